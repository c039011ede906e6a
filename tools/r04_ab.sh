#!/bin/bash
# round 4, call ab: the LN-fold epilogues' constants (row statistics, d, c) prefetched into the wave's staging slice by LDS-DMA during the K loop
# (libvithip.so) against loading them inside the epilogue (libvithip_abl_nocpre.so = -DVH_PP_CPRE=0): tests, hashes, interleaved bench
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > $out/ab_tests.txt 2>&1 || { tail -15 $out/ab_tests.txt; exit 1; }
tail -1 $out/ab_tests.txt
python -m pytest tests/test_gpu_vit.py -x -q -m gpu -k "tiled or bits or parity" > $out/ab_tests2.txt 2>&1 || { tail -15 $out/ab_tests2.txt; exit 1; }
tail -1 $out/ab_tests2.txt
for lib in libvithip_abl_nocpre.so libvithip.so; do for dt in bf16 fp16; do echo -n "$lib $dt: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 100 --every 20 --dtype $dt 2>&1 | tail -1; done; done | tee $out/ab_hashes.txt
for i in 1 2 3 4 5; do for lib in libvithip_abl_nocpre.so libvithip.so; do for dt in bf16 fp16; do
  echo -n "$lib-$dt: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --dtype $dt $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done; done > $out/ab_cpre.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/ab_cpre.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1 ms median', st.median(x[1] for x in v[k]), [x[0] for x in v[k]])
PY
for lib in libvithip_abl_nocpre.so libvithip.so; do VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/ab_stages_$lib.txt > /dev/null; echo "$lib: $(grep -E 'qkv_gemm|fc1_gemm' $out/ab_stages_$lib.txt | awk '{printf "%s %s  ", $1, $2}')"; done
