#!/bin/bash
# round 4, call ae: fp8 contexts, the q|k|v epilogue's constants (+ weight scales) prefetched through LDS-DMA (asm DMAs with scalar bases):
# libvithip.so against -DVH_PP_CPRE=0; fp8 tests, hashes, interleaved bench of config 5
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "gemm or tiled or deterministic or full_size" > $out/ae_tests.txt 2>&1 || { tail -20 $out/ae_tests.txt; exit 1; }
tail -1 $out/ae_tests.txt
for lib in libvithip_abl_nocpre.so libvithip.so; do echo -n "$lib fp8: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 100 --every 20 --dtype fp8 2>&1 | tail -1; done | tee $out/ae_hashes.txt
for i in 1 2 3 4 5; do for lib in libvithip_abl_nocpre.so libvithip.so; do
  echo -n "$lib: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --dtype fp8 --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/ae_fp8_cpre.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/ae_fp8_cpre.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), [x[0] for x in v[k]])
PY
for lib in libvithip_abl_nocpre.so libvithip.so; do VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --dtype fp8 --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/ae_stages_$lib.txt > /dev/null; echo "$lib: $(grep -E 'qkv_gemm|fc1_gemm' $out/ae_stages_$lib.txt | awk '{printf "%s %s  ", $1, $2}')"; done
