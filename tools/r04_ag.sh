#!/bin/bash
# round 4, call ag: split-residual epilogue with its bias and first-pass planes prefetched through LDS-DMA (libvithip.so) against -DVH_PP_RPRE=0; was: whole-tile (counted waits per line instead of vmcnt(0) at
# every other pass): unit tests, hashes, interleaved A/B against the previous library
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split or resid or patch or persistent" > $out/ag_tests.txt 2>&1 || { tail -15 $out/ag_tests.txt; exit 1; }
tail -1 $out/ag_tests.txt
for lib in libvithip_abl_norpre.so libvithip.so; do for dt in bf16 fp16; do echo -n "$lib $dt: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 40 --every 20 --dtype $dt 2>&1 | tail -1; done; done | tee $out/ag_hashes.txt
for i in 1 2 3 4 5; do for lib in libvithip_abl_norpre.so libvithip.so; do
  echo -n "$lib: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/ag_resid_nobranch.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/ag_resid_nobranch.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), [x[0] for x in v[k]])
PY
for lib in libvithip_abl_norpre.so libvithip.so; do VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/ag_stages_$lib.txt > /dev/null; echo "$lib: $(grep -E 'proj_gemm|fc2_gemm|patch_gemm' $out/ag_stages_$lib.txt | awk '{printf "%s %s  ", $1, $2}')"; done
