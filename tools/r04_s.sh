#!/bin/bash
# round 4, call s: split-residual epilogue with its hi / lo addends in flight several passes ahead (buffer instructions: one per-lane
# offset for all lines): libraries with VH_RS_DEPTH = 1 (product) 2 3 4 5, interleaved; logits hash of each; stage tables
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
python -m pytest tests/test_gpu_vit.py tests/test_gpu_ops.py -x -q -m gpu > $out/s_tests.txt 2>&1 || { tail -15 $out/s_tests.txt; exit 1; }
tail -2 $out/s_tests.txt
for lib in libvithip_abl_old.so libvithip_abl_rs1.so libvithip_abl_rs2.so libvithip_abl_rs3.so libvithip.so libvithip_abl_rs5.so; do
  for dt in bf16 fp16; do echo -n "$lib $dt: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 40 --every 20 --dtype $dt 2>&1 | tail -1; done
  echo -n "$lib vit_large b16 bf16: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 4 --every 2 --batch 16 --config vit_large_384 2>&1 | tail -1
done > $out/s_hashes.txt 2>&1
cat $out/s_hashes.txt
for i in 1 2 3 4; do for lib in libvithip_abl_old.so libvithip_abl_rs1.so libvithip_abl_rs2.so libvithip_abl_rs3.so libvithip.so libvithip_abl_rs5.so; do
  echo -n "$lib: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/s_depth.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/s_depth.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in v: print(k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1 ms median', st.median(x[1] for x in v[k]), 'n', len(v[k]))
PY
for lib in libvithip_abl_old.so libvithip_abl_rs2.so libvithip.so; do VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/s_stages_$lib.txt > /dev/null; done
grep -E "qkv|fc1|fc2|proj" $out/s_stages_*.txt
