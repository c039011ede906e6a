#!/bin/bash
# round 4, call u: last-round tiles spread over the XCDs (blockIdx < rem) + start stagger of the workgroups without one
# (VH_PP_STAGGER="ticks per K-tile, ticks fc1 / q|k|v, ticks residual GEMMs", 10 ns ticks): hashes, then interleaved bench runs
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "gemm" > $out/u_tests.txt 2>&1 || { tail -15 $out/u_tests.txt; exit 1; }
tail -1 $out/u_tests.txt
for st in 0,0,0 87,250,600; do for b in 512 16; do
  echo -n "stagger $st b$b: "; VH_PP_STAGGER=$st timeout -k 10 120 python tools/soak.py --steps 4 --every 2 --batch $b 2>&1 | tail -1
done; done 2>&1 | tee $out/u_hashes.txt
SET="0,0,0 87,250,600 58,170,400 120,350,800 0,0,600 87,0,600 30,0,200"
for i in 1 2 3; do
  echo -n "old: "; VITHIP_LIB=$PWD/$L/libvithip_abl_old.so timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  for t in $SET; do
  echo -n "$t: "; VH_PP_STAGGER=$t timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/u_stagger.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/u_stagger.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in v: print(k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1 ms median', st.median(x[1] for x in v[k]), 'n', len(v[k]))
PY
for t in 0,0,0 87,250,600 120,350,800; do VH_PP_STAGGER=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/u_stages_$t.txt > /dev/null; done
VITHIP_LIB=$PWD/$L/libvithip_abl_old.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/u_stages_old.txt > /dev/null
grep -E "qkv|fc1|fc2|proj" $out/u_stages_*.txt
