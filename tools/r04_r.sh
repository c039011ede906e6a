#!/bin/bash
# round 4, call r: start stagger of the persistent GEMM's workgroups with slack (VH_PP_STAGGER="ticks per K-tile, ticks (fc1 / q|k|v), ticks (residual GEMMs)",
# 10 ns ticks): do the epilogue bursts of 256 lock-stepped workgroups cost HBM time that a half-period offset gives back?
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
python -m pytest tests/test_gpu_vit.py -x -q -m gpu -k "tiled or bits or parity" > $out/r_tests.txt 2>&1 || { tail -5 $out/r_tests.txt; exit 1; }
VH_PP_STAGGER=87,300,600 python -m pytest tests/test_gpu_vit.py -x -q -m gpu -k "tiled or bits" > $out/r_tests_stag.txt 2>&1 || { tail -5 $out/r_tests_stag.txt; exit 1; }
SET="0,0,0 87,250,600 58,170,400 120,350,800 87,250,0 0,0,600 40,100,300"
for i in 1 2 3; do for t in $SET; do
  echo -n "VH_PP_STAGGER=$t: "; VH_PP_STAGGER=$t timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/r_stagger.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/r_stagger.txt'):
    m=re.match(r'VH_PP_STAGGER=(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in v: print('VH_PP_STAGGER='+k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1 ms median', st.median(x[1] for x in v[k]), 'n', len(v[k]))
PY
for t in 0,0,0 87,250,600; do VH_PP_STAGGER=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/r_stages_$t.txt > /dev/null; done
grep -E "qkv|fc1|fc2|proj" $out/r_stages_0,0,0.txt $out/r_stages_87,250,600.txt
