#!/bin/bash
# round 4, call y: e4m3 hidden activation tiled (fp8 contexts, VH_H_TILED): bits, then interleaved A/B of config 5
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > $out/y_tests.txt 2>&1 || { tail -20 $out/y_tests.txt; exit 1; }
tail -1 $out/y_tests.txt
for t in 0 1; do echo -n "VH_H_TILED=$t fp8: "; VH_H_TILED=$t timeout -k 10 120 python tools/soak.py --steps 40 --every 20 --dtype fp8 2>&1 | tail -1; done | tee $out/y_hashes.txt
for i in 1 2 3 4 5; do for t in 0 1; do
  echo -n "tiled$t: "; VH_H_TILED=$t timeout -k 10 200 python bench.py --dtype fp8 --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/y_fp8_tiled.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/y_fp8_tiled.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1 ms median', st.median(x[1] for x in v[k]), [x[0] for x in v[k]])
PY
for t in 0 1; do VH_H_TILED=$t timeout -k 10 200 python bench.py --dtype fp8 --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/y_stages_$t.txt > /dev/null; echo "tiled$t: $(grep -E 'fc1_gemm|fc2_gemm' $out/y_stages_$t.txt | awk '{printf "%s %s  ", $1, $2}')"; done
