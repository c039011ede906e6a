#!/bin/bash
# round 4, call n: tiled hidden activation, longer A/B (VH_H_TILED=0|1, eight interleaved pairs of 40 steps), bf16 then fp16
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
for dt in bf16 fp16; do for i in 1 2 3 4 5 6 7 8; do for t in 0 1; do
  echo -n "$dt VH_H_TILED=$t: "; VH_H_TILED=$t timeout -k 10 200 python bench.py $NOX --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done; done > $out/n_h_tiled_ab_long.txt 2>&1
python - <<'PY'
import re,collections
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/n_h_tiled_ab_long.txt'):
    m=re.match(r'(\w+) VH_H_TILED=(\d): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[(m.group(1),m.group(2))].append((float(m.group(3)),float(m.group(5))))
import statistics as st
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), 'mean', round(st.mean(x[0] for x in v[k]),1), 'fc1 ms median', st.median(x[1] for x in v[k]), 'n', len(v[k]))
PY
