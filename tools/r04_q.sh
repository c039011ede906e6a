#!/bin/bash
# round 4, call q: TIMING-ONLY -- q|k|v and fc1 read both operands through the tiled DMA path (VH_X_TILED_EXP=1; the operands are
# NOT tiled, the logits of that mode are garbage): what a tiled residual hi plane could buy
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
for i in 1 2 3 4 5 6; do for t in 0 1; do
  echo -n "VH_X_TILED_EXP=$t: "; VH_X_TILED_EXP=$t timeout -k 10 200 python bench.py $NOX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/q_x_tiled_exp.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/q_x_tiled_exp.txt'):
    m=re.match(r'VH_X_TILED_EXP=(\d): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print('VH_X_TILED_EXP='+k, 'images/s median', st.median(x[0] for x in v[k]), 'fc1 ms median', st.median(x[1] for x in v[k]), 'n', len(v[k]))
PY
for t in 0 1; do VH_X_TILED_EXP=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/q_stages$t.txt > /dev/null; done
grep -E "qkv|fc1" $out/q_stages0.txt $out/q_stages1.txt
