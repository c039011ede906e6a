#!/bin/bash
# round 4, call p: attention output tiled -> tiled out-projection (VH_ATT_TILED=0|1; h tiled in both): bitwise tests, attention tests, A/B
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_ops.py -x -q -k "tiled_hidden or attention or same_bits or full_size_config_4" > $out/p_tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/p_tests.log
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40"
for dt in bf16 fp16; do for i in 1 2 3 4 5 6; do for t in 0 1; do
  echo -n "$dt VH_ATT_TILED=$t: "; VH_ATT_TILED=$t timeout -k 10 200 python bench.py $NOX --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done; done > $out/p_att_tiled_ab.txt 2>&1
for t in 0 1; do VH_ATT_TILED=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/p_stages_att$t.txt > /dev/null; done
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/p_att_tiled_ab.txt'):
    m=re.match(r'(\w+) VH_ATT_TILED=(\d): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[(m.group(1),m.group(2))].append(float(m.group(3)))
for k in sorted(v): print(k, 'images/s median', st.median(v[k]), 'mean', round(st.mean(v[k]),1), 'n', len(v[k]))
PY
grep -E "attention|proj" $out/p_stages_att0.txt $out/p_stages_att1.txt
