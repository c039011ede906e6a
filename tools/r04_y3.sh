#!/bin/bash
# round 4, call y3: fp8 contexts, q|k|v head-major as well (VH_QKV_HM): bits, then interleaved A/B of config 5
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "tiled or deterministic or logits_track or full_size" > $out/y3_tests.txt 2>&1 || { tail -20 $out/y3_tests.txt; exit 1; }
tail -1 $out/y3_tests.txt
for t in 0 1; do echo -n "VH_QKV_HM=$t fp8: "; VH_QKV_HM=$t timeout -k 10 120 python tools/soak.py --steps 40 --every 20 --dtype fp8 2>&1 | tail -1; done | tee $out/y3_hashes.txt
for i in 1 2 3 4 5; do for t in 0 1; do
  echo -n "hm$t: "; VH_QKV_HM=$t timeout -k 10 200 python bench.py --dtype fp8 --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $out/y3_fp8_att_tiled.txt 2>&1
python - <<'PY'
import re,collections,statistics as st
v=collections.defaultdict(list)
for l in open('gpurun_out/r04/y3_fp8_att_tiled.txt'):
    m=re.match(r'(\S+): ([\d.]+) ([\d.]+) ([\d.]+)',l)
    if m: v[m.group(1)].append((float(m.group(2)),float(m.group(4))))
for k in sorted(v): print(k, 'images/s median', st.median(x[0] for x in v[k]), [x[0] for x in v[k]])
PY
for t in 0 1; do VH_QKV_HM=$t timeout -k 10 200 python bench.py --dtype fp8 --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/y3_stages_$t.txt > /dev/null; echo "hm$t: $(grep -E 'attention|proj_gemm' $out/y3_stages_$t.txt | awk '{printf "%s %s  ", $1, $2}')"; done
