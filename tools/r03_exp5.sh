#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
SECONDS=0
python bench.py > $O/e5_bench_default.json 2> $O/e5_bench_default.err; echo "bench rc=$? wall ${SECONDS}s"; tail -3 $O/e5_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/e5_bench_default.json").read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict): print(k, json.dumps(v)[:700])
    else: print(k, v)
PY
