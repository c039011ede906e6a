#!/bin/bash
# round-3 experiment 14: one memset of the attention work-queue counters per forward instead of one per launch
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_fp8.py tests/test_gpu_group.py -x -q > $O/e14_tests.log 2>&1; rc=$?; tail -3 $O/e14_tests.log; [ $rc = 0 ] || exit 1
for r in 1 2 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['step_ms'], flush=True)"
done
