#!/bin/bash
# round-3 experiment 18: the fp8 path's residual as an e4m3 plane (the operand) + a bf16 plane
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_fp8.py -x -q > $O/e18_fp8.log 2>&1; rc=$?; tail -5 $O/e18_fp8.log; [ $rc = 0 ] || exit 1
for s in 0 1; do
  VH_RESID_SPLIT=$s timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --dtype fp8 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp8 split=$s', d['value'], d['ms_per_step'], d['parity']['worst'], d['parity']['median'], flush=True)" | tee -a $O/e18_bench.txt
done
VH_RESID_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity --dtype fp8 --stages 2>&1 >/dev/null | tail -13 > $O/e18_stages_split0.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity --dtype fp8 --stages 2>&1 >/dev/null | tail -13 > $O/e18_stages_split1.txt
paste $O/e18_stages_split0.txt $O/e18_stages_split1.txt | cut -c1-160
