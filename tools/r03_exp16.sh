#!/bin/bash
# round-3 experiment 16: the split residual's lo plane as one scaled e4m3 byte per element (6 instead of 8 bytes per element and update)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q > $O/e16_ops.log 2>&1; rc=$?; tail -5 $O/e16_ops.log; [ $rc = 0 ] || exit 1
timeout -k 10 400 python tools/parity_stats.py > $O/e16_parity.txt 2>&1; tail -12 $O/e16_parity.txt
for r in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['parity']['worst'], d['parity']['median'], 'fp16', d['fp16']['value'], d['fp16']['parity']['worst'], flush=True)" | tee -a $O/e16_bench.txt
done
timeout -k 10 200 python tools/torch_matmul_calib.py --no-torch --rounds 2 --iters 200 2>&1 | tee $O/e16_calib.txt
