#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit.py -x -q -s -k "split or massive or rowstats or persistent_walks or 64_vit_b or outlier or class_token" > $O/e22_tests.log 2>&1; rc=$?; grep "massive\]\|fold\]\|outliers\]\|cls tail\]\|passed\|failed" $O/e22_tests.log | tail -12; [ $rc = 0 ] || { tail -30 $O/e22_tests.log; exit 1; }
timeout -k 10 400 python tools/parity_stats.py 2>&1 | grep "fold=default" 
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['parity']['worst'], d['parity']['median'], 'fp16', d['fp16']['value'], d['fp16']['parity']['worst'], d['fp16']['parity']['median'])"
