#!/bin/bash
# round-3 experiment 19: VH_FLAG_CLS_TAIL (opt-in): tests, then the default bench line with its cls_tail object
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -x -q -k "class_token_tail or error_reporting or logits_match_oracle" > $O/e19_tests.log 2>&1; rc=$?; tail -5 $O/e19_tests.log; grep "cls tail" $O/e19_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/e19_bench.json 2> $O/e19_bench.err; python -c "
import json
d=json.loads(open('$O/e19_bench.json').read().strip().splitlines()[-1]); print('default', d['value'], 'cls_tail', d['cls_tail_bf16_b512']['value'], d['cls_tail_bf16_b512']['parity']['worst'], d['cls_tail_bf16_b512']['parity']['median'], 'default parity', d['parity']['worst'], d['parity']['median'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity --cls-tail --stages 2>&1 >/dev/null | tail -13
