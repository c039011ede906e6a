#!/bin/bash
# small batches: eager launches vs graph replay
cd "${GRAFT_REPO_ROOT:-.}"
for b in 1 4 16 64; do for g in "" "--graph"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity --batch $b --steps 200 --warmup 20 $g | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b $g', d['value'], d['ms_per_step'], flush=True)"
done; done
