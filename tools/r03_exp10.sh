#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for st in 0 2 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity --streams $st | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $st, d['value'], d['ms_per_step'], d['roofline']['frac'] if d['roofline'] else None, d['roofline']['avg_launch_ms'] if d['roofline'] else None, flush=True)"
done; done
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --backend gloo --same-device --batch 128 --no-cpu-baseline --no-extra-configs 2>/dev/null | cut -c1-400
