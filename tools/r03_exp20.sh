#!/bin/bash
# batch-size sweep of the default configuration (bf16 and fp16), short runs: images/s and ms per forward
cd "${GRAFT_REPO_ROOT:-.}"
for dt in bf16 fp16; do for b in 1 8 32 64 128 256 384 512 768 1024; do
  st=$(( b >= 256 ? 20 : 100 ))
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity --dtype $dt --batch $b --steps $st --warmup 5 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt batch %5d  %9.1f images/s  %8.3f ms per forward  (%.1f %% of the 2.5 PF nominal peak over the whole forward)' % ($b, d['value'], d['ms_per_step'], 100*d['forward_mfma_frac']), flush=True)"
done; done
