#!/bin/bash
# cache policy of the attention kernel's K/V/Q LDS-DMA loads (read-once data): default / nt / sc1
cd "${GRAFT_REPO_ROOT:-.}"
L=$PWD/vit-fpga_amd
for r in 1 2; do for v in "" _b_nt _b_sc1; do
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('policy[$v]', d['value'], d['ms_per_step'], flush=True)"
done; done
VITHIP_LIB=$L/libvithip_b_nt.so timeout -k 10 100 python tools/attn_bench.py 2>&1 | tail -2
timeout -k 10 100 python tools/attn_bench.py 2>&1 | tail -2
