#!/bin/bash
# round-3 experiment 15: would a one-byte lo plane (6 instead of 8 bytes per element and residual update) shorten the split-residual epilogues?
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
L=$PWD/vit-fpga_amd
for v in "" _abl_lo8 "" _abl_lo8; do
  echo "## libvithip$v.so" | tee -a $O/e15_lo8.txt
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 200 python tools/torch_matmul_calib.py --no-torch --rounds 2 --iters 200 2>&1 | grep -v "^#" | grep "proj\|fc2" | tee -a $O/e15_lo8.txt || exit 1
done
for v in "" _abl_lo8; do
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench$v', d['value'], d['ms_per_step'], flush=True)" | tee -a $O/e15_lo8.txt
done
