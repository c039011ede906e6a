#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
L=$PWD/vit-fpga_amd
for r in 1 2 3; do for v in "" _abl_prio; do
  echo "## libvithip$v.so" | tee -a $O/e8_attn_prio.txt
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep "in the forward" | tee -a $O/e8_attn_prio.txt
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 200 python tools/attn_bench.py --config vit_large_384 --batch 256 --dtype fp16 2>&1 | grep "in the forward" | tee -a $O/e8_attn_prio.txt
done; done
