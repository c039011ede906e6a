#!/bin/bash
# round-3 experiment 12: main-loop ablation with the clock of each variant (tools/mainloop_ablation.py)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
L=$PWD/vit-fpga_amd
for v in "m0:as is (no epilogue)" "m4:fragments read once" "m8:no DMA in the loop" "m12:MFMAs only (real operands)" "m2:no fragment reads" "m1:no MFMAs" "m3:DMA stream only" "m0:as is (no epilogue)"; do
  VITHIP_LIB=$L/libvithip_diag_${v%%:*}.so timeout -k 10 200 python tools/mainloop_ablation.py --label "${v#*:}" --shape fc1,qkv 2>&1 | tee -a $O/e12_mainloop.txt || exit 1
done
