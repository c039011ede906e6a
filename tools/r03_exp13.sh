#!/bin/bash
# round-3 experiment 13: the MFMA + barrier skeleton of the K loop on its own (no memory operation in the loop)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
L=$PWD/vit-fpga_amd
for v in "m12:MFMAs only (real operands)" "m28:MFMAs only, no barriers" "m44:MFMAs only, no s_setprio" "m12:MFMAs only (real operands)"; do
  VITHIP_LIB=$L/libvithip_diag_${v%%:*}.so timeout -k 10 200 python tools/mainloop_ablation.py --label "${v#*:}" --shape fc1,qkv 2>&1 | tee -a $O/e13_skeleton.txt || exit 1
done
