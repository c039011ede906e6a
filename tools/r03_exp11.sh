#!/bin/bash
# round-3 experiment 11: what the K loop of the persistent GEMM consists of, by ablation of the main loop (timing-only
# builds, no epilogue in any of them: VH_EPI_ABL=64): m0 = as is, m2 = no fragment reads, m1 = no MFMAs, m3 = neither
# (DMA stream + counted waits + barriers only).  Launch times (calib tool) and, from the stamped builds, cycles per
# K-tile and the clock each variant runs at.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
L=$PWD/vit-fpga_amd
for v in e64 m2 m1 m3 e64; do
  echo "## libvithip_abl_$v.so" | tee -a $O/e11_abl.txt
  VITHIP_LIB=$L/libvithip_abl_$v.so timeout -k 10 200 python tools/torch_matmul_calib.py --no-torch --rounds 2 2>&1 | grep -v "^#" | tee -a $O/e11_abl.txt || exit 1
done
for v in m0 m2 m1 m3; do
  echo "## libvithip_diag_$v.so" | tee -a $O/e11_anat.txt
  VITHIP_LIB=$L/libvithip_diag_$v.so timeout -k 10 200 python tools/gemm_anatomy.py --seconds 2 2>&1 | grep -v "^      \|^        " | tee -a $O/e11_anat.txt || exit 1
done
