#!/bin/bash
# round-3 experiment 6: super-column width of the persistent tile order, per GEMM shape (launch time, same box)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
for r in 1 2; do
for sn in -1 0 2 3 4 6 9; do
  echo "## VH_PP_SN=$sn" | tee -a $O/e6_supercol.txt
  VH_PP_SN=$sn timeout -k 10 200 python tools/torch_matmul_calib.py --no-torch --rounds 2 2>&1 | grep "qkv\|fc1" | tee -a $O/e6_supercol.txt || exit 1
done
done
