#!/bin/bash
# round-3 experiment 3: what the 16-bit epilogues cost, by ablation (timing-only builds), then the GPU test suite
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
L=$PWD/vit-fpga_amd
for v in "" _abl_e16 _abl_e32 _abl_e48 _abl_e64 ""; do
  echo "## libvithip$v.so" | tee -a $O/e3_abl.txt
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 200 python tools/torch_matmul_calib.py --no-torch --rounds 2 2>&1 | grep -v "^#" | tee -a $O/e3_abl.txt || exit 1
done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/e3_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/e3_gpu_tests.log
