#!/bin/bash
# round 4, call b: the guard tests (calibration at load, fp8 max|x|, graphs after a trip) + ViT-L/16-384 fp16 parity on 64 images, both LayerNorm paths
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_fp8.py -x -q -s -k "common_mode or backstop or tripped_guard or massive or outlier or fp8_emulation" > $out/b_guard_tests.log 2>&1; echo "guard tests rc=$?"; tail -5 $out/b_guard_tests.log
PARITY_CONFIG=vit_large_384 PARITY_N=64 PARITY_DTYPES=fp16 timeout -k 10 500 python tools/parity_stats.py > $out/b_parity_vitl64.txt 2>&1
PARITY_FOLD=off PARITY_CONFIG=vit_large_384 PARITY_N=64 PARITY_DTYPES=fp16 timeout -k 10 500 python tools/parity_stats.py >> $out/b_parity_vitl64.txt 2>&1
cat $out/b_parity_vitl64.txt
