#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit.py tests/test_gpu_fp8.py -x -q > $O/e7_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/e7_tests.log
timeout -k 10 200 python tools/attn_bench.py > $O/e7_attn_bench.txt 2>&1; timeout -k 10 200 python tools/attn_bench.py --config vit_large_384 --batch 256 --dtype fp16 >> $O/e7_attn_bench.txt 2>&1; cat $O/e7_attn_bench.txt
VITHIP_LIB=$PWD/vit-fpga_amd/libvithip_diag.so timeout -k 10 100 python tools/attn_anatomy.py > $O/e7_attn_anatomy.txt 2>&1; head -12 $O/e7_attn_anatomy.txt
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --no-parity | cut -c1-200; done
