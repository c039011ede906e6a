#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "patch_split or resid_split or persistent" > $O/e9_ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/e9_ops.log
timeout -k 10 1000 python -m pytest tests/test_gpu_vit.py tests/test_gpu_group.py tests/test_host_cpp.py -x -q > $O/e9_vit.log 2>&1; echo "vit rc=$?"; tail -5 $O/e9_vit.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --stages 2>&1 | cut -c1-260
