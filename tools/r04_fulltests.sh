#!/bin/bash
# the whole GPU suite + smoke() in one call (tail of the log -> gpurun_out/r04/full_gpu_tests.log)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > $O/full_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -8 $O/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
