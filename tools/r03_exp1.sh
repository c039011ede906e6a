#!/bin/bash
# round-3 experiment 1: register-path 16-bit epilogue + LDS-prefetched epilogue vectors (A) against the staged form (B)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03; mkdir -p $O
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q > $O/e1_ops_tests.log 2>&1; echo "ops tests rc=$?" | tee -a $O/e1_status.txt
tail -3 $O/e1_ops_tests.log
for r in 1 2; do
  VITHIP_LIB=$PWD/vit-fpga_amd/libvithip_b.so timeout -k 10 300 python tools/torch_matmul_calib.py --no-torch >> $O/e1_calib_b.txt 2>&1 || exit 1
  timeout -k 10 300 python tools/torch_matmul_calib.py --no-torch >> $O/e1_calib_a.txt 2>&1 || exit 1
done
cat $O/e1_calib_b.txt $O/e1_calib_a.txt
VITHIP_LIB=$PWD/vit-fpga_amd/libvithip_b.so timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/e1_bench_b.json 2> $O/e1_bench_b.err || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/e1_bench_a.json 2> $O/e1_bench_a.err || exit 1
python - <<'PY'
import json
for n in ("b","a"):
    d=json.loads(open(f"gpurun_out/r03/e1_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"], d.get("parity"))
PY
VITHIP_LIB=$PWD/vit-fpga_amd/libvithip_diag.so timeout -k 10 300 python tools/gemm_anatomy.py > $O/e1_anatomy.txt 2>&1; head -60 $O/e1_anatomy.txt
