set -e
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -k "gemm" > gpurun_out/v7_tests.log 2>&1 || { tail -15 gpurun_out/v7_tests.log; exit 1; }
tail -2 gpurun_out/v7_tests.log
for v in 5 7; do
  echo "== VH_GEMM_PP=$v bf16"; VH_GEMM_PP=$v timeout -k 10 200 python bench.py --no-cpu-baseline --stages 2>&1 | grep -E "qkv_gemm|proj_gemm|fc1_gemm|fc2_gemm|\"value\"" | cut -c1-120
done
for v in 5 7; do
  echo "== VH_GEMM_PP=$v fp8"; VH_GEMM_PP=$v timeout -k 10 200 python bench.py --no-cpu-baseline --dtype fp8 --stages 2>&1 | grep -E "qkv_gemm|proj_gemm|fc1_gemm|fc2_gemm|\"value\"" | cut -c1-120
done
