# super-column tile order A/B (VH_PP_SN): GEMM op tests, then stage times for sn = 0 (n fastest), 2, 3, 4, 6
set -e
VH_PP_SN=3 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > gpurun_out/sn_tests.log 2>&1 || { tail -15 gpurun_out/sn_tests.log; exit 1; }
tail -1 gpurun_out/sn_tests.log
for sn in 0 2 3 4 6 0; do
  echo -n "sn=$sn "; VH_PP_SN=$sn timeout -k 10 200 python bench.py --no-cpu-baseline --stages 2>&1 | grep -E "\"value\"|qkv_gemm|fc1_gemm|fc2_gemm|proj_gemm" | sed "s/launches.*//" | tr "\n" " " | sed "s/{\"metric.*\"value\": \([0-9.]*\).*/ value \1/"; echo
done
