# tail overlap A/B: parity tests first, then the default bench with the overlap off / on, bf16 and fp8
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_fp8.py -x -q > gpurun_out/tail_tests.log 2>&1 || { tail -20 gpurun_out/tail_tests.log; exit 1; }
tail -1 gpurun_out/tail_tests.log
for t in 0 1 0 1; do
  echo "== VH_TAIL_OVERLAP=$t bf16"; VH_TAIL_OVERLAP=$t timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo
done
for t in 0 1; do
  echo "== VH_TAIL_OVERLAP=$t fp8"; VH_TAIL_OVERLAP=$t timeout -k 10 200 python bench.py --no-cpu-baseline --dtype fp8 2>&1 | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | tr "\n" " "; echo
done
