# attention A/B: operator tests, then the stage tables of ViT-B/16 b512 bf16 and ViT-L/16-384 b256 fp16
set -e
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -k "attention" > gpurun_out/attn_tests.log 2>&1 || { tail -15 gpurun_out/attn_tests.log; exit 1; }
tail -1 gpurun_out/attn_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --stages 2>&1 | grep -E "attention|total|\"value\"" | cut -c1-110
timeout -k 10 300 python bench.py --config vit_large_384 --dtype fp16 --batch 256 --steps 5 --warmup 1 --no-cpu-baseline --stages 2>&1 | grep -E "attention|total|\"value\"" | cut -c1-110
