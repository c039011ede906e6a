#!/bin/bash
# round 4, call d: multi-slab attention ring (long sequences): parity tests, A/B against the ring form at ViT-L/16-384 b256
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $out/d_attn_tests.log 2>&1; echo "attention tests rc=$?"; tail -3 $out/d_attn_tests.log
for ms in 0 1 0 1; do echo "== VH_ATTN_MS=$ms"; VH_ATTN_MS=$ms timeout -k 10 200 python tools/attn_bench.py --config vit_large_384 --batch 256 --dtype fp16 2>&1 | head -2; done > $out/d_attn_ms_ab.txt 2>&1
cat $out/d_attn_ms_ab.txt
