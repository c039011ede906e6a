#!/bin/bash
# round 4, call e: 4-wave multi-slab ring (one wave per SIMD): parity, then timing at T = 384 (NS 3), 512 (NS 4), 577 (NS 5, spills) vs the ring form
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > $out/e_attn_tests.log 2>&1; echo "attention tests rc=$?"; tail -3 $out/e_attn_tests.log
for T in 384 512 577; do for ms in 0 1 0 1; do echo -n "VH_ATTN_MS=$ms "; VH_ATTN_MS=$ms timeout -k 10 200 python tools/attn_bench.py --config vit_large_384 --batch 256 --dtype fp16 --tokens $T 2>&1 | tail -1; done; done > $out/e_attn_ms_ab.txt 2>&1
cat $out/e_attn_ms_ab.txt
