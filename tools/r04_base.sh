#!/bin/bash
# round-4 baseline of the tree as round 3 left it: default bench line + ViT-L/16-384 fp16 parity on 16 images
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 400 python bench.py > $out/a_bench_default.json 2> $out/a_bench_default.err || exit 1
cut -c1-400 $out/a_bench_default.json
PARITY_CONFIG=vit_large_384 PARITY_N=16 PARITY_DTYPES=fp16 timeout -k 10 500 python tools/parity_stats.py > $out/a_parity_vitl.txt 2>&1
PARITY_CONFIG=vit_large_384 PARITY_N=16 PARITY_DTYPES=fp16 VH_RESID_SPLIT=0 timeout -k 10 500 python tools/parity_stats.py >> $out/a_parity_vitl.txt 2>&1
cat $out/a_parity_vitl.txt
