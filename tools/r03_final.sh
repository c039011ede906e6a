#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out; tag=r03f
timeout -k 10 400 python bench.py > $out/${tag}_bench_default_final.json 2> $out/${tag}_bench_default_final.err || exit 1
timeout -k 10 200 python tools/attn_bench.py > $out/${tag}_attn_bench.txt 2>&1 || exit 1
timeout -k 10 200 python tools/attn_bench.py --config vit_large_384 --batch 256 --dtype fp16 >> $out/${tag}_attn_bench.txt 2>&1
cat $out/${tag}_attn_bench.txt; cut -c1-300 $out/${tag}_bench_default_final.json
