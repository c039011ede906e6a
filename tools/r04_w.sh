#!/bin/bash
# round 4, call w: where is fp16 slower than bf16?  stage tables of both, twice, interleaved
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
for i in 1 2; do for dt in bf16 fp16; do
  timeout -k 10 200 python bench.py --dtype $dt --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/w_stages_${dt}_$i.txt > $out/w_line_${dt}_$i.json
  echo "$dt $i: $(python -c "import json; d=json.load(open('$out/w_line_${dt}_$i.json')); print(d['value'])") $(grep -E 'qkv_gemm|attention|proj_gemm|fc1_gemm|fc2_gemm|ln_stats' $out/w_stages_${dt}_$i.txt | awk '{printf "%s %s  ", $1, $2}')"
done; done | tee $out/w_fp16_vs_bf16_stages.txt
