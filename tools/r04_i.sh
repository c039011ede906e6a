#!/bin/bash
# round 4, call i: patch embedding with the gather inside the GEMM (VH_PATCH_FUSED=1): bitwise test, A/B of the forward and of the stages
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -x -q -s -k "gather_inside or vit_large_384_fp16_on_16" > $out/i_tests.log 2>&1; echo "tests rc=$?"; grep -E "parity|passed|failed" $out/i_tests.log | tail -4
NOX="--no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs"
for i in 1 2 3; do for f in 0 1; do
  echo -n "VH_PATCH_FUSED=$f: "; VH_PATCH_FUSED=$f timeout -k 10 200 python bench.py $NOX --stages 2> $out/i_stages_fused$f.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done > $out/i_patch_fused_ab.txt 2>&1
cat $out/i_patch_fused_ab.txt
for f in 0 1; do echo "== VH_PATCH_FUSED=$f"; grep -iE "im2col|patch|cls|stat" $out/i_stages_fused$f.txt | head -6; done
