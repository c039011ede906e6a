#!/bin/bash
# round 4, call aa: TIMING-ONLY -- the 16-bit GELU epilogue (tiled fc1) and every other epilogue without the loads of their constants
# (row statistics, bias, c): the upper bound of what prefetching them through LDS during the K loop could buy (garbage logits)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
for i in 1 2 3; do for lib in libvithip.so libvithip_abl_noconst.so; do
  VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/aa_stages_$lib.txt > /dev/null
  echo "$lib: $(grep -E 'qkv_gemm|fc1_gemm|fc2_gemm|proj_gemm' $out/aa_stages_$lib.txt | awk '{printf "%s %s  ", $1, $2}')"
done; done | tee $out/aa_epilogue_constants_ablation.txt
