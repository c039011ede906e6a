#!/bin/bash
# round 4, call t: TIMING-ONLY ablations of the split-residual epilogue (general form): what do its loads, its stores, its row partials cost?
# VH_EPI_ABL: 1 no partial-sum stores, 4 no plane stores, 8 no plane loads (garbage logits)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
for lib in libvithip_abl_old.so libvithip_abl_ea1.so libvithip_abl_ea4.so libvithip_abl_ea8.so libvithip_abl_ea12.so libvithip_abl_ea5.so libvithip_abl_ea13.so; do
  VITHIP_LIB=$PWD/$L/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-fp16-line --no-extra-configs --stages 2> $out/t_stages_$lib.txt > /dev/null
  echo "$lib: $(grep -E 'proj_gemm|fc2_gemm' $out/t_stages_$lib.txt | awk '{printf "%s %s ms  ", $1, $2}')"
done | tee $out/t_resid_epilogue_ablation.txt
