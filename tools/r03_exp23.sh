#!/bin/bash
# which scale for the lo byte?  64-image ViT-B statistics and 4-image ViT-L/384 parity for three pairs of scales (A/B libraries)
cd "${GRAFT_REPO_ROOT:-.}"
L=$PWD/vit-fpga_amd
for v in "" _b_s512 _b_s1024; do
  echo "## libvithip$v.so"
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 400 python tools/parity_stats.py 2>&1 | grep "fold=default" | grep -v fp8 | cut -c1-150
  VITHIP_LIB=$L/libvithip$v.so timeout -k 10 400 python bench.py --no-cpu-baseline --no-extra-configs --no-fp16-line --config vit_large_384 --dtype fp16 --batch 256 --steps 3 --warmup 1 --parity-images 4 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vit_large_384 fp16, 4 images: worst', d['parity']['worst'], 'median', d['parity']['median'], flush=True)"
done
