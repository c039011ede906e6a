#!/bin/bash
# round 4, call s0: which form changed its bits?  unit tests of the split-residual epilogue, then logits hashes old / new at batch 512, 2, 16
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
L=vit-fpga_amd
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "split or resid" > $out/s0_tests.txt 2>&1; tail -12 $out/s0_tests.txt
for lib in libvithip_abl_old.so libvithip.so libvithip_abl_rs1.so libvithip_abl_rs5.so; do for b in 512 2 16; do
  echo -n "$lib b$b: "; VITHIP_LIB=$PWD/$L/$lib timeout -k 10 120 python tools/soak.py --steps 4 --every 2 --batch $b 2>&1 | tail -1
done; done 2>&1 | tee $out/s0_hashes.txt
