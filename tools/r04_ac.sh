#!/bin/bash
# round 4, call ac: phase stamps of K-tile 0, 1, 2 and the middle one of a persistent workgroup's ninth tile (diagnostic builds
# -DVH_DIAG_KT=n -DVH_DIAG_REC_IT=8): what do the first counted waits of a tile pay for the previous tile's stores (one in-order vmcnt)?
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r04; mkdir -p $out
for kt in -1 0 1 2; do
  echo "== K-tile $kt (tile 8 of the workgroup)"; VITHIP_LIB=$PWD/vit-fpga_amd/libvithip_diag_kt$kt.so timeout -k 10 200 python tools/gemm_anatomy.py 2>&1 | grep -E "^ +[0-9]+ +[0-9]+ +[0-9]+ |wave 0 \(G0\)|wave 4 \(G1\)"
done | tee $out/ac_ktile_phase_stamps.txt
