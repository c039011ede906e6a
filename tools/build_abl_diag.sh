#!/bin/bash
# tools/build_abl_diag.sh <bits>: the STAMPED timing-only ablation build tools/mainloop_ablation.py reads ->
# vit-fpga_amd/libvithip_diag_m<bits>.so  (-DVH_DIAG_STAMPS -DVH_EPI_ABL=64 -DVH_MAIN_ABL=<bits> on the two GEMM sources; the other
# objects are the diagnostic build's: run `make -C vit-fpga_amd diag` first)
set -e
cd "$(dirname "$0")/../vit-fpga_amd"
B=$1
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result -DVH_DIAG_STAMPS -DVH_EPI_ABL=64 -DVH_MAIN_ABL=$B"
/opt/rocm/bin/hipcc $F -c csrc/kernels_gemm5.hip -o /tmp/diag_m${B}_g5.o &
/opt/rocm/bin/hipcc $F -c csrc/kernels_gemm.hip -o /tmp/diag_m${B}_g.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvithip_diag_m$B.so /tmp/diag_m${B}_g.o /tmp/diag_m${B}_g5.o csrc/diag_kernels_attn.o csrc/diag_kernels_misc.o csrc/diag_kernels_patch.o csrc/diag_vithip_api.o -ldl -lpthread
echo built libvithip_diag_m$B.so
