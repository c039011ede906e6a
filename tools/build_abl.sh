#!/bin/bash
# tools/build_abl.sh <name> <hipcc -D flags...>: timing-only ablation build of the library -> vit-fpga_amd/libvithip_abl_<name>.so
# (only kernels_gemm*.hip are recompiled with the flags; the other objects are the product build's)
set -e
cd "$(dirname "$0")/../vit-fpga_amd"
N=$1; shift
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $F "$@" -c csrc/kernels_gemm5.hip -o /tmp/abl_${N}_g5.o &
/opt/rocm/bin/hipcc $F "$@" -c csrc/kernels_gemm.hip -o /tmp/abl_${N}_g.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvithip_abl_$N.so /tmp/abl_${N}_g.o /tmp/abl_${N}_g5.o csrc/kernels_attn.o csrc/kernels_misc.o csrc/kernels_patch.o csrc/vithip_api.o -ldl -lpthread
echo built libvithip_abl_$N.so
