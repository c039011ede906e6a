# timing ablations of the ring attention kernel (libvithip_abl<N>.so built with ABFLAGS=-DVH_ATTN_ABL=<N>; results wrong):
# 1 no exp, 2 no refill DMA, 4 no tile barriers, 8 no PV MFMAs, 16 no QK MFMAs, 32 no LDS reads, 64 no max
for a in 0 $(ls vit-fpga_amd/libvithip_abl*.so | sed 's/.*abl\([0-9]*\).so/\1/' | sort -n) 0; do
  lib=vit-fpga_amd/libvithip_abl$a.so; [ $a = 0 ] && lib=vit-fpga_amd/libvithip.so
  echo -n "abl=$a "; VITHIP_LIB=$PWD/$lib timeout -k 10 100 python tools/attn_bench.py 2>&1 | tail -1 | cut -c1-90
done
