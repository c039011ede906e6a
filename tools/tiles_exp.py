import sys; sys.path.insert(0,"vit-fpga_amd/python"); sys.path.insert(0,"tests")
import vithip
N,K=3072,768
for epi,name in ((vithip.EPI_BIAS,"bias16"),(vithip.EPI_BIAS_GELU,"gelu16"),(vithip.EPI_BIAS_F32,"f32")):
    for mt in (1,2,4,8,16,21):
        M=mt*256
        ms=vithip.bench_gemm(M,N,K,epi,vithip.DTYPE_BF16,variant=5,iters=50)
        print(f"{name} tiles {mt*12:4d}: {ms*1e3:7.1f} us")
