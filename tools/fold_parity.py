import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "vit-fpga_amd/python")
import numpy as np, oracle_lib as O, vh_synth as S, vithip
def rel(a,b): return float(np.abs(a-b).max()/np.abs(b).max())
for name,batch,dts in (("vit_base",4,(vithip.DTYPE_FP16,vithip.DTYPE_BF16)),("vit_large_384",1,(vithip.DTYPE_FP16,))):
    cfg=S.CONFIGS[name]; blob=S.make_blob(cfg,0); images=S.make_images(cfg,1,batch)
    ref=O.vit_forward(cfg,blob,images)
    for dt in dts:
        ctx=vithip.VitContext(cfg,dtype=dt,max_batch=batch); ctx.load_weights(blob); got=ctx.forward(images); ctx.close()
        print(name, "fold=%s"%os.environ.get("VH_LN_FOLD","0"), "fp16" if dt==vithip.DTYPE_FP16 else "bf16", "%.3e"%rel(got,ref), "top1", float((got.argmax(1)==ref.argmax(1)).mean()))
