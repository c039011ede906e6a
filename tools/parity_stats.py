#!/usr/bin/env python3
"""Parity statistics on a larger sample than the unit tests use: ViT-B/16, 64 images, against the CPU oracle
(PARITY_CONFIG / PARITY_N / PARITY_DTYPES select another model, sample size, operand types).
PARITY_FOLD=on|off selects the LayerNorm path through vh_config.flags (default: the library's, folded for ViT-B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib as O, vh_synth as S, vithip
cfg_name = os.environ.get("PARITY_CONFIG", "vit_base"); cfg = S.CONFIGS[cfg_name]; n = int(os.environ.get("PARITY_N", "64"))
dtypes = os.environ.get("PARITY_DTYPES", "fp16,bf16,fp8").split(",")
blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, n)
ref = O.vit_forward(cfg, blob, images)
fold = os.environ.get("PARITY_FOLD", "default")
flags = {"on": vithip.FLAG_LN_FOLD_ON, "off": vithip.FLAG_LN_FOLD_OFF}.get(fold, 0)
for name, dt in (("fp16", vithip.DTYPE_FP16), ("bf16", vithip.DTYPE_BF16), ("fp8", vithip.DTYPE_FP8)):
    if name not in dtypes: continue
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=n, flags=flags); ctx.load_weights(blob); got = ctx.forward(images)
    guard = ctx.ln_guard(); ctx.close()
    per = np.abs(got - ref).max(1) / np.abs(ref).max()
    rms = np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    top1 = (got.argmax(1) == ref.argmax(1)).mean()
    top5 = np.mean([len(set(np.argsort(-got[i])[:5]) & set(np.argsort(-ref[i])[:5])) / 5 for i in range(n)])
    print(f"{cfg_name} {n} images {name} LayerNorm fold={fold}: max|d|/max|ref| worst {per.max():.3e} median {np.median(per):.3e}; "
          f"rms {rms:.3e}; top-1 agreement {top1:.3f}, top-5 overlap {top5:.3f}; fold guard: max |row mean|/sigma {guard[0]:.3f} "
          f"(threshold {guard[1]:.2f}, tripped {guard[2]})")
