#!/usr/bin/env python3
"""Where does the 16-bit error of the logits come from?  CPU only (oracle/, test infrastructure): the fp32 forward is
re-run with ONE of the device's MFMA-operand rounding points switched on at a time (oracle_vit_forward_emul16), and
the distance to the plain fp32 logits is reported in the parity metric of the tests (max|d| / max|ref| per image).
Independent rounding errors add in quadrature, so the squares of the single-point medians show each point's share.

  python tools/parity_attribution.py [--config vit_base] [--images 16] [--dtype fp16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import vh_synth as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"])
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    cfg = S.CONFIGS[args.config]
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, args.seed, args.images)
    ref = O.vit_forward(cfg, blob, images)
    dt = 0 if args.dtype == "bf16" else 1
    B = O.EMUL_BITS
    plain = (B["weights"] | B["ln_out"] | B["qkv"] | B["probs"] | B["attn_out"] | B["gelu_out"] | B["patches"] | B["cls_rows"]
             | B["head_weights"])
    head = B["cls_rows"] | B["head_weights"]
    rows = [(k, v) for k, v in B.items() if k != "ln_folded"]
    rows += [("ln_folded (operand = raw residual)", B["ln_folded"]),
             ("ALL, stand-alone LayerNorm", plain),
             ("ALL, LayerNorm folded", (plain & ~B["ln_out"]) | B["ln_folded"]),
             ("ALL but the head (fp32 head), stand-alone LN", plain & ~head),
             ("ALL but the head (fp32 head), LN folded", ((plain & ~B["ln_out"]) | B["ln_folded"]) & ~head)]
    print(f"# {args.config}, {args.images} images (seed {args.seed}), {args.dtype}: max|d|/max|ref| per image vs the fp32 oracle")
    print(f"{'rounding point':48s} {'worst':>10s} {'median':>10s} {'median^2 share':>15s}")
    res = []
    for name, mask in rows:
        got = O.vit_forward_emul16(cfg, blob, images, dt, mask)
        per = np.abs(got - ref).max(1) / np.abs(ref).max()
        res.append((name, per.max(), float(np.median(per))))
    tot = sum(m * m for n, w, m in res if not n.startswith("ALL") and not n.startswith("ln_folded"))
    for name, w, m in res:
        share = f"{100 * m * m / tot:14.1f}%" if not name.startswith("ALL") else ""
        print(f"{name:48s} {w:10.3e} {m:10.3e} {share:>15s}")


if __name__ == "__main__":
    main()
