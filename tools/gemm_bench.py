#!/usr/bin/env python3
"""GEMM micro-benchmark over the shapes of the ViT forward (run on the GPU box).

  python tools/gemm_bench.py [--variants 1,2,3,4] [--batch 512] [--config vit_base] [--dtype bf16]

Prints, per (shape, epilogue, variant), the average launch time and the algorithmic TFLOP/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import vh_synth as S  # noqa: E402
import vithip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="2,5,6")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    cfg = S.CONFIGS[args.config]
    D, Mh, T = cfg["dim"], cfg["mlp_dim"], S.tokens(cfg)
    M = args.batch * T
    dt = vithip.DTYPE_BF16 if args.dtype == "bf16" else vithip.DTYPE_FP16
    shapes = [("qkv", M, 3 * D, D, vithip.EPI_BIAS), ("proj", M, D, D, vithip.EPI_BIAS_RESID),
              ("fc1", M, Mh, D, vithip.EPI_BIAS_GELU), ("fc2", M, D, Mh, vithip.EPI_BIAS_RESID)]
    variants = [int(v) for v in args.variants.split(",")]
    total = {v: 0.0 for v in variants}
    for name, m, n, k, epi in shapes:
        flops = 2.0 * m * n * k
        for v in variants:
            best = min(vithip.bench_gemm(m, n, k, epi, dt, v, args.iters) for _ in range(args.rounds))
            total[v] += best
            print(f"{name:5s} M={m} N={n:5d} K={k:5d} variant {v}: {best * 1e3:8.1f} us  {flops / best / 1e9:8.1f} TFLOP/s",
                  flush=True)
    for v in variants:
        print(f"variant {v}: sum of the four GEMMs {total[v] * 1e3:8.1f} us per layer")


if __name__ == "__main__":
    main()
