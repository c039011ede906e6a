#!/usr/bin/env python3
"""Soak test: run the batch-512 forward many times and require bit-identical logits every time.

A latent hazard in the LDS pipelines (a missing wait, a barrier one phase off) would show as rare bit flips, not as a
test failure on one run.  usage: python tools/soak.py [--steps 3000] [--every 20] [--dtype bf16|fp16|fp8] [--config vit_base]
"""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp8"])
    args = ap.parse_args()
    import numpy as np
    import vh_synth as S
    import vithip

    cfg = S.CONFIGS[args.config]
    dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16, "fp8": vithip.DTYPE_FP8}[args.dtype]
    B = args.batch
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=B)
    ctx.init_weights_seeded(0)
    din = vithip.DeviceBuffer(B * cfg["image_size"] ** 2 * cfg["channels"] * 4)
    dout = vithip.DeviceBuffer(B * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, B, din.ptr)
    ref = None
    t0 = time.time()
    done = 0
    while done < args.steps:
        n = min(args.every, args.steps - done)
        ctx.forward_device_async(din.ptr, B, dout.ptr, steps=n)
        ctx.synchronize()
        done += n
        h = hashlib.sha256(dout.to_numpy(np.uint8, (B * cfg["classes"] * 4,)).tobytes()).hexdigest()
        if ref is None:
            ref = h
        elif h != ref:
            print(f"MISMATCH after {done} forwards: {h[:16]} != {ref[:16]}")
            sys.exit(1)
        if done % (args.every * 25) == 0:
            print(f"{done} forwards, {time.time() - t0:.0f} s, logits hash {ref[:16]}", flush=True)
    print(f"OK: {done} forwards of {args.config} {args.dtype} batch {B}, every {args.every}th logits buffer identical ({ref[:16]})")


if __name__ == "__main__":
    main()
