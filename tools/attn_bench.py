#!/usr/bin/env python3
"""Attention micro-benchmark (run on the GPU box): the fused attention kernel alone on the shape of a ViT forward.

  python tools/attn_bench.py [--config vit_base] [--batch 512] [--dtype bf16] [--iters 20]

Prints the average launch time (host clock around `iters` back-to-back launches and a copy that drains the stream) and
the HBM rate of the algorithmic bytes (qkv read once + the output written once).  Under rocprofv3 --pmc it is the
cheap way to get per-kernel counters without the rest of the forward.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import vh_synth as S  # noqa: E402
import vithip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tokens", type=int, default=0, help="operator tap only, on this sequence length instead of the model's")
    a = ap.parse_args()
    cfg = S.CONFIGS[a.config]
    D, H, T = cfg["dim"], cfg["heads"], S.tokens(cfg)
    dt = vithip.DTYPE_BF16 if a.dtype == "bf16" else vithip.DTYPE_FP16
    if a.tokens:
        T = a.tokens
        tap(a, dt, D, H, T)
        return
    byts = a.batch * T * 4 * D * 2
    # (1) device time per launch inside the forward: hip events on the context's stream around every attention launch
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=a.batch)
    ctx.init_weights_seeded(0)
    din = vithip.DeviceBuffer(a.batch * cfg["image_size"] ** 2 * cfg["channels"] * 4)
    dout = vithip.DeviceBuffer(a.batch * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, a.batch, din.ptr)
    ctx.forward_device_async(din.ptr, a.batch, dout.ptr, steps=2)
    ctx.synchronize()
    ctx.set_stage_timing("attention")
    ctx.forward_device_async(din.ptr, a.batch, dout.ptr, steps=max(1, a.iters // 4))
    avg_ms, min_ms, n = ctx.get_stage_timing()
    ctx.set_stage_timing(None)
    ctx.close(); din.free(); dout.free()
    print(f"attention {a.config} b{a.batch} T={T} H={H} {a.dtype}, in the forward (hip events, {n} launches): "
          f"{avg_ms * 1e3:8.1f} us average, {min_ms * 1e3:8.1f} us min  {byts / (avg_ms * 1e3) / 1e6:6.2f} TB/s algorithmic", flush=True)
    tap(a, dt, D, H, T)


def tap(a, dt, D, H, T):
    # (2) the operator tap, timed from the host: every call allocates its work-queue counter, zeroes it, launches and
    # SYNCHRONISES -- 30-40 us of host work per call on top of the kernel.  Round 2 quoted this figure (178-188 us)
    # beside the kernel trace's 150 us; the difference is the tap, not the kernel.
    n_el = a.batch * T * 3 * D
    f32 = vithip.DeviceBuffer(n_el * 4)
    vithip.op_fill(f32.ptr, n_el, 7, 1, 0, 1.0)
    qkv = vithip.DeviceBuffer(n_el * 2)
    vithip.op_cast(f32.ptr, qkv.ptr, n_el, dt)
    f32.free()
    out = vithip.DeviceBuffer(a.batch * T * D * 2)
    drain = lambda: out.to_numpy(np.uint16, (8,))
    for _ in range(3):
        vithip.op_attention(qkv.ptr, a.batch, T, H, out.ptr, dt)
    drain()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        vithip.op_attention(qkv.ptr, a.batch, T, H, out.ptr, dt)
    drain()
    us = (time.perf_counter() - t0) / a.iters * 1e6
    print(f"   T={T} H={H} b{a.batch}: the launch through the synchronising operator tap, host clock: {us:8.1f} us per call (kernel + allocation, memset, "
          f"launch and a stream synchronisation per call)", flush=True)


if __name__ == "__main__":
    main()
