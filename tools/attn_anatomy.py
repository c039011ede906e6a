#!/usr/bin/env python3
"""Where the ring attention kernel's waves spend their time (diagnostic build: make -C vit-fpga_amd diag).

  VITHIP_LIB=vit-fpga_amd/libvithip_diag.so python tools/attn_anatomy.py [--config vit_base --batch 512 --dtype bf16]

Prints, per wave index of the workgroup, the share of the wave's shader-clock time in each phase of the item loop.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vh_synth as S  # noqa: E402
import vithip  # noqa: E402

PH = ["top wait", "top barrier", "Q + tile 0", "mid barrier", "mid compute", "last wait", "last barrier", "bookkeeping", "last tile", "norm+store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    cfg = S.CONFIGS[a.config]
    D, H, T = cfg["dim"], cfg["heads"], S.tokens(cfg)
    dt = vithip.DTYPE_BF16 if a.dtype == "bf16" else vithip.DTYPE_FP16
    n = a.batch * T * 3 * D
    f32 = vithip.DeviceBuffer(n * 4)
    vithip.op_fill(f32.ptr, n, 7, 1, 0, 1.0)
    qkv = vithip.DeviceBuffer(n * 2)
    vithip.op_cast(f32.ptr, qkv.ptr, n, dt)
    out = vithip.DeviceBuffer(a.batch * T * D * 2)
    for _ in range(3):
        vithip.op_attention(qkv.ptr, a.batch, T, H, out.ptr, dt)
    out.to_numpy(np.uint16, (8,))
    buf = np.zeros(1024 * 16 * 16, dtype=np.uint64)
    rc = vithip.lib().vh_diag_attn_read(buf.ctypes.data_as(C.c_void_p), buf.size)
    assert rc == 0, rc
    raw = buf.reshape(1024, 16, 16)
    d = raw[:, :, :10].astype(np.float64)
    nw = int((d.sum(2) > 0).sum(1).max())
    wgs = int((d[:, 0].sum(1) > 0).sum())
    print(f"{a.config} b{a.batch} T={T} {a.dtype}: {wgs} workgroups x {nw} waves; mean shader clocks per wave: {d[:wgs, :nw].sum(2).mean():.0f}")
    print("wave  " + "  ".join(f"{p:>12s}" for p in PH))
    for w in range(nw):
        m = d[:wgs, w].mean(0)
        print(f"{w:4d}  " + "  ".join(f"{x / m.sum() * 100:11.1f}%" for x in m))
    m = d[:wgs, :nw].mean((0, 1))
    print(" all  " + "  ".join(f"{x / m.sum() * 100:11.1f}%" for x in m))
    print(" clk  " + "  ".join(f"{x:12.0f}" for x in m))
    # s_memrealtime (100 MHz): when the workgroups started and ended relative to the first start
    st, en = raw[:wgs, 0, 10].astype(np.int64), raw[:wgs, 0, 11].astype(np.int64)
    t0 = st.min()
    print(f"workgroup start (us after the first): median {np.median(st - t0) / 100:.1f}, p90 {np.percentile(st - t0, 90) / 100:.1f}, max {(st.max() - t0) / 100:.1f}; "
          f"end: min {(en.min() - t0) / 100:.1f}, median {np.median(en - t0) / 100:.1f}, max {(en.max() - t0) / 100:.1f}; "
          f"lifetime median {np.median(en - st) / 100:.1f} us -> in-kernel clock {np.median(d[:wgs, 0].sum(1)) / (np.median(en - st) / 100) / 1e3:.2f} GHz")
    hw = raw[:wgs, 0, 12]
    xcc = ((hw >> 32) & 0xF).astype(np.int64)
    se = ((hw >> 13) & 0x7).astype(np.int64)     # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    cu = ((hw >> 8) & 0xF).astype(np.int64)
    life = (en - st) / 100.0
    print("lifetime (us) by XCC:", " ".join(f"{x}:{life[xcc == x].mean():.1f}" for x in range(8)))
    print("lifetime (us) by SE :", " ".join(f"{x}:{life[se == x].mean():.1f}" for x in sorted(set(se))))
    print("lifetime (us) by CU :", " ".join(f"{x}:{life[cu == x].mean():.1f}({(cu == x).sum()})" for x in sorted(set(cu))))
    key = xcc * 1000 + se * 16 + cu
    per = np.bincount(np.unique(key, return_inverse=True)[1])
    print("workgroups per (XCC, SE, CU):", np.bincount(per).tolist(), "(index = workgroups on one CU)")
    for n in sorted(set(per)):
        sel = np.isin(key, np.unique(key)[per == n])
        print(f"  CUs holding {n}: lifetime mean {life[sel].mean():.1f} us")
    if os.environ.get("ATTN_TKDBG"):   # diagnostic build with -DVH_ATTN_TKDBG: the items the ticket exchange delivered
        t = raw[:wgs, :nw, 13:16].astype(np.int64)
        print("ticket exchange, workgroups 0..3, wave 0 and last:", [(t[i, 0].tolist(), t[i, nw - 1].tolist()) for i in range(4)])
        same = all((t[:, w] == t[:, 0]).all() for w in range(nw))
        vals = np.sort(t[:, 0].ravel())
        print(f"all waves agree: {same}; delivered items: min {vals.min()}, max {vals.max()}, distinct {len(set(vals.tolist()))} of {vals.size}")
    simd = ((raw[:wgs, :nw, 12] >> 4) & 3).astype(np.int64)
    print("waves per SIMD within a workgroup (first 4 workgroups):", [np.bincount(simd[i], minlength=4).tolist() for i in range(4)])


if __name__ == "__main__":
    main()
