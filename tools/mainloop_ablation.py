#!/usr/bin/env python3
"""What the K loop of the persistent 256x256x64 GEMM consists of: one timing-only ablation build per run
(VITHIP_LIB = a libvithip_diag_m<bits>.so built with -DVH_DIAG_STAMPS -DVH_EPI_ABL=64 -DVH_MAIN_ABL=<bits>, see
kernels_gemm5.hip), the GEMM launched back to back for about a second (steady clock), then launch time from HIP events
and -- from the in-kernel stamps of the last launches -- the shader clock and the cycles one K-tile takes.
  VITHIP_LIB=vit-fpga_amd/libvithip_diag_m3.so python tools/mainloop_ablation.py [--shape fc1]
Results are garbage by construction (operands partly not read, nothing stored): timing only.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
import vithip  # noqa: E402

SHAPES = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="fc1,fc2")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--label", default="")
    args = ap.parse_args()
    L = vithip.lib()
    L.vh_diag_stamps_arm.argtypes = [ctypes.c_int, ctypes.c_int]
    L.vh_diag_stamps_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.vh_diag_stamps_count.restype = ctypes.c_longlong
    M = 100864
    max_wgs = 512
    for name in args.shape.split(","):
        N, K = SHAPES[name]
        t = vithip.bench_gemm(M, N, K, vithip.EPI_BIAS, vithip.DTYPE_BF16, 0, 20)
        iters = max(50, int(args.seconds / (t * 1e-3)))
        if L.vh_diag_stamps_arm(8, max_wgs):
            raise SystemExit("vh_diag_stamps_arm failed (is VITHIP_LIB a diag build?)")
        t = min(vithip.bench_gemm(M, N, K, vithip.EPI_BIAS, vithip.DTYPE_BF16, 0, iters) for _ in range(2)) * 1e3
        n = min(8, int(L.vh_diag_stamps_count()))
        buf = np.zeros((max_wgs, 8, 16), dtype=np.uint64)
        meta = (ctypes.c_longlong * 8)()
        clk, main_us, cyc = [], [], []
        for age in range(n):
            if L.vh_diag_stamps_read(age, buf.ctypes.data, max_wgs, meta):
                continue
            grid = int(meta[5])
            s = buf[:grid, 0, :].astype(np.int64)
            s = s[s[:, 2] > s[:, 1]]
            if not len(s):
                continue
            d_rt = (s[:, 2] - s[:, 1]).astype(np.float64)      # 10 ns units
            d_ct = (s[:, 6] - s[:, 5]).astype(np.float64)      # shader cycles
            clk.append(np.median(d_ct / d_rt) * 0.1)
            main_us.append(np.median(d_rt) / 100.0)
            cyc.append(np.median(d_ct))
        phases = {}
        for age in range(n):
            if L.vh_diag_stamps_read(age, buf.ctypes.data, max_wgs, meta):
                continue
            grid = int(meta[5])
            for w_ in (0, 4):
                ph = buf[:grid, w_, 8:16].astype(np.int64)
                ph = ph[(ph[:, 0] > 0) & (ph[:, 7] > ph[:, 0])]
                if len(ph):
                    phases.setdefault(w_, []).append(ph)
        nk = K // 64
        fl = 2.0 * M * N * K
        c = float(np.mean(clk)) if clk else float("nan")
        print(f"{args.label:28s} {name:5s} launch {t:7.1f} us {fl / t / 1e6:7.1f} TF | clock {c:5.3f} GHz | K loop of one tile "
              f"{float(np.mean(main_us)) if main_us else float('nan'):6.2f} us = {float(np.mean(cyc)) / nk if cyc else float('nan'):6.0f} cycles per K-tile "
              f"(2048 = the MFMAs alone) | launch / tile rounds: {t * c * 1e3 / (-(-(M // 256 * (N // 256)) // 256)) / nk:6.0f} cycles per K-tile", flush=True)
        for w_, lst in sorted(phases.items()):
            ph = np.concatenate(lst)
            d = np.median(np.diff(ph, axis=1), axis=0)
            print(f"{'':28s}       wave {w_} (G{w_ // 4}), cycles: L0 {d[0]:5.0f} | barrier {d[1]:5.0f} | C0 {d[2]:5.0f} | barrier {d[3]:5.0f} | L1 {d[4]:5.0f} | barrier {d[5]:5.0f} | C1 {d[6]:5.0f} "
                  f"| these seven {np.median(ph[:, 7] - ph[:, 0]):5.0f}", flush=True)


if __name__ == "__main__":
    main()
