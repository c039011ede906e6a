#!/usr/bin/env python3
"""How long does a GEMM epilogue take on a CU that has the memory system (almost) to itself?

Needs the diagnostic build (VITHIP_LIB=vit-fpga_amd/libvithip_diag.so).  Launches the one-tile-per-workgroup form (variant 5)
of a GEMM with only a few tiles -- far fewer than CUs, so neither HBM nor the fabric is loaded -- and reads the in-kernel
stamps: main loop / epilogue issue / store drain per wave group.  Growing the tile count until the chip is full shows what
part of an epilogue is the CU's own latency chain and what part is shared bandwidth.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VITHIP_LIB", os.path.join(ROOT, "vit-fpga_amd", "libvithip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
import numpy as np  # noqa: E402
import vithip  # noqa: E402

EPI = {vithip.EPI_BIAS: "bias", vithip.EPI_LNFOLD: "lnfold", vithip.EPI_LNFOLD_GELU: "lnfold+gelu", vithip.EPI_RESID_SPLIT: "resid-split"}


def main():
    L = vithip.lib()
    L.vh_diag_stamps_arm.argtypes = [ctypes.c_int, ctypes.c_int]
    L.vh_diag_stamps_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    max_wgs = 8192
    buf = np.zeros((max_wgs, 8, 16), dtype=np.uint64)
    meta = (ctypes.c_longlong * 8)()
    print(f"{'epilogue':12s} {'N':>5s} {'K':>5s} {'tiles':>6s} | main us | epilogue issue us: G0   G1 | + drain us: G0   G1 | tile us (G1 end)")
    for epi, N, K in ((vithip.EPI_BIAS, 2304, 768), (vithip.EPI_LNFOLD, 2304, 768), (vithip.EPI_LNFOLD_GELU, 3072, 768),
                      (vithip.EPI_RESID_SPLIT, 768, 768), (vithip.EPI_RESID_SPLIT, 768, 3072)):
        for tiles_m in (2, 8, 32, 85, 394):
            M = 256 * tiles_m
            if L.vh_diag_stamps_arm(8, max_wgs):
                raise SystemExit("not the diag build")
            vithip.bench_gemm(M, N, K, epi, vithip.DTYPE_BF16, 5, 4)
            if L.vh_diag_stamps_read(0, buf.ctypes.data, max_wgs, meta):
                continue
            grid = int(meta[5])
            s = buf[:grid].astype(np.int64)          # [wg][wave][stamp]
            ok = (s[:, :, 4] > 0).all(axis=1)
            s = s[ok]
            if not len(s):
                continue
            base = s[:, 0:1, 0]
            main_us = ((s[:, 0, 2] - s[:, 0, 1]).mean()) / 100.0
            e_iss = ((s[:, :, 3] - s[:, :, 2]).mean(axis=0)) / 100.0
            e_drn = ((s[:, :, 4] - s[:, :, 2]).mean(axis=0)) / 100.0
            end = ((s[:, :, 4] - base).mean(axis=0)) / 100.0
            print(f"{EPI[epi]:12s} {N:5d} {K:5d} {grid:6d} | {main_us:7.2f} | {e_iss[:4].mean():17.2f} {e_iss[4:].mean():5.2f} | "
                  f"{e_drn[:4].mean():12.2f} {e_drn[4:].mean():5.2f} | {end[4:].mean():8.2f}", flush=True)


if __name__ == "__main__":
    main()
