#!/usr/bin/env python3
"""Anatomy of the ping-pong GEMM launches inside the real forward, from in-kernel clock stamps.

Needs the DIAGNOSTIC build (make -C vit-fpga_amd diag -> libvithip_diag.so, -DVH_DIAG_STAMPS): wave 0 of every
workgroup stamps s_memrealtime (100 MHz) at kernel entry / first K-tile visible / main loop done / epilogue issued /
stores drained, s_memtime (shader clock) around the main loop, and its CU's hardware id.  The product library contains
none of this.  Protocol (MI355X_MICROARCH.md, "DVFS give-back" item 6): >= 2 s of back-to-back forwards on random
data, then the stamps of the most recent launches are read.  Never quote this build's run time; read its SHARES.

  VITHIP_LIB=vit-fpga_amd/libvithip_diag.so python tools/gemm_anatomy.py [--dtype bf16] [--batch 512] [--seconds 3]
"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VITHIP_LIB", os.path.join(ROOT, "vit-fpga_amd", "libvithip_diag.so"))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import vh_synth as S  # noqa: E402
import vithip  # noqa: E402

EPI = {0: "bias", 1: "bias+gelu", 2: "bias+resid", 3: "bias->f32", 4: "patch", 5: "lnfold", 6: "lnfold+gelu", 7: "resid+ln-stats"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp8"])
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--launches", type=int, default=64, help="most recent ping-pong launches to analyse")
    ap.add_argument("--clock-json", default=None,
                    help="write the in-kernel clock of the fc1 GEMM's main loop (with the hash of the kernel sources) to this "
                         "file: bench.py's roofline.attainable reads profiles/*_fc1_clock.json")
    args = ap.parse_args()

    L = vithip.lib()
    L.vh_diag_stamps_arm.argtypes = [ctypes.c_int, ctypes.c_int]
    L.vh_diag_stamps_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.vh_diag_stamps_count.restype = ctypes.c_longlong

    cfg = S.CONFIGS[args.config]
    dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16, "fp8": vithip.DTYPE_FP8}[args.dtype]
    B = args.batch
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=B)
    ctx.init_weights_seeded(0)
    din = vithip.DeviceBuffer(B * cfg["image_size"] ** 2 * cfg["channels"] * 4)
    dout = vithip.DeviceBuffer(B * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, B, din.ptr)
    ctx.forward_device_async(din.ptr, B, dout.ptr, steps=2)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.forward_device_async(din.ptr, B, dout.ptr, steps=4)
    ctx.synchronize()
    per = (time.perf_counter() - t0) / 4
    max_wgs = 8192
    if L.vh_diag_stamps_arm(args.launches, max_wgs):
        raise SystemExit("vh_diag_stamps_arm failed (is VITHIP_LIB the diag build?)")
    steps = max(2, int(args.seconds / per))
    ctx.forward_device_async(din.ptr, B, dout.ptr, steps=steps)
    ctx.synchronize()
    print(f"# {args.config} {args.dtype} batch {B}: {steps} forwards back to back, {per * 1e3:.2f} ms each (diag build: do not quote)")

    groups = {}
    n = min(args.launches, int(L.vh_diag_stamps_count()))
    buf = np.zeros((max_wgs, 8, 16), dtype=np.uint64)   # [workgroup][wave][stamp]: 0-7 tile stamps, 8-15 phase stamps of one K-tile
    meta = (ctypes.c_longlong * 8)()
    for age in range(n):
        if L.vh_diag_stamps_read(age, buf.ctypes.data, max_wgs, meta):
            continue
        M, N, K, epi, f8, grid, variant = [int(meta[i]) for i in range(7)]
        groups.setdefault((M, N, K, epi, f8, variant), []).append(buf[:grid].copy())
    waves = {}

    print(f"{'shape (M N K epilogue)':44s} {'launches':>8s} {'tiles':>6s} {'span us':>8s} | per tile, us: {'prologue':>8s} {'main':>7s} {'epi issue':>9s} {'drain':>6s} "
          f"{'gap':>6s} | {'clk GHz':>7s} {'MFMA busy% of main':>18s} {'main share%':>11s} {'PF':>6s}")
    for key, launches in sorted(groups.items(), key=lambda kv: -kv[0][1] * kv[0][2]):
        M, N, K, epi, f8, variant = key
        pro, mainl, epi_t, drain, gaps, clk, span = [], [], [], [], [], [], []
        wv = np.zeros((8, 4))
        nwv = 0
        for st8 in launches:
            s8 = st8.astype(np.int64)
            okw = (s8[:, :, 4] > 0).all(axis=1)
            if okw.any():
                # per wave, relative to wave 0's top-of-tile stamp: end of main loop, epilogue issued, iteration end
                base = s8[okw, 0:1, 0]
                wv[:, 0] += ((s8[okw, :, 1] - base).mean(axis=0)) / 100.0
                wv[:, 1] += ((s8[okw, :, 2] - base).mean(axis=0)) / 100.0
                wv[:, 2] += ((s8[okw, :, 3] - base).mean(axis=0)) / 100.0
                wv[:, 3] += ((s8[okw, :, 4] - base).mean(axis=0)) / 100.0
                nwv += 1
            st = st8[:, 0, :]
            s = st.astype(np.int64)
            ok = s[:, 4] > 0
            s = s[ok]
            if not len(s):
                continue
            pro.append((s[:, 1] - s[:, 0]).mean() / 100.0)
            mainl.append((s[:, 2] - s[:, 1]).mean() / 100.0)
            epi_t.append((s[:, 3] - s[:, 2]).mean() / 100.0)
            drain.append((s[:, 4] - s[:, 3]).mean() / 100.0)
            span.append((s[:, 4].max() - s[:, 0].min()) / 100.0)
            d_rt = (s[:, 2] - s[:, 1]).astype(np.float64)
            d_ct = (s[:, 6] - s[:, 5]).astype(np.float64)
            clk.append(np.median(d_ct / np.maximum(d_rt, 1.0)) * 0.1)   # cycles per 10 ns -> GHz
            # turnaround on one CU: entry of the next workgroup minus the drained stamp of the previous one
            hw = st[ok][:, 7]
            cu = ((hw >> np.uint64(32)) & np.uint64(0xF)) * np.uint64(65536) + (hw & np.uint64(0xFF00))   # xcc | se/sh/cu bits
            order = np.lexsort((s[:, 0], cu.astype(np.int64)))
            cs, e0, e4 = cu.astype(np.int64)[order], s[order, 0], s[order, 4]
            same = cs[1:] == cs[:-1]
            g = (e0[1:] - e4[:-1])[same]
            if len(g):
                gaps.append(np.median(g) / 100.0)
        if not mainl:
            continue
        nk = K // (128 if f8 else 64)
        mfma_cycles = nk * 2048.0          # per SIMD per tile: 2 waves x 64 MFMAs x 16 cycles per K-tile (fp8: 32 x 32)
        c = float(np.mean(clk))
        m_us = float(np.mean(mainl))
        if args.clock_json and N == cfg["mlp_dim"] and K == cfg["dim"] and not f8 == (args.dtype != "fp8"):
            import json
            sys.path.insert(0, ROOT)
            import bench
            json.dump({"source_sha": bench.source_sha(), "config": args.config, "batch": B, "dtype": args.dtype,
                       "clock_ghz": round(c, 4), "kernel": "fc1 GEMM main loop (d s_memtime / d s_memrealtime x 100 MHz, median over "
                       "workgroups, diagnostic build, after >= 2 s of back-to-back forwards)", "launches": len(launches)},
                      open(args.clock_json, "w"), indent=1)
        busy = 100.0 * mfma_cycles / (m_us * c * 1e3) if m_us > 0 else 0.0
        tile_us = np.mean(pro) + m_us + np.mean(epi_t) + np.mean(drain) + (np.mean(gaps) if gaps else 0.0)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        pf = 2.0 * M * N * K / (np.mean(span) * 1e-6) / 1e15
        print(f"{M:7d} {N:5d} {K:5d} {EPI.get(epi, str(epi)):>16s}{' e4m3' if f8 else '':5s} v{variant} {len(launches):8d} {tiles:6d} {np.mean(span):8.1f} | "
              f"{'':13s} {np.mean(pro):8.2f} {m_us:7.2f} {np.mean(epi_t):9.2f} {np.mean(drain):6.2f} {np.mean(gaps) if gaps else float('nan'):6.2f} | "
              f"{c:7.3f} {busy:18.1f} {100.0 * m_us / tile_us:11.1f} {pf:6.3f}")
        # are the workgroups of a launch in step?  spread over the workgroups of the absolute time (chip-wide 100 MHz
        # counter) at which each one finished the stamped tile's main loop, i.e. started its epilogue
        sp = []
        for st8 in launches:
            s0 = st8[:, 0, :].astype(np.int64)
            s0 = s0[s0[:, 4] > 0]
            if len(s0) > 8:
                t = s0[:, 2] / 100.0
                sp.append([np.percentile(t, q) - np.median(t) for q in (5, 25, 75, 95)] + [t.std()])
        if sp:
            a = np.mean(np.array(sp), axis=0)
            print(f"      epilogue start across the workgroups of a launch, us around the median: p5 {a[0]:+.2f}  p25 {a[1]:+.2f}  p75 {a[2]:+.2f}  p95 {a[3]:+.2f}  (sigma {a[4]:.2f})")
        # one K-tile (the middle one of the stamped tile) phase by phase, shader cycles, median over workgroups and launches:
        # work = until the phase's last instruction has issued (L phases: reads back + DMA issued + counted wait), wait = at
        # the barrier that closes it.  G0 = wave 0, G1 = wave 4 (one barrier behind).
        if not f8:
            rows = []
            for w_ in (0, 4):
                ph = np.concatenate([st8[:, w_, 8:16].astype(np.int64) for st8 in launches])
                ph = ph[(ph[:, 0] > 0) & (ph[:, 7] > ph[:, 0])]
                if len(ph):
                    d = np.diff(ph, axis=1)
                    rows.append((w_, np.median(d, axis=0), np.median(ph[:, 7] - ph[:, 0])))
            if rows:
                print("      one K-tile, shader cycles (median): L0 work | L0 barrier | C0 issue | C0 barrier | L1 work+wait | L1 barrier | C1 issue+wait |  sum of these 7")
                for w_, d, tot in rows:
                    print(f"        wave {w_} (G{w_ // 4}): " + " ".join(f"{x:12.0f}" for x in d) + f" | {tot:8.0f}   (K-tile at this clock: {m_us * c * 1e3 / nk:6.0f} cycles, MFMA 2048 per SIMD)")
        if nwv:
            wv /= nwv
            print("      per wave (us after wave 0's top-of-tile stamp): loop entered | main loop done | epilogue issued | iteration end")
            for w in range(8):
                print(f"        wave {w}: {wv[w, 0]:7.2f} {wv[w, 1]:7.2f} {wv[w, 2]:7.2f} {wv[w, 3]:7.2f}")
    ctx.close()


if __name__ == "__main__":
    main()
