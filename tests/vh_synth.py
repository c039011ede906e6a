"""Synthetic-data generator and canonical weight-blob layout, numpy edition.

Third, independent statement of the generator specified in DESIGN.md "synthetic data"
(the other two: oracle/vit_oracle.c `oracle_fill`, vit-fpga_amd/csrc `fill_kernel`).  Tests
check that all three agree bit for bit.  Also knows the canonical fp32 blob layout so that
fixtures can be produced by libraries that want named tensors (tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np

U64 = np.uint64
GOLD = U64(0x9E3779B97F4A7C15)
M1 = U64(0xBF58476D1CE4E5B9)
M2 = U64(0x94D049BB133111EB)
IH4_STD = 37837.22725

TID_PATCH_W, TID_PATCH_B, TID_CLS, TID_POS, TID_LAYER0, TID_FINAL = 1, 2, 3, 4, 16, 0x7000

CONFIGS = {
    # name: image, patch, channels, dim, heads, mlp, layers, classes
    "vit_tiny": dict(image_size=224, patch_size=16, channels=3, dim=192, heads=3, mlp_dim=768,
                     layers=12, classes=1000),
    "vit_base": dict(image_size=224, patch_size=16, channels=3, dim=768, heads=12, mlp_dim=3072,
                     layers=12, classes=1000),
    "vit_large_384": dict(image_size=384, patch_size=16, channels=3, dim=1024, heads=16,
                          mlp_dim=4096, layers=24, classes=1000),
    # small odd-shaped nets for fast tests: ragged token counts, few layers
    "vit_micro": dict(image_size=64, patch_size=16, channels=3, dim=128, heads=2, mlp_dim=256,
                      layers=2, classes=40),
    "vit_mini": dict(image_size=96, patch_size=16, channels=3, dim=192, heads=3, mlp_dim=384,
                     layers=3, classes=100),
    # odd corners: one channel, 32-pixel patches, 4-row patch grid, class count not a tile multiple
    "vit_gray": dict(image_size=128, patch_size=32, channels=1, dim=128, heads=2, mlp_dim=320,
                     layers=2, classes=12),
    # fp8 path (VH_DTYPE_FP8) needs dim and mlp_dim to be multiples of 128
    "vit_q8": dict(image_size=96, patch_size=16, channels=3, dim=256, heads=4, mlp_dim=640,
                   layers=3, classes=100),
}


def mix64(x):
    x = np.asarray(x, dtype=U64)
    with np.errstate(over="ignore"):
        x = x + GOLD
        x = x ^ (x >> U64(30))
        x = x * M1
        x = x ^ (x >> U64(27))
        x = x * M2
        x = x ^ (x >> U64(31))
    return x


def fill(n, seed, tensor_id, kind, sigma=0.0, offset=0.0, start=0):
    """kind 0: uniform[-1,1); 1: offset + IrwinHall4*sigma; 2: constant offset."""
    stream = mix64(mix64(U64(seed)) ^ U64(tensor_id))
    out = np.empty(n, dtype=np.float32)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = np.arange(start + s, start + e, dtype=U64)
        w = mix64(stream ^ idx)
        if kind == 0:
            u = (w >> U64(40)).astype(np.int64) - (1 << 23)
            out[s:e] = u.astype(np.float32) * np.float32(1.0 / 8388608.0)
        elif kind == 1:
            ssum = ((w & U64(0xFFFF)).astype(np.int64) + ((w >> U64(16)) & U64(0xFFFF)).astype(np.int64)
                    + ((w >> U64(32)) & U64(0xFFFF)).astype(np.int64)
                    + ((w >> U64(48)) & U64(0xFFFF)).astype(np.int64) - 131070)
            scale = np.float64(np.float32(sigma)) / IH4_STD
            v = (ssum.astype(np.float64) * scale).astype(np.float32)
            out[s:e] = np.float32(offset) + v
        else:
            out[s:e] = np.float32(offset)
    return out


def tokens(cfg):
    g = cfg["image_size"] // cfg["patch_size"]
    return 1 + g * g


def tensor_table(cfg):
    """[(name, shape, tensor_id, sigma, offset)] in canonical blob order."""
    D, M, C, P, CH = cfg["dim"], cfg["mlp_dim"], cfg["classes"], cfg["patch_size"], cfg["channels"]
    T = tokens(cfg)
    sw, sb, sg = 0.02, 0.02, 0.05
    t = [("patch.weight", (D, CH, P, P), TID_PATCH_W, sw, 0.0),
         ("patch.bias", (D,), TID_PATCH_B, sb, 0.0),
         ("cls", (D,), TID_CLS, sw, 0.0),
         ("pos", (T, D), TID_POS, sw, 0.0)]
    for l in range(cfg["layers"]):
        b = TID_LAYER0 + 16 * l
        t += [(f"l{l}.ln1.weight", (D,), b + 0, sg, 1.0), (f"l{l}.ln1.bias", (D,), b + 1, sb, 0.0),
              (f"l{l}.q.weight", (D, D), b + 2, sw, 0.0), (f"l{l}.q.bias", (D,), b + 3, sb, 0.0),
              (f"l{l}.k.weight", (D, D), b + 4, sw, 0.0), (f"l{l}.k.bias", (D,), b + 5, sb, 0.0),
              (f"l{l}.v.weight", (D, D), b + 6, sw, 0.0), (f"l{l}.v.bias", (D,), b + 7, sb, 0.0),
              (f"l{l}.o.weight", (D, D), b + 8, sw, 0.0), (f"l{l}.o.bias", (D,), b + 9, sb, 0.0),
              (f"l{l}.ln2.weight", (D,), b + 10, sg, 1.0), (f"l{l}.ln2.bias", (D,), b + 11, sb, 0.0),
              (f"l{l}.fc1.weight", (M, D), b + 12, sw, 0.0), (f"l{l}.fc1.bias", (M,), b + 13, sb, 0.0),
              (f"l{l}.fc2.weight", (D, M), b + 14, sw, 0.0), (f"l{l}.fc2.bias", (D,), b + 15, sb, 0.0)]
    t += [("lnf.weight", (D,), TID_FINAL + 0, sg, 1.0), ("lnf.bias", (D,), TID_FINAL + 1, sb, 0.0),
          ("head.weight", (C, D), TID_FINAL + 2, sw, 0.0), ("head.bias", (C,), TID_FINAL + 3, sb, 0.0)]
    return t


def param_count(cfg):
    return sum(int(np.prod(s)) for _, s, _, _, _ in tensor_table(cfg))


def blob_header(cfg, ln_eps=1e-6):
    h = np.zeros(64, dtype=np.uint8)
    h[:7] = np.frombuffer(b"VHBLOB1", dtype=np.uint8)
    ints = np.array([cfg[k] for k in ("image_size", "patch_size", "channels", "dim", "heads",
                                      "mlp_dim", "layers", "classes")], dtype=np.int32)
    h[8:40] = ints.view(np.uint8)
    h[40:44] = np.array([ln_eps], dtype=np.float32).view(np.uint8)
    return h


def make_tensors(cfg, seed):
    return {name: fill(int(np.prod(shape)), seed, tid, 1, sigma, off).reshape(shape)
            for name, shape, tid, sigma, off in tensor_table(cfg)}


def pack_blob(cfg, tensors, ln_eps=1e-6):
    parts = [blob_header(cfg, ln_eps)]
    for name, shape, *_ in tensor_table(cfg):
        a = np.ascontiguousarray(tensors[name], dtype=np.float32)
        assert a.shape == tuple(shape), (name, a.shape, shape)
        parts.append(a.reshape(-1).view(np.uint8))
    return np.concatenate(parts)


def make_blob(cfg, seed, ln_eps=1e-6):
    return pack_blob(cfg, make_tensors(cfg, seed), ln_eps)


def make_images(cfg, seed, batch, tensor_id=0x100):
    n = batch * cfg["image_size"] * cfg["image_size"] * cfg["channels"]
    return fill(n, seed, tensor_id, 0).reshape(batch, cfg["image_size"], cfg["image_size"],
                                               cfg["channels"])


def flops_per_image(cfg):
    """Algorithmic FLOPs (2 x MACs of every GEMM + QK^T + PV), SURVEY.md §8d."""
    D, M, C, L = cfg["dim"], cfg["mlp_dim"], cfg["classes"], cfg["layers"]
    T = tokens(cfg)
    kp = cfg["patch_size"] ** 2 * cfg["channels"]
    mac = (T - 1) * kp * D
    mac += L * (T * D * 3 * D + 2 * T * T * D + T * D * D + 2 * T * D * M)
    mac += D * C
    return 2 * mac
