#!/usr/bin/env python3
"""Calibration line only (never on the product path): torch.matmul (hipBLASLt/rocBLAS through PyTorch) on the four
per-layer GEMM shapes of ViT-B/16 batch 512, same operand distribution as vh_bench_gemm (uniform[-1,1) activations,
sigma=0.02 weights), plain GEMM without epilogue, next to libvithip's kernel on the same device in the same process.
  python tools/torch_matmul_calib.py [--iters 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--no-torch", action="store_true", help="skip the torch.matmul column (A/B runs of two library builds)")
    args = ap.parse_args()
    import vithip
    if args.no_torch:
        M = 100864
        print(f"# {os.environ.get('VITHIP_LIB', 'libvithip.so')}")
        for name, N, K, epi in [("qkv", 2304, 768, vithip.EPI_LNFOLD), ("proj", 768, 768, vithip.EPI_RESID_SPLIT),
                                ("fc1", 3072, 768, vithip.EPI_LNFOLD_GELU), ("fc2", 768, 3072, vithip.EPI_RESID_SPLIT)]:
            v = min(vithip.bench_gemm(M, N, K, vithip.EPI_BIAS, vithip.DTYPE_BF16, 0, args.iters) for _ in range(args.rounds)) * 1e3
            e = min(vithip.bench_gemm(M, N, K, epi, vithip.DTYPE_BF16, 0, args.iters) for _ in range(args.rounds)) * 1e3
            fl = 2.0 * M * N * K
            print(f"{name:6s} {N:5d} {K:5d} | plain bias: {v:8.1f} us {fl / v / 1e6:7.1f} TF | layer epilogue: {e:8.1f} us {fl / e / 1e6:7.1f} TF | +{100 * (e / v - 1):5.1f} %", flush=True)
        return
    import torch
    M = 100864
    # the epilogues the folded layer loop really launches (vh_bench_gemm allocates their statistics / second plane)
    shapes = [("qkv", 2304, 768, vithip.EPI_LNFOLD), ("proj", 768, 768, vithip.EPI_RESID_SPLIT),
              ("fc1", 3072, 768, vithip.EPI_LNFOLD_GELU), ("fc2", 768, 3072, vithip.EPI_RESID_SPLIT)]
    dev = "cuda:0"
    print(f"{'shape':6s} {'N':>5s} {'K':>5s} | torch.matmul bf16 (no epilogue): us  TF | libvithip plain bias epilogue: us  TF | libvithip layer epilogue: us  TF")
    for name, N, K, epi in shapes:
        a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        wt = w.t()
        best_t = 1e9
        best_v = 1e9
        best_e = 1e9
        for _ in range(args.rounds):   # interleaved rounds in one process (cdna_hip_programming.md rule 24)
            for _ in range(3):
                torch.matmul(a, wt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                torch.matmul(a, wt)
            e1.record()
            torch.cuda.synchronize()
            best_t = min(best_t, e0.elapsed_time(e1) / args.iters * 1e3)
            best_v = min(best_v, vithip.bench_gemm(M, N, K, vithip.EPI_BIAS, vithip.DTYPE_BF16, 0, args.iters) * 1e3)
            best_e = min(best_e, vithip.bench_gemm(M, N, K, epi, vithip.DTYPE_BF16, 0, args.iters) * 1e3)
        fl = 2.0 * M * N * K
        print(f"{name:6s} {N:5d} {K:5d} | {best_t:9.1f} {fl / best_t / 1e6:7.1f} | {best_v:9.1f} {fl / best_v / 1e6:7.1f} | {best_e:9.1f} {fl / best_e / 1e6:7.1f}", flush=True)
        del a, w, wt


if __name__ == "__main__":
    main()
