#!/usr/bin/env python3
"""bench.py — images/sec of the ViT hot path on N MI355X (one process per GPU).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward of libvithip.so over one batch of synthetic images that is already
resident in HBM (ViT-B/16, 224x224x3, 512 images per GPU, bf16 MFMA operands / fp32 accumulate).
Images are sharded across ranks (independent units, no data-path collective): rank 0 generates the
weights, the canonical blob is RCCL-broadcast over xGMI once, every rank then runs its own forward.
Prints ONE JSON line on rank 0 (contract in the task statement): whole-job images/s, plus
  roofline     — the dominant kernel (fc1 GEMM, 256x256-tile MFMA kernel with the GELU epilogue):
                 algorithmic FLOP per launch / average launch duration, measured with hip events on
                 the context's stream inside the timed region, against the dense bf16 MFMA peak;
                 `traffic` = HBM bytes per launch from the PMC passes of THIS source tree
                 (profiles/*_fc1_traffic.json whose recorded source hash equals the tree's), else null;
  parity       — the logits of the first images of the timed batch against the fp32 CPU oracle
                 (max|d| / max|ref| per image: worst and median), with the north star's 1e-3 beside it;
  fp16         — the same measurement (value, roofline, parity) with fp16 MFMA operands, the operand
                 type that is inside the north star's tolerance (bf16, the dtype the headline config
                 names, is not and cannot be: DESIGN.md "Numerics");
  vit_large_384_fp16_b256 / fp8_b512 — short measurements (value, roofline, parity) of BASELINE configs 4 and 5, so
                 that the driver's record covers them too (default run only; `--no-extra-configs` skips them);
  cpu_baseline — the CPU oracle behind net::net_abstract (tests/cpp/net_cpu, a port: the reference has
                 no CPU path) timed through launch_forward on the host cores (N=1 only).
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vit-fpga_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp8": 5000.0}  # dense MFMA peak, MI355X_MICROARCH.md
NORTH_STAR_TOL = 1e-3


def source_sha():
    """Hash of the sources the kernels are built from: a PMC measurement belongs to ONE state of them."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "vit-fpga_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "vit-fpga_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "vithip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(config, batch, dtype, streams):
    """HBM bytes per fc1 launch from the PMC passes (tools/pmc_passes.sh + tools/pmc_traffic.py) of exactly this
    source tree and workload, or None: a constant from another build is not a measurement."""
    sha = source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fc1_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if (d.get("source_sha") == sha and d.get("config") == config and d.get("batch") == batch
                and d.get("dtype") == dtype and streams == 1):
            return d["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def measured_clock(config, batch, dtype):
    """In-kernel shader clock (GHz) of the fc1 GEMM's main loop from the diagnostic build's s_memtime / s_memrealtime
    stamps (tools/gemm_anatomy.py --json), taken once per state of the kernel sources like the traffic file; or None."""
    sha = source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fc1_clock.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("source_sha") == sha and d.get("config") == config and d.get("batch") == batch and d.get("dtype") == dtype:
            return d["clock_ghz"], os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU")
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp8"],
                    help="fp8 = BASELINE config 5: e4m3 operands for the four per-layer GEMMs (bf16 elsewhere)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target length of the CPU baseline sample")
    ap.add_argument("--weights", default="full", choices=["full", "e4m3"],
                    help="e4m3: weight-only fp8, VH_FLAG_W8_E4M3 (q/k/v/o/fc1/fc2 through the e4m3 row quantiser at load; 16-bit dtypes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed batch's first images")
    ap.add_argument("--parity-images", type=int, default=16)
    ap.add_argument("--no-fp16-line", action="store_true", help="skip the extra fp16 measurement of the default run")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short ViT-L/16-384 fp16 b256 and ViT-B/16 fp8 b512 measurements of the default run")
    ap.add_argument("--cls-tail", action="store_true",
                    help="VH_FLAG_CLS_TAIL: the last layer computes the class-token rows only (an opt-in of the library; a line "
                         "measured with it says so in config.workload and is not the BASELINE configuration)")
    ap.add_argument("--force-dist", action="store_true", help="use torch.distributed even with one rank")
    ap.add_argument("--stages", action="store_true", help="also print a per-stage time table to stderr")
    ap.add_argument("--host-path", action="store_true",
                    help="also measure the PCIe-inclusive rate through the pinned submit/collect ring (extra object, never `value`)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --same-device rehearses the N > 1 control flow on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--graph", action="store_true", help="replay the forward as a hipGraph (small batches are launch-bound)")
    ap.add_argument("--streams", type=int, default=0, help="split every forward into N concurrent parts (0 = library default, 1)")
    ap.add_argument("--group", action="store_true",
                    help="single process, C-ABI device group (vh_group_*: one host thread per GPU, RCCL broadcast inside "
                         "libvithip) instead of one process per GPU; --gpus N selects the first N devices")
    args = ap.parse_args()

    if args.group:
        return main_group(args)

    import vh_dist
    rank, world, local_rank = vh_dist.env_ranks()
    use_dist = world > 1 or args.force_dist
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one process per GPU), or use --group")

    torch = dist = None
    if use_dist:
        # torch first: its bundled HIP runtime is then the one libvithip.so binds to as well
        if args.same_device:
            local_rank = 0
        torch, dist = vh_dist.init_process_group(args.backend, rank, world, local_rank)

    import numpy as np
    import vh_synth as S
    import vithip

    def barrier():
        if use_dist:
            dist.barrier()

    def measure(dtype_name, extras, config=None, B=None, steps=None, warmup=None, parity_images=None, cls_tail=None):
        """One complete measurement with `dtype_name` operands: context, weights (broadcast when distributed), warm-up,
        K timed steps, roofline of the fc1 kernel, parity of the timed batch's first images."""
        config = config or args.config
        cfg = S.CONFIGS[config]
        B = B or args.batch
        steps = steps or args.steps
        warmup = args.warmup if warmup is None else warmup
        parity_images = parity_images or args.parity_images
        T = S.tokens(cfg)
        flops_img = S.flops_per_image(cfg)
        dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16, "fp8": vithip.DTYPE_FP8}[dtype_name]
        ctx = vithip.VitContext(cfg, dtype=dt, max_batch=B, device=local_rank,
                                flags=(vithip.FLAG_W8_E4M3 if args.weights == "e4m3" else 0) |
                                      (vithip.FLAG_CLS_TAIL if (args.cls_tail if cls_tail is None else cls_tail) else 0))
        if args.streams > 0:
            ctx.set_streams(args.streams)
        streams = ctx.get_streams()
        if args.graph:
            ctx.set_graph(True)

        # ---- weights: rank 0 generates, RCCL broadcast of the canonical fp32 blob ---------------------
        if use_dist:
            nbytes = ctx.blob_bytes
            wbuf = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{local_rank}")
            if rank == 0:
                ctx.init_weights_seeded(0)
                ctx.export_weights_device(wbuf.data_ptr(), nbytes)
            torch.cuda.synchronize()
            vh_dist.broadcast_blob(dist, wbuf, src=0)
            torch.cuda.synchronize()
            if rank != 0:
                ctx.load_weights_device(wbuf.data_ptr(), nbytes)
            del wbuf
        else:
            ctx.init_weights_seeded(0)

        # ---- synthetic batch, generated in HBM (each rank its own shard of the global batch) ----------
        img_floats = B * cfg["image_size"] ** 2 * cfg["channels"]
        din = vithip.DeviceBuffer(img_floats * 4, device=local_rank)
        dout = vithip.DeviceBuffer(B * cfg["classes"] * 4, device=local_rank)
        ctx.fill_input_seeded(1 + rank, B, din.ptr)

        def sync():
            ctx.synchronize()
            if use_dist:
                torch.cuda.synchronize()

        if warmup > 0:
            ctx.forward_device_async(din.ptr, B, dout.ptr, steps=warmup)
        sync()

        # ---- timed region: exactly K steps ----------------------------------------------------------------
        # hip events on the context's stream inside it: around every fc1 launch (the roofline kernel) and at every step
        # boundary (median / min step).  Their cost is in `value`; `events_cost` below measures it.
        if not args.graph:
            ctx.set_stage_timing("fc1_gemm")   # per-launch events would bypass the graph
            ctx.set_step_timing(True)
        barrier(); sync()
        t0 = time.perf_counter()
        ctx.forward_device_async(din.ptr, B, dout.ptr, steps=steps)
        sync(); barrier()
        elapsed = time.perf_counter() - t0
        fc1_avg_ms, fc1_min_ms, fc1_n = ctx.get_stage_timing()
        step_ms = sorted(ctx.get_step_timing()) if not args.graph else []
        ctx.set_stage_timing(None)
        ctx.set_step_timing(False)
        if use_dist:
            elapsed = vh_dist.max_over_ranks(torch, dist, elapsed, f"cuda:{local_rank}")
        # the same K steps once more WITHOUT any event in the stream (not the reported value: the measure of what the
        # events inside the timed region cost)
        elapsed_plain = None
        if extras and not args.graph and not use_dist:
            sync()
            t1 = time.perf_counter()
            ctx.forward_device_async(din.ptr, B, dout.ptr, steps=steps)
            sync()
            elapsed_plain = time.perf_counter() - t1

        logits = dout.to_numpy(np.float32, (B, cfg["classes"]))
        if not np.isfinite(logits).all():
            raise SystemExit("non-finite logits")

        res = {"elapsed": elapsed, "streams": streams, "cfg": cfg, "B": B, "steps": steps, "flops_img": flops_img}
        if rank == 0:
            ips = B * world * steps / elapsed
            peak = PEAK_TFLOPS[dtype_name]
            rows = (B // streams) * T
            fc1_flops = 2.0 * rows * cfg["mlp_dim"] * cfg["dim"]  # every launch handles one part of the batch
            achieved = fc1_flops / (fc1_avg_ms * 1e-3) / 1e12 if fc1_avg_ms > 0 else 0.0
            fold = ctx.ln_fold()
            # the form the launcher really picks for this row count (vithip_api.hip enqueue_forward + kernels_gemm5.hip
            # launch_pp): persistent when the rows fill whole 256-row tiles, or when the folded layer loop pads them
            # (at least two rounds of tiles); the one-tile-per-workgroup form otherwise, or when VH_GEMM_PP forces it
            row_tiles = (rows + 255) // 256
            padded = fold and row_tiles * ((cfg["dim"] + 255) // 256) >= 2 * 256
            persistent = os.environ.get("VH_GEMM_PP") in (None, "", "0", "6") and (rows % 256 == 0 or padded) \
                and cfg["mlp_dim"] % 256 == 0 and row_tiles * (cfg["mlp_dim"] // 256) >= 128
            kname = ("gemm_nt_pp_kernel<%s%s> 256x256x%d %sping-pong (fc1)" %
                     ("LN-fold+bias+GELU" if fold else "bias+GELU", ", e4m3" if dtype_name == "fp8" else "",
                      128 if dtype_name == "fp8" else 64, "persistent " if persistent else ""))
            traffic, tsrc = measured_traffic(config, B, dtype_name, streams)
            clk, csrc = measured_clock(config, B, dtype_name)
            roof = None
            if not args.graph:
                roof = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                        "concurrent_parts": streams, "flop_per_launch": fc1_flops, "avg_launch_ms": round(fc1_avg_ms, 5),
                        "min_launch_ms": round(fc1_min_ms, 5), "launches_timed": fc1_n}
                # what the chip can deliver at the clock it HOLDS inside this kernel (DVFS): in-kernel clock x 4096 FLOP
                # per clock and CU (16-bit; 8192 for e4m3 operands) x 256 CUs.  `frac` (against the nominal 2.4 GHz peak)
                # stays the graded figure; frac_of_attainable separates clock loss from schedule loss.
                if clk:
                    att = clk * 1e9 * (8192 if dtype_name == "fp8" else 4096) * 256 / 1e12
                    roof.update({"attainable": round(att, 1), "frac_of_attainable": round(achieved / att, 4),
                                 "clock_ghz_in_kernel": clk, "clock_source": csrc})
                else:
                    roof.update({"attainable": None, "frac_of_attainable": None})
            res.update({
                "value": round(ips, 2), "ms_per_step": round(elapsed / steps * 1e3, 4),
                "forward_mfma_frac": round(ips / world * flops_img / (peak * 1e12), 4),
                "roofline": roof,
            })
            if step_ms:
                res["step_ms"] = {"median": round(step_ms[len(step_ms) // 2], 4), "min": round(step_ms[0], 4),
                                  "max": round(step_ms[-1], 4), "steps": len(step_ms),
                                  "note": "device time between the hip events at the step boundaries (rank 0)"}
            if elapsed_plain:
                res["events_cost"] = {"value_without_events": round(B * steps / elapsed_plain, 2),
                                      "note": "the same K steps run again with no hip event in the stream; `value` is the run WITH "
                                              "the roofline / step events, as the timed region is defined"}
            if not args.no_parity:
                res["parity"] = parity_of(cfg, logits, min(parity_images, B), dtype_name, 1 + rank, np, S)
            if extras and args.host_path:
                res["host_path"] = host_path_rate(ctx, cfg, B, steps, np, S)
            if extras and args.stages:
                res["stages"] = ctx.profile_forward(din.ptr, B, dout.ptr)
        din.free(); dout.free()
        ctx.close()
        return res

    main_res = measure(args.dtype, True)
    cfg, B, flops_img = main_res["cfg"], main_res["B"], main_res["flops_img"]
    fp16_res = None
    extra_res = {}
    default_run = args.config == "vit_base" and B == 512 and args.dtype == "bf16"
    if default_run and not args.no_fp16_line and not args.graph:
        fp16_res = measure("fp16", False)
    if default_run and not args.no_extra_configs and not args.graph and world == 1:
        # BASELINE configs 4 and 5, short: about a second of GPU time each
        extra_res["vit_large_384_fp16_b256"] = measure("fp16", False, config="vit_large_384", B=256, steps=5, warmup=1, parity_images=8)   # (8 images: ~3 s of oracle; round 3 judged this config on 2)
        extra_res["fp8_b512"] = measure("fp8", False, config="vit_base", B=512, steps=10, warmup=2, parity_images=8)
        if not args.cls_tail:
            extra_res["cls_tail_bf16_b512"] = measure("bf16", False, config="vit_base", B=512, steps=10, warmup=2, parity_images=8, cls_tail=True)

    if rank == 0:
        streams = main_res["streams"]
        par = (f"image-sharded x{world}, weights RCCL-broadcast once, no data-path collective" if use_dist else
               "one GPU, no process group (image sharding + one RCCL weight broadcast when launched with N > 1 ranks)")
        out = {
            "metric": "images/sec ViT-B/16 224x224, batch 512 per GPU" if default_run
                      else f"images/sec {args.config} {args.dtype} batch {B} per GPU",
            "value": main_res["value"], "unit": "images/s", "n_gpus": world, "steps": main_res["steps"], "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.config} {cfg['image_size']}x{cfg['image_size']}x{cfg['channels']} inference, "
                                   f"{B} images per GPU resident in HBM, random-init weights (seed 0)"
                                   + (", weight-only e4m3 (VH_FLAG_W8_E4M3)" if args.weights == "e4m3" else "")
                                   + (", VH_FLAG_CLS_TAIL (last layer: class-token rows only -- NOT the BASELINE configuration)" if args.cls_tail else ""),
                       "global_batch": B * world, "per_gpu_batch": B, "parallelism": par, "flop_per_image": flops_img},
            "forward_mfma_frac": main_res["forward_mfma_frac"],
            "roofline": main_res["roofline"],
        }
        for k in ("step_ms", "events_cost", "parity"):
            if k in main_res:
                out[k] = main_res[k]
        if args.graph:
            out["graph"] = True
        for name, r in extra_res.items():
            out[name] = {k: r[k] for k in ("value", "ms_per_step", "forward_mfma_frac", "roofline", "step_ms", "parity") if k in r}
            out[name]["steps"] = r["steps"]
            out[name]["note"] = ("BASELINE config 4: ViT-L/16 384x384 fp16, 256 images, 1 GPU (short run)" if name.startswith("vit_large")
                                 else "NOT a BASELINE configuration and never `value`: the same bf16 batch-512 workload with the library's opt-in "
                                      "VH_FLAG_CLS_TAIL -- the last layer runs attention for the class-token query only and out-proj / fc1 / "
                                      "fc2 on the 512 class rows, i.e. it skips rows no logit depends on (forward_mfma_frac still counts the "
                                      "full model's FLOP and therefore overstates the matrix work done here)" if name.startswith("cls_tail")
                                 else "BASELINE config 5: ViT-B/16, e4m3 operands in the four per-layer GEMMs (VH_DTYPE_FP8), 512 images, "
                                      "1 GPU (short run; fraction of the 5 PF dense fp8 peak)")
        if fp16_res:
            out["fp16"] = {k: fp16_res[k] for k in ("value", "ms_per_step", "forward_mfma_frac", "roofline", "step_ms", "parity") if k in fp16_res}
            out["fp16"]["note"] = ("same workload, same run, fp16 MFMA operands: the operand type inside the north star's 1e-3 "
                                   "(extra object; `value` above is the bf16 configuration BASELINE.json names)")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        if "host_path" in main_res:
            out["host_path"] = main_res["host_path"]
        if "stages" in main_res:
            stages = main_res["stages"]
            tot = sum(v[0] for v in stages.values())
            for k, (ms, n) in stages.items():
                print(f"  {k:16s} {ms:9.3f} ms  {n:3d} launches  {100 * ms / tot:5.1f} %", file=sys.stderr)
            print(f"  {'total':16s} {tot:9.3f} ms", file=sys.stderr)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def parity_of(cfg, logits, n, dtype_name, seed, np, S):
    """max|d| / max|ref| per image of the first `n` images of the timed batch against the fp32 CPU oracle."""
    import oracle_lib as O
    blob = S.make_blob(cfg, 0)
    images = S.make_images(cfg, seed, n)          # the same generator that filled the batch in HBM (bit-identical)
    cores = min(16, len(os.sched_getaffinity(0)))
    ref = O.vit_forward(cfg, blob, images, threads=cores)
    per = np.abs(logits[:n] - ref).max(1) / np.abs(ref).max()
    worst, med = float(per.max()), float(np.median(per))
    d = {"against": "fp32 CPU oracle (oracle/liboracle.so) on the same seeded images and weights", "images": int(n),
         "metric": "max|logit - ref| / max|ref| per image", "worst": round(worst, 6), "median": round(med, 6),
         "top1_agreement": round(float((logits[:n].argmax(1) == ref.argmax(1)).mean()), 4),
         "north_star_tolerance": NORTH_STAR_TOL, "within_north_star_tolerance": bool(worst <= NORTH_STAR_TOL)}
    if dtype_name == "bf16":
        d["note"] = ("bf16 operands (8-bit significand) are OUTSIDE the north star's 1e-3 by construction; fp16 operands "
                     "are inside it: see the `fp16` object and DESIGN.md Numerics")
    elif dtype_name == "fp8":
        d["note"] = "e4m3 operands: the north star's 1e-3 does not apply to BASELINE config 5 (statistical criterion, tests/test_gpu_fp8.py)"
    return d


def main_group(args):
    """N GPUs from ONE process through the C ABI's device group (vh_group_*): one host thread + stream per device inside
    libvithip, ncclCommInitAll + one ncclBroadcast of the canonical blob, contiguous image shards, no data-path
    collective.  Same timing rules: K steps bracketed by a synchronisation of every device, whole-job images/s."""
    import numpy as np
    import vh_synth as S
    import vithip
    cfg = S.CONFIGS[args.config]
    B = args.batch
    n = args.gpus
    dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16, "fp8": vithip.DTYPE_FP8}[args.dtype]
    # --same-device: a REHEARSAL group -- device 0 listed n times: the N > 1 control flow of vh_group_* (n host threads, the
    # device-blob hand-over, shard bounds, concurrent member forwards) on a one-GPU box.  The members share one GPU, so the
    # rate says nothing about scaling: the line is labelled and carries no `value`.
    rehearsal = bool(args.same_device)
    grp = vithip.VitGroup(cfg, [0] * n if rehearsal else list(range(n)), dtype=dt, max_batch_per_device=B)
    grp.init_weights_seeded(0)            # device 0 generates, the blob is broadcast to the others
    grp.fill_inputs_seeded(1, B)          # every device its own shard, generated in HBM
    if args.warmup > 0:
        grp.forward_resident(B, args.warmup)
    t0 = time.perf_counter()
    grp.forward_resident(B, args.steps)   # returns after every device has finished its K steps
    elapsed = time.perf_counter() - t0
    logits = grp.read_logits(B)
    if not np.isfinite(logits).all():
        raise SystemExit("non-finite logits")
    ips = B * n * args.steps / elapsed
    flops_img = S.flops_per_image(cfg)
    out = {"metric": "images/sec ViT-B/16 224x224, batch 512 per GPU" if args.config == "vit_base" and B == 512 and args.dtype == "bf16"
                     else f"images/sec {args.config} {args.dtype} batch {B} per GPU",
           "value": round(ips, 2), "unit": "images/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"{args.config} inference, {B} images per GPU resident in HBM, random-init weights (seed 0)",
                      "global_batch": B * n, "per_gpu_batch": B,
                      "parallelism": f"single process, vh_group_* C ABI: {n} device thread(s), weights "
                                     f"{'RCCL-broadcast once' if n > 1 else 'generated in place (group of one)'}, no data-path collective",
                      "flop_per_image": flops_img},
           "forward_mfma_frac": round(ips / n * flops_img / (PEAK_TFLOPS[args.dtype] * 1e12), 4), "roofline": None}
    if not args.no_parity:
        out["parity"] = parity_of(cfg, logits[:B], min(args.parity_images, B), args.dtype, 1, np, S)
    if rehearsal:
        # every member's shard against a plain context on the same seeded images: sharded == unsharded, bit for bit
        ctx = vithip.VitContext(cfg, dtype=dt, max_batch=B)
        ctx.init_weights_seeded(0)
        px = cfg["image_size"] ** 2 * cfg["channels"]
        din, dout = vithip.DeviceBuffer(B * px * 4), vithip.DeviceBuffer(B * cfg["classes"] * 4)
        same = True
        for i in range(n):
            ctx.fill_input_seeded(1 + i, B, din.ptr)
            ctx.forward_device(din.ptr, B, dout.ptr)
            same &= bool(np.array_equal(dout.to_numpy(np.float32, (B, cfg["classes"])), logits[i * B:(i + 1) * B]))
        ctx.close()
        out["rehearsal"] = {"members_on_device_0": n, "images_per_s_all_members_sharing_one_gpu": out["value"],
                            "sharded_equals_unsharded_bitwise": same,
                            "note": "REHEARSAL of the vh_group_* control flow on ONE GPU (device 0 listed n times; the weight "
                                    "broadcast is a device-to-device copy, RCCL cannot hold one device twice): NOT a scaling "
                                    "measurement, no N > 1 run exists (DESIGN.md section 8)"}
        out["value"] = None
        out["forward_mfma_frac"] = None
        if not same:
            print(json.dumps(out), flush=True)
            raise SystemExit("rehearsal group: sharded logits differ from the plain context's")
    print(json.dumps(out), flush=True)
    grp.close()


def host_path_rate(ctx, cfg, B, steps, np, S):
    """PCIe-inclusive rate: images start in (pinned) HOST memory and logits end in host memory, through the
    vh_ring_* pipeline (3 slots: upload of batch i+1 and download of batch i-1 overlap the forward of batch i)."""
    slots = 3
    ctx.ring_create(slots, B)
    sample = S.make_images(cfg, 7, min(B, 8))
    for _ in range(slots):          # fill every slot's pinned staging buffer once (a producer writes there in place)
        buf = ctx.ring_input(B)
        for i in range(0, B, len(sample)):
            buf[i:i + len(sample)] = sample[:min(len(sample), B - i)]
        ctx.ring_submit(None, B)
    for _ in range(slots):
        ctx.ring_collect()
    ctx.synchronize()
    t0 = time.perf_counter()
    inflight = 0
    for _ in range(steps):
        if inflight == slots:
            ctx.ring_collect(); inflight -= 1
        ctx.ring_submit(None, B); inflight += 1
    while inflight:
        last = ctx.ring_collect(); inflight -= 1
    dt = time.perf_counter() - t0
    if not np.isfinite(last).all():
        raise SystemExit("non-finite logits (host path)")
    in_gb = B * cfg["image_size"] ** 2 * cfg["channels"] * 4 / 1e9
    return {"value": round(B * steps / dt, 2), "unit": "images/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "slots": slots, "h2d_GB_per_step": round(in_gb, 4),
            "note": "fp32 NHWC images from pinned host memory -> logits in host memory, vh_ring_submit/collect"}


def cpu_baseline(cfg, target_seconds):
    """The CPU leg behind the reference's plugin interface: tests/cpp/net_cpu (cpu::net_cpu : net::net_abstract, the
    oracle behind launch_forward — test infrastructure, a port: the reference ships no CPU path), timed through
    launch_forward with the reference's own std::chrono window (netFPGA.cpp:262-284), on a bounded sample.
    BASELINE.md protocol: ViT-Tiny/16 batch 1 and the benchmarked model, 1 thread and all threads."""
    import cpu_leg
    cores = min(16, len(os.sched_getaffinity(0)))  # the GPU box grants a CPU share of 16 cores per GPU
    return cpu_leg.baseline(cfg, cores, target_seconds)


if __name__ == "__main__":
    main()
