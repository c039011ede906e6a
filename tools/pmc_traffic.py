#!/usr/bin/env python3
"""Write profiles/<tag>_fc1_traffic.json from the PMC passes of tools/pmc_passes.sh: HBM bytes per fc1 launch
(2 x FETCH_SIZE per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE), stamped with the hash of the kernel
sources it was measured on.  bench.py reports it as roofline.traffic only while that hash equals the tree's.

  python tools/pmc_traffic.py <tag> gpurun_out/pmc_<tag>_fetch gpurun_out/pmc_<tag>_write [--config vit_base --batch 512 --dtype bf16]
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_launch(d, counter, match):
    tot, n = 0.0, 0
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and match(r["Kernel_Name"]):
                tot += float(r["Counter_Value"])
                n += 1
    if not n:
        raise SystemExit(f"no {counter} rows for the fc1 kernel under {d}")
    return tot / n, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--config", default="vit_base")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    import bench
    # the fc1 kernel: ping-pong GEMM with the GELU epilogue (epilogue id 6 = LN-fold + GELU, 1 = bias + GELU)
    tn = {"bf16": "BF16", "fp16": "FP16"}[a.dtype]   # launches of the other storage type (bench.py's fp16 line) are left out
    is_fc1 = lambda k: "gemm_nt_pp_kernel" in k and (f"{tn}, 6" in k or f"{tn}, 1" in k)
    fetch_kib, n1 = per_launch(a.fetch_dir, "FETCH_SIZE", is_fc1)
    write_kib, n2 = per_launch(a.write_dir, "WRITE_SIZE", is_fc1)
    import vh_synth as S  # noqa: E402  (tests/ is on the path through bench)
    cfg = S.CONFIGS[a.config]
    rows = a.batch * S.tokens(cfg)
    algorithmic = rows * cfg["dim"] * 2 + cfg["mlp_dim"] * cfg["dim"] * 2 + rows * cfg["mlp_dim"] * 2
    out = {"kernel": "gemm_nt_pp_kernel<LN-fold+bias+GELU> (fc1)", "config": a.config, "batch": a.batch, "dtype": a.dtype,
           "fetch_size_bytes_raw": fetch_kib * 1024, "fetch_size_bytes_x2_corrected": 2 * fetch_kib * 1024,
           "write_size_bytes": write_kib * 1024, "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
           "algorithmic_bytes": algorithmic, "launches": [n1, n2], "source_sha": bench.source_sha(),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only (tools/pmc_passes.sh); "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section"}
    path = os.path.join(ROOT, "profiles", f"{a.tag}_fc1_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, json.dumps(out))


if __name__ == "__main__":
    main()
