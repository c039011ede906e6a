"""bench.py reports roofline.traffic only from a PMC file that belongs to the tree's kernel sources and to the workload
being run; anything else is null (no stale constants).  CPU-only: exercises the two helpers, no device."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.fixture
def traffic_file(monkeypatch, tmp_path):
    """A scratch repo root holding one traffic file stamped with the real tree's hash."""
    (tmp_path / "profiles").mkdir()
    sha = bench.source_sha()
    rec = {"source_sha": sha, "config": "vit_base", "batch": 512, "dtype": "bf16", "hbm_bytes_per_launch": 1.25e9}
    path = tmp_path / "profiles" / "zz_fc1_traffic.json"
    path.write_text(json.dumps(rec))
    real_sha = bench.source_sha
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_sha", lambda: sha)
    return path, rec, real_sha


def test_traffic_is_reported_for_the_matching_tree_and_workload(traffic_file):
    path, rec, _ = traffic_file
    got, src = bench.measured_traffic("vit_base", 512, "bf16", 1)
    assert got == rec["hbm_bytes_per_launch"] and src.endswith("zz_fc1_traffic.json")


@pytest.mark.parametrize("change", [{"source_sha": "0" * 16}, {"batch": 256}, {"dtype": "fp16"}, {"config": "vit_tiny"}])
def test_traffic_is_null_for_another_tree_or_workload(traffic_file, change):
    path, rec, _ = traffic_file
    path.write_text(json.dumps({**rec, **change}))
    assert bench.measured_traffic("vit_base", 512, "bf16", 1) == (None, None)


def test_traffic_is_null_with_concurrent_parts_and_for_a_damaged_file(traffic_file):
    path, rec, _ = traffic_file
    assert bench.measured_traffic("vit_base", 512, "bf16", 2) == (None, None)   # a launch no longer isolates one kernel
    path.write_text("{ not json")
    assert bench.measured_traffic("vit_base", 512, "bf16", 1) == (None, None)


def test_the_source_hash_follows_the_kernel_sources(tmp_path, monkeypatch):
    real = bench.source_sha()
    assert len(real) == 16 and real == bench.source_sha()
    # a copy of the tree's kernel sources with one byte appended to one of them hashes differently
    import shutil
    for sub in ("vit-fpga_amd/csrc", "include"):
        os.makedirs(tmp_path / sub)
    for f in os.listdir(os.path.join(ROOT, "vit-fpga_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            shutil.copy(os.path.join(ROOT, "vit-fpga_amd", "csrc", f), tmp_path / "vit-fpga_amd" / "csrc" / f)
    shutil.copy(os.path.join(ROOT, "include", "vithip.h"), tmp_path / "include" / "vithip.h")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.source_sha() == real
    with open(tmp_path / "vit-fpga_amd" / "csrc" / "kernels_attn.hip", "a") as fh:
        fh.write("\n")
    assert bench.source_sha() != real
