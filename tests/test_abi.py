"""CPU suite: the C-ABI library loads here (no GPU) and exports exactly what include/vithip.h
declares; host-side argument checking works; compute entry points fail loudly without a device
(there is no CPU fallback in the product path)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import vh_synth as S
import vithip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vithip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", vithip.LIB_PATH], text=True)
    exported = set(re.findall(r" T (vh_[a-z0-9_]+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert set(names) == set(vithip.SYMBOLS), sorted(set(names) ^ set(vithip.SYMBOLS))
    assert vithip.lib().vh_abi_version() == 1


def test_header_is_plain_c_and_the_c_example_builds(tmp_path):
    # the boundary is a C ABI: the header must compile as C99 with warnings as errors, and the plain-C example links
    hdr = os.path.join(ROOT, "include")
    probe = tmp_path / "probe.c"
    probe.write_text('#include "vithip.h"\nint main(void) { return vh_abi_version() == VH_ABI_VERSION ? 0 : 1; }\n')
    lib_dir = os.path.dirname(vithip.LIB_PATH)
    for src, exe in ((str(probe), tmp_path / "probe"), (os.path.join(ROOT, "examples", "classify.c"), tmp_path / "classify")):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", hdr, src, "-L", lib_dir,
                               "-lvithip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)])
    assert subprocess.call([str(tmp_path / "probe")]) == 0


def test_tools_and_entry_points_compile():
    # the measurement scripts are part of the evidence trail (profiles/README.md names them): keep them importable
    import py_compile
    for rel in ("bench.py", "__graft_entry__.py", "tools/soak.py", "tools/parity_stats.py", "tools/pmc_summary.py",
                "tools/gemm_bench.py", "tools/tiles_exp.py", "tools/fold_parity.py", "tests/golden/make_golden.py",
                "tools/gemm_anatomy.py", "tools/torch_matmul_calib.py", "tools/parity_attribution.py", "tools/pmc_traffic.py",
                "tests/cpu_leg.py", "tools/attn_bench.py", "tools/attn_anatomy.py"):
        py_compile.compile(os.path.join(ROOT, rel), doraise=True)
    for rel in ("tools/power_probe.sh", "tools/ab_variant7.sh", "tools/ab_tail.sh", "tools/ab_supercol.sh", "tools/ab_attn.sh",
                "tools/pmc_passes.sh", "tools/pmc_attn.sh", "tools/ab_attn_abl.sh", "tools/collect_round_evidence.sh", "tools/ab_supercol_traffic.sh"):
        subprocess.check_call(["bash", "-n", os.path.join(ROOT, rel)])


def test_no_torch_or_oracle_dependency_in_the_product_library():
    out = subprocess.check_output(["readelf", "-d", vithip.LIB_PATH], text=True)
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any(("torch" in n) or ("oracle" in n) or ("c10" in n) or ("rccl" in n) for n in needed), needed   # RCCL: dlopen, groups of > 1 only
    src = os.path.join(ROOT, "vit-fpga_amd")
    for dirpath, _, files in os.walk(src):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".py")):
                t = open(os.path.join(dirpath, f)).read()
                for forbidden in ("liboracle", "oracle_lib", "oracle.h", "oracle_vit_", "oracle_mlp_"):
                    assert forbidden not in t, (f, forbidden)


def test_blob_size_and_config_validation_without_device():
    L = vithip.lib()
    for name in ("vit_micro", "vit_tiny", "vit_base", "vit_large_384"):
        cfg = S.CONFIGS[name]
        c = vithip.make_config(cfg, max_batch=4)
        assert L.vh_weight_blob_bytes(C.byref(c)) == 64 + 4 * S.param_count(cfg)
    bad = vithip.make_config(dict(S.CONFIGS["vit_micro"], dim=100))
    assert L.vh_weight_blob_bytes(C.byref(bad)) == 0
    bad = vithip.make_config(dict(S.CONFIGS["vit_micro"], classes=41))
    assert L.vh_weight_blob_bytes(C.byref(bad)) == 0


@pytest.mark.skipif(vithip.device_count() > 0, reason="checks the no-device behaviour")
def test_compute_entry_points_fail_loudly_without_a_gpu():
    with pytest.raises(vithip.VhError) as e:
        vithip.VitContext(S.CONFIGS["vit_micro"])
    assert e.value.code == 5  # VH_ERR_NO_DEVICE
    with pytest.raises(vithip.VhError):
        vithip.MlpContext(4, [3, 2])
    with pytest.raises(vithip.VhError):
        vithip.DeviceBuffer(1024)
    with pytest.raises(vithip.VhError):
        vithip.op_fill(None, 16, 1, 1, 0)


def test_host_side_16bit_helpers():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 0.0], dtype=np.float32)
    assert np.array_equal(vithip.from16(vithip.to16(x, vithip.DTYPE_BF16), vithip.DTYPE_BF16),
                          np.array([1.0, 1.0, 1.015625, -2.5, 0.0], dtype=np.float32))
    assert np.array_equal(vithip.from16(vithip.to16(x, vithip.DTYPE_FP16), vithip.DTYPE_FP16)[:2],
                          np.array([1.0, 1.00390625], dtype=np.float32))


def test_group_shard_bounds_match_the_multi_process_sharding():
    """vh_group_shard_bounds (C ABI, single process) and vh_dist.shard_bounds (one process per GPU) split a batch the same
    way; together the shards cover the batch exactly once."""
    import vh_dist
    for batch, n in ((4096, 8), (10, 3), (2, 4), (7, 7), (1, 1), (513, 2)):
        cover = []
        for r in range(n):
            lo, hi = vithip.group_shard_bounds(batch, n, r)
            assert (lo, hi) == vh_dist.shard_bounds(batch, n, r)
            cover += list(range(lo, hi))
        assert cover == list(range(batch))


def test_group_create_without_a_device_fails_loudly():
    if vithip.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(vithip.VhError) as e:
        vithip.VitGroup(S.CONFIGS["vit_micro"], [0, 1])
    assert e.value.code == 5   # VH_ERR_NO_DEVICE
