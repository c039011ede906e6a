"""Weight blob on disk: header parsing is host-only (runs without a GPU); save / load round
trips run on the device."""
import numpy as np
import pytest

import vh_synth as S

vithip = pytest.importorskip("vithip")


def test_file_header_is_validated_on_the_host(tmp_path):
    cfg = S.CONFIGS["vit_micro"]
    blob = S.make_blob(cfg, 4)
    good = tmp_path / "micro.vhblob"
    blob.tofile(good)
    got, eps = vithip.blob_file_config(good)
    assert got == cfg and abs(eps - 1e-6) < 1e-12
    bad = blob.copy(); bad[0] ^= 0xFF
    bad.tofile(tmp_path / "magic.vhblob")
    blob[:-4].tofile(tmp_path / "short.vhblob")
    np.concatenate([blob, blob[:8]]).tofile(tmp_path / "long.vhblob")
    odd = blob.copy(); odd[8 + 12:8 + 16] = np.frombuffer(np.int32(100).tobytes(), np.uint8)   # dim = 100: unsupported
    odd.tofile(tmp_path / "dim.vhblob")
    # a crafted header: absurd layer / class / image counts must be refused from the header alone (no allocation is driven
    # by the file: the bounds of check_config apply before any size is computed), never abort the process
    for off, val in ((8 + 24, 2 ** 31 - 1), (8 + 24, 5000), (8 + 28, 2 ** 30), (8 + 0, 2 ** 20), (8 + 20, 2 ** 24)):
        crafted = blob.copy(); crafted[off:off + 4] = np.frombuffer(np.int32(val).tobytes(), np.uint8)
        crafted.tofile(tmp_path / f"crafted_{off}_{val}.vhblob")
        with pytest.raises(vithip.VhError):
            vithip.blob_file_config(tmp_path / f"crafted_{off}_{val}.vhblob")
    for name in ("magic", "short", "long", "dim", "missing"):
        with pytest.raises(vithip.VhError):
            vithip.blob_file_config(tmp_path / f"{name}.vhblob")
    # vh_blob_file_read: the file as a memory blob, checksum verified when present
    buf = np.empty(blob.size, np.uint8)
    assert vithip.lib().vh_blob_file_read(str(good).encode(), buf.ctypes.data, buf.size) == 0 and np.array_equal(buf, blob)
    assert vithip.lib().vh_blob_file_read(str(good).encode(), buf.ctypes.data, buf.size - 4) != 0


def test_weight_blob_size_formula_matches_the_layout():
    for name, cfg in S.CONFIGS.items():
        c = vithip.make_config(cfg)
        assert vithip.lib().vh_weight_blob_bytes(c) == 64 + 4 * S.param_count(cfg), name
    bad = vithip.make_config(S.CONFIGS["vit_micro"], flags=64)       # unknown flag bits are rejected
    assert vithip.lib().vh_weight_blob_bytes(bad) == 0


@pytest.mark.gpu
def test_save_load_round_trip_and_damage_detection(tmp_path):
    cfg = S.CONFIGS["vit_mini"]
    blob, images = S.make_blob(cfg, 6), S.make_images(cfg, 7, 3)
    a = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=3)
    a.load_weights(blob)
    want = a.forward(images)
    path = tmp_path / "mini.vhblob"
    a.save_weights_file(path)
    a.close()
    on_disk = np.fromfile(path, dtype=np.uint8)
    assert on_disk.size == blob.size and np.array_equal(on_disk[64:], blob[64:]) and np.array_equal(on_disk[:44], blob[:44])
    assert on_disk[52] == 1 and on_disk[44:52].any()              # flags bit 0 + a checksum
    shape, _ = vithip.blob_file_config(path)
    b = vithip.VitContext(shape, dtype=vithip.DTYPE_FP16, max_batch=3)
    b.load_weights_file(path)
    assert np.array_equal(b.forward(images), want)
    assert np.array_equal(b.export_weights(), blob)                # resident form = memory form (checksum words clear)
    # a blob written without checksum (memory form) loads too
    blob.tofile(tmp_path / "plain.vhblob")
    b.load_weights_file(tmp_path / "plain.vhblob")
    assert np.array_equal(b.forward(images), want)
    # damage: one flipped payload byte / wrong model / truncated
    hurt = on_disk.copy(); hurt[5000] ^= 1
    hurt.tofile(tmp_path / "hurt.vhblob")
    S.make_blob(S.CONFIGS["vit_micro"], 1).tofile(tmp_path / "other.vhblob")
    on_disk[:-8].tofile(tmp_path / "cut.vhblob")
    for name in ("hurt", "other", "cut", "nope"):
        with pytest.raises(vithip.VhError):
            b.load_weights_file(tmp_path / f"{name}.vhblob")
    assert np.array_equal(b.forward(images), want)                 # failed loads leave the context usable
    b.close()
