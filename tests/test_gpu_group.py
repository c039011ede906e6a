"""The C-ABI device group (vh_group_*: N GPUs of one node from one process) and the weight hand-over a non-zero rank
performs in the multi-process form.  A one-GPU box can run: a group of one, a REHEARSAL group (GPU 0 listed several
times: the same host threads, image shards and device-blob path, the broadcast being a device-to-device copy), and RCCL
on a single-rank communicator (dlopen + ncclCommInitAll + ncclBroadcast).  N > 1 distinct devices: the driver's run."""
import numpy as np
import pytest

import vh_synth as S

pytestmark = pytest.mark.gpu

vithip = pytest.importorskip("vithip")


def plain(cfg, blob, images, dtype):
    ctx = vithip.VitContext(cfg, dtype=dtype, max_batch=len(images))
    ctx.load_weights(blob)
    out = ctx.forward(images)
    ctx.close()
    return out


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0]])
def test_group_forward_equals_one_context_bitwise(devices):
    cfg = S.CONFIGS["vit_mini"]
    blob, images = S.make_blob(cfg, 3), S.make_images(cfg, 4, 7)
    want = plain(cfg, blob, images, vithip.DTYPE_BF16)
    g = vithip.VitGroup(cfg, devices, dtype=vithip.DTYPE_BF16, max_batch_per_device=7)
    assert g.size() == len(devices)
    g.load_weights(blob)                      # member 0 uploads, the others receive the canonical blob and convert it
    assert np.array_equal(g.forward(images), want)              # ragged shards (7 over 2 / 4 members)
    assert np.array_equal(g.forward(images[:1]), want[:1])      # fewer images than members
    assert np.array_equal(g.forward(images[2:5]), want[2:5])
    with pytest.raises(vithip.VhError):
        g.forward(S.make_images(cfg, 4, 7 * len(devices) + 1))  # beyond the per-device capacity
    g.close()


def test_rehearsal_group_of_eight_members_shards_batch_512_bitwise():
    """The driver's 8-GPU shape rehearsed on one GPU: eight members (device 0 eight times -- eight host threads, eight
    contexts, eight device-blob hand-overs), ViT-B/16, batch 8 x 64 = 512 images given as ONE host batch; the group's
    logits equal the unsharded batch-512 forward of one context bit for bit.  (No N > 1 run on distinct GPUs exists.)"""
    cfg = S.CONFIGS["vit_base"]
    n, per = 8, 64
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=n * per)
    ctx.init_weights_seeded(0)
    blob = ctx.export_weights()
    px = cfg["image_size"] ** 2 * cfg["channels"]
    din, dout = vithip.DeviceBuffer(n * per * px * 4), vithip.DeviceBuffer(n * per * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, n * per, din.ptr)
    ctx.forward_device(din.ptr, n * per, dout.ptr)
    want = dout.to_numpy(np.float32, (n * per, cfg["classes"]))
    images = din.to_numpy(np.float32, (n * per, cfg["image_size"], cfg["image_size"], cfg["channels"]))
    ctx.close()
    g = vithip.VitGroup(cfg, [0] * n, dtype=vithip.DTYPE_BF16, max_batch_per_device=per)
    assert g.size() == n
    g.load_weights(blob)
    got = g.forward(images)
    for r in range(n):                       # the shard bounds the group used
        lo, hi = vithip.group_shard_bounds(n * per, n, r)
        assert (lo, hi) == (r * per, (r + 1) * per)
    g.close()
    assert np.isfinite(want).all() and np.array_equal(got, want)


def test_group_seeded_weights_and_resident_path():
    cfg = S.CONFIGS["vit_tiny"]
    B = 3
    g = vithip.VitGroup(cfg, [0, 0], dtype=vithip.DTYPE_FP16, max_batch_per_device=B)
    g.init_weights_seeded(5)
    g.fill_inputs_seeded(1, B)                # member i generates the shard of "rank" i: seed 1 + i
    g.forward_resident(B, steps=2)
    got = g.read_logits(B)
    g.close()
    blob = S.make_blob(cfg, 5)
    for i in range(2):
        want = plain(cfg, blob, S.make_images(cfg, 1 + i, B), vithip.DTYPE_FP16)
        assert np.array_equal(got[i * B:(i + 1) * B], want), i


def test_group_of_one_through_rccl(monkeypatch):
    """VH_GROUP_FORCE_RCCL=1: even a group of one binds librccl.so, creates its communicator and sends the blob through
    ncclBroadcast (root = only rank): the RCCL calls of the N > 1 path, on the hardware that is available here."""
    monkeypatch.setenv("VH_GROUP_FORCE_RCCL", "1")
    cfg = S.CONFIGS["vit_micro"]
    blob, images = S.make_blob(cfg, 11), S.make_images(cfg, 12, 3)
    g = vithip.VitGroup(cfg, [0], dtype=vithip.DTYPE_BF16, max_batch_per_device=3)
    g.load_weights(blob)
    got = g.forward(images)
    g.close()
    assert np.array_equal(got, plain(cfg, blob, images, vithip.DTYPE_BF16))


def test_group_of_two_distinct_devices_through_rccl():
    """The real N > 1 path: two DISTINCT devices, ncclCommInitAll + grouped ncclBroadcast over xGMI, one host thread per
    device.  Skips -- explicitly, it does not pass as a rehearsal -- on a box with fewer than two GPUs; until a multi-GPU
    run exists the RCCL broadcast between distinct devices is unverified on hardware."""
    if vithip.device_count() < 2:
        pytest.skip("needs 2 GPUs: the RCCL broadcast between distinct devices has never run on hardware (one-GPU boxes)")
    cfg = S.CONFIGS["vit_mini"]
    blob, images = S.make_blob(cfg, 3), S.make_images(cfg, 4, 7)
    want = plain(cfg, blob, images, vithip.DTYPE_BF16)
    g = vithip.VitGroup(cfg, [0, 1], dtype=vithip.DTYPE_BF16, max_batch_per_device=7)
    g.load_weights(blob)
    assert np.array_equal(g.forward(images), want)
    g.close()


def test_non_zero_rank_weight_hand_over_then_shard_forward():
    """What rank r > 0 of the one-process-per-GPU form does (bench.py, vh_dist): rank 0 exports the canonical blob into a
    device buffer, the buffer is broadcast (here: used as is), the other rank calls vh_load_weights_device on it and runs
    its own contiguous shard.  Sharded == unsharded, bit for bit."""
    cfg = S.CONFIGS["vit_tiny"]
    images = S.make_images(cfg, 1, 6)
    a = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=6)
    a.init_weights_seeded(0)
    full = a.forward(images)
    wire = vithip.DeviceBuffer(a.blob_bytes)
    a.export_weights_device(wire.ptr, a.blob_bytes)
    b = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=3)
    b.load_weights_device(wire.ptr, a.blob_bytes)
    wire.free()
    for r in range(2):
        lo, hi = vithip.group_shard_bounds(6, 2, r)
        ctx = a if r == 0 else b
        assert np.array_equal(ctx.forward(images[lo:hi]), full[lo:hi]), r
    assert np.array_equal(b.export_weights(), a.export_weights())
    a.close(); b.close()
