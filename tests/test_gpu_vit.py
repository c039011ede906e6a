"""GPU parity, end to end: the ViT forward of libvithip.so (through the C ABI) against the CPU
oracle and against the committed golden logits (tests/golden/*.npz, produced offline from an
independent fp64 implementation).

Tolerance.  The north star asks for logits within 1e-3 (relative, fp32 reference).  The metric used
here is  max|logit_gpu - logit_ref| / max|logit_ref|  over the whole batch.
  * fp16 MFMA operands (11-bit significand): asserted at 1e-3 = THE NORTH STAR'S TOLERANCE, on every model, on 32 ViT-B
    images with both LayerNorm paths, and on ViT-L/16 at 384x384.
  * bf16 MFMA operands (8-bit significand), the dtype BASELINE.json's headline config names: OUTSIDE the north star's
    tolerance by construction -- each GEMM input carries ~2^-9 relative rounding error and the measured distance is
    4e-3..7.5e-3, eight times the fp16 figure, exactly the ratio of the two significands -- even with the fp32
    residual stream, fp32 LayerNorm/softmax statistics, fp32 accumulation and the fp32 classifier head this build
    keeps.  Those cases carry "outside_north_star_tolerance" in their test ids and are held to a regression bound of
    1e-2, which is NOT a parity claim.
"""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
import vh_synth as S

pytestmark = pytest.mark.gpu

vithip = pytest.importorskip("vithip")
NORTH_STAR = 1e-3
TOL = {vithip.DTYPE_FP16: NORTH_STAR, vithip.DTYPE_BF16: 1e-2}   # bf16: regression bound only, see the module docstring
NAME = {vithip.DTYPE_FP16: "fp16", vithip.DTYPE_BF16: "bf16"}
DT_PARAMS = [pytest.param(vithip.DTYPE_FP16, id="fp16-north_star_1e-3"),
             pytest.param(vithip.DTYPE_BF16, id="bf16-outside_north_star_tolerance-bound_1e-2")]
HERE = os.path.dirname(os.path.abspath(__file__))


def rel(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("dt", DT_PARAMS)
@pytest.mark.parametrize("name,batch", [("vit_micro", 5), ("vit_mini", 3), ("vit_tiny", 3), ("vit_base", 2), ("vit_gray", 7)])
def test_logits_match_oracle(dt, name, batch):
    cfg = S.CONFIGS[name]
    blob = S.make_blob(cfg, seed=0)
    images = S.make_images(cfg, seed=1, batch=batch)
    ref, ref_hidden = O.vit_forward(cfg, blob, images, want_hidden=True)
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=batch)
    ctx.load_weights(blob)
    got = ctx.forward(images)
    hid = ctx.debug_read(0, ref_hidden.size).reshape(ref_hidden.shape)
    e_h, e_l = rel(hid, ref_hidden), rel(got, ref)
    top1 = float((got.argmax(1) == ref.argmax(1)).mean())
    print(f"\n[parity] {name} b{batch} {NAME[dt]}: logits {e_l:.3e} hidden {e_h:.3e} top1-agree {top1:.2f}")
    assert np.isfinite(got).all()
    assert e_l <= TOL[dt], e_l
    ctx.close()


def test_vit_large_384_long_sequence_config():
    # BASELINE.json config 4 as a parity case: ViT-L/16 at 384x384 (T = 577 tokens, 24 layers, fp16).
    # Two images keep the CPU oracle at ~0.8 TFLOP (the same two the full-size test of config 4 checks, so the oracle
    # result is computed once per session); exercises the 152 KiB LDS-resident K/V path.
    cfg = S.CONFIGS["vit_large_384"]
    blob = S.make_blob(cfg, seed=0)
    images = S.make_images(cfg, seed=1, batch=2)
    ref = O.vit_forward(cfg, blob, images)
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2)
    ctx.load_weights(blob)
    got = ctx.forward(images)
    e = rel(got, ref)
    print(f"\n[parity] vit_large_384 b2 fp16: logits {e:.3e}")
    assert np.isfinite(got).all() and e <= NORTH_STAR, e
    ctx.close()


def test_vit_large_384_fp16_on_16_images_median_inside_worst_at_the_edge_of_the_north_star_tolerance():
    """BASELINE config 4 on a sample that shows its margin: ViT-L/16 at 384x384, fp16, 16 images, default flags (the guarded
    fold, the two-plane residual with the one-byte lo plane).  What 24 layers of 16-bit operands leave, measured in round 4:
    median 7.8-8.0e-4; the WORST of 16 images 9.2e-4 on one build and 1.02e-3 on the next (the GELU polynomial changed, to a
    MORE accurate one: which image is worst, and by how much, is decided by rounding noise); 64 images: 1.03-1.07e-3 on either
    LayerNorm path, and the fp32-residual build measures the same as the planes (profiles/r04_a_*, r04_b_*).  So the honest
    claim for config 4 is NOT "every image inside 1e-3", and this test does not pretend it: it holds the median inside 9e-4,
    at least 14 of the 16 images inside 1e-3, and the worst inside 1.15e-3 -- a regression bound on the tail, labelled as such
    (DESIGN.md section 5).  ViT-B/16, the headline model, is asserted at 1e-3 on 64 images below."""
    cfg = S.CONFIGS["vit_large_384"]
    n = 16
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, n)
    ref = O.vit_forward(cfg, blob, images)
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=n)
    ctx.load_weights(blob)
    assert ctx.ln_fold()
    got = ctx.forward(images)
    ctx.close()
    per = np.abs(got - ref).max(1) / np.abs(ref).max()
    inside = int((per <= NORTH_STAR).sum())
    print(f"\n[parity] vit_large_384 fp16 {n} images, default flags: worst {per.max():.3e} median {np.median(per):.3e}; "
          f"{inside} of {n} images inside 1e-3")
    assert np.isfinite(got).all()
    assert np.median(per) <= 9.0e-4, np.median(per)
    assert inside >= 14, (inside, per.max())
    assert per.max() <= 1.15e-3, per.max()             # tail regression bound: NOT the north star's tolerance
    assert (got.argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("flags,label", [(vithip.FLAG_LN_FOLD_ON, "folded"), (vithip.FLAG_LN_FOLD_OFF, "stand-alone")])
def test_fp16_is_inside_the_north_star_tolerance_on_64_vit_b_images(flags, label):
    """The configuration that carries the parity claim: ViT-B/16, fp16 operands, 64 images, BOTH LayerNorm paths; the worst
    image must be inside 1e-3 (not the mean, not a sample of two).  (Round 2 asserted 32 here and kept 64 in a tool.)"""
    cfg = S.CONFIGS["vit_base"]
    n = 64
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, n)
    ref = O.vit_forward(cfg, blob, images)
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=n, flags=flags)
    ctx.load_weights(blob)
    assert ctx.ln_fold() == (flags == vithip.FLAG_LN_FOLD_ON)
    got = ctx.forward(images)
    ctx.close()
    per = np.abs(got - ref).max(1) / np.abs(ref).max()
    print(f"\n[parity] vit_base fp16 {n} images, LayerNorm {label}: worst {per.max():.3e} median {np.median(per):.3e}")
    assert per.max() <= NORTH_STAR, per.max()


@pytest.mark.parametrize("name,batch,dtype_name", [("vit_base", 3, "bf16"), ("vit_base", 70, "fp16"), ("vit_large_384", 1, "fp16")])
def test_patch_embedding_with_the_gather_inside_the_gemm_gives_the_same_bits(monkeypatch, name, batch, dtype_name):
    """VH_PATCH_FUSED=1: the patch GEMM reads the NHWC fp32 images itself (kernels_patch.hip: register-path A loader, no patch
    matrix in memory) instead of im2col + GEMM.  Same conversion of every pixel, same k order in every MFMA, same epilogue:
    the logits must be bit-identical -- also where the last row tile is ragged (batch 3 and 70: 588 / 13 720 GEMM rows)."""
    cfg = S.CONFIGS[name]
    dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16}[dtype_name]
    images = S.make_images(cfg, 5, batch)
    outs = []
    for fused in ("0", "1"):
        monkeypatch.setenv("VH_PATCH_FUSED", fused)
        ctx = vithip.VitContext(cfg, dtype=dt, max_batch=batch)
        ctx.init_weights_seeded(0)
        assert ctx.ln_fold()
        outs.append(ctx.forward(images))
        ctx.close()
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("name,batch,dtype_name", [("vit_base", 256, "bf16"), ("vit_base", 300, "fp16"), ("vit_large_384", 64, "fp16")])
def test_tiled_hidden_activation_gives_the_same_bits_as_the_row_major_one(monkeypatch, name, batch, dtype_name):
    """Round 4: where both MLP GEMMs take the persistent form, fc1 writes the hidden activation h in a 16-row-blocked layout
    straight from its registers (no LDS transposition) and fc2's operand DMA reads h and a tiled copy of its weights in that
    layout; the attention kernel writes its output in the same layout for the out-projection (VH_ATT_TILED), and reads q|k|v
    HEAD-MAJOR ([3][heads][rows][64]) as the projection's epilogue then writes it (VH_QKV_HM).  Natural column order
    inside every chunk, so every MFMA sees the same k order: the logits must equal the row-major path's bit for bit (VH_H_TILED=0)
    -- on exact tiles (batch 256), on rows padded to whole tiles (300) and on ViT-L (T = 577: the ring without Q staging)."""
    cfg = S.CONFIGS[name]
    dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16}[dtype_name]
    px = cfg["image_size"] ** 2 * cfg["channels"]
    din, dout = vithip.DeviceBuffer(batch * px * 4), vithip.DeviceBuffer(batch * cfg["classes"] * 4)
    outs, used, used_hm = [], [], []
    # row-major; h tiled only; h and the attention output tiled; q|k|v head-major as well (the default)
    for tiled, att, hm in (("0", "1", "1"), ("1", "0", "1"), ("1", "1", "0"), ("1", "1", "1")):
        monkeypatch.setenv("VH_H_TILED", tiled)
        monkeypatch.setenv("VH_ATT_TILED", att)
        monkeypatch.setenv("VH_QKV_HM", hm)
        ctx = vithip.VitContext(cfg, dtype=dt, max_batch=batch)
        ctx.init_weights_seeded(0)
        ctx.fill_input_seeded(1, batch, din.ptr)
        ctx.forward_device(din.ptr, batch, dout.ptr)
        outs.append(dout.to_numpy(np.float32, (batch, cfg["classes"])))
        used.append(int(ctx.debug_read(3, 1)[0]))
        used_hm.append(int(ctx.debug_read(4, 1)[0]))
        small = ctx.forward(din.to_numpy(np.float32, (batch, cfg["image_size"], cfg["image_size"], cfg["channels"]))[:2])   # a batch too small for the persistent form
        assert int(ctx.debug_read(3, 1)[0]) == 0 and np.array_equal(small, outs[-1][:2])
        ctx.close()
    assert used == [0, 1, 1, 1] and used_hm == [0, 0, 0, 1], (used, used_hm)
    assert np.isfinite(outs[0]).all() and all(np.array_equal(outs[0], o) for o in outs[1:])


def test_layernorm_fold_is_a_property_of_the_configuration_not_of_the_workspace_size():
    """The same image gives the same logit BITS from a context sized for 1 image and from one sized for 300 (the fold used
    to switch on with max_batch, so the numerics depended on how the workspace had been sized)."""
    cfg = S.CONFIGS["vit_base"]
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, 1)
    outs = []
    for mb in (1, 300):
        ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=mb)
        ctx.load_weights(blob)
        assert ctx.ln_fold()
        outs.append(ctx.forward(images))
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype_name", ["bf16", "fp16"])
def test_one_image_gives_the_same_bits_at_batch_1_128_300_and_512(dtype_name):
    """The GEMM form depends on the row count -- 128x128 tiles at batch 1, one 256x256 tile per workgroup at 128, the
    persistent form on rows padded to whole tiles at 300, on exact tiles at 512 -- and so does the attention grid.  Every
    form adds a row's products in the same order, so image 0 must come out bit for bit the same from all of them (what
    sharded == unsharded rests on)."""
    cfg = S.CONFIGS["vit_base"]
    dt = {"bf16": vithip.DTYPE_BF16, "fp16": vithip.DTYPE_FP16}[dtype_name]
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=512, flags=vithip.FLAG_LN_FOLD_ON)
    ctx.init_weights_seeded(0)
    px = cfg["image_size"] ** 2 * cfg["channels"]
    din, dout = vithip.DeviceBuffer(512 * px * 4), vithip.DeviceBuffer(512 * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, 512, din.ptr)
    rows = {}
    for b in (1, 128, 300, 512):
        ctx.forward_device(din.ptr, b, dout.ptr)
        rows[b] = dout.to_numpy(np.float32, (b, cfg["classes"]))
    ctx.close()
    assert np.isfinite(rows[512]).all()
    for b in (1, 128, 300):
        assert np.array_equal(rows[b], rows[512][:b]), b


def test_folded_layernorm_with_a_large_common_mode_row_mean():
    """The folded path multiplies the RAW rounded residual, so its operand rounding error is relative to |x| and not to
    |x - mean|: it grows like sqrt(1 + (mean/sigma)^2) with a common-mode offset of the rows.  Drive the residual stream
    with such an offset (position embedding + 1.5 in every channel, |mean| / sigma ~ 5) and check (a) the stand-alone path
    (VH_FLAG_LN_FOLD_OFF), which normalises in fp32 before rounding, stays inside the north star's tolerance, (b) the folded
    path degrades by no more than the predicted factor.  This is the documented reason for the flag (include/vithip.h)."""
    cfg = S.CONFIGS["vit_base"]
    t = S.make_tensors(cfg, 0)
    t["pos"] = t["pos"] + np.float32(1.5)
    blob = S.pack_blob(cfg, t)
    images = S.make_images(cfg, 1, 4)
    ref, hid = O.vit_forward(cfg, blob, images, n_layers=0, want_hidden=True)
    ratio = float(np.abs(hid.mean(1)).mean() / hid.std(1).mean())
    ref = O.vit_forward(cfg, blob, images)
    err = {}
    for flags in (vithip.FLAG_LN_FOLD_OFF, vithip.FLAG_LN_FOLD_ON):
        ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=4, flags=flags)
        ctx.load_weights(blob)
        err[flags] = rel(ctx.forward(images), ref)
        ctx.close()
    print(f"\n[fold] |row mean| / sigma at the input of layer 0 = {ratio:.1f}: stand-alone {err[vithip.FLAG_LN_FOLD_OFF]:.3e}, "
          f"folded {err[vithip.FLAG_LN_FOLD_ON]:.3e}")
    assert ratio > 3.0
    assert err[vithip.FLAG_LN_FOLD_OFF] <= NORTH_STAR
    assert err[vithip.FLAG_LN_FOLD_ON] <= 1.5 * NORTH_STAR * np.sqrt(1.0 + ratio * ratio)
    # DEFAULT flags = the guarded fold.  The offset is a property of the WEIGHTS, so the calibration forward at the end of
    # load_weights already measures it and leaves the context on the stand-alone LayerNorm: the FIRST forward of every
    # entry point -- the asynchronous device-pointer one included (round 3: it returned the first batch from the tripped
    # fold) -- is inside the tolerance.
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=4)
    assert ctx.ln_fold()
    ctx.load_weights(blob)
    seen, thresh, tripped = ctx.ln_guard()
    print(f"[fold] default flags: calibration at load saw {seen:.2f} (threshold {thresh:.2f}), tripped {tripped}, folded now {ctx.ln_fold()}")
    assert tripped and seen > 3.0 and not ctx.ln_fold()
    got = ctx.forward(images)
    assert rel(got, ref) <= NORTH_STAR
    d_in, d_out = vithip.DeviceBuffer.from_numpy(images), vithip.DeviceBuffer(4 * cfg["classes"] * 4)
    ctx2 = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=4)
    ctx2.load_weights(blob)
    assert not ctx2.ln_fold()
    ctx2.forward_device_async(d_in.ptr, 4, d_out.ptr, steps=1)     # the FIRST asynchronous call
    ctx2.synchronize()
    got2 = d_out.to_numpy(np.float32, (4, cfg["classes"]))
    print(f"[fold] first asynchronous forward: {rel(got2, ref):.3e}")
    assert rel(got2, ref) <= NORTH_STAR
    assert np.array_equal(got2, got)           # the stand-alone path, same bits as the synchronous forward
    # new weights re-arm the guard and restore the configured path (the calibration of the new weights stays below the threshold)
    ctx2.load_weights(S.make_blob(cfg, 0))
    seen2, _, tripped2 = ctx2.ln_guard()
    assert ctx2.ln_fold() and not tripped2 and 0.0 < seen2 < thresh
    ctx2.close()
    ctx.close()


def test_the_per_forward_guard_is_the_backstop_for_what_only_the_data_can_cause():
    """A trip the calibration image cannot see: every patch-kernel entry carries a small common component, so an image with
    a DC level gets a common-mode offset in every channel, while the zero-mean calibration image does not.  The load keeps
    the fold; the first batch of DC images trips the guard; vh_synchronize (or the next forward entry) switches the context."""
    cfg = S.CONFIGS["vit_base"]
    t = S.make_tensors(cfg, 0)
    t["patch.weight"] = t["patch.weight"] + np.float32(0.0015)
    blob = S.pack_blob(cfg, t)
    images = (S.make_images(cfg, 1, 2) * np.float32(0.1) + np.float32(0.9)).astype(np.float32)
    ref = O.vit_forward(cfg, blob, images)
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2)
    ctx.load_weights(blob)
    seen0, thresh, tripped0 = ctx.ln_guard()
    assert ctx.ln_fold() and not tripped0, (seen0, thresh)          # the calibration image does not show it
    d_in, d_out = vithip.DeviceBuffer.from_numpy(images), vithip.DeviceBuffer(2 * cfg["classes"] * 4)
    ctx.forward_device_async(d_in.ptr, 2, d_out.ptr, steps=1)
    ctx.synchronize()                                                # polls the guard: the switch happens here
    seen, _, tripped = ctx.ln_guard()
    print(f"\n[fold] data-induced offset: calibration {seen0:.2f}, batch {seen:.2f} (threshold {thresh:.2f}), tripped {tripped}")
    assert tripped and seen > thresh and not ctx.ln_fold()
    ctx.forward_device_async(d_in.ptr, 2, d_out.ptr, steps=1)
    ctx.synchronize()
    assert rel(d_out.to_numpy(np.float32, (2, cfg["classes"])), ref) <= NORTH_STAR
    assert rel(ctx.forward(images), ref) <= NORTH_STAR
    ctx.close()


def test_new_weights_after_a_tripped_guard_do_not_replay_the_old_paths_graphs():
    """Graph replay on: a context that tripped holds captured launch sequences of the stand-alone-LayerNorm path.  A weight
    load restores the folded path and re-folds the weights, so those graphs must go (replaying them against gamma-folded
    weights would be silently wrong): logits after the reload equal an eager context's bit for bit."""
    cfg = S.CONFIGS["vit_base"]
    t = S.make_tensors(cfg, 0)
    t["pos"] = t["pos"] + np.float32(1.5)
    bad, good = S.pack_blob(cfg, t), S.make_blob(cfg, 0)
    images = S.make_images(cfg, 1, 2)
    d_in, d_out = vithip.DeviceBuffer.from_numpy(images), vithip.DeviceBuffer(2 * cfg["classes"] * 4)
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2)
    ctx.set_graph(True)
    ctx.load_weights(bad)
    assert not ctx.ln_fold()
    for _ in range(3):                                               # eager once, then captured and replayed
        ctx.forward_device(d_in.ptr, 2, d_out.ptr)
    assert ctx.get_graph()[1] >= 1
    ctx.load_weights(good)
    assert ctx.ln_fold() and ctx.get_graph()[1] == 0
    for _ in range(3):
        ctx.forward_device(d_in.ptr, 2, d_out.ptr)
    got = d_out.to_numpy(np.float32, (2, cfg["classes"]))
    ctx.close()
    eager = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2)
    eager.load_weights(good)
    want = eager.forward(images)
    eager.close()
    assert np.array_equal(got, want)


def test_default_guarded_fold_stays_folded_on_the_synthetic_nets_and_on_outlier_channels():
    """What trained ViTs show is not a common-mode offset but a few OUTLIER CHANNELS: 6 channels of the position embedding,
    the class token and the patch bias scaled by 50.  That leaves |row mean| / sigma small (the guard must not trip) and
    costs the folded operand nothing -- 16-bit rounding is relative per element -- so both LayerNorm paths stay inside the
    north star's 1e-3 (fp16)."""
    cfg = S.CONFIGS["vit_base"]
    t = S.make_tensors(cfg, 0)
    ch = np.array([5, 77, 190, 333, 501, 700])
    for name in ("pos", "cls", "patch.bias"):
        t[name] = t[name].copy()
        t[name][..., ch] *= np.float32(50.0)
    blob = S.pack_blob(cfg, t)
    images = S.make_images(cfg, 1, 4)
    ref = O.vit_forward(cfg, blob, images)
    for flags, name in ((0, "guarded fold (default)"), (vithip.FLAG_LN_FOLD_OFF, "stand-alone")):
        ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=4, flags=flags)
        ctx.load_weights(blob)
        e = rel(ctx.forward(images), ref)
        seen, thresh, tripped = ctx.ln_guard()
        print(f"\n[outliers] {name}: {e:.3e}; guard saw {seen:.3f} of {thresh:.2f}")
        assert e <= NORTH_STAR, (name, e)
        if flags == 0:
            assert ctx.ln_fold() and not tripped and 0.0 < seen < thresh
        ctx.close()


@pytest.mark.parametrize("dt", DT_PARAMS)
def test_logits_match_golden_fixtures(dt):
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))):
        g = np.load(path)
        name = os.path.basename(path).split("_s")[0]
        cfg = S.CONFIGS[name]
        wseed, iseed, batch = [int(v) for v in g["meta"]]
        ctx = vithip.VitContext(cfg, dtype=dt, max_batch=batch)
        ctx.load_weights(S.make_blob(cfg, wseed))
        got = ctx.forward(S.make_images(cfg, iseed, batch))
        e = rel(got, g["logits_f64"])
        print(f"\n[golden] {name} {NAME[dt]}: {e:.3e}")
        assert e <= TOL[dt], (name, e)
        ctx.close()


def test_folded_layernorm_path_agrees_with_the_separate_layernorm_path():
    # ViT-B (dim % 256 == 0) runs with LayerNorm folded into the q|k|v and fc1 GEMMs; VH_FLAG_LN_FOLD_OFF selects the
    # stand-alone LayerNorm kernel.  Both must sit within tolerance of the oracle and close to each other.
    cfg = S.CONFIGS["vit_base"]
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, 2)
    ref = O.vit_forward(cfg, blob, images)
    outs = {}
    for fold in (vithip.FLAG_LN_FOLD_ON, vithip.FLAG_LN_FOLD_OFF):
        ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2, flags=fold)
        ctx.load_weights(blob)
        outs[fold] = ctx.forward(images)
        ctx.set_streams(2)                                   # concurrent parts slice the statistics buffers
        assert ctx.get_streams() == 2 and np.array_equal(ctx.forward(images), outs[fold]), fold
        ctx.close()
        assert rel(outs[fold], ref) <= NORTH_STAR, (fold, rel(outs[fold], ref))
    a, b = outs[vithip.FLAG_LN_FOLD_ON], outs[vithip.FLAG_LN_FOLD_OFF]
    print(f"\n[fold] fp16 folded vs separate LN: {rel(a, b):.3e}")
    assert not np.array_equal(a, b)   # they really are different code paths
    assert rel(a, b) <= 2 * NORTH_STAR   # each is within the tolerance of the oracle (asserted above): of each other, twice that


def test_layer_by_layer_against_oracle():
    # run 0, 1, 2 ... layers and compare the residual stream: localises a wrong stage
    cfg = S.CONFIGS["vit_mini"]
    blob, images = S.make_blob(cfg, 3), S.make_images(cfg, 4, 2)
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2)
    ctx.load_weights(blob)
    for nl in range(cfg["layers"] + 1):
        _, ref_h = O.vit_forward(cfg, blob, images, n_layers=nl, want_hidden=True)
        ctx.debug_set_layers(nl)
        ctx.forward(images)
        hid = ctx.debug_read(0, ref_h.size).reshape(ref_h.shape)
        assert rel(hid, ref_h) <= 1e-3, (nl, rel(hid, ref_h))
    ctx.close()


def test_deterministic_and_batch_independent():
    # image i's logits do not depend on what else is in the batch nor on the batch size: this is
    # what makes image-sharding across GPUs exact (sharded == unsharded, bitwise)
    cfg = S.CONFIGS["vit_tiny"]
    blob = S.make_blob(cfg, 0)
    images = S.make_images(cfg, 1, 6)
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=6)
    ctx.load_weights(blob)
    full = ctx.forward(images)
    again = ctx.forward(images)
    assert np.array_equal(full, again)
    for lo, hi in ((0, 3), (3, 6), (2, 3), (5, 6)):
        part = ctx.forward(images[lo:hi])
        assert np.array_equal(part, full[lo:hi]), (lo, hi)
    ctx.close()


def test_full_size_batch_512_properties():
    # BASELINE config 2 at its full size (ViT-B/16, bf16, 512 images = 100 864 token rows).  The oracle cannot run
    # 512 ViT-B images in test time, so the full-size run is checked through size-independent properties:
    #   * images are independent: any sub-batch run alone reproduces its rows of the big run bit for bit
    #     (this is also what makes the 8-GPU image sharding exact),
    #   * a permutation of the images permutes the logits,
    #   * and eight of its rows are compared with the oracle directly.
    cfg = S.CONFIGS["vit_base"]
    B = 512
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=B)
    ctx.init_weights_seeded(0)
    blob = ctx.export_weights()
    din = vithip.DeviceBuffer(B * 224 * 224 * 3 * 4)
    dout = vithip.DeviceBuffer(B * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, B, din.ptr)
    ctx.forward_device(din.ptr, B, dout.ptr)
    full = dout.to_numpy(np.float32, (B, cfg["classes"]))
    assert np.isfinite(full).all()
    images = din.to_numpy(np.float32, (B, 224, 224, 3))
    assert np.array_equal(images, S.make_images(cfg, 1, B))          # device generator == numpy generator
    for lo, hi in ((0, 8), (250, 263), (511, 512)):
        assert np.array_equal(ctx.forward(images[lo:hi]), full[lo:hi]), (lo, hi)
    perm = np.array([300, 7, 511, 0, 128, 64])
    assert np.array_equal(ctx.forward(images[perm]), full[perm])
    ref = O.vit_forward(cfg, blob, images[:8])
    e = rel(full[:8], ref)
    print(f"\n[full size] vit_base b512 bf16: rows 0-7 vs oracle {e:.3e}")
    assert e <= TOL[vithip.DTYPE_BF16]
    ctx.close()


def test_full_size_config_4_vit_large_384_fp16_batch_256_properties():
    # BASELINE.json config 4 at its FULL size: ViT-L/16 at 384x384 (577 tokens, 24 layers), fp16, 256 images = 147 712
    # token rows, 453 MB of input.  Size-independent properties as for config 2: sub-batches run alone (from both ends and
    # the middle) reproduce their rows bit for bit, a permutation of images permutes the logits, the run is
    # deterministic -- plus two rows against the oracle AT THE NORTH STAR'S TOLERANCE (fp16).
    import ctypes as C
    cfg = S.CONFIGS["vit_large_384"]
    B, per = 256, 384 * 384 * 3
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=B)
    ctx.init_weights_seeded(0)
    blob = ctx.export_weights()
    din = vithip.DeviceBuffer(B * per * 4)
    dout = vithip.DeviceBuffer(B * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, B, din.ptr)
    ctx.forward_device(din.ptr, B, dout.ptr)
    full = dout.to_numpy(np.float32, (B, cfg["classes"]))
    assert np.isfinite(full).all()
    ctx.forward_device(din.ptr, B, dout.ptr)
    assert np.array_equal(dout.to_numpy(np.float32, (B, cfg["classes"])), full)      # deterministic

    def rows(lo, hi):
        imgs = np.empty((hi - lo, 384, 384, 3), np.float32)
        assert vithip.lib().vh_memcpy_d2h(0, imgs.ctypes.data, C.c_void_p(din.ptr + lo * per * 4), imgs.nbytes) == 0
        return imgs

    for lo, hi in ((0, 3), (127, 130), (255, 256)):
        assert np.array_equal(ctx.forward(rows(lo, hi)), full[lo:hi]), (lo, hi)
    perm = [200, 3, 255, 0]
    assert np.array_equal(ctx.forward(np.concatenate([rows(i, i + 1) for i in perm])), full[perm])
    first = rows(0, 2)
    assert np.array_equal(first, S.make_images(cfg, 1, 2))                            # device generator == numpy generator
    ref = O.vit_forward(cfg, blob, first)
    e = rel(full[:2], ref)
    print(f"\n[full size] vit_large_384 b256 fp16: rows 0-1 vs oracle {e:.3e}")
    assert e <= NORTH_STAR, e
    ctx.close()


@pytest.mark.parametrize("B", [100, 300])
@pytest.mark.parametrize("dtype_name", ["fp16", "bf16", "fp8"])
def test_a_batch_that_is_no_multiple_of_256_rows_agrees_bitwise_with_its_sub_batches(dtype_name, B):
    # Token-row counts that are not multiples of the 256-row GEMM tile.  100 images (19 700 rows): few tiles, the forward
    # takes the ping-pong GEMM's one-tile-per-workgroup form with a ragged last row of tiles.  300 images (59 100 rows):
    # the folded layer loop runs its GEMMs on 59 136 rows -- whole tiles, the persistent form -- and the 36 padding rows
    # hold whatever the arena holds; nothing of them may reach a logit.  Images are independent: sub-batches from both
    # ends and the middle, which run through the small-shape kernels, must reproduce their rows of the big run bit for
    # bit -- for every operand type, and again on a second run (the padding rows have changed by then).
    dt = {"fp16": vithip.DTYPE_FP16, "bf16": vithip.DTYPE_BF16, "fp8": vithip.DTYPE_FP8}[dtype_name]
    cfg = S.CONFIGS["vit_base"]
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=B)
    ctx.init_weights_seeded(0)
    images = S.make_images(cfg, 1, B)
    full = ctx.forward(images)
    assert np.isfinite(full).all()
    for lo, hi in ((0, 3), (B // 2, B // 2 + 3), (B - 3, B)):
        assert np.array_equal(ctx.forward(images[lo:hi]), full[lo:hi]), (dtype_name, lo, hi)
    assert np.array_equal(ctx.forward(images), full)
    ctx.set_streams(3)     # three concurrent parts: only the last one may pad (the others have a neighbour behind their rows)
    assert np.array_equal(ctx.forward(images), full)
    ctx.close()


def test_batch_4096_crosses_the_2_to_31_element_mark():
    # BASELINE config 3's global batch on ONE device: 806 912 token rows; the MLP hidden tensor has 2.48e9 elements,
    # so every row * width product in the kernels must be 64-bit.  Checked through batch independence: images from
    # the far end of the batch, run alone, must reproduce their rows of the big run bit for bit.
    import ctypes as C
    cfg = S.CONFIGS["vit_base"]
    B, per = 4096, 224 * 224 * 3
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=B)
    ctx.init_weights_seeded(0)
    din = vithip.DeviceBuffer(B * per * 4)
    dout = vithip.DeviceBuffer(B * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, B, din.ptr)
    ctx.forward_device(din.ptr, B, dout.ptr)
    full = dout.to_numpy(np.float32, (B, cfg["classes"]))
    assert np.isfinite(full).all()
    for lo, hi in ((0, 2), (2047, 2050), (4093, 4096)):
        imgs = np.empty((hi - lo, 224, 224, 3), np.float32)
        rc = vithip.lib().vh_memcpy_d2h(0, imgs.ctypes.data, C.c_void_p(din.ptr + lo * per * 4), imgs.nbytes)
        assert rc == 0
        assert np.array_equal(ctx.forward(imgs), full[lo:hi]), (lo, hi)
    ctx.close()


def test_concurrent_parts_give_bit_identical_logits():
    # vh_set_streams(n): the batch runs as n contiguous parts on n streams; images are independent, so every n
    # (and uneven splits: 5 images in 2, 3, 4 parts) must reproduce the single-stream logits bit for bit
    cfg = S.CONFIGS["vit_tiny"]
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=5)
    ctx.load_weights(S.make_blob(cfg, 0))
    images = S.make_images(cfg, 1, 5)
    ctx.set_streams(1)
    ref = ctx.forward(images)
    for n in (2, 3, 4):
        ctx.set_streams(n)
        assert ctx.get_streams() == n
        assert np.array_equal(ctx.forward(images), ref), n
        assert np.array_equal(ctx.forward(images[:1]), ref[:1]), n   # fewer images than parts
    with pytest.raises(vithip.VhError):
        ctx.set_streams(9)
    ctx.close()


def test_ring_pipelined_submit_collect_is_fifo_and_equals_forward():
    # the in-flight ring (reference pattern: filter_image / get_filtered_image, netFPGA.cpp:292-365): results come
    # back in submission order, bit-identical to the synchronous forward, for ragged batch sizes, with both ways
    # of handing over the input (caller buffer / the slot's pinned buffer filled in place)
    cfg = S.CONFIGS["vit_tiny"]
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=4)
    ctx.load_weights(S.make_blob(cfg, 2))
    sizes = [4, 1, 3, 4, 2, 4, 1]
    batches = [S.make_images(cfg, 10 + i, n) for i, n in enumerate(sizes)]
    refs = [ctx.forward(b) for b in batches]
    with pytest.raises(vithip.VhError):
        ctx.ring_submit(batches[0])          # no ring yet
    ctx.ring_create(3, 4)
    assert ctx.ring_free_slots() == 3
    with pytest.raises(vithip.VhError) as e:
        ctx.ring_collect()                   # "PILA VACIA"
    assert e.value.code == 7
    got = []
    for i, b in enumerate(batches):
        if ctx.ring_free_slots() == 0:
            with pytest.raises(vithip.VhError) as e:
                ctx.ring_submit(b)           # "PILA LLENA"
            assert e.value.code == 6
            got.append(ctx.ring_collect())
        if i % 2:
            ctx.ring_input(len(b))[...] = b  # fill the pinned slot in place
            ctx.ring_submit(None, len(b))
        else:
            ctx.ring_submit(b)
    while ctx.ring_free_slots() < 3:
        got.append(ctx.ring_collect())
    assert len(got) == len(refs)
    for g, r in zip(got, refs):
        assert g.shape == r.shape and np.array_equal(g, r)
    with pytest.raises(vithip.VhError):
        ctx.ring_submit(S.make_images(cfg, 1, 5))   # larger than a slot
    # the synchronous path still works with a ring present, and with concurrent parts under the ring
    ctx.set_streams(2)
    ctx.ring_submit(batches[0])
    ctx.ring_submit(batches[2])
    assert np.array_equal(ctx.ring_collect(), refs[0])
    assert np.array_equal(ctx.forward(batches[1]), refs[1])
    assert np.array_equal(ctx.ring_collect(), refs[2])
    ctx.close()


def test_tail_overlap_split_launch_is_bit_identical(monkeypatch):
    # VH_TAIL_OVERLAP=1: residual GEMMs are launched as [full rounds of tiles] + [tail round] and the LayerNorm of the
    # finished rows runs beside the tail round on a helper stream.  Needs more tiles than CUs, i.e. a large batch.
    cfg = S.CONFIGS["vit_mini"]                     # 37 tokens, dim 192: tiles_n = 1
    B = 2200                                        # 81 400 rows = 318 M-tiles > 256 CUs, 62 in the tail round
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("VH_TAIL_OVERLAP", flag)
        ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=B)
        ctx.init_weights_seeded(4)
        din = vithip.DeviceBuffer(B * 96 * 96 * 3 * 4)
        dout = vithip.DeviceBuffer(B * cfg["classes"] * 4)
        ctx.fill_input_seeded(5, B, din.ptr)
        ctx.forward_device(din.ptr, B, dout.ptr)
        outs[flag] = dout.to_numpy(np.float32, (B, cfg["classes"]))
        # the option must really split: 2 residual GEMMs per layer, the last fc2 has no LayerNorm behind it
        splits = int(ctx.debug_read(2, 1)[0])
        assert splits == (2 * cfg["layers"] - 1 if flag == "1" else 0), (flag, splits)
        ctx.close()
    assert np.isfinite(outs["0"]).all() and np.array_equal(outs["0"], outs["1"])


def test_graph_replay_is_bit_identical_to_eager_launches():
    # vh_set_graph: the launch sequence is captured once per (input, output, batch) and replayed; the first forward
    # at a batch size runs eagerly, the second is captured, later ones replay.  Same bits in every case, for the
    # host path, the device path, concurrent parts and the ring.
    cfg = S.CONFIGS["vit_tiny"]
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=4)
    ctx.load_weights(S.make_blob(cfg, 1))
    imgs = {b: S.make_images(cfg, 20 + b, b) for b in (1, 3, 4)}
    want = {b: ctx.forward(imgs[b]) for b in imgs}
    ctx.set_graph(True)
    assert ctx.get_graph() == (True, 0)
    for rep in range(4):
        for b in (1, 3, 4, 1):
            assert np.array_equal(ctx.forward(imgs[b]), want[b]), (rep, b)
    on, cached = ctx.get_graph()
    assert on and cached == 3                                   # one graph per batch size on the host path
    other = S.make_images(cfg, 99, 3)
    assert np.array_equal(ctx.forward(other), vithip_eager(cfg, other))   # new data through the same graph
    din, dout = vithip.DeviceBuffer.from_numpy(imgs[4]), vithip.DeviceBuffer(4 * cfg["classes"] * 4)
    for _ in range(3):
        ctx.forward_device(din.ptr, 4, dout.ptr)
        assert np.array_equal(dout.to_numpy(np.float32, want[4].shape), want[4])
    ctx.set_streams(2)                                          # drops the graphs; parts fork/join inside the capture
    for _ in range(3):
        assert np.array_equal(ctx.forward(imgs[3]), want[3])
    ctx.ring_create(2, 4)
    for _ in range(3):
        ctx.ring_submit(imgs[4]); ctx.ring_submit(imgs[1])
        assert np.array_equal(ctx.ring_collect(), want[4]) and np.array_equal(ctx.ring_collect(), want[1])
    ctx.set_graph(False)
    assert ctx.get_graph() == (False, 0) and np.array_equal(ctx.forward(imgs[3]), want[3])
    ctx.close()


def vithip_eager(cfg, images):
    c = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=len(images))
    c.load_weights(S.make_blob(cfg, 1))
    out = c.forward(images)
    c.close()
    return out


def test_contexts_can_be_created_and_destroyed_repeatedly():
    # no leaked device memory / streams / events across create-use-destroy cycles (incl. ring, graphs, filter ring)
    cfg = S.CONFIGS["vit_micro"]
    blob, images = S.make_blob(cfg, 8), S.make_images(cfg, 9, 2)
    first = None
    free0 = None
    for i in range(12):
        ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16 if i % 2 else vithip.DTYPE_BF16, max_batch=2)
        ctx.load_weights(blob)
        ctx.set_graph(i % 3 == 0)
        ctx.ring_create(2, 2)
        ctx.ring_submit(images)
        got = ctx.ring_collect()
        got2 = ctx.forward(images)
        assert np.array_equal(got, got2)
        if i == 1:
            first = got
        if i % 2 and i > 1:
            assert np.array_equal(got, first)
        f = vithip.FilterPipeline(32, 48, slots=3)
        f.submit(np.zeros((32, 48), np.uint8)); f.collect(); f.close()
        ctx.close()
        free = vithip.device_free_bytes()
        if i == 2:
            free0 = free
        if i > 2:
            assert free >= free0 - (64 << 20), (i, free0, free)   # allow allocator slack, catch per-cycle leaks
    assert first is not None


def test_plain_c_example_runs_and_agrees_with_the_python_binding(tmp_path):
    import re
    import subprocess
    root = os.path.dirname(HERE)
    lib_dir = os.path.dirname(vithip.LIB_PATH)
    exe = tmp_path / "classify"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "classify.c"),
                           "-L", lib_dir, "-lvithip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)])
    cfg = S.CONFIGS["vit_base"]
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=2)
    ctx.init_weights_seeded(0)
    want = ctx.forward(S.make_images(cfg, 1, 2))
    path = tmp_path / "b16.vhblob"
    ctx.save_weights_file(path)
    ctx.close()
    for arg in ("-", str(path)):                                  # seeded weights, then the same weights from a file
        out = subprocess.check_output([str(exe), arg, "2"], text=True)
        got = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"class (\d+) \(logit ([-0-9.]+)\)", out)]
        assert [g[0] for g in got] == list(want.argmax(1)), out
        assert np.allclose([g[1] for g in got], want.max(1), atol=1e-4)


def test_two_contexts_on_two_host_threads():
    # per-instance device state (no globals, unlike the reference's namespace-level singletons, netFPGA.cpp:21-56):
    # distinct contexts are usable from distinct threads at the same time and do not disturb each other
    import threading
    cfgs = [S.CONFIGS["vit_tiny"], S.CONFIGS["vit_mini"]]
    dts = [vithip.DTYPE_BF16, vithip.DTYPE_FP16]
    data = [(S.make_blob(c, 3 + i), S.make_images(c, 4 + i, 3)) for i, c in enumerate(cfgs)]
    want = []
    for (blob, imgs), c, dt in zip(data, cfgs, dts):
        ctx = vithip.VitContext(c, dtype=dt, max_batch=3)
        ctx.load_weights(blob)
        want.append(ctx.forward(imgs))
        ctx.close()
    errors = []

    def worker(i):
        try:
            ctx = vithip.VitContext(cfgs[i], dtype=dts[i], max_batch=3)
            ctx.load_weights(data[i][0])
            for _ in range(25):
                if not np.array_equal(ctx.forward(data[i][1]), want[i]):
                    errors.append((i, "logits differ"))
                    break
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in (0, 1)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_device_resident_path_equals_host_path():
    cfg = S.CONFIGS["vit_mini"]
    batch = 4
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=batch)
    ctx.load_weights(S.make_blob(cfg, 5))
    images = S.make_images(cfg, 6, batch)
    host = ctx.forward(images)
    din = vithip.DeviceBuffer.from_numpy(images)
    dout = vithip.DeviceBuffer(batch * cfg["classes"] * 4)
    ctx.forward_device(din.ptr, batch, dout.ptr)
    assert np.array_equal(dout.to_numpy(np.float32, host.shape), host)
    assert ctx.last_forward_us() > 0 and ctx.last_kernel_ms() > 0
    prof = ctx.profile_forward(din.ptr, batch, dout.ptr)
    assert prof["fc1_gemm"][1] == cfg["layers"] and prof["layernorm"][1] == 2 * cfg["layers"]
    ctx.close()


def test_seeded_weights_on_device_equal_the_generators():
    cfg = S.CONFIGS["vit_mini"]
    ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_BF16, max_batch=2)
    ctx.init_weights_seeded(77)
    blob = ctx.export_weights()
    assert np.array_equal(blob, S.make_blob(cfg, 77))
    assert np.array_equal(blob, O.make_blob(cfg, 77))
    # synthetic images generated in HBM == numpy generator
    din = vithip.DeviceBuffer(2 * cfg["image_size"] ** 2 * 3 * 4)
    ctx.fill_input_seeded(9, 2, din.ptr)
    img = din.to_numpy(np.float32, (2, cfg["image_size"], cfg["image_size"], 3))
    assert np.array_equal(img, S.make_images(cfg, 9, 2))
    # and a forward on them matches the oracle
    ref = O.vit_forward(cfg, blob, img)
    dout = vithip.DeviceBuffer(2 * cfg["classes"] * 4)
    ctx.forward_device(din.ptr, 2, dout.ptr)
    assert rel(dout.to_numpy(np.float32, ref.shape), ref) <= TOL[vithip.DTYPE_BF16]
    ctx.close()


def test_weight_roundtrip_through_device_blob():
    # the broadcast payload: export to a device buffer, load another context from it
    cfg = S.CONFIGS["vit_micro"]
    a = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2)
    a.load_weights(S.make_blob(cfg, 1))
    buf = vithip.DeviceBuffer(a.blob_bytes)
    a.export_weights_device(buf.ptr, a.blob_bytes)
    b = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=2)
    b.load_weights_device(buf.ptr, a.blob_bytes)
    images = S.make_images(cfg, 2, 2)
    assert np.array_equal(a.forward(images), b.forward(images))
    a.close(), b.close()


def test_error_reporting():
    cfg = S.CONFIGS["vit_micro"]
    ctx = vithip.VitContext(cfg, max_batch=2)
    images = S.make_images(cfg, 1, 2)
    with pytest.raises(vithip.VhError, match="before weights"):
        ctx.forward(images)
    with pytest.raises(vithip.VhError, match="bytes"):
        ctx.load_weights(np.zeros(100, dtype=np.uint8))
    bad = S.make_blob(cfg, 0).copy()
    bad[0] = ord("X")
    with pytest.raises(vithip.VhError, match="magic"):
        ctx.load_weights(bad)
    other = S.make_blob(S.CONFIGS["vit_mini"], 0)
    with pytest.raises(vithip.VhError):
        ctx.load_weights(other)
    ctx.load_weights(S.make_blob(cfg, 0))
    with pytest.raises(vithip.VhError, match="max_batch"):
        ctx.forward(S.make_images(cfg, 1, 3))
    bad_cfg = dict(cfg, dim=100)
    with pytest.raises(vithip.VhError):
        vithip.VitContext(bad_cfg)
    ctx.close()


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_mlp_mode_matches_oracle(act):
    # the reference's real launch_forward semantics: dense-layer chain over net_data's layout
    n_ins, npl = 37, [64, 130, 5]
    n_params = n_ins * 64 + 64 * 130 + 130 * 5
    params, bias = O.mlp_random_params(n_params, sum(npl), seed=3)
    params *= 0.2
    x = S.fill(n_ins, 1, 1, 0)
    ref = O.mlp_forward(n_ins, npl, params, bias, act, x)
    m = vithip.MlpContext(n_ins, npl, activation=act)
    m.load_params(params, bias)
    got = m.forward(x)[0]
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
    many = m.forward(np.stack([x, -x, 0.5 * x]))
    assert np.array_equal(many[0], got)
    m.close()


@pytest.mark.parametrize("dt", DT_PARAMS)
def test_class_token_tail_flag_gives_the_full_forward_logits_to_rounding(dt):
    """VH_FLAG_CLS_TAIL (opt-in): the last layer computes only what the logits need -- attention for the class-token query,
    out-proj / fc1 / fc2 on the `batch` class rows.  Its logits must be the full forward's to rounding (the one-query attention
    is another kernel, so not bitwise), inside the same tolerance against the oracle, deterministic and independent of what
    else is in the batch; a model the folded path does not apply to ignores the flag (bitwise the default)."""
    cfg = S.CONFIGS["vit_base"]
    blob = S.make_blob(cfg, seed=0)
    images = S.make_images(cfg, seed=1, batch=6)
    ref = O.vit_forward(cfg, blob, images)
    full = vithip.VitContext(cfg, dtype=dt, max_batch=6)
    full.load_weights(blob)
    want = full.forward(images)
    full.close()
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=6, flags=vithip.FLAG_CLS_TAIL)
    ctx.load_weights(blob)
    got = ctx.forward(images)
    assert np.isfinite(got).all()
    print(f"\n[cls tail] {NAME[dt]}: vs the full forward {rel(got, want):.3e}, vs the oracle {rel(got, ref):.3e} (full: {rel(want, ref):.3e})")
    assert rel(got, want) <= (2e-4 if dt == vithip.DTYPE_FP16 else 2e-3)
    assert rel(got, ref) <= TOL[dt]
    assert np.array_equal(got, ctx.forward(images))
    for lo, hi in ((0, 3), (4, 5), (5, 6)):
        assert np.array_equal(ctx.forward(images[lo:hi]), got[lo:hi]), (lo, hi)
    ctx.close()
    tiny = S.CONFIGS["vit_tiny"]                      # dim 192: no fold, so no split planes and no tail
    tb, ti = S.make_blob(tiny, 0), S.make_images(tiny, 1, 3)
    outs = []
    for flags in (0, vithip.FLAG_CLS_TAIL):
        c2 = vithip.VitContext(tiny, dtype=dt, max_batch=3, flags=flags)
        c2.load_weights(tb)
        outs.append(c2.forward(ti))
        c2.close()
    assert np.array_equal(outs[0], outs[1])


def test_massive_activations_do_not_saturate_the_one_byte_lo_plane():
    """Trained ViTs carry a few "massive activations": single channels of the residual stream at hundreds to thousands while
    the rest is O(1).  The split residual's lo byte holds (x - hi) * 256 (fp16) / * 32 (bf16) in e4m3, i.e. at most |x| / 8:
    it saturates only beyond |x| = 3 584.  Drive one channel to about 2 000 through the position embedding and check that the
    folded default path (planes) is as close to the fp32 oracle as the stand-alone path (fp32 residual) is."""
    cfg = S.CONFIGS["vit_base"]
    t = S.make_tensors(cfg, 0)
    t["pos"] = t["pos"].copy()
    t["pos"][..., 301] += np.float32(2000.0)
    blob = S.pack_blob(cfg, t)
    images = S.make_images(cfg, 1, 4)
    ref, hid = O.vit_forward(cfg, blob, images, want_hidden=True)
    assert np.abs(hid[..., 301]).max() > 1500.0           # the channel is still massive after the last layer
    err = {}
    for flags in (vithip.FLAG_LN_FOLD_OFF, 0):
        ctx = vithip.VitContext(cfg, dtype=vithip.DTYPE_FP16, max_batch=4, flags=flags)
        ctx.load_weights(blob)
        got = ctx.forward(images)
        err[flags] = rel(got, ref)
        if flags == 0:
            seen, thresh, tripped = ctx.ln_guard()
            assert ctx.ln_fold() and not tripped, (seen, thresh)
            x = ctx.debug_read(0, hid.size).reshape(hid.shape)
            # the planes reproduce the massive channel to the pair's precision (2^-15 of its value), not to fp16's 2^-11
            assert np.abs(x[..., 301] - hid[..., 301]).max() <= 2e-3 * np.abs(hid[..., 301]).max()
        ctx.close()
    print(f"\n[massive] one channel at ~2000: stand-alone {err[vithip.FLAG_LN_FOLD_OFF]:.3e}, folded + planes {err[0]:.3e}")
    assert np.isfinite(list(err.values())).all()
    assert err[0] <= max(NORTH_STAR, 1.25 * err[vithip.FLAG_LN_FOLD_OFF])
