"""GPU parity of the fp8 path (BASELINE config 5: ViT-B/16, e4m3 GEMM operands on v_mfma_scale_f32_16x16x128_f8f6f4).

What can be pinned exactly, is: the casts (bit-exact against the oracle's quantiser), the weight quantiser
(bit-exact), and the GEMM on GIVEN e4m3 operands (fp32 accumulation: only the summation order differs).
The end-to-end logits cannot be pinned to 1e-3: every cast rounds by up to 6 %, and a quantised pipeline is
chaotic (a last-bit difference ahead of a cast flips a rounding), so two correct implementations of the same
data flow drift apart to the fp8 noise level.  The end-to-end criterion is therefore statistical and written
out in test_logits_*: the device must be as close to the fp32 oracle as the oracle's own emulation of the fp8
data flow is (x1.5), and close to that emulation at the same level.
"""
import numpy as np
import pytest

import oracle_lib as O
import vh_synth as S

pytestmark = pytest.mark.gpu

vithip = pytest.importorskip("vithip")
FP8 = vithip.DTYPE_FP8
_KEEP = []


def dev(a):
    b = vithip.DeviceBuffer.from_numpy(a)
    _KEEP.append(b)
    return b


@pytest.fixture(autouse=True)
def _release_buffers():
    yield
    for b in _KEEP:
        b.free()
    _KEEP.clear()


def rel(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


def test_cast_is_bit_exact_e4m3_rne_saturating():
    n = 1 << 18
    x = (S.fill(n, 5, 9, 0) * np.exp2(np.floor(S.fill(n, 6, 9, 0) * 12))).astype(np.float32)
    x[:8] = [448.0, -448.0, 449.0, 1e9, -1e9, 0.0, 2.0 ** -10, 3 * 2.0 ** -10]
    out = vithip.DeviceBuffer(n)
    vithip.op_cast(dev(x).ptr, out.ptr, n, FP8)
    got = out.to_numpy(np.uint8, (n,))
    want = O.e4m3_bytes(x)
    same = (got == want) | ((x == 0) & ((got & 0x7F) == 0))      # -0 / +0
    assert same.all(), np.flatnonzero(~same)[:10]


def test_weight_quantiser_is_bit_exact():
    rows, cols = 300, 768
    w = S.fill(rows * cols, 11, 3, 1, 0.02).reshape(rows, cols)
    w[7] = 0.0
    w8 = vithip.DeviceBuffer(rows * cols)
    sc = vithip.DeviceBuffer(rows * 4)
    vithip.op_quantize_rows(dev(w).ptr, rows, cols, 0.125, w8.ptr, sc.ptr)
    want8, _, want_sc = O.quantize_rows(w, post=0.125)
    assert np.array_equal(sc.to_numpy(np.float32, (rows,)), want_sc)
    got8 = w8.to_numpy(np.uint8, (rows, cols))
    assert np.array_equal(got8 & 0x7F, want8 & 0x7F) and np.array_equal(got8[w != 0], want8[w != 0])


def _operands(M, N, K, seed):
    a = O.quant_e4m3((S.fill(M * K, seed, 1, 0) * 3.0).reshape(M, K))
    w = S.fill(N * K, seed, 2, 1, 0.02).reshape(N, K)
    w8, wq, sc = O.quantize_rows(w)
    bias = S.fill(N, seed, 3, 1, 0.1)
    return a, O.e4m3_bytes(a), w8, wq, sc, bias


@pytest.mark.parametrize("variant", [5, 7])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 260, 384), (1000, 768, 768), (64, 3072, 256), (5, 8, 128)])
def test_gemm_fp32_out_matches_fp32_accumulation(M, N, K, variant):
    a, a8, w8, wq, sc, bias = _operands(M, N, K, 21)
    ref = O.linear(a, wq) * sc[None, :] + bias[None, :]
    out = vithip.DeviceBuffer(M * N * 4)
    vithip.op_gemm_fp8(dev(a8).ptr, dev(w8).ptr, dev(sc).ptr, dev(bias).ptr, out.ptr, M, N, K, vithip.EPI_BIAS_F32, variant)
    got = out.to_numpy(np.float32, (M, N))
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-5 * scale, np.abs(got - ref).max() / scale
    # residual form: out += ...
    x0 = S.fill(M * N, 22, 4, 0).reshape(M, N)
    xb = dev(x0)
    vithip.op_gemm_fp8(dev(a8).ptr, dev(w8).ptr, dev(sc).ptr, dev(bias).ptr, xb.ptr, M, N, K, vithip.EPI_BIAS_RESID, variant)
    got = xb.to_numpy(np.float32, (M, N))
    assert np.abs(got - (x0 + ref)).max() <= 2e-5 * max(scale, 1.0)


def test_gemm_random_ragged_shapes_with_canary_rows():
    rng = np.random.default_rng(5)
    for i in range(10):
        M, N, K = int(rng.integers(1, 701)), 4 * int(rng.integers(1, 276)), 128 * int(rng.integers(1, 9))
        a, a8, w8, wq, sc, bias = _operands(M, N, K, 70 + i)
        pre = O.linear(a, wq) * sc[None, :] + bias[None, :]
        if i % 2 == 0:
            x0 = S.fill((M + 2) * N, 80 + i, 4, 0).reshape(M + 2, N)
            buf = dev(x0)
            vithip.op_gemm_fp8(dev(a8).ptr, dev(w8).ptr, dev(sc).ptr, dev(bias).ptr, buf.ptr + N * 4, M, N, K, vithip.EPI_BIAS_RESID,
                               5 + 2 * (i % 4 == 0))
            got = buf.to_numpy(np.float32, (M + 2, N))
            assert np.array_equal(got[0], x0[0]) and np.array_equal(got[-1], x0[-1]), (M, N, K)
            assert np.abs(got[1:-1] - (x0[1:-1] + pre)).max() <= 2e-5 * max(np.abs(pre).max(), 1.0), (M, N, K)
        else:
            canary = np.full((M + 2, N), 0x7B, np.uint8)
            buf = dev(canary)
            vithip.op_gemm_fp8(dev(a8).ptr, dev(w8).ptr, dev(sc).ptr, dev(bias).ptr, buf.ptr + N, M, N, K, vithip.EPI_BIAS_GELU)
            raw = buf.to_numpy(np.uint8, (M + 2, N))
            assert (raw[0] == 0x7B).all() and (raw[-1] == 0x7B).all(), (M, N, K)
            got8, want8 = vithip.from_e4m3(raw[1:-1]), O.quant_e4m3(O.gelu(pre))
            step = np.maximum(np.abs(want8), 2.0 ** -6) * 2.0 ** -3 + 1e-12
            assert np.all(np.abs(got8 - want8) <= step * 1.001), (M, N, K)


def test_gemm_integer_exact_asymmetric():
    # small integers and halves are exact in e4m3 and in the fp32 accumulator: any lane-map mistake shows
    M, N, K = 272, 264, 256
    rng = np.random.default_rng(3)
    a = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
    w = rng.integers(-3, 4, size=(N, K)).astype(np.float32) * 0.5
    a[:, 0] += 8.0 * (np.arange(M) % 3)            # asymmetric in m
    w[:, 1] += 1.0 * (np.arange(N) % 5)            # and in n
    a, w = O.quant_e4m3(a), O.quant_e4m3(w)
    sc = np.ones(N, np.float32)
    bias = np.zeros(N, np.float32)
    out = vithip.DeviceBuffer(M * N * 4)
    vithip.op_gemm_fp8(dev(O.e4m3_bytes(a)).ptr, dev(O.e4m3_bytes(w)).ptr, dev(sc).ptr, dev(bias).ptr, out.ptr, M, N, K,
                       vithip.EPI_BIAS_F32)
    assert np.array_equal(out.to_numpy(np.float32, (M, N)), a.astype(np.float64) @ w.astype(np.float64).T)


@pytest.mark.parametrize("M,N,K", [(512, 512, 256), (300, 260, 128), (197, 768, 768)])
def test_gemm_bf16_and_gelu_e4m3_outputs(M, N, K):
    a, a8, w8, wq, sc, bias = _operands(M, N, K, 31)
    pre = O.linear(a, wq) * sc[None, :] + bias[None, :]
    out16 = vithip.DeviceBuffer(M * N * 2)
    vithip.op_gemm_fp8(dev(a8).ptr, dev(w8).ptr, dev(sc).ptr, dev(bias).ptr, out16.ptr, M, N, K, vithip.EPI_BIAS)
    got = vithip.from_bf16_bits(out16.to_numpy(np.uint16, (M, N)))
    assert np.all(np.abs(got - pre) <= 2.0 ** -8 * np.abs(pre) * 1.01 + 2e-5 * np.abs(pre).max())
    # fc1 form: gelu -> e4m3.  Compare decoded values: within one e4m3 step of the oracle's, almost all identical
    # (fp32 summation order moves a value across a rounding boundary now and then)
    out8 = vithip.DeviceBuffer(M * N)
    vithip.op_gemm_fp8(dev(a8).ptr, dev(w8).ptr, dev(sc).ptr, dev(bias).ptr, out8.ptr, M, N, K, vithip.EPI_BIAS_GELU)
    got8 = vithip.from_e4m3(out8.to_numpy(np.uint8, (M, N)))
    g = O.gelu(pre)
    want8 = O.quant_e4m3(g)
    assert np.isfinite(got8).all()
    step = np.maximum(np.abs(want8), 2.0 ** -6) * 2.0 ** -3 + 1e-12
    assert np.all(np.abs(got8 - want8) <= step * 1.001)
    # (e4m3 results use the degree-6 GELU polynomial since round 4: |error| <= 5.7e-4, a quarter of e4m3's smallest step -- it
    #  moves a value across a rounding boundary a little more often than fp32 summation order alone did: 0.995 before)
    assert (got8 == want8).mean() >= 0.97, (got8 == want8).mean()


def test_layernorm_and_attention_write_e4m3():
    rows, dim = 333, 768
    x = S.fill(rows * dim, 41, 1, 0).reshape(rows, dim) * 2.0
    gm, bt = 1.0 + S.fill(dim, 41, 2, 1, 0.05), S.fill(dim, 41, 3, 1, 0.02)
    out = vithip.DeviceBuffer(rows * dim)
    vithip.op_layernorm(dev(x).ptr, rows, dim, dim, dev(gm).ptr, dev(bt).ptr, 1e-6, out.ptr, FP8)
    got = vithip.from_e4m3(out.to_numpy(np.uint8, (rows, dim)))
    ref = O.layernorm(x, gm, bt)
    want = O.quant_e4m3(ref)
    assert (got == want).mean() >= 0.999 and np.all(np.abs(got - ref) <= np.maximum(np.abs(ref), 2.0 ** -6) * 2.0 ** -4 * 1.05)
    batch, tokens, heads = 2, 197, 4
    D = heads * 64
    qkv = O.round_bf16((S.fill(batch * tokens * 3 * D, 42, 1, 0) * 1.5).reshape(batch * tokens, 3 * D))
    # q columns x VH_ATTN_Q_SCALE (64^-1/2 * log2 e) rounded to bf16 as the qkv GEMM delivers them; the oracle gets the
    # same q back in its own convention (test_gpu_ops.prescale_q)
    q_scale = np.float32(0.125 * 1.4426950408889634)
    pre = qkv.copy()
    pre[:, :D] = O.round_bf16(pre[:, :D] * q_scale)
    qkv = pre.astype(np.float64)
    qkv[:, :D] /= np.float64(q_scale)
    ref = O.attention(qkv.astype(np.float32), batch, tokens, heads)
    o8 = vithip.DeviceBuffer(batch * tokens * D)
    vithip.op_attention(dev(vithip.to16(pre, vithip.DTYPE_BF16)).ptr, batch, tokens, heads, o8.ptr, FP8)
    got = vithip.from_e4m3(o8.to_numpy(np.uint8, (batch * tokens, D)))
    assert np.isfinite(got).all()
    # bf16 P/O rounding inside the kernel (1.2e-2, test_gpu_ops.ATT_TOL) plus half an e4m3 step on the way out
    assert np.all(np.abs(got - ref) <= 2.0 ** -4 * np.abs(ref) + 1.2e-2 * np.abs(ref).max())


@pytest.mark.parametrize("name,batch,flags,folded", [("vit_q8", 4, 0, False), ("vit_base", 2, 0, True),
                                                     ("vit_base", 2, vithip.FLAG_LN_FOLD_OFF, False)])
def test_logits_track_the_fp8_emulation_and_the_fp32_forward(name, batch, flags, folded):
    # vit_q8 (mlp_dim 640) cannot fold its LayerNorms; ViT-B folds them by default (e4m3 copy of the RAW residual rows as
    # the operand of q|k|v and fc1, oracle_vit_forward_fp8_folded) and runs the stand-alone LayerNorm on request
    cfg = S.CONFIGS[name]
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, batch)
    ref32 = O.vit_forward(cfg, blob, images)
    emu = O.vit_forward(cfg, blob, images, fp8="folded" if folded else True)
    ctx = vithip.VitContext(cfg, dtype=FP8, max_batch=batch, flags=flags)
    assert ctx.ln_fold() == folded
    ctx.load_weights(blob)
    got = ctx.forward(images)
    ctx.close()
    e_emu32, e_gpu32, e_gpuemu = rel(emu, ref32), rel(got, ref32), rel(got, emu)
    # normalised RMS is the steadier statistic of a noisy pipeline; max-norm is reported next to it
    rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    r_emu32, r_gpu32, r_gpuemu = rms(emu, ref32), rms(got, ref32), rms(got, emu)
    top1 = float((got.argmax(1) == ref32.argmax(1)).mean())
    print(f"\n[fp8] {name} b{batch} {'folded LN' if folded else 'stand-alone LN'}: max-norm emu-fp32 {e_emu32:.3e} gpu-fp32 {e_gpu32:.3e} gpu-emu {e_gpuemu:.3e}; "
          f"rms emu-fp32 {r_emu32:.3e} gpu-fp32 {r_gpu32:.3e} gpu-emu {r_gpuemu:.3e}; top1 agree {top1:.2f}")
    assert np.isfinite(got).all()
    assert r_gpu32 <= 1.5 * r_emu32 + 1e-3          # as close to the truth as the emulated data flow is
    assert r_gpuemu <= 1.5 * r_emu32 + 1e-3         # and as close to the emulation
    assert e_gpu32 <= 0.25                            # sanity bound in the max-norm


def test_massive_activation_switches_the_fp8_context_to_the_stand_alone_layernorm():
    """The folded fp8 path feeds the RAW residual rows to q|k|v and fc1 as e4m3, which saturates at 448.  One channel at
    about 2 000 (the "massive activations" of trained ViTs, driven through the position embedding as in test_gpu_vit's
    fp16 case) would be clipped in the operand.  The guard's second word tracks max |x|: the calibration forward at weight
    load sees it and leaves the context on the stand-alone LayerNorm, whose NORMALISED operand fits e4m3 -- the device is
    then as close to fp32 as the oracle's emulation of that data flow; forced to stay folded it is far off."""
    cfg = S.CONFIGS["vit_base"]
    t = S.make_tensors(cfg, 0)
    t["pos"] = t["pos"].copy()
    t["pos"][..., 301] += np.float32(2000.0)
    blob = S.pack_blob(cfg, t)
    images = S.make_images(cfg, 1, 4)
    ref32 = O.vit_forward(cfg, blob, images)
    emu = O.vit_forward(cfg, blob, images, fp8=True)                  # stand-alone LayerNorm data flow
    rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    ctx = vithip.VitContext(cfg, dtype=FP8, max_batch=4)
    assert ctx.ln_fold()
    ctx.load_weights(blob)
    amax, lim = ctx.fp8_guard()
    _, _, tripped = ctx.ln_guard()
    assert lim == 448.0 and amax > 1500.0 and tripped and not ctx.ln_fold(), (amax, lim, tripped)
    d_in, d_out = vithip.DeviceBuffer.from_numpy(images), vithip.DeviceBuffer(4 * cfg["classes"] * 4)
    ctx.forward_device_async(d_in.ptr, 4, d_out.ptr, steps=1)        # first call of the asynchronous entry point
    ctx.synchronize()
    got = d_out.to_numpy(np.float32, (4, cfg["classes"]))
    ctx.close()
    forced = vithip.VitContext(cfg, dtype=FP8, max_batch=4, flags=vithip.FLAG_LN_FOLD_ON)
    forced.load_weights(blob)
    clipped = forced.forward(images)
    forced.close()
    r_emu32, r_gpu32, r_clip32 = rms(emu, ref32), rms(got, ref32), rms(clipped, ref32)
    print(f"\n[fp8 massive] channel at ~2000: max|x| bound {amax:.0f}; rms emu-fp32 {r_emu32:.3e} gpu-fp32 {r_gpu32:.3e}; "
          f"forced fold (operand clipped at 448) {r_clip32:.3e}")
    assert np.isfinite(got).all()
    assert r_gpu32 <= 1.5 * r_emu32 + 1e-3
    # the synthetic nets stay folded: their rows are O(1)
    ok = vithip.VitContext(cfg, dtype=FP8, max_batch=1)
    ok.load_weights(S.make_blob(cfg, 0))
    a2, _ = ok.fp8_guard()
    assert ok.ln_fold() and 0.0 < a2 < 448.0, a2
    ok.close()


def test_full_size_config_5_vit_base_fp8_batch_512_properties():
    # BASELINE.json config 5 at its FULL size: ViT-B/16, e4m3 GEMM operands, 512 images.  fp8 is outside the north
    # star's tolerance by construction, so the parity statement is the one of the small fp8 cases (the GPU follows the
    # oracle's e4m3 emulation of the same data flow as closely as that emulation follows fp32), made on rows of the
    # FULL-size run, next to the size-independent properties: determinism, sub-batch and permutation bit-exactness.
    cfg = S.CONFIGS["vit_base"]
    B = 512
    ctx = vithip.VitContext(cfg, dtype=FP8, max_batch=B)
    ctx.init_weights_seeded(0)
    blob = ctx.export_weights()
    din = vithip.DeviceBuffer(B * 224 * 224 * 3 * 4)
    dout = vithip.DeviceBuffer(B * cfg["classes"] * 4)
    ctx.fill_input_seeded(1, B, din.ptr)
    ctx.forward_device(din.ptr, B, dout.ptr)
    full = dout.to_numpy(np.float32, (B, cfg["classes"]))
    assert np.isfinite(full).all()
    ctx.forward_device(din.ptr, B, dout.ptr)
    assert np.array_equal(dout.to_numpy(np.float32, (B, cfg["classes"])), full)
    images = din.to_numpy(np.float32, (B, 224, 224, 3))
    for lo, hi in ((0, 5), (255, 258), (511, 512)):
        assert np.array_equal(ctx.forward(images[lo:hi]), full[lo:hi]), (lo, hi)
    perm = np.array([400, 9, 511, 0, 77])
    assert np.array_equal(ctx.forward(images[perm]), full[perm])
    ref32 = O.vit_forward(cfg, blob, images[:4])
    assert ctx.ln_fold()
    emu = O.vit_forward(cfg, blob, images[:4], fp8="folded")
    rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    r_emu32, r_gpu32, r_gpuemu = rms(emu, ref32), rms(full[:4], ref32), rms(full[:4], emu)
    print(f"\n[full size] vit_base b512 fp8: rows 0-3 rms emu-fp32 {r_emu32:.3e} gpu-fp32 {r_gpu32:.3e} gpu-emu {r_gpuemu:.3e}; "
          f"max-norm gpu-fp32 {rel(full[:4], ref32):.3e}")
    assert r_gpu32 <= 1.5 * r_emu32 + 1e-3 and r_gpuemu <= 1.5 * r_emu32 + 1e-3
    ctx.close()


def weight_only_e4m3_blob(cfg, blob):
    """The blob whose q/k/v/o/fc1/fc2 matrices went through the oracle's e4m3 row quantiser and back (fp32 values)."""
    out = blob.copy()
    off = 64
    for name, shape, *_ in S.tensor_table(cfg):
        n = int(np.prod(shape))
        if name.endswith(".weight") and name.split(".")[1] in ("q", "k", "v", "o", "fc1", "fc2"):
            w = out[off:off + 4 * n].view(np.float32).reshape(shape)
            _, wq, sc = O.quantize_rows(w)
            w[...] = wq * sc[:, None]
        off += 4 * n
    return out


@pytest.mark.parametrize("dt,name", [(vithip.DTYPE_FP16, "fp16"), (vithip.DTYPE_BF16, "bf16")])
def test_weight_only_e4m3_is_the_16_bit_forward_of_the_quantised_model(dt, name):
    # SURVEY.md section 7 option (a): e4m3 WEIGHTS (one scale per output channel), 16-bit activations, next to option
    # (b) = VH_DTYPE_FP8 (both operands e4m3).  Kernel parity: the flag's forward equals the plain 16-bit forward of
    # the model whose weights went through the quantiser on the host.  Accuracy price: both options against the fp32
    # oracle of the ORIGINAL weights, same images.
    cfg = S.CONFIGS["vit_base"]
    batch = 4
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, batch)
    ref32 = O.vit_forward(cfg, blob, images)
    blob_q = weight_only_e4m3_blob(cfg, blob)
    ref_q = O.vit_forward(cfg, blob_q, images)
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=batch, flags=vithip.FLAG_W8_E4M3)
    ctx.load_weights(blob)
    got = ctx.forward(images)
    ctx.close()
    ctx = vithip.VitContext(cfg, dtype=dt, max_batch=batch)
    ctx.load_weights(blob_q)
    host_quantised = ctx.forward(images)
    ctx.close()
    ctx = vithip.VitContext(cfg, dtype=FP8, max_batch=batch)
    ctx.load_weights(blob)
    both = ctx.forward(images)
    ctx.close()
    assert np.array_equal(got, host_quantised)                  # device quantiser == oracle quantiser, bit for bit
    e_kernel, e_a, e_b = rel(got, ref_q), rel(got, ref32), rel(both, ref32)
    print(f"\n[fp8 options] vit_base b{batch} {name}: (a) weight-only e4m3 vs its own fp32 model {e_kernel:.3e}, "
          f"vs the original fp32 model {e_a:.3e}; (b) both operands e4m3 vs the original fp32 model {e_b:.3e}")
    assert e_kernel <= (1e-3 if dt == vithip.DTYPE_FP16 else 1e-2)
    assert e_a < e_b
    with pytest.raises(vithip.VhError):
        vithip.VitContext(cfg, dtype=FP8, max_batch=1, flags=vithip.FLAG_W8_E4M3)


def test_fp8_is_deterministic_batch_independent_and_rejects_bad_dims():
    cfg = S.CONFIGS["vit_q8"]
    ctx = vithip.VitContext(cfg, dtype=FP8, max_batch=5)
    ctx.init_weights_seeded(3)
    images = S.make_images(cfg, 2, 5)
    full = ctx.forward(images)
    assert np.array_equal(full, ctx.forward(images))
    assert np.array_equal(ctx.forward(images[2:4]), full[2:4])
    ctx.set_streams(2)
    assert np.array_equal(ctx.forward(images), full)
    ctx.close()
    with pytest.raises(vithip.VhError):
        vithip.VitContext(S.CONFIGS["vit_tiny"], dtype=FP8, max_batch=1)   # dim 192 is not a multiple of 128


@pytest.mark.parametrize("batch", [256, 300])
def test_tiled_e4m3_hidden_activation_gives_the_same_bits_as_the_row_major_one(monkeypatch, batch):
    """Round 4: with e4m3 operands too, fc1's epilogue writes the (e4m3) hidden activation in the tiled layout of the e4m3 operand
    straight from its registers (a 4 x 4 dword transpose over the four 16-lane rows: two v_permlane32_swap, two v_permlane16_swap)
    and fc2's operand DMA reads it and a tiled copy of its weight bytes; the attention kernel stores its e4m3 result in the same
    layout for the out-projection (VH_ATT_TILED).  Byte order inside every 16-byte chunk is the natural one, so
    every MFMA sees the same k order: bit-identical logits (VH_H_TILED=0), on exact tiles and on rows padded to whole tiles; a batch
    too small for the persistent form takes the row-major path and gives the same bits again."""
    cfg = S.CONFIGS["vit_base"]
    px = cfg["image_size"] ** 2 * cfg["channels"]
    din, dout = vithip.DeviceBuffer(batch * px * 4), vithip.DeviceBuffer(batch * cfg["classes"] * 4)
    outs, used = [], []
    # row-major; h tiled only; the attention output (e4m3, tiled for out-proj) as well; q|k|v (bf16) head-major as well (the default)
    for tiled, att, hm in (("0", "1", "1"), ("1", "0", "1"), ("1", "1", "0"), ("1", "1", "1")):
        monkeypatch.setenv("VH_H_TILED", tiled)
        monkeypatch.setenv("VH_ATT_TILED", att)
        monkeypatch.setenv("VH_QKV_HM", hm)
        ctx = vithip.VitContext(cfg, dtype=FP8, max_batch=batch)
        ctx.init_weights_seeded(0)
        ctx.fill_input_seeded(1, batch, din.ptr)
        ctx.forward_device(din.ptr, batch, dout.ptr)
        outs.append(dout.to_numpy(np.float32, (batch, cfg["classes"])))
        used.append((int(ctx.debug_read(3, 1)[0]), int(ctx.debug_read(4, 1)[0])))
        small = ctx.forward(din.to_numpy(np.float32, (batch, cfg["image_size"], cfg["image_size"], cfg["channels"]))[:2])
        assert int(ctx.debug_read(3, 1)[0]) == 0 and np.array_equal(small, outs[-1][:2])
        ctx.close()
    assert used == [(0, 0), (1, 0), (1, 0), (1, 1)], used
    assert np.isfinite(outs[0]).all() and all(np.array_equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("epi", ["bias", "gelu", "resid"])
def test_gemm_persistent_form_equals_one_tile_per_workgroup_form(epi):
    """Variant 6 (persistent: a workgroup walks several tiles and prefetches the next tile's K-tiles from inside the
    epilogue; the default for the 16-bit / e4m3 results) against variant 5 on shapes with more tiles than CUs and a
    ragged last row of tiles: same arithmetic, same order, identical bytes."""
    for M, N, K in ((256 * 70 + 9, 1024, 256), (256 * 130, 768, 128)):
        a, a8, w8, wq, sc, bias = _operands(M, N, K, 31)
        A, W, SC, Bv = dev(a8), dev(w8), dev(sc), dev(bias)
        outs = []
        x0 = S.fill(M * N, 32, 4, 0).reshape(M, N) if epi == "resid" else None
        for variant in (5, 6):
            nbytes = M * N * {"bias": 2, "gelu": 1, "resid": 4}[epi]
            out = vithip.DeviceBuffer.from_numpy(x0) if epi == "resid" else vithip.DeviceBuffer(nbytes)
            vithip.op_gemm_fp8(A.ptr, W.ptr, SC.ptr, Bv.ptr, out.ptr, M, N, K,
                               {"bias": vithip.EPI_BIAS, "gelu": vithip.EPI_BIAS_GELU, "resid": vithip.EPI_BIAS_RESID}[epi], variant)
            outs.append(out.to_numpy(np.uint8, (nbytes,)).copy())
            out.free()
        assert np.array_equal(outs[0], outs[1]), f"{(outs[0] != outs[1]).sum()} bytes differ ({M}x{N}x{K})"


@pytest.mark.parametrize("variant", [5, 6])
def test_gemm_folded_layer_epilogues_on_e4m3_operands(variant):
    """The epilogues of the fp8 path's folded-LayerNorm layer loop, operator by operator (vh_op_gemm_fp8_ex), against fp64
    arithmetic on the same e4m3 operands: LNFOLD (bf16 out), RESID_LN (fp32 residual + e4m3 copy + partial row sums) and
    RESID_SPLIT (the residual as an e4m3 plane -- the next GEMM's operand -- plus a bf16 plane: 3 bytes per element)."""
    M, N, K = 256 * 3 + 57, 512, 256
    a, a8, w8, wq, sc, bias = _operands(M, N, K, 47)
    pre = (O.linear(a, wq) * sc[None, :]).astype(np.float64)
    A, W, SC, Bv = dev(a8), dev(w8), dev(sc), dev(bias)
    scale = np.abs(pre).max()
    # ---- LNFOLD: rstd * (s * acc - mean * c) + d
    st = np.stack([S.fill(M, 48, 1, 1, 0.1), 1.0 + np.abs(S.fill(M, 48, 2, 0))], axis=1).astype(np.float32)
    cvec = S.fill(N, 48, 3, 1, 0.1)
    out = vithip.DeviceBuffer(M * N * 2)
    vithip.op_gemm_fp8_ex(A.ptr, W.ptr, SC.ptr, Bv.ptr, out.ptr, M, N, K, vithip.EPI_LNFOLD, c_ptr=dev(cvec).ptr,
                          stats_ptr=dev(st).ptr, variant=variant)
    want = st[:, 1:2] * (pre - st[:, 0:1] * cvec[None, :]) + bias[None, :]
    got = vithip.from_bf16_bits(out.to_numpy(np.uint16, (M, N)))
    assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) * 1.01 + 3e-5 * np.abs(want).max())
    # ---- RESID_LN: fp32 x += ...; e4m3 copy; partial sums
    x0 = (S.fill((M + 2) * N, 48, 4, 0) * 3.0).reshape(M + 2, N)
    ref = x0[1:-1].astype(np.float64) + pre + bias[None, :]
    xb, c8, parts = dev(x0.copy()), vithip.DeviceBuffer(M * N), vithip.DeviceBuffer((N // 64) * M * 8)
    vithip.op_gemm_fp8_ex(A.ptr, W.ptr, SC.ptr, Bv.ptr, xb.ptr + N * 4, M, N, K, vithip.EPI_RESID_LN, out16_ptr=c8.ptr,
                          partials_ptr=parts.ptr, variant=variant)
    x1 = xb.to_numpy(np.float32, (M + 2, N))
    assert np.array_equal(x1[0], x0[0]) and np.array_equal(x1[-1], x0[-1])
    assert np.abs(x1[1:-1] - ref).max() <= 3e-5 * max(scale, np.abs(ref).max())
    g8, w8q = vithip.from_e4m3(c8.to_numpy(np.uint8, (M, N))), O.quant_e4m3(x1[1:-1])
    assert (g8 == w8q).all()
    p = parts.to_numpy(np.float32, (N // 64, M, 2))
    r64 = ref.reshape(M, N // 64, 64)
    assert np.abs(p[:, :, 0].T - r64.sum(2)).max() <= 2e-4 * np.abs(ref).max()
    assert np.abs(p[:, :, 1].T - (r64 ** 2).sum(2)).max() <= 1e-4 * (r64 ** 2).sum(2).max()
    c8.free(); parts.free()
    # ---- RESID_SPLIT: hi = e4m3(x) (saturating), lo = bf16(x - hi)
    x0[5, :7] = [600.0, -700.0, 448.0, 449.0, 0.0, 1e-5, -3e-4]          # beyond e4m3's range: lo must carry the rest
    hi0 = O.quant_e4m3(x0)
    lo0 = vithip.from_bf16_bits(vithip.to_bf16_bits(x0 - hi0))
    ref = (hi0 + lo0)[1:-1].astype(np.float64) + pre + bias[None, :]
    hb, lb = dev(O.e4m3_bytes(x0)), dev(vithip.to_bf16_bits(x0 - hi0))
    parts = vithip.DeviceBuffer((N // 64) * M * 8)
    vithip.op_gemm_fp8_ex(A.ptr, W.ptr, SC.ptr, Bv.ptr, hb.ptr + N, M, N, K, vithip.EPI_RESID_SPLIT, out16_ptr=lb.ptr + N * 2,
                          partials_ptr=parts.ptr, variant=variant)
    h = vithip.from_e4m3(hb.to_numpy(np.uint8, (M + 2, N)))
    l = vithip.from_bf16_bits(lb.to_numpy(np.uint16, (M + 2, N)))
    for plane, before in ((h, hi0), (l, lo0)):
        assert np.array_equal(plane[0], before[0]) and np.array_equal(plane[-1], before[-1])   # canary rows
    got = h[1:-1].astype(np.float64) + l[1:-1]
    big = np.abs(ref).max()
    # the pair: bf16's half ulp of a residue of at most half an e4m3 step (1/16 of |x|; more where hi saturates)
    resid = np.maximum(np.abs(ref) * 2.0 ** -4, np.abs(ref) - 448.0)
    assert (np.abs(got - ref) <= resid * 2.0 ** -8 * 1.05 + 3e-5 * big).all()
    # hi is the e4m3 rounding of the updated value (up to the last-bit effect of the summation order)
    want_h = O.quant_e4m3(ref.astype(np.float32))
    step = np.maximum(np.abs(want_h), 2.0 ** -6) * 2.0 ** -3 + 1e-12
    assert np.all(np.abs(h[1:-1] - want_h) <= step * 1.001) and (h[1:-1] == want_h).mean() >= 0.995
    p = parts.to_numpy(np.float32, (N // 64, M, 2))
    r64 = ref.reshape(M, N // 64, 64)
    assert np.abs(p[:, :, 0].T - r64.sum(2)).max() <= 2e-4 * big
    assert np.abs(p[:, :, 1].T - (r64 ** 2).sum(2)).max() <= 1e-4 * (r64 ** 2).sum(2).max()
    parts.free(); out.free()
