"""GPU parity, operator level: every HIP kernel of the hot path, driven alone through the C ABI
(vh_op_*), against the CPU oracle on the same seeded inputs.

Tolerances (written here, per the north star "within 1e-3 relative fp32" for the end-to-end
logits; operator outputs are 16-bit so the bound is the 16-bit rounding step):
  bf16 output: 1 ulp = 2^-8 relative  -> |d| <= 2^-8 |ref| + eps
  fp16 output: 1 ulp = 2^-11 relative
  fp32 output of a GEMM with 16-bit-exact inputs: only summation order differs -> 2e-5 * scale
"""
import numpy as np
import pytest

import oracle_lib as O
import vh_synth as S

pytestmark = pytest.mark.gpu

vithip = pytest.importorskip("vithip")
DT = [vithip.DTYPE_BF16, vithip.DTYPE_FP16]
ULP = {vithip.DTYPE_BF16: 2.0 ** -8, vithip.DTYPE_FP16: 2.0 ** -11}


def rnd16(a, dt):
    return O.round_bf16(a) if dt == vithip.DTYPE_BF16 else O.round_fp16(a)


_KEEP = []


def dev(a):
    """Upload and keep the allocation alive until the test ends (a temporary DeviceBuffer would be
    freed by its destructor before the kernel that reads it runs)."""
    b = vithip.DeviceBuffer.from_numpy(a)
    _KEEP.append(b)
    return b


@pytest.fixture(autouse=True)
def _release_buffers():
    yield
    for b in _KEEP:
        b.free()
    _KEEP.clear()


def assert_close16(got, ref, dt, extra=0.0):
    """got is a 16-bit kernel output (as fp32), ref the fp32 oracle value."""
    tol = ULP[dt] * np.abs(ref) * 1.01 + extra + 1e-30
    bad = np.abs(got - ref) > tol
    assert not bad.any(), (f"{bad.sum()} / {bad.size} elements outside 1 ulp; worst "
                           f"{np.abs(got - ref).max():.3e} at ref {ref.flat[np.abs(got - ref).argmax()]:.3e}")


def test_fill_matches_numpy_and_oracle_bitwise():
    n = 100003
    for kind, sigma, tid in ((0, 0.0, 0x100), (1, 0.02, 7), (1, 0.05, 0x7001)):
        buf = vithip.DeviceBuffer(n * 4)
        vithip.op_fill(buf.ptr, n, 12345, tid, kind, sigma)
        got = buf.to_numpy(np.float32, (n,))
        assert np.array_equal(got, S.fill(n, 12345, tid, kind, sigma))
        assert np.array_equal(got, O.fill(n, 12345, tid, kind, sigma))


@pytest.mark.parametrize("dt", DT)
def test_cast_is_round_to_nearest_even(dt):
    x = np.concatenate([S.fill(4096, 1, 1, 0) * 3.0, np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 65504.0, 1e-8, -1e-8],
                                                                 dtype=np.float32)])
    out = vithip.DeviceBuffer(x.size * 2)
    vithip.op_cast(dev(x).ptr, out.ptr, x.size, dt)
    got = out.to_numpy(np.uint16, x.shape)
    assert np.array_equal(got, vithip.to16(x, dt))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("image,patch,batch", [(64, 16, 3), (224, 16, 2), (96, 16, 1), (64, 32, 2)])
def test_im2col(dt, image, patch, batch):
    x = S.fill(batch * image * image * 3, 5, 9, 0).reshape(batch, image, image, 3)
    ref = rnd16(O.im2col(x, patch), dt)
    out = vithip.DeviceBuffer(ref.size * 2)
    vithip.op_im2col(dev(x).ptr, batch, image, patch, 3, out.ptr, dt)
    got = vithip.from16(out.to_numpy(np.uint16, ref.shape), dt)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,dim", [(5, 128), (197, 192), (1001, 768), (64, 1024), (3, 64)])
def test_layernorm(dt, rows, dim):
    x = (S.fill(rows * dim, 3, 1, 0) * 2.0 + 0.3).reshape(rows, dim)
    g = S.fill(dim, 3, 2, 1, 0.05, 1.0)
    b = S.fill(dim, 3, 3, 1, 0.02, 0.0)
    ref = O.layernorm(x, g, b, 1e-6)
    out = vithip.DeviceBuffer(rows * dim * 2)
    vithip.op_layernorm(dev(x).ptr, rows, dim, dim, dev(g).ptr, dev(b).ptr, 1e-6, out.ptr, dt)
    got = vithip.from16(out.to_numpy(np.uint16, (rows, dim)), dt)
    assert_close16(got, ref, dt, extra=2e-6)


@pytest.mark.parametrize("dt", DT)
def test_layernorm_strided_rows(dt):
    # the final LN reads only token 0 of every image: row stride = T*D
    B, T, D = 4, 17, 128
    x = S.fill(B * T * D, 4, 1, 0).reshape(B, T, D)
    g = S.fill(D, 4, 2, 1, 0.05, 1.0)
    b = S.fill(D, 4, 3, 1, 0.02, 0.0)
    ref = O.layernorm(x[:, 0, :].copy(), g, b, 1e-6)
    out = vithip.DeviceBuffer(B * D * 2)
    vithip.op_layernorm(dev(x).ptr, B, D, T * D, dev(g).ptr, dev(b).ptr, 1e-6, out.ptr, dt)
    assert_close16(vithip.from16(out.to_numpy(np.uint16, (B, D)), dt), ref, dt, extra=2e-6)


GEMM_SHAPES = [  # M, N, K — ragged M, N not a tile multiple, every K class of the forward
    (197, 576, 192), (394, 192, 768), (51, 40, 128), (256, 256, 64), (300, 1000, 768),
    (1024, 768, 768), (777, 2304, 768), (512, 768, 3072), (34, 128, 768),
]


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7])
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_bias_f32(dt, variant, M, N, K):
    a = rnd16(S.fill(M * K, 7, 1, 0).reshape(M, K), dt)
    w = rnd16(S.fill(N * K, 7, 2, 1, 0.05).reshape(N, K), dt)
    bias = S.fill(N, 7, 3, 1, 0.1)
    ref = O.linear(a, w, bias)
    out = vithip.DeviceBuffer(M * N * 4)
    vithip.op_gemm(dev(vithip.to16(a, dt)).ptr, dev(vithip.to16(w, dt)).ptr, dev(bias).ptr, out.ptr, M, N, K,
                   vithip.EPI_BIAS_F32, dt, variant=variant)
    got = out.to_numpy(np.float32, (M, N))
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-5 * scale, np.abs(got - ref).max() / scale


def _random_shapes(n, seed, kstep):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 701)), 4 * int(rng.integers(1, 276)), kstep * int(rng.integers(1, 1024 // kstep + 1))) for _ in range(n)]


@pytest.mark.parametrize("variant", [1, 5, 7])
def test_gemm_random_ragged_shapes_with_canary_rows(variant):
    # 16 seeded random (M, N, K): every epilogue class on shapes no tile divides, and the output buffer carries one
    # extra row of canaries on each side that no store may touch
    for i, (M, N, K) in enumerate(_random_shapes(16, 100 + variant, 64)):
        dt = DT[i % 2]
        a = rnd16(S.fill(M * K, 30 + i, 1, 0).reshape(M, K), dt)
        w = rnd16(S.fill(N * K, 30 + i, 2, 1, 0.05).reshape(N, K), dt)
        bias = S.fill(N, 30 + i, 3, 1, 0.1)
        pre = O.linear(a, w, bias)
        A, W, Bv = dev(vithip.to16(a, dt)), dev(vithip.to16(w, dt)), dev(bias)
        if i % 3 == 0:      # fp32 out, residual form
            x0 = S.fill((M + 2) * N, 60 + i, 4, 0).reshape(M + 2, N)
            buf = dev(x0)
            vithip.op_gemm(A.ptr, W.ptr, Bv.ptr, buf.ptr + N * 4, M, N, K, vithip.EPI_BIAS_RESID, dt, variant=variant)
            got = buf.to_numpy(np.float32, (M + 2, N))
            assert np.array_equal(got[0], x0[0]) and np.array_equal(got[-1], x0[-1]), (M, N, K)
            assert np.abs(got[1:-1] - (x0[1:-1] + pre)).max() <= 2e-5 * max(np.abs(pre).max(), 1.0), (M, N, K)
        else:               # 16-bit out, plain or GELU
            gelu = i % 3 == 2
            canary = np.full((M + 2, N), 0x7B7B, np.uint16)
            buf = dev(canary)
            vithip.op_gemm(A.ptr, W.ptr, Bv.ptr, buf.ptr + N * 2, M, N, K, vithip.EPI_BIAS_GELU if gelu else vithip.EPI_BIAS, dt,
                           variant=variant)
            raw = buf.to_numpy(np.uint16, (M + 2, N))
            assert (raw[0] == 0x7B7B).all() and (raw[-1] == 0x7B7B).all(), (M, N, K)
            assert_close16(vithip.from16(raw[1:-1], dt), O.gelu(pre) if gelu else pre, dt, extra=4e-5 * np.abs(pre).max())


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7])
@pytest.mark.parametrize("dt", DT)
def test_gemm_integer_exact_asymmetric(dt, variant):
    # exact small-integer data with an asymmetric W catches any row/col or k-order mix-up bitwise
    M, N, K = 300, 260, 128
    rng = np.random.default_rng(0)
    a = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
    w = rng.integers(-4, 5, size=(N, K)).astype(np.float32)
    bias = np.arange(N, dtype=np.float32)
    ref = a @ w.T + bias
    out = vithip.DeviceBuffer(M * N * 4)
    vithip.op_gemm(dev(vithip.to16(a, dt)).ptr, dev(vithip.to16(w, dt)).ptr, dev(bias).ptr, out.ptr, M, N, K,
                   vithip.EPI_BIAS_F32, dt, variant=variant)
    assert np.array_equal(out.to_numpy(np.float32, (M, N)), ref)


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7])
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("epi", ["bias", "gelu", "resid"])
def test_gemm_epilogues(dt, variant, epi):
    M, N, K = 333, 320, 192
    a = rnd16(S.fill(M * K, 8, 1, 0).reshape(M, K), dt)
    w = rnd16(S.fill(N * K, 8, 2, 1, 0.1).reshape(N, K), dt)
    bias = S.fill(N, 8, 3, 1, 0.1)
    lin = O.linear(a, w, bias)
    A, W, Bv = dev(vithip.to16(a, dt)), dev(vithip.to16(w, dt)), dev(bias)
    if epi == "resid":
        x0 = S.fill(M * N, 8, 4, 0).reshape(M, N)
        out = dev(x0)
        vithip.op_gemm(A.ptr, W.ptr, Bv.ptr, out.ptr, M, N, K, vithip.EPI_BIAS_RESID, dt, variant=variant)
        got = out.to_numpy(np.float32, (M, N))
        ref = x0 + lin
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    else:
        ref = lin if epi == "bias" else O.gelu(lin)
        out = vithip.DeviceBuffer(M * N * 2)
        vithip.op_gemm(A.ptr, W.ptr, Bv.ptr, out.ptr, M, N, K,
                       vithip.EPI_BIAS if epi == "bias" else vithip.EPI_BIAS_GELU, dt, variant=variant)
        got = vithip.from16(out.to_numpy(np.uint16, (M, N)), dt)
        assert_close16(got, ref, dt, extra=3e-5 * np.abs(lin).max())


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7])
@pytest.mark.parametrize("dt", DT)
def test_gemm_patch_epilogue(dt, variant):
    # rows of the patch matrix are remapped to token rows 1..NP of each image, + pos-emb
    B, NP, N, K = 3, 16, 128, 768
    T = NP + 1
    a = rnd16(S.fill(B * NP * K, 9, 1, 0).reshape(B * NP, K), dt)
    w = rnd16(S.fill(N * K, 9, 2, 1, 0.02).reshape(N, K), dt)
    bias = S.fill(N, 9, 3, 1, 0.02)
    pos = S.fill(T * N, 9, 4, 1, 0.02).reshape(T, N)
    lin = O.linear(a, w, bias).reshape(B, NP, N)
    x = np.full((B, T, N), 7.5, dtype=np.float32)
    ref = x.copy()
    ref[:, 1:, :] = lin + pos[None, 1:, :]
    out = dev(x)
    vithip.op_gemm(dev(vithip.to16(a, dt)).ptr, dev(vithip.to16(w, dt)).ptr, dev(bias).ptr, out.ptr, B * NP, N, K,
                   vithip.EPI_PATCH, dt, aux_ptr=dev(pos).ptr, aux_i=NP, variant=variant)
    got = out.to_numpy(np.float32, (B, T, N))
    assert np.array_equal(got[:, 0, :], x[:, 0, :])  # CLS rows untouched
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


def test_gemm_rejects_bad_shapes():
    b = vithip.DeviceBuffer(1 << 16)
    with pytest.raises(vithip.VhError):
        vithip.op_gemm(b.ptr, b.ptr, b.ptr, b.ptr, 16, 16, 48, vithip.EPI_BIAS, vithip.DTYPE_BF16)  # K % 64
    with pytest.raises(vithip.VhError):
        vithip.op_gemm(b.ptr, b.ptr, b.ptr, b.ptr, 16, 18, 64, vithip.EPI_BIAS, vithip.DTYPE_BF16)  # N % 4
    with pytest.raises(vithip.VhError):
        vithip.op_gemm(b.ptr, b.ptr, b.ptr, b.ptr, 0, 16, 64, vithip.EPI_BIAS, vithip.DTYPE_BF16)   # empty


Q_SCALE = np.float32(0.125 * 1.4426950408889634)   # VH_ATTN_Q_SCALE: 64^-1/2 * log2(e), folded into Wq by the forward


def prescale_q(qkv, D, dt):
    """The kernel's input (q columns x Q_SCALE, rounded to the storage type as the qkv GEMM would) and the oracle's
    input that corresponds to it EXACTLY (the rounded q divided by the scale again, in fp32/fp64): the rounding of the
    scaled q belongs to the GEMM in front, not to the attention kernel under test."""
    pre = qkv.astype(np.float32).copy()
    pre[:, :D] = rnd16(pre[:, :D] * Q_SCALE, dt)
    ref_in = pre.astype(np.float64)
    ref_in[:, :D] /= np.float64(Q_SCALE)   # the oracle scales by 64^-1/2 itself and works in base e
    return pre, ref_in.astype(np.float32)


ATT_TOL = {vithip.DTYPE_BF16: 1.2e-2, vithip.DTYPE_FP16: 1.5e-3}  # P and O are rounded to 16 bit


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("batch,tokens,heads", [(2, 17, 2), (1, 37, 3), (2, 197, 3), (1, 32, 1), (1, 64, 2),
                                                (1, 1, 1), (1, 257, 2), (1, 577, 2)])
def test_attention(dt, batch, tokens, heads):
    D = heads * 64
    qkv = rnd16((S.fill(batch * tokens * 3 * D, 10, 1, 0) * 1.5).reshape(batch * tokens, 3 * D), dt)
    pre, qkv = prescale_q(qkv, D, dt)
    ref = O.attention(qkv, batch, tokens, heads)
    out = vithip.DeviceBuffer(batch * tokens * D * 2)
    vithip.op_attention(dev(vithip.to16(pre, dt)).ptr, batch, tokens, heads, out.ptr, dt)
    got = vithip.from16(out.to_numpy(np.uint16, (batch * tokens, D)), dt)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err <= ATT_TOL[dt], err


def test_attention_token_counts_around_every_kernel_boundary():
    # one-shot form (T <= 96), staged ring (97..~208), plain ring (1 slab up to 384, 2 slabs beyond), LDS limit (640):
    # token counts on both sides of each boundary and of the 32-key tile / 8-row group edges, several items per workgroup
    # where that is cheap (fp16, the tighter tolerance)
    dt = vithip.DTYPE_FP16
    for tokens, batch, heads in ((33, 3, 2), (64, 2, 1), (65, 2, 2), (96, 5, 1), (97, 5, 1), (100, 700, 1), (128, 3, 2), (129, 2, 1),
                                 (160, 2, 1), (161, 300, 2), (193, 2, 1), (200, 2, 3), (208, 2, 1), (209, 2, 1), (224, 300, 1),
                                 (225, 2, 1), (256, 2, 2), (289, 1, 2), (384, 1, 2), (385, 1, 3), (512, 1, 1), (513, 2, 1),
                                 (608, 1, 2), (609, 1, 1), (640, 1, 2)):
        D = heads * 64
        qkv = rnd16((S.fill(batch * tokens * 3 * D, 13, tokens, 0) * 1.5).reshape(batch * tokens, 3 * D), dt)
        pre, qkv = prescale_q(qkv, D, dt)
        ref = O.attention(qkv, batch, tokens, heads)
        out = vithip.DeviceBuffer(batch * tokens * D * 2)
        vithip.op_attention(dev(vithip.to16(pre, dt)).ptr, batch, tokens, heads, out.ptr, dt)
        got = vithip.from16(out.to_numpy(np.uint16, (batch * tokens, D)), dt)
        err = np.abs(got - ref).reshape(batch, -1).max(1) / np.abs(ref).max()
        assert np.isfinite(got).all() and err.max() <= ATT_TOL[dt], (tokens, batch, heads, int(err.argmax()), float(err.max()))


@pytest.mark.parametrize("batch,tokens,heads,what", [
    (92, 197, 12, "staged ring: 1104 items on 512 workgroups = two static items each, then 80 tickets from the work queue"),
    (25, 577, 12, "plain ring: 300 heads x 2 slabs on 256 workgroups = K/V shared by the slabs, refilled for a second head")])
def test_attention_persistent_workgroups_walk_several_items(batch, tokens, heads, what):
    # the shapes of test_attention give every workgroup ONE item; these make the persistent loops, the ring refill across
    # items, the slab sharing and the device work queue do real work (fp16: the tighter tolerance)
    dt = vithip.DTYPE_FP16
    D = heads * 64
    qkv = rnd16((S.fill(batch * tokens * 3 * D, 12, 1, 0) * 1.5).reshape(batch * tokens, 3 * D), dt)
    pre, qkv = prescale_q(qkv, D, dt)
    ref = O.attention(qkv, batch, tokens, heads)
    out = vithip.DeviceBuffer(batch * tokens * D * 2)
    for _ in range(2):   # twice: the queue counter is re-armed by every launch
        vithip.op_attention(dev(vithip.to16(pre, dt)).ptr, batch, tokens, heads, out.ptr, dt)
        got = vithip.from16(out.to_numpy(np.uint16, (batch * tokens, D)), dt)
        assert np.isfinite(got).all()
        per_image = np.abs(got - ref).reshape(batch, -1).max(1) / np.abs(ref).max()
        assert per_image.max() <= ATT_TOL[dt], (what, int(per_image.argmax()), float(per_image.max()))


@pytest.mark.parametrize("tokens", [197, 577])   # the staged ring; the multi-slab ring of the long sequences (round 4)
@pytest.mark.parametrize("dt", DT)
def test_attention_spiked_scores_force_rescale(dt, tokens):
    # one key row aligned with one query row far above the rest: the running max jumps in a late
    # tile, exercising the online-softmax rescale of O and l (rule: a rare branch needs its own test)
    batch, heads = 1, 1
    D = 64
    qkv = (S.fill(tokens * 3 * D, 11, 1, 0) * 0.5).reshape(tokens, 3 * D)
    qkv[5, :D] = 4.0
    qkv[170, D:2 * D] = 4.0     # score(5,170) = 64*16/8 = 128 >> others
    qkv[40, :D] = -3.0
    qkv[3, D:2 * D] = -3.0      # early spike for query 40, then nothing larger
    qkv = rnd16(qkv, dt)
    pre, qkv = prescale_q(qkv, D, dt)
    ref = O.attention(qkv, batch, tokens, heads)
    out = vithip.DeviceBuffer(tokens * D * 2)
    vithip.op_attention(dev(vithip.to16(pre, dt)).ptr, batch, tokens, heads, out.ptr, dt)
    got = vithip.from16(out.to_numpy(np.uint16, (tokens, D)), dt)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.abs(ref).max() <= ATT_TOL[dt]
    # query 5 attends essentially only to key 170
    assert np.abs(got[5] - qkv[170, 2 * D:]).max() <= 2 * ULP[dt] * np.abs(qkv[170, 2 * D:]).max() + 1e-6


# ---- LayerNorm folded into the GEMMs (LNFOLD / RESID_LN epilogues + helpers) -----------------------------------------
@pytest.mark.parametrize("dt", DT)
def test_fold_ln_weights(dt):
    rows, dim, scale = 70, 192, 0.125
    w = S.fill(rows * dim, 20, 1, 1, 0.05).reshape(rows, dim)
    b = S.fill(rows, 20, 2, 1, 0.05)
    g = S.fill(dim, 20, 3, 1, 0.05, 1.0)
    be = S.fill(dim, 20, 4, 1, 0.05)
    w16, c, d = vithip.DeviceBuffer(rows * dim * 2), vithip.DeviceBuffer(rows * 4), vithip.DeviceBuffer(rows * 4)
    vithip.op_fold_ln(dev(w).ptr, dev(b).ptr, dev(g).ptr, dev(be).ptr, rows, dim, scale, w16.ptr, c.ptr, d.ptr, dt)
    got_w = vithip.from16(w16.to_numpy(np.uint16, (rows, dim)), dt)
    want_w = rnd16((np.float32(scale) * g)[None, :] * w, dt)
    assert np.array_equal(got_w, want_w)
    assert np.abs(c.to_numpy(np.float32, (rows,)) - want_w.astype(np.float64).sum(1)).max() <= 1e-5
    want_d = scale * ((w.astype(np.float64) * be[None, :]).sum(1) + b)
    assert np.abs(d.to_numpy(np.float32, (rows,)) - want_d).max() <= 1e-6


@pytest.mark.parametrize("dt", DT)
def test_rowstats_cast_and_finalize(dt):
    rows, dim = 301, 768
    x = (S.fill(rows * dim, 21, 1, 0) * 2.0 + 0.4).reshape(rows, dim)
    x16, st = vithip.DeviceBuffer(rows * dim * 2), vithip.DeviceBuffer(rows * 8)
    vithip.op_rowstats_cast(dev(x).ptr, rows, dim, 1e-6, x16.ptr, st.ptr, dt)
    assert np.array_equal(x16.to_numpy(np.uint16, (rows, dim)), vithip.to16(x, dt))
    x64 = x.astype(np.float64)
    want = np.stack([x64.mean(1), 1.0 / np.sqrt(x64.var(1) + 1e-6)], 1)
    assert np.abs(st.to_numpy(np.float32, (rows, 2)) - want).max() <= 2e-6 * np.abs(want).max()
    nblk = dim // 64
    parts = np.stack([np.stack([x64[:, i * 64:(i + 1) * 64].sum(1), (x64[:, i * 64:(i + 1) * 64] ** 2).sum(1)], 1) for i in range(nblk)])
    st2 = vithip.DeviceBuffer(rows * 8)
    vithip.op_finalize_stats(dev(parts.astype(np.float32)).ptr, nblk, rows, dim, 1e-6, st2.ptr)
    assert np.abs(st2.to_numpy(np.float32, (rows, 2)) - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7])
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_lnfold_equals_layernorm_then_linear(dt, variant, gelu):
    M, N, K = 333, 512, 256
    x = (S.fill(M * K, 22, 1, 0) * 1.5 + 0.2).reshape(M, K)
    w = S.fill(N * K, 22, 2, 1, 0.05).reshape(N, K)
    b = S.fill(N, 22, 3, 1, 0.05)
    g = S.fill(K, 22, 4, 1, 0.05, 1.0)
    be = S.fill(K, 22, 5, 1, 0.05)
    w16, c, d = vithip.DeviceBuffer(N * K * 2), vithip.DeviceBuffer(N * 4), vithip.DeviceBuffer(N * 4)
    vithip.op_fold_ln(dev(w).ptr, dev(b).ptr, dev(g).ptr, dev(be).ptr, N, K, 1.0, w16.ptr, c.ptr, d.ptr, dt)
    x16, st = vithip.DeviceBuffer(M * K * 2), vithip.DeviceBuffer(M * 8)
    vithip.op_rowstats_cast(dev(x).ptr, M, K, 1e-6, x16.ptr, st.ptr, dt)
    out = vithip.DeviceBuffer(M * N * 2)
    vithip.op_gemm_ex(x16.ptr, w16.ptr, d.ptr, out.ptr, M, N, K, vithip.EPI_LNFOLD_GELU if gelu else vithip.EPI_LNFOLD, dt,
                      aux_ptr=c.ptr, stats_ptr=st.ptr, variant=variant)
    got = vithip.from16(out.to_numpy(np.uint16, (M, N)), dt)
    # (1) exact statement of what the kernel computes (same rounded operands), float64
    x64, xr = x.astype(np.float64), rnd16(x, dt).astype(np.float64)
    mean, rstd = x64.mean(1), 1.0 / np.sqrt(x64.var(1) + 1e-6)
    wr = rnd16(g[None, :] * w, dt).astype(np.float64)
    lin = rstd[:, None] * (xr @ wr.T - mean[:, None] * wr.sum(1)[None, :]) + ((w.astype(np.float64) * be[None, :]).sum(1) + b)[None, :]
    ref = O.gelu(lin.astype(np.float32)) if gelu else lin.astype(np.float32)
    assert_close16(got, ref, dt, extra=5e-5 * np.abs(lin).max())
    # (2) and it is LayerNorm followed by the linear layer (oracle, fp32) up to the 16-bit operand rounding
    true = O.linear(O.layernorm(x, g, be, 1e-6), w, b)
    true = O.gelu(true) if gelu else true
    assert np.abs(got - true).max() / np.abs(true).max() <= (3e-2 if dt == vithip.DTYPE_BF16 else 4e-3)


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7])
@pytest.mark.parametrize("dt", DT)
def test_gemm_resid_ln_epilogue(dt, variant):
    M, N, K = 400, 512, 192
    a = rnd16(S.fill(M * K, 23, 1, 0).reshape(M, K), dt)
    w = rnd16(S.fill(N * K, 23, 2, 1, 0.1).reshape(N, K), dt)
    bias = S.fill(N, 23, 3, 1, 0.1)
    x0 = S.fill(M * N, 23, 4, 0).reshape(M, N)
    ref = x0 + O.linear(a, w, bias)
    out, out16, parts = dev(x0), vithip.DeviceBuffer(M * N * 2), vithip.DeviceBuffer((N // 64) * M * 8)
    vithip.op_gemm_ex(dev(vithip.to16(a, dt)).ptr, dev(vithip.to16(w, dt)).ptr, dev(bias).ptr, out.ptr, M, N, K,
                      vithip.EPI_RESID_LN, dt, out16_ptr=out16.ptr, partials_ptr=parts.ptr, variant=variant)
    got = out.to_numpy(np.float32, (M, N))
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    assert np.array_equal(out16.to_numpy(np.uint16, (M, N)), vithip.to16(got, dt))       # exact cast of what was stored
    p = parts.to_numpy(np.float32, (N // 64, M, 2))
    g64 = got.astype(np.float64).reshape(M, N // 64, 64)
    assert np.abs(p[:, :, 0].T - g64.sum(2)).max() <= 1e-4
    assert np.abs(p[:, :, 1].T - (g64 ** 2).sum(2)).max() <= 1e-4 * (g64 ** 2).sum(2).max()


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7])
@pytest.mark.parametrize("dt", DT)
def test_gemm_resid_split_epilogue(dt, variant):
    """The residual stream as two planes, x = hi + lo (VH_EPI_RESID_SPLIT): hi = T(x), 16 bit, the next GEMM's operand; lo = what
    that rounding dropped, one scaled e4m3 byte.  (hi, lo) += A W^T + bias, checked against the oracle's fp32 update of the SAME
    starting value hi0 + lo0: the new hi is the 16-bit rounding of the new x, hi + lo reproduces it to the planes' joint
    precision (lo's own rounding: 2^-4 of half an ulp of hi, i.e. 12 / 15 significant bits), the per-64-column row sums are
    those of the new x (taken before lo is rounded), rows outside M are untouched (ragged M, canary rows)."""
    M, N, K = 700, 512, 192
    a = rnd16(S.fill(M * K, 29, 1, 0).reshape(M, K), dt)
    w = rnd16(S.fill(N * K, 29, 2, 1, 0.1).reshape(N, K), dt)
    bias = S.fill(N, 29, 3, 1, 0.1)
    x0 = (S.fill((M + 2) * N, 29, 4, 0) * 3.0).reshape(M + 2, N)
    hi0 = rnd16(x0, dt)
    lo0_8 = vithip.to_lo8(x0 - hi0, dt)
    lo0 = vithip.from_lo8(lo0_8, dt)
    ref = (hi0 + lo0)[1:-1] + O.linear(a, w, bias)
    hi, lo = dev(vithip.to16(hi0, dt)), dev(lo0_8)
    parts = vithip.DeviceBuffer((N // 64) * M * 8)
    vithip.op_gemm_ex(dev(vithip.to16(a, dt)).ptr, dev(vithip.to16(w, dt)).ptr, dev(bias).ptr, hi.ptr + N * 2, M, N, K,
                      vithip.EPI_RESID_SPLIT, dt, out16_ptr=lo.ptr + N, partials_ptr=parts.ptr, variant=variant)
    h = vithip.from16(hi.to_numpy(np.uint16, (M + 2, N)), dt)
    l = vithip.from_lo8(lo.to_numpy(np.uint8, (M + 2, N)), dt)
    for plane, before in ((h, hi0), (l, lo0)):
        assert np.array_equal(plane[0], before[0]) and np.array_equal(plane[-1], before[-1])   # canary rows
    got = h[1:-1].astype(np.float64) + l[1:-1]
    scale = np.abs(ref).max()
    # lo is a 4-bit-significand rounding of a residue of at most half an ulp of hi (+ the byte's underflow step)
    assert (np.abs(got - ref) <= ULP[dt] * np.abs(ref) * 2.0 ** -4 * 1.05 + 2.0 ** -10 / vithip.LO8_SCALE[dt] + 2e-5 * scale).all()
    # hi is the 16-bit rounding of the updated value (up to the last-bit effect of the GEMM's summation order)
    assert (np.abs(h[1:-1] - ref) <= ULP[dt] * np.abs(ref) * 1.01 + 2e-5 * scale).all()
    p = parts.to_numpy(np.float32, (N // 64, M, 2))
    g64 = ref.astype(np.float64).reshape(M, N // 64, 64)      # the statistics are those of the updated x BEFORE its planes are rounded
    assert np.abs(p[:, :, 0].T - g64.sum(2)).max() <= 2e-4 * scale
    assert np.abs(p[:, :, 1].T - (g64 ** 2).sum(2)).max() <= 1e-4 * (g64 ** 2).sum(2).max()


@pytest.mark.parametrize("variant", [1, 2, 5, 6])
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,NP", [(3, 196), (37, 16), (5, 100)])
def test_gemm_patch_split_epilogue(dt, variant, B, NP):
    """VH_EPI_PATCH_SPLIT: the patch embedding written directly as the split residual.  Row m = image * NP + p of the GEMM
    lands on token row image * (NP + 1) + 1 + p as hi = T(x), lo = lo8(x - hi) of x = A W^T + bias + pos[1 + p], with that
    row's per-64-column (sum, sum of squares); class-token rows stay untouched (canary).  Patch counts above and below the
    128-row wave tile (one or several image boundaries inside it), ragged M."""
    N, K = 256, 192
    T = NP + 1
    M = B * NP
    a = rnd16(S.fill(M * K, 41, 1, 0).reshape(M, K), dt)
    w = rnd16(S.fill(N * K, 41, 2, 1, 0.1).reshape(N, K), dt)
    bias = S.fill(N, 41, 3, 1, 0.1)
    pos = S.fill(T * N, 41, 4, 1, 0.5).reshape(T, N)
    ref = O.linear(a, w, bias).reshape(B, NP, N) + pos[None, 1:, :]
    canary = np.full((B * T, N), 0x3c00 if dt == vithip.DTYPE_FP16 else 0x3f80, dtype=np.uint16)      # 1.0 everywhere
    canary8 = np.full((B * T, N), 0x38, dtype=np.uint8)                                                # e4m3 1.0
    hi, lo = dev(canary), dev(canary8)
    parts = vithip.DeviceBuffer((N // 64) * B * T * 8)
    pz = np.full((N // 64, B * T, 2), -7.0, dtype=np.float32)
    vithip.lib().vh_memcpy_h2d(0, parts.ptr, pz.ctypes.data, pz.nbytes)
    vithip.op_gemm_ex(dev(vithip.to16(a, dt)).ptr, dev(vithip.to16(w, dt)).ptr, dev(bias).ptr, hi.ptr, M, N, K,
                      vithip.EPI_PATCH_SPLIT, dt, aux_ptr=dev(pos).ptr, aux_i=NP, out16_ptr=lo.ptr, partials_ptr=parts.ptr,
                      variant=variant)
    h16 = hi.to_numpy(np.uint16, (B, T, N))
    l8 = lo.to_numpy(np.uint8, (B, T, N))
    assert (h16[:, 0] == canary[0, 0]).all() and (l8[:, 0] == 0x38).all()                   # class-token rows untouched
    h, l = vithip.from16(h16, dt)[:, 1:], vithip.from_lo8(l8, dt)[:, 1:]
    got = h.astype(np.float64) + l
    scale = np.abs(ref).max()
    assert (np.abs(got - ref) <= ULP[dt] * np.abs(ref) * 2.0 ** -4 * 1.05 + 2.0 ** -10 / vithip.LO8_SCALE[dt] + 2e-5 * scale).all()
    assert (np.abs(h - ref) <= ULP[dt] * np.abs(ref) * 1.01 + 2e-5 * scale).all()
    assert (np.abs(l) <= ULP[dt] * np.abs(h) * 1.07 + 1e-30).all()                             # lo is a rounding residue: at most half an ulp of hi (+ its own rounding)
    p = parts.to_numpy(np.float32, (N // 64, B, T, 2))
    assert (p[:, :, 0] == -7.0).all()                                                           # no statistics for class-token rows
    g64 = ref.astype(np.float64).reshape(B, NP, N // 64, 64)   # statistics of x before its planes are rounded
    assert np.abs(np.moveaxis(p[:, :, 1:, 0], 0, 2) - g64.sum(3)).max() <= 2e-4 * scale
    assert np.abs(np.moveaxis(p[:, :, 1:, 1], 0, 2) - (g64 ** 2).sum(3)).max() <= 1e-4 * (g64 ** 2).sum(3).max()


def test_rowstats_split_planes_and_statistics():
    for dt in DT:
        rows, dim = 333, 768
        x = (S.fill(rows * dim, 31, 5, 0) * 2.0 + 0.3).reshape(rows, dim)
        hi, lo, st = vithip.DeviceBuffer(rows * dim * 2), vithip.DeviceBuffer(rows * dim), vithip.DeviceBuffer(rows * 8)
        vithip.op_rowstats_split(dev(x).ptr, rows, dim, 1e-6, hi.ptr, lo.ptr, st.ptr, dt)
        h16 = hi.to_numpy(np.uint16, (rows, dim))
        assert np.array_equal(h16, vithip.to16(x, dt))
        h = vithip.from16(h16, dt)
        assert np.array_equal(lo.to_numpy(np.uint8, (rows, dim)), vithip.to_lo8(x - h, dt))
        s = st.to_numpy(np.float32, (rows, 2))
        x64 = x.astype(np.float64)
        assert np.abs(s[:, 0] - x64.mean(1)).max() <= 1e-6
        assert np.abs(s[:, 1] - 1.0 / np.sqrt(x64.var(1) + 1e-6)).max() <= 1e-5 * s[:, 1].max()
        for b in (hi, lo, st):
            b.free()


@pytest.mark.parametrize("dt", [vithip.DTYPE_BF16, vithip.DTYPE_FP16])
@pytest.mark.parametrize("epi", ["bias", "gelu", "lnfold_gelu", "resid_ln", "resid_split"])
def test_gemm_persistent_walks_several_tiles_per_workgroup(dt, epi):
    """The persistent ping-pong form (variant 6, the default for 16-bit results): with more tiles than CUs a workgroup
    runs 2-3 tiles back to back, prefetching the next tile's first K-tile from inside the epilogue.  Same arithmetic in
    the same order as the one-tile-per-workgroup form (variant 5), so the results must be bit-identical to it; the
    oracle pins both.  Ragged M, K = 3 K-tiles / 1 K-tile / 2 K-tiles / 5 K-tiles (from four K-tiles on, the LN-fold epilogues of
    the persistent form get their constants -- row statistics, d, c -- through LDS-DMA during the K loop instead of loading them)."""
    rng = np.random.default_rng(7)
    for M, N, K in ((256 * 141 + 37, 1024, 192), (256 * 70 + 5, 2048, 64), (256 * 67, 1280, 128), (256 * 72, 768, 320)):
        a = (rng.random((M, K), dtype=np.float32) * 2 - 1)
        w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        b = (rng.standard_normal(N) * 0.1).astype(np.float32)
        a16, w16 = vithip.to16(a, dt), vithip.to16(w, dt)
        A, W, Bv = dev(a16), dev(w16), dev(b)
        outs = []
        st = np.stack([rng.standard_normal(M) * 0.1, 1.0 + rng.random(M)], axis=1).astype(np.float32)
        cvec = (rng.standard_normal(N) * 0.1).astype(np.float32)
        x0 = (rng.random((M, N), dtype=np.float32) - 0.5) if epi in ("resid_ln", "resid_split") else None
        for variant in (5, 6):
            out = vithip.DeviceBuffer(M * N * 2)
            if epi in ("resid_ln", "resid_split"):
                # residual read-modify-write (fp32 + 16-bit copy, or the 16-bit + one-byte planes) + per-64-column row sums: every
                # output must be identical
                split = epi == "resid_split"
                xb = vithip.DeviceBuffer.from_numpy(vithip.to16(x0, dt) if split else x0)
                o16 = vithip.DeviceBuffer.from_numpy(vithip.to_lo8(x0 - vithip.from16(vithip.to16(x0, dt), dt), dt)) if split \
                    else vithip.DeviceBuffer(M * N * 2)
                parts = vithip.DeviceBuffer((N // 64) * M * 2 * 4)
                vithip.op_gemm_ex(A.ptr, W.ptr, Bv.ptr, xb.ptr, M, N, K, vithip.EPI_RESID_SPLIT if split else vithip.EPI_RESID_LN, dt,
                                  out16_ptr=o16.ptr, partials_ptr=parts.ptr, variant=variant)
                outs.append(np.concatenate([xb.to_numpy(np.uint16, (M * N * (1 if split else 2),)), o16.to_numpy(np.uint16, (M * N // (2 if split else 1),)),
                                            parts.to_numpy(np.uint16, ((N // 64) * M * 4,))]))
                xb.free(); o16.free(); parts.free(); out.free()
                continue
            if epi == "lnfold_gelu":
                vithip.op_gemm_ex(A.ptr, W.ptr, Bv.ptr, out.ptr, M, N, K, vithip.EPI_LNFOLD_GELU, dt,
                                  aux_ptr=dev(cvec).ptr, stats_ptr=dev(st).ptr, variant=variant)
            else:
                vithip.op_gemm(A.ptr, W.ptr, Bv.ptr, out.ptr, M, N, K,
                               vithip.EPI_BIAS if epi == "bias" else vithip.EPI_BIAS_GELU, dt, variant=variant)
            outs.append(out.to_numpy(np.uint16, (M, N)).copy())
            out.free()
        assert np.array_equal(outs[0], outs[1]), f"variant 6 differs from variant 5 at {(outs[0] != outs[1]).sum()} of {M * N} elements ({M}x{N}x{K})"
        if epi == "bias":
            ref = vithip.from16(a16, dt).astype(np.float64) @ vithip.from16(w16, dt).astype(np.float64).T + b
            got = vithip.from16(outs[1], dt)
            assert np.abs(got - ref).max() <= (2e-2 if dt == vithip.DTYPE_BF16 else 3e-3) * max(1.0, np.abs(ref).max())
