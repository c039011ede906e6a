"""CPU suite: pins the oracle (oracle/*.c) before anything on the GPU is trusted.

  * against the committed golden vectors (tests/golden/*.npz: fp64 logits and residual-stream taps of
    an independent implementation, see tests/golden/make_golden.py) — the reference itself holds no
    fixtures, so this is what stands between the oracle and "parity unpinned" for the ViT path;
  * its operators against plain numpy float64 restatements;
  * the three statements of the synthetic-data generator against each other (numpy vs C);
  * the MLP-mode restatement against a numpy statement of the reference's layout
    (netFPGA.cpp:68-76, 91-106).
"""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
import vh_synth as S

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_golden_fixtures_present():
    names = {os.path.basename(p).split("_s")[0] for p in GOLDEN}
    assert {"vit_micro", "vit_mini", "vit_tiny", "vit_base"} <= names


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_golden(path):
    g = np.load(path)
    cfg = S.CONFIGS[os.path.basename(path).split("_s")[0]]
    wseed, iseed, batch = [int(v) for v in g["meta"]]
    tensors = S.make_tensors(cfg, wseed)
    # the generator has not drifted since the fixture was written
    cs = np.array([float(v.astype(np.float64).sum()) for v in tensors.values()][:8])
    assert np.allclose(cs, g["weights_checksum"], rtol=0, atol=1e-9)
    images = S.make_images(cfg, iseed, batch)
    assert abs(float(images.astype(np.float64).sum()) - float(g["images_checksum"][0])) < 1e-9
    blob = S.pack_blob(cfg, tensors)
    logits, hidden = O.vit_forward(cfg, blob, images, want_hidden=True)
    assert rel(logits, g["logits_f64"]) <= 5e-6
    assert rel(hidden[:64], g["hidden_last_f64"]) <= 5e-6
    # the fp32 run of the independent implementation agrees with its own fp64 run to the same order,
    # i.e. the oracle is as close to exact as an fp32 implementation gets
    assert rel(g["logits_f32"], g["logits_f64"]) <= 5e-6
    # taps: embedding output and the stream after layer 1
    _, emb = O.vit_forward(cfg, blob, images, n_layers=0, want_hidden=True)
    assert rel(emb[:64], g["embed_f64"]) <= 5e-6
    _, h1 = O.vit_forward(cfg, blob, images, n_layers=1, want_hidden=True)
    assert rel(h1[:64], g["hidden_l1_f64"]) <= 5e-6


def test_generator_numpy_equals_c():
    for kind, sigma, off, tid in ((0, 0.0, 0.0, 0x100), (1, 0.02, 0.0, 5), (1, 0.05, 1.0, 0x7000), (2, 0.0, 3.5, 1)):
        a = S.fill(70001, 99, tid, kind, sigma, off)
        b = O.fill(70001, 99, tid, kind, sigma, off)
        assert np.array_equal(a, b)
    u = S.fill(1 << 18, 1, 0x100, 0)
    assert u.min() >= -1.0 and u.max() < 1.0 and abs(float(u.mean())) < 5e-3  # value range of defines.h:11-12
    n = S.fill(1 << 18, 1, 7, 1, 0.02)
    assert abs(float(n.std()) - 0.02) < 2e-4 and abs(float(n.mean())) < 2e-4 and np.abs(n).max() <= 0.02 * 3.4642


@pytest.mark.parametrize("name", ["vit_micro", "vit_mini", "vit_tiny"])
def test_blob_layout_numpy_equals_c(name):
    cfg = S.CONFIGS[name]
    a, b = S.make_blob(cfg, 4), O.make_blob(cfg, 4)
    assert a.nbytes == 64 + 4 * S.param_count(cfg)
    assert np.array_equal(a, b)


def test_flop_and_param_counts_match_survey():
    assert S.flops_per_image(S.CONFIGS["vit_base"]) == 35_127_656_448
    assert S.flops_per_image(S.CONFIGS["vit_tiny"]) == 2_507_366_400
    assert S.flops_per_image(S.CONFIGS["vit_large_384"]) == 382_132_600_832
    assert S.param_count(S.CONFIGS["vit_base"]) == 86_567_656
    assert S.param_count(S.CONFIGS["vit_tiny"]) == 5_717_416


def test_oracle_operators_against_numpy_float64():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((37, 96)).astype(np.float32)
    w = rng.standard_normal((50, 96)).astype(np.float32)
    b = rng.standard_normal(50).astype(np.float32)
    assert rel(O.linear(a, w, b), a.astype(np.float64) @ w.astype(np.float64).T + b) <= 2e-6
    assert rel(O.linear(a[:1], w[:3], None), a[:1].astype(np.float64) @ w[:3].astype(np.float64).T) <= 2e-6
    x = rng.standard_normal((9, 64)).astype(np.float32) * 3 + 1
    g, be = rng.standard_normal(64).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    x64 = x.astype(np.float64)
    ln = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-6) * g + be
    assert rel(O.layernorm(x, g, be, 1e-6), ln) <= 2e-6
    from math import erf
    v = np.linspace(-6, 6, 1001).astype(np.float32)
    assert np.abs(O.gelu(v) - np.array([0.5 * t * (1 + erf(t / np.sqrt(2))) for t in v.astype(np.float64)])).max() <= 1e-6
    B, T, H = 2, 13, 3
    qkv = rng.standard_normal((B * T, 3 * H * 64)).astype(np.float32)
    q, k, vv = [qkv[:, i * H * 64:(i + 1) * H * 64].astype(np.float64).reshape(B, T, H, 64).transpose(0, 2, 1, 3) for i in range(3)]
    s = q @ k.transpose(0, 1, 3, 2) / 8.0
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = (p @ vv).transpose(0, 2, 1, 3).reshape(B * T, H * 64)
    assert rel(O.attention(qkv, B, T, H), ref) <= 2e-6
    img = rng.standard_normal((2, 32, 32, 3)).astype(np.float32)
    col = O.im2col(img, 16)
    assert col.shape == (8, 768)
    assert np.array_equal(col[5].reshape(16, 16, 3), img[1, 0:16, 16:32, :])  # image 1, patch (0,1)
    r = np.array([1.0, 1.00390625, 1.01171875, -1.00390625, 3.0e-39], dtype=np.float32)
    assert np.array_equal(O.round_bf16(r)[:4], np.array([1.0, 1.0, 1.015625, -1.0], dtype=np.float32))
    assert np.array_equal(O.round_fp16(r)[:2], np.array([1.0, 1.00390625], dtype=np.float32))


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_mlp_oracle_against_numpy_statement_of_the_reference_layout(act):
    n_ins, npl = 7, [5, 9, 3]
    n_params = 7 * 5 + 5 * 9 + 9 * 3                       # netFPGA.cpp:68-76
    params, bias = O.mlp_random_params(n_params, sum(npl), seed=1)
    assert set(np.round(params * 100).astype(int)) <= set(range(-100, 100))  # netFPGA.cpp:82-88 value set
    x = S.fill(n_ins, 2, 1, 0)
    a = x.astype(np.float64)
    po = bo = 0
    fan = n_ins
    for n in npl:                                          # layer-major, neuron-major, input-minor
        W = params[po:po + n * fan].astype(np.float64).reshape(n, fan)
        z = W @ a + bias[bo:bo + n]
        from math import erf
        a = {0: z, 1: np.clip(z, 0, 1), 2: np.maximum(z, 0), 3: np.clip(z, -1, 1),
             4: np.array([0.5 * t * (1 + erf(t / np.sqrt(2))) for t in z])}[act]
        a = a.astype(np.float32).astype(np.float64)
        po, bo, fan = po + n * fan, bo + n, n
    got = O.mlp_forward(n_ins, npl, params, bias, act, x)
    assert np.abs(got - a).max() <= 1e-6
