"""MLP-mode training (init_gradient / launch_gradient).  The reference's bodies are commented-out code
(netFPGA.cpp:518-580), so PARITY IS UNPINNED: what is checked is (CPU) that the oracle's gradient is the gradient of the
loss it states -- against central finite differences of its own forward -- and that the loop does what include/vithip.h
says (error before the update, threshold ends the loop, later entries 0); (GPU) that the device follows the oracle."""
import numpy as np
import pytest

import oracle_lib as O

ACT_IDENTITY, ACT_RELU2, ACT_RELU, ACT_HARDTANH, ACT_GELU = range(5)


def net(seed, n_ins=6, n_p_l=(9, 7, 4), n_sets=5, scale=0.5):
    rng = np.random.default_rng(seed)
    fan, n_params = n_ins, 0
    for n in n_p_l:
        n_params += n * fan
        fan = n
    params = (rng.standard_normal(n_params) * scale / np.sqrt(n_ins)).astype(np.float32)
    bias = (rng.standard_normal(sum(n_p_l)) * 0.1).astype(np.float32)
    ins = rng.uniform(-1, 1, (n_sets, n_ins)).astype(np.float32)
    outs = rng.uniform(-0.5, 0.5, (n_sets, n_p_l[-1])).astype(np.float32)
    return n_ins, list(n_p_l), params, bias, ins, outs


def loss(n_ins, n_p_l, params, bias, act, ins, outs):
    """mean over the sets of 1/2 |a_L - t|^2, from the oracle's own forward, in double"""
    tot = 0.0
    for x, t in zip(ins, outs):
        y = O.mlp_forward(n_ins, n_p_l, params, bias, act, x).astype(np.float64)
        tot += 0.5 * float(((y - t) ** 2).sum())
    return tot / len(ins)


@pytest.mark.parametrize("act", [ACT_IDENTITY, ACT_GELU])
def test_oracle_update_is_the_gradient_of_the_stated_loss(act):
    # smooth activations only: the piecewise-linear ones have kinks a finite difference can straddle
    n_ins, n_p_l, params, bias, ins, outs = net(3)
    lr = 1.0
    p1, b1, err = O.mlp_train(n_ins, n_p_l, params, bias, act, ins, outs, 1, -1.0, lr)
    g_p, g_b = (params - p1) / lr, (bias - b1) / lr          # what one update subtracted = the mean gradient
    rng = np.random.default_rng(0)
    h = 1e-2
    for idx in rng.choice(params.size, 12, replace=False):
        pp, pm = params.copy(), params.copy()
        pp[idx] += h
        pm[idx] -= h
        fd = (loss(n_ins, n_p_l, pp, bias, act, ins, outs) - loss(n_ins, n_p_l, pm, bias, act, ins, outs)) / (2 * h)
        assert abs(fd - g_p[idx]) <= 2e-3 * max(1.0, abs(fd)), (idx, fd, g_p[idx])
    for idx in rng.choice(bias.size, 6, replace=False):
        bp, bm = bias.copy(), bias.copy()
        bp[idx] += h
        bm[idx] -= h
        fd = (loss(n_ins, n_p_l, params, bp, act, ins, outs) - loss(n_ins, n_p_l, params, bm, act, ins, outs)) / (2 * h)
        assert abs(fd - g_b[idx]) <= 2e-3 * max(1.0, abs(fd)), (idx, fd, g_b[idx])
    # the reported error of the iteration is the L1 output error BEFORE the update
    want = sum(np.abs(O.mlp_forward(n_ins, n_p_l, params, bias, act, x) - t).sum() for x, t in zip(ins, outs))
    assert abs(err[0] - want) <= 1e-4 * want


@pytest.mark.parametrize("act", [ACT_IDENTITY, ACT_RELU2, ACT_RELU, ACT_HARDTANH, ACT_GELU])
def test_oracle_training_reduces_the_error_and_honours_the_threshold(act):
    n_ins, n_p_l, params, bias, ins, outs = net(7)
    outs = np.clip(outs + 0.5, 0.05, 0.95)       # inside the range of the clamping activations
    _, _, err = O.mlp_train(n_ins, n_p_l, params, bias, act, ins, outs, 200, -1.0, 0.2)
    assert np.isfinite(err).all() and err[-1] < 0.6 * err[0], (act, err[0], err[-1])
    # a threshold ends the loop at the first iteration at or below it; later entries keep the reference's initial 0
    thr = float(err[50])
    _, _, e2 = O.mlp_train(n_ins, n_p_l, params, bias, act, ins, outs, 200, thr, 0.2)
    stop = int(np.argmax(e2 <= thr))
    assert e2[stop] <= thr and (e2[:stop] > thr).all() and (e2[stop + 1:] == 0).all()
    assert np.array_equal(e2[:stop + 1], err[:stop + 1])          # same trajectory up to the stop
    # zero iterations: nothing happens
    p0, b0, e0 = O.mlp_train(n_ins, n_p_l, params, bias, act, ins, outs, 0, -1.0, 0.2)
    assert e0.size == 0 and np.array_equal(p0, params) and np.array_equal(b0, bias)


@pytest.mark.gpu
@pytest.mark.parametrize("act", [ACT_IDENTITY, ACT_RELU2, ACT_RELU, ACT_HARDTANH, ACT_GELU])
def test_device_training_follows_the_oracle(act):
    vithip = pytest.importorskip("vithip")
    n_ins, n_p_l, params, bias, ins, outs = net(11, n_ins=70, n_p_l=(130, 33, 5), n_sets=9)
    outs = np.clip(outs + 0.5, 0.05, 0.95)
    iters, lr = 25, 0.1
    p_ref, b_ref, e_ref = O.mlp_train(n_ins, n_p_l, params, bias, act, ins, outs, iters, -1.0, lr)
    m = vithip.MlpContext(n_ins, n_p_l, activation=act)
    m.load_params(params, bias)
    with pytest.raises(vithip.VhError):
        m.launch_gradient(3, -1.0, lr)                # before init_gradient
    m.init_gradient(ins, outs)
    err = m.launch_gradient(iters, -1.0, lr)
    p, b = m.read_params()
    assert m.last_gradient_us() > 0
    # fp32 sums in different orders (wave-strided dot products on the device, sequential on the CPU), 25 dependent steps
    assert np.allclose(err, e_ref, rtol=2e-4, atol=1e-5), (act, np.abs(err - e_ref).max())
    assert np.allclose(p, p_ref, rtol=0, atol=2e-4 * np.abs(p_ref).max()), np.abs(p - p_ref).max()
    assert np.allclose(b, b_ref, rtol=0, atol=2e-4 * max(1e-3, np.abs(b_ref).max()))
    assert err[-1] < err[0]
    # the trained net is what the forward now runs
    y = m.forward(ins)
    want = np.stack([O.mlp_forward(n_ins, n_p_l, p, b, act, x) for x in ins])
    assert np.allclose(y, want, rtol=1e-4, atol=1e-5)
    # threshold: ends the loop, later entries 0; a second launch continues from the trained state
    thr = float(err[-1]) * 2.0
    e3 = m.launch_gradient(5, thr, lr)
    assert e3[0] <= thr and (e3[1:] == 0).all()
    m.close()
