"""filter_image pipeline.  The reference's `image_process` kernel is absent, so the arithmetic
(3x3 binomial blur / Sobel, replicated borders) is this build's definition — PARITY UNPINNED by the reference; what
is reproduced from the reference is the ring behaviour (netFPGA.cpp:292-365).  The oracle is checked on the host
against hand-computable cases; the HIP kernel is bit-exact against the oracle."""
import numpy as np
import pytest

import oracle_lib as O

vithip = pytest.importorskip("vithip")


def frame(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w), dtype=np.uint8)


def test_oracle_filter_known_answers():
    flat = np.full((9, 11), 77, np.uint8)
    assert np.array_equal(O.filter3x3(flat, 0), flat) and not O.filter3x3(flat, 1).any()
    imp = np.zeros((7, 7), np.uint8); imp[3, 3] = 160
    assert np.array_equal(O.filter3x3(imp, 0)[2:5, 2:5], np.array([[10, 20, 10], [20, 40, 20], [10, 20, 10]]))
    edge = np.zeros((6, 8), np.uint8); edge[:, 4:] = 50              # vertical step: |gx| = 4 * 50 on both sides of it
    s = O.filter3x3(edge, 1)
    assert np.all(s[:, 3:5] == 200) and not s[:, :3].any() and not s[:, 5:].any()
    big = np.zeros((4, 4), np.uint8); big[:, 2:] = 255
    assert O.filter3x3(big, 1).max() == 255                          # saturates
    # against a straightforward numpy statement with edge padding
    f = frame(23, 31, 1).astype(np.int32)
    p = np.pad(f, 1, mode="edge")
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]])
    want = sum(k[r, c] * p[r:r + 23, c:c + 31] for r in range(3) for c in range(3))
    assert np.array_equal(O.filter3x3(f.astype(np.uint8), 0), ((want + 8) >> 4).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [vithip.FILTER_BLUR3, vithip.FILTER_SOBEL3])
@pytest.mark.parametrize("h,w", [(1080, 1920), (37, 53), (5, 7), (1, 1), (64, 4), (3, 1030), (2, 8)])
def test_filter_is_bit_exact(kind, h, w):
    f = frame(h, w, h * 7 + w)
    pipe = vithip.FilterPipeline(h, w, slots=2, kind=kind)
    pipe.submit(f)
    got = pipe.collect()
    pipe.close()
    assert np.array_equal(got, O.filter3x3(f, kind))


@pytest.mark.gpu
def test_ring_is_fifo_with_overflow_and_underflow_reports():
    h, w, slots = 120, 200, 24                                        # the reference's BATCH_SIZE
    pipe = vithip.FilterPipeline(h, w, slots=slots)
    with pytest.raises(vithip.VhError) as e:
        pipe.collect()
    assert e.value.code == 7                                          # "PILA VACIA"
    frames = [frame(h, w, i) for i in range(slots + 6)]
    for f in frames[:slots]:
        pipe.submit(f)
    assert pipe.free_slots() == 0
    with pytest.raises(vithip.VhError) as e:
        pipe.submit(frames[slots])
    assert e.value.code == 6                                          # "PILA LLENA": the frame is not queued
    out = [pipe.collect() for _ in range(3)]
    for f in frames[slots:slots + 3]:                                 # wrap around
        pipe.submit(f)
    while pipe.free_slots() < slots:
        out.append(pipe.collect())
    assert len(out) == slots + 3
    for got, f in zip(out, frames[:slots + 3]):
        assert np.array_equal(got, O.filter3x3(f, 0))
    pipe.close()
    with pytest.raises(vithip.VhError):
        vithip.FilterPipeline(0, 10)
