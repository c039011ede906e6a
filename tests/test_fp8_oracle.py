"""CPU checks of the e4m3 quantiser the fp8 path (BASELINE config 5, VH_DTYPE_FP8) is judged against:
oracle/vit_oracle.c's routines against an independent table-driven numpy statement of OCP e4m3fn."""
import numpy as np

import oracle_lib as O
import vh_synth as S


def e4m3_values():
    b = np.arange(256)
    e, m = (b >> 3) & 15, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (1 + m / 8.0) * 2.0 ** (e - 7.0))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return np.where(b >= 128, -v, v)


def test_decode_table_and_exact_round_trip():
    tab = e4m3_values()
    for b in range(256):
        got = O.lib().oracle_e4m3_to_float(b)
        if np.isnan(tab[b]):
            assert np.isnan(got)
            continue
        assert got == np.float32(tab[b])
        if tab[b] != 0:                                   # +-0 both decode to 0
            assert O.lib().oracle_e4m3_from_float(float(tab[b])) == b
    assert np.nanmax(tab) == 448.0


def test_round_to_nearest_even_and_saturation():
    tab = e4m3_values()
    pos = np.sort(tab[(tab >= 0) & ~np.isnan(tab)])
    pos = np.unique(pos)
    f = O.lib().oracle_e4m3_from_float
    dec = O.lib().oracle_e4m3_to_float
    for lo, hi in zip(pos[:-1], pos[1:]):
        mid = np.float32(0.5 * (lo + hi))
        below, above = np.nextafter(mid, np.float32(0)), np.nextafter(mid, np.float32(1e9))
        assert dec(f(float(below))) == np.float32(lo) and dec(f(float(above))) == np.float32(hi)
        tie = dec(f(float(mid)))                          # ties go to the even mantissa
        assert tie in (np.float32(lo), np.float32(hi))
        assert (f(float(mid)) & 1) == 0
        assert dec(f(float(-mid))) == -tie
    for big in (448.0, 449.0, 464.0, 465.0, 1e30, np.inf):
        assert dec(f(big)) == 448.0 and dec(f(-big)) == -448.0
    assert f(float("nan")) == 0x7F


def test_quantised_values_are_nearest_representable():
    x = (S.fill(200000, 3, 5, 0) * 8.0).astype(np.float32)
    q = O.quant_e4m3(x)
    vals = np.unique(e4m3_values()[~np.isnan(e4m3_values())]).astype(np.float64)
    idx = np.searchsorted(vals, x.astype(np.float64))
    idx = np.clip(idx, 1, len(vals) - 1)
    lo, hi = vals[idx - 1], vals[idx]
    nearest = np.minimum(np.abs(lo - x), np.abs(hi - x))
    assert np.all(np.abs(q.astype(np.float64) - x) <= nearest + 1e-12)
    rel = np.abs(q - x) / np.maximum(np.abs(x), 2.0 ** -6)
    assert rel.max() <= 2.0 ** -4 + 1e-6                  # half an ulp of a 3-bit mantissa


def test_row_quantiser_scales_and_fp8_forward_is_close_to_fp32_forward():
    w = (S.fill(16 * 256, 9, 2, 1, 0.02)).reshape(16, 256)
    w[3] = 0.0
    w8, wq, sc = O.quantize_rows(w, post=0.125)
    amax = np.abs(w).max(1)
    assert np.allclose(sc[amax > 0], amax[amax > 0] / 448.0 * 0.125, rtol=1e-6) and sc[3] == np.float32(0.125)
    assert np.abs(wq).max() == 448.0 and np.all(wq[3] == 0)
    deq = wq * (sc / 0.125)[:, None]
    assert np.abs(deq - w).max() <= 2.0 ** -4 * np.abs(w).max()
    # the emulated fp8 forward stays within fp8's noise of the fp32 forward (and is not identical to it)
    cfg = S.CONFIGS["vit_q8"]
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, 2)
    ref = O.vit_forward(cfg, blob, images)
    q = O.vit_forward(cfg, blob, images, fp8=True)
    err = np.abs(q - ref).max() / np.abs(ref).max()
    assert 1e-4 < err < 0.15, err


def test_folded_fp8_emulation_is_the_same_model_with_another_rounding_pattern():
    # oracle_vit_forward_fp8_folded: the LayerNorm folded into q|k|v and fc1 (e4m3 copy of the RAW rows as the operand).
    # Without e4m3 rounding the two data flows are the same function; with it they differ by fp8 noise only: the folded
    # emulation is as close to fp32 as the plain one (within a factor 1.5), and not identical to it.
    cfg = S.CONFIGS["vit_q8"]
    blob, images = S.make_blob(cfg, 0), S.make_images(cfg, 1, 2)
    ref = O.vit_forward(cfg, blob, images)
    plain = O.vit_forward(cfg, blob, images, fp8=True)
    folded = O.vit_forward(cfg, blob, images, fp8="folded")
    rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    r_plain, r_folded = rms(plain, ref), rms(folded, ref)
    assert not np.array_equal(plain, folded)
    assert 1e-4 < r_folded < 1.5 * r_plain + 1e-3, (r_plain, r_folded)
