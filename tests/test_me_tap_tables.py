"""CPU tier: the arithmetic of the refinement-stage kernel's two interpolation passes (vvenc_amd/csrc/me.hip: firstPass, predRow), restated in numpy on the tap tables the
library itself builds for a plan (vvhip_get_me_tap_tables_host: no device involved), against the oracle's two-pass sub-pel prediction (InterSearch.cpp:818-848 through
InterpolationFilter.cpp:285-441).

The kernel does not add the reference's constants where the reference adds them:
  * first pass: tap pairs scaled by 2^(8 - shift1), the value is bytes 1..2 of the 32-bit sum; stored as value + 2^(headRoom - 1) WITHOUT the reference's -8192;
  * second pass: taps scaled by 2^(16 - shift2), accumulators start at zero (the stored offset x the tap sum of 64 IS the reference's rounding term + 8192 << 6), clip on the
    scaled sum, the sample is its upper half; zero vertical phase = a plain shift of the stored value, or the same filter with the taps of phase 0.
This test pins what those rewrites rely on (every tap set sums to 64, the scaled pairs fit 16 bits, nothing overflows 32 bits) and that the results are the reference's."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TAP_SUPPORT = {0: (0, 7), 1: (1, 6), 2: (2, 5)}          # filter_mode -> (K0, K1): 8 taps, 6 taps, the 4-tap search set


def tables(bd):
    from vvenc_amd.lib import load_library
    L = load_library()
    out = np.zeros(6 * 192, np.int32)
    assert L.vvhip_get_me_tap_tables_host(bd, out.ctypes.data_as(C.c_void_p)) == 0
    return out.reshape(6, 192)


def support(mode, alt):
    return (1, 6) if (alt and mode == 2) else TAP_SUPPORT[mode]          # (the alternative half-sample filter has six taps: its table uses the 6-tap instance)


@pytest.mark.parametrize("bd", [8, 9, 10])
def test_tap_tables_scaling_and_sums(bd):
    head = max(14 - bd, 2)
    v_scale, h_scale = 1 << (10 - head), 1 << (2 + head)
    t = tables(bd)
    for mode in range(3):
        for alt in range(2):
            tab = t[mode * 2 + alt]
            v = tab[:128].reshape(16, 8)
            assert (v % v_scale == 0).all()
            taps = v // v_scale
            assert (taps.sum(1) == 64).all(), (mode, alt)                      # what folding the constants into the stored first-pass value relies on
            assert (np.abs(v) < 32768).all()                                   # v_mad_i32_i16 reads 16 bits of the tap
            assert list(taps[0]) == [0, 0, 0, 64, 0, 0, 0, 0]                  # phase 0 = the copy: a zero-phase lane may run the filter
            k0, k1 = support(mode, alt)
            assert (taps[:, :k0] == 0).all() and (taps[:, k1 + 1:] == 0).all()  # nothing outside the instance's tap support
            pairs = tab[128:].view(np.int16).reshape(16, 4, 2).astype(np.int64)
            npair = (k1 - k0 + 1) // 2
            for f in range(16):
                for i in range(npair):
                    assert pairs[f, i, 0] == taps[f, k0 + 2 * i] * h_scale and pairs[f, i, 1] == taps[f, k0 + 2 * i + 1] * h_scale, (mode, alt, f, i)
                assert (pairs[f, npair:] == 0).all()
            # 32-bit accumulators: first pass |taps| x scale x sample, second pass |taps| x scale x the 16-bit first-pass value
            assert np.abs(taps).sum(1).max() * h_scale * ((1 << bd) - 1) + (1 << (head + 7)) < 2 ** 31
            assert np.abs(taps).sum(1).max() * v_scale * 32767 < 2 ** 31


def kernel_prediction(tab, mode, alt, bd, ref, y0, x0, w, h, tx, ty, filter_zero_phase):
    """the stage kernel's arithmetic for one position: displacement (tx, ty) in 1/16 sample from the block at (y0, x0) of `ref`"""
    head = max(14 - bd, 2)
    k0, k1 = support(mode, alt)
    nt = k1 - k0 + 1
    v = tab[:128].reshape(16, 8).astype(np.int64)
    pairs = tab[128:].view(np.int16).reshape(16, 4, 2).astype(np.int64)
    sx, fx, sy, fy = tx >> 4, tx & 15, ty >> 4, ty & 15
    rows_t = h + nt
    # ---- first pass: tmp[r][x] <-> plane row y0 + k0 - 4 + r, column x0 + x + sx
    tmp = np.zeros((rows_t + 1, w), np.int64)                                   # (+ 1: the row a zero-phase lane's zero taps may touch)
    for r in range(rows_t):
        row = ref[y0 + k0 - 4 + r].astype(np.int64)
        for x in range(w):
            p = x0 + x + sx
            if fx:
                acc = (1 << (head - 1)) << 8
                for i in range(nt // 2):
                    acc += pairs[fx, i, 0] * row[p + k0 - 3 + 2 * i] + pairs[fx, i, 1] * row[p + k0 - 3 + 2 * i + 1]
                assert -2 ** 31 <= acc < 2 ** 31
                val = (acc >> 8) & 0xffff                                       # bytes 1..2 of the accumulator (v_perm_b32)
                tmp[r, x] = val - 65536 if val >= 32768 else val
            else:
                tmp[r, x] = (row[p] << head) + (1 << (head - 1))
            assert -32768 <= tmp[r, x] < 32768
    # ---- second pass
    maxv = (1 << bd) - 1
    out = np.zeros((h, w), np.int64)
    for y in range(h):
        for x in range(w):
            if fy or filter_zero_phase:
                acc = 0
                for t in range(nt):
                    acc += v[fy, k0 + t] * tmp[y + sy + t + 1, x]
                assert -2 ** 31 <= acc < 2 ** 31
                acc = min(max(acc, 0), (maxv << 16) | 0xffff)                  # v_med3_i32 on the scaled sum
                out[y, x] = acc >> 16
            else:
                out[y, x] = min(max(tmp[y + sy + 4 - k0, x] >> head, 0), maxv)
    return out.astype(np.int16)


@pytest.mark.parametrize("bd", [10, 8])
def test_two_pass_arithmetic_equals_the_reference(oracle, bd):
    rng = np.random.default_rng(5 + bd)
    top = 1 << bd
    H, W = 64, 96
    ref = rng.integers(0, top, (H, W), dtype=np.int16)
    ref[8:24, 8:40] = rng.choice([0, top - 1], (16, 32))                        # extreme edges: the filters over- and undershoot, both clips act
    t = tables(bd)
    checked = 0
    for mode in range(3):
        for alt in (0, 1):
            tab = t[mode * 2 + alt]
            for (w, h) in ((8, 8), (16, 4)):
                for i_frac in (2, 1):
                    for k in range(9):
                        rx, ry = (k % 3) - 1, (k // 3) - 1
                        bq = (0, 0) if i_frac == 2 else (int(rng.integers(-1, 2)) * 2, int(rng.integers(-1, 2)) * 2)
                        tx, ty = (rx + bq[0]) * i_frac * 4, (ry + bq[1]) * i_frac * 4
                        if not (-16 <= tx <= 16 and -16 <= ty <= 16):
                            continue
                        y0, x0 = 12 + int(rng.integers(0, 8)), 10 + int(rng.integers(0, 16))
                        exp = oracle.if_pred_luma_me((ref, y0 + (ty >> 4), x0 + (tx >> 4)), w, h, tx & 15, ty & 15, bd, bool(alt), mode)
                        for as_filter in ((False, True) if (ty & 15) == 0 else (False,)):
                            got = kernel_prediction(tab, mode, alt, bd, ref, y0, x0, w, h, tx, ty, as_filter)
                            assert np.array_equal(got, exp), (bd, mode, alt, w, h, tx, ty, as_filter)
                            checked += 1
    assert checked > 200
