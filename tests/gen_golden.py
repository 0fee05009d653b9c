#!/usr/bin/env python3
"""Writes tests/golden/*.npz: seeded inputs + the outputs of THE REFERENCE ITSELF
(oracle/_ref/libvvenc_ref.so, compiled from /root/reference by oracle/ref/Makefile; x86-SIMD row,
which the reference's own unit tests hold equal to its scalar row).

Run in the build container (needs /root/reference):   python tests/gen_golden.py
The fixtures are committed; tests/test_oracle_golden.py and the -m gpu parity tests replay them on
machines where /root/reference does not exist.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import DCT2, DCT8, DST7, RefLib, build_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rand_plane(rng, h, w, bits=10):
    return rng.integers(0, 1 << bits, size=(h, w), dtype=np.int16)


def synth_pair(rng, h, w, shift=(3, 1), noise=6):
    yy, xx = np.mgrid[0:h + 32, 0:w + 32]
    base = 512 + 180 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 120 * np.sin((xx + yy) / 11.0) + 60 * np.sin(xx / 3.1) * np.sin(yy / 4.3)
    base = base + rng.normal(0, 12, base.shape)
    a = np.clip(base[16:16 + h, 16:16 + w], 0, 1023)
    b = np.clip(base[16 + shift[1]:16 + shift[1] + h, 16 + shift[0]:16 + shift[0] + w] + rng.normal(0, noise, (h, w)), 0, 1023)
    return a.astype(np.int16), b.astype(np.int16)


def gen_distortion_ext(R, R0):
    """GEO masked SAD (DF_SAD_WITH_MASK) and fixed-weight SSE (m_fxdWtdPredPtr) -> distortion_ext.npz"""
    rng = np.random.default_rng(20260924)
    org, cur = rand_plane(rng, 96, 160), rand_plane(rng, 96, 160)
    yy, xx = np.mgrid[0:200, 0:300]
    mask = np.clip((xx - yy) // 3 + 4 + rng.integers(-1, 2, (200, 300)), 0, 8).astype(np.int16)
    mcases, mout = [], []
    for w in (8, 16, 32, 64):
        for h in (8, 16, 32, 64):
            for ss in (0, 1):
                for step_x in (1, -1):
                    ox, oy, cx, cy = (int(rng.integers(0, 160 - w)), int(rng.integers(0, 96 - h)), int(rng.integers(0, 160 - w)), int(rng.integers(0, 96 - h)))
                    mx, my = int(rng.integers(70, 150)), int(rng.integers(0, 60))
                    v = R.sad_mask((org, oy, ox), (cur, cy, cx), (mask, my, mx), step_x, -step_x * w, w, h, ss)
                    assert v == R0.sad_mask((org, oy, ox), (cur, cy, cx), (mask, my, mx), step_x, -step_x * w, w, h, ss)
                    mcases.append((ox, oy, cx, cy, mx, my, step_x, -step_x * w, w, h, ss))
                    mout.append(v)
    wcases, wout = [], []
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            ox, oy, cx, cy = (int(rng.integers(0, 160 - w)), int(rng.integers(0, 96 - h)), int(rng.integers(0, 160 - w)), int(rng.integers(0, 96 - h)))
            wt = int(rng.integers(1, 1 << 17))
            v = R.fix_weighted_sse((org, oy, ox), (cur, cy, cx), w, h, wt)
            assert v == R0.fix_weighted_sse((org, oy, ox), (cur, cy, cx), w, h, wt)
            wcases.append((ox, oy, cx, cy, w, h, wt))
            wout.append(v)
    np.savez_compressed(os.path.join(OUT, "distortion_ext.npz"), org=org, cur=cur, mask=mask,
                        mask_cases=np.array(mcases, np.int32), mask_out=np.array(mout, np.uint64),
                        wsse_cases=np.array(wcases, np.int64), wsse_out=np.array(wout, np.uint64))


def gen_interp(R, R0):
    """interpolation filter slots, luma dispatch, prediction blocks -> interp.npz"""
    rng = np.random.default_rng(20260925)
    plane = rng.integers(0, 1024, size=(112, 176)).astype(np.int16)
    plane[30:90, 40:140] = np.clip(512 + 200 * np.sin(np.mgrid[0:60, 0:100][1] / 6.0) + rng.normal(0, 20, (60, 100)), 0, 1023).astype(np.int16)
    inter = (rng.integers(0, 1 << 14, size=(112, 176)) - 8192).astype(np.int16)
    d = dict(plane=plane, inter=inter)
    slot_cases, slot_out = [], []
    for n, set_, phases in ((8, 0, (1, 8, 13)), (6, 1, (2, 8)), (6, 3, (0,)), (4, 2, (3, 16, 29))):
        for p in phases:
            _, c = R.if_coeff(set_, p)
            for (w, h) in ((4, 4), (8, 8), (16, 4), (32, 16), (64, 24)):
                for vertical in (0, 1):
                    for first in (0, 1):
                        for last in (0, 1):
                            src = (plane if first else inter, 24, 32)
                            o = R.if_filter(n, vertical, first, last, 10, src, w, h, c)
                            assert np.array_equal(o, R0.if_filter(n, vertical, first, last, 10, src, w, h, c))
                            slot_cases.append((n, set_, p, w, h, vertical, first, last))
                            slot_out.append(o.ravel())
    copy_cases, copy_out = [], []
    for (w, h) in ((4, 4), (8, 16), (64, 8)):
        for first, last, bi in ((1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 0, 1)):
            src = (plane if first else inter, 24, 32)
            o = R.if_copy(first, last, 10, src, w, h, bool(bi))
            assert np.array_equal(o, R0.if_copy(first, last, 10, src, w, h, bool(bi)))
            copy_cases.append((w, h, first, last, bi)); copy_out.append(o.ravel())
    pred_cases, pred_out = [], []
    for bd in (8, 10):
        pl = (plane >> (10 - bd)).astype(np.int16)
        d["plane%d" % bd] = pl
        for (w, h) in ((4, 4), (8, 4), (8, 8), (16, 8), (16, 16), (32, 32), (64, 16), (128, 64), (4, 8), (24, 8)):
            for (xf, yf) in ((0, 0), (8, 0), (0, 8), (8, 8), (3, 0), (0, 13), (5, 11), (15, 1), (4, 12)):
                for alt in (0, 1):
                    if alt and (xf % 8 or yf % 8):
                        continue
                    for rnd in (1, 0):
                        o = R.if_pred_luma((pl, 24, 20), w, h, xf, yf, bool(rnd), bd, bool(alt))
                        assert np.array_equal(o, R0.if_pred_luma((pl, 24, 20), w, h, xf, yf, bool(rnd), bd, bool(alt)))
                        pred_cases.append((bd, w, h, xf, yf, alt, rnd)); pred_out.append(o.ravel())
    me_cases, me_out = [], []
    for (w, h) in ((8, 8), (16, 16), (32, 16), (64, 64)):
        for (xf, yf) in ((8, 0), (0, 8), (8, 8), (4, 0), (12, 4), (4, 12), (0, 4), (8, 12)):
            for alt in (0, 1):
                for rt in (0, 1, 2):
                    o = R.if_pred_luma_me((plane, 24, 20), w, h, xf, yf, 10, bool(alt), rt)
                    assert np.array_equal(o, R0.if_pred_luma_me((plane, 24, 20), w, h, xf, yf, 10, bool(alt), rt))
                    me_cases.append((w, h, xf, yf, alt, rt)); me_out.append(o.ravel())
    d.update(slot_cases=np.array(slot_cases, np.int32), slot_out=np.concatenate(slot_out), copy_cases=np.array(copy_cases, np.int32), copy_out=np.concatenate(copy_out),
             pred_cases=np.array(pred_cases, np.int32), pred_out=np.concatenate(pred_out), me_cases=np.array(me_cases, np.int32), me_out=np.concatenate(me_out))
    np.savez_compressed(os.path.join(OUT, "interp.npz"), **d)


def gen_mctf_apply(R, R0):
    """whole-picture MCTF::bilateralFilter outputs of the reference's SCALAR row (the x86 row is held to +-1 of it by the
    reference's own unit test; here they agree exactly, asserted) -> mctf_apply.npz"""
    from oracle.oracle import Oracle
    O = Oracle()          # only for the (pinned) motion fields that feed the filter
    rng = np.random.default_rng(20260926)
    d = {}
    cases = []
    for k, (w, h, bd, qp, low_res, nrefs) in enumerate(((96, 64, 10, 32, 1, 4), (80, 48, 8, 27, 0, 2), (128, 72, 10, 40, 1, 4), (64, 64, 10, 22, 0, 3))):
        yy, xx = np.mgrid[0:h + 16, 0:w + 16]
        base = (512 + 200 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 90 * np.sin((xx + yy) / 5.0)) * ((1 << bd) / 1024.0)
        def pic(dx, dy, noise):
            y = np.clip(base[8 + dy:8 + dy + h, 8 + dx:8 + dx + w] + rng.normal(0, noise, (h, w)), 0, (1 << bd) - 1).astype(np.int16)
            u = np.clip(y[::2, ::2] // 2 + (1 << (bd - 2)), 0, (1 << bd) - 1).astype(np.int16)
            v = np.clip((1 << bd) - 1 - y[::2, ::2] // 3, 0, (1 << bd) - 1).astype(np.int16)
            return y, u, v
        org = pic(0, 0, 4)
        refs = [pic(*m, n) for m, n in (((1, 0), 4), ((-1, 1), 5), ((2, -1), 4), ((-2, 0), 6))[:nrefs]]
        ref_index = [0, 0, 1, 1][:nrefs]
        mvs = [O.mctf_me(org[0], r[0], bd, 16, 4, False)[4].ravel() for r in refs]
        out0 = R0.mctf_bilateral(org, refs, mvs, ref_index, bd, qp, 16, bool(low_res), True, 0.95)
        out1 = R.mctf_bilateral(org, refs, mvs, ref_index, bd, qp, 16, bool(low_res), True, 0.95)
        for c in range(3):
            assert np.abs(out0[c].astype(np.int32) - out1[c]).max() <= 1
            d["c%d_org%d" % (k, c)] = org[c]
            d["c%d_out%d" % (k, c)] = out0[c]
            for i, r in enumerate(refs):
                d["c%d_ref%d_%d" % (k, i, c)] = r[c]
        for i, m in enumerate(mvs):
            d["c%d_mv%d" % (k, i)] = m
        cases.append((w, h, bd, qp, low_res, nrefs) + tuple(ref_index) + (0,) * (4 - nrefs))
    d["cases"] = np.array(cases, np.int32)
    np.savez_compressed(os.path.join(OUT, "mctf_apply.npz"), **d)


def gen_dmvr(R, R0):
    """DMVR refinement search results of the reference's pieces (bilinear prediction, SAD / SAD-X5, error surface) -> dmvr.npz"""
    rng = np.random.default_rng(20260927)
    yy, xx = np.mgrid[0:120, 0:192]
    tex = 512 + 220 * np.sin(xx / 6.0) * np.cos(yy / 5.0) + 80 * np.sin((xx - yy) / 3.0)
    r0 = np.clip(tex + rng.normal(0, 6, tex.shape), 0, 1023).astype(np.int16)
    d = dict(r0=r0)
    cases, outs = [], []
    for si, (sx, sy) in enumerate(((0, 0), (1, 0), (-1, 1), (2, -2))):
        r1 = np.clip(np.roll(tex, (2 * sy, 2 * sx), (0, 1)) + rng.normal(0, 6, tex.shape), 0, 1023).astype(np.int16)
        d["r1_%d" % si] = r1
        for (dx, dy) in ((16, 16), (8, 8), (16, 8), (8, 16)):
            for k in range(12):
                x0, y0 = int(rng.integers(16, 192 - 32)), int(rng.integers(16, 120 - 32))
                f = [int(v) for v in rng.integers(0, 16, 4)]
                if k < 3:
                    f[k] = 0; f[3 - k] = 0
                a = R.dmvr_refine((r0, y0, x0), (r1, y0, x0), (f[0], f[1]), (f[2], f[3]), dx, dy, 10)
                assert a == R0.dmvr_refine((r0, y0, x0), (r1, y0, x0), (f[0], f[1]), (f[2], f[3]), dx, dy, 10)
                cases.append((si, x0, y0, f[0], f[1], f[2], f[3], dx, dy)); outs.append(a)
    d["cases"] = np.array(cases, np.int32); d["out"] = np.array(outs, np.int64)
    np.savez_compressed(os.path.join(OUT, "dmvr.npz"), **d)


def gen_alf(R, R0):
    """ALF classification and covariance statistics of the reference (x86 row == scalar row) -> alf.npz.
    Statistics are stored as the float32 bit patterns of two CTUs per case (the whole array is checked by its sum of bit patterns)."""
    rng = np.random.default_rng(20260928)
    d = {}
    cases = []
    for k, (h, w, ctu, smooth) in enumerate(((144, 208, 128, False), (136, 200, 64, True), (72, 96, 32, True))):
        yy, xx = np.mgrid[0:h, 0:w]
        if smooth:
            base = 512 + 90 * np.sin(xx / 41.0) * np.cos(yy / 33.0) + 40 * np.sin((xx - yy) / 17.0) + rng.normal(0, 1.5, (h, w)) + 25 * (((xx // 24) + (yy // 40)) % 2)
        else:
            base = 512 + 200 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 80 * np.sin((xx + 2 * yy) / 5.0) + rng.normal(0, 20, (h, w))
        rec = np.clip(base, 0, 1023).astype(np.int16)
        org = np.clip(rec.astype(np.int32) + rng.integers(-12, 13, (h, w)), 0, 1023).astype(np.int16)
        cls = R.alf_classify(rec, 10, ctu, ctu - 4)
        assert np.array_equal(cls, R0.alf_classify(rec, 10, ctu, ctu - 4))
        st = R.alf_stats_plane(org, rec, ctu, 7, cls, ctu, ctu - 4)
        assert np.array_equal(st.view(np.uint32), R0.alf_stats_plane(org, rec, ctu, 7, cls, ctu, ctu - 4).view(np.uint32))
        c_org, c_rec = np.ascontiguousarray(org[::2, ::2]), np.ascontiguousarray(rec[::2, ::2])
        sc = R.alf_stats_plane(c_org, c_rec, ctu // 2, 5, None, ctu // 2, ctu // 2 - 2)
        assert np.array_equal(sc.view(np.uint32), R0.alf_stats_plane(c_org, c_rec, ctu // 2, 5, None, ctu // 2, ctu // 2 - 2).view(np.uint32))
        d["c%d_org" % k] = org; d["c%d_rec" % k] = rec; d["c%d_cls" % k] = cls
        d["c%d_luma_head" % k] = st[:2].view(np.uint32).copy(); d["c%d_luma_sum" % k] = np.array([int(st.view(np.uint32).astype(np.uint64).sum())], np.uint64)
        d["c%d_chroma_head" % k] = sc[:2].view(np.uint32).copy(); d["c%d_chroma_sum" % k] = np.array([int(sc.view(np.uint32).astype(np.uint64).sum())], np.uint64)
        # CC-ALF: the "ALF-filtered" chroma plane is a perturbed copy (any plane serves: the statistics only read it)
        slf = np.clip(c_rec.astype(np.int32) + rng.integers(-3, 4, c_rec.shape), 0, 1023).astype(np.int16)
        cc = R.ccalf_stats_plane(c_org, slf, rec, ctu // 2, ctu, ctu - 4)
        assert np.array_equal(cc.view(np.uint32), R0.ccalf_stats_plane(c_org, slf, rec, ctu // 2, ctu, ctu - 4).view(np.uint32))
        d["c%d_slf" % k] = slf; d["c%d_ccalf" % k] = cc.view(np.uint32).copy()
        cases.append((h, w, ctu))
    d["cases"] = np.array(cases, np.int32)
    np.savez_compressed(os.path.join(OUT, "alf.npz"), **d)


def gen_alf_filter(R, R0):
    """ALF / CC-ALF filtering by the reference's table entries (x86 row == scalar row), CTU by CTU like reconstructCTU / applyCcAlfFilterCTU -> alf_filter.npz"""
    rng = np.random.default_rng(20260929)
    d = {}
    cases = []
    for k, (h, w, ctu, nonlinear) in enumerate(((144, 208, 128, False), (136, 200, 64, True), (72, 96, 32, True))):
        yy, xx = np.mgrid[0:h, 0:w]
        rec = np.clip(512 + 200 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 80 * np.sin((xx + 2 * yy) / 5.0) + rng.normal(0, 20, (h, w)), 0, 1023).astype(np.int16)
        cls = R.alf_classify(rec, 10, ctu, ctu - 4)
        nctu = -(-h // ctu) * -(-w // ctu)
        clips = np.array([1024, 128, 32, 8], np.int16)
        coeff = rng.integers(-40, 41, (3, 25, 13)).astype(np.int16); coeff[..., 12] = 0
        clip = clips[rng.integers(0, 4, (3, 25, 13))] if nonlinear else np.full((3, 25, 13), 1024, np.int16)
        ctu_set = rng.integers(-1, 3, nctu).astype(np.int16)
        luma = R.alf_filter_plane(rec, ctu, 10, 7, coeff, clip, ctu_set, cls, None, ctu, ctu - 4)
        assert np.array_equal(luma, R0.alf_filter_plane(rec, ctu, 10, 7, coeff, clip, ctu_set, cls, None, ctu, ctu - 4))
        c_rec = np.ascontiguousarray(rec[::2, ::2])
        c_coeff = rng.integers(-40, 41, (4, 1, 13)).astype(np.int16); c_coeff[..., 12] = 0
        c_clip = clips[rng.integers(0, 4, (4, 1, 13))] if nonlinear else np.full((4, 1, 13), 1024, np.int16)
        c_set = rng.integers(-1, 4, nctu).astype(np.int16)
        chroma = R.alf_filter_plane(c_rec, ctu // 2, 10, 5, c_coeff, c_clip, c_set, None, None, ctu // 2, ctu // 2 - 2)
        assert np.array_equal(chroma, R0.alf_filter_plane(c_rec, ctu // 2, 10, 5, c_coeff, c_clip, c_set, None, None, ctu // 2, ctu // 2 - 2))
        cc_coeff = np.zeros((4, 8), np.int16)
        cc_coeff[:, :7] = np.array([0, 1, 2, 4, 8, 16, 32, 64], np.int16)[rng.integers(0, 8, (4, 7))] * rng.choice([-1, 1], (4, 7))
        cc_ctu = rng.integers(0, 5, nctu).astype(np.uint8)
        cc = R.ccalf_filter_plane(chroma, rec, ctu // 2, 10, cc_coeff, cc_ctu, ctu, ctu - 4)
        assert np.array_equal(cc, R0.ccalf_filter_plane(chroma, rec, ctu // 2, 10, cc_coeff, cc_ctu, ctu, ctu - 4))
        for name, v in (("rec", rec), ("cls", cls), ("coeff", coeff), ("clip", clip), ("set", ctu_set), ("luma", luma), ("c_coeff", c_coeff), ("c_clip", c_clip), ("c_set", c_set),
                        ("chroma", chroma), ("cc_coeff", cc_coeff), ("cc_ctu", cc_ctu), ("cc", cc)):
            d["c%d_%s" % (k, name)] = v
        cases.append((h, w, ctu, int(nonlinear)))
    d["cases"] = np.array(cases, np.int32)
    np.savez_compressed(os.path.join(OUT, "alf_filter.npz"), **d)


def main():
    build_ref()
    R = RefLib(1)
    R0 = RefLib(0)
    os.makedirs(OUT, exist_ok=True)
    if "--alf-filter-only" in sys.argv:
        gen_alf_filter(R, R0)
        return
    gen_distortion_ext(R, R0)
    gen_interp(R, R0)
    gen_mctf_apply(R, R0)
    gen_dmvr(R, R0)
    gen_alf(R, R0)
    gen_alf_filter(R, R0)
    if "--ext-only" in sys.argv:
        return
    rng = np.random.default_rng(20260923)

    # ---- distortion: one 160x96 plane pair, a list of (func, x, y, cx, cy, w, h, subShift) cases ----
    org, cur = rand_plane(rng, 160, 224), rand_plane(rng, 160, 224)
    cur[40:120, 60:180] = np.clip(org[40:120, 60:180].astype(np.int32) + rng.integers(-3, 4, (80, 120)), 0, 1023).astype(np.int16)
    cases, outs = [], []
    funcs = ["SAD", "SSE", "HAD", "HAD_fast"]
    for fi, f in enumerate(funcs):
        for w in (2, 4, 8, 16, 32, 64, 128):
            for h in (2, 4, 8, 16, 32, 64, 128):
                if f in ("SAD", "SSE") and w < 4:
                    continue
                for rep in range(2):
                    ox, oy = int(rng.integers(0, 224 - w)), int(rng.integers(0, 160 - h))
                    cx, cy = int(rng.integers(0, 224 - w)), int(rng.integers(0, 160 - h))
                    if rep:
                        ox, oy = int(rng.integers(60, 181 - w)) if w <= 120 else 60, int(rng.integers(40, 121 - h)) if h <= 80 else 0
                        cx, cy = ox, oy
                    ss = int(rng.integers(0, 2)) if f == "SAD" else 0
                    v = R.dist(f, (org, oy, ox), (cur, cy, cx), w, h, 10, ss)
                    assert v == R0.dist(f, (org, oy, ox), (cur, cy, cx), w, h, 10, ss)
                    cases.append((fi, ox, oy, cx, cy, w, h, ss))
                    outs.append(v)
    x5_cases, x5_out = [], []
    for w in (8, 16):
        for h in (8, 16):
            ox, oy, cx, cy = 20, 10, 100, 50
            x5_cases.append((ox, oy, cx, cy, w, h))
            x5_out.append(R.sad_x5((org, oy, ox), (cur, cy, cx), w, h, 1, True))
    h2_cases, h2_in, h2_out = [], [], []
    for w in (4, 8, 16, 32):
        for h in (4, 8, 16, 32):
            a, b = rand_plane(rng, h, w), rand_plane(rng, h, w)
            b2 = np.clip(a.astype(np.int32) + rng.integers(-3, 4, (h, w)), 0, 1023).astype(np.int16)
            for bb in (b, b2):
                h2_cases.append((w, h))
                h2_in.append(np.concatenate([a.ravel(), bb.ravel()]))
                h2_out.append(R.dist("HAD_2SAD", a, bb, w, h))
    np.savez_compressed(os.path.join(OUT, "distortion.npz"), org=org, cur=cur, funcs=np.array(funcs),
                        cases=np.array(cases, np.int32), out=np.array(outs, np.uint64),
                        x5_cases=np.array(x5_cases, np.int32), x5_out=np.array(x5_out, np.uint64),
                        h2_cases=np.array(h2_cases, np.int32), h2_in=np.concatenate(h2_in), h2_out=np.array(h2_out, np.uint64))

    # ---- transforms ----
    d = {}
    for t, name, logs in ((DCT2, "DCT2", range(1, 7)), (DCT8, "DCT8", range(2, 6)), (DST7, "DST7", range(2, 6))):
        for l in logs:
            d["mat_%s_%d" % (name, 1 << l)] = R.tr_matrix(t, l)
    tcases, tin, tcoef, tq, trec = [], [], [], [], []
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            combos = [(DCT2, DCT2)]
            if 4 <= w <= 32 and 4 <= h <= 32:
                combos += [(DST7, DST7), (DCT8, DST7), (DST7, DCT8), (DCT8, DCT8)]
            for th, tv in combos:
                bd = 10
                resi = rng.integers(-(1 << bd), 1 << bd, size=(h, w)).astype(np.int16)
                if rng.random() < 0.5:
                    resi = (resi // 16).astype(np.int16)
                coef = R.xT(resi, th, tv, bd)
                assert np.array_equal(coef, R0.xT(resi, th, tv, bd))
                cq = (coef // 5 * 5).astype(np.int32)
                rec = R.xIT(cq, th, tv, bd)
                assert np.array_equal(rec, R0.xIT(cq, th, tv, bd))
                tcases.append((w, h, th, tv, bd))
                tin.append(resi.ravel()); tcoef.append(coef.ravel()); tq.append(cq.ravel()); trec.append(rec.ravel())
    d.update(cases=np.array(tcases, np.int32), resi=np.concatenate(tin), coef=np.concatenate(tcoef),
             coef_in=np.concatenate(tq), rec=np.concatenate(trec))
    np.savez_compressed(os.path.join(OUT, "transform.npz"), **d)

    # ---- quant ----
    from oracle.oracle import Oracle
    O = Oracle()   # only for the (validated) parameter derivation helpers, outputs come from the reference
    d = {}
    for lw in range(0, 7):
        for lh in range(0, 7):
            d["scan_%d_%d" % (lw, lh)] = R.scan_order(lw, lh).astype(np.uint16 if lw + lh <= 12 else np.uint32)
    q, iq = R.quant_scales()
    d["quant_scales"], d["inv_quant_scales"] = q, iq
    qcases, qin, qlev, qdu, qdeq, qneed = [], [], [], [], [], []
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            for qp in (34, 44, 57):
                irap = int(rng.integers(0, 2))
                qc, qbits, add = O.quant_params(w, h, 10, qp, irap)
                mag = int(rng.choice([1 << 9, 1 << 12, 1 << 15]))
                coef = rng.integers(-mag, mag, size=(h, w)).astype(np.int32)
                coef[rng.random((h, w)) < 0.5] = 0
                coef[:, 32:] = 0
                coef[32:, :] = 0
                lev, du, s, last = R.quant_core(coef, qc, qbits, add, 8, sign_hiding=True)
                sc, rs, imax = O.dequant_params(w, h, 10, qp)
                deq = R.dequant_core(lev, sc, rs, imax)
                nqc, nqbits, nadd, num = O.need_rdoq_params(w, h, 10, qp, 1)
                small = (coef // max(1, mag >> 4)).astype(np.int32)
                need = (R.need_rdoq(coef.ravel()[:num], nqc, nadd, nqbits), R.need_rdoq(small.ravel()[:num], nqc, nadd, nqbits))
                qcases.append((w, h, qp, irap, qc, qbits, s, last, sc, rs, imax, nqc, nqbits, num, need[0], need[1], max(1, mag >> 4)))
                qdu_masked = du.copy()
                keep = np.zeros(h * w, bool)
                keep[R.scan_order(w.bit_length() - 1, h.bit_length() - 1)[: last + 1]] = True
                qdu_masked[~keep] = 0
                qin.append(coef.ravel()); qlev.append(lev.ravel()); qdu.append(qdu_masked); qdeq.append(deq.ravel())
    d.update(cases=np.array(qcases, np.int64), coef=np.concatenate(qin), level=np.concatenate(qlev),
             deltaU=np.concatenate(qdu), dequant=np.concatenate(qdeq))
    np.savez_compressed(os.path.join(OUT, "quant.npz"), **d)

    # ---- MCTF ----
    d = {}
    f8, f4 = R.mctf_filters()
    d["filter8"], d["filter4"] = f8, f4
    org, buf = rand_plane(rng, 96, 112), rand_plane(rng, 96, 112)
    ecases, eout = [], []
    for w in range(8, 65, 8):
        for h in range(8, 65, 8):
            ox, oy, bx, by = 8, 8, int(rng.integers(4, 40)), int(rng.integers(4, 20))
            ecases.append((0, ox, oy, bx, by, w, h, 0, 0)); eout.append(R.mctf_err_int((org, oy, ox), (buf, by, bx), w, h))
            for tap4 in (0, 1):
                fx, fy = int(rng.integers(0, 16)), int(rng.integers(1, 16))
                ecases.append((1 + tap4, ox, oy, bx, by, w, h, fx, fy))
                eout.append(R.mctf_err_frac(tap4, (org, oy, ox), (buf, by, bx), w, h, fx, fy, 10))
    for fx in range(16):
        for fy in range(16):
            if fx or fy:
                for tap4 in (0, 1):
                    ecases.append((1 + tap4, 8, 8, 30, 12, 16, 16, fx, fy))
                    eout.append(R.mctf_err_frac(tap4, (org, 8, 8), (buf, 12, 30), 16, 16, fx, fy, 10))
    vcases, vout = [], []
    for w in (8, 16, 32):
        for h in (8, 16, 32):
            vcases.append((5, 3, w, h)); vout.append(R.mctf_calc_var((org, 3, 5), w, h))
    d.update(org=org, buf=buf, err_cases=np.array(ecases, np.int32), err_out=np.array(eout, np.int32),
             var_cases=np.array(vcases, np.int32), var_out=np.array(vout, np.float64), sub_out=R.mctf_subsample(org))
    me_cfgs = [(176, 144, 4, 16, 0), (200, 120, 4, 8, 0), (176, 144, 0, 16, 0), (256, 136, 2, 16, 1), (328, 200, 4, 16, 1)]
    d["me_cfgs"] = np.array(me_cfgs, np.int32)
    for i, (w, h, speed, unit, add) in enumerate(me_cfgs):
        a, b = synth_pair(np.random.default_rng(500 + i), h, w, shift=(3 + i, 1 + (i & 1)))
        lv = R.mctf_me(a, b, 10, unit, speed, bool(add))
        lv0 = R0.mctf_me(a, b, 10, unit, speed, bool(add))
        d["me%d_org" % i], d["me%d_ref" % i] = a, b
        for k in range(5):
            if lv[k] is not None:
                assert all(np.array_equal(lv[k][f], lv0[k][f]) for f in ("x", "y", "error"))
                d["me%d_l%d" % (i, k)] = lv[k]
    np.savez_compressed(os.path.join(OUT, "mctf.npz"), **d)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
