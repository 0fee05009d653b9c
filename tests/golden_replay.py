"""Replays tests/golden/*.npz (inputs + outputs of the reference itself, written by tests/gen_golden.py)
against any backend exposing the Oracle method names (oracle.oracle.Oracle, or the HIP backend
adapter in tests/hip_backend.py).  Bit-exact (tolerance 0) everywhere."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def check_distortion(be):
    g = load("distortion")
    org, cur = g["org"], g["cur"]
    funcs = [str(f) for f in g["funcs"]]
    for (fi, ox, oy, cx, cy, w, h, ss), exp in zip(g["cases"], g["out"]):
        got = be.dist(funcs[fi], (org, int(oy), int(ox)), (cur, int(cy), int(cx)), int(w), int(h), 10, int(ss))
        assert got == int(exp), (funcs[fi], w, h, ss, got, int(exp))
    for (ox, oy, cx, cy, w, h), exp in zip(g["x5_cases"], g["x5_out"]):
        got = be.sad_x5((org, int(oy), int(ox)), (cur, int(cy), int(cx)), int(w), int(h), 1, True)
        assert np.array_equal(got, exp), (w, h)
    off = 0
    for (w, h), exp in zip(g["h2_cases"], g["h2_out"]):
        n = int(w) * int(h)
        a = np.ascontiguousarray(g["h2_in"][off:off + n].reshape(h, w))
        b = np.ascontiguousarray(g["h2_in"][off + n:off + 2 * n].reshape(h, w))
        off += 2 * n
        assert be.dist("HAD_2SAD", a, b, int(w), int(h)) == int(exp), (w, h)


def check_distortion_ext(be):
    g = load("distortion_ext")
    org, cur, mask = g["org"], g["cur"], g["mask"]
    for (ox, oy, cx, cy, mx, my, sx, ms2, w, h, ss), exp in zip(g["mask_cases"], g["mask_out"]):
        got = be.sad_mask((org, int(oy), int(ox)), (cur, int(cy), int(cx)), (mask, int(my), int(mx)), int(sx), int(ms2), int(w), int(h), int(ss))
        assert got == int(exp), ("mask", w, h, ss, sx, got, int(exp))
    for (ox, oy, cx, cy, w, h, wt), exp in zip(g["wsse_cases"], g["wsse_out"]):
        got = be.fix_weighted_sse((org, int(oy), int(ox)), (cur, int(cy), int(cx)), int(w), int(h), int(wt))
        assert got == int(exp), ("wsse", w, h, wt, got, int(exp))


def check_transform_matrices(be):
    g = load("transform")
    for t, name, logs in ((0, "DCT2", range(1, 7)), (1, "DCT8", range(2, 6)), (2, "DST7", range(2, 6))):
        for l in logs:
            assert np.array_equal(be.tr_matrix(t, l), g["mat_%s_%d" % (name, 1 << l)]), (name, l)


def check_transform(be):
    g = load("transform")
    off = 0
    for (w, h, th, tv, bd) in g["cases"]:
        n = int(w) * int(h)
        resi = g["resi"][off:off + n].reshape(h, w)
        coef = g["coef"][off:off + n].reshape(h, w)
        cin = g["coef_in"][off:off + n].reshape(h, w)
        rec = g["rec"][off:off + n].reshape(h, w)
        off += n
        assert np.array_equal(be.xT(resi, int(th), int(tv), int(bd)), coef), ("xT", w, h, th, tv)
        assert np.array_equal(be.xIT(cin, int(th), int(tv), int(bd)), rec), ("xIT", w, h, th, tv)


def check_scan(be):
    g = load("quant")
    for lw in range(0, 7):
        for lh in range(0, 7):
            assert np.array_equal(be.scan_order(lw, lh), g["scan_%d_%d" % (lw, lh)].astype(np.uint32)), (lw, lh)


def check_quant(be):
    g = load("quant")
    off = 0
    for c in g["cases"]:
        w, h, qp, irap, qc, qbits, s, last, sc, rs, imax, nqc, nqbits, num, need0, need1, div = [int(v) for v in c]
        n = w * h
        coef = g["coef"][off:off + n].reshape(h, w)
        lev = g["level"][off:off + n].reshape(h, w)
        du = g["deltaU"][off:off + n]
        deq = g["dequant"][off:off + n].reshape(h, w)
        off += n
        add = (171 if irap else 85) << (qbits - 9)
        assert be.quant_params(w, h, 10, qp, irap) == (qc, qbits, add)
        q, d, ssum, slast = be.quant_core(coef, qc, qbits, add, 8)
        assert (ssum, slast) == (s, last), (w, h, qp)
        assert np.array_equal(q, lev), (w, h, qp)
        keep = np.zeros(n, bool)
        keep[be.scan_order(w.bit_length() - 1, h.bit_length() - 1)[: last + 1]] = True
        assert np.array_equal(np.where(keep, d, 0), du), (w, h, qp)
        assert be.dequant_params(w, h, 10, qp) == (sc, rs, imax)
        assert np.array_equal(be.dequant_core(lev, sc, rs, imax), deq), (w, h, qp)
        nadd = 171 << (nqbits - 9)
        assert be.need_rdoq_params(w, h, 10, qp, 1) == (nqc, nqbits, nadd, num)
        small = (coef // div).astype(np.int32)
        assert be.need_rdoq(coef.ravel()[:num], nqc, nadd, nqbits) == need0
        assert be.need_rdoq(small.ravel()[:num], nqc, nadd, nqbits) == need1


def check_quant_qp(be):
    """same golden cases through the (qp, flags) entry points (what the C ABI exposes)"""
    g = load("quant")
    off = 0
    for c in g["cases"]:
        w, h, qp, irap, qc, qbits, s, last, sc, rs, imax, nqc, nqbits, num, need0, need1, div = [int(v) for v in c]
        n = w * h
        coef = g["coef"][off:off + n].reshape(h, w)
        lev = g["level"][off:off + n].reshape(h, w)
        du = g["deltaU"][off:off + n]
        deq = g["dequant"][off:off + n].reshape(h, w)
        off += n
        q, d, ssum, slast = be.quant_tu(coef, qp, irap, 8)
        assert (ssum, slast) == (s, last), (w, h, qp, (ssum, slast), (s, last))
        assert np.array_equal(q, lev), (w, h, qp)
        keep = np.zeros(n, bool)
        keep[be.scan_order(w.bit_length() - 1, h.bit_length() - 1)[: last + 1]] = True
        assert np.array_equal(np.where(keep, d, 0), du), (w, h, qp)
        assert np.array_equal(be.dequant_tu(lev, qp), deq), (w, h, qp)
        small = (coef // div).astype(np.int32)
        if h <= 32:   # the TU-level entry reads the first w*min(h,32) coefficients itself
            assert be.need_rdoq_tu(coef, qp, 1) == need0, (w, h, qp)
            assert be.need_rdoq_tu(small, qp, 1) == need1, (w, h, qp)


def check_mctf_kernels(be):
    g = load("mctf")
    org, buf = g["org"], g["buf"]
    for (kind, ox, oy, bx, by, w, h, fx, fy), exp in zip(g["err_cases"], g["err_out"]):
        o, b = (org, int(oy), int(ox)), (buf, int(by), int(bx))
        if kind == 0:
            got = be.mctf_err_int(o, b, int(w), int(h))
        else:
            got = be.mctf_err_frac(int(kind) - 1, o, b, int(w), int(h), int(fx), int(fy), 10)
        assert got == int(exp), (kind, w, h, fx, fy)
    for (x, y, w, h), exp in zip(g["var_cases"], g["var_out"]):
        assert be.mctf_calc_var((org, int(y), int(x)), int(w), int(h)) == float(exp)
    assert np.array_equal(be.mctf_subsample(org), g["sub_out"])


def check_mctf_me(be, which=None):
    g = load("mctf")
    for i, (w, h, speed, unit, add) in enumerate(g["me_cfgs"]):
        if which is not None and i not in which:
            continue
        lv = be.mctf_me(g["me%d_org" % i], g["me%d_ref" % i], 10, int(unit), int(speed), bool(add))
        for k in range(5):
            key = "me%d_l%d" % (i, k)
            if key not in g:
                assert lv[k] is None
                continue
            exp = g[key]
            for f in ("x", "y", "error"):
                assert np.array_equal(lv[k][f], exp[f]), (i, k, f, np.argwhere(lv[k][f] != exp[f])[:4])
            if k == 4:
                assert np.array_equal(lv[k]["rmsme"], exp["rmsme"]) and np.array_equal(lv[k]["overlap"], exp["overlap"])


def check_interp(be, slots=True):
    """interpolation filter fixtures (reference outputs).  `be` needs if_filter/if_copy (slots) and if_pred_luma;
    the two-pass sub-pel form is checked through if_pred_luma_me when the backend has it, else through if_pred_luma(mode=reduce_tap)."""
    g = load("interp")
    plane, inter = g["plane"], g["inter"]
    if slots:
        off = 0
        for (n, set_, p, w, h, vertical, first, last) in g["slot_cases"]:
            _, c = _coeff(be, int(set_), int(p))
            exp = g["slot_out"][off:off + w * h].reshape(h, w); off += w * h
            got = be.if_filter(int(n), int(vertical), int(first), int(last), 10, (plane if first else inter, 24, 32), int(w), int(h), c)
            assert np.array_equal(got, exp), ("slot", n, set_, p, w, h, vertical, first, last)
        off = 0
        for (w, h, first, last, bi) in g["copy_cases"]:
            exp = g["copy_out"][off:off + w * h].reshape(h, w); off += w * h
            got = be.if_copy(int(first), int(last), 10, (plane if first else inter, 24, 32), int(w), int(h), bool(bi))
            assert np.array_equal(got, exp), ("copy", w, h, first, last, bi)
    off = 0
    for (bd, w, h, xf, yf, alt, rnd) in g["pred_cases"]:
        exp = g["pred_out"][off:off + w * h].reshape(h, w); off += w * h
        got = be.if_pred_luma((g["plane%d" % bd], 24, 20), int(w), int(h), int(xf), int(yf), bool(rnd), int(bd), bool(alt))
        assert np.array_equal(got, exp), ("pred", bd, w, h, xf, yf, alt, rnd)
    off = 0
    for (w, h, xf, yf, alt, rt) in g["me_cases"]:
        exp = g["me_out"][off:off + w * h].reshape(h, w); off += w * h
        if hasattr(be, "if_pred_luma_me"):
            got = be.if_pred_luma_me((plane, 24, 20), int(w), int(h), int(xf), int(yf), 10, bool(alt), int(rt))
        else:
            got = be.if_pred_luma((plane, 24, 20), int(w), int(h), int(xf), int(yf), True, 10, bool(alt), int(rt))
        assert np.array_equal(got, exp), ("me", w, h, xf, yf, alt, rt)


_COEFF_CACHE = {}


def _coeff(be, set_, p):
    if hasattr(be, "if_coeff"):
        return be.if_coeff(set_, p)
    if not _COEFF_CACHE:
        from oracle.oracle import Oracle
        _COEFF_CACHE["o"] = Oracle()
    return _COEFF_CACHE["o"].if_coeff(set_, p)


def check_mctf_apply(be):
    """whole-picture bilateral temporal filter against the reference's scalar row, exact"""
    g = load("mctf_apply")
    for k, row in enumerate(g["cases"]):
        w, h, bd, qp, low_res, nrefs = [int(v) for v in row[:6]]
        ref_index = [int(v) for v in row[6:6 + nrefs]]
        org = tuple(g["c%d_org%d" % (k, c)] for c in range(3))
        refs = [tuple(g["c%d_ref%d_%d" % (k, i, c)] for c in range(3)) for i in range(nrefs)]
        mvs = [g["c%d_mv%d" % (k, i)] for i in range(nrefs)]
        out = be.mctf_bilateral(org, refs, mvs, ref_index, bd, qp, 16, bool(low_res), True, 0.95)
        for c in range(3):
            assert np.array_equal(out[c], g["c%d_out%d" % (k, c)]), ("mctf apply", k, c, int(np.abs(out[c].astype(np.int32) - g["c%d_out%d" % (k, c)]).max()))


def check_dmvr(be):
    """DMVR refinement search fixtures"""
    g = load("dmvr")
    r0 = g["r0"]
    for (si, x0, y0, f0x, f0y, f1x, f1y, dx, dy), exp in zip(g["cases"], g["out"]):
        got = be.dmvr_refine((r0, int(y0), int(x0)), (g["r1_%d" % si], int(y0), int(x0)), (int(f0x), int(f0y)), (int(f1x), int(f1y)), int(dx), int(dy), 10)
        assert tuple(got) == tuple(int(v) for v in exp), (si, x0, y0, dx, dy, got, exp)


def check_alf(be):
    """ALF classification + covariance statistics fixtures (float32 bit patterns)"""
    g = load("alf")
    for k, (h, w, ctu) in enumerate(g["cases"]):
        org, rec = g["c%d_org" % k], g["c%d_rec" % k]
        ctu = int(ctu)
        cls = be.alf_classify(rec, 10, ctu, ctu - 4)
        assert np.array_equal(np.asarray(cls), g["c%d_cls" % k]), ("alf classes", k)
        st = np.asarray(be.alf_stats_plane(org, rec, ctu, 7, g["c%d_cls" % k], ctu, ctu - 4)).view(np.uint32)
        assert np.array_equal(st[:2], g["c%d_luma_head" % k]) and int(st.astype(np.uint64).sum()) == int(g["c%d_luma_sum" % k][0]), ("alf luma", k)
        c_org, c_rec = np.ascontiguousarray(org[::2, ::2]), np.ascontiguousarray(rec[::2, ::2])
        sc = np.asarray(be.alf_stats_plane(c_org, c_rec, ctu // 2, 5, None, ctu // 2, ctu // 2 - 2)).view(np.uint32)
        assert np.array_equal(sc[:2], g["c%d_chroma_head" % k]) and int(sc.astype(np.uint64).sum()) == int(g["c%d_chroma_sum" % k][0]), ("alf chroma", k)
        cc = np.asarray(be.ccalf_stats_plane(c_org, g["c%d_slf" % k], rec, ctu // 2, ctu, ctu - 4)).view(np.uint32)
        assert np.array_equal(cc, g["c%d_ccalf" % k]), ("cc-alf", k)


def check_alf_filter(be):
    """ALF / CC-ALF filtering fixtures: whole filtered planes of the reference (linear and non-linear entries, disabled CTUs, virtual-boundary rows)"""
    g = load("alf_filter")
    for k, (h, w, ctu, nonlinear) in enumerate(g["cases"]):
        ctu = int(ctu)
        rec = g["c%d_rec" % k]
        luma = be.alf_filter_plane(rec, ctu, 10, 7, g["c%d_coeff" % k], g["c%d_clip" % k], g["c%d_set" % k], g["c%d_cls" % k], None, ctu, ctu - 4)
        assert np.array_equal(np.asarray(luma), g["c%d_luma" % k]), ("alf filter luma", k)
        c_rec = np.ascontiguousarray(rec[::2, ::2])
        chroma = be.alf_filter_plane(c_rec, ctu // 2, 10, 5, g["c%d_c_coeff" % k], g["c%d_c_clip" % k], g["c%d_c_set" % k], None, None, ctu // 2, ctu // 2 - 2)
        assert np.array_equal(np.asarray(chroma), g["c%d_chroma" % k]), ("alf filter chroma", k)
        cc = be.ccalf_filter_plane(g["c%d_chroma" % k], rec, ctu // 2, 10, g["c%d_cc_coeff" % k], g["c%d_cc_ctu" % k], ctu, ctu - 4)
        assert np.array_equal(np.asarray(cc), g["c%d_cc" % k]), ("cc-alf filter", k)
