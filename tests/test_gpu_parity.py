"""-m gpu: the HIP path (through the C ABI) against (1) the committed golden vectors = outputs of the reference itself,
(2) the CPU oracle on seeded random batches, (3) size-independent properties at BASELINE sizes (1080p / 4K planes).
Everything is bit-exact (tolerance 0): integer kernels, and the two IEEE-double corners (rectangular Hadamard tiles,
MCTF error normalisation) are compiled without FMA contraction."""
import os

import numpy as np
import pytest

import golden_replay as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from hip_backend import HipBackend
    return HipBackend()


def rand_plane(rng, h, w, bits=10):
    return rng.integers(0, 1 << bits, size=(h, w), dtype=np.int16)


# ---------------------------------------------------------------- golden (reference outputs) ----
def test_golden_distortion(hip):
    G.check_distortion(hip)


def test_golden_distortion_ext(hip):
    G.check_distortion_ext(hip)


def test_golden_dmvr(hip):
    G.check_dmvr(hip)


def test_golden_alf(hip):
    G.check_alf(hip)


def test_golden_alf_filter(hip):
    G.check_alf_filter(hip)


def test_golden_mctf_apply(hip):
    G.check_mctf_apply(hip)


def test_golden_interp(hip):
    G.check_interp(hip)


def test_golden_transform(hip):
    G.check_transform_matrices(hip)
    G.check_scan(hip)
    G.check_transform(hip)


def test_golden_quant(hip):
    G.check_quant_qp(hip)


def test_golden_mctf_kernels(hip):
    G.check_mctf_kernels(hip)


def test_golden_mctf_me(hip):
    G.check_mctf_me(hip)


# ---------------------------------------------------------------- oracle on random batches ----
@pytest.mark.parametrize("func", ["SAD", "SSE", "HAD", "HAD_fast", "HAD_2SAD"])
def test_dist_batches_vs_oracle(hip, oracle, func):
    rng = np.random.default_rng(100)
    org, cur = rand_plane(rng, 200, 328), rand_plane(rng, 200, 328)
    cur[50:150, 100:300] = np.clip(org[50:150, 100:300].astype(np.int32) + rng.integers(-4, 5, (100, 200)), 0, 1023).astype(np.int16)
    if func in ("SAD", "SSE"):      # signed samples too (residual-domain use)
        org[150:, :] -= 512
        cur[150:, :] -= 700
    sizes = [(w, h) for w in (4, 8, 16, 32, 64, 128) for h in (4, 8, 16, 32, 64, 128)]
    if func.startswith("HAD"):
        sizes += [(2, 2), (2, 8), (8, 2), (4, 2)]
    if func in ("SAD", "SSE"):
        sizes += [(2, 4), (12, 8), (24, 16), (6, 6)]
    for (w, h) in sizes:
        n = 37
        items = [(int(rng.integers(0, 328 - w + 1)), int(rng.integers(0, 200 - h + 1)),
                  int(rng.integers(0, 328 - w + 1)), int(rng.integers(0, 200 - h + 1))) for _ in range(n)]
        for ss in ((0, 1) if func == "SAD" and h >= 2 else (0,)):
            got = hip.dist_many(func, org, cur, items, w, h, 10, ss)
            for k, (ox, oy, cx, cy) in enumerate(items):
                if func == "HAD_2SAD":
                    a = np.ascontiguousarray(org[oy:oy + h, ox:ox + w]); b = np.ascontiguousarray(cur[cy:cy + h, cx:cx + w])
                    exp = oracle.dist(func, a, b, w, h)
                else:
                    exp = oracle.dist(func, (org, oy, ox), (cur, cy, cx), w, h, 10, ss)
                assert int(got[k]) == exp, (func, w, h, ss, k, int(got[k]), exp)


def test_sad_x5_vs_oracle(hip, oracle):
    rng = np.random.default_rng(101)
    org, cur = rand_plane(rng, 64, 96), rand_plane(rng, 64, 96)
    for w in (8, 16):
        for h in (8, 16):
            for centre in (True, False):
                a = hip.sad_x5((org, 5, 10), (cur, 9, 40), w, h, 1, centre)
                b = oracle.sad_x5((org, 5, 10), (cur, 9, 40), w, h, 1, centre)
                if not centre:
                    a = a.copy(); a[2] = 0; b[2] = 0
                assert np.array_equal(a, b)


def test_sad_mask_and_weighted_sse_batches_vs_oracle(hip, oracle):
    """DF_SAD_WITH_MASK and m_fxdWtdPredPtr in batch form: per-candidate mask offsets / weights, odd sizes, signed masks"""
    rng = np.random.default_rng(102)
    org, cur = rand_plane(rng, 120, 200), rand_plane(rng, 120, 200)
    mask = rng.integers(-9, 9, size=(260, 448)).astype(np.int16)      # 448 = a multiple of 64: the device plane keeps this pitch, so reads that run across a row end see the same samples
    hp = hip.hp
    po, pc, pm = hip._plane(org), hip._plane(cur), hip._plane(mask)
    for (w, h) in ((4, 4), (8, 8), (16, 8), (8, 32), (64, 64), (12, 6), (128, 16)):
        n = 29
        pos = [(int(rng.integers(0, 200 - w + 1)), int(rng.integers(0, 120 - h + 1)), int(rng.integers(0, 200 - w + 1)), int(rng.integers(0, 120 - h + 1)),
                int(rng.integers(130, 280)), int(rng.integers(0, 120))) for _ in range(n)]
        d_it = hp.to_device(np.array([(oy * po.stride + ox, cy * pc.stride + cx) for (ox, oy, cx, cy, _, _) in pos], np.int32))
        d_mo = hp.to_device(np.array([my * pm.stride + mx for (_, _, _, _, mx, my) in pos], np.int32))
        for ss in ((0, 1) if h % 2 == 0 else (0,)):
            for step_x, ms2 in ((1, -w), (-1, w), (1, 3), (2, -5)):
                got = hp.sad_mask_batch(po, pc, pm, step_x, ms2, d_it, d_mo, n, w, h, ss).cpu().numpy().view(np.uint64)
                for k, (ox, oy, cx, cy, mx, my) in enumerate(pos):
                    exp = oracle.sad_mask((org, oy, ox), (cur, cy, cx), (mask, my, mx), step_x, ms2, w, h, ss)
                    assert int(got[k]) == exp, ("mask", w, h, ss, step_x, ms2, k, int(got[k]), exp)
        if w % 2 == 0:
            wts = rng.integers(1, 1 << 20, size=n).astype(np.uint32)
            got = hp.fix_weighted_sse_batch(po, pc, d_it, hp.to_device(wts.view(np.int32)), n, w, h).cpu().numpy().view(np.uint64)
            for k, (ox, oy, cx, cy, _, _) in enumerate(pos):
                exp = oracle.fix_weighted_sse((org, oy, ox), (cur, cy, cx), w, h, int(wts[k]))
                assert int(got[k]) == exp, ("wsse", w, h, k, int(got[k]), exp)


def test_tcoeffops_slots_vs_oracle(hip, oracle):
    """the five g_tCoeffOps slot kinds through the C ABI (device pointers), caller-supplied matrices"""
    import torch
    hp = hip.hp
    rng = np.random.default_rng(166)

    def dev(a):
        return hp.to_device(np.ascontiguousarray(a))
    for t, logs in ((0, range(2, 7)), (2, range(2, 6)), (1, range(2, 6))):
        for l in logs:
            n = 1 << l
            tc = hip.tr_matrix(t, l)
            d_tc = dev(tc)
            for line in (4, 8, 32, 64):
                skip, skip2 = (line // 2 if line >= 8 else 0), (n // 2 if n >= 32 else 0)
                for red, cut in ((line, n), (line - skip, n - skip2)):
                    src = rng.integers(-(1 << 15), 1 << 15, size=(line, n)).astype(np.int32)
                    shift = int(rng.integers(1, 12))
                    d_dst = torch.zeros(n * line, dtype=torch.int32, device=hp.device)
                    hp._ck(hp.L.vvhip_fast_fwd_core(hp.ctx, n, d_tc.data_ptr(), dev(src).data_ptr(), d_dst.data_ptr(), line, red, cut, shift))
                    assert np.array_equal(d_dst.cpu().numpy().reshape(n, line), oracle.fast_fwd_core(tc, src, line, red, cut, shift)), ("fwd", t, n, line, red, cut)
                    srci = rng.integers(-(1 << 15), 1 << 15, size=(n, line)).astype(np.int32)
                    d0 = rng.integers(-1000, 1000, size=(line, n)).astype(np.int32)
                    d_acc = dev(d0)
                    hp._ck(hp.L.vvhip_fast_inv_core(hp.ctx, n, d_tc.data_ptr(), dev(srci).data_ptr(), d_acc.data_ptr(), line, red, cut))
                    assert np.array_equal(d_acc.cpu().numpy().reshape(line, n), oracle.fast_inv_core(tc, srci, d0, line, red, cut)), ("inv", t, n, line, red, cut)
    for w in (4, 8, 16, 64):
        for h in (2, 8, 64):
            buf = rng.integers(-(1 << 24), 1 << 24, size=(h, w + 4)).astype(np.int32)
            d = dev(buf)
            hp._ck(hp.L.vvhip_round_clip(hp.ctx, d.data_ptr(), w, h, w + 4, -32768, 32767, 64, 7))
            assert np.array_equal(d.cpu().numpy().reshape(h, w + 4), oracle.round_clip(buf, w, h, w + 4, -32768, 32767, 64, 7)), ("clip", w, h)
            src = rng.integers(-(1 << 17), 1 << 17, size=(h, w)).astype(np.int32)
            d_pel = torch.full((h * (w + 8),), -77, dtype=torch.int16, device=hp.device)
            hp._ck(hp.L.vvhip_cpy_resi(hp.ctx, dev(src).data_ptr(), d_pel.data_ptr(), w + 8, w, h))
            assert np.array_equal(d_pel.cpu().numpy().reshape(h, w + 8), oracle.cpy_resi(src, w, h, w + 8)), ("cpyResi", w, h)
            pel = rng.integers(-(1 << 15), 1 << 15, size=(h, w + 8)).astype(np.int16)
            d_c = torch.zeros(h * w, dtype=torch.int32, device=hp.device)
            hp._ck(hp.L.vvhip_cpy_coeff(hp.ctx, dev(pel).data_ptr(), w + 8, d_c.data_ptr(), w, h))
            assert np.array_equal(d_c.cpu().numpy().reshape(h, w), oracle.cpy_coeff(pel, w, h)), ("cpyCoeff", w, h)


def test_sad_surface_vs_oracle(hip, oracle):
    from vvenc_amd.hotpath import Plane
    rng = np.random.default_rng(102)
    org, ref = rand_plane(rng, 160, 256), rand_plane(rng, 160, 256)
    hp = hip.hp
    po, pr = Plane.from_numpy(hp.device, org), Plane.from_numpy(hp.device, ref)
    for (w, h, ss, rx, ry) in [(8, 8, 0, 4, 3), (16, 16, 1, 8, 8), (32, 32, 1, 16, 8), (64, 64, 1, 6, 6), (4, 4, 0, 2, 2)]:
        blocks = [(int(rng.integers(rx, 256 - w - rx)), int(rng.integers(ry, 160 - h - ry))) for _ in range(5)]
        oo = np.array([y * po.stride + x for x, y in blocks], np.int32)
        ro = np.array([y * pr.stride + x for x, y in blocks], np.int32)
        out = hp.sad_surface(po, pr, hp.to_device(oo), hp.to_device(ro), len(blocks), w, h, ss, rx, ry).cpu().numpy()
        out = out.reshape(len(blocks), 2 * ry + 1, 2 * rx + 1)
        for b, (x, y) in enumerate(blocks):
            for dy in range(-ry, ry + 1):
                for dx in range(-rx, rx + 1):
                    assert int(out[b, dy + ry, dx + rx]) == oracle.dist("SAD", (org, y, x), (ref, y + dy, x + dx), w, h, 10, ss)


def _tr_types(w, h):
    yield 0, 0
    if 4 <= w <= 32 and 4 <= h <= 32:
        for th in (2, 1):
            for tv in (2, 1):
                yield th, tv
    if 4 <= w <= 32:
        yield 2, 0
    if 4 <= h <= 32:
        yield 0, 2


def test_transform_batches_vs_oracle(hip, oracle):
    rng = np.random.default_rng(103)
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            for th, tv in _tr_types(w, h):
                for bd in (8, 10):
                    n = 19
                    resi = rng.integers(-(1 << bd), 1 << bd, size=(n, h, w)).astype(np.int16)
                    resi[0] = (1 << bd) - 1
                    resi[1] = -(1 << bd)
                    got = hip.xT_many(resi, th, tv, bd)
                    exp = np.stack([oracle.xT(resi[i], th, tv, bd) for i in range(n)])
                    assert np.array_equal(got, exp), ("xT", w, h, th, tv, bd)
                    coef = (exp // 3).astype(np.int32)
                    coef[2] = rng.integers(-(1 << 15), 1 << 15, size=(h, w))
                    got = hip.xIT_many(coef, th, tv, bd)
                    exp = np.stack([oracle.xIT(coef[i], th, tv, bd) for i in range(n)])
                    assert np.array_equal(got, exp), ("xIT", w, h, th, tv, bd)


def test_quant_batches_vs_oracle(hip, oracle):
    from vvenc_amd.hotpath import HotPath
    rng = np.random.default_rng(104)
    hp = hip.hp
    for w in (2, 4, 8, 16, 32, 64):
        for h in (2, 4, 8, 16, 32, 64):
            n = 23
            qps = rng.integers(12, 64, size=n)
            irap = rng.integers(0, 2, size=n)
            luma = rng.integers(0, 2, size=n)
            mags = rng.choice([8, 1 << 9, 1 << 12, 1 << 15], size=n)
            coef = np.stack([rng.integers(-m, m, size=(h, w)) for m in mags]).astype(np.int32)
            coef[rng.random(coef.shape) < 0.5] = 0
            coef[3] = 0
            d_coef = hp.to_device(coef.ravel())
            d_qp = hp.to_device(HotPath.tu_qp(qps, irap, luma))
            for thr in (8, 4):
                lev, du, s, last = hp.quant(d_coef, n, w, h, d_qp, 10, thr, True)
                lev, du, s, last = lev.cpu().numpy().reshape(n, h, w), du.cpu().numpy().reshape(n, h * w), s.cpu().numpy(), last.cpu().numpy()
                for i in range(n):
                    c = coef[i].copy()
                    c[:, 32:] = c[:, 32:]   # zero-out region is ignored by the quantiser (never scanned)
                    e = oracle.quant_tu(c, int(qps[i]), int(irap[i]), thr)
                    assert (int(s[i]), int(last[i])) == (e[2], e[3]), (w, h, i, thr)
                    assert np.array_equal(lev[i], e[0]), (w, h, i)
                    keep = np.zeros(h * w, bool)
                    keep[oracle.scan_order(w.bit_length() - 1, h.bit_length() - 1)[: e[3] + 1]] = True
                    assert np.array_equal(np.where(keep, du[i], 0), np.where(keep, e[1], 0)), (w, h, i)
            deq = hp.dequant(hp.to_device(lev.ravel()), n, w, h, d_qp, 10).cpu().numpy().reshape(n, h, w)
            need = hp.need_rdoq(d_coef, n, w, h, d_qp, 10).cpu().numpy()
            for i in range(n):
                assert np.array_equal(deq[i], oracle.dequant_tu(lev[i], int(qps[i]))), (w, h, i)
                assert int(need[i]) == oracle.need_rdoq_tu(coef[i], int(qps[i]), int(luma[i])), (w, h, i)


def test_fused_tu_rdo_vs_oracle(hip, oracle):
    from vvenc_amd.hotpath import STATS_DTYPE, HotPath, Plane
    rng = np.random.default_rng(105)
    hp = hip.hp
    for w in (4, 8, 16, 32, 64):
        for h in (4, 8, 16, 32, 64):
            for th, tv in [(0, 0)] + ([(2, 2), (1, 2)] if w <= 32 and h <= 32 else []):
                n = 17
                scale = rng.choice([4, 32, 256, 1023], size=n)
                resi = np.stack([rng.integers(-s, s + 1, size=(h, w)) for s in scale]).astype(np.int16)
                qps = rng.integers(20, 60, size=n)
                irap = rng.integers(0, 2, size=n)
                plane = Plane.from_numpy(hp.device, resi.reshape(n * h, w))
                d_off = hp.to_device(np.arange(n, dtype=np.int32) * (h * plane.stride))
                d_qp = hp.to_device(HotPath.tu_qp(qps, irap, 1))
                lev, rec, st = hp.tu_rdo(plane, d_off, n, w, h, d_qp, th, tv, 10, 8)
                lev, rec = lev.cpu().numpy().reshape(n, h, w), rec.cpu().numpy().reshape(n, h, w)
                st = st.cpu().numpy().view(STATS_DTYPE).reshape(n)
                for i in range(n):
                    el, er, es = oracle.tu_rdo(resi[i], int(qps[i]), int(irap[i]), th, tv, 10, 8, 1)
                    assert np.array_equal(lev[i], el), ("lev", w, h, th, tv, i)
                    assert np.array_equal(rec[i], er), ("rec", w, h, th, tv, i)
                    assert (int(st["abs_sum"][i]), int(st["last_scan_pos"][i]), int(st["need_rdoq"][i]), int(st["sse"][i])) == \
                        (es["abs_sum"], es["last_scan_pos"], es["need_rdoq"], es["sse"]), (w, h, th, tv, i)


def test_fused_tu_matrix_core_vs_oracle_and_row_kernel(hip, oracle, monkeypatch):
    """the matrix-core form of the fused TU kernel (8/16/32-point square TUs as int8 MFMA products): every transform-type pair, bit depths 8 and 10,
    ragged tile counts, unaligned residual offsets, per-TU QPs over the whole range, residual magnitudes up to the 16-bit limit (the 64-bit
    quantiser path) — against the oracle for conforming residuals and against the dot-product row kernel for everything"""
    import torch
    from vvenc_amd.hotpath import STATS_DTYPE, HotPath
    rng = np.random.default_rng(511)
    hp = hip.hp
    H, W = 200, 328
    for bd in (10, 8):
        for S in (8, 16, 32):
            for th in (0, 1, 2):
                for tv in (0, 1, 2):
                    n = int(rng.integers(1, 70))
                    amp = rng.choice([2, 40, (1 << bd) - 1, 5000, 32767], size=n)
                    resi = np.zeros((H, W), np.int16)
                    off = np.zeros(n, np.int32)
                    pr_stride = None
                    blocks = []
                    ys = rng.integers(0, H - S + 1, n); xs = rng.integers(0, W - S + 1, n)
                    # non-overlapping content is not required: TUs only read the plane
                    resi[:] = rng.integers(-40, 41, size=(H, W))
                    for i in range(n):
                        if amp[i] > 40:
                            resi[ys[i]:ys[i] + S, xs[i]:xs[i] + S] = rng.integers(-int(amp[i]), int(amp[i]) + 1, size=(S, S))
                    pr = hp.plane(resi, 0)
                    off = (ys * pr.stride + xs).astype(np.int32)
                    qps = rng.integers(0, 64, size=n); irap = rng.integers(0, 2, size=n)
                    d_off = hp.to_device(off); d_qp = hp.to_device(HotPath.tu_qp(qps, irap, 1))
                    monkeypatch.delenv("VVHIP_TU_KERNEL", raising=False)
                    lev, rec, st = hp.tu_rdo(pr, d_off, n, S, S, d_qp, th, tv, bd, 8)
                    monkeypatch.setenv("VVHIP_TU_KERNEL", "row")
                    lev2, rec2, st2 = hp.tu_rdo(pr, d_off, n, S, S, d_qp, th, tv, bd, 8)
                    monkeypatch.delenv("VVHIP_TU_KERNEL", raising=False)
                    assert torch.equal(lev, lev2) and torch.equal(rec, rec2) and torch.equal(st, st2), (bd, S, th, tv)
                    lev, rec = lev.cpu().numpy().reshape(n, S, S), rec.cpu().numpy().reshape(n, S, S)
                    st = st.cpu().numpy().view(STATS_DTYPE).reshape(n)
                    cur = resi                      # plane content at call time
                    for i in range(n):
                        blk = cur[ys[i]:ys[i] + S, xs[i]:xs[i] + S]
                        if np.abs(blk).max() >= (1 << bd):
                            continue                # beyond the bitDepth contract: covered by the row-kernel comparison above
                        el, er, es = oracle.tu_rdo(blk, int(qps[i]), int(irap[i]), th, tv, bd, 8, 1)
                        assert np.array_equal(lev[i], el), ("lev", bd, S, th, tv, i)
                        assert np.array_equal(rec[i], er), ("rec", bd, S, th, tv, i)
                        assert (int(st["abs_sum"][i]), int(st["last_scan_pos"][i]), int(st["need_rdoq"][i]), int(st["sse"][i])) == \
                            (es["abs_sum"], es["last_scan_pos"], es["need_rdoq"], es["sse"]), (bd, S, th, tv, i)


def synth_pair(rng, h, w, shift=(3, 1), noise=6):
    yy, xx = np.mgrid[0:h + 32, 0:w + 32]
    base = 512 + 180 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 120 * np.sin((xx + yy) / 11.0) + 60 * np.sin(xx / 3.1) * np.sin(yy / 4.3)
    base = base + rng.normal(0, 12, base.shape)
    a = np.clip(base[16:16 + h, 16:16 + w], 0, 1023)
    b = np.clip(base[16 + shift[1]:16 + shift[1] + h, 16 + shift[0]:16 + shift[0] + w] + rng.normal(0, noise, (h, w)), 0, 1023)
    return a.astype(np.int16), b.astype(np.int16)


@pytest.mark.parametrize("cfg", [(416, 240, 4, 16, False), (360, 200, 0, 16, False), (200, 136, 2, 8, False), (640, 360, 4, 16, True)])
def test_mctf_levels_vs_oracle(hip, oracle, cfg):
    w, h, speed, unit, add = cfg
    org, ref = synth_pair(np.random.default_rng(106 + w), h, w, shift=(5, 2))
    exp = oracle.mctf_me(org, ref, 10, unit, speed, add)
    got = hip.mctf_me_levels(org, ref, 10, unit, speed, add)
    for k in range(5):
        if exp[k] is None:
            continue
        for f in ("x", "y", "error"):
            assert np.array_equal(got[k][f], exp[k][f]), (k, f, np.argwhere(got[k][f] != exp[k][f])[:5])
    assert np.array_equal(got[4]["rmsme"], exp[4]["rmsme"]) and np.array_equal(got[4]["overlap"], exp[4]["overlap"])
    full = hip.mctf_me(org, ref, 10, unit, speed, add)[4]
    for f in ("x", "y", "error", "rmsme", "overlap"):
        assert np.array_equal(full[f], exp[4][f]), f


def test_mctf_references_searched_together(hip, oracle):
    """all references of a picture advance through the hierarchy in the same launches (blockIdx.y = reference; more than 8 references go in chunks): every
    reference's field equals the field of a call with that reference alone, and the oracle's for two of them"""
    from vvenc_amd.hotpath import HotPath
    hp = hip.hp
    w, h = 416, 240
    rng = np.random.default_rng(4242)
    org, _ = synth_pair(rng, h, w, shift=(0, 0))
    refs = []
    for k in range(10):
        _, r = synth_pair(np.random.default_rng(4242), h, w, shift=(k % 5 - 2, (k * 3) % 4 - 1), noise=2 + k)
        refs.append(r)
    pc = hp.plane(org, 128)
    prs = [hp.plane(r, 128) for r in refs]
    out, dims = hp.mctf_motion_estimation(pc, prs, 10, 16, 4, False)
    fields = [HotPath.mv_to_numpy(o, dims) for o in out]
    for k in (0, 3, 7, 8, 9):
        alone, _ = hp.mctf_motion_estimation(pc, [prs[k]], 10, 16, 4, False)
        assert np.array_equal(HotPath.mv_to_numpy(alone[0], dims), fields[k]), k
    for k in (1, 9):
        exp = oracle.mctf_me(org, refs[k], 10, 16, 4, False)[4]
        for f in ("x", "y", "error", "rmsme", "overlap"):
            assert np.array_equal(fields[k][f], exp[f]), (k, f)
    assert any(not np.array_equal(fields[0]["x"], fields[k]["x"]) for k in range(1, 10))


def test_mctf_1080p_vs_oracle(hip, oracle):
    """BASELINE config-2 geometry: 1920x1080 10-bit, faster preset (speed 4, unit 16, extra 1/8 level)."""
    org, ref = synth_pair(np.random.default_rng(1080), 1080, 1920, shift=(3, 1), noise=3)
    exp = oracle.mctf_me(org, ref, 10, 16, 4, True)[4]
    got = hip.mctf_me(org, ref, 10, 16, 4, True)[4]
    for f in ("x", "y", "error", "rmsme", "overlap"):
        assert np.array_equal(got[f], exp[f]), (f, np.argwhere(got[f] != exp[f])[:5])


# ---------------------------------------------------------------- size-independent properties at BASELINE sizes ----
def test_properties_1080p_4k(hip):
    """(1) cost surface == candidate-list SAD for the same displacements; (2) SAD(x, x) == 0 and HAD(x, x) == 0;
    (3) SSE/SAD of a plane against itself shifted equals numpy's; (4) fused TU pipeline == unfused pipeline; all on full planes."""
    from vvenc_amd.hotpath import STATS_DTYPE, HotPath, Plane
    hp = hip.hp
    for (W, H) in ((1920, 1080), (3840, 2160)):
        rng = np.random.default_rng(W)
        org = rng.integers(0, 1024, size=(H, W), dtype=np.int16)
        ref = np.roll(org, (2, -3), (0, 1))
        ref = np.clip(ref.astype(np.int32) + rng.integers(-8, 9, size=(H, W)), 0, 1023).astype(np.int16)
        po, pr = hp.plane(org, 80), hp.plane(ref, 80)
        S = 32
        bx, by = np.meshgrid(np.arange(0, W - S + 1, S), np.arange(0, H - S + 1, S))
        bx, by = bx.ravel(), by.ravel()
        nb = bx.size
        oo = (by * po.stride + bx).astype(np.int32)
        d_oo = hp.to_device(oo)
        # (2) identical blocks
        items = np.stack([oo, oo], 1)
        for f in ("SAD", "SSE", "HAD", "HAD_fast"):
            z = hp.dist_batch(f, po, po, hp.to_device(items), nb, S, S, 1 if f == "SAD" else 0)
            assert int(z.abs().max()) == 0
        # (1) + (3): every block, displacement window +-4: surface vs list vs numpy
        R = 4
        surf = hp.sad_surface(po, pr, d_oo, d_oo, nb, S, S, 1, R, R).cpu().numpy().reshape(nb, 2 * R + 1, 2 * R + 1)
        for (dx, dy) in ((0, 0), (-3, 2), (4, -4), (-4, 4)):
            it = np.stack([oo, oo + dy * pr.stride + dx], 1).astype(np.int32)
            lst = hp.dist_batch("SAD", po, pr, hp.to_device(it), nb, S, S, 1).cpu().numpy()
            assert np.array_equal(lst, surf[:, dy + R, dx + R].astype(np.int64)), (W, dx, dy)
            full = hp.dist_batch("SAD", po, pr, hp.to_device(it), nb, S, S, 0).cpu().numpy()
            sse = hp.dist_batch("SSE", po, pr, hp.to_device(it), nb, S, S, 0).cpu().numpy()
            # numpy check on interior blocks (inside the picture after displacement)
            pad_ref = np.pad(ref.astype(np.int64), 80, mode="edge")
            sel = rng.choice(nb, 64, replace=False)
            for b in sel:
                x, y = int(bx[b]), int(by[b])
                a = org[y:y + S, x:x + S].astype(np.int64)
                c = pad_ref[80 + y + dy:80 + y + dy + S, 80 + x + dx:80 + x + dx + S]
                assert int(full[b]) == int(np.abs(a - c).sum())
                assert int(sse[b]) == int(((a - c) ** 2).sum())
        # (4) fused == unfused on residual = org - ref, every 32x32 TU of the frame
        resi = (org.astype(np.int32) - ref.astype(np.int32)).astype(np.int16)
        pres = hp.plane(resi, 0)
        off = (by * pres.stride + bx).astype(np.int32)
        d_off = hp.to_device(off)
        qps = rng.integers(24, 50, size=nb)
        d_qp = hp.to_device(HotPath.tu_qp(qps, 0, 1))
        lev, rec, st = hp.tu_rdo(pres, d_off, nb, S, S, d_qp)
        coef = hp.fwd_transform(pres, d_off, nb, S, S)
        need = hp.need_rdoq(coef, nb, S, S, d_qp)
        lev2, _, s2, last2 = hp.quant(coef, nb, S, S, d_qp, 10, 8, False)
        deq = hp.dequant(lev2, nb, S, S, d_qp)
        prec = Plane(hp.device, W, H, 0)
        hp.inv_transform(deq, nb, S, S, prec, hp.to_device((by * prec.stride + bx).astype(np.int32)))
        st = st.cpu().numpy().view(STATS_DTYPE).reshape(nb)
        assert np.array_equal(lev.cpu().numpy(), lev2.cpu().numpy())
        assert np.array_equal(st["abs_sum"], s2.cpu().numpy()) and np.array_equal(st["last_scan_pos"], last2.cpu().numpy())
        assert np.array_equal(st["need_rdoq"], need.cpu().numpy().astype(np.int32))
        rec_blocks = rec.cpu().numpy().reshape(nb, S, S)
        full_rec = prec.visible().cpu().numpy()
        d = resi.astype(np.int64)
        for b in rng.choice(nb, 64, replace=False):
            x, y = int(bx[b]), int(by[b])
            assert np.array_equal(rec_blocks[b], full_rec[y:y + S, x:x + S])
            assert int(st["sse"][b]) == int(((d[y:y + S, x:x + S] - rec_blocks[b]) ** 2).sum())


def test_dist_multi_equals_batches(hip):
    """vvhip_dist_multi (merged launches) == one vvhip_dist_batch per job, for every function and a mix of mergeable / unmergeable sizes"""
    import torch
    hp = hip.hp
    rng = np.random.default_rng(107)
    org, cur = rand_plane(rng, 256, 384), rand_plane(rng, 256, 384)
    po, pc = hp.plane(org, 0), hp.plane(cur, 0)
    allj, allr = [], []
    for func in ("SAD", "SSE", "HAD", "HAD_fast"):
        jobs, refs = [], []
        for (w, h, ss) in [(8, 8, 0), (16, 16, 1), (4, 4, 0), (32, 32, 1), (64, 64, 1), (16, 8, 0), (128, 128, 1), (8, 8, 0), (24, 24, 0), (32, 32, 0), (64, 64, 0)]:
            if func != "SAD":
                ss = 0
            n = int(rng.integers(1, 300))
            it = np.stack([rng.integers(0, 256 - h, n) * po.stride + rng.integers(0, 384 - w, n),
                           rng.integers(0, 256 - h, n) * pc.stride + rng.integers(0, 384 - w, n)], 1).astype(np.int32)
            d_it = hp.to_device(it)
            out = torch.full((n,), -1, dtype=torch.int64, device=hp.device)
            jobs.append((w, h, ss, n, d_it, out))
            refs.append(hp.dist_batch(func, po, pc, d_it, n, w, h, ss).cpu().numpy())
        hp.dist_multi(func, po, pc, jobs)
        for (w, h, ss, n, _, out), ref in zip(jobs, refs):
            assert np.array_equal(out.cpu().numpy(), ref), (func, w, h, ss)
        allj += [(func,) + j for j in jobs]; allr += refs
    # every function in ONE job table (vvhip_dist_multi_func): SAD and SSE jobs interleaved share launches, so do HAD and HAD_fast
    order = rng.permutation(len(allj))
    mixed = []
    for k in order:
        f, w, h, ss, n, it, _ = allj[k]
        mixed.append((f, w, h, ss, n, it, torch.full((n,), -1, dtype=torch.int64, device=hp.device)))
    hp.dist_multi_func(po, pc, mixed)
    for k, j in zip(order, mixed):
        assert np.array_equal(j[6].cpu().numpy(), allr[k]), ("mixed", j[0], j[1], j[2], j[3])


@pytest.mark.parametrize("case", ["bipred-10", "samples-10-flag", "samples-12", "samples-8-flag"])
def test_dist_multi_hadamard_input_ranges(hip, oracle, case):
    """the merged Hadamard launches against the oracle on every input domain the encoder has: the bi-prediction pattern 2*org - pred (signed, |difference|
    up to 2046; the general packed tile), plain samples with VVHIP_DIST_FLAG_SAMPLES (all-packed tile), and 12-bit samples (hadTile8MultiKernel, 32-bit tile)"""
    import torch
    hp = hip.hp
    rng = np.random.default_rng(311)
    bd = {"bipred-10": 10, "samples-10-flag": 10, "samples-12": 12, "samples-8-flag": 8}[case]
    H, W = 192, 320
    if case == "bipred-10":
        org = rng.integers(-1023, 2047, size=(H, W)).astype(np.int16)
        org[:64, :64] = np.where(rng.integers(0, 2, (64, 64)) == 1, 2046, -1023).astype(np.int16)     # extremes
    else:
        org = rand_plane(rng, H, W, bd)
    cur = rand_plane(rng, H, W, bd)
    if case == "bipred-10":
        cur[:64, :64] = np.where(org[:64, :64] > 0, 0, 1023).astype(np.int16)
    po, pc = hp.plane(org, 0), hp.plane(cur, 0)
    jobs, pos = [], []
    for func in ("HAD", "HAD_fast"):
        for S in (8, 16, 32, 64):
            n = 40
            oy, ox = rng.integers(0, H - S + 1, n), rng.integers(0, W - S + 1, n)
            cy, cx = rng.integers(0, H - S + 1, n), rng.integers(0, W - S + 1, n)
            oy[0], ox[0], cy[0], cx[0] = 0, 0, 0, 0
            it = np.stack([oy * po.stride + ox, cy * pc.stride + cx], 1).astype(np.int32)
            jobs.append((func, S, S, 0, n, hp.to_device(it), torch.full((n,), -1, dtype=torch.int64, device=hp.device)))
            pos.append((oy, ox, cy, cx))
    hp.dist_multi_func(po, pc, hp.make_dist_fjobs(jobs, flags=hp.DIST_FLAG_SAMPLES if case.endswith("flag") else 0), bd)
    for (func, S, _, _, n, _, out), (oy, ox, cy, cx) in zip(jobs, pos):
        got = out.cpu().numpy()
        for k in range(n):
            assert int(got[k]) == oracle.dist(func, (org, int(oy[k]), int(ox[k])), (cur, int(cy[k]), int(cx[k])), S, S, bd), (case, func, S, k)


def test_tu_rdo_multi_equals_batches(hip):
    """vvhip_tu_rdo_multi (square 8/16/32 lists merged into one launch, others alone) == one vvhip_tu_rdo_batch per job"""
    import torch
    from vvenc_amd.hotpath import HotPath, DCT2, DST7, DCT8
    hp = hip.hp
    rng = np.random.default_rng(208)
    resi = rng.integers(-300, 300, size=(192, 320)).astype(np.int16)
    pr = hp.plane(resi, 0)
    jobs, refs = [], []
    for (w, h, th, tv) in [(8, 8, DCT2, DCT2), (32, 32, DCT2, DCT2), (16, 16, DST7, DST7), (16, 16, DCT2, DCT2), (4, 4, DST7, DST7), (32, 32, DST7, DCT8), (64, 64, DCT2, DCT2),
                           (16, 8, DCT2, DCT2), (8, 8, DCT8, DST7)]:
        n = int(rng.integers(1, 200))
        off = (rng.integers(0, 192 - h + 1, n) * pr.stride + rng.integers(0, 320 - w + 1, n)).astype(np.int32)
        d_off = hp.to_device(off)
        d_qp = hp.to_device(HotPath.tu_qp(rng.integers(20, 50, size=n), int(rng.integers(0, 2)), 1))
        lv = torch.full((n * w * h,), -5, dtype=torch.int16, device=hp.device)
        rc = torch.full((n * w * h,), -5, dtype=torch.int16, device=hp.device)
        st = torch.zeros((n, 24), dtype=torch.uint8, device=hp.device)
        jobs.append((w, h, th, tv, n, 8, d_off, d_qp, lv, rc, st))
        a, b, c = hp.tu_rdo(pr, d_off, n, w, h, d_qp, th, tv, 10, 8)
        refs.append((a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()))
    hp.tu_rdo_multi(pr, jobs, 10)
    for (w, h, th, tv, n, _, _, _, lv, rc, st), (a, b, c) in zip(jobs, refs):
        assert np.array_equal(lv.cpu().numpy(), a), ("level", w, h, th, tv)
        assert np.array_equal(rc.cpu().numpy(), b), ("rec", w, h, th, tv)
        assert np.array_equal(st.cpu().numpy(), c), ("stats", w, h, th, tv)


@pytest.mark.parametrize("mix", [{64: 30, 32: 70, 16: 150, 8: 300, 4: 500}, {64: 900, 32: 2200, 16: 700}, {32: 9000, 64: 300, 4: 3000, 8: 4000}], ids=["short-lists", "one-round-budget", "long-lists"])
def test_tu_rdo_multi_strided_launch_shapes(hip, mix):
    """vvhip_tu_rdo_multi_strided on compact residual blocks (pitch = width), the launch shapes of the matrix-core form: every list of a picture in ONE launch with the short lists
    first, a wave budget that makes the launch one resident round (several tiles per wave), and separate launches per kernel instance for long lists — all equal to the
    generic per-job kernel ($VVHIP_TU_GENERIC, read per call)"""
    import torch
    from vvenc_amd.hotpath import HotPath, DCT2, DST7
    hp = hip.hp
    rng = np.random.default_rng(77)
    spec = [(w, n, DCT2) for w, n in mix.items()] + [(16, 40, DST7), (8, 25, DST7)]
    total = sum(n * w * w for w, n, _ in spec)
    pool = hp.to_device(rng.integers(-400, 400, total, dtype=np.int16))

    def run():
        jobs, strides, at = [], [], 0
        for w, n, tr in spec:
            off = hp.to_device((at + np.arange(n, dtype=np.int32) * w * w).astype(np.int32))
            qp = hp.to_device(HotPath.tu_qp(np.random.default_rng(w + n).integers(22, 48, size=n), 1, 1))
            lv = torch.full((n * w * w,), -7, dtype=torch.int16, device=hp.device)
            rc = torch.full((n * w * w,), -7, dtype=torch.int16, device=hp.device)
            st = torch.zeros((n, 24), dtype=torch.uint8, device=hp.device)
            jobs.append((w, w, tr, tr, n, 8, off, qp, lv, rc, st)); strides.append(w)
            at += n * w * w
        hp.tu_rdo_multi_strided(pool, strides, jobs)
        torch.cuda.synchronize()
        return [(j[8].cpu().numpy(), j[9].cpu().numpy(), j[10].cpu().numpy()) for j in jobs]

    got = run()
    os.environ["VVHIP_TU_GENERIC"] = "1"
    try:
        exp = run()
    finally:
        del os.environ["VVHIP_TU_GENERIC"]
    for (w, n, tr), g, e in zip(spec, got, exp):
        for name, a, b in zip(("level", "rec", "stats"), g, e):
            assert np.array_equal(a, b), (name, w, n, tr)


def test_subpel_candidates_vs_oracle(hip, oracle):
    """batched sub-pel prediction blocks and their distortion (what xPatternRefinement scores), every tap mode"""
    from vvenc_amd.hotpath import SUBPEL_DTYPE
    hp = hip.hp
    rng = np.random.default_rng(301)
    yy, xx = np.mgrid[0:160, 0:256]
    ref = np.clip(512 + 200 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 25, (160, 256)), 0, 1023).astype(np.int16)
    org = np.clip(np.roll(ref, (1, 2), (0, 1)).astype(np.int32) + rng.integers(-6, 7, (160, 256)), 0, 1023).astype(np.int16)
    po, pr = hp.plane(org, 0), hp.plane(ref, 0)
    for (w, h) in ((8, 8), (16, 16), (32, 16), (64, 64), (4, 4), (8, 4), (128, 32)):
        n = 23
        it = np.zeros(n, SUBPEL_DTYPE)
        pos = [(int(rng.integers(8, 256 - w - 8)), int(rng.integers(8, 160 - h - 8)), int(rng.integers(8, 256 - w - 8)), int(rng.integers(8, 160 - h - 8)),
                int(rng.integers(0, 16)), int(rng.integers(0, 16))) for _ in range(n)]
        for mode, alt, funcs in ((0, False, ("SAD", "HAD", "HAD_fast", "SSE")), (0, True, ("HAD",)), (1, False, ("HAD_fast",)), (2, True, ("SAD",))):
            q = 8 if alt else 1
            for k, (ox, oy, rx, ry, fx, fy) in enumerate(pos):
                it[k] = (oy * po.stride + ox, ry * pr.stride + rx, fx // q * q % 16, fy // q * q % 16)
            d_it = hp.to_device(it)
            pred = hp.interp_luma_batch(pr, d_it, n, w, h, 10, True, mode, alt).cpu().numpy().reshape(n, h, w)
            exp_pred = []
            for k, (ox, oy, rx, ry, _, _) in enumerate(pos):
                fx, fy = int(it[k]["frac_x"]), int(it[k]["frac_y"])
                e = oracle.if_pred_luma((ref, ry, rx), w, h, fx, fy, True, 10, alt) if mode == 0 else oracle.if_pred_luma_me((ref, ry, rx), w, h, fx, fy, 10, alt, mode)
                assert np.array_equal(pred[k], e), ("pred", w, h, mode, alt, fx, fy)
                exp_pred.append(np.ascontiguousarray(e))
            for func in funcs:
                got = hp.subpel_dist_batch(func, po, pr, d_it, n, w, h, 10, mode, alt).cpu().numpy().view(np.uint64)
                for k, (ox, oy, _, _, _, _) in enumerate(pos):
                    exp = oracle.dist(func, (org, oy, ox), exp_pred[k], w, h, 10, 0)
                    assert int(got[k]) == exp, ("dist", func, w, h, mode, alt, k, int(got[k]), exp)
        # the 14-bit intermediate a bi-prediction average consumes
        pred = hp.interp_luma_batch(pr, d_it, n, w, h, 10, False, 0, False).cpu().numpy().reshape(n, h, w)
        for k, (_, _, rx, ry, _, _) in enumerate(pos):
            fx, fy = int(it[k]["frac_x"]), int(it[k]["frac_y"])
            assert np.array_equal(pred[k], oracle.if_pred_luma((ref, ry, rx), w, h, fx, fy, False, 10, False)), ("bi", w, h, fx, fy)


def test_mctf_apply_1080p_vs_oracle(hip, oracle):
    """SURVEY 8f rank 2 at BASELINE size: device motion estimation -> device bilateral filter (luma + 4:2:0 chroma) equals the oracle fed
    with the same motion fields; all float corners included (exact)"""
    import torch
    from vvenc_amd.workload import synth_frame_pair
    hp = hip.hp
    W, H = 1920, 1080
    cur, ref = synth_frame_pair(W, H, 77)
    refs_y = [ref, np.roll(ref, (1, -2), (0, 1))]
    def yuv(y):
        return (y, np.clip(y[::2, ::2] // 2 + 256, 0, 1023).astype(np.int16), np.clip(1023 - y[::2, ::2] // 3, 0, 1023).astype(np.int16))
    org = yuv(cur)
    refs = [yuv(r) for r in refs_y]
    pc = hp.plane(cur, 128)
    prs = [hp.plane(r, 128) for r in refs_y]
    outs, dims = hp.mctf_motion_estimation(pc, prs, 10, 16, 4, True)
    mvs = [o.cpu().numpy().view(np.uint8).reshape(-1) for o in outs]
    from vvenc_amd.hotpath import MV_DTYPE
    mvs = [m.view(MV_DTYPE) for m in mvs]
    got = hp.mctf_bilateral(org, refs, mvs, [0, 1], 10, 32, 16, True, True, 0.95)
    exp = oracle.mctf_bilateral(org, refs, mvs, [0, 1], 10, 32, 16, True, True, 0.95)
    for c in range(3):
        assert np.array_equal(got[c], exp[c]), (c, int(np.abs(got[c].astype(np.int32) - exp[c]).max()))
        assert not np.array_equal(got[c], org[c])


def test_dmvr_refine_vs_oracle(hip, oracle):
    """batched DMVR refinement search (bilinear predictions, 25 mirrored SADs, error surface) == oracle (== reference rows)"""
    from vvenc_amd.hotpath import DMVR_ITEM_DTYPE, DMVR_RESULT_DTYPE
    hp = hip.hp
    rng = np.random.default_rng(401)
    yy, xx = np.mgrid[0:200, 0:320]
    tex = 512 + 220 * np.sin(xx / 6.0) * np.cos(yy / 5.0) + 80 * np.sin((xx - yy) / 3.0)
    for bd in (10, 8):
        sc = (1 << bd) / 1024.0
        r0 = np.clip((tex + rng.normal(0, 6, tex.shape)) * sc, 0, (1 << bd) - 1).astype(np.int16)
        moved = 0
        for (sx, sy) in ((0, 0), (1, -1), (-2, 1), (2, 2)):
            r1 = np.clip((np.roll(tex, (2 * sy, 2 * sx), (0, 1)) + rng.normal(0, 6, tex.shape)) * sc, 0, (1 << bd) - 1).astype(np.int16)
            p0, p1 = hp.plane(r0, 0), hp.plane(r1, 0)
            for (dx, dy) in ((16, 16), (8, 8), (16, 8), (8, 16)):
                n = 41
                it = np.zeros(n, DMVR_ITEM_DTYPE)
                pos = [(int(rng.integers(8, 320 - dx - 8)), int(rng.integers(8, 200 - dy - 8))) for _ in range(n)]
                it["ref0_off"] = [y * p0.stride + x for (x, y) in pos]
                it["ref1_off"] = [y * p1.stride + x for (x, y) in pos]
                for f in ("frac0_x", "frac0_y", "frac1_x", "frac1_y"):
                    it[f] = rng.integers(0, 16, n)
                it[:5]["frac0_x"] = 0; it[:3]["frac0_y"] = 0; it[2:7]["frac1_x"] = 0; it[:8]["frac1_y"] = 0
                res = hp.dmvr_refine_batch(p0, p1, hp.to_device(it), n, dx, dy, bd).cpu().numpy().reshape(-1).view(DMVR_RESULT_DTYPE)
                for k, (x, y) in enumerate(pos):
                    exp = oracle.dmvr_refine((r0, y, x), (r1, y, x), (int(it[k]["frac0_x"]), int(it[k]["frac0_y"])), (int(it[k]["frac1_x"]), int(it[k]["frac1_y"])), dx, dy, bd)
                    got = (int(res[k]["mvd_x"]), int(res[k]["mvd_y"]), int(res[k]["min_cost"]))
                    assert got == exp, (bd, sx, sy, dx, dy, k, got, exp)
                    moved += got[0] != 0 or got[1] != 0
        assert moved > 100


def test_subpel_refine_equals_candidate_list(hip):
    """vvhip_subpel_refine_batch (LDS window, shared horizontal passes) == vvhip_subpel_dist_batch on the expanded candidate list, which is
    pinned to the oracle/reference above; half-sample and quarter-sample stages, every function and tap mode, fractional base vectors"""
    from vvenc_amd.hotpath import SUBPEL_DTYPE
    hp = hip.hp
    rng = np.random.default_rng(302)
    yy, xx = np.mgrid[0:200, 0:320]
    ref = np.clip(512 + 200 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 25, (200, 320)), 0, 1023).astype(np.int16)
    org = np.clip(np.roll(ref, (1, 2), (0, 1)).astype(np.int32) + rng.integers(-6, 7, (200, 320)), 0, 1023).astype(np.int16)
    po, pr = hp.plane(org, 0), hp.plane(ref, 0)
    half = [(-8, 0), (8, 0), (0, -8), (0, 8), (-8, -8), (8, -8), (-8, 8), (8, 8)]
    quarter = [(-4, 0), (4, 0), (0, -4), (0, 4), (-4, -4), (4, -4), (-4, 4), (4, 4), (0, 0)]
    wide = [(-16, 16), (16, -16), (3, -13), (-11, 7)]
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (32, 16)):
        nb = 19
        bases = np.zeros(nb, SUBPEL_DTYPE)
        for k in range(nb):
            ox, oy = int(rng.integers(8, 320 - w - 8)), int(rng.integers(8, 200 - h - 8))
            rx, ry = int(rng.integers(8, 320 - w - 8)), int(rng.integers(8, 200 - h - 8))
            bases[k] = (oy * po.stride + ox, ry * pr.stride + rx, int(rng.integers(0, 4)) * 4, int(rng.integers(0, 4)) * 4)
        d_b = hp.to_device(bases)
        for offs, mode, alt, funcs in ((half, 0, False, ("HAD", "HAD_fast", "SAD", "SSE")), (quarter, 0, False, ("HAD_fast",)), (half, 1, False, ("HAD",)),
                                       (quarter, 2, True, ("SAD",)), (wide, 0, True, ("SSE", "HAD_fast"))):
            cand = np.zeros(nb * len(offs), SUBPEL_DTYPE)
            for b in range(nb):
                for k, (dx, dy) in enumerate(offs):
                    tx, ty = int(bases[b]["frac_x"]) + dx, int(bases[b]["frac_y"]) + dy
                    cand[b * len(offs) + k] = (bases[b]["org_off"], int(bases[b]["ref_off"]) + (ty >> 4) * pr.stride + (tx >> 4), tx & 15, ty & 15)
            d_c = hp.to_device(cand)
            for func in funcs:
                got = hp.subpel_refine_batch(func, po, pr, d_b, nb, offs, w, h, 10, mode, alt).cpu().numpy()
                exp = hp.subpel_dist_batch(func, po, pr, d_c, cand.size, w, h, 10, mode, alt).cpu().numpy()
                assert np.array_equal(got, exp), (w, h, func, mode, alt, np.argwhere(got != exp)[:4].ravel(), got[:4], exp[:4])


def test_bad_arguments_are_reported_not_crashed(hip):
    """error convention of the C ABI: negative code + message from vvhip_last_error, the context stays usable"""
    import ctypes as C
    import torch
    from vvenc_amd.lib import VVHipError
    hp = hip.hp
    plane = hp.plane(np.zeros((64, 64), np.int16), 0)
    items = hp.to_device(np.zeros((4, 2), np.int32))
    out = torch.zeros(4, dtype=torch.int64, device=hp.device)
    plane8 = hp.plane(np.zeros((64, 64), np.int16), 8)
    cls8 = torch.zeros((16, 16, 2), dtype=torch.uint8, device=hp.device)
    bad = [
        lambda: hp.dist_batch("SAD", plane, plane, items, 4, 3, 8),                                   # width 3
        lambda: hp.dist_batch("HAD", plane, plane, items, 4, 256, 256),                               # block larger than 128
        lambda: hp.sad_x5_batch(plane, plane, items, 1, 32, 8),                                       # X5 exists for widths 8 and 16 only
        lambda: hp.fwd_transform(plane, items[:, 0].contiguous(), 4, 128, 8),                         # no 128-point transform
        lambda: hp.fwd_transform(plane, items[:, 0].contiguous(), 4, 64, 64, 2, 2),                   # DST-7 stops at 32
        lambda: hp.interp_luma_batch(plane, items, 1, 256, 8),                                        # block too large
        lambda: hp.subpel_refine_batch("HAD", plane, plane, items, 1, [(40, 0)], 8, 8),               # offset beyond one sample
        lambda: hp.dmvr_refine_batch(plane, plane, items, 1, 32, 16),                                 # DMVR sub-blocks are at most 16x16
        lambda: hp.mctf_apply_plane(plane, [plane], [items], 1, 0, [1.0], 1.0, 1.0, 10, 64),          # unit 64 unsupported
        lambda: hp.alf_classify(hp.plane(np.zeros((66, 64), np.int16), 8)),                             # height not a multiple of 4
        lambda: hp.alf_classify(plane8, 10, 96, 92),                                                  # CTU height not a power of two
        lambda: hp.alf_stats_plane(plane, plane8, 256, 7, cls8),                                      # statistics unit larger than 128
        lambda: hp.alf_stats_plane(plane, plane8, 64, 9, cls8),                                       # filter length 9
        lambda: hp.alf_stats_plane(plane, plane8, 128, 7, cls8, ctu_in_unit=48),                      # unit not a multiple of the CTU
        lambda: hp.ccalf_stats_plane(plane, plane, plane8, 128),                                      # chroma CTU 128 = luma 256
        lambda: hp.alf_filter_plane(plane8, plane, 64, 10, 7, items, None, items),                    # 7x7 needs the classes
        lambda: hp.alf_filter_plane(plane8, plane, 64, 10, 5, items, None, items, cls8),              # 5x5 has one class
        lambda: hp.alf_filter_plane(plane8, plane, 64, 14, 5, items, None, items),                    # bit depth
        lambda: hp.ccalf_filter_plane(plane, plane8, 32, 10, items, items, 96, 92),                   # CTU height not a power of two
    ]
    for k, f in enumerate(bad):
        with pytest.raises(VVHipError) as e:
            f()
        assert "vvhip" in str(e.value) or "vvenc_hip" in str(e.value), (k, str(e.value))
    # still alive
    good = hp.dist_batch("SAD", plane, plane, items, 4, 8, 8, out=out)
    assert int(good.sum().item()) == 0


def _alf_pictures(rng, h, w, smooth):
    yy, xx = np.mgrid[0:h, 0:w]
    if smooth:
        base = 512 + 90 * np.sin(xx / 41.0) * np.cos(yy / 33.0) + 40 * np.sin((xx - yy) / 17.0) + rng.normal(0, 1.5, (h, w)) + 25 * (((xx // 24) + (yy // 40)) % 2)
    else:
        base = 512 + 200 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 80 * np.sin((xx + 2 * yy) / 5.0) + rng.normal(0, 20, (h, w))
    rec = np.clip(base, 0, 1023).astype(np.int16)
    org = np.clip(rec.astype(np.int32) + rng.integers(-12, 13, (h, w)), 0, 1023).astype(np.int16)
    return org, rec


@pytest.mark.parametrize("cfg", [(272, 400, 128, False), (264, 392, 128, True), (136, 200, 64, True), (64, 64, 32, False)])
def test_alf_classification_and_statistics_vs_oracle(hip, oracle, cfg):
    """ALF classification (integer) and covariance statistics (float sums in the reference's order) — bit-exact, luma 7x7 with
    25 classes and chroma 5x5, partial CTUs, virtual-boundary rows"""
    h, w, ctu, smooth = cfg
    hp = hip.hp
    org, rec = _alf_pictures(np.random.default_rng(700 + h), h, w, smooth)
    vbh, vbp = ctu, ctu - 4
    prec, porg = hp.plane(rec, 8), hp.plane(org, 0)
    d_cls = hp.alf_classify(prec, 10, vbh, vbp)
    exp_cls = oracle.alf_classify(rec, 10, vbh, vbp)
    assert np.array_equal(d_cls.cpu().numpy(), exp_cls)
    got = hp.alf_stats_plane(porg, prec, ctu, 7, d_cls, vbh, vbp).cpu().numpy()
    exp = oracle.alf_stats_plane(org, rec, ctu, 7, exp_cls, vbh, vbp)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), ("luma", np.abs(got - exp).max())
    c_org, c_rec = np.ascontiguousarray(org[::2, ::2]), np.ascontiguousarray(rec[::2, ::2])
    if c_rec.shape[0] % 4 == 0 and c_rec.shape[1] % 4 == 0:
        pcr, pco = hp.plane(c_rec, 8), hp.plane(c_org, 0)
        got = hp.alf_stats_plane(pco, pcr, ctu // 2, 5, None, ctu // 2, ctu // 2 - 2).cpu().numpy()
        exp = oracle.alf_stats_plane(c_org, c_rec, ctu // 2, 5, None, ctu // 2, ctu // 2 - 2)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), ("chroma", np.abs(got - exp).max())
    # chains continued from given records (statistics units spanning several CTUs): a second picture accumulated onto the first one's records
    org2, rec2 = _alf_pictures(np.random.default_rng(900 + h), h, w, not smooth)
    p2r, p2o = hp.plane(rec2, 8), hp.plane(org2, 0)
    first = hp.alf_stats_plane(porg, prec, ctu, 7, d_cls, vbh, vbp)
    got = hp.alf_stats_plane(p2o, p2r, ctu, 7, d_cls, vbh, vbp, init=first, out=first).cpu().numpy()
    exp = oracle.alf_stats_plane(org2, rec2, ctu, 7, exp_cls, vbh, vbp, init=oracle.alf_stats_plane(org, rec, ctu, 7, exp_cls, vbh, vbp))
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), "continued chains"
    # statistics units of 2x2 CTUs (alfUnitSize 128 over 64x64 CTUs): chains through the unit's CTUs in raster order
    if ctu <= 64:
        got = hp.alf_stats_plane(porg, prec, 2 * ctu, 7, d_cls, vbh, vbp, ctu_in_unit=ctu).cpu().numpy()
        exp = oracle.alf_stats_plane(org, rec, 2 * ctu, 7, exp_cls, vbh, vbp, ctu_in_unit=ctu)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), "units of 2x2 CTUs"
    # blocks marked unused are skipped
    cls2 = exp_cls.copy(); cls2[::3, 1::2] = 255
    got = hp.alf_stats_plane(porg, prec, ctu, 7, hp.to_device(cls2), vbh, vbp).cpu().numpy()
    exp = oracle.alf_stats_plane(org, rec, ctu, 7, cls2, vbh, vbp)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), "unused blocks"


@pytest.mark.parametrize("cfg", [(272, 400, 64), (136, 200, 32), (128, 128, 64), (64, 96, 16)])
def test_ccalf_statistics_vs_oracle(hip, oracle, cfg):
    """CC-ALF covariance records (getBlkStatsCcAlf): 7 luma differences vs org - ALF-filtered chroma, float sums in the reference's order; partial CTUs, virtual
    boundary rows, the boundary-free last CTU row, and continued chains"""
    h, w, ctu_c = cfg
    hp = hip.hp
    rng = np.random.default_rng(1000 + h)
    _, rec = _alf_pictures(rng, h, w, False)
    slf = np.clip(500 + 0.3 * (rec[::2, ::2].astype(np.float64) - 512) + rng.normal(0, 4, (h // 2, w // 2)), 0, 1023).astype(np.int16)
    org = np.clip(slf.astype(np.int32) + rng.integers(-9, 10, slf.shape), 0, 1023).astype(np.int16)
    prec, pslf, porg = hp.plane(rec, 8), hp.plane(slf, 0), hp.plane(org, 0)
    got = hp.ccalf_stats_plane(porg, pslf, prec, ctu_c, 2 * ctu_c, 2 * ctu_c - 4)
    exp = oracle.ccalf_stats_plane(org, slf, rec, ctu_c, 2 * ctu_c, 2 * ctu_c - 4)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), exp.view(np.uint32)), np.abs(got.cpu().numpy() - exp).max()
    got2 = hp.ccalf_stats_plane(porg, pslf, prec, ctu_c, 2 * ctu_c, 2 * ctu_c - 4, init=got, out=got).cpu().numpy()
    exp2 = oracle.ccalf_stats_plane(org, slf, rec, ctu_c, 2 * ctu_c, 2 * ctu_c - 4, init=exp)
    assert np.array_equal(got2.view(np.uint32), exp2.view(np.uint32)), "continued chains"


def _alf_filter_sets(rng, num_sets, num_classes, bd, nonlinear):
    coeff = rng.integers(-40, 41, (num_sets, num_classes, 13)).astype(np.int16)
    coeff[..., 12] = 0
    clips = np.array([1 << bd, 1 << (bd - 3), 1 << (bd - 5), 1 << max(1, bd - 7)], np.int16)
    clip = clips[rng.integers(0, 4, (num_sets, num_classes, 13))] if nonlinear else np.full((num_sets, num_classes, 13), 1 << bd, np.int16)
    return coeff, np.ascontiguousarray(clip, np.int16)


@pytest.mark.parametrize("cfg", [(272, 400, 128, False, 10), (264, 1048, 64, True, 10), (136, 200, 64, True, 8), (64, 64, 32, False, 10), (20, 12, 16, False, 10), (72, 104, 32, False, 12)])
def test_alf_filtering_vs_oracle(hip, oracle, cfg):
    """SURVEY 8f rank 4, apply side: filterBlk 7x7 (25 classes, 4 transposes) and 5x5 over the enabled CTUs of a plane — linear entry (no clipping values), non-linear
    entry, rows folded at the virtual boundary and the 3-bit larger shift next to it, disabled CTUs untouched, partial tiles / CTUs, strided unaligned destination"""
    h, w, ctu, smooth, bd = cfg
    hp = hip.hp
    rng = np.random.default_rng(1500 + h)
    _, rec = _alf_pictures(rng, h, w, smooth)
    if bd == 8:
        rec = (rec >> 2).astype(np.int16)
    if bd == 12:
        rec = ((rec.astype(np.int32) << 2) + rng.integers(0, 4, rec.shape)).astype(np.int16)
    cls = oracle.alf_classify(rec, bd, ctu, ctu - 4)
    nctu = -(-h // ctu) * -(-w // ctu)
    for nonlinear in (False, True):
        coeff, clip = _alf_filter_sets(rng, 3, 25, bd, nonlinear)
        ctu_set = rng.integers(-1, 3, nctu).astype(np.int16)
        exp = oracle.alf_filter_plane(rec, ctu, bd, 7, coeff, clip, ctu_set, cls, None, ctu, ctu - 4)
        got = hip.alf_filter_plane(rec, ctu, bd, 7, coeff, clip, ctu_set, cls, None, ctu, ctu - 4, linear_entry=not nonlinear)
        assert np.array_equal(got, exp), ("luma", nonlinear, np.argwhere(got != exp)[:4])
        if not nonlinear:                                   # the non-linear entry with clipping values that never bite gives the same picture
            assert np.array_equal(hip.alf_filter_plane(rec, ctu, bd, 7, coeff, clip, ctu_set, cls, None, ctu, ctu - 4), exp)
        c_rec = np.ascontiguousarray(rec[::2, ::2])
        if c_rec.shape[0] % 4 == 0 and c_rec.shape[1] % 4 == 0:
            coeff, clip = _alf_filter_sets(rng, 4, 1, bd, nonlinear)
            ctu_set = rng.integers(-1, 4, nctu).astype(np.int16)
            exp = oracle.alf_filter_plane(c_rec, ctu // 2, bd, 5, coeff, clip, ctu_set, None, None, ctu // 2, ctu // 2 - 2)
            got = hip.alf_filter_plane(c_rec, ctu // 2, bd, 5, coeff, clip, ctu_set, None, None, ctu // 2, ctu // 2 - 2, linear_entry=not nonlinear)
            assert np.array_equal(got, exp), ("chroma", nonlinear)
    # destination with an odd stride and a 2-byte aligned origin: the 16-bit store path
    coeff, clip = _alf_filter_sets(rng, 2, 25, bd, True)
    ctu_set = rng.integers(-1, 2, nctu).astype(np.int16)
    import torch
    from vvenc_amd.hotpath import Plane
    dst = Plane(hp.device, w, h, 1, stride=w + 3)
    dst.storage[1:1 + h, 1:1 + w] = torch.from_numpy(rec).to(hp.device)
    hp.alf_filter_plane(hp.plane(rec, 4), dst, ctu, bd, 7, hp.to_device(coeff), hp.to_device(clip), hp.to_device(ctu_set), hp.to_device(cls), ctu, ctu - 4)
    assert np.array_equal(dst.visible().cpu().numpy(), oracle.alf_filter_plane(rec, ctu, bd, 7, coeff, clip, ctu_set, cls, None, ctu, ctu - 4))


@pytest.mark.parametrize("cfg", [(272, 400, 64, 10), (136, 200, 32, 8), (128, 128, 64, 10), (64, 96, 16, 10)])
def test_ccalf_filtering_vs_oracle(hip, oracle, cfg):
    """filterBlkCcAlf over a chroma plane: 7 luma differences, per-CTU filter choice, rows next to the luma virtual boundary"""
    h, w, ctu_c, bd = cfg
    rng = np.random.default_rng(1600 + h)
    _, rec = _alf_pictures(rng, h, w, False)
    if bd == 8:
        rec = (rec >> 2).astype(np.int16)
    chroma = np.clip((1 << (bd - 1)) + rng.normal(0, 1 << (bd - 2), (h // 2, w // 2)), 0, (1 << bd) - 1).astype(np.int16)
    coeff = np.zeros((4, 8), np.int16)
    coeff[:, :7] = np.array([0, 1, 2, 4, 8, 16, 32, 64], np.int16)[rng.integers(0, 8, (4, 7))] * rng.choice([-1, 1], (4, 7))
    nctu = -(-(h // 2) // ctu_c) * -(-(w // 2) // ctu_c)
    ctu_filter = rng.integers(0, 5, nctu).astype(np.uint8)
    exp = oracle.ccalf_filter_plane(chroma, rec, ctu_c, bd, coeff, ctu_filter, 2 * ctu_c, 2 * ctu_c - 4)
    got = hip.ccalf_filter_plane(chroma, rec, ctu_c, bd, coeff, ctu_filter, 2 * ctu_c, 2 * ctu_c - 4)
    assert np.array_equal(got, exp)
    assert not np.array_equal(exp, chroma)


def test_launch_graph_replays_batch_calls(hip, oracle):
    """vvhip_graph_*: a fixed sequence of batch calls recorded once on the context's stream and replayed — results equal the direct calls', also after the inputs changed"""
    import torch
    hp = hip.hp
    rng = np.random.default_rng(77)
    org = rng.integers(0, 1024, (128, 192), dtype=np.int16)
    cur = np.clip(np.roll(org, (1, -2), (0, 1)).astype(np.int32) + rng.integers(-6, 7, org.shape), 0, 1023).astype(np.int16)
    po, pc = hp.plane(org, 16), hp.plane(cur, 16)
    bx, by = np.meshgrid(np.arange(0, 192 - 16 + 1, 16), np.arange(0, 128 - 16 + 1, 16))
    items = np.stack([(by.ravel() * po.stride + bx.ravel()), ((by.ravel() + 1) * pc.stride + bx.ravel() - 2)], 1).astype(np.int32)
    d_items = hp.to_device(items)
    n = items.shape[0]
    out_sad = torch.zeros(n, dtype=torch.int64, device=hp.device)
    out_had = torch.zeros(n, dtype=torch.int64, device=hp.device)

    def calls():
        hp.dist_batch("SAD", po, pc, d_items, n, 16, 16, out=out_sad)
        hp.dist_batch("HAD", po, pc, d_items, n, 16, 16, out=out_had)

    calls()
    torch.cuda.synchronize()
    exp_sad, exp_had = out_sad.clone(), out_had.clone()
    g = hp.graph_capture(calls)
    out_sad.zero_(); out_had.zero_()
    hp.graph_launch(g)
    torch.cuda.synchronize()
    assert torch.equal(out_sad, exp_sad) and torch.equal(out_had, exp_had)
    # same graph, new picture content in the same buffers
    cur2 = np.clip(cur.astype(np.int32) + rng.integers(-20, 21, cur.shape), 0, 1023).astype(np.int16)
    pc.storage[pc.pad:pc.pad + pc.height, pc.pad:pc.pad + pc.width] = torch.from_numpy(cur2).to(hp.device)
    hp.graph_launch(g)
    torch.cuda.synchronize()
    got = out_sad.cpu().numpy()
    o32, c32 = org.astype(np.int64), cur2.astype(np.int64)
    ref = np.array([np.abs(o32[y:y + 16, x:x + 16] - c32[y + 1:y + 17, x - 2:x + 14]).sum() if x >= 2 and y + 17 <= 128 else -1 for x, y in zip(bx.ravel(), by.ravel())])
    ok = ref >= 0
    assert np.array_equal(got[ok], ref[ok])
    hp.graph_destroy(g)


def test_alf_4k_vs_oracle(hip, oracle):
    """SURVEY 8f rank 4 at BASELINE's largest picture (3840x2160, CTU 128): classes, luma covariance records (float bit patterns) and the filtered luma plane against the
    oracle; size-independent property on top: an all-zero filter set returns the unfiltered plane"""
    hp = hip.hp
    h, w, ctu = 2160, 3840, 128
    org, rec = _alf_pictures(np.random.default_rng(2160), h, w, False)
    prec, porg = hp.plane(rec, 8), hp.plane(org, 0)
    d_cls = hp.alf_classify(prec, 10, ctu, ctu - 4)
    exp_cls = oracle.alf_classify(rec, 10, ctu, ctu - 4)
    assert np.array_equal(d_cls.cpu().numpy(), exp_cls)
    got = hp.alf_stats_plane(porg, prec, ctu, 7, d_cls, ctu, ctu - 4).cpu().numpy()
    exp = oracle.alf_stats_plane(org, rec, ctu, 7, exp_cls, ctu, ctu - 4)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    rng = np.random.default_rng(2161)
    coeff, clip = _alf_filter_sets(rng, 3, 25, 10, True)
    nctu = -(-h // ctu) * -(-w // ctu)
    ctu_set = rng.integers(-1, 3, nctu).astype(np.int16)
    exp = oracle.alf_filter_plane(rec, ctu, 10, 7, coeff, clip, ctu_set, exp_cls, None, ctu, ctu - 4)
    got = hip.alf_filter_plane(rec, ctu, 10, 7, coeff, clip, ctu_set, exp_cls, None, ctu, ctu - 4)
    assert np.array_equal(got, exp)
    zero = np.zeros_like(coeff)
    assert np.array_equal(hip.alf_filter_plane(rec, ctu, 10, 7, zero, clip, np.zeros(nctu, np.int16), exp_cls, None, ctu, ctu - 4), rec)


@pytest.mark.parametrize("aligned", [True, False])
def test_dist_multi_func_tiled_equals_row_major(hip, oracle, aligned):
    """the 8x8-tiled plane copies (one cache line = one 8x8 tile): vvhip_dist_multi_func_tiled == vvhip_dist_multi_func == oracle for every function and size, original blocks
    on the tile grid (a picture tiling) or anywhere, candidates anywhere incl. inside the margins; general and VVHIP_DIST_FLAG_SAMPLES Hadamard forms"""
    import torch
    hp = hip.hp
    rng = np.random.default_rng(911 + int(aligned))
    H, W, pad = 136, 200, 24
    org, cur = rand_plane(rng, H, W), rand_plane(rng, H, W)
    po, pc = hp.plane(org, pad), hp.plane(cur, pad)
    to, tc = hp.tile_plane(po), hp.tile_plane(pc)
    opad = np.pad(org, pad, mode="edge")
    cpad = np.pad(cur, pad, mode="edge")
    for flags in (0, hp.DIST_FLAG_SAMPLES):
        jobs, pos = [], []
        for func in ("SAD", "SSE", "HAD", "HAD_fast"):
            for (S, ss) in ((8, 0), (16, 1), (32, 1), (64, 1), (16, 0)):
                if func != "SAD":
                    ss = 0
                n = 37
                ox = rng.integers(0, (W - S) // 8 + 1, n) * 8 if aligned else rng.integers(-pad, W + pad - S + 1, n)
                oy = rng.integers(0, (H - S) // 8 + 1, n) * 8 if aligned else rng.integers(-pad, H + pad - S + 1, n)
                cx, cy = rng.integers(-pad, W + pad - S + 1, n), rng.integers(-pad, H + pad - S + 1, n)
                it = np.stack([oy * po.stride + ox, cy * pc.stride + cx], 1).astype(np.int32)
                jobs.append((func, S, S, ss, n, hp.to_device(it), torch.full((n,), -1, dtype=torch.int64, device=hp.device)))
                pos.append((ox, oy, cx, cy))
        plain = [(f, w, h, ss, n, it, torch.full((n,), -2, dtype=torch.int64, device=hp.device)) for (f, w, h, ss, n, it, _) in jobs]
        hp.dist_multi_func_tiled(po, pc, to, tc, hp.make_dist_fjobs(jobs, flags=flags), 10)
        hp.dist_multi_func(po, pc, hp.make_dist_fjobs(plain, flags=flags), 10)
        # + the one-sample-shifted copy of the reference plane (candidates at odd sample addresses read it with dword-aligned loads): with and without the tiled copies,
        # packed (10-bit) and 32-bit (12-bit call) Hadamard forms
        sh = hp.shift_plane(pc)
        # the three copies in one launch (vvhip_planes_derive) == the single-copy entry points
        to2, tc2, sh2 = torch.full_like(to, -7), torch.full_like(tc, -7), torch.full_like(sh, -7)
        hp.planes_derive(po, pc, to2, tc2, sh2)
        n_t = (po.storage.shape[0] + 7) // 8 * ((po.stride + 7) // 8) * 64
        assert torch.equal(to2[:n_t], to[:n_t]) and torch.equal(tc2[:n_t], tc[:n_t]) and torch.equal(sh2, sh)
        assert np.array_equal(sh.cpu().numpy().ravel()[:-1], pc.storage.cpu().numpy().ravel()[1:]) and int(sh.cpu().numpy().ravel()[-1]) == 0
        for (t_o, t_c, bd) in ((to, tc, 10), (None, None, 10), (None, None, 12)):
            shifted = [(f, w, h, ss, n, it, torch.full((n,), -3, dtype=torch.int64, device=hp.device)) for (f, w, h, ss, n, it, _) in jobs]
            hp.dist_multi_func_tiled(po, pc, t_o, t_c, hp.make_dist_fjobs(shifted, flags=flags), bd, cur_shift=sh)
            for (func, S, _, ss, n, _, out2), (_, _, _, _, _, _, out3) in zip(plain, shifted):
                assert np.array_equal(out2.cpu().numpy(), out3.cpu().numpy()), (aligned, flags, func, S, bd, t_o is None)
        for (func, S, _, ss, n, _, out), (_, _, _, _, _, _, out2), (ox, oy, cx, cy) in zip(jobs, plain, pos):
            got, ref = out.cpu().numpy(), out2.cpu().numpy()
            assert np.array_equal(got, ref), (aligned, flags, func, S)
            for k in (0, 7, n - 1):
                assert int(got[k]) == oracle.dist(func, (opad, int(oy[k]) + pad, int(ox[k]) + pad), (cpad, int(cy[k]) + pad, int(cx[k]) + pad), S, S, 10, ss), (func, S, k)


def test_transfer_and_multi_device_entry_points(hip):
    """the round-2 C-ABI additions: pinned host areas, in-place pinning, strided copies, device enumeration and the device-to-device picture copy (same device here)"""
    import ctypes as C
    L, ctx = hip.hp.L, hip.hp.ctx
    assert L.vvhip_device_count() >= 1 and L.vvhip_get_device(ctx) == 0 and L.vvhip_make_current(ctx) == 0
    rng = np.random.default_rng(5)
    src = rng.integers(-2000, 2000, size=(96, 200)).astype(np.int16)
    d = C.c_void_p(); d2 = C.c_void_p()
    assert L.vvhip_malloc(ctx, C.byref(d), src.nbytes) == 0 and L.vvhip_malloc(ctx, C.byref(d2), src.nbytes) == 0
    # strided upload of a window, device-to-device copy, strided download
    win = src[8:72, 16:144]
    assert L.vvhip_upload_2d(ctx, d, 128 * 2, src.ctypes.data + (8 * 200 + 16) * 2, 200 * 2, 128 * 2, 64) == 0
    assert L.vvhip_copy_peer(ctx, d2, ctx, d, 128 * 2 * 64) == 0
    back = np.zeros((64, 160), np.int16)
    assert L.vvhip_download_2d(ctx, back.ctypes.data, 160 * 2, d2, 128 * 2, 128 * 2, 64) == 0
    assert np.array_equal(back[:, :128], win) and not back[:, 128:].any()
    # pinned area owned by the library's caller + asynchronous download
    h = C.c_void_p()
    assert L.vvhip_host_alloc(ctx, C.byref(h), src.nbytes) == 0
    assert L.vvhip_upload(ctx, d, src.ctypes.data, src.nbytes) == 0
    assert L.vvhip_download_async(ctx, h, d, src.nbytes) == 0 and L.vvhip_sync(ctx) == 0
    assert np.array_equal(np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_int16)), shape=src.shape), src)
    assert L.vvhip_host_free(ctx, h) == 0
    # in-place pinning of a caller buffer (whole pages), upload from inside it, unpin
    big = np.zeros(1 << 20, np.int16)
    big[:] = rng.integers(0, 1024, big.size)
    a = (big.ctypes.data + 4095) & ~4095
    n = ((big.ctypes.data + big.nbytes) & ~4095) - a
    assert L.vvhip_host_register(ctx, C.c_void_p(a), n) == 0
    assert L.vvhip_host_register(ctx, C.c_void_p(a), n) == 0          # registering again is not an error
    d3 = C.c_void_p()
    assert L.vvhip_malloc(ctx, C.byref(d3), n) == 0
    assert L.vvhip_upload(ctx, d3, C.c_void_p(a), n) == 0
    chk = np.zeros(n // 2, np.int16)
    assert L.vvhip_download(ctx, chk.ctypes.data, d3, n) == 0
    assert np.array_equal(chk, big[(a - big.ctypes.data) // 2:(a - big.ctypes.data) // 2 + n // 2])
    assert L.vvhip_host_unregister(ctx, C.c_void_p(a)) == 0
    for p in (d, d2, d3):
        assert L.vvhip_free(ctx, p) == 0


def test_bench_multi_rank_path_on_one_device(tmp_path):
    """bench.py's N > 1 path (different pictures of one sequence per rank, PictureExchange broadcasts on their own stream inside the timed region, three compute streams) with
    two ranks sharing the one GPU and gloo as the collective back-end ($VVHIP_SHARE_DEVICE / $VVHIP_DIST_BACKEND): the code the driver runs under RCCL on 2/4/8 GPUs"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VVHIP_SHARE_DEVICE="1", VVHIP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--width", "640", "--height", "384", "--exchange-every", "1", "--detail", "bench_detail_two_ranks.json"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith('{"metric"') and len(last) <= 8192, (len(last), last[:200])          # the driver-facing line is the LAST line of stdout and compact
    d = json.loads(last)
    # one broadcast per step inside the timed region (--exchange-every 1); over gloo no fabric figure is printed; the two other N-GPU points are reported next to `value`
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["exchange"]["pictures"] == 6 and d["exchange"]["backend"] == "gloo" and "exchange_ms_per_picture" not in d["exchange"], d
    assert d["no_exchange"]["value"] > 0 and d["exchange_per_gop_cycle"]["value"] > 0 and d["exchange_per_gop_cycle"]["every_steps"] == 32, d
    assert d["exchange_every_reference"]["value"] > 0 and d["exchange_every_reference"]["every_steps"] == 2, d
    assert d["parity"]["status"] == "bit-exact", d["parity"]
    # round 6: a broadcast WITH A CONSUMER — the steps again with leg C, a filtered picture's reference originals arriving through the exchange from the
    # neighbouring rank; the fields computed from the received planes equal the fields from the rank's own copies, and a corrupted slot is noticed (self-test on every consumer)
    assert d["value_with_mctf"] > 0 and d["parity_exchange"]["status"] == "bit-exact", d.get("parity_exchange")
    assert d["parity_exchange"]["jobs_checked_over_ranks"] >= 2 and d["parity_exchange"]["ranks_that_detected_a_corrupted_slot"] == 2, d["parity_exchange"]
    assert d["mctf_exchanges"]["exchanges_in_the_timed_steps"] >= 2 and d["mctf_exchanges"]["bytes_per_exchange"] > 0, d["mctf_exchanges"]
    # the metric's N-GPU form: one encoder instance per rank on its own GPU (here: both on the one GPU), GOP chunks of one sequence, CPU kernels vs --SIMD=HIP
    inst = d["e2e_instances"]
    assert inst["instances"] == 2 and inst["chunk_bitstreams_identical"] is True and inst["cpu_fps_aggregate"] > 0 and inst["hip_fps_aggregate"] > 0, inst
    full = json.load(open(os.path.join(root, d["detail"])))                                     # everything else of the run: next to the line
    assert full["exchange"]["bytes_per_picture"] > 0 and full["value"] == pytest.approx(d["value"], rel=1e-4)
    assert len({p["md5"] for p in full["e2e_instances"]["per_instance"]}) == 2, full["e2e_instances"]          # two different chunks
