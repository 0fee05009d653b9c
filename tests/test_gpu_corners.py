"""-m gpu: the corners of the round-4 arithmetic re-formulations, ON THE DEVICE (tests/test_me_kernel_algebra.py pins the algebra in numpy; uniform random data never reaches these):

  * the packed 16-bit lane-team Hadamard of the stage and item kernels (me.hip hadTeamPk / hadTeamCross, biased last stage) at the EXTREMES of its operand range: whole blocks of
    org = 2 * 2^bd - 2 / -(2^bd - 1) (the bi-prediction pattern's range) against cur = 0 / 2^bd - 1, constant, checkerboard, stripes and random-of-extremes, every tile kind of
    the reference's ladder (16x8, 8x16, 8x4, 4x8, 16x16_fast, 8x8, 4x4, 2x2), bit depths 10 and 8, against oracle.dist                       RdCost.cpp:1225-1322, 1324-1766, 1818-1938
  * the DMVR packed minimum (cost << 6 | order) against the reference's strict `<` scan: 25-way ties (flat planes) and pairwise ties (identical references: cost(d) == cost(-d)),
    all four sub-block sizes — the centre / the first position in raster order must win                                                        InterPrediction.cpp:1330-1366
  * a stage bundle whose first unit is SHORTER than a later one (the wave's LDS slice was sized from the leader only)                 me.hip vvhip_me_plan_create
Tolerance 0."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_gpu_me_shapes import _run_plan  # noqa: E402


@pytest.fixture(scope="module")
def env():
    from vvenc_amd.hotpath import HotPath
    from oracle.oracle import Oracle
    return HotPath(), Oracle()


def _pattern(kind, a, b, H, W, rng):
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == "const_a":
        v = np.full((H, W), a)
    elif kind == "const_b":
        v = np.full((H, W), b)
    elif kind == "checker":
        v = np.where((xx + yy) & 1, a, b)
    elif kind == "cols":
        v = np.where(xx & 1, a, b)
    elif kind == "rows":
        v = np.where(yy & 1, a, b)
    elif kind == "blocks4":                       # 4x4 blocks of alternating extremes: drives the middle Hadamard stages, not only the first / last
        v = np.where(((xx >> 2) + (yy >> 2)) & 1, a, b)
    else:
        v = np.where(rng.integers(0, 2, (H, W)) == 1, a, b)
    return np.ascontiguousarray(v.astype(np.int16))


# every tile kind of the ladder (RdCost.cpp:1818-1938): 16x8 tiles (w > h, h % 8 == 0, w % 16 == 0), 8x16, 8x4 (w > h, h % 4 == 0, w % 8 == 0), 4x8, 16x16_fast (HAD_fast, square, % 32),
# 8x8, 4x4, 2x2
ITEM_SHAPES = [(16, 8), (32, 16), (128, 64), (8, 16), (16, 32), (64, 128), (8, 4), (16, 4), (4, 8), (4, 16), (32, 32), (64, 64), (128, 128), (8, 8), (16, 16), (4, 4), (2, 2), (2, 8), (8, 2)]
STAGE_SHAPES = [(16, 8), (32, 16), (128, 64), (8, 16), (16, 32), (64, 128), (8, 4), (16, 4), (4, 8), (4, 16), (32, 32), (64, 64), (128, 128), (8, 8), (16, 16)]
ORG_KINDS = ("const_a", "const_b", "checker", "cols", "rows", "blocks4", "random")
CUR_KINDS = ("const_a", "const_b", "checker", "random")


@pytest.mark.parametrize("bd", [10, 8])
def test_hadamard_table_calls_on_saturated_operands(env, bd):
    """item kernels: HAD / HAD_fast / HAD_2SAD (and SAD / SSE next to them) on blocks whose every difference is at the end of the range"""
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    hp, orc = env
    rng = np.random.default_rng(500 + bd)
    top = 1 << bd
    hi, lo = 2 * top - 2, -(top - 1)
    H, W, M = 160, 192, 16
    compared = {}
    for ok in ORG_KINDS:
        for ck in CUR_KINDS:
            org_np = _pattern(ok, hi, lo, H, W, rng)
            cur_np = _pattern(ck, top - 1, 0, H, W, rng)
            org, cur = hp.plane(org_np, 8), hp.plane(cur_np, M)
            cur_pad = np.pad(cur_np, M, mode="edge")
            items, exp, tag = [], [], []
            for (w, h) in ITEM_SHAPES:
                for f in ("HAD", "HAD_fast", "HAD_2SAD", "SAD", "SSE"):
                    for (x, y, cx, cy) in ((0, 0, 0, 0), (2, 2, 5, 3)):          # aligned, and a shifted pair (odd displacement flips the parity of the second operand's pattern)
                        items.append((y * org.stride + x, cy * cur.stride + cx, 0, 1, DF[f], 0, w, h))
                        if f == "HAD_2SAD":          # (the reference's entry takes compact operands: RdCost.cpp:1778)
                            e = orc.dist(f, (np.ascontiguousarray(org_np[y:y + h, x:x + w]), 0, 0), (np.ascontiguousarray(cur_pad[M + cy:M + cy + h, M + cx:M + cx + w]), 0, 0), w, h, bd, 0)
                        else:
                            e = orc.dist(f, (org_np, y, x), (cur_pad, M + cy, M + cx), w, h, bd, 0)
                        exp.append(e)
                        tag.append((ok, ck, f, w, h, x, y))
            it = np.array(items, RP.ME_ITEM)
            planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (cur.storage.data_ptr() + 2 * cur.origin, cur.stride)]
            _, _, ic = _run_plan(hp, planes, np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), np.zeros(0, RP.ME_STAGE_JOB), it, None, bd)
            bad = np.nonzero(ic[:len(exp)] != np.array(exp, np.int64))[0]
            assert bad.size == 0, [(tag[i], int(ic[i]), exp[i]) for i in bad[:6]]
            for t in tag:
                compared[(t[3], t[4])] = compared.get((t[3], t[4]), 0) + 1
    assert all(compared.get(s, 0) >= 5 * 2 * len(ORG_KINDS) * len(CUR_KINDS) for s in ITEM_SHAPES), compared
    # the extreme of the extremes is really in the set: an 8x8 tile whose DC coefficient is 64 * (2^(bd+1) - 2) — beyond 16 bits
    assert orc.dist("HAD", (_pattern("const_a", hi, lo, 8, 8, rng), 0, 0), (np.zeros((8, 8), np.int16), 0, 0), 8, 8, bd, 0) > 0


@pytest.mark.parametrize("bd,filter_mode", [(10, 2), (10, 1), (10, 0), (8, 2)])
def test_hadamard_stages_on_saturated_operands(env, bd, filter_mode):
    """stage kernels: interpolation of reference planes made of the two extreme sample values, then HAD / HAD_fast / SAD against originals at the ends of the bi-prediction range"""
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    from vvenc_amd.recorded import REFINE_H, REFINE_Q
    hp, orc = env
    rng = np.random.default_rng(600 + bd + filter_mode)
    top = 1 << bd
    hi, lo = 2 * top - 2, -(top - 1)
    H, W, M = 192, 224, 24
    compared = {}
    for ok in ORG_KINDS:
        for ck in CUR_KINDS:
            org_np = _pattern(ok, hi, lo, H, W, rng)
            ref_np = _pattern(ck, top - 1, 0, H, W, rng)
            org, ref = hp.plane(org_np, 8), hp.plane(ref_np, M)
            ref_pad = np.pad(ref_np, M, mode="edge")
            stages, meta = [], []
            for (w, h) in STAGE_SHAPES:
                for f in ("HAD", "HAD_fast", "SAD"):
                    for i_frac in (2, 1):
                        x, y = int(rng.integers(0, (W - w) // 2 + 1)) * 2, int(rng.integers(0, (H - h) // 2 + 1)) * 2
                        bx, by = min(max(x + int(rng.integers(-2, 3)), 0), W - w), min(max(y + int(rng.integers(-2, 3)), 0), H - h)
                        alt = int(filter_mode != 2 and i_frac == 2 and rng.integers(0, 2))
                        stages.append((y * org.stride + x, by * ref.stride + bx, w, h, 0, 1, i_frac, filter_mode, alt, DF[f], 0, 0, 511, 0))
                        meta.append((x, y, bx, by, w, h, i_frac, alt, f))
            sj = np.array(stages, RP.ME_STAGE_JOB)
            planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (ref.storage.data_ptr() + 2 * ref.origin, ref.stride)]
            _, sc, _ = _run_plan(hp, planes, np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), sj, np.zeros(0, RP.ME_ITEM), None, bd)
            bad = []
            for s, (x, y, bx, by, w, h, i_frac, alt, f) in enumerate(meta):
                refine = REFINE_H if i_frac == 2 else REFINE_Q
                for k in range(9):
                    tx, ty = int(refine[k][0]) * i_frac * 4, int(refine[k][1]) * i_frac * 4
                    pred = orc.if_pred_luma_me((ref_pad, M + by + (ty >> 4), M + bx + (tx >> 4)), w, h, tx & 15, ty & 15, bd, bool(alt), filter_mode)
                    e = orc.dist(f, (org_np, y, x), (pred, 0, 0), w, h, bd, 0)
                    compared[(w, h)] = compared.get((w, h), 0) + 1
                    if int(sc[s, k]) != e:
                        bad.append(((ok, ck), (w, h), f, i_frac, k, int(sc[s, k]), e))
            assert not bad, bad[:6]
    assert all(compared.get(s, 0) >= 9 * 6 * len(ORG_KINDS) * len(CUR_KINDS) for s in STAGE_SHAPES), compared


def test_dmvr_ties_centre_and_first_in_raster_win(env):
    """25-way ties (flat planes, equal and different levels), pairwise ties (identical references with equal fractions: cost(d) == cost(-d) for all 12 pairs), column / row ties
    (planes that vary along one axis only): mvd and min_cost == the reference's strict-`<` scan (oracle.dmvr_refine), all four sub-block sizes, bit depths 10 and 8"""
    from vvenc_amd.hotpath import DMVR_ITEM_DTYPE, DMVR_RESULT_DTYPE
    hp, orc = env
    rng = np.random.default_rng(777)
    H, W = 96, 160
    yy, xx = np.mgrid[0:H, 0:W]
    n_checked = n_tied_nonzero = 0
    for bd in (10, 8):
        top = (1 << bd) - 1
        tex = np.clip((512 + 300 * np.sin(xx / 3.7) * np.cos(yy / 2.9) + rng.normal(0, 40, (H, W))) * (top + 1) / 1024.0, 0, top).astype(np.int16)
        xonly = np.clip((512 + 400 * np.sin(xx / 2.3) + 0 * yy) * (top + 1) / 1024.0, 0, top).astype(np.int16)
        yonly = np.clip((512 + 400 * np.sin(yy / 2.1) + 0 * xx) * (top + 1) / 1024.0, 0, top).astype(np.int16)
        flat = lambda v: np.full((H, W), v, np.int16)
        cases = [("flat equal", flat(top // 2), flat(top // 2), True), ("flat zero", flat(0), flat(0), True), ("flat different", flat(top), flat(0), True), ("flat off by one", flat(300 * (top + 1) // 1024), flat(300 * (top + 1) // 1024 + 1), True),
                 ("identical texture", tex, tex, True), ("identical texture, free fractions", tex, tex, False), ("x only", xonly, xonly, False), ("y only", yonly, yonly, False),
                 ("x only vs texture", xonly, tex, False), ("checkerboard", np.where((xx + yy) & 1, top, 0).astype(np.int16), np.where((xx + yy) & 1, top, 0).astype(np.int16), True)]
        for name, r0, r1, same_frac in cases:
            p0, p1 = hp.plane(r0, 0), hp.plane(r1, 0)
            for (dx, dy) in ((16, 16), (8, 8), (16, 8), (8, 16)):
                n = 24
                it = np.zeros(n, DMVR_ITEM_DTYPE)
                pos = [(int(rng.integers(8, W - dx - 8)), int(rng.integers(8, H - dy - 8))) for _ in range(n)]
                it["ref0_off"] = [y * p0.stride + x for (x, y) in pos]
                it["ref1_off"] = [y * p1.stride + x for (x, y) in pos]
                for f in ("frac0_x", "frac0_y"):
                    it[f] = rng.integers(0, 16, n)
                it["frac1_x"], it["frac1_y"] = (it["frac0_x"], it["frac0_y"]) if same_frac else (rng.integers(0, 16, n), rng.integers(0, 16, n))
                it[:4]["frac0_x"] = 0; it[:4]["frac1_x"] = 0; it[2:6]["frac0_y"] = 0; it[2:6]["frac1_y"] = 0
                res = hp.dmvr_refine_batch(p0, p1, hp.to_device(it), n, dx, dy, bd).cpu().numpy().reshape(-1).view(DMVR_RESULT_DTYPE)
                for k, (x, y) in enumerate(pos):
                    exp = orc.dmvr_refine((r0, y, x), (r1, y, x), (int(it[k]["frac0_x"]), int(it[k]["frac0_y"])), (int(it[k]["frac1_x"]), int(it[k]["frac1_y"])), dx, dy, bd)
                    got = (int(res[k]["mvd_x"]), int(res[k]["mvd_y"]), int(res[k]["min_cost"]))
                    assert got == exp, (bd, name, dx, dy, k, got, exp)
                    n_checked += 1
                    n_tied_nonzero += int(exp[2] > 0)
                if name.startswith("flat"):
                    assert all(int(r["mvd_x"]) == 0 and int(r["mvd_y"]) == 0 for r in res), (bd, name, dx, dy)          # every cost equal: the centre wins
    assert n_checked == 2 * 10 * 4 * 24 and n_tied_nonzero > 200


def test_stage_bundle_with_a_taller_unit_behind_a_short_leader(env):
    """bundles group single-unit stages by unit width, launch class and deal — not by height.  [8x4 with nine positions, 8x32 with one position] fits one work budget;
    the wave's LDS slice has to hold the 8x32 unit's rows although the 8x4 leads.  Same for 16-wide and 4-wide units and for both sub-pel tap sets"""
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    from vvenc_amd.recorded import REFINE_H
    hp, orc = env
    rng = np.random.default_rng(99)
    bd = 10
    H, W, M = 256, 256, 24
    org_np = rng.integers(0, 1024, (H, W), dtype=np.int16)
    ref_np = rng.integers(0, 1024, (H, W), dtype=np.int16)
    org, ref = hp.plane(org_np, 8), hp.plane(ref_np, M)
    ref_pad = np.pad(ref_np, M, mode="edge")
    for filter_mode in (2, 0):
        stages, meta = [], []
        for rep in range(6):                    # (picture order = ref_off order inside a sub-class: short leader first, then the tall unit)
            for (w, hs, ht) in ((8, 4, 32), (16, 4, 32), (4, 8, 32), (8, 8, 32)):
                y0 = 8 + rep * 40
                for (h, mask, x) in ((hs, 511, 8 + 24 * (w // 4)), (ht, 1 << int(rng.integers(0, 9)), 8 + 24 * (w // 4) + 1)):
                    stages.append((y0 * org.stride + x, y0 * ref.stride + x, w, h, 0, 1, 2, filter_mode, 0, DF["HAD"], 0, 0, mask, 0))
                    meta.append((x, y0, w, h, mask))
        sj = np.array(stages, RP.ME_STAGE_JOB)
        planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (ref.storage.data_ptr() + 2 * ref.origin, ref.stride)]
        _, sc, _ = _run_plan(hp, planes, np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), sj, np.zeros(0, RP.ME_ITEM), None, bd)
        n = 0
        for s, (x, y, w, h, mask) in enumerate(meta):
            for k in range(9):
                if not (mask >> k) & 1:
                    assert sc[s, k] == 0
                    continue
                tx, ty = int(REFINE_H[k][0]) * 8, int(REFINE_H[k][1]) * 8
                pred = orc.if_pred_luma_me((ref_pad, M + y + (ty >> 4), M + x + (tx >> 4)), w, h, tx & 15, ty & 15, bd, False, filter_mode)
                assert int(sc[s, k]) == orc.dist("HAD", (org_np, y, x), (pred, 0, 0), w, h, bd, 0), (filter_mode, s, k, (w, h))
                n += 1
        assert n == 6 * 4 * (9 + 1)


def test_fused_tu_all_zero_level_shortcut_vs_oracle(env):
    """round 5: tiles (waves) whose every TU quantises to all-zero levels skip the quantiser loop, the level staging and both inverse passes (tuMxBody / tuMx64Body): what they
    write must be what the long way writes — levels 0, reconstruction 0, abs sum 0, the oracle's last position / need-RDOQ flag / SSE — into output buffers pre-filled with
    garbage.  Lists: all TUs tiny; whole tiles tiny next to tiles with levels; ONE TU with levels inside an otherwise all-zero tile; ragged last tiles; per-TU QPs; all sizes of the
    matrix-core launch (4 .. 64), DCT-2 and DST-7 / DCT-8 pairs, bit depths 10 and 8"""
    import torch
    from vvenc_amd.hotpath import STATS_DTYPE, HotPath
    hp, orc = env
    rng = np.random.default_rng(4242)
    n_zero_tus = n_level_tus = 0
    for bd in (10, 8):
        for S in (4, 8, 16, 32, 64):
            tpt = 1 if S == 64 else (32 // S) ** 2                                   # TUs per 32x32 tile = per wave
            for (th, tv) in ((0, 0),) + (((2, 2), (1, 2)) if S <= 32 else ()):
                for pattern in ("all tiny", "tiles alternate", "one TU with levels per tile"):
                    n = 3 * tpt + max(1, tpt // 3) if S < 64 else 7                  # whole tiles + a ragged one
                    big = np.zeros(n, bool)
                    if pattern == "tiles alternate":
                        big = (np.arange(n) // tpt) % 2 == 1
                    elif pattern == "one TU with levels per tile":
                        starts = np.arange(0, n, tpt)
                        big[np.minimum(starts + rng.integers(0, tpt, starts.size), n - 1)] = True
                    amp = min(300, (1 << bd) - 1)                                            # (residuals inside the bit depth's range: the interface's contract)
                    resi = np.where(big[:, None, None], rng.integers(-amp, amp + 1, (n, S, S)), rng.integers(-1, 2, (n, S, S))).astype(np.int16)
                    resi[~big] *= (rng.integers(0, 3, (n, 1, 1))[~big] > 0)           # some exactly-zero residuals too
                    qps = np.where(big, rng.integers(22, 40, n), rng.integers(30, 52, n))
                    irap = rng.integers(0, 2, n)
                    pool = hp.to_device(resi.reshape(-1))
                    off = hp.to_device((np.arange(n, dtype=np.int32) * S * S).astype(np.int32))
                    qp = hp.to_device(HotPath.tu_qp(qps, irap, 1))
                    lev = torch.full((n * S * S,), 0x7777, dtype=torch.int16, device=hp.device)
                    rec = torch.full((n * S * S,), 0x5555, dtype=torch.int16, device=hp.device)
                    st = torch.full((n, STATS_DTYPE.itemsize), 0xEE, dtype=torch.uint8, device=hp.device)
                    hp.tu_rdo_multi_strided(pool, [S], [(S, S, th, tv, n, 8, off, qp, lev, rec, st)], bd)
                    torch.cuda.synchronize()
                    lv, rc = lev.cpu().numpy().reshape(n, S, S), rec.cpu().numpy().reshape(n, S, S)
                    sv = st.cpu().numpy().view(STATS_DTYPE).reshape(n)
                    for i in range(n):
                        el, er, es = orc.tu_rdo(resi[i], int(qps[i]), int(irap[i]), th, tv, bd, 8, 1)
                        assert np.array_equal(lv[i], el), ("lev", bd, S, th, tv, pattern, i)
                        assert np.array_equal(rc[i], er), ("rec", bd, S, th, tv, pattern, i)
                        got = (int(sv["abs_sum"][i]), int(sv["last_scan_pos"][i]), int(sv["need_rdoq"][i]), int(sv["sse"][i]))
                        assert got == (es["abs_sum"], es["last_scan_pos"], es["need_rdoq"], es["sse"]), (bd, S, th, tv, pattern, i, got, es)
                        n_zero_tus += es["abs_sum"] == 0
                        n_level_tus += es["abs_sum"] != 0
    assert n_zero_tus > 1000 and n_level_tus > 300, (n_zero_tus, n_level_tus)


def test_fused_tu_sparse_outputs_leave_all_zero_tus_untouched(env):
    """round 6 (vvhip_tu_set_sparse_outputs): with the sparse contract a TU whose levels are all zero gets its statistics ONLY.  Garbage-prefilled level / reconstruction buffers:
    a TU with levels must hold the oracle's levels and reconstruction; a TU without must be either untouched (0x7777 / 0x5555 everywhere: the whole tile took the shortcut) or
    all zero (it shared its tile with a TU that has levels) — never anything else; statistics always the oracle's; whole all-zero tiles MUST be untouched (that is the point);
    and switching the contract back restores the dense behaviour on the same context"""
    import torch
    from vvenc_amd.hotpath import STATS_DTYPE, HotPath
    hp, orc = env
    rng = np.random.default_rng(777)
    untouched = zeroed = with_levels = 0
    try:
        hp.tu_set_sparse_outputs(True)
        for bd in (10, 8):
            for S in (8, 16, 32, 64):
                tpt = 1 if S == 64 else (32 // S) ** 2
                for (th, tv) in ((0, 0),) + (((2, 2),) if S <= 32 else ()):
                    for pattern in ("all tiny", "tiles alternate", "one TU with levels per tile"):
                        n = 3 * tpt + max(1, tpt // 3) if S < 64 else 7
                        big = np.zeros(n, bool)
                        if pattern == "tiles alternate":
                            big = (np.arange(n) // tpt) % 2 == 1
                        elif pattern == "one TU with levels per tile":
                            starts = np.arange(0, n, tpt)
                            big[np.minimum(starts + rng.integers(0, tpt, starts.size), n - 1)] = True
                        amp = min(300, (1 << bd) - 1)
                        resi = np.where(big[:, None, None], rng.integers(-amp, amp + 1, (n, S, S)), rng.integers(-1, 2, (n, S, S))).astype(np.int16)
                        qps = np.where(big, rng.integers(22, 40, n), rng.integers(30, 52, n))
                        irap = rng.integers(0, 2, n)
                        pool = hp.to_device(resi.reshape(-1))
                        off = hp.to_device((np.arange(n, dtype=np.int32) * S * S).astype(np.int32))
                        qp = hp.to_device(HotPath.tu_qp(qps, irap, 1))
                        lev = torch.full((n * S * S,), 0x7777, dtype=torch.int16, device=hp.device)
                        rec = torch.full((n * S * S,), 0x5555, dtype=torch.int16, device=hp.device)
                        st = torch.full((n, STATS_DTYPE.itemsize), 0xEE, dtype=torch.uint8, device=hp.device)
                        hp.tu_rdo_multi_strided(pool, [S], [(S, S, th, tv, n, 8, off, qp, lev, rec, st)], bd)
                        torch.cuda.synchronize()
                        lv, rc = lev.cpu().numpy().reshape(n, S, S), rec.cpu().numpy().reshape(n, S, S)
                        sv = st.cpu().numpy().view(STATS_DTYPE).reshape(n)
                        exp = [orc.tu_rdo(resi[i], int(qps[i]), int(irap[i]), th, tv, bd, 8, 1) for i in range(n)]
                        for i, (el, er, es) in enumerate(exp):
                            got = (int(sv["abs_sum"][i]), int(sv["last_scan_pos"][i]), int(sv["need_rdoq"][i]), int(sv["sse"][i]))
                            assert got == (es["abs_sum"], es["last_scan_pos"], es["need_rdoq"], es["sse"]), (bd, S, th, tv, pattern, i, got, es)
                            if es["abs_sum"] != 0:
                                assert np.array_equal(lv[i], el) and np.array_equal(rc[i], er), (bd, S, pattern, i)
                                with_levels += 1
                            else:
                                kept = bool((lv[i] == 0x7777).all() and (rc[i] == 0x5555).all())
                                zero = bool((lv[i] == 0).all() and (rc[i] == 0).all())
                                assert kept or zero, ("neither untouched nor zero", bd, S, th, tv, pattern, i)
                                tile_all_zero = all(e[2]["abs_sum"] == 0 for e in exp[(i // tpt) * tpt:(i // tpt + 1) * tpt])
                                if tile_all_zero:
                                    assert kept, ("an all-zero tile wrote its outputs", bd, S, th, tv, pattern, i)
                                untouched += kept
                                zeroed += zero
        assert untouched > 300 and with_levels > 150, (untouched, zeroed, with_levels)
    finally:
        hp.tu_set_sparse_outputs(False)
    # dense again: the same all-tiny list writes zeros everywhere
    S, n = 16, 8
    resi = rng.integers(-1, 2, (n, S, S)).astype(np.int16)
    pool = hp.to_device(resi.reshape(-1))
    off = hp.to_device((np.arange(n, dtype=np.int32) * S * S).astype(np.int32))
    qp = hp.to_device(HotPath.tu_qp(np.full(n, 45), np.zeros(n, int), 1))
    lev = torch.full((n * S * S,), 0x7777, dtype=torch.int16, device=hp.device)
    rec = torch.full((n * S * S,), 0x5555, dtype=torch.int16, device=hp.device)
    st = torch.zeros((n, STATS_DTYPE.itemsize), dtype=torch.uint8, device=hp.device)
    hp.tu_rdo_multi_strided(pool, [S], [(S, S, 0, 0, n, 8, off, qp, lev, rec, st)], 10)
    torch.cuda.synchronize()
    assert int(lev.abs().sum()) == 0 and int(rec.abs().sum()) == 0


def test_quant_core_lfnst_rule_vs_oracle(env):
    """round 6: QuantCore's first-coefficient-group rule for LFNST TUs on the device (vvhip_quant_core_lfnst; Quant.cpp:149-159) — the binding no longer hands those TUs to the
    CPU entry.  Against the oracle (pinned to the reference's own entry with CodingUnit::lfnstIdx set, tests/test_oracle_vs_reference.py): levels of the WHOLE block (zeros
    behind the first group), abs sum, last position, deltaU inside the quantised range; lfnst_idx 0 = the plain core"""
    hp, orc = env
    rng = np.random.default_rng(2024)
    n = 0
    for (w, h) in ((4, 4), (8, 8), (4, 8), (8, 4), (16, 16), (4, 16), (16, 8), (32, 32), (8, 32), (64, 64)):
        for qp in (12 + 22, 12 + 37):
            qc, qbits, add = orc.quant_params(w, h, 10, qp, 0)
            for lfnst in (0, 1, 2):
                coef = rng.integers(-(1 << 13), 1 << 13, size=(h, w)).astype(np.int32)
                coef[rng.random((h, w)) < 0.3] = 0
                if w > 32:
                    coef[:, 32:] = 0
                if h > 32:
                    coef[32:, :] = 0
                e = orc.quant_core(coef, qc, qbits, add, 8, lfnst_idx=lfnst)
                g = hp.quant_core(coef, qc, qbits, add, 8, lfnst_idx=lfnst)
                assert np.array_equal(g[0], e[0]) and g[2] == e[2] and g[3] == e[3], (w, h, qp, lfnst, g[2:], e[2:])
                nz = np.zeros(h * w, bool)
                nz[orc.scan_order(w.bit_length() - 1, h.bit_length() - 1)[: e[3] + 1]] = True
                assert np.array_equal(g[1][nz], e[1][nz]), ("deltaU", w, h, qp, lfnst)
                n += 1
    assert n == 60


def test_masked_sad_large_operands_sum_in_64_bits(env):
    """masked SADs take any int16 operands (include/vvenc_hip.h) and the reference sums in a 64-bit Distortion (RdCost.cpp:2062-2093): a 128x128 block of large
    differences and weights passes 2^32 — the wave takes its 64-bit form; GEO-range operands next to it keep the 32-bit form.  Against the numpy restatement of the loop"""
    import torch
    from vvenc_amd import replay as RP
    hp, _ = env
    rng = np.random.default_rng(31)
    H, W, M = 160, 192, 16
    cases = [("extremes", 32767, -32768, 32767), ("large", 30000, -30000, 1000), ("geo", 1023, 0, 8)]          # (weights >= 0: what a mask is)
    for name, hi, lo, wmax in cases:
        org_np = rng.integers(lo, hi + 1, (H, W)).astype(np.int16)
        cur_np = rng.integers(lo, hi + 1, (H, W)).astype(np.int16)
        if name == "extremes":
            org_np[:], cur_np[:] = hi, lo                                                  # every difference 65535
        org, cur = hp.plane(org_np, 8), hp.plane(cur_np, M)
        items, exp, at = [], [], 0
        pool_np = np.zeros(1 << 17, np.int16)
        for (w, h, ss) in ((128, 128, 0), (128, 128, 1), (64, 64, 0), (32, 16, 0), (8, 8, 0), (16, 4, 0), (4, 8, 0)):
            rows = h >> ss
            wg = (np.full((rows, w), wmax) if name == "extremes" else rng.integers(min(0, wmax), max(0, wmax) + 1, (rows, w))).astype(np.int16)
            pool_np[at:at + rows * w] = wg.reshape(-1)
            x, y, cx, cy = 8, 6, 11, 9
            items.append((y * org.stride + x, cy * cur.stride + cx, at, 0, 1, 2, ss, w, h, 0))
            a = org_np[y:y + h:1 << ss, x:x + w].astype(np.int64)
            b = cur_np[cy:cy + h:1 << ss, cx:cx + w].astype(np.int64)
            exp.append((int((np.abs(a - b) * wg.astype(np.int64)).sum()) << ss) & 0xFFFFFFFFFFFFFFFF)          # (Distortion is unsigned 64-bit)
            at += (rows * w + 7) & ~7
        pool = torch.from_numpy(pool_np).to(hp.device)
        mi = np.array(items, RP.ME_MASK_ITEM)
        planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (cur.storage.data_ptr() + 2 * cur.origin, cur.stride), (pool.data_ptr(), 0)]
        _, _, ic = _run_plan(hp, planes, np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), np.zeros(0, RP.ME_STAGE_JOB), np.zeros(0, RP.ME_ITEM), mi, 10)
        got = ic[:len(exp)].view(np.uint64)
        assert [int(g) for g in got] == exp, (name, [int(g) for g in got], exp)
    assert exp is not None
