"""-m gpu: motion-search plans (vvenc_amd/csrc/me.hip) against the ORACLE directly, over every block shape preset medium's CTU 128 + multi-type tree produces.

What the plan returns is compared with oracle/ (the CPU restatement pinned to the compiled reference), not with another HIP kernel:
  * integer windows      xGetSAD* incl. the subShift rule                         RdCost.cpp:301-644       rectangular 8..128 x 4..128
  * refinement stages    the search's interpolation (4-tap / 6-tap + alternative half-sample / 8-tap) + SAD / HAD / HAD_fast   InterSearch.cpp:760-880,
                         with the reference's tile ladder (16x8, 8x16, 8x4, 4x8 in double precision, 16x16_fast, 8x8)           RdCost.cpp:1818-1938
  * table calls          SSE / SAD / HAD / HAD_fast / HAD_2SAD on 2..128 x 2..128 (4x4 and 2x2 tiles included), pool (compact) operands
  * masked SADs          xGetSADwMask                                             RdCost.cpp:2062-2093
at bit depths 10 and 8, unsigned and signed (bi-prediction pattern) originals.  Tolerance 0.  Negative cases: what plan creation must reject."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES_ME = [(w, h) for w in (8, 16, 32, 64, 128) for h in (4, 8, 16, 32, 64, 128)]
SHAPES_STAGE = SHAPES_ME + [(4, 8), (4, 16), (4, 32)]
SHAPES_ITEM = [(w, h) for w in (2, 4, 8, 16, 32, 64, 128) for h in (2, 4, 8, 16, 32, 64, 128)]


@pytest.fixture(scope="module")
def env():
    from vvenc_amd.hotpath import HotPath
    from oracle.oracle import Oracle
    return HotPath(), Oracle()


def _run_plan(hp, planes, ij, cands, sj, items, mask_items, bd, max_window=16):
    import torch
    from vvenc_amd import replay as RP
    plan = hp.me_plan_create(ij, cands, sj, items, bd, max_window, mask_items=mask_items)
    tab = (RP.MePlane * 16)()
    for k, (ptr, stride) in enumerate(planes):
        tab[k] = RP.MePlane(ptr, stride, 0)
    cc = torch.full((max(1, cands.size),), -1, dtype=torch.int64, device=hp.device)
    sc = torch.full((max(1, 9 * sj.size),), -1, dtype=torch.int64, device=hp.device)            # (not zeros: the kernel must write every stage's nine costs itself)
    ic = torch.full((max(1, items.size + (0 if mask_items is None else mask_items.size)),), -1, dtype=torch.int64, device=hp.device)
    hp.me_plan_run(plan, tab, len(planes), cc, sc, ic)
    torch.cuda.synchronize()
    hp.me_plan_destroy(plan)
    return cc.cpu().numpy(), sc.cpu().numpy()[:9 * sj.size].reshape(-1, 9), ic.cpu().numpy()


def _setup(hp, rng, bd, signed_org, H=320, W=384, M=80):
    top = 1 << bd
    if signed_org:                               # bi-prediction pattern 2 * org - pred: |value| < 2^(bd + 1)
        org_np = rng.integers(-(top - 1), 2 * top - 1, size=(H, W), dtype=np.int16)
    else:
        org_np = rng.integers(0, top, size=(H, W), dtype=np.int16)
    ref_np = rng.integers(0, top, size=(H, W), dtype=np.int16)
    org, ref = hp.plane(org_np, 8), hp.plane(ref_np, M)
    ref_pad = np.pad(ref_np, M, mode="edge")
    return org_np, ref_np, ref_pad, org, ref, M


@pytest.mark.parametrize("bd,signed_org,seed", [(10, False, 1), (10, True, 2), (8, False, 3)])
def test_integer_windows_all_shapes_vs_oracle(env, bd, signed_org, seed):
    from vvenc_amd import replay as RP
    hp, orc = env
    rng = np.random.default_rng(seed)
    org_np, ref_np, ref_pad, org, ref, M = _setup(hp, rng, bd, signed_org)
    H, W = org_np.shape
    jobs, cands, exp = [], [], []
    for (w, h) in SHAPES_ME:
        for ss in ((0, 1) if h > 8 else (0,)):
            x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
            n = int(rng.integers(2, 24))
            spread = int(rng.choice([1, 3, 9, 30]))
            dx = np.clip(rng.integers(-spread, spread + 1, n), -M + 8 - x, None)
            dy = np.clip(rng.integers(-spread, spread + 1, n), -M + 8 - y, None)
            jobs.append((y * org.stride + x, y * ref.stride + x, w, h, 0, 1, ss, 0, 0, 0, 0, 0, len(cands), n))
            for a, b in zip(dx, dy):
                cands.append((int(a), int(b)))
                exp.append(orc.dist("SAD", (org_np, y, x), (ref_pad, M + y + int(b), M + x + int(a)), w, h, bd, ss))
    ij = np.array(jobs, RP.ME_INT_JOB)
    pc = np.array(cands, RP.ME_CAND)
    planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (ref.storage.data_ptr() + 2 * ref.origin, ref.stride)]
    cc, _, _ = _run_plan(hp, planes, ij, pc, np.zeros(0, RP.ME_STAGE_JOB), np.zeros(0, RP.ME_ITEM), None, bd)
    bad = np.nonzero(cc[:len(exp)] != np.array(exp, np.int64))[0]
    assert bad.size == 0, (bad[:8], [(cands[i], int(cc[i]), exp[i]) for i in bad[:4]])


def test_integer_windows_repeated_positions(env):
    """a search call lists its start point again and again (half of a recorded picture's integer candidates repeat a position of their call): the plan scores a position once and
    stores the cost to every entry that listed it — also when a position is listed more often than its team has lanes (8x4: four), and when the repeats are far apart in the list"""
    from vvenc_amd import replay as RP
    hp, orc = env
    bd = 10
    rng = np.random.default_rng(77)
    org_np, ref_np, ref_pad, org, ref, M = _setup(hp, rng, bd, False)
    H, W = org_np.shape
    jobs, cands, exp = [], [], []
    for (w, h, ss) in [(8, 4, 0), (8, 8, 0), (16, 16, 1), (64, 64, 1), (32, 8, 0), (128, 64, 1)]:
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        start = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
        lst = [start] * 11 + [(start[0] + 1, start[1]), start, (start[0], start[1] - 1)] * 3 + [(int(a), int(b)) for a, b in rng.integers(-20, 21, (9, 2))] + [start] * 70
        jobs.append((y * org.stride + x, y * ref.stride + x, w, h, 0, 1, ss, 0, 0, 0, 0, 0, len(cands), len(lst)))
        memo = {}
        for a, b in lst:
            cands.append((a, b))
            if (a, b) not in memo:
                memo[(a, b)] = orc.dist("SAD", (org_np, y, x), (ref_pad, M + y + b, M + x + a), w, h, bd, ss)
            exp.append(memo[(a, b)])
    ij = np.array(jobs, RP.ME_INT_JOB)
    pc = np.array(cands, RP.ME_CAND)
    planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (ref.storage.data_ptr() + 2 * ref.origin, ref.stride)]
    cc, _, _ = _run_plan(hp, planes, ij, pc, np.zeros(0, RP.ME_STAGE_JOB), np.zeros(0, RP.ME_ITEM), None, bd)
    bad = np.nonzero(cc[:len(exp)] != np.array(exp, np.int64))[0]
    assert bad.size == 0, (bad[:8], [(cands[i], int(cc[i]), exp[i]) for i in bad[:4]])


@pytest.mark.parametrize("filter_mode,func,bd,signed_org,seed", [(2, "HAD", 10, False, 11), (2, "HAD_fast", 10, True, 12), (2, "SAD", 10, False, 13), (1, "HAD", 10, False, 14),
                                                                 (1, "SAD", 10, True, 15), (0, "HAD", 10, False, 16), (0, "HAD_fast", 8, False, 17), (2, "HAD", 8, False, 18)])
def test_refinement_stages_all_shapes_vs_oracle(env, filter_mode, func, bd, signed_org, seed):
    """every block shape through the stage kernel of one tap set: half- and quarter-sample stages, random position masks, the alternative half-sample filter"""
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    from vvenc_amd.recorded import REFINE_H, REFINE_Q
    hp, orc = env
    rng = np.random.default_rng(seed)
    org_np, ref_np, ref_pad, org, ref, M = _setup(hp, rng, bd, signed_org)
    H, W = org_np.shape
    stages, meta = [], []
    for (w, h) in SHAPES_STAGE:
        for rep in range(2):
            x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
            i_frac = int(rng.choice([1, 2]))
            bq = (0, 0) if i_frac == 2 else (int(rng.integers(-1, 2)) * 2, int(rng.integers(-1, 2)) * 2)          # quarter stage around a half-sample winner
            if i_frac == 1 and rng.integers(0, 3) == 0:
                bq = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
            mask = int(rng.integers(1, 512))
            mask = sum(1 << k for k in range(9) if (mask >> k) & 1 and _stage_in_reach(i_frac, bq, 1 << k))          # (plan creation rejects positions outside the staged rows)
            if not mask:
                bq, mask = (0, 0), 1
            alt = int(filter_mode != 2 and rng.integers(0, 2)) if filter_mode != 2 else int(rng.integers(0, 4) == 0)
            bx, by = x + int(rng.integers(-4, 5)), y + int(rng.integers(-4, 5))
            stages.append((y * org.stride + x, by * ref.stride + bx, w, h, 0, 1, i_frac, filter_mode, alt, DF[func], bq[0], bq[1], mask, 0))
            meta.append((x, y, bx, by, w, h, i_frac, bq, mask, alt))
    sj = np.array(stages, RP.ME_STAGE_JOB)
    planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (ref.storage.data_ptr() + 2 * ref.origin, ref.stride)]
    _, sc, _ = _run_plan(hp, planes, np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), sj, np.zeros(0, RP.ME_ITEM), None, bd)
    bad, compared = [], {}
    for s, (x, y, bx, by, w, h, i_frac, bq, mask, alt) in enumerate(meta):
        refine = REFINE_H if i_frac == 2 else REFINE_Q
        for k in range(9):
            if not (mask >> k) & 1:
                if sc[s, k] != 0:
                    bad.append((s, k, "unmasked position is not 0"))
                continue
            tx, ty = (int(refine[k][0]) + bq[0]) * i_frac * 4, (int(refine[k][1]) + bq[1]) * i_frac * 4
            if not (-16 <= tx <= 16 and -16 <= ty <= 16):
                continue
            pred = orc.if_pred_luma_me((ref_pad, M + by + (ty >> 4), M + bx + (tx >> 4)), w, h, tx & 15, ty & 15, bd, bool(alt), filter_mode)
            e = orc.dist(func, (org_np, y, x), (pred, 0, 0), w, h, bd, 0)
            compared[(w, h)] = compared.get((w, h), 0) + 1
            if int(sc[s, k]) != e:
                bad.append((s, k, (w, h), i_frac, bq, alt, int(sc[s, k]), e))
    assert not bad, bad[:8]
    # (positions outside the staged rows are skipped above — plan creation rejects them, the masks are built inside the reach: every shape must still have been compared)
    assert all(compared.get(sh, 0) > 0 for sh in SHAPES_STAGE), {sh: compared.get(sh, 0) for sh in SHAPES_STAGE}


def _stage_in_reach(i_frac, bq, mask):
    from vvenc_amd.recorded import REFINE_H, REFINE_Q
    refine = REFINE_H if i_frac == 2 else REFINE_Q
    return all(-16 <= (int(refine[k][0]) + bq[0]) * i_frac * 4 <= 16 and -16 <= (int(refine[k][1]) + bq[1]) * i_frac * 4 <= 16 for k in range(9) if (mask >> k) & 1)


@pytest.mark.parametrize("bd,signed_org,seed", [(10, False, 21), (10, True, 22), (8, False, 23)])
def test_table_calls_all_shapes_vs_oracle(env, bd, signed_org, seed):
    """every function on every shape 2..128 x 2..128, operands in planes and in a compact pool (stride 0 entry of the plane table); masked SADs on compact weight blocks"""
    import torch
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    hp, orc = env
    rng = np.random.default_rng(seed)
    org_np, ref_np, ref_pad, org, ref, M = _setup(hp, rng, bd, signed_org)
    H, W = org_np.shape
    top = 1 << bd
    pool_np = rng.integers(0, top, size=1 << 18, dtype=np.int16)
    pool = torch.from_numpy(pool_np).to(hp.device)
    items, exp, masks, mexp, mnext = [], [], [], [], 0
    for (w, h) in SHAPES_ITEM:
        for f in ("SSE", "SAD", "HAD", "HAD_fast", "HAD_2SAD"):
            for rep in range(2):
                ss = int(f == "SAD" and h > 8 and rng.integers(0, 2))
                x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
                cx, cy = int(rng.integers(-16, W - w + 16)), int(rng.integers(-16, H - h + 16))
                # (the oracle's HAD_2SAD entry takes compact operands like the reference's: RdCost.cpp:1778)
                oblk = (np.ascontiguousarray(org_np[y:y + h, x:x + w]), 0, 0) if f == "HAD_2SAD" else (org_np, y, x)
                if rep == 0:
                    items.append((y * org.stride + x, cy * ref.stride + cx, 0, 1, DF[f], ss, w, h))
                    cblk = (np.ascontiguousarray(ref_pad[M + cy:M + cy + h, M + cx:M + cx + w]), 0, 0) if f == "HAD_2SAD" else (ref_pad, M + cy, M + cx)
                    exp.append(orc.dist(f, oblk, cblk, w, h, bd, ss))
                else:
                    off = int(rng.integers(0, pool_np.size // 2 - w * h - 64)) & ~1          # compact pool block as the second operand (lower half of the pool; the weights sit in the upper)
                    items.append((y * org.stride + x, off, 0, 2, DF[f], ss, w, h))
                    blk = pool_np[off:off + w * h].reshape(h, w).copy()
                    exp.append(orc.dist(f, oblk, (blk, 0, 0), w, h, bd, ss))
        ss = int(h > 8 and rng.integers(0, 2))
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        cx, cy = int(rng.integers(-16, W - w + 16)), int(rng.integers(-16, H - h + 16))
        moff = pool_np.size // 2 + mnext
        mnext += (w * h + 7) & ~7
        rows = h >> ss
        pool_np[moff:moff + w * rows] = rng.integers(0, 9, w * rows)                         # GEO weights 0..8
        masks.append((y * org.stride + x, cy * ref.stride + cx, moff, 0, 1, 2, ss, w, h, 0))
        mblk = pool_np[moff:moff + w * rows].reshape(rows, w).copy()
        # compact weights, one row of w per evaluated row; without row sub-sampling that is (stepX 1, maskStride w, maskStride2 -w) of the oracle's entry
        mexp.append(orc.sad_mask((org_np, y, x), (ref_pad, M + cy, M + cx), (mblk, 0, 0), 1, -w, w, h, 0) if ss == 0 else None)
    pool = torch.from_numpy(pool_np).to(hp.device)
    it = np.array(items, RP.ME_ITEM)
    mi = np.array(masks, RP.ME_MASK_ITEM)
    planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (ref.storage.data_ptr() + 2 * ref.origin, ref.stride), (pool.data_ptr(), 0)]
    _, _, ic = _run_plan(hp, planes, np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), np.zeros(0, RP.ME_STAGE_JOB), it, mi, bd)
    bad = np.nonzero(ic[:len(exp)] != np.array(exp, np.int64))[0]
    assert bad.size == 0, [(items[i][4:], int(ic[i]), exp[i]) for i in bad[:8]]
    # masked SADs: numpy restatement of xGetSADwMask (RdCost.cpp:2062-2093) on the compact weights (the oracle's entry is pinned to the reference in test_oracle_vs_reference)
    for k, m in enumerate(masks):
        oo, co, mo, _, _, _, ss, w, h, _ = m
        y, x = divmod(oo, org.stride)
        cy, cx = divmod(co + M * ref.stride + M, ref.stride)
        rows = h >> ss
        a = org_np[y:y + h:1 << ss, x:x + w].astype(np.int64)
        b = ref_pad[cy:cy + h:1 << ss, cx:cx + w].astype(np.int64)
        wgt = pool_np[mo:mo + w * rows].reshape(rows, w).astype(np.int64)
        e = int((np.abs(a - b) * wgt).sum()) << ss
        assert int(ic[len(exp) + k]) == e, ("masked SAD", (w, h, ss), int(ic[len(exp) + k]), e)
        assert mexp[k] is None or mexp[k] == e, ("masked SAD: oracle vs restatement", (w, h))


def test_plan_creation_rejects_what_the_kernels_do_not_cover(env):
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    from vvenc_amd.lib import VVHipError
    hp, _ = env
    e_int, e_c, e_st, e_it = np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), np.zeros(0, RP.ME_STAGE_JOB), np.zeros(0, RP.ME_ITEM)

    def stage(w, h, i_frac, bq, mask, func="HAD"):
        sj = np.zeros(1, RP.ME_STAGE_JOB)
        sj[0] = (0, 0, w, h, 0, 1, i_frac, 2, 0, DF[func], bq[0], bq[1], mask, 0)
        return sj
    hp.me_plan_destroy(hp.me_plan_create(e_int, e_c, stage(16, 16, 2, (0, 0), 511), e_it, 10, 0))
    hp.me_plan_destroy(hp.me_plan_create(e_int, e_c, stage(16, 16, 1, (2, -2), 511), e_it, 10, 0))
    # a half-sample stage around a base two half samples away needs rows the kernel does not stage (vertical displacement +-24 / +-32 sixteenths)
    for bad in (stage(16, 16, 2, (0, 2), 511), stage(16, 16, 2, (0, -2), 0b10), stage(16, 16, 2, (3, 0), 1 << 4), stage(16, 16, 2, (-2, 0), 1 << 3),
                stage(4, 4, 2, (0, 0), 1), stage(24, 16, 2, (0, 0), 1), stage(256, 16, 2, (0, 0), 1), stage(16, 16, 3, (0, 0), 1), stage(16, 16, 2, (0, 0), 1, "SSE")):
        with pytest.raises(VVHipError):
            hp.me_plan_create(e_int, e_c, bad, e_it, 10, 0)
    # ... while the same base is fine when no masked position leaves the staged rows
    hp.me_plan_destroy(hp.me_plan_create(e_int, e_c, stage(16, 16, 2, (0, 2), 0b1010), e_it, 10, 0))
    hp.me_plan_destroy(hp.me_plan_create(e_int, e_c, stage(16, 16, 1, (3, -3), 511), e_it, 10, 0))          # (a quarter-sample stage stays inside for every base)
    it = np.zeros(1, RP.ME_ITEM)
    for w, h, f, ss in ((3, 8, "SAD", 0), (8, 256, "SAD", 0), (8, 8, "SSE", 1), (1, 1, "SAD", 0)):
        it[0] = (0, 0, 0, 1, DF[f], ss, w, h)
        with pytest.raises(VVHipError):
            hp.me_plan_create(e_int, e_c, e_st, it, 10, 0)
    ij = np.zeros(1, RP.ME_INT_JOB)
    ij[0] = (0, 0, 4, 8, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1)                                           # 4-wide blocks are table calls, not windows
    with pytest.raises(VVHipError):
        hp.me_plan_create(ij, np.zeros(1, RP.ME_CAND), e_st, e_it, 10, 0)


def test_stage_without_evaluated_positions_reads_nine_zeros(env):
    """a stage whose mask is empty still gets its nine costs written (0) by every run — no stale memory, no clearing launch"""
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    hp, _ = env
    rng = np.random.default_rng(5)
    org_np, ref_np, ref_pad, org, ref, M = _setup(hp, rng, 10, False)
    sj = np.zeros(3, RP.ME_STAGE_JOB)
    sj[0] = (40 * org.stride + 40, 40 * ref.stride + 40, 16, 16, 0, 1, 2, 2, 0, DF["HAD"], 0, 0, 0, 0)            # empty mask
    sj[1] = (40 * org.stride + 40, 40 * ref.stride + 40, 64, 64, 0, 1, 2, 2, 0, DF["HAD_fast"], 0, 0, 0, 0)      # empty mask, a stage shared by two waves
    sj[2] = (40 * org.stride + 40, 40 * ref.stride + 40, 16, 16, 0, 1, 2, 2, 0, DF["HAD"], 0, 0, 1, 0)
    planes = [(org.storage.data_ptr() + 2 * org.origin, org.stride), (ref.storage.data_ptr() + 2 * ref.origin, ref.stride)]
    _, sc, _ = _run_plan(hp, planes, np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), sj, np.zeros(0, RP.ME_ITEM), None, 10)
    assert (sc[0] == 0).all() and (sc[1] == 0).all() and sc[2, 0] > 0 and (sc[2, 1:] == 0).all(), sc


def test_plan_run_rejects_a_short_plane_table(env):
    import torch
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF
    from vvenc_amd.lib import VVHipError
    hp, _ = env
    it = np.zeros(1, RP.ME_ITEM)
    it[0] = (0, 0, 0, 3, DF["SAD"], 0, 8, 8)
    plan = hp.me_plan_create(np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), np.zeros(0, RP.ME_STAGE_JOB), it, 10, 0)
    buf = torch.zeros(4096, dtype=torch.int16, device=hp.device)
    tab = (RP.MePlane * 16)()
    for k in range(4):
        tab[k] = RP.MePlane(buf.data_ptr(), 64, 0)
    out = torch.zeros(1, dtype=torch.int64, device=hp.device)
    with pytest.raises(VVHipError):
        hp.me_plan_run(plan, tab, 3, out, out, out)                                              # the item names plane 3, the table has 3 entries
    hp.me_plan_run(plan, tab, 4, out, out, out)
    torch.cuda.synchronize()
    hp.me_plan_destroy(plan)
