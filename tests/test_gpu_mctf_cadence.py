"""GPU tier: north-star leg C in the measurement (round 6) — the asynchronous motion-estimation entry, the library's scored-candidate counters and per-class timers, and
tools/bench_mctf.py's GOP-cadence workload (fields and filtered planes against the reference compiled here / the oracle)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from hip_backend import HipBackend
    return HipBackend()


def _pair(seed, h, w, shift=(5, 2), noise=6):
    from test_gpu_parity import synth_pair
    return synth_pair(np.random.default_rng(seed), h, w, shift=shift, noise=noise)


def test_async_motion_estimation_equals_the_synchronous_call(hip, oracle):
    """vvhip_mctf_motion_estimation_async: same launches, no host wait — on a forked context's own stream while the default stream runs other work; the fields are the
    synchronous call's (and the oracle's) once the stream has passed them"""
    import torch
    from vvenc_amd.hotpath import HotPath
    hp = hip.hp
    w, h = 416, 240
    org, r0 = _pair(11, h, w)
    _, r1 = _pair(11, h, w, shift=(-2, 1), noise=3)
    pc, prs = hp.plane(org, 128), [hp.plane(r0, 128), hp.plane(r1, 128)]
    sync_out, dims = hp.mctf_motion_estimation(pc, prs, 10, 16, 4, False)
    lane = hp.fork(torch.cuda.Stream())
    async_out, _ = lane.mctf_motion_estimation(pc, prs, 10, 16, 4, False, wait=False)
    busy = torch.ones(1 << 20, device=hp.device)
    for _ in range(4):
        busy = busy * 1.0001                       # (other work on the default stream while the lane's launches run)
    torch.cuda.synchronize()
    for k, ref in enumerate((r0, r1)):
        a, s = HotPath.mv_to_numpy(async_out[k], dims), HotPath.mv_to_numpy(sync_out[k], dims)
        exp = oracle.mctf_me(org, ref, 10, 16, 4, False)[4]
        for f in ("x", "y", "error", "rmsme", "overlap"):
            assert np.array_equal(a[f], s[f]) and np.array_equal(a[f], exp[f]), (k, f)
    # a second call on the same lane reuses its scratch in stream order
    lane.mctf_motion_estimation(pc, prs[::-1], 10, 16, 4, False, out=async_out, wait=False)
    torch.cuda.synchronize()
    assert all(np.array_equal(HotPath.mv_to_numpy(async_out[0], dims)[f], HotPath.mv_to_numpy(sync_out[1], dims)[f]) for f in ("x", "y", "error"))
    lane.close()


@pytest.mark.parametrize("cfg", [(416, 240, False), (640, 360, True)])
def test_scored_candidate_counters_against_the_references_schedule(hip, oracle, cfg):
    """the counters behind the MCTF rows' algorithmic bytes: deterministic, consistent (bytes = candidates x the block's figure), never above the number of motionErrorLuma
    calls the reference's schedule makes on the same pictures (oracle.mctf_me_counted) and not far below it (the device only skips candidates that cannot win: a vector equal
    to the current best, the above / left vector equal to one already scored); switching the counters on changes no result"""
    from vvenc_amd.hotpath import HotPath
    hp = hip.hp
    w, h, add = cfg
    org, ref = _pair(77 + w, h, w)
    pc, pr = hp.plane(org, 128), [hp.plane(ref, 128)]
    plain, dims = hp.mctf_motion_estimation(pc, pr, 10, 16, 4, add)
    hp.mctf_set_stats(True)
    counted, _ = hp.mctf_motion_estimation(pc, pr, 10, 16, 4, add)
    st1 = hp.mctf_get_stats()
    hp.mctf_set_stats(True)                                             # (re-arming zeroes the counters)
    hp.mctf_motion_estimation(pc, pr, 10, 16, 4, add, out=counted)
    st2 = hp.mctf_get_stats()
    hp.mctf_set_stats(False)
    assert st1 == st2
    a, b = HotPath.mv_to_numpy(plain[0], dims), HotPath.mv_to_numpy(counted[0], dims)
    assert all(np.array_equal(a[f], b[f]) for f in ("x", "y", "error", "rmsme", "overlap"))
    exp, ref_calls = oracle.mctf_me_counted(org, ref, 10, 16, 4, add)
    assert all(np.array_equal(a[f], exp[4][f]) for f in ("x", "y", "error"))
    tot = {k: sum(st1[p][k] for p in st1) for k in ("int", "int_bytes", "frac", "frac_bytes", "grid", "grid_window_bytes", "ring", "ring_window_bytes")}
    for ph in st1.values():
        assert (ph["int"] == 0) == (ph["int_bytes"] == 0) and (ph["frac"] == 0) == (ph["frac_bytes"] == 0) and (ph["grid"] == 0) == (ph["grid_window_bytes"] == 0)
        # a dense grid's window form is never more than its positions at 4 w h each, and at least a window of range 3 + the block per 81 positions
        assert ph["grid_window_bytes"] <= ph["grid"] * 4 * 32 * 32 and ph["grid_window_bytes"] * 81 >= ph["grid"] * ((32 + 6) ** 2 * 2 + 2048)
        assert ph["int_bytes"] <= ph["int"] * 4 * 32 * 32 and ph["frac_bytes"] <= ph["frac"] * ((32 + 3) * (32 + 3) * 2 + 2 * 32 * 32)
        assert ph["int_bytes"] >= ph["int"] * 4 * 8 * 8 and ph["frac_bytes"] >= ph["frac"] * ((8 + 3) * (8 + 3) * 2 + 2 * 8 * 8)
    assert st1["search"]["int"] > 0 and st1["search"]["ring"] > 0
    # a ring's window form is below its positions at the per-candidate figure (8 positions share the window) and above one window per 8 positions
    ring_pc = tot["ring"] * ((16 + 3) * (16 + 3) * 2 + 2 * 16 * 16)
    assert tot["ring_window_bytes"] < ring_pc and tot["ring_window_bytes"] * 8 >= tot["ring"] * ((8 + 4) * (8 + 4) * 2 + 2 * 8 * 8)
    n_dev, n_ref = tot["int"] + tot["frac"] + tot["grid"] + tot["ring"], ref_calls["int"] + ref_calls["frac"]
    assert n_dev <= n_ref, (tot, ref_calls)
    assert n_dev >= 0.75 * n_ref, (tot, ref_calls)
    assert tot["int_bytes"] + tot["grid"] * 4 * 32 * 32 + tot["frac_bytes"] + tot["ring"] * ((8 + 3) * (8 + 3) * 2 + 2 * 8 * 8) <= ref_calls["int_bytes"] + ref_calls["frac_bytes"]   # >= per candidate
    assert tot["int_bytes"] + tot["grid_window_bytes"] + tot["frac_bytes"] + tot["ring_window_bytes"] < ref_calls["int_bytes"] + ref_calls["frac_bytes"]      # the window form (what the rows use)
    with pytest.raises(Exception):
        hp.mctf_get_stats()                                             # off: asking for them is an error, not zeros


def test_per_class_timers(hip):
    hp = hip.hp
    org, ref = _pair(5, 240, 416)
    pc, pr = hp.plane(org, 128), [hp.plane(ref, 128)]
    hp.mctf_set_timing(True)
    hp.mctf_motion_estimation(pc, pr, 10, 16, 4, False)
    t = hp.mctf_last_times()
    hp.mctf_set_timing(False)
    assert len(t) == 5 and all(x >= 0 for x in t) and t[0] > 0 and t[2] > 0 and sum(t) < 50.0
    with pytest.raises(Exception):
        hp.mctf_last_times()


def test_gop_cadence_workload_parity_and_bytes(hip):
    """tools/bench_mctf.MctfCadence at 416x240 (unit 16 like the BASELINE resolutions): every job's fields and filtered planes against the checker bench.py uses (the
    reference compiled here where oracle/_ref exists, the oracle otherwise); the algorithmic bytes of a cycle are positive for every class and the filter's follow the formula"""
    import torch
    import bench_mctf as BM
    hp = hip.hp
    mc = BM.MctfCadence(hp, 416, 240, lane=hp.fork(torch.cuda.Stream()))
    res = mc.parity()
    assert res["status"] == "bit-exact" and res["fields"] == 12 and res["planes"] == 12, res
    cyc = mc.count()
    assert all(cyc[k] > 0 for k in ("MCTF_search", "MCTF_nb", "MCTF_apply")), cyc
    assert cyc["MCTF_apply"] == sum(BM.apply_alg_bytes(416, 240, len(j[2])) for j in BM.JOBS)
    assert mc.candidates_per_cycle["int"] > 0 and mc.candidates_per_cycle["frac"] > 0
    # the cadence: issuing the 32 steps of a cycle queues exactly the four jobs
    issued = [mc.issue_step(s) for s in range(32)]
    torch.cuda.synchronize()
    assert [j[1] for j in issued if j] == [32, 8, 24, 16]
    mc.lane.close()


@pytest.mark.parametrize("kind", ["noise", "periodic", "flat_noise", "object"])
def test_fixed_point_phase_b_on_content_that_fires_the_recurrence(hip, oracle, kind):
    """phase B as a fixed-point iteration (mctf.hip, meFixKernel): content on which the above / left tests DO change many blocks and chains of changes run through the
    field — pure noise, a periodic texture with many equal matches, near-flat pictures with a little noise, a moving object on a static background — against the oracle's
    sequential row loop, every level (the coarser levels' fields feed the next level's candidates: one wrong vector anywhere shows up)"""
    rng = np.random.default_rng({"noise": 1, "periodic": 2, "flat_noise": 3, "object": 4}[kind])
    w, h = 416, 240
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "noise":
        org = rng.integers(0, 1024, (h, w)).astype(np.int16)
        ref = rng.integers(0, 1024, (h, w)).astype(np.int16)
    elif kind == "periodic":
        base = (512 + 300 * np.sin(xx * 2 * np.pi / 8.0) * np.sin(yy * 2 * np.pi / 8.0))
        org = np.clip(base + rng.normal(0, 2, base.shape), 0, 1023).astype(np.int16)
        ref = np.clip(np.roll(base, (3, 5), (0, 1)) + rng.normal(0, 2, base.shape), 0, 1023).astype(np.int16)
    elif kind == "flat_noise":
        org = np.clip(500 + rng.normal(0, 1.5, (h, w)), 0, 1023).astype(np.int16)
        ref = np.clip(500 + rng.normal(0, 1.5, (h, w)), 0, 1023).astype(np.int16)
    else:
        bg = np.clip(400 + 100 * np.sin(xx / 9.0) + rng.normal(0, 4, (h, w)), 0, 1023)
        org, ref = bg.copy(), bg.copy()
        tex = 700 + 200 * np.sin(np.mgrid[0:96, 0:96][1] / 3.0) * np.cos(np.mgrid[0:96, 0:96][0] / 4.0)
        org[60:156, 100:196] = tex
        ref[49:145, 123:219] = tex
        org, ref = org.astype(np.int16), ref.astype(np.int16)
    exp = oracle.mctf_me(org, ref, 10, 16, 4, False)
    got = hip.mctf_me_levels(org, ref, 10, 16, 4, False)
    for k in range(5):
        if exp[k] is None:
            continue
        for f in ("x", "y", "error"):
            assert np.array_equal(got[k][f], exp[k][f]), (kind, k, f, np.argwhere(got[k][f] != exp[k][f])[:5])
    full = hip.mctf_me(org, ref, 10, 16, 4, False)[4]
    for f in ("x", "y", "error", "rmsme", "overlap"):
        assert np.array_equal(full[f], exp[4][f]), (kind, f)
