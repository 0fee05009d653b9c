"""CPU tier: the host logic of bench.py's north-star leg C (tools/bench_mctf.py) — the GOP cadence (which steps queue which filtered picture with how many references,
MCTF.cpp:745-800 / vvencCfg.cpp:1498-1509), the filter's algorithmic-byte formula against a block-by-block count, and the CPU baseline's picture-level entry of the compiled
reference (oracle/_ref: vvref_mctf_cycle_timed) against its own per-pair entry and against the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_cadence_follows_the_gop():
    import bench_mctf as BM
    from bench_common import STEP_LAYERS
    jobs = [(s, BM.job_of_step(s)) for s in range(32) if BM.job_of_step(s)]
    assert [(s, j[1]) for s, j in jobs] == [(0, 32), (8, 8), (20, 24), (24, 16)]
    for s, j in jobs:
        assert STEP_LAYERS[s] == j[0] and j[1] % 8 == 0
        assert len(j[2]) == (4 if j[1] % 16 == 0 else 2)                    # MCTFSpeed 4: 2 references per side for the 16- / 32-multiples, 1 for the 8-multiples
        assert sorted(abs(r - j[1]) for r in j[2]) == ([1, 1, 2, 2] if len(j[2]) == 4 else [1, 1])
    assert BM.motion_estimations_per_cycle() == 12
    assert BM.job_of_step(32 + 8)[1] == 8 and BM.job_of_step(5) is None
    # the strengths of vvencCfg.cpp:1503-1507 at QP 32, GOP 32 (numFrames 3)
    qp, n = 32, 3
    exp = [min(2.0, max(0.0, (qp - 4.0) / 8.0)) / (n - i) for i in range(n)]
    exp[n - 1] = min(1.5, max(0.0, (qp - 4.0) * 3.0 / 32.0))
    by_poc = {j[1]: j[3] for j in BM.JOBS}
    assert abs(by_poc[8] - exp[0]) < 1e-12 and abs(by_poc[16] - exp[1]) < 1e-12 and abs(by_poc[32] - exp[2]) < 1e-12


@pytest.mark.parametrize("geom", [(416, 240, 2), (1920, 1080, 4), (200, 136, 3)])
def test_filter_bytes_formula(geom):
    import bench_mctf as BM
    w, h, n = geom
    total = 0
    for cs in (0, 1, 1):
        pw, ph, b = w >> cs, h >> cs, 16 >> cs
        for by in range(0, ph, b):
            for bx in range(0, pw, b):
                bw, bh = min(b, pw - bx), min(b, ph - by)
                total += 2 * bw * bh + 2 * bw * bh + n * ((bw + 5) * (bh + 5) * 2 + 24)
    assert BM.apply_alg_bytes(w, h, n) == total


@pytest.mark.ref
def test_reference_cycle_entry_against_its_pair_entry_and_the_oracle(oracle):
    from oracle.oracle import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/libvvenc_ref.so not built (needs /root/reference)")
    import bench_mctf as BM
    ref = RefLib(1)
    pics = BM.cycle_pictures(416, 240)
    assert [len(p[1]) for p in pics] == [4, 4, 2, 2]
    f0, o0, s0 = ref.mctf_cycle_timed(pics, 0)
    f2, o2, s2 = ref.mctf_cycle_timed(pics, 2)
    assert s0[0] > 0 and s0[1] > 0 and s2[0] > 0
    for p in range(4):
        for a, b in zip(f0[p], f2[p]):
            assert all(np.array_equal(a[k], b[k]) for k in ("x", "y", "error", "rmsme", "overlap"))          # threads change nothing
        assert all(np.array_equal(a, b) for a, b in zip(o0[p], o2[p]))
    # its fields = the per-pair entry's = the oracle's; its filtered planes = the whole-picture entry's on those fields
    cur, refs, idx, strength = pics[2]
    for k, r in enumerate(refs):
        e = ref.mctf_me(cur[0], r[0], 10, 16, 4, False)[4]
        o = oracle.mctf_me(cur[0], r[0], 10, 16, 4, False)[4]
        for f in ("x", "y", "error", "rmsme", "overlap"):
            assert np.array_equal(f0[2][k][f], e[f]) and np.array_equal(e[f], o[f]), (k, f)
    b = ref.mctf_bilateral(cur, refs, f0[2], idx, 10, 32, 16, False, True, strength)
    assert all(np.array_equal(x, y) for x, y in zip(b, o0[2]))


@pytest.mark.ref
def test_cpu_baseline_mctf_object():
    from oracle.oracle import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/libvvenc_ref.so not built (needs /root/reference)")
    import bench_mctf as BM
    r = BM.cpu_baseline_mctf(416, 240, 2)
    assert r["kind"] == "reference" and r["cores"] == 2 and r["value"] > 0 and r["value_1thread"] > 0 and "sample" in r
    assert abs(r["value_1thread"] * 32 - (r["me_ms_per_cycle_1thread"] + r["filter_ms_per_cycle_1thread"])) < 0.2
