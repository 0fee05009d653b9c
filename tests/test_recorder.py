"""CPU tier: the work-list recorder of the binding (bindings/vvenc/vvenc_hip_recorder.*, hook bit 131072) and the host side of the replay (vvenc_amd/recorded.py, replay.py).

The encoder built with the binding records its own hot-path calls while it encodes on the CPU kernels (no device involved).  Checked here:
  * recording does not change the encode (bitstream = the plain reference encoder's);
  * the record layouts the loader mirrors, the picture header, the sample-pair counter (what CommonLib/SearchSpaceCounter.cpp counts);
  * CONSISTENCY with the reference: the reference's own x86-SIMD table entries, driven over the recorded lists on host copies of the recorded planes (oracle/_ref's
    vvref_run_recorded_mt: plain distortion lists, and the sub-pel stages rebuilt from their descriptors with the reference's InterpolationFilter), return exactly the costs
    the encoder computed — i.e. operands, positions, tap sets and masks were recorded right;
  * the host lists built for the device (window jobs, stage jobs, items, TU groups) cover every recorded call."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import e2e_util  # noqa: E402


def need():
    if not (os.path.exists(e2e_util.REF_SO) and os.path.exists(e2e_util.REF_HIP_SO)):
        pytest.skip("needs oracle/_ref/libvvenc_ref.so and bindings/vvenc/_build/libvvenc_hip_enc.so (built from /root/reference)")


@pytest.fixture(scope="module")
def recording(tmp_path_factory):
    need()
    from vvenc_amd import recorded as R
    d = tmp_path_factory.mktemp("rec")
    res = R.record(str(d), 416, 240, 17, pocs=(), threads=4)
    return res, R.load_dir(str(d))


def test_recording_leaves_the_encode_unchanged(recording):
    import e2e_fps
    res, pics = recording
    cpu = e2e_fps.run(dict(w=416, h=240, frames=17, threads=4, mask=0))
    assert res["md5"] == cpu["md5"] and res["bytes"] == cpu["bytes"]
    assert len(pics) == 17


def test_record_layouts_and_counters(recording):
    from vvenc_amd import recorded as R
    _, pics = recording
    assert R.DT["me"].itemsize == 36 and R.DT["cand"].itemsize == 24 and R.DT["stage"].itemsize == 96 and R.DT["dist"].itemsize == 44 and R.DT["tu"].itemsize == 24 and R.DT["dmvr"].itemsize == 48
    inter = [p for p in pics.values() if p.slice_type != 2]
    assert inter and any(p.slice_type == 2 for p in pics.values())
    for p in pics.values():
        s = p.summary()
        # every table call is either an integer candidate, an evaluated sub-pel position or a plain call: the three add up to the counter (w x h per call, DMVR excluded)
        assert s["int_candidate_pairs"] + s["subpel_pairs"] + s["other_pairs"] == p.sample_pairs, s
        assert int(p.meta["sample_pairs_in_subpel_stages"]) == s["subpel_pairs"]
        assert p.planes[0]["kind"] == 0 and p.planes[0]["width"] == 416 and all(p.planes[i]["kind"] == 1 for i in range(3, p.planes.size))
    for p in inter:
        assert p.me.size and p.cand.size and p.stage.size and p.tu.size
        assert (p.me["nCand"].sum() == p.cand.size) and (p.me["nStage"].sum() == p.stage.size)
        assert set(np.unique(p.stage["iFrac"])) <= {1, 2} and (p.stage["reduceTap"] == 2).all() and (p.stage["hadMode"] == 2).all()      # preset faster: 4-tap search filter, HAD_fast


def test_reference_entries_reproduce_the_recorded_costs(recording):
    import bench
    from vvenc_amd.replay import RecordedLists
    _, pics = recording
    n_dist = n_stage = 0
    for poc, pic in pics.items():
        L = RecordedLists(pic)
        assert L.items_dropped == 0 and L.nothing_dropped and L.mask_items.size == 0
        # coverage: every recorded candidate is a window candidate or an item, every stage a stage job, every TU in a group
        assert L.plan_cands.size + L.items.size == pic.cand.size + pic.dist.size
        assert L.stage_jobs.size == pic.stage.size and sum(g["n"] for g in L.tu_groups) == pic.tu.size
        J = bench.ReferenceJobs(L, with_outputs=True)
        J.run(4, 1)
        exp = np.concatenate([L.cand_expected, L.item_expected])
        got = np.zeros(exp.size, np.uint64)
        for sel, out in J.dist_groups:
            got[sel] = out
        assert np.array_equal(got, exp), (poc, int((got != exp).sum()))
        for sel, out in J.stage_groups:
            assert np.array_equal(out[L.stage_evaluated[sel]], L.stage_expected[sel][L.stage_evaluated[sel]]), poc
        n_dist += exp.size
        n_stage += int(L.stage_evaluated.sum())
    assert n_dist > 50000 and n_stage > 3000, (n_dist, n_stage)


def test_medium_preset_lists_are_complete_and_consistent(tmp_path):
    """preset medium (CTU 128 + multi-type tree, GEO): rectangular blocks up to 128x128 and masked SADs.  The host lists hold EVERY recorded call (nothing dropped), and the
    reference's own entries driven over them — generic-shape SADs, the rectangular Hadamard tiles, DF_SAD_WITH_MASK on the recorded weight blocks, stages rebuilt from their
    descriptors — return the costs the encoder computed"""
    need()
    import bench
    from vvenc_amd import recorded as R
    from vvenc_amd.replay import RecordedLists
    R.record(str(tmp_path), 416, 240, 9, pocs=(4, 8), threads=4, preset="medium")
    pics = R.load_dir(str(tmp_path))
    assert len(pics) == 2
    shapes, n_mask = set(), 0
    for poc, pic in pics.items():
        L = RecordedLists(pic)
        assert L.nothing_dropped, L.dropped
        assert L.plan_cands.size + L.items.size + L.mask_items.size == pic.cand.size + pic.dist.size
        assert L.stage_jobs.size == pic.stage.size and sum(g["n"] for g in L.tu_groups) == pic.tu.size
        shapes |= set(zip(L.stage_jobs["width"].tolist(), L.stage_jobs["height"].tolist())) | set(zip(L.int_jobs["width"].tolist(), L.int_jobs["height"].tolist()))
        assert any(g["w"] != g["h"] for g in L.tu_groups)
        J = bench.ReferenceJobs(L, with_outputs=True)
        J.run(4, 1)
        exp = np.concatenate([L.cand_expected, L.item_expected])
        got = np.zeros(exp.size, np.uint64)
        for sel, out in J.dist_groups:
            got[sel] = out
        assert np.array_equal(got, exp), (poc, int((got != exp).sum()))
        gm = np.zeros(L.mask_items.size, np.uint64)
        for sel, out in J.mask_groups:
            gm[sel] = out
        assert np.array_equal(gm, L.mask_expected), (poc, int((gm != L.mask_expected).sum()))
        n_mask += gm.size
        for sel, out in J.stage_groups:
            assert np.array_equal(out[L.stage_evaluated[sel]], L.stage_expected[sel][L.stage_evaluated[sel]]), poc
    assert any(w != h for w, h in shapes) and any(w == 128 for w, h in shapes) and n_mask > 10000, (shapes, n_mask)


def test_fast_preset_lists_are_complete_and_consistent(tmp_path):
    """preset fast (BASELINE configs[4]'s preset: affine, BDOF, MMVD, two references per list, DepQuant, LFNST — vvencCfg.cpp:2751-2819): the host lists hold EVERY recorded
    call (nothing dropped) and the reference's own entries driven over them return the costs the encoder computed — the CPU twin of tests/test_gpu_replay.py's fast-preset replay"""
    need()
    import bench
    from vvenc_amd import recorded as R
    from vvenc_amd.replay import RecordedLists
    R.record(str(tmp_path), 416, 240, 9, pocs=(4, 8), threads=4, preset="fast")
    pics = R.load_dir(str(tmp_path))
    assert len(pics) == 2
    refs, n_calls = set(), 0
    for poc, pic in pics.items():
        refs |= set(pic.me["refPlane"].tolist())
        L = RecordedLists(pic)
        assert L.nothing_dropped, L.dropped
        assert L.plan_cands.size + L.items.size + L.mask_items.size == pic.cand.size + pic.dist.size
        assert L.stage_jobs.size == pic.stage.size and sum(g["n"] for g in L.tu_groups) == pic.tu.size
        J = bench.ReferenceJobs(L, with_outputs=True)
        J.run(4, 1)
        exp = np.concatenate([L.cand_expected, L.item_expected])
        got = np.zeros(exp.size, np.uint64)
        for sel, out in J.dist_groups:
            got[sel] = out
        assert np.array_equal(got, exp), (poc, int((got != exp).sum()))
        for sel, out in J.stage_groups:
            assert np.array_equal(out[L.stage_evaluated[sel]], L.stage_expected[sel][L.stage_evaluated[sel]]), poc
        n_calls += exp.size
    assert len(refs) >= 2 and n_calls > 40000, (refs, n_calls)


def test_early_exit_partial_sums_are_flagged_not_recorded(tmp_path):
    """64-wide SADs: the x86 row returns a partial sum once it exceeds the best cost so far (x86/RdCostX86.h:390-410); the record holds the full value and flags the call"""
    need()
    from vvenc_amd import recorded as R
    R.record(str(tmp_path), 416, 240, 9, pocs=(4,), threads=1)
    pic = R.load_dir(str(tmp_path))[4]
    c = pic.cand
    w = pic.me["w"][c["me"]]
    flagged = c["pad"] & 0xff
    assert (flagged[w < 64] == 0).all()
    assert flagged[w == 64].any()                                   # the synthetic pan makes the search leave its start point: some candidates do exit early
    # a flagged value is what the full sum is: recompute a few from the planes
    org, _ = pic.plane_array(0)
    k = np.nonzero((flagged == 1) & (pic.me["patternPool"][c["me"]] < 0))[0][:20]
    for i in k:
        m = pic.me[c["me"][i]]
        arr, mg = pic.plane_array(int(m["refPlane"]))
        o = org[m["cuY"]:m["cuY"] + 64, m["cuX"]:m["cuX"] + 64].astype(np.int64)
        r = arr[c["y"][i] + mg:c["y"][i] + mg + 64, c["x"][i] + mg:c["x"][i] + mg + 64].astype(np.int64)
        ss = int(c["subShift"][i])
        assert int(np.abs(o[::1 << ss] - r[::1 << ss]).sum()) << ss == int(c["cost"][i])


def test_window_kernel_byte_accounting(recording):
    """the three byte counts bench.py reports for the integer windows (vvenc_amd/replay.py): every listed candidate with its whole block (round 4's figure: 3.6 x the HBM peak at
    4K) >= every DISTINCT position with its block (a work rate) >= the job's window read once + its block + 8 B per distinct position (SURVEY 8d "with window reuse": the class's
    algorithmic bytes) — recomputed here job by job with plain Python sets and boxes"""
    from vvenc_amd.replay import RecordedLists
    _, pics = recording
    checked = 0
    for poc, pic in pics.items():
        L = RecordedLists(pic, unique_bytes=False)
        if not L.plan_cands.size:
            continue
        assert L.alg_bytes_int_window <= L.alg_bytes_int_per_position <= L.alg_bytes_int_all_candidates
        assert L.int_positions_distinct <= L.plan_cands.size and L.alg_bytes_by_kernel["ME_int"] == L.alg_bytes_int_window
        tot_all = tot_pos = tot_win = n_distinct = 0
        for j in L.int_jobs:
            w, h, ss = int(j["width"]), int(j["height"]), int(j["sub_shift"])
            c = L.plan_cands[int(j["first_cand"]):int(j["first_cand"]) + int(j["n_cand"])]
            pos = {(int(a), int(b)) for a, b in zip(c["dx"], c["dy"])}
            rows = h >> ss
            tot_all += c.size * (4 * w * rows + 8)
            tot_pos += len(pos) * (4 * w * rows + 8)
            n_distinct += len(pos)
            cols = max(p[0] for p in pos) - min(p[0] for p in pos) + w
            touched = set()                                              # reference rows the job's candidates read (subShift 1: every second row from the candidate's dy on)
            for _, dy in pos:
                touched.update(range(dy, dy + h, 1 << ss))
            tot_win += 2 * cols * len(touched) + 2 * w * rows + 8 * len(pos)
        assert (tot_all, tot_pos, n_distinct) == (L.alg_bytes_int_all_candidates, L.alg_bytes_int_per_position, L.int_positions_distinct), poc
        # the library's figure is the job's BOUNDING window (per row parity under subShift: what the kernel stages in LDS) — never below the rows the candidates touch, and not
        # far above them (a search's late, far-apart candidates leave gaps inside the box)
        assert tot_win <= L.alg_bytes_int_window <= 1.5 * tot_win, (poc, tot_win, L.alg_bytes_int_window)
        checked += 1
    assert checked >= 8
