"""-m gpu: motion-search plans (vvenc_amd/csrc/me.hip) and the replay of work lists recorded from the reference encoder.

  * plans against the per-function kernels of dist.hip / interp.hip (themselves pinned to the reference by the golden vectors) on seeded random jobs: every block size,
    split windows, all three tap sets, SAD / HAD / HAD_fast stages, signed bi-prediction patterns, every table-call function incl. 4x4;
  * recorded lists (the encoder built with the binding records them on the box: no /root/reference needed) replayed through the C ABI and compared with the values the REAL
    encoder computed — integer SADs, sub-pel Hadamards, table calls, DMVR vectors and costs — and, for the TU lists, with the reference's x86-SIMD entries;
  * the untested corners VERDICT r2 named: the row-wavefront fallback of the MCTF sweep, 8K planes through the tiled / shifted list paths.
Everything is bit-exact (tolerance 0)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def hp():
    from vvenc_amd.hotpath import HotPath
    return HotPath()


def _plan_vs_kernels(hp, seed, sizes, filter_mode, func_stage, max_window, signed_org=False, n_jobs=40, bd=10):
    """random integer jobs / stages / items on one original and one reference plane, scored by a plan and by the per-function kernels"""
    import torch
    from vvenc_amd import replay as RP
    from vvenc_amd.hotpath import DF, SUBPEL_DTYPE
    rng = np.random.default_rng(seed)
    H, W, M = 256, 320, 80
    top = 1 << bd
    lo = -(top - 1) if signed_org else 0
    org = hp.plane(rng.integers(lo, 2 * top - 1 if signed_org else top, size=(H, W), dtype=np.int16), 8)
    ref = hp.plane(rng.integers(0, top, size=(H, W), dtype=np.int16), M)
    jobs, cands, stages, items = [], [], [], []
    for _ in range(n_jobs):
        S = int(rng.choice(sizes))
        x, y = int(rng.integers(0, W - S)), int(rng.integers(0, H - S))
        n = int(rng.integers(1, 30))
        spread = int(rng.choice([1, 2, 6, 40]))
        dx = rng.integers(-spread, spread + 1, n)
        dy = rng.integers(-spread, spread + 1, n)
        jobs.append((y * org.stride + x, y * ref.stride + x, S, S, 0, 1, int(S > 8 and rng.integers(0, 2)), len(cands), n))
        cands += [(int(a), int(b)) for a, b in zip(dx, dy)]
        i_frac = int(rng.choice([1, 2]))
        bq = (0, 0) if i_frac == 2 else (int(rng.integers(-2, 3)), int(rng.integers(-2, 3)))
        mask = int(rng.integers(1, 512))
        bx, by = x + int(rng.integers(-4, 5)), y + int(rng.integers(-4, 5))
        stages.append((y * org.stride + x, by * ref.stride + bx, S, S, 0, 1, i_frac, filter_mode, int(rng.integers(0, 2)) if filter_mode != 2 or rng.integers(0, 4) == 0 else 0, func_stage, bq[0], bq[1], mask))
    for _ in range(6 * n_jobs):
        S = int(rng.choice([4, 8, 16, 32, 64]))
        f = int(rng.choice([DF["SSE"], DF["SAD"], DF["HAD"], DF["HAD_fast"], DF["HAD_2SAD"]]))
        x, y, cx, cy = int(rng.integers(0, W - S)), int(rng.integers(0, H - S)), int(rng.integers(-16, W - S + 16)), int(rng.integers(-16, H - S + 16))
        items.append((y * org.stride + x, cy * ref.stride + cx, 0, 1, f, int(f == DF["SAD"] and S > 8 and rng.integers(0, 2)), S, S))
    ij = np.zeros(len(jobs), RP.ME_INT_JOB)
    for k, j in enumerate(jobs):
        ij[k] = (j[0], j[1], j[2], j[3], j[4], j[5], j[6], 0, 0, 0, 0, 0, j[7], j[8])
    pc = np.array(cands, RP.ME_CAND)
    sj = np.zeros(len(stages), RP.ME_STAGE_JOB)
    for k, s in enumerate(stages):
        sj[k] = (s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], s[9], s[10], s[11], s[12], 0)
    it = np.zeros(len(items), RP.ME_ITEM)
    for k, v in enumerate(items):
        it[k] = v
    plan = hp.me_plan_create(ij, pc, sj, it, bd, max_window)
    tab = (RP.MePlane * 16)()
    tab[0] = RP.MePlane(org.storage.data_ptr() + 2 * org.origin, org.stride, 0)
    tab[1] = RP.MePlane(ref.storage.data_ptr() + 2 * ref.origin, ref.stride, 0)
    cc = torch.zeros(len(cands), dtype=torch.int64, device=hp.device)
    sc = torch.full((9 * len(stages),), -1, dtype=torch.int64, device=hp.device)
    ic = torch.zeros(len(items), dtype=torch.int64, device=hp.device)
    hp.me_plan_run(plan, tab, 2, cc, sc, ic)
    torch.cuda.synchronize()
    cc, sc, ic = cc.cpu().numpy(), sc.cpu().numpy().reshape(-1, 9), ic.cpu().numpy()
    hp.me_plan_destroy(plan)
    # ---- the same through the per-function kernels
    name = {v: k for k, v in DF.items()}
    for k, j in enumerate(jobs):
        its = np.array([(j[0], j[1] + dy * ref.stride + dx) for dx, dy in cands[j[7]:j[7] + j[8]]], np.int32)
        exp = hp.dist_batch("SAD", org, ref, hp.to_device(its), len(its), j[2], j[3], j[6], bd).cpu().numpy()
        assert np.array_equal(cc[j[7]:j[7] + j[8]], exp), ("integer job", k, j, cc[j[7]:j[7] + j[8]][:6], exp[:6])
    from vvenc_amd.recorded import REFINE_H, REFINE_Q
    for k, s in enumerate(stages):
        refine = REFINE_H if s[6] == 2 else REFINE_Q
        q = (refine + np.array([s[10], s[11]])) * s[6] * 4                        # 1/16 sample
        sel = [p for p in range(9) if (s[12] >> p) & 1]
        its = np.zeros(len(sel), SUBPEL_DTYPE)
        for a, p in enumerate(sel):
            tx, ty = int(q[p][0]), int(q[p][1])
            its[a] = (s[0], s[1] + (ty >> 4) * ref.stride + (tx >> 4), tx & 15, ty & 15)
        exp = hp.subpel_dist_batch(name[s[9]], org, ref, hp.to_device(its), len(sel), s[2], s[3], bd, s[7], bool(s[8])).cpu().numpy()
        got = sc[k][sel]
        assert np.array_equal(got, exp), ("stage", k, s, got, exp)
        assert all(sc[k][p] == 0 for p in range(9) if p not in sel)
    for k, v in enumerate(items):
        exp = hp.dist_batch(name[v[4]], org, ref, hp.to_device(np.array([(v[0], v[1])], np.int32)), 1, v[6], v[7], v[5], bd).cpu().numpy()[0]
        assert ic[k] == exp, ("item", k, v, ic[k], exp)


@pytest.mark.parametrize("filter_mode,func", [(2, "HAD_fast"), (2, "HAD"), (2, "SAD"), (0, "HAD_fast"), (1, "HAD"), (0, "SAD")])
def test_me_plan_vs_per_function_kernels(hp, filter_mode, func):
    from vvenc_amd.hotpath import DF
    _plan_vs_kernels(hp, 100 + filter_mode * 7 + DF[func], (8, 16, 32, 64), filter_mode, DF[func], max_window=16)


def test_me_plan_split_windows_and_signed_patterns(hp):
    """candidates spread over +-40 samples with an 8-sample window limit (many windows per job); originals with the range of the bi-prediction pattern 2 * org - pred"""
    from vvenc_amd.hotpath import DF
    _plan_vs_kernels(hp, 7, (8, 16, 32, 64), 2, DF["HAD_fast"], max_window=8, signed_org=True)
    _plan_vs_kernels(hp, 8, (8, 64), 0, DF["HAD"], max_window=24, signed_org=True, n_jobs=12)


def test_me_plan_8_bit(hp):
    """bit depth 8 (BASELINE configs[0]): no headroom shift in the first pass, 14 - 8 = 6 bits in the second; every tap set"""
    from vvenc_amd.hotpath import DF
    _plan_vs_kernels(hp, 31, (8, 16, 32, 64), 2, DF["HAD_fast"], max_window=16, bd=8, n_jobs=24)
    _plan_vs_kernels(hp, 32, (8, 16, 64), 0, DF["HAD"], max_window=16, bd=8, n_jobs=16)
    _plan_vs_kernels(hp, 33, (16, 32), 1, DF["SAD"], max_window=16, bd=8, signed_org=True, n_jobs=16)


def test_me_plan_rejects_bad_jobs(hp):
    from vvenc_amd import replay as RP
    from vvenc_amd.lib import VVHipError
    ij = np.zeros(1, RP.ME_INT_JOB)
    ij[0] = (0, 0, 12, 12, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1)                      # 12x12: not a block size of the path
    with pytest.raises(VVHipError):
        hp.me_plan_create(ij, np.zeros(1, RP.ME_CAND), np.zeros(0, RP.ME_STAGE_JOB), np.zeros(0, RP.ME_ITEM), 10, 0)
    with pytest.raises(VVHipError):
        hp.me_plan_create(np.zeros(0, RP.ME_INT_JOB), np.zeros(0, RP.ME_CAND), np.zeros(0, RP.ME_STAGE_JOB), np.zeros(0, RP.ME_ITEM), 12, 0)      # bit depth 12: outside the packed tile's range


# ---------------------------------------------------------------------------------------------------------------- recorded lists
def _record(tmp, width, height, frames, pocs, preset="faster"):
    from vvenc_amd import recorded as R
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_util
    if not os.path.exists(e2e_util.REF_HIP_SO):
        pytest.skip("bindings/vvenc/_build/libvvenc_hip_enc.so (the encoder with the binding) is not built")
    R.record(str(tmp), width, height, frames, pocs=pocs, threads=4, preset=preset)
    return R.load_dir(str(tmp))


def _replay_and_check(hp, pics, with_tu_twin=True):
    import torch
    from vvenc_amd.replay import RecordedWorkload
    sys.path.insert(0, ROOT)
    import bench
    tot = {}
    for poc, pic in pics.items():
        wl = RecordedWorkload(hp, pic)
        wl.run()
        for k, (n, bad) in wl.check_against_recording().items():
            assert bad == 0, (poc, k, n, bad)
            tot[k] = tot.get(k, 0) + n
        assert wl.nothing_dropped, (poc, wl.dropped)                     # every recorded call is in a list of the plan / a TU group
        assert wl.plan_cands.size + wl.items.size + wl.mask_items.size == pic.cand.size + pic.dist.size and wl.stage_jobs.size == pic.stage.size
        assert sum(g["n"] for g in wl.tu_groups) == pic.tu.size
    if with_tu_twin:
        wls = {i: RecordedWorkload(hp, pic) for i, pic in enumerate(pics.values())}
        r = bench.parity_check(wls)
        assert r["status"] == "bit-exact", r
        tot["tus"] = r["checked"].get("tus_sse_abssum_last_needrdoq_levels", 0)
    return tot


def test_recorded_lists_small_clip(hp, tmp_path):
    """416x240, 17 frames, preset faster: every recorded picture replays bit-exactly (all list kinds present)"""
    pics = _record(tmp_path, 416, 240, 17, pocs=())
    assert len(pics) == 17
    tot = _replay_and_check(hp, pics)
    assert tot["integer_candidates"] > 10000 and tot["subpel_positions"] > 1000 and tot["table_calls"] > 10000 and tot["tus"] > 1000, tot


def test_recorded_lists_medium_preset(hp, tmp_path):
    """preset medium (BASELINE configs[3]'s preset: CTU 128, multi-type tree -> rectangular blocks 4..128, GEO's masked SADs, two references per list): NOTHING of the
    recording is left out, every recorded output replays bit-exactly, the TU lists (rectangular, DST-7, 2-wide chroma) against the reference's own entries"""
    pics = _record(tmp_path, 416, 240, 9, pocs=(4, 8), preset="medium")
    assert len(pics) == 2
    shapes = set()
    for pic in pics.values():
        shapes |= set(zip(pic.me["w"].tolist(), pic.me["h"].tolist()))
    assert any(w != h for w, h in shapes) and any(w == 128 or h == 128 for w, h in shapes), shapes          # the recording does hold what the test is about
    tot = _replay_and_check(hp, pics)
    assert tot["integer_candidates"] > 50000 and tot["subpel_positions"] > 5000 and tot["table_calls"] > 30000 and tot["masked_sad_calls"] > 10000 and tot["tus"] > 5000, tot


def test_recorded_lists_medium_preset_4k_picture(hp, tmp_path):
    """one 3840x2160 picture of a preset-medium encode (BASELINE configs[3] geometry: 128x128 blocks in numbers): nothing dropped, bit-exact"""
    pics = _record(tmp_path, 3840, 2160, 9, pocs=(4,), preset="medium")
    assert len(pics) == 1 and (pics[4].me["w"] == 128).sum() > 100
    tot = _replay_and_check(hp, pics)
    assert tot["integer_candidates"] > 200000 and tot["subpel_positions"] > 20000 and tot["tus"] > 10000, tot


def test_recorded_lists_fast_preset(hp, tmp_path):
    """preset fast (BASELINE configs[4]'s preset, vvencCfg.cpp:2751-2819: affine, BDOF, MMVD, two references per list, DepQuant, LFNST on top of faster): every call the
    encoder makes through the tables while those tools search — affine's gradient-search distortions, MMVD / merge pruning, the second reference of a list — lands in the
    lists (NOTHING dropped), and every recorded output replays bit-exactly; TU lists against the reference's own entries"""
    pics = _record(tmp_path, 416, 240, 9, pocs=(4, 8), preset="fast")
    assert len(pics) == 2
    refs = set()
    for pic in pics.values():
        refs |= set(pic.me["refPlane"].tolist())
    assert len(refs) >= 2, refs
    tot = _replay_and_check(hp, pics)
    assert tot["integer_candidates"] > 20000 and tot["subpel_positions"] > 2000 and tot["table_calls"] > 10000 and tot["tus"] > 2000, tot


def test_recorded_lists_fast_preset_4k_picture(hp, tmp_path):
    """one 3840x2160 picture of a preset-fast encode (the 4K stand-in for BASELINE configs[4]'s 8K geometry: the same tools, a recording that fits the test budget):
    nothing dropped, bit-exact"""
    pics = _record(tmp_path, 3840, 2160, 9, pocs=(4,), preset="fast")
    assert len(pics) == 1
    tot = _replay_and_check(hp, pics)
    assert tot["integer_candidates"] > 100000 and tot["subpel_positions"] > 20000 and tot["tus"] > 10000, tot


def test_recorded_lists_1080p_layers(hp, tmp_path):
    """BASELINE configs[1] geometry: the six layer pictures bench.py replays, full 65-frame encode"""
    sys.path.insert(0, ROOT)
    import bench
    pics = _record(tmp_path, 1920, 1080, 65, pocs=sorted(bench.LAYER_POCS.values()))
    tot = _replay_and_check(hp, pics)
    pairs = np.mean([p.sample_pairs / (1.5 * 1920 * 1080) for p in pics.values() if p.slice_type != 2])
    assert 40 < pairs < 70, pairs                                    # ~54 x 1.5 W H sample pairs per B picture


def test_recorded_lists_4k_layers(hp, tmp_path):
    """BASELINE configs[2] geometry: the six layer pictures of the 3840x2160 x 65 preset-faster encode that bench.py's 4K pass replays (`value_4k`): bit-exact against the
    encoder's recorded costs and the reference's TU entries"""
    sys.path.insert(0, ROOT)
    import bench
    pics = _record(tmp_path, 3840, 2160, 65, pocs=sorted(bench.LAYER_POCS.values()))
    assert len(pics) == 6
    tot = _replay_and_check(hp, pics)
    assert tot["integer_candidates"] > 500000 and tot["subpel_positions"] > 100000 and tot["table_calls"] > 1000000 and tot["tus"] > 200000, tot


# ---------------------------------------------------------------------------------------------------------------- corners named by VERDICT r2
WAVEFRONT = r'''
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from hip_backend import HipBackend
from oracle.oracle import Oracle
from test_gpu_parity import synth_pair
hip, orc = HipBackend(), Oracle()
bad = 0
for (w, h, speed, unit, add) in [(416, 240, 4, 16, False), (200, 136, 2, 8, False)]:
    org, ref = synth_pair(np.random.default_rng(106 + w), h, w, shift=(5, 2))
    exp = orc.mctf_me(org, ref, 10, unit, speed, add)[4]
    got = hip.mctf_me(org, ref, 10, unit, speed, add)[4]
    bad += sum(int((got[f] != exp[f]).sum()) for f in ("x", "y", "error", "rmsme", "overlap"))
print(json.dumps({"bad": bad}))
'''


def test_mctf_row_wavefront_fallback_vs_oracle():
    """VVHIP_MCTF_DIAG=0 forces meWavefrontKernel (the fallback of the diagonal sweep for pictures whose diagonals exceed one workgroup): same fields as the oracle"""
    env = dict(os.environ, VVHIP_MCTF_DIAG="0")
    r = subprocess.run([sys.executable, "-c", WAVEFRONT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["bad"] == 0


def test_8k_planes_tiled_and_shifted_lists(hp):
    """7680x4320 planes (BASELINE configs[4] geometry): the 8x8 SAD / SSE lists on the tiled copies and the Hadamard lists on the shifted copy equal the plain lists at the far
    corner of the picture (coordinates beyond 2^12 in the tiled address arithmetic)"""
    import torch
    rng = np.random.default_rng(8000)
    W, H, M = 7680, 4320, 80
    org = hp.plane(rng.integers(0, 1024, size=(H, W), dtype=np.int16), 0, extend=False)
    ref = hp.plane(rng.integers(0, 1024, size=(H, W), dtype=np.int16), M)
    org_t, ref_t, ref_s = hp.tile_plane(org), hp.tile_plane(ref), hp.shift_plane(ref)
    n = 4000
    bx = np.concatenate([rng.integers(0, W // 8, n // 2), rng.integers(W // 8 - 40, W // 8, n // 2)]) * 8
    by = np.concatenate([rng.integers(0, H // 8, n // 2), rng.integers(H // 8 - 40, H // 8, n // 2)]) * 8
    dx, dy = rng.integers(-24, 25, n), rng.integers(-24, 25, n)
    items = np.stack([by * org.stride + bx, (by + dy) * ref.stride + bx + dx], 1).astype(np.int32)
    d_items = hp.to_device(items)
    for func in ("SAD", "SSE", "HAD", "HAD_fast"):
        plain = hp.dist_batch(func, org, ref, d_items, n, 8, 8, 0, 10).cpu().numpy()
        out = torch.zeros(n, dtype=torch.int64, device=hp.device)
        jobs = hp.make_dist_fjobs([(func, 8, 8, 0, n, d_items, out)], flags=hp.DIST_FLAG_SAMPLES)
        hp.dist_multi_func_tiled(org, ref, org_t, ref_t, jobs, 10, cur_shift=ref_s)
        assert np.array_equal(out.cpu().numpy(), plain), func
