#!/usr/bin/env python3
"""Side-by-side timings for the SURVEY §8f rows on the GPU box: the device entry point vs the reference's own CPU entries (oracle/_ref,
AVX2 row) on the same work lists — and the results compared while at it.  Test infrastructure (uses oracle/_ref); prints a markdown table.

  python tests/perf_side_by_side.py [--threads 8]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import MV_DTYPE, RefLib  # noqa: E402
from vvenc_amd.hotpath import DMVR_ITEM_DTYPE, DMVR_RESULT_DTYPE, SUBPEL_DTYPE, HotPath  # noqa: E402
from vvenc_amd.workload import synth_frame_pair  # noqa: E402


def gpu_us(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    hp = HotPath()
    R = RefLib(1)
    L = R.L
    W, H = 1920, 1080
    cur, ref = synth_frame_pair(W, H, W)
    yy_, xx_ = np.mgrid[0:H, 0:W]
    rows = []

    # ---- f1: 16 sub-pel HAD candidates per 16x16 block
    S = 16
    margin = 80
    curp = np.ascontiguousarray(np.pad(cur, margin, mode="edge")); refp = np.ascontiguousarray(np.pad(ref, margin, mode="edge"))
    po, pr = hp.plane(cur, margin), hp.plane(ref, margin)
    assert po.stride == curp.shape[1]
    bx, by = np.meshgrid(np.arange(0, W - S + 1, S), np.arange(0, H - S + 1, S)); bx, by = bx.ravel(), by.ravel()
    nb = bx.size
    rng = np.random.default_rng(5)
    offs = [(-2, 0), (2, 0), (0, -2), (0, 2), (-2, -2), (2, -2), (-2, 2), (2, 2), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)]
    mvx, mvy = rng.integers(-12, 13, nb) * 4 + 12, rng.integers(-12, 13, nb) * 4 + 4
    it = np.zeros(nb * len(offs), SUBPEL_DTYPE)
    for k, (dx, dy) in enumerate(offs):
        qx, qy = mvx + dx, mvy + dy
        sl = slice(k * nb, (k + 1) * nb)
        it["org_off"][sl] = by * po.stride + bx
        it["ref_off"][sl] = (by + (qy >> 2)) * pr.stride + bx + (qx >> 2)
        it["frac_x"][sl] = (qx & 3) << 2
        it["frac_y"][sl] = (qy & 3) << 2
    n = it.size
    d_it = hp.to_device(it)
    out = torch.empty(n, dtype=torch.int64, device=hp.device)
    us = gpu_us(lambda: hp.subpel_dist_batch("HAD_fast", po, pr, d_it, n, S, S, 10, 0, False, out=out))
    cpu_out = np.zeros(n, np.uint64)
    L.vvref_subpel_batch_mt.restype = C.c_double
    org0 = curp.ctypes.data + 2 * (margin * curp.shape[1] + margin); ref0 = refp.ctypes.data + 2 * (margin * refp.shape[1] + margin)
    def cpu_subpel(threads):
        return L.vvref_subpel_batch_mt(C.c_void_p(org0), curp.shape[1], C.c_void_p(ref0), refp.shape[1], it.ctypes.data_as(C.c_void_p), n, S, S, 10,
                                       R._df["HAD_fast"], threads, cpu_out.ctypes.data_as(C.c_void_p))
    t1 = cpu_subpel(1); tn = cpu_subpel(args.threads)
    same = np.array_equal(out.cpu().numpy().view(np.uint64), cpu_out)
    rows.append(("f1 sub-pel candidates (interpolate + HAD_fast), 1080p, 16x16 blocks x 16 positions = %d candidates" % n, us, t1 * 1e6, tn * 1e6, same))

    # ---- f3: DMVR refinement of every 16x16 sub-block
    ditems = np.zeros(nb, DMVR_ITEM_DTYPE)
    ditems["ref0_off"] = by * po.stride + bx
    ditems["ref1_off"] = (by + 1) * pr.stride + bx + 3
    for f in ("frac0_x", "frac0_y", "frac1_x", "frac1_y"):
        ditems[f] = rng.integers(0, 16, nb)
    d_d = hp.to_device(ditems)
    dres = hp.dmvr_refine_batch(po, pr, d_d, nb, 16, 16, 10)
    us = gpu_us(lambda: hp.dmvr_refine_batch(po, pr, d_d, nb, 16, 16, 10, out=dres))
    cres = np.zeros(nb, DMVR_RESULT_DTYPE)
    L.vvref_dmvr_batch_mt.restype = C.c_double
    def cpu_dmvr(threads):
        return L.vvref_dmvr_batch_mt(C.c_void_p(org0), curp.shape[1], C.c_void_p(ref0), refp.shape[1], ditems.ctypes.data_as(C.c_void_p), nb, 16, 16, 10, threads,
                                     cres.ctypes.data_as(C.c_void_p))
    t1 = cpu_dmvr(1); tn = cpu_dmvr(args.threads)
    g = dres.cpu().numpy().reshape(-1).view(DMVR_RESULT_DTYPE)
    same = np.array_equal(g["mvd_x"], cres["mvd_x"]) and np.array_equal(g["mvd_y"], cres["mvd_y"]) and np.array_equal(g["min_cost"], cres["min_cost"])
    rows.append(("f3 DMVR refinement search, 1080p, %d sub-blocks 16x16" % nb, us, t1 * 1e6, tn * 1e6, same))

    # ---- MCTF: hierarchical ME (row a18) and f2 bilateral filter, 1080p, 4 references
    def yuv(y):
        return (y, np.clip(y[::2, ::2] // 2 + 256, 0, 1023).astype(np.int16), np.clip(1023 - y[::2, ::2] // 3, 0, 1023).astype(np.int16))
    refs_np = [yuv(np.roll(ref, (k, -2 * k), (0, 1))) for k in range(4)]
    org_np = yuv(cur)
    pc = hp.plane(cur, 128)
    prl = [hp.plane(r[0], 128) for r in refs_np]
    outs, dims = hp.mctf_motion_estimation(pc, prl, 10, 16, 4, True)
    us_me = gpu_us(lambda: hp.mctf_motion_estimation(pc, prl, 10, 16, 4, True, out=outs), 3)
    t0 = time.perf_counter(); cpu_mvs = [R.mctf_me(cur, r[0], 10, 16, 4, True)[4].ravel() for r in refs_np]; t_me = time.perf_counter() - t0
    gmv = [o.cpu().numpy().reshape(-1).view(MV_DTYPE) for o in outs]
    same = all(np.array_equal(a, b) for a, b in zip(gmv, cpu_mvs))
    rows.append(("a18 MCTF hierarchical ME, 1080p, 4 references (CPU: reference single-threaded row walk, incl. pyramid)", us_me, t_me * 1e6, float("nan"), same))
    planes = [(hp.plane(org_np[c], 128 >> (1 if c else 0)), [hp.plane(r[c], 128 >> (1 if c else 0)) for r in refs_np], 1 if c else 0) for c in range(3)]
    strengths = [hp.REF_STRENGTHS[0][k] for k in (0, 0, 1, 1)]
    params = [hp.mctf_filter_params(32, 10, 0.95, c > 0) for c in range(3)]
    fout = [None, None, None]
    def run_filter():
        for c, (po_, prs_, cs) in enumerate(planes):
            fout[c] = hp.mctf_apply_plane(po_, prs_, outs, dims[0], cs, strengths, params[c][1], params[c][0], 10, 16, True, 32, out=fout[c])
    us_f = gpu_us(run_filter, 5)
    t0 = time.perf_counter(); cpu_f = R.mctf_bilateral(org_np, refs_np, cpu_mvs, [0, 0, 1, 1], 10, 32, 16, True, True, 0.95); t_f = time.perf_counter() - t0
    diff = max(int(np.abs(fout[c].visible().cpu().numpy().astype(np.int32) - cpu_f[c]).max()) for c in range(3))
    rows.append(("f2 MCTF bilateral filter, 1080p Y+U+V, 4 references (CPU: reference AVX2 row, 1 thread, incl. plane set-up; max |diff| %d, allowed 1)" % diff,
                 us_f, t_f * 1e6, float("nan"), diff <= 1))

    # ---- f4: ALF encoder statistics (classification + luma 7x7 + two chroma 5x5 planes), 1080p
    a_rec = np.clip(512 + 140 * np.sin(xx_ / 23.0) * np.cos(yy_ / 19.0) + 60 * np.sin((xx_ + 2 * yy_) / 9.0) + rng.normal(0, 6, (H, W)) + 30 * (((xx_ // 48) + (yy_ // 32)) % 2), 0, 1023).astype(np.int16)
    a_org = np.clip(a_rec.astype(np.int32) + rng.integers(-10, 11, (H, W)), 0, 1023).astype(np.int16)
    c_rec, c_org = np.ascontiguousarray(a_rec[::2, ::2][:536]), np.ascontiguousarray(a_org[::2, ::2][:536])
    prec, porg, pcr, pco = hp.plane(a_rec, 8), hp.plane(a_org, 0), hp.plane(c_rec, 8), hp.plane(c_org, 0)
    d_cls = hp.alf_classify(prec)
    st = hp.alf_stats_plane(porg, prec, 128, 7, d_cls)
    sc = hp.alf_stats_plane(pco, pcr, 64, 5, None, 64, 62)
    def run_alf():
        hp.alf_classify(prec, out=d_cls)
        hp.alf_stats_plane(porg, prec, 128, 7, d_cls, out=st)
        hp.alf_stats_plane(pco, pcr, 64, 5, None, 64, 62, out=sc)
        hp.alf_stats_plane(pco, pcr, 64, 5, None, 64, 62, out=sc)
    us_alf = gpu_us(run_alf, 10)
    t0 = time.perf_counter()
    cr = R.alf_classify(a_rec); sr = R.alf_stats_plane(a_org, a_rec, 128, 7, cr)
    scr = R.alf_stats_plane(c_org, c_rec, 64, 5, None, 64, 62); R.alf_stats_plane(c_org, c_rec, 64, 5, None, 64, 62)
    t_alf = time.perf_counter() - t0
    same = np.array_equal(d_cls.cpu().numpy(), cr) and np.array_equal(st.cpu().numpy().view(np.uint32), sr.view(np.uint32)) and np.array_equal(sc.cpu().numpy().view(np.uint32), scr.view(np.uint32))
    rows.append(("f4 ALF statistics, 1080p: classification of 129 600 blocks + covariance records of 135 luma CTUs x 25 classes + 2 chroma planes (CPU: reference AVX2 row, "
                 "1 thread, incl. the wrapper's plane padding; floats compared bit for bit)", us_alf, t_alf * 1e6, float("nan"), same))

    print("| row / work list | MI355X (us) | reference AVX2, 1 thread (us) | reference AVX2, %d threads (us) | results equal |" % args.threads)
    print("|---|---|---|---|---|")
    for name, g_us, c1, cn, same in rows:
        print("| %s | %.1f | %.0f | %s | %s |" % (name, g_us, c1, "%.0f" % cn if cn == cn else "n/a", "yes" if same else "NO"))
    if not all(r[4] for r in rows):
        sys.exit(1)


if __name__ == "__main__":
    main()
