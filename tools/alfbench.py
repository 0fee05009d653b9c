#!/usr/bin/env python3
"""ALF statistics on a 1080p picture: device kernels vs the reference's own entries (oracle/_ref, AVX2 row, 1 thread).
Development / measurement aid (uses oracle/_ref): python tools/alfbench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vvenc_amd.hotpath import HotPath
from oracle.oracle import RefLib

hp = HotPath()
R = RefLib(1)
W, H = 1920, 1080
rng = np.random.default_rng(9)
yy, xx = np.mgrid[0:H, 0:W]
rec = np.clip(512 + 140 * np.sin(xx / 23.0) * np.cos(yy / 19.0) + 60 * np.sin((xx + 2 * yy) / 9.0) + rng.normal(0, 6, (H, W)) + 30 * (((xx // 48) + (yy // 32)) % 2), 0, 1023).astype(np.int16)
org = np.clip(rec.astype(np.int32) + rng.integers(-10, 11, (H, W)), 0, 1023).astype(np.int16)
prec, porg = hp.plane(rec, 8), hp.plane(org, 0)
c_rec, c_org = np.ascontiguousarray(rec[::2, ::2][:540 // 4 * 4]), np.ascontiguousarray(org[::2, ::2][:540 // 4 * 4])
pcr, pco = hp.plane(c_rec, 8), hp.plane(c_org, 0)

def gpu_us(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

d_cls = hp.alf_classify(prec)
st = hp.alf_stats_plane(porg, prec, 128, 7, d_cls)
sc = hp.alf_stats_plane(pco, pcr, 64, 5, None, 64, 62)
t_cls = gpu_us(lambda: hp.alf_classify(prec, out=d_cls))
t_l = gpu_us(lambda: hp.alf_stats_plane(porg, prec, 128, 7, d_cls, out=st))
t_c = gpu_us(lambda: hp.alf_stats_plane(pco, pcr, 64, 5, None, 64, 62, out=sc))
t0 = time.perf_counter(); cr = R.alf_classify(rec); c_cls = time.perf_counter() - t0
t0 = time.perf_counter(); sr = R.alf_stats_plane(org, rec, 128, 7, cr); c_l = time.perf_counter() - t0
t0 = time.perf_counter(); scr = R.alf_stats_plane(c_org, c_rec, 64, 5, None, 64, 62); c_c = time.perf_counter() - t0
ok = np.array_equal(d_cls.cpu().numpy(), cr) and np.array_equal(st.cpu().numpy().view(np.uint32), sr.view(np.uint32)) and np.array_equal(sc.cpu().numpy().view(np.uint32), scr.view(np.uint32))
print("classes used:", np.count_nonzero(np.bincount(cr[..., 0].ravel(), minlength=25)))
print("| stage (1080p) | MI355X (us) | reference AVX2, 1 thread (us, incl. wrapper copies) |")
print("|---|---|---|")
print("| classification, 129 600 blocks | %.1f | %.0f |" % (t_cls, c_cls * 1e6))
print("| luma statistics 7x7, 135 CTUs x 25 classes | %.1f | %.0f |" % (t_l, c_l * 1e6))
print("| chroma statistics 5x5 (one plane) | %.1f | %.0f |" % (t_c, c_c * 1e6))
print("results equal:", ok)
sys.exit(0 if ok else 1)
