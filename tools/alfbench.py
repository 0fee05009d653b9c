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
# ---- apply side: filterBlk 7x7 / 5x5 and CC-ALF over the whole picture (every CTU enabled, 3 filter sets)
nctu = 9 * 15
coeff = rng.integers(-40, 41, (3, 25, 13)).astype(np.int16); coeff[..., 12] = 0
clip = np.array([1024, 128, 32, 8], np.int16)[rng.integers(0, 4, (3, 25, 13))]
lin = np.full((3, 25, 13), 1024, np.int16)
ctu_set = rng.integers(0, 3, nctu).astype(np.int16)
d_coeff, d_clip, d_set = hp.to_device(coeff), hp.to_device(clip), hp.to_device(ctu_set)
pdst, pcdst = hp.plane(rec, 0), hp.plane(c_rec, 0)
c_coeff, c_clip = np.ascontiguousarray(coeff[:, :1]), np.ascontiguousarray(clip[:, :1])
d_ccoeff, d_cclip = hp.to_device(c_coeff), hp.to_device(c_clip)
cc_coeff = np.zeros((4, 8), np.int16); cc_coeff[:, :7] = np.array([0, 1, 2, 4, 8, 16, 32, 64], np.int16)[rng.integers(0, 8, (4, 7))] * rng.choice([-1, 1], (4, 7))
cc_ctl = rng.integers(1, 5, nctu).astype(np.uint8)
d_cc, d_ctl = hp.to_device(cc_coeff), hp.to_device(cc_ctl)
t_fl = gpu_us(lambda: hp.alf_filter_plane(prec, pdst, 128, 10, 7, d_coeff, None, d_set, d_cls))
g_lin = pdst.visible().cpu().numpy()
t_fn = gpu_us(lambda: hp.alf_filter_plane(prec, pdst, 128, 10, 7, d_coeff, d_clip, d_set, d_cls))
g_non = pdst.visible().cpu().numpy()
t_fc = gpu_us(lambda: hp.alf_filter_plane(pcr, pcdst, 64, 10, 5, d_ccoeff, d_cclip, d_set, None, 64, 62))
g_chr = pcdst.visible().cpu().numpy()
t_cc = gpu_us(lambda: hp.ccalf_filter_plane(pcdst, prec, 64, 10, d_cc, d_ctl))
pcdst = hp.plane(g_chr, 0); hp.ccalf_filter_plane(pcdst, prec, 64, 10, d_cc, d_ctl); g_cc = pcdst.visible().cpu().numpy()
t0 = time.perf_counter(); r_lin = R.alf_filter_plane(rec, 128, 10, 7, coeff, lin, ctu_set, cr); c_fl = time.perf_counter() - t0
t0 = time.perf_counter(); r_non = R.alf_filter_plane(rec, 128, 10, 7, coeff, clip, ctu_set, cr); c_fn = time.perf_counter() - t0
t0 = time.perf_counter(); r_chr = R.alf_filter_plane(c_rec, 64, 10, 5, c_coeff, c_clip, ctu_set, None, None, 64, 62); c_fc = time.perf_counter() - t0
t0 = time.perf_counter(); r_cc = R.ccalf_filter_plane(r_chr, rec[:2 * c_rec.shape[0]], 64, 10, cc_coeff, cc_ctl); c_cc = time.perf_counter() - t0
ok2 = np.array_equal(g_lin, r_lin) and np.array_equal(g_non, r_non) and np.array_equal(g_chr, r_chr) and np.array_equal(g_cc, r_cc)
print("| luma filtering 7x7, linear entry | %.1f | %.0f |" % (t_fl, c_fl * 1e6))
print("| luma filtering 7x7, clipping values | %.1f | %.0f |" % (t_fn, c_fn * 1e6))
print("| chroma filtering 5x5 (one plane), clipping values | %.1f | %.0f |" % (t_fc, c_fc * 1e6))
print("| CC-ALF filtering (one plane) | %.1f | %.0f |" % (t_cc, c_cc * 1e6))
print("results equal:", ok, ok2)
sys.exit(0 if ok and ok2 else 1)
