#!/usr/bin/env python3
"""times the DMVR refinement search: every 16x16 sub-block of a 1920x1080 / 3840x2160 picture pair"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vvenc_amd.hotpath import HotPath, DMVR_ITEM_DTYPE
from vvenc_amd.workload import synth_frame_pair
hp = HotPath()
for (W, H) in ((1920, 1080), (3840, 2160)):
    cur, ref = synth_frame_pair(W, H, W)
    p0, p1 = hp.plane(cur, 32), hp.plane(ref, 32)
    bx, by = np.meshgrid(np.arange(0, W - 15, 16), np.arange(0, H - 15, 16))
    n = bx.size
    rng = np.random.default_rng(3)
    it = np.zeros(n, DMVR_ITEM_DTYPE)
    it["ref0_off"] = by.ravel() * p0.stride + bx.ravel()
    it["ref1_off"] = (by.ravel() + 1) * p1.stride + bx.ravel() + 3
    for f in ("frac0_x", "frac0_y", "frac1_x", "frac1_y"):
        it[f] = rng.integers(0, 16, n)
    d_it = hp.to_device(it)
    out = hp.dmvr_refine_batch(p0, p1, d_it, n, 16, 16, 10)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): hp.dmvr_refine_batch(p0, p1, d_it, n, 16, 16, 10, out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    alg = n * (2 * 21 * 21 * 2 + 16)          # two (16+5)^2 reference windows in, 16 B result
    print("DMVR %dx%d: %d sub-blocks 16x16, %.1f us, %.2f Msub-blocks/s, %.1f GB/s algorithmic (25 SAD positions + 2 bilinear predictions each)" % (W, H, n, us, n / us, alg / us / 1e3))
