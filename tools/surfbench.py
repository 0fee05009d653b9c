#!/usr/bin/env python3
"""times vvhip_sad_surface (full-search cost surface) over every block of a 1080p frame"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload
hp = HotPath()
wl = FrameWorkload(hp, 1920, 1080)
def timeit(fn, reps=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (S, R) in ((8, 8), (16, 16), (32, 16), (64, 16), (16, 32)):
    bx, by = np.meshgrid(np.arange(0, 1920 - S + 1, S), np.arange(0, 1080 - S + 1, S))
    oo = hp.to_device((by.ravel() * wl.org.stride + bx.ravel()).astype(np.int32))
    ro = hp.to_device((by.ravel() * wl.ref.stride + bx.ravel()).astype(np.int32))
    nb = bx.size
    ss = 1 if S > 8 else 0
    out = torch.empty(nb * (2 * R + 1) ** 2, dtype=torch.int32, device=hp.device)
    us = timeit(lambda: hp.sad_surface(wl.org, wl.ref, oo, ro, nb, S, S, ss, R, R, out=out))
    cands = nb * (2 * R + 1) ** 2
    pairs = cands * S * (S >> ss)
    print("surface S=%2d R=%2d blocks=%6d cands=%9d : %9.1f us  %7.2f Tpairs/s  per-candidate-algorithmic %8.1f TB/s" % (S, R, nb, cands, us, pairs / us / 1e6, cands * 4 * S * (S >> ss) / us / 1e6))
