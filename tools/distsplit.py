#!/usr/bin/env python3
"""where the time of a merged distortion launch goes: the launch with every single job alone, with all but one job, and complete (HIP events, many repetitions)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload

hp = HotPath()
W, H = (3840, 2160) if "--4k" in sys.argv else (1920, 1080)
wl = FrameWorkload(hp, W, H)
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200


def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for cls, funcs, flags in (("SAD_SSE", ("SAD", "SSE"), hp.DIST_FLAG_SAMPLES), ("HAD_fast", ("HAD_fast",), hp.DIST_FLAG_SAMPLES)):
    jobs = [(f, S, S, ss, n, it, out) for (f, S, ss, n, it, out, _) in wl.dist_jobs if f in funcs]
    def run(sel):
        tab = hp.make_dist_fjobs(sel, flags=flags)
        return timeit(lambda: hp.dist_multi_func_tiled(wl.org, wl.ref, wl.org_tiled, wl.ref_tiled, tab, wl.bit_depth, cur_shift=wl.ref_shift))
    print("%s all jobs: %.2f us" % (cls, run(jobs)))
    for j in jobs:
        print("   only %-8s %2d: %6.2f us     without it: %6.2f us" % (j[0], j[1], run([j]), run([k for k in jobs if k is not j])))
