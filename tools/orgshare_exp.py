#!/usr/bin/env python3
"""experiment: how much of the SAD / HAD list time is the per-candidate re-read of the original block?  Same lists, but every candidate's
org_off points at ONE block (org rows then hit the same cache lines for all lanes) vs the real lists."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload
hp = HotPath()
wl = FrameWorkload(hp, 1920, 1080)
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for func in ("SAD", "HAD_fast"):
    real, same = [], []
    for (f, S, ss, n, d_items, d_out, items) in wl.dist_jobs:
        if f != func: continue
        it2 = items.copy(); it2[:, 0] = items[0, 0]
        real.append((f, S, S, ss, n, d_items, d_out))
        same.append((f, S, S, ss, n, hp.to_device(it2), d_out))
    tr, ts = hp.make_dist_fjobs(real), hp.make_dist_fjobs(same)
    print("%-8s real lists %.1f us   single-org lists %.1f us" % (func, timeit(lambda: hp.dist_multi_func(wl.org, wl.ref, tr, 10)), timeit(lambda: hp.dist_multi_func(wl.org, wl.ref, ts, 10))))
