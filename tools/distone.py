#!/usr/bin/env python3
"""one merged distortion launch of the 1080p work lists, repeated: python tools/distone.py SAD8,SAD16,SSE64,HAD_fast8 [--row-major] [--reps N]   (for tools/pmc_probe.py / tools/ktrace.py)"""
import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload

hp = HotPath(); wl = FrameWorkload(hp, 1920, 1080)
names = sys.argv[1].split(",")
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
J = {"%s%d" % (f, S): (f, S, S, ss, n, it, out) for (f, S, ss, n, it, out, _) in wl.dist_jobs}
sel = [J[n] for n in names]
tab = hp.make_dist_fjobs(sel, flags=hp.DIST_FLAG_SAMPLES if names[0].startswith("HAD") else 0)
for _ in range(reps):
    if "--row-major" in sys.argv: hp.dist_multi_func(wl.org, wl.ref, tab, wl.bit_depth)
    else: hp.dist_multi_func_tiled(wl.org, wl.ref, wl.org_tiled, wl.ref_tiled, tab, wl.bit_depth, cur_shift=wl.ref_shift)
torch.cuda.synchronize()
