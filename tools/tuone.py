#!/usr/bin/env python3
"""the merged fused-TU launch of the 1080p work lists, repeated: python tools/tuone.py [8,16,32] [--reps N]   (for tools/pmc_probe.py / tools/ktrace.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload

hp = HotPath(); wl = FrameWorkload(hp, 1920, 1080)
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else [8, 16, 32]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
tab = hp.make_tu_jobs([(S, S, 0, 0, n, 8, d_off, d_qp, lvl, rec, st) for (S, n, d_off, d_qp, lvl, rec, st, _, _) in wl.tu_jobs if S in sizes])
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): hp.tu_rdo_multi(wl.resi, tab, wl.bit_depth)
a.record()
for _ in range(reps): hp.tu_rdo_multi(wl.resi, tab, wl.bit_depth)
b.record(); torch.cuda.synchronize()
print("TU %s: %.2f us per launch" % (sizes, a.elapsed_time(b) / reps * 1e3))
