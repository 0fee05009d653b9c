#!/usr/bin/env python3
"""Per-part times of the motion-search plan of recorded pictures (1080p clip of bench.py, cached under /tmp): refinement stages, integer windows (large / small LDS class), table
calls — HIP events inside the plan run, medians of N runs.   python tools/me_parts.py [pocs, default 2,5,15] [--reps N] [--window W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from vvenc_amd.hotpath import HotPath
from vvenc_amd import replay

pocs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else [2, 5, 15]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 30
win = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 16
hp = HotPath()
pics, _ = bench.prepare_recordings(1920, 1080, 65, pocs)
for poc in pocs:
    wl = replay.RecordedWorkload(hp, pics[poc], max_window=win)
    hp.me_plan_set_timing(wl.plan, True)
    for _ in range(3):
        wl.run_me()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        wl.run_me()
        torch.cuda.synchronize()
        t.append(hp.me_plan_last_times(wl.plan))
    m = np.median(np.array(t), 0) * 1000
    print("poc %d: stage %.1f  int_big %.1f  int_small %.1f  item %.1f us   plan %s  jobs %d cands %d stages %d items %d" %
          (poc, m[0], m[1], m[2], m[3], wl.me_info, wl.int_jobs.size, wl.plan_cands.size, wl.stage_jobs.size, wl.items.size))
