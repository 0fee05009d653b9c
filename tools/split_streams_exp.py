#!/usr/bin/env python3
"""experiment: the picture's distortion lists split into more launches on more HIP streams (3 launches / 3 streams is the product path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload
hp = HotPath(); wl = FrameWorkload(hp, 1920, 1080)
J = wl.dist_jobs
def tab(sel, flags=0):
    return hp.make_dist_fjobs([(f, S, S, ss, n, it, out) for (f, S, ss, n, it, out, _) in J if sel(f, S)], flags=flags)
splits = {
  "3 launches": [("d", wl.fjob_tables["SAD_SSE"]), ("d", wl.fjob_tables["HAD_fast"]), ("t", None)],
  "5 launches (8x8 lists apart)": [("d", tab(lambda f, S: f in ("SAD", "SSE") and S == 8)), ("d", tab(lambda f, S: f in ("SAD", "SSE") and S > 8)),
                                  ("d", tab(lambda f, S: f == "HAD_fast" and S == 8, hp.DIST_FLAG_SAMPLES)), ("d", tab(lambda f, S: f == "HAD_fast" and S > 8, hp.DIST_FLAG_SAMPLES)), ("t", None)],
}
for name, calls in splits.items():
    streams = [torch.cuda.Stream() for _ in calls]
    def run():
        for st, (k, t) in zip(streams, calls):
            with torch.cuda.stream(st):
                hp.use_torch_stream()
                if k == "d":
                    hp.dist_multi_func_tiled(wl.org, wl.ref, wl.org_tiled, wl.ref_tiled, t, 10)
                else:
                    hp.tu_rdo_multi(wl.resi, wl.tu_table, 10)
        hp.use_torch_stream()
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): run()
    torch.cuda.synchronize()
    print("%-32s %.1f us per picture" % (name, (time.perf_counter() - t0) / 100 * 1e6))
