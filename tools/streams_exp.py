#!/usr/bin/env python3
"""experiment: a frame's launches spread over several HIP streams (independent work lists) vs one stream"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload

hp = HotPath()
wl = FrameWorkload(hp, 1920, 1080)
streams = [torch.cuda.Stream() for _ in range(6)]

def on(stream, fn):
    with torch.cuda.stream(stream):
        hp.use_torch_stream()
        fn()

def step_multi(nstreams):
    jobs = []
    for func in ("SAD", "HAD_fast", "SSE"):
        jobs.append(lambda f=func: hp.dist_multi(f, wl.org, wl.ref, wl.job_tables[f], 10))
    for (S, n, d_off, d_qp, lvl, rec, st, _, _) in wl.tu_jobs:
        jobs.append(lambda S=S, n=n, d_off=d_off, d_qp=d_qp, lvl=lvl, rec=rec, st=st: hp.tu_rdo(wl.resi, d_off, n, S, S, d_qp, 0, 0, 10, 8, lvl, rec, st))
    for i, j in enumerate(jobs):
        on(streams[i % nstreams], j)

for ns in (1, 2, 3, 6):
    for _ in range(5): step_multi(ns)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps): step_multi(ns)
    torch.cuda.synchronize()
    print("streams=%d: %.1f us per frame" % (ns, (time.perf_counter() - t0) / reps * 1e6))
