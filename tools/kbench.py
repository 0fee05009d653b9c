#!/usr/bin/env python3
"""per-kernel micro-benchmark on the 1080p work lists (HIP events, many repetitions)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload

hp = HotPath()
wl = FrameWorkload(hp, 1920, 1080)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50

def timeit(fn, reps=reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us

for (func, S, ss, n, d_items, d_out, _) in wl.dist_jobs:
    us = timeit(lambda: hp.dist_batch(func, wl.org, wl.ref, d_items, n, S, S, ss, 10, out=d_out))
    by = n * (4 * S * (S >> ss) + 8)
    print("%-9s S=%2d n=%7d  %8.2f us  alg %7.1f GB/s  %6.2f Gpairs/s" % (func, S, n, us, by / us / 1e3, n * S * S / us / 1e3))
for (S, n, d_off, d_qp, lvl, rec, st, _, _) in wl.tu_jobs:
    us = timeit(lambda: hp.tu_rdo(wl.resi, d_off, n, S, S, d_qp, 0, 0, 10, 8, lvl, rec, st))
    print("TU fused  S=%2d n=%7d  %8.2f us  alg %7.1f GB/s" % (S, n, us, n * (6 * S * S + 24) / us / 1e3))
    us = timeit(lambda: hp.fwd_transform(wl.resi, d_off, n, S, S))
    print("  fwd only            %8.2f us" % us)
us = timeit(lambda: hp.tu_rdo_multi(wl.resi, wl.tu_table, wl.bit_depth))
print("TU merged launch (8+16+32)   %8.2f us" % us)
for c in ("SAD_SSE", "HAD_fast"):
    us = timeit(lambda: hp.dist_multi_func(wl.org, wl.ref, wl.fjob_tables[c], wl.bit_depth))
    print("%-8s merged launch       %8.2f us" % (c, us))
# host overhead of a whole step
t0 = time.perf_counter()
for _ in range(200): wl.run()
torch.cuda.synchronize()
print("whole step (no events): %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(200): wl.run()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host-side enqueue time per step: %.1f us" % ((t1 - t0) / 200 * 1e6))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    hp.use_torch_stream()
    wl.run(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        hp.use_torch_stream()
        wl.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize()
print("whole step as hipGraph replay: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
