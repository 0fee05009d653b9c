#!/usr/bin/env python3
"""One fused-TU launch of a given mix of square TU lists on compact residual blocks (what the recorded replay issues): python tools/tu_mix.py 64:955,32:2133,16:600,8:800,4:600 [--reps N]
prints microseconds per call.  Per-wave latency of a size: a list short enough to be one resident round (e.g. 64:256)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vvenc_amd.hotpath import HotPath

hp = HotPath()
dev = hp.device
spec = [(int(a), int(b)) for a, b in (x.split(":") for x in sys.argv[1].split(","))]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
amp = int(sys.argv[sys.argv.index("--amp") + 1]) if "--amp" in sys.argv else 300          # --amp 1: residuals that quantise to all-zero levels (the shortcut of round 5)
rng = np.random.default_rng(7)
total = sum(n * w * w for w, n in spec)
pool = hp.to_device(rng.integers(-amp, amp + 1, total, dtype=np.int16))
jobs, strides, at, keep = [], [], 0, []
for w, n in spec:
    off = hp.to_device((at + np.arange(n, dtype=np.int32) * w * w).astype(np.int32))
    qf = np.zeros((n, 2), np.int16); qf[:, 0] = 32
    qp = hp.to_device(qf)
    lv = torch.empty(n * w * w, dtype=torch.int16, device=dev); rc = torch.empty_like(lv); st = torch.empty((n, 24), dtype=torch.uint8, device=dev)
    keep += [off, qp, lv, rc, st]
    jobs.append((w, w, 0, 0, n, 8, off, qp, lv, rc, st)); strides.append(w)
    at += n * w * w
tab = hp.make_tu_jobs(jobs)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
hp.use_torch_stream()
for _ in range(5):
    hp.tu_rdo_multi_strided(pool, strides, tab)
torch.cuda.synchronize()
a.record()
for _ in range(reps):
    hp.tu_rdo_multi_strided(pool, strides, tab)
b.record(); torch.cuda.synchronize()
print("TU mix %s amp %d: %.2f us per call" % (sys.argv[1], amp, a.elapsed_time(b) / reps * 1e3))
