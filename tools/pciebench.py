#!/usr/bin/env python3
"""PCIe-inclusive rate of the frame workload (DESIGN §4 note): per step the three planes (original, reference, residual) and the work-item tables come from pinned
host memory, the frame's three launches run, and every result (distortions, levels, reconstructed residuals, TU statistics) goes back to pinned host memory.
Not the headline value: bench.py times the hot path with its inputs resident in HBM.   usage: python tools/pciebench.py [--width 1920 --height 1080 --steps 30]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import FrameWorkload

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080); ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
hp = HotPath()
wl = FrameWorkload(hp, a.width, a.height, seed=1080)
ups = [wl.org.storage, wl.ref.storage, wl.resi.storage] + [j[4] for j in wl.dist_jobs] + [t for j in wl.tu_jobs for t in (j[2], j[3])]
downs = [j[5] for j in wl.dist_jobs] + [t for j in wl.tu_jobs for t in (j[4], j[5], j[6]) if t is not None]
h_up = [torch.empty(t.shape, dtype=t.dtype).pin_memory().copy_(t.cpu()) for t in ups]
h_dn = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in downs]
up_b = sum(t.numel() * t.element_size() for t in ups); dn_b = sum(t.numel() * t.element_size() for t in downs)

def step():
    for d, h in zip(ups, h_up): d.copy_(h, non_blocking=True)
    wl.run(None)
    for h, d in zip(h_dn, downs): h.copy_(d, non_blocking=True)

for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
print("%dx%d: %.3f ms per frame = %.0f frames/s with %.1f MB up and %.1f MB down per frame (pinned host memory, one stream; %.1f GB/s aggregate)" %
      (a.width, a.height, ms, 1e3 / ms, up_b / 1e6, dn_b / 1e6, (up_b + dn_b) / ms / 1e6))
