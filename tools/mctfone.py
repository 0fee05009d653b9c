#!/usr/bin/env python3
"""the MCTF job of one filtered picture (tools/bench_mctf.py: POC 32 against 4 references, search + filter of Y, U, V), N times on one stream — the run tools/ktrace.py and
tools/pmc_probe.py wrap: python tools/mctfone.py [width height [poc [n]]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench_mctf as BM  # noqa: E402
from vvenc_amd.hotpath import HotPath  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
poc = int(sys.argv[3]) if len(sys.argv) > 3 else 32
n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
hp = HotPath("cuda:0")
mc = BM.MctfCadence(hp, w, h)
job = [j for j in BM.JOBS if j[1] == poc][0]
for _ in range(n):
    mc.issue(job)
torch.cuda.synchronize()
