import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from hip_backend import HipBackend
from oracle.oracle import Oracle
hip, orc = HipBackend(), Oracle()
rng = np.random.default_rng(7)
org = rng.integers(0, 1024, size=(96, 128), dtype=np.int16)
cur = rng.integers(0, 1024, size=(96, 128), dtype=np.int16)
for (w,h) in [(4,2),(4,4),(8,8),(16,16),(64,64),(2,2)]:
  for (ox,oy,cx,cy) in [(0,0,0,0),(1,0,0,0),(0,0,1,0),(2,0,2,1),(3,1,5,3),(8,8,8,8)]:
    for ss in (0,1):
        if h>>ss < 1: continue
        g = int(hip.dist_many("SAD", org, cur, [(ox,oy,cx,cy)], w, h, 10, ss)[0])
        e = orc.dist("SAD", (org,oy,ox),(cur,cy,cx), w,h,10,ss)
        print(w,h,ox,oy,cx,cy,ss,g,e,"OK" if g==e else "BAD")
