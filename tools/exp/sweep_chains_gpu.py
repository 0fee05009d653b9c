"""experiment (GPU side): phase-A vectors (dev library, VVHIP_MCTF_NO_SWEEP=1) and final vectors of the final MCTF level -> gpurun_out/sweep_fields_<w>.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import e2e_fps
from vvenc_amd.hotpath import HotPath
w, h = int(sys.argv[1]), int(sys.argv[2])
hp = HotPath("cuda:0")
y, u, v = e2e_fps.synth_clip(w, h, 65)
out = {}
for poc, refs in ((32, (30, 31, 33, 34)), (8, (7, 9))):
    cur = hp.plane(y[poc], 128); prs = [hp.plane(y[r], 128) for r in refs]
    f, dims = hp.mctf_motion_estimation(cur, prs, 10, 16, 4, w >= 1920)
    for k, r in enumerate(refs):
        out["%s_%d_%d" % (os.environ.get("TAG", "F"), poc, r)] = HotPath.mv_to_numpy(f[k], dims)
np.savez(os.path.join(ROOT, "gpurun_out", "sweep_fields_%s_%d.npz" % (os.environ.get("TAG", "F"), w)), **out)
print("saved", list(out))
