#!/usr/bin/env python3
"""times the MCTF hierarchical motion estimation (vvhip_mctf_motion_estimation) on 1080p / 4K synthetic pictures"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vvenc_amd.hotpath import HotPath
from vvenc_amd.workload import synth_frame_pair
hp = HotPath()
ME_ONLY = "--me1080" in sys.argv          # only the 1080p 4-reference search (clean kernel traces)
for (W, H) in ((1920, 1080),) if ME_ONLY else ((1920, 1080), (3840, 2160)):
    cur, ref = synth_frame_pair(W, H, W)
    pc = hp.plane(cur, 128)
    refs = [hp.plane(np.roll(ref, (k, -2 * k), (0, 1)), 128) for k in range(4)]
    for nref in (4,) if ME_ONLY else (1, 4):
        out, dims = hp.mctf_motion_estimation(pc, refs[:nref], 10, 16, 4, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            hp.mctf_motion_estimation(pc, refs[:nref], 10, 16, 4, True, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("%dx%d refs=%d : %.2f ms per filtered picture (%.2f ms per reference), blocks %dx%d" % (W, H, nref, ms, ms / nref, dims[0], dims[1]))

if ME_ONLY:
    sys.exit(0)
# ---- apply side: bilateral temporal filter of the three 4:2:0 planes with 4 references
from vvenc_amd.hotpath import MV_DTYPE
for (W, H) in ((1920, 1080), (3840, 2160)):
    cur, ref = synth_frame_pair(W, H, W)
    def yuv(y):
        return (y, np.clip(y[::2, ::2] // 2 + 256, 0, 1023).astype(np.int16), np.clip(1023 - y[::2, ::2] // 3, 0, 1023).astype(np.int16))
    refs_np = [yuv(np.roll(ref, (k, -2 * k), (0, 1))) for k in range(4)]
    org_np = yuv(cur)
    pc = hp.plane(cur, 128)
    prl = [hp.plane(r[0], 128) for r in refs_np]
    outs, dims = hp.mctf_motion_estimation(pc, prl, 10, 16, 4, True)
    planes = []
    for c in range(3):
        cs = 1 if c else 0
        planes.append((hp.plane(org_np[c], 128 >> cs), [hp.plane(r[c], 128 >> cs) for r in refs_np], cs))
    strengths = [hp.REF_STRENGTHS[0][k] for k in (0, 0, 1, 1)]
    params = [hp.mctf_filter_params(32, 10, 0.95, c > 0) for c in range(3)]
    def run():
        for c, (po, prs, cs) in enumerate(planes):
            hp.mctf_apply_plane(po, prs, outs, dims[0], cs, strengths, params[c][1], params[c][0], 10, 16, True, 32)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    samples = W * H * 1.5
    print("%dx%d bilateral filter, 4 references, Y+U+V: %.3f ms per filtered picture (%.1f Gsamples/s out, reads %.1f GB/s of reference samples)" % (W, H, ms, samples / ms / 1e6, samples * 4 * 2 / ms / 1e6))
