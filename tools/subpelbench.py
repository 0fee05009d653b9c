#!/usr/bin/env python3
"""times the fractional-ME stage: per S x S block 16 sub-pel candidates (8 half-pel + 8 quarter-pel positions around an
integer vector) -> interpolated prediction block -> HAD against the original; 1920x1080 10-bit synthetic frame"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vvenc_amd.hotpath import HotPath, SUBPEL_DTYPE
from vvenc_amd.workload import synth_frame_pair
hp = HotPath()
W, H = 1920, 1080
cur, ref = synth_frame_pair(W, H, W)
po, pr = hp.plane(cur, 0), hp.plane(ref, 80)
rng = np.random.default_rng(5)
offs = [(-2, 0), (2, 0), (0, -2), (0, 2), (-2, -2), (2, -2), (-2, 2), (2, 2), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)]   # quarter-sample units
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for S in (8, 16, 32, 64):
    bx, by = np.meshgrid(np.arange(0, W - S + 1, S), np.arange(0, H - S + 1, S))
    bx, by = bx.ravel(), by.ravel()
    nb = bx.size
    mvx, mvy = rng.integers(-12, 13, nb) * 4 + 12, rng.integers(-12, 13, nb) * 4 + 4        # integer vectors, quarter units
    it = np.zeros(nb * len(offs), SUBPEL_DTYPE)
    k = 0
    for (dx, dy) in offs:
        qx, qy = mvx + dx, mvy + dy
        it["org_off"][k:k + nb] = by * po.stride + bx
        it["ref_off"][k:k + nb] = (by + (qy >> 2)) * pr.stride + bx + (qx >> 2)
        it["frac_x"][k:k + nb] = (qx & 3) << 2
        it["frac_y"][k:k + nb] = (qy & 3) << 2
        k += nb
    n = it.size
    d_it = hp.to_device(it)
    out = torch.empty(n, dtype=torch.int64, device=hp.device)
    pred = torch.empty(n * S * S, dtype=torch.int16, device=hp.device)
    t_pred = timeit(lambda: hp.interp_luma_batch(pr, d_it, n, S, S, 10, True, 0, False, out=pred))
    t_all = timeit(lambda: hp.subpel_dist_batch("HAD_fast", po, pr, d_it, n, S, S, 10, 0, False, out=out))
    # the same 16 positions through the pattern-refinement entry (window + original staged once per block)
    bases = np.zeros(nb, SUBPEL_DTYPE)
    bases["org_off"] = by * po.stride + bx
    bases["ref_off"] = (by + (mvy >> 2)) * pr.stride + bx + (mvx >> 2)
    bases["frac_x"] = (mvx & 3) << 2
    bases["frac_y"] = (mvy & 3) << 2
    d_b = hp.to_device(bases)
    offs16 = [(4 * dx, 4 * dy) for (dx, dy) in offs]
    out2 = torch.empty(n, dtype=torch.int64, device=hp.device)
    t_ref = timeit(lambda: hp.subpel_refine_batch("HAD_fast", po, pr, d_b, nb, offs16, S, S, 10, 0, False, out=out2))
    ok = torch.equal(out2.view(nb, 16).t().contiguous().view(-1), out)
    alg = n * ((S + 7) * (S + 7) * 2 + S * S * 2 + 8)
    print("subpel S=%2d cands=%8d : interp %8.1f us, interp+HAD %8.1f us (%7.1f GB/s algorithmic) | refine entry %8.1f us (%7.1f GB/s algorithmic, %6.2f Gsamples/s), equal=%s" % (S, n, t_pred, t_all, alg / t_all / 1e3, t_ref, alg / t_ref / 1e3, n * S * S / t_ref / 1e3, ok))
