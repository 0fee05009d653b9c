#!/usr/bin/env python3
"""experiment (optimistic upper bound of plans over P pictures): every list of a recorded picture replicated P times inside ONE plan / ONE TU call on the SAME planes (so the
copies share caches — a real merge of P different pictures can only be slower), five streams; ms per picture-equivalent against P = 1"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from bench_common import LAYER_POCS, GOP_WEIGHT, prepare_recordings
from vvenc_amd.hotpath import HotPath
from vvenc_amd.replay import RecordedWorkload

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
hp = HotPath("cuda:0")
pics, _ = prepare_recordings(w, h, 65, sorted(LAYER_POCS.values()))
lanes = [hp.fork(torch.cuda.Stream()) for _ in range(5)]
res = {}
for P in (1, 2, 4):
    per_layer = {}
    for layer, poc in LAYER_POCS.items():
        wl = RecordedWorkload(hp, pics[poc], unique_bytes=False)
        if P > 1:
            ij = np.concatenate([wl.int_jobs] * P)
            if wl.int_jobs.size:
                for k in range(P):
                    ij["first_cand"][k * wl.int_jobs.size:(k + 1) * wl.int_jobs.size] += k * wl.plan_cands.size
            cands = np.concatenate([wl.plan_cands] * P); sj = np.concatenate([wl.stage_jobs] * P); it = np.concatenate([wl.items] * P); mi = np.concatenate([wl.mask_items] * P)
            wl.plan = hp.me_plan_create(ij, cands, sj, it, wl.bit_depth, 16 if False else hp.me_plan_info(wl.plan)["lds_bytes"] * 0 + 0, mask_items=mi) if False else hp.me_plan_create(ij, cands, sj, it, wl.bit_depth, 0, mask_items=mi)
            dev = hp.device
            wl.cand_cost = torch.zeros(max(1, cands.size), dtype=torch.int64, device=dev)
            wl.stage_cost = torch.zeros(max(1, 9 * sj.size), dtype=torch.int64, device=dev)
            wl.item_cost = torch.zeros(max(1, it.size + mi.size), dtype=torch.int64, device=dev)
            if wl.tu_table:
                jobs = []
                for g in wl.tu_groups:
                    jobs.append((g["w"], g["h"], g["tr_hor"], g["tr_ver"], g["n"], 8, g["d_off"], g["d_qp"], g["level"], g["rec"], g["stats"]))
                wl.tu_table = hp.make_tu_jobs(jobs * P)
                wl.tu_strides = (C.c_int32 * (len(jobs) * P))(*([g["w"] for g in wl.tu_groups] * P))
        calls = list(wl.bind_lanes(lanes))
        if P > 1:      # DMVR: P times the launches
            calls = calls + [c for c in calls[-len(wl.dmvr_groups):]] * (P - 1) if wl.dmvr_groups else calls
        for _ in range(3):
            for c in calls: c()
        torch.cuda.synchronize()
        n = 12 if layer else 4
        t = time.perf_counter()
        for _ in range(n):
            for c in calls: c()
        torch.cuda.synchronize()
        per_layer[layer] = 1000.0 * (time.perf_counter() - t) / n / P
    ms = sum(GOP_WEIGHT[l] * per_layer[l] for l in per_layer) / 32.0
    print("P = %d: GOP-weighted %.4f ms per picture-equivalent = %.0f pictures/s; by layer %s" % (P, ms, 1000.0 / ms, {l: round(v, 4) for l, v in per_layer.items()}), flush=True)
