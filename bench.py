#!/usr/bin/env python3
"""bench.py — frames/sec of the MI355X hot path (BASELINE.json metric) + roofline + CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

A step = one pass of the hot path over one synthetic 1920x1080 10-bit frame (vvenc_amd/workload.py: 12 distortion launches
= 11 SAD + 8 HAD_fast + 1 SSE candidates per 8/16/32/64 block, and 3 fused transform/quant/dequant/inverse launches over the
8/16/32 TU tilings; ~1.66e8 sample pairs + 6.2e6 coefficients, the per-frame totals measured on the reference).
All inputs are resident in HBM before the timed region.  With N GPUs every rank owns its own pictures (round-robin picture
sharding, no data-path collective): value = N*K frames / max-over-ranks time.

Extra JSON objects: "roofline" for the dominant kernel class (HIP events on the launch stream inside the timed region,
algorithmic bytes from SURVEY §8d), "cpu_baseline" (the reference's own AVX2 table entries from oracle/_ref, or the C port,
timed on the host cores over a bounded sample of the same work lists).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vvenc_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


class EventTimers:
    """HIP events on the launch stream (torch's current stream == the context's stream) bracketing each kernel CLASS once per
    step (its launches are issued back to back).  Events are pre-allocated: nothing is created inside the timed region."""

    def __init__(self, classes, steps, launches_per_class):
        self.launches_per_class = launches_per_class
        self.pool = {k: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] for k in classes}
        self.idx = {k: 0 for k in classes}

    def start(self, key):
        self.pool[key][self.idx[key]][0].record()

    def stop(self, key):
        self.pool[key][self.idx[key]][1].record()
        self.idx[key] += 1

    def summary(self):
        out = {}
        for k, lst in self.pool.items():
            ms = [a.elapsed_time(b) for a, b in lst[:self.idx[k]]]
            nl = len(ms) * self.launches_per_class[k]
            out[k] = {"launches": nl, "total_ms": float(sum(ms)), "avg_ms": float(sum(ms) / nl)}
        return out


class OnlyClass:
    """forwards start/stop for one kernel class, ignores the others"""

    def __init__(self, inner, cls):
        self.inner, self.cls = inner, cls

    def start(self, key):
        if key == self.cls:
            self.inner.start(key)

    def stop(self, key):
        if key == self.cls:
            self.inner.stop(key)


def cpu_baseline(wl, budget_s=12.0):
    """Times the CPU path on the host cores over the SAME work lists and scales to frames/sec.
    kind "reference": the reference's own x86-SIMD (AVX2) table entries — oracle/_ref/libvvenc_ref.so, compiled from /root/reference —
    driven by a C++ std::thread pool inside the library (one ctypes call, no Python in the timed loop), every kernel class and size;
    kind "port": oracle/liboracle.so (scalar C restatement, one core, distortion lists only) when the reference build is absent."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    org = np.ascontiguousarray(wl.org.storage.cpu().numpy())
    ref = np.ascontiguousarray(wl.ref.storage.cpu().numpy())
    resi = np.ascontiguousarray(wl.resi.storage.cpu().numpy())
    org_p = org.ctypes.data + 2 * wl.org.origin
    ref_p = ref.ctypes.data + 2 * wl.ref.origin
    if O.RefLib.available():
        R = O.RefLib(1)
        L = R.L

        class FrameJob(C.Structure):
            _fields_ = [("kind", C.c_int32), ("df", C.c_int32), ("size", C.c_int32), ("subShift", C.c_int32),
                        ("items", C.c_void_p), ("aux", C.c_void_p), ("n", C.c_int32), ("pad", C.c_int32)]
        keep, jobs = [], []
        for (func, S, ss, n, _, _, items) in wl.dist_jobs:
            it = np.ascontiguousarray(items)
            keep.append(it)
            jobs.append(FrameJob(0, R._df[func], S, ss, it.ctypes.data, None, n, 0))
        for (S, n, _, _, _, _, _, off, qps) in wl.tu_jobs:
            o = np.ascontiguousarray(off)
            qf = np.zeros((n, 2), np.int16)
            qf[:, 0] = qps
            qf[:, 1] = 2
            keep += [o, qf]
            jobs.append(FrameJob(1, 0, S, 0, o.ctypes.data, qf.ctypes.data, n, 0))
        arr = (FrameJob * len(jobs))(*jobs)
        L.vvref_run_jobs_mt.restype = C.c_double
        L.vvref_run_jobs_mt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]

        def run(threads, passes):
            return L.vvref_run_jobs_mt(org_p, wl.org.stride, ref_p, wl.ref.stride, resi.ctypes.data, wl.resi.stride, wl.bit_depth,
                                       arr, len(jobs), threads, passes)
        out = {}
        cand = sorted({1, min(cores, 8), min(cores, 32), max(1, cores // 2), cores})
        for threads in cand:                               # thread-count sweep: report the best the host can do
            dt1 = run(threads, 1)
            passes = int(max(1, min(2000, (budget_s / (2.0 * len(cand))) / max(dt1, 1e-4))))
            dt = run(threads, passes)
            out[threads] = (passes / dt, passes, dt)
        best = max(out, key=lambda t: out[t][0])
        fps, passes, dt = out[best]
        return {"value": fps, "unit": "frames/s", "cores": best, "kind": "reference", "host_cpus": cores,
                "sweep_fps": {str(t): round(out[t][0], 2) for t in cand},
                "sample": "%d full passes over one frame's work lists (every kernel class, all block sizes) on %d std::threads in %.1f s wall; "
                          "reference x86-SIMD (AVX2) table entries called back-to-back; best of a thread-count sweep" % (passes, best, dt)}
    orc = O.Oracle()
    L = orc.L
    L.orc_dist_batch.restype = None
    L.orc_dist_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    fidx = {"SSE": 0, "SAD": 1, "HAD": 2, "HAD_fast": 3}
    frac = 0.05
    t0 = time.perf_counter()
    for (func, S, ss, n, _, _, items) in wl.dist_jobs:
        m = max(1, int(n * frac))
        sl = np.ascontiguousarray(items[:m])
        outb = np.zeros(m, np.uint64)
        L.orc_dist_batch(fidx[func], org_p, wl.org.stride, ref_p, wl.ref.stride, S, S, ss, sl.ctypes.data, m, outb.ctypes.data)
    dt = time.perf_counter() - t0
    return {"value": frac / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%.0f%% of one frame's distortion work lists (transform/quant not included) through the scalar C oracle, 1 thread, %.1f s" % (100 * frac, dt)}


def pmc_traffic(cls, args):
    """HBM-side bytes per launch of the dominant class from the committed PMC passes (profiles/pmc_r01.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH doubled per the gfx950 correction).  Counters cannot be collected inside
    this run; null when the profile does not cover the requested workload."""
    try:
        if (args.width, args.height) != (1920, 1080):
            return None
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_r01.json")))
        return d["classes"][cls]["traffic_bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip per-kernel HIP events inside the timed region")
    ap.add_argument("--bcast-ref", action="store_true", help="also broadcast the reference picture from its owner every step (RCCL)")
    ap.add_argument("--graph", type=int, default=1, help="extra measurement: the frame's launches replayed from a HIP graph (0 = skip)")
    ap.add_argument("--overlap-streams", type=int, default=3, help="extra measurement: the frame's launches on this many HIP streams (0/1 = skip)")
    ap.add_argument("--with-subpel", action="store_true", help="also run the fractional-ME stage per step (16 interpolated HAD_fast candidates per block; SURVEY 8f rank 1)")
    ap.add_argument("--with-mctf", type=int, default=0, help="also run the MCTF hierarchical ME against this many references per step")
    args = ap.parse_args()

    rank, local_rank, world = sharding.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: vvenc_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    from vvenc_amd.hotpath import HotPath
    from vvenc_amd.workload import FrameWorkload
    hp = HotPath("cuda:%d" % local_rank)
    wl = FrameWorkload(hp, args.width, args.height, seed=1080 + rank)
    if args.with_subpel:
        wl.enable_subpel()
    mctf_refs = []
    if args.with_mctf:
        cur128 = hp.plane(wl.cur_np, 128)
        mctf_refs = [hp.plane(np.roll(wl.ref_np, (k, -k), (0, 1)), 128) for k in range(args.with_mctf)]

    def step(timers=None):
        if args.bcast_ref and world > 1:
            sharding.broadcast_picture(wl.ref.storage, src_rank=0)
        wl.run(timers)
        if mctf_refs:
            hp.mctf_motion_estimation(cur128, mctf_refs, wl.bit_depth, 16, 4, args.width >= 1920)

    ap_launches = wl.class_launches_merged if wl.merged else wl.class_launches
    # warm-up: every kernel class is bracketed by events (per-class breakdown + choice of the dominant class) ...
    # (the first warm-up step carries the cold-launch costs and is left out of the per-class figures when there is more than one)
    wtimers = None if args.no_kernel_timers else EventTimers(list(ap_launches), max(args.warmup, 1), ap_launches)
    nwarm = max(args.warmup, 1) if wtimers is not None else args.warmup
    for i in range(nwarm):
        step(wtimers if (i > 0 or nwarm == 1) else None)
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    # ... timed region: only the dominant class keeps its event pair (2 event records per step instead of 8: events are not free)
    timers = None
    if wtimers is not None:
        wsum = wtimers.summary()
        dom_cls = max(wsum, key=lambda k: wsum[k]["total_ms"])
        timers = OnlyClass(EventTimers([dom_cls], args.steps, ap_launches), dom_cls)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timers)
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt, device="cuda")

    # extra (not `value`): the same steps with the three independent launches of a frame on separate HIP streams
    overlap = None
    if args.overlap_streams > 1 and wl.merged:
        streams = [torch.cuda.Stream() for _ in range(args.overlap_streams)]
        for _ in range(max(args.warmup, 1)):
            wl.run_overlapped(streams)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.run_overlapped(streams)
        torch.cuda.synchronize()
        dto = sharding.max_over_ranks(time.perf_counter() - t1, device="cuda")
        overlap = {"streams": args.overlap_streams, "value": args.steps * world / dto, "unit": "frames/s", "ms_per_step": 1000.0 * dto / args.steps,
                   "note": "same work, the 3 launches of a frame issued on separate HIP streams (no per-kernel events); not the headline value"}
    graph = None
    if args.graph and wl.merged and not mctf_refs and not args.with_subpel:
        # the frame's launches recorded once as a HIP graph and replayed with one hipGraphLaunch per step (the call sequence of a picture is fixed).
        # An extra measurement must never take the headline line down: failures are caught per rank and every rank runs the same collectives.
        gh, err, dtl = None, None, 0.0
        try:
            gh = hp.graph_capture(lambda: wl.run(None))
            for _ in range(max(args.warmup, 1)):
                hp.graph_launch(gh)
            torch.cuda.synchronize()
        except Exception as e:
            err = str(e)[:200]
        sharding.barrier()
        if err is None:
            try:
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    hp.graph_launch(gh)
                torch.cuda.synchronize()
                dtl = time.perf_counter() - t1
                hp.graph_destroy(gh)
            except Exception as e:
                err = str(e)[:200]
        dtg = sharding.max_over_ranks(dtl, device="cuda")
        bad = sharding.max_over_ranks(0.0 if err is None else 1.0, device="cuda")
        if bad > 0.0:
            graph = {"error": err or "failed on another rank"}
        else:
            graph = {"value": args.steps * world / dtg, "unit": "frames/s", "ms_per_step": 1000.0 * dtg / args.steps,
                     "note": "same work, the 3 launches of a frame captured once as a HIP graph and replayed (no per-kernel events); not the headline value"}
    if rank != 0:
        return
    frames = args.steps * world
    fps = frames / dt
    out = {
        "metric": "frames/sec + bit-exact vs CPU, 1080p/4K 10-bit preset=faster at 1/2/4/8 GPU",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i16", "data": "synthetic",
        "config": {"workload": "%dx%d 10-bit synthetic frame, preset=faster hot-path work lists: SAD/SATD(HAD_fast)/SSE candidate batches "
                               "(8..64 blocks, 20 candidates/block) + fused DCT-2/quant/dequant/IDCT TU batches (8..32)" % (args.width, args.height),
                   "sample_pairs_per_frame": int(wl.pairs), "coefficients_per_frame": int(wl.coefs), "launches_per_frame": 3 if wl.merged else len(wl.dist_jobs) + len(wl.tu_jobs),
                   "sharding": "pictures round-robin over ranks, no data-path collective" + (", reference-picture RCCL broadcast per step" if args.bcast_ref else ""),
                   "mctf_refs_per_step": args.with_mctf, "subpel_candidates_per_block": 16 if args.with_subpel else 0},
    }
    if timers is not None:
        ks = wsum                                             # all classes, measured over the warm-up steps
        nw = max(nwarm - 1, 1)
        for k in ks:
            ks[k]["alg_bytes_per_frame"] = int(wl.alg_bytes[k])
            ks[k]["alg_GBps"] = wl.alg_bytes[k] * nw / (ks[k]["total_ms"] * 1e-3) / 1e9
            ks[k]["measured_over"] = "%d warm-up steps" % nw
        dom = dom_cls
        td = timers.inner.summary()[dom]                      # the dominant class, measured over the timed region
        td["alg_bytes_per_frame"] = int(wl.alg_bytes[dom])
        td["alg_GBps"] = wl.alg_bytes[dom] * args.steps / (td["total_ms"] * 1e-3) / 1e9
        td["measured_over"] = "%d timed steps" % args.steps
        ks[dom] = td
        launches_per_frame = ks[dom]["launches"] / args.steps
        out["kernels"] = ks
        out["roofline"] = {"bound": "hbm", "kernel": {"SAD_SSE": "sadSseMixedKernel", "SAD": "sadSseMultiKernel<SAD>", "SSE": "sadSseMultiKernel<SSE>", "HAD_fast": "hadTile8PkMultiKernel", "TU": "tuMxMultiKernel<false>", "SUBPEL": "refinePredKernel + hadTileKernel", "TU8": "tuMxMultiKernel<false>", "TU16": "tuMxMultiKernel<false>", "TU32": "tuMxMultiKernel<false>"}[dom],
                           "achieved": ks[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ks[dom]["alg_GBps"] / HBM_PEAK_GBS,
                           "traffic": pmc_traffic(dom, args),
                           "alg_bytes_per_launch": wl.alg_bytes[dom] / launches_per_frame, "avg_launch_ms": ks[dom]["avg_ms"],
                           "note": "algorithmic bytes = 4*w*h per candidate (+8 B result), rows halved under subShift; fused TU = 6*w*h + 24 B (SURVEY 8d)"}
    if overlap is not None:
        out["overlap"] = overlap
    if graph is not None:
        out["graph"] = graph
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(wl)
        except Exception as e:   # the baseline is a report, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
