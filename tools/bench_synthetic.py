#!/usr/bin/env python3
"""tools/bench_synthetic.py — the rounds 1-2 bench on SYNTHETIC uniform work lists (vvenc_amd/workload.py), kept as a kernel-level benchmark of the list
kernels of dist.hip / trquant.hip; the driver-facing bench.py replays work lists recorded from the reference encoder and imports this file's MCTF / encoder helpers.

bench_synthetic.py — frames/sec of the MI355X hot path (BASELINE.json metric) with roofline, in-run parity, the MCTF stage, a 4K pass, the end-to-end encoder and the CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

`value` (the headline, BASELINE configs[1] "SAD/SATD + DCT batch"): a step = one pass of the hot path over one synthetic 1920x1080 10-bit picture
(vvenc_amd/workload.py: 11 SAD + 8 HAD_fast + 1 SSE candidates per 8/16/32/64 block and the fused xT -> quant -> dequant -> xIT -> SSE pipeline over the 8/16/32 TU
tilings = the per-frame totals measured on the reference), three launches, inputs resident in HBM.  With N GPUs every rank works on a DIFFERENT picture of one
sequence; the reference picture of step s+1 is published by its owner to every rank (RCCL broadcast of luma + chroma over xGMI, vvenc_amd/sharding.PictureExchange) inside
the timed region, overlapped with the kernels of step s: value = N*K pictures / max-over-ranks time, "scaling": "weak".

Extra objects of the same JSON line (rank 0; each can be switched off, each failure is reported in place and never costs the headline):
  roofline      dominant kernel class: algorithmic bytes per launch / HIP-event launch time (the per-candidate figure of SURVEY §8d, labelled nominal), the same time against the
                L2 ceiling (frac_l2), the unique bytes of the launch against HBM (frac_hbm_unique), and counter traffic per launch (FETCH_SIZE x2 + WRITE_SIZE, collected
                by this run's own rocprofv3 --pmc passes when rocprofv3 is there, else read from the committed profile)
  parity        every output of the timed launches (all distortion candidates, every TU's SSE / abs-sum) compared with the reference's x86-SIMD table entries on the same
                lists, and one MCTF motion field with the oracle: "bit-exact" or the mismatch count
  mctf          BASELINE configs[2] stage at 1080p: hierarchical ME against 4 references + bilateral filter of Y, U, V; ms per picture, critical-path bound
  pass_4k       the same three launches + the MCTF stage on a 3840x2160 picture (configs[2] geometry)
  kernel_trace  rocprofv3 --kernel-trace of a short inner run (per-kernel average durations, incl. the MCTF kernels)
  e2e           the real reference encoder, 1080p x 65 frames, preset faster: CPU kernels vs --SIMD=HIP (whole-picture stages on the device), fps + bitstream md5 equality
  cpu_baseline  the reference's own AVX2 table entries on the host cores over the same work lists (thread sweep, dynamic chunking), or the scalar C port
"""
import argparse
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vvenc_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0    # aggregate L2 bandwidth of the 8 XCDs, same guide (L2 section)
KERNEL_OF_CLASS = {"SAD_SSE": "sadSseMixedKernel", "HAD_fast": "hadTile8PkMultiKernel<false>", "TU": "tuMxMultiKernel<false>"}


class EventTimers:
    """HIP events on the launch stream (torch's current stream == the context's stream) bracketing each kernel CLASS once per
    step (its launches are issued back to back).  Events are pre-allocated: nothing is created inside the timed region."""

    def __init__(self, classes, steps, launches_per_class):
        self.launches_per_class = launches_per_class
        self.pool = {k: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] for k in classes}
        self.idx = {k: 0 for k in classes}

    def start(self, key, stream=None):
        if key in self.pool:
            self.pool[key][self.idx[key]][0].record(stream) if stream is not None else self.pool[key][self.idx[key]][0].record()

    def stop(self, key, stream=None):
        if key in self.pool:
            self.pool[key][self.idx[key]][1].record(stream) if stream is not None else self.pool[key][self.idx[key]][1].record()
            self.idx[key] += 1

    def summary(self):
        out = {}
        for k, lst in self.pool.items():
            ms = [a.elapsed_time(b) for a, b in lst[:self.idx[k]]]
            nl = max(1, len(ms) * self.launches_per_class[k])
            out[k] = {"launches": nl, "total_ms": float(sum(ms)), "avg_ms": float(sum(ms) / nl)}
        return out


def timed_ms(fn, reps, warm=1):
    """average wall-clock ms of fn() on the current stream, HIP events"""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


# ------------------------------------------------------------------------------------------------------------------------ CPU side
def host_cpu_info():
    info = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_" + os.path.basename(p)] = open(p).read().strip()
        except OSError:
            pass
    return info


class RefJobs:
    """the frame's work lists as job records of oracle/_ref's multi-threaded driver (vvref_run_jobs_mt): the reference's own x86-SIMD table entries"""

    class FrameJob(C.Structure):
        _fields_ = [("kind", C.c_int32), ("df", C.c_int32), ("size", C.c_int32), ("subShift", C.c_int32),
                    ("items", C.c_void_p), ("aux", C.c_void_p), ("n", C.c_int32), ("pad", C.c_int32), ("out", C.c_void_p)]

    def __init__(self, wl, with_outputs):
        from oracle import oracle as O
        self.R = O.RefLib(1)
        self.L = self.R.L
        self.wl = wl
        self.org = np.ascontiguousarray(wl.org.storage.cpu().numpy())
        self.ref = np.ascontiguousarray(wl.ref.storage.cpu().numpy())
        self.resi = np.ascontiguousarray(wl.resi.storage.cpu().numpy())
        self.keep, jobs, self.outs = [], [], []
        for (func, S, ss, n, _, _, items) in wl.dist_jobs:
            it = np.ascontiguousarray(items)
            o = np.zeros(n, np.uint64) if with_outputs else None
            self.keep.append(it)
            self.outs.append(o)
            jobs.append(self.FrameJob(0, self.R._df[func], S, ss, it.ctypes.data, None, n, 0, o.ctypes.data if o is not None else None))
        for (S, n, _, _, _, _, _, off, qps) in wl.tu_jobs:
            o_ = np.ascontiguousarray(off)
            qf = np.zeros((n, 2), np.int16)
            qf[:, 0] = qps
            qf[:, 1] = 2
            o = np.zeros(n, np.uint64) if with_outputs else None
            self.keep += [o_, qf]
            self.outs.append(o)
            jobs.append(self.FrameJob(1, 0, S, 0, o_.ctypes.data, qf.ctypes.data, n, 0, o.ctypes.data if o is not None else None))
        self.arr = (self.FrameJob * len(jobs))(*jobs)
        self.n = len(jobs)
        self.L.vvref_run_jobs_mt.restype = C.c_double
        self.L.vvref_run_jobs_mt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]

    def run(self, threads, passes):
        wl = self.wl
        return self.L.vvref_run_jobs_mt(self.org.ctypes.data + 2 * wl.org.origin, wl.org.stride, self.ref.ctypes.data + 2 * wl.ref.origin, wl.ref.stride,
                                        self.resi.ctypes.data, wl.resi.stride, wl.bit_depth, self.arr, self.n, threads, passes)


def cpu_baseline(wl, budget_s=12.0):
    """Times the CPU path on the host cores over the SAME work lists and scales to frames/sec.
    kind "reference": the reference's own x86-SIMD (AVX2) table entries — oracle/_ref/libvvenc_ref.so, compiled from /root/reference — driven by std::threads inside
    the library that pull 256-item chunks from one atomic counter (no Python, no allocation in the timed loop), every kernel class and size;
    kind "port": oracle/liboracle.so (scalar C restatement, one core, distortion lists only) when the reference build is absent."""
    from oracle import oracle as O
    info = host_cpu_info()
    cores = info["affinity"]
    if O.RefLib.available():
        J = RefJobs(wl, with_outputs=False)
        out = {}
        cand = sorted({1, min(cores, 8), min(cores, 16), min(cores, 32), min(cores, 64), cores})
        for threads in cand:                               # thread-count sweep: report the best the host can do
            dt1 = J.run(threads, 1)
            passes = int(max(1, min(2000, (budget_s / (2.0 * len(cand))) / max(dt1, 1e-4))))
            dt = J.run(threads, passes)
            out[threads] = (passes / dt, passes, dt)
        best = max(out, key=lambda t: out[t][0])
        fps, passes, dt = out[best]
        one = out[1][0]
        return {"value": fps, "unit": "frames/s", "cores": best, "kind": "reference", "host": info,
                "sweep_fps": {str(t): round(out[t][0], 2) for t in cand},
                "parallel_efficiency": {str(t): round(out[t][0] / (one * t), 3) for t in cand},
                "sample": "%d full passes over one frame's work lists (every kernel class, all block sizes) on %d std::threads in %.1f s wall; "
                          "reference x86-SIMD (AVX2) table entries called back-to-back, 256-item chunks from one atomic counter; best of a thread-count sweep" % (passes, best, dt),
                "scaling_note": "threads beyond what the sweep's efficiency column supports do not help: the lists are memory-side work on ~25 MB of planes and index lists per pass, "
                                "and the lease's usable cores are what `host` shows (affinity / cgroup quota), not os.cpu_count()"}
    orc = O.Oracle()
    L = orc.L
    L.orc_dist_batch.restype = None
    L.orc_dist_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    fidx = {"SSE": 0, "SAD": 1, "HAD": 2, "HAD_fast": 3}
    org = np.ascontiguousarray(wl.org.storage.cpu().numpy())
    ref = np.ascontiguousarray(wl.ref.storage.cpu().numpy())
    frac = 0.05
    t0 = time.perf_counter()
    for (func, S, ss, n, _, _, items) in wl.dist_jobs:
        m = max(1, int(n * frac))
        sl = np.ascontiguousarray(items[:m])
        outb = np.zeros(m, np.uint64)
        L.orc_dist_batch(fidx[func], org.ctypes.data + 2 * wl.org.origin, wl.org.stride, ref.ctypes.data + 2 * wl.ref.origin, wl.ref.stride, S, S, ss, sl.ctypes.data, m, outb.ctypes.data)
    dt = time.perf_counter() - t0
    return {"value": frac / dt, "unit": "frames/s", "cores": 1, "kind": "port", "host": info,
            "sample": "%.0f%% of one frame's distortion work lists (transform/quant not included) through the scalar C oracle, 1 thread, %.1f s" % (100 * frac, dt)}


def parity_check(hp, wl, mctf=None):
    """every result of the launches this run timed against the CPU reference on the same lists (bit-exact or counted)"""
    from oracle import oracle as O
    res = {"status": None, "checked": {}, "mismatches": 0}
    if O.RefLib.available():
        J = RefJobs(wl, with_outputs=True)
        J.run(min(16, len(os.sched_getaffinity(0))), 1)
        res["against"] = "the reference's x86-SIMD table entries (oracle/_ref, compiled from the reference) on the same work lists"
        k = 0
        nd = nt = 0
        for job in wl.dist_jobs:
            got = job[5].cpu().numpy().view(np.uint64)
            res["mismatches"] += int((got != J.outs[k]).sum())
            nd += got.size
            k += 1
        from vvenc_amd.hotpath import STATS_DTYPE
        for job in wl.tu_jobs:
            st = job[6].cpu().numpy().view(STATS_DTYPE).reshape(-1)
            res["mismatches"] += int((st["sse"] != J.outs[k]).sum())
            nt += st.size
            k += 1
        res["checked"] = {"distortion_candidates": nd, "tus_sse_after_fwd_quant_dequant_inv": nt}
    else:
        orc = O.Oracle()
        res["against"] = "the scalar C oracle (oracle/liboracle.so) on a 2% sample of every distortion list"
        org = np.ascontiguousarray(wl.org.storage.cpu().numpy())
        ref = np.ascontiguousarray(wl.ref.storage.cpu().numpy())
        L = orc.L
        L.orc_dist_batch.restype = None
        L.orc_dist_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        fidx = {"SSE": 0, "SAD": 1, "HAD": 2, "HAD_fast": 3}
        nd = 0
        for (func, S, ss, n, _, d_out, items) in wl.dist_jobs:
            sel = np.arange(0, n, 50)
            sl = np.ascontiguousarray(items[sel])
            outb = np.zeros(sel.size, np.uint64)
            L.orc_dist_batch(fidx[func], org.ctypes.data + 2 * wl.org.origin, wl.org.stride, ref.ctypes.data + 2 * wl.ref.origin, wl.ref.stride, S, S, ss, sl.ctypes.data, sel.size, outb.ctypes.data)
            res["mismatches"] += int((d_out.cpu().numpy().view(np.uint64)[sel] != outb).sum())
            nd += sel.size
        res["checked"] = {"distortion_candidates": nd}
    if mctf is not None:
        cur_np, ref_np, field = mctf
        h, w = cur_np.shape                                 # the whole picture (the scalar oracle takes a fraction of a second)
        t0 = time.perf_counter()
        orc = O.Oracle()
        exp = orc.mctf_me(np.ascontiguousarray(cur_np[:h, :w]), np.ascontiguousarray(ref_np[:h, :w]), wl.bit_depth, 16, 4, w >= 1920)[4]
        cp = hp.plane(np.ascontiguousarray(cur_np[:h, :w]), 128)
        rp = hp.plane(np.ascontiguousarray(ref_np[:h, :w]), 128)
        outs, dims = hp.mctf_motion_estimation(cp, [rp], wl.bit_depth, 16, 4, w >= 1920)
        got = hp.mv_to_numpy(outs[0], dims)
        bad = sum(int((got[k] != exp[k]).sum()) for k in ("x", "y", "error", "rmsme", "overlap"))
        res["mismatches"] += bad
        res["checked"]["mctf_motion_vectors"] = int(got.size)
        res["mctf_oracle_s"] = round(time.perf_counter() - t0, 2)
    res["status"] = "bit-exact" if res["mismatches"] == 0 else "MISMATCH"
    return res


# ------------------------------------------------------------------------------------------------------------------------ device side
def mctf_stage(hp, wl, refs=4, reps=5):
    """BASELINE configs[2] stage: hierarchical motion estimation of one picture against `refs` references (one vvhip_mctf_motion_estimation call: pyramid levels and search
    stages of all references share launches) and the bilateral filter of Y, U, V with the fields just found"""
    W, H, bd = wl.width, wl.height, wl.bit_depth
    cur = hp.plane(wl.cur_np, 128)
    ref_np = [np.roll(wl.ref_np, (k, -2 * k), (0, 1)) for k in range(refs)]
    ref_pl = [hp.plane(r, 128) for r in ref_np]
    add_level = W >= 1920
    out = {"width": W, "height": H, "references": refs, "unit": 16, "mctf_speed": 4}
    outs, dims = hp.mctf_motion_estimation(cur, ref_pl, bd, 16, 4, add_level)
    out["me_ms_per_picture"] = timed_ms(lambda: hp.mctf_motion_estimation(cur, ref_pl, bd, 16, 4, add_level, out=outs), reps)
    out["me_ms_one_reference"] = timed_ms(lambda: hp.mctf_motion_estimation(cur, ref_pl[:1], bd, 16, 4, add_level, out=outs[:1]), reps)
    # MCTF-filtered pictures are independent of each other (MCTF.cpp:666-724: originals only): k pictures in flight, one context + stream + host thread each (the call ends with a
    # host synchronisation: the hand-off abort flag).  The sweep of a picture (phase B) is one workgroup per reference: alone it leaves the device empty.
    import threading
    import time as _t
    for k in (2, 4):
        try:
            ctxs = [hp.fork(torch.cuda.Stream()) for _ in range(k)]
            fouts = [ctxs[i].mctf_motion_estimation(cur, ref_pl, bd, 16, 4, add_level)[0] for i in range(k)]
            torch.cuda.synchronize()

            def work(i, n):
                for _ in range(n):
                    ctxs[i].mctf_motion_estimation(cur, ref_pl, bd, 16, 4, add_level, out=fouts[i])
            n = max(2, reps)
            th = [threading.Thread(target=work, args=(i, n)) for i in range(k)]
            t0 = _t.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            out["me_ms_per_picture_%d_in_flight" % k] = 1000.0 * (_t.perf_counter() - t0) / (n * k)
            same = all(bool(torch.equal(fouts[i][r], outs[r])) for i in range(k) for r in range(refs))
            out["in_flight_fields_identical"] = bool(out.get("in_flight_fields_identical", True) and same)
            for c in ctxs:
                c.close()
        except Exception as e:
            out["me_in_flight_error"] = str(e)[:200]
    # bilateral filter: luma + both chroma planes (4:2:0), the fields stay on the device
    def yuv(y):
        return (y, np.clip(y[::2, ::2] // 2 + 256, 0, (1 << bd) - 1).astype(np.int16), np.clip((1 << bd) - 1 - y[::2, ::2] // 3, 0, (1 << bd) - 1).astype(np.int16))
    o3 = yuv(wl.cur_np)
    r3 = [yuv(r) for r in ref_np]
    planes_o = [hp.plane(o3[c], 128 >> (1 if c else 0)) for c in range(3)]
    planes_r = [[hp.plane(r[c], 128 >> (1 if c else 0)) for r in r3] for c in range(3)]
    outp = [hp.plane(np.zeros_like(o3[c]), 0) for c in range(3)]
    strengths = [hp.REF_STRENGTHS[0][min(k, 5)] for k in (0, 0, 1, 1)[:refs]]
    prm = [hp.mctf_filter_params(32, bd, 0.95, c > 0) for c in range(3)]
    mv_w = dims[0]

    def apply():
        for c in range(3):
            hp.mctf_apply_plane(planes_o[c], planes_r[c], outs, mv_w, 1 if c else 0, strengths, prm[c][1], prm[c][0], bd, 16, False, 32, out=outp[c])      # (6-tap: m_lowResFltApply is never set, MCTF.h:190)
    out["filter_ms_per_picture"] = timed_ms(apply, reps)
    nb = dims[0] * dims[1]
    out["blocks_final_level"] = nb
    out["bound"] = ("phase A (candidate scoring, all blocks of a level in parallel): VALU + LDS (4-tap separable interpolation per candidate, ~60 candidates per block on the final level); "
                    "phase B (above/left predictor test, MCTF.cpp:1289-1306): critical path of rows+cols dependent steps per level (%d+%d on the final level) x the hand-off latency; "
                    "HBM traffic is the pyramid (~1.33 x 2 planes x %d references), far below either" % (dims[1], dims[0], refs))
    out["hbm_bytes_unique"] = int(1.34 * W * H * 2 * (1 + refs))
    out["me_frac_hbm_unique"] = out["hbm_bytes_unique"] / (out["me_ms_per_picture"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out, (wl.cur_np, ref_np[0], None)


def unique_bytes(wl, cls):
    """HBM bytes one launch of the class must move at least: the planes it reads once + its index lists + its results"""
    planes = 2 * (wl.org.storage.numel() + wl.ref.storage.numel())
    if cls == "TU":
        return int(2 * wl.resi.storage.numel() + sum(j[1] * (4 + 4 + 2 * 2 * j[0] * j[0] + 24) for j in wl.tu_jobs))
    funcs = ("SAD", "SSE") if cls == "SAD_SSE" else (cls,)
    n = sum(j[3] for j in wl.dist_jobs if j[0] in funcs)
    return int(planes + n * (8 + 8))


def run_inner_profile(args, kind):
    """one rocprofv3 pass over a short inner run of this script; returns the rocpd database path"""
    outdir = os.path.join("/tmp", "vvhip_prof_%d_%s" % (os.getpid(), kind))
    shutil.rmtree(outdir, ignore_errors=True)
    prof = {"trace": ["--kernel-trace", "--stats"], "fetch": ["--pmc", "FETCH_SIZE"], "write": ["--pmc", "WRITE_SIZE"]}[kind]
    cmd = ["rocprofv3"] + prof + ["-d", outdir, "--", sys.executable, os.path.abspath(__file__), "--inner", "--steps", "10" if kind == "trace" else "4", "--warmup", "2",
                                  "--width", str(args.width), "--height", str(args.height)]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 %s pass: rc %d: %s" % (kind, r.returncode, r.stdout[-400:]))
    dbs = sorted(glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True), key=os.path.getmtime)
    if not dbs:
        raise RuntimeError("rocprofv3 %s pass left no database" % kind)
    return dbs[-1], outdir


def live_profile(args):
    """kernel trace + the two PMC passes of a short inner run (separate passes, as the MI355X guide prescribes); FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 B)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_round as P
    out = {}
    db, d1 = run_inner_profile(args, "trace")
    rows = P.kernel_table(db)
    tot = sum(r[2] for r in rows) or 1
    out["kernel_trace"] = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --inner --steps 10 --warmup 2 (3 frame launches per step + 2 MCTF stages)",
                           "kernels": [{"name": k.replace("(anonymous namespace)::", "")[:90], "calls": n, "avg_us": round(av / 1e3, 2), "total_us": round(s / 1e3, 1), "pct": round(100.0 * s / tot, 1)}
                                       for k, n, s, av, mn, mx in rows[:14]]}
    classes = {}
    dirs = [d1]
    for kind, key in (("fetch", "fetch_kb"), ("write", "write_kb")):
        db, d = run_inner_profile(args, kind)
        dirs.append(d)
        for k, c, n, s, av in P.counter_table(db):
            cls = P.class_of(k)
            if cls:
                e = classes.setdefault(cls, {"fetch_kb": 0.0, "write_kb": 0.0, "n_fetch_kb": 0, "n_write_kb": 0})
                e[key] += s
                e["n_" + key] += n
    out["pmc"] = {cls: {"fetch_bytes_per_launch_x2_corrected": 2.0 * 1024.0 * e["fetch_kb"] / max(1, e["n_fetch_kb"]), "write_bytes_per_launch": 1024.0 * e["write_kb"] / max(1, e["n_write_kb"])}
                  for cls, e in classes.items()}
    for c in out["pmc"].values():
        c["traffic_bytes_per_launch"] = c["fetch_bytes_per_launch_x2_corrected"] + c["write_bytes_per_launch"]
    for d in dirs:
        shutil.rmtree(d, ignore_errors=True)
    return out


def file_pmc(cls, args):
    """fallback: the committed PMC passes of the same command (profiles/pmc_r0x.json)"""
    if (args.width, args.height) != (1920, 1080):
        return None, None
    for name in ("pmc_r02.json", "pmc_r01.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return d["classes"][cls]["traffic_bytes_per_launch"], "profiles/" + name
        except Exception:
            continue
    return None, None


def e2e_encoder(frames, threads):
    """the real reference encoder end to end (SURVEY §8d metric): CPU kernels vs --SIMD=HIP, same clip, same threads; subprocesses (the SIMD level is process-wide)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_fps
    import e2e_util
    if not (os.path.exists(e2e_util.REF_SO) and os.path.exists(e2e_util.REF_HIP_SO)):
        return {"skipped": "oracle/_ref (the compiled reference encoder, with and without the binding) is not built"}
    prod = 16 + 128 + 8192 + 65536
    # one discarded run (clip cache, page cache, clocks), then five alternating pairs; medians (single runs of a 1.4 s encode scatter by +-4 %)
    e2e_fps.run(dict(w=1920, h=1080, frames=frames, threads=threads, mask=0), timeout=600)
    runs = [e2e_fps.run(dict(w=1920, h=1080, frames=frames, threads=threads, mask=m), timeout=600) for m in (0, prod) * 5]
    med = lambda v: sorted(v)[len(v) // 2]
    cpu = med([r["fps"] for r in runs if r["mask"] == 0])
    hip = med([r["fps"] for r in runs if r["mask"] == prod])
    return {"clip": "1920x1080 10-bit synthetic (config-2 generator), %d frames, preset faster, QP 32" % frames, "threads": threads,
            "cpu_fps": round(cpu, 2), "hip_fps": round(hip, 2), "speedup": round(hip / cpu, 3), "runs_fps": [round(r["fps"], 2) for r in runs],
            "runs_order": "cpu, hip alternating, five pairs after one discarded run; cpu_fps / hip_fps are medians",
            "bitstreams_identical": len({r["md5"] for r in runs}) == 1, "md5": runs[0]["md5"],
            "device_stages": "MCTF motion estimation (all references of a picture per call) + bilateral filter, ALF statistics + ALF filtering of whole pictures (--SIMD=HIP production mask %d)" % prod,
            "pcie_MB_per_picture": runs[1].get("pcie_MB_per_picture"), "median_of": 5,
            "note": "the encoder's CTU-level control flow (mode decision, CABAC, RDOQ) stays on the host and bounds the gain (SURVEY §6: the hot path is 30-35% of one thread)"}


# ------------------------------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip per-kernel HIP events inside the timed region")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-mctf", action="store_true")
    ap.add_argument("--no-4k", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the rocprofv3 passes of the inner run (kernel trace, FETCH_SIZE, WRITE_SIZE)")
    ap.add_argument("--e2e-frames", type=int, default=65)
    ap.add_argument("--e2e-threads", type=int, default=8)
    ap.add_argument("--graph", type=int, default=1, help="extra measurement: the frame's launches replayed from a HIP graph (0 = skip)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the three launches of a picture are issued on (1 = one stream, serialized)")
    ap.add_argument("--with-subpel", action="store_true", help="also run the fractional-ME stage per step (16 interpolated HAD_fast candidates per block; SURVEY 8f rank 1)")
    ap.add_argument("--static-copies", action="store_true", help="make the derived plane copies (tiled, shifted) once outside the timed region instead of per step")
    ap.add_argument("--inner", action="store_true", help="(internal) the short run rocprofv3 wraps: frame launches + MCTF stages, no extras, no output line")
    args = ap.parse_args()

    rank, local_rank, world = sharding.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: vvenc_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    from vvenc_amd.hotpath import HotPath, Plane
    from vvenc_amd.workload import FrameWorkload
    hp = HotPath("cuda:%d" % local_rank)
    # rank r works on picture r + step*world of ONE sequence (pan of the same texture); the reference picture is the same on every rank
    wl = FrameWorkload(hp, args.width, args.height, seed=1080, frame_index=rank)
    if args.with_subpel:
        wl.enable_subpel()

    if args.inner:
        # the launches of a step, serialized on one stream: the two derive launches of the picture's plane copies (as the lanes issue them) + the three list launches
        sh2 = torch.empty_like(wl.ref.storage) if wl.shifted else None
        for _ in range(args.warmup + args.steps):
            if (wl.tiled or wl.shifted) and not args.static_copies:
                hp.planes_derive(wl.org, wl.ref, wl.org_tiled if wl.tiled else None, wl.ref_tiled if wl.tiled else None, wl.ref_shift if wl.shifted else None)
                if wl.shifted:
                    hp.planes_derive(wl.org, wl.ref, None, None, sh2)
            wl.run(None)
        mctf_stage(hp, wl, 4, reps=2)
        torch.cuda.synchronize()
        return

    # ---- the reference-picture exchange of the sharded sequence (N > 1): DPB ring of two pictures (luma + 2 chroma planes with margins)
    ex, ref_planes = None, None
    if world > 1:
        shp = tuple(wl.ref.storage.shape)
        cshape = (shp[0] // 2, shp[1] // 2)
        ex = sharding.PictureExchange([shp, cshape, cshape], slots=2, device=hp.device)
        ref_planes, ref_tiled, ref_shift = [], [], []
        for s in range(2):
            ex.slots[s][0].copy_(wl.ref.storage)
            pl = Plane(hp.device, wl.ref.width, wl.ref.height, wl.ref.pad, wl.ref.stride)
            pl.storage = ex.slots[s][0]
            ref_planes.append(pl)
            ref_tiled.append(hp.tile_plane(pl) if wl.tiled else None)      # (for the serialized extras; the timed steps derive their own, see `derive`)
            ref_shift.append(hp.shift_plane(pl) if wl.shifted else None)

        ex.publish(0, 0)

    step_no = [0]
    # Every step is a different picture, so the copies this library derives from a picture's planes (8x8-tiled original and reference, one-sample-shifted reference) are made
    # INSIDE the step, by the lanes that read them, in front of their launches (--static-copies: made once, outside the timed region — the regime of the `single_stream` / `graph` extras)
    derive = (wl.tiled or wl.shifted) and wl.merged and args.streams > 1 and (world > 1 or not args.static_copies)      # (N > 1: a received reference picture always gets its copies)
    # the three launches of a picture are independent work lists: each goes to its own HIP stream (they share the device, and the steps pipeline per stream)
    streams = [torch.cuda.Stream() for _ in range(3)] if (args.streams > 1 and wl.merged) else None

    def step(timers=None):
        if ex is not None:
            s = step_no[0]
            ex.publish(s + 1, (s + 1) % world, readers=streams or ())     # the next picture's reference is in flight while this picture's launches run
            ex.wait(s, streams)
            wl.ref, wl.ref_tiled, wl.ref_shift = ref_planes[s % 2], ref_tiled[s % 2], ref_shift[s % 2]
            step_no[0] += 1
        if streams:
            wl.run_overlapped(streams, timers, derive=derive)
        else:
            wl.run(timers)

    classes = wl.class_launches_merged if wl.merged else wl.class_launches
    for _ in range(args.warmup):
        step(None)
    torch.cuda.synchronize()
    # Untimed settling of the device (besides the W warm-up steps): the FIRST process on a freshly acquired box stalls ~30 ms once, a few milliseconds into its first phase of
    # multi-queue concurrency (measured: 50 steps drain in 34 ms instead of 2 ms; any earlier GPU process on the box, however small, removes it).  Batches of the same steps until
    # two consecutive batches agree and at least 150 ms have passed; world > 1: a fixed count, so that every rank publishes the same pictures.
    if world == 1:
        tw, prev, batches = time.perf_counter(), None, 0
        while True:
            tb = time.perf_counter()
            for _ in range(50):
                step(None)
            torch.cuda.synchronize()
            cur = time.perf_counter() - tb
            batches += 1
            if (time.perf_counter() - tw > 0.15 and prev is not None and abs(cur - prev) < 0.2 * min(cur, prev)) or batches >= 200:
                break
            prev = cur
    else:
        for _ in range(100):
            step(None)
        torch.cuda.synchronize()
    # K steps with the three launches SERIALIZED on one stream and every class bracketed by events: the regime in which a kernel's launch duration is its own (roofline),
    # and the one the rocprofv3 trace of the inner run shows.  Outside the timed region (events are not free: ~3 us of host time each).
    stimers, dom_cls = None, None
    if not args.no_kernel_timers:
        wl.run(None)
        stimers = EventTimers(list(classes), args.steps, classes)
        for _ in range(args.steps):
            wl.run(stimers)
        torch.cuda.synchronize()
        ssum = stimers.summary()
        dom_cls = max(ssum, key=lambda k: ssum[k]["total_ms"])
        for _ in range(2):
            step(None)
        torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    # timed region: only the dominant class keeps an event pair, on its own stream (its launch duration while the classes share the device)
    timers = EventTimers([dom_cls], args.steps, classes) if dom_cls else None
    enq = [0.0] * (args.steps + 1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(timers)
        enq[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    enq[0] = t0
    enq_us = sorted(1e6 * (b - a) for a, b in zip(enq[:-1], enq[1:]))
    dt = sharding.max_over_ranks(dt, device="cuda")
    if ex is not None:
        wl.ref, wl.ref_tiled, wl.ref_shift = ref_planes[0], ref_tiled[0], ref_shift[0]

    # extra (not `value`): the same K steps on the three streams with the derived plane copies made ONCE (what a sequence of lists against the same picture costs)
    static_copies = None
    if streams and derive and world == 1:
        torch.cuda.synchronize()
        for _ in range(max(args.warmup, 1)):
            wl.run_overlapped(streams, None, derive=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.run_overlapped(streams, None, derive=False)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        static_copies = {"value": args.steps / dts, "unit": "frames/s", "ms_per_step": 1000.0 * dts / args.steps,
                         "note": "same three launches on three streams, the tiled / shifted plane copies made once outside the timed steps (lists of one picture); not the headline value"}
    # extra (not `value`): the same K steps serialized on one stream without any event
    overlap = None
    if streams:
        for _ in range(max(args.warmup, 1)):
            wl.run(None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.run(None)
        torch.cuda.synchronize()
        dto = sharding.max_over_ranks(time.perf_counter() - t1, device="cuda")
        overlap = {"streams": 1, "value": args.steps * world / dto, "unit": "frames/s", "ms_per_step": 1000.0 * dto / args.steps,
                   "note": "same work, the 3 launches of a picture serialized on one HIP stream (no events, no picture exchange, derived plane copies made once); not the headline value"}
    graph = None
    if args.graph and wl.merged and not args.with_subpel:
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
        graph = {"error": err or "failed on another rank"} if bad > 0.0 else \
            {"value": args.steps * world / dtg, "unit": "frames/s", "ms_per_step": 1000.0 * dtg / args.steps,
             "note": "same work, the 3 launches of a frame captured once as a HIP graph and replayed (no per-kernel events, no picture exchange); not the headline value"}
    if rank != 0:
        return

    frames = args.steps * world
    out = {
        "metric": "frames/sec + bit-exact vs CPU, 1080p/4K 10-bit preset=faster at 1/2/4/8 GPU",
        "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i16", "data": "synthetic",
        "host_enqueue_us_per_step": {"p50": round(enq_us[len(enq_us) // 2], 1), "p90": round(enq_us[int(len(enq_us) * 0.9)], 1), "max": round(enq_us[-1], 1),
                                     "drain_ms_after_last_enqueue": round(1000.0 * (t0 + dt_local - enq[-1]), 3)},
        "config": {"workload": "%dx%d 10-bit synthetic picture, preset=faster hot-path work lists: SAD/SATD(HAD_fast)/SSE candidate batches "
                               "(8..64 blocks, 20 candidates/block) + fused DCT-2/quant/dequant/IDCT TU batches (8..32); BASELINE configs[1]" % (args.width, args.height),
                   "sample_pairs_per_frame": int(wl.pairs), "coefficients_per_frame": int(wl.coefs), "launches_per_frame": (5 if derive else 3) if wl.merged else len(wl.dist_jobs) + len(wl.tu_jobs), "hip_streams": len(streams) if streams else 1,
                   "sharding": "one picture per rank and step, pictures of one sequence round-robin over ranks, no data-path collective"
                               + (", reference picture (luma + chroma, %.1f MB) RCCL-broadcast from its owner every step inside the timed region, overlapped with the launches"
                                  % (sum(p.numel() * 2 for p in ex.slots[0]) / 1e6) if ex is not None else ""),
                   "subpel_candidates_per_block": 16 if args.with_subpel else 0},
    }
    if static_copies:
        out["static_derived_copies"] = static_copies
    if ex is not None:
        out["exchange"] = {"bytes_per_rank": int(ex.bytes_published), "pictures": step_no[0] + 1, "collective": "broadcast (RCCL)", "overlapped": True}
    if wl.tiled or wl.shifted:
        out["derived_copies"] = ("8x8-tiled original + reference and one-sample-shifted reference (SAD / SSE lane, one launch) + one-sample-shifted reference (Hadamard lane): made for every "
                                 "step's picture INSIDE the timed region, on the lane that reads them" if derive else "made once outside the timed region (--static-copies / one stream)")

    live = None
    if not args.no_profile and world == 1 and shutil.which("rocprofv3"):
        try:
            live = live_profile(args)
            out["kernel_trace"] = live["kernel_trace"]
        except Exception as e:
            out["kernel_trace"] = {"error": str(e)[:300]}

    if stimers is not None:
        ks = stimers.summary()                                # every class, K steps, launches serialized on one stream
        conc = timers.summary() if timers is not None else {}
        for k in ks:
            ks[k]["alg_bytes_per_frame"] = int(wl.alg_bytes[k])
            ks[k]["alg_GBps"] = wl.alg_bytes[k] * args.steps / (ks[k]["total_ms"] * 1e-3) / 1e9
            ks[k]["measured_over"] = "%d steps, launches serialized on one stream" % args.steps
            if k in conc:
                ks[k]["avg_ms_in_timed_region"] = conc[k]["avg_ms"]      # on its own stream, concurrently with the other two classes
        for k in ks:
            ub = unique_bytes(wl, k) if k in ("SAD_SSE", "HAD_fast", "TU") else None
            if ub:
                ks[k]["unique_bytes_per_launch"] = ub
                ks[k]["frac_hbm_unique"] = ub / (ks[k]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                ks[k]["frac_l2"] = ks[k]["alg_GBps"] / L2_PEAK_GBS
        out["kernels"] = ks
        lpf = classes[dom_cls]
        traffic, tsrc = None, None
        if live and dom_cls in live.get("pmc", {}):
            traffic, tsrc = live["pmc"][dom_cls]["traffic_bytes_per_launch"], "this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over the inner run (FETCH_SIZE x2, gfx950)"
            out["pmc"] = live["pmc"]
        else:
            traffic, tsrc = file_pmc(dom_cls, args)
        avg_s = ks[dom_cls]["avg_ms"] * 1e-3
        alg_launch = wl.alg_bytes[dom_cls] / lpf
        out["roofline"] = {"bound": "hbm", "kernel": KERNEL_OF_CLASS.get(dom_cls, dom_cls),
                           "achieved": ks[dom_cls]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ks[dom_cls]["alg_GBps"] / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": tsrc,
                           "basis": "NOMINAL: per-candidate algorithmic bytes (4*w*h per candidate + 8 B result, rows halved under subShift; fused TU = 6*w*h + 24 B; SURVEY 8d) / HIP-event launch time. "
                                    "Candidates of a block share their original block and overlap in the reference window, so most of these bytes are served by L2 / Infinity Cache; the physical positions are the next three fields",
                           "alg_bytes_per_launch": alg_launch, "avg_launch_ms": ks[dom_cls]["avg_ms"],
                           "avg_launch_ms_measured": "HIP events on the launch stream, %d steps with the picture's launches serialized (the kernel alone on the device; the rocprofv3 trace of the inner run shows the same regime); "
                                                     "in the timed region the three classes run concurrently on three streams and this class's launches last avg_launch_ms_concurrent" % args.steps,
                           "avg_launch_ms_concurrent": ks[dom_cls].get("avg_ms_in_timed_region"),
                           "aggregate_alg_GBps_timed_region": sum(wl.alg_bytes[k] for k in classes) * frames / dt / 1e9,
                           "frac_l2": ks[dom_cls]["alg_GBps"] / L2_PEAK_GBS, "l2_peak_GBps": L2_PEAK_GBS,
                           "unique_bytes_per_launch": ks[dom_cls].get("unique_bytes_per_launch"), "frac_hbm_unique": ks[dom_cls].get("frac_hbm_unique"),
                           "frac_hbm_traffic": (traffic / avg_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                           "limiter": "L1 (TCP) access rate and VALU issue, not a memory level: one L1 access per 64-byte granule and instruction (rocprofv3 TCP_TOTAL_CACHE_ACCESSES: 9.7 M per "
                                      "launch = 15.8 us at 256 CUs x 2.4 GHz) next to 7.8 M VALU wave-instructions (12.8 us of the SIMDs); frac > 1 on the nominal basis only says that the "
                                      "candidates' bytes are re-read from L1 / L2, never from HBM (DESIGN §6, profiles/r02_d_pmc_merged_launches.log)"}
    if overlap is not None:
        out["single_stream"] = overlap
    if graph is not None:
        out["graph"] = graph

    mctf_parity = None
    if not args.no_mctf and world == 1:
        try:
            out["mctf"], mctf_parity = mctf_stage(hp, wl, 4)
        except Exception as e:
            out["mctf"] = {"error": str(e)[:300]}
    if not args.no_parity:
        try:
            wl.run(None)
            torch.cuda.synchronize()
            out["parity"] = parity_check(hp, wl, mctf_parity)
        except Exception as e:
            out["parity"] = {"status": "not checked", "error": str(e)[:300]}
    if not args.no_4k and world == 1 and (args.width, args.height) == (1920, 1080):
        try:
            w4 = FrameWorkload(hp, 3840, 2160, seed=2160)
            # per-class launch durations: launches serialized on one stream, every class bracketed by events (the regime of the 1080p `kernels` block) ...
            t4 = EventTimers(list(w4.class_launches_merged), 12, w4.class_launches_merged)
            w4.run(None)
            for _ in range(10):
                w4.run(t4)
            torch.cuda.synchronize()
            s4 = t4.summary()
            # ... and the picture rate with the three launches on their streams, no events (the regime of `value`)
            run4 = (lambda t: w4.run_overlapped(streams, t, derive=derive)) if streams else w4.run
            run4(None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                run4(None)
            torch.cuda.synchronize()
            d4 = time.perf_counter() - t1
            m4, _ = mctf_stage(hp, w4, 4, reps=3)
            out["pass_4k"] = {"config": "BASELINE configs[2]: 3840x2160 10-bit picture, the same three launches + the MCTF stage (hierarchical ME vs 4 references, 5 pyramid levels; bilateral filter)",
                              "frame_launches_ms_per_step": 1000.0 * d4 / 10, "frames_per_s_launches_only": 10 / d4,
                              "kernels": {k: {"avg_ms": v["avg_ms"], "alg_GBps": w4.alg_bytes[k] / (v["avg_ms"] * 1e-3) / 1e9, "frac_l2": w4.alg_bytes[k] / (v["avg_ms"] * 1e-3) / 1e9 / L2_PEAK_GBS,
                                              "frac_hbm_unique": unique_bytes(w4, k) / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS} for k, v in s4.items()},
                              "mctf": m4,
                              "ms_per_picture_with_mctf": 1000.0 * d4 / 10 + m4["me_ms_per_picture"] + m4["filter_ms_per_picture"]}
            del w4
        except Exception as e:
            out["pass_4k"] = {"error": str(e)[:300]}
    if not args.no_e2e and world == 1:
        try:
            out["e2e"] = e2e_encoder(args.e2e_frames, args.e2e_threads)
        except Exception as e:
            out["e2e"] = {"error": str(e)[:300]}
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(wl)
        except Exception as e:   # the baseline is a report, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
