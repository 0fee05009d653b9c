#!/usr/bin/env python3
"""bench.py — pictures/sec of the MI355X hot path on work lists RECORDED from the reference encoder (1080p = `value`, 3840x2160 = `value_4k`), with a physical roofline,
in-run parity against the encoder's own values, the MCTF stage, the end-to-end encoder (1080p and 4K; N > 1: one encoder instance per GPU) and the CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

`value` (BASELINE configs[1]): a step = ONE picture's hot-path work exactly as the reference encoder produced it.  Before the clock starts the encoder built with the
binding (bindings/vvenc) encodes the 1920x1080 10-bit config-2 clip (65 frames, preset faster) on its CPU kernels with the work-list recorder on (hook bit 131072): every call
through RdCost's table, every InterSearch::xMotionEstimation with its integer candidates and xPatternRefinement stages, every TU of TrQuant::xT with its residual, every DMVR
sub-block — for one picture of each temporal layer.  The lists + the pictures' planes are uploaded once; step s replays the picture of layer
STEP_LAYERS[s mod 32]: a low-discrepancy interleaving of the GOP's layer mix (1 x TL0, 1 x TL1, 2 x TL2, 4 x TL3, 8 x TL4, 16 x TL5 per 32) that starts at the key picture, so any
prefix of K steps is close to the mix (20 steps = 1 x TL0 (intra), 0 x TL1, 1 x TL2, 3 x TL3, 5 x TL4, 10 x TL5: time-weighted within ~1 % of the cycle's mean):
    motion-search plan   integer candidates (LDS windows) + sub-pel refinement stages (interpolation fused with the Hadamard) + merge / AMVP / intra / SSE table calls
    TU lists             fused xT -> needRdoq -> quant -> dequant -> xIT -> SSE, luma + chroma, DCT-2 / DST-7, 4..64
    DMVR lists           bilinear prediction + 25-point search + error surface per sub-block
on five HIP streams.  With N GPUs rank r replays position k + 32 r / N of the same cycle at its step k (N different pictures of one sequence at any time); the reconstructed picture a sharded encoder would hand to the ranks encoding the
pictures that reference it is broadcast (RCCL) every --exchange-every steps inside the timed region, overlapped; value = N * K pictures / max-over-ranks time, "weak".

Extra objects of the JSON line (rank 0; each can be switched off; a failure is reported in place and never costs the headline):
  value_4k ...  the same replay on lists recorded from the 3840x2160 x 65 encode (BASELINE configs[2]'s geometry): value_4k, ms_per_step_4k, kernels_4k, roofline_4k, parity_4k, cpu_baseline_4k
  roofline      dominant kernel: PHYSICAL position — fabric traffic per launch from this run's own rocprofv3 --pmc passes (FETCH_SIZE x the factor calibrated on this GPU for the
                kernel's access pattern, tools/calib_fetch.py, + WRITE_SIZE) / launch time / 8 TB/s — next to the L1 access and VALU issue fractions that actually bound these
                kernels; unique_bytes (the union of what the launch reads and writes) and traffic_over_unique; SURVEY 8d's per-candidate figure is kept as nominal_alg_GBps
  kernels       every kernel of a step: launches and average duration per layer and GOP-weighted, algorithmic and unique bytes
  parity        every value the timed launches produced against (a) the costs the REAL encoder computed while it was recorded and (b) the reference's x86-SIMD entries
                driven over the same TU lists (SSE, abs sums, last scan positions, need-RDOQ flags, level checksums): "bit-exact" or the mismatch count
  mctf          BASELINE configs[2] stage at 1080p and 4K: hierarchical ME against 4 references + bilateral filter, ms per picture
  e2e / e2e_4k  the real encoder, 1080p x 65 and 3840x2160 x 65, preset faster: CPU kernels vs --SIMD=HIP, fps + bitstream md5 equality
  e2e_instances N > 1: one encoder instance per rank / GPU over GOP chunks of one sequence (EncoderLib/EncGOP.cpp:1647-1651 chunking), aggregate fps CPU vs --SIMD=HIP, per-chunk md5
  config3_medium_4k  one 3840x2160 picture of a preset-medium encode (BASELINE configs[3]: CTU 128, rectangular blocks, GEO) through the same path: nothing dropped, bit-exact, ms per picture
  cpu_baseline  the reference's own AVX2 entries on the host cores over the same recorded lists: median of 5 passes per layer on pinned threads, GOP-weighted
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# a picture's five launch groups go to five HIP streams: the runtime maps streams onto 4 hardware queues by default (two groups would share one and serialize);
# measured on the recorded 1080p lists: 2 / 4 / 8 queues -> 8 519 / 12 036 / 12 326 pictures/s
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vvenc_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
CLOCK_GHZ, N_CU, N_SIMD = 2.4, 256, 1024
LAYER_POCS = {0: 31, 1: 15, 2: 23, 3: 3, 4: 5, 5: 2}          # one recorded picture per temporal layer of the 65-frame encode (its GOP anchors at POC 31 / 63)
KERNEL_NAMES = {"ME_stage": "meStageKernel", "ME_int": "meIntKernel", "ME_item": "meItemKernel", "TU": "tuMxMultiKernel", "DMVR": "dmvrRefineKernel"}


# The replay order of the 32 pictures of a GOP cycle (1 x TL0, 1 x TL1, 2 x TL2, 4 x TL3, 8 x TL4, 16 x TL5): a low-discrepancy interleaving, NOT the coding order — the recorded
# pictures are independent work for the device, and a run of K steps should hold the layers close to their GOP share whatever K is.  The cycle starts at its key (intra) picture;
# the first 20 positions hold 1 x TL0, 0 x TL1, 1 x TL2, 3 x TL3, 5 x TL4, 10 x TL5 (time-weighted within ~1 % of the whole cycle's mean on the recorded 1080p lists), 32 = the exact mix.
STEP_LAYERS = (0, 5, 4, 5, 3, 5, 4, 5, 2, 5, 4, 5, 3, 5, 4, 5, 3, 5, 4, 5, 2, 5, 4, 5, 1, 5, 4, 5, 3, 5, 4, 5)


def layer_of_step(s):
    """temporal layer of the picture step s replays"""
    return STEP_LAYERS[s % 32]


def step_of_rank(k, rank, world):
    """N ranks: rank r's k-th step is position k + r * (32 / N) of the same cycle — at any time the ranks work on N different pictures of one sequence, and every rank's window
    of K steps holds (nearly) the same layer mix (taking every N-th position instead would hand one rank all the heavy layers: the odd positions are all TL5)"""
    return k + (rank * 32) // max(1, world)


GOP_WEIGHT = {l: sum(1 for s in range(32) if layer_of_step(s) == l) for l in range(6)}


# ---------------------------------------------------------------------------------------------------------------------- recording
def prepare_recordings(width, height, frames, pocs, tag="faster", threads=8):
    """the recorded lists of the pictures `pocs` (cached under /tmp: the rocprofv3 passes and later runs reuse them)"""
    from vvenc_amd import recorded as R
    d = os.path.join("/tmp", "vvhip_rec_%dx%d_%d_%s" % (width, height, frames, tag))
    info = {"dir": d, "recorded_now": False}
    need = [p for p in pocs if not os.path.exists(os.path.join(d, "poc%d.json" % p))]
    if need:
        t0 = time.perf_counter()
        res = R.record(d, width, height, frames, pocs=need, threads=threads, preset=tag)
        info.update(recorded_now=True, record_s=round(time.perf_counter() - t0, 2), encoder_md5=res["md5"], encoder_s=round(res["secs"], 2))
    return {p: R.RecordedPicture(os.path.join(d, "poc%d" % p)) for p in pocs}, info


# ---------------------------------------------------------------------------------------------------------------------- reference twin (parity + CPU baseline)
class RecJob(C.Structure):
    _fields_ = [("kind", C.c_int32), ("df", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("subShift", C.c_int32), ("trHor", C.c_int32), ("trVer", C.c_int32), ("n", C.c_int32),
                ("org", C.c_void_p), ("cur", C.c_void_p), ("orgStride", C.c_int32), ("curStride", C.c_int32), ("items", C.c_void_p), ("aux", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p)]


REC_STAGE = np.dtype([("org_off", "<i4"), ("ref_off", "<i4"), ("base_qx", "i1"), ("base_qy", "i1"), ("i_frac", "u1"), ("filter_mode", "u1"), ("alt_hpel", "u1"), ("had_mode", "u1"), ("mask", "<u2")])


class ReferenceJobs:
    """a recorded picture's lists as job records of oracle/_ref's multi-threaded driver (vvref_run_recorded_mt): the reference's own x86-SIMD table entries on host copies of
    the same planes, pool and lists the device replays.  Test infrastructure: used by the parity check and the cpu_baseline leg only."""

    def __init__(self, wl, with_outputs):
        from oracle import oracle as O
        self.L = O.RefLib(1).L
        self.L.vvref_run_recorded_mt.restype = C.c_double
        self.L.vvref_run_recorded_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        wl = getattr(wl, "lists", wl)                       # the HOST lists (vvenc_amd.replay.RecordedLists) of a device workload
        self.wl, self.keep, jobs = wl, [], []
        host_planes = [pl.storage for pl in wl.planes]
        pool = wl.pool

        def base(idx, w):
            """(address of sample (0,0), row pitch) of plane-table entry idx for blocks of width w (the pool holds compact blocks: pitch = width)"""
            if idx < wl.n_pic_planes:
                pl = wl.planes[idx]
                return host_planes[idx].ctypes.data + 2 * pl.origin, pl.stride
            return pool.ctypes.data, int(w)
        df_of = {0: 0, 1: 8, 2: 16, 3: 26, 4: 24}          # C ABI function code -> DFunc base of the reference's table (TypeDef.h:339-382)
        # integer candidates + plain table calls as distortion lists grouped by (function, size, subShift, operand planes)
        ij, pc = wl.int_jobs, wl.plan_cands
        recs = []
        if pc.size:
            jidx = np.repeat(np.arange(ij.size), ij["n_cand"])
            ref_stride = np.array([base(int(p), 0)[1] for p in ij["ref_plane"]], np.int64)
            cur_off = ij["ref_off"][jidx].astype(np.int64) + pc["dy"].astype(np.int64) * ref_stride[jidx] + pc["dx"]
            recs.append(np.stack([np.full(pc.size, 1), ij["width"][jidx], ij["height"][jidx], ij["sub_shift"][jidx], ij["org_plane"][jidx], ij["ref_plane"][jidx], ij["org_off"][jidx], cur_off], 1).astype(np.int64))
        it = wl.items
        if it.size:
            recs.append(np.stack([it["func"], it["width"], it["height"], it["sub_shift"], it["org_plane"], it["cur_plane"], it["org_off"], it["cur_off"]], 1).astype(np.int64))
        self.dist_groups = []
        if recs:
            allr = np.concatenate(recs)
            key = allr[:, :6]
            uniq, inv = np.unique(key, axis=0, return_inverse=True)
            inv = inv.ravel()
            for g, (func, w, h, ss, po, pcu) in enumerate(uniq):
                sel = np.nonzero(inv == g)[0]
                items = np.ascontiguousarray(allr[sel][:, 6:8].astype(np.int32))
                out = np.zeros(sel.size, np.uint64) if with_outputs else None
                (ob, os_), (cb, cs) = base(int(po), w), base(int(pcu), w)
                if int(func) == 4:
                    # HAD_2SAD's SAD part assumes compact, 32-byte aligned operands (CHECKD + _mm256_load_si256, x86/RdCostX86.h:2556-2600; the encoder calls it on IntraSearch's
                    # compact buffers): gather both operands of the list into aligned compact buffers for the reference entry
                    w_, h_ = int(w), int(h)
                    yy, xx = np.mgrid[0:h_, 0:w_]

                    def gather(pidx, offs):
                        if pidx < wl.n_pic_planes:
                            pl = wl.planes[pidx]
                            flat, o0, st = pl.storage.reshape(-1), pl.origin, pl.stride
                        else:
                            flat, o0, st = pool, 0, w_
                        idx = (o0 + offs.astype(np.int64))[:, None, None] + yy[None] * st + xx[None]
                        buf = np.zeros(sel.size * w_ * h_ + 32, np.int16)
                        shift = (-buf.ctypes.data // 2) % 16                     # first sample at a 32-byte boundary
                        buf[shift:shift + sel.size * w_ * h_] = flat[idx].reshape(-1)
                        self.keep.append(buf)
                        return buf.ctypes.data + 2 * shift
                    ob, cb = gather(int(po), items[:, 0]), gather(int(pcu), items[:, 1])
                    os_ = cs = w_
                    items = np.ascontiguousarray(np.stack([np.arange(sel.size) * w_ * h_] * 2, 1).astype(np.int32))
                self.keep += [items, out]
                self.dist_groups.append((sel, out))
                jobs.append(RecJob(0, df_of[int(func)], int(w), int(h), int(ss), 0, 0, sel.size, ob, cb, os_, cs, items.ctypes.data, None, out.ctypes.data if out is not None else None, None))
        # masked SADs (GEO): grouped by (size, subShift, operand planes); the weight blocks are compact pool blocks
        mi = getattr(wl, "mask_items", np.zeros(0))
        self.mask_groups = []
        if mi.size:
            key = np.stack([mi["width"], mi["height"], mi["sub_shift"], mi["org_plane"], mi["cur_plane"]], 1).astype(np.int64)
            uniq, inv = np.unique(key, axis=0, return_inverse=True)
            inv = inv.ravel()
            for g, (w, h, ss, po, pcu) in enumerate(uniq):
                sel = np.nonzero(inv == g)[0]
                items = np.ascontiguousarray(np.stack([mi["org_off"][sel], mi["cur_off"][sel], mi["mask_off"][sel]], 1).astype(np.int32))
                out = np.zeros(sel.size, np.uint64) if with_outputs else None
                (ob, os_), (cb, cs) = base(int(po), w), base(int(pcu), w)
                self.keep += [items, out]
                self.mask_groups.append((sel, out))
                jobs.append(RecJob(3, 25, int(w), int(h), int(ss), 0, 0, sel.size, ob, cb, os_, cs, items.ctypes.data, pool.ctypes.data, out.ctypes.data if out is not None else None, None))
        self.n_cands = int(pc.size)
        # TU lists
        self.tu_outs = []
        for g in wl.tu_groups:
            off = np.ascontiguousarray(g["off"])
            qf = np.ascontiguousarray(g["qf"])
            out = np.zeros(g["n"], np.uint64) if with_outputs else None
            out2 = np.zeros((g["n"], 4), np.int32) if with_outputs else None
            self.keep += [off, qf, out, out2]
            self.tu_outs.append((out, out2))
            jobs.append(RecJob(1, 0, g["w"], g["h"], 0, g["tr_hor"], g["tr_ver"], g["n"], pool.ctypes.data, None, g["w"], 0, off.ctypes.data, qf.ctypes.data,
                               out.ctypes.data if out is not None else None, out2.ctypes.data if out2 is not None else None))
        # refinement stages grouped by (size, planes)
        sj = wl.stage_jobs
        self.stage_groups = []
        if sj.size:
            key = np.stack([sj["width"], sj["height"], sj["org_plane"], sj["ref_plane"]], 1).astype(np.int64)
            uniq, inv = np.unique(key, axis=0, return_inverse=True)
            inv = inv.ravel()
            for g, (w, h, po, pr) in enumerate(uniq):
                sel = np.nonzero(inv == g)[0]
                st = np.zeros(sel.size, REC_STAGE)
                for f in ("org_off", "ref_off", "base_qx", "base_qy", "i_frac", "filter_mode", "alt_hpel", "mask"):
                    st[f] = sj[f][sel]
                st["had_mode"] = np.array([0, 0, 1, 2, 0], np.uint8)[sj["func"][sel]]          # SSE(unused) / SAD -> 0, HAD -> 1, HAD_fast -> 2
                out = np.zeros((sel.size, 9), np.uint64) if with_outputs else None
                (ob, os_), (cb, cs) = base(int(po), w), base(int(pr), w)
                self.keep += [st, out]
                self.stage_groups.append((sel, out))
                jobs.append(RecJob(2, 0, int(w), int(h), 0, 0, 0, sel.size, ob, cb, os_, cs, st.ctypes.data, None, out.ctypes.data if out is not None else None, None))
        self.arr = (RecJob * max(1, len(jobs)))(*jobs)
        self.n = len(jobs)

    def run(self, threads, passes):
        return self.L.vvref_run_recorded_mt(self.arr, self.n, self.wl.bit_depth, threads, passes)


def host_cpu_info():
    info = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_" + os.path.basename(p)] = open(p).read().strip()
        except OSError:
            pass
    return info


def usable_cores(info):
    n = info["affinity"]
    q = info.get("cgroup_cpu.max", "")
    try:
        a, b = q.split()
        if a != "max":
            n = min(n, max(1, int(int(a) / int(b))))
    except Exception:
        pass
    return n


def cpu_baseline(workloads, passes=5):
    """the reference's own x86-SIMD (AVX2) entries over the SAME recorded lists on the host cores: per layer one warm pass + `passes` timed passes on threads pinned to distinct
    CPUs; median per layer, GOP-weighted pictures/s; the spread of the passes is reported"""
    from oracle import oracle as O
    info = host_cpu_info()
    if not O.RefLib.available():
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref (the compiled reference) is not built", "host": info}
    cores = usable_cores(info)
    os.environ["VVREF_PIN"] = "1"
    per_layer, spread, t_all = {}, {}, time.perf_counter()
    for layer, wl in workloads.items():
        J = ReferenceJobs(wl, with_outputs=False)
        ts = sorted(J.run(cores, 1) for _ in range(passes))
        per_layer[layer] = ts[len(ts) // 2]
        spread[layer] = (ts[0], ts[-1])
        del J
    tot_w = sum(GOP_WEIGHT[l] for l in per_layer)
    sec = lambda pick: sum(GOP_WEIGHT[l] * pick(l) for l in per_layer) / tot_w
    return {"value": 1.0 / sec(lambda l: per_layer[l]), "unit": "frames/s", "cores": cores, "kind": "reference", "host": info, "passes": passes, "threads_pinned": True,
            "value_fastest_passes": 1.0 / sec(lambda l: spread[l][0]), "value_slowest_passes": 1.0 / sec(lambda l: spread[l][1]),
            "seconds_per_picture_by_layer": {str(l): round(v, 4) for l, v in per_layer.items()},
            "seconds_per_picture_min_max_by_layer": {str(l): [round(a, 4), round(b, 4)] for l, (a, b) in spread.items()},
            "sample": "median of %d full passes (after a warm-up pass) over every recorded list of one picture per temporal layer — integer SAD candidates, sub-pel refinement stages "
                      "(one first pass per horizontal position like xPatternRefinement, then second pass + Hadamard per evaluated position), merge / AMVP / intra / SSE table "
                      "calls, masked SADs, the fused TU pipeline's twin — through the reference's x86-SIMD (AVX2) entries on %d std::threads pinned to distinct CPUs, pulling chunks from "
                      "one atomic counter; GOP-weighted over the layers; %.1f s wall in total; DMVR lists not included" % (passes, cores, time.perf_counter() - t_all)}


def parity_check(workloads):
    """(a) device vs the values the real encoder computed while the lists were recorded; (b) device TU results vs the reference's x86-SIMD entries on the same lists"""
    from oracle import oracle as O
    from vvenc_amd.hotpath import STATS_DTYPE
    res = {"status": None, "mismatches": 0, "checked": {}, "against": "the costs the reference encoder itself computed when the lists were recorded (integer SAD, sub-pel Hadamard, "
           "table calls, DMVR vectors + costs)"}
    tot = {}
    for layer, wl in workloads.items():
        wl.run()
        for k, (n, bad) in wl.check_against_recording().items():
            a = tot.setdefault(k, [0, 0])
            a[0] += n
            a[1] += bad
    for k, (n, bad) in tot.items():
        res["checked"][k] = n
        res["mismatches"] += bad
    if O.RefLib.available():
        n_tu = bad_tu = 0
        cores = usable_cores(host_cpu_info())
        for layer, wl in workloads.items():
            J = ReferenceJobs(wl, with_outputs=True)
            J.arr = (RecJob * max(1, len(wl.tu_groups)))(*[j for j in J.arr[:J.n] if j.kind == 1])      # TU jobs only (the rest is checked against the recording)
            J.n = len(wl.tu_groups)
            J.run(cores, 1)
            torch.cuda.synchronize()
            for g, (sse, st4) in zip(wl.tu_groups, J.tu_outs):
                st = g["stats"].cpu().numpy().view(STATS_DTYPE).reshape(-1)
                lv = g["level"].view(g["n"], -1).to(torch.int64)
                idx = torch.arange(1, lv.shape[1] + 1, device=lv.device, dtype=torch.int64)
                cs = ((lv * idx).sum(1) & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)
                bad = (st["sse"] != sse) | (st["abs_sum"] != st4[:, 0]) | (st["need_rdoq"] != st4[:, 2]) | (cs != st4[:, 3].view(np.uint32))
                has = st4[:, 0] != 0
                bad |= has & (st["last_scan_pos"] != st4[:, 1])          # (the last position is defined when a level is non-zero)
                bad_tu += int(bad.sum())
                n_tu += g["n"]
            del J
        res["checked"]["tus_sse_abssum_last_needrdoq_levels"] = n_tu
        res["mismatches"] += bad_tu
        res["against"] += "; TU outputs against the reference's x86-SIMD entries (oracle/_ref) on the same residuals"
    res["status"] = "bit-exact" if res["mismatches"] == 0 else "MISMATCH"
    return res


# ---------------------------------------------------------------------------------------------------------------------- profiling passes
def run_inner_profile(width, height, prof, steps, tag, mode="--inner"):
    outdir = os.path.join("/tmp", "vvhip_prof_%d_%s" % (os.getpid(), tag))
    shutil.rmtree(outdir, ignore_errors=True)
    cmd = ["rocprofv3"] + prof + ["-d", outdir, "--", sys.executable, os.path.abspath(__file__), mode, "--steps", str(steps), "--warmup", "0",
                                  "--width", str(width), "--height", str(height)]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if r.returncode != 0:
        raise RuntimeError("rocprofv3 %s pass: rc %d: %s" % (tag, r.returncode, r.stdout[-400:]))
    dbs = sorted(glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True), key=os.path.getmtime)
    if not dbs:
        raise RuntimeError("rocprofv3 %s pass left no database" % tag)
    return dbs[-1], outdir


def class_of_kernel(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for cls, sub in KERNEL_NAMES.items():
        if n.startswith(sub):
            return cls
    return None


ALL_COUNTERS = (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib"), ("TCP_TOTAL_CACHE_ACCESSES_sum", "l1_accesses"), ("SQ_INSTS_VALU", "valu_insts"), ("TCC_HIT_sum", "l2_hits"), ("TCC_MISS_sum", "l2_misses"))


def live_profile(width, height, counters=ALL_COUNTERS):
    """kernel trace + one --pmc pass per counter over a short inner run (32 steps = one GOP cycle, launches serialized): per kernel class the average duration and the RAW
    counters per launch (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; the calibrated byte factors are applied by the caller)"""
    import profile_round as P
    out, dirs = {}, []
    db, d = run_inner_profile(width, height, ["--kernel-trace", "--stats"], 32, "trace")
    dirs.append(d)
    rows = P.kernel_table(db)
    tot = sum(r[2] for r in rows) or 1
    out["kernel_trace"] = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --inner --steps 32 --width %d --height %d (one GOP cycle of recorded pictures, launches serialized on one stream)" % (width, height),
                           "kernels": [{"name": k.replace("(anonymous namespace)::", "")[:90], "calls": n, "avg_us": round(av / 1e3, 2), "total_us": round(s / 1e3, 1), "pct": round(100.0 * s / tot, 1)}
                                       for k, n, s, av, mn, mx in rows[:12]]}
    cls = {}
    for k, n, s, av, mn, mx in rows:
        c = class_of_kernel(k)
        if c:
            e = cls.setdefault(c, {"launches": 0, "total_ns": 0.0})
            e["launches"] += n
            e["total_ns"] += s
    for counter, key in counters:
        try:
            db, d = run_inner_profile(width, height, ["--pmc", counter], 32, counter)
            dirs.append(d)
            for k, c, n, s, av in P.counter_table(db):
                kc = class_of_kernel(k)
                if kc:
                    e = cls.setdefault(kc, {})
                    e[key] = e.get(key, 0.0) + s
                    e["n_" + key] = e.get("n_" + key, 0) + n
        except Exception as ex:
            out.setdefault("pmc_errors", []).append("%s: %s" % (counter, str(ex)[:160]))
    out["per_class"] = cls
    for d in dirs:
        shutil.rmtree(d, ignore_errors=True)
    return out


# which calibration pattern (tools/calib/fetch_calib.hip) a kernel class's reads look like: per-lane 16-byte row gathers out of picture planes, or streams of compact blocks
FETCH_PATTERN = {"ME_stage": "rows16", "ME_int": "rows16", "ME_item": "rows16", "DMVR": "rows16", "TU": "stream16"}


def counter_calibration():
    """FETCH_SIZE / WRITE_SIZE factors measured on THIS GPU against known byte counts (tools/calib_fetch.py; two short rocprofv3 passes).  Falls back to the guide's figure for
    wide coalesced reads (x2) and x1 for writes, and says so."""
    try:
        import calib_fetch
        c = calib_fetch.calibrate()
        f = {k: v["factor"] for k, v in c["patterns"].items() if v.get("factor")}
        if not {"rows16", "stream16", "store8"} <= set(f):
            raise RuntimeError("patterns missing: %s" % sorted(f))
        return {"measured": True, "factors": {k: round(v, 4) for k, v in f.items()}, "how": "tools/calib_fetch.py: every byte of a 512 MiB buffer read / written once per pattern, "
                "factor = known bytes / (counter x 1024)", "pattern_of_class": FETCH_PATTERN}
    except Exception as e:
        return {"measured": False, "factors": {"rows16": 2.0, "stream16": 2.0, "store8": 1.0}, "how": "calibration failed (%s): MI355X_MICROARCH.md's x2 for wide coalesced reads, x1 for writes" % str(e)[:120],
                "pattern_of_class": FETCH_PATTERN}


def roofline_objects(kern, live, calib, unique_by_class, profile_md=None):
    """-> (roofline of the dominant kernel class, the same positions for every class) from the per-class raw counters"""
    dom = max(kern, key=lambda k: kern[k]["avg_ms_per_picture"])
    roof = {"kernel": KERNEL_NAMES[dom], "class": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": kern[dom]["avg_ms_per_picture"],
            "nominal_alg_GBps": kern[dom]["nominal_alg_GBps"], "alg_bytes_per_launch": kern[dom]["alg_bytes_per_picture"],
            "basis": "achieved = ALGORITHMIC bytes of one launch (SURVEY 8d: 4 w h bytes per scored position, DESIGN 5) / the launch's average duration in the rocprofv3 kernel trace; "
                     "frac = achieved / 8 TB/s.  traffic = bytes that crossed the L2's memory side per launch (rocprofv3 --pmc FETCH_SIZE x the factor calibrated on this GPU for the "
                     "kernel's access pattern + WRITE_SIZE, separate passes over the inner run of this command); FETCH_SIZE counts the L2's fabric read requests, hits in the 256 MB "
                     "Infinity Cache INCLUDED, so frac_physical = traffic / duration / 8 TB/s is an UPPER bound of the HBM fraction.  The kernel stages its windows in LDS and the XCD-band "
                     "schedule keeps a band of the planes in each L2: traffic is ~ the unique bytes (traffic_over_unique), an order of magnitude below the algorithmic bytes, and what "
                     "binds the launch is VALU issue (binding_resource), not HBM"}
    allk = {}
    if not live or not live.get("per_class"):
        roof.update({"bound": "hbm", "traffic": None, "achieved": kern[dom]["nominal_alg_GBps"], "frac": kern[dom]["nominal_alg_GBps"] / HBM_PEAK_GBS, "frac_physical": None,
                     "note": "no live PMC pass (rocprofv3 absent or --no-profile): HIP-event duration per picture instead of the trace's per-launch average, no counter traffic"})
        return roof, allk
    fac = calib["factors"]
    for k, c in live["per_class"].items():
        n = max(1, c.get("launches", 1))
        t_k = (c["total_ns"] / n) * 1e-9 if c.get("total_ns") else None
        if not t_k:
            continue
        pk = lambda key: (c.get(key, 0.0) / max(1, c.get("n_" + key, 0))) if c.get("n_" + key) else None
        f_, w_, l_, v_, h_, m_ = pk("fetch_kib"), pk("write_kib"), pk("l1_accesses"), pk("valu_insts"), pk("l2_hits"), pk("l2_misses")
        ff = fac.get(FETCH_PATTERN.get(k, "rows16"), 2.0)
        traffic = (f_ * 1024.0 * ff + (w_ or 0.0) * 1024.0 * fac.get("store8", 1.0)) if f_ is not None else None
        lpp = n / 32.0                                        # launches per picture (the inner run is one GOP cycle of 32 pictures)
        uniq = unique_by_class.get(k)
        allk[k] = {"kernel": KERNEL_NAMES.get(k, k), "launches_per_picture": round(lpp, 2), "avg_launch_us": round(t_k * 1e6, 2),
                   "fabric_traffic_MB_per_launch": round(traffic / 1e6, 2) if traffic is not None else None,
                   "fabric_traffic_bounds_MB": [round((f_ * 1024.0 + (w_ or 0) * 1024.0) / 1e6, 2), round((f_ * 2048.0 + (w_ or 0) * 1024.0) / 1e6, 2)] if f_ is not None else None,
                   "alg_MB_per_launch": round(kern[k]["alg_bytes_per_picture"] / max(1e-9, lpp) / 1e6, 2) if k in kern else None,
                   "alg_frac_of_hbm_peak": round(kern[k]["alg_bytes_per_picture"] / max(1e-9, lpp) / t_k / 1e9 / HBM_PEAK_GBS, 4) if k in kern else None,
                   "fetch_factor": round(ff, 3), "fabric_frac_of_hbm_peak": round(traffic / t_k / 1e9 / HBM_PEAK_GBS, 4) if traffic is not None else None,
                   "unique_MB_per_picture": round(uniq / 1e6, 2) if uniq else None,
                   "traffic_over_unique": round(traffic * lpp / uniq, 2) if traffic is not None and uniq else None,
                   "l2_hit_rate": round(h_ / (h_ + m_), 3) if h_ is not None and m_ is not None and h_ + m_ > 0 else None,
                   "l1_access_frac": round(l_ / (N_CU * CLOCK_GHZ * 1e9 * t_k), 4) if l_ else None,
                   "valu_issue_frac": round(v_ * 4.0 / (N_SIMD * CLOCK_GHZ * 1e9 * t_k), 4) if v_ else None}
    if dom in allk:
        d = allk[dom]
        c = live["per_class"][dom]
        n = max(1, c.get("launches", 1))
        t_s = (c["total_ns"] / n) * 1e-9
        traffic = d["fabric_traffic_MB_per_launch"] * 1e6 if d["fabric_traffic_MB_per_launch"] is not None else None
        roof.update({"traffic": traffic, "avg_launch_ms": t_s * 1e3, "launches_per_picture": d["launches_per_picture"], "ms_per_picture": kern[dom]["avg_ms_per_picture"],
                     "alg_bytes_per_launch": kern[dom]["alg_bytes_per_picture"] / max(1e-9, n / 32.0),
                     "avg_launch_ms_measured": "rocprofv3 kernel trace of the inner run (launches serialized), averaged over the class's launches of one GOP cycle",
                     "achieved": kern[dom]["alg_bytes_per_picture"] / max(1e-9, n / 32.0) / t_s / 1e9,
                     "frac": kern[dom]["alg_bytes_per_picture"] / max(1e-9, n / 32.0) / t_s / 1e9 / HBM_PEAK_GBS,
                     "achieved_physical": (traffic / t_s / 1e9) if traffic else None, "frac_physical": (traffic / t_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "frac_note": "frac = algorithmic bytes / time / 8 TB/s (the task's definition; rounds 2-3 reported the counter-based figure here, now frac_physical: 0.27 in round 3 with 95 MB of "
                                  "fabric traffic per launch, ~0.05 now with ~15 MB for a shorter launch — it FALLS when re-reads are removed)",
                     "fetch_factor": d["fetch_factor"], "traffic_bounds_MB": d["fabric_traffic_bounds_MB"],
                     "unique_bytes_per_picture": unique_by_class.get(dom), "traffic_over_unique": d["traffic_over_unique"], "l2_hit_rate": d["l2_hit_rate"],
                     "traffic_over_alg_bytes": (traffic * (n / 32.0) / kern[dom]["alg_bytes_per_picture"]) if traffic and kern[dom]["alg_bytes_per_picture"] else None,
                     "l1_access_frac": d["l1_access_frac"], "valu_issue_frac": d["valu_issue_frac"]})
        fr = {"hbm": roof["frac_physical"] or 0.0, "l1_access": roof["l1_access_frac"] or 0.0, "valu": roof["valu_issue_frac"] or 0.0}
        order = sorted(fr, key=lambda k: -fr[k])
        roof["bound"] = "hbm"                                                        # the ceiling `peak` / `frac` are quoted against (the task's roofline object: hbm | mfma; no MFMA on this path)
        roof["bound_physical"] = "+".join(k for k in order if fr[k] >= 0.6 * fr[order[0]] and fr[k] > 0) or "hbm"
        roof["binding_resource"], roof["binding_frac"] = order[0], fr[order[0]]      # the resource closest to its ceiling and how close
        roof["valu_rate_note"] = "valu_issue_frac counts 4 cycles at %.1f GHz (1.67 ns) per wave instruction; measured sustained issue on this GPU is 1.8 ns per instruction and SIMD for the VOP3 / DPP / packed forms these kernels use (profiles/r04_valu_rate.log): the fraction of the ATTAINABLE issue rate is ~1.08 x valu_issue_frac" % CLOCK_GHZ
        roof["limiter"] = "fractions of the launch time: fabric traffic %.3f of the HBM peak, L1 (TCP) access slots %.3f (one access per 64-byte granule and instruction, %d CUs x %.1f GHz), VALU issue slots %.3f " \
                          "(wave instructions x 4 cycles / %d SIMDs); the rest is latency the resident waves do not cover" % (fr["hbm"], fr["l1_access"], N_CU, CLOCK_GHZ, fr["valu"], N_SIMD)
    else:
        roof.update({"bound": "hbm", "traffic": None, "achieved": kern[dom]["nominal_alg_GBps"], "frac": kern[dom]["nominal_alg_GBps"] / HBM_PEAK_GBS, "frac_physical": None, "note": "the PMC passes did not see the dominant kernel"})
    if profile_md:
        try:
            with open(profile_md, "w") as f:
                f.write("# rocprofv3 summary of `python bench.py` (written by bench.py --profile-md from its own passes)\n\n")
                f.write("Inner run: `%s`\n\n" % live["kernel_trace"]["command"])
                f.write("## rocprofv3 --kernel-trace --stats\n\n| kernel | calls | avg us | total us | % |\n|---|---|---|---|---|\n")
                for r in live["kernel_trace"]["kernels"]:
                    f.write("| `%s` | %d | %.2f | %.1f | %.1f |\n" % (r["name"], r["calls"], r["avg_us"], r["total_us"], r["pct"]))
                f.write("\n## rocprofv3 --pmc, one pass per counter; per launch.  Fabric traffic = FETCH_SIZE x 1024 x the calibrated factor of the class's access pattern + WRITE_SIZE x 1024 "
                        "(calibration: %s; factors %s).  FETCH_SIZE includes Infinity-Cache hits.\n\n"
                        "| class | kernel | launches per picture | avg launch us | fabric traffic MB | [x1, x2] bounds MB | frac of 8 TB/s | unique MB per picture | traffic / unique | L2 hit rate | L1 access frac | VALU issue frac |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n"
                        % (calib["how"], json.dumps(calib["factors"])))
                for k, r in allk.items():
                    f.write("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |\n" % (k, r["kernel"], r["launches_per_picture"], r["avg_launch_us"], r["fabric_traffic_MB_per_launch"], r["fabric_traffic_bounds_MB"],
                                                                                                  r["fabric_frac_of_hbm_peak"], r["unique_MB_per_picture"], r["traffic_over_unique"], r["l2_hit_rate"], r["l1_access_frac"], r["valu_issue_frac"]))
                f.write("\nHIP-event time per picture (GOP-weighted, launches serialized): " + ", ".join("%s %.1f us" % (k, kern[k]["avg_ms_per_picture"] * 1e3) for k in kern) + "\n")
        except Exception as e:
            roof["profile_md_error"] = str(e)[:200]
    return roof, allk


def mctf_profile(args):
    """the MCTF motion estimation under rocprofv3 (7 calls of one 1080p picture against 4 references): per picture the time of the parallel candidate scoring, of the sequential
    sweep (= the critical path of phase B: one workgroup per reference) and the VALU issue fraction of the scoring kernel"""
    import profile_round as P
    out, dirs = {}, []
    db, d = run_inner_profile(args.width, args.height, ["--kernel-trace", "--stats"], 1, "mctf_trace", "--inner-mctf")
    dirs.append(d)
    calls = 7.0
    t = {}
    for k, n, sm, av, mn, mx in P.kernel_table(db):
        name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith("me") or name.startswith("subsample") or name.startswith("extend") or name.startswith("initMvs"):
            t[name] = t.get(name, 0.0) + sm
    out["us_per_picture_by_kernel"] = {k: round(v / 1e3 / calls, 1) for k, v in sorted(t.items(), key=lambda kv: -kv[1])}
    out["phase_a_us"] = round(t.get("meSearchKernel", 0.0) / 1e3 / calls, 1)
    out["critical_path_us"] = round((t.get("meDiagKernel", 0.0) + t.get("meWavefrontKernel", 0.0)) / 1e3 / calls, 1)
    out["critical_path_note"] = "the anti-diagonal sweep of phase B (MCTF.cpp:1289-1306): one workgroup per reference, cols + rows dependent steps per level; everything else of the call is parallel over blocks"
    try:
        db, d = run_inner_profile(args.width, args.height, ["--pmc", "SQ_INSTS_VALU"], 1, "mctf_valu", "--inner-mctf")
        dirs.append(d)
        for k, c, n, sm, av in P.counter_table(db):
            if "meSearchKernel" in k and t.get("meSearchKernel"):
                out["phase_a_valu_issue_frac"] = round(sm * 4.0 / (N_SIMD * CLOCK_GHZ * 1e9 * t["meSearchKernel"] * 1e-9), 3)
    except Exception as ex:
        out["pmc_error"] = str(ex)[:160]
    for d in dirs:
        shutil.rmtree(d, ignore_errors=True)
    return out


# ---------------------------------------------------------------------------------------------------------------------- device side
class Mctf1080:
    """what tools/bench_synthetic.mctf_stage needs from a workload: one picture pair of the config-2 generator"""

    def __init__(self, width, height, bit_depth=10):
        from vvenc_amd.workload import synth_frame_pair
        self.width, self.height, self.bit_depth = width, height, bit_depth
        self.cur_np, self.ref_np = synth_frame_pair(width, height, 1080 if width == 1920 else 2160, bit_depth)


def e2e_encoder(width, height, frames, threads, pairs):
    """the real reference encoder end to end (SURVEY 8d metric): CPU kernels vs --SIMD=HIP, same clip, same threads; subprocesses (the SIMD level is process-wide)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_fps
    import e2e_util
    if not (os.path.exists(e2e_util.REF_SO) and os.path.exists(e2e_util.REF_HIP_SO)):
        return {"skipped": "the compiled reference encoder (oracle/_ref) and the encoder with the binding (bindings/vvenc/_build) are not both built"}
    prod = e2e_production_mask()
    e2e_fps.run(dict(w=width, h=height, frames=frames, threads=threads, mask=0), timeout=900)          # discarded run: clip cache, page cache, clocks
    runs = [e2e_fps.run(dict(w=width, h=height, frames=frames, threads=threads, mask=m), timeout=900) for m in (0, prod) * pairs]
    med = lambda v: sorted(v)[len(v) // 2]
    cpu = med([r["fps"] for r in runs if r["mask"] == 0])
    hip = med([r["fps"] for r in runs if r["mask"] == prod])
    return {"clip": "%dx%d 10-bit synthetic (config-2 generator), %d frames, preset faster, QP 32" % (width, height, frames), "threads": threads,
            "cpu_fps": round(cpu, 2), "hip_fps": round(hip, 2), "speedup": round(hip / cpu, 3), "runs_fps": [round(r["fps"], 2) for r in runs],
            "runs_order": "cpu, hip alternating, %d pairs after one discarded run; cpu_fps / hip_fps are medians" % pairs,
            "bitstreams_identical": len({r["md5"] for r in runs}) == 1, "md5": runs[0]["md5"], "hook_mask": prod,
            "device_stages": "MCTF motion estimation (all references of a picture per call) + bilateral filter, ALF statistics of whole pictures (--SIMD=HIP production mask)",
            "pcie_MB_per_picture": runs[1].get("pcie_MB_per_picture")}


def e2e_production_mask():
    return 16 + 128 + 8192


# ---------------------------------------------------------------------------------------------------------------------- N encoder instances (N > 1)
def e2e_instances(rank, local_rank, world, width=1920, height=1080, frames=65):
    """The BASELINE metric at N GPUs: N encoder instances, one per rank / GPU, each with its share of the host cores, over GOP chunks of ONE sequence (chunk r = frames
    r * 33 .. r * 33 + 32 of the config-2 generator's endless clip; every chunk starts with its own intra picture like a closed-GOP segment — how a sequence is split for
    chunk-parallel encoding; inside one encoder the reference's own GOP parallelism is EncGOP.cpp:1647-1651 / vvencCfg.cpp:2188-2199).  All instances run at the same
    time, first with CPU kernels, then with --SIMD=HIP on their GPU: aggregate fps = N * frames / the slowest instance's ENCODE time (the encoder's own clock around its
    encode loop: process start, `import torch` and context creation of an instance are not part of a sequence's frame rate; the wall-clock figure is reported next to it);
    per-chunk md5 CPU == HIP."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_fps
    import e2e_util
    if not (os.path.exists(e2e_util.REF_SO) and os.path.exists(e2e_util.REF_HIP_SO)):
        return {"skipped": "the compiled reference encoder (oracle/_ref) and the encoder with the binding (bindings/vvenc/_build) are not both built"} if rank == 0 else None
    threads = max(1, usable_cores(host_cpu_info()) // world)
    ndev = max(1, torch.cuda.device_count())
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = str(local_rank % ndev)                     # the instance sees ONE device: its rank's GPU
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VVHIP_SHARE_DEVICE", "VVHIP_DIST_BACKEND"):
        env.pop(k, None)
    prod = e2e_production_mask()
    cfg = dict(w=width, h=height, frames=frames, first=rank * frames, threads=threads)
    e2e_fps.synth_clip_chunk(width, height, rank * frames, frames)          # (the chunk's clip is made before the clock starts; the instances load it from the cache)
    res = {}
    for name, mask in (("warm", 0), ("cpu", 0), ("hip", prod)):
        sharding.barrier()
        t0 = time.perf_counter()
        try:
            r = e2e_fps.run(dict(cfg, mask=mask), timeout=1200, env=env)
        except Exception as e:
            r = {"md5": "error: " + str(e)[-200:], "fps": 0.0}
        dt = sharding.max_over_ranks(time.perf_counter() - t0, device="cuda")          # (device tensors: RCCL has no host reductions)
        enc = sharding.max_over_ranks(float(r.get("secs") or 1e9), device="cuda")        # the slowest instance's encode time
        res[name] = (r, dt, enc)
    same = 1.0 if res["cpu"][0]["md5"] == res["hip"][0]["md5"] and not res["cpu"][0]["md5"].startswith("error") else 0.0
    all_same = -sharding.max_over_ranks(-same, device="cuda")                # min over ranks
    gathered = [None] * world
    if dist.is_initialized():
        dist.all_gather_object(gathered, {"rank": rank, "chunk_first_frame": rank * frames, "cpu_fps": round(res["cpu"][0]["fps"], 2), "hip_fps": round(res["hip"][0]["fps"], 2), "md5": res["cpu"][0]["md5"][:12],
                                          "md5_hip": res["hip"][0]["md5"][:12]})
    if rank != 0:
        return None
    cpu_fps, hip_fps = world * frames / res["cpu"][2], world * frames / res["hip"][2]
    cpu_wall, hip_wall = world * frames / res["cpu"][1], world * frames / res["hip"][1]
    return {"instances": world, "frames_per_chunk": frames, "threads_per_instance": threads, "clip": "%dx%d 10-bit, chunk r = frames %d r .. %d r + %d of one endless config-2 sequence, preset faster" % (width, height, frames, frames, frames - 1),
            "cpu_fps_aggregate": round(cpu_fps, 2), "hip_fps_aggregate": round(hip_fps, 2), "speedup": round(hip_fps / cpu_fps, 3) if cpu_fps else None,
            "cpu_fps_aggregate_wall": round(cpu_wall, 2), "hip_fps_aggregate_wall": round(hip_wall, 2),
            "chunk_bitstreams_identical": bool(all_same == 1.0), "hook_mask": prod, "per_instance": gathered,
            "timing": "aggregate = N x frames / the slowest instance's encode time (all instances start at one barrier and run concurrently); _wall: from the barrier to the slowest instance's exit, i.e. "
                      "with process start, `import torch` and HIP context creation of the instance (≈1.5 s, a one-off per sequence, not per chunk of a long one); one discarded CPU run first",
            "note": "one encoder process is host-bound (DESIGN 7): N-GPU frames/s in the sense of the metric is N instances; it scales with the host cores each instance gets, the GPUs are never the limit"}


# ---------------------------------------------------------------------------------------------------------------------- BASELINE configs[3]'s lists (preset medium)
def replay_medium_4k(hp, streams=5):
    """one 3840x2160 picture of a preset-MEDIUM encode (BASELINE configs[3]'s geometry and preset: CTU 128, multi-type tree -> rectangular blocks 4..128, GEO masked SADs, two
    references per list) through the same batched path: nothing of the recording left out, every output against the encoder's own values, time per picture"""
    from vvenc_amd.replay import RecordedWorkload
    pics, info = prepare_recordings(3840, 2160, 9, [4], tag="medium", threads=16)
    wl = RecordedWorkload(hp, pics[4])
    lanes = [hp.fork(torch.cuda.Stream()) for _ in range(streams)]
    wl.bind_lanes(lanes)
    for _ in range(3):
        wl.run_lanes()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        wl.run_lanes()
    torch.cuda.synchronize()
    ms = 1000.0 * (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        wl.run()
    torch.cuda.synchronize()
    ms1 = 1000.0 * (time.perf_counter() - t0) / n
    par = parity_check({5: wl})
    me = wl.pic.me
    shapes = sorted({(int(w), int(h)) for w, h in zip(me["w"].tolist(), me["h"].tolist())})
    out = {"clip": "3840x2160 10-bit synthetic config-2 clip, 9 frames, preset medium, picture POC 4 (TL5)", "ms_per_picture": round(ms, 4), "pictures_per_s": round(1000.0 / ms, 1),
           "ms_per_picture_single_stream": round(ms1, 4), "sample_pairs": int(wl.pic.sample_pairs), "sample_pairs_per_1p5WH": round(wl.pic.sample_pairs / (1.5 * 3840 * 2160), 1),
           "recorded_calls_outside_the_lists": wl.dropped, "nothing_dropped": bool(wl.nothing_dropped), "parity": par,
           "work": {"me_calls": int(me.size), "me_block_shapes": ["%dx%d" % s_ for s_ in shapes], "integer_candidates": int(wl.plan_cands.size), "subpel_stages": int(wl.stage_jobs.size),
                    "table_calls": int(wl.items.size), "masked_sad_calls": int(wl.mask_items.size), "tus": int(sum(g["n"] for g in wl.tu_groups)),
                    "tu_shapes": sorted({"%dx%d" % (g["w"], g["h"]) for g in wl.tu_groups}), "dmvr_subblocks": int(sum(g["n"] for g in wl.dmvr_groups)), "plan": wl.me_info},
           "recording": info}
    try:
        cb = cpu_baseline({5: wl}, passes=3)
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "passes", "threads_pinned", "seconds_per_picture_by_layer") if k in cb}
    except Exception as e:
        out["cpu_baseline"] = {"error": str(e)[:200]}
    for c in lanes:
        c.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------- one replay pass (a resolution)
def replay_pass(args, hp, rank, world, width, height, steps, warmup):
    """records (or loads) the layer pictures of the width x height encode, puts them on the device, times `steps` steps and collects the per-kernel times.
    -> (core fields of the JSON line, workloads, kern)"""
    from vvenc_amd.replay import RecordedWorkload
    rec_info = {}
    if rank == 0:
        pics, rec_info = prepare_recordings(width, height, 65, sorted(LAYER_POCS.values()))
    sharding.barrier()
    if rank != 0:
        pics, _ = prepare_recordings(width, height, 65, sorted(LAYER_POCS.values()))
    workloads = {layer: RecordedWorkload(hp, pics[poc]) for layer, poc in LAYER_POCS.items()}
    for l, wl in workloads.items():
        if not wl.nothing_dropped:
            raise RuntimeError("layer %d: recorded calls outside the lists: %s" % (l, wl.dropped))

    # ---- lanes: the refinement stages, integer windows and table calls of the plan, the TU lists and the DMVR lists of a picture are independent work: five HIP streams
    lanes, streams = None, []
    if args.streams > 1:
        streams = [torch.cuda.Stream() for _ in range(5 if args.streams >= 5 else 3)]
        lanes = [hp.fork(s) for s in streams]
        for wl in workloads.values():
            wl.bind_lanes(lanes)

    # ---- the reference-picture exchange of the sharded sequence (N > 1): ring of two reconstructed pictures (luma + 2 chroma planes with margins)
    ex = None
    if world > 1:
        m = 80
        shp = (height + 2 * m, ((width + 2 * m + 7) // 8) * 8)
        ex = sharding.PictureExchange([shp, (shp[0] // 2, shp[1] // 2), (shp[0] // 2, shp[1] // 2)], slots=2, device=hp.device)
        ex.publish(0, 0)
    step_no = [0]                                        # this rank's step count k; it replays position step_of_rank( k ) of the cycle
    ex_count = [0]

    def step():
        k = step_no[0]
        s = step_of_rank(k, rank, world)
        if ex is not None:
            if k % args.exchange_every == 0:
                e = k // args.exchange_every
                ex.publish(e + 1, (e + 1) % world, readers=streams if lanes else ())      # the next reference picture is in flight while this picture's launches run
                ex_count[0] += 1
                ex.wait(e, streams if lanes else None)
        wl = workloads[layer_of_step(s)]
        if lanes:
            wl.run_lanes()
        else:
            wl.run()
        step_no[0] = k + 1

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # untimed settling (a freshly acquired box stalls once, a few milliseconds into its first multi-queue phase): whole GOP cycles until two agree
    if world == 1:
        prev = None
        for _ in range(40):
            tb = time.perf_counter()
            for _ in range(32):
                step()
            torch.cuda.synchronize()
            cur = time.perf_counter() - tb
            if prev is not None and abs(cur - prev) < 0.15 * min(cur, prev):
                break
            prev = cur
    else:
        for _ in range(64):
            step()
        torch.cuda.synchronize()

    # ---- per-kernel durations: every layer's launches serialized on one stream, HIP events around each kernel (inside the library for the plan's kernels); outside the timed region
    per_layer = {}
    for layer, wl in workloads.items():
        hp.me_plan_set_timing(wl.plan, True)
        acc = {"ME_stage": 0.0, "ME_int": 0.0, "ME_item": 0.0, "TU": 0.0, "DMVR": 0.0}
        reps = 8
        for it in range(reps + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            wl.run_me()
            e[0].record()
            wl.run_tu()
            e[1].record()
            wl.run_dmvr()
            e[2].record()
            torch.cuda.synchronize()
            t = hp.me_plan_last_times(wl.plan)
            if it == 0:
                continue
            acc["ME_stage"] += t[0]
            acc["ME_int"] += t[1] + t[2]
            acc["ME_item"] += t[3]
            acc["TU"] += e[0].elapsed_time(e[1])
            acc["DMVR"] += e[1].elapsed_time(e[2])
        hp.me_plan_set_timing(wl.plan, False)
        per_layer[layer] = {k: v / reps for k, v in acc.items()}

    # ---- THE timed region: exactly `steps` steps between barrier + synchronize on both sides, max over ranks
    step_no[0] = 0
    sharding.barrier()
    torch.cuda.synchronize()
    enq = [0.0] * (steps + 1)
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        enq[i + 1] = time.perf_counter()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    enq[0] = t0
    enq_us = sorted(1e6 * (b - a) for a, b in zip(enq[:-1], enq[1:]))
    dt = sharding.max_over_ranks(dt_local, device="cuda")

    # extras (not `value`): the same K steps serialized on one stream; per layer, the multi-stream time of one picture (-> the GOP-weighted rate); N > 1: without the picture exchange
    serial = None
    if lanes:
        t1 = time.perf_counter()
        for i in range(steps):
            workloads[layer_of_step(step_of_rank(i, rank, world))].run()
        torch.cuda.synchronize()
        dts = sharding.max_over_ranks(time.perf_counter() - t1, device="cuda")
        serial = {"value": steps * world / dts, "unit": "frames/s", "ms_per_step": 1000.0 * dts / steps, "note": "same pictures, every launch on ONE stream (no picture exchange); not the headline value"}
    layer_ms = {}
    if world == 1:
        for l, wl in workloads.items():
            n = 12 if l else 4
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                wl.run_lanes() if lanes else wl.run()
            torch.cuda.synchronize()
            layer_ms[l] = 1000.0 * (time.perf_counter() - t1) / n
    no_exchange = None
    if ex is not None:
        sharding.barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            wl = workloads[layer_of_step(step_of_rank(i, rank, world))]
            wl.run_lanes() if lanes else wl.run()
        torch.cuda.synchronize()
        dtn = sharding.max_over_ranks(time.perf_counter() - t1, device="cuda")
        no_exchange = {"value": steps * world / dtn, "unit": "frames/s", "ms_per_step": 1000.0 * dtn / steps, "note": "same steps without the reference-picture broadcast: kernel scaling alone"}
    if rank != 0:
        return None, workloads, None

    frames = steps * world
    pairs_by_layer = {l: workloads[l].pic.sample_pairs for l in workloads}
    wsum = float(sum(GOP_WEIGHT.values()))
    lay_seq = [layer_of_step(step_of_rank(i, rank, world)) for i in range(steps)]
    out = {
        "value": frames / dt, "unit": "frames/s", "steps": steps, "warmup": warmup, "ms_per_step": 1000.0 * dt / steps,
        "host_enqueue_us_per_step": {"p50": round(enq_us[len(enq_us) // 2], 1), "p90": round(enq_us[int(len(enq_us) * 0.9)], 1), "max": round(enq_us[-1], 1),
                                     "drain_ms_after_last_enqueue": round(1000.0 * (t0 + dt_local - enq[-1]), 3)},
        "config": {"workload": "work lists RECORDED from the reference encoder (bindings/vvenc recorder, hook bit 131072): %dx%d 10-bit synthetic config-2 clip, 65 frames, preset faster; one picture "
                               "per temporal layer (POC %s); step s replays the picture of layer STEP_LAYERS[s mod 32], a low-discrepancy interleaving of the GOP's 1 : 1 : 2 : 4 : 8 : 16 layer mix that starts at the key picture (bench.py); BASELINE configs[%d]"
                               % (width, height, ", ".join("%d = TL%d" % (p, l) for l, p in LAYER_POCS.items()), 1 if width == 1920 else 2),
                   "pictures_per_32_steps_by_layer": {str(l): GOP_WEIGHT[l] for l in GOP_WEIGHT},
                   "pictures_in_the_timed_steps_by_layer": {str(l): lay_seq.count(l) for l in range(6)},
                   "sample_pairs_per_frame": int(sum(GOP_WEIGHT[l] * pairs_by_layer[l] for l in pairs_by_layer) / wsum),
                   "sample_pairs_per_frame_by_layer": {str(l): int(v) for l, v in pairs_by_layer.items()},
                   "sample_pairs_note": "counted like the reference's own counter (CommonLib/SearchSpaceCounter.cpp:106-165 via RdCost.cpp:150-156: w x h per table call, DMVR excluded): "
                                        "%.1f x 1.5 W H GOP-weighted" % (sum(GOP_WEIGHT[l] * pairs_by_layer[l] for l in pairs_by_layer) / wsum / (1.5 * width * height)),
                   "coefficients_per_frame": int(sum(GOP_WEIGHT[l] * workloads[l].tu_coefficients for l in workloads) / wsum),
                   "work_per_layer": {str(l): {"me_calls": int(workloads[l].pic.me.size), "integer_candidates": int(workloads[l].plan_cands.size), "integer_positions_distinct": workloads[l].distinct_positions,
                                               "subpel_stages": int(workloads[l].stage_jobs.size), "subpel_positions": int(workloads[l].stage_evaluated.sum()), "table_calls": int(workloads[l].items.size), "tus": int(sum(g["n"] for g in workloads[l].tu_groups)),
                                               "dmvr_subblocks": int(sum(g["n"] for g in workloads[l].dmvr_groups)), "plan": workloads[l].me_info} for l in workloads},
                   "recorded_calls_outside_the_lists": 0,
                   "subpel_candidates_per_block": round(float(np.mean([workloads[l].stage_evaluated.sum() / max(1, workloads[l].pic.me.size) for l in workloads if workloads[l].pic.me.size])), 2),
                   "launches_per_frame": "motion-search plan (refinement stages per tap set + integer windows + table calls) + 1 TU launch (+ one per rectangular TU shape) + 0-1 DMVR launch per reference pair",
                   "schedule": "workgroups of every launch in XCD-band order: XCD x (workgroup index mod 8) takes the x-th contiguous eighth of each class in picture order ($VVHIP_ME_XCD_BAND=0: heaviest first)",
                   "hip_streams": len(lanes) if lanes else 1, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "recording": rec_info,
                   "sharding": "one picture per rank and step: rank r replays position k + 32 r / N of the cycle at its step k (N different pictures of one sequence at any time, the same layer mix per rank), no data-path collective"
                               + (", reconstructed picture (luma + chroma, %.1f MB) RCCL-broadcast from its owner every %d step(s) inside the timed region, overlapped with the launches"
                                  % (sum(p.numel() * 2 for p in ex.slots[0]) / 1e6, args.exchange_every) if ex is not None else "")},
    }
    if layer_ms:
        ms = sum(GOP_WEIGHT[l] * layer_ms[l] for l in layer_ms) / wsum
        out["gop_weighted"] = {"value": 1000.0 / ms, "unit": "frames/s", "ms_per_picture_by_layer": {str(l): round(v, 4) for l, v in layer_ms.items()},
                               "note": "per layer: the picture's launches on the five streams, back to back, 12 times (4 for the intra picture); weighted 1 : 1 : 2 : 4 : 8 : 16 — what `value` converges to over whole GOP cycles"}
    if serial:
        out["single_stream"] = serial
    if ex is not None:
        out["exchange"] = {"pictures": ex_count[0], "bytes_per_rank": int(ex.bytes_published), "collective": "broadcast (RCCL)", "overlapped": True, "every_steps": args.exchange_every,
                           "exchange_ms_per_picture": round(1000.0 * (dt - (steps * world / no_exchange["value"])) / steps, 4) if no_exchange else None}
        out["no_exchange"] = no_exchange

    # ---- kernels
    alg = {l: {"ME_stage": None, "ME_int": None, "ME_item": None, "TU": workloads[l].alg_bytes_tu, "DMVR": workloads[l].alg_bytes_dmvr} for l in workloads}
    for l, wl in workloads.items():
        alg[l].update(wl.alg_bytes_by_kernel)
    kern = {}
    for k in ("ME_stage", "ME_int", "ME_item", "TU", "DMVR"):
        ms = sum(GOP_WEIGHT[l] * per_layer[l][k] for l in per_layer) / wsum
        ab = sum(GOP_WEIGHT[l] * (alg[l][k] or 0) for l in per_layer) / wsum
        ub = sum(GOP_WEIGHT[l] * ((workloads[l].unique_bytes_by_kernel or {}).get(k) or 0) for l in per_layer) / wsum
        kern[k] = {"kernel": KERNEL_NAMES[k], "avg_ms_per_picture": ms, "ms_by_layer": {str(l): round(per_layer[l][k], 4) for l in per_layer}, "alg_bytes_per_picture": int(ab),
                   "unique_bytes_per_picture": int(ub), "nominal_alg_GBps": (ab / (ms * 1e-3) / 1e9) if ms > 0 else None}
    out["kernels"] = kern
    out["kernels_measured"] = "HIP events on the launch stream around every kernel (inside vvhip_me_plan_run for the plan's kernels), 8 passes per layer with the launches serialized; GOP-weighted"
    return out, workloads, kern


# ---------------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--streams", type=int, default=5, help="HIP streams a picture's independent launch groups are issued on: 5 = refinement stages / integer windows / table calls / TU / DMVR, "
                                                           "3 = motion-search plan / TU / DMVR, 1 = serialized")
    ap.add_argument("--exchange-every", type=int, default=2, help="N > 1: one reference-picture broadcast every this many steps (a step is one picture; default 2: 16 of the 32 "
                    "pictures of a random-access GOP cycle, temporal layers 0-4, are references of other pictures and have to reach the other GPUs, the 16 of layer 5 do not)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-mctf", action="store_true")
    ap.add_argument("--no-4k", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-medium", action="store_true", help="skip the replay of the preset-medium 4K picture (BASELINE configs[3]'s lists)")
    ap.add_argument("--profile-md", default=None, help="also write the rocprofv3 summary of this run (kernel table + counters per kernel class) as markdown to this path (the 4K pass: PATH with _4k before the extension)")
    ap.add_argument("--no-profile", action="store_true", help="skip the rocprofv3 passes of the inner run (kernel trace + one pass per counter)")
    ap.add_argument("--e2e-threads", type=int, default=8)
    ap.add_argument("--inner-mctf", action="store_true", help="(internal) the short run rocprofv3 wraps for the MCTF stage: six motion estimations of one 1080p picture against 4 references")
    ap.add_argument("--inner", action="store_true", help="(internal) the short run rocprofv3 wraps: the recorded pictures' launches serialized on one stream, no extras, no output line")
    args = ap.parse_args()

    rank, local_rank, world = sharding.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: vvenc_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    from vvenc_amd.hotpath import HotPath
    from vvenc_amd.replay import RecordedWorkload
    hp = HotPath("cuda:%d" % local_rank)
    if args.inner_mctf:
        wl = Mctf1080(args.width, args.height)
        cur = hp.plane(wl.cur_np, 128)
        refs = [hp.plane(np.roll(wl.ref_np, (k, -2 * k), (0, 1)), 128) for k in range(4)]
        outs, _ = hp.mctf_motion_estimation(cur, refs, wl.bit_depth, 16, 4, args.width >= 1920)
        for _ in range(6):
            hp.mctf_motion_estimation(cur, refs, wl.bit_depth, 16, 4, args.width >= 1920, out=outs)
        torch.cuda.synchronize()
        return
    if args.inner:
        pics, _ = prepare_recordings(args.width, args.height, 65, sorted(LAYER_POCS.values()))
        workloads = {layer: RecordedWorkload(hp, pics[poc], unique_bytes=False) for layer, poc in LAYER_POCS.items()}
        for s in range(args.warmup + args.steps):
            workloads[layer_of_step(s)].run()
        torch.cuda.synchronize()
        return

    core, workloads, kern = replay_pass(args, hp, rank, world, args.width, args.height, args.steps, args.warmup)
    inst = None
    if world > 1 and not args.no_e2e:
        try:
            inst = e2e_instances(rank, local_rank, world)
        except Exception as e:
            inst = {"error": str(e)[:300]} if rank == 0 else None
    if rank != 0:
        return
    out = {"metric": "frames/sec + bit-exact vs CPU, 1080p/4K 10-bit preset=faster at 1/2/4/8 GPU", "value": core["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": core["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i16", "data": "synthetic"}
    out.update({k: v for k, v in core.items() if k not in out})
    if inst is not None:
        out["e2e_instances"] = inst

    def profiled(width, height, kern_, workloads_, counters, md):
        """roofline objects of one resolution from this run's own rocprofv3 passes"""
        live = None
        res = {}
        if not args.no_profile and world == 1 and shutil.which("rocprofv3"):
            try:
                live = live_profile(width, height, counters)
                res["kernel_trace"] = live["kernel_trace"]
                if live.get("pmc_errors"):
                    res["pmc_errors"] = live["pmc_errors"]
            except Exception as e:
                res["kernel_trace"] = {"error": str(e)[:300]}
        wsum = float(sum(GOP_WEIGHT.values()))
        uniq = {k: sum(GOP_WEIGHT[l] * ((workloads_[l].unique_bytes_by_kernel or {}).get(k) or 0) for l in workloads_) / wsum for k in KERNEL_NAMES}
        roof, allk = roofline_objects(kern_, live, calib, uniq, md)
        res["roofline"] = roof
        if allk:
            res["roofline_all_kernels"] = allk
        return res

    calib = counter_calibration() if (not args.no_profile and world == 1 and shutil.which("rocprofv3")) else {"measured": False, "factors": {"rows16": 2.0, "stream16": 2.0, "store8": 1.0}, "how": "not run", "pattern_of_class": FETCH_PATTERN}
    out["counter_calibration"] = calib
    out.update(profiled(args.width, args.height, kern, workloads, ALL_COUNTERS, args.profile_md))

    if not args.no_parity:
        try:
            out["parity"] = parity_check(workloads)
        except Exception as e:
            out["parity"] = {"status": "not checked", "error": str(e)[:300]}
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(workloads)
        except Exception as e:   # the baseline is a report, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (e,)}

    # ---- the 3840x2160 replay (BASELINE: "1080p/4K"): the same pass on lists recorded from the 4K x 65 encode
    if world == 1 and not args.no_4k and (args.width, args.height) == (1920, 1080):
        del workloads
        torch.cuda.empty_cache()
        try:
            c4, w4, k4 = replay_pass(args, hp, rank, world, 3840, 2160, args.steps, max(4, args.warmup // 2))
            out["value_4k"], out["ms_per_step_4k"], out["steps_4k"] = c4["value"], c4["ms_per_step"], c4["steps"]
            out["config_4k"] = c4["config"]
            for k in ("gop_weighted", "single_stream", "kernels"):
                if k in c4:
                    out[k + "_4k"] = c4[k]
            md4 = (os.path.splitext(args.profile_md)[0] + "_4k" + os.path.splitext(args.profile_md)[1]) if args.profile_md else None
            for k, v in profiled(3840, 2160, k4, w4, ALL_COUNTERS[:2] + ALL_COUNTERS[3:4], md4).items():
                out[k + "_4k"] = v
            if not args.no_parity:
                try:
                    out["parity_4k"] = parity_check(w4)
                except Exception as e:
                    out["parity_4k"] = {"status": "not checked", "error": str(e)[:300]}
            if not args.no_cpu_baseline:
                try:
                    out["cpu_baseline_4k"] = cpu_baseline(w4, passes=3)
                except Exception as e:
                    out["cpu_baseline_4k"] = {"value": None, "sample": "failed: %r" % (e,)}
            del w4
            torch.cuda.empty_cache()
        except Exception as e:
            out["value_4k"] = None
            out["error_4k"] = str(e)[:400]

    if world == 1 and not args.no_4k and not args.no_medium and (args.width, args.height) == (1920, 1080):
        try:
            out["config3_medium_4k"] = replay_medium_4k(hp)
        except Exception as e:
            out["config3_medium_4k"] = {"error": str(e)[:300]}
    if not args.no_mctf and world == 1:
        try:
            import bench_synthetic as BS
            m, _ = BS.mctf_stage(hp, Mctf1080(args.width, args.height), 4)
            out["mctf"] = {k: v for k, v in m.items() if k != "me_frac_hbm_unique"}
            if not args.no_profile and shutil.which("rocprofv3"):
                try:
                    out["mctf"]["profile"] = mctf_profile(args)
                except Exception as e:
                    out["mctf"]["profile"] = {"error": str(e)[:200]}
            if not args.no_4k:
                m4, _ = BS.mctf_stage(hp, Mctf1080(3840, 2160), 4, reps=3)
                out["mctf_4k"] = {k: v for k, v in m4.items() if k != "me_frac_hbm_unique"}
        except Exception as e:
            out["mctf"] = {"error": str(e)[:300]}
    if not args.no_e2e and world == 1:
        try:
            out["e2e"] = e2e_encoder(1920, 1080, 65, args.e2e_threads, 3)
        except Exception as e:
            out["e2e"] = {"error": str(e)[:300]}
        if not args.no_4k:
            try:
                out["e2e_4k"] = e2e_encoder(3840, 2160, 65, args.e2e_threads, 3)
            except Exception as e:
                out["e2e_4k"] = {"error": str(e)[:300]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
