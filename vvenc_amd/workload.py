"""Synthetic per-frame hot-path workload (BASELINE.json configs[1..2]).

One "frame" = the work the reference pushes through its kernel tables for one inter picture at preset
faster, sized from the survey's measurements (~54 x 1.5*W*H sample pairs through
DistParam::distFunc and ~2.05 x 1.5*W*H transformed coefficients per B-frame):

  distortion  for every aligned SxS luma block, S in {8,16,32,64} (CU sizes of preset faster):
                11 SAD candidates  (integer ME with bIntegerET: start + predictors + two 4-point rings;
                                    subShift=1 for S>8, RdCost.cpp:187-193)
                 8 HAD_fast cands  (pruned half/quarter-pel refinement, InterSearch.cpp:760-972)
                 1 SSE             (getDistPart after reconstruction)
              -> 4 sizes x 20 = 80 x W*H sample pairs  (~ 54 x 1.5 = 81)
  transform   the luma plane tiled once each with 8x8, 16x16, 32x32 TUs through the fused
              xT -> needRdoq/quant -> dequant -> xIT -> SSE pipeline  -> 3 x W*H coefficients (~ 2.05 x 1.5)

Candidate displacements are seeded pseudo-random integer vectors within +-16 samples of a global pan.
All of it is generated on the host once and stays resident in HBM; nothing here computes distortion.
"""
import os

import numpy as np

from .hotpath import HotPath, Plane

SIZES = (8, 16, 32, 64)
TU_SIZES = (8, 16, 32)
N_SAD, N_HAD, N_SSE = 11, 8, 1


def synth_frame_pair(width, height, seed, bit_depth=10, frame_index=0):
    """SURVEY §8d generator family: textured base + pan + per-frame noise (current, reference).  frame_index > 0: another picture of the same sequence (the current
    picture is taken further along the pan and gets its own noise; the reference picture is the same for every index)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height + 64, 0:width + 64].astype(np.float32)
    base = 512 + 180 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 120 * np.sin((xx + yy) / 11.0) + 60 * np.sin(xx / 3.1) * np.sin(yy / 4.3)
    base += rng.normal(0, 12, base.shape).astype(np.float32)
    maxv = (1 << bit_depth) - 1
    scale = maxv / 1023.0
    ref = np.clip((base[31:31 + height, 29:29 + width] + rng.normal(0, 3, (height, width))) * scale, 0, maxv)   # pan (3,1)
    k = int(frame_index) % 8
    cur = base[32 + k:32 + k + height, 32 + 3 * k:32 + 3 * k + width]
    if frame_index:
        cur = cur + np.random.default_rng(seed + 7919 * int(frame_index)).normal(0, 3, (height, width)).astype(np.float32)
    cur = np.clip(cur * scale, 0, maxv)
    return cur.astype(np.int16), ref.astype(np.int16)


class FrameWorkload:
    """Device-resident work lists for one frame geometry."""

    def __init__(self, hp: HotPath, width=1920, height=1080, seed=1080, margin=80, bit_depth=10, frame_index=0):
        self.hp, self.width, self.height, self.bit_depth = hp, width, height, bit_depth
        cur, ref = synth_frame_pair(width, height, seed, bit_depth, frame_index)
        self.cur_np, self.ref_np = cur, ref
        self.org = hp.plane(cur, margin)              # original picture (padding like PelStorage margins)
        self.ref = hp.plane(ref, margin)              # reconstructed reference picture, margin = CTU+16 (EncStage.h:311)
        resi = (cur.astype(np.int32) - ref.astype(np.int32)).astype(np.int16)
        self.resi = hp.plane(resi, 0)
        # 8x8-tiled copies of the two planes, resident like the planes themselves (made once per picture: tile_ref() again when the reference plane changes)
        self.tiled = os.environ.get("VVHIP_WORKLOAD_TILED", "1") != "0"
        self.org_tiled = hp.tile_plane(self.org) if self.tiled else None
        self.ref_tiled = hp.tile_plane(self.ref) if self.tiled else None
        # one-sample-shifted copy of the reference plane (dword-aligned loads for candidates at odd sample addresses), resident like the tiled copies
        self.shifted = os.environ.get("VVHIP_WORKLOAD_SHIFT1", "1") != "0"
        self.ref_shift = hp.shift_plane(self.ref) if self.shifted else None
        rng = np.random.default_rng(seed + 1)
        self.dist_jobs = []     # (func, S, sub_shift, n, d_items, d_out, host_items)
        self.pairs = 0
        self.alg_bytes = {}
        for S in SIZES:
            bx, by = np.meshgrid(np.arange(0, width - S + 1, S), np.arange(0, height - S + 1, S))
            bx, by = bx.ravel(), by.ravel()
            nb = bx.size
            for func, ncand in (("SAD", N_SAD), ("HAD_fast", N_HAD), ("SSE", N_SSE)):
                dx = rng.integers(-16, 17, size=(nb, ncand)) + 3
                dy = rng.integers(-16, 17, size=(nb, ncand)) + 1
                org_off = np.repeat((by * self.org.stride + bx)[:, None], ncand, 1)
                cur_off = (by[:, None] + dy) * self.ref.stride + (bx[:, None] + dx)
                items = np.stack([org_off.ravel(), cur_off.ravel()], 1).astype(np.int32)
                n = items.shape[0]
                ss = 1 if (func == "SAD" and S > 8) else 0
                d_items = hp.to_device(items)
                d_out = hp.to_device(np.zeros(n, np.int64))
                self.dist_jobs.append((func, S, ss, n, d_items, d_out, items))
                self.pairs += n * S * S
                # algorithmic bytes: 4*w*h per candidate (org + cur, int16), rows halved under subShift, + 8 B result
                self.alg_bytes[func] = self.alg_bytes.get(func, 0) + n * (4 * S * (S >> ss) + 8)
        self.tu_jobs = []       # (S, n, d_off, d_qp, d_level, d_rec, d_stats)
        self.coefs = 0
        for S in TU_SIZES:
            bx, by = np.meshgrid(np.arange(0, width - S + 1, S), np.arange(0, height - S + 1, S))
            off = (by.ravel() * self.resi.stride + bx.ravel()).astype(np.int32)
            n = off.size
            qps = rng.integers(30, 48, size=n)      # QP 32 +- with qpBdOffset 12 -> base QP around 44; spread like QPA
            d_qp = hp.to_device(HotPath.tu_qp(qps, 0, 1))
            import torch
            lvl = torch.empty(n * S * S, dtype=torch.int16, device=hp.device)
            rec = torch.empty(n * S * S, dtype=torch.int16, device=hp.device)
            st = torch.empty((n, 24), dtype=torch.uint8, device=hp.device)
            self.tu_jobs.append((S, n, hp.to_device(off), d_qp, lvl, rec, st, off, qps))
            self.coefs += n * S * S
            # fused pipeline: read residual 2 B, write level 2 B + reconstructed residual 2 B per sample, 24 B stats per TU
            self.alg_bytes["TU%d" % S] = n * (6 * S * S + 24)

        # launches of one kernel class are issued back to back so a class can be bracketed by ONE pair of stream events
        self.dist_jobs.sort(key=lambda j: ("SAD", "HAD_fast", "SSE").index(j[0]))
        self.class_launches = {"SAD": len(SIZES), "HAD_fast": len(SIZES), "SSE": len(SIZES)}
        self.class_launches.update({"TU%d" % S: 1 for S in TU_SIZES})        # each fused TU size is its own kernel instantiation

        self.merged = os.environ.get("VVHIP_WORKLOAD_UNMERGED", "0") != "1"      # one vvhip_dist_multi launch per function (all block sizes) instead of one launch per size
        self.job_tables = {func: hp.make_dist_jobs([(S, S, ss, n, it, out) for (f, S, ss, n, it, out, _) in self.dist_jobs if f == func])
                           for func in ("SAD", "HAD_fast", "SSE")}
        self.tu_table = hp.make_tu_jobs([(S, S, 0, 0, n, 8, d_off, d_qp, lvl, rec, st) for (S, n, d_off, d_qp, lvl, rec, st, _, _) in self.tu_jobs])
        self.alg_bytes["TU"] = sum(self.alg_bytes["TU%d" % S] for S in TU_SIZES)
        self.class_launches_merged = {"SAD_SSE": 1, "HAD_fast": 1, "TU": 1}
        self.alg_bytes["SAD_SSE"] = self.alg_bytes["SAD"] + self.alg_bytes["SSE"]
        # SAD and SSE lists share one launch (vvhip_dist_multi_func), the Hadamard lists another
        self.fjob_tables = {"SAD_SSE": hp.make_dist_fjobs([(f, S, S, ss, n, it, out) for (f, S, ss, n, it, out, _) in self.dist_jobs if f in ("SAD", "SSE")], flags=hp.DIST_FLAG_SAMPLES),
                            # original vs reconstructed picture samples: the Hadamard jobs may use the all-packed tile (VVHIP_DIST_FLAG_SAMPLES)
                            "HAD_fast": hp.make_dist_fjobs([(f, S, S, ss, n, it, out) for (f, S, ss, n, it, out, _) in self.dist_jobs if f == "HAD_fast"],
                                                           flags=hp.DIST_FLAG_SAMPLES)}

    # ---- optional fractional-ME stage: 16 sub-pel positions (8 half-sample + 8 quarter-sample neighbours of a seeded
    # base vector) per block of every size, interpolated and scored with HAD_fast like InterSearch::xPatternRefinement — one
    # vvhip_subpel_refine_batch call per block size
    SUBPEL_OFFSETS = [(-8, 0), (8, 0), (0, -8), (0, 8), (-8, -8), (8, -8), (-8, 8), (8, 8), (-4, 0), (4, 0), (0, -4), (0, 4), (-4, -4), (4, -4), (-4, 4), (4, 4)]

    def enable_subpel(self):
        import torch
        from .hotpath import SUBPEL_DTYPE
        hp = self.hp
        rng = np.random.default_rng(4242)
        self.subpel_jobs = []
        self.alg_bytes["SUBPEL"] = 0
        for S in SIZES:
            bx, by = np.meshgrid(np.arange(0, self.width - S + 1, S), np.arange(0, self.height - S + 1, S))
            bx, by = bx.ravel(), by.ravel()
            nb = bx.size
            mvx, mvy = rng.integers(-12, 13, nb) * 4 + 12, rng.integers(-12, 13, nb) * 4 + 4          # base vectors, quarter-sample units
            bases = np.zeros(nb, SUBPEL_DTYPE)
            bases["org_off"] = by * self.org.stride + bx
            bases["ref_off"] = (by + (mvy >> 2)) * self.ref.stride + bx + (mvx >> 2)
            bases["frac_x"] = (mvx & 3) << 2
            bases["frac_y"] = (mvy & 3) << 2
            n = nb * len(self.SUBPEL_OFFSETS)
            self.subpel_jobs.append((S, nb, hp.to_device(bases), torch.empty(n, dtype=torch.int64, device=hp.device), bases))
            # per position: (S+7)^2 reference samples in, S^2 original samples in, 8 B out
            self.alg_bytes["SUBPEL"] += n * (2 * (S + 7) * (S + 7) + 2 * S * S + 8)
        self.class_launches["SUBPEL"] = 2 * len(SIZES)
        self.class_launches_merged["SUBPEL"] = 2 * len(SIZES)

    def run_subpel(self, timers=None):
        if timers is not None:
            timers.start("SUBPEL")
        for (S, nb, d_b, out, _) in self.subpel_jobs:
            self.hp.subpel_refine_batch("HAD_fast", self.org, self.ref, d_b, nb, self.SUBPEL_OFFSETS, S, S, self.bit_depth, 0, False, out=out)
        if timers is not None:
            timers.stop("SUBPEL")

    def tile_ref(self, ref=None, out=None):
        """the reference plane changed (a new picture arrived): refresh its derived copies (8x8-tiled, one-sample-shifted)"""
        if self.tiled:
            self.ref_tiled = self.hp.tile_plane(ref if ref is not None else self.ref, out if out is not None else self.ref_tiled)
        if self.shifted:
            self.ref_shift = self.hp.shift_plane(ref if ref is not None else self.ref, self.ref_shift)
        return self.ref_tiled

    def _dist(self, cls):
        if self.tiled or self.shifted:
            self.hp.dist_multi_func_tiled(self.org, self.ref, self.org_tiled, self.ref_tiled, self.fjob_tables[cls], self.bit_depth, cur_shift=self.ref_shift)
        else:
            self.hp.dist_multi_func(self.org, self.ref, self.fjob_tables[cls], self.bit_depth)

    def _lanes(self, streams, derive=False):
        """one forked context per stream, and per class a pre-bound launch (the arguments of a picture's three launches never change between steps, except the reference
        plane of the sharded run: the bound calls are rebuilt when it does).
        derive: every step is a new picture, so the copies the library derives from its planes are made per step, by the lane that reads them and in front of its launch (stream
        order, no events): the SAD / SSE lane makes the tiled copies of both planes and its shifted reference copy in ONE launch (vvhip_planes_derive), the Hadamard lane its own
        shifted copy.  Without it the lanes read the copies made when the workload was built (tile_ref() refreshes them)."""
        import ctypes as C
        import torch
        key = (tuple(id(s) for s in streams), self.ref.storage.data_ptr(), id(self.ref_tiled), id(self.ref_shift), bool(derive))
        if getattr(self, "_lane_key", None) != key:
            if getattr(self, "_lane_ctx", None) is None or self._lane_streams != key[0]:
                self._lane_ctx = [self.hp.fork(s) for s in streams]
                self._lane_streams = key[0]
            self._lane_cache = getattr(self, "_lane_cache", {})
            ck = key[1:]
            if ck not in self._lane_cache:
                hp, calls = self.hp, []
                self._tiled_struct = getattr(self, "_tiled_struct", [])
                if derive and (self.tiled or self.shifted) and getattr(self, "_lane_copies", None) is None:
                    # per-lane derived copies (each lane rewrites its own, in stream order behind its previous launch)
                    self._lane_copies = {"SAD_SSE": (torch.empty_like(self.org_tiled) if self.tiled else None, torch.empty_like(self.ref_tiled) if self.tiled else None,
                                                     torch.empty_like(self.ref.storage) if self.shifted else None),
                                         "HAD_fast": (None, None, torch.empty_like(self.ref.storage) if self.shifted else None)}
                rows_o, rows_c = self.org.storage.shape[0], self.ref.storage.shape[0]
                for i, cls in enumerate(("SAD_SSE", "HAD_fast", "TU")):
                    lane = self._lane_ctx[i % len(self._lane_ctx)]
                    if cls == "TU":
                        tab = self.tu_table
                        calls.append((cls, lane, None, lane.bound("vvhip_tu_rdo_multi", self.resi.buf_ptr, self.resi.stride, self.bit_depth, tab[0], tab[1])))
                        continue
                    tab = self.fjob_tables[cls]
                    pre = None
                    o_t, r_t, r_s = (self.org_tiled if self.tiled else None), (self.ref_tiled if self.tiled else None), (self.ref_shift if self.shifted else None)
                    if derive and (self.tiled or self.shifted):
                        o_t, r_t, r_s = self._lane_copies[cls]
                        if cls == "HAD_fast":
                            o_t, r_t = None, None                  # (the Hadamard lists do not read the tiled copies)
                        if o_t is not None or r_t is not None or r_s is not None:
                            pre = lane.bound("vvhip_planes_derive", self.org.storage.data_ptr(), self.org.stride, rows_o, o_t.data_ptr() if o_t is not None else None,
                                             self.ref.storage.data_ptr(), self.ref.stride, rows_c, r_t.data_ptr() if r_t is not None else None, r_s.data_ptr() if r_s is not None else None)
                    t = hp._TiledPlanes(o_t.data_ptr() if o_t is not None else None, r_t.data_ptr() if r_t is not None else None, self.org.pad, self.ref.pad,
                                        r_s.data_ptr() + 2 * self.ref.origin if r_s is not None else None)
                    self._tiled_struct.append(t)          # (kept alive: the bound call holds a pointer to it)
                    calls.append((cls, lane, pre, lane.bound("vvhip_dist_multi_func_tiled", self.org.buf_ptr, self.org.stride, self.ref.buf_ptr, self.ref.stride, C.byref(t), self.bit_depth,
                                                             tab[0], tab[1])))
                self._lane_cache[ck] = calls
            self._lane_calls = self._lane_cache[ck]
            self._lane_key = key
        return self._lane_calls

    def run_overlapped(self, streams, timers=None, derive=False):
        """the same three launches, each on its own HIP stream (they are independent work lists): they share the device and successive steps pipeline per stream.
        Every stream has its own context (HotPath.fork) and every launch is a pre-bound call, so a step costs the host three (derive: five) foreign calls and nothing else.
        timers: per-class HIP events, recorded on the class's own stream (around the class's launch, behind its derive launch)"""
        for cls, lane, pre, call in self._lanes(streams, derive):
            if pre is not None:
                pre()
            if timers is not None and cls in timers.pool:
                timers.start(cls, lane.stream)
                call()
                timers.stop(cls, lane.stream)
            else:
                call()

    # one pass of the hot path over the frame: 2 merged distortion launches (SAD+SSE, Hadamard) + 1 merged fused-TU launch (or 12 + 3 per-size ones)
    def run(self, timers=None):
        hp = self.hp
        prev = None
        if self.merged:
            for cls in ("SAD_SSE", "HAD_fast"):
                if timers is not None:
                    timers.start(cls)
                self._dist(cls)
                if timers is not None:
                    timers.stop(cls)
        else:
            for (func, S, ss, n, d_items, d_out, _) in self.dist_jobs:
                if timers is not None and func != prev:
                    if prev is not None:
                        timers.stop(prev)
                    timers.start(func)
                    prev = func
                hp.dist_batch(func, self.org, self.ref, d_items, n, S, S, ss, self.bit_depth, out=d_out)
            if timers is not None:
                timers.stop(prev)
        if self.merged:
            if timers is not None:
                timers.start("TU")
            hp.tu_rdo_multi(self.resi, self.tu_table, self.bit_depth)
            if timers is not None:
                timers.stop("TU")
            if getattr(self, "subpel_jobs", None):
                self.run_subpel(timers)
            return
        for (S, n, d_off, d_qp, lvl, rec, st, _, _) in self.tu_jobs:
            if timers is not None:
                timers.start("TU%d" % S)
            hp.tu_rdo(self.resi, d_off, n, S, S, d_qp, 0, 0, self.bit_depth, 8, lvl, rec, st)
            if timers is not None:
                timers.stop("TU%d" % S)

    def checksum(self):
        """order-independent digest of every result of the last run (used by tests: equal across ranks / reruns)"""
        import torch
        acc = 0
        for job in self.dist_jobs:
            acc ^= int(torch.sum(job[5]).item()) & 0xFFFFFFFFFFFF
        for job in self.tu_jobs:
            acc ^= int(torch.sum(job[4].to(torch.int64)).item()) & 0xFFFFFFFFFFFF
            acc ^= (int(torch.sum(job[5].to(torch.int64)).item()) << 1) & 0xFFFFFFFFFFFF
        return acc
