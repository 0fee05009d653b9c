"""bench.py's north-star leg C: the MCTF pre-filter (hierarchical block matching + bilateral filter) of the pictures a GOP cycle filters, at the GOP's cadence.

What the reference does at preset faster (MCTF 2, MCTFSpeed 4, GOP 32, QP 32; vvencCfg.cpp:1498-1509, CommonLib/MCTF.cpp:595-599, :745-800): MCTFFrames = {8, 16, 32}; a picture whose
POC is a multiple of 32 or of 16 is filtered against 2 neighbours on each side (4 motion estimations), a multiple of 8 against 1 on each side (2) — 12 motion estimations and 4
bilateral filters per 32 pictures — always from ORIGINAL pictures (independent of the encode).  The adaptive extra references (:803-860) are data-dependent and not issued here.
In STEP_LAYERS (tools/bench_common.py) the TL0 picture is the POC-32 one, TL1 POC 16, the two TL2 pictures POC 8 and 24.

Host logic (JOBS, the byte formulas, the cadence) is importable without torch; the device side needs vvenc_amd."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (temporal layer, POC, reference POCs, overall strength = MCTFStrengths[mctfIdx] at QP 32 / GOP 32: vvencCfg.cpp:1503-1507)
JOBS = ((0, 32, (30, 31, 33, 34), 1.5), (1, 16, (14, 15, 17, 18), 1.0), (2, 8, (7, 9), 2.0 / 3.0), (2, 24, (23, 25), 2.0 / 3.0))
UNIT, SPEED, QP, BIT_DEPTH = 16, 4, 32, 10
MCTF_KERNELS = {"MCTF_search": "meSearchKernel", "MCTF_nb": "meNeighbourKernel", "MCTF_fix": "meFixKernel", "MCTF_apply": "mctfApplyKernel"}


def jobs_of_layer(layer):
    return [j for j in JOBS if j[0] == layer]


def job_of_step(s):
    """the MCTF job step s of the GOP cycle issues (None for the 28 pictures that are not filtered): the k-th TL2 step of a cycle takes the k-th TL2 job"""
    from bench_common import STEP_LAYERS
    s %= 32
    layer = STEP_LAYERS[s]
    js = jobs_of_layer(layer)
    if not js:
        return None
    k = sum(1 for t in range(s) if STEP_LAYERS[t] == layer)
    return js[k % len(js)]


def motion_estimations_per_cycle():
    return sum(len(job_of_step(s)[2]) for s in range(32) if job_of_step(s))


def apply_alg_bytes(width, height, n_refs, unit=UNIT, taps=6):
    """algorithmic bytes of the bilateral filter of one 4:2:0 picture (SURVEY 8d's accounting carried to MCTF.cpp:1399-1487): per block the original in and the filtered block
    out (2 w h each) and per reference the compensated block's support ((w + taps - 1)(h + taps - 1) 2) + its motion record (24 B)"""
    total = 0
    for cs in (0, 1, 1):
        w, h, b = width >> cs, height >> cs, unit >> cs
        for by in range(0, h, b):
            bh = min(b, h - by)
            nx_full, rem = divmod(w, b)
            for bw, n in ((b, nx_full), (rem, 1 if rem else 0)):
                total += n * (4 * bw * bh + n_refs * ((bw + taps - 1) * (bh + taps - 1) * 2 + 24))
    return int(total)


class MctfCadence:
    """the four filtered pictures of a GOP cycle resident on the device (originals of the encoded clip: tests/e2e_fps.synth_clip), issued asynchronously on their own lane"""

    def __init__(self, hp, width, height, lane=None, more_lanes=()):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import e2e_fps
        from vvenc_amd.hotpath import MV_DTYPE
        self.hp, self.width, self.height = hp, width, height
        self.lane = lane if lane is not None else hp
        self.lanes = [self.lane] + list(more_lanes)                     # the cadence deals the cycle's jobs over these in turn (issue_step); everything else runs on the first
        y, u, v = e2e_fps.synth_clip(width, height, 65)
        pocs = sorted({p for _, poc, refs, _ in JOBS for p in (poc,) + refs})
        self.np_planes = {p: (y[p], u[p], v[p]) for p in pocs}
        self.planes = {p: tuple(hp.plane(self.np_planes[p][c], 128 >> (1 if c else 0)) for c in range(3)) for p in pocs}
        self.add_level = width >= 1920                                  # MCTF.cpp:768
        self.mv_w, self.mv_h = (width + UNIT - 1) // UNIT, (height + UNIT - 1) // UNIT
        self.jobs = {}
        for layer, poc, refs, strength in JOBS:
            fields = [torch.empty((self.mv_w * self.mv_h, MV_DTYPE.itemsize), dtype=torch.uint8, device=hp.device) for _ in refs]
            outs = [hp.plane(np.zeros_like(self.np_planes[poc][c]), 0) for c in range(3)]
            prm = [hp.mctf_filter_params(QP, BIT_DEPTH, strength, c > 0) for c in range(3)]
            rs = [hp.REF_STRENGTHS[0][abs(r - poc) - 1] for r in refs]          # m_refStrengths row of random access, MCTF.cpp:112-117, index = |POC offset| - 1
            self.jobs[poc] = dict(layer=layer, poc=poc, refs=refs, strength=strength, fields=fields, outs=outs, prm=prm, ref_strengths=rs)
        torch.cuda.synchronize()

    # ---- issue
    def issue(self, job, ctx=None, me=True, apply=True):
        """queues the job's motion estimation (all references, one call) and the filter of Y, U, V on the lane's stream; no host wait"""
        c = ctx or self.lane
        j = self.jobs[job[1]]
        cur = self.planes[j["poc"]]
        if me:
            c.mctf_motion_estimation(cur[0], [self.planes[r][0] for r in j["refs"]], BIT_DEPTH, UNIT, SPEED, self.add_level, out=j["fields"], wait=False)
        if apply:
            for comp in range(3):
                sigma, scaling = j["prm"][comp]
                c.mctf_apply_plane(cur[comp], [self.planes[r][comp] for r in j["refs"]], j["fields"], self.mv_w, 1 if comp else 0, j["ref_strengths"], scaling, sigma,
                                   BIT_DEPTH, UNIT, False, QP, out=j["outs"][comp])

    # ---- N > 1: the references of a job arrive through the picture exchange (vvenc_amd/sharding.PictureExchange): slot = up to four luma planes with their margins
    def exchange_shapes(self):
        p = self.planes[JOBS[0][1]][0]
        return [tuple(p.storage.shape)] * 4

    def fill_slot(self, job, slot):
        """owner side: the reference originals of `job` (luma incl. margins) into the slot, on the current stream"""
        for i, r in enumerate(job[2]):
            slot[i].copy_(self.planes[r][0].storage)

    def slot_planes(self, job, slot):
        from vvenc_amd.hotpath import Plane
        cur = self.planes[job[1]][0]
        out = []
        for i in range(len(job[2])):
            p = Plane.__new__(Plane)
            p.width, p.height, p.pad, p.stride, p.storage = cur.width, cur.height, cur.pad, cur.stride, slot[i]
            out.append(p)
        return out

    def issue_from_slot(self, job, slot, fields=None):
        """consumer side: the job's motion estimation against the RECEIVED reference planes (the fields go to `fields`, default the job's exchange fields), then the filter
        (luma references from the slot too, chroma local)"""
        import torch
        from vvenc_amd.hotpath import MV_DTYPE
        c = self.lane
        j = self.jobs[job[1]]
        if "fields_x" not in j:
            j["fields_x"] = [torch.empty((self.mv_w * self.mv_h, MV_DTYPE.itemsize), dtype=torch.uint8, device=self.hp.device) for _ in j["refs"]]
        f = fields if fields is not None else j["fields_x"]
        cur = self.planes[j["poc"]]
        refs_y = self.slot_planes(job, slot)
        c.mctf_motion_estimation(cur[0], refs_y, BIT_DEPTH, UNIT, SPEED, self.add_level, out=f, wait=False)
        for comp in range(3):
            sigma, scaling = j["prm"][comp]
            rp = refs_y if comp == 0 else [self.planes[r][comp] for r in j["refs"]]
            c.mctf_apply_plane(cur[comp], rp, f, self.mv_w, 1 if comp else 0, j["ref_strengths"], scaling, sigma, BIT_DEPTH, UNIT, False, QP, out=j["outs"][comp])
        j["consumed"] = j.get("consumed", 0) + 1

    def exchange_parity(self, last_slots):
        """the fields computed from RECEIVED planes against the fields from this rank's own copies, for every job this rank consumed; + the self-test that the consumer really
        reads the slot: a slot with a few samples changed must give different fields.  last_slots: {poc: slot of the job's last exchange}"""
        import torch
        res = {"jobs_checked": 0, "mismatches": 0, "corruption_detected": None}
        for job in JOBS:
            j = self.jobs[job[1]]
            if not j.get("consumed"):
                continue
            torch.cuda.synchronize()
            got = [f.clone() for f in j["fields_x"]]
            self.issue(job, apply=False)
            torch.cuda.synchronize()
            res["jobs_checked"] += 1
            res["mismatches"] += sum(0 if bool(torch.equal(a, b)) else 1 for a, b in zip(got, j["fields"]))
            if res["corruption_detected"] is None and job[1] in last_slots:
                slot = last_slots[job[1]]
                pad = self.planes[job[1]][0].pad
                keep = slot[0][pad + 64:pad + 128, pad + 64:pad + 192].clone()
                slot[0][pad + 64:pad + 128, pad + 64:pad + 192] = 1023 - keep          # "a wrong broadcast": part of the first reference inverted
                torch.cuda.synchronize()                                                 # (written on the default stream, read on the lane's)
                tmp = [torch.empty_like(f) for f in got]
                self.issue_from_slot(job, slot, fields=tmp)
                torch.cuda.synchronize()
                res["corruption_detected"] = any(not bool(torch.equal(a, b)) for a, b in zip(tmp, got))
                slot[0][pad + 64:pad + 128, pad + 64:pad + 192] = keep
        res["status"] = "bit-exact" if res["mismatches"] == 0 and res["jobs_checked"] > 0 and res["corruption_detected"] is not False else ("not checked" if res["jobs_checked"] == 0 else "MISMATCH")
        return res

    def issue_step(self, s):
        job = job_of_step(s)
        if job is not None:
            self.issue(job, ctx=self.lanes[JOBS.index(job) % len(self.lanes)])
        return job

    # ---- algorithmic bytes (per GOP cycle and per class), from the library's scored-candidate counters
    def count(self):
        hp = self.lane
        import torch
        per_job = {}
        for job in JOBS:
            hp.mctf_set_stats(True)
            self.issue(job, apply=False)
            st = hp.mctf_get_stats()
            per_job[job[1]] = st
        hp.mctf_set_stats(False)
        torch.cuda.synchronize()
        self.stats = per_job
        cyc = {"MCTF_search": 0, "MCTF_nb": 0, "MCTF_fix": 0, "MCTF_apply": 0}
        cand = {"int": 0, "frac": 0, "grid_positions": 0, "ring_positions": 0}
        per_cand = 0
        for job in JOBS:
            st = per_job[job[1]]
            for cls, ph in (("MCTF_search", "search"), ("MCTF_nb", "neighbour"), ("MCTF_fix", "sweep")):
                # one-by-one candidates at 4 w h / (w + 3)(h + 3) 2 + 2 w h each; the dense integer grids AND the refinement rings (positions within half a sample of
                # their centre, horizontal passes shared) in its WINDOW form (window + block read once + 8 B per position)
                cyc[cls] += st[ph]["int_bytes"] + st[ph]["frac_bytes"] + st[ph]["grid_window_bytes"] + st[ph]["ring_window_bytes"]
                per_cand += st[ph]["int_bytes"] + st[ph]["frac_bytes"] + st[ph]["grid"] * 4 * 32 * 32 + st[ph]["ring"] * ((UNIT + 3) * (UNIT + 3) * 2 + 2 * UNIT * UNIT)
                cand["int"] += st[ph]["int"]
                cand["frac"] += st[ph]["frac"]
                cand["grid_positions"] += st[ph]["grid"]
                cand["ring_positions"] += st[ph]["ring"]
            cyc["MCTF_apply"] += apply_alg_bytes(self.width, self.height, len(job[2]))
        self.alg_bytes_per_cycle = cyc
        self.per_candidate_bytes_per_cycle = per_cand          # every scored position at its own 4 w h: a work rate (LDS-level reuse), not memory traffic
        self.candidates_per_cycle = cand
        # unique bytes of a cycle: every plane a job touches once (pyramid 1 + 1/4 + 1/16 (+ 1/64) of the luma planes in and out; the filter's three planes in, one out) + fields
        pyr = 1 + 0.25 + 0.0625 + (0.015625 if self.add_level else 0)
        wh = self.width * self.height
        uniq = {"MCTF_search": 0, "MCTF_nb": 0, "MCTF_fix": 0, "MCTF_apply": 0}
        for job in JOBS:
            n = len(job[2])
            fld = self.mv_w * self.mv_h * 24
            uniq["MCTF_search"] += int(2 * wh * pyr * (1 + n) + n * fld * 1.34)
            uniq["MCTF_nb"] += int(2 * wh * pyr * (1 + n) + n * fld * 1.34 * 2)
            uniq["MCTF_fix"] += int(n * fld * 1.34 * 3)
            uniq["MCTF_apply"] += int(2 * wh * 1.5 * (2 + n) + n * fld)
        self.unique_bytes_per_cycle = uniq
        return cyc

    # ---- parity against the reference compiled here (oracle/_ref): fields bit-exact vs its x86 row (== scalar row), filtered planes vs its scalar row
    def parity(self, pocs=None):
        import torch
        sys.path.insert(0, ROOT)
        from oracle.oracle import Oracle, RefLib
        from vvenc_amd.hotpath import HotPath
        use_ref = RefLib.available()
        me_chk = RefLib(1) if use_ref else Oracle()
        flt_chk = RefLib(0) if use_ref else Oracle()
        res = {"checker": "reference (oracle/_ref: x86 row for the fields, scalar row for the filter)" if use_ref else "oracle (C restatement)", "fields": 0, "field_mismatches": 0,
               "planes": 0, "plane_mismatches": 0, "pictures": []}
        for job in JOBS:
            if pocs is not None and job[1] not in pocs:
                continue
            self.issue(job)
            torch.cuda.synchronize()
            j = self.jobs[job[1]]
            cur = self.np_planes[j["poc"]]
            mvs = []
            for k, r in enumerate(j["refs"]):
                got = HotPath.mv_to_numpy(j["fields"][k], (self.mv_w, self.mv_h))
                exp = me_chk.mctf_me(cur[0], self.np_planes[r][0], BIT_DEPTH, UNIT, SPEED, self.add_level)[4]
                res["fields"] += 1
                ok = all(np.array_equal(got[f], exp[f]) for f in ("x", "y", "error", "rmsme", "overlap"))
                res["field_mismatches"] += 0 if ok else 1
                mvs.append(exp)
            exp3 = flt_chk.mctf_bilateral(cur, [self.np_planes[r] for r in j["refs"]], mvs, [abs(r - j["poc"]) - 1 for r in j["refs"]], BIT_DEPTH, QP, UNIT, False, True, j["strength"])
            for comp in range(3):
                res["planes"] += 1
                res["plane_mismatches"] += 0 if np.array_equal(j["outs"][comp].visible().cpu().numpy(), exp3[comp]) else 1
            res["pictures"].append(j["poc"])
        res["status"] = "bit-exact" if res["field_mismatches"] == 0 and res["plane_mismatches"] == 0 else "MISMATCH"
        return res


def cycle_pictures(width, height):
    """the GOP cycle's four filtered pictures as numpy planes: [(cur (Y, U, V), [ref (Y, U, V)], [ref index], overall strength)] (oracle.RefLib.mctf_cycle_timed's input)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import e2e_fps
    y, u, v = e2e_fps.synth_clip(width, height, 65)
    P = lambda p: (y[p], u[p], v[p])
    return [(P(poc), [P(r) for r in refs], [abs(r - poc) - 1 for r in refs], strength) for _, poc, refs, strength in JOBS]


def cpu_baseline_mctf(width, height, threads):
    """the reference's own MCTF (oracle/_ref, x86 row) on the host cores over the GOP cycle's four filtered pictures — per picture its pyramid, motionEstimationLuma's five
    levels for every reference, bilateralFilter on Y, U, V (what MCTF::filter does, MCTF.cpp:726-870, without the adaptive extra references): one pass on ONE thread, three
    passes on `threads` threads (best taken: a shared host's idle cores need a second to come up).  -> ms per GOP-weighted picture (a cycle / 32) and the cycle's parts"""
    sys.path.insert(0, ROOT)
    from oracle.oracle import RefLib
    if not RefLib.available():
        return {"skipped": "oracle/_ref is not built"}
    ref = RefLib(1)
    pics = cycle_pictures(width, height)
    out = {"unit": "ms per GOP-weighted picture", "kind": "reference", "cores": int(threads),
           "sample": "the GOP cycle's 4 filtered pictures (%dx%d; 12 motion estimations of 5 levels + 4 bilateral filters of Y, U, V; the cycle's cost / 32 pictures): 1 pass on one thread, "
                     "best of 3 on %d threads ((picture, reference) estimations as independent jobs, the filter on the reference's own thread pool)" % (width, height, threads)}
    _, _, (me1, f1) = ref.mctf_cycle_timed(pics, 0)
    out["value_1thread"] = round(1000.0 * (me1 + f1) / 32.0, 3)
    out["me_ms_per_cycle_1thread"], out["filter_ms_per_cycle_1thread"] = round(1000.0 * me1, 1), round(1000.0 * f1, 1)
    if threads > 1:
        best = None
        for _ in range(3):
            _, _, (me, f) = ref.mctf_cycle_timed(pics, threads)
            if best is None or me + f < best[0] + best[1]:
                best = (me, f)
        out["value"] = round(1000.0 * (best[0] + best[1]) / 32.0, 3)
        out["me_ms_per_cycle"], out["filter_ms_per_cycle"] = round(1000.0 * best[0], 1), round(1000.0 * best[1], 1)
    else:
        out["value"] = out["value_1thread"]
    return out
