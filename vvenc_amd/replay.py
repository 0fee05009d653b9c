"""Replay of RECORDED work lists on the MI355X (bench.py, tests): turns a recorded picture (vvenc_amd/recorded.py) into device-resident planes, a motion-search plan
(vvhip_me_plan_*: integer candidates, sub-pel refinement stages and plain table calls in one launch), the TU lists of the fused transform pipeline
(vvhip_tu_rdo_multi_strided) and the DMVR list (vvhip_dmvr_refine_batch), and checks the results against the costs the REAL encoder computed while it was recorded.

Plane table of a picture's plan (<= 16 entries):  0..2 original Y / Cb / Cr (as the encoder's CTU copies read them), 3.. luma reconstruction of each reference picture
(with its margin), then ONE entry with stride 0 for the sample pool (compact blocks: the row pitch of a pool block is the block's width).

Nothing of a recording is left out (round 4): every block shape of preset medium's CTU 128 + multi-type tree (width and height independent powers of two 2..128) goes through the
plan, GEO's masked SADs included; `dropped` counts what could not be placed and the tests assert it is empty.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import recorded as R
from .hotpath import DF, HotPath, Plane, STATS_DTYPE, DMVR_ITEM_DTYPE, DMVR_RESULT_DTYPE

ME_INT_JOB = np.dtype([("org_off", "<i4"), ("ref_off", "<i4"), ("width", "<i2"), ("height", "<i2"), ("org_plane", "u1"), ("ref_plane", "u1"), ("sub_shift", "u1"), ("reserved", "u1"),
                       ("min_dx", "<i2"), ("min_dy", "<i2"), ("win_w", "<i2"), ("win_h", "<i2"), ("first_cand", "<i4"), ("n_cand", "<i4")])
ME_CAND = np.dtype([("dx", "<i2"), ("dy", "<i2")])
ME_STAGE_JOB = np.dtype([("org_off", "<i4"), ("ref_off", "<i4"), ("width", "<i2"), ("height", "<i2"), ("org_plane", "u1"), ("ref_plane", "u1"), ("i_frac", "u1"), ("filter_mode", "u1"),
                         ("alt_hpel", "u1"), ("func", "u1"), ("base_qx", "i1"), ("base_qy", "i1"), ("mask", "<u2"), ("reserved", "<u2")])
ME_ITEM = np.dtype([("org_off", "<i4"), ("cur_off", "<i4"), ("org_plane", "u1"), ("cur_plane", "u1"), ("func", "u1"), ("sub_shift", "u1"), ("width", "<i2"), ("height", "<i2")])
ME_MASK_ITEM = np.dtype([("org_off", "<i4"), ("cur_off", "<i4"), ("mask_off", "<i4"), ("org_plane", "u1"), ("cur_plane", "u1"), ("mask_plane", "u1"), ("sub_shift", "u1"),
                         ("width", "<i2"), ("height", "<i2"), ("reserved", "<i4")])
assert ME_INT_JOB.itemsize == 32 and ME_STAGE_JOB.itemsize == 24 and ME_ITEM.itemsize == 16 and ME_MASK_ITEM.itemsize == 24


class MePlane(C.Structure):
    _fields_ = [("d_base", C.c_void_p), ("stride", C.c_int32), ("reserved", C.c_int32)]


FAMILY_TO_FUNC = {"SSE": DF["SSE"], "SAD": DF["SAD"], "HAD": DF["HAD"], "HAD_fast": DF["HAD_fast"], "HAD_2SAD": DF["HAD_2SAD"]}
FUNC_SAD_MASK = 5          # (host-side marker only: masked SADs are their own list of the plan)


def _family_codes(df):
    """table index (DFunc, CommonLib/TypeDef.h:339-382) -> C ABI function code"""
    df = df.astype(np.int32)
    out = np.full(df.shape, 255, np.int32)
    out[df < 8] = DF["SSE"]
    out[(df >= 8) & (df < 16)] = DF["SAD"]
    out[(df >= 16) & (df < 24)] = DF["HAD"]
    out[df == 24] = DF["HAD_2SAD"]
    out[df == 25] = FUNC_SAD_MASK
    out[df >= 26] = DF["HAD_fast"]
    return out


def _pow2(v):
    v = v.astype(np.int64)
    return (v > 0) & ((v & (v - 1)) == 0)


def _covered_samples(shape, origin_xy, x0, y0, x1, y1):
    """number of samples of a plane (storage shape, sample (0,0) at origin_xy) covered by the union of the rectangles [x0, x1) x [y0, y1) (arrays, plane coordinates)"""
    if x0.size == 0:
        return 0
    H, W = shape
    ox, oy = origin_xy
    a0, b0 = np.clip(x0.astype(np.int64) + ox, 0, W), np.clip(y0.astype(np.int64) + oy, 0, H)
    a1, b1 = np.clip(x1.astype(np.int64) + ox, 0, W), np.clip(y1.astype(np.int64) + oy, 0, H)
    d = np.zeros((H + 1, W + 1), np.int32)
    np.add.at(d, (b0, a0), 1)
    np.add.at(d, (b1, a1), 1)
    np.add.at(d, (b0, a1), -1)
    np.add.at(d, (b1, a0), -1)
    return int((d.cumsum(0).cumsum(1) > 0).sum())


def _covered_1d(n, off, length):
    """number of elements of a buffer of n covered by the union of the intervals [off, off + length) (sorted, merged: no dense array — a 4K picture's pool is 600 M samples)"""
    if off.size == 0:
        return 0
    a = np.clip(off.astype(np.int64), 0, n)
    b = np.clip(off.astype(np.int64) + length.astype(np.int64), 0, n)
    o = np.argsort(a, kind="stable")
    a, b = a[o], b[o]
    end = np.maximum.accumulate(b)
    prev_end = np.concatenate([[0], end[:-1]])
    return int(np.maximum(end - np.maximum(a, prev_end), 0).sum())


class HostPlane:
    """a picture plane in host memory with the layout the device planes get: `pad` samples of margin, row pitch a multiple of 8 samples"""

    def __init__(self, arr_with_margin, margin, pad):
        h2, w2 = arr_with_margin.shape
        self.width, self.height, self.pad = w2 - 2 * margin, h2 - 2 * margin, pad
        self.stride = ((self.width + 2 * pad + 7) // 8) * 8
        self.storage = np.zeros((self.height + 2 * pad + 1, self.stride), np.int16)          # (+ one row of slack: chunk loads may run past the last row)
        inner = arr_with_margin if margin == pad else np.pad(arr_with_margin[margin:margin + self.height, margin:margin + self.width], pad, mode="edge")
        self.storage[:self.height + 2 * pad, :self.width + 2 * pad] = inner
        self.origin = pad * self.stride + pad


class RecordedLists:
    """one recorded picture as HOST-side lists in the layouts of the C ABI (vvhip_me_* records, TU / DMVR groups) over host planes; RecordedWorkload puts them on the device,
    the tests' reference driver runs them on the CPU"""

    def __init__(self, pic: R.RecordedPicture, bit_depth=10, unique_bytes=True):
        self.pic, self.bit_depth = pic, bit_depth
        # ---- planes: originals get 8 replicated samples of margin (no job reads them; chunk loads may), reconstructions keep theirs
        self.planes = []
        for i in range(pic.planes.size):
            arr, m = pic.plane_array(i)
            self.planes.append(HostPlane(np.asarray(arr), m, m if m else 8))
        self.pool = np.concatenate([np.asarray(pic.pool), np.zeros(256, np.int16)])          # (+ slack)
        self.n_pic_planes = len(self.planes)
        self.pool_index = self.n_pic_planes                     # ONE pool entry, stride 0: a pool block's row pitch is its own width
        assert self.n_pic_planes + 1 <= 16
        self.n_planes = self.n_pic_planes + 1
        strides = np.array([pl.stride for pl in self.planes], np.int64)
        self.dropped = {}

        me, cand, st, d = pic.me, pic.cand, pic.stage, pic.dist
        # ---- motion-estimation jobs: original block = plane 0 at the CU, or a pool pattern (bi-prediction: 2 * org - other prediction)
        pooled = me["patternPool"] >= 0
        me_org_plane = np.where(pooled, self.pool_index, 0)
        me_org_off = np.where(pooled, me["patternPool"].astype(np.int64), me["cuY"].astype(np.int64) * strides[0] + me["cuX"])
        ref_ok = me["refPlane"] >= 3
        me_ref_stride = np.where(ref_ok, strides[np.clip(me["refPlane"], 0, self.n_pic_planes - 1)], 0)
        shape_ok = _pow2(me["w"]) & _pow2(me["h"]) & (me["w"] <= 128) & (me["h"] <= 128)
        # integer candidates: SAD candidates of blocks at least 8 wide go into windows; anything else (the Hadamards of the integer refinement, 4-wide blocks) becomes a plain item
        c_me = cand["me"]
        c_func = _family_codes(cand["df"])
        int_ok = (c_func == DF["SAD"]) & shape_ok[c_me] & (me["w"][c_me] >= 8) & (me["h"][c_me] >= 4) & ref_ok[c_me]
        # per ME: every candidate must share the subShift to form one job (they do: one setDistParam per search); split off the rest
        first_ss = np.zeros(me.size, np.int64)
        if cand.size:
            first_idx = me["firstCand"].clip(0, max(cand.size - 1, 0))
            first_ss = cand["subShift"][first_idx].astype(np.int64)
        int_ok &= cand["subShift"] == first_ss[c_me]
        sel = np.nonzero(int_ok)[0]
        self.cand_index = sel                                            # recorded candidate -> plan candidate (same order)
        pc = np.zeros(sel.size, ME_CAND)
        pc["dx"] = cand["x"][sel] - me["cuX"][c_me[sel]]
        pc["dy"] = cand["y"][sel] - me["cuY"][c_me[sel]]
        # jobs = runs of consecutive selected candidates with the same me
        jobs = np.zeros(0, ME_INT_JOB)
        if sel.size:
            mes = c_me[sel]
            brk = np.nonzero(np.diff(mes) != 0)[0] + 1
            starts = np.concatenate([[0], brk])
            counts = np.diff(np.concatenate([starts, [sel.size]]))
            jm = mes[starts]
            jobs = np.zeros(starts.size, ME_INT_JOB)
            jobs["org_off"] = me_org_off[jm]
            jobs["ref_off"] = me["cuY"][jm].astype(np.int64) * me_ref_stride[jm] + me["cuX"][jm]
            jobs["width"], jobs["height"] = me["w"][jm], me["h"][jm]
            jobs["org_plane"], jobs["ref_plane"] = me_org_plane[jm], me["refPlane"][jm]
            jobs["sub_shift"] = first_ss[jm]
            jobs["first_cand"], jobs["n_cand"] = starts, counts
        self.int_jobs, self.plan_cands = jobs, pc
        self.cand_expected = cand["cost"][sel].copy()

        # ---- refinement stages (any block shape; 4x4 inter blocks do not exist)
        s_me = st["me"]
        s_ok = ref_ok[s_me] & shape_ok[s_me] & ~((me["w"][s_me] == 4) & (me["h"][s_me] == 4)) if st.size else np.zeros(0, bool)
        self.dropped["stages"] = int((~s_ok).sum())
        ssel = np.nonzero(s_ok)[0]
        self.stage_index = ssel
        sj = np.zeros(ssel.size, ME_STAGE_JOB)
        if ssel.size:
            sm = s_me[ssel]
            sj["org_off"] = me_org_off[sm]
            sj["ref_off"] = st["baseY"][ssel].astype(np.int64) * me_ref_stride[sm] + st["baseX"][ssel]
            sj["width"], sj["height"] = me["w"][sm], me["h"][sm]
            sj["org_plane"], sj["ref_plane"] = me_org_plane[sm], me["refPlane"][sm]
            sj["i_frac"], sj["filter_mode"], sj["alt_hpel"] = st["iFrac"][ssel], st["reduceTap"][ssel], st["altHpel"][ssel]
            sj["func"] = np.array([DF["SAD"], DF["HAD"], DF["HAD_fast"]], np.uint8)[st["hadMode"][ssel]]
            sj["base_qx"], sj["base_qy"] = st["baseHor"][ssel], st["baseVer"][ssel]
            ev = st["cost"][ssel] != R.SKIPPED
            sj["mask"] = (ev * (1 << np.arange(9))).sum(1)
            self.stage_expected, self.stage_evaluated = st["cost"][ssel].copy(), ev
        else:
            self.stage_expected, self.stage_evaluated = np.zeros((0, 9), np.uint64), np.zeros((0, 9), bool)
        self.stage_jobs = sj

        # ---- plain table calls: the recorder's `dist` records + the integer candidates that did not fit a window job
        def operand(plane, x, y):
            pl = plane.astype(np.int64)
            is_pool = pl < 0
            st_ = strides[np.clip(pl, 0, self.n_pic_planes - 1)]
            off = np.where(is_pool, x.astype(np.int64), y.astype(np.int64) * st_ + x)
            return np.where(is_pool, self.pool_index, pl), off
        items = np.zeros(d.size, ME_ITEM)
        d_func = _family_codes(d["df"]) if d.size else np.zeros(0, np.int32)
        if d.size:
            po, oo = operand(d["org_plane"], d["org_x"], d["org_y"])
            pcu, oc = operand(d["cur_plane"], d["cur_x"], d["cur_y"])
            items["org_off"], items["cur_off"], items["org_plane"], items["cur_plane"] = oo, oc, po, pcu
            items["func"], items["sub_shift"], items["width"], items["height"] = np.minimum(d_func, 255), d["subShift"], d["w"], d["h"]
        rest = np.nonzero(~int_ok)[0]
        extra = np.zeros(rest.size, ME_ITEM)
        if rest.size:
            rm = c_me[rest]
            extra["org_off"], extra["org_plane"] = me_org_off[rm], me_org_plane[rm]
            extra["cur_off"] = cand["y"][rest].astype(np.int64) * me_ref_stride[rm] + cand["x"][rest]
            extra["cur_plane"], extra["func"], extra["sub_shift"] = me["refPlane"][rm], c_func[rest], cand["subShift"][rest]
            extra["width"], extra["height"] = me["w"][rm], me["h"][rm]
        all_items = np.concatenate([items, extra])
        expected = np.concatenate([d["cost"], cand["cost"][rest]]) if all_items.size else np.zeros(0, np.uint64)
        is_mask = np.concatenate([d_func == FUNC_SAD_MASK, np.zeros(rest.size, bool)]) if all_items.size else np.zeros(0, bool)
        mask_pool = np.concatenate([d["maskPool"].astype(np.int64), np.full(rest.size, -1, np.int64)]) if all_items.size else np.zeros(0, np.int64)
        ok = _pow2(all_items["width"]) & _pow2(all_items["height"]) & (all_items["width"] >= 2) & (all_items["height"] >= 2) & (all_items["width"] <= 128) & (all_items["height"] <= 128) & \
            (all_items["org_plane"] < 16) & (all_items["cur_plane"] < 16) & ((all_items["sub_shift"] == 0) | (all_items["func"] == DF["SAD"]) | is_mask)
        plain = ok & ~is_mask & (all_items["func"] <= 4)
        masked = ok & is_mask & (mask_pool >= 0)
        self.items_dropped = int((~(plain | masked)).sum())
        self.dropped["table_calls"] = self.items_dropped
        # picture order (what a caller that owns a picture's lists hands over; the recorder's order is per worker thread): by the row of the original block, relative to its plane's
        # height; calls between two pool blocks stay behind their predecessor.  me.hip keeps this order inside every (shape, function) class and deals contiguous eighths to the XCDs
        heights = np.array([pl.height for pl in self.planes] + [1], np.float64)
        org_pic = all_items["org_plane"] < self.n_pic_planes
        rowf = np.where(org_pic, (all_items["org_off"].astype(np.int64) // np.append(strides, 1)[np.minimum(all_items["org_plane"], self.n_pic_planes)]) / heights[np.minimum(all_items["org_plane"], self.n_pic_planes)], np.nan)
        last = np.maximum.accumulate(np.where(org_pic, np.arange(all_items.size), -1))
        rowf = np.where(last >= 0, rowf[np.maximum(last, 0)], 0.0) if all_items.size else rowf
        order = np.argsort(rowf, kind="stable")
        all_items, expected, is_mask, mask_pool, plain, masked = all_items[order], expected[order], is_mask[order], mask_pool[order], plain[order], masked[order]
        self.items, self.item_expected = np.ascontiguousarray(all_items[plain]), expected[plain]
        mi = np.zeros(int(masked.sum()), ME_MASK_ITEM)
        if mi.size:
            src = all_items[masked]
            for f in ("org_off", "cur_off", "org_plane", "cur_plane", "sub_shift", "width", "height"):
                mi[f] = src[f]
            mi["mask_off"], mi["mask_plane"] = mask_pool[masked], self.pool_index
        self.mask_items, self.mask_expected = mi, expected[masked]

        # ---- TU lists: one job per (size, transform types); residual blocks live in the pool (pitch = width)
        tu = pic.tu
        self.tu_groups = []
        n_tu_dropped = 0
        if tu.size:
            key = np.stack([tu["w"], tu["h"], tu["trHor"], tu["trVer"]], 1).astype(np.int64)
            uniq, inv = np.unique(key, axis=0, return_inverse=True)
            for g, (w, h, th, tv) in enumerate(uniq):
                sel_t = np.nonzero(inv.ravel() == g)[0]
                if int(w) not in (2, 4, 8, 16, 32, 64) or int(h) not in (2, 4, 8, 16, 32, 64):
                    n_tu_dropped += sel_t.size
                    continue
                qf = np.zeros((sel_t.size, 2), np.int16)
                qf[:, 0] = tu["qp"][sel_t]
                qf[:, 1] = (tu["flags"][sel_t] & 1) | (((tu["flags"][sel_t] >> 1) & 1) << 1)
                self.tu_groups.append(dict(w=int(w), h=int(h), tr_hor=int(th), tr_ver=int(tv), n=int(sel_t.size), index=sel_t, off=tu["pool"][sel_t].astype(np.int32), qf=qf))
        self.dropped["tus"] = int(n_tu_dropped)
        self.tu_coefficients = int(sum(g["n"] * g["w"] * g["h"] for g in self.tu_groups))
        # ---- DMVR: one list per (reference 0, reference 1, sub-block size)
        self.dmvr_groups = []
        dm = pic.dmvr
        if dm.size:
            key = np.stack([dm["ref0Plane"], dm["ref1Plane"], dm["dx"], dm["dy"]], 1).astype(np.int64)
            uniq, inv = np.unique(key, axis=0, return_inverse=True)
            for g, (r0, r1, dx, dy) in enumerate(uniq):
                sel_d = np.nonzero(inv.ravel() == g)[0]
                sel_d = sel_d[np.argsort(dm["y0"][sel_d].astype(np.int64) * strides[r0] + dm["x0"][sel_d], kind="stable")]          # picture order (the recorder's order is per worker thread)
                it = np.zeros(sel_d.size, DMVR_ITEM_DTYPE)
                # recorded (x, y): top-left of the CU's bilinear search area = 2 samples (DMVR_NUM_ITERATION) before the sub-block at the merge vector, which is what the entry point takes
                it["ref0_off"] = (dm["y0"][sel_d].astype(np.int64) + 2) * strides[r0] + dm["x0"][sel_d] + 2
                it["ref1_off"] = (dm["y1"][sel_d].astype(np.int64) + 2) * strides[r1] + dm["x1"][sel_d] + 2
                for f in ("frac0_x", "frac0_y", "frac1_x", "frac1_y"):
                    it[f] = dm[f.replace("_", "")][sel_d]
                self.dmvr_groups.append(dict(r0=int(r0), r1=int(r1), dx=int(dx), dy=int(dy), n=int(sel_d.size), index=sel_d, items=it))
        self.nothing_dropped = not any(self.dropped.values())

        # ---- accounting (SURVEY 8d figures per unit)
        ev_pairs = int((self.stage_evaluated.sum(1) * self.stage_jobs["width"].astype(np.int64) * self.stage_jobs["height"]).sum()) if self.stage_jobs.size else 0
        cand_pairs = int((me["w"][c_me[sel]].astype(np.int64) * me["h"][c_me[sel]]).sum()) if sel.size else 0
        item_pairs = int((self.items["width"].astype(np.int64) * self.items["height"]).sum()) + int((mi["width"].astype(np.int64) * mi["height"]).sum())
        self.pairs = {"integer_candidates": cand_pairs, "subpel_positions": ev_pairs, "table_calls": item_pairs}
        # integer windows (both figures): per scored position 4 w h' + 8 bytes (h' = rows after subShift) over the DISTINCT positions of a job — vvhip_me_plan_create
        # scores a position once however often the search lists it — and "with window reuse" the job's window read once: the rows and columns its candidates touch + the
        # original block + 8 bytes per position.  The window figure is the class's algorithmic bytes (a job IS one block against its window); the per-position figure counts the
        # same samples once per candidate and is a work rate, not memory traffic
        self.alg_bytes_int_all_candidates = self.alg_bytes_int_per_position = self.alg_bytes_int_window = self.int_positions_distinct = 0
        if sel.size:
            ss_c = cand["subShift"][sel].astype(np.int64)
            w_c, h_c = me["w"][c_me[sel]].astype(np.int64), me["h"][c_me[sel]].astype(np.int64)
            per_cand = 4 * w_c * (h_c >> ss_c) + 8
            cj = np.repeat(np.arange(jobs.size), jobs["n_cand"])
            dx, dy = pc["dx"].astype(np.int64), pc["dy"].astype(np.int64)
            _, first = np.unique((cj << 32) | ((dx & 0xffff) << 16) | (dy & 0xffff), return_index=True)
            self.int_positions_distinct = int(first.size)
            self.alg_bytes_int_all_candidates = int(per_cand.sum())
            self.alg_bytes_int_per_position = int(per_cand[first].sum())
            fc = jobs["first_cand"].astype(np.int64)
            nc = jobs["n_cand"].astype(np.int64)
            # (np.*.reduceat over fc: every job must own a non-empty, contiguous, ascending run of candidates — otherwise it silently returns the element AT fc[i])
            if jobs.size and not (nc.min() >= 1 and fc[0] == 0 and np.array_equal(fc[1:], fc[:-1] + nc[:-1]) and fc[-1] + nc[-1] == dx.size):
                raise ValueError("window jobs: candidate runs must be non-empty, contiguous and in job order (first_cand / n_cand)")
            big = 1 << 20
            jw, jh, jss = jobs["width"].astype(np.int64), jobs["height"].astype(np.int64), jobs["sub_shift"].astype(np.int64)
            cols = np.maximum.reduceat(dx, fc) - np.minimum.reduceat(dx, fc) + jw
            rows = np.zeros(jobs.size, np.int64)
            for par in (0, 1):            # subShift 1: a candidate reads every second row from its own dy on -> the even and the odd candidates touch different row sets
                m = (dy & 1) == par
                hi, lo = np.maximum.reduceat(np.where(m, dy, -big), fc), np.minimum.reduceat(np.where(m, dy, big), fc)
                rows += np.where(hi >= lo, np.where(jss > 0, (hi - lo) // 2 + (jh >> jss), 0), 0)
            lo_all, hi_all = np.minimum.reduceat(dy, fc), np.maximum.reduceat(dy, fc)
            rows = np.where(jss > 0, np.minimum(rows, hi_all - lo_all + jh), hi_all - lo_all + jh)
            npos = np.bincount(cj[first], minlength=jobs.size)
            self.alg_bytes_int_window = int((2 * cols * rows + 2 * jw * (jh >> jss) + 8 * npos).sum())
        self.alg_bytes_by_kernel = {"ME_int": self.alg_bytes_int_window, "ME_stage": 0, "ME_item": 0}
        # a sub-pel position reads (w + 3)(h + 3) reference samples with the 4-tap search filter ((w + 7)(h + 7) with 8 taps) + w * h original samples, writes 8 bytes
        if self.stage_jobs.size:
            taps = np.where(self.stage_jobs["filter_mode"] == 2, 3, np.where(self.stage_jobs["filter_mode"] == 1, 5, 7)).astype(np.int64)
            w_, h_ = self.stage_jobs["width"].astype(np.int64), self.stage_jobs["height"].astype(np.int64)
            self.alg_bytes_by_kernel["ME_stage"] = int((self.stage_evaluated.sum(1) * (2 * (w_ + taps) * (h_ + taps) + 2 * w_ * h_ + 8)).sum())
        self.alg_bytes_by_kernel["ME_item"] = int((4 * self.items["width"].astype(np.int64) * (self.items["height"].astype(np.int64) >> self.items["sub_shift"]) + 8).sum()) + \
            int((6 * mi["width"].astype(np.int64) * (mi["height"].astype(np.int64) >> mi["sub_shift"]) + 8).sum())
        self.alg_bytes_me = sum(self.alg_bytes_by_kernel.values())
        # ---- unique bytes per kernel class: the union of everything a class reads (picture-plane rectangles, pool blocks, its records) + what it writes — the floor of its
        #      memory traffic; counter traffic over this figure is the over-fetch (~10x)
        self.unique_bytes_by_kernel = None
        if not unique_bytes:          # (the short runs rocprofv3 wraps skip this accounting: it is host work per picture)
            self.alg_bytes_tu = int(sum(g["n"] * (6 * g["w"] * g["h"] + 24) for g in self.tu_groups))
            self.alg_bytes_dmvr = int(sum(g["n"] * (2 * 2 * (g["dx"] + 5) * (g["dy"] + 5) + 16) for g in self.dmvr_groups))
            return

        def plane_union(pairs):
            """pairs: list of (plane index array, x0, y0, x1, y1) in plane coordinates; pool operands as (offset, length) under plane index pool_index"""
            tot = 0
            for pidx in range(self.n_pic_planes):
                xs = [np.concatenate([p[k][p[0] == pidx] for p in pairs]) if pairs else np.zeros(0, np.int64) for k in range(1, 5)]
                pl = self.planes[pidx]
                tot += 2 * _covered_samples(pl.storage.shape, (pl.pad, pl.pad), *xs)
            return tot

        def xy(off, stride):
            off = off.astype(np.int64)
            y = np.floor_divide(off + 4096 * stride, stride) - 4096          # (offsets into the margin are negative)
            return off - y * stride, y
        st_n = np.append(strides, 1)
        uniq = {}
        # integer windows: the candidates' bounding window per job + the original block
        if jobs.size:
            cj = np.repeat(np.arange(jobs.size), jobs["n_cand"])
            mn = lambda v: np.minimum.reduceat(v, jobs["first_cand"]) if v.size else v
            mx = lambda v: np.maximum.reduceat(v, jobs["first_cand"]) if v.size else v
            rx, ry = xy(jobs["ref_off"], st_n[jobs["ref_plane"]])
            w_, h_ = jobs["width"].astype(np.int64), jobs["height"].astype(np.int64)
            ref_r = (jobs["ref_plane"].astype(np.int64), rx + mn(pc["dx"].astype(np.int64)), ry + mn(pc["dy"].astype(np.int64)), rx + mx(pc["dx"].astype(np.int64)) + w_, ry + mx(pc["dy"].astype(np.int64)) + h_)
            op = jobs["org_plane"].astype(np.int64)
            ox_, oy_ = xy(jobs["org_off"], st_n[np.minimum(op, self.n_pic_planes)])
            org_r = (op, ox_, oy_, ox_ + w_, oy_ + h_)
            pool_sel = op == self.pool_index
            uniq["ME_int"] = plane_union([ref_r, org_r]) + 2 * _covered_1d(self.pool.size, jobs["org_off"][pool_sel], (w_ * h_)[pool_sel]) + 32 * jobs.size + 12 * pc.size
        else:
            uniq["ME_int"] = 0
        if sj.size:
            taps = np.where(sj["filter_mode"] == 2, 4, np.where(sj["filter_mode"] == 1, 6, 8)).astype(np.int64)
            rx, ry = xy(sj["ref_off"], st_n[sj["ref_plane"]])
            w_, h_ = sj["width"].astype(np.int64), sj["height"].astype(np.int64)
            ref_r = (sj["ref_plane"].astype(np.int64), rx - taps // 2, ry - taps // 2, rx + w_ + taps // 2 + 1, ry + h_ + taps // 2 + 1)          # the nine positions' window (offsets -1 .. 1)
            op = sj["org_plane"].astype(np.int64)
            ox_, oy_ = xy(sj["org_off"], st_n[np.minimum(op, self.n_pic_planes)])
            pool_sel = op == self.pool_index
            uniq["ME_stage"] = plane_union([ref_r, (op, ox_, oy_, ox_ + w_, oy_ + h_)]) + 2 * _covered_1d(self.pool.size, sj["org_off"][pool_sel], (w_ * h_)[pool_sel]) + (24 + 72) * sj.size
        else:
            uniq["ME_stage"] = 0
        rects, pool_off, pool_len = [], [], []
        for arr, fields in ((self.items, (("org_plane", "org_off"), ("cur_plane", "cur_off"))), (mi, (("org_plane", "org_off"), ("cur_plane", "cur_off"), ("mask_plane", "mask_off")))):
            if not arr.size:
                continue
            w_, h_ = arr["width"].astype(np.int64), arr["height"].astype(np.int64)
            for fp, fo in fields:
                op = arr[fp].astype(np.int64)
                x_, y_ = xy(arr[fo], st_n[np.minimum(op, self.n_pic_planes)])
                rects.append((op, x_, y_, x_ + w_, y_ + h_))
                sel_p = op == self.pool_index
                pool_off.append(arr[fo][sel_p].astype(np.int64))
                pool_len.append((w_ * h_)[sel_p])
        uniq["ME_item"] = (plane_union(rects) + 2 * _covered_1d(self.pool.size, np.concatenate(pool_off), np.concatenate(pool_len)) if rects else 0) + 24 * self.items.size + 32 * mi.size
        uniq["TU"] = (2 * _covered_1d(self.pool.size, np.concatenate([g["off"].astype(np.int64) for g in self.tu_groups]), np.concatenate([np.full(g["n"], g["w"] * g["h"], np.int64) for g in self.tu_groups]))
                      + int(sum(g["n"] * (4 * g["w"] * g["h"] + 24 + 8) for g in self.tu_groups))) if self.tu_groups else 0
        du = 0
        for g in self.dmvr_groups:
            it = g["items"]
            r0 = (np.full(g["n"], g["r0"], np.int64),) + tuple(v for v in (lambda x_, y_: (x_ - 2, y_ - 2, x_ + g["dx"] + 3, y_ + g["dy"] + 3))(*xy(it["ref0_off"], strides[g["r0"]])))
            r1 = (np.full(g["n"], g["r1"], np.int64),) + tuple(v for v in (lambda x_, y_: (x_ - 2, y_ - 2, x_ + g["dx"] + 3, y_ + g["dy"] + 3))(*xy(it["ref1_off"], strides[g["r1"]])))
            du += plane_union([r0, r1]) + g["n"] * (16 + 16)
        uniq["DMVR"] = du
        self.unique_bytes_by_kernel = {k: int(v) for k, v in uniq.items()}
        self.alg_bytes_tu = int(sum(g["n"] * (6 * g["w"] * g["h"] + 24) for g in self.tu_groups))
        self.alg_bytes_dmvr = int(sum(g["n"] * (2 * 2 * (g["dx"] + 5) * (g["dy"] + 5) + 16) for g in self.dmvr_groups))


class RecordedWorkload:
    """one recorded picture, resident in HBM, ready to replay"""

    def __init__(self, hp: HotPath, pic, max_window=16, bit_depth=10, unique_bytes=True):
        lists = pic if isinstance(pic, RecordedLists) else RecordedLists(pic, bit_depth, unique_bytes)
        self.hp, self.lists, self.pic, self.bit_depth = hp, lists, lists.pic, lists.bit_depth
        dev = hp.device
        for k in ("n_pic_planes", "pool_index", "n_planes", "mask_items", "mask_expected", "dropped", "nothing_dropped", "int_jobs", "plan_cands", "cand_expected", "cand_index", "stage_jobs", "stage_index", "stage_expected", "stage_evaluated", "items",
                  "item_expected", "items_dropped", "tu_coefficients", "pairs", "alg_bytes_me", "alg_bytes_int_all_candidates", "alg_bytes_int_per_position", "alg_bytes_int_window", "int_positions_distinct", "alg_bytes_by_kernel", "alg_bytes_tu", "alg_bytes_dmvr", "unique_bytes_by_kernel"):
            setattr(self, k, getattr(lists, k))
        self.planes = []
        for hpl in lists.planes:
            pl = Plane(dev, hpl.width, hpl.height, hpl.pad)
            assert pl.stride == hpl.stride
            pl.storage = torch.from_numpy(hpl.storage).to(dev)
            self.planes.append(pl)
        self.pool = torch.from_numpy(lists.pool).to(dev)
        tab = (MePlane * 16)()
        for i, pl in enumerate(self.planes):
            tab[i] = MePlane(pl.storage.data_ptr() + 2 * pl.origin, pl.stride, 0)
        tab[self.pool_index] = MePlane(self.pool.data_ptr(), 0, 0)            # stride 0: compact blocks, row pitch = block width
        self.plane_table = tab
        # ---- the plan + result buffers
        self.plan = hp.me_plan_create(self.int_jobs, self.plan_cands, self.stage_jobs, self.items, bit_depth, max_window, mask_items=self.mask_items)
        self.cand_cost = torch.zeros(max(1, self.plan_cands.size), dtype=torch.int64, device=dev)
        self.stage_cost = torch.zeros(max(1, 9 * self.stage_jobs.size), dtype=torch.int64, device=dev)
        self.item_cost = torch.zeros(max(1, self.items.size + self.mask_items.size), dtype=torch.int64, device=dev)
        self.me_info = hp.me_plan_info(self.plan)
        self._me_call = hp.bound("vvhip_me_plan_run", self.plan, C.cast(self.plane_table, C.c_void_p), self.n_planes, C.c_void_p(self.cand_cost.data_ptr()),
                                 C.c_void_p(self.stage_cost.data_ptr()), C.c_void_p(self.item_cost.data_ptr()))

        # ---- TU lists and DMVR lists on the device
        self.tu_groups, tu_jobs, strides_l = [], [], []
        for g in lists.tu_groups:
            n, w, h = g["n"], g["w"], g["h"]
            d = dict(g, d_off=hp.to_device(g["off"]), d_qp=hp.to_device(g["qf"]), level=torch.empty(n * w * h, dtype=torch.int16, device=dev),
                     rec=torch.empty(n * w * h, dtype=torch.int16, device=dev), stats=torch.empty((n, 24), dtype=torch.uint8, device=dev))
            self.tu_groups.append(d)
            tu_jobs.append((w, h, g["tr_hor"], g["tr_ver"], n, 8, d["d_off"], d["d_qp"], d["level"], d["rec"], d["stats"]))
            strides_l.append(w)
        self.tu_table = hp.make_tu_jobs(tu_jobs) if tu_jobs else None
        # sparse outputs (round 6): a TU whose levels are all zero gets only its statistics — xEstimateInterResidualQT reads neither levels nor reconstruction of such a TU
        # (EncoderLib/InterSearch.cpp:3696-3714); $VVHIP_TU_SPARSE=0: the dense contract (every output written)
        self.tu_sparse = os.environ.get("VVHIP_TU_SPARSE", "1") != "0"
        hp.tu_set_sparse_outputs(self.tu_sparse)
        self.tu_strides = (C.c_int32 * max(1, len(strides_l)))(*strides_l)
        self._tu_call = hp.bound("vvhip_tu_rdo_multi_strided", C.c_void_p(self.pool.data_ptr()), C.cast(self.tu_strides, C.c_void_p), bit_depth, self.tu_table[0], self.tu_table[1]) if tu_jobs else None
        self.dmvr_groups = [dict(g, d_items=hp.to_device(g["items"]), out=torch.zeros((g["n"], 16), dtype=torch.uint8, device=dev)) for g in lists.dmvr_groups]
        self._dmvr_calls = [hp.bound("vvhip_dmvr_refine_batch", self.planes[g["r0"]].buf_ptr, self.planes[g["r0"]].stride, self.planes[g["r1"]].buf_ptr, self.planes[g["r1"]].stride,
                                     C.c_void_p(g["d_items"].data_ptr()), g["n"], g["dx"], g["dy"], bit_depth, C.c_void_p(g["out"].data_ptr())) for g in self.dmvr_groups]

    # ------------------------------------------------------------------------------------------------------------------
    def bind_lanes(self, lanes):
        """lanes: HotPath contexts (HotPath.fork) bound to HIP streams -> the picture's launches as pre-bound calls.  Five lanes: refinement stages / integer windows / table calls of
        the motion-search plan, TU lists, DMVR lists (all independent: they share the device); three lanes: the whole plan on lane 0"""
        args = (self.plan, C.cast(self.plane_table, C.c_void_p), self.n_planes, C.c_void_p(self.cand_cost.data_ptr()), C.c_void_p(self.stage_cost.data_ptr()), C.c_void_p(self.item_cost.data_ptr()))
        if len(lanes) >= 5:
            me = [lanes[0].bound("vvhip_me_plan_run_parts", *args, 1), lanes[1].bound("vvhip_me_plan_run_parts", *args, 2), lanes[2].bound("vvhip_me_plan_run_parts", *args, 4)]
            hp1, hp2 = lanes[3], lanes[4]
        else:
            me = [lanes[0].bound("vvhip_me_plan_run", *args)]
            hp1, hp2 = lanes[1], lanes[2]
        hp1.tu_set_sparse_outputs(self.tu_sparse)
        # (round 5: the 64x64 TU lists as a launch of their own on a sixth stream was measured and is SLOWER — step 68.4 -> 77.7 us — the second launch's fixed ~5.5 us and its
        #  queue slot cost more than the long 64-point waves gain from leaving the other sizes' launch; profiles/r05_tu_kernel.md)
        tus = [hp1.bound("vvhip_tu_rdo_multi_strided", C.c_void_p(self.pool.data_ptr()), C.cast(self.tu_strides, C.c_void_p), self.bit_depth, self.tu_table[0], self.tu_table[1])] if self.tu_table else []
        dm = [hp2.bound("vvhip_dmvr_refine_batch", self.planes[g["r0"]].buf_ptr, self.planes[g["r0"]].stride, self.planes[g["r1"]].buf_ptr, self.planes[g["r1"]].stride,
                        C.c_void_p(g["d_items"].data_ptr()), g["n"], g["dx"], g["dy"], self.bit_depth, C.c_void_p(g["out"].data_ptr())) for g in self.dmvr_groups]
        # $VVHIP_BENCH_SKIP_LANES=stage,int,item,tu,dmvr: leave launch groups out (measurement aid: what each group costs the five-stream step; results are then incomplete)
        skip = set(filter(None, os.environ.get("VVHIP_BENCH_SKIP_LANES", "").split(",")))
        if skip:
            if len(me) == 3:
                me = [c for c, name in zip(me, ("stage", "int", "item")) if name not in skip]
            tus = [] if "tu" in skip else tus
            dm = [] if "dmvr" in skip else dm
        self._lane_calls = [c for c in me + tus + dm if c is not None]
        return self._lane_calls

    def run_lanes(self):
        for c in self._lane_calls:
            c()

    def run_me(self):
        self._me_call()

    def update_tu_alg_bytes(self):
        """algorithmic bytes of the TU lists under the sparse-output contract, from the statistics of the last run: a TU with abs_sum == 0 moves its residual in and 24 bytes
        out (2 w h + 24), any other the full 6 w h + 24.  Dense contract: 6 w h + 24 for every TU."""
        from .hotpath import STATS_DTYPE
        torch.cuda.synchronize()
        tot = zero_tus = zero_area = area = 0
        for g in self.tu_groups:
            wh = g["w"] * g["h"]
            nz = int((g["stats"].cpu().numpy().view(STATS_DTYPE).reshape(-1)["abs_sum"] == 0).sum()) if self.tu_sparse else 0
            tot += nz * (2 * wh + 24) + (g["n"] - nz) * (6 * wh + 24)
            zero_tus += nz
            zero_area += nz * wh
            area += g["n"] * wh
        self.alg_bytes_tu = int(tot)
        self.tu_zero_share = {"tus": zero_tus, "area_share": round(zero_area / area, 4) if area else None}
        return self.alg_bytes_tu

    def run_tu(self):
        if self._tu_call is not None:
            self._tu_call()

    def run_dmvr(self):
        for c in self._dmvr_calls:
            c()

    def run(self, timers=None):
        """one pass over the picture's recorded hot-path work on the context's stream"""
        for cls, fn in (("ME", self.run_me), ("TU", self.run_tu), ("DMVR", self.run_dmvr)):
            if timers is not None:
                timers.start(cls)
            fn()
            if timers is not None:
                timers.stop(cls)

    @property
    def distinct_positions(self):
        """integer candidates with a position no earlier candidate of the same window job has: what the plan scores (vvhip_me_plan_create keeps one candidate per distinct
        position of a job; the search lists its start point again and again)"""
        return self.int_positions_distinct

    @property
    def class_launches(self):
        return {"ME": 1, "TU": max(1, len({(g["w"] in (4, 64)) for g in self.tu_groups})), "DMVR": max(1, len(self.dmvr_groups))}

    # ------------------------------------------------------------------------------------------------------------------
    def check_against_recording(self):
        """device results vs what the reference encoder computed when the lists were recorded -> dict of mismatch counts"""
        torch.cuda.synchronize()
        out = {}
        got = self.cand_cost.cpu().numpy().view(np.uint64)[:self.plan_cands.size]
        out["integer_candidates"] = (int(self.plan_cands.size), int((got != self.cand_expected).sum()))
        gs = self.stage_cost.cpu().numpy().view(np.uint64)[:9 * self.stage_jobs.size].reshape(-1, 9)
        out["subpel_positions"] = (int(self.stage_evaluated.sum()), int(((gs != self.stage_expected) & self.stage_evaluated).sum()))
        gi = self.item_cost.cpu().numpy().view(np.uint64)[:self.items.size + self.mask_items.size]
        out["table_calls"] = (int(self.items.size), int((gi[:self.items.size] != self.item_expected).sum()))
        if self.mask_items.size:
            out["masked_sad_calls"] = (int(self.mask_items.size), int((gi[self.items.size:] != self.mask_expected).sum()))
        dm = self.pic.dmvr
        n_d = bad_d = 0
        for g in self.dmvr_groups:
            res = g["out"].cpu().numpy().view(DMVR_RESULT_DTYPE).reshape(-1)
            e = dm[g["index"]]
            bad_d += int(((res["mvd_x"] != e["mvdX"]) | (res["mvd_y"] != e["mvdY"]) | (res["min_cost"] != e["minCost"])).sum())
            n_d += g["n"]
        out["dmvr_subblocks"] = (n_d, bad_d)
        return out
