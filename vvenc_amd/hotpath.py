"""HotPath — Python host layer over the C ABI (include/vvenc_hip.h).

Mirrors the reference's kernel-object interface for this path in batched form:
  RdCost      -> dist_batch / sad_x5_batch / sad_surface     (CommonLib/RdCost.h:117-121)
  TCoeffOps   -> fwd_transform / inv_transform               (CommonLib/TrQuant_EMT.h:63-91 via TrQuant::xT/xIT)
  Quant       -> quant / dequant / need_rdoq / tu_rdo        (CommonLib/Quant.h:143-151)
  MCTF        -> mctf_error_batch / mctf_me_level / mctf_motion_estimation (CommonLib/MCTF.h:160-170)

All tensors are torch CUDA tensors (HBM resident); only raw pointers cross the ABI.  torch is
plumbing here (device memory + streams), nothing is computed with torch ops.
"""
import ctypes as C

import numpy as np
import torch

from .lib import VVHipError, load_library

DF = {"SSE": 0, "SAD": 1, "HAD": 2, "HAD_fast": 3, "HAD_2SAD": 4}
DCT2, DCT8, DST7 = 0, 1, 2

MV_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("error", "<i4"), ("rmsme", "<i4"), ("overlap", "<f8")])
DMVR_ITEM_DTYPE = np.dtype([("ref0_off", "<i4"), ("ref1_off", "<i4"), ("frac0_x", "<i2"), ("frac0_y", "<i2"), ("frac1_x", "<i2"), ("frac1_y", "<i2")])
DMVR_RESULT_DTYPE = np.dtype([("mvd_x", "<i2"), ("mvd_y", "<i2"), ("pad", "<i4"), ("min_cost", "<u8")])
SUBPEL_DTYPE = np.dtype([("org_off", "<i4"), ("ref_off", "<i4"), ("frac_x", "<i2"), ("frac_y", "<i2")])
STATS_DTYPE = np.dtype([("abs_sum", "<i4"), ("last_scan_pos", "<i4"), ("need_rdoq", "<i4"), ("pad", "<i4"), ("sse", "<u8")])


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return C.c_void_p(t.data_ptr())


class Plane:
    """A picture plane in HBM with `pad` samples of margin on every side (int16 samples).

    `buf_ptr` addresses sample (0,0); stride is in samples, like the reference's CPelBuf
    (CommonLib/Buffer.h:149-159).
    """

    def __init__(self, device, width, height, pad=0, stride=None):
        self.width, self.height, self.pad = int(width), int(height), int(pad)
        self.stride = int(stride) if stride else ((self.width + 2 * self.pad + 7) // 8) * 8
        self.storage = torch.zeros((self.height + 2 * self.pad, self.stride), dtype=torch.int16, device=device)

    @classmethod
    def from_numpy(cls, device, arr, pad=0):
        arr = np.ascontiguousarray(arr, np.int16)
        p = cls(device, arr.shape[1], arr.shape[0], pad)
        p.storage[p.pad:p.pad + p.height, p.pad:p.pad + p.width] = torch.from_numpy(arr).to(device)
        return p

    @property
    def origin(self):
        return self.pad * self.stride + self.pad      # element offset of sample (0,0) inside storage

    @property
    def buf_ptr(self):
        return C.c_void_p(self.storage.data_ptr() + 2 * self.origin)

    def offset(self, x, y):
        return y * self.stride + x

    def visible(self):
        return self.storage[self.pad:self.pad + self.height, self.pad:self.pad + self.width]


class HotPath:
    def __init__(self, device=None):
        self.L = load_library()
        if not torch.cuda.is_available():
            raise VVHipError("no GPU visible: vvenc_amd has no CPU fallback (HIP path only)")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        ctx = C.c_void_p()
        rc = self.L.vvhip_create(C.byref(ctx), idx)
        if rc != 0:
            raise VVHipError("vvhip_create failed (%d): %s" % (rc, self.L.vvhip_last_error(None).decode()))
        self.ctx = ctx
        self.use_torch_stream()

    def fork(self, stream=None):
        """a second context on the same device, bound for good to `stream` (a torch.cuda.Stream): launches of independent work lists go to their own streams without any
        per-call stream switching on the host (what the binding's worker threads do with their per-thread contexts).
        stream=None: the lane IS the new context's own stream (wrapped as a torch ExternalStream for events and waits) — no second stream per lane: the runtime deals HIP
        streams onto $GPU_MAX_HW_QUEUES hardware queues in creation order, and a lane that shares its queue with another busy lane is serialized behind it
        (profiles/r06_hw_queues.log)"""
        o = object.__new__(HotPath)
        o.L, o.device = self.L, self.device
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        ctx = C.c_void_p()
        rc = self.L.vvhip_create(C.byref(ctx), idx)
        if rc != 0:
            raise VVHipError("vvhip_create failed (%d): %s" % (rc, self.L.vvhip_last_error(None).decode()))
        o.ctx = ctx
        if stream is None:
            with torch.cuda.device(self.device):
                o.stream = torch.cuda.ExternalStream(int(self.L.vvhip_get_stream(ctx)))
            return o
        o.stream = stream
        o._ck(self.L.vvhip_set_stream(ctx, C.c_void_p(stream.cuda_stream)))
        return o

    def bound(self, fn_name, *args):
        """a zero-argument callable issuing L.<fn_name>(ctx, *args) with the ctypes arguments converted once (per-step launches of fixed work lists)"""
        fn, ctx, a, ck = getattr(self.L, fn_name), self.ctx, args, self._ck

        def call():
            rc = fn(ctx, *a)
            if rc:
                ck(rc)
        return call

    def close(self):
        if getattr(self, "ctx", None):
            self.L.vvhip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing ----
    def _ck(self, rc):
        if rc != 0:
            raise VVHipError("vvenc_hip error %d: %s" % (rc, self.L.vvhip_last_error(self.ctx).decode()))

    def use_torch_stream(self):
        """run on torch's current stream so tensor ops and kernels are ordered"""
        with torch.cuda.device(self.device):
            self._ck(self.L.vvhip_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    # ---- launch graphs: record a fixed sequence of batch calls once, replay it with one hipGraphLaunch ----
    def graph_capture(self, fn):
        """runs fn() under stream capture (fn may only issue batch calls whose scratch already exists) -> graph handle for graph_launch"""
        self.use_own_stream()                        # torch's current stream may be the legacy default stream, which cannot be captured
        try:
            self._ck(self.L.vvhip_graph_begin(self.ctx))
            try:
                fn()
            finally:
                g = C.c_void_p()
                rc = self.L.vvhip_graph_end(self.ctx, C.byref(g))
            self._ck(rc)
        finally:
            self.use_torch_stream()
        return g

    def graph_launch(self, g):
        self._ck(self.L.vvhip_graph_launch(self.ctx, g))

    def graph_destroy(self, g):
        self.L.vvhip_graph_destroy(g)

    def use_own_stream(self):
        self._ck(self.L.vvhip_use_own_stream(self.ctx))

    def stream_handle(self):
        return self.L.vvhip_get_stream(self.ctx)

    def sync(self):
        self._ck(self.L.vvhip_sync(self.ctx))

    def to_device(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype) if dtype is not None else np.ascontiguousarray(arr)
        if a.dtype.fields is not None:
            return torch.from_numpy(a.view(np.uint8).reshape(a.shape + (a.dtype.itemsize,))).to(self.device)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        if a.dtype == np.uint16:
            a = a.view(np.int16)
        return torch.from_numpy(a).to(self.device)

    def plane(self, arr, pad=0, extend=True):
        p = Plane.from_numpy(self.device, arr, pad)
        if pad and extend:
            self.extend_border(p)
        return p

    @staticmethod
    def items(pairs):
        """numpy (n,2) int32 array of (org_off, cur_off)"""
        return np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)

    # ---- (A) distortion ----
    def dist_batch(self, func, org, cur, d_items, n, w, h, sub_shift=0, bit_depth=10, out=None):
        if out is None:
            out = torch.empty(n, dtype=torch.int64, device=self.device)
        self._ck(self.L.vvhip_dist_batch(self.ctx, DF[func] if isinstance(func, str) else func, org.buf_ptr, org.stride,
                                         cur.buf_ptr, cur.stride, w, h, sub_shift, bit_depth, _ptr(d_items), n, _ptr(out)))
        return out

    class _DistJob(C.Structure):
        _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sub_shift", C.c_int32), ("n", C.c_int32), ("d_items", C.c_void_p), ("d_out", C.c_void_p)]

    def make_dist_jobs(self, jobs):
        """jobs: list of (w, h, sub_shift, n, d_items, d_out) -> prepared host job table (keep the tensors alive while it is in use)"""
        arr = (self._DistJob * len(jobs))(*[self._DistJob(w, h, ss, n, it.data_ptr(), out.data_ptr()) for (w, h, ss, n, it, out) in jobs])
        return arr, len(jobs), jobs

    def dist_multi(self, func, org, cur, jobs, bit_depth=10):
        """one merged launch per function where the kernels allow it; `jobs` = list of tuples or a table from make_dist_jobs"""
        if isinstance(jobs, list):
            jobs = self.make_dist_jobs(jobs)
        self._ck(self.L.vvhip_dist_multi(self.ctx, DF[func] if isinstance(func, str) else func, org.buf_ptr, org.stride, cur.buf_ptr, cur.stride,
                                         bit_depth, jobs[0], jobs[1]))

    class _DistFJob(C.Structure):
        _fields_ = [("func", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("sub_shift", C.c_int32), ("n", C.c_int32), ("flags", C.c_int32),
                    ("d_items", C.c_void_p), ("d_out", C.c_void_p)]

    DIST_FLAG_SAMPLES = 1      # VVHIP_DIST_FLAG_SAMPLES: both operands are samples in [0, 2^bit_depth)

    def make_dist_fjobs(self, jobs, flags=0):
        """jobs: list of (func, w, h, sub_shift, n, d_items, d_out) -> prepared host job table for dist_multi_func"""
        arr = (self._DistFJob * len(jobs))(*[self._DistFJob(DF[f] if isinstance(f, str) else f, w, h, ss, n, flags, it.data_ptr(), out.data_ptr()) for (f, w, h, ss, n, it, out) in jobs])
        return arr, len(jobs), jobs

    def dist_multi_func(self, org, cur, jobs, bit_depth=10):
        """a frame's distortion lists with a function per job; SAD+SSE and HAD+HAD_fast jobs share launches"""
        if isinstance(jobs, list):
            jobs = self.make_dist_fjobs(jobs)
        self._ck(self.L.vvhip_dist_multi_func(self.ctx, org.buf_ptr, org.stride, cur.buf_ptr, cur.stride, bit_depth, jobs[0], jobs[1]))

    # ---- 8x8-tiled copies of planes (one 128-byte cache line = one 8x8 tile): what the 8x8 SAD / SSE lists and every Hadamard list read on the MI355X
    class _TiledPlanes(C.Structure):
        _fields_ = [("d_org_tiled", C.c_void_p), ("d_cur_tiled", C.c_void_p), ("org_margin", C.c_int32), ("cur_margin", C.c_int32), ("d_cur_shift1", C.c_void_p)]

    def tile_plane(self, plane, out=None):
        """tiled copy of a whole padded Plane (re-tile after the plane changes); returns the int16 tensor holding it"""
        rows = plane.storage.shape[0]
        n = int(self.L.vvhip_tiled8_elems(plane.stride, rows))
        if out is None:
            out = torch.empty(n, dtype=torch.int16, device=self.device)
        self._ck(self.L.vvhip_plane_tile8(self.ctx, plane.storage.data_ptr(), plane.stride, rows, out.data_ptr()))
        return out

    def shift_plane(self, plane, out=None):
        """copy of a whole padded Plane shifted by one sample (out[k] == storage[k + 1]): odd sample addresses become dword-aligned loads; re-make after the plane changes"""
        if out is None:
            out = torch.empty_like(plane.storage)
        self._ck(self.L.vvhip_plane_shift1(self.ctx, plane.storage.data_ptr(), plane.storage.numel(), out.data_ptr()))
        return out

    def planes_derive(self, org, cur, org_tiled=None, cur_tiled=None, cur_shift=None):
        """the derived copies of a picture's two planes in one launch (outputs that are None are skipped)"""
        self._ck(self.L.vvhip_planes_derive(self.ctx, org.storage.data_ptr(), org.stride, org.storage.shape[0], org_tiled.data_ptr() if org_tiled is not None else None,
                                            cur.storage.data_ptr(), cur.stride, cur.storage.shape[0], cur_tiled.data_ptr() if cur_tiled is not None else None,
                                            cur_shift.data_ptr() if cur_shift is not None else None))

    def dist_multi_func_tiled(self, org, cur, org_tiled, cur_tiled, jobs, bit_depth=10, cur_shift=None):
        """dist_multi_func with the tiled copies of both planes (either may be None) and the one-sample-shifted copy of the reference plane at hand (identical results)"""
        if isinstance(jobs, list):
            jobs = self.make_dist_fjobs(jobs)
        t = self._TiledPlanes(org_tiled.data_ptr() if org_tiled is not None else None, cur_tiled.data_ptr() if cur_tiled is not None else None, org.pad, cur.pad,
                              cur_shift.data_ptr() + 2 * cur.origin if cur_shift is not None else None)
        self._ck(self.L.vvhip_dist_multi_func_tiled(self.ctx, org.buf_ptr, org.stride, cur.buf_ptr, cur.stride, C.byref(t), bit_depth, jobs[0], jobs[1]))

    def sad_x5_batch(self, org, cur, d_items, n, w, h, sub_shift=1, calc_centre=True, out=None):
        if out is None:
            out = torch.zeros(n * 5, dtype=torch.int64, device=self.device)
        self._ck(self.L.vvhip_sad_x5_batch(self.ctx, org.buf_ptr, org.stride, cur.buf_ptr, cur.stride, w, h, sub_shift,
                                           int(calc_centre), _ptr(d_items), n, _ptr(out)))
        return out

    def sad_mask_batch(self, org, cur, mask, step_x, mask_stride2, d_items, d_mask_off, n, w, h, sub_shift=0, bit_depth=10, out=None):
        """GEO masked SAD (DF_SAD_WITH_MASK); `mask` is a Plane, d_mask_off per-candidate first-sample offsets (or None)"""
        if out is None:
            out = torch.zeros(n, dtype=torch.int64, device=self.device)
        self._ck(self.L.vvhip_sad_mask_batch(self.ctx, org.buf_ptr, org.stride, cur.buf_ptr, cur.stride, mask.buf_ptr, mask.stride, step_x, mask_stride2,
                                             w, h, sub_shift, bit_depth, _ptr(d_items), _ptr(d_mask_off) if d_mask_off is not None else None, n, _ptr(out)))
        return out

    def fix_weighted_sse_batch(self, org, cur, d_items, d_weights, n, w, h, bit_depth=10, out=None):
        """fixed-weight SSE (RdCost::m_fxdWtdPredPtr); d_weights: uint32 per candidate (stored in an int32 tensor)"""
        if out is None:
            out = torch.zeros(n, dtype=torch.int64, device=self.device)
        self._ck(self.L.vvhip_fix_weighted_sse_batch(self.ctx, org.buf_ptr, org.stride, cur.buf_ptr, cur.stride, w, h, bit_depth,
                                                     _ptr(d_items), _ptr(d_weights), n, _ptr(out)))
        return out

    def sad_surface(self, org, ref, d_org_off, d_ref_off, n_blocks, w, h, sub_shift, range_x, range_y, out=None):
        if out is None:
            out = torch.empty(n_blocks * (2 * range_x + 1) * (2 * range_y + 1), dtype=torch.int32, device=self.device)
        self._ck(self.L.vvhip_sad_surface(self.ctx, org.buf_ptr, org.stride, ref.buf_ptr, ref.stride, w, h, sub_shift,
                                          range_x, range_y, _ptr(d_org_off), _ptr(d_ref_off), n_blocks, _ptr(out)))
        return out

    # ---- (B) transform / quant ----
    def fwd_transform(self, resi, d_off, n, w, h, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10, out=None):
        if out is None:
            out = torch.empty(n * w * h, dtype=torch.int32, device=self.device)
        self._ck(self.L.vvhip_fwd_transform_batch(self.ctx, resi.buf_ptr, resi.stride, _ptr(d_off), n, w, h, tr_hor, tr_ver, bit_depth, _ptr(out)))
        return out

    def inv_transform(self, d_coef, n, w, h, resi_out, d_off, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10):
        self._ck(self.L.vvhip_inv_transform_batch(self.ctx, _ptr(d_coef), n, w, h, tr_hor, tr_ver, bit_depth,
                                                  resi_out.buf_ptr, resi_out.stride, _ptr(d_off)))
        return resi_out

    @staticmethod
    def tu_qp(qps, irap=0, luma=1):
        a = np.zeros((len(qps), 2), np.int16)
        a[:, 0] = qps
        a[:, 1] = (np.asarray(irap, np.int16) & 1) | ((np.asarray(luma, np.int16) & 1) << 1)
        return a

    def quant(self, d_coef, n, w, h, d_qp, bit_depth=10, thr_val=8, want_delta_u=True):
        level = torch.empty(n * w * h, dtype=torch.int16, device=self.device)
        du = torch.zeros(n * w * h, dtype=torch.int32, device=self.device) if want_delta_u else None
        s = torch.empty(n, dtype=torch.int32, device=self.device)
        last = torch.empty(n, dtype=torch.int32, device=self.device)
        self._ck(self.L.vvhip_quant_batch(self.ctx, _ptr(d_coef), n, w, h, bit_depth, _ptr(d_qp), thr_val, _ptr(level), _ptr(du), _ptr(s), _ptr(last)))
        return level, du, s, last

    def dequant(self, d_level, n, w, h, d_qp, bit_depth=10):
        out = torch.empty(n * w * h, dtype=torch.int32, device=self.device)
        self._ck(self.L.vvhip_dequant_batch(self.ctx, _ptr(d_level), n, w, h, bit_depth, _ptr(d_qp), _ptr(out)))
        return out

    def need_rdoq(self, d_coef, n, w, h, d_qp, bit_depth=10):
        out = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._ck(self.L.vvhip_need_rdoq_batch(self.ctx, _ptr(d_coef), n, w, h, bit_depth, _ptr(d_qp), _ptr(out)))
        return out

    def tu_rdo(self, resi, d_off, n, w, h, d_qp, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10, thr_val=8, level=None, rec=None, stats=None):
        if level is None:
            level = torch.empty(n * w * h, dtype=torch.int16, device=self.device)
        if rec is None:
            rec = torch.empty(n * w * h, dtype=torch.int16, device=self.device)
        if stats is None:
            stats = torch.empty((n, STATS_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        self._ck(self.L.vvhip_tu_rdo_batch(self.ctx, resi.buf_ptr, resi.stride, _ptr(d_off), n, w, h, tr_hor, tr_ver, bit_depth,
                                           _ptr(d_qp), thr_val, _ptr(level), _ptr(rec), _ptr(stats)))
        return level, rec, stats

    class _TuJob(C.Structure):
        _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tr_hor", C.c_int32), ("tr_ver", C.c_int32), ("n", C.c_int32), ("thr_val", C.c_int32),
                    ("d_resi_off", C.c_void_p), ("d_qp", C.c_void_p), ("d_level", C.c_void_p), ("d_rec_resi", C.c_void_p), ("d_stats", C.c_void_p)]

    def make_tu_jobs(self, jobs):
        """jobs: list of (w, h, tr_hor, tr_ver, n, thr_val, d_off, d_qp, level, rec, stats) -> prepared host job table (tensors must stay alive)"""
        arr = (self._TuJob * len(jobs))(*[self._TuJob(w, h, th, tv, n, thr, off.data_ptr(), qp.data_ptr(),
                                                      lv.data_ptr() if lv is not None else None, rc.data_ptr() if rc is not None else None,
                                                      st.data_ptr() if st is not None else None)
                                          for (w, h, th, tv, n, thr, off, qp, lv, rc, st) in jobs])
        return arr, len(jobs), jobs

    def tu_rdo_multi(self, resi, jobs, bit_depth=10):
        """all TU lists of a residual plane in one call (square 8/16/32 sizes share one launch)"""
        if isinstance(jobs, list):
            jobs = self.make_tu_jobs(jobs)
        self._ck(self.L.vvhip_tu_rdo_multi(self.ctx, resi.buf_ptr, resi.stride, bit_depth, jobs[0], jobs[1]))

    def tu_rdo_multi_strided(self, d_resi, strides, jobs, bit_depth=10):
        """the same with a residual row pitch per job (compact per-TU blocks in one buffer: pitch = width)"""
        if isinstance(jobs, list):
            jobs = self.make_tu_jobs(jobs)
        arr = (C.c_int32 * len(strides))(*strides)
        self._ck(self.L.vvhip_tu_rdo_multi_strided(self.ctx, _ptr(d_resi), C.cast(arr, C.c_void_p), bit_depth, jobs[0], jobs[1]))

    # ---- motion-search plans: integer candidates + sub-pel stages + plain table calls of a picture in one launch ----
    def me_plan_create(self, int_jobs, cands, stage_jobs, items, bit_depth=10, max_window=0, mask_items=None):
        """numpy record arrays (vvenc_amd/replay.py dtypes mirror vvhip_me_int_job / _cand / _stage_job / _item / _mask_item) -> plan handle"""
        keep = [np.ascontiguousarray(a) for a in (int_jobs, cands, stage_jobs, items)]
        plan = C.c_void_p()
        if mask_items is None or not mask_items.size:
            self._ck(self.L.vvhip_me_plan_create(self.ctx, keep[0].ctypes.data if keep[0].size else None, int(keep[0].size), keep[1].ctypes.data if keep[1].size else None, int(keep[1].size),
                                                 keep[2].ctypes.data if keep[2].size else None, int(keep[2].size), keep[3].ctypes.data if keep[3].size else None, int(keep[3].size),
                                                 bit_depth, max_window, C.byref(plan)))
            return plan
        keep.append(np.ascontiguousarray(mask_items))

        class Lists(C.Structure):          # vvhip_me_lists
            _fields_ = [(n, t) for k in range(5) for n, t in (("p%d" % k, C.c_void_p), ("n%d" % k, C.c_int32))]
        L = Lists()
        for k, a in enumerate(keep):
            setattr(L, "p%d" % k, a.ctypes.data if a.size else None)
            setattr(L, "n%d" % k, int(a.size))
        self._ck(self.L.vvhip_me_plan_create_lists(self.ctx, C.cast(C.pointer(L), C.c_void_p), bit_depth, max_window, C.byref(plan)))
        return plan

    def me_plan_destroy(self, plan):
        self.L.vvhip_me_plan_destroy(self.ctx, plan)

    def me_plan_info(self, plan):
        v = [C.c_int() for _ in range(4)]
        self._ck(self.L.vvhip_me_plan_info(plan, *[C.cast(C.pointer(x), C.c_void_p) for x in v]))
        return {"waves_int": v[0].value, "waves_stage": v[1].value, "waves_item": v[2].value, "lds_bytes": v[3].value}

    def me_plan_set_timing(self, plan, on=True):
        self._ck(self.L.vvhip_me_plan_set_timing(self.ctx, plan, 1 if on else 0))

    def me_plan_last_times(self, plan):
        """ms of the last run's parts: refinement stages, integer windows (one launch for both LDS classes), 0, table calls"""
        ms = (C.c_float * 4)()
        self._ck(self.L.vvhip_me_plan_last_times(self.ctx, plan, C.cast(ms, C.c_void_p)))
        return [float(x) for x in ms]

    def me_plan_run(self, plan, plane_table, n_planes, cand_cost, stage_cost, item_cost):
        self._ck(self.L.vvhip_me_plan_run(self.ctx, plan, C.cast(plane_table, C.c_void_p), n_planes, _ptr(cand_cost), _ptr(stage_cost), _ptr(item_cost)))

    # ---- DMVR refinement search ----
    def dmvr_refine_batch(self, ref0, ref1, d_items, n, dx, dy, bit_depth=10, out=None):
        """d_items: DMVR_ITEM_DTYPE records -> tensor of DMVR_RESULT_DTYPE records (as uint8 rows)"""
        if out is None:
            out = torch.empty((n, DMVR_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        self._ck(self.L.vvhip_dmvr_refine_batch(self.ctx, ref0.buf_ptr, ref0.stride, ref1.buf_ptr, ref1.stride, _ptr(d_items), n, dx, dy, bit_depth, _ptr(out)))
        return out

    # ---- ALF encoder statistics ----
    ALF_REC = 183

    def alf_classify(self, rec, bit_depth=10, vb_ctu_height=128, vb_pos=124, out=None):
        """rec: luma Plane with a replicated margin >= 4 -> uint8 tensor (H/4, W/4, 2) {classIdx, transposeIdx}"""
        if out is None:
            out = torch.empty((rec.height // 4, rec.width // 4, 2), dtype=torch.uint8, device=self.device)
        self._ck(self.L.vvhip_alf_classify(self.ctx, rec.buf_ptr, rec.stride, rec.width, rec.height, bit_depth, vb_ctu_height, vb_pos, _ptr(out)))
        return out

    def alf_stats_plane(self, org, rec, ctu_size, filter_length, d_cls=None, vb_ctu_height=128, vb_pos=124, out=None, init=None, ctu_in_unit=None):
        """covariance records of every CTU of a plane -> float32 tensor (numCtus, 25 or 1, ALF_REC): E[13][13], y[13], pixAcc"""
        nctu = ((rec.width + ctu_size - 1) // ctu_size) * ((rec.height + ctu_size - 1) // ctu_size)
        ncls = 25 if d_cls is not None else 1
        if out is None:
            out = torch.empty((nctu, ncls, self.ALF_REC), dtype=torch.float32, device=self.device)
        self._ck(self.L.vvhip_alf_stats_plane_units(self.ctx, org.buf_ptr, org.stride, rec.buf_ptr, rec.stride, rec.width, rec.height, ctu_size, ctu_in_unit or ctu_size, filter_length,
                                              _ptr(d_cls) if d_cls is not None else None, vb_ctu_height, vb_pos, _ptr(init) if init is not None else None, _ptr(out)))
        return out

    def ccalf_stats_plane(self, org_c, slf_c, rec_luma, ctu_size_c, vb_ctu_height=128, vb_pos=124, shift=(1, 1), out=None, init=None):
        """CC-ALF covariance records of every chroma CTU -> float32 tensor (numCtus, ALF_REC); org_c / slf_c chroma Planes, rec_luma with margin >= 2"""
        nctu = ((slf_c.width + ctu_size_c - 1) // ctu_size_c) * ((slf_c.height + ctu_size_c - 1) // ctu_size_c)
        if out is None:
            out = torch.empty((nctu, self.ALF_REC), dtype=torch.float32, device=self.device)
        self._ck(self.L.vvhip_ccalf_stats_plane(self.ctx, org_c.buf_ptr, org_c.stride, slf_c.buf_ptr, slf_c.stride, rec_luma.buf_ptr, rec_luma.stride, slf_c.width, slf_c.height,
                                                ctu_size_c, shift[0], shift[1], vb_ctu_height, vb_pos, rec_luma.height, _ptr(init) if init is not None else None, _ptr(out)))
        return out

    def alf_filter_plane(self, src, dst, ctu_size, bit_depth, filter_length, d_coeff, d_clip, d_ctu_set, d_cls=None, vb_ctu_height=128, vb_pos=124):
        """filterBlk over the enabled CTUs: src Plane (replicated margin >= 4) -> dst Plane (CTUs with d_ctu_set < 0 keep their samples).
        d_coeff / d_clip: int16 (numSets, 25 or 1, 13); d_clip None = linear filters; d_ctu_set: int16 per CTU"""
        self._ck(self.L.vvhip_alf_filter_plane(self.ctx, src.buf_ptr, src.stride, dst.buf_ptr, dst.stride, src.width, src.height, ctu_size, bit_depth, filter_length,
                                               _ptr(d_cls) if d_cls is not None else None, _ptr(d_coeff), _ptr(d_clip) if d_clip is not None else None, _ptr(d_ctu_set), vb_ctu_height, vb_pos))
        return dst

    def ccalf_filter_plane(self, dst_c, rec_luma, ctu_size_c, bit_depth, d_coeff, d_ctu_filter, vb_ctu_height=128, vb_pos=124, shift=(1, 1)):
        """filterBlkCcAlf: corrects the chroma Plane dst_c in place from rec_luma (margin >= 2).  d_coeff: int16 (numFilters, 8); d_ctu_filter: uint8 per CTU (0 = off)"""
        self._ck(self.L.vvhip_ccalf_filter_plane(self.ctx, dst_c.buf_ptr, dst_c.stride, rec_luma.buf_ptr, rec_luma.stride, dst_c.width, dst_c.height, ctu_size_c, shift[0], shift[1], bit_depth,
                                                 _ptr(d_coeff), _ptr(d_ctu_filter), vb_ctu_height, vb_pos))
        return dst_c

    # ---- MCTF apply side ----
    REF_STRENGTHS = ((0.84375, 0.6, 0.4286, 0.3333, 0.2727, 0.2308), (1.12500, 1.0, 0.7143, 0.5556, 0.4545, 0.3846))      # MCTF.cpp:112-117

    def mctf_filter_params(self, qp, bit_depth, overall_strength, chroma):
        s, w = C.c_double(), C.c_double()
        self._ck(self.L.vvhip_mctf_filter_params(qp, bit_depth, overall_strength, int(chroma), C.byref(s), C.byref(w)))
        return s.value, w.value

    def mctf_apply_plane(self, org, refs, d_mvs, mv_w, chroma_shift, ref_strengths, weight_scaling, sigma_sq, bit_depth=10, unit=16, low_res=True, qp=32, out=None):
        """bilateral temporal filter of one plane: org / refs are Planes (same stride), d_mvs a list of device motion fields (MV_DTYPE records)"""
        if out is None:
            out = Plane(self.device, org.width, org.height, 0)
        assert all(r.stride == refs[0].stride for r in refs)
        rp = (C.c_void_p * len(refs))(*[r.buf_ptr.value for r in refs])
        mp = (C.c_void_p * len(refs))(*[m.data_ptr() for m in d_mvs])
        rs = (C.c_double * len(refs))(*[float(x) for x in ref_strengths])
        self._ck(self.L.vvhip_mctf_apply_plane(self.ctx, org.buf_ptr, org.stride, org.width, org.height, chroma_shift, bit_depth, unit, int(low_res), qp, len(refs),
                                               rp, refs[0].stride, mp, mv_w, rs, weight_scaling, sigma_sq, out.buf_ptr, out.stride))
        return out

    def mctf_bilateral(self, org_yuv, refs_yuv, mvs_np, ref_index, bit_depth=10, qp=32, unit=16, low_res=True, pic_reordering=True, overall_strength=0.95):
        """MCTF::bilateralFilter on 4:2:0 numpy planes (uploads with the reference's margins); returns numpy (Y, U, V)"""
        h, w = org_yuv[0].shape
        mv_w = (w + unit - 1) // unit
        d_mvs = [self.to_device(np.ascontiguousarray(m)) for m in mvs_np]
        strengths = [self.REF_STRENGTHS[0 if pic_reordering else 1][k] for k in ref_index]
        outs = []
        for c in range(3):
            cs = 1 if c else 0
            po = self.plane(org_yuv[c], 128 >> cs)
            prs = [self.plane(r[c], 128 >> cs) for r in refs_yuv]
            sigma, scaling = self.mctf_filter_params(qp, bit_depth, overall_strength, c > 0)
            o = self.mctf_apply_plane(po, prs, d_mvs, mv_w, cs, strengths, scaling, sigma, bit_depth, unit, low_res, qp)
            outs.append(o.visible().cpu().numpy())
        return tuple(outs)

    # ---- sub-pel interpolation ----
    def if_filter(self, taps, vertical, first, last, bit_depth, d_src, src_off, src_stride, d_dst, dst_off, dst_stride, w, h, coeff):
        """one pass on one block with the caller's taps (InterpolationFilter::m_filterHor / m_filterVer slot)"""
        c = np.ascontiguousarray(coeff, np.int16)
        self._ck(self.L.vvhip_if_filter(self.ctx, taps, int(vertical), int(first), int(last), bit_depth, C.c_void_p(d_src.data_ptr() + 2 * src_off), src_stride,
                                        C.c_void_p(d_dst.data_ptr() + 2 * dst_off), dst_stride, w, h, c.ctypes.data_as(C.c_void_p)))

    def if_copy(self, first, last, bit_depth, d_src, src_off, src_stride, d_dst, dst_off, dst_stride, w, h, bi_mc=False):
        self._ck(self.L.vvhip_if_copy(self.ctx, int(first), int(last), bit_depth, C.c_void_p(d_src.data_ptr() + 2 * src_off), src_stride,
                                      C.c_void_p(d_dst.data_ptr() + 2 * dst_off), dst_stride, w, h, int(bi_mc)))

    def interp_luma_batch(self, ref, d_items, n, w, h, bit_depth=10, rnd_res=True, filter_mode=0, use_alt_hpel=False, out=None):
        """motion-compensated luma prediction blocks at 1/16-sample vectors (d_items: SUBPEL_DTYPE records) -> (n, h, w) int16"""
        if out is None:
            out = torch.empty(n * w * h, dtype=torch.int16, device=self.device)
        self._ck(self.L.vvhip_interp_luma_batch(self.ctx, ref.buf_ptr, ref.stride, _ptr(d_items), n, w, h, bit_depth, int(rnd_res), filter_mode,
                                                int(use_alt_hpel), _ptr(out)))
        return out

    def subpel_dist_batch(self, func, org, ref, d_items, n, w, h, bit_depth=10, filter_mode=0, use_alt_hpel=False, out=None):
        """distortion of sub-pel candidates: interpolate at the fractional vector, score against the original block"""
        if out is None:
            out = torch.empty(n, dtype=torch.int64, device=self.device)
        self._ck(self.L.vvhip_subpel_dist_batch(self.ctx, DF[func] if isinstance(func, str) else func, org.buf_ptr, org.stride, ref.buf_ptr, ref.stride,
                                                w, h, bit_depth, filter_mode, int(use_alt_hpel), _ptr(d_items), n, _ptr(out)))
        return out

    def subpel_refine_batch(self, func, org, ref, d_bases, n_blocks, offsets, w, h, bit_depth=10, filter_mode=0, use_alt_hpel=False, out=None):
        """one xPatternRefinement stage for many blocks: offsets = [(dx, dy), ...] in 1/16 sample around each block's base vector -> (n_blocks, len(offsets)) costs"""
        offs = np.ascontiguousarray(offsets, np.int16).reshape(-1, 2)
        if out is None:
            out = torch.empty(n_blocks * offs.shape[0], dtype=torch.int64, device=self.device)
        self._ck(self.L.vvhip_subpel_refine_batch(self.ctx, DF[func] if isinstance(func, str) else func, org.buf_ptr, org.stride, ref.buf_ptr, ref.stride, w, h, bit_depth,
                                                  filter_mode, int(use_alt_hpel), _ptr(d_bases), n_blocks, offs.ctypes.data_as(C.c_void_p), offs.shape[0], _ptr(out)))
        return out

    # ---- (C) MCTF ----
    def extend_border(self, plane):
        self._ck(self.L.vvhip_extend_border(self.ctx, plane.buf_ptr, plane.stride, plane.width, plane.height, plane.pad))

    def mctf_subsample(self, src, pad=128):
        dst = Plane(self.device, src.width // 2, src.height // 2, pad)
        self._ck(self.L.vvhip_mctf_subsample(self.ctx, src.buf_ptr, src.stride, src.width, src.height, dst.buf_ptr, dst.stride, pad))
        return dst

    def mctf_error_batch(self, org, buf, d_items, n, w, h, tap4, bit_depth=10):
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        self._ck(self.L.vvhip_mctf_error_batch(self.ctx, org.buf_ptr, org.stride, buf.buf_ptr, buf.stride, w, h, int(tap4), bit_depth, _ptr(d_items), n, _ptr(out)))
        return out

    def mctf_calc_var_batch(self, org, d_off, n, w, h):
        out = torch.empty(n, dtype=torch.int64, device=self.device)
        self._ck(self.L.vvhip_mctf_calc_var_batch(self.ctx, org.buf_ptr, org.stride, w, h, _ptr(d_off), n, _ptr(out)))
        return out

    def new_mv_field(self, w, h):
        t = torch.empty((h * w, MV_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        self._ck(self.L.vvhip_mctf_init_mvs(self.ctx, _ptr(t), w * h))
        return t

    def mctf_me_level(self, org, buf, block_size, prev, prev_dims, factor, double_res, mvs, mvs_dims, search_pattern=2, low_res=1, bit_depth=10, unit=16):
        pw, ph = prev_dims if prev is not None else (0, 0)
        self._ck(self.L.vvhip_mctf_me_level(self.ctx, org.buf_ptr, org.stride, buf.buf_ptr, buf.stride, org.width, org.height, block_size,
                                            _ptr(prev), pw, ph, factor, int(double_res), search_pattern, int(low_res), bit_depth, unit,
                                            _ptr(mvs), mvs_dims[0], mvs_dims[1]))
        return mvs

    def mctf_motion_estimation(self, cur, refs, bit_depth=10, unit=16, speed=4, add_level=None, out=None, wait=True):
        """cur / refs: Plane objects with identical geometry and pad >= 128 (borders extended).  wait=False: queued on the context's stream, no host wait
        (vvhip_mctf_motion_estimation_async)."""
        if add_level is None:
            add_level = cur.width >= 1920            # MCTF.cpp:768
        ow, oh = (cur.width + unit - 1) // unit, (cur.height + unit - 1) // unit
        if out is None:
            out = [torch.empty((ow * oh, MV_DTYPE.itemsize), dtype=torch.uint8, device=self.device) for _ in refs]
        ref_ptrs = (C.c_void_p * len(refs))(*[r.buf_ptr.value for r in refs])
        out_ptrs = (C.c_void_p * len(refs))(*[o.data_ptr() for o in out])
        for r in refs:
            assert (r.width, r.height, r.stride, r.pad) == (cur.width, cur.height, cur.stride, cur.pad)
        fn = self.L.vvhip_mctf_motion_estimation if wait else self.L.vvhip_mctf_motion_estimation_async
        self._ck(fn(self.ctx, cur.buf_ptr, ref_ptrs, len(refs), cur.stride, cur.width, cur.height, cur.pad, bit_depth, unit, speed, int(bool(add_level)), out_ptrs))
        return out, (ow, oh)

    def mctf_set_stats(self, on):
        self._ck(self.L.vvhip_mctf_set_stats(self.ctx, int(bool(on))))

    def quant_core(self, coef, quant_coeff, q_bits, add, thr_val=8, lfnst_idx=0):
        """QuantCore's own argument list on ONE block (numpy int32 h x w) -> (levels h x w int16, deltaU w*h int32, abs sum, last scan position); lfnst_idx > 0: the
        first-coefficient-group rule of LFNST TUs (vvhip_quant_core_lfnst)"""
        coef = np.ascontiguousarray(coef, np.int32)
        h, w = coef.shape
        d_c = self.to_device(coef.reshape(-1))
        d_l = torch.full((h * w,), 0x7777, dtype=torch.int16, device=self.device)
        d_u = torch.zeros(h * w, dtype=torch.int32, device=self.device)
        d_s = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._ck(self.L.vvhip_quant_core_lfnst(self.ctx, _ptr(d_c), w, h, quant_coeff, q_bits, add, thr_val, int(lfnst_idx), _ptr(d_l), _ptr(d_u), C.c_void_p(d_s.data_ptr()),
                                               C.c_void_p(d_s.data_ptr() + 4)))
        torch.cuda.synchronize()
        sv = d_s.cpu().numpy()
        return d_l.cpu().numpy().reshape(h, w), d_u.cpu().numpy(), int(sv[0]), int(sv[1])

    def tu_set_sparse_outputs(self, on):
        """vvhip_tu_set_sparse_outputs: the fused TU launches write no levels / reconstruction for TUs whose levels are all zero (the caller treats them as zero)"""
        self._ck(self.L.vvhip_tu_set_sparse_outputs(self.ctx, int(bool(on))))

    def mctf_set_timing(self, on):
        self._ck(self.L.vvhip_mctf_set_timing(self.ctx, int(bool(on))))

    def mctf_last_times(self):
        """ms of the last motion-estimation call per class: (candidate scoring, neighbour scoring, sweep, final normalisation, rest)"""
        a = (C.c_float * 5)()
        self._ck(self.L.vvhip_mctf_last_times(self.ctx, a))
        return tuple(float(x) for x in a)

    def mctf_get_stats(self):
        """scored candidates of the MCTF search since mctf_set_stats(True): {phase: {int, int_bytes, frac, frac_bytes, grid, grid_window_bytes, ring, ring_window_bytes}}
        (include/vvenc_hip.h)"""
        a = np.zeros(24, np.uint64)
        self._ck(self.L.vvhip_mctf_get_stats(self.ctx, a.ctypes.data_as(C.c_void_p)))
        keys = ("int", "int_bytes", "frac", "frac_bytes", "grid", "grid_window_bytes", "ring", "ring_window_bytes")
        return {ph: {k: int(a[8 * i + j]) for j, k in enumerate(keys)} for i, ph in enumerate(("search", "neighbour", "sweep"))}

    @staticmethod
    def mv_to_numpy(t, dims):
        return t.cpu().numpy().view(MV_DTYPE).reshape(dims[1], dims[0])
