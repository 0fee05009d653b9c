"""ctypes/numpy bindings of the CPU parity oracle and of the compiled reference.

TEST INFRASTRUCTURE ONLY — may be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never from vvenc_amd/ (the product path).

* `Oracle`  wraps oracle/liboracle.so   (our C restatement, oracle/vvenc_oracle.c)
* `RefLib`  wraps oracle/_ref/libvvenc_ref.so (the reference itself, compiled from /root/reference
            by oracle/ref/Makefile; present only where that build has been run / shipped)

Both expose the same method names so tests can run one body against either.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libvvenc_ref.so")

DCT2, DCT8, DST7 = 0, 1, 2

MV_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("error", "<i4"), ("rmsme", "<i4"), ("overlap", "<f8")])


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE])


def build_ref():
    """Compile the reference from /root/reference (only possible where it exists)."""
    if not os.path.isdir("/root/reference/source/Lib"):
        return False
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "ref")])
    return True


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i16(a):
    a = np.asarray(a)
    assert a.dtype == np.int16
    return a


def _view_ptr(a, off_elems=0):
    """pointer to element `off_elems` of the (C-contiguous) array a"""
    return C.c_void_p(a.ctypes.data + off_elems * a.itemsize)


class _Base:
    """Helpers shared by both libraries: 2-D numpy views in, python ints out.

    A 'view' is (array2d, y0, x0): the block's top-left sample inside a larger C-contiguous
    int16 array, so negative displacements / margins work exactly like the reference's
    pointer arithmetic.
    """

    @staticmethod
    def _ptr_stride(view):
        if isinstance(view, tuple):
            arr, y0, x0 = view
        else:
            arr, y0, x0 = view, 0, 0
        arr = _i16(arr)
        assert arr.flags["C_CONTIGUOUS"] and arr.ndim == 2
        stride = arr.shape[1]
        return _view_ptr(arr, y0 * stride + x0), stride

    # ---- interpolation filter (InterpolationFilter.cpp) ----
    def if_coeff(self, set_, phase):
        out = np.zeros(8, np.int16)
        n = getattr(self.L, self._pfx + "if_coeff")(set_, phase, _p(out))
        return n, out

    @staticmethod
    def _if_src(src):
        """src: (array, y, x) view with enough margin around the block for the taps"""
        arr, y0, x0 = src
        arr = _i16(arr)
        assert arr.flags["C_CONTIGUOUS"] and arr.ndim == 2
        return _view_ptr(arr, y0 * arr.shape[1] + x0), arr.shape[1]

    def if_filter(self, n, vertical, first, last, bd, src, w, h, coeff):
        ps, ss = self._if_src(src)
        dst = np.full((h + 2, w + 32), -99, np.int16)      # slack: the x86 rows store whole vectors
        c = np.ascontiguousarray(coeff, np.int16)
        self._if_call("if_filter", n, int(vertical), int(first), int(last), bd, ps, ss, _p(dst), w + 32, w, h, _p(c))
        return dst[:h, :w].copy()

    def if_copy(self, first, last, bd, src, w, h, bi_mc=False):
        ps, ss = self._if_src(src)
        dst = np.full((h + 2, w + 32), -99, np.int16)      # slack: the x86 rows store whole vectors
        self._if_call("if_copy", int(first), int(last), bd, ps, ss, _p(dst), w + 32, w, h, int(bi_mc))
        return dst[:h, :w].copy()

    def if_luma_1d(self, vertical, src, w, h, frac, first, last, bd=10, alt=False, reduce_tap=0):
        ps, ss = self._if_src(src)
        dst = np.full((h + 2, w + 32), -99, np.int16)      # slack: the x86 rows store whole vectors
        self._if_call("if_luma_1d", int(vertical), ps, ss, _p(dst), w + 32, w, h, frac, int(first), int(last), bd, int(alt), reduce_tap)
        return dst[:h, :w].copy()

    def if_pred_luma(self, ref, w, h, xfrac, yfrac, rnd=True, bd=10, alt=False):
        ps, ss = self._if_src(ref)
        dst = np.full((h + 2, w + 32), -99, np.int16)      # slack: the x86 rows store whole vectors
        self._if_call("if_pred_luma", ps, ss, _p(dst), w + 32, w, h, xfrac, yfrac, int(rnd), bd, int(alt))
        return dst[:h, :w].copy()

    def if_pred_luma_me(self, ref, w, h, xfrac, yfrac, bd=10, alt=False, reduce_tap=0):
        """the two passes InterSearch::xPatternRefinement / xExtDIFUpSampling* run for one sub-pel position (InterSearch.cpp:818-848):
        horizontal pass with isLast=false over the rows the vertical taps need, vertical pass isFirst=false isLast=true"""
        arr, y0, x0 = ref
        rows = h + 7
        tmp = self.if_luma_1d(0, (arr, y0 - 3, x0), w, rows, xfrac, 1, 0, bd, alt, reduce_tap)
        pad = np.zeros((rows + 8, w + 16), np.int16)
        pad[4:4 + rows, 8:8 + w] = tmp
        return self.if_luma_1d(1, (pad, 4 + 3, 8), w, h, yfrac, 0, 1, bd, alt, reduce_tap)

    def _if_call(self, name, *a):
        f = getattr(self.L, self._pfx + name)
        f.restype = None
        if self._pfx == "vvref_":
            a = (self.simd,) + a
        f(*a)

    # ---- MCTF apply side (MCTF.cpp:259-518, 1399-1552) ----
    REF_STRENGTHS = ((0.84375, 0.6, 0.4286, 0.3333, 0.2727, 0.2308), (1.12500, 1.0, 0.7143, 0.5556, 0.4545, 0.3846))      # MCTF.cpp:112-117

    def mctf_apply_frac(self, tap4, src, w, h, fx, fy, bit_depth=10, chroma=False):
        ps, ss = self._ptr_stride(src)
        dst = np.full((h + 2, w + 32), -99, np.int16)
        a = (int(tap4), ps, C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w + 32), w, h, fx, fy, bit_depth)
        if self._pfx == "vvref_":
            a = (self.simd, int(chroma)) + a
        f = getattr(self.L, self._pfx + "mctf_apply_frac"); f.restype = None; f(*a)
        return dst[:h, :w].copy()

    def mctf_planar_correction(self, ref, block, w, h, bit_depth, motion_error):
        pr, rs = self._ptr_stride(ref)
        dst = np.full((h + 2, w + 32), 0, np.int16)
        dst[:h, :w] = block
        a = (pr, C.c_ssize_t(rs), _p(dst), C.c_ssize_t(w + 32), w, h, bit_depth, int(motion_error))
        if self._pfx == "vvref_":
            a = (self.simd,) + a
        f = getattr(self.L, self._pfx + "mctf_planar_correction"); f.restype = None; f(*a)
        return dst[:h, :w].copy()

    def mctf_apply_block(self, src, corrected, verror, ref_strengths, weight_scaling, sigma_sq, w, h, bit_depth=10):
        ps, ss = self._ptr_stride(src)
        n = len(corrected)
        keep = [_aligned(np.ascontiguousarray(c, np.int16)) for c in corrected]
        ptrs = (C.c_void_p * n)(*[k.ctypes.data for k in keep])
        ve = np.ascontiguousarray(verror, np.int32)
        rs = np.ascontiguousarray(ref_strengths, np.float64)
        dst = np.full((h + 2, w + 32), -99, np.int16)
        a = (ps, C.c_ssize_t(ss), _p(dst), C.c_ssize_t(w + 32), w, h, bit_depth, ptrs, n, _p(ve), _p(rs), C.c_double(weight_scaling), C.c_double(sigma_sq))
        if self._pfx == "vvref_":
            a = (self.simd,) + a
        f = getattr(self.L, self._pfx + "mctf_apply_block"); f.restype = None; f(*a)
        return dst[:h, :w].copy()

    @staticmethod
    def mctf_filter_params(qp, bit_depth, overall_strength, chroma):
        """sigmaSq / weightScaling as MCTF::bilateralFilter and xFinalizeBlkLine derive them (MCTF.cpp:1491-1501, :1417)"""
        luma_sigma = 9.0 * (128.0 + 3.0 / 256.0 * qp * qp * qp)
        bdw = 1024.0 / (1 << bit_depth)
        sigma = (30.0 * 30.0 if chroma else luma_sigma) / (bdw * bdw)
        return sigma, overall_strength * (0.55 if chroma else 0.4)

    # ---- DMVR refinement search ----
    def if_bilinear(self, src, w, h, fx, fy, bd=10):
        ps, ss = self._if_src(src)
        dst = np.full((h + 2, w + 32), -99, np.int16)
        self._if_call("if_bilinear", ps, ss, _p(dst), w + 32, w, h, fx, fy, bd)
        return dst[:h, :w].copy()

    def dmvr_subpel_error_surface(self, sad5):
        s5 = np.ascontiguousarray(sad5, np.uint64)
        d = np.zeros(2, np.int32)
        f = getattr(self.L, self._pfx + "dmvr_subpel_error_surface"); f.restype = None
        f(_p(s5), _p(d))
        return d

    def dmvr_refine(self, ref0, ref1, frac0, frac1, dx, dy, bd=10):
        """ref0 / ref1: (array, y, x) = the sub-block's integer position for the merge vector; frac = (x, y) in 1/16; -> (mvd_x, mvd_y, minCost)"""
        p0, s0 = self._if_src(ref0)
        p1, s1 = self._if_src(ref1)
        mvd = np.zeros(2, np.int16)
        f = getattr(self.L, self._pfx + "dmvr_refine"); f.restype = C.c_uint64
        a = (p0, s0, frac0[0], frac0[1], p1, s1, frac1[0], frac1[1], dx, dy, bd, _p(mvd))
        if self._pfx == "vvref_":
            a = (self.simd,) + a
        cost = f(*a)
        return int(mvd[0]), int(mvd[1]), int(cost)

    # ---- ALF encoder statistics ----
    ALF_REC = 13 * 13 + 13 + 1

    @staticmethod
    def alf_pad(plane, margin=8):
        """replicated border, like the reference's extended m_tempBuf -> (padded array, margin)"""
        return np.ascontiguousarray(np.pad(np.ascontiguousarray(plane, np.int16), margin, mode="edge")), margin

    def alf_classify(self, rec, bit_depth=10, vb_ctu_height=128, vb_pos=124):
        """rec: (H, W) int16 luma (H, W multiples of 4) -> (H/4, W/4, 2) uint8 {classIdx, transposeIdx}"""
        h, w = rec.shape
        pad, m = self.alf_pad(rec)
        cls = np.zeros((h // 4, w // 4, 2), np.uint8)
        base = pad.ctypes.data + 2 * (m * pad.shape[1] + m)
        f = getattr(self.L, self._pfx + "alf_classify"); f.restype = None if self._pfx == "orc_" else C.c_int
        a = (C.c_void_p(base), C.c_ssize_t(pad.shape[1]) if self._pfx == "orc_" else pad.shape[1], w, h, bit_depth + 4, vb_ctu_height, vb_pos)
        if self._pfx == "vvref_":
            a = a + (self.simd,)
        f(*a, _p(cls))
        return cls

    def alf_stats_plane(self, org, rec, ctu_size, filter_length, cls=None, vb_ctu_height=128, vb_pos=124, init=None, ctu_in_unit=None):
        """-> (numCtus, numClasses, ALF_REC) float32: E[13][13], y[13], pixAcc per CTU and class, accumulated in the reference's order"""
        h, w = rec.shape
        pad, m = self.alf_pad(rec)
        org = np.ascontiguousarray(org, np.int16)
        ncls = 25 if cls is not None else 1
        nctu = ((w + ctu_size - 1) // ctu_size) * ((h + ctu_size - 1) // ctu_size)
        out = np.zeros((nctu, ncls, self.ALF_REC), np.float32) if init is None else np.ascontiguousarray(init, np.float32).copy().reshape(nctu, ncls, self.ALF_REC)
        base = pad.ctypes.data + 2 * (m * pad.shape[1] + m)
        clsp = _p(np.ascontiguousarray(cls, np.uint8)) if cls is not None else None
        assert init is None or self._pfx == "orc_"
        if ctu_in_unit is not None:
            assert init is None
            f = getattr(self.L, self._pfx + "alf_stats_plane_units"); f.restype = None if self._pfx == "orc_" else C.c_int
            if self._pfx == "orc_":
                f(_p(org), C.c_ssize_t(org.shape[1]), C.c_void_p(base), C.c_ssize_t(pad.shape[1]), w, h, ctu_size, ctu_in_unit, filter_length, clsp, vb_ctu_height, vb_pos, _p(out))
            else:
                f(_p(org), org.shape[1], C.c_void_p(base), pad.shape[1], w, h, ctu_size, ctu_in_unit, filter_length, clsp, vb_ctu_height, vb_pos, self.simd, _p(out))
            return out
        f = getattr(self.L, self._pfx + ("alf_stats_plane" if init is None else "alf_stats_plane_acc")); f.restype = None if self._pfx == "orc_" else C.c_int
        if self._pfx == "orc_":
            f(_p(org), C.c_ssize_t(org.shape[1]), C.c_void_p(base), C.c_ssize_t(pad.shape[1]), w, h, ctu_size, filter_length, clsp, vb_ctu_height, vb_pos, _p(out))
        else:
            f(_p(org), org.shape[1], C.c_void_p(base), pad.shape[1], w, h, ctu_size, filter_length, clsp, vb_ctu_height, vb_pos, self.simd, _p(out))
        return out

    def ccalf_stats_plane(self, org_c, slf_c, rec_luma, ctu_size_c, vb_ctu_height=128, vb_pos=124, init=None):
        """CC-ALF covariance records per chroma CTU (4:2:0) -> (numCtus, ALF_REC) float32; only E[:7,:7], y[:7], pixAcc are defined"""
        hc, wc = slf_c.shape
        pad, m = self.alf_pad(rec_luma)
        org_c, slf_c = np.ascontiguousarray(org_c, np.int16), np.ascontiguousarray(slf_c, np.int16)
        nctu = ((wc + ctu_size_c - 1) // ctu_size_c) * ((hc + ctu_size_c - 1) // ctu_size_c)
        out = np.zeros((nctu, self.ALF_REC), np.float32) if init is None else np.ascontiguousarray(init, np.float32).copy().reshape(nctu, self.ALF_REC)
        base = pad.ctypes.data + 2 * (m * pad.shape[1] + m)
        f = getattr(self.L, self._pfx + "ccalf_stats_plane"); f.restype = None if self._pfx == "orc_" else C.c_int
        if self._pfx == "orc_":
            f(_p(org_c), C.c_ssize_t(org_c.shape[1]), _p(slf_c), C.c_ssize_t(slf_c.shape[1]), C.c_void_p(base), C.c_ssize_t(pad.shape[1]), wc, hc, ctu_size_c, 1, 1,
              vb_ctu_height, vb_pos, rec_luma.shape[0], _p(out))
        else:
            assert init is None
            f(_p(org_c), org_c.shape[1], _p(slf_c), slf_c.shape[1], C.c_void_p(base), pad.shape[1], wc, hc, ctu_size_c, vb_ctu_height, vb_pos, rec_luma.shape[0], self.simd, _p(out))
        return out

    def alf_filter_plane(self, src, ctu_size, bit_depth, filter_length, coeff_sets, clip_sets, ctu_set, cls=None, dst=None, vb_ctu_height=128, vb_pos=124):
        """filterBlk over the enabled CTUs of a plane.  src: (H, W) int16; coeff_sets / clip_sets: (numSets, numClasses, 13) int16;
        ctu_set: (numCtus,) int16, < 0 = CTU keeps dst.  -> filtered plane (dst defaults to a copy of src)"""
        h, w = src.shape
        pad, m = self.alf_pad(src)
        out = np.ascontiguousarray(src if dst is None else dst, np.int16).copy()
        base = pad.ctypes.data + 2 * (m * pad.shape[1] + m)
        cf, cp = np.ascontiguousarray(coeff_sets, np.int16), np.ascontiguousarray(clip_sets, np.int16)
        cs = np.ascontiguousarray(ctu_set, np.int16)
        clsp = _p(np.ascontiguousarray(cls, np.uint8)) if cls is not None else None
        f = getattr(self.L, self._pfx + "alf_filter_plane"); f.restype = None if self._pfx == "orc_" else C.c_int
        if self._pfx == "orc_":
            f(C.c_void_p(base), C.c_ssize_t(pad.shape[1]), _p(out), C.c_ssize_t(w), w, h, ctu_size, bit_depth, filter_length, clsp, _p(cf), _p(cp), _p(cs), vb_ctu_height, vb_pos)
        else:
            f(C.c_void_p(base), pad.shape[1], _p(out), w, w, h, ctu_size, bit_depth, filter_length, clsp, _p(cf), _p(cp), _p(cs), vb_ctu_height, vb_pos,
              int(bool((cp != (1 << bit_depth)).any())), self.simd)
        return out

    def ccalf_filter_plane(self, dst_c, rec_luma, ctu_size_c, bit_depth, coeff, ctu_filter, vb_ctu_height=128, vb_pos=124):
        """filterBlkCcAlf over a chroma plane (4:2:0).  coeff: (numFilters, 8) int16; ctu_filter: (numCtus,) uint8, 0 = off, k = filter k-1 -> corrected plane"""
        hc, wc = dst_c.shape
        pad, m = self.alf_pad(rec_luma)
        out = np.ascontiguousarray(dst_c, np.int16).copy()
        base = pad.ctypes.data + 2 * (m * pad.shape[1] + m)
        cf, fl = np.ascontiguousarray(coeff, np.int16), np.ascontiguousarray(ctu_filter, np.uint8)
        f = getattr(self.L, self._pfx + "ccalf_filter_plane"); f.restype = None if self._pfx == "orc_" else C.c_int
        if self._pfx == "orc_":
            f(_p(out), C.c_ssize_t(wc), C.c_void_p(base), C.c_ssize_t(pad.shape[1]), wc, hc, ctu_size_c, 1, 1, bit_depth, _p(cf), _p(fl), vb_ctu_height, vb_pos)
        else:
            f(_p(out), wc, C.c_void_p(base), pad.shape[1], wc, hc, ctu_size_c, bit_depth, _p(cf), _p(fl), vb_ctu_height, vb_pos, self.simd)
        return out

    # ---- g_tCoeffOps table slots (TrQuant_EMT.h:63-91), caller's matrix ----
    def fast_fwd_core(self, tc, src, line, reduced_line, cutoff, shift):
        """tc: (N, N) int16, src: (line, N) int32 -> dst (N, line) int32 (entries outside reduced_line x cutoff stay 0)"""
        n = tc.shape[0]
        tc, src = _aligned(np.ascontiguousarray(tc, np.int16)), _aligned(np.ascontiguousarray(src, np.int32))
        dst = _aligned(np.zeros((n, line), np.int32))
        self._slot("fast_fwd_core", n, _p(tc), _p(src), _p(dst), line, reduced_line, cutoff, shift)
        return dst

    def fast_inv_core(self, it, src, dst0, lines, reduced_lines, rows):
        """it: (N, N) int16, src: (N, lines) int32, dst0: (lines, N) int32 start values -> accumulated copy"""
        n = it.shape[0]
        it, src = _aligned(np.ascontiguousarray(it, np.int16)), _aligned(np.ascontiguousarray(src, np.int32))
        dst = _aligned(np.ascontiguousarray(dst0, np.int32))
        self._slot("fast_inv_core", n, _p(it), _p(src), _p(dst), lines, reduced_lines, rows)
        return dst

    def round_clip(self, buf, w, h, stride, mn, mx, rnd, shift):
        dst = _aligned(np.ascontiguousarray(buf, np.int32))
        self._slot("round_clip", None, _p(dst), w, h, stride, mn, mx, rnd, shift)
        return dst

    def cpy_resi(self, src, w, h, stride):
        src = _aligned(np.ascontiguousarray(src, np.int32))
        dst = np.full((h, stride), -77, np.int16)
        self._slot("cpy_resi", None, _p(src), _p(dst), C.c_ssize_t(stride), w, h)
        return dst

    def cpy_coeff(self, src, w, h):
        """src: (h, stride) int16"""
        src = np.ascontiguousarray(src, np.int16)
        dst = _aligned(np.zeros((h, w), np.int32))
        self._slot("cpy_coeff", None, _p(src), C.c_ssize_t(src.shape[1]), _p(dst), w, h)
        return dst


class Oracle(_Base):
    _pfx = "orc_"
    def __init__(self):
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "vvenc_oracle.c")):
            build_oracle()
        L = self.L = C.CDLL(ORACLE_SO)
        u64, i32, vp, i64 = C.c_uint64, C.c_int, C.c_void_p, C.c_int64
        L.orc_sad.restype = u64
        L.orc_sad.argtypes = [vp, i32, vp, i32, i32, i32, i32]
        L.orc_sse.restype = u64
        L.orc_sse.argtypes = [vp, i32, vp, i32, i32, i32]
        L.orc_had.restype = u64
        L.orc_had.argtypes = [vp, i32, vp, i32, i32, i32, i32]
        L.orc_had_2sad.restype = u64
        L.orc_had_2sad.argtypes = [vp, vp, i32, i32]
        L.orc_sad_x5.restype = None
        L.orc_sad_x5.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, i32]
        L.orc_sad_mask.restype = u64
        L.orc_sad_mask.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32]
        L.orc_fix_weighted_sse.restype = u64
        L.orc_fix_weighted_sse.argtypes = [vp, i32, vp, i32, i32, i32, C.c_uint32]
        L.orc_tr_matrix.argtypes = [i32, i32, vp]
        L.orc_fwd_1d.argtypes = [i32, i32, vp, vp, i32, i32, i32, i32]
        L.orc_inv_1d.argtypes = [i32, i32, vp, vp, i32, i32, i32, i32, i32, i32]
        L.orc_xT.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32]
        L.orc_xIT.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32]
        L.orc_scan_order.argtypes = [i32, i32, vp]
        L.orc_quant_params.restype = None
        L.orc_quant_params.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp]
        L.orc_dequant_params.restype = None
        L.orc_dequant_params.argtypes = [i32, i32, i32, i32, vp, vp, vp]
        L.orc_need_rdoq_params.restype = None
        L.orc_need_rdoq_params.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp]
        L.orc_quant_core.restype = None
        L.orc_quant_core.argtypes = [vp, vp, vp, i32, i32, i32, i32, i64, i32, vp, vp]
        L.orc_dequant_core.restype = None
        L.orc_dequant_core.argtypes = [i32, i32, i32, vp, C.c_size_t, vp, i32, i32, i32]
        L.orc_need_rdoq.argtypes = [vp, C.c_size_t, i32, i64, i32]
        L.orc_mctf_err_int.argtypes = [vp, C.c_ssize_t, vp, C.c_ssize_t, i32, i32]
        L.orc_mctf_err_frac.argtypes = [i32, vp, C.c_ssize_t, vp, C.c_ssize_t, i32, i32, i32, i32, i32]
        L.orc_mctf_calc_var.restype = C.c_double
        L.orc_mctf_calc_var.argtypes = [vp, C.c_ssize_t, i32, i32]
        L.orc_mctf_subsample.restype = None
        L.orc_mctf_subsample.argtypes = [vp, i32, i32, i32, vp, i32]
        L.orc_extend_border.restype = None
        L.orc_extend_border.argtypes = [vp, i32, i32, i32, i32]
        L.orc_mctf_me.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]

    name = "oracle"

    def _slot(self, name, n, *a):
        f = getattr(self.L, "orc_" + name)
        f.restype = None
        if n is not None:
            a = (n,) + a
        f(*[C.c_void_p(x) if isinstance(x, int) and x > (1 << 32) else x for x in a])

    # ---- distortion ----
    def dist(self, func, org, cur, w, h, bit_depth=10, sub_shift=0):
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        L = self.L
        if func == "SAD":
            return L.orc_sad(po, so, pc, sc, w, h, sub_shift)
        if func == "SSE":
            return L.orc_sse(po, so, pc, sc, w, h)
        if func == "HAD":
            return L.orc_had(po, so, pc, sc, w, h, 0)
        if func == "HAD_fast":
            return L.orc_had(po, so, pc, sc, w, h, 1)
        if func == "HAD_2SAD":
            assert so == w and sc == w
            return L.orc_had_2sad(po, pc, w, h)
        raise ValueError(func)

    def sad_x5(self, org, cur, w, h, sub_shift=1, calc_centre=True):
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        out = np.zeros(5, np.uint64)
        self.L.orc_sad_x5(po, so, pc, sc, w, h, sub_shift, _p(out), int(calc_centre))
        return out

    def fix_weighted_sse(self, org, cur, w, h, weight):
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        return self.L.orc_fix_weighted_sse(po, so, pc, sc, w, h, weight)

    def sad_mask(self, org, cur, mask, step_x, mask_stride2, w, h, sub_shift=0):
        """mask: (array, y, x) view whose first sample is the first mask sample read; mask_stride = the array's row pitch"""
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        pm, sm = self._ptr_stride(mask)
        return self.L.orc_sad_mask(po, so, pc, sc, pm, sm, step_x, mask_stride2, w, h, sub_shift)

    # ---- transforms ----
    def tr_matrix(self, tr_type, log2n):
        n = 1 << log2n
        out = np.zeros((n, n), np.int16)
        rc = self.L.orc_tr_matrix(tr_type, log2n, _p(out))
        return out if rc == 0 else None

    def fwd_1d(self, tr_type, log2n, src, shift, line, skip, skip2):
        src = np.ascontiguousarray(src, np.int32)
        dst = np.zeros((1 << log2n) * line, np.int32)
        rc = self.L.orc_fwd_1d(tr_type, log2n, _p(src), _p(dst), shift, line, skip, skip2)
        assert rc == 0
        return dst

    def inv_1d(self, tr_type, log2n, src, shift, line, skip, skip2, cmin=-32768, cmax=32767):
        src = np.ascontiguousarray(src, np.int32)
        dst = np.zeros((1 << log2n) * line, np.int32)
        rc = self.L.orc_inv_1d(tr_type, log2n, _p(src), _p(dst), shift, line, skip, skip2, cmin, cmax)
        assert rc == 0
        return dst

    def xT(self, resi, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10):
        resi = np.ascontiguousarray(resi, np.int16)
        h, w = resi.shape
        coef = np.zeros((h, w), np.int32)
        rc = self.L.orc_xT(_p(resi), w, _p(coef), w, h, tr_hor, tr_ver, bit_depth)
        assert rc == 0, rc
        return coef

    def xIT(self, coef, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10):
        coef = np.ascontiguousarray(coef, np.int32)
        h, w = coef.shape
        resi = np.zeros((h, w), np.int16)
        rc = self.L.orc_xIT(_p(coef), _p(resi), w, w, h, tr_hor, tr_ver, bit_depth)
        assert rc == 0, rc
        return resi

    # ---- quant ----
    def scan_order(self, log2w, log2h):
        out = np.zeros(1 << (log2w + log2h), np.uint32)
        self.L.orc_scan_order(log2w, log2h, _p(out))
        return out

    def quant_params(self, w, h, bit_depth, qp, is_irap):
        a, b, c = C.c_int(), C.c_int(), C.c_int64()
        self.L.orc_quant_params(w, h, bit_depth, qp, int(is_irap), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def dequant_params(self, w, h, bit_depth, qp):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_dequant_params(w, h, bit_depth, qp, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def need_rdoq_params(self, w, h, bit_depth, qp, is_luma):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        self.L.orc_need_rdoq_params(w, h, bit_depth, qp, int(is_luma), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return a.value, b.value, c.value, d.value

    def quant_core(self, coef, quant_coeff, q_bits, add, thr_val=8, sign_hiding=False, lfnst_idx=0):
        coef = np.ascontiguousarray(coef, np.int32)
        h, w = coef.shape
        q = np.zeros((h, w), np.int16)
        du = np.zeros(h * w, np.int32)
        s, last = C.c_int32(), C.c_int()
        self.L.orc_quant_core_lfnst.restype = None
        self.L.orc_quant_core_lfnst(_p(coef), _p(q), _p(du), w, h, quant_coeff, q_bits, C.c_int64(add), thr_val, int(lfnst_idx), C.byref(s), C.byref(last))
        return q, du, s.value, last.value

    def dequant_core(self, q, scale, right_shift, input_max, tr_max=32767):
        q = np.ascontiguousarray(q, np.int16)
        h, w = q.shape
        coef = np.zeros((h, w), np.int32)
        self.L.orc_dequant_core(w - 1, h - 1, scale, _p(q), w, _p(coef), right_shift, input_max, tr_max)
        return coef

    def need_rdoq(self, coef, quant_coeff, offset, shift):
        coef = np.ascontiguousarray(coef, np.int32).ravel()
        return int(self.L.orc_need_rdoq(_p(coef), coef.size, quant_coeff, offset, shift))

    # (qp, flags)-level entry points, same names as tests/hip_backend.py
    def quant_tu(self, coef, qp, irap, thr_val=8, bit_depth=10):
        h, w = np.asarray(coef).shape
        qc, qbits, add = self.quant_params(w, h, bit_depth, qp, irap)
        return self.quant_core(coef, qc, qbits, add, thr_val)

    def dequant_tu(self, level, qp, bit_depth=10):
        h, w = np.asarray(level).shape
        sc, rs, imax = self.dequant_params(w, h, bit_depth, qp)
        return self.dequant_core(level, sc, rs, imax)

    def need_rdoq_tu(self, coef, qp, is_luma=1, bit_depth=10):
        h, w = np.asarray(coef).shape
        qc, qbits, add, num = self.need_rdoq_params(w, h, bit_depth, qp, is_luma)
        return self.need_rdoq(np.asarray(coef).ravel()[:num], qc, add, qbits)

    def tu_rdo(self, resi, qp, irap, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10, thr_val=8, is_luma=1):
        """the fused pipeline's definition, composed from the pinned pieces: xT -> needRdoq/QuantCore -> DeQuantCore -> xIT -> SSE"""
        resi = np.ascontiguousarray(resi, np.int16)
        coef = self.xT(resi, tr_hor, tr_ver, bit_depth)
        need = self.need_rdoq_tu(coef, qp, is_luma, bit_depth)
        lev, _, s, last = self.quant_tu(coef, qp, irap, thr_val, bit_depth)
        rec = self.xIT(self.dequant_tu(lev, qp, bit_depth), tr_hor, tr_ver, bit_depth)
        d = resi.astype(np.int64) - rec.astype(np.int64)
        return lev, rec, dict(abs_sum=s, last_scan_pos=last, need_rdoq=need, sse=int((d * d).sum()))

    # ---- MCTF ----
    def mctf_err_int(self, org, buf, w, h):
        po, so = self._ptr_stride(org)
        pb, sb = self._ptr_stride(buf)
        return self.L.orc_mctf_err_int(po, so, pb, sb, w, h)

    def mctf_err_frac(self, tap4, org, buf, w, h, fx, fy, bit_depth=10):
        po, so = self._ptr_stride(org)
        pb, sb = self._ptr_stride(buf)
        return self.L.orc_mctf_err_frac(int(tap4), po, so, pb, sb, w, h, fx, fy, bit_depth)

    def mctf_calc_var(self, org, w, h):
        po, so = self._ptr_stride(org)
        return self.L.orc_mctf_calc_var(po, so, w, h)

    def mctf_subsample(self, plane):
        plane = np.ascontiguousarray(plane, np.int16)
        h, w = plane.shape
        out = np.zeros((h // 2, w // 2), np.int16)
        self.L.orc_mctf_subsample(_p(plane), w, w, h, _p(out), w // 2)
        return out

    def mctf_bilateral(self, org, refs, mvs, ref_index, bit_depth=10, qp=32, unit=16, low_res=True, pic_reordering=True, overall_strength=0.95):
        """org: (Y, U, V) arrays; refs: list of (Y, U, V); mvs: list of MV_DTYPE arrays (final level); -> (Y, U, V) filtered"""
        h, w = org[0].shape
        wb = (w + unit - 1) // unit
        strengths = np.array([self.REF_STRENGTHS[0 if pic_reordering else 1][k] for k in ref_index], np.float64)
        out = []
        for c in range(3):
            cs = 1 if c else 0
            pad = 128 >> cs
            po = np.ascontiguousarray(np.pad(org[c], pad, mode="edge"), np.int16)
            prs = [np.ascontiguousarray(np.pad(r[c], pad, mode="edge"), np.int16) for r in refs]
            stride = po.shape[1]
            sigma, scaling = self.mctf_filter_params(qp, bit_depth, overall_strength, c > 0)
            o = np.zeros_like(org[c], dtype=np.int16)
            rp = (C.c_void_p * len(refs))(*[_view_ptr(p, pad * stride + pad).value for p in prs])
            keep = [np.ascontiguousarray(m) for m in mvs]
            mp = (C.c_void_p * len(refs))(*[k.ctypes.data for k in keep])
            f = self.L.orc_mctf_bilateral_plane
            f.restype = None
            f(_view_ptr(po, pad * stride + pad), C.c_ssize_t(stride), org[c].shape[1], org[c].shape[0], cs, bit_depth, unit, int(low_res), qp, len(refs), rp,
              C.c_ssize_t(stride), mp, wb, _p(strengths), C.c_double(scaling), C.c_double(sigma), _p(o), C.c_ssize_t(o.shape[1]))
            out.append(o)
        return tuple(out)

    def mctf_me(self, org, ref, bit_depth=10, unit=16, speed=4, add_level=None):
        return _mctf_me(self.L.orc_mctf_me, None, org, ref, bit_depth, unit, speed, add_level)

    def mctf_me_counted(self, org, ref, bit_depth=10, unit=16, speed=4, add_level=None):
        """mctf_me + the motionErrorLuma calls it made: {int, int_bytes, frac, frac_bytes} (the reference's schedule: every call of estimateLumaLn)"""
        self.L.orc_mctf_count_reset.restype = None
        self.L.orc_mctf_count_get.restype = None
        self.L.orc_mctf_count_reset()
        res = self.mctf_me(org, ref, bit_depth, unit, speed, add_level)
        c = np.zeros(4, np.uint64)
        self.L.orc_mctf_count_get(_p(c))
        return res, {"int": int(c[0]), "int_bytes": int(c[1]), "frac": int(c[2]), "frac_bytes": int(c[3])}


def _mctf_me(fn, simd, org, ref, bit_depth, unit, speed, add_level):
    org = np.ascontiguousarray(org, np.int16)
    ref = np.ascontiguousarray(ref, np.int16)
    h, w = org.shape
    if add_level is None:
        add_level = w >= 1920   # MCTF.cpp:768
    dims = [(w // (unit * 16) + 1, h // (unit * 16) + 1), (w // (unit * 8) + 1, h // (unit * 8) + 1),
            (w // (unit * 4) + 1, h // (unit * 4) + 1), (w // (unit * 2) + 1, h // (unit * 2) + 1),
            ((w + unit - 1) // unit, (h + unit - 1) // unit)]
    outs = [np.zeros(dw * dh, MV_DTYPE) for dw, dh in dims]
    ptrs = (C.c_void_p * 5)(*[o.ctypes.data for o in outs])
    ld = np.zeros(10, np.int32)
    args = [_p(org), _p(ref), w, h, bit_depth, unit, speed, int(bool(add_level)), ptrs, _p(ld)]
    if simd is not None:
        args = [int(simd)] + args
    rc = fn(*args)
    assert rc == 0
    res = []
    for k in range(5):
        dw, dh = int(ld[2 * k]), int(ld[2 * k + 1])
        if dw == 0:
            res.append(None)
            continue
        assert (dw, dh) == dims[k]
        res.append(outs[k].reshape(dh, dw))
    return res


class RefLib(_Base):
    _pfx = "vvref_"
    """The reference's own kernels (scalar row simd=0, x86 SIMD row simd=1)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, simd=0):
        self.simd = int(simd)
        self.name = "reference[%s]" % ("simd" if simd else "scalar")
        L = self.L = C.CDLL(REF_SO)
        u64, i32, vp, i64 = C.c_uint64, C.c_int, C.c_void_p, C.c_int64
        L.vvref_df.argtypes = [C.c_char_p]
        L.vvref_dist.restype = u64
        L.vvref_dist.argtypes = [i32, i32, vp, i32, vp, i32, i32, i32, i32, i32]
        L.vvref_sad_x5.restype = None
        L.vvref_sad_x5.argtypes = [i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32]
        L.vvref_fix_weighted_sse.restype = u64
        L.vvref_fix_weighted_sse.argtypes = [i32, vp, i32, vp, i32, i32, i32, i32, C.c_uint32]
        L.vvref_sad_mask.restype = u64
        L.vvref_sad_mask.argtypes = [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32]
        L.vvref_tr_matrix.argtypes = [i32, i32, vp]
        L.vvref_fwd_1d.argtypes = [i32, i32, i32, vp, vp, i32, i32, i32, i32]
        L.vvref_inv_1d.argtypes = [i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32]
        L.vvref_xT.argtypes = [i32, vp, i32, vp, i32, i32, i32, i32, i32]
        L.vvref_xIT.argtypes = [i32, vp, vp, i32, i32, i32, i32, i32, i32]
        L.vvref_scan_order.argtypes = [i32, i32, vp]
        L.vvref_quant_scales.restype = None
        L.vvref_quant_scales.argtypes = [vp, vp]
        L.vvref_dequant_core.restype = None
        L.vvref_dequant_core.argtypes = [i32, i32, i32, i32, vp, C.c_size_t, vp, i32, i32, i32]
        L.vvref_need_rdoq_core.argtypes = [i32, vp, C.c_size_t, i32, i64, i32]
        L.vvref_quant_core.argtypes = [vp, vp, vp, i32, i32, i32, i32, i64, i32, i32, vp, vp]
        L.vvref_mctf_err_int.argtypes = [i32, vp, C.c_ssize_t, vp, C.c_ssize_t, i32, i32, i32]
        L.vvref_mctf_err_frac.argtypes = [i32, i32, vp, C.c_ssize_t, vp, C.c_ssize_t, i32, i32, i32, i32, i32, i32]
        L.vvref_mctf_filters.restype = None
        L.vvref_mctf_filters.argtypes = [vp, vp]
        L.vvref_mctf_calc_var.restype = C.c_double
        L.vvref_mctf_calc_var.argtypes = [i32, vp, C.c_ssize_t, i32, i32]
        L.vvref_mctf_subsample.restype = None
        L.vvref_mctf_subsample.argtypes = [vp, i32, i32, vp]
        L.vvref_mctf_me.argtypes = [i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
        self._df = {n: L.vvref_df(n.encode()) for n in ("SSE", "SAD", "HAD", "HAD_fast", "HAD_2SAD")}
        assert (L.vvref_tr_type(b"DCT2"), L.vvref_tr_type(b"DCT8"), L.vvref_tr_type(b"DST7")) == (DCT2, DCT8, DST7)

    def _slot(self, name, n, *a):
        f = getattr(self.L, "vvref_" + name)
        f.restype = None
        a = (self.simd,) + ((n.bit_length() - 1,) if n is not None else ()) + a
        f(*[C.c_void_p(x) if isinstance(x, int) and x > (1 << 32) else x for x in a])

    def dist(self, func, org, cur, w, h, bit_depth=10, sub_shift=0):
        if func == "HAD_2SAD":   # RdCost.cpp:1778 "assumes compact, aligned buffering": the SIMD row uses aligned loads
            org, cur = _aligned(np.ascontiguousarray(org)), _aligned(np.ascontiguousarray(cur))
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        return self.L.vvref_dist(self.simd, self._df[func], po, so, pc, sc, w, h, bit_depth, sub_shift)

    def sad_x5(self, org, cur, w, h, sub_shift=1, calc_centre=True):
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        out = np.zeros(5, np.uint64)
        self.L.vvref_sad_x5(self.simd, po, so, pc, sc, w, h, 10, sub_shift, _p(out), int(calc_centre))
        return out

    def fix_weighted_sse(self, org, cur, w, h, weight):
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        return self.L.vvref_fix_weighted_sse(self.simd, po, so, pc, sc, w, h, 10, weight)

    def sad_mask(self, org, cur, mask, step_x, mask_stride2, w, h, sub_shift=0):
        po, so = self._ptr_stride(org)
        pc, sc = self._ptr_stride(cur)
        pm, sm = self._ptr_stride(mask)
        return self.L.vvref_sad_mask(self.simd, po, so, pc, sc, pm, sm, step_x, mask_stride2, w, h, 10, sub_shift)

    def tr_matrix(self, tr_type, log2n):
        n = 1 << log2n
        out = np.zeros((n, n), np.int16)
        rc = self.L.vvref_tr_matrix(tr_type, log2n, _p(out))
        return out if rc == 0 else None

    def fwd_1d(self, tr_type, log2n, src, shift, line, skip, skip2):
        src = _aligned(np.ascontiguousarray(src, np.int32))
        dst = _aligned(np.zeros((1 << log2n) * line, np.int32))
        rc = self.L.vvref_fwd_1d(self.simd, tr_type, log2n, _p(src), _p(dst), shift, line, skip, skip2)
        assert rc == 0
        return np.array(dst)

    def inv_1d(self, tr_type, log2n, src, shift, line, skip, skip2, cmin=-32768, cmax=32767):
        src = _aligned(np.ascontiguousarray(src, np.int32))
        dst = _aligned(np.zeros((1 << log2n) * line, np.int32))
        rc = self.L.vvref_inv_1d(self.simd, tr_type, log2n, _p(src), _p(dst), shift, line, skip, skip2, cmin, cmax)
        assert rc == 0
        return np.array(dst)

    def xT(self, resi, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10):
        resi = np.ascontiguousarray(resi, np.int16)
        h, w = resi.shape
        coef = np.zeros((h, w), np.int32)
        rc = self.L.vvref_xT(self.simd, _p(resi), w, _p(coef), w, h, tr_hor, tr_ver, bit_depth)
        assert rc == 0, rc
        return coef

    def xIT(self, coef, tr_hor=DCT2, tr_ver=DCT2, bit_depth=10):
        coef = np.ascontiguousarray(coef, np.int32)
        h, w = coef.shape
        resi = np.zeros((h, w), np.int16)
        rc = self.L.vvref_xIT(self.simd, _p(coef), _p(resi), w, w, h, tr_hor, tr_ver, bit_depth)
        assert rc == 0, rc
        return resi

    def scan_order(self, log2w, log2h):
        out = np.zeros(1 << (log2w + log2h), np.uint32)
        self.L.vvref_scan_order(log2w, log2h, _p(out))
        return out

    def quant_scales(self):
        q, iq = np.zeros(12, np.int32), np.zeros(12, np.int32)
        self.L.vvref_quant_scales(_p(q), _p(iq))
        return q.reshape(2, 6), iq.reshape(2, 6)

    def quant_core(self, coef, quant_coeff, q_bits, add, thr_val=8, sign_hiding=False, lfnst_idx=0):
        coef = _aligned(np.ascontiguousarray(coef, np.int32))
        h, w = coef.shape
        q = _aligned(np.zeros((h, w), np.int16))
        du = _aligned(np.zeros(h * w, np.int32))
        s, last = C.c_int32(), C.c_int()
        self.L.vvref_quant_core_lfnst(_p(coef), _p(q), _p(du), w, h, quant_coeff, q_bits, C.c_int64(add), int(sign_hiding), thr_val, int(lfnst_idx), C.byref(s), C.byref(last))
        return np.array(q), np.array(du), s.value, last.value

    def dequant_core(self, q, scale, right_shift, input_max, tr_max=32767):
        q = _aligned(np.ascontiguousarray(q, np.int16))
        h, w = q.shape
        coef = _aligned(np.zeros((h, w), np.int32))
        self.L.vvref_dequant_core(self.simd, w - 1, h - 1, scale, _p(q), w, _p(coef), right_shift, input_max, tr_max)
        return np.array(coef)

    def need_rdoq(self, coef, quant_coeff, offset, shift):
        coef = _aligned(np.ascontiguousarray(coef, np.int32).ravel())
        return int(self.L.vvref_need_rdoq_core(self.simd, _p(coef), coef.size, quant_coeff, offset, shift))

    def mctf_err_int(self, org, buf, w, h, besterror=2 ** 31 - 1):
        po, so = self._ptr_stride(org)
        pb, sb = self._ptr_stride(buf)
        return self.L.vvref_mctf_err_int(self.simd, po, so, pb, sb, w, h, besterror)

    def mctf_err_frac(self, tap4, org, buf, w, h, fx, fy, bit_depth=10, besterror=2 ** 31 - 1):
        po, so = self._ptr_stride(org)
        pb, sb = self._ptr_stride(buf)
        return self.L.vvref_mctf_err_frac(self.simd, int(tap4), po, so, pb, sb, w, h, fx, fy, bit_depth, besterror)

    def mctf_filters(self):
        f8, f4 = np.zeros((16, 8), np.int16), np.zeros((16, 4), np.int16)
        self.L.vvref_mctf_filters(_p(f8), _p(f4))
        return f8, f4

    def mctf_calc_var(self, org, w, h):
        po, so = self._ptr_stride(org)
        return self.L.vvref_mctf_calc_var(self.simd, po, so, w, h)

    def mctf_subsample(self, plane):
        plane = np.ascontiguousarray(plane, np.int16)
        h, w = plane.shape
        out = np.zeros((h // 2, w // 2), np.int16)
        self.L.vvref_mctf_subsample(_p(plane), w, h, _p(out))
        return out

    def mctf_bilateral(self, org, refs, mvs, ref_index, bit_depth=10, qp=32, unit=16, low_res=True, pic_reordering=True, overall_strength=0.95):
        h, w = org[0].shape
        keep = [np.ascontiguousarray(p, np.int16) for p in org] + [np.ascontiguousarray(p, np.int16) for r in refs for p in r]
        op = (C.c_void_p * 3)(*[k.ctypes.data for k in keep[:3]])
        rp = (C.c_void_p * (3 * len(refs)))(*[k.ctypes.data for k in keep[3:]])
        km = [np.ascontiguousarray(m) for m in mvs]
        mp = (C.c_void_p * len(refs))(*[k.ctypes.data for k in km])
        idx = np.ascontiguousarray(ref_index, np.int32)
        outs = [np.zeros_like(p, dtype=np.int16) for p in org]
        outp = (C.c_void_p * 3)(*[o.ctypes.data for o in outs])
        rc = self.L.vvref_mctf_bilateral(self.simd, w, h, bit_depth, qp, unit, int(low_res), int(pic_reordering), op, len(refs), rp, mp, _p(idx),
                                         C.c_double(overall_strength), outp)
        assert rc == 0
        return tuple(outs)

    def mctf_me(self, org, ref, bit_depth=10, unit=16, speed=4, add_level=None):
        return _mctf_me(self.L.vvref_mctf_me, self.simd, org, ref, bit_depth, unit, speed, add_level)

    def mctf_cycle_timed(self, pictures, threads=0, bit_depth=10, qp=32, unit=16, speed=4, add_level=None):
        """what MCTF::filter does for a list of filtered pictures — per picture every reference's motion estimation, then the bilateral filter.  threads == 0: on the calling
        thread; threads > 0: the (picture, reference) estimations as independent jobs on that many threads, the filter on the reference's own thread pool.  Planes are set up
        before its clocks start.  pictures: [(cur (Y, U, V), [ref (Y, U, V), ...], [ref index = |POC offset| - 1, ...], overall strength), ...]
        -> ([final fields per reference] per picture, [filtered (Y, U, V)] per picture, (wall seconds of the motion estimations, of the filters))"""
        h, w = pictures[0][0][0].shape
        if add_level is None:
            add_level = w >= 1920
        keep_c = [np.ascontiguousarray(p, np.int16) for cur, _, _, _ in pictures for p in cur]
        keep_r = [np.ascontiguousarray(p, np.int16) for _, refs, _, _ in pictures for r in refs for p in r]
        cp = (C.c_void_p * len(keep_c))(*[k.ctypes.data for k in keep_c])
        rp = (C.c_void_p * len(keep_r))(*[k.ctypes.data for k in keep_r])
        nrefs = np.array([len(refs) for _, refs, _, _ in pictures], np.int32)
        idx = np.array([i for _, _, ri, _ in pictures for i in ri], np.int32)
        strength = np.array([s for _, _, _, s in pictures], np.float64)
        wb, hb = (w + unit - 1) // unit, (h + unit - 1) // unit
        fields = [np.zeros(wb * hb, MV_DTYPE) for _ in range(int(nrefs.sum()))]
        fp = (C.c_void_p * len(fields))(*[f.ctypes.data for f in fields])
        outs = [np.zeros_like(p, dtype=np.int16) for p in keep_c]
        outp = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        secs = np.zeros(2, np.float64)
        rc = self.L.vvref_mctf_cycle_timed(self.simd, w, h, bit_depth, qp, unit, speed, int(bool(add_level)), int(threads), len(pictures), cp, _p(nrefs), rp, _p(idx), _p(strength),
                                           fp, outp, _p(secs))
        assert rc == 0
        f_by_pic, k = [], 0
        for n in nrefs:
            f_by_pic.append([f.reshape(hb, wb) for f in fields[k:k + int(n)]])
            k += int(n)
        return f_by_pic, [tuple(outs[3 * i:3 * i + 3]) for i in range(len(pictures))], (float(secs[0]), float(secs[1]))


def _aligned(a, align=64):
    """copy `a` into a 64-byte aligned buffer (the reference's SIMD uses aligned loads on transform buffers)"""
    n = a.nbytes
    raw = np.zeros(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + n].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out
