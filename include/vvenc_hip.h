/*
 * vvenc_hip.h — C ABI of libvvenc_hip.so: the MI355X (gfx950) back-end for VVenC's block-level
 * RDO hot path (distortion kernels, transform + scalar quantisation, MCTF block matching).
 *
 * This is the drop-in boundary one level below the reference's C++ function-pointer tables:
 * the table-shaped shim (vvenc_amd/csrc/host/) forwards batches of CU/PU/TU
 * candidates to these entry points.  All bulk pointers are DEVICE pointers (prefix d_) into HBM
 * unless the name says `_host`; every call is asynchronous on the context's HIP stream and ordered
 * with respect to the other calls on the same context.  Results are bit-exact with the
 * reference's scalar and x86-SIMD kernels (citations: paths below /root/reference/source/Lib/).
 *
 * Error convention: every function returns VVHIP_OK (0) or a negative VVHIP_E_* code and records
 * a message retrievable with vvhip_last_error().  The reference has no error codes at this level
 * (programming errors THROW, CommonLib/TypeDef.h:635-636); the shim turns a non-zero return into
 * the same exception.  There is NO CPU fallback anywhere behind this ABI.
 *
 * Types: Pel = int16_t, TCoeff = int32_t, TCoeffSig = int16_t, TMatrixCoeff = int16_t,
 * Distortion = uint64_t (CommonLib/TypeDef.h:181-192).
 */
#ifndef VVENC_HIP_H
#define VVENC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VVHIP_API __attribute__((visibility("default")))

enum {
  VVHIP_OK           = 0,
  VVHIP_E_ARG        = -1,  /* invalid argument (size/type combination the reference THROWs on)  */
  VVHIP_E_HIP        = -2,  /* a HIP runtime call failed; message holds hipGetErrorString          */
  VVHIP_E_NOMEM      = -3,
  VVHIP_E_UNSUPPORTED= -4
};

typedef struct vvhip_ctx vvhip_ctx;

/* ---------------------------------------------------------------------------------------------
 * Context, stream and raw device-memory helpers (so that a C/C++ host needs nothing but this ABI)
 * ------------------------------------------------------------------------------------------- */
VVHIP_API int         vvhip_create( vvhip_ctx** out, int device );     /* own non-blocking stream + ROM tables in HBM */
VVHIP_API void        vvhip_destroy( vvhip_ctx* ctx );
VVHIP_API const char* vvhip_last_error( const vvhip_ctx* ctx );        /* ctx may be NULL: error of vvhip_create       */
VVHIP_API int         vvhip_set_stream( vvhip_ctx* ctx, void* hip_stream ); /* borrow a caller's hipStream_t (NULL = the default stream) */
VVHIP_API int         vvhip_use_own_stream( vvhip_ctx* ctx );          /* back to the context's private stream          */
VVHIP_API void*       vvhip_get_stream( vvhip_ctx* ctx );
VVHIP_API int         vvhip_sync( vvhip_ctx* ctx );                    /* hipStreamSynchronize                          */
VVHIP_API int         vvhip_set_blocking_sync( vvhip_ctx* ctx, int on ); /* on: vvhip_sync and the downloads wait on a blocking event (the thread sleeps) instead of hipStreamSynchronize —
                                                                        * for hosts whose cores are all busy (an encoder's worker threads); default off (lowest latency) */
VVHIP_API int         vvhip_sync_all_devices( vvhip_ctx* ctx );        /* hipDeviceSynchronize on every device, the calling thread's current device restored: before host memory other
                                                                        * contexts may still be copying from is unpinned or released */
/* Launch graphs: the batch entry points only enqueue kernels on the context's stream, so a frame's fixed sequence of calls (the lists of one picture: same tables,
 * same buffers) can be recorded once and replayed with one hipGraphLaunch instead of one dispatch per call.  Between begin and end no call may synchronise, allocate
 * (call every entry point once beforehand so scratch buffers exist) or touch another stream; the stream must not be the legacy default stream (vvhip_use_own_stream).  No counterpart in the reference: its table entries are synchronous calls. */
typedef struct vvhip_graph vvhip_graph;
VVHIP_API int         vvhip_graph_begin( vvhip_ctx* ctx );                         /* hipStreamBeginCapture on the context's stream */
VVHIP_API int         vvhip_graph_end( vvhip_ctx* ctx, vvhip_graph** out );        /* end capture + instantiate                     */
VVHIP_API int         vvhip_graph_launch( vvhip_ctx* ctx, vvhip_graph* graph );    /* replay on the context's stream                */
VVHIP_API void        vvhip_graph_destroy( vvhip_graph* graph );
VVHIP_API int         vvhip_malloc( vvhip_ctx* ctx, void** d_ptr, size_t bytes );
VVHIP_API int         vvhip_free( vvhip_ctx* ctx, void* d_ptr );
VVHIP_API int         vvhip_upload( vvhip_ctx* ctx, void* d_dst, const void* host_src, size_t bytes );   /* async on stream */
VVHIP_API int         vvhip_download( vvhip_ctx* ctx, void* host_dst, const void* d_src, size_t bytes ); /* async + sync    */
VVHIP_API int         vvhip_download_async( vvhip_ctx* ctx, void* host_dst, const void* d_src, size_t bytes ); /* no sync: pair with vvhip_sync; host_dst should be pinned */
/* strided forms (pitches and width in BYTES): a picture window moves without a packing pass on the host (hipMemcpy2DAsync)               */
VVHIP_API int         vvhip_upload_2d( vvhip_ctx* ctx, void* d_dst, size_t dst_pitch, const void* host_src, size_t src_pitch, size_t width_bytes, size_t rows );
VVHIP_API int         vvhip_download_2d( vvhip_ctx* ctx, void* host_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width_bytes, size_t rows ); /* async + sync */
/* Pin a caller-owned host range in place (hipHostRegister) so that uploads / downloads touching it are true asynchronous DMA at full PCIe rate instead of
 * being staged through the runtime's bounce buffers.  The encoder's picture buffers are recycled for the whole run: register once, unregister before free.   */
VVHIP_API int         vvhip_host_register( vvhip_ctx* ctx, const void* host_ptr, size_t bytes );
VVHIP_API int         vvhip_host_unregister( vvhip_ctx* ctx, const void* host_ptr );
/* pinned host memory owned by the library's caller (hipHostMalloc): download / upload areas that are not the encoder's own buffers                                  */
VVHIP_API int         vvhip_host_alloc( vvhip_ctx* ctx, void** host_ptr, size_t bytes );
VVHIP_API int         vvhip_host_free( vvhip_ctx* ctx, void* host_ptr );
/* Completion marks for work a worker thread leaves behind (hipEvent_t without timing): recorded on the recording context's stream, awaited by ANY thread — a picture stage
 * issued piecewise by several workers (the ALF statistics of a picture, one band per row task: EncoderLib/EncSlice.cpp:1135-1167) is collected by the thread that needs the
 * whole (EncAdaptiveLoopFilter::deriveFilter, EncoderLib/EncAdaptiveLoopFilter.cpp:1757) without a stream synchronisation per piece.                                          */
VVHIP_API int         vvhip_event_create( vvhip_ctx* ctx, void** event );
VVHIP_API int         vvhip_event_record( vvhip_ctx* ctx, void* event );          /* on the context's stream */
VVHIP_API int         vvhip_event_wait( vvhip_ctx* ctx, void* event );            /* host wait (ctx: error reporting only; an event never recorded is complete) */
VVHIP_API int         vvhip_event_destroy( vvhip_ctx* ctx, void* event );
/* Several GPUs in one process (one context per device and worker thread): number of devices, the device of a context, and a device-to-device copy of a
 * picture over xGMI (hipMemcpyPeerAsync on dst's stream, ordered after the work already queued on src's stream) — how an original or reconstructed
 * picture reaches the GPU that serves the pictures depending on it without a round trip through the host.                                        */
VVHIP_API int         vvhip_device_count( void );
VVHIP_API int         vvhip_get_device( const vvhip_ctx* ctx );
VVHIP_API int         vvhip_make_current( vvhip_ctx* ctx );              /* hipSetDevice( the context's device ) for the calling thread: call when a thread switches GPUs */
VVHIP_API int         vvhip_copy_peer( vvhip_ctx* dst_ctx, void* d_dst, vvhip_ctx* src_ctx, const void* d_src, size_t bytes );
VVHIP_API const char* vvhip_version( void );

/* ---------------------------------------------------------------------------------------------
 * (A) Distortion kernels — replaces RdCost::m_afpDistortFunc[0][DF_*] / m_afpDistortFuncX5
 *     (CommonLib/RdCost.h:117-121; scalar CommonLib/RdCost.cpp:301-2093; SIMD CommonLib/x86/RdCostX86.h)
 *
 * One call evaluates n candidates of ONE block size with ONE function.  A candidate is a pair of
 * sample offsets relative to d_org / d_cur (the DistParam's org.buf / cur.buf; offsets may be
 * negative = inside the picture margin).  out[i] is what distFunc(DistParam) returns.
 * ------------------------------------------------------------------------------------------- */
enum {                    /* reference table rows (CommonLib/TypeDef.h:339-382)                         */
  VVHIP_DF_SSE      = 0,  /* DF_SSE + log2(w)        xGetSSE*        RdCost.cpp:651-1000               */
  VVHIP_DF_SAD      = 1,  /* DF_SAD + log2(w)        xGetSAD*        RdCost.cpp:301-644  (subShift)    */
  VVHIP_DF_HAD      = 2,  /* DF_HAD + log2(w)        xGetHADs<false> RdCost.cpp:1818-1938              */
  VVHIP_DF_HAD_FAST = 3,  /* DF_HAD_fast + log2(w)   xGetHADs<true>                                    */
  VVHIP_DF_HAD_2SAD = 4   /* DF_HAD_2SAD             xGetHAD2SADs    RdCost.cpp:1768-1816              */
};

typedef struct { int32_t org_off; int32_t cur_off; } vvhip_dist_item;

VVHIP_API int vvhip_dist_batch( vvhip_ctx* ctx, int func,
                                const int16_t* d_org, int org_stride,
                                const int16_t* d_cur, int cur_stride,
                                int width, int height, int sub_shift, int bit_depth,
                                const vvhip_dist_item* d_items, int n, uint64_t* d_out );

/* Several batches of ONE function over the same plane pair (e.g. all block sizes of a frame's SAD work list) in as few launches as
 * possible: semantically n_jobs x vvhip_dist_batch; jobs the merged kernels cannot take run as separate launches.  `jobs` is a HOST array. */
typedef struct { int32_t width, height, sub_shift, n; const vvhip_dist_item* d_items; uint64_t* d_out; } vvhip_dist_job;
VVHIP_API int vvhip_dist_multi( vvhip_ctx* ctx, int func, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride, int bit_depth,
                                const vvhip_dist_job* jobs_host, int n_jobs );

/* The same with a function per job: consecutive jobs of one kernel family — SAD and SSE, or HAD and HAD_fast — share a launch (a frame's
 * SAD and SSE lists are both short, memory-side work; together they fill the device better).
 * flags: VVHIP_DIST_FLAG_SAMPLES = the caller asserts that BOTH operands of the job hold samples in [0, 2^bit_depth) (original vs reconstructed /
 * predicted picture samples).  Without it the Hadamard jobs accept everything the encoder hands to a HAD table entry at that bit depth, including
 * the bi-prediction pattern 2*org - pred (values -(2^bd - 1) .. 2*(2^bd - 1), EncoderLib/InterSearch.cpp:1996-2003), through a slightly wider tile;
 * results are identical wherever both forms are defined.  An SSE job with the flag (bit depth <= 12; here the flag may also stand for residuals, |operand| < 2^bit_depth)
 * squares packed 16-bit differences (v_dot2_i32_i16).  vvhip_dist_multi (no flags field) always uses the general form.                 */
#define VVHIP_DIST_FLAG_SAMPLES 1
typedef struct { int32_t func, width, height, sub_shift, n, flags; const vvhip_dist_item* d_items; uint64_t* d_out; } vvhip_dist_fjob;
VVHIP_API int vvhip_dist_multi_func( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride, int bit_depth,
                                     const vvhip_dist_fjob* jobs_host, int n_jobs );

/* The same lists on 8x8-TILED copies of the planes (no counterpart in the reference: a memory-layout decision for the MI355X).  A 128-byte cache line holds one 8x8 tile of
 * int16 samples; in a row-major plane an 8x8 block is eight 16-byte pieces in eight lines, each its own L1 access (29 accesses per 8x8 SAD candidate, 10.5 on the tiled
 * copy: the lists run at the L1's access rate).  vvhip_plane_tile8 makes the tiled copy of a padded plane (d_base = first sample of the padded plane, `rows` lines of `stride` samples; d_tiled holds
 * vvhip_tiled8_elems( stride, rows ) samples; re-tile when the plane changes).  vvhip_dist_multi_func_tiled = vvhip_dist_multi_func, identical results, with the 8x8 SAD / SSE jobs
 * of bit depths <= 10 reading the tiled copies (the Hadamard jobs were measured slower there and stay on the row-major planes) (tiled may be NULL: no difference to vvhip_dist_multi_func; either tiled pointer may be NULL: those jobs stay on the row-major planes).  *_margin = samples of margin around
 * sample (0,0) of the plane the tiled copy was made from (d_org / d_cur still address sample (0,0) of the row-major planes, used by the other jobs).                          */
typedef struct { const int16_t* d_org_tiled; const int16_t* d_cur_tiled; int32_t org_margin, cur_margin;
                 const int16_t* d_cur_shift1;   /* optional (NULL: none): the sample of a ONE-SAMPLE-SHIFTED copy of the reference plane buffer that corresponds to d_cur, i.e.
                                                   d_cur_shift1[k] == d_cur[k + 1] for every sample of the padded plane (vvhip_plane_shift1 over the whole buffer).  Candidates at odd
                                                   sample addresses (half of all motion vectors) are then read from the copy with dword-aligned loads — an only 2-byte-aligned
                                                   16-byte lane load is split by the memory pipeline and costs 1.8x; every job on the row-major planes uses it */
               } vvhip_tiled_planes;
/* d_dst[i] = d_src[i + 1] for i < elems - 1, d_dst[elems - 1] = 0; both buffers with the same alignment modulo 4 bytes */
VVHIP_API int vvhip_plane_shift1( vvhip_ctx* ctx, const int16_t* d_src, size_t elems, int16_t* d_dst );
/* all three derived copies of a picture's planes in one launch (what a caller does once per picture): the 8x8-tiled copy of the original plane, the 8x8-tiled and the one-sample-shifted
 * copy of the reference plane.  *_base = first sample of the padded plane buffer, `rows` lines of `stride` samples; any output may be NULL (skipped).
 * d_cur_shift1 covers the whole buffer (stride * rows samples) like vvhip_plane_shift1. */
VVHIP_API int vvhip_planes_derive( vvhip_ctx* ctx, const int16_t* d_org_base, int org_stride, int org_rows, int16_t* d_org_tiled,
                                   const int16_t* d_cur_base, int cur_stride, int cur_rows, int16_t* d_cur_tiled, int16_t* d_cur_shift1 );
VVHIP_API size_t vvhip_tiled8_elems( int stride, int rows );
VVHIP_API int vvhip_plane_tile8( vvhip_ctx* ctx, const int16_t* d_base, int stride, int rows, int16_t* d_tiled );
VVHIP_API int vvhip_dist_multi_func_tiled( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride, const vvhip_tiled_planes* tiled_host, int bit_depth,
                                           const vvhip_dist_fjob* jobs_host, int n_jobs );

/* DMVR 5-position SAD: RdCost::xGetSAD8X5 / xGetSAD16X5 (RdCost.cpp:1984-2034). width 8 or 16.
 * d_out5[5*i+k] = SAD(org+k, cur-k) >> 1; entry 2 is left untouched when calc_centre == 0.       */
VVHIP_API int vvhip_sad_x5_batch( vvhip_ctx* ctx,
                                  const int16_t* d_org, int org_stride,
                                  const int16_t* d_cur, int cur_stride,
                                  int width, int height, int sub_shift, int calc_centre,
                                  const vvhip_dist_item* d_items, int n, uint64_t* d_out5 );

/* GEO masked SAD: RdCost::xGetSADwMask (RdCost.cpp:2062-2093), table slot DF_SAD_WITH_MASK (TypeDef.h:339-382).
 *   d_out[i] = ( sum over processed rows r, columns x of |org - cur| * mask[...] ) << sub_shift
 * with the mask walked exactly as the reference does: +step_x per sample, + mask_stride*(1<<sub_shift) + mask_stride2 per
 * processed row, starting at d_mask + d_mask_off[i] (d_mask_off may be NULL: every candidate starts at d_mask).               */
VVHIP_API int vvhip_sad_mask_batch( vvhip_ctx* ctx,
                                    const int16_t* d_org, int org_stride,
                                    const int16_t* d_cur, int cur_stride,
                                    const int16_t* d_mask, int mask_stride, int step_x, int mask_stride2,
                                    int width, int height, int sub_shift, int bit_depth,
                                    const vvhip_dist_item* d_items, const int32_t* d_mask_off, int n, uint64_t* d_out );

/* Fixed-weight SSE: RdCost::m_fxdWtdPredPtr = fixWeightedSSE_Core (RdCost.cpp:1948-1982, RdCost.h:117).
 *   d_out[i] = sum ( int )( ( weight_i * d*d + 2^15 ) >> 16 ),  width even or 1.                                               */
VVHIP_API int vvhip_fix_weighted_sse_batch( vvhip_ctx* ctx,
                                            const int16_t* d_org, int org_stride,
                                            const int16_t* d_cur, int cur_stride,
                                            int width, int height, int bit_depth,
                                            const vvhip_dist_item* d_items, const uint32_t* d_weights, int n, uint64_t* d_out );

/* Full-window SAD cost surface for one block size: for block b (top-left org sample offset
 * d_block_org_off[b], co-located reference offset d_block_ref_off[b]) and every integer displacement
 * (dx,dy), |dx| <= range_x, |dy| <= range_y:
 *   d_out[(b*(2*range_y+1) + dy+range_y)*(2*range_x+1) + dx+range_x] = xGetSAD(org, ref + dy*stride + dx)
 * (same subShift rule).  Exact integers, so any subset a host search asks for is bit-exact
 * (SURVEY §7 "cost-surface precompute").  The reference window is staged once in LDS.            */
VVHIP_API int vvhip_sad_surface( vvhip_ctx* ctx,
                                 const int16_t* d_org, int org_stride,
                                 const int16_t* d_ref, int ref_stride,
                                 int width, int height, int sub_shift,
                                 int range_x, int range_y,
                                 const int32_t* d_block_org_off, const int32_t* d_block_ref_off, int n_blocks,
                                 uint32_t* d_out );

/* ---------------------------------------------------------------------------------------------
 * (B) Transforms + scalar quantisation — replaces g_tCoeffOps.* as driven by TrQuant::xT / xIT
 *     (CommonLib/TrQuant.cpp:481-655, TrQuant_EMT.cpp:152-420,1917-2000) and Quant::xQuant /
 *     xDeQuant / xNeedRdoq (CommonLib/Quant.cpp:132-278 with the parameter derivation of
 *     Quant::quant :735-833, Quant::dequant :520-610, Quant::xNeedRDOQ :835-891).
 *
 * A batch holds n TUs of ONE size and ONE transform-type pair.  TU i reads/writes its residual
 * at d_resi + d_resi_off[i] (stride resi_stride); coefficient / level arrays are compact,
 * n * width * height, TU-major, raster inside the TU (what TrQuant::m_plTempCoeff holds).
 * ------------------------------------------------------------------------------------------- */
enum { VVHIP_DCT2 = 0, VVHIP_DCT8 = 1, VVHIP_DST7 = 2 };   /* TransType, CommonLib/TypeDef.h */

VVHIP_API int vvhip_fwd_transform_batch( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* d_resi_off,
                                         int n, int width, int height, int tr_hor, int tr_ver, int bit_depth,
                                         int32_t* d_coef );
VVHIP_API int vvhip_inv_transform_batch( vvhip_ctx* ctx, const int32_t* d_coef,
                                         int n, int width, int height, int tr_hor, int tr_ver, int bit_depth,
                                         int16_t* d_resi, int resi_stride, const int32_t* d_resi_off );

/* per-TU quantiser control: qp = QpParam::Qp (base QP incl. qpBdOffset); flags bit0 = slice isIRAP
 * (iAdd 171 instead of 85, Quant.cpp:775), bit1 = luma (need-RDOQ offset 171 instead of 256, :874)   */
typedef struct { int16_t qp; int16_t flags; } vvhip_tu_qp;

VVHIP_API int vvhip_quant_batch( vvhip_ctx* ctx, const int32_t* d_coef, int n, int width, int height, int bit_depth,
                                 const vvhip_tu_qp* d_qp, int thr_val,
                                 int16_t* d_level, int32_t* d_delta_u /* may be NULL */,
                                 int32_t* d_abs_sum, int32_t* d_last_scan_pos );
VVHIP_API int vvhip_dequant_batch( vvhip_ctx* ctx, const int16_t* d_level, int n, int width, int height, int bit_depth,
                                   const vvhip_tu_qp* d_qp, int32_t* d_coef );
VVHIP_API int vvhip_need_rdoq_batch( vvhip_ctx* ctx, const int32_t* d_coef, int n, int width, int height, int bit_depth,
                                     const vvhip_tu_qp* d_qp, uint8_t* d_need );

/* Raw-parameter forms with exactly the argument lists of the reference's table entries Quant::xDeQuant / Quant::xNeedRdoq
 * (CommonLib/Quant.h:143-151; QuantCore / DeQuantCore / needRdoqCore, Quant.cpp:132-278;
 * vvhip_quant_core takes QuantCore's defaultQuantisationCoefficient, iQBits, iAdd, m_thrVal; lfnstIdx == 0 — vvhip_quant_core_lfnst any), for the table-shaped shim: ONE block per call.
 * d_level is strided (level_stride), d_coef compact (max_x+1) x (max_y+1); *d_need receives 0/1.                            */
VVHIP_API int vvhip_dequant_core( vvhip_ctx* ctx, int max_x, int max_y, int scale, const int16_t* d_level, size_t level_stride, int32_t* d_coef,
                                  int right_shift, int input_maximum, int32_t transform_maximum );
VVHIP_API int vvhip_quant_core( vvhip_ctx* ctx, const int32_t* d_coef, int width, int height, int quant_coeff, int q_bits, int64_t add, int thr_val,
                                int16_t* d_level /* w*h compact */, int32_t* d_delta_u /* may be NULL */, int32_t* d_abs_sum, int32_t* d_last_scan_pos );
/* QuantCore for a TU of a coding unit with lfnst_idx > 0 (CodingUnit::lfnstIdx, presets fast / medium): only the first coefficient group is quantised, its first 8 scan
 * positions for 4x4 and 8x8 TUs (Quant.cpp:149-159); lfnst_idx == 0 is vvhip_quant_core.  Replaces the same table entry (Quant::xQuant, Quant.h:143-146).            */
VVHIP_API int vvhip_quant_core_lfnst( vvhip_ctx* ctx, const int32_t* d_coef, int width, int height, int quant_coeff, int q_bits, int64_t add, int thr_val, int lfnst_idx,
                                      int16_t* d_level /* w*h compact */, int32_t* d_delta_u /* may be NULL */, int32_t* d_abs_sum, int32_t* d_last_scan_pos );
VVHIP_API int vvhip_need_rdoq_core( vvhip_ctx* ctx, const int32_t* d_coef, size_t num_coeff, int quant_coeff, int64_t offset, int shift, uint8_t* d_need );

/* Fused TU pipeline of the residual RDO loop (InterSearch::xEstimateInterResidualQT,
 * EncoderLib/InterSearch.cpp:3663-3714): xT -> xNeedRDOQ + QuantCore -> DeQuantCore -> xIT -> SSE(resi, rec).
 * Reads 2*w*h bytes, writes 2*w*h (levels) + 2*w*h (reconstructed residual) + 24 bytes of statistics per TU;
 * intermediate coefficients never leave the CU (LDS).  Any output pointer may be NULL.          */
typedef struct { int32_t abs_sum; int32_t last_scan_pos; int32_t need_rdoq; int32_t pad; uint64_t sse; } vvhip_tu_stats;

VVHIP_API int vvhip_tu_rdo_batch( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* d_resi_off,
                                  int n, int width, int height, int tr_hor, int tr_ver, int bit_depth,
                                  const vvhip_tu_qp* d_qp, int thr_val,
                                  int16_t* d_level, int16_t* d_rec_resi /* compact n*w*h */, vvhip_tu_stats* d_stats );

/* A frame's TU work lists in one call: every job is one vvhip_tu_rdo_batch argument set on the same residual plane.  Square 8/16/32
 * TUs (any transform types) are merged into ONE launch, largest size first — the per-size launches are each too small to fill
 * the device; other shapes run as individual launches.  Results are identical to per-job vvhip_tu_rdo_batch calls.               */
typedef struct
{
  int32_t width, height, tr_hor, tr_ver, n, thr_val;
  const int32_t*     d_resi_off;
  const vvhip_tu_qp* d_qp;
  int16_t*           d_level;
  int16_t*           d_rec_resi;
  vvhip_tu_stats*    d_stats;
} vvhip_tu_job;
VVHIP_API int vvhip_tu_rdo_multi( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, int bit_depth, const vvhip_tu_job* jobs_host, int n_jobs );
/* The same with a residual row pitch PER JOB (resi_strides_host[i] for jobs_host[i]): lists whose residual blocks are stored compactly, one after the other (pitch = width), as a
 * caller that gathers the TUs of many CUs into one buffer has them — e.g. every TU InterSearch::xEstimateInterResidualQT tests for a picture (EncoderLib/InterSearch.cpp:3663-3714). */
VVHIP_API int vvhip_tu_rdo_multi_strided( vvhip_ctx* ctx, const int16_t* d_resi, const int32_t* resi_strides_host, int bit_depth, const vvhip_tu_job* jobs_host, int n_jobs );

/* Sparse outputs of vvhip_tu_rdo_multi / _multi_strided (default off = every output of every TU is written).  On: for a TU whose levels are ALL ZERO (stats.abs_sum == 0)
 * d_level and d_rec_resi are UNSPECIFIED — left untouched wherever the matrix-core launch takes a whole 32x32 tile (or a 64x64 TU) through its all-zero shortcut, zeros
 * otherwise — and the caller treats them as zero.  That is what the reference's caller does: xEstimateInterResidualQT reads neither the levels nor the reconstruction of a TU
 * whose uiAbsSum is 0 (EncoderLib/InterSearch.cpp:3696-3714: invTransformNxN only runs for a non-zero abs sum).  Statistics (abs_sum, last_scan_pos, need_rdoq, sse) are always
 * written.  An all-zero tile then moves 2 w h bytes in and 24 bytes per TU out instead of 6 w h + 24.                                                                          */
VVHIP_API int vvhip_tu_set_sparse_outputs( vvhip_ctx* ctx, int on );

/* ---- the g_tCoeffOps table slots one-to-one (CommonLib/TrQuant_EMT.h:63-91), device pointers, caller's matrix ------------------
 * vvhip_fast_fwd_core  <- fastFwdCore_2D/_1D[log2(tr_size)-2]  (TrQuant_EMT.cpp:1973-2000):
 *     dst[j*line + i] = ( sum_k src[i*tr_size + k] * tc[j*tr_size + k] + 2^(shift-1) ) >> shift,  i < reduced_line, j < cutoff
 * vvhip_fast_inv_core  <- fastInvCore[log2(tr_size)-2]         (:1953-1970): dst[i*tr_size + j] += sum_{k<rows} src[k*lines + i] * it[k*tr_size + j]
 *     (ACCUMULATES: the reference's caller zeroes dst first, :159)
 * vvhip_round_clip     <- roundClip4/8 (clipCore :1941-1950), vvhip_cpy_resi <- cpyResi4/8 (:1929-1938), vvhip_cpy_coeff <- cpyCoeff4/8 (:1917-1926).
 * The batched path uses the fused 2-D entries above; these give every slot of the table a device provider.                     */
VVHIP_API int vvhip_fast_fwd_core( vvhip_ctx* ctx, int tr_size, const int16_t* d_tc, const int32_t* d_src, int32_t* d_dst,
                                   unsigned line, unsigned reduced_line, unsigned cutoff, int shift );
VVHIP_API int vvhip_fast_inv_core( vvhip_ctx* ctx, int tr_size, const int16_t* d_it, const int32_t* d_src, int32_t* d_dst,
                                   unsigned lines, unsigned reduced_lines, unsigned rows );
VVHIP_API int vvhip_round_clip( vvhip_ctx* ctx, int32_t* d_dst, unsigned width, unsigned height, unsigned stride,
                                int32_t out_min, int32_t out_max, int32_t round, int32_t shift );
VVHIP_API int vvhip_cpy_resi( vvhip_ctx* ctx, const int32_t* d_src, int16_t* d_dst, ptrdiff_t stride, unsigned width, unsigned height );
VVHIP_API int vvhip_cpy_coeff( vvhip_ctx* ctx, const int16_t* d_src, ptrdiff_t stride, int32_t* d_dst, unsigned width, unsigned height );

/* ======================================================================================================================
 * SURVEY 8f rank 1 — sub-pel interpolation for fractional motion estimation / motion compensation
 * ====================================================================================================================== */
/* One pass of the separable interpolation on one block with the caller's taps: the table slots
 * InterpolationFilter::m_filterHor / m_filterVer [tap index][isFirst][isLast] (CommonLib/InterpolationFilter.h:113-114),
 * scalar core filter<N,isVertical,isFirst,isLast> (InterpolationFilter.cpp:356-441).  taps = 8, 6, 4 or 2 (bilinear); coeff_host is the
 * table ROW as the reference passes it (8 entries for 8 and 6 taps — the 6-tap core skips the first entry —, 4 for 4 taps).  */
VVHIP_API int vvhip_if_filter( vvhip_ctx* ctx, int taps, int is_vertical, int is_first, int is_last, int bit_depth,
                               const int16_t* d_src, int src_stride, int16_t* d_dst, int dst_stride, int width, int height,
                               const int16_t* coeff_host );
/* m_filterCopy[isFirst][isLast] (InterpolationFilter.h:115, filterCopy :255-333). */
VVHIP_API int vvhip_if_copy( vvhip_ctx* ctx, int is_first, int is_last, int bit_depth,
                             const int16_t* d_src, int src_stride, int16_t* d_dst, int dst_stride, int width, int height, int bi_mc_for_dmvr );

/* A sub-pel candidate: original block at org_off, reference block whose integer position is ref_off, vector fraction in 1/16 sample. */
typedef struct { int32_t org_off, ref_off; int16_t frac_x, frac_y; } vvhip_subpel_item;

/* Motion-compensated luma prediction blocks (compact, d_out[i*w*h + y*w + x]): the passes InterPredInterpolation::xPredInterBlk runs
 * (CommonLib/InterPrediction.cpp:832-865; no BDOF/DMVR) — frac_y == 0: horizontal only, frac_x == 0: vertical only, else horizontal
 * (14-bit intermediate) then vertical.  rnd_res = !bi (InterPrediction.cpp:775): 1 = final samples clipped to the bit depth, 0 = the
 * 14-bit intermediate a bi-prediction average consumes.
 * filter_mode 0: the motion-compensation taps (8-tap m_lumaFilter; 6-tap m_lumaFilter4x4 for 4x4 blocks);
 *             1 / 2: the reduced sets of the fast sub-pel search (m_meReduceTap 1 / 2: m_lumaFilter4x4 / m_chromaFilter[frac << 1],
 *             InterpolationFilter.cpp:586-593).  use_alt_hpel: m_lumaAltHpelIFilter at phase 8 (IMV_HPEL).                       */
VVHIP_API int vvhip_interp_luma_batch( vvhip_ctx* ctx, const int16_t* d_ref, int ref_stride, const vvhip_subpel_item* d_items, int n,
                                       int width, int height, int bit_depth, int rnd_res, int filter_mode, int use_alt_hpel,
                                       int16_t* d_out );

/* What InterSearch::xPatternRefinement computes per tested position (EncoderLib/InterSearch.cpp:850-873, minus the MV-bit cost the
 * host adds): distortion (func = VVHIP_DF_SAD / _HAD / _HAD_FAST / _SSE, subShift 0) between the original block and the block
 * interpolated at the candidate's fractional vector.  One call = all sub-pel candidates of a stage for many blocks.               */
VVHIP_API int vvhip_subpel_dist_batch( vvhip_ctx* ctx, int func, const int16_t* d_org, int org_stride, const int16_t* d_ref, int ref_stride,
                                       int width, int height, int bit_depth, int filter_mode, int use_alt_hpel,
                                       const vvhip_subpel_item* d_items, int n, uint64_t* d_out );

/* Pattern refinement in one call: for every block, n_offsets sub-pel positions around ITS base vector (d_bases[b]: original block, integer
 * position and fraction of the base vector) — candidate k is base + offsets_host[k] (dx, dy in 1/16 sample, |.| <= 16) —
 *   d_out[b * n_offsets + k] = distortion( original block, block interpolated at candidate k )       (func as vvhip_subpel_dist_batch)
 * i.e. one stage of InterSearch::xPatternRefinement (EncoderLib/InterSearch.cpp:760-880: the 8 half-sample neighbours, then the 8
 * quarter-sample neighbours) for many blocks.  Same values as vvhip_subpel_dist_batch on the expanded candidate list; the reference
 * window is staged in LDS once per block and the horizontal pass is shared between positions with the same
 * horizontal offset; the distortion kernels then score the prediction blocks.  width a multiple of 8, blocks up to 64x64; the
 * reference plane needs 5 samples of margin beyond every block.                                                                                                       */
VVHIP_API int vvhip_subpel_refine_batch( vvhip_ctx* ctx, int func, const int16_t* d_org, int org_stride, const int16_t* d_ref, int ref_stride,
                                         int width, int height, int bit_depth, int filter_mode, int use_alt_hpel,
                                         const vvhip_subpel_item* d_bases, int n_blocks, const int16_t* offsets_host, int n_offsets, uint64_t* d_out );

/* ======================================================================================================================
 * Motion-search plans (round 3): what ONE picture's inter search pushes through the distortion tables, in ONE launch.
 * Built for work lists with the shape the encoder really produces (recorded from it, vvenc_amd/recorded.py): per InterSearch::xMotionEstimation call
 * (EncoderLib/InterSearch.cpp:1976-2130) ~20 integer candidates inside a few samples of each other (start points, 4-point diamond + square at distance 1,
 * :2385-2410) followed by one or two xPatternRefinement stages (:760-880) of <= 9 sub-pel positions around the winner; blocks 4..64 square at the fast presets (most sample pairs in
 * 64x64 blocks), rectangular 4..128 at preset medium; plus merge / AMVP / intra / residual distortions on compact prediction blocks.
 *   integer job  : the block's candidate positions as displacements from ref_off; the kernel stages the candidates' bounding window of the reference plane in
 *                  LDS once (samples biased for v_sad_u16; odd displacements are taken from the even dword below with v_alignbit) and scores all candidates from it.
 *                  cost[first_cand + i] = xGetSAD*( org block, ref block at (dx_i, dy_i) ) incl. the subShift rule (RdCost.cpp:301-644).
 *   stage job    : position k (k = 0..8, evaluated when bit k of mask is set) lies ( refine[k] + (base_qx, base_qy) ) * i_frac quarter samples from the block at
 *                  ref_off, refine = s_acMvRefineH (i_frac 2) / s_acMvRefineQ (i_frac 1) (InterSearch.cpp:67-91).  The kernel interpolates like
 *                  xPatternRefinement's planes (filter_mode / alt_hpel as vvhip_interp_luma_batch; horizontal pass shared between positions) and scores each
 *                  prediction against the original block directly from LDS — the predictions never exist in HBM.  cost[9 * stage + k] as the table entry
 *                  `func` (VVHIP_DF_SAD / _HAD / _HAD_FAST) returns it; positions outside the mask read 0 (every stage's nine costs are written by every run).
 *                  Every evaluated position must lie within one sample of the block at ref_off vertically and horizontally ((refine + base) * i_frac in -4..4
 *                  quarter samples): the kernel stages exactly that neighbourhood; plan creation rejects anything else.
 *   item         : one plain table call: func on (org block, cur block) of any two planes / pools of the plan's plane table.
 *   masked item  : DF_SAD_WITH_MASK (xGetSADwMask, RdCost.cpp:2062-2093: GEO partitions of preset medium and slower): sum |org - cur| * mask over the rows the
 *                  subShift rule keeps, << subShift.  The mask block is COMPACT (row pitch = width, one row per evaluated row: the caller has applied the
 *                  reference's stepX / maskStride / maskStride2 walk).  Its costs follow the plain items': d_item_cost[n_items + i].
 * Offsets are in samples from sample (0,0) of the job's plane (negative = margin); planes must be readable 16 bytes beyond every block / window row they hold.
 * A plane table entry with stride 0 is a pool of COMPACT blocks: the row pitch of a block read from it is the block's own width.
 * Shapes (round 4: everything preset medium's CTU 128 + multi-type tree produces): width and height independent powers of two — integer jobs 8..128 x 4..128,
 * stage jobs 4..128 x 4..128 (not 4x4), items 2..128 x 2..128; the Hadamard family follows the reference's tile ladder (16x8, 8x16, 8x4, 4x8 with the
 * double-precision normalisation, 16x16_fast, 8x8, 4x4, 2x2: RdCost.cpp:1818-1938); bit depths <= 10 (the packed Hadamard tile, like the reference's x86 rows).
 * Operand ranges of the Hadamard family (stage jobs and items): what the encoder hands to these table entries at the plan's bit depth — samples in [0, 2^bit_depth) or
 * bi-prediction patterns 2 org - pred in (-2^bit_depth, 2^(bit_depth+1)) against prediction samples, i.e. |org - cur| < 2^(bit_depth+1): the first two butterfly stages run on
 * packed 16-bit pairs.  SAD / SSE / masked-SAD items take any int16 operands (masked SADs: weights >= 0; sums in 64 bits outside the GEO range of samples and weights).
 * Limits: fewer than 2^24 candidates and 2^22 stage jobs per plan.
 * Integer jobs: positions a job lists more than once (the search re-scores its start point) are scored once; every listed candidate still gets its cost.
 * A plan owns device copies of the job tables and the schedule derived from them (which wave takes which jobs, heaviest first); running it is one launch per kind.
 * ====================================================================================================================== */
typedef struct { const int16_t* d_base; int32_t stride; int32_t reserved; } vvhip_me_plane;
typedef struct { int32_t org_off, ref_off; int16_t width, height; uint8_t org_plane, ref_plane, sub_shift, reserved; int16_t min_dx, min_dy, win_w, win_h; int32_t first_cand, n_cand; } vvhip_me_int_job;
typedef struct { int16_t dx, dy; } vvhip_me_cand;
typedef struct { int32_t org_off, ref_off; int16_t width, height; uint8_t org_plane, ref_plane, i_frac, filter_mode, alt_hpel, func; int8_t base_qx, base_qy; uint16_t mask; uint16_t reserved; } vvhip_me_stage_job;
typedef struct { int32_t org_off, cur_off; uint8_t org_plane, cur_plane, func, sub_shift; int16_t width, height; } vvhip_me_item;
typedef struct { int32_t org_off, cur_off, mask_off; uint8_t org_plane, cur_plane, mask_plane, sub_shift; int16_t width, height; int32_t reserved; } vvhip_me_mask_item;
typedef struct
{
  const vvhip_me_int_job* int_jobs; int32_t n_int_jobs; const vvhip_me_cand* cands; int32_t n_cands;
  const vvhip_me_stage_job* stage_jobs; int32_t n_stage_jobs; const vvhip_me_item* items; int32_t n_items;
  const vvhip_me_mask_item* mask_items; int32_t n_mask_items;
} vvhip_me_lists;
typedef struct vvhip_me_plan vvhip_me_plan;
/* all job arrays are HOST arrays (copied); any of the three lists may be empty.  win_* / min_* of an integer job may be left 0: the library derives the window from the candidates
 * and splits jobs whose candidates spread further than max_window samples (0: default 24) into several windows. */
VVHIP_API int  vvhip_me_plan_create( vvhip_ctx* ctx, const vvhip_me_int_job* int_jobs, int n_int_jobs, const vvhip_me_cand* cands, int n_cands,
                                     const vvhip_me_stage_job* stage_jobs, int n_stage_jobs, const vvhip_me_item* items, int n_items,
                                     int bit_depth, int max_window, vvhip_me_plan** out );
/* the same with every list in one record (adds the masked items) */
VVHIP_API int  vvhip_me_plan_create_lists( vvhip_ctx* ctx, const vvhip_me_lists* lists, int bit_depth, int max_window, vvhip_me_plan** out );
VVHIP_API void vvhip_me_plan_destroy( vvhip_ctx* ctx, vvhip_me_plan* plan );
/* planes_host: the plan's plane table for THIS run (<= 16 entries; device pointers to sample (0,0)).  d_cand_cost: n_cands, d_stage_cost: 9 * n_stage_jobs,
 * d_item_cost: n_items + n_mask_items Distortion values; pointers of empty lists may be NULL.                                                               */
VVHIP_API int  vvhip_me_plan_run( vvhip_ctx* ctx, const vvhip_me_plan* plan, const vvhip_me_plane* planes_host, int n_planes,
                                  uint64_t* d_cand_cost, uint64_t* d_stage_cost, uint64_t* d_item_cost );
/* the same, restricted to some of the plan's three independent parts (bit 0: refinement stages, bit 1: integer windows, bit 2: table calls): a caller with several
 * contexts / streams issues the parts on different streams so that they share the device (bench.py) */
VVHIP_API int  vvhip_me_plan_run_parts( vvhip_ctx* ctx, const vvhip_me_plan* plan, const vvhip_me_plane* planes_host, int n_planes,
                                        uint64_t* d_cand_cost, uint64_t* d_stage_cost, uint64_t* d_item_cost, int parts );
/* per-kernel HIP events inside vvhip_me_plan_run (measurements: bench.py's roofline): with timing on, vvhip_me_plan_last_times waits for the last run and returns the
 * milliseconds of its parts — [0] refinement-stage kernel(s), [1] integer windows (both LDS classes: one launch), [2] 0 (was: the small
 * windows' own launch), [3] table calls. */
VVHIP_API int  vvhip_me_plan_set_timing( vvhip_ctx* ctx, vvhip_me_plan* plan, int on );
VVHIP_API int  vvhip_me_plan_last_times( vvhip_ctx* ctx, const vvhip_me_plan* plan, float* ms4_host );
/* what the schedule looks like (for measurements): waves per list and LDS bytes per wave */
VVHIP_API int  vvhip_me_plan_info( const vvhip_me_plan* plan, int* waves_int, int* waves_stage, int* waves_item, int* lds_bytes );

/* ROM accessors (host memory out): the tables the kernels use, for parity checks against
 * g_trCore* (CommonLib/RomTr.cpp:364-449) and getScanOrder (CommonLib/Rom.h:104).               */
VVHIP_API int vvhip_get_tr_matrix_host( int tr_type, int log2_size, int16_t* host_out );
VVHIP_API int vvhip_get_scan_order_host( int log2_w, int log2_h, uint32_t* host_out );
/* the six tap tables of a motion-search plan's refinement-stage kernels at this bit depth (8..10), 6 x 192 dwords, table = filter_mode * 2 + alt_hpel:
 * [0..127] 16 phases x 8 window taps (tap k multiplies the sample at offset k - 3) of the SECOND pass, scaled by 2^(16 - shift2), shift2 = 6 + headRoom, headRoom = 14 - bit_depth
 * (InterpolationFilter.cpp:394-400: the filtered sample is the upper half of the 32-bit sum); [128..191] 16 phases x 4 packed int16 pairs (taps K0 + 2i, K0 + 2i + 1 of the table's
 * tap support: 4-tap search set 2..5, 6 taps / alternative half-sample filter 1..6, 8 taps 0..7) of the FIRST pass, scaled by 2^(8 - shift1), shift1 = 6 - headRoom (:401-408).     */
VVHIP_API int vvhip_get_me_tap_tables_host( int bit_depth, int32_t* host_out );

/* ---------------------------------------------------------------------------------------------
 * (C) MCTF block matching — replaces MCTF::m_motionErrorLumaInt8 / m_motionErrorLumaFrac8[2] /
 *     m_calcVar and the search schedule around them (CommonLib/MCTF.h:160-170,
 *     CommonLib/MCTF.cpp:122-257, 520-546, 1072-1397).
 * ------------------------------------------------------------------------------------------- */
typedef struct { int32_t org_off; int32_t buf_off; int16_t fx; int16_t fy; } vvhip_mctf_item;

/* out[i] = motionErrorLumaInt (fx==fy==0) or motionErrorLumaFrac{6,4} with the reference's
 * 1/16-pel filter rows fx, fy (tap4 != 0: m_interpolationFilter4, else m_interpolationFilter8).
 * The full error is returned (the reference's `> besterror` early exit is semantically inert,
 * MCTF.cpp:138-141 and callers :1201,:1223).                                                     */
VVHIP_API int vvhip_mctf_error_batch( vvhip_ctx* ctx, const int16_t* d_org, int org_stride,
                                      const int16_t* d_buf, int buf_stride, int width, int height,
                                      int tap4, int bit_depth, const vvhip_mctf_item* d_items, int n, int32_t* d_out );

/* calcVarCore (MCTF.cpp:520-546) for n blocks of one size: out[i] = variance * 256 as int64
 * (the double the reference returns is exactly out[i] / 256.0).                                   */
VVHIP_API int vvhip_mctf_calc_var_batch( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, int width, int height,
                                         const int32_t* d_off, int n, int64_t* d_out_x256 );

/* MCTF::subsampleLuma (MCTF.cpp:1072-1097) incl. the border extension by `pad` samples.
 * d_dst points at sample (0,0) of a plane with >= pad samples of margin on every side.           */
VVHIP_API int vvhip_mctf_subsample( vvhip_ctx* ctx, const int16_t* d_src, int src_stride, int src_width, int src_height,
                                    int16_t* d_dst, int dst_stride, int pad );
VVHIP_API int vvhip_extend_border( vvhip_ctx* ctx, int16_t* d_plane, int stride, int width, int height, int pad );

/* layout of MotionVector (CommonLib/MCTF.h:72-82); x,y in 1/16 pel */
typedef struct { int32_t x, y, error, rmsme; double overlap; } vvhip_mv;

/* One level of MCTF::motionEstimationLuma (MCTF.cpp:1329-1397) = estimateLumaLn (:1166-1327) for every
 * block of the level, including the above/left wavefront dependency, in one call.
 * d_prev (may be NULL) is the coarser level's field, prev_w x prev_h, `factor` as in the reference.
 * d_mvs is mvs_w x mvs_h and must have been initialised with vvhip_mctf_init_mvs (default MotionVector).
 * search_pattern / low_res_filter: MCTF::m_searchPttrn / m_lowResFltSearch (MCTF.cpp:598-599).    */
VVHIP_API int vvhip_mctf_init_mvs( vvhip_ctx* ctx, vvhip_mv* d_mvs, int count );
VVHIP_API int vvhip_mctf_me_level( vvhip_ctx* ctx,
                                   const int16_t* d_org, int org_stride, const int16_t* d_buf, int buf_stride,
                                   int width, int height, int block_size,
                                   const vvhip_mv* d_prev, int prev_w, int prev_h, int factor, int double_res,
                                   int search_pattern, int low_res_filter, int bit_depth, int unit_size,
                                   vvhip_mv* d_mvs, int mvs_w, int mvs_h );

/* MCTF::motionEstimationMCTF (MCTF.cpp:666-707) for one current picture against n_refs reference
 * pictures: builds the 2x/4x(/8x) pyramids, runs every level for all references concurrently.
 * Planes are luma, `pad` (>= 128 = MCTF_PADDING) samples of replicated margin, pointers at sample (0,0).
 * d_mvs_out[r] receives ceil(w/unit) x ceil(h/unit) motion vectors of reference r.                 */
VVHIP_API int vvhip_mctf_motion_estimation( vvhip_ctx* ctx, const int16_t* d_cur, const int16_t* const* d_refs_host_array,
                                            int n_refs, int stride, int width, int height, int pad, int bit_depth,
                                            int unit_size, int mctf_speed, int add_level,
                                            vvhip_mv* const* d_mvs_out_host_array );

/* The same call WITHOUT the host wait at its end: every launch is queued on the context's stream and the call returns (like every other entry point); results are valid
 * once the stream has passed them.  What bench.py's sixth stream issues at the GOP's cadence while the other five run a picture's lists.  (A field whose anti-diagonals
 * exceed a workgroup — more than 320 blocks on the shorter side: beyond 8K — takes the row hand-off with its abort flag; the call then waits like the synchronous one.)
 * Replaces the same reference loop as above, MCTF.cpp:666-724 called per reference from MCTF::filter :779-800.                                              */
VVHIP_API int vvhip_mctf_motion_estimation_async( vvhip_ctx* ctx, const int16_t* d_cur, const int16_t* const* d_refs_host_array,
                                                  int n_refs, int stride, int width, int height, int pad, int bit_depth,
                                                  int unit_size, int mctf_speed, int add_level,
                                                  vvhip_mv* const* d_mvs_out_host_array );

/* Scored-candidate counters of the MCTF search (measurement: the algorithmic bytes of SURVEY 8d are priced per SCORED candidate, and which candidates estimateLumaLn
 * scores depends on the data — MCTF.cpp:1189-1306).  set_stats( 1 ) allocates and zeroes 24 counters and switches the counting kernel instances on, set_stats( 0 )
 * switches them off.  get_stats copies them to the host (waits for the stream).  Three phases x 8 counters:
 *   out[0..7]   phase A (estimateLumaLn up to the above/left tests): [0] integer candidates scored one by one, [1] their bytes (4 w h each), [2] fractional candidates scored
 *               one by one, [3] their bytes ((w + taps - 1)(h + taps - 1) 2 + 2 w h each; taps = 4 with the low-resolution search filter, 6 otherwise), [4] positions of the
 *               dense integer grids (MCTF.cpp:1216-1228) scored out of one staged window, [5] those windows' bytes in SURVEY 8d's window form ((w + 2R)^2 2 + 2 w h per block
 *               + 8 per position); the per-candidate figure of the grid positions is [4] x 4 w h with w = h = 32 (only full blocks take that path), [6] positions of the
 *               refinement rings (:1229-1288) whose horizontal passes are shared, [7] those rings' bytes in window form (( w + 4 )( h + 4 ) 2 + 2 w h + 8 per position: the
 *               positions of a ring lie within half a sample of its centre); per-candidate figure [6] x ((w + 3)(h + 3) 2 + 2 w h)
 *   out[8..15]  the above/left candidates scored in parallel at the neighbours' phase-A vectors, out[16..23] those scored while the recurrence is resolved.             */
VVHIP_API int vvhip_mctf_set_stats( vvhip_ctx* ctx, int on );
/* Per-class device time of the LAST motion-estimation call of the context (measurement; HIP events on the context's stream around every launch while on):
 * ms5 = { candidate scoring (meSearchKernel, all levels), neighbour scoring, sweep, final normalisation, everything else (pyramids, field initialisation) }.   */
VVHIP_API int vvhip_mctf_set_timing( vvhip_ctx* ctx, int on );
VVHIP_API int vvhip_mctf_last_times( vvhip_ctx* ctx, float* ms5 );
VVHIP_API int vvhip_mctf_get_stats( vvhip_ctx* ctx, uint64_t* out24 );

/* ======================================================================================================================
 * SURVEY 8f rank 2 — MCTF apply side: motion-compensated bilateral temporal filter of one component plane
 * (MCTF::bilateralFilter / xFinalizeBlkLine, CommonLib/MCTF.cpp:1399-1552, with applyFrac8Core_6Tap/_4Tap :259-358,
 * applyPlanarCorrectionCore :372-421 and applyBlockCore :423-518 fused per block).  Equals the reference's scalar row AND its x86 row
 * (CommonLib/x86/MCTFX86.h:861-1440): the two agree sample for sample (tests/test_oracle_vs_reference.py holds both to tolerance 0).
 *   d_org / d_refs[i] : sample (0,0) of planes whose margins cover the motion vectors (MCTF_PADDING 128 luma, 64 chroma)
 *   d_mvs[i]          : final-level motion field of reference i (what vvhip_mctf_motion_estimation returns), mv_w blocks per row
 *   chroma_shift      : 0 luma, 1 the chroma planes of 4:2:0 (vectors and block size are halved)
 *   ref_strengths     : host array, m_refStrengths[row][|POC offset| - 1] per reference (MCTF.cpp:112-117)
 *   weight_scaling / sigma_sq : per channel, see vvhip_mctf_filter_params
 * d_refs / d_mvs are HOST arrays of device pointers.                                                                              */
VVHIP_API int vvhip_mctf_apply_plane( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, int width, int height, int chroma_shift,
                                      int bit_depth, int unit_size, int low_res_flt_apply, int qp, int num_refs,
                                      const int16_t* const* d_refs, int ref_stride, const vvhip_mv* const* d_mvs, int mv_w,
                                      const double* ref_strengths, double weight_scaling, double sigma_sq,
                                      int16_t* d_out, int out_stride );
/* sigmaSq and weightScaling exactly as MCTF::bilateralFilter (:1491-1501) and xFinalizeBlkLine (:1417) derive them (host only). */
VVHIP_API int vvhip_mctf_filter_params( int qp, int bit_depth, double overall_strength, int is_chroma, double* sigma_sq, double* weight_scaling );

/* ======================================================================================================================
 * SURVEY 8f rank 3 — DMVR refinement search (decoder-normative, integer): DMVR::xProcessDMVR, CommonLib/InterPrediction.cpp:1262-1392.
 * Per sub-block (dx, dy in {8, 16}; DMVR_SUBCU_SIZE 16): bilinear prediction of both lists around the (already clipped) merge vectors
 * (InterpolationFilter::filterN2_2D), centre cost with early exit, 25-point mirrored SAD search (dmvrSadX5 semantics, strict < in the
 * reference's scan order), parametric sub-pel error surface.
 *   ref0_off / ref1_off : the sub-block's integer position for the merge vector of list 0 / 1 (mv >> 4), frac* = mv & 15
 *   result              : mvd = cu.mvdL0SubPu[num] in 1/16 sample (list 1 moves by -mvd), min_cost = the value the BDOF switch compares
 *                         with 2*dx*dy (:1386).  The final motion compensation stays with the caller (vvhip_interp_luma_batch).          */
typedef struct { int32_t ref0_off, ref1_off; int16_t frac0_x, frac0_y, frac1_x, frac1_y; } vvhip_dmvr_item;
typedef struct { int16_t mvd_x, mvd_y; int32_t pad; uint64_t min_cost; } vvhip_dmvr_result;
VVHIP_API int vvhip_dmvr_refine_batch( vvhip_ctx* ctx, const int16_t* d_ref0, int stride0, const int16_t* d_ref1, int stride1,
                                       const vvhip_dmvr_item* d_items, int n, int dx, int dy, int bit_depth, vvhip_dmvr_result* d_out );

/* ======================================================================================================================
 * SURVEY 8f rank 4 — ALF encoder statistics (P_ALF, 19.5 % of single-thread time at preset faster).
 *   vvhip_alf_classify    <- AdaptiveLoopFilter::m_deriveClassificationBlk (deriveClassificationBlk, CommonLib/AdaptiveLoopFilter.cpp:524-728):
 *       class index / transpose index of every 4x4 luma block of a picture; d_cls holds 2 bytes per block {classIdx, transposeIdx},
 *       width/4 blocks per row.  vb_ctu_height / vb_pos = m_alfVBLumaCTUHeight / m_alfVBLumaPos (:411-415: CTU height, CTU height - 4).
 *   vvhip_alf_stats_plane <- EncAdaptiveLoopFilter::getPreBlkStats + m_getPreBlkStatsAccum for every CTU of a plane
 *       (EncoderLib/EncAdaptiveLoopFilter.cpp:3376-3541, :3266-3319), linear filters (numBins 1): filter_length 7 with d_cls (luma, 25
 *       classes) or 5 with d_cls == NULL (chroma, one class; ctu_size / vb_* in chroma samples).  d_out: [numCtus][numClasses][183] floats per
 *       record = E[13][13] (row-major, symmetric), y[13], pixAcc.  The float additions happen in the reference's order (4x4 blocks of a
 *       CTU in raster order, per class): results are bit-identical.  Blocks classified {255, 255} are skipped (m_ALF_UNUSED_CLASSIDX).
 *       d_init (optional, same layout as d_out; may alias it): the records the additions start from — a statistics unit that spans several CTUs
 *       (alfUnitSize > CTU size: getStatisticsASU, :1568-1590) continues its float chains from CTU to CTU.
 * d_rec carries a replicated border of >= 4 samples (the reference reads its extended m_tempBuf); width / height multiples of 4.      */
#define VVHIP_ALF_REC 183
VVHIP_API int vvhip_alf_classify( vvhip_ctx* ctx, const int16_t* d_rec, int stride, int width, int height, int bit_depth, int vb_ctu_height, int vb_pos, uint8_t* d_cls );
VVHIP_API int vvhip_alf_stats_plane( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_rec, int rec_stride, int width, int height, int ctu_size,
                                     int filter_length, const uint8_t* d_cls /* NULL: chroma */, int vb_ctu_height, int vb_pos, const float* d_init /* may be NULL */, float* d_out );

/* The same with statistics units larger than a CTU (alfUnitSize > CTU size, EncAdaptiveLoopFilter::getStatisticsASU :1568-1590): one record set per unit of
 * unit_size x unit_size samples (<= 128), the float chains run through the unit's CTUs (ctu_size, raster order) and the blocks inside each CTU.          */
VVHIP_API int vvhip_alf_stats_plane_units( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_rec, int rec_stride, int width, int height, int unit_size,
                                           int ctu_size, int filter_length, const uint8_t* d_cls, int vb_ctu_height, int vb_pos, const float* d_init, float* d_out );

/* CC-ALF statistics <- EncAdaptiveLoopFilter::getBlkStatsCcAlf per chroma CTU (EncoderLib/EncAdaptiveLoopFilter.cpp:6061-6357; local terms
 * calcCovariance4CcAlf :6359-6422): 7 luma differences around the co-located luma sample against org - ALF-filtered chroma (d_slf_c).  One record of
 * VVHIP_ALF_REC floats per chroma CTU: E[0..6][0..6] (row pitch 13), y[0..6], pixAcc; float additions in the reference's order (bit-identical).
 * d_rec_luma carries a replicated border of >= 2 samples; shift_x / shift_y = chroma subsampling (4:2:0: 1, 1); vb_* and pic_height in luma samples
 * (the last CTU row has no virtual boundary, :6079-6082); d_init as in vvhip_alf_stats_plane.                                                  */
VVHIP_API int vvhip_ccalf_stats_plane( vvhip_ctx* ctx, const int16_t* d_org_c, int org_stride, const int16_t* d_slf_c, int slf_stride, const int16_t* d_rec_luma, int rec_stride,
                                       int width_c, int height_c, int ctu_size_c, int shift_x, int shift_y, int vb_ctu_height, int vb_pos, int pic_height,
                                       const float* d_init /* may be NULL */, float* d_out );

/* ALF filtering of a plane <- AdaptiveLoopFilter::m_filter7x7Blk / m_filter5x5Blk (filterBlk<ALF_FILTER_7 / ALF_FILTER_5>, CommonLib/AdaptiveLoopFilter.cpp:730-967) applied to
 * every enabled CTU the way EncAdaptiveLoopFilter::reconstructCTU does when no slice / tile / virtual picture boundary crosses the CTU (EncoderLib/EncAdaptiveLoopFilter.cpp:2035-2066).
 * filter_length 7 with d_cls (luma: class / transpose index per 4x4 block as written by vvhip_alf_classify, 25 classes) or 5 with d_cls == NULL (chroma, one class).
 * d_coeff / d_clip: [num_sets][numClasses][13] int16 — the reference's m_coeffApsLuma / m_fixedFilterSetCoeffDec / m_chromaCoeffFinal rows and the matching clipping values;
 * d_clip == NULL selects the linear table entries (m_filter*Blk[0]: the x86 row ignores the clipping values there).  d_ctu_set[ctu]: filter set (luma: alfCtuFilterIndex;
 * chroma: m_ctuAlternative) of the CTU, < 0 = m_ctuEnableFlag off (the CTU's samples in d_dst stay untouched).  d_src carries a replicated border of >= 4 samples and must not
 * overlap d_dst; any strides (in samples) and alignment.  vb_ctu_height / vb_pos as in vvhip_alf_classify (chroma: m_alfVBChmaCTUHeight / m_alfVBChmaPos).                       */
VVHIP_API int vvhip_alf_filter_plane( vvhip_ctx* ctx, const int16_t* d_src, ptrdiff_t src_stride, int16_t* d_dst, ptrdiff_t dst_stride, int width, int height, int ctu_size, int bit_depth,
                                      int filter_length, const uint8_t* d_cls, const int16_t* d_coeff, const int16_t* d_clip /* NULL: linear */, const int16_t* d_ctu_set,
                                      int vb_ctu_height, int vb_pos );

/* CC-ALF filtering of a chroma plane <- AdaptiveLoopFilter::m_filterCcAlf (filterBlkCcAlf, CommonLib/AdaptiveLoopFilter.cpp:969-1058) as EncAdaptiveLoopFilter::applyCcAlfFilterCTU
 * drives it (EncoderLib/EncAdaptiveLoopFilter.cpp:6606-6699): d_dst_c (the ALF-filtered chroma plane) is corrected in place from the unfiltered luma d_rec_luma (replicated border
 * >= 2).  d_coeff: [num_filters][8] int16 (ccAlfCoeff, 7 used); d_ctu_filter[ctu]: 0 = off, k = filter k-1 (m_ccAlfFilterControl).  vb_* in luma samples.                        */
VVHIP_API int vvhip_ccalf_filter_plane( vvhip_ctx* ctx, int16_t* d_dst_c, ptrdiff_t dst_stride, const int16_t* d_rec_luma, ptrdiff_t rec_stride, int width_c, int height_c, int ctu_size_c,
                                        int shift_x, int shift_y, int bit_depth, const int16_t* d_coeff, const uint8_t* d_ctu_filter, int vb_ctu_height, int vb_pos );

#ifdef __cplusplus
}
#endif
#endif /* VVENC_HIP_H */
