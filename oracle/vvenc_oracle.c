/*
 * vvenc_oracle.c — CPU restatement of the VVenC block-level RDO hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP kernels in
 * vvenc_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load it; the product path (libvvenc_hip.so and the Python/C++ host layers) never links,
 * imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-exactly
 *   (a) against the reference itself, compiled from /root/reference by oracle/ref/Makefile into
 *       oracle/_ref/libvvenc_ref.so (scalar row AND x86-SIMD row of the reference's dispatch
 *       tables) — tests/test_oracle_vs_reference.py, runs where /root/reference exists;
 *   (b) against golden vectors written from that library by tests/gen_golden.py and committed
 *       under tests/golden/ — tests/test_oracle_golden.py, runs everywhere.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference/source/Lib/CommonLib/).  It is a restatement, not a copy: generic loops replace
 * the reference's unrolled butterflies, the Hadamard tiles share one Walsh–Hadamard routine,
 * transform matrices are rebuilt from the standard's coefficient lists, and the MCTF search is
 * written as a per-block function.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include "vvenc_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Distortion                                                                      (RdCost.cpp)
 * ---------------------------------------------------------------------------------------------- */

/* RdCost.cpp:301-336 (xGetSAD) and the width-specialised twins :338-644 — rows 0,s,2s,.. with
 * s = 1<<subShift, result << subShift.  DISTORTION_PRECISION_ADJUSTMENT == 0 (TypeDef.h:171). */
uint64_t orc_sad(const int16_t *org, int orgStride, const int16_t *cur, int curStride,
                 int w, int h, int subShift)
{
    const int step = 1 << subShift;
    uint64_t sum = 0;
    for (int y = 0; y < h; y += step)
        for (int x = 0; x < w; x++)
            sum += (uint64_t)abs((int)org[y * orgStride + x] - (int)cur[y * curStride + x]);
    return sum << subShift;
}

/* RdCost.cpp:651-1000 (xGetSSE*): 32-bit difference, 64-bit accumulation. */
uint64_t orc_sse(const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h)
{
    uint64_t sum = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int32_t d = (int32_t)org[y * orgStride + x] - (int32_t)cur[y * curStride + x];
            sum += (uint64_t)((int64_t)d * d);
        }
    return sum;
}

/* In-place unnormalised Walsh–Hadamard transform of n (power of two) values spaced `stride` apart.
 * The SATD is a sum of |coefficients| so coefficient ORDER and SIGN are irrelevant; the DC term is
 * the plain sum and lands at index 0 for this butterfly. */
static void wht(int32_t *v, int n, int stride)
{
    for (int len = 1; len < n; len <<= 1)
        for (int i = 0; i < n; i += len << 1)
            for (int j = i; j < i + len; j++) {
                const int32_t a = v[j * stride], b = v[(j + len) * stride];
                v[j * stride] = a + b;
                v[(j + len) * stride] = a - b;
            }
}

/* Sum of |H·D·Hᵀ| over a tw×th tile with |DC| replaced by |DC|>>2 — the common tail of every
 * xCalcHADs* (RdCost.cpp:1020, :1119-1120, :1218-1219, :1317-1318, :1465-1466, ...). */
static int64_t had_tile_sum(int32_t *d, int tw, int th)
{
    for (int y = 0; y < th; y++) wht(d + y * tw, tw, 1);
    for (int x = 0; x < tw; x++) wht(d + x, th, tw);
    int64_t s = 0;
    for (int i = 0; i < tw * th; i++) s += llabs((long long)d[i]);
    const int64_t dc = llabs((long long)d[0]);
    return s - dc + (dc >> 2);
}

static int64_t had_tile(const int16_t *org, int os, const int16_t *cur, int cs, int tw, int th)
{
    int32_t d[16 * 16];
    for (int y = 0; y < th; y++)
        for (int x = 0; x < tw; x++)
            d[y * tw + x] = (int32_t)org[y * os + x] - (int32_t)cur[y * cs + x];
    return had_tile_sum(d, tw, th);
}

/* RdCost.cpp:1126-1223 (xCalcHADs16x16_fast): 2x2 rounded averages of org and cur separately,
 * 8x8 Hadamard of their difference, ((sad+2)>>2)<<2. */
static int64_t had_tile_16x16_fast(const int16_t *org, int os, const int16_t *cur, int cs)
{
    int32_t d[64];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            const int16_t *o = org + 2 * y * os + 2 * x, *c = cur + 2 * y * cs + 2 * x;
            const int32_t ao = (o[0] + o[1] + o[os] + o[os + 1] + 2) >> 2;
            const int32_t ac = (c[0] + c[1] + c[cs] + c[cs + 1] + 2) >> 2;
            d[y * 8 + x] = ao - ac;
        }
    const int64_t s = had_tile_sum(d, 8, 8);
    return ((s + 2) >> 2) << 2;
}

/* RdCost.cpp:1818-1938 (xGetHADs<fastHad>): tile-selection ladder + per-tile normalisation.
 * Returns UINT64_MAX for sizes the reference THROWs on. */
uint64_t orc_had(const int16_t *org, int os, const int16_t *cur, int cs, int w, int h, int fast)
{
    int tw, th, mode; /* mode: 0 = (s+?)>>k integer norm, 1 = double sqrt norm, 2 = 16x16 fast */
    if (w > h && (h & 7) == 0 && (w & 15) == 0)      { tw = 16; th = 8;  mode = 1; }
    else if (w < h && (w & 7) == 0 && (h & 15) == 0) { tw = 8;  th = 16; mode = 1; }
    else if (w > h && (h & 3) == 0 && (w & 7) == 0)  { tw = 8;  th = 4;  mode = 1; }
    else if (w < h && (w & 3) == 0 && (h & 7) == 0)  { tw = 4;  th = 8;  mode = 1; }
    else if (fast && (h % 32 == 0) && (w % 32 == 0) && h == w) { tw = 16; th = 16; mode = 2; }
    else if ((h % 8 == 0) && (w % 8 == 0))           { tw = 8;  th = 8;  mode = 0; }
    else if ((h % 4 == 0) && (w % 4 == 0))           { tw = 4;  th = 4;  mode = 0; }
    else if ((h % 2 == 0) && (w % 2 == 0))           { tw = 2;  th = 2;  mode = 0; }
    else return UINT64_MAX;

    uint64_t sum = 0;
    for (int y = 0; y < h; y += th)
        for (int x = 0; x < w; x += tw) {
            const int16_t *o = org + y * os + x, *c = cur + y * cs + x;
            if (mode == 2) { sum += (uint64_t)had_tile_16x16_fast(o, os, c, cs); continue; }
            const int64_t s = had_tile(o, os, c, cs, tw, th);
            if (mode == 1) {
                /* RdCost.cpp:1467,1606,1682,1763: sad = (int)(sad / sqrt(tw*th) * 2), `int sad` */
                const int si = (int)s;
                sum += (uint64_t)(int)(si / sqrt((double)tw * th) * 2);
            } else if (tw == 8) sum += (uint64_t)((s + 2) >> 2);   /* :1317-1319 */
            else if (tw == 4)   sum += (uint64_t)((s + 1) >> 1);   /* :1119-1121 */
            else                sum += (uint64_t)s;                /* :1020-1023 */
        }
    return sum;
}

/* RdCost.cpp:1768-1816 (xGetHAD2SADs): min(HAD, 2*SAD); compact buffers (stride == width). */
uint64_t orc_had_2sad(const int16_t *org, const int16_t *cur, int w, int h)
{
    const uint64_t hadv = orc_had(org, w, cur, w, w, h, 0);
    const uint64_t sadv = orc_sad(org, w, cur, w, w, h, 0);
    return hadv < 2 * sadv ? hadv : 2 * sadv;
}

/* RdCost.cpp:1984-2034 (xGetSAD8X5 / xGetSAD16X5): org+k / cur-k, k=0..4, each SAD >> 1. */
void orc_sad_x5(const int16_t *org, int os, const int16_t *cur, int cs, int w, int h, int subShift,
                uint64_t cost[5], int calcCentre)
{
    for (int k = 0; k < 5; k++) {
        if (k == 2 && !calcCentre) continue;
        cost[k] = orc_sad(org + k, os, cur - k, cs, w, h, subShift) >> 1;
    }
}

/* RdCost.cpp:2062-2093 (xGetSADwMask). */
uint64_t orc_sad_mask(const int16_t *org, int os, const int16_t *cur, int cs, const int16_t *mask,
                      int maskStride, int stepX, int maskStride2, int w, int h, int subShift)
{
    const int step = 1 << subShift;
    uint64_t sum = 0;
    for (int y = 0; y < h; y += step) {
        for (int x = 0; x < w; x++) {
            sum += (uint64_t)(abs((int)org[x] - (int)cur[x]) * (int)*mask);
            mask += stepX;
        }
        org += os * step;
        cur += cs * step;
        mask += maskStride * step;
        mask += maskStride2;
    }
    return sum << subShift;
}

/* RdCost.cpp:1948-1982 (fixWeightedSSE_Core), even widths. */
uint64_t orc_fix_weighted_sse(const int16_t *org, int os, const int16_t *cur, int cs, int w, int h,
                              uint32_t weight)
{
    uint64_t sum = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int32_t d = (int32_t)org[y * os + x] - (int32_t)cur[y * cs + x];
            sum += (uint64_t)(int32_t)(((int64_t)weight * (d * d) + (1 << 15)) >> 16);
        }
    return sum;
}

/* Candidate-list batch (CPU-baseline timing helper; same loop shape as InterSearch::xTZSearchHelp,
 * EncoderLib/InterSearch.cpp:410-438).  func: 0 SSE, 1 SAD, 2 HAD, 3 HAD_fast. */
void orc_dist_batch(int func, const int16_t *org, int os, const int16_t *cur, int cs, int w, int h, int subShift,
                    const int32_t *items /* n x {org_off, cur_off} */, int n, uint64_t *out)
{
    for (int i = 0; i < n; i++) {
        const int16_t *o = org + items[2 * i], *c = cur + items[2 * i + 1];
        out[i] = func == 0 ? orc_sse(o, os, c, cs, w, h) : func == 1 ? orc_sad(o, os, c, cs, w, h, subShift)
               : orc_had(o, os, c, cs, w, h, func == 3);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Transform matrices                                             (RomTr.cpp:364-449, Rom.h:164-179)
 * The VVC integer kernels are fully determined by one coefficient per distinct angle:
 *   DCT-2: T[k][n] = ±c2[m], m = (2n+1)k folded into [0,64]   (64-point; N-point = rows k*64/N, cols < N)
 *   DST-7: T[k][n] = ±s7_N[j], j = (2k+1)(n+1) folded into [0,N]
 *   DCT-8: T[k][n] = (-1)^k DST-7[k][N-1-n]
 * The coefficient lists are the standard's (H.266 §8.7.4.2 / RomTr.cpp macro bodies); the
 * construction is checked entry-by-entry against the reference's g_trCore* tables in tests/.
 * ---------------------------------------------------------------------------------------------- */
static const int16_t kDct2Angle[65] = {
    64, 91, 90, 90, 90, 90, 90, 90, 89, 88, 88, 87, 87, 86, 85, 84, 83, 83, 82, 81, 80, 79, 78, 77, 75, 73,
    73, 71, 70, 69, 67, 65, 64, 62, 61, 59, 57, 56, 54, 52, 50, 48, 46, 44, 43, 41, 38, 37, 36, 33, 31, 28,
    25, 24, 22, 20, 18, 15, 13, 11, 9,  7,  4,  2,  0 };
static const int16_t kDst7Row0_4[4]   = { 29, 55, 74, 84 };
static const int16_t kDst7Row0_8[8]   = { 17, 32, 46, 60, 71, 78, 85, 86 };
static const int16_t kDst7Row0_16[16] = { 8, 17, 25, 33, 40, 48, 55, 62, 68, 73, 77, 81, 85, 87, 88, 88 };
static const int16_t kDst7Row0_32[32] = { 4,  9,  13, 17, 21, 26, 30, 34, 38, 42, 46, 50, 53, 56, 60, 63,
                                          66, 68, 72, 74, 77, 78, 80, 82, 84, 85, 86, 87, 88, 89, 90, 90 };

static int16_t dct2_entry(int N, int k, int n)
{
    int m = ((2 * n + 1) * (k * (64 / N))) % 256, s = 1;
    if (m > 128) m = 256 - m;
    if (m > 64) { m = 128 - m; s = -1; }
    return (int16_t)(s * kDct2Angle[m]);
}
static int16_t dst7_entry(int N, int k, int n)
{
    const int16_t *row0 = N == 4 ? kDst7Row0_4 : N == 8 ? kDst7Row0_8 : N == 16 ? kDst7Row0_16 : kDst7Row0_32;
    int j = ((2 * k + 1) * (n + 1)) % (4 * N + 2), s = 1;
    if (j > 2 * N + 1) { j -= 2 * N + 1; s = -1; }
    if (j > N) j = 2 * N + 1 - j;
    return j == 0 ? 0 : (int16_t)(s * row0[j - 1]);
}

int orc_tr_matrix(int trType, int log2N, int16_t *out)
{
    const int N = 1 << log2N;
    if (trType == ORC_DCT2) { if (log2N < 1 || log2N > 6) return -1; }
    else if (trType == ORC_DCT8 || trType == ORC_DST7) { if (log2N < 2 || log2N > 5) return -1; }
    else return -1;
    for (int k = 0; k < N; k++)
        for (int n = 0; n < N; n++) {
            int16_t v;
            if (trType == ORC_DCT2) v = dct2_entry(N, k, n);
            else if (trType == ORC_DST7) v = dst7_entry(N, k, n);
            else v = (int16_t)(((k & 1) ? -1 : 1) * dst7_entry(N, k, N - 1 - n));
            out[k * N + n] = v;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * 1-D transforms                                                              (TrQuant_EMT.cpp)
 * ---------------------------------------------------------------------------------------------- */

/* Forward: TrQuant_EMT.cpp:366-420 (_fastForwardMM) + :1973-2000 (fastFwdCore); the N=2/4 butterflies
 * (:197-229, :265-300, :1106-1141, :1507-1542) are the same sums.  32-bit wrap-around arithmetic. */
int orc_fwd_1d(int trType, int log2N, const int32_t *src, int32_t *dst, int shift, int line,
               int skipLine, int skipLine2)
{
    int16_t T[64 * 64];
    if (orc_tr_matrix(trType, log2N, T)) return -1;
    const int N = 1 << log2N, reducedLine = line - skipLine, cutoff = N - skipLine2;
    const uint32_t rnd = shift > 0 ? (1u << (shift - 1)) : 0;
    for (int j = 0; j < N; j++)
        for (int i = 0; i < line; i++) {
            if (i >= reducedLine || j >= cutoff) { dst[j * line + i] = 0; continue; }
            uint32_t sum = 0;
            for (int k = 0; k < N; k++) sum += (uint32_t)src[i * N + k] * (uint32_t)(int32_t)T[j * N + k];
            dst[j * line + i] = (int32_t)(sum + rnd) >> shift;
        }
    return 0;
}

/* Inverse: TrQuant_EMT.cpp:152-194 (_fastInverseMM) + :1953-1970 (fastInvCore_) + :1941-1950 (clipCore);
 * DCT-2 butterflies N=2,4,8 (:231-257, :310-362, :492-552) are the same sums. */
int orc_inv_1d(int trType, int log2N, const int32_t *src, int32_t *dst, int shift, int line,
               int skipLine, int skipLine2, int32_t clipMin, int32_t clipMax)
{
    int16_t T[64 * 64];
    if (orc_tr_matrix(trType, log2N, T)) return -1;
    const int N = 1 << log2N, reducedLine = line - skipLine, cutoff = N - skipLine2;
    const uint32_t rnd = 1u << (shift - 1);
    for (int i = 0; i < line; i++)
        for (int j = 0; j < N; j++) {
            if (i >= reducedLine) { dst[i * N + j] = 0; continue; }
            uint32_t sum = 0;
            for (int k = 0; k < cutoff; k++) sum += (uint32_t)src[k * line + i] * (uint32_t)(int32_t)T[k * N + j];
            int32_t v = (int32_t)(sum + rnd) >> shift;
            dst[i * N + j] = v < clipMin ? clipMin : (v > clipMax ? clipMax : v);
        }
    return 0;
}

/* The g_tCoeffOps table slots themselves (TrQuant_EMT.h:63-91), with the caller's matrix pointer: scalar cores
 * TrQuant_EMT.cpp:1917-2000.  32-bit wrap-around arithmetic as the reference's int sums / SIMD mullo. */
void orc_fast_fwd_core(int trSize, const int16_t *tc, const int32_t *src, int32_t *dst, unsigned line, unsigned reducedLine,
                       unsigned cutoff, int shift)
{   /* :1973-2000 */
    const uint32_t rnd = 1u << (shift - 1);
    for (unsigned i = 0; i < reducedLine; i++)
        for (unsigned j = 0; j < cutoff; j++) {
            uint32_t sum = 0;
            for (int k = 0; k < trSize; k++) sum += (uint32_t)src[i * trSize + k] * (uint32_t)(int32_t)tc[j * trSize + k];
            dst[j * line + i] = (int32_t)(sum + rnd) >> shift;
        }
}

void orc_fast_inv_core(int trSize, const int16_t *it, const int32_t *src, int32_t *dst, unsigned lines, unsigned reducedLines,
                       unsigned rows)
{   /* :1953-1970: accumulates into dst (the caller zeroed it, :159) */
    for (unsigned i = 0; i < reducedLines; i++)
        for (int j = 0; j < trSize; j++) {
            uint32_t sum = (uint32_t)dst[i * trSize + j];
            for (unsigned k = 0; k < rows; k++) sum += (uint32_t)src[k * lines + i] * (uint32_t)(int32_t)it[k * trSize + j];
            dst[i * trSize + j] = (int32_t)sum;
        }
}

void orc_round_clip(int32_t *dst, unsigned w, unsigned h, unsigned stride, int32_t mn, int32_t mx, int32_t round, int32_t shift)
{   /* clipCore :1941-1950 */
    for (unsigned y = 0; y < h; y++)
        for (unsigned x = 0; x < w; x++) {
            const int32_t v = (int32_t)((uint32_t)dst[y * stride + x] + (uint32_t)round) >> shift;
            dst[y * stride + x] = v < mn ? mn : (v > mx ? mx : v);
        }
}

void orc_cpy_resi(const int32_t *src, int16_t *dst, ptrdiff_t stride, unsigned w, unsigned h)
{   /* cpyResiCore :1929-1938 */
    for (unsigned y = 0; y < h; y++)
        for (unsigned x = 0; x < w; x++) dst[y * stride + x] = (int16_t)src[y * w + x];
}

void orc_cpy_coeff(const int16_t *src, ptrdiff_t stride, int32_t *dst, unsigned w, unsigned h)
{   /* cpyCoeffCore :1917-1926 */
    for (unsigned y = 0; y < h; y++)
        for (unsigned x = 0; x < w; x++) dst[y * w + x] = src[y * stride + x];
}

static int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) l++; return l; }

static void tr_skips(int w, int h, int trHor, int trVer, int *skipW, int *skipH)
{
    /* TrQuant.cpp:496-497 / :587-588 (no LFNST) */
    *skipW = (trHor != ORC_DCT2 && w == 32) ? 16 : (w > 32 ? w - 32 : 0);
    *skipH = (trVer != ORC_DCT2 && h == 32) ? 16 : (h > 32 ? h - 32 : 0);
}

/* TrQuant::xT, TrQuant.cpp:481-564 (2-D case, maxLog2TrDynamicRange = 15, no LFNST). */
int orc_xT(const int16_t *resi, int resiStride, int32_t *coef, int w, int h, int trHor, int trVer, int bitDepth)
{
    if (w < 2 || h < 2 || w > 64 || h > 64) return -2;
    static _Thread_local int32_t block[64 * 64], tmp[64 * 64];
    int skipW, skipH;
    tr_skips(w, h, trHor, trVer, &skipW, &skipH);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) block[y * w + x] = resi[y * resiStride + x];
    const int shift1 = ilog2(w) + bitDepth + 6 - 15;   /* :544 */
    const int shift2 = ilog2(h) + 6;                   /* :545 */
    if (shift1 < 0) return -1;
    if (orc_fwd_1d(trHor, ilog2(w), block, tmp, shift1, h, 0, skipW)) return -1;        /* :548 */
    if (orc_fwd_1d(trVer, ilog2(h), tmp, coef, shift2, w, skipW, skipH)) return -1;     /* :549 */
    return 0;
}

/* TrQuant::xIT, TrQuant.cpp:567-655 (2-D case). */
int orc_xIT(const int32_t *coef, int16_t *resi, int resiStride, int w, int h, int trHor, int trVer, int bitDepth)
{
    if (w < 2 || h < 2 || w > 64 || h > 64) return -2;
    static _Thread_local int32_t block[64 * 64], tmp[64 * 64];
    int skipW, skipH;
    tr_skips(w, h, trHor, trVer, &skipW, &skipH);
    const int32_t cmin = -(1 << 15), cmax = (1 << 15) - 1;
    const int shift1 = 6 + 1;                /* :608 */
    const int shift2 = (6 + 15 - 1) - bitDepth; /* :609 */
    if (orc_inv_1d(trVer, ilog2(h), coef, tmp, shift1, w, skipW, skipH, cmin, cmax)) return -1;   /* :612 */
    if (orc_inv_1d(trHor, ilog2(w), tmp, block, shift2, h, 0, skipW, cmin, cmax)) return -1;      /* :613 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) resi[y * resiStride + x] = (int16_t)block[y * w + x];
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Scalar quantisation                                                              (Quant.cpp)
 * ---------------------------------------------------------------------------------------------- */
const int orc_quant_scales[2][6]     = { { 26214, 23302, 20560, 18396, 16384, 14564 }, { 18396, 16384, 14564, 13107, 11651, 10280 } }; /* Rom.cpp:1390-1394 */
const int orc_inv_quant_scales[2][6] = { { 40, 45, 51, 57, 64, 72 }, { 57, 64, 72, 80, 90, 102 } };                                  /* Rom.cpp:1396-1400 */

/* Coefficient-group geometry: Rom.cpp:1138-1148 (g_log2SbbSize[log2w][log2h] = {log2CGw, log2CGh}). */
void orc_cg_size(int log2w, int log2h, int *log2CGw, int *log2CGh)
{
    if (log2w >= 2 && log2h >= 2) { *log2CGw = 2; *log2CGh = 2; return; }
    if (log2w == 0) { *log2CGw = 0; *log2CGh = log2h < 4 ? log2h : 4; return; }
    if (log2h == 0) { *log2CGh = 0; *log2CGw = log2w < 4 ? log2w : 4; return; }
    /* one dimension == 2 samples */
    if (log2w == 1) { if (log2h <= 2) { *log2CGw = 1; *log2CGh = 1; } else { *log2CGw = 1; *log2CGh = 3; } return; }
    /* log2h == 1 */
    if (log2w <= 2) { *log2CGw = 1; *log2CGh = 1; } else { *log2CGw = 3; *log2CGh = 1; }
}

/* Up-right diagonal scan of a bw×bh array (ScanGenerator, Rom.cpp:1098-1136): each anti-diagonal is
 * walked from its bottom-left end to its top-right end. */
static void diag_scan(int bw, int bh, int *xs, int *ys)
{
    int n = 0;
    for (int d = 0; d < bw + bh - 1; d++) {
        int y = d < bh ? d : bh - 1;
        int x = d - y;
        while (y >= 0 && x < bw) { xs[n] = x; ys[n] = y; n++; x++; y--; }
    }
}

/* getScanOrder(SCAN_GROUPED_4x4, log2w, log2h) (Rom.cpp:1203-1284 generator, baked table at Rom.cpp).
 * Positions beyond the 32x32 zero-out region are filled with w*h-1 like the reference. Returns w*h. */
int orc_scan_order(int log2w, int log2h, uint32_t *out)
{
    const int w = 1 << log2w, h = 1 << log2h, total = w * h;
    int lcw, lch;
    orc_cg_size(log2w, log2h, &lcw, &lch);
    const int gw = 1 << lcw, gh = 1 << lch;
    const int wInG = (w < 32 ? w : 32) >> lcw, hInG = (h < 32 ? h : 32) >> lch;
    if (w > 32 || h > 32) for (int i = 0; i < total; i++) out[i] = (uint32_t)(total - 1);
    int gx[64], gy[64], px[16], py[16];
    diag_scan(wInG, hInG, gx, gy);
    diag_scan(gw, gh, px, py);
    const int gsize = gw * gh;
    for (int g = 0; g < wInG * hInG; g++)
        for (int p = 0; p < gsize; p++)
            out[g * gsize + p] = (uint32_t)((gy[g] * gh + py[p]) * w + gx[g] * gw + px[p]);
    return total;
}

/* Quant::quant parameter derivation, Quant.cpp:769-775 (no transform skip, no scaling lists). */
void orc_quant_params(int w, int h, int bitDepth, int qp, int isIRAP, int *quantCoeff, int *iQBits, int64_t *iAdd)
{
    const int l = ilog2(w) + ilog2(h);
    const int sqrt2 = l & 1;                              /* TU::needsSqrt2Scale, UnitTools.cpp:3616-3621 */
    const int trShift = 15 - bitDepth - (l >> 1) + (sqrt2 ? -1 : 0);   /* Quant.h:69-72 */
    *quantCoeff = orc_quant_scales[sqrt2][qp % 6];
    *iQBits = 14 + qp / 6 + trShift;
    *iAdd = (int64_t)(isIRAP ? 171 : 85) << (*iQBits - 9);
}

/* Quant::dequant parameter derivation, Quant.cpp:554-561, :601-607. */
void orc_dequant_params(int w, int h, int bitDepth, int qp, int *scale, int *rightShift, int *inputMaximum)
{
    const int l = ilog2(w) + ilog2(h);
    const int sqrt2 = l & 1;
    const int trShift = 15 - bitDepth - (l >> 1) + (sqrt2 ? -1 : 0);
    *scale = orc_inv_quant_scales[sqrt2][qp % 6];
    *rightShift = 6 - (trShift + qp / 6);
    int tgt = 32 + *rightShift - 7;            /* (sizeof(Intermediate_Int)*8 + rightShift) - (IQUANT_SHIFT+1) */
    if (tgt > 16) tgt = 16;                    /* min(maxLog2TrDynamicRange + 1, ..) */
    *inputMaximum = (1 << (tgt - 1)) - 1;
}

/* Quant::xNeedRDOQ parameter derivation, Quant.cpp:852-874 (no DepQuant). */
void orc_need_rdoq_params(int w, int h, int bitDepth, int qp, int isLuma, int *quantCoeff, int *iQBits, int64_t *iAdd, int *numCoeff)
{
    int64_t dummy;
    orc_quant_params(w, h, bitDepth, qp, 0, quantCoeff, iQBits, &dummy);
    *iAdd = (int64_t)(isLuma ? 171 : 256) << (*iQBits - 9);
    *numCoeff = w * (h < 32 ? h : 32);
}

/* QuantCore, Quant.cpp:132-230.  qcoef is w*h compact (stride w).  lfnstIdx > 0 (a TU whose coefficients went through the low-frequency non-separable transform): only the
 * FIRST coefficient group is looked at (iCGNum = 1, :152-153), and only its first 8 scan positions for 4x4 and 8x8 TUs (:156-159). */
void orc_quant_core_lfnst(const int32_t *coef, int16_t *qcoef, int32_t *deltaU, int w, int h, int quantCoeff,
                          int iQBits, int64_t iAdd, int thrVal, int lfnstIdx, int32_t *absSumOut, int *lastScanPosOut);
void orc_quant_core(const int32_t *coef, int16_t *qcoef, int32_t *deltaU, int w, int h, int quantCoeff,
                    int iQBits, int64_t iAdd, int thrVal, int32_t *absSumOut, int *lastScanPosOut)
{
    orc_quant_core_lfnst(coef, qcoef, deltaU, w, h, quantCoeff, iQBits, iAdd, thrVal, 0, absSumOut, lastScanPosOut);
}
void orc_quant_core_lfnst(const int32_t *coef, int16_t *qcoef, int32_t *deltaU, int w, int h, int quantCoeff,
                          int iQBits, int64_t iAdd, int thrVal, int lfnstIdx, int32_t *absSumOut, int *lastScanPosOut)
{
    const int log2w = ilog2(w), log2h = ilog2(h);
    uint32_t *scan = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)w * h);
    orc_scan_order(log2w, log2h, scan);
    int lcw, lch;
    orc_cg_size(log2w, log2h, &lcw, &lch);
    const int log2CG = lcw + lch, cgSize = 1 << log2CG;
    const int cgNum = lfnstIdx > 0 ? 1 : ((w < 32 ? w : 32) * (h < 32 ? h : 32)) >> log2CG;                 /* :152 */
    int scanPos = (cgNum << log2CG) - 1;
    if (lfnstIdx > 0 && ((w == 4 && h == 4) || (w == 8 && h == 8))) scanPos = 7;                               /* :156-159 */
    for (; scanPos > 0; scanPos--)
        if (coef[scan[scanPos]]) break;                                     /* :162-167 */

    int32_t thres = iQBits ? (int32_t)((int64_t)thrVal << (iQBits - 1)) : (int32_t)((int64_t)(thrVal >> 1) << iQBits);
    const int32_t useThres = thres / (quantCoeff << 2);                      /* :173-180 */
    const int is4x4 = log2CG == 4 && lcw == 2;
    for (int subSet = scanPos >> log2CG; subSet >= 1; subSet--) {            /* :184-208 */
        if (is4x4 && scanPos >= 16) {
            const int inCG = scanPos & (cgSize - 1);
            int allSmaller = 1;
            for (int k = inCG, p = scanPos; allSmaller && k >= 0; k--, p--)
                allSmaller &= abs(coef[scan[p]]) <= useThres;
            if (allSmaller) { scanPos -= inCG + 1; continue; }
            break;
        }
    }
    memset(qcoef, 0, sizeof(int16_t) * (size_t)w * h);
    int32_t absSum = 0;
    for (int p = 0; p <= scanPos; p++) {                                     /* :213-227 */
        const uint32_t bp = scan[p];
        const int32_t lvl = coef[bp];
        const int64_t t = (int64_t)abs(lvl) * quantCoeff;
        const int32_t q = (int32_t)((t + iAdd) >> iQBits);
        deltaU[bp] = (int32_t)((t - ((int64_t)q << iQBits)) >> (iQBits - 8));
        absSum += q;
        int32_t v = lvl < 0 ? -q : q;
        v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
        qcoef[bp] = (int16_t)v;
    }
    *absSumOut = absSum;
    *lastScanPosOut = scanPos;
    free(scan);
}

/* DeQuantCore, Quant.cpp:232-262. */
void orc_dequant_core(int maxX, int maxY, int scale, const int16_t *q, size_t qStride, int32_t *coef,
                      int rightShift, int inputMaximum, int32_t transformMaximum)
{
    const int32_t inMin = -(inputMaximum + 1), trMin = -(transformMaximum + 1);
    for (int y = 0, n = 0; y <= maxY; y++)
        for (int x = 0; x <= maxX; x++, n++) {
            int32_t c = q[x + y * qStride];
            c = c < inMin ? inMin : (c > inputMaximum ? inputMaximum : c);
            int32_t v;
            if (rightShift > 0) v = (int32_t)((uint32_t)(c * scale) + (1u << (rightShift - 1))) >> rightShift;
            else                v = (int32_t)((uint32_t)(c * scale) * (1u << -rightShift));
            coef[n] = v < trMin ? trMin : (v > transformMaximum ? transformMaximum : v);
        }
}

/* needRdoqCore, Quant.cpp:264-278. */
int orc_need_rdoq(const int32_t *coef, size_t num, int quantCoeff, int64_t offset, int shift)
{
    for (size_t i = 0; i < num; i++) {
        const int64_t t = (int64_t)llabs((long long)coef[i]) * quantCoeff;
        if ((int32_t)((t + offset) >> shift) != 0) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * MCTF block matching                                                               (MCTF.cpp)
 * ---------------------------------------------------------------------------------------------- */
const int16_t orc_mctf_filter6[16][8] = { /* MCTF.cpp:72-90, taps [1..6] used */
    { 0, 0, 0, 64, 0, 0, 0, 0 },    { 0, 1, -3, 64, 4, -2, 0, 0 },   { 0, 1, -6, 62, 9, -3, 1, 0 },   { 0, 2, -8, 60, 14, -5, 1, 0 },
    { 0, 2, -9, 57, 19, -7, 2, 0 }, { 0, 3, -10, 53, 24, -8, 2, 0 }, { 0, 3, -11, 50, 29, -9, 2, 0 }, { 0, 3, -11, 44, 35, -10, 3, 0 },
    { 0, 1, -7, 38, 38, -7, 1, 0 }, { 0, 3, -10, 35, 44, -11, 3, 0 }, { 0, 2, -9, 29, 50, -11, 3, 0 }, { 0, 2, -8, 24, 53, -10, 3, 0 },
    { 0, 2, -7, 19, 57, -9, 2, 0 }, { 0, 1, -5, 14, 60, -8, 2, 0 },  { 0, 1, -3, 9, 62, -6, 1, 0 },   { 0, 0, -2, 4, 64, -3, 1, 0 } };
const int16_t orc_mctf_filter4[16][4] = { /* MCTF.cpp:92-110 */
    { 0, 64, 0, 0 },   { -2, 62, 4, 0 },   { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 },
    { -6, 46, 28, -4 }, { -4, 42, 30, -4 }, { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 },
    { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, { 0, 4, 62, -2 } };

/* motionErrorLumaInt, MCTF.cpp:122-145.  The early exit (`> besterror` → return partial) is honoured
 * only in the sense the callers rely on: any return value > besterror is equivalent. We return the
 * FULL sum (≥ any partial sum), like the HIP kernels do. */
int orc_mctf_err_int(const int16_t *org, ptrdiff_t os, const int16_t *buf, ptrdiff_t bs, int w, int h)
{
    int32_t e = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int d = org[y * os + x] - buf[y * bs + x];
            e += d * d;
        }
    return e;
}

static inline int clip_pel(int v, int maxv) { return v < 0 ? 0 : (v > maxv ? maxv : v); }

/* motionErrorLumaFrac6 / Frac4, MCTF.cpp:147-257: horizontal pass → clip → vertical pass → clip → SSE.
 * tap4: taps [0..3] at offsets -1..+2; else taps [1..6] at offsets -2..+3.  Full sum (see above). */
int orc_mctf_err_frac(int tap4, const int16_t *org, ptrdiff_t os, const int16_t *buf, ptrdiff_t bs,
                      int w, int h, int fx, int fy, int bitDepth)
{
    const int maxv = (1 << bitDepth) - 1;
    const int nt = tap4 ? 4 : 6, off = tap4 ? 1 : 2;
    const int16_t *xf = tap4 ? orc_mctf_filter4[fx] : orc_mctf_filter6[fx] + 1;
    const int16_t *yf = tap4 ? orc_mctf_filter4[fy] : orc_mctf_filter6[fy] + 1;
    int16_t tmp[(64 + 8) * 64];
    const int rows = h + nt - 1;
    for (int r = 0; r < rows; r++) {
        const int16_t *srow = buf + (ptrdiff_t)(r - off) * bs;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < nt; t++) s += xf[t] * srow[x - off + t];
            tmp[r * 64 + x] = (int16_t)clip_pel((s + 32) >> 6, maxv);
        }
    }
    int32_t e = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < nt; t++) s += yf[t] * tmp[(y + t) * 64 + x];
            s = clip_pel((s + 32) >> 6, maxv);
            const int d = s - org[y * os + x];
            e += d * d;
        }
    return e;
}

/* calcVarCore, MCTF.cpp:520-546. */
double orc_mctf_calc_var(const int16_t *org, ptrdiff_t stride, int w, int h)
{
    int avg = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) avg += org[x + y * stride];
    avg <<= 4;
    avg = avg / (w * h);
    int64_t var = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int pix = org[x + y * stride] << 4;
            var += (pix - avg) * (pix - avg);
        }
    return var / 256.0;
}

/* A padded luma plane: buf points at sample (0,0); `pad` replicated samples on every side
 * (PelStorage::create(..., margin = MCTF_PADDING) + extendBorderPel, MCTF.cpp:608-612, :1076, :1096). */
typedef struct { int16_t *base; int16_t *buf; int w, h, stride, pad; } plane_t;

static void plane_alloc(plane_t *p, int w, int h, int pad)
{
    p->w = w; p->h = h; p->pad = pad; p->stride = w + 2 * pad;
    p->base = (int16_t *)malloc(sizeof(int16_t) * (size_t)p->stride * (h + 2 * pad));
    p->buf = p->base + (size_t)pad * p->stride + pad;
}
static void plane_free(plane_t *p) { free(p->base); p->base = 0; }

void orc_extend_border(int16_t *buf, int stride, int w, int h, int pad)
{
    for (int y = 0; y < h; y++) {
        int16_t *r = buf + (ptrdiff_t)y * stride;
        for (int x = 1; x <= pad; x++) { r[-x] = r[0]; r[w - 1 + x] = r[w - 1]; }
    }
    for (int y = 1; y <= pad; y++) {
        memcpy(buf - (ptrdiff_t)y * stride - pad, buf - pad, sizeof(int16_t) * (size_t)(w + 2 * pad));
        memcpy(buf + (ptrdiff_t)(h - 1 + y) * stride - pad, buf + (ptrdiff_t)(h - 1) * stride - pad, sizeof(int16_t) * (size_t)(w + 2 * pad));
    }
}

/* MCTF::subsampleLuma, MCTF.cpp:1072-1097 (visible area; caller extends the border). */
void orc_mctf_subsample(const int16_t *src, int srcStride, int w, int h, int16_t *dst, int dstStride)
{
    const int nw = w / 2, nh = h / 2;
    for (int y = 0; y < nh; y++)
        for (int x = 0; x < nw; x++) {
            const int16_t *a = src + (ptrdiff_t)(2 * y) * srcStride + 2 * x;
            dst[(ptrdiff_t)y * dstStride + x] = (int16_t)((a[0] + a[srcStride] + a[1] + a[srcStride + 1] + 2) >> 2);
        }
}

/* Calls of motionErrorLuma since the last reset and their algorithmic bytes: { integer-vector calls, 4 w h each, fractional-vector calls,
 * (w + taps - 1)(h + taps - 1) 2 + 2 w h each }.  The REFERENCE's schedule (every call estimateLumaLn makes, MCTF.cpp:1189-1306): the upper bound the device library's
 * own scored-candidate counters (vvhip_mctf_get_stats) are checked against in tests/ — the device skips candidates that cannot win.  Not thread-safe (tests only). */
static uint64_t g_mctf_calls[4];
void orc_mctf_count_reset(void) { g_mctf_calls[0] = g_mctf_calls[1] = g_mctf_calls[2] = g_mctf_calls[3] = 0; }
void orc_mctf_count_get(uint64_t *out4) { for (int i = 0; i < 4; i++) out4[i] = g_mctf_calls[i]; }

/* MCTF::motionErrorLuma, MCTF.cpp:1099-1164. */
static int me_error(const plane_t *org, const plane_t *ref, int x, int y, int dx, int dy, int bs,
                    int lowRes, int bitDepth)
{
    const int fx = dx & 15, fy = dy & 15;
    int w = bs < org->w - x ? bs : org->w - x; w &= ~7;
    int h = bs < org->h - y ? bs : org->h - y; h &= ~7;
    const int16_t *o = org->buf + x + (ptrdiff_t)y * org->stride;
    if ((fx | fy) == 0) { g_mctf_calls[0]++; g_mctf_calls[1] += (uint64_t)(4 * w * h); }
    else { const int t = lowRes ? 3 : 5; g_mctf_calls[2]++; g_mctf_calls[3] += (uint64_t)((w + t) * (h + t) * 2 + 2 * w * h); }
    if ((fx | fy) == 0) {
        dx /= 16; dy /= 16;          /* C division: both are exact multiples here */
        return orc_mctf_err_int(o, org->stride, ref->buf + x + dx + (ptrdiff_t)(y + dy) * ref->stride, ref->stride, w, h);
    }
    dx >>= 4; dy >>= 4;
    return orc_mctf_err_frac(lowRes, o, org->stride, ref->buf + x + dx + (ptrdiff_t)(y + dy) * ref->stride, ref->stride,
                             w, h, fx, fy, bitDepth);
}

typedef struct { int w, h; orc_mv_t *v; } mvfield_t;

#define TRY(DX, DY) do { const int e_ = me_error(org, ref, bx, by, (DX), (DY), bs, lowRes, bitDepth); \
                         if (e_ < best.error) { best.x = (DX); best.y = (DY); best.error = e_; } } while (0)

/* MCTF::estimateLumaLn, MCTF.cpp:1166-1327, for one block; `mvs` holds the final MVs of the blocks
 * above and to the left (raster order is a valid topological order of that dependency). */
static orc_mv_t me_block(const plane_t *org, const plane_t *ref, int bx, int by, int bs, const mvfield_t *prev,
                         int factor, int doubleRes, const mvfield_t *mvs, int searchPttrn, int lowRes,
                         int bitDepth, int unitSize)
{
    orc_mv_t best = { 0, 0, INT_MAX, 65535, 0.0 };
    int range = doubleRes ? 0 : (searchPttrn == 2 ? 3 : 5);
    if (!prev) range = 8;
    else {
        for (int py = -1; py <= 1; py++) {
            const int ty = by / (2 * bs) + py;
            if (ty < 0 || ty >= prev->h) continue;
            for (int px = -1; px <= 1; px++) {
                const int tx = bx / (2 * bs) + px;
                if (tx < 0 || tx >= prev->w) continue;
                const orc_mv_t *old = &prev->v[ty * prev->w + tx];
                TRY(old->x * factor, old->y * factor);
            }
        }
        TRY(0, 0);
    }
    orc_mv_t pb = best;
    const int d = (!prev && searchPttrn == 2) ? 2 : 1;
    for (int y2 = pb.y / 16 - range; y2 <= pb.y / 16 + range; y2 += d)
        for (int x2 = pb.x / 16 - range; x2 <= pb.x / 16 + range; x2 += d)
            TRY(x2 * 16, y2 * 16);
    if (doubleRes) {
        pb = best;
        int dr = searchPttrn ? 6 : 12;
        const int d1 = searchPttrn == 2 ? 6 : 4;
        for (int y2 = -dr; y2 <= dr; y2 += d1)
            for (int x2 = -dr; x2 <= dr; x2 += d1)
                if (x2 || y2) TRY(pb.x + x2, pb.y + y2);
        pb = best;
        for (int y2 = -2; y2 <= 2; y2 += 2)
            for (int x2 = -2; x2 <= 2; x2 += 2)
                if (x2 || y2) TRY(pb.x + x2, pb.y + y2);
        pb = best;
        for (int y2 = -1; y2 <= 1; y2++)
            for (int x2 = -1; x2 <= 1; x2++)
                if (x2 || y2) TRY(pb.x + x2, pb.y + y2);
    }
    if (by > 0) { const orc_mv_t *a = &mvs->v[((by - bs) / bs) * mvs->w + bx / bs]; TRY(a->x, a->y); }
    if (bx > 0) { const orc_mv_t *l = &mvs->v[(by / bs) * mvs->w + (bx - bs) / bs]; TRY(l->x, l->y); }
    if (doubleRes) {                                                   /* MCTF.cpp:1308-1321 */
        int w = bs < org->w - bx ? bs : org->w - bx; w &= ~7;
        int h = bs < org->h - by ? bs : org->h - by; h &= ~7;
        const double bdScale = (double)(1 << (2 * (10 - bitDepth)));
        const double dvar = orc_mctf_calc_var(org->buf + bx + (ptrdiff_t)by * org->stride, org->stride, w, h) * bdScale;
        const double mse = best.error * bdScale / (double)(w * h);
        best.error = (int)(20 * ((best.error * bdScale + 5.0) / (dvar + 5.0)) + mse / 50.0);
        best.rmsme = (int32_t)(uint16_t)(0.5 + sqrt(mse));
        best.overlap = ((double)w * h) / (unitSize * unitSize);
    }
    return best;
}

/* MCTF::motionEstimationLuma (single-thread branch), MCTF.cpp:1388-1396 + loop bounds of :1174. */
static void me_level(mvfield_t *mvs, const plane_t *org, const plane_t *ref, int bs, const mvfield_t *prev, int factor,
                     int doubleRes, int searchPttrn, int lowRes, int bitDepth, int unitSize)
{
    for (int by = 0; by + 8 <= org->h; by += bs)
        for (int bx = 0; bx + 8 <= org->w; bx += bs)
            mvs->v[(by / bs) * mvs->w + bx / bs] =
                me_block(org, ref, bx, by, bs, prev, factor, doubleRes, mvs, searchPttrn, lowRes, bitDepth, unitSize);
}

static void field_alloc(mvfield_t *f, int w, int h)
{
    f->w = w; f->h = h; f->v = (orc_mv_t *)malloc(sizeof(orc_mv_t) * (size_t)w * h);
    for (int i = 0; i < w * h; i++) { f->v[i].x = 0; f->v[i].y = 0; f->v[i].error = INT_MAX; f->v[i].rmsme = 65535; f->v[i].overlap = 0.0; }
}

/* MCTF::motionEstimationMCTF, MCTF.cpp:666-707, for one (current, reference) pair of compact luma planes.
 * levelOut[0..4] = 1/8 (only if addLevel), 1/4, 1/2, 1/1 @2*unit, final @unit; may be NULL.  levelDims[2k..2k+1] = w,h. */
int orc_mctf_me(const int16_t *orgLuma, const int16_t *refLuma, int width, int height, int bitDepth, int unitSize,
                int mctfSpeed, int addLevel, orc_mv_t **levelOut, int *levelDims)
{
    const int pad = 128;                                                /* MCTF_PADDING, CommonDef.h:520 */
    const int lowRes = mctfSpeed > 0;                                   /* MCTF.cpp:598 */
    const int pttrn = mctfSpeed > 0 ? (mctfSpeed >= 3 ? 2 : 1) : 0;     /* MCTF.cpp:599 */
    plane_t o[4], r[4];
    const int16_t *src[2] = { orgLuma, refLuma };
    plane_t *pl[2] = { o, r };
    for (int s = 0; s < 2; s++) {
        plane_alloc(&pl[s][0], width, height, pad);
        for (int y = 0; y < height; y++) memcpy(pl[s][0].buf + (ptrdiff_t)y * pl[s][0].stride, src[s] + (size_t)y * width, sizeof(int16_t) * width);
        orc_extend_border(pl[s][0].buf, pl[s][0].stride, width, height, pad);
        for (int l = 1; l < 4; l++) {
            plane_alloc(&pl[s][l], pl[s][l - 1].w / 2, pl[s][l - 1].h / 2, pad);
            orc_mctf_subsample(pl[s][l - 1].buf, pl[s][l - 1].stride, pl[s][l - 1].w, pl[s][l - 1].h, pl[s][l].buf, pl[s][l].stride);
            orc_extend_border(pl[s][l].buf, pl[s][l].stride, pl[s][l].w, pl[s][l].h, pad);
        }
    }
    const int u = unitSize;
    mvfield_t mvm, mv0, mv1, mv2, mvs;
    field_alloc(&mvm, width / (u * 16) + 1, height / (u * 16) + 1);
    field_alloc(&mv0, width / (u * 8) + 1, height / (u * 8) + 1);
    field_alloc(&mv1, width / (u * 4) + 1, height / (u * 4) + 1);
    field_alloc(&mv2, width / (u * 2) + 1, height / (u * 2) + 1);
    field_alloc(&mvs, (width + u - 1) / u, (height + u - 1) / u);
    if (addLevel) {
        me_level(&mvm, &o[3], &r[3], 2 * u, NULL, 1, 0, pttrn, lowRes, bitDepth, u);
        me_level(&mv0, &o[2], &r[2], 2 * u, &mvm, 2, 0, pttrn, lowRes, bitDepth, u);
    } else {
        me_level(&mv0, &o[2], &r[2], 2 * u, NULL, 1, 0, pttrn, lowRes, bitDepth, u);
    }
    me_level(&mv1, &o[1], &r[1], 2 * u, &mv0, 2, 0, pttrn, lowRes, bitDepth, u);
    me_level(&mv2, &o[0], &r[0], 2 * u, &mv1, 2, 0, pttrn, lowRes, bitDepth, u);
    me_level(&mvs, &o[0], &r[0], u, &mv2, 1, 1, pttrn, lowRes, bitDepth, u);

    mvfield_t *f[5] = { &mvm, &mv0, &mv1, &mv2, &mvs };
    for (int k = 0; k < 5; k++) {
        const int present = k > 0 || addLevel;
        levelDims[2 * k] = present ? f[k]->w : 0;
        levelDims[2 * k + 1] = present ? f[k]->h : 0;
        if (present && levelOut && levelOut[k]) memcpy(levelOut[k], f[k]->v, sizeof(orc_mv_t) * (size_t)f[k]->w * f[k]->h);
        free(f[k]->v);
    }
    for (int l = 0; l < 4; l++) { plane_free(&o[l]); plane_free(&r[l]); }
    return 0;
}

/* ================================================================================================
 * SURVEY §8f rank 1 — sub-pel interpolation                               (InterpolationFilter.cpp)
 * ==============================================================================================*/
/* Tap tables (the VVC luma 8-tap, luma 6-tap "4x4/affine", alternative half-pel and chroma 4-tap sets, InterpolationFilter.cpp:64-142)
 * are mirror-symmetric in the phase: row P-p is row p reversed.  Only phases 0..P/2 are stored. */
static const int8_t if_luma8_half[9][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { -1, 2, -5, 62, 8, -3, 1, 0 }, { -1, 3, -8, 60, 13, -4, 1, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
    { -1, 4, -11, 40, 40, -11, 4, -1 } };
static const int8_t if_luma6_half[9][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { 0, 1, -5, 62, 8, -3, 1, 0 }, { 0, 2, -8, 60, 13, -4, 1, 0 },
    { 0, 3, -10, 58, 17, -5, 1, 0 }, { 0, 3, -11, 52, 26, -8, 2, 0 }, { 0, 2, -9, 47, 31, -10, 3, 0 }, { 0, 3, -11, 45, 34, -10, 3, 0 },
    { 0, 3, -11, 40, 40, -11, 3, 0 } };
static const int8_t if_alt_hpel[8] = { 0, 3, 9, 20, 20, 9, 3, 0 };
static const int8_t if_chroma_half[17][4] = {
    { 0, 64, 0, 0 }, { -1, 63, 2, 0 }, { -2, 62, 4, 0 }, { -2, 60, 7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 },
    { -4, 55, 15, -2 }, { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 },
    { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 } };

/* set: 0 = m_lumaFilter (8 taps), 1 = m_lumaFilter4x4 (8 entries, used as 6 taps from entry 1), 2 = m_chromaFilter (4 taps, phase 0..32),
 * 3 = m_lumaAltHpelIFilter (phase ignored), 4 = m_bilinearFilterPrec4 (2 taps).  Returns the tap count the reference filters with;
 * coeff[] receives the table row exactly as the reference passes it (8 entries for sets 0/1/3). */
int orc_if_coeff(int set, int phase, int16_t coeff[8])
{
    for (int i = 0; i < 8; i++) coeff[i] = 0;
    if (set == 0 || set == 1) {
        if (phase < 0 || phase > 16) return -1;
        const int8_t (*t)[8] = set == 0 ? if_luma8_half : if_luma6_half;
        /* the table rows are mirror images in the phase around the two centre taps 3 and 4: row[16-p][k] = row[p][7-k] */
        for (int k = 0; k < 8; k++) coeff[k] = phase <= 8 ? t[phase][k] : t[16 - phase][7 - k];
        return set == 0 ? 8 : 6;
    }
    if (set == 2) {
        if (phase < 0 || phase > 32) return -1;
        for (int k = 0; k < 4; k++) coeff[k] = phase <= 16 ? if_chroma_half[phase][k] : if_chroma_half[32 - phase][3 - k];
        return 4;
    }
    if (set == 3) { for (int k = 0; k < 8; k++) coeff[k] = if_alt_hpel[k]; return 6; }
    if (set == 4) { if (phase < 0 || phase > 15) return -1; coeff[0] = (int16_t)(16 - phase); coeff[1] = (int16_t)phase; return 2; }
    return -1;
}

/* InterpolationFilter::filter<N,isVertical,isFirst,isLast> (InterpolationFilter.cpp:356-441).  coeff = the table row (for N == 6 the
 * reference skips its first entry, :361-364).  Output is truncated to Pel like the reference's `Pel val`. */
void orc_if_filter(int N, int isVertical, int isFirst, int isLast, int bitDepth, const int16_t *src, int srcStride, int16_t *dst,
                   int dstStride, int width, int height, const int16_t *coeff)
{
    if (N == 6) coeff++;
    const int cStride = isVertical ? srcStride : 1;
    src -= (N / 2 - 1) * cStride;
    const int headRoom = 14 - bitDepth > 2 ? 14 - bitDepth : 2;
    int shift = 6, offset;
    if (N != 2) {
        if (isLast) { shift += isFirst ? 0 : headRoom; offset = 1 << (shift - 1); offset += isFirst ? 0 : (8192 << 6); }
        else        { shift -= isFirst ? headRoom : 0; offset = isFirst ? -(8192 << shift) : 0; }
    } else {
        if (isFirst) { shift = 4 - (10 - bitDepth); offset = 1 << (shift - 1); }
        else         { shift = 4; offset = 1 << (shift - 1); }
    }
    const int maxv = (1 << bitDepth) - 1;
    for (int row = 0; row < height; row++) {
        for (int col = 0; col < width; col++) {
            int sum = 0;
            for (int k = 0; k < N; k++) sum += src[col + k * cStride] * coeff[k];
            int16_t val = (int16_t)((sum + offset) >> shift);
            if (isLast) val = val < 0 ? 0 : (val > maxv ? (int16_t)maxv : val);
            dst[col] = val;
        }
        src += srcStride;
        dst += dstStride;
    }
}

/* InterpolationFilter::filterCopy<isFirst,isLast> (:255-333). */
void orc_if_copy(int isFirst, int isLast, int bitDepth, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width,
                 int height, int biMCForDMVR)
{
    const int shift = 14 - bitDepth > 2 ? 14 - bitDepth : 2, maxv = (1 << bitDepth) - 1;
    for (int row = 0; row < height; row++, src += srcStride, dst += dstStride)
        for (int col = 0; col < width; col++) {
            if (isFirst == isLast) dst[col] = src[col];
            else if (isFirst) dst[col] = biMCForDMVR ? (int16_t)(src[col] << (10 - bitDepth))
                                                      : (int16_t)((int16_t)((uint16_t)src[col] << shift) - 8192);
            else {
                /* rightShiftU( val + offset, shift ) on the int promotion: arithmetic shift (CommonDef.h:755), :319-322 */
                int16_t val = (int16_t)(((int)src[col] + (int)(int16_t)((1 << (shift - 1)) + 8192)) >> shift);
                dst[col] = val < 0 ? 0 : (val > maxv ? (int16_t)maxv : val);
            }
        }
}

/* Luma filter choice of InterpolationFilter::filterHor/filterVer (:557-601, :617-661) for nFilterIdx 0. */
static int if_luma_set(int frac, int width, int height, int useAltHpelIf, int reduceTap, int vertical)
{
    if (reduceTap == 0 || (useAltHpelIf && frac == 8)) {
        if (useAltHpelIf && frac == 8) return 3;
        if ((width == 4 && height == 4) || (!vertical && width == 4 && height == 4 + 8 - 1)) return 1;
        return 0;
    }
    return reduceTap == 1 ? 1 : 2;
}

/* InterpolationFilter::filterHor / filterVer, luma (compID Y), nFilterIdx 0. */
void orc_if_luma_1d(int vertical, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width, int height, int frac,
                    int isFirst, int isLast, int bitDepth, int useAltHpelIf, int reduceTap)
{
    if (frac == 0) {
        if (!vertical) { isFirst = 1; }
        orc_if_copy(isFirst, isLast, bitDepth, src, srcStride, dst, dstStride, width, height, 0);   /* :559-565, :619-622 */
        return;
    }
    int16_t c[8];
    const int set = if_luma_set(frac, width, height, useAltHpelIf, reduceTap, vertical);
    const int N = orc_if_coeff(set, set == 2 ? frac << 1 : frac, c);
    orc_if_filter(N, vertical, vertical ? isFirst : 1, isLast, bitDepth, src, srcStride, dst, dstStride, width, height, c);
}

/* Motion-compensated luma prediction block at a 1/16-sample vector: the dispatch of InterPredInterpolation::xPredInterBlk
 * (InterPrediction.cpp:832-865) without BDOF/DMVR; rndRes = !bi.  The fused 2-D entries (filter4x4/8xH/16xH, :682-761) compute the
 * same two passes with int intermediates (:826-935). */
void orc_if_pred_luma(const int16_t *ref, int refStride, int16_t *dst, int dstStride, int width, int height, int xFrac, int yFrac,
                      int rndRes, int bitDepth, int useAltHpelIf)
{
    if (yFrac == 0) { orc_if_luma_1d(0, ref, refStride, dst, dstStride, width, height, xFrac, 1, rndRes, bitDepth, useAltHpelIf, 0); return; }
    if (xFrac == 0) { orc_if_luma_1d(1, ref, refStride, dst, dstStride, width, height, yFrac, 1, rndRes, bitDepth, useAltHpelIf, 0); return; }
    static _Thread_local int16_t tmp[(128 + 8) * 128];
    int16_t ch[8], cv[8];
    int sh, sv;
    if (width == 4 && height == 4) sh = sv = useAltHpelIf ? 3 : 1;      /* filter4x4 :688-693: the alternative row replaces BOTH directions */
    else { sh = (useAltHpelIf && xFrac == 8) ? 3 : 0; sv = (useAltHpelIf && yFrac == 8) ? 3 : 0; }   /* filter8xH/16xH :719-720,:747-748; generic :860-864 */
    /* the fused entries run 8 taps over the full 8-entry rows; rows of the 6-tap sets start and end with 0, so 6 taps from entry 1 are the same sums */
    const int Nh = orc_if_coeff(sh, xFrac, ch), Nv = orc_if_coeff(sv, yFrac, cv);
    orc_if_filter(Nh, 0, 1, 0, bitDepth, ref - 3 * refStride, refStride, tmp, width, width, height + 7, ch);
    orc_if_filter(Nv, 1, 0, rndRes, bitDepth, tmp + 3 * width, width, dst, dstStride, width, height, cv);
}

/* ================================================================================================
 * SURVEY §8f rank 2 — MCTF apply side (motion-compensated bilateral temporal filter)   (MCTF.cpp)
 * Float corners follow the reference's SCALAR row operation by operation (float / double mix as the
 * C++ expressions promote); the reference's own unit test allows +-1 between its scalar and SIMD rows.
 * ==============================================================================================*/
/* applyFrac8Core_6Tap / _4Tap, MCTF.cpp:259-358: two passes, first pass NOT clipped (only truncated to Pel), second clipped. */
void orc_mctf_apply_frac(int tap4, const int16_t *org, ptrdiff_t os, int16_t *dst, ptrdiff_t ds, int w, int h, int fx, int fy,
                         int bitDepth)
{
    const int maxv = (1 << bitDepth) - 1;
    static _Thread_local int16_t tmp[64 + 8][64];
    if (tap4) {
        const int16_t *xf = orc_mctf_filter4[fx], *yf = orc_mctf_filter4[fy];
        for (int by = 0; by < h + 3; by++)
            for (int bx = 0; bx < w; bx++) {
                const int16_t *p = org + (by - 1) * os + bx - 1;
                int sum = xf[0] * p[0] + xf[1] * p[1] + xf[2] * p[2] + xf[3] * p[3];
                tmp[by][bx] = (int16_t)((sum + 32) >> 6);
            }
        for (int by = 0; by < h; by++)
            for (int bx = 0; bx < w; bx++) {
                int sum = yf[0] * tmp[by][bx] + yf[1] * tmp[by + 1][bx] + yf[2] * tmp[by + 2][bx] + yf[3] * tmp[by + 3][bx];
                dst[by * ds + bx] = (int16_t)clip_pel((sum + 32) >> 6, maxv);
            }
    } else {
        const int16_t *xf = orc_mctf_filter6[fx], *yf = orc_mctf_filter6[fy];
        for (int by = 1; by < h + 6; by++)
            for (int bx = 0; bx < w; bx++) {
                const int16_t *p = org + (by - 3) * os + bx - 3;
                int sum = 0;
                for (int k = 1; k <= 6; k++) sum += xf[k] * p[k];
                tmp[by][bx] = (int16_t)((sum + 32) >> 6);
            }
        for (int by = 0; by < h; by++)
            for (int bx = 0; bx < w; bx++) {
                int sum = 0;
                for (int k = 1; k <= 6; k++) sum += yf[k] * tmp[by + k][bx];
                dst[by * ds + bx] = (int16_t)clip_pel((sum + 32) >> 6, maxv);
            }
    }
}

/* applyPlanarCorrectionCore, MCTF.cpp:372-421 (w, h powers of two). */
void orc_mctf_planar_correction(const int16_t *ref, ptrdiff_t rs, int16_t *dst, ptrdiff_t ds, int w, int h, int bitDepth,
                                uint16_t motionError)
{
    static const int32_t xSzm[6] = { 0, 1, 20, 336, 5440, 87296 };
    const int32_t blockSize = w * h, log2Width = ilog2(w), maxPel = (1 << bitDepth) - 1;
    const uint32_t me2 = (uint32_t)motionError * (uint32_t)motionError;
    const int32_t mWeight = (int32_t)(me2 < 512u ? me2 : 512u);
    const int32_t xSum = (blockSize * (w - 1)) >> 1;
    int32_t x1yzm = 0, x2yzm = 0, ySum = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int32_t z = dst[y * ds + x] - ref[y * rs + x];
            x1yzm += x * z; x2yzm += y * z; ySum += z;
        }
    const int64_t denom = blockSize * xSzm[log2Width];
    int64_t numer = (int64_t)mWeight * ((int64_t)x1yzm * blockSize - xSum * ySum);
    int32_t b1 = (int32_t)((numer < 0 ? numer - (denom >> 1) : numer + (denom >> 1)) / denom);
    b1 = b1 < INT16_MIN ? INT16_MIN : (b1 > INT16_MAX ? INT16_MAX : b1);
    numer = (int64_t)mWeight * ((int64_t)x2yzm * blockSize - xSum * ySum);
    int32_t b2 = (int32_t)((numer < 0 ? numer - (denom >> 1) : numer + (denom >> 1)) / denom);
    b2 = b2 > INT16_MAX ? INT16_MAX : (b2 < INT16_MIN ? INT16_MIN : b2);
    const int32_t b0 = (mWeight * ySum - (b1 + b2) * xSum + (blockSize >> 1)) >> (log2Width << 1);
    if (b0 == 0 && b1 == 0 && b2 == 0) return;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int32_t p = (b0 + b1 * x + b2 * y + 256) >> 9;
            const int32_t z = dst[y * ds + x] - p;
            dst[y * ds + x] = (int16_t)(z < 0 ? 0 : (z > maxPel ? maxPel : z));
        }
}

static float mctf_fast_exp(float n, float d)
{   /* MCTF.cpp:359-367 */
    float x = 1.0f + n / (d * 1024);
    x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
    return x;
}

/* applyBlockCore, MCTF.cpp:423-518 — and, sample for sample, the reference's x86 row applyBlockSIMD (CommonLib/x86/MCTFX86.h:1207-1440) as well: the two differ in ONE operation,
 * the rounding of the blended sample — ( Pel )( newVal + 0.5 ) with a double literal here, _mm_add_ps( v, 0.5f ) + truncation there (:1424-1427) — and the single-precision sum
 * is exact whenever it could change the integer part (v and v + 0.5 share a binade unless the sum crosses a power of two, where the integer part is that power either way);
 * -a / b == a / -b in IEEE arithmetic covers the other textual difference.  tests/test_oracle_vs_reference.py holds both rows to tolerance 0 (the reference's unit test allows
 * them +-1, test/vvenc_unit_test/vvenc_unit_test.cpp:1280-1282). */
/* corrected[i]: compact w x h blocks. */
void orc_mctf_apply_block(const int16_t *src, ptrdiff_t ss, int16_t *dst, ptrdiff_t ds, int w, int h, int bitDepth,
                          const int16_t *const *corrected, int numRefs, const int *verror, const double *refStrengths,
                          double weightScaling, double sigmaSq)
{
    const int16_t maxv = (int16_t)((1 << bitDepth) - 1);
    int vnoise[16] = { 0 };
    float vsw[16] = { 0 }, vww[16] = { 0 };
    int minError = INT32_MAX;
    for (int i = 0; i < numRefs; i++) {
        int64_t variance = 0, diffsum = 0;
        const int16_t *ref = corrected[i];
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int diff = src[y * ss + x] - ref[y * w + x];
                variance += diff * diff;
                if (x != w - 1) { const int dR = src[y * ss + x + 1] - ref[y * w + x + 1]; diffsum += (dR - diff) * (dR - diff); }
                if (y != h - 1) { const int dD = src[(y + 1) * ss + x] - ref[(y + 1) * w + x]; diffsum += (dD - diff) * (dD - diff); }
            }
        variance *= (int64_t)1 << (2 * (10 - bitDepth));
        diffsum *= (int64_t)1 << (2 * (10 - bitDepth));
        const int cntV = w * h, cntD = 2 * cntV - w - h;
        vnoise[i] = (int)round((15.0 * cntD / cntV * variance + 5.0) / (diffsum + 5.0));
        minError = verror[i] < minError ? verror[i] : minError;
    }
    for (int i = 0; i < numRefs; i++) {
        const int error = verror[i], noise = vnoise[i];
        float ww = 1, sw = 1;
        ww *= (noise < 25) ? 1.0 : 0.6;
        sw *= (noise < 25) ? 1.0 : 0.8;
        ww *= (error < 50) ? 1.2 : ((error > 100) ? 0.6 : 1.0);
        sw *= (error < 50) ? 1.0 : 0.8;
        ww *= ((minError + 1.0) / (error + 1.0));
        vww[i] = ww * weightScaling * refStrengths[i];
        vsw[i] = sw * 2 * sigmaSq;
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int16_t orgVal = src[y * ss + x];
            float temporalWeightSum = 1.0;
            float newVal = (float)orgVal;
            for (int i = 0; i < numRefs; i++) {
                const int refVal = corrected[i][y * w + x];
                const int diff = refVal - orgVal;
                const float diffSq = diff * diff;
                float weight = vww[i] * mctf_fast_exp(-diffSq, vsw[i]);
                newVal += weight * refVal;
                temporalWeightSum += weight;
            }
            newVal /= temporalWeightSum;
            int16_t sampleVal = (int16_t)(newVal + 0.5);
            sampleVal = sampleVal < 0 ? 0 : (sampleVal > maxv ? maxv : sampleVal);
            dst[y * ds + x] = sampleVal;
        }
}

/* MCTF::bilateralFilter + xFinalizeBlkLine, MCTF.cpp:1399-1552, one component plane (csx = csy = 0 luma, 1 chroma of 4:2:0).
 * org / refs[i]: sample (0,0) of planes with enough margin for the vectors; mvs[i]: the final-level motion field of reference i
 * (mvW blocks per row); weightScaling / sigmaSq as xFinalizeBlkLine derives them for the component. */
void orc_mctf_bilateral_plane(const int16_t *org, ptrdiff_t orgStride, int width, int height, int cs, int bitDepth, int unitSize,
                              int lowResFltApply, int qp, int numRefs, const int16_t *const *refs, ptrdiff_t refStride,
                              const orc_mv_t *const *mvs, int mvW, const double *refStrengths, double weightScaling, double sigmaSq,
                              int16_t *out, ptrdiff_t outStride)
{
    const int blk = unitSize >> cs;
    static _Thread_local int16_t bufs[16][64 * 64];
    for (int by = 0, yb = 0; by < height; by += blk, yb++)
        for (int bx = 0, xb = 0; bx < width; bx += blk, xb++) {
            const int h = blk < height - by ? blk : height - by, w = blk < width - bx ? blk : width - bx;
            const int16_t *corrected[16];
            int verror[16];
            for (int i = 0; i < numRefs; i++) {
                const orc_mv_t *mv = &mvs[i][yb * mvW + xb];
                const int dx = mv->x >> cs, dy = mv->y >> cs, xInt = mv->x >> (4 + cs), yInt = mv->y >> (4 + cs);
                const int16_t *src = refs[i] + (ptrdiff_t)(by + yInt) * refStride + bx + xInt;
                orc_mctf_apply_frac(lowResFltApply, src, refStride, bufs[i], w, w, h, dx & 15, dy & 15, bitDepth);
                if (mv->rmsme > 0 && qp <= 32 && w == h && w <= 32)          /* "deblocking", :1473-1476 */
                    orc_mctf_planar_correction(org + (ptrdiff_t)by * orgStride + bx, orgStride, bufs[i], w, w, h, bitDepth, (uint16_t)mv->rmsme);
                corrected[i] = bufs[i];
                verror[i] = mv->error;
            }
            orc_mctf_apply_block(org + (ptrdiff_t)by * orgStride + bx, orgStride, out + (ptrdiff_t)by * outStride + bx, outStride, w, h,
                                 bitDepth, corrected, numRefs, verror, refStrengths, weightScaling, sigmaSq);
        }
}

/* ================================================================================================
 * SURVEY §8f rank 3 — DMVR refinement search                                 (InterPrediction.cpp)
 * ==============================================================================================*/
/* InterpolationFilter::filterN2_2D (InterpolationFilter.cpp:662-673) with scalarFilterN2_2D (:675-681): the bilinear prediction DMVR
 * searches on (10-bit internal precision, no clipping). */
void orc_if_bilinear(const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int fracX, int fracY, int bitDepth)
{
    int16_t ch[8], cv[8];
    orc_if_coeff(4, fracX, ch);
    orc_if_coeff(4, fracY, cv);
    if (fracX && fracY) {
        static _Thread_local int16_t tmp[(128 + 8) * (128 + 8)];
        orc_if_filter(2, 0, 1, 0, bitDepth, src, srcStride, tmp, w, w, h + 1, ch);
        orc_if_filter(2, 1, 0, 0, bitDepth, tmp, w, dst, dstStride, w, h, cv);
    } else if (fracX) orc_if_filter(2, 0, 1, 0, bitDepth, src, srcStride, dst, dstStride, w, h, ch);
    else if (fracY)   orc_if_filter(2, 1, 1, 0, bitDepth, src, srcStride, dst, dstStride, w, h, cv);
    else              orc_if_copy(1, 0, bitDepth, src, srcStride, dst, dstStride, w, h, 1);
}

static int32_t dmvr_div_maxq7(int64_t N, int64_t D)
{   /* div_for_maxq7, InterPrediction.cpp:1131-1165: three steps of restoring division, quotient magnitude <= 7 */
    int32_t sign = 0, q = 0;
    if (N < 0) { sign = 1; N = -N; }
    D <<= 3;
    if (N >= D) { N -= D; q++; }
    q <<= 1;
    D >>= 1;
    if (N >= D) { N -= D; q++; }
    q <<= 1;
    if (N >= (D >> 1)) q++;
    return sign ? -q : q;
}

/* xSubPelErrorSrfc (:1167-1187): parametric error surface from the centre cost and its four neighbours (left, top, right, bottom). */
void orc_dmvr_subpel_error_surface(const uint64_t sad[5], int32_t deltaMv[2])
{
    for (int hv = 0; hv < 2; hv++) {
        const int64_t numerator = (int64_t)((sad[hv + 1] - sad[hv + 3]) << 4);
        const int64_t denominator = (int64_t)(sad[hv + 1] + sad[hv + 3] - (sad[0] << 1));
        if (denominator != 0) {
            if (sad[hv + 1] != sad[0] && sad[hv + 3] != sad[0]) deltaMv[hv] = dmvr_div_maxq7(numerator, denominator);
            else deltaMv[hv] = sad[hv + 1] == sad[0] ? -8 : 8;
        }
    }
}

/* The refinement search of DMVR::xProcessDMVR for one sub-block (InterPrediction.cpp:1312-1384): l0c / l1c point at the CENTRE position of
 * the two bilinear predictions (which extend 2 samples to every side).  mvd = cu.mvdL0SubPu (1/16 sample); returns the final minCost
 * (the BDOF switch compares it with 2*dx*dy, :1386). */
uint64_t orc_dmvr_search(const int16_t *l0c, const int16_t *l1c, int stride, int dx, int dy, int16_t mvd[2])
{
    uint64_t sadArray[25];
    uint64_t minCost = orc_sad(l0c, stride, l1c, stride, dx, dy, 1) >> 1;      /* DF_SAD, subShift 1 (:1310, :1332) */
    minCost -= minCost >> 2;
    mvd[0] = mvd[1] = 0;
    if (minCost < (uint64_t)(dx * dy)) return minCost;
    int16_t delta[2] = { 0, 0 }, total[2] = { 0, 0 };
    sadArray[12] = minCost;
    for (int ver = -2; ver <= 2; ver++) {
        const ptrdiff_t offset = -2 + (ptrdiff_t)ver * stride;
        orc_sad_x5(l0c + offset, stride, l1c - offset, stride, dx, dy, 1, &sadArray[(ver + 2) * 5], ver != 0);
        for (int hor = -2; hor <= 2; hor++) {
            const uint64_t cost = sadArray[(ver + 2) * 5 + hor + 2];
            if (cost < minCost) { minCost = cost; delta[0] = (int16_t)hor; delta[1] = (int16_t)ver; }
        }
    }
    total[0] = (int16_t)(delta[0] * 16); total[1] = (int16_t)(delta[1] * 16);
    /* xDMVRSubPixelErrorSurface (:1227-1244): only when the best integer offset is not on the border of the 5x5 window */
    if (abs(total[0]) != (2 << 4) && abs(total[1]) != (2 << 4)) {
        const uint64_t *p = &sadArray[12 + delta[1] * 5 + delta[0]];
        const uint64_t sb[5] = { p[0], p[-1], p[-5], p[1], p[5] };
        int32_t t[2] = { 0, 0 };
        orc_dmvr_subpel_error_surface(sb, t);
        total[0] = (int16_t)(total[0] + t[0]); total[1] = (int16_t)(total[1] + t[1]);
    }
    mvd[0] = total[0]; mvd[1] = total[1];
    return minCost;
}

/* One DMVR sub-block end to end: bilinear prediction of both lists around the (clipped) merge vectors, then the search.
 * ref0/ref1 point at the sub-block's integer position for the merge vector (mv >> 4); frac* = mv & 15. */
uint64_t orc_dmvr_refine(const int16_t *ref0, int stride0, int fx0, int fy0, const int16_t *ref1, int stride1, int fx1, int fy1, int dx,
                         int dy, int bitDepth, int16_t mvd[2])
{
    static _Thread_local int16_t p0[(16 + 4) * (16 + 4)], p1[(16 + 4) * (16 + 4)];
    const int bs = dx + 4;
    orc_if_bilinear(ref0 - 2 * stride0 - 2, stride0, p0, bs, dx + 4, dy + 4, fx0, fy0, bitDepth);      /* mergeMV - 2 samples (:1285-1288) */
    orc_if_bilinear(ref1 - 2 * stride1 - 2, stride1, p1, bs, dx + 4, dy + 4, fx1, fy1, bitDepth);
    return orc_dmvr_search(p0 + 2 * bs + 2, p1 + 2 * bs + 2, bs, dx, dy, mvd);
}


/* ============================================================================================================================
 * ALF encoder statistics.
 *   classification   AdaptiveLoopFilter::deriveClassificationBlk   CommonLib/AdaptiveLoopFilter.cpp:524-728 (restated per 4x4 block: the
 *                    reference's laplacian rows / 4-column sums depend on absolute positions only)
 *   ELocal           EncAdaptiveLoopFilter::calcLinCovariance4      EncoderLib/EncAdaptiveLoopFilter.cpp:3707-3921 (linear filters, numBins 1)
 *   accumulation     getPreBlkStats :3376-3541 + getPreBlkStatsAccum :3266-3319 (x86: one int32 sum per 4x4 block and entry, converted to
 *                    float and ADDED in float — x86/EncAdaptiveLoopFilterX86.h:160-236; the order of the float additions is the raster
 *                    order of the 4x4 blocks inside the CTU area, per class)
 * ============================================================================================================================ */
static int alf_abs(int v) { return v < 0 ? -v : v; }

void orc_alf_classify(const int16_t *rec, ptrdiff_t stride, int width, int height, int shift, int vbCTUHeight, int vbPos, uint8_t *cls)
{
  static const int th[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
  static const int transposeTable[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };
  const int bw = width / 4;
  for (int Y = 0; Y < height; Y += 4)
    for (int X = 0; X < width; X += 4)
    {
      int lap[4][4][4];                                         /* [dir][row pair r][col pair c], 2x2-subsampled positions (Y-2+2r, X-2+2c) */
      for (int r = 0; r < 4; r++)
      {
        const int y = Y - 2 + 2 * r;                            /* row of src1 (:558) */
        const int16_t *s1 = rec + (ptrdiff_t)y * stride, *s0 = s1 - stride, *s2 = s1 + stride, *s3 = s1 + 2 * stride;
        if (y > 0 && (y & (vbCTUHeight - 1)) == vbPos - 2) s3 = s2;        /* :559-566 */
        else if (y > 0 && (y & (vbCTUHeight - 1)) == vbPos) s0 = s1;
        for (int c = 0; c < 4; c++)
        {
          const int x = X - 2 + 2 * c;
          const int16_t *pY = s1 + x, *pYdown = s0 + x, *pYup = s2 + x, *pYup2 = s3 + x;
          const int16_t y0 = (int16_t)(pY[0] << 1), yup1 = (int16_t)(pYup[1] << 1);
          lap[0][r][c] = alf_abs(y0 - pYdown[0] - pYup[0]) + alf_abs(yup1 - pY[1] - pYup2[1]);       /* VER  :583 */
          lap[1][r][c] = alf_abs(y0 - pY[1] - pY[-1]) + alf_abs(yup1 - pYup[2] - pYup[0]);           /* HOR  :584 */
          lap[2][r][c] = alf_abs(y0 - pYdown[-1] - pYup[1]) + alf_abs(yup1 - pY[0] - pYup2[2]);      /* DIAG0 :585 */
          lap[3][r][c] = alf_abs(y0 - pYup[-1] - pYdown[1]) + alf_abs(yup1 - pYup2[0] - pY[2]);      /* DIAG1 :586 */
        }
      }
      const int ym = Y % vbCTUHeight;
      const int r0 = ym == vbPos ? 1 : 0, r1 = ym == vbPos - 4 ? 3 : 4;    /* :630-650 */
      int sum[4];
      for (int d = 0; d < 4; d++)
      {
        sum[d] = 0;
        for (int r = r0; r < r1; r++) for (int c = 0; c < 4; c++) sum[d] += lap[d][r][c];
      }
      const int sumV = sum[0], sumH = sum[1], sumD0 = sum[2], sumD1 = sum[3];
      const int tempAct = sumV + sumH;
      const int yb = Y & (vbCTUHeight - 1);
      int activity = (tempAct * ((yb == vbPos - 4 || yb == vbPos) ? 96 : 64)) >> shift;                /* :655-663 */
      activity = activity < 0 ? 0 : (activity > 15 ? 15 : activity);
      int classIdx = th[activity];
      int hv1, hv0, d1, d0, hvd1, hvd0, dirTempHV, dirTempD, mainDirection, secondaryDirection;
      if (sumV > sumH) { hv1 = sumV; hv0 = sumH; dirTempHV = 1; } else { hv1 = sumH; hv0 = sumV; dirTempHV = 3; }
      if (sumD0 > sumD1) { d1 = sumD0; d0 = sumD1; dirTempD = 0; } else { d1 = sumD1; d0 = sumD0; dirTempD = 2; }
      if ((uint32_t)d1 * (uint32_t)hv0 > (uint32_t)hv1 * (uint32_t)d0) { hvd1 = d1; hvd0 = d0; mainDirection = dirTempD; secondaryDirection = dirTempHV; }
      else { hvd1 = hv1; hvd0 = hv0; mainDirection = dirTempHV; secondaryDirection = dirTempD; }
      int directionStrength = 0;
      if (hvd1 > 2 * hvd0) directionStrength = 1;
      if (hvd1 * 2 > 9 * hvd0) directionStrength = 2;
      if (directionStrength) classIdx += (((mainDirection & 1) << 1) + directionStrength) * 5;
      uint8_t *o = cls + 2 * ((size_t)(Y / 4) * bw + X / 4);
      o[0] = (uint8_t)classIdx;
      o[1] = (uint8_t)transposeTable[mainDirection * 2 + (secondaryDirection >> 1)];
    }
}

static int alf_clip_idx(int clipToBdry, int i, int clip) { return clipToBdry ? (i > clip ? i : clip) : i; }

/* ELocal of one 4x4 block (rec -> its top-left sample): [numCoeff][16] int16, sample index = row * 4 + column */
void orc_alf_elocal(const int16_t *rec, ptrdiff_t stride, int L, int transposeIdx, const int clipTopRow[4], const int clipBotRow[4], int16_t *ELocal)
{
  for (int ii = 0; ii < 4; ii++)
  {
    const int16_t *r0 = rec + ii * stride;
    const int cb = clipBotRow[ii] != 4;                       /* :3438-3455 picks the clipping instantiation */
    const int ct = clipTopRow[ii], cbr = clipBotRow[ii];
    int k = 0;
#define ALF_OFF0(i) ((ptrdiff_t)alf_clip_idx(cb, (i), ct) * stride)
#define ALF_OFF1(i) (-(ptrdiff_t)alf_clip_idx(cb, (i), -cbr) * stride)
#define ALF_PUT(P0, P1) { for (int x = 0; x < 4; x++) ELocal[k * 16 + ii * 4 + x] = (int16_t)((P0)[x] + (P1)[x] - (int16_t)(r0[x] << 1)); k++; }
    if (transposeIdx == 0)
    {
      for (int i = -L; i < 0; i++) for (int j = -L - i; j <= L + i; j++) ALF_PUT(r0 + ALF_OFF0(i) + j, r0 + ALF_OFF1(i) - j)
      for (int j = -L; j < 0; j++) ALF_PUT(r0 + j, r0 - j)
    }
    else if (transposeIdx == 1)
    {
      for (int j = -L; j < 0; j++) for (int i = -L - j; i <= L + j; i++) ALF_PUT(r0 + j + ALF_OFF0(i), r0 - j + ALF_OFF1(i))
      for (int i = -L; i < 0; i++) ALF_PUT(r0 + ALF_OFF0(i), r0 + ALF_OFF1(i))
    }
    else if (transposeIdx == 2)
    {
      for (int i = -L; i < 0; i++) for (int j = L + i; j >= -L - i; j--) ALF_PUT(r0 + ALF_OFF0(i) + j, r0 + ALF_OFF1(i) - j)
      for (int j = -L; j < 0; j++) ALF_PUT(r0 + j, r0 - j)
    }
    else
    {
      for (int j = -L; j < 0; j++) for (int i = L + j; i >= -L - j; i--) ALF_PUT(r0 + j + ALF_OFF0(i), r0 - j + ALF_OFF1(i))
      for (int i = -L; i < 0; i++) ALF_PUT(r0 + ALF_OFF0(i), r0 + ALF_OFF1(i))
    }
    for (int x = 0; x < 4; x++) ELocal[k * 16 + ii * 4 + x] = r0[x];
#undef ALF_OFF0
#undef ALF_OFF1
#undef ALF_PUT
  }
}

void orc_alf_stats_area(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int x0, int y0, int w, int h, int filterLength,
                        const uint8_t *cls, int clsStride, int vbCTUHeight, int vbPos, float *out)
{
  const int L = filterLength >> 1, nc = filterLength * filterLength / 4 + 1;
  for (int i = 0; i < h; i += 4)
  {
    int clipTopRow[4] = { -4, -4, -4, -4 }, clipBotRow[4] = { 4, 4, 4, 4 };
    for (int ii = 0; ii < 4; ii++)
    {
      const int vbDistance = ((y0 + i + ii) % vbCTUHeight) - vbPos;       /* :3399-3411 */
      if (vbDistance >= -3 && vbDistance < 0) { clipBotRow[ii] = -vbDistance - 1; clipTopRow[ii] = -clipBotRow[ii]; }
      else if (vbDistance >= 0 && vbDistance < 3) { clipTopRow[ii] = -vbDistance; clipBotRow[ii] = -clipTopRow[ii]; }
    }
    for (int j = 0; j < w; j += 4)
    {
      int classIdx = 0, transposeIdx = 0;
      if (cls)
      {
        const uint8_t *c = cls + 2 * ((size_t)((y0 + i) / 4) * clsStride + (x0 + j) / 4);
        if (c[0] == 255 && c[1] == 255) continue;                       /* m_ALF_UNUSED_CLASSIDX / _TRANSPOSIDX :3416 */
        classIdx = c[0]; transposeIdx = c[1];
      }
      const int16_t *o = org + (ptrdiff_t)(y0 + i) * orgStride + x0 + j, *r = rec + (ptrdiff_t)(y0 + i) * recStride + x0 + j;
      int16_t yLocal[16], ELocal[13 * 16];
      for (int ii = 0; ii < 4; ii++) for (int jj = 0; jj < 4; jj++) yLocal[ii * 4 + jj] = (int16_t)(o[jj + ii * orgStride] - r[jj + ii * recStride]);
      orc_alf_elocal(r, recStride, L, transposeIdx, clipTopRow, clipBotRow, ELocal);
      float *E = out + (size_t)classIdx * ORC_ALF_REC, *yv = E + 169, *pix = E + 182;
      for (int k = 0; k < nc; k++)
      {
        for (int l = k; l < nc; l++)
        {
          int32_t sum = 0;
          for (int p = 0; p < 16; p++) sum += (int32_t)ELocal[l * 16 + p] * ELocal[k * 16 + p];
          E[k * 13 + l] += (float)sum;
        }
        int32_t sum = 0;
        for (int p = 0; p < 16; p++) sum += (int32_t)ELocal[k * 16 + p] * yLocal[p];
        yv[k] += (float)sum;
      }
      int32_t sum = 0;
      for (int p = 0; p < 16; p++) sum += (int32_t)yLocal[p] * yLocal[p];
      *pix += (float)sum;
    }
  }
  const int numClasses = cls ? 25 : 1;                                  /* :3493-3512 mirror the upper triangle */
  for (int c = 0; c < numClasses; c++)
  {
    float *E = out + (size_t)c * ORC_ALF_REC;
    for (int k = 1; k < nc; k++) for (int l = 0; l < k; l++) E[k * 13 + l] = E[l * 13 + k];
  }
}

void orc_alf_stats_plane(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int width, int height, int ctuSize, int filterLength,
                         const uint8_t *cls, int vbCTUHeight, int vbPos, float *out)
{
  const int numClasses = cls ? 25 : 1, ctusX = (width + ctuSize - 1) / ctuSize, ctusY = (height + ctuSize - 1) / ctuSize;
  memset(out, 0, sizeof(float) * (size_t)ctusX * ctusY * numClasses * ORC_ALF_REC);
  orc_alf_stats_plane_acc(org, orgStride, rec, recStride, width, height, ctuSize, filterLength, cls, vbCTUHeight, vbPos, out);
}

/* the same walk continuing from the records already in out: what a statistics unit of several CTUs does (getStatisticsASU :1568-1590) */
void orc_alf_stats_plane_acc(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int width, int height, int ctuSize, int filterLength,
                             const uint8_t *cls, int vbCTUHeight, int vbPos, float *out)
{
  const int numClasses = cls ? 25 : 1, ctusX = (width + ctuSize - 1) / ctuSize, ctusY = (height + ctuSize - 1) / ctuSize;
  for (int cy = 0; cy < ctusY; cy++)
    for (int cx = 0; cx < ctusX; cx++)
    {
      const int x0 = cx * ctuSize, y0 = cy * ctuSize;
      const int w = x0 + ctuSize > width ? width - x0 : ctuSize, h = y0 + ctuSize > height ? height - y0 : ctuSize;
      orc_alf_stats_area(org, orgStride, rec, recStride, x0, y0, w, h, filterLength, cls, width / 4, vbCTUHeight, vbPos,
                         out + (size_t)(cy * ctusX + cx) * numClasses * ORC_ALF_REC);
    }
}


/* ---- CC-ALF statistics: getBlkStatsCcAlf (EncAdaptiveLoopFilter.cpp:6061-6357) per chroma CTU, scalar form (:6318-6345) with the x86 form's
 * float additions (one int32 sum per 4x4 block and entry).  Only coefficients 0..6 are defined (the x86 loop also touches row 7 of an
 * uninitialised buffer; those entries are not used by the derivation). */
void orc_ccalf_stats_plane(const int16_t *orgC, ptrdiff_t orgStride, const int16_t *slfC, ptrdiff_t slfStride, const int16_t *recLuma, ptrdiff_t recStride,
                           int widthC, int heightC, int ctuSizeC, int sx, int sy, int vbCTUHeight, int vbPosIn, int picHeight, float *out)
{
  const int ctusX = (widthC + ctuSizeC - 1) / ctuSizeC, ctusY = (heightC + ctuSizeC - 1) / ctuSizeC;
  const int dx1 = 1 << sx;
  for (int cy = 0; cy < ctusY; cy++)
    for (int cx = 0; cx < ctusX; cx++)
    {
      const int x0 = cx * ctuSizeC, y0 = cy * ctuSizeC;
      const int w = x0 + ctuSizeC > widthC ? widthC - x0 : ctuSizeC, h = y0 + ctuSizeC > heightC ? heightC - y0 : ctuSizeC;
      const int yPos = y0 << sy;
      const int vbPos = (yPos + (ctuSizeC << sy)) >= picHeight ? picHeight : vbPosIn;          /* :6079-6082 */
      float *E = out + (size_t)(cy * ctusX + cx) * ORC_ALF_REC, *yv = E + 169, *pix = E + 182;
      for (int i = 0; i < h; i += 4)
        for (int j = 0; j < w; j += 4)
        {
          int16_t EL[7][16], yL[16];
          for (int ii = 0; ii < 4; ii++)
          {
            const int vbd = (((i + ii) << sy) % vbCTUHeight) - vbPos;                            /* :6100-6103 (rows relative to the CTU) */
            const int16_t *r0 = recLuma + (ptrdiff_t)((y0 + i + ii) << sy) * recStride + ((x0 + j) << sx);
            const int16_t *rm1 = r0 - recStride, *rp1 = r0 + recStride, *rp2 = r0 + 2 * recStride;
            if (vbd == -2 || vbd == 1) rp2 = rp1;                                               /* :6368-6376 */
            else if (vbd == -1 || vbd == 0) { rm1 = r0; rp2 = rp1 = r0; }
            for (int jj = 0; jj < 4; jj++)
            {
              const int d = jj * dx1;
              const int16_t c = r0[d];
              EL[0][ii * 4 + jj] = (int16_t)(rm1[d] - c);
              EL[1][ii * 4 + jj] = (int16_t)(r0[d - 1] - c);
              EL[2][ii * 4 + jj] = (int16_t)(r0[d + 1] - c);
              EL[3][ii * 4 + jj] = (int16_t)(rp1[d - 1] - c);
              EL[4][ii * 4 + jj] = (int16_t)(rp1[d] - c);
              EL[5][ii * 4 + jj] = (int16_t)(rp1[d + 1] - c);
              EL[6][ii * 4 + jj] = (int16_t)(rp2[d] - c);
              yL[ii * 4 + jj] = (int16_t)(orgC[(ptrdiff_t)(y0 + i + ii) * orgStride + x0 + j + jj] - slfC[(ptrdiff_t)(y0 + i + ii) * slfStride + x0 + j + jj]);
            }
          }
          for (int k = 0; k < 7; k++)
          {
            for (int l = k; l < 7; l++)
            {
              int32_t sum = 0;
              for (int p = 0; p < 16; p++) sum += (int32_t)EL[k][p] * EL[l][p];
              E[k * 13 + l] += (float)sum;
            }
            int32_t sum = 0;
            for (int p = 0; p < 16; p++) sum += (int32_t)EL[k][p] * yL[p];
            yv[k] += (float)sum;
          }
          int32_t sum = 0;
          for (int p = 0; p < 16; p++) sum += (int32_t)yL[p] * yL[p];
          *pix += (float)sum;
        }
      for (int k = 1; k < 7; k++) for (int l = 0; l < k; l++) E[k * 13 + l] = E[l * 13 + k];    /* :6349-6355 */
    }
}


void orc_alf_stats_plane_units(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int width, int height, int unitSize, int ctuSize, int filterLength,
                               const uint8_t *cls, int vbCTUHeight, int vbPos, float *out)
{
  const int numClasses = cls ? 25 : 1, ux = (width + unitSize - 1) / unitSize, uy = (height + unitSize - 1) / unitSize;
  memset(out, 0, sizeof(float) * (size_t)ux * uy * numClasses * ORC_ALF_REC);
  for (int ay = 0; ay < uy; ay++)
    for (int ax = 0; ax < ux; ax++)
      for (int y0 = ay * unitSize; y0 < (ay + 1) * unitSize && y0 < height; y0 += ctuSize)      /* :1582-1588 */
        for (int x0 = ax * unitSize; x0 < (ax + 1) * unitSize && x0 < width; x0 += ctuSize)
        {
          const int w = x0 + ctuSize > width ? width - x0 : ctuSize, h = y0 + ctuSize > height ? height - y0 : ctuSize;
          orc_alf_stats_area(org, orgStride, rec, recStride, x0, y0, w, h, filterLength, cls, width / 4, vbCTUHeight, vbPos,
                             out + (size_t)(ay * ux + ax) * numClasses * ORC_ALF_REC);
        }
}


/* ---- ALF filtering: AdaptiveLoopFilter::filterBlk<ALF_FILTER_7 / ALF_FILTER_5> (AdaptiveLoopFilter.cpp:730-967) for every CTU of a plane the way
 * EncAdaptiveLoopFilter::reconstructCTU calls it when no slice / tile / virtual picture boundary crosses the CTU (EncAdaptiveLoopFilter.cpp:2035-2066).
 * Tap pairs as (dy, dx) of the first sample of the point-symmetric pair; rows beyond the virtual boundary fold back onto the nearest allowed row. */
static const int8_t alf_tap7[12][2] = { {3,0}, {2,1}, {2,0}, {2,-1}, {1,2}, {1,1}, {1,0}, {1,-1}, {1,-2}, {0,3}, {0,2}, {0,1} };
static const int8_t alf_tap5[6][2]  = { {2,0}, {1,1}, {1,0}, {1,-1}, {0,2}, {0,1} };
/* coefficient order per transposeIdx (:805-849) */
static const uint8_t alf_perm7[4][13] = { {0,1,2,3,4,5,6,7,8,9,10,11,12}, {9,4,10,8,1,5,11,7,3,0,2,6,12}, {0,3,2,1,8,7,6,5,4,9,10,11,12}, {9,8,10,4,3,7,11,5,1,0,2,6,12} };
static const uint8_t alf_perm5[4][7]  = { {0,1,2,3,4,5,6}, {4,1,5,3,0,2,6}, {0,3,2,1,4,5,6}, {4,3,5,1,0,2,6} };

static int alf_clamp(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }

void orc_alf_filter_plane(const int16_t *src, ptrdiff_t srcStride, int16_t *dst, ptrdiff_t dstStride, int width, int height, int ctuSize, int bitDepth, int filterLength,
                          const uint8_t *cls, const int16_t *coeffSets, const int16_t *clipSets, const int16_t *ctuSet, int vbCTUHeight, int vbPos)
{
  const int numClasses = cls ? 25 : 1, taps = filterLength == 7 ? 12 : 6, ctusX = (width + ctuSize - 1) / ctuSize, maxVal = (1 << bitDepth) - 1;
  for (int y = 0; y < height; y++)
  {
    const int yVb = y & (vbCTUHeight - 1);
    const int dist = yVb < vbPos ? vbPos - 1 - yVb : yVb - vbPos;                 /* rows the filter may reach on either side (:881-897) */
    const int nearVb = yVb == vbPos - 1 || yVb == vbPos;                           /* :898-899 */
    for (int x = 0; x < width; x++)
    {
      const int set = ctuSet[(y / ctuSize) * ctusX + x / ctuSize];
      if (set < 0) continue;                                                       /* m_ctuEnableFlag off: the CTU keeps its samples */
      int classIdx = 0, tr = 0;
      if (cls) { const uint8_t *c = cls + 2 * ((size_t)(y / 4) * (width / 4) + x / 4); classIdx = c[0]; tr = c[1]; }
      const int16_t *cf = coeffSets + ((size_t)set * numClasses + classIdx) * 13, *cl = clipSets + ((size_t)set * numClasses + classIdx) * 13;
      const int16_t *p = src + (ptrdiff_t)y * srcStride + x;
      const int cur = p[0];
      int sum = 0;
      for (int k = 0; k < taps; k++)
      {
        int dy = filterLength == 7 ? alf_tap7[k][0] : alf_tap5[k][0];
        const int dx = filterLength == 7 ? alf_tap7[k][1] : alf_tap5[k][1];
        const int src_k = filterLength == 7 ? alf_perm7[tr][k] : alf_perm5[tr][k];
        if (dy > dist) dy = dist;
        const int c = cl[src_k];
        const int a = p[dy * srcStride + dx] - cur, b = p[-dy * srcStride - dx] - cur;
        sum += cf[src_k] * (alf_clamp(-c, c, a) + alf_clamp(-c, c, b));           /* clipALF, AdaptiveLoopFilter.h:85-88 */
      }
      sum = nearVb ? (sum + 512) >> 10 : (sum + 64) >> 7;                          /* :940-947, m_NUM_BITS 8 */
      dst[(ptrdiff_t)y * dstStride + x] = (int16_t)alf_clamp(0, maxVal, sum + cur);
    }
  }
}

/* ---- CC-ALF filtering: AdaptiveLoopFilter::filterBlkCcAlf (AdaptiveLoopFilter.cpp:969-1058) over the CTUs of a chroma plane as
 * EncAdaptiveLoopFilter::applyCcAlfFilterCTU drives it (EncAdaptiveLoopFilter.cpp:6606-6699, branch without virtual picture boundaries). */
void orc_ccalf_filter_plane(int16_t *dstC, ptrdiff_t dstStride, const int16_t *recLuma, ptrdiff_t recStride, int widthC, int heightC, int ctuSizeC, int sx, int sy, int bitDepth,
                            const int16_t *coeff, const uint8_t *ctuFilter, int vbCTUHeight, int vbPos)
{
  const int ctusX = (widthC + ctuSizeC - 1) / ctuSizeC, maxVal = (1 << bitDepth) - 1, half = (1 << bitDepth) >> 1;
  for (int y = 0; y < heightC; y++)
  {
    const int pos = (y << sy) & (vbCTUHeight - 1);
    if (sy == 0 && (pos == vbPos || pos == vbPos + 1)) continue;                   /* :1016-1019 */
    ptrdiff_t o1 = recStride, o2 = -recStride, o3 = 2 * recStride;
    if (pos == vbPos - 2 || pos == vbPos + 1) o3 = o1;                             /* :1020-1029 */
    else if (pos == vbPos - 1 || pos == vbPos) o1 = o2 = o3 = 0;
    for (int x = 0; x < widthC; x++)
    {
      const int f = ctuFilter[(y / ctuSizeC) * ctusX + x / ctuSizeC];
      if (!f) continue;
      const int16_t *cf = coeff + (size_t)(f - 1) * 8;
      const int16_t *l = recLuma + (ptrdiff_t)(y << sy) * recStride + (x << sx);
      const int c = l[0];
      int sum = cf[0] * (l[o2] - c) + cf[1] * (l[-1] - c) + cf[2] * (l[1] - c) + cf[3] * (l[o1 - 1] - c) + cf[4] * (l[o1] - c) + cf[5] * (l[o1 + 1] - c) + cf[6] * (l[o3] - c);
      sum = (sum + 64) >> 7;                                                       /* m_scaleBits 7 */
      sum = alf_clamp(0, maxVal, sum + half) - half;
      int16_t *d = dstC + (ptrdiff_t)y * dstStride + x;
      *d = (int16_t)alf_clamp(0, maxVal, sum + *d);
    }
  }
}
