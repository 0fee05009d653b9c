/* vvenc_oracle.h — C interface of the CPU parity oracle (TEST INFRASTRUCTURE ONLY, see vvenc_oracle.c). */
#ifndef VVENC_ORACLE_H
#define VVENC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* transform types: numbering of the reference's TransType enum (CommonLib/TypeDef.h: DCT2=0, DCT8=1, DST7=2) */
enum { ORC_DCT2 = 0, ORC_DCT8 = 1, ORC_DST7 = 2 };

/* one MCTF motion vector, layout of MotionVector (CommonLib/MCTF.h:72-82): 1/16-pel x,y */
typedef struct { int32_t x, y, error, rmsme; double overlap; } orc_mv_t;

uint64_t orc_sad(const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int subShift);
uint64_t orc_sse(const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h);
uint64_t orc_had(const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int fast);
uint64_t orc_had_2sad(const int16_t *org, const int16_t *cur, int w, int h);
void     orc_sad_x5(const int16_t *org, int orgStride, const int16_t *cur, int curStride, int w, int h, int subShift,
                    uint64_t cost[5], int calcCentre);
uint64_t orc_sad_mask(const int16_t *org, int os, const int16_t *cur, int cs, const int16_t *mask, int maskStride,
                      int stepX, int maskStride2, int w, int h, int subShift);
uint64_t orc_fix_weighted_sse(const int16_t *org, int os, const int16_t *cur, int cs, int w, int h, uint32_t weight);

void orc_dist_batch(int func, const int16_t *org, int os, const int16_t *cur, int cs, int w, int h, int subShift,
                    const int32_t *items, int n, uint64_t *out);

int  orc_tr_matrix(int trType, int log2N, int16_t *out);
void orc_fast_fwd_core(int trSize, const int16_t *tc, const int32_t *src, int32_t *dst, unsigned line, unsigned reducedLine, unsigned cutoff, int shift);
void orc_fast_inv_core(int trSize, const int16_t *it, const int32_t *src, int32_t *dst, unsigned lines, unsigned reducedLines, unsigned rows);
void orc_round_clip(int32_t *dst, unsigned w, unsigned h, unsigned stride, int32_t mn, int32_t mx, int32_t round, int32_t shift);
void orc_cpy_resi(const int32_t *src, int16_t *dst, ptrdiff_t stride, unsigned w, unsigned h);
void orc_cpy_coeff(const int16_t *src, ptrdiff_t stride, int32_t *dst, unsigned w, unsigned h);
int  orc_fwd_1d(int trType, int log2N, const int32_t *src, int32_t *dst, int shift, int line, int skipLine, int skipLine2);
int  orc_inv_1d(int trType, int log2N, const int32_t *src, int32_t *dst, int shift, int line, int skipLine, int skipLine2,
                int32_t clipMin, int32_t clipMax);
int  orc_xT(const int16_t *resi, int resiStride, int32_t *coef, int w, int h, int trHor, int trVer, int bitDepth);
int  orc_xIT(const int32_t *coef, int16_t *resi, int resiStride, int w, int h, int trHor, int trVer, int bitDepth);

extern const int orc_quant_scales[2][6];
extern const int orc_inv_quant_scales[2][6];
void orc_cg_size(int log2w, int log2h, int *log2CGw, int *log2CGh);
int  orc_scan_order(int log2w, int log2h, uint32_t *out);
void orc_quant_params(int w, int h, int bitDepth, int qp, int isIRAP, int *quantCoeff, int *iQBits, int64_t *iAdd);
void orc_dequant_params(int w, int h, int bitDepth, int qp, int *scale, int *rightShift, int *inputMaximum);
void orc_need_rdoq_params(int w, int h, int bitDepth, int qp, int isLuma, int *quantCoeff, int *iQBits, int64_t *iAdd, int *numCoeff);
void orc_quant_core(const int32_t *coef, int16_t *qcoef, int32_t *deltaU, int w, int h, int quantCoeff, int iQBits,
                    int64_t iAdd, int thrVal, int32_t *absSumOut, int *lastScanPosOut);
/* QuantCore with CodingUnit::lfnstIdx (Quant.cpp:149-159: first coefficient group only; 8 positions for 4x4 / 8x8 TUs) */
void orc_quant_core_lfnst(const int32_t *coef, int16_t *qcoef, int32_t *deltaU, int w, int h, int quantCoeff, int iQBits,
                          int64_t iAdd, int thrVal, int lfnstIdx, int32_t *absSumOut, int *lastScanPosOut);
void orc_dequant_core(int maxX, int maxY, int scale, const int16_t *q, size_t qStride, int32_t *coef, int rightShift,
                      int inputMaximum, int32_t transformMaximum);
int  orc_need_rdoq(const int32_t *coef, size_t num, int quantCoeff, int64_t offset, int shift);

extern const int16_t orc_mctf_filter6[16][8];
extern const int16_t orc_mctf_filter4[16][4];
int    orc_mctf_err_int(const int16_t *org, ptrdiff_t os, const int16_t *buf, ptrdiff_t bs, int w, int h);
int    orc_mctf_err_frac(int tap4, const int16_t *org, ptrdiff_t os, const int16_t *buf, ptrdiff_t bs, int w, int h,
                         int fx, int fy, int bitDepth);
double orc_mctf_calc_var(const int16_t *org, ptrdiff_t stride, int w, int h);
void   orc_extend_border(int16_t *buf, int stride, int w, int h, int pad);
void   orc_mctf_subsample(const int16_t *src, int srcStride, int w, int h, int16_t *dst, int dstStride);
int    orc_mctf_me(const int16_t *orgLuma, const int16_t *refLuma, int width, int height, int bitDepth, int unitSize,
                   int mctfSpeed, int addLevel, orc_mv_t **levelOut, int *levelDims);
/* motionErrorLuma calls of orc_mctf_me since the last reset: { integer calls, their bytes (4 w h), fractional calls, their bytes } (the reference's schedule, MCTF.cpp:1189-1306) */
void   orc_mctf_count_reset(void);
void   orc_mctf_count_get(uint64_t *out4);

/* sub-pel interpolation (InterpolationFilter.cpp) */
int  orc_if_coeff(int set, int phase, int16_t coeff[8]);
void orc_if_filter(int N, int isVertical, int isFirst, int isLast, int bitDepth, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width, int height, const int16_t *coeff);
void orc_if_copy(int isFirst, int isLast, int bitDepth, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width, int height, int biMCForDMVR);
void orc_if_luma_1d(int vertical, const int16_t *src, int srcStride, int16_t *dst, int dstStride, int width, int height, int frac, int isFirst, int isLast, int bitDepth, int useAltHpelIf, int reduceTap);
void orc_if_pred_luma(const int16_t *ref, int refStride, int16_t *dst, int dstStride, int width, int height, int xFrac, int yFrac, int rndRes, int bitDepth, int useAltHpelIf);

/* MCTF apply (MCTF.cpp:259-518, 1399-1552) */
void orc_mctf_apply_frac(int tap4, const int16_t *org, ptrdiff_t os, int16_t *dst, ptrdiff_t ds, int w, int h, int fx, int fy, int bitDepth);
void orc_mctf_planar_correction(const int16_t *ref, ptrdiff_t rs, int16_t *dst, ptrdiff_t ds, int w, int h, int bitDepth, uint16_t motionError);
void orc_mctf_apply_block(const int16_t *src, ptrdiff_t ss, int16_t *dst, ptrdiff_t ds, int w, int h, int bitDepth, const int16_t *const *corrected,
                          int numRefs, const int *verror, const double *refStrengths, double weightScaling, double sigmaSq);
void orc_mctf_bilateral_plane(const int16_t *org, ptrdiff_t orgStride, int width, int height, int cs, int bitDepth, int unitSize, int lowResFltApply,
                              int qp, int numRefs, const int16_t *const *refs, ptrdiff_t refStride, const orc_mv_t *const *mvs, int mvW,
                              const double *refStrengths, double weightScaling, double sigmaSq, int16_t *out, ptrdiff_t outStride);

/* DMVR refinement search (InterPrediction.cpp:1167-1187, 1227-1244, 1312-1384; InterpolationFilter.cpp:662-681) */
void     orc_if_bilinear(const int16_t *src, int srcStride, int16_t *dst, int dstStride, int w, int h, int fracX, int fracY, int bitDepth);
void     orc_dmvr_subpel_error_surface(const uint64_t sad[5], int32_t deltaMv[2]);
uint64_t orc_dmvr_search(const int16_t *l0c, const int16_t *l1c, int stride, int dx, int dy, int16_t mvd[2]);
uint64_t orc_dmvr_refine(const int16_t *ref0, int stride0, int fx0, int fy0, const int16_t *ref1, int stride1, int fx1, int fy1, int dx, int dy,
                         int bitDepth, int16_t mvd[2]);

/* ALF encoder statistics (AdaptiveLoopFilter.cpp:524-728, EncAdaptiveLoopFilter.cpp:3266-3541, 3707-3921;
 * x86/EncAdaptiveLoopFilterX86.h:160-236).  Planes carry a replicated border of >= 4 samples (the reference works on its extended m_tempBuf).
 * cls: 2 bytes per 4x4 block {classIdx, transposeIdx}, (width/4) per row.  Statistics record per class: ORC_ALF_REC floats =
 * E[13][13] (row-major, symmetric), y[13], pixAcc; for the 5x5 chroma shape only the first 7 rows/columns are used.                 */
#define ORC_ALF_REC (13 * 13 + 13 + 1)
void orc_alf_classify(const int16_t *rec, ptrdiff_t stride, int width, int height, int shift, int vbCTUHeight, int vbPos, uint8_t *cls);
void orc_alf_elocal(const int16_t *rec, ptrdiff_t stride, int halfFilterLength, int transposeIdx, const int clipTopRow[4], const int clipBotRow[4], int16_t *ELocal /* [numCoeff][16] */);
void orc_alf_stats_area(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int x0, int y0, int w, int h, int filterLength,
                        const uint8_t *cls /* NULL: one class */, int clsStride /* blocks per picture row */, int vbCTUHeight, int vbPos, float *out /* [numClasses][ORC_ALF_REC], accumulated into */);
void orc_alf_stats_plane(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int width, int height, int ctuSize, int filterLength,
                         const uint8_t *cls, int vbCTUHeight, int vbPos, float *out /* [numCtus][numClasses][ORC_ALF_REC], zeroed here */);
void orc_alf_stats_plane_acc(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int width, int height, int ctuSize, int filterLength,
                             const uint8_t *cls, int vbCTUHeight, int vbPos, float *out /* continues from the records already there */);

/* statistics units of unitSize x unitSize samples made of CTUs (getStatisticsASU :1568-1590): CTUs of a unit in raster order, one record set per unit */
void orc_alf_stats_plane_units(const int16_t *org, ptrdiff_t orgStride, const int16_t *rec, ptrdiff_t recStride, int width, int height, int unitSize, int ctuSize, int filterLength,
                               const uint8_t *cls, int vbCTUHeight, int vbPos, float *out /* [numUnits][numClasses][ORC_ALF_REC], zeroed here */);

/* CC-ALF statistics (EncAdaptiveLoopFilter.cpp:6061-6357 getBlkStatsCcAlf, :6359-6422 calcCovariance4CcAlf): per chroma CTU one record (the first
 * 7 rows / columns of E, y[0..6], pixAcc); local terms = 7 luma differences around the co-located luma sample, target = org - ALF-filtered chroma.
 * recLuma carries a replicated border >= 2.  sx / sy = chroma subsampling shifts; vbCTUHeight / vbPos / picHeight in luma samples.                */
void orc_ccalf_stats_plane(const int16_t *orgC, ptrdiff_t orgStride, const int16_t *slfC, ptrdiff_t slfStride, const int16_t *recLuma, ptrdiff_t recStride,
                           int widthC, int heightC, int ctuSizeC, int sx, int sy, int vbCTUHeight, int vbPos, int picHeight, float *out /* [numCtus][ORC_ALF_REC], continues from the records there */);

/* ALF / CC-ALF filtering of a plane (AdaptiveLoopFilter.cpp:730-967 filterBlk, :969-1058 filterBlkCcAlf; call sites EncAdaptiveLoopFilter.cpp:2035-2066, :6606-6699).
 * src carries a replicated border of >= 3 (7x7) / 2 (5x5) samples; coeffSets / clipSets: [numSets][numClasses][13] (25 classes with cls, 1 without);
 * ctuSet[ctu] < 0: CTU not filtered (dst untouched).  CC-ALF: coeff [numFilters][8], ctuFilter[ctu] 0 = off, k = filter k-1; dstC is corrected in place. */
void orc_alf_filter_plane(const int16_t *src, ptrdiff_t srcStride, int16_t *dst, ptrdiff_t dstStride, int width, int height, int ctuSize, int bitDepth, int filterLength,
                          const uint8_t *cls, const int16_t *coeffSets, const int16_t *clipSets, const int16_t *ctuSet, int vbCTUHeight, int vbPos);
void orc_ccalf_filter_plane(int16_t *dstC, ptrdiff_t dstStride, const int16_t *recLuma, ptrdiff_t recStride, int widthC, int heightC, int ctuSizeC, int sx, int sy, int bitDepth,
                            const int16_t *coeff, const uint8_t *ctuFilter, int vbCTUHeight, int vbPos);

#ifdef __cplusplus
}
#endif
#endif
