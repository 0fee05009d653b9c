// vvenc_hip_shim.h — C++ host side ABOVE the C ABI (include/vvenc_hip.h): the table-shaped mirror of the reference's
// kernel objects for this path.  Same names, argument meaning and calling convention as the reference so that the
// encoder's control logic (EncCu / InterSearch / TrQuant / MCTF) can call through it unmodified:
//
//   vvhip::RdCost::m_afpDistortFunc[2][DF_TOTAL_FUNCTIONS], m_afpDistortFuncX5[2]   <- CommonLib/RdCost.h:117-121
//   vvhip::DistParam / CPelBuf / FpDistFunc / FpDistFuncX5                          <- CommonLib/RdCost.h:74-111, Buffer.h:149-159
//   vvhip::TCoeffOps (cpyResi/cpyCoeff/fastInvCore/fastFwdCore_2D/roundClip)        <- CommonLib/TrQuant_EMT.h:63-91
//   vvhip::QuantOps (xQuant/xDeQuant/xNeedRdoq core signatures)                     <- CommonLib/Quant.h:143-151
//   vvhip::MCTFOps  (m_motionErrorLumaInt8, m_motionErrorLumaFrac8[2], m_calcVar)   <- CommonLib/MCTF.h:160-170
//
// Two ways to call:
//   (1) table entry, one candidate per call — the reference's synchronous signature.  Pointers are HOST pointers; if they
//       fall inside a picture registered with registerPicture() the call only ships two offsets, otherwise the two blocks
//       are staged.  Correct everywhere and re-entrant (per-thread contexts), but one launch + one round trip per call: plumbing, not speed.
//   (2) batching: enqueue( DistParam ) -> ticket, flush(), result( ticket ) — what the thin shim in INTEGRATION.md uses
//       inside xTZSearch / xPatternRefinement (all positions of a ring / raster / refinement are known before the first
//       cost is needed) and for whole MCTF levels.
// Errors: like the reference there are no return codes; any failure throws vvhip::Exception (THROW, TypeDef.h:635-636).
// There is no CPU fallback: without a GPU create() throws.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/vvenc_hip.h"

namespace vvhip {

using Pel        = int16_t;
using TCoeff     = int32_t;
using TCoeffSig  = int16_t;
using TMatrixCoeff = int16_t;
using Distortion = uint64_t;

struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };

// DFunc, CommonLib/TypeDef.h:339-382 (same order: index = base + log2(width))
enum DFunc
{
  DF_SSE = 0, DF_SSE2, DF_SSE4, DF_SSE8, DF_SSE16, DF_SSE32, DF_SSE64, DF_SSE128,
  DF_SAD = 8, DF_SAD2, DF_SAD4, DF_SAD8, DF_SAD16, DF_SAD32, DF_SAD64, DF_SAD128,
  DF_HAD = 16, DF_HAD2, DF_HAD4, DF_HAD8, DF_HAD16, DF_HAD32, DF_HAD64, DF_HAD128,
  DF_HAD_2SAD = 24, DF_SAD_WITH_MASK = 25,
  DF_HAD_fast = 26, DF_HAD2_fast, DF_HAD4_fast, DF_HAD8_fast, DF_HAD16_fast, DF_HAD32_fast, DF_HAD64_fast, DF_HAD128_fast,
  DF_TOTAL_FUNCTIONS = 34
};

struct CPelBuf { const Pel* buf = nullptr; int stride = 0; unsigned width = 0, height = 0; };

class DistParam;
typedef Distortion ( *FpDistFunc )( const DistParam& );
typedef void ( *FpDistFuncX5 )( const DistParam&, Distortion*, bool );

class DistParam
{
public:
  CPelBuf      org, cur;
  FpDistFunc   distFunc  = nullptr;
  FpDistFuncX5 dmvrSadX5 = nullptr;
  int          bitDepth  = 0;
  int          subShift  = 0;
  int          compID    = 0;
  bool         applyWeight = false;
  Distortion   maximumDistortionForEarlyExit = ~0ull;   // honoured as in the SIMD rows: ignored (full sums are returned)
  const void*    wpCur   = nullptr;                     // weighted prediction is refused like the reference does (RdCost.cpp:303-306)
  const CPelBuf* orgLuma = nullptr;
  // GEO masked SAD (DF_SAD_WITH_MASK, RdCost.h:100-103): mask walks +stepX per sample, +maskStride*(1<<subShift)+maskStride2 per row
  const Pel*   mask        = nullptr;
  int          maskStride  = 0;
  int          stepX       = 0;
  int          maskStride2 = 0;
};
typedef Distortion ( *FpFxdWtdDistFunc )( const DistParam&, uint32_t fixedWeight );      // RdCost.h:117

// Worker contexts and the registry of host pictures mirrored in HBM.
//   * Every encoder worker thread gets its OWN context (HIP stream, staging areas) on the GPU it currently serves, created on first use and recycled when the
//     thread ends: table entries called concurrently by the reference's workers (one RdCost / TrQuant / Quant per worker, shared g_tCoeffOps / MCTF; SURVEY 8b
//     "Threading") never wait on each other on the host.
//   * Several GPUs in one process: selectGpu() binds the calling thread to a device (pictures are mapped to devices by the binding: one picture <-> one device,
//     SURVEY 8e); mirrors live in a per-GPU registry shared by that GPU's contexts; copyMirror() moves a picture between GPUs over xGMI.
//   * Host buffers the encoder recycles (picture planes) can be pinned in place once (pinHost) so that their transfers are asynchronous DMA at PCIe rate.
class Device
{
public:
  static Device& get();                      // the calling thread's context on its selected GPU (creates it on first use; throws without a GPU)
  static void    selectGpu( int gpu );       // bind the calling thread to a device (< 0: the default, $VVHIP_DEVICE or 0)
  static int     selectedGpu();              // the calling thread's binding as selectGpu set it (< 0: default) — scopes save and restore it
  static int     gpuCount();                 // devices visible to the process
  static int     defaultGpu();
  int        gpu() const { return m_gpu; }
  vvhip_ctx* ctx() const { return m_ctx; }
  // Mirror a host plane (sample (0,0) at `origin`, `margin` samples around a w x h picture, line pitch `stride`) in this GPU's HBM.
  // Re-register (or call updatePicture) after the host changed it (e.g. a reference picture was reconstructed).
  // findable = false: the mirror is used through its id only and never substituted for host pointers by the per-call table entries
  // (for pictures whose host buffer may be rewritten while the mirror is kept)
  int  registerPicture( const Pel* origin, int stride, int width, int height, int margin, bool findable = true, bool upload = true );
  void updatePicture( int id );
  // rows [y0, y0 + rows) of the picture (y relative to sample row 0; the margins are rows < 0 and >= height), whole padded lines: a reconstructed picture is
  // mirrored CTU row by CTU row as the encoder finishes (and border-extends) it.  Synchronous: other threads' contexts may read the rows when this returns.
  void updatePictureRows( int id, int y0, int rows );
  void unregisterPicture( int id );
  // the same picture on another GPU: allocates a mirror there and fills it device-to-device (hipMemcpyPeerAsync over xGMI); returns the id in `dst`'s registry
  int  copyMirrorTo( int id, Device& dst );
  struct Mirror { const Pel* hostBase; const Pel* hostEnd; const Pel* origin; int stride, width, height, margin; int16_t* dBase; int16_t* dOrigin; bool live; bool findable; bool reference; };
  // (find / findReference / mirror hand out COPIES taken under the registry's lock: another thread may register into a freed slot at any time; the device memory of a picture
  //  is released by unregisterPicture only — the caller that owns the picture's life cycle — and hipFree itself waits for the device's work in flight)
  struct Found { bool ok = false; Mirror m; explicit operator bool() const { return ok; } const Mirror* operator->() const { return &m; } const Mirror& operator*() const { return m; } };
  Found find( const Pel* p ) const;  // which registered picture of this GPU contains host pointer p (nullptr: none)
  // reference pictures (reconstructions mirrored row by row, setReference): looked up by the motion-search entry points only — never by the generic table entries,
  // which may be handed the same host buffer while it is being rewritten as the current picture's reconstruction
  void setReference( int id );
  Found findReference( const Pel* p ) const;
  Mirror mirror( int id ) const;
  void check( int rc, const char* what ) const;
  int16_t* staging( size_t bytes );          // grow-only device scratch of THIS context for unregistered (compact temp) buffers
  void*    stagingAux( size_t bytes );
  // pin a recycled host buffer in place (hipHostRegister, once per range; false: left pageable).  $VVHIP_PIN=0 switches pinning off.
  static bool pinHost( const void* p, size_t bytes );
  static void unpinAll();                    // before the owner frees the buffers (encoder close)
  // traffic over PCIe / calls since the process started (all contexts): what a binding prints per picture
  struct Stats { uint64_t uploadBytes, downloadBytes, uploads, downloads, contexts, residentReferenceCalls; };
  static Stats stats();
private:
  explicit Device( int gpu );
  ~Device();
  friend struct DevicePool;
  int        m_gpu = 0;
  vvhip_ctx* m_ctx = nullptr;
  int16_t* m_stage = nullptr; size_t m_stageBytes = 0;
  void* m_aux = nullptr; size_t m_auxBytes = 0;
};

// grow-only pinned host area owned by the shim (hipHostMalloc through the C ABI): where picture-sized results land before they are scattered into the encoder's buffers
class PinnedBuffer
{
public:
  Pel* get( size_t elems );                  // at least `elems` samples (contents are not preserved when it grows)
  ~PinnedBuffer() {}                         // (released with the process: HIP teardown order at exit is not ours to rely on)
private:
  Pel* m_p = nullptr; size_t m_elems = 0;
};

class RdCost
{
public:
  FpDistFunc   m_afpDistortFunc[2][DF_TOTAL_FUNCTIONS];
  FpDistFuncX5 m_afpDistortFuncX5[2];
  FpFxdWtdDistFunc m_fxdWtdPredPtr;            // fixWeightedSSE_Core, RdCost.cpp:1948-1982
  void create( bool enableOpt = true );      // both rows point at the HIP entries (bit-exact for every bit depth)

  // RdCost::setDistParam, CommonLib/RdCost.cpp:158-206 (subShiftMode / useHadamard semantics identical)
  void setDistParam( DistParam& dp, const CPelBuf& org, const Pel* refY, int refStride, int bitDepth, int compID, int subShiftMode = 0, int useHadamard = 0 );
  // RdCost::setDistParamGeo, CommonLib/RdCost.cpp:2036-2060
  void setDistParamGeo( DistParam& dp, const CPelBuf& org, const Pel* refY, int refStride, const Pel* mask, int maskStride, int stepX, int maskStride2, int bitDepth, int compID );
  // RdCost::getDistPart, CommonLib/RdCost.cpp:267-291 (luma; chroma weighting stays with the caller)
  Distortion getDistPart( const CPelBuf& org, const CPelBuf& cur, int bitDepth, DFunc eDFunc );

  // One stage of InterSearch::xPatternRefinement (EncoderLib/InterSearch.cpp:760-880) in ONE device call: the distortion of the original block
  // against the reference block interpolated at each of n (<= 9) displacements qpel[i] = (hor, ver) in quarter samples around refBlk
  // (= pattern->buf, the block at the best integer vector; the reference picture's margin must be readable, 6 samples are touched).
  // hadMode 0 SAD, 1 HAD, 2 HAD_fast (m_bUseHADME / m_fastHad); reduceTap = m_meReduceTap; returns false for shapes the device entry
  // does not take (the caller keeps its CPU path).  The MV-bit cost and the strict-< update stay with the caller's loop.
  bool patternRefineCosts( const CPelBuf& org, const Pel* refBlk, int refStride, const int ( *qpel )[2], int n, int bitDepth, int hadMode, int reduceTap, bool useAltHpelIf,
                           Distortion* out );

  // The candidate positions of one integer-search stage (a TZ diamond round, a raster row, a refinement star; InterSearch::xTZSearchHelp,
  // EncoderLib/InterSearch.cpp:410-438) in ONE device call: out[i] = func( org, refBase + xy[i][1]*refStride + xy[i][0] ) with the given
  // subShift.  Host buffers: the original block and the bounding window of the positions are staged per call.
  void distAtPositions( int func, const CPelBuf& org, const Pel* refBase, int refStride, int subShift, int bitDepth, const int ( *xy )[2], int n, Distortion* out );

  // ---- batching (2) ----
  int        enqueue( const DistParam& dp );         // dp.distFunc must be one of this object's table entries
  void       flush();                                // one launch per (function, block size, subShift, plane pair) group
  Distortion result( int ticket ) const { return m_results[ticket]; }
  void       clear() { m_pending.clear(); m_results.clear(); }
private:
  struct Pending { int func, w, h, subShift; Device::Mirror mo, mc; int32_t orgOff, curOff; };
  std::vector<Pending>    m_pending;
  std::vector<Distortion> m_results;
};

// TCoeffOps, CommonLib/TrQuant_EMT.h:63-91 — same member names and signatures, host pointers, one call = one (synchronous) launch.
struct TCoeffOps
{
  TCoeffOps();
  // the table's own ten slots, reference signatures (host pointers; dst of fastInvCore is accumulated into, as in the reference)
  void ( *cpyResi8 )( const TCoeff* src, Pel* dst, ptrdiff_t stride, unsigned width, unsigned height );
  void ( *cpyResi4 )( const TCoeff* src, Pel* dst, ptrdiff_t stride, unsigned width, unsigned height );
  void ( *cpyCoeff8 )( const Pel* src, ptrdiff_t stride, TCoeff* dst, unsigned width, unsigned height );
  void ( *cpyCoeff4 )( const Pel* src, ptrdiff_t stride, TCoeff* dst, unsigned width, unsigned height );
  void ( *fastInvCore[5] )( const TMatrixCoeff* it, const TCoeff* src, TCoeff* dst, unsigned lines, unsigned reducedLines, unsigned rows );
  void ( *fastFwdCore_2D[5] )( const TMatrixCoeff* it, const TCoeff* src, TCoeff* dst, unsigned lines, unsigned reducedLines, unsigned cutoff, int shift );
  void ( *fastFwdCore_1D[5] )( const TMatrixCoeff* it, const TCoeff* src, TCoeff* dst, unsigned lines, unsigned reducedLines, unsigned cutoff, int shift );
  void ( *roundClip4 )( TCoeff* dst, unsigned width, unsigned height, unsigned stride, const TCoeff outputMin, const TCoeff outputMax, const TCoeff round, const TCoeff shift );
  void ( *roundClip8 )( TCoeff* dst, unsigned width, unsigned height, unsigned stride, const TCoeff outputMin, const TCoeff outputMax, const TCoeff round, const TCoeff shift );
  // 2-D drivers with the signature of TrQuant::xT / xIT's inner work (TrQuant.cpp:481-655): what a batching integration calls instead
  // of two 1-D passes + copies, because both passes (and the Pel <-> TCoeff conversion) are fused on the device.
  void ( *fwdTransform2D )( const Pel* resi, ptrdiff_t stride, TCoeff* coef, unsigned width, unsigned height, int trTypeHor, int trTypeVer, int bitDepth );
  void ( *invTransform2D )( const TCoeff* coef, Pel* resi, ptrdiff_t stride, unsigned width, unsigned height, int trTypeHor, int trTypeVer, int bitDepth );
};
extern TCoeffOps g_tCoeffOps;

// Quant::xQuant/xDeQuant/xNeedRdoq (CommonLib/Quant.h:143-151) with the TransformUnit reduced to what QuantCore reads (width, height).
struct QuantOps
{
  QuantOps();
  void ( *xDeQuant )( const int maxX, const int maxY, const int scale, const TCoeffSig* const piQCoef, const size_t piQCfStride, TCoeff* const piCoef,
                      const int rightShift, const int inputMaximum, const TCoeff transformMaximum );
  bool ( *xNeedRdoq )( const TCoeff* pCoeff, size_t numCoeff, int quantCoeff, int64_t offset, int shift );
  void ( *xQuant )( unsigned width, unsigned height, const TCoeff* piCoef, TCoeffSig* piQCoef, TCoeff& uiAbsSum, int& lastScanPos, TCoeff* deltaU,
                    const int qp, const bool isIRAP, const int bitDepth, const TCoeff thrVal );
  // QuantCore's own argument list (Quant.cpp:132): what a trampoline for Quant::xQuant forwards (piQCoef compact, stride = width)
  void ( *xQuantCore )( unsigned width, unsigned height, const TCoeff* piCoef, TCoeffSig* piQCoef, TCoeff& uiAbsSum, int& lastScanPos, TCoeff* deltaU,
                        const int defaultQuantisationCoefficient, const int iQBits, const int64_t iAdd, const TCoeff thrVal );
  // the same for a TU whose coding unit has lfnstIdx > 0: QuantCore's first-coefficient-group rule (Quant.cpp:149-159)
  void ( *xQuantCoreLfnst )( unsigned width, unsigned height, const TCoeff* piCoef, TCoeffSig* piQCoef, TCoeff& uiAbsSum, int& lastScanPos, TCoeff* deltaU,
                             const int defaultQuantisationCoefficient, const int iQBits, const int64_t iAdd, const TCoeff thrVal, const int lfnstIdx );
};

// DMVR refinement search of one CU in one device call (DMVR::xProcessDMVR, CommonLib/InterPrediction.cpp:1262-1392).
struct DMVROps
{
  // ref0 / ref1: the reference samples at the CU's position displaced by the integer part of the (clipped) merge vectors MINUS 2 samples in
  // both directions (= what the reference hands to its bilinear xPredInterBlk, :1285-1302); frac* = the vectors' 1/16 fractions.
  // Sub-blocks dx x dy (<= 16) in raster order: mvd[2*num], mvd[2*num+1] = cu.mvdL0SubPu[num]; minCost[num] = the value compared with 2*dx*dy (:1386).
  bool refineCu( const Pel* ref0, int stride0, int fx0, int fy0, const Pel* ref1, int stride1, int fx1, int fy1, int cuWidth, int cuHeight, int dx, int dy, int bitDepth,
                 int16_t* mvd, uint64_t* minCost );
};

// ALF encoder statistics: whole-plane forms of AdaptiveLoopFilter::m_deriveClassificationBlk (CommonLib/AdaptiveLoopFilter.h,
// table entry set at AdaptiveLoopFilter.cpp:73) and EncAdaptiveLoopFilter::getPreBlkStats + m_getPreBlkStatsAccum (EncAdaptiveLoopFilter.h:438).
// rec points at sample (0,0) of a plane that carries a replicated border of >= 4 samples (the reference's extended m_tempBuf).
struct ALFOps
{
  // cls: 2 bytes per 4x4 block {classIdx, transposeIdx} = AlfClassifier, width/4 per row.  vbCTUHeight / vbPos = m_alfVBLumaCTUHeight / m_alfVBLumaPos.
  bool deriveClassification( const Pel* rec, int recStride, int width, int height, int bitDepth, int vbCTUHeight, int vbPos, uint8_t* cls );
  // per CTU and class one record of 183 floats: E[13][13], y[13], pixAcc (AlfCovariance with numBins 1); cls == nullptr: chroma (filterLength 5, one class);
  // init (optional, same layout): the values the float chains start from (statistics units that span several CTUs)
  bool getStatistics( const Pel* org, int orgStride, const Pel* rec, int recStride, int width, int height, int ctuSize, int filterLength,
                      const uint8_t* cls, int vbCTUHeight, int vbPos, float* out, const float* init = nullptr );
  // whole-picture form (4:2:0): every plane is uploaded once; classes of the luma blocks (picture raster) and the records of every statistics unit
  // (unitSize x unitSize luma samples, made of CTUs of ctuSize: EncAdaptiveLoopFilter::getStatisticsASU) for the enabled components.
  // stats[c]: [numUnits][c == 0 ? 25 : 1][183]
  bool pictureStatistics( const Pel* const rec[3], const int recStride[3], const Pel* const org[3], const int orgStride[3], int width, int height, int bitDepth,
                          int ctuSize, int unitSize, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3], uint8_t* cls, float* const stats[3] );
  // The same picture call IN BANDS, issued by the encoder's row tasks while the picture's SAO is still running elsewhere (EncoderLib/EncSlice.cpp:1135-1167: the statistics task
  // of a CTU row starts when the row below it has left SAO, i.e. when every sample the row's statistics read is final): statisticsBegin sets the picture up (any thread),
  // statisticsBand( u ) uploads the rows of statistics-unit row u (+ 4 border rows on either side) from the encoder's planes, runs classification + statistics of those units and
  // requests their download into pinned host memory — all asynchronous on the CALLING thread's stream, completion marked by an event; statisticsEnd (the thread that runs
  // EncAdaptiveLoopFilter::deriveFilter, EncoderLib/EncAdaptiveLoopFilter.cpp:1757) waits for the marks and hands out the host copies.  Same kernels on the same samples as
  // pictureStatistics (a band starts on a CTU boundary: every position relative to the virtual boundaries is the one it has in the picture): bit-identical records.
  // The caller serialises the three calls per object (the binding's per-ALF-object lock); bands may come in any order, each exactly once.
  bool statisticsBegin( const int recStride[3], const int orgStride[3], int width, int height, int bitDepth, int ctuSize, int unitSize, int vbLumaH, int vbLumaPos,
                        int vbChromaH, int vbChromaPos, const bool enabled[3] );
  int  statisticsBands() const { return m_band.rows; }                                           // unit rows of the picture begun last
  bool statisticsBand( int unitRow, const Pel* const rec[3], const Pel* const org[3] );
  bool statisticsEnd( const Pel* const rec[3], const uint8_t** cls, const float* stats[3] );      // false: not every band was issued (the caller falls back to pictureStatistics)
  // EncAdaptiveLoopFilter::getBlkStatsCcAlf per chroma CTU (4:2:0): org / slf = chroma planes (slf = ALF-filtered), recLuma with a replicated border >= 2;
  // one record per chroma CTU (E[0..6][0..6], y[0..6], pixAcc), vb* / picHeight in luma samples
  bool getStatisticsCcAlf( const Pel* orgC, int orgStride, const Pel* slfC, int slfStride, const Pel* recLuma, int recStride, int widthC, int heightC, int ctuSizeC,
                           int vbCTUHeight, int vbPos, int picHeight, float* out, const float* init = nullptr );
  // AdaptiveLoopFilter::m_filter7x7Blk / m_filter5x5Blk (CommonLib/AdaptiveLoopFilter.h:129-136) over the enabled CTUs of a plane, the way EncAdaptiveLoopFilter::reconstructCTU
  // drives them: src with a replicated border >= 4; dst receives the filtered samples of the CTUs with ctuSet[ctu] >= 0 (filter set of the CTU), the others keep theirs.
  // coeffSets / clipSets: [numSets][cls ? 25 : 1][13]; clipSets == nullptr selects the linear entries ([0]).  cls as written by deriveClassification (luma), nullptr for chroma.
  bool filterPlane( const Pel* src, int srcStride, Pel* dst, int dstStride, int width, int height, int ctuSize, int bitDepth, int filterLength, const uint8_t* cls,
                    const short* coeffSets, const short* clipSets, int numSets, const short* ctuSet, int vbCTUHeight, int vbPos );
  // whole-picture form (4:2:0) of the ALF reconstruction: luma with numLumaSets filter sets (lumaCtuSet == nullptr: luma off), both chroma planes with the numChromaSets
  // alternatives (chromaCtuSet[c] == nullptr: plane off); src[c] = the unfiltered planes with their replicated border, dst[c] = the reconstruction picture
  bool filterPicture( const Pel* const src[3], const int srcStride[3], Pel* const dst[3], const int dstStride[3], int width, int height, int bitDepth, int ctuSize,
                      const uint8_t* cls, const short* lumaCoeff, const short* lumaClip, int numLumaSets, const short* lumaCtuSet,
                      const short* chromaCoeff, const short* chromaClip, int numChromaSets, const short* const chromaCtuSet[2],
                      int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos );
  // AdaptiveLoopFilter::m_filterCcAlf (:124) over a chroma plane (4:2:0) as applyCcAlfFilterCTU drives it: dstC corrected in place; coeff [numFilters][8]; ctuFilter[ctu] 0 = off
  bool filterCcAlf( Pel* dstC, int dstStride, const Pel* recLuma, int recStride, int widthC, int heightC, int ctuSizeC, int bitDepth, const int16_t* coeff, int numFilters,
                    const uint8_t* ctuFilter, int vbCTUHeight, int vbPos );
  // One ALFOps object per EncAdaptiveLoopFilter: pictureStatistics leaves the unfiltered planes (with their border) and the block classes of ITS picture in HBM;
  // filterPicture, handed the same host planes afterwards (statistics -> derivation on the host -> filtering, EncAdaptiveLoopFilter.cpp:1440-1520, :1902), skips
  // their upload.  dropResident() when the host planes change without a statistics call.
  void dropResident() { m_res.valid = false; }
  ~ALFOps();
private:
  struct Resident { bool valid = false; int gpu = -1, width = 0, height = 0; const Pel* rec[3] = { nullptr, nullptr, nullptr }; int stride[3] = { 0, 0, 0 };
                    int16_t* d = nullptr; size_t elems = 0, off[3] = { 0, 0, 0 }; uint8_t* dCls = nullptr; size_t clsBytes = 0; };
  Resident m_res;
  struct Banded { bool open = false; int gpu = -1, rows = 0, issued = 0, width = 0, height = 0, bitDepth = 0, ctuSize = 0, unitSize = 0, vbLumaH = 0, vbLumaPos = 0, vbChromaH = 0, vbChromaPos = 0;
                  bool enabled[3] = { false, false, false }; int rp[3] = { 0, 0, 0 }, op[3] = { 0, 0, 0 }; size_t rOff[3] = { 0, 0, 0 }, oOff[3] = { 0, 0, 0 }, stOff[3] = { 0, 0, 0 };
                  int16_t* dOrg = nullptr; size_t orgElems = 0; char* dSt = nullptr; size_t stBytes = 0; char* host = nullptr; size_t hostBytes = 0, nCls = 0;
                  std::vector<void*> events; std::vector<char> done; };
  Banded m_band;
  PinnedBuffer m_down;          // download area of filterPlane
  bool filterPlaneImpl( const Pel* src, int srcStride, const int16_t* dSrcResident, const uint8_t* dClsResident, Pel* dst, int dstStride, int width, int height, int ctuSize, int bitDepth,
                        int filterLength, const uint8_t* cls, const short* coeffSets, const short* clipSets, int numSets, const short* ctuSet, int vbCTUHeight, int vbPos );
};

// MCTF table, CommonLib/MCTF.h:160-170
struct MCTFOps
{
  MCTFOps();
  int ( *m_motionErrorLumaInt8 )( const Pel* org, const ptrdiff_t origStride, const Pel* buf, const ptrdiff_t buffStride, const int w, const int h, const int besterror );
  int ( *m_motionErrorLumaFrac8[2] )( const Pel* org, const ptrdiff_t origStride, const Pel* buf, const ptrdiff_t buffStride, const int w, const int h,
                                      const int16_t* xFilter, const int16_t* yFilter, const int bitDepth, const int besterror );
  double ( *m_calcVar )( const Pel* org, const ptrdiff_t origStride, const int w, const int h );
  // whole-picture replacement of MCTF::motionEstimationMCTF (MCTF.cpp:666-707) for pictures registered with Device:
  // out[r] receives ceil(w/unit) x ceil(h/unit) vvhip_mv (== MotionVector, MCTF.h:72-82) for reference r
  void motionEstimation( int curPicId, const int* refPicIds, int nRefs, int bitDepth, int unitSize, int mctfSpeed, bool addLevel, vvhip_mv** out );
  // whole-picture replacement of MCTF::bilateralFilter (MCTF.cpp:1489-1552) for planes registered with Device, one id per
  // component plane (numComp 1 or 3, 4:2:0): orgIds[c], refIds[3*r + c]; mvs[r] = final-level motion field of reference r (host, as
  // motionEstimation returned it); refStrengths[r] = m_refStrengths[row][index] (MCTF.cpp:112-117,1480); out[c] receives the filtered plane.
  // Equals the reference's scalar row (its x86 row is within +-1 by the reference's own unit test).
  void bilateralFilter( const int* orgIds, const int* refIds, int nRefs, const vvhip_mv* const* mvs, const double* refStrengths, int qp, int bitDepth, int unitSize,
                        bool lowResFltApply, double overallStrength, int numComp, Pel* const* out, const int* outStride );
};

// InterpolationFilter, CommonLib/InterpolationFilter.h:70-155: the function-pointer tables of the separable
// sub-pel filter with the reference's signatures (host pointers, one synchronous launch per call) and the two public dispatchers.
struct ClpRng { int bd; static constexpr int min() { return 0; } int max() const { return ( 1 << bd ) - 1; } };      // CommonDef.h:542-548
typedef int16_t TFilterCoeff;
class InterpolationFilter
{
public:
  InterpolationFilter();
  void initInterpolationFilter( bool /*enable*/ ) {}
  // [tap index: 0 = 8, 1 = 4, 2 = 2 (bilinear), 3 = 6 taps][isFirst][isLast]
  void ( *m_filterHor[4][2][2] )( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, TFilterCoeff const* coeff );
  void ( *m_filterVer[4][2][2] )( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, TFilterCoeff const* coeff );
  void ( *m_filterCopy[2][2] )( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, bool biMCForDMVR );
  // fused two-pass entries [0 = 8 taps, 1 = 4 taps (, 2 = 2 taps)][isLast]: horizontal pass (14-bit) then vertical pass
  void ( *m_filter4x4[2][2] )( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, TFilterCoeff const* coeffH, TFilterCoeff const* coeffV );
  void ( *m_filter8xH[3][2] )( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, TFilterCoeff const* coeffH, TFilterCoeff const* coeffV );
  void ( *m_filter16xH[3][2] )( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, TFilterCoeff const* coeffH, TFilterCoeff const* coeffV );
  // luma dispatch of InterpolationFilter::filterHor / filterVer (InterpolationFilter.cpp:557-661; nFilterIdx 0)
  void filterHor( Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, int frac, bool isLast, const ClpRng& clpRng, bool useAltHpelIf = false, int reduceTap = 0 );
  void filterVer( Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, int frac, bool isFirst, bool isLast, const ClpRng& clpRng, bool useAltHpelIf = false, int reduceTap = 0 );
  // tap rows as the reference's static tables hold them (8 entries for the luma sets, 4 for chroma)
  static const TFilterCoeff* lumaFilter( int frac );          // m_lumaFilter[frac]
  static const TFilterCoeff* lumaFilter4x4( int frac );       // m_lumaFilter4x4[frac]
  static const TFilterCoeff* lumaAltHpelIFilter();            // m_lumaAltHpelIFilter
  static const TFilterCoeff* chromaFilter( int frac32 );      // m_chromaFilter[frac32]
};

} // namespace vvhip
