// vvenc_hip_binding.h — the hook surface of the VVenC <-> MI355X binding (PRODUCT: what a VVenC maintainer adds to the encoder).
// bindings/vvenc/apply_binding.py inserts one-line calls to g_vvhipHooks into ten reference translation units (the copies are made at build time, nothing of the
// reference is stored here); vvenc_hip_binding.cpp points the hooks at the table-shaped shim (vvenc_amd/csrc/host/vvenc_hip_shim.h) above the C ABI.
// Selection at run time goes through the reference's own switch: --SIMD=HIP[:mask] / vvenc_set_SIMD_extension("HIP[:mask]") (vvencimpl.cpp:800-851).
// All hooks are null by default: the patched library then behaves exactly like the unpatched one (its bitstreams are the CPU encoder's, tests/test_e2e_bitstream.py).
#pragma once
#include <cstddef>
#include <cstdint>

namespace vvenc {
class RdCost; class Quant; class MCTF; class InterpolationFilter; class Mv; class DistParam; struct MotionVector; struct PelStorage;
template<class T> struct AreaBuf;
template<class T> struct Array2D;
}

struct VvhipHooks
{
  void ( *initRdCost )( vvenc::RdCost* );
  void ( *initQuant )( vvenc::Quant* );
  void ( *initMCTF )( vvenc::MCTF* );
  // whole-picture MCTF search on the device: mctfWants answers up front (the CPU pyramid of the original is then skipped), mctfPrefetch scores ALL references of the
  // first loop of MCTF::filter (MCTF.cpp:788-796) in ONE device call, mctfMe hands the field of one reference to motionEstimationMCTF (from that call, or computed now)
  bool ( *mctfWants )( const vvenc::MCTF* m, int width, int height );
  void ( *mctfPrefetch )( vvenc::MCTF* m, const void* picFifoDeque, int dropFront, int dropBack, const vvenc::PelStorage& orig, bool addLevel, int filterPoc );
  // the worker thread starts a CTU task of the picture with this POC: binds the thread to the picture's GPU (one picture <-> one device)
  void ( *bindPicture )( int poc );
  // a CTU row of a picture's luma reconstruction is final (in-loop filters done, borders extended, EncSlice.cpp:1381-1386): rows [y0, y0 + rows) incl. margin rows go to
  // the device mirror of that picture on every GPU in use — the motion-search hooks then address reference pictures in HBM instead of staging windows per call
  void ( *reconRows )( const int16_t* origin, int stride, int width, int height, int margin, int y0, int rows );
  // EncAdaptiveLoopFilter::deriveFilter starts on a new picture (resets the whole-picture filtering bookkeeping of this ALF object)
  void ( *alfBeginPicture )( const void* owner, int poc );
  bool ( *fwd2D )( const int16_t* resi, ptrdiff_t stride, int32_t* coef, unsigned width, unsigned height, int trTypeHor, int trTypeVer, int bitDepth );
  bool ( *inv2D )( const int32_t* coef, int16_t* resi, ptrdiff_t stride, unsigned width, unsigned height, int trTypeHor, int trTypeVer, int bitDepth );
  void ( *initIF )( vvenc::InterpolationFilter* );
  bool ( *patternCosts )( const vvenc::AreaBuf<const int16_t>* key, const vvenc::AreaBuf<const int16_t>* pattern, int baseHor, int baseVer, int iFrac, const vvenc::Mv* refine,
                          int bitDepth, int hadMode, int reduceTap, bool useAltHpelIf, uint64_t* cost9 );
  // integer TZ search: all positions of a diamond round are scored by one device call up front; xTZSearchHelp looks the SAD up
  void ( *tzReset )();
  void ( *tzPrefetch )( const vvenc::DistParam* dp, const int16_t* refY, int refStride, int startX, int startY, int dist, bool corners, int left, int right, int top, int bottom );
  bool ( *tzLookup )( const vvenc::DistParam* dp, const int16_t* refY, int x, int y, uint64_t* sad );
  bool ( *dmvrSearch )( const int16_t* ref0, int stride0, int fx0, int fy0, const int16_t* ref1, int stride1, int fx1, int fy1, int cuWidth, int cuHeight, int dx, int dy, int bitDepth,
                        int16_t* mvd, uint64_t* minCost );
  bool ( *mctfApply )( const vvenc::MCTF*, const vvenc::PelStorage& orgPic, void* srcFrameInfoDeque, vvenc::PelStorage& newOrgPic, double overallStrength );
  // ALF statistics of one CTU: classes of the luma 4x4 blocks and the covariance records of up to three components
  bool ( *alfCtu )( const int16_t* const rec[3], const int recStride[3], const int16_t* const org[3], const int orgStride[3], int width, int height, int chromaShift,
                    int bitDepth, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3], uint8_t* cls /* [h/4][w/4][2] */, float* const stats[3] /* in: the records to continue from, out: updated */ );
  // ALF statistics of a whole picture in one call (classes in picture raster, one record set per statistics unit)
  bool ( *alfPicture )( const void* owner, int poc, const int16_t* const rec[3], const int recStride[3], const int16_t* const org[3], const int orgStride[3], int width, int height, int bitDepth,
                        int ctuSize, int unitSize, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3], uint8_t* cls, float* const stats[3] );
  // the same picture IN BANDS: called by the statistics task of every CTU row (its last CTU; EncSlice.cpp:1135-1167) while other rows are still in SAO — the row's statistics unit
  // goes to the device asynchronously as soon as its CTU rows are complete, alfPicture then only collects (false: the picture goes up as a whole inside alfPicture)
  bool ( *alfRow )( const void* owner, int poc, int ctuRow, const int16_t* const rec[3], const int recStride[3], const int16_t* const org[3], const int orgStride[3], int width, int height,
                    int bitDepth, int ctuSize, int unitSize, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3] );
  // whether the whole-picture statistics call pays for this picture size and thread count (it sits in the serial filter derivation: worth it when the thread pool is saturated)
  bool ( *alfPictureOn )( int numCtusInPic, int numThreads );
  // CC-ALF statistics of one CTU and chroma component: record of 183 floats (E[0..6][0..6] with row pitch 13, y[0..6], pixAcc)
  bool ( *ccAlfCtu )( const int16_t* orgC, int orgStride, const int16_t* slfC, int slfStride, const int16_t* recLuma, int recStride, int widthC, int heightC,
                      int vbCTUHeight, int vbPos, int picHeightFromHere, float* record );
  // ALF filtering of one CTU block (reconstructCTU, branch without virtual picture boundaries): classifier = the CTU's AlfClassifier array (row pitch MAX_CU_SIZE / 4) or
  // nullptr (chroma, 5x5); clip == nullptr: the linear table entry.  CC-ALF: the chroma block is corrected in place from the unfiltered luma.
  bool ( *alfFilterBlk )( const void* classifier, int16_t* dst, int dstStride, const int16_t* src, int srcStride, int width, int height, int filterLength,
                          const short* coeff, const short* clip, int bitDepth, int vbCTUHeight, int vbPos );
  bool ( *ccAlfFilterBlk )( int16_t* dstC, int dstStride, const int16_t* recLuma, int recStride, int widthC, int heightC, const int16_t* coeff, int bitDepth, int vbCTUHeight, int vbPos );
  // ALF reconstruction of a whole picture by the first CTU task that reaches it (the others find it done): owner / poc identify the picture
  bool ( *alfFilterPicture )( const void* owner, int poc, const int16_t* const src[3], const int srcStride[3], int16_t* const dst[3], const int dstStride[3], int width, int height, int bitDepth,
                              int ctuSize, const uint8_t* cls, const short* lumaCoeff, const short* lumaClip, int numLumaSets, const short* lumaCtuSet, const short* chromaCoeff,
                              const short* chromaClip, int numChromaSets, const short* const chromaCtuSet[2], int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, bool alreadyDoneOnly );
  bool ( *mctfMe )( vvenc::MCTF*, const vvenc::PelStorage& refPic, const vvenc::PelStorage& orig, vvenc::Array2D<vvenc::MotionVector>& mvs, bool addLevel, int refPoc, int curPoc );
  // ---- work-list recorder (hook bit 131072, vvenc_hip_recorder.h): the encoder runs on its CPU kernels and every call of the hot path is written down with the operands' places
  void ( *recPicture )( const void* picture );                      // a worker thread starts a CTU task of this Picture
  void ( *recCu )( const void* codingStructure );                   // EncCu::xCompressCU starts on a block (where its compact copy of the original maps to)
  void ( *recMeBegin )( int cuX, int cuY, int w, int h, int list, int refIdx, int refPoc, bool bi, const int16_t* pattern, int patternStride, const int16_t* refY, int refStride );
  void ( *recMeEnd )();
  void ( *recStageBegin )( const int16_t* patternRoi, int baseHor, int baseVer, int iFrac, int hadMode, int reduceTap, bool altHpel );
  void ( *recStageCost )( int i, uint64_t dist );
  void ( *recStageEnd )();
  void ( *recTu )( const void* transformUnit, int comp, const int16_t* resi, long stride, int w, int h, int trHor, int trVer, int bitDepth );
  void ( *recDmvrBegin )( const void* cu, const int16_t* ref0, int stride0, int fx0, int fy0, const int16_t* ref1, int stride1, int fx1, int fy1, int cuW, int cuH, int dx, int dy );
  void ( *recDmvrResult )( int num, int mvdX, int mvdY, uint64_t minCost );
  // ---- batched call sites of the residual loop and the merge pruning (hook bits 262144 / 524288)
  // residual loop: forward transforms (DCT-2) of the CU's component TUs in one device round trip; tuLookup hands a block's coefficients to TrQuant::xT when the residual it is asked
  // to transform is one of the prefetched blocks (compared sample by sample: a pure memo of the transform)
  void ( *tuPrefetch )( const int16_t* const resi[3], const int strides[3], const int widths[3], const int heights[3], int n, int bitDepth );
  bool ( *tuLookup )( const int16_t* resi, ptrdiff_t stride, int32_t* coef, unsigned width, unsigned height );
  // all merge candidates of a CU in one device call: predictions are compact w x h blocks, pitch = w * h samples from pred0 on; sad / satd may be null
  bool ( *mergeCosts )( const int16_t* org, int orgStride, const int16_t* const* preds, const int* predStrides, int n, int w, int h, int bitDepth, int hadMode, uint64_t* costs );
};
extern VvhipHooks g_vvhipHooks;

// run-time selection (called by the patched VVEncImpl::setSIMDExtension for a request that starts with "HIP"):
//   "HIP"            the production set: whole-picture stages on the device (MCTF search + filter, ALF statistics + filtering)
//   "HIP:<mask>"     explicit hook mask (decimal or 0x...), see vvenc_hip_install
// returns 0, or -1 when no MI355X context can be created (the encoder then reports the SIMD request as unsupported)
extern "C" int vvenc_hip_select( const char* spec );
// an encoder instance closes (VVEncImpl::uninit, before its buffers are freed): resident pictures are dropped, in-place host pins released.  Hooks stay installed.
extern "C" void vvenc_hip_release( void );
