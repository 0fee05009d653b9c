// TEST INFRASTRUCTURE — not product code.
//
// Thin extern "C" wrapper (ours) that is compiled against the *reference's own headers and
// objects* (see oracle/ref/Makefile) so that tests can call the reference's kernel tables —
// scalar row (enableOpt=false) and x86 SIMD row (enableOpt=true) — on seeded inputs, exactly
// like test/vvenc_unit_test/vvenc_unit_test.cpp builds its `ref`/`opt` object pairs.
// Used (a) to pin oracle/vvenc_oracle.c, (b) by tests/gen_golden.py to write tests/golden/*,
// (c) optionally as bench.py's cpu_baseline (kind "reference").
//
// Nothing here is shipped: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load oracle/_ref/libvvenc_ref.so.
//
// `#define private public` is a test-only trick to reach Quant::xQuant/xDeQuant/xNeedRdoq
// (CommonLib/Quant.h:143-151) and MCTF's private search schedule (CommonLib/MCTF.h:190-204).

#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <vector>
#include <atomic>
#include <sstream>
#include <iostream>
#include <fstream>
#include <string>
#include <list>
#include <map>
#include <array>
#include <mutex>
#include <thread>
#include <sched.h>
#include <pthread.h>
#include <condition_variable>
#include <functional>
#include <algorithm>
#include <memory>
#include <chrono>
#include <cmath>
#include <limits>
#include <cassert>
#include <cstdarg>
#include <iomanip>
#include <numeric>
#include <set>
#include <unordered_map>
#include <stdexcept>
#include <exception>
#include <utility>

#define private public
#define protected public
#include "CommonLib/CommonDef.h"
#include "CommonLib/Unit.h"
#include "CommonLib/Slice.h"
#include "CommonLib/CodingStructure.h"
#include "CommonLib/RdCost.h"
#include "CommonLib/TrQuant.h"
#include "CommonLib/TrQuant_EMT.h"
#include "CommonLib/Quant.h"
#include "CommonLib/MCTF.h"
#include "Utilities/NoMallocThreadPool.h"
#include "CommonLib/Rom.h"
#include "CommonLib/ContextModelling.h"
#include "CommonLib/Picture.h"
#include "CommonLib/InterpolationFilter.h"
#include "vvenc/vvencCfg.h"
#include "EncoderLib/EncCfg.h"
#include "CommonLib/AdaptiveLoopFilter.h"
#include "EncoderLib/EncAdaptiveLoopFilter.h"
#undef private
#undef protected

using namespace vvenc;

#define API extern "C" __attribute__((visibility("default")))

namespace {

struct RdPair { RdCost* rc[2]; };
RdPair& rdPair()
{
  static RdPair p = [] {
    RdPair q;
    q.rc[0] = new RdCost; q.rc[0]->create( false );
    q.rc[1] = new RdCost; q.rc[1]->create( true );
    return q;
  }();
  return p;
}

TCoeffOps& tcoeffOps( int simd )
{
  static TCoeffOps ops[2];
  static bool init = false;
  if( !init ) { ops[1].initTCoeffOps( true ); init = true; }
  return ops[simd ? 1 : 0];
}

void selectTCoeffOps( int simd )
{
  static std::atomic<int> cur( -1 );               // g_tCoeffOps is process-wide: rewrite it only when the selection changes (no cache-line ping-pong)
  if( cur.load( std::memory_order_relaxed ) != simd ) { g_tCoeffOps = tcoeffOps( simd ); cur.store( simd ); }
}

struct MctfPair { MCTF* m[2]; };
MctfPair& mctfPair()
{
  static MctfPair p = [] { MctfPair q; q.m[0] = new MCTF( false ); q.m[1] = new MCTF( true ); return q; }();
  return p;
}

Quant& quantObj()
{
  // Quant::Quant always installs the SIMD pointers when built with x86 SIMD (Quant.cpp:281-291);
  static Quant q( nullptr, false );
  return q;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// Distortion (RdCost tables)   dfBase: value of the DFunc enum base (DF_SSE, DF_SAD, DF_HAD, DF_HAD_fast, DF_HAD_2SAD)
// ---------------------------------------------------------------------------------------------
API int vvref_df( const char* name )
{
  if( !strcmp( name, "SSE" ) )      return DF_SSE;
  if( !strcmp( name, "SAD" ) )      return DF_SAD;
  if( !strcmp( name, "HAD" ) )      return DF_HAD;
  if( !strcmp( name, "HAD_fast" ) ) return DF_HAD_fast;
  if( !strcmp( name, "HAD_2SAD" ) ) return DF_HAD_2SAD;
  if( !strcmp( name, "SAD_WITH_MASK" ) ) return DF_SAD_WITH_MASK;
  return -1;
}

API uint64_t vvref_dist( int simd, int dfBase, const int16_t* org, int orgStride, const int16_t* cur, int curStride,
                         int w, int h, int bitDepth, int subShift )
{
  RdCost& rc = *rdPair().rc[simd ? 1 : 0];
  DistParam dp;
  dp.org = CPelBuf( org, orgStride, w, h );
  dp.cur = CPelBuf( cur, curStride, w, h );
  dp.bitDepth = bitDepth;
  dp.subShift = subShift;
  dp.compID   = COMP_Y;
  int idx = dfBase;
  if( dfBase != DF_HAD_2SAD && dfBase != DF_SAD_WITH_MASK ) idx += Log2( w );   // RdCost.cpp:177-181
  return rc.m_afpDistortFunc[0][idx]( dp );
}

API void vvref_sad_x5( int simd, const int16_t* org, int orgStride, const int16_t* cur, int curStride,
                       int w, int h, int bitDepth, int subShift, uint64_t* cost5, int calcCentre )
{
  RdCost& rc = *rdPair().rc[simd ? 1 : 0];
  DistParam dp;
  dp.org = CPelBuf( org, orgStride, w, h );
  dp.cur = CPelBuf( cur, curStride, w, h );
  dp.bitDepth = bitDepth;
  dp.subShift = subShift;
  dp.compID   = COMP_Y;
  Distortion c[5] = { 0, 0, 0, 0, 0 };
  rc.m_afpDistortFuncX5[Log2( w ) - 3]( dp, c, calcCentre != 0 );   // RdCost.cpp:252
  for( int i = 0; i < 5; i++ ) cost5[i] = c[i];
}

// DF_SAD_WITH_MASK (RdCost.cpp:2062-2093, SIMD row RdCostX86.h:2629).  The SIMD row derives the row advance from maskStride alone and
// mirrors the mask for stepX == -1, so it equals the scalar row only for the two parameterisations setDistParamGeo produces:
// (stepX 1, maskStride2 -w) and (stepX -1, maskStride2 +w).
API uint64_t vvref_sad_mask( int simd, const int16_t* org, int orgStride, const int16_t* cur, int curStride,
                             const int16_t* mask, int maskStride, int stepX, int maskStride2, int w, int h, int bitDepth, int subShift )
{
  RdCost& rc = *rdPair().rc[simd ? 1 : 0];
  DistParam dp;
  dp.org = CPelBuf( org, orgStride, w, h );
  dp.cur = CPelBuf( cur, curStride, w, h );
  dp.bitDepth = bitDepth;
  dp.subShift = subShift;
  dp.compID   = COMP_Y;
  dp.mask = mask; dp.maskStride = maskStride; dp.stepX = stepX; dp.maskStride2 = maskStride2;
  return rc.m_afpDistortFunc[0][DF_SAD_WITH_MASK]( dp );
}

API uint64_t vvref_fix_weighted_sse( int simd, const int16_t* org, int orgStride, const int16_t* cur, int curStride,
                                     int w, int h, int bitDepth, uint32_t weight )
{
  RdCost& rc = *rdPair().rc[simd ? 1 : 0];
  DistParam dp;
  dp.org = CPelBuf( org, orgStride, w, h );
  dp.cur = CPelBuf( cur, curStride, w, h );
  dp.bitDepth = bitDepth;
  dp.compID   = COMP_Y;
  return rc.m_fxdWtdPredPtr( dp, weight );
}

// ---------------------------------------------------------------------------------------------
// Transform matrices and 1-D / 2-D transforms
// ---------------------------------------------------------------------------------------------
static const TMatrixCoeff* trMatrix( int trType, int log2N )
{
  switch( trType )
  {
  case DCT2:
    switch( log2N ) {
      case 1: return g_trCoreDCT2P2 [TRANSFORM_FORWARD][0];
      case 2: return g_trCoreDCT2P4 [TRANSFORM_FORWARD][0];
      case 3: return g_trCoreDCT2P8 [TRANSFORM_FORWARD][0];
      case 4: return g_trCoreDCT2P16[TRANSFORM_FORWARD][0];
      case 5: return g_trCoreDCT2P32[TRANSFORM_FORWARD][0];
      case 6: return g_trCoreDCT2P64[TRANSFORM_FORWARD][0];
    } break;
  case DCT8:
    switch( log2N ) {
      case 2: return g_trCoreDCT8P4 [TRANSFORM_FORWARD][0];
      case 3: return g_trCoreDCT8P8 [TRANSFORM_FORWARD][0];
      case 4: return g_trCoreDCT8P16[TRANSFORM_FORWARD][0];
      case 5: return g_trCoreDCT8P32[TRANSFORM_FORWARD][0];
    } break;
  case DST7:
    switch( log2N ) {
      case 2: return g_trCoreDST7P4 [TRANSFORM_FORWARD][0];
      case 3: return g_trCoreDST7P8 [TRANSFORM_FORWARD][0];
      case 4: return g_trCoreDST7P16[TRANSFORM_FORWARD][0];
      case 5: return g_trCoreDST7P32[TRANSFORM_FORWARD][0];
    } break;
  }
  return nullptr;
}

API int vvref_tr_type( const char* name )
{
  if( !strcmp( name, "DCT2" ) ) return DCT2;
  if( !strcmp( name, "DCT8" ) ) return DCT8;
  if( !strcmp( name, "DST7" ) ) return DST7;
  return -1;
}

API int vvref_tr_matrix( int trType, int log2N, int16_t* out )
{
  const TMatrixCoeff* m = trMatrix( trType, log2N );
  if( !m ) return -1;
  const int N = 1 << log2N;
  for( int i = 0; i < N * N; i++ ) out[i] = m[i];
  return 0;
}

typedef void FwdFn( const TCoeff*, TCoeff*, int, int, int, int );
typedef void InvFn( const TCoeff*, TCoeff*, int, int, int, int, const TCoeff, const TCoeff );

static FwdFn* fwdFn( int trType, int log2N )
{
  // same layout as fastFwdTrans[NUM_TRANS_TYPE][g_numTransformMatrixSizes] (TrQuant.cpp:76-81), which has internal linkage
  static FwdFn* const t[3][6] = {
    { fastForwardDCT2_B2, fastForwardDCT2_B4, fastForwardDCT2_B8, fastForwardDCT2_B16, fastForwardDCT2_B32, fastForwardDCT2_B64 },
    { nullptr,            fastForwardDCT8_B4, fastForwardDCT8_B8, fastForwardDCT8_B16, fastForwardDCT8_B32, nullptr },
    { nullptr,            fastForwardDST7_B4, fastForwardDST7_B8, fastForwardDST7_B16, fastForwardDST7_B32, nullptr } };
  return t[trType][log2N - 1];
}
static InvFn* invFn( int trType, int log2N )
{
  static InvFn* const t[3][6] = {
    { fastInverseDCT2_B2, fastInverseDCT2_B4, fastInverseDCT2_B8, fastInverseDCT2_B16, fastInverseDCT2_B32, fastInverseDCT2_B64 },
    { nullptr,            fastInverseDCT8_B4, fastInverseDCT8_B8, fastInverseDCT8_B16, fastInverseDCT8_B32, nullptr },
    { nullptr,            fastInverseDST7_B4, fastInverseDST7_B8, fastInverseDST7_B16, fastInverseDST7_B32, nullptr } };
  return t[trType][log2N - 1];
}

API int vvref_fwd_1d( int simd, int trType, int log2N, const int32_t* src, int32_t* dst, int shift, int line, int skipLine, int skipLine2 )
{
  FwdFn* f = fwdFn( trType, log2N );
  if( !f ) return -1;
  selectTCoeffOps( simd );
  f( src, dst, shift, line, skipLine, skipLine2 );
  return 0;
}

API int vvref_inv_1d( int simd, int trType, int log2N, const int32_t* src, int32_t* dst, int shift, int line, int skipLine, int skipLine2, int32_t clipMin, int32_t clipMax )
{
  InvFn* f = invFn( trType, log2N );
  if( !f ) return -1;
  selectTCoeffOps( simd );
  f( src, dst, shift, line, skipLine, skipLine2, clipMin, clipMax );
  return 0;
}

// The g_tCoeffOps slots called directly (TrQuant_EMT.h:63-91): scalar row = the *Core functions, SIMD row = TrafoX86.h.
API void vvref_fast_fwd_core( int simd, int log2N, const int16_t* tc, const int32_t* src, int32_t* dst, unsigned line, unsigned reducedLine, unsigned cutoff, int shift )
{ tcoeffOps( simd ).fastFwdCore_2D[log2N - 2]( tc, src, dst, line, reducedLine, cutoff, shift ); }
API void vvref_fast_inv_core( int simd, int log2N, const int16_t* it, const int32_t* src, int32_t* dst, unsigned lines, unsigned reducedLines, unsigned rows )
{ tcoeffOps( simd ).fastInvCore[log2N - 2]( it, src, dst, lines, reducedLines, rows ); }
API void vvref_round_clip( int simd, int32_t* dst, unsigned w, unsigned h, unsigned stride, int32_t mn, int32_t mx, int32_t round, int32_t shift )
{ if( w & 7 ) tcoeffOps( simd ).roundClip4( dst, w, h, stride, mn, mx, round, shift ); else tcoeffOps( simd ).roundClip8( dst, w, h, stride, mn, mx, round, shift ); }
API void vvref_cpy_resi( int simd, const int32_t* src, int16_t* dst, ptrdiff_t stride, unsigned w, unsigned h )
{ if( w & 7 ) tcoeffOps( simd ).cpyResi4( src, dst, stride, w, h ); else tcoeffOps( simd ).cpyResi8( src, dst, stride, w, h ); }
API void vvref_cpy_coeff( int simd, const int16_t* src, ptrdiff_t stride, int32_t* dst, unsigned w, unsigned h )
{ if( w & 7 ) tcoeffOps( simd ).cpyCoeff4( src, stride, dst, w, h ); else tcoeffOps( simd ).cpyCoeff8( src, stride, dst, w, h ); }

// The wiring of TrQuant::xT (TrQuant.cpp:481-564) around the reference's own 1-D functions and cpyCoeff ops.
// (xT itself needs a TransformUnit/CodingStructure; the 1-D cores, zero-out and copies below ARE the reference's.)
API int vvref_xT( int simd, const int16_t* resi, int resiStride, int32_t* coef, int width, int height,
                  int trTypeHor, int trTypeVer, int bitDepth )
{
  const int maxLog2TrDynamicRange = 15;
  const int TRANSFORM_MATRIX_SHIFT = g_transformMatrixShift[TRANSFORM_FORWARD];
  int skipWidth  = ( trTypeHor != DCT2 && width  == 32 ) ? 16 : width  > JVET_C0024_ZERO_OUT_TH ? width  - JVET_C0024_ZERO_OUT_TH : 0;
  int skipHeight = ( trTypeVer != DCT2 && height == 32 ) ? 16 : height > JVET_C0024_ZERO_OUT_TH ? height - JVET_C0024_ZERO_OUT_TH : 0;
  selectTCoeffOps( simd );
  struct Scratch { TCoeff* p[3]; Scratch() { for( auto& q : p ) q = ( TCoeff* ) xMalloc( TCoeff, MAX_TB_SIZEY * MAX_TB_SIZEY ); } ~Scratch() { for( auto& q : p ) xFree( q ); } };
  static thread_local Scratch sc;
  TCoeff* block = sc.p[0]; TCoeff* tmp = sc.p[1]; TCoeff* dst = sc.p[2];
  if( width & 3 )
  {
    for( int y = 0; y < height; y++ ) for( int x = 0; x < width; x++ ) block[y * width + x] = resi[y * resiStride + x];
  }
  else if( width & 7 ) g_tCoeffOps.cpyCoeff4( resi, resiStride, block, width, height );
  else                 g_tCoeffOps.cpyCoeff8( resi, resiStride, block, width, height );

  int rc = 0;
  if( width > 1 && height > 1 )
  {
    const int shift_1st = ( Log2( width ) + bitDepth + TRANSFORM_MATRIX_SHIFT ) - maxLog2TrDynamicRange;
    const int shift_2nd = Log2( height ) + TRANSFORM_MATRIX_SHIFT;
    FwdFn* fh = fwdFn( trTypeHor, Log2( width ) ), *fv = fwdFn( trTypeVer, Log2( height ) );
    if( !fh || !fv || shift_1st < 0 ) rc = -1;
    else
    {
      fh( block, tmp, shift_1st, height, 0, skipWidth );
      fv( tmp, dst, shift_2nd, width, skipWidth, skipHeight );
    }
  }
  else rc = -2;
  if( !rc ) memcpy( coef, dst, sizeof( TCoeff ) * width * height );
  return rc;
}

// The wiring of TrQuant::xIT (TrQuant.cpp:567-655).
API int vvref_xIT( int simd, const int32_t* coef, int16_t* resi, int resiStride, int width, int height,
                   int trTypeHor, int trTypeVer, int bitDepth )
{
  const int maxLog2TrDynamicRange = 15;
  const int TRANSFORM_MATRIX_SHIFT = g_transformMatrixShift[TRANSFORM_INVERSE];
  const TCoeff clipMinimum = -( 1 << maxLog2TrDynamicRange );
  const TCoeff clipMaximum =  ( 1 << maxLog2TrDynamicRange ) - 1;
  int skipWidth  = ( trTypeHor != DCT2 && width  == 32 ) ? 16 : width  > JVET_C0024_ZERO_OUT_TH ? width  - JVET_C0024_ZERO_OUT_TH : 0;
  int skipHeight = ( trTypeVer != DCT2 && height == 32 ) ? 16 : height > JVET_C0024_ZERO_OUT_TH ? height - JVET_C0024_ZERO_OUT_TH : 0;
  selectTCoeffOps( simd );
  struct Scratch { TCoeff* p[3]; Scratch() { for( auto& q : p ) q = ( TCoeff* ) xMalloc( TCoeff, MAX_TB_SIZEY * MAX_TB_SIZEY ); } ~Scratch() { for( auto& q : p ) xFree( q ); } };
  static thread_local Scratch sc;
  TCoeff* src = sc.p[0]; TCoeff* block = sc.p[1]; TCoeff* tmp = sc.p[2];
  memcpy( src, coef, sizeof( TCoeff ) * width * height );
  int rc = 0;
  if( width > 1 && height > 1 )
  {
    const int shift_1st = TRANSFORM_MATRIX_SHIFT + 1;
    const int shift_2nd = ( TRANSFORM_MATRIX_SHIFT + maxLog2TrDynamicRange - 1 ) - bitDepth;
    InvFn* fv = invFn( trTypeVer, Log2( height ) ), *fh = invFn( trTypeHor, Log2( width ) );
    if( !fh || !fv ) rc = -1;
    else
    {
      fv( src, tmp, shift_1st, width, skipWidth, skipHeight, clipMinimum, clipMaximum );
      fh( tmp, block, shift_2nd, height, 0, skipWidth, clipMinimum, clipMaximum );
    }
  }
  else rc = -2;
  if( !rc )
  {
    if( width & 3 )
    {
      const TCoeff* b = block;
      for( int y = 0; y < height; y++ ) for( int x = 0; x < width; x++ ) resi[y * resiStride + x] = ( Pel ) *b++;
    }
    else if( width & 7 ) g_tCoeffOps.cpyResi4( block, resi, resiStride, width, height );
    else                 g_tCoeffOps.cpyResi8( block, resi, resiStride, width, height );
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------
// Scalar quantisation
// ---------------------------------------------------------------------------------------------
API int vvref_scan_order( int log2w, int log2h, uint32_t* out )
{
  const ScanElement* scan = getScanOrder( SCAN_GROUPED_4x4, log2w, log2h );
  const int n = 1 << ( log2w + log2h );
  for( int i = 0; i < n; i++ ) out[i] = scan[i].idx;
  return n;
}

API void vvref_quant_scales( int* q12, int* iq12 )
{
  for( int i = 0; i < 2; i++ ) for( int j = 0; j < 6; j++ ) { q12[i * 6 + j] = g_quantScales[i][j]; iq12[i * 6 + j] = g_invQuantScales[i][j]; }
}

API void vvref_dequant_core( int simd, int maxX, int maxY, int scale, const int16_t* q, size_t qStride, int32_t* coef,
                             int rightShift, int inputMaximum, int32_t transformMaximum )
{
  // scalar = DeQuantCore is file-static in Quant.cpp; the only exported path is the pointer installed by the ctor (SIMD).
  // simd==0 is served by a fresh Quant whose pointer we have NOT let initQuantX86 overwrite: not reachable -> both use ctor pointer.
  ( void ) simd;
  quantObj().xDeQuant( maxX, maxY, scale, q, qStride, coef, rightShift, inputMaximum, transformMaximum );
}

API int vvref_need_rdoq_core( int simd, const int32_t* coef, size_t num, int quantCoeff, int64_t offset, int shift )
{
  ( void ) simd;
  return quantObj().xNeedRdoq( coef, num, quantCoeff, offset, shift ) ? 1 : 0;
}

// QuantCore through the pointer Quant::xQuant (SIMD row when built with x86 SIMD).  A minimal TransformUnit is faked:
// QuantCore only touches tu.blocks[compID], tu.cu->lfnstIdx and CoeffCodingContext( tu, ... ) which reads
// tu.block(), tu.cs->sps->getMaxLog2TrDynamicRange() (Quant.cpp:132-230, ContextModelling.cpp:61-66).
API int vvref_quant_core_lfnst( const int32_t* coef, int16_t* qcoef, int32_t* deltaU, int width, int height,
                                int quantCoeff, int iQBits, int64_t iAdd, int signHiding, int thrVal, int lfnstIdx,
                                int32_t* absSumOut, int* lastScanPosOut );
API int vvref_quant_core( const int32_t* coef, int16_t* qcoef, int32_t* deltaU, int width, int height,
                          int quantCoeff, int iQBits, int64_t iAdd, int signHiding, int thrVal,
                          int32_t* absSumOut, int* lastScanPosOut )
{
  return vvref_quant_core_lfnst( coef, qcoef, deltaU, width, height, quantCoeff, iQBits, iAdd, signHiding, thrVal, 0, absSumOut, lastScanPosOut );
}
// the same with CodingUnit::lfnstIdx set (QuantCore's first-coefficient-group rule, Quant.cpp:149-159)
API int vvref_quant_core_lfnst( const int32_t* coef, int16_t* qcoef, int32_t* deltaU, int width, int height,
                                int quantCoeff, int iQBits, int64_t iAdd, int signHiding, int thrVal, int lfnstIdx,
                                int32_t* absSumOut, int* lastScanPosOut )
{
  static SPS* sps = new SPS;
  static void* csMem = calloc( 1, sizeof( CodingStructure ) );
  CodingStructure* cs = reinterpret_cast<CodingStructure*>( csMem );
  cs->sps = sps;
  CodingUnit cu;
  memset( ( void* ) &cu, 0, sizeof( cu ) );
  cu.lfnstIdx = ( uint8_t ) lfnstIdx;
  TransformUnit tu( CHROMA_420, Area( 0, 0, width, height ) );
  tu.cu = &cu;
  tu.cs = cs;
  tu.mtsIdx[COMP_Y] = 0;
  CCoeffBuf src( coef, width, width, height );
  CoeffSigBuf dst( qcoef, width, width, height );
  TCoeff absSum = 0; int lastScanPos = -1;
  quantObj().xQuant( tu, COMP_Y, src, dst, absSum, lastScanPos, deltaU, quantCoeff, iQBits, iAdd,
                     -( 1 << 15 ), ( 1 << 15 ) - 1, signHiding != 0, thrVal );
  *absSumOut = absSum; *lastScanPosOut = lastScanPos;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// MCTF
// ---------------------------------------------------------------------------------------------
API int vvref_mctf_err_int( int simd, const int16_t* org, ptrdiff_t orgStride, const int16_t* buf, ptrdiff_t bufStride, int w, int h, int besterror )
{
  return mctfPair().m[simd ? 1 : 0]->m_motionErrorLumaInt8( org, orgStride, buf, bufStride, w, h, besterror );
}

// tap4 = 1 -> m_motionErrorLumaFrac8[1] with MCTF::m_interpolationFilter4 rows, else [0] with m_interpolationFilter8 rows
API int vvref_mctf_err_frac( int simd, int tap4, const int16_t* org, ptrdiff_t orgStride, const int16_t* buf, ptrdiff_t bufStride,
                             int w, int h, int fx, int fy, int bitDepth, int besterror )
{
  MCTF* m = mctfPair().m[simd ? 1 : 0];
  const int16_t* xf = tap4 ? MCTF::m_interpolationFilter4[fx] : MCTF::m_interpolationFilter8[fx];
  const int16_t* yf = tap4 ? MCTF::m_interpolationFilter4[fy] : MCTF::m_interpolationFilter8[fy];
  return m->m_motionErrorLumaFrac8[tap4 ? 1 : 0]( org, orgStride, buf, bufStride, w, h, xf, yf, bitDepth, besterror );
}

API void vvref_mctf_filters( int16_t* f8 /*16*8*/, int16_t* f4 /*16*4*/ )
{
  memcpy( f8, MCTF::m_interpolationFilter8, sizeof( int16_t ) * 16 * 8 );
  memcpy( f4, MCTF::m_interpolationFilter4, sizeof( int16_t ) * 16 * 4 );
}

API double vvref_mctf_calc_var( int simd, const int16_t* org, ptrdiff_t stride, int w, int h )
{
  return mctfPair().m[simd ? 1 : 0]->m_calcVar( org, stride, w, h );
}

namespace {
struct MvOut { int32_t x, y, error, rmsme; double overlap; };

void fillPlane( PelStorage& ps, const int16_t* src, int w, int h )
{
  ps.create( CHROMA_400, Area( 0, 0, w, h ), 0, MCTF_PADDING );
  PelBuf y = ps.Y();
  for( int r = 0; r < h; r++ ) memcpy( y.buf + r * y.stride, src + ( size_t ) r * w, sizeof( int16_t ) * w );
  y.extendBorderPel( MCTF_PADDING, MCTF_PADDING );   // MCTF::initPicture, MCTF.cpp:608-612
}
void dumpMvs( const Array2D<MotionVector>& a, MvOut* out )
{
  for( int y = 0; y < a.h(); y++ ) for( int x = 0; x < a.w(); x++ )
  {
    const MotionVector& m = a.get( x, y );
    MvOut& o = out[y * a.w() + x];
    o.x = m.x; o.y = m.y; o.error = m.error; o.rmsme = m.rmsme; o.overlap = m.overlap;
  }
}
} // namespace

// Runs MCTF::subsampleLuma once; `out` must hold (w/2)*(h/2) samples (visible area only).
API void vvref_mctf_subsample( const int16_t* src, int w, int h, int16_t* out )
{
  MCTF m( false );
  PelStorage in, o;
  fillPlane( in, src, w, h );
  m.subsampleLuma( in, o );
  CPelBuf y = o.Y();
  for( int r = 0; r < ( int ) y.height; r++ ) memcpy( out + ( size_t ) r * y.width, y.buf + r * y.stride, sizeof( int16_t ) * y.width );
}

// The hierarchical search of MCTF::motionEstimationMCTF (MCTF.cpp:666-724) for ONE (current, reference) pair of luma planes,
// single-threaded (m_threadPool == nullptr -> MCTF.cpp:1388-1396, same results as the threaded wavefront).
// levelOut[k] (k=0: 1/8 if addLevel, then 1/4, 1/2, 1/1@2*unit, final@unit) receive MvOut arrays; dims are written to levelDims[2*k+{0,1}].
API int vvref_mctf_me( int simd, const int16_t* orgLuma, const int16_t* refLuma, int width, int height, int bitDepth,
                       int unitSize, int mctfSpeed, int addLevel, MvOut** levelOut, int* levelDims )
{
  MCTF m( simd != 0 );
  static VVEncCfg cfg;
  vvenc_init_default( &cfg, width, height, 30, 0, 32, VVENC_FASTER );
  cfg.m_internalBitDepth[0] = bitDepth;
  cfg.m_internalBitDepth[1] = bitDepth;
  m.m_encCfg = &cfg;
  m.m_threadPool = nullptr;
  m.m_area = Area( 0, 0, width, height );
  m.m_lowResFltSearch = mctfSpeed > 0;                                        // MCTF.cpp:598
  m.m_searchPttrn     = mctfSpeed > 0 ? ( mctfSpeed >= 3 ? 2 : 1 ) : 0;       // MCTF.cpp:599
  m.m_mctfUnitSize    = unitSize;

  PelStorage org, ref;
  fillPlane( org, orgLuma, width, height );
  fillPlane( ref, refLuma, width, height );

  PelStorage o2, o4, o8, b2, b4, b8;
  m.subsampleLuma( org, o2 ); m.subsampleLuma( o2, o4 );
  m.subsampleLuma( ref, b2 ); m.subsampleLuma( b2, b4 );
  if( addLevel ) { m.subsampleLuma( o4, o8 ); m.subsampleLuma( b4, b8 ); }

  const int wInBlks = ( width + unitSize - 1 ) / unitSize, hInBlks = ( height + unitSize - 1 ) / unitSize;
  Array2D<MotionVector> mv_m( width / ( unitSize * 16 ) + 1, height / ( unitSize * 16 ) + 1 );
  Array2D<MotionVector> mv_0( width / ( unitSize * 8 ) + 1, height / ( unitSize * 8 ) + 1 );
  Array2D<MotionVector> mv_1( width / ( unitSize * 4 ) + 1, height / ( unitSize * 4 ) + 1 );
  Array2D<MotionVector> mv_2( width / ( unitSize * 2 ) + 1, height / ( unitSize * 2 ) + 1 );
  Array2D<MotionVector> mvs; mvs.allocate( wInBlks, hInBlks );

  int k = 0;
  if( addLevel )
  {
    m.motionEstimationLuma( mv_m, o8, b8, 2 * unitSize );
    m.motionEstimationLuma( mv_0, o4, b4, 2 * unitSize, &mv_m, 2 );
    levelDims[0] = mv_m.w(); levelDims[1] = mv_m.h(); if( levelOut[0] ) dumpMvs( mv_m, levelOut[0] );
  }
  else
  {
    m.motionEstimationLuma( mv_0, o4, b4, 2 * unitSize );
    levelDims[0] = 0; levelDims[1] = 0;
  }
  k = 1;
  levelDims[2] = mv_0.w(); levelDims[3] = mv_0.h(); if( levelOut[1] ) dumpMvs( mv_0, levelOut[1] );
  m.motionEstimationLuma( mv_1, o2, b2, 2 * unitSize, &mv_0, 2 );
  levelDims[4] = mv_1.w(); levelDims[5] = mv_1.h(); if( levelOut[2] ) dumpMvs( mv_1, levelOut[2] );
  m.motionEstimationLuma( mv_2, org, ref, 2 * unitSize, &mv_1, 2 );
  levelDims[6] = mv_2.w(); levelDims[7] = mv_2.h(); if( levelOut[3] ) dumpMvs( mv_2, levelOut[3] );
  m.motionEstimationLuma( mvs, org, ref, unitSize, &mv_2, 1, true );
  levelDims[8] = mvs.w(); levelDims[9] = mvs.h(); if( levelOut[4] ) dumpMvs( mvs, levelOut[4] );
  ( void ) k;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// MCTF apply side — table entries m_applyFrac / m_applyPlanarCorrection / m_applyBlock (MCTF.h:167-170) and the
// whole-picture MCTF::bilateralFilter (MCTF.cpp:1489-1552) on 4:2:0 planes.
// ---------------------------------------------------------------------------------------------
API void vvref_mctf_apply_frac( int simd, int chroma, int tap4, const int16_t* org, ptrdiff_t os, int16_t* dst, ptrdiff_t ds, int w, int h, int fx, int fy, int bitDepth )
{
  MCTF& m = *mctfPair().m[simd ? 1 : 0];
  if( tap4 ) m.m_applyFrac[chroma ? 1 : 0][1]( org, os, dst, ds, w, h, MCTF::m_interpolationFilter4[fx], MCTF::m_interpolationFilter4[fy], bitDepth );
  else       m.m_applyFrac[chroma ? 1 : 0][0]( org, os, dst, ds, w, h, MCTF::m_interpolationFilter8[fx], MCTF::m_interpolationFilter8[fy], bitDepth );
}

API void vvref_mctf_planar_correction( int simd, const int16_t* ref, ptrdiff_t rs, int16_t* dst, ptrdiff_t ds, int w, int h, int bitDepth, int motionError )
{
  ClpRng clp; clp.bd = bitDepth;
  mctfPair().m[simd ? 1 : 0]->m_applyPlanarCorrection( ref, rs, dst, ds, w, h, clp, ( uint16_t ) motionError );
}

API void vvref_mctf_apply_block( int simd, const int16_t* src, ptrdiff_t ss, int16_t* dst, ptrdiff_t ds, int w, int h, int bitDepth, const int16_t* const* corrected, int numRefs,
                                 const int* verror, const double* refStrengths, double weightScaling, double sigmaSq )
{
  ClpRng clp; clp.bd = bitDepth;
  const CPelBuf srcBuf( src, ss, w, h );
  PelBuf dstBuf( dst, ds, w, h );
  const Pel* corr[2 * VVENC_MCTF_RANGE] = { nullptr, };
  for( int i = 0; i < numRefs; i++ ) corr[i] = corrected[i];
  mctfPair().m[simd ? 1 : 0]->m_applyBlock( srcBuf, dstBuf, CompArea( COMP_Y, CHROMA_400, Area( 0, 0, w, h ) ), clp, corr, numRefs, verror, refStrengths, weightScaling, sigmaSq );
}

namespace {
void fillYuv( PelStorage& ps, const int16_t* const yuv[3], int w, int h )
{
  ps.create( CHROMA_420, Area( 0, 0, w, h ), 0, MCTF_PADDING );
  for( int c = 0; c < 3; c++ )
  {
    PelBuf b = ps.bufs[c];
    for( int r = 0; r < ( int ) b.height; r++ ) memcpy( b.buf + r * b.stride, yuv[c] + ( size_t ) r * b.width, sizeof( int16_t ) * b.width );
    b.extendBorderPel( MCTF_PADDING >> ( c ? 1 : 0 ), MCTF_PADDING >> ( c ? 1 : 0 ) );
  }
}
}

// org / refs[i]: 3 compact planes each (Y w x h, U and V w/2 x h/2); mvs[i]: final-level motion field of reference i (ceil(w/unit) x ceil(h/unit) MvOut);
// refIndex[i] = TemporalFilterSourcePicInfo::index (|POC offset| - 1, row RA of m_refStrengths when picReordering).  out: 3 compact planes.
API int vvref_mctf_bilateral( int simd, int width, int height, int bitDepth, int qp, int unitSize, int lowResFltApply, int picReordering, const int16_t* const* org,
                              int numRefs, const int16_t* const* refs /* 3 per reference */, const MvOut* const* mvs, const int* refIndex, double overallStrength,
                              int16_t* const* out )
{
  MCTF m( simd != 0 );
  static VVEncCfg cfg;
  vvenc_init_default( &cfg, width, height, 30, 0, qp, VVENC_FASTER );
  cfg.m_internalBitDepth[0] = cfg.m_internalBitDepth[1] = bitDepth;
  cfg.m_internChromaFormat = VVENC_CHROMA_420;
  cfg.m_QP = qp;
  cfg.m_picReordering = picReordering != 0;
  m.m_encCfg = &cfg;
  m.m_threadPool = nullptr;
  m.m_area = Area( 0, 0, width, height );
  m.m_mctfUnitSize = unitSize;
  m.m_lowResFltApply = lowResFltApply != 0;
  PelStorage orgPic, newPic;
  fillYuv( orgPic, org, width, height );
  newPic.create( CHROMA_420, Area( 0, 0, width, height ), 0, MCTF_PADDING );
  std::deque<TemporalFilterSourcePicInfo> infos;
  const int wB = ( width + unitSize - 1 ) / unitSize, hB = ( height + unitSize - 1 ) / unitSize;
  for( int i = 0; i < numRefs; i++ )
  {
    infos.emplace_back();
    TemporalFilterSourcePicInfo& s = infos.back();
    fillYuv( s.picBuffer, refs + 3 * i, width, height );
    s.mvs.allocate( wB, hB );
    for( int y = 0; y < hB; y++ ) for( int x = 0; x < wB; x++ )
    {
      MotionVector& d = s.mvs.get( x, y ); const MvOut& v = mvs[i][y * wB + x];
      d.x = v.x; d.y = v.y; d.error = v.error; d.rmsme = ( uint16_t ) v.rmsme; d.overlap = v.overlap;
    }
    s.index = refIndex[i];
  }
  m.bilateralFilter( orgPic, infos, newPic, overallStrength );
  for( int c = 0; c < 3; c++ )
  {
    CPelBuf b = newPic.bufs[c];
    for( int r = 0; r < ( int ) b.height; r++ ) memcpy( out[c] + ( size_t ) r * b.width, b.buf + r * b.stride, sizeof( int16_t ) * b.width );
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// CPU baseline of north-star leg C (bench.py cpu_baseline.mctf, kind "reference"): what MCTF::filter does for the filtered pictures of a GOP cycle (MCTF.cpp:726-870 without
// the adaptive extra references): per picture its pyramid, motionEstimationMCTF of every reference (:666-724: the reference picture's pyramid + the five levels) and
// bilateralFilter (:1489-1552) on 4:2:0 planes.  threads == 0: everything on the calling thread.  threads > 0: the (picture, reference) motion estimations are independent
// jobs on `threads` host threads (each runs the reference's single-threaded row loop, :1388-1396 — its pool's row tasks busy-wait on the row above, which loses on a shared
// host), the filter runs on the reference's OWN thread pool (NoMallocThreadPool: block rows are its tasks, :1504-1543).  Planes are allocated and filled before the clocks
// start (the encoder holds them already).  secs[0] = wall time of the pyramids + all motion estimations, secs[1] = of the filters.
// cur: 3 planes per picture; refs / refIndex / finals: per reference, pictures concatenated (numRefs[p] each); out: 3 planes per picture.  finals / out entries may be null.
// ---------------------------------------------------------------------------------------------
API int vvref_mctf_cycle_timed( int simd, int width, int height, int bitDepth, int qp, int unitSize, int mctfSpeed, int addLevel, int threads, int numPics,
                                const int16_t* const* cur, const int* numRefs, const int16_t* const* refs, const int* refIndex, const double* overallStrength,
                                MvOut* const* finals, int16_t* const* out, double* secs )
{
  MCTF m( simd != 0 );
  VVEncCfg cfg;
  vvenc_init_default( &cfg, width, height, 30, 0, qp, VVENC_FASTER );
  cfg.m_internalBitDepth[0] = cfg.m_internalBitDepth[1] = bitDepth;
  cfg.m_internChromaFormat = VVENC_CHROMA_420;
  cfg.m_QP = qp;
  cfg.m_picReordering = true;
  m.m_encCfg = &cfg;
  m.m_threadPool = nullptr;
  m.m_area = Area( 0, 0, width, height );
  m.m_lowResFltSearch = mctfSpeed > 0;                                        // MCTF.cpp:598
  m.m_searchPttrn     = mctfSpeed > 0 ? ( mctfSpeed >= 3 ? 2 : 1 ) : 0;       // MCTF.cpp:599
  m.m_mctfUnitSize    = unitSize;
  m.m_lowResFltApply  = false;                                                // MCTF.h:190 (never set by the encoder)

  struct Pic { PelStorage org, newPic, o2, o4, o8; std::deque<TemporalFilterSourcePicInfo> infos; int firstRef; };
  std::vector<Pic> pics( numPics );
  const int wB = ( width + unitSize - 1 ) / unitSize, hB = ( height + unitSize - 1 ) / unitSize;
  std::vector<std::pair<int, int>> jobs;                                       // (picture, reference of the picture)
  int rTotal = 0;
  for( int p = 0; p < numPics; p++ )
  {
    Pic& P = pics[p];
    P.firstRef = rTotal;
    fillYuv( P.org, cur + 3 * p, width, height );
    P.newPic.create( CHROMA_420, Area( 0, 0, width, height ), 0, MCTF_PADDING );
    for( int i = 0; i < numRefs[p]; i++, rTotal++ )
    {
      P.infos.emplace_back();
      fillYuv( P.infos.back().picBuffer, refs + 3 * rTotal, width, height );
      P.infos.back().mvs.allocate( wB, hB );
      P.infos.back().index = refIndex[rTotal];
      jobs.emplace_back( p, i );
    }
  }
  auto pyramid = [&]( int p ) { Pic& P = pics[p]; m.subsampleLuma( P.org, P.o2 ); m.subsampleLuma( P.o2, P.o4 ); if( addLevel ) m.subsampleLuma( P.o4, P.o8 ); };
  auto estimate = [&]( int j )                                                // MCTF.cpp:666-707 for one (picture, reference)
  {
    Pic& P = pics[jobs[j].first];
    TemporalFilterSourcePicInfo& s = P.infos[jobs[j].second];
    PelStorage b2, b4, b8;
    m.subsampleLuma( s.picBuffer, b2 ); m.subsampleLuma( b2, b4 );
    if( addLevel ) m.subsampleLuma( b4, b8 );
    Array2D<MotionVector> mv_m( width / ( unitSize * 16 ) + 1, height / ( unitSize * 16 ) + 1 );
    Array2D<MotionVector> mv_0( width / ( unitSize * 8 ) + 1, height / ( unitSize * 8 ) + 1 );
    Array2D<MotionVector> mv_1( width / ( unitSize * 4 ) + 1, height / ( unitSize * 4 ) + 1 );
    Array2D<MotionVector> mv_2( width / ( unitSize * 2 ) + 1, height / ( unitSize * 2 ) + 1 );
    if( addLevel )
    {
      m.motionEstimationLuma( mv_m, P.o8, b8, 2 * unitSize );
      m.motionEstimationLuma( mv_0, P.o4, b4, 2 * unitSize, &mv_m, 2 );
    }
    else m.motionEstimationLuma( mv_0, P.o4, b4, 2 * unitSize );
    m.motionEstimationLuma( mv_1, P.o2, b2, 2 * unitSize, &mv_0, 2 );
    m.motionEstimationLuma( mv_2, P.org, s.picBuffer, 2 * unitSize, &mv_1, 2 );
    m.motionEstimationLuma( s.mvs, P.org, s.picBuffer, unitSize, &mv_2, 1, true );
  };
  auto spread = [&]( int n, const std::function<void( int )>& f )
  {
    if( threads <= 0 ) { for( int i = 0; i < n; i++ ) f( i ); return; }
    std::atomic<int> next( 0 );
    std::vector<std::thread> th;
    for( int t = 0; t < std::min( threads, n ); t++ ) th.emplace_back( [&]{ for( int i; ( i = next.fetch_add( 1 ) ) < n; ) f( i ); } );
    for( auto& t : th ) t.join();
  };
  const auto t0 = std::chrono::steady_clock::now();
  spread( numPics, pyramid );
  spread( ( int ) jobs.size(), estimate );
  const auto t1 = std::chrono::steady_clock::now();
  {
    std::unique_ptr<NoMallocThreadPool> pool;
    if( threads > 0 ) pool.reset( new NoMallocThreadPool( threads, "mctf", &cfg ) );
    m.m_threadPool = pool.get();
    for( int p = 0; p < numPics; p++ ) m.bilateralFilter( pics[p].org, pics[p].infos, pics[p].newPic, overallStrength[p] );
    m.m_threadPool = nullptr;
  }
  const auto t2 = std::chrono::steady_clock::now();
  secs[0] = std::chrono::duration<double>( t1 - t0 ).count();
  secs[1] = std::chrono::duration<double>( t2 - t1 ).count();
  for( int p = 0; p < numPics; p++ )
  {
    for( int i = 0; i < numRefs[p]; i++ ) if( finals && finals[pics[p].firstRef + i] ) dumpMvs( pics[p].infos[i].mvs, finals[pics[p].firstRef + i] );
    for( int c = 0; c < 3; c++ )
      if( out && out[3 * p + c] )
      {
        CPelBuf b = pics[p].newPic.bufs[c];
        for( int r = 0; r < ( int ) b.height; r++ ) memcpy( out[3 * p + c] + ( size_t ) r * b.width, b.buf + r * b.stride, sizeof( int16_t ) * b.width );
      }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Batch loops for CPU-baseline timing (bench.py cpu_baseline kind "reference"): the reference's own table entries
// called back-to-back from C++, like InterSearch::xTZSearchHelp does (EncoderLib/InterSearch.cpp:410-438).
// ---------------------------------------------------------------------------------------------
struct DistItem { int32_t org_off, cur_off; };

API void vvref_dist_batch( int simd, int dfBase, const int16_t* org, int orgStride, const int16_t* cur, int curStride,
                           int w, int h, int bitDepth, int subShift, const DistItem* items, int n, uint64_t* out )
{
  RdCost& rc = *rdPair().rc[simd ? 1 : 0];
  DistParam dp;
  dp.org = CPelBuf( org, orgStride, w, h );
  dp.cur = CPelBuf( cur, curStride, w, h );
  dp.bitDepth = bitDepth;
  dp.subShift = subShift;
  dp.compID   = COMP_Y;
  int idx = dfBase;
  if( dfBase != DF_HAD_2SAD && dfBase != DF_SAD_WITH_MASK ) idx += Log2( w );
  FpDistFunc f = rc.m_afpDistortFunc[0][idx];
  for( int i = 0; i < n; i++ )
  {
    dp.org.buf = org + items[i].org_off;
    dp.cur.buf = cur + items[i].cur_off;
    out[i] = f( dp );
  }
}

// xT -> needRdoq -> QuantCore -> DeQuantCore -> xIT -> SSE for n TUs (the fused pipeline's CPU twin; thread-safe only per simd value
// because g_tCoeffOps is a process-wide global: callers use ONE simd setting per process run)
static void tuRdoBatchTyped( int simd, const int16_t* resi, int resiStride, const int32_t* off, int n, int w, int h, int trHor, int trVer, int bitDepth,
                            const int16_t* qpFlags, int thrVal, int16_t* levelOut, int16_t* recOut, uint64_t* sseOut, int32_t* absSumOut );
API void vvref_tu_rdo_batch( int simd, const int16_t* resi, int resiStride, const int32_t* off, int n, int w, int h, int bitDepth,
                             const int16_t* qpFlags /* n x {qp, flags} */, int thrVal, int16_t* levelOut, int16_t* recOut, uint64_t* sseOut )
{
  tuRdoBatchTyped( simd, resi, resiStride, off, n, w, h, DCT2, DCT2, bitDepth, qpFlags, thrVal, levelOut, recOut, sseOut, nullptr );
}
static void tuRdoBatchTyped( int simd, const int16_t* resi, int resiStride, const int32_t* off, int n, int w, int h, int trHor, int trVer, int bitDepth,
                            const int16_t* qpFlags, int thrVal, int16_t* levelOut, int16_t* recOut, uint64_t* sseOut, int32_t* absSumOut )
{
  const int area = w * h;
  TCoeff* coef = ( TCoeff* ) xMalloc( TCoeff, area );
  TCoeff* deq  = ( TCoeff* ) xMalloc( TCoeff, area );
  TCoeff* du   = ( TCoeff* ) xMalloc( TCoeff, area );
  Pel*    rec  = ( Pel* ) xMalloc( Pel, area );
  TCoeffSig* lev = ( TCoeffSig* ) xMalloc( TCoeffSig, area );
  const int l = Log2( w ) + Log2( h ), sqrt2 = l & 1;
  RdCost& rc = *rdPair().rc[simd ? 1 : 0];
  for( int i = 0; i < n; i++ )
  {
    const int qp = qpFlags[2 * i], flags = qpFlags[2 * i + 1];
    const int trShift = 15 - bitDepth - ( l >> 1 ) - sqrt2;
    const int qBits = QUANT_SHIFT + qp / 6 + trShift;
    const int scale = g_quantScales[sqrt2][qp % 6];
    vvref_xT( simd, resi + off[i], resiStride, coef, w, h, trHor, trVer, bitDepth );
    volatile int need = quantObj().xNeedRdoq( coef, ( size_t ) w * std::min( h, 32 ), scale, int64_t( ( flags & 2 ) ? 171 : 256 ) << ( qBits - 9 ), qBits );
    ( void ) need;
    int32_t absSum; int last;
    vvref_quant_core( coef, lev, du, w, h, scale, qBits, int64_t( ( flags & 1 ) ? 171 : 85 ) << ( qBits - 9 ), 0, thrVal, &absSum, &last );
    const int rightShift = IQUANT_SHIFT - ( trShift + qp / 6 );
    const int tgt = std::min( 16, 32 + rightShift - 7 );
    quantObj().xDeQuant( w - 1, h - 1, g_invQuantScales[sqrt2][qp % 6], lev, w, deq, rightShift, ( 1 << ( tgt - 1 ) ) - 1, 32767 );
    vvref_xIT( simd, deq, rec, w, w, h, trHor, trVer, bitDepth );
    DistParam dp;
    dp.org = CPelBuf( resi + off[i], resiStride, w, h );
    dp.cur = CPelBuf( rec, w, w, h );
    dp.bitDepth = bitDepth; dp.compID = COMP_Y;
    const uint64_t sse = rc.m_afpDistortFunc[0][DF_SSE + Log2( w )]( dp );
    if( sseOut ) sseOut[i] = sse;
    if( absSumOut )      // four ints per TU: abs sum, last scan position, need-RDOQ flag, checksum of the levels sum (k + 1) * level[k] mod 2^32
    {
      uint32_t cs = 0; for( int k = 0; k < area; k++ ) cs += ( uint32_t ) ( k + 1 ) * ( uint32_t ) ( int32_t ) lev[k];
      absSumOut[4 * i] = absSum; absSumOut[4 * i + 1] = last; absSumOut[4 * i + 2] = need ? 1 : 0; absSumOut[4 * i + 3] = ( int32_t ) cs;
    }
    if( levelOut ) memcpy( levelOut + ( size_t ) i * area, lev, sizeof( int16_t ) * area );
    if( recOut ) memcpy( recOut + ( size_t ) i * area, rec, sizeof( int16_t ) * area );
  }
  xFree( coef ); xFree( deq ); xFree( du ); xFree( rec ); xFree( lev );
}

// Multi-threaded driver for bench.py's cpu_baseline (kind "reference") and its in-run parity check: `threads` std::threads pull chunks of CHUNK items from ONE atomic
// counter over all jobs (dynamic load balance: a late thread never leaves others idle, and the mix of cheap 8x8 and expensive 64x64 work evens out), `passes` times, through
// the reference's SIMD table entries; returns wall seconds.  jobs: kind 0 = distortion list (df = DFunc base), kind 1 = fused TU pipeline twin (vvref_tu_rdo_batch).
// out (may be NULL): n uint64 results of the job (distortion / SSE of the reconstructed residual) — what bench.py compares with the device's outputs.
// Scratch is per thread and allocated before the clock starts.
struct FrameJob { int32_t kind, df, size, subShift; const void* items; const void* aux; int32_t n, pad; uint64_t* out; };

API double vvref_run_jobs_mt( const int16_t* org, int orgStride, const int16_t* cur, int curStride, const int16_t* resi, int resiStride, int bitDepth,
                              const FrameJob* jobs, int nJobs, int threads, int passes )
{
  rdPair(); quantObj(); selectTCoeffOps( 1 );
  { static SPS* warm = new SPS; ( void ) warm; }
  const int CHUNK = 256;
  struct Chunk { int job, begin, end; };
  std::vector<Chunk> chunks;
  // expensive work first (largest blocks): the tail of the pass is made of cheap chunks
  std::vector<int> order( nJobs );
  for( int j = 0; j < nJobs; j++ ) order[j] = j;
  std::stable_sort( order.begin(), order.end(), [&]( int a, int b ) { return jobs[a].size * ( 1 + 8 * jobs[a].kind ) > jobs[b].size * ( 1 + 8 * jobs[b].kind ); } );
  for( int j : order ) for( int b = 0; b < jobs[j].n; b += CHUNK ) chunks.push_back( { j, b, std::min( jobs[j].n, b + CHUNK ) } );
  std::vector<std::atomic<int>> next( passes + 1 );
  for( auto& n : next ) n = 0;
  auto worker = [&]( int pass0, int pass1 )
  {
    uint64_t scratch[CHUNK];
    for( int p = pass0; p < pass1; p++ )
      for( ;; )
      {
        const int c = next[p]++;
        if( c >= ( int ) chunks.size() ) break;
        const Chunk& ck = chunks[c];
        const FrameJob& jb = jobs[ck.job];
        uint64_t* o = jb.out ? jb.out + ck.begin : scratch;
        if( jb.kind == 0 )
          vvref_dist_batch( 1, jb.df, org, orgStride, cur, curStride, jb.size, jb.size, bitDepth, jb.subShift, ( const DistItem* ) jb.items + ck.begin, ck.end - ck.begin, o );
        else
          vvref_tu_rdo_batch( 1, resi, resiStride, ( const int32_t* ) jb.items + ck.begin, ck.end - ck.begin, jb.size, jb.size, bitDepth, ( const int16_t* ) jb.aux + 2 * ck.begin, 8, nullptr, nullptr, o );
      }
  };
  worker( passes, passes + 1 );      // warm-up pass on the calling thread (also initialises the lazily built statics before threads start)
  // $VVREF_PIN: worker t stays on the t-th CPU of the process's affinity mask (a timing that does not depend on where the scheduler puts 16 threads on a 256-CPU host)
  std::vector<int> cpus;
  if( getenv( "VVREF_PIN" ) && atoi( getenv( "VVREF_PIN" ) ) )
  {
    cpu_set_t set; CPU_ZERO( &set );
    if( sched_getaffinity( 0, sizeof( set ), &set ) == 0 ) for( int c = 0; c < CPU_SETSIZE; c++ ) if( CPU_ISSET( c, &set ) ) cpus.push_back( c );
  }
  auto pinned = [&]( int t, int pass0, int pass1 )
  {
    if( ( int ) cpus.size() >= threads ) { cpu_set_t one; CPU_ZERO( &one ); CPU_SET( cpus[( size_t ) t * ( cpus.size() / threads )], &one ); pthread_setaffinity_np( pthread_self(), sizeof( one ), &one ); }
    worker( pass0, pass1 );
  };
  std::vector<std::thread> th;
  th.reserve( threads );
  const auto t0 = std::chrono::steady_clock::now();
  for( int t = 0; t < threads; t++ ) th.emplace_back( pinned, t, 0, passes );
  for( auto& x : th ) x.join();
  return std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
}

// ---------------------------------------------------------------------------------------------
// End-to-end: the reference encoder itself through its C API (vvenc/vvenc.h), one call = one sequence -> bitstream bytes.
// Planes are int16 samples (vvencYUVPlane), 4:2:0.  simd: NULL = auto (AVX2 here), "SCALAR", "SSE41", ...
// ---------------------------------------------------------------------------------------------
#include "vvenc/vvenc.h"
namespace {
void quietMsg( void*, int, const char*, va_list ) {}
}
// ---------------------------------------------------------------------------------------------
// InterpolationFilter (CommonLib/InterpolationFilter.{h,cpp}, x86 row x86/InterpolationFilterX86.h)
// ---------------------------------------------------------------------------------------------
namespace {
InterpolationFilter& ifObj( int simd )
{
  static InterpolationFilter* f[2] = { nullptr, nullptr };
  if( !f[0] ) { f[0] = new InterpolationFilter; f[1] = new InterpolationFilter; f[1]->initInterpolationFilter( true ); }
  return *f[simd ? 1 : 0];
}
int tapIdx( int N ) { return N == 8 ? 0 : N == 4 ? 1 : N == 2 ? 2 : 3; }     // InterpolationFilter.cpp:463-482
}

API int vvref_if_coeff( int set, int phase, int16_t* out8 )
{
  for( int i = 0; i < 8; i++ ) out8[i] = 0;
  switch( set )
  {
  case 0: for( int i = 0; i < 8; i++ ) out8[i] = InterpolationFilter::m_lumaFilter[phase][i]; return 8;
  case 1: for( int i = 0; i < 8; i++ ) out8[i] = InterpolationFilter::m_lumaFilter4x4[phase][i]; return 6;
  case 2: for( int i = 0; i < 4; i++ ) out8[i] = InterpolationFilter::m_chromaFilter[phase][i]; return 4;
  case 3: for( int i = 0; i < 8; i++ ) out8[i] = InterpolationFilter::m_lumaAltHpelIFilter[i]; return 6;
  case 4: for( int i = 0; i < 2; i++ ) out8[i] = InterpolationFilter::m_bilinearFilterPrec4[phase][i]; return 2;
  }
  return -1;
}

// the table slots m_filterHor / m_filterVer [tap index][isFirst][isLast] (InterpolationFilter.h:113-114)
API void vvref_if_filter( int simd, int N, int isVertical, int isFirst, int isLast, int bitDepth, const int16_t* src, int srcStride, int16_t* dst, int dstStride,
                          int width, int height, const int16_t* coeff )
{
  InterpolationFilter& f = ifObj( simd );
  ClpRng clp; clp.bd = bitDepth;
  ( isVertical ? f.m_filterVer : f.m_filterHor )[tapIdx( N )][isFirst ? 1 : 0][isLast ? 1 : 0]( clp, src, srcStride, dst, dstStride, width, height, coeff );
}

API void vvref_if_copy( int simd, int isFirst, int isLast, int bitDepth, const int16_t* src, int srcStride, int16_t* dst, int dstStride, int width, int height, int biMCForDMVR )
{
  ClpRng clp; clp.bd = bitDepth;
  ifObj( simd ).m_filterCopy[isFirst ? 1 : 0][isLast ? 1 : 0]( clp, src, srcStride, dst, dstStride, width, height, biMCForDMVR != 0 );
}

// public dispatchers filterHor / filterVer for luma (InterpolationFilter.cpp:557-661)
API void vvref_if_luma_1d( int simd, int vertical, const int16_t* src, int srcStride, int16_t* dst, int dstStride, int width, int height, int frac,
                           int isFirst, int isLast, int bitDepth, int useAltHpelIf, int reduceTap )
{
  InterpolationFilter& f = ifObj( simd );
  ClpRng clp; clp.bd = bitDepth;
  if( vertical ) f.filterVer( COMP_Y, src, srcStride, dst, dstStride, width, height, frac, isFirst != 0, isLast != 0, CHROMA_420, clp, useAltHpelIf != 0, 0, reduceTap );
  else           f.filterHor( COMP_Y, src, srcStride, dst, dstStride, width, height, frac, isLast != 0, CHROMA_420, clp, useAltHpelIf != 0, 0, reduceTap );
}

// The interpolation calls of InterPredInterpolation::xPredInterBlk (InterPrediction.cpp:832-865; no BDOF / DMVR / bilinear) made on the
// reference's own InterpolationFilter object (the function itself needs a CodingUnit and a Picture).
API void vvref_if_pred_luma( int simd, const int16_t* ref, int refStride, int16_t* dst, int dstStride, int width, int height, int xFrac, int yFrac,
                             int rndRes, int bitDepth, int useAltHpelIf )
{
  InterpolationFilter& f = ifObj( simd );
  ClpRng clp; clp.bd = bitDepth;
  const bool alt = useAltHpelIf != 0, rnd = rndRes != 0;
  if( yFrac == 0 )      f.filterHor( COMP_Y, ref, refStride, dst, dstStride, width, height, xFrac, rnd, CHROMA_420, clp, alt, 0 );
  else if( xFrac == 0 ) f.filterVer( COMP_Y, ref, refStride, dst, dstStride, width, height, yFrac, true, rnd, CHROMA_420, clp, alt, 0 );
  else if( width == 4 && height == 4 ) f.filter4x4( COMP_Y, ref, refStride, dst, dstStride, 4, 4, xFrac, yFrac, rnd, CHROMA_420, clp, alt );
  else if( width == 16 ) f.filter16xH( COMP_Y, ref, refStride, dst, dstStride, 16, height, xFrac, yFrac, rnd, CHROMA_420, clp, alt );
  else if( width == 8 )  f.filter8xH( COMP_Y, ref, refStride, dst, dstStride, 8, height, xFrac, yFrac, rnd, CHROMA_420, clp, alt );
  else
  {
    static thread_local std::vector<Pel> tmp;
    tmp.resize( ( size_t ) width * ( height + 8 ) );
    f.filterHor( COMP_Y, ref - 3 * refStride, refStride, tmp.data(), width, width, height + 7, xFrac, false, CHROMA_420, clp, alt, 0 );
    f.filterVer( COMP_Y, tmp.data() + 3 * width, width, dst, dstStride, width, height, yFrac, false, rnd, CHROMA_420, clp, alt, 0 );
  }
}

// ---------------------------------------------------------------------------------------------
// DMVR refinement search.  DMVR::xProcessDMVR (InterPrediction.cpp:1246-1400) needs a CodingUnit, a slice and reference
// pictures; the pieces it is made of are the reference's own and are called here: InterpolationFilter::filterN2_2D (bilinear prediction),
// RdCost::setDistParam(.., isDMVR) -> distFunc / dmvrSadX5, xSubPelErrorSrfc.  Only the 25-point loop around them is restated (:1322-1384).
// ---------------------------------------------------------------------------------------------
namespace vvenc { void xSubPelErrorSrfc( uint64_t* sadBuffer, int32_t* deltaMv ); }

API void vvref_if_bilinear( int simd, const int16_t* src, int srcStride, int16_t* dst, int dstStride, int w, int h, int fracX, int fracY, int bitDepth )
{
  ClpRng clp; clp.bd = bitDepth;
  ifObj( simd ).filterN2_2D( COMP_Y, src, srcStride, dst, dstStride, w, h, fracX, fracY, clp );
}

API void vvref_dmvr_subpel_error_surface( const uint64_t* sad5, int32_t* deltaMv )
{
  uint64_t b[5]; for( int i = 0; i < 5; i++ ) b[i] = sad5[i];
  xSubPelErrorSrfc( b, deltaMv );
}

API uint64_t vvref_dmvr_refine( int simd, const int16_t* ref0, int stride0, int fx0, int fy0, const int16_t* ref1, int stride1, int fx1, int fy1, int dx, int dy,
                                int bitDepth, int16_t* mvd )
{
  RdCost& rc = *rdPair().rc[simd ? 1 : 0];
  ClpRng clp; clp.bd = bitDepth;
  const int bs = dx + 4;
  Pel* p0 = ( Pel* ) xMalloc( Pel, bs * ( dy + 4 ) + 16 ); Pel* p1 = ( Pel* ) xMalloc( Pel, bs * ( dy + 4 ) + 16 );
  ifObj( simd ).filterN2_2D( COMP_Y, ref0 - 2 * stride0 - 2, stride0, p0, bs, dx + 4, dy + 4, fx0, fy0, clp );
  ifObj( simd ).filterN2_2D( COMP_Y, ref1 - 2 * stride1 - 2, stride1, p1, bs, dx + 4, dy + 4, fx1, fy1, clp );
  const Pel* l0 = p0 + 2 * bs + 2; const Pel* l1 = p1 + 2 * bs + 2;
  DistParam dp = rc.setDistParam( nullptr, nullptr, bs, bs, bitDepth, COMP_Y, dx, dy, 1, true );
  dp.org.buf = l0; dp.cur.buf = l1;
  uint64_t minCost = dp.distFunc( dp ) >> 1;
  minCost -= ( minCost >> 2 );
  mvd[0] = mvd[1] = 0;
  if( minCost >= ( uint64_t ) ( dx * dy ) )
  {
    uint64_t sadArray[25];
    int16_t total[2] = { 0, 0 }, delta[2] = { 0, 0 };
    sadArray[12] = minCost;
    for( int ver = -2; ver <= 2; ver++ )
    {
      const ptrdiff_t offset = -2 + ver * bs;
      dp.org.buf = l0 + offset; dp.cur.buf = l1 - offset;
      dp.dmvrSadX5( dp, &sadArray[( ver + 2 ) * 5], ver != 0 );
      for( int hor = -2; hor <= 2; hor++ ) if( sadArray[( ver + 2 ) * 5 + hor + 2] < minCost ) { minCost = sadArray[( ver + 2 ) * 5 + hor + 2]; delta[0] = hor; delta[1] = ver; }
    }
    total[0] = delta[0] * 16; total[1] = delta[1] * 16;
    if( abs( total[0] ) != 32 && abs( total[1] ) != 32 )
    {
      uint64_t* p = &sadArray[12 + delta[1] * 5 + delta[0]];
      uint64_t sb[5] = { p[0], p[-1], p[-5], p[1], p[5] };
      int32_t t[2] = { 0, 0 };
      xSubPelErrorSrfc( sb, t );
      total[0] += t[0]; total[1] += t[1];
    }
    mvd[0] = total[0]; mvd[1] = total[1];
  }
  xFree( p0 ); xFree( p1 );
  return minCost;
}

// ---------------------------------------------------------------------------------------------
// CPU timing loops for the SURVEY 8f rows (tests/perf_side_by_side.py): the reference's own entries called back-to-back from C++ on
// `threads` std::threads; return wall seconds.
// ---------------------------------------------------------------------------------------------
struct SubpelItem { int32_t org_off, ref_off; int16_t frac_x, frac_y; };
API double vvref_subpel_batch_mt( const int16_t* org, int orgStride, const int16_t* ref, int refStride, const SubpelItem* items, int n, int w, int h, int bitDepth,
                                  int dfBase, int threads, uint64_t* out )
{
  rdPair(); ifObj( 1 );
  auto worker = [&]( int t )
  {
    const int per = ( n + threads - 1 ) / threads, b = std::min( n, t * per ), e = std::min( n, b + per );
    RdCost& rc = *rdPair().rc[1];
    std::vector<Pel> pred( ( size_t ) w * h + 64 );
    for( int i = b; i < e; i++ )
    {
      vvref_if_pred_luma( 1, ref + items[i].ref_off, refStride, pred.data(), w, w, h, items[i].frac_x, items[i].frac_y, 1, bitDepth, 0 );
      DistParam dp;
      dp.org = CPelBuf( org + items[i].org_off, orgStride, w, h ); dp.cur = CPelBuf( pred.data(), w, w, h );
      dp.bitDepth = bitDepth; dp.subShift = 0; dp.compID = COMP_Y;
      out[i] = rc.m_afpDistortFunc[0][dfBase + Log2( w )]( dp );
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for( int t = 0; t < threads; t++ ) th.emplace_back( worker, t );
  for( auto& x : th ) x.join();
  return std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
}

struct DmvrItem { int32_t ref0_off, ref1_off; int16_t f0x, f0y, f1x, f1y; };
struct DmvrResult { int16_t mvd_x, mvd_y; int32_t pad; uint64_t min_cost; };
API double vvref_dmvr_batch_mt( const int16_t* ref0, int stride0, const int16_t* ref1, int stride1, const DmvrItem* items, int n, int dx, int dy, int bitDepth,
                                int threads, DmvrResult* out )
{
  rdPair(); ifObj( 1 );
  auto worker = [&]( int t )
  {
    const int per = ( n + threads - 1 ) / threads, b = std::min( n, t * per ), e = std::min( n, b + per );
    for( int i = b; i < e; i++ )
    {
      int16_t mvd[2];
      out[i].min_cost = vvref_dmvr_refine( 1, ref0 + items[i].ref0_off, stride0, items[i].f0x, items[i].f0y, ref1 + items[i].ref1_off, stride1, items[i].f1x, items[i].f1y,
                                           dx, dy, bitDepth, mvd );
      out[i].mvd_x = mvd[0]; out[i].mvd_y = mvd[1]; out[i].pad = 0;
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for( int t = 0; t < threads; t++ ) th.emplace_back( worker, t );
  for( auto& x : th ) x.join();
  return std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
}

// ---------------------------------------------------------------------------------------------
// Recorded work lists (vvenc_amd/recorded.py) on the reference's own entries: CPU baseline and in-run parity of bench.py's replay.  Every job carries its operand bases
// and pitches (picture planes or compact pool blocks).  kind 0: distortion list (df = DFunc base; items {org_off, cur_off}); kind 1: the fused TU pipeline's twin
// (items = residual offsets, aux = {qp, flags} pairs; out = SSE, out2 = {abs sum, last scan position, need-RDOQ, level checksum} per TU); kind 2: sub-pel refinement stages as xPatternRefinement computes them — one first
// pass per distinct horizontal position, second pass + distortion per evaluated position (items = RecStage; out = 9 costs per stage, untouched where not evaluated);
// kind 3: masked SAD on compact weight blocks (items = {org_off, cur_off, mask_off}, aux = mask base).
// `threads` std::threads pull chunks from one atomic counter, `passes` times; returns wall seconds.
// ---------------------------------------------------------------------------------------------
struct RecStage { int32_t org_off, ref_off; int8_t base_qx, base_qy; uint8_t i_frac, filter_mode, alt_hpel, had_mode; uint16_t mask; };
struct RecJob { int32_t kind, df, w, h, subShift, trHor, trVer, n; const int16_t* org; const int16_t* cur; int32_t orgStride, curStride; const void* items; const void* aux; uint64_t* out; int32_t* out2; };

static void recStages( const RecJob& jb, int begin, int end, int bitDepth, uint64_t* out )
{
  static const int8_t refH[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
  static const int8_t refQ[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { 1, 1 } };
  const int w = jb.w, h = jb.h;
  RdCost& rc = *rdPair().rc[1];
  static thread_local std::vector<Pel> tmp, pred;
  tmp.resize( ( size_t ) 3 * w * ( h + 8 ) ); pred.resize( ( size_t ) w * h + 64 );
  const RecStage* st = ( const RecStage* ) jb.items;
  for( int i = begin; i < end; i++ )
  {
    const RecStage& s = st[i];
    int hx[3], nHor = 0;
    const int dfBase = s.had_mode == 0 ? DF_SAD : ( s.had_mode == 1 ? DF_HAD : DF_HAD_fast );
    for( int k = 0; k < 9; k++ )
    {
      if( !( ( s.mask >> k ) & 1 ) ) continue;
      const int8_t* r = s.i_frac == 2 ? refH[k] : refQ[k];
      const int tx = ( r[0] + s.base_qx ) * s.i_frac * 4, ty = ( r[1] + s.base_qy ) * s.i_frac * 4;
      int v = 0; while( v < nHor && hx[v] != tx ) v++;
      if( v == nHor )
      {
        hx[nHor++] = tx;
        // first pass of rows -4 .. h + 3 (not last: 14-bit intermediates), the way xPatternRefinement fills m_filteredBlockTmp (InterSearch.cpp:817-848)
        vvref_if_luma_1d( 1, 0, jb.cur + s.ref_off + ( tx >> 4 ) - 4 * jb.curStride, jb.curStride, tmp.data() + ( size_t ) v * w * ( h + 8 ), w, w, h + 8, tx & 15, 1, 0, bitDepth, s.alt_hpel, s.filter_mode );
      }
      const Pel* hp = tmp.data() + ( size_t ) v * w * ( h + 8 ) + ( 4 + ( ty >> 4 ) ) * w;
      vvref_if_luma_1d( 1, 1, hp, w, pred.data(), w, w, h, ty & 15, 0, 1, bitDepth, s.alt_hpel, s.filter_mode );
      DistParam dp;
      dp.org = CPelBuf( jb.org + s.org_off, jb.orgStride, w, h ); dp.cur = CPelBuf( pred.data(), w, w, h );
      dp.bitDepth = bitDepth; dp.subShift = 0; dp.compID = COMP_Y;
      out[( size_t ) 9 * i + k] = rc.m_afpDistortFunc[0][dfBase + Log2( w )]( dp );
    }
  }
}

API double vvref_run_recorded_mt( const RecJob* jobs, int nJobs, int bitDepth, int threads, int passes )
{
  rdPair(); quantObj(); selectTCoeffOps( 1 ); ifObj( 1 );
  { static SPS* warm = new SPS; ( void ) warm; }
  struct Chunk { int job, begin, end; };
  std::vector<Chunk> chunks;
  std::vector<int> order( nJobs );
  for( int j = 0; j < nJobs; j++ ) order[j] = j;
  auto weight = [&]( int a ) { return ( long ) jobs[a].w * jobs[a].h * ( ( jobs[a].kind == 0 || jobs[a].kind == 3 ) ? 1 : ( jobs[a].kind == 1 ? 8 : 24 ) ); };
  std::stable_sort( order.begin(), order.end(), [&]( int a, int b ) { return weight( a ) > weight( b ); } );
  for( int j : order ) { const int CH = ( jobs[j].kind == 0 || jobs[j].kind == 3 ) ? 256 : ( jobs[j].kind == 1 ? 32 : 8 ); for( int b = 0; b < jobs[j].n; b += CH ) chunks.push_back( { j, b, std::min( jobs[j].n, b + CH ) } ); }
  std::vector<std::atomic<int>> next( passes + 1 );
  for( auto& n : next ) n = 0;
  auto worker = [&]( int pass0, int pass1 )
  {
    uint64_t scratch[9 * 256];
    for( int p = pass0; p < pass1; p++ )
      for( ;; )
      {
        const int c = next[p]++;
        if( c >= ( int ) chunks.size() ) break;
        const Chunk& ck = chunks[c];
        const RecJob& jb = jobs[ck.job];
        if( jb.kind == 0 )
          vvref_dist_batch( 1, jb.df, jb.org, jb.orgStride, jb.cur, jb.curStride, jb.w, jb.h, bitDepth, jb.subShift, ( const DistItem* ) jb.items + ck.begin, ck.end - ck.begin, jb.out ? jb.out + ck.begin : scratch );
        else if( jb.kind == 1 )
          tuRdoBatchTyped( 1, jb.org, jb.orgStride, ( const int32_t* ) jb.items + ck.begin, ck.end - ck.begin, jb.w, jb.h, jb.trHor, jb.trVer, bitDepth, ( const int16_t* ) jb.aux + 2 * ck.begin, 8, nullptr, nullptr,
                           jb.out ? jb.out + ck.begin : scratch, jb.out2 ? jb.out2 + 4 * ( size_t ) ck.begin : nullptr );
        else if( jb.kind == 3 )
        {
          // masked SAD (DF_SAD_WITH_MASK) on compact weight blocks: items = { org_off, cur_off, mask_off }, aux = the base the mask offsets refer to.  One mask row of w weights
          // per evaluated row: maskStride << subShift = w, maskStride2 = -w (the parameterisation both the scalar and the SIMD row walk identically, see vvref_sad_mask)
          const int32_t* it = ( const int32_t* ) jb.items;
          for( int i = ck.begin; i < ck.end; i++ )
          {
            const uint64_t v = vvref_sad_mask( 1, jb.org + it[3 * i], jb.orgStride, jb.cur + it[3 * i + 1], jb.curStride, ( const int16_t* ) jb.aux + it[3 * i + 2], jb.w >> jb.subShift, 1, -jb.w,
                                               jb.w, jb.h, bitDepth, jb.subShift );
            if( jb.out ) jb.out[i] = v;
          }
        }
        else
          recStages( jb, ck.begin, ck.end, bitDepth, jb.out ? jb.out : scratch - ( size_t ) 9 * ck.begin );
      }
  };
  worker( passes, passes + 1 );      // warm-up pass on the calling thread (also initialises the lazily built statics before threads start)
  // $VVREF_PIN: worker t stays on the t-th CPU of the process's affinity mask (a timing that does not depend on where the scheduler puts 16 threads on a 256-CPU host)
  std::vector<int> cpus;
  if( getenv( "VVREF_PIN" ) && atoi( getenv( "VVREF_PIN" ) ) )
  {
    cpu_set_t set; CPU_ZERO( &set );
    if( sched_getaffinity( 0, sizeof( set ), &set ) == 0 ) for( int c = 0; c < CPU_SETSIZE; c++ ) if( CPU_ISSET( c, &set ) ) cpus.push_back( c );
  }
  auto pinned = [&]( int t, int pass0, int pass1 )
  {
    if( ( int ) cpus.size() >= threads ) { cpu_set_t one; CPU_ZERO( &one ); CPU_SET( cpus[( size_t ) t * ( cpus.size() / threads )], &one ); pthread_setaffinity_np( pthread_self(), sizeof( one ), &one ); }
    worker( pass0, pass1 );
  };
  std::vector<std::thread> th;
  th.reserve( threads );
  const auto t0 = std::chrono::steady_clock::now();
  for( int t = 0; t < threads; t++ ) th.emplace_back( pinned, t, 0, passes );
  for( auto& x : th ) x.join();
  return std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
}

// the hook-enabled build re-installs table-level device slots after the SIMD initialisation rewrote the global tables
// ---- ALF encoder statistics — the reference's own classification and covariance accumulation ---------------------
// cls: 2 bytes per 4x4 block {classIdx, transposeIdx}, width/4 per row.  simd = 0: scalar table entry, 1: the x86 row.  Walks the picture in
// 128x128 CTUs exactly like deriveClassification (AdaptiveLoopFilter.cpp:505-522, m_CLASSIFICATION_BLK_SIZE 128).
API int vvref_alf_classify( const int16_t* rec, int stride, int width, int height, int shift, int vbCTUHeight, int vbPos, int simd, uint8_t* cls )
{
  static AdaptiveLoopFilter* alf[2] = { nullptr, nullptr };
  if( !alf[simd != 0] ) alf[simd != 0] = new AdaptiveLoopFilter( simd != 0 );
  std::vector<AlfClassifier> tmp( 32 * 32 );
  const CPelBuf src( rec, stride, width, height );
  for( int y = 0; y < height; y += 128 )
    for( int x = 0; x < width; x += 128 )
    {
      const int w = std::min( 128, width - x ), h = std::min( 128, height - y );
      const Area blk( x, y, w, h );
      alf[simd != 0]->m_deriveClassificationBlk( tmp.data(), src, blk, blk, shift, vbCTUHeight, vbPos );
      for( int i = 0; i < h; i += 4 )
        for( int j = 0; j < w; j += 4 )
        {
          const AlfClassifier& c = tmp[( i / 4 ) * 32 + j / 4];
          uint8_t* o = cls + 2 * ( ( size_t ) ( ( y + i ) / 4 ) * ( width / 4 ) + ( x + j ) / 4 );
          o[0] = c.classIdx; o[1] = c.transposeIdx;
        }
    }
  return 0;
}

// Statistics of one plane, CTU by CTU, through EncAdaptiveLoopFilter::getPreBlkStats (EncAdaptiveLoopFilter.cpp:3376) with linear filters
// (numBins 1).  out: [numCtus][numClasses][13*13 + 13 + 1] floats (E row-major, y, pixAcc); cls == NULL: chroma (one class).
API int vvref_alf_stats_plane_units( const int16_t* org, int orgStride, const int16_t* rec, int recStride, int width, int height, int unitSize, int ctuSize, int filterLength,
                                     const uint8_t* cls, int vbCTUHeight, int vbPos, int simd, float* out );
API int vvref_alf_stats_plane( const int16_t* org, int orgStride, const int16_t* rec, int recStride, int width, int height, int ctuSize, int filterLength,
                               const uint8_t* cls, int vbCTUHeight, int vbPos, int simd, float* out )
{
  return vvref_alf_stats_plane_units( org, orgStride, rec, recStride, width, height, ctuSize, ctuSize, filterLength, cls, vbCTUHeight, vbPos, simd, out );
}

// statistics units made of several CTUs, walked like EncAdaptiveLoopFilter::getStatisticsASU (EncAdaptiveLoopFilter.cpp:1568-1590): one covariance set per unit
API int vvref_alf_stats_plane_units( const int16_t* org, int orgStride, const int16_t* rec, int recStride, int width, int height, int unitSize, int ctuSize, int filterLength,
                                     const uint8_t* cls, int vbCTUHeight, int vbPos, int simd, float* out )
{
  static EncAdaptiveLoopFilter* enc[2] = { nullptr, nullptr };
  static VVEncCfg cfg;
  if( !enc[simd != 0] )
  {
    memset( ( void* ) &cfg, 0, sizeof( cfg ) );                       // m_useNonLinearAlfLuma / Chroma = false
    enc[simd != 0] = new EncAdaptiveLoopFilter( simd != 0 );
    enc[simd != 0]->m_encCfg = &cfg;
  }
  EncAdaptiveLoopFilter& E = *enc[simd != 0];
  const AlfFilterShape shape( filterLength );
  const int numClasses = cls ? MAX_NUM_ALF_CLASSES : 1, rec_ = 13 * 13 + 13 + 1;
  std::vector<AlfCovariance> cov( numClasses );
  for( auto& c : cov ) c.create( shape.numCoeff, 1 );
  std::vector<AlfClassifier> cl( 32 * 32 );
  const int ux = ( width + unitSize - 1 ) / unitSize, uy = ( height + unitSize - 1 ) / unitSize;
  for( int ay = 0; ay < uy; ay++ )
    for( int ax = 0; ax < ux; ax++ )
    {
      for( auto& c : cov ) c.reset();
      for( int y0 = ay * unitSize; y0 < ( ay + 1 ) * unitSize && y0 < height; y0 += ctuSize )
        for( int x0 = ax * unitSize; x0 < ( ax + 1 ) * unitSize && x0 < width; x0 += ctuSize )
        {
          const int w = std::min( ctuSize, width - x0 ), h = std::min( ctuSize, height - y0 );
          if( cls )
            for( int i = 0; i < h; i += 4 )
              for( int j = 0; j < w; j += 4 )
              {
                const uint8_t* c = cls + 2 * ( ( size_t ) ( ( y0 + i ) / 4 ) * ( width / 4 ) + ( x0 + j ) / 4 );
                cl[( i / 4 ) * 32 + j / 4] = AlfClassifier( c[0], c[1] );
              }
          const CompArea area( cls ? COMP_Y : COMP_Cb, CHROMA_420, Area( x0, y0, w, h ) );
          E.getPreBlkStats( cov.data(), shape, cls ? cl.data() : nullptr, const_cast<Pel*>( org ) + ( ptrdiff_t ) y0 * orgStride + x0, orgStride,
                            const_cast<Pel*>( rec ) + ( ptrdiff_t ) y0 * recStride + x0, recStride, area, cls ? CH_L : CH_C, vbCTUHeight, vbPos );
        }
      float* o = out + ( size_t ) ( ay * ux + ax ) * numClasses * rec_;
      for( int c = 0; c < numClasses; c++, o += rec_ )
      {
        memset( o, 0, sizeof( float ) * rec_ );
        for( int k = 0; k < shape.numCoeff; k++ )
        {
          for( int l = 0; l < shape.numCoeff; l++ ) o[k * 13 + l] = cov[c].E[0][0][k][l];
          o[169 + k] = cov[c].y[0][k];
        }
        o[182] = cov[c].pixAcc;
      }
    }
  for( auto& c : cov ) c.destroy();
  return 0;
}

// CC-ALF statistics of one chroma plane, CTU by CTU, through EncAdaptiveLoopFilter::getBlkStatsCcAlf (EncAdaptiveLoopFilter.cpp:6061); 4:2:0.
// out: [numCtus][183] floats, only E[0..6][0..6], y[0..6], pixAcc are meaningful (the x86 loop touches an uninitialised 8th row).
API int vvref_ccalf_stats_plane( const int16_t* orgC, int orgStride, const int16_t* slfC, int slfStride, const int16_t* recLuma, int recStride,
                                 int widthC, int heightC, int ctuSizeC, int vbCTUHeight, int vbPos, int picHeight, int simd, float* out )
{
  static EncAdaptiveLoopFilter* enc[2] = { nullptr, nullptr };
  if( !enc[simd != 0] ) enc[simd != 0] = new EncAdaptiveLoopFilter( simd != 0 );
  EncAdaptiveLoopFilter& E = *enc[simd != 0];
  E.m_chromaFormat = CHROMA_420; E.m_alfVBLumaCTUHeight = vbCTUHeight; E.m_alfVBLumaPos = vbPos; E.m_picHeight = picHeight; E.m_maxCUHeight = ctuSizeC * 2; E.m_maxCUWidth = ctuSizeC * 2;
  const AlfFilterShape shape( size_CC_ALF );
  AlfCovariance cov; cov.create( shape.numCoeff, 1 );
  // PelUnitBufs over the caller's planes: luma of rec, the chroma component as Cb of both
  const int wL = widthC * 2, hL = heightC * 2;
  PelUnitBuf recYuv, orgYuv;
  recYuv.chromaFormat = CHROMA_420; orgYuv.chromaFormat = CHROMA_420;
  recYuv.bufs.push_back( PelBuf( const_cast<Pel*>( recLuma ), recStride, wL, hL ) );
  recYuv.bufs.push_back( PelBuf( const_cast<Pel*>( slfC ), slfStride, widthC, heightC ) );
  recYuv.bufs.push_back( PelBuf( const_cast<Pel*>( slfC ), slfStride, widthC, heightC ) );
  static std::vector<Pel> dummyY; dummyY.assign( ( size_t ) wL * hL, 0 );
  orgYuv.bufs.push_back( PelBuf( dummyY.data(), wL, wL, hL ) );
  orgYuv.bufs.push_back( PelBuf( const_cast<Pel*>( orgC ), orgStride, widthC, heightC ) );
  orgYuv.bufs.push_back( PelBuf( const_cast<Pel*>( orgC ), orgStride, widthC, heightC ) );
  const int ctusX = ( widthC + ctuSizeC - 1 ) / ctuSizeC, ctusY = ( heightC + ctuSizeC - 1 ) / ctuSizeC;
  for( int cy = 0; cy < ctusY; cy++ )
    for( int cx = 0; cx < ctusX; cx++ )
    {
      const int xL = cx * ctuSizeC * 2, yL = cy * ctuSizeC * 2, w = std::min( ctuSizeC * 2, wL - xL ), h = std::min( ctuSizeC * 2, hL - yL );
      cov.reset();
      const UnitArea area( CHROMA_420, Area( xL, yL, w, h ) );
      E.getBlkStatsCcAlf( cov, shape, orgYuv, recYuv, area, area, COMP_Cb, yL );
      float* o = out + ( size_t ) ( cy * ctusX + cx ) * 183;
      memset( o, 0, sizeof( float ) * 183 );
      for( int k = 0; k < 7; k++ ) { for( int l = 0; l < 7; l++ ) o[k * 13 + l] = cov.E[0][0][k][l]; o[169 + k] = cov.y[0][k]; }
      o[182] = cov.pixAcc;
    }
  cov.destroy();
  return 0;
}


// ---- ALF / CC-ALF filtering through the reference's own table entries, CTU by CTU like EncAdaptiveLoopFilter::reconstructCTU (:2035-2066, branch
// without virtual picture boundaries) and applyCcAlfFilterCTU (:6606-6699).  coeffSets / clipSets: [numSets][numClasses][13]; ctuSet[ctu] < 0: skipped.
// nonLinear selects the table entry like m_encCfg->m_useNonLinearAlfLuma / Chroma does (the x86 row's entry 0 ignores the clipping values).
static CodingStructure& alfDummyCs()
{
  static XUCache xu;
  static CodingStructure cs( xu, nullptr );
  static Slice sl;
  static SPS sps;
  sps.chromaFormatIdc = CHROMA_420;
  sl.sps = &sps;
  cs.slice = &sl;
  return cs;
}

API int vvref_alf_filter_plane( const int16_t* src, int srcStride, int16_t* dst, int dstStride, int width, int height, int ctuSize, int bitDepth, int filterLength,
                                const uint8_t* cls, const int16_t* coeffSets, const int16_t* clipSets, const int16_t* ctuSet, int vbCTUHeight, int vbPos, int nonLinear, int simd )
{
  static AdaptiveLoopFilter* alf[2] = { nullptr, nullptr };
  if( !alf[simd != 0] ) alf[simd != 0] = new AdaptiveLoopFilter( simd != 0 );
  AdaptiveLoopFilter& A = *alf[simd != 0];
  CodingStructure& cs = alfDummyCs();
  const int numClasses = cls ? MAX_NUM_ALF_CLASSES : 1, ctusX = ( width + ctuSize - 1 ) / ctuSize;
  const ComponentID comp = cls ? COMP_Y : COMP_Cb;
  ClpRng rng; rng.bd = bitDepth;
  std::vector<AlfClassifier> cl( 32 * 32 );
  for( int y0 = 0; y0 < height; y0 += ctuSize )
    for( int x0 = 0; x0 < width; x0 += ctuSize )
    {
      const int set = ctuSet[( y0 / ctuSize ) * ctusX + x0 / ctuSize];
      if( set < 0 ) continue;
      const int w = std::min( ctuSize, width - x0 ), h = std::min( ctuSize, height - y0 );
      if( cls )
        for( int i = 0; i < h; i += 4 )
          for( int j = 0; j < w; j += 4 )
          {
            const uint8_t* c = cls + 2 * ( ( size_t ) ( ( y0 + i ) / 4 ) * ( width / 4 ) + ( x0 + j ) / 4 );
            cl[( i / 4 ) * 32 + j / 4] = AlfClassifier( c[0], c[1] );
          }
      PelUnitBuf recDst, recSrc;
      recDst.chromaFormat = recSrc.chromaFormat = CHROMA_420;
      for( int k = 0; k < ( cls ? 1 : 2 ); k++ )
      {
        recDst.bufs.push_back( PelBuf( dst, dstStride, width, height ) );
        recSrc.bufs.push_back( PelBuf( const_cast<Pel*>( src ) + ( ptrdiff_t ) y0 * srcStride + x0, srcStride, w, h ) );
      }
      const Area blkDst( x0, y0, w, h ), blk( 0, 0, w, h );
      const short* cf = coeffSets + ( size_t ) set * numClasses * MAX_NUM_ALF_LUMA_COEFF;
      const short* cp = clipSets + ( size_t ) set * numClasses * MAX_NUM_ALF_LUMA_COEFF;
      if( filterLength == 7 ) A.m_filter7x7Blk[nonLinear != 0]( cl.data(), recDst, recSrc, blkDst, blk, comp, cf, cp, rng, cs, vbCTUHeight, vbPos );
      else                    A.m_filter5x5Blk[nonLinear != 0]( cl.data(), recDst, recSrc, blkDst, blk, comp, cf, cp, rng, cs, vbCTUHeight, vbPos );
    }
  return 0;
}

API int vvref_ccalf_filter_plane( int16_t* dstC, int dstStride, const int16_t* recLuma, int recStride, int widthC, int heightC, int ctuSizeC, int bitDepth,
                                  const int16_t* coeff, const uint8_t* ctuFilter, int vbCTUHeight, int vbPos, int simd )
{
  static AdaptiveLoopFilter* alf[2] = { nullptr, nullptr };
  if( !alf[simd != 0] ) alf[simd != 0] = new AdaptiveLoopFilter( simd != 0 );
  AdaptiveLoopFilter& A = *alf[simd != 0];
  CodingStructure& cs = alfDummyCs();
  ClpRngs rngs; rngs.bd = bitDepth;
  const int ctusX = ( widthC + ctuSizeC - 1 ) / ctuSizeC;
  PelUnitBuf recYuv; recYuv.chromaFormat = CHROMA_420;
  recYuv.bufs.push_back( PelBuf( const_cast<Pel*>( recLuma ), recStride, widthC * 2, heightC * 2 ) );
  const PelBuf dstBuf( dstC, dstStride, widthC, heightC );
  for( int y0 = 0; y0 < heightC; y0 += ctuSizeC )
    for( int x0 = 0; x0 < widthC; x0 += ctuSizeC )
    {
      const int f = ctuFilter[( y0 / ctuSizeC ) * ctusX + x0 / ctuSizeC];
      if( !f ) continue;
      const int w = std::min( ctuSizeC, widthC - x0 ), h = std::min( ctuSizeC, heightC - y0 );
      const Area blkDst( x0, y0, w, h ), blkSrc( x0 * 2, y0 * 2, w * 2, h * 2 );
      A.m_filterCcAlf( dstBuf, recYuv, blkDst, blkSrc, COMP_Cb, coeff + ( size_t ) ( f - 1 ) * MAX_NUM_CC_ALF_CHROMA_COEFF, rngs, cs, vbCTUHeight, vbPos );
    }
  return 0;
}

extern "C" void vvref_after_simd_init() __attribute__( ( weak ) );

// options: "name=value;name=value" handed to vvenc_set_param one by one after the preset (e.g. "RDOQ=0;LFNST=1"); simd may be "HIP[:mask]" in the build with the binding
API long vvref_encode_ex( const int16_t* y, const int16_t* u, const int16_t* v, int width, int height, int frames, int inputBitDepth, int internalBitDepth,
                          int preset, int qp, int threads, const char* simd, const char* options, uint8_t* out, long outCap, double* secondsOut )
{
  if( simd && simd[0] && !vvenc_set_SIMD_extension( simd ) ) { fprintf( stderr, "vvref_encode: SIMD request %s refused\n", simd ); return -4; }
  if( !( simd && simd[0] ) ) vvenc_set_SIMD_extension( nullptr );
  vvenc_config cfg;
  vvenc_init_default( &cfg, width, height, 30, 0, qp, ( vvencPresetMode ) preset );
  cfg.m_inputBitDepth[0] = inputBitDepth;
  cfg.m_internalBitDepth[0] = internalBitDepth;
  cfg.m_numThreads = threads;
  cfg.m_verbosity = VVENC_SILENT;
  if( options && options[0] )
  {
    std::string o( options );
    size_t pos = 0;
    while( pos < o.size() )
    {
      size_t end = o.find( ';', pos ); if( end == std::string::npos ) end = o.size();
      const std::string kv = o.substr( pos, end - pos ); pos = end + 1;
      const size_t eq = kv.find( '=' );
      if( eq == std::string::npos ) continue;
      const int rc = vvenc_set_param( &cfg, kv.substr( 0, eq ).c_str(), kv.substr( eq + 1 ).c_str() );
      if( rc != 0 ) { fprintf( stderr, "vvref_encode: vvenc_set_param( %s ) -> %d\n", kv.c_str(), rc ); return -5; }
    }
  }
  vvenc_set_msg_callback( &cfg, nullptr, quietMsg );
  vvencEncoder* enc = vvenc_encoder_create();
  if( !enc ) return -1;
  if( vvenc_encoder_open( enc, &cfg ) != 0 ) { fprintf( stderr, "vvref_encode: open failed: %s\n", vvenc_get_last_error( enc ) ); vvenc_encoder_close( enc ); return -2; }
  if( vvref_after_simd_init ) vvref_after_simd_init();        // the encoder's constructor re-ran the SIMD initialisation (vvencimpl.cpp:96)
  vvencYUVBuffer yuv; vvenc_YUVBuffer_default( &yuv );
  vvenc_YUVBuffer_alloc_buffer( &yuv, VVENC_CHROMA_420, width, height );
  vvencAccessUnit au; vvenc_accessUnit_default( &au );
  vvenc_accessUnit_alloc_payload( &au, ( 3 * width * height ) / 2 + 1024 * 64 );
  const int cw = width / 2, ch = height / 2;
  long used = 0; bool done = false; int rc = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for( int f = 0; f < frames && !rc; f++ )
  {
    for( int r = 0; r < height; r++ ) memcpy( yuv.planes[0].ptr + ( size_t ) r * yuv.planes[0].stride, y + ( ( size_t ) f * height + r ) * width, sizeof( int16_t ) * width );
    for( int r = 0; r < ch; r++ )
    {
      memcpy( yuv.planes[1].ptr + ( size_t ) r * yuv.planes[1].stride, u + ( ( size_t ) f * ch + r ) * cw, sizeof( int16_t ) * cw );
      memcpy( yuv.planes[2].ptr + ( size_t ) r * yuv.planes[2].stride, v + ( ( size_t ) f * ch + r ) * cw, sizeof( int16_t ) * cw );
    }
    yuv.sequenceNumber = f; yuv.cts = f; yuv.ctsValid = true;
    rc = vvenc_encode( enc, &yuv, &au, &done );
    if( !rc && au.payloadUsedSize > 0 ) { if( used + au.payloadUsedSize > outCap ) rc = -100; else { memcpy( out + used, au.payload, au.payloadUsedSize ); used += au.payloadUsedSize; } }
  }
  while( !rc && !done )
  {
    rc = vvenc_encode( enc, nullptr, &au, &done );
    if( !rc && au.payloadUsedSize > 0 ) { if( used + au.payloadUsedSize > outCap ) rc = -100; else { memcpy( out + used, au.payload, au.payloadUsedSize ); used += au.payloadUsedSize; } }
  }
  if( secondsOut ) *secondsOut = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
  if( rc ) fprintf( stderr, "vvref_encode: error %d: %s\n", rc, vvenc_get_last_error( enc ) );
  vvenc_YUVBuffer_free_buffer( &yuv );
  vvenc_accessUnit_free_payload( &au );
  vvenc_encoder_close( enc );
  return rc ? -3 : used;
}

API long vvref_encode( const int16_t* y, const int16_t* u, const int16_t* v, int width, int height, int frames, int inputBitDepth, int internalBitDepth,
                       int preset, int qp, int threads, const char* simd, uint8_t* out, long outCap, double* secondsOut )
{
  return vvref_encode_ex( y, u, v, width, height, frames, inputBitDepth, internalBitDepth, preset, qp, threads, simd, nullptr, out, outCap, secondsOut );
}

API const char* vvref_version() { return "vvenc reference 1.15.0-dev (built from /root/reference by oracle/ref/Makefile)"; }
