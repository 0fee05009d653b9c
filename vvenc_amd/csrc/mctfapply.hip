// mctfapply.hip — SURVEY §8f rank 2: the apply side of the motion-compensated temporal pre-filter.
//
// Reference behaviour (scalar row, followed operation by operation — the float corners keep the C++ promotions of the source):
//   applyFrac8Core_6Tap / _4Tap       CommonLib/MCTF.cpp:259-358   (first pass truncated to Pel, NOT clipped; second pass clipped)
//   applyPlanarCorrectionCore         :372-421                     (fixed-point plane fit of the compensation error, "deblocking")
//   applyBlockCore                    :423-518                     (noise estimate, per-reference weights, bilateral blend with fastExp :359-367)
//   MCTF::xFinalizeBlkLine            :1399-1487                   (per block: compensate every reference, correct, blend)
//   MCTF::bilateralFilter             :1489-1552                   (sigma / strength per channel, all block rows)
// The reference's unit test allows +-1 between its scalar and x86 rows here (test/vvenc_unit_test/vvenc_unit_test.cpp:1280-1282); this
// kernel follows the scalar row (IEEE float/double, no contraction: -ffp-contract=off, correctly rounded division) — which IS the x86 row sample for
// sample on x86 hosts: the rows differ only in "+ 0.5" (double) vs "+ 0.5f" (single), and the single-precision sum is exact wherever the integer part
// could change (both rows held to tolerance 0 by the CPU tests of the repository's checker against the compiled reference, round 6).
#include <math.h>
#include <stdlib.h>
#include "common.h"

namespace {

__constant__ __attribute__( ( aligned( 16 ) ) ) int16_t cApply6[16][8] = {      // MCTF::m_interpolationFilter8 (MCTF.cpp:72-90), taps 1..6 are used
  { 0, 0, 0, 64, 0, 0, 0, 0 },    { 0, 1, -3, 64, 4, -2, 0, 0 },    { 0, 1, -6, 62, 9, -3, 1, 0 },    { 0, 2, -8, 60, 14, -5, 1, 0 },
  { 0, 2, -9, 57, 19, -7, 2, 0 }, { 0, 3, -10, 53, 24, -8, 2, 0 },  { 0, 3, -11, 50, 29, -9, 2, 0 },  { 0, 3, -11, 44, 35, -10, 3, 0 },
  { 0, 1, -7, 38, 38, -7, 1, 0 }, { 0, 3, -10, 35, 44, -11, 3, 0 }, { 0, 2, -9, 29, 50, -11, 3, 0 },  { 0, 2, -8, 24, 53, -10, 3, 0 },
  { 0, 2, -7, 19, 57, -9, 2, 0 }, { 0, 1, -5, 14, 60, -8, 2, 0 },   { 0, 1, -3, 9, 62, -6, 1, 0 },    { 0, 0, -2, 4, 64, -3, 1, 0 } };
__constant__ int16_t cApply4[16][4] = {      // MCTF::m_interpolationFilter4 (MCTF.cpp:92-110)
  { 0, 64, 0, 0 },    { -2, 62, 4, 0 },   { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
  { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, { 0, 4, 62, -2 } };

constexpr int MAX_REFS = 12;     // 2 * VVENC_MCTF_RANGE
constexpr int MAX_BLK  = 32;     // unit sizes 8/16/32 (luma), halves for 4:2:0 chroma

struct ApplyArgs
{
  const int16_t* refs[MAX_REFS];
  const vvhip_mv* mvs[MAX_REFS];
  double refStrengths[MAX_REFS];
  double weightScaling, sigmaSq;
  int numRefs, cs, bitDepth, blk, lowRes, qp, mvW, width, height;
  int generic;       // $VVHIP_MCTF_APPLY_GENERIC=1: every block through the general path (A/B measurements; results identical)
};

__device__ __forceinline__ float fastExp( float n, float d )     // MCTF.cpp:359-367
{
  float x = 1.0f + n / ( d * 1024 );
  x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x; x *= x;
  return x;
}

#define WAVE_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }

// ---- packed helpers of the full-block path (round 6) ---------------------------------------------------------------------------------------------------
typedef uint32_t au32x4 __attribute__( ( ext_vector_type( 4 ) ) );
typedef short as16x2 __attribute__( ( ext_vector_type( 2 ) ) );
struct __attribute__( ( packed, aligned( 2 ) ) ) AU4  { uint32_t v; };
struct __attribute__( ( packed, aligned( 2 ) ) ) AU16 { au32x4 v; };
__device__ __forceinline__ uint32_t ald4( const int16_t* p ) { return reinterpret_cast<const AU4*>( p )->v; }
__device__ __forceinline__ au32x4 ald16( const int16_t* p ) { return reinterpret_cast<const AU16*>( p )->v; }
__device__ __forceinline__ int adot2( uint32_t a, uint32_t b, int c ) { return __builtin_amdgcn_sdot2( __builtin_bit_cast( as16x2, a ), __builtin_bit_cast( as16x2, b ), c, false ); }
__device__ __forceinline__ uint32_t apk( int lo, int hi ) { return ( uint32_t ) ( lo & 0xffff ) | ( ( uint32_t ) hi << 16 ); }
__device__ __forceinline__ int alo( uint32_t v ) { return ( int ) ( int16_t ) ( v & 0xffffu ); }
__device__ __forceinline__ int ahi( uint32_t v ) { return ( int ) ( ( int32_t ) v >> 16 ); }
__device__ __forceinline__ int aclip( int v, int maxv ) { return v < 0 ? 0 : ( v > maxv ? maxv : v ); }

// ( numer +- denom / 2 ) / denom of applyPlanarCorrection (MCTF.cpp:403-411), kept out of line: the 64-bit division is ~100 instructions and the kernel holds six copies otherwise
__device__ __attribute__( ( noinline ) ) int planarDiv( long long numer, long long denom )
{
  return ( int ) ( ( numer < 0 ? numer - ( denom >> 1 ) : numer + ( denom >> 1 ) ) / denom );
}

// One reference of a FULL B x B block (B = 16 luma / 8 chroma at unit 16; 6-tap filter) by one wavefront — the same integers as the general path below, restated on sample
// PAIRS: the horizontal pass takes two outputs from one 16-byte request (tap pairs as v_dot2_i32_i16, the odd-phase pairs through v_alignbit), the vertical pass gives every
// lane a 2 x 2 patch (column pair xp, rows 2 rg, 2 rg + 1: seven dword rows of the intermediate, pairs of rows interleaved with v_perm), and that patch STAYS in registers
// through the planar correction and the noise estimate (its right / lower neighbours' differences arrive by three lane shuffles); block sums are 32-bit (a block has <= 256
// samples of |diff| <= 1023: variance < 2^28, diffsum < 2^31).  sT: ( B + 5 ) x B intermediate of the wave; corr: B x B result (raster, pitch B).
template<int B>
__device__ __forceinline__ void applyRefFull( const int16_t* __restrict__ src, int refStride, const int16_t* __restrict__ orgBlk, int orgStride, int dxF, int dyF, int maxv,
                                              unsigned rmsme, bool planar, int bitDepth, int16_t* __restrict__ sT, int16_t* __restrict__ corr, int& noiseOut, int lane )
{
  constexpr int HP = B / 2, LOG2HP = B == 16 ? 3 : 2, LOG2B = B == 16 ? 4 : 3;
  {
    const int16_t* xf = cApply6[dxF];
    const uint32_t t12 = apk( xf[1], xf[2] ), t34 = apk( xf[3], xf[4] ), t56 = apk( xf[5], xf[6] );
    for( int e = lane; e < ( B + 5 ) * HP; e += 64 )       // intermediate row rr <-> source row rr - 2
    {
      const int rr = e >> LOG2HP, xp = e & ( HP - 1 );
      const au32x4 d = ald16( src + ( ptrdiff_t ) ( rr - 2 ) * refStride + 2 * xp - 2 );      // samples x - 2 .. x + 5 of the row, x = 2 xp
      const int o0 = adot2( d.x, t12, adot2( d.y, t34, adot2( d.z, t56, 32 ) ) );
      const int o1 = adot2( __builtin_amdgcn_alignbit( d.y, d.x, 16 ), t12, adot2( __builtin_amdgcn_alignbit( d.z, d.y, 16 ), t34, adot2( __builtin_amdgcn_alignbit( d.w, d.z, 16 ), t56, 32 ) ) );
      *reinterpret_cast<uint32_t*>( sT + rr * B + 2 * xp ) = apk( o0 >> 6, o1 >> 6 );       // ( Pel ) truncation, no clip (MCTF.cpp:287-300)
    }
  }
  WAVE_SYNC();
  const bool act = lane < HP * HP;                       // B = 8: sixteen lanes hold the block
  const int xp = lane & ( HP - 1 ), rg = ( lane >> LOG2HP ) & ( HP - 1 ), x = 2 * xp, y = 2 * rg;
  int c00, c01, c10, c11;
  {
    const int16_t* yf = cApply6[dyF];
    const uint32_t u12 = apk( yf[1], yf[2] ), u34 = apk( yf[3], yf[4] ), u56 = apk( yf[5], yf[6] );
    const uint32_t* q = reinterpret_cast<const uint32_t*>( sT + y * B + x );
    uint32_t R[7];
#pragma unroll
    for( int k = 0; k < 7; k++ ) R[k] = q[k * HP];
#define ALO( A, Bv ) __builtin_amdgcn_perm( Bv, A, 0x05040100u )
#define AHI( A, Bv ) __builtin_amdgcn_perm( Bv, A, 0x07060302u )
    c00 = aclip( adot2( ALO( R[0], R[1] ), u12, adot2( ALO( R[2], R[3] ), u34, adot2( ALO( R[4], R[5] ), u56, 32 ) ) ) >> 6, maxv );
    c01 = aclip( adot2( AHI( R[0], R[1] ), u12, adot2( AHI( R[2], R[3] ), u34, adot2( AHI( R[4], R[5] ), u56, 32 ) ) ) >> 6, maxv );
    c10 = aclip( adot2( ALO( R[1], R[2] ), u12, adot2( ALO( R[3], R[4] ), u34, adot2( ALO( R[5], R[6] ), u56, 32 ) ) ) >> 6, maxv );
    c11 = aclip( adot2( AHI( R[1], R[2] ), u12, adot2( AHI( R[3], R[4] ), u34, adot2( AHI( R[5], R[6] ), u56, 32 ) ) ) >> 6, maxv );
#undef ALO
#undef AHI
  }
  const uint32_t og0 = ald4( orgBlk + ( ptrdiff_t ) y * orgStride + x ), og1 = ald4( orgBlk + ( ptrdiff_t ) ( y + 1 ) * orgStride + x );
  const int o00 = alo( og0 ), o01 = ahi( og0 ), o10 = alo( og1 ), o11 = ahi( og1 );
  if( planar )                                           // MCTF.cpp:372-421 (wave-uniform)
  {
    const int z00 = act ? c00 - o00 : 0, z01 = act ? c01 - o01 : 0, z10 = act ? c10 - o10 : 0, z11 = act ? c11 - o11 : 0;
    const int x1yzm = ( int ) vvhipGroupSum32( ( uint32_t ) ( x * ( z00 + z10 ) + ( x + 1 ) * ( z01 + z11 ) ), 64, lane );
    const int x2yzm = ( int ) vvhipGroupSum32( ( uint32_t ) ( y * ( z00 + z01 ) + ( y + 1 ) * ( z10 + z11 ) ), 64, lane );
    const int ySum  = ( int ) vvhipGroupSum32( ( uint32_t ) ( z00 + z01 + z10 + z11 ), 64, lane );
    const int xSzm[6] = { 0, 1, 20, 336, 5440, 87296 };
    constexpr int blockSize = B * B, log2Width = LOG2B;
    const unsigned me2 = rmsme * rmsme;
    const int mWeight = ( int ) ( me2 < 512u ? me2 : 512u );
    constexpr int xSum = ( blockSize * ( B - 1 ) ) >> 1;
    const long long denom = ( long long ) blockSize * xSzm[log2Width];
    long long numer = ( long long ) mWeight * ( ( long long ) x1yzm * blockSize - xSum * ySum );
    int b1 = planarDiv( numer, denom );
    b1 = b1 < -32768 ? -32768 : ( b1 > 32767 ? 32767 : b1 );
    numer = ( long long ) mWeight * ( ( long long ) x2yzm * blockSize - xSum * ySum );
    int b2 = planarDiv( numer, denom );
    b2 = b2 > 32767 ? 32767 : ( b2 < -32768 ? -32768 : b2 );
    const int b0 = ( mWeight * ySum - ( b1 + b2 ) * xSum + ( blockSize >> 1 ) ) >> ( log2Width << 1 );
    if( b0 != 0 || b1 != 0 || b2 != 0 )
    {
      const int p00 = b0 + b1 * x + b2 * y + 256;
      c00 = aclip( c00 - ( p00 >> 9 ), maxv );
      c01 = aclip( c01 - ( ( p00 + b1 ) >> 9 ), maxv );
      c10 = aclip( c10 - ( ( p00 + b2 ) >> 9 ), maxv );
      c11 = aclip( c11 - ( ( p00 + b1 + b2 ) >> 9 ), maxv );
    }
  }
  if( act )
  {
    *reinterpret_cast<uint32_t*>( corr + y * B + x ) = apk( c00, c01 );
    *reinterpret_cast<uint32_t*>( corr + ( y + 1 ) * B + x ) = apk( c10, c11 );
  }
  // ---- noise estimate (MCTF.cpp:445-477): diff = org - corrected; variance = sum diff^2; diffsum = sum of squared differences of horizontally / vertically adjacent diffs
  const int d00 = o00 - c00, d01 = o01 - c01, d10 = o10 - c10, d11 = o11 - c11;
  const uint32_t P0 = apk( d00, d01 ), P1 = apk( d10, d11 );
  const uint32_t nr0 = ( uint32_t ) __shfl_down( ( int ) P0, 1 ), nr1 = ( uint32_t ) __shfl_down( ( int ) P1, 1 ), nd = ( uint32_t ) __shfl_down( ( int ) P0, HP );
  uint32_t var = 0, ds = 0;
  if( act )
  {
    var = ( uint32_t ) ( d00 * d00 + d01 * d01 + d10 * d10 + d11 * d11 );
    int t;
    t = d01 - d00; ds += ( uint32_t ) ( t * t );  t = d11 - d10; ds += ( uint32_t ) ( t * t );                           // right neighbours inside the patch
    t = d10 - d00; ds += ( uint32_t ) ( t * t );  t = d11 - d01; ds += ( uint32_t ) ( t * t );                           // lower neighbours inside the patch
    if( xp != HP - 1 ) { t = alo( nr0 ) - d01; ds += ( uint32_t ) ( t * t ); t = alo( nr1 ) - d11; ds += ( uint32_t ) ( t * t ); }      // column 2 xp + 2 of the next lane
    if( rg != HP - 1 ) { t = alo( nd ) - d10; ds += ( uint32_t ) ( t * t ); t = ahi( nd ) - d11; ds += ( uint32_t ) ( t * t ); }        // row 2 rg + 2 of the lane HP further on
  }
  const uint32_t varSum = vvhipGroupSum32( var, 64, lane ), dsSum = vvhipGroupSum32( ds, 64, lane );
  long long variance = ( long long ) varSum, diffsum = ( long long ) dsSum;
  variance *= ( long long ) 1 << ( 2 * ( 10 - bitDepth ) );
  diffsum  *= ( long long ) 1 << ( 2 * ( 10 - bitDepth ) );
  constexpr int cntV = B * B, cntD = 2 * cntV - B - B;
  noiseOut = ( int ) round( ( 15.0 * cntD / cntV * variance + 5.0 ) / ( diffsum + 5.0 ) );
  WAVE_SYNC();                                           // sT is reused by the wave's next reference
}

// FOUR references of a full 8 x 8 block (the chroma planes at unit 16) in one pass of the wave: lane = 16 ref + 4 rg + xp — an 8 x 8 block is sixteen 2 x 2 patches, so one
// reference leaves three quarters of a wave idle and repeats the per-reference scalar work (plane fit with its two 64-bit divisions, noise estimate in double) per pass; with
// the references side by side that work runs once per lane group.  Same integers as applyRefFull<8> per reference.  Per-reference inputs (plane pointer, vector, rmsme) come
// from LDS (filled with uniform indices by the caller: kernel arguments are not indexed per lane).  sT: 4 x ( 13 x 8 ) intermediates; corrBase + ref * corrPitch: results.
struct ApplyRef8 { const int16_t* src; int dxF, dyF; unsigned rmsme; int planar; };
__device__ __forceinline__ void applyRefs8( const ApplyRef8* __restrict__ sRef, int nRef, int refStride, const int16_t* __restrict__ orgBlk, int orgStride, int maxv, int bitDepth,
                                            int16_t* __restrict__ sT, int16_t* __restrict__ corrBase, int corrPitch, int* __restrict__ sNoise, int lane )
{
  constexpr int B = 8, HP = 4, ROWS = B + 5, ITEMS = ROWS * HP;      // 52 sample pairs of the intermediate per reference
  for( int e = lane; e < nRef * ITEMS; e += 64 )
  {
    const int ref = e / ITEMS, within = e - ref * ITEMS, rr = within >> 2, xp = within & 3;
    const ApplyRef8 R = sRef[ref];
    const au32x4 tp = *reinterpret_cast<const au32x4*>( cApply6[R.dxF] );                      // taps 0..7 as pairs; wanted: (1,2) (3,4) (5,6)
    const uint32_t t12 = __builtin_amdgcn_alignbit( tp.y, tp.x, 16 ), t34 = __builtin_amdgcn_alignbit( tp.z, tp.y, 16 ), t56 = __builtin_amdgcn_alignbit( tp.w, tp.z, 16 );
    const au32x4 d = ald16( R.src + ( ptrdiff_t ) ( rr - 2 ) * refStride + 2 * xp - 2 );
    const int o0 = adot2( d.x, t12, adot2( d.y, t34, adot2( d.z, t56, 32 ) ) );
    const int o1 = adot2( __builtin_amdgcn_alignbit( d.y, d.x, 16 ), t12, adot2( __builtin_amdgcn_alignbit( d.z, d.y, 16 ), t34, adot2( __builtin_amdgcn_alignbit( d.w, d.z, 16 ), t56, 32 ) ) );
    *reinterpret_cast<uint32_t*>( sT + ref * ( ROWS * B ) + rr * B + 2 * xp ) = apk( o0 >> 6, o1 >> 6 );
  }
  WAVE_SYNC();
  const int ref = lane >> 4, xp = lane & 3, rg = ( lane >> 2 ) & 3, x = 2 * xp, y = 2 * rg;
  const bool act = ref < nRef;
  const ApplyRef8 R = sRef[act ? ref : 0];
  int c00, c01, c10, c11;
  {
    const au32x4 tp = *reinterpret_cast<const au32x4*>( cApply6[R.dyF] );
    const uint32_t u12 = __builtin_amdgcn_alignbit( tp.y, tp.x, 16 ), u34 = __builtin_amdgcn_alignbit( tp.z, tp.y, 16 ), u56 = __builtin_amdgcn_alignbit( tp.w, tp.z, 16 );
    const uint32_t* q = reinterpret_cast<const uint32_t*>( sT + ( act ? ref : 0 ) * ( ROWS * B ) + y * B + x );
    uint32_t Rw[7];
#pragma unroll
    for( int k = 0; k < 7; k++ ) Rw[k] = q[k * HP];
#define ALO( A, Bv ) __builtin_amdgcn_perm( Bv, A, 0x05040100u )
#define AHI( A, Bv ) __builtin_amdgcn_perm( Bv, A, 0x07060302u )
    c00 = aclip( adot2( ALO( Rw[0], Rw[1] ), u12, adot2( ALO( Rw[2], Rw[3] ), u34, adot2( ALO( Rw[4], Rw[5] ), u56, 32 ) ) ) >> 6, maxv );
    c01 = aclip( adot2( AHI( Rw[0], Rw[1] ), u12, adot2( AHI( Rw[2], Rw[3] ), u34, adot2( AHI( Rw[4], Rw[5] ), u56, 32 ) ) ) >> 6, maxv );
    c10 = aclip( adot2( ALO( Rw[1], Rw[2] ), u12, adot2( ALO( Rw[3], Rw[4] ), u34, adot2( ALO( Rw[5], Rw[6] ), u56, 32 ) ) ) >> 6, maxv );
    c11 = aclip( adot2( AHI( Rw[1], Rw[2] ), u12, adot2( AHI( Rw[3], Rw[4] ), u34, adot2( AHI( Rw[5], Rw[6] ), u56, 32 ) ) ) >> 6, maxv );
#undef ALO
#undef AHI
  }
  const uint32_t og0 = ald4( orgBlk + ( ptrdiff_t ) y * orgStride + x ), og1 = ald4( orgBlk + ( ptrdiff_t ) ( y + 1 ) * orgStride + x );
  const int o00 = alo( og0 ), o01 = ahi( og0 ), o10 = alo( og1 ), o11 = ahi( og1 );
  if( __builtin_amdgcn_ballot_w64( act && R.planar ) )      // MCTF.cpp:372-421 for the groups whose reference wants it (the others end with b0 = b1 = b2 = 0: unchanged)
  {
    const int z00 = c00 - o00, z01 = c01 - o01, z10 = c10 - o10, z11 = c11 - o11;
    const int x1yzm = ( int ) vvhipGroupSum32( ( uint32_t ) ( x * ( z00 + z10 ) + ( x + 1 ) * ( z01 + z11 ) ), 16, lane );
    const int x2yzm = ( int ) vvhipGroupSum32( ( uint32_t ) ( y * ( z00 + z01 ) + ( y + 1 ) * ( z10 + z11 ) ), 16, lane );
    const int ySum  = ( int ) vvhipGroupSum32( ( uint32_t ) ( z00 + z01 + z10 + z11 ), 16, lane );
    constexpr int blockSize = B * B, log2Width = 3, xSum = ( blockSize * ( B - 1 ) ) >> 1;
    const unsigned me2 = R.rmsme * R.rmsme;
    const int mWeight = ( int ) ( me2 < 512u ? me2 : 512u );
    const long long denom = ( long long ) blockSize * 336;                                        // xSzm[3]
    int b1 = planarDiv( ( long long ) mWeight * ( ( long long ) x1yzm * blockSize - xSum * ySum ), denom );
    b1 = b1 < -32768 ? -32768 : ( b1 > 32767 ? 32767 : b1 );
    int b2 = planarDiv( ( long long ) mWeight * ( ( long long ) x2yzm * blockSize - xSum * ySum ), denom );
    b2 = b2 > 32767 ? 32767 : ( b2 < -32768 ? -32768 : b2 );
    const int b0 = ( mWeight * ySum - ( b1 + b2 ) * xSum + ( blockSize >> 1 ) ) >> ( log2Width << 1 );
    if( act && R.planar && ( b0 != 0 || b1 != 0 || b2 != 0 ) )
    {
      const int p00 = b0 + b1 * x + b2 * y + 256;
      c00 = aclip( c00 - ( p00 >> 9 ), maxv );
      c01 = aclip( c01 - ( ( p00 + b1 ) >> 9 ), maxv );
      c10 = aclip( c10 - ( ( p00 + b2 ) >> 9 ), maxv );
      c11 = aclip( c11 - ( ( p00 + b1 + b2 ) >> 9 ), maxv );
    }
  }
  if( act )
  {
    int16_t* corr = corrBase + ref * corrPitch;
    *reinterpret_cast<uint32_t*>( corr + y * B + x ) = apk( c00, c01 );
    *reinterpret_cast<uint32_t*>( corr + ( y + 1 ) * B + x ) = apk( c10, c11 );
  }
  const int d00 = o00 - c00, d01 = o01 - c01, d10 = o10 - c10, d11 = o11 - c11;
  const uint32_t P0 = apk( d00, d01 ), P1 = apk( d10, d11 );
  const uint32_t nr0 = ( uint32_t ) __shfl_down( ( int ) P0, 1 ), nr1 = ( uint32_t ) __shfl_down( ( int ) P1, 1 ), nd = ( uint32_t ) __shfl_down( ( int ) P0, HP );
  uint32_t var = 0, ds = 0;
  if( act )
  {
    var = ( uint32_t ) ( d00 * d00 + d01 * d01 + d10 * d10 + d11 * d11 );
    int t;
    t = d01 - d00; ds += ( uint32_t ) ( t * t );  t = d11 - d10; ds += ( uint32_t ) ( t * t );
    t = d10 - d00; ds += ( uint32_t ) ( t * t );  t = d11 - d01; ds += ( uint32_t ) ( t * t );
    if( xp != HP - 1 ) { t = alo( nr0 ) - d01; ds += ( uint32_t ) ( t * t ); t = alo( nr1 ) - d11; ds += ( uint32_t ) ( t * t ); }
    if( rg != HP - 1 ) { t = alo( nd ) - d10; ds += ( uint32_t ) ( t * t ); t = ahi( nd ) - d11; ds += ( uint32_t ) ( t * t ); }
  }
  const uint32_t varSum = vvhipGroupSum32( var, 16, lane ), dsSum = vvhipGroupSum32( ds, 16, lane );
  long long variance = ( long long ) varSum, diffsum = ( long long ) dsSum;
  variance *= ( long long ) 1 << ( 2 * ( 10 - bitDepth ) );
  diffsum  *= ( long long ) 1 << ( 2 * ( 10 - bitDepth ) );
  constexpr int cntV = B * B, cntD = 2 * cntV - B - B;
  const int noise = ( int ) round( ( 15.0 * cntD / cntV * variance + 5.0 ) / ( diffsum + 5.0 ) );
  if( act && ( lane & 15 ) == 0 ) sNoise[ref] = noise;
  WAVE_SYNC();
}

// One WAVEFRONT per filter block, four consecutive blocks of a block row per workgroup (round 6; was one workgroup per block with a wave per reference: an 8 x 8 chroma block
// kept 256 threads, a barrier and the per-block scalar work — weights, plane fit, noise — busy for 64 samples, and a two-reference picture idled half the waves).  The wave
// walks the references one after the other (wave-level synchronisation only, private LDS scratch, DPP reductions) and blends its block with all 64 lanes.
__global__ void __launch_bounds__( 256 )
mctfApplyKernel( const int16_t* __restrict__ org, int orgStride, int refStride, int16_t* __restrict__ out, int outStride, ApplyArgs A )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sCorrAll[4][MAX_REFS][MAX_BLK * MAX_BLK / 4];     // blocks up to 16x16 (256 samples) per reference; see host check
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmpAll[4][4 * ( 8 + 5 ) * 8 + 16];      // ( 16 + 5 ) x 16 of a 16 x 16 block's reference, 4 x 13 x 8 of an 8 x 8 block's four
  __shared__ ApplyRef8 sRef8All[4][4];
  __shared__ int sNoiseAll[4][MAX_REFS], sErrAll[4][MAX_REFS];
  __shared__ float sVwwAll[4][MAX_REFS], sVswAll[4][MAX_REFS];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
  const int blk = A.blk, bxI = blockIdx.x * 4 + wave, byI = blockIdx.y;
  const int bx = bxI * blk, by = byI * blk;
  if( bx >= A.width ) return;                                                      // (no workgroup barrier below: a wave may leave)
  const int w = min( blk, A.width - bx ), h = min( blk, A.height - by );
  const int maxv = ( 1 << A.bitDepth ) - 1;
  const int16_t* orgBlk = org + ( ptrdiff_t ) by * orgStride + bx;
  int16_t* sTmp = sTmpAll[wave];
  int16_t ( *sCorr )[MAX_BLK * MAX_BLK / 4] = sCorrAll[wave];
  int* sNoise = sNoiseAll[wave]; int* sErr = sErrAll[wave];
  float* sVww = sVwwAll[wave]; float* sVsw = sVswAll[wave];

  const bool refs8 = !A.lowRes && w == 8 && h == 8 && A.bitDepth <= 10 && !A.generic;      // full 8 x 8 blocks: four references per pass (applyRefs8)
  if( refs8 )
    for( int i0 = 0; i0 < A.numRefs; i0 += 4 )
    {
      const int nr = min( 4, A.numRefs - i0 );
      for( int k = 0; k < nr; k++ )      // (uniform indices into the kernel arguments; lane 0 files the per-reference inputs)
      {
        const vvhip_mv mv = A.mvs[i0 + k][byI * A.mvW + bxI];
        const int dx = mv.x >> A.cs, dy = mv.y >> A.cs, xInt = mv.x >> ( 4 + A.cs ), yInt = mv.y >> ( 4 + A.cs );
        if( lane == 0 )
        {
          ApplyRef8 r; r.src = A.refs[i0 + k] + ( ptrdiff_t ) ( by + yInt ) * refStride + bx + xInt; r.dxF = dx & 15; r.dyF = dy & 15;
          r.rmsme = ( unsigned ) ( uint16_t ) mv.rmsme; r.planar = mv.rmsme > 0 && A.qp <= 32;
          sRef8All[wave][k] = r; sErr[i0 + k] = mv.error;
        }
      }
      WAVE_SYNC();
      applyRefs8( sRef8All[wave], nr, refStride, orgBlk, orgStride, maxv, A.bitDepth, sTmp, sCorr[i0], MAX_BLK * MAX_BLK / 4, sNoise + i0, lane );
    }
  for( int i = 0; i < ( refs8 ? 0 : A.numRefs ); i++ )      // (a uniform index: the per-reference kernel arguments are scalar loads — indexed per lane they become 100+ vector registers of copies)
  {
    const vvhip_mv mv = A.mvs[i][byI * A.mvW + bxI];
    const int dx = mv.x >> A.cs, dy = mv.y >> A.cs, xInt = mv.x >> ( 4 + A.cs ), yInt = mv.y >> ( 4 + A.cs );
    const int16_t* src = A.refs[i] + ( ptrdiff_t ) ( by + yInt ) * refStride + bx + xInt;
    int16_t* corr = sCorr[i];
    // full 16 x 16 / 8 x 8 blocks with the 6-tap filter (what the encoder runs: m_lowResFltApply is never set, MCTF.h:190): the packed path
    if( !A.lowRes && w == h && ( w == 16 || w == 8 ) && A.bitDepth <= 10 && !A.generic )
    {
      int noise = 0;
      const bool planar = mv.rmsme > 0 && A.qp <= 32;
      if( w == 16 ) applyRefFull<16>( src, refStride, orgBlk, orgStride, dx & 15, dy & 15, maxv, ( unsigned ) ( uint16_t ) mv.rmsme, planar, A.bitDepth, sTmp, corr, noise, lane );
      else          applyRefFull<8>( src, refStride, orgBlk, orgStride, dx & 15, dy & 15, maxv, ( unsigned ) ( uint16_t ) mv.rmsme, planar, A.bitDepth, sTmp, corr, noise, lane );
      if( lane == 0 ) { sNoise[i] = noise; sErr[i] = mv.error; }
      continue;
    }
    // ---- applyFrac: horizontal pass into sTmp (Pel truncation, no clip), vertical pass into corr (clip)
    if( A.lowRes )
    {
      const int16_t* xf = cApply4[dx & 15]; const int16_t* yf = cApply4[dy & 15];
      for( int e = lane; e < ( h + 3 ) * w; e += 64 )
      {
        const int r = e / w, x = e - r * w;
        const int16_t* p = src + ( ptrdiff_t ) ( r - 1 ) * refStride + x - 1;
        const int sum = xf[0] * p[0] + xf[1] * p[1] + xf[2] * p[2] + xf[3] * p[3];
        sTmp[r * w + x] = ( int16_t ) ( ( sum + 32 ) >> 6 );
      }
      WAVE_SYNC();
      for( int e = lane; e < h * w; e += 64 )
      {
        const int y = e / w, x = e - y * w;
        const int sum = yf[0] * sTmp[y * w + x] + yf[1] * sTmp[( y + 1 ) * w + x] + yf[2] * sTmp[( y + 2 ) * w + x] + yf[3] * sTmp[( y + 3 ) * w + x];
        const int v = ( sum + 32 ) >> 6;
        corr[e] = ( int16_t ) ( v < 0 ? 0 : ( v > maxv ? maxv : v ) );
      }
    }
    else
    {
      const int16_t* xf = cApply6[dx & 15]; const int16_t* yf = cApply6[dy & 15];
      for( int e = lane; e < ( h + 5 ) * w; e += 64 )       // rows 1 .. h+5 of the reference's temp array <-> source rows -2 .. h+2
      {
        const int r = e / w + 1, x = e - ( r - 1 ) * w;
        const int16_t* p = src + ( ptrdiff_t ) ( r - 3 ) * refStride + x - 3;
        int sum = 0;
#pragma unroll
        for( int k = 1; k <= 6; k++ ) sum += xf[k] * p[k];
        sTmp[r * w + x] = ( int16_t ) ( ( sum + 32 ) >> 6 );
      }
      WAVE_SYNC();
      for( int e = lane; e < h * w; e += 64 )
      {
        const int y = e / w, x = e - y * w;
        int sum = 0;
#pragma unroll
        for( int k = 1; k <= 6; k++ ) sum += yf[k] * sTmp[( y + k ) * w + x];
        const int v = ( sum + 32 ) >> 6;
        corr[e] = ( int16_t ) ( v < 0 ? 0 : ( v > maxv ? maxv : v ) );
      }
    }
    WAVE_SYNC();
    // ---- planar correction of the compensated block (MCTF.cpp:1473-1476): every lane evaluates the (wave-uniform) plane parameters
    if( mv.rmsme > 0 && A.qp <= 32 && w == h && w <= 32 )
    {
      int x1 = 0, x2 = 0, ys = 0;
      for( int e = lane; e < h * w; e += 64 )
      {
        const int y = e / w, x = e - y * w;
        const int z = ( int ) corr[e] - ( int ) orgBlk[( ptrdiff_t ) y * orgStride + x];
        x1 += x * z; x2 += y * z; ys += z;
      }
      // the block sums fit int32 (|z| < 2^12, x < 32, <= 1024 samples): modular 32-bit wave sums are exact
      const int x1yzm = ( int ) vvhipGroupSum32( ( uint32_t ) x1, 64, lane ), x2yzm = ( int ) vvhipGroupSum32( ( uint32_t ) x2, 64, lane ), ySum = ( int ) vvhipGroupSum32( ( uint32_t ) ys, 64, lane );
      const int xSzm[6] = { 0, 1, 20, 336, 5440, 87296 };
      const int blockSize = w * h; int log2Width = 0; while( ( 2 << log2Width ) <= w ) log2Width++;
      const unsigned me2 = ( unsigned ) ( uint16_t ) mv.rmsme * ( unsigned ) ( uint16_t ) mv.rmsme;
      const int mWeight = ( int ) ( me2 < 512u ? me2 : 512u );
      const int xSum = ( blockSize * ( w - 1 ) ) >> 1;
      const long long denom = ( long long ) blockSize * xSzm[log2Width];
      long long numer = ( long long ) mWeight * ( ( long long ) x1yzm * blockSize - xSum * ySum );
      int b1 = planarDiv( numer, denom );
      b1 = b1 < -32768 ? -32768 : ( b1 > 32767 ? 32767 : b1 );
      numer = ( long long ) mWeight * ( ( long long ) x2yzm * blockSize - xSum * ySum );
      int b2 = planarDiv( numer, denom );
      b2 = b2 > 32767 ? 32767 : ( b2 < -32768 ? -32768 : b2 );
      const int b0 = ( mWeight * ySum - ( b1 + b2 ) * xSum + ( blockSize >> 1 ) ) >> ( log2Width << 1 );
      if( b0 != 0 || b1 != 0 || b2 != 0 )
        for( int e = lane; e < h * w; e += 64 )
        {
          const int y = e / w, x = e - y * w;
          const int p = ( b0 + b1 * x + b2 * y + 256 ) >> 9;
          const int z = ( int ) corr[e] - p;
          corr[e] = ( int16_t ) ( z < 0 ? 0 : ( z > maxv ? maxv : z ) );
        }
      WAVE_SYNC();
    }
    // ---- noise estimate of reference i (MCTF.cpp:445-477)
    {
      unsigned long long variance = 0, diffsum = 0;
      for( int e = lane; e < h * w; e += 64 )
      {
        const int y = e / w, x = e - y * w;
        const int diff = ( int ) orgBlk[( ptrdiff_t ) y * orgStride + x] - ( int ) corr[e];
        variance += ( unsigned ) ( diff * diff );
        if( x != w - 1 ) { const int dR = ( int ) orgBlk[( ptrdiff_t ) y * orgStride + x + 1] - ( int ) corr[e + 1]; diffsum += ( unsigned ) ( ( dR - diff ) * ( dR - diff ) ); }
        if( y != h - 1 ) { const int dD = ( int ) orgBlk[( ptrdiff_t ) ( y + 1 ) * orgStride + x] - ( int ) corr[e + w]; diffsum += ( unsigned ) ( ( dD - diff ) * ( dD - diff ) ); }
      }
      long long var = ( long long ) vvhipGroupSum64( variance, 64, lane ), dsum = ( long long ) vvhipGroupSum64( diffsum, 64, lane );      // per-lane values < 2^50
      if( lane == 0 )
      {
        var  *= ( long long ) 1 << ( 2 * ( 10 - A.bitDepth ) );
        dsum *= ( long long ) 1 << ( 2 * ( 10 - A.bitDepth ) );
        const int cntV = w * h, cntD = 2 * cntV - w - h;
        sNoise[i] = ( int ) round( ( 15.0 * cntD / cntV * var + 5.0 ) / ( dsum + 5.0 ) );
        sErr[i] = mv.error;
      }
    }
  }
  WAVE_SYNC();

  // ---- per-reference weights (MCTF.cpp:479-500): thread i evaluates reference i's pair once (round 6: every thread used to evaluate all of them, 12 unrolled slots of
  //      double arithmetic and 24 live registers), the blend reads them as LDS broadcasts
  {
    int minError = 0x7fffffff;
    for( int i = 0; i < A.numRefs; i++ ) minError = sErr[i] < minError ? sErr[i] : minError;
    for( int i = 0; i < A.numRefs; i++ )      // (uniform index into the kernel arguments; lane i keeps reference i's pair)
    {
      const int error = sErr[i], noise = sNoise[i];
      float ww = 1, sw = 1;
      ww *= ( noise < 25 ) ? 1.0 : 0.6;
      sw *= ( noise < 25 ) ? 1.0 : 0.8;
      ww *= ( error < 50 ) ? 1.2 : ( ( error > 100 ) ? 0.6 : 1.0 );
      sw *= ( error < 50 ) ? 1.0 : 0.8;
      ww *= ( ( minError + 1.0 ) / ( error + 1.0 ) );
      const float vw = ww * A.weightScaling * A.refStrengths[i], vs = sw * 2 * A.sigmaSq;
      if( lane == i ) { sVww[i] = vw; sVsw[i] = vs; }
    }
  }
  WAVE_SYNC();
  // ---- the blend (MCTF.cpp:501-517)
  const int nRefs = A.numRefs;
  for( int e = lane; e < h * w; e += 64 )
  {
    const int y = e / w, x = e - y * w;
    const int16_t orgVal = orgBlk[( ptrdiff_t ) y * orgStride + x];
    float temporalWeightSum = 1.0;
    float newVal = ( float ) orgVal;
    for( int i = 0; i < nRefs; i++ )
    {
      const int refVal = sCorr[i][e];
      const int diff = refVal - orgVal;
      const float diffSq = diff * diff;
      const float weight = sVww[i] * fastExp( -diffSq, sVsw[i] );
      newVal += weight * refVal;
      temporalWeightSum += weight;
    }
    newVal /= temporalWeightSum;
    int16_t sampleVal = ( int16_t ) ( newVal + 0.5 );
    sampleVal = sampleVal < 0 ? ( int16_t ) 0 : ( sampleVal > maxv ? ( int16_t ) maxv : sampleVal );
    out[( ptrdiff_t ) ( by + y ) * outStride + bx + x] = sampleVal;
  }
}

} // namespace

extern "C" {

int vvhip_mctf_apply_plane( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, int width, int height, int chroma_shift, int bit_depth, int unit_size,
                            int low_res_flt_apply, int qp, int num_refs, const int16_t* const* d_refs, int ref_stride, const vvhip_mv* const* d_mvs, int mv_w,
                            const double* ref_strengths, double weight_scaling, double sigma_sq, int16_t* d_out, int out_stride )
{
  if( !ctx ) return VVHIP_E_ARG;
  const int blk = unit_size >> chroma_shift;
  if( !d_org || !d_out || !d_refs || !d_mvs || !ref_strengths || width < 1 || height < 1 || chroma_shift < 0 || chroma_shift > 1 || bit_depth < 8 || bit_depth > 12 ||
      num_refs < 1 || num_refs > MAX_REFS || blk < 4 || blk > 16 || ( blk & ( blk - 1 ) ) || mv_w < ( width + blk - 1 ) / blk )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_mctf_apply_plane: %dx%d unit %d shift %d refs %d (unit >> shift must be 4, 8 or 16; <= %d references)", width, height, unit_size,
                       chroma_shift, num_refs, MAX_REFS );
  ApplyArgs a;
  for( int i = 0; i < MAX_REFS; i++ ) { a.refs[i] = i < num_refs ? d_refs[i] : nullptr; a.mvs[i] = i < num_refs ? d_mvs[i] : nullptr; a.refStrengths[i] = i < num_refs ? ref_strengths[i] : 0.0; }
  a.weightScaling = weight_scaling; a.sigmaSq = sigma_sq; a.numRefs = num_refs; a.cs = chroma_shift; a.bitDepth = bit_depth; a.blk = blk; a.lowRes = low_res_flt_apply ? 1 : 0;
  a.qp = qp; a.mvW = mv_w; a.width = width; a.height = height;
  static const int generic = []{ const char* e = getenv( "VVHIP_MCTF_APPLY_GENERIC" ); return e ? atoi( e ) : 0; }();
  a.generic = generic;
  const dim3 grid( ( ( width + blk - 1 ) / blk + 3 ) / 4, ( height + blk - 1 ) / blk );      // four blocks of a block row per workgroup, one per wavefront
  hipLaunchKernelGGL( mctfApplyKernel, grid, dim3( 256 ), 0, ctx->stream, d_org, org_stride, ref_stride, d_out, out_stride, a );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_mctf_filter_params( int qp, int bit_depth, double overall_strength, int is_chroma, double* sigma_sq, double* weight_scaling )
{
  if( !sigma_sq || !weight_scaling || bit_depth < 8 || bit_depth > 16 ) return VVHIP_E_ARG;
  const double lumaSigmaSq = 9.0 * ( 128.0 + 3.0 / 256.0 * qp * qp * qp );          // MCTF.cpp:68,1491 (m_sigmaMultiplier 9.0)
  const double chromaSigmaSq = 30 * 30;
  const double bitDepthDiffWeighting = 1024.0 / ( double ) ( 1 << bit_depth );      // :1498-1499
  *sigma_sq = ( is_chroma ? chromaSigmaSq : lumaSigmaSq ) / ( bitDepthDiffWeighting * bitDepthDiffWeighting );
  *weight_scaling = overall_strength * ( is_chroma ? 0.55 : 0.4 );                  // :67,1417 (m_chromaFactor 0.55)
  return VVHIP_OK;
}

} // extern "C"
