// trquant.hip — forward / inverse DCT-2, DST-7, DCT-8 and scalar (de)quantisation for gfx950.
//
// Reference semantics
//   forward 2-D   TrQuant::xT            CommonLib/TrQuant.cpp:481-564  ->  _fastForwardMM / fastFwdCore   TrQuant_EMT.cpp:366-420,1973-2000  (x86: fastFwd_SSE  x86/TrafoX86.h:310-643)
//   inverse 2-D   TrQuant::xIT           CommonLib/TrQuant.cpp:567-655  ->  _fastInverseMM / fastInvCore_  TrQuant_EMT.cpp:152-194,1953-1970 + clipCore :1941 (x86: fastInv_SSE :60-308)
//   quantiser     Quant::quant           CommonLib/Quant.cpp:735-833    ->  QuantCore   :132-230
//   dequantiser   Quant::dequant         CommonLib/Quant.cpp:520-610    ->  DeQuantCore :232-262
//   RDOQ pre-test Quant::xNeedRDOQ       CommonLib/Quant.cpp:835-891    ->  needRdoqCore :264-278
// The N=2/4/8 butterflies of the reference compute the same integer sums as the matrix form, so one matrix-form kernel
// covers every size and type.
//
// Arithmetic: every 1-D pass is int16 x int16 -> int32 (v_dot2_i32_i16, two MACs per lane per instruction).  The operands
// of a pass are 16-bit by construction: residuals are Pel, the kernel matrices are 8-bit, the forward intermediate fits 16 bits
// for every residual of bitDepth-bit samples (|r| <= 2^bitDepth  =>  |tmp| <= 64*2^9 = 32768-ish, DC row bound), dequantised
// coefficients and the inverse intermediate are clipped to [-32768, 32767] by the reference itself.  For inputs outside that
// contract the passes saturate their *input* to 16 bits exactly like the reference's x86 row does (_mm_packs_epi32,
// x86/TrafoX86.h:101,364) — i.e. results equal the reference's SIMD path for all inputs and its scalar path for conforming ones.
//
// LDS layout (per TU, int16 unless noted; "pitch" = row length rounded to an odd number of 16-byte chunks so that consecutive
// lanes reading consecutive rows with ds_read_b128 hit disjoint banks):
//   R0  residual rows [h][pw]            | later: dequantised coefficients, transposed [w][ph]
//   R1  forward intermediate [w][ph]     | later: levels (raster) -> HBM | later: inverse intermediate [h][pw]
//   R2  coefficients int32 [h][w]        | later: reconstructed residual (raster int16) -> HBM
// In every pass consecutive lanes own consecutive ROWS of the data operand (each lane streams its row with 16-byte reads)
// while the matrix row is a wave-wide LDS broadcast.  Nothing but the residual, the levels, the reconstruction and 24 bytes
// of statistics per TU crosses HBM in the fused kernel.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {

typedef uint32_t u32x2 __attribute__( ( ext_vector_type( 2 ) ) );
typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
typedef short    s16x2 __attribute__( ( ext_vector_type( 2 ) ) );

__device__ __forceinline__ int clip3i( int lo, int hi, int v ) { return v < lo ? lo : ( v > hi ? hi : v ); }
__device__ __forceinline__ int sat16( int v ) { return clip3i( -32768, 32767, v ); }
__device__ __forceinline__ int dot2( uint32_t a, uint32_t b, int acc )
{
  return __builtin_amdgcn_sdot2( __builtin_bit_cast( s16x2, a ), __builtin_bit_cast( s16x2, b ), acc, false );
}

struct TrGeom
{
  int w, h, log2w, log2h;
  int skipW, skipH;          // TrQuant.cpp:496-497 / :587-588 (LFNST off)
  int shift1, shift2;
};

// LDS carve-up, all offsets in int16 elements
struct TuLay
{
  int pw, ph;                // row pitch of rows of length w / h
  int r1, r2, slot;          // region offsets inside a TU slot (R0 at 0) and slot size
  int mTh, mTv, mThT, mTvT;  // matrix offsets from the matrix base (natural [freq][sample] and transposed [sample][freq])
  int matElems;
};

// sum_k a[k] * b[k], k < n; a and b aligned to min(n,8)*2 bytes
__device__ __forceinline__ int dotRow( const int16_t* a, const int16_t* b, int n )
{
  int acc = 0;
  if( n >= 8 )
  {
    for( int k = 0; k < n; k += 8 )
    {
      const u32x4 x = *reinterpret_cast<const u32x4*>( a + k ), y = *reinterpret_cast<const u32x4*>( b + k );
      acc = dot2( x.x, y.x, acc ); acc = dot2( x.y, y.y, acc ); acc = dot2( x.z, y.z, acc ); acc = dot2( x.w, y.w, acc );
    }
  }
  else if( n == 4 )
  {
    const u32x2 x = *reinterpret_cast<const u32x2*>( a ), y = *reinterpret_cast<const u32x2*>( b );
    acc = dot2( x.x, y.x, acc ); acc = dot2( x.y, y.y, acc );
  }
  else acc = dot2( *reinterpret_cast<const uint32_t*>( a ), *reinterpret_cast<const uint32_t*>( b ), 0 );
  return acc;
}

struct Lds
{
  int16_t* tu;         // TU slots
  const int16_t* mat;  // matrices
};

// ---- forward: R0 = residual rows  ->  R2 = coefficients (int32 raster, zero-out applied) -----------------------------------
__device__ __forceinline__ void fwd2d( const TrGeom& g, const TuLay& y, const Lds& L, int nTu, int tid, int nthr )
{
  const int w = g.w, h = g.h, la = g.log2w + g.log2h, area = 1 << la;
  const int cutW = w - g.skipW, cutH = h - g.skipH;
  const int rnd1 = g.shift1 > 0 ? 1 << ( g.shift1 - 1 ) : 0, rnd2 = 1 << ( g.shift2 - 1 );
  // pass 1 (rows): tmp[j][i] = sat16( ( sum_k blk[i][k] * Th[j][k] + rnd ) >> shift1 ), j < w - skipW     (TrQuant.cpp:548)
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 ), i = p & ( h - 1 ), j = p >> g.log2h;
    int16_t* s = L.tu + t * y.slot;
    int v = 0;
    if( j < cutW ) v = sat16( ( int ) ( ( uint32_t ) dotRow( s + i * y.pw, L.mat + y.mTh + j * w, w ) + ( uint32_t ) rnd1 ) >> g.shift1 );
    s[y.r1 + j * y.ph + i] = ( int16_t ) v;
  }
  __syncthreads();
  // pass 2 (columns): coef[j2][i2] = ( sum_k tmp[i2][k] * Tv[j2][k] + rnd ) >> shift2, i2 < w - skipW, j2 < h - skipH   (TrQuant.cpp:549)
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 ), i2 = p & ( w - 1 ), j2 = p >> g.log2w;
    int16_t* s = L.tu + t * y.slot;
    int v = 0;
    if( i2 < cutW && j2 < cutH ) v = ( int ) ( ( uint32_t ) dotRow( s + y.r1 + i2 * y.ph, L.mat + y.mTv + j2 * h, h ) + ( uint32_t ) rnd2 ) >> g.shift2;
    reinterpret_cast<int32_t*>( s + y.r2 )[p] = v;
  }
  __syncthreads();
}

// ---- inverse: R0 = coefficients transposed [i][k] (int16)  ->  R2 = residual (raster int16) ---------------------------------
__device__ __forceinline__ void inv2d( const TrGeom& g, const TuLay& y, const Lds& L, int nTu, int tid, int nthr )
{
  const int w = g.w, h = g.h, la = g.log2w + g.log2h, area = 1 << la;
  const int cutW = w - g.skipW;
  const int rnd1 = 1 << ( g.shift1 - 1 ), rnd2 = 1 << ( g.shift2 - 1 );
  // pass 1 (columns): t1[j][i] = clip( ( sum_k coef[k][i] * Tv[k][j] + rnd ) >> shift1 ), i < w - skipW else 0           (TrQuant.cpp:612)
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 ), i = p & ( w - 1 ), j = p >> g.log2w;
    int16_t* s = L.tu + t * y.slot;
    int v = 0;
    if( i < cutW ) v = sat16( ( int ) ( ( uint32_t ) dotRow( s + i * y.ph, L.mat + y.mTvT + j * h, h ) + ( uint32_t ) rnd1 ) >> g.shift1 );
    s[y.r1 + j * y.pw + i] = ( int16_t ) v;
  }
  __syncthreads();
  // pass 2 (rows): rec[i2][j2] = clip( ( sum_k t1[i2][k] * Th[k][j2] + rnd ) >> shift2 )                                  (TrQuant.cpp:613)
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 ), i2 = p & ( h - 1 ), j2 = p >> g.log2h;
    int16_t* s = L.tu + t * y.slot;
    const int v = sat16( ( int ) ( ( uint32_t ) dotRow( s + y.r1 + i2 * y.pw, L.mat + y.mThT + j2 * w, w ) + ( uint32_t ) rnd2 ) >> g.shift2 );
    s[y.r2 + i2 * w + j2] = ( int16_t ) v;
  }
  __syncthreads();
}

__device__ __forceinline__ Lds carve( unsigned char* raw, const TrGeom& g, const TuLay& y, int tpb, const int16_t* matH, const int16_t* matV, bool needInv, bool needFwd, int tid, int nthr )
{
  Lds L;
  int16_t* m = reinterpret_cast<int16_t*>( raw );
  L.mat = m;
  L.tu = m + ( ( y.matElems + 7 ) & ~7 );
  const int w = g.w, h = g.h;
  if( needFwd )
  {
    for( int i = tid; i < w * w; i += nthr ) m[y.mTh + i] = matH[i];
    if( y.mTv != y.mTh ) for( int i = tid; i < h * h; i += nthr ) m[y.mTv + i] = matV[i];
  }
  if( needInv )
  {
    for( int i = tid; i < w * w; i += nthr ) { const int k = i >> g.log2w, j = i & ( w - 1 ); m[y.mThT + j * w + k] = matH[i]; }
    if( y.mTvT != y.mThT ) for( int i = tid; i < h * h; i += nthr ) { const int k = i >> g.log2h, j = i & ( h - 1 ); m[y.mTvT + j * h + k] = matV[i]; }
  }
  return L;
}

// residual rows -> R0 (cpyCoeff, TrQuant_EMT.cpp:1917-1926); 16-byte global loads when w >= 8
__device__ __forceinline__ void loadResi( const TrGeom& g, const TuLay& y, const Lds& L, const int16_t* resi, int resiStride, const int32_t* resiOff,
                                          int tu0, int nTu, int tid, int nthr )
{
  const int w = g.w, la = g.log2w + g.log2h, area = 1 << la;
  if( w >= 8 )
  {
    struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
    const int lc = la - 3;                                  // chunks of 8 samples per TU
    for( int o = tid; o < ( nTu << lc ); o += nthr )
    {
      const int t = o >> lc, c = o & ( ( 1 << lc ) - 1 ), yy = c >> ( g.log2w - 3 ), x = ( c & ( ( w >> 3 ) - 1 ) ) << 3;
      const u32x4 v = reinterpret_cast<const U16*>( resi + resiOff[tu0 + t] + ( ptrdiff_t ) yy * resiStride + x )->v;
      *reinterpret_cast<u32x4*>( L.tu + t * y.slot + yy * y.pw + x ) = v;
    }
  }
  else
    for( int o = tid; o < ( nTu << la ); o += nthr )
    {
      const int t = o >> la, p = o & ( area - 1 ), yy = p >> g.log2w, x = p & ( w - 1 );
      L.tu[t * y.slot + yy * y.pw + x] = resi[resiOff[tu0 + t] + ( ptrdiff_t ) yy * resiStride + x];
    }
}

__global__ void __launch_bounds__( 256 )
fwdTransformKernel( const int16_t* __restrict__ resi, int resiStride, const int32_t* __restrict__ resiOff, int n,
                    TrGeom g, TuLay y, int tpb, const int16_t* __restrict__ matH, const int16_t* __restrict__ matV,
                    int32_t* __restrict__ coef )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  const int tid = threadIdx.x, nthr = blockDim.x, la = g.log2w + g.log2h, area = 1 << la;
  const int tu0 = blockIdx.x * tpb, nTu = min( tpb, n - tu0 );
  const Lds L = carve( smemRaw, g, y, tpb, matH, matV, false, true, tid, nthr );
  loadResi( g, y, L, resi, resiStride, resiOff, tu0, nTu, tid, nthr );
  __syncthreads();
  fwd2d( g, y, L, nTu, tid, nthr );
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 );
    coef[( size_t ) tu0 * area + o] = reinterpret_cast<const int32_t*>( L.tu + t * y.slot + y.r2 )[p];
  }
}

__global__ void __launch_bounds__( 256 )
invTransformKernel( const int32_t* __restrict__ coef, int n, TrGeom g, TuLay y, int tpb,
                    const int16_t* __restrict__ matH, const int16_t* __restrict__ matV,
                    int16_t* __restrict__ resi, int resiStride, const int32_t* __restrict__ resiOff )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  const int tid = threadIdx.x, nthr = blockDim.x, w = g.w, h = g.h, la = g.log2w + g.log2h, area = 1 << la;
  const int tu0 = blockIdx.x * tpb, nTu = min( tpb, n - tu0 );
  const Lds L = carve( smemRaw, g, y, tpb, matH, matV, true, false, tid, nthr );
  const int cutH = h - g.skipH;
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 ), k = p >> g.log2w, i = p & ( w - 1 );       // coefficient row k (vertical frequency), column i
    const int v = k < cutH ? sat16( coef[( size_t ) tu0 * area + o] ) : 0;                 // rows >= cutoff are never read by the reference (TrQuant_EMT.cpp:165)
    L.tu[t * y.slot + i * y.ph + k] = ( int16_t ) v;
  }
  __syncthreads();
  inv2d( g, y, L, nTu, tid, nthr );
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 ), yy = p >> g.log2w, x = p & ( w - 1 );
    resi[resiOff[tu0 + t] + ( ptrdiff_t ) yy * resiStride + x] = L.tu[t * y.slot + y.r2 + p];   // cpyResi, TrQuant_EMT.cpp:1929-1938
  }
}

// --------------------------------------------------------------------------------------------
// Quantiser parameter derivation (device copy of Quant.cpp:769-775, :554-561/:601-607, :852-874)
// --------------------------------------------------------------------------------------------
__constant__ int cQuantScales[2][6]    = { { 26214, 23302, 20560, 18396, 16384, 14564 }, { 18396, 16384, 14564, 13107, 11651, 10280 } };  // Rom.cpp:1390-1394
__constant__ int cInvQuantScales[2][6] = { { 40, 45, 51, 57, 64, 72 }, { 57, 64, 72, 80, 90, 102 } };                                     // Rom.cpp:1396-1400

struct QGeom { int w, h, log2w, log2h, bitDepth, log2CG, cgIs4x4, numScan /* scan positions inside the 32x32 zero-out region */; };

__device__ __forceinline__ void quantParams( const QGeom& q, int qp, int& scale, int& qBits )
{
  const int l = q.log2w + q.log2h, sqrt2 = l & 1;
  const int trShift = 15 - q.bitDepth - ( l >> 1 ) - sqrt2;
  scale = cQuantScales[sqrt2][qp % 6];
  qBits = 14 + qp / 6 + trShift;
}

// One team of LPC lanes per TU.  Scan positions are strided over the team.
__global__ void __launch_bounds__( 256 )
quantKernel( const int32_t* __restrict__ coef, int n, QGeom q, int log2Lpc, const vvhip_tu_qp* __restrict__ qps, int rawScale, int rawQBits, long long rawAdd, int thrVal,
             const uint16_t* __restrict__ scan, int16_t* __restrict__ level, int32_t* __restrict__ deltaU,
             int32_t* __restrict__ absSumOut, int32_t* __restrict__ lastPosOut, int lfnst = 0 )
{
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lpc = 1 << log2Lpc, tu = gid >> log2Lpc, lt = gid & ( lpc - 1 );
  const bool valid = tu < n;
  const int area = q.w * q.h;
  const int32_t* src = coef + ( size_t ) ( valid ? tu : 0 ) * area;
  int16_t* dst = level + ( size_t ) ( valid ? tu : 0 ) * area;
  int scale = 1, qBits = 16;
  int64_t add = 0;
  if( valid && qps )
  {
    const vvhip_tu_qp p = qps[tu];
    quantParams( q, p.qp, scale, qBits );
    add = ( int64_t ) ( ( p.flags & 1 ) ? 171 : 85 ) << ( qBits - 9 );               // Quant.cpp:775
  }
  else if( valid ) { scale = rawScale; qBits = rawQBits; add = rawAdd; }              // table-entry form: QuantCore's own argument list
  const int num = valid ? q.numScan : 0;

  // (1) last non-zero scan position (Quant.cpp:162-167); 0 if none.  LFNST TUs (CodingUnit::lfnstIdx > 0): only the first coefficient group is looked at, its first 8 positions
  //     for 4x4 and 8x8 TUs (iCGNum = 1, :149-159) — everything behind stays zero, and the coefficient-group test (2) has nothing to test (last < 16)
  const int numLast = !lfnst ? num : min( num, ( ( q.w == 4 && q.h == 4 ) || ( q.w == 8 && q.h == 8 ) ) ? 8 : ( 1 << q.log2CG ) );
  int last = 0;
  for( int p = lt; p < numLast; p += lpc ) if( src[scan[p]] != 0 ) last = p;         // p increases -> keeps the largest
  for( int o = lpc >> 1; o > 0; o >>= 1 ) last = max( last, __shfl_xor( last, o ) );

  // (2) coefficient-group early zero-out (Quant.cpp:173-208): only for 4x4 CGs, only CGs >= 1
  if( q.cgIs4x4 && last >= 16 )
  {
    const int32_t thres = qBits ? ( int32_t ) ( ( int64_t ) thrVal << ( qBits - 1 ) ) : ( int32_t ) ( ( int64_t ) ( thrVal >> 1 ) << qBits );
    const int32_t useThres = thres / ( scale << 2 );
    uint32_t bigLo = 0, bigHi = 0;                                                    // bit g: CG g holds a |coef| > useThres
    for( int p = lt; p <= last; p += lpc )
      if( abs( src[scan[p]] ) > useThres ) { const int cg = p >> 4; if( cg < 32 ) bigLo |= 1u << cg; else bigHi |= 1u << ( cg - 32 ); }
    for( int o = lpc >> 1; o > 0; o >>= 1 ) { bigLo |= __shfl_xor( bigLo, o ); bigHi |= __shfl_xor( bigHi, o ); }
    const uint64_t big = ( ( ( uint64_t ) bigHi << 32 ) | bigLo ) & ~1ull;          // CG 0 is never tested
    const int topCg = last >> 4;
    if( big == 0 ) last = 15;
    else
    {
      const int g = 63 - __clzll( ( long long ) big );
      if( g != topCg ) last = g * 16 + 15;
    }
  }

  // (3) quantise scan positions 0..last (Quant.cpp:213-227); everything else is zero.  Every output sample has exactly
  //     one owner lane: scan position p owns raster position scan[p]; the zero-out region (x >= 32 or y >= 32) is owned by raster index.
  uint32_t absSum = 0;
  for( int p = lt; p < num; p += lpc )
  {
    const int bp = scan[p];
    int16_t lv = 0;
    if( p <= last )
    {
      const int32_t c = src[bp];
      const int64_t t = ( int64_t ) abs( c ) * scale;
      const int32_t m = ( int32_t ) ( ( t + add ) >> qBits );
      if( deltaU ) deltaU[( size_t ) tu * area + bp] = ( int32_t ) ( ( t - ( ( int64_t ) m << qBits ) ) >> ( qBits - 8 ) );
      absSum += ( uint32_t ) m;
      lv = ( int16_t ) clip3i( -32768, 32767, c < 0 ? -m : m );
    }
    dst[bp] = lv;
  }
  if( q.w > 32 || q.h > 32 )
    for( int i = lt; i < ( valid ? area : 0 ); i += lpc )
    {
      const int y = i >> q.log2w, x = i & ( q.w - 1 );
      if( x >= 32 || y >= 32 ) dst[i] = 0;
    }
  for( int o = lpc >> 1; o > 0; o >>= 1 ) absSum += __shfl_xor( absSum, o );
  if( valid && lt == 0 ) { absSumOut[tu] = ( int32_t ) absSum; lastPosOut[tu] = last; }
}

__global__ void __launch_bounds__( 256 )
dequantKernel( const int16_t* __restrict__ level, long total, QGeom q, const vvhip_tu_qp* __restrict__ qps, int32_t* __restrict__ coef )
{
  const int area = q.w * q.h;
  for( long i = ( long ) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += ( long ) gridDim.x * blockDim.x )
  {
    const int tu = ( int ) ( i / area );
    const int qp = qps[tu].qp;
    const int l = q.log2w + q.log2h, sqrt2 = l & 1;
    const int trShift = 15 - q.bitDepth - ( l >> 1 ) - sqrt2;
    const int scale = cInvQuantScales[sqrt2][qp % 6];
    const int rightShift = 6 - ( trShift + qp / 6 );                                 // Quant.cpp:561
    int tgt = 32 + rightShift - 7; if( tgt > 16 ) tgt = 16;                           // Quant.cpp:606
    const int inMax = ( 1 << ( tgt - 1 ) ) - 1, inMin = -( inMax + 1 );
    const int c = clip3i( inMin, inMax, ( int ) level[i] );
    int32_t v;
    if( rightShift > 0 ) v = ( int32_t ) ( ( uint32_t ) ( c * scale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;   // Quant.cpp:244
    else                 v = ( int32_t ) ( ( uint32_t ) ( c * scale ) << ( -rightShift ) );                           // Quant.cpp:257
    coef[i] = clip3i( -32768, 32767, v );
  }
}

__global__ void __launch_bounds__( 256 )
needRdoqKernel( const int32_t* __restrict__ coef, int n, QGeom q, int log2Lpc, const vvhip_tu_qp* __restrict__ qps, uint8_t* __restrict__ need )
{
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lpc = 1 << log2Lpc, tu = gid >> log2Lpc, lt = gid & ( lpc - 1 );
  const bool valid = tu < n;
  const int area = q.w * q.h;
  const int num = valid ? q.w * min( q.h, 32 ) : 0;                                   // efArea, Quant.cpp:841-842
  int scale = 1, qBits = 16;
  int64_t add = 0;
  if( valid )
  {
    const vvhip_tu_qp p = qps[tu];
    quantParams( q, p.qp, scale, qBits );
    add = ( int64_t ) ( ( p.flags & 2 ) ? 171 : 256 ) << ( qBits - 9 );              // Quant.cpp:874
  }
  const int32_t* src = coef + ( size_t ) ( valid ? tu : 0 ) * area;
  int any = 0;
  for( int i = lt; i < num; i += lpc )
  {
    const int64_t t = ( int64_t ) abs( src[i] ) * scale;
    any |= ( int32_t ) ( ( t + add ) >> qBits ) != 0;
  }
  for( int o = lpc >> 1; o > 0; o >>= 1 ) any |= __shfl_xor( any, o );
  if( valid && lt == 0 ) need[tu] = ( uint8_t ) any;
}

// --------------------------------------------------------------------------------------------
// Fused TU pipeline (InterSearch::xEstimateInterResidualQT inner sequence, EncoderLib/InterSearch.cpp:3663-3714):
//   xT -> xNeedRDOQ -> QuantCore -> DeQuantCore -> xIT -> SSE( residual, reconstructed residual )
// A workgroup owns ~1024 samples worth of TUs.  The significance structure of QuantCore (last non-zero scan position,
// per-coefficient-group threshold test) is gathered with ONE wave ballot per 64 scan positions — no shuffles, no atomics — and
// resolved by one lane per TU; sums (abs-sum, SSE) use the hardware wave reduction.
// --------------------------------------------------------------------------------------------
struct TuPar   // per-TU quantiser constants + reduction targets, in LDS
{
  int scale, qBits, iscale, rightShift, inMax, useThres, last; uint32_t absSum;
  long long add, addN; unsigned long long sse; uint32_t need, pad;
  unsigned long long nz[16], big[16];
};

__device__ __forceinline__ unsigned long long grpAdd64( unsigned long long e, int G )
{
  uint32_t lo = ( uint32_t ) e, hi = ( uint32_t ) ( e >> 32 );
  for( int sft = G >> 1; sft > 0; sft >>= 1 )
  {
    const unsigned long long other = ( ( unsigned long long ) __shfl_xor( hi, sft ) << 32 ) | __shfl_xor( lo, sft );
    e += other; lo = ( uint32_t ) e; hi = ( uint32_t ) ( e >> 32 );
  }
  return e;
}

// (the body of tuRdoKernel / tuRdoGenMultiKernel: workgroup `blk` of one job's launch geometry)
__device__ __forceinline__ void
tuRdoBody( unsigned char* smemRaw, const int blk, const int16_t* __restrict__ resi, int resiStride, const int32_t* __restrict__ resiOff, int n,
           const TrGeom& gf, const TrGeom& gi, const TuLay& y, const QGeom& q, int tpb, const int16_t* __restrict__ matH, const int16_t* __restrict__ matV,
           const uint16_t* __restrict__ scan, const vvhip_tu_qp* __restrict__ qps, int thrVal,
           int16_t* __restrict__ level, int16_t* __restrict__ rec, vvhip_tu_stats* __restrict__ stats, int phaseLimit )
{
  const int tid = threadIdx.x, nthr = blockDim.x, w = gf.w, la = gf.log2w + gf.log2h, area = 1 << la;
  const int tu0 = blk * tpb, nTu = min( tpb, n - tu0 );
  const int G = area < 64 ? area : 64, lane = tid & 63, sub = lane & ( G - 1 );
  const Lds L = carve( smemRaw, gf, y, tpb, matH, matV, true, true, tid, nthr );
  TuPar* par = reinterpret_cast<TuPar*>( L.tu + ( ( tpb * y.slot + 7 ) & ~7 ) );
  for( int t = tid; t < nTu; t += nthr )
  {
    const vvhip_tu_qp qq = qps[tu0 + t];
    TuPar P;
    quantParams( q, qq.qp, P.scale, P.qBits );
    P.add  = ( long long ) ( ( qq.flags & 1 ) ? 171 : 85 ) << ( P.qBits - 9 );           // Quant.cpp:775
    P.addN = ( long long ) ( ( qq.flags & 2 ) ? 171 : 256 ) << ( P.qBits - 9 );          // Quant.cpp:874
    const int32_t thres = P.qBits ? ( int32_t ) ( ( int64_t ) thrVal << ( P.qBits - 1 ) ) : ( int32_t ) ( ( int64_t ) ( thrVal >> 1 ) << P.qBits );
    P.useThres = thres / ( P.scale << 2 );                                                // Quant.cpp:173-180
    const int l2 = q.log2w + q.log2h, sqrt2 = l2 & 1, trShift = 15 - q.bitDepth - ( l2 >> 1 ) - sqrt2;
    P.iscale = cInvQuantScales[sqrt2][qq.qp % 6];                                          // Quant.cpp:601
    P.rightShift = 6 - ( trShift + qq.qp / 6 );                                            // Quant.cpp:561
    int tgt = 32 + P.rightShift - 7; if( tgt > 16 ) tgt = 16;                              // Quant.cpp:606
    P.inMax = ( 1 << ( tgt - 1 ) ) - 1;
    P.last = 0; P.absSum = 0; P.sse = 0; P.need = 0; P.pad = 0;
    for( int c = 0; c < 16; c++ ) { P.nz[c] = 0; P.big[c] = 0; }
    par[t] = P;
  }
  loadResi( gf, y, L, resi, resiStride, resiOff, tu0, nTu, tid, nthr );
  __syncthreads();
  if( phaseLimit == 1 ) return;
  fwd2d( gf, y, L, nTu, tid, nthr );                        // R2 = coefficients (int32 raster)
  if( phaseLimit == 2 ) return;

  // ---- significance ballots: thread <-> (TU t, scan position p) for nz/big, (TU t, raster position p) for the need-RDOQ test
  const int efArea = q.w * min( q.h, 32 );
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 );
    const int32_t* c = reinterpret_cast<const int32_t*>( L.tu + t * y.slot + y.r2 );
    const TuPar& P = par[t];
    const bool need = p < efArea && ( int32_t ) ( ( ( int64_t ) abs( c[p] ) * P.scale + P.addN ) >> P.qBits ) != 0;   // needRdoqCore, Quant.cpp:264-278
    int cs = 0;
    if( p < q.numScan ) cs = c[scan[p]];
    const unsigned long long mNz = __ballot( cs != 0 ), mBig = __ballot( abs( cs ) > P.useThres ), mNeed = __ballot( need );
    if( sub == 0 )
    {
      const int sh = lane & ~( G - 1 );
      const unsigned long long gm = G == 64 ? ~0ull : ( ( 1ull << G ) - 1 );
      const int chunk = p >> 6;
      if( chunk < 16 ) { par[t].nz[chunk] = ( mNz >> sh ) & gm; par[t].big[chunk] = ( mBig >> sh ) & gm; }
      if( ( mNeed >> sh ) & gm ) par[t].need = 1;             // benign race: every writer stores 1
    }
  }
  __syncthreads();
  if( phaseLimit == 3 ) return;
  for( int t = tid; t < nTu; t += nthr )                    // one lane per TU resolves QuantCore's scan logic (Quant.cpp:162-208)
  {
    int last = 0;
    for( int c = 15; c >= 0; c-- ) if( par[t].nz[c] ) { last = c * 64 + 63 - __clzll( ( long long ) par[t].nz[c] ); break; }
    if( q.cgIs4x4 && last >= 16 )
    {
      int g2 = -1;
      for( int c = last >> 6; c >= 0 && g2 < 0; c-- )
      {
        unsigned long long m = par[t].big[c];
        if( c == 0 ) m &= ~0xFFFFull;                         // CG 0 is never tested
        if( c == ( last >> 6 ) && ( last & 63 ) != 63 ) m &= ( 1ull << ( ( last & 63 ) + 1 ) ) - 1;
        if( m ) g2 = c * 4 + ( ( 63 - __clzll( ( long long ) m ) ) >> 4 );
      }
      if( g2 < 0 ) last = 15;
      else if( g2 != ( last >> 4 ) ) last = g2 * 16 + 15;
    }
    par[t].last = last;
  }
  __syncthreads();
  if( phaseLimit == 4 ) return;
  // ---- QuantCore + DeQuantCore.  R1 <- levels (raster int16), R0 <- dequantised coefficients transposed [x][y]
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 );
    int16_t* s = L.tu + t * y.slot;
    const int32_t* c = reinterpret_cast<const int32_t*>( s + y.r2 );
    const TuPar& P = par[t];
    int bp = p; bool inScan = false;
    if( p < q.numScan ) { bp = scan[p]; inScan = true; }
    else
    {
      // enumerate the zero-out region (x >= 32 or y >= 32) with the remaining indices
      const int r = p - q.numScan, wz = q.w - min( q.w, 32 ), rowsTop = min( q.h, 32 );
      if( r < rowsTop * wz ) { const int yy = r / wz, x = 32 + ( r - yy * wz ); bp = yy * q.w + x; }
      else bp = rowsTop * q.w + ( r - rowsTop * wz );
    }
    int lv = 0, dq = 0;
    if( inScan && p <= P.last )
    {
      const int32_t cv = c[bp];
      const uint32_t m = ( uint32_t ) ( int32_t ) ( ( ( int64_t ) abs( cv ) * P.scale + P.add ) >> P.qBits );          // Quant.cpp:219-220
      if( m )
      {
        atomicAdd( &par[t].absSum, m );
        lv = clip3i( -32768, 32767, cv < 0 ? -( int32_t ) m : ( int32_t ) m );
        const int cl = clip3i( -( P.inMax + 1 ), P.inMax, lv );                                                           // DeQuantCore, Quant.cpp:232-262
        int32_t v;
        if( P.rightShift > 0 ) v = ( int32_t ) ( ( uint32_t ) ( cl * P.iscale ) + ( 1u << ( P.rightShift - 1 ) ) ) >> P.rightShift;
        else                   v = ( int32_t ) ( ( uint32_t ) ( cl * P.iscale ) << ( -P.rightShift ) );
        dq = clip3i( -32768, 32767, v );
      }
    }
    const int by = bp >> q.log2w, bx = bp & ( q.w - 1 );
    s[y.r1 + bp] = ( int16_t ) lv;
    s[bx * y.ph + by] = ( int16_t ) dq;
  }
  __syncthreads();
  if( phaseLimit == 5 ) return;
  if( level )
  {
    if( area >= 8 )
    {
      const int lc = la - 3;
      for( int o = tid; o < ( nTu << lc ); o += nthr )
      {
        const int t = o >> lc, c8 = ( o & ( ( 1 << lc ) - 1 ) ) << 3;
        *reinterpret_cast<u32x4*>( level + ( size_t ) ( tu0 + t ) * area + c8 ) = *reinterpret_cast<const u32x4*>( L.tu + t * y.slot + y.r1 + c8 );
      }
    }
    else
      for( int o = tid; o < ( nTu << la ); o += nthr ) level[( size_t ) tu0 * area + o] = L.tu[( o >> la ) * y.slot + y.r1 + ( o & ( area - 1 ) )];
    __syncthreads();
  }
  if( phaseLimit == 6 ) return;
  inv2d( gi, y, L, nTu, tid, nthr );                        // R2 = reconstructed residual (raster int16)
  if( phaseLimit == 7 ) return;
  if( w >= 8 )
  {
    // 8 samples per lane: 16-byte LDS read of the reconstruction, 16-byte residual load and reconstruction store, DPP sum over the TU's lanes
    struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
    const int lc = la - 3, Gc = ( area >> 3 ) < 64 ? ( area >> 3 ) : 64;
    for( int o = tid; o < ( nTu << lc ); o += nthr )
    {
      const int t = o >> lc, c = o & ( ( 1 << lc ) - 1 ), yy = c >> ( gf.log2w - 3 ), x = ( c & ( ( w >> 3 ) - 1 ) ) << 3;
      const u32x4 rv = *reinterpret_cast<const u32x4*>( L.tu + t * y.slot + y.r2 + ( c << 3 ) );
      const u32x4 ov = reinterpret_cast<const U16*>( resi + resiOff[tu0 + t] + ( ptrdiff_t ) yy * resiStride + x )->v;
      if( rec ) *reinterpret_cast<u32x4*>( rec + ( size_t ) ( tu0 + t ) * area + ( c << 3 ) ) = rv;
      const uint32_t rr[4] = { rv.x, rv.y, rv.z, rv.w }, oo[4] = { ov.x, ov.y, ov.z, ov.w };
      unsigned long long e = 0;
#pragma unroll
      for( int k = 0; k < 4; k++ )
      {
        const int d0 = ( int ) ( int16_t ) ( oo[k] & 0xffff ) - ( int ) ( int16_t ) ( rr[k] & 0xffff ), d1 = ( ( int ) oo[k] >> 16 ) - ( ( int ) rr[k] >> 16 );
        e += ( unsigned long long ) ( ( long long ) d0 * d0 ) + ( unsigned long long ) ( ( long long ) d1 * d1 );
      }
      e = vvhipGroupSum64( e, Gc, lane );
      if( ( lane & ( Gc - 1 ) ) == 0 && e ) atomicAdd( &par[t].sse, e );
    }
  }
  else
  for( int o = tid; o < ( nTu << la ); o += nthr )
  {
    const int t = o >> la, p = o & ( area - 1 ), yy = p >> gf.log2w, x = p & ( w - 1 );
    const int r = L.tu[t * y.slot + y.r2 + p];
    if( rec ) rec[( size_t ) tu0 * area + o] = ( int16_t ) r;
    const int d = ( int ) resi[resiOff[tu0 + t] + ( ptrdiff_t ) yy * resiStride + x] - r;
    unsigned long long e = grpAdd64( ( unsigned long long ) ( ( long long ) d * d ), G );
    if( sub == 0 && e ) atomicAdd( &par[t].sse, e );
  }
  __syncthreads();
  if( stats )
    for( int t = tid; t < nTu; t += nthr )
    {
      vvhip_tu_stats st; st.abs_sum = ( int32_t ) par[t].absSum; st.last_scan_pos = par[t].last; st.need_rdoq = ( int32_t ) par[t].need; st.pad = 0; st.sse = par[t].sse;
      stats[tu0 + t] = st;
    }
}

__global__ void __launch_bounds__( 256 )
tuRdoKernel( const int16_t* __restrict__ resi, int resiStride, const int32_t* __restrict__ resiOff, int n,
             TrGeom gf, TrGeom gi, TuLay y, QGeom q, int tpb, const int16_t* __restrict__ matH, const int16_t* __restrict__ matV,
             const uint16_t* __restrict__ scan, const vvhip_tu_qp* __restrict__ qps, int thrVal,
             int16_t* __restrict__ level, int16_t* __restrict__ rec, vvhip_tu_stats* __restrict__ stats, int phaseLimit )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  tuRdoBody( smemRaw, blockIdx.x, resi, resiStride, resiOff, n, gf, gi, y, q, tpb, matH, matV, scan, qps, thrVal, level, rec, stats, phaseLimit );
}

// Every TU list of a picture that the matrix-core kernel does not take — the rectangular shapes of preset medium's multi-type tree (TrQuant.cpp:481-564 with the
// non-square shifts :544-545 and trShift :769-772 of Quant.cpp), 2-wide chroma, 64-point DST — in ONE launch: a job table in device memory (geometry, LDS layout, matrices,
// lists), workgroup -> job by the jobs' first workgroup.  (One tuRdoKernel launch per (shape, types) list was 45 launches for a 4K medium picture: 420 us, mostly launch gaps.)
struct TuGenJob
{
  int32_t resiStride, n, tpb, thrVal, blockStart, pad;
  TrGeom gf, gi; TuLay y; QGeom q;
  const int32_t* resiOff; const int16_t* matH; const int16_t* matV; const uint16_t* scan; const vvhip_tu_qp* qps; int16_t* level; int16_t* rec; vvhip_tu_stats* stats;
};

__global__ void __launch_bounds__( 256 )
tuRdoGenMultiKernel( const int16_t* __restrict__ resi, const TuGenJob* __restrict__ jobs, int nJobs, int phaseLimit )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  // which job: the last one whose first workgroup is <= this workgroup (nJobs <= 64: one lane per job, one ballot)
  const int lane = threadIdx.x & 63;
  const int st = lane < nJobs ? jobs[lane].blockStart : 0x7fffffff;
  const int k = __builtin_popcountll( __ballot( ( int ) blockIdx.x >= st ) ) - 1;
  const TuGenJob& J = jobs[__builtin_amdgcn_readfirstlane( k )];
  tuRdoBody( smemRaw, ( int ) blockIdx.x - J.blockStart, resi, J.resiStride, J.resiOff, J.n, J.gf, J.gi, J.y, J.q, J.tpb, J.matH, J.matV, J.scan, J.qps, J.thrVal, J.level, J.rec, J.stats, phaseLimit );
}


// --------------------------------------------------------------------------------------------
// Fused TU pipeline, row-per-lane form for square N x N TUs (N = 8, 16, 32).
// The N lanes of a TU sit in ONE wavefront, so the whole pipeline needs no workgroup barrier after the ROM is staged:
//   lane r holds residual row r in registers -> forward rows (all N frequencies of its row) -> LDS transpose ->
//   lane c holds column c of the intermediate -> forward columns -> the N coefficients of column c stay in registers ->
//   QuantCore / DeQuantCore on registers (significance through DPP max/or over the TU's lanes) ->
//   inverse columns straight from registers -> LDS transpose -> inverse rows -> reconstruction row + SSE against the
//   residual row that never left the registers.
// Matrix rows are wave-wide LDS broadcasts (16-byte reads), every multiply is v_dot2_i32_i16; ~6x fewer instructions per
// sample than the sample-per-lane kernel above (which remains for non-square, 4-, 2- and 64-point TUs).
// --------------------------------------------------------------------------------------------
struct TuRowArgs
{
  const int32_t* resiOff; int n;
  TrGeom gf, gi; QGeom q;
  const int16_t* matH; const int16_t* matV; const uint16_t* scan;
  const vvhip_tu_qp* qps; int thrVal;
  int16_t* level; int16_t* rec; vvhip_tu_stats* stats;
  int phaseLimit;          // profiling aid ($VVHIP_TU_PHASES): stop after phase k, 0 = run everything
  int groupStride;         // workgroups assigned to this job (a workgroup walks groups b, b + groupStride, ...)
};

template<int N, int SPLIT> struct TuRowLds
{
  static constexpr int LPT = N * SPLIT, TPB = 256 / LPT, ND = N / 2, NO = N / SPLIT, P = N == 8 ? 8 : N + 8, LINES = 256 / SPLIT;
  static constexpr int oMat = 0, oInv = oMat + 4 * N * N * 2, oTile = oInv + N * N * 2, oCoef = ( oTile + TPB * N * P * 2 + 15 ) & ~15,
                       bytes = oCoef + NO * 256 * 4;
};

template<int N, int SPLIT>
__device__ __forceinline__ void
tuRdoRowBody( unsigned char* __restrict__ smem, const int blockIndex, const int16_t* __restrict__ resi, const int resiStride, const TuRowArgs& A )
{
  // SPLIT lanes share one row / column: lane (line, part) produces outputs [part*NO, (part+1)*NO) of its line; LPT lanes per TU (<= 64).
  // Output loops are rolled (8 outputs per trip); per-lane coefficients live in a private LDS column (lane-major, conflict-free).
  typedef TuRowLds<N, SPLIT> L;
  constexpr int LPT = N * SPLIT, TPB = 256 / LPT, ND = N / 2, NC = N / 8, NO = N / SPLIT, NOC = NO / 8;
  constexpr int P = L::P;                                     // tile row pitch (int16): odd number of 16-byte chunks
  static_assert( LPT <= 64 && NO >= 8 && ( SPLIT == 1 || SPLIT == 2 ), "geometry" );
  int16_t  ( *sMat )[N * N]   = reinterpret_cast<int16_t ( * )[N * N]>( smem + L::oMat );       // Th, Tv, Th^T, Tv^T
  uint16_t* sInv              = reinterpret_cast<uint16_t*>( smem + L::oInv );                  // raster position -> scan position
  int16_t  ( *sTile )[N * P]  = reinterpret_cast<int16_t ( * )[N * P]>( smem + L::oTile );
  int32_t  ( *sCoef )[256]    = reinterpret_cast<int32_t ( * )[256]>( smem + L::oCoef );        // [output][thread]: private per-lane coefficients
  struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
  const int32_t* __restrict__ resiOff = A.resiOff; const int n = A.n;
  const TrGeom& gf = A.gf; const TrGeom& gi = A.gi; const QGeom& q = A.q;
  const int16_t* __restrict__ matH = A.matH; const int16_t* __restrict__ matV = A.matV; const uint16_t* __restrict__ scan = A.scan;
  const vvhip_tu_qp* __restrict__ qps = A.qps; const int thrVal = A.thrVal;
  int16_t* __restrict__ level = A.level; int16_t* __restrict__ rec = A.rec; vvhip_tu_stats* __restrict__ stats = A.stats;

  const int tid = threadIdx.x;
  const int tl = tid / LPT, li = tid & ( LPT - 1 ), r = li / SPLIT, part = li & ( SPLIT - 1 ), lane = tid & 63;
  const int line = tid / SPLIT;                               // row / column index inside the workgroup
  const int o0 = part * NO;                                   // first output index of this lane
  for( int i = tid; i < N * N; i += 256 )
  {
    const int k = i / N, j = i - k * N;
    const int16_t a = matH[i], b = matV[i];
    sMat[0][i] = a; sMat[1][i] = b; sMat[2][j * N + k] = a; sMat[3][j * N + k] = b;
    sInv[scan[i]] = ( uint16_t ) i;
  }
  __syncthreads();
  if( A.phaseLimit == 1 ) return;
  // persistent form: a workgroup stages the ROM once and walks groups blockIndex, blockIndex + groupStride, ... (all LDS hand-offs below are inside a wave)
  for( int grp = blockIndex; grp * TPB < n; grp += A.groupStride )
  {
    const int tu = grp * TPB + tl;
    const bool valid = tu < n;
    int16_t* tile = sTile[tl];
    // the dependent global loads (offset -> residual row, QP) are issued before the ROM is staged so that their latencies overlap
    const int16_t* src = resi + ( valid ? resiOff[tu] : 0 ) + ( ptrdiff_t ) r * resiStride;
    const vvhip_tu_qp qq = valid ? qps[tu] : vvhip_tu_qp{ 32, 0 };
    uint32_t x[ND];
  #pragma unroll
    for( int c = 0; c < NC; c++ )
    {
      u32x4 v = { 0, 0, 0, 0 };
      if( valid ) v = reinterpret_cast<const U16*>( src + 8 * c )->v;
      x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }

  #define DOT_ROW( ACC, VEC, MROW ) { ACC = 0; _Pragma( "unroll" ) for( int c_ = 0; c_ < NC; c_++ ) {                          \
        const u32x4 m_ = *reinterpret_cast<const u32x4*>( ( MROW ) + 8 * c_ );                                                  \
        ACC = dot2( VEC[4 * c_], m_.x, ACC ); ACC = dot2( VEC[4 * c_ + 1], m_.y, ACC ); ACC = dot2( VEC[4 * c_ + 2], m_.z, ACC ); ACC = dot2( VEC[4 * c_ + 3], m_.w, ACC ); } }
  #define WAVE_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }

    // ---- forward rows: tmp[j][r] = sat16( ( sum_k blk[r][k] * Th[j][k] + rnd ) >> shift1 )        (cpyCoeff + TrQuant.cpp:548)
    {
      const int rnd1 = gf.shift1 > 0 ? 1 << ( gf.shift1 - 1 ) : 0;
  #pragma unroll 2
      for( int jj = 0; jj < NO; jj++ )
      {
        const int j = o0 + jj;
        int acc;
        DOT_ROW( acc, x, &sMat[0][j * N] );
        tile[j * P + r] = ( int16_t ) sat16( ( int ) ( ( uint32_t ) acc + ( uint32_t ) rnd1 ) >> gf.shift1 );
      }
    }
    WAVE_SYNC();
    if( A.phaseLimit == 2 ) return;
    // ---- forward columns (line = horizontal frequency c): coef[j2] = ( sum_k tmp[c][k] * Tv[j2][k] + rnd ) >> shift2   (TrQuant.cpp:549)
    const int cidx = r;
    {
      uint32_t y[ND];
  #pragma unroll
      for( int c = 0; c < NC; c++ ) { const u32x4 v = *reinterpret_cast<const u32x4*>( &tile[cidx * P + 8 * c] ); y[4 * c] = v.x; y[4 * c + 1] = v.y; y[4 * c + 2] = v.z; y[4 * c + 3] = v.w; }
      const int rnd2 = 1 << ( gf.shift2 - 1 );
      const bool colLive = cidx < N - gf.skipW;
  #pragma unroll 2
      for( int jj = 0; jj < NO; jj++ )
      {
        const int j2 = o0 + jj;
        int acc;
        DOT_ROW( acc, y, &sMat[1][j2 * N] );
        sCoef[jj][tid] = ( colLive && j2 < N - gf.skipH ) ? ( int ) ( ( uint32_t ) acc + ( uint32_t ) rnd2 ) >> gf.shift2 : 0;
      }
    }
    if( A.phaseLimit == 3 ) return;
    // ---- quantiser constants of this TU
    int scale, qBits;
    quantParams( q, qq.qp, scale, qBits );
    const long long add  = ( long long ) ( ( qq.flags & 1 ) ? 171 : 85 ) << ( qBits - 9 );            // Quant.cpp:775
    const long long addN = ( long long ) ( ( qq.flags & 2 ) ? 171 : 256 ) << ( qBits - 9 );           // Quant.cpp:874
    const int32_t thres = qBits ? ( int32_t ) ( ( int64_t ) thrVal << ( qBits - 1 ) ) : ( int32_t ) ( ( int64_t ) ( thrVal >> 1 ) << qBits );
    const int useThres = thres / ( scale << 2 );                                                      // Quant.cpp:173-180
    const int trShift = 15 - q.bitDepth - q.log2w;
    const int iscale = cInvQuantScales[0][qq.qp % 6];                                                  // Quant.cpp:601 (square: no sqrt2)
    const int rightShift = 6 - ( trShift + qq.qp / 6 );                                                // Quant.cpp:561
    int tgt = 32 + rightShift - 7; if( tgt > 16 ) tgt = 16;                                            // Quant.cpp:606
    const int inMax = ( 1 << ( tgt - 1 ) ) - 1;

    // ---- significance: last non-zero scan position, need-RDOQ flag, coefficient-group test (Quant.cpp:162-208, :264-278)
    // needRdoqCore asks whether ANY coefficient quantises to non-zero with the RDOQ offset: the quantiser is monotonic in |c|, so it is
    // one 64-bit test on the TU's largest magnitude.
    uint32_t last = 0, maxAbs = 0;
    for( int jj = 0; jj < NO; jj++ )
    {
      const uint32_t si = sInv[( o0 + jj ) * N + cidx];
      const int c = sCoef[jj][tid];
      const uint32_t ac = ( uint32_t ) abs( c );
      maxAbs = ac > maxAbs ? ac : maxAbs;
      last = ( c != 0 && si > last ) ? si : last;
    }
    last = vvhipGroupMax32( last, LPT, lane );
    maxAbs = vvhipGroupMax32( maxAbs, LPT, lane );
    const uint32_t need = ( uint32_t ) ( ( int32_t ) ( ( ( int64_t ) maxAbs * scale + addN ) >> qBits ) != 0 );
    if( last >= 16 )
    {
      uint32_t lo = 0, hi = 0;
      for( int jj = 0; jj < NO; jj++ )
      {
        const uint32_t si = sInv[( o0 + jj ) * N + cidx];
        if( si >= 16 && si <= last && abs( sCoef[jj][tid] ) > useThres ) { const int cg = si >> 4; if( cg < 32 ) lo |= 1u << cg; else hi |= 1u << ( cg - 32 ); }
      }
      lo = vvhipGroupOr32( lo, LPT, lane );
      hi = N > 16 ? vvhipGroupOr32( hi, LPT, lane ) : 0u;
      const unsigned long long big = ( ( unsigned long long ) hi << 32 ) | lo;
      if( big == 0 ) last = 15;
      else { const uint32_t g2 = 63 - __clzll( ( long long ) big ); if( g2 != ( last >> 4 ) ) last = g2 * 16 + 15; }
    }
    if( A.phaseLimit == 4 ) return;
    // ---- QuantCore + DeQuantCore (Quant.cpp:213-227, :232-262), branch-free per coefficient: level pairs -> tile (transpose for the raster
    // store), dequantised pairs -> the lane's own sCoef column (rows 0..NO/2-1, already consumed).  When every |c| of the wave fits 16 bits
    // the level is a 24-bit multiply-add in 32 bits (|c|*scale < 2^31, add < 2^29.5); otherwise the 64-bit form.
    uint32_t absSum = 0;
    const bool narrow = __builtin_amdgcn_ballot_w64( maxAbs >= 65536u || qBits > 30 ) == 0ull;
    const uint32_t add32 = ( uint32_t ) add;
    const int rndDq = rightShift > 0 ? 1 << ( rightShift - 1 ) : 0;
  #define TU_LEVEL_PAIR( MEXPR )                                                                                                       \
    for( int jj = 0; jj < NO; jj += 2 )                                                                                                \
    {                                                                                                                                  \
      int lv[2], dq[2];                                                                                                                \
      _Pragma( "unroll" ) for( int e = 0; e < 2; e++ )                                                                                 \
      {                                                                                                                                \
        const int cv = sCoef[jj + e][tid];                                                                                             \
        const uint32_t ac = ( uint32_t ) abs( cv );                                                                                    \
        uint32_t m = MEXPR;                                                                                                            \
        m = sInv[( o0 + jj + e ) * N + cidx] <= last ? m : 0u;                                                                         \
        absSum += m;                                                                                                                   \
        const int sm = cv < 0 ? -( int32_t ) m : ( int32_t ) m;                                                                        \
        lv[e] = clip3i( -32768, 32767, sm );                                                                                           \
        const int cl = clip3i( -( inMax + 1 ), inMax, lv[e] );                                                                         \
        const int pr = __mul24( cl, iscale );                                                                                          \
        const int32_t v = rightShift > 0 ? ( int32_t ) ( ( uint32_t ) pr + ( uint32_t ) rndDq ) >> rightShift : ( int32_t ) ( ( uint32_t ) pr << ( -rightShift ) ); \
        dq[e] = clip3i( -32768, 32767, v );                                                                                            \
      }                                                                                                                                \
      tile[( o0 + jj ) * P + cidx] = ( int16_t ) lv[0];                                                                                \
      tile[( o0 + jj + 1 ) * P + cidx] = ( int16_t ) lv[1];                                                                            \
      sCoef[jj >> 1][tid] = ( int32_t ) ( ( uint32_t ) ( dq[0] & 0xffff ) | ( ( uint32_t ) dq[1] << 16 ) );                            \
    }
    if( narrow ) { TU_LEVEL_PAIR( ( ( uint32_t ) __umul24( ac, ( uint32_t ) scale ) + add32 ) >> qBits ) }
    else         { TU_LEVEL_PAIR( ( uint32_t ) ( int32_t ) ( ( ( int64_t ) ac * scale + add ) >> qBits ) ) }
  #undef TU_LEVEL_PAIR
    if( A.phaseLimit == 5 ) return;
    absSum = vvhipGroupSum32( absSum, LPT, lane );
    WAVE_SYNC();
    // ---- levels: raster rows -> HBM (16-byte stores)
    if( level && valid )
  #pragma unroll
      for( int c = 0; c < NOC; c++ )
        *reinterpret_cast<u32x4*>( level + ( size_t ) tu * N * N + r * N + o0 + 8 * c ) = *reinterpret_cast<const u32x4*>( &tile[r * P + o0 + 8 * c] );
    WAVE_SYNC();
    if( A.phaseLimit == 6 ) return;
    // ---- inverse columns: t1[j][c] = clip( ( sum_k deq[k][c] * Tv[k][j] + 64 ) >> 7 ), c < N - skipW   (TrQuant.cpp:612)
    {
      uint32_t dqp[ND];
  #pragma unroll
      for( int k = 0; k < ND; k++ ) dqp[k] = ( uint32_t ) sCoef[k % ( NO / 2 )][line * SPLIT + k / ( NO / 2 )];     // pair k of the column: lane part k / (NO/2), its row k % (NO/2)
      const int rnd1 = 1 << ( gi.shift1 - 1 );
      const bool colLive = cidx < N - gi.skipW;
  #pragma unroll 2
      for( int jj = 0; jj < NO; jj++ )
      {
        const int j = o0 + jj;
        int acc;
        DOT_ROW( acc, dqp, &sMat[3][j * N] );
        tile[j * P + cidx] = ( int16_t ) ( colLive ? sat16( ( int ) ( ( uint32_t ) acc + ( uint32_t ) rnd1 ) >> gi.shift1 ) : 0 );
      }
    }
    WAVE_SYNC();
    if( A.phaseLimit == 7 ) return;
    // ---- inverse rows: rec[r][j2] = clip( ( sum_k t1[r][k] * Th[k][j2] + rnd ) >> shift2 ); SSE against the residual row (re-read: L2 hit)
    unsigned long long sse = 0;
    {
      uint32_t t[ND];
  #pragma unroll
      for( int c = 0; c < NC; c++ ) { const u32x4 v = *reinterpret_cast<const u32x4*>( &tile[r * P + 8 * c] ); t[4 * c] = v.x; t[4 * c + 1] = v.y; t[4 * c + 2] = v.z; t[4 * c + 3] = v.w; }
      const int rnd2 = 1 << ( gi.shift2 - 1 );
      for( int c8 = 0; c8 < NOC; c8++ )
      {
        u32x4 xv = { 0, 0, 0, 0 };
        if( valid ) xv = reinterpret_cast<const U16*>( src + o0 + 8 * c8 )->v;
        const uint32_t xs[4] = { xv.x, xv.y, xv.z, xv.w };
        uint32_t rp[4];
  #pragma unroll
        for( int pr = 0; pr < 4; pr++ )
        {
          int rv[2];
  #pragma unroll
          for( int e = 0; e < 2; e++ )
          {
            int acc;
            DOT_ROW( acc, t, &sMat[2][( o0 + 8 * c8 + 2 * pr + e ) * N] );
            rv[e] = sat16( ( int ) ( ( uint32_t ) acc + ( uint32_t ) rnd2 ) >> gi.shift2 );
          }
          rp[pr] = ( uint32_t ) ( rv[0] & 0xffff ) | ( ( uint32_t ) rv[1] << 16 );
          const int d0 = ( int ) ( int16_t ) ( xs[pr] & 0xffff ) - rv[0], d1 = ( ( int ) xs[pr] >> 16 ) - rv[1];
          sse += ( unsigned long long ) ( ( long long ) d0 * d0 ) + ( unsigned long long ) ( ( long long ) d1 * d1 );
        }
        if( rec && valid )
        {
          u32x4 v; v.x = rp[0]; v.y = rp[1]; v.z = rp[2]; v.w = rp[3];
          *reinterpret_cast<u32x4*>( rec + ( size_t ) tu * N * N + r * N + o0 + 8 * c8 ) = v;
        }
      }
    }
  #undef DOT_ROW
  #undef WAVE_SYNC
    sse = vvhipGroupSum64( sse, LPT, lane );
    if( stats && valid && li == 0 )
    {
      vvhip_tu_stats st; st.abs_sum = ( int32_t ) absSum; st.last_scan_pos = ( int32_t ) last; st.need_rdoq = ( int32_t ) need; st.pad = 0; st.sse = sse;
      stats[tu] = st;
    }
  }
}

template<int N, int SPLIT>
__global__ void __launch_bounds__( 256 )
tuRdoRowKernel( const int16_t* __restrict__ resi, int resiStride, TuRowArgs args )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smem[TuRowLds<N, SPLIT>::bytes];
  tuRdoRowBody<N, SPLIT>( smem, blockIdx.x, resi, resiStride, args );
}

// Several square TU sizes of one residual plane in ONE launch (largest first): the per-size launches are each too small to fill
// the 1024 SIMDs (a 1080p frame has 1980 32x32 TUs = 2 waves per SIMD), together they overlap.
struct TuMultiJobs { int nJobs; int blockStart[4]; int size[4]; TuRowArgs j[4]; };

__global__ void __launch_bounds__( 256 )
tuRdoRowMultiKernel( const int16_t* __restrict__ resi, int resiStride, TuMultiJobs jobs )
{
  constexpr int B32 = TuRowLds<32, 2>::bytes, B16 = TuRowLds<16, 2>::bytes, B8 = TuRowLds<8, 1>::bytes;
  __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smem[B32 > B16 ? ( B32 > B8 ? B32 : B8 ) : ( B16 > B8 ? B16 : B8 )];
  int k = 0;
#pragma unroll
  for( int i = 1; i < 4; i++ ) if( i < jobs.nJobs && ( int ) blockIdx.x >= jobs.blockStart[i] ) k = i;
  const int blk = blockIdx.x - jobs.blockStart[k];
  if( jobs.size[k] == 32 )      tuRdoRowBody<32, 2>( smem, blk, resi, resiStride, jobs.j[k] );
  else if( jobs.size[k] == 16 ) tuRdoRowBody<16, 2>( smem, blk, resi, resiStride, jobs.j[k] );
  else                          tuRdoRowBody<8, 1>( smem, blk, resi, resiStride, jobs.j[k] );
}


// --------------------------------------------------------------------------------------------
// Fused TU pipeline on the matrix cores, square N x N TUs (N = 8, 16, 32): ONE wavefront per 32x32 tile of (32/N)^2 TUs.
// Every 1-D pass is a 32x32x32 integer matrix product (v_mfma_i32_32x32x32_i8).  The kernel matrices are 8-bit; the 16-bit data operand
// is split into its high byte and (low byte - 128), two products per pass, recombined as (hi << 8) + lo with the 128 * sum(matrix row)
// correction and the rounding offset preloaded into the accumulator.  All sums are exact (|sum| <= 32 * 2^15 * 90 < 2^31), so the
// results equal the dot-product kernels above bit for bit.  For N < 32 the matrices are block-diagonal (common.h: VvhipTuMxOps).
// The four passes alternate the side the data sits on, and the K-slots of the operands are bound to the rows a lane's result registers
// hold (mxIdx), so results feed the next pass from the lane's own registers — no LDS transpose, no cross-lane traffic between passes:
//   residual (A: lane = row y, slots x)         x  Th   (B)  ->  D1: lane = hor. frequency k, registers = y
//   Tv (A)   x  D1 (B: lane = k, slots y)                    ->  D2: lane = k, registers = ver. frequency k2     = coefficients
//   QuantCore / DeQuantCore on the 16 registers; significance / abs-sum by DPP over the TU's lanes
//   dequantised (A: lane = k, slots k2)         x  Tv^T (B)  ->  D3: lane = y, registers = k
//   Th^T (A) x  D3 (B: lane = y, slots k)                    ->  D4: lane = y, registers = x   = reconstruction, same layout as the input
// LDS is used only to turn the level registers into 16-byte raster stores.  Waves are independent (no workgroup barrier).
// --------------------------------------------------------------------------------------------
#ifndef TUMX_ZERO_MIN_N
#define TUMX_ZERO_MIN_N 8      /* smallest TU size whose tiles take the all-zero shortcut of tuMxBody */
#endif
typedef int v4i  __attribute__( ( ext_vector_type( 4 ) ) );
typedef int v16i __attribute__( ( ext_vector_type( 16 ) ) );

struct TuMxArgs
{
  const int32_t* resiOff; int n;
  int shF1, shF2, shI1, shI2, skipW, skipH;
  QGeom q;
  const VvhipTuMxOps* opH; const VvhipTuMxOps* opV; const uint16_t* pos;
  const vvhip_tu_qp* qps; int thrVal;
  int16_t* level; int16_t* rec; vvhip_tu_stats* stats;
  int tiles;               // 32x32 tiles of this job
  int phaseLimit;          // bits 0..7: profiling aid ($VVHIP_TU_PHASES): skip the rest of a tile after phase k, 0 = run everything; bit 8: sparse outputs (vvhip_tu_set_sparse_outputs)
  int waveStride;          // waves assigned to this job (wave w walks tiles w, w + waveStride, ...)
  int resiStride;          // row pitch of this job's residual blocks; 0: the launch's common pitch (vvhip_tu_rdo_multi_strided: compact per-TU blocks, pitch = width)
};

// 16 values of 16-bit range -> the two byte operands (slot s = register s): low bytes - 128 (xor 0x80) and high bytes
__device__ __forceinline__ void mxSplit( const int* d, v4i& lo, v4i& hi )
{
#pragma unroll
  for( int g = 0; g < 4; g++ )
  {
    const uint32_t p01 = __builtin_amdgcn_perm( ( uint32_t ) d[4 * g + 1], ( uint32_t ) d[4 * g], 0x05040100u );
    const uint32_t p23 = __builtin_amdgcn_perm( ( uint32_t ) d[4 * g + 3], ( uint32_t ) d[4 * g + 2], 0x05040100u );
    lo[g] = ( int ) ( __builtin_amdgcn_perm( p23, p01, 0x06040200u ) ^ 0x80808080u );
    hi[g] = ( int ) __builtin_amdgcn_perm( p23, p01, 0x07050301u );
  }
}

// max over aligned groups of G lanes of two packed unsigned 16-bit values at once (v_pk_max_u16)
typedef unsigned short u16x2v __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ uint32_t pkMaxU16( uint32_t a, uint32_t b )
{
  return __builtin_bit_cast( uint32_t, __builtin_elementwise_max( __builtin_bit_cast( u16x2v, a ), __builtin_bit_cast( u16x2v, b ) ) );
}
__device__ __forceinline__ uint32_t tuMxGroupMaxPk16( uint32_t v, int G, int lane )
{
  v = pkMaxU16( v, ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR1 ) );
  v = pkMaxU16( v, ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR2 ) );
  if( G >= 8 ) v = pkMaxU16( v, ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_HALF_MIRROR ) );
  if( G >= 16 ) v = pkMaxU16( v, ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_MIRROR ) );
  if( G >= 32 )
  {
    const uint32_t r0 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 0 ),  r1 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 16 );
    const uint32_t r2 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 32 ), r3 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 48 );
    const uint32_t a = pkMaxU16( r0, r1 ), b = pkMaxU16( r2, r3 );
    v = G == 64 ? pkMaxU16( a, b ) : ( lane < 32 ? a : b );
  }
  return v;
}

__device__ __forceinline__ uint32_t sadU32( uint32_t a, uint32_t b, uint32_t acc ) { uint32_t r; asm( "v_sad_u32 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( b ), "v"( acc ) ); return r; }
// clip( x, lo, hi ) with run-time bounds lo <= hi as ONE v_med3_i32 (the compiler emits min + max when it cannot prove lo <= hi)
__device__ __forceinline__ int med3i( int x, int lo, int hi ) { int r; asm( "v_med3_i32 %0, %1, %2, %3" : "=v"( r ) : "v"( x ), "v"( lo ), "v"( hi ) ); return r; }

// quantiser constants of one TU from its QP (Quant.cpp:775, :874, :173-180, :561, :601-606); the scale tables as select chains (no memory)
struct TuMxQ { int scale, qBits, thres, iscale, rightShift, inMax; long long addQ, addN; };
__device__ __forceinline__ int tuMxSel6( int r, int a0, int a1, int a2, int a3, int a4, int a5 )
{
  return r == 0 ? a0 : r == 1 ? a1 : r == 2 ? a2 : r == 3 ? a3 : r == 4 ? a4 : a5;
}
__device__ __forceinline__ TuMxQ tuMxParams( const QGeom& q, const vvhip_tu_qp qq, const int thrVal )
{
  TuMxQ P;
  const int per = qq.qp / 6, rem = qq.qp - 6 * per;
  const int trShift = 15 - q.bitDepth - q.log2w;                                              // square TU: no sqrt2 scaling
  P.scale = tuMxSel6( rem, 26214, 23302, 20560, 18396, 16384, 14564 );                        // Rom.cpp:1390
  P.iscale = tuMxSel6( rem, 40, 45, 51, 57, 64, 72 );                                         // Rom.cpp:1396
  P.qBits = 14 + per + trShift;
  P.addQ = ( long long ) ( ( qq.flags & 1 ) ? 171 : 85 ) << ( P.qBits - 9 );
  P.addN = ( long long ) ( ( qq.flags & 2 ) ? 171 : 256 ) << ( P.qBits - 9 );
  P.thres = P.qBits ? ( int32_t ) ( ( int64_t ) thrVal << ( P.qBits - 1 ) ) : ( int32_t ) ( ( int64_t ) ( thrVal >> 1 ) << P.qBits );
  P.rightShift = 6 - ( trShift + per );
  int tgt = 32 + P.rightShift - 7; if( tgt > 16 ) tgt = 16;
  P.inMax = ( 1 << ( tgt - 1 ) ) - 1;
  return P;
}

// 16 pre-saturation values -> byte operands of sat16( value ): v_cvt_pk_i16_i32 saturates and packs two values per instruction
typedef short s16x2v __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ void mxSplitSat( const int* d, v4i& lo, v4i& hi )
{
#pragma unroll
  for( int g = 0; g < 4; g++ )
  {
    const uint32_t p01 = __builtin_bit_cast( uint32_t, __builtin_amdgcn_cvt_pk_i16( d[4 * g], d[4 * g + 1] ) );
    const uint32_t p23 = __builtin_bit_cast( uint32_t, __builtin_amdgcn_cvt_pk_i16( d[4 * g + 2], d[4 * g + 3] ) );
    lo[g] = ( int ) ( __builtin_amdgcn_perm( p23, p01, 0x06040200u ) ^ 0x80808080u );
    hi[g] = ( int ) __builtin_amdgcn_perm( p23, p01, 0x07050301u );
  }
}

template<int N>
__device__ __forceinline__ void
tuMxBody( int16_t* __restrict__ stage, int32_t* __restrict__ sInit, v4i* __restrict__ sOps, const int waveIndex, const int16_t* __restrict__ resi, const int resiStride, const TuMxArgs& A )
{
  // A lane's 16 registers are 16 consecutive rows (coefficient side) / samples (residual side) 16h .. 16h+15 of its column / row:
  // R TUs per lane with VPR registers each; a TU's lanes are G consecutive lanes (N = 32: the same 32 lanes of both halves).
  constexpr int TPS = 32 / N, TPT = TPS * TPS, R = N >= 16 ? 1 : 16 / N, VPR = 16 / R, G = N == 32 ? 64 : N;
  constexpr int PS = N < 8 ? N : 8, NP = 16 / PS;                     // a lane's 16 samples are fetched / stored in NP runs of PS samples (one TU each)
  constexpr int LP = 40;                                              // staging pitch (int16): rows 80 bytes apart
  struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
  struct __attribute__( ( packed, aligned( 2 ) ) ) U8 { u32x2 v; };
  const int lane = threadIdx.x & 63, h = lane >> 5, c32 = lane & 31;
  const int blkL = c32 / N, inL = c32 % N;                            // the lane's row (residual side) / column (coefficient side): TU block, index inside
  const int blk0 = ( 16 * h ) / N;                                    // first TU block along the register direction
  const v16i zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define WAVE_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  // one 1-D pass: the product of the high bytes first, shifted up and joined with the accumulator preload (rounding + the 128 x row-sum correction), then the product of the
  // low bytes ON TOP of it — one 16-register accumulator per pass instead of two (hi and lo side by side cost the 8/16/32-point bodies ~30 registers: 159 -> 130 for N = 32)
#define MX_PASS( HA, HB, LA, LB, C, SH ) { v16i acc_ = __builtin_amdgcn_mfma_i32_32x32x32_i8( HA, HB, zero16, 0, 0, 0 );                      \
    _Pragma( "unroll" ) for( int v = 0; v < 16; v++ ) acc_[v] = ( acc_[v] << 8 ) + C[v];                                                     \
    acc_ = __builtin_amdgcn_mfma_i32_32x32x32_i8( LA, LB, acc_, 0, 0, 0 );                                                                   \
    _Pragma( "unroll" ) for( int v = 0; v < 16; v++ ) d[v] = acc_[v] >> ( SH ); }
#define TUMX_KEEP( ARR ) { int k_ = 0; _Pragma( "unroll" ) for( int v = 0; v < 16; v++ ) k_ ^= ARR[v]; if( k_ == 0x12345678 ) A.stats[0].pad = 1; }   /* phase profiling: keeps the values live */

  // residual of a tile: lane = row Y of the tile, samples X = 16h .. 16h+15 (the K-slots of this lane) in NP runs of PS samples.  The wave's first
  // tile is requested BEFORE the constants below: the offset -> residual chain and the ROM -> LDS chain overlap.
  auto loadResidual = [&]( const int tile, uint32_t* xr )
  {
#pragma unroll
    for( int c = 0; c < NP; c++ )
    {
      const int X0 = 16 * h + PS * c;
      const int tu = tile * TPT + blkL * TPS + X0 / N;
      const bool ok = tile < A.tiles && tu < A.n;
      const int16_t* src = resi + ( ok ? A.resiOff[tu] : 0 ) + ( ptrdiff_t ) inL * resiStride + X0 % N;
      if( PS == 8 )
      {
        u32x4 v = { 0, 0, 0, 0 };
        if( ok ) v = reinterpret_cast<const U16*>( src )->v;
        xr[4 * c] = v.x; xr[( 4 * c + 1 ) & 7] = v.y; xr[( 4 * c + 2 ) & 7] = v.z; xr[( 4 * c + 3 ) & 7] = v.w;
      }
      else
      {
        u32x2 v = { 0, 0 };
        if( ok ) v = reinterpret_cast<const U8*>( src )->v;
        xr[( 2 * c ) & 7] = v.x; xr[( 2 * c + 1 ) & 7] = v.y;
      }
    }
  };
  uint32_t xr[8];
  loadResidual( waveIndex, xr );

  // ---- per-wave constants.  The zero-out of the 32-point DST-7 / DCT-8 (coefficients beyond 16 dropped, TrQuant.cpp:496-497) is folded
  // into the operands: a dead column k gets an all-zero Th operand (its intermediate becomes 0), a dead row k2 an all-zero Tv operand row;
  // with the correction dropped as well the result is ( rnd >> shift ) = 0.
  const int rndF1 = A.shF1 > 0 ? 1 << ( A.shF1 - 1 ) : 0, rndF2 = 1 << ( A.shF2 - 1 ), rndI1 = 1 << ( A.shI1 - 1 ), rndI2 = 1 << ( A.shI2 - 1 );
  const bool liveCol = inL < N - A.skipW, liveRow = mxSigma( c32 ) % N < N - A.skipH;
  const v4i zero4 = { 0, 0, 0, 0 };
  // the two forward passes' operands live in LDS (lane-private 16-byte slots), fetched right before their products: registers less through the quantiser
  sOps[lane]       = liveCol ? *reinterpret_cast<const v4i*>( A.opH->nat[lane] ) : zero4;
  sOps[64 + lane]  = liveRow ? *reinterpret_cast<const v4i*>( A.opV->rowP[lane] ) : zero4;
  // (the two inverse passes' operands are read from the records when a tile gets that far — most tiles quantise to nothing — instead of sitting in LDS: 2 KB per wave, round 5)
  const int cP1 = ( liveCol ? A.opH->rowSum[c32] : 0 ) + rndF1, cI1 = A.opV->colSum[c32] + rndI1;
  // accumulator preloads of the two passes whose matrix sits on the A side (they depend on the result register = logical row 16h + v): LDS
  sInit[c32]      = ( c32 % N < N - A.skipH ? A.opV->rowSum[c32] : 0 ) + rndF2;
  sInit[32 + c32] = A.opH->colSum[c32] + rndI2;
  uint32_t pos[16];
  {
    const u32x4 p0 = *reinterpret_cast<const u32x4*>( A.pos + lane * 16 ), p1 = *reinterpret_cast<const u32x4*>( A.pos + lane * 16 + 8 );
    const uint32_t pw[8] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w };
#pragma unroll
    for( int v = 0; v < 16; v++ ) pos[v] = ( v & 1 ) ? pw[v >> 1] >> 16 : pw[v >> 1] & 0xffffu;
  }
  WAVE_SYNC();
  if( ( A.phaseLimit & 0xff ) == 1 ) return;

  for( int tile = waveIndex; tile < A.tiles; tile += A.waveStride )
  {
    v4i aLo, aHi;
    if( tile != waveIndex ) loadResidual( tile, xr );
#pragma unroll
    for( int g = 0; g < 4; g++ )
    {
      aLo[g] = ( int ) ( __builtin_amdgcn_perm( xr[2 * g + 1], xr[2 * g], 0x06040200u ) ^ 0x80808080u );
      aHi[g] = ( int ) __builtin_amdgcn_perm( xr[2 * g + 1], xr[2 * g], 0x07050301u );
    }
    vvhip_tu_qp qqs[R];                                               // QPs of the lane's TUs on the coefficient side: column block blkL, row blocks blk0 + r
#pragma unroll
    for( int r = 0; r < R; r++ ) { const int tu = tile * TPT + ( blk0 + r ) * TPS + blkL; qqs[r] = tu < A.n ? A.qps[tu] : vvhip_tu_qp{ 32, 0 }; }
    if( ( A.phaseLimit & 0xff ) == 2 ) { int k_ = 0; _Pragma( "unroll" ) for( int g = 0; g < 4; g++ ) k_ ^= aLo[g] ^ aHi[g]; if( k_ == 0x12345678 ) A.stats[0].pad = 1; continue; }

    int d[16];
    v4i bLo, bHi;
    // ---- forward rows: tmp[y][k] = sat16( ( sum_x blk[y][x] * Th[k][x] + rnd ) >> shift1 )                 (TrQuant.cpp:548)
    {
      v16i c;
#pragma unroll
      for( int v = 0; v < 16; v++ ) c[v] = cP1;
      const v4i opP1 = sOps[lane];
      MX_PASS( aHi, opP1, aLo, opP1, c, A.shF1 );
      mxSplitSat( d, bLo, bHi );
    }
    // ---- forward columns: coef[k2][k] = ( sum_y Tv[k2][y] * tmp[y][k] + rnd ) >> shift2                      (TrQuant.cpp:549)
    {
      v16i c;
#pragma unroll
      for( int g = 0; g < 4; g++ ) { const v4i t = *reinterpret_cast<const v4i*>( &sInit[h * 16 + 4 * g] ); c[4 * g] = t.x; c[4 * g + 1] = t.y; c[4 * g + 2] = t.z; c[4 * g + 3] = t.w; }
      const v4i opP2 = sOps[64 + lane];
      MX_PASS( opP2, bHi, opP2, bLo, c, A.shF2 );
    }
    if( ( A.phaseLimit & 0xff ) == 3 ) { TUMX_KEEP( d ); continue; }

    // ---- per TU: significance, QuantCore, DeQuantCore on the registers (the lane's TUs are walked in lockstep: their reduction chains overlap).
    // Significance: last non-zero scan position; largest magnitude (need-RDOQ is one test on it: the quantiser is monotonic); highest scan
    // position >= 16 whose magnitude passes the coefficient-group threshold |c| > thres / (4 * scale)  <=>  |c| * scale > thres >> 2 (such a
    // coefficient is non-zero, hence <= last).  The product is the one the level needs anyway (24-bit multiply while |c| < 2^16; larger
    // magnitudes take the 64-bit path and redo the test there)                                             (Quant.cpp:162-208, :264-278)
    uint32_t last[R], need[R], big[R];
    bool wide = false, someLevel = false;
#pragma unroll
    for( int r = 0; r < R; r++ )
    {
      const TuMxQ P = tuMxParams( A.q, qqs[r], A.thrVal );
      uint32_t l = 0, mx = 0, bg = 0;
      const int thr4 = P.thres >> 2;
#pragma unroll
      for( int v = r * VPR; v < ( r + 1 ) * VPR; v++ )
      {
        const uint32_t ac = ( uint32_t ) abs( d[v] );
        mx = ac > mx ? ac : mx;
        l = ( ac != 0 && pos[v] > l ) ? pos[v] : l;
        bg = ( ( int ) __umul24( ac, ( uint32_t ) P.scale ) > thr4 && pos[v] > bg ) ? pos[v] : bg;
      }
      { const uint32_t lb = tuMxGroupMaxPk16( l | ( bg << 16 ), G, lane ); l = lb & 0xffffu; bg = lb >> 16; }     // both < 1024
      mx = vvhipGroupMax32( mx, G, lane );
      need[r] = ( uint32_t ) ( ( int32_t ) ( ( ( int64_t ) mx * P.scale + P.addN ) >> P.qBits ) != 0 );
      // the TU's largest magnitude quantises to level 0 -> EVERY level of the TU is 0 (the quantiser is monotonic in |c|; the group threshold only removes levels)
      someLevel |= ( ( ( int64_t ) mx * P.scale + P.addQ ) >> P.qBits ) != 0;
      wide |= mx >= 65536u || P.qBits > 30 || P.qBits < 9;
      last[r] = l; big[r] = bg;
    }
    const bool narrow = __builtin_amdgcn_ballot_w64( wide ) == 0ull;
    if( !narrow )
#pragma unroll
      for( int r = 0; r < R; r++ )
      {
        const TuMxQ P = tuMxParams( A.q, qqs[r], A.thrVal );
        uint32_t bg = 0;
#pragma unroll
        for( int v = r * VPR; v < ( r + 1 ) * VPR; v++ )
          bg = ( ( long long ) abs( d[v] ) * ( P.scale << 2 ) > ( long long ) P.thres && pos[v] > bg ) ? pos[v] : bg;
        big[r] = vvhipGroupMax32( bg, G, lane );
      }
#pragma unroll
    for( int r = 0; r < R; r++ )
      if( last[r] >= 16 )
      {
        if( big[r] < 16 ) last[r] = 15;
        else if( ( big[r] >> 4 ) != ( last[r] >> 4 ) ) last[r] = ( big[r] >> 4 ) * 16 + 15;
      }
    if( ( A.phaseLimit & 0xff ) == 4 ) { TUMX_KEEP( d ); continue; }
    // Every TU of the tile quantises to all-zero levels (WAVE-UNIFORM; most TUs an encoder's RDO tries at its usual QPs: 72-98 % of the area of the recorded 1080p lists): the
    // dequantised coefficients are 0, both inverse passes give ( 0 + rnd ) >> shift = 0, the reconstructed residual is 0 and the SSE is the residual's energy — the quantiser
    // loop, the level staging and the two inverse passes are skipped; what is written is what the long way writes (levels 0, rec 0, abs sum 0, the same last / need-RDOQ).
    // The reference does the same one level up: TrQuant::invTransformNxN is only called for a non-zero abs sum (InterSearch.cpp:3696-3714).
    // SSE of the lane's 16 samples against the re-read residual + the reconstruction's raster stores + the TUs' sums.  When every residual and reconstructed sample of the wave is
    // within +-4095 (any residual of <= 12-bit video) a difference fits 14 bits, a lane's 16 squares fit 32 bits and the sum is eight packed subtractions + eight v_dot2_i32_i16;
    // otherwise (arbitrary int16 input) the 64-bit multiply-adds.  -4096 <= x <= 4095  <=>  ( uint16 ) ( x + 4096 ) < 8192: one packed add and one and-or per pair of samples
    typedef unsigned short u16x2t __attribute__( ( ext_vector_type( 2 ) ) );
#define TUMX_TAIL( STORE_REC ) { \
    unsigned long long sse[R]; \
_Pragma( "unroll" ) \
    for( int r = 0; r < R; r++ ) sse[r] = 0; \
    uint32_t rpAll[8]; \
    uint32_t magn = 0; \
_Pragma( "unroll" ) \
    for( int k = 0; k < 8; k++ ) \
    { \
      rpAll[k] = __builtin_bit_cast( uint32_t, __builtin_amdgcn_cvt_pk_i16( d[2 * k], d[2 * k + 1] ) ); \
      magn |= __builtin_bit_cast( uint32_t, __builtin_bit_cast( u16x2t, rpAll[k] ) + __builtin_bit_cast( u16x2t, 0x10001000u ) ) & 0xe000e000u; \
      magn |= __builtin_bit_cast( uint32_t, __builtin_bit_cast( u16x2t, xr2[k] ) + __builtin_bit_cast( u16x2t, 0x10001000u ) ) & 0xe000e000u; \
    } \
    const bool smallDiff = __builtin_amdgcn_ballot_w64( magn != 0 ) == 0ull; \
    uint32_t sse32[R]; \
_Pragma( "unroll" ) \
    for( int r = 0; r < R; r++ ) sse32[r] = 0; \
_Pragma( "unroll" ) \
    for( int c = 0; c < NP; c++ ) \
    { \
      const int X0 = 16 * h + PS * c; \
      const int tu = tile * TPT + blkL * TPS + X0 / N; \
      uint32_t rp[PS / 2]; \
_Pragma( "unroll" ) \
      for( int k = 0; k < PS / 2; k++ ) \
      { \
        const int xi = ( PS / 2 ) * c + k; \
        rp[k] = rpAll[xi]; \
        const int slot = ( PS * c ) / N < R ? ( PS * c ) / N : 0; \
        if( smallDiff ) \
        { \
          const uint32_t df = __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2, xr2[xi] ) - __builtin_bit_cast( s16x2, rp[k] ) ); \
          sse32[slot] = ( uint32_t ) dot2( df, df, ( int ) sse32[slot] ); \
        } \
        else \
        { \
          const int e0 = ( int ) ( int16_t ) ( xr2[xi] & 0xffff ) - ( int ) ( int16_t ) ( rp[k] & 0xffff ), e1 = ( ( int ) xr2[xi] >> 16 ) - ( ( int ) rp[k] >> 16 ); \
          sse[slot] += ( unsigned long long ) ( ( long long ) e0 * e0 ) + ( unsigned long long ) ( ( long long ) e1 * e1 ); \
        } \
      } \
      if( ( STORE_REC ) && A.rec && tu < A.n ) \
      { \
        int16_t* dst = A.rec + ( size_t ) tu * N * N + inL * N + X0 % N; \
        if( PS == 8 ) { u32x4 v; v.x = rp[0]; v.y = rp[1 % ( PS / 2 )]; v.z = rp[2 % ( PS / 2 )]; v.w = rp[3 % ( PS / 2 )]; *reinterpret_cast<u32x4*>( dst ) = v; } \
        else          { u32x2 v; v.x = rp[0]; v.y = rp[1 % ( PS / 2 )]; *reinterpret_cast<u32x2*>( dst ) = v; } \
      } \
    } \
_Pragma( "unroll" ) \
    for( int r = 0; r < R; r++ ) \
    { \
      const unsigned long long t = vvhipGroupSum64( smallDiff ? ( unsigned long long ) sse32[r] : sse[r], G, lane ); \
      const int tu = tile * TPT + blkL * TPS + blk0 + r; \
      if( A.stats && ( N < 32 || h == 0 ) && inL == r && tu < A.n ) A.stats[tu].sse = t; \
    } }
    const bool allZero = ( N >= TUMX_ZERO_MIN_N ) && __builtin_amdgcn_ballot_w64( someLevel ) == 0ull;
    uint32_t xr2[8];
    // the residual once more, for the SSE (L2 hit; cheaper than holding 8 registers through the quantiser)
#define TUMX_REREAD() { _Pragma( "unroll" ) for( int c = 0; c < NP; c++ )                                                                \
      {                                                                                                                                 \
        const int X0 = 16 * h + PS * c;                                                                                                 \
        const int tu = tile * TPT + blkL * TPS + X0 / N;                                                                                \
        const int16_t* src = resi + ( tu < A.n ? A.resiOff[tu] : 0 ) + ( ptrdiff_t ) inL * resiStride + X0 % N;                         \
        if( PS == 8 )                                                                                                                   \
        {                                                                                                                               \
          u32x4 v = { 0, 0, 0, 0 };                                                                                                     \
          if( tu < A.n ) v = reinterpret_cast<const U16*>( src )->v;                                                                    \
          xr2[4 * c] = v.x; xr2[( 4 * c + 1 ) & 7] = v.y; xr2[( 4 * c + 2 ) & 7] = v.z; xr2[( 4 * c + 3 ) & 7] = v.w;                   \
        }                                                                                                                               \
        else                                                                                                                            \
        {                                                                                                                               \
          u32x2 v = { 0, 0 };                                                                                                           \
          if( tu < A.n ) v = reinterpret_cast<const U8*>( src )->v;                                                                     \
          xr2[( 2 * c ) & 7] = v.x; xr2[( 2 * c + 1 ) & 7] = v.y;                                                                       \
        }                                                                                                                               \
      } }
    if( allZero )
    {
      // (keeping the first read's registers alive instead — -DTUMX_ZERO_KEEP_RESI=1 — makes all-zero 32-point lists 12 % faster on their own, 17.9 -> 15.7 us for 8 192 tiles,
      //  and the five-stream step of the recorded lists 5 % SLOWER, 66.2 -> 70.8 us: measured twice each in one call, profiles/r05_tu_zero_shortcut.log; not the default)
#if TUMX_ZERO_KEEP_RESI
#pragma unroll
      for( int k = 0; k < 8; k++ ) xr2[k] = xr[k];
#else
      TUMX_REREAD()
#endif
      //                                               // (requested first: the zero stores below pass under its latency)
#pragma unroll
      for( int r = 0; r < R; r++ )
      {
        const int tu = tile * TPT + ( blk0 + r ) * TPS + blkL;
        if( A.stats && ( N < 32 || h == 0 ) && inL == r && tu < A.n )
        {
          int32_t* st = reinterpret_cast<int32_t*>( A.stats + tu );
          st[0] = 0; st[1] = ( int32_t ) last[r]; st[2] = ( int32_t ) need[r]; st[3] = 0;
        }
      }
      // zeros for levels AND reconstruction in the raster mapping of the level stores (a wave instruction writes whole 64-byte runs; the long way's reconstruction stores go
      // row by row from the lanes that hold the rows: two half-filled requests per row)
      // (sparse outputs, vvhip_tu_set_sparse_outputs: a TU whose abs_sum is 0 gets NEITHER levels NOR reconstruction — the caller treats them as zero, as the reference does,
      //  which reads neither for such a TU, InterSearch.cpp:3696-3714; the tile then moves its residual in and 24 bytes per TU out)
      if( !( A.phaseLimit & 0x100 ) )
#pragma unroll
      for( int u = 0; u < 16 / PS; u++ )
      {
        const int q = lane + 64 * u, Y = q / ( 32 / PS ), X = PS * ( q % ( 32 / PS ) );
        const int tu = tile * TPT + ( Y / N ) * TPS + X / N;
        if( tu < A.n )
        {
          const size_t at = ( size_t ) tu * N * N + ( Y % N ) * N + X % N;
          if( PS == 8 ) { if( A.level ) *reinterpret_cast<u32x4*>( A.level + at ) = u32x4{ 0, 0, 0, 0 }; if( A.rec ) *reinterpret_cast<u32x4*>( A.rec + at ) = u32x4{ 0, 0, 0, 0 }; }
          else          { if( A.level ) *reinterpret_cast<u32x2*>( A.level + at ) = u32x2{ 0, 0 }; if( A.rec ) *reinterpret_cast<u32x2*>( A.rec + at ) = u32x2{ 0, 0 }; }
        }
      }
#pragma unroll
      for( int v = 0; v < 16; v++ ) d[v] = 0;
      TUMX_TAIL( false )
      continue;
    }
    // levels -> staging tile (raster), dequantised values replace the coefficients.  When every |c| fits 16 bits the level is a 24-bit
    // multiply-add in 32 bits (|c| * scale < 2^31, add < 2^29.5); otherwise the 64-bit form.  DeQuantCore's two shift directions are one
    // formula: a left shift is folded into the multiplier, a right shift carries its rounding offset.                (Quant.cpp:213-262)
    uint32_t absSum[R];
#define TUMX_LEVELS( MEXPR )                                                                                                           \
    _Pragma( "unroll" ) for( int r = 0; r < R; r++ )                                                                                   \
    {                                                                                                                                  \
      const TuMxQ P = tuMxParams( A.q, qqs[r], A.thrVal );                                                                             \
      uint32_t sum = 0;                                                                                                                \
      const int rsPos = P.rightShift > 0 ? P.rightShift : 0, rndDq = P.rightShift > 0 ? 1 << ( P.rightShift - 1 ) : 0;                \
      const int iscaleL = P.rightShift < 0 ? P.iscale << ( -P.rightShift ) : P.iscale;                                                 \
      _Pragma( "unroll" ) for( int v = r * VPR; v < ( r + 1 ) * VPR; v++ )                                                             \
      {                                                                                                                                \
        const int cv = d[v];                                                                                                           \
        const uint32_t ac = ( uint32_t ) abs( cv );                                                                                    \
        uint32_t m = MEXPR;                                                                                                            \
        m = pos[v] <= last[r] ? m : 0u;                                                                                                \
        sum += m;                                                                                                                      \
        const int sm = cv < 0 ? -( int32_t ) m : ( int32_t ) m;                                                                        \
        const int lv = clip3i( -32768, 32767, sm );                                                                                    \
        stage[( 16 * h + v ) * LP + c32] = ( int16_t ) lv;                                                                             \
        const int cl = med3i( lv, ~P.inMax, P.inMax );                                                                                 \
        const int32_t w_ = ( int32_t ) ( ( uint32_t ) __mul24( cl, iscaleL ) + ( uint32_t ) rndDq ) >> rsPos;   /* |cl| < 2^15, multiplier < 2^23 */ \
        d[v] = clip3i( -32768, 32767, w_ );                                                                                            \
      }                                                                                                                                \
      absSum[r] = sum;                                                                                                                 \
    }
    if( narrow )
    {
      // signed form: sign(c) * ( ( |c| * scale + add ) >> qBits )  ==  ( c * scale + ( c < 0 ? 2^qBits - 1 - add : add ) ) >> qBits (arithmetic),
      // because -floor( X / 2^q ) == floor( ( -X + 2^q - 1 ) / 2^q ); |c * scale| + add < 2^31 here.  |level| sums through v_sad_u32 on biased values.
#pragma unroll
      for( int r = 0; r < R; r++ )
      {
        const TuMxQ P = tuMxParams( A.q, qqs[r], A.thrVal );
        uint32_t sum = 0;
        const int addP = ( int ) P.addQ, addM = ( int ) ( ( 1u << P.qBits ) - 1u ) - addP;
        const int rsPos = P.rightShift > 0 ? P.rightShift : 0, rndDq = P.rightShift > 0 ? 1 << ( P.rightShift - 1 ) : 0;
        const int iscaleL = P.rightShift < 0 ? P.iscale << ( -P.rightShift ) : P.iscale;
#pragma unroll
        for( int v = r * VPR; v < ( r + 1 ) * VPR; v++ )
        {
          const int cv = d[v];
          int sm = ( __mul24( cv, P.scale ) + ( cv < 0 ? addM : addP ) ) >> P.qBits;
          sm = pos[v] <= last[r] ? sm : 0;
          sum = sadU32( ( uint32_t ) ( sm + ( 1 << 24 ) ), 1u << 24, sum );                                      // |sm| < 2^24 (qBits >= 9)
          const int lv = clip3i( -32768, 32767, sm );
          stage[( 16 * h + v ) * LP + c32] = ( int16_t ) lv;
          const int cl = med3i( lv, ~P.inMax, P.inMax );
          const int32_t w_ = ( int32_t ) ( ( uint32_t ) __mul24( cl, iscaleL ) + ( uint32_t ) rndDq ) >> rsPos;   // |cl| < 2^15, multiplier < 2^23
          d[v] = clip3i( -32768, 32767, w_ );
        }
        absSum[r] = sum;
      }
    }
    else { TUMX_LEVELS( ( uint32_t ) ( int32_t ) ( ( ( int64_t ) ac * P.scale + P.addQ ) >> P.qBits ) ) }
#undef TUMX_LEVELS
#pragma unroll
    for( int r = 0; r < R; r++ )
    {
      const uint32_t sum = vvhipGroupSum32( absSum[r], G, lane );
      const int tu = tile * TPT + ( blk0 + r ) * TPS + blkL;
      if( A.stats && ( N < 32 || h == 0 ) && inL == r && tu < A.n )
      {
        int32_t* st = reinterpret_cast<int32_t*>( A.stats + tu );
        st[0] = ( int32_t ) sum; st[1] = ( int32_t ) last[r]; st[2] = ( int32_t ) need[r]; st[3] = 0;
      }
    }
    if( ( A.phaseLimit & 0xff ) == 5 ) { TUMX_KEEP( d ); continue; }
    WAVE_SYNC();
    // ---- levels: staging rows -> raster in runs of PS samples (run q: row q / (32 / PS) of the tile, samples PS * (q % (32 / PS)) ..)
    if( A.level )
#pragma unroll
      for( int u = 0; u < 16 / PS; u++ )
      {
        const int q = lane + 64 * u, Y = q / ( 32 / PS ), X = PS * ( q % ( 32 / PS ) );
        const int tu = tile * TPT + ( Y / N ) * TPS + X / N;
        if( tu < A.n )
        {
          int16_t* dst = A.level + ( size_t ) tu * N * N + ( Y % N ) * N + X % N;
          if( PS == 8 ) *reinterpret_cast<u32x4*>( dst ) = *reinterpret_cast<const u32x4*>( &stage[Y * LP + X] );
          else          *reinterpret_cast<u32x2*>( dst ) = *reinterpret_cast<const u32x2*>( &stage[Y * LP + X] );
        }
      }
    WAVE_SYNC();

    if( ( A.phaseLimit & 0xff ) == 6 ) { TUMX_KEEP( d ); continue; }
    // ---- inverse columns: t1[y][k] = clip( ( sum_k2 deq[k2][k] * Tv[k2][y] + 64 ) >> 7 )                  (TrQuant.cpp:612)
    mxSplit( d, aLo, aHi );
    {
      v16i c;
#pragma unroll
      for( int v = 0; v < 16; v++ ) c[v] = cI1;
      const v4i opI1 = *reinterpret_cast<const v4i*>( A.opV->natT[lane] );
      MX_PASS( aHi, opI1, aLo, opI1, c, A.shI1 );
      mxSplitSat( d, bLo, bHi );
    }
    // ---- inverse rows: rec[y][x] = clip( ( sum_k t1[y][k] * Th[k][x] + rnd ) >> shift2 ); SSE against the residual (re-read: L2 hit, issued
    // ahead of the matrix products — cheaper than holding 8 registers through the quantiser)                                  (:613)
    TUMX_REREAD()
    {
      v16i c;
#pragma unroll
      for( int g = 0; g < 4; g++ ) { const v4i t = *reinterpret_cast<const v4i*>( &sInit[32 + h * 16 + 4 * g] ); c[4 * g] = t.x; c[4 * g + 1] = t.y; c[4 * g + 2] = t.z; c[4 * g + 3] = t.w; }
      const v4i opI2 = *reinterpret_cast<const v4i*>( A.opH->colP[lane] );
      MX_PASS( opI2, bHi, opI2, bLo, c, A.shI2 );
    }
    if( ( A.phaseLimit & 0xff ) == 7 ) { TUMX_KEEP( d ); continue; }
    TUMX_TAIL( true )
  }
#undef WAVE_SYNC
#undef TUMX_KEEP
#undef MX_PASS
#undef TUMX_REREAD
#undef TUMX_TAIL
}

// --------------------------------------------------------------------------------------------
// 64x64 TUs (DCT-2 only; 32x32 coefficients survive the zero-out): one wave per TU.  The residual is two row tiles of 32 rows, every row in two
// chunks of 32 samples; a contraction over 64 is two products accumulating into the same registers.  Quantiser section = the 32-point one.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void
tuMx64Body( int16_t* __restrict__ stage, int32_t* __restrict__ sInit /* [96] */, v4i* __restrict__ sOps /* [512] */, const int waveIndex,
            const int16_t* __restrict__ resi, const int resiStride, const TuMxArgs& A )
{
  constexpr int LP = 40;
  struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
  const VvhipTuMx64Ops* __restrict__ O = reinterpret_cast<const VvhipTuMx64Ops*>( A.opH );
  const int lane = threadIdx.x & 63, h = lane >> 5, c32 = lane & 31;
  const v16i zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define WAVE_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  const int rndF1 = 1 << ( A.shF1 - 1 ), rndF2 = 1 << ( A.shF2 - 1 ), rndI1 = 1 << ( A.shI1 - 1 ), rndI2 = 1 << ( A.shI2 - 1 );
  // operand slots in LDS: 0,1 natX[c]  2,3 rowPY[t]   (natTY[t], colPX[c] of the inverse passes: from the record, see below)
#pragma unroll
  for( int q = 0; q < 2; q++ )
  {
    sOps[( 0 + q ) * 64 + lane] = *reinterpret_cast<const v4i*>( O->natX[q][lane] );
    sOps[( 2 + q ) * 64 + lane] = *reinterpret_cast<const v4i*>( O->rowPY[q][lane] );
    // (the operands of the two INVERSE passes are read from the record when a TU gets that far — 2–12 % of the 64x64 TUs of the recorded lists have a level at all — instead of
    //  sitting in LDS: the instance needs the 4 KB of operand slots per wave the other sizes need, not 8: 44.5 -> 28.2 KB per workgroup, round 5)
  }
  const int cP1 = O->rowSum[c32] + rndF1;
  const int cI1[2] = { O->colSum[c32] + rndI1, O->colSum[32 + c32] + rndI1 };
  sInit[c32] = O->rowSum[c32] + rndF2;                         // forward columns: by result row k2 = 16h + v
  sInit[32 + lane] = O->colSum[lane] + rndI2;                  // inverse rows: by result column x = 32c + 16h + v
  uint32_t pos[16];
  {
    const u32x4 p0 = *reinterpret_cast<const u32x4*>( &O->pos[lane][0] ), p1 = *reinterpret_cast<const u32x4*>( &O->pos[lane][8] );
    const uint32_t pw[8] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w };
#pragma unroll
    for( int v = 0; v < 16; v++ ) pos[v] = ( v & 1 ) ? pw[v >> 1] >> 16 : pw[v >> 1] & 0xffffu;
  }
  WAVE_SYNC();

  for( int tu = waveIndex; tu < A.n; tu += A.waveStride )
  {
    const int16_t* src = resi + A.resiOff[tu];
    const vvhip_tu_qp qq = A.qps[tu];
    int d[16];
    v4i bLo[2], bHi[2];
    // ---- forward rows, two row tiles: tmp[y][k] = sat16( ( sum_{x<64} blk[y][x] * T[k][x] + rnd ) >> shift1 ), k < 32            (TrQuant.cpp:548)
#pragma unroll
    for( int t = 0; t < 2; t++ )
    {
      // (one accumulator: the high-byte products of both chunks, shifted up and joined with the preload, then the low-byte products on top — as MX_PASS of tuMxBody)
      v16i acc = zero16;
      v4i aLo[2];
#pragma unroll
      for( int c = 0; c < 2; c++ )
      {
        const int16_t* p = src + ( ptrdiff_t ) ( 32 * t + c32 ) * resiStride + 32 * c + 16 * h;
        const u32x4 x0 = reinterpret_cast<const U16*>( p )->v, x1 = reinterpret_cast<const U16*>( p + 8 )->v;
        const uint32_t xr[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
        v4i aHi;
#pragma unroll
        for( int g = 0; g < 4; g++ )
        {
          aLo[c][g] = ( int ) ( __builtin_amdgcn_perm( xr[2 * g + 1], xr[2 * g], 0x06040200u ) ^ 0x80808080u );
          aHi[g] = ( int ) __builtin_amdgcn_perm( xr[2 * g + 1], xr[2 * g], 0x07050301u );
        }
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( aHi, sOps[c * 64 + lane], acc, 0, 0, 0 );
      }
#pragma unroll
      for( int v = 0; v < 16; v++ ) acc[v] = ( acc[v] << 8 ) + cP1;
#pragma unroll
      for( int c = 0; c < 2; c++ ) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( aLo[c], sOps[c * 64 + lane], acc, 0, 0, 0 );
#pragma unroll
      for( int v = 0; v < 16; v++ ) d[v] = acc[v] >> A.shF1;
      mxSplitSat( d, bLo[t], bHi[t] );
    }
    // ---- forward columns: coef[k2][k] = ( sum_{y<64} T[k2][y] * tmp[y][k] + rnd ) >> shift2, k2 < 32                              (TrQuant.cpp:549)
    {
      v16i acc = zero16;
#pragma unroll
      for( int t = 0; t < 2; t++ ) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( sOps[( 2 + t ) * 64 + lane], bHi[t], acc, 0, 0, 0 );
#pragma unroll
      for( int g = 0; g < 4; g++ )
      {
        const v4i t4 = *reinterpret_cast<const v4i*>( &sInit[h * 16 + 4 * g] );
        acc[4 * g] = ( acc[4 * g] << 8 ) + t4.x; acc[4 * g + 1] = ( acc[4 * g + 1] << 8 ) + t4.y; acc[4 * g + 2] = ( acc[4 * g + 2] << 8 ) + t4.z; acc[4 * g + 3] = ( acc[4 * g + 3] << 8 ) + t4.w;
      }
#pragma unroll
      for( int t = 0; t < 2; t++ ) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( sOps[( 2 + t ) * 64 + lane], bLo[t], acc, 0, 0, 0 );
#pragma unroll
      for( int v = 0; v < 16; v++ ) d[v] = acc[v] >> A.shF2;
    }
    // ---- significance, QuantCore, DeQuantCore on the 32x32 coefficients: exactly the 32-point section of tuMxBody
    const TuMxQ P = tuMxParams( A.q, qq, A.thrVal );
    uint32_t last = 0, mx = 0, big = 0;
    const int thr4 = P.thres >> 2;
#pragma unroll
    for( int v = 0; v < 16; v++ )
    {
      const uint32_t ac = ( uint32_t ) abs( d[v] );
      mx = ac > mx ? ac : mx;
      last = ( ac != 0 && pos[v] > last ) ? pos[v] : last;
      big = ( ( int ) __umul24( ac, ( uint32_t ) P.scale ) > thr4 && pos[v] > big ) ? pos[v] : big;
    }
    { const uint32_t lb = tuMxGroupMaxPk16( last | ( big << 16 ), 64, lane ); last = lb & 0xffffu; big = lb >> 16; }
    mx = vvhipGroupMax32( mx, 64, lane );
    const uint32_t need = ( uint32_t ) ( ( int32_t ) ( ( ( int64_t ) mx * P.scale + P.addN ) >> P.qBits ) != 0 );
    const bool narrow = __builtin_amdgcn_ballot_w64( mx >= 65536u || P.qBits > 30 || P.qBits < 9 ) == 0ull;      // (wave-uniform: mx is the TU's maximum on every lane)
    if( !narrow )
    {
      big = 0;
#pragma unroll
      for( int v = 0; v < 16; v++ ) big = ( ( long long ) abs( d[v] ) * ( P.scale << 2 ) > ( long long ) P.thres && pos[v] > big ) ? pos[v] : big;
      big = vvhipGroupMax32( big, 64, lane );
    }
    if( last >= 16 )
    {
      if( big < 16 ) last = 15;
      else if( ( big >> 4 ) != ( last >> 4 ) ) last = ( big >> 4 ) * 16 + 15;
    }
    // the TU's largest magnitude quantises to level 0 -> every level is 0, the reconstructed residual is 0 and the SSE is the residual's energy (see tuMxBody): no quantiser
    // loop, no staging, no inverse passes — 88-98 % of the 64x64 TUs of the recorded 1080p lists
    if( __builtin_amdgcn_ballot_w64( ( ( ( int64_t ) mx * P.scale + P.addQ ) >> P.qBits ) != 0 ) == 0ull )
    {
      if( A.stats && lane == 0 )
      {
        int32_t* st = reinterpret_cast<int32_t*>( A.stats + tu );
        st[0] = 0; st[1] = ( int32_t ) last; st[2] = ( int32_t ) need; st[3] = 0;
      }
      unsigned long long sse0 = 0;
      typedef unsigned short u16x2z __attribute__( ( ext_vector_type( 2 ) ) );
#pragma nounroll
      for( int tc = 0; tc < 4; tc++ )                                  // (not unrolled: four chunks' requests at once would cost the long way's registers)
        {
          const int t = tc >> 1, c = tc & 1;
          const int16_t* p = src + ( ptrdiff_t ) ( 32 * t + c32 ) * resiStride + 32 * c + 16 * h;
          const u32x4 x0 = reinterpret_cast<const U16*>( p )->v, x1 = reinterpret_cast<const U16*>( p + 8 )->v;
          const uint32_t xr[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
          const u32x4 z4 = { 0, 0, 0, 0 };
          // (levels and reconstruction zeros in the same raster runs: whole 64-byte requests; 512 runs of 8 samples per array, 2 per lane and chunk)
          const int q0 = lane + 64 * ( 4 * t + 2 * c );
          const bool dense = !( A.phaseLimit & 0x100 );      // (sparse outputs: nothing but the statistics for a TU whose levels are all zero, see tuMxBody)
          if( A.rec && dense )
          {
            *reinterpret_cast<u32x4*>( A.rec + ( size_t ) tu * 4096 + ( size_t ) q0 * 8 ) = z4;
            *reinterpret_cast<u32x4*>( A.rec + ( size_t ) tu * 4096 + ( size_t ) ( q0 + 64 ) * 8 ) = z4;
          }
          if( A.level && dense )
          {
            *reinterpret_cast<u32x4*>( A.level + ( size_t ) tu * 4096 + ( size_t ) q0 * 8 ) = z4;
            *reinterpret_cast<u32x4*>( A.level + ( size_t ) tu * 4096 + ( size_t ) ( q0 + 64 ) * 8 ) = z4;
          }
          uint32_t magn = 0;
#pragma unroll
          for( int k = 0; k < 8; k++ ) magn |= __builtin_bit_cast( uint32_t, __builtin_bit_cast( u16x2z, xr[k] ) + __builtin_bit_cast( u16x2z, 0x10001000u ) ) & 0xe000e000u;
          if( __builtin_amdgcn_ballot_w64( magn != 0 ) == 0ull )
          {
            uint32_t s32 = 0;
#pragma unroll
            for( int k = 0; k < 8; k++ ) s32 = ( uint32_t ) dot2( xr[k], xr[k], ( int ) s32 );
            sse0 += s32;
          }
          else
#pragma unroll
            for( int k = 0; k < 8; k++ )
            {
              const int e0 = ( int ) ( int16_t ) ( xr[k] & 0xffff ), e1 = ( int ) xr[k] >> 16;
              sse0 += ( unsigned long long ) ( ( long long ) e0 * e0 ) + ( unsigned long long ) ( ( long long ) e1 * e1 );
            }
        }
      sse0 = vvhipGroupSum64( sse0, 64, lane );
      if( A.stats && lane == 0 ) A.stats[tu].sse = sse0;
      continue;
    }
    uint32_t sum = 0;
    {
      const int addP = ( int ) P.addQ, addM = ( int ) ( ( 1u << ( P.qBits & 31 ) ) - 1u ) - addP;
      const int rsPos = P.rightShift > 0 ? P.rightShift : 0, rndDq = P.rightShift > 0 ? 1 << ( P.rightShift - 1 ) : 0;
      const int iscaleL = P.rightShift < 0 ? P.iscale << ( -P.rightShift ) : P.iscale;
#define TU64_LEVEL( SMEXPR )                                                                                                         \
      _Pragma( "unroll" ) for( int v = 0; v < 16; v++ )                                                                                 \
      {                                                                                                                                 \
        const int cv = d[v];                                                                                                            \
        int sm = SMEXPR;                                                                                                                \
        sm = pos[v] <= last ? sm : 0;                                                                                                   \
        sum += ( uint32_t ) abs( sm );                                                                                                  \
        const int lv = clip3i( -32768, 32767, sm );                                                                                     \
        stage[( 16 * h + v ) * LP + c32] = ( int16_t ) lv;                                                                              \
        const int cl = med3i( lv, ~P.inMax, P.inMax );                                                                                  \
        const int32_t w_ = ( int32_t ) ( ( uint32_t ) __mul24( cl, iscaleL ) + ( uint32_t ) rndDq ) >> rsPos;                           \
        d[v] = clip3i( -32768, 32767, w_ );                                                                                             \
      }
      if( narrow ) { TU64_LEVEL( ( __mul24( cv, P.scale ) + ( cv < 0 ? addM : addP ) ) >> P.qBits ) }
      else { TU64_LEVEL( ( cv < 0 ? -( int ) ( ( ( int64_t ) abs( cv ) * P.scale + P.addQ ) >> P.qBits ) : ( int ) ( ( ( int64_t ) abs( cv ) * P.scale + P.addQ ) >> P.qBits ) ) ) }
#undef TU64_LEVEL
    }
    sum = vvhipGroupSum32( sum, 64, lane );
    if( A.stats && lane == 0 )
    {
      int32_t* st = reinterpret_cast<int32_t*>( A.stats + tu );
      st[0] = ( int32_t ) sum; st[1] = ( int32_t ) last; st[2] = ( int32_t ) need; st[3] = 0;
    }
    WAVE_SYNC();
    // ---- levels: 64x64 raster, the 32x32 region from the staging tile, zeros elsewhere (512 runs of 8 samples, 8 per lane)
    if( A.level )
#pragma unroll
      for( int u = 0; u < 8; u++ )
      {
        const int q = lane + 64 * u, Y = q >> 3, X = 8 * ( q & 7 );
        u32x4 v = { 0, 0, 0, 0 };
        if( Y < 32 && X < 32 ) v = *reinterpret_cast<const u32x4*>( &stage[Y * LP + X] );
        *reinterpret_cast<u32x4*>( A.level + ( size_t ) tu * 4096 + Y * 64 + X ) = v;
      }
    WAVE_SYNC();
    // ---- inverse columns, two row tiles: t1[y][k] = clip( ( sum_{k2<32} deq[k2][k] * T[k2][y] + 64 ) >> 7 )                       (TrQuant.cpp:612)
    v4i aLo, aHi;
    mxSplit( d, aLo, aHi );
#pragma unroll
    for( int t = 0; t < 2; t++ )
    {
      v16i c;
#pragma unroll
      for( int v = 0; v < 16; v++ ) c[v] = cI1[t];
      const v4i op = *reinterpret_cast<const v4i*>( O->natTY[t][lane] );
      v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( aHi, op, zero16, 0, 0, 0 );
#pragma unroll
      for( int v = 0; v < 16; v++ ) acc[v] = ( acc[v] << 8 ) + c[v];
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( aLo, op, acc, 0, 0, 0 );
#pragma unroll
      for( int v = 0; v < 16; v++ ) d[v] = acc[v] >> A.shI1;
      mxSplitSat( d, bLo[t], bHi[t] );
    }
    // ---- inverse rows: rec[y][x] = clip( ( sum_{k<32} t1[y][k] * T[k][x] + rnd ) >> shift2 ), row tile t x column chunk c; SSE vs the residual (:613)
    unsigned long long sse = 0;
#pragma unroll
    for( int t = 0; t < 2; t++ )
#pragma unroll
      for( int c = 0; c < 2; c++ )
      {
        const int16_t* p = src + ( ptrdiff_t ) ( 32 * t + c32 ) * resiStride + 32 * c + 16 * h;
        const u32x4 x0 = reinterpret_cast<const U16*>( p )->v, x1 = reinterpret_cast<const U16*>( p + 8 )->v;
        const uint32_t xr[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
        v16i ci;
#pragma unroll
        for( int g = 0; g < 4; g++ ) { const v4i t4 = *reinterpret_cast<const v4i*>( &sInit[32 + 32 * c + h * 16 + 4 * g] ); ci[4 * g] = t4.x; ci[4 * g + 1] = t4.y; ci[4 * g + 2] = t4.z; ci[4 * g + 3] = t4.w; }
        const v4i op = *reinterpret_cast<const v4i*>( O->colPX[c][lane] );
        v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( op, bHi[t], zero16, 0, 0, 0 );
#pragma unroll
        for( int v = 0; v < 16; v++ ) acc[v] = ( acc[v] << 8 ) + ci[v];
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8( op, bLo[t], acc, 0, 0, 0 );
        // SSE as in tuMxBody: residual and reconstruction within +-4095 on every lane (any residual of <= 12-bit video) -> packed subtraction + v_dot2_i32_i16 in 32 bits
        // (16 squares < 2^30), else the 64-bit multiply-adds
        typedef unsigned short u16x2t __attribute__( ( ext_vector_type( 2 ) ) );
        uint32_t rp[8], magn = 0;
#pragma unroll
        for( int k = 0; k < 8; k++ )
        {
          rp[k] = __builtin_bit_cast( uint32_t, __builtin_amdgcn_cvt_pk_i16( acc[2 * k] >> A.shI2, acc[2 * k + 1] >> A.shI2 ) );
          magn |= __builtin_bit_cast( uint32_t, __builtin_bit_cast( u16x2t, rp[k] ) + __builtin_bit_cast( u16x2t, 0x10001000u ) ) & 0xe000e000u;
          magn |= __builtin_bit_cast( uint32_t, __builtin_bit_cast( u16x2t, xr[k] ) + __builtin_bit_cast( u16x2t, 0x10001000u ) ) & 0xe000e000u;
        }
        if( __builtin_amdgcn_ballot_w64( magn != 0 ) == 0ull )
        {
          uint32_t s32 = 0;
#pragma unroll
          for( int k = 0; k < 8; k++ )
          {
            const uint32_t df = __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2, xr[k] ) - __builtin_bit_cast( s16x2, rp[k] ) );
            s32 = ( uint32_t ) dot2( df, df, ( int ) s32 );
          }
          sse += s32;
        }
        else
#pragma unroll
          for( int k = 0; k < 8; k++ )
          {
            const int e0 = ( int ) ( int16_t ) ( xr[k] & 0xffff ) - ( int ) ( int16_t ) ( rp[k] & 0xffff ), e1 = ( ( int ) xr[k] >> 16 ) - ( ( int ) rp[k] >> 16 );
            sse += ( unsigned long long ) ( ( long long ) e0 * e0 ) + ( unsigned long long ) ( ( long long ) e1 * e1 );
          }
        if( A.rec )
        {
          int16_t* dst = A.rec + ( size_t ) tu * 4096 + ( 32 * t + c32 ) * 64 + 32 * c + 16 * h;
          u32x4 a, b; a.x = rp[0]; a.y = rp[1]; a.z = rp[2]; a.w = rp[3]; b.x = rp[4]; b.y = rp[5]; b.z = rp[6]; b.w = rp[7];
          *reinterpret_cast<u32x4*>( dst ) = a; *reinterpret_cast<u32x4*>( dst + 8 ) = b;
        }
      }
    sse = vvhipGroupSum64( sse, 64, lane );
    if( A.stats && lane == 0 ) A.stats[tu].sse = sse;
  }
#undef WAVE_SYNC
}

// --------------------------------------------------------------------------------------------
// 64x64 TUs over TWO waves (round 4): wave p of the pair owns row tile p — rows 32p .. 32p + 31 — in both row passes (forward rows, inverse rows: independent per
// row), and contributes its tile's half of the forward column pass (a contraction over the 64 rows = the sum of the two tiles' products): the two halves meet through LDS, after
// which BOTH waves hold the 32x32 coefficients and run the quantiser section on them (redundantly: it is a quarter of the wave's work and needs every coefficient for its
// last-position / coefficient-group decisions), then each does the inverse column and row passes of its own tile.  Per TU a wave issues 12 instead of 24 matrix products and half
// of the byte splits, clips, SSE terms and stores.  The pair meets twice per TU (coefficient halves, SSE) on an LDS counter: the two waves belong to one workgroup, so both are
// resident; no workgroup barrier (the other pair of the workgroup runs its own TUs).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void tuMxPairSync( uint32_t* ctr, uint32_t& gen, int lane )
{
  __builtin_amdgcn_fence( __ATOMIC_RELEASE, "workgroup" );
  if( lane == 0 ) atomicAdd( ctr, 1u );
  gen += 2;
  while( *reinterpret_cast<volatile uint32_t*>( ctr ) < gen ) __builtin_amdgcn_s_sleep( 1 );
  __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "workgroup" );
}

__device__ __forceinline__ void
tuMx64PairBody( int16_t* __restrict__ stage, int32_t* __restrict__ sInit /* [96] */, v4i* __restrict__ sOps /* [512] */, int32_t* __restrict__ xch /* [2][64][17]: the pair's exchange area */,
                uint32_t* __restrict__ ctr, const int p /* 0 / 1: which wave of the pair */, const int pairIndex, const int pairStride,
                const int16_t* __restrict__ resi, const int resiStride, const TuMxArgs& A )
{
  constexpr int LP = 40, XP = 17;                 // (exchange rows of 17 dwords: lanes hit different banks)
  struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
  const VvhipTuMx64Ops* __restrict__ O = reinterpret_cast<const VvhipTuMx64Ops*>( A.opH );
  const int lane = threadIdx.x & 63, h = lane >> 5, c32 = lane & 31;
  const v16i zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define WAVE_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  const int rndF1 = 1 << ( A.shF1 - 1 ), rndF2 = 1 << ( A.shF2 - 1 ), rndI1 = 1 << ( A.shI1 - 1 ), rndI2 = 1 << ( A.shI2 - 1 );
  // operand slots (as tuMx64Body): 0,1 natX[c]  2 rowPY[p]  4 natTY[p]  6,7 colPX[c]
#pragma unroll
  for( int q = 0; q < 2; q++ )
  {
    sOps[( 0 + q ) * 64 + lane] = *reinterpret_cast<const v4i*>( O->natX[q][lane] );
    sOps[( 6 + q ) * 64 + lane] = *reinterpret_cast<const v4i*>( O->colPX[q][lane] );
  }
  sOps[2 * 64 + lane] = *reinterpret_cast<const v4i*>( O->rowPY[p][lane] );
  sOps[4 * 64 + lane] = *reinterpret_cast<const v4i*>( O->natTY[p][lane] );
  const int cP1 = O->rowSum[c32] + rndF1;
  const int cI1 = O->colSum[32 * p + c32] + rndI1;
  sInit[c32] = p == 0 ? O->rowSum[c32] + rndF2 : 0;            // forward columns: correction + rounding enter the sum once (wave 0's half)
  sInit[32 + lane] = O->colSum[lane] + rndI2;                  // inverse rows: by result column x = 32c + 16h + v
  uint32_t pos[16];
  {
    const u32x4 p0 = *reinterpret_cast<const u32x4*>( &O->pos[lane][0] ), p1 = *reinterpret_cast<const u32x4*>( &O->pos[lane][8] );
    const uint32_t pw[8] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w };
#pragma unroll
    for( int v = 0; v < 16; v++ ) pos[v] = ( v & 1 ) ? pw[v >> 1] >> 16 : pw[v >> 1] & 0xffffu;
  }
  WAVE_SYNC();
  uint32_t gen = 0;
  int32_t* xMine = xch + p * 64 * XP + lane * XP;
  const int32_t* xOther = xch + ( p ^ 1 ) * 64 * XP + lane * XP;

  for( int tu = pairIndex; tu < A.n; tu += pairStride )
  {
    const int16_t* src = resi + A.resiOff[tu] + ( ptrdiff_t ) ( 32 * p ) * resiStride;      // the wave's row tile
    const vvhip_tu_qp qq = A.qps[tu];
    int d[16];
    v4i bLo, bHi;
    // ---- forward rows of the own tile: tmp[y][k] = sat16( ( sum_{x<64} blk[y][x] * T[k][x] + rnd ) >> shift1 ), k < 32            (TrQuant.cpp:548)
    {
      v16i lo, hi = zero16;
#pragma unroll
      for( int v = 0; v < 16; v++ ) lo[v] = cP1;
#pragma unroll
      for( int c = 0; c < 2; c++ )
      {
        const int16_t* q = src + ( ptrdiff_t ) c32 * resiStride + 32 * c + 16 * h;
        const u32x4 x0 = reinterpret_cast<const U16*>( q )->v, x1 = reinterpret_cast<const U16*>( q + 8 )->v;
        const uint32_t xr[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
        v4i aLo, aHi;
#pragma unroll
        for( int g = 0; g < 4; g++ )
        {
          aLo[g] = ( int ) ( __builtin_amdgcn_perm( xr[2 * g + 1], xr[2 * g], 0x06040200u ) ^ 0x80808080u );
          aHi[g] = ( int ) __builtin_amdgcn_perm( xr[2 * g + 1], xr[2 * g], 0x07050301u );
        }
        const v4i op = sOps[c * 64 + lane];
        lo = __builtin_amdgcn_mfma_i32_32x32x32_i8( aLo, op, lo, 0, 0, 0 );
        hi = __builtin_amdgcn_mfma_i32_32x32x32_i8( aHi, op, hi, 0, 0, 0 );
      }
#pragma unroll
      for( int v = 0; v < 16; v++ ) d[v] = ( ( hi[v] << 8 ) + lo[v] ) >> A.shF1;
      mxSplitSat( d, bLo, bHi );
    }
    // ---- forward columns, the own tile's half of the contraction over y: the halves meet in LDS; coef[k2][k] = ( sum_{y<64} T[k2][y] * tmp[y][k] + rnd ) >> shift2  (:549)
    {
      v16i lo, hi = zero16;
#pragma unroll
      for( int g = 0; g < 4; g++ ) { const v4i t4 = *reinterpret_cast<const v4i*>( &sInit[h * 16 + 4 * g] ); lo[4 * g] = t4.x; lo[4 * g + 1] = t4.y; lo[4 * g + 2] = t4.z; lo[4 * g + 3] = t4.w; }
      const v4i op = sOps[2 * 64 + lane];
      lo = __builtin_amdgcn_mfma_i32_32x32x32_i8( op, bLo, lo, 0, 0, 0 );
      hi = __builtin_amdgcn_mfma_i32_32x32x32_i8( op, bHi, hi, 0, 0, 0 );
#pragma unroll
      for( int v = 0; v < 16; v++ ) { d[v] = ( hi[v] << 8 ) + lo[v]; xMine[v] = d[v]; }
      tuMxPairSync( ctr, gen, lane );
#pragma unroll
      for( int v = 0; v < 16; v++ ) d[v] = ( d[v] + xOther[v] ) >> A.shF2;
    }
    // ---- significance, QuantCore, DeQuantCore on the 32x32 coefficients (both waves, identical): exactly the 32-point section of tuMxBody
    const TuMxQ P = tuMxParams( A.q, qq, A.thrVal );
    uint32_t last = 0, mx = 0, big = 0;
    const int thr4 = P.thres >> 2;
#pragma unroll
    for( int v = 0; v < 16; v++ )
    {
      const uint32_t ac = ( uint32_t ) abs( d[v] );
      mx = ac > mx ? ac : mx;
      last = ( ac != 0 && pos[v] > last ) ? pos[v] : last;
      big = ( ( int ) __umul24( ac, ( uint32_t ) P.scale ) > thr4 && pos[v] > big ) ? pos[v] : big;
    }
    { const uint32_t lb = tuMxGroupMaxPk16( last | ( big << 16 ), 64, lane ); last = lb & 0xffffu; big = lb >> 16; }
    mx = vvhipGroupMax32( mx, 64, lane );
    const uint32_t need = ( uint32_t ) ( ( int32_t ) ( ( ( int64_t ) mx * P.scale + P.addN ) >> P.qBits ) != 0 );
    const bool narrow = !( mx >= 65536u || P.qBits > 30 || P.qBits < 9 );
    if( !narrow )
    {
      big = 0;
#pragma unroll
      for( int v = 0; v < 16; v++ ) big = ( ( long long ) abs( d[v] ) * ( P.scale << 2 ) > ( long long ) P.thres && pos[v] > big ) ? pos[v] : big;
      big = vvhipGroupMax32( big, 64, lane );
    }
    if( last >= 16 )
    {
      if( big < 16 ) last = 15;
      else if( ( big >> 4 ) != ( last >> 4 ) ) last = ( big >> 4 ) * 16 + 15;
    }
    uint32_t sum = 0;
    {
      const int addP = ( int ) P.addQ, addM = ( int ) ( ( 1u << ( P.qBits & 31 ) ) - 1u ) - addP;
      const int rsPos = P.rightShift > 0 ? P.rightShift : 0, rndDq = P.rightShift > 0 ? 1 << ( P.rightShift - 1 ) : 0;
      const int iscaleL = P.rightShift < 0 ? P.iscale << ( -P.rightShift ) : P.iscale;
#pragma unroll
      for( int v = 0; v < 16; v++ )
      {
        const int cv = d[v];
        int sm;
        if( narrow ) sm = ( __mul24( cv, P.scale ) + ( cv < 0 ? addM : addP ) ) >> P.qBits;
        else { const int m = ( int ) ( ( ( int64_t ) abs( cv ) * P.scale + P.addQ ) >> P.qBits ); sm = cv < 0 ? -m : m; }
        sm = pos[v] <= last ? sm : 0;
        sum += ( uint32_t ) abs( sm );
        const int lv = clip3i( -32768, 32767, sm );
        stage[( 16 * h + v ) * LP + c32] = ( int16_t ) lv;
        const int cl = med3i( lv, ~P.inMax, P.inMax );
        const int32_t w_ = ( int32_t ) ( ( uint32_t ) __mul24( cl, iscaleL ) + ( uint32_t ) rndDq ) >> rsPos;
        d[v] = clip3i( -32768, 32767, w_ );
      }
    }
    sum = vvhipGroupSum32( sum, 64, lane );
    if( A.stats && p == 0 && lane == 0 )
    {
      int32_t* st = reinterpret_cast<int32_t*>( A.stats + tu );
      st[0] = ( int32_t ) sum; st[1] = ( int32_t ) last; st[2] = ( int32_t ) need; st[3] = 0;
    }
    WAVE_SYNC();
    // ---- levels: 64x64 raster, the 32x32 region from the (own) staging tile, zeros elsewhere: 512 runs of 8 samples, the wave stores rows 32p .. 32p + 31 (4 runs per lane)
    if( A.level )
#pragma unroll
      for( int u = 0; u < 4; u++ )
      {
        const int q = lane + 64 * ( 4 * p + u ), Y = q >> 3, X = 8 * ( q & 7 );
        u32x4 v = { 0, 0, 0, 0 };
        if( Y < 32 && X < 32 ) v = *reinterpret_cast<const u32x4*>( &stage[Y * LP + X] );
        *reinterpret_cast<u32x4*>( A.level + ( size_t ) tu * 4096 + Y * 64 + X ) = v;
      }
    WAVE_SYNC();
    // ---- inverse columns of the own tile: t1[y][k] = clip( ( sum_{k2<32} deq[k2][k] * T[k2][y] + 64 ) >> 7 ), y in the tile                 (TrQuant.cpp:612)
    v4i aLo, aHi;
    mxSplit( d, aLo, aHi );
    {
      v16i c;
#pragma unroll
      for( int v = 0; v < 16; v++ ) c[v] = cI1;
      const v4i op = sOps[4 * 64 + lane];
      const v16i lo = __builtin_amdgcn_mfma_i32_32x32x32_i8( aLo, op, c, 0, 0, 0 );
      const v16i hi = __builtin_amdgcn_mfma_i32_32x32x32_i8( aHi, op, zero16, 0, 0, 0 );
#pragma unroll
      for( int v = 0; v < 16; v++ ) d[v] = ( ( hi[v] << 8 ) + lo[v] ) >> A.shI1;
      mxSplitSat( d, bLo, bHi );
    }
    // ---- inverse rows of the own tile x the two column chunks; SSE vs the residual (:613)
    unsigned long long sse = 0;
#pragma unroll
    for( int c = 0; c < 2; c++ )
    {
      const int16_t* q = src + ( ptrdiff_t ) c32 * resiStride + 32 * c + 16 * h;
      const u32x4 x0 = reinterpret_cast<const U16*>( q )->v, x1 = reinterpret_cast<const U16*>( q + 8 )->v;
      const uint32_t xr[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
      v16i ci;
#pragma unroll
      for( int g = 0; g < 4; g++ ) { const v4i t4 = *reinterpret_cast<const v4i*>( &sInit[32 + 32 * c + h * 16 + 4 * g] ); ci[4 * g] = t4.x; ci[4 * g + 1] = t4.y; ci[4 * g + 2] = t4.z; ci[4 * g + 3] = t4.w; }
      const v4i op = sOps[( 6 + c ) * 64 + lane];
      const v16i lo = __builtin_amdgcn_mfma_i32_32x32x32_i8( op, bLo, ci, 0, 0, 0 );
      const v16i hi = __builtin_amdgcn_mfma_i32_32x32x32_i8( op, bHi, zero16, 0, 0, 0 );
      uint32_t rp[8];
#pragma unroll
      for( int k = 0; k < 8; k++ )
      {
        const int v0 = ( ( hi[2 * k] << 8 ) + lo[2 * k] ) >> A.shI2, v1 = ( ( hi[2 * k + 1] << 8 ) + lo[2 * k + 1] ) >> A.shI2;
        rp[k] = __builtin_bit_cast( uint32_t, __builtin_amdgcn_cvt_pk_i16( v0, v1 ) );
        const int e0 = ( int ) ( int16_t ) ( xr[k] & 0xffff ) - ( int ) ( int16_t ) ( rp[k] & 0xffff ), e1 = ( ( int ) xr[k] >> 16 ) - ( ( int ) rp[k] >> 16 );
        sse += ( unsigned long long ) ( ( long long ) e0 * e0 ) + ( unsigned long long ) ( ( long long ) e1 * e1 );
      }
      if( A.rec )
      {
        int16_t* dst = A.rec + ( size_t ) tu * 4096 + ( 32 * p + c32 ) * 64 + 32 * c + 16 * h;
        u32x4 a, b; a.x = rp[0]; a.y = rp[1]; a.z = rp[2]; a.w = rp[3]; b.x = rp[4]; b.y = rp[5]; b.z = rp[6]; b.w = rp[7];
        *reinterpret_cast<u32x4*>( dst ) = a; *reinterpret_cast<u32x4*>( dst + 8 ) = b;
      }
    }
    sse = vvhipGroupSum64( sse, 64, lane );
    // the two tiles' SSE meet in the exchange area (slot 16 of lane 0's row: not part of the coefficient exchange), the first wave stores the TU's sum
    if( lane == 0 ) { xMine[16] = ( int32_t ) ( uint32_t ) sse; xMine[XP + 16] = ( int32_t ) ( uint32_t ) ( sse >> 32 ); }
    tuMxPairSync( ctr, gen, lane );
    if( A.stats && p == 0 && lane == 0 )
    {
      const unsigned long long o = ( unsigned long long ) ( uint32_t ) xOther[16] | ( ( unsigned long long ) ( uint32_t ) xOther[XP + 16] << 32 );
      A.stats[tu].sse = sse + o;
    }
  }
#undef WAVE_SYNC
}

struct TuMxJobs { int nJobs; int pair64; int waveStart[8]; int size[8]; TuMxArgs j[8]; };

// Three instances, by what the launch's lists need: KIND 0 the 8/16/32-point bodies, KIND 1 also the 4-point body (four TUs per lane), KIND 2 also the 64-point body.
// Registers (round 5, tools/kernel_regs.py): 148 / 164 / 166, no scratch — every instance runs three waves per SIMD.  Round 4 held 168 (1 spill) / 168 (33 spills) / 232 (two waves
// per SIMD for EVERY size of a launch with 64x64 TUs): each 1-D pass kept the high-byte and the low-byte product in two 16-register accumulators side by side; the passes now run the
// high-byte product first and the low-byte product on top of it (MX_PASS), and the 64-point body takes its SSE through the packed 32-bit form like the others.
// Measured (recorded 1080p mix, same box): the TU launch alone 17.75 -> 16.9 us, the five-stream step 73.0 -> 69.1 us (the freed registers let the other streams' waves share the SIMDs).
// Throughput of the 32-point lists is 467 tiles/us = 80 % of the VALU issue bound (1 059 wave instructions per tile): the instruction count, not occupancy, is what is left.
// KIND 3 = KIND 2 with a 64x64 TU over a wave pair (tuMx64PairBody, $VVHIP_TU_PAIR64=1: measured slower, not the default) — its own instance because the pair's exchange area
// is 17 KB of LDS per workgroup that the default launch must not reserve (the TU workgroups share their CUs with the motion-search kernels of the other streams).
template<int KIND>
__global__ void __launch_bounds__( 256, KIND == 3 ? 2 : 3 )
tuMxMultiKernel( const int16_t* __restrict__ resi, int resiStride, TuMxJobs jobs )
{
  constexpr bool WITH4 = KIND >= 1, WITH64 = KIND >= 2, PAIR64 = KIND == 3;
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t stage[4][32 * 40];
  __shared__ __attribute__( ( aligned( 16 ) ) ) int32_t sInit[4][WITH64 ? 96 : 64];
  __shared__ v4i sOps[4][PAIR64 ? 512 : ( WITH64 ? 256 : 128 )];      // forward-pass operands per wave (8/16/32-point: 2 slots, 64-point: 4; the pair form keeps all 8)
  __shared__ int32_t xch[PAIR64 ? 2 : 1][PAIR64 ? 2 * 64 * 17 : 1];      // 64x64 TUs over a wave pair: the pair's exchange area
  __shared__ uint32_t pairCtr[2];
  if( PAIR64 ) { if( threadIdx.x < 2 ) pairCtr[threadIdx.x] = 0; __syncthreads(); }      // (before any wave leaves)
  const int wv = __builtin_amdgcn_readfirstlane( ( int ) ( threadIdx.x >> 6 ) );
  const int wave = blockIdx.x * 4 + wv;
  int k = 0;
#pragma unroll
  for( int i = 1; i < 8; i++ ) if( i < jobs.nJobs && wave >= jobs.waveStart[i] ) k = i;
  const int w = wave - jobs.waveStart[k];
  if( w >= jobs.j[k].waveStride ) return;
  const int rs = jobs.j[k].resiStride ? jobs.j[k].resiStride : resiStride;
  if( jobs.size[k] == 32 )      tuMxBody<32>( stage[wv], sInit[wv], sOps[wv], w, resi, rs, jobs.j[k] );
  else if( jobs.size[k] == 16 ) tuMxBody<16>( stage[wv], sInit[wv], sOps[wv], w, resi, rs, jobs.j[k] );
  else if( jobs.size[k] == 8 )  tuMxBody<8>( stage[wv], sInit[wv], sOps[wv], w, resi, rs, jobs.j[k] );
  else if( WITH4 && jobs.size[k] == 4 )   tuMxBody<4>( stage[wv], sInit[wv], sOps[wv], w, resi, rs, jobs.j[k] );
  else if( WITH64 && jobs.size[k] == 64 )
  {
    // two waves per 64x64 TU: waves 2i, 2i + 1 of the job = the pair ( wv & ~1, wv | 1 ) of this workgroup (the host keeps the job's first wave and wave count even)
    if( PAIR64 ) tuMx64PairBody( stage[wv], sInit[wv], sOps[wv], xch[wv >> 1], &pairCtr[wv >> 1], w & 1, w >> 1, jobs.j[k].waveStride >> 1, resi, rs, jobs.j[k] );
    else         tuMx64Body( stage[wv], sInit[wv], sOps[wv], w, resi, rs, jobs.j[k] );
  }
}

__global__ void __launch_bounds__( 256 )
dequantCoreKernel( int maxX, int maxY, int scale, const int16_t* __restrict__ q, size_t qStride, int32_t* __restrict__ coef, int rightShift, int inMax, int32_t trMax )
{
  const int w = maxX + 1, total = w * ( maxY + 1 );
  for( int n = blockIdx.x * blockDim.x + threadIdx.x; n < total; n += gridDim.x * blockDim.x )
  {
    const int y = n / w, x = n - y * w;
    const int c = clip3i( -( inMax + 1 ), inMax, ( int ) q[x + y * qStride] );                                            // Quant.cpp:243
    int32_t v;
    if( rightShift > 0 ) v = ( int32_t ) ( ( uint32_t ) ( c * scale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;    // :244
    else                 v = ( int32_t ) ( ( uint32_t ) ( c * scale ) << ( -rightShift ) );                              // :257
    coef[n] = clip3i( -( trMax + 1 ), trMax, v );
  }
}

__global__ void __launch_bounds__( 256 )
needRdoqCoreKernel( const int32_t* __restrict__ coef, size_t num, int quantCoeff, long long offset, int shift, uint8_t* __restrict__ need )
{
  int any = 0;
  for( size_t i = threadIdx.x; i < num; i += blockDim.x ) any |= ( int32_t ) ( ( ( int64_t ) abs( coef[i] ) * quantCoeff + offset ) >> shift ) != 0;   // Quant.cpp:264-278
  __shared__ int sAny;
  if( threadIdx.x == 0 ) sAny = 0;
  __syncthreads();
  if( any ) sAny = 1;
  __syncthreads();
  if( threadIdx.x == 0 ) *need = ( uint8_t ) sAny;
}

// ------------------------------------------------------------------------------------------------------------------
bool makeGeom( int w, int h, int trHor, int trVer, int bitDepth, bool inverse, TrGeom& g )
{
  if( !isPow2( w ) || !isPow2( h ) || w < 2 || h < 2 || w > 64 || h > 64 ) return false;
  g.w = w; g.h = h; g.log2w = ilog2i( w ); g.log2h = ilog2i( h );
  auto okType = []( int t, int n ) { return t == VVHIP_DCT2 ? true : ( ( t == VVHIP_DCT8 || t == VVHIP_DST7 ) && n >= 4 && n <= 32 ); };
  if( !okType( trHor, w ) || !okType( trVer, h ) ) return false;
  g.skipW = ( trHor != VVHIP_DCT2 && w == 32 ) ? 16 : ( w > 32 ? w - 32 : 0 );
  g.skipH = ( trVer != VVHIP_DCT2 && h == 32 ) ? 16 : ( h > 32 ? h - 32 : 0 );
  if( !inverse ) { g.shift1 = g.log2w + bitDepth + 6 - 15; g.shift2 = g.log2h + 6; if( g.shift1 < 0 ) return false; }   // TrQuant.cpp:544-545
  else           { g.shift1 = 6 + 1; g.shift2 = ( 6 + 15 - 1 ) - bitDepth; if( g.shift2 < 1 ) return false; }           // TrQuant.cpp:608-609
  return true;
}

bool makeQGeom( int w, int h, int bitDepth, QGeom& q )
{
  if( !isPow2( w ) || !isPow2( h ) || w < 1 || h < 1 || w > 64 || h > 64 || w * h < 2 ) return false;
  q.w = w; q.h = h; q.log2w = ilog2i( w ); q.log2h = ilog2i( h ); q.bitDepth = bitDepth;
  int cgw, cgh;
  vvhip_cg_size( q.log2w, q.log2h, &cgw, &cgh );
  q.log2CG = cgw + cgh;
  q.cgIs4x4 = ( q.log2CG == 4 && cgw == 2 ) ? 1 : 0;
  q.numScan = ( w < 32 ? w : 32 ) * ( h < 32 ? h : 32 );
  return true;
}

int rowPitch( int n )   // odd number of 16-byte (8-sample) chunks per row: conflict-free ds_read_b128 across consecutive rows
{
  const int ch = n < 8 ? n : 8;
  int pu = n / ch;
  if( pu > 1 && !( pu & 1 ) ) pu++;
  return pu * ch;
}

TuLay makeLayout( const TrGeom& g, int trHor, int trVer )
{
  TuLay y;
  y.pw = rowPitch( g.w ); y.ph = rowPitch( g.h );
  const int area = g.w * g.h;
  int r = g.h * y.pw > g.w * y.ph ? g.h * y.pw : g.w * y.ph;
  if( r < area ) r = area;
  r = ( r + 7 ) & ~7;
  y.r1 = r; y.r2 = 2 * r; y.slot = 2 * r + 2 * area;
  const bool same = g.w == g.h && trHor == trVer;
  y.mTh = 0; y.mTv = same ? 0 : g.w * g.w;
  const int nat = same ? g.w * g.w : g.w * g.w + g.h * g.h;
  y.mThT = nat; y.mTvT = same ? nat : nat + g.w * g.w;
  y.matElems = 2 * nat;
  return y;
}

size_t trSmemBytes( const TuLay& y, int tpb )
{
  return ( size_t ) ( ( ( y.matElems + 7 ) & ~7 ) + ( ( tpb * y.slot + 7 ) & ~7 ) ) * sizeof( int16_t ) + ( size_t ) tpb * 320 + 64;   // + per-TU TuPar (312 B)
}

int launchTuRdo( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* d_resi_off, int n, const TrGeom& gf, const TrGeom& gi, const TuLay& y,
                 const QGeom& q, int tpb, int tr_hor, int tr_ver, const vvhip_tu_qp* d_qp, int thr_val, int16_t* d_level, int16_t* d_rec_resi, vvhip_tu_stats* d_stats );

int teamLog2( int positions ) { int l = 0; while( ( 1 << ( l + 1 ) ) <= positions && l < 6 ) l++; return l; }

// ---------------------------------------------------------------------------------------------
// The g_tCoeffOps table slots one-to-one (TrQuant_EMT.h:63-91): 1-D matrix cores with the CALLER's matrix, round/clip and the
// Pel <-> TCoeff copies.  The fused 2-D kernels above are what the batched path uses; these exist so that every slot of the
// table has a device provider with the reference's signature.  One thread per output, 32-bit wrap-around sums.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__( 256 )
fastFwdCoreKernel( int trSize, const int16_t* __restrict__ tc, const int32_t* __restrict__ src, int32_t* __restrict__ dst, unsigned line, unsigned reducedLine, unsigned cutoff, int shift )
{
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= reducedLine * cutoff ) return;
  const unsigned j = idx / reducedLine, i = idx - j * reducedLine;
  uint32_t sum = 0;
  for( int k = 0; k < trSize; k++ ) sum += ( uint32_t ) src[i * trSize + k] * ( uint32_t ) ( int32_t ) tc[j * trSize + k];     // TrQuant_EMT.cpp:1987-1991
  dst[j * line + i] = ( int32_t ) ( sum + ( 1u << ( shift - 1 ) ) ) >> shift;
}

__global__ void __launch_bounds__( 256 )
fastInvCoreKernel( int trSize, const int16_t* __restrict__ it, const int32_t* __restrict__ src, int32_t* __restrict__ dst, unsigned lines, unsigned reducedLines, unsigned rows )
{
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= reducedLines * trSize ) return;
  const unsigned i = idx / trSize, j = idx - i * trSize;
  uint32_t sum = ( uint32_t ) dst[idx];                                                                                       // accumulates (:1964)
  for( unsigned k = 0; k < rows; k++ ) sum += ( uint32_t ) src[k * lines + i] * ( uint32_t ) ( int32_t ) it[k * trSize + j];
  dst[idx] = ( int32_t ) sum;
}

__global__ void __launch_bounds__( 256 )
roundClipKernel( int32_t* __restrict__ dst, unsigned w, unsigned h, unsigned stride, int32_t mn, int32_t mx, int32_t round, int32_t shift )
{
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= w * h ) return;
  const unsigned y = idx / w, x = idx - y * w;
  const int32_t v = ( int32_t ) ( ( uint32_t ) dst[y * stride + x] + ( uint32_t ) round ) >> shift;                            // clipCore :1943
  dst[y * stride + x] = v < mn ? mn : ( v > mx ? mx : v );
}

__global__ void __launch_bounds__( 256 )
cpyResiKernel( const int32_t* __restrict__ src, int16_t* __restrict__ dst, ptrdiff_t stride, unsigned w, unsigned h )
{
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= w * h ) return;
  const unsigned y = idx / w, x = idx - y * w;
  dst[( ptrdiff_t ) y * stride + x] = ( int16_t ) src[idx];
}

__global__ void __launch_bounds__( 256 )
cpyCoeffKernel( const int16_t* __restrict__ src, ptrdiff_t stride, int32_t* __restrict__ dst, unsigned w, unsigned h )
{
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= w * h ) return;
  const unsigned y = idx / w, x = idx - y * w;
  dst[idx] = src[( ptrdiff_t ) y * stride + x];
}

} // namespace

// fused square-TU kernel: 0 = matrix cores (default), 1 = dot-product row kernel ($VVHIP_TU_KERNEL=row)
static int tuKernelForm() { const char* e = getenv( "VVHIP_TU_KERNEL" ); return e && !strcmp( e, "row" ) ? 1 : 0; }     // read per call: tests switch it
static int tuRepeat() { static const int v = getenv( "VVHIP_TU_REPEAT" ) ? atoi( getenv( "VVHIP_TU_REPEAT" ) ) : 2; return v < 1 ? 1 : v; }     // groups per workgroup
// profiling aid (tools/tuphase.py): stop the fused kernels after phase k — changes the results, so it only exists in builds with -DVVHIP_DEV_KNOBS
#ifdef VVHIP_DEV_KNOBS
static int tuPhaseLimit() { static const int v = getenv( "VVHIP_TU_PHASES" ) ? atoi( getenv( "VVHIP_TU_PHASES" ) ) : 0; return v; }
#else
static int tuPhaseLimit() { return 0; }
#endif

extern "C" {


int vvhip_fwd_transform_batch( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* d_resi_off,
                               int n, int width, int height, int tr_hor, int tr_ver, int bit_depth, int32_t* d_coef )
{
  if( !ctx ) return VVHIP_E_ARG;
  TrGeom g;
  if( n < 0 || !makeGeom( width, height, tr_hor, tr_ver, bit_depth, false, g ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_fwd_transform_batch: unsupported %dx%d types (%d,%d) bitDepth %d", width, height, tr_hor, tr_ver, bit_depth );
  if( n == 0 ) return VVHIP_OK;
  const int area = width * height;
  const int tpb = area >= 256 ? 1 : 256 / area;
  const TuLay y = makeLayout( g, tr_hor, tr_ver );
  const size_t smem = trSmemBytes( y, tpb );
  hipLaunchKernelGGL( fwdTransformKernel, dim3( ( n + tpb - 1 ) / tpb ), dim3( 256 ), smem, ctx->stream,
                      d_resi, resi_stride, d_resi_off, n, g, y, tpb,
                      ctx->d_trMat + trMatOffset( tr_hor, g.log2w ), ctx->d_trMat + trMatOffset( tr_ver, g.log2h ), d_coef );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_inv_transform_batch( vvhip_ctx* ctx, const int32_t* d_coef, int n, int width, int height, int tr_hor, int tr_ver, int bit_depth,
                               int16_t* d_resi, int resi_stride, const int32_t* d_resi_off )
{
  if( !ctx ) return VVHIP_E_ARG;
  TrGeom g;
  if( n < 0 || !makeGeom( width, height, tr_hor, tr_ver, bit_depth, true, g ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_inv_transform_batch: unsupported %dx%d types (%d,%d) bitDepth %d", width, height, tr_hor, tr_ver, bit_depth );
  if( n == 0 ) return VVHIP_OK;
  const int area = width * height;
  const int tpb = area >= 256 ? 1 : 256 / area;
  const TuLay y = makeLayout( g, tr_hor, tr_ver );
  const size_t smem = trSmemBytes( y, tpb );
  hipLaunchKernelGGL( invTransformKernel, dim3( ( n + tpb - 1 ) / tpb ), dim3( 256 ), smem, ctx->stream,
                      d_coef, n, g, y, tpb, ctx->d_trMat + trMatOffset( tr_hor, g.log2w ), ctx->d_trMat + trMatOffset( tr_ver, g.log2h ),
                      d_resi, resi_stride, d_resi_off );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_quant_batch( vvhip_ctx* ctx, const int32_t* d_coef, int n, int width, int height, int bit_depth, const vvhip_tu_qp* d_qp, int thr_val,
                       int16_t* d_level, int32_t* d_delta_u, int32_t* d_abs_sum, int32_t* d_last_scan_pos )
{
  if( !ctx ) return VVHIP_E_ARG;
  QGeom q;
  if( n < 0 || !makeQGeom( width, height, bit_depth, q ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_quant_batch: unsupported TU %dx%d", width, height );
  if( n == 0 ) return VVHIP_OK;
  const int log2Lpc = teamLog2( q.numScan );
  const long threads = ( long ) n << log2Lpc;
  hipLaunchKernelGGL( quantKernel, dim3( ( unsigned ) ( ( threads + 255 ) / 256 ) ), dim3( 256 ), 0, ctx->stream,
                      d_coef, n, q, log2Lpc, d_qp, 0, 0, 0ll, thr_val, ctx->d_scan + scanOffset( q.log2w, q.log2h ), d_level, d_delta_u, d_abs_sum, d_last_scan_pos );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_quant_core_lfnst( vvhip_ctx* ctx, const int32_t* d_coef, int width, int height, int quant_coeff, int q_bits, int64_t add, int thr_val, int lfnst_idx,
                            int16_t* d_level, int32_t* d_delta_u, int32_t* d_abs_sum, int32_t* d_last_scan_pos )
{
  if( !ctx ) return VVHIP_E_ARG;
  QGeom q;
  if( !makeQGeom( width, height, 10, q ) || q_bits < 9 || lfnst_idx < 0 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_quant_core: unsupported TU %dx%d / qBits %d / lfnstIdx %d", width, height, q_bits, lfnst_idx );
  const int log2Lpc = teamLog2( q.numScan );
  hipLaunchKernelGGL( quantKernel, dim3( 1 ), dim3( 256 ), 0, ctx->stream,
                      d_coef, 1, q, log2Lpc, ( const vvhip_tu_qp* ) nullptr, quant_coeff, q_bits, ( long long ) add, thr_val,
                      ctx->d_scan + scanOffset( q.log2w, q.log2h ), d_level, d_delta_u, d_abs_sum, d_last_scan_pos, lfnst_idx > 0 ? 1 : 0 );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_quant_core( vvhip_ctx* ctx, const int32_t* d_coef, int width, int height, int quant_coeff, int q_bits, int64_t add, int thr_val,
                      int16_t* d_level, int32_t* d_delta_u, int32_t* d_abs_sum, int32_t* d_last_scan_pos )
{
  return vvhip_quant_core_lfnst( ctx, d_coef, width, height, quant_coeff, q_bits, add, thr_val, 0, d_level, d_delta_u, d_abs_sum, d_last_scan_pos );
}

int vvhip_dequant_batch( vvhip_ctx* ctx, const int16_t* d_level, int n, int width, int height, int bit_depth, const vvhip_tu_qp* d_qp, int32_t* d_coef )
{
  if( !ctx ) return VVHIP_E_ARG;
  QGeom q;
  if( n < 0 || !makeQGeom( width, height, bit_depth, q ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dequant_batch: unsupported TU %dx%d", width, height );
  if( n == 0 ) return VVHIP_OK;
  const long total = ( long ) n * width * height;
  long blocks = ( total + 255 ) / 256; if( blocks > 8192 ) blocks = 8192;
  hipLaunchKernelGGL( dequantKernel, dim3( ( unsigned ) blocks ), dim3( 256 ), 0, ctx->stream, d_level, total, q, d_qp, d_coef );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_need_rdoq_batch( vvhip_ctx* ctx, const int32_t* d_coef, int n, int width, int height, int bit_depth, const vvhip_tu_qp* d_qp, uint8_t* d_need )
{
  if( !ctx ) return VVHIP_E_ARG;
  QGeom q;
  if( n < 0 || !makeQGeom( width, height, bit_depth, q ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_need_rdoq_batch: unsupported TU %dx%d", width, height );
  if( n == 0 ) return VVHIP_OK;
  const int log2Lpc = teamLog2( width * ( height < 32 ? height : 32 ) );
  const long threads = ( long ) n << log2Lpc;
  hipLaunchKernelGGL( needRdoqKernel, dim3( ( unsigned ) ( ( threads + 255 ) / 256 ) ), dim3( 256 ), 0, ctx->stream, d_coef, n, q, log2Lpc, d_qp, d_need );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_dequant_core( vvhip_ctx* ctx, int max_x, int max_y, int scale, const int16_t* d_level, size_t level_stride, int32_t* d_coef,
                        int right_shift, int input_maximum, int32_t transform_maximum )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( max_x < 0 || max_y < 0 || max_x > 127 || max_y > 127 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dequant_core: bad block" );
  const int total = ( max_x + 1 ) * ( max_y + 1 );
  hipLaunchKernelGGL( dequantCoreKernel, dim3( ( total + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, max_x, max_y, scale, d_level, level_stride, d_coef,
                      right_shift, input_maximum, transform_maximum );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

static int tuRdoMulti( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* strides, int bit_depth, const vvhip_tu_job* jobs, int n_jobs );

int vvhip_tu_rdo_multi( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, int bit_depth, const vvhip_tu_job* jobs, int n_jobs )
{
  return tuRdoMulti( ctx, d_resi, resi_stride, nullptr, bit_depth, jobs, n_jobs );
}

int vvhip_tu_set_sparse_outputs( vvhip_ctx* ctx, int on )
{
  if( !ctx ) return VVHIP_E_ARG;
  ctx->tuSparse = on != 0;
  return VVHIP_OK;
}

int vvhip_tu_rdo_multi_strided( vvhip_ctx* ctx, const int16_t* d_resi, const int32_t* resi_strides_host, int bit_depth, const vvhip_tu_job* jobs, int n_jobs )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n_jobs && !resi_strides_host ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_tu_rdo_multi_strided: no strides" );
  return tuRdoMulti( ctx, d_resi, 0, resi_strides_host, bit_depth, jobs, n_jobs );
}

static int tuRdoMulti( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* strides, int bit_depth, const vvhip_tu_job* jobs, int n_jobs )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n_jobs < 0 || ( n_jobs && !jobs ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_tu_rdo_multi: bad job table" );
  // square 8/16/32 TUs share the row-per-lane kernel: up to 4 of them go into one launch, largest size first; anything else runs alone
  auto mergeable = []( const vvhip_tu_job& j ) { return j.n > 0 && j.width == j.height && ( j.width == 8 || j.width == 16 || j.width == 32 ||
                                                        ( tuKernelForm() == 0 && ( j.width == 4 || ( j.width == 64 && j.tr_hor == VVHIP_DCT2 && j.tr_ver == VVHIP_DCT2 ) ) ) ); };     // 4x4 / 64x64 only in the matrix-core form
  int order[64], nm = 0;
  std::vector<TuGenJob> gen;
  for( int i = 0; i < n_jobs; i++ )
  {
    if( jobs[i].n < 0 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_tu_rdo_multi: job %d has n < 0", i );
    if( mergeable( jobs[i] ) && nm < 64 && !getenv( "VVHIP_TU_GENERIC" ) && ( !strides || tuKernelForm() == 0 ) ) order[nm++] = i;
    else if( jobs[i].n > 0 )
    {
      static const bool separate = getenv( "VVHIP_TU_GENERIC_SEPARATE" ) != nullptr;      // (A/B: one launch per list, the round-3 route)
      const vvhip_tu_job& jb = jobs[i];
      const bool rowForm = jb.width == jb.height && ( jb.width == 8 || jb.width == 16 || jb.width == 32 );      // (only reached with the row kernels selected: they have their own launches)
      if( separate || rowForm || gen.size() >= 64 )
      {
        const int rc = vvhip_tu_rdo_batch( ctx, d_resi, strides ? strides[i] : resi_stride, jb.d_resi_off, jb.n, jb.width, jb.height, jb.tr_hor, jb.tr_ver, bit_depth,
                                           jb.d_qp, jb.thr_val, jb.d_level, jb.d_rec_resi, jb.d_stats );
        if( rc ) return rc;
        continue;
      }
      // everything the matrix-core kernel does not take goes into ONE launch of the generic pipeline (tuRdoGenMultiKernel)
      TuGenJob g; memset( &g, 0, sizeof( g ) );
      if( !makeGeom( jb.width, jb.height, jb.tr_hor, jb.tr_ver, bit_depth, false, g.gf ) || !makeGeom( jb.width, jb.height, jb.tr_hor, jb.tr_ver, bit_depth, true, g.gi ) || !makeQGeom( jb.width, jb.height, bit_depth, g.q ) )
        return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_tu_rdo_multi: unsupported %dx%d types (%d,%d) bitDepth %d", jb.width, jb.height, jb.tr_hor, jb.tr_ver, bit_depth );
      g.y = makeLayout( g.gf, jb.tr_hor, jb.tr_ver );
      const int area = jb.width * jb.height;
      g.tpb = area >= 1024 ? 1 : 1024 / area;          // ~1024 samples per workgroup (as vvhip_tu_rdo_batch's generic route)
      g.resiStride = strides ? strides[i] : resi_stride; g.n = jb.n; g.thrVal = jb.thr_val;
      g.resiOff = jb.d_resi_off; g.matH = ctx->d_trMat + trMatOffset( jb.tr_hor, g.gf.log2w ); g.matV = ctx->d_trMat + trMatOffset( jb.tr_ver, g.gf.log2h );
      g.scan = ctx->d_scan + scanOffset( g.q.log2w, g.q.log2h ); g.qps = jb.d_qp; g.level = jb.d_level; g.rec = jb.d_rec_resi; g.stats = jb.d_stats;
      gen.push_back( g );
    }
  }
  if( !gen.empty() )
  {
    // largest workgroups (LDS, duration) first; the table lives in device memory and is re-uploaded only when it differs from the previous call's (a caller replays a picture's lists)
    std::stable_sort( gen.begin(), gen.end(), []( const TuGenJob& a, const TuGenJob& b ) { return a.gf.w * a.gf.h > b.gf.w * b.gf.h; } );
    size_t smem = 0; int blocks = 0;
    for( TuGenJob& g : gen ) { g.blockStart = blocks; blocks += ( g.n + g.tpb - 1 ) / g.tpb; smem = std::max( smem, trSmemBytes( g.y, g.tpb ) ); }
    const size_t bytes = gen.size() * sizeof( TuGenJob );
    if( ctx->tuGenBytes < bytes )
    {
      if( ctx->d_tuGen ) { VVHIP_CHECK_HIP( ctx, hipStreamSynchronize( ctx->stream ) ); ( void ) hipFree( ctx->d_tuGen ); ctx->d_tuGen = nullptr; ctx->tuGenBytes = 0; }
      VVHIP_CHECK_HIP( ctx, hipMalloc( &ctx->d_tuGen, 64 * sizeof( TuGenJob ) ) );
      ctx->tuGenBytes = 64 * sizeof( TuGenJob ); ctx->tuGenLast.clear();
    }
    // (the cache is keyed on the table's content AND the stream it was uploaded on: a caller that switched the context's stream may still have the previous launch reading the
    //  table on the old stream.  The old stream may be a borrowed handle that no longer exists (vvhip_set_stream), so it is never touched: the new stream waits for the EVENT the
    //  context recorded behind that launch, then the table is uploaded again; the bookkeeping is updated before anything can fail.  ADVICE r4 / r5)
    if( ctx->tuGenStream != ctx->stream )
    {
      const bool wait = !ctx->tuGenLast.empty() && ctx->tuGenEventRecorded;
      ctx->tuGenLast.clear(); ctx->tuGenStream = ctx->stream;
      if( wait ) VVHIP_CHECK_HIP( ctx, hipStreamWaitEvent( ctx->stream, ctx->tuGenEvent, 0 ) );
    }
    if( ctx->tuGenLast.size() != bytes || memcmp( ctx->tuGenLast.data(), gen.data(), bytes ) != 0 )
    {
      // (stream-ordered behind the previous launch that still reads the old table; the host copy is staged before the call returns)
      ctx->tuGenLast.assign( reinterpret_cast<const unsigned char*>( gen.data() ), reinterpret_cast<const unsigned char*>( gen.data() ) + bytes );
      VVHIP_CHECK_HIP( ctx, hipMemcpyAsync( ctx->d_tuGen, ctx->tuGenLast.data(), bytes, hipMemcpyHostToDevice, ctx->stream ) );
    }
    if( smem > 64 * 1024 ) VVHIP_CHECK_HIP( ctx, hipFuncSetAttribute( ( const void* ) tuRdoGenMultiKernel, hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) smem ) );
    hipLaunchKernelGGL( tuRdoGenMultiKernel, dim3( ( unsigned ) blocks ), dim3( 256 ), smem, ctx->stream, d_resi, static_cast<const TuGenJob*>( ctx->d_tuGen ), ( int ) gen.size(), tuPhaseLimit() );
    VVHIP_LAUNCH_CHECK( ctx );
    if( !ctx->tuGenEvent ) VVHIP_CHECK_HIP( ctx, hipEventCreateWithFlags( &ctx->tuGenEvent, hipEventDisableTiming ) );
    VVHIP_CHECK_HIP( ctx, hipEventRecord( ctx->tuGenEvent, ctx->stream ) );
    ctx->tuGenEventRecorded = true;
  }
  // launch groups, largest size first (a wave of the largest size runs longest: it has to start first).  Matrix-core form: up to 8 jobs per launch; the 4x4 and 64x64
  // variants live in a second kernel instance (more code, same register bound) — a picture's lists go into ONE launch of that instance when they are small (a recorded
  // B picture is ~3 000 waves: three launches of a few hundred waves each would mostly be ramp-up and drain), into one launch per kind otherwise.  Row form: 4 jobs, 8/16/32 only.
  const bool mx = tuKernelForm() == 0;
  const int groupMax = mx ? 8 : 4;
  auto isExtra = [&]( int i ) { return jobs[order[i]].width == 4 || jobs[order[i]].width == 64; };
  auto tilesOf = [&]( int i ) { const vvhip_tu_job& jb = jobs[order[i]]; const int tpt = jb.width == 64 ? 1 : ( 32 / jb.width ) * ( 32 / jb.width ); return ( long ) ( jb.n + tpt - 1 ) / tpt; };
  long allTiles = 0;
  for( int i = 0; i < nm; i++ ) allTiles += tilesOf( i );
  // (round 5, recorded 4K lists, profiles/r05_tu_launch_sweep.log: all of a picture's lists in ONE launch with a budget of 8 192 waves instead of two launches of 4 096 / 3 072:
  //  TU time per picture 66 -> 52 us, five-stream step 244 -> 227 us; 1080p unchanged.  The limits below only split what is far beyond a 4K picture)
  static const long oneLaunchTiles = getenv( "VVHIP_TU_ONE_LAUNCH_TILES" ) ? atol( getenv( "VVHIP_TU_ONE_LAUNCH_TILES" ) ) : 65536;
  static const long repeat1Tiles = getenv( "VVHIP_TU_REPEAT1_TILES" ) ? atol( getenv( "VVHIP_TU_REPEAT1_TILES" ) ) : 16384;
  const bool oneLaunch = mx && nm <= groupMax && allTiles <= oneLaunchTiles;
  // short lists first: a size that only a handful of waves run finds its code in no instruction cache (the five bodies are 74 KB) and those waves take several times a warm
  // wave's duration — started first, they finish under the long lists instead of after them
  static const long smallFirstTiles = getenv( "VVHIP_TU_SMALL_FIRST" ) ? atol( getenv( "VVHIP_TU_SMALL_FIRST" ) ) : 256;
  auto isShort = [&]( int i ) { return oneLaunch && tilesOf( i ) < smallFirstTiles; };
  auto before = [&]( int a, int b ) { const bool ea = isExtra( a ), eb = isExtra( b ), sa = isShort( a ), sb = isShort( b );
                                      return ( !oneLaunch && ea != eb ) ? !ea : ( sa != sb ? sa : jobs[order[a]].width > jobs[order[b]].width ); };
  for( int a = 1; a < nm; a++ ) for( int b = a; b > 0 && before( b, b - 1 ); b-- ) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  int nRegular = 0;
  while( !oneLaunch && nRegular < nm && !isExtra( nRegular ) ) nRegular++;
  for( int first = 0, next = 0; first < nm; first = next )
  {
    next = first + groupMax;
    if( !oneLaunch && first < nRegular && next > nRegular ) next = nRegular;          // do not mix the two kinds
    const int groupEnd = next < nm ? next : nm;

    TuMultiJobs mj; mj.nJobs = 0;
    TuRowArgs rowArgs[8];
    int nJobs = 0;
    long blocks = 0, groupTiles = 0;
    for( int i = first; i < groupEnd; i++ ) groupTiles += tilesOf( i );
    // tiles per wave (matrix-core form): the launch should be ONE resident round of waves (3 per SIMD) with balanced work — a second round that is a tenth full costs a whole
    // wave duration.  Work budget B per wave in 32x32-tile units (a 64x64 TU counts 4): the smallest B for which the launch fits; capped, long lists simply take several rounds.
    static const long residentEnv = getenv( "VVHIP_TU_RESIDENT_WAVES" ) ? atol( getenv( "VVHIP_TU_RESIDENT_WAVES" ) ) : 0;
    // (round 4, when the 64-point instance held 2 048 waves; on the recorded mix 64:955 32:2 133 16:600 8:800 4:600 = 3 298 tiles a budget for 2 048 / 3 072 / 4 096 waves gives 22.4 / 20.7 / 17.6 us:
    //  with its long 64x64 waves in front, one tile per wave and a second partial round beat fewer, longer waves)
    const long residentWaves = residentEnv ? residentEnv : 8192;      // (round 4: 4 096 with 64-point lists at two waves per SIMD, 3 072 without; every instance holds three now)
    int budget = 0;
    // $VVHIP_TU_PAIR64=1: a 64x64 TU as a PAIR of waves (tuMx64PairBody), each worth two 32x32-tile units.  Measured (round 4, tools/tu_mix.py) and NOT the default: one TU
    // per launch slot 8.8 -> 7.85 us, but 955 TUs 9.9 -> 10.7 us and the recorded mix 64:955 32:2133 16:600 8:800 4:600 18.5 -> 21.8 us — a wave's life is its set-up and its
    // memory latencies (operand records, residual rows), which a second wave repeats instead of halving, and at two waves per SIMD the doubled wave count costs a further round
    static const bool pair64 = getenv( "VVHIP_TU_PAIR64" ) && atoi( getenv( "VVHIP_TU_PAIR64" ) ) == 1;
    auto wavesPer = [&]( int i ) { return ( jobs[order[i]].width == 64 && pair64 ) ? 2 : 1; };
    auto repeatOf = [&]( int i ) { const int work = jobs[order[i]].width == 64 ? ( pair64 ? 2 : 4 ) : 1; const int r = budget / work; return r < 1 ? 1 : r; };
    if( mx && groupTiles <= repeat1Tiles * 4 )
    {
      for( budget = 1; budget < 8; budget++ )
      {
        long waves = 0;
        for( int i = first; i < groupEnd; i++ ) waves += wavesPer( i ) * ( ( tilesOf( i ) + repeatOf( i ) - 1 ) / repeatOf( i ) );
        if( waves <= residentWaves ) break;
      }
    }
    for( int i = first; i < groupEnd; i++ )
    {
      const vvhip_tu_job& jb = jobs[order[i]];
      TuRowArgs& ra = rowArgs[nJobs];
      if( !makeGeom( jb.width, jb.height, jb.tr_hor, jb.tr_ver, bit_depth, false, ra.gf ) || !makeGeom( jb.width, jb.height, jb.tr_hor, jb.tr_ver, bit_depth, true, ra.gi ) ||
          !makeQGeom( jb.width, jb.height, bit_depth, ra.q ) )
        return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_tu_rdo_multi: unsupported %dx%d types (%d,%d) bitDepth %d", jb.width, jb.height, jb.tr_hor, jb.tr_ver, bit_depth );
      ra.resiOff = jb.d_resi_off; ra.n = jb.n;
      ra.matH = ctx->d_trMat + trMatOffset( jb.tr_hor, ra.gf.log2w ); ra.matV = ctx->d_trMat + trMatOffset( jb.tr_ver, ra.gf.log2h );
      ra.scan = ctx->d_scan + scanOffset( ra.q.log2w, ra.q.log2h );
      ra.qps = jb.d_qp; ra.thrVal = jb.thr_val; ra.level = jb.d_level; ra.rec = jb.d_rec_resi; ra.stats = jb.d_stats; ra.phaseLimit = tuPhaseLimit();
      if( !mx )
      {
        const int tpb = 256 / ( jb.width * ( jb.width == 8 ? 1 : 2 ) );
        mj.blockStart[nJobs] = ( int ) blocks; mj.size[nJobs] = jb.width;
        const int groups = ( jb.n + tpb - 1 ) / tpb;
        ra.groupStride = ( groups + tuRepeat() - 1 ) / tuRepeat();
        blocks += ra.groupStride;
        mj.j[nJobs] = ra;
      }
      nJobs++;
    }
    if( mx )
    {
      // matrix-core form: one wave per 32x32 tile of (32/N)^2 TUs
      TuMxJobs xj; xj.nJobs = nJobs; xj.pair64 = pair64 ? 1 : 0;
      long waves = 0;
      for( int i = 0; i < nJobs; i++ )
      {
        const vvhip_tu_job& jb = jobs[order[first + i]];
        const TuRowArgs& ra = rowArgs[i];
        TuMxArgs& xa = xj.j[i];
        const int z = ra.gf.log2w - 2, tpt = jb.width == 64 ? 1 : ( 32 / jb.width ) * ( 32 / jb.width );
        xa.resiOff = jb.d_resi_off; xa.n = jb.n;
        xa.shF1 = ra.gf.shift1; xa.shF2 = ra.gf.shift2; xa.shI1 = ra.gi.shift1; xa.shI2 = ra.gi.shift2; xa.skipW = ra.gf.skipW; xa.skipH = ra.gf.skipH;
        xa.q = ra.q;
        xa.opH = ctx->d_tuMx + jb.tr_hor * 4 + z; xa.opV = ctx->d_tuMx + jb.tr_ver * 4 + z; xa.pos = ctx->d_tuMxPos + z * 64 * 16;
        if( jb.width == 64 ) { xa.opH = reinterpret_cast<const VvhipTuMxOps*>( ctx->d_tuMx64 ); xa.opV = xa.opH; xa.pos = nullptr; }
        xa.qps = jb.d_qp; xa.thrVal = jb.thr_val; xa.level = jb.d_level; xa.rec = jb.d_rec_resi; xa.stats = jb.d_stats;
        xa.tiles = ( jb.n + tpt - 1 ) / tpt; xa.phaseLimit = tuPhaseLimit() | ( ctx->tuSparse ? 0x100 : 0 );
        const int repeat = budget ? repeatOf( first + i ) : tuRepeat();
        xa.waveStride = wavesPer( first + i ) * ( ( xa.tiles + repeat - 1 ) / repeat );
        xa.resiStride = strides ? strides[order[first + i]] : 0;
        if( wavesPer( first + i ) == 2 ) waves = ( waves + 1 ) & ~1l;          // a pair = waves 2g, 2g + 1 of the launch (one workgroup)
        xj.waveStart[i] = ( int ) waves; xj.size[i] = jb.width;
        waves += xa.waveStride;
      }
      for( int i = nJobs; i < 8; i++ ) { xj.waveStart[i] = 0x7fffffff; xj.size[i] = 0; }
      bool any4 = false;
      for( int i = 0; i < xj.nJobs; i++ ) any4 |= xj.size[i] == 4 || xj.size[i] == 64;
      bool any64 = false;
      for( int i = 0; i < xj.nJobs; i++ ) any64 |= xj.size[i] == 64;
      const dim3 grid( ( unsigned ) ( ( waves + 3 ) / 4 ) );
      // $VVHIP_TU_LDS_PAD: extra LDS bytes reserved per workgroup (unused) = fewer TU workgroups per CU.  A TU wave holds 168 / 232 registers: three / two of them fill a SIMD's
      // register file and nothing of the other streams' kernels can share that SIMD while they wait for memory (measurements of the five-stream step: DESIGN 6)
      static const int ldsPad = getenv( "VVHIP_TU_LDS_PAD" ) ? atoi( getenv( "VVHIP_TU_LDS_PAD" ) ) : 0;
      static bool padSet = false;
      if( ldsPad > 0 && !padSet )
      {
        padSet = true;
        ( void ) hipFuncSetAttribute( ( const void* ) tuMxMultiKernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsPad );
        ( void ) hipFuncSetAttribute( ( const void* ) tuMxMultiKernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsPad );
        ( void ) hipFuncSetAttribute( ( const void* ) tuMxMultiKernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsPad );
        ( void ) hipFuncSetAttribute( ( const void* ) tuMxMultiKernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsPad );
      }
      if( any64 && pair64 ) hipLaunchKernelGGL( tuMxMultiKernel<3>, grid, dim3( 256 ), ldsPad, ctx->stream, d_resi, resi_stride, xj );
      else if( any64 ) hipLaunchKernelGGL( tuMxMultiKernel<2>, grid, dim3( 256 ), ldsPad, ctx->stream, d_resi, resi_stride, xj );
      else if( any4 ) hipLaunchKernelGGL( tuMxMultiKernel<1>, grid, dim3( 256 ), ldsPad, ctx->stream, d_resi, resi_stride, xj );
      else            hipLaunchKernelGGL( tuMxMultiKernel<0>, grid, dim3( 256 ), ldsPad, ctx->stream, d_resi, resi_stride, xj );
    }
    else
    {
      mj.nJobs = nJobs;
      for( int i = nJobs; i < 4; i++ ) { mj.blockStart[i] = 0x7fffffff; mj.size[i] = 0; }
      hipLaunchKernelGGL( tuRdoRowMultiKernel, dim3( ( unsigned ) blocks ), dim3( 256 ), 0, ctx->stream, d_resi, resi_stride, mj );
    }
    VVHIP_LAUNCH_CHECK( ctx );
  }
  return VVHIP_OK;
}

static bool trSizeOk( int n ) { return n == 4 || n == 8 || n == 16 || n == 32 || n == 64; }

int vvhip_fast_fwd_core( vvhip_ctx* ctx, int tr_size, const int16_t* d_tc, const int32_t* d_src, int32_t* d_dst, unsigned line, unsigned reduced_line, unsigned cutoff, int shift )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !trSizeOk( tr_size ) || !d_tc || !d_src || !d_dst || reduced_line > line || cutoff > ( unsigned ) tr_size || line > 64 || shift < 1 || shift > 31 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_fast_fwd_core: trSize %d line %u/%u cutoff %u shift %d", tr_size, reduced_line, line, cutoff, shift );
  const unsigned total = reduced_line * cutoff;
  if( !total ) return VVHIP_OK;
  hipLaunchKernelGGL( fastFwdCoreKernel, dim3( ( total + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, tr_size, d_tc, d_src, d_dst, line, reduced_line, cutoff, shift );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_fast_inv_core( vvhip_ctx* ctx, int tr_size, const int16_t* d_it, const int32_t* d_src, int32_t* d_dst, unsigned lines, unsigned reduced_lines, unsigned rows )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !trSizeOk( tr_size ) || !d_it || !d_src || !d_dst || reduced_lines > lines || rows > ( unsigned ) tr_size || lines > 64 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_fast_inv_core: trSize %d lines %u/%u rows %u", tr_size, reduced_lines, lines, rows );
  const unsigned total = reduced_lines * tr_size;
  if( !total ) return VVHIP_OK;
  hipLaunchKernelGGL( fastInvCoreKernel, dim3( ( total + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, tr_size, d_it, d_src, d_dst, lines, reduced_lines, rows );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_round_clip( vvhip_ctx* ctx, int32_t* d_dst, unsigned width, unsigned height, unsigned stride, int32_t out_min, int32_t out_max, int32_t round, int32_t shift )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_dst || width > stride || shift < 0 || shift > 31 || ( uint64_t ) width * height > ( 1u << 24 ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_round_clip: bad arguments" );
  if( !( width * height ) ) return VVHIP_OK;
  hipLaunchKernelGGL( roundClipKernel, dim3( ( width * height + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_dst, width, height, stride, out_min, out_max, round, shift );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_cpy_resi( vvhip_ctx* ctx, const int32_t* d_src, int16_t* d_dst, ptrdiff_t stride, unsigned width, unsigned height )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_src || !d_dst || ( uint64_t ) width * height > ( 1u << 24 ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_cpy_resi: bad arguments" );
  if( !( width * height ) ) return VVHIP_OK;
  hipLaunchKernelGGL( cpyResiKernel, dim3( ( width * height + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_src, d_dst, stride, width, height );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_cpy_coeff( vvhip_ctx* ctx, const int16_t* d_src, ptrdiff_t stride, int32_t* d_dst, unsigned width, unsigned height )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_src || !d_dst || ( uint64_t ) width * height > ( 1u << 24 ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_cpy_coeff: bad arguments" );
  if( !( width * height ) ) return VVHIP_OK;
  hipLaunchKernelGGL( cpyCoeffKernel, dim3( ( width * height + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_src, stride, d_dst, width, height );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_need_rdoq_core( vvhip_ctx* ctx, const int32_t* d_coef, size_t num_coeff, int quant_coeff, int64_t offset, int shift, uint8_t* d_need )
{
  if( !ctx ) return VVHIP_E_ARG;
  hipLaunchKernelGGL( needRdoqCoreKernel, dim3( 1 ), dim3( 256 ), 0, ctx->stream, d_coef, num_coeff, quant_coeff, ( long long ) offset, shift, d_need );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_tu_rdo_batch( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* d_resi_off, int n, int width, int height,
                        int tr_hor, int tr_ver, int bit_depth, const vvhip_tu_qp* d_qp, int thr_val,
                        int16_t* d_level, int16_t* d_rec_resi, vvhip_tu_stats* d_stats )
{
  if( !ctx ) return VVHIP_E_ARG;
  TrGeom gf, gi;
  QGeom q;
  if( n < 0 || !makeGeom( width, height, tr_hor, tr_ver, bit_depth, false, gf ) || !makeGeom( width, height, tr_hor, tr_ver, bit_depth, true, gi ) || !makeQGeom( width, height, bit_depth, q ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_tu_rdo_batch: unsupported %dx%d types (%d,%d) bitDepth %d", width, height, tr_hor, tr_ver, bit_depth );
  if( n == 0 ) return VVHIP_OK;
  const int area = width * height;
  if( width == height && ( width == 4 || width == 8 || width == 16 || width == 32 || ( width == 64 && tr_hor == VVHIP_DCT2 && tr_ver == VVHIP_DCT2 ) ) && !getenv( "VVHIP_TU_GENERIC" ) && tuKernelForm() == 0 )
  {
    vvhip_tu_job jb; jb.width = width; jb.height = height; jb.tr_hor = tr_hor; jb.tr_ver = tr_ver; jb.n = n; jb.thr_val = thr_val;
    jb.d_resi_off = d_resi_off; jb.d_qp = d_qp; jb.d_level = d_level; jb.d_rec_resi = d_rec_resi; jb.d_stats = d_stats;
    return vvhip_tu_rdo_multi( ctx, d_resi, resi_stride, bit_depth, &jb, 1 );
  }
  if( width == height && ( width == 8 || width == 16 || width == 32 ) && !getenv( "VVHIP_TU_GENERIC" ) )
  {
    const int16_t* mh = ctx->d_trMat + trMatOffset( tr_hor, gf.log2w );
    const int16_t* mv = ctx->d_trMat + trMatOffset( tr_ver, gf.log2h );
    const uint16_t* sc = ctx->d_scan + scanOffset( q.log2w, q.log2h );
    TuRowArgs ra; ra.resiOff = d_resi_off; ra.n = n; ra.gf = gf; ra.gi = gi; ra.q = q; ra.matH = mh; ra.matV = mv; ra.scan = sc;
    ra.qps = d_qp; ra.thrVal = thr_val; ra.level = d_level; ra.rec = d_rec_resi; ra.stats = d_stats; ra.phaseLimit = tuPhaseLimit();
#define ROWK( NN, SP ) { const int groups = ( n + ( 256 / ( NN * SP ) ) - 1 ) / ( 256 / ( NN * SP ) ); ra.groupStride = ( groups + tuRepeat() - 1 ) / tuRepeat(); \
                         hipLaunchKernelGGL( ( tuRdoRowKernel<NN, SP> ), dim3( ra.groupStride ), dim3( 256 ), 0, ctx->stream, d_resi, resi_stride, ra ); }
    static const int split16 = getenv( "VVHIP_TU_SPLIT16" ) ? atoi( getenv( "VVHIP_TU_SPLIT16" ) ) : 2;
    if( width == 8 ) ROWK( 8, 1 ) else if( width == 16 ) { if( split16 == 2 ) ROWK( 16, 2 ) else ROWK( 16, 1 ) } else ROWK( 32, 2 )
#undef ROWK
    VVHIP_LAUNCH_CHECK( ctx );
    return VVHIP_OK;
  }
  const TuLay y = makeLayout( gf, tr_hor, tr_ver );
  const int tpbF = area >= 1024 ? 1 : 1024 / area;          // ~1024 samples per workgroup: 4 independent iterations per thread and phase
  return launchTuRdo( ctx, d_resi, resi_stride, d_resi_off, n, gf, gi, y, q, tpbF, tr_hor, tr_ver, d_qp, thr_val, d_level, d_rec_resi, d_stats );
}

} // extern "C"

namespace {
int launchTuRdo( vvhip_ctx* ctx, const int16_t* d_resi, int resi_stride, const int32_t* d_resi_off, int n, const TrGeom& gf, const TrGeom& gi, const TuLay& y,
                 const QGeom& q, int tpb, int tr_hor, int tr_ver, const vvhip_tu_qp* d_qp, int thr_val, int16_t* d_level, int16_t* d_rec_resi, vvhip_tu_stats* d_stats )
{
  const int phaseLimit = tuPhaseLimit();
  const size_t smem = trSmemBytes( y, tpb );
  if( smem > 64 * 1024 ) VVHIP_CHECK_HIP( ctx, hipFuncSetAttribute( ( const void* ) tuRdoKernel, hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) smem ) );
  hipLaunchKernelGGL( tuRdoKernel, dim3( ( n + tpb - 1 ) / tpb ), dim3( 256 ), smem, ctx->stream,
                      d_resi, resi_stride, d_resi_off, n, gf, gi, y, q, tpb,
                      ctx->d_trMat + trMatOffset( tr_hor, gf.log2w ), ctx->d_trMat + trMatOffset( tr_ver, gf.log2h ),
                      ctx->d_scan + scanOffset( q.log2w, q.log2h ), d_qp, thr_val, d_level, d_rec_resi, d_stats, phaseLimit );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}
} // namespace
