// trquant.hip — forward / inverse DCT-2, DST-7, DCT-8 and scalar (de)quantisation for gfx950.
//
// Reference semantics
//   forward 2-D   TrQuant::xT            CommonLib/TrQuant.cpp:481-564  ->  _fastForwardMM / fastFwdCore   TrQuant_EMT.cpp:366-420,1973-2000
//   inverse 2-D   TrQuant::xIT           CommonLib/TrQuant.cpp:567-655  ->  _fastInverseMM / fastInvCore_  TrQuant_EMT.cpp:152-194,1953-1970 + clipCore :1941
//   quantiser     Quant::quant           CommonLib/Quant.cpp:735-833    ->  QuantCore   :132-230
//   dequantiser   Quant::dequant         CommonLib/Quant.cpp:520-610    ->  DeQuantCore :232-262
//   RDOQ pre-test Quant::xNeedRDOQ       CommonLib/Quant.cpp:835-891    ->  needRdoqCore :264-278
// The N=2/4/8 butterflies of the reference compute the same integer sums as the matrix form (32-bit
// wrap-around arithmetic), so one matrix-form kernel covers every size and type.
//
// Layout: one workgroup owns TPB transform units (TPB = 256 / (w*h) for small TUs, else 1).  The
// residual block, the intermediate and both kernel matrices live in LDS; matrices are stored so that
// consecutive lanes (consecutive output frequencies / samples) read consecutive LDS words while the
// input operand is a wave-wide broadcast.  Coefficients never touch HBM between the two 1-D passes.
#include "common.h"

namespace {

__device__ __forceinline__ int clip3i( int lo, int hi, int v ) { return v < lo ? lo : ( v > hi ? hi : v ); }

struct TrGeom
{
  int w, h, log2w, log2h;
  int skipW, skipH;          // TrQuant.cpp:496-497 / :587-588 (LFNST off)
  int shift1, shift2;
};

// --------------------------------------------------------------------------------------------
// 2-D transforms on LDS-resident TUs.  Both kernel matrices stay in their natural [frequency][sample]
// layout; operands are arranged so that, in every pass, consecutive lanes read consecutive LDS words
// of one operand while the other operand is a wave-wide broadcast:
//   forward  pass 1: lanes over block rows i     : blkT[k][i] (transposed residual) x Th[j][k] (broadcast)  -> tmp[i][j]
//            pass 2: lanes over hor. freq.  i2   : tmp[k][i2]                       x Tv[j2][k] (broadcast) -> coef[j2][i2] (raster)
//   inverse  pass 1: lanes over columns i        : coef[k][i]                       x Tv[k][j] (broadcast)  -> t1[j][i]
//            pass 2: lanes over columns j2       : Th[k][j2]                        x t1[i2][k] (broadcast) -> rec[i2][j2] (raster)
// Strided LDS *writes* use a +1 padded pitch (conflict-free).
// --------------------------------------------------------------------------------------------
struct TuLds
{
  int32_t* a;      // [TPB][pitchA * ...]  transposed residual (fwd in) / dequantised coefficients (inv in), later output staging
  int32_t* b;      // intermediate
  const int16_t* th;
  const int16_t* tv;
};

__device__ __forceinline__ int ldsTuStride( const TrGeom& g ) { return max( g.w * ( g.h + 1 ), g.h * ( g.w + 1 ) ); }   // words per TU slot

// in : a[t][x*(h+1) + y] = resi[y][x]        out: a[t][j2*w + i2] = coef (raster, zero-out applied)
__device__ __forceinline__ void fwd2dLds( const TrGeom& g, const TuLds& L, int nTu, int tid, int nthr )
{
  const int w = g.w, h = g.h, area = w * h, slot = ldsTuStride( g );
  const int cutW = w - g.skipW, cutH = h - g.skipH;
  const uint32_t rnd1 = g.shift1 > 0 ? 1u << ( g.shift1 - 1 ) : 0u, rnd2 = 1u << ( g.shift2 - 1 );
  // pass 1: tmp[i*(w+1) + j] = ( sum_k blk[i][k] * Th[j][k] + rnd ) >> shift1, j < w - skipW        (TrQuant.cpp:548)
  for( int o = tid; o < nTu * area; o += nthr )
  {
    const int t = o / area, p = o - t * area, j = p / h, i = p - j * h;             // i fastest
    int32_t v = 0;
    if( j < cutW )
    {
      const int32_t* src = L.a + t * slot + i;
      const int16_t* m = L.th + j * w;
      uint32_t acc = 0;
      for( int k = 0; k < w; k++ ) acc += ( uint32_t ) src[k * ( h + 1 )] * ( uint32_t ) ( int32_t ) m[k];
      v = ( int32_t ) ( acc + rnd1 ) >> g.shift1;
    }
    L.b[t * slot + i * ( w + 1 ) + j] = v;
  }
  __syncthreads();
  // pass 2: coef[j2*w + i2] = ( sum_k tmp[k][i2] * Tv[j2][k] + rnd ) >> shift2, i2 < w - skipW, j2 < h - skipH   (TrQuant.cpp:549)
  for( int o = tid; o < nTu * area; o += nthr )
  {
    const int t = o / area, p = o - t * area, j2 = p / w, i2 = p - j2 * w;         // i2 fastest
    int32_t v = 0;
    if( i2 < cutW && j2 < cutH )
    {
      const int32_t* src = L.b + t * slot + i2;
      const int16_t* m = L.tv + j2 * h;
      uint32_t acc = 0;
      for( int k = 0; k < h; k++ ) acc += ( uint32_t ) src[k * ( w + 1 )] * ( uint32_t ) ( int32_t ) m[k];
      v = ( int32_t ) ( acc + rnd2 ) >> g.shift2;
    }
    L.a[t * slot + p] = v;
  }
  __syncthreads();
}

// in : a[t][k*w + i] = coefficients (raster)  out: a[t][i2*w + j2] = residual (raster), clipped to int16
__device__ __forceinline__ void inv2dLds( const TrGeom& g, const TuLds& L, int nTu, int tid, int nthr )
{
  const int w = g.w, h = g.h, area = w * h, slot = ldsTuStride( g );
  const int cutW = w - g.skipW, cutH = h - g.skipH;
  const int32_t cmin = -32768, cmax = 32767;
  const uint32_t rnd1 = 1u << ( g.shift1 - 1 ), rnd2 = 1u << ( g.shift2 - 1 );
  // pass 1 (columns): t1[j*w + i] = clip( ( sum_{k<cutH} coef[k][i] * Tv[k][j] + rnd ) >> shift1 ), i < w - skipW, else 0   (TrQuant.cpp:612)
  for( int o = tid; o < nTu * area; o += nthr )
  {
    const int t = o / area, p = o - t * area, j = p / w, i = p - j * w;             // i fastest
    int32_t v = 0;
    if( i < cutW )
    {
      const int32_t* src = L.a + t * slot + i;
      const int16_t* m = L.tv + j;
      uint32_t acc = 0;
      for( int k = 0; k < cutH; k++ ) acc += ( uint32_t ) src[k * w] * ( uint32_t ) ( int32_t ) m[k * h];
      v = clip3i( cmin, cmax, ( int32_t ) ( acc + rnd1 ) >> g.shift1 );
    }
    L.b[t * slot + p] = v;
  }
  __syncthreads();
  // pass 2 (rows): rec[i2*w + j2] = clip( ( sum_{k<cutW} t1[i2][k] * Th[k][j2] + rnd ) >> shift2 )                          (TrQuant.cpp:613)
  for( int o = tid; o < nTu * area; o += nthr )
  {
    const int t = o / area, p = o - t * area, i2 = p / w, j2 = p - i2 * w;         // j2 fastest
    const int32_t* src = L.b + t * slot + i2 * w;
    const int16_t* m = L.th + j2;
    uint32_t acc = 0;
    for( int k = 0; k < cutW; k++ ) acc += ( uint32_t ) src[k] * ( uint32_t ) ( int32_t ) m[k * w];
    L.a[t * slot + p] = clip3i( cmin, cmax, ( int32_t ) ( acc + rnd2 ) >> g.shift2 );
  }
  __syncthreads();
}

__device__ __forceinline__ TuLds carveLds( unsigned char* raw, const TrGeom& g, int tpb, const int16_t* matH, const int16_t* matV, int tid, int nthr )
{
  TuLds L;
  const int slot = ldsTuStride( g );
  L.a = reinterpret_cast<int32_t*>( raw );
  L.b = L.a + tpb * slot;
  int16_t* th = reinterpret_cast<int16_t*>( L.b + tpb * slot );
  int16_t* tv = th + g.w * g.w;
  for( int i = tid; i < g.w * g.w; i += nthr ) th[i] = matH[i];
  for( int i = tid; i < g.h * g.h; i += nthr ) tv[i] = matV[i];
  L.th = th; L.tv = tv;
  return L;
}

// transposed residual load: a[t][x*(h+1) + y] = resi[y][x]   (cpyCoeff, TrQuant_EMT.cpp:1917-1926)
__device__ __forceinline__ void loadResiT( const TrGeom& g, const TuLds& L, const int16_t* resi, int resiStride, const int32_t* resiOff,
                                           int tu0, int nTu, int tid, int nthr )
{
  const int w = g.w, h = g.h, area = w * h, slot = ldsTuStride( g );
  for( int i = tid; i < nTu * area; i += nthr )
  {
    const int t = i / area, p = i - t * area, y = p / w, x = p - y * w;
    L.a[t * slot + x * ( h + 1 ) + y] = resi[resiOff[tu0 + t] + ( ptrdiff_t ) y * resiStride + x];
  }
}

__global__ void __launch_bounds__( 256 )
fwdTransformKernel( const int16_t* __restrict__ resi, int resiStride, const int32_t* __restrict__ resiOff, int n,
                    TrGeom g, int tpb, const int16_t* __restrict__ matH, const int16_t* __restrict__ matV,
                    int32_t* __restrict__ coef )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  const int tid = threadIdx.x, nthr = blockDim.x, area = g.w * g.h, slot = ldsTuStride( g );
  const int tu0 = blockIdx.x * tpb, nTu = min( tpb, n - tu0 );
  const TuLds L = carveLds( smemRaw, g, tpb, matH, matV, tid, nthr );
  loadResiT( g, L, resi, resiStride, resiOff, tu0, nTu, tid, nthr );
  __syncthreads();
  fwd2dLds( g, L, nTu, tid, nthr );
  for( int i = tid; i < nTu * area; i += nthr ) { const int t = i / area, p = i - t * area; coef[( size_t ) tu0 * area + i] = L.a[t * slot + p]; }
}

__global__ void __launch_bounds__( 256 )
invTransformKernel( const int32_t* __restrict__ coef, int n, TrGeom g, int tpb,
                    const int16_t* __restrict__ matH, const int16_t* __restrict__ matV,
                    int16_t* __restrict__ resi, int resiStride, const int32_t* __restrict__ resiOff )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  const int tid = threadIdx.x, nthr = blockDim.x, w = g.w, area = g.w * g.h, slot = ldsTuStride( g );
  const int tu0 = blockIdx.x * tpb, nTu = min( tpb, n - tu0 );
  const TuLds L = carveLds( smemRaw, g, tpb, matH, matV, tid, nthr );
  for( int i = tid; i < nTu * area; i += nthr ) { const int t = i / area, p = i - t * area; L.a[t * slot + p] = coef[( size_t ) tu0 * area + i]; }
  __syncthreads();
  inv2dLds( g, L, nTu, tid, nthr );
  for( int i = tid; i < nTu * area; i += nthr )
  {
    const int t = i / area, p = i - t * area, y = p / w, x = p - y * w;
    resi[resiOff[tu0 + t] + ( ptrdiff_t ) y * resiStride + x] = ( int16_t ) L.a[t * slot + p];     // cpyResi, TrQuant_EMT.cpp:1929-1938
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
quantKernel( const int32_t* __restrict__ coef, int n, QGeom q, int log2Lpc, const vvhip_tu_qp* __restrict__ qps, int thrVal,
             const uint16_t* __restrict__ scan, int16_t* __restrict__ level, int32_t* __restrict__ deltaU,
             int32_t* __restrict__ absSumOut, int32_t* __restrict__ lastPosOut )
{
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lpc = 1 << log2Lpc, tu = gid >> log2Lpc, lt = gid & ( lpc - 1 );
  const bool valid = tu < n;
  const int area = q.w * q.h;
  const int32_t* src = coef + ( size_t ) ( valid ? tu : 0 ) * area;
  int16_t* dst = level + ( size_t ) ( valid ? tu : 0 ) * area;
  int scale = 1, qBits = 16;
  int64_t add = 0;
  if( valid )
  {
    const vvhip_tu_qp p = qps[tu];
    quantParams( q, p.qp, scale, qBits );
    add = ( int64_t ) ( ( p.flags & 1 ) ? 171 : 85 ) << ( qBits - 9 );               // Quant.cpp:775
  }
  const int num = valid ? q.numScan : 0;

  // (1) last non-zero scan position (Quant.cpp:162-167); 0 if none
  int last = 0;
  for( int p = lt; p < num; p += lpc ) if( src[scan[p]] != 0 ) last = p;             // p increases -> keeps the largest
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
// Coefficients, levels and the intermediate of both transforms stay in LDS; HBM sees 2 B/sample in and
// 2+2 B/sample out plus 24 B of statistics per TU.
// --------------------------------------------------------------------------------------------
struct TuRed { int last; uint32_t bigLo, bigHi; uint32_t absSum; int need; int pad; unsigned long long sse; };

__global__ void __launch_bounds__( 256 )
tuRdoKernel( const int16_t* __restrict__ resi, int resiStride, const int32_t* __restrict__ resiOff, int n,
             TrGeom gf, TrGeom gi, QGeom q, int tpb, const int16_t* __restrict__ matH, const int16_t* __restrict__ matV,
             const uint16_t* __restrict__ scan, const vvhip_tu_qp* __restrict__ qps, int thrVal,
             int16_t* __restrict__ level, int16_t* __restrict__ rec, vvhip_tu_stats* __restrict__ stats )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  const int tid = threadIdx.x, nthr = blockDim.x, w = gf.w, area = gf.w * gf.h, slot = ldsTuStride( gf );
  const int tu0 = blockIdx.x * tpb, nTu = min( tpb, n - tu0 );
  const TuLds L = carveLds( smemRaw, gf, tpb, matH, matV, tid, nthr );
  TuRed* red = reinterpret_cast<TuRed*>( smemRaw + ( ( ( size_t ) 2 * tpb * slot * sizeof( int32_t ) + ( size_t ) ( gf.w * gf.w + gf.h * gf.h ) * sizeof( int16_t ) + 15 ) & ~( size_t ) 15 ) );
  for( int t = tid; t < nTu; t += nthr ) { TuRed r; r.last = 0; r.bigLo = r.bigHi = 0; r.absSum = 0; r.need = 0; r.pad = 0; r.sse = 0; red[t] = r; }
  loadResiT( gf, L, resi, resiStride, resiOff, tu0, nTu, tid, nthr );
  __syncthreads();
  fwd2dLds( gf, L, nTu, tid, nthr );                       // L.a[t] = coefficients, raster

  // ---- need-RDOQ pre-test + last significant scan position
  const int efArea = q.w * min( q.h, 32 );
  for( int o = tid; o < nTu * area; o += nthr )
  {
    const int t = o / area, p = o - t * area;
    const vvhip_tu_qp qq = qps[tu0 + t];
    int scale, qBits;
    quantParams( q, qq.qp, scale, qBits );
    if( p < efArea )
    {
      const int64_t addN = ( int64_t ) ( ( qq.flags & 2 ) ? 171 : 256 ) << ( qBits - 9 );
      const int64_t tt = ( int64_t ) abs( L.a[t * slot + p] ) * scale;
      if( ( int32_t ) ( ( tt + addN ) >> qBits ) != 0 ) red[t].need = 1;          // benign race: all writers store 1
    }
    if( p < q.numScan && L.a[t * slot + scan[p]] != 0 ) atomicMax( &red[t].last, p );
  }
  __syncthreads();
  if( q.cgIs4x4 )
  {
    for( int o = tid; o < nTu * area; o += nthr )
    {
      const int t = o / area, p = o - t * area;
      const int last = red[t].last;
      if( last >= 16 && p <= last && p < q.numScan )
      {
        const vvhip_tu_qp qq = qps[tu0 + t];
        int scale, qBits;
        quantParams( q, qq.qp, scale, qBits );
        const int32_t thres = qBits ? ( int32_t ) ( ( int64_t ) thrVal << ( qBits - 1 ) ) : ( int32_t ) ( ( int64_t ) ( thrVal >> 1 ) << qBits );
        const int32_t useThres = thres / ( scale << 2 );
        if( abs( L.a[t * slot + scan[p]] ) > useThres )
        {
          const int cg = p >> 4;
          if( cg > 0 ) { if( cg < 32 ) atomicOr( &red[t].bigLo, 1u << cg ); else atomicOr( &red[t].bigHi, 1u << ( cg - 32 ) ); }
        }
      }
    }
    __syncthreads();
    for( int t = tid; t < nTu; t += nthr )
    {
      const int last = red[t].last;
      if( last >= 16 )
      {
        const uint64_t big = ( ( uint64_t ) red[t].bigHi << 32 ) | red[t].bigLo;
        if( big == 0 ) red[t].last = 15;
        else { const int g2 = 63 - __clzll( ( long long ) big ); if( g2 != ( last >> 4 ) ) red[t].last = g2 * 16 + 15; }
      }
    }
    __syncthreads();
  }
  // ---- QuantCore + DeQuantCore: b[t][raster] = dequantised coefficient (0 beyond `last`), levels to HBM
  for( int o = tid; o < nTu * area; o += nthr )
  {
    const int t = o / area, p = o - t * area;
    int bp = p; bool inScan = false;
    if( p < q.numScan ) { bp = scan[p]; inScan = true; }
    else if( q.w > 32 || q.h > 32 )
    {
      // enumerate the zero-out region (x >= 32 or y >= 32) with the remaining indices
      const int r = p - q.numScan;                     // 0 .. area - numScan - 1
      const int wz = q.w - min( q.w, 32 );             // columns right of the region
      const int rowsTop = min( q.h, 32 );
      if( r < rowsTop * wz ) { const int y = r / wz, x = 32 + ( r - y * wz ); bp = y * q.w + x; }
      else { const int r2 = r - rowsTop * wz; bp = rowsTop * q.w + r2; }
    }
    int16_t lv = 0; int32_t dq = 0;
    if( inScan && p <= red[t].last )
    {
      const vvhip_tu_qp qq = qps[tu0 + t];
      int scale, qBits;
      quantParams( q, qq.qp, scale, qBits );
      const int64_t add = ( int64_t ) ( ( qq.flags & 1 ) ? 171 : 85 ) << ( qBits - 9 );
      const int32_t c = L.a[t * slot + bp];
      const int64_t tt = ( int64_t ) abs( c ) * scale;
      const int32_t m = ( int32_t ) ( ( tt + add ) >> qBits );
      if( m ) atomicAdd( &red[t].absSum, ( uint32_t ) m );
      lv = ( int16_t ) clip3i( -32768, 32767, c < 0 ? -m : m );
      // DeQuantCore (Quant.cpp:232-262) with Quant::dequant's parameters (:554-561,:601-607)
      const int l2 = q.log2w + q.log2h, sqrt2 = l2 & 1;
      const int trShift = 15 - q.bitDepth - ( l2 >> 1 ) - sqrt2;
      const int iscale = cInvQuantScales[sqrt2][qq.qp % 6];
      const int rightShift = 6 - ( trShift + qq.qp / 6 );
      int tgt = 32 + rightShift - 7; if( tgt > 16 ) tgt = 16;
      const int inMax = ( 1 << ( tgt - 1 ) ) - 1;
      const int cl = clip3i( -( inMax + 1 ), inMax, ( int ) lv );
      int32_t v;
      if( rightShift > 0 ) v = ( int32_t ) ( ( uint32_t ) ( cl * iscale ) + ( 1u << ( rightShift - 1 ) ) ) >> rightShift;
      else                 v = ( int32_t ) ( ( uint32_t ) ( cl * iscale ) << ( -rightShift ) );
      dq = clip3i( -32768, 32767, v );
    }
    L.b[t * slot + bp] = dq;
    if( level ) level[( size_t ) ( tu0 + t ) * area + bp] = lv;
  }
  __syncthreads();
  // move dequantised coefficients to the inverse transform's input buffer
  for( int o = tid; o < nTu * area; o += nthr ) { const int t = o / area, p = o - t * area; L.a[t * slot + p] = L.b[t * slot + p]; }
  __syncthreads();
  inv2dLds( gi, L, nTu, tid, nthr );                       // L.a[t] = reconstructed residual, raster
  for( int o = tid; o < nTu * area; o += nthr )
  {
    const int t = o / area, p = o - t * area, y = p / w, x = p - y * w;
    const int r = L.a[t * slot + p];
    if( rec ) rec[( size_t ) ( tu0 + t ) * area + p] = ( int16_t ) r;
    const int d = ( int ) resi[resiOff[tu0 + t] + ( ptrdiff_t ) y * resiStride + x] - r;
    if( d ) atomicAdd( &red[t].sse, ( unsigned long long ) ( ( long long ) d * d ) );
  }
  __syncthreads();
  if( stats )
    for( int t = tid; t < nTu; t += nthr )
    {
      vvhip_tu_stats st; st.abs_sum = ( int32_t ) red[t].absSum; st.last_scan_pos = red[t].last; st.need_rdoq = red[t].need; st.pad = 0; st.sse = red[t].sse;
      stats[tu0 + t] = st;
    }
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

size_t trSmemBytes( const TrGeom& g, int tpb )
{
  const int slot = g.w * ( g.h + 1 ) > g.h * ( g.w + 1 ) ? g.w * ( g.h + 1 ) : g.h * ( g.w + 1 );
  return ( size_t ) 2 * tpb * slot * sizeof( int32_t ) + ( size_t ) ( g.w * g.w + g.h * g.h ) * sizeof( int16_t ) + 64;
}

int teamLog2( int positions ) { int l = 0; while( ( 1 << ( l + 1 ) ) <= positions && l < 6 ) l++; return l; }

} // namespace

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
  const size_t smem = trSmemBytes( g, tpb );
  hipLaunchKernelGGL( fwdTransformKernel, dim3( ( n + tpb - 1 ) / tpb ), dim3( 256 ), smem, ctx->stream,
                      d_resi, resi_stride, d_resi_off, n, g, tpb,
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
  const size_t smem = trSmemBytes( g, tpb );
  hipLaunchKernelGGL( invTransformKernel, dim3( ( n + tpb - 1 ) / tpb ), dim3( 256 ), smem, ctx->stream,
                      d_coef, n, g, tpb, ctx->d_trMat + trMatOffset( tr_hor, g.log2w ), ctx->d_trMat + trMatOffset( tr_ver, g.log2h ),
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
                      d_coef, n, q, log2Lpc, d_qp, thr_val, ctx->d_scan + scanOffset( q.log2w, q.log2h ), d_level, d_delta_u, d_abs_sum, d_last_scan_pos );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
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
  const int tpb = area >= 256 ? 1 : 256 / area;
  const size_t smem = ( ( trSmemBytes( gf, tpb ) + 15 ) & ~( size_t ) 15 ) + ( size_t ) tpb * sizeof( TuRed ) + 16;
  hipLaunchKernelGGL( tuRdoKernel, dim3( ( n + tpb - 1 ) / tpb ), dim3( 256 ), smem, ctx->stream,
                      d_resi, resi_stride, d_resi_off, n, gf, gi, q, tpb,
                      ctx->d_trMat + trMatOffset( tr_hor, gf.log2w ), ctx->d_trMat + trMatOffset( tr_ver, gf.log2h ),
                      ctx->d_scan + scanOffset( q.log2w, q.log2h ), d_qp, thr_val, d_level, d_rec_resi, d_stats );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

} // extern "C"
