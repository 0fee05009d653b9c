// interp.hip — SURVEY §8f rank 1: sub-pel interpolation for fractional motion estimation / motion compensation.
//
// Reference behaviour (all integer, bit-exact):
//   InterpolationFilter::filter<N,isVertical,isFirst,isLast>      CommonLib/InterpolationFilter.cpp:356-441
//   InterpolationFilter::filterCopy<isFirst,isLast>               :255-333
//   InterpolationFilter::filterHor / filterVer (luma dispatch)    :557-661
//   InterPredInterpolation::xPredInterBlk (which passes run)      CommonLib/InterPrediction.cpp:832-865
//   InterSearch::xPatternRefinement (sub-pel candidates + cost)   EncoderLib/InterSearch.cpp:760-880
// One pass: val = ( sum_k c[k] * src[(k - (N/2-1)) * step] + offset ) >> shift, truncated to Pel, clipped to [0, 2^bd-1] when it is the
// last pass.  A first-and-not-last pass keeps 14-bit precision minus IF_INTERNAL_OFFS; the second pass undoes both.
#include <stdlib.h>
#include "common.h"

namespace {

// VVC tap sets, phases 0..P/2; row P-p is row p reversed (InterpolationFilter.cpp:64-142)
__constant__ int8_t cLuma8[9][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { -1, 2, -5, 62, 8, -3, 1, 0 }, { -1, 3, -8, 60, 13, -4, 1, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
  { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 }, { -1, 4, -11, 40, 40, -11, 4, -1 } };
__constant__ int8_t cLuma6[9][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { 0, 1, -5, 62, 8, -3, 1, 0 }, { 0, 2, -8, 60, 13, -4, 1, 0 }, { 0, 3, -10, 58, 17, -5, 1, 0 },
  { 0, 3, -11, 52, 26, -8, 2, 0 }, { 0, 2, -9, 47, 31, -10, 3, 0 }, { 0, 3, -11, 45, 34, -10, 3, 0 }, { 0, 3, -11, 40, 40, -11, 3, 0 } };
__constant__ int8_t cAltHpel[8] = { 0, 3, 9, 20, 20, 9, 3, 0 };
__constant__ int8_t cChroma4[17][4] = {
  { 0, 64, 0, 0 }, { -1, 63, 2, 0 }, { -2, 62, 4, 0 }, { -2, 60, 7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 }, { -4, 55, 15, -2 }, { -4, 54, 16, -2 },
  { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 } };

enum { SET_LUMA8 = 0, SET_LUMA6 = 1, SET_CHROMA4 = 2, SET_ALT = 3 };

// taps as an 8-entry window around the sample (window entry j multiplies src[(j - 3) * step]); k0/k1 = first / last entry that is read
struct Taps { int c[8]; int k0, k1; };

__device__ __forceinline__ Taps loadTaps( int set, int phase )
{
  Taps t;
#pragma unroll
  for( int j = 0; j < 8; j++ ) t.c[j] = 0;
  if( set == SET_LUMA8 || set == SET_LUMA6 )
  {
#pragma unroll
    for( int j = 0; j < 8; j++ )
    {
      const int p = phase <= 8 ? phase : 16 - phase, jj = phase <= 8 ? j : 7 - j;
      t.c[j] = set == SET_LUMA8 ? cLuma8[p][jj] : cLuma6[p][jj];
    }
    t.k0 = set == SET_LUMA8 ? 0 : 1; t.k1 = set == SET_LUMA8 ? 7 : 6;
  }
  else if( set == SET_ALT )
  {
#pragma unroll
    for( int j = 0; j < 8; j++ ) t.c[j] = cAltHpel[j];
    t.k0 = 1; t.k1 = 6;
  }
  else      // 4 chroma taps sit on window entries 2..5; phase is in 1/32
  {
#pragma unroll
    for( int j = 0; j < 4; j++ ) t.c[2 + j] = phase <= 16 ? cChroma4[phase][j] : cChroma4[32 - phase][3 - j];
    t.k0 = 2; t.k1 = 5;
  }
  return t;
}

struct PassGeom { int shift, offset, clipMax; };     // clipMax < 0: no clip

__host__ __device__ inline PassGeom passGeom( int isFirst, int isLast, int bitDepth, int taps = 8 )    // InterpolationFilter.cpp:388-423
{
  const int headRoom = 14 - bitDepth > 2 ? 14 - bitDepth : 2;
  PassGeom g; g.shift = 6; g.clipMax = isLast ? ( 1 << bitDepth ) - 1 : -1;
  if( taps == 2 )      // bilinear taps of DMVR's search (IF_FILTER_PREC_BILINEAR 4, IF_INTERNAL_PREC_BILINEAR 10), :410-423
  {
    g.shift = isFirst ? 4 - ( 10 - bitDepth ) : 4;
    g.offset = 1 << ( g.shift - 1 );
    return g;
  }
  if( isLast ) { g.shift += isFirst ? 0 : headRoom; g.offset = ( 1 << ( g.shift - 1 ) ) + ( isFirst ? 0 : ( 8192 << 6 ) ); }
  else         { g.shift -= isFirst ? headRoom : 0; g.offset = isFirst ? -( 8192 << g.shift ) : 0; }
  return g;
}

__device__ __forceinline__ int16_t finish( int sum, const PassGeom& g )
{
  int16_t v = ( int16_t ) ( ( sum + g.offset ) >> g.shift );          // Pel val (:433)
  if( g.clipMax >= 0 ) v = v < 0 ? ( int16_t ) 0 : ( v > g.clipMax ? ( int16_t ) g.clipMax : v );
  return v;
}

// -------- table-slot form: one block, one pass, caller's coefficients (m_filterHor / m_filterVer [taps][isFirst][isLast]) --------
struct SlotCoeff { int16_t c[8]; };

__global__ void __launch_bounds__( 256 )
ifSlotKernel( const int16_t* __restrict__ src, int srcStride, int16_t* __restrict__ dst, int dstStride, int width, int height, int N, int step, SlotCoeff co, PassGeom g )
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= width * height ) return;
  const int y = idx / width, x = idx - y * width;
  const int16_t* p = src + ( ptrdiff_t ) y * srcStride + x - ( N / 2 - 1 ) * step;
  int sum = 0;
  for( int k = 0; k < N; k++ ) sum += ( int ) p[( ptrdiff_t ) k * step] * co.c[k];
  dst[( ptrdiff_t ) y * dstStride + x] = finish( sum, g );
}

// m_filterCopy[isFirst][isLast] (:255-333); mode 0 copy, 1 first-not-last, 2 last-not-first, 3 first pass of DMVR's bilinear MC
__global__ void __launch_bounds__( 256 )
ifCopyKernel( const int16_t* __restrict__ src, int srcStride, int16_t* __restrict__ dst, int dstStride, int width, int height, int mode, int bitDepth )
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if( idx >= width * height ) return;
  const int y = idx / width, x = idx - y * width;
  const int16_t s = src[( ptrdiff_t ) y * srcStride + x];
  const int shift = 14 - bitDepth > 2 ? 14 - bitDepth : 2, maxv = ( 1 << bitDepth ) - 1;
  int16_t v;
  if( mode == 0 ) v = s;
  else if( mode == 1 ) v = ( int16_t ) ( ( int16_t ) ( ( uint16_t ) s << shift ) - 8192 );
  else if( mode == 3 ) v = ( int16_t ) ( s << ( 10 - bitDepth ) );
  else
  {
    const int16_t t = ( int16_t ) ( ( ( int ) s + ( int ) ( int16_t ) ( ( 1 << ( shift - 1 ) ) + 8192 ) ) >> shift );    // rightShiftU on the int promotion (:319-322)
    v = t < 0 ? ( int16_t ) 0 : ( t > maxv ? ( int16_t ) maxv : t );
  }
  dst[( ptrdiff_t ) y * dstStride + x] = v;
}

// -------- batched prediction blocks: n blocks of w x h at 1/16-sample vectors, compact output --------
// filterMode 0: the tap sets xPredInterBlk uses (8 taps; 6-tap set for 4x4 blocks); 1 / 2: the reduced sets of the fast sub-pel search
// (m_meReduceTap: 6-tap set / 4-tap chroma set); the alternative half-pel filter replaces phase 8 when useAlt (for 4x4 in mode 0: always).
__device__ __forceinline__ int tapSet( int frac, int w, int h, int filterMode, int useAlt, bool both )
{
  // 4x4 blocks: filter4x4 swaps in the alternative row for BOTH directions whatever the phase (:692-693); the 1-D dispatch only at phase 8 (:570-580)
  if( filterMode == 0 ) return ( w == 4 && h == 4 ) ? ( ( useAlt && ( both || frac == 8 ) ) ? SET_ALT : SET_LUMA6 ) : ( ( useAlt && frac == 8 ) ? SET_ALT : SET_LUMA8 );
  if( useAlt && frac == 8 ) return SET_ALT;
  return filterMode == 1 ? SET_LUMA6 : SET_CHROMA4;
}

__global__ void __launch_bounds__( 256 )
ifPredBatchKernel( const int16_t* __restrict__ ref, int refStride, const vvhip_subpel_item* __restrict__ items, int n, int w, int h, int bitDepth,
                   int rndRes, int filterMode, int useAlt, int blocksPerWg, int16_t* __restrict__ out, vvhip_dist_item* __restrict__ distItems )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmp[];       // blocksPerWg x (h + 7) x w first-pass samples
  const int tpb = 256 / blocksPerWg, sub = threadIdx.x / tpb, t = threadIdx.x - sub * tpb;
  const int blk = blockIdx.x * blocksPerWg + sub;
  const bool valid = blk < n;
  vvhip_subpel_item it = { 0, 0, 0, 0 };
  if( valid ) it = items[blk];
  const int xf = it.frac_x & 15, yf = it.frac_y & 15;
  const int16_t* src = ref + it.ref_off;
  int16_t* dst = out + ( size_t ) blk * w * h;
  int16_t* tmp = sTmp + ( size_t ) sub * ( h + 7 ) * w;
  if( valid && t == 0 && distItems ) { vvhip_dist_item d; d.org_off = it.org_off; d.cur_off = blk * w * h; distItems[blk] = d; }

  const bool both = xf != 0 && yf != 0;          // uniform per sub-block; barriers below are reached by every thread
  if( valid && both )
  {
    const Taps th = loadTaps( tapSet( xf, w, h, filterMode, useAlt, true ), filterMode == 2 && !( useAlt && xf == 8 ) ? xf << 1 : xf );
    const PassGeom g1 = passGeom( 1, 0, bitDepth );
    for( int i = t; i < ( h + 7 ) * w; i += tpb )
    {
      const int r = i / w, x = i - r * w;
      const int16_t* p = src + ( ptrdiff_t ) ( r - 3 ) * refStride + x - 3;
      int sum = 0;
#pragma unroll
      for( int j = 0; j < 8; j++ ) if( j >= th.k0 && j <= th.k1 ) sum += ( int ) p[j] * th.c[j];
      tmp[i] = finish( sum, g1 );
    }
  }
  __syncthreads();
  if( !valid ) return;
  if( both )
  {
    const Taps tv = loadTaps( tapSet( yf, w, h, filterMode, useAlt, true ), filterMode == 2 && !( useAlt && yf == 8 ) ? yf << 1 : yf );
    const PassGeom g2 = passGeom( 0, rndRes, bitDepth );
    for( int i = t; i < h * w; i += tpb )
    {
      const int y = i / w, x = i - y * w;
      const int16_t* p = tmp + y * w + x;             // row (y + 3) - 3 of the first-pass buffer
      int sum = 0;
#pragma unroll
      for( int j = 0; j < 8; j++ ) if( j >= tv.k0 && j <= tv.k1 ) sum += ( int ) p[j * w] * tv.c[j];
      dst[i] = finish( sum, g2 );
    }
  }
  else if( xf != 0 || yf != 0 )
  {
    const bool ver = xf == 0;
    const int f = ver ? yf : xf;
    const Taps tt = loadTaps( tapSet( f, w, h, filterMode, useAlt, false ), filterMode == 2 && !( useAlt && f == 8 ) ? f << 1 : f );
    const PassGeom g = passGeom( 1, rndRes, bitDepth );
    const int step = ver ? refStride : 1;
    for( int i = t; i < h * w; i += tpb )
    {
      const int y = i / w, x = i - y * w;
      const int16_t* p = src + ( ptrdiff_t ) y * refStride + x - 3 * step;
      int sum = 0;
#pragma unroll
      for( int j = 0; j < 8; j++ ) if( j >= tt.k0 && j <= tt.k1 ) sum += ( int ) p[( ptrdiff_t ) j * step] * tt.c[j];
      dst[i] = finish( sum, g );
    }
  }
  else
  {
    const int shift = 14 - bitDepth > 2 ? 14 - bitDepth : 2;
    for( int i = t; i < h * w; i += tpb )
    {
      const int y = i / w, x = i - y * w;
      const int16_t s = src[( ptrdiff_t ) y * refStride + x];
      dst[i] = rndRes ? s : ( int16_t ) ( ( int16_t ) ( ( uint16_t ) s << shift ) - 8192 );      // copy / filterCopy<true,false> (:559-565)
    }
  }
}

// ---- wide form (w a multiple of 8): a thread produces 8 horizontally adjacent samples from a register window / 16-byte LDS rows ----
struct __attribute__( ( packed, aligned( 2 ) ) ) Chunk8 { int16_t v[8]; };

__device__ __forceinline__ void mac8( int ( &acc )[8], const int ( &win )[15], const Taps& t )
{
#pragma unroll
  for( int i = 0; i < 8; i++ )
  {
    int s = 0;
#pragma unroll
    for( int j = 0; j < 8; j++ ) s = __mul24( win[i + j], t.c[j] ) + s;       // |sample| < 2^15, |tap| < 2^7: 24-bit multiply is exact
    acc[i] = s;
  }
}

// 15 samples x0-3 .. x0+11 of one row: two 8-sample loads that overlap by one sample (no read past x0+11)
__device__ __forceinline__ void loadWindow( const int16_t* rowAtX0, int ( &win )[15] )
{
  const Chunk8 a = *reinterpret_cast<const Chunk8*>( rowAtX0 - 3 ), b = *reinterpret_cast<const Chunk8*>( rowAtX0 + 4 );
#pragma unroll
  for( int k = 0; k < 8; k++ ) win[k] = a.v[k];
#pragma unroll
  for( int k = 1; k < 8; k++ ) win[7 + k] = b.v[k];
}

struct __attribute__( ( aligned( 16 ) ) ) Row8 { int16_t v[8]; };      // 16-byte aligned 8-sample segment (LDS rows, compact outputs)

__device__ __forceinline__ void store8( int16_t* dst, const int ( &acc )[8], const PassGeom& g )
{
  Row8 o;
#pragma unroll
  for( int i = 0; i < 8; i++ ) o.v[i] = finish( acc[i], g );
  *reinterpret_cast<Row8*>( dst ) = o;
}

__global__ void __launch_bounds__( 256 )
ifPredBatchWideKernel( const int16_t* __restrict__ ref, int refStride, const vvhip_subpel_item* __restrict__ items, int n, int w, int h, int bitDepth,
                       int rndRes, int filterMode, int useAlt, int blocksPerWg, int16_t* __restrict__ out, vvhip_dist_item* __restrict__ distItems )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmp[];       // blocksPerWg x (h + 7) x w first-pass samples
  const int tpb = 256 / blocksPerWg, sub = threadIdx.x / tpb, t = threadIdx.x - sub * tpb;
  const int blk = blockIdx.x * blocksPerWg + sub;
  const bool valid = blk < n;
  vvhip_subpel_item it = { 0, 0, 0, 0 };
  if( valid ) it = items[blk];
  const int xf = it.frac_x & 15, yf = it.frac_y & 15;
  const int16_t* src = ref + it.ref_off;
  int16_t* dst = out + ( size_t ) blk * w * h;
  int16_t* tmp = sTmp + ( size_t ) sub * ( h + 7 ) * w;
  if( valid && t == 0 && distItems ) { vvhip_dist_item d; d.org_off = it.org_off; d.cur_off = blk * w * h; distItems[blk] = d; }
  const int w8 = w >> 3;
  const bool both = xf != 0 && yf != 0;
  if( valid && both )
  {
    const Taps th = loadTaps( tapSet( xf, w, h, filterMode, useAlt, true ), filterMode == 2 && !( useAlt && xf == 8 ) ? xf << 1 : xf );
    const PassGeom g1 = passGeom( 1, 0, bitDepth );
    for( int i = t; i < ( h + 7 ) * w8; i += tpb )
    {
      const int r = i / w8, x0 = ( i - r * w8 ) << 3;
      int win[15], acc[8];
      loadWindow( src + ( ptrdiff_t ) ( r - 3 ) * refStride + x0, win );
      mac8( acc, win, th );
      store8( tmp + r * w + x0, acc, g1 );
    }
  }
  __syncthreads();
  if( !valid ) return;
  if( both )
  {
    const Taps tv = loadTaps( tapSet( yf, w, h, filterMode, useAlt, true ), filterMode == 2 && !( useAlt && yf == 8 ) ? yf << 1 : yf );
    const PassGeom g2 = passGeom( 0, rndRes, bitDepth );
    for( int i = t; i < h * w8; i += tpb )
    {
      const int y = i / w8, x0 = ( i - y * w8 ) << 3;
      int acc[8];
#pragma unroll
      for( int k = 0; k < 8; k++ ) acc[k] = 0;
#pragma unroll
      for( int j = 0; j < 8; j++ )
      {
        const Row8 row = *reinterpret_cast<const Row8*>( tmp + ( y + j ) * w + x0 );      // 16-byte aligned LDS row segment
#pragma unroll
        for( int k = 0; k < 8; k++ ) acc[k] = __mul24( ( int ) row.v[k], tv.c[j] ) + acc[k];
      }
      store8( dst + y * w + x0, acc, g2 );
    }
  }
  else if( xf != 0 )
  {
    const Taps tt = loadTaps( tapSet( xf, w, h, filterMode, useAlt, false ), filterMode == 2 && !( useAlt && xf == 8 ) ? xf << 1 : xf );
    const PassGeom g = passGeom( 1, rndRes, bitDepth );
    for( int i = t; i < h * w8; i += tpb )
    {
      const int y = i / w8, x0 = ( i - y * w8 ) << 3;
      int win[15], acc[8];
      loadWindow( src + ( ptrdiff_t ) y * refStride + x0, win );
      mac8( acc, win, tt );
      store8( dst + y * w + x0, acc, g );
    }
  }
  else if( yf != 0 )
  {
    const Taps tt = loadTaps( tapSet( yf, w, h, filterMode, useAlt, false ), filterMode == 2 && !( useAlt && yf == 8 ) ? yf << 1 : yf );
    const PassGeom g = passGeom( 1, rndRes, bitDepth );
    for( int i = t; i < h * w8; i += tpb )
    {
      const int y = i / w8, x0 = ( i - y * w8 ) << 3;
      int acc[8];
#pragma unroll
      for( int k = 0; k < 8; k++ ) acc[k] = 0;
#pragma unroll
      for( int j = 0; j < 8; j++ )
        if( j >= tt.k0 && j <= tt.k1 )          // rows outside the tap support are not touched (they may lie outside the picture margin)
        {
          const Chunk8 row = *reinterpret_cast<const Chunk8*>( src + ( ptrdiff_t ) ( y + j - 3 ) * refStride + x0 );
#pragma unroll
          for( int k = 0; k < 8; k++ ) acc[k] = __mul24( ( int ) row.v[k], tt.c[j] ) + acc[k];
        }
      store8( dst + y * w + x0, acc, g );
    }
  }
  else
  {
    const int shift = 14 - bitDepth > 2 ? 14 - bitDepth : 2;
    for( int i = t; i < h * w8; i += tpb )
    {
      const int y = i / w8, x0 = ( i - y * w8 ) << 3;
      Chunk8 c = *reinterpret_cast<const Chunk8*>( src + ( ptrdiff_t ) y * refStride + x0 );
      if( !rndRes )
#pragma unroll
        for( int k = 0; k < 8; k++ ) c.v[k] = ( int16_t ) ( ( int16_t ) ( ( uint16_t ) c.v[k] << shift ) - 8192 );
      Row8 o;
#pragma unroll
      for( int k = 0; k < 8; k++ ) o.v[k] = c.v[k];
      *reinterpret_cast<Row8*>( dst + y * w + x0 ) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Pattern refinement (InterSearch::xPatternRefinement, EncoderLib/InterSearch.cpp:760-880): K sub-pel positions around ONE base vector per
// block.  One block per 64 or 256 threads: the reference window (h+9) x (w+10) is staged in LDS once (coalesced HBM reads); the
// horizontal pass runs once per DISTINCT horizontal offset (3 for the 8-neighbour stages — the reference shares its planes the same
// way) over all window rows into LDS; the vertical passes of all K positions then run concurrently and write the prediction blocks
// (compact, candidate-list order) for the distortion kernels.  Every position is horizontal-then-vertical with the copy forms for a
// zero phase: the same values as the single-pass forms the reference's dispatch uses for one-directional vectors (floor(floor(a)+b)/c
// = floor((a+b)/c) for integer b, c).
// ---------------------------------------------------------------------------------------------
struct RefineOffsets { int n, nHor; int16_t dx[16], dy[16], horDx[16]; int8_t horOf[16]; };

// base vectors x offsets -> explicit candidate list (large blocks go through the per-candidate kernel: enough parallelism per candidate)
__global__ void __launch_bounds__( 256 )
refineExpandKernel( const vvhip_subpel_item* __restrict__ bases, int nBlocks, int refStride, RefineOffsets offs, vvhip_subpel_item* __restrict__ cand )
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if( i >= nBlocks * offs.n ) return;
  const int b = i / offs.n, k = i - b * offs.n;
  const vvhip_subpel_item base = bases[b];
  const int tx = ( base.frac_x & 15 ) + offs.dx[k], ty = ( base.frac_y & 15 ) + offs.dy[k];
  vvhip_subpel_item c; c.org_off = base.org_off; c.ref_off = base.ref_off + ( ty >> 4 ) * refStride + ( tx >> 4 ); c.frac_x = ( int16_t ) ( tx & 15 ); c.frac_y = ( int16_t ) ( ty & 15 );
  cand[i] = c;
}

__global__ void __launch_bounds__( 256 )
refinePredKernel( const int16_t* __restrict__ ref, int refStride, const vvhip_subpel_item* __restrict__ bases, int nBlocks, int w, int h, int bitDepth, int filterMode,
                  int useAlt, int blocksPerWg, RefineOffsets offs, int16_t* __restrict__ pred, vvhip_dist_item* __restrict__ distItems )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sRef[];
  const int tpb = 256 / blocksPerWg, sub = threadIdx.x / tpb, t = threadIdx.x - sub * tpb;
  const int blk = blockIdx.x * blocksPerWg + sub;
  const bool valid = blk < nBlocks;
  const int wp = w + 10, rows = h + 9, w8 = w >> 3;
  const int oTmp = ( rows * wp + 7 ) & ~7, slotElems = oTmp + offs.nHor * rows * w;            // every region starts on 16 bytes (w is a multiple of 8)
  int16_t* win = sRef + ( size_t ) sub * slotElems;      // window rows y-4 .. y+h+4, columns x-4 .. x+w+5
  int16_t* tmp = win + oTmp;                              // [horizontal variant][window row][w]: first-pass samples, column shift applied
  vvhip_subpel_item base = { 0, 0, 0, 0 };
  if( valid )
  {
    base = bases[blk];
    const int16_t* src = ref + base.ref_off - 4 * ( ptrdiff_t ) refStride - 4;
    for( int e = t; e < rows * wp; e += tpb ) { const int r = e / wp, c = e - r * wp; win[e] = src[( ptrdiff_t ) r * refStride + c]; }
    if( blockIdx.y == 0 ) for( int k = t; k < offs.n; k += tpb ) { vvhip_dist_item d; d.org_off = base.org_off; d.cur_off = ( blk * offs.n + k ) * w * h; distItems[( size_t ) blk * offs.n + k] = d; }
  }
#define REFINE_SYNC() { if( tpb == 64 ) { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); }
  REFINE_SYNC();
  const int sh = 14 - bitDepth > 2 ? 14 - bitDepth : 2, maxv = ( 1 << bitDepth ) - 1;
  if( valid )
  {
    // ---- horizontal pass, every distinct horizontal offset x every window row: tmp[v][r][x] <-> picture column x + sx
    const PassGeom g1 = passGeom( 1, 0, bitDepth );
    for( int i = t; i < offs.nHor * rows * w8; i += tpb )
    {
      const int v = i / ( rows * w8 ), rem = i - v * rows * w8, r = rem / w8, x0 = ( rem - r * w8 ) << 3;
      const int txx = ( base.frac_x & 15 ) + offs.horDx[v], sx = txx >> 4, fx = txx & 15;
      int16_t* dst = tmp + ( v * rows + r ) * w + x0;
      if( fx )
      {
        const Taps th = loadTaps( tapSet( fx, w, h, filterMode, useAlt, true ), filterMode == 2 && !( useAlt && fx == 8 ) ? fx << 1 : fx );
        const int16_t* p = win + r * wp + x0 + sx + 1;
        int wv[15], accv[8];
#pragma unroll
        for( int j = 0; j < 15; j++ ) wv[j] = p[j];
        mac8( accv, wv, th );
        store8( dst, accv, g1 );
      }
      else
      {
        Row8 o;
#pragma unroll
        for( int c = 0; c < 8; c++ ) o.v[c] = ( int16_t ) ( ( int16_t ) ( ( uint16_t ) win[r * wp + x0 + c + sx + 4] << sh ) - 8192 );       // filterCopy<true,false>
        *reinterpret_cast<Row8*>( dst ) = o;
      }
    }
  }
  REFINE_SYNC();
  if( !valid ) return;
  // ---- vertical passes of this workgroup's share of the positions (gridDim.y splits them when there are few blocks): prediction row y <->
  // window rows y + sy + 1 .. + 8 (taps), last pass (clip)
  const PassGeom g2 = passGeom( 0, 1, bitDepth );
  const int kPer = ( offs.n + gridDim.y - 1 ) / gridDim.y, k0 = blockIdx.y * kPer, k1 = min( offs.n, k0 + kPer );
  for( int i = t; i < ( k1 - k0 ) * h * w8; i += tpb )
  {
    const int k = k0 + i / ( h * w8 ), rem = i - ( k - k0 ) * h * w8, y = rem / w8, x0 = ( rem - y * w8 ) << 3;
    const int tyy = ( base.frac_y & 15 ) + offs.dy[k], sy = tyy >> 4, fy = tyy & 15;
    const int16_t* tv_ = tmp + offs.horOf[k] * rows * w;
    int16_t* dst = pred + ( ( size_t ) blk * offs.n + k ) * w * h + y * w + x0;
    if( fy )
    {
      const Taps tv = loadTaps( tapSet( fy, w, h, filterMode, useAlt, true ), filterMode == 2 && !( useAlt && fy == 8 ) ? fy << 1 : fy );
      int accv[8];
#pragma unroll
      for( int c = 0; c < 8; c++ ) accv[c] = 0;
#pragma unroll
      for( int j = 0; j < 8; j++ )
      {
        const Row8 row = *reinterpret_cast<const Row8*>( tv_ + ( y + sy + 1 + j ) * w + x0 );
#pragma unroll
        for( int c = 0; c < 8; c++ ) accv[c] = __mul24( ( int ) row.v[c], tv.c[j] ) + accv[c];
      }
      store8( dst, accv, g2 );
    }
    else
    {
      const Row8 row = *reinterpret_cast<const Row8*>( tv_ + ( y + sy + 4 ) * w + x0 );
      Row8 o;
#pragma unroll
      for( int c = 0; c < 8; c++ )
      {
        const int16_t v = ( int16_t ) ( ( ( int ) row.v[c] + ( int ) ( int16_t ) ( ( 1 << ( sh - 1 ) ) + 8192 ) ) >> sh );     // filterCopy<false,true>
        o.v[c] = v < 0 ? ( int16_t ) 0 : ( v > maxv ? ( int16_t ) maxv : v );
      }
      *reinterpret_cast<Row8*>( dst ) = o;
    }
  }
#undef REFINE_SYNC
}

} // namespace

extern "C" {

int vvhip_if_filter( vvhip_ctx* ctx, int taps, int is_vertical, int is_first, int is_last, int bit_depth,
                     const int16_t* d_src, int src_stride, int16_t* d_dst, int dst_stride, int width, int height, const int16_t* coeff_host )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( ( taps != 8 && taps != 6 && taps != 4 && taps != 2 ) || ( taps == 2 && is_first && bit_depth > 10 ) || !d_src || !d_dst || !coeff_host || width < 1 || height < 1 || width > 4096 || height > 4096 || bit_depth < 8 || bit_depth > 12 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_if_filter: taps %d block %dx%d bitDepth %d", taps, width, height, bit_depth );
  SlotCoeff co;
  for( int k = 0; k < 8; k++ ) co.c[k] = 0;
  for( int k = 0; k < taps; k++ ) co.c[k] = coeff_host[k + ( taps == 6 ? 1 : 0 )];         // the 6-tap cores skip the row's first entry (:361-364)
  const PassGeom g = passGeom( is_first != 0, is_last != 0, bit_depth, taps );
  hipLaunchKernelGGL( ifSlotKernel, dim3( ( width * height + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_src, src_stride, d_dst, dst_stride, width, height, taps,
                      is_vertical ? src_stride : 1, co, g );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_if_copy( vvhip_ctx* ctx, int is_first, int is_last, int bit_depth, const int16_t* d_src, int src_stride, int16_t* d_dst, int dst_stride,
                   int width, int height, int bi_mc_for_dmvr )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_src || !d_dst || width < 1 || height < 1 || width > 4096 || height > 4096 || bit_depth < 8 || bit_depth > 12 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_if_copy: bad arguments" );
  const int mode = ( !is_first == !is_last ) ? 0 : is_first ? ( bi_mc_for_dmvr ? 3 : 1 ) : 2;
  hipLaunchKernelGGL( ifCopyKernel, dim3( ( width * height + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_src, src_stride, d_dst, dst_stride, width, height, mode, bit_depth );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

static int launchPred( vvhip_ctx* ctx, const int16_t* d_ref, int ref_stride, const vvhip_subpel_item* d_items, int n, int width, int height, int bit_depth,
                       int rnd_res, int filter_mode, int use_alt_hpel, int16_t* d_out, vvhip_dist_item* d_dist_items )
{
  if( ( width & 7 ) == 0 )
  {
    int tpb = 16; while( tpb < 256 && tpb * 8 < width * height ) tpb <<= 1;          // one thread per 8 output samples, at least 16 threads per block
    const int bpw = 256 / tpb;
    const size_t smem = ( size_t ) bpw * ( height + 7 ) * width * sizeof( int16_t );
    hipLaunchKernelGGL( ifPredBatchWideKernel, dim3( ( n + bpw - 1 ) / bpw ), dim3( 256 ), smem, ctx->stream, d_ref, ref_stride, d_items, n, width, height, bit_depth,
                        rnd_res ? 1 : 0, filter_mode, use_alt_hpel ? 1 : 0, bpw, d_out, d_dist_items );
    VVHIP_LAUNCH_CHECK( ctx );
    return VVHIP_OK;
  }
  int bpw = 256 / ( width * height ); if( bpw < 1 ) bpw = 1; if( bpw > 8 ) bpw = 8;
  const size_t smem = ( size_t ) bpw * ( height + 7 ) * width * sizeof( int16_t );
  hipLaunchKernelGGL( ifPredBatchKernel, dim3( ( n + bpw - 1 ) / bpw ), dim3( 256 ), smem, ctx->stream, d_ref, ref_stride, d_items, n, width, height, bit_depth,
                      rnd_res ? 1 : 0, filter_mode, use_alt_hpel ? 1 : 0, bpw, d_out, d_dist_items );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

static bool predArgsOk( int width, int height, int bit_depth, int filter_mode, int n )
{
  return width >= 4 && height >= 4 && width <= 128 && height <= 128 && bit_depth >= 8 && bit_depth <= 12 && filter_mode >= 0 && filter_mode <= 2 && n >= 0;
}

int vvhip_interp_luma_batch( vvhip_ctx* ctx, const int16_t* d_ref, int ref_stride, const vvhip_subpel_item* d_items, int n,
                             int width, int height, int bit_depth, int rnd_res, int filter_mode, int use_alt_hpel, int16_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !predArgsOk( width, height, bit_depth, filter_mode, n ) || ( n && ( !d_ref || !d_items || !d_out ) ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_interp_luma_batch: block %dx%d bitDepth %d mode %d", width, height, bit_depth, filter_mode );
  if( n == 0 ) return VVHIP_OK;
  return launchPred( ctx, d_ref, ref_stride, d_items, n, width, height, bit_depth, rnd_res, filter_mode, use_alt_hpel, d_out, nullptr );
}

int vvhip_subpel_dist_batch( vvhip_ctx* ctx, int func, const int16_t* d_org, int org_stride, const int16_t* d_ref, int ref_stride,
                             int width, int height, int bit_depth, int filter_mode, int use_alt_hpel,
                             const vvhip_subpel_item* d_items, int n, uint64_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !predArgsOk( width, height, bit_depth, filter_mode, n ) || ( n && ( !d_org || !d_ref || !d_items || !d_out ) ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_subpel_dist_batch: block %dx%d bitDepth %d mode %d", width, height, bit_depth, filter_mode );
  if( n == 0 ) return VVHIP_OK;
  // predictions go to a grow-only scratch (compact blocks), then the ordinary distortion kernels score them against the originals
  const size_t predBytes = ( ( size_t ) n * width * height * sizeof( int16_t ) + 255 ) & ~( size_t ) 255, need = predBytes + ( size_t ) n * sizeof( vvhip_dist_item );
  if( need > ctx->subpelBytes )
  {
    VVHIP_CHECK_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
    if( ctx->d_subpel ) VVHIP_CHECK_HIP( ctx, hipFree( ctx->d_subpel ) );
    ctx->d_subpel = nullptr; ctx->subpelBytes = 0;
    VVHIP_CHECK_HIP( ctx, hipMalloc( &ctx->d_subpel, need + need / 4 ) );
    ctx->subpelBytes = need + need / 4;
  }
  int16_t* pred = static_cast<int16_t*>( ctx->d_subpel );
  vvhip_dist_item* di = reinterpret_cast<vvhip_dist_item*>( static_cast<char*>( ctx->d_subpel ) + predBytes );
  const int rc = launchPred( ctx, d_ref, ref_stride, d_items, n, width, height, bit_depth, 1, filter_mode, use_alt_hpel, pred, di );
  if( rc ) return rc;
  return vvhip_dist_batch( ctx, func, d_org, org_stride, pred, width, width, height, 0, bit_depth, di, n, d_out );
}

int vvhip_subpel_refine_batch( vvhip_ctx* ctx, int func, const int16_t* d_org, int org_stride, const int16_t* d_ref, int ref_stride,
                               int width, int height, int bit_depth, int filter_mode, int use_alt_hpel,
                               const vvhip_subpel_item* d_bases, int n_blocks, const int16_t* offsets_host, int n_offsets, uint64_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !predArgsOk( width, height, bit_depth, filter_mode, n_blocks ) || ( width & 7 ) || width > 64 || height > 64 || n_offsets < 1 || n_offsets > 16 || !offsets_host ||
      ( n_blocks && ( !d_org || !d_ref || !d_bases || !d_out ) ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_subpel_refine_batch: block %dx%d (width a multiple of 8, up to 64x64), %d offsets (1..16)", width, height, n_offsets );
  RefineOffsets ro; ro.n = n_offsets; ro.nHor = 0;
  for( int k = 0; k < 16; k++ ) { ro.dx[k] = k < n_offsets ? offsets_host[2 * k] : 0; ro.dy[k] = k < n_offsets ? offsets_host[2 * k + 1] : 0; ro.horDx[k] = 0; ro.horOf[k] = 0; }
  for( int k = 0; k < n_offsets; k++ )
  {
    if( ro.dx[k] < -16 || ro.dx[k] > 16 || ro.dy[k] < -16 || ro.dy[k] > 16 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_subpel_refine_batch: offsets are limited to +-16 (one sample)" );
    int v = 0; while( v < ro.nHor && ro.horDx[v] != ro.dx[k] ) v++;
    if( v == ro.nHor ) ro.horDx[ro.nHor++] = ro.dx[k];          // distinct horizontal offsets share one first pass
    ro.horOf[k] = ( int8_t ) v;
  }
  if( n_blocks == 0 ) return VVHIP_OK;
  const size_t n = ( size_t ) n_blocks * n_offsets;
  const size_t predBytes = ( n * width * height * sizeof( int16_t ) + 255 ) & ~( size_t ) 255, itemBytes = ( n * sizeof( vvhip_dist_item ) + 255 ) & ~( size_t ) 255;
  const size_t need = predBytes + itemBytes + n * sizeof( vvhip_subpel_item );
  if( need > ctx->subpelBytes )
  {
    VVHIP_CHECK_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
    if( ctx->d_subpel ) VVHIP_CHECK_HIP( ctx, hipFree( ctx->d_subpel ) );
    ctx->d_subpel = nullptr; ctx->subpelBytes = 0;
    VVHIP_CHECK_HIP( ctx, hipMalloc( &ctx->d_subpel, need + need / 4 ) );
    ctx->subpelBytes = need + need / 4;
  }
  int16_t* pred = static_cast<int16_t*>( ctx->d_subpel );
  vvhip_dist_item* di = reinterpret_cast<vvhip_dist_item*>( static_cast<char*>( ctx->d_subpel ) + predBytes );
  if( width * height > 1024 )
  {
    vvhip_subpel_item* cand = reinterpret_cast<vvhip_subpel_item*>( static_cast<char*>( ctx->d_subpel ) + predBytes + itemBytes );
    hipLaunchKernelGGL( refineExpandKernel, dim3( ( unsigned ) ( ( n + 255 ) / 256 ) ), dim3( 256 ), 0, ctx->stream, d_bases, n_blocks, ref_stride, ro, cand );
    VVHIP_LAUNCH_CHECK( ctx );
    const int rc = launchPred( ctx, d_ref, ref_stride, cand, ( int ) n, width, height, bit_depth, 1, filter_mode, use_alt_hpel, pred, di );
    if( rc ) return rc;
    return vvhip_dist_batch( ctx, func, d_org, org_stride, pred, width, width, height, 0, bit_depth, di, ( int ) n, d_out );
  }
  const int rows = height + 9, wp = width + 10;
  const size_t slotElems = ( size_t ) ( ( rows * wp + 7 ) & ~7 ) + ( size_t ) ro.nHor * rows * width;
  int tpb = width * height <= 256 ? 64 : 256;
  if( slotElems * ( 256 / tpb ) * sizeof( int16_t ) > 150 * 1024 ) tpb = 256;
  const int bpw = 256 / tpb;
  const size_t smem = slotElems * bpw * sizeof( int16_t );
  if( smem > 150 * 1024 ) return vvhip_fail( ctx, VVHIP_E_UNSUPPORTED, "vvhip_subpel_refine_batch: %d distinct horizontal offsets of a %dx%d block need %zu B of LDS", ro.nHor, width, height, smem );
  if( smem > 64 * 1024 ) VVHIP_CHECK_HIP( ctx, hipFuncSetAttribute( ( const void* ) refinePredKernel, hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) smem ) );
  const int nWg = ( n_blocks + bpw - 1 ) / bpw;
  const int splitK = 1;      // (the kernel can spread the positions over gridDim.y workgroups; re-staging the window costs more than it gains)
  hipLaunchKernelGGL( refinePredKernel, dim3( nWg, splitK ), dim3( 256 ), smem, ctx->stream, d_ref, ref_stride, d_bases, n_blocks, width, height,
                      bit_depth, filter_mode, use_alt_hpel ? 1 : 0, bpw, ro, pred, di );
  VVHIP_LAUNCH_CHECK( ctx );
  return vvhip_dist_batch( ctx, func, d_org, org_stride, pred, width, width, height, 0, bit_depth, di, ( int ) n, d_out );
}

} // extern "C"
