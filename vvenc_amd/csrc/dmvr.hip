// dmvr.hip — SURVEY §8f rank 3: decoder-side motion vector refinement search (decoder-normative, all integer).
//
// Reference behaviour:
//   DMVR::xProcessDMVR, refinement search part          CommonLib/InterPrediction.cpp:1262-1392
//     bilinear prediction of both lists (mergeMV - 2)   :1280-1302 -> InterpolationFilter::filterN2_2D  InterpolationFilter.cpp:662-681
//     centre cost, early exit                           :1330-1337
//     25-point mirrored search with dmvrSadX5           :1344-1366 (RdCost::xGetSAD8X5/16X5, RdCost.cpp:1984-2034)
//     parametric sub-pel error surface                  :1227-1244, xSubPelErrorSrfc :1167-1187, div_for_maxq7 :1131-1165
// One wavefront per sub-block (<= 16x16): both bilinear predictions (dx+4) x (dy+4) are built in LDS, 50 lanes evaluate the 25 mirrored
// positions (two row halves each), one lane replays the reference's scan order (strict <) and the error surface.
#include "common.h"

namespace {

constexpr int DMVR_PITCH = 20;                 // dx + 4 <= 20
constexpr int DMVR_ELEMS = DMVR_PITCH * 20;

__device__ __forceinline__ int16_t bilinearSample( const int16_t* p, int stride, int fx, int fy, int bitDepth )
{
  // filterN2_2D: both fractions -> horizontal first pass on rows y, y+1 (isFirst, not last), vertical second pass; one fraction -> a single
  // first pass; none -> filterCopy<true,false>(biMCForDMVR): sample << (10 - bitDepth).  Every pass truncates to Pel.
  const int sh1 = 4 - ( 10 - bitDepth ), of1 = 1 << ( sh1 - 1 );
  if( fx && fy )
  {
    const int16_t t0 = ( int16_t ) ( ( ( 16 - fx ) * p[0] + fx * p[1] + of1 ) >> sh1 );
    const int16_t t1 = ( int16_t ) ( ( ( 16 - fx ) * p[stride] + fx * p[stride + 1] + of1 ) >> sh1 );
    return ( int16_t ) ( ( ( 16 - fy ) * t0 + fy * t1 + 8 ) >> 4 );
  }
  if( fx ) return ( int16_t ) ( ( ( 16 - fx ) * p[0] + fx * p[1] + of1 ) >> sh1 );
  if( fy ) return ( int16_t ) ( ( ( 16 - fy ) * p[0] + fy * p[stride] + of1 ) >> sh1 );
  return ( int16_t ) ( p[0] << ( 10 - bitDepth ) );
}

__device__ __forceinline__ int divMaxQ7( long long N, long long D )       // div_for_maxq7
{
  int sign = 0, q = 0;
  if( N < 0 ) { sign = 1; N = -N; }
  D <<= 3;
  if( N >= D ) { N -= D; q++; }
  q <<= 1;
  D >>= 1;
  if( N >= D ) { N -= D; q++; }
  q <<= 1;
  if( N >= ( D >> 1 ) ) q++;
  return sign ? -q : q;
}

__global__ void __launch_bounds__( 256 )
dmvrRefineKernel( const int16_t* __restrict__ ref0, int stride0, const int16_t* __restrict__ ref1, int stride1, const vvhip_dmvr_item* __restrict__ items, int n,
                  int dx, int dy, int bitDepth, vvhip_dmvr_result* __restrict__ out )
{
  __shared__ int16_t sPred[4][2][DMVR_ELEMS];
  __shared__ uint32_t sCost[4][25][2];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * 4 + wv;
  if( blk >= n ) return;                                   // whole waves leave together; no workgroup barrier below
  const vvhip_dmvr_item it = items[blk];
  int16_t* p0 = sPred[wv][0]; int16_t* p1 = sPred[wv][1];
  const int bw = dx + 4, bh = dy + 4;
  const int16_t* s0 = ref0 + it.ref0_off - 2 * stride0 - 2;       // mergeMV - (2, 2) samples (:1285-1288)
  const int16_t* s1 = ref1 + it.ref1_off - 2 * stride1 - 2;
  for( int e = lane; e < bw * bh; e += 64 )
  {
    const int y = e / bw, x = e - y * bw;
    p0[y * DMVR_PITCH + x] = bilinearSample( s0 + ( ptrdiff_t ) y * stride0 + x, stride0, it.frac0_x & 15, it.frac0_y & 15, bitDepth );
    p1[y * DMVR_PITCH + x] = bilinearSample( s1 + ( ptrdiff_t ) y * stride1 + x, stride1, it.frac1_x & 15, it.frac1_y & 15, bitDepth );
  }
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();

  // mirrored SAD on every second row (subShift 1; (sum << 1) >> 1 == sum): position q = (ver + 2) * 5 + hor + 2, lanes 2q, 2q+1 take the row halves
  const int16_t* c0 = p0 + 2 * DMVR_PITCH + 2; const int16_t* c1 = p1 + 2 * DMVR_PITCH + 2;
  if( lane < 50 )
  {
    const int q = lane >> 1, half = lane & 1, ver = q / 5 - 2, hor = q - ( q / 5 ) * 5 - 2;
    const int off = hor + ver * DMVR_PITCH;
    const int rows = dy >> 1, r0 = half * ( ( rows + 1 ) >> 1 ), r1 = half ? rows : ( ( rows + 1 ) >> 1 );
    uint32_t sum = 0;
    for( int r = r0; r < r1; r++ )
      for( int x = 0; x < dx; x++ )
      {
        const int d = ( int ) c0[2 * r * DMVR_PITCH + x + off] - ( int ) c1[2 * r * DMVR_PITCH + x - off];
        sum += ( uint32_t ) ( d < 0 ? -d : d );
      }
    sCost[wv][q][half] = sum;
  }
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();

  if( lane == 0 )
  {
    unsigned long long sad[25];
#pragma unroll
    for( int q = 0; q < 25; q++ ) sad[q] = ( unsigned long long ) sCost[wv][q][0] + sCost[wv][q][1];
    // centre: distFunc(SAD, subShift 1) >> 1, minus a quarter (:1332-1333); the X5 costs are SAD >> 1 without that reduction
    unsigned long long minCost = sad[12];
    minCost -= minCost >> 2;
    int tx = 0, ty = 0;
    if( minCost >= ( unsigned long long ) ( dx * dy ) )
    {
      sad[12] = minCost;
      int bh_ = 0, bv_ = 0;
      for( int ver = -2; ver <= 2; ver++ )
        for( int hor = -2; hor <= 2; hor++ )
        {
          const unsigned long long cost = sad[( ver + 2 ) * 5 + hor + 2];
          if( cost < minCost ) { minCost = cost; bh_ = hor; bv_ = ver; }
        }
      tx = bh_ * 16; ty = bv_ * 16;
      if( bh_ != 2 && bh_ != -2 && bv_ != 2 && bv_ != -2 )                       // xDMVRSubPixelErrorSurface (:1230-1231)
      {
        const unsigned long long* p = &sad[12 + bv_ * 5 + bh_];
        const unsigned long long sb[5] = { p[0], p[-1], p[-5], p[1], p[5] };
        int t[2] = { 0, 0 };
#pragma unroll
        for( int hv = 0; hv < 2; hv++ )
        {
          const long long num = ( long long ) ( ( sb[hv + 1] - sb[hv + 3] ) << 4 );
          const long long den = ( long long ) ( sb[hv + 1] + sb[hv + 3] - ( sb[0] << 1 ) );
          if( den != 0 )
          {
            if( sb[hv + 1] != sb[0] && sb[hv + 3] != sb[0] ) t[hv] = divMaxQ7( num, den );
            else t[hv] = sb[hv + 1] == sb[0] ? -8 : 8;
          }
        }
        tx += t[0]; ty += t[1];
      }
    }
    vvhip_dmvr_result r; r.mvd_x = ( int16_t ) tx; r.mvd_y = ( int16_t ) ty; r.pad = 0; r.min_cost = minCost;
    out[blk] = r;
  }
}

} // namespace

extern "C" {

int vvhip_dmvr_refine_batch( vvhip_ctx* ctx, const int16_t* d_ref0, int stride0, const int16_t* d_ref1, int stride1, const vvhip_dmvr_item* d_items, int n,
                             int dx, int dy, int bit_depth, vvhip_dmvr_result* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n < 0 || ( dx != 8 && dx != 16 ) || ( dy != 8 && dy != 16 ) || bit_depth < 8 || bit_depth > 10 || ( n && ( !d_ref0 || !d_ref1 || !d_items || !d_out ) ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dmvr_refine_batch: sub-block %dx%d (8 or 16 per side, DMVR_SUBCU_SIZE 16) bitDepth %d (<= 10: the bilinear taps keep 10-bit precision)", dx, dy, bit_depth );
  if( n == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( dmvrRefineKernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, d_ref0, stride0, d_ref1, stride1, d_items, n, dx, dy, bit_depth, d_out );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

} // extern "C"
