// dmvr.hip — SURVEY §8f rank 3: decoder-side motion vector refinement search (decoder-normative, all integer).
//
// Reference behaviour:
//   DMVR::xProcessDMVR, refinement search part          CommonLib/InterPrediction.cpp:1262-1392
//     bilinear prediction of both lists (mergeMV - 2)   :1280-1302 -> InterpolationFilter::filterN2_2D  InterpolationFilter.cpp:662-681
//     centre cost, early exit                           :1330-1337
//     25-point mirrored search with dmvrSadX5           :1344-1366 (RdCost::xGetSAD8X5/16X5, RdCost.cpp:1984-2034)
//     parametric sub-pel error surface                  :1227-1244, xSubPelErrorSrfc :1167-1187, div_for_maxq7 :1131-1165
// One wavefront per sub-block (<= 16x16).  Three lane-parallel steps through LDS, then a one-lane epilogue:
//   1  rows 0 .. dy+4 of both source windows straight from the planes: a lane = 8 adjacent samples of a row (one 16-byte load + one dword), horizontal taps as
//      v_dot2_i32_i16 on sample pairs (v_alignbit makes the odd pairs), Pel truncation by packing;
//   2  vertical taps on row pairs (v_perm interleaves the rows for v_dot2), results stored biased for v_sad_u16;
//   3  25 mirrored positions x dy/2 rows: dy/2 lanes per position (lane = row), a row is dx/2 dwords of each prediction; the 15 positions with an even horizontal offset read
//      whole dwords, the 10 odd ones go through v_alignbit — separate passes (wave-uniform code), rows fully unrolled (the sub-block size is a template parameter); DPP sum;
//   the reference's scan (strict <, centre first) as ONE packed minimum over 25 lanes: key = cost << 6 | (centre ? 0 : raster index + 1); then the error surface.
#include "common.h"

namespace {

constexpr int DP = 24;                         // LDS row pitch in samples (dx + 4 <= 20 used, dword reads reach 22)

typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
typedef short dmvrS2 __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ int dmvrDot2( uint32_t a, uint32_t b, int c ) { return __builtin_amdgcn_sdot2( __builtin_bit_cast( dmvrS2, a ), __builtin_bit_cast( dmvrS2, b ), c, false ); }
__device__ __forceinline__ uint32_t dmvrPack( int lo, int hi ) { return ( ( uint32_t ) lo & 0xffffu ) | ( ( uint32_t ) hi << 16 ); }      // truncation to Pel like the reference's casts

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

VVHIP_GROUP_REDUCE( dmvrGroupMin32, ( o < v ? o : v ) )

template<int dx, int dy>
__global__ void __launch_bounds__( 256 )
dmvrRefineKernel( const int16_t* __restrict__ ref0, int stride0, const int16_t* __restrict__ ref1, int stride1, const vvhip_dmvr_item* __restrict__ items, int n,
                  int bitDepth, vvhip_dmvr_result* __restrict__ out )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sT[4][2][22 * DP];      // first-pass rows
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sR[4][2][20 * DP];      // bilinear predictions, biased
  __shared__ uint32_t sCost[4][32];
  struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
  struct __attribute__( ( packed, aligned( 2 ) ) ) U4 { uint32_t v; };
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // workgroups are dealt round-robin to the 8 XCDs: XCD x takes the x-th contiguous eighth of the list (a caller's list is in picture order: every private L2 then streams one
  // band of the two reference planes instead of sub-blocks from the whole picture; results do not depend on it)
  const int q = ( int ) gridDim.x >> 3, g = ( int ) blockIdx.x < ( q << 3 ) ? ( ( int ) blockIdx.x & 7 ) * q + ( ( int ) blockIdx.x >> 3 ) : ( int ) blockIdx.x;
  const int blk = g * 4 + wv;
  if( blk >= n ) return;                                   // whole waves leave together; no workgroup barrier below
  const vvhip_dmvr_item it = items[blk];
  constexpr int bw = dx + 4, bh = dy + 4, segs = ( bw + 7 ) >> 3;
  const int sh1 = 4 - ( 10 - bitDepth ), of1 = 1 << ( sh1 - 1 );
  const int16_t* s0 = ref0 + it.ref0_off - 2 * stride0 - 2;       // mergeMV - (2, 2) samples (:1285-1288)
  const int16_t* s1 = ref1 + it.ref1_off - 2 * stride1 - 2;
#define DMVR_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  // ---- 1: filterN2_2D first pass (fraction x != 0) or the samples themselves, rows 0 .. bh (the second pass of row bh - 1 reads row bh).  Both trips' loads first.
  {
    const int perList = ( bh + 1 ) * segs, jobs = 2 * perList;
    u32x4 A[2]; uint32_t B[2]; int at[2], fxs[2]; bool ok[2];
#pragma unroll
    for( int q = 0; q < 2; q++ )
    {
      const int jb = lane + 64 * q;
      ok[q] = jb < jobs;
      const int jv = ok[q] ? jb : 0, list = jv >= perList, rem = jv - list * perList, y = segs == 3 ? rem / 3 : rem >> 1, x0 = ( rem - y * segs ) * 8;      // segs is 2 or 3
      const int16_t* src = ( list ? s1 + ( ptrdiff_t ) y * stride1 : s0 + ( ptrdiff_t ) y * stride0 ) + x0;
      fxs[q] = ( list ? it.frac1_x : it.frac0_x ) & 15;
      at[q] = list * 22 * DP + y * DP + x0;
      A[q] = reinterpret_cast<const U16*>( src )->v; B[q] = reinterpret_cast<const U4*>( src + 8 )->v;
    }
#pragma unroll
    for( int q = 0; q < 2; q++ )
    {
      if( !ok[q] ) continue;
      u32x4 o = A[q];
      if( fxs[q] )
      {
        const uint32_t w = dmvrPack( 16 - fxs[q], fxs[q] );
        const uint32_t e[5] = { A[q].x, A[q].y, A[q].z, A[q].w, B[q] };
        uint32_t r[4];
#pragma unroll
        for( int i = 0; i < 4; i++ )
          r[i] = dmvrPack( dmvrDot2( e[i], w, of1 ) >> sh1, dmvrDot2( __builtin_amdgcn_alignbit( e[i + 1], e[i], 16 ), w, of1 ) >> sh1 );
        o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
      }
      *reinterpret_cast<u32x4*>( &sT[wv][0][at[q]] ) = o;
    }
  }
  DMVR_SYNC();
  // ---- 2: second pass (fraction y != 0), the copy's precision shift when neither fraction is set; stored with the sign bit flipped (v_sad_u16 on biased values is exact)
  {
    const int perList = bh * segs, jobs = 2 * perList;
    for( int jb = lane; jb < jobs; jb += 64 )
    {
      const int list = jb >= perList, rem = jb - list * perList, y = segs == 3 ? rem / 3 : rem >> 1, x0 = ( rem - y * segs ) * 8;
      const int fx = ( list ? it.frac1_x : it.frac0_x ) & 15, fy = ( list ? it.frac1_y : it.frac0_y ) & 15;
      const u32x4 a = *reinterpret_cast<const u32x4*>( &sT[wv][list][y * DP + x0] );
      u32x4 o = a;
      if( fy )
      {
        const u32x4 b = *reinterpret_cast<const u32x4*>( &sT[wv][list][( y + 1 ) * DP + x0] );
        const int rnd = fx ? 8 : of1, sh = fx ? 4 : sh1;
        const uint32_t w = dmvrPack( 16 - fy, fy );
        const uint32_t aw[4] = { a.x, a.y, a.z, a.w }, bw_[4] = { b.x, b.y, b.z, b.w };
        uint32_t r[4];
#pragma unroll
        for( int i = 0; i < 4; i++ )
          r[i] = dmvrPack( dmvrDot2( __builtin_amdgcn_perm( bw_[i], aw[i], 0x05040100u ), w, rnd ) >> sh, dmvrDot2( __builtin_amdgcn_perm( bw_[i], aw[i], 0x07060302u ), w, rnd ) >> sh );
        o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
      }
      else if( !fx )
      {
        const int s = 10 - bitDepth;
        const uint32_t aw[4] = { a.x, a.y, a.z, a.w };
        uint32_t r[4];
#pragma unroll
        for( int i = 0; i < 4; i++ ) r[i] = dmvrPack( ( int ) ( int16_t ) ( aw[i] & 0xffffu ) << s, ( ( int ) aw[i] >> 16 ) << s );
        o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
      }
      o.x ^= 0x80008000u; o.y ^= 0x80008000u; o.z ^= 0x80008000u; o.w ^= 0x80008000u;
      *reinterpret_cast<u32x4*>( &sR[wv][list][y * DP + x0] ) = o;
    }
  }
  DMVR_SYNC();
  // ---- 3: mirrored SAD on every second row (subShift 1; (sum << 1) >> 1 == sum): position q = (ver + 2) * 5 + hor + 2; `rows` lanes per position, lane = row.
  //      Even horizontal offsets (hor = -2, 0, 2: 15 positions) start at a dword of both predictions (columns 2 + hor, 2 - hor); odd ones (10 positions) at an odd column of both
  {
    constexpr int rows = dy >> 1, dw = dx >> 1, PPW = 64 / rows;                  // positions per wave pass
    const int r = lane & ( rows - 1 ), slot = lane / rows;
#pragma unroll
    for( int pass = 0; pass < ( 15 + PPW - 1 ) / PPW; pass++ )                    // even offsets: idx = 0 .. 14 -> ver = idx / 3 - 2, hor = 2 ( idx % 3 ) - 2
    {
      const int idx = pass * PPW + slot;
      const bool valid = idx < 15;
      const int iv = valid ? idx : 7, v3 = ( iv * 11 ) >> 5, ver = v3 - 2, hor = 2 * ( iv - 3 * v3 ) - 2;      // ( i * 11 ) >> 5 == i / 3 for i < 16
      const uint32_t* p0 = reinterpret_cast<const uint32_t*>( &sR[wv][0][( 2 + 2 * r + ver ) * DP + 2 + hor] );
      const uint32_t* p1 = reinterpret_cast<const uint32_t*>( &sR[wv][1][( 2 + 2 * r - ver ) * DP + 2 - hor] );
      uint32_t sum = 0;
#pragma unroll
      for( int i = 0; i < dw; i++ ) sum = __builtin_amdgcn_sad_u16( p0[i], p1[i], sum );
      sum = vvhipGroupSum32( valid ? sum : 0u, rows, lane );
      if( valid && r == 0 ) sCost[wv][( ver + 2 ) * 5 + hor + 2] = sum;
    }
#pragma unroll
    for( int pass = 0; pass < ( 10 + PPW - 1 ) / PPW; pass++ )                    // odd offsets: idx = 0 .. 9 -> ver = idx / 2 - 2, hor = 2 ( idx % 2 ) - 1
    {
      const int idx = pass * PPW + slot;
      const bool valid = idx < 10;
      const int iv = valid ? idx : 4, ver = ( iv >> 1 ) - 2, hor = 2 * ( iv & 1 ) - 1;
      const uint32_t* p0 = reinterpret_cast<const uint32_t*>( &sR[wv][0][( 2 + 2 * r + ver ) * DP + 1 + hor] );      // column 2 + hor is odd: the dword below it
      const uint32_t* p1 = reinterpret_cast<const uint32_t*>( &sR[wv][1][( 2 + 2 * r - ver ) * DP + 1 - hor] );
      uint32_t sum = 0, x0 = p0[0], y0 = p1[0];
#pragma unroll
      for( int i = 0; i < dw; i++ )
      {
        const uint32_t x1 = p0[i + 1], y1 = p1[i + 1];
        sum = __builtin_amdgcn_sad_u16( __builtin_amdgcn_alignbit( x1, x0, 16 ), __builtin_amdgcn_alignbit( y1, y0, 16 ), sum );
        x0 = x1; y0 = y1;
      }
      sum = vvhipGroupSum32( valid ? sum : 0u, rows, lane );
      if( valid && r == 0 ) sCost[wv][( ver + 2 ) * 5 + hor + 2] = sum;
    }
  }
  DMVR_SYNC();

  {
    // (the 25 costs stay in LDS: the error surface indexes them dynamically — a private array would live in scratch memory)
    uint32_t* sad = sCost[wv];
    // centre: distFunc(SAD, subShift 1) >> 1, minus a quarter (:1332-1333); the X5 costs are SAD >> 1 without that reduction
    const uint32_t centre = sad[12] - ( sad[12] >> 2 );
    unsigned long long minCost = centre;
    int tx = 0, ty = 0;
    if( centre >= ( uint32_t ) ( dx * dy ) )                                        // (wave-uniform)
    {
      // the reference walks ver = -2 .. 2, hor = -2 .. 2 and takes a position only when it is STRICTLY cheaper than the best so far, starting from the centre: the winner is
      // the cheapest position, ties going to the centre, then to the first in raster order — the minimum of cost << 6 | ( centre ? 0 : q + 1 ) (costs < 2^18 at <= 10 bits)
      const uint32_t cq = lane == 12 ? centre : sad[lane < 25 ? lane : 12];
      const uint32_t key = lane < 25 ? ( cq << 6 | ( lane == 12 ? 0u : ( uint32_t ) lane + 1u ) ) : 0xffffffffu;
      const uint32_t best = dmvrGroupMin32( key, 64, lane );                        // (lanes 25 .. 63: all ones)
      const int qb = ( best & 63u ) ? ( int ) ( best & 63u ) - 1 : 12, bv_ = ( qb * 13 >> 6 ) - 2, bh_ = qb - 5 * ( qb * 13 >> 6 ) - 2;      // ( q * 13 ) >> 6 == q / 5 for q < 25
      minCost = best >> 6;
      DMVR_SYNC();
      if( lane == 0 ) sad[12] = centre;
      DMVR_SYNC();
      tx = bh_ * 16; ty = bv_ * 16;
      if( bh_ != 2 && bh_ != -2 && bv_ != 2 && bv_ != -2 )                       // xDMVRSubPixelErrorSurface (:1230-1231)
      {
        const uint32_t* p = &sad[12 + bv_ * 5 + bh_];
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
    if( lane == 0 )
    {
      vvhip_dmvr_result r; r.mvd_x = ( int16_t ) tx; r.mvd_y = ( int16_t ) ty; r.pad = 0; r.min_cost = minCost;
      out[blk] = r;
    }
  }
#undef DMVR_SYNC
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
  const dim3 grid( ( n + 3 ) / 4 ), block( 256 );
  if( dx == 16 && dy == 16 )     hipLaunchKernelGGL( ( dmvrRefineKernel<16, 16> ), grid, block, 0, ctx->stream, d_ref0, stride0, d_ref1, stride1, d_items, n, bit_depth, d_out );
  else if( dx == 16 )            hipLaunchKernelGGL( ( dmvrRefineKernel<16, 8> ), grid, block, 0, ctx->stream, d_ref0, stride0, d_ref1, stride1, d_items, n, bit_depth, d_out );
  else if( dy == 16 )            hipLaunchKernelGGL( ( dmvrRefineKernel<8, 16> ), grid, block, 0, ctx->stream, d_ref0, stride0, d_ref1, stride1, d_items, n, bit_depth, d_out );
  else                           hipLaunchKernelGGL( ( dmvrRefineKernel<8, 8> ), grid, block, 0, ctx->stream, d_ref0, stride0, d_ref1, stride1, d_items, n, bit_depth, d_out );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

} // extern "C"
