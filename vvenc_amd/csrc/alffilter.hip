// alffilter.hip — SURVEY §8f rank 4, second half: applying the ALF and CC-ALF filters to a picture (the data-parallel stage that follows the encoder's filter derivation).
// Compiled with -fno-slp-vectorize (see Makefile): the SLP vectoriser turns the register window of alfFilterKernel back into a private array (scratch).
#include "common.h"

namespace {

// ---- ALF filtering: AdaptiveLoopFilter::filterBlk<ALF_FILTER_7 / ALF_FILTER_5> (CommonLib/AdaptiveLoopFilter.cpp:730-967; table entries m_filter7x7Blk / m_filter5x5Blk,
// :76-79), for every enabled CTU of a plane as EncAdaptiveLoopFilter::reconstructCTU drives it without virtual picture boundaries (EncoderLib/EncAdaptiveLoopFilter.cpp:2035-2066).
// One lane per 4x4 block (one class / transpose index / CTU per lane).  A workgroup stages a 256x16 tile plus its halo in LDS with plain 16-bit loads (no alignment or
// stride constraint on the caller's plane), every lane then holds its 12-sample-wide window in registers; block rows next to the virtual boundary (wave-uniform: a wave
// covers one block row) take the fold-back of the rows from LDS instead.
struct AlfFilterArgs
{
  const int16_t* src; int16_t* dst; const uint8_t* cls; const int16_t* coeff; const int16_t* clip; const int16_t* ctuSet;
  ptrdiff_t srcStride, dstStride;
  int width, height, ctuSize, ctusX, maxVal, vbH, vbPos, numClasses;
};

__host__ __device__ constexpr int alfDy( int fl, int k ) { return fl == 7 ? ( k < 1 ? 3 : k < 4 ? 2 : k < 9 ? 1 : 0 ) : ( k < 1 ? 2 : k < 4 ? 1 : 0 ); }
__host__ __device__ constexpr int alfDx( int fl, int k ) { return fl == 7 ? ( k == 0 ? 0 : k < 4 ? 2 - k : k < 9 ? 6 - k : 12 - k ) : ( k == 0 ? 0 : k < 4 ? 2 - k : 6 - k ); }
// coefficient order per transposeIdx (:805-849), one nibble per tap
__host__ __device__ constexpr uint64_t alfNib( int a0, int a1, int a2, int a3, int a4, int a5, int a6 = 0, int a7 = 0, int a8 = 0, int a9 = 0, int a10 = 0, int a11 = 0 )
{
  return ( uint64_t ) a0 | ( uint64_t ) a1 << 4 | ( uint64_t ) a2 << 8 | ( uint64_t ) a3 << 12 | ( uint64_t ) a4 << 16 | ( uint64_t ) a5 << 20 | ( uint64_t ) a6 << 24 | ( uint64_t ) a7 << 28 |
         ( uint64_t ) a8 << 32 | ( uint64_t ) a9 << 36 | ( uint64_t ) a10 << 40 | ( uint64_t ) a11 << 44;
}
__device__ __forceinline__ int alfMed3( int x, int lo, int hi ) { int r; asm( "v_med3_i32 %0, %1, %2, %3" : "=v"( r ) : "v"( x ), "v"( lo ), "v"( hi ) ); return r; }

constexpr int ALF_FT_W = 256, ALF_FT_H = 16, ALF_FT_P = ALF_FT_W + 8;              // tile of a workgroup; LDS row: x0-4 .. x0+259

template<int FL, bool NONLIN>
__global__ void __launch_bounds__( 256 )
alfFilterKernel( const AlfFilterArgs A )
{
  constexpr int NT = FL == 7 ? 12 : 6, R = FL / 2, ROWS = ALF_FT_H + 2 * R, WR = 4 + 2 * R;
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sT[ROWS][ALF_FT_P];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int x0 = blockIdx.x * ALF_FT_W, y0 = blockIdx.y * ALF_FT_H;
  // stage the tile: row rr <-> y0 - R + rr (<= height + R - 1), column c <-> x0 - 4 + c (<= width + 3): the plane's replicated border covers both
  for( int rr = wv; rr < ROWS; rr += 4 )
  {
    const int ys = min( y0 - R + rr, A.height + R - 1 );
    const int16_t* row = A.src + ( ptrdiff_t ) ys * A.srcStride;
#pragma unroll
    for( int i = 0; i < 5; i++ )
    {
      const int c = lane + 64 * i;
      if( c < ALF_FT_P ) sT[rr][c] = row[min( x0 - 4 + c, A.width + 3 )];
    }
  }
  __syncthreads();
  const int x = x0 + 4 * lane, y = y0 + 4 * wv;
  if( x >= A.width || y >= A.height ) return;
  const int set = A.ctuSet[( y / A.ctuSize ) * A.ctusX + x / A.ctuSize];
  if( set < 0 ) return;                                                              // m_ctuEnableFlag off: the CTU keeps its samples
  int classIdx = 0, tr = 0;
  if( A.cls ) { const uint8_t* c = A.cls + 2 * ( ( size_t ) ( y >> 2 ) * ( A.width >> 2 ) + ( x >> 2 ) ); classIdx = c[0]; tr = c[1]; }
  constexpr uint64_t P7[4] = { alfNib( 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 ), alfNib( 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6 ), alfNib( 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11 ), alfNib( 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6 ) };
  constexpr uint64_t P5[4] = { alfNib( 0, 1, 2, 3, 4, 5 ), alfNib( 4, 1, 5, 3, 0, 2 ), alfNib( 0, 3, 2, 1, 4, 5 ), alfNib( 4, 3, 5, 1, 0, 2 ) };
  const uint64_t pm = FL == 7 ? ( tr == 0 ? P7[0] : tr == 1 ? P7[1] : tr == 2 ? P7[2] : P7[3] ) : ( tr == 0 ? P5[0] : tr == 1 ? P5[1] : tr == 2 ? P5[2] : P5[3] );
  const int16_t* cfp = A.coeff + ( ( size_t ) set * A.numClasses + classIdx ) * 13;
  const int16_t* clp = NONLIN ? A.clip + ( ( size_t ) set * A.numClasses + classIdx ) * 13 : nullptr;
  int cf[NT], cl[NT], cfSum = 0;
#pragma unroll
  for( int k = 0; k < NT; k++ )
  {
    const int q = ( int ) ( pm >> ( 4 * k ) ) & 15;
    cf[k] = cfp[q]; cfSum += cf[k];
    cl[k] = NONLIN ? clp[q] : 0;
  }
  // window: rows y - R .. y + 3 + R, columns x - 4 .. x + 7 (dword d of a row = columns x - 4 + 2d, x - 3 + 2d)
  uint32_t w[WR][6];
#pragma unroll
  for( int rr = 0; rr < WR; rr++ )
  {
    const uint2* q = ( const uint2* ) &sT[4 * wv + rr][4 * lane];
    const uint2 a = q[0], b = q[1], c = q[2];
    w[rr][0] = a.x; w[rr][1] = a.y; w[rr][2] = b.x; w[rr][3] = b.y; w[rr][4] = c.x; w[rr][5] = c.y;
  }
#define ALF_S( rr, c ) ( int ) ( ( ( c ) & 1 ) ? w[rr][( ( c ) + 4 ) >> 1] >> 16 : w[rr][( ( c ) + 4 ) >> 1] & 0xffffu )
  const bool dstWide = ( ( ( uintptr_t ) A.dst | ( uintptr_t ) ( A.dstStride * 2 ) ) & 7 ) == 0;
#pragma unroll
  for( int r = 0; r < 4; r++ )
  {
    const int yVb = ( y + r ) & ( A.vbH - 1 );
    const int dist = yVb < A.vbPos ? A.vbPos - 1 - yVb : yVb - A.vbPos;            // rows the filter may reach on either side of the virtual boundary (:881-897)
    int o[4];
    if( dist >= R )                                                                  // wave-uniform
    {
#pragma unroll
      for( int j = 0; j < 4; j++ )
      {
        const int cur = ALF_S( r + R, j );
        int sum = 0;
#pragma unroll
        for( int k = 0; k < NT; k++ )
        {
          const int dy = alfDy( FL, k ), dx = alfDx( FL, k );
          const int a = ALF_S( r + R + dy, j + dx ), b = ALF_S( r + R - dy, j - dx );
          if( NONLIN ) sum += cf[k] * ( alfMed3( a - cur, -cl[k], cl[k] ) + alfMed3( b - cur, -cl[k], cl[k] ) );
          else         sum += cf[k] * ( a + b );
        }
        if( !NONLIN ) sum -= 2 * cur * cfSum;
        o[j] = min( max( cur + ( ( sum + 64 ) >> 7 ), 0 ), A.maxVal );
      }
    }
    else
    {
      const int sh = dist == 0 ? 10 : 7;                                             // rows next to the boundary: three more bits (:940-947)
#pragma unroll
      for( int j = 0; j < 4; j++ )
      {
        const int16_t* p = &sT[4 * wv + r + R][4 * lane + 4 + j];
        const int cur = p[0];
        int sum = 0;
#pragma unroll
        for( int k = 0; k < NT; k++ )
        {
          const int dy = min( alfDy( FL, k ), dist ), dx = alfDx( FL, k );
          const int a = p[dy * ALF_FT_P + dx] - cur, b = p[-dy * ALF_FT_P - dx] - cur;
          if( NONLIN ) sum += cf[k] * ( alfMed3( a, -cl[k], cl[k] ) + alfMed3( b, -cl[k], cl[k] ) );
          else         sum += cf[k] * ( a + b );
        }
        o[j] = min( max( cur + ( ( sum + ( 1 << ( sh - 1 ) ) ) >> sh ), 0 ), A.maxVal );
      }
    }
    int16_t* d = A.dst + ( ptrdiff_t ) ( y + r ) * A.dstStride + x;
    if( dstWide ) *( uint2* ) d = make_uint2( ( uint32_t ) o[0] | ( uint32_t ) o[1] << 16, ( uint32_t ) o[2] | ( uint32_t ) o[3] << 16 );
    else { d[0] = ( int16_t ) o[0]; d[1] = ( int16_t ) o[1]; d[2] = ( int16_t ) o[2]; d[3] = ( int16_t ) o[3]; }
  }
#undef ALF_S
}

// ---- CC-ALF filtering: AdaptiveLoopFilter::filterBlkCcAlf (CommonLib/AdaptiveLoopFilter.cpp:969-1058; table entry m_filterCcAlf, :74) over the CTUs of a chroma plane as
// EncAdaptiveLoopFilter::applyCcAlfFilterCTU drives it (EncoderLib/EncAdaptiveLoopFilter.cpp:6606-6699).  One lane per chroma sample: 8 luma samples, 7 coefficients.
struct CcAlfFilterArgs
{
  int16_t* dst; const int16_t* luma; const int16_t* coeff; const uint8_t* ctuFilter;
  ptrdiff_t dstStride, lumaStride;
  int widthC, heightC, ctuSizeC, ctusX, sx, sy, maxVal, half, vbH, vbPos;
};

__global__ void __launch_bounds__( 256 )
ccAlfFilterKernel( const CcAlfFilterArgs A )
{
  const int x = blockIdx.x * 64 + ( threadIdx.x & 63 ), y = blockIdx.y * 4 + ( threadIdx.x >> 6 );
  if( x >= A.widthC || y >= A.heightC ) return;
  const int f = A.ctuFilter[( y / A.ctuSizeC ) * A.ctusX + x / A.ctuSizeC];
  if( !f ) return;
  const int pos = ( y << A.sy ) & ( A.vbH - 1 );
  if( A.sy == 0 && ( pos == A.vbPos || pos == A.vbPos + 1 ) ) return;              // :1016-1019
  ptrdiff_t o1 = A.lumaStride, o2 = -A.lumaStride, o3 = 2 * A.lumaStride;
  if( pos == A.vbPos - 2 || pos == A.vbPos + 1 ) o3 = o1;                          // :1020-1029
  else if( pos == A.vbPos - 1 || pos == A.vbPos ) o1 = o2 = o3 = 0;
  const int16_t* cf = A.coeff + ( f - 1 ) * 8;
  const int16_t* l = A.luma + ( ptrdiff_t ) ( y << A.sy ) * A.lumaStride + ( x << A.sx );
  const int c = l[0];
  int sum = cf[0] * ( l[o2] - c ) + cf[1] * ( l[-1] - c ) + cf[2] * ( l[1] - c ) + cf[3] * ( l[o1 - 1] - c ) + cf[4] * ( l[o1] - c ) + cf[5] * ( l[o1 + 1] - c ) + cf[6] * ( l[o3] - c );
  sum = ( sum + 64 ) >> 7;                                                         // m_scaleBits 7
  sum = min( max( sum + A.half, 0 ), A.maxVal ) - A.half;
  int16_t* d = A.dst + ( ptrdiff_t ) y * A.dstStride + x;
  *d = ( int16_t ) min( max( sum + *d, 0 ), A.maxVal );
}

} // namespace

extern "C" {

int vvhip_alf_filter_plane( vvhip_ctx* ctx, const int16_t* d_src, ptrdiff_t src_stride, int16_t* d_dst, ptrdiff_t dst_stride, int width, int height, int ctu_size, int bit_depth,
                            int filter_length, const uint8_t* d_cls, const int16_t* d_coeff, const int16_t* d_clip, const int16_t* d_ctu_set, int vb_ctu_height, int vb_pos )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width < 4 || height < 4 || ( width & 3 ) || ( height & 3 ) || ( filter_length != 7 && filter_length != 5 ) || ctu_size < 8 || ( ctu_size & 3 ) || bit_depth < 8 || bit_depth > 12 ||
      vb_ctu_height < 4 || ( vb_ctu_height & ( vb_ctu_height - 1 ) ) || vb_pos < 0 || !d_src || !d_dst || !d_coeff || !d_ctu_set || ( filter_length == 5 && d_cls ) || ( filter_length == 7 && !d_cls ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_alf_filter_plane: %dx%d (multiples of 4), CTU %d, filter length %d (7 with classes / 5 without), bitDepth %d", width, height, ctu_size, filter_length, bit_depth );
  AlfFilterArgs A;
  A.src = d_src; A.dst = d_dst; A.cls = d_cls; A.coeff = d_coeff; A.clip = d_clip; A.ctuSet = d_ctu_set; A.srcStride = src_stride; A.dstStride = dst_stride;
  A.width = width; A.height = height; A.ctuSize = ctu_size; A.ctusX = ( width + ctu_size - 1 ) / ctu_size; A.maxVal = ( 1 << bit_depth ) - 1; A.vbH = vb_ctu_height; A.vbPos = vb_pos;
  A.numClasses = d_cls ? 25 : 1;
  const dim3 grid( ( width + ALF_FT_W - 1 ) / ALF_FT_W, ( height + ALF_FT_H - 1 ) / ALF_FT_H ), block( 256 );
  if( filter_length == 7 ) { if( d_clip ) hipLaunchKernelGGL( ( alfFilterKernel<7, true> ), grid, block, 0, ctx->stream, A ); else hipLaunchKernelGGL( ( alfFilterKernel<7, false> ), grid, block, 0, ctx->stream, A ); }
  else                     { if( d_clip ) hipLaunchKernelGGL( ( alfFilterKernel<5, true> ), grid, block, 0, ctx->stream, A ); else hipLaunchKernelGGL( ( alfFilterKernel<5, false> ), grid, block, 0, ctx->stream, A ); }
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_ccalf_filter_plane( vvhip_ctx* ctx, int16_t* d_dst_c, ptrdiff_t dst_stride, const int16_t* d_rec_luma, ptrdiff_t rec_stride, int width_c, int height_c, int ctu_size_c,
                              int shift_x, int shift_y, int bit_depth, const int16_t* d_coeff, const uint8_t* d_ctu_filter, int vb_ctu_height, int vb_pos )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width_c < 1 || height_c < 1 || ctu_size_c < 4 || shift_x < 0 || shift_x > 1 || shift_y < 0 || shift_y > 1 || bit_depth < 8 || bit_depth > 12 ||
      vb_ctu_height < 4 || ( vb_ctu_height & ( vb_ctu_height - 1 ) ) || !d_dst_c || !d_rec_luma || !d_coeff || !d_ctu_filter )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_ccalf_filter_plane: chroma %dx%d, chroma CTU %d, shifts %d/%d, bitDepth %d", width_c, height_c, ctu_size_c, shift_x, shift_y, bit_depth );
  CcAlfFilterArgs A;
  A.dst = d_dst_c; A.luma = d_rec_luma; A.coeff = d_coeff; A.ctuFilter = d_ctu_filter; A.dstStride = dst_stride; A.lumaStride = rec_stride; A.widthC = width_c; A.heightC = height_c;
  A.ctuSizeC = ctu_size_c; A.ctusX = ( width_c + ctu_size_c - 1 ) / ctu_size_c; A.sx = shift_x; A.sy = shift_y; A.maxVal = ( 1 << bit_depth ) - 1; A.half = ( 1 << bit_depth ) >> 1;
  A.vbH = vb_ctu_height; A.vbPos = vb_pos;
  hipLaunchKernelGGL( ccAlfFilterKernel, dim3( ( width_c + 63 ) / 64, ( height_c + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, A );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

} // extern "C"
