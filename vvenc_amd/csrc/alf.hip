// alf.hip — SURVEY §8f rank 4: ALF encoder statistics (the largest stage outside the north-star path: ≈19.5 % of single-thread time at preset faster).
//
// Reference behaviour:
//   classification   AdaptiveLoopFilter::deriveClassificationBlk          CommonLib/AdaptiveLoopFilter.cpp:524-728  (table entry m_deriveClassificationBlk, :73)
//   local terms      EncAdaptiveLoopFilter::calcLinCovariance4            EncoderLib/EncAdaptiveLoopFilter.cpp:3544-3921 (linear filters: numBins 1)
//   accumulation     EncAdaptiveLoopFilter::getPreBlkStats :3376-3541  +  m_getPreBlkStatsAccum (scalar :3266-3319, x86 x86/EncAdaptiveLoopFilterX86.h:160-236)
// Classification is integer and independent per 4x4 block: one lane per block.
// The covariance terms are FLOAT sums of per-block int32 sums, so the order of the additions is part of the result: the reference walks the
// 4x4 blocks of a CTU in raster order and adds each block's term to the entry of the block's class.  One workgroup per CTU keeps that
// order: per block row the local terms of all blocks are built in LDS by all threads, then every (entry) lane pair walks the blocks in
// order — integer dot product of two 16-sample rows, v_cvt_f32_i32, one float add into the register of the block's class (25 registers per
// lane; the class is wave-uniform).  IEEE round-to-nearest on both sides: results are bit-identical with the reference.
#include "common.h"

namespace {

constexpr int ALF_REC = 13 * 13 + 13 + 1;          // floats per (CTU, class): E[13][13] row-major, y[13], pixAcc
constexpr int ALF_WIN_P = 128 + 8;                 // staged rec row: 3 samples left / right of the CTU (+2 alignment)
constexpr int ALF_MAXB = 32;                       // 4x4 blocks per CTU row (CTU <= 128)
constexpr int ALF_ROWS = 14;                       // 13 local-term rows + the org-rec row
constexpr int ALF_NE = 13 * 14 / 2 + 13 + 1;       // entries of a record that are accumulated: E upper triangle, y, pixAcc

struct AlfTap { int8_t i, j; };
struct AlfTaps { AlfTap t[2][4][12]; };            // [shape 0: 7x7, 1: 5x5][transposeIdx][k]: tap A = (off0(i), +j), tap B = (off1(i), -j)

__global__ void __launch_bounds__( 256 )
alfClassifyKernel( const int16_t* __restrict__ rec, int stride, int width, int height, int shift, int vbH, int vbPos, uint8_t* __restrict__ cls )
{
  const int bw = width >> 2, nb = bw * ( height >> 2 );
  const int b = blockIdx.x * 256 + threadIdx.x;
  if( b >= nb ) return;
  const int Y = ( b / bw ) * 4, X = ( b - ( b / bw ) * bw ) * 4;
  const int ym = Y % vbH;
  const int r0 = ym == vbPos ? 1 : 0, r1 = ym == vbPos - 4 ? 3 : 4;                            // rows of 2x2 positions that count (:630-650)
  int sumV = 0, sumH = 0, sumD0 = 0, sumD1 = 0;
  for( int r = r0; r < r1; r++ )
  {
    const int y = Y - 2 + 2 * r;
    const int16_t* s1 = rec + ( ptrdiff_t ) y * stride + X - 3;
    const int16_t* s0 = s1 - stride; const int16_t* s2 = s1 + stride; const int16_t* s3 = s1 + 2 * stride;
    if( y > 0 && ( y & ( vbH - 1 ) ) == vbPos - 2 ) s3 = s2;                                   // :559-566
    else if( y > 0 && ( y & ( vbH - 1 ) ) == vbPos ) s0 = s1;
    int a0[10], a1[10], a2[10], a3[10];                                                        // columns X-3 .. X+6
#pragma unroll
    for( int c = 0; c < 10; c++ ) { a0[c] = s0[c]; a1[c] = s1[c]; a2[c] = s2[c]; a3[c] = s3[c]; }
#pragma unroll
    for( int c = 0; c < 4; c++ )
    {
      const int x = 1 + 2 * c;                                                                 // index of sample X-2+2c in the arrays
      const int y0 = ( int16_t ) ( a1[x] << 1 ), yup1 = ( int16_t ) ( a2[x + 1] << 1 );
      sumV  += abs( y0 - a0[x] - a2[x] )         + abs( yup1 - a1[x + 1] - a3[x + 1] );        // :583-586
      sumH  += abs( y0 - a1[x + 1] - a1[x - 1] ) + abs( yup1 - a2[x + 2] - a2[x] );
      sumD0 += abs( y0 - a0[x - 1] - a2[x + 1] ) + abs( yup1 - a1[x] - a3[x + 2] );
      sumD1 += abs( y0 - a2[x - 1] - a0[x + 1] ) + abs( yup1 - a3[x] - a1[x + 2] );
    }
  }
  const int tempAct = sumV + sumH;
  const int yb = Y & ( vbH - 1 );
  int activity = ( tempAct * ( ( yb == vbPos - 4 || yb == vbPos ) ? 96 : 64 ) ) >> shift;      // :655-663
  activity = activity < 0 ? 0 : ( activity > 15 ? 15 : activity );
  const int th = activity == 0 ? 0 : activity == 1 ? 1 : activity < 7 ? 2 : activity < 15 ? 3 : 4;     // th[] :530
  int classIdx = th;
  int hv1, hv0, d1, d0, hvd1, hvd0, dirHV, dirD, mainDir, secDir;
  if( sumV > sumH ) { hv1 = sumV; hv0 = sumH; dirHV = 1; } else { hv1 = sumH; hv0 = sumV; dirHV = 3; }
  if( sumD0 > sumD1 ) { d1 = sumD0; d0 = sumD1; dirD = 0; } else { d1 = sumD1; d0 = sumD0; dirD = 2; }
  if( ( uint32_t ) d1 * ( uint32_t ) hv0 > ( uint32_t ) hv1 * ( uint32_t ) d0 ) { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
  else { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
  int strength = 0;
  if( hvd1 > 2 * hvd0 ) strength = 1;
  if( hvd1 * 2 > 9 * hvd0 ) strength = 2;
  if( strength ) classIdx += ( ( ( mainDir & 1 ) << 1 ) + strength ) * 5;
  const int tIdx = mainDir * 2 + ( secDir >> 1 );                                              // transposeTable { 0, 1, 0, 2, 2, 3, 1, 3 } :722
  const int transposeIdx = tIdx == 0 ? 0 : tIdx == 1 ? 1 : tIdx == 2 ? 0 : tIdx == 3 ? 2 : tIdx == 4 ? 2 : tIdx == 5 ? 3 : tIdx == 6 ? 1 : 3;
  cls[2 * b] = ( uint8_t ) classIdx; cls[2 * b + 1] = ( uint8_t ) transposeIdx;
}

struct AlfStatArgs
{
  const int16_t* org; const int16_t* rec; const uint8_t* cls; int32_t* sums; const float* init; float* out;
  const int16_t* slf; int slfStride, sx, sy, picHeight;      // CC-ALF: ALF-filtered chroma, chroma subsampling shifts, luma picture height
  int subBlk;                                                // traversal sub-unit (CTU inside a statistics unit) in 4x4 blocks per side
  int orgStride, recStride, width, height, ctuSize, ctusX, nc, shape, vbH, vbPos, blocksPerCtuRow;
};

__device__ __forceinline__ void alfEntryRows( int e, int nc, int& ra, int& rb )      // entry -> the two local-term rows whose dot product it is
{
  const int nTri = nc * ( nc + 1 ) / 2;
  if( e < nTri ) { int k = 0, rem = e; while( rem >= nc - k ) { rem -= nc - k; k++; } ra = k; rb = k + rem; }
  else if( e < nTri + nc ) { ra = e - nTri; rb = 13; }
  else { ra = 13; rb = 13; }
}

// Kernel A — everything that does not depend on the order: one workgroup per (CTU, block row): the rec window and org - rec of the row go to
// LDS, all threads build the local terms of the row's 4x4 blocks (calcLinCovariance4) and the int32 dot products of every (block, entry)
// pair -> sums[ctu][block][entry] (exact integers).
// MODE 0: ALF (rec = the plane itself), MODE 1: CC-ALF (rec = luma, org / slf = chroma; getBlkStatsCcAlf :6061-6357, calcCovariance4CcAlf :6359-6422)
template<int MODE>
__global__ void __launch_bounds__( 256 )
alfBlockSumsKernel( AlfStatArgs A, AlfTaps T )
{
  __shared__ int16_t sRec[10][ALF_WIN_P];                    // rows y-3 .. y+6 of the block row, columns x0-3 ..
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sLoc[ALF_MAXB][ALF_ROWS][16];
  __shared__ uint8_t sTr[ALF_MAXB];
  __shared__ uint8_t sPair[ALF_NE][2];
  __shared__ AlfTap sTap[4][12];                             // taps of this shape: kernel arguments are not indexable per lane without a memory load per item
  const int tid = threadIdx.x;
  const int ctu = blockIdx.x, cx = ctu % A.ctusX, cy = ctu / A.ctusX;
  const int x0 = cx * A.ctuSize, y0 = cy * A.ctuSize, i = blockIdx.y * 4;
  const int w = min( A.ctuSize, A.width - x0 ), h = min( A.ctuSize, A.height - y0 );
  if( i >= h ) return;
  const int nb = w >> 2, nc = A.nc;
  const int nE = nc * ( nc + 1 ) / 2 + nc + 1;
  if( tid < nE ) { int ra, rb; alfEntryRows( tid, nc, ra, rb ); sPair[tid][0] = ( uint8_t ) ra; sPair[tid][1] = ( uint8_t ) rb; }
  if( tid >= 192 && tid < 240 ) sTap[( tid - 192 ) / 12][( tid - 192 ) % 12] = T.t[A.shape][( tid - 192 ) / 12][( tid - 192 ) % 12];
  if( MODE == 0 )
    for( int t = tid; t < 10 * ( ( w + 6 + 1 ) >> 1 ); t += 256 )
    {
      const int pw = ( w + 6 + 1 ) >> 1, r = t / pw, c = 2 * ( t - r * pw );
      const int16_t* src = A.rec + ( ptrdiff_t ) ( y0 + i - 3 + r ) * A.recStride + x0 - 3 + c;
      sRec[r][c] = src[0]; sRec[r][c + 1] = src[1];                                             // (the margin of >= 4 covers the odd last column)
    }
  else
  {
    // luma rows (y << sy) - 1 .. ((y + 3) << sy) + 2, columns (x0 << sx) - 1 .. ((x0 + w - 1) << sx) + 1
    const int nr = ( 3 << A.sy ) + 4, ncol = ( ( w - 1 ) << A.sx ) + 3, pw = ( ncol + 1 ) >> 1;
    for( int t = tid; t < nr * pw; t += 256 )
    {
      const int r = t / pw, c = 2 * ( t - r * pw );
      const int16_t* src = A.rec + ( ptrdiff_t ) ( ( ( y0 + i ) << A.sy ) - 1 + r ) * A.recStride + ( x0 << A.sx ) - 1 + c;
      sRec[r][c] = src[0]; sRec[r][c + 1] = src[1];
    }
  }
  if( tid < nb ) sTr[tid] = A.cls ? A.cls[2 * ( ( size_t ) ( ( y0 + i ) >> 2 ) * ( A.width >> 2 ) + ( x0 >> 2 ) + tid ) + 1] & 3 : 0;
  __syncthreads();
  // local terms (:3423-3457, :3707-3921).  ALF: a lane pair per (block, sample row) — the centre samples, the virtual-boundary reach and the
  // block's tap table are set up once, lane 0 / 1 of the pair take the even / odd term rows (row 13 = org - rec).
  if( MODE == 0 )
  {
    const int item = tid >> 1, half = tid & 1;
    if( item < nb * 4 )
    {
      const int b = item >> 2, ii = item & 3;
      const int16_t* c0 = &sRec[3 + ii][3 + 4 * b];
      const int cen[4] = { c0[0], c0[1], c0[2], c0[3] };
      const int vbd = ( ( y0 + i + ii ) & ( A.vbH - 1 ) ) - A.vbPos;        // vertical reach against the virtual boundary (:3394-3411)
      int clipTop = -4, clipBot = 4;
      if( vbd >= -3 && vbd < 0 ) { clipBot = -vbd - 1; clipTop = -clipBot; }
      else if( vbd >= 0 && vbd < 3 ) { clipTop = -vbd; clipBot = -clipTop; }
      const AlfTap* taps = sTap[sTr[b]];
      for( int k = half; k < ALF_ROWS; k += 2 )
      {
        int v[4];
        if( k == 13 )
        {
          const int16_t* o = A.org + ( ptrdiff_t ) ( y0 + i + ii ) * A.orgStride + x0 + 4 * b;
#pragma unroll
          for( int x = 0; x < 4; x++ ) v[x] = ( int16_t ) ( o[x] - cen[x] );
        }
        else if( k == nc - 1 )
        {
#pragma unroll
          for( int x = 0; x < 4; x++ ) v[x] = cen[x];
        }
        else if( k < nc - 1 )
        {
          const AlfTap tp = taps[k];
          int o0 = tp.i, o1 = -( int ) tp.i;
          if( clipBot != 4 && tp.i != 0 ) { o0 = max( ( int ) tp.i, clipTop ); o1 = -max( ( int ) tp.i, -clipBot ); }     // the clipped form only when clipBotRow != 4 (:3438)
          const int16_t* pa = c0 + o0 * ALF_WIN_P + tp.j;
          const int16_t* pb = c0 + o1 * ALF_WIN_P - tp.j;
#pragma unroll
          for( int x = 0; x < 4; x++ ) v[x] = ( int16_t ) ( pa[x] + pb[x] - ( int16_t ) ( cen[x] << 1 ) );
        }
        else continue;
        uint32_t* dst = reinterpret_cast<uint32_t*>( &sLoc[b][k][ii * 4] );
        dst[0] = ( uint32_t ) ( v[0] & 0xffff ) | ( ( uint32_t ) v[1] << 16 );
        dst[1] = ( uint32_t ) ( v[2] & 0xffff ) | ( ( uint32_t ) v[3] << 16 );
      }
    }
  }
  else
  for( int t = tid; t < nb * ALF_ROWS * 4; t += 256 )
  {
    const int b = t / ( ALF_ROWS * 4 ), rem = t - b * ( ALF_ROWS * 4 ), k = rem >> 2, ii = rem & 3;
    int16_t* dst = &sLoc[b][k][ii * 4];
    if( MODE == 1 )
    {
      if( k == 13 )
      {
        const int16_t* o = A.org + ( ptrdiff_t ) ( y0 + i + ii ) * A.orgStride + x0 + 4 * b;
        const int16_t* f = A.slf + ( ptrdiff_t ) ( y0 + i + ii ) * A.slfStride + x0 + 4 * b;
#pragma unroll
        for( int x = 0; x < 4; x++ ) dst[x] = ( int16_t ) ( o[x] - f[x] );
      }
      else if( k < 7 )
      {
        // rows of the luma window: centre row r0, above / below / two below, folded at the virtual boundary (:6368-6376; no boundary in the last CTU row :6079)
        const int vbPos = ( ( y0 << A.sy ) + ( A.ctuSize << A.sy ) ) >= A.picHeight ? A.picHeight : A.vbPos;
        const int vbd = ( ( ( i + ii ) << A.sy ) & ( A.vbH - 1 ) ) - vbPos;
        const int r0 = 1 + ( ii << A.sy );
        int rm1 = r0 - 1, rp1 = r0 + 1, rp2 = r0 + 2;
        if( vbd == -2 || vbd == 1 ) rp2 = rp1;
        else if( vbd == -1 || vbd == 0 ) { rm1 = r0; rp1 = r0; rp2 = r0; }
        const int row = k == 0 ? rm1 : k < 3 ? r0 : k < 6 ? rp1 : rp2;
        const int dx = ( k == 1 || k == 3 ) ? -1 : ( k == 2 || k == 5 ) ? 1 : 0;
#pragma unroll
        for( int x = 0; x < 4; x++ )
        {
          const int col = 1 + ( ( 4 * b + x ) << A.sx );
          dst[x] = ( int16_t ) ( sRec[row][col + dx] - sRec[r0][col] );
        }
      }
      continue;
    }
    const int16_t* c0 = &sRec[3 + ii][3 + 4 * b];
    if( k == 13 )
    {
      const int16_t* o = A.org + ( ptrdiff_t ) ( y0 + i + ii ) * A.orgStride + x0 + 4 * b;
#pragma unroll
      for( int x = 0; x < 4; x++ ) dst[x] = ( int16_t ) ( o[x] - c0[x] );
    }
    else if( k == nc - 1 )
    {
#pragma unroll
      for( int x = 0; x < 4; x++ ) dst[x] = c0[x];
    }
    else if( k < nc - 1 )
    {
      // vertical reach of this sample row against the virtual boundary (:3394-3411); the clipped form only when clipBotRow != 4 (:3438)
      const int vbd = ( ( y0 + i + ii ) & ( A.vbH - 1 ) ) - A.vbPos;
      int clipTop = -4, clipBot = 4;
      if( vbd >= -3 && vbd < 0 ) { clipBot = -vbd - 1; clipTop = -clipBot; }
      else if( vbd >= 0 && vbd < 3 ) { clipTop = -vbd; clipBot = -clipTop; }
      const AlfTap tp = sTap[sTr[b]][k];
      int o0 = tp.i, o1 = -( int ) tp.i;
      if( clipBot != 4 && tp.i != 0 ) { o0 = max( ( int ) tp.i, clipTop ); o1 = -max( ( int ) tp.i, -clipBot ); }
      const int16_t* pa = c0 + o0 * ALF_WIN_P + tp.j;
      const int16_t* pb = c0 + o1 * ALF_WIN_P - tp.j;
#pragma unroll
      for( int x = 0; x < 4; x++ ) dst[x] = ( int16_t ) ( pa[x] + pb[x] - ( int16_t ) ( c0[x] << 1 ) );
    }
  }
  __syncthreads();
  // layout [ctu][block row][entry][32 blocks]: kernel B's lane (= entry) reads one block row as 128 contiguous bytes
  int32_t* sums = A.sums + ( ( size_t ) ctu * A.blocksPerCtuRow + blockIdx.y ) * ALF_NE * ALF_MAXB;
  for( int t = tid; t < nb * nE; t += 256 )
  {
    const int e = t / nb, b = t - e * nb;
    const int4* pa = reinterpret_cast<const int4*>( &sLoc[b][sPair[e][0]][0] );
    const int4* pb = reinterpret_cast<const int4*>( &sLoc[b][sPair[e][1]][0] );
    typedef short s2 __attribute__( ( ext_vector_type( 2 ) ) );
    int s = 0;
#pragma unroll
    for( int q = 0; q < 2; q++ )
    {
      const int4 va = pa[q], vb = pb[q];
      s = __builtin_amdgcn_sdot2( __builtin_bit_cast( s2, va.x ), __builtin_bit_cast( s2, vb.x ), s, false );
      s = __builtin_amdgcn_sdot2( __builtin_bit_cast( s2, va.y ), __builtin_bit_cast( s2, vb.y ), s, false );
      s = __builtin_amdgcn_sdot2( __builtin_bit_cast( s2, va.z ), __builtin_bit_cast( s2, vb.z ), s, false );
      s = __builtin_amdgcn_sdot2( __builtin_bit_cast( s2, va.w ), __builtin_bit_cast( s2, vb.w ), s, false );
    }
    sums[e * ALF_MAXB + b] = s;
  }
}

// Kernel B — the ordered part only: one workgroup (2 waves) per CTU, lane = entry.  Walks the CTU's 4x4 blocks in raster order and adds
// (float) sum into the accumulator of the block's class.  The 25 accumulators are ONE 32-wide vector value and the class is wave-uniform
// (lane b of the wave holds the classes of column b; v_readlane -> SGPR), so acc[class] += f is an indexed register access
// (s_set_gpr_idx), not a branch tree.  Sums of a whole block row (<= 32 blocks) are fetched one row ahead of the additions, the classes of
// the whole CTU up front: no HBM latency inside the ordered loop.  NCLS = 25 (luma) or 1 (chroma).
template<int NCLS, bool UNITS>      // UNITS: statistics units of several CTUs (traversal table); otherwise one CTU per unit: plain raster walk
__global__ void __launch_bounds__( 128 )
alfOrderedAddKernel( AlfStatArgs A )
{
  const int tid = threadIdx.x, lane = tid & 63;
  const int ctu = blockIdx.x, cx = ctu % A.ctusX, cy = ctu / A.ctusX;
  const int x0 = cx * A.ctuSize, y0 = cy * A.ctuSize;
  const int w = min( A.ctuSize, A.width - x0 ), h = min( A.ctuSize, A.height - y0 );
  const int nb = w >> 2, rows = h >> 2, nc = A.nc;
  const int nE = nc * ( nc + 1 ) / 2 + nc + 1;
  const int e = tid < nE ? tid : 0;
  const int4* sums = reinterpret_cast<const int4*>( A.sums + ( size_t ) ctu * A.blocksPerCtuRow * ALF_NE * ALF_MAXB + ( size_t ) e * ALF_MAXB );
  constexpr int ROW_I4 = ALF_NE * ALF_MAXB / 4;                // int4 per block row
  typedef float f32x32 __attribute__( ( ext_vector_type( 32 ) ) );
  f32x32 acc;
#pragma unroll
  for( int c = 0; c < 32; c++ ) acc[c] = 0.0f;
  if( A.init && tid < nE )                                     // continue the chains of an earlier CTU of the same statistics unit
  {
    int ra, rb; alfEntryRows( tid, nc, ra, rb );
    const float* in = A.init + ( size_t ) ctu * NCLS * ALF_REC + ( rb == 13 ? ( ra == 13 ? 182 : 169 + ra ) : ra * 13 + rb );
#pragma unroll
    for( int c = 0; c < NCLS; c++ ) acc[c] = in[c * ALF_REC];
  }
  int clsRow[ALF_MAXB];
#pragma unroll
  for( int r = 0; r < ALF_MAXB; r++ )
  {
    clsRow[r] = 0xffff;
    if( NCLS > 1 && r < rows && lane < nb ) { const uint8_t* c = A.cls + 2 * ( ( size_t ) ( ( y0 >> 2 ) + r ) * ( A.width >> 2 ) + ( x0 >> 2 ) + lane ); clsRow[r] = ( int ) c[0] | ( ( int ) c[1] << 8 ); }
  }
  // ring of 4 block rows of sums (8 x 16-byte loads each): row br + 3 is requested before row br is added — the scratch array comes from
  // HBM / Infinity Cache with ~2 us latency and a row's additions take ~0.6 us; two waves per CU own the whole register file
  int4 r0[8], r1[8], r2[8], r3[8];
  // traversal: a statistics unit may consist of several CTUs (alfUnitSize > CTU size): CTU by CTU in raster order, blocks in raster order inside
  // a CTU (getStatisticsASU, :1568-1590).  Step s = (CTU row sy, CTU column sx, block row br inside the CTU); subBlk = CTU size in blocks.
  const int subBlk = A.subBlk, nSubX = ( nb + subBlk - 1 ) / subBlk, nSubY = ( rows + subBlk - 1 ) / subBlk, steps = nSubY * nSubX * subBlk;
  __shared__ uint16_t sStep[4 * ALF_MAXB + 8];               // step -> unit block row | first block << 8 (computed once: no divisions in the ordered loop)
  for( int st = tid; st < steps + 8; st += 128 )
  {
    const int sy = st / ( nSubX * subBlk ), rem = st - sy * nSubX * subBlk, sx = rem / subBlk;
    sStep[st] = ( uint16_t ) ( ( sy * subBlk + ( rem - sx * subBlk ) ) | ( ( sx * subBlk ) << 8 ) );
  }
  __syncthreads();
#define ALF_STEP( S, UR, Q0 ) const int code_ = UNITS ? __builtin_amdgcn_readfirstlane( ( int ) sStep[S] ) : ( S ); const int UR = code_ & 0xff, Q0 = code_ >> 8;
#define ALF_FETCH( DST, S ) { ALF_STEP( S, ur_, q0_ ) ( void ) q0_; const int rr_ = ur_ < rows ? ur_ : rows - 1;                                \
    _Pragma( "unroll" ) for( int q = 0; q < 8; q++ ) DST[q] = sums[( size_t ) rr_ * ROW_I4 + q]; }
#define ALF_ADD( SRC, S ) if( ( S ) < steps ) { ALF_STEP( S, ur_, q0_ ) if( ur_ < rows ) { int myCls = 0;                                      \
    const int q1_ = min( nb, q0_ + subBlk );                                                                                             \
    _Pragma( "unroll" ) for( int r = 0; r < ALF_MAXB; r++ ) myCls = r == ur_ ? clsRow[r] : myCls;                                        \
    _Pragma( "unroll" ) for( int q = 0; q < ALF_MAXB; q++ ) if( UNITS ? ( q >= q0_ && q < q1_ ) : q < nb )                                \
    {                                                                                                                                    \
      const int4 v_ = SRC[q >> 2];                                                                                                       \
      const float f = ( float ) ( ( q & 3 ) == 0 ? v_.x : ( q & 3 ) == 1 ? v_.y : ( q & 3 ) == 2 ? v_.z : v_.w );                        \
      if( NCLS == 1 ) acc[0] += f;                                                                                                       \
      else { const int ct = __builtin_amdgcn_readlane( myCls, q ); if( ct != 0xffff ) acc[ct & 31] += f; }   /* 0xffff: m_ALF_UNUSED_CLASSIDX / _TRANSPOSIDX (:3416) */ \
    } } }
  ALF_FETCH( r0, 0 ) ALF_FETCH( r1, 1 ) ALF_FETCH( r2, 2 )
#pragma unroll 1
  for( int st = 0; st < steps; st += 4 )
  {
    ALF_FETCH( r3, st + 3 ) ALF_ADD( r0, st )
    ALF_FETCH( r0, st + 4 ) ALF_ADD( r1, st + 1 )
    ALF_FETCH( r1, st + 5 ) ALF_ADD( r2, st + 2 )
    ALF_FETCH( r2, st + 6 ) ALF_ADD( r3, st + 3 )
  }
#undef ALF_STEP
#undef ALF_FETCH
#undef ALF_ADD
  // records: E symmetric (upper triangle mirrored, :3493-3512), y, pixAcc; slots of unused coefficients are 0
  float* out = A.out + ( size_t ) ctu * NCLS * ALF_REC;
  for( int t = tid; t < NCLS * ALF_REC; t += 128 )
  {
    const int r = t % ALF_REC;
    bool written = false;
    if( r < 169 ) { const int k = r / 13, l = r % 13; written = k < nc && l < nc; } else if( r < 182 ) written = r - 169 < nc; else written = true;
    if( !written ) out[t] = 0.0f;
  }
  if( tid < nE )
  {
    int ra, rb; alfEntryRows( tid, nc, ra, rb );
#pragma unroll
    for( int c = 0; c < NCLS; c++ )
    {
      float* o = out + c * ALF_REC;
      if( rb == 13 ) o[ra == 13 ? 182 : 169 + ra] = acc[c];
      else { o[ra * 13 + rb] = acc[c]; o[rb * 13 + ra] = acc[c]; }
    }
  }
}

void buildTaps( AlfTaps& T )
{
  for( int shape = 0; shape < 2; shape++ )
  {
    const int L = shape == 0 ? 3 : 2;
    for( int tr = 0; tr < 4; tr++ )
    {
      int k = 0;
      AlfTap* t = T.t[shape][tr];
      for( int q = 0; q < 12; q++ ) t[q] = AlfTap{ 0, 0 };
      if( tr == 0 )      { for( int i = -L; i < 0; i++ ) for( int j = -L - i; j <= L + i; j++ ) t[k++] = AlfTap{ ( int8_t ) i, ( int8_t ) j }; for( int j = -L; j < 0; j++ ) t[k++] = AlfTap{ 0, ( int8_t ) j }; }
      else if( tr == 1 ) { for( int j = -L; j < 0; j++ ) for( int i = -L - j; i <= L + j; i++ ) t[k++] = AlfTap{ ( int8_t ) i, ( int8_t ) j }; for( int i = -L; i < 0; i++ ) t[k++] = AlfTap{ ( int8_t ) i, 0 }; }
      else if( tr == 2 ) { for( int i = -L; i < 0; i++ ) for( int j = L + i; j >= -L - i; j-- ) t[k++] = AlfTap{ ( int8_t ) i, ( int8_t ) j }; for( int j = -L; j < 0; j++ ) t[k++] = AlfTap{ 0, ( int8_t ) j }; }
      else               { for( int j = -L; j < 0; j++ ) for( int i = L + j; i >= -L - j; i-- ) t[k++] = AlfTap{ ( int8_t ) i, ( int8_t ) j }; for( int i = -L; i < 0; i++ ) t[k++] = AlfTap{ ( int8_t ) i, 0 }; }
    }
  }
}

} // namespace

static int alfLaunchStats( vvhip_ctx* ctx, AlfStatArgs& A, int width, int height, int ctu_size, bool ccalf )
{
  static const AlfTaps taps = []{ AlfTaps t; buildTaps( t ); return t; }();       // thread-safe one-time initialisation
  const int ctus = A.ctusX * ( ( height + ctu_size - 1 ) / ctu_size );
  const size_t need = ( size_t ) ctus * A.blocksPerCtuRow * ALF_NE * ALF_MAXB * sizeof( int32_t );           // per-block int32 sums between the two kernels
  if( need > ctx->scratchBytes )
  {
    VVHIP_CHECK_HIP( ctx, hipStreamSynchronize( ctx->stream ) );
    if( ctx->d_scratch ) ( void ) hipFree( ctx->d_scratch );
    ctx->d_scratch = nullptr; ctx->scratchBytes = 0;
    VVHIP_CHECK_HIP( ctx, hipMalloc( &ctx->d_scratch, need ) );
    ctx->scratchBytes = need;
  }
  A.sums = ( int32_t* ) ctx->d_scratch;
  if( ccalf ) hipLaunchKernelGGL( alfBlockSumsKernel<1>, dim3( ctus, A.blocksPerCtuRow ), dim3( 256 ), 0, ctx->stream, A, taps );
  else        hipLaunchKernelGGL( alfBlockSumsKernel<0>, dim3( ctus, A.blocksPerCtuRow ), dim3( 256 ), 0, ctx->stream, A, taps );
  VVHIP_LAUNCH_CHECK( ctx );
  const bool units = A.subBlk != A.blocksPerCtuRow;
  if( A.cls ) { if( units ) hipLaunchKernelGGL( ( alfOrderedAddKernel<25, true> ), dim3( ctus ), dim3( 128 ), 0, ctx->stream, A ); else hipLaunchKernelGGL( ( alfOrderedAddKernel<25, false> ), dim3( ctus ), dim3( 128 ), 0, ctx->stream, A ); }
  else        { if( units ) hipLaunchKernelGGL( ( alfOrderedAddKernel<1, true> ), dim3( ctus ), dim3( 128 ), 0, ctx->stream, A ); else hipLaunchKernelGGL( ( alfOrderedAddKernel<1, false> ), dim3( ctus ), dim3( 128 ), 0, ctx->stream, A ); }
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

extern "C" {

int vvhip_alf_classify( vvhip_ctx* ctx, const int16_t* d_rec, int stride, int width, int height, int bit_depth, int vb_ctu_height, int vb_pos, uint8_t* d_cls )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width < 4 || height < 4 || ( width & 3 ) || ( height & 3 ) || bit_depth < 8 || bit_depth > 12 || vb_ctu_height < 8 || ( vb_ctu_height & ( vb_ctu_height - 1 ) ) ||
      vb_pos < 4 || vb_pos > vb_ctu_height || !d_rec || !d_cls )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_alf_classify: %dx%d (multiples of 4), bitDepth %d, virtual boundary %d/%d (CTU height a power of two)", width, height, bit_depth, vb_pos, vb_ctu_height );
  const int nb = ( width >> 2 ) * ( height >> 2 );
  hipLaunchKernelGGL( alfClassifyKernel, dim3( ( nb + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_rec, stride, width, height, bit_depth + 4, vb_ctu_height, vb_pos, d_cls );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_alf_stats_plane_units( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_rec, int rec_stride, int width, int height, int unit_size, int ctu_size,
                                 int filter_length, const uint8_t* d_cls, int vb_ctu_height, int vb_pos, const float* d_init, float* d_out );

int vvhip_alf_stats_plane( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_rec, int rec_stride, int width, int height, int ctu_size,
                           int filter_length, const uint8_t* d_cls, int vb_ctu_height, int vb_pos, const float* d_init, float* d_out )
{
  return vvhip_alf_stats_plane_units( ctx, d_org, org_stride, d_rec, rec_stride, width, height, ctu_size, ctu_size, filter_length, d_cls, vb_ctu_height, vb_pos, d_init, d_out );
}

int vvhip_alf_stats_plane_units( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_rec, int rec_stride, int width, int height, int unit_size, int ctu_in_unit,
                                 int filter_length, const uint8_t* d_cls, int vb_ctu_height, int vb_pos, const float* d_init, float* d_out )
{
  const int ctu_size = unit_size;
  if( !ctx ) return VVHIP_E_ARG;
  if( width < 4 || height < 4 || ( width & 3 ) || ( height & 3 ) || ( filter_length != 7 && filter_length != 5 ) || ctu_size < 8 || ctu_size > 128 || ( ctu_size & 3 ) ||
      vb_ctu_height < 4 || ( vb_ctu_height & ( vb_ctu_height - 1 ) ) || vb_pos < 0 || !d_org || !d_rec || !d_out || ctu_in_unit < 8 || ( ctu_in_unit & 3 ) || unit_size % ctu_in_unit )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_alf_stats_plane: %dx%d (multiples of 4), CTU %d (<= 128), filter length %d (7 luma / 5 chroma)", width, height, ctu_size, filter_length );
  AlfStatArgs A;
  A.org = d_org; A.rec = d_rec; A.cls = d_cls; A.init = d_init; A.out = d_out; A.slf = nullptr; A.slfStride = 0; A.sx = 0; A.sy = 0; A.picHeight = 0; A.orgStride = org_stride; A.recStride = rec_stride; A.width = width; A.height = height;
  A.ctuSize = ctu_size; A.ctusX = ( width + ctu_size - 1 ) / ctu_size; A.nc = filter_length * filter_length / 4 + 1;
  A.shape = filter_length == 7 ? 0 : 1; A.vbH = vb_ctu_height; A.vbPos = vb_pos; A.blocksPerCtuRow = ctu_size >> 2; A.subBlk = ctu_in_unit >> 2;
  return alfLaunchStats( ctx, A, width, height, ctu_size, false );
}

int vvhip_ccalf_stats_plane( vvhip_ctx* ctx, const int16_t* d_org_c, int org_stride, const int16_t* d_slf_c, int slf_stride, const int16_t* d_rec_luma, int rec_stride,
                             int width_c, int height_c, int ctu_size_c, int shift_x, int shift_y, int vb_ctu_height, int vb_pos, int pic_height, const float* d_init, float* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width_c < 4 || height_c < 4 || ( width_c & 3 ) || ( height_c & 3 ) || ctu_size_c < 8 || ( ctu_size_c << shift_x ) > 128 || ( ctu_size_c & 3 ) || shift_x < 0 || shift_x > 1 || shift_y < 0 || shift_y > 1 ||
      vb_ctu_height < 4 || ( vb_ctu_height & ( vb_ctu_height - 1 ) ) || !d_org_c || !d_slf_c || !d_rec_luma || !d_out )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_ccalf_stats_plane: chroma %dx%d (multiples of 4), chroma CTU %d, shifts %d/%d", width_c, height_c, ctu_size_c, shift_x, shift_y );
  AlfStatArgs A;
  A.org = d_org_c; A.rec = d_rec_luma; A.cls = nullptr; A.init = d_init; A.out = d_out; A.orgStride = org_stride; A.recStride = rec_stride; A.width = width_c; A.height = height_c;
  A.slf = d_slf_c; A.slfStride = slf_stride; A.sx = shift_x; A.sy = shift_y; A.picHeight = pic_height;
  A.ctuSize = ctu_size_c; A.ctusX = ( width_c + ctu_size_c - 1 ) / ctu_size_c; A.nc = 7;
  A.shape = 1; A.vbH = vb_ctu_height; A.vbPos = vb_pos; A.blocksPerCtuRow = ctu_size_c >> 2; A.subBlk = ctu_size_c >> 2;
  return alfLaunchStats( ctx, A, width_c, height_c, ctu_size_c, true );
}

} // extern "C"
