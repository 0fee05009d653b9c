// dist.hip — distortion kernels (SAD / SSE / Hadamard-SATD / DMVR SADx5 / SAD cost surface) for gfx950.
//
// Reference semantics: CommonLib/RdCost.cpp:301-2093 (scalar) == CommonLib/x86/RdCostX86.h (SIMD);
// table rows RdCost::m_afpDistortFunc[0][DF_*] (RdCost.h:120).  Everything is integer except the
// rectangular Hadamard tiles' (int)(sad / sqrt(w*h) * 2) which is IEEE double divide + multiply.
//
// Execution model: a *team* of LPC consecutive lanes (1..64, power of two) owns one candidate, so a
// 64-lane wavefront evaluates 64/LPC small candidates at once and a 64x64 candidate fills a whole
// wave.  Lanes read 8/16-byte row segments straight from the picture planes (which stay resident in
// L2 / Infinity Cache across the thousands of candidates of a frame); costs are reduced with
// wave-level xor-shuffles; one lane stores the 64-bit Distortion.
#include <stdlib.h>
#include <vector>
#include "common.h"
#include <algorithm>

namespace {

typedef uint32_t u32x2 __attribute__( ( ext_vector_type( 2 ) ) );
typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
struct __attribute__( ( packed, aligned( 2 ) ) ) U4  { uint32_t v; };
struct __attribute__( ( packed, aligned( 2 ) ) ) U8  { u32x2 v; };
struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };

// picture samples are only 2-byte aligned (arbitrary motion vectors): gfx950 handles unaligned dword..dwordx4 loads
__device__ __forceinline__ uint32_t ld4( const int16_t* p ) { return reinterpret_cast<const U4*>( p )->v; }
__device__ __forceinline__ u32x2    ld8( const int16_t* p ) { return reinterpret_cast<const U8*>( p )->v; }
__device__ __forceinline__ u32x4    ld16( const int16_t* p ) { return reinterpret_cast<const U16*>( p )->v; }

// A 16-byte lane load at an address that is only 2-byte aligned (odd sample offset: half of all motion vectors) is split by the memory pipeline into dword pieces — measured,
// a Hadamard list whose candidates all sit at odd x costs 1.8x the same list at even x, and a wave with mixed parities pays the odd price.  With a second copy of the
// reference plane shifted by ONE sample at hand (shift1[k] == plane[k + 1]), an odd address reads the copy one sample earlier: every load is dword-aligned.
// delta = ( shift1 - plane ) - 1 in samples, 0 without the copy (the load is then the plain unaligned one).
__device__ __forceinline__ const int16_t* shiftSel( const int16_t* p, ptrdiff_t delta )
{
  return ( reinterpret_cast<uintptr_t>( p ) & 2 ) ? p + delta : p;
}

__device__ __forceinline__ int lo16( uint32_t v ) { return ( int ) ( int16_t ) ( v & 0xffffu ); }
__device__ __forceinline__ int hi16( uint32_t v ) { return ( int ) ( ( int32_t ) v >> 16 ); }

// |a.lo-b.lo| + |a.hi-b.hi| + acc for SIGNED int16 pairs: bias both to unsigned order, then v_sad_u16
__device__ __forceinline__ uint32_t sadPair( uint32_t a, uint32_t b, uint32_t acc )
{
  return __builtin_amdgcn_sad_u16( a ^ 0x80008000u, b ^ 0x80008000u, acc );
}

__device__ __forceinline__ uint32_t teamSum( uint32_t v, int lpc ) { return vvhipGroupSum32( v, lpc, threadIdx.x & 63 ); }
__device__ __forceinline__ uint64_t teamSum( uint64_t v, int lpc ) { return vvhipGroupSum64( v, lpc, threadIdx.x & 63 ); }

enum { MODE_SAD = 0, MODE_SSE = 1, MODE_SAD_X5 = 2, MODE_SAD_MIN2 = 3, MODE_SSE_PK = 4 };

// MODE_SSE_PK: SSE of operands the caller declares to be samples of a bit depth <= 12 (VVHIP_DIST_FLAG_SAMPLES): a difference fits 16 bits, eight squares fit 32 —
// packed subtraction + v_dot2_i32_i16 per sample pair instead of two widened subtractions and two 64-bit multiply-adds
typedef short s16x2q __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ int sqPair( uint32_t a, uint32_t b, int acc )
{
  const s16x2q d = __builtin_bit_cast( s16x2q, a ) - __builtin_bit_cast( s16x2q, b );
  return __builtin_amdgcn_sdot2( d, d, acc, false );
}

// ---------------------------------------------------------------------------------------------
// SAD / SSE over a candidate list.  CH = samples per lane per row segment (2, 4 or 8).
// chunk c of a candidate = (row c / lpr, segment c % lpr); lanes of a team stride over chunks.
// ---------------------------------------------------------------------------------------------
// U candidates per lane team are processed together: their (dependent) item and row loads are all in flight before the first reduction —
// small blocks have one chunk per lane, so without this a wave has only two loads outstanding.
template<int CH, int MODE, int U = 1>
__device__ __forceinline__ void
sadSseBody( int blockIndex, const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride,
            int lpr /* lanes (segments) per row = w / CH */, int lprShift /* log2(lpr) or -1 */, int rowsEff, int subShift, int log2Lpc,
            const vvhip_dist_item* __restrict__ items, int n, int calcCentre, uint64_t* __restrict__ out, ptrdiff_t curShift = 0 )
{
  const int gid  = blockIndex * blockDim.x + threadIdx.x;
  const int lpc  = 1 << log2Lpc;
  const int lt   = gid & ( lpc - 1 );
  const int nTeams = MODE == MODE_SAD_X5 ? n * 5 : n;
  const int step = 1 << subShift;
  int team[U], k[U];
  bool valid[U];
  const int16_t* po[U]; const int16_t* pc[U];
#pragma unroll
  for( int u = 0; u < U; u++ )
  {
    team[u] = ( gid >> log2Lpc ) * U + u;
    valid[u] = team[u] < nTeams;
    int orgOff = 0, curOff = 0; k[u] = 0;
    if( valid[u] )
    {
      const int idx = MODE == MODE_SAD_X5 ? team[u] / 5 : team[u];
      k[u] = MODE == MODE_SAD_X5 ? team[u] - idx * 5 : 0;
      const vvhip_dist_item it = items[idx];
      orgOff = it.org_off + k[u];      // RdCost.cpp:1988-2001: org.buf += k, cur.buf -= k
      curOff = it.cur_off - k[u];
    }
    po[u] = org + orgOff; pc[u] = cur + curOff;
    if( CH == 8 && curShift && !( curStride & 1 ) ) pc[u] = shiftSel( pc[u], curShift );      // (even stride: every row of the candidate has the parity of its first)
  }
  const int chunks = lpr * rowsEff;

  uint32_t sad[U];
  uint64_t sse[U];
#pragma unroll
  for( int u = 0; u < U; u++ ) { sad[u] = 0; sse[u] = 0; }
  for( int c = lt; c < chunks; c += lpc )
  {
    const int r = lprShift >= 0 ? c >> lprShift : c / lpr, s = c - r * lpr;
    const int y = r * step;
    uint32_t va[U][CH / 2], vb[U][CH / 2];
#pragma unroll
    for( int u = 0; u < U; u++ )
    {
      const int16_t* a = po[u] + ( ptrdiff_t ) y * orgStride + s * CH;
      const int16_t* b = pc[u] + ( ptrdiff_t ) y * curStride + s * CH;
      if( CH == 2 ) { va[u][0] = ld4( a ); vb[u][0] = ld4( b ); }
      else if( CH == 4 ) { u32x2 x = ld8( a ), z = ld8( b ); va[u][0] = x.x; va[u][CH / 2 - 1] = x.y; vb[u][0] = z.x; vb[u][CH / 2 - 1] = z.y; }
      else { u32x4 x, z = ld16( b );
             // candidates of one block follow each other in the lists: the original rows are fetched once per lane team (L1 request rate is what
             // bounds this kernel, not the bytes: a cache-hot reload costs as much as a miss that hits L2)
             if( u > 0 && po[u] == po[0] ) { x.x = va[0][0]; x.y = va[0][1 % ( CH / 2 )]; x.z = va[0][2 % ( CH / 2 )]; x.w = va[0][3 % ( CH / 2 )]; }
             else x = ld16( a );
             va[u][0] = x.x; va[u][1 % ( CH / 2 )] = x.y; va[u][2 % ( CH / 2 )] = x.z; va[u][3 % ( CH / 2 )] = x.w;
             vb[u][0] = z.x; vb[u][1 % ( CH / 2 )] = z.y; vb[u][2 % ( CH / 2 )] = z.z; vb[u][3 % ( CH / 2 )] = z.w; }
    }
#pragma unroll
    for( int u = 0; u < U; u++ )
      if( MODE == MODE_SSE_PK )
      {
        int e = 0;
#pragma unroll
        for( int i = 0; i < CH / 2; i++ ) e = sqPair( va[u][i], vb[u][i], e );
        sse[u] += ( uint32_t ) e;
      }
      else
#pragma unroll
      for( int i = 0; i < CH / 2; i++ )
      {
        if( MODE == MODE_SSE )
        {
          const int d0 = lo16( va[u][i] ) - lo16( vb[u][i] ), d1 = hi16( va[u][i] ) - hi16( vb[u][i] );
          sse[u] += ( uint64_t ) ( ( int64_t ) d0 * d0 ) + ( uint64_t ) ( ( int64_t ) d1 * d1 );
        }
        else sad[u] = sadPair( va[u][i], vb[u][i], sad[u] );
      }
  }

#pragma unroll
  for( int u = 0; u < U; u++ )
  {
    if( MODE == MODE_SSE || MODE == MODE_SSE_PK )
    {
      const uint64_t t = teamSum( sse[u], lpc );
      if( valid[u] && lt == 0 ) out[team[u]] = t;
    }
    else
    {
      const uint32_t t = teamSum( sad[u], lpc );
      if( valid[u] && lt == 0 )
      {
        const uint64_t v = ( uint64_t ) t << subShift;                 // RdCost.cpp:334
        if( MODE == MODE_SAD ) out[team[u]] = v;
        else if( MODE == MODE_SAD_X5 ) { if( k[u] != 2 || calcCentre ) out[team[u]] = v >> 1; }   // RdCost.cpp:2003-2007
        else { const uint64_t h = out[team[u]]; out[team[u]] = h < 2 * v ? h : 2 * v; }          // RdCost.cpp:1815
      }
    }
  }
}

template<int CH, int MODE>
__global__ void __launch_bounds__( 256 )
sadSseKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride,
              int lpr, int lprShift, int rowsEff, int subShift, int log2Lpc,
              const vvhip_dist_item* __restrict__ items, int n, int calcCentre, uint64_t* __restrict__ out )
{
  sadSseBody<CH, MODE>( blockIdx.x, org, orgStride, cur, curStride, lpr, lprShift, rowsEff, subShift, log2Lpc, items, n, calcCentre, out );
}

// Several (function-compatible) batches in ONE launch: a workgroup finds its job from the block-range table (wave-uniform scalar
// work) and runs the same body.  Removes the launch gaps and the tails of the short per-size launches of a frame's work lists.
#ifndef VVHIP_DIST_U
#define VVHIP_DIST_U 2
#endif
constexpr int DIST_U = VVHIP_DIST_U;   // candidates per lane team in the merged SAD / SSE launches (they share the original rows when they belong to one block)
struct DistJobGeom { int lpr, lprShift, rowsEff, subShift, log2Lpc, n, blockStart, nBlocks, fast16, tilesX, tilesPerCand, sse, tiled, shift; const vvhip_dist_item* items; uint64_t* out; };
// 8x8-tiled copies of the two planes (vvhip_plane_tile8): a 128-byte cache line = ONE 8x8 tile of int16 samples.  An 8x8 candidate of a row-major plane is eight 16-byte
// pieces in eight cache lines — every piece is an L1 access of its own (and a line fill out of L2), which is what bounds the 8x8 lists; in the tiled copy the same
// candidate lies in at most four lines, an aligned original block in one (29 -> 10.5 L1 accesses per SAD candidate).
struct Tiled8 { const int16_t* org; const int16_t* cur; int orgTpr, curTpr, orgBias, curBias, orgStride, curStride; unsigned long long orgMagic, curMagic; };
struct DistMultiJobs { int nJobs, xcdRemap; DistJobGeom j[8]; Tiled8 T; ptrdiff_t curShift; };

// sample offset relative to sample (0,0) of a row-major plane -> coordinates inside the padded plane (bias = margin * stride + margin; magic = 2^40 / stride + 1)
__device__ __forceinline__ void tiledXY( int off, int bias, int stride, unsigned long long magic, int& x, int& y )
{
  const uint32_t o = ( uint32_t ) ( off + bias );
  int q = ( int ) ( ( ( unsigned long long ) o * magic ) >> 40 );
  int r = ( int ) o - q * stride;
  if( r < 0 ) { q--; r += stride; } else if( r >= stride ) { q++; r -= stride; }
  x = r; y = q;
}
// eight samples (x .. x+7, y) of a tiled plane: two aligned 16-byte tile rows, funnel-shifted by x & 7
__device__ __forceinline__ u32x4 tiledRow8( const int16_t* __restrict__ t, int tpr, int x, int y )
{
  const int16_t* p = t + ( ( size_t ) ( y >> 3 ) * tpr + ( x >> 3 ) ) * 64 + ( y & 7 ) * 8;
  const u32x4 a = *reinterpret_cast<const u32x4*>( p );
  const int sh = x & 7;
  if( __builtin_amdgcn_ballot_w64( sh != 0 ) == 0ull ) return a;                       // the whole wave is tile-aligned (original blocks of a picture tiling)
  const u32x4 b = *reinterpret_cast<const u32x4*>( p + 64 );                            // (reads the next tile also when sh == 0: inside the padded plane by construction)
  const uint32_t s0 = a.x, s1 = a.y, s2 = a.z, s3 = a.w, s4 = b.x, s5 = b.y, s6 = b.z, s7 = b.w;
  const bool k1 = ( sh & 2 ) != 0, k2 = ( sh & 4 ) != 0;
  const uint32_t u0 = k1 ? s1 : s0, u1 = k1 ? s2 : s1, u2 = k1 ? s3 : s2, u3 = k1 ? s4 : s3, u4 = k1 ? s5 : s4, u5 = k1 ? s6 : s5, u6 = k1 ? s7 : s6;
  const uint32_t v0 = k2 ? u2 : u0, v1 = k2 ? u3 : u1, v2 = k2 ? u4 : u2, v3 = k2 ? u5 : u3, v4 = k2 ? u6 : u4;
  const uint32_t bits = ( sh & 1 ) * 16;
  u32x4 o;
  o.x = __builtin_amdgcn_alignbit( v1, v0, bits ); o.y = __builtin_amdgcn_alignbit( v2, v1, bits ); o.z = __builtin_amdgcn_alignbit( v3, v2, bits ); o.w = __builtin_amdgcn_alignbit( v4, v3, bits );
  return o;
}

// SAD / SSE of 8x8 candidates on the tiled copies.  The list is VALU-issue-bound, not memory-bound (four v_sad_u16 per lane against an offset -> (x, y) division, tile addresses
// and a funnel shift), so nothing is computed twice: a wave takes 64 candidates, lane L decomposes candidate L, and in round j the eight lanes of team t (lane = row) work on
// candidate 8t + j, whose coordinates they fetch from lane 8t + j (ds_bpermute); lane j of the team keeps that round's total, so candidate L's result ends in lane L:
// 64 coalesced result stores per wave.
// (general SSE, arbitrary int16 operands: 64-bit row sums — the plain form, 8 lanes per candidate and 8 candidates per wave; workgroups are sized for it by the host)
__device__ __forceinline__ void
sse8TiledGeneralBody( int blockIndex, const Tiled8& T, const vvhip_dist_item* __restrict__ items, int n, uint64_t* __restrict__ out )
{
  const int gid = blockIndex * blockDim.x + threadIdx.x, r = gid & 7;
  // this workgroup's candidates are [blockIndex * blockDim, + blockDim): eight passes of blockDim / 8
  for( int pass = 0; pass < 8; pass++ )
  {
    const int cand = blockIndex * blockDim.x + pass * ( blockDim.x >> 3 ) + ( threadIdx.x >> 3 );
    const bool valid = cand < n;
    int ox = 0, oy = 0, cx = 0, cy = 0;
    if( valid )
    {
      const vvhip_dist_item it = items[cand];
      tiledXY( it.org_off, T.orgBias, T.orgStride, T.orgMagic, ox, oy );
      tiledXY( it.cur_off, T.curBias, T.curStride, T.curMagic, cx, cy );
    }
    const u32x4 a = tiledRow8( T.org, T.orgTpr, ox, oy + r ), b = tiledRow8( T.cur, T.curTpr, cx, cy + r );
    const uint32_t as[4] = { a.x, a.y, a.z, a.w }, bs[4] = { b.x, b.y, b.z, b.w };
    uint64_t e = 0;
#pragma unroll
    for( int i = 0; i < 4; i++ ) { const int d0 = lo16( as[i] ) - lo16( bs[i] ), d1 = hi16( as[i] ) - hi16( bs[i] ); e += ( uint64_t ) ( ( int64_t ) d0 * d0 ) + ( uint64_t ) ( ( int64_t ) d1 * d1 ); }
    const uint64_t t = vvhipGroupSum64( e, 8, threadIdx.x & 63 );
    if( valid && r == 0 ) out[cand] = t;
  }
}

template<int MODE>
__device__ __forceinline__ void
sadSse8TiledBody( int blockIndex, const Tiled8& T, const vvhip_dist_item* __restrict__ items, int n, uint64_t* __restrict__ out )
{
  if( MODE == MODE_SSE ) { sse8TiledGeneralBody( blockIndex, T, items, n, out ); return; }
  const int lane = threadIdx.x & 63, r = lane & 7;
  const int cand = blockIndex * blockDim.x + threadIdx.x;                  // the candidate this lane decomposes and stores
  const bool valid = cand < n;
  uint32_t po = 0, pc = 0;                                                  // ( x | y << 16 ) in the padded planes (both < 2^16: checked on the host)
  if( valid )
  {
    const vvhip_dist_item it = items[cand];
    int ox, oy, cx, cy;
    tiledXY( it.org_off, T.orgBias, T.orgStride, T.orgMagic, ox, oy );
    tiledXY( it.cur_off, T.curBias, T.curStride, T.curMagic, cx, cy );
    po = ( uint32_t ) ox | ( ( uint32_t ) oy << 16 ); pc = ( uint32_t ) cx | ( ( uint32_t ) cy << 16 );
  }
  uint32_t tot = 0;
#pragma unroll 2
  for( int j = 0; j < 8; j++ )
  {
    const int srcLane = ( ( lane & ~7 ) + j ) << 2;
    const uint32_t qo = ( uint32_t ) __builtin_amdgcn_ds_bpermute( srcLane, ( int ) po ), qc = ( uint32_t ) __builtin_amdgcn_ds_bpermute( srcLane, ( int ) pc );
    const u32x4 a = tiledRow8( T.org, T.orgTpr, ( int ) ( qo & 0xffffu ), ( int ) ( qo >> 16 ) + r ), b = tiledRow8( T.cur, T.curTpr, ( int ) ( qc & 0xffffu ), ( int ) ( qc >> 16 ) + r );
    uint32_t e;
    if( MODE == MODE_SSE_PK ) { int q = sqPair( a.x, b.x, 0 ); q = sqPair( a.y, b.y, q ); q = sqPair( a.z, b.z, q ); q = sqPair( a.w, b.w, q ); e = ( uint32_t ) q; }      // 8 squares < 2^26 each
    else { e = sadPair( a.x, b.x, 0 ); e = sadPair( a.y, b.y, e ); e = sadPair( a.z, b.z, e ); e = sadPair( a.w, b.w, e ); }
    const uint32_t t = vvhipGroupSum32( e, 8, lane );          // every lane of the team gets the candidate's total; lane j of the team keeps it
    tot = r == j ? t : tot;
  }
  if( valid ) out[cand] = tot;
}

// Workgroups are dealt round-robin to the 8 XCDs (private L2 each).  Work lists are in raster order of the picture, so giving XCD x
// the x-th contiguous eighth of a list makes every L2 stream one band of the planes instead of all of them.
__device__ __forceinline__ int xcdBand( int local, int nBlocks )
{
  const int q = nBlocks >> 3;
  return local < ( q << 3 ) ? ( local & 7 ) * q + ( local >> 3 ) : local;
}

template<int MODE>
__global__ void __launch_bounds__( 256 )
sadSseMultiKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride, DistMultiJobs jobs )
{
  int k = 0;
#pragma unroll
  for( int i = 1; i < 8; i++ ) if( i < jobs.nJobs && ( int ) blockIdx.x >= jobs.j[i].blockStart ) k = i;
  const DistJobGeom& g = jobs.j[k];
  int blk = blockIdx.x - g.blockStart;
  if( jobs.xcdRemap ) blk = xcdBand( blk, g.nBlocks );
  if( MODE == MODE_SSE && g.sse == 2 )
  {
    if( g.tiled ) sadSse8TiledBody<MODE_SSE_PK>( blk, jobs.T, g.items, g.n, g.out );
    else          sadSseBody<8, MODE_SSE_PK, DIST_U>( blk, org, orgStride, cur, curStride, g.lpr, g.lprShift, g.rowsEff, g.subShift, g.log2Lpc, g.items, g.n, 0, g.out, g.shift ? jobs.curShift : 0 );
  }
  else if( g.tiled ) sadSse8TiledBody<MODE>( blk, jobs.T, g.items, g.n, g.out );
  else          sadSseBody<8, MODE, DIST_U>( blk, org, orgStride, cur, curStride, g.lpr, g.lprShift, g.rowsEff, g.subShift, g.log2Lpc, g.items, g.n, 0, g.out, g.shift ? jobs.curShift : 0 );
}

// SAD and SSE lists of one frame in the same launch (per-job mode): both are short, memory-side kernels with the same geometry
__global__ void __launch_bounds__( 256 )
sadSseMixedKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride, DistMultiJobs jobs )
{
  int k = 0;
#pragma unroll
  for( int i = 1; i < 8; i++ ) if( i < jobs.nJobs && ( int ) blockIdx.x >= jobs.j[i].blockStart ) k = i;
  const DistJobGeom& g = jobs.j[k];
  int blk = blockIdx.x - g.blockStart;
  if( jobs.xcdRemap ) blk = xcdBand( blk, g.nBlocks );
  if( g.sse == 2 )
  {
    if( g.tiled ) sadSse8TiledBody<MODE_SSE_PK>( blk, jobs.T, g.items, g.n, g.out );
    else          sadSseBody<8, MODE_SSE_PK, DIST_U>( blk, org, orgStride, cur, curStride, g.lpr, g.lprShift, g.rowsEff, g.subShift, g.log2Lpc, g.items, g.n, 0, g.out, g.shift ? jobs.curShift : 0 );
  }
  else if( g.tiled ) { if( g.sse ) sadSse8TiledBody<MODE_SSE>( blk, jobs.T, g.items, g.n, g.out ); else sadSse8TiledBody<MODE_SAD>( blk, jobs.T, g.items, g.n, g.out ); }
  else if( g.sse ) sadSseBody<8, MODE_SSE, DIST_U>( blk, org, orgStride, cur, curStride, g.lpr, g.lprShift, g.rowsEff, g.subShift, g.log2Lpc, g.items, g.n, 0, g.out, g.shift ? jobs.curShift : 0 );
  else        sadSseBody<8, MODE_SAD, DIST_U>( blk, org, orgStride, cur, curStride, g.lpr, g.lprShift, g.rowsEff, g.subShift, g.log2Lpc, g.items, g.n, 0, g.out, g.shift ? jobs.curShift : 0 );
}

// ---------------------------------------------------------------------------------------------
// Hadamard SATD.  A tile of TW x TH differences is held one row per lane (TH lanes = a tile team):
// horizontal Walsh-Hadamard in registers, vertical across lanes with xor-shuffles.  SATD is a sum of
// absolute coefficients, so butterfly order/sign conventions are irrelevant; only the DC term
// (plain sum of differences) is special: |DC| is replaced by |DC| >> 2 in every tile type.
// FAST16: the 16x16_fast tile = 8x8 Hadamard of 2x2-averaged org and cur (RdCost.cpp:1126-1223).
// ---------------------------------------------------------------------------------------------
template<int TW> __device__ __forceinline__ void loadRowDiff( const int16_t* a, const int16_t* b, int ( &d )[TW] )
{
  if( TW == 2 ) { const uint32_t x = ld4( a ), z = ld4( b ); d[0] = lo16( x ) - lo16( z ); d[1 % TW] = hi16( x ) - hi16( z ); }
  else if( TW == 4 )
  {
    const u32x2 x = ld8( a ), z = ld8( b );
    d[0] = lo16( x.x ) - lo16( z.x ); d[1 % TW] = hi16( x.x ) - hi16( z.x ); d[2 % TW] = lo16( x.y ) - lo16( z.y ); d[3 % TW] = hi16( x.y ) - hi16( z.y );
  }
  else
  {
#pragma unroll
    for( int q = 0; q < TW / 8; q++ )
    {
      const u32x4 x = ld16( a + 8 * q ), z = ld16( b + 8 * q );
      const uint32_t xs[4] = { x.x, x.y, x.z, x.w }, zs[4] = { z.x, z.y, z.z, z.w };
#pragma unroll
      for( int i = 0; i < 4; i++ )
      {
        d[( 8 * q + 2 * i ) % TW]     = lo16( xs[i] ) - lo16( zs[i] );
        d[( 8 * q + 2 * i + 1 ) % TW] = hi16( xs[i] ) - hi16( zs[i] );
      }
    }
  }
}

// rounded 2x2 averages of 16 samples x 2 rows -> 8 values
__device__ __forceinline__ void avgRow16( const int16_t* p, int stride, int ( &o )[8] )
{
  const u32x4 a0 = ld16( p ), a1 = ld16( p + 8 ), b0 = ld16( p + stride ), b1 = ld16( p + stride + 8 );
  const uint32_t t[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
  const uint32_t u[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
  for( int i = 0; i < 8; i++ ) o[i] = ( lo16( t[i] ) + hi16( t[i] ) + lo16( u[i] ) + hi16( u[i] ) + 2 ) >> 2;
}

template<int TW, int TH, bool FAST16>
__global__ void __launch_bounds__( 256 )
hadKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride,
           int tilesX, int tilesPerCand, int log2Lpc,
           const vvhip_dist_item* __restrict__ items, int n, uint64_t* __restrict__ out )
{
  constexpr int PX = FAST16 ? 16 : TW;     // picture samples covered by a tile horizontally / vertically
  constexpr int PY = FAST16 ? 16 : TH;
  const int gid  = blockIdx.x * blockDim.x + threadIdx.x;
  const int lpc  = 1 << log2Lpc;
  const int cand = gid >> log2Lpc;
  const int lt   = gid & ( lpc - 1 );
  const bool valid = cand < n;
  const int row  = lt & ( TH - 1 );        // my row inside the tile
  const int tt   = lt / TH;                // tile team inside the candidate
  const int teams = lpc / TH;

  int orgOff = 0, curOff = 0;
  if( valid ) { const vvhip_dist_item it = items[cand]; orgOff = it.org_off; curOff = it.cur_off; }

  uint64_t sum = 0;
  const int tiles = valid ? tilesPerCand : 0;
  // all lanes of a wave run the same trip count (teams own tiles tt, tt+teams, ..); tilesPerCand % teams == 0 or teams == 1 by construction
  for( int t = tt; t < tiles; t += teams )
  {
    const int ty = t / tilesX, tx = t - ty * tilesX;
    int d[TW];
    if( FAST16 )
    {
      int ao[8], ac[8];
      avgRow16( org + orgOff + ( ptrdiff_t ) ( ty * PY + 2 * row ) * orgStride + tx * PX, orgStride, ao );
      avgRow16( cur + curOff + ( ptrdiff_t ) ( ty * PY + 2 * row ) * curStride + tx * PX, curStride, ac );
#pragma unroll
      for( int i = 0; i < 8; i++ ) d[i % TW] = ao[i] - ac[i];
    }
    else
      loadRowDiff<TW>( org + orgOff + ( ptrdiff_t ) ( ty * PY + row ) * orgStride + tx * PX,
                       cur + curOff + ( ptrdiff_t ) ( ty * PY + row ) * curStride + tx * PX, d );

    // horizontal WHT (in registers)
#pragma unroll
    for( int len = 1; len < TW; len <<= 1 )
#pragma unroll
      for( int i = 0; i < TW; i += 2 * len )
#pragma unroll
        for( int j = i; j < i + len; j++ ) { const int a = d[j], b = d[j + len]; d[j] = a + b; d[j + len] = a - b; }
    // vertical WHT across the TH lanes of the tile team, DPP only.  Mirror pairings (i <-> 15-i, i <-> 7-i) followed by the
    // quad xor-2 / xor-1 pairings form a valid Hadamard factorisation (directions 1111,0111,0010,0001 are independent over GF(2));
    // coefficient order and sign differ from the reference's butterflies, the multiset of |coefficients| and the DC lane (0) do not.
#define VSTAGE( CTRL, BIT ) { const bool upper = ( row & ( BIT ) ) != 0; _Pragma( "unroll" ) \
      for( int i = 0; i < TW; i++ ) { const int o = VVHIP_DPP( d[i], CTRL ); d[i] = upper ? o - d[i] : d[i] + o; } }
    if( TH >= 16 ) VSTAGE( VVHIP_DPP_MIRROR, 8 )
    if( TH >= 8 )  VSTAGE( VVHIP_DPP_HALF_MIRROR, 4 )
    if( TH >= 4 )  VSTAGE( VVHIP_DPP_XOR2, 2 )
    if( TH >= 2 )  VSTAGE( VVHIP_DPP_XOR1, 1 )
#undef VSTAGE
    uint32_t s = 0;
#pragma unroll
    for( int i = 0; i < TW; i++ ) s += ( uint32_t ) abs( d[i] );
    if( row == 0 ) { const uint32_t dc = ( uint32_t ) abs( d[0] ); s = s - dc + ( dc >> 2 ); }
    s = vvhipGroupSum32( s, TH, threadIdx.x & 63 );
    if( row == 0 )
    {
      uint32_t v;
      if( FAST16 )              v = ( ( s + 2 ) >> 2 ) << 2;                                         // RdCost.cpp:1220-1222
      else if( TW != TH )       v = ( uint32_t ) ( int ) ( ( double ) ( int ) s / __builtin_sqrt( ( double ) ( TW * TH ) ) * 2 );  // :1467,1606,1682,1763
      else if( TW == 8 )        v = ( s + 2 ) >> 2;                                                  // :1319
      else if( TW == 4 )        v = ( s + 1 ) >> 1;                                                  // :1121
      else                      v = s;                                                               // :1020-1023
      sum += v;
    }
  }
  sum = vvhipGroupSum64( sum, lpc, threadIdx.x & 63 );
  if( valid && lt == 0 ) out[cand] = sum;
}

// ---------------------------------------------------------------------------------------------
// Hadamard SATD, one whole tile per LANE (tiles up to 64 differences: 8x8, 16x16_fast, 8x4, 4x8, 4x4, 2x2).
// Both butterfly directions run on registers: no cross-lane traffic at all and ~2-3x fewer instructions per
// sample than the row-per-lane form above (which remains for the 128-difference 16x8 / 8x16 tiles).
// A candidate's tiles sit on LPC consecutive lanes (LPC = min(#tiles, 64), power of two) and are summed with DPP.
// ---------------------------------------------------------------------------------------------
template<int TW, int TH, bool FAST16>
__device__ __forceinline__ void
hadTileBody( int blockIndex, const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride,
             int tilesX, int tilesPerCand, int log2Lpc,
             const vvhip_dist_item* __restrict__ items, int n, uint64_t* __restrict__ out, ptrdiff_t curShift = 0 )
{
  constexpr int PX = FAST16 ? 16 : TW, PY = FAST16 ? 16 : TH;
  const int gid  = blockIndex * blockDim.x + threadIdx.x;
  const int lpc  = 1 << log2Lpc;
  const int cand = gid >> log2Lpc;
  const int lt   = gid & ( lpc - 1 );
  const bool valid = cand < n;
  int orgOff = 0, curOff = 0;
  if( valid ) { const vvhip_dist_item it = items[cand]; orgOff = it.org_off; curOff = it.cur_off; }
  const int tiles = valid ? tilesPerCand : 0;

  uint32_t sum = 0;
  for( int t = lt; t < tiles; t += lpc )
  {
    const int ty = t / tilesX, tx = t - ty * tilesX;
    const int16_t* po = org + orgOff + ( ptrdiff_t ) ( ty * PY ) * orgStride + tx * PX;
    const int16_t* pc = cur + curOff + ( ptrdiff_t ) ( ty * PY ) * curStride + tx * PX;
    if( ( FAST16 || TW == 8 ) && curShift && !( curStride & 1 ) ) pc = shiftSel( pc, curShift );
    int d[TH][TW];
#pragma unroll
    for( int r = 0; r < TH; r++ )
    {
      if( FAST16 )
      {
        int ao[8], ac[8];
        avgRow16( po + ( ptrdiff_t ) ( 2 * r ) * orgStride, orgStride, ao );
        avgRow16( pc + ( ptrdiff_t ) ( 2 * r ) * curStride, curStride, ac );
#pragma unroll
        for( int i = 0; i < 8; i++ ) d[r][i % TW] = ao[i] - ac[i];
      }
      else loadRowDiff<TW>( po + ( ptrdiff_t ) r * orgStride, pc + ( ptrdiff_t ) r * curStride, d[r] );
      // horizontal WHT of this row
#pragma unroll
      for( int len = 1; len < TW; len <<= 1 )
#pragma unroll
        for( int i = 0; i < TW; i += 2 * len )
#pragma unroll
          for( int j = i; j < i + len; j++ ) { const int a = d[r][j], b = d[r][j + len]; d[r][j] = a + b; d[r][j + len] = a - b; }
    }
    // vertical WHT
#pragma unroll
    for( int len = 1; len < TH; len <<= 1 )
#pragma unroll
      for( int i = 0; i < TH; i += 2 * len )
#pragma unroll
        for( int j = i; j < i + len; j++ )
#pragma unroll
          for( int c = 0; c < TW; c++ ) { const int a = d[j][c], b = d[j + len][c]; d[j][c] = a + b; d[j + len][c] = a - b; }
    uint32_t s = 0;
#pragma unroll
    for( int r = 0; r < TH; r++ )
#pragma unroll
      for( int c = 0; c < TW; c++ ) s += ( uint32_t ) abs( d[r][c] );
    const uint32_t dc = ( uint32_t ) abs( d[0][0] );
    s = s - dc + ( dc >> 2 );
    uint32_t v;
    if( FAST16 )        v = ( ( s + 2 ) >> 2 ) << 2;                                                                   // RdCost.cpp:1220-1222
    else if( TW != TH ) v = ( uint32_t ) ( int ) ( ( double ) ( int ) s / __builtin_sqrt( ( double ) ( TW * TH ) ) * 2 );  // :1682,1763
    else if( TW == 8 )  v = ( s + 2 ) >> 2;                                                                            // :1319
    else if( TW == 4 )  v = ( s + 1 ) >> 1;                                                                            // :1121
    else                v = s;                                                                                         // :1020-1023
    sum += v;
  }
  // per-lane sums stay below 2^32 (<= 4 tiles x 64 x 2^20); the candidate total can exceed it only for full-range int16 data -> 64-bit group sum
  const uint64_t tot = vvhipGroupSum64( sum, lpc, threadIdx.x & 63 );
  if( valid && lt == 0 ) out[cand] = tot;
}

// ---------------------------------------------------------------------------------------------
// 8x8 / 16x16_fast Hadamard tile per lane in PACKED 16-bit arithmetic, for bit depths <= 10 (the same contract as the reference's x86
// rows: xCalcHAD8x8_SSE / xCalcHAD16x16_fast_SSE CHECK( iBitDepth > 10 ), x86/RdCostX86.h:655-659,800-804).
// The 64 differences of a tile are 32 dwords of two samples.  Two forms:
//   WIDE (default): inputs may be anything the encoder hands to a <= 10-bit HAD entry, including the bi-prediction pattern 2*org - pred
//     (values -1023..2046, InterSearch.cpp:1996-2003), so |difference| <= 2047.  Four butterfly stages over the dword index run packed
//     (v_pk_add_i16 / v_pk_sub_i16: 2047 * 16 < 32768), the fifth is carried out in 32 bits (the reference's x86 row widens after its third
//     stage, x86/RdCostX86.h:700-760), the sixth never has to be: |a + b| + |a - b| = 2 * max( |a|, |b| ).  The 2x2 averages of the fast tile
//     use an arithmetic packed shift (signed inputs).
//   NARROW (job flag VVHIP_DIST_FLAG_SAMPLES: the caller asserts both operands are samples in [0, 2^bit_depth)): |difference| <= 1023, all
//     five dword stages stay packed (1023 * 32 < 32768).  Fewer instructions; identical results on that domain.
// ---------------------------------------------------------------------------------------------
typedef short s16x2v __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ uint32_t pkAdd( uint32_t a, uint32_t b ) { return __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2v, a ) + __builtin_bit_cast( s16x2v, b ) ); }
__device__ __forceinline__ uint32_t pkSub( uint32_t a, uint32_t b ) { return __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2v, a ) - __builtin_bit_cast( s16x2v, b ) ); }
__device__ __forceinline__ uint32_t pkAbs( uint32_t a )
{
  const s16x2v v = __builtin_bit_cast( s16x2v, a ), z = { 0, 0 };
  return __builtin_bit_cast( uint32_t, __builtin_elementwise_max( v, z - v ) );
}

// rounded 2x2 averages of 16 samples x 2 rows -> 8 values as 4 packed dwords; SIGNED: inputs may be negative (arithmetic shift)
template<bool SIGNED>
__device__ __forceinline__ void avgRow16PkData( const u32x4 a0, const u32x4 a1, const u32x4 b0, const u32x4 b1, uint32_t ( &o )[4] )
{
  const uint32_t t[8] = { pkAdd( a0.x, b0.x ), pkAdd( a0.y, b0.y ), pkAdd( a0.z, b0.z ), pkAdd( a0.w, b0.w ), pkAdd( a1.x, b1.x ), pkAdd( a1.y, b1.y ), pkAdd( a1.z, b1.z ), pkAdd( a1.w, b1.w ) };
#pragma unroll
  for( int i = 0; i < 4; i++ )
  {
    const uint32_t lo = __builtin_amdgcn_perm( t[2 * i + 1], t[2 * i], 0x05040100u ), hi = __builtin_amdgcn_perm( t[2 * i + 1], t[2 * i], 0x07060302u );   // (t0.lo, t1.lo), (t0.hi, t1.hi)
    const uint32_t s = pkAdd( pkAdd( lo, hi ), 0x00020002u );
    if( SIGNED ) o[i] = __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2v, s ) >> 2 );   // v_pk_ashrrev_i16 (|sum| <= 4 * 2046 + 2)
    else         o[i] = ( s >> 2 ) & 0x3fff3fffu;                                               // both halves >> 2 (sums are < 2^13, unsigned)
  }
}
template<bool SIGNED>
__device__ __forceinline__ void avgRow16Pk( const int16_t* p, int stride, uint32_t ( &o )[4] )
{
  avgRow16PkData<SIGNED>( ld16( p ), ld16( p + 8 ), ld16( p + stride ), ld16( p + stride + 8 ), o );
}

// TILED: the samples come from the 8x8-tiled copies of the planes (Tiled8): an 8x8 Hadamard tile of the original is one cache line, of the candidate at most four
template<bool FAST16, bool WIDE, bool TILED = false>
__device__ __forceinline__ void
hadTilePkBody( int blockIndex, const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride,
               int tilesX, int tilesPerCand, int log2Lpc,
               const vvhip_dist_item* __restrict__ items, int n, uint64_t* __restrict__ out, const Tiled8* __restrict__ T = nullptr, ptrdiff_t curShift = 0 )
{
  constexpr int PX = FAST16 ? 16 : 8, PY = FAST16 ? 16 : 8;
  const int gid  = blockIndex * blockDim.x + threadIdx.x;
  const int lpc  = 1 << log2Lpc;
  const int cand = gid >> log2Lpc;
  const int lt   = gid & ( lpc - 1 );
  const bool valid = cand < n;
  int orgOff = 0, curOff = 0;
  if( valid ) { const vvhip_dist_item it = items[cand]; orgOff = it.org_off; curOff = it.cur_off; }
  const int tiles = valid ? tilesPerCand : 0;
  int ox = 0, oy = 0, cx = 0, cy = 0;
  if( TILED )
  {
    tiledXY( orgOff, T->orgBias, T->orgStride, T->orgMagic, ox, oy );
    tiledXY( curOff, T->curBias, T->curStride, T->curMagic, cx, cy );
  }

  uint32_t sum = 0;
  for( int t = lt; t < tiles; t += lpc )
  {
    const int ty = t / tilesX, tx = t - ty * tilesX;
    const int16_t* po = org + orgOff + ( ptrdiff_t ) ( ty * PY ) * orgStride + tx * PX;
    const int16_t* pc = cur + curOff + ( ptrdiff_t ) ( ty * PY ) * curStride + tx * PX;
    if( !TILED && curShift && !( curStride & 1 ) ) pc = shiftSel( pc, curShift );
    uint32_t d[32];                                           // dword 4 * r + q: differences (r, 2q), (r, 2q + 1)
#pragma unroll
    for( int r = 0; r < 8; r++ )
    {
      if( FAST16 )
      {
        uint32_t ao[4], ac[4];
        if( TILED )
        {
          const int X = ox + tx * PX, Y = oy + ty * PY + 2 * r, CX = cx + tx * PX, CY = cy + ty * PY + 2 * r;
          avgRow16PkData<WIDE>( tiledRow8( T->org, T->orgTpr, X, Y ), tiledRow8( T->org, T->orgTpr, X + 8, Y ), tiledRow8( T->org, T->orgTpr, X, Y + 1 ), tiledRow8( T->org, T->orgTpr, X + 8, Y + 1 ), ao );
          avgRow16PkData<WIDE>( tiledRow8( T->cur, T->curTpr, CX, CY ), tiledRow8( T->cur, T->curTpr, CX + 8, CY ), tiledRow8( T->cur, T->curTpr, CX, CY + 1 ), tiledRow8( T->cur, T->curTpr, CX + 8, CY + 1 ), ac );
        }
        else
        {
          avgRow16Pk<WIDE>( po + ( ptrdiff_t ) ( 2 * r ) * orgStride, orgStride, ao );
          avgRow16Pk<WIDE>( pc + ( ptrdiff_t ) ( 2 * r ) * curStride, curStride, ac );
        }
#pragma unroll
        for( int q = 0; q < 4; q++ ) d[4 * r + q] = pkSub( ao[q], ac[q] );
      }
      else
      {
        const u32x4 x = TILED ? tiledRow8( T->org, T->orgTpr, ox + tx * PX, oy + ty * PY + r ) : ld16( po + ( ptrdiff_t ) r * orgStride );
        const u32x4 z = TILED ? tiledRow8( T->cur, T->curTpr, cx + tx * PX, cy + ty * PY + r ) : ld16( pc + ( ptrdiff_t ) r * curStride );
        d[4 * r] = pkSub( x.x, z.x ); d[4 * r + 1] = pkSub( x.y, z.y ); d[4 * r + 2] = pkSub( x.z, z.z ); d[4 * r + 3] = pkSub( x.w, z.w );
      }
    }
    // Walsh-Hadamard stages over the dword index (both halves in parallel): five packed (NARROW) or four packed + one in 32 bits (WIDE)
#pragma unroll
    for( int len = 1; len < ( WIDE ? 16 : 32 ); len <<= 1 )
#pragma unroll
      for( int i = 0; i < 32; i += 2 * len )
#pragma unroll
        for( int j = i; j < i + len; j++ ) { const uint32_t a = d[j], b = d[j + len]; d[j] = pkAdd( a, b ); d[j + len] = pkSub( a, b ); }
    uint32_t s;
    if( WIDE )
    {
      // fifth stage (dwords j, j + 16) in 32 bits; sixth stage + magnitudes through 2 max( |lo|, |hi| ); the pair holding the DC is dword 0's sum
      uint32_t m = 0, dcTerm = 0;
#pragma unroll
      for( int j = 0; j < 16; j++ )
      {
        const int al = ( int ) ( int16_t ) ( d[j] & 0xffffu ), ah = ( int ) d[j] >> 16, bl = ( int ) ( int16_t ) ( d[j + 16] & 0xffffu ), bh = ( int ) d[j + 16] >> 16;
        const int pl = al + bl, ph = ah + bh, ml = al - bl, mh = ah - bh;
        const uint32_t aml = ( uint32_t ) abs( ml ), amh = ( uint32_t ) abs( mh );
        m += aml > amh ? aml : amh;
        if( j == 0 ) { const uint32_t dc = ( uint32_t ) abs( pl + ph ); dcTerm = ( uint32_t ) abs( pl - ph ) + ( dc >> 2 ); }
        else { const uint32_t apl = ( uint32_t ) abs( pl ), aph = ( uint32_t ) abs( ph ); m += apl > aph ? apl : aph; }
      }
      s = 2 * m + dcTerm;
    }
    else
    {
      // sixth stage + sum of magnitudes: |a + b| + |a - b| = 2 max( |a|, |b| ); dword 0 holds the DC (a + b), which counts a quarter
      uint32_t m = 0;
#pragma unroll
      for( int i = 1; i < 32; i++ ) { const uint32_t ax = pkAbs( d[i] ); const uint32_t lo = ax & 0xffffu, hi = ax >> 16; m += lo > hi ? lo : hi; }
      const int a0 = ( int ) ( int16_t ) ( d[0] & 0xffffu ), b0 = ( int ) d[0] >> 16;
      const uint32_t dc = ( uint32_t ) abs( a0 + b0 );
      s = 2 * m + ( uint32_t ) abs( a0 - b0 ) + ( dc >> 2 );
    }
    sum += FAST16 ? ( ( s + 2 ) >> 2 ) << 2 : ( s + 2 ) >> 2;                                 // RdCost.cpp:1220-1222 / :1319
  }
  const uint64_t tot = vvhipGroupSum64( sum, lpc, threadIdx.x & 63 );
  if( valid && lt == 0 ) out[cand] = tot;
}

template<bool WIDE>
__global__ void __launch_bounds__( 256 )
hadTile8PkMultiKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride, DistMultiJobs jobs )
{
  int k = 0;
#pragma unroll
  for( int i = 1; i < 8; i++ ) if( i < jobs.nJobs && ( int ) blockIdx.x >= jobs.j[i].blockStart ) k = i;
  const DistJobGeom& g = jobs.j[k];
  int blk = blockIdx.x - g.blockStart;
  if( jobs.xcdRemap ) blk = xcdBand( blk, g.nBlocks );
  // (the TILED instantiations of the body are not dispatched: measured, the Hadamard lists lose on the tiled copies — the funnel shifts cost the per-lane-tile kernel more
  //  than the saved line fills bring, 34.7 -> 69 us with every size tiled, 44 us with the 8x8 candidates alone — and merely compiling them into this kernel costs its other
  //  jobs registers: 34.7 -> 41 us)
  if( g.fast16 ) hadTilePkBody<true, WIDE>( blk, org, orgStride, cur, curStride, g.tilesX, g.tilesPerCand, g.log2Lpc, g.items, g.n, g.out, nullptr, g.shift ? jobs.curShift : 0 );
  else           hadTilePkBody<false, WIDE>( blk, org, orgStride, cur, curStride, g.tilesX, g.tilesPerCand, g.log2Lpc, g.items, g.n, g.out, nullptr, g.shift ? jobs.curShift : 0 );
}

template<int TW, int TH, bool FAST16>
__global__ void __launch_bounds__( 256 )
hadTileKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride,
               int tilesX, int tilesPerCand, int log2Lpc,
               const vvhip_dist_item* __restrict__ items, int n, uint64_t* __restrict__ out )
{
  hadTileBody<TW, TH, FAST16>( blockIdx.x, org, orgStride, cur, curStride, tilesX, tilesPerCand, log2Lpc, items, n, out );
}

__global__ void __launch_bounds__( 256 )
hadTile8MultiKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride, DistMultiJobs jobs )
{
  int k = 0;
#pragma unroll
  for( int i = 1; i < 8; i++ ) if( i < jobs.nJobs && ( int ) blockIdx.x >= jobs.j[i].blockStart ) k = i;
  const DistJobGeom& g = jobs.j[k];
  int blk = blockIdx.x - g.blockStart;
  if( jobs.xcdRemap ) blk = xcdBand( blk, g.nBlocks );
  if( g.fast16 ) hadTileBody<8, 8, true>( blk, org, orgStride, cur, curStride, g.tilesX, g.tilesPerCand, g.log2Lpc, g.items, g.n, g.out, g.shift ? jobs.curShift : 0 );
  else           hadTileBody<8, 8, false>( blk, org, orgStride, cur, curStride, g.tilesX, g.tilesPerCand, g.log2Lpc, g.items, g.n, g.out, g.shift ? jobs.curShift : 0 );
}

// ---------------------------------------------------------------------------------------------
// GEO masked SAD (RdCost.cpp:2062-2093) and fixed-weight SSE (RdCost.cpp:1948-1982): one wave per candidate, lanes stride the
// (processed row, column) pairs.  Both are off the BASELINE presets' path (GEO off, luma-level dQP off): correctness first.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__( 256 )
sadMaskKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride,
               const int16_t* __restrict__ mask, int maskRowAdvance, int stepX, int w, int rowsEff, int subShift,
               const vvhip_dist_item* __restrict__ items, const int32_t* __restrict__ maskOff, int n, uint64_t* __restrict__ out )
{
  const int lane = threadIdx.x & 63, cand = blockIdx.x * 4 + ( threadIdx.x >> 6 );
  if( cand >= n ) return;
  const int16_t* po = org + items[cand].org_off;
  const int16_t* pc = cur + items[cand].cur_off;
  const int16_t* pm = mask + ( maskOff ? maskOff[cand] : 0 );
  const int step = 1 << subShift;
  unsigned long long sum = 0;
  for( int i = lane; i < rowsEff * w; i += 64 )
  {
    const int r = i / w, x = i - r * w;
    const int d = ( int ) po[( ptrdiff_t ) r * step * orgStride + x] - ( int ) pc[( ptrdiff_t ) r * step * curStride + x];
    sum += ( unsigned long long ) ( long long ) ( ( d < 0 ? -d : d ) * ( int ) pm[( ptrdiff_t ) r * maskRowAdvance + x * stepX] );   // int product, then widened (:2081)
  }
  // per-lane partial sums may be "negative" (sign-extended) with a negative mask: add the two 32-bit halves separately, mod 2^64
  const uint32_t lo = ( uint32_t ) sum, hi = ( uint32_t ) ( sum >> 32 );
  const unsigned long long sLo = vvhipGroupSum64( lo, 64, lane ), sHi = vvhipGroupSum64( hi, 64, lane );
  if( lane == 0 ) out[cand] = ( sLo + ( sHi << 32 ) ) << subShift;
}

__global__ void __launch_bounds__( 256 )
fixWeightedSseKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ cur, int curStride, int w, int h,
                      const vvhip_dist_item* __restrict__ items, const uint32_t* __restrict__ weights, int n, uint64_t* __restrict__ out )
{
  const int lane = threadIdx.x & 63, cand = blockIdx.x * 4 + ( threadIdx.x >> 6 );
  if( cand >= n ) return;
  const int16_t* po = org + items[cand].org_off;
  const int16_t* pc = cur + items[cand].cur_off;
  const long long wt = weights[cand];
  unsigned long long sum = 0;
  for( int i = lane; i < w * h; i += 64 )
  {
    const int r = i / w, x = i - r * w;
    const int d = ( int ) po[( ptrdiff_t ) r * orgStride + x] - ( int ) pc[( ptrdiff_t ) r * curStride + x];
    sum += ( unsigned long long ) ( long long ) ( int ) ( ( wt * ( d * d ) + ( 1 << 15 ) ) >> 16 );       // Intermediate_Int cast (:1950)
  }
  const uint32_t lo = ( uint32_t ) sum, hi = ( uint32_t ) ( sum >> 32 );
  const unsigned long long sLo = vvhipGroupSum64( lo, 64, lane ), sHi = vvhipGroupSum64( hi, 64, lane );
  if( lane == 0 ) out[cand] = sLo + ( sHi << 32 );          // width 1 counts its single column twice and halves (:1981): same value
}

// ---------------------------------------------------------------------------------------------
// SAD cost surface: one workgroup per block; the (w+2rx) x (h'+2ry) reference window is staged in
// LDS once (coalesced row reads), the block's original rows live in LDS too; each wave then walks
// displacements, lanes split the block's row segments.
// ---------------------------------------------------------------------------------------------
// LDS-tiled full-search SAD.  One workgroup per block:
//   * the reference window (w + 2*rx) x (h + 2*ry) is read from the picture ONCE (coalesced rows) into LDS, twice: copy A as is and copy B
//     shifted by one sample, so that any 2-sample pair (x + dx, x + dx + 1), x even, is an aligned 32-bit LDS word in A (dx even) or B (dx odd);
//   * the block's original rows are staged too and read back as wave-wide broadcasts;
//   * lanes own CANDIDATES (consecutive dx, 4 consecutive dy each): per original word a lane issues 4 window reads and 4 v_sad_u16, i.e. the
//     picture data is re-used (2rx+1)(2ry+1) times from LDS and never re-fetched.
//   * a block's displacement rows can be split over several workgroups (gridDim.y) so that big blocks still fill 256 CUs; every workgroup
//     stages only the window rows its displacement rows touch.  Samples are stored biased (x ^ 0x8000) so v_sad_u16 is exact for signed Pel.
template<int KDY>
__global__ void __launch_bounds__( 256 )
sadSurfaceKernel( const int16_t* __restrict__ org, int orgStride, const int16_t* __restrict__ ref, int refStride,
                  int w, int h, int subShift, int rx, int ry, int pitchDw, int copyBOffDw, int groupsPerSplit,
                  const int32_t* __restrict__ blkOrgOff, const int32_t* __restrict__ blkRefOff, uint32_t* __restrict__ out )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char smemRaw[];
  uint32_t* sWin = reinterpret_cast<uint32_t*>( smemRaw );                     // copy A at 0, copy B at copyBOffDw (dwords)
  const int step = 1 << subShift, rowsEff = h >> subShift, wDw = w >> 1;
  const int nx = 2 * rx + 1, ny = 2 * ry + 1, nyG = ( ny + KDY - 1 ) / KDY;
  const int g0 = blockIdx.y * groupsPerSplit, g1 = min( nyG, g0 + groupsPerSplit );   // displacement-row groups of this workgroup
  const int dyFirst = g0 * KDY;                                                          // first displacement row (0-based, = dy + ry)
  const int winW = w + 2 * rx, winRows = h + ( g1 - g0 ) * KDY;                          // rows [dyFirst, dyFirst + winRows) of the full window
  uint32_t* sOrg = sWin + 2 * copyBOffDw;
  uint16_t* a16 = reinterpret_cast<uint16_t*>( sWin );
  uint16_t* b16 = reinterpret_cast<uint16_t*>( sWin + copyBOffDw );
  uint16_t* o16 = reinterpret_cast<uint16_t*>( sOrg );

  const int b = blockIdx.x, tid = threadIdx.x;
  const int16_t* po = org + blkOrgOff[b];
  const int16_t* pr = ref + blkRefOff[b] + ( ptrdiff_t ) ( dyFirst - ry ) * refStride - rx;
  const int rowsAvail = h + 2 * ry - dyFirst;                                            // window rows that exist below dyFirst
  for( int i = tid; i < rowsEff * w; i += blockDim.x ) { const int r = i / w, x = i - r * w; o16[i] = ( uint16_t ) po[( ptrdiff_t ) ( r * step ) * orgStride + x] ^ 0x8000u; }
  const int pitch = 2 * pitchDw;
  for( int i = tid; i < winRows * ( winW + 1 ); i += blockDim.x )
  {
    const int r = i / ( winW + 1 ), x = i - r * ( winW + 1 );
    uint16_t v = 0;
    if( r < rowsAvail && x < winW ) v = ( uint16_t ) pr[( ptrdiff_t ) r * refStride + x] ^ 0x8000u;
    if( x < winW ) a16[r * pitch + x] = v;
    if( x > 0 ) b16[r * pitch + x - 1] = v;          // B[i] = A[i + 1]
  }
  __syncthreads();

  for( int m = tid; m < nx * ( g1 - g0 ); m += blockDim.x )
  {
    const int gl = m / nx, mx = m - gl * nx, my0 = ( g0 + gl ) * KDY;
    const uint32_t* base = sWin + ( ( mx & 1 ) ? copyBOffDw : 0 ) + ( mx >> 1 ) + gl * KDY * pitchDw;
    uint32_t acc[KDY];
#pragma unroll
    for( int k = 0; k < KDY; k++ ) acc[k] = 0;
    for( int r = 0; r < rowsEff; r++ )
    {
      const uint32_t* rowp = base + r * step * pitchDw;
      const uint32_t* op = sOrg + r * wDw;
      for( int xp = 0; xp < wDw; xp++ )
      {
        const uint32_t o = op[xp];
#pragma unroll
        for( int k = 0; k < KDY; k++ ) acc[k] = __builtin_amdgcn_sad_u16( rowp[k * pitchDw + xp], o, acc[k] );
      }
    }
#pragma unroll
    for( int k = 0; k < KDY; k++ )
      if( my0 + k < ny ) out[( size_t ) b * nx * ny + ( my0 + k ) * nx + mx] = acc[k] << subShift;
  }
}

__device__ __forceinline__ void shift1Unit( size_t i, const int16_t* __restrict__ src, size_t elems, int16_t* __restrict__ dst );
// dst[i] = src[i + 1] (the last element 0): thread = eight samples
__global__ void __launch_bounds__( 256 )
shift1Kernel( const int16_t* __restrict__ src, size_t elems, int16_t* __restrict__ dst )
{
  shift1Unit( ( ( size_t ) blockIdx.x * 256 + threadIdx.x ) * 8, src, elems, dst );
}
__device__ __forceinline__ void shift1Unit( size_t i, const int16_t* __restrict__ src, size_t elems, int16_t* __restrict__ dst )
{
  if( i >= elems ) return;
  if( i + 8 <= elems && !( ( reinterpret_cast<uintptr_t>( dst + i ) | reinterpret_cast<uintptr_t>( src + i ) ) & 15 ) )
  {
    // an aligned vector + the dword behind it, funnel-shifted by one sample (a 16-byte load at a 2-byte-aligned address would be split into dwords by the memory pipeline)
    const u32x4 v = *reinterpret_cast<const u32x4*>( src + i );
    uint32_t nx = 0;
    if( i + 10 <= elems ) nx = *reinterpret_cast<const uint32_t*>( src + i + 8 ); else if( i + 8 < elems ) nx = ( uint32_t ) ( uint16_t ) src[i + 8];
    u32x4 o;
    o.x = __builtin_amdgcn_alignbit( v.y, v.x, 16 ); o.y = __builtin_amdgcn_alignbit( v.z, v.y, 16 ); o.z = __builtin_amdgcn_alignbit( v.w, v.z, 16 ); o.w = __builtin_amdgcn_alignbit( nx, v.w, 16 );
    *reinterpret_cast<u32x4*>( dst + i ) = o;
    return;
  }
  for( size_t k = i; k < elems && k < i + 8; k++ ) dst[k] = k + 1 < elems ? src[k + 1] : ( int16_t ) 0;
}

// row-major padded plane -> 8x8-tiled copy.  A workgroup takes a band of 8 rows x 64 tiles: coalesced row reads (64 x 16 bytes per row), transposed through LDS (pitch 9 tile rows:
// conflict-free), coalesced tile writes (64 x 128 bytes).  Grid = ceil(rows / 8) x ceil(tpr / 64) workgroups.
constexpr int TILE_BAND = 64;
__device__ __forceinline__ void tile8Band( int blk, const int16_t* __restrict__ src, int stride, int rows, int tpr, int16_t* __restrict__ dst, u32x4* __restrict__ sT )
{
  const int groups = ( tpr + TILE_BAND - 1 ) / TILE_BAND, band = blk / groups, tx0 = ( blk - band * groups ) * TILE_BAND, t = threadIdx.x;
#pragma unroll
  for( int k = 0; k < 2; k++ )
  {
    const int p = t + 256 * k, r = p >> 6, tx = p & 63, y = band * 8 + r, X = tx0 + tx;
    u32x4 v = { 0, 0, 0, 0 };
    if( X < tpr && y < rows )
    {
      const int16_t* q = src + ( size_t ) y * stride + 8 * X;
      if( 8 * X + 8 <= stride ) v = ld16( q );
      else { int16_t tmp[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; for( int j = 0; 8 * X + j < stride; j++ ) tmp[j] = q[j]; v = *reinterpret_cast<const u32x4*>( tmp ); }
    }
    sT[tx * 9 + r] = v;
  }
  __syncthreads();
#pragma unroll
  for( int k = 0; k < 2; k++ )
  {
    const int q = t + 256 * k, tile = q >> 3, r = q & 7, X = tx0 + tile;
    if( X < tpr ) *reinterpret_cast<u32x4*>( dst + ( ( size_t ) band * tpr + X ) * 64 + r * 8 ) = sT[tile * 9 + r];
  }
}
__device__ __forceinline__ void shift1Unit( size_t i, const int16_t* __restrict__ src, size_t elems, int16_t* __restrict__ dst );

__global__ void __launch_bounds__( 256 )
tile8Kernel( const int16_t* __restrict__ src, int stride, int rows, int tpr, int16_t* __restrict__ dst )
{
  __shared__ u32x4 sT[TILE_BAND * 9];
  tile8Band( blockIdx.x, src, stride, rows, tpr, dst, sT );
}

// every copy the library derives from a picture's planes in ONE launch: block ranges [0, b0) tile the original plane, [b0, b1) tile the reference plane, [b1, ..) shift it
struct DeriveArgs { const int16_t* org; int orgStride, orgRows, orgTpr; int16_t* orgTiled; const int16_t* cur; int curStride, curRows, curTpr; int16_t* curTiled; int16_t* curShift; size_t curElems; int b0, b1; };
__global__ void __launch_bounds__( 256 )
deriveKernel( DeriveArgs a )
{
  __shared__ u32x4 sT[TILE_BAND * 9];
  const int b = blockIdx.x;
  if( b < a.b0 )      tile8Band( b, a.org, a.orgStride, a.orgRows, a.orgTpr, a.orgTiled, sT );
  else if( b < a.b1 ) tile8Band( b - a.b0, a.cur, a.curStride, a.curRows, a.curTpr, a.curTiled, sT );
  else                shift1Unit( ( ( size_t ) ( b - a.b1 ) * 256 + threadIdx.x ) * 8, a.cur, a.curElems, a.curShift );
}

int pow2Floor( int v ) { int p = 1; while( p * 2 <= v ) p <<= 1; return p; }

template<int MODE>
int launchSadSse( vvhip_ctx* ctx, const int16_t* d_org, int os, const int16_t* d_cur, int cs, int w, int h, int subShift,
                  const vvhip_dist_item* items, int n, int calcCentre, uint64_t* out )
{
  const int rowsEff = h >> subShift;
  const int CH = ( w % 8 == 0 ) ? 8 : ( w % 4 == 0 ) ? 4 : 2;
  const int lpr = w / CH;
  int lpc = pow2Floor( lpr * rowsEff );
  if( lpc > 64 ) lpc = 64;
  const int log2Lpc = ilog2i( lpc );
  const long teams = ( long ) n * ( MODE == MODE_SAD_X5 ? 5 : 1 );
  const long threads = teams * lpc;
  const int block = 256;
  const unsigned grid = ( unsigned ) ( ( threads + block - 1 ) / block );
  if( grid == 0 ) return VVHIP_OK;
#define LAUNCH( C ) hipLaunchKernelGGL( ( sadSseKernel<C, MODE> ), dim3( grid ), dim3( block ), 0, ctx->stream, \
                                        d_org, os, d_cur, cs, lpr, isPow2( lpr ) ? ilog2i( lpr ) : -1, rowsEff, subShift, log2Lpc, items, n, calcCentre, out )
  if( CH == 8 ) LAUNCH( 8 ); else if( CH == 4 ) LAUNCH( 4 ); else LAUNCH( 2 );
#undef LAUNCH
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

template<int TW, int TH, bool FAST16>
int launchHad( vvhip_ctx* ctx, const int16_t* d_org, int os, const int16_t* d_cur, int cs, int w, int h,
               const vvhip_dist_item* items, int n, uint64_t* out )
{
  constexpr int PX = FAST16 ? 16 : TW, PY = FAST16 ? 16 : TH;
  const int tilesX = w / PX, tiles = tilesX * ( h / PY );
  int lpc = TH * pow2Floor( tiles );      // tiles per candidate is a power of two for power-of-two blocks
  if( lpc > 64 ) lpc = 64;
  if( tiles % ( lpc / TH ) != 0 ) lpc = TH;   // odd tile counts (non power-of-two blocks): one tile team walks all tiles
  const int log2Lpc = ilog2i( lpc );
  const long threads = ( long ) n * lpc;
  const int block = 256;
  const unsigned grid = ( unsigned ) ( ( threads + block - 1 ) / block );
  if( grid == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( ( hadKernel<TW, TH, FAST16> ), dim3( grid ), dim3( block ), 0, ctx->stream,
                      d_org, os, d_cur, cs, tilesX, tiles, log2Lpc, items, n, out );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

template<int TW, int TH, bool FAST16>
int launchHadTile( vvhip_ctx* ctx, const int16_t* d_org, int os, const int16_t* d_cur, int cs, int w, int h,
                   const vvhip_dist_item* items, int n, uint64_t* out )
{
  constexpr int PX = FAST16 ? 16 : TW, PY = FAST16 ? 16 : TH;
  const int tilesX = w / PX, tiles = tilesX * ( h / PY );
  int lpc = pow2Floor( tiles );
  if( lpc > 64 ) lpc = 64;
  if( tiles % lpc != 0 ) lpc = 1;
  const int log2Lpc = ilog2i( lpc );
  const long threads = ( long ) n * lpc;
  const int block = 256;
  const unsigned grid = ( unsigned ) ( ( threads + block - 1 ) / block );
  if( grid == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( ( hadTileKernel<TW, TH, FAST16> ), dim3( grid ), dim3( block ), 0, ctx->stream,
                      d_org, os, d_cur, cs, tilesX, tiles, log2Lpc, items, n, out );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

// tile-selection ladder of xGetHADs<fastHad>, RdCost.cpp:1836-1935
int launchHadLadder( vvhip_ctx* ctx, bool fast, const int16_t* d_org, int os, const int16_t* d_cur, int cs, int w, int h,
                     const vvhip_dist_item* items, int n, uint64_t* out )
{
  if( w > h && ( h & 7 ) == 0 && ( w & 15 ) == 0 )      return launchHad<16, 8, false>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  else if( w < h && ( w & 7 ) == 0 && ( h & 15 ) == 0 ) return launchHad<8, 16, false>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  else if( w > h && ( h & 3 ) == 0 && ( w & 7 ) == 0 )  return launchHadTile<8, 4, false>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  else if( w < h && ( w & 3 ) == 0 && ( h & 7 ) == 0 )  return launchHadTile<4, 8, false>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  else if( fast && h % 32 == 0 && w % 32 == 0 && h == w ) return launchHadTile<8, 8, true>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  else if( h % 8 == 0 && w % 8 == 0 )                   return launchHadTile<8, 8, false>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  else if( h % 4 == 0 && w % 4 == 0 )                   return launchHadTile<4, 4, false>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  else if( h % 2 == 0 && w % 2 == 0 )                   return launchHadTile<2, 2, false>( ctx, d_org, os, d_cur, cs, w, h, items, n, out );
  return vvhip_fail( ctx, VVHIP_E_ARG, "Hadamard: invalid size %dx%d (reference THROWs \"Invalid size\", RdCost.cpp:1934)", w, h );
}

} // namespace

extern "C" {

int vvhip_dist_batch( vvhip_ctx* ctx, int func, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride,
                      int width, int height, int sub_shift, int bit_depth, const vvhip_dist_item* d_items, int n, uint64_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n < 0 || width < 1 || height < 1 || width > 128 || height > 128 || sub_shift < 0 || sub_shift > 1 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dist_batch: bad geometry %dx%d subShift %d n %d", width, height, sub_shift, n );
  if( bit_depth > 10 && bit_depth != 12 ) { /* any depth works: arithmetic is exact for all int16 inputs */ }
  if( n == 0 ) return VVHIP_OK;
  if( !d_org || !d_cur || !d_items || !d_out ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dist_batch: NULL pointer" );
  switch( func )
  {
  case VVHIP_DF_SAD:
    if( ( width & 1 ) || ( height >> sub_shift ) < 1 || ( height & ( ( 1 << sub_shift ) - 1 ) ) )
      return vvhip_fail( ctx, VVHIP_E_UNSUPPORTED, "SAD: width must be even and height a multiple of 1<<subShift (%dx%d)", width, height );
    return launchSadSse<MODE_SAD>( ctx, d_org, org_stride, d_cur, cur_stride, width, height, sub_shift, d_items, n, 0, d_out );
  case VVHIP_DF_SSE:
    if( width & 1 ) return vvhip_fail( ctx, VVHIP_E_UNSUPPORTED, "SSE: width must be even (%d)", width );
    return launchSadSse<MODE_SSE>( ctx, d_org, org_stride, d_cur, cur_stride, width, height, 0, d_items, n, 0, d_out );
  case VVHIP_DF_HAD:
    return launchHadLadder( ctx, false, d_org, org_stride, d_cur, cur_stride, width, height, d_items, n, d_out );
  case VVHIP_DF_HAD_FAST:
    return launchHadLadder( ctx, true, d_org, org_stride, d_cur, cur_stride, width, height, d_items, n, d_out );
  case VVHIP_DF_HAD_2SAD:
  {
    int rc = launchHadLadder( ctx, false, d_org, org_stride, d_cur, cur_stride, width, height, d_items, n, d_out );
    if( rc ) return rc;
    return launchSadSse<MODE_SAD_MIN2>( ctx, d_org, org_stride, d_cur, cur_stride, width, height, 0, d_items, n, 0, d_out );
  }
  default:
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dist_batch: unknown function %d", func );
  }
}

// jobs with their own function each: consecutive jobs of the same kernel family (SAD/SSE, or HAD/HAD_fast) share a launch
static int distMultiFunc( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride, int bit_depth, const vvhip_dist_fjob* jobs, int n_jobs,
                          const vvhip_tiled_planes* tiled = nullptr )
{
  // mergeable: SAD / SSE with width % 8 == 0, Hadamard whose ladder ends on the 8x8 or 16x16_fast tile; everything else runs as separate launches
  auto family = [&]( const vvhip_dist_fjob& jb ) -> int {
    if( jb.n <= 0 || jb.width < 8 || jb.height < 1 || jb.width > 128 || jb.height > 128 ) return 0;
    if( jb.func == VVHIP_DF_SAD ) return ( ( jb.width & 7 ) == 0 && jb.sub_shift >= 0 && jb.sub_shift <= 1 && ( jb.height >> jb.sub_shift ) >= 1 && !( jb.height & ( ( 1 << jb.sub_shift ) - 1 ) ) ) ? 1 : 0;
    if( jb.func == VVHIP_DF_SSE ) return ( jb.width & 7 ) == 0 ? 1 : 0;
    if( jb.func == VVHIP_DF_HAD || jb.func == VVHIP_DF_HAD_FAST ) return ( jb.width == jb.height && ( jb.width & 7 ) == 0 ) ? ( ( jb.flags & VVHIP_DIST_FLAG_SAMPLES ) ? 3 : 2 ) : 0;
    return 0; };
  int i = 0;
  while( i < n_jobs )
  {
    const int fam = family( jobs[i] );
    if( !fam )
    {
      if( jobs[i].n > 0 )
      {
        const int rc = vvhip_dist_batch( ctx, jobs[i].func, d_org, org_stride, d_cur, cur_stride, jobs[i].width, jobs[i].height, jobs[i].sub_shift, bit_depth, jobs[i].d_items, jobs[i].n, jobs[i].d_out );
        if( rc ) return rc;
      }
      i++;
      continue;
    }
    DistMultiJobs mj; mj.nJobs = 0; mj.T = {};
    // one-sample-shifted copy of the reference plane (see shiftSel): both buffers must share their alignment modulo 4 bytes
    static const int useShift = []{ const char* e = getenv( "VVHIP_SHIFT1" ); return e ? atoi( e ) : 1; }();
    mj.curShift = ( useShift && tiled && tiled->d_cur_shift1 && !( ( reinterpret_cast<uintptr_t>( d_cur ) ^ reinterpret_cast<uintptr_t>( tiled->d_cur_shift1 ) ) & 3 ) )
                  ? ( tiled->d_cur_shift1 - d_cur ) - 1 : 0;
    static const int useTiled = []{ const char* e = getenv( "VVHIP_TILED" ); return e ? atoi( e ) : 1; }();
    const bool haveTiled = tiled && useTiled && tiled->d_org_tiled && tiled->d_cur_tiled && bit_depth <= 10 && org_stride < 65536 && cur_stride < 65536;
    if( haveTiled )
    {
      mj.T.org = tiled->d_org_tiled; mj.T.cur = tiled->d_cur_tiled;
      mj.T.orgStride = org_stride; mj.T.curStride = cur_stride;
      mj.T.orgTpr = ( org_stride + 7 ) / 8; mj.T.curTpr = ( cur_stride + 7 ) / 8;
      mj.T.orgBias = tiled->org_margin * org_stride + tiled->org_margin; mj.T.curBias = tiled->cur_margin * cur_stride + tiled->cur_margin;
      mj.T.orgMagic = ( 1ull << 40 ) / ( unsigned ) org_stride + 1; mj.T.curMagic = ( 1ull << 40 ) / ( unsigned ) cur_stride + 1;
    }
    static const int xcdRemapEnv = []{ const char* e = getenv( "VVHIP_XCD_REMAP" ); return e ? atoi( e ) : 1; }();
    mj.xcdRemap = xcdRemapEnv;
    long blocks = 0;
    bool anySad = false, anySse = false;
    // the jobs of a launch, heaviest workgroups first: workgroups start in grid order, so the long ones (64x64 SSE: eight row pieces per lane) must not be the
    // last to start while the one-piece-per-lane 8x8 lists have long drained
    int order[8], nOrder = 0;
    while( i < n_jobs && nOrder < 8 && family( jobs[i] ) == fam ) order[nOrder++] = i++;
    static const int sortJobs = []{ const char* e = getenv( "VVHIP_DIST_SORT" ); return e ? atoi( e ) : 1; }();
    auto weight = [&]( const vvhip_dist_fjob& jb ) -> long {
      if( fam == 1 ) { const int rows = jb.func == VVHIP_DF_SSE ? jb.height : jb.height >> jb.sub_shift; int l = pow2Floor( jb.width / 8 * rows ); if( l > 64 ) l = 64; return ( long ) jb.width / 8 * rows / l; }
      const bool f16 = jb.func == VVHIP_DF_HAD_FAST && jb.width % 32 == 0;
      const int px = f16 ? 16 : 8, tpc = ( jb.width / px ) * ( jb.height / px ); int l = pow2Floor( tpc ); if( l > 64 ) l = 64; if( tpc % l ) l = 1;
      return ( long ) tpc / l * ( f16 ? 4 : 1 ); };
    if( sortJobs ) std::stable_sort( order, order + nOrder, [&]( int a, int b ) { return weight( jobs[a] ) > weight( jobs[b] ); } );
    for( int oi = 0; oi < nOrder; oi++ )
    {
      const vvhip_dist_fjob& jb = jobs[order[oi]];
      DistJobGeom& g = mj.j[mj.nJobs];
      g.items = jb.d_items; g.out = jb.d_out; g.n = jb.n; g.blockStart = ( int ) blocks; g.fast16 = 0; g.tilesX = 0; g.tilesPerCand = 0;
      g.lpr = 0; g.lprShift = 0; g.rowsEff = 0; g.subShift = 0; g.sse = 0; g.tiled = 0;
      // the shifted copy pays where a lane's 16 bytes are a granule access of their own: every Hadamard tile row, SAD rows of <= 16 samples.  Wider rows are several
      // adjacent lanes that coalesce either way, and splitting a block's candidates over two copies of its window only costs L1 hits there (measured: 32-wide SAD and
      // the SSE lists lose 10-30 %, 16-wide SAD gains 25 %, the Hadamard lists 25-40 %)
      g.shift = ( fam >= 2 || ( jb.func == VVHIP_DF_SAD && jb.width <= 16 ) ) ? 1 : 0;
      int lpc;
      bool tiledSad = false;
      if( fam == 1 )
      {
        g.sse = jb.func == VVHIP_DF_SSE ? ( ( ( jb.flags & VVHIP_DIST_FLAG_SAMPLES ) && bit_depth <= 12 ) ? 2 : 1 ) : 0; ( g.sse ? anySse : anySad ) = true;
        g.subShift = g.sse ? 0 : jb.sub_shift;
        g.rowsEff = jb.height >> g.subShift; g.lpr = jb.width / 8; g.lprShift = isPow2( g.lpr ) ? ilog2i( g.lpr ) : -1;
        lpc = pow2Floor( g.lpr * g.rowsEff ); if( lpc > 64 ) lpc = 64;
        // 8x8 lists (every row of the block is read): the tiled copies, 8 lanes per candidate
        if( haveTiled && jb.width == 8 && jb.height == 8 && g.subShift == 0 ) { g.tiled = 1; tiledSad = true; lpc = 8; }
      }
      else
      {
        g.tiled = 0;      // Hadamard lists stay on the row-major planes (see hadTile8PkMultiKernel)
        g.fast16 = ( jb.func == VVHIP_DF_HAD_FAST && jb.width % 32 == 0 ) ? 1 : 0;
        const int px = g.fast16 ? 16 : 8;
        g.tilesX = jb.width / px; g.tilesPerCand = g.tilesX * ( jb.height / px );
        lpc = pow2Floor( g.tilesPerCand ); if( lpc > 64 ) lpc = 64;
        if( g.tilesPerCand % lpc ) lpc = 1;
      }
      g.log2Lpc = ilog2i( lpc );
      // workgroup size: lane teams never talk to each other, so the Hadamard launches use single-wave workgroups (a wave's registers are free again
      // the moment it retires: 34.6 -> 31.3 us); the SAD / SSE launch is indifferent (34.5 / 35.2 / 35.0 us for 256 / 128 / 64) and keeps 256
      static const int wgEnv = []{ const char* e = getenv( "VVHIP_DIST_WG" ); const int v = e ? atoi( e ) : 0; return ( v == 64 || v == 128 || v == 256 ) ? v : 0; }();
      const int wgSize = wgEnv ? wgEnv : ( fam >= 2 ? 64 : 256 );
      g.nBlocks = ( fam == 1 && !tiledSad ) ? ( int ) ( ( ( ( long ) jb.n + DIST_U - 1 ) / DIST_U * lpc + wgSize - 1 ) / wgSize ) : ( int ) ( ( ( long ) jb.n * lpc + wgSize - 1 ) / wgSize );
      if( tiledSad ) g.nBlocks = ( jb.n + wgSize - 1 ) / wgSize;            // the tiled body: one candidate per lane (sadSse8TiledBody)
      blocks += g.nBlocks;
      mj.nJobs++;
    }
    static const int wgEnvL = []{ const char* e = getenv( "VVHIP_DIST_WG" ); const int v = e ? atoi( e ) : 0; return ( v == 64 || v == 128 || v == 256 ) ? v : 0; }();
    const int wgSizeL = wgEnvL ? wgEnvL : ( fam >= 2 ? 64 : 256 );
    // bit depths <= 10: the packed 16-bit tile (the reference's x86 rows have the same limit); VVHIP_HAD_PK=0 forces the 32-bit form
    static const int hadPk = []{ const char* e = getenv( "VVHIP_HAD_PK" ); return e ? atoi( e ) : 1; }();
    if( fam == 3 && bit_depth <= 10 && hadPk )      hipLaunchKernelGGL( hadTile8PkMultiKernel<false>, dim3( ( unsigned ) blocks ), dim3( wgSizeL ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, mj );
    else if( fam >= 2 && bit_depth <= 10 && hadPk ) hipLaunchKernelGGL( hadTile8PkMultiKernel<true>, dim3( ( unsigned ) blocks ), dim3( wgSizeL ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, mj );
    else if( fam >= 2 )       hipLaunchKernelGGL( hadTile8MultiKernel, dim3( ( unsigned ) blocks ), dim3( wgSizeL ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, mj );
    else if( anySad && anySse ) hipLaunchKernelGGL( sadSseMixedKernel, dim3( ( unsigned ) blocks ), dim3( wgSizeL ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, mj );
    else if( anySad )         hipLaunchKernelGGL( ( sadSseMultiKernel<MODE_SAD> ), dim3( ( unsigned ) blocks ), dim3( wgSizeL ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, mj );
    else                      hipLaunchKernelGGL( ( sadSseMultiKernel<MODE_SSE> ), dim3( ( unsigned ) blocks ), dim3( wgSizeL ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, mj );
    VVHIP_LAUNCH_CHECK( ctx );
  }
  return VVHIP_OK;
}

int vvhip_dist_multi_func( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride, int bit_depth, const vvhip_dist_fjob* jobs, int n_jobs )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n_jobs < 0 || ( n_jobs && !jobs ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dist_multi_func: bad job list" );
  return distMultiFunc( ctx, d_org, org_stride, d_cur, cur_stride, bit_depth, jobs, n_jobs );
}

int vvhip_dist_multi_func_tiled( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride, const vvhip_tiled_planes* tiled, int bit_depth,
                                 const vvhip_dist_fjob* jobs, int n_jobs )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n_jobs < 0 || ( n_jobs && !jobs ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dist_multi_func_tiled: bad job list" );
  if( tiled && ( tiled->org_margin < 0 || tiled->cur_margin < 0 || org_stride < 8 || cur_stride < 8 ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dist_multi_func_tiled: bad tiled-plane geometry" );
  return distMultiFunc( ctx, d_org, org_stride, d_cur, cur_stride, bit_depth, jobs, n_jobs, tiled );
}

static long tile8Blocks( int rows, int tpr ) { return ( long ) ( ( rows + 7 ) / 8 ) * ( ( tpr + TILE_BAND - 1 ) / TILE_BAND ); }

int vvhip_planes_derive( vvhip_ctx* ctx, const int16_t* d_org_base, int org_stride, int org_rows, int16_t* d_org_tiled,
                         const int16_t* d_cur_base, int cur_stride, int cur_rows, int16_t* d_cur_tiled, int16_t* d_cur_shift1 )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( ( d_org_tiled && ( !d_org_base || org_stride < 8 || org_rows < 1 ) ) || ( ( d_cur_tiled || d_cur_shift1 ) && ( !d_cur_base || cur_stride < 8 || cur_rows < 1 ) ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_planes_derive: bad arguments" );
  DeriveArgs a = {};
  a.org = d_org_base; a.orgStride = org_stride; a.orgRows = org_rows; a.orgTpr = ( org_stride + 7 ) / 8; a.orgTiled = d_org_tiled;
  a.cur = d_cur_base; a.curStride = cur_stride; a.curRows = cur_rows; a.curTpr = ( cur_stride + 7 ) / 8; a.curTiled = d_cur_tiled; a.curShift = d_cur_shift1;
  a.curElems = ( size_t ) cur_stride * cur_rows;
  const long n0 = d_org_tiled ? tile8Blocks( org_rows, a.orgTpr ) : 0, n1 = d_cur_tiled ? tile8Blocks( cur_rows, a.curTpr ) : 0;
  const long n2 = d_cur_shift1 ? ( long ) ( ( a.curElems + 7 ) / 8 ) : 0;
  a.b0 = ( int ) n0; a.b1 = a.b0 + ( int ) n1;
  const long blocks = a.b1 + ( n2 + 255 ) / 256;
  if( !blocks ) return VVHIP_OK;
  hipLaunchKernelGGL( deriveKernel, dim3( ( unsigned ) blocks ), dim3( 256 ), 0, ctx->stream, a );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_plane_shift1( vvhip_ctx* ctx, const int16_t* d_src, size_t elems, int16_t* d_dst )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_src || !d_dst || elems < 1 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_plane_shift1: bad arguments" );
  hipLaunchKernelGGL( shift1Kernel, dim3( ( unsigned ) ( ( elems + 8 * 256 - 1 ) / ( 8 * 256 ) ) ), dim3( 256 ), 0, ctx->stream, d_src, elems, d_dst );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

size_t vvhip_tiled8_elems( int stride, int rows )
{
  return ( size_t ) ( ( rows + 7 ) / 8 ) * ( ( stride + 7 ) / 8 ) * 64 + 128;      // (+ two tiles of slack: the funnel shift may touch the tile after the last one)
}

int vvhip_plane_tile8( vvhip_ctx* ctx, const int16_t* d_base, int stride, int rows, int16_t* d_tiled )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_base || !d_tiled || stride < 8 || rows < 1 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_plane_tile8: bad arguments" );
  const int tpr = ( stride + 7 ) / 8;
  hipLaunchKernelGGL( tile8Kernel, dim3( ( unsigned ) tile8Blocks( rows, tpr ) ), dim3( 256 ), 0, ctx->stream, d_base, stride, rows, tpr, d_tiled );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_dist_multi( vvhip_ctx* ctx, int func, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride, int bit_depth,
                      const vvhip_dist_job* jobs, int n_jobs )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n_jobs < 0 || ( n_jobs && !jobs ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_dist_multi: bad job list" );
  std::vector<vvhip_dist_fjob> fj( n_jobs );
  for( int i = 0; i < n_jobs; i++ ) { fj[i].func = func; fj[i].width = jobs[i].width; fj[i].height = jobs[i].height; fj[i].sub_shift = jobs[i].sub_shift; fj[i].n = jobs[i].n; fj[i].flags = 0;
                                      fj[i].d_items = jobs[i].d_items; fj[i].d_out = jobs[i].d_out; }
  return distMultiFunc( ctx, d_org, org_stride, d_cur, cur_stride, bit_depth, fj.data(), n_jobs );
}

int vvhip_sad_x5_batch( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride,
                        int width, int height, int sub_shift, int calc_centre, const vvhip_dist_item* d_items, int n, uint64_t* d_out5 )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( ( width != 8 && width != 16 ) || height < 1 || height > 128 || sub_shift < 0 || sub_shift > 1 || n < 0 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_sad_x5_batch: width must be 8 or 16 (m_afpDistortFuncX5[log2w-3], RdCost.cpp:252), got %dx%d", width, height );
  if( n == 0 ) return VVHIP_OK;
  return launchSadSse<MODE_SAD_X5>( ctx, d_org, org_stride, d_cur, cur_stride, width, height, sub_shift, d_items, n, calc_centre, d_out5 );
}

int vvhip_sad_mask_batch( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride,
                          const int16_t* d_mask, int mask_stride, int step_x, int mask_stride2,
                          int width, int height, int sub_shift, int bit_depth,
                          const vvhip_dist_item* d_items, const int32_t* d_mask_off, int n, uint64_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_org || !d_cur || !d_mask || width < 1 || height < 1 || width > 128 || height > 128 || sub_shift < 0 || sub_shift > 1 || n < 0 ||
      bit_depth < 8 || bit_depth > 16 || ( height & ( ( 1 << sub_shift ) - 1 ) ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_sad_mask_batch: bad arguments (%dx%d subShift %d)", width, height, sub_shift );
  if( n == 0 ) return VVHIP_OK;
  // the mask pointer advances by stepX per sample, then by maskStride*step + maskStride2 per processed row (RdCost.cpp:2079-2088)
  const int rowAdvance = width * step_x + mask_stride * ( 1 << sub_shift ) + mask_stride2;
  hipLaunchKernelGGL( sadMaskKernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, d_mask, rowAdvance, step_x,
                      width, height >> sub_shift, sub_shift, d_items, d_mask_off, n, d_out );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_fix_weighted_sse_batch( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride,
                                  int width, int height, int bit_depth, const vvhip_dist_item* d_items, const uint32_t* d_weights, int n, uint64_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !d_org || !d_cur || !d_weights || width < 1 || height < 1 || width > 128 || height > 128 || n < 0 || bit_depth < 8 || bit_depth > 16 || ( ( width & 1 ) && width != 1 ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_fix_weighted_sse_batch: width must be even or 1 (RdCost.cpp:1967), got %dx%d", width, height );
  if( n == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( fixWeightedSseKernel, dim3( ( n + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, d_org, org_stride, d_cur, cur_stride, width, height, d_items, d_weights, n, d_out );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_sad_surface( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_ref, int ref_stride,
                       int width, int height, int sub_shift, int range_x, int range_y,
                       const int32_t* d_block_org_off, const int32_t* d_block_ref_off, int n_blocks, uint32_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width < 2 || height < 2 || width > 128 || height > 128 || ( width & 1 ) || sub_shift < 0 || sub_shift > 1 || range_x < 0 || range_y < 0 || n_blocks < 0 ||
      ( height & ( ( 1 << sub_shift ) - 1 ) ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_sad_surface: bad geometry" );
  if( n_blocks == 0 ) return VVHIP_OK;
  const int rowsEff = height >> sub_shift;
  const int nx = 2 * range_x + 1, ny = 2 * range_y + 1;
  // displacement rows per lane (power of two): minimise wave trips x (window reads + one org read)
  int kdy = 1, bestCost = 1 << 30;
  for( int k = 1; k <= 8; k *= 2 )
  {
    const int slots = nx * ( ( ny + k - 1 ) / k ), cost = ( ( slots + 63 ) / 64 ) * ( k + 1 );
    if( cost < bestCost ) { bestCost = cost; kdy = k; }
  }
  const int nyG = ( ny + kdy - 1 ) / kdy;
  int splits = ( 1024 + n_blocks - 1 ) / n_blocks; if( splits > nyG ) splits = nyG; if( splits < 1 ) splits = 1;
  const int groupsPerSplit = ( nyG + splits - 1 ) / splits;
  splits = ( nyG + groupsPerSplit - 1 ) / groupsPerSplit;
  const int winW = width + 2 * range_x, winRows = height + groupsPerSplit * kdy;
  int pitchDw = ( winW + 2 ) / 2; pitchDw |= 1;                                   // odd dword pitch: consecutive window rows start on different banks
  int copyB = pitchDw * winRows; copyB = ( ( copyB + 31 ) & ~31 ) + 16;           // copy B sits 16 banks away from copy A
  const size_t smem = ( size_t ) ( 2 * copyB + ( rowsEff * width ) / 2 + 8 ) * sizeof( uint32_t );
  if( smem > 160 * 1024 ) return vvhip_fail( ctx, VVHIP_E_UNSUPPORTED, "vvhip_sad_surface: window %dx%d needs %zu B of LDS (> 160 KiB)", winW, winRows, smem );
#define SURF( K ) { if( smem > 64 * 1024 ) VVHIP_CHECK_HIP( ctx, hipFuncSetAttribute( ( const void* ) sadSurfaceKernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, ( int ) smem ) ); \
    hipLaunchKernelGGL( ( sadSurfaceKernel<K> ), dim3( n_blocks, splits ), dim3( 256 ), smem, ctx->stream, d_org, org_stride, d_ref, ref_stride, width, height, sub_shift, \
                        range_x, range_y, pitchDw, copyB, groupsPerSplit, d_block_org_off, d_block_ref_off, d_out ); }
  switch( kdy ) { case 1: SURF( 1 ) break; case 2: SURF( 2 ) break; case 4: SURF( 4 ) break; default: SURF( 8 ) break; }
#undef SURF
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

} // extern "C"
