// mctf.hip — MCTF hierarchical block matching for gfx950.
//
// Reference semantics (CommonLib/MCTF.cpp):
//   motionErrorLumaInt :122-145, motionErrorLumaFrac6/4 :147-257 (clip after each pass), calcVarCore :520-546,
//   subsampleLuma :1072-1097, motionErrorLuma :1099-1164, estimateLumaLn :1166-1327 (search schedule, strict-< argmin in scan
//   order), motionEstimationLuma :1329-1397 (row tasks with above/left hand-off), motionEstimationMCTF :666-707 (pyramid).
//
// How the schedule is mapped to the GPU (per level, per (current, reference) pair):
//   phase A  meSearchKernel    one wavefront per block; the candidate list of estimateLumaLn up to (not including) the above/left
//                              tests is walked in the reference's order, each candidate's error is computed by the 64 lanes
//                              together (16-byte row segments, two-pass sub-pel filter through LDS).  All blocks independent.
//   phase B  the above/left tests are a true recurrence (block (x,y) needs the FINAL vectors of (x,y-1) and (x-1,y)), critical path
//            cols + rows blocks.  The expensive part of a step — the error of a candidate vector, ~1-2 us of dependent loads —
//            is taken OFF the critical path:
//              meNeighbourKernel  (all blocks in parallel) scores every block at the phase-A vectors of its upper and left
//                                 neighbour: in smooth fields the neighbours' final vectors ARE their phase-A vectors;
//              meDiagKernel       one workgroup per reference sweeps the anti-diagonals, a lane per block of the diagonal: the
//                                 neighbours' final vectors come from LDS (written on the previous diagonal), their errors from
//                                 the records above; only a vector no record knows is scored on the spot by the lane's wavefront.
//            A step is a few LDS operations and two LDS-only barriers instead of two candidate evaluations.
//            meWavefrontKernel (one wavefront per block row, rows hand vectors down through {tag,x,y} granules) remains for fields whose
//            diagonals exceed a workgroup.
//   phase C  meFinalizeKernel  (final 1/16-pel level only) variance-normalised error, rmsme, overlap in IEEE double, no FMA
//                              contraction (-ffp-contract=off), MCTF.cpp:1308-1321.
// The `> besterror` early exits of the reference are semantically inert (callers only use errors < best.error), so kernels
// always produce the full sum.
#include "common.h"
#include <stdlib.h>
#include <vector>

namespace {

typedef uint32_t u32x2 __attribute__( ( ext_vector_type( 2 ) ) );
typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
struct __attribute__( ( packed, aligned( 2 ) ) ) U4  { uint32_t v; };
struct __attribute__( ( packed, aligned( 2 ) ) ) U8  { u32x2 v; };
struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
__device__ __forceinline__ uint32_t ld4( const int16_t* p ) { return reinterpret_cast<const U4*>( p )->v; }
__device__ __forceinline__ u32x2 ld8( const int16_t* p ) { return reinterpret_cast<const U8*>( p )->v; }
__device__ __forceinline__ u32x4 ld16( const int16_t* p ) { return reinterpret_cast<const U16*>( p )->v; }
__device__ __forceinline__ int lo16( uint32_t v ) { return ( int ) ( int16_t ) ( v & 0xffffu ); }
__device__ __forceinline__ int hi16( uint32_t v ) { return ( int ) ( ( int32_t ) v >> 16 ); }

// LDS hand-over inside ONE wavefront (DS operations of a wave execute in order; the fence keeps the compiler from reordering them)
#define ME_WAVE_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }

__device__ __forceinline__ int waveSum( int v )       // DPP row operations + 4 readlanes instead of six ds_bpermute shuffles
{
  return ( int ) vvhipGroupSum32( ( uint32_t ) v, 64, threadIdx.x & 63 );
}

// 1/16-pel interpolation filters of the MCTF search.  Row f of kFilter4 = MCTF::m_interpolationFilter4[f] (MCTF.cpp:92-110),
// row f of kFilter6 = taps [1..6] of MCTF::m_interpolationFilter8[f] (MCTF.cpp:72-90; taps 0 and 7 are zero and unused :164-169).
__constant__ int8_t kFilter4[16][4] = {
  { 0, 64, 0, 0 },    { -2, 62, 4, 0 },   { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
  { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, { 0, 4, 62, -2 } };
__constant__ int8_t kFilter6[16][6] = {
  { 0, 0, 64, 0, 0, 0 },     { 1, -3, 64, 4, -2, 0 },   { 1, -6, 62, 9, -3, 1 },   { 2, -8, 60, 14, -5, 1 },  { 2, -9, 57, 19, -7, 2 },  { 3, -10, 53, 24, -8, 2 },
  { 3, -11, 50, 29, -9, 2 }, { 3, -11, 44, 35, -10, 3 }, { 1, -7, 38, 38, -7, 1 },  { 3, -10, 35, 44, -11, 3 }, { 2, -9, 29, 50, -11, 3 }, { 2, -8, 24, 53, -10, 3 },
  { 2, -7, 19, 57, -9, 2 },  { 1, -5, 14, 60, -8, 2 },  { 1, -3, 9, 62, -6, 1 },   { 0, -2, 4, 64, -3, 1 } };

// packed 16-bit helpers: two samples per dword.  v_dot2_i32_i16 multiplies both halves and accumulates in 32 bits — exact for 10-bit samples,
// 7-bit taps and 11-bit differences (sums of squares stay below 2^31 for the block sizes of the search, as in the reference's int).
typedef short s16x2 __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ int sdot2( uint32_t a, uint32_t b, int c ) { return __builtin_amdgcn_sdot2( __builtin_bit_cast( s16x2, a ), __builtin_bit_cast( s16x2, b ), c, false ); }
__device__ __forceinline__ uint32_t pk16( int lo, int hi ) { return ( uint32_t ) ( lo & 0xffff ) | ( ( uint32_t ) hi << 16 ); }
__device__ __forceinline__ uint32_t pkSub16( uint32_t a, uint32_t b ) { return __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2, a ) - __builtin_bit_cast( s16x2, b ) ); }
// clip to [0, maxVal] as ONE v_med3_i32 (with a run-time bound the compiler emits v_max + v_min: two of the ~13 instructions of a vertical filter step)
__device__ __forceinline__ int clipPel( int v, int maxVal ) { int r; asm( "v_med3_i32 %0, %1, 0, %2" : "=v"( r ) : "v"( v ), "v"( maxVal ) ); return r; }

// ---- error of one candidate, computed by a whole wavefront (all 64 lanes must call) -------------------------------------------
// integer displacement: sum (org - buf)^2 over w x h, w,h multiples of 8 (MCTF.cpp:122-145)
__device__ __forceinline__ int waveErrorInt( const int16_t* org, int os, const int16_t* buf, int bs, int w, int h, int lane )
{
  const int segs = w >> 3, total = segs * h;
  int e = 0;
  for( int i = lane; i < total; i += 64 )
  {
    const int y = i / segs, s = i - y * segs;
    const u32x4 a = ld16( org + ( ptrdiff_t ) y * os + 8 * s ), b = ld16( buf + ( ptrdiff_t ) y * bs + 8 * s );
    const uint32_t d0 = pkSub16( a.x, b.x ), d1 = pkSub16( a.y, b.y ), d2 = pkSub16( a.z, b.z ), d3 = pkSub16( a.w, b.w );
    e = sdot2( d0, d0, e ); e = sdot2( d1, d1, e ); e = sdot2( d2, d2, e ); e = sdot2( d3, d3, e );
  }
  return waveSum( e );
}

// ---- the 4-tap sub-pel error in two separable pieces, packed arithmetic (blocks 8 or 16 wide; the search of MCTFSpeed > 0) -------------------
// horizontal pass of `rows` rows starting at src (= first row, block column 0; taps reach columns -1 .. w+2): dst[r * w + x] = clip( ( sum_t f[t] * src[r][x - 1 + t] + 32 ) >> 6 )
__device__ __forceinline__ void horPass4( const int16_t* src, int bs, int rows, int w, int fx, int maxVal, int16_t* dst, int lane )
{
  const uint32_t c01 = pk16( kFilter4[fx][0], kFilter4[fx][1] ), c23 = pk16( kFilter4[fx][2], kFilter4[fx][3] );
  const int pshift = w == 16 ? 3 : 2, pairs = 1 << pshift;
  for( int i = lane; i < ( rows << pshift ); i += 64 )
  {
    const int r = i >> pshift, x = 2 * ( i & ( pairs - 1 ) );
    const int16_t* p = src + ( ptrdiff_t ) r * bs + x - 1;
    const u32x2 v = ld8( p );                      // (s0,s1) (s2,s3)
    const uint32_t v4 = ld4( p + 4 );              // (s4,s5)
    const uint32_t a = __builtin_amdgcn_alignbit( v.y, v.x, 16 ), b = __builtin_amdgcn_alignbit( v4, v.y, 16 );      // (s1,s2) (s3,s4)
    const int t0 = clipPel( sdot2( v.x, c01, sdot2( v.y, c23, 32 ) ) >> 6, maxVal );
    const int t1 = clipPel( sdot2( a, c01, sdot2( b, c23, 32 ) ) >> 6, maxVal );
    *reinterpret_cast<uint32_t*>( dst + r * w + x ) = pk16( t0, t1 );
  }
}
// vertical pass over rows tmp[0 .. h+2] + squared error against the original block; returns this lane's partial sum
__device__ __forceinline__ int verError4( const int16_t* org, int os, const int16_t* tmp, int w, int h, int fy, int maxVal, int lane )
{
  const uint32_t c01 = pk16( kFilter4[fy][0], kFilter4[fy][1] ), c23 = pk16( kFilter4[fy][2], kFilter4[fy][3] );
  const int pshift = w == 16 ? 3 : 2, pairs = 1 << pshift;
  int e = 0;
  for( int i = lane; i < ( h << pshift ); i += 64 )
  {
    const int y = i >> pshift, x = 2 * ( i & ( pairs - 1 ) );
    const int16_t* q = tmp + y * w + x;
    const uint32_t v0 = *reinterpret_cast<const uint32_t*>( q ), v1 = *reinterpret_cast<const uint32_t*>( q + w ),
                   v2 = *reinterpret_cast<const uint32_t*>( q + 2 * w ), v3 = *reinterpret_cast<const uint32_t*>( q + 3 * w );
    const uint32_t l01 = __builtin_amdgcn_perm( v1, v0, 0x05040100u ), h01 = __builtin_amdgcn_perm( v1, v0, 0x07060302u );      // (v0.lo,v1.lo) (v0.hi,v1.hi)
    const uint32_t l23 = __builtin_amdgcn_perm( v3, v2, 0x05040100u ), h23 = __builtin_amdgcn_perm( v3, v2, 0x07060302u );
    const int s0 = clipPel( sdot2( l01, c01, sdot2( l23, c23, 32 ) ) >> 6, maxVal );
    const int s1 = clipPel( sdot2( h01, c01, sdot2( h23, c23, 32 ) ) >> 6, maxVal );
    const uint32_t d = pkSub16( pk16( s0, s1 ), ld4( org + ( ptrdiff_t ) y * os + x ) );
    e = sdot2( d, d, e );
  }
  return e;
}

// fractional displacement (MCTF.cpp:147-257): horizontal pass -> clip -> LDS -> vertical pass -> clip -> squared error.
// TAP4: 4-tap filter rows (offsets -1..+2), else 6-tap (offsets -2..+3).  sTmp: (h + NT - 1) x w int16, private to the wave.
template<bool TAP4>
__device__ __forceinline__ int waveErrorFrac( const int16_t* org, int os, const int16_t* buf, int bs, int w, int h, int fx, int fy,
                                              int maxVal, int16_t* sTmp, int lane )
{
  constexpr int NT = TAP4 ? 4 : 6, OFF = TAP4 ? 1 : 2;
  int xf[NT], yf[NT];
#pragma unroll
  for( int t = 0; t < NT; t++ ) { xf[t] = TAP4 ? kFilter4[fx][t % 4] : kFilter6[fx][t % 6]; yf[t] = TAP4 ? kFilter4[fy][t % 4] : kFilter6[fy][t % 6]; }
  const int rows = h + NT - 1;
  const int pairs = w >> 1;                     // two horizontally adjacent outputs per lane-iteration
  for( int i = lane; i < rows * pairs; i += 64 )
  {
    const int r = i / pairs, x = 2 * ( i - r * pairs );
    const int16_t* p = buf + ( ptrdiff_t ) ( r - OFF ) * bs + x - OFF;
    int smp[NT + 1];
    if( TAP4 ) { const u32x2 v = ld8( p ); smp[0] = lo16( v.x ); smp[1] = hi16( v.x ); smp[2] = lo16( v.y ); smp[3] = hi16( v.y ); smp[4 % ( NT + 1 )] = p[4]; }
    else       { const u32x4 v = ld16( p ); smp[0] = lo16( v.x ); smp[1] = hi16( v.x ); smp[2] = lo16( v.y ); smp[3] = hi16( v.y );
                 smp[4 % ( NT + 1 )] = lo16( v.z ); smp[5 % ( NT + 1 )] = hi16( v.z ); smp[6 % ( NT + 1 )] = lo16( v.w ); }
    int s0 = 0, s1 = 0;
#pragma unroll
    for( int t = 0; t < NT; t++ ) { s0 += xf[t] * smp[t]; s1 += xf[t] * smp[t + 1]; }
    s0 = min( max( ( s0 + 32 ) >> 6, 0 ), maxVal );
    s1 = min( max( ( s1 + 32 ) >> 6, 0 ), maxVal );
    *reinterpret_cast<uint32_t*>( sTmp + r * w + x ) = ( uint32_t ) ( s0 & 0xffff ) | ( ( uint32_t ) s1 << 16 );
  }
  ME_WAVE_SYNC();                                // sTmp is private to the wave: its LDS writes above are ordered before the reads below
  int e = 0;
  for( int i = lane; i < h * pairs; i += 64 )
  {
    const int y = i / pairs, x = 2 * ( i - y * pairs );
    int s0 = 0, s1 = 0;
#pragma unroll
    for( int t = 0; t < NT; t++ )
    {
      const uint32_t v = *reinterpret_cast<const uint32_t*>( sTmp + ( y + t ) * w + x );
      s0 += yf[t] * lo16( v ); s1 += yf[t] * hi16( v );
    }
    s0 = min( max( ( s0 + 32 ) >> 6, 0 ), maxVal );
    s1 = min( max( ( s1 + 32 ) >> 6, 0 ), maxVal );
    const uint32_t o = ld4( org + ( ptrdiff_t ) y * os + x );
    const int d0 = s0 - lo16( o ), d1 = s1 - hi16( o );
    e += d0 * d0 + d1 * d1;
  }
  ME_WAVE_SYNC();                                // sTmp is reused by the next candidate
  return waveSum( e );
}

struct MeGeom
{
  const int16_t* org; int orgStride;
  const int16_t* buf; int bufStride;
  int width, height, bs;
  int lowRes, maxVal;
};

// The references of a picture are searched independently of each other (MCTF.cpp:666-707 loops over them): every level runs all of them in ONE launch,
// blockIdx.y = reference.  Same geometry for all; per reference its plane, the coarser level's field, the field written and the hand-off granules.
constexpr int ME_MAX_REFS = 8;
struct MeRefs
{
  const int16_t* buf[ME_MAX_REFS]; const vvhip_mv* prev[ME_MAX_REFS]; vvhip_mv* mvs[ME_MAX_REFS]; unsigned long long* gran[ME_MAX_REFS];
};

// MCTF::motionErrorLuma, MCTF.cpp:1099-1164 (x,y block origin; dx,dy in 1/16 pel)
__device__ __forceinline__ int meError( const MeGeom& g, int x, int y, int dx, int dy, int16_t* sTmp, int lane )
{
  const int fx = dx & 15, fy = dy & 15;
  const int w = min( g.bs, g.width - x ) & ~7, h = min( g.bs, g.height - y ) & ~7;
  const int16_t* o = g.org + x + ( ptrdiff_t ) y * g.orgStride;
  if( ( fx | fy ) == 0 )
    return waveErrorInt( o, g.orgStride, g.buf + x + dx / 16 + ( ptrdiff_t ) ( y + dy / 16 ) * g.bufStride, g.bufStride, w, h, lane );
  const int16_t* b = g.buf + x + ( dx >> 4 ) + ( ptrdiff_t ) ( y + ( dy >> 4 ) ) * g.bufStride;
  return g.lowRes ? waveErrorFrac<true>( o, g.orgStride, b, g.bufStride, w, h, fx, fy, g.maxVal, sTmp, lane )
                  : waveErrorFrac<false>( o, g.orgStride, b, g.bufStride, w, h, fx, fy, g.maxVal, sTmp, lane );
}

// Integer-vector candidates of a FULL 32 x 32 block (every level but the last works on 32 x 32 blocks and integer vectors only: ~60 candidates per block).  The original
// block stays in registers for all of them (8 dwords per lane: rows l >> 2 and 16 + (l >> 2), 16-byte segment l & 3); a candidate is two 16-byte loads, eight packed
// subtractions and eight dot products per lane — no per-candidate index arithmetic.
struct Org32 { u32x4 a, b; int off; };
__device__ __forceinline__ Org32 loadOrg32( const int16_t* o, int os, int cs, int lane )
{
  const int r = lane >> 2, s = lane & 3;
  Org32 O; O.a = ld16( o + ( ptrdiff_t ) r * os + 8 * s ); O.b = ld16( o + ( ptrdiff_t ) ( r + 16 ) * os + 8 * s ); O.off = r * cs + 8 * s;
  return O;
}
__device__ __forceinline__ int errInt32( const Org32& O, const int16_t* c, int cs, int lane )
{
  const u32x4 x = ld16( c + O.off ), y = ld16( c + O.off + 16 * cs );
  int e = 0;
  uint32_t d;
  d = pkSub16( O.a.x, x.x ); e = sdot2( d, d, e ); d = pkSub16( O.a.y, x.y ); e = sdot2( d, d, e ); d = pkSub16( O.a.z, x.z ); e = sdot2( d, d, e ); d = pkSub16( O.a.w, x.w ); e = sdot2( d, d, e );
  d = pkSub16( O.b.x, y.x ); e = sdot2( d, d, e ); d = pkSub16( O.b.y, y.y ); e = sdot2( d, d, e ); d = pkSub16( O.b.z, y.z ); e = sdot2( d, d, e ); d = pkSub16( O.b.w, y.w ); e = sdot2( d, d, e );
  return waveSum( e );
}

// A candidate equal to the current best vector cannot win (same error, the update needs a strictly smaller one): it is not evaluated.
#define ME_TRY( DX, DY ) do { const int dx_ = ( DX ), dy_ = ( DY );                                                              \
                              if( bestE == 0x7fffffff || dx_ != bestX || dy_ != bestY ) {                                       \
                                ME_COUNT( dx_, dy_ );                                                                            \
                                const int e_ = ME_ERROR( dx_, dy_ );                                                             \
                                if( e_ < bestE ) { bestX = dx_; bestY = dy_; bestE = e_; } } } while( 0 )
#define ME_ERROR( DX, DY ) meError( g, bx, by, ( DX ), ( DY ), sTmp, lane )
// scored candidates, counted the way SURVEY 8d prices them: a vector with both phases zero is an integer candidate (4 w h bytes), any other a fractional one
// ((w + taps - 1)(h + taps - 1) 2 + 2 w h bytes).  Only the STATS instances count (vvhip_mctf_set_stats); everywhere else the macro is empty.
#define ME_COUNT( DX, DY )
struct MeCount { int nInt, nFrac, nGrid, gridBytes, nRing, ringBytes; };      // nGrid: positions of a dense integer grid scored out of one staged window, gridBytes: that window + the block, read once; nRing / ringBytes: the same for the positions of a refinement ring

// One refinement ring of estimateLumaLn's final level (MCTF.cpp:1229-1288): the 8 positions (cx + x2, cy + y2), x2, y2 in {-a, 0, a} without the centre, tested in the
// reference's order (y2 outer, x2 inner) with its strict-< update.  The three x positions share their horizontal passes: one pass per x position over the rows any of its
// y positions needs, then a vertical pass + error per candidate (3 horizontal + 8 vertical passes instead of 8 + 8).  Same integers as motionErrorLumaFrac4 per candidate
// (an integer position through the filters is the identity: row 0 of the filter table is {0,64,0,0} and samples are already inside the clipping range).
template<bool STATS>
__device__ __forceinline__ void meRing3( const MeGeom& g, int bx, int by, int cx, int cy, int a, int& bestX, int& bestY, int& bestE, int16_t* sTmp, int lane, MeCount& cnt )
{
  const int w = min( g.bs, g.width - bx ) & ~7, h = min( g.bs, g.height - by ) & ~7;
  const bool shared = g.lowRes && ( w == 8 || w == 16 ) && h <= 16;
  if( !shared )
  {
    for( int y2 = -a; y2 <= a; y2 += a )
      for( int x2 = -a; x2 <= a; x2 += a )
        if( x2 || y2 )
        {
          const int dx_ = cx + x2, dy_ = cy + y2;
          if( dx_ != bestX || dy_ != bestY )
          {
            if( STATS ) { if( ( ( dx_ | dy_ ) & 15 ) == 0 ) cnt.nInt++; else cnt.nFrac++; }
            const int e_ = meError( g, bx, by, dx_, dy_, sTmp, lane ); if( e_ < bestE ) { bestX = dx_; bestY = dy_; bestE = e_; }
          }
        }
    return;
  }
  const int iyMin = ( cy - a ) >> 4, iyMax = ( cy + a ) >> 4, nrows = iyMax - iyMin + h + 3, region = nrows * w;
  const int16_t* o = g.org + bx + ( ptrdiff_t ) by * g.orgStride;
#pragma unroll
  for( int v = 0; v < 3; v++ )
  {
    const int X = cx + ( v - 1 ) * a;
    horPass4( g.buf + bx + ( X >> 4 ) + ( ptrdiff_t ) ( by + iyMin - 1 ) * g.bufStride, g.bufStride, nrows, w, X & 15, g.maxVal, sTmp + v * region, lane );
  }
  ME_WAVE_SYNC();
  for( int y2 = -a; y2 <= a; y2 += a )
  {
    const int Y = cy + y2, ro = ( Y >> 4 ) - iyMin;
#pragma unroll
    for( int v = 0; v < 3; v++ )
    {
      if( v == 1 && y2 == 0 ) continue;
      const int X = cx + ( v - 1 ) * a;
      if( X == bestX && Y == bestY ) continue;
      if( STATS ) cnt.nRing++;
      const int e_ = waveSum( verError4( o, g.orgStride, sTmp + v * region + ro * w, w, h, Y & 15, g.maxVal, lane ) );
      if( e_ < bestE ) { bestX = X; bestY = Y; bestE = e_; }
    }
  }
  // (SURVEY 8d's window form carried to a ring: its positions lie within half a sample of the centre — the ( w + 3 + 1 ) x ( h + 3 + 1 ) samples their 4-tap supports cover are
  //  read once (three horizontal passes over them), the block once, 8 bytes per position; the per-candidate figure is positions x ( ( w + 3 )( h + 3 ) 2 + 2 w h ))
  if( STATS ) cnt.ringBytes += ( w + 4 ) * ( h + 4 ) * 2 + 2 * w * h + 64;
  ME_WAVE_SYNC();                                // sTmp is reused by the next ring / candidate
}

// ---- phase B as a fixed-point iteration (round 6) -------------------------------------------------------------------------------------------------
// The above / left tests make block (x, y) depend on the FINAL vectors of (x, y-1) and (x-1, y): final( b ) = f( phaseA( b ), final( up ).vec, final( left ).vec ), a recurrence
// over a DAG with ONE solution.  The sweep walks it in topological order (nbx + nby - 1 dependent steps per level, one workgroup per reference: 230 us of a 1080p picture's
// 690 with the rest of the device idle).  Measured on the BASELINE clips, the recurrence hardly ever fires: 3..9 of 8 160 blocks change on the final level, chains of <= 3
// blocks.  So the solution is reached from the other side:
//   iteration 0   every block in parallel (inside meNeighbourKernel, which already holds everything needed): f( phaseA( b ), phaseA( up ).vec, phaseA( left ).vec ) — exact
//                 for every block whose neighbours keep their phase-A vectors; blocks whose result differs from their phase-A vector go to a list
//   iteration k   (meFixKernel, one workgroup per reference) only the right / lower neighbours of the blocks that changed in iteration k-1 are evaluated again with their
//                 neighbours' current vectors; a block that changes goes to the next list.  Ends when a list is empty.
// Chaotic iteration over a DAG converges to its unique solution whatever the order (by induction over the topological order: a block is evaluated again after the last
// change of each predecessor), so the fields are the sweep's — and the reference's — bit for bit.  Vectors are read and written as ONE 8-byte word; a block that reads a
// neighbour just before it changes is evaluated again in the next iteration (the neighbour is in that iteration's list).
__device__ __attribute__( ( noinline ) ) int meErrorCall( const MeGeom* g, int x, int y, int dx, int dy, int16_t* sTmp, int lane );      // (defined with the sweep kernel)
struct FixRec { int ownX, ownY, ownE, eU, eL, upX, upY, leftX, leftY; };                         // 36 bytes per block
struct FixLists { int* list[3]; int* stamp; int* count; };                                      // per reference: 3 x blocks ints, blocks ints, 4 ints (count[0] = blocks changed in iteration 0)
constexpr int ME_FIX_THREADS = 1024;

// f( own; vU, vL ) with the errors a record knows; `need` (out): 0 = resolved, 1 = vU has to be scored, 2 = vL has to be scored (then call again with the score)
struct FixBest { int x, y, e; };
__device__ __forceinline__ int fixKnown( const FixRec& r, int vx, int vy, int e1x, int e1y, int e1 )
{
  if( vx == r.ownX && vy == r.ownY ) return r.ownE;
  if( r.eU >= 0 && vx == r.upX && vy == r.upY ) return r.eU;
  if( r.eL >= 0 && vx == r.leftX && vy == r.leftY ) return r.eL;
  if( e1 >= 0 && vx == e1x && vy == e1y ) return e1;
  return -1;
}


// ---- phase A -------------------------------------------------------------------------------------------------------------
#undef ME_COUNT
#define ME_COUNT( DX, DY ) do { if( STATS ) { if( ( ( ( DX ) | ( DY ) ) & 15 ) == 0 ) cnt.nInt++; else cnt.nFrac++; } } while( 0 )
// scored-candidate counters of one launch class: { integer candidates, their algorithmic bytes, fractional candidates, their algorithmic bytes }
__device__ __forceinline__ void meCountFlush( unsigned long long* st, const MeCount& cnt, const MeGeom& g, int bx, int by, int lane )
{
  if( lane != 0 || !st ) return;
  const int w = min( g.bs, g.width - bx ) & ~7, h = min( g.bs, g.height - by ) & ~7, t = g.lowRes ? 3 : 5;
  if( cnt.nInt )  { atomicAdd( st + 0, ( unsigned long long ) cnt.nInt );  atomicAdd( st + 1, ( unsigned long long ) cnt.nInt * ( unsigned long long ) ( 4 * w * h ) ); }
  if( cnt.nFrac ) { atomicAdd( st + 2, ( unsigned long long ) cnt.nFrac ); atomicAdd( st + 3, ( unsigned long long ) cnt.nFrac * ( unsigned long long ) ( ( w + t ) * ( h + t ) * 2 + 2 * w * h ) ); }
  if( cnt.nGrid ) { atomicAdd( st + 4, ( unsigned long long ) cnt.nGrid ); atomicAdd( st + 5, ( unsigned long long ) cnt.gridBytes ); }
  if( cnt.nRing ) { atomicAdd( st + 6, ( unsigned long long ) cnt.nRing ); atomicAdd( st + 7, ( unsigned long long ) cnt.ringBytes ); }
}

// ---- the final level's full 16 x 16 blocks with the 4-tap search filter and search pattern 2 (MCTFSpeed >= 3: what presets faster .. medium run), round 6 --------------
// Eight candidates at a time instead of one: lane = 8 c + p scores COLUMN PAIR p (samples 2p, 2p + 1 of all 16 rows) of candidate c, the block's original column pair stays
// in 16 registers for every candidate of the block, a candidate's error is a sum over its 8 lanes (three DPP steps), and the reference's "first strictly smaller in scan
// order" is one minimum over keys ( error << 3 | order ) — errors of a 16 x 16 block at <= 10 bits stay below 2^29.  Same integers as motionErrorLumaInt / Frac4 per candidate.
__device__ __forceinline__ uint32_t groupSum8( uint32_t v )
{
  v += ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR1 ); v += ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR2 ); v += ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_HALF_MIRROR );
  return v;
}
__device__ __forceinline__ uint32_t waveMinOfGroups8( uint32_t key )      // every lane of an 8-lane group holds the group's key -> the minimum over the wave, uniform
{
  const uint32_t o = ( uint32_t ) VVHIP_DPP( key, VVHIP_DPP_MIRROR ); key = o < key ? o : key;
  const uint32_t r0 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) key, 0 ), r1 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) key, 16 );
  const uint32_t r2 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) key, 32 ), r3 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) key, 48 );
  const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
  return a < b ? a : b;
}
// integer vector (vx, vy) (multiples of 16) of this lane's candidate against the block at `blkBuf` (= buf + bx + by * stride): key of the lane's candidate group
__device__ __forceinline__ uint32_t intKey16( const uint32_t ( &orgCol )[16], const int16_t* blkBuf, int bufStride, int vx, int vy, bool valid, int lane )
{
  const int16_t* b = blkBuf + vx / 16 + ( ptrdiff_t ) ( vy / 16 ) * bufStride + 2 * ( lane & 7 );
  uint32_t rows[16];
#pragma unroll
  for( int i = 0; i < 16; i++ ) rows[i] = ld4( b + ( ptrdiff_t ) i * bufStride );
  int e = 0;
#pragma unroll
  for( int i = 0; i < 16; i++ ) { const uint32_t d = pkSub16( orgCol[i], rows[i] ); e = sdot2( d, d, e ); }
  const uint32_t sum = groupSum8( ( uint32_t ) e );
  return valid ? ( sum << 3 ) | ( uint32_t ) ( lane >> 3 ) : 0xffffffffu;
}
// the three horizontal passes of a ring in ONE loop, four outputs per item (a 16-byte request = the 7 samples four outputs need): 3 x nrows x 4 items for the wave instead of
// three loops of nrows x 8 two-output items (cut builds, 1080p / 4 references: the horizontal passes were 46 of the final level's 124 us — instruction-bound: serving their
// loads from a hot region changed nothing).  dst region v: nrows x 16 samples; X0..X2: the passes' horizontal positions (1/16 sample); src: row 0 of the regions, block column 0.
__device__ __forceinline__ void horPass4x3( const int16_t* src, int bs, int nrows, int X0, int X1, int X2, int maxVal, int16_t* dst, int region, int lane )
{
  const int n4 = nrows * 4;
  for( int e = lane; e < 3 * n4; e += 64 )
  {
    const int v = ( e >= n4 ) + ( e >= 2 * n4 ), within = e - v * n4, r = within >> 2, x = 4 * ( within & 3 );
    const int X = v == 0 ? X0 : ( v == 1 ? X1 : X2 ), fx = X & 15;
    const uint32_t c01 = pk16( kFilter4[fx][0], kFilter4[fx][1] ), c23 = pk16( kFilter4[fx][2], kFilter4[fx][3] );
    const u32x4 d = ld16( src + ( X >> 4 ) + ( ptrdiff_t ) r * bs + x - 1 );                   // samples x - 1 .. x + 6
    const uint32_t o0 = __builtin_amdgcn_alignbit( d.y, d.x, 16 ), o1 = __builtin_amdgcn_alignbit( d.z, d.y, 16 ), o2 = __builtin_amdgcn_alignbit( d.w, d.z, 16 );
    const int t0 = clipPel( sdot2( d.x, c01, sdot2( d.y, c23, 32 ) ) >> 6, maxVal );
    const int t1 = clipPel( sdot2( o0, c01, sdot2( o1, c23, 32 ) ) >> 6, maxVal );
    const int t2 = clipPel( sdot2( d.y, c01, sdot2( d.z, c23, 32 ) ) >> 6, maxVal );
    const int t3 = clipPel( sdot2( o1, c01, sdot2( o2, c23, 32 ) ) >> 6, maxVal );
    u32x2 o; o.x = pk16( t0, t1 ); o.y = pk16( t2, t3 );
    *reinterpret_cast<u32x2*>( dst + v * region + r * 16 + x ) = o;
  }
}

// one refinement ring (the 8 positions around (cx, cy) at distance a, reference order) -> updates best; sTmp: 3 regions of the horizontal passes
template<bool STATS>
__device__ __forceinline__ void meRing16( const MeGeom& g, int bx, int by, const uint32_t ( &orgCol )[16], int cx, int cy, int a, int& bestX, int& bestY, int& bestE, int16_t* sTmp, int lane, MeCount& cnt )
{
  const int iyMin = ( cy - a ) >> 4, iyMax = ( cy + a ) >> 4, nrows = iyMax - iyMin + 16 + 3, region = nrows * 16;
  horPass4x3( g.buf + bx + ( ptrdiff_t ) ( by + iyMin - 1 ) * g.bufStride, g.bufStride, nrows, cx - a, cx, cx + a, g.maxVal, sTmp, region, lane );
  ME_WAVE_SYNC();
  const int c = lane >> 3, p = lane & 7;
  const int cc = c + ( c >= 4 ), j = ( cc * 11 ) >> 5, v = cc - 3 * j;          // position in the 3 x 3 scan without its centre: y offset index j (outer), x offset index v (inner)
  const int Y = cy + ( j - 1 ) * a, fy = Y & 15, ro = ( Y >> 4 ) - iyMin;
  const uint32_t c01 = pk16( kFilter4[fy][0], kFilter4[fy][1] ), c23 = pk16( kFilter4[fy][2], kFilter4[fy][3] );
  const uint32_t* q = reinterpret_cast<const uint32_t*>( sTmp + v * region + ro * 16 + 2 * p );      // a row of a region is 16 samples = 8 dwords
  uint32_t r0 = q[0], r1 = q[8], r2 = q[16];
  uint32_t L0 = __builtin_amdgcn_perm( r1, r0, 0x05040100u ), H0 = __builtin_amdgcn_perm( r1, r0, 0x07060302u );      // ( row k, row k + 1 ) low / high samples
  uint32_t L1 = __builtin_amdgcn_perm( r2, r1, 0x05040100u ), H1 = __builtin_amdgcn_perm( r2, r1, 0x07060302u );
  int e = 0;
#pragma unroll
  for( int i = 0; i < 16; i++ )
  {
    const uint32_t r3 = q[8 * ( i + 3 )];
    const uint32_t L2 = __builtin_amdgcn_perm( r3, r2, 0x05040100u ), H2 = __builtin_amdgcn_perm( r3, r2, 0x07060302u );
    const int s0 = clipPel( sdot2( L0, c01, sdot2( L2, c23, 32 ) ) >> 6, g.maxVal );
    const int s1 = clipPel( sdot2( H0, c01, sdot2( H2, c23, 32 ) ) >> 6, g.maxVal );
    const uint32_t d = pkSub16( pk16( s0, s1 ), orgCol[i] );
    e = sdot2( d, d, e );
    L0 = L1; H0 = H1; L1 = L2; H1 = H2; r2 = r3;
  }
  const uint32_t key = waveMinOfGroups8( ( groupSum8( ( uint32_t ) e ) << 3 ) | ( uint32_t ) c );
  if( STATS ) { cnt.nRing += 8; cnt.ringBytes += ( 16 + 4 ) * ( 16 + 4 ) * 2 + 2 * 16 * 16 + 64; }      // (the ring in SURVEY 8d's window form, see meRing3)
  const int eMin = ( int ) ( key >> 3 );
  if( eMin < bestE )
  {
    const int cm = ( int ) ( key & 7u ), ccm = cm + ( cm >= 4 ), jm = ( ccm * 11 ) >> 5, vm = ccm - 3 * jm;
    bestX = cx + ( vm - 1 ) * a; bestY = cy + ( jm - 1 ) * a; bestE = eMin;
  }
  ME_WAVE_SYNC();                                // sTmp is reused by the next ring
}
// estimateLumaLn of one full 16 x 16 block of the final level up to (not including) the above / left tests (MCTF.cpp:1189-1288 with doubleRes, m_searchPttrn == 2)
template<bool STATS>
__device__ __forceinline__ bool meFinal16( const MeGeom& g, int bx, int by, const vvhip_mv* __restrict__ prev, int prevW, int prevH, int factor,
                                           int& bestX, int& bestY, int& bestE, int16_t* sTmp, int lane, MeCount& cnt )
{
  // the inherited vectors of the 3 x 3 coarser blocks (:1189-1208), then the zero vector (:1210-1214): candidates 0..9 in the reference's order, eight at a time
  int vx[2], vy[2]; bool valid[2];
#pragma unroll
  for( int round = 0; round < 2; round++ )
  {
    const int idx = 8 * round + ( lane >> 3 );
    vx[round] = 0; vy[round] = 0; valid[round] = idx == 9;
    if( idx < 9 )
    {
      const int pyI = ( idx * 11 ) >> 5, pxI = idx - 3 * pyI;
      const int ty = by / 32 + pyI - 1, tx = bx / 32 + pxI - 1;
      valid[round] = ty >= 0 && ty < prevH && tx >= 0 && tx < prevW;
      if( valid[round] ) { const vvhip_mv old = prev[ty * prevW + tx]; vx[round] = old.x * factor; vy[round] = old.y * factor; }
    }
  }
  // (the coarser levels only produce integer vectors; should a caller's field hold a fractional one, the block takes the general path)
  if( __ballot( ( valid[0] && ( ( vx[0] | vy[0] ) & 15 ) ) || ( valid[1] && ( ( vx[1] | vy[1] ) & 15 ) ) ) ) return false;
  uint32_t orgCol[16];
  {
    const int16_t* o = g.org + bx + ( ptrdiff_t ) by * g.orgStride + 2 * ( lane & 7 );
#pragma unroll
    for( int i = 0; i < 16; i++ ) orgCol[i] = ld4( o + ( ptrdiff_t ) i * g.orgStride );
  }
  const int16_t* blkBuf = g.buf + bx + ( ptrdiff_t ) by * g.bufStride;
#pragma unroll
  for( int round = 0; round < 2; round++ )
  {
    const uint32_t key = waveMinOfGroups8( intKey16( orgCol, blkBuf, g.bufStride, vx[round], vy[round], valid[round], lane ) );
    if( STATS ) cnt.nInt += __popcll( __ballot( valid[round] && ( lane & 7 ) == 0 ) );
    if( key != 0xffffffffu && ( int ) ( key >> 3 ) < bestE )
    {
      const int src = ( int ) ( key & 7u ) * 8;
      bestX = __builtin_amdgcn_readlane( vx[round], src ); bestY = __builtin_amdgcn_readlane( vy[round], src ); bestE = ( int ) ( key >> 3 );
    }
  }
  // (:1216-1228 with range 0: the one position is the best vector itself — same error, cannot win)
  int pbx = bestX, pby = bestY;
  meRing16<STATS>( g, bx, by, orgCol, pbx, pby, 6, bestX, bestY, bestE, sTmp, lane, cnt );
  pbx = bestX; pby = bestY;
  meRing16<STATS>( g, bx, by, orgCol, pbx, pby, 2, bestX, bestY, bestE, sTmp, lane, cnt );
  pbx = bestX; pby = bestY;
  meRing16<STATS>( g, bx, by, orgCol, pbx, pby, 1, bestX, bestY, bestE, sTmp, lane, cnt );
  return true;
}

template<bool STATS>
__global__ void __launch_bounds__( 64 )
meSearchKernel( MeGeom g, const MeRefs R, int nbx, int prevW, int prevH, int factor, int doubleRes, int searchPttrn, int mvsW, unsigned long long* stats, int fixBlocks )
{
  MeCount cnt = { 0, 0, 0, 0, 0, 0 };
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmp[48 * 52];      // the integer-grid window of a 32 x 32 block with range 8 (48 rows, pitch 50); > ( 32 + 5 ) * 32 of the sub-pel passes
  const int lane = threadIdx.x;
  g.buf = R.buf[blockIdx.y];
  const vvhip_mv* __restrict__ prev = R.prev[blockIdx.y];
  vvhip_mv* __restrict__ mvs = R.mvs[blockIdx.y];
  const int blk = blockIdx.x;
  const int byi = blk / nbx, bxi = blk - byi * nbx;
  const int bs = g.bs, bx = bxi * bs, by = byi * bs;

  int bestX = 0, bestY = 0, bestE = 0x7fffffff;
  // the final level's full 16 x 16 blocks (4-tap search filter, search pattern 2, integer inherited vectors, <= 10 bits): eight candidates at a time
  // (doubleRes bit 1 = the path is enabled: $VVHIP_MCTF_FINAL16=0 switches it off for A/B measurements, results identical)
  if( ( doubleRes & 2 ) && bs == 16 && g.lowRes && searchPttrn == 2 && prev && bx + 16 <= g.width && by + 16 <= g.height && g.maxVal <= 1023 &&
      meFinal16<STATS>( g, bx, by, prev, prevW, prevH, factor, bestX, bestY, bestE, sTmp, lane, cnt ) )
  {
    if( lane == 0 )
    {
      vvhip_mv& m = mvs[byi * mvsW + bxi];
      m.x = bestX; m.y = bestY; m.error = bestE;
      if( fixBlocks )
      {
        int* lists = reinterpret_cast<int*>( reinterpret_cast<char*>( R.gran[blockIdx.y] ) + ( size_t ) fixBlocks * sizeof( FixRec ) );
        lists[3 * ( size_t ) fixBlocks + blk] = 0;
        if( blk == 0 ) lists[4 * ( size_t ) fixBlocks] = 0;
      }
    }
    if( STATS ) meCountFlush( stats, cnt, g, bx, by, lane );
    return;
  }
  // full 32 x 32 blocks: integer vectors are scored from registers (errInt32); anything else takes the general path
  const bool full32 = bs == 32 && bx + 32 <= g.width && by + 32 <= g.height;
  Org32 O32 = {};
  if( full32 ) O32 = loadOrg32( g.org + bx + ( ptrdiff_t ) by * g.orgStride, g.orgStride, g.bufStride, lane );
#undef ME_ERROR
#define ME_ERROR( DX, DY ) ( ( full32 && ( ( ( DX ) | ( DY ) ) & 15 ) == 0 ) ? errInt32( O32, g.buf + bx + ( DX ) / 16 + ( ptrdiff_t ) ( by + ( DY ) / 16 ) * g.bufStride, g.bufStride, lane ) \
                                                                           : meError( g, bx, by, ( DX ), ( DY ), sTmp, lane ) )
  int range = doubleRes ? 0 : ( searchPttrn == 2 ? 3 : 5 );              // MCTF.cpp:1178
  if( !prev ) range = 8;                                                  // :1183-1186
  else
  {
    for( int py = -1; py <= 1; py++ )                                     // :1189-1208
    {
      const int ty = by / ( 2 * bs ) + py;
      if( ty < 0 || ty >= prevH ) continue;
      for( int px = -1; px <= 1; px++ )
      {
        const int tx = bx / ( 2 * bs ) + px;
        if( tx < 0 || tx >= prevW ) continue;
        const vvhip_mv old = prev[ty * prevW + tx];
        ME_TRY( old.x * factor, old.y * factor );
      }
    }
    ME_TRY( 0, 0 );                                                       // :1210-1214
  }
  {
    const int pbx = bestX, pby = bestY;                                   // :1216-1228
    const int d = ( !prev && searchPttrn == 2 ) ? 2 : 1;
    if( full32 && range > 0 )
    {
      // the dense integer grid of a full 32 x 32 block from an LDS WINDOW: ( 32 + 2 range )^2 samples staged once (every request in flight before the first is used),
      // then every position is two rows of five dword reads per lane (v_alignbit for odd columns) against the original block held in registers — a candidate no longer
      // waits for global memory, and consecutive candidates are independent (their reductions overlap).  Same scan order and strict-< update as the reference; a position
      // equal to the current best vector has the same error and cannot win, so evaluating it is harmless.
      const int gx0 = pbx / 16 - range, gy0 = pby / 16 - range, W = 32 + 2 * range;
      const int PS = 2 * ( ( ( W + 3 ) >> 1 ) | 1 ), cpr = ( W + 7 ) >> 3, total = W * cpr;      // odd dword pitch: the five-dword reads of a 32-lane group cover all banks
      const uint32_t cprInv = ( 65536u + ( uint32_t ) cpr - 1u ) / ( uint32_t ) cpr;              // i / cpr for i < 288, cpr <= 6: exact
      const int16_t* src = g.buf + bx + gx0 + ( ptrdiff_t ) ( by + gy0 ) * g.bufStride;
      for( int i0 = lane; i0 < total; i0 += 5 * 64 )
      {
        u32x4 v[5]; int at[5], left[5];
#pragma unroll
        for( int q = 0; q < 5; q++ )
        {
          const int i = i0 + 64 * q < total ? i0 + 64 * q : i0, r = ( int ) ( ( ( uint32_t ) i * cprInv ) >> 16 ), c = i - r * cpr;
          at[q] = r * PS + 8 * c; left[q] = ( PS >> 1 ) - 4 * c;
          v[q] = ld16( src + ( ptrdiff_t ) r * g.bufStride + 8 * c );
        }
#pragma unroll
        for( int q = 0; q < 5; q++ )
          if( i0 + 64 * q < total )
          {
            uint32_t* dst = reinterpret_cast<uint32_t*>( sTmp + at[q] );
            dst[0] = v[q].x; if( left[q] > 1 ) dst[1] = v[q].y; if( left[q] > 2 ) dst[2] = v[q].z; if( left[q] > 3 ) dst[3] = v[q].w;
          }
      }
      ME_WAVE_SYNC();
      const int r = lane >> 2, sgm = lane & 3;
      const uint32_t* rowA = reinterpret_cast<const uint32_t*>( sTmp + r * PS + 8 * sgm );
      for( int gy = 0; gy <= 2 * range; gy += d )
      {
        for( int gx = 0; gx <= 2 * range; gx += d )
        {
          const uint32_t* pa = rowA + ( ( gy * PS + gx ) >> 1 );             // ( gy * PS + gx ) & ~1 samples on: PS and 8 sgm are even, the parity is gx's
          const uint32_t* pb = pa + 8 * PS;                                   // 16 rows further down
          const uint32_t sh = ( gx & 1 ) * 16;
          const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], a4 = pa[4], b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3], b4 = pb[4];
          int e = 0; uint32_t df;
          df = pkSub16( O32.a.x, __builtin_amdgcn_alignbit( a1, a0, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.a.y, __builtin_amdgcn_alignbit( a2, a1, sh ) ); e = sdot2( df, df, e );
          df = pkSub16( O32.a.z, __builtin_amdgcn_alignbit( a3, a2, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.a.w, __builtin_amdgcn_alignbit( a4, a3, sh ) ); e = sdot2( df, df, e );
          df = pkSub16( O32.b.x, __builtin_amdgcn_alignbit( b1, b0, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.b.y, __builtin_amdgcn_alignbit( b2, b1, sh ) ); e = sdot2( df, df, e );
          df = pkSub16( O32.b.z, __builtin_amdgcn_alignbit( b3, b2, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.b.w, __builtin_amdgcn_alignbit( b4, b3, sh ) ); e = sdot2( df, df, e );
          const int e_ = waveSum( e );
          if( e_ < bestE ) { bestX = ( gx0 + gx ) * 16; bestY = ( gy0 + gy ) * 16; bestE = e_; }
        }
      }
      // (SURVEY 8d's window form: "(w + 2R)(h + 2R) 2 + w h 2 bytes per block for (2R + 1)^2 candidates" + 8 per position; the per-candidate figure 4 w h each is reported
      //  next to it as a work rate: st[4] x 4 w h)
      if( STATS ) { const int per = 2 * range / d + 1; cnt.nGrid += per * per; cnt.gridBytes += W * W * 2 + 32 * 32 * 2 + 8 * per * per; }
      ME_WAVE_SYNC();                                                         // sTmp is reused by the refinement rings
    }
    else
    for( int y2 = pby / 16 - range; y2 <= pby / 16 + range; y2 += d )
      for( int x2 = pbx / 16 - range; x2 <= pbx / 16 + range; x2 += d )
        ME_TRY( x2 * 16, y2 * 16 );
  }
  if( doubleRes )                                                         // :1229-1288
  {
    int pbx = bestX, pby = bestY;
    const int dr = searchPttrn ? 6 : 12, d1 = searchPttrn == 2 ? 6 : 4;
    if( d1 == dr ) meRing3<STATS>( g, bx, by, pbx, pby, dr, bestX, bestY, bestE, sTmp, lane, cnt );      // the 3 x 3 ring of search pattern 2
    else
      for( int y2 = -dr; y2 <= dr; y2 += d1 )
        for( int x2 = -dr; x2 <= dr; x2 += d1 )
          if( x2 || y2 ) ME_TRY( pbx + x2, pby + y2 );
    pbx = bestX; pby = bestY;
    meRing3<STATS>( g, bx, by, pbx, pby, 2, bestX, bestY, bestE, sTmp, lane, cnt );
    pbx = bestX; pby = bestY;
    meRing3<STATS>( g, bx, by, pbx, pby, 1, bestX, bestY, bestE, sTmp, lane, cnt );
  }
  if( lane == 0 )
  {
    vvhip_mv& m = mvs[byi * mvsW + bxi];
    m.x = bestX; m.y = bestY; m.error = bestE;
    if( fixBlocks )      // the fixed-point form of phase B: this block's stamp and (first block) the changed-block count start at zero — no separate clearing launch
    {
      int* lists = reinterpret_cast<int*>( reinterpret_cast<char*>( R.gran[blockIdx.y] ) + ( size_t ) fixBlocks * sizeof( FixRec ) );
      lists[3 * ( size_t ) fixBlocks + blk] = 0;
      if( blk == 0 ) lists[4 * ( size_t ) fixBlocks] = 0;
    }
  }
  if( STATS ) meCountFlush( stats, cnt, g, bx, by, lane );
}

#undef ME_ERROR
#define ME_ERROR( DX, DY ) meError( g, bx, by, ( DX ), ( DY ), sTmp, lane )
#undef ME_COUNT
#define ME_COUNT( DX, DY )

// ---- phase A of the coarse levels, four wavefronts per block (round 6) ---------------------------------------------------------------------------------------
// The levels above the full-resolution ones have a few hundred blocks: one wavefront per block leaves the device empty while every wave walks its ~60-90 candidates one
// after the other (25 + 20 + 37 us of a 1080p call's 274 us of candidate scoring, with 160 / 540 / 2 040 waves in flight).  Here a workgroup of four waves shares a FULL
// 32 x 32 block: the inherited vectors are dealt to the waves (all requests in flight at once), the window of the dense integer grid is staged by all 256 threads, the grid
// positions are dealt round-robin, and the reference's "first strictly smaller in scan order" becomes a minimum over ( error, scan index ).  Blocks cut by the picture edge
// deal their candidates the same way and score each with the general routine.  Integer vectors only (every level but the last).
constexpr int ME_COOP_WAVES = 4;
__global__ void __launch_bounds__( 64 * ME_COOP_WAVES )
meSearchCoopKernel( MeGeom g, const MeRefs R, int nbx, int prevW, int prevH, int factor, int searchPttrn, int mvsW, int fixBlocks )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sWin[48 * 52];
  __shared__ int sCandE[12], sCandX[12], sCandY[12];
  __shared__ int sBestE[ME_COOP_WAVES], sBestP[ME_COOP_WAVES];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane( t >> 6 );
  g.buf = R.buf[blockIdx.y];
  const vvhip_mv* __restrict__ prev = R.prev[blockIdx.y];
  vvhip_mv* __restrict__ mvs = R.mvs[blockIdx.y];
  const int blk = blockIdx.x, byi = blk / nbx, bxi = blk - byi * nbx;
  const int bx = bxi * 32, by = byi * 32;
  // full blocks: original in registers, grid out of a staged window; blocks cut by the picture edge (8 .. 24 samples on a side): the same dealing of candidates, each scored
  // by the general routine (wave-uniform choice, workgroup-uniform control flow)
  const bool full = bx + 32 <= g.width && by + 32 <= g.height;
  Org32 O32 = {};
  if( full ) O32 = loadOrg32( g.org + bx + ( ptrdiff_t ) by * g.orgStride, g.orgStride, g.bufStride, lane );
  int bestX = 0, bestY = 0, bestE = 0x7fffffff;
  int range = searchPttrn == 2 ? 3 : 5;                                   // MCTF.cpp:1178 (not the final level)
  if( !prev ) range = 8;                                                  // :1183-1186
  else
  {
    // the 3 x 3 coarser blocks' vectors (:1189-1208) and the zero vector (:1210-1214): candidate k = 0..9 in the reference's order; wave w scores k = w, w + 4, w + 8
    if( t < 12 ) sCandE[t] = 0x7fffffff;
    __syncthreads();
    int cx[3], cy[3]; bool ok[3]; u32x4 va[3], vb[3];
#pragma unroll
    for( int q = 0; q < 3; q++ )
    {
      const int k = wave + ME_COOP_WAVES * q;
      cx[q] = 0; cy[q] = 0; ok[q] = k == 9;
      if( k < 9 )
      {
        const int py = ( k * 11 ) >> 5, px = k - 3 * py, ty = by / 64 + py - 1, tx = bx / 64 + px - 1;
        ok[q] = ty >= 0 && ty < prevH && tx >= 0 && tx < prevW;
        if( ok[q] ) { const vvhip_mv old = prev[ty * prevW + tx]; cx[q] = old.x * factor; cy[q] = old.y * factor; }
      }
      if( full )
      {
        const int16_t* c = g.buf + bx + cx[q] / 16 + ( ptrdiff_t ) ( by + cy[q] / 16 ) * g.bufStride;
        va[q] = ld16( c + O32.off ); vb[q] = ld16( c + O32.off + 16 * g.bufStride );      // (an unused slot reads the zero-vector position: in range)
      }
    }
#pragma unroll
    for( int q = 0; q < 3; q++ )
    {
      int e = 0; uint32_t d;
      if( !full )
      {
        const int k = wave + ME_COOP_WAVES * q;
        const int ev = meError( g, bx, by, cx[q], cy[q], sWin, lane );
        if( lane == 0 && ok[q] && k < 10 ) { sCandE[k] = ev; sCandX[k] = cx[q]; sCandY[k] = cy[q]; }
        continue;
      }
      d = pkSub16( O32.a.x, va[q].x ); e = sdot2( d, d, e ); d = pkSub16( O32.a.y, va[q].y ); e = sdot2( d, d, e ); d = pkSub16( O32.a.z, va[q].z ); e = sdot2( d, d, e ); d = pkSub16( O32.a.w, va[q].w ); e = sdot2( d, d, e );
      d = pkSub16( O32.b.x, vb[q].x ); e = sdot2( d, d, e ); d = pkSub16( O32.b.y, vb[q].y ); e = sdot2( d, d, e ); d = pkSub16( O32.b.z, vb[q].z ); e = sdot2( d, d, e ); d = pkSub16( O32.b.w, vb[q].w ); e = sdot2( d, d, e );
      e = waveSum( e );
      const int k = wave + ME_COOP_WAVES * q;
      if( lane == 0 && ok[q] && k < 10 ) { sCandE[k] = e; sCandX[k] = cx[q]; sCandY[k] = cy[q]; }
    }
    __syncthreads();
    for( int k = 0; k < 10; k++ )                                         // the reference's order, strict <
    {
      const int e = sCandE[k];
      if( e < bestE ) { bestE = e; bestX = sCandX[k]; bestY = sCandY[k]; }
    }
  }
  // the dense integer grid around the best vector so far (:1216-1228)
  const int d = ( !prev && searchPttrn == 2 ) ? 2 : 1;
  const int gx0 = bestX / 16 - range, gy0 = bestY / 16 - range, W = 32 + 2 * range;
  const int PS = 2 * ( ( ( W + 3 ) >> 1 ) | 1 ), cpr = ( W + 7 ) >> 3, total = W * cpr;
  const uint32_t cprInv = ( 65536u + ( uint32_t ) cpr - 1u ) / ( uint32_t ) cpr;
  const int16_t* src = g.buf + bx + gx0 + ( ptrdiff_t ) ( by + gy0 ) * g.bufStride;
  if( full )
  for( int i0 = t; i0 < total; i0 += 2 * 64 * ME_COOP_WAVES )
  {
    u32x4 v[2]; int at[2], left[2];
#pragma unroll
    for( int q = 0; q < 2; q++ )
    {
      const int i = i0 + 64 * ME_COOP_WAVES * q < total ? i0 + 64 * ME_COOP_WAVES * q : i0, r = ( int ) ( ( ( uint32_t ) i * cprInv ) >> 16 ), c = i - r * cpr;
      at[q] = r * PS + 8 * c; left[q] = ( PS >> 1 ) - 4 * c;
      v[q] = ld16( src + ( ptrdiff_t ) r * g.bufStride + 8 * c );
    }
#pragma unroll
    for( int q = 0; q < 2; q++ )
      if( i0 + 64 * ME_COOP_WAVES * q < total )
      {
        uint32_t* dst = reinterpret_cast<uint32_t*>( sWin + at[q] );
        dst[0] = v[q].x; if( left[q] > 1 ) dst[1] = v[q].y; if( left[q] > 2 ) dst[2] = v[q].z; if( left[q] > 3 ) dst[3] = v[q].w;
      }
  }
  __syncthreads();
  {
    const int r = lane >> 2, sgm = lane & 3, per = 2 * range / d + 1, nPos = per * per;
    const uint32_t* rowA = reinterpret_cast<const uint32_t*>( sWin + r * PS + 8 * sgm );
    int locE = 0x7fffffff, locP = 0x7fffffff;
    for( int p = wave; p < nPos; p += ME_COOP_WAVES )
    {
      const int iy = p / per, gy = iy * d, gx = ( p - iy * per ) * d;
      if( !full )
      {
        const int e_ = meError( g, bx, by, ( gx0 + gx ) * 16, ( gy0 + gy ) * 16, sWin, lane );
        if( e_ < locE ) { locE = e_; locP = p; }
        continue;
      }
      const uint32_t* pa = rowA + ( ( gy * PS + gx ) >> 1 );
      const uint32_t* pb = pa + 8 * PS;
      const uint32_t sh = ( gx & 1 ) * 16;
      const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], a4 = pa[4], b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3], b4 = pb[4];
      int e = 0; uint32_t df;
      df = pkSub16( O32.a.x, __builtin_amdgcn_alignbit( a1, a0, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.a.y, __builtin_amdgcn_alignbit( a2, a1, sh ) ); e = sdot2( df, df, e );
      df = pkSub16( O32.a.z, __builtin_amdgcn_alignbit( a3, a2, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.a.w, __builtin_amdgcn_alignbit( a4, a3, sh ) ); e = sdot2( df, df, e );
      df = pkSub16( O32.b.x, __builtin_amdgcn_alignbit( b1, b0, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.b.y, __builtin_amdgcn_alignbit( b2, b1, sh ) ); e = sdot2( df, df, e );
      df = pkSub16( O32.b.z, __builtin_amdgcn_alignbit( b3, b2, sh ) ); e = sdot2( df, df, e ); df = pkSub16( O32.b.w, __builtin_amdgcn_alignbit( b4, b3, sh ) ); e = sdot2( df, df, e );
      const int e_ = waveSum( e );
      if( e_ < locE ) { locE = e_; locP = p; }                           // (p increases: the wave's first minimum)
    }
    if( lane == 0 ) { sBestE[wave] = locE; sBestP[wave] = locP; }
    __syncthreads();
    int gE = 0x7fffffff, gP = 0x7fffffff;
#pragma unroll
    for( int w2 = 0; w2 < ME_COOP_WAVES; w2++ )
    {
      const int e = sBestE[w2], p = sBestP[w2];
      if( e < gE || ( e == gE && p < gP ) ) { gE = e; gP = p; }          // smallest error, then the earliest position of the scan
    }
    if( gE < bestE )
    {
      const int iy = gP / per;
      bestX = ( gx0 + ( gP - iy * per ) * d ) * 16; bestY = ( gy0 + iy * d ) * 16; bestE = gE;
    }
  }
  if( t == 0 )
  {
    vvhip_mv& m = mvs[byi * mvsW + bxi];
    m.x = bestX; m.y = bestY; m.error = bestE;
    if( fixBlocks )
    {
      int* lists = reinterpret_cast<int*>( reinterpret_cast<char*>( R.gran[blockIdx.y] ) + ( size_t ) fixBlocks * sizeof( FixRec ) );
      lists[3 * ( size_t ) fixBlocks + blk] = 0;
      if( blk == 0 ) lists[4 * ( size_t ) fixBlocks] = 0;
    }
  }
}

// ---- phase B -------------------------------------------------------------------------------------------------------------
// granule = { tag (hi 32) , x (bits 16..31), y (bits 0..15) }; tag == 1 marks "final"
__device__ __forceinline__ uint64_t packGranule( int x, int y ) { return ( 1ull << 32 ) | ( ( uint64_t ) ( uint16_t ) ( int16_t ) x << 16 ) | ( uint16_t ) ( int16_t ) y; }

__global__ void __launch_bounds__( 64 )
meWavefrontKernel( MeGeom g, const MeRefs R, int nbx, int mvsW, int* abortFlag )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmp[( 32 + 5 ) * 32];
  const int lane = threadIdx.x;
  g.buf = R.buf[blockIdx.y];
  vvhip_mv* __restrict__ mvs = R.mvs[blockIdx.y];
  unsigned long long* granules = R.gran[blockIdx.y];                      // nby x nbx, zeroed
  const int byi = blockIdx.x, bs = g.bs, by = byi * bs;
  int leftX = 0, leftY = 0;
  for( int bxi = 0; bxi < nbx; bxi++ )
  {
    const int bx = bxi * bs;
    vvhip_mv& m = mvs[byi * mvsW + bxi];
    int bestX = m.x, bestY = m.y, bestE = m.error;
    int aboveX = 0, aboveY = 0; bool haveAbove = false;
    if( byi > 0 )                                                         // MCTF.cpp:1289-1297
    {
      unsigned long long gr = 0;
      if( lane == 0 )
      {
        const unsigned long long* src = granules + ( size_t ) ( byi - 1 ) * nbx + bxi;
        unsigned spins = 0;
        const unsigned long long t0 = wall_clock64();                      // constant-rate counter (100 MHz): the bound is TIME (2 s), not iterations — a preempted or
        while( true )                                                     // time-shared producer must not look like a dead one
        {
          gr = __hip_atomic_load( src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
          if( ( gr >> 32 ) == 1ull ) break;
          if( ( ++spins & 255u ) == 0 && ( __hip_atomic_load( abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) || wall_clock64() - t0 > 200000000ull ) )
          { __hip_atomic_store( abortFlag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); gr = ~0ull; break; }
          __builtin_amdgcn_s_sleep( 2 );
        }
      }
      const uint32_t glo = __shfl( ( uint32_t ) gr, 0 ), ghi = __shfl( ( uint32_t ) ( gr >> 32 ), 0 );
      if( ghi != 1u ) return;                                             // aborted: bounded spin expired (reported by the host)
      aboveX = ( int ) ( int16_t ) ( glo >> 16 ); aboveY = ( int ) ( int16_t ) ( glo & 0xffffu ); haveAbove = true;
      ME_TRY( aboveX, aboveY );
    }
    // the left block's vector: evaluated unless it is the vector just tested (same error: loses to it or ties with the unchanged best)
    if( bxi > 0 && !( haveAbove && leftX == aboveX && leftY == aboveY ) ) ME_TRY( leftX, leftY );                  // MCTF.cpp:1298-1306
    leftX = bestX; leftY = bestY;
    if( lane == 0 )
    {
      m.x = bestX; m.y = bestY; m.error = bestE;
      __hip_atomic_store( granules + ( size_t ) byi * nbx + bxi, packGranule( bestX, bestY ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    }
  }
}

// ---- phase B, parallel part: errors at the neighbours' phase-A vectors ---------------------------------------------------------
struct NbRec { int eU, eL, upX, upY, leftX, leftY; };      // e < 0: not scored (the vector equals one whose error is known)

__global__ void __launch_bounds__( 64 )
meNeighbourKernel( MeGeom g, const MeRefs R, int nbx, int mvsW, unsigned long long* stats, int fix, int nBlocks )
{
  MeCount cnt = { 0, 0, 0, 0, 0, 0 };
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmp[( 32 + 5 ) * 32];
  const int lane = threadIdx.x;
  g.buf = R.buf[blockIdx.y];
  const vvhip_mv* __restrict__ mvs = R.mvs[blockIdx.y];
  NbRec* __restrict__ nb = reinterpret_cast<NbRec*>( R.gran[blockIdx.y] );
  const int blk = blockIdx.x, byi = blk / nbx, bxi = blk - byi * nbx;
  const int bs = g.bs, bx = bxi * bs, by = byi * bs;
  const vvhip_mv own = mvs[byi * mvsW + bxi];
  NbRec r; r.eU = -1; r.eL = -1; r.upX = r.upY = r.leftX = r.leftY = 0;
  if( byi > 0 )
  {
    const vvhip_mv up = mvs[( byi - 1 ) * mvsW + bxi];
    r.upX = up.x; r.upY = up.y;
    if( up.x != own.x || up.y != own.y ) { r.eU = meError( g, bx, by, up.x, up.y, sTmp, lane ); if( ( ( up.x | up.y ) & 15 ) == 0 ) cnt.nInt++; else cnt.nFrac++; }
  }
  if( bxi > 0 )
  {
    const vvhip_mv lf = mvs[byi * mvsW + bxi - 1];
    r.leftX = lf.x; r.leftY = lf.y;
    if( ( lf.x != own.x || lf.y != own.y ) && !( byi > 0 && lf.x == r.upX && lf.y == r.upY ) )
    { r.eL = meError( g, bx, by, lf.x, lf.y, sTmp, lane ); if( ( ( lf.x | lf.y ) & 15 ) == 0 ) cnt.nInt++; else cnt.nFrac++; }
  }
  if( lane == 0 )
  {
    if( !fix ) nb[byi * nbx + bxi] = r;
    else
    {
      // iteration 0 of the fixed-point form: the record carries the block's own phase-A result, and the block is resolved against its neighbours' phase-A vectors
      unsigned long long* base = R.gran[blockIdx.y];
      FixRec* recs = reinterpret_cast<FixRec*>( base );
      FixRec f; f.ownX = own.x; f.ownY = own.y; f.ownE = own.error; f.eU = r.eU; f.eL = r.eL; f.upX = r.upX; f.upY = r.upY; f.leftX = r.leftX; f.leftY = r.leftY;
      recs[byi * nbx + bxi] = f;
      bool changed = false;
      int bestE = own.error;
      if( byi > 0 && r.eU >= 0 && r.eU < bestE ) { bestE = r.eU; changed = true; }
      if( bxi > 0 && r.eL >= 0 && r.eL < bestE ) { changed = true; }
      if( changed )
      {
        int* lists = reinterpret_cast<int*>( reinterpret_cast<char*>( base ) + ( size_t ) nBlocks * sizeof( FixRec ) );      // [3 lists][stamps][counts]
        int* count = lists + 4 * ( size_t ) nBlocks;
        lists[atomicAdd( count, 1 )] = byi * nbx + bxi;
      }
    }
  }
  if( stats ) meCountFlush( stats, cnt, g, bx, by, lane );
}

// ---- phase B, iterations >= 1 of the fixed-point form: one workgroup per reference, a wavefront per block to evaluate ---------------------------------------------
__global__ void __launch_bounds__( ME_FIX_THREADS )
meFixKernel( MeGeom g, const MeRefs R, int nbx, int nby, int mvsW, unsigned long long* stats )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmpAll[ME_FIX_THREADS / 64][( 32 + 5 ) * 32];
  __shared__ MeGeom sG;
  __shared__ int sN[2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int WAVES = ME_FIX_THREADS / 64;
  int16_t* sTmp = sTmpAll[wave];
  g.buf = R.buf[blockIdx.x];
  if( t == 0 ) sG = g;
  vvhip_mv* __restrict__ mvs = R.mvs[blockIdx.x];
  const int nBlocks = nbx * nby, bs = g.bs;
  unsigned long long* base = R.gran[blockIdx.x];
  const FixRec* __restrict__ recs = reinterpret_cast<const FixRec*>( base );
  int* lists = reinterpret_cast<int*>( reinterpret_cast<char*>( base ) + ( size_t ) nBlocks * sizeof( FixRec ) );
  int* stamp = lists + 3 * ( size_t ) nBlocks;
  int* count = lists + 4 * ( size_t ) nBlocks;
  int n = __hip_atomic_load( count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  if( n == 0 ) return;                                                        // (the common case on smooth content: nothing to do)
  __syncthreads();
  auto vecOf = [&]( int b ) -> unsigned long long {                           // a block's current vector as one word (x low, y high)
    const int by_ = b / nbx, bx_ = b - by_ * nbx;
    return __hip_atomic_load( reinterpret_cast<const unsigned long long*>( &mvs[by_ * mvsW + bx_] ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP ); };
  // f( own; up's vector, left's vector ), errors from the record or scored by this wavefront (MCTF.cpp:1289-1306)
  auto resolve = [&]( int b, const FixRec& r, FixBest& best ) {
    const int by_ = b / nbx, bx_ = b - by_ * nbx;
    best.x = r.ownX; best.y = r.ownY; best.e = r.ownE;
    int upX = 0, upY = 0, e1 = -1;
    const bool haveUp = by_ > 0;
    if( haveUp )
    {
      const unsigned long long v = vecOf( b - nbx );
      upX = ( int ) ( uint32_t ) v; upY = ( int ) ( uint32_t ) ( v >> 32 );
      if( upX != best.x || upY != best.y )
      {
        e1 = fixKnown( r, upX, upY, 0, 0, -1 );
        if( e1 < 0 )
        {
          e1 = meErrorCall( &sG, bx_ * bs, by_ * bs, upX, upY, sTmp, lane );
          if( stats ) { MeCount c1 = { 0, 0, 0, 0, 0, 0 }; c1.nInt = ( ( upX | upY ) & 15 ) == 0; c1.nFrac = !c1.nInt; meCountFlush( stats, c1, sG, bx_ * bs, by_ * bs, lane ); }
        }
        if( e1 < best.e ) { best.x = upX; best.y = upY; best.e = e1; }
      }
    }
    if( bx_ > 0 )
    {
      const unsigned long long v = vecOf( b - 1 );
      const int lfX = ( int ) ( uint32_t ) v, lfY = ( int ) ( uint32_t ) ( v >> 32 );
      if( !( haveUp && lfX == upX && lfY == upY ) && ( lfX != best.x || lfY != best.y ) )
      {
        int e2 = fixKnown( r, lfX, lfY, upX, upY, e1 );
        if( e2 < 0 )
        {
          e2 = meErrorCall( &sG, bx_ * bs, by_ * bs, lfX, lfY, sTmp, lane );
          if( stats ) { MeCount c1 = { 0, 0, 0, 0, 0, 0 }; c1.nInt = ( ( lfX | lfY ) & 15 ) == 0; c1.nFrac = !c1.nInt; meCountFlush( stats, c1, sG, bx_ * bs, by_ * bs, lane ); }
        }
        if( e2 < best.e ) { best.x = lfX; best.y = lfY; best.e = e2; }
      }
    } };
  auto store = [&]( int b, const FixBest& best ) {
    if( lane != 0 ) return;
    const int by_ = b / nbx, bx_ = b - by_ * nbx;
    vvhip_mv& m = mvs[by_ * mvsW + bx_];
    __hip_atomic_store( reinterpret_cast<unsigned long long*>( &m ), ( unsigned long long ) ( uint32_t ) best.x | ( ( unsigned long long ) ( uint32_t ) best.y << 32 ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
    m.error = best.e; };
  // iteration 0's results: every listed block against its neighbours' PHASE-A vectors (all in its record: nothing to score, nothing read from the field), then written
  int* cur = lists;
  for( int i = wave; i < n; i += WAVES )
  {
    const int b = cur[i];
    const FixRec r = recs[b];
    FixBest best = { r.ownX, r.ownY, r.ownE };
    if( b >= nbx && r.eU >= 0 && r.eU < best.e ) { best.x = r.upX; best.y = r.upY; best.e = r.eU; }
    if( b % nbx > 0 && r.eL >= 0 && r.eL < best.e ) { best.x = r.leftX; best.y = r.leftY; best.e = r.eL; }
    store( b, best );
  }
  __threadfence_block();
  __syncthreads();
  int* dep = lists + nBlocks;
  int* nxt = lists + 2 * ( size_t ) nBlocks;
  const int maxIt = nbx + nby + 2;
  for( int it = 1; n > 0 && it <= maxIt; it++ )
  {
    if( t == 0 ) { sN[0] = 0; sN[1] = 0; }
    __syncthreads();
    // the blocks to evaluate again: right and lower neighbours of the blocks that changed, once each
    for( int i = t; i < 2 * n; i += ME_FIX_THREADS )
    {
      const int c = cur[i >> 1], cy = c / nbx, cx = c - cy * nbx;
      const int d = ( i & 1 ) ? ( cy + 1 < nby ? c + nbx : -1 ) : ( cx + 1 < nbx ? c + 1 : -1 );
      if( d >= 0 && atomicExch( &stamp[d], it ) != it ) dep[atomicAdd( &sN[0], 1 )] = d;
    }
    __threadfence_block();
    __syncthreads();
    const int nd = sN[0];
    for( int j = wave; j < nd; j += WAVES )
    {
      const int b = dep[j];
      const FixRec r = recs[b];
      FixBest best;
      resolve( b, r, best );
      const unsigned long long old = vecOf( b );
      if( ( int ) ( uint32_t ) old != best.x || ( int ) ( uint32_t ) ( old >> 32 ) != best.y )
      {
        store( b, best );
        if( lane == 0 ) nxt[atomicAdd( &sN[1], 1 )] = b;
      }
    }
    __threadfence_block();
    __syncthreads();
    n = sN[1];
    int* tmp = cur; cur = nxt; nxt = tmp;      // (the changed list just read becomes the next iteration's output list)
    __syncthreads();
  }
}

// ---- phase B, sequential part: anti-diagonal sweep -------------------------------------------------------------------------------
// workgroup barrier that orders LDS traffic only: the sweep's global loads (fetched a diagonal ahead) and stores (results, nobody in the launch reads them back)
// stay in flight across it
#define ME_LDS_BARRIER() asm volatile( "s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory" )
constexpr int ME_DIAG_MAX_THREADS = 320, ME_DIAG_MAX_COLS = 1024;

// the rare on-the-spot evaluation of the sweep, kept out of line: the sweep's loop stays a few dozen instructions in registers (inlined, the two-pass error code made
// the compiler spill the loop's state to scratch — a global-memory round trip on every step of the chain: 219 -> 128 us on the final level of a 1080p picture)
__device__ __attribute__( ( noinline ) ) int meErrorCall( const MeGeom* g, int x, int y, int dx, int dy, int16_t* sTmp, int lane ) { return meError( *g, x, y, dx, dy, sTmp, lane ); }

__global__ void __launch_bounds__( ME_DIAG_MAX_THREADS )
meDiagKernel( MeGeom g, const MeRefs R, int nbx, int nby, int mvsW, unsigned long long* stats )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmpAll[ME_DIAG_MAX_THREADS / 64][( 32 + 5 ) * 32];
  __shared__ int finX[ME_DIAG_MAX_COLS], finY[ME_DIAG_MAX_COLS];      // final vector of the last finished block of every block column
  __shared__ MeGeom sG;
  const int t = threadIdx.x, lane = t & 63;
  int16_t* sTmp = sTmpAll[t >> 6];
  g.buf = R.buf[blockIdx.x];
  if( t == 0 ) sG = g;
  __syncthreads();
  vvhip_mv* __restrict__ mvs = R.mvs[blockIdx.x];
  const NbRec* __restrict__ nb = reinterpret_cast<const NbRec*>( R.gran[blockIdx.x] );
  const int bs = g.bs, nd = nbx + nby - 1;
  // the block's own phase-A result and its neighbour record do not depend on the sweep: they are fetched one diagonal ahead
  // (two ahead, and three diagonals at a time parked in LDS, both measured slower: 166 / 312 us against 128 us)
  int nX = 0, nY = 0, nE = 0, nEU = -1, nEL = -1, nUX = 0, nUY = 0, nLX = 0, nLY = 0;
  auto fetch = [&]( int d ) {
    const int xlo = d - ( nby - 1 ) > 0 ? d - ( nby - 1 ) : 0, xhi = d < nbx - 1 ? d : nbx - 1;
    if( d < nd && t <= xhi - xlo )
    {
      const int x = xlo + t, y = d - x;
      const vvhip_mv o = mvs[y * mvsW + x]; const NbRec r = nb[y * nbx + x];
      nX = o.x; nY = o.y; nE = o.error; nEU = r.eU; nEL = r.eL; nUX = r.upX; nUY = r.upY; nLX = r.leftX; nLY = r.leftY;
    } };
  fetch( 0 );
  for( int d = 0; d < nd; d++ )
  {
    const int xlo = d - ( nby - 1 ) > 0 ? d - ( nby - 1 ) : 0, xhi = d < nbx - 1 ? d : nbx - 1;
    const bool active = t <= xhi - xlo;
    const int x = xlo + t, y = d - x;
    const int ownX = nX, ownY = nY, ownE = nE, rEU = nEU, rEL = nEL, rUX = nUX, rUY = nUY, rLX = nLX, rLY = nLY;
    fetch( d + 1 );
    int bestX = ownX, bestY = ownY, bestE = ownE;
    int upX = 0, upY = 0, lfX = 0, lfY = 0;
    if( active )
    {
      if( y > 0 ) { upX = finX[x]; upY = finY[x]; }
      if( x > 0 ) { lfX = finX[x - 1]; lfY = finY[x - 1]; }
    }
    ME_LDS_BARRIER();                                                       // every lane has read the previous diagonal's vectors
    // error of vector (vx, vy) for this block if a record knows it, else -1
    auto known = [&]( int vx, int vy, int e1x, int e1y, int e1 ) -> int {
      if( vx == ownX && vy == ownY ) return ownE;
      if( rEU >= 0 && vx == rUX && vy == rUY ) return rEU;
      if( rEL >= 0 && vx == rLX && vy == rLY ) return rEL;
      if( e1 >= 0 && vx == e1x && vy == e1y ) return e1;
      return -1; };
    // scores the vectors no record knows, one at a time, by the whole wavefront of the lane that needs it
    auto scoreMissing = [&]( bool missing, int vx, int vy, int& e ) {
      unsigned long long m = __ballot( missing );
      while( m )
      {
        const int l = __ffsll( ( long long ) m ) - 1;
        const int lx = __shfl( x, l ), ly = __shfl( y, l ), lvx = __shfl( vx, l ), lvy = __shfl( vy, l );
        const int v = meErrorCall( &sG, lx * bs, ly * bs, lvx, lvy, sTmp, lane );
        if( stats ) { MeCount c1 = { 0, 0, 0, 0, 0, 0 }; c1.nInt = ( ( lvx | lvy ) & 15 ) == 0; c1.nFrac = !c1.nInt; meCountFlush( stats, c1, sG, lx * bs, ly * bs, lane ); }
        if( lane == l ) e = v;
        m &= m - 1;
      } };
    // the upper block's final vector (MCTF.cpp:1289-1297)
    const bool try1 = active && y > 0 && ( upX != bestX || upY != bestY );
    int e1 = try1 ? known( upX, upY, 0, 0, -1 ) : -1;
    scoreMissing( try1 && e1 < 0, upX, upY, e1 );
    if( try1 && e1 < bestE ) { bestX = upX; bestY = upY; bestE = e1; }
    // the left block's final vector, unless it is the vector just tested (:1298-1306)
    const bool try2 = active && x > 0 && !( y > 0 && lfX == upX && lfY == upY ) && ( lfX != bestX || lfY != bestY );
    int e2 = try2 ? known( lfX, lfY, upX, upY, try1 ? e1 : -1 ) : -1;
    scoreMissing( try2 && e2 < 0, lfX, lfY, e2 );
    if( try2 && e2 < bestE ) { bestX = lfX; bestY = lfY; bestE = e2; }
    if( active )
    {
      vvhip_mv& m = mvs[y * mvsW + x];
      m.x = bestX; m.y = bestY; m.error = bestE;
      finX[x] = bestX; finY[x] = bestY;
    }
    ME_LDS_BARRIER();
  }
}

// ---- phase C -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__( 256 )
meFinalizeKernel( MeGeom g, const MeRefs R, int nRefs, int nbx, int nby, int bitDepth, int unitSize, int mvsW )
{
  const int i = blockIdx.x * 4 + ( threadIdx.x >> 6 ), lane = threadIdx.x & 63;          // one wave per block; the block's variance serves every reference
  if( i >= nbx * nby ) return;
  const int byi = i / nbx, bxi = i - byi * nbx, bx = bxi * g.bs, by = byi * g.bs;
  const int w = min( g.bs, g.width - bx ) & ~7, h = min( g.bs, g.height - by ) & ~7;
  const int16_t* o = g.org + bx + ( ptrdiff_t ) by * g.orgStride;
  int sum = 0;                                                            // calcVarCore, MCTF.cpp:520-546
  for( int k = lane; k < w * h; k += 64 ) { const int y = k / w, x = k - y * w; sum += o[( ptrdiff_t ) y * g.orgStride + x]; }
  int avg = waveSum( sum );
  avg <<= 4;
  avg = avg / ( w * h );
  long long var = 0;
  for( int k = lane; k < w * h; k += 64 ) { const int y = k / w, x = k - y * w; const int p = ( o[( ptrdiff_t ) y * g.orgStride + x] << 4 ) - avg; var += p * p; }
#pragma unroll
  for( int d = 32; d >= 1; d >>= 1 ) var += __shfl_xor( var, d );
  if( lane >= nRefs ) return;                                             // lane r finishes reference r
  vvhip_mv& m = R.mvs[lane][byi * mvsW + bxi];
  const double bdScale = ( double ) ( 1 << ( 2 * ( 10 - bitDepth ) ) );   // MCTF.cpp:1314-1320
  const double dvar = ( ( double ) var / 256.0 ) * bdScale;
  const double mse  = m.error * bdScale / ( double ) ( w * h );
  const int    e    = ( int ) ( 20 * ( ( m.error * bdScale + 5.0 ) / ( dvar + 5.0 ) ) + mse / 50.0 );
  m.rmsme   = ( int32_t ) ( uint16_t ) ( 0.5 + __builtin_sqrt( mse ) );
  m.overlap = ( ( double ) w * h ) / ( unitSize * unitSize );
  m.error   = e;
}

__global__ void initMvsKernel( vvhip_mv* mvs, int count )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if( i < count ) { vvhip_mv m; m.x = 0; m.y = 0; m.error = 0x7fffffff; m.rmsme = 65535; m.overlap = 0.0; mvs[i] = m; }   // MotionVector(), MCTF.h:79
}

// ---- stand-alone batches (table-entry shaped) -------------------------------------------------------------------------------
__global__ void __launch_bounds__( 64 )
mctfErrorBatchKernel( const int16_t* __restrict__ org, int os, const int16_t* __restrict__ buf, int bs, int w, int h, int tap4, int maxVal,
                      const vvhip_mctf_item* __restrict__ items, int32_t* __restrict__ out )
{
  __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t sTmp[( 64 + 5 ) * 64];
  const int lane = threadIdx.x;
  const vvhip_mctf_item it = items[blockIdx.x];
  const int16_t* o = org + it.org_off;
  const int16_t* b = buf + it.buf_off;
  int e;
  if( ( it.fx | it.fy ) == 0 ) e = waveErrorInt( o, os, b, bs, w, h, lane );
  else if( tap4 )              e = waveErrorFrac<true>( o, os, b, bs, w, h, it.fx, it.fy, maxVal, sTmp, lane );
  else                         e = waveErrorFrac<false>( o, os, b, bs, w, h, it.fx, it.fy, maxVal, sTmp, lane );
  if( lane == 0 ) out[blockIdx.x] = e;
}

__global__ void __launch_bounds__( 64 )
calcVarBatchKernel( const int16_t* __restrict__ org, int os, int w, int h, const int32_t* __restrict__ off, int64_t* __restrict__ out )
{
  const int lane = threadIdx.x;
  const int16_t* o = org + off[blockIdx.x];
  int s = 0;
  for( int i = lane; i < w * h; i += 64 ) { const int y = i / w, x = i - y * w; s += o[( ptrdiff_t ) y * os + x]; }
  s = waveSum( s );
  int avg = s << 4;
  avg = avg / ( w * h );
  long long v = 0;
  for( int i = lane; i < w * h; i += 64 ) { const int y = i / w, x = i - y * w; const int p = ( o[( ptrdiff_t ) y * os + x] << 4 ) - avg; v += p * p; }
  int lo = ( int ) ( uint32_t ) v, hi = ( int ) ( v >> 32 );
  // 64-bit wave sum through two 32-bit shuffles per step
  for( int o2 = 32; o2 > 0; o2 >>= 1 )
  {
    const uint32_t olo = ( uint32_t ) __shfl_xor( lo, o2 ); const int ohi = __shfl_xor( hi, o2 );
    const long long other = ( ( long long ) ohi << 32 ) | olo;
    const long long mine = ( ( long long ) hi << 32 ) | ( uint32_t ) lo;
    const long long t = mine + other;
    lo = ( int ) ( uint32_t ) t; hi = ( int ) ( t >> 32 );
  }
  if( lane == 0 ) out[blockIdx.x] = ( ( long long ) hi << 32 ) | ( uint32_t ) lo;
}

__global__ void __launch_bounds__( 256 )
subsampleKernel( const int16_t* __restrict__ src, int ss, int16_t* __restrict__ dst, int ds, int nw, int nh )
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if( x >= nw || y >= nh ) return;
  const int16_t* a = src + ( ptrdiff_t ) ( 2 * y ) * ss + 2 * x;
  dst[( ptrdiff_t ) y * ds + x] = ( int16_t ) ( ( a[0] + a[ss] + a[1] + a[ss + 1] + 2 ) >> 2 );   // MCTF.cpp:1091
}

// left/right replication for rows [0,h)
__global__ void __launch_bounds__( 256 )
extendLRKernel( int16_t* plane, int stride, int w, int h, int pad )
{
  const int y = blockIdx.x;
  int16_t* r = plane + ( ptrdiff_t ) y * stride;
  const int16_t l = r[0], rr = r[w - 1];
  for( int x = threadIdx.x; x < pad; x += blockDim.x ) { r[-1 - x] = l; r[w + x] = rr; }
}
// top/bottom replication of whole padded rows
__global__ void __launch_bounds__( 256 )
extendTBKernel( int16_t* plane, int stride, int w, int h, int pad )
{
  const int y = blockIdx.x;   // 0..2*pad-1
  const int16_t* src = y < pad ? plane - pad : plane + ( ptrdiff_t ) ( h - 1 ) * stride - pad;
  int16_t* dst = y < pad ? plane - ( ptrdiff_t ) ( y + 1 ) * stride - pad : plane + ( ptrdiff_t ) ( h + ( y - pad ) ) * stride - pad;
  for( int x = threadIdx.x; x < w + 2 * pad; x += blockDim.x ) dst[x] = src[x];
}

// the pyramid level of several pictures (current + references) in one launch each: blockIdx.z / .y = picture
constexpr int ME_MAX_PICS = 1 + ME_MAX_REFS;
struct MePlanes { const int16_t* src[ME_MAX_PICS]; int16_t* dst[ME_MAX_PICS]; };

__global__ void __launch_bounds__( 256 )
subsampleBatchKernel( const MePlanes P, int ss, int ds, int nw, int nh )
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if( x >= nw || y >= nh ) return;
  const int16_t* a = P.src[blockIdx.z] + ( ptrdiff_t ) ( 2 * y ) * ss + 2 * x;
  P.dst[blockIdx.z][( ptrdiff_t ) y * ds + x] = ( int16_t ) ( ( a[0] + a[ss] + a[1] + a[ss + 1] + 2 ) >> 2 );   // MCTF.cpp:1091
}
__global__ void __launch_bounds__( 128 )
extendLRBatchKernel( const MePlanes P, int stride, int w, int h, int pad )
{
  int16_t* r = P.dst[blockIdx.y] + ( ptrdiff_t ) blockIdx.x * stride;
  const int16_t l = r[0], rr = r[w - 1];
  for( int x = threadIdx.x; x < pad; x += blockDim.x ) { r[-1 - x] = l; r[w + x] = rr; }
}
__global__ void __launch_bounds__( 256 )
extendTBBatchKernel( const MePlanes P, int stride, int w, int h, int pad )
{
  int16_t* plane = P.dst[blockIdx.y];
  const int y = blockIdx.x;   // 0..2*pad-1
  const int16_t* src = y < pad ? plane - pad : plane + ( ptrdiff_t ) ( h - 1 ) * stride - pad;
  int16_t* dst = y < pad ? plane - ( ptrdiff_t ) ( y + 1 ) * stride - pad : plane + ( ptrdiff_t ) ( h + ( y - pad ) ) * stride - pad;
  for( int x = threadIdx.x; x < w + 2 * pad; x += blockDim.x ) dst[x] = src[x];
}

// one pyramid level of several pictures INCLUDING its replicated border, one launch (round 6: was subsample + left/right + top/bottom = three launches per level):
// the output sample at (x, y) of the padded plane is the 2 x 2 average at the clamped position — what extendBorderPel replicates (MCTF.cpp:1072-1097)
__global__ void __launch_bounds__( 256 )
pyramidLevelKernel( const MePlanes P, int ss, int ds, int nw, int nh, int pad )
{
  const int x = ( int ) ( blockIdx.x * blockDim.x + threadIdx.x ) - pad, y = ( int ) blockIdx.y - pad;
  if( x >= nw + pad ) return;
  const int cx = min( max( x, 0 ), nw - 1 ), cy = min( max( y, 0 ), nh - 1 );
  const int16_t* a = P.src[blockIdx.z] + ( ptrdiff_t ) ( 2 * cy ) * ss + 2 * cx;
  P.dst[blockIdx.z][( ptrdiff_t ) y * ds + x] = ( int16_t ) ( ( a[0] + a[ss] + a[1] + a[ss + 1] + 2 ) >> 2 );
}

// default MotionVector() for the level fields of every reference (one contiguous array) and the result fields (R.mvs) in one launch: blockIdx.y = nRefs addresses the array
__global__ void initMvsAllKernel( const MeRefs R, int nRefs, int count, vvhip_mv* fields, int fieldCount )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  vvhip_mv m; m.x = 0; m.y = 0; m.error = 0x7fffffff; m.rmsme = 65535; m.overlap = 0.0;                                    // MotionVector(), MCTF.h:79
  if( ( int ) blockIdx.y < nRefs ) { if( i < count ) R.mvs[blockIdx.y][i] = m; }
  else for( int k = i; k < fieldCount; k += gridDim.x * blockDim.x ) fields[k] = m;
}

__global__ void initMvsBatchKernel( const MeRefs R, int count )       // R.mvs[blockIdx.y]: the result field of a reference
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if( i < count ) { vvhip_mv m; m.x = 0; m.y = 0; m.error = 0x7fffffff; m.rmsme = 65535; m.overlap = 0.0; R.mvs[blockIdx.y][i] = m; }   // MotionVector(), MCTF.h:79
}

int ensureScratch( vvhip_ctx* ctx, size_t bytes )
{
  if( ctx->scratchBytes >= bytes ) return VVHIP_OK;
  if( ctx->d_scratch ) { VVHIP_CHECK_HIP( ctx, hipStreamSynchronize( ctx->stream ) ); VVHIP_CHECK_HIP( ctx, hipFree( ctx->d_scratch ) ); ctx->d_scratch = nullptr; ctx->scratchBytes = 0; }
  hipError_t e = hipMalloc( &ctx->d_scratch, bytes );
  if( e != hipSuccess ) return vvhip_fail( ctx, VVHIP_E_NOMEM, "scratch hipMalloc(%zu): %s", bytes, hipGetErrorString( e ) );
  ctx->scratchBytes = bytes;
  return VVHIP_OK;
}

// per-class timing of a motion-estimation call (vvhip_mctf_set_timing): one event per mark, created on first use and kept
void meMark( vvhip_ctx* ctx, int tag )
{
  if( !ctx->mctfTiming ) return;
  const size_t k = ctx->mctfEvTag.size();
  if( k >= ctx->mctfEv.size() ) { hipEvent_t e = nullptr; if( hipEventCreate( &e ) != hipSuccess ) return; ctx->mctfEv.push_back( e ); }
  if( hipEventRecord( ctx->mctfEv[k], ctx->stream ) == hipSuccess ) ctx->mctfEvTag.push_back( tag );
}

// one hierarchy level for nRefs references at once (R.gran[r]: nbx * nby granules per reference, one contiguous region starting at R.gran[0] with pitch granPitch)
int meLevel( vvhip_ctx* ctx, const int16_t* d_org, int os, int bsd, int width, int height, int bs, const MeRefs& R, int nRefs, size_t granPitch,
             int prevW, int prevH, int factor, int doubleRes, int pttrn, int lowRes, int bitDepth, int unit, int mvsW, int mvsH, int* d_abort, int* usedHandOff = nullptr )
{
  // blocks processed: bx + 8 <= width, by + 8 <= height (MCTF.cpp:1174,1357)
  const int nbx = width >= 8 ? ( width - 8 ) / bs + 1 : 0, nby = height >= 8 ? ( height - 8 ) / bs + 1 : 0;
  if( nbx <= 0 || nby <= 0 ) return VVHIP_OK;
  if( nbx > mvsW || nby > mvsH ) return vvhip_fail( ctx, VVHIP_E_ARG, "MCTF level: motion field %dx%d too small for %dx%d blocks", mvsW, mvsH, nbx, nby );
  MeGeom g; g.org = d_org; g.orgStride = os; g.buf = nullptr; g.bufStride = bsd; g.width = width; g.height = height; g.bs = bs;
  g.lowRes = lowRes; g.maxVal = ( 1 << bitDepth ) - 1;
  meMark( ctx, 1 );
  unsigned long long* st = ctx->d_mctfStats;      // (null unless vvhip_mctf_set_stats switched the counters on: 3 phases x 8 counters, include/vvenc_hip.h)
  // phase B: $VVHIP_MCTF_DIAG = 2 (default) the fixed-point form (any field size), 1 the anti-diagonal sweep (fields up to 320 blocks on the shorter side), 0 the row hand-off
  static const int useDiag = []{ const char* e = getenv( "VVHIP_MCTF_DIAG" ); return e ? atoi( e ) : 2; }();
  const int fixBlocks = useDiag >= 2 ? nbx * nby : 0;
  static const int final16 = []{ const char* e = getenv( "VVHIP_MCTF_FINAL16" ); return e ? atoi( e ) : 1; }();
  const int dblArg = doubleRes ? ( 1 | ( final16 ? 2 : 0 ) ) : 0;
  // the coarse levels (few blocks: one wave per block leaves the device empty) give their full 32 x 32 blocks to four waves each; the blocks at the picture edge and
  // every level with enough blocks to fill the device take the one-wave-per-block kernel ($VVHIP_MCTF_COOP=0: always; the counting instance always)
  static const int coopOn = []{ const char* e = getenv( "VVHIP_MCTF_COOP" ); return e ? atoi( e ) : 1; }();
  static const long coopMax = []{ const char* e = getenv( "VVHIP_MCTF_COOP_MAX" ); return e ? atol( e ) : 2048l; }();
  if( coopOn && !st && !doubleRes && bs == 32 && ( long ) nbx * nby * nRefs <= coopMax )
    hipLaunchKernelGGL( meSearchCoopKernel, dim3( nbx * nby, nRefs ), dim3( 64 * ME_COOP_WAVES ), 0, ctx->stream, g, R, nbx, prevW, prevH, factor, pttrn, mvsW, fixBlocks );
  else if( st ) hipLaunchKernelGGL( meSearchKernel<true>,  dim3( nbx * nby, nRefs ), dim3( 64 ), 0, ctx->stream, g, R, nbx, prevW, prevH, factor, dblArg, pttrn, mvsW, st, fixBlocks );
  else          hipLaunchKernelGGL( meSearchKernel<false>, dim3( nbx * nby, nRefs ), dim3( 64 ), 0, ctx->stream, g, R, nbx, prevW, prevH, factor, dblArg, pttrn, mvsW, st, fixBlocks );
  VVHIP_LAUNCH_CHECK( ctx );
  meMark( ctx, 2 );
  const int diagLen = nbx < nby ? nbx : nby;
#ifdef VVHIP_DEV_KNOBS      // (development aid, changes the results: no phase B at all — what the near-empty launches of phase B cost a GOP cycle, tools/exp/small_launches.sh)
  static const int noPhaseB = getenv( "VVHIP_MCTF_NO_PHASE_B" ) ? atoi( getenv( "VVHIP_MCTF_NO_PHASE_B" ) ) : 0;      // 1: neither launch, 2: the neighbour tests but not the resolution
  if( noPhaseB == 1 ) { meMark( ctx, 3 ); }
  else if( noPhaseB == 2 && fixBlocks )
  {
    hipLaunchKernelGGL( meNeighbourKernel, dim3( nbx * nby, nRefs ), dim3( 64 ), 0, ctx->stream, g, R, nbx, mvsW, st ? st + 8 : nullptr, 1, fixBlocks );
    meMark( ctx, 3 );
  }
  else
#endif
  if( fixBlocks )
  {
    // (the granule area holds, per reference: one FixRec per block, three block lists, the stamps and the counts: 52 bytes per block + 16 <= 7 granules per block)
    hipLaunchKernelGGL( meNeighbourKernel, dim3( nbx * nby, nRefs ), dim3( 64 ), 0, ctx->stream, g, R, nbx, mvsW, st ? st + 8 : nullptr, 1, fixBlocks );
    meMark( ctx, 3 );
    hipLaunchKernelGGL( meFixKernel, dim3( nRefs ), dim3( ME_FIX_THREADS ), 0, ctx->stream, g, R, nbx, nby, mvsW, st ? st + 16 : nullptr );
    VVHIP_LAUNCH_CHECK( ctx );
  }
  else if( useDiag && diagLen <= ME_DIAG_MAX_THREADS && nbx <= ME_DIAG_MAX_COLS )
  {
    // (the granule area holds the neighbour records here: 3 granules = one NbRec per block)
    hipLaunchKernelGGL( meNeighbourKernel, dim3( nbx * nby, nRefs ), dim3( 64 ), 0, ctx->stream, g, R, nbx, mvsW, st ? st + 8 : nullptr, 0, 0 );
    meMark( ctx, 3 );
#ifdef VVHIP_DEV_KNOBS      // (development aid, changes the results: the final level without its above / left tests = the phase-A vectors, for the analysis of the sweep's chains)
    if( !( doubleRes && getenv( "VVHIP_MCTF_NO_SWEEP" ) ) )
#endif
    hipLaunchKernelGGL( meDiagKernel, dim3( nRefs ), dim3( ( ( diagLen + 63 ) / 64 ) * 64 ), 0, ctx->stream, g, R, nbx, nby, mvsW, st ? st + 16 : nullptr );
    VVHIP_LAUNCH_CHECK( ctx );
  }
  else
  {
    if( usedHandOff ) *usedHandOff = 1;
    VVHIP_CHECK_HIP( ctx, hipMemsetAsync( R.gran[0], 0, sizeof( unsigned long long ) * ( granPitch * ( size_t ) ( nRefs - 1 ) + ( size_t ) nbx * nby ), ctx->stream ) );
    // blocks are dispatched in linear order (x fastest): the row a wave waits for — same reference, one row up — always has the smaller linear index
    hipLaunchKernelGGL( meWavefrontKernel, dim3( nby, nRefs ), dim3( 64 ), 0, ctx->stream, g, R, nbx, mvsW, d_abort );
    VVHIP_LAUNCH_CHECK( ctx );
  }
  meMark( ctx, 4 );
  if( doubleRes )
  {
    hipLaunchKernelGGL( meFinalizeKernel, dim3( ( nbx * nby + 3 ) / 4 ), dim3( 256 ), 0, ctx->stream, g, R, nRefs, nbx, nby, bitDepth, unit, mvsW );
    VVHIP_LAUNCH_CHECK( ctx );
    meMark( ctx, 5 );
  }
  return VVHIP_OK;
}

} // namespace

extern "C" {

int vvhip_mctf_error_batch( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_buf, int buf_stride, int width, int height,
                            int tap4, int bit_depth, const vvhip_mctf_item* d_items, int n, int32_t* d_out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width < 8 || height < 8 || width > 64 || height > 64 || ( width & 7 ) || ( height & 7 ) || n < 0 || bit_depth < 8 || bit_depth > 12 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_mctf_error_batch: block %dx%d must be a multiple of 8 in 8..64 (MCTF.cpp:1113)", width, height );
  if( n == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( mctfErrorBatchKernel, dim3( n ), dim3( 64 ), 0, ctx->stream, d_org, org_stride, d_buf, buf_stride, width, height, tap4,
                      ( 1 << bit_depth ) - 1, d_items, d_out );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_mctf_calc_var_batch( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, int width, int height, const int32_t* d_off, int n, int64_t* d_out_x256 )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width < 1 || height < 1 || width > 128 || height > 128 || n < 0 ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_mctf_calc_var_batch: bad size" );
  if( n == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( calcVarBatchKernel, dim3( n ), dim3( 64 ), 0, ctx->stream, d_org, org_stride, width, height, d_off, ( int64_t* ) d_out_x256 );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_extend_border( vvhip_ctx* ctx, int16_t* d_plane, int stride, int width, int height, int pad )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( width < 1 || height < 1 || pad < 0 || stride < width + 2 * pad ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_extend_border: bad geometry" );
  if( pad == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( extendLRKernel, dim3( height ), dim3( 128 ), 0, ctx->stream, d_plane, stride, width, height, pad );
  hipLaunchKernelGGL( extendTBKernel, dim3( 2 * pad ), dim3( 256 ), 0, ctx->stream, d_plane, stride, width, height, pad );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_mctf_subsample( vvhip_ctx* ctx, const int16_t* d_src, int src_stride, int src_width, int src_height, int16_t* d_dst, int dst_stride, int pad )
{
  if( !ctx ) return VVHIP_E_ARG;
  const int nw = src_width / 2, nh = src_height / 2;
  if( nw < 1 || nh < 1 || dst_stride < nw + 2 * pad ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_mctf_subsample: bad geometry" );
  hipLaunchKernelGGL( subsampleKernel, dim3( ( nw + 255 ) / 256, nh ), dim3( 256 ), 0, ctx->stream, d_src, src_stride, d_dst, dst_stride, nw, nh );
  VVHIP_LAUNCH_CHECK( ctx );
  return vvhip_extend_border( ctx, d_dst, dst_stride, nw, nh, pad );
}

int vvhip_mctf_init_mvs( vvhip_ctx* ctx, vvhip_mv* d_mvs, int count )
{
  if( !ctx || count < 0 ) return VVHIP_E_ARG;
  if( count == 0 ) return VVHIP_OK;
  hipLaunchKernelGGL( initMvsKernel, dim3( ( count + 255 ) / 256 ), dim3( 256 ), 0, ctx->stream, d_mvs, count );
  VVHIP_LAUNCH_CHECK( ctx );
  return VVHIP_OK;
}

int vvhip_mctf_me_level( vvhip_ctx* ctx, const int16_t* d_org, int org_stride, const int16_t* d_buf, int buf_stride, int width, int height, int block_size,
                         const vvhip_mv* d_prev, int prev_w, int prev_h, int factor, int double_res, int search_pattern, int low_res_filter,
                         int bit_depth, int unit_size, vvhip_mv* d_mvs, int mvs_w, int mvs_h )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( ( block_size != 8 && block_size != 16 && block_size != 32 ) || width < 8 || height < 8 || bit_depth < 8 || bit_depth > 10 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_mctf_me_level: block size %d (8/16/32), bit depth %d (8..10, MCTF.cpp:1313)", block_size, bit_depth );
  const int nbx = ( width - 8 ) / block_size + 1, nby = ( height - 8 ) / block_size + 1;
  const size_t need = 7 * sizeof( unsigned long long ) * ( ( size_t ) nbx * nby + 64 ) + 256;      // per block: one FixRec + three list slots + a stamp (52 bytes <= 7 granules; the sweep's NbRec needs 3)
  int rc = ensureScratch( ctx, need );
  if( rc ) return rc;
  int* d_abort = reinterpret_cast<int*>( ctx->d_scratch );
  unsigned long long* d_gr = reinterpret_cast<unsigned long long*>( reinterpret_cast<char*>( ctx->d_scratch ) + 256 );
  VVHIP_CHECK_HIP( ctx, hipMemsetAsync( d_abort, 0, 256, ctx->stream ) );
  MeRefs R = {};
  R.buf[0] = d_buf; R.prev[0] = d_prev; R.mvs[0] = d_mvs; R.gran[0] = d_gr;
  rc = meLevel( ctx, d_org, org_stride, buf_stride, width, height, block_size, R, 1, 0, prev_w, prev_h, factor, double_res, search_pattern,
                low_res_filter, bit_depth, unit_size, mvs_w, mvs_h, d_abort );
  if( rc ) return rc;
  int aborted = 0;
  VVHIP_CHECK_HIP( ctx, hipMemcpyAsync( &aborted, d_abort, sizeof( int ), hipMemcpyDeviceToHost, ctx->stream ) );
  VVHIP_CHECK_HIP( ctx, vvhip_wait_stream( ctx ) );
  if( aborted ) return vvhip_fail( ctx, VVHIP_E_HIP, "vvhip_mctf_me_level: wavefront hand-off timed out" );
  return VVHIP_OK;
}

int vvhip_mctf_set_stats( vvhip_ctx* ctx, int on )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( on && !ctx->d_mctfStats ) VVHIP_CHECK_HIP( ctx, hipMalloc( reinterpret_cast<void**>( &ctx->d_mctfStats ), 24 * sizeof( unsigned long long ) ) );
  if( on ) VVHIP_CHECK_HIP( ctx, hipMemsetAsync( ctx->d_mctfStats, 0, 24 * sizeof( unsigned long long ), ctx->stream ) );
  if( !on && ctx->d_mctfStats ) { VVHIP_CHECK_HIP( ctx, vvhip_wait_stream( ctx ) ); ( void ) hipFree( ctx->d_mctfStats ); ctx->d_mctfStats = nullptr; }
  return VVHIP_OK;
}

int vvhip_mctf_set_timing( vvhip_ctx* ctx, int on )
{
  if( !ctx ) return VVHIP_E_ARG;
  ctx->mctfTiming = on != 0; ctx->mctfEvTag.clear();
  return VVHIP_OK;
}

int vvhip_mctf_last_times( vvhip_ctx* ctx, float* ms5 )
{
  if( !ctx || !ms5 || !ctx->mctfTiming || ctx->mctfEvTag.size() < 2 ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipEventSynchronize( ctx->mctfEv[ctx->mctfEvTag.size() - 1] ) );
  for( int k = 0; k < 5; k++ ) ms5[k] = 0.f;
  for( size_t k = 1; k < ctx->mctfEvTag.size(); k++ )
  {
    float ms = 0.f;
    VVHIP_CHECK_HIP( ctx, hipEventElapsedTime( &ms, ctx->mctfEv[k - 1], ctx->mctfEv[k] ) );
    const int tag = ctx->mctfEvTag[k];
    ms5[tag == 2 ? 0 : tag == 3 ? 1 : tag == 4 ? 2 : tag == 5 ? 3 : 4] += ms;
  }
  return VVHIP_OK;
}

int vvhip_mctf_get_stats( vvhip_ctx* ctx, uint64_t* out24 )
{
  if( !ctx || !out24 ) return VVHIP_E_ARG;
  if( !ctx->d_mctfStats ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_mctf_get_stats: the counters are off (vvhip_mctf_set_stats)" );
  VVHIP_CHECK_HIP( ctx, hipMemcpyAsync( out24, ctx->d_mctfStats, 24 * sizeof( unsigned long long ), hipMemcpyDeviceToHost, ctx->stream ) );
  VVHIP_CHECK_HIP( ctx, vvhip_wait_stream( ctx ) );
  return VVHIP_OK;
}

static int mctfMotionEstimation( vvhip_ctx* ctx, const int16_t* d_cur, const int16_t* const* d_refs, int n_refs, int stride, int width, int height,
                                 int pad, int bit_depth, int unit_size, int mctf_speed, int add_level, vvhip_mv* const* d_mvs_out, bool wait );

int vvhip_mctf_motion_estimation( vvhip_ctx* ctx, const int16_t* d_cur, const int16_t* const* d_refs, int n_refs, int stride, int width, int height,
                                  int pad, int bit_depth, int unit_size, int mctf_speed, int add_level, vvhip_mv* const* d_mvs_out )
{
  return mctfMotionEstimation( ctx, d_cur, d_refs, n_refs, stride, width, height, pad, bit_depth, unit_size, mctf_speed, add_level, d_mvs_out, true );
}

int vvhip_mctf_motion_estimation_async( vvhip_ctx* ctx, const int16_t* d_cur, const int16_t* const* d_refs, int n_refs, int stride, int width, int height,
                                        int pad, int bit_depth, int unit_size, int mctf_speed, int add_level, vvhip_mv* const* d_mvs_out )
{
  return mctfMotionEstimation( ctx, d_cur, d_refs, n_refs, stride, width, height, pad, bit_depth, unit_size, mctf_speed, add_level, d_mvs_out, false );
}

static int mctfMotionEstimation( vvhip_ctx* ctx, const int16_t* d_cur, const int16_t* const* d_refs, int n_refs, int stride, int width, int height,
                                 int pad, int bit_depth, int unit_size, int mctf_speed, int add_level, vvhip_mv* const* d_mvs_out, bool wait )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( n_refs < 0 || width < 64 || height < 64 || pad < 128 || stride < width + 2 * pad || ( unit_size != 8 && unit_size != 16 ) || bit_depth < 8 || bit_depth > 10 || mctf_speed < 0 || mctf_speed > 4 )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_mctf_motion_estimation: bad arguments (%dx%d stride %d pad %d unit %d bitDepth %d speed %d)", width, height, stride, pad, unit_size, bit_depth, mctf_speed );
  if( n_refs == 0 ) return VVHIP_OK;
  const int lowRes = mctf_speed > 0;                                        // MCTF.cpp:598
  const int pttrn  = mctf_speed > 0 ? ( mctf_speed >= 3 ? 2 : 1 ) : 0;      // MCTF.cpp:599
  const int u = unit_size;
  const int P = 128;                                                        // MCTF_PADDING of the internal pyramid planes
  // ---- scratch layout: [abort 256 B][granules][pyramid planes cur L1..L3][per ref: planes L1..L3][per ref: mv fields L(-1)..L2]
  int lw[4], lh[4], ls[4];
  lw[0] = width; lh[0] = height; ls[0] = stride;
  for( int l = 1; l < 4; l++ ) { lw[l] = lw[l - 1] / 2; lh[l] = lh[l - 1] / 2; ls[l] = ( lw[l] + 2 * P + 7 ) & ~7; }
  size_t planeElems[4] = { 0, 0, 0, 0 };
  for( int l = 1; l < 4; l++ ) planeElems[l] = ( size_t ) ls[l] * ( lh[l] + 2 * P );
  const size_t pyrElems = planeElems[1] + planeElems[2] + planeElems[3];
  const int fw[4] = { width / ( u * 16 ) + 1, width / ( u * 8 ) + 1, width / ( u * 4 ) + 1, width / ( u * 2 ) + 1 };     // MCTF.cpp:682-684,694
  const int fh[4] = { height / ( u * 16 ) + 1, height / ( u * 8 ) + 1, height / ( u * 4 ) + 1, height / ( u * 2 ) + 1 };
  size_t fieldElems = 0;
  for( int k = 0; k < 4; k++ ) fieldElems += ( size_t ) fw[k] * fh[k];
  const int outW = ( width + u - 1 ) / u, outH = ( height + u - 1 ) / u;      // MCTF.cpp:671-672
  const size_t granElems = 7 * ( ( size_t ) outW * outH + 64 );      // per reference: one FixRec + three list slots + a stamp per block of the finest level (52 bytes <= 7 granules)
  size_t off = 256;
  const size_t offGran = off;  off += granElems * sizeof( unsigned long long ) * ( size_t ) ( n_refs < ME_MAX_REFS ? n_refs : ME_MAX_REFS );
  off = ( off + 255 ) & ~( size_t ) 255;
  const size_t offPyr = off;   off += ( size_t ) ( 1 + n_refs ) * pyrElems * sizeof( int16_t );
  off = ( off + 255 ) & ~( size_t ) 255;
  const size_t offFld = off;   off += ( size_t ) n_refs * fieldElems * sizeof( vvhip_mv );
  int rc = ensureScratch( ctx, off );
  if( rc ) return rc;
  char* base = reinterpret_cast<char*>( ctx->d_scratch );
  int* d_abort = reinterpret_cast<int*>( base );
  unsigned long long* d_gr = reinterpret_cast<unsigned long long*>( base + offGran );
  ctx->mctfEvTag.clear();
  meMark( ctx, 0 );
  VVHIP_CHECK_HIP( ctx, hipMemsetAsync( d_abort, 0, 256, ctx->stream ) );

  auto planePtr = [&]( int pic, int l ) -> int16_t* {   // pic 0 = current, 1.. = references; l = 1..3; returns pointer to sample (0,0)
    int16_t* p = reinterpret_cast<int16_t*>( base + offPyr ) + ( size_t ) pic * pyrElems;
    for( int k = 1; k < l; k++ ) p += planeElems[k];
    return p + ( size_t ) P * ls[l] + P;
  };
  int usedHandOff = 0;
  const int levels = add_level ? 3 : 2;
  for( int l = 1; l <= levels; l++ )                                                     // MCTF.cpp:689-690,696,779-784: one level of every picture per launch
    for( int p0 = 0; p0 <= n_refs; p0 += ME_MAX_PICS )
    {
      const int np = n_refs + 1 - p0 < ME_MAX_PICS ? n_refs + 1 - p0 : ME_MAX_PICS;
      MePlanes Q = {};
      for( int k = 0; k < np; k++ )
      {
        const int pic = p0 + k;
        Q.src[k] = l == 1 ? ( pic == 0 ? d_cur : d_refs[pic - 1] ) : planePtr( pic, l - 1 );
        Q.dst[k] = planePtr( pic, l );
      }
      hipLaunchKernelGGL( pyramidLevelKernel, dim3( ( lw[l] + 2 * P + 255 ) / 256, lh[l] + 2 * P, np ), dim3( 256 ), 0, ctx->stream, Q, ls[l - 1], ls[l], lw[l], lh[l], P );
      VVHIP_LAUNCH_CHECK( ctx );
    }
  bool fieldsInitialised = false;
  for( int r0 = 0; r0 < n_refs; r0 += ME_MAX_REFS )                                    // all references of a chunk advance through the hierarchy together
  {
    const int nr = n_refs - r0 < ME_MAX_REFS ? n_refs - r0 : ME_MAX_REFS;
    vvhip_mv* f[ME_MAX_REFS][4];
    MeRefs R = {};
    for( int k = 0; k < nr; k++ )
    {
      f[k][0] = reinterpret_cast<vvhip_mv*>( base + offFld ) + ( size_t ) ( r0 + k ) * fieldElems;
      for( int l = 1; l < 4; l++ ) f[k][l] = f[k][l - 1] + ( size_t ) fw[l - 1] * fh[l - 1];
      R.mvs[k] = d_mvs_out[r0 + k];
      R.gran[k] = d_gr + ( size_t ) k * granElems;
    }
    // (the first chunk's launch also initialises the level fields of ALL references: blockIdx.y = nr)
    hipLaunchKernelGGL( initMvsAllKernel, dim3( ( outW * outH + 255 ) / 256, nr + ( fieldsInitialised ? 0 : 1 ) ), dim3( 256 ), 0, ctx->stream, R, nr, outW * outH,
                        reinterpret_cast<vvhip_mv*>( base + offFld ), ( int ) ( fieldElems * n_refs ) );
    fieldsInitialised = true;
    VVHIP_LAUNCH_CHECK( ctx );
    auto level = [&]( int l, int bs, int inField, int outField, int factor, int dbl ) -> int      // l: pyramid level (0 = full resolution); fields 0..3, 4 = result
    {
      for( int k = 0; k < nr; k++ )
      {
        R.buf[k]  = l ? planePtr( r0 + k + 1, l ) : d_refs[r0 + k];
        R.prev[k] = inField < 0 ? nullptr : f[k][inField];
        R.mvs[k]  = outField == 4 ? d_mvs_out[r0 + k] : f[k][outField];
      }
      return meLevel( ctx, l ? planePtr( 0, l ) : d_cur, ls[l], ls[l], lw[l], lh[l], bs, R, nr, granElems, inField < 0 ? 0 : fw[inField], inField < 0 ? 0 : fh[inField], factor, dbl, pttrn, lowRes,
                      bit_depth, u, outField == 4 ? outW : fw[outField], outField == 4 ? outH : fh[outField], d_abort, &usedHandOff );
    };
    if( add_level ) { rc = level( 3, 2 * u, -1, 0, 1, 0 ); if( rc ) return rc; }        // MCTF.cpp:692-699
    rc = level( 2, 2 * u, add_level ? 0 : -1, 1, 2, 0 );  if( rc ) return rc;           // :698 / :702
    rc = level( 1, 2 * u, 1, 2, 2, 0 );                   if( rc ) return rc;           // :704
    rc = level( 0, 2 * u, 2, 3, 2, 0 );                   if( rc ) return rc;           // :705
    rc = level( 0, u, 3, 4, 1, 1 );                       if( rc ) return rc;           // :707
  }
  meMark( ctx, 6 );
  // the sweep kernels cannot fail; only the row hand-off of fields whose diagonals exceed a workgroup has an abort flag to read — the asynchronous entry waits for it too
  if( !wait && !usedHandOff ) return VVHIP_OK;
  int aborted = 0;
  if( usedHandOff ) VVHIP_CHECK_HIP( ctx, hipMemcpyAsync( &aborted, d_abort, sizeof( int ), hipMemcpyDeviceToHost, ctx->stream ) );
  VVHIP_CHECK_HIP( ctx, vvhip_wait_stream( ctx ) );
  if( aborted ) return vvhip_fail( ctx, VVHIP_E_HIP, "vvhip_mctf_motion_estimation: wavefront hand-off timed out" );
  return VVHIP_OK;
}

} // extern "C"
