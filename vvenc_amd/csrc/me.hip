// me.hip — motion-search plans: the distortion work of one picture's inter search as a handful of launches (include/vvenc_hip.h, "Motion-search plans").
//
// Shaped by the work lists the reference encoder really produces (recorded from it: bindings/vvenc/vvenc_hip_recorder.*, vvenc_amd/recorded.py), not by uniform
// synthetic ones: an InterSearch::xMotionEstimation call (EncoderLib/InterSearch.cpp:1976-2130) scores ~20 integer positions that lie within a few samples of each
// other (start points, then the 4-point diamond and square at distance 1, :2385-2410; HALF of them repeat a position of the same call — scored once), then one or two xPatternRefinement stages
// (:760-880) of <= 9 sub-pel positions; at preset faster two thirds of all sample pairs belong to 64x64 blocks, preset medium (CTU 128 + multi-type tree) brings every
// rectangular shape 4..128 and GEO's masked SADs.  Kinds of work, one wave per unit:
//   integer job   the bounding window of the job's candidates is staged ONCE in LDS (samples biased for v_sad_u16), the original block next to it; every candidate is then
//                 scored from LDS by a team of lanes (dword reads at the even address below the candidate + v_alignbit for odd displacements): the L1 sees the window once
//                 per job instead of once per candidate.  Three LDS classes, a launch each when they differ much.      xGetSAD*, CommonLib/RdCost.cpp:301-644
//   stage bundle  a few (stage, <= 32 rows x <= 64 columns) units of one unit width.  Per unit: horizontal pass of the <= 3 distinct horizontal positions straight from the
//                 plane into LDS (14-bit intermediates, InterpolationFilter.cpp:356-441), then a team of lanes per (position, Hadamard tile): vertical pass of the lane's own
//                 row(s) out of LDS, difference to the original block, the tile's transform in registers and across the team (DPP) — the prediction never exists in memory.
//                 The reference's tile ladder: 16x8, 8x16, 8x4, 4x8, 16x16_fast, 8x8.                    xGetHADs<fast>, RdCost.cpp:1818-1938; tiles :1028-1766
//                 Six launch classes: tap support x (the fast presets' square shapes with compile-time tile constants / any shape).
//   item bundle   plain table calls (merge / AMVP / intra candidates, residual SSE) on blocks of any two planes or pools: lane teams on row chunks (SAD, SSE, masked SAD),
//                 lane teams per Hadamard tile, one lane per 4x4 block / 2x2 tile.  Lean instance for what the fast presets call, generic instance for the rest.
// These are short-lived waves: what they cost is their chain of dependent memory accesses and — measured at 1.8 ns per wave instruction and SIMD for every form used here
// (tools/exp/valu_rate.hip) — their instruction COUNT: scaled taps so that shifts become byte selections (v_perm_b32), v_mad_i32_i16 with op_sel, packed butterfly stages,
// scalar bases with 32-bit offsets, wave-uniform branches.  Hence also: job tables in schedule order (no order -> record
// indirection), every global request of a job issued before the first wait, candidate records / plane table / tap tables staged in LDS once per wave, per-unit derived data
// precomputed by plan creation, a cap on the serial work of one workgroup (candidates per window, row groups per bundle, units per wave), and — round 4 — every launch's
// workgroups in XCD-band order (xcdBandOrder): each XCD's private L2 streams one horizontal band of the picture.
// Results are bit-exact with the reference's table entries (tests: the CPU restatement directly over every shape, tests/test_gpu_me_shapes.py; the recorded costs of the real encoder,
// tests/test_gpu_replay.py; the per-function kernels of dist.hip / interp.hip).
#include <stdlib.h>
#ifndef VVHIP_ME_HU
#define VVHIP_ME_HU 2
#endif
// register budgets (waves per SIMD the compiler must make room for; 512 registers per lane and SIMD): without them the allocator takes what the default occupancy target
// leaves (87 / 68 registers) although it needs no spill at 80 / 52 — and these kernels are short-lived waves whose latencies only more resident waves hide
#ifndef VVHIP_ME_STAGE_WAVES
#define VVHIP_ME_STAGE_WAVES 6
#endif
#ifndef VVHIP_ME_ITEM_WAVES
#define VVHIP_ME_ITEM_WAVES 8
#endif
#include <string.h>
#include <algorithm>
#include <vector>
#include "common.h"

struct vvhip_me_plan
{
  int bitDepth = 0, nCands = 0, nStages = 0, nItems = 0, nMaskItems = 0, maxPlane = 0;
  int wavesInt = 0, wavesStage = 0, wavesItem = 0, wavesItemMain = 0, ldsInt = 0, ldsStage = 0;      // wavesItemMain: the leading item waves the lean body takes (the rest: generic body, own launch)
  int intBig = 0, ldsIntSmall = 0;          // the first intBig windows need up to ldsInt bytes of LDS, the others at most ldsIntSmall (two launches: small blocks keep their occupancy)
  int intLarge = 0, ldsIntMid = 0;          // of the intBig windows, the first intLarge are the > 24 KB class; the others need at most ldsIntMid
  bool stageAtomic = false;                 // some stages add their units' sums with atomics: the cost array is cleared in front of the stage launches
  bool intSplit = false;                    // the large windows need far more LDS than four small ones: two launches (the small windows keep their occupancy)
  bool timing = false; hipEvent_t ev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };      // optional per-kernel events of the last run (vvhip_me_plan_set_timing)
  // stage bundles per launch class, in schedule order: class = 2 * tap support (4-tap search set, 6 taps / alternative half-pel, 8 taps) + generic (0: the square 8..64 blocks of
  // the fast presets, every tile quantity a compile-time constant; 1: any shape)
  int stageSetWaves[6] = { 0, 0, 0, 0, 0, 0 }, stageSetLds[6] = { 0, 0, 0, 0, 0, 0 };
  void* d_blob = nullptr;                  // one allocation: every table below
  const void* d_intJobs = nullptr; const void* d_cands = nullptr; const void* d_candOut = nullptr; const void* d_stageJobs = nullptr; const void* d_stageOrder = nullptr; const void* d_stageWaves = nullptr;
  const void* d_items = nullptr; const void* d_itemOrder = nullptr; const void* d_itemWaves = nullptr; const void* d_tapTables = nullptr; const void* d_maskItems = nullptr;
};

namespace {

typedef uint32_t u32x2 __attribute__( ( ext_vector_type( 2 ) ) );
typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
typedef short s16x2 __attribute__( ( ext_vector_type( 2 ) ) );
struct __attribute__( ( packed, aligned( 2 ) ) ) U8  { u32x2 v; };
struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };
#define ME_GLOBAL __attribute__( ( address_space( 1 ) ) )      /* a pointer that went through LDS (the item kernels' plane table) is a generic pointer to the compiler: flat loads; these say "global" */
__device__ __forceinline__ u32x2 ld8( const int16_t* p )  { return ( ( const ME_GLOBAL U8* ) p )->v; }
__device__ __forceinline__ u32x4 ld16( const int16_t* p ) { return ( ( const ME_GLOBAL U16* ) p )->v; }
// 16 bytes at a 32-bit unsigned byte offset from a WAVE-UNIFORM base: global_load with scalar base + vector offset — one multiply-add per address instead of a 64-bit
// multiply and three 64-bit adds per lane
__device__ __forceinline__ u32x4 ld16o( const char* base, uint32_t byteOff ) { return reinterpret_cast<const U16*>( base + byteOff )->v; }
__device__ __forceinline__ u32x2 ld8o( const char* base, uint32_t byteOff ) { return reinterpret_cast<const U8*>( base + byteOff )->v; }
__device__ __forceinline__ int lo16( uint32_t v ) { return ( int ) ( int16_t ) ( v & 0xffffu ); }
__device__ __forceinline__ int hi16( uint32_t v ) { return ( int ) ( ( int32_t ) v >> 16 ); }
__device__ __forceinline__ uint32_t pkAdd( uint32_t a, uint32_t b ) { return __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2, a ) + __builtin_bit_cast( s16x2, b ) ); }
__device__ __forceinline__ uint32_t pkSub( uint32_t a, uint32_t b ) { return __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2, a ) - __builtin_bit_cast( s16x2, b ) ); }
__device__ __forceinline__ uint32_t pack2( int lo, int hi ) { return ( ( uint32_t ) lo & 0xffffu ) | ( ( uint32_t ) hi << 16 ); }
constexpr uint32_t BIAS = 0x80008000u;      // signed int16 pair -> unsigned order (v_sad_u16 on biased operands is |a - b| of the signed values)

struct MePlanes { const int16_t* p[16]; int stride[16]; };

// ---- plan-side job records (device) ------------------------------------------------------------------------------------------------------------
struct IntJob   { int32_t orgOff, refOff; int16_t w, h; uint8_t orgPlane, refPlane, subShift, pad; int16_t minDx, minDy, winW, winH; int32_t firstCand, nCand; };   // one window
struct PlanCand { int16_t dx, dy; uint32_t out; };      // out: the candidate's first entry in the plan's output list | ( entries - 1 ) << 24 — every entry is a position of the caller's cost array (identical positions of a job are scored once)
struct WaveSpan { int32_t first, count; };                                                                                                               // into an order array
// one stage unit of the schedule: the job + what every lane of the wave used to re-derive from it per unit (the evaluated positions grouped by their horizontal displacement,
// the <= 3 distinct displacements and how many positions use each): plan creation does it once (the unit skeleton)
struct StageUnit
{
  vvhip_me_stage_job j;
  int16_t hx[3]; uint8_t nHor, nPos;          // distinct horizontal displacements (1/16 sample) in first-seen order
  uint8_t cnt0, cnt1, cnt2, pad;              // positions per displacement
  int32_t order;                               // stage index | band << 24 | half << 27 | ST_UNIT_CONT | ST_UNIT_MORE
  int32_t pos[9];                              // slot -> k | ( tx + 64 ) << 8 | ( ty + 64 ) << 20, grouped by displacement (variant 0 first)
  int32_t pad2[3];
};
static_assert( sizeof( StageUnit ) == 88, "StageUnit layout" );
struct MeArgs
{
  const IntJob* intJobs; const PlanCand* cands; const int32_t* candOut; int wavesInt;
  const StageUnit* stageUnits; const WaveSpan* stageWaves; int wavesStage;
  const int32_t* tapTables;      // [6 = filter_mode * 2 + alt_hpel][192]: stageTapTables
  const vvhip_me_item* items; const int32_t* itemOrder; const WaveSpan* itemWaves; int wavesItem;
  const vvhip_me_mask_item* maskItems;      // in schedule order; their waves follow the plain items' (WaveSpan.count < 0), their costs follow the plain items' costs
  uint64_t* candCost; uint64_t* stageCost; uint64_t* itemCost;
  int bitDepth;
};

// 4x4 Hadamard of 16 differences (xCalcHADs4x4, RdCost.cpp:1028-1124): sum of |coefficients| with the DC a quarter, then ( satd + 1 ) >> 1
__device__ __forceinline__ uint32_t hadamard4x4( int ( &d )[16] )
{
#pragma unroll
  for( int r = 0; r < 4; r++ )
  {
    const int a = d[4 * r], b = d[4 * r + 1], c = d[4 * r + 2], e = d[4 * r + 3];
    const int s0 = a + b, s1 = a - b, s2 = c + e, s3 = c - e;
    d[4 * r] = s0 + s2; d[4 * r + 1] = s1 + s3; d[4 * r + 2] = s0 - s2; d[4 * r + 3] = s1 - s3;
  }
  uint32_t s = 0;
#pragma unroll
  for( int c = 0; c < 4; c++ )
  {
    const int a = d[c], b = d[4 + c], e = d[8 + c], f = d[12 + c];
    const int s0 = a + b, s1 = a - b, s2 = e + f, s3 = e - f;
    const int v0 = s0 + s2, v1 = s1 + s3, v2 = s0 - s2, v3 = s1 - s3;
    s += ( c == 0 ? ( ( uint32_t ) abs( v0 ) >> 2 ) : ( uint32_t ) abs( v0 ) ) + ( uint32_t ) abs( v1 ) + ( uint32_t ) abs( v2 ) + ( uint32_t ) abs( v3 );
  }
  return ( s + 1 ) >> 1;
}

// =================================================================================================================================================
// (A) integer candidates of one window
// =================================================================================================================================================
// LDS row pitch of a window in samples: an ODD number of dwords.  A candidate row is read as five consecutive dwords per lane (ds_read_b32: 32-lane groups, bank = dword mod 32);
// the lanes of a group sit 4 dwords apart along a row and one row apart across rows, so with a pitch that is a multiple of 4 dwords every lane of the group hits one of 8 banks
// (4-way conflict on every read — the kernel was LDS-cycle-bound), with an odd pitch the rows land on different residues and the group covers all 32 banks
__device__ __forceinline__ int winPitch( int winW ) { return 2 * ( ( ( winW + 3 ) >> 1 ) | 1 ); }

// ONE_WAVE: the job belongs to one wave of the workgroup (small windows: four jobs per 256-thread workgroup, each in its own LDS slice, wave-level synchronisation only)
template<bool ONE_WAVE>
__device__ __forceinline__ void intBody( const MePlanes& P, const MeArgs& a, int wave, int16_t* lds )
{
  const IntJob j = a.intJobs[wave];
  const int tid = ONE_WAVE ? ( int ) ( threadIdx.x & 63 ) : ( int ) threadIdx.x, nthr = ONE_WAVE ? 64 : ( int ) blockDim.x, lane = tid & 63;
  const int w = j.w, ss = j.subShift, rowsEff = j.h >> ss, lpr = w >> 3, lprShift = 31 - __builtin_clz( lpr );
  const int pitch = winPitch( j.winW ), half0 = ( j.winH + 1 ) >> 1;
  int16_t* win = lds;
  int16_t* orgL = lds + ( ( j.winH * pitch + 7 ) & ~7 );                 // rowsEff x w, compact (16-byte rows)
  PlanCand* candL = reinterpret_cast<PlanCand*>( orgL + rowsEff * w );    // the job's candidates (8 bytes each)
  {
    // every global request of the job is issued before the first one is waited for: the candidate records, the original block (<= 4 chunks per lane: any block up to
    // 64x64, 128x128 with row sub-sampling), then the window in batches of four chunks per lane — the job is one memory latency + the LDS work, not four latencies in a row
    PlanCand myCand = { 0, 0, 0u };
    if( tid < j.nCand ) myCand = a.cands[j.firstCand + tid];
    const char* orgB = reinterpret_cast<const char*>( P.p[j.orgPlane] + j.orgOff );      // (wave-uniform bases, 32-bit offsets: ld16o)
    const int os = P.stride[j.orgPlane] ? P.stride[j.orgPlane] : w, m = rowsEff * lpr;      // (stride 0: a pool of compact blocks)
    u32x4 ov[4]; int oat[4];
#pragma unroll
    for( int q = 0; q < 4; q++ )
    {
      const int i = tid + nthr * q < m ? tid + nthr * q : 0, r = i >> lprShift, c = i & ( lpr - 1 );
      oat[q] = r * w + c * 8;
      ov[q] = ld16o( orgB, ( uint32_t ) ( __mul24( r << ss, os ) + c * 8 ) * 2u );
    }
    const char* refB = reinterpret_cast<const char*>( P.p[j.refPlane] + j.refOff + ( ptrdiff_t ) j.minDy * P.stride[j.refPlane] + j.minDx );
    // window rows: with row sub-sampling the even and the odd rows are two separate halves (a candidate reads every second row: consecutive rows of one half)
    const int cpr = ( j.winW + 2 + 7 ) >> 3, n = j.winH * cpr, rs = P.stride[j.refPlane];      // 16-byte chunks per row
    // i / cpr as ( i * ceil( 2^20 / cpr ) ) >> 20: exact for i < 4096 (a window has at most 152 rows x 20 chunks), both factors below 2^24 — full-rate 24-bit multiplies
    const uint32_t cprInv = ( ( 1u << 20 ) + ( uint32_t ) cpr - 1u ) / ( uint32_t ) cpr;                          // wave-uniform, once per job (cpr >= 2)
    for( int i0 = tid; i0 < n; i0 += 4 * nthr )                        // four loads in flight per lane (the loop is latency-bound otherwise)
    {
      u32x4 v[4]; int at[4], cc[4];
#pragma unroll
      for( int q = 0; q < 4; q++ )
      {
        const int i = i0 + nthr * q < n ? i0 + nthr * q : i0, r = ( int ) ( __umul24( ( uint32_t ) i, cprInv ) >> 20 ), c = i - __mul24( r, cpr );
        const int dr = ss ? ( ( r & 1 ) ? half0 : 0 ) + ( r >> 1 ) : r;
        at[q] = __mul24( dr, pitch ) + c * 8; cc[q] = c;
        v[q] = ld16o( refB, ( uint32_t ) ( __mul24( r, rs ) + c * 8 ) * 2u );
      }
#pragma unroll
      for( int q = 0; q < 4; q++ )
        if( i0 + nthr * q < n )
        {
          uint32_t* d = reinterpret_cast<uint32_t*>( win + at[q] );      // dword stores: the rows are not 16-byte aligned; the last chunk of a row is cut at the pitch
          const int left = ( pitch >> 1 ) - cc[q] * 4;
          d[0] = v[q].x ^ BIAS; if( left > 1 ) d[1] = v[q].y ^ BIAS; if( left > 2 ) d[2] = v[q].z ^ BIAS; if( left > 3 ) d[3] = v[q].w ^ BIAS;
        }
    }
#pragma unroll
    for( int q = 0; q < 4; q++ )
      if( tid + nthr * q < m ) { u32x4 x = ov[q]; x.x ^= BIAS; x.y ^= BIAS; x.z ^= BIAS; x.w ^= BIAS; *reinterpret_cast<u32x4*>( orgL + oat[q] ) = x; }
    for( int i0 = tid + 4 * nthr; i0 < m; i0 += nthr )                 // (blocks beyond four chunks per lane: 128x128 without row sub-sampling, large blocks of a one-wave job)
    {
      const int r = i0 >> lprShift, c = i0 & ( lpr - 1 );
      u32x4 x = ld16o( orgB, ( uint32_t ) ( __mul24( r << ss, os ) + c * 8 ) * 2u ); x.x ^= BIAS; x.y ^= BIAS; x.z ^= BIAS; x.w ^= BIAS;
      *reinterpret_cast<u32x4*>( orgL + r * w + c * 8 ) = x;
    }
    if( tid < j.nCand ) candL[tid] = myCand;
    for( int i = tid + nthr; i < j.nCand; i += nthr ) candL[i] = a.cands[j.firstCand + i];
  }
  if( ONE_WAVE ) { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  else __syncthreads();
  const int chunks = rowsEff * lpr;
  int lpc = 64; while( lpc > chunks ) lpc >>= 1;                     // lanes per candidate: a power of two <= min( 64, chunks )   (width, height and so chunks are powers of two)
  const int teams = nthr / lpc, lt = tid & ( lpc - 1 ), team = tid / lpc;
  for( int c0 = 0; c0 < j.nCand; c0 += teams )
  {
    const int ci = c0 + team;
    const bool valid = ci < j.nCand;
    const PlanCand cd = candL[valid ? ci : 0];
    // where the cost goes: the search scores its start point again and again (half of a recorded picture's integer positions repeat one of the same call) — plan creation
    // keeps one candidate per distinct position and the list of cost-array entries it stands for; lane k of the team stores entry k (requested here, ahead of the rows)
    const int myOut = ( valid && ( uint32_t ) lt <= ( cd.out >> 24 ) ) ? a.candOut[( cd.out & 0xffffffu ) + lt] : -1;
    const int x = cd.dx - j.minDx, y = cd.dy - j.minDy;
    const int rowBase = ss ? ( ( y & 1 ) ? half0 : 0 ) + ( y >> 1 ) : y;
    const int16_t* base = win + rowBase * pitch + ( x & ~1 );
    const uint32_t sh = ( x & 1 ) * 16;
    uint32_t sad = 0;
    // a lane's chunks are lpc apart: the same column piece, rowStep rows further down — both LDS addresses advance by constants (chunks and lpc are powers of two, lpc >= lpr)
    const uint32_t* pc = reinterpret_cast<const uint32_t*>( base + ( lt >> lprShift ) * pitch + ( lt & ( lpr - 1 ) ) * 8 );
    const u32x4* po = reinterpret_cast<const u32x4*>( orgL + ( lt >> lprShift ) * w + ( lt & ( lpr - 1 ) ) * 8 );
    const int rowStep = lpc >> lprShift, curStep = rowStep * ( pitch >> 1 ), orgStep = rowStep * ( w >> 3 );
#pragma unroll 2
    for( int it = chunks / lpc; it > 0; it-- )
    {
      const u32x4 o = *po;
      const uint32_t v0 = pc[0], v1 = pc[1], v2 = pc[2], v3 = pc[3], v4 = pc[4];
      sad = __builtin_amdgcn_sad_u16( __builtin_amdgcn_alignbit( v1, v0, sh ), o.x, sad );
      sad = __builtin_amdgcn_sad_u16( __builtin_amdgcn_alignbit( v2, v1, sh ), o.y, sad );
      sad = __builtin_amdgcn_sad_u16( __builtin_amdgcn_alignbit( v3, v2, sh ), o.z, sad );
      sad = __builtin_amdgcn_sad_u16( __builtin_amdgcn_alignbit( v4, v3, sh ), o.w, sad );
      pc += curStep; po += orgStep;
    }
    const uint32_t t = vvhipGroupSum32( sad, lpc, lane );
    if( myOut >= 0 ) a.candCost[myOut] = ( uint64_t ) t << ss;                    // RdCost.cpp:334
  }
}

// =================================================================================================================================================
// (B) sub-pel refinement stages
// =================================================================================================================================================
static const int8_t kLuma8[9][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { -1, 2, -5, 62, 8, -3, 1, 0 }, { -1, 3, -8, 60, 13, -4, 1, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
  { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 }, { -1, 4, -11, 40, 40, -11, 4, -1 } };
static const int8_t kLuma6[9][8] = {
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { 0, 1, -5, 62, 8, -3, 1, 0 }, { 0, 2, -8, 60, 13, -4, 1, 0 }, { 0, 3, -10, 58, 17, -5, 1, 0 },
  { 0, 3, -11, 52, 26, -8, 2, 0 }, { 0, 2, -9, 47, 31, -10, 3, 0 }, { 0, 3, -11, 45, 34, -10, 3, 0 }, { 0, 3, -11, 40, 40, -11, 3, 0 } };
static const int8_t kAltHpel[8] = { 0, 3, 9, 20, 20, 9, 3, 0 };
static const int8_t kChroma4[17][4] = {
  { 0, 64, 0, 0 }, { -1, 63, 2, 0 }, { -2, 62, 4, 0 }, { -2, 60, 7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 }, { -4, 55, 15, -2 }, { -4, 54, 16, -2 },
  { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 } };

// the 8 window taps (entry k multiplies the sample at offset k - 3) of phase `frac` (1/16 sample) in the tap set the stage's search uses:
// filter_mode 0 = 8 taps, 1 = 6 taps, 2 = the 4 chroma taps at twice the phase (m_meReduceTap, InterpolationFilter.cpp:586-593); alt half-pel at phase 8
// HOST: the kernels read these from six precomputed tables in the plan (a tap looked up in constant memory per unit was a dependent memory access on every short-lived unit)
static int stageTap( int frac, int k, int filterMode, int altHpel )
{
  if( altHpel && frac == 8 ) return kAltHpel[k];
  if( filterMode == 2 )
  {
    const int ph = frac << 1;
    return ( k < 2 || k > 5 ) ? 0 : ( ph <= 16 ? kChroma4[ph][k - 2] : kChroma4[32 - ph][5 - k] );
  }
  const int p = frac <= 8 ? frac : 16 - frac, kk = frac <= 8 ? k : 7 - k;
  return filterMode == 0 ? kLuma8[p][kk] : kLuma6[p][kk];
}

// HOST: the six tables of a plan, per (filter_mode, alternative half-sample filter) 192 dwords:
//   [0 .. 127]   16 phases x 8 window taps for the SECOND (vertical) pass, scaled by 2^( 16 - shift2 ), shift2 = 6 + headRoom: the filtered sample is the accumulator's upper half (predRow)
//   [128 .. 191] 16 phases x 4 packed tap pairs ( K0 + 2 i, K0 + 2 i + 1 ) of the tap support the table's kernel instance uses, for the first (horizontal) pass, scaled by
//                2^( 8 - shift1 ), shift1 = 6 - headRoom (<= 64 x 256: 16 bits): the first-pass value is bytes 1..2 of the accumulator (firstPass)
static std::vector<int32_t> stageTapTables( int bitDepth )
{
  const int headRoom = 14 - bitDepth > 2 ? 14 - bitDepth : 2, vScale = 1 << ( 10 - headRoom ), hScale = 1 << ( 2 + headRoom );      // 2^( 16 - shift2 ), 2^( 8 - shift1 )
  std::vector<int32_t> tapTab( 6 * 192, 0 );
  for( int mode = 0; mode < 3; mode++ ) for( int alt = 0; alt < 2; alt++ )
  {
    int32_t* t = &tapTab[( mode * 2 + alt ) * 192];
    const int set = ( mode == 2 && !alt ) ? 0 : ( mode == 0 ? 2 : 1 ), k0 = set == 0 ? 2 : ( set == 1 ? 1 : 0 ), np = set == 0 ? 2 : ( set == 1 ? 3 : 4 );
    for( int f = 0; f < 16; f++ )
    {
      for( int k = 0; k < 8; k++ ) t[f * 8 + k] = stageTap( f, k, mode, alt ) * vScale;
      for( int i = 0; i < 4; i++ ) t[128 + f * 4 + i] = i < np ? ( int32_t ) ( ( ( uint32_t ) ( stageTap( f, k0 + 2 * i, mode, alt ) * hScale ) & 0xffffu ) | ( ( uint32_t ) ( stageTap( f, k0 + 2 * i + 1, mode, alt ) * hScale ) << 16 ) ) : 0;
    }
  }
  return tapTab;
}

// position k of a stage -> displacement in 1/16 sample from the stage's integer base: ( refine[k] + base ) * iFrac quarter samples
// refinement offsets s_acMvRefineH / s_acMvRefineQ (InterSearch.cpp:67-91) as 2-bit fields ( offset + 1 ) of literals: no table in memory
//   H: (0,0) (0,-1) (0,1) (-1,0) (1,0) (-1,-1) (1,-1) (-1,1) (1,1)      Q: (0,0) (0,-1) (0,1) (-1,-1) (1,-1) (-1,0) (1,0) (-1,1) (1,1)
__device__ __forceinline__ void stagePos( const vvhip_me_stage_job& j, int k, int& tx, int& ty )
{
  constexpr uint32_t RX  = 1u | 1u << 2 | 1u << 4 | 0u << 6 | 2u << 8 | 0u << 10 | 2u << 12 | 0u << 14 | 2u << 16;        // both tables have the same x offsets
  constexpr uint32_t RYH = 1u | 0u << 2 | 2u << 4 | 1u << 6 | 1u << 8 | 0u << 10 | 0u << 12 | 2u << 14 | 2u << 16;
  constexpr uint32_t RYQ = 1u | 0u << 2 | 2u << 4 | 0u << 6 | 0u << 8 | 1u << 10 | 1u << 12 | 2u << 14 | 2u << 16;
  const int rx = ( int ) ( ( RX >> ( 2 * k ) ) & 3u ) - 1, ry = ( int ) ( ( ( j.i_frac == 2 ? RYH : RYQ ) >> ( 2 * k ) ) & 3u ) - 1;
  tx = ( rx + j.base_qx ) * j.i_frac * 4; ty = ( ry + j.base_qy ) * j.i_frac * 4;
}

typedef short s16x2h __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ int dot2( uint32_t a, uint32_t b, int c ) { return __builtin_amdgcn_sdot2( __builtin_bit_cast( s16x2h, a ), __builtin_bit_cast( s16x2h, b ), c, false ); }
// One bundle of (stage, band) units.  K0 .. K1: the window taps the bundle's filter set can use (wave-uniform: 4-tap search 2..5, 6 taps / alternative half-pel 1..6, 8 taps 0..7).
// A band = <= 16 rows of a block (one row of 16x16_fast tiles); the bands of a block are independent units (their partial costs are added with integer atomics, which
// commute: results do not depend on the schedule).  Per unit two cooperative phases:
//   H   first pass of the <= 3 distinct horizontal positions, rows band + K0 - 4 .. band + BH + K1 - 4, straight from the plane into LDS: 8 outputs per lane from two
//       overlapping 16-byte loads, tap PAIRS as v_dot2_i32_i16 on the even / odd sample pairs of the window (no unpacking);
//   VD  eight lanes per (position, tile), lane r = tile row r: second pass of the lane's own prediction row(s) out of LDS (v_mad_i32_i16 on scaled taps: predRow), clip, difference to
//       the original row(s), horizontal butterflies in registers, vertical ones across the eight lanes with DPP (the factorisation of dist.hip's hadKernel), |DC| >> 2,
//       per-tile normalisation — the prediction never leaves registers.
// the second pass's taps of one vertical phase: prepared once per (lane, slot) and shared by the slot's prediction rows
template<int NT> struct VTaps { int c[NT]; };
template<int K0, int K1>
__device__ __forceinline__ void loadVTaps( const int* tapL, int fyk, VTaps<K1 - K0 + 1>& vt )
{
#pragma unroll
  for( int t = 0; t < K1 - K0 + 1; t++ ) vt.c[t] = tapL[fyk * 8 + K0 + t];
}
// acc + ( low / high 16-bit half of v ) * ( low half of c ): v_mad_i32_i16, op_sel picks the half — one full-rate instruction per product and no ( c, 0 ) / ( 0, c ) operand
// pairs to prepare (profiles/r04_valu_rate.log: same issue rate as v_dot2_i32_i16, semantics checked on the GPU)
__device__ __forceinline__ int madLo( uint32_t v, int c, int acc ) { int r; asm( "v_mad_i32_i16 %0, %1, %2, %3" : "=v"( r ) : "v"( v ), "v"( c ), "v"( acc ) ); return r; }
__device__ __forceinline__ int madHi( uint32_t v, int c, int acc ) { int r; asm( "v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"( r ) : "v"( v ), "v"( c ), "v"( acc ) ); return r; }
__device__ __forceinline__ int mulLo( uint32_t v, int c ) { int r; asm( "v_mad_i32_i16 %0, %1, %2, 0" : "=v"( r ) : "v"( v ), "v"( c ) ); return r; }
__device__ __forceinline__ int med3( int v, int hi ) { int r; asm( "v_med3_i32 %0, %1, 0, %2" : "=v"( r ) : "v"( v ), "v"( hi ) ); return r; }      // clip to [0, hi] (hi is not a literal: the compiler emits compare + select + min)
__device__ __forceinline__ int mulHi( uint32_t v, int c ) { int r; asm( "v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"( r ) : "v"( v ), "v"( c ) ); return r; }
// 8 clipped prediction samples of band row y (columns x0 .. x0 + 7) of a position, as four sample pairs.
// Where the reference adds its constants is moved, the values are the same:
//  * the first pass leaves its 14-bit intermediates in LDS as value + 2^( headRoom - 1 ) WITHOUT the reference's -8192 offset (InterpolationFilter.cpp:401-408): every tap set
//    sums to 64, so the second pass's rounding term ( 1 << ( shift2 - 1 ) ) + ( 8192 << 6 ) (:394-400) is exactly what the stored offset contributes to the tap sum — the
//    accumulators start at zero, and the zero-phase copy (:309-322) is a plain shift;
//  * the plan's vertical taps are scaled by 2^( 16 - shift2 ): the sum >> shift2 is the upper half of the accumulator — clip to [0, max] on the scaled value (v_med3_i32
//    against ( max << 16 ) | 0xffff), then ONE v_perm_b32 shifts and packs two samples ( 3 instructions per sample pair instead of 5 ).
// anyFrac (WAVE-UNIFORM): some lane of the wave has a fractional vertical phase -> every lane filters (a zero-phase lane with the taps of phase 0, ( 0, 64, 0, 0 ): the same
// value as the copy; its rows beyond the staged ones meet zero taps); no lane has one -> the copy on packed pairs.  No divergent branch per row.
template<int K0, int K1>
__device__ __forceinline__ void predRow( const int16_t* tv /* first-pass band of the position's horizontal variant, at column x0 */, int w, int y, int syk, bool anyFrac,
                                         const VTaps<K1 - K0 + 1>& vt, int headRoom, uint32_t maxPk, uint32_t ( &o )[4] )
{
  constexpr int NT = K1 - K0 + 1;
  if( anyFrac )
  {
    int acc[8];
#pragma unroll
    for( int t = 0; t < NT; t++ )
    {
      const int c = vt.c[t];
      const u32x4 r = *reinterpret_cast<const u32x4*>( tv + __mul24( y + syk + t + 1, w ) );
      if( t == 0 )
      {
        acc[0] = mulLo( r.x, c ); acc[1] = mulHi( r.x, c ); acc[2] = mulLo( r.y, c ); acc[3] = mulHi( r.y, c );
        acc[4] = mulLo( r.z, c ); acc[5] = mulHi( r.z, c ); acc[6] = mulLo( r.w, c ); acc[7] = mulHi( r.w, c );
      }
      else
      {
        acc[0] = madLo( r.x, c, acc[0] ); acc[1] = madHi( r.x, c, acc[1] ); acc[2] = madLo( r.y, c, acc[2] ); acc[3] = madHi( r.y, c, acc[3] );
        acc[4] = madLo( r.z, c, acc[4] ); acc[5] = madHi( r.z, c, acc[5] ); acc[6] = madLo( r.w, c, acc[6] ); acc[7] = madHi( r.w, c, acc[7] );
      }
    }
    const int maxS = ( int ) ( ( ( maxPk & 0xffffu ) << 16 ) | 0xffffu );
#pragma unroll
    for( int i = 0; i < 8; i++ ) acc[i] = med3( acc[i], maxS );
#pragma unroll
    for( int i = 0; i < 4; i++ ) o[i] = __builtin_amdgcn_perm( ( uint32_t ) acc[2 * i + 1], ( uint32_t ) acc[2 * i], 0x07060302u );
  }
  else
  {
    const u32x4 r = *reinterpret_cast<const u32x4*>( tv + __mul24( y + syk + 4 - K0, w ) );
    const uint32_t rw[4] = { r.x, r.y, r.z, r.w };
    const s16x2 sh = { ( short ) headRoom, ( short ) headRoom }, z = { 0, 0 };
#pragma unroll
    for( int i = 0; i < 4; i++ )
      o[i] = __builtin_bit_cast( uint32_t, __builtin_elementwise_min( __builtin_elementwise_max( __builtin_bit_cast( s16x2, rw[i] ) >> sh, z ), __builtin_bit_cast( s16x2, maxPk ) ) );
  }
}

// 4 rounded 2x2 averages ( a + b rows, 8 columns ) as ints
// (operands are samples or bi-prediction patterns 2 org - pred, |value| < 2^13: the packed vertical sum stays inside 16 bits; the horizontal sum + rounding is a signed dot product with (1, 1))
__device__ __forceinline__ void avgInts( const uint32_t ( &ra )[4], const uint32_t ( &rb )[4], int ( &o )[4] )
{
#pragma unroll
  for( int i = 0; i < 4; i++ ) o[i] = dot2( pkAdd( ra[i], rb[i] ), 0x00010001u, 2 ) >> 2;
}

// ---- Hadamard tiles --------------------------------------------------------------------------------------------------------------------------
// The reference's tile ladder (xGetHADs, RdCost.cpp:1818-1938: first match wins).  Every tile type is scored by a TEAM of LT lanes that hold 8 differences each
// (the tile's 8 x LT = 16, 32, 64 or 128 values): an 8-point Walsh-Hadamard transform in the lane's registers, then log2( LT ) butterfly stages across the lanes with DPP —
//   8x8   lane r = row r                                   16x16_fast  lane r = rows 2r, 2r + 1, 2x2 averages of 16 columns (RdCost.cpp:1126-1223)
//   16x8  lane r = row r & 7, columns 8 ( r >> 3 ) ..      8x16        lane r = row r                      (128 values: 16 lanes)
//   8x4   lane r = row r                                   4x8         lane r = rows 2r, 2r + 1 of 4 columns (the register transform then covers x0, x1, y0)
//   4x4   lane r = rows 2r, 2r + 1
// The pairings (in registers: the three index bits a lane holds; across lanes: i <-> 15 - i, i <-> 7 - i, i ^ 2, i ^ 1) are linearly independent over GF(2), so each is a
// valid Hadamard factorisation: the multiset of |coefficients| is the reference's, and the DC coefficient — the only one treated specially (|DC| >> 2 in every tile type) —
// ends in register 0 of lane 0.  Normalisation per tile type as the reference: RdCost.cpp:1119-1121 (4x4), :1317-1319 (8x8), :1218-1222 (16x16_fast),
// :1467 / :1606 (16x8 / 8x16: ( int ) ( sad / sqrt( 16.0 * 8 ) * 2 ) in IEEE double), :1682 / :1763 (8x4 / 4x8: sqrt( 4.0 * 8 )).
enum { TK_8x8 = 0, TK_16F = 1, TK_16x8 = 2, TK_8x16 = 3, TK_8x4 = 4, TK_4x8 = 5, TK_4x4 = 6, TK_2x2 = 7, TK_ROWS = 8 /* no transform: rows of 8 samples (SAD-scored stages) */ };
__host__ __device__ __forceinline__ int hadTileKind( int w, int h, bool fast )
{
  if( w > h && !( h & 7 ) && !( w & 15 ) ) return TK_16x8;
  if( w < h && !( w & 7 ) && !( h & 15 ) ) return TK_8x16;
  if( w > h && !( h & 3 ) && !( w & 7 ) )  return TK_8x4;
  if( w < h && !( w & 3 ) && !( h & 7 ) )  return TK_4x8;
  if( fast && w == h && !( w & 31 ) )      return TK_16F;
  if( !( w & 7 ) && !( h & 7 ) )           return TK_8x8;
  if( !( w & 3 ) && !( h & 3 ) )           return TK_4x4;
  return TK_2x2;
}
__host__ __device__ __forceinline__ int tileW( int kind ) { return ( kind == TK_16F || kind == TK_16x8 ) ? 16 : ( ( kind == TK_4x8 || kind == TK_4x4 ) ? 4 : ( kind == TK_2x2 ? 2 : 8 ) ); }
__host__ __device__ __forceinline__ int tileH( int kind ) { return ( kind == TK_16F || kind == TK_8x16 ) ? 16 : ( ( kind == TK_8x4 || kind == TK_4x4 ) ? 4 : ( kind == TK_2x2 ? 2 : 8 ) ); }
__host__ __device__ __forceinline__ int tileLanes( int kind ) { return ( kind == TK_16x8 || kind == TK_8x16 ) ? 16 : ( ( kind == TK_8x4 || kind == TK_4x8 ) ? 4 : ( kind == TK_4x4 ? 2 : ( kind == TK_2x2 ? 1 : 8 ) ) ); }

__device__ __forceinline__ uint32_t hadNorm( uint32_t s, int kind )
{
  if( kind == TK_8x8 || kind == TK_16F ) { const uint32_t v = ( s + 2 ) >> 2; return kind == TK_16F ? v << 2 : v; }
  if( kind == TK_4x4 ) return ( s + 1 ) >> 1;
  if( kind == TK_16x8 || kind == TK_8x16 ) return ( uint32_t ) ( int ) ( ( double ) ( int ) s / __builtin_sqrt( 16.0 * 8 ) * 2 );
  if( kind == TK_8x4 || kind == TK_4x8 )   return ( uint32_t ) ( int ) ( ( double ) ( int ) s / __builtin_sqrt( 4.0 * 8 ) * 2 );
  return s;
}

// the team's transform: d = the lane's 8 differences, r = the lane's index inside its team of LT lanes (teams are aligned groups of consecutive lanes).
// Returns the tile's normalised SATD in every lane of the team.  |d| < 2^23 / 128 on entry (differences of <= 12-bit values).
// hadTeamCross: the stages across the lanes + sum, on values the lane has already transformed in its registers
__device__ __forceinline__ uint32_t sadU32( uint32_t a, uint32_t b, uint32_t acc ) { uint32_t r; asm( "v_sad_u32 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( b ), "v"( acc ) ); return r; }
// `done`: the lane-pair distances ( 8 / 4 / 2 ) whose stage the caller has already run (on packed pairs, hadTeamPkD)
__device__ __forceinline__ uint32_t hadTeamCross( int ( &d )[8], int r, int LT, int kind, int lane, int done = 0 )
{
  // upper lane of a pair: other - own, lower: own + other; |d| < 2^23.  The LAST stage (lane pairs i ^ 1: every team has it) adds 2^31 with its multiply (v_mad_i32_i24):
  // |coefficient| = |biased - 2^31| as unsigned numbers, so the sum of magnitudes is one v_sad_u32 per coefficient instead of subtract + max + add
  constexpr uint32_t B31 = 0x80000000u;
#define ME_VSTAGE( CTRL, BIT ) { const int sgn = ( r & ( BIT ) ) ? -1 : 1; _Pragma( "unroll" ) \
  for( int i = 0; i < 8; i++ ) { const int t = __mul24( d[i], sgn ); d[i] = VVHIP_DPP( d[i], CTRL ) + t; } }
  if( LT >= 16 && !( done & 8 ) ) ME_VSTAGE( VVHIP_DPP_MIRROR, 8 )
  if( LT >= 8 && !( done & 4 ) )  ME_VSTAGE( VVHIP_DPP_HALF_MIRROR, 4 )
  if( LT >= 4 && !( done & 2 ) )  ME_VSTAGE( VVHIP_DPP_XOR2, 2 )
#undef ME_VSTAGE
  {                                                                                   // (LT >= 2 for every tile type that comes here)
    const int sgn = ( r & 1 ) ? -1 : 1;
#pragma unroll
    for( int i = 0; i < 8; i++ )
    {
      int t; asm( "v_mad_i32_i24 %0, %1, %2, %3" : "=v"( t ) : "v"( d[i] ), "v"( sgn ), "s"( B31 ) );      // (written out: the compiler turns a multiply by +-1 plus a constant into four instructions)
      d[i] = ( int ) ( ( uint32_t ) VVHIP_DPP( d[i], VVHIP_DPP_XOR1 ) + ( uint32_t ) t );
    }
  }
  uint32_t s = 0;
#pragma unroll
  for( int i = 0; i < 8; i++ ) s = sadU32( ( uint32_t ) d[i], B31, s );
  if( r == 0 ) { const uint32_t dc = sadU32( ( uint32_t ) d[0], B31, 0u ); s = s - dc + ( dc >> 2 ); }
  s = vvhipGroupSum32( s, LT, lane );
  return hadNorm( s, kind );
}
__device__ __forceinline__ uint32_t hadTeam( int ( &d )[8], int r, int LT, int kind, int lane )
{
#pragma unroll
  for( int len = 1; len < 8; len <<= 1 )
#pragma unroll
    for( int i = 0; i < 8; i += 2 * len )
#pragma unroll
      for( int q = i; q < i + len; q++ ) { const int x = d[q], z = d[q + len]; d[q] = x + z; d[q + len] = x - z; }
  return hadTeamCross( d, r, LT, kind, lane );
}
// the same from the lane's two operand rows as PACKED sample pairs (o - p per 16-bit half): the two register stages that pair whole dwords run on packed pairs
// (v_pk_add_i16 / v_pk_sub_i16: sums of four differences, |d| <= 2^( bitDepth + 1 ) with bitDepth <= 10 -> 14 bits), the stage inside a dword is the unpacking itself
// (lo + hi, lo - hi as dot products with ( 1, 1 ) / ( 1, -1 )): 20 instructions for difference + register transform instead of 32.  The stages of a Hadamard transform commute;
// the all-plus coefficient (DC) still ends in register 0.
// Round 6: the first TWO stages across the lanes run on the packed pairs too — v_mov_b32_dpp + v_pk_mad_i16 ( own * ( +-1, +-1 ) + other ) per dword, 8 instructions per
// stage instead of 16 ( v_mul_i32_i24 + v_add_u32_dpp per value ).  Range: a difference is |o - p| <= 2 ( 2^bitDepth - 1 ) ( o a bi-prediction pattern 2 org - pred, p a
// sample; bitDepth <= 10 is checked at plan creation ) = 2046; two register stages + two lane stages sum 16 of them: 32736 <= 32767 — 16 bits hold every intermediate
// ( tests/test_gpu_corners.py drives exactly these extremes ).
__device__ __forceinline__ uint32_t pkMad( uint32_t a, uint32_t b, uint32_t c ) { uint32_t r; asm( "v_pk_mad_i16 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( b ), "v"( c ) ); return r; }
__device__ __forceinline__ uint32_t hadTeamPkD( const uint32_t ( &D )[4], int r, int LT, int kind, int lane )
{
  const uint32_t E0 = pkAdd( D[0], D[1] ), E1 = pkSub( D[0], D[1] ), E2 = pkAdd( D[2], D[3] ), E3 = pkSub( D[2], D[3] );
  uint32_t F[4] = { pkAdd( E0, E2 ), pkAdd( E1, E3 ), pkSub( E0, E2 ), pkSub( E1, E3 ) };
#define ME_VSTAGE_PK( CTRL, BIT ) { const uint32_t sg = ( r & ( BIT ) ) ? 0xffffffffu : 0x00010001u; _Pragma( "unroll" ) \
  for( int q = 0; q < 4; q++ ) F[q] = pkMad( F[q], sg, ( uint32_t ) VVHIP_DPP( F[q], CTRL ) ); }
  int done = 0;
  if( LT >= 16 )     { ME_VSTAGE_PK( VVHIP_DPP_MIRROR, 8 ) ME_VSTAGE_PK( VVHIP_DPP_HALF_MIRROR, 4 ) done = 8 | 4; }
  else if( LT >= 8 ) { ME_VSTAGE_PK( VVHIP_DPP_HALF_MIRROR, 4 ) ME_VSTAGE_PK( VVHIP_DPP_XOR2, 2 ) done = 4 | 2; }
  else if( LT >= 4 ) { ME_VSTAGE_PK( VVHIP_DPP_XOR2, 2 ) done = 2; }
#undef ME_VSTAGE_PK
  int d[8];
#pragma unroll
  for( int q = 0; q < 4; q++ ) { d[2 * q] = dot2( F[q], 0x00010001u, 0 ); d[2 * q + 1] = dot2( F[q], 0xffff0001u, 0 ); }
  return hadTeamCross( d, r, LT, kind, lane, done );
}
__device__ __forceinline__ uint32_t hadTeamPk( const uint32_t ( &o )[4], const uint32_t ( &p )[4], int r, int LT, int kind, int lane )
{
  const uint32_t D[4] = { pkSub( o[0], p[0] ), pkSub( o[1], p[1] ), pkSub( o[2], p[2] ), pkSub( o[3], p[3] ) };
  return hadTeamPkD( D, r, LT, kind, lane );
}
// two ints ( each inside 16 bits ) as a packed pair
__device__ __forceinline__ uint32_t pkInts( int lo, int hi ) { return __builtin_amdgcn_perm( ( uint32_t ) hi, ( uint32_t ) lo, 0x05040100u ); }

// dot product on top of a wave-uniform constant (a scalar register as the third operand: no accumulator initialisation per output)
__device__ __forceinline__ int dot2s( uint32_t a, uint32_t b, int c ) { int r; asm( "v_dot2_i32_i16 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( b ), "s"( c ) ); return r; }

// First (horizontal) pass of a stage unit: tmp[vl][r][x] <-> plane row y0 + K0 - 4 + r, column x + ( displacement of variant v0 + vl >> 4 ), for nH = variants x rowsT x G
// units of 8 outputs.  HU units per lane and trip, all their loads issued before the first is used (a unit is short: without it every trip of the wave waits out a full
// memory latency).  8 outputs per lane from two overlapping 16-byte loads, tap PAIRS as v_dot2_i32_i16 on the even / odd sample pairs of the window (no unpacking).
// The plan's tap pairs are scaled by 2^( 8 - shift1 ): ( sum + offset ) >> shift1 sits in bytes 1..2 of the accumulator, ONE v_perm_b32 shifts and packs two outputs.
// ONE: the pass holds a single horizontal variant (units of 32 / 64 columns): its displacement, phase and taps are wave-uniform.
// refB: the unit's plane row y0 + K0 - 4, 8 samples left of the unit's column 0 (every sample offset below is >= 0).
template<int K0, int K1, bool ONE>
__device__ __forceinline__ void firstPass( const char* refB, int rs, int nH, int G, int log2G, int rowsT, int ldsPitch, int v0, int hx0, int hx1, int hx2,
                                           const uint32_t* tapP, int16_t* tmp, int tid, int offS, int headRoom, uint32_t biasPk )
{
  constexpr int NT = K1 - K0 + 1, NP = NT / 2, B0 = ( NT - 2 ) / 2, HU = VVHIP_ME_HU;
  const int txOne = v0 == 0 ? hx0 : ( v0 == 1 ? hx1 : hx2 );
  for( int ub = tid; ub < nH; ub += HU * 64 )
  {
    u32x4 LA[HU], LB[HU]; int fxs[HU], at[HU]; bool ok[HU];
#pragma unroll
    for( int q = 0; q < HU; q++ )
    {
      const int u = ub + q * 64;
      ok[q] = u < nH;
      const int uu = ok[q] ? u : ub;
      const int x0 = ( uu & ( G - 1 ) ) << 3, rr = uu >> log2G;
      int r = rr, txv = txOne;
      if( !ONE )
      {
        const int vl = ( rr >= rowsT ) + ( rr >= 2 * rowsT ), v = v0 + vl;                          // (<= 3 variants: no division)
        r = rr - __mul24( vl, rowsT ); txv = v == 0 ? hx0 : ( v == 1 ? hx1 : hx2 );
      }
      fxs[q] = txv & 15; at[q] = __mul24( rr, ldsPitch ) + x0;
      const uint32_t off = ( uint32_t ) ( __mul24( r, rs ) + x0 + ( txv >> 4 ) + 8 ) * 2u;        // bytes from refB to the output's integer position p
      // window samples s[0 .. 6 + NT] = p[K0 - 3 ..]: A = s[0..7] as even pairs W[0..3], B = s[NT - 1 .. NT + 6] as odd pairs S[B0 .. B0 + 3]; zero phase: A = p[0..7]
      // (a zero-phase unit is a copy of A: its second request would be read for nothing — a third of the variants of a half-sample stage, round 5)
      LA[q] = ld16o( refB, fxs[q] ? off + ( uint32_t ) ( 2 * ( K0 - 3 ) ) : off );
      if( fxs[q] ) LB[q] = ld16o( refB, off + ( uint32_t ) ( 2 * ( K0 - 3 + NT - 1 ) ) );
    }
#pragma unroll
    for( int q = 0; q < HU; q++ )
    {
      if( !ok[q] ) continue;
      const int fxv = fxs[q];
      const u32x4 A = LA[q];
      u32x4 ov;
      if( fxv )
      {
        const u32x4 B = LB[q];                                  // (requested for non-zero phases only: never read otherwise)
        uint32_t W[4 + NP], S[4 + NP];
        W[0] = A.x; W[1] = A.y; W[2] = A.z; W[3] = A.w;
        S[B0] = B.x; S[B0 + 1] = B.y; S[B0 + 2] = B.z; S[B0 + 3] = B.w;
#pragma unroll
        for( int m = B0 - 1; m >= 0; m-- ) S[m] = __builtin_amdgcn_alignbit( W[m + 1], W[m], 16 );      // the missing pairs by v_alignbit
#pragma unroll
        for( int m = 4; m < 4 + NP - 1; m++ ) W[m] = __builtin_amdgcn_alignbit( S[m], S[m - 1], 16 );
        uint32_t cp[NP];
#pragma unroll
        for( int i = 0; i < NP; i++ ) cp[i] = tapP[fxv * 4 + i];
        uint32_t o[4];
#pragma unroll
        for( int qq = 0; qq < 4; qq++ )
        {
          int e = dot2s( W[qq], cp[0], offS ), d = dot2s( S[qq], cp[0], offS );
#pragma unroll
          for( int i = 1; i < NP; i++ ) { e = dot2( W[qq + i], cp[i], e ); d = dot2( S[qq + i], cp[i], d ); }
          o[qq] = __builtin_amdgcn_perm( ( uint32_t ) d, ( uint32_t ) e, 0x06050201u );
        }
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
      }
      else
      {
        // filterCopy<true,false>: ( sample << headRoom ) - 8192 (InterpolationFilter.cpp:285-296), stored with the folded constants like the filtered rows
        const uint32_t aw[4] = { A.x, A.y, A.z, A.w };
        const s16x2 sh = { ( short ) headRoom, ( short ) headRoom };
        uint32_t o[4];
#pragma unroll
        for( int qq = 0; qq < 4; qq++ ) o[qq] = pkAdd( __builtin_bit_cast( uint32_t, __builtin_bit_cast( s16x2, aw[qq] ) << sh ), biasPk );
        ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
      }
      *reinterpret_cast<u32x4*>( tmp + at[q] ) = ov;
    }
  }
}

// a stage unit in the schedule: stage index | band of 32 rows << 24 | 64-column half << 27 | continues the previous unit's sums << 28 | the next unit continues << 29 | atomic << 30
// bits 0..21 the stage index, bits 22..23 the horizontal variant the unit is restricted to + 1 (0: all variants of the stage — narrow units; wide units are split, see plan creation)
constexpr int ST_STAGE_MASK = 0x3fffff, ST_VAR_SHIFT = 22;
constexpr int ST_UNIT_CONT = 1 << 28, ST_UNIT_MORE = 1 << 29, ST_UNIT_ATOMIC = 1 << 30;      // ATOMIC: a stage of more than two units — every unit a wave of its own, sums added to the (cleared) cost array

// GEN = false: the shapes of the fast presets (CTU 64, quad-tree only) — square 8..64, tiles 8x8 / 16x16_fast / SAD rows, eight lanes per tile: every tile quantity is a
// compile-time constant.  GEN = true: any shape (CTU 128 + multi-type tree); such units run in their own launch (their own registers, their own LDS size).
template<int K0, int K1, bool GEN>
__device__ __forceinline__ void stageBody( const MePlanes& P, const MeArgs& a, const WaveSpan span, int16_t* lds, uint32_t* pairCost, const int wv )
{
  constexpr int NT = K1 - K0 + 1;
  // a unit belongs to ONE wave: the workgroup's other waves work on other bundles (their own LDS slice); every hand-over through LDS is inside the wave
#define ST_SYNC() { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
  const int tid = threadIdx.x & 63, nthr = 64, lane = tid, bd = a.bitDepth;
  const int headRoom = 14 - bd > 2 ? 14 - bd : 2;
  // first (not last) pass, InterpolationFilter.cpp:401-408: ( sum >> shift1 ) - 8192; kept in LDS as ( sum >> shift1 ) + 2^( headRoom - 1 ) instead — the second pass's
  // constants ( :394-400 ) folded into the stored value, see predRow
  const int offS = ( 1 << ( headRoom - 1 ) ) << 8;                                   // the stored offset on the scaled accumulator (firstPass)
  const uint32_t biasPk = ( uint32_t ) ( 1 << ( headRoom - 1 ) ) * 0x00010001u;
  const uint32_t maxPk = ( uint32_t ) ( ( 1 << bd ) - 1 ) * 0x00010001u;
  int* tapL = reinterpret_cast<int*>( lds );                                         // [16 phases][8] taps of the current unit's stage
  uint32_t* tapP = reinterpret_cast<uint32_t*>( lds ) + 128;                         // [16 phases][4] tap pairs (K0 + 2i, K0 + 2i + 1)
  int* posL = reinterpret_cast<int*>( lds ) + 128 + 64;                              // [9] evaluated positions: k | ( tx + 64 ) << 8 | ( ty + 64 ) << 20
  uint32_t* costL = reinterpret_cast<uint32_t*>( lds ) + 128 + 64 + 16;              // [9] sums of the current stage
  int16_t* tmp = lds + 2 * ( 128 + 64 + 16 + 16 );

  int curTab = -1;
  if( K0 == 2 )                                                                      // the 4-tap search set has ONE table (filter_mode 2, no alternative half-sample filter):
  {                                                                                  // requested at once, next to the first unit's record instead of behind it
    for( int i = tid; i < 192; i += nthr ) reinterpret_cast<int*>( lds )[i] = a.tapTables[4 * 192 + i];
    curTab = 4;
  }
  for( int si = 0; si < span.count; si++ )
  {
    const StageUnit* up = a.stageUnits + span.first + si;                        // (wave-uniform address: scalar loads; the table is in schedule order, one record per unit)
    const vvhip_me_stage_job j = up->j;
    const int unit = up->order, stage = unit & ST_STAGE_MASK, varOnly = ( unit >> ST_VAR_SHIFT ) & 3;      // varOnly: 0 = every variant, v + 1 = variant v alone
    // the unit: <= 32 rows x <= 64 columns of the block (a band of one 64-column half).  Blocks of more than one unit (h > 32 or w > 64) are shared by the two waves of the
    // workgroup; a wave adds the sums of its units (ST_UNIT_CONT / _MORE) before the two waves meet
    const int w = j.width, h = j.height, uw = w < 64 ? w : 64, uwH = uw < 8 ? 8 : uw, G = uwH >> 3, log2G = 31 - __builtin_clz( G );
    const int BH = h < 32 ? h : 32, rowsT = BH + NT, y0 = ( ( unit >> 24 ) & 7 ) * BH, xoff = ( ( unit >> 27 ) & 1 ) * 64;
    const int ldsPitch = uwH + 8;                                                  // LDS row pitch: an odd number of 16-byte chunks (rows of a tile column land in different banks)
    const int rs = P.stride[j.ref_plane];
    const char* refB = reinterpret_cast<const char*>( P.p[j.ref_plane] + j.ref_off + xoff + ( ptrdiff_t ) ( y0 + K0 - 4 ) * rs - 8 );      // (wave-uniform)
    const char* orgB = reinterpret_cast<const char*>( P.p[j.org_plane] + j.org_off + xoff );
    const int os = P.stride[j.org_plane] ? P.stride[j.org_plane] : w;             // (stride 0: a compact pool block — bi-prediction patterns)
    // the evaluated positions and their distinct horizontal displacements (<= 3: the refinement offsets are -1, 0, 1; one first pass each, shared like the reference's planes)
    // come precomputed with the unit; posL: the positions grouped by displacement (a pass of the unit works on the positions of the variants it holds in LDS)
    const int hx0 = up->hx[0], hx1 = up->hx[1], hx2 = up->hx[2], nHor = up->nHor, nPos = up->nPos, cnt0 = up->cnt0, cnt1 = up->cnt1;
    const int myPos = tid < 9 ? up->pos[tid] : 0;                                // (requested before the wait below)
    ST_SYNC();                                                                    // the previous unit's readers are done with the tables and tmp
    if( tid < 9 ) posL[tid] = myPos;
    // the tap tables of the unit's (tap set, alternative half-sample filter) from the plan, 192 dwords: fetched when they differ from the previous unit's (a bundle of the
    // 4-tap search set never changes them)
    const int tabId = j.filter_mode * 2 + ( j.alt_hpel ? 1 : 0 );
    if( tabId != curTab )
    {
      for( int i = tid; i < 192; i += nthr ) reinterpret_cast<int*>( lds )[i] = a.tapTables[tabId * 192 + i];
      curTab = tabId;
    }
    if( !( unit & ST_UNIT_CONT ) && tid < 9 ) costL[tid] = 0;
    // the tile type follows from the BLOCK's shape (the reference's ladder), the unit holds whole tiles of it
    // (SAD-scored stages: no transform — the lanes are dealt like an 8x8 / 8x4 tile's, or like the 4x8 tile's for a 4-wide block)
    const int kind = j.func == VVHIP_DF_SAD ? TK_ROWS : ( GEN ? hadTileKind( w, h, j.func == VVHIP_DF_HAD_FAST ) : ( ( j.func == VVHIP_DF_HAD_FAST && ( w & 31 ) == 0 ) ? TK_16F : TK_8x8 ) );
    const bool rows4 = GEN && ( kind == TK_4x8 || ( kind == TK_ROWS && uw == 4 ) );      // two rows of four samples per lane
    const int log2PW = !GEN ? ( kind == TK_16F ? 4 : 3 ) : ( rows4 ? 2 : ( kind == TK_ROWS ? 3 : 31 - __builtin_clz( tileW( kind ) ) ) );
    const int log2PH = !GEN ? ( kind == TK_16F ? 4 : 3 ) : ( rows4 ? 3 : ( kind == TK_ROWS ? ( BH < 8 ? 31 - __builtin_clz( BH ) : 3 ) : 31 - __builtin_clz( tileH( kind ) ) ) );
    const int log2LT = !GEN ? 3 : ( rows4 ? 2 : ( kind == TK_ROWS ? log2PH : 31 - __builtin_clz( tileLanes( kind ) ) ) ), LT = 1 << log2LT;
    // (everything is a power of two: tile counts by shifts — no integer division per unit)
    const int log2TX = ( 31 - __builtin_clz( uw ) ) - log2PW, log2TB = log2TX + ( 31 - __builtin_clz( BH ) ) - log2PH, tilesX = 1 << log2TX;
    // Units of 32 and 64 columns keep ONE horizontal variant in LDS at a time (a pass = first pass of the variant, then every position that uses it: with 32-row bands a
    // position of a 64-wide unit is exactly 64 lanes of second-pass work), narrower ones all (<= 3) of them: 6 KB of LDS per wave instead of 9.5 — the kernel is
    // occupancy-bound — and half as many units for the 64x64 blocks.
    const int vpp = uw <= 16 ? nHor : 1;
    for( int v0 = varOnly ? varOnly - 1 : 0; v0 < ( varOnly ? varOnly : nHor ); v0 += vpp )
    {
    const int nV = nHor - v0 < vpp ? nHor - v0 : vpp;
    const int pBeg = v0 == 0 ? 0 : ( v0 == 1 ? cnt0 : cnt0 + cnt1 ), pEnd = v0 + nV >= 3 ? nPos : ( v0 + nV == 2 ? cnt0 + cnt1 : ( v0 + nV == 1 ? cnt0 : 0 ) );
    ST_SYNC();                                                                    // tables and positions are written; the previous pass's readers are done with tmp
    // ---- H: tmp[v - v0][r][x] <-> plane row y0 + K0 - 4 + r, column x + sx[v] (firstPass)
#if defined( VVHIP_ME_CUT ) && ( VVHIP_ME_CUT & 1 )      // (cut build for phase timing, tools/me_cut.sh: no first pass — results are wrong by construction)
    const int nH = 0;
#else
    const int nH = nV * rowsT * G;
#endif
    if( nV == 1 ) firstPass<K0, K1, true>( refB, rs, nH, G, log2G, rowsT, ldsPitch, v0, hx0, hx1, hx2, tapP, tmp, tid, offS, headRoom, biasPk );
    else          firstPass<K0, K1, false>( refB, rs, nH, G, log2G, rowsT, ldsPitch, v0, hx0, hx1, hx2, tapP, tmp, tid, offS, headRoom, biasPk );
    ST_SYNC();     
    // ---- VD: LT lanes per (position, tile); what a lane holds: see the tile table above.  A lane keeps its part of a tile for EVERY position of the pass (the positions that
    //      do not fit side by side into the wave follow in an inner loop): the original rows — for 16x16_fast tiles their 2x2 averages — are fetched and prepared once per pass
    //      instead of once per position (<= 9 times: the original block was two thirds of this kernel's L1 accesses)
#if defined( VVHIP_ME_CUT ) && ( VVHIP_ME_CUT & 2 )      // (cut build: no second pass / distortion)
    const int nPP = 0;
#else
    const int nPP = pEnd - pBeg;                                                   // positions of this pass
#endif
    const int log2L = log2TB + log2LT, log2Lw = log2L < 6 ? log2L : 6;            // lanes per position; of them inside one wave trip
    const int trips = log2L > 6 ? 1 << ( log2L - 6 ) : 1, PL = 64 >> log2Lw, plLane = tid >> log2Lw;      // wave trips per position; positions side by side in a trip, the lane's
    for( int c = 0; c < trips && nPP > 0; c++ )
    {
      const int sl = ( c << 6 ) + ( tid & ( ( 1 << log2Lw ) - 1 ) ), r = sl & ( LT - 1 ), t = sl >> log2LT;
      const int tyi = t >> log2TX, txi = t & ( tilesX - 1 );
      // the lane's original rows (16x16_fast: their 2x2 averages, RdCost.cpp:1138-1160)
      uint32_t o4[4] = { 0, 0, 0, 0 };                                              // (16x16_fast: the eight averages as packed pairs)
      if( kind == TK_16F )
      {
        const uint32_t po = ( uint32_t ) ( __mul24( y0 + tyi * 16 + 2 * r, os ) + txi * 16 ) * 2u, po1 = po + ( uint32_t ) os * 2u;
        const u32x4 c0v = ld16o( orgB, po ), c1v = ld16o( orgB, po + 16 ), e0 = ld16o( orgB, po1 ), e1 = ld16o( orgB, po1 + 16 );
        int ao[4];
        { const uint32_t oa[4] = { c0v.x, c0v.y, c0v.z, c0v.w }, ob[4] = { e0.x, e0.y, e0.z, e0.w }; avgInts( oa, ob, ao ); o4[0] = pkInts( ao[0], ao[1] ); o4[1] = pkInts( ao[2], ao[3] ); }
        { const uint32_t oa[4] = { c1v.x, c1v.y, c1v.z, c1v.w }, ob[4] = { e1.x, e1.y, e1.z, e1.w }; avgInts( oa, ob, ao ); o4[2] = pkInts( ao[0], ao[1] ); o4[3] = pkInts( ao[2], ao[3] ); }
      }
      else if( rows4 )
      {
        const uint32_t po = ( uint32_t ) __mul24( y0 + tyi * 8 + 2 * r, os ) * 2u;
        const u32x2 oa = ld8o( orgB, po ), ob = ld8o( orgB, po + ( uint32_t ) os * 2u );
        o4[0] = oa.x; o4[1] = oa.y; o4[2] = ob.x; o4[3] = ob.y;
      }
      else
      {
        const int row = ( GEN && kind == TK_16x8 ) ? tyi * 8 + ( r & 7 ) : ( tyi << log2PH ) + r, col = ( GEN && kind == TK_16x8 ) ? txi * 16 + 8 * ( r >> 3 ) : txi * 8;
        const u32x4 ovv = ld16o( orgB, ( uint32_t ) ( __mul24( y0 + row, os ) + col ) * 2u );
        o4[0] = ovv.x; o4[1] = ovv.y; o4[2] = ovv.z; o4[3] = ovv.w;
      }
      for( int p0 = 0; p0 < nPP; p0 += PL )
      {
      const int pl = p0 + plLane;
      const bool valid = pl < nPP;
      const int pi = pBeg + ( valid ? pl : 0 );
      const int pk = posL[pi], txk = ( ( pk >> 8 ) & 0xfff ) - 64, tyk = ( ( pk >> 20 ) & 0xfff ) - 64;
      const int hv = ( txk == hx0 ? 0 : ( ( nHor > 1 && txk == hx1 ) ? 1 : 2 ) ) - v0, syk = tyk >> 4, fyk = tyk & 15;
      const int16_t* tvp = tmp + hv * rowsT * ldsPitch;
      const bool anyFrac = __builtin_amdgcn_ballot_w64( fyk != 0 ) != 0;              // wave-uniform: see predRow
      VTaps<NT> vt;
      if( anyFrac ) loadVTaps<K0, K1>( tapL, fyk, vt );
      uint32_t sres;
      if( kind == TK_16F )
      {
        uint32_t D[4];                                                              // original - prediction averages, packed pairs
        const int16_t* tv = tvp + txi * 16;
        uint32_t pa[4], pb[4]; int ap[4];
        predRow<K0, K1>( tv, ldsPitch, tyi * 16 + 2 * r, syk, anyFrac, vt, headRoom, maxPk, pa );
        predRow<K0, K1>( tv, ldsPitch, tyi * 16 + 2 * r + 1, syk, anyFrac, vt, headRoom, maxPk, pb );
        avgInts( pa, pb, ap );
        D[0] = pkSub( o4[0], pkInts( ap[0], ap[1] ) ); D[1] = pkSub( o4[1], pkInts( ap[2], ap[3] ) );
        predRow<K0, K1>( tv + 8, ldsPitch, tyi * 16 + 2 * r, syk, anyFrac, vt, headRoom, maxPk, pa );
        predRow<K0, K1>( tv + 8, ldsPitch, tyi * 16 + 2 * r + 1, syk, anyFrac, vt, headRoom, maxPk, pb );
        avgInts( pa, pb, ap );
        D[2] = pkSub( o4[2], pkInts( ap[0], ap[1] ) ); D[3] = pkSub( o4[3], pkInts( ap[2], ap[3] ) );
        sres = hadTeamPkD( D, r, GEN ? LT : 8, TK_16F, lane );
      }
      else
      {
        // the lane's two operand rows as packed sample pairs: original o4[], prediction p4[]
        uint32_t p4[4];
        if( rows4 )
        {
          // a 4-wide block (one tile column): two rows of four samples per lane; the first pass worked on 8 columns, the upper four are not part of the block
          const int row = tyi * 8 + 2 * r;
          uint32_t pa[4], pb[4];
          predRow<K0, K1>( tvp, ldsPitch, row, syk, anyFrac, vt, headRoom, maxPk, pa );
          predRow<K0, K1>( tvp, ldsPitch, row + 1, syk, anyFrac, vt, headRoom, maxPk, pb );
          p4[0] = pa[0]; p4[1] = pa[1]; p4[2] = pb[0]; p4[3] = pb[1];
        }
        else
        {
          const int row = ( GEN && kind == TK_16x8 ) ? tyi * 8 + ( r & 7 ) : ( tyi << log2PH ) + r, col = ( GEN && kind == TK_16x8 ) ? txi * 16 + 8 * ( r >> 3 ) : txi * 8;
          predRow<K0, K1>( tvp + col, ldsPitch, row, syk, anyFrac, vt, headRoom, maxPk, p4 );
        }
        if( kind == TK_ROWS )
        {
          uint32_t sa = 0;
#pragma unroll
          for( int i = 0; i < 4; i++ ) sa = __builtin_amdgcn_sad_u16( o4[i] ^ BIAS, p4[i] ^ BIAS, sa );
          sres = vvhipGroupSum32( sa, LT, lane );
        }
        else sres = hadTeamPk( o4, p4, r, GEN ? LT : 8, GEN ? kind : TK_8x8, lane );
      }
      if( valid && r == 0 ) atomicAdd( &costL[pk & 0xff], sres );
      }      // positions
    }        // trips
    }      // passes
    ST_SYNC();     
    if( unit & ST_UNIT_MORE ) continue;                                            // the wave's next unit belongs to the same stage and adds to the same sums
    // the stage's nine costs (0 for positions outside the mask).  A block of several units is shared by the two waves of this workgroup (the schedule puts them side by
    // side, half of the units each): the second wave hands its sums over through LDS, the first stores the totals — no atomics on the cost array, no clearing of it before the launch
    if( GEN && ( unit & ST_UNIT_ATOMIC ) )
    {
      // a stage of more than two units (128-wide or 128-high blocks): the units are independent waves, their sums meet in the cost array (cleared in front of the launch;
      // integer additions commute: the result does not depend on the schedule)
      if( tid < 9 && ( ( j.mask >> tid ) & 1 ) ) atomicAdd( reinterpret_cast<unsigned long long*>( a.stageCost ) + ( size_t ) 9 * stage + tid, ( unsigned long long ) costL[tid] );
    }
    else
    {
      // a unit restricted to one variant stores the costs of ITS positions (slots ownBeg .. ownEnd of the grouped position list); variant 0's wave also stores the zeros of the
      // positions outside the mask.  An unrestricted unit: all nine.
      const int ownBeg = varOnly <= 1 ? 0 : ( varOnly == 2 ? cnt0 : cnt0 + cnt1 ), ownEnd = !varOnly ? nPos : ( varOnly == 1 ? cnt0 : ( varOnly == 2 ? cnt0 + cnt1 : nPos ) );
      const bool pair = !( h <= 32 && w <= 64 );
      if( pair ) { if( wv == 1 && tid < 9 ) pairCost[tid] = costL[tid]; __syncthreads(); }
      if( !pair || wv == 0 )
      {
        if( !varOnly ) { if( tid < 9 ) a.stageCost[( size_t ) 9 * stage + tid] = ( ( j.mask >> tid ) & 1 ) ? costL[tid] + ( pair ? pairCost[tid] : 0u ) : 0u; }
        else
        {
          if( tid >= ownBeg && tid < ownEnd ) { const int k = posL[tid] & 0xff; a.stageCost[( size_t ) 9 * stage + k] = costL[k] + ( pair ? pairCost[k] : 0u ); }
          if( varOnly == 1 && tid < 9 && !( ( j.mask >> tid ) & 1 ) ) a.stageCost[( size_t ) 9 * stage + tid] = 0u;
        }
      }
      if( pair ) __syncthreads();                                                // (a workgroup may hold several such pairs in a row: pairCost is read before the next pair writes it)
    }
  }
}

// =================================================================================================================================================
// (C) plain table calls on blocks of any two planes
// =================================================================================================================================================
__device__ __forceinline__ uint32_t ld4( const int16_t* p ) { struct __attribute__( ( packed, aligned( 2 ) ) ) U4 { uint32_t v; }; return ( ( const ME_GLOBAL U4* ) p )->v; }

// masked SAD (xGetSADwMask, RdCost.cpp:2062-2093): lane teams on row chunks like the plain SAD; the mask block is compact, one row per evaluated row
__device__ __forceinline__ void maskItemBody( const MeArgs& a, const WaveSpan span, int nItems, const int16_t* const* planeL, const int* strideL, int lane )
{
  const int first = span.first, count = -span.count;
  const vvhip_me_mask_item f = a.maskItems[first];
  const int w = f.width, h = f.height, ss = f.sub_shift;
  const int cw = w >= 8 ? 8 : ( w >= 4 ? 4 : 2 ), lprShift = ( 31 - __builtin_clz( w ) ) - ( 31 - __builtin_clz( cw ) ), lpr = 1 << lprShift, rowsEff = h >> ss, chunks = rowsEff * lpr;
  const int lpcShift = chunks >= 64 ? 6 : 31 - __builtin_clz( chunks ), lpc = 1 << lpcShift;          // lanes per item: min( 64, chunks ), a power of two
  const int teams = 64 >> lpcShift, lt = lane & ( lpc - 1 ), team = lane >> lpcShift;
  for( int i0 = 0; i0 < count; i0 += teams )
  {
    const int ii = i0 + team;
    const bool valid = ii < count;
    const int idx = a.itemOrder[nItems + first + ( valid ? ii : 0 )];
    const vvhip_me_mask_item it = a.maskItems[first + ( valid ? ii : 0 )];
    const int os = strideL[it.org_plane] ? strideL[it.org_plane] : w, cs = strideL[it.cur_plane] ? strideL[it.cur_plane] : w;
    const int16_t* po = planeL[it.org_plane] + it.org_off; const int16_t* pc = planeL[it.cur_plane] + it.cur_off; const int16_t* pm = planeL[it.mask_plane] + it.mask_off;
    const int r0 = lt >> lprShift, s0 = lt & ( lpr - 1 ), rowStep = lpc >> lprShift;
    // 32-bit sums while every sample of the wave's operands lies in [0, 4096) and every weight in [0, 16) (GEO: samples and weights 0..8): a product is < 2^16, a lane's <= 256
    // products < 2^24, the item's sum < 2^30; any other int16 operands (the interface's contract; the reference sums in 64 bits, RdCost.cpp:2062-2093): the wave repeats the item
    // with 64-bit sums (a 128x128 block of large differences overflowed the 32-bit form)
    uint32_t sum = 0, bits = 0, mbits = 0;
    for( int r = r0; r < rowsEff; r += rowStep )
    {
      const int16_t* qa = po + ( ptrdiff_t ) ( r << ss ) * os + s0 * cw; const int16_t* qb = pc + ( ptrdiff_t ) ( r << ss ) * cs + s0 * cw; const int16_t* qm = pm + r * w + s0 * cw;
      uint32_t va[4] = { 0, 0, 0, 0 }, vb[4] = { 0, 0, 0, 0 }, vm[4] = { 0, 0, 0, 0 };
      if( cw == 8 )      { const u32x4 x = ld16( qa ), z = ld16( qb ), m = ld16( qm ); va[0] = x.x; va[1] = x.y; va[2] = x.z; va[3] = x.w; vb[0] = z.x; vb[1] = z.y; vb[2] = z.z; vb[3] = z.w; vm[0] = m.x; vm[1] = m.y; vm[2] = m.z; vm[3] = m.w; }
      else if( cw == 4 ) { const u32x2 x = ld8( qa ), z = ld8( qb ), m = ld8( qm ); va[0] = x.x; va[1] = x.y; vb[0] = z.x; vb[1] = z.y; vm[0] = m.x; vm[1] = m.y; }
      else               { va[0] = ld4( qa ); vb[0] = ld4( qb ); vm[0] = ld4( qm ); }
#pragma unroll
      for( int q = 0; q < 4; q++ )
      {
        bits |= va[q] | vb[q]; mbits |= vm[q];
        sum += ( uint32_t ) ( abs( lo16( va[q] ) - lo16( vb[q] ) ) * lo16( vm[q] ) + abs( hi16( va[q] ) - hi16( vb[q] ) ) * hi16( vm[q] ) );
      }
    }
    uint64_t t;
    if( __builtin_amdgcn_ballot_w64( ( ( bits & 0xf000f000u ) | ( mbits & 0xfff0fff0u ) ) != 0 ) == 0ull ) t = vvhipGroupSum32( sum, lpc, lane );
    else
    {
      uint64_t sum64 = 0;
      for( int r = r0; r < rowsEff; r += rowStep )
      {
        const int16_t* qa = po + ( ptrdiff_t ) ( r << ss ) * os + s0 * cw; const int16_t* qb = pc + ( ptrdiff_t ) ( r << ss ) * cs + s0 * cw; const int16_t* qm = pm + r * w + s0 * cw;
        for( int x = 0; x < cw; x++ ) sum64 += ( uint64_t ) ( ( int64_t ) abs( ( int ) qa[x] - ( int ) qb[x] ) * ( int64_t ) qm[x] );          // (Distortion is unsigned 64-bit; the product as the reference forms it)
      }
      t = vvhipGroupSum64( sum64, lpc, lane );
    }
    if( valid && lt == 0 ) a.itemCost[idx] = ( uint64_t ) t << ss;                 // RdCost.cpp:2090
  }
}

// one chunk (CW = 8, 4 or 2 samples) of both operands of a SAD / SSE item as CW / 2 packed pairs
template<int CW>
__device__ __forceinline__ void itemChunk( const int16_t* qa, const int16_t* qb, uint32_t ( &va )[CW / 2], uint32_t ( &vb )[CW / 2] )
{
  if( CW == 8 )      { const u32x4 x = ld16( qa ), z = ld16( qb ); va[0] = x.x; va[1] = x.y; va[2 % ( CW / 2 )] = x.z; va[3 % ( CW / 2 )] = x.w; vb[0] = z.x; vb[1] = z.y; vb[2 % ( CW / 2 )] = z.z; vb[3 % ( CW / 2 )] = z.w; }
  else if( CW == 4 ) { const u32x2 x = ld8( qa ), z = ld8( qb ); va[0] = x.x; va[1 % ( CW / 2 )] = x.y; vb[0] = z.x; vb[1 % ( CW / 2 )] = z.y; }
  else               { va[0] = ld4( qa ); vb[0] = ld4( qb ); }
}
// SAD / SSE of one item by a team of lanes: the lane's nIt chunks of CW samples start at pa / pb and lie stepA / stepB samples apart.  Returns the LANE's sum (SAD: 32 bits).
// SSE: what the encoder hands to this entry are samples — when every sample of the wave's operands lies in [0, 4096) a difference fits 13 bits, a lane's <= 256 squares fit
// 32 bits (256 x 4095^2 < 2^32) and a sample pair costs a packed subtraction and a v_dot2_i32_i16 (+ one v_or3_b32 that collects the operands' bits for the test); otherwise
// (any int16 operands: the interface's contract) the wave repeats the item with 64-bit multiply-adds.
template<int CW>
__device__ __forceinline__ uint64_t itemSadSse( bool isSad, const int16_t* pa, const int16_t* pb, ptrdiff_t stepA, ptrdiff_t stepB, int nIt )
{
  if( isSad )                                                                      // (wave-uniform: the span's function)
  {
    uint32_t sad = 0;
    for( int it = nIt; it > 0; it--, pa += stepA, pb += stepB )
    {
      uint32_t va[CW / 2], vb[CW / 2];
      itemChunk<CW>( pa, pb, va, vb );
#pragma unroll
      for( int q = 0; q < CW / 2; q++ ) sad = __builtin_amdgcn_sad_u16( va[q] ^ BIAS, vb[q] ^ BIAS, sad );
    }
    return sad;
  }
  uint32_t acc = 0, bits = 0;
  {
    const int16_t* qa = pa; const int16_t* qb = pb;
    for( int it = nIt; it > 0; it--, qa += stepA, qb += stepB )
    {
      uint32_t va[CW / 2], vb[CW / 2];
      itemChunk<CW>( qa, qb, va, vb );
#pragma unroll
      for( int q = 0; q < CW / 2; q++ ) { bits |= va[q] | vb[q]; const uint32_t df = pkSub( va[q], vb[q] ); acc = ( uint32_t ) dot2( df, df, ( int ) acc ); }
    }
  }
  if( __builtin_amdgcn_ballot_w64( ( bits & 0xf000f000u ) != 0 ) == 0ull ) return acc;
  uint64_t sse = 0;
  for( int it = nIt; it > 0; it--, pa += stepA, pb += stepB )
  {
    uint32_t va[CW / 2], vb[CW / 2];
    itemChunk<CW>( pa, pb, va, vb );
#pragma unroll
    for( int q = 0; q < CW / 2; q++ ) { const int d0 = lo16( va[q] ) - lo16( vb[q] ), d1 = hi16( va[q] ) - hi16( vb[q] ); sse += ( uint64_t ) ( ( int64_t ) d0 * d0 ) + ( uint64_t ) ( ( int64_t ) d1 * d1 ); }
  }
  return sse;
}

// GEN = false: what the fast presets call (and every SAD / SSE at least four samples wide): lane teams on row chunks, 8x8 / 16x16_fast tiles with eight lanes per tile, the 4x4
// block — the round-3 body, 57 registers.  GEN = true (its own launch): the rectangular tiles, 2x2 tiles, two-sample-wide blocks, masked SADs.
template<bool GEN>
__device__ __forceinline__ void itemBody( const MePlanes& P, const MeArgs& a, int wave, int nItems )
{
  // the plane table in LDS: an item's planes are per-lane indices, and a per-lane index into the kernel arguments is a memory access behind the item record — one more link
  // in a chain (wave record -> item -> plane -> samples) that is all a short-lived wave does
  __shared__ const int16_t* planeL[16];
  __shared__ int strideL[16];
  const WaveSpan span = a.itemWaves[wave];
  const int lane = threadIdx.x & 63;
  if( lane < 16 ) { planeL[lane] = P.p[lane]; strideL[lane] = P.stride[lane]; }      // (every wave of the workgroup writes the same values)
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();
  if( GEN && span.count < 0 ) { maskItemBody( a, span, nItems, planeL, strideL, lane ); return; }
  const vvhip_me_item first = a.items[span.first];                                  // every item of the span has this function and geometry (the table is in schedule order)
  const int w = first.width, h = first.height, func = first.func, ss = func == VVHIP_DF_SAD ? first.sub_shift : 0;
  if( func == VVHIP_DF_SAD || func == VVHIP_DF_SSE )
  {
    const int cw = w >= 8 ? 8 : ( ( !GEN || w >= 4 ) ? 4 : 2 ), lprShift = ( 31 - __builtin_clz( w ) ) - ( 31 - __builtin_clz( cw ) ), lpr = 1 << lprShift, rowsEff = h >> ss, chunks = rowsEff * lpr;
    const int lpcShift = chunks >= 64 ? 6 : 31 - __builtin_clz( chunks ), lpc = 1 << lpcShift;          // lanes per item: min( 64, chunks ), a power of two
    const int teams = 64 >> lpcShift, lt = lane & ( lpc - 1 ), team = lane >> lpcShift;
    for( int i0 = 0; i0 < span.count; i0 += teams )
    {
      const int ii = i0 + team;
      const bool valid = ii < span.count;
      const int idx = a.itemOrder[span.first + ( valid ? ii : 0 )];            // where the result goes
      const vvhip_me_item it = a.items[span.first + ( valid ? ii : 0 )];
      const int16_t* po = planeL[it.org_plane] + it.org_off; const int os = strideL[it.org_plane] ? strideL[it.org_plane] : w;      // (stride 0: a pool of compact blocks)
      const int16_t* pc = planeL[it.cur_plane] + it.cur_off; const int cs = strideL[it.cur_plane] ? strideL[it.cur_plane] : w;
      // a lane's chunks are lpc apart: the same column piece, rowStep rows further down (lpr, chunks, lpc are powers of two, lpc >= lpr): no division, constant address steps.
      // (four chunks per lane in flight were measured slower: 23.8 -> 24.6 us, the intra picture 123 -> 143 us)
      const int r0 = lt >> lprShift, s0 = lt & ( lpr - 1 ), rowStep = lpc >> lprShift;
      const int16_t* pa = po + ( ptrdiff_t ) ( r0 << ss ) * os + s0 * cw;
      const int16_t* pb = pc + ( ptrdiff_t ) ( r0 << ss ) * cs + s0 * cw;
      const ptrdiff_t stepA = ( ptrdiff_t ) ( rowStep << ss ) * os, stepB = ( ptrdiff_t ) ( rowStep << ss ) * cs;
      const int nIt = chunks >> lpcShift;
      const bool isSad = func == VVHIP_DF_SAD;
      const uint64_t mine = cw == 8 ? itemSadSse<8>( isSad, pa, pb, stepA, stepB, nIt ) : ( ( !GEN || cw == 4 ) ? itemSadSse<4>( isSad, pa, pb, stepA, stepB, nIt ) : itemSadSse<2>( isSad, pa, pb, stepA, stepB, nIt ) );
      if( isSad ) { const uint32_t t = vvhipGroupSum32( ( uint32_t ) mine, lpc, lane ); if( valid && lt == 0 ) a.itemCost[idx] = ( uint64_t ) t << ss; }
      else        { const uint64_t t = vvhipGroupSum64( mine, lpc, lane ); if( valid && lt == 0 ) a.itemCost[idx] = t; }
    }
    return;
  }
  // Hadamard family: the reference's tile ladder (hadTileKind), a team of lanes per tile (table in front of hadTeam; every sample of an item is requested at once: one memory
  // latency per item), one lane per 4x4 block and per 2x2 tile.  HAD_2SAD = min( HAD, 2 SAD ) (RdCost.cpp:1768-1816; its SAD runs over all samples of the block).
  const int kind = GEN ? hadTileKind( w, h, func == VVHIP_DF_HAD_FAST ) : ( w == 4 ? TK_4x4 : ( ( func == VVHIP_DF_HAD_FAST && ( w & 31 ) == 0 ) ? TK_16F : TK_8x8 ) );
  if( !GEN && kind == TK_4x4 )                                           // (only the 4x4 block: 4 x N and N x 4 blocks use the 4x8 / 8x4 tiles)
  {
    const int ii = lane;
    const bool valid = ii < span.count;
    const int idx = a.itemOrder[span.first + ( valid ? ii : 0 )];
    const vvhip_me_item it = a.items[span.first + ( valid ? ii : 0 )];
    const int16_t* qa = planeL[it.org_plane] + it.org_off; const int os = strideL[it.org_plane] ? strideL[it.org_plane] : w;
    const int16_t* qb = planeL[it.cur_plane] + it.cur_off; const int cs = strideL[it.cur_plane] ? strideL[it.cur_plane] : w;
    int d[16]; uint32_t sad = 0;
#pragma unroll
    for( int r = 0; r < 4; r++ )
    {
      const u32x2 x = ld8( qa + ( ptrdiff_t ) r * os ), z = ld8( qb + ( ptrdiff_t ) r * cs );
      d[4 * r] = lo16( x.x ) - lo16( z.x ); d[4 * r + 1] = hi16( x.x ) - hi16( z.x ); d[4 * r + 2] = lo16( x.y ) - lo16( z.y ); d[4 * r + 3] = hi16( x.y ) - hi16( z.y );
      if( func == VVHIP_DF_HAD_2SAD ) { sad = __builtin_amdgcn_sad_u16( x.x ^ BIAS, z.x ^ BIAS, sad ); sad = __builtin_amdgcn_sad_u16( x.y ^ BIAS, z.y ^ BIAS, sad ); }
    }
    const uint64_t tot = hadamard4x4( d ), s2 = 2ull * sad;
    if( valid ) a.itemCost[idx] = ( func == VVHIP_DF_HAD_2SAD && s2 < tot ) ? s2 : tot;
    return;
  }
  __shared__ uint32_t accAll[4][128];                                 // per wave and item of its span: Hadamard sum, SAD
  uint32_t* accL = accAll[( threadIdx.x >> 6 ) & 3];
  accL[lane] = 0; accL[64 + lane] = 0;
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();
  const int PW = GEN ? tileW( kind ) : ( kind == TK_16F ? 16 : 8 ), PH = GEN ? tileH( kind ) : PW, LT = GEN ? tileLanes( kind ) : 8, log2LT = GEN ? 31 - __builtin_clz( LT ) : 3;
  const int log2TX = ( 31 - __builtin_clz( w ) ) - ( 31 - __builtin_clz( PW ) ), tilesX = 1 << log2TX, log2Slots = log2TX + ( 31 - __builtin_clz( h ) ) - ( 31 - __builtin_clz( PH ) ) + log2LT;
  const int slotsPerItem = 1 << log2Slots, total = span.count * slotsPerItem;
  for( int s0 = 0; s0 < total; s0 += 64 )
  {
    const int sl = s0 + lane;
    const bool valid = sl < total;
    const int sv = valid ? sl : 0, ii = sv >> log2Slots, rem = sv & ( slotsPerItem - 1 ), t = rem >> log2LT, r = rem & ( LT - 1 );      // (powers of two)
    const vvhip_me_item it = a.items[span.first + ii];
    const int os = strideL[it.org_plane] ? strideL[it.org_plane] : w, cs = strideL[it.cur_plane] ? strideL[it.cur_plane] : w;
    const int tyi = t >> log2TX, txi = t & ( tilesX - 1 );
    const int16_t* qa = planeL[it.org_plane] + it.org_off + ( ptrdiff_t ) ( tyi * PH ) * os + txi * PW;
    const int16_t* qb = planeL[it.cur_plane] + it.cur_off + ( ptrdiff_t ) ( tyi * PH ) * cs + txi * PW;
    uint32_t sad = 0, sres;
    if( GEN && kind == TK_2x2 )
    {
      // xCalcHADs2x2 (RdCost.cpp:1006-1026): one lane per tile
      const uint32_t x0 = ld4( qa ), x1 = ld4( qa + os ), z0 = ld4( qb ), z1 = ld4( qb + cs );
      const int d0 = lo16( x0 ) - lo16( z0 ), d1 = hi16( x0 ) - hi16( z0 ), d2 = lo16( x1 ) - lo16( z1 ), d3 = hi16( x1 ) - hi16( z1 );
      const int m0 = d0 + d2, m1 = d1 + d3, m2 = d0 - d2, m3 = d1 - d3;
      sres = ( ( uint32_t ) abs( m0 + m1 ) >> 2 ) + ( uint32_t ) abs( m0 - m1 ) + ( uint32_t ) abs( m2 + m3 ) + ( uint32_t ) abs( m2 - m3 );
      sad = ( uint32_t ) ( abs( d0 ) + abs( d1 ) + abs( d2 ) + abs( d3 ) );
    }
    else
    {
      if( kind == TK_16F )
      {
        const int16_t* p0 = qa + ( ptrdiff_t ) ( 2 * r ) * os; const int16_t* p1 = qb + ( ptrdiff_t ) ( 2 * r ) * cs;
        const u32x4 a0 = ld16( p0 ), a1 = ld16( p0 + 8 ), b0 = ld16( p0 + os ), b1 = ld16( p0 + os + 8 );
        const u32x4 c0 = ld16( p1 ), c1 = ld16( p1 + 8 ), e0 = ld16( p1 + cs ), e1 = ld16( p1 + cs + 8 );
        int ao[4], ac[4]; uint32_t D[4];
        { const uint32_t x[4] = { a0.x, a0.y, a0.z, a0.w }, y[4] = { b0.x, b0.y, b0.z, b0.w }; avgInts( x, y, ao ); }
        { const uint32_t x[4] = { c0.x, c0.y, c0.z, c0.w }, y[4] = { e0.x, e0.y, e0.z, e0.w }; avgInts( x, y, ac ); }
        D[0] = pkInts( ao[0] - ac[0], ao[1] - ac[1] ); D[1] = pkInts( ao[2] - ac[2], ao[3] - ac[3] );
        { const uint32_t x[4] = { a1.x, a1.y, a1.z, a1.w }, y[4] = { b1.x, b1.y, b1.z, b1.w }; avgInts( x, y, ao ); }
        { const uint32_t x[4] = { c1.x, c1.y, c1.z, c1.w }, y[4] = { e1.x, e1.y, e1.z, e1.w }; avgInts( x, y, ac ); }
        D[2] = pkInts( ao[0] - ac[0], ao[1] - ac[1] ); D[3] = pkInts( ao[2] - ac[2], ao[3] - ac[3] );
        sres = hadTeamPkD( D, r, GEN ? LT : 8, TK_16F, lane );
      }
      else
      {
        uint32_t xw[4], zw[4];                                           // the lane's operand rows as packed sample pairs
        if( GEN && kind == TK_4x8 )
        {
          const u32x2 xa = ld8( qa + ( ptrdiff_t ) ( 2 * r ) * os ), xb = ld8( qa + ( ptrdiff_t ) ( 2 * r + 1 ) * os ), za = ld8( qb + ( ptrdiff_t ) ( 2 * r ) * cs ), zb = ld8( qb + ( ptrdiff_t ) ( 2 * r + 1 ) * cs );
          xw[0] = xa.x; xw[1] = xa.y; xw[2] = xb.x; xw[3] = xb.y; zw[0] = za.x; zw[1] = za.y; zw[2] = zb.x; zw[3] = zb.y;
        }
        else
        {
          const int row = ( GEN && kind == TK_16x8 ) ? ( r & 7 ) : r, col = ( GEN && kind == TK_16x8 ) ? 8 * ( r >> 3 ) : 0;
          const u32x4 x = ld16( qa + ( ptrdiff_t ) row * os + col ), z = ld16( qb + ( ptrdiff_t ) row * cs + col );
          xw[0] = x.x; xw[1] = x.y; xw[2] = x.z; xw[3] = x.w; zw[0] = z.x; zw[1] = z.y; zw[2] = z.z; zw[3] = z.w;
        }
        if( func == VVHIP_DF_HAD_2SAD )
        {
#pragma unroll
          for( int i = 0; i < 4; i++ ) sad = __builtin_amdgcn_sad_u16( xw[i] ^ BIAS, zw[i] ^ BIAS, sad );
        }
        sres = hadTeamPk( xw, zw, r, GEN ? LT : 8, GEN ? kind : TK_8x8, lane );
      }
      if( func == VVHIP_DF_HAD_2SAD ) sad = vvhipGroupSum32( sad, GEN ? LT : 8, lane );
    }
    if( valid && r == 0 ) { atomicAdd( &accL[ii], sres ); if( func == VVHIP_DF_HAD_2SAD ) atomicAdd( &accL[64 + ii], sad ); }
  }
  __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier();
  if( lane < span.count )
  {
    const int idx = a.itemOrder[span.first + lane];
    const uint64_t tot = accL[lane], s2 = 2ull * accL[64 + lane];
    a.itemCost[idx] = ( func == VVHIP_DF_HAD_2SAD && s2 < tot ) ? s2 : tot;
  }
}

// three kernels (their register budgets differ widely); a plan's run launches the ones it needs back to back
// (one instance per tap support: the 4-tap search filter of the fast presets must not pay the registers of the 8-tap window)
// two waves per workgroup, one bundle each: independent (wave-level synchronisation, own LDS slice) except for the two bands of a 64x64 block, which the schedule gives to
// the two waves of one workgroup (they add their sums through LDS)
template<int K0, int K1, bool GEN>
__global__ void __launch_bounds__( 128 ) __attribute__( ( amdgpu_waves_per_eu( VVHIP_ME_STAGE_WAVES, VVHIP_ME_STAGE_WAVES ) ) )
meStageKernel( MePlanes P, MeArgs a, int firstWave, int nWaves, int ldsPerWave )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t meLds[];
  __shared__ uint32_t pairCost[16];
  const int wv = __builtin_amdgcn_readfirstlane( ( int ) ( threadIdx.x >> 6 ) ), wave = blockIdx.x * 2 + wv;
  if( wave < nWaves ) stageBody<K0, K1, GEN>( P, a, a.stageWaves[firstWave + wave], meLds + wv * ( ldsPerWave >> 1 ), pairCost, wv );
}

// workgroups 0 .. nBig - 1: one large window each (four waves share it); the others: four small windows each, one per wave
__global__ void __launch_bounds__( 256 )
meIntKernel( MePlanes P, MeArgs a, int nBig, int ldsSmall, int blockBase )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t meLds[];
  const int blk = ( int ) blockIdx.x + blockBase;                                   // (the two window classes may be two launches: vvhip_me_plan.intSplit)
  if( blk < nBig ) { intBody<false>( P, a, blk, meLds ); return; }
  const int wv = __builtin_amdgcn_readfirstlane( ( int ) ( threadIdx.x >> 6 ) );
  const int job = nBig + ( blk - nBig ) * 4 + wv;
  if( job < a.wavesInt ) intBody<true>( P, a, job, meLds + wv * ( ldsSmall >> 1 ) );
}

// WAVES independent waves per workgroup (wave-level synchronisation only).  A B picture's ~13 000 one-pass waves are bound by the rate workgroups start at: four per workgroup
// 23.6 -> 21.2 us (eight: 19.2, but long lists lose: a workgroup holds its slots until its slowest wave ends — the intra picture's 168 000 waves 116 -> 129 / 143 us), so long
// lists keep single-wave workgroups.  GEN: see itemBody (waves firstWave .. of the plan's item schedule)
// spansPerWave: a wave takes this many consecutive spans of the schedule (long lists: the intra picture's 168 000 one-pass waves are bound by the rate workgroups start at)
template<int WAVES, bool GEN>
__global__ void __launch_bounds__( 64 * WAVES ) __attribute__( ( amdgpu_waves_per_eu( GEN ? VVHIP_ME_ITEM_WAVES - 1 : VVHIP_ME_ITEM_WAVES, GEN ? VVHIP_ME_ITEM_WAVES - 1 : VVHIP_ME_ITEM_WAVES ) ) )      // (the generic body needs 3 registers more than 8 waves leave: 7 there, no scratch)
meItemKernel( MePlanes P, MeArgs a, int nItems, int firstWave, int nWaves, int spansPerWave )
{
  // (the wave's index as a scalar: everything derived from its span record — function, geometry, loop bounds — is then wave-uniform for the compiler too: scalar loads and branches
  //  instead of per-lane copies under exec masks)
  const int wave = ( ( int ) blockIdx.x * WAVES + __builtin_amdgcn_readfirstlane( ( int ) ( threadIdx.x >> 6 ) ) ) * spansPerWave;
  for( int q = 0; q < spansPerWave; q++ )
    if( wave + q < nWaves ) { itemBody<GEN>( P, a, firstWave + wave + q, nItems ); __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront" ); __builtin_amdgcn_wave_barrier(); }
}

// XCD-aware order of a launch's workgroups.  Workgroups are dealt round-robin to the 8 XCDs, each with a private 4 MB L2.  The entries of a class —
// `group` consecutive schedule entries are one workgroup — are sorted by their position in the picture (raster order of the reference offset = horizontal bands); workgroup
// `base + l` of the launch sits on XCD ( base + l ) % 8 and is handed the next entry of the ( ( base + l ) % 8 )-th contiguous eighth of the class: every L2 streams one band of
// the planes instead of blocks from the whole picture.  perm[l] = which sorted workgroup runs as the l-th of the class.
std::vector<int> xcdBandOrder( int nGroups, int base )
{
  std::vector<int> perm( nGroups );
  const int q = ( nGroups + 7 ) / 8;
  int cursor[8], endOf[8];
  for( int x = 0; x < 8; x++ ) { cursor[x] = std::min( nGroups, x * q ); endOf[x] = std::min( nGroups, ( x + 1 ) * q ); }
  for( int l = 0; l < nGroups; l++ )
  {
    int x = ( base + l ) & 7;
    for( int t = 0; t < 8 && cursor[x] >= endOf[x]; t++ ) x = ( x + 1 ) & 7;      // (uneven split: a finished eighth borrows from its neighbour)
    perm[l] = cursor[x]++;
  }
  return perm;
}
bool xcdBandOn() { static const bool on = !( getenv( "VVHIP_ME_XCD_BAND" ) && atoi( getenv( "VVHIP_ME_XCD_BAND" ) ) == 0 ); return on; }

int hostWinPitch( int winW ) { return 2 * ( ( ( winW + 3 ) >> 1 ) | 1 ); }
int hostWinSamples( int winW, int winH ) { return ( winH * hostWinPitch( winW ) + 7 ) & ~7; }      // the window part of a job's LDS, in samples (the original block behind it is 16-byte aligned)

} // namespace

#ifndef VVHIP_ME_KERNELS_ONLY      // (tools/exp/persist_stage.hip includes this file for its kernels only)
extern "C" {

static int mePlanCreate( vvhip_ctx* ctx, const vvhip_me_lists& L, int bit_depth, int max_window, vvhip_me_plan** out );

// host memory out: the six interpolation tap tables a plan of this bit depth gives its stage kernels (stageTapTables) — for checks of the kernels' arithmetic without a device
int vvhip_get_me_tap_tables_host( int bit_depth, int32_t* host_out )
{
  if( bit_depth < 8 || bit_depth > 10 || !host_out ) return VVHIP_E_ARG;
  const std::vector<int32_t> t = stageTapTables( bit_depth );
  memcpy( host_out, t.data(), t.size() * sizeof( int32_t ) );
  return VVHIP_OK;
}

int vvhip_me_plan_create( vvhip_ctx* ctx, const vvhip_me_int_job* int_jobs, int n_int_jobs, const vvhip_me_cand* cands, int n_cands,
                          const vvhip_me_stage_job* stage_jobs, int n_stage_jobs, const vvhip_me_item* items, int n_items, int bit_depth, int max_window, vvhip_me_plan** out )
{
  vvhip_me_lists L; L.int_jobs = int_jobs; L.n_int_jobs = n_int_jobs; L.cands = cands; L.n_cands = n_cands; L.stage_jobs = stage_jobs; L.n_stage_jobs = n_stage_jobs;
  L.items = items; L.n_items = n_items; L.mask_items = nullptr; L.n_mask_items = 0;
  return mePlanCreate( ctx, L, bit_depth, max_window, out );
}

int vvhip_me_plan_create_lists( vvhip_ctx* ctx, const vvhip_me_lists* lists, int bit_depth, int max_window, vvhip_me_plan** out )
{
  if( !ctx || !lists ) return VVHIP_E_ARG;
  return mePlanCreate( ctx, *lists, bit_depth, max_window, out );
}

static int mePlanCreate( vvhip_ctx* ctx, const vvhip_me_lists& L, int bit_depth, int max_window, vvhip_me_plan** out )
{
  if( !ctx || !out ) return VVHIP_E_ARG;
  *out = nullptr;
  const vvhip_me_int_job* int_jobs = L.int_jobs; const vvhip_me_cand* cands = L.cands; const vvhip_me_stage_job* stage_jobs = L.stage_jobs; const vvhip_me_item* items = L.items;
  const vvhip_me_mask_item* mask_items = L.mask_items;
  const int n_int_jobs = L.n_int_jobs, n_cands = L.n_cands, n_stage_jobs = L.n_stage_jobs, n_items = L.n_items, n_mask = L.n_mask_items;
  if( n_int_jobs < 0 || n_cands < 0 || n_stage_jobs < 0 || n_items < 0 || n_mask < 0 || ( n_int_jobs && ( !int_jobs || !cands ) ) || ( n_stage_jobs && !stage_jobs ) || ( n_items && !items ) || ( n_mask && !mask_items ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_create: bad lists" );
  if( n_cands >= ( 1 << 24 ) ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_create: %d candidates (the schedule packs the output-list position into 24 bits)", n_cands );
  if( n_stage_jobs > ST_STAGE_MASK ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_create: %d stage jobs (the schedule packs the stage index into 22 bits)", n_stage_jobs );
  if( bit_depth < 8 || bit_depth > 10 ) return vvhip_fail( ctx, VVHIP_E_UNSUPPORTED, "vvhip_me_plan_create: bit depth %d (the packed Hadamard tile covers <= 10)", bit_depth );
  if( max_window <= 0 ) max_window = 24;
  // block shapes: width and height independent powers of two (CTU 128 + multi-type tree: RdCost.cpp:301-336 generic SAD, the rectangular Hadamard tiles :1324-1766)
  auto shapeOk = []( int w, int h, int minW, int minH ) { return isPow2( w ) && isPow2( h ) && w >= minW && h >= minH && w <= 128 && h <= 128; };

  // ---- integer jobs: one window per cluster of candidates (greedy in list order: a candidate joins the first window it keeps within max_window)
  std::vector<IntJob> ij; std::vector<PlanCand> pc; std::vector<int32_t> candOut;      // candOut: the cost-array entries of the kept candidates, consecutive per candidate
  int ldsInt = 0;
  static const bool dedupe = !( getenv( "VVHIP_ME_DEDUPE" ) && atoi( getenv( "VVHIP_ME_DEDUPE" ) ) == 0 );      // (0: every position scored as often as the caller lists it — A/B measurements)
  static const int candCap = getenv( "VVHIP_ME_CAND_CAP" ) ? atoi( getenv( "VVHIP_ME_CAND_CAP" ) ) : 16;      // recorded 1080p B pictures, window launch with events: 16 / 32 / 64 / none -> 22.0 / 24.7 / 28.8 / 29.0 us
  for( int i = 0; i < n_int_jobs; i++ )
  {
    const vvhip_me_int_job& s = int_jobs[i];
    if( !shapeOk( s.width, s.height, 8, 4 ) || s.org_plane > 15 || s.ref_plane > 15 || s.sub_shift > 1 || ( s.height >> s.sub_shift ) < 1 || s.first_cand < 0 || s.n_cand < 0 || s.first_cand + s.n_cand > n_cands )
      return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_create: integer job %d (%dx%d, candidates %d+%d)", i, s.width, s.height, s.first_cand, s.n_cand );
    struct WinCand { int16_t dx, dy; std::vector<int32_t> outs; };
    struct Win { int x0, y0, x1, y1; std::vector<WinCand> c; };
    // a kept candidate stands for at most as many entries as its team has lanes (each lane stores one), min( 64, 16-byte chunks of the evaluated rows )
    const int teamLanes = std::min( 64, ( s.height >> s.sub_shift ) * ( s.width >> 3 ) );
    std::vector<Win> wins;
    // a 128-wide block's window must still fit the LDS of a workgroup: its reach shrinks with the block
    const int reach = std::min( max_window, s.width * s.height >= 128 * 64 ? 16 : max_window );
    for( int k = 0; k < s.n_cand; k++ )
    {
      const vvhip_me_cand& c = cands[s.first_cand + k];
      WinCand p; p.dx = c.dx; p.dy = c.dy; p.outs.assign( 1, s.first_cand + k );
      bool placed = false;
      if( dedupe )
        for( Win& wn : wins )
        {
          for( WinCand& q : wn.c ) if( q.dx == c.dx && q.dy == c.dy && ( int ) q.outs.size() < teamLanes ) { q.outs.push_back( s.first_cand + k ); placed = true; break; }
          if( placed ) break;
        }
      if( placed ) continue;
      for( Win& wn : wins )
      {
        if( ( int ) wn.c.size() >= candCap ) continue;          // a window's candidates are one workgroup's serial work: long lists (raster searches: ~300 positions) are cut
        const int x0 = std::min( wn.x0, ( int ) c.dx ), y0 = std::min( wn.y0, ( int ) c.dy ), x1 = std::max( wn.x1, ( int ) c.dx ), y1 = std::max( wn.y1, ( int ) c.dy );
        if( x1 - x0 <= reach && y1 - y0 <= reach ) { wn.x0 = x0; wn.y0 = y0; wn.x1 = x1; wn.y1 = y1; wn.c.push_back( p ); placed = true; break; }
      }
      if( !placed ) { Win wn; wn.x0 = wn.x1 = c.dx; wn.y0 = wn.y1 = c.dy; wn.c.push_back( p ); wins.push_back( wn ); }
    }
    for( const Win& wn : wins )
    {
      IntJob j; j.orgOff = s.org_off; j.refOff = s.ref_off; j.w = s.width; j.h = s.height; j.orgPlane = s.org_plane; j.refPlane = s.ref_plane; j.subShift = s.sub_shift; j.pad = 0;
      j.minDx = ( int16_t ) wn.x0; j.minDy = ( int16_t ) wn.y0; j.winW = ( int16_t ) ( wn.x1 - wn.x0 + s.width ); j.winH = ( int16_t ) ( wn.y1 - wn.y0 + s.height );
      j.firstCand = ( int32_t ) pc.size(); j.nCand = ( int32_t ) wn.c.size();
      for( const WinCand& q : wn.c )
      {
        PlanCand pcd; pcd.dx = q.dx; pcd.dy = q.dy; pcd.out = ( uint32_t ) candOut.size() | ( uint32_t ) ( q.outs.size() - 1 ) << 24;
        pc.push_back( pcd ); candOut.insert( candOut.end(), q.outs.begin(), q.outs.end() );
      }
      ij.push_back( j );
      ldsInt = std::max( ldsInt, ( hostWinSamples( j.winW, j.winH ) + ( s.height >> s.sub_shift ) * s.width ) * 2 + j.nCand * ( int ) sizeof( PlanCand ) );
    }
  }
  // windows that need much LDS first (their own launch), inside each class heaviest first
  auto ldsOf = []( const IntJob& j ) { return ( hostWinSamples( j.winW, j.winH ) + ( j.h >> j.subShift ) * j.w ) * 2 + j.nCand * ( int ) sizeof( PlanCand ); };      // window + original + candidate records
  // three LDS classes, one launch each when they differ much: a launch's dynamic LDS is its largest job's, so one 128x128 window (58 KB) would cap every 64x64 window (17 KB) of
  // the same launch at two workgroups per CU.  Class 0: > 24 KB, 1: > 6 KB (one workgroup per window), 2: the small ones (four windows per workgroup, one per wave)
  const int ldsSmallCap = 6 * 1024, ldsMidCap = 24 * 1024;
  const bool band = xcdBandOn();
  auto classOf = [&]( const IntJob& j ) { const int l = ldsOf( j ); return l > ldsMidCap ? 0 : ( l > ldsSmallCap ? 1 : 2 ); };
  std::stable_sort( ij.begin(), ij.end(), [&]( const IntJob& a, const IntJob& b ) { const int ca = classOf( a ), cb = classOf( b ); if( ca != cb ) return ca < cb;
                    if( band ) return a.refOff < b.refOff;                                    // picture order inside a class: see xcdBandOrder
                    return ( long ) a.nCand * a.w * ( a.h >> a.subShift ) + ( long ) a.winW * a.winH > ( long ) b.nCand * b.w * ( b.h >> b.subShift ) + ( long ) b.winW * b.winH; } );
  int intLarge = 0, intBig = 0, ldsIntSmall = 0, ldsIntMid = 0;
  for( const IntJob& j : ij ) { const int c = classOf( j ); if( c == 0 ) intLarge++; if( c <= 1 ) intBig++; if( c == 1 ) ldsIntMid = std::max( ldsIntMid, ldsOf( j ) ); if( c == 2 ) ldsIntSmall = std::max( ldsIntSmall, ldsOf( j ) ); }
  if( band && !ij.empty() )
  {
    // workgroups: one window each for classes 0 and 1, then four small windows each (meIntKernel).  Inside an XCD's eighth the heaviest jobs start first (a band's planes fit
    // the L2 whatever the order inside it, and a raster search's windows are far heavier than a diamond's: in plain picture order the launch ended on a tail of them — 30.7 -> 34.9 us)
    auto weightOf = []( const IntJob& j ) { return ( long ) j.nCand * j.w * ( j.h >> j.subShift ) + ( long ) j.winW * j.winH; };
    auto heavyFirstPerEighth = [&]( int begin, int n, int per ) { const int q = ( ( n + per - 1 ) / per + 7 ) / 8 * per;
      for( int x = 0; x < 8 && q; x++ ) { const int a = std::min( n, x * q ), b = std::min( n, ( x + 1 ) * q );
        std::stable_sort( ij.begin() + begin + a, ij.begin() + begin + b, [&]( const IntJob& u, const IntJob& v ) { return weightOf( u ) > weightOf( v ); } ); } };
    heavyFirstPerEighth( 0, intLarge, 1 );
    heavyFirstPerEighth( intLarge, intBig - intLarge, 1 );
    heavyFirstPerEighth( intBig, ( int ) ij.size() - intBig, 4 );
    std::vector<IntJob> src( ij );
    const std::vector<int> pl = xcdBandOrder( intLarge, 0 );
    for( int l = 0; l < intLarge; l++ ) ij[l] = src[pl[l]];
    const std::vector<int> pb = xcdBandOrder( intBig - intLarge, intLarge );
    for( int l = 0; l < intBig - intLarge; l++ ) ij[intLarge + l] = src[intLarge + pb[l]];
    const int nSmall = ( int ) src.size() - intBig, nGr = ( nSmall + 3 ) / 4;
    const std::vector<int> ps = xcdBandOrder( nGr, intBig );
    for( int l = 0, o = intBig; l < nGr; l++ ) for( int k = 0; k < 4 && ps[l] * 4 + k < nSmall; k++ ) ij[o++] = src[intBig + ps[l] * 4 + k];
    // (a short last group may land in the middle: the groups behind it then start up to three jobs early — any four consecutive jobs are a valid workgroup)
  }

  // ---- stage units: (stage, band of <= 32 rows, half of <= 64 columns); a wave takes a bundle of units of one unit width and tap support worth ~160 second-pass row groups.
  //      A stage of several units (h > 32 or w > 64) is shared by the two waves of ONE workgroup, half of its units each.
  std::vector<WaveSpan> stWaves;
  int ldsStage = 0;
  for( int i = 0; i < n_stage_jobs; i++ )
  {
    const vvhip_me_stage_job& s = stage_jobs[i];
    bool ok = shapeOk( s.width, s.height, 4, 4 ) && !( s.width == 4 && s.height == 4 ) && s.org_plane <= 15 && s.ref_plane <= 15 && ( s.i_frac == 1 || s.i_frac == 2 ) && s.filter_mode <= 2 &&
              ( s.func == VVHIP_DF_SAD || s.func == VVHIP_DF_HAD || s.func == VVHIP_DF_HAD_FAST ) && !( s.mask >> 9 );
    // every evaluated position within one sample of the block (in 1/16 sample: -16 .. 16): the kernel stages the band's rows K0 - 4 .. BH + K1 - 4 and nothing else, so a
    // vertical displacement beyond that would read another variant's rows or the tap tables (s_acMvRefineH / Q offsets are -1 .. 1; InterSearch.cpp:67-91)
    for( int k = 0; ok && k < 9; k++ )
      if( ( s.mask >> k ) & 1 )
      {
        static const int8_t rxT[9] = { 0, 0, 0, -1, 1, -1, 1, -1, 1 }, ryH[9] = { 0, -1, 1, 0, 0, -1, -1, 1, 1 }, ryQ[9] = { 0, -1, 1, -1, -1, 0, 0, 1, 1 };
        const int tx = ( rxT[k] + s.base_qx ) * s.i_frac * 4, ty = ( ( s.i_frac == 2 ? ryH[k] : ryQ[k] ) + s.base_qy ) * s.i_frac * 4;
        if( tx < -16 || tx > 16 || ty < -16 || ty > 16 ) ok = false;
      }
    if( !ok )
      return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_create: stage job %d (%dx%d, iFrac %d, mode %d, func %d, base %d,%d, mask %x)", i, s.width, s.height, s.i_frac, s.filter_mode, s.func, s.base_qx, s.base_qy, s.mask );
  }
  // $VVHIP_ME_ATOMIC_STAGES=0: stages of more than two units as two waves of four (two) units each, the first form of round 4
  static const bool atomicStages = !( getenv( "VVHIP_ME_ATOMIC_STAGES" ) && atoi( getenv( "VVHIP_ME_ATOMIC_STAGES" ) ) == 0 );
  auto tapSetOf = []( const vvhip_me_stage_job& s ) { return ( s.filter_mode == 2 && !s.alt_hpel ) ? 0 : ( s.filter_mode == 0 ? 2 : 1 ); };      // which tap support the bundle's kernel instance uses
  auto setOf = [&]( const vvhip_me_stage_job& s ) { const bool sq = s.width == s.height && s.width >= 8 && s.width <= 64; return 2 * tapSetOf( s ) + ( sq ? 0 : 1 ); };      // launch class
  auto unitW = []( const vvhip_me_stage_job& s ) { return std::min( ( int ) s.width, 64 ); };
  auto unitH = []( const vvhip_me_stage_job& s ) { return std::min( ( int ) s.height, 32 ); };
  auto unitsOf = [&]( const vvhip_me_stage_job& s ) { return ( s.width / unitW( s ) ) * ( s.height / unitH( s ) ); };
  // how a stage's units are dealt: 0 = shared by the two waves of one workgroup (two units; more when the atomic form is off), 1 = more than two units, every unit a wave of its
  // own adding into the cleared cost array, 2 = a single unit (bundled with others)
  auto dealOf = [&]( const vvhip_me_stage_job& s ) { const int n = unitsOf( s ); return n == 1 ? 2 : ( ( n > 2 && atomicStages ) ? 1 : 0 ); };
  auto unitWork = [&]( const vvhip_me_stage_job& s ) { return __builtin_popcount( s.mask ) * std::max( 1, unitW( s ) / 8 ) * unitH( s ); };      // 8-sample row groups of the second pass
  // A wide unit's horizontal variants as waves of their own — when the stage launch would not fill the chip anyway: below $VVHIP_ME_SPLIT_VARIANTS units in the picture
  // (default 8 192 ~ 1.3 resident rounds of stage waves; 0 = never, the form up to round 4; results identical).  Measured on the recorded lists (profiles/r05_stage_variants.log):
  // 1080p (4 554 -> 10 970 waves) stage launch 31.6 -> 29.0 us, GOP-weighted rate +4 %; 4K (~20 000 waves before the split) 71 -> 78 us — there the chip is full and the extra
  // waves only repeat the per-wave set-up.
  static const long splitBelow = getenv( "VVHIP_ME_SPLIT_VARIANTS" ) ? atol( getenv( "VVHIP_ME_SPLIT_VARIANTS" ) ) : 8192;
  long unitsUnsplit = 0;
  for( int i = 0; i < n_stage_jobs; i++ ) unitsUnsplit += unitsOf( stage_jobs[i] );
  const bool stageSplitVariants = unitsUnsplit < splitBelow;
  std::vector<int32_t> stOrder;
  bool hasAtomic = false;
  for( int i = 0; i < n_stage_jobs; i++ )      // (a stage without evaluated positions still gets its unit: the kernel writes its nine zeros)
  {
    const vvhip_me_stage_job& s = stage_jobs[i];
    const int bands = s.height / unitH( s ), halves = s.width / unitW( s ), n = bands * halves;
    // Units of 32 and 64 columns hold ONE horizontal variant in LDS at a time (stageBody): their <= 3 variants are independent pieces of work with disjoint positions — a wave
    // each (round 5).  A 64x32 unit with nine positions was ~3 300 instructions in one wave and the launch lasted as long as its slowest such wave (the launch's 4 554 waves do not
    // even fill the chip once); three waves of a third each, nothing computed twice.  (Narrow units keep their variants together: they share one first pass.)
    int nVar = 1;
    if( stageSplitVariants && unitW( s ) >= 32 && ( n <= 2 || atomicStages ) )
    {
      int hx[3], nh = 0;
      for( int k = 0; k < 9; k++ )
        if( ( s.mask >> k ) & 1 )
        {
          static const int8_t rxT[9] = { 0, 0, 0, -1, 1, -1, 1, -1, 1 };
          const int tx = ( rxT[k] + s.base_qx ) * s.i_frac * 4;
          int v = 0; while( v < nh && hx[v] != tx ) v++;
          if( v == nh ) hx[nh++] = tx;
        }
      nVar = nh > 1 ? nh : 1;
    }
    for( int vr = 0; vr < nVar; vr++ )
    {
      const int vbits = nVar > 1 ? ( vr + 1 ) << ST_VAR_SHIFT : 0;
      if( n > 2 && atomicStages ) { hasAtomic = true; for( int u = 0; u < n; u++ ) stOrder.push_back( i | vbits | ( ( u % bands ) << 24 ) | ( ( u / bands ) << 27 ) | ST_UNIT_ATOMIC ); continue; }
      const int perWave = n > 1 ? n / 2 : 1;
      for( int u = 0; u < n; u++ )
        stOrder.push_back( i | vbits | ( ( u % bands ) << 24 ) | ( ( u / bands ) << 27 ) | ( ( u % perWave ) ? ST_UNIT_CONT : 0 ) | ( ( u % perWave ) != perWave - 1 ? ST_UNIT_MORE : 0 ) );
    }
  }
  // per tap support: the stages of several units first (their waves must be the pairs 2g, 2g + 1 of the launch), then by unit width and work
  std::stable_sort( stOrder.begin(), stOrder.end(), [&]( int a, int b ) { const auto& x = stage_jobs[a & ST_STAGE_MASK]; const auto& y = stage_jobs[b & ST_STAGE_MASK];
                    const int px = dealOf( x ), py = dealOf( y );
                    if( setOf( x ) != setOf( y ) ) return setOf( x ) < setOf( y );
                    if( px != py ) return px < py;
                    if( unitW( x ) != unitW( y ) ) return unitW( x ) > unitW( y );
                    return band ? x.ref_off < y.ref_off : unitWork( x ) > unitWork( y ); } );      // picture order inside a sub-class (xcdBandOrder) / heaviest first
  int setWaves[6] = { 0, 0, 0, 0, 0, 0 }, setLds[6] = { 0, 0, 0, 0, 0, 0 };
  static const int bundleWork = getenv( "VVHIP_ME_BUNDLE_WORK" ) ? atoi( getenv( "VVHIP_ME_BUNDLE_WORK" ) ) : 80;      // recorded 1080p lists, round 3: 80 / 160 / 320 / 640 / 1280 -> 59.0 / 58.7 / 61.9 / 67.9 / 71.1 us; round 4 (XCD-band order): 80 / 120 / 160 / 240 / 320 -> 41.0 / 44.2 / 44.3 / 44.5 / 43.3 us
  for( size_t i = 0; i < stOrder.size(); )
  {
    const vvhip_me_stage_job& s0 = stage_jobs[stOrder[i] & ST_STAGE_MASK];
    int count = 0, work = 0;
    if( dealOf( s0 ) == 0 ) count = unitsOf( s0 ) / 2;      // half of a shared stage's units: a wave of its own, next to its sibling
    else if( dealOf( s0 ) == 1 ) count = 1;                  // a unit of an atomic stage
    else
      while( i + count < stOrder.size() && count < 8 )
      {
        const vvhip_me_stage_job& s = stage_jobs[stOrder[i + count] & ST_STAGE_MASK];
        if( unitW( s ) != unitW( s0 ) || setOf( s ) != setOf( s0 ) || dealOf( s ) != 2 || ( count && work + unitWork( s ) > bundleWork ) ) break;
        work += unitWork( s ); count++;
      }
    WaveSpan sp; sp.first = ( int32_t ) i; sp.count = count; stWaves.push_back( sp );
    setWaves[setOf( s0 )]++;
    // the wave's LDS slice must hold the TALLEST unit of the bundle: bundles group by unit width, launch class and deal, not by height — in the rectangular classes a short
    // full-mask leader (8x4, nine positions) can be followed by a tall unit with few evaluated positions (8x32, one position) inside the same work budget
    int bh = unitH( s0 );
    for( int u = 1; u < count; u++ ) bh = std::max( bh, unitH( stage_jobs[stOrder[i + u] & ST_STAGE_MASK] ) );
    const int nt = tapSetOf( s0 ) == 0 ? 4 : ( tapSetOf( s0 ) == 1 ? 6 : 8 ), uw = std::max( 8, unitW( s0 ) ), vpp = unitW( s0 ) <= 16 ? 3 : 1;
    const int ldsUnit = ( 2 * ( 128 + 64 + 16 + 16 ) + vpp * ( bh + nt ) * ( uw + 8 ) ) * 2;      // tables + the first-pass bands a pass holds (row pitch unit width + 8)
    setLds[setOf( s0 )] = std::max( setLds[setOf( s0 )], ( ldsUnit + 15 ) & ~15 );
    ldsStage = std::max( ldsStage, ldsUnit );
    i += count;
  }

  if( band )
  {
    // per launch class and sub-class (shared stages / unit width): the workgroups (wave pairs) in XCD-band order
    std::vector<WaveSpan> src( stWaves );
    for( int k = 0, first = 0; k < 6; first += setWaves[k], k++ )
    {
      auto subOf = [&]( int w ) { const vvhip_me_stage_job& s = stage_jobs[stOrder[src[first + w].first] & ST_STAGE_MASK]; return ( 2 - dealOf( s ) ) * 1024 + unitW( s ); };
      for( int w0 = 0; w0 < setWaves[k]; )
      {
        int w1 = w0;
        while( w1 < setWaves[k] && subOf( w1 ) == subOf( w0 ) ) w1++;
        {
          const int a0 = ( w0 + 1 ) & ~1, nGr = ( w1 - a0 ) / 2;      // whole workgroups of the sub-class (one that starts on an odd wave shares its first workgroup with the previous sub-class; shared stages come first and are pairs)
          const std::vector<int> pm = xcdBandOrder( std::max( nGr, 0 ), a0 / 2 );
          for( int l = 0; l < nGr; l++ ) { stWaves[first + a0 + 2 * l] = src[first + a0 + 2 * pm[l]]; stWaves[first + a0 + 2 * l + 1] = src[first + a0 + 2 * pm[l] + 1]; }
        }
        w0 = w1;
      }
    }
  }
  // the kernel relies on it: the two halves of a shared stage's units are the waves 2g, 2g + 1 of their tap support's launch
  for( int k = 0, first = 0; k < 6; first += setWaves[k], k++ )
    for( int w = 0; w < setWaves[k]; w++ )
    {
      const WaveSpan& sp = stWaves[first + w];
      const int u = stOrder[sp.first];
      const vvhip_me_stage_job& s = stage_jobs[u & ST_STAGE_MASK];
      if( dealOf( s ) != 0 ) continue;
      const int sib = ( w & 1 ) ? w - 1 : w + 1;
      if( sp.count != unitsOf( s ) / 2 || sib >= setWaves[k] || stWaves[first + sib].count != sp.count || ( stOrder[stWaves[first + sib].first] & ( ST_STAGE_MASK | ( 3 << ST_VAR_SHIFT ) ) ) != ( u & ( ST_STAGE_MASK | ( 3 << ST_VAR_SHIFT ) ) ) )
        return vvhip_fail( ctx, VVHIP_E_HIP, "vvhip_me_plan_create: schedule error (the units of stage %d are not one workgroup)", u & ST_STAGE_MASK );
    }

  // ---- item bundles: same function and geometry, a few team passes per wave
  std::vector<int32_t> itOrder( ( size_t ) n_items + n_mask ); std::vector<WaveSpan> itWaves;
  for( int i = 0; i < n_items; i++ )
  {
    const vvhip_me_item& s = items[i];
    const bool fOk = s.func == VVHIP_DF_SAD || s.func == VVHIP_DF_SSE || s.func == VVHIP_DF_HAD || s.func == VVHIP_DF_HAD_FAST || s.func == VVHIP_DF_HAD_2SAD;
    if( !shapeOk( s.width, s.height, 2, 2 ) || !fOk || s.org_plane > 15 || s.cur_plane > 15 || s.sub_shift > 1 || ( s.sub_shift && ( s.func != VVHIP_DF_SAD || s.height < 2 ) ) )
      return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_create: item %d (func %d, %dx%d)", i, s.func, s.width, s.height );
    itOrder[i] = i;
  }
  // the generic body's items (rectangular Hadamard tiles, 2x2 tiles, two-sample-wide blocks) behind the others: their waves are a launch of their own
  auto itemGen = [&]( int i ) { const auto& s = items[i]; if( s.func == VVHIP_DF_SAD || s.func == VVHIP_DF_SSE ) return s.width < 4;
                                const int kind = hadTileKind( s.width, s.height, s.func == VVHIP_DF_HAD_FAST ); return !( kind == TK_8x8 || kind == TK_16F || ( kind == TK_4x4 && s.width == 4 && s.height == 4 ) ); };
  auto itemKey = [&]( int i ) { const auto& s = items[i]; return ( itemGen( i ) ? 0l : 1l << 48 ) | ( ( long ) s.width << 32 ) | ( ( long ) s.height << 16 ) | ( s.func << 8 ) | s.sub_shift; };
  std::stable_sort( itOrder.begin(), itOrder.begin() + n_items, [&]( int a, int b ) { return itemKey( a ) > itemKey( b ); } );
  int wavesItemMain = 0;
  for( int i = 0; i < n_items; )
  {
    const vvhip_me_item& s0 = items[itOrder[i]];
    int lanesPer;
    if( s0.func == VVHIP_DF_SAD || s0.func == VVHIP_DF_SSE ) { const int cw = s0.width >= 8 ? 8 : ( s0.width >= 4 ? 4 : 2 ); lanesPer = ( s0.height >> ( s0.func == VVHIP_DF_SAD ? s0.sub_shift : 0 ) ) * ( s0.width / cw ); }
    else { const int kind = hadTileKind( s0.width, s0.height, s0.func == VVHIP_DF_HAD_FAST ); lanesPer = kind == TK_4x4 ? 1 : ( s0.width / tileW( kind ) ) * ( s0.height / tileH( kind ) ) * tileLanes( kind ); }
    if( lanesPer > 64 ) lanesPer = 64;
    const int perWave = std::max( 1, 64 / lanesPer );                // one pass of lane teams per wave: the items are latency-bound, waves are what overlaps them
    int count = 0;
    while( i + count < n_items && count < perWave && itemKey( itOrder[i + count] ) == itemKey( itOrder[i] ) ) count++;
    WaveSpan sp; sp.first = i; sp.count = count; itWaves.push_back( sp );
    if( !itemGen( itOrder[i] ) ) wavesItemMain = ( int ) itWaves.size();
    i += count;
  }
  // Round 6: BAND-MAJOR interleave of the classes.  A call's original block is read by every class that scores it (SAD, SSE, Hadamard lists of the same CU);
  // with one class after the other an XCD streams its eighth of class A through its 4 MB L2 before class B comes back to the same originals: at 4K the table calls moved
  // 354 MB for 290 MB of unique bytes.  Here XCD x (workgroup index mod 8) walks its eighth of the PICTURE in sub-bands and runs every class's waves of a sub-band before the
  // next sub-band.  Picture position of a wave = the index its first item had in the caller's list (the lists arrive in picture order).  MEASURED SLOWER (same box, A/B twice:
  // 4K 81.2 -> 84.0 us per picture, 1080p 21.8 -> 25.6 us; 4 / 16 / 64 sub-bands alike): waves of different functions side by side cost more than the originals' second fetch
  // from the Infinity Cache — so it is OFF; $VVHIP_ME_ITEM_INTERLEAVE=<sub-bands per eighth> switches it on for measurements (results identical).
  static const int interleaveSub = []{ const char* e = getenv( "VVHIP_ME_ITEM_INTERLEAVE" ); return e ? atoi( e ) : 0; }();
  if( band && interleaveSub > 0 && wavesItemMain > 64 && wavesItemMain <= 65536 && n_items > 0 )
  {
    const std::vector<WaveSpan> src( itWaves.begin(), itWaves.begin() + wavesItemMain );
    const int nb = 8 * interleaveSub;
    std::vector<int> idx( wavesItemMain ), bandOf( wavesItemMain );
    for( int w = 0; w < wavesItemMain; w++ ) { idx[w] = w; bandOf[w] = ( int ) std::min<long>( nb - 1, ( long ) itOrder[src[w].first] * nb / n_items ); }
    std::stable_sort( idx.begin(), idx.end(), [&]( int a, int b ) { return bandOf[a] < bandOf[b]; } );      // (stable: inside a sub-band the classes keep their order, heaviest first)
    std::vector<int> qBeg( 9, wavesItemMain );
    for( int k = wavesItemMain - 1; k >= 0; k-- ) qBeg[bandOf[idx[k]] / interleaveSub] = k;
    for( int x = 7; x >= 0; x-- ) if( qBeg[x] > qBeg[x + 1] ) qBeg[x] = qBeg[x + 1];                          // (an XCD without waves)
    int cursor[8];
    for( int x = 0; x < 8; x++ ) cursor[x] = qBeg[x];
    int outW = 0;
    for( int l = 0; outW < wavesItemMain; l++ )
    {
      int x = l & 7;
      for( int t = 0; t < 8 && cursor[x] >= qBeg[x + 1]; t++ ) x = ( x + 1 ) & 7;                           // (a finished eighth borrows from its neighbour)
      for( int k = 0; k < 4 && outW < wavesItemMain; k++ )
      {
        if( cursor[x] >= qBeg[x + 1] ) { int t = 0; for( ; t < 8 && cursor[x] >= qBeg[x + 1]; t++ ) x = ( x + 1 ) & 7; if( t == 8 ) break; }
        itWaves[outW++] = src[idx[cursor[x]++]];
      }
    }
  }
  else
  if( band && wavesItemMain <= 65536 )      // (four-wave workgroups; the long lists of an intra picture run one wave per workgroup in list order)
  {
    std::vector<WaveSpan> src( itWaves );
    auto keyOfWave = [&]( int w ) { return itemKey( itOrder[src[w].first] ); };
    for( int w0 = 0; w0 < wavesItemMain; )
    {
      int w1 = w0;
      while( w1 < wavesItemMain && keyOfWave( w1 ) == keyOfWave( w0 ) ) w1++;
      const int a0 = ( w0 + 3 ) & ~3, nGr = ( w1 - a0 ) / 4;      // whole workgroups inside the class
      if( nGr > 8 )
      {
        const std::vector<int> pm = xcdBandOrder( nGr, a0 / 4 );
        for( int l = 0; l < nGr; l++ ) for( int k = 0; k < 4; k++ ) itWaves[a0 + 4 * l + k] = src[a0 + 4 * pm[l] + k];
      }
      w0 = w1;
    }
  }
  // masked items: their waves follow (count < 0 marks them), their schedule entries and costs sit behind the plain items'
  for( int i = 0; i < n_mask; i++ )
  {
    const vvhip_me_mask_item& s = mask_items[i];
    if( !shapeOk( s.width, s.height, 2, 2 ) || s.org_plane > 15 || s.cur_plane > 15 || s.mask_plane > 15 || s.sub_shift > 1 || ( s.sub_shift && s.height < 2 ) )
      return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_create: masked item %d (%dx%d)", i, s.width, s.height );
    itOrder[n_items + i] = i;
  }
  auto maskKey = [&]( int i ) { const auto& s = mask_items[i]; return ( ( long ) s.width << 32 ) | ( ( long ) s.height << 16 ) | s.sub_shift; };
  std::stable_sort( itOrder.begin() + n_items, itOrder.end(), [&]( int a, int b ) { return maskKey( a ) > maskKey( b ); } );
  for( int i = 0; i < n_mask; )
  {
    const vvhip_me_mask_item& s0 = mask_items[itOrder[n_items + i]];
    const int cw = s0.width >= 8 ? 8 : ( s0.width >= 4 ? 4 : 2 );
    const int lanesPer = std::min( 64, ( s0.height >> s0.sub_shift ) * ( s0.width / cw ) ), perWave = std::max( 1, 64 / lanesPer );
    int count = 0;
    while( i + count < n_mask && count < perWave && maskKey( itOrder[n_items + i + count] ) == maskKey( itOrder[n_items + i] ) ) count++;
    WaveSpan sp; sp.first = i; sp.count = -count; itWaves.push_back( sp );
    i += count;
  }

  int maxPlane = 0;                                                   // the run's plane table must cover every index the lists use
  for( int i = 0; i < n_int_jobs; i++ ) maxPlane = std::max( maxPlane, ( int ) std::max( int_jobs[i].org_plane, int_jobs[i].ref_plane ) );
  for( int i = 0; i < n_stage_jobs; i++ ) maxPlane = std::max( maxPlane, ( int ) std::max( stage_jobs[i].org_plane, stage_jobs[i].ref_plane ) );
  for( int i = 0; i < n_items; i++ ) maxPlane = std::max( maxPlane, ( int ) std::max( items[i].org_plane, items[i].cur_plane ) );
  for( int i = 0; i < n_mask; i++ ) maxPlane = std::max( maxPlane, ( int ) std::max( mask_items[i].mask_plane, std::max( mask_items[i].org_plane, mask_items[i].cur_plane ) ) );
  // ---- the tables go to the device in SCHEDULE order (a wave reads its jobs at the schedule index: no order -> record indirection on the critical path of a short-lived wave;
  //      the order arrays only say where a result goes)
  std::vector<StageUnit> stUnits( stOrder.size() );
  for( size_t i = 0; i < stOrder.size(); i++ )
  {
    StageUnit& u = stUnits[i];
    memset( &u, 0, sizeof( u ) );
    u.j = stage_jobs[stOrder[i] & ST_STAGE_MASK]; u.order = stOrder[i];
    static const int8_t rxT[9] = { 0, 0, 0, -1, 1, -1, 1, -1, 1 }, ryH[9] = { 0, -1, 1, 0, 0, -1, -1, 1, 1 }, ryQ[9] = { 0, -1, 1, -1, -1, 0, 0, 1, 1 };      // s_acMvRefineH / Q, InterSearch.cpp:67-91
    int tx[9], ty[9], var[9], cnt[3] = { 0, 0, 0 }, nHor = 0, nPos = 0;
    for( int k = 0; k < 9; k++ )
    {
      if( !( ( u.j.mask >> k ) & 1 ) ) continue;
      tx[k] = ( rxT[k] + u.j.base_qx ) * u.j.i_frac * 4; ty[k] = ( ( u.j.i_frac == 2 ? ryH[k] : ryQ[k] ) + u.j.base_qy ) * u.j.i_frac * 4;
      int v = 0; while( v < nHor && u.hx[v] != tx[k] ) v++;
      if( v == nHor ) u.hx[nHor++] = ( int16_t ) tx[k];
      var[k] = v; cnt[v]++; nPos++;
    }
    u.nHor = ( uint8_t ) nHor; u.nPos = ( uint8_t ) nPos; u.cnt0 = ( uint8_t ) cnt[0]; u.cnt1 = ( uint8_t ) cnt[1]; u.cnt2 = ( uint8_t ) cnt[2];
    int slot[3] = { 0, cnt[0], cnt[0] + cnt[1] };
    for( int k = 0; k < 9; k++ ) if( ( u.j.mask >> k ) & 1 ) u.pos[slot[var[k]]++] = k | ( ( tx[k] + 64 ) << 8 ) | ( ( ty[k] + 64 ) << 20 );
  }
  std::vector<vvhip_me_item> itSorted( n_items );
  for( int i = 0; i < n_items; i++ ) itSorted[i] = items[itOrder[i]];
  std::vector<vvhip_me_mask_item> mkSorted( n_mask );
  for( int i = 0; i < n_mask; i++ ) { mkSorted[i] = mask_items[itOrder[n_items + i]]; itOrder[n_items + i] += n_items; }      // (the order entry = where the cost goes)
  // ---- the interpolation tap tables the stage kernels stage into LDS (stageTapTables)
  const std::vector<int32_t> tapTab = stageTapTables( bit_depth );
  // ---- one device allocation for every table
  auto pad = []( size_t b ) { return ( b + 255 ) & ~( size_t ) 255; };
  const size_t bInt = pad( ij.size() * sizeof( IntJob ) ), bCand = pad( pc.size() * sizeof( PlanCand ) ), bCandO = pad( candOut.size() * 4 ), bSt = pad( stUnits.size() * sizeof( StageUnit ) ),
               bStO = pad( stOrder.size() * 4 ), bStW = pad( stWaves.size() * sizeof( WaveSpan ) ), bIt = pad( ( size_t ) n_items * sizeof( vvhip_me_item ) ), bItO = pad( itOrder.size() * 4 ),
               bItW = pad( itWaves.size() * sizeof( WaveSpan ) ), bMk = pad( mkSorted.size() * sizeof( vvhip_me_mask_item ) );
  const size_t bTap = pad( tapTab.size() * 4 );
  const size_t total = bInt + bCand + bCandO + bSt + bStO + bStW + bIt + bItO + bItW + bTap + bMk + 256;
  std::vector<char> host( total, 0 );
  size_t o = 0;
  auto put = [&]( const void* src, size_t bytes, size_t padded ) { const size_t at = o; if( bytes ) memcpy( host.data() + o, src, bytes ); o += padded; return at; };
  const size_t oInt = put( ij.data(), ij.size() * sizeof( IntJob ), bInt ), oCand = put( pc.data(), pc.size() * sizeof( PlanCand ), bCand ), oCandO = put( candOut.data(), candOut.size() * 4, bCandO ),
               oSt = put( stUnits.data(), stUnits.size() * sizeof( StageUnit ), bSt ), oStO = put( stOrder.data(), stOrder.size() * 4, bStO ),
               oStW = put( stWaves.data(), stWaves.size() * sizeof( WaveSpan ), bStW ), oIt = put( itSorted.data(), ( size_t ) n_items * sizeof( vvhip_me_item ), bIt ),
               oItO = put( itOrder.data(), itOrder.size() * 4, bItO ), oItW = put( itWaves.data(), itWaves.size() * sizeof( WaveSpan ), bItW ), oTap = put( tapTab.data(), tapTab.size() * 4, bTap ),
               oMk = put( mkSorted.data(), mkSorted.size() * sizeof( vvhip_me_mask_item ), bMk );
  vvhip_me_plan* p = new vvhip_me_plan;
  hipError_t e = hipMalloc( &p->d_blob, total );
  if( e != hipSuccess ) { delete p; return vvhip_fail( ctx, VVHIP_E_NOMEM, "vvhip_me_plan_create: hipMalloc( %zu ): %s", total, hipGetErrorString( e ) ); }
  e = hipMemcpyAsync( p->d_blob, host.data(), total, hipMemcpyHostToDevice, ctx->stream );
  if( e == hipSuccess ) e = hipStreamSynchronize( ctx->stream );
  if( e != hipSuccess ) { ( void ) hipFree( p->d_blob ); delete p; return vvhip_fail( ctx, VVHIP_E_HIP, "vvhip_me_plan_create: upload: %s", hipGetErrorString( e ) ); }
  char* b = static_cast<char*>( p->d_blob );
  p->d_intJobs = b + oInt; p->d_cands = b + oCand; p->d_candOut = b + oCandO; p->d_stageJobs = b + oSt; p->d_stageOrder = b + oStO; p->d_stageWaves = b + oStW; p->d_items = b + oIt; p->d_itemOrder = b + oItO; p->d_itemWaves = b + oItW; p->d_tapTables = b + oTap; p->d_maskItems = b + oMk;
  p->bitDepth = bit_depth; p->nCands = n_cands; p->nStages = n_stage_jobs; p->nItems = n_items; p->nMaskItems = n_mask; p->maxPlane = maxPlane; p->stageAtomic = hasAtomic;
  p->wavesInt = ( int ) ij.size(); p->wavesStage = ( int ) stWaves.size(); p->wavesItem = ( int ) itWaves.size(); p->wavesItemMain = wavesItemMain;
  p->ldsInt = ( ldsInt + 15 ) & ~15; p->ldsStage = ( ldsStage + 15 ) & ~15;
  for( int k = 0; k < 6; k++ ) { p->stageSetWaves[k] = setWaves[k]; p->stageSetLds[k] = setLds[k]; }
  p->intBig = intBig; p->ldsIntSmall = ( ldsIntSmall + 15 ) & ~15; p->intLarge = intLarge; p->ldsIntMid = ( ldsIntMid + 15 ) & ~15;
  // 128-wide blocks make the large windows' LDS several times what four small windows need: their own launch then, so that the small windows keep their occupancy
  p->intSplit = intBig > 0 && intBig < ( int ) ij.size() && p->ldsInt > 2 * 4 * p->ldsIntSmall && p->ldsInt > 32 * 1024;
  if( p->ldsInt > 160 * 1024 || p->ldsStage > 64 * 1024 )
  { ( void ) hipFree( p->d_blob ); delete p; return vvhip_fail( ctx, VVHIP_E_UNSUPPORTED, "vvhip_me_plan_create: %d / %d bytes of LDS per wave (max_window too large?)", ldsInt, ldsStage ); }
  *out = p;
  return VVHIP_OK;
}

void vvhip_me_plan_destroy( vvhip_ctx* ctx, vvhip_me_plan* plan )
{
  if( !plan ) return;
  if( ctx ) ( void ) hipStreamSynchronize( ctx->stream );
  if( plan->d_blob ) ( void ) hipFree( plan->d_blob );
  for( hipEvent_t e : plan->ev ) if( e ) ( void ) hipEventDestroy( e );
  delete plan;
}

int vvhip_me_plan_info( const vvhip_me_plan* plan, int* waves_int, int* waves_stage, int* waves_item, int* lds_bytes )
{
  if( !plan ) return VVHIP_E_ARG;
  if( waves_int ) *waves_int = plan->wavesInt;
  if( waves_stage ) *waves_stage = plan->wavesStage;
  if( waves_item ) *waves_item = plan->wavesItem;
  if( lds_bytes ) *lds_bytes = plan->ldsInt > plan->ldsStage ? plan->ldsInt : plan->ldsStage;
  return VVHIP_OK;
}

int vvhip_me_plan_set_timing( vvhip_ctx* ctx, vvhip_me_plan* plan, int on )
{
  if( !ctx || !plan ) return VVHIP_E_ARG;
  if( on ) for( hipEvent_t& e : plan->ev ) if( !e ) VVHIP_CHECK_HIP( ctx, hipEventCreate( &e ) );
  plan->timing = on != 0;
  return VVHIP_OK;
}

int vvhip_me_plan_last_times( vvhip_ctx* ctx, const vvhip_me_plan* plan, float* ms4 )
{
  if( !ctx || !plan || !ms4 || !plan->timing ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipEventSynchronize( plan->ev[4] ) );
  for( int k = 0; k < 4; k++ ) VVHIP_CHECK_HIP( ctx, hipEventElapsedTime( &ms4[k], plan->ev[k], plan->ev[k + 1] ) );
  return VVHIP_OK;
}

static int mePlanRun( vvhip_ctx* ctx, const vvhip_me_plan* plan, const vvhip_me_plane* planes_host, int n_planes, uint64_t* d_cand_cost, uint64_t* d_stage_cost, uint64_t* d_item_cost, int parts );

int vvhip_me_plan_run( vvhip_ctx* ctx, const vvhip_me_plan* plan, const vvhip_me_plane* planes_host, int n_planes, uint64_t* d_cand_cost, uint64_t* d_stage_cost, uint64_t* d_item_cost )
{
  return mePlanRun( ctx, plan, planes_host, n_planes, d_cand_cost, d_stage_cost, d_item_cost, 7 );
}

int vvhip_me_plan_run_parts( vvhip_ctx* ctx, const vvhip_me_plan* plan, const vvhip_me_plane* planes_host, int n_planes, uint64_t* d_cand_cost, uint64_t* d_stage_cost, uint64_t* d_item_cost, int parts )
{
  return mePlanRun( ctx, plan, planes_host, n_planes, d_cand_cost, d_stage_cost, d_item_cost, parts );
}

static int mePlanRun( vvhip_ctx* ctx, const vvhip_me_plan* plan, const vvhip_me_plane* planes_host, int n_planes, uint64_t* d_cand_cost, uint64_t* d_stage_cost, uint64_t* d_item_cost, int parts )
{
  if( !ctx || !plan ) return VVHIP_E_ARG;
  if( !planes_host || n_planes < 1 || n_planes > 16 || ( plan->nCands && !d_cand_cost ) || ( plan->nStages && !d_stage_cost ) || ( ( plan->nItems || plan->nMaskItems ) && !d_item_cost ) )
    return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_run: bad arguments" );
  if( plan->maxPlane >= n_planes ) return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_me_plan_run: the plan's lists use plane %d, the table has %d", plan->maxPlane, n_planes );
  MePlanes P;
  for( int i = 0; i < 16; i++ ) { P.p[i] = planes_host[i < n_planes ? i : 0].d_base; P.stride[i] = planes_host[i < n_planes ? i : 0].stride; }
  MeArgs a;
  a.intJobs = static_cast<const IntJob*>( plan->d_intJobs ); a.cands = static_cast<const PlanCand*>( plan->d_cands ); a.candOut = static_cast<const int32_t*>( plan->d_candOut ); a.wavesInt = plan->wavesInt;
  a.stageUnits = static_cast<const StageUnit*>( plan->d_stageJobs );
  a.stageWaves = static_cast<const WaveSpan*>( plan->d_stageWaves ); a.wavesStage = plan->wavesStage; a.tapTables = static_cast<const int32_t*>( plan->d_tapTables );
  a.items = static_cast<const vvhip_me_item*>( plan->d_items ); a.itemOrder = static_cast<const int32_t*>( plan->d_itemOrder ); a.itemWaves = static_cast<const WaveSpan*>( plan->d_itemWaves ); a.wavesItem = plan->wavesItem;
  a.maskItems = static_cast<const vvhip_me_mask_item*>( plan->d_maskItems );
  a.candCost = d_cand_cost; a.stageCost = d_stage_cost; a.itemCost = d_item_cost; a.bitDepth = plan->bitDepth;
  const bool tm = plan->timing && parts == 7;
  const bool doStage = ( parts & 1 ) != 0, doInt = ( parts & 2 ) != 0, doItem = ( parts & 4 ) != 0;
  if( tm ) VVHIP_CHECK_HIP( ctx, hipEventRecord( plan->ev[0], ctx->stream ) );
  if( plan->stageAtomic && doStage ) VVHIP_CHECK_HIP( ctx, hipMemsetAsync( d_stage_cost, 0, ( size_t ) 9 * 8 * plan->nStages, ctx->stream ) );
  int firstWave = 0;
  static const int ldsPadExp = getenv( "VVHIP_ME_LDS_PAD" ) ? atoi( getenv( "VVHIP_ME_LDS_PAD" ) ) : 0;      // experiment: occupancy sensitivity of the stage kernel
  constexpr int stW = 2;                                                                // waves per workgroup (see meStageKernel)
  // one launch per class (tap support x fast-preset shapes / any shape), each with the LDS its own units need; inside a class: shared stages first, then by unit width
  // (splitting the wide blocks' bundles from the small blocks' or giving them four-wave workgroups was measured slower)
#define ME_STAGE_LAUNCH( CLS, K0, K1, GEN ) \
  if( plan->stageSetWaves[CLS] && doStage ) { const size_t ldsSt = ( ( size_t ) plan->stageSetLds[CLS] + ldsPadExp + 15 ) & ~( size_t ) 15; \
    hipLaunchKernelGGL( ( meStageKernel<K0, K1, GEN> ), dim3( ( unsigned ) ( ( plan->stageSetWaves[CLS] + stW - 1 ) / stW ) ), dim3( 64 * stW ), ( size_t ) stW * ldsSt, ctx->stream, P, a, firstWave, plan->stageSetWaves[CLS], ( int ) ldsSt ); } \
  firstWave += plan->stageSetWaves[CLS];
  ME_STAGE_LAUNCH( 0, 2, 5, false ) ME_STAGE_LAUNCH( 1, 2, 5, true )
  ME_STAGE_LAUNCH( 2, 1, 6, false ) ME_STAGE_LAUNCH( 3, 1, 6, true )
  ME_STAGE_LAUNCH( 4, 0, 7, false ) ME_STAGE_LAUNCH( 5, 0, 7, true )
#undef ME_STAGE_LAUNCH
  if( tm ) VVHIP_CHECK_HIP( ctx, hipEventRecord( plan->ev[1], ctx->stream ) );
  if( plan->wavesInt && doInt )
  {
    // one launch for both window classes (two launches of a few thousand short-lived waves each were mostly ramp-up and drain: 20.6 + 18.0 us on a recorded 1080p picture)
    const int nSmall = plan->wavesInt - plan->intBig, lds = std::max( plan->intBig ? plan->ldsInt : 0, nSmall ? 4 * plan->ldsIntSmall : 0 );
    if( plan->ldsInt > 64 * 1024 ) VVHIP_CHECK_HIP( ctx, hipFuncSetAttribute( ( const void* ) meIntKernel, hipFuncAttributeMaxDynamicSharedMemorySize, plan->ldsInt ) );
    const int nLarge = plan->intLarge, nMid = plan->intBig - nLarge;
    // one launch per LDS class where the classes differ much (each class keeps the occupancy its own windows allow), one launch for everything otherwise
    const bool splitLarge = nLarge > 0 && nMid > 0 && plan->ldsInt > plan->ldsIntMid + plan->ldsIntMid / 2;
    const bool splitSmall = plan->intSplit;
    if( !splitLarge && !splitSmall ) hipLaunchKernelGGL( meIntKernel, dim3( ( unsigned ) ( plan->intBig + ( nSmall + 3 ) / 4 ) ), dim3( 256 ), ( size_t ) lds, ctx->stream, P, a, plan->intBig, plan->ldsIntSmall, 0 );
    else
    {
      if( splitLarge )
      {
        hipLaunchKernelGGL( meIntKernel, dim3( ( unsigned ) nLarge ), dim3( 256 ), ( size_t ) plan->ldsInt, ctx->stream, P, a, plan->intBig, plan->ldsIntSmall, 0 );
        const int rest = nMid + ( splitSmall ? 0 : ( nSmall + 3 ) / 4 ), ldsRest = std::max( plan->ldsIntMid, splitSmall ? 0 : 4 * plan->ldsIntSmall );
        hipLaunchKernelGGL( meIntKernel, dim3( ( unsigned ) rest ), dim3( 256 ), ( size_t ) ldsRest, ctx->stream, P, a, plan->intBig, plan->ldsIntSmall, nLarge );
      }
      else hipLaunchKernelGGL( meIntKernel, dim3( ( unsigned ) plan->intBig ), dim3( 256 ), ( size_t ) plan->ldsInt, ctx->stream, P, a, plan->intBig, plan->ldsIntSmall, 0 );
      if( splitSmall && nSmall ) hipLaunchKernelGGL( meIntKernel, dim3( ( unsigned ) ( ( nSmall + 3 ) / 4 ) ), dim3( 256 ), ( size_t ) 4 * plan->ldsIntSmall, ctx->stream, P, a, plan->intBig, plan->ldsIntSmall, plan->intBig );
    }
  }
  if( tm ) VVHIP_CHECK_HIP( ctx, hipEventRecord( plan->ev[2], ctx->stream ) );
  if( tm ) VVHIP_CHECK_HIP( ctx, hipEventRecord( plan->ev[3], ctx->stream ) );
  if( plan->wavesItem && doItem )
  {
    const int nMain = plan->wavesItemMain, nGen = plan->wavesItem - nMain;
    static const int longSpans = getenv( "VVHIP_ME_ITEM_SPANS" ) ? atoi( getenv( "VVHIP_ME_ITEM_SPANS" ) ) : 1;
    auto launch = [&]( bool gen, int first, int n )
    {
      if( n <= 65536 )
      {
        if( gen ) hipLaunchKernelGGL( ( meItemKernel<4, true> ), dim3( ( unsigned ) ( ( n + 3 ) / 4 ) ), dim3( 256 ), 0, ctx->stream, P, a, plan->nItems, first, n, 1 );
        else      hipLaunchKernelGGL( ( meItemKernel<4, false> ), dim3( ( unsigned ) ( ( n + 3 ) / 4 ) ), dim3( 256 ), 0, ctx->stream, P, a, plan->nItems, first, n, 1 );
      }
      else
      {
        const int spw = longSpans < 1 ? 1 : longSpans, nw = ( n + spw - 1 ) / spw;
        if( gen ) hipLaunchKernelGGL( ( meItemKernel<1, true> ), dim3( ( unsigned ) nw ), dim3( 64 ), 0, ctx->stream, P, a, plan->nItems, first, n, spw );
        else      hipLaunchKernelGGL( ( meItemKernel<1, false> ), dim3( ( unsigned ) nw ), dim3( 64 ), 0, ctx->stream, P, a, plan->nItems, first, n, spw );
      }
    };
    if( nMain ) launch( false, 0, nMain );
    if( nGen ) launch( true, nMain, nGen );
  }
  VVHIP_LAUNCH_CHECK( ctx );
  if( tm ) VVHIP_CHECK_HIP( ctx, hipEventRecord( plan->ev[4], ctx->stream ) );
  return VVHIP_OK;
}

} // extern "C"
#endif
