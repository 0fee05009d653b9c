// common.h — internal declarations shared by the HIP translation units of libvvenc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/vvenc_hip.h"

struct vvhip_ctx
{
  int          device     = 0;
  hipStream_t  stream     = nullptr;   // stream every launch goes to
  hipStream_t  ownStream  = nullptr;
  std::string  lastError;

  // ROM in HBM (uploaded once by vvhip_create)
  int16_t*     d_trMat    = nullptr;   // all transform matrices, see trMatOffset()
  uint16_t*    d_scan     = nullptr;   // grouped diagonal scan orders, see scanOffset()
  struct VvhipTuMxOps* d_tuMx = nullptr;   // matrix-core operand records of the fused TU kernel: [type][size 4,8,16,32], see VvhipTuMxOps
  struct VvhipTuMx64Ops* d_tuMx64 = nullptr;   // the 64-point DCT-2 record
  uint16_t*    d_tuMxPos  = nullptr;   // scan position of every (lane, register) of a 32x32 tile, per size: [4][64][16]
  // scratch for vvhip_mctf_motion_estimation (grown on demand)
  void*        d_scratch  = nullptr;
  size_t       scratchBytes = 0;
  // optional per-class events of the last vvhip_mctf_motion_estimation[_async] call (vvhip_mctf_set_timing): tag 0 call start, 1 before a level's candidate scoring, 2 after it,
  // 3 after the neighbour scoring, 4 after the sweep, 5 after the final normalisation, 6 call end
  bool         mctfTiming = false;
  std::vector<hipEvent_t> mctfEv;
  std::vector<int> mctfEvTag;
  unsigned long long* d_mctfStats = nullptr;   // scored-candidate counters of the MCTF search (vvhip_mctf_set_stats): 12 x uint64, null = off
  // scratch of vvhip_subpel_dist_batch: predicted blocks + distortion items (grown on demand)
  void*        d_subpel   = nullptr;
  size_t       subpelBytes = 0;
  // job table of the generic TU pipeline's merged launch (trquant.hip: tuRdoGenMultiKernel) + the host copy of what it holds
  void*        d_tuGen    = nullptr;
  size_t       tuGenBytes = 0;
  std::vector<unsigned char> tuGenLast;
  bool         tuSparse   = false;     // vvhip_tu_set_sparse_outputs: the fused TU launches of vvhip_tu_rdo_multi[_strided] write no levels / reconstruction for TUs whose levels are all zero
  hipStream_t  tuGenStream = nullptr;      // the stream the cached generic-TU job table was uploaded on (trquant.hip: tuRdoMulti); compared only, never used as a handle
  hipEvent_t   tuGenEvent  = nullptr;      // recorded behind every launch that reads the table: a caller that switches streams orders the new stream behind it
  bool         tuGenEventRecorded = false;
  // how the host waits for the stream (vvhip_set_blocking_sync): false = hipStreamSynchronize (the runtime's low-latency wait), true = a blocking event — the calling thread
  // sleeps, which matters when the host's cores are all busy encoding
  bool         blockingSync = false;
  hipEvent_t   syncEvent  = nullptr;
};

// every host wait of the library on a context's stream goes through this
inline hipError_t vvhip_wait_stream( vvhip_ctx* ctx )
{
  if( !ctx->blockingSync ) return hipStreamSynchronize( ctx->stream );
  if( !ctx->syncEvent ) return hipErrorInvalidHandle;      // (created by vvhip_create on the context's device)
  const hipError_t e = hipEventRecord( ctx->syncEvent, ctx->stream );
  return e != hipSuccess ? e : hipEventSynchronize( ctx->syncEvent );
}

int vvhip_fail( vvhip_ctx* ctx, int code, const char* fmt, ... );

#define VVHIP_CHECK_HIP( ctx, expr )                                                        \
  do { hipError_t e_ = ( expr );                                                            \
       if( e_ != hipSuccess ) return vvhip_fail( ctx, VVHIP_E_HIP, "%s: %s (%s:%d)", #expr, \
                                                 hipGetErrorString( e_ ), __FILE__, __LINE__ ); } while( 0 )

#define VVHIP_LAUNCH_CHECK( ctx ) VVHIP_CHECK_HIP( ctx, hipGetLastError() )

// ---- ROM layout -------------------------------------------------------------------------------
// transform matrices: type-major, sizes 2..64; entry (type, log2N) lives at trMatOffset(type, log2N), N*N int16
static inline __host__ __device__ int trMatOffset( int trType, int log2N )
{
  // sum_{l=1}^{log2N-1} 4^l = (4^log2N - 4) / 3 ; each type gets a full 2..64 slot (5460 entries)
  return trType * 5460 + ( ( 1 << ( 2 * log2N ) ) - 4 ) / 3;
}
static const int kTrMatTotal = 3 * 5460;

// scan orders for log2w, log2h in 0..6: entry lives at scanOffset(log2w, log2h), (w*h) uint16
static inline __host__ __device__ int scanOffset( int log2w, int log2h )
{
  // prefix over (lw, lh) in row-major order of 2^(lw+lh): sum_{a<lw} 2^a*(127) + 2^lw * (2^lh - 1)
  return ( ( 1 << log2w ) - 1 ) * 127 + ( 1 << log2w ) * ( ( 1 << log2h ) - 1 );
}
static const int kScanTotal = 127 * 127;

// ---- matrix-core operands of the fused TU kernel (trquant.hip, tuMx*) ---------------------------------------------------------
// A 32x32 tile holds (32/N)^2 TUs of size N side by side; "big" is the block-diagonal 32x32 matrix of 32/N copies of the N-point kernel
// matrix (all entries fit 8 bits).  v_mfma_i32_32x32x32_i8 takes 16 bytes per lane for each operand: lane l = 32*h + r supplies row r (A) /
// column r (B) and K-slots (h, 0..15); result register v of lane l holds row mxHw(h, v), column r.  The data operand of every pass is the
// previous pass's result registers (slot s = register s), so the matrix operand's K-slot (h, s) is bound to index mxHw(h, s); and the
// A-side matrices supply their rows in the order mxSigma, which makes result register (h, v) hold logical row 16h + v — a lane's 16
// registers are 16 consecutive rows of ONE column: one 16-point TU, two 8-point TUs or half a 32-point TU.
static inline __host__ __device__ int mxHw( int h, int s ) { return 8 * ( s >> 2 ) + 4 * h + ( s & 3 ); }
static inline __host__ __device__ int mxSigma( int i )     { return 16 * ( ( i >> 2 ) & 1 ) + 4 * ( i >> 3 ) + ( i & 3 ); }   // mxSigma( mxHw( h, v ) ) == 16h + v
struct VvhipTuMxOps
{
  int8_t  nat[64][16];         // B side, forward rows:     lane l: big[l % 32][16h + s]                   (data slots = samples 16h + s of the lane's row)
  int8_t  rowP[64][16];        // A side, forward columns:  lane l: big[mxSigma(l % 32)][mxHw(h, s)]
  int8_t  natT[64][16];        // B side, inverse columns:  lane l: big[16h + s][l % 32]
  int8_t  colP[64][16];        // A side, inverse rows:     lane l: big[mxHw(h, s)][mxSigma(l % 32)]
  int32_t rowSum[32];          // 128 * sum_x big[r][x]: corrections for the (byte - 128) form of the data's low bytes
  int32_t colSum[32];          // 128 * sum_k big[k][c]
};

// 64-point DCT-2 with its zero-out (only the 32 low frequencies per direction survive, TrQuant.cpp:496-497): one wave per 64x64 TU, the
// residual as two row tiles x two column chunks of 32, every contraction over 64 split into two accumulating products.
struct VvhipTuMx64Ops
{
  int8_t  natX[2][64][16];     // forward rows,    B side, x chunk c:  T[l % 32][32c + 16h + s]
  int8_t  rowPY[2][64][16];    // forward columns, A side, row tile t: T[mxSigma(l % 32)][32t + mxHw(h, s)]
  int8_t  natTY[2][64][16];    // inverse columns, B side, row tile t: T[16h + s][32t + l % 32]
  int8_t  colPX[2][64][16];    // inverse rows,    A side, x chunk c:  T[mxHw(h, s)][32c + mxSigma(l % 32)]
  int32_t rowSum[32];          // 128 * sum_{x < 64} T[k][x]
  int32_t colSum[64];          // 128 * sum_{k < 32} T[k][x]
  uint16_t pos[64][16];        // scan position of coefficient (16h + v, l % 32) inside the 64x64 TU
};

void vvhip_build_tr_matrix( int trType, int log2N, int16_t* out );          // host
void vvhip_build_scan_order( int log2w, int log2h, uint32_t* out );         // host
void vvhip_cg_size( int log2w, int log2h, int* log2CGw, int* log2CGh );     // host

static inline int ilog2i( int v ) { int l = 0; while( ( 1 << ( l + 1 ) ) <= v ) l++; return l; }
static inline bool isPow2( int v ) { return v > 0 && ( v & ( v - 1 ) ) == 0; }

// ---- cross-lane helpers (device) ----------------------------------------------------------------------------------------
// Sum over aligned groups of G (2..64, power of two) consecutive lanes using DPP only (no LDS crossbar traffic): quad_perm
// xor-1 / xor-2, row_half_mirror (8 lanes), row_mirror (16 lanes), then v_readlane of the four 16-lane row totals.
// The result is valid in every lane of the group.
#ifdef __HIPCC__
#define VVHIP_DPP( v, ctrl ) __builtin_amdgcn_mov_dpp( ( int ) ( v ), ( ctrl ), 0xf, 0xf, true )
#define VVHIP_DPP_XOR1        0xB1    /* quad_perm [1,0,3,2] */
#define VVHIP_DPP_XOR2        0x4E    /* quad_perm [2,3,0,1] */
#define VVHIP_DPP_HALF_MIRROR 0x141   /* lane i <-> 7-i inside 8 lanes  */
#define VVHIP_DPP_MIRROR      0x140   /* lane i <-> 15-i inside 16 lanes */

__device__ __forceinline__ uint32_t vvhipGroupSum32( uint32_t v, int G, int lane )
{
  if( G >= 2 )  v += ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR1 );
  if( G >= 4 )  v += ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR2 );
  if( G >= 8 )  v += ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_HALF_MIRROR );
  if( G >= 16 ) v += ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_MIRROR );
  if( G >= 32 )
  {
    const uint32_t r0 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 0 ),  r1 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 16 );
    const uint32_t r2 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 32 ), r3 = ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 48 );
    v = G == 64 ? r0 + r1 + r2 + r3 : ( lane < 32 ? r0 + r1 : r2 + r3 );
  }
  return v;
}
#define VVHIP_GROUP_REDUCE( NAME, OP )                                                                                         \
__device__ __forceinline__ uint32_t NAME( uint32_t v, int G, int lane )                                                          \
{                                                                                                                                \
  if( G >= 2 )  { const uint32_t o = ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR1 ); v = OP; }                                      \
  if( G >= 4 )  { const uint32_t o = ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_XOR2 ); v = OP; }                                      \
  if( G >= 8 )  { const uint32_t o = ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_HALF_MIRROR ); v = OP; }                               \
  if( G >= 16 ) { const uint32_t o = ( uint32_t ) VVHIP_DPP( v, VVHIP_DPP_MIRROR ); v = OP; }                                    \
  if( G >= 32 )                                                                                                                  \
  {                                                                                                                              \
    uint32_t r[4] = { ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 0 ), ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 16 ),       \
                      ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 32 ), ( uint32_t ) __builtin_amdgcn_readlane( ( int ) v, 48 ) };    \
    uint32_t a, b;                                                                                                               \
    { const uint32_t v = r[0], o = r[1]; a = OP; } { const uint32_t v = r[2], o = r[3]; b = OP; }                                \
    if( G == 64 ) { const uint32_t v = a, o = b; a = OP; b = a; }                                                                \
    v = lane < 32 ? a : b;                                                                                                       \
  }                                                                                                                              \
  return v;                                                                                                                      \
}
VVHIP_GROUP_REDUCE( vvhipGroupMax32, ( o > v ? o : v ) )
VVHIP_GROUP_REDUCE( vvhipGroupOr32, ( v | o ) )

// exact 64-bit group sum of per-lane values < 2^50 through two 32-bit limbs (24-bit split)
__device__ __forceinline__ unsigned long long vvhipGroupSum64( unsigned long long e, int G, int lane )
{
  const uint32_t lo = vvhipGroupSum32( ( uint32_t ) ( e & 0xFFFFFFull ), G, lane ), hi = vvhipGroupSum32( ( uint32_t ) ( e >> 24 ), G, lane );
  return ( ( unsigned long long ) hi << 24 ) + lo;
}
#endif
