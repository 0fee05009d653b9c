// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts in the access patterns of this repository's kernels
// (VERDICT r3 #3a; /opt/skills/guides/MI355X_MICROARCH.md: "other access widths ... calibrate on a known byte count in your own access pattern").
// Every kernel touches each byte of a 512 MiB buffer (far beyond the 32 MiB of L2 and the 256 MiB Infinity Cache) exactly once, so the bytes that must cross the
// L2's memory side are the buffer size whatever the request width:
//   stream16      16 bytes per lane, lanes consecutive, 16-byte aligned          (the guide's case: FETCH_SIZE reports half)
//   stream16_odd  the same stream starting 2 bytes into the buffer               (a candidate row at an odd x: 2-byte aligned 16-byte loads)
//   rows16        lane teams of eight: lane r reads 16 bytes of row y0 + r of an 8x8 block; the blocks of a wave are neighbours in x (stage / item / window kernels:
//                 per-lane row gathers out of a plane with a pitch of 4160 samples)
//   rec8          8 bytes per lane, consecutive                                   (job / item records)
//   store8        8 bytes per lane, consecutive                                   (Distortion results)            -> WRITE_SIZE
//   store16       16 bytes per lane, consecutive                                  (levels / residuals)             -> WRITE_SIZE
// tools/calib_fetch.py runs this binary under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and writes the factors
// known bytes / ( counter x 1024 ) per pattern.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__( ( ext_vector_type( 4 ) ) );
typedef uint32_t u32x2 __attribute__( ( ext_vector_type( 2 ) ) );
struct __attribute__( ( packed, aligned( 2 ) ) ) U16 { u32x4 v; };

#define CK( x ) do { hipError_t e_ = ( x ); if( e_ != hipSuccess ) { fprintf( stderr, "%s: %s\n", #x, hipGetErrorString( e_ ) ); exit( 1 ); } } while( 0 )

__global__ void __launch_bounds__( 256 ) stream16( const u32x4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink )
{
  uint32_t acc = 0;
  for( size_t i = ( size_t ) blockIdx.x * 256 + threadIdx.x; i < n16; i += ( size_t ) gridDim.x * 256 ) { const u32x4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if( acc == 0x12345678u ) sink[0] = acc;
}
__global__ void __launch_bounds__( 256 ) stream16_odd( const int16_t* __restrict__ src, size_t n16, uint32_t* __restrict__ sink )
{
  uint32_t acc = 0;
  for( size_t i = ( size_t ) blockIdx.x * 256 + threadIdx.x; i < n16; i += ( size_t ) gridDim.x * 256 ) { const u32x4 v = reinterpret_cast<const U16*>( src + 1 + 8 * i )->v; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if( acc == 0x12345678u ) sink[0] = acc;
}
// plane of `rows` rows x `pitch` samples; 8x8 blocks tile it (x0 = 8 bx + xo, xo odd: unaligned like a motion vector).  lane = ( team, r ): team = block, r = row
__global__ void __launch_bounds__( 256 ) rows16( const int16_t* __restrict__ plane, int pitch, int blocksX, size_t nBlocks, int xo, uint32_t* __restrict__ sink )
{
  uint32_t acc = 0;
  for( size_t t = ( size_t ) blockIdx.x * 32 + ( threadIdx.x >> 3 ); t < nBlocks; t += ( size_t ) gridDim.x * 32 )
  {
    const size_t by = t / blocksX, bx = t - by * blocksX;
    const int r = threadIdx.x & 7;
    const u32x4 v = reinterpret_cast<const U16*>( plane + ( by * 8 + r ) * ( size_t ) pitch + bx * 8 + xo )->v;
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if( acc == 0x12345678u ) sink[0] = acc;
}
__global__ void __launch_bounds__( 256 ) rec8( const u32x2* __restrict__ src, size_t n8, uint32_t* __restrict__ sink )
{
  uint32_t acc = 0;
  for( size_t i = ( size_t ) blockIdx.x * 256 + threadIdx.x; i < n8; i += ( size_t ) gridDim.x * 256 ) { const u32x2 v = src[i]; acc += v.x ^ v.y; }
  if( acc == 0x12345678u ) sink[0] = acc;
}
__global__ void __launch_bounds__( 256 ) store8( u32x2* __restrict__ dst, size_t n8 )
{
  for( size_t i = ( size_t ) blockIdx.x * 256 + threadIdx.x; i < n8; i += ( size_t ) gridDim.x * 256 ) { u32x2 v; v.x = ( uint32_t ) i; v.y = 7; dst[i] = v; }
}
__global__ void __launch_bounds__( 256 ) store16( u32x4* __restrict__ dst, size_t n16 )
{
  for( size_t i = ( size_t ) blockIdx.x * 256 + threadIdx.x; i < n16; i += ( size_t ) gridDim.x * 256 ) { u32x4 v; v.x = ( uint32_t ) i; v.y = 7; v.z = 9; v.w = 11; dst[i] = v; }
}

int main()
{
  const size_t bytes = ( size_t ) 512 << 20;
  void* buf = nullptr; uint32_t* sink = nullptr;
  CK( hipMalloc( &buf, bytes + 4096 ) ); CK( hipMalloc( &sink, 64 ) );
  CK( hipMemset( buf, 1, bytes + 4096 ) ); CK( hipDeviceSynchronize() );
  const int grid = 256 * 8;
  const int pitch = 4160, rowsN = ( int ) ( bytes / 2 / pitch ) & ~7, blocksX = ( pitch - 8 ) / 8;
  const size_t nBlocks = ( size_t ) ( rowsN / 8 ) * blocksX;
  for( int rep = 0; rep < 2; rep++ )      // (two dispatches each: the averages tools/calib_fetch.py takes are per dispatch)
  {
    hipLaunchKernelGGL( stream16, dim3( grid ), dim3( 256 ), 0, 0, ( const u32x4* ) buf, bytes / 16, sink );
    hipLaunchKernelGGL( stream16_odd, dim3( grid ), dim3( 256 ), 0, 0, ( const int16_t* ) buf, bytes / 16, sink );
    hipLaunchKernelGGL( rows16, dim3( grid ), dim3( 256 ), 0, 0, ( const int16_t* ) buf, pitch, blocksX, nBlocks, 3, sink );
    hipLaunchKernelGGL( rec8, dim3( grid ), dim3( 256 ), 0, 0, ( const u32x2* ) buf, bytes / 8, sink );
    hipLaunchKernelGGL( store8, dim3( grid ), dim3( 256 ), 0, 0, ( u32x2* ) buf, bytes / 8 );
    hipLaunchKernelGGL( store16, dim3( grid ), dim3( 256 ), 0, 0, ( u32x4* ) buf, bytes / 16 );
    CK( hipDeviceSynchronize() );
  }
  // the known byte counts, one line per kernel (read by tools/calib_fetch.py)
  printf( "KNOWN stream16 %zu\nKNOWN stream16_odd %zu\nKNOWN rows16 %zu\nKNOWN rec8 %zu\nKNOWN store8 %zu\nKNOWN store16 %zu\n",
          bytes, bytes, ( size_t ) nBlocks * 128, bytes, bytes, bytes );
  CK( hipFree( buf ) ); CK( hipFree( sink ) );
  return 0;
}
