// valu_rate.hip — issue rate of a few gfx950 VALU instructions the sub-pel stage kernel could use (v_mad_i32_i16 with op_sel against v_dot2_i32_i16, v_perm_b32, v_med3_i32,
// packed 16-bit shifts), and a check of v_mad_i32_i16's op_sel semantics.  One workgroup of 64 lanes per SIMD slot, N dependent-free instructions per loop trip:
// cycles per instruction = elapsed / ( trips x instructions ) per wave with ONE wave per SIMD (no overlap between waves).
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/valu_rate.hip -o tools/exp/valu_rate ; run on the GPU box: tools/exp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define REP8( X ) X( 0 ) X( 1 ) X( 2 ) X( 3 ) X( 4 ) X( 5 ) X( 6 ) X( 7 )

template<int OP>
__global__ void __launch_bounds__( 64 ) rate( const uint32_t* in, uint32_t* out, int trips )
{
  uint32_t x = in[threadIdx.x], y = in[64 + threadIdx.x];
  int a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  for( int t = 0; t < trips; t++ )
  {
#define DOT( i )  asm volatile( "v_dot2_i32_i16 %0, %1, %2, %0" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define MAD( i )  asm volatile( "v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define M24( i )  asm volatile( "v_mad_i32_i24 %0, %1, %2, %0" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define PERM( i ) asm volatile( "v_perm_b32 %0, %0, %1, %2" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define MED( i )  asm volatile( "v_med3_i32 %0, %0, %1, %2" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define PKSH( i ) asm volatile( "v_pk_ashrrev_i16 %0, 1, %0" : "+v"( a##i ) );
#define PKSUB( i ) asm volatile( "v_pk_sub_i16 %0, %0, %1" : "+v"( a##i ) : "v"( x ) );
#define MUL24( i ) asm volatile( "v_mul_i32_i24 %0, %0, %1" : "+v"( a##i ) : "v"( x ) );
#define ADD( i )  asm volatile( "v_add_u32 %0, %0, %1" : "+v"( a##i ) : "v"( x ) );
#define DOTC( i ) asm volatile( "v_dot2c_i32_i16 %0, %1, %2" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define ADD3( i ) asm volatile( "v_add3_u32 %0, %0, %1, %2" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define ADD64( i ) asm volatile( "v_add_u32_e64 %0, %0, %1" : "+v"( a##i ) : "v"( x ) );
#define SAD( i )  asm volatile( "v_sad_u16 %0, %1, %2, %0" : "+v"( a##i ) : "v"( x ), "v"( y ) );
#define ALIGN( i ) asm volatile( "v_alignbit_b32 %0, %0, %1, 16" : "+v"( a##i ) : "v"( x ) );
#define ADDDPP( i ) asm volatile( "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"( a##i ) : "v"( x ) );
#define PKADD( i ) asm volatile( "v_pk_add_i16 %0, %0, %1" : "+v"( a##i ) : "v"( x ) );
#define SDWA( i ) asm volatile( "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"( a##i ) : "v"( x ) );
    if( OP == 9 )  { REP8( DOTC ) REP8( DOTC ) }
    if( OP == 10 ) { REP8( ADD3 ) REP8( ADD3 ) }
    if( OP == 11 ) { REP8( ADD64 ) REP8( ADD64 ) }
    if( OP == 12 ) { REP8( SAD ) REP8( SAD ) }
    if( OP == 13 ) { REP8( ALIGN ) REP8( ALIGN ) }
    if( OP == 14 ) { REP8( ADDDPP ) REP8( ADDDPP ) }
    if( OP == 15 ) { REP8( PKADD ) REP8( PKADD ) }
    if( OP == 16 ) { REP8( SDWA ) REP8( SDWA ) }
    if( OP == 0 ) { REP8( DOT ) REP8( DOT ) }
    if( OP == 1 ) { REP8( MAD ) REP8( MAD ) }
    if( OP == 2 ) { REP8( M24 ) REP8( M24 ) }
    if( OP == 3 ) { REP8( PERM ) REP8( PERM ) }
    if( OP == 4 ) { REP8( MED ) REP8( MED ) }
    if( OP == 5 ) { REP8( PKSH ) REP8( PKSH ) }
    if( OP == 6 ) { REP8( PKSUB ) REP8( PKSUB ) }
    if( OP == 7 ) { REP8( MUL24 ) REP8( MUL24 ) }
    if( OP == 8 ) { REP8( ADD ) REP8( ADD ) }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void semantics( const uint32_t* in, int* out )
{
  const uint32_t x = in[threadIdx.x], y = in[64 + threadIdx.x]; const int acc = 1000;
  int r0, r1, r2, r3;
  asm volatile( "v_mad_i32_i16 %0, %1, %2, %3" : "=v"( r0 ) : "v"( x ), "v"( y ), "v"( acc ) );                        // x.lo * y.lo + acc
  asm volatile( "v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"( r1 ) : "v"( x ), "v"( y ), "v"( acc ) );     // x.hi * y.lo + acc
  asm volatile( "v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"( r2 ) : "v"( x ), "v"( y ), "v"( acc ) );     // x.lo * y.hi + acc
  asm volatile( "v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"( r3 ) : "v"( x ), "v"( y ), "v"( acc ) );     // x.hi * y.hi + acc
  out[4 * threadIdx.x] = r0; out[4 * threadIdx.x + 1] = r1; out[4 * threadIdx.x + 2] = r2; out[4 * threadIdx.x + 3] = r3;
}

template<int OP> static void run( const char* name, const uint32_t* dIn, uint32_t* dOut, int wavesPerSimd = 1 )
{
  const int trips = 20000, blocks = 1024 * wavesPerSimd;      // 1024 = one wave per SIMD (256 CUs x 4)
  hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
  rate<OP><<<blocks, 64>>>( dIn, dOut, 100 );
  hipDeviceSynchronize();
  hipEventRecord( e0 ); rate<OP><<<blocks, 64>>>( dIn, dOut, trips ); hipEventRecord( e1 ); hipEventSynchronize( e1 );
  float ms = 0; hipEventElapsedTime( &ms, e0, e1 );
  printf( "%-22s %d wave(s) per SIMD %8.3f ms  %6.2f ns per instruction and SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n", name, wavesPerSimd, ms, ms * 1e6 / ( ( double ) trips * 16 * wavesPerSimd ) );
}

int main()
{
  std::vector<uint32_t> h( 128 );
  for( int i = 0; i < 128; i++ ) h[i] = ( uint32_t ) ( ( ( i * 2654435761u ) >> 7 ) ^ ( i * 40503u << 13 ) );
  uint32_t* dIn; uint32_t* dOut; int* dSem;
  hipMalloc( &dIn, 512 ); hipMalloc( &dOut, 8 * 1024 * 64 * 4 ); hipMalloc( &dSem, 64 * 16 );
  hipMemcpy( dIn, h.data(), 512, hipMemcpyHostToDevice );
  semantics<<<1, 64>>>( dIn, dSem );
  std::vector<int> s( 256 ); hipMemcpy( s.data(), dSem, 1024, hipMemcpyDeviceToHost );
  int bad = 0;
  for( int i = 0; i < 64; i++ )
  {
    const int xl = ( int16_t ) ( h[i] & 0xffff ), xh = ( int16_t ) ( h[i] >> 16 ), yl = ( int16_t ) ( h[64 + i] & 0xffff ), yh = ( int16_t ) ( h[64 + i] >> 16 );
    const int want[4] = { xl * yl + 1000, xh * yl + 1000, xl * yh + 1000, xh * yh + 1000 };
    for( int k = 0; k < 4; k++ ) if( s[4 * i + k] != want[k] ) { if( bad < 8 ) printf( "lane %d form %d: got %d want %d\n", i, k, s[4 * i + k], want[k] ); bad++; }
  }
  printf( "v_mad_i32_i16 op_sel semantics: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as expected", bad );
  run<0>( "v_dot2_i32_i16", dIn, dOut ); run<1>( "v_mad_i32_i16 op_sel", dIn, dOut ); run<2>( "v_mad_i32_i24", dIn, dOut ); run<3>( "v_perm_b32", dIn, dOut );
  run<4>( "v_med3_i32", dIn, dOut ); run<5>( "v_pk_ashrrev_i16", dIn, dOut ); run<6>( "v_pk_sub_i16", dIn, dOut ); run<7>( "v_mul_i32_i24", dIn, dOut ); run<8>( "v_add_u32", dIn, dOut );
  for( int w = 2; w <= 8; w *= 2 ) { run<8>( "v_add_u32", dIn, dOut, w ); run<1>( "v_mad_i32_i16 op_sel", dIn, dOut, w ); run<0>( "v_dot2_i32_i16", dIn, dOut, w ); }
  for( int w = 4; w <= 8; w *= 2 )
  {
    run<9>( "v_dot2c_i32_i16 (VOP2)", dIn, dOut, w ); run<10>( "v_add3_u32", dIn, dOut, w ); run<11>( "v_add_u32_e64", dIn, dOut, w ); run<12>( "v_sad_u16", dIn, dOut, w );
    run<13>( "v_alignbit_b32", dIn, dOut, w ); run<14>( "v_add_u32_dpp", dIn, dOut, w ); run<15>( "v_pk_add_i16", dIn, dOut, w ); run<16>( "v_sub_u32_sdwa", dIn, dOut, w );
    run<2>( "v_mad_i32_i24", dIn, dOut, w ); run<3>( "v_perm_b32", dIn, dOut, w ); run<4>( "v_med3_i32", dIn, dOut, w ); run<7>( "v_mul_i32_i24", dIn, dOut, w );
  }
  return 0;
}
