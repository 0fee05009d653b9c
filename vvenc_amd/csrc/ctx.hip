// ctx.hip — context, stream, device-memory helpers and ROM tables of libvvenc_hip.so.
//
// ROM tables are rebuilt here from the standard's coefficient lists rather than stored:
//   * DCT-2 / DST-7 / DCT-8 integer kernels  == g_trCore*      (CommonLib/RomTr.cpp:364-449)
//   * grouped up-right diagonal coefficient scans == getScanOrder(SCAN_GROUPED_4x4,..) (CommonLib/Rom.cpp:1098-1284)
// and are checked entry-by-entry against the reference in tests/ (golden + live reference).
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <atomic>
#include <vector>

static std::string g_createError;

int vvhip_fail( vvhip_ctx* ctx, int code, const char* fmt, ... )
{
  char buf[512];
  va_list ap;
  va_start( ap, fmt );
  vsnprintf( buf, sizeof( buf ), fmt, ap );
  va_end( ap );
  if( ctx ) ctx->lastError = buf; else g_createError = buf;
  return code;
}

// ---- transform matrices ----------------------------------------------------------------------
// One coefficient per distinct angle (H.266 8.7.4.2): DCT-2 via m = (2n+1)k folded to [0,64];
// DST-7 via j = (2k+1)(n+1) folded to [0,N]; DCT-8[k][n] = (-1)^k DST-7[k][N-1-n].
static const int16_t kDct2Cos[65] = {
  64, 91, 90, 90, 90, 90, 90, 90, 89, 88, 88, 87, 87, 86, 85, 84, 83, 83, 82, 81, 80, 79, 78, 77, 75, 73, 73, 71, 70, 69, 67, 65, 64,
  62, 61, 59, 57, 56, 54, 52, 50, 48, 46, 44, 43, 41, 38, 37, 36, 33, 31, 28, 25, 24, 22, 20, 18, 15, 13, 11, 9, 7, 4, 2, 0 };
static const int16_t kDst7Sin4[4]   = { 29, 55, 74, 84 };
static const int16_t kDst7Sin8[8]   = { 17, 32, 46, 60, 71, 78, 85, 86 };
static const int16_t kDst7Sin16[16] = { 8, 17, 25, 33, 40, 48, 55, 62, 68, 73, 77, 81, 85, 87, 88, 88 };
static const int16_t kDst7Sin32[32] = { 4, 9, 13, 17, 21, 26, 30, 34, 38, 42, 46, 50, 53, 56, 60, 63,
                                        66, 68, 72, 74, 77, 78, 80, 82, 84, 85, 86, 87, 88, 89, 90, 90 };

static int16_t dct2At( int N, int k, int n )
{
  int m = ( ( 2 * n + 1 ) * k * ( 64 / N ) ) & 255;
  int sign = 1;
  if( m > 128 ) m = 256 - m;
  if( m > 64 ) { m = 128 - m; sign = -1; }
  return ( int16_t ) ( sign * kDct2Cos[m] );
}

static int16_t dst7At( int N, int k, int n )
{
  const int16_t* s = N == 4 ? kDst7Sin4 : N == 8 ? kDst7Sin8 : N == 16 ? kDst7Sin16 : kDst7Sin32;
  const int period = 2 * N + 1;
  int j = ( ( 2 * k + 1 ) * ( n + 1 ) ) % ( 2 * period );
  int sign = 1;
  if( j > period ) { j -= period; sign = -1; }
  if( j > N ) j = period - j;
  return j ? ( int16_t ) ( sign * s[j - 1] ) : ( int16_t ) 0;
}

void vvhip_build_tr_matrix( int trType, int log2N, int16_t* out )
{
  const int N = 1 << log2N;
  for( int k = 0; k < N; k++ )
    for( int n = 0; n < N; n++ )
    {
      int16_t v = 0;
      if( trType == VVHIP_DCT2 ) v = dct2At( N, k, n );
      else if( log2N >= 2 && log2N <= 5 )
        v = trType == VVHIP_DST7 ? dst7At( N, k, n ) : ( int16_t ) ( ( k & 1 ? -1 : 1 ) * dst7At( N, k, N - 1 - n ) );
      out[k * N + n] = v;
    }
}

// ---- coefficient scan ------------------------------------------------------------------------
void vvhip_cg_size( int log2w, int log2h, int* cgw, int* cgh )   // g_log2SbbSize, CommonLib/Rom.cpp:1138-1148
{
  if( log2w >= 2 && log2h >= 2 ) { *cgw = 2; *cgh = 2; }
  else if( log2w == 0 ) { *cgw = 0; *cgh = log2h < 4 ? log2h : 4; }
  else if( log2h == 0 ) { *cgh = 0; *cgw = log2w < 4 ? log2w : 4; }
  else if( log2w == 1 ) { *cgw = 1; *cgh = log2h <= 2 ? 1 : 3; }
  else                  { *cgh = 1; *cgw = log2w <= 2 ? 1 : 3; }
}

static void upRightDiagonal( int bw, int bh, std::vector<int>& xs, std::vector<int>& ys )
{
  xs.clear(); ys.clear();
  for( int d = 0; d < bw + bh - 1; d++ )
    for( int y = d < bh ? d : bh - 1, x = d - y; y >= 0 && x < bw; x++, y-- ) { xs.push_back( x ); ys.push_back( y ); }
}

void vvhip_build_scan_order( int log2w, int log2h, uint32_t* out )
{
  const int w = 1 << log2w, h = 1 << log2h;
  int cgw, cgh;
  vvhip_cg_size( log2w, log2h, &cgw, &cgh );
  const int gw = 1 << cgw, gh = 1 << cgh, gsize = gw * gh;
  const int wInG = ( w < 32 ? w : 32 ) >> cgw, hInG = ( h < 32 ? h : 32 ) >> cgh;
  for( int i = 0; i < w * h; i++ ) out[i] = ( uint32_t ) ( w * h - 1 );   // zero-out region filler (reference does the same)
  std::vector<int> gx, gy, px, py;
  upRightDiagonal( wInG, hInG, gx, gy );
  upRightDiagonal( gw, gh, px, py );
  for( int g = 0; g < wInG * hInG; g++ )
    for( int p = 0; p < gsize; p++ )
      out[g * gsize + p] = ( uint32_t ) ( ( gy[g] * gh + py[p] ) * w + gx[g] * gw + px[p] );
}

extern "C" {

int vvhip_get_tr_matrix_host( int tr_type, int log2_size, int16_t* host_out )
{
  if( tr_type < 0 || tr_type > 2 || log2_size < ( tr_type == VVHIP_DCT2 ? 1 : 2 ) || log2_size > ( tr_type == VVHIP_DCT2 ? 6 : 5 ) ) return VVHIP_E_ARG;
  vvhip_build_tr_matrix( tr_type, log2_size, host_out );
  return VVHIP_OK;
}

int vvhip_get_scan_order_host( int log2_w, int log2_h, uint32_t* host_out )
{
  if( log2_w < 0 || log2_w > 6 || log2_h < 0 || log2_h > 6 ) return VVHIP_E_ARG;
  vvhip_build_scan_order( log2_w, log2_h, host_out );
  return VVHIP_OK;
}

const char* vvhip_version( void ) { return "vvenc_hip 0.1 (gfx950)"; }

// devices that hold (or held) a context of this process: vvhip_sync_all_devices waits for these only — touching every visible GPU would create a primary context on each
// (eight on an MI355X node) and stall callers that hold a lock while they wait
static std::atomic<uint64_t> g_liveDevices{ 0 };

int vvhip_create( vvhip_ctx** out, int device )
{
  if( !out ) return vvhip_fail( nullptr, VVHIP_E_ARG, "vvhip_create: out == NULL" );
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount( &count );
  if( e != hipSuccess || count <= 0 )
    return vvhip_fail( nullptr, VVHIP_E_HIP, "vvhip_create: no HIP device available (%s); this library has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString( e ) : "device count 0" );
  if( device < 0 || device >= count ) return vvhip_fail( nullptr, VVHIP_E_ARG, "vvhip_create: device %d out of range (%d devices)", device, count );
  // released to the caller only on success: every early return below destroys the context with whatever it already owns
  struct Guard { vvhip_ctx* c; ~Guard() { if( c ) vvhip_destroy( c ); } } guard{ new vvhip_ctx };
  vvhip_ctx* ctx = guard.c;
  ctx->device = device;
  VVHIP_CHECK_HIP( nullptr, hipSetDevice( device ) );
  VVHIP_CHECK_HIP( nullptr, hipStreamCreateWithFlags( &ctx->ownStream, hipStreamNonBlocking ) );
  ctx->stream = ctx->ownStream;
  // the blocking wait's event belongs to the context's device: created here, where that device is current (a lazily created one would land on whatever device the
  // calling thread has selected and fail on the context's stream)
  VVHIP_CHECK_HIP( nullptr, hipEventCreateWithFlags( &ctx->syncEvent, hipEventBlockingSync | hipEventDisableTiming ) );
  if( device < 64 ) g_liveDevices.fetch_or( 1ull << device );

  std::vector<int16_t> mats( kTrMatTotal, 0 );
  for( int t = 0; t < 3; t++ )
    for( int l = ( t == VVHIP_DCT2 ? 1 : 2 ); l <= ( t == VVHIP_DCT2 ? 6 : 5 ); l++ )
      vvhip_build_tr_matrix( t, l, mats.data() + trMatOffset( t, l ) );
  std::vector<uint16_t> scans( kScanTotal, 0 );
  std::vector<uint32_t> tmp( 4096 );
  for( int lw = 0; lw <= 6; lw++ )
    for( int lh = 0; lh <= 6; lh++ )
    {
      vvhip_build_scan_order( lw, lh, tmp.data() );
      uint16_t* dst = scans.data() + scanOffset( lw, lh );
      for( int i = 0; i < ( 1 << ( lw + lh ) ); i++ ) dst[i] = ( uint16_t ) tmp[i];
    }
  // matrix-core operand records (common.h: VvhipTuMxOps) and per-register scan positions for 4-, 8-, 16- and 32-point square TUs: record [type * 4 + log2N - 2]
  std::vector<VvhipTuMxOps> mx( 12 );
  std::vector<uint16_t> mxPos( 4 * 64 * 16 );
  for( int z = 0; z < 4; z++ )
  {
    const int l2 = 2 + z, n = 1 << l2;
    for( int t = 0; t < 3; t++ )
    {
      const int16_t* m = mats.data() + trMatOffset( t, l2 );
      auto big = [&]( int r, int c ) -> int { return ( r / n == c / n ) ? m[( r % n ) * n + ( c % n )] : 0; };
      VvhipTuMxOps& o = mx[t * 4 + z];
      for( int r = 0; r < 32; r++ )
      {
        int rs = 0, cs = 0;
        for( int c = 0; c < 32; c++ )
        {
          if( big( r, c ) < -128 || big( r, c ) > 127 ) return vvhip_fail( nullptr, VVHIP_E_HIP, "vvhip_create: transform matrix entry outside 8 bits" );
          rs += big( r, c ); cs += big( c, r );
        }
        o.rowSum[r] = 128 * rs; o.colSum[r] = 128 * cs;
      }
      for( int l = 0; l < 64; l++ )
        for( int k = 0; k < 16; k++ )
        {
          const int h = l / 32, r = l % 32;
          o.nat[l][k]  = ( int8_t ) big( r, 16 * h + k );
          o.rowP[l][k] = ( int8_t ) big( mxSigma( r ), mxHw( h, k ) );
          o.natT[l][k] = ( int8_t ) big( 16 * h + k, r );
          o.colP[l][k] = ( int8_t ) big( mxHw( h, k ), mxSigma( r ) );
        }
    }
    const uint16_t* sc = scans.data() + scanOffset( l2, l2 );          // scan position -> raster position
    std::vector<uint16_t> inv( n * n );
    for( int i = 0; i < n * n; i++ ) inv[sc[i]] = ( uint16_t ) i;
    for( int l = 0; l < 64; l++ )
      for( int v = 0; v < 16; v++ ) mxPos[( z * 64 + l ) * 16 + v] = inv[( ( 16 * ( l / 32 ) + v ) % n ) * n + ( l % 32 ) % n];
  }
  {
    std::unique_ptr<VvhipTuMx64Ops> o64p( new VvhipTuMx64Ops );      // (local: contexts may be created concurrently, one per device / worker thread)
    VvhipTuMx64Ops& o64 = *o64p;
    const int16_t* m = mats.data() + trMatOffset( VVHIP_DCT2, 6 );
    auto T = [&]( int r, int c ) -> int { return m[r * 64 + c]; };
    for( int r = 0; r < 32; r++ ) { int rs = 0; for( int c = 0; c < 64; c++ ) { if( T( r, c ) < -128 || T( r, c ) > 127 ) return vvhip_fail( nullptr, VVHIP_E_HIP, "vvhip_create: 64-point matrix entry outside 8 bits" ); rs += T( r, c ); } o64.rowSum[r] = 128 * rs; }
    for( int c = 0; c < 64; c++ ) { int cs = 0; for( int r = 0; r < 32; r++ ) cs += T( r, c ); o64.colSum[c] = 128 * cs; }
    for( int q = 0; q < 2; q++ )
      for( int l = 0; l < 64; l++ )
        for( int k = 0; k < 16; k++ )
        {
          const int h = l / 32, r = l % 32;
          o64.natX[q][l][k]  = ( int8_t ) T( r, 32 * q + 16 * h + k );
          o64.rowPY[q][l][k] = ( int8_t ) T( mxSigma( r ), 32 * q + mxHw( h, k ) );
          o64.natTY[q][l][k] = ( int8_t ) T( 16 * h + k, 32 * q + r );
          o64.colPX[q][l][k] = ( int8_t ) T( mxHw( h, k ), 32 * q + mxSigma( r ) );
        }
    const uint16_t* sc = scans.data() + scanOffset( 6, 6 );            // scan position -> raster position (64-wide), first 32*32 positions = the 32x32 region
    std::vector<uint16_t> inv( 64 * 64, 0xffff );
    for( int i = 0; i < 32 * 32; i++ ) inv[sc[i]] = ( uint16_t ) i;
    for( int l = 0; l < 64; l++ ) for( int v = 0; v < 16; v++ ) o64.pos[l][v] = inv[( 16 * ( l / 32 ) + v ) * 64 + l % 32];
    VVHIP_CHECK_HIP( nullptr, hipMalloc( ( void** ) &ctx->d_tuMx64, sizeof( o64 ) ) );
    VVHIP_CHECK_HIP( nullptr, hipMemcpy( ctx->d_tuMx64, &o64, sizeof( o64 ), hipMemcpyHostToDevice ) );
  }
  VVHIP_CHECK_HIP( nullptr, hipMalloc( ( void** ) &ctx->d_tuMx, mx.size() * sizeof( VvhipTuMxOps ) ) );
  VVHIP_CHECK_HIP( nullptr, hipMalloc( ( void** ) &ctx->d_tuMxPos, mxPos.size() * sizeof( uint16_t ) ) );
  VVHIP_CHECK_HIP( nullptr, hipMemcpy( ctx->d_tuMx, mx.data(), mx.size() * sizeof( VvhipTuMxOps ), hipMemcpyHostToDevice ) );
  VVHIP_CHECK_HIP( nullptr, hipMemcpy( ctx->d_tuMxPos, mxPos.data(), mxPos.size() * sizeof( uint16_t ), hipMemcpyHostToDevice ) );
  VVHIP_CHECK_HIP( nullptr, hipMalloc( ( void** ) &ctx->d_trMat, mats.size() * sizeof( int16_t ) ) );
  VVHIP_CHECK_HIP( nullptr, hipMalloc( ( void** ) &ctx->d_scan, scans.size() * sizeof( uint16_t ) ) );
  VVHIP_CHECK_HIP( nullptr, hipMemcpy( ctx->d_trMat, mats.data(), mats.size() * sizeof( int16_t ), hipMemcpyHostToDevice ) );
  VVHIP_CHECK_HIP( nullptr, hipMemcpy( ctx->d_scan, scans.data(), scans.size() * sizeof( uint16_t ), hipMemcpyHostToDevice ) );
  guard.c = nullptr;
  *out = ctx;
  return VVHIP_OK;
}

void vvhip_destroy( vvhip_ctx* ctx )
{
  if( !ctx ) return;
  ( void ) hipSetDevice( ctx->device );
  if( ctx->ownStream ) ( void ) hipStreamSynchronize( ctx->ownStream );
  if( ctx->d_trMat ) ( void ) hipFree( ctx->d_trMat );
  if( ctx->d_scan ) ( void ) hipFree( ctx->d_scan );
  if( ctx->d_tuMx ) ( void ) hipFree( ctx->d_tuMx );
  if( ctx->d_tuMxPos ) ( void ) hipFree( ctx->d_tuMxPos );
  if( ctx->d_tuMx64 ) ( void ) hipFree( ctx->d_tuMx64 );
  if( ctx->d_scratch ) ( void ) hipFree( ctx->d_scratch );
  if( ctx->d_subpel ) ( void ) hipFree( ctx->d_subpel );
  if( ctx->d_tuGen ) ( void ) hipFree( ctx->d_tuGen );
  if( ctx->syncEvent ) ( void ) hipEventDestroy( ctx->syncEvent );
  if( ctx->tuGenEvent ) ( void ) hipEventDestroy( ctx->tuGenEvent );
  if( ctx->d_mctfStats ) ( void ) hipFree( ctx->d_mctfStats );
  for( hipEvent_t e : ctx->mctfEv ) ( void ) hipEventDestroy( e );
  if( ctx->ownStream ) ( void ) hipStreamDestroy( ctx->ownStream );
  delete ctx;
}

const char* vvhip_last_error( const vvhip_ctx* ctx ) { return ctx ? ctx->lastError.c_str() : g_createError.c_str(); }

int vvhip_set_stream( vvhip_ctx* ctx, void* hip_stream )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( hip_stream )
  {
    // a borrowed stream must live on the context's device: the ROM tables and scratch areas do
    hipDevice_t dev = 0;
    if( hipStreamGetDevice( ( hipStream_t ) hip_stream, &dev ) == hipSuccess && ( int ) dev != ctx->device )
      return vvhip_fail( ctx, VVHIP_E_ARG, "vvhip_set_stream: the stream belongs to device %d, the context to device %d", ( int ) dev, ctx->device );
    ( void ) hipGetLastError();
  }
  ctx->stream = ( hipStream_t ) hip_stream;
  return VVHIP_OK;
}

int vvhip_use_own_stream( vvhip_ctx* ctx )
{
  if( !ctx ) return VVHIP_E_ARG;
  ctx->stream = ctx->ownStream;
  return VVHIP_OK;
}

void* vvhip_get_stream( vvhip_ctx* ctx ) { return ctx ? ( void* ) ctx->stream : nullptr; }

int vvhip_set_blocking_sync( vvhip_ctx* ctx, int on )
{
  if( !ctx ) return VVHIP_E_ARG;
  ctx->blockingSync = on != 0;
  return VVHIP_OK;
}

int vvhip_sync( vvhip_ctx* ctx )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, vvhip_wait_stream( ctx ) );
  return VVHIP_OK;
}

int vvhip_sync_all_devices( vvhip_ctx* ctx )
{
  if( !ctx ) return VVHIP_E_ARG;
  int prev = 0, n = 0;
  VVHIP_CHECK_HIP( ctx, hipGetDevice( &prev ) );
  VVHIP_CHECK_HIP( ctx, hipGetDeviceCount( &n ) );
  const uint64_t live = g_liveDevices.load();
  hipError_t first = hipSuccess;
  for( int d = 0; d < n && d < 64; d++ )
  {
    if( !( ( live >> d ) & 1 ) ) continue;                 // no context of this library ever lived there: nothing of ours can be in flight
    hipError_t e = hipSetDevice( d );
    if( e == hipSuccess ) e = hipDeviceSynchronize();
    if( e != hipSuccess && first == hipSuccess ) first = e;
  }
  ( void ) hipSetDevice( prev );
  VVHIP_CHECK_HIP( ctx, first );
  return VVHIP_OK;
}

struct vvhip_graph { hipGraph_t graph; hipGraphExec_t exec; };

int vvhip_graph_begin( vvhip_ctx* ctx )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipStreamBeginCapture( ctx->stream, hipStreamCaptureModeThreadLocal ) );
  return VVHIP_OK;
}

int vvhip_graph_end( vvhip_ctx* ctx, vvhip_graph** out )
{
  if( !ctx || !out ) return VVHIP_E_ARG;
  *out = nullptr;
  hipGraph_t g = nullptr;
  VVHIP_CHECK_HIP( ctx, hipStreamEndCapture( ctx->stream, &g ) );
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate( &e, g, nullptr, nullptr, 0 );
  if( err != hipSuccess ) { ( void ) hipGraphDestroy( g ); return vvhip_fail( ctx, VVHIP_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString( err ) ); }
  *out = new vvhip_graph{ g, e };
  return VVHIP_OK;
}

int vvhip_graph_launch( vvhip_ctx* ctx, vvhip_graph* graph )
{
  if( !ctx || !graph ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipGraphLaunch( graph->exec, ctx->stream ) );
  return VVHIP_OK;
}

void vvhip_graph_destroy( vvhip_graph* graph )
{
  if( !graph ) return;
  ( void ) hipGraphExecDestroy( graph->exec );
  ( void ) hipGraphDestroy( graph->graph );
  delete graph;
}

int vvhip_malloc( vvhip_ctx* ctx, void** d_ptr, size_t bytes )
{
  if( !ctx || !d_ptr ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipSetDevice( ctx->device ) );
  hipError_t e = hipMalloc( d_ptr, bytes ? bytes : 1 );
  if( e != hipSuccess ) return vvhip_fail( ctx, VVHIP_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString( e ) );
  return VVHIP_OK;
}

int vvhip_free( vvhip_ctx* ctx, void* d_ptr )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipFree( d_ptr ) );
  return VVHIP_OK;
}

int vvhip_upload( vvhip_ctx* ctx, void* d_dst, const void* host_src, size_t bytes )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipMemcpyAsync( d_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream ) );
  return VVHIP_OK;
}

int vvhip_download( vvhip_ctx* ctx, void* host_dst, const void* d_src, size_t bytes )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipMemcpyAsync( host_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream ) );
  VVHIP_CHECK_HIP( ctx, vvhip_wait_stream( ctx ) );
  return VVHIP_OK;
}

int vvhip_download_async( vvhip_ctx* ctx, void* host_dst, const void* d_src, size_t bytes )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipMemcpyAsync( host_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream ) );
  return VVHIP_OK;
}

int vvhip_upload_2d( vvhip_ctx* ctx, void* d_dst, size_t dst_pitch, const void* host_src, size_t src_pitch, size_t width_bytes, size_t rows )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !rows || !width_bytes ) return VVHIP_OK;
  VVHIP_CHECK_HIP( ctx, hipMemcpy2DAsync( d_dst, dst_pitch, host_src, src_pitch, width_bytes, rows, hipMemcpyHostToDevice, ctx->stream ) );
  return VVHIP_OK;
}

int vvhip_download_2d( vvhip_ctx* ctx, void* host_dst, size_t dst_pitch, const void* d_src, size_t src_pitch, size_t width_bytes, size_t rows )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( !rows || !width_bytes ) return VVHIP_OK;
  VVHIP_CHECK_HIP( ctx, hipMemcpy2DAsync( host_dst, dst_pitch, d_src, src_pitch, width_bytes, rows, hipMemcpyDeviceToHost, ctx->stream ) );
  VVHIP_CHECK_HIP( ctx, vvhip_wait_stream( ctx ) );
  return VVHIP_OK;
}

int vvhip_host_register( vvhip_ctx* ctx, const void* host_ptr, size_t bytes )
{
  if( !ctx || !host_ptr || !bytes ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipSetDevice( ctx->device ) );
  hipError_t e = hipHostRegister( const_cast<void*>( host_ptr ), bytes, hipHostRegisterPortable );
  if( e == hipErrorHostMemoryAlreadyRegistered ) { ( void ) hipGetLastError(); return VVHIP_OK; }
  if( e != hipSuccess ) { ( void ) hipGetLastError(); return vvhip_fail( ctx, VVHIP_E_HIP, "hipHostRegister(%zu): %s", bytes, hipGetErrorString( e ) ); }
  return VVHIP_OK;
}

int vvhip_host_unregister( vvhip_ctx* ctx, const void* host_ptr )
{
  if( !ctx || !host_ptr ) return VVHIP_E_ARG;
  hipError_t e = hipHostUnregister( const_cast<void*>( host_ptr ) );
  if( e != hipSuccess ) { ( void ) hipGetLastError(); return vvhip_fail( ctx, VVHIP_E_HIP, "hipHostUnregister: %s", hipGetErrorString( e ) ); }
  return VVHIP_OK;
}

int vvhip_host_alloc( vvhip_ctx* ctx, void** host_ptr, size_t bytes )
{
  if( !ctx || !host_ptr ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipSetDevice( ctx->device ) );
  hipError_t e = hipHostMalloc( host_ptr, bytes ? bytes : 1, hipHostMallocPortable );
  if( e != hipSuccess ) { ( void ) hipGetLastError(); return vvhip_fail( ctx, VVHIP_E_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString( e ) ); }
  return VVHIP_OK;
}

int vvhip_host_free( vvhip_ctx* ctx, void* host_ptr )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipHostFree( host_ptr ) );
  return VVHIP_OK;
}

int vvhip_event_create( vvhip_ctx* ctx, void** event )
{
  if( !ctx || !event ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipSetDevice( ctx->device ) );
  hipEvent_t e;
  VVHIP_CHECK_HIP( ctx, hipEventCreateWithFlags( &e, hipEventDisableTiming ) );
  *event = e;
  return VVHIP_OK;
}

int vvhip_event_record( vvhip_ctx* ctx, void* event )
{
  if( !ctx || !event ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipEventRecord( static_cast<hipEvent_t>( event ), ctx->stream ) );
  return VVHIP_OK;
}

int vvhip_event_wait( vvhip_ctx* ctx, void* event )
{
  if( !ctx || !event ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipEventSynchronize( static_cast<hipEvent_t>( event ) ) );
  return VVHIP_OK;
}

int vvhip_event_destroy( vvhip_ctx* ctx, void* event )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( event ) VVHIP_CHECK_HIP( ctx, hipEventDestroy( static_cast<hipEvent_t>( event ) ) );
  return VVHIP_OK;
}

int vvhip_device_count( void )
{
  int count = 0;
  if( hipGetDeviceCount( &count ) != hipSuccess ) { ( void ) hipGetLastError(); return 0; }
  return count;
}

int vvhip_get_device( const vvhip_ctx* ctx ) { return ctx ? ctx->device : -1; }

int vvhip_make_current( vvhip_ctx* ctx )
{
  if( !ctx ) return VVHIP_E_ARG;
  VVHIP_CHECK_HIP( ctx, hipSetDevice( ctx->device ) );
  return VVHIP_OK;
}

int vvhip_copy_peer( vvhip_ctx* dst_ctx, void* d_dst, vvhip_ctx* src_ctx, const void* d_src, size_t bytes )
{
  if( !dst_ctx || !src_ctx || !d_dst || !d_src ) return VVHIP_E_ARG;
  if( !bytes ) return VVHIP_OK;
  // order the copy after what the source context has queued (the producer of d_src), then run it on the destination's stream
  hipEvent_t ev = nullptr;
  int before = -1;
  ( void ) hipGetDevice( &before );                      // the calling thread's current device is restored on every path (the shim caches it per thread)
  VVHIP_CHECK_HIP( dst_ctx, hipSetDevice( src_ctx->device ) );
  hipError_t e = hipEventCreateWithFlags( &ev, hipEventDisableTiming );
  if( e == hipSuccess ) e = hipEventRecord( ev, src_ctx->stream );
  if( e == hipSuccess ) e = hipSetDevice( dst_ctx->device );
  if( e == hipSuccess ) e = hipStreamWaitEvent( dst_ctx->stream, ev, 0 );
  if( e == hipSuccess ) e = hipMemcpyPeerAsync( d_dst, dst_ctx->device, d_src, src_ctx->device, bytes, dst_ctx->stream );
  if( ev ) ( void ) hipEventDestroy( ev );
  if( before >= 0 ) ( void ) hipSetDevice( before );
  if( e != hipSuccess ) return vvhip_fail( dst_ctx, VVHIP_E_HIP, "vvhip_copy_peer: %s", hipGetErrorString( e ) );
  return VVHIP_OK;
}

} // extern "C"
