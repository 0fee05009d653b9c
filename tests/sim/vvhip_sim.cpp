// vvhip_sim.cpp — TEST DOUBLE of libvvenc_hip.so for the CPU test tier.  TEST INFRASTRUCTURE ONLY.
//
// Exports the entry points of include/vvenc_hip.h that the host side (vvenc_amd/csrc/host shim, bindings/vvenc) calls, with "device"
// memory = host memory and every kernel replaced by the CPU oracle (oracle/vvenc_oracle.c).  It exists so that the HOST LOGIC above the
// C ABI — staging, picture residency, batching and replay in the encoder binding, per-thread contexts, picture -> device mapping — can be
// exercised without a GPU (pytest -m "not gpu": the real reference encoder runs through shim + binding + this double and must emit the
// CPU encoder's bitstream).  It proves nothing about the HIP kernels; those are compared with the oracle by the -m gpu tests.
//
// It is never linked into or loaded by the product: tests inject it with LD_PRELOAD (symbol interposition over libvvenc_hip.so), and
// vvhip_create refuses to run when a real HIP device is visible unless VVHIP_SIM_FORCE=1, so it cannot stand in for the GPU by accident.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <unistd.h>

#include "../../include/vvenc_hip.h"
#include "../../oracle/vvenc_oracle.h"

struct vvhip_ctx { int device; std::string lastError; };
static std::string g_createError;
static std::atomic<long> g_calls{ 0 }, g_uploadBytes{ 0 }, g_downloadBytes{ 0 }, g_ctxCreated{ 0 };

static int fail( vvhip_ctx* ctx, int code, const char* fmt, ... )
{
  char buf[512]; va_list ap; va_start( ap, fmt ); vsnprintf( buf, sizeof( buf ), fmt, ap ); va_end( ap );
  ( ctx ? ctx->lastError : g_createError ) = buf;
  return code;
}
#define UNSUPPORTED( name ) return fail( ctx, VVHIP_E_UNSUPPORTED, "vvhip_sim: %s is not provided by the test double", name )

extern "C" {

// counters for the host-logic tests: [0] kernel-type calls, [1] bytes uploaded, [2] bytes downloaded, [3] contexts created
VVHIP_API void vvhip_sim_counters( long* out4 ) { out4[0] = g_calls; out4[1] = g_uploadBytes; out4[2] = g_downloadBytes; out4[3] = g_ctxCreated; }
VVHIP_API int  vvhip_sim_device_count() { const char* e = getenv( "VVHIP_SIM_DEVICES" ); return e ? atoi( e ) : 1; }

int vvhip_create( vvhip_ctx** out, int device )
{
  if( !out ) return fail( nullptr, VVHIP_E_ARG, "vvhip_create: out == NULL" );
  *out = nullptr;
  if( device < 0 || device >= vvhip_sim_device_count() ) return fail( nullptr, VVHIP_E_ARG, "vvhip_create: device %d out of range (%d simulated devices)", device, vvhip_sim_device_count() );
  const char* f = getenv( "VVHIP_SIM_FORCE" );
  if( !( f && atoi( f ) ) && ( access( "/dev/kfd", 0 ) == 0 ) )
    return fail( nullptr, VVHIP_E_HIP, "vvhip_sim: a GPU is present on this host; the test double only runs on GPU-less hosts (VVHIP_SIM_FORCE=1 overrides)" );
  *out = new vvhip_ctx{ device, "" };
  g_ctxCreated++;
  return VVHIP_OK;
}
void        vvhip_destroy( vvhip_ctx* ctx ) { delete ctx; }
const char* vvhip_last_error( const vvhip_ctx* ctx ) { return ctx ? ctx->lastError.c_str() : g_createError.c_str(); }
int         vvhip_set_stream( vvhip_ctx* ctx, void* ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_use_own_stream( vvhip_ctx* ctx ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
void*       vvhip_get_stream( vvhip_ctx* ) { return nullptr; }
int         vvhip_sync( vvhip_ctx* ctx ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_sync_all_devices( vvhip_ctx* ctx ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_set_blocking_sync( vvhip_ctx* ctx, int ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_device_count( void ) { return vvhip_sim_device_count(); }
int         vvhip_get_device( const vvhip_ctx* ctx ) { return ctx ? ctx->device : -1; }
int         vvhip_make_current( vvhip_ctx* ctx ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_graph_begin( vvhip_ctx* ctx ) { UNSUPPORTED( "vvhip_graph_begin" ); }
int         vvhip_graph_end( vvhip_ctx* ctx, vvhip_graph** ) { UNSUPPORTED( "vvhip_graph_end" ); }
int         vvhip_graph_launch( vvhip_ctx* ctx, vvhip_graph* ) { UNSUPPORTED( "vvhip_graph_launch" ); }
void        vvhip_graph_destroy( vvhip_graph* ) {}
int         vvhip_malloc( vvhip_ctx* ctx, void** p, size_t bytes ) { if( !ctx || !p ) return VVHIP_E_ARG; *p = malloc( bytes ? bytes : 1 ); return *p ? VVHIP_OK : VVHIP_E_NOMEM; }
int         vvhip_free( vvhip_ctx* ctx, void* p ) { if( !ctx ) return VVHIP_E_ARG; free( p ); return VVHIP_OK; }
int         vvhip_upload( vvhip_ctx* ctx, void* d, const void* h, size_t bytes ) { if( !ctx ) return VVHIP_E_ARG; memcpy( d, h, bytes ); g_uploadBytes += ( long ) bytes; return VVHIP_OK; }
int         vvhip_download( vvhip_ctx* ctx, void* h, const void* d, size_t bytes ) { if( !ctx ) return VVHIP_E_ARG; memcpy( h, d, bytes ); g_downloadBytes += ( long ) bytes; return VVHIP_OK; }
int         vvhip_download_async( vvhip_ctx* ctx, void* h, const void* d, size_t bytes ) { return vvhip_download( ctx, h, d, bytes ); }
int         vvhip_upload_2d( vvhip_ctx* ctx, void* d, size_t dpitch, const void* h, size_t spitch, size_t widthBytes, size_t rows )
{
  if( !ctx ) return VVHIP_E_ARG;
  for( size_t r = 0; r < rows; r++ ) memcpy( ( char* ) d + r * dpitch, ( const char* ) h + r * spitch, widthBytes );
  g_uploadBytes += ( long ) ( widthBytes * rows );
  return VVHIP_OK;
}
int         vvhip_download_2d( vvhip_ctx* ctx, void* h, size_t dpitch, const void* d, size_t spitch, size_t widthBytes, size_t rows )
{
  if( !ctx ) return VVHIP_E_ARG;
  for( size_t r = 0; r < rows; r++ ) memcpy( ( char* ) h + r * dpitch, ( const char* ) d + r * spitch, widthBytes );
  g_downloadBytes += ( long ) ( widthBytes * rows );
  return VVHIP_OK;
}
int         vvhip_copy_peer( vvhip_ctx* dst, void* d_dst, vvhip_ctx* src, const void* d_src, size_t bytes ) { if( !dst || !src ) return VVHIP_E_ARG; memcpy( d_dst, d_src, bytes ); return VVHIP_OK; }
int         vvhip_host_register( vvhip_ctx* ctx, const void*, size_t ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_host_unregister( vvhip_ctx* ctx, const void* ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_host_alloc( vvhip_ctx* ctx, void** p, size_t bytes ) { if( !ctx || !p ) return VVHIP_E_ARG; *p = malloc( bytes ? bytes : 1 ); return *p ? VVHIP_OK : VVHIP_E_NOMEM; }
int         vvhip_host_free( vvhip_ctx* ctx, void* p ) { if( !ctx ) return VVHIP_E_ARG; free( p ); return VVHIP_OK; }
int         vvhip_event_create( vvhip_ctx* ctx, void** e ) { if( !ctx || !e ) return VVHIP_E_ARG; *e = malloc( 1 ); return *e ? VVHIP_OK : VVHIP_E_NOMEM; }      // (every call of the sim is complete when it returns)
int         vvhip_event_record( vvhip_ctx* ctx, void* e ) { return ctx && e ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_event_wait( vvhip_ctx* ctx, void* e ) { return ctx && e ? VVHIP_OK : VVHIP_E_ARG; }
int         vvhip_event_destroy( vvhip_ctx* ctx, void* e ) { if( !ctx ) return VVHIP_E_ARG; free( e ); return VVHIP_OK; }
const char* vvhip_version( void ) { return "vvenc_hip SIMULATED (CPU oracle test double; not a product build)"; }

// ------------------------------------------------------------------------------------------------ (A) distortion
int vvhip_dist_batch( vvhip_ctx* ctx, int func, const int16_t* o, int os, const int16_t* c, int cs, int w, int h, int ss, int, const vvhip_dist_item* it, int n, uint64_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ )
  {
    const int16_t* po = o + it[i].org_off; const int16_t* pc = c + it[i].cur_off;
    switch( func )
    {
    case VVHIP_DF_SSE:      out[i] = orc_sse( po, os, pc, cs, w, h ); break;
    case VVHIP_DF_SAD:      out[i] = orc_sad( po, os, pc, cs, w, h, ss ); break;
    case VVHIP_DF_HAD:      out[i] = orc_had( po, os, pc, cs, w, h, 0 ); break;
    case VVHIP_DF_HAD_FAST: out[i] = orc_had( po, os, pc, cs, w, h, 1 ); break;
    case VVHIP_DF_HAD_2SAD:
    {
      const uint64_t hd = orc_had( po, os, pc, cs, w, h, 0 ), sd = 2 * orc_sad( po, os, pc, cs, w, h, 0 );
      out[i] = hd < sd ? hd : sd; break;
    }
    default: return fail( ctx, VVHIP_E_ARG, "vvhip_dist_batch: unknown function %d", func );
    }
  }
  return VVHIP_OK;
}
int vvhip_dist_multi_func( vvhip_ctx* ctx, const int16_t* o, int os, const int16_t* c, int cs, int bd, const vvhip_dist_fjob* jobs, int n )
{
  for( int i = 0; i < n; i++ ) { const int rc = vvhip_dist_batch( ctx, jobs[i].func, o, os, c, cs, jobs[i].width, jobs[i].height, jobs[i].sub_shift, bd, jobs[i].d_items, jobs[i].n, jobs[i].d_out ); if( rc ) return rc; }
  return VVHIP_OK;
}
int vvhip_plane_shift1( vvhip_ctx* ctx, const int16_t* s, size_t n, int16_t* d ) { if( !ctx ) return VVHIP_E_ARG; for( size_t i = 0; i < n; i++ ) d[i] = i + 1 < n ? s[i + 1] : ( int16_t ) 0; return VVHIP_OK; }
int vvhip_planes_derive( vvhip_ctx* ctx, const int16_t* ob, int os, int orows, int16_t* ot, const int16_t* cb, int cs, int crows, int16_t* ct, int16_t* sh )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( ot ) vvhip_plane_tile8( ctx, ob, os, orows, ot );
  if( ct ) vvhip_plane_tile8( ctx, cb, cs, crows, ct );
  if( sh ) vvhip_plane_shift1( ctx, cb, ( size_t ) cs * crows, sh );
  return VVHIP_OK;
}
size_t vvhip_tiled8_elems( int stride, int rows ) { return ( size_t ) ( ( rows + 7 ) / 8 ) * ( ( stride + 7 ) / 8 ) * 64 + 128; }
int vvhip_plane_tile8( vvhip_ctx* ctx, const int16_t* base, int stride, int rows, int16_t* tiled )
{
  if( !ctx ) return VVHIP_E_ARG;
  const int tpr = ( stride + 7 ) / 8;
  for( int y = 0; y < rows; y++ ) for( int x = 0; x < stride; x++ ) tiled[( ( size_t ) ( y >> 3 ) * tpr + ( x >> 3 ) ) * 64 + ( y & 7 ) * 8 + ( x & 7 )] = base[( size_t ) y * stride + x];
  return VVHIP_OK;
}
int vvhip_dist_multi_func_tiled( vvhip_ctx* ctx, const int16_t* o, int os, const int16_t* c, int cs, const vvhip_tiled_planes*, int bd, const vvhip_dist_fjob* jobs, int n )
{ return vvhip_dist_multi_func( ctx, o, os, c, cs, bd, jobs, n ); }      // (the layout is a device-side matter: same results by contract)
int vvhip_dist_multi( vvhip_ctx* ctx, int func, const int16_t* o, int os, const int16_t* c, int cs, int bd, const vvhip_dist_job* jobs, int n )
{
  for( int i = 0; i < n; i++ ) { const int rc = vvhip_dist_batch( ctx, func, o, os, c, cs, jobs[i].width, jobs[i].height, jobs[i].sub_shift, bd, jobs[i].d_items, jobs[i].n, jobs[i].d_out ); if( rc ) return rc; }
  return VVHIP_OK;
}
int vvhip_sad_x5_batch( vvhip_ctx* ctx, const int16_t* o, int os, const int16_t* c, int cs, int w, int h, int ss, int centre, const vvhip_dist_item* it, int n, uint64_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ ) orc_sad_x5( o + it[i].org_off, os, c + it[i].cur_off, cs, w, h, ss, out + 5 * i, centre );
  return VVHIP_OK;
}
int vvhip_sad_mask_batch( vvhip_ctx* ctx, const int16_t* o, int os, const int16_t* c, int cs, const int16_t* m, int ms, int stepX, int ms2, int w, int h, int ss, int,
                          const vvhip_dist_item* it, const int32_t* mo, int n, uint64_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ ) out[i] = orc_sad_mask( o + it[i].org_off, os, c + it[i].cur_off, cs, m + ( mo ? mo[i] : 0 ), ms, stepX, ms2, w, h, ss );
  return VVHIP_OK;
}
int vvhip_fix_weighted_sse_batch( vvhip_ctx* ctx, const int16_t* o, int os, const int16_t* c, int cs, int w, int h, int, const vvhip_dist_item* it, const uint32_t* wt, int n, uint64_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ ) out[i] = orc_fix_weighted_sse( o + it[i].org_off, os, c + it[i].cur_off, cs, w, h, wt[i] );
  return VVHIP_OK;
}
int vvhip_sad_surface( vvhip_ctx* ctx, const int16_t* o, int os, const int16_t* r, int rs, int w, int h, int ss, int rx, int ry, const int32_t* bo, const int32_t* br, int nb, uint32_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  const int nx = 2 * rx + 1, ny = 2 * ry + 1;
  for( int b = 0; b < nb; b++ ) for( int dy = -ry; dy <= ry; dy++ ) for( int dx = -rx; dx <= rx; dx++ )
    out[( ( size_t ) b * ny + dy + ry ) * nx + dx + rx] = ( uint32_t ) orc_sad( o + bo[b], os, r + br[b] + dy * rs + dx, rs, w, h, ss );
  return VVHIP_OK;
}

// ------------------------------------------------------------------------------------------------ (B) transforms + quantisation
int vvhip_fwd_transform_batch( vvhip_ctx* ctx, const int16_t* resi, int rs, const int32_t* off, int n, int w, int h, int th, int tv, int bd, int32_t* coef )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ ) if( orc_xT( resi + off[i], rs, coef + ( size_t ) i * w * h, w, h, th, tv, bd ) ) return fail( ctx, VVHIP_E_ARG, "fwd transform %dx%d types %d/%d", w, h, th, tv );
  return VVHIP_OK;
}
int vvhip_inv_transform_batch( vvhip_ctx* ctx, const int32_t* coef, int n, int w, int h, int th, int tv, int bd, int16_t* resi, int rs, const int32_t* off )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ ) if( orc_xIT( coef + ( size_t ) i * w * h, resi + off[i], rs, w, h, th, tv, bd ) ) return fail( ctx, VVHIP_E_ARG, "inv transform %dx%d types %d/%d", w, h, th, tv );
  return VVHIP_OK;
}
int vvhip_quant_core( vvhip_ctx* ctx, const int32_t* coef, int w, int h, int qc, int qbits, int64_t add, int thr, int16_t* lev, int32_t* du, int32_t* absSum, int32_t* last )
{
  return vvhip_quant_core_lfnst( ctx, coef, w, h, qc, qbits, add, thr, 0, lev, du, absSum, last );
}
int vvhip_quant_core_lfnst( vvhip_ctx* ctx, const int32_t* coef, int w, int h, int qc, int qbits, int64_t add, int thr, int lfnstIdx, int16_t* lev, int32_t* du, int32_t* absSum, int32_t* last )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  std::vector<int32_t> d( ( size_t ) w * h );
  int l = -1; int32_t s = 0;
  orc_quant_core_lfnst( coef, lev, du ? du : d.data(), w, h, qc, qbits, add, thr, lfnstIdx, &s, &l );
  if( absSum ) *absSum = s;
  if( last ) *last = l;
  return VVHIP_OK;
}
int vvhip_quant_batch( vvhip_ctx* ctx, const int32_t* coef, int n, int w, int h, int bd, const vvhip_tu_qp* qp, int thr, int16_t* lev, int32_t* du, int32_t* absSum, int32_t* last )
{
  for( int i = 0; i < n; i++ )
  {
    int qc, qbits; int64_t add;
    orc_quant_params( w, h, bd, qp[i].qp, qp[i].flags & 1, &qc, &qbits, &add );
    const int rc = vvhip_quant_core( ctx, coef + ( size_t ) i * w * h, w, h, qc, qbits, add, thr, lev + ( size_t ) i * w * h, du ? du + ( size_t ) i * w * h : nullptr, absSum + i, last + i );
    if( rc ) return rc;
  }
  return VVHIP_OK;
}
int vvhip_dequant_core( vvhip_ctx* ctx, int maxX, int maxY, int scale, const int16_t* lev, size_t ls, int32_t* coef, int rs, int imax, int32_t tmax )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  orc_dequant_core( maxX, maxY, scale, lev, ls, coef, rs, imax, tmax );
  return VVHIP_OK;
}
int vvhip_dequant_batch( vvhip_ctx* ctx, const int16_t* lev, int n, int w, int h, int bd, const vvhip_tu_qp* qp, int32_t* coef )
{
  for( int i = 0; i < n; i++ )
  {
    int scale, rs, imax;
    orc_dequant_params( w, h, bd, qp[i].qp, &scale, &rs, &imax );
    memset( coef + ( size_t ) i * w * h, 0, sizeof( int32_t ) * w * h );
    const int rc = vvhip_dequant_core( ctx, w - 1, h - 1, scale, lev + ( size_t ) i * w * h, w, coef + ( size_t ) i * w * h, rs, imax, 32767 );
    if( rc ) return rc;
  }
  return VVHIP_OK;
}
int vvhip_need_rdoq_core( vvhip_ctx* ctx, const int32_t* coef, size_t num, int qc, int64_t off, int shift, uint8_t* need )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  *need = ( uint8_t ) orc_need_rdoq( coef, num, qc, off, shift );
  return VVHIP_OK;
}
int vvhip_need_rdoq_batch( vvhip_ctx* ctx, const int32_t* coef, int n, int w, int h, int bd, const vvhip_tu_qp* qp, uint8_t* need )
{
  for( int i = 0; i < n; i++ )
  {
    int qc, qbits, num; int64_t add;
    orc_need_rdoq_params( w, h, bd, qp[i].qp, ( qp[i].flags >> 1 ) & 1, &qc, &qbits, &add, &num );
    const int rc = vvhip_need_rdoq_core( ctx, coef + ( size_t ) i * w * h, num, qc, add, qbits, need + i );
    if( rc ) return rc;
  }
  return VVHIP_OK;
}
int vvhip_tu_rdo_batch( vvhip_ctx* ctx, const int16_t* resi, int rs, const int32_t* off, int n, int w, int h, int th, int tv, int bd, const vvhip_tu_qp* qp, int thr,
                        int16_t* lev, int16_t* rec, vvhip_tu_stats* st )
{
  if( !ctx ) return VVHIP_E_ARG;
  std::vector<int32_t> coef( ( size_t ) w * h ), deq( ( size_t ) w * h );
  std::vector<int16_t> l( ( size_t ) w * h ), r( ( size_t ) w * h );
  for( int i = 0; i < n; i++ )
  {
    const int32_t o = off[i], zero = 0;
    int rc = vvhip_fwd_transform_batch( ctx, resi + o, rs, &zero, 1, w, h, th, tv, bd, coef.data() ); if( rc ) return rc;
    uint8_t need = 0; int32_t absSum = 0, last = -1;
    rc = vvhip_need_rdoq_batch( ctx, coef.data(), 1, w, h, bd, qp + i, &need ); if( rc ) return rc;
    rc = vvhip_quant_batch( ctx, coef.data(), 1, w, h, bd, qp + i, thr, l.data(), nullptr, &absSum, &last ); if( rc ) return rc;
    rc = vvhip_dequant_batch( ctx, l.data(), 1, w, h, bd, qp + i, deq.data() ); if( rc ) return rc;
    rc = vvhip_inv_transform_batch( ctx, deq.data(), 1, w, h, th, tv, bd, r.data(), w, &zero ); if( rc ) return rc;
    uint64_t sse = 0;
    for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ ) { const int64_t d = ( int64_t ) resi[o + y * rs + x] - r[y * w + x]; sse += ( uint64_t ) ( d * d ); }
    if( lev ) memcpy( lev + ( size_t ) i * w * h, l.data(), sizeof( int16_t ) * w * h );
    if( rec ) memcpy( rec + ( size_t ) i * w * h, r.data(), sizeof( int16_t ) * w * h );
    if( st ) { st[i].abs_sum = absSum; st[i].last_scan_pos = last; st[i].need_rdoq = need; st[i].pad = 0; st[i].sse = sse; }
  }
  return VVHIP_OK;
}
int vvhip_tu_rdo_multi( vvhip_ctx* ctx, const int16_t* resi, int rs, int bd, const vvhip_tu_job* j, int n )
{
  for( int i = 0; i < n; i++ )
  {
    const int rc = vvhip_tu_rdo_batch( ctx, resi, rs, j[i].d_resi_off, j[i].n, j[i].width, j[i].height, j[i].tr_hor, j[i].tr_ver, bd, j[i].d_qp, j[i].thr_val, j[i].d_level, j[i].d_rec_resi, j[i].d_stats );
    if( rc ) return rc;
  }
  return VVHIP_OK;
}
int vvhip_fast_fwd_core( vvhip_ctx* ctx, int trSize, const int16_t* tc, const int32_t* src, int32_t* dst, unsigned line, unsigned red, unsigned cut, int shift )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_fast_fwd_core( trSize, tc, src, dst, line, red, cut, shift ); return VVHIP_OK; }
int vvhip_fast_inv_core( vvhip_ctx* ctx, int trSize, const int16_t* it, const int32_t* src, int32_t* dst, unsigned lines, unsigned red, unsigned rows )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_fast_inv_core( trSize, it, src, dst, lines, red, rows ); return VVHIP_OK; }
int vvhip_round_clip( vvhip_ctx* ctx, int32_t* dst, unsigned w, unsigned h, unsigned stride, int32_t mn, int32_t mx, int32_t round, int32_t shift )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_round_clip( dst, w, h, stride, mn, mx, round, shift ); return VVHIP_OK; }
int vvhip_cpy_resi( vvhip_ctx* ctx, const int32_t* src, int16_t* dst, ptrdiff_t stride, unsigned w, unsigned h )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_cpy_resi( src, dst, stride, w, h ); return VVHIP_OK; }
int vvhip_cpy_coeff( vvhip_ctx* ctx, const int16_t* src, ptrdiff_t stride, int32_t* dst, unsigned w, unsigned h )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_cpy_coeff( src, stride, dst, w, h ); return VVHIP_OK; }
int vvhip_get_tr_matrix_host( int type, int l2, int16_t* out ) { return orc_tr_matrix( type, l2, out ) ? VVHIP_E_ARG : VVHIP_OK; }
int vvhip_get_scan_order_host( int lw, int lh, uint32_t* out ) { orc_scan_order( lw, lh, out ); return VVHIP_OK; }

// ------------------------------------------------------------------------------------------------ interpolation
int vvhip_if_filter( vvhip_ctx* ctx, int taps, int vert, int first, int last, int bd, const int16_t* src, int ss, int16_t* dst, int ds, int w, int h, const int16_t* coeff )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_if_filter( taps, vert, first, last, bd, src, ss, dst, ds, w, h, coeff ); return VVHIP_OK; }
int vvhip_if_copy( vvhip_ctx* ctx, int first, int last, int bd, const int16_t* src, int ss, int16_t* dst, int ds, int w, int h, int bi )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_if_copy( first, last, bd, src, ss, dst, ds, w, h, bi ); return VVHIP_OK; }

static void predOne( const int16_t* ref, int rs, int w, int h, int fx, int fy, int rnd, int bd, int mode, int alt, int16_t* out )
{
  if( mode == 0 ) { orc_if_pred_luma( ref, rs, out, w, w, h, fx, fy, rnd, bd, alt ); return; }
  // the two passes of the fast sub-pel search (InterSearch.cpp:818-848): horizontal isFirst over h + 7 rows, vertical isLast
  const int rows = h + 7;
  std::vector<int16_t> tmp( ( size_t ) rows * w );
  orc_if_luma_1d( 0, ref - 3 * rs, rs, tmp.data(), w, w, rows, fx, 1, 0, bd, alt, mode );
  orc_if_luma_1d( 1, tmp.data() + 3 * w, w, out, w, w, h, fy, 0, 1, bd, alt, mode );
}
int vvhip_interp_luma_batch( vvhip_ctx* ctx, const int16_t* ref, int rs, const vvhip_subpel_item* it, int n, int w, int h, int bd, int rnd, int mode, int alt, int16_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ ) predOne( ref + it[i].ref_off, rs, w, h, it[i].frac_x, it[i].frac_y, rnd, bd, mode, alt, out + ( size_t ) i * w * h );
  return VVHIP_OK;
}
int vvhip_subpel_dist_batch( vvhip_ctx* ctx, int func, const int16_t* org, int os, const int16_t* ref, int rs, int w, int h, int bd, int mode, int alt, const vvhip_subpel_item* it, int n, uint64_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  std::vector<int16_t> pred( ( size_t ) w * h );
  for( int i = 0; i < n; i++ )
  {
    predOne( ref + it[i].ref_off, rs, w, h, it[i].frac_x, it[i].frac_y, 1, bd, mode, alt, pred.data() );
    const vvhip_dist_item di = { it[i].org_off, 0 };
    const int rc = vvhip_dist_batch( ctx, func, org, os, pred.data(), w, w, h, 0, bd, &di, 1, out + i ); if( rc ) return rc;
  }
  return VVHIP_OK;
}
int vvhip_subpel_refine_batch( vvhip_ctx* ctx, int func, const int16_t* org, int os, const int16_t* ref, int rs, int w, int h, int bd, int mode, int alt,
                               const vvhip_subpel_item* bases, int nb, const int16_t* offs, int no, uint64_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( ( w & 7 ) || w > 64 || h > 64 || no < 1 || no > 16 ) return fail( ctx, VVHIP_E_ARG, "vvhip_subpel_refine_batch: block %dx%d, %d offsets", w, h, no );
  std::vector<vvhip_subpel_item> c( ( size_t ) nb * no );
  for( int b = 0; b < nb; b++ ) for( int k = 0; k < no; k++ )
  {
    const int tx = ( bases[b].frac_x & 15 ) + offs[2 * k], ty = ( bases[b].frac_y & 15 ) + offs[2 * k + 1];
    vvhip_subpel_item& q = c[( size_t ) b * no + k];
    q.org_off = bases[b].org_off; q.ref_off = bases[b].ref_off + ( ty >> 4 ) * rs + ( tx >> 4 ); q.frac_x = ( int16_t ) ( tx & 15 ); q.frac_y = ( int16_t ) ( ty & 15 );
  }
  return vvhip_subpel_dist_batch( ctx, func, org, os, ref, rs, w, h, bd, mode, alt, c.data(), nb * no, out );
}

// ------------------------------------------------------------------------------------------------ (C) MCTF
int vvhip_mctf_error_batch( vvhip_ctx* ctx, const int16_t* org, int os, const int16_t* buf, int bs, int w, int h, int tap4, int bd, const vvhip_mctf_item* it, int n, int32_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ )
    out[i] = ( it[i].fx | it[i].fy ) ? orc_mctf_err_frac( tap4, org + it[i].org_off, os, buf + it[i].buf_off, bs, w, h, it[i].fx, it[i].fy, bd )
                                     : orc_mctf_err_int( org + it[i].org_off, os, buf + it[i].buf_off, bs, w, h );
  return VVHIP_OK;
}
int vvhip_mctf_calc_var_batch( vvhip_ctx* ctx, const int16_t* org, int os, int w, int h, const int32_t* off, int n, int64_t* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ ) out[i] = ( int64_t ) llround( orc_mctf_calc_var( org + off[i], os, w, h ) * 256.0 );
  return VVHIP_OK;
}
int vvhip_mctf_subsample( vvhip_ctx* ctx, const int16_t* src, int ss, int sw, int sh, int16_t* dst, int ds, int pad )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  orc_mctf_subsample( src, ss, sw, sh, dst, ds ); orc_extend_border( dst, ds, sw / 2, sh / 2, pad );
  return VVHIP_OK;
}
int vvhip_extend_border( vvhip_ctx* ctx, int16_t* p, int stride, int w, int h, int pad ) { if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_extend_border( p, stride, w, h, pad ); return VVHIP_OK; }
int vvhip_mctf_init_mvs( vvhip_ctx* ctx, vvhip_mv*, int ) { UNSUPPORTED( "vvhip_mctf_init_mvs" ); }
int vvhip_mctf_me_level( vvhip_ctx* ctx, const int16_t*, int, const int16_t*, int, int, int, int, const vvhip_mv*, int, int, int, int, int, int, int, int, vvhip_mv*, int, int ) { UNSUPPORTED( "vvhip_mctf_me_level" ); }
int vvhip_mctf_motion_estimation( vvhip_ctx* ctx, const int16_t* cur, const int16_t* const* refs, int nRefs, int stride, int w, int h, int, int bd, int unit, int speed, int addLevel, vvhip_mv* const* outs )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  std::vector<int16_t> o( ( size_t ) w * h ), r( ( size_t ) w * h );
  for( int y = 0; y < h; y++ ) memcpy( &o[( size_t ) y * w], cur + ( ptrdiff_t ) y * stride, sizeof( int16_t ) * w );
  const int mw = ( w + unit - 1 ) / unit, mh = ( h + unit - 1 ) / unit;
  for( int k = 0; k < nRefs; k++ )
  {
    for( int y = 0; y < h; y++ ) memcpy( &r[( size_t ) y * w], refs[k] + ( ptrdiff_t ) y * stride, sizeof( int16_t ) * w );
    std::vector<orc_mv_t> fin( ( size_t ) mw * mh );
    orc_mv_t* lv[5] = { nullptr, nullptr, nullptr, nullptr, fin.data() };
    int dims[10];
    orc_mctf_me( o.data(), r.data(), w, h, bd, unit, speed, addLevel, lv, dims );
    static_assert( sizeof( orc_mv_t ) == sizeof( vvhip_mv ), "motion vector layout" );
    memcpy( outs[k], fin.data(), sizeof( vvhip_mv ) * fin.size() );
  }
  return VVHIP_OK;
}
int vvhip_mctf_motion_estimation_async( vvhip_ctx* ctx, const int16_t* cur, const int16_t* const* refs, int nRefs, int stride, int w, int h, int pad, int bd, int unit, int speed, int addLevel, vvhip_mv* const* outs )
{
  return vvhip_mctf_motion_estimation( ctx, cur, refs, nRefs, stride, w, h, pad, bd, unit, speed, addLevel, outs );      // (the test double has no streams: every call is complete on return)
}
int vvhip_tu_set_sparse_outputs( vvhip_ctx* ctx, int ) { return ctx ? VVHIP_OK : VVHIP_E_ARG; }      // (the test double always writes every output: a valid "unspecified")
int vvhip_mctf_set_stats( vvhip_ctx* ctx, int ) { UNSUPPORTED( "vvhip_mctf_set_stats" ); }
int vvhip_mctf_get_stats( vvhip_ctx* ctx, uint64_t* ) { UNSUPPORTED( "vvhip_mctf_get_stats" ); }
int vvhip_mctf_set_timing( vvhip_ctx* ctx, int ) { UNSUPPORTED( "vvhip_mctf_set_timing" ); }
int vvhip_mctf_last_times( vvhip_ctx* ctx, float* ) { UNSUPPORTED( "vvhip_mctf_last_times" ); }
int vvhip_mctf_filter_params( int qp, int bd, double strength, int chroma, double* sigmaSq, double* weightScaling )
{
  const double lumaSigmaSq = 9.0 * ( 128.0 + 3.0 / 256.0 * qp * qp * qp ), chromaSigmaSq = 30 * 30;       // MCTF.cpp:1491-1492 (m_sigmaMultiplier 9)
  const double bdw = 1024.0 / ( 1 << bd );
  *sigmaSq = ( chroma ? chromaSigmaSq : lumaSigmaSq ) / ( bdw * bdw );
  *weightScaling = strength * ( chroma ? 0.55 : 0.4 );
  return VVHIP_OK;
}
int vvhip_mctf_apply_plane( vvhip_ctx* ctx, const int16_t* org, int os, int w, int h, int cs, int bd, int unit, int lowRes, int qp, int nRefs, const int16_t* const* refs, int rs,
                            const vvhip_mv* const* mvs, int mvW, const double* strengths, double ws, double sigmaSq, int16_t* out, int outStride )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  orc_mctf_bilateral_plane( org, os, w, h, cs, bd, unit, lowRes, qp, nRefs, refs, rs, reinterpret_cast<const orc_mv_t* const*>( mvs ), mvW, strengths, ws, sigmaSq, out, outStride );
  return VVHIP_OK;
}

// ------------------------------------------------------------------------------------------------ DMVR
int vvhip_dmvr_refine_batch( vvhip_ctx* ctx, const int16_t* r0, int s0, const int16_t* r1, int s1, const vvhip_dmvr_item* it, int n, int dx, int dy, int bd, vvhip_dmvr_result* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  for( int i = 0; i < n; i++ )
  {
    int16_t mvd[2] = { 0, 0 };
    out[i].min_cost = orc_dmvr_refine( r0 + it[i].ref0_off, s0, it[i].frac0_x, it[i].frac0_y, r1 + it[i].ref1_off, s1, it[i].frac1_x, it[i].frac1_y, dx, dy, bd, mvd );
    out[i].mvd_x = mvd[0]; out[i].mvd_y = mvd[1]; out[i].pad = 0;
  }
  return VVHIP_OK;
}

// ------------------------------------------------------------------------------------------------ ALF
int vvhip_alf_classify( vvhip_ctx* ctx, const int16_t* rec, int stride, int w, int h, int bd, int vbH, int vbPos, uint8_t* cls )
{ if( !ctx ) return VVHIP_E_ARG; g_calls++; orc_alf_classify( rec, stride, w, h, bd + 4, vbH, vbPos, cls ); return VVHIP_OK; }
int vvhip_alf_stats_plane( vvhip_ctx* ctx, const int16_t* org, int os, const int16_t* rec, int rs, int w, int h, int ctu, int fl, const uint8_t* cls, int vbH, int vbPos, const float* init, float* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  if( init )
  {
    const size_t n = ( size_t ) ( ( w + ctu - 1 ) / ctu ) * ( ( h + ctu - 1 ) / ctu ) * ( cls ? 25 : 1 ) * ORC_ALF_REC;
    if( init != out ) memmove( out, init, n * sizeof( float ) );
    orc_alf_stats_plane_acc( org, os, rec, rs, w, h, ctu, fl, cls, vbH, vbPos, out );
  }
  else orc_alf_stats_plane( org, os, rec, rs, w, h, ctu, fl, cls, vbH, vbPos, out );
  return VVHIP_OK;
}
int vvhip_alf_stats_plane_units( vvhip_ctx* ctx, const int16_t* org, int os, const int16_t* rec, int rs, int w, int h, int unit, int ctu, int fl, const uint8_t* cls, int vbH, int vbPos, const float* init, float* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  if( init ) return fail( ctx, VVHIP_E_UNSUPPORTED, "vvhip_sim: vvhip_alf_stats_plane_units with start records" );
  g_calls++;
  orc_alf_stats_plane_units( org, os, rec, rs, w, h, unit, ctu, fl, cls, vbH, vbPos, out );
  return VVHIP_OK;
}
int vvhip_ccalf_stats_plane( vvhip_ctx* ctx, const int16_t* orgC, int os, const int16_t* slfC, int ss, const int16_t* recL, int rs, int wc, int hc, int ctuC, int sx, int sy, int vbH, int vbPos, int picH,
                             const float* init, float* out )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  const size_t n = ( size_t ) ( ( wc + ctuC - 1 ) / ctuC ) * ( ( hc + ctuC - 1 ) / ctuC ) * ORC_ALF_REC;
  if( init ) { if( init != out ) memmove( out, init, n * sizeof( float ) ); } else memset( out, 0, n * sizeof( float ) );
  orc_ccalf_stats_plane( orgC, os, slfC, ss, recL, rs, wc, hc, ctuC, sx, sy, vbH, vbPos, picH, out );
  return VVHIP_OK;
}
int vvhip_alf_filter_plane( vvhip_ctx* ctx, const int16_t* src, ptrdiff_t ss, int16_t* dst, ptrdiff_t ds, int w, int h, int ctu, int bd, int fl, const uint8_t* cls, const int16_t* coeff, const int16_t* clip,
                            const int16_t* ctuSet, int vbH, int vbPos )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  std::vector<int16_t> lin;
  if( !clip )
  {
    // linear entries: clipping values = the sample range (no clipping takes place)
    int maxSet = 0; const int nCtu = ( ( w + ctu - 1 ) / ctu ) * ( ( h + ctu - 1 ) / ctu );
    for( int i = 0; i < nCtu; i++ ) if( ctuSet[i] > maxSet ) maxSet = ctuSet[i];
    lin.assign( ( size_t ) ( maxSet + 1 ) * ( cls ? 25 : 1 ) * 13, ( int16_t ) ( 1 << bd ) );
    clip = lin.data();
  }
  orc_alf_filter_plane( src, ss, dst, ds, w, h, ctu, bd, fl, cls, coeff, clip, ctuSet, vbH, vbPos );
  return VVHIP_OK;
}
int vvhip_ccalf_filter_plane( vvhip_ctx* ctx, int16_t* dstC, ptrdiff_t ds, const int16_t* recL, ptrdiff_t rs, int wc, int hc, int ctuC, int sx, int sy, int bd, const int16_t* coeff, const uint8_t* ctuFilter, int vbH, int vbPos )
{
  if( !ctx ) return VVHIP_E_ARG;
  g_calls++;
  orc_ccalf_filter_plane( dstC, ds, recL, rs, wc, hc, ctuC, sx, sy, bd, coeff, ctuFilter, vbH, vbPos );
  return VVHIP_OK;
}

} // extern "C"
