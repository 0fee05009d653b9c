// vvenc_hip_shim.cpp — implementation of the table-shaped host mirror (see vvenc_hip_shim.h) on top of the C ABI only.
#include "vvenc_hip_shim.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

namespace vvhip {

static int ilog2( unsigned v ) { int l = 0; while( ( 1u << ( l + 1 ) ) <= v ) l++; return l; }

// ------------------------------------------------------------------------------------------------ Device
namespace {
std::atomic<uint64_t> g_upBytes{ 0 }, g_downBytes{ 0 }, g_ups{ 0 }, g_downs{ 0 }, g_contexts{ 0 }, g_refCalls{ 0 };
// every transfer of this translation unit is counted (Device::stats): the macros below route the C ABI's copy calls through these
// host ranges pinned in place (Device::pinHost): whole pages INSIDE a recycled picture buffer.  A copy whose host range starts inside a pinned range must end inside it
// (the runtime treats it as pinned as a whole), so an upload is cut at the borders of the pinned ranges it touches: the page-aligned body goes as asynchronous DMA, the
// sub-page head and tail as ordinary (staged) copies.
std::mutex g_pinLock;
std::map<uintptr_t, size_t> g_pinned;      // page-aligned base -> bytes
inline int countedUpload( vvhip_ctx* c, void* d, const void* h, size_t n )
{
  g_upBytes += n; g_ups++;
  uintptr_t s = reinterpret_cast<uintptr_t>( h ); const uintptr_t e = s + n;
  char* dst = static_cast<char*>( d );
  while( s < e )
  {
    uintptr_t cut = e;
    {
      std::lock_guard<std::mutex> g( g_pinLock );
      auto it = g_pinned.upper_bound( s );
      if( it != g_pinned.begin() && std::prev( it )->first + std::prev( it )->second > s ) cut = std::min( e, std::prev( it )->first + std::prev( it )->second );      // inside a pinned range: up to its end
      else if( it != g_pinned.end() && it->first < e ) cut = it->first;                                                                                              // pageable: up to the next pinned range
    }
    const int rc = vvhip_upload( c, dst, reinterpret_cast<const void*>( s ), cut - s );
    if( rc ) return rc;
    dst += cut - s; s = cut;
  }
  return VVHIP_OK;
}
inline int countedDownload( vvhip_ctx* c, void* h, const void* d, size_t n ) { g_downBytes += n; g_downs++; return vvhip_download( c, h, d, n ); }
inline int countedDownloadAsync( vvhip_ctx* c, void* h, const void* d, size_t n ) { g_downBytes += n; g_downs++; return vvhip_download_async( c, h, d, n ); }
#define vvhip_upload countedUpload
#define vvhip_download countedDownload
#define vvhip_download_async countedDownloadAsync

// mirrors of one GPU, shared by that GPU's contexts.  Readers get copies made under the lock (a slot is reused after unregisterPicture).
struct Registry { mutable std::mutex m; std::deque<Device::Mirror> mirrors; };
Registry& registry( int gpu )
{
  static std::mutex m; static std::map<int, std::unique_ptr<Registry>> regs;
  std::lock_guard<std::mutex> g( m );
  std::unique_ptr<Registry>& r = regs[gpu];
  if( !r ) r.reset( new Registry );
  return *r;
}
} // namespace

// contexts outlive their threads: a finished worker's context goes back to the pool and serves the next new thread (no HIP teardown at thread or process exit)
struct DevicePool
{
  std::mutex m; std::multimap<int, Device*> idle;
  Device* take( int gpu )
  {
    { std::lock_guard<std::mutex> g( m ); auto it = idle.find( gpu ); if( it != idle.end() ) { Device* d = it->second; idle.erase( it ); return d; } }
    return new Device( gpu );
  }
  void give( Device* d ) { std::lock_guard<std::mutex> g( m ); idle.emplace( d->gpu(), d ); }
  static DevicePool& get() { static DevicePool* p = new DevicePool; return *p; }      // leaked on purpose (see above)
};
namespace {
struct ThreadDevices
{
  int selected = -1; std::map<int, Device*> byGpu; int current = -2;
  ~ThreadDevices() { for( auto& kv : byGpu ) DevicePool::get().give( kv.second ); }
};
thread_local ThreadDevices t_dev;
}

int Device::defaultGpu() { static const int g = []{ const char* e = getenv( "VVHIP_DEVICE" ); return e ? atoi( e ) : 0; }(); return g; }
// $VVHIP_LOGICAL_GPUS = N: the shim presents N devices whatever the box has; logical device g lives on physical device g mod (physical count).  Every structure above the
// C ABI (per-device registries, thread -> device binding, device-to-device picture copies) then runs exactly as on N GPUs — how the sharded path is tested on a 1-GPU box.
static int logicalGpus() { static const int n = []{ const char* e = getenv( "VVHIP_LOGICAL_GPUS" ); return e ? atoi( e ) : 0; }(); return n; }
int Device::gpuCount() { return logicalGpus() > 0 ? logicalGpus() : vvhip_device_count(); }
void Device::selectGpu( int gpu ) { t_dev.selected = gpu; }
int Device::selectedGpu() { return t_dev.selected; }

Device& Device::get()
{
  ThreadDevices& t = t_dev;
  const int gpu = t.selected >= 0 ? t.selected : defaultGpu();
  Device*& d = t.byGpu[gpu];
  if( !d ) d = DevicePool::get().take( gpu );
  if( t.current != gpu ) { d->check( vvhip_make_current( d->m_ctx ), "vvhip_make_current" ); t.current = gpu; }
  return *d;
}

Device::Device( int gpu ) : m_gpu( gpu )
{
  const int phys = vvhip_device_count();
  const int rc = vvhip_create( &m_ctx, logicalGpus() > 0 && phys > 0 ? gpu % phys : gpu );
  if( rc != VVHIP_OK ) throw Exception( std::string( "vvhip::Device: " ) + vvhip_last_error( nullptr ) );
  // the shim's callers are an encoder's worker threads: while they wait for the device their core should go to another worker ($VVHIP_SYNC=spin: the runtime's busy wait)
  static const bool spin = []{ const char* e = getenv( "VVHIP_SYNC" ); return e && !strcmp( e, "spin" ); }();
  vvhip_set_blocking_sync( m_ctx, spin ? 0 : 1 );
  g_contexts++;
}
Device::~Device() {}

Device::Stats Device::stats() { return { g_upBytes.load(), g_downBytes.load(), g_ups.load(), g_downs.load(), g_contexts.load(), g_refCalls.load() }; }

void Device::check( int rc, const char* what ) const
{
  if( rc != VVHIP_OK ) throw Exception( std::string( what ) + ": " + vvhip_last_error( m_ctx ) );
}

bool Device::pinHost( const void* p, size_t bytes )
{
  static const bool on = []{ const char* e = getenv( "VVHIP_PIN" ); return !e || atoi( e ) != 0; }();
  if( !on || !p || bytes < ( 256u << 10 ) ) return false;      // picture-sized, recycled buffers only
  // whole pages inside the buffer: a neighbouring allocation never shares a pinned page
  const uintptr_t page = 4096, a = ( reinterpret_cast<uintptr_t>( p ) + page - 1 ) & ~( page - 1 ), e = ( reinterpret_cast<uintptr_t>( p ) + bytes ) & ~( page - 1 );
  if( e <= a ) return false;
  Device& dev = Device::get();
  std::lock_guard<std::mutex> g( g_pinLock );
  auto it = g_pinned.upper_bound( a );
  if( it != g_pinned.begin() ) { auto pr = std::prev( it ); if( pr->first <= a && pr->first + pr->second >= e ) return true; }       // already inside a pinned range
  // ranges that overlap the new one were pinned for an earlier incarnation of (part of) this buffer: drop them first — after every device has finished the copies other worker
  // contexts may still have in flight from them (uploads are issued outside this lock)
  bool synced = false;
  for( it = g_pinned.begin(); it != g_pinned.end(); )
  {
    if( !synced && it->first < e && it->first + it->second > a ) { vvhip_sync_all_devices( dev.ctx() ); synced = true; }
    if( it->first < e && it->first + it->second > a ) { vvhip_host_unregister( dev.ctx(), reinterpret_cast<void*>( it->first ) ); it = g_pinned.erase( it ); }
    else ++it;
  }
  if( vvhip_host_register( dev.ctx(), reinterpret_cast<void*>( a ), e - a ) != VVHIP_OK ) return false;
  g_pinned[a] = e - a;
  return true;
}

void Device::unpinAll()
{
  Device& dev = Device::get();
  std::lock_guard<std::mutex> g( g_pinLock );
  if( !g_pinned.empty() ) vvhip_sync_all_devices( dev.ctx() );
  for( auto& kv : g_pinned ) vvhip_host_unregister( dev.ctx(), reinterpret_cast<void*>( kv.first ) );
  g_pinned.clear();
}

Pel* PinnedBuffer::get( size_t elems )
{
  if( elems > m_elems )
  {
    Device& dev = Device::get();
    if( m_p ) { vvhip_sync( dev.ctx() ); vvhip_host_free( dev.ctx(), m_p ); m_p = nullptr; m_elems = 0; }
    void* p = nullptr;
    const size_t want = elems + elems / 4;
    dev.check( vvhip_host_alloc( dev.ctx(), &p, want * sizeof( Pel ) ), "pinned host area" );
    m_p = static_cast<Pel*>( p ); m_elems = want;
  }
  return m_p;
}

int Device::registerPicture( const Pel* origin, int stride, int width, int height, int margin, bool findable, bool upload )
{
  Mirror m;
  m.origin = origin; m.stride = stride; m.width = width; m.height = height; m.margin = margin; m.live = true; m.findable = findable; m.reference = false;
  m.hostBase = origin - ( ptrdiff_t ) margin * stride - margin;
  const size_t elems = ( size_t ) stride * ( height + 2 * margin );
  m.hostEnd = m.hostBase + elems;
  void* d = nullptr;
  check( vvhip_malloc( m_ctx, &d, elems * sizeof( Pel ) ), "registerPicture" );
  m.dBase = static_cast<int16_t*>( d );
  m.dOrigin = m.dBase + ( ptrdiff_t ) margin * stride + margin;
  Registry& r = registry( m_gpu );
  int id = -1;
  {
    std::lock_guard<std::mutex> g( r.m );
    for( size_t i = 0; i < r.mirrors.size(); i++ ) if( !r.mirrors[i].live ) { id = ( int ) i; break; }      // reuse the slot of an unregistered picture
    if( id < 0 ) { r.mirrors.push_back( m ); id = ( int ) r.mirrors.size() - 1; } else r.mirrors[id] = m;
  }
  if( upload ) updatePicture( id );
  return id;
}

Device::Mirror Device::mirror( int id ) const
{
  Registry& r = registry( m_gpu );
  std::lock_guard<std::mutex> g( r.m );
  return r.mirrors.at( id );
}

void Device::updatePicture( int id )
{
  const Mirror m = mirror( id );
  check( vvhip_upload( m_ctx, m.dBase, m.hostBase, ( size_t ) ( m.hostEnd - m.hostBase ) * sizeof( Pel ) ), "updatePicture" );
  check( vvhip_sync( m_ctx ), "updatePicture" );
}

void Device::updatePictureRows( int id, int y0, int rows )
{
  const Mirror m = mirror( id );
  if( y0 < -m.margin ) { rows -= -m.margin - y0; y0 = -m.margin; }
  if( y0 + rows > m.height + m.margin ) rows = m.height + m.margin - y0;
  if( rows <= 0 ) return;
  const ptrdiff_t off = ( ptrdiff_t ) ( y0 + m.margin ) * m.stride;
  check( vvhip_upload( m_ctx, m.dBase + off, m.hostBase + off, ( size_t ) rows * m.stride * sizeof( Pel ) ), "updatePictureRows" );
  check( vvhip_sync( m_ctx ), "updatePictureRows" );
}

void Device::unregisterPicture( int id )
{
  Registry& r = registry( m_gpu );
  int16_t* d = nullptr;
  {
    std::lock_guard<std::mutex> g( r.m );
    Mirror& m = r.mirrors.at( id );
    if( m.live ) { d = m.dBase; m.live = false; m.hostBase = m.hostEnd = nullptr; }
  }
  if( d ) vvhip_free( m_ctx, d );
}

int Device::copyMirrorTo( int id, Device& dst )
{
  const Mirror m = mirror( id );
  const int did = dst.registerPicture( m.origin, m.stride, m.width, m.height, m.margin, m.findable, false );
  const Mirror dm = dst.mirror( did );
  dst.check( vvhip_copy_peer( dst.ctx(), dm.dBase, m_ctx, m.dBase, ( size_t ) ( m.hostEnd - m.hostBase ) * sizeof( Pel ) ), "vvhip_copy_peer" );
  dst.check( vvhip_sync( dst.ctx() ), "vvhip_copy_peer" );
  return did;
}

Device::Found Device::find( const Pel* p ) const
{
  Registry& r = registry( m_gpu );
  std::lock_guard<std::mutex> g( r.m );
  for( const Mirror& m : r.mirrors ) if( m.live && m.findable && p >= m.hostBase && p < m.hostEnd ) { Found f; f.ok = true; f.m = m; return f; }
  return Found();
}

void Device::setReference( int id )
{
  Registry& r = registry( m_gpu );
  std::lock_guard<std::mutex> g( r.m );
  Mirror& m = r.mirrors.at( id );
  m.reference = true; m.findable = false;
}

Device::Found Device::findReference( const Pel* p ) const
{
  Registry& r = registry( m_gpu );
  std::lock_guard<std::mutex> g( r.m );
  for( const Mirror& m : r.mirrors ) if( m.live && m.reference && p >= m.hostBase && p < m.hostEnd ) { Found f; f.ok = true; f.m = m; return f; }
  return Found();
}

int16_t* Device::staging( size_t bytes )
{
  if( bytes > m_stageBytes )
  {
    if( m_stage ) { vvhip_sync( m_ctx ); vvhip_free( m_ctx, m_stage ); }
    void* d = nullptr;
    check( vvhip_malloc( m_ctx, &d, bytes * 2 ), "staging" );
    m_stage = static_cast<int16_t*>( d ); m_stageBytes = bytes * 2;
  }
  return m_stage;
}

void* Device::stagingAux( size_t bytes )
{
  if( bytes > m_auxBytes )
  {
    if( m_aux ) { vvhip_sync( m_ctx ); vvhip_free( m_ctx, m_aux ); }
    check( vvhip_malloc( m_ctx, &m_aux, bytes * 2 ), "stagingAux" );
    m_auxBytes = bytes * 2;
  }
  return m_aux;
}

// ------------------------------------------------------------------------------------------------ RdCost
namespace {

// table entries are re-entrant like the reference's (one RdCost per worker thread): every thread stages through its own context, no host lock

struct Resolved { const int16_t* dBase; int stride; int32_t off; };

// host block -> device pointer: inside a registered picture (offset only) or staged copy (compact w x h)
Resolved resolve( Device& dev, const CPelBuf& b, int w, int h, int extraLeft, int extraRight, int16_t*& stageCursor, std::vector<Pel>& hostTmp )
{
  if( const auto m = dev.find( b.buf ) )
    if( m->stride == b.stride ) return { m->dOrigin, m->stride, ( int32_t ) ( b.buf - m->origin ) };
  const int ww = w + extraLeft + extraRight;
  hostTmp.resize( ( size_t ) ww * h );
  for( int y = 0; y < h; y++ ) memcpy( &hostTmp[( size_t ) y * ww], b.buf - extraLeft + ( ptrdiff_t ) y * b.stride, sizeof( Pel ) * ww );
  int16_t* d = stageCursor;
  dev.check( vvhip_upload( dev.ctx(), d, hostTmp.data(), hostTmp.size() * sizeof( Pel ) ), "stage block" );
  dev.check( vvhip_sync( dev.ctx() ), "stage block" );      // hostTmp is reused right away
  stageCursor += ( hostTmp.size() + 7 ) & ~( size_t ) 7;
  return { d, ww, extraLeft };
}

Distortion callOne( int func, const DistParam& dp )
{
  Device& dev = Device::get();
  const int w = dp.org.width, h = dp.org.height;
  int16_t* cursor = dev.staging( ( size_t ) 4 * ( w + 8 ) * h * sizeof( Pel ) + 64 );
  std::vector<Pel> t0, t1;
  const Resolved o = resolve( dev, dp.org, w, h, 0, 0, cursor, t0 );
  const Resolved c = resolve( dev, dp.cur, w, h, 0, 0, cursor, t1 );
  struct Io { vvhip_dist_item it; uint64_t out; } io;
  io.it.org_off = o.off; io.it.cur_off = c.off; io.out = 0;
  char* aux = static_cast<char*>( dev.stagingAux( sizeof( Io ) ) );
  dev.check( vvhip_upload( dev.ctx(), aux, &io, sizeof( io ) ), "dist item" );
  // org and cur may live in different allocations: the ABI takes one base per operand, offsets are relative to each
  dev.check( vvhip_dist_batch( dev.ctx(), func, o.dBase, o.stride, c.dBase, c.stride, w, h, dp.subShift, dp.bitDepth,
                               reinterpret_cast<vvhip_dist_item*>( aux ), 1, reinterpret_cast<uint64_t*>( aux + offsetof( Io, out ) ) ), "vvhip_dist_batch" );
  dev.check( vvhip_download( dev.ctx(), &io.out, aux + offsetof( Io, out ), sizeof( uint64_t ) ), "dist result" );
  return io.out;
}

template<int F> Distortion distEntry( const DistParam& dp )
{
  if( dp.applyWeight ) throw Exception( " no support" );      // RdCost.cpp:303-306
  return callOne( F, dp );
}

// DF_SAD_WITH_MASK: the mask window the walk touches is staged as a compact buffer and the walk re-expressed on it
Distortion sadMaskEntry( const DistParam& dp )
{
  if( dp.applyWeight ) throw Exception( " no support" );
  if( !dp.mask ) throw Exception( "vvhip::RdCost: DF_SAD_WITH_MASK entry called without a mask" );
  Device& dev = Device::get();
  const int w = dp.org.width, h = dp.org.height, step = 1 << dp.subShift, rowsEff = h >> dp.subShift;
  const ptrdiff_t rowAdvance = ( ptrdiff_t ) w * dp.stepX + ( ptrdiff_t ) dp.maskStride * step + dp.maskStride2;
  // gather exactly the mask samples read, in (processed row, column) order: compact rowsEff x w, stepX 1, row advance w
  std::vector<Pel> m( ( size_t ) rowsEff * w ), t0, t1;
  for( int r = 0; r < rowsEff; r++ ) for( int x = 0; x < w; x++ ) m[( size_t ) r * w + x] = dp.mask[r * rowAdvance + ( ptrdiff_t ) x * dp.stepX];
  int16_t* cursor = dev.staging( ( size_t ) 6 * ( w + 8 ) * h * sizeof( Pel ) + 64 );
  const Resolved o = resolve( dev, dp.org, w, h, 0, 0, cursor, t0 );
  const Resolved c = resolve( dev, dp.cur, w, h, 0, 0, cursor, t1 );
  int16_t* dMask = cursor;
  dev.check( vvhip_upload( dev.ctx(), dMask, m.data(), m.size() * sizeof( Pel ) ), "stage mask" );
  struct Io { vvhip_dist_item it; uint64_t out; } io;
  io.it.org_off = o.off; io.it.cur_off = c.off; io.out = 0;
  char* aux = static_cast<char*>( dev.stagingAux( sizeof( Io ) ) );
  dev.check( vvhip_upload( dev.ctx(), aux, &io, sizeof( io ) ), "dist item" );
  // compact mask: +1 per sample leaves the pointer at the next row already -> maskStride 0, maskStride2 0 after the w steps
  dev.check( vvhip_sad_mask_batch( dev.ctx(), o.dBase, o.stride, c.dBase, c.stride, dMask, 0, 1, 0, w, h, dp.subShift, dp.bitDepth,
                                   reinterpret_cast<vvhip_dist_item*>( aux ), nullptr, 1, reinterpret_cast<uint64_t*>( aux + offsetof( Io, out ) ) ), "vvhip_sad_mask_batch" );
  dev.check( vvhip_download( dev.ctx(), &io.out, aux + offsetof( Io, out ), sizeof( uint64_t ) ), "dist result" );
  return io.out;
}

Distortion fxdWtdEntry( const DistParam& dp, uint32_t fixedWeight )
{
  Device& dev = Device::get();
  const int w = dp.org.width, h = dp.org.height;
  int16_t* cursor = dev.staging( ( size_t ) 4 * ( w + 8 ) * h * sizeof( Pel ) + 64 );
  std::vector<Pel> t0, t1;
  const Resolved o = resolve( dev, dp.org, w, h, 0, 0, cursor, t0 );
  const Resolved c = resolve( dev, dp.cur, w, h, 0, 0, cursor, t1 );
  struct Io { vvhip_dist_item it; uint32_t weight, pad; uint64_t out; } io;
  io.it.org_off = o.off; io.it.cur_off = c.off; io.weight = fixedWeight; io.pad = 0; io.out = 0;
  char* aux = static_cast<char*>( dev.stagingAux( sizeof( Io ) ) );
  dev.check( vvhip_upload( dev.ctx(), aux, &io, sizeof( io ) ), "dist item" );
  dev.check( vvhip_fix_weighted_sse_batch( dev.ctx(), o.dBase, o.stride, c.dBase, c.stride, w, h, dp.bitDepth, reinterpret_cast<vvhip_dist_item*>( aux ),
                                           reinterpret_cast<uint32_t*>( aux + offsetof( Io, weight ) ), 1, reinterpret_cast<uint64_t*>( aux + offsetof( Io, out ) ) ),
             "vvhip_fix_weighted_sse_batch" );
  dev.check( vvhip_download( dev.ctx(), &io.out, aux + offsetof( Io, out ), sizeof( uint64_t ) ), "dist result" );
  return io.out;
}

template<int LOG2W> void sadX5Entry( const DistParam& dp, Distortion* cost, bool calcCentre )
{
  Device& dev = Device::get();
  const int w = dp.org.width, h = dp.org.height;
  int16_t* cursor = dev.staging( ( size_t ) 4 * ( w + 16 ) * h * sizeof( Pel ) + 64 );
  std::vector<Pel> t0, t1;
  const Resolved o = resolve( dev, dp.org, w, h, 0, 4, cursor, t0 );     // org + k, k = 0..4
  const Resolved c = resolve( dev, dp.cur, w, h, 4, 0, cursor, t1 );     // cur - k
  struct Io { vvhip_dist_item it; uint64_t out[5]; } io;
  io.it.org_off = o.off; io.it.cur_off = c.off;
  for( int k = 0; k < 5; k++ ) io.out[k] = cost[k];
  char* aux = static_cast<char*>( dev.stagingAux( sizeof( Io ) ) );
  dev.check( vvhip_upload( dev.ctx(), aux, &io, sizeof( io ) ), "x5 item" );
  dev.check( vvhip_sad_x5_batch( dev.ctx(), o.dBase, o.stride, c.dBase, c.stride, w, h, dp.subShift, calcCentre ? 1 : 0,
                                 reinterpret_cast<vvhip_dist_item*>( aux ), 1, reinterpret_cast<uint64_t*>( aux + offsetof( Io, out ) ) ), "vvhip_sad_x5_batch" );
  dev.check( vvhip_download( dev.ctx(), io.out, aux + offsetof( Io, out ), sizeof( io.out ) ), "x5 result" );
  for( int k = 0; k < 5; k++ ) if( k != 2 || calcCentre ) cost[k] = io.out[k];
}

int funcOfEntry( const RdCost& rc, FpDistFunc f )
{
  if( f == rc.m_afpDistortFunc[0][DF_SSE] ) return VVHIP_DF_SSE;
  if( f == rc.m_afpDistortFunc[0][DF_SAD] ) return VVHIP_DF_SAD;
  if( f == rc.m_afpDistortFunc[0][DF_HAD] ) return VVHIP_DF_HAD;
  if( f == rc.m_afpDistortFunc[0][DF_HAD_fast] ) return VVHIP_DF_HAD_FAST;
  if( f == rc.m_afpDistortFunc[0][DF_HAD_2SAD] ) return VVHIP_DF_HAD_2SAD;
  return -1;
}

} // namespace

void RdCost::create( bool /*enableOpt*/ )
{
  Device::get();     // throws here, not at the first distortion call, when there is no GPU
  for( int row = 0; row < 2; row++ )
  {
    for( int i = 0; i < 8; i++ )
    {
      m_afpDistortFunc[row][DF_SSE + i]      = distEntry<VVHIP_DF_SSE>;
      m_afpDistortFunc[row][DF_SAD + i]      = distEntry<VVHIP_DF_SAD>;
      m_afpDistortFunc[row][DF_HAD + i]      = distEntry<VVHIP_DF_HAD>;
      m_afpDistortFunc[row][DF_HAD_fast + i] = distEntry<VVHIP_DF_HAD_FAST>;
    }
    m_afpDistortFunc[row][DF_HAD_2SAD]      = distEntry<VVHIP_DF_HAD_2SAD>;
    m_afpDistortFunc[row][DF_SAD_WITH_MASK] = sadMaskEntry;
  }
  m_fxdWtdPredPtr = fxdWtdEntry;
  m_afpDistortFuncX5[0] = sadX5Entry<3>;
  m_afpDistortFuncX5[1] = sadX5Entry<4>;
}

void RdCost::distAtPositions( int func, const CPelBuf& org, const Pel* refBase, int refStride, int subShift, int bitDepth, const int ( *xy )[2], int n, Distortion* out )
{
  if( n <= 0 ) return;
  Device& dev = Device::get();
  const int w = org.width, h = org.height;
  int x0 = xy[0][0], x1 = xy[0][0], y0 = xy[0][1], y1 = xy[0][1];
  for( int i = 1; i < n; i++ ) { x0 = std::min( x0, xy[i][0] ); x1 = std::max( x1, xy[i][0] ); y0 = std::min( y0, xy[i][1] ); y1 = std::max( y1, xy[i][1] ); }
  const auto mr = dev.findReference( refBase );
  const bool resident = mr && mr->stride == refStride;      // the reference picture is mirrored in HBM (binding: reconstructed pictures, row by row): only offsets travel
  if( resident ) g_refCalls++;
  const int pw = resident ? 0 : x1 - x0 + w, ph = resident ? 0 : y1 - y0 + h;
  std::vector<Pel> host( ( size_t ) w * h + ( size_t ) pw * ph );
  for( int y = 0; y < h; y++ ) memcpy( &host[( size_t ) y * w], org.buf + ( ptrdiff_t ) y * org.stride, sizeof( Pel ) * w );
  Pel* win = host.data() + ( size_t ) w * h;
  for( int y = 0; y < ph; y++ ) memcpy( win + ( size_t ) y * pw, refBase + ( ptrdiff_t ) ( y0 + y ) * refStride + x0, sizeof( Pel ) * pw );
  int16_t* dArea = dev.staging( host.size() * sizeof( Pel ) + 256 );
  dev.check( vvhip_upload( dev.ctx(), dArea, host.data(), host.size() * sizeof( Pel ) ), "search stage" );
  std::vector<vvhip_dist_item> items( n );
  const ptrdiff_t base = resident ? refBase - mr->origin : 0;
  for( int i = 0; i < n; i++ )
  {
    items[i].org_off = 0;
    items[i].cur_off = resident ? ( int32_t ) ( base + ( ptrdiff_t ) xy[i][1] * refStride + xy[i][0] ) : ( xy[i][1] - y0 ) * pw + ( xy[i][0] - x0 );
  }
  char* aux = static_cast<char*>( dev.stagingAux( ( size_t ) n * ( sizeof( vvhip_dist_item ) + sizeof( uint64_t ) ) + 64 ) );
  uint64_t* dOut = reinterpret_cast<uint64_t*>( aux + ( ( ( size_t ) n * sizeof( vvhip_dist_item ) + 15 ) & ~( size_t ) 15 ) );
  dev.check( vvhip_upload( dev.ctx(), aux, items.data(), ( size_t ) n * sizeof( vvhip_dist_item ) ), "search stage" );
  dev.check( vvhip_dist_batch( dev.ctx(), func, dArea, w, resident ? mr->dOrigin : dArea + ( size_t ) w * h, resident ? refStride : pw, w, h, subShift, bitDepth,
                               reinterpret_cast<vvhip_dist_item*>( aux ), n, dOut ), "vvhip_dist_batch" );
  std::vector<uint64_t> res( n );
  dev.check( vvhip_download( dev.ctx(), res.data(), dOut, ( size_t ) n * sizeof( uint64_t ) ), "search stage" );
  for( int i = 0; i < n; i++ ) out[i] = res[i];
}

bool RdCost::patternRefineCosts( const CPelBuf& org, const Pel* refBlk, int refStride, const int ( *qpel )[2], int n, int bitDepth, int hadMode, int reduceTap, bool useAltHpelIf,
                                 Distortion* out )
{
  const int w = org.width, h = org.height;
  if( ( w & 7 ) || w > 64 || h > 64 || h < 4 || n < 1 || n > 16 || reduceTap < 0 || reduceTap > 2 ) return false;
  if( ( hadMode == 1 || hadMode == 2 ) && ( h & 3 ) ) return false;
  Device& dev = Device::get();
  const auto mr = dev.findReference( refBlk );
  const bool resident = mr && mr->stride == refStride;      // reference picture mirrored in HBM: the block's window is not staged
  if( resident ) g_refCalls++;
  const int M0 = 5, M1 = 6, pitch = resident ? refStride : w + M0 + M1, rows = resident ? 0 : h + M0 + M1;
  std::vector<Pel> host( ( size_t ) w * h + ( resident ? 0 : ( size_t ) pitch * rows ) );
  for( int y = 0; y < h; y++ ) memcpy( &host[( size_t ) y * w], org.buf + ( ptrdiff_t ) y * org.stride, sizeof( Pel ) * w );
  Pel* win = host.data() + ( size_t ) w * h;
  for( int y = 0; y < rows; y++ ) memcpy( win + ( size_t ) y * pitch, refBlk + ( ptrdiff_t ) ( y - M0 ) * refStride - M0, sizeof( Pel ) * pitch );
  int16_t* dArea = dev.staging( host.size() * sizeof( Pel ) + 256 );
  dev.check( vvhip_upload( dev.ctx(), dArea, host.data(), host.size() * sizeof( Pel ) ), "pattern refinement" );
  struct Io { vvhip_subpel_item base; uint32_t pad; uint64_t cost[16]; } io;
  memset( &io, 0, sizeof( io ) );
  io.base.org_off = 0; io.base.ref_off = resident ? ( int32_t ) ( refBlk - mr->origin ) : M0 * pitch + M0; io.base.frac_x = 0; io.base.frac_y = 0;
  char* aux = static_cast<char*>( dev.stagingAux( sizeof( Io ) ) );
  dev.check( vvhip_upload( dev.ctx(), aux, &io, sizeof( io ) ), "pattern refinement" );
  const int16_t* dRef = resident ? mr->dOrigin : dArea + ( size_t ) w * h;
  int16_t offs[32];
  for( int i = 0; i < n; i++ ) { offs[2 * i] = ( int16_t ) ( qpel[i][0] * 4 ); offs[2 * i + 1] = ( int16_t ) ( qpel[i][1] * 4 ); }
  const int func = hadMode == 0 ? VVHIP_DF_SAD : hadMode == 1 ? VVHIP_DF_HAD : VVHIP_DF_HAD_FAST;
  dev.check( vvhip_subpel_refine_batch( dev.ctx(), func, dArea, w, dRef, pitch, w, h, bitDepth, reduceTap, useAltHpelIf ? 1 : 0,
                                        reinterpret_cast<vvhip_subpel_item*>( aux ), 1, offs, n, reinterpret_cast<uint64_t*>( aux + offsetof( Io, cost ) ) ), "vvhip_subpel_refine_batch" );
  dev.check( vvhip_download( dev.ctx(), io.cost, aux + offsetof( Io, cost ), sizeof( uint64_t ) * n ), "pattern refinement" );
  for( int i = 0; i < n; i++ ) out[i] = io.cost[i];
  return true;
}

void RdCost::setDistParamGeo( DistParam& dp, const CPelBuf& org, const Pel* refY, int refStride, const Pel* mask, int maskStride, int stepX, int maskStride2, int bitDepth, int compID )
{
  dp.bitDepth = bitDepth; dp.compID = compID;
  dp.org = org;
  dp.cur.buf = refY; dp.cur.stride = refStride; dp.cur.width = org.width; dp.cur.height = org.height;
  dp.mask = mask; dp.maskStride = maskStride; dp.stepX = stepX; dp.maskStride2 = maskStride2;     // subShift is left as it is, as in the reference
  dp.maximumDistortionForEarlyExit = ~0ull;
  dp.distFunc = m_afpDistortFunc[0][DF_SAD_WITH_MASK];
}

void RdCost::setDistParam( DistParam& dp, const CPelBuf& org, const Pel* refY, int refStride, int bitDepth, int compID, int subShiftMode, int useHadamard )
{
  dp.bitDepth = bitDepth; dp.compID = compID;
  dp.org = org;
  dp.cur.buf = refY; dp.cur.stride = refStride; dp.cur.width = org.width; dp.cur.height = org.height;
  dp.maximumDistortionForEarlyExit = ~0ull;
  const int base = ( bitDepth > 10 || dp.applyWeight ) ? 1 : 0;
  if( !useHadamard ) dp.distFunc = m_afpDistortFunc[base][DF_SAD + ilog2( org.width )];
  else               dp.distFunc = m_afpDistortFunc[base][( useHadamard == 1 ? DF_HAD : DF_HAD_fast ) + ilog2( org.width )];
  dp.subShift = 0;
  if( subShiftMode == 1 ) { if( org.height > 8 && org.width <= 128 ) dp.subShift = 1; }
  else if( subShiftMode == 2 ) { if( org.height > 8 ) dp.subShift = 1; }
}

Distortion RdCost::getDistPart( const CPelBuf& org, const CPelBuf& cur, int bitDepth, DFunc eDFunc )
{
  DistParam dp; dp.org = org; dp.cur = cur; dp.bitDepth = bitDepth;
  const int base = bitDepth > 10 ? 1 : 0;
  return m_afpDistortFunc[base][eDFunc + ilog2( org.width )]( dp );
}

int RdCost::enqueue( const DistParam& dp )
{
  Device& dev = Device::get();
  const auto mo = dev.find( dp.org.buf );
  const auto mc = dev.find( dp.cur.buf );
  const int func = funcOfEntry( *this, dp.distFunc );
  if( !mo || !mc || mo->stride != dp.org.stride || mc->stride != dp.cur.stride || func < 0 )
    throw Exception( "RdCost::enqueue: org/cur must point into pictures registered with vvhip::Device and distFunc must be a table entry" );
  Pending p; p.func = func; p.w = dp.org.width; p.h = dp.org.height; p.subShift = dp.subShift; p.mo = *mo; p.mc = *mc;
  p.orgOff = ( int32_t ) ( dp.org.buf - mo->origin ); p.curOff = ( int32_t ) ( dp.cur.buf - mc->origin );
  m_pending.push_back( p );
  m_results.push_back( 0 );
  return ( int ) m_results.size() - 1;
}

void RdCost::flush()
{
  Device& dev = Device::get();
  const size_t first = m_results.size() - m_pending.size();
  // group by (func, w, h, subShift, planes): one launch per group
  std::map<std::tuple<int, int, int, int, const void*, const void*>, std::vector<int>> groups;
  for( size_t i = 0; i < m_pending.size(); i++ )
  {
    const Pending& p = m_pending[i];
    groups[std::make_tuple( p.func, p.w, p.h, p.subShift, ( const void* ) p.mo.dOrigin, ( const void* ) p.mc.dOrigin )].push_back( ( int ) i );
  }
  const size_t n = m_pending.size();
  char* aux = static_cast<char*>( dev.stagingAux( n * ( sizeof( vvhip_dist_item ) + sizeof( uint64_t ) ) + 64 ) );
  vvhip_dist_item* dItems = reinterpret_cast<vvhip_dist_item*>( aux );
  uint64_t* dOut = reinterpret_cast<uint64_t*>( aux + ( ( n * sizeof( vvhip_dist_item ) + 15 ) & ~( size_t ) 15 ) );
  std::vector<vvhip_dist_item> items( n );
  std::vector<int> order; order.reserve( n );
  for( auto& kv : groups ) for( int i : kv.second ) { items[order.size()].org_off = m_pending[i].orgOff; items[order.size()].cur_off = m_pending[i].curOff; order.push_back( i ); }
  dev.check( vvhip_upload( dev.ctx(), dItems, items.data(), n * sizeof( vvhip_dist_item ) ), "flush items" );
  size_t pos = 0;
  for( auto& kv : groups )
  {
    const Pending& p = m_pending[kv.second[0]];
    dev.check( vvhip_dist_batch( dev.ctx(), p.func, p.mo.dOrigin, p.mo.stride, p.mc.dOrigin, p.mc.stride, p.w, p.h, p.subShift, 10,
                                 dItems + pos, ( int ) kv.second.size(), dOut + pos ), "flush launch" );
    pos += kv.second.size();
  }
  std::vector<uint64_t> out( n );
  dev.check( vvhip_download( dev.ctx(), out.data(), dOut, n * sizeof( uint64_t ) ), "flush results" );
  for( size_t k = 0; k < n; k++ ) m_results[first + order[k]] = out[k];
  m_pending.clear();
}

// ------------------------------------------------------------------------------------------------ TCoeffOps
namespace {

void fwd2D( const Pel* resi, ptrdiff_t stride, TCoeff* coef, unsigned w, unsigned h, int trHor, int trVer, int bitDepth )
{
  Device& dev = Device::get();
  const size_t area = ( size_t ) w * h;
  std::vector<Pel> tmp( area );
  for( unsigned y = 0; y < h; y++ ) memcpy( &tmp[( size_t ) y * w], resi + y * stride, sizeof( Pel ) * w );
  int16_t* dResi = dev.staging( area * sizeof( Pel ) + 64 );
  char* aux = static_cast<char*>( dev.stagingAux( area * sizeof( TCoeff ) + 64 ) );
  int32_t zero = 0;
  dev.check( vvhip_upload( dev.ctx(), dResi, tmp.data(), area * sizeof( Pel ) ), "fwd2D" );
  dev.check( vvhip_upload( dev.ctx(), aux, &zero, sizeof( zero ) ), "fwd2D" );
  dev.check( vvhip_fwd_transform_batch( dev.ctx(), dResi, ( int ) w, reinterpret_cast<int32_t*>( aux ), 1, ( int ) w, ( int ) h, trHor, trVer, bitDepth,
                                        reinterpret_cast<int32_t*>( aux + 64 ) ), "vvhip_fwd_transform_batch" );
  dev.check( vvhip_download( dev.ctx(), coef, aux + 64, area * sizeof( TCoeff ) ), "fwd2D" );
}

void inv2D( const TCoeff* coef, Pel* resi, ptrdiff_t stride, unsigned w, unsigned h, int trHor, int trVer, int bitDepth )
{
  Device& dev = Device::get();
  const size_t area = ( size_t ) w * h;
  int16_t* dResi = dev.staging( area * sizeof( Pel ) + 64 );
  char* aux = static_cast<char*>( dev.stagingAux( area * sizeof( TCoeff ) + 64 ) );
  int32_t zero = 0;
  dev.check( vvhip_upload( dev.ctx(), aux, &zero, sizeof( zero ) ), "inv2D" );
  dev.check( vvhip_upload( dev.ctx(), aux + 64, coef, area * sizeof( TCoeff ) ), "inv2D" );
  dev.check( vvhip_inv_transform_batch( dev.ctx(), reinterpret_cast<int32_t*>( aux + 64 ), 1, ( int ) w, ( int ) h, trHor, trVer, bitDepth,
                                        dResi, ( int ) w, reinterpret_cast<int32_t*>( aux ) ), "vvhip_inv_transform_batch" );
  std::vector<Pel> tmp( area );
  dev.check( vvhip_download( dev.ctx(), tmp.data(), dResi, area * sizeof( Pel ) ), "inv2D" );
  for( unsigned y = 0; y < h; y++ ) memcpy( resi + y * stride, &tmp[( size_t ) y * w], sizeof( Pel ) * w );
}

// ---- the table's own slots (host pointers, one synchronous launch each) ----
char* auxArea( Device& dev, size_t bytes ) { return static_cast<char*>( dev.stagingAux( bytes + 256 ) ); }

template<int N> void fwdCore( const TMatrixCoeff* tc, const TCoeff* src, TCoeff* dst, unsigned line, unsigned reducedLine, unsigned cutoff, int shift )
{
  Device& dev = Device::get();
  const size_t mB = sizeof( TMatrixCoeff ) * N * N, sB = sizeof( TCoeff ) * ( size_t ) line * N, dB = sizeof( TCoeff ) * ( size_t ) line * cutoff;
  char* aux = auxArea( dev, mB + sB + dB );
  char* dM = aux; char* dS = aux + ( ( mB + 63 ) & ~( size_t ) 63 ); char* dD = dS + ( ( sB + 63 ) & ~( size_t ) 63 );
  dev.check( vvhip_upload( dev.ctx(), dM, tc, mB ), "fastFwdCore" );
  dev.check( vvhip_upload( dev.ctx(), dS, src, sizeof( TCoeff ) * ( size_t ) reducedLine * N ), "fastFwdCore" );
  dev.check( vvhip_upload( dev.ctx(), dD, dst, dB ), "fastFwdCore" );          // entries the core does not write keep the caller's values
  dev.check( vvhip_fast_fwd_core( dev.ctx(), N, reinterpret_cast<int16_t*>( dM ), reinterpret_cast<int32_t*>( dS ), reinterpret_cast<int32_t*>( dD ), line, reducedLine, cutoff, shift ), "vvhip_fast_fwd_core" );
  dev.check( vvhip_download( dev.ctx(), dst, dD, dB ), "fastFwdCore" );
}

template<int N> void invCore( const TMatrixCoeff* it, const TCoeff* src, TCoeff* dst, unsigned lines, unsigned reducedLines, unsigned rows )
{
  Device& dev = Device::get();
  const size_t mB = sizeof( TMatrixCoeff ) * N * N, sB = sizeof( TCoeff ) * ( size_t ) lines * N, dB = sizeof( TCoeff ) * ( size_t ) reducedLines * N;
  char* aux = auxArea( dev, mB + sB + dB );
  char* dM = aux; char* dS = aux + ( ( mB + 63 ) & ~( size_t ) 63 ); char* dD = dS + ( ( sB + 63 ) & ~( size_t ) 63 );
  dev.check( vvhip_upload( dev.ctx(), dM, it, mB ), "fastInvCore" );
  dev.check( vvhip_upload( dev.ctx(), dS, src, sizeof( TCoeff ) * ( size_t ) rows * lines ), "fastInvCore" );
  dev.check( vvhip_upload( dev.ctx(), dD, dst, dB ), "fastInvCore" );          // accumulates into the caller's (zeroed) dst
  dev.check( vvhip_fast_inv_core( dev.ctx(), N, reinterpret_cast<int16_t*>( dM ), reinterpret_cast<int32_t*>( dS ), reinterpret_cast<int32_t*>( dD ), lines, reducedLines, rows ), "vvhip_fast_inv_core" );
  dev.check( vvhip_download( dev.ctx(), dst, dD, dB ), "fastInvCore" );
}

void roundClipSlot( TCoeff* dst, unsigned width, unsigned height, unsigned stride, const TCoeff outputMin, const TCoeff outputMax, const TCoeff round, const TCoeff shift )
{
  if( !width || !height ) return;
  Device& dev = Device::get();
  const size_t bytes = sizeof( TCoeff ) * ( ( size_t ) ( height - 1 ) * stride + width );
  char* aux = auxArea( dev, bytes );
  dev.check( vvhip_upload( dev.ctx(), aux, dst, bytes ), "roundClip" );
  dev.check( vvhip_round_clip( dev.ctx(), reinterpret_cast<int32_t*>( aux ), width, height, stride, outputMin, outputMax, round, shift ), "vvhip_round_clip" );
  dev.check( vvhip_download( dev.ctx(), dst, aux, bytes ), "roundClip" );
}

void cpyResiSlot( const TCoeff* src, Pel* dst, ptrdiff_t stride, unsigned width, unsigned height )
{
  if( !width || !height ) return;
  Device& dev = Device::get();
  const size_t area = ( size_t ) width * height;
  char* aux = auxArea( dev, area * ( sizeof( TCoeff ) + sizeof( Pel ) ) );
  char* dD = aux + ( ( area * sizeof( TCoeff ) + 63 ) & ~( size_t ) 63 );
  dev.check( vvhip_upload( dev.ctx(), aux, src, area * sizeof( TCoeff ) ), "cpyResi" );
  dev.check( vvhip_cpy_resi( dev.ctx(), reinterpret_cast<int32_t*>( aux ), reinterpret_cast<int16_t*>( dD ), width, width, height ), "vvhip_cpy_resi" );   // compact on the device ...
  std::vector<Pel> tmp( area );
  dev.check( vvhip_download( dev.ctx(), tmp.data(), dD, area * sizeof( Pel ) ), "cpyResi" );
  for( unsigned y = 0; y < height; y++ ) memcpy( dst + ( ptrdiff_t ) y * stride, &tmp[( size_t ) y * width], sizeof( Pel ) * width );                      // ... strided on the host
}

void cpyCoeffSlot( const Pel* src, ptrdiff_t stride, TCoeff* dst, unsigned width, unsigned height )
{
  if( !width || !height ) return;
  Device& dev = Device::get();
  const size_t area = ( size_t ) width * height;
  std::vector<Pel> tmp( area );
  for( unsigned y = 0; y < height; y++ ) memcpy( &tmp[( size_t ) y * width], src + ( ptrdiff_t ) y * stride, sizeof( Pel ) * width );
  char* aux = auxArea( dev, area * ( sizeof( TCoeff ) + sizeof( Pel ) ) );
  char* dD = aux + ( ( area * sizeof( Pel ) + 63 ) & ~( size_t ) 63 );
  dev.check( vvhip_upload( dev.ctx(), aux, tmp.data(), area * sizeof( Pel ) ), "cpyCoeff" );
  dev.check( vvhip_cpy_coeff( dev.ctx(), reinterpret_cast<int16_t*>( aux ), width, reinterpret_cast<int32_t*>( dD ), width, height ), "vvhip_cpy_coeff" );
  dev.check( vvhip_download( dev.ctx(), dst, dD, area * sizeof( TCoeff ) ), "cpyCoeff" );
}

} // namespace

TCoeffOps::TCoeffOps()
{
  fwdTransform2D = fwd2D;
  invTransform2D = inv2D;
  cpyResi4 = cpyResi8 = cpyResiSlot;
  cpyCoeff4 = cpyCoeff8 = cpyCoeffSlot;
  roundClip4 = roundClip8 = roundClipSlot;
  fastInvCore[0] = invCore<4>;  fastInvCore[1] = invCore<8>;  fastInvCore[2] = invCore<16>;  fastInvCore[3] = invCore<32>;  fastInvCore[4] = invCore<64>;
  fastFwdCore_2D[0] = fastFwdCore_1D[0] = fwdCore<4>;   fastFwdCore_2D[1] = fastFwdCore_1D[1] = fwdCore<8>;   fastFwdCore_2D[2] = fastFwdCore_1D[2] = fwdCore<16>;
  fastFwdCore_2D[3] = fastFwdCore_1D[3] = fwdCore<32>;  fastFwdCore_2D[4] = fastFwdCore_1D[4] = fwdCore<64>;
}
TCoeffOps g_tCoeffOps;

// ------------------------------------------------------------------------------------------------ InterpolationFilter
namespace {

struct IfTables
{
  TFilterCoeff luma8[17][8], luma6[17][8], alt[8], chroma[33][4];
  IfTables()
  {
    // phases 0..8 (luma, 1/16) and 0..16 (chroma, 1/32); the other half is the mirror image (InterpolationFilter.cpp:64-142)
    static const int8_t l8[9][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { -1, 2, -5, 62, 8, -3, 1, 0 }, { -1, 3, -8, 60, 13, -4, 1, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                     { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 }, { -1, 4, -11, 40, 40, -11, 4, -1 } };
    static const int8_t l6[9][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { 0, 1, -5, 62, 8, -3, 1, 0 }, { 0, 2, -8, 60, 13, -4, 1, 0 }, { 0, 3, -10, 58, 17, -5, 1, 0 },
                                     { 0, 3, -11, 52, 26, -8, 2, 0 }, { 0, 2, -9, 47, 31, -10, 3, 0 }, { 0, 3, -11, 45, 34, -10, 3, 0 }, { 0, 3, -11, 40, 40, -11, 3, 0 } };
    static const int8_t c4[17][4] = { { 0, 64, 0, 0 }, { -1, 63, 2, 0 }, { -2, 62, 4, 0 }, { -2, 60, 7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 }, { -4, 55, 15, -2 },
                                      { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 } };
    static const int8_t a[8] = { 0, 3, 9, 20, 20, 9, 3, 0 };
    for( int p = 0; p <= 16; p++ ) for( int k = 0; k < 8; k++ ) { luma8[p][k] = p <= 8 ? l8[p][k] : l8[16 - p][7 - k]; luma6[p][k] = p <= 8 ? l6[p][k] : l6[16 - p][7 - k]; }
    for( int p = 0; p <= 32; p++ ) for( int k = 0; k < 4; k++ ) chroma[p][k] = p <= 16 ? c4[p][k] : c4[32 - p][3 - k];
    for( int k = 0; k < 8; k++ ) alt[k] = a[k];
  }
};
const IfTables& ifTables() { static const IfTables t; return t; }

// stage the (w + left + right) x (h + top + bottom) neighbourhood of a host block compactly on the device; returns the device pointer of the block's sample (0,0)
int16_t* stageRegion( Device& dev, int16_t* dArea, const Pel* src, int srcStride, int w, int h, int left, int right, int top, int bottom, std::vector<Pel>& tmp, int& pitch )
{
  pitch = w + left + right;
  const int rows = h + top + bottom;
  tmp.resize( ( size_t ) pitch * rows );
  for( int y = 0; y < rows; y++ ) memcpy( &tmp[( size_t ) y * pitch], src + ( ptrdiff_t ) ( y - top ) * srcStride - left, sizeof( Pel ) * pitch );
  dev.check( vvhip_upload( dev.ctx(), dArea, tmp.data(), tmp.size() * sizeof( Pel ) ), "interpolation source" );
  return dArea + ( size_t ) top * pitch + left;
}

void fetchBlock( Device& dev, const int16_t* dBlk, Pel* dst, int dstStride, int w, int h, std::vector<Pel>& tmp )
{
  tmp.resize( ( size_t ) w * h );
  dev.check( vvhip_download( dev.ctx(), tmp.data(), dBlk, tmp.size() * sizeof( Pel ) ), "interpolation result" );
  for( int y = 0; y < h; y++ ) memcpy( dst + ( ptrdiff_t ) y * dstStride, &tmp[( size_t ) y * w], sizeof( Pel ) * w );
}

template<int N, bool VER, bool FIRST, bool LAST>
void ifSlot( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, TFilterCoeff const* coeff )
{
  if( width <= 0 || height <= 0 ) return;
  Device& dev = Device::get();
  const int lo = N / 2 - 1, hi = N / 2;
  const size_t srcElems = ( size_t ) ( width + ( VER ? 0 : N ) ) * ( height + ( VER ? N : 0 ) ), dstElems = ( size_t ) width * height;
  int16_t* area = dev.staging( ( srcElems + dstElems ) * sizeof( Pel ) + 256 );
  std::vector<Pel> tmp;
  int pitch;
  const int16_t* dSrc = stageRegion( dev, area, src, srcStride, width, height, VER ? 0 : lo, VER ? 0 : hi, VER ? lo : 0, VER ? hi : 0, tmp, pitch );
  int16_t* dDst = area + ( ( srcElems + 63 ) & ~( size_t ) 63 );
  dev.check( vvhip_if_filter( dev.ctx(), N, VER, FIRST, LAST, clpRng.bd, dSrc, pitch, dDst, width, width, height, coeff ), "vvhip_if_filter" );
  fetchBlock( dev, dDst, dst, dstStride, width, height, tmp );
}

template<bool FIRST, bool LAST>
void ifCopySlot( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, bool biMCForDMVR )
{
  if( width <= 0 || height <= 0 ) return;
  Device& dev = Device::get();
  const size_t elems = ( size_t ) width * height;
  int16_t* area = dev.staging( 2 * elems * sizeof( Pel ) + 256 );
  std::vector<Pel> tmp;
  int pitch;
  const int16_t* dSrc = stageRegion( dev, area, src, srcStride, width, height, 0, 0, 0, 0, tmp, pitch );
  int16_t* dDst = area + ( ( elems + 63 ) & ~( size_t ) 63 );
  dev.check( vvhip_if_copy( dev.ctx(), FIRST, LAST, clpRng.bd, dSrc, pitch, dDst, width, width, height, biMCForDMVR ), "vvhip_if_copy" );
  fetchBlock( dev, dDst, dst, dstStride, width, height, tmp );
}

// fused entries (filterWxH_N8 / _N4 / _N2, InterpolationFilter.cpp:772-1010): horizontal first pass over the rows the vertical taps need, then the vertical pass
template<int N, bool LAST>
void ifFused( const ClpRng& clpRng, Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, TFilterCoeff const* coeffH, TFilterCoeff const* coeffV )
{
  if( width <= 0 || height <= 0 ) return;
  Device& dev = Device::get();
  const int lo = N / 2 - 1, hi = N / 2, rows = height + N - 1;
  const size_t srcElems = ( size_t ) ( width + N ) * rows, midElems = ( size_t ) width * rows, dstElems = ( size_t ) width * height;
  int16_t* area = dev.staging( ( srcElems + midElems + dstElems ) * sizeof( Pel ) + 512 );
  std::vector<Pel> tmp;
  int pitch;
  const int16_t* dSrc = stageRegion( dev, area, src, srcStride, width, height, lo, hi, lo, hi, tmp, pitch );
  int16_t* dMid = area + ( ( srcElems + 63 ) & ~( size_t ) 63 );
  int16_t* dDst = dMid + ( ( midElems + 63 ) & ~( size_t ) 63 );
  // the 8-entry rows of the 6-tap sets begin and end with 0: running 8 taps on them is the same sum (the reference's fused cores do exactly that)
  dev.check( vvhip_if_filter( dev.ctx(), N, 0, 1, 0, clpRng.bd, dSrc - ( ptrdiff_t ) lo * pitch, pitch, dMid, width, width, rows, coeffH ), "vvhip_if_filter (hor)" );
  dev.check( vvhip_if_filter( dev.ctx(), N, 1, 0, LAST, clpRng.bd, dMid + ( size_t ) lo * width, width, dDst, width, width, height, coeffV ), "vvhip_if_filter (ver)" );
  fetchBlock( dev, dDst, dst, dstStride, width, height, tmp );
}

} // namespace

InterpolationFilter::InterpolationFilter()
{
  Device::get();
#define IF_FILL( T, IDX, N ) T[IDX][0][0] = ifSlot<N, T##_V, false, false>; T[IDX][0][1] = ifSlot<N, T##_V, false, true>; T[IDX][1][0] = ifSlot<N, T##_V, true, false>; T[IDX][1][1] = ifSlot<N, T##_V, true, true>;
  constexpr bool m_filterHor_V = false, m_filterVer_V = true;
  IF_FILL( m_filterHor, 0, 8 ) IF_FILL( m_filterHor, 1, 4 ) IF_FILL( m_filterHor, 2, 2 ) IF_FILL( m_filterHor, 3, 6 )
  IF_FILL( m_filterVer, 0, 8 ) IF_FILL( m_filterVer, 1, 4 ) IF_FILL( m_filterVer, 2, 2 ) IF_FILL( m_filterVer, 3, 6 )
#undef IF_FILL
  m_filterCopy[0][0] = ifCopySlot<false, false>; m_filterCopy[0][1] = ifCopySlot<false, true>; m_filterCopy[1][0] = ifCopySlot<true, false>; m_filterCopy[1][1] = ifCopySlot<true, true>;
  m_filter4x4[0][0] = ifFused<8, false>; m_filter4x4[0][1] = ifFused<8, true>; m_filter4x4[1][0] = ifFused<4, false>; m_filter4x4[1][1] = ifFused<4, true>;
  m_filter8xH[0][0] = m_filter16xH[0][0] = ifFused<8, false>; m_filter8xH[0][1] = m_filter16xH[0][1] = ifFused<8, true>;
  m_filter8xH[1][0] = m_filter16xH[1][0] = ifFused<4, false>; m_filter8xH[1][1] = m_filter16xH[1][1] = ifFused<4, true>;
  m_filter8xH[2][0] = m_filter16xH[2][0] = ifFused<2, false>; m_filter8xH[2][1] = m_filter16xH[2][1] = ifFused<2, true>;
}

const TFilterCoeff* InterpolationFilter::lumaFilter( int frac )    { return ifTables().luma8[frac]; }
const TFilterCoeff* InterpolationFilter::lumaFilter4x4( int frac ) { return ifTables().luma6[frac]; }
const TFilterCoeff* InterpolationFilter::lumaAltHpelIFilter()      { return ifTables().alt; }
const TFilterCoeff* InterpolationFilter::chromaFilter( int frac32 ) { return ifTables().chroma[frac32]; }

void InterpolationFilter::filterHor( Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, int frac, bool isLast, const ClpRng& clpRng, bool useAltHpelIf, int reduceTap )
{
  if( frac == 0 ) { m_filterCopy[1][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, false ); return; }        // :559-565 (copyBuffer == filterCopy<true,true>)
  if( frac < 0 || frac >= 16 ) throw Exception( "Invalid fraction" );
  if( reduceTap == 0 || ( useAltHpelIf && frac == 8 ) )
  {
    if( useAltHpelIf && frac == 8 ) m_filterHor[3][1][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaAltHpelIFilter() );
    else if( ( width == 4 && height == 4 ) || ( width == 4 && height == 4 + 8 - 1 ) ) m_filterHor[3][1][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaFilter4x4( frac ) );
    else m_filterHor[0][1][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaFilter( frac ) );
  }
  else if( reduceTap == 1 ) m_filterHor[3][1][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaFilter4x4( frac ) );
  else                      m_filterHor[1][1][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, chromaFilter( frac << 1 ) );
}

void InterpolationFilter::filterVer( Pel const* src, int srcStride, Pel* dst, int dstStride, int width, int height, int frac, bool isFirst, bool isLast, const ClpRng& clpRng, bool useAltHpelIf, int reduceTap )
{
  if( frac == 0 ) { m_filterCopy[isFirst][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, false ); return; }  // :619-622
  if( frac < 0 || frac >= 16 ) throw Exception( "Invalid fraction" );
  if( reduceTap == 0 || ( useAltHpelIf && frac == 8 ) )
  {
    if( useAltHpelIf && frac == 8 ) m_filterVer[3][isFirst][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaAltHpelIFilter() );
    else if( width == 4 && height == 4 ) m_filterVer[3][isFirst][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaFilter4x4( frac ) );
    else m_filterVer[0][isFirst][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaFilter( frac ) );
  }
  else if( reduceTap == 1 ) m_filterVer[3][isFirst][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, lumaFilter4x4( frac ) );
  else                      m_filterVer[1][isFirst][isLast]( clpRng, src, srcStride, dst, dstStride, width, height, chromaFilter( frac << 1 ) );
}

// ------------------------------------------------------------------------------------------------ DMVROps
bool DMVROps::refineCu( const Pel* ref0, int stride0, int fx0, int fy0, const Pel* ref1, int stride1, int fx1, int fy1, int cuWidth, int cuHeight, int dx, int dy, int bitDepth,
                        int16_t* mvd, uint64_t* minCost )
{
  if( ( dx != 8 && dx != 16 ) || ( dy != 8 && dy != 16 ) || cuWidth % dx || cuHeight % dy || bitDepth > 10 ) return false;
  Device& dev = Device::get();
  // the bilinear prediction of the (w+4) x (h+4) area reads one more column / row: stage (w+5) x (h+5) of each list, compact
  const int pw = cuWidth + 5, ph = cuHeight + 5;
  std::vector<Pel> host( ( size_t ) 2 * pw * ph );
  for( int y = 0; y < ph; y++ )
  {
    memcpy( &host[( size_t ) y * pw], ref0 + ( ptrdiff_t ) y * stride0, sizeof( Pel ) * pw );
    memcpy( &host[( size_t ) ( ph + y ) * pw], ref1 + ( ptrdiff_t ) y * stride1, sizeof( Pel ) * pw );
  }
  int16_t* dArea = dev.staging( host.size() * sizeof( Pel ) + 256 );
  dev.check( vvhip_upload( dev.ctx(), dArea, host.data(), host.size() * sizeof( Pel ) ), "DMVR windows" );
  const int nx = cuWidth / dx, ny = cuHeight / dy, n = nx * ny;
  std::vector<vvhip_dmvr_item> items( n );
  for( int by = 0; by < ny; by++ ) for( int bx = 0; bx < nx; bx++ )
  {
    vvhip_dmvr_item& it = items[by * nx + bx];
    // the device entry subtracts the 2-sample search margin itself: hand it the position 2 samples inside the staged window
    it.ref0_off = ( by * dy + 2 ) * pw + bx * dx + 2; it.ref1_off = it.ref0_off;
    it.frac0_x = ( int16_t ) fx0; it.frac0_y = ( int16_t ) fy0; it.frac1_x = ( int16_t ) fx1; it.frac1_y = ( int16_t ) fy1;
  }
  char* aux = static_cast<char*>( dev.stagingAux( ( size_t ) n * ( sizeof( vvhip_dmvr_item ) + sizeof( vvhip_dmvr_result ) ) + 64 ) );
  vvhip_dmvr_result* dRes = reinterpret_cast<vvhip_dmvr_result*>( aux + ( ( ( size_t ) n * sizeof( vvhip_dmvr_item ) + 15 ) & ~( size_t ) 15 ) );
  dev.check( vvhip_upload( dev.ctx(), aux, items.data(), ( size_t ) n * sizeof( vvhip_dmvr_item ) ), "DMVR items" );
  dev.check( vvhip_dmvr_refine_batch( dev.ctx(), dArea, pw, dArea + ( size_t ) pw * ph, pw, reinterpret_cast<vvhip_dmvr_item*>( aux ), n, dx, dy, bitDepth, dRes ), "vvhip_dmvr_refine_batch" );
  std::vector<vvhip_dmvr_result> res( n );
  dev.check( vvhip_download( dev.ctx(), res.data(), dRes, ( size_t ) n * sizeof( vvhip_dmvr_result ) ), "DMVR results" );
  for( int i = 0; i < n; i++ ) { mvd[2 * i] = res[i].mvd_x; mvd[2 * i + 1] = res[i].mvd_y; minCost[i] = res[i].min_cost; }
  return true;
}

// ------------------------------------------------------------------------------------------------ ALFOps
namespace {
// stages a plane with its 4-sample border compactly: returns the device pointer of sample (0,0) and the pitch
int16_t* stageBordered( Device& dev, const Pel* p, int stride, int w, int h, int border, size_t slotOffset, int& pitch )
{
  pitch = ( w + 2 * border + 7 ) & ~7;
  std::vector<Pel> host( ( size_t ) pitch * ( h + 2 * border ) );
  for( int y = -border; y < h + border; y++ ) memcpy( &host[( size_t ) ( y + border ) * pitch], p + ( ptrdiff_t ) y * stride - border, sizeof( Pel ) * ( w + 2 * border ) );
  int16_t* base = dev.staging( slotOffset + host.size() * sizeof( Pel ) + 256 ) + slotOffset / sizeof( Pel );
  dev.check( vvhip_upload( dev.ctx(), base, host.data(), host.size() * sizeof( Pel ) ), "ALF plane" );
  return base + ( size_t ) border * pitch + border;
}
} // namespace

bool ALFOps::deriveClassification( const Pel* rec, int recStride, int width, int height, int bitDepth, int vbCTUHeight, int vbPos, uint8_t* cls )
{
  if( ( width & 3 ) || ( height & 3 ) || width < 4 || height < 4 ) return false;
  Device& dev = Device::get();
  int pitch;
  const int16_t* dRec = stageBordered( dev, rec, recStride, width, height, 4, 0, pitch );
  const size_t n = ( size_t ) ( width / 4 ) * ( height / 4 ) * 2;
  uint8_t* dCls = static_cast<uint8_t*>( dev.stagingAux( n + 64 ) );
  dev.check( vvhip_alf_classify( dev.ctx(), dRec, pitch, width, height, bitDepth, vbCTUHeight, vbPos, dCls ), "vvhip_alf_classify" );
  dev.check( vvhip_download( dev.ctx(), cls, dCls, n ), "ALF classes" );
  return true;
}

bool ALFOps::getStatistics( const Pel* org, int orgStride, const Pel* rec, int recStride, int width, int height, int ctuSize, int filterLength,
                            const uint8_t* cls, int vbCTUHeight, int vbPos, float* out, const float* init )
{
  if( ( width & 3 ) || ( height & 3 ) || width < 4 || height < 4 || ( filterLength != 7 && filterLength != 5 ) || ctuSize > 128 ) return false;
  Device& dev = Device::get();
  // one staging allocation: [rec with border][org compact]
  const int recPitchGuess = ( width + 8 + 7 ) & ~7;
  const size_t recBytes = ( ( size_t ) recPitchGuess * ( height + 8 ) * sizeof( Pel ) + 255 ) & ~( size_t ) 255;
  const int orgPitch = ( width + 7 ) & ~7;
  std::vector<Pel> horg( ( size_t ) orgPitch * height );
  for( int y = 0; y < height; y++ ) memcpy( &horg[( size_t ) y * orgPitch], org + ( ptrdiff_t ) y * orgStride, sizeof( Pel ) * width );
  dev.staging( recBytes + horg.size() * sizeof( Pel ) + 512 );                         // grow once, so that the second request does not move the first
  int pitch;
  const int16_t* dRec = stageBordered( dev, rec, recStride, width, height, 4, 0, pitch );
  int16_t* dOrg = dev.staging( recBytes + horg.size() * sizeof( Pel ) + 512 ) + recBytes / sizeof( Pel );
  dev.check( vvhip_upload( dev.ctx(), dOrg, horg.data(), horg.size() * sizeof( Pel ) ), "ALF org plane" );
  const int numClasses = cls ? 25 : 1, ctus = ( ( width + ctuSize - 1 ) / ctuSize ) * ( ( height + ctuSize - 1 ) / ctuSize );
  const size_t nCls = cls ? ( size_t ) ( width / 4 ) * ( height / 4 ) * 2 : 0, outBytes = ( size_t ) ctus * numClasses * VVHIP_ALF_REC * sizeof( float );
  char* aux = static_cast<char*>( dev.stagingAux( ( ( nCls + 255 ) & ~( size_t ) 255 ) + outBytes + 64 ) );
  float* dOut = reinterpret_cast<float*>( aux + ( ( nCls + 255 ) & ~( size_t ) 255 ) );
  if( cls ) dev.check( vvhip_upload( dev.ctx(), aux, cls, nCls ), "ALF classes" );
  if( init ) dev.check( vvhip_upload( dev.ctx(), dOut, init, outBytes ), "ALF start records" );
  dev.check( vvhip_alf_stats_plane( dev.ctx(), dOrg, orgPitch, dRec, pitch, width, height, ctuSize, filterLength, cls ? reinterpret_cast<const uint8_t*>( aux ) : nullptr,
                                    vbCTUHeight, vbPos, init ? dOut : nullptr, dOut ), "vvhip_alf_stats_plane" );
  dev.check( vvhip_download( dev.ctx(), out, dOut, outBytes ), "ALF statistics" );
  return true;
}

bool ALFOps::pictureStatistics( const Pel* const rec[3], const int recStride[3], const Pel* const org[3], const int orgStride[3], int width, int height, int bitDepth,
                                int ctuSize, int unitSize, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3], uint8_t* cls, float* const stats[3] )
{
  if( ( width & 7 ) || ( height & 7 ) || unitSize > 128 || unitSize % ctuSize ) return false;
  Device& dev = Device::get();
  // the planes go up as they lie in the encoder's buffers (their own strides; rec with 4 border rows / columns): six uploads, no host repacking.
  // The unfiltered planes land in this object's resident area (kept for filterPicture), the originals in the context's staging area.
  int w[3], h[3], rp[3], op[3]; size_t rOff[3], oOff[3], recTotal = 0, orgTotal = 0;
  for( int c = 0; c < 3; c++ )
  {
    w[c] = c ? width >> 1 : width; h[c] = c ? height >> 1 : height;
    rp[c] = recStride[c]; op[c] = orgStride[c];
    rOff[c] = recTotal; recTotal += ( ( size_t ) rp[c] * ( h[c] + 8 ) + 8 + 127 ) & ~( size_t ) 127;
  }
  for( int c = 0; c < 3; c++ ) { oOff[c] = orgTotal; orgTotal += ( ( size_t ) op[c] * h[c] + 127 ) & ~( size_t ) 127; }
  m_res.valid = false;
  if( m_res.gpu != dev.gpu() && m_res.dCls )      // the class buffer lives (and is read by the filter kernels) on the same GPU as the planes: it moves with them
  {
    vvhip_free( dev.ctx(), m_res.dCls ); m_res.dCls = nullptr; m_res.clsBytes = 0;
  }
  if( m_res.gpu != dev.gpu() || m_res.elems < recTotal )
  {
    if( m_res.d ) { vvhip_free( dev.ctx(), m_res.d ); m_res.d = nullptr; }      // (hipFree finds the owning device itself)
    void* p = nullptr;
    dev.check( vvhip_malloc( dev.ctx(), &p, recTotal * sizeof( Pel ) + 256 ), "ALF resident planes" );
    m_res.d = static_cast<int16_t*>( p ); m_res.elems = recTotal; m_res.gpu = dev.gpu();
  }
  int16_t* dR = m_res.d;
  int16_t* d = dev.staging( orgTotal * sizeof( Pel ) + 256 );
  for( int c = 0; c < 3; c++ )
  {
    // rows -4 .. h+3, starting 4 samples left of column 0; the last row ends at its right border
    const size_t recElems = ( size_t ) rp[c] * ( h[c] + 7 ) + w[c] + 8;
    Device::pinHost( rec[c] - ( ptrdiff_t ) 4 * rp[c] - 4, recElems * sizeof( Pel ) );
    dev.check( vvhip_upload( dev.ctx(), dR + rOff[c], rec[c] - ( ptrdiff_t ) 4 * rp[c] - 4, recElems * sizeof( Pel ) ), "ALF rec plane" );
    const size_t orgElems = ( size_t ) op[c] * ( h[c] - 1 ) + w[c];
    Device::pinHost( org[c], orgElems * sizeof( Pel ) );
    dev.check( vvhip_upload( dev.ctx(), d + oOff[c], org[c], orgElems * sizeof( Pel ) ), "ALF org plane" );
  }
  const int units = ( ( width + unitSize - 1 ) / unitSize ) * ( ( height + unitSize - 1 ) / unitSize );
  const size_t nCls = ( size_t ) ( width / 4 ) * ( height / 4 ) * 2;
  const size_t stBytes[3] = { ( size_t ) units * 25 * VVHIP_ALF_REC * sizeof( float ), ( size_t ) units * VVHIP_ALF_REC * sizeof( float ), ( size_t ) units * VVHIP_ALF_REC * sizeof( float ) };
  if( m_res.clsBytes < nCls )
  {
    if( m_res.dCls ) vvhip_free( dev.ctx(), m_res.dCls );
    void* p = nullptr;
    dev.check( vvhip_malloc( dev.ctx(), &p, nCls + 256 ), "ALF resident classes" );
    m_res.dCls = static_cast<uint8_t*>( p ); m_res.clsBytes = nCls;
  }
  char* aux = static_cast<char*>( dev.stagingAux( stBytes[0] + stBytes[1] + stBytes[2] + 64 ) );
  uint8_t* dCls = m_res.dCls;
  float* dSt[3] = { reinterpret_cast<float*>( aux ), reinterpret_cast<float*>( aux + stBytes[0] ), reinterpret_cast<float*>( aux + stBytes[0] + stBytes[1] ) };
  const int16_t* dRec[3]; const int16_t* dOrg[3];
  for( int c = 0; c < 3; c++ ) { dRec[c] = dR + rOff[c] + ( size_t ) 4 * rp[c] + 4; dOrg[c] = d + oOff[c]; }     // sample (0,0) of each plane
  dev.check( vvhip_alf_classify( dev.ctx(), dRec[0], rp[0], width, height, bitDepth, vbLumaH, vbLumaPos, dCls ), "vvhip_alf_classify" );
  if( enabled[0] ) dev.check( vvhip_alf_stats_plane_units( dev.ctx(), dOrg[0], op[0], dRec[0], rp[0], width, height, unitSize, ctuSize, 7, dCls, vbLumaH, vbLumaPos, nullptr, dSt[0] ), "vvhip_alf_stats_plane_units" );
  for( int c = 1; c < 3; c++ )
    if( enabled[c] ) dev.check( vvhip_alf_stats_plane_units( dev.ctx(), dOrg[c], op[c], dRec[c], rp[c], w[c], h[c], unitSize >> 1, ctuSize >> 1, 5, nullptr, vbChromaH, vbChromaPos, nullptr, dSt[c] ), "vvhip_alf_stats_plane_units (chroma)" );
  dev.check( vvhip_download( dev.ctx(), cls, dCls, nCls ), "ALF classes" );
  for( int c = 0; c < 3; c++ ) if( enabled[c] ) dev.check( vvhip_download( dev.ctx(), stats[c], dSt[c], stBytes[c] ), "ALF statistics" );
  for( int c = 0; c < 3; c++ ) { m_res.rec[c] = rec[c]; m_res.stride[c] = rp[c]; m_res.off[c] = rOff[c] + ( size_t ) 4 * rp[c] + 4; }
  m_res.width = width; m_res.height = height; m_res.valid = true;
  return true;
}

// ---- the picture call in bands (see the header)
bool ALFOps::statisticsBegin( const int recStride[3], const int orgStride[3], int width, int height, int bitDepth, int ctuSize, int unitSize, int vbLumaH, int vbLumaPos,
                              int vbChromaH, int vbChromaPos, const bool enabled[3] )
{
  Banded& B = m_band;
  B.open = false;
  if( ( width & 7 ) || ( height & 7 ) || unitSize > 128 || unitSize % ctuSize || ( ctuSize & ( ctuSize - 1 ) ) ) return false;
  Device& dev = Device::get();
  size_t recTotal = 0, orgTotal = 0;
  for( int c = 0; c < 3; c++ )
  {
    const int h = c ? height >> 1 : height;
    B.rp[c] = recStride[c]; B.op[c] = orgStride[c]; B.enabled[c] = enabled[c];
    B.rOff[c] = recTotal; recTotal += ( ( size_t ) B.rp[c] * ( h + 8 ) + 8 + 127 ) & ~( size_t ) 127;      // the layout pictureStatistics leaves behind for filterPicture
    B.oOff[c] = orgTotal; orgTotal += ( ( size_t ) B.op[c] * h + 127 ) & ~( size_t ) 127;
  }
  m_res.valid = false;
  if( m_res.gpu != dev.gpu() && m_res.dCls ) { vvhip_free( dev.ctx(), m_res.dCls ); m_res.dCls = nullptr; m_res.clsBytes = 0; }
  if( m_res.gpu != dev.gpu() || m_res.elems < recTotal )
  {
    if( m_res.d ) { vvhip_free( dev.ctx(), m_res.d ); m_res.d = nullptr; }
    void* p = nullptr;
    dev.check( vvhip_malloc( dev.ctx(), &p, recTotal * sizeof( Pel ) + 256 ), "ALF resident planes" );
    m_res.d = static_cast<int16_t*>( p ); m_res.elems = recTotal; m_res.gpu = dev.gpu();
  }
  const int unitsX = ( width + unitSize - 1 ) / unitSize, rows = ( height + unitSize - 1 ) / unitSize;
  const size_t nCls = ( size_t ) ( width / 4 ) * ( height / 4 ) * 2, rec1 = ( size_t ) unitsX * rows * VVHIP_ALF_REC * sizeof( float );
  B.stOff[0] = 0; B.stOff[1] = 25 * rec1; B.stOff[2] = 26 * rec1;
  if( m_res.clsBytes < nCls )
  {
    if( m_res.dCls ) vvhip_free( dev.ctx(), m_res.dCls );
    void* p = nullptr;
    dev.check( vvhip_malloc( dev.ctx(), &p, nCls + 256 ), "ALF resident classes" );
    m_res.dCls = static_cast<uint8_t*>( p ); m_res.clsBytes = nCls;
  }
  if( B.gpu != dev.gpu() || B.orgElems < orgTotal )
  {
    if( B.dOrg ) vvhip_free( dev.ctx(), B.dOrg );
    void* p = nullptr;
    dev.check( vvhip_malloc( dev.ctx(), &p, orgTotal * sizeof( Pel ) + 256 ), "ALF original planes" );
    B.dOrg = static_cast<int16_t*>( p ); B.orgElems = orgTotal;
  }
  if( B.gpu != dev.gpu() || B.stBytes < 27 * rec1 )
  {
    if( B.dSt ) vvhip_free( dev.ctx(), B.dSt );
    void* p = nullptr;
    dev.check( vvhip_malloc( dev.ctx(), &p, 27 * rec1 + 256 ), "ALF statistics records" );
    B.dSt = static_cast<char*>( p ); B.stBytes = 27 * rec1;
  }
  if( B.hostBytes < nCls + 27 * rec1 + 64 )
  {
    if( B.host ) vvhip_host_free( dev.ctx(), B.host );
    void* p = nullptr;
    dev.check( vvhip_host_alloc( dev.ctx(), &p, nCls + 27 * rec1 + 64 ), "ALF statistics download area" );
    B.host = static_cast<char*>( p ); B.hostBytes = nCls + 27 * rec1 + 64;
  }
  while( ( int ) B.events.size() < rows ) { void* e = nullptr; dev.check( vvhip_event_create( dev.ctx(), &e ), "ALF band mark" ); B.events.push_back( e ); }
  B.done.assign( ( size_t ) rows, 0 );
  B.gpu = dev.gpu(); B.rows = rows; B.issued = 0; B.width = width; B.height = height; B.bitDepth = bitDepth; B.ctuSize = ctuSize; B.unitSize = unitSize; B.nCls = nCls;
  B.vbLumaH = vbLumaH; B.vbLumaPos = vbLumaPos; B.vbChromaH = vbChromaH; B.vbChromaPos = vbChromaPos;
  B.open = true;
  return true;
}

bool ALFOps::statisticsBand( int unitRow, const Pel* const rec[3], const Pel* const org[3] )
{
  Banded& B = m_band;
  Device& dev = Device::get();
  if( !B.open || unitRow < 0 || unitRow >= B.rows || B.done[unitRow] || dev.gpu() != B.gpu ) return false;
  const int unitsX = ( B.width + B.unitSize - 1 ) / B.unitSize;
  const size_t clsAt = ( size_t ) ( unitRow * B.unitSize / 4 ) * ( B.width / 4 ) * 2;
  for( int c = 0; c < 3; c++ )
  {
    const int w = c ? B.width >> 1 : B.width, h = c ? B.height >> 1 : B.height, us = c ? B.unitSize >> 1 : B.unitSize, cs = c ? B.ctuSize >> 1 : B.ctuSize;
    const int y0 = unitRow * us, bh = std::min( us, h - y0 ), rp = B.rp[c], op = B.op[c];
    // rows y0 - 4 .. y0 + bh + 3 of the unfiltered plane, from 4 samples left of column 0 to the last row's right border: the band and everything its statistics read;
    // the rows two bands share arrive twice with the same values (both bands' row tasks run after the rows are final)
    const size_t recElems = ( size_t ) rp * ( bh + 7 ) + w + 8, orgElems = ( size_t ) op * ( bh - 1 ) + w;
    const size_t planeElems = ( size_t ) rp * ( h + 7 ) + w + 8;
    Device::pinHost( rec[c] - ( ptrdiff_t ) 4 * rp - 4, planeElems * sizeof( Pel ) );
    Device::pinHost( org[c], ( ( size_t ) op * ( h - 1 ) + w ) * sizeof( Pel ) );
    int16_t* dBandRec = m_res.d + B.rOff[c] + ( size_t ) y0 * rp;                 // <-> host row y0 - 4, column -4
    int16_t* dBandOrg = B.dOrg + B.oOff[c] + ( size_t ) y0 * op;
    dev.check( vvhip_upload( dev.ctx(), dBandRec, rec[c] + ( ptrdiff_t ) ( y0 - 4 ) * rp - 4, recElems * sizeof( Pel ) ), "ALF rec band" );
    dev.check( vvhip_upload( dev.ctx(), dBandOrg, org[c] + ( ptrdiff_t ) y0 * op, orgElems * sizeof( Pel ) ), "ALF org band" );
    const int16_t* dRec = dBandRec + ( size_t ) 4 * rp + 4;                        // sample ( 0, y0 )
    float* dSt = reinterpret_cast<float*>( B.dSt + B.stOff[c] ) + ( size_t ) unitRow * unitsX * ( c ? 1 : 25 ) * VVHIP_ALF_REC;
    if( c == 0 ) dev.check( vvhip_alf_classify( dev.ctx(), dRec, rp, w, bh, B.bitDepth, B.vbLumaH, B.vbLumaPos, m_res.dCls + clsAt ), "vvhip_alf_classify (band)" );
    if( !B.enabled[c] ) continue;
    dev.check( vvhip_alf_stats_plane_units( dev.ctx(), dBandOrg, op, dRec, rp, w, bh, us, cs, c ? 5 : 7, c ? nullptr : m_res.dCls + clsAt, c ? B.vbChromaH : B.vbLumaH,
                                            c ? B.vbChromaPos : B.vbLumaPos, nullptr, dSt ), "vvhip_alf_stats_plane_units (band)" );
    const size_t bytes = ( size_t ) unitsX * ( c ? 1 : 25 ) * VVHIP_ALF_REC * sizeof( float );
    dev.check( vvhip_download_async( dev.ctx(), B.host + B.nCls + B.stOff[c] + ( size_t ) unitRow * bytes, dSt, bytes ), "ALF statistics band" );
  }
  const int y0 = unitRow * B.unitSize, bh = std::min( B.unitSize, B.height - y0 );
  dev.check( vvhip_download_async( dev.ctx(), B.host + clsAt, m_res.dCls + clsAt, ( size_t ) ( bh / 4 ) * ( B.width / 4 ) * 2 ), "ALF classes band" );
  dev.check( vvhip_event_record( dev.ctx(), B.events[unitRow] ), "ALF band mark" );
  B.done[unitRow] = 1; B.issued++;
  return true;
}

bool ALFOps::statisticsEnd( const Pel* const rec[3], const uint8_t** cls, const float* stats[3] )
{
  Banded& B = m_band;
  if( !B.open || B.issued != B.rows ) return false;
  Device& dev = Device::get();
  for( int u = 0; u < B.rows; u++ ) dev.check( vvhip_event_wait( dev.ctx(), B.events[u] ), "ALF band mark" );
  *cls = reinterpret_cast<const uint8_t*>( B.host );
  for( int c = 0; c < 3; c++ ) stats[c] = reinterpret_cast<const float*>( B.host + B.nCls + B.stOff[c] );
  for( int c = 0; c < 3; c++ ) { m_res.rec[c] = rec[c]; m_res.stride[c] = B.rp[c]; m_res.off[c] = B.rOff[c] + ( size_t ) 4 * B.rp[c] + 4; }
  m_res.width = B.width; m_res.height = B.height; m_res.valid = true;      // the unfiltered planes and the classes stay in HBM for filterPicture
  B.open = false;
  return true;
}

ALFOps::~ALFOps() {}      // (device areas are released with the process: HIP teardown order at exit is not ours to rely on)

bool ALFOps::getStatisticsCcAlf( const Pel* orgC, int orgStride, const Pel* slfC, int slfStride, const Pel* recLuma, int recStride, int widthC, int heightC, int ctuSizeC,
                                 int vbCTUHeight, int vbPos, int picHeight, float* out, const float* init )
{
  if( ( widthC & 3 ) || ( heightC & 3 ) || widthC < 4 || heightC < 4 || ctuSizeC > 64 ) return false;
  Device& dev = Device::get();
  const int wL = widthC * 2, hL = heightC * 2;
  const int recPitch = ( wL + 8 + 7 ) & ~7;
  const size_t recBytes = ( ( size_t ) recPitch * ( hL + 8 ) * sizeof( Pel ) + 255 ) & ~( size_t ) 255;
  const int cPitch = ( widthC + 7 ) & ~7;
  std::vector<Pel> hc( ( size_t ) 2 * cPitch * heightC );
  for( int y = 0; y < heightC; y++ )
  {
    memcpy( &hc[( size_t ) y * cPitch], orgC + ( ptrdiff_t ) y * orgStride, sizeof( Pel ) * widthC );
    memcpy( &hc[( size_t ) ( heightC + y ) * cPitch], slfC + ( ptrdiff_t ) y * slfStride, sizeof( Pel ) * widthC );
  }
  dev.staging( recBytes + hc.size() * sizeof( Pel ) + 512 );
  int pitch;
  const int16_t* dRec = stageBordered( dev, recLuma, recStride, wL, hL, 4, 0, pitch );
  int16_t* dC = dev.staging( recBytes + hc.size() * sizeof( Pel ) + 512 ) + recBytes / sizeof( Pel );
  dev.check( vvhip_upload( dev.ctx(), dC, hc.data(), hc.size() * sizeof( Pel ) ), "CC-ALF chroma planes" );
  const int ctus = ( ( widthC + ctuSizeC - 1 ) / ctuSizeC ) * ( ( heightC + ctuSizeC - 1 ) / ctuSizeC );
  const size_t outBytes = ( size_t ) ctus * VVHIP_ALF_REC * sizeof( float );
  float* dOut = static_cast<float*>( dev.stagingAux( outBytes + 64 ) );
  if( init ) dev.check( vvhip_upload( dev.ctx(), dOut, init, outBytes ), "CC-ALF start records" );
  dev.check( vvhip_ccalf_stats_plane( dev.ctx(), dC, cPitch, dC + ( size_t ) cPitch * heightC, cPitch, dRec, pitch, widthC, heightC, ctuSizeC, 1, 1, vbCTUHeight, vbPos, picHeight,
                                      init ? dOut : nullptr, dOut ), "vvhip_ccalf_stats_plane" );
  dev.check( vvhip_download( dev.ctx(), out, dOut, outBytes ), "CC-ALF statistics" );
  return true;
}

bool ALFOps::filterPlane( const Pel* src, int srcStride, Pel* dst, int dstStride, int width, int height, int ctuSize, int bitDepth, int filterLength, const uint8_t* cls,
                          const short* coeffSets, const short* clipSets, int numSets, const short* ctuSet, int vbCTUHeight, int vbPos )
{
  return filterPlaneImpl( src, srcStride, nullptr, nullptr, dst, dstStride, width, height, ctuSize, bitDepth, filterLength, cls, coeffSets, clipSets, numSets, ctuSet, vbCTUHeight, vbPos );
}

bool ALFOps::filterPlaneImpl( const Pel* src, int srcStride, const int16_t* dSrcResident, const uint8_t* dClsResident, Pel* dst, int dstStride, int width, int height, int ctuSize, int bitDepth,
                              int filterLength, const uint8_t* cls, const short* coeffSets, const short* clipSets, int numSets, const short* ctuSet, int vbCTUHeight, int vbPos )
{
  if( ( width & 3 ) || ( height & 3 ) || width < 4 || height < 4 || ( filterLength != 7 && filterLength != 5 ) || ( filterLength == 7 ) != ( cls != nullptr ) || numSets < 1 ) return false;
  Device& dev = Device::get();
  // staging: [src with border (unless resident)][dst compact]; aux: [classes][coefficients][clipping values][CTU sets]
  const int srcPitchGuess = ( width + 8 + 7 ) & ~7;
  const size_t srcBytes = dSrcResident ? 0 : ( ( size_t ) srcPitchGuess * ( height + 8 ) * sizeof( Pel ) + 255 ) & ~( size_t ) 255;
  const int dstPitch = ( width + 7 ) & ~7;
  const size_t dstElems = ( size_t ) dstPitch * height;
  Pel* down = m_down.get( dstElems );
  dev.staging( srcBytes + dstElems * sizeof( Pel ) + 512 );
  int pitch = srcStride;
  const int16_t* dSrc = dSrcResident ? dSrcResident : stageBordered( dev, src, srcStride, width, height, 4, 0, pitch );
  int16_t* dDst = dev.staging( srcBytes + dstElems * sizeof( Pel ) + 512 ) + srcBytes / sizeof( Pel );      // not initialised: only the enabled CTUs are read back
  const int numClasses = cls ? 25 : 1, ctus = ( ( width + ctuSize - 1 ) / ctuSize ) * ( ( height + ctuSize - 1 ) / ctuSize );
  const size_t nCls = ( ( cls && !dClsResident ? ( size_t ) ( width / 4 ) * ( height / 4 ) * 2 : 0 ) + 255 ) & ~( size_t ) 255;
  const size_t nCoef = ( ( size_t ) numSets * numClasses * 13 * sizeof( short ) + 255 ) & ~( size_t ) 255, nSet = ( size_t ) ctus * sizeof( short );
  char* aux = static_cast<char*>( dev.stagingAux( nCls + 2 * nCoef + nSet + 64 ) );
  if( cls && !dClsResident ) dev.check( vvhip_upload( dev.ctx(), aux, cls, ( size_t ) ( width / 4 ) * ( height / 4 ) * 2 ), "ALF classes" );
  dev.check( vvhip_upload( dev.ctx(), aux + nCls, coeffSets, ( size_t ) numSets * numClasses * 13 * sizeof( short ) ), "ALF coefficients" );
  if( clipSets ) dev.check( vvhip_upload( dev.ctx(), aux + nCls + nCoef, clipSets, ( size_t ) numSets * numClasses * 13 * sizeof( short ) ), "ALF clipping values" );
  dev.check( vvhip_upload( dev.ctx(), aux + nCls + 2 * nCoef, ctuSet, nSet ), "ALF CTU filter sets" );
  const uint8_t* dCls = !cls ? nullptr : dClsResident ? dClsResident : reinterpret_cast<const uint8_t*>( aux );
  dev.check( vvhip_alf_filter_plane( dev.ctx(), dSrc, pitch, dDst, dstPitch, width, height, ctuSize, bitDepth, filterLength, dCls,
                                     reinterpret_cast<const int16_t*>( aux + nCls ), clipSets ? reinterpret_cast<const int16_t*>( aux + nCls + nCoef ) : nullptr,
                                     reinterpret_cast<const int16_t*>( aux + nCls + 2 * nCoef ), vbCTUHeight, vbPos ), "vvhip_alf_filter_plane" );
  dev.check( vvhip_download( dev.ctx(), down, dDst, dstElems * sizeof( Pel ) ), "ALF filtered plane" );
  const int ctusX = ( width + ctuSize - 1 ) / ctuSize;
  for( int c = 0; c < ctus; c++ )
  {
    if( ctuSet[c] < 0 ) continue;                                                        // the CTU keeps the caller's samples
    const int x0 = ( c % ctusX ) * ctuSize, y0 = ( c / ctusX ) * ctuSize, w = std::min( ctuSize, width - x0 ), h = std::min( ctuSize, height - y0 );
    for( int y = y0; y < y0 + h; y++ ) memcpy( dst + ( ptrdiff_t ) y * dstStride + x0, down + ( size_t ) y * dstPitch + x0, sizeof( Pel ) * w );
  }
  return true;
}

bool ALFOps::filterPicture( const Pel* const src[3], const int srcStride[3], Pel* const dst[3], const int dstStride[3], int width, int height, int bitDepth, int ctuSize,
                            const uint8_t* cls, const short* lumaCoeff, const short* lumaClip, int numLumaSets, const short* lumaCtuSet,
                            const short* chromaCoeff, const short* chromaClip, int numChromaSets, const short* const chromaCtuSet[2],
                            int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos )
{
  // the planes pictureStatistics left in HBM serve when this call is about the same host planes (and runs on the same GPU)
  const bool res = m_res.valid && m_res.gpu == Device::get().gpu() && m_res.width == width && m_res.height == height &&
                   src[0] == m_res.rec[0] && src[1] == m_res.rec[1] && src[2] == m_res.rec[2] && srcStride[0] == m_res.stride[0] && srcStride[1] == m_res.stride[1] && srcStride[2] == m_res.stride[2];
  if( lumaCtuSet && !filterPlaneImpl( src[0], srcStride[0], res ? m_res.d + m_res.off[0] : nullptr, res ? m_res.dCls : nullptr, dst[0], dstStride[0], width, height, ctuSize, bitDepth, 7, cls,
                                      lumaCoeff, lumaClip, numLumaSets, lumaCtuSet, vbLumaH, vbLumaPos ) ) return false;
  for( int c = 0; c < 2; c++ )
    if( chromaCtuSet[c] && !filterPlaneImpl( src[1 + c], srcStride[1 + c], res ? m_res.d + m_res.off[1 + c] : nullptr, nullptr, dst[1 + c], dstStride[1 + c], width / 2, height / 2, ctuSize / 2,
                                             bitDepth, 5, nullptr, chromaCoeff, chromaClip, numChromaSets, chromaCtuSet[c], vbChromaH, vbChromaPos ) ) return false;
  return true;
}

bool ALFOps::filterCcAlf( Pel* dstC, int dstStride, const Pel* recLuma, int recStride, int widthC, int heightC, int ctuSizeC, int bitDepth, const int16_t* coeff, int numFilters,
                          const uint8_t* ctuFilter, int vbCTUHeight, int vbPos )
{
  if( widthC < 1 || heightC < 1 || numFilters < 1 ) return false;
  Device& dev = Device::get();
  const int wL = widthC * 2, hL = heightC * 2;
  const int recPitch = ( wL + 8 + 7 ) & ~7;
  const size_t recBytes = ( ( size_t ) recPitch * ( hL + 8 ) * sizeof( Pel ) + 255 ) & ~( size_t ) 255;
  const int cPitch = ( widthC + 7 ) & ~7;
  std::vector<Pel> hc( ( size_t ) cPitch * heightC );
  for( int y = 0; y < heightC; y++ ) memcpy( &hc[( size_t ) y * cPitch], dstC + ( ptrdiff_t ) y * dstStride, sizeof( Pel ) * widthC );
  dev.staging( recBytes + hc.size() * sizeof( Pel ) + 512 );
  int pitch;
  const int16_t* dRec = stageBordered( dev, recLuma, recStride, wL, hL, 4, 0, pitch );
  int16_t* dC = dev.staging( recBytes + hc.size() * sizeof( Pel ) + 512 ) + recBytes / sizeof( Pel );
  dev.check( vvhip_upload( dev.ctx(), dC, hc.data(), hc.size() * sizeof( Pel ) ), "CC-ALF chroma plane" );
  const int ctus = ( ( widthC + ctuSizeC - 1 ) / ctuSizeC ) * ( ( heightC + ctuSizeC - 1 ) / ctuSizeC );
  const size_t nCoef = ( ( size_t ) numFilters * 8 * sizeof( int16_t ) + 255 ) & ~( size_t ) 255;
  char* aux = static_cast<char*>( dev.stagingAux( nCoef + ctus + 64 ) );
  dev.check( vvhip_upload( dev.ctx(), aux, coeff, ( size_t ) numFilters * 8 * sizeof( int16_t ) ), "CC-ALF coefficients" );
  dev.check( vvhip_upload( dev.ctx(), aux + nCoef, ctuFilter, ctus ), "CC-ALF filter control" );
  dev.check( vvhip_ccalf_filter_plane( dev.ctx(), dC, cPitch, dRec, pitch, widthC, heightC, ctuSizeC, 1, 1, bitDepth, reinterpret_cast<const int16_t*>( aux ),
                                       reinterpret_cast<const uint8_t*>( aux + nCoef ), vbCTUHeight, vbPos ), "vvhip_ccalf_filter_plane" );
  dev.check( vvhip_download( dev.ctx(), hc.data(), dC, hc.size() * sizeof( Pel ) ), "CC-ALF corrected plane" );
  for( int y = 0; y < heightC; y++ ) memcpy( dstC + ( ptrdiff_t ) y * dstStride, &hc[( size_t ) y * cPitch], sizeof( Pel ) * widthC );
  return true;
}

// ------------------------------------------------------------------------------------------------ MCTFOps
namespace {

// the reference hands the kernels ROWS of its static filter tables (MCTF.cpp:1142-1143,1157-1158); recover the phase from the taps
const int16_t kF4[16][4] = { { 0, 64, 0, 0 }, { -2, 62, 4, 0 }, { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
                             { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, { 0, 4, 62, -2 } };
const int16_t kF8[16][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 64, 4, -2, 0, 0 }, { 0, 1, -6, 62, 9, -3, 1, 0 }, { 0, 2, -8, 60, 14, -5, 1, 0 }, { 0, 2, -9, 57, 19, -7, 2, 0 },
                             { 0, 3, -10, 53, 24, -8, 2, 0 }, { 0, 3, -11, 50, 29, -9, 2, 0 }, { 0, 3, -11, 44, 35, -10, 3, 0 }, { 0, 1, -7, 38, 38, -7, 1, 0 }, { 0, 3, -10, 35, 44, -11, 3, 0 },
                             { 0, 2, -9, 29, 50, -11, 3, 0 }, { 0, 2, -8, 24, 53, -10, 3, 0 }, { 0, 2, -7, 19, 57, -9, 2, 0 }, { 0, 1, -5, 14, 60, -8, 2, 0 }, { 0, 1, -3, 9, 62, -6, 1, 0 }, { 0, 0, -2, 4, 64, -3, 1, 0 } };

int phaseOf( const int16_t* f, bool tap4 )
{
  for( int p = 0; p < 16; p++ )
    if( tap4 ? !memcmp( f, kF4[p], sizeof( kF4[p] ) ) : !memcmp( f, kF8[p], sizeof( kF8[p] ) ) ) return p;
  throw Exception( "motionErrorLumaFrac: filter row is not one of MCTF::m_interpolationFilter{4,8}" );
}

int errorOne( const Pel* org, ptrdiff_t os, const Pel* buf, ptrdiff_t bs, int w, int h, int fx, int fy, bool tap4, int bitDepth )
{
  Device& dev = Device::get();
  const int M = 4;      // halo for the 6-tap filter: 2 left/above, 3 right/below
  int16_t* cursor = dev.staging( ( size_t ) 2 * ( w + 2 * M ) * ( h + 2 * M ) * sizeof( Pel ) + 128 );
  struct Io { vvhip_mctf_item it; int32_t out; } io;
  const int16_t* dOrg; const int16_t* dBuf; int sOrg, sBuf;
  std::vector<Pel> t;
  const auto mo = dev.find( org );
  const auto mb = dev.find( buf );
  if( mo && mo->stride == os ) { dOrg = mo->dOrigin; sOrg = mo->stride; io.it.org_off = ( int32_t ) ( org - mo->origin ); }
  else
  {
    t.resize( ( size_t ) w * h );
    for( int y = 0; y < h; y++ ) memcpy( &t[( size_t ) y * w], org + y * os, sizeof( Pel ) * w );
    dev.check( vvhip_upload( dev.ctx(), cursor, t.data(), t.size() * sizeof( Pel ) ), "mctf org" );
    dev.check( vvhip_sync( dev.ctx() ), "mctf org" );
    dOrg = cursor; sOrg = w; io.it.org_off = 0; cursor += ( t.size() + 7 ) & ~( size_t ) 7;
  }
  if( mb && mb->stride == bs ) { dBuf = mb->dOrigin; sBuf = mb->stride; io.it.buf_off = ( int32_t ) ( buf - mb->origin ); }
  else
  {
    const int ww = w + 2 * M, hh = h + 2 * M;
    t.resize( ( size_t ) ww * hh );
    for( int y = 0; y < hh; y++ ) memcpy( &t[( size_t ) y * ww], buf + ( y - M ) * bs - M, sizeof( Pel ) * ww );
    dev.check( vvhip_upload( dev.ctx(), cursor, t.data(), t.size() * sizeof( Pel ) ), "mctf buf" );
    dev.check( vvhip_sync( dev.ctx() ), "mctf buf" );
    dBuf = cursor; sBuf = ww; io.it.buf_off = M * ww + M;
  }
  io.it.fx = ( int16_t ) fx; io.it.fy = ( int16_t ) fy; io.out = 0;
  char* aux = static_cast<char*>( dev.stagingAux( sizeof( Io ) ) );
  dev.check( vvhip_upload( dev.ctx(), aux, &io, sizeof( io ) ), "mctf item" );
  dev.check( vvhip_mctf_error_batch( dev.ctx(), dOrg, sOrg, dBuf, sBuf, w, h, tap4 ? 1 : 0, bitDepth, reinterpret_cast<vvhip_mctf_item*>( aux ), 1,
                                     reinterpret_cast<int32_t*>( aux + offsetof( Io, out ) ) ), "vvhip_mctf_error_batch" );
  dev.check( vvhip_download( dev.ctx(), &io.out, aux + offsetof( Io, out ), sizeof( int32_t ) ), "mctf result" );
  return io.out;
}

int errInt( const Pel* org, const ptrdiff_t os, const Pel* buf, const ptrdiff_t bs, const int w, const int h, const int /*besterror*/ )
{
  return errorOne( org, os, buf, bs, w, h, 0, 0, true, 10 );
}
int errFrac6( const Pel* org, const ptrdiff_t os, const Pel* buf, const ptrdiff_t bs, const int w, const int h, const int16_t* xf, const int16_t* yf, const int bd, const int )
{
  return errorOne( org, os, buf, bs, w, h, phaseOf( xf, false ), phaseOf( yf, false ), false, bd );
}
int errFrac4( const Pel* org, const ptrdiff_t os, const Pel* buf, const ptrdiff_t bs, const int w, const int h, const int16_t* xf, const int16_t* yf, const int bd, const int )
{
  return errorOne( org, os, buf, bs, w, h, phaseOf( xf, true ), phaseOf( yf, true ), true, bd );
}

double calcVarOne( const Pel* org, const ptrdiff_t os, const int w, const int h )
{
  Device& dev = Device::get();
  std::vector<Pel> t( ( size_t ) w * h );
  for( int y = 0; y < h; y++ ) memcpy( &t[( size_t ) y * w], org + y * os, sizeof( Pel ) * w );
  int16_t* d = dev.staging( t.size() * sizeof( Pel ) + 64 );
  struct Io { int32_t off; int32_t pad; int64_t out; } io = { 0, 0, 0 };
  char* aux = static_cast<char*>( dev.stagingAux( sizeof( Io ) ) );
  dev.check( vvhip_upload( dev.ctx(), d, t.data(), t.size() * sizeof( Pel ) ), "calcVar" );
  dev.check( vvhip_upload( dev.ctx(), aux, &io, sizeof( io ) ), "calcVar" );
  dev.check( vvhip_mctf_calc_var_batch( dev.ctx(), d, w, w, h, reinterpret_cast<int32_t*>( aux ), 1, reinterpret_cast<int64_t*>( aux + offsetof( Io, out ) ) ), "vvhip_mctf_calc_var_batch" );
  dev.check( vvhip_download( dev.ctx(), &io.out, aux + offsetof( Io, out ), sizeof( int64_t ) ), "calcVar" );
  return io.out / 256.0;      // MCTF.cpp:545
}

} // namespace

MCTFOps::MCTFOps()
{
  Device::get();
  m_motionErrorLumaInt8 = errInt;
  m_motionErrorLumaFrac8[0] = errFrac6;
  m_motionErrorLumaFrac8[1] = errFrac4;
  m_calcVar = calcVarOne;
}

void MCTFOps::motionEstimation( int curPicId, const int* refPicIds, int nRefs, int bitDepth, int unitSize, int mctfSpeed, bool addLevel, vvhip_mv** out )
{
  Device& dev = Device::get();
  const Device::Mirror& cur = dev.mirror( curPicId );
  std::vector<const int16_t*> refs( nRefs );
  for( int r = 0; r < nRefs; r++ )
  {
    const Device::Mirror& m = dev.mirror( refPicIds[r] );
    if( m.stride != cur.stride || m.width != cur.width || m.height != cur.height || m.margin != cur.margin )
      throw Exception( "MCTFOps::motionEstimation: reference picture geometry differs from the current picture" );
    refs[r] = m.dOrigin;
  }
  const size_t count = ( size_t ) ( ( cur.width + unitSize - 1 ) / unitSize ) * ( ( cur.height + unitSize - 1 ) / unitSize );
  vvhip_mv* dOut = static_cast<vvhip_mv*>( dev.stagingAux( count * nRefs * sizeof( vvhip_mv ) + 64 ) );
  std::vector<vvhip_mv*> outs( nRefs );
  for( int r = 0; r < nRefs; r++ ) outs[r] = dOut + count * r;
  dev.check( vvhip_mctf_motion_estimation( dev.ctx(), cur.dOrigin, refs.data(), nRefs, cur.stride, cur.width, cur.height, cur.margin, bitDepth, unitSize,
                                           mctfSpeed, addLevel ? 1 : 0, outs.data() ), "vvhip_mctf_motion_estimation" );
  for( int r = 0; r < nRefs; r++ ) dev.check( vvhip_download( dev.ctx(), out[r], outs[r], count * sizeof( vvhip_mv ) ), "motion vectors" );
}

void MCTFOps::bilateralFilter( const int* orgIds, const int* refIds, int nRefs, const vvhip_mv* const* mvs, const double* refStrengths, int qp, int bitDepth, int unitSize,
                               bool lowResFltApply, double overallStrength, int numComp, Pel* const* out, const int* outStride )
{
  Device& dev = Device::get();
  const Device::Mirror& y = dev.mirror( orgIds[0] );
  const int mvW = ( y.width + unitSize - 1 ) / unitSize, mvH = ( y.height + unitSize - 1 ) / unitSize;
  const size_t count = ( size_t ) mvW * mvH;
  vvhip_mv* dMv = static_cast<vvhip_mv*>( dev.stagingAux( count * nRefs * sizeof( vvhip_mv ) + 64 ) );
  std::vector<const vvhip_mv*> dMvs( nRefs );
  for( int r = 0; r < nRefs; r++ ) { dev.check( vvhip_upload( dev.ctx(), dMv + count * r, mvs[r], count * sizeof( vvhip_mv ) ), "motion vectors" ); dMvs[r] = dMv + count * r; }
  static thread_local PinnedBuffer tmpBuf;       // download area of this thread
  for( int c = 0; c < numComp; c++ )
  {
    const Device::Mirror& o = dev.mirror( orgIds[c] );
    std::vector<const int16_t*> refs( nRefs );
    for( int r = 0; r < nRefs; r++ )
    {
      const Device::Mirror& m = dev.mirror( refIds[3 * r + c] );
      if( m.stride != o.stride || m.width != o.width || m.height != o.height ) throw Exception( "MCTFOps::bilateralFilter: reference plane geometry differs from the original's" );
      refs[r] = m.dOrigin;
    }
    double sigmaSq, weightScaling;
    dev.check( vvhip_mctf_filter_params( qp, bitDepth, overallStrength, c > 0, &sigmaSq, &weightScaling ), "vvhip_mctf_filter_params" );
    const size_t elems = ( size_t ) o.width * o.height;
    int16_t* dOut = dev.staging( elems * sizeof( Pel ) + 64 );
    dev.check( vvhip_mctf_apply_plane( dev.ctx(), o.dOrigin, o.stride, o.width, o.height, c > 0 ? 1 : 0, bitDepth, unitSize, lowResFltApply ? 1 : 0, qp, nRefs, refs.data(), o.stride,
                                       dMvs.data(), mvW, refStrengths, weightScaling, sigmaSq, dOut, o.width ), "vvhip_mctf_apply_plane" );
    Pel* tmp = tmpBuf.get( elems );
    dev.check( vvhip_download( dev.ctx(), tmp, dOut, elems * sizeof( Pel ) ), "filtered plane" );
    for( int r = 0; r < o.height; r++ ) memcpy( out[c] + ( ptrdiff_t ) r * outStride[c], tmp + ( size_t ) r * o.width, sizeof( Pel ) * o.width );
  }
}

// ------------------------------------------------------------------------------------------------ QuantOps
namespace {

void deQuantOne( const int maxX, const int maxY, const int scale, const TCoeffSig* const q, const size_t qStride, TCoeff* const coef, const int rightShift, const int inputMaximum, const TCoeff transformMaximum )
{
  Device& dev = Device::get();
  const int w = maxX + 1, h = maxY + 1;
  std::vector<TCoeffSig> t( ( size_t ) w * h );
  for( int y = 0; y < h; y++ ) memcpy( &t[( size_t ) y * w], q + y * qStride, sizeof( TCoeffSig ) * w );
  int16_t* dQ = dev.staging( t.size() * sizeof( TCoeffSig ) + 64 );
  int32_t* dC = static_cast<int32_t*>( dev.stagingAux( t.size() * sizeof( TCoeff ) + 64 ) );
  dev.check( vvhip_upload( dev.ctx(), dQ, t.data(), t.size() * sizeof( TCoeffSig ) ), "xDeQuant" );
  dev.check( vvhip_dequant_core( dev.ctx(), maxX, maxY, scale, dQ, ( size_t ) w, dC, rightShift, inputMaximum, transformMaximum ), "vvhip_dequant_core" );
  dev.check( vvhip_download( dev.ctx(), coef, dC, t.size() * sizeof( TCoeff ) ), "xDeQuant" );
}

bool needRdoqOne( const TCoeff* c, size_t num, int quantCoeff, int64_t offset, int shift )
{
  Device& dev = Device::get();
  char* aux = static_cast<char*>( dev.stagingAux( 64 + num * sizeof( TCoeff ) ) );
  uint8_t need = 0;
  dev.check( vvhip_upload( dev.ctx(), aux + 64, c, num * sizeof( TCoeff ) ), "xNeedRdoq" );
  dev.check( vvhip_need_rdoq_core( dev.ctx(), reinterpret_cast<int32_t*>( aux + 64 ), num, quantCoeff, offset, shift, reinterpret_cast<uint8_t*>( aux ) ), "vvhip_need_rdoq_core" );
  dev.check( vvhip_download( dev.ctx(), &need, aux, 1 ), "xNeedRdoq" );
  return need != 0;
}

void quantImpl( unsigned w, unsigned h, const TCoeff* coef, TCoeffSig* q, TCoeff& absSum, int& lastScanPos, TCoeff* deltaU, const int qp, const bool isIRAP, const int bitDepth, const TCoeff thrVal,
                bool raw, int rawScale, int rawQBits, int64_t rawAdd, int lfnstIdx = 0 );

void quantOne( unsigned w, unsigned h, const TCoeff* coef, TCoeffSig* q, TCoeff& absSum, int& lastScanPos, TCoeff* deltaU, const int qp, const bool isIRAP, const int bitDepth, const TCoeff thrVal )
{
  quantImpl( w, h, coef, q, absSum, lastScanPos, deltaU, qp, isIRAP, bitDepth, thrVal, false, 0, 0, 0 );
}
void quantCoreOne( unsigned w, unsigned h, const TCoeff* coef, TCoeffSig* q, TCoeff& absSum, int& lastScanPos, TCoeff* deltaU, const int quantCoeff, const int iQBits, const int64_t iAdd, const TCoeff thrVal )
{
  quantImpl( w, h, coef, q, absSum, lastScanPos, deltaU, 0, false, 10, thrVal, true, quantCoeff, iQBits, iAdd );
}

void quantCoreLfnstOne( unsigned w, unsigned h, const TCoeff* coef, TCoeffSig* q, TCoeff& absSum, int& lastScanPos, TCoeff* deltaU, const int quantCoeff, const int iQBits, const int64_t iAdd, const TCoeff thrVal,
                        const int lfnstIdx )
{
  quantImpl( w, h, coef, q, absSum, lastScanPos, deltaU, 0, false, 10, thrVal, true, quantCoeff, iQBits, iAdd, lfnstIdx );
}

void quantImpl( unsigned w, unsigned h, const TCoeff* coef, TCoeffSig* q, TCoeff& absSum, int& lastScanPos, TCoeff* deltaU, const int qp, const bool isIRAP, const int bitDepth, const TCoeff thrVal,
                bool raw, int rawScale, int rawQBits, int64_t rawAdd, int lfnstIdx )
{
  Device& dev = Device::get();
  const size_t area = ( size_t ) w * h;
  struct Hdr { vvhip_tu_qp qp; int32_t absSum, last; } hdr; hdr.qp.qp = ( int16_t ) qp; hdr.qp.flags = ( int16_t ) ( ( isIRAP ? 1 : 0 ) | 2 ); hdr.absSum = 0; hdr.last = 0;
  char* aux = static_cast<char*>( dev.stagingAux( 64 + area * ( sizeof( TCoeff ) * 2 + sizeof( TCoeffSig ) ) + 64 ) );
  int32_t* dCoef = reinterpret_cast<int32_t*>( aux + 64 );
  int32_t* dDu = dCoef + area;
  int16_t* dLev = reinterpret_cast<int16_t*>( dDu + area );
  dev.check( vvhip_upload( dev.ctx(), aux, &hdr, sizeof( hdr ) ), "quant" );
  dev.check( vvhip_upload( dev.ctx(), dCoef, coef, area * sizeof( TCoeff ) ), "quant" );
  if( raw )
    dev.check( vvhip_quant_core_lfnst( dev.ctx(), dCoef, ( int ) w, ( int ) h, rawScale, rawQBits, rawAdd, thrVal, lfnstIdx, dLev, deltaU ? dDu : nullptr,
                                       reinterpret_cast<int32_t*>( aux + offsetof( Hdr, absSum ) ), reinterpret_cast<int32_t*>( aux + offsetof( Hdr, last ) ) ), "vvhip_quant_core" );
  else
  dev.check( vvhip_quant_batch( dev.ctx(), dCoef, 1, ( int ) w, ( int ) h, bitDepth, reinterpret_cast<vvhip_tu_qp*>( aux ), thrVal, dLev, deltaU ? dDu : nullptr,
                                reinterpret_cast<int32_t*>( aux + offsetof( Hdr, absSum ) ), reinterpret_cast<int32_t*>( aux + offsetof( Hdr, last ) ) ), "vvhip_quant_batch" );
  dev.check( vvhip_download( dev.ctx(), q, dLev, area * sizeof( TCoeffSig ) ), "quant" );
  if( deltaU ) dev.check( vvhip_download( dev.ctx(), deltaU, dDu, area * sizeof( TCoeff ) ), "quant" );
  dev.check( vvhip_download( dev.ctx(), &hdr, aux, sizeof( hdr ) ), "quant" );
  absSum = hdr.absSum; lastScanPos = hdr.last;
}

} // namespace

QuantOps::QuantOps()
{
  xDeQuant = deQuantOne;
  xNeedRdoq = needRdoqOne;
  xQuant = quantOne;
  xQuantCore = quantCoreOne;
  xQuantCoreLfnst = quantCoreLfnstOne;
}

} // namespace vvhip
