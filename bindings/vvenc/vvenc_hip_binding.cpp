// vvenc_hip_binding.cpp — the VVenC side of the MI355X binding (PRODUCT; built into bindings/vvenc/_build/libvvenc_hip_enc.so by bindings/vvenc/Makefile).
// Bodies of the hooks apply_binding.py inserts into the encoder (vvenc_hip_binding.h): they point the reference's kernel tables and picture-level stages at the
// table-shaped C++ shim (vvenc_amd/csrc/host/vvenc_hip_shim.h) above the C ABI — per-call trampolines for RdCost::m_afpDistortFunc / m_afpDistortFuncX5,
// Quant::xQuant / xDeQuant / xNeedRdoq, MCTF::m_motionErrorLuma* / m_calcVar, the fused 2-D transform entries of TrQuant::xT / xIT, the batched search sites, the
// whole-picture MCTF and ALF stages, picture residency and the picture -> GPU mapping.  INTEGRATION.md section 2 walks through it.
#include <cstdint>
#include <cstring>
#include <sstream>
#include <iostream>
#include <fstream>
#include <string>
#include <vector>
#include <list>
#include <map>
#include <array>
#include <deque>
#include <mutex>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <algorithm>
#include <memory>
#include <chrono>
#include <cmath>
#include <limits>
#include <cassert>
#include <cstdarg>
#include <iomanip>
#include <numeric>
#include <set>
#include <unordered_map>
#include <stdexcept>
#include <exception>
#include <utility>

#define private public
#define protected public
#include "CommonLib/CommonDef.h"
#include "CommonLib/Unit.h"
#include "CommonLib/RdCost.h"
#include "CommonLib/Quant.h"
#include "CommonLib/MCTF.h"
#include "CommonLib/TrQuant_EMT.h"
#include "CommonLib/InterpolationFilter.h"
#include "CommonLib/Mv.h"
#include "CommonLib/Picture.h"
#include "EncoderLib/EncCfg.h"
#undef private
#undef protected

#include "vvenc_hip_binding.h"
#include "vvenc_hip_recorder.h"
#include "../../vvenc_amd/csrc/host/vvenc_hip_shim.h"
#include "EncoderLib/EncStage.h"

VvhipHooks g_vvhipHooks = {};

namespace {

vvhip::RdCost*   g_rd = nullptr;
vvhip::QuantOps* g_q  = nullptr;
vvhip::MCTFOps*  g_m  = nullptr;
std::atomic<uint64_t> g_calls[10];    // dist, x5, fwd, inv, quant, dequant, needrdoq, mctf, interpolation, whole-picture MCTF filter
std::atomic<uint64_t> g_lfnstQuantFallbacks{ 0 }, g_lfnstQuantDevice{ 0 }, g_mctfDeviceCalls{ 0 };

vvhip::DistParam conv( const vvenc::DistParam& dp )
{
  vvhip::DistParam d;
  d.org.buf = dp.org.buf; d.org.stride = ( int ) dp.org.stride; d.org.width = dp.org.width; d.org.height = dp.org.height;
  d.cur.buf = dp.cur.buf; d.cur.stride = ( int ) dp.cur.stride; d.cur.width = dp.org.width; d.cur.height = dp.org.height;
  d.bitDepth = dp.bitDepth; d.subShift = dp.subShift; d.applyWeight = dp.applyWeight;
  d.mask = dp.mask; d.maskStride = dp.maskStride; d.stepX = dp.stepX; d.maskStride2 = dp.maskStride2;
  return d;
}

template<int IDX> vvenc::Distortion distTramp( const vvenc::DistParam& dp )
{
  g_calls[0]++;
  return g_rd->m_afpDistortFunc[0][IDX]( conv( dp ) );
}
template<int IDX> void x5Tramp( const vvenc::DistParam& dp, vvenc::Distortion* cost, bool calcCentre )
{
  g_calls[1]++;
  vvhip::Distortion c[5]; for( int i = 0; i < 5; i++ ) c[i] = cost[i];
  g_rd->m_afpDistortFuncX5[IDX]( conv( dp ), c, calcCentre );
  for( int i = 0; i < 5; i++ ) cost[i] = c[i];
}

template<int I> struct Fill { static void go( vvenc::RdCost* rc ) { rc->m_afpDistortFunc[0][I] = distTramp<I>; Fill<I - 1>::go( rc ); } };
vvenc::Distortion fxdWtdTramp( const vvenc::DistParam& dp, uint32_t fixedWeight )
{
  g_calls[0]++;
  return g_rd->m_fxdWtdPredPtr( conv( dp ), fixedWeight );
}
template<> struct Fill<-1> { static void go( vvenc::RdCost* ) {} };

void initRdCost( vvenc::RdCost* rc )
{
  static_assert( ( int ) vvenc::DF_TOTAL_FUNCTIONS == ( int ) vvhip::DF_TOTAL_FUNCTIONS && ( int ) vvenc::DF_HAD_fast == ( int ) vvhip::DF_HAD_fast &&
                 ( int ) vvenc::DF_HAD_2SAD == ( int ) vvhip::DF_HAD_2SAD && ( int ) vvenc::DF_SAD == ( int ) vvhip::DF_SAD, "DFunc numbering" );
  Fill<vvenc::DF_TOTAL_FUNCTIONS - 1>::go( rc );      // row 0 (bitDepth <= 10); row 1 stays the scalar copy made before SIMD init (RdCost.cpp:126)
  rc->m_afpDistortFuncX5[0] = x5Tramp<0>;
  rc->m_afpDistortFuncX5[1] = x5Tramp<1>;
  rc->m_fxdWtdPredPtr = fxdWtdTramp;
}

// the CPU entry this binding replaced (the same function for every Quant object).  Until round 6 LFNST TUs kept going there; QuantCore's first-coefficient-group rule for
// them (iCGNum = 1, Quant.cpp:149-159) is on the device now (vvhip_quant_core_lfnst) — $VVHIP_LFNST_QUANT_ON_CPU=1 restores the old route (counter 20 counts those TUs)
decltype( vvenc::Quant::xQuant ) g_cpuQuant = nullptr;
static const bool g_lfnstOnCpu = []{ const char* e = getenv( "VVHIP_LFNST_QUANT_ON_CPU" ); return e && atoi( e ) != 0; }();

void xQuantTramp( const vvenc::TransformUnit tu, const vvenc::ComponentID compID, const vvenc::CCoeffBuf& piCoef, vvenc::CoeffSigBuf piQCoef, vvenc::TCoeff& uiAbsSum,
                  int& lastScanPos, vvenc::TCoeff* deltaU, const int defaultQuantisationCoefficient, const int iQBits, const int64_t iAdd,
                  const vvenc::TCoeff entropyCodingMinimum, const vvenc::TCoeff entropyCodingMaximum, const bool signHiding, const vvenc::TCoeff m_thrVal )
{
  if( tu.cu->lfnstIdx && g_cpuQuant && g_lfnstOnCpu )
  {
    g_lfnstQuantFallbacks++;
    g_cpuQuant( tu, compID, piCoef, piQCoef, uiAbsSum, lastScanPos, deltaU, defaultQuantisationCoefficient, iQBits, iAdd, entropyCodingMinimum, entropyCodingMaximum, signHiding, m_thrVal );
    return;
  }
  g_calls[4]++;
  const unsigned w = tu.blocks[compID].width, h = tu.blocks[compID].height;
  std::vector<vvenc::TCoeffSig> lev( ( size_t ) w * h );
  std::vector<vvenc::TCoeff> src( ( size_t ) w * h );
  for( unsigned y = 0; y < h; y++ ) memcpy( &src[( size_t ) y * w], piCoef.buf + y * piCoef.stride, sizeof( vvenc::TCoeff ) * w );
  if( tu.cu->lfnstIdx ) { g_lfnstQuantDevice++; g_q->xQuantCoreLfnst( w, h, src.data(), lev.data(), uiAbsSum, lastScanPos, deltaU, defaultQuantisationCoefficient, iQBits, iAdd, m_thrVal, tu.cu->lfnstIdx ); }
  else g_q->xQuantCore( w, h, src.data(), lev.data(), uiAbsSum, lastScanPos, deltaU, defaultQuantisationCoefficient, iQBits, iAdd, m_thrVal );
  for( unsigned y = 0; y < h; y++ ) memcpy( piQCoef.buf + y * piQCoef.stride, &lev[( size_t ) y * w], sizeof( vvenc::TCoeffSig ) * w );
}
void xDeQuantTramp( const int maxX, const int maxY, const int scale, const vvenc::TCoeffSig* const q, const size_t qStride, vvenc::TCoeff* const coef,
                    const int rightShift, const int inputMaximum, const vvenc::TCoeff transformMaximum )
{
  g_calls[5]++;
  g_q->xDeQuant( maxX, maxY, scale, q, qStride, coef, rightShift, inputMaximum, transformMaximum );
}
bool xNeedRdoqTramp( const vvenc::TCoeff* c, size_t n, int qc, int64_t off, int shift ) { g_calls[6]++; return g_q->xNeedRdoq( c, n, qc, off, shift ); }

void initQuant( vvenc::Quant* q )
{
  if( q->xQuant != xQuantTramp ) g_cpuQuant = q->xQuant;
  q->xQuant = xQuantTramp;
  q->xDeQuant = xDeQuantTramp;
  q->xNeedRdoq = xNeedRdoqTramp;
}

int errInt( const vvenc::Pel* o, const ptrdiff_t os, const vvenc::Pel* b, const ptrdiff_t bs, const int w, const int h, const int best ) { g_calls[7]++; return g_m->m_motionErrorLumaInt8( o, os, b, bs, w, h, best ); }
int errF6( const vvenc::Pel* o, const ptrdiff_t os, const vvenc::Pel* b, const ptrdiff_t bs, const int w, const int h, const int16_t* xf, const int16_t* yf, const int bd, const int best )
{ g_calls[7]++; return g_m->m_motionErrorLumaFrac8[0]( o, os, b, bs, w, h, xf, yf, bd, best ); }
int errF4( const vvenc::Pel* o, const ptrdiff_t os, const vvenc::Pel* b, const ptrdiff_t bs, const int w, const int h, const int16_t* xf, const int16_t* yf, const int bd, const int best )
{ g_calls[7]++; return g_m->m_motionErrorLumaFrac8[1]( o, os, b, bs, w, h, xf, yf, bd, best ); }
double calcVar( const vvenc::Pel* o, const ptrdiff_t os, const int w, const int h ) { return g_m->m_calcVar( o, os, w, h ); }

void initMCTF( vvenc::MCTF* m )
{
  m->m_motionErrorLumaInt8 = errInt;
  m->m_motionErrorLumaFrac8[0] = errF6;
  m->m_motionErrorLumaFrac8[1] = errF4;
  m->m_calcVar = calcVar;
}

bool fwd2D( const int16_t* resi, ptrdiff_t stride, int32_t* coef, unsigned w, unsigned h, int th, int tv, int bd )
{
  g_calls[2]++;
  vvhip::g_tCoeffOps.fwdTransform2D( resi, stride, coef, w, h, th, tv, bd );
  return true;
}
bool inv2D( const int32_t* coef, int16_t* resi, ptrdiff_t stride, unsigned w, unsigned h, int th, int tv, int bd )
{
  g_calls[3]++;
  vvhip::g_tCoeffOps.invTransform2D( coef, resi, stride, w, h, th, tv, bd );
  return true;
}

// ---- residual loop of a CU (InterSearch::xEstimateInterResidualQT, EncoderLib/InterSearch.cpp:3584-3714): the DCT-2 forward transforms of its component TUs in ONE device
// round trip (one upload of the compact residual blocks, one launch per distinct TU size, one download), kept per worker thread until TrQuant::xT asks for them.
struct TuMemo { int w = 0, h = 0; std::vector<int16_t> resi; std::vector<int32_t> coef; };
thread_local TuMemo t_tuMemo[3];
thread_local int t_tuMemoN = 0;
std::atomic<uint64_t> g_tuPrefetches{ 0 }, g_tuLookups{ 0 }, g_tuHits{ 0 }, g_tuPrefetchNs{ 0 };
struct SiteTimer { std::atomic<uint64_t>& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit SiteTimer( std::atomic<uint64_t>& a ) : acc( a ) {}
  ~SiteTimer() { acc += ( uint64_t ) std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now() - t0 ).count(); } };

void tuPrefetch( const int16_t* const resi[3], const int strides[3], const int widths[3], const int heights[3], int n, int bitDepth )
{
  t_tuMemoN = 0;
  SiteTimer timer( g_tuPrefetchNs );
  vvhip::Device& dev = vvhip::Device::get();
  size_t total = 0;
  for( int c = 0; c < n; c++ )
  {
    const int w = widths[c], h = heights[c];
    if( w != h || ( w != 4 && w != 8 && w != 16 && w != 32 && w != 64 ) ) return;                        // square TUs of the fused path only; anything else stays on the CPU
    total += ( size_t ) w * h;
  }
  std::vector<int16_t> host( total );
  std::vector<int32_t> offs( n );
  size_t at = 0;
  for( int c = 0; c < n; c++ )
  {
    TuMemo& m = t_tuMemo[c]; m.w = widths[c]; m.h = heights[c];
    m.resi.resize( ( size_t ) m.w * m.h ); m.coef.resize( ( size_t ) m.w * m.h );
    for( int y = 0; y < m.h; y++ ) memcpy( &m.resi[( size_t ) y * m.w], resi[c] + ( ptrdiff_t ) y * strides[c], sizeof( int16_t ) * m.w );
    memcpy( &host[at], m.resi.data(), sizeof( int16_t ) * m.resi.size() );
    offs[c] = ( int32_t ) at; at += m.resi.size();
  }
  int16_t* dResi = dev.staging( total * sizeof( int16_t ) + 256 );
  char* aux = static_cast<char*>( dev.stagingAux( total * sizeof( int32_t ) + 64 + 256 ) );
  int32_t* dOff = reinterpret_cast<int32_t*>( aux );
  int32_t* dCoef = reinterpret_cast<int32_t*>( aux + 64 );
  dev.check( vvhip_upload( dev.ctx(), dResi, host.data(), total * sizeof( int16_t ) ), "TU residuals" );
  dev.check( vvhip_upload( dev.ctx(), dOff, offs.data(), n * sizeof( int32_t ) ), "TU offsets" );
  // blocks of one size share a launch (the two chroma TUs); compact blocks: pitch = width
  for( int c = 0; c < n; )
  {
    int k = 1; while( c + k < n && widths[c + k] == widths[c] ) k++;
    dev.check( vvhip_fwd_transform_batch( dev.ctx(), dResi, widths[c], dOff + c, k, widths[c], heights[c], VVHIP_DCT2, VVHIP_DCT2, bitDepth, dCoef + offs[c] ), "vvhip_fwd_transform_batch" );
    c += k;
  }
  std::vector<int32_t> coef( total );
  dev.check( vvhip_download( dev.ctx(), coef.data(), dCoef, total * sizeof( int32_t ) ), "TU coefficients" );
  for( int c = 0; c < n; c++ ) memcpy( t_tuMemo[c].coef.data(), &coef[offs[c]], sizeof( int32_t ) * t_tuMemo[c].coef.size() );
  t_tuMemoN = n;
  g_tuPrefetches++;
}

bool tuLookup( const int16_t* resi, ptrdiff_t stride, int32_t* coef, unsigned width, unsigned height )
{
  g_tuLookups++;
  for( int c = 0; c < t_tuMemoN; c++ )
  {
    const TuMemo& m = t_tuMemo[c];
    if( m.w != ( int ) width || m.h != ( int ) height ) continue;
    bool same = true;
    for( unsigned y = 0; y < height && same; y++ ) same = memcmp( resi + ( ptrdiff_t ) y * stride, &m.resi[( size_t ) y * width], sizeof( int16_t ) * width ) == 0;
    if( !same ) continue;
    memcpy( coef, m.coef.data(), sizeof( int32_t ) * m.coef.size() );
    g_tuHits++;
    return true;
  }
  return false;
}

// ---- merge SATD pruning (EncCu::addRegularCandsToPruningList, EncoderLib/EncCu.cpp:2266-2300): the predictions of all regular merge candidates of a CU against the CU's
// original in ONE device call (upload original + predictions as compact blocks, one vvhip_dist_batch, download the costs)
std::atomic<uint64_t> g_mergeCalls{ 0 }, g_mergeCands{ 0 }, g_mergeNs{ 0 };
bool mergeCosts( const int16_t* org, int orgStride, const int16_t* const* preds, const int* predStrides, int n, int w, int h, int bitDepth, int hadMode, uint64_t* costs )
{
  if( n < 1 || n > 16 || bitDepth > 10 ) return false;
  SiteTimer timer( g_mergeNs );
  vvhip::Device& dev = vvhip::Device::get();
  const size_t blk = ( size_t ) w * h;
  std::vector<int16_t> host( blk * ( n + 1 ) );
  for( int y = 0; y < h; y++ ) memcpy( &host[( size_t ) y * w], org + ( ptrdiff_t ) y * orgStride, sizeof( int16_t ) * w );
  for( int i = 0; i < n; i++ ) for( int y = 0; y < h; y++ ) memcpy( &host[blk * ( i + 1 ) + ( size_t ) y * w], preds[i] + ( ptrdiff_t ) y * predStrides[i], sizeof( int16_t ) * w );
  std::vector<vvhip_dist_item> items( n );
  for( int i = 0; i < n; i++ ) { items[i].org_off = 0; items[i].cur_off = ( int32_t ) ( blk * ( i + 1 ) ); }
  int16_t* dBlk = dev.staging( host.size() * sizeof( int16_t ) + 256 );
  char* aux = static_cast<char*>( dev.stagingAux( 16 * ( sizeof( vvhip_dist_item ) + sizeof( uint64_t ) ) + 64 ) );
  vvhip_dist_item* dItems = reinterpret_cast<vvhip_dist_item*>( aux );
  uint64_t* dOut = reinterpret_cast<uint64_t*>( aux + 16 * sizeof( vvhip_dist_item ) );
  dev.check( vvhip_upload( dev.ctx(), dBlk, host.data(), host.size() * sizeof( int16_t ) ), "merge blocks" );
  dev.check( vvhip_upload( dev.ctx(), dItems, items.data(), n * sizeof( vvhip_dist_item ) ), "merge items" );
  dev.check( vvhip_dist_batch( dev.ctx(), hadMode == 2 ? VVHIP_DF_HAD_FAST : VVHIP_DF_HAD, dBlk, w, dBlk, w, w, h, 0, bitDepth, dItems, n, dOut ), "vvhip_dist_batch" );
  dev.check( vvhip_download( dev.ctx(), costs, dOut, n * sizeof( uint64_t ) ), "merge costs" );
  g_mergeCalls++; g_mergeCands += n;
  return true;
}

// ---- several GPUs: one picture <-> one device.  $VVHIP_GPUS = number of devices this encoder spreads its pictures over ("all": every visible device;
// default 1 = the default device only).  MCTF-filtered pictures are independent of each other (originals only, MCTF.cpp:666-724): filtered picture k goes to device
// k mod N; the CTU tasks and in-loop stages of a picture run on device poc mod N (bindPicture / the whole-picture ALF hooks).  Pictures a device needs that already
// sit in another device's HBM are copied device-to-device (hipMemcpyPeerAsync over xGMI) instead of crossing PCIe again.
int numGpus()
{
  static const int n = []{
    const char* e = getenv( "VVHIP_GPUS" );
    const int have = vvhip::Device::gpuCount();
    int want = !e ? 1 : ( !strcmp( e, "all" ) ? have : atoi( e ) );
    if( want < 1 ) want = 1;
    return want > have ? ( have > 0 ? have : 1 ) : want; }();
  return n;
}
int gpuOfFilteredPicture( int filterPoc ) { static std::atomic<int> next{ 0 }; static std::mutex m; static std::map<int, int> of;
  std::lock_guard<std::mutex> g( m ); auto it = of.find( filterPoc ); if( it != of.end() ) return it->second;
  const int gpu = ( vvhip::Device::defaultGpu() + next++ ) % numGpus(); if( of.size() > 64 ) of.erase( of.begin() ); of[filterPoc] = gpu; return gpu; }
int gpuOfPicture( int poc ) { return ( vvhip::Device::defaultGpu() + ( poc < 0 ? 0 : poc ) ) % numGpus(); }
struct GpuScope      // binds the calling thread to a device for one hook call; scopes nest (dropResident inside a search call, a worker thread bound to its picture's GPU)
{
  explicit GpuScope( int gpu ) { if( numGpus() > 1 ) { prev = vvhip::Device::selectedGpu(); vvhip::Device::selectGpu( gpu ); on = true; } }
  ~GpuScope() { if( on ) vvhip::Device::selectGpu( prev ); }
  bool on = false; int prev = -1;
};
void bindPicture( int poc ) { if( numGpus() > 1 ) vvhip::Device::selectGpu( gpuOfPicture( poc ) ); }

// ---- resident original pictures: MCTF reads every original picture many times (as the current picture once, as a neighbour of up to 2*range filtered pictures,
// luma for the search and all planes for the filter).  The binding keeps the planes mirrored in HBM — keyed by host buffer, picture order count (buffers are recycled
// for later pictures) AND device — with least-recently-used eviction; the host planes are pinned in place once (recycled buffers), so an upload is one DMA.
struct ResidentPic { const vvenc::Pel* key; int poc; int gpu; int ids[3]; int nComp; uint64_t stamp; };
std::vector<ResidentPic> g_resident;
std::mutex g_mctfLock;                                      // the MCTF stage runs on one thread; the lock makes the hooks safe for parallel-GOP set-ups too
uint64_t g_residentStamp = 0;
std::atomic<uint64_t> g_residentUploads{ 0 }, g_residentHits{ 0 }, g_residentPeerCopies{ 0 };

void dropResident( size_t i )
{
  GpuScope sc( g_resident[i].gpu );
  vvhip::Device& dev = vvhip::Device::get();
  for( int c = 0; c < g_resident[i].nComp; c++ ) dev.unregisterPicture( g_resident[i].ids[c] );
  g_resident.erase( g_resident.begin() + i );
}

// mirror ids of the picture's first nComp planes ON THE CALLING THREAD'S GPU; poc < 0: match the most recent entry of this host buffer (the search of the same
// filter call validated it), uploading (uncached identity) when there is none
ResidentPic& residentPlanes( const vvenc::PelStorage& ps, int poc, int nComp )
{
  vvhip::Device& dev = vvhip::Device::get();
  const int gpu = dev.gpu();
  const vvenc::Pel* key = ps.bufs[0].buf;
  int found = -1, other = -1;
  for( size_t i = 0; i < g_resident.size(); )
  {
    ResidentPic& e = g_resident[i];
    if( e.key == key && poc >= 0 && e.poc != poc ) { dropResident( i ); continue; }            // the host buffer now holds another picture
    if( e.key == key && e.gpu == gpu && ( found < 0 || e.stamp > g_resident[found].stamp ) ) found = ( int ) i;
    if( e.key == key && e.gpu != gpu && ( other < 0 || e.nComp > g_resident[other].nComp ) ) other = ( int ) i;
    i++;
  }
  if( found < 0 )
  {
    const size_t cap = 16 * ( size_t ) numGpus();
    if( g_resident.size() >= cap )
    {
      size_t lru = 0;
      for( size_t i = 1; i < g_resident.size(); i++ ) if( g_resident[i].stamp < g_resident[lru].stamp ) lru = i;
      if( ( int ) lru < other ) other--; else if( ( int ) lru == other ) other = -1;
      dropResident( lru );
    }
    ResidentPic e; e.key = key; e.poc = poc; e.gpu = gpu; e.nComp = 0; e.ids[0] = e.ids[1] = e.ids[2] = -1; e.stamp = 0;
    g_resident.push_back( e );
    found = ( int ) g_resident.size() - 1;
  }
  else g_residentHits++;
  ResidentPic& e = g_resident[found];
  for( int c = e.nComp; c < nComp; c++ )
  {
    const vvenc::CPelBuf b = ps.bufs[c];
    const int margin = vvenc::MCTF_PADDING >> ( c ? 1 : 0 );
    if( other >= 0 && g_resident[other].nComp > c )
    {
      // the plane already sits in another GPU's HBM: device-to-device over xGMI
      const ResidentPic& o = g_resident[other];
      vvhip::Device::selectGpu( o.gpu ); vvhip::Device& src = vvhip::Device::get();
      vvhip::Device::selectGpu( gpu );
      e.ids[c] = src.copyMirrorTo( o.ids[c], dev );
      g_residentPeerCopies++;
    }
    else
    {
      vvhip::Device::pinHost( b.buf - ( ptrdiff_t ) margin * b.stride - margin, ( size_t ) b.stride * ( b.height + 2 * margin ) * sizeof( vvenc::Pel ) );
      e.ids[c] = dev.registerPicture( b.buf, ( int ) b.stride, b.width, b.height, margin, false );      // id use only: never aliased by the per-call entries
      g_residentUploads++;
    }
  }
  if( nComp > e.nComp ) e.nComp = nComp;
  e.stamp = ++g_residentStamp;
  return e;
}

// ---- whole hierarchical ME of MCTF::motionEstimationMCTF on the GPU; pictures carry MCTF_PADDING extended margins (MCTF.cpp:608-612).
// mctfPrefetch: the references of the first loop of MCTF::filter are known before the loop starts -> ONE device call scores them all (their pyramid levels and
// search stages share launches, vvhip_mctf_motion_estimation); mctfMe then hands each field over, or computes a single reference (the adaptive extra ones).
bool mctfWants( const vvenc::MCTF* m, int width, int height ) { return width >= 64 && height >= 64 && ( m->m_mctfUnitSize == 8 || m->m_mctfUnitSize == 16 ); }

struct MeCache { int curPoc = -1; std::map<int, std::vector<vvhip_mv>> fields; } g_meCache;

void motionFieldsOnDevice( vvenc::MCTF* m, const vvenc::PelStorage& orig, int curPoc, const std::vector<const vvenc::PelStorage*>& refs, const std::vector<int>& refPocs, bool addLevel )
{
  const int nRefs = ( int ) refs.size();
  if( !nRefs ) return;
  const int unit = m->m_mctfUnitSize;
  const vvenc::CPelBuf o = orig.Y();
  const size_t count = ( size_t ) ( ( o.width + unit - 1 ) / unit ) * ( ( o.height + unit - 1 ) / unit );
  const int idO = residentPlanes( orig, curPoc, 1 ).ids[0];
  std::vector<int> idR( nRefs );
  for( int r = 0; r < nRefs; r++ )
  {
    if( refs[r]->Y().stride != o.stride ) throw vvhip::Exception( "HIP MCTF: original pictures with different strides" );
    idR[r] = residentPlanes( *refs[r], refPocs[r], 1 ).ids[0];
  }
  std::vector<vvhip_mv*> outs( nRefs );
  for( int r = 0; r < nRefs; r++ ) { std::vector<vvhip_mv>& f = g_meCache.fields[refPocs[r]]; f.resize( count ); outs[r] = f.data(); }
  g_m->motionEstimation( idO, idR.data(), nRefs, m->m_encCfg->m_internalBitDepth[0], unit, m->m_encCfg->m_vvencMCTF.MCTFSpeed, addLevel, outs.data() );
  g_mctfDeviceCalls++;
}

void mctfPrefetch( vvenc::MCTF* m, const void* picFifoDeque, int dropFront, int dropBack, const vvenc::PelStorage& orig, bool addLevel, int filterPoc )
{
  std::lock_guard<std::mutex> g( g_mctfLock );
  GpuScope sc( gpuOfFilteredPicture( filterPoc ) );
  const std::deque<vvenc::Picture*>& fifo = *static_cast<const std::deque<vvenc::Picture*>*>( picFifoDeque );
  g_meCache.curPoc = filterPoc; g_meCache.fields.clear();
  std::vector<const vvenc::PelStorage*> refs; std::vector<int> pocs; std::vector<vvenc::PelStorage> keep;
  keep.reserve( fifo.size() );
  for( int i = dropFront; i < ( int ) fifo.size() - dropBack; i++ )
  {
    if( fifo[i]->poc == filterPoc ) continue;
    keep.emplace_back(); keep.back().createFromBuf( fifo[i]->getOrigBuf() );
    pocs.push_back( fifo[i]->poc );
  }
  for( auto& k : keep ) refs.push_back( &k );
  motionFieldsOnDevice( m, orig, filterPoc, refs, pocs, addLevel );
}

bool mctfMe( vvenc::MCTF* m, const vvenc::PelStorage& refPic, const vvenc::PelStorage& orig, vvenc::Array2D<vvenc::MotionVector>& mvs, bool addLevel, int refPoc, int curPoc )
{
  if( !mctfWants( m, orig.Y().width, orig.Y().height ) ) return false;       // below the device entry point's minimum: the table-entry path
  std::lock_guard<std::mutex> g( g_mctfLock );
  GpuScope sc( gpuOfFilteredPicture( curPoc ) );
  if( g_meCache.curPoc != curPoc ) { g_meCache.curPoc = curPoc; g_meCache.fields.clear(); }
  auto it = g_meCache.fields.find( refPoc );
  if( it == g_meCache.fields.end() )
  {
    motionFieldsOnDevice( m, orig, curPoc, { &refPic }, { refPoc }, addLevel );
    it = g_meCache.fields.find( refPoc );
  }
  const std::vector<vvhip_mv>& out = it->second;
  if( out.size() != ( size_t ) mvs.w() * mvs.h() ) throw vvhip::Exception( "HIP MCTF: motion field size" );
  for( int y = 0; y < mvs.h(); y++ ) for( int x = 0; x < mvs.w(); x++ )
  {
    vvenc::MotionVector& d = mvs.get( x, y ); const vvhip_mv& s = out[( size_t ) y * mvs.w() + x];
    d.x = s.x; d.y = s.y; d.error = s.error; d.rmsme = ( uint16_t ) s.rmsme; d.overlap = s.overlap;
  }
  g_calls[7] += 1000000;      // marks "whole-picture ME ran on the device"
  return true;
}

// whole MCTF::bilateralFilter (MCTF.cpp:1489-1552) on the GPU: every plane of the original and of the references is mirrored, the motion fields
// the (CPU or GPU) search left in srcFrameInfo[i].mvs are uploaded, one launch per component plane
bool mctfApply( const vvenc::MCTF* m, const vvenc::PelStorage& orgPic, void* infoDeque, vvenc::PelStorage& newOrgPic, double overallStrength )
{
  std::deque<vvenc::TemporalFilterSourcePicInfo>& info = *static_cast<std::deque<vvenc::TemporalFilterSourcePicInfo>*>( infoDeque );
  const int nRefs = ( int ) info.size(), unit = m->m_mctfUnitSize;
  const int numComp = ( int ) vvenc::getNumberValidComponents( m->m_encCfg->m_internChromaFormat );
  if( nRefs < 1 || nRefs > 12 || ( unit != 8 && unit != 16 ) || ( numComp == 3 && m->m_encCfg->m_internChromaFormat != CHROMA_420 ) ) return false;
  std::lock_guard<std::mutex> g( g_mctfLock );
  GpuScope sc( gpuOfFilteredPicture( m->m_filterPoc ) );
  std::vector<int> orgIds( 3, -1 ), refIds( 3 * nRefs, -1 );
  const bool searchOnDevice = g_vvhipHooks.mctfMe != nullptr && mctfWants( m, orgPic.Y().width, orgPic.Y().height );   // then the pictures of this filter call are already resident (and validated by POC)
  if( !searchOnDevice ) while( !g_resident.empty() ) dropResident( 0 );
  { const ResidentPic& e = residentPlanes( orgPic, -1, numComp ); for( int c = 0; c < numComp; c++ ) orgIds[c] = e.ids[c]; }
  for( int r = 0; r < nRefs; r++ ) { const ResidentPic& e = residentPlanes( info[r].picBuffer, -1, numComp ); for( int c = 0; c < numComp; c++ ) refIds[3 * r + c] = e.ids[c]; }
  const int mvW = info[0].mvs.w(), mvH = info[0].mvs.h();
  std::vector<std::vector<vvhip_mv>> mvs( nRefs, std::vector<vvhip_mv>( ( size_t ) mvW * mvH ) );
  std::vector<const vvhip_mv*> mvPtr( nRefs );
  std::vector<double> strengths( nRefs );
  for( int r = 0; r < nRefs; r++ )
  {
    for( int y = 0; y < mvH; y++ ) for( int x = 0; x < mvW; x++ )
    {
      const vvenc::MotionVector& s = info[r].mvs.get( x, y ); vvhip_mv& d = mvs[r][( size_t ) y * mvW + x];
      d.x = s.x; d.y = s.y; d.error = s.error; d.rmsme = s.rmsme; d.overlap = s.overlap;
    }
    mvPtr[r] = mvs[r].data();
    strengths[r] = vvenc::MCTF::m_refStrengths[m->m_encCfg->m_picReordering ? 0 : 1][info[r].index];      // MCTF.cpp:1405,1480
  }
  vvenc::Pel* outs[3]; int strides[3];
  for( int c = 0; c < numComp; c++ ) { outs[c] = newOrgPic.bufs[c].buf; strides[c] = ( int ) newOrgPic.bufs[c].stride; }
  g_m->bilateralFilter( orgIds.data(), refIds.data(), nRefs, mvPtr.data(), strengths.data(), m->m_encCfg->m_QP, m->m_encCfg->m_internalBitDepth[0], unit, m->m_lowResFltApply,
                        overallStrength, numComp, outs, strides );
  if( !searchOnDevice ) while( !g_resident.empty() ) dropResident( 0 );
  g_calls[9]++;
  return true;
}

// ---- resident reconstruction pictures: every finished (border-extended) CTU row of a picture's luma reconstruction is uploaded to that picture's mirror on every GPU
// in use (the "broadcast of reference pictures" of SURVEY 8e, host -> each device: the producer is the host).  Mirrors are keyed by the host buffer (buffers are recycled
// for later pictures: rows are simply overwritten as the next picture finishes them) and marked reference-only.
struct ReconMirror { int ids[16]; int stride, width, height, margin; };
std::mutex g_reconLock;
std::map<const int16_t*, ReconMirror> g_recon;
std::atomic<uint64_t> g_reconRowUploads{ 0 };

void reconRows( const int16_t* origin, int stride, int width, int height, int margin, int y0, int rows )
{
  const int nGpus = std::min( numGpus(), 16 );
  ReconMirror rm;
  {
    std::lock_guard<std::mutex> g( g_reconLock );
    auto it = g_recon.find( origin );
    if( it != g_recon.end() && ( it->second.stride != stride || it->second.width != width || it->second.height != height || it->second.margin != margin ) )
    {
      for( int k = 0; k < nGpus; k++ ) { GpuScope sc( ( vvhip::Device::defaultGpu() + k ) % numGpus() ); vvhip::Device::get().unregisterPicture( it->second.ids[k] ); }
      g_recon.erase( it ); it = g_recon.end();
    }
    if( it == g_recon.end() )
    {
      ReconMirror n; n.stride = stride; n.width = width; n.height = height; n.margin = margin;
      vvhip::Device::pinHost( origin - ( ptrdiff_t ) margin * stride - margin, ( size_t ) stride * ( height + 2 * margin ) * sizeof( int16_t ) );
      for( int k = 0; k < nGpus; k++ )
      {
        GpuScope sc( ( vvhip::Device::defaultGpu() + k ) % numGpus() );
        vvhip::Device& dev = vvhip::Device::get();
        n.ids[k] = dev.registerPicture( origin, stride, width, height, margin, false, false );
        dev.setReference( n.ids[k] );
      }
      it = g_recon.emplace( origin, n ).first;
    }
    rm = it->second;
  }
  for( int k = 0; k < nGpus; k++ )
  {
    GpuScope sc( ( vvhip::Device::defaultGpu() + k ) % numGpus() );
    vvhip::Device::get().updatePictureRows( rm.ids[k], y0, rows );
  }
  g_reconRowUploads++;
}

// one xPatternRefinement stage (EncoderLib/InterSearch.cpp:760-880) scored by ONE batched device call: positions = (refine[i] + base) * iFrac in
// quarter samples around the block at the best integer vector; the encoder's loop replays the costs (skip / break rules, MV bits, strict <)
std::atomic<uint64_t> g_patternCalls( 0 );
bool patternCosts( const vvenc::CPelBuf* key, const vvenc::CPelBuf* pattern, int baseHor, int baseVer, int iFrac, const vvenc::Mv* refine, int bitDepth, int hadMode, int reduceTap,
                   bool useAltHpelIf, uint64_t* cost9 )
{
  vvhip::CPelBuf org; org.buf = key->buf; org.stride = ( int ) key->stride; org.width = key->width; org.height = key->height;
  int q[9][2];
  for( int i = 0; i < 9; i++ ) { q[i][0] = ( refine[i].hor + baseHor ) * iFrac; q[i][1] = ( refine[i].ver + baseVer ) * iFrac; }
  vvhip::Distortion c[9];
  if( !g_rd->patternRefineCosts( org, pattern->buf, ( int ) pattern->stride, q, 9, bitDepth, hadMode, reduceTap, useAltHpelIf, c ) ) return false;
  for( int i = 0; i < 9; i++ ) cost9[i] = c[i];
  g_patternCalls++;
  return true;
}

// ---- integer TZ search (InterSearch::xTZ8PointDiamondSearch, EncoderLib/InterSearch.cpp:557-760): the positions a diamond round may test are
// scored by ONE device call when the round starts; xTZSearchHelp (:410-438) then takes its SAD from this table and runs its own update
// logic (MV-bit cost, strict <, early-exit bookkeeping).  Values are exact, so a position the round skips is simply never looked up.
struct TzTable { const int16_t* org = nullptr; const int16_t* ref = nullptr; int subShift = -1, n = 0; int xy[24][2]; uint64_t sad[24]; };
thread_local TzTable t_tz;
std::atomic<uint64_t> g_tzRounds( 0 ), g_tzHits( 0 );

void tzReset() { t_tz.n = 0; t_tz.org = nullptr; }

void tzPrefetch( const vvenc::DistParam* dp, const int16_t* refY, int refStride, int sx, int sy, int d, bool corners, int left, int right, int top, int bottom )
{
  TzTable& t = t_tz;
  t.n = 0; t.org = nullptr;
  if( dp->applyWeight || dp->org.width < 4 || dp->org.width > 128 || dp->org.height > 128 ) return;
  auto add = [&]( int x, int y ) { if( x >= left && x <= right && y >= top && y <= bottom && t.n < 24 ) { t.xy[t.n][0] = x; t.xy[t.n][1] = y; t.n++; } };
  if( d == 1 )
  {
    add( sx, sy - 1 ); add( sx - 1, sy ); add( sx + 1, sy ); add( sx, sy + 1 );
    if( corners ) { add( sx - 1, sy - 1 ); add( sx + 1, sy - 1 ); add( sx - 1, sy + 1 ); add( sx + 1, sy + 1 ); }
  }
  else if( d <= 8 )
  {
    const int h = d >> 1;
    add( sx, sy - d ); add( sx - h, sy - h ); add( sx + h, sy - h ); add( sx - d, sy ); add( sx + d, sy ); add( sx - h, sy + h ); add( sx + h, sy + h ); add( sx, sy + d );
  }
  else
  {
    add( sx, sy - d ); add( sx - d, sy ); add( sx + d, sy ); add( sx, sy + d );
    for( int i = 1; i < 4; i++ )
    {
      const int q = ( d >> 2 ) * i;
      add( sx - q, sy - d + q ); add( sx + q, sy - d + q ); add( sx - q, sy + d - q ); add( sx + q, sy + d - q );
    }
  }
  if( t.n == 0 ) return;
  vvhip::CPelBuf org; org.buf = dp->org.buf; org.stride = ( int ) dp->org.stride; org.width = dp->org.width; org.height = dp->org.height;
  vvhip::Distortion c[24];
  g_rd->distAtPositions( VVHIP_DF_SAD, org, refY, refStride, dp->subShift, dp->bitDepth, t.xy, t.n, c );
  for( int i = 0; i < t.n; i++ ) t.sad[i] = c[i];
  t.org = dp->org.buf; t.ref = refY; t.subShift = dp->subShift;
  g_tzRounds++;
}

bool tzLookup( const vvenc::DistParam* dp, const int16_t* refY, int x, int y, uint64_t* sad )
{
  const TzTable& t = t_tz;
  if( t.org != dp->org.buf || t.ref != refY || t.subShift != dp->subShift ) return false;
  for( int i = 0; i < t.n; i++ ) if( t.xy[i][0] == x && t.xy[i][1] == y ) { *sad = t.sad[i]; g_tzHits++; return true; }
  return false;
}

// the refinement search of DMVR::xProcessDMVR for all sub-blocks of one CU in one device call (InterPrediction.cpp:1312-1392)
std::atomic<uint64_t> g_dmvrCalls( 0 );
bool dmvrSearch( const int16_t* ref0, int stride0, int fx0, int fy0, const int16_t* ref1, int stride1, int fx1, int fy1, int cuWidth, int cuHeight, int dx, int dy, int bitDepth,
                 int16_t* mvd, uint64_t* minCost )
{
  static vvhip::DMVROps ops;
  if( !ops.refineCu( ref0, stride0, fx0, fy0, ref1, stride1, fx1, fy1, cuWidth, cuHeight, dx, dy, bitDepth, mvd, minCost ) ) return false;
  g_dmvrCalls++;
  return true;
}

// ---- InterpolationFilter tables: every slot of m_filterHor / m_filterVer / m_filterCopy / m_filter4x4 / m_filter8xH /
// m_filter16xH forwards to the shim's slot of the same index (the only difference between the two signatures is the ClpRng type)
vvhip::InterpolationFilter* g_if = nullptr;
template<int I, int F, int L> void ifHorT( const vvenc::ClpRng& c, vvenc::Pel const* s, int ss, vvenc::Pel* d, int ds, int w, int h, vvenc::TFilterCoeff const* co )
{ g_calls[8]++; const vvhip::ClpRng cc = { c.bd }; g_if->m_filterHor[I][F][L]( cc, s, ss, d, ds, w, h, co ); }
template<int I, int F, int L> void ifVerT( const vvenc::ClpRng& c, vvenc::Pel const* s, int ss, vvenc::Pel* d, int ds, int w, int h, vvenc::TFilterCoeff const* co )
{ g_calls[8]++; const vvhip::ClpRng cc = { c.bd }; g_if->m_filterVer[I][F][L]( cc, s, ss, d, ds, w, h, co ); }
template<int F, int L> void ifCopyT( const vvenc::ClpRng& c, vvenc::Pel const* s, int ss, vvenc::Pel* d, int ds, int w, int h, bool bi )
{ g_calls[8]++; const vvhip::ClpRng cc = { c.bd }; g_if->m_filterCopy[F][L]( cc, s, ss, d, ds, w, h, bi ); }
template<int K, int I, int L> void ifFusedT( const vvenc::ClpRng& c, vvenc::Pel const* s, int ss, vvenc::Pel* d, int ds, int w, int h, vvenc::TFilterCoeff const* ch, vvenc::TFilterCoeff const* cv )
{
  g_calls[8]++;
  const vvhip::ClpRng cc = { c.bd };
  ( K == 0 ? g_if->m_filter4x4[I][L] : K == 1 ? g_if->m_filter8xH[I][L] : g_if->m_filter16xH[I][L] )( cc, s, ss, d, ds, w, h, ch, cv );
}
template<int I> void fillIf1D( vvenc::InterpolationFilter* f )
{
  f->m_filterHor[I][0][0] = ifHorT<I, 0, 0>; f->m_filterHor[I][0][1] = ifHorT<I, 0, 1>; f->m_filterHor[I][1][0] = ifHorT<I, 1, 0>; f->m_filterHor[I][1][1] = ifHorT<I, 1, 1>;
  f->m_filterVer[I][0][0] = ifVerT<I, 0, 0>; f->m_filterVer[I][0][1] = ifVerT<I, 0, 1>; f->m_filterVer[I][1][0] = ifVerT<I, 1, 0>; f->m_filterVer[I][1][1] = ifVerT<I, 1, 1>;
}
void initIF( vvenc::InterpolationFilter* f )
{
  // the luma / chroma tap tables (indices 0 = 8, 1 = 4, 3 = 6 taps) and the copies; the bilinear (DMVR) entries stay on the CPU
  fillIf1D<0>( f ); fillIf1D<1>( f ); fillIf1D<3>( f );
  f->m_filterCopy[0][1] = ifCopyT<0, 1>; f->m_filterCopy[1][0] = ifCopyT<1, 0>; f->m_filterCopy[1][1] = ifCopyT<1, 1>; f->m_filterCopy[0][0] = ifCopyT<0, 0>;
  f->m_filter4x4[0][0] = ifFusedT<0, 0, 0>; f->m_filter4x4[0][1] = ifFusedT<0, 0, 1>; f->m_filter4x4[1][0] = ifFusedT<0, 1, 0>; f->m_filter4x4[1][1] = ifFusedT<0, 1, 1>;
  f->m_filter8xH[0][0] = ifFusedT<1, 0, 0>; f->m_filter8xH[0][1] = ifFusedT<1, 0, 1>; f->m_filter8xH[1][0] = ifFusedT<1, 1, 0>; f->m_filter8xH[1][1] = ifFusedT<1, 1, 1>;
  f->m_filter16xH[0][0] = ifFusedT<2, 0, 0>; f->m_filter16xH[0][1] = ifFusedT<2, 0, 1>; f->m_filter16xH[1][0] = ifFusedT<2, 1, 0>; f->m_filter16xH[1][1] = ifFusedT<2, 1, 1>;
}

} // namespace

static int g_slotMask = 0;

// bit5: the reference's global g_tCoeffOps takes the shim's ten slots AS THEY ARE — identical signatures, no trampoline, no source
// patch (the table is a public global, TrQuant_EMT.h:91).  Called again after every SIMD (re-)initialisation.
template<int K> void fwdSlot( const vvenc::TMatrixCoeff* tc, const vvenc::TCoeff* src, vvenc::TCoeff* dst, unsigned line, unsigned red, unsigned cut, int shift )
{ g_calls[2]++; vvhip::g_tCoeffOps.fastFwdCore_2D[K]( tc, src, dst, line, red, cut, shift ); }
template<int K> void invSlot( const vvenc::TMatrixCoeff* it, const vvenc::TCoeff* src, vvenc::TCoeff* dst, unsigned lines, unsigned red, unsigned rows )
{ g_calls[3]++; vvhip::g_tCoeffOps.fastInvCore[K]( it, src, dst, lines, red, rows ); }

extern "C" __attribute__( ( visibility( "default" ) ) ) void vvref_after_simd_init()
{
  if( !( g_slotMask & 32 ) ) return;
  vvenc::TCoeffOps& t = vvenc::g_tCoeffOps;
  const vvhip::TCoeffOps& h = vvhip::g_tCoeffOps;
  t.cpyResi4 = h.cpyResi4; t.cpyResi8 = h.cpyResi8; t.cpyCoeff4 = h.cpyCoeff4; t.cpyCoeff8 = h.cpyCoeff8;
  t.roundClip4 = h.roundClip4; t.roundClip8 = h.roundClip8;
  // (the two matrix cores go through a counting wrapper so the test can see they were used; h's entries have the same signature)
  t.fastFwdCore_2D[0] = fwdSlot<0>; t.fastFwdCore_2D[1] = fwdSlot<1>; t.fastFwdCore_2D[2] = fwdSlot<2>; t.fastFwdCore_2D[3] = fwdSlot<3>; t.fastFwdCore_2D[4] = fwdSlot<4>;
  t.fastFwdCore_1D[0] = fwdSlot<0>; t.fastFwdCore_1D[1] = fwdSlot<1>; t.fastFwdCore_1D[2] = fwdSlot<2>; t.fastFwdCore_1D[3] = fwdSlot<3>; t.fastFwdCore_1D[4] = fwdSlot<4>;
  t.fastInvCore[0] = invSlot<0>; t.fastInvCore[1] = invSlot<1>; t.fastInvCore[2] = invSlot<2>; t.fastInvCore[3] = invSlot<3>; t.fastInvCore[4] = invSlot<4>;
}

std::atomic<uint64_t> g_alfCtus{ 0 };
bool alfCtu( const int16_t* const rec[3], const int recStride[3], const int16_t* const org[3], const int orgStride[3], int width, int height, int chromaShift,
             int bitDepth, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3], uint8_t* cls, float* const stats[3] )
{
  static thread_local vvhip::ALFOps alf;      // (the object carries per-thread download areas)
  if( ( width & 3 ) || ( height & 3 ) || ( ( width >> chromaShift ) & 3 ) || ( ( height >> chromaShift ) & 3 ) ) return false;
  if( !alf.deriveClassification( rec[0], recStride[0], width, height, bitDepth, vbLumaH, vbLumaPos, cls ) ) return false;
  if( enabled[0] && !alf.getStatistics( org[0], orgStride[0], rec[0], recStride[0], width, height, 128, 7, cls, vbLumaH, vbLumaPos, stats[0], stats[0] ) ) return false;
  for( int c = 1; c < 3; c++ )
    if( enabled[c] && !alf.getStatistics( org[c], orgStride[c], rec[c], recStride[c], width >> chromaShift, height >> chromaShift, 128, 5, nullptr, vbChromaH, vbChromaPos, stats[c], stats[c] ) ) return false;
  g_alfCtus++;
  return true;
}

// ---- whole-picture ALF stages: one vvhip::ALFOps per EncAdaptiveLoopFilter object (it keeps the unfiltered planes and the classes of its current picture in HBM
// between the statistics call and the filtering call), on the picture's GPU
std::mutex g_alfPicLock;
struct AlfPictureState { std::unique_ptr<vvhip::ALFOps> ops; int donePoc = -1; bool done = false; std::mutex busy;
                         bool bandsOpen = false; int bandPoc = 0; std::vector<char> rowSeen; };      // the picture whose statistics are being issued in bands (alfRow)
std::map<const void*, std::unique_ptr<AlfPictureState>> g_alfState;
AlfPictureState& alfState( const void* owner )
{
  std::lock_guard<std::mutex> g( g_alfPicLock );
  std::unique_ptr<AlfPictureState>& st = g_alfState[owner];
  if( !st ) { st.reset( new AlfPictureState ); st->ops.reset( new vvhip::ALFOps ); }
  return *st;
}

// EncAdaptiveLoopFilter::deriveFilter starts on a picture: whatever this ALF object filtered before is history (same POC in a later pass / encode included)
void alfBeginPicture( const void* owner, int /*poc*/ )
{
  AlfPictureState& st = alfState( owner );
  std::lock_guard<std::mutex> g( st.busy );
  st.done = false; st.donePoc = -1;
}

std::atomic<uint64_t> g_alfPictures{ 0 };
// The whole-picture statistics call runs inside the serial part of the ALF stage (deriveFilter): one upload + three launches + a download per picture against CPU work that is
// spread over the worker threads.  It pays when the pool is saturated — measured: 1080p (510 CTUs) with 2 / 4 threads +16 / +14 %, 4K (2040 CTUs) with 8 threads +8...13 %,
// 1080p with 8 threads +-0 (profiles/r02_e2e_encoder_fps.md, r03) — so the hook is taken from $VVHIP_ALF_MIN_CTUS_PER_THREAD CTUs per encoder thread on (default 100).
// Round 6: the picture goes to the device in bands from the row tasks (alfRow below) and the serial part is 0.16 ms instead of 7.5 ms per 1080p picture
// (profiles/r06_alf_bands.log).  The rule stays: with the serial part gone the hook is neutral at 1080p / 8 threads on a quiet host (48.8 against 48.5 fps, eight runs each) and
// loses on a loaded one (47.3 against 48.3, five runs each, tools/exp/e2e_queues_ab.py: the row tasks' uploads compete with the encoder's own memory traffic).
bool alfPictureOn( int numCtusInPic, int numThreads )
{
  static const int minPerThread = []{ const char* e = getenv( "VVHIP_ALF_MIN_CTUS_PER_THREAD" ); return e ? atoi( e ) : 100; }();
  return numCtusInPic >= minPerThread * ( numThreads > 0 ? numThreads : 1 );
}
// Statistics in bands: the statistics task of CTU row y runs when row y + 1 has left SAO (EncSlice.cpp:1135-1141) — every sample the row's statistics read is
// final.  When the CTU rows of a statistics-unit row are all reported, the unit row goes up + is computed + comes down asynchronously on the reporting worker's stream; the
// serial part (deriveFilter -> alfPicture) then waits for marks that are long complete instead of moving the whole picture.  $VVHIP_ALF_BANDS=0: the picture as a whole.
std::atomic<uint64_t> g_alfBands{ 0 }, g_alfBandPictures{ 0 }, g_alfRowNs{ 0 }, g_alfPictureNs{ 0 };      // (ns: wall time inside alfRow on the workers / inside alfPicture = the serial part)
// $VVHIP_ALF_TIMELINE=<file>: one line per alfRow / alfPicture call (measurement aid: "row|picture poc row-or--1 wait_lock_us inside_us t_us")
void alfTimeline( const char* what, int poc, int row, uint64_t waitNs, uint64_t insideNs )
{
  static FILE* f = []{ const char* e = getenv( "VVHIP_ALF_TIMELINE" ); return e && *e ? fopen( e, "a" ) : ( FILE* ) nullptr; }();
  if( !f ) return;
  static std::mutex m; static const auto t0 = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> g( m );
  fprintf( f, "%s %d %d %.1f %.1f %.1f\n", what, poc, row, waitNs / 1e3, insideNs / 1e3, std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now() - t0 ).count() / 1e3 );
  fflush( f );
}
struct NsScope { std::atomic<uint64_t>& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                 explicit NsScope( std::atomic<uint64_t>& a ) : acc( a ) {}
                 ~NsScope() { acc += ( uint64_t ) std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now() - t0 ).count(); } };
bool alfRow( const void* owner, int poc, int ctuRow, const int16_t* const rec[3], const int recStride[3], const int16_t* const org[3], const int orgStride[3], int width, int height,
             int bitDepth, int ctuSize, int unitSize, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3] )
{
  static const bool on = []{ const char* e = getenv( "VVHIP_ALF_BANDS" ); return !e || atoi( e ) != 0; }();
  if( !on ) return false;
  NsScope ns( g_alfRowNs );
  AlfPictureState& st = alfState( owner );
  const auto tl0 = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> g( st.busy );
  const auto tl1 = std::chrono::steady_clock::now();
  struct Tl { int poc, row; std::chrono::steady_clock::time_point a, b; ~Tl() { alfTimeline( "row", poc, row, ( uint64_t ) std::chrono::duration_cast<std::chrono::nanoseconds>( b - a ).count(),
                ( uint64_t ) std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now() - b ).count() ); } } tl{ poc, ctuRow, tl0, tl1 };
  GpuScope sc( gpuOfPicture( poc ) );
  const int ctuRows = ( height + ctuSize - 1 ) / ctuSize;
  if( ctuRow < 0 || ctuRow >= ctuRows ) return false;
  if( !st.bandsOpen || st.bandPoc != poc || ( int ) st.rowSeen.size() != ctuRows || st.rowSeen[ctuRow] )      // a new picture (or the same POC again: another pass)
  {
    st.bandsOpen = st.ops->statisticsBegin( recStride, orgStride, width, height, bitDepth, ctuSize, unitSize, vbLumaH, vbLumaPos, vbChromaH, vbChromaPos, enabled );
    st.bandPoc = poc; st.rowSeen.assign( ( size_t ) ctuRows, 0 );
  }
  if( !st.bandsOpen ) return false;
  st.rowSeen[ctuRow] = 1;
  const int perUnit = unitSize / ctuSize, u = ctuRow / perUnit;
  for( int r = u * perUnit; r < std::min( ctuRows, ( u + 1 ) * perUnit ); r++ ) if( !st.rowSeen[r] ) return true;      // the unit row waits for its other CTU rows
  if( !st.ops->statisticsBand( u, rec, org ) ) { st.bandsOpen = false; return false; }
  g_alfBands++;
  return true;
}

bool alfPicture( const void* owner, int poc, const int16_t* const rec[3], const int recStride[3], const int16_t* const org[3], const int orgStride[3], int width, int height, int bitDepth,
                 int ctuSize, int unitSize, int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, const bool enabled[3], uint8_t* cls, float* const stats[3] )
{
  NsScope ns( g_alfPictureNs );
  AlfPictureState& st = alfState( owner );
  const auto tl0 = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> g( st.busy );
  const auto tl1 = std::chrono::steady_clock::now();
  struct Tl { int poc; std::chrono::steady_clock::time_point a, b; ~Tl() { alfTimeline( "picture", poc, -1, ( uint64_t ) std::chrono::duration_cast<std::chrono::nanoseconds>( b - a ).count(),
                ( uint64_t ) std::chrono::duration_cast<std::chrono::nanoseconds>( std::chrono::steady_clock::now() - b ).count() ); } } tl{ poc, tl0, tl1 };
  GpuScope sc( gpuOfPicture( poc ) );
  if( st.bandsOpen && st.bandPoc == poc )
  {
    st.bandsOpen = false;
    const uint8_t* hCls = nullptr; const float* hSt[3] = { nullptr, nullptr, nullptr };
    if( st.ops->statisticsEnd( rec, &hCls, hSt ) )
    {
      const int units = ( ( width + unitSize - 1 ) / unitSize ) * ( ( height + unitSize - 1 ) / unitSize );
      memcpy( cls, hCls, ( size_t ) ( width / 4 ) * ( height / 4 ) * 2 );
      for( int c = 0; c < 3; c++ ) if( enabled[c] ) memcpy( stats[c], hSt[c], ( size_t ) units * ( c ? 1 : 25 ) * 183 * sizeof( float ) );
      g_alfPictures++; g_alfBandPictures++;
      return true;
    }
  }
  if( !st.ops->pictureStatistics( rec, recStride, org, orgStride, width, height, bitDepth, ctuSize, unitSize, vbLumaH, vbLumaPos, vbChromaH, vbChromaPos, enabled, cls, stats ) ) return false;
  g_alfPictures++;
  return true;
}

std::atomic<uint64_t> g_ccAlfCtus{ 0 };
bool ccAlfCtu( const int16_t* orgC, int orgStride, const int16_t* slfC, int slfStride, const int16_t* recLuma, int recStride, int widthC, int heightC,
               int vbCTUHeight, int vbPos, int picHeightFromHere, float* record )
{
  static thread_local vvhip::ALFOps alf;      // (the object carries per-thread download areas)
  if( !alf.getStatisticsCcAlf( orgC, orgStride, slfC, slfStride, recLuma, recStride, widthC, heightC, 64, vbCTUHeight, vbPos, picHeightFromHere, record ) ) return false;
  g_ccAlfCtus++;
  return true;
}

std::atomic<uint64_t> g_alfFilterBlks{ 0 }, g_ccAlfFilterBlks{ 0 };
bool alfFilterBlk( const void* classifier, int16_t* dst, int dstStride, const int16_t* src, int srcStride, int width, int height, int filterLength,
                   const short* coeff, const short* clip, int bitDepth, int vbCTUHeight, int vbPos )
{
  static thread_local vvhip::ALFOps alf;      // (the object carries per-thread download areas)
  if( ( width & 3 ) || ( height & 3 ) || width > 128 || height > 128 ) return false;
  uint8_t cls[32 * 32 * 2];
  if( classifier )
  {
    const uint8_t* c = static_cast<const uint8_t*>( classifier );                  // AlfClassifier = { uint8_t classIdx, transposeIdx }, 32 per row
    for( int i = 0; i < height / 4; i++ ) memcpy( cls + ( size_t ) i * ( width / 4 ) * 2, c + ( size_t ) i * 32 * 2, ( size_t ) ( width / 4 ) * 2 );
  }
  const short set0 = 0;
  if( !alf.filterPlane( src, srcStride, dst, dstStride, width, height, 128, bitDepth, filterLength, classifier ? cls : nullptr, coeff, clip, 1, &set0, vbCTUHeight, vbPos ) ) return false;
  g_alfFilterBlks++;
  return true;
}

bool ccAlfFilterBlk( int16_t* dstC, int dstStride, const int16_t* recLuma, int recStride, int widthC, int heightC, const int16_t* coeff, int bitDepth, int vbCTUHeight, int vbPos )
{
  static thread_local vvhip::ALFOps alf;      // (the object carries per-thread download areas)
  const uint8_t on = 1;
  if( !alf.filterCcAlf( dstC, dstStride, recLuma, recStride, widthC, heightC, 64, bitDepth, coeff, 1, &on, vbCTUHeight, vbPos ) ) return false;
  g_ccAlfFilterBlks++;
  return true;
}

std::atomic<uint64_t> g_alfFilterPictures{ 0 };
bool alfFilterPicture( const void* owner, int poc, const int16_t* const src[3], const int srcStride[3], int16_t* const dst[3], const int dstStride[3], int width, int height, int bitDepth,
                       int ctuSize, const uint8_t* cls, const short* lumaCoeff, const short* lumaClip, int numLumaSets, const short* lumaCtuSet, const short* chromaCoeff,
                       const short* chromaClip, int numChromaSets, const short* const chromaCtuSet[2], int vbLumaH, int vbLumaPos, int vbChromaH, int vbChromaPos, bool alreadyDoneOnly )
{
  AlfPictureState& st = alfState( owner );
  std::lock_guard<std::mutex> g( st.busy );                                       // later CTU tasks of the picture wait here until the first one has finished it
  if( st.done && st.donePoc == poc ) return true;                                 // (reset by alfBeginPicture when deriveFilter starts on the next picture)
  if( alreadyDoneOnly ) return false;
  GpuScope sc( gpuOfPicture( poc ) );
  if( !st.ops->filterPicture( src, srcStride, dst, dstStride, width, height, bitDepth, ctuSize, cls, lumaCoeff, lumaClip, numLumaSets, lumaCtuSet, chromaCoeff, chromaClip, numChromaSets,
                              chromaCtuSet, vbLumaH, vbLumaPos, vbChromaH, vbChromaPos ) ) return false;
  st.done = true; st.donePoc = poc;
  g_alfFilterPictures++;
  return true;
}

// ---- installation.  Hook mask (also the argument of --SIMD=HIP:<mask>):
//   1 RdCost tables   2 fused 2-D transforms (TrQuant::xT / xIT)   4 Quant cores   8 MCTF table entries   16 MCTF whole-picture motion estimation   32 g_tCoeffOps slots
//   64 InterpolationFilter tables   128 MCTF bilateral filter   256 batched sub-pel refinement stages   512 DMVR refinement search per CU   1024 TZ diamond rounds
//   2048 ALF statistics per CTU   4096 CC-ALF statistics per CTU   8192 ALF statistics per picture   16384 ALF filtering per CTU block   32768 CC-ALF filtering per CTU block
//   65536 ALF filtering per picture   262144 residual loop: a CU's component TUs forward-transformed in one device round trip   524288 merge SATD pruning: a CU's regular merge candidates in one device call
//   131072 work-list RECORDER (the encoder keeps its CPU kernels; vvenc_hip_recorder.h; needs no device)
// VVENC_HIP_PRODUCTION: the stages that take whole pictures off the host AND pay at every thread count (what --SIMD=HIP selects): MCTF search + filter, whole-picture ALF
// statistics.  Whole-picture ALF filtering (65536) is bit-exact too but stays out: the first reconstruction task of a picture filters while the others wait on it, which
// costs 2.7 % at 8 encoder threads (profiles/r02_e2e_encoder_fps.md) — a production mask must never lose.
static const int VVENC_HIP_PRODUCTION = 16 + 128 + 8192;
static const int VVENC_HIP_RECORD = 131072;
void recInitRdCost( vvenc::RdCost* rc ) { vvrec::initRdCost( rc ); }

extern "C" __attribute__( ( visibility( "default" ) ) ) int vvenc_hip_install( int mask )
{
  g_slotMask = mask;
  try
  {
    if( ( mask & ~VVENC_HIP_RECORD ) && !g_rd ) { g_rd = new vvhip::RdCost; g_rd->create( true ); g_q = new vvhip::QuantOps; g_m = new vvhip::MCTFOps; g_if = new vvhip::InterpolationFilter; }
  }
  catch( const std::exception& e ) { fprintf( stderr, "vvenc_hip_install: %s\n", e.what() ); return -1; }
  const bool rec = ( mask & VVENC_HIP_RECORD ) != 0;
  if( rec && !vvrec::active() ) { fprintf( stderr, "vvenc_hip_install: the recorder needs $VVHIP_RECORD_DIR\n" ); return -1; }
  g_vvhipHooks.initRdCost = rec ? recInitRdCost : ( mask & 1 ) ? initRdCost : nullptr;      // (the recorder wraps the CPU entries: it excludes the device table)
  g_vvhipHooks.recPicture = rec ? vvrec::picture : nullptr; g_vvhipHooks.recCu = rec ? vvrec::cu : nullptr;
  g_vvhipHooks.recMeBegin = rec ? vvrec::meBegin : nullptr; g_vvhipHooks.recMeEnd = rec ? vvrec::meEnd : nullptr;
  g_vvhipHooks.recStageBegin = rec ? vvrec::stageBegin : nullptr; g_vvhipHooks.recStageCost = rec ? vvrec::stageCost : nullptr; g_vvhipHooks.recStageEnd = rec ? vvrec::stageEnd : nullptr;
  g_vvhipHooks.recTu = rec ? vvrec::tu : nullptr; g_vvhipHooks.recDmvrBegin = rec ? vvrec::dmvrBegin : nullptr; g_vvhipHooks.recDmvrResult = rec ? vvrec::dmvrResult : nullptr;
  g_vvhipHooks.fwd2D      = ( mask & 2 ) ? fwd2D : nullptr;
  g_vvhipHooks.inv2D      = ( mask & 2 ) ? inv2D : nullptr;
  g_vvhipHooks.initQuant  = ( mask & 4 ) ? initQuant : nullptr;
  g_vvhipHooks.initMCTF   = ( mask & 8 ) ? initMCTF : nullptr;
  g_vvhipHooks.mctfMe     = ( mask & 16 ) ? mctfMe : nullptr;
  g_vvhipHooks.mctfWants  = ( mask & 16 ) ? mctfWants : nullptr;
  g_vvhipHooks.mctfPrefetch = ( mask & 16 ) ? mctfPrefetch : nullptr;
  g_vvhipHooks.bindPicture = ( mask & ~VVENC_HIP_RECORD ) ? bindPicture : nullptr;
  g_vvhipHooks.reconRows = ( mask & ( 256 | 1024 ) ) ? reconRows : nullptr;      // the search sites that address reference pictures in HBM
  g_vvhipHooks.initIF     = ( mask & 64 ) ? initIF : nullptr;
  g_vvhipHooks.mctfApply  = ( mask & 128 ) ? mctfApply : nullptr;
  g_vvhipHooks.patternCosts = ( mask & 256 ) ? patternCosts : nullptr;
  g_vvhipHooks.dmvrSearch = ( mask & 512 ) ? dmvrSearch : nullptr;
  g_vvhipHooks.alfCtu = ( mask & 2048 ) ? alfCtu : nullptr; g_alfCtus = 0;
  g_vvhipHooks.ccAlfCtu = ( mask & 4096 ) ? ccAlfCtu : nullptr; g_ccAlfCtus = 0;
  g_vvhipHooks.alfPicture = ( mask & 8192 ) ? alfPicture : nullptr; g_alfPictures = 0;
  g_vvhipHooks.alfPictureOn = alfPictureOn;
  g_vvhipHooks.alfRow = ( mask & 8192 ) ? alfRow : nullptr; g_alfBands = 0; g_alfBandPictures = 0; g_alfRowNs = 0; g_alfPictureNs = 0;
  g_vvhipHooks.alfBeginPicture = ( mask & ( 8192 | 65536 ) ) ? alfBeginPicture : nullptr;
  g_vvhipHooks.alfFilterBlk = ( mask & 16384 ) ? alfFilterBlk : nullptr; g_alfFilterBlks = 0;
  g_vvhipHooks.ccAlfFilterBlk = ( mask & 32768 ) ? ccAlfFilterBlk : nullptr; g_ccAlfFilterBlks = 0;
  g_vvhipHooks.alfFilterPicture = ( mask & 65536 ) ? alfFilterPicture : nullptr; g_alfFilterPictures = 0;
  { std::lock_guard<std::mutex> g( g_alfPicLock ); for( auto& kv : g_alfState ) { kv.second->done = false; kv.second->donePoc = -1; kv.second->bandsOpen = false; kv.second->ops->dropResident(); } }
  g_vvhipHooks.mergeCosts = ( mask & 524288 ) ? mergeCosts : nullptr; g_mergeCalls = 0; g_mergeCands = 0; g_mergeNs = 0; g_tuPrefetchNs = 0;
  g_vvhipHooks.tuPrefetch = ( mask & 262144 ) ? tuPrefetch : nullptr; g_vvhipHooks.tuLookup = ( mask & 262144 ) ? tuLookup : nullptr; g_tuPrefetches = 0; g_tuLookups = 0; g_tuHits = 0;
  g_vvhipHooks.tzReset = ( mask & 1024 ) ? tzReset : nullptr; g_vvhipHooks.tzPrefetch = ( mask & 1024 ) ? tzPrefetch : nullptr; g_vvhipHooks.tzLookup = ( mask & 1024 ) ? tzLookup : nullptr;
  g_tzRounds = 0; g_tzHits = 0;
  g_dmvrCalls = 0;
  g_patternCalls = 0;
  g_lfnstQuantFallbacks = 0; g_lfnstQuantDevice = 0; g_mctfDeviceCalls = 0;
  for( auto& c : g_calls ) c = 0;
  return 0;
}
extern "C" __attribute__( ( visibility( "default" ) ) ) int vvref_install_hip_hooks( int mask ) { return vvenc_hip_install( mask ); }      // (name the tests grew up with)

extern "C" __attribute__( ( visibility( "default" ) ) ) int vvenc_hip_select( const char* spec )
{
  if( !spec || strncmp( spec, "HIP", 3 ) != 0 ) return -1;
  int mask = VVENC_HIP_PRODUCTION;
  if( spec[3] == ':' ) mask = ( int ) strtol( spec + 4, nullptr, 0 );
  else if( spec[3] != 0 ) return -1;
  return vvenc_hip_install( mask );
}

extern "C" __attribute__( ( visibility( "default" ) ) ) void vvenc_hip_release( void )
{
  if( g_vvhipHooks.recPicture ) vvrec::flush();                         // the recorder writes its lists when the encoder closes
  if( !g_rd ) return;                                                  // the binding was never installed: nothing lives on a device
  try
  {
    { std::lock_guard<std::mutex> g( g_mctfLock ); while( !g_resident.empty() ) dropResident( 0 ); g_meCache.curPoc = -1; g_meCache.fields.clear(); }
    { std::lock_guard<std::mutex> g( g_alfPicLock ); for( auto& kv : g_alfState ) { kv.second->done = false; kv.second->donePoc = -1; kv.second->bandsOpen = false; kv.second->ops->dropResident(); } }
    {
      std::lock_guard<std::mutex> g( g_reconLock );
      for( auto& kv : g_recon ) for( int k = 0; k < std::min( numGpus(), 16 ); k++ ) { GpuScope sc( ( vvhip::Device::defaultGpu() + k ) % numGpus() ); vvhip::Device::get().unregisterPicture( kv.second.ids[k] ); }
      g_recon.clear();
    }
    vvhip::Device::unpinAll();
  }
  catch( const std::exception& e ) { fprintf( stderr, "vvenc_hip_release: %s\n", e.what() ); }
}

extern "C" __attribute__( ( visibility( "default" ) ) ) void vvref_hip_hook_calls( uint64_t* out8 )
{
  for( int i = 0; i < 8; i++ ) out8[i] = g_calls[i];
}
// counters: 0-9 table-entry classes (8 interpolation, 9 MCTF filter pictures), 10 sub-pel stages, 11 DMVR searches, 12 TZ rounds, 13 TZ hits, 14 ALF CTUs, 15 CC-ALF CTUs,
// 16 ALF statistics pictures, 17 ALF filter blocks, 18 CC-ALF filter blocks, 19 ALF filter pictures, 20 LFNST TUs left to the CPU quantiser, 21 MCTF device ME calls,
// 22 original-picture uploads, 23 resident hits, 24 device-to-device picture copies, 25 PCIe bytes up, 26 PCIe bytes down, 27 worker contexts, 28 GPUs in use,
// 29 reconstruction CTU-row uploads, 30 search-stage calls served from a resident reference picture, 31 residual-loop prefetches (CUs), 32 transforms served from them, 33 lookups, 34 merge-pruning device calls (CUs), 35 merge candidates scored, 36/37 wall ns spent inside the residual-loop / merge-pruning device calls (all threads)
extern "C" __attribute__( ( visibility( "default" ) ) ) void vvref_hip_hook_calls_ex( uint64_t* out, int n )
{
  for( int i = 0; i < n && i < 10; i++ ) out[i] = g_calls[i];
  if( n > 10 ) out[10] = g_patternCalls;
  if( n > 11 ) out[11] = g_dmvrCalls;
  if( n > 12 ) out[12] = g_tzRounds;
  if( n > 13 ) out[13] = g_tzHits;
  if( n > 14 ) out[14] = g_alfCtus;
  if( n > 15 ) out[15] = g_ccAlfCtus;
  if( n > 16 ) out[16] = g_alfPictures;
  if( n > 17 ) out[17] = g_alfFilterBlks;
  if( n > 18 ) out[18] = g_ccAlfFilterBlks;
  if( n > 19 ) out[19] = g_alfFilterPictures;
  if( n > 20 ) out[20] = g_lfnstQuantFallbacks;
  if( n > 21 ) out[21] = g_mctfDeviceCalls;
  if( n > 22 ) out[22] = g_residentUploads;
  if( n > 23 ) out[23] = g_residentHits;
  if( n > 24 ) out[24] = g_residentPeerCopies;
  if( n > 29 ) out[29] = g_reconRowUploads;
  if( n > 28 )
  {
    const vvhip::Device::Stats st = vvhip::Device::stats();
    out[25] = st.uploadBytes; out[26] = st.downloadBytes; out[27] = st.contexts; out[28] = ( uint64_t ) numGpus();
    if( n > 30 ) out[30] = st.residentReferenceCalls;
    if( n > 31 ) out[31] = g_tuPrefetches;
    if( n > 32 ) out[32] = g_tuHits;
    if( n > 33 ) out[33] = g_tuLookups;
    if( n > 34 ) out[34] = g_mergeCalls;
    if( n > 35 ) out[35] = g_mergeCands;
    if( n > 36 ) out[36] = g_tuPrefetchNs;
    if( n > 37 ) out[37] = g_mergeNs;
    if( n > 39 ) out[39] = g_alfBands;              // statistics-unit rows issued asynchronously by row tasks (round 6)
    if( n > 40 ) out[40] = g_alfBandPictures;       // pictures whose statistics were collected from bands
    if( n > 41 ) out[41] = g_alfRowNs;              // wall ns inside alfRow (worker threads, all pictures)
    if( n > 42 ) out[42] = g_alfPictureNs;          // wall ns inside alfPicture (the serial part of the ALF stage)
    if( n > 38 ) out[38] = g_lfnstQuantDevice;      // LFNST TUs quantised on the device (round 6; 20 = those left to the CPU entry: 0 unless $VVHIP_LFNST_QUANT_ON_CPU=1)
  }
}
