// vvenc_hip_recorder.cpp — see vvenc_hip_recorder.h.  Compiled against the reference's headers (it walks Picture / Slice / CodingStructure / TransformUnit).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>
#include <array>
#include <deque>
#include <list>
#include <sstream>
#include <iostream>
#include <fstream>
#include <thread>
#include <condition_variable>
#include <functional>
#include <chrono>
#include <cmath>
#include <limits>
#include <cassert>
#include <cstdarg>
#include <iomanip>
#include <numeric>
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
#include "CommonLib/Picture.h"
#include "CommonLib/Slice.h"
#include "CommonLib/CodingStructure.h"
#undef private
#undef protected

#include "vvenc_hip_recorder.h"

namespace vvrec {
namespace {

struct PlaneInfo { PlaneRec rec; const int16_t* origin; const int16_t* bufStart; const int16_t* bufEnd; };

struct ThreadLog
{
  std::vector<MeRec> me; std::vector<CandRec> cand; std::vector<StageRec> stage; std::vector<DistRec> dist; std::vector<TuRec> tu; std::vector<DmvrRec> dmvr;
  std::vector<int16_t> pool;
  std::map<std::array<intptr_t, 7>, int32_t> masks;      // GEO weight blocks already in the pool: (pointer, stepX, maskStride, maskStride2, w, h, subShift) -> offset
  uint64_t pairs[3] = { 0, 0, 0 };      // sample pairs through the table: luma, chroma, inside sub-pel stages (subset of luma)
  uint64_t calls = 0;
};

struct PicState
{
  int poc = -1, tlayer = 0, sliceType = 0, qp = 0, width = 0, height = 0;
  std::vector<PlaneInfo> planes;            // [0..2] original Y, Cb, Cr; then the luma reconstruction of every reference picture
  std::vector<int16_t> planeData;           // concatenated samples of the planes (as dumped)
  std::vector<std::unique_ptr<ThreadLog>> logs;
  std::mutex m;
};

std::mutex g_lock;
std::map<int, std::unique_ptr<PicState>> g_pics;            // by POC
std::set<int> g_wanted; bool g_all = true, g_light = false; std::string g_dir; bool g_envRead = false;
vvenc::FpDistFunc g_cpu[vvenc::DF_TOTAL_FUNCTIONS] = {};

void readEnv()
{
  if( g_envRead ) return;
  g_envRead = true;
  const char* d = getenv( "VVHIP_RECORD_DIR" ); if( d ) g_dir = d;
  g_light = getenv( "VVHIP_RECORD_LIGHT" ) != nullptr;          // statistics only: no plane samples, no pool (offsets -2)
  const char* p = getenv( "VVHIP_RECORD_POCS" );
  if( p && *p ) { g_all = false; std::stringstream ss( p ); std::string tok; while( std::getline( ss, tok, ',' ) ) if( !tok.empty() ) g_wanted.insert( atoi( tok.c_str() ) ); }
}

struct CuMap { const int16_t* buf[3]; int stride[3]; int x[3], y[3], w[3], h[3]; };

std::atomic<int> g_epoch{ 0 };                     // flush() invalidates what the worker threads remember

struct ThreadState
{
  int epoch = -1;
  const void* pic = nullptr; int pocSeen = -1; PicState* ps = nullptr; ThreadLog* log = nullptr;
  CuMap cus[12]; int nCus = 0, nextCu = 0;
  // motion-estimation context
  bool inMe = false, inStage = false; int meIdx = -1; int mePlane = -1; int stageIdx = -1;
  // DMVR context
  bool inDmvr = false; DmvrRec dmvr; int dmvrCuW = 0, dmvrDx = 0, dmvrDy = 0;
  // pool de-duplication (the last few compact blocks copied)
  struct Recent { const int16_t* p; int w, h, stride; uint32_t sum; int32_t off; } recent[8]; int nRecent = 0, nextRecent = 0;
};
thread_local ThreadState t_s;

uint32_t checksum( const int16_t* p, int stride, int w, int h )
{
  uint32_t s = 2166136261u;
  for( int y = 0; y < h; y++ ) for( int x = 0; x < w; x++ ) s = ( s ^ ( uint16_t ) p[y * stride + x] ) * 16777619u;
  return s;
}

int32_t toPool( const int16_t* p, int stride, int w, int h )
{
  ThreadState& t = t_s;
  if( g_light ) return -2;
  const uint32_t sum = checksum( p, stride, w, h );
  for( int i = 0; i < t.nRecent; i++ ) { const auto& r = t.recent[i]; if( r.p == p && r.w == w && r.h == h && r.stride == stride && r.sum == sum ) return r.off; }
  std::vector<int16_t>& pool = t.log->pool;
  while( pool.size() & 7 ) pool.push_back( 0 );                     // blocks start 16-byte aligned
  const int32_t off = ( int32_t ) pool.size();
  for( int y = 0; y < h; y++ ) pool.insert( pool.end(), p + ( ptrdiff_t ) y * stride, p + ( ptrdiff_t ) y * stride + w );
  t.recent[t.nextRecent] = { p, w, h, stride, sum, off };
  t.nextRecent = ( t.nextRecent + 1 ) & 7; if( t.nRecent < 8 ) t.nRecent++;
  return off;
}

bool inPlane( const PlaneInfo& pl, const int16_t* p, int stride, int w, int h, int& x, int& y )
{
  if( p < pl.bufStart || p >= pl.bufEnd || stride != pl.rec.stride ) return false;
  const ptrdiff_t off = p - pl.bufStart;
  y = ( int ) ( off / stride ) - pl.rec.margin; x = ( int ) ( off % stride ) - pl.rec.margin;
  return x >= -pl.rec.margin && y >= -pl.rec.margin && x + w <= pl.rec.width + pl.rec.margin && y + h <= pl.rec.height + pl.rec.margin;
}

Operand classify( const int16_t* p, int stride, int w, int h, bool allowPool = true )
{
  ThreadState& t = t_s;
  Operand o; o.pad = 0;
  for( int i = 0; i < t.nCus; i++ )
  {
    const CuMap& c = t.cus[i];
    for( int k = 0; k < 3; k++ )
    {
      if( !c.buf[k] || stride != c.stride[k] || p < c.buf[k] || p >= c.buf[k] + ( ptrdiff_t ) c.h[k] * c.stride[k] ) continue;
      const ptrdiff_t off = p - c.buf[k];
      const int dy = ( int ) ( off / stride ), dx = ( int ) ( off % stride );
      if( dx + w > c.w[k] || dy + h > c.h[k] ) continue;
      o.plane = ( int16_t ) k; o.x = c.x[k] + dx; o.y = c.y[k] + dy;
      return o;
    }
  }
  for( size_t i = 3; i < t.ps->planes.size(); i++ )
  {
    int x, y;
    if( inPlane( t.ps->planes[i], p, stride, w, h, x, y ) ) { o.plane = ( int16_t ) i; o.x = x; o.y = y; return o; }
  }
  o.plane = -1; o.x = allowPool ? toPool( p, stride, w, h ) : -1; o.y = w;
  return o;
}

template<int IDX> vvenc::Distortion recTramp( const vvenc::DistParam& dp )
{
  const vvenc::Distortion d = g_cpu[IDX]( dp );
  ThreadState& t = t_s;
  if( !t.log || t.epoch != g_epoch ) return d;
  if( dp.dmvrSadX5 ) return d;                                 // DMVR's own centre cost (setDistParam( .., isDMVR ), RdCost.cpp:228-265): recorded by dmvrResult; decoder-side work, like the reference's counter (:256-262)
  const int w = dp.org.width, h = dp.org.height;
  t.log->calls++;
  t.log->pairs[dp.compID == vvenc::COMP_Y || dp.compID == vvenc::MAX_NUM_COMP ? 0 : 1] += ( uint64_t ) w * h;
  if( t.inStage ) { t.log->pairs[2] += ( uint64_t ) w * h; return d; }                // (recorded by stageCost)
  if( t.inMe )
  {
    int x, y;
    if( t.mePlane >= 0 && inPlane( t.ps->planes[t.mePlane], dp.cur.buf, ( int ) dp.cur.stride, w, h, x, y ) )
    {
      // The x86 rows of the 64-wide SAD return a PARTIAL sum as soon as it exceeds maximumDistortionForEarlyExit (checked every fourth processed row,
      // x86/RdCostX86.h:390-410, 535-555; the scalar row after every row, RdCost.cpp:326): any such value only ever loses the comparison with the best cost.  The record holds what
      // the same table entry returns with the exit disabled — the value a device list delivers — and flags that the encoder saw the partial one.
      vvenc::Distortion full = d;
      if( w >= 64 && dp.maximumDistortionForEarlyExit != vvenc::MAX_DISTORTION ) { vvenc::DistParam q = dp; q.maximumDistortionForEarlyExit = vvenc::MAX_DISTORTION; full = g_cpu[IDX]( q ); }
      CandRec c; c.me = t.meIdx; c.x = x; c.y = y; c.df = ( uint8_t ) IDX; c.subShift = ( uint8_t ) dp.subShift; c.pad0 = full != d; c.pad1 = 0; c.cost = full;
      t.log->cand.push_back( c );
      t.log->me[t.meIdx].nCand++;
      return d;
    }
  }
  DistRec r; r.df = ( uint8_t ) IDX; r.subShift = ( uint8_t ) dp.subShift; r.bitDepth = ( uint8_t ) dp.bitDepth; r.ctx = t.inMe ? 1 : 0; r.w = ( int16_t ) w; r.h = ( int16_t ) h;
  r.org = classify( dp.org.buf, ( int ) dp.org.stride, w, h );
  r.cur = classify( dp.cur.buf, ( int ) dp.cur.stride, w, h );
  r.cost = d;
  r.maskPool = -1;
  if( IDX == vvenc::DF_SAD_WITH_MASK && dp.mask && !g_light )
  {
    // the weights as the call walks them (xGetSADwMask, RdCost.cpp:2062-2093): per evaluated row w values stepX apart, then maskStride << subShift and maskStride2 further
    const std::array<intptr_t, 7> key = { ( intptr_t ) dp.mask, dp.stepX, dp.maskStride, dp.maskStride2, w, h, dp.subShift };
    auto f = t.log->masks.find( key );
    if( f == t.log->masks.end() )
    {
      std::vector<int16_t>& pool = t.log->pool;
      while( pool.size() & 7 ) pool.push_back( 0 );
      const int32_t off = ( int32_t ) pool.size();
      const int16_t* m = dp.mask;
      for( int y = 0; y < ( h >> dp.subShift ); y++ )
      {
        for( int x = 0; x < w; x++ ) { pool.push_back( *m ); m += dp.stepX; }
        m += ( ptrdiff_t ) dp.maskStride * ( 1 << dp.subShift ) + dp.maskStride2;
      }
      f = t.log->masks.emplace( key, off ).first;
    }
    r.maskPool = f->second;
  }
  t.log->dist.push_back( r );
  return d;
}
template<int I> struct Wrap { static void go( vvenc::RdCost* rc ) { if( rc->m_afpDistortFunc[0][I] != recTramp<I> ) g_cpu[I] = rc->m_afpDistortFunc[0][I]; rc->m_afpDistortFunc[0][I] = recTramp<I>; Wrap<I - 1>::go( rc ); } };
template<> struct Wrap<-1> { static void go( vvenc::RdCost* ) {} };

int addPlane( PicState& ps, int kind, int poc, int comp, const vvenc::CPelBuf& b, int margin, bool withMargin )
{
  PlaneInfo pl;
  pl.rec.kind = kind; pl.rec.poc = poc; pl.rec.comp = comp; pl.rec.width = b.width; pl.rec.height = b.height; pl.rec.stride = ( int ) b.stride; pl.rec.margin = withMargin ? margin : 0;
  pl.origin = b.buf; pl.bufStart = b.buf - ( ptrdiff_t ) margin * b.stride - margin; pl.bufEnd = b.buf + ( ptrdiff_t ) ( b.height + margin ) * b.stride;
  if( !withMargin ) { pl.bufStart = b.buf; pl.bufEnd = b.buf + ( ptrdiff_t ) b.height * b.stride; }
  pl.rec.fileOffset = ( int64_t ) ps.planeData.size();
  const int m = pl.rec.margin;
  // dumped with its margin, tightly packed: (height + 2m) rows of (width + 2m) samples
  if( !g_light ) for( int y = -m; y < ( int ) b.height + m; y++ ) ps.planeData.insert( ps.planeData.end(), b.buf + ( ptrdiff_t ) y * b.stride - m, b.buf + ( ptrdiff_t ) y * b.stride + b.width + m );
  ps.planes.push_back( pl );
  return ( int ) ps.planes.size() - 1;
}

} // namespace

bool active() { readEnv(); return !g_dir.empty(); }

void initRdCost( void* rdCost ) { Wrap<vvenc::DF_TOTAL_FUNCTIONS - 1>::go( static_cast<vvenc::RdCost*>( rdCost ) ); }

void picture( const void* picture )
{
  ThreadState& t = t_s;
  if( t.epoch == g_epoch && t.pic == picture && t.pocSeen == static_cast<const vvenc::Picture*>( picture )->poc ) return;
  readEnv();
  const vvenc::Picture* pic = static_cast<const vvenc::Picture*>( picture );
  t.epoch = g_epoch; t.pocSeen = pic->poc;
  t.pic = picture; t.ps = nullptr; t.log = nullptr; t.nCus = 0; t.nextCu = 0; t.inMe = t.inStage = t.inDmvr = false; t.nRecent = 0;
  if( !g_all && !g_wanted.count( pic->poc ) ) return;
  std::lock_guard<std::mutex> g( g_lock );
  std::unique_ptr<PicState>& slot = g_pics[pic->poc];
  if( !slot )
  {
    slot.reset( new PicState );
    PicState& ps = *slot;
    ps.poc = pic->poc;
    const vvenc::Slice* sl = pic->slices.empty() ? nullptr : pic->slices[0];
    ps.tlayer = sl ? ( int ) sl->TLayer : 0; ps.sliceType = sl ? ( int ) sl->sliceType : 0; ps.qp = sl ? sl->sliceQp : 0;
    const vvenc::CPelUnitBuf org = const_cast<vvenc::Picture*>( pic )->getFilteredOrigBuffer().valid() ? pic->getFiltOrigBuf() : pic->getOrigBuf();
    ps.width = org.Y().width; ps.height = org.Y().height;
    for( int c = 0; c < 3; c++ ) addPlane( ps, 0, pic->poc, c, org.bufs[c], 0, false );
    if( sl ) for( int l = 0; l < 2; l++ ) for( int i = 0; i < sl->numRefIdx[l]; i++ )
    {
      const vvenc::Picture* rp = sl->getRefPic( vvenc::RefPicList( l ), i );
      if( !rp ) continue;
      bool have = false;
      for( const PlaneInfo& pl : ps.planes ) if( pl.rec.kind == 1 && pl.rec.poc == rp->poc ) have = true;
      if( !have ) addPlane( ps, 1, rp->poc, 0, rp->getRecoBuf( vvenc::COMP_Y ), ( int ) rp->margin, true );
    }
  }
  t.ps = slot.get();
  t.ps->logs.emplace_back( new ThreadLog );
  t.log = t.ps->logs.back().get();
}

void cu( const void* codingStructure )
{
  ThreadState& t = t_s;
  if( !t.log || t.epoch != g_epoch ) return;
  const vvenc::CodingStructure* cs = static_cast<const vvenc::CodingStructure*>( codingStructure );
  if( !cs->m_org ) return;
  CuMap m;
  for( int c = 0; c < 3; c++ )
  {
    const bool ok = c < ( int ) cs->m_org->bufs.size() && c < ( int ) cs->area.blocks.size() && cs->area.blocks[c].valid();
    m.buf[c] = ok ? cs->m_org->bufs[c].buf : nullptr;
    if( !ok ) continue;
    m.stride[c] = ( int ) cs->m_org->bufs[c].stride; m.x[c] = cs->area.blocks[c].x; m.y[c] = cs->area.blocks[c].y; m.w[c] = cs->area.blocks[c].width; m.h[c] = cs->area.blocks[c].height;
  }
  for( int i = 0; i < t.nCus; i++ ) if( t.cus[i].buf[0] == m.buf[0] ) { t.cus[i] = m; return; }       // the per-depth buffer now holds this block
  t.cus[t.nextCu] = m; t.nextCu = ( t.nextCu + 1 ) % 12; if( t.nCus < 12 ) t.nCus++;
}

void meBegin( int cuX, int cuY, int w, int h, int list, int /*refIdx*/, int /*refPoc*/, bool bi, const int16_t* pattern, int patternStride, const int16_t* refY, int refStride )
{
  ThreadState& t = t_s;
  if( !t.log || t.epoch != g_epoch ) return;
  MeRec m; m.cuX = cuX; m.cuY = cuY; m.w = ( int16_t ) w; m.h = ( int16_t ) h; m.bi = bi; m.list = ( uint8_t ) list;
  m.firstCand = ( int32_t ) t.log->cand.size(); m.nCand = 0; m.firstStage = ( int32_t ) t.log->stage.size(); m.nStage = 0;
  t.mePlane = -1;
  for( size_t i = 3; i < t.ps->planes.size(); i++ ) { int x, y; if( inPlane( t.ps->planes[i], refY, refStride, w, h, x, y ) ) { t.mePlane = ( int ) i; break; } }
  m.refPlane = ( int16_t ) t.mePlane;
  const Operand o = classify( pattern, patternStride, w, h );
  m.patternPool = ( o.plane == 0 && o.x == cuX && o.y == cuY ) ? -1 : ( o.plane == -1 ? o.x : toPool( pattern, patternStride, w, h ) );
  t.meIdx = ( int ) t.log->me.size();
  t.log->me.push_back( m );
  t.inMe = true; t.inStage = false;
}

void meEnd() { t_s.inMe = false; t_s.inStage = false; }

void stageBegin( const int16_t* patternRoi, int baseHor, int baseVer, int iFrac, int hadMode, int reduceTap, bool altHpel )
{
  ThreadState& t = t_s;
  if( !t.log || t.epoch != g_epoch || !t.inMe || t.mePlane < 0 ) return;
  const MeRec& m = t.log->me[t.meIdx];
  int x, y;
  if( !inPlane( t.ps->planes[t.mePlane], patternRoi, t.ps->planes[t.mePlane].rec.stride, m.w, m.h, x, y ) ) return;
  StageRec s; s.me = t.meIdx; s.baseX = x; s.baseY = y; s.baseHor = ( int16_t ) baseHor; s.baseVer = ( int16_t ) baseVer; s.iFrac = ( uint8_t ) iFrac; s.hadMode = ( uint8_t ) hadMode;
  s.reduceTap = ( uint8_t ) reduceTap; s.altHpel = altHpel; s.pad = 0;
  for( int i = 0; i < 9; i++ ) s.cost[i] = ~0ull;
  t.stageIdx = ( int ) t.log->stage.size();
  t.log->stage.push_back( s );
  t.log->me[t.meIdx].nStage++;
  t.inStage = true;
}
void stageCost( int i, uint64_t dist ) { ThreadState& t = t_s; if( t.log && t.epoch == g_epoch && t.inStage && i >= 0 && i < 9 ) t.log->stage[t.stageIdx].cost[i] = dist; }
void stageEnd() { t_s.inStage = false; }

void tu( const void* transformUnit, int comp, const int16_t* resi, long stride, int w, int h, int trHor, int trVer, int bitDepth )
{
  ThreadState& t = t_s;
  if( !t.log || t.epoch != g_epoch ) return;
  const vvenc::TransformUnit& u = *static_cast<const vvenc::TransformUnit*>( transformUnit );
  const vvenc::QpParam qp( u, vvenc::ComponentID( comp ), true );
  TuRec r; r.comp = ( uint8_t ) comp; r.trHor = ( uint8_t ) trHor; r.trVer = ( uint8_t ) trVer;
  r.flags = ( u.cs->slice->isIRAP() ? 1 : 0 ) | ( comp == 0 ? 2 : 0 ) | ( u.cu->predMode == vvenc::MODE_INTRA ? 4 : 0 );
  r.w = ( int16_t ) w; r.h = ( int16_t ) h; r.qp = ( int16_t ) qp.Qp( false ); r.bitDepth = ( int16_t ) bitDepth; r.x = u.blocks[comp].x; r.y = u.blocks[comp].y;
  r.pool = toPool( resi, ( int ) stride, w, h );
  t.log->tu.push_back( r );
}

void dmvrBegin( const void* /*cu*/, const int16_t* ref0, int stride0, int fx0, int fy0, const int16_t* ref1, int stride1, int fx1, int fy1, int cuW, int cuH, int dx, int dy )
{
  ThreadState& t = t_s;
  t.inDmvr = false;
  if( !t.log || t.epoch != g_epoch ) return;
  DmvrRec d; memset( &d, 0, sizeof( d ) );
  d.ref0Plane = d.ref1Plane = -1;
  for( size_t i = 3; i < t.ps->planes.size(); i++ )
  {
    int x, y;
    if( d.ref0Plane < 0 && inPlane( t.ps->planes[i], ref0, stride0, cuW, cuH, x, y ) ) { d.ref0Plane = ( int16_t ) i; d.x0 = x; d.y0 = y; }
    if( d.ref1Plane < 0 && inPlane( t.ps->planes[i], ref1, stride1, cuW, cuH, x, y ) ) { d.ref1Plane = ( int16_t ) i; d.x1 = x; d.y1 = y; }
  }
  if( d.ref0Plane < 0 || d.ref1Plane < 0 ) return;
  d.frac0x = ( int16_t ) fx0; d.frac0y = ( int16_t ) fy0; d.frac1x = ( int16_t ) fx1; d.frac1y = ( int16_t ) fy1; d.dx = ( int16_t ) dx; d.dy = ( int16_t ) dy;
  t.dmvr = d; t.dmvrCuW = cuW; t.dmvrDx = dx; t.dmvrDy = dy; t.inDmvr = true;
}

void dmvrResult( int num, int mvdX, int mvdY, uint64_t minCost )
{
  ThreadState& t = t_s;
  if( !t.log || t.epoch != g_epoch || !t.inDmvr ) return;
  const int perRow = t.dmvrCuW / t.dmvrDx;
  DmvrRec d = t.dmvr;
  const int sx = ( num % perRow ) * t.dmvrDx, sy = ( num / perRow ) * t.dmvrDy;
  d.x0 += sx; d.y0 += sy; d.x1 += sx; d.y1 += sy; d.mvdX = ( int16_t ) mvdX; d.mvdY = ( int16_t ) mvdY; d.minCost = minCost;
  t.log->dmvr.push_back( d );
}

namespace {
template<class T> void writeArr( FILE* f, const std::vector<T>& v ) { if( !v.empty() ) fwrite( v.data(), sizeof( T ), v.size(), f ); }
}

void flush()
{
  readEnv();
  std::lock_guard<std::mutex> g( g_lock );
  g_epoch++;
  if( g_dir.empty() ) { g_pics.clear(); return; }
  for( auto& kv : g_pics )
  {
    PicState& ps = *kv.second;
    // merge the per-thread logs: indices into me / cand / stage and pool offsets are rebased
    ThreadLog all;
    for( auto& lp : ps.logs )
    {
      ThreadLog& l = *lp;
      const int32_t me0 = ( int32_t ) all.me.size(), cand0 = ( int32_t ) all.cand.size(), stage0 = ( int32_t ) all.stage.size();
      while( all.pool.size() & 7 ) all.pool.push_back( 0 );
      const int32_t pool0 = ( int32_t ) all.pool.size();
      for( MeRec m : l.me ) { m.firstCand += cand0; m.firstStage += stage0; if( m.patternPool >= 0 ) m.patternPool += pool0; all.me.push_back( m ); }
      for( CandRec c : l.cand ) { c.me += me0; all.cand.push_back( c ); }
      for( StageRec s : l.stage ) { s.me += me0; all.stage.push_back( s ); }
      for( DistRec d : l.dist ) { if( d.org.plane < 0 && d.org.x >= 0 ) d.org.x += pool0; if( d.cur.plane < 0 && d.cur.x >= 0 ) d.cur.x += pool0; if( d.maskPool >= 0 ) d.maskPool += pool0; all.dist.push_back( d ); }
      for( TuRec u : l.tu ) { u.pool += pool0; all.tu.push_back( u ); }
      all.dmvr.insert( all.dmvr.end(), l.dmvr.begin(), l.dmvr.end() );
      all.pool.insert( all.pool.end(), l.pool.begin(), l.pool.end() );
      for( int i = 0; i < 3; i++ ) all.pairs[i] += l.pairs[i];
      all.calls += l.calls;
    }
    const std::string base = g_dir + "/poc" + std::to_string( ps.poc );
    FILE* f = fopen( ( base + ".bin" ).c_str(), "wb" );
    if( !f ) { fprintf( stderr, "vvrec: cannot write %s.bin\n", base.c_str() ); continue; }
    std::vector<PlaneRec> planes; for( const PlaneInfo& pl : ps.planes ) planes.push_back( pl.rec );
    struct Sec { const char* name; size_t bytes, count; };
    const Sec secs[] = { { "planes", planes.size() * sizeof( PlaneRec ), planes.size() }, { "plane_data", ps.planeData.size() * 2, ps.planeData.size() }, { "me", all.me.size() * sizeof( MeRec ), all.me.size() },
                         { "cand", all.cand.size() * sizeof( CandRec ), all.cand.size() }, { "stage", all.stage.size() * sizeof( StageRec ), all.stage.size() },
                         { "dist", all.dist.size() * sizeof( DistRec ), all.dist.size() }, { "tu", all.tu.size() * sizeof( TuRec ), all.tu.size() },
                         { "dmvr", all.dmvr.size() * sizeof( DmvrRec ), all.dmvr.size() }, { "pool", all.pool.size() * 2, all.pool.size() } };
    writeArr( f, planes ); writeArr( f, ps.planeData ); writeArr( f, all.me ); writeArr( f, all.cand ); writeArr( f, all.stage ); writeArr( f, all.dist ); writeArr( f, all.tu ); writeArr( f, all.dmvr ); writeArr( f, all.pool );
    fclose( f );
    FILE* j = fopen( ( base + ".json" ).c_str(), "w" );
    if( !j ) continue;
    fprintf( j, "{\"poc\": %d, \"tlayer\": %d, \"slice_type\": %d, \"slice_qp\": %d, \"width\": %d, \"height\": %d, \"table_calls\": %llu, \"sample_pairs_luma\": %llu, \"sample_pairs_chroma\": %llu, "
                "\"sample_pairs_in_subpel_stages\": %llu, \"threads\": %d, \"sections\": [", ps.poc, ps.tlayer, ps.sliceType, ps.qp, ps.width, ps.height, ( unsigned long long ) all.calls,
             ( unsigned long long ) all.pairs[0], ( unsigned long long ) all.pairs[1], ( unsigned long long ) all.pairs[2], ( int ) ps.logs.size() );
    size_t off = 0;
    for( size_t i = 0; i < sizeof( secs ) / sizeof( secs[0] ); i++ )
    {
      fprintf( j, "%s{\"name\": \"%s\", \"offset\": %zu, \"bytes\": %zu, \"count\": %zu}", i ? ", " : "", secs[i].name, off, secs[i].bytes, secs[i].count );
      off += secs[i].bytes;
    }
    fprintf( j, "]}\n" );
    fclose( j );
  }
  g_pics.clear();
}

} // namespace vvrec
