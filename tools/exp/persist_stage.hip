// persist_stage.hip — ONE experiment for DESIGN 7a: what does a sub-pel refinement stage cost a SYNCHRONOUS caller when no kernel is launched at all?
//
// The encoder's xPatternRefinement site (hook bit 256) pays launch + hipStreamSynchronize per stage (48 us per call measured in round 2/3).  Here ONE persistent workgroup
// spins on a request ring in fine-grained host-mapped memory: the host writes a stage record (the same StageUnit the plan's kernels read) and a sequence number, the
// resident kernel runs me.hip's stage body on it (reference and original pictures already resident in HBM), writes the nine costs into host-mapped memory and a completion
// number; the host spins on that.  Nothing is launched, nothing is synchronised: what remains is the PCIe round trip of the doorbell + the kernel's own latency chain.
// Compared in the same process with the launch form (the same unit through meStageKernel + hipStreamSynchronize).  Results are checked to be identical.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVVHIP_ME_KERNELS_ONLY tools/exp/persist_stage.hip -o tools/exp/persist_stage
//   run:   timeout 60 tools/exp/persist_stage [calls=2000] [size=16]      -> one JSON line
#include <chrono>
#include <thread>
#include "../../vvenc_amd/csrc/me.hip"

#define CK( x ) do { hipError_t e_ = ( x ); if( e_ != hipSuccess ) { fprintf( stderr, "%s: %s\n", #x, hipGetErrorString( e_ ) ); exit( 1 ); } } while( 0 )

struct Ring { volatile uint32_t req; uint32_t pad0[15]; volatile uint32_t done; uint32_t pad1[15]; };      // two cache lines: the host writes req, the device writes done

// one workgroup of two waves (the stage kernel's shape); wave 0 serves the ring, wave 1 only joins the barriers a shared stage would need (none here: blocks <= 32 rows)
__global__ void __launch_bounds__( 128 )
persistStageKernel( MePlanes P, MeArgs a, Ring* ring, int nCalls, int ldsPerWave, long long spinLimit )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) int16_t meLds[];
  __shared__ uint32_t pairCost[16];
  const int wv = __builtin_amdgcn_readfirstlane( ( int ) ( threadIdx.x >> 6 ) );
  if( wv != 0 ) return;
  for( int call = 1; call <= nCalls; call++ )
  {
    long long spins = 0;
    // the doorbell: system-scope acquire load of the request number (lane-uniform: every lane polls the same word; fine-grained host memory is not cached)
    for( ;; )
    {
      const uint32_t r = __hip_atomic_load( const_cast<uint32_t*>( &ring->req ), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM );
      if( r == 0xffffffffu || ++spins > spinLimit ) return;                        // the host gave up / went away: a bounded spin never leaves a kernel behind
      if( r >= ( uint32_t ) call ) break;
    }
    __builtin_amdgcn_s_dcache_inv();                                                // the unit record is read through the scalar cache
    WaveSpan span; span.first = call - 1; span.count = 1;                           // request `call` = unit record call - 1 (host-mapped), costs -> 9 * ( call - 1 ) (host-mapped)
    stageBody<2, 5, false>( P, a, span, meLds, pairCost, 0 );
    __threadfence_system();
    if( ( threadIdx.x & 63 ) == 0 ) __hip_atomic_store( const_cast<uint32_t*>( &ring->done ), ( uint32_t ) call, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM );
  }
}

int main( int argc, char** argv )
{
  const int nCalls = argc > 1 ? atoi( argv[1] ) : 2000, S = argc > 2 ? atoi( argv[2] ) : 16;
  const int W = 1920, H = 1080, M = 80, stride = ( W + 2 * M + 7 ) / 8 * 8, rows = H + 2 * M + 1;
  std::vector<int16_t> hostPlane( ( size_t ) stride * rows );
  srand( 7 );
  for( auto& v : hostPlane ) v = ( int16_t ) ( rand() & 1023 );
  int16_t* dOrg = nullptr; int16_t* dRef = nullptr;
  CK( hipMalloc( &dOrg, hostPlane.size() * 2 ) ); CK( hipMalloc( &dRef, hostPlane.size() * 2 ) );
  CK( hipMemcpy( dRef, hostPlane.data(), hostPlane.size() * 2, hipMemcpyHostToDevice ) );
  for( auto& v : hostPlane ) v = ( int16_t ) ( rand() & 1023 );
  CK( hipMemcpy( dOrg, hostPlane.data(), hostPlane.size() * 2, hipMemcpyHostToDevice ) );
  // tap tables exactly as plan creation builds them (4-tap search set only: table 4)
  const std::vector<int32_t> tapTab = stageTapTables( 10 );
  int32_t* dTap = nullptr; CK( hipMalloc( &dTap, tapTab.size() * 4 ) ); CK( hipMemcpy( dTap, tapTab.data(), tapTab.size() * 4, hipMemcpyHostToDevice ) );
  // host-mapped, fine-grained: the request records, the costs, the ring
  StageUnit* hUnits = nullptr; uint64_t* hCost = nullptr; Ring* hRing = nullptr;
  CK( hipHostMalloc( ( void** ) &hUnits, sizeof( StageUnit ) * nCalls, hipHostMallocMapped | hipHostMallocCoherent ) );
  CK( hipHostMalloc( ( void** ) &hCost, 8 * 9 * ( size_t ) nCalls, hipHostMallocMapped | hipHostMallocCoherent ) );
  CK( hipHostMalloc( ( void** ) &hRing, sizeof( Ring ), hipHostMallocMapped | hipHostMallocCoherent ) );
  memset( ( void* ) hRing, 0, sizeof( Ring ) ); memset( hCost, 0xff, 8 * 9 * ( size_t ) nCalls );
  StageUnit* dUnits; uint64_t* dCost; Ring* dRing;
  CK( hipHostGetDevicePointer( ( void** ) &dUnits, hUnits, 0 ) ); CK( hipHostGetDevicePointer( ( void** ) &dCost, hCost, 0 ) ); CK( hipHostGetDevicePointer( ( void** ) &dRing, ( void* ) hRing, 0 ) );
  auto makeUnit = [&]( int i, StageUnit& u )
  {
    memset( &u, 0, sizeof( u ) );
    const int x = 64 + ( rand() % ( W - 192 ) ), y = 64 + ( rand() % ( H - 192 ) );
    u.j.org_off = ( y + M ) * stride + x + M; u.j.ref_off = ( y + M + ( rand() % 5 ) - 2 ) * stride + x + M + ( rand() % 5 ) - 2; u.j.width = ( int16_t ) S; u.j.height = ( int16_t ) S;
    u.j.org_plane = 0; u.j.ref_plane = 1; u.j.i_frac = 2; u.j.filter_mode = 2; u.j.alt_hpel = 0; u.j.func = S >= 32 ? VVHIP_DF_HAD_FAST : VVHIP_DF_HAD; u.j.mask = 0x1ff;
    u.order = i;
    static const int8_t rx[9] = { 0, 0, 0, -1, 1, -1, 1, -1, 1 }, ry[9] = { 0, -1, 1, 0, 0, -1, -1, 1, 1 };
    int cnt[3] = { 0, 0, 0 }, var[9], nHor = 0;
    for( int k = 0; k < 9; k++ ) { const int tx = rx[k] * 8; int v = 0; while( v < nHor && u.hx[v] != tx ) v++; if( v == nHor ) u.hx[nHor++] = ( int16_t ) tx; var[k] = v; cnt[v]++; }
    u.nHor = ( uint8_t ) nHor; u.nPos = 9; u.cnt0 = ( uint8_t ) cnt[0]; u.cnt1 = ( uint8_t ) cnt[1]; u.cnt2 = ( uint8_t ) cnt[2];
    int slot[3] = { 0, cnt[0], cnt[0] + cnt[1] };
    for( int k = 0; k < 9; k++ ) u.pos[slot[var[k]]++] = k | ( ( rx[k] * 8 + 64 ) << 8 ) | ( ( ry[k] * 8 + 64 ) << 20 );
  };
  std::vector<StageUnit> units( nCalls );
  for( int i = 0; i < nCalls; i++ ) makeUnit( i, units[i] );
  MePlanes P; for( int i = 0; i < 16; i++ ) { P.p[i] = ( i & 1 ) ? dRef : dOrg; P.stride[i] = stride; }
  MeArgs a; memset( &a, 0, sizeof( a ) ); a.tapTables = dTap; a.bitDepth = 10;
  const int ldsSt = ( ( 2 * ( 128 + 64 + 16 + 16 ) + 3 * ( 32 + 4 ) * ( 64 + 8 ) ) * 2 + 15 ) & ~15;
  hipStream_t sPersist, sLaunch; CK( hipStreamCreateWithFlags( &sPersist, hipStreamNonBlocking ) ); CK( hipStreamCreateWithFlags( &sLaunch, hipStreamNonBlocking ) );

  // ---- (1) the launch form: one meStageKernel launch + hipStreamSynchronize per stage, unit records and costs in DEVICE memory (uploaded once: the best case of that form)
  StageUnit* devUnits; uint64_t* devCost; WaveSpan* devSpans;
  CK( hipMalloc( &devUnits, sizeof( StageUnit ) * nCalls ) ); CK( hipMalloc( &devCost, 8 * 9 * ( size_t ) nCalls ) ); CK( hipMalloc( &devSpans, sizeof( WaveSpan ) * nCalls ) );
  CK( hipMemcpy( devUnits, units.data(), sizeof( StageUnit ) * nCalls, hipMemcpyHostToDevice ) );
  { std::vector<WaveSpan> sp( nCalls ); for( int i = 0; i < nCalls; i++ ) { sp[i].first = i; sp[i].count = 1; } CK( hipMemcpy( devSpans, sp.data(), sizeof( WaveSpan ) * nCalls, hipMemcpyHostToDevice ) ); }
  MeArgs al = a; al.stageUnits = devUnits; al.stageWaves = devSpans; al.stageCost = devCost;
  std::vector<double> tLaunch;
  std::vector<uint64_t> got( 9 );
  for( int i = 0; i < nCalls; i++ )
  {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL( ( meStageKernel<2, 5, false> ), dim3( 1 ), dim3( 128 ), ( size_t ) 2 * ldsSt, sLaunch, P, al, i, 1, ldsSt );
    CK( hipMemcpyAsync( got.data(), devCost + 9 * ( size_t ) i, 72, hipMemcpyDeviceToHost, sLaunch ) );          // the nine costs back to the caller
    CK( hipStreamSynchronize( sLaunch ) );
    tLaunch.push_back( std::chrono::duration<double, std::micro>( std::chrono::steady_clock::now() - t0 ).count() );
  }
  std::vector<uint64_t> ref( 9 * ( size_t ) nCalls );
  CK( hipMemcpy( ref.data(), devCost, 8 * 9 * ( size_t ) nCalls, hipMemcpyDeviceToHost ) );

  // ---- (2) the persistent form
  MeArgs ap = a; ap.stageUnits = dUnits; ap.stageCost = dCost;
  hipLaunchKernelGGL( persistStageKernel, dim3( 1 ), dim3( 128 ), ( size_t ) 2 * ldsSt, sPersist, P, ap, dRing, nCalls, ldsSt, 20000000ll );
  std::this_thread::sleep_for( std::chrono::milliseconds( 20 ) );      // the kernel is resident and polling
  std::vector<double> tPersist;
  bool timeout = false;
  for( int i = 0; i < nCalls && !timeout; i++ )
  {
    const auto t0 = std::chrono::steady_clock::now();
    hUnits[i] = units[i];                                                           // the request record (88 bytes) into host-mapped memory
    __atomic_store_n( &hRing->req, ( uint32_t ) ( i + 1 ), __ATOMIC_RELEASE );       // the doorbell
    while( __atomic_load_n( &hRing->done, __ATOMIC_ACQUIRE ) < ( uint32_t ) ( i + 1 ) )
      if( std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count() > 2.0 ) { timeout = true; break; }
    tPersist.push_back( std::chrono::duration<double, std::micro>( std::chrono::steady_clock::now() - t0 ).count() );
  }
  if( timeout ) __atomic_store_n( &hRing->req, 0xffffffffu, __ATOMIC_RELEASE );
  CK( hipStreamSynchronize( sPersist ) );
  size_t bad = 0;
  for( size_t i = 0; i < 9 * ( size_t ) ( timeout ? 0 : nCalls ); i++ ) bad += hCost[i] != ref[i];
  auto med = []( std::vector<double> v, double q ) { std::sort( v.begin(), v.end() ); return v.empty() ? 0.0 : v[( size_t ) ( q * ( v.size() - 1 ) )]; };
  printf( "{\"experiment\": \"sub-pel refinement stage of one %dx%d block for a synchronous caller\", \"calls\": %d, "
          "\"launch_plus_sync_us\": {\"p50\": %.2f, \"p10\": %.2f, \"p90\": %.2f}, \"persistent_ring_us\": {\"p50\": %.2f, \"p10\": %.2f, \"p90\": %.2f}, "
          "\"persistent_timed_out\": %s, \"results_identical\": %s, \"mismatches\": %zu}\n",
          S, S, nCalls, med( tLaunch, 0.5 ), med( tLaunch, 0.1 ), med( tLaunch, 0.9 ), med( tPersist, 0.5 ), med( tPersist, 0.1 ), med( tPersist, 0.9 ),
          timeout ? "true" : "false", ( !timeout && bad == 0 ) ? "true" : "false", bad );
  return 0;
}
