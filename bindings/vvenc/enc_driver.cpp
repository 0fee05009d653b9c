// enc_driver.cpp — a plain-C entry point around the public vvenc API (include/vvenc/vvenc.h) of the encoder built WITH the MI355X binding, so that ctypes callers
// (tests/, bench.py, tools/) can encode a clip held in memory and get the bitstream back.  PRODUCT side: no oracle code, nothing but the encoder's own API.
//
//   long vvenc_hip_encode( y, u, v, width, height, frames, inputBitDepth, internalBitDepth, preset, qp, threads, simd, options, out, outCap, &seconds )
//     y / u / v : planar int16 samples, frames x height x width (chroma 4:2:0: height/2 x width/2), tightly packed
//     simd      : NULL / "" (the encoder's default), "SCALAR", "SSE41", "AVX2", ... or "HIP[:mask]" (vvenc_set_SIMD_extension; vvencimpl.cpp:800-851 as patched by apply_binding.py)
//     options   : "name=value;name=value" handed to vvenc_set_param after the preset
//     returns the bitstream size (> 0) or a negative error; *seconds = wall time of the encode loop (first picture in .. last access unit out)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "vvenc/vvenc.h"
#include "vvenc/vvencCfg.h"

extern "C" void vvref_after_simd_init();        // vvenc_hip_binding.cpp: re-points the global g_tCoeffOps when hook bit 32 is set

namespace {
void silent( void*, int, const char*, va_list ) {}

bool applyOptions( vvenc_config& cfg, const char* options )
{
  if( !options ) return true;
  const std::string all( options );
  for( size_t at = 0; at < all.size(); )
  {
    size_t semi = all.find( ';', at );
    if( semi == std::string::npos ) semi = all.size();
    const std::string item = all.substr( at, semi - at );
    at = semi + 1;
    const size_t eq = item.find( '=' );
    if( eq == std::string::npos ) continue;
    if( vvenc_set_param( &cfg, item.substr( 0, eq ).c_str(), item.substr( eq + 1 ).c_str() ) != 0 )
    {
      fprintf( stderr, "vvenc_hip_encode: option '%s' refused\n", item.c_str() );
      return false;
    }
  }
  return true;
}

struct Sink
{
  uint8_t* out; long cap; long used = 0;
  bool take( const vvencAccessUnit& au )
  {
    if( au.payloadUsedSize <= 0 ) return true;
    if( used + au.payloadUsedSize > cap ) return false;
    memcpy( out + used, au.payload, au.payloadUsedSize );
    used += au.payloadUsedSize;
    return true;
  }
};
}

extern "C" __attribute__( ( visibility( "default" ) ) )
long vvenc_hip_encode( const int16_t* y, const int16_t* u, const int16_t* v, int width, int height, int frames, int inputBitDepth, int internalBitDepth,
                       int preset, int qp, int threads, const char* simd, const char* options, uint8_t* out, long outCap, double* seconds )
{
  const bool wantSimd = simd && simd[0];
  if( !vvenc_set_SIMD_extension( wantSimd ? simd : nullptr ) && wantSimd )
  {
    fprintf( stderr, "vvenc_hip_encode: SIMD request '%s' refused\n", simd );
    return -4;
  }
  vvenc_config cfg;
  vvenc_init_default( &cfg, width, height, 30, 0, qp, ( vvencPresetMode ) preset );
  cfg.m_inputBitDepth[0]    = inputBitDepth;
  cfg.m_internalBitDepth[0] = internalBitDepth;
  cfg.m_numThreads          = threads;
  cfg.m_verbosity           = VVENC_SILENT;
  if( !applyOptions( cfg, options ) ) return -5;
  vvenc_set_msg_callback( &cfg, nullptr, silent );

  vvencEncoder* enc = vvenc_encoder_create();
  if( !enc ) return -1;
  if( vvenc_encoder_open( enc, &cfg ) != 0 )
  {
    fprintf( stderr, "vvenc_hip_encode: open failed: %s\n", vvenc_get_last_error( enc ) );
    vvenc_encoder_close( enc );
    return -2;
  }
  vvref_after_simd_init();          // opening the encoder re-ran the SIMD initialisation of the global transform table (vvencimpl.cpp:96)

  vvencYUVBuffer pic;
  vvenc_YUVBuffer_default( &pic );
  vvenc_YUVBuffer_alloc_buffer( &pic, VVENC_CHROMA_420, width, height );
  vvencAccessUnit au;
  vvenc_accessUnit_default( &au );
  vvenc_accessUnit_alloc_payload( &au, ( 3 * width * height ) / 2 + ( 64 << 10 ) );

  Sink sink{ out, outCap };
  const int16_t* src[3] = { y, u, v };
  const int pw[3] = { width, width / 2, width / 2 }, ph[3] = { height, height / 2, height / 2 };
  int  rc   = 0;
  bool done = false;
  const auto start = std::chrono::steady_clock::now();
  for( int f = 0; f < frames && !rc; f++ )
  {
    for( int c = 0; c < 3; c++ )
    {
      const int16_t* from = src[c] + ( size_t ) f * pw[c] * ph[c];
      for( int r = 0; r < ph[c]; r++ ) memcpy( pic.planes[c].ptr + ( size_t ) r * pic.planes[c].stride, from + ( size_t ) r * pw[c], sizeof( int16_t ) * pw[c] );
    }
    pic.sequenceNumber = f;
    pic.cts            = f;
    pic.ctsValid       = true;
    rc = vvenc_encode( enc, &pic, &au, &done );
    if( !rc && !sink.take( au ) ) rc = -100;
  }
  while( !rc && !done )
  {
    rc = vvenc_encode( enc, nullptr, &au, &done );
    if( !rc && !sink.take( au ) ) rc = -100;
  }
  if( seconds ) *seconds = std::chrono::duration<double>( std::chrono::steady_clock::now() - start ).count();
  if( rc ) fprintf( stderr, "vvenc_hip_encode: error %d: %s\n", rc, vvenc_get_last_error( enc ) );
  vvenc_YUVBuffer_free_buffer( &pic );
  vvenc_accessUnit_free_payload( &au );
  vvenc_encoder_close( enc );
  return rc ? -3 : sink.used;
}
