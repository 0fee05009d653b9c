// vvenc_hip_recorder.h — work-list recorder of the VVenC <-> MI355X binding (hook bit 131072, --SIMD=HIP:131072).
//
// Records what the encoder's OWN search pushes through the kernel tables of the hot path for selected pictures, so that bench.py / the tests replay REAL work
// lists on the device instead of synthetic ones:
//   * every call through RdCost::m_afpDistortFunc[0][*] (the CPU entry runs, its result is recorded): function, size, subShift, where the operands live
//     (original plane / a reference picture's reconstruction plane at (x, y), or a compact block copied into the sample pool), the distortion returned;
//   * inside InterSearch::xMotionEstimation (EncoderLib/InterSearch.cpp:1976-2130): the integer candidates of the search in call order and every
//     xPatternRefinement stage (:760-880) as a descriptor (base vector, 9 offsets, filter set) with the costs of the positions the encoder evaluated;
//   * every forward transform TrQuant::xT (CommonLib/TrQuant.cpp:481-564): component, position, size, transform types, QP, and the residual block;
//   * every DMVR sub-block search (CommonLib/InterPrediction.cpp:1312-1392) with its result.
// Output: $VVHIP_RECORD_DIR/poc<N>.* (little-endian raw arrays + one JSON index), pictures selected by $VVHIP_RECORD_POCS ("16,24", default: all).
// The record layouts are mirrored by vvenc_amd/recorded.py.  Nothing here touches a device: recording works on any host.
#pragma once
#include <cstdint>

namespace vvrec {

#pragma pack( push, 1 )
struct Operand { int16_t plane; int16_t pad; int32_t x, y; };            // plane >= 0: index into the picture's plane table, block at sample (x, y); plane == -1: pool, x = sample offset, y = stride
struct MeRec    { int32_t cuX, cuY; int16_t w, h; int16_t refPlane; uint8_t bi, list; int32_t patternPool; int32_t firstCand, nCand, firstStage, nStage; };   // patternPool < 0: the original block at (cuX, cuY)
struct CandRec  { int32_t me; int32_t x, y; uint8_t df, subShift, pad0 /* 1: the encoder's own call returned an early-exit partial sum, cost holds the full one */, pad1; uint64_t cost; };   // block position in the reference plane
struct StageRec { int32_t me; int32_t baseX, baseY; int16_t baseHor, baseVer; uint8_t iFrac, hadMode, reduceTap, altHpel; int32_t pad; uint64_t cost[9]; };   // cost ~0ull: position skipped by the encoder
struct DistRec  { uint8_t df, subShift, bitDepth, ctx; int16_t w, h; Operand org, cur; uint64_t cost; int32_t maskPool; };      // calls outside xMotionEstimation; maskPool >= 0 (DF_SAD_WITH_MASK): the weights the call walked, as a compact w x ( h >> subShift ) pool block
struct TuRec    { uint8_t comp, trHor, trVer, flags; int16_t w, h; int16_t qp, bitDepth; int32_t x, y; int32_t pool; };                                        // flags: 1 IRAP, 2 luma, 4 intra CU
struct DmvrRec  { int16_t ref0Plane, ref1Plane; int32_t x0, y0, x1, y1; int16_t frac0x, frac0y, frac1x, frac1y; int16_t dx, dy; int16_t mvdX, mvdY; int32_t pad; uint64_t minCost; };
struct PlaneRec { int32_t kind /* 0 original (as the CTU copies read it), 1 reconstruction of a reference picture */, poc, comp, width, height, stride, margin; int64_t fileOffset; };
#pragma pack( pop )

bool active();                                        // $VVHIP_RECORD_DIR is set
// hook bodies (signatures of VvhipHooks)
void initRdCost( void* rdCost );                      // wraps row 0 of the table with recording trampolines around the CPU entries
void picture( const void* picture );                  // a worker thread starts a CTU task of this picture
void cu( const void* codingStructure );               // EncCu::xCompressCU starts on a block: where its compact original copy maps to
void meBegin( int cuX, int cuY, int w, int h, int list, int refIdx, int refPoc, bool bi, const int16_t* pattern, int patternStride, const int16_t* refY, int refStride );
void meEnd();
void stageBegin( const int16_t* patternRoi, int baseHor, int baseVer, int iFrac, int hadMode, int reduceTap, bool altHpel );
void stageCost( int i, uint64_t dist );
void stageEnd();
void tu( const void* transformUnit, int comp, const int16_t* resi, long stride, int w, int h, int trHor, int trVer, int bitDepth );
void dmvrBegin( const void* cu, const int16_t* ref0, int stride0, int fx0, int fy0, const int16_t* ref1, int stride1, int fx1, int fy1, int cuW, int cuH, int dx, int dy );
void dmvrResult( int num, int mvdX, int mvdY, uint64_t minCost );
void flush();                                         // writes everything recorded so far (encoder close)

}
