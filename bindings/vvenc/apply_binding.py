#!/usr/bin/env python3
"""The VVenC side of the binding as an executable patch (PRODUCT: what a VVenC maintainer applies; INTEGRATION.md section 2 walks through it).

Writes hook-enabled COPIES of eleven reference translation units into an output directory (build output, git-ignored): the script inserts one-line calls to
g_vvhipHooks (bindings/vvenc/vvenc_hip_binding.h) at anchor lines it looks up in the files where they lie under the reference tree — nothing of the reference is
stored in this repository.  With every hook null the patched encoder is the unpatched one (bitstream tests).  Run-time selection: --SIMD=HIP[:mask]
(VVEncImpl::setSIMDExtension, source/Lib/vvenc/vvencimpl.cpp:800-851).

usage: apply_binding.py <reference source/Lib dir> <output dir>"""
import os
import sys

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def patch(name, edits, sub="CommonLib"):
    s = open(os.path.join(src, sub, name)).read()
    for kind, anchor, text in edits:
        assert s.count(anchor) == 1, (name, anchor, s.count(anchor))
        if kind == "before":
            s = s.replace(anchor, text + anchor)
        elif kind == "after":
            s = s.replace(anchor, anchor + text)
        else:
            s = s.replace(anchor, text)
    open(os.path.join(out, name), "w").write(s)


INC = '\n#include "vvenc_hip_binding.h"\n'
patch("RdCost.cpp", [
    ("after", '#include "RdCost.h"', INC),
    ("before", "  m_costMode      = VVENC_COST_STANDARD_LOSSY;",
     "  if( enableOpt && g_vvhipHooks.initRdCost ) g_vvhipHooks.initRdCost( this );\n"),
])
patch("Quant.cpp", [
    ("after", '#include "Quant.h"', INC),
    ("after", "  initQuantX86();\n#endif\n", "  if( g_vvhipHooks.initQuant ) g_vvhipHooks.initQuant( this );\n"),
])
patch("TrQuant.cpp", [
    ("after", '#include "TrQuant.h"', INC),
    ("before", "  TCoeff* block = m_blk;",
     "  if( g_vvhipHooks.recTu && width > 1 && height > 1 ) g_vvhipHooks.recTu( &tu, ( int ) compID, resi.buf, ( long ) resi.stride, width, height, trTypeHor, trTypeVer, bitDepth );\n"
     "  if( g_vvhipHooks.tuLookup && width > 2 && height > 2 && !tu.cu->lfnstIdx && trTypeHor == DCT2 && trTypeVer == DCT2 &&\n"
     "      g_vvhipHooks.tuLookup( resi.buf, resi.stride, dstCoeff.buf, width, height ) ) return;\n"
     "  if( g_vvhipHooks.fwd2D && width > 1 && height > 1 && !tu.cu->lfnstIdx &&\n"
     "      g_vvhipHooks.fwd2D( resi.buf, resi.stride, dstCoeff.buf, width, height, trTypeHor, trTypeVer, bitDepth ) ) return;\n"),
    ("before", "  TCoeff *block = m_blk;",
     "  if( g_vvhipHooks.inv2D && width > 1 && height > 1 && !( tu.cs->sps->LFNST && tu.cu->lfnstIdx ) &&\n"
     "      g_vvhipHooks.inv2D( pCoeff.buf, pResidual.buf, pResidual.stride, width, height, trTypeHor, trTypeVer, bitDepth ) ) return;\n"),
])
patch("MCTF.cpp", [
    ("after", '#include "MCTF.h"', INC),
    ("after", "    initMCTF_ARM();\n#endif\n  }\n", "  if( enableOpt && g_vvhipHooks.initMCTF ) g_vvhipHooks.initMCTF( this );\n"),
    ("before", "    Array2D<MotionVector> mv_0(width / (m_mctfUnitSize * 8) + 1, height / (m_mctfUnitSize * 8) + 1);",
     "    if( !( g_vvhipHooks.mctfMe && g_vvhipHooks.mctfMe( this, srcPic.picBuffer, origBuf, srcPic.mvs, addLevel, curPic->poc, m_filterPoc ) ) )\n    {\n"),
    ("after", "    motionEstimationLuma(srcPic.mvs, origBuf, srcPic.picBuffer, m_mctfUnitSize, &mv_2, 1, true);\n", "    }\n"),
    ("replace", "    subsampleLuma( origBuf,         origSubsampled2 );\n    subsampleLuma( origSubsampled2, origSubsampled4 );\n    if (condAddLevel)\n    {\n      subsampleLuma(origSubsampled4, origSubsampled8);\n    }\n",
     "    const bool hipMe = g_vvhipHooks.mctfMe && g_vvhipHooks.mctfWants && g_vvhipHooks.mctfWants( this, m_area.width, m_area.height );\n"
     "    if( !hipMe )\n    {\n"
     "    subsampleLuma( origBuf,         origSubsampled2 );\n    subsampleLuma( origSubsampled2, origSubsampled4 );\n    if (condAddLevel)\n    {\n      subsampleLuma(origSubsampled4, origSubsampled8);\n    }\n"
     "    }\n"),
    ("before", "    for ( int i = dropFramesFront; i < picFifo.size() - dropFramesBack; i++ )\n    {\n      Picture* curPic = picFifo[ i ];\n      if ( curPic->poc == m_filterPoc )",
     "    if( hipMe && g_vvhipHooks.mctfPrefetch ) g_vvhipHooks.mctfPrefetch( this, &picFifo, dropFramesFront, dropFramesBack, origBuf, condAddLevel, m_filterPoc );\n"),
    ("before", "  const double lumaSigmaSq = m_sigmaMultiplier * ( 128.0 + 3.0 / 256.0 * m_encCfg->m_QP * m_encCfg->m_QP * m_encCfg->m_QP );",
     "  if( g_vvhipHooks.mctfApply && g_vvhipHooks.mctfApply( this, orgPic, &srcFrameInfo, newOrgPic, overallStrength ) ) return;\n"),
])
patch("InterpolationFilter.cpp", [
    ("after", '#include "InterpolationFilter.h"', INC),
    ("before", "void InterpolationFilter::initInterpolationFilter( bool enable )\n{\n",
     ""),   # (anchor check only: the hook goes at the end of the function body)
    ("after", "    initInterpolationFilterARM();\n#endif\n  }\n#endif\n", "  if( enable && g_vvhipHooks.initIF ) g_vvhipHooks.initIF( this );\n"),
])
# DMVR: the refinement search of every sub-block of the CU in one device call; the reference's loop then only copies results
patch("InterPrediction.cpp", [
    ("after", '#include "InterPrediction.h"', INC),
    ("before", "    DistParam distParam = m_pcRdCost->setDistParam( nullptr, nullptr, bilinearBufStride, bilinearBufStride, bd, COMP_Y, dx, dy, 1, true );",
     "    int16_t  hipMvd[2 * MAX_NUM_SUBCU_DMVR]; uint64_t hipCost[MAX_NUM_SUBCU_DMVR];\n"
     "    bool hipDmvr = false;\n"
     "    if( g_vvhipHooks.dmvrSearch || g_vvhipHooks.recDmvrBegin )\n"
     "    {\n"
     "      const Picture* r0 = cu.slice->getRefPic( L0, cu.refIdx[L0] ); const Picture* r1 = cu.slice->getRefPic( L1, cu.refIdx[L1] );\n"
     "      const int s0 = r0->getRecoBufStride( COMP_Y ), s1 = r1->getRecoBufStride( COMP_Y );\n"
     "      const Pel* p0 = r0->getRecoBufPtr( COMP_Y ) + ( puPos.x + ( mergeMVL0.hor >> MV_FRACTIONAL_BITS_INTERNAL ) ) + ( puPos.y + ( mergeMVL0.ver >> MV_FRACTIONAL_BITS_INTERNAL ) ) * s0;\n"
     "      const Pel* p1 = r1->getRecoBufPtr( COMP_Y ) + ( puPos.x + ( mergeMVL1.hor >> MV_FRACTIONAL_BITS_INTERNAL ) ) + ( puPos.y + ( mergeMVL1.ver >> MV_FRACTIONAL_BITS_INTERNAL ) ) * s1;\n"
     "      if( g_vvhipHooks.recDmvrBegin ) g_vvhipHooks.recDmvrBegin( &cu, p0, s0, mergeMVL0.hor & 15, mergeMVL0.ver & 15, p1, s1, mergeMVL1.hor & 15, mergeMVL1.ver & 15, cu.lwidth(), cu.lheight(), dx, dy );\n"
     "      hipDmvr = g_vvhipHooks.dmvrSearch && g_vvhipHooks.dmvrSearch( p0, s0, mergeMVL0.hor & 15, mergeMVL0.ver & 15, p1, s1, mergeMVL1.hor & 15, mergeMVL1.ver & 15,\n"
     "                                         cu.lwidth(), cu.lheight(), dx, dy, bd, hipMvd, hipCost );\n"
     "    }\n"),
    ("before", "        distParam.org.buf = addrL0;\n        distParam.cur.buf = addrL1;\n        minCost  = distParam.distFunc( distParam ) >> 1;",
     "        if( hipDmvr ) { cu.mvdL0SubPu[num] = Mv( hipMvd[2 * num], hipMvd[2 * num + 1] ); minCost = hipCost[num]; } else {\n"),
    ("before", "        bioAppliedType[num] = ( minCost < bioEnabledThres ) ? false : bioApplied;",
     "        }\n        if( g_vvhipHooks.recDmvrResult ) g_vvhipHooks.recDmvrResult( num, cu.mvdL0SubPu[num].hor, cu.mvdL0SubPu[num].ver, minCost );\n"),
])

# batched call site: all sub-pel positions of one xPatternRefinement stage are scored by ONE device call up
# front; the reference's own loop (skip rules, break rules, MV-bit costs, strict < update, patternId bookkeeping) then replays them
patch("InterSearch.cpp", [
    ("after", '#include "InterSearch.h"', INC),
    # integer TZ search: a diamond round's positions are scored by one device call when the round starts, xTZSearchHelp looks them up
    ("replace", "  m_cDistParam.cur.buf = piRefSrch;\n\n  uiSad = m_cDistParam.distFunc( m_cDistParam );",
     "  m_cDistParam.cur.buf = piRefSrch;\n\n  if( !( g_vvhipHooks.tzLookup && g_vvhipHooks.tzLookup( &m_cDistParam, rcStruct.piRefY, iSearchX, iSearchY, &uiSad ) ) )\n    uiSad = m_cDistParam.distFunc( m_cDistParam );"),
    ("replace", "  rcStruct.uiBestRound += 1;\n\n  if ( iDist == 1 )",
     "  rcStruct.uiBestRound += 1;\n  if( g_vvhipHooks.tzPrefetch ) g_vvhipHooks.tzPrefetch( &m_cDistParam, rcStruct.piRefY, rcStruct.iRefStride, iStartX, iStartY, iDist, bCheckCornersAtDist1, sr.left, sr.right, sr.top, sr.bottom );\n\n  if ( iDist == 1 )"),
    ("before", "  if( cu.cs->picture->useME )\n  {\n    switch ( m_motionEstimationSearchMethodSCC )", "  if( g_vvhipHooks.tzReset ) g_vvhipHooks.tzReset();\n"),
    ("replace", "  cStruct.uiBestSad     = MAX_DISTORTION;\n\n\n  CodedCUInfo &relatedCU = m_modeCtrl->getBlkInfo( cu );",
     "  cStruct.uiBestSad     = MAX_DISTORTION;\n"
     "  if( g_vvhipHooks.recMeBegin ) g_vvhipHooks.recMeBegin( cu.lx(), cu.ly(), cu.lwidth(), cu.lheight(), ( int ) refPicList, iRefIdxPred, refPic->poc, bBi, pcPatternKey->buf, ( int ) pcPatternKey->stride, buf.buf, ( int ) buf.stride );\n\n"
     "  CodedCUInfo &relatedCU = m_modeCtrl->getBlkInfo( cu );"),
    ("before", "  DTRACE(g_trace_ctx, D_ME, \"   MECost<L%d,%d>: %6d (%d)  MV:%d,%d\\n\"", "  if( g_vvhipHooks.recMeEnd ) g_vvhipHooks.recMeEnd();\n"),
    ("before", "  rcMvFrac = pcMvRefine[uiDirecBest];", "  if( g_vvhipHooks.recStageEnd ) g_vvhipHooks.recStageEnd();\n"),
    # residual loop (xEstimateInterResidualQT): the forward transforms of all component TUs of the CU in ONE device round trip before the component loop; TrQuant::xT then finds
    # its coefficients ready (tuLookup in front of the CPU cores).  need-RDOQ / RDOQ / rate estimation stay where they are.
    ("before", "    for( uint32_t c = 0; c < numTBlocks; c++ )\n    {\n      const ComponentID compID    = ComponentID(c);\n      const CompArea&   compArea  = tu.blocks[compID];\n      const int channelBitDepth   = sps.bitDepths[toChannelType(compID)];",
     "    if( g_vvhipHooks.tuPrefetch && !cu.lfnstIdx && !tu.noResidual )\n"
     "    {\n"
     "      const Pel* hR[3]; int hS[3], hW[3], hH[3], hN = 0;\n"
     "      for( uint32_t c = 0; c < numTBlocks; c++ ) if( tu.blocks[c].valid() && tu.blocks[c].width > 2 && tu.blocks[c].height > 2 )\n"
     "      { const CPelBuf b = orgResiBuf.get( ComponentID( c ) ); hR[hN] = b.buf; hS[hN] = ( int ) b.stride; hW[hN] = b.width; hH[hN] = b.height; hN++; }\n"
     "      if( hN ) g_vvhipHooks.tuPrefetch( hR, hS, hW, hH, hN, sps.bitDepths[CH_L] );\n"
     "    }\n"),
    ("after", "  const Mv* pcMvRefine = (iFrac == 2 ? s_acMvRefineH : s_acMvRefineQ);\n",
     "  if( g_vvhipHooks.recStageBegin ) g_vvhipHooks.recStageBegin( pattern->buf, baseRefMv.hor, baseRefMv.ver, iFrac, m_pcEncCfg->m_bUseHADME ? ( m_pcEncCfg->m_fastHad ? 2 : 1 ) : 0, reduceTap, useAltHpelIf );\n"
     "  uint64_t hipCost[9];\n"
     "  const bool hipOk = g_vvhipHooks.patternCosts && g_vvhipHooks.patternCosts( pcPatternKey, pattern, baseRefMv.hor, baseRefMv.ver, iFrac, pcMvRefine, clpRng.bd,\n"
     "                         m_pcEncCfg->m_bUseHADME ? ( m_pcEncCfg->m_fastHad ? 2 : 1 ) : 0, reduceTap, useAltHpelIf, hipCost );\n"),
    ("replace", "    m_cDistParam.cur.buf   = piRefPos;\n    uiDist = m_cDistParam.distFunc( m_cDistParam );\n",
     "    m_cDistParam.cur.buf   = piRefPos;\n    uiDist = hipOk ? hipCost[i] : m_cDistParam.distFunc( m_cDistParam );\n"
     "    if( g_vvhipHooks.recStageCost ) g_vvhipHooks.recStageCost( ( int ) i, uiDist );\n"),
], sub="EncoderLib")

# ALF statistics of a CTU: classification + covariance records from ONE hook call; the reference's own accumulators are
# filled from the returned records (they start at zero: one CTU per statistics unit).  Virtual-boundary CTUs and non-linear ALF stay on the CPU.
patch("EncAdaptiveLoopFilter.cpp", [
    ("after", '#include "EncAdaptiveLoopFilter.h"', INC),
    ("before", "    Area blk( xPos, yPos, width, height );\n    deriveClassification(",
     "    bool hipAlf = false;\n"
     "    if( g_vvhipHooks.alfCtu && !m_encCfg->m_useNonLinearAlfLuma && !m_encCfg->m_useNonLinearAlfChroma && numberOfComponents == 3 )\n"
     "    {\n"
     "      static thread_local std::vector<float> hipStats( 3 * MAX_NUM_ALF_CLASSES * 183 );\n"
     "      static thread_local std::vector<uint8_t> hipCls( 32 * 32 * 2 );\n"
     "      const UnitArea hArea( m_chromaFormat, Area( xPos, yPos, width, height ) );\n"
     "      const Pel* hRec[3]; const Pel* hOrg[3]; int hRs[3], hOs[3]; bool hEn[3]; float* hSt[3];\n"
     "      for( int c = 0; c < 3; c++ )\n"
     "      {\n"
     "        const CompArea& ca = hArea.block( ComponentID( c ) );\n"
     "        hRec[c] = recYuv.get( ComponentID( c ) ).bufAt( ca ); hRs[c] = recYuv.get( ComponentID( c ) ).stride;\n"
     "        hOrg[c] = orgYuv.get( ComponentID( c ) ).bufAt( ca ); hOs[c] = orgYuv.get( ComponentID( c ) ).stride;\n"
     "        hEn[c] = m_alfFilterStatEnabled[c]; hSt[c] = hipStats.data() + ( size_t ) c * MAX_NUM_ALF_CLASSES * 183;\n"
     "        // the records the float chains continue from: a statistics unit may span several CTUs (getStatisticsASU)\n"
     "        const int nCls = c ? 1 : MAX_NUM_ALF_CLASSES, nCo = m_filterShapes[toChannelType( ComponentID( c ) )].numCoeff;\n"
     "        if( hEn[c] ) for( int k = 0; k < nCls; k++ )\n"
     "        {\n"
     "          const AlfCovariance& cov = m_alfCovariance[c][asuRsAddr][k]; float* r = hSt[c] + ( size_t ) k * 183;\n"
     "          memset( r, 0, 183 * sizeof( float ) );\n"
     "          for( int a = 0; a < nCo; a++ ) { for( int b = 0; b < nCo; b++ ) r[a * 13 + b] = cov.E[0][0][a < b ? a : b][a < b ? b : a]; r[169 + a] = cov.y[0][a]; }\n"
     "          r[182] = cov.pixAcc;\n"
     "        }\n"
     "      }\n"
     "      hipAlf = g_vvhipHooks.alfCtu( hRec, hRs, hOrg, hOs, width, height, getComponentScaleX( COMP_Cb, m_chromaFormat ), m_inputBitDepth[CH_L],\n"
     "                                    m_alfVBLumaCTUHeight, m_alfVBLumaPos, m_alfVBChmaCTUHeight, m_alfVBChmaPos, hEn, hipCls.data(), hSt );\n"
     "      if( hipAlf )\n"
     "      {\n"
     "        AlfClassifier* cl = &m_classifier[numClassBlocksInCTU * ctuRsAddr];\n"
     "        bool used[MAX_NUM_ALF_CLASSES] = { false };\n"
     "        for( int i = 0; i < height; i += 4 ) for( int j = 0; j < width; j += 4 )\n"
     "        {\n"
     "          const uint8_t* c = hipCls.data() + 2 * ( ( i / 4 ) * ( width / 4 ) + j / 4 );\n"
     "          cl[( i / 4 ) * ( MAX_CU_SIZE / 4 ) + j / 4] = AlfClassifier( c[0], c[1] ); used[c[0]] = true;\n"
     "        }\n"
     "        for( int c = 0; c < 3; c++ )\n"
     "        {\n"
     "          if( !hEn[c] ) continue;\n"
     "          const int nCls = c ? 1 : MAX_NUM_ALF_CLASSES, nCo = m_filterShapes[toChannelType( ComponentID( c ) )].numCoeff;\n"
     "          for( int k = 0; k < nCls; k++ )\n"
     "          {\n"
     "            if( c == 0 && !used[k] ) continue;\n"
     "            AlfCovariance& cov = m_alfCovariance[c][asuRsAddr][k]; const float* r = hSt[c] + ( size_t ) k * 183;\n"
     "            for( int a = 0; a < nCo; a++ ) { for( int b = 0; b < nCo; b++ ) cov.E[0][0][a][b] = r[a * 13 + b]; cov.y[0][a] = r[169 + a]; }\n"
     "            cov.pixAcc = r[182]; cov.all0 = false;\n"
     "          }\n"
     "        }\n"
     "      }\n"
     "    }\n"
     "    if( !hipAlf )\n    {\n"),
    ("before", "  }  \n}\n\nvoid EncAdaptiveLoopFilter::copyCTUforALF(", "    }\n"),
    # whole-picture ALF statistics: the per-CTU tasks do nothing, the first thing deriveFilter does is ONE device call for the picture
    ("before", "  const PreCalcValues& pcv = *cs.pcv;\n  const int xC = ( ctuRsAddr % pcv.widthInCtus ) << pcv.maxCUSizeLog2;",
     "  if( ( g_vvhipHooks.alfPicture && g_vvhipHooks.alfPictureOn( ( int ) m_numCTUsInPic, m_encCfg->m_numThreads ) && m_encCfg->m_ifpLines == 0 && !m_accumStatCTUWise && !m_encCfg->m_useNonLinearAlfLuma && !m_encCfg->m_useNonLinearAlfChroma && m_chromaFormat == CHROMA_420 && cs.pps->getNumTiles() == 1 && cs.pps->numSlicesInPic == 1 && !cs.picHeader->virtualBoundariesEnabled ) )\n"
     "  {\n"
     "    // the row task's last CTU: every sample the statistics of this CTU row read is final (EncSlice.cpp:1135-1141) -> the row's statistics unit goes to the device now, asynchronously\n"
     "    if( g_vvhipHooks.alfRow && ( ctuRsAddr % cs.pcv->widthInCtus ) == cs.pcv->widthInCtus - 1 )\n"
     "    {\n"
     "      const Pel* hRec[3]; const Pel* hOrg[3]; int hRs[3], hOs[3]; bool hEn[3];\n"
     "      PelUnitBuf hOrgYuv = pic.getOrigBuf();\n"
     "      for( int c = 0; c < 3; c++ )\n"
     "      {\n"
     "        hRec[c] = m_tempBuf.get( ComponentID( c ) ).buf; hRs[c] = m_tempBuf.get( ComponentID( c ) ).stride;\n"
     "        hOrg[c] = hOrgYuv.get( ComponentID( c ) ).buf; hOs[c] = hOrgYuv.get( ComponentID( c ) ).stride; hEn[c] = m_alfFilterStatEnabled[c];\n"
     "      }\n"
     "      g_vvhipHooks.alfRow( this, cs.picture->poc, ctuRsAddr / cs.pcv->widthInCtus, hRec, hRs, hOrg, hOs, m_picWidth, m_picHeight, m_inputBitDepth[CH_L], m_maxCUHeight, m_maxAsuHeight,\n"
     "                           m_alfVBLumaCTUHeight, m_alfVBLumaPos, m_alfVBChmaCTUHeight, m_alfVBChmaPos, hEn );\n"
     "    }\n"
     "    return;\n"
     "  }\n"),
    ("after", "  initCABACEstimator( cs.slice );\n\n  // Accumulate ALF statistic\n",
     "  if( g_vvhipHooks.alfBeginPicture ) g_vvhipHooks.alfBeginPicture( this, cs.picture->poc );\n"
     "  if( ( g_vvhipHooks.alfPicture && g_vvhipHooks.alfPictureOn( ( int ) m_numCTUsInPic, m_encCfg->m_numThreads ) && m_encCfg->m_ifpLines == 0 && !m_accumStatCTUWise && !m_encCfg->m_useNonLinearAlfLuma && !m_encCfg->m_useNonLinearAlfChroma && m_chromaFormat == CHROMA_420 && cs.pps->getNumTiles() == 1 && cs.pps->numSlicesInPic == 1 && !cs.picHeader->virtualBoundariesEnabled ) && numCtus == ( int ) m_numCTUsInPic )\n"
     "  {\n"
     "    static thread_local std::vector<uint8_t> hCls; static thread_local std::vector<float> hSt[3];\n"
     "    hCls.resize( ( size_t ) ( m_picWidth / 4 ) * ( m_picHeight / 4 ) * 2 );\n"
     "    const Pel* hRec[3]; const Pel* hOrg[3]; int hRs[3], hOs[3]; bool hEn[3]; float* hP[3];\n"
     "    PelUnitBuf hOrgYuv = pic.getOrigBuf();\n"
     "    for( int c = 0; c < 3; c++ )\n"
     "    {\n"
     "      hRec[c] = m_tempBuf.get( ComponentID( c ) ).buf; hRs[c] = m_tempBuf.get( ComponentID( c ) ).stride;\n"
     "      hOrg[c] = hOrgYuv.get( ComponentID( c ) ).buf; hOs[c] = hOrgYuv.get( ComponentID( c ) ).stride;\n"
     "      hEn[c] = m_alfFilterStatEnabled[c]; hSt[c].resize( ( size_t ) m_numAsusInPic * ( c ? 1 : MAX_NUM_ALF_CLASSES ) * 183 ); hP[c] = hSt[c].data();\n"
     "    }\n"
     "    const bool hOk = g_vvhipHooks.alfPicture( this, cs.picture->poc, hRec, hRs, hOrg, hOs, m_picWidth, m_picHeight, m_inputBitDepth[CH_L], m_maxCUHeight, m_maxAsuHeight,\n"
     "                                              m_alfVBLumaCTUHeight, m_alfVBLumaPos, m_alfVBChmaCTUHeight, m_alfVBChmaPos, hEn, hCls.data(), hP );\n"
     "    CHECK( !hOk, \"HIP ALF picture statistics failed\" );\n"
     "    const int hBw = m_picWidth / 4, hCtuBlk = ( MAX_CU_SIZE * MAX_CU_SIZE ) >> 4;\n"
     "    for( int ctu = 0; ctu < ( int ) m_numCTUsInPic; ctu++ )\n"
     "    {\n"
     "      const int x0 = ( ctu % m_numCTUsInWidth ) * m_maxCUWidth, y0 = ( ctu / m_numCTUsInWidth ) * m_maxCUHeight;\n"
     "      const int w = std::min( m_maxCUWidth, m_picWidth - x0 ), h = std::min( m_maxCUHeight, m_picHeight - y0 );\n"
     "      for( int i = 0; i < h; i += 4 ) for( int j = 0; j < w; j += 4 )\n"
     "      {\n"
     "        const uint8_t* c = hCls.data() + 2 * ( ( size_t ) ( ( y0 + i ) / 4 ) * hBw + ( x0 + j ) / 4 );\n"
     "        m_classifier[hCtuBlk * ctu + ( i / 4 ) * ( MAX_CU_SIZE / 4 ) + j / 4] = AlfClassifier( c[0], c[1] );\n"
     "      }\n"
     "    }\n"
     "    for( int a = 0; a < m_numAsusInPic; a++ )\n"
     "    {\n"
     "      const int x0 = ( a % m_numAsusInWidth ) * m_maxAsuWidth, y0 = ( a / m_numAsusInWidth ) * m_maxAsuHeight;\n"
     "      const int w = std::min( m_maxAsuWidth, m_picWidth - x0 ), h = std::min( m_maxAsuHeight, m_picHeight - y0 );\n"
     "      bool used[MAX_NUM_ALF_CLASSES] = { false };\n"
     "      for( int i = 0; i < h; i += 4 ) for( int j = 0; j < w; j += 4 ) used[hCls[2 * ( ( size_t ) ( ( y0 + i ) / 4 ) * hBw + ( x0 + j ) / 4 )]] = true;\n"
     "      for( int c = 0; c < 3; c++ )\n"
     "      {\n"
     "        if( !hEn[c] ) continue;\n"
     "        const int nCls = c ? 1 : MAX_NUM_ALF_CLASSES, nCo = m_filterShapes[toChannelType( ComponentID( c ) )].numCoeff;\n"
     "        for( int k = 0; k < nCls; k++ )\n"
     "        {\n"
     "          AlfCovariance& cov = m_alfCovariance[c][a][k]; const float* r = hP[c] + ( ( size_t ) a * nCls + k ) * 183;\n"
     "          cov.reset();\n"
     "          if( c == 0 && !used[k] ) continue;\n"
     "          for( int p = 0; p < nCo; p++ ) { for( int q = 0; q < nCo; q++ ) cov.E[0][0][p][q] = r[p * 13 + q]; cov.y[0][p] = r[169 + p]; }\n"
     "          cov.pixAcc = r[182]; cov.all0 = false;\n"
     "        }\n"
     "      }\n"
     "    }\n"
     "  }\n"),
    # CC-ALF statistics of a CTU and chroma component (one CTU per covariance, reset before): the record comes from the device
    ("replace", "    getBlkStatsCcAlf( m_alfCovarianceCcAlf[compIdx - 1][filterIdx][ctuRsAddr], m_filterShapesCcAlf[compIdx - 1], orgYuv, recYuv, area, area, compID, yPos );\n  }\n}",
     "    bool hipCc = false;\n"
     "    if( g_vvhipHooks.ccAlfCtu && m_chromaFormat == CHROMA_420 )\n"
     "    {\n"
     "      float r[183];\n"
     "      const CompArea& ca = area.block( compID );\n"
     "      // CTU-local call: the last CTU row has no virtual boundary (:6079-6082) -> a boundary position no row reaches\n"
     "      const int hipVbPos = ( yPos + m_maxCUHeight ) >= m_picHeight ? 1 << 20 : m_alfVBLumaPos; const int picHFromHere = 1 << 30;\n"
     "      hipCc = g_vvhipHooks.ccAlfCtu( orgYuv.get( compID ).bufAt( ca ), orgYuv.get( compID ).stride, recYuv.get( compID ).bufAt( area.chromaPos() ), recYuv.get( compID ).stride,\n"
     "                                     recYuv.get( COMP_Y ).bufAt( area.lumaPos() ), recYuv.get( COMP_Y ).stride, ca.width, ca.height, m_alfVBLumaCTUHeight, hipVbPos, picHFromHere, r );\n"
     "      if( hipCc )\n"
     "      {\n"
     "        AlfCovariance& cov = m_alfCovarianceCcAlf[compIdx - 1][filterIdx][ctuRsAddr];\n"
     "        for( int a = 0; a < 7; a++ ) { for( int b = 0; b < 7; b++ ) cov.E[0][0][a][b] += r[a * 13 + b]; cov.y[0][a] += r[169 + a]; }\n"
     "        cov.pixAcc += r[182];\n"
     "      }\n"
     "    }\n"
     "    if( !hipCc )\n"
     "    getBlkStatsCcAlf( m_alfCovarianceCcAlf[compIdx - 1][filterIdx][ctuRsAddr], m_filterShapesCcAlf[compIdx - 1], orgYuv, recYuv, area, area, compID, yPos );\n  }\n}"),
    # ALF filtering of a CTU (reconstructCTU, the branch without virtual picture boundaries): the table entries' work goes to the device block by block
    ("replace", "        m_filter7x7Blk[nonLinAlfLuma]( &m_classifier[numClassBlocksInCTU * ctuRsAddr], recBuf, recExtBufCTU, blk, blkSrc, COMP_Y, coeff, clip, clpRngs[COMP_Y], cs, m_alfVBLumaCTUHeight, m_alfVBLumaPos );",
     "        if( !( g_vvhipHooks.alfFilterBlk && g_vvhipHooks.alfFilterBlk( &m_classifier[numClassBlocksInCTU * ctuRsAddr], recBuf.get( COMP_Y ).bufAt( blk.x, blk.y ), recBuf.get( COMP_Y ).stride,\n"
     "                 recExtBufCTU.get( COMP_Y ).buf, recExtBufCTU.get( COMP_Y ).stride, width, height, 7, coeff, nonLinAlfLuma ? clip : nullptr, clpRngs[COMP_Y].bd, m_alfVBLumaCTUHeight, m_alfVBLumaPos ) ) )\n"
     "        m_filter7x7Blk[nonLinAlfLuma]( &m_classifier[numClassBlocksInCTU * ctuRsAddr], recBuf, recExtBufCTU, blk, blkSrc, COMP_Y, coeff, clip, clpRngs[COMP_Y], cs, m_alfVBLumaCTUHeight, m_alfVBLumaPos );"),
    ("replace", "          m_filter5x5Blk[nonLinAlfChroma]( m_classifier, recBuf, recExtBufCTU, blk, blkSrc, compID, m_chromaCoeffFinal[alt_num], m_chromaClippFinal[alt_num], clpRngs[compIdx], cs, m_alfVBChmaCTUHeight, m_alfVBChmaPos );",
     "          if( !( g_vvhipHooks.alfFilterBlk && g_vvhipHooks.alfFilterBlk( nullptr, recBuf.get( compID ).bufAt( blk.x, blk.y ), recBuf.get( compID ).stride, recExtBufCTU.get( compID ).buf, recExtBufCTU.get( compID ).stride,\n"
     "                   blk.width, blk.height, 5, m_chromaCoeffFinal[alt_num], nonLinAlfChroma ? m_chromaClippFinal[alt_num] : nullptr, clpRngs[compIdx].bd, m_alfVBChmaCTUHeight, m_alfVBChmaPos ) ) )\n"
     "          m_filter5x5Blk[nonLinAlfChroma]( m_classifier, recBuf, recExtBufCTU, blk, blkSrc, compID, m_chromaCoeffFinal[alt_num], m_chromaClippFinal[alt_num], clpRngs[compIdx], cs, m_alfVBChmaCTUHeight, m_alfVBChmaPos );"),
    ("replace", "        m_filterCcAlf( dstBuf, recYuvExt, blkDst, blkSrc, compID, filterCoeff, clpRngs, cs, m_alfVBLumaCTUHeight, m_alfVBLumaPos );",
     "        if( !( g_vvhipHooks.ccAlfFilterBlk && m_chromaFormat == CHROMA_420 && g_vvhipHooks.ccAlfFilterBlk( const_cast<Pel*>( dstBuf.bufAt( blkDst.x, blkDst.y ) ), dstBuf.stride, recYuvExt.get( COMP_Y ).bufAt( blkSrc.x, blkSrc.y ),\n"
     "                 recYuvExt.get( COMP_Y ).stride, blkDst.width, blkDst.height, filterCoeff, clpRngs[compID].bd, m_alfVBLumaCTUHeight, m_alfVBLumaPos ) ) )\n"
     "        m_filterCcAlf( dstBuf, recYuvExt, blkDst, blkSrc, compID, filterCoeff, clpRngs, cs, m_alfVBLumaCTUHeight, m_alfVBLumaPos );"),
    # ALF reconstruction of the whole picture in one shim call: the first reconstruction task of a picture filters every enabled CTU of the three planes, the other tasks
    # find the picture done (they wait on the hook's lock while it is in progress) and return
    ("before", "  // copy unfiltered reco (including padded / extented area)\n",
     "  if( g_vvhipHooks.alfFilterPicture && m_chromaFormat == CHROMA_420 && cs.pps->getNumTiles() == 1 && cs.pps->numSlicesInPic == 1 && !cs.picHeader->virtualBoundariesEnabled && m_encCfg->m_ifpLines == 0 )\n"
     "  {\n"
     "    if( g_vvhipHooks.alfFilterPicture( this, cs.picture->poc, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, 0, 0, true ) ) return;\n"
     "    const int hNumCtus = ( int ) m_numCTUsInPic, hBw = m_picWidth / 4, hCtuBlk = ( MAX_CU_SIZE * MAX_CU_SIZE ) >> 4;\n"
     "    std::vector<uint8_t> hCls( ( size_t ) hBw * ( m_picHeight / 4 ) * 2 );\n"
     "    std::vector<short> hSet[3];\n"
     "    const short* hFilterIdx = cs.picture->m_alfCtbFilterIndex.data();\n"
     "    bool hAny[3] = { false, false, false };\n"
     "    for( int c = 0; c < 3; c++ )\n"
     "    {\n"
     "      hSet[c].resize( hNumCtus );\n"
     "      for( int ctu = 0; ctu < hNumCtus; ctu++ )\n"
     "      {\n"
     "        hSet[c][ctu] = !m_ctuEnableFlag[c][ctu] ? -1 : c == 0 ? hFilterIdx[ctu] : m_ctuAlternative[c][ctu];\n"
     "        hAny[c] |= hSet[c][ctu] >= 0;\n"
     "      }\n"
     "    }\n"
     "    for( int ctu = 0; ctu < hNumCtus; ctu++ )\n"
     "    {\n"
     "      const int x0 = ( ctu % m_numCTUsInWidth ) * m_maxCUWidth, y0 = ( ctu / m_numCTUsInWidth ) * m_maxCUHeight;\n"
     "      const int w = std::min( m_maxCUWidth, m_picWidth - x0 ), h = std::min( m_maxCUHeight, m_picHeight - y0 );\n"
     "      for( int i = 0; i < h; i += 4 ) for( int j = 0; j < w; j += 4 )\n"
     "      {\n"
     "        const AlfClassifier& k = m_classifier[hCtuBlk * ctu + ( i / 4 ) * ( MAX_CU_SIZE / 4 ) + j / 4];\n"
     "        uint8_t* o = hCls.data() + 2 * ( ( size_t ) ( ( y0 + i ) / 4 ) * hBw + ( x0 + j ) / 4 ); o[0] = k.classIdx; o[1] = k.transposeIdx;\n"
     "      }\n"
     "    }\n"
     "    // luma filter sets in the order of alfCtbFilterIndex: the fixed sets, then the APS sets of the slice\n"
     "    const int hSetLen = MAX_NUM_ALF_CLASSES * MAX_NUM_ALF_LUMA_COEFF, hNumSets = NUM_FIXED_FILTER_SETS + ALF_CTB_MAX_NUM_APS;\n"
     "    std::vector<short> hCoeff( ( size_t ) hNumSets * hSetLen ), hClip( hCoeff.size() );\n"
     "    for( int k = 0; k < NUM_FIXED_FILTER_SETS; k++ ) { memcpy( &hCoeff[( size_t ) k * hSetLen], m_fixedFilterSetCoeffDec[k], sizeof( short ) * hSetLen ); memcpy( &hClip[( size_t ) k * hSetLen], m_clipDefault, sizeof( short ) * hSetLen ); }\n"
     "    for( int k = 0; k < ALF_CTB_MAX_NUM_APS; k++ ) { memcpy( &hCoeff[( size_t ) ( NUM_FIXED_FILTER_SETS + k ) * hSetLen], m_coeffApsLuma[k], sizeof( short ) * hSetLen ); memcpy( &hClip[( size_t ) ( NUM_FIXED_FILTER_SETS + k ) * hSetLen], m_clippApsLuma[k], sizeof( short ) * hSetLen ); }\n"
     "    PelUnitBuf& hRecBuf = cs.getRecoBufRef();\n"
     "    const Pel* hSrc[3]; Pel* hDst[3]; int hSs[3], hDs[3];\n"
     "    for( int c = 0; c < 3; c++ ) { hSrc[c] = m_tempBuf.get( ComponentID( c ) ).buf; hSs[c] = m_tempBuf.get( ComponentID( c ) ).stride; hDst[c] = hRecBuf.get( ComponentID( c ) ).buf; hDs[c] = hRecBuf.get( ComponentID( c ) ).stride; }\n"
     "    const short* hChromaSet[2] = { hAny[1] ? hSet[1].data() : nullptr, hAny[2] ? hSet[2].data() : nullptr };\n"
     "    const bool hOk = g_vvhipHooks.alfFilterPicture( this, cs.picture->poc, hSrc, hSs, hDst, hDs, m_picWidth, m_picHeight, m_inputBitDepth[CH_L], m_maxCUHeight, hCls.data(), hCoeff.data(),\n"
     "        m_encCfg->m_useNonLinearAlfLuma ? hClip.data() : nullptr, hNumSets, hAny[0] ? hSet[0].data() : nullptr, &m_chromaCoeffFinal[0][0], m_encCfg->m_useNonLinearAlfChroma ? &m_chromaClippFinal[0][0] : nullptr,\n"
     "        VVENC_MAX_NUM_ALF_ALTERNATIVES_CHROMA, hChromaSet, m_alfVBLumaCTUHeight, m_alfVBLumaPos, m_alfVBChmaCTUHeight, m_alfVBChmaPos, false );\n"
     "    CHECK( !hOk, \"HIP ALF picture filtering failed\" );\n"
     "    return;\n"
     "  }\n"),
], sub="EncoderLib")

# the recorder learns where a block's compact copy of the original maps to (EncCu keeps one copy per partitioning depth, EncCu.cpp:556-567,1398-1407)
patch("EncCu.cpp", [
    ("after", '#include "EncCu.h"', INC),
    ("before", "  m_modeCtrl.initBlk( tempCS->area, slice.pic->poc );", "  if( g_vvhipHooks.recCu ) g_vvhipHooks.recCu( tempCS );\n"),
    # merge SATD pruning (addRegularCandsToPruningList): the predictions of all regular merge candidates of the CU are scored by ONE device call after the loop that generates them;
    # the list insertions then happen in the same order with the same costs (the item pool gets room for a CU's candidates to be outside the list at once)
    ("replace", "  m_mergeItemList.init( encCfg.m_maxMergeRdCandNumTotal, m_pcEncCfg->m_Geo > 1 ? 3 : 1, chromaFormat, uiMaxSize, uiMaxSize );",
     "  m_mergeItemList.init( encCfg.m_maxMergeRdCandNumTotal, g_vvhipHooks.mergeCosts ? MRG_MAX_NUM_CANDS + 3 : ( m_pcEncCfg->m_Geo > 1 ? 3 : 1 ), chromaFormat, uiMaxSize, uiMaxSize );"),
    ("replace", "         = pu.ciip\n         = false;\n\n  for( uint32_t uiMergeCand = 0; uiMergeCand < mergeCtx.numValidMergeCand; uiMergeCand++ )\n  {\n    if( sameMv[uiMergeCand] ) continue;\n\n    mergeCtx.setMergeInfo   ( pu, uiMergeCand );",
     "         = pu.ciip\n         = false;\n\n"
     "  // (a lossless test mode scores with DF_SAD, EncCu.cpp:1961: the device site covers the Hadamard families only, so it requires that the caller's distParam really holds the\n"
     "  //  Hadamard entry it is about to replace — anything else stays with distParam.distFunc)\n"
     "  const bool hipMrg = g_vvhipHooks.mergeCosts != nullptr && ( localUnitArea.lwidth() & 7 ) == 0 && localUnitArea.lwidth() == localUnitArea.lheight() && localUnitArea.lwidth() <= 64 &&\n"
     "                      distParam.distFunc == m_cRdCost.m_afpDistortFunc[0][( m_pcEncCfg->m_fastHad ? DF_HAD_fast : DF_HAD ) + Log2( localUnitArea.lwidth() )];\n"
     "  MergeItem* hipItem[MRG_MAX_NUM_CANDS]; const Pel* hipPred[MRG_MAX_NUM_CANDS]; int hipStride[MRG_MAX_NUM_CANDS]; uint64_t hipBits[MRG_MAX_NUM_CANDS]; int hipN = 0;\n"
     "  for( uint32_t uiMergeCand = 0; uiMergeCand < mergeCtx.numValidMergeCand; uiMergeCand++ )\n  {\n    if( sameMv[uiMergeCand] ) continue;\n\n    mergeCtx.setMergeInfo   ( pu, uiMergeCand );"),
    ("replace", "    regularMerge->cost      = calcLumaCost4MergePrediction( ctxStart, dstBuf, sqrtLambdaForFirstPassIntra, pu, distParam );\n"
                "    if( CU::checkDMVRCondition( pu ) ) std::copy_n( pu.mvdL0SubPu, getDmvrMvdNum( pu ), m_subPuMvOffset[uiMergeCand].data() );\n"
                "    m_mergeItemList         . insertMergeItemToList( regularMerge );\n  }\n}",
     "    if( hipMrg )\n"
     "    {\n"
     "      hipItem[hipN] = regularMerge; hipPred[hipN] = dstBuf.Y().buf; hipStride[hipN] = ( int ) dstBuf.Y().stride;\n"
     "      m_CABACEstimator->getCtx() = ctxStart; hipBits[hipN] = xCalcPuMeBits( pu ); hipN++;\n"
     "    }\n"
     "    else\n"
     "    regularMerge->cost      = calcLumaCost4MergePrediction( ctxStart, dstBuf, sqrtLambdaForFirstPassIntra, pu, distParam );\n"
     "    if( CU::checkDMVRCondition( pu ) ) std::copy_n( pu.mvdL0SubPu, getDmvrMvdNum( pu ), m_subPuMvOffset[uiMergeCand].data() );\n"
     "    if( !hipMrg )\n"
     "    m_mergeItemList         . insertMergeItemToList( regularMerge );\n  }\n"
     "  if( hipN )\n"
     "  {\n"
     "    uint64_t hipCost[MRG_MAX_NUM_CANDS];\n"
     "    const bool hipOk = g_vvhipHooks.mergeCosts( distParam.org.buf, ( int ) distParam.org.stride, hipPred, hipStride, hipN, distParam.org.width, distParam.org.height, distParam.bitDepth,\n"
     "                                                m_pcEncCfg->m_fastHad ? 2 : 1, hipCost );\n"
     "    for( int i = 0; i < hipN; i++ )\n"
     "    {\n"
     "      Distortion dist = hipCost[i];\n"
     "      if( !hipOk ) { distParam.cur.buf = hipPred[i]; distParam.cur.stride = hipStride[i]; dist = distParam.distFunc( distParam ); }\n"
     "      hipItem[i]->cost = ( double ) dist + ( double ) hipBits[i] * sqrtLambdaForFirstPassIntra;\n"
     "      m_uiSadBestForQPA = std::min( dist, m_uiSadBestForQPA );\n"
     "      m_mergeItemList.insertMergeItemToList( hipItem[i] );\n"
     "    }\n"
     "  }\n}"),
], sub="EncoderLib")

# one picture <-> one device: a worker thread that starts a CTU task of a picture binds itself to that picture's GPU (several GPUs only)
patch("EncSlice.cpp", [
    ("after", '#include "EncSlice.h"', INC),
    ("after", "  CtuEncParam* ctuEncParam       = static_cast<CtuEncParam*>( taskParam );\n  Picture* pic                   = ctuEncParam->pic;\n",
     "  if( !checkReadyState && g_vvhipHooks.bindPicture ) g_vvhipHooks.bindPicture( pic->poc );\n"
     "  if( !checkReadyState && g_vvhipHooks.recPicture ) g_vvhipHooks.recPicture( pic );\n"),
    # a CTU row of the reconstruction is final (borders extended): its luma rows go to the picture's device mirrors
    ("before", "          // for IFP lines synchro, do an additional increment signaling that CTU row is ready",
     "          if( g_vvhipHooks.reconRows )\n"
     "          {\n"
     "            const CPelBuf hY = recoBuf.Y();\n"
     "            const int hTop = ctuPosY == 0 ? margin : 0, hBot = ctuPosY + 1 == pcv.heightInCtus ? margin : 0;\n"
     "            g_vvhipHooks.reconRows( hY.buf, ( int ) hY.stride, hY.width, hY.height, margin, y - hTop, height + hTop + hBot );\n"
     "          }\n"),
], sub="EncoderLib")

# run-time selection through the reference's own switch: --SIMD=HIP[:mask] / vvenc_set_SIMD_extension( "HIP" ) installs the binding, the CPU levels below it stay at their best
patch("vvencimpl.cpp", [
    ("after", '#include "vvencimpl.h"', INC),
    ("replace", "  const std::string simdReqStr( simdId ? simdId : \"\" );\n",
     "  std::string hipReq( simdId ? simdId : \"\" );\n"
     "  bool hipSelected = false;\n"
     "  if( hipReq.compare( 0, 3, \"HIP\" ) == 0 )\n"
     "  {\n"
     "    if( vvenc_hip_select( hipReq.c_str() ) != 0 ) { MsgLog msg; msg.log( VVENC_ERROR, \"\\nrequested SIMD level (%s) not available: no MI355X context\\n\", hipReq.c_str() ); return nullptr; }\n"
     "    hipSelected = true; hipReq.clear();\n"
     "  }\n"
     "  const std::string simdReqStr( hipReq );\n"),
    # the encoder is being closed: its picture buffers are about to be freed -> the binding drops device mirrors and host pins BEFORE that happens
    ("before", "      m_pEncLib->uninitEncoderLib();\n      delete m_pEncLib;", "      vvenc_hip_release();\n"),
    ("replace", "    return read_x86_extension_name().c_str();\n# endif   // !TARGET_SIMD_ARM",
     "    if( hipSelected ) { static std::string hipName; hipName = \"HIP+\" + read_x86_extension_name(); return hipName.c_str(); }\n"
     "    return read_x86_extension_name().c_str();\n# endif   // !TARGET_SIMD_ARM"),
], sub="vvenc")
print("binding applied to", out)
