/* welship_leaf.h -- layer (3c) of the C ABI: the leaf primitives with EXACTLY the reference's function-pointer typedefs, one export
 * per table slot (95 exports; several slots share one, e.g. the three pfSetMemZero* or the two pfCopy16x16*), so that each can be stored into SWelsFuncPtrList / SMcFunc / DeblockingFunc without a cast
 * (codec/encoder/core/inc/wels_func_ptr_def.h:58-188, codec/common/inc/mc.h:40-53; SURVEY.md 8b "signatures a C-ABI replacement must
 * export (leaf level)").  integration/welship_hooks.cpp installs them with WELS_HIP_LEAVES=1 (static_asserts there tie every export
 * to its typedef) and tests/test_leaf_gpu.py runs the unmodified encoder loop on top of them against the reference's C functions.
 *
 * This layer is for integration bring-up and parity checking: every call stages its few hundred bytes through HBM and runs one
 * wavefront, i.e. costs tens of microseconds -- the reason the throughput path is the frame-level entry points of welship.h (the
 * survey's "why leaf-level alone is not the GPU design").  The arithmetic is the device code of the fused macroblock kernels
 * (csrc/hip/prims_kernels.h).  The typedefs have no error channel: a call without a usable device, or one whose launch fails,
 * prints the HIP error to stderr and abort()s -- never a silent CPU result.  Ask WelsHipLeafAvailable() before installing.
 * Thread-safe (slice threads call through the shared table, wels_task_encoder.cpp:148-199): staging buffers and queues come from
 * a pool, one per concurrent caller. */
#ifndef WELSHIP_LEAF_H_
#define WELSHIP_LEAF_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* WELSHIP_OK when a device is usable, WELSHIP_ERR_NO_DEVICE otherwise (welship.h) */
int WelsHipLeafAvailable (void);
/* calls served since the library was loaded (the installer's trace and the tests read it) */
uint64_t WelsHipLeafCalls (void);

/* ---- SSampleDealingFunc (wels_func_ptr_def.h:147-177): PSampleSadSatdCostFunc pfSampleSad[] / pfSampleSatd[], PSample4SadCostFunc
 * pfSample4Sad[] indexed BLOCK_16x16 .. BLOCK_4x8 (sample.cpp:336-357; sad_common.cpp:44-165, sample.cpp:47-156) */
int32_t WelsHipSampleSad16x16 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSad16x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSad8x16 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSad8x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSad4x4 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSad8x4 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSad4x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSatd16x16 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSatd16x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSatd8x16 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSatd8x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSatd4x4 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSatd8x4 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
int32_t WelsHipSampleSatd4x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
void WelsHipSampleSadFour16x16 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2, int32_t* pSad);
void WelsHipSampleSadFour16x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2, int32_t* pSad);
void WelsHipSampleSadFour8x16 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2, int32_t* pSad);
void WelsHipSampleSadFour8x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2, int32_t* pSad);
void WelsHipSampleSadFour4x4 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2, int32_t* pSad);
void WelsHipSampleSadFour8x4 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2, int32_t* pSad);
void WelsHipSampleSadFour4x8 (uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2, int32_t* pSad);

/* ---- transform / quantisation (encode_mb_aux.cpp:161-451, table :473-489): PDctFunc pfDctT4 / pfDctFourT4, PQuantizationFunc
 * pfQuantization4x4 / pfQuantizationFour4x4, PQuantizationDcFunc pfQuantizationDc4x4, PQuantizationMaxFunc pfQuantizationFour4x4Max,
 * PQuantizationHadamardFunc pfQuantizationHadamard2x2, PQuantizationSkipFunc pfQuantizationHadamard2x2Skip,
 * PTransformHadamard4x4Func pfTransformHadamard4x4Dc, PScanFunc pfScan4x4 / pfScan4x4Ac, PCalculateSingleCtrFunc
 * pfCalculateSingleCtr4x4, PGetNoneZeroCountFunc pfGetNoneZeroCount */
void WelsHipDctT4 (int16_t* pDct, uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
void WelsHipDctFourT4 (int16_t* pDct, uint8_t* pSample1, int32_t iStride1, uint8_t* pSample2, int32_t iStride2);
void WelsHipQuant4x4 (int16_t* pDct, const int16_t* pFF, const int16_t* pMF);
void WelsHipQuant4x4Dc (int16_t* pDct, int16_t iFF, int16_t iMF);
void WelsHipQuantFour4x4 (int16_t* pDct, const int16_t* pFF, const int16_t* pMF);
void WelsHipQuantFour4x4Max (int16_t* pDct, const int16_t* pFF, const int16_t* pMF, int16_t* pMax);
int32_t WelsHipHadamardQuant2x2 (int16_t* pRes, const int16_t kiFF, int16_t iMF, int16_t* pDct, int16_t* pBlock);
int32_t WelsHipHadamardQuant2x2Skip (int16_t* pRes, int16_t iFF, int16_t iMF);
void WelsHipHadamardT4Dc (int16_t* pLumaDc, int16_t* pDct);
void WelsHipScan4x4DcAc (int16_t* pLevel, int16_t* pDct);
void WelsHipScan4x4Ac (int16_t* pLevel, int16_t* pDct);
int32_t WelsHipCalculateSingleCtr4x4 (int16_t* pDct);
int32_t WelsHipGetNoneZeroCount (int16_t* pLevel);

/* ---- reconstruction (decode_mb_aux.cpp:107-233, table :252-258): PDeQuantizationFunc pfDequantization4x4 / pfDequantizationFour4x4,
 * PDeQuantizationHadamardFunc pfDequantizationIHadamard4x4, PIDctFunc pfIDctT4 / pfIDctFourT4 / pfIDctI16x16Dc */
void WelsHipDequant4x4 (int16_t* pRes, const uint16_t* kpQpTable);
void WelsHipDequantFour4x4 (int16_t* pRes, const uint16_t* kpQpTable);
void WelsHipDequantIHadamard4x4 (int16_t* pRes, const uint16_t kuiMF);
void WelsHipIDctT4Rec (uint8_t* pRec, int32_t iStride, uint8_t* pPred, int32_t iPredStride, int16_t* pRes);
void WelsHipIDctFourT4Rec (uint8_t* pRec, int32_t iStride, uint8_t* pPred, int32_t iPredStride, int16_t* pRes);
void WelsHipIDctRecI16x16Dc (uint8_t* pRec, int32_t iStride, uint8_t* pPred, int32_t iPredStride, int16_t* pRes);

/* ---- SMcFunc (codec/common/inc/mc.h:40-53; mc.cpp:162-378, table :4529-4534): PWelsMcFunc pMcLumaFunc / pMcChromaFunc,
 * PWelsLumaHalfpelMcFunc pfLumaHalfpelHor / Ver / Cen (widths and heights 4..17), PWelsSampleAveragingFunc pfSampleAveraging */
void WelsHipMcLuma (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int16_t iMvX, int16_t iMvY, int32_t iWidth, int32_t iHeight);
void WelsHipMcChroma (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int16_t iMvX, int16_t iMvY, int32_t iWidth, int32_t iHeight);
void WelsHipMcHorVer20 (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int32_t iWidth, int32_t iHeight);
void WelsHipMcHorVer02 (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int32_t iWidth, int32_t iHeight);
void WelsHipMcHorVer22 (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int32_t iWidth, int32_t iHeight);
void WelsHipPixelAvg (uint8_t* pDst, int32_t iDstStride, const uint8_t* pSrcA, int32_t iSrcAStride, const uint8_t* pSrcB, int32_t iSrcBStride, int32_t iWidth, int32_t iHeight);

/* ---- the 28 intra predictors, PGetIntraPredFunc (get_intra_predictor.cpp:79-646, intra_pred_common.cpp:47-77):
 * pfGetLumaI4x4Pred[I4_PRED_V .. I4_PRED_VL_TOP] (pPred 4x4, pitch 4), pfGetLumaI16x16Pred[I16_PRED_V .. I16_PRED_DC_128] (pPred 16x16,
 * pitch 16), pfGetChromaPred[C_PRED_DC .. C_PRED_DC_128] (pPred 8x8, pitch 8); pRef = the block's first sample in the reconstruction */
void WelsHipI4x4LumaPredV (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredH (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredDc (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredDcLeft (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredDcTop (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredDcNA (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredDDL (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredDDLTop (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredDDR (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredVL (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredVLTop (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredVR (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredHU (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI4x4LumaPredHD (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI16x16LumaPredV (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI16x16LumaPredH (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI16x16LumaPredDc (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI16x16LumaPredPlane (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI16x16LumaPredDcLeft (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI16x16LumaPredDcTop (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipI16x16LumaPredDcNA (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipIChromaPredDc (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipIChromaPredH (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipIChromaPredV (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipIChromaPredPlane (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipIChromaPredDcLeft (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipIChromaPredDcTop (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);
void WelsHipIChromaPredDcNA (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride);

/* ---- DeblockingFunc (wels_func_ptr_def.h:77-102; deblocking_common.cpp:5-181): PLumaDeblockingLT4Func / EQ4Func pfLumaDeblocking*{Ver,Hor},
 * PChromaDeblockingLT4Func / EQ4Func pfChromaDeblocking*{Ver,Hor}.  "V" filters across a horizontal edge (neighbours one stride apart),
 * "H" across a vertical one, as the reference's DeblockLumaLt4V_c / ..H_c do. */
void WelsHipDeblockLumaLt4V (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc);
void WelsHipDeblockLumaEq4V (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta);
void WelsHipDeblockLumaLt4H (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc);
void WelsHipDeblockLumaEq4H (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta);
void WelsHipDeblockChromaLt4V (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc);
void WelsHipDeblockChromaEq4V (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta);
void WelsHipDeblockChromaLt4H (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc);
void WelsHipDeblockChromaEq4H (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta);

/* ---- PCopyFunc (wels_func_ptr_def.h:61, slots :238-245: pfCopy16x16Aligned, pfCopy16x16NotAligned, pfCopy8x8Aligned, pfCopy16x8NotAligned,
 * pfCopy8x16Aligned, pfCopy4x4, pfCopy8x4, pfCopy4x8; copy_mb.cpp:48-111) and PSetMemoryZero (:58, slots :285-287: pfSetMemZeroSize8,
 * pfSetMemZeroSize64Aligned16, pfSetMemZeroSize64; copy_mb.cpp:38-46) */
void WelsHipCopy4x4 (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS);
void WelsHipCopy8x4 (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS);
void WelsHipCopy4x8 (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS);
void WelsHipCopy8x8 (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS);
void WelsHipCopy16x8 (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS);
void WelsHipCopy8x16 (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS);
void WelsHipCopy16x16 (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS);
void WelsHipSetMemZero (void* pDst, int32_t iSize);

/* ---- PIntraPred4x4Combined3Func / PIntraPred16x16Combined3Func / PIntraPred8x8Combined3Func (wels_func_ptr_def.h:129-133; slots :166-170
 * pfIntra4x4Combined3Satd, pfIntra16x16Combined3Satd / Sad, pfIntra8x8Combined3Satd / Sad, from which InitIntraAnalysisVaaInfo / the complexity
 * mode pick :174-176).  The reference's C build leaves these slots NULL (sample.cpp:363-367); its `_c` functions (sample.cpp:153-331) are what
 * the exports restate: V / H / DC of a block predicted and costed in one call, the best mode and cost returned. */
int32_t WelsHipIntra4x4Combined3Satd (uint8_t* pDec, int32_t iDecStride, uint8_t* pEnc, int32_t iEncStride, uint8_t* pDst, int32_t* pBestMode,
                                     int32_t iLambda2, int32_t iLambda1, int32_t iLambda0);
int32_t WelsHipIntra16x16Combined3Satd (uint8_t* pDec, int32_t iDecStride, uint8_t* pEnc, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda, uint8_t* pDst);
int32_t WelsHipIntra16x16Combined3Sad (uint8_t* pDec, int32_t iDecStride, uint8_t* pEnc, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda, uint8_t* pDst);
int32_t WelsHipIntra8x8Combined3Satd (uint8_t* pDecCb, int32_t iDecStride, uint8_t* pEncCb, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda,
                                     uint8_t* pDstChroma, uint8_t* pDecCr, uint8_t* pEncCr);
int32_t WelsHipIntra8x8Combined3Sad (uint8_t* pDecCb, int32_t iDecStride, uint8_t* pEncCb, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda,
                                    uint8_t* pDstChroma, uint8_t* pDecCr, uint8_t* pEncCr);

#ifdef __cplusplus
}
#endif
#endif  /* WELSHIP_LEAF_H_ */
