// oracle/ref_prims_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" trampolines onto the reference's own C-fallback primitives (the `_c` functions the
// encoder installs into SWelsFuncPtrList when no SIMD flag is set), with the signatures of
// oracle/prims/oracle_prims.h and the prefix ref_.  Built into oracle/_ref/libref_prims.so by
// oracle/Makefile from the sources under /root/reference; tests/test_oracle_prims.py uses it to pin
// the plain-C restatement (liboracle_prims.so) against the real thing.  Function tables are obtained
// the way the reference's own unit tests do: Init*(…, 0) with CPU flag 0 ("fake backends").
#include <string.h>
#include "wels_func_ptr_def.h"
#include "sample.h"
#include "sad_common.h"
#include "encode_mb_aux.h"
#include "decode_mb_aux.h"
#include "get_intra_predictor.h"
#include "mc.h"
#include "deblocking_common.h"
#include "svc_encode_mb.h"
#include "wels_common_defs.h"
#include "copy_mb.h"

using namespace WelsEnc;
namespace WelsEnc { void WelsSetMemZero_c (void* pDst, int32_t iSize); }     // codec/encoder/core/inc/encoder.h:122
namespace WelsVP {   // codec/processing/src/vaacalc/vaacalculation.h:85
void VAACalcSad_c (const uint8_t* pCurData, const uint8_t* pRefData, int32_t iPicWidth, int32_t iPicHeight, int32_t iPicStride,
                   int32_t* pFrameSad, int32_t* pSad8x8);
// codec/processing/src/downsample/downsample.h:100-121 (the C fallbacks of SDownsampleFuncs)
void DyadicBilinearDownsampler_c (uint8_t* pDst, const int32_t kiDstStride, uint8_t* pSrc, const int32_t kiSrcStride, const int32_t kiSrcWidth, const int32_t kiSrcHeight);
void DyadicBilinearQuarterDownsampler_c (uint8_t* pDst, const int32_t kiDstStride, uint8_t* pSrc, const int32_t kiSrcStride, const int32_t kiSrcWidth, const int32_t kiSrcHeight);
void DyadicBilinearOneThirdDownsampler_c (uint8_t* pDst, const int32_t kiDstStride, uint8_t* pSrc, const int32_t kiSrcStride, const int32_t kiSrcWidth, const int32_t kiDstHeight);
void GeneralBilinearFastDownsampler_c (uint8_t* pDst, const int32_t kiDstStride, const int32_t kiDstWidth, const int32_t kiDstHeight,
                                       uint8_t* pSrc, const int32_t kiSrcStride, const int32_t kiSrcWidth, const int32_t kiSrcHeight);
void GeneralBilinearAccurateDownsampler_c (uint8_t* pDst, const int32_t kiDstStride, const int32_t kiDstWidth, const int32_t kiDstHeight,
                                           uint8_t* pSrc, const int32_t kiSrcStride, const int32_t kiSrcWidth, const int32_t kiSrcHeight);
}

namespace {
struct Tables {
  SWelsFuncPtrList fl;
  SMcFunc mc;
  Tables() {
    memset (&fl, 0, sizeof (fl));
    WelsInitSampleSadFunc (&fl, 0);
    WelsInitEncodingFuncs (&fl, 0);
    WelsInitReconstructionFuncs (&fl, 0);
    WelsInitIntraPredFuncs (&fl, 0);
    WelsCommon::InitMcFunc (&mc, 0);
  }
};
Tables& T() { static Tables t; return t; }
}

extern "C" {

int32_t ref_sad (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb) {
  return T().fl.sSampleDealingFuncs.pfSampleSad[blk] ((uint8_t*)a, sa, (uint8_t*)b, sb);
}
void ref_sad_four (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb, int32_t* out4) {
  T().fl.sSampleDealingFuncs.pfSample4Sad[blk] ((uint8_t*)a, sa, (uint8_t*)b, sb, out4);
}
int32_t ref_satd (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb) {
  return T().fl.sSampleDealingFuncs.pfSampleSatd[blk] ((uint8_t*)a, sa, (uint8_t*)b, sb);
}
void ref_dct4x4 (int16_t* dct, const uint8_t* p1, int32_t s1, const uint8_t* p2, int32_t s2) { T().fl.pfDctT4 (dct, (uint8_t*)p1, s1, (uint8_t*)p2, s2); }
void ref_hadamard_t4_dc (int16_t* out, const int16_t* dct256) { T().fl.pfTransformHadamard4x4Dc (out, (int16_t*)dct256); }
void ref_quant4x4 (int16_t* d, int qp, int intra) { T().fl.pfQuantization4x4 (d, g_kiQuantInterFF[qp + (intra ? 6 : 0)], g_kiQuantMF[qp]); }
int32_t ref_quant4x4_max (int16_t* d, int qp, int intra) {
  int16_t blk[64], mx[4];
  memset (blk, 0, sizeof (blk));
  memcpy (blk, d, 32);
  T().fl.pfQuantizationFour4x4Max (blk, g_kiQuantInterFF[qp + (intra ? 6 : 0)], g_kiQuantMF[qp], mx);
  memcpy (d, blk, 32);
  return mx[0];
}
void ref_quant4x4_dc (int16_t* d, int16_t ff, int16_t mf) { T().fl.pfQuantizationDc4x4 (d, ff, mf); }
int32_t ref_hadamard_quant2x2 (int16_t* rs, int16_t ff, int16_t mf, int16_t* dct4, int16_t* blk4) { return T().fl.pfQuantizationHadamard2x2 (rs, ff, mf, dct4, blk4); }
int32_t ref_hadamard_quant2x2_skip (const int16_t* rs, int16_t ff, int16_t mf) { return T().fl.pfQuantizationHadamard2x2Skip ((int16_t*)rs, ff, mf); }
void ref_scan4x4_dcac (int16_t* lv, const int16_t* d) { T().fl.pfScan4x4 (lv, (int16_t*)d); }
void ref_scan4x4_ac (int16_t* lv, const int16_t* d) { T().fl.pfScan4x4Ac (lv, (int16_t*)d); }
int32_t ref_single_ctr4x4 (const int16_t* lv) { return T().fl.pfCalculateSingleCtr4x4 ((int16_t*)lv); }
int32_t ref_nonzero_count (const int16_t* lv) { return T().fl.pfGetNoneZeroCount ((int16_t*)lv); }
void ref_dequant4x4 (int16_t* r, int qp) { T().fl.pfDequantization4x4 (r, WelsCommon::g_kuiDequantCoeff[qp]); }
void ref_dequant_ihadamard4x4 (int16_t* r, int qp) {       // dispatch of svc_encode_mb.cpp:99-105
  if (qp < 12) { WelsIHadamard4x4Dc (r); WelsDequantLumaDc4x4 (r, qp); }
  else T().fl.pfDequantizationIHadamard4x4 (r, WelsCommon::g_kuiDequantCoeff[qp][0] >> 2);
}
void ref_dequant_ihadamard2x2_dc (int16_t* d, int qp) { WelsDequantIHadamard2x2Dc (d, WelsCommon::g_kuiDequantCoeff[qp][0]); }
void ref_idct4x4_rec (uint8_t* rec, int32_t rs, const uint8_t* pred, int32_t ps, const int16_t* d) { T().fl.pfIDctT4 (rec, rs, (uint8_t*)pred, ps, (int16_t*)d); }

void ref_pred_i4x4 (int mode, uint8_t* pred, const uint8_t* ref, int32_t st) { T().fl.pfGetLumaI4x4Pred[mode] (pred, (uint8_t*)ref, st); }
void ref_pred_i16x16 (int mode, uint8_t* pred, const uint8_t* ref, int32_t st) { T().fl.pfGetLumaI16x16Pred[mode] (pred, (uint8_t*)ref, st); }
void ref_pred_chroma (int mode, uint8_t* pred, const uint8_t* ref, int32_t st) { T().fl.pfGetChromaPred[mode] (pred, (uint8_t*)ref, st); }

void ref_mc_luma (const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int mvx, int mvy, int w, int h) { T().mc.pMcLumaFunc (src, ss, dst, ds, (int16_t)mvx, (int16_t)mvy, w, h); }
void ref_mc_chroma (const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int mvx, int mvy, int w, int h) { T().mc.pMcChromaFunc (src, ss, dst, ds, (int16_t)mvx, (int16_t)mvy, w, h); }

// NB the reference's naming: ...V filters across a horizontal edge (samples step by the stride), ...H across a vertical edge
void ref_deblock_luma_lt4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta, const int8_t* tc4) {
  if (horizontal) DeblockLumaLt4H_c (pix, stride, alpha, beta, (int8_t*)tc4); else DeblockLumaLt4V_c (pix, stride, alpha, beta, (int8_t*)tc4);
}
void ref_deblock_luma_eq4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta) {
  if (horizontal) DeblockLumaEq4H_c (pix, stride, alpha, beta); else DeblockLumaEq4V_c (pix, stride, alpha, beta);
}
void ref_deblock_chroma_lt4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta, const int8_t* tc4) {
  if (horizontal) DeblockChromaLt4H2_c (pix, stride, alpha, beta, (int8_t*)tc4); else DeblockChromaLt4V2_c (pix, stride, alpha, beta, (int8_t*)tc4);
}
void ref_deblock_chroma_eq4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta) {
  if (horizontal) DeblockChromaEq4H2_c (pix, stride, alpha, beta); else DeblockChromaEq4V2_c (pix, stride, alpha, beta);
}

void ref_vaa_sad8x8 (const uint8_t* cur, const uint8_t* ref, int32_t stride, int32_t* sad4) {
  int32_t frame_sad = 0;
  WelsVP::VAACalcSad_c (cur, ref, 16, 16, stride, &frame_sad, sad4);
}

// mode numbering of include/welship.h (WELSHIP_DS_*): 0 half, 1 quarter, 2 one third, 3 general fast, 4 general accurate
void ref_downsample (int mode, uint8_t* dst, int32_t dst_stride, int32_t dst_w, int32_t dst_h, uint8_t* src, int32_t src_stride, int32_t src_w, int32_t src_h) {
  switch (mode) {
  case 0: WelsVP::DyadicBilinearDownsampler_c (dst, dst_stride, src, src_stride, src_w, src_h); break;
  case 1: WelsVP::DyadicBilinearQuarterDownsampler_c (dst, dst_stride, src, src_stride, src_w, src_h); break;
  case 2: WelsVP::DyadicBilinearOneThirdDownsampler_c (dst, dst_stride, src, src_stride, src_w, dst_h); break;
  case 3: WelsVP::GeneralBilinearFastDownsampler_c (dst, dst_stride, dst_w, dst_h, src, src_stride, src_w, src_h); break;
  default: WelsVP::GeneralBilinearAccurateDownsampler_c (dst, dst_stride, dst_w, dst_h, src, src_stride, src_w, src_h); break;
  }
}


// the Combined3 intra costs (sample.cpp:153-331: `_c` functions no C build installs, sample.cpp:363-367), block copies and memory clearing
// (copy_mb.cpp:38-111) -- the references of the leaf exports of include/welship_leaf.h that round 6 added
int32_t ref_intra4x4_combined3_satd (uint8_t* dec, int32_t ds, uint8_t* enc, int32_t es, uint8_t* dst, int32_t* mode, int32_t l2, int32_t l1, int32_t l0) {
  return WelsSampleSatdIntra4x4Combined3_c (dec, ds, enc, es, dst, mode, l2, l1, l0);
}
int32_t ref_intra16x16_combined3 (int satd, uint8_t* dec, int32_t ds, uint8_t* enc, int32_t es, int32_t* mode, int32_t lambda, uint8_t* dst) {
  return satd ? WelsSampleSatdIntra16x16Combined3_c (dec, ds, enc, es, mode, lambda, dst) : WelsSampleSadIntra16x16Combined3_c (dec, ds, enc, es, mode, lambda, dst);
}
int32_t ref_intra8x8_combined3 (int satd, uint8_t* dec_cb, int32_t ds, uint8_t* enc_cb, int32_t es, int32_t* mode, int32_t lambda, uint8_t* dst, uint8_t* dec_cr, uint8_t* enc_cr) {
  return satd ? WelsSampleSatdIntra8x8Combined3_c (dec_cb, ds, enc_cb, es, mode, lambda, dst, dec_cr, enc_cr)
              : WelsSampleSadIntra8x8Combined3_c (dec_cb, ds, enc_cb, es, mode, lambda, dst, dec_cr, enc_cr);
}
void ref_copy (int w, int h, uint8_t* dst, int32_t ds, uint8_t* src, int32_t ss) {
  PCopyFunc f = w == 4 && h == 4 ? WelsCopy4x4_c : w == 8 && h == 4 ? WelsCopy8x4_c : w == 4 && h == 8 ? WelsCopy4x8_c : w == 8 && h == 8 ? WelsCopy8x8_c
              : w == 16 && h == 8 ? WelsCopy16x8_c : w == 8 && h == 16 ? WelsCopy8x16_c : WelsCopy16x16_c;
  f (dst, ds, src, ss);
}
void ref_quant_rows (int qp, int intra, int16_t* ff8, int16_t* mf8) { memcpy (ff8, g_kiQuantInterFF[qp + (intra ? 6 : 0)], 16); memcpy (mf8, g_kiQuantMF[qp], 16); }
void ref_set_mem_zero (void* dst, int32_t size) { WelsSetMemZero_c (dst, size); }

}  // extern "C"
