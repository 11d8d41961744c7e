// cavlc_bits.h -- how many bits the CAVLC writer will spend on a macroblock, counted on the device.
//
// Reference behaviour restated (codec/encoder/core/src): set_mb_syn_cavlc.cpp:84-232 CavlcParamCal_c / WriteBlockResidualCavlc,
// svc_set_mb_syn_cavlc.cpp:58-245 WelsSpatialWriteMbPred / WelsSpatialWriteSubMbPred, :260-322 WelsSpatialWriteMbSyn, :324-440
// WelsWriteMbResidual -- lengths only, nothing is written.  What the count is for: rate control with one slice per picture sets
// the QP of a group of macroblocks from the bits the groups before it produced (WelsRcMbInitGom / RcCalculateGomQp,
// ratectl.cpp:748-775,1239-1262); WelsRcMbInfoUpdateGom measures a macroblock as the bitstream position after it minus the
// position before it.  Two terms of that difference depend on the macroblocks BEFORE this one in coding order and are left to
// whoever walks the macroblocks in that order: ue(mb_skip_run) in front of a coded macroblock of a P slice, and se(mb_qp_delta)
// of a macroblock that codes one (WH_BITS_HAS_QP_DELTA).  Everything else is counted here, one lane per 4x4 block.
#pragma once
#include "mb_common.h"

#define WH_BITS_HAS_QP_DELTA 0x40000000      /* flag in the count: the macroblock codes mb_qp_delta (cbp > 0 or Intra16x16) */

WH_FN int wh_ue_bits (unsigned v) { return 2 * (31 - __builtin_clz (v + 1u)) + 1; }       // (v + 1 >= 1 for every code number that occurs)
WH_FN int wh_se_bits_c (int v) { return wh_ue_bits ((unsigned) (v > 0 ? 2 * v - 1 : -2 * v)); }

// residual_block_cavlc (7.3.5.3.2 / 9.2): lv[0..end_idx] zig-zag levels; nc: 0..16, or 17 for ChromaDCLevel.
WH_FN int wh_cavlc_block_bits (const int16_t* lv, int end_idx, int nc, bool has_coeff) {
  // (a block whose total_coeff is 0 is coded as empty without looking at its levels, which may be stale after the zeroing
  //  heuristics: iCalRunLevelFlag, svc_encode_mb.cpp:283-287)
  int i = has_coeff ? end_idx : -1;
  while (i >= 0 && lv[i] == 0) --i;
  int total = 0, total_zeros = 0, t1 = 0, bits = 0, suffix_len = 0;
  bool t1_open = true;
  int run_bits = 0, zeros_seen = 0;
  // first pass: total_coeff, trailing ones, total_zeros (highest frequency first, as the syntax codes them)
  int j = i;
  while (j >= 0) {
    const int v = lv[j--];
    if (t1_open && total < 3 && (v == 1 || v == -1)) ++t1; else t1_open = false;
    ++total;
    while (j >= 0 && lv[j] == 0) { ++total_zeros; --j; }
  }
  const uint16_t tok = kWhCoeffToken[(kWhNcClass[nc] * 17 + total) * 4 + t1];
  bits += (int) (tok >> 8);
  if (total == 0) return bits;
  bits += t1;
  // second pass: level codes after the trailing ones, run_before
  suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
  int k = 0, zeros_left = total_zeros;
  j = i;
  while (j >= 0) {
    const int val = lv[j--];
    int zeros = 0;
    while (j >= 0 && lv[j] == 0) { ++zeros; --j; }
    if (k >= t1) {
      int code = (val > 0) ? 2 * val - 2 : -2 * val - 1;            // levelCode
      if (k == t1 && t1 < 3) code -= 2;
      int prefix = code >> suffix_len, suffix_size = suffix_len;
      if (prefix >= 14 && prefix < 30 && suffix_len == 0) { prefix = 14; suffix_size = 4; }
      else if (prefix >= 15) { prefix = 15; suffix_size = 12; }
      bits += prefix + 1 + suffix_size;
      if (suffix_len == 0) suffix_len = 1;
      const int thr = 3 << (suffix_len - 1);
      if ((val > thr || val < -thr) && suffix_len < 6) ++suffix_len;
    }
    if (k + 1 < total && zeros_left > 0) {
      const int zl = zeros_left > 7 ? 7 : zeros_left;
      run_bits += (int) (kWhRunBefore[zl * 15 + zeros] >> 8);
      zeros_left -= zeros;
    }
    ++k;
  }
  (void)zeros_seen;
  if (total < end_idx + 1) bits += (int) (((nc != 17) ? kWhTotalZeros[total * 16 + total_zeros] : kWhTotalZerosChromaDc[total * 4 + total_zeros]) >> 8);
  return bits + run_bits;
}

WH_FN int wh_nc_of (int na, int nb) {          // WELS_NON_ZERO_COUNT_AVERAGE (macros.h:135-139), -1 = unavailable
  int nc = na + nb + 1;
  nc >>= (na != -1 && nb != -1);
  nc += (na == -1 && nb == -1);
  return nc;
}
WH_FN int wh_blk_raster (int b) { return (((b >> 1) & 1) | ((b >> 2) & 2)) * 4 + ((b & 1) | ((b >> 1) & 2)); }

// Residual bits of the macroblock whose levels and total_coeff counts are in the tile (lv_luma / lv_dc / lv_cdc / lv_cac, nzc).
// nzc_left / nzc_top: the neighbour macroblocks' total_coeff arrays (WhMbState::nzc layout) when they belong to the slice, else NULL.
WH_FN int wh_mb_residual_bits (WhMbLds& S, int mb_type, int cbp, const uint8_t* nzc_left, const uint8_t* nzc_top) {
  const int cbp_l = cbp & 15, cbp_c = cbp >> 4;
  int bits;
  WV_SUM (bits, lane, ([&] () -> int {
    if (lane < 16) {                                               // luma block, luma4x4BlkIdx = lane
      const bool i16 = mb_type == WH_MB_I16x16;
      if (i16 ? cbp_l == 0 : ! (cbp_l & (1 << (lane >> 2)))) return 0;
      const int r = wh_blk_raster (lane);
      const int na = (r & 3) ? (int)S.nzc[r - 1] : (nzc_left ? (int)nzc_left[r + 3] : -1);
      const int nb = (r >> 2) ? (int)S.nzc[r - 4] : (nzc_top ? (int)nzc_top[r + 12] : -1);
      return wh_cavlc_block_bits (&S.lv_luma[lane * 16], i16 ? 14 : 15, wh_nc_of (na, nb), S.nzc[r] > 0);
    }
    if (lane == 16) {                                              // Intra16x16 DC
      if (mb_type != WH_MB_I16x16) return 0;
      const int na = nzc_left ? (int)nzc_left[3] : -1, nb = nzc_top ? (int)nzc_top[12] : -1;
      return wh_cavlc_block_bits (S.lv_dc, 15, wh_nc_of (na, nb), true);
    }
    if (lane < 19) return cbp_c ? wh_cavlc_block_bits (&S.lv_cdc[(lane - 17) * 4], 3, 17, true) : 0;       // chroma DC
    if (lane < 27) {                                               // chroma AC: plane p, raster 2x2 index c
      if (! (cbp_c & 2)) return 0;
      const int i = lane - 19, p = i >> 2, c = i & 3;
      const int na = (c & 1) ? (int)S.nzc[16 + p * 4 + c - 1] : (nzc_left ? (int)nzc_left[16 + p * 4 + c + 1] : -1);
      const int nb = (c >> 1) ? (int)S.nzc[16 + p * 4 + c - 2] : (nzc_top ? (int)nzc_top[16 + p * 4 + c + 2] : -1);
      return wh_cavlc_block_bits (&S.lv_cac[i * 16], 14, wh_nc_of (na, nb), S.nzc[16 + p * 4 + c] > 0);
    }
    return 0; }) ());
  return bits;
}

// Header bits of an intra macroblock (mb_type, prediction modes, coded_block_pattern); `p_slice`: mb_type numbers are offset by 5.
WH_FN int wh_mb_intra_header_bits (const WhMbLds& S, int mb_type, int cbp, int i16_mode_std, int chroma_mode_std, bool p_slice) {
  const int off = p_slice ? 5 : 0;
  int bits;
  if (mb_type == WH_MB_I4x4) {
    const int prev = __builtin_popcount ((unsigned)S.i4_prev & 0xffffu);
    bits = wh_ue_bits ((unsigned)off) + 16 + 3 * (16 - prev) + wh_ue_bits ((unsigned)chroma_mode_std) + wh_ue_bits (kWhCbpCodeIntra[cbp]);
  } else {
    bits = wh_ue_bits ((unsigned) (1 + off + i16_mode_std + ((cbp >> 4) << 2) + ((cbp & 15) ? 12 : 0))) + wh_ue_bits ((unsigned)chroma_mode_std);
  }
  return bits;
}
