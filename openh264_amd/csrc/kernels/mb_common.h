// mb_common.h -- LDS tile of one macroblock wave + residual coding pipelines shared by the
// intra and inter MB kernels.
//
// Reference behaviour restated here (all under codec/encoder/core/src unless noted):
//   svc_encode_mb.cpp:54-137   WelsEncRecI16x16Y      -> wh_encrec_i16
//   svc_encode_mb.cpp:139-178  WelsEncRecI4x4Y        -> wh_encrec_i4
//   svc_encode_mb.cpp:244-312  WelsEncRecUV           -> wh_encrec_chroma
//   encode_mb_aux.cpp / decode_mb_aux.cpp primitives  -> prims.h
#pragma once
#include "prims.h"

// ---- the per-wave LDS tile ----------------------------------------------------------------------
// rec_y : reconstructed luma incl. one row above and one column left:  pixel (x,y), x in [-8,23],
//         y in [-1,15]  at  rec_y[(y+1)*32 + x + 8]
// rec_c : same for Cb/Cr: pixel (x,y), x in [-4,11], y in [-1,7] at rec_c[p][(y+1)*16 + x + 4]
typedef struct WhMbLds {
  uint8_t  enc_y[256];
  uint8_t  enc_c[128];        // Cb 8x8 then Cr 8x8, stride 8
  uint8_t  rec_y[17 * 32];
  uint8_t  rec_c[2][9 * 16];
  uint8_t  pred_y[256];       // stride 16
  uint8_t  pred_c[128];       // Cb then Cr, stride 8
  uint8_t  pred4[9 * 16];     // candidate 4x4 predictions, [mode][y*4+x]
  int16_t  res[384];          // residual coefficients: luma blk*16 (luma4x4BlkIdx order), Cb 256.., Cr 320..
  int16_t  tmp[256];          // transform scratch: 16 blocks of coefficients
  int32_t  part[64];          // reduction partials
  int32_t  part2[16];         // per-block words of wh_quant_blocks; 64 bytes of table (Intra4x4 edge filters, neighbour counts)
  int16_t  dc[16];
  int16_t  cdc[8];
  int8_t   i4m[25];           // neighbour Intra4x4PredMode cache: [(by+1)*5 + bx+1]
  uint8_t  nzc[24];
  alignas (8) int16_t lv_luma[256];   // zig-zag levels per luma4x4BlkIdx (copied out as 8-byte words)
  int16_t  lv_dc[16];
  alignas (8) int16_t lv_cac[128];    // chroma AC levels Cb 0..3, Cr 4..7
  int16_t  lv_cdc[8];
  int8_t   i4_rem[16];
  uint16_t i4_prev;
  uint8_t  pad_[2];
#if defined(WH_PROF)
  uint32_t prof[32];          // phase-profiling accumulators of this wave (WH_PROF_MARK; profiling build only)
#endif
} WhMbLds;

#define WH_RY(S, x, y) ((S).rec_y[((y) + 1) * 32 + (x) + 8])
#define WH_RC(S, p, x, y) ((S).rec_c[p][((y) + 1) * 16 + (x) + 4])

// position class (0/1/2) helpers for quant tables
// The table row is wave-uniform (QP) and only the position class varies per lane: three uniform loads (scalar cache,
// lgkmcnt) + selects instead of a per-lane gather from global memory (vmcnt, which would also serialise behind any
// LDS-DMA prefetch in flight).
// (by the parities of the position's row and column, not by the class number: wherever one of the two is a compile-time constant after unrolling -- a lane's four
//  coefficients of a row, or of a column -- the lookup is ONE v_cndmask on a compare shared by all of the lane's lookups; by class number it was two compares, two selects
//  and the class arithmetic per coefficient, more than the quantisation itself: round 6)
WH_FN int wh_sel3_rc (int a, int b, int d, bool row_odd, bool col_odd) { return row_odd ? (col_odd ? d : b) : (col_odd ? b : a); }
WH_FN int wh_sel3 (int a, int b, int d, int pos) { return wh_sel3_rc (a, b, d, ((pos >> 2) & 1) != 0, (pos & 1) != 0); }
WH_FN int wh_mf (int qp, int pos) { return wh_sel3 (kWhQuantMF[qp * 3], kWhQuantMF[qp * 3 + 1], kWhQuantMF[qp * 3 + 2], pos); }
WH_FN int wh_ff_row (int ffrow, int pos) { return wh_sel3 (kWhQuantFF[ffrow * 3], kWhQuantFF[ffrow * 3 + 1], kWhQuantFF[ffrow * 3 + 2], pos); }
WH_FN int wh_ff_intra (int qp, int pos) { return wh_ff_row (qp + 6, pos); }
WH_FN int wh_ff_inter (int qp, int pos) { return wh_ff_row (qp, pos); }
WH_FN int wh_dq (int qp, int pos) { return wh_sel3 (kWhDequant[qp * 3], kWhDequant[qp * 3 + 1], kWhDequant[qp * 3 + 2], pos); }
// position (row r, column c) of a 4x4 block with r, c in 0..3 given separately (a caller whose lane index is only known to be < 4 by a guard)
WH_FN int wh_mf_rc (int qp, int r, int c) { return wh_sel3_rc (kWhQuantMF[qp * 3], kWhQuantMF[qp * 3 + 1], kWhQuantMF[qp * 3 + 2], (r & 1) != 0, (c & 1) != 0); }
WH_FN int wh_ff_intra_rc (int qp, int r, int c) { return wh_sel3_rc (kWhQuantFF[(qp + 6) * 3], kWhQuantFF[(qp + 6) * 3 + 1], kWhQuantFF[(qp + 6) * 3 + 2], (r & 1) != 0, (c & 1) != 0); }
WH_FN int wh_dq_rc (int qp, int r, int c) { return wh_sel3_rc (kWhDequant[qp * 3], kWhDequant[qp * 3 + 1], kWhDequant[qp * 3 + 2], (r & 1) != 0, (c & 1) != 0); }

// ---- forward DCT of N 4x4 blocks: res[blk*16 + r*4 + c] = T(enc - pred) -------------------------
// `nblk` blocks; block b covers enc rows/cols given by (ex[b],ey[b]) through the callbacks below.
// Implemented for the three tile shapes we need.
//
// luma 16x16: block index = luma4x4BlkIdx, enc = S.enc_y (stride 16), pred = S.pred_y (stride 16)
WH_FN void wh_dct_luma16 (WhMbLds& S) {
  WV_LANES_BEGIN (lane)
  const int b = lane >> 2, r = lane & 3;
  const int px = wh_blk_x (b) * 4, py = wh_blk_y (b) * 4 + r;
  const uint8_t* e = &S.enc_y[py * 16 + px];
  const uint8_t* p = &S.pred_y[py * 16 + px];
  int16_t o0, o1, o2, o3;
  wh_fdct4 (e[0] - p[0], e[1] - p[1], e[2] - p[2], e[3] - p[3], &o0, &o1, &o2, &o3);
  int16_t* t = &S.tmp[b * 16 + r * 4];
  t[0] = o0; t[1] = o1; t[2] = o2; t[3] = o3;
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  const int b = lane >> 2, c = lane & 3;
  const int16_t* t = &S.tmp[b * 16 + c];
  int16_t o0, o1, o2, o3;
  wh_fdct4 (t[0], t[4], t[8], t[12], &o0, &o1, &o2, &o3);
  int16_t* o = &S.res[b * 16 + c];
  o[0] = o0; o[4] = o1; o[8] = o2; o[12] = o3;
  WV_LANES_END
}

// chroma: 8 blocks (Cb 0..3, Cr 4..7 raster 2x2), enc = S.enc_c, pred = S.pred_c, out res[256 + blk*16]
WH_FN void wh_dct_chroma (WhMbLds& S) {
  WV_LANES_BEGIN (lane)
  if (lane < 32) {
    const int b = lane >> 2, r = lane & 3;
    const int pl = b >> 2, bb = b & 3;
    const int px = (bb & 1) * 4, py = (bb >> 1) * 4 + r;
    const uint8_t* e = &S.enc_c[pl * 64 + py * 8 + px];
    const uint8_t* p = &S.pred_c[pl * 64 + py * 8 + px];
    int16_t o0, o1, o2, o3;
    wh_fdct4 (e[0] - p[0], e[1] - p[1], e[2] - p[2], e[3] - p[3], &o0, &o1, &o2, &o3);
    int16_t* t = &S.tmp[b * 16 + r * 4];
    t[0] = o0; t[1] = o1; t[2] = o2; t[3] = o3;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < 32) {
    const int b = lane >> 2, c = lane & 3;
    const int16_t* t = &S.tmp[b * 16 + c];
    int16_t o0, o1, o2, o3;
    wh_fdct4 (t[0], t[4], t[8], t[12], &o0, &o1, &o2, &o3);
    int16_t* o = &S.res[256 + b * 16 + c];
    o[0] = o0; o[4] = o1; o[8] = o2; o[12] = o3;
  }
  WV_LANES_END
}

// ---- inverse transform + reconstruction of the 16 luma blocks from S.res into the rec tile ------
// rec = clip(pred + idct(res)); pred = S.pred_y
WH_FN void wh_idct_luma16 (WhMbLds& S) {
  WV_LANES_BEGIN (lane)
  const int b = lane >> 2, r = lane & 3;
  const int16_t* c = &S.res[b * 16 + r * 4];
  int16_t t0, t1, t2, t3;
  wh_idct4_h (c[0], c[1], c[2], c[3], &t0, &t1, &t2, &t3);
  int16_t* t = &S.tmp[b * 16 + r * 4];
  t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3;
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  const int b = lane >> 2, c = lane & 3;
  const int16_t* t = &S.tmp[b * 16 + c];
  int r0, r1, r2, r3;
  wh_idct4_v (t[0], t[4], t[8], t[12], &r0, &r1, &r2, &r3);
  const int px = wh_blk_x (b) * 4 + c, py = wh_blk_y (b) * 4;
  WH_RY (S, px, py + 0) = wh_clip255 (S.pred_y[(py + 0) * 16 + px] + r0);
  WH_RY (S, px, py + 1) = wh_clip255 (S.pred_y[(py + 1) * 16 + px] + r1);
  WH_RY (S, px, py + 2) = wh_clip255 (S.pred_y[(py + 2) * 16 + px] + r2);
  WH_RY (S, px, py + 3) = wh_clip255 (S.pred_y[(py + 3) * 16 + px] + r3);
  WV_LANES_END
}

WH_FN void wh_idct_chroma (WhMbLds& S) {
  WV_LANES_BEGIN (lane)
  if (lane < 32) {
    const int b = lane >> 2, r = lane & 3;
    const int16_t* c = &S.res[256 + b * 16 + r * 4];
    int16_t t0, t1, t2, t3;
    wh_idct4_h (c[0], c[1], c[2], c[3], &t0, &t1, &t2, &t3);
    int16_t* t = &S.tmp[b * 16 + r * 4];
    t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < 32) {
    const int b = lane >> 2, c = lane & 3;
    const int pl = b >> 2, bb = b & 3;
    const int16_t* t = &S.tmp[b * 16 + c];
    int r0, r1, r2, r3;
    wh_idct4_v (t[0], t[4], t[8], t[12], &r0, &r1, &r2, &r3);
    const int px = (bb & 1) * 4 + c, py = (bb >> 1) * 4;
    const uint8_t* p = &S.pred_c[pl * 64];
    WH_RC (S, pl, px, py + 0) = wh_clip255 (p[(py + 0) * 8 + px] + r0);
    WH_RC (S, pl, px, py + 1) = wh_clip255 (p[(py + 1) * 8 + px] + r1);
    WH_RC (S, pl, px, py + 2) = wh_clip255 (p[(py + 2) * 8 + px] + r2);
    WH_RC (S, pl, px, py + 3) = wh_clip255 (p[(py + 3) * 8 + px] + r3);
  }
  WV_LANES_END
}

// ---- cost functions over the full MB tiles -------------------------------------------------------
// SAD / SATD of enc_y vs pred_y (sad_common.cpp:44-80, sample.cpp:47-156); SATD rounds per 4x4.
WH_FN int wh_cost_luma16 (WhMbLds& S, int use_satd) {
  int cost;
  if (!use_satd) {
    WV_SUM (cost, lane, (wh_abs (S.enc_y[lane * 4 + 0] - S.pred_y[lane * 4 + 0]) + wh_abs (S.enc_y[lane * 4 + 1] - S.pred_y[lane * 4 + 1]) +
                          wh_abs (S.enc_y[lane * 4 + 2] - S.pred_y[lane * 4 + 2]) + wh_abs (S.enc_y[lane * 4 + 3] - S.pred_y[lane * 4 + 3])));
    return cost;
  }
  WV_LANES_BEGIN (lane)
  const int b = lane >> 2, r = lane & 3;                     // raster 4x4 block here
  const int px = (b & 3) * 4, py = (b >> 2) * 4 + r;
  const uint8_t* e = &S.enc_y[py * 16 + px];
  const uint8_t* p = &S.pred_y[py * 16 + px];
  int o0, o1, o2, o3;
  wh_had4 (e[0] - p[0], e[1] - p[1], e[2] - p[2], e[3] - p[3], &o0, &o1, &o2, &o3);
  int16_t* t = &S.tmp[b * 16 + r * 4];
  t[0] = (int16_t)o0; t[1] = (int16_t)o1; t[2] = (int16_t)o2; t[3] = (int16_t)o3;
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  const int b = lane >> 2, c = lane & 3;
  const int16_t* t = &S.tmp[b * 16 + c];
  int o0, o1, o2, o3;
  wh_had4 (t[0], t[4], t[8], t[12], &o0, &o1, &o2, &o3);
  S.part[lane] = wh_abs (o0) + wh_abs (o1) + wh_abs (o2) + wh_abs (o3);
  WV_LANES_END
  WV_SUM (cost, lane, (lane < 16 ? ((S.part[lane * 4] + S.part[lane * 4 + 1] + S.part[lane * 4 + 2] + S.part[lane * 4 + 3] + 1) >> 1) : 0));
  return cost;
}

// chroma 8x8 Cb + Cr cost (enc_c vs pred_c)
WH_FN int wh_cost_chroma (WhMbLds& S, int use_satd) {
  int cost;
  if (!use_satd) {
    WV_SUM (cost, lane, (lane < 32 ? (wh_abs (S.enc_c[lane * 4 + 0] - S.pred_c[lane * 4 + 0]) + wh_abs (S.enc_c[lane * 4 + 1] - S.pred_c[lane * 4 + 1]) +
                                        wh_abs (S.enc_c[lane * 4 + 2] - S.pred_c[lane * 4 + 2]) + wh_abs (S.enc_c[lane * 4 + 3] - S.pred_c[lane * 4 + 3])) : 0));
    return cost;
  }
  WV_LANES_BEGIN (lane)
  if (lane < 32) {
    const int b = lane >> 2, r = lane & 3;
    const int pl = b >> 2, bb = b & 3;
    const int px = (bb & 1) * 4, py = (bb >> 1) * 4 + r;
    const uint8_t* e = &S.enc_c[pl * 64 + py * 8 + px];
    const uint8_t* p = &S.pred_c[pl * 64 + py * 8 + px];
    int o0, o1, o2, o3;
    wh_had4 (e[0] - p[0], e[1] - p[1], e[2] - p[2], e[3] - p[3], &o0, &o1, &o2, &o3);
    int16_t* t = &S.tmp[b * 16 + r * 4];
    t[0] = (int16_t)o0; t[1] = (int16_t)o1; t[2] = (int16_t)o2; t[3] = (int16_t)o3;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < 32) {
    const int b = lane >> 2, c = lane & 3;
    const int16_t* t = &S.tmp[b * 16 + c];
    int o0, o1, o2, o3;
    wh_had4 (t[0], t[4], t[8], t[12], &o0, &o1, &o2, &o3);
    S.part[lane] = wh_abs (o0) + wh_abs (o1) + wh_abs (o2) + wh_abs (o3);
  }
  WV_LANES_END
  WV_SUM (cost, lane, (lane < 8 ? ((S.part[lane * 4] + S.part[lane * 4 + 1] + S.part[lane * 4 + 2] + S.part[lane * 4 + 3] + 1) >> 1) : 0));
  return cost;
}

// ---- Intra16x16 luma residual pipeline (svc_encode_mb.cpp:54-137) -------------------------------
// in : enc_y, pred_y       out: lv_dc, lv_luma (AC), nzc[0..15], rec tile;  returns cbp luma (0 / 15)
WH_FN int wh_encrec_i16 (WhMbLds& S, int qp) {
  wh_dct_luma16 (S);
  // DC gather + 4x4 Hadamard (encode_mb_aux.cpp:280-311 WelsHadamardT4Dc_c): rows first
  WV_LANES_BEGIN (lane)
  if (lane < 4) {                       // raster row `lane` of block DCs
    int d[4];
    for (int x = 0; x < 4; ++x) {
      const int by = lane, bx = x;      // raster block (bx,by) -> luma4x4BlkIdx
      const int b = (bx & 1) | ((by & 1) << 1) | ((bx & 2) << 1) | ((by & 2) << 2);
      d[x] = S.res[b * 16];
    }
    const int s0 = d[0] + d[3], s3 = d[0] - d[3], s1 = d[1] + d[2], s2 = d[1] - d[2];
    int32_t* p = &S.part[lane * 4];
    p[0] = s0 + s1; p[2] = s0 - s1; p[1] = s3 + s2; p[3] = s3 - s2;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < 4) {                       // column `lane`
    const int32_t* p = &S.part[lane];
    const int s0 = p[0] + p[12], s3 = p[0] - p[12], s1 = p[4] + p[8], s2 = p[4] - p[8];
    const int ff = wh_ff_intra (qp, 0) << 1, mf = wh_mf (qp, 0) >> 1;   // pfQuantizationDc4x4 (aDctT4Dc, pFF[0]<<1, pMF[0]>>1)
    S.dc[lane]      = wh_quant1_t ((int16_t)wh_clip3 ((s0 + s1 + 1) >> 1, -32768, 32767), ff, mf);
    S.dc[lane + 8]  = wh_quant1_t ((int16_t)wh_clip3 ((s0 - s1 + 1) >> 1, -32768, 32767), ff, mf);
    S.dc[lane + 4]  = wh_quant1_t ((int16_t)wh_clip3 ((s3 + s2 + 1) >> 1, -32768, 32767), ff, mf);
    S.dc[lane + 12] = wh_quant1_t ((int16_t)wh_clip3 ((s3 - s2 + 1) >> 1, -32768, 32767), ff, mf);
  }
  WV_LANES_END
  // AC quant (WelsQuantFour4x4_c, intra FF) + scans + counts
  WV_LANES_BEGIN (lane)
  for (int k = 0; k < 4; ++k) {
    const int i = lane * 4 + k, pos = i & 15;
    S.res[i] = wh_quant1_t (S.res[i], wh_ff_intra (qp, pos), wh_mf (qp, pos));
  }
  if (lane < 16) S.lv_dc[lane] = S.dc[wh_zigzag (lane)];
  WV_LANES_END
  int nz_ac, nz_dc;
  WV_LANES_BEGIN (lane)
  {                                     // 4 zig-zag AC levels per lane: block lane>>2, k = (lane&3)*4..+3
    const int b = lane >> 2;
    int cnt = 0;
    for (int q = 0; q < 4; ++q) {
      const int k = (lane & 3) * 4 + q;
      const int16_t v = (k < 15) ? S.res[b * 16 + wh_zigzag (k + 1)] : (int16_t)0;
      S.lv_luma[b * 16 + k] = v;
      cnt += (v != 0);
    }
    S.part[lane] = cnt;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < 16) {
    const int n = S.part[lane * 4] + S.part[lane * 4 + 1] + S.part[lane * 4 + 2] + S.part[lane * 4 + 3];
    S.nzc[wh_blk_y (lane) * 4 + wh_blk_x (lane)] = (uint8_t)n;
    S.part2[lane] = n;
  }
  WV_LANES_END
  WV_SUM (nz_ac, lane, (lane < 16 ? S.part2[lane] : 0));
  WV_SUM (nz_dc, lane, (lane < 16 ? (S.lv_dc[lane] != 0) : 0));
  // DC dequant (decode_mb_aux.cpp:40-125)
  if (nz_dc > 0) {
    if (qp < 12) {
      // WelsIHadamard4x4Dc then WelsDequantLumaDc4x4
      WV_LANES_BEGIN (lane)
      if (lane < 4) {                   // row pass, in place (int16)
        int16_t* p = &S.dc[lane * 4];
        const int16_t t0 = (int16_t) (p[0] + p[2]), t1 = (int16_t) (p[0] - p[2]);
        const int16_t t2 = (int16_t) (p[1] - p[3]), t3 = (int16_t) (p[1] + p[3]);
        p[0] = (int16_t) (t0 + t3); p[1] = (int16_t) (t1 + t2); p[2] = (int16_t) (t1 - t2); p[3] = (int16_t) (t0 - t3);
      }
      WV_LANES_END
      WV_LANES_BEGIN (lane)
      if (lane < 4) {
        int16_t* p = &S.dc[lane];
        const int16_t t0 = (int16_t) (p[0] + p[8]), t1 = (int16_t) (p[0] - p[8]);
        const int16_t t2 = (int16_t) (p[4] - p[12]), t3 = (int16_t) (p[4] + p[12]);
        const int dqv = kWhDequant[(qp % 6) * 3 + 0];
        const int qf0 = qp / 6, qf1 = 2 - qf0, qf0s = 1 << (1 - qf0);
        p[0]  = (int16_t) (((int16_t) (t0 + t3) * dqv + qf0s) >> qf1);
        p[4]  = (int16_t) (((int16_t) (t1 + t2) * dqv + qf0s) >> qf1);
        p[8]  = (int16_t) (((int16_t) (t1 - t2) * dqv + qf0s) >> qf1);
        p[12] = (int16_t) (((int16_t) (t0 - t3) * dqv + qf0s) >> qf1);
      }
      WV_LANES_END
    } else {
      // WelsDequantIHadamard4x4_c (pRes, g_kuiDequantCoeff[qp][0] >> 2)
      WV_LANES_BEGIN (lane)
      if (lane < 4) {
        int16_t* p = &S.dc[lane * 4];
        const int16_t t0 = (int16_t) (p[0] + p[2]), t1 = (int16_t) (p[0] - p[2]);
        const int16_t t2 = (int16_t) (p[1] - p[3]), t3 = (int16_t) (p[1] + p[3]);
        p[0] = (int16_t) (t0 + t3); p[1] = (int16_t) (t1 + t2); p[2] = (int16_t) (t1 - t2); p[3] = (int16_t) (t0 - t3);
      }
      WV_LANES_END
      WV_LANES_BEGIN (lane)
      if (lane < 4) {
        int16_t* p = &S.dc[lane];
        const int16_t t0 = (int16_t) (p[0] + p[8]), t1 = (int16_t) (p[0] - p[8]);
        const int16_t t2 = (int16_t) (p[4] - p[12]), t3 = (int16_t) (p[4] + p[12]);
        const int mfq = wh_dq (qp, 0) >> 2;
        p[0]  = (int16_t) ((t0 + t3) * mfq);
        p[4]  = (int16_t) ((t1 + t2) * mfq);
        p[8]  = (int16_t) ((t1 - t2) * mfq);
        p[12] = (int16_t) ((t0 - t3) * mfq);
      }
      WV_LANES_END
    }
  }
  // AC dequant (WelsDequantFour4x4_c, int16 wrap) and DC insertion; when no AC survived the
  // quantised AC values are all zero already, so the generic inverse transform reproduces both the
  // DC-only (WelsIDctRecI16x16Dc_c) and the copy-prediction special cases of the reference.
  WV_LANES_BEGIN (lane)
  for (int k = 0; k < 4; ++k) {
    const int i = lane * 4 + k, pos = i & 15;
    if (pos == 0) {
      const int b = i >> 4;
      S.res[i] = S.dc[wh_blk_y (b) * 4 + wh_blk_x (b)];
    } else {
      S.res[i] = (int16_t) (S.res[i] * wh_dq (qp, pos));
    }
  }
  WV_LANES_END
  wh_idct_luma16 (S);
  return nz_ac > 0 ? 15 : 0;
}

// ---- Intra4x4 residual pipeline of one block (svc_encode_mb.cpp:139-178) ------------------------
// pred = S.pred4[mode_slot*16..], writes lv_luma[b], nzc, i4m / i4_rem of the block, the rec tile; returns nzc of the block.
// Four lane blocks: rows of the forward transform | columns + quantisation (the count of non-zero levels is a quad sum of the four
// lanes' shares, no 64-lane reduction) | zig-zag scan + dequantisation + rows of the inverse | columns of the inverse + reconstruction.
WH_FN int wh_encrec_i4 (WhMbLds& S, int b, int pred_slot, int qp, int8_t rem) {
  const int bx = wh_blk_x (b) * 4, by = wh_blk_y (b) * 4;
  WV_LANES_BEGIN (lane)
  if (lane < 4) {
    const uint8_t* e = &S.enc_y[(by + lane) * 16 + bx];
    const uint8_t* p = &S.pred4[pred_slot * 16 + lane * 4];
    int16_t o0, o1, o2, o3;
    wh_fdct4 (e[0] - p[0], e[1] - p[1], e[2] - p[2], e[3] - p[3], &o0, &o1, &o2, &o3);
    int16_t* t = &S.tmp[lane * 4];
    t[0] = o0; t[1] = o1; t[2] = o2; t[3] = o3;
  } else if (lane == 4) {
    S.i4m[((by >> 2) + 1) * 5 + (bx >> 2) + 1] = (int8_t)pred_slot;
    S.i4_rem[b] = rem;
  }
  WV_LANES_END
  int nz, nz1, nz2, nz3;
  WV_QUADSUM4 (nz, nz1, nz2, nz3, lane, (lane < 4 ? ([&] () {
    const int16_t* t = &S.tmp[lane];
    int16_t o[4];
    wh_fdct4 (t[0], t[4], t[8], t[12], &o[0], &o[1], &o[2], &o[3]);
    int cnt = 0;
    for (int k = 0; k < 4; ++k) {
      const int pos = k * 4 + lane;
      const int16_t q = wh_quant1_t (o[k], wh_ff_intra_rc (qp, k, lane), wh_mf_rc (qp, k, lane));
      S.res[b * 16 + pos] = q;
      cnt += q != 0;
    }
    return cnt; }) () : 0));
  (void)nz1; (void)nz2; (void)nz3;
  WV_SYNC();
  WV_LANES_BEGIN (lane)
  if (lane < 16) S.lv_luma[b * 16 + lane] = S.res[b * 16 + wh_zigzag (lane)];
  if (lane == 16) S.nzc[(by >> 2) * 4 + (bx >> 2)] = (uint8_t)nz;
  if (nz > 0) {
    if (lane < 4) {                     // dequant (WelsDequant4x4_c, int16 wrap) + horizontal inverse
      const int16_t* c = &S.res[b * 16 + lane * 4];
      int16_t t0, t1, t2, t3;
      wh_idct4_h ((int16_t) (c[0] * wh_dq_rc (qp, lane, 0)), (int16_t) (c[1] * wh_dq_rc (qp, lane, 1)),
                  (int16_t) (c[2] * wh_dq_rc (qp, lane, 2)), (int16_t) (c[3] * wh_dq_rc (qp, lane, 3)), &t0, &t1, &t2, &t3);
      int16_t* t = &S.tmp[lane * 4];
      t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3;
    }
  } else if (lane >= 32 && lane < 48) {
    const int l = lane - 32;
    WH_RY (S, bx + (l & 3), by + (l >> 2)) = S.pred4[pred_slot * 16 + l];
  }
  WV_LANES_END
  if (nz > 0) {
    WV_LANES_BEGIN (lane)
    if (lane < 4) {
      const int16_t* t = &S.tmp[lane];
      int r0, r1, r2, r3;
      wh_idct4_v (t[0], t[4], t[8], t[12], &r0, &r1, &r2, &r3);
      const uint8_t* p = &S.pred4[pred_slot * 16 + lane];
      WH_RY (S, bx + lane, by + 0) = wh_clip255 (p[0] + r0);
      WH_RY (S, bx + lane, by + 1) = wh_clip255 (p[4] + r1);
      WH_RY (S, bx + lane, by + 2) = wh_clip255 (p[8] + r2);
      WH_RY (S, bx + lane, by + 3) = wh_clip255 (p[12] + r3);
    }
    WV_LANES_END
  }
  return nz;
}

// The same for TWO blocks at once -- bA on lanes 0..31, bB on lanes 32..63 (intra_mb.h: the blocks of one 2:1 diagonal of the 4x4 grid are independent) --
// with the lane roles of wh_encrec_i4 inside each half.  predB: the second block's candidate predictions (the first one's are S.pred4); the transform
// scratch is S.tmp[0..15] for the first block, S.tmp[16..31] for the second.
WH_FN void wh_encrec_i4_pair (WhMbLds& S, int bA, int bB, int slotA, int slotB, int qp, int8_t remA, int8_t remB, const uint8_t* predB, int* nzA, int* nzB) {
#define WH_I4P_HALF const int h = lane >> 5, ll = lane & 31, b = h ? bB : bA, slot = h ? slotB : slotA, bx = wh_blk_x (b) * 4, by = wh_blk_y (b) * 4; const uint8_t* pr = h ? predB : S.pred4
  WV_LANES_BEGIN (lane)
  {
    WH_I4P_HALF;
    if (ll < 4) {
      const uint8_t* e = &S.enc_y[(by + ll) * 16 + bx];
      const uint8_t* p = &pr[slot * 16 + ll * 4];
      int16_t o0, o1, o2, o3;
      wh_fdct4 (e[0] - p[0], e[1] - p[1], e[2] - p[2], e[3] - p[3], &o0, &o1, &o2, &o3);
      int16_t* t = &S.tmp[h * 16 + ll * 4];
      t[0] = o0; t[1] = o1; t[2] = o2; t[3] = o3;
    } else if (ll == 4) {
      S.i4m[((by >> 2) + 1) * 5 + (bx >> 2) + 1] = (int8_t)slot;
      S.i4_rem[b] = h ? remB : remA;
    }
  }
  WV_LANES_END
  WvLaneArr nzt;
#if defined(WH_EMU)
  memset (&nzt, 0, sizeof (nzt));
#else
  nzt = 0;
#endif
  WV_QUADSUM_TAB (nzt, lane, ((lane & 31) < 4 ? ([&] () {
    const int h = lane >> 5, ll = lane & 31, b = h ? bB : bA;
    const int16_t* t = &S.tmp[h * 16 + ll];
    int16_t o[4];
    wh_fdct4 (t[0], t[4], t[8], t[12], &o[0], &o[1], &o[2], &o[3]);
    int cnt = 0;
    for (int k = 0; k < 4; ++k) {
      const int pos = k * 4 + ll;
      const int16_t q = wh_quant1_t (o[k], wh_ff_intra_rc (qp, k, ll), wh_mf_rc (qp, k, ll));
      S.res[b * 16 + pos] = q;
      cnt += q != 0;
    }
    return cnt; }) () : 0));
  WV_SYNC();
  const int nz0 = WV_LGET (nzt, 0), nz1 = WV_LGET (nzt, 32);
  *nzA = nz0; *nzB = nz1;
  WV_LANES_BEGIN (lane)
  {
    WH_I4P_HALF;
    const int nz = h ? nz1 : nz0;
    if (ll < 16) S.lv_luma[b * 16 + ll] = S.res[b * 16 + wh_zigzag (ll)];
    if (ll == 16) S.nzc[(by >> 2) * 4 + (bx >> 2)] = (uint8_t)nz;
    if (nz > 0) {
      if (ll < 4) {                     // dequant (WelsDequant4x4_c, int16 wrap) + horizontal inverse
        const int16_t* c = &S.res[b * 16 + ll * 4];
        int16_t t0, t1, t2, t3;
        wh_idct4_h ((int16_t) (c[0] * wh_dq_rc (qp, ll, 0)), (int16_t) (c[1] * wh_dq_rc (qp, ll, 1)),
                    (int16_t) (c[2] * wh_dq_rc (qp, ll, 2)), (int16_t) (c[3] * wh_dq_rc (qp, ll, 3)), &t0, &t1, &t2, &t3);
        int16_t* t = &S.tmp[h * 16 + ll * 4];
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3;
      }
    } else if (ll >= 16) {
      const int l = ll - 16;
      WH_RY (S, bx + (l & 3), by + (l >> 2)) = pr[slot * 16 + l];
    }
  }
  WV_LANES_END
  if (nz0 > 0 || nz1 > 0) {
    WV_LANES_BEGIN (lane)
    {
      WH_I4P_HALF;
      if (ll < 4 && (h ? nz1 : nz0) > 0) {
        const int16_t* t = &S.tmp[h * 16 + ll];
        int r0, r1, r2, r3;
        wh_idct4_v (t[0], t[4], t[8], t[12], &r0, &r1, &r2, &r3);
        const uint8_t* p = &pr[slot * 16 + ll];
        WH_RY (S, bx + ll, by + 0) = wh_clip255 (p[0] + r0);
        WH_RY (S, bx + ll, by + 1) = wh_clip255 (p[4] + r1);
        WH_RY (S, bx + ll, by + 2) = wh_clip255 (p[8] + r2);
        WH_RY (S, bx + ll, by + 3) = wh_clip255 (p[12] + r3);
      }
    }
    WV_LANES_END
  }
#undef WH_I4P_HALF
}

// JVT-O079 "single coefficient" score of a block whose levels are all +-1 (WelsGetNoneZeroCount / the run table of svc_encode_mb.cpp:
// 3 for a coefficient right behind the previous one in zig-zag order, 2 after a run of one or two zeros, 1 after three to five, 0
// beyond), from the mask of non-zero zig-zag positions.  Bit-parallel: a coefficient scores [run < 1] + [run < 3] + [run < 6], and
// "run < d" says that one of the d positions below it is occupied -- positions below zero count as occupied (the run before the
// first coefficient is its index).
WH_FN int wh_single_ctr_mask (unsigned m) {
  const unsigned R = m << 6, M = R | 63u;
  const unsigned a1 = M << 1, a3 = a1 | (M << 2) | (M << 3), a6 = a3 | (M << 4) | (M << 5) | (M << 6);
  return __builtin_popcount (R & a1) + __builtin_popcount (R & a3) + __builtin_popcount (R & a6);
}
// What wh_quant_blocks leaves per block in S.part2[blk]: the mask of non-zero zig-zag positions, the block's score and two flags
#define WH_Q_MASK(w) ((unsigned)(w) & 0xffffu)                 /* bit k: zig-zag position k + skip_dc holds a level */
#define WH_Q_SCORE(w) ((int)(((unsigned)(w) >> 16) & 0xffu))   /* 9 when a level exceeds 1, else the single-coefficient score of the mask */
#define WH_Q_BIG(w) ((int)(((unsigned)(w) >> 24) & 1u))        /* some |level| > 1 */
#define WH_Q_ANY(w) ((int)(((unsigned)(w) >> 25) & 1u))        /* some level != 0 */
// quantise S.res[base + 0 .. 16*nblk) into `dst` (rounding offsets of table row `ffrow`: qp for inter, qp + 6 for intra); S.part2[blk] = the
// block's WH_Q_* word (`skip_dc`: position 0 is not part of the scan).  Lane 4 b + q quantises the raster positions 4 q .. 4 q + 3 of
// block b and leaves its share of the mask (its four positions' zig-zag indices) and its largest |level| in S.part; lane b gathers.
WH_FN void wh_quant_blocks (WhMbLds& S, int base, int nblk, int qp, int ffrow, int16_t* dst, int skip_dc) {
  WV_LANES_BEGIN (lane)
  if (lane < nblk * 4) {
    int16_t mx = 0;
    unsigned pm = 0;
    // zig-zag index of raster position p, a nibble each: 0 1 5 6 | 2 4 7 12 | 3 8 11 13 | 9 10 14 15
    const unsigned inv4 = (unsigned) (0xFEA9DB83C7426510ULL >> (16 * (lane & 3))) & 0xffffu;
    for (int k = 0; k < 4; ++k) {
      const int i = lane * 4 + k, pos = i & 15;
      int16_t a;
      dst[i] = wh_quant1_abs_t (S.res[base + i], wh_ff_row (ffrow, pos), wh_mf (qp, pos), &a);
      if (mx < a) mx = a;
      pm |= (unsigned) (a != 0) << ((inv4 >> (4 * k)) & 15u);
    }
    S.part[lane] = (int32_t) (pm | ((unsigned) (uint16_t)mx << 16));
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < nblk) {
    const unsigned w0 = (unsigned)S.part[lane * 4], w1 = (unsigned)S.part[lane * 4 + 1], w2 = (unsigned)S.part[lane * 4 + 2], w3 = (unsigned)S.part[lane * 4 + 3];
    const unsigned m = ((w0 | w1 | w2 | w3) & 0xffffu) >> skip_dc;
    const int mx = wh_max (wh_max ((int) (w0 >> 16), (int) (w1 >> 16)), wh_max ((int) (w2 >> 16), (int) (w3 >> 16)));      // (|level| as uint16: 0 .. 32768)
    const unsigned big = mx > 1, any = mx != 0;
    const unsigned score = big ? 9u : (unsigned)wh_single_ctr_mask (m);      // (no level at all: empty mask, score 0)
    S.part2[lane] = (int32_t) (m | (score << 16) | (big << 24) | (any << 25));
  }
  WV_LANES_END
}

// ---- chroma residual pipeline for both planes (svc_encode_mb.cpp:244-312 WelsEncRecUV) -----------
// in: enc_c, pred_c.  out: lv_cdc, lv_cac, nzc[16..23], S.res[256..383] ready for wh_idct_chroma.  Returns chroma cbp (0,1,2).
WH_FN int wh_encrec_chroma (WhMbLds& S, int qpc, int is_intra) {
  wh_dct_chroma (S);
  const int ffrow = is_intra ? qpc + 6 : qpc;
  // 2x2 DC Hadamard + quant per plane (encode_mb_aux.cpp:247-277 WelsHadamardQuant2x2_c)
  WV_LANES_BEGIN (lane)
  if (lane < 2) {
    int16_t* r = &S.res[256 + lane * 64];
    const int16_t s0 = (int16_t) (r[0] + r[32]), s1 = (int16_t) (r[0] - r[32]);
    const int16_t s2 = (int16_t) (r[16] + r[48]), s3 = (int16_t) (r[16] - r[48]);
    r[0] = 0; r[16] = 0; r[32] = 0; r[48] = 0;
    const int ff = kWhQuantFF[ffrow * 3 + 0] << 1, mf = kWhQuantMF[qpc * 3 + 0] >> 1;
    int16_t* d = &S.cdc[lane * 4];
    d[0] = wh_quant1_t ((int16_t) (s0 + s2), ff, mf);
    d[1] = wh_quant1_t ((int16_t) (s0 - s2), ff, mf);
    d[2] = wh_quant1_t ((int16_t) (s1 + s3), ff, mf);
    d[3] = wh_quant1_t ((int16_t) (s1 - s3), ff, mf);
    for (int k = 0; k < 4; ++k) S.lv_cdc[lane * 4 + k] = d[k];
  }
  WV_LANES_END
  // AC quant with per-block max and non-zero mask (WelsQuantFour4x4Max_c)
  wh_quant_blocks (S, 256, 8, qpc, ffrow, &S.res[256], 1);
  // keep a plane's AC when its JVT-O079 score reaches 7 (inter; the reference stops adding at 7, which cannot change
  // the test) -- intra keeps every plane that has a non-zero level
  int sc0, sc1, nzdc0, nzdc1;          // (four quad sums in one pass: lanes 0..7 the blocks' scores per plane, lanes 8..15 the planes' DC levels)
#define WH_CSCORE(l) (is_intra ? (WH_Q_ANY (S.part2[l]) ? 7 : 0) : WH_Q_SCORE (S.part2[l]))
  WV_QUADSUM4 (sc0, sc1, nzdc0, nzdc1, lane, (lane < 8 ? WH_CSCORE (lane & 7) : lane < 16 ? (int) (S.cdc[lane & 7] != 0) : 0));
#undef WH_CSCORE
  const int keep0 = sc0 >= 7, keep1 = sc1 >= 7;
  WV_LANES_BEGIN (lane)
  {
    if (lane < 32) {                      // scans (WelsScan4x4Ac_c) -- a dropped plane leaves stale levels behind, nzc = 0 says so
      const int b = lane >> 2;
      for (int q = 0; q < 4; ++q) {
        const int k = (lane & 3) * 4 + q;
        S.lv_cac[b * 16 + k] = (k < 15) ? S.res[256 + b * 16 + wh_zigzag (k + 1)] : (int16_t)0;
      }
    }
    if (lane < 8) {
      int n = 0;
      if (lane < 4 ? keep0 : keep1) n = __builtin_popcount (WH_Q_MASK (S.part2[lane]));
      S.nzc[16 + lane] = (uint8_t)n;
    }
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  for (int h = 0; h < 2; ++h) {           // dequant (or clear) the 128 AC coefficients, two per lane
    const int i = lane + 64 * h, keep = h == 0 ? keep0 : keep1;
    S.res[256 + i] = keep ? (int16_t) (S.res[256 + i] * wh_dq (qpc, i & 15)) : (int16_t)0;
  }
  WV_LANES_END
  if (nzdc0 > 0 || nzdc1 > 0) {
    // WelsDequantIHadamard2x2Dc (decode_mb_aux.cpp:127-137)
    WV_LANES_BEGIN (lane)
    if (lane < 2 && (lane == 0 ? nzdc0 : nzdc1) > 0) {
      int16_t* d = &S.cdc[lane * 4];
      const int16_t su = (int16_t) (d[0] + d[2]), du = (int16_t) (d[0] - d[2]);
      const int16_t sd = (int16_t) (d[1] + d[3]), dd = (int16_t) (d[1] - d[3]);
      const int mf = wh_dq (qpc, 0);
      int16_t* r = &S.res[256 + lane * 64];
      r[0]  = (int16_t) (((su + sd) * mf) >> 1);
      r[16] = (int16_t) (((su - sd) * mf) >> 1);
      r[32] = (int16_t) (((du + dd) * mf) >> 1);
      r[48] = (int16_t) (((du - dd) * mf) >> 1);
    }
    WV_LANES_END
  }
  return (keep0 || keep1) ? 2 : (nzdc0 > 0 || nzdc1 > 0) ? 1 : 0;
}
