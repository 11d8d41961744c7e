// intra_mb.h -- mode decision + reconstruction of one intra macroblock by one wavefront.
//
// Reference behaviour restated (codec/encoder/core/src):
//   svc_base_layer_md.cpp:956-962   WelsMdIntraMb
//   svc_base_layer_md.cpp:365-417   WelsMdI16x16            (generic loop: the C build has no Combined3)
//   svc_base_layer_md.cpp:418-546   WelsMdI4x4              (complexity >= MEDIUM)
//   svc_base_layer_md.cpp:548-865   WelsMdI4x4Fast          (complexity LOW, gated by md.cpp:435-475,497-503)
//   svc_base_layer_md.cpp:867-930   WelsMdIntraChroma
//   svc_base_layer_md.cpp:2023-2039 WelsMdIntraSecondaryModesEnc
//   get_intra_predictor.cpp:79-613, common/src/intra_pred_common.cpp:47-77  (predictors)
#pragma once
#include "mb_common.h"

// ---- neighbour availability (svc_encode_slice.cpp:138-174 UpdateMbNeighbor: same slice only) ----
#define WH_AV_LEFT 1
#define WH_AV_TOP 2
#define WH_AV_TOPLEFT 4
#define WH_AV_TOPRIGHT 8

WH_FN int wh_slice_of_mb (const WhSeqParams& P, int mbxy) {
  int s = 0;
  for (int i = 1; i < P.num_slices; ++i) s += (mbxy >= P.slice_first_mb[i]);
  return s;
}
// every neighbour has a smaller MB address, so "same slice" == "not before the first MB of this slice"
WH_FN int wh_mb_avail_in_slice (const WhSeqParams& P, int mbx, int mby, int slice_first) {
  const int w = P.mb_w, xy = mby * w + mbx;
  int av = 0;
  if (mbx > 0 && xy - 1 >= slice_first) av |= WH_AV_LEFT;
  if (mby > 0) {
    if (xy - w >= slice_first) av |= WH_AV_TOP;
    if (mbx > 0 && xy - w - 1 >= slice_first) av |= WH_AV_TOPLEFT;
    if (mbx < w - 1 && xy - w + 1 >= slice_first) av |= WH_AV_TOPRIGHT;
  }
  return av;
}
WH_FN int wh_mb_avail (const WhSeqParams& P, int mbx, int mby) {
  return wh_mb_avail_in_slice (P, mbx, mby, P.slice_first_mb[wh_slice_of_mb (P, mby * P.mb_w + mbx)]);
}

// ---- load source MB + reconstructed neighbours into the LDS tile ---------------------------------
// Split into a fetch half (global loads into registers) and a commit half (LDS stores) so that a caller can put
// further loads between the two and pay the HBM latency once.
typedef struct WhTileRegs { uint32_t y, c, nb; } WhTileRegs;

// source samples of the MB (never written on the device: may be fetched long before the MB is processed)
WH_FN void wh_tile_fetch_src (int lane, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, WhTileRegs* r) {
  // (macroblock-tiled source picture, WH_SRC_*: lane = (row, 4-sample segment) of the luma block = its byte order; the same for chroma)
  r->y = * (const WH_G uint32_t*) ((const WH_G uint8_t*)J.src[0] + WH_SRC_Y_OFF (P.mb_w, mbx, mby, 0, 0) + lane * 4);
  r->c = 0;
  if (lane < 32) r->c = * (const WH_G uint32_t*) ((const WH_G uint8_t*)J.src[0] + WH_SRC_C_OFF (P.mb_w, mbx, mby, 0, 0, 0) + lane * 4);
}
// reconstructed neighbour samples (written by the neighbour MBs: only after they are done)
// (the neighbour loads of a macroblock as ONE batch: measured on the MI355X in round 3, MD launch 13.65 -> 13.18 ms against one load per role)
// One 32-bit load per lane whatever its role, at an address that is valid for every lane, and a select afterwards.  With one
// load per role inside `if / else if` the compiler merges the results through a phi and waits for each load at the end of its
// branch: the macroblock then pays one L2 round trip per role (three to four in a row) instead of one.  The column lanes read
// the aligned word that ENDS with their sample (x = -4 .. -1) and keep its top byte; lanes without a role repeat lane 0's word.
// (the three planes as VALUES: a select between loads of job fields would be folded into one load at a selected offset, which
// keeps a job descriptor that lives in registers from staying there)
// (round 6: both layouts only compute the ADDRESS of the lane's word; the one load follows the uniform choice between them, and the
//  column lanes' shift waits until the commit.  With a load in each branch the compiler merged the loaded VALUES through a phi and waited
//  for the load at the end of its branch -- the P kernel then paid that L2 round trip before it had even issued the loads of the
//  neighbours' states: "batch-1: loads issued" 3.2 k of a macroblock's 47 k cycles, profiles/r06_p1080p_detail_phase_cycles_before.txt)
WH_FN const WH_G uint8_t* wh_tile_nb_addr_planes (int lane, const WhSeqParams& P, const WH_G uint8_t* rec0, const WH_G uint8_t* rec1, const WH_G uint8_t* rec2, int mbx, int mby) {
  const bool top_y = lane < 7, col_y = lane >= 16 && lane < 32, top_c = lane >= 32 && lane < 38, col_c = lane >= 48;
  const bool luma = !(top_c || col_c);                                   // (idle lanes take lane 0's role)
  const int pl = top_c ? (lane - 32) / 3 : (lane - 48) >> 3;             // chroma plane of the chroma roles
  const int row = col_y ? mby * 16 + (lane - 16) : col_c ? mby * 8 + (lane & 7) : top_c ? mby * 8 - 1 : mby * 16 - 1;
  const int x = top_y ? mbx * 16 + lane * 4 - 4 : top_c ? mbx * 8 + ((lane - 32) % 3) * 4 - 4 : luma ? mbx * 16 - 4 : mbx * 8 - 4;
  const WH_G uint8_t* base = luma ? rec0 : (pl & 1) ? rec2 : rec1;
  int sy = P.rec_stride_y, sc = P.rec_stride_c;
  WH_UNIFORM_VALUE (sy); WH_UNIFORM_VALUE (sc);
  return base + (ptrdiff_t)row * (luma ? sy : sc) + x;
}
// The same roles when the unfiltered reconstruction is kept macroblock by macroblock (WhPicJob::rec_blk): the word is the neighbour block's
// row 15 / 7 (top roles: the macroblock above-left, above, above-right) or the last word of one of its rows (column roles: the left one).
// A neighbour outside the picture is replaced by a block that exists (macroblock 0 / the last one): its samples are never used.
// (selects, no branches: lane roles as `if / else if` become a chain of exec-mask updates)
WH_FN const WH_G uint8_t* wh_tile_nb_addr_blk (int lane, const WhSeqParams& P, const WH_G uint8_t* blk, int mbx, int mby) {
  const bool top_y = lane < 7, col_y = lane >= 16 && lane < 32, top_c = lane >= 32 && lane < 38, col_c = lane >= 48;
  const int w = P.mb_w, xy = mby * w + mbx;
  const int cpl = lane >= 35 ? 1 : 0, ck = (lane - 32) - 3 * cpl;       // top_c: plane and word (0: above-left, 1 / 2: above)
  const int k = top_y ? lane : 0;                                        // (idle lanes take lane 0's role)
  const int nb_top = top_c ? (ck == 0 ? xy - w - 1 : xy - w) : (k == 0 ? xy - w - 1 : k < 5 ? xy - w : xy - w + 1);
  const int off_top = top_c ? 256 + cpl * 64 + 56 + (ck == 0 ? 4 : (ck - 1) * 4) : 240 + (k == 0 ? 12 : k < 5 ? (k - 1) * 4 : (k - 5) * 4);
  const int off_col = col_y ? (lane - 16) * 16 + 12 : 256 + ((lane - 48) >> 3) * 64 + (lane & 7) * 8 + 4;
  int nb = (col_y || col_c) ? xy - 1 : nb_top;
  const int off = (col_y || col_c) ? off_col : off_top;
  nb = nb < 0 ? 0 : nb;                                                  // (xy - w + 1 <= xy: never beyond the picture)
  return blk + (size_t)nb * WH_SRC_MB_BYTES + off;
}
// r->nb: the lane's word as loaded; the column roles keep its top byte (wh_tile_commit_nb)
WH_FN void wh_tile_fetch_nb_planes (int lane, const WhSeqParams& P, const WH_G uint8_t* rec0, const WH_G uint8_t* rec1, const WH_G uint8_t* rec2, int mbx, int mby, WhTileRegs* r) {
  r->nb = * (const WH_G uint32_t*)wh_tile_nb_addr_planes (lane, P, rec0, rec1, rec2, mbx, mby);
}
// (X: the neighbours may have been coded by another workgroup -- wave.h wh_ld_x32)
template <bool X = false>
WH_FN void wh_tile_fetch_nb (int lane, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, WhTileRegs* r) {
  const WH_G uint8_t* a = J.rec_blk ? wh_tile_nb_addr_blk (lane, P, (const WH_G uint8_t*)J.rec_blk, mbx, mby)
                                    : wh_tile_nb_addr_planes (lane, P, (const WH_G uint8_t*)J.rec[0], (const WH_G uint8_t*)J.rec[1], (const WH_G uint8_t*)J.rec[2], mbx, mby);
  r->nb = wh_ld_x32<X> ((const WH_G uint32_t*)a);
}
WH_FN void wh_tile_fetch (int lane, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, WhTileRegs* r) {
  wh_tile_fetch_src (lane, P, J, mbx, mby, r);
  wh_tile_fetch_nb (lane, P, J, mbx, mby, r);
}
// the neighbour samples only (the P kernel's source samples arrive by LDS-DMA: inter_mb.h wh_inter_cold_fetch)
WH_FN void wh_tile_commit_nb (WhMbLds& S, int lane, const WhTileRegs* r) {
  if (lane < 7) {
    * (uint32_t*)&S.rec_y[0 * 32 + lane * 4 - 4 + 8] = r->nb;
  } else if (lane >= 16 && lane < 32) {
    WH_RY (S, -1, lane - 16) = (uint8_t) (r->nb >> 24);
  } else if (lane >= 32 && lane < 38) {
    const int pl = (lane - 32) / 3, x = ((lane - 32) % 3) * 4 - 4;
    * (uint32_t*)&S.rec_c[pl][0 * 16 + x + 4] = r->nb;
  } else if (lane >= 48) {
    const int pl = (lane - 48) >> 3, y = lane & 7;
    WH_RC (S, pl, -1, y) = (uint8_t) (r->nb >> 24);
  }
}
WH_FN void wh_tile_commit (WhMbLds& S, int lane, const WhTileRegs* r) {
  {
    const int row = lane >> 2, seg = lane & 3;
    * (uint32_t*)&S.enc_y[row * 16 + seg * 4] = r->y;           // (= enc_y[4 * lane]: the source blocks are stored in lane order)
  }
  if (lane < 32) {
    const int pl = lane >> 4, row = (lane >> 1) & 7, half = lane & 1;
    * (uint32_t*)&S.enc_c[pl * 64 + row * 8 + half * 4] = r->c;
  }
  wh_tile_commit_nb (S, lane, r);
}
WH_FN void wh_load_mb_tile (WhMbLds& S, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  WV_LANES_BEGIN (lane)
  WhTileRegs r;
  wh_tile_fetch (lane, P, J, mbx, mby, &r);
  wh_tile_commit (S, lane, &r);
  WV_LANES_END
}

// ---- Intra16x16 predictors into S.pred_y (mode numbering = the reference's I16_PRED_*) ----------
enum { WH_I16_V = 0, WH_I16_H = 1, WH_I16_DC = 2, WH_I16_P = 3, WH_I16_DC_L = 4, WH_I16_DC_T = 5, WH_I16_DC_128 = 6 };

WH_FN void wh_pred_i16 (WhMbLds& S, int mode, int sum_t, int sum_l, int pl_b, int pl_c, int pl_a) {
  WV_LANES_BEGIN (lane)
  const int row = lane >> 2, x0 = (lane & 3) * 4;
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + k;
    int v;
    switch (mode) {
    case WH_I16_V: v = WH_RY (S, x, -1); break;
    case WH_I16_H: v = WH_RY (S, -1, row); break;
    case WH_I16_DC: v = (sum_t + sum_l + 16) >> 5; break;
    case WH_I16_DC_L: v = (sum_l + 8) >> 4; break;
    case WH_I16_DC_T: v = (sum_t + 8) >> 4; break;
    case WH_I16_P: v = wh_clip255 ((pl_a + pl_b * (x - 7) + pl_c * (row - 7) + 16) >> 5); break;
    default: v = 128; break;
    }
    S.pred_y[row * 16 + x] = (uint8_t)v;
  }
  WV_LANES_END
}

// four adjacent samples (x0 .. x0+3 of row y) of an Intra16x16 prediction, packed -- for costing a mode without
// materialising it in LDS
WH_FN uint32_t wh_pred_i16_4 (const WhMbLds& S, int mode, int x0, int y, int sum_t, int sum_l, int pl_b, int pl_c, int pl_a) {
  switch (mode) {
  case WH_I16_V: return * (const uint32_t*)&S.rec_y[0 * 32 + x0 + 8];
  case WH_I16_H: return 0x01010101u * (uint32_t)WH_RY (S, -1, y);
  case WH_I16_DC: return 0x01010101u * (uint32_t) ((sum_t + sum_l + 16) >> 5);
  case WH_I16_DC_L: return 0x01010101u * (uint32_t) ((sum_l + 8) >> 4);
  case WH_I16_DC_T: return 0x01010101u * (uint32_t) ((sum_t + 8) >> 4);
  case WH_I16_P: {
    const int base = pl_a + pl_c * (y - 7) + 16;
    return (uint32_t)wh_clip255 ((base + pl_b * (x0 - 7)) >> 5) | ((uint32_t)wh_clip255 ((base + pl_b * (x0 - 6)) >> 5) << 8) |
           ((uint32_t)wh_clip255 ((base + pl_b * (x0 - 5)) >> 5) << 16) | ((uint32_t)wh_clip255 ((base + pl_b * (x0 - 4)) >> 5) << 24);
  }
  default: return 0x80808080u;
  }
}

// chroma predictors into S.pred_c (numbering = the reference's C_PRED_*)
enum { WH_C_DC = 0, WH_C_H = 1, WH_C_V = 2, WH_C_P = 3, WH_C_DC_L = 4, WH_C_DC_T = 5, WH_C_DC_128 = 6 };

WH_FN void wh_pred_chroma (WhMbLds& S, int mode, const int* st /*[pl][2] top sums*/, const int* sl /*[pl][2] left sums*/,
                           const int* pb, const int* pc, const int* pa) {
  WV_LANES_BEGIN (lane)
  if (lane < 32) {
    const int pl = lane >> 4, row = (lane >> 1) & 7, x0 = (lane & 1) * 4;
    // (selects, not st[pl * 2]: a local array indexed by a per-lane value is very slow)
    const int t0 = pl ? st[2] : st[0], t1 = pl ? st[3] : st[1], l0 = pl ? sl[2] : sl[0], l1 = pl ? sl[3] : sl[1];
    const int ppa = pl ? pa[1] : pa[0], ppb = pl ? pb[1] : pb[0], ppc = pl ? pc[1] : pc[0];
    for (int k = 0; k < 4; ++k) {
      const int x = x0 + k;
      int v;
      switch (mode) {
      case WH_C_V: v = WH_RC (S, pl, x, -1); break;
      case WH_C_H: v = WH_RC (S, pl, -1, row); break;
      case WH_C_DC:
        if (row < 4) v = (x < 4) ? ((t0 + l0 + 4) >> 3) : ((t1 + 2) >> 2);
        else         v = (x < 4) ? ((l1 + 2) >> 2) : ((t1 + l1 + 4) >> 3);
        break;
      case WH_C_DC_L: v = (row < 4) ? ((l0 + 2) >> 2) : ((l1 + 2) >> 2); break;
      case WH_C_DC_T: v = (x < 4) ? ((t0 + 2) >> 2) : ((t1 + 2) >> 2); break;
      case WH_C_P: v = wh_clip255 ((ppa + ppb * (x - 3) + ppc * (row - 3) + 16) >> 5); break;
      default: v = 128; break;
      }
      S.pred_c[pl * 64 + row * 8 + x] = (uint8_t)v;
    }
  }
  WV_LANES_END
}

// The same predictors for the mode decision, one 4-sample row of a 4x4 block per lane, as a packed word: pl = plane, bb = block (raster in the 8x8), r = row of the
// block; t0 / t1 / l0 / l1 = the plane's top / left half sums, ppa / ppb / ppc = its plane parameters.
WH_FN uint32_t wh_pred_chroma4 (const WhMbLds& S, int mode, int pl, int bb, int r, int t0, int t1, int l0, int l1, int ppa, int ppb, int ppc) {
  const int px = (bb & 1) * 4, py = (bb >> 1) * 4 + r;
  int v;
  switch (mode) {
  case WH_C_V: return wh_ld4u (S.rec_c[pl], 4 + px);
  case WH_C_H: v = WH_RC (S, pl, -1, py); break;
  case WH_C_DC: v = bb == 0 ? (t0 + l0 + 4) >> 3 : bb == 1 ? (t1 + 2) >> 2 : bb == 2 ? (l1 + 2) >> 2 : (t1 + l1 + 4) >> 3; break;
  case WH_C_DC_L: v = ((bb < 2 ? l0 : l1) + 2) >> 2; break;
  case WH_C_DC_T: v = (((bb & 1) ? t1 : t0) + 2) >> 2; break;
  case WH_C_P: {
    const int base = ppa + ppc * (py - 3) + 16 + ppb * (px - 3);
    return (uint32_t)wh_clip255 (base >> 5) | ((uint32_t)wh_clip255 ((base + ppb) >> 5) << 8) | ((uint32_t)wh_clip255 ((base + 2 * ppb) >> 5) << 16) |
           ((uint32_t)wh_clip255 ((base + 3 * ppb) >> 5) << 24);
  }
  default: v = 128; break;
  }
  return (uint32_t)v * 0x01010101u;
}
// lane's term of the edge sums of the chroma decision: quads 0..3 the top halves (plane * 2 + half), 4..7 the left halves, 8 / 9 the plane mode's horizontal
// gradient of Cb / Cr, 10 / 11 the vertical one (get_intra_predictor.cpp:530-570 WelsIChromaPredPlane_c)
WH_FN int wh_chroma_edge_term (const WhMbLds& S, int lane, bool has_t, bool has_l, bool plane) {
  const int k = lane >> 2, j = lane & 3;
  if (k < 4) return has_t ? WH_RC (S, k >> 1, (k & 1) * 4 + j, -1) : 0;
  if (k < 8) return has_l ? WH_RC (S, (k - 4) >> 1, -1, (k & 1) * 4 + j) : 0;
  if (k < 12 && plane) {
    const int pl = k & 1;
    return k < 10 ? (j + 1) * (WH_RC (S, pl, 4 + j, -1) - WH_RC (S, pl, 2 - j, -1)) : (j + 1) * (WH_RC (S, pl, -1, 4 + j) - WH_RC (S, pl, -1, 2 - j));
  }
  return 0;
}

// ---- one row (4 pixels) of an Intra4x4 prediction, standard mode numbering 0..8 ------------------
// E (0..12): L3 L2 L1 L0 TL T0..T7   (so p[-1,j] = E (3-j), p[i,-1] = E (5+i), TL = E (4)), packed four per word: the
// index varies per lane (mode, position), and a byte array would end up in scratch memory
typedef struct WhE13 { uint32_t w[4]; } WhE13;
WH_FN int wh_e13 (const WhE13& E, int i) {
  const uint32_t v = i < 4 ? E.w[0] : i < 8 ? E.w[1] : i < 12 ? E.w[2] : E.w[3];
  return (int) ((v >> (8 * (i & 3))) & 255u);
}
#define WH_F3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)
#define WH_F2(a, b) (((a) + (b) + 1) >> 1)
WH_FN int wh_pred4_px (int mode, int x, int y, const WhE13& EE, int dcval) {
#define E(i) wh_e13 (EE, (i))
  switch (mode) {
  case 0: return E (5 + x);                                   // V
  case 1: return E (3 - y);                                   // H
  case 2: return dcval;                                      // DC family
  case 3:                                                    // DDL
    if (x == 3 && y == 3) return (E (5 + 6) + 3 * E (5 + 7) + 2) >> 2;
    return WH_F3 (E (5 + x + y), E (5 + x + y + 1), E (5 + x + y + 2));
  case 4:                                                    // DDR
    return WH_F3 (E (4 + x - y - 1), E (4 + x - y), E (4 + x - y + 1));
  case 5: {                                                  // VR
    const int z = 2 * x - y, i = x - (y >> 1);
    if (z >= 0) return (z & 1) ? WH_F3 (E (5 + i - 2), E (5 + i - 1), E (5 + i)) : WH_F2 (E (5 + i - 1), E (5 + i));
    if (z == -1) return WH_F3 (E (3), E (4), E (5));
    return WH_F3 (E (3 - (y - 1)), E (3 - (y - 2)), E (3 - (y - 3)));
  }
  case 6: {                                                  // HD
    const int z = 2 * y - x, j = y - (x >> 1);
    if (z >= 0) return (z & 1) ? WH_F3 (E (3 - (j - 2)), E (3 - (j - 1)), E (3 - j)) : WH_F2 (E (3 - (j - 1)), E (3 - j));
    if (z == -1) return WH_F3 (E (3), E (4), E (5));
    return WH_F3 (E (5 + x - 1), E (5 + x - 2), E (5 + x - 3));
  }
  case 7: {                                                  // VL
    const int i = x + (y >> 1);
    return (y & 1) ? WH_F3 (E (5 + i), E (5 + i + 1), E (5 + i + 2)) : WH_F2 (E (5 + i), E (5 + i + 1));
  }
  default: {                                                 // HU
    const int z = x + 2 * y, j = y + (x >> 1);
    if (z > 5) return E (0);
    if (z == 5) return (E (1) + 3 * E (0) + 2) >> 2;
    return (z & 1) ? WH_F3 (E (3 - j), E (3 - (j + 1)), E (3 - (j + 2))) : WH_F2 (E (3 - j), E (3 - (j + 1)));
  }
  }
#undef E
}

// The same nine predictors as a TABLE LOOK-UP (what the macroblock bodies use; wh_pred4_px stays the definition and serves the leaf
// primitive).  Every predicted sample is one of three functions of the edge samples at an index that only depends on (mode, x, y):
// with X = E(0), E(0) .. E(12), E(12) (the ends doubled: the two "3 x outer sample" cases of DDL and HU become ordinary 3-tap
// cases), RAW[a] = X[a], F2[a] = (X[a] + X[a+1] + 1) >> 1, F3[a] = (X[a] + 2 X[a+1] + X[a+2] + 2) >> 2 -- or the DC value.  A lane per
// index fills the 49-byte table (0..14 RAW, 16..30 F2, 32..46 F3, 48 DC), then lane (mode, row) fetches its four samples with the
// byte offsets below: nine modes cost one pass without a nine-way divergent switch.  Generated from wh_pred4_px's index arithmetic
// and checked against it for every (mode, x, y) on random edges (tools/gen_tables.py --i4).
WH_TABLE uint32_t kWhI4Desc[36] = {     // [mode * 4 + row]: byte x = table offset of sample (x, row)
  0x09080706u, 0x09080706u, 0x09080706u, 0x09080706u, 0x04040404u, 0x03030303u, 0x02020202u, 0x01010101u, 0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u,
  0x29282726u, 0x2a292827u, 0x2b2a2928u, 0x2c2b2a29u, 0x27262524u, 0x26252423u, 0x25242322u, 0x24232221u, 0x18171615u, 0x27262524u, 0x17161523u, 0x26252422u,
  0x26252414u, 0x24142313u, 0x23132212u, 0x22122111u, 0x19181716u, 0x29282726u, 0x1a191817u, 0x2a292827u, 0x21122213u, 0x20112112u, 0x01012011u, 0x01010101u };
// edge sample X[i] (i = 0 .. 16; beyond 14: E(12) again) of the 4x4 block at (bx, by) in 4-sample units, from the reconstruction tile
WH_FN int wh_i4_edge (const WhMbLds& S, int bx, int by, int i) {
  const int e = i < 1 ? 0 : i > 13 ? 12 : i - 1;
  const int x = e < 5 ? bx * 4 - 1 : bx * 4 + e - 5, y = e < 4 ? by * 4 + 3 - e : by * 4 - 1;
  return WH_RY (S, x, y);
}
// the 49-byte table of one block into `t` (lanes 0 .. 15; lane 15: the DC value for the block's availability)
WH_FN void wh_i4_fill_table (const WhMbLds& S, uint8_t* t, int lane, int bx, int by, bool a_l, bool a_t) {
  if (lane < 15) {
    const int x0 = wh_i4_edge (S, bx, by, lane), x1 = wh_i4_edge (S, bx, by, lane + 1), x2 = wh_i4_edge (S, bx, by, lane + 2);
    t[lane] = (uint8_t)x0; t[16 + lane] = (uint8_t)WH_F2 (x0, x1); t[32 + lane] = (uint8_t)WH_F3 (x0, x1, x2);
  } else if (lane == 15) {
    const uint32_t tw = * (const uint32_t*)&WH_RY (S, bx * 4, by * 4 - 1);
    const int sum_t4 = (int) ((tw & 255u) + ((tw >> 8) & 255u) + ((tw >> 16) & 255u) + (tw >> 24));
    const int sum_l4 = WH_RY (S, bx * 4 - 1, by * 4) + WH_RY (S, bx * 4 - 1, by * 4 + 1) + WH_RY (S, bx * 4 - 1, by * 4 + 2) + WH_RY (S, bx * 4 - 1, by * 4 + 3);
    t[48] = (uint8_t) ((a_l && a_t) ? (sum_l4 + sum_t4 + 4) >> 3 : a_l ? (sum_l4 + 2) >> 2 : a_t ? (sum_t4 + 2) >> 2 : 128);
  }
}

// ---- the intra MB ---------------------------------------------------------------------------------
// Leaves: S.lv_*, S.nzc, S.i4_rem/i4_prev, the rec tile.  Returns through *o.
typedef struct WhIntraResult {
  int mb_type, cbp, i16_mode_std, chroma_mode_std, cost_luma, cost_chroma;
} WhIntraResult;

// `inter_cost`: in P slices the I16x16 cost must beat the best inter cost so far, otherwise nothing is
// encoded and false is returned (WelsMdFirstIntraMode); I slices pass INT_MAX.
// Intra16x16 mode costs of the MB (neighbour sums, plane parameters, best mode): needs only the tile, so a caller can
// compute it early (the P kernel does, underneath its window loads).
typedef struct WhI16Cost { int sum_t, sum_l, pl_a, pl_b, pl_c, best_mode, best_cost; } WhI16Cost;
WH_FN void wh_i16_costs (WhMbLds& S, int avail, int use_satd, int lambda, WhI16Cost* k) {
  const int av3 = avail & 7;
  const bool has_l = (avail & WH_AV_LEFT) != 0, has_t = (avail & WH_AV_TOP) != 0;

  // ---------------- I16x16 mode decision ----------------
  int sum_t = 0, sum_l = 0, pl_a = 0, pl_b = 0, pl_c = 0;
  {
    // the four sums of the neighbour samples side by side, one per DPP row (one pass instead of four wave sums): row 0 the top
    // samples, row 1 the left ones, rows 2 / 3 the eight terms of the plane predictor's H / V
    int h, v;
    const bool plane = av3 == 7;
    WV_ROWSUM4 (sum_t, sum_l, h, v, lane, ([&] () {
      const int r = lane >> 4, k = lane & 15, k7 = k & 7;
      if (r == 0) return has_t ? (int)WH_RY (S, k, -1) : 0;
      if (r == 1) return has_l ? (int)WH_RY (S, -1, k) : 0;
      if (!plane || k >= 8) return 0;
      return r == 2 ? (k7 + 1) * ((int)WH_RY (S, 8 + k7, -1) - (int)WH_RY (S, 6 - k7, -1)) : (k7 + 1) * ((int)WH_RY (S, -1, 8 + k7) - (int)WH_RY (S, -1, 6 - k7)); }) ());
    if (plane) {
      pl_a = (WH_RY (S, -1, 15) + WH_RY (S, 15, -1)) << 4;
      pl_b = (5 * h + 32) >> 6;
      pl_c = (5 * v + 32) >> 6;
    }
  }
  // candidate modes in the reference's order (scalars, not an array: nothing here is indexed at run time)
  int m0, m1, m2, m3, ncand;
  if (has_l && has_t) { m0 = WH_I16_V; m1 = WH_I16_H; m2 = WH_I16_DC; m3 = WH_I16_P; ncand = (av3 == 7) ? 4 : 3; }
  else if (has_l) { m0 = WH_I16_DC_L; m1 = WH_I16_H; m2 = m0; m3 = m0; ncand = 2; }
  else if (has_t) { m0 = WH_I16_DC_T; m1 = WH_I16_V; m2 = m0; m3 = m0; ncand = 2; }
  else { m0 = WH_I16_DC_128; m1 = m0; m2 = m0; m3 = m0; ncand = 1; }
  // cost of every candidate in one pass, straight from the neighbour samples (no prediction is written to LDS yet);
  // lambda * BsSizeUE (g_kiMapModeI16x16[mode]):  V -> 1 bit, H/DC* -> 3, Plane -> 5
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if (!use_satd) {
    int p01, p23;
#define WH_I16_SAD(m) wh_sad4 (* (const uint32_t*)&S.enc_y[lane * 4], wh_pred_i16_4 (S, (m), (lane & 3) * 4, lane >> 2, sum_t, sum_l, pl_b, pl_c, pl_a))
    WV_SUM2 (p01, p23, lane, (WH_I16_SAD (m0) | (ncand > 1 ? WH_I16_SAD (m1) << 16 : 0)), ((ncand > 2 ? WH_I16_SAD (m2) : 0) | (ncand > 3 ? WH_I16_SAD (m3) << 16 : 0)));
#undef WH_I16_SAD
    c0 = p01 & 0xffff; c1 = (int) ((unsigned)p01 >> 16); c2 = p23 & 0xffff; c3 = (int) ((unsigned)p23 >> 16);
  } else {
    // SATD layout: lane quad = raster 4x4 block, lane & 3 = row inside it
#define WH_I16_X0 (((lane >> 2) & 3) * 4)
#define WH_I16_Y ((lane >> 4) * 4 + (lane & 3))
#define WH_I16_SATD(dst, m) WV_SATD_ROWS (dst, lane, true, * (const uint32_t*)&S.enc_y[WH_I16_Y * 16 + WH_I16_X0], \
                                          wh_pred_i16_4 (S, (m), WH_I16_X0, WH_I16_Y, sum_t, sum_l, pl_b, pl_c, pl_a))
    WH_I16_SATD (c0, m0);
    if (ncand > 1) WH_I16_SATD (c1, m1);
    if (ncand > 2) WH_I16_SATD (c2, m2);
    if (ncand > 3) WH_I16_SATD (c3, m3);
#undef WH_I16_SATD
#undef WH_I16_X0
#undef WH_I16_Y
  }
  int best_mode = m0, best_cost = 0x7fffffff, last_mode = -1;
#define WH_I16_TRY(m, c) do { const int _c = (c) + lambda * (((m) == WH_I16_V) ? 1 : ((m) == WH_I16_P) ? 5 : 3); if (_c < best_cost) { best_cost = _c; best_mode = (m); } } while (0)
  WH_I16_TRY (m0, c0);
  if (ncand > 1) WH_I16_TRY (m1, c1);
  if (ncand > 2) WH_I16_TRY (m2, c2);
  if (ncand > 3) WH_I16_TRY (m3, c3);
#undef WH_I16_TRY
  k->sum_t = sum_t; k->sum_l = sum_l; k->pl_a = pl_a; k->pl_b = pl_b; k->pl_c = pl_c; k->best_mode = best_mode; k->best_cost = best_cost;
  (void)last_mode;
}

// CPLX >= 0: the complexity mode is known at compile time (the P kernel's variant for LOW complexity launches, inter_mb.h)
// X: the slice is coded by several workgroups (hip_backend.hip k_inter_split): the neighbours' states were stored write-through by another compute unit
// and are loaded past the caches like their samples (wave.h wh_ld_x32)
template <int CPLX = -1, bool X = false>
WH_FN bool wh_intra_md_enc_p (WhMbLds& S, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, int avail, int qp, int qpc,
                              int inter_cost, WhIntraResult* o, const WhI16Cost* pre = nullptr, int stale_cbp = 0) {
  const int lambda = kWhLambda[qp];
  const int use_satd = CPLX >= 0 ? (CPLX > 0) : (P.complexity > 0);
  const bool has_l = (avail & WH_AV_LEFT) != 0, has_t = (avail & WH_AV_TOP) != 0;
  // phase cycles of an I picture's macroblocks (a P macroblock's intra test is part of the P body's own phases)
#if defined(WH_EMU) || !defined(WH_PROF)
#define WH_PROF_MARK_I(id) ((void)0)
#define WH_PROF_SUB_I(id) ((void)0)
#else
  unsigned long long _wh_t0 = (P.prof && J.slice_type == WH_SLICE_I) ? (unsigned long long)__builtin_readcyclecounter() : 0ULL;
#define WH_PROF_MARK_I(id) do { if (J.slice_type == WH_SLICE_I) WH_PROF_MARK_RAW (P, S, id); } while (0)
#if defined(WH_PROF_I4)          /* sub-phases of the Intra4x4 pair steps (tools/phase_profile.py ... intra with a -DWH_PROF -DWH_PROF_I4 library) */
#define WH_PROF_SUB_I(id) WH_PROF_MARK_I (id)
#else
#define WH_PROF_SUB_I(id) ((void)0)
#endif
#endif
  WhI16Cost own;
  if (!pre) { wh_i16_costs (S, avail, use_satd, lambda, &own); pre = &own; }
  WH_PROF_MARK_I (2);       // Intra16x16 mode costs
  const int sum_t = pre->sum_t, sum_l = pre->sum_l, pl_a = pre->pl_a, pl_b = pre->pl_b, pl_c = pre->pl_c;
  const int best_mode = pre->best_mode, best_cost = pre->best_cost, last_mode = -1;
  const int av3 = avail & 7;
  if (!(best_cost < inter_cost)) return false;
  int cost_luma = best_cost;
  int mb_type = WH_MB_I16x16;

  // ---------------- I4x4 (fine partition) ----------------
  bool try_i4 = true;
  if (!use_satd) {
    // MdIntraAnalysisVaaInfo (md.cpp:435-475,497-503): variance of the sixteen 4x4 means >= 150
    WV_LANES_BEGIN (lane)
    const int b = lane >> 2, r = lane & 3;
    const uint8_t* e = &S.enc_y[((b >> 2) * 4 + r) * 16 + (b & 3) * 4];
    S.part[lane] = e[0] + e[1] + e[2] + e[3];
    WV_LANES_END
    int sum_avg, sum_sqr;
    WV_SUM (sum_avg, lane, (lane < 16 ? ((S.part[lane * 4] + S.part[lane * 4 + 1] + S.part[lane * 4 + 2] + S.part[lane * 4 + 3]) >> 4) : 0));
    WV_SUM (sum_sqr, lane, (lane < 16 ? (((S.part[lane * 4] + S.part[lane * 4 + 1] + S.part[lane * 4 + 2] + S.part[lane * 4 + 3]) >> 4) *
                                          ((S.part[lane * 4] + S.part[lane * 4 + 1] + S.part[lane * 4 + 2] + S.part[lane * 4 + 3]) >> 4)) : 0));
    try_i4 = (sum_sqr - ((sum_avg * sum_avg) >> 4)) >= 150;
  }
  int cbp = 0;
  if (try_i4) {
    // neighbour Intra4x4PredMode cache (md.cpp:51-130 FillNeighborCacheIntra): in the tile (what the store reads) and in a lane table
    // (what the sixteen blocks' predicted modes are read from: v_readlane instead of an LDS round trip per block)
    WvLaneArr i4t, dsc;
#if defined(WH_EMU)
    memset (&i4t, 0, sizeof (i4t)); memset (&dsc, 0, sizeof (dsc));
#else
    i4t = 0; dsc = 0;
#endif
    WV_LSET_IF (dsc, lane, lane < 36, (int)kWhI4Desc[lane]);
    WV_LSET_IF (i4t, lane, lane < 25, ([&] () {
      // type + the mode word of the state a lane needs in one batch (a state that exists for every lane: its own MB's when it has
      // no neighbour to ask), the selects afterwards -- see wh_tile_fetch_nb
      const int cx = lane % 5, cy = lane / 5;
      const bool from_t = cy == 0 && cx > 0 && has_t, from_l = cx == 0 && cy > 0 && has_l;
      const WH_G WhMbState* n = (const WH_G WhMbState*)J.mbs + (from_t ? (mby - 1) * P.mb_w + mbx : from_l ? mby * P.mb_w + mbx - 1 : mby * P.mb_w + mbx);
      const int idx = from_t ? 12 + cx - 1 : from_l ? (cy - 1) * 4 + 3 : 0;
      // (X: a plain load could hit a line this compute unit's L1 holds from before the neighbour -- or this very macroblock's predecessor in the
      //  picture buffer, read as the lane's placeholder above -- was written: a write-through store does not touch that copy.  Found as one
      //  mismatching P picture in one of six runs of the GPU tier, round 6.)
      const int type = X ? (int) (wh_ld_x32<X> ((const WH_G uint32_t*)n) & 0xffu) : (int)n->mb_type;
      const int8_t mode = X ? (int8_t) ((wh_ld_x32<X> ((const WH_G uint32_t*)n + 2 + (idx >> 2)) >> (8 * (idx & 3))) & 0xffu) : n->i4_mode[idx];
      const int8_t v = (from_t || from_l) ? (type == WH_MB_I4x4 ? mode : (int8_t)2) : (int8_t) - 1;
      S.i4m[lane] = v;
      return (int)v; }) ());
    WV_SYNC();
    const int lam4 = lambda << 2;
    int cost4 = 0;
    uint16_t prev_flags = 0;
    bool completed = true;
    WH_PROF_MARK_I (3);     // texture analysis + neighbour mode cache
    // sample availability of 4x4 block b (the reference tabulates it: g_kiNeighborIntraToI4x4)
    auto blk_avail = [&] (int b, bool& a_l, bool& a_t, bool& a_tl, bool& a_tr) {
      const int bx = wh_blk_x (b), by = wh_blk_y (b);
      a_l = bx > 0 || has_l;
      a_t = by > 0 || has_t;
      if (bx > 0 && by > 0) a_tl = true;
      else if (bx > 0) a_tl = has_t;
      else if (by > 0) a_tl = has_l;
      else a_tl = (avail & WH_AV_TOPLEFT) != 0;
      if (by == 0) a_tr = (bx < 3) ? has_t : ((avail & WH_AV_TOPRIGHT) != 0);
      else a_tr = ((0x5744 >> b) & 1) != 0;   // blocks whose top-right 4x4 is already reconstructed inside this MB (in CODING order: what the decoder has)
    };
    // the block's best mode from the cost table `ct` (mode m at lane cbase + m * cstep)
    auto decide = [&] (const WvLaneArr& ct, int cbase, int cstep, bool a_l, bool a_t, bool a_tl, bool a_tr, int pred_mode, int& bmode, int& bcost) {
      // cost of standard mode m incl. the mode-signalling term lambda[pred_mode == m]
#define WH_C4(m) ((use_satd ? ((WV_LGET (ct, cbase + (m) * cstep) + 1) >> 1) : WV_LGET (ct, cbase + (m) * cstep)) + ((pred_mode == (m)) ? lambda : lam4))
      // candidate order of the reference (g_kiIntra4AvailMode rows), standard numbering
      // (the list is packed four bits per entry: an int array indexed at run time would live in scratch memory)
      unsigned long long list = 0;
      int n = 0;
#define WH_PUSH(m) do { list |= (unsigned long long) (m) << (4 * n); ++n; } while (0)
#define WH_LIST(i) ((int) ((list >> (4 * (i))) & 15ULL))
      if (a_l && a_t) {
        WH_PUSH (2); WH_PUSH (1); WH_PUSH (0); WH_PUSH (8);
        if (a_tr) { WH_PUSH (3); WH_PUSH (7); }
        if (a_tl) { WH_PUSH (4); WH_PUSH (5); WH_PUSH (6); }
      } else if (a_l) { WH_PUSH (2); WH_PUSH (1); WH_PUSH (8); }
      else if (a_t) { WH_PUSH (2); WH_PUSH (0); if (a_tr) { WH_PUSH (3); WH_PUSH (7); } }
      else { WH_PUSH (2); }
      if (!use_satd && (n == 9 || n == 7)) {
        // WelsMdI4x4Fast decision tree (svc_base_layer_md.cpp:598-826)
        bmode = 2; bcost = WH_C4 (2);
        const int c_h = WH_C4 (1);
        if (c_h < bcost) { bmode = 1; bcost = c_h; }
        const int c_v = WH_C4 (0);
        if (c_v < bcost) { bmode = 0; bcost = c_v; }
        if (c_v < c_h) {
          if (n == 9) {
            bool fake = true;
            const int c_vr = WH_C4 (5);
            if (c_vr < bcost) { bmode = 5; bcost = c_vr; }
            if (c_vr < c_v) fake = false;
            const int c_vl = WH_C4 (7);
            if (c_vl < bcost) { bmode = 7; bcost = c_vl; }
            if (c_vl < c_v) fake = false;
            if (!fake) {
              if (c_vr < c_vl) { const int c = WH_C4 (4); if (c < bcost) { bmode = 4; bcost = c; } }
              else { const int c = WH_C4 (3); if (c < bcost) { bmode = 3; bcost = c; } }
            }
          } else {
            const int c_ddr = WH_C4 (4);
            if (c_ddr < bcost) { bmode = 4; bcost = c_ddr; }
            const int c_vr = WH_C4 (5);
            if (c_vr < bcost) { bmode = 5; bcost = c_vr; }
          }
        } else {
          bool fake = true;
          const int c_hd = WH_C4 (6);
          if (c_hd < bcost) { bmode = 6; bcost = c_hd; }
          if (c_hd < c_h) fake = false;
          const int c_hu = WH_C4 (8);
          if (c_hu < bcost) { bmode = 8; bcost = c_hu; }
          if (c_hu < c_h) fake = false;
          if (!fake) {
            if (c_hd < c_hu) { const int c = WH_C4 (4); if (c < bcost) { bmode = 4; bcost = c; } }
            else if (n == 9) { const int c = WH_C4 (3); if (c < bcost) { bmode = 3; bcost = c; } }
          }
        }
      } else {
        bmode = WH_LIST (0); bcost = 0x7fffffff;
        for (int i = 0; i < n; ++i) {
          const int c = WH_C4 (WH_LIST (i));
          if (c < bcost) { bcost = c; bmode = WH_LIST (i); }
        }
      }
#undef WH_C4
#undef WH_PUSH
#undef WH_LIST
    };
    // one block, start to finish (rounds 1-5: all sixteen, in coding order); false = the MB's Intra4x4 attempt is over
    auto single_block = [&] (int b) -> bool {
      const int bx = wh_blk_x (b), by = wh_blk_y (b);
      bool a_l, a_t, a_tl, a_tr;
      blk_avail (b, a_l, a_t, a_tl, a_tr);
      // predicted mode
      const int m_left = WV_LGET (i4t, (by + 1) * 5 + bx), m_top = WV_LGET (i4t, by * 5 + bx + 1);
      const int pred_mode = (m_left == -1 || m_top == -1) ? 2 : wh_min (m_left, m_top);
      // candidate predictions: the block's filter table (see kWhI4Desc), then lane (mode m = lane >> 2, row r = lane & 3) fetches its
      // four samples, stores them for the encode step and costs them; the modes' costs are quad sums, read from a lane table
      uint8_t* const tbl = (uint8_t*)S.part2;
      WV_LANES_BEGIN (lane)
      wh_i4_fill_table (S, tbl, lane, bx, by, a_l, a_t);
      WV_LANES_END
      WvLaneArr ct;
#if defined(WH_EMU)
      memset (&ct, 0, sizeof (ct));
#else
      ct = 0;
#endif
      WV_QUADSUM_TAB (ct, lane, (lane < 36 ? ([&] () {
        const int m = lane >> 2, r = lane & 3;
        const uint32_t d = (uint32_t)WV_LOWN (dsc, lane);
        const uint32_t px = (uint32_t)tbl[d & 255u] | ((uint32_t)tbl[(d >> 8) & 255u] << 8) | ((uint32_t)tbl[(d >> 16) & 255u] << 16) | ((uint32_t)tbl[d >> 24] << 24);
        * (uint32_t*)&S.pred4[m * 16 + r * 4] = px;
        const uint32_t e = * (const uint32_t*)&S.enc_y[(by * 4 + r) * 16 + bx * 4];
        if (!use_satd) return wh_sad4 (e, px);
        int o0, o1, o2, o3;
        wh_had4 ((int) (e & 255u) - (int) (px & 255u), (int) ((e >> 8) & 255u) - (int) ((px >> 8) & 255u), (int) ((e >> 16) & 255u) - (int) ((px >> 16) & 255u),
                 (int) (e >> 24) - (int) (px >> 24), &o0, &o1, &o2, &o3);
        int16_t* t = &S.tmp[m * 16 + r * 4];
        t[0] = (int16_t)o0; t[1] = (int16_t)o1; t[2] = (int16_t)o2; t[3] = (int16_t)o3;
        return 0; }) () : 0));
      WV_SYNC();                           // (the lanes' stores above are read by other lanes below and in the encode step)
      if (use_satd) {
        WV_QUADSUM_TAB (ct, lane, (lane < 36 ? ([&] () {
          const int m = lane >> 2, c = lane & 3;
          const int16_t* t = &S.tmp[m * 16 + c];
          int o0, o1, o2, o3;
          wh_had4 (t[0], t[4], t[8], t[12], &o0, &o1, &o2, &o3);
          return wh_abs (o0) + wh_abs (o1) + wh_abs (o2) + wh_abs (o3); }) () : 0));
      }
      int bmode, bcost;
      decide (ct, 0, 4, a_l, a_t, a_tl, a_tr, pred_mode, bmode, bcost);
      cost4 += bcost;
      if (cost4 >= cost_luma) { completed = false; return false; }
      if (pred_mode == bmode) prev_flags |= (uint16_t) (1u << b);
      const int rem = (bmode < pred_mode) ? bmode : bmode - 1;
      WV_LSET (i4t, (by + 1) * 5 + bx + 1, bmode);
      // (the block's mode, rem_intra4x4_pred_mode and total_coeff go into the tile inside the encode step's own lane blocks)
      const int nz = wh_encrec_i4 (S, b, bmode, qp, (int8_t) ((pred_mode == bmode) ? 0 : rem));
      if (nz > 0) cbp |= 1 << (b >> 2);
      return true;
    };
    // Two blocks at once, one per half of the wave (round 6; SAD costs, i.e. LOW complexity).  The sixteen blocks of a macroblock are no chain: a
    // block predicts from its left, upper, upper-left and -- where coding order has it -- upper-right neighbour, so the blocks of one 2:1
    // diagonal of the 4x4 grid (bx + 2 by) are independent: ten steps, six of them with two blocks -- (2,0)+(0,1), (3,0)+(1,1), (2,1)+(0,2),
    // (3,1)+(1,2), (2,2)+(0,3), (3,2)+(1,3) -- instead of sixteen.  What coding order decides stays as it is: upper-right availability is the table
    // above, and the early end of the attempt (svc_base_layer_md.cpp:418-546: the costs are added up in coding order until they reach the
    // Intra16x16 cost) depends on the blocks' order only through WHEN it happens, never WHETHER: the costs are not negative, so some prefix of
    // the coding order reaches the limit exactly when the sum of ALL of them does, and an attempt that ends leaves nothing behind that is read
    // (Intra16x16 then codes every block again).  Lanes 0..31 hold the step's first block, lanes 32..63 the second.
    auto block_pair = [&] (int bA, int bB) -> bool {
      const int bxA = wh_blk_x (bA), byA = wh_blk_y (bA), bxB = wh_blk_x (bB), byB = wh_blk_y (bB);
      bool alA, atA, atlA, atrA, alB, atB, atlB, atrB;
      blk_avail (bA, alA, atA, atlA, atrA);
      blk_avail (bB, alB, atB, atlB, atrB);
      const int mlA = WV_LGET (i4t, (byA + 1) * 5 + bxA), mtA = WV_LGET (i4t, byA * 5 + bxA + 1), mlB = WV_LGET (i4t, (byB + 1) * 5 + bxB), mtB = WV_LGET (i4t, byB * 5 + bxB + 1);
      const int pmA = (mlA == -1 || mtA == -1) ? 2 : wh_min (mlA, mtA), pmB = (mlB == -1 || mtB == -1) ? 2 : wh_min (mlB, mtB);
      uint8_t* const tbl2 = (uint8_t*)S.part;                 // two filter tables, 64 bytes apart
      uint8_t* const predB = (uint8_t*)&S.res[256];           // the second block's nine candidates (the chroma coefficients' place: not in use before the chroma step)
      WV_LANES_BEGIN (lane)
      {
        const int h = lane >> 5;
        wh_i4_fill_table (S, tbl2 + 64 * h, lane & 31, h ? bxB : bxA, h ? byB : byA, h ? alB : alA, h ? atB : atA);
      }
      WV_LANES_END
      // lane (half h, mode m = (lane & 31) >> 1, rows 2 p and 2 p + 1 with p = lane & 1): fetches, stores and costs two rows; a mode's cost is the sum of its
      // two lanes.  The descriptor of (mode, row) sits in lane 4 m + row of `dsc`
      WvLaneArr d0, d1, ct;
#if defined(WH_EMU)
      memset (&d0, 0, sizeof (d0)); memset (&d1, 0, sizeof (d1)); memset (&ct, 0, sizeof (ct));
#else
      d0 = 0; d1 = 0; ct = 0;
#endif
      WV_LSHUF (d0, dsc, lane, ((lane & 31) < 18 ? ((lane & 31) >> 1) * 4 + (lane & 1) * 2 : 0));
      WV_LSHUF (d1, dsc, lane, ((lane & 31) < 18 ? ((lane & 31) >> 1) * 4 + (lane & 1) * 2 + 1 : 0));
      WV_PAIRSUM_TAB (ct, lane, ((lane & 31) < 18 ? ([&] () {
        const int h = lane >> 5, m = (lane & 31) >> 1, r0 = (lane & 1) * 2;
        const uint8_t* tbl = tbl2 + 64 * h;
        uint8_t* pr = h ? predB : S.pred4;
        const int bx = h ? bxB : bxA, by = h ? byB : byA;
        int c = 0;
        for (int k = 0; k < 2; ++k) {
          const uint32_t d = (uint32_t) (k ? WV_LOWN (d1, lane) : WV_LOWN (d0, lane));
          const uint32_t px = (uint32_t)tbl[d & 255u] | ((uint32_t)tbl[(d >> 8) & 255u] << 8) | ((uint32_t)tbl[(d >> 16) & 255u] << 16) | ((uint32_t)tbl[d >> 24] << 24);
          * (uint32_t*)&pr[m * 16 + (r0 + k) * 4] = px;
          c += wh_sad4 (* (const uint32_t*)&S.enc_y[(by * 4 + r0 + k) * 16 + bx * 4], px);
        }
        return c; }) () : 0));
      WV_SYNC();
      WH_PROF_SUB_I (0);      /* detail: a pair step's filter tables, candidates and costs */
      int bmA, bcA, bmB, bcB;
      decide (ct, 0, 2, alA, atA, atlA, atrA, pmA, bmA, bcA);
      decide (ct, 32, 2, alB, atB, atlB, atrB, pmB, bmB, bcB);
      WH_PROF_SUB_I (1);      /* detail: both decision trees */
      cost4 += bcA + bcB;
      if (cost4 >= cost_luma) { completed = false; return false; }
      if (pmA == bmA) prev_flags |= (uint16_t) (1u << bA);
      if (pmB == bmB) prev_flags |= (uint16_t) (1u << bB);
      WV_LSET (i4t, (byA + 1) * 5 + bxA + 1, bmA);
      WV_LSET (i4t, (byB + 1) * 5 + bxB + 1, bmB);
      const int remA = (pmA == bmA) ? 0 : (bmA < pmA) ? bmA : bmA - 1, remB = (pmB == bmB) ? 0 : (bmB < pmB) ? bmB : bmB - 1;
      int nzA, nzB;
      wh_encrec_i4_pair (S, bA, bB, bmA, bmB, qp, (int8_t)remA, (int8_t)remB, predB, &nzA, &nzB);
      WH_PROF_SUB_I (8);      /* detail: both blocks' transform, quantisation, reconstruction */
      if (nzA > 0) cbp |= 1 << (bA >> 2);
      if (nzB > 0) cbp |= 1 << (bB >> 2);
      return true;
    };
    if (use_satd) {
      for (int b = 0; b < 16; ++b) if (!single_block (b)) break;
    } else {
      // (the steps' blocks as nibbles, the second one 15 + 1 = none; luma4x4BlkIdx: (2,0) = 4, (0,1) = 2, (3,0) = 5, (1,1) = 3, (2,1) = 6, (0,2) = 8, ...)
      const unsigned long long first = 0xFEBA763210ULL, second = 0x00DC985400ULL;
      for (int st = 0; st < 10; ++st) {
        const int bA = (int) ((first >> (4 * st)) & 15ULL), bB = (int) ((second >> (4 * st)) & 15ULL);
        if (!(bB ? block_pair (bA, bB) : single_block (bA))) break;
      }
    }
    if (completed) cost4 += (lambda << 4) + (lambda << 3);
    if (completed && cost4 < cost_luma) {
      mb_type = WH_MB_I4x4;
      cost_luma = cost4;
      WV_LANES_BEGIN (lane)
      if (lane == 0) S.i4_prev = prev_flags;
      WV_LANES_END
    }
  }
  WH_PROF_MARK_I (4);       // the sixteen Intra4x4 blocks
  if (mb_type == WH_MB_I16x16) {
    if (last_mode != best_mode) wh_pred_i16 (S, best_mode, sum_t, sum_l, pl_b, pl_c, pl_a);
    cbp = wh_encrec_i16 (S, qp);
  }
  WH_PROF_MARK_I (5);       // Intra16x16 encode (when it won)

  // ---------------- chroma ----------------
  // The twelve edge sums (top / left halves of both planes, the plane mode's gradients) as quad sums of ONE pass, then two candidates per pass side by
  // side on the halves of the wave: lane = (candidate, 4x4 block, row), predictions and source rows as packed words in registers, only the winner goes to
  // S.pred_c (round 6; before: a pass over LDS bytes per candidate and a wave sum per edge sum).  Order and strict '<' as WelsMdIntraChroma (md.cpp:391-433).
  WvLaneArr es;
  WV_QUADSUM_TAB (es, lane, wh_chroma_edge_term (S, lane, has_t, has_l, av3 == 7));
  int st[4], sl[4], cpa[2] = {0, 0}, cpb[2] = {0, 0}, cpc[2] = {0, 0};
  for (int i = 0; i < 4; ++i) { st[i] = WV_LGET (es, 4 * i); sl[i] = WV_LGET (es, 16 + 4 * i); }
  if (av3 == 7) {
    for (int pl = 0; pl < 2; ++pl) {
      const int h = WV_LGET (es, 32 + 4 * pl), v = WV_LGET (es, 40 + 4 * pl);
      cpa[pl] = (WH_RC (S, pl, -1, 7) + WH_RC (S, pl, 7, -1)) << 4;
      cpb[pl] = (17 * h + 16) >> 5;
      cpc[pl] = (17 * v + 16) >> 5;
    }
  }
  int q0, q1, q2, q3, nc;
  if (has_l && has_t) { q0 = WH_C_V; q1 = WH_C_H; q2 = WH_C_DC; q3 = WH_C_P; nc = (av3 == 7) ? 4 : 3; }
  else if (has_l) { q0 = WH_C_DC_L; q1 = WH_C_H; q2 = q0; q3 = q0; nc = 2; }
  else if (has_t) { q0 = WH_C_DC_T; q1 = WH_C_V; q2 = q0; q3 = q0; nc = 2; }
  else { q0 = WH_C_DC_128; q1 = q0; q2 = q0; q3 = q0; nc = 1; }
  WvLaneArr pr0, pr1;
  WV_DECLARE_LANE (lane);
  int cc0, cc1 = 0x7fffffff, cc2 = 0x7fffffff, cc3 = 0x7fffffff;
#define WH_CPOS(lane) const int pl_ = ((lane) >> 4) & 1, bb_ = ((lane) >> 2) & 3, r_ = (lane) & 3
#define WH_CENC(lane) wh_ld4u (S.enc_c, (((lane) >> 4) & 1) * 64 + (((((lane) >> 2) & 3) >> 1) * 4 + ((lane) & 3)) * 8 + ((((lane) >> 2) & 3) & 1) * 4)
  WV_LANE_EVAL (lane, WH_CPOS (lane);
                WV_LOWN (pr0, lane) = (int)wh_pred_chroma4 (S, lane < 32 ? q0 : q1, pl_, bb_, r_, pl_ ? st[2] : st[0], pl_ ? st[3] : st[1], pl_ ? sl[2] : sl[0], pl_ ? sl[3] : sl[1],
                                                            pl_ ? cpa[1] : cpa[0], pl_ ? cpb[1] : cpb[0], pl_ ? cpc[1] : cpc[0]));
  if (use_satd) WV_SATD_ROWS_HALVES (cc0, cc1, lane, true, WH_CENC (lane), (uint32_t)WV_LOWN (pr0, lane));
  else { int d0, d1, d2, d3; WV_ROWSUM4 (d0, d1, d2, d3, lane, wh_sad4 (WH_CENC (lane), (uint32_t)WV_LOWN (pr0, lane))); cc0 = d0 + d1; cc1 = d2 + d3; }
  if (nc > 2) {
    WV_LANE_EVAL (lane, WH_CPOS (lane);
                  WV_LOWN (pr1, lane) = (int)wh_pred_chroma4 (S, lane < 32 ? q2 : q3, pl_, bb_, r_, pl_ ? st[2] : st[0], pl_ ? st[3] : st[1], pl_ ? sl[2] : sl[0], pl_ ? sl[3] : sl[1],
                                                              pl_ ? cpa[1] : cpa[0], pl_ ? cpb[1] : cpb[0], pl_ ? cpc[1] : cpc[0]));
    if (use_satd) WV_SATD_ROWS_HALVES (cc2, cc3, lane, true, WH_CENC (lane), (uint32_t)WV_LOWN (pr1, lane));
    else { int d0, d1, d2, d3; WV_ROWSUM4 (d0, d1, d2, d3, lane, wh_sad4 (WH_CENC (lane), (uint32_t)WV_LOWN (pr1, lane))); cc2 = d0 + d1; cc3 = d2 + d3; }
  } else {
    WV_LANE_EVAL (lane, WV_LOWN (pr1, lane) = 0);
  }
  int cbest = q0, cbest_cost = 0x7fffffff, cidx = 0;
  for (int i = 0; i < nc; ++i) {
    const int m = i == 0 ? q0 : i == 1 ? q1 : i == 2 ? q2 : q3;
    // lambda * BsSizeUE (g_kiMapModeIntraChroma[mode]): DC* -> 1 bit, H/V -> 3, Plane -> 5
    const int bits = (m == WH_C_H || m == WH_C_V) ? 3 : (m == WH_C_P) ? 5 : 1;
    const int c = (i == 0 ? cc0 : i == 1 ? cc1 : i == 2 ? cc2 : cc3) + lambda * bits;
    if (c < cbest_cost) { cbest_cost = c; cbest = m; cidx = i; }
  }
  WV_LANES_BEGIN (lane)
  if ((lane >> 5) == (cidx & 1)) {
    WH_CPOS (lane);
    * (uint32_t*)&S.pred_c[pl_ * 64 + ((bb_ >> 1) * 4 + r_) * 8 + (bb_ & 1) * 4] = (uint32_t) ((cidx & 2) ? WV_LOWN (pr1, lane) : WV_LOWN (pr0, lane));
  }
  WV_LANES_END
#undef WH_CPOS
#undef WH_CENC
  int cbp_c = wh_encrec_chroma (S, qpc, 1);
  wh_idct_chroma (S);
  if (mb_type == WH_MB_I4x4 && stale_cbp) {      // WhMbCtl::stale_cbp: only Intra4x4 keeps what an earlier pass left
    cbp |= stale_cbp & 15;
    const int c0 = stale_cbp >> 4;
    if (cbp_c != 2) cbp_c = (c0 == 2) ? 2 : (cbp_c | c0);
  }

  WH_PROF_MARK_I (6);       // chroma: mode decision + encode
#undef WH_PROF_MARK_I
#undef WH_PROF_SUB_I
  o->mb_type = mb_type;
  o->cbp = cbp | (cbp_c << 4);
  o->i16_mode_std = (best_mode <= 3) ? best_mode : 2;
  o->chroma_mode_std = (cbest <= 3) ? cbest : 0;
  o->cost_luma = cost_luma;
  o->cost_chroma = cbest_cost;
  (void)mbx; (void)mby;
  return true;
}
WH_FN void wh_intra_md_enc (WhMbLds& S, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, int avail, int qp, int qpc,
                            WhIntraResult* o, int stale_cbp = 0) {
  (void)wh_intra_md_enc_p (S, P, J, mbx, mby, avail, qp, qpc, 0x7fffffff, o, nullptr, stale_cbp);
}
