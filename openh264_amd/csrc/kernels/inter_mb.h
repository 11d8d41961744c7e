// inter_mb.h -- mode decision, motion estimation and reconstruction of one P-slice macroblock by
// one wavefront.
//
// Reference behaviour restated (codec/encoder/core/src unless noted):
//   svc_base_layer_md.cpp:1858-1904  WelsMdInterMb            (decision tree)
//   svc_base_layer_md.cpp:1906-1921  WelsMdInterJudgePskip
//   svc_base_layer_md.cpp:1423-1540  WelsMdPSkipEnc           (skip test incl. quant-to-zero checks)
//   svc_base_layer_md.cpp:978-1118   WelsMdP16x16 / P16x8 / P8x16 / P8x8
//   svc_base_layer_md.cpp:1238-1339  WelsMdInterFinePartition[Vaa]
//   svc_base_layer_md.cpp:1573-1827  WelsMdInterMbRefinement
//   svc_base_layer_md.cpp:1829-1856  WelsMdFirstIntraMode
//   svc_base_layer_md.cpp:1937-1995  WelsMdInterDoubleCheckPskip / WelsMdInterEncode / SaveSadAndRefMbType
//   svc_motion_estimate.cpp:170-379  WelsMotionEstimateSearch / InitialPoint / WelsDiamondSearch
//   md.cpp:132-253                   FillNeighborCacheInterWithoutBGD
//   md.cpp:389-433                   MdInterAnalysisVaaInfo_c
//   md.cpp:575-769                   MeRefineFracPixel / MeRefineQuarPixel
//   md.cpp:797-910                   MvdCostInit / PredictSad / PredictSadSkip
//   mv_pred.cpp:45-147               PredMv / PredInter16x8Mv / PredInter8x16Mv / PredSkipMv
//   svc_encode_mb.cpp:180-242,325-381 WelsEncInterY / WelsTryPYskip / WelsTryPUVskip
//   codec/common/src/mc.cpp:100-386  luma 6-tap quarter-pel + chroma bilinear interpolation (= H.264 8.4.2.2)
//   codec/processing/src/vaacalc/vaacalcfuncs.cpp:254-330 VAACalcSad_c (8x8 SADs vs the previous source frame)
#pragma once
#include "frame_kernels.h"

#define WH_REF_NOT_AVAIL (-2)
#define WH_REF_NOT_IN_LIST (-1)
#define WH_WIN_STRIDE 64
#define WH_WIN_ROWS 56
#define WH_WIN_MARGIN 19          // diamond (16) + quarter/half-pel taps (3)

typedef struct WhInterLds {
  WhMbLds m;
  uint8_t win[WH_WIN_ROWS * WH_WIN_STRIDE];   // reference search window (luma), see wh_win_load
  uint8_t cand[256];                          // candidate luma prediction, stride 16
  uint8_t skip_y[256];                        // P_Skip prediction
  uint8_t skip_c[128];
  int16_t mvc[30][2];                         // motion vector cache, 5 rows x 6 cols (row 0 / col 0 = neighbours)
  int8_t  refc[32];                           // reference index cache
  int16_t mvp_out[16][2];                     // predictor used for the mvd of each 4x4 (raster)
  int16_t mv_out[16][2];
} WhInterLds;

// ---- mvd cost: lambda * bits(se(mvd))  (md.cpp:797-824, svc_enc_golomb.h BsSizeSE) --------------
WH_FN int wh_se_bits (int v) {
  if (v == 0) return 1;
  unsigned k = (unsigned) (v > 0 ? 2 * v - 1 : -2 * v) + 1u;
  int n = 0;
  while (k > 1) { k >>= 1; ++n; }
  return 2 * n + 1;
}
WH_FN int wh_mvd_cost (int lambda, int dx, int dy) { return (int) (uint16_t) (lambda * wh_se_bits (dx)) + (int) (uint16_t) (lambda * wh_se_bits (dy)); }

// ---- H.264 luma sample interpolation from a byte tile (stride st), integer position p, frac (fx,fy) ----
WH_FN int wh_tap6 (int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
WH_FN int wh_mc_b (const uint8_t* p) { return wh_clip255 ((wh_tap6 (p[-2], p[-1], p[0], p[1], p[2], p[3]) + 16) >> 5); }
WH_FN int wh_mc_h (const uint8_t* p, int st) { return wh_clip255 ((wh_tap6 (p[-2 * st], p[-st], p[0], p[st], p[2 * st], p[3 * st]) + 16) >> 5); }
WH_FN int wh_mc_j (const uint8_t* p, int st) {
  int v[6];
  for (int k = 0; k < 6; ++k) { const uint8_t* r = p + (k - 2) * st; v[k] = wh_tap6 (r[-2], r[-1], r[0], r[1], r[2], r[3]); }
  return wh_clip255 ((wh_tap6 (v[0], v[1], v[2], v[3], v[4], v[5]) + 512) >> 10);
}
WH_FN int wh_mc_luma_px (const uint8_t* p, int st, int fx, int fy) {
  switch (fy * 4 + fx) {
  case 0: return p[0];
  case 1: return (p[0] + wh_mc_b (p) + 1) >> 1;
  case 2: return wh_mc_b (p);
  case 3: return (p[1] + wh_mc_b (p) + 1) >> 1;
  case 4: return (p[0] + wh_mc_h (p, st) + 1) >> 1;
  case 5: return (wh_mc_b (p) + wh_mc_h (p, st) + 1) >> 1;
  case 6: return (wh_mc_b (p) + wh_mc_j (p, st) + 1) >> 1;
  case 7: return (wh_mc_b (p) + wh_mc_h (p + 1, st) + 1) >> 1;
  case 8: return wh_mc_h (p, st);
  case 9: return (wh_mc_h (p, st) + wh_mc_j (p, st) + 1) >> 1;
  case 10: return wh_mc_j (p, st);
  case 11: return (wh_mc_j (p, st) + wh_mc_h (p + 1, st) + 1) >> 1;
  case 12: return (p[st] + wh_mc_h (p, st) + 1) >> 1;
  case 13: return (wh_mc_h (p, st) + wh_mc_b (p + st) + 1) >> 1;
  case 14: return (wh_mc_j (p, st) + wh_mc_b (p + st) + 1) >> 1;
  default: return (wh_mc_h (p + 1, st) + wh_mc_b (p + st) + 1) >> 1;
  }
}
// chroma (mc.cpp:349-378): bilinear with eighth-sample weights
WH_FN int wh_mc_chroma_px (const uint8_t* p, int st, int dx, int dy) {
  return ((8 - dx) * (8 - dy) * p[0] + dx * (8 - dy) * p[1] + (8 - dx) * dy * p[st] + dx * dy * p[st + 1] + 32) >> 6;
}

// ---- reference window ---------------------------------------------------------------------------
// Loads luma pixels [px-19, px+bw+19) x [py-19, py+bh+19) of the reference picture (picture
// coordinates) into S.win.  Returns the picture coordinates of window element (0,0) in *ox,*oy
// (ox is aligned down to 4 for word loads).
WH_FN void wh_win_load (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, int px, int py, int bw, int bh, int* ox, int* oy) {
  const int x0 = (px - WH_WIN_MARGIN) & ~3, y0 = py - WH_WIN_MARGIN;
  const int words = ((px + bw + WH_WIN_MARGIN) - x0 + 3) >> 2;       // <= 16
  const int rows = bh + 2 * WH_WIN_MARGIN;                            // <= 54
  WV_LANES_BEGIN (lane)
  for (int i = lane; i < rows * 16; i += 64) {
    const int r = i >> 4, wd = i & 15;
    if (wd < words) {
      const uint8_t* s = J.ref[0] + (ptrdiff_t) (y0 + r) * P.rec_stride_y + x0 + wd * 4;
      * (uint32_t*)&S.win[r * WH_WIN_STRIDE + wd * 4] = * (const uint32_t*)s;
    }
  }
  WV_LANES_END
  *ox = x0; *oy = y0;
}

// SAD of a bw x bh block of enc (at ex,ey inside the MB) against a byte tile; lanes = bw*bh/4
WH_FN int wh_sad_tile (const WhInterLds& S, int ex, int ey, int bw, int bh, const uint8_t* t, int st) {
  int s;
  const int per_row = bw >> 2, n = per_row * bh;
  WV_SUM (s, lane, (lane < n ? (wh_abs (S.m.enc_y[(ey + lane / per_row) * 16 + ex + (lane % per_row) * 4 + 0] - t[(lane / per_row) * st + (lane % per_row) * 4 + 0]) +
                                 wh_abs (S.m.enc_y[(ey + lane / per_row) * 16 + ex + (lane % per_row) * 4 + 1] - t[(lane / per_row) * st + (lane % per_row) * 4 + 1]) +
                                 wh_abs (S.m.enc_y[(ey + lane / per_row) * 16 + ex + (lane % per_row) * 4 + 2] - t[(lane / per_row) * st + (lane % per_row) * 4 + 2]) +
                                 wh_abs (S.m.enc_y[(ey + lane / per_row) * 16 + ex + (lane % per_row) * 4 + 3] - t[(lane / per_row) * st + (lane % per_row) * 4 + 3])) : 0));
  return s;
}

// SATD (4x4 Hadamard, rounded per 4x4) of a bw x bh block of enc against a tile with stride st
WH_FN int wh_satd_tile (WhInterLds& S, int ex, int ey, int bw, int bh, const uint8_t* t, int st) {
  const int nb = (bw >> 2) * (bh >> 2), bpr = bw >> 2;      // 4x4 blocks
  WV_LANES_BEGIN (lane)
  if (lane < nb * 4) {
    const int b = lane >> 2, r = lane & 3;
    const int bx = (b % bpr) * 4, by = (b / bpr) * 4 + r;
    const uint8_t* e = &S.m.enc_y[(ey + by) * 16 + ex + bx];
    const uint8_t* p = &t[by * st + bx];
    int o0, o1, o2, o3;
    wh_had4 (e[0] - p[0], e[1] - p[1], e[2] - p[2], e[3] - p[3], &o0, &o1, &o2, &o3);
    int16_t* q = &S.m.tmp[b * 16 + r * 4];
    q[0] = (int16_t)o0; q[1] = (int16_t)o1; q[2] = (int16_t)o2; q[3] = (int16_t)o3;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < nb * 4) {
    const int b = lane >> 2, c = lane & 3;
    const int16_t* q = &S.m.tmp[b * 16 + c];
    int o0, o1, o2, o3;
    wh_had4 (q[0], q[4], q[8], q[12], &o0, &o1, &o2, &o3);
    S.m.part[lane] = wh_abs (o0) + wh_abs (o1) + wh_abs (o2) + wh_abs (o3);
  }
  WV_LANES_END
  int s;
  WV_SUM (s, lane, (lane < nb ? ((S.m.part[lane * 4] + S.m.part[lane * 4 + 1] + S.m.part[lane * 4 + 2] + S.m.part[lane * 4 + 3] + 1) >> 1) : 0));
  return s;
}

// Build the luma prediction of a bw x bh block for quarter-pel mv (relative to block position
// bpx,bpy in picture coords) from the window into dst (stride dst_st).
WH_FN void wh_mc_luma_from_win (WhInterLds& S, int ox, int oy, int bpx, int bpy, int mvx, int mvy, int bw, int bh, uint8_t* dst, int dst_st) {
  const int ix = bpx + (mvx >> 2) - ox, iy = bpy + (mvy >> 2) - oy, fx = mvx & 3, fy = mvy & 3;
  const int per_row = bw >> 2, n = per_row * bh;
  WV_LANES_BEGIN (lane)
  if (lane < n) {
    const int r = lane / per_row, c = (lane % per_row) * 4;
    const uint8_t* p = &S.win[(iy + r) * WH_WIN_STRIDE + ix + c];
    for (int k = 0; k < 4; ++k) dst[r * dst_st + c + k] = (uint8_t)wh_mc_luma_px (p + k, WH_WIN_STRIDE, fx, fy);
  }
  WV_LANES_END
}

// Luma prediction straight from the reference picture in HBM (P_Skip test: one-off position).
WH_FN void wh_mc_luma_from_ref (const WhSeqParams& P, const WhPicJob& J, int bpx, int bpy, int mvx, int mvy, uint8_t* dst) {
  const int fx = mvx & 3, fy = mvy & 3;
  const uint8_t* base = J.ref[0] + (ptrdiff_t) (bpy + (mvy >> 2)) * P.rec_stride_y + bpx + (mvx >> 2);
  WV_LANES_BEGIN (lane)
  const int r = lane >> 2, c = (lane & 3) * 4;
  const uint8_t* p = base + (ptrdiff_t)r * P.rec_stride_y + c;
  for (int k = 0; k < 4; ++k) dst[r * 16 + c + k] = (uint8_t)wh_mc_luma_px (p + k, P.rec_stride_y, fx, fy);
  WV_LANES_END
}

// Chroma prediction of a cw x ch block (both planes) at chroma block position (cx,cy) inside the MB.
WH_FN void wh_mc_chroma (const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, int cx, int cy, int cw, int ch, int mvx, int mvy, uint8_t* dst /*Cb at 0, Cr at 64, stride 8*/) {
  const int dx = mvx & 7, dy = mvy & 7;
  const int n = cw * ch;
  WV_LANES_BEGIN (lane)
  for (int i = lane; i < 2 * n; i += 64) {
    const int pl = i / n, k = i % n, x = cx + k % cw, y = cy + k / cw;
    const uint8_t* p = J.ref[1 + pl] + (ptrdiff_t) (mby * 8 + y + (mvy >> 3)) * P.rec_stride_c + mbx * 8 + x + (mvx >> 3);
    dst[pl * 64 + y * 8 + x] = (uint8_t)wh_mc_chroma_px (p, P.rec_stride_c, dx, dy);
  }
  WV_LANES_END
}

// ---- motion vector prediction on the 5x6 cache ----------------------------------------------------
WH_FN int wh_cidx (int bx, int by) { return (by + 1) * 6 + bx + 1; }
WH_FN void wh_pred_mv (const WhInterLds& S, int bx, int by, int w, int ref, int* mx, int* my) {
  const int li = (by + 1) * 6 + bx, ti = by * 6 + bx + 1;
  const int lref = S.refc[li], tref = S.refc[ti];
  int di = ti + w;
  if (S.refc[di] == WH_REF_NOT_AVAIL) di = ti - 1;
  const int dref = S.refc[di];
  if (tref == WH_REF_NOT_AVAIL && dref == WH_REF_NOT_AVAIL && lref != WH_REF_NOT_AVAIL) { *mx = S.mvc[li][0]; *my = S.mvc[li][1]; return; }
  const int match = (ref == lref) | ((ref == tref) << 1) | ((ref == dref) << 2);
  if (match == 1) { *mx = S.mvc[li][0]; *my = S.mvc[li][1]; }
  else if (match == 2) { *mx = S.mvc[ti][0]; *my = S.mvc[ti][1]; }
  else if (match == 4) { *mx = S.mvc[di][0]; *my = S.mvc[di][1]; }
  else { *mx = wh_median3 (S.mvc[li][0], S.mvc[ti][0], S.mvc[di][0]); *my = wh_median3 (S.mvc[li][1], S.mvc[ti][1], S.mvc[di][1]); }
}
WH_FN void wh_pred_16x8 (const WhInterLds& S, int part, int ref, int* mx, int* my) {
  if (part == 0) { if (ref == S.refc[1]) { *mx = S.mvc[1][0]; *my = S.mvc[1][1]; return; } }
  else { if (ref == S.refc[18]) { *mx = S.mvc[18][0]; *my = S.mvc[18][1]; return; } }
  wh_pred_mv (S, 0, part * 2, 4, ref, mx, my);
}
WH_FN void wh_pred_8x16 (const WhInterLds& S, int part, int ref, int* mx, int* my) {
  if (part == 0) { if (ref == S.refc[6]) { *mx = S.mvc[6][0]; *my = S.mvc[6][1]; return; } }
  else {
    int idx = 5;
    if (S.refc[5] == WH_REF_NOT_AVAIL) idx = 2;
    if (ref == S.refc[idx]) { *mx = S.mvc[idx][0]; *my = S.mvc[idx][1]; return; }
  }
  wh_pred_mv (S, part * 2, 0, 2, ref, mx, my);
}
WH_FN void wh_pred_skip_mv (const WhInterLds& S, int* mx, int* my) {
  const int lref = S.refc[6], tref = S.refc[1];
  if (lref == WH_REF_NOT_AVAIL || tref == WH_REF_NOT_AVAIL || (lref == 0 && S.mvc[6][0] == 0 && S.mvc[6][1] == 0) ||
      (tref == 0 && S.mvc[1][0] == 0 && S.mvc[1][1] == 0)) { *mx = 0; *my = 0; return; }
  wh_pred_mv (S, 0, 0, 4, 0, mx, my);
}
// write mv/ref into the cache rectangle (bx,by,w,h in 4x4 units)
WH_FN void wh_cache_set (WhInterLds& S, int bx, int by, int w, int h, int ref, int mx, int my) {
  WV_LANES_BEGIN (lane)
  if (lane < w * h) {
    const int i = wh_cidx (bx + lane % w, by + lane / w);
    S.refc[i] = (int8_t)ref; S.mvc[i][0] = (int16_t)mx; S.mvc[i][1] = (int16_t)my;
  }
  WV_LANES_END
}

// ---- one motion search (WelsMotionEstimateSearch) -------------------------------------------------
typedef struct WhMe {
  int bx, by, bw, bh;        // block inside the MB (pixels)
  int mvpx, mvpy;            // predictor (quarter-pel)
  int sad_pred;              // uiSadPred
  int mvx, mvy;              // result (quarter-pel)
  int sad_cost, satd_cost;   // uiSadCost / uiSatdCost
  int satd_raw;              // uSadPredISatd.uiSatd (complexity >= MEDIUM)
  int ox, oy;                // window origin of the last load
} WhMe;

typedef struct WhMeCtx {
  int mbx, mby, lambda, use_satd;
  int minx, miny, maxx, maxy;        // sMvStartMin / sMvStartMax (integer pel)
} WhMeCtx;

WH_FN int wh_sad_ref_global (const WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, const WhMe& me, const WhMeCtx& C, int imx, int imy) {
  const uint8_t* base = J.ref[0] + (ptrdiff_t) (C.mby * 16 + me.by + imy) * P.rec_stride_y + C.mbx * 16 + me.bx + imx;
  return wh_sad_tile (S, me.bx, me.by, me.bw, me.bh, base, P.rec_stride_y);
}

WH_FN void wh_motion_search (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, const WhMeCtx& C, WhMe& me,
                             const int16_t (*mvc_list)[2], int n_mvc) {
  // initial point (svc_motion_estimate.cpp:222-284)
  int bmx = wh_clip3 ((2 + me.mvpx) >> 2, C.minx, C.maxx), bmy = wh_clip3 ((2 + me.mvpy) >> 2, C.miny, C.maxy);
  int best = wh_sad_ref_global (S, P, J, me, C, bmx, bmy) + wh_mvd_cost (C.lambda, bmx * 4 - me.mvpx, bmy * 4 - me.mvpy);
  for (int i = 0; i < n_mvc; ++i) {
    const int cx = wh_clip3 ((2 + mvc_list[i][0]) >> 2, C.minx, C.maxx), cy = wh_clip3 ((2 + mvc_list[i][1]) >> 2, C.miny, C.maxy);
    if (cx != bmx || cy != bmy) {
      const int c = wh_sad_ref_global (S, P, J, me, C, cx, cy) + wh_mvd_cost (C.lambda, cx * 4 - me.mvpx, cy * 4 - me.mvpy);
      if (c < best) { best = c; bmx = cx; bmy = cy; }
    }
  }
  // the window serves the diamond search and the later fractional refinement
  wh_win_load (S, P, J, C.mbx * 16 + me.bx + bmx, C.mby * 16 + me.by + bmy, me.bw, me.bh, &me.ox, &me.oy);
  if (!(best < me.sad_pred)) {
    // WelsDiamondSearch (svc_motion_estimate.cpp:335-379)
    int dx = bmx * 4 - me.mvpx, dy = bmy * 4 - me.mvpy;
    int px = C.mbx * 16 + me.bx + bmx - me.ox, py = C.mby * 16 + me.by + bmy - me.oy;     // position inside the window
    for (int it = 0; it < 16; ++it) {
      const int cmx = (dx + me.mvpx) >> 2, cmy = (dy + me.mvpy) >> 2;
      if (!(cmx >= C.minx && cmx < C.maxx && cmy >= C.miny && cmy < C.maxy)) continue;
      const uint8_t* t = &S.win[py * WH_WIN_STRIDE + px];
      // four SADs: up, down, left, right -- two packed 16-bit partial sums per reduction
      int pud, plr;
      const int per_row = me.bw >> 2, n = per_row * me.bh;
      WV_SUM (pud, lane, (lane < n ? ([&] () { const int r = lane / per_row, c = (lane % per_row) * 4; const uint8_t* e = &S.m.enc_y[(me.by + r) * 16 + me.bx + c];
                                                const uint8_t* u = t + (r - 1) * WH_WIN_STRIDE + c; const uint8_t* d = t + (r + 1) * WH_WIN_STRIDE + c;
                                                int su = 0, sd = 0; for (int k = 0; k < 4; ++k) { su += wh_abs (e[k] - u[k]); sd += wh_abs (e[k] - d[k]); } return su | (sd << 16); }) () : 0));
      WV_SUM (plr, lane, (lane < n ? ([&] () { const int r = lane / per_row, c = (lane % per_row) * 4; const uint8_t* e = &S.m.enc_y[(me.by + r) * 16 + me.bx + c];
                                                const uint8_t* l = t + r * WH_WIN_STRIDE + c - 1; const uint8_t* rr = t + r * WH_WIN_STRIDE + c + 1;
                                                int sl = 0, sr = 0; for (int k = 0; k < 4; ++k) { sl += wh_abs (e[k] - l[k]); sr += wh_abs (e[k] - rr[k]); } return sl | (sr << 16); }) () : 0));
      const int c0 = (pud & 0xffff) + wh_mvd_cost (C.lambda, dx, dy - 4);
      const int c1 = ((unsigned)pud >> 16) + wh_mvd_cost (C.lambda, dx, dy + 4);
      const int c2 = (plr & 0xffff) + wh_mvd_cost (C.lambda, dx - 4, dy);
      const int c3 = ((unsigned)plr >> 16) + wh_mvd_cost (C.lambda, dx + 4, dy);
      const int in_cost = best;
      int ix = 0, iy = 0;
      if (c0 < best) { best = c0; ix = 0; iy = 1; }
      if (c1 < best) { best = c1; ix = 0; iy = -1; }
      if (c2 < best) { best = c2; ix = 1; iy = 0; }
      if (c3 < best) { best = c3; ix = -1; iy = 0; }
      if (best == in_cost) break;
      dx -= ix * 4; dy -= iy * 4;
      px -= ix; py -= iy;
    }
    bmx = (dx + me.mvpx) >> 2; bmy = (dy + me.mvpy) >> 2;
  }
  me.mvx = bmx * 4; me.mvy = bmy * 4;
  me.sad_cost = best; me.satd_cost = best; me.satd_raw = 0;
  if (C.use_satd) {   // CalculateSatdCost (complexity >= MEDIUM)
    const uint8_t* t = &S.win[(C.mby * 16 + me.by + bmy - me.oy) * WH_WIN_STRIDE + C.mbx * 16 + me.bx + bmx - me.ox];
    me.satd_raw = wh_satd_tile (S, me.bx, me.by, me.bw, me.bh, t, WH_WIN_STRIDE);
    me.satd_cost = me.satd_raw + wh_mvd_cost (C.lambda, me.mvx - me.mvpx, me.mvy - me.mvpy);
  }
}

// ---- fractional refinement (MeRefineFracPixel): returns through me.mvx/mvy/satd_cost, writes the
// final luma prediction of the block into S.m.pred_y -------------------------------------------------
WH_FN void wh_refine_frac (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, const WhMeCtx& C, WhMe& me, int satd_in_md) {
  const int bpx = C.mbx * 16 + me.bx, bpy = C.mby * 16 + me.by;
  // (re)load the window around the integer result: the other partitions' searches have reused S.win
  wh_win_load (S, P, J, bpx + (me.mvx >> 2), bpy + (me.mvy >> 2), me.bw, me.bh, &me.ox, &me.oy);
  uint8_t* dst = &S.m.pred_y[me.by * 16 + me.bx];
  int best;
  if (satd_in_md) best = me.satd_raw + wh_mvd_cost (C.lambda, me.mvx - me.mvpx, me.mvy - me.mvpy);   // uiSatd of the integer search
  else {
    wh_mc_luma_from_win (S, me.ox, me.oy, bpx, bpy, me.mvx, me.mvy, me.bw, me.bh, S.cand, 16);
    best = wh_satd_tile (S, me.bx, me.by, me.bw, me.bh, S.cand, 16) + wh_mvd_cost (C.lambda, me.mvx - me.mvpx, me.mvy - me.mvpy);
  }
  int bmx = me.mvx, bmy = me.mvy;
  // half-pel candidates: top, bottom, left, right
  const int hdx[4] = {0, 0, -2, 2}, hdy[4] = {-2, 2, 0, 0};
  int hbest = -1;
  for (int k = 0; k < 4; ++k) {
    const int cx = me.mvx + hdx[k], cy = me.mvy + hdy[k];
    wh_mc_luma_from_win (S, me.ox, me.oy, bpx, bpy, cx, cy, me.bw, me.bh, S.cand, 16);
    const int c = wh_satd_tile (S, me.bx, me.by, me.bw, me.bh, S.cand, 16) + wh_mvd_cost (C.lambda, cx - me.mvpx, cy - me.mvpy);
    if (c < best) { best = c; hbest = k; }
  }
  const int hx = hbest < 0 ? me.mvx : me.mvx + hdx[hbest], hy = hbest < 0 ? me.mvy : me.mvy + hdy[hbest];
  bmx = hx; bmy = hy;
  // quarter-pel candidates around the best half/integer position: top, bottom, left, right
  const int qdx[4] = {0, 0, -1, 1}, qdy[4] = {-1, 1, 0, 0};
  for (int k = 0; k < 4; ++k) {
    const int cx = hx + qdx[k], cy = hy + qdy[k];
    wh_mc_luma_from_win (S, me.ox, me.oy, bpx, bpy, cx, cy, me.bw, me.bh, S.cand, 16);
    const int c = wh_satd_tile (S, me.bx, me.by, me.bw, me.bh, S.cand, 16) + wh_mvd_cost (C.lambda, cx - me.mvpx, cy - me.mvpy);
    if (c < best) { best = c; bmx = cx; bmy = cy; }
  }
  me.mvx = bmx; me.mvy = bmy; me.satd_cost = best;
  wh_mc_luma_from_win (S, me.ox, me.oy, bpx, bpy, bmx, bmy, me.bw, me.bh, dst, 16);
}

// ---- inter luma residual (WelsEncInterY) on S.m.res after wh_dct_luma16; returns cbp luma -----------
WH_FN int wh_enc_inter_y (WhMbLds& S, int qp) {
  // quant with per-block max (inter rounding)
  WV_LANES_BEGIN (lane)
  {
    int16_t mx = 0;
    for (int k = 0; k < 4; ++k) {
      const int i = lane * 4 + k, pos = i & 15;
      int16_t a;
      S.res[i] = wh_quant1_abs (S.res[i], wh_ff_inter (qp, pos), wh_mf (qp, pos), &a);
      if (mx < a) mx = a;
    }
    S.part[lane] = mx;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  if (lane < 16) S.amax[lane] = (int16_t)wh_max (wh_max (S.part[lane * 4], S.part[lane * 4 + 1]), wh_max (S.part[lane * 4 + 2], S.part[lane * 4 + 3]));
  {
    const int b = lane >> 2;
    for (int q = 0; q < 4; ++q) { const int k = (lane & 3) * 4 + q; S.lv_luma[b * 16 + k] = S.res[b * 16 + wh_zigzag (k)]; }
  }
  WV_LANES_END
  int ctr8[4], ctr_mb = 0;
  for (int i = 0; i < 4; ++i) {
    ctr8[i] = 0;
    for (int j = 0; j < 4; ++j) {
      const int mx = S.amax[i * 4 + j];
      if (mx != 0) {
        if (mx > 1) ctr8[i] += 9;
        else if (ctr8[i] < 6) ctr8[i] += wh_single_ctr (&S.lv_luma[(i * 4 + j) * 16]);
      }
    }
    ctr_mb += ctr8[i];
  }
  int cbp = 0;
  if (ctr_mb >= 6) for (int i = 0; i < 4; ++i) if (ctr8[i] >= 4) cbp |= 1 << i;
  WV_LANES_BEGIN (lane)
  {
    const int b = lane >> 2, on = (cbp >> (b >> 2)) & 1;
    for (int k = 0; k < 4; ++k) {
      const int i = lane * 4 + k, pos = i & 15;
      S.res[i] = on ? (int16_t) (S.res[i] * wh_dq (qp, pos)) : (int16_t)0;
    }
    if (lane < 16) {
      int n = 0;
      if ((cbp >> (lane >> 2)) & 1) for (int k = 0; k < 16; ++k) n += (S.lv_luma[lane * 16 + k] != 0);
      S.nzc[wh_blk_y (lane) * 4 + wh_blk_x (lane)] = (uint8_t)n;
    }
  }
  WV_LANES_END
  return cbp;
}

// quant-to-zero tests of the P_Skip path (WelsTryPYskip / WelsTryPUVskip); operate on copies in S.tmp
WH_FN bool wh_try_py_skip (WhMbLds& S, int qp) {
  WV_LANES_BEGIN (lane)
  {
    int16_t mx = 0;
    for (int k = 0; k < 4; ++k) {
      const int i = lane * 4 + k, pos = i & 15;
      int16_t a;
      S.tmp[i] = wh_quant1_abs (S.res[i], wh_ff_inter (qp, pos), wh_mf (qp, pos), &a);
      if (mx < a) mx = a;
    }
    S.part[lane] = mx;
  }
  WV_LANES_END
  int ctr = 0;
  for (int b = 0; b < 16; ++b) {
    const int mx = wh_max (wh_max (S.part[b * 4], S.part[b * 4 + 1]), wh_max (S.part[b * 4 + 2], S.part[b * 4 + 3]));
    if (mx > 1) return false;
    if (mx == 1) {
      int16_t lv[16];
      for (int k = 0; k < 16; ++k) lv[k] = S.tmp[b * 16 + wh_zigzag (k)];
      ctr += wh_single_ctr (lv);
    }
    if (ctr >= 6) return false;
  }
  return true;
}
WH_FN bool wh_try_puv_skip (WhMbLds& S, int pl, int qpc) {
  const int16_t* r = &S.res[256 + pl * 64];
  // WelsHadamardQuant2x2Skip_c (encode_mb_aux.cpp:226-245)
  const int ff = wh_ff_inter (qpc, 0) << 1, mf = wh_mf (qpc, 0) >> 1;
  const int16_t thr = (int16_t) (((1 << 16) - 1) / mf - ff);
  const int16_t s0 = (int16_t) (r[0] + r[32]), s1 = (int16_t) (r[0] - r[32]), s2 = (int16_t) (r[16] + r[48]), s3 = (int16_t) (r[16] - r[48]);
  const int16_t d0 = (int16_t) (s0 + s2), d1 = (int16_t) (s0 - s2), d2 = (int16_t) (s1 + s3), d3 = (int16_t) (s1 - s3);
  if (wh_abs (d0) > thr || wh_abs (d1) > thr || wh_abs (d2) > thr || wh_abs (d3) > thr) return false;
  WV_LANES_BEGIN (lane)
  if (lane < 16) {
    int16_t mx = 0;
    for (int k = 0; k < 4; ++k) {
      const int i = lane * 4 + k, pos = i & 15;
      int16_t a;
      S.tmp[i] = wh_quant1_abs (r[i], wh_ff_inter (qpc, pos), wh_mf (qpc, pos), &a);
      if (mx < a) mx = a;
    }
    S.part[lane] = mx;
  }
  WV_LANES_END
  int ctr = 0;
  for (int b = 0; b < 4; ++b) {
    const int mx = wh_max (wh_max (S.part[b * 4], S.part[b * 4 + 1]), wh_max (S.part[b * 4 + 2], S.part[b * 4 + 3]));
    if (mx > 1) return false;
    if (mx == 1) {
      int16_t lv[16];
      for (int k = 0; k < 15; ++k) lv[k] = S.tmp[b * 16 + wh_zigzag (k + 1)];
      lv[15] = 0;
      ctr += wh_single_ctr (lv);
    }
    if (ctr >= 7) return false;
  }
  return true;
}

// ---- the P macroblock -----------------------------------------------------------------------------
WH_FN void wh_inter_mb_body (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  WhMbLds& M = S.m;
  const int w = P.mb_w, xy = mby * w + mbx;
  const int avail = wh_mb_avail (P, mbx, mby);
  const int qp = wh_clip3 (J.qp, 0, 51);
  const int qpc = kWhChromaQp[wh_clip3 (qp + P.chroma_qp_offset, 0, 51)];
  const int lambda = kWhLambda[qp];
  const int use_satd = P.complexity > 0;        // pfMdCost == SATD, pfCalculateSatd == CalculateSatdCost
  const bool md_using_sad = !use_satd;          // bMdUsingSad (svc_encode_slice.cpp:699)
  const int slice_idc = wh_slice_of_mb (P, xy);
  WH_PROF_DECL (P);
  wh_load_mb_tile (M, P, J, mbx, mby);

  // ---- neighbour cache (FillNeighborCacheInterWithoutBGD) ----
  const WhMbState* Lm = (avail & WH_AV_LEFT) ? &J.mbs[xy - 1] : nullptr;
  const WhMbState* Tm = (avail & WH_AV_TOP) ? &J.mbs[xy - w] : nullptr;
  const WhMbState* TLm = (avail & WH_AV_TOPLEFT) ? &J.mbs[xy - w - 1] : nullptr;
  const WhMbState* TRm = (avail & WH_AV_TOPRIGHT) ? &J.mbs[xy - w + 1] : nullptr;
  const bool l_inter = Lm && WH_IS_INTER (Lm->mb_type), t_inter = Tm && WH_IS_INTER (Tm->mb_type);
  const bool tl_inter = TLm && WH_IS_INTER (TLm->mb_type), tr_inter = TRm && WH_IS_INTER (TRm->mb_type);
  WV_LANES_BEGIN (lane)
  if (lane < 30) {
    const int r = lane / 6, c = lane % 6;
    int ref = WH_REF_NOT_AVAIL, mx = 0, my = 0;
    if (r == 0 && c == 0) { ref = TLm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; if (tl_inter) { ref = TLm->ref_idx[3]; mx = TLm->mv[15][0]; my = TLm->mv[15][1]; } }
    else if (r == 0 && c == 5) { ref = TRm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; if (tr_inter) { ref = TRm->ref_idx[2]; mx = TRm->mv[12][0]; my = TRm->mv[12][1]; } }
    else if (r == 0) { ref = Tm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; if (t_inter) { ref = Tm->ref_idx[2 + ((c - 1) >> 1)]; mx = Tm->mv[12 + c - 1][0]; my = Tm->mv[12 + c - 1][1]; } }
    else if (c == 0) { ref = Lm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; if (l_inter) { ref = Lm->ref_idx[((r - 1) >> 1) * 2 + 1]; mx = Lm->mv[(r - 1) * 4 + 3][0]; my = Lm->mv[(r - 1) * 4 + 3][1]; } }
    else { ref = WH_REF_NOT_AVAIL; }          // inside the MB: not coded yet (the reference pre-marks 9,11,17,21,23)
    S.refc[lane] = (int8_t)ref; S.mvc[lane][0] = (int16_t)mx; S.mvc[lane][1] = (int16_t)my;
  }
  WV_LANES_END
  // neighbour SAD / skip context, order of the reference's caches: [0] top-left, [1] top, [2] top-right, [3] left
  int sadc[4], skc[4], sadsk[4];
  {
    const WhMbState* nb[4] = {TLm, Tm, TRm, Lm};
    const bool ni[4] = {tl_inter, t_inter, tr_inter, l_inter};
    for (int k = 0; k < 4; ++k) {
      sadc[k] = ni[k] ? nb[k]->sad_cost[0] : 0;
      skc[k] = (ni[k] && nb[k]->mb_type == WH_MB_PSKIP) ? 1 : 0;
      sadsk[k] = skc[k] ? nb[k]->skip_sad : 0;
    }
  }
  const bool ref_is_p = J.ref_is_p != 0;
  const int ref_mb_type = ref_is_p ? J.ref_mbs[xy].mb_type : WH_MB_NONE;
  WhMeCtx C;
  C.mbx = mbx; C.mby = mby; C.lambda = lambda; C.use_satd = use_satd;
  C.minx = wh_max (- ((mbx + 1) << 4) + 3, -P.mv_range); C.miny = wh_max (- ((mby + 1) << 4) + 3, -P.mv_range);
  C.maxx = wh_min (((P.mb_w - mbx) << 4) - 3, P.mv_range); C.maxy = wh_min (((P.mb_h - mby) << 4) - 3, P.mv_range);

  WH_PROF_MARK (P, 0);   // tile + neighbour cache
  int mb_type = WH_MB_P16x16, cbp = 0, cost_luma = 0, cost_skip_mb = 0, sad_cost0 = 0;
  int p16x = 0, p16y = 0;                       // sP16x16Mv
  int skx = 0, sky = 0;
  bool done = false;

  // ---- P_Skip test (WelsMdInterJudgePskip / WelsMdPSkipEnc) ----
  const bool l_sk = Lm && Lm->mb_type == WH_MB_PSKIP, t_sk = Tm && Tm->mb_type == WH_MB_PSKIP;
  const bool tl_sk = TLm && TLm->mb_type == WH_MB_PSKIP, tr_sk = TRm && TRm->mb_type == WH_MB_PSKIP;
  const bool try_skip = l_sk || t_sk || tl_sk || tr_sk;
  const bool keep_skip = l_sk && t_sk && tr_sk;
  bool b_skip = false;
  if ((ref_is_p && ref_mb_type == WH_MB_PSKIP) || try_skip) {
    // PredictSadSkip (md.cpp:872-910)
    int sad_pred_skip;
    {
      const int rb = S.refc[1], ra = S.refc[6];
      int rc = S.refc[5];
      const int sb = skc[1] ? sadsk[1] : 0, sa = skc[3] ? sadsk[3] : 0;
      int sc = skc[2] ? sadsk[2] : 0, skip_c = skc[2];
      if (rc == WH_REF_NOT_AVAIL) { rc = S.refc[0]; sc = skc[0] ? sadsk[0] : 0; skip_c = skc[0]; }
      if (rb == WH_REF_NOT_AVAIL && rc == WH_REF_NOT_AVAIL && ra != WH_REF_NOT_AVAIL) sad_pred_skip = sa;
      else {
        const int cnt = ((0 == ra) && skc[3]) | (((0 == rb) && skc[1]) << 1) | (((0 == rc) && skip_c) << 2);
        sad_pred_skip = cnt == 1 ? sa : cnt == 2 ? sb : cnt == 4 ? sc : wh_median3 (sa, sb, sc);
      }
    }
    wh_pred_skip_mv (S, &skx, &sky);
    const int nx = (mbx << 4) + (skx >> 2), ny = (mby << 4) + (sky >> 2);
    if (!(nx < -29 || nx > (P.mb_w << 4) + 12 || ny < -29 || ny > (P.mb_h << 4) + 12)) {
      wh_mc_luma_from_ref (P, J, mbx * 16, mby * 16, skx, sky, S.skip_y);
      wh_mc_chroma (P, J, mbx, mby, 0, 0, 8, 8, skx, sky, S.skip_c);
      const int sad_l = wh_sad_tile (S, 0, 0, 16, 16, S.skip_y, 16);
      int sad_c;
      WV_SUM (sad_c, lane, (lane < 32 ? (wh_abs (M.enc_c[lane * 4] - S.skip_c[lane * 4]) + wh_abs (M.enc_c[lane * 4 + 1] - S.skip_c[lane * 4 + 1]) +
                                          wh_abs (M.enc_c[lane * 4 + 2] - S.skip_c[lane * 4 + 2]) + wh_abs (M.enc_c[lane * 4 + 3] - S.skip_c[lane * 4 + 3])) : 0));
      const int sad_mb = sad_l + sad_c;
      bool ok = sad_mb == 0 || sad_mb < sad_pred_skip || (ref_is_p && ref_mb_type == WH_MB_PSKIP && sad_mb < J.ref_mbs[xy].skip_sad);
      if (!ok) {
        // residual would quantise to nothing?  (WelsDctMb + WelsTryPYskip + WelsTryPUVskip)
        WV_LANES_BEGIN (lane)
        for (int k = 0; k < 4; ++k) M.pred_y[lane * 4 + k] = S.skip_y[lane * 4 + k];
        if (lane < 32) for (int k = 0; k < 4; ++k) M.pred_c[lane * 4 + k] = S.skip_c[lane * 4 + k];
        WV_LANES_END
        wh_dct_luma16 (M);
        if (wh_try_py_skip (M, qp)) {
          wh_dct_chroma (M);
          if (wh_try_puv_skip (M, 0, qpc) && wh_try_puv_skip (M, 1, qpc)) ok = true;
        }
      }
      if (ok) {
        b_skip = true;
        cost_luma = md_using_sad ? sad_l : wh_satd_tile (S, 0, 0, 16, 16, S.skip_y, 16);
        // pSadCost[0] is only refreshed when bMdUsingSad; otherwise the SMB entry keeps the previous frame's value
        sad_cost0 = md_using_sad ? sad_l : J.ref_mbs[xy].sad_cost[0];
        cost_skip_mb = sad_mb;
        p16x = skx; p16y = sky;
      }
    }
  }
  if (b_skip && keep_skip) { mb_type = WH_MB_PSKIP; done = true; }
  WH_PROF_MARK (P, 1);   // P_Skip test

  WhMe me16;
  if (!done && !b_skip) {
    // PredictSad (md.cpp:826-870)
    int sad_pred;
    {
      const int rb = S.refc[1], ra = S.refc[6];
      int rc = S.refc[5], sc = sadc[2];
      if (rc == WH_REF_NOT_AVAIL) { rc = S.refc[0]; sc = sadc[0]; }
      if (rb == WH_REF_NOT_AVAIL && rc == WH_REF_NOT_AVAIL && ra != WH_REF_NOT_AVAIL) sad_pred = sadc[3];
      else {
        const int cnt = (0 == ra) | ((0 == rb) << 1) | ((0 == rc) << 2);
        sad_pred = cnt == 1 ? sadc[3] : cnt == 2 ? sadc[1] : cnt == 4 ? sc : wh_median3 (sadc[3], sadc[1], sc);
      }
      const int v = sad_pred << 6;
      sad_pred = ((v - (v >> 3) + (v >> 5)) + 32) >> 6;
    }
    // ---- P16x16 (WelsMdP16x16): candidates = base (0), left/top P16x16 mv, co-located right/below of the ref ----
    int16_t mvcl[5][2];
    int nm = 0;
    mvcl[nm][0] = 0; mvcl[nm][1] = 0; ++nm;
    if (Lm) { mvcl[nm][0] = Lm->p16mv[0]; mvcl[nm][1] = Lm->p16mv[1]; ++nm; }
    if (Tm) { mvcl[nm][0] = Tm->p16mv[0]; mvcl[nm][1] = Tm->p16mv[1]; ++nm; }
    if (ref_is_p) {
      if (mbx < P.mb_w - 1) { mvcl[nm][0] = J.ref_mbs[xy + 1].p16mv[0]; mvcl[nm][1] = J.ref_mbs[xy + 1].p16mv[1]; ++nm; }
      if (mby < P.mb_h - 1) { mvcl[nm][0] = J.ref_mbs[xy + w].p16mv[0]; mvcl[nm][1] = J.ref_mbs[xy + w].p16mv[1]; ++nm; }
    }
    me16.bx = 0; me16.by = 0; me16.bw = 16; me16.bh = 16; me16.sad_pred = sad_pred;
    wh_pred_mv (S, 0, 0, 4, 0, &me16.mvpx, &me16.mvpy);
    wh_motion_search (S, P, J, C, me16, mvcl, nm);
    p16x = me16.mvx; p16y = me16.mvy;
    cost_luma = me16.satd_cost;
    mb_type = WH_MB_P16x16;

    // remember sad_pred for the partitions below
    me16.sad_pred = sad_pred;
  }

  WH_PROF_MARK (P, 2);   // P16x16 motion search
  // ---- secondary modes (WelsMdInterSecondaryModesEnc) ----
  bool intra = false;
  WhIntraResult ir;
  if (!done) {
    // WelsMdFirstIntraMode: I16x16 cost vs the inter/skip cost so far
    if (wh_intra_md_enc_p (M, P, J, mbx, mby, avail, qp, qpc, cost_luma, &ir)) { intra = true; done = true; }
  }
  if (!done && b_skip) { mb_type = WH_MB_PSKIP; done = true; }
  WH_PROF_MARK (P, 3);   // I16x16 test (+ intra encode when intra wins)

  int sub_type[4] = {0, 0, 0, 0};
  if (!done) {
    // ---- fine partitions ----
    WhMe me8[4], me168[2], me816[2];
    const int zero_mvc[1][2] = {{0, 0}};
    const int16_t zmv[1][2] = {{0, 0}};
    (void)zero_mvc;
    auto do_p8x8 = [&] () {
      int c = 0;
      for (int i = 0; i < 4; ++i) {
        WhMe& m = me8[i];
        m.bx = (i & 1) * 8; m.by = (i >> 1) * 8; m.bw = 8; m.bh = 8; m.sad_pred = me16.sad_pred >> 2;
        wh_pred_mv (S, (i & 1) * 2, (i >> 1) * 2, 2, 0, &m.mvpx, &m.mvpy);
        wh_motion_search (S, P, J, C, m, zmv, 1);
        wh_cache_set (S, (i & 1) * 2, (i >> 1) * 2, 2, 2, 0, m.mvx, m.mvy);
        c += m.satd_cost;
      }
      return c;
    };
    auto do_p16x8 = [&] () {
      int c = 0;
      for (int i = 0; i < 2; ++i) {
        WhMe& m = me168[i];
        m.bx = 0; m.by = i * 8; m.bw = 16; m.bh = 8; m.sad_pred = me16.sad_pred >> 1;
        wh_pred_16x8 (S, i, 0, &m.mvpx, &m.mvpy);
        wh_motion_search (S, P, J, C, m, zmv, 1);
        wh_cache_set (S, 0, i * 2, 4, 2, 0, m.mvx, m.mvy);
        c += m.satd_cost;
      }
      return c;
    };
    auto do_p8x16 = [&] () {
      int c = 0;
      for (int i = 0; i < 2; ++i) {
        WhMe& m = me816[i];
        m.bx = i * 8; m.by = 0; m.bw = 8; m.bh = 16; m.sad_pred = me16.sad_pred >> 1;
        wh_pred_8x16 (S, i, 0, &m.mvpx, &m.mvpy);
        wh_motion_search (S, P, J, C, m, zmv, 1);
        wh_cache_set (S, i * 2, 0, 2, 4, 0, m.mvx, m.mvy);
        c += m.satd_cost;
      }
      return c;
    };
    int best_cost = cost_luma;
    if (!use_satd) {
      // WelsMdInterFinePartitionVaa: partition set chosen from the sign pattern of the four 8x8 SADs
      // between this source MB and the previous source frame (VAACalcSad_c + MdInterAnalysisVaaInfo_c)
      int s8[4];
      for (int k = 0; k < 4; ++k) {
        const int ex = (k & 1) * 8, ey = (k >> 1) * 8;
        const uint8_t* pv = J.prev_src_y + (size_t) (mby * 16 + ey) * P.src_stride_y + mbx * 16 + ex;
        s8[k] = wh_sad_tile (S, ex, ey, 8, 8, pv, P.src_stride_y);
      }
      int sign = 15;
      {
        const int avg = (s8[0] + s8[1] + s8[2] + s8[3]) >> 2;
        int var = 0;
        for (int k = 0; k < 4; ++k) { const int d = (s8[k] >> 6) - (avg >> 6); var += d * d; }
        if (var >= 20) sign = ((s8[0] > avg) << 3) | ((s8[1] > avg) << 2) | ((s8[2] > avg) << 1) | (s8[3] > avg);
      }
      if (sign != 15) {
        if (sign == 3 || sign == 12) { const int c = do_p16x8(); if (c < best_cost) { best_cost = c; mb_type = WH_MB_P16x8; } }
        else if (sign == 5 || sign == 10) { const int c = do_p8x16(); if (c < best_cost) { best_cost = c; mb_type = WH_MB_P8x16; } }
        else if (sign == 6 || sign == 9) { const int c = do_p8x8(); if (c < best_cost) { best_cost = c; mb_type = WH_MB_P8x8; } }
        else {
          const int c8 = do_p8x8();
          if (c8 < best_cost) {
            best_cost = c8; mb_type = WH_MB_P8x8;
            const int c1 = do_p16x8(); if (c1 <= best_cost) { best_cost = c1; mb_type = WH_MB_P16x8; }
            const int c2 = do_p8x16(); if (c2 <= best_cost) { best_cost = c2; mb_type = WH_MB_P8x16; }
          }
        }
        cost_luma = best_cost;
      }
    } else {
      // WelsMdInterFinePartition
      int c = do_p8x8();
      if (c < best_cost) {
        mb_type = WH_MB_P8x8;
        const int c1 = do_p16x8(); if (c1 <= c) { c = c1; mb_type = WH_MB_P16x8; }
        const int c2 = do_p8x16(); if (c2 <= c) { c = c2; mb_type = WH_MB_P8x16; }
      }
    }

    WH_PROF_MARK (P, 4);   // fine partitions
    // ---- refinement (WelsMdInterMbRefinement) ----
    const int satd_in_md = use_satd;     // bSatdInMdFlag: pfMeCost == pfMdCost == SATD
    int best_sad = 0, best_satd = 0;
    auto put_mv = [&] (int bx4, int by4, int w4, int h4, int mvx, int mvy, int px, int py) {
      WV_LANES_BEGIN (lane)
      if (lane < w4 * h4) {
        const int r = (by4 + lane / w4) * 4 + bx4 + lane % w4;
        S.mv_out[r][0] = (int16_t)mvx; S.mv_out[r][1] = (int16_t)mvy; S.mvp_out[r][0] = (int16_t)px; S.mvp_out[r][1] = (int16_t)py;
      }
      WV_LANES_END
    };
    if (mb_type == WH_MB_P16x16) {
      wh_refine_frac (S, P, J, C, me16, satd_in_md);
      wh_cache_set (S, 0, 0, 4, 4, 0, me16.mvx, me16.mvy);
      put_mv (0, 0, 4, 4, me16.mvx, me16.mvy, me16.mvpx, me16.mvpy);
      best_sad = me16.sad_cost; best_satd = me16.satd_cost;
      wh_mc_chroma (P, J, mbx, mby, 0, 0, 8, 8, me16.mvx, me16.mvy, M.pred_c);
      // iCostSkipMb of a 16x16 MB = SAD of its final prediction (luma + chroma)
      const int sl = wh_sad_tile (S, 0, 0, 16, 16, M.pred_y, 16);
      int sc;
      WV_SUM (sc, lane, (lane < 32 ? (wh_abs (M.enc_c[lane * 4] - M.pred_c[lane * 4]) + wh_abs (M.enc_c[lane * 4 + 1] - M.pred_c[lane * 4 + 1]) +
                                       wh_abs (M.enc_c[lane * 4 + 2] - M.pred_c[lane * 4 + 2]) + wh_abs (M.enc_c[lane * 4 + 3] - M.pred_c[lane * 4 + 3])) : 0));
      cost_skip_mb = sl + sc;
    } else if (mb_type == WH_MB_P16x8) {
      for (int i = 0; i < 2; ++i) {
        WhMe& m = me168[i];
        wh_pred_16x8 (S, i, 0, &m.mvpx, &m.mvpy);
        wh_refine_frac (S, P, J, C, m, satd_in_md);
        wh_cache_set (S, 0, i * 2, 4, 2, 0, m.mvx, m.mvy);
        put_mv (0, i * 2, 4, 2, m.mvx, m.mvy, m.mvpx, m.mvpy);
        best_sad += m.sad_cost; best_satd += m.satd_cost;
        wh_mc_chroma (P, J, mbx, mby, 0, i * 4, 8, 4, m.mvx, m.mvy, M.pred_c);
      }
    } else if (mb_type == WH_MB_P8x16) {
      for (int i = 0; i < 2; ++i) {
        WhMe& m = me816[i];
        wh_pred_8x16 (S, i, 0, &m.mvpx, &m.mvpy);
        wh_refine_frac (S, P, J, C, m, satd_in_md);
        wh_cache_set (S, i * 2, 0, 2, 4, 0, m.mvx, m.mvy);
        put_mv (i * 2, 0, 2, 4, m.mvx, m.mvy, m.mvpx, m.mvpy);
        best_sad += m.sad_cost; best_satd += m.satd_cost;
        wh_mc_chroma (P, J, mbx, mby, i * 4, 0, 4, 8, m.mvx, m.mvy, M.pred_c);
      }
    } else {   // P8x8, all sub types 8x8
      WV_LANES_BEGIN (lane)
      if (lane == 0) { S.refc[9] = WH_REF_NOT_AVAIL; S.refc[21] = WH_REF_NOT_AVAIL; }
      WV_LANES_END
      for (int i = 0; i < 4; ++i) {
        WhMe& m = me8[i];
        wh_pred_mv (S, (i & 1) * 2, (i >> 1) * 2, 2, 0, &m.mvpx, &m.mvpy);
        wh_refine_frac (S, P, J, C, m, satd_in_md);
        wh_cache_set (S, (i & 1) * 2, (i >> 1) * 2, 2, 2, 0, m.mvx, m.mvy);
        put_mv ((i & 1) * 2, (i >> 1) * 2, 2, 2, m.mvx, m.mvy, m.mvpx, m.mvpy);
        best_sad += m.sad_cost; best_satd += m.satd_cost;
        wh_mc_chroma (P, J, mbx, mby, (i & 1) * 4, (i >> 1) * 4, 4, 4, m.mvx, m.mvy, M.pred_c);
      }
    }
    sad_cost0 = best_sad;
    cost_luma = md_using_sad ? best_sad : best_satd;

    WH_PROF_MARK (P, 5);   // fractional refinement + chroma MC
    // ---- encode (WelsMdInterEncode) ----
    wh_dct_luma16 (M);
    cbp = wh_enc_inter_y (M, qp);
    const int cbp_c = wh_encrec_chroma (M, qpc, 0);
    cbp |= cbp_c << 4;
    wh_idct_luma16 (M);
    wh_idct_chroma (M);
    // ---- WelsMdInterDoubleCheckPskip ----
    if (mb_type == WH_MB_P16x16 && cbp == 0) {
      // PredSkipMv against the neighbour cache (row 0 / col 0 are untouched by the partition updates)
      int sx, sy;
      wh_pred_skip_mv (S, &sx, &sy);
      if (sx == me16.mvx && sy == me16.mvy) { mb_type = WH_MB_PSKIP; skx = sx; sky = sy; }
    }
  }

  WH_PROF_MARK (P, 6);   // residual coding
  // ---- store ----
  const bool is_skip = mb_type == WH_MB_PSKIP;
  if (intra) {
    WV_LANES_BEGIN (lane)
    if (lane < 16) { J.mbs[xy].mv[lane][0] = 0; J.mbs[xy].mv[lane][1] = 0; J.records[xy].mvd[lane][0] = 0; J.records[xy].mvd[lane][1] = 0; }
    if (lane < 4) { J.mbs[xy].ref_idx[lane] = -1; J.records[xy].ref_idx[lane] = -1; J.records[xy].sub_type[lane] = 0; }
    if (lane == 0) { J.mbs[xy].sad_cost[0] = 0; J.mbs[xy].p16mv[0] = (int16_t)p16x; J.mbs[xy].p16mv[1] = (int16_t)p16y; J.mbs[xy].skip_sad = 0; }
    WV_LANES_END
    wh_store_mb (M, P, J, mbx, mby, ir.mb_type, ir.cbp, qp, qpc, ir.i16_mode_std, ir.chroma_mode_std, ir.cost_luma, slice_idc);
    return;
  }
  if (is_skip && b_skip) {
    // decided P_Skip: reconstruction = skip prediction, no residual (WelsRecPskip)
    WV_LANES_BEGIN (lane)
    {
      const int row = lane >> 2, seg = lane & 3;
      for (int k = 0; k < 4; ++k) WH_RY (M, seg * 4 + k, row) = S.skip_y[row * 16 + seg * 4 + k];
    }
    if (lane < 32) {
      const int pl = lane >> 4, row = (lane >> 1) & 7, half = lane & 1;
      for (int k = 0; k < 4; ++k) WH_RC (M, pl, half * 4 + k, row) = S.skip_c[pl * 64 + row * 8 + half * 4 + k];
    }
    if (lane < 24) M.nzc[lane] = 0;
    WV_LANES_END
    cbp = 0;
  }
  WV_LANES_BEGIN (lane)
  if (lane < 16) {
    const int mvx = is_skip ? skx : S.mv_out[lane][0], mvy = is_skip ? sky : S.mv_out[lane][1];
    J.mbs[xy].mv[lane][0] = (int16_t)mvx; J.mbs[xy].mv[lane][1] = (int16_t)mvy;
    J.records[xy].mvd[lane][0] = is_skip ? (int16_t)0 : (int16_t) (mvx - S.mvp_out[lane][0]);
    J.records[xy].mvd[lane][1] = is_skip ? (int16_t)0 : (int16_t) (mvy - S.mvp_out[lane][1]);
    if (!(cbp & 15) || is_skip) { for (int k = 0; k < 16; ++k) M.lv_luma[lane * 16 + k] = 0; }
  }
  if (lane < 4) { J.mbs[xy].ref_idx[lane] = 0; J.records[xy].ref_idx[lane] = 0; J.records[xy].sub_type[lane] = (uint8_t)sub_type[lane]; }
  if (lane == 0) {
    J.mbs[xy].sad_cost[0] = sad_cost0; J.mbs[xy].p16mv[0] = (int16_t)p16x; J.mbs[xy].p16mv[1] = (int16_t)p16y;
    J.mbs[xy].skip_sad = is_skip ? cost_skip_mb : 0;
  }
  WV_LANES_END
  wh_store_mb (M, P, J, mbx, mby, mb_type, cbp, qp, qpc, 0, 0, cost_luma, slice_idc);
  WH_PROF_MARK (P, 7);   // store
}
