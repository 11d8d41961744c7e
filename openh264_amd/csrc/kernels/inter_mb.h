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
// Screen content (iUsageType == SCREEN_CONTENT_REAL_TIME; template parameter SCC of the macroblock body):
//   svc_mode_decision.cpp:293-541    JudgeStaticSkip / JudgeScrollSkip / SvcMdSCDMbEnc / WelsMdInterJudgeSCDPskip
//   svc_mode_decision.cpp:553-667    TryModeMerge / WelsMdInterFinePartitionVaaOnScreen
//   svc_motion_estimate.cpp:170-218  WelsMotionEstimateSearchStatic / ...Scrolled
//   svc_motion_estimate.cpp:385-412  CheckDirectionalMv
//   svc_motion_estimate.cpp:568-646  LineFullSearch_c / WelsMotionCrossSearch
//   svc_motion_estimate.cpp:880-1097 SetFeatureSearchIn / FeatureSearchOne / WelsDiamondCrossSearch / WelsDiamondCrossFeatureSearch
#pragma once
#include "frame_kernels.h"
#include "../common/gom_rc.h"

// Performance shape (MI355X): the wave pays HBM latency twice per macroblock -- one batch of loads for the source
// tile, the neighbour pixels and the neighbour / co-located MB states, and one batch for the reference search
// windows (luma 64x64, chroma 2 x 32x32) centred on the 16x16 motion vector predictor.  Everything after that
// (candidate SADs, diamond search, fractional refinement, motion compensation, residual coding) runs from LDS and
// registers; a window is re-fetched only when a search start or a prediction block falls outside it.

// test-build statistics of the search windows (WELSHIP_WIN_STATS=1 prints them at exit): 0 P macroblocks, 1 reloads (wh_win_ensure misses),
// 2 speculative windows adopted, 3 first loads after a refused / missing speculation, 4 reloads in the middle of a diamond walk,
// 5 chroma predictions made inside the chroma window, 6 read from the picture
#if defined(WH_EMU)
#include <stdio.h>
static long g_wh_win_stat[40];
// 8.. : what the macroblocks became and what their searches cost (WELSHIP_WIN_STATS=1): 8 decided P_Skip, 9 background / static, 10 intra, 11 P16x16, 12 P16x8,
// 13 P8x16, 14 P8x8, 15 renamed P_Skip (double check); 16 / 17 / 18 searches of 16x16 / 16x8+8x16 / 8x8 blocks, 19 / 20 / 21 their diamond steps,
// 22 / 23 / 24 refinements of 16x16 / 16x8+8x16 / 8x8 blocks, 25 P_Skip tests, 26 of which went through the transform
static void wh_win_stat_dump() {
  if (!getenv ("WELSHIP_WIN_STATS")) return;
  fprintf (stderr, "welship mb stat: skip %ld bg/static %ld intra %ld p16x16 %ld p16x8 %ld p8x16 %ld p8x8 %ld renamed-skip %ld | searches 16x16 %ld 16x8/8x16 %ld 8x8 %ld, diamond steps %ld %ld %ld | refinements %ld %ld %ld | skip tests %ld with transform %ld\n",
           g_wh_win_stat[8], g_wh_win_stat[9], g_wh_win_stat[10], g_wh_win_stat[11], g_wh_win_stat[12], g_wh_win_stat[13], g_wh_win_stat[14], g_wh_win_stat[15], g_wh_win_stat[16], g_wh_win_stat[17], g_wh_win_stat[18],
           g_wh_win_stat[19], g_wh_win_stat[20], g_wh_win_stat[21], g_wh_win_stat[22], g_wh_win_stat[23], g_wh_win_stat[24], g_wh_win_stat[25], g_wh_win_stat[26]);
  fprintf (stderr, "welship window stat: P macroblocks %ld, reloads %ld, adopted %ld, first loads %ld, mid-walk reloads %ld, chroma predictions inside the window %ld / from the picture %ld\n", g_wh_win_stat[0], g_wh_win_stat[1], g_wh_win_stat[2], g_wh_win_stat[3], g_wh_win_stat[4], g_wh_win_stat[5], g_wh_win_stat[6]);
}
struct WhWinStatInit { WhWinStatInit() { atexit (wh_win_stat_dump); } };
static WhWinStatInit g_wh_win_stat_init;
#define WH_STAT_WIN(i) (++g_wh_win_stat[i])
#define WH_STAT_BLK(base, bw, bh) (++g_wh_win_stat[(base) + ((bw) == 16 && (bh) == 16 ? 0 : (bw) == 8 && (bh) == 8 ? 2 : 1)])
#else
#define WH_STAT_WIN(i) ((void)0)
#define WH_STAT_BLK(base, bw, bh) ((void)0)
#endif

#define WH_REF_NOT_AVAIL (-2)
#define WH_REF_NOT_IN_LIST (-1)
// The windows are SMALL and follow the search (round 5; rounds 1-4 held everything a 16-step diamond could ever reach, 56 rows of
// luma and 32 of chroma: 6.4 KB of a wave's 12.7 KB of LDS, which kept a CU at 12 waves): a search start only demands WH_WIN_START
// samples of room around its block (wh_win_need), the diamond counts its steps against the room it started with and has the window
// follow it when that is used up (wh_motion_search), and every other reader states what it reads (wh_win_ensure).  What a window holds
// never changes a result -- only whether a reload is needed.
#define WH_WIN_STRIDE 80          // five tile columns (common/wh_types.h WH_TILE_*): 16 + 2 * 19 samples wherever the block sits in its tile column
#ifndef WH_WIN_ROWS
#define WH_WIN_ROWS 38            // 16 + 2 * 11 rows = 190 16-byte pieces: three loads per lane.  (Rows of 80 bytes also spread the lanes of a block read over all LDS banks; 64 did not.)
#endif
#define WH_WIN_MARGIN_X 19        // what a (re)load leaves around the block it is centred on, horizontally ...
#ifndef WH_WIN_MARGIN_Y
#define WH_WIN_MARGIN_Y 11        // ... and vertically (a 16-row block: all 38 rows)
#endif
#ifndef WH_WIN_START
#define WH_WIN_START 5            // room (samples in every direction) a search start asks for: what the fractional refinement reads around a block that does not move (measured on the reference 1080p clip with the test build: profiles/r05_window_reload_statistics.txt)
#endif
#define WH_CWIN_STRIDE 64         // chroma window: 32 samples x 24 rows of BOTH planes, stored as the tiles are -- per row four 16-byte pieces,
#ifndef WH_CWIN_ROWS
#define WH_CWIN_ROWS 24           //   each 8 Cb samples then the 8 Cr samples at the same position (a block outside it is read from the picture)
#endif
#define WH_CWIN_COLS 32

// partition slots: one motion search result each (WhMeTab)
#define WH_SLOT_16x16 0
#define WH_SLOT_8x8 1             // 1..4
#define WH_SLOT_16x8 5            // 5,6
#define WH_SLOT_8x16 7            // 7,8

// The reference search windows of a wave: a separate LDS object (its own __shared__ variable on the GPU) because they are
// filled by LDS-DMA -- the compiler then knows that reads of the MB tile cannot alias a window load still in flight.
typedef struct alignas (16) WhWinLds {
  alignas (16) uint8_t win[WH_WIN_ROWS * WH_WIN_STRIDE + 16];            // luma, see wh_win_load_luma
  alignas (16) uint8_t cwin[WH_CWIN_ROWS * WH_CWIN_STRIDE + 16];         // chroma, Cb and Cr interleaved in 8-sample groups (wh_cwin_off)
} WhWinLds;

typedef struct alignas (16) WhInterLds {
  WhMbLds m;
  // (the P_Skip prediction lives in m.pred_y / m.pred_c: whenever a macroblock makes another prediction it is not a decided skip any more)
  uint32_t nb[5 * 36];                                      // WhMbState copies: top-left, top, top-right, left, co-located (reference picture)
  int16_t co_mv[2][2];                                      // sP16x16Mv of the reference picture's MBs to the right / below (MUST follow nb: wh_inter_cold_fetch)
  uint32_t sad_cost0_in;                                    // the layer's pSadCost[0] of this macroblock (WhPicJob::sad_cost0), when the host supplies it
  int16_t mvp_out[16][2];                                   // predictor used for the mvd of each 4x4 (raster)
  int16_t mv_out[16][2];
} WhInterLds;

// The "cold" inputs of a wave's NEXT macroblock (wh_inter_cold_fetch) are copied by LDS-DMA straight to where the body reads them -- the source
// samples into m.enc_y / m.enc_c, the reference picture's macroblock state into nb[144..] / co_mv -- while the wave waits for that macroblock's
// neighbours: the macroblock in hand is complete by then (rounds 1-4 staged them in 800 bytes of their own and moved them at the top of the
// body).  Only the previous source picture's luma block keeps a staging area: it is a separate LDS object (a different __shared__ variable on
// the GPU), which the four words of the host's own 8x8 SADs share when the pre-processing supplies them.
typedef struct alignas (16) WhInterStage {
  uint32_t cold_pv[64];
} WhInterStage;
typedef struct WhWin { int x0, y0, cx0, cy0; WhWinLds* b; } WhWin;     // picture coordinates of element (0,0) of win / cwin + where they live

// ---- mvd cost: lambda * bits(se(mvd))  (md.cpp:797-824, svc_enc_golomb.h BsSizeSE) --------------
// (branch-free: the count is evaluated on the scalar unit eight times per diamond step, and as a shift loop it was five instructions
//  and a taken branch per bit of the difference)
WH_FN int wh_se_bits (int v) {
  const unsigned k = v > 0 ? 2u * (unsigned)v : 1u - 2u * (unsigned)v;      // codeNum + 1 of se(v): 2v for v > 0, 1 - 2v otherwise (v = 0: 1)
  return 2 * (31 - __builtin_clz (k)) + 1;
}
WH_FN int wh_mvd_cost (int lambda, int dx, int dy) { return (int) (uint16_t) (lambda * wh_se_bits (dx)) + (int) (uint16_t) (lambda * wh_se_bits (dy)); }

// ---- H.264 luma sample interpolation (8.4.2.2.1; mc.cpp:100-347), four horizontally adjacent samples per lane ----
// `w` is the LDS window, `o` the byte offset of the integer sample G of the first of the four.
// (Tried in round 6: the shift-and-add form (a + f) + 5 m, m = 4 (c + d) - (b + e), which spares the second stage of the centre half sample its
//  quarter-rate 32-bit multiplies -- the 16-wave kernel then needed 150 registers and spilled 22 of them to scratch: MD launch 7.2 -> 8.3 ms.
//  tests/test_abi.py now checks that no mode-decision kernel that ships spills.)
WH_FN int wh_tap6 (int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
#define WH_BYTE(v, k) ((int) (((v) >> (8 * (k))) & 255u))
WH_FN uint32_t wh_pack4 (int a, int b, int c, int d) { return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24); }
// unclipped horizontal 6-tap sums of the four samples at o (needs bytes o-2 .. o+6)
WH_FN void wh_htaps4 (const uint8_t* w, int o, int* t0, int* t1, int* t2, int* t3) {
  const uint32_t a = wh_ld4u (w, o - 2), b = wh_ld4u (w, o + 2), c = wh_ld4u (w, o + 6);
  const int q0 = WH_BYTE (a, 0), q1 = WH_BYTE (a, 1), q2 = WH_BYTE (a, 2), q3 = WH_BYTE (a, 3);
  const int q4 = WH_BYTE (b, 0), q5 = WH_BYTE (b, 1), q6 = WH_BYTE (b, 2), q7 = WH_BYTE (b, 3), q8 = WH_BYTE (c, 0);
  *t0 = wh_tap6 (q0, q1, q2, q3, q4, q5); *t1 = wh_tap6 (q1, q2, q3, q4, q5, q6);
  *t2 = wh_tap6 (q2, q3, q4, q5, q6, q7); *t3 = wh_tap6 (q3, q4, q5, q6, q7, q8);
}
WH_FN uint32_t wh_mc_b4 (const uint8_t* w, int o) {          // half-sample b (between G and its right neighbour)
  int t0, t1, t2, t3;
  wh_htaps4 (w, o, &t0, &t1, &t2, &t3);
  return wh_pack4 (wh_clip255 ((t0 + 16) >> 5), wh_clip255 ((t1 + 16) >> 5), wh_clip255 ((t2 + 16) >> 5), wh_clip255 ((t3 + 16) >> 5));
}
WH_FN uint32_t wh_mc_h4 (const uint8_t* w, int o) {          // half-sample h (between G and the sample below)
  const uint32_t r0 = wh_ld4u (w, o - 2 * WH_WIN_STRIDE), r1 = wh_ld4u (w, o - WH_WIN_STRIDE), r2 = wh_ld4u (w, o);
  const uint32_t r3 = wh_ld4u (w, o + WH_WIN_STRIDE), r4 = wh_ld4u (w, o + 2 * WH_WIN_STRIDE), r5 = wh_ld4u (w, o + 3 * WH_WIN_STRIDE);
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] = wh_clip255 ((wh_tap6 (WH_BYTE (r0, k), WH_BYTE (r1, k), WH_BYTE (r2, k), WH_BYTE (r3, k), WH_BYTE (r4, k), WH_BYTE (r5, k)) + 16) >> 5);
  return wh_pack4 (v[0], v[1], v[2], v[3]);
}
WH_FN uint32_t wh_mc_j4 (const uint8_t* w, int o) {          // centre half-sample j
  int t[6][4];
#pragma unroll
  for (int k = 0; k < 6; ++k) wh_htaps4 (w, o + (k - 2) * WH_WIN_STRIDE, &t[k][0], &t[k][1], &t[k][2], &t[k][3]);
  int v[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) v[x] = wh_clip255 ((wh_tap6 (t[0][x], t[1][x], t[2][x], t[3][x], t[4][x], t[5][x]) + 512) >> 10);
  return wh_pack4 (v[0], v[1], v[2], v[3]);
}
WH_FN uint32_t wh_mc4 (const uint8_t* w, int o, int fx, int fy) {
  switch (fy * 4 + fx) {
  case 0: return wh_ld4u (w, o);
  case 1: return wh_avg4 (wh_ld4u (w, o), wh_mc_b4 (w, o));
  case 2: return wh_mc_b4 (w, o);
  case 3: return wh_avg4 (wh_ld4u (w, o + 1), wh_mc_b4 (w, o));
  case 4: return wh_avg4 (wh_ld4u (w, o), wh_mc_h4 (w, o));
  case 5: return wh_avg4 (wh_mc_b4 (w, o), wh_mc_h4 (w, o));
  case 6: return wh_avg4 (wh_mc_b4 (w, o), wh_mc_j4 (w, o));
  case 7: return wh_avg4 (wh_mc_b4 (w, o), wh_mc_h4 (w, o + 1));
  case 8: return wh_mc_h4 (w, o);
  case 9: return wh_avg4 (wh_mc_h4 (w, o), wh_mc_j4 (w, o));
  case 10: return wh_mc_j4 (w, o);
  case 11: return wh_avg4 (wh_mc_j4 (w, o), wh_mc_h4 (w, o + 1));
  case 12: return wh_avg4 (wh_ld4u (w, o + WH_WIN_STRIDE), wh_mc_h4 (w, o));
  case 13: return wh_avg4 (wh_mc_h4 (w, o), wh_mc_b4 (w, o + WH_WIN_STRIDE));
  case 14: return wh_avg4 (wh_mc_j4 (w, o), wh_mc_b4 (w, o + WH_WIN_STRIDE));
  default: return wh_avg4 (wh_mc_h4 (w, o + 1), wh_mc_b4 (w, o + WH_WIN_STRIDE));
  }
}
// chroma (mc.cpp:349-378): bilinear with eighth-sample weights
WH_FN int wh_mc_chroma_w (int a, int b, int c, int d, int dx, int dy) {
  return ((8 - dx) * (8 - dy) * a + dx * (8 - dy) * b + (8 - dx) * dy * c + dx * dy * d + 32) >> 6;
}

// ---- reference windows ----------------------------------------------------------------------------
// A window is a WH_WIN_STRIDE x WH_WIN_ROWS block of the border-expanded reference luma (plus 32 x 32 of both chroma planes),
// fetched from the picture's TILED twin (common/wh_types.h): the origin is a tile column (x0 a multiple of 16, chroma 8) and a
// multiple of four rows, so each 16-byte piece a lane moves is one row of one 128-byte tile and a window costs the fabric the
// 35-40 lines it is made of instead of one or two lines per row.  Origins are clamped so that the whole window lies inside the
// expanded picture (32 luma / 16 chroma samples each side): loads need no per-lane clamping.
#define WH_WIN_PIECES (WH_WIN_ROWS * (WH_WIN_STRIDE / 16))          // 16-byte pieces of the luma window, row-major: piece p = row p / 5, tile column p % 5
#define WH_WIN_LOADS ((WH_WIN_PIECES + 63) / 64)                     // load instructions per lane
#define WH_CWIN_PIECES (WH_CWIN_ROWS * (WH_CWIN_STRIDE / 16))
#define WH_CWIN_LOADS ((WH_CWIN_PIECES + 63) / 64)
// a window origin in [lo, hi] that is a multiple of `align` when the range holds one (one tile row fewer to fetch), else the middle
WH_FN int wh_win_pick (int lo, int hi, int align) { const int a = hi & ~ (align - 1); return a >= lo ? a : (lo + hi) >> 1; }
WH_FN void wh_win_place (const WhSeqParams& P, WhWin& W, int cx, int cy) {      // leaves W.b alone      // (cx,cy): luma position the 16x16 block is centred on
  // luma: the block +- WH_WIN_MARGIN_X / _Y must be inside; columns: 24..39 samples either side of the block
  W.x0 = wh_clip3 ((cx - 24) & ~15, -32, P.mb_w * 16 + 32 - WH_WIN_STRIDE);
  W.y0 = wh_clip3 (wh_win_pick (cy + 16 + WH_WIN_MARGIN_Y - WH_WIN_ROWS, cy - WH_WIN_MARGIN_Y, 8), -32, P.mb_h * 16 + 32 - WH_WIN_ROWS);
  // chroma: 8..15 samples either side of the 8x8 block, 6+ rows above and below (whatever lies outside is read from the picture)
  const int ccx = cx >> 1, ccy = cy >> 1;
  W.cx0 = wh_clip3 ((ccx - 8) & ~7, -16, P.mb_w * 8 + 16 - WH_CWIN_COLS);
  W.cy0 = wh_clip3 (wh_win_pick (ccy - (WH_CWIN_ROWS - 9) / 2 - 2, ccy - (WH_CWIN_ROWS - 9) / 2 + 1, 8), -16, P.mb_h * 8 + 16 - WH_CWIN_ROWS);
}
// piece p of the luma window / chroma window in the tiled reference picture
// (WH_TILE_Y_OFF / WH_TILE_C_OFF of common/wh_types.h in 32-bit arithmetic with 24-bit multiplies: a tiled picture is far below 4 GB, a tile row and
//  the tiles per row far below 2^24 -- the macros' size_t products are 64-bit multiplies, four quarter-rate instructions per piece)
WH_FN uint32_t wh_tile_y_off32 (int stride_y, int x, int y) {
  return ((wh_mul_u24 ((uint32_t) ((y + 32) >> 3), (uint32_t) (stride_y >> 4)) + (uint32_t) ((x + 32) >> 4)) << 7) + (uint32_t) (((y + 32) & 7) << 4);
}
WH_FN uint32_t wh_tile_c_off32 (int stride_c, int x, int y) {
  return ((wh_mul_u24 ((uint32_t) ((y + 16) >> 3), (uint32_t) (stride_c >> 3)) + (uint32_t) ((x + 16) >> 3)) << 7) + (uint32_t) (((y + 16) & 7) << 4);
}
WH_FN const WH_G uint8_t* wh_win_src_luma (int p, const WhSeqParams& P, const WhPicJob& J, const WhWin& W) {
  const int r = (p * 205) >> 10, c = p - 5 * r;            // p / 5, p % 5 for p < 1024
  return (const WH_G uint8_t*)J.ref_tiles[0] + wh_tile_y_off32 (P.rec_stride_y, W.x0 + 16 * c, W.y0 + r);
}
WH_FN const WH_G uint8_t* wh_win_src_chroma (int p, const WhSeqParams& P, const WhPicJob& J, const WhWin& W) {
  return (const WH_G uint8_t*)J.ref_tiles[1] + wh_tile_c_off32 (P.rec_stride_c, W.cx0 + 8 * (p & 3), W.cy0 + (p >> 2));
}
// byte offset of chroma sample (x, y) (window coordinates) of plane pl inside cwin
WH_FN int wh_cwin_off (int pl, int x, int y) { return y * WH_CWIN_STRIDE + ((x >> 3) << 4) + (pl << 3) + (x & 7); }
// Window loads are LDS-DMA (16 bytes per lane, no register holds the data).  wh_win_issue_* only starts them; the data
// may be used after WV_ASYNC_WAIT().
WH_FN void wh_win_issue_luma (const WhSeqParams& P, const WhPicJob& J, WhWin& W) {
  WV_LANES_BEGIN (lane)
#pragma unroll
  for (int k = 0; k < WH_WIN_LOADS; ++k) if (64 * k + lane < WH_WIN_PIECES) wh_ld_async16 (wh_win_src_luma (64 * k + lane, P, J, W), &W.b->win[1024 * k], lane);
  WV_LANES_END
}
WH_FN void wh_win_issue_chroma (const WhSeqParams& P, const WhPicJob& J, WhWin& W) {
  WV_LANES_BEGIN (lane)
#pragma unroll
  for (int k = 0; k < WH_CWIN_LOADS; ++k) if (64 * k + lane < WH_CWIN_PIECES) wh_ld_async16 (wh_win_src_chroma (64 * k + lane, P, J, W), &W.b->cwin[1024 * k], lane);
  WV_LANES_END
}
WH_FN void wh_win_load_luma (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W) {
  (void)S;
  wh_win_issue_luma (P, J, W);
  WV_ASYNC_WAIT();
}
// first load of a macroblock: luma + both chroma windows in one batch (issue only)
WH_FN void wh_win_issue_all (const WhSeqParams& P, const WhPicJob& J, WhWin& W, int cx, int cy) {
  wh_win_place (P, W, cx, cy);
  WV_LANES_BEGIN (lane)
  {
#pragma unroll
    for (int k = 0; k < WH_WIN_LOADS; ++k) if (64 * k + lane < WH_WIN_PIECES) wh_ld_async16 (wh_win_src_luma (64 * k + lane, P, J, W), &W.b->win[1024 * k], lane);
#pragma unroll
    for (int k = 0; k < WH_CWIN_LOADS; ++k) if (64 * k + lane < WH_CWIN_PIECES) wh_ld_async16 (wh_win_src_chroma (64 * k + lane, P, J, W), &W.b->cwin[1024 * k], lane);
  }
  WV_LANES_END
}
WH_FN bool wh_win_covers (const WhWin& W, int x0, int y0, int x1, int y1) {
  return x0 >= W.x0 && y0 >= W.y0 && x1 <= W.x0 + WH_WIN_STRIDE && y1 <= W.y0 + WH_WIN_ROWS;
}
// make sure luma [x0,x1) x [y0,y1) is inside the window (extent <= 65 x WH_WIN_ROWS)
WH_FN void wh_win_ensure (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W, int x0, int y0, int x1, int y1) {
  if (wh_win_covers (W, x0, y0, x1, y1)) return;
  // (x1 - 80 <= x0 - 16: a tile column in between always exists)
  W.x0 = wh_clip3 (wh_win_pick (x1 - WH_WIN_STRIDE, x0, 16), -32, P.mb_w * 16 + 32 - WH_WIN_STRIDE);
  W.y0 = wh_clip3 (wh_win_pick (y1 - WH_WIN_ROWS, y0, 8), -32, P.mb_h * 16 + 32 - WH_WIN_ROWS);
  WH_STAT_WIN (1);
  wh_win_load_luma (S, P, J, W);
}
// Room of the bw x bh block at (px, py): how many samples it can move in EVERY direction with its SAD still read from the window (a
// 4-byte read at any byte offset takes 3 bytes beyond the block); negative: the block itself is not (wholly) inside
WH_FN int wh_win_room (const WhWin& W, int px, int py, int bw, int bh) {
  return wh_min (wh_min (px - W.x0, W.x0 + WH_WIN_STRIDE - (px + bw + 3)), wh_min (py - W.y0, W.y0 + WH_WIN_ROWS - (py + bh)));
}
// ... at least `need` of it, else the window is reloaded around the block (WH_WIN_MARGIN_X / _Y either side; the picture's expanded border
// bounds the window, never the room: a search does not leave the picture by more than the border either)
WH_FN void wh_win_need (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W, int px, int py, int bw, int bh, int need) {
  if (wh_win_room (W, px, py, bw, bh) >= need) return;
  wh_win_ensure (S, P, J, W, px - WH_WIN_MARGIN_X, py - WH_WIN_MARGIN_Y, px + bw + 3 + WH_WIN_MARGIN_X, py + bh + WH_WIN_MARGIN_Y);
}

// Speculative fetch of a macroblock's windows, issued together with its cold inputs as soon as the wave holds the ticket --
// before the neighbours are final, so before the motion vector predictor is known.  The guess is the most recent final
// 16x16 vector of the slice (camera motion and smooth motion fields make it a good one); the macroblock adopts the window when
// it covers what the search around the real predictor needs, otherwise it loads its own (exactly the round-1 behaviour).
// Either way every sample the macroblock reads is the reference picture's: the guess changes timing, never a result.
WH_FN void wh_win_speculate (const WhSeqParams& P, const WhPicJob& J, WhWin& W, int mbx, int mby, int guess_mv) {
  const int gx = wh_clip3 ((2 + (int) (int16_t) (guess_mv & 0xffff)) >> 2, -P.mv_range, P.mv_range), gy = wh_clip3 ((2 + (guess_mv >> 16)) >> 2, -P.mv_range, P.mv_range);
  wh_win_issue_all (P, J, W, mbx * 16 + gx, mby * 16 + gy);
}

// ---- lane geometry -----------------------------------------------------------------------------------
// SAD layout: lane = (row, 4-pixel segment) of a bw x bh block, lanes [0, bw*bh/4)
WH_FN int wh_sl_row (int lane, int bw) { return bw == 16 ? lane >> 2 : lane >> 1; }
WH_FN int wh_sl_col (int lane, int bw) { return bw == 16 ? (lane & 3) * 4 : (lane & 1) * 4; }
// SATD layout: lane quad = one 4x4 block (raster inside the partition), lane & 3 = row of the block
WH_FN int wh_tl_row (int lane, int bw) { const int b = lane >> 2; return (bw == 16 ? b >> 2 : b >> 1) * 4 + (lane & 3); }
WH_FN int wh_tl_col (int lane, int bw) { const int b = lane >> 2; return (bw == 16 ? b & 3 : b & 1) * 4; }
WH_FN uint32_t wh_enc4 (const WhInterLds& S, int x, int y) { return * (const uint32_t*)&S.m.enc_y[y * 16 + x]; }

// SAD of the bw x bh block at (ex,ey) of the source MB against the window at offset wo
WH_FN int wh_sad_win (const WhInterLds& S, const WhWin& W, int ex, int ey, int bw, int bh, int wo) {
  int s;
  const int n = (bw * bh) >> 2;
  WV_SUM (s, lane, (lane < n ? wh_sad4 (wh_enc4 (S, ex + wh_sl_col (lane, bw), ey + wh_sl_row (lane, bw)),
                                        wh_ld4u (W.b->win, wo + wh_sl_row (lane, bw) * WH_WIN_STRIDE + wh_sl_col (lane, bw))) : 0));
  return s;
}
// same against the reference picture in HBM (search candidates far away from the window)
WH_FN int wh_sad_global (const WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, int ex, int ey, int bw, int bh, int px, int py) {
  int s;
  const int n = (bw * bh) >> 2;
  const WH_G uint8_t* ref = (const WH_G uint8_t*)J.ref[0];
  WV_SUM (s, lane, (lane < n ? ([&] () { const WH_G uint8_t* t = ref + (ptrdiff_t) (py + wh_sl_row (lane, bw)) * P.rec_stride_y + px + wh_sl_col (lane, bw);
                                          return wh_sad4 (wh_enc4 (S, ex + wh_sl_col (lane, bw), ey + wh_sl_row (lane, bw)), wh_pack4 (t[0], t[1], t[2], t[3])); }) () : 0));
  return s;
}
// test-build statistics of the screen-content paths (which of them a test clip reaches at all): WELSHIP_SCC_STATS=1 prints them
#if defined(WH_EMU)
#include <stdio.h>
enum { WH_ST_STATIC_SKIP, WH_ST_SCROLL_SKIP, WH_ST_SCD_P16, WH_ST_CROSS_V, WH_ST_CROSS_H, WH_ST_FME, WH_ST_FME_HIT, WH_ST_DIR_TAKEN, WH_ST_P8X8, WH_ST_MERGE, WH_ST_FIXED, WH_ST_N, WH_ST_FME_SADS = WH_ST_N, WH_ST_FME_LISTED, WH_ST_LINE_SADS, WH_ST_ALL };
static long g_wh_scc_stat[WH_ST_ALL];
static void wh_scc_stat_dump() {
  if (!getenv ("WELSHIP_SCC_STATS")) return;
  static const char* n[WH_ST_N] = {"static_skip", "scroll_skip", "scd_p16x16", "cross_vertical", "cross_horizontal", "feature_search", "feature_hit", "directional_taken", "p8x8", "merged", "fixed_8x8"};
  for (int i = 0; i < WH_ST_N; ++i) fprintf (stderr, "welship scc stat %s %ld\n", n[i], g_wh_scc_stat[i]);
  fprintf (stderr, "welship scc work feature_candidates_listed %ld feature_sads %ld line_sads %ld\n", g_wh_scc_stat[WH_ST_FME_LISTED], g_wh_scc_stat[WH_ST_FME_SADS], g_wh_scc_stat[WH_ST_LINE_SADS]);
}
struct WhSccStatInit { WhSccStatInit() { atexit (wh_scc_stat_dump); } };
static WhSccStatInit g_wh_scc_stat_init;
#define WH_STAT(i) (++g_wh_scc_stat[i])
#else
#define WH_STAT(i) ((void)0)
#endif

// ---- screen content: SADs straight from the reference picture, one candidate per LANE --------------------------------
// (cross search and feature search look at hundreds of positions far apart: no window can hold them)
#if defined(WH_EMU)
WH_FN uint32_t wh_ldg4u (const uint8_t* p) { uint32_t v; memcpy (&v, p, 4); return v; }
#else
// four bytes at any byte address of device memory, from the two aligned words around it
WH_FN uint32_t wh_ldg4u (const WH_G uint8_t* p) {
  const WH_G uint32_t* w = (const WH_G uint32_t*) ((uintptr_t)p & ~ (uintptr_t)3);
  return __builtin_amdgcn_alignbyte (w[1], w[0], (uint32_t) (uintptr_t)p & 3u);
}
#endif
// this lane's SAD of the bw x bh source block at (ex,ey) against the block at p (any alignment)
WH_FN int wh_sad_lane_g (const WhInterLds& S, const WH_G uint8_t* p, int stride, int ex, int ey, int bw, int bh) {
  int s = 0;
  for (int r = 0; r < bh; ++r) {
    const WH_G uint8_t* q = p + (ptrdiff_t)r * stride;
#if defined(WH_EMU)
    for (int c = 0; c < bw; c += 4) s += wh_sad4 (wh_enc4 (S, ex + c, ey + r), wh_ldg4u (q + c));
#else
    const WH_G uint32_t* w = (const WH_G uint32_t*) ((uintptr_t)q & ~ (uintptr_t)3);
    const uint32_t sh = (uint32_t) (uintptr_t)q & 3u;
    uint32_t a = w[0];
    for (int c = 0; c < bw; c += 4) { const uint32_t b = w[(c >> 2) + 1]; s += wh_sad4 (wh_enc4 (S, ex + c, ey + r), __builtin_amdgcn_alignbyte (b, a, sh)); a = b; }
#endif
  }
  return s;
}

// SAD of the whole source MB against a 16x16 byte tile in LDS (stride 16)
WH_FN int wh_sad_mb_tile (const WhInterLds& S, const uint8_t* t) {
  int s;
  WV_SUM (s, lane, wh_sad4 (* (const uint32_t*)&S.m.enc_y[lane * 4], * (const uint32_t*)&t[lane * 4]));
  return s;
}
WH_FN int wh_sad_chroma_tile (const WhInterLds& S, const uint8_t* t) {
  int s;
  WV_SUM (s, lane, (lane < 32 ? wh_sad4 (* (const uint32_t*)&S.m.enc_c[lane * 4], * (const uint32_t*)&t[lane * 4]) : 0));
  return s;
}

// ---- motion compensation into LDS tiles -----------------------------------------------------------
// luma: bw x bh block at MB offset (bx,by), quarter-pel mv; dst stride 16, dst addressed like the MB
WH_FN void wh_mc_luma_to (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W, int mbx, int mby, int bx, int by, int bw, int bh,
                          int mvx, int mvy, uint8_t* dst) {
  const int ipx = mbx * 16 + bx + (mvx >> 2), ipy = mby * 16 + by + (mvy >> 2), fx = mvx & 3, fy = mvy & 3;
  wh_win_ensure (S, P, J, W, ipx - 2, ipy - 2, ipx + bw + 7, ipy + bh + 3);
  const int wo = (ipy - W.y0) * WH_WIN_STRIDE + ipx - W.x0, n = (bw * bh) >> 2;
  WV_LANES_BEGIN (lane)
  if (lane < n) {
    const int r = wh_sl_row (lane, bw), c = wh_sl_col (lane, bw);
    * (uint32_t*)&dst[(by + r) * 16 + bx + c] = wh_mc4 (W.b->win, wo + r * WH_WIN_STRIDE + c, fx, fy);
  }
  WV_LANES_END
}
// chroma: cw x ch block (both planes) at chroma offset (cx,cy) inside the MB; dst: Cb at 0, Cr at 64, stride 8
WH_FN void wh_mc_chroma_to (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, const WhWin& W, int mbx, int mby, int cx, int cy, int cw, int ch,
                            int mvx, int mvy, uint8_t* dst) {
  const int dx = mvx & 7, dy = mvy & 7;
  const int ipx = mbx * 8 + cx + (mvx >> 3), ipy = mby * 8 + cy + (mvy >> 3);
  const int n = cw * ch, sh = cw == 8 ? 3 : 2;
  const bool in_win = ipx >= W.cx0 && ipy >= W.cy0 && ipx + cw + 1 <= W.cx0 + WH_CWIN_COLS && ipy + ch + 1 <= W.cy0 + WH_CWIN_ROWS;
  if (in_win) WH_STAT_WIN (5); else WH_STAT_WIN (6);
  if (in_win) {
    // A lane makes four adjacent samples of one row of one plane: per source row the 8-sample group its first sample lies in and the
    // first word of the next group (12 bytes of that plane; wh_cwin_off), of which it takes samples s .. s + 3 (A) and s + 1 .. s + 4
    // (B, their right neighbours).  The weighted sum runs on two samples per register (fields of 16 bits: 64 x 255 + 32 fits).
    const int wx = ipx - W.cx0, wy = ipy - W.cy0;
    const uint32_t w00 = (uint32_t) ((8 - dx) * (8 - dy)), w10 = (uint32_t) (dx * (8 - dy)), w01 = (uint32_t) ((8 - dx) * dy), w11 = (uint32_t) (dx * dy);
    WV_LANES_BEGIN (lane)
    if (lane < (n >> 1)) {                  // n / 4 lanes per plane
      const int per = n >> 2, pl = lane >= per, k = lane - pl * per;
      const int x = cw == 8 ? (k & 1) * 4 : 0, y = cw == 8 ? k >> 1 : k;
      const int p = wx + x, sft = p & 7;
      const uint32_t* g = (const uint32_t*)&W.b->cwin[(wy + y) * WH_CWIN_STRIDE + ((p >> 3) << 4) + (pl << 3)];
      uint32_t r[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint32_t d0 = g[q * (WH_CWIN_STRIDE / 4)], d1 = g[q * (WH_CWIN_STRIDE / 4) + 1], d2 = g[q * (WH_CWIN_STRIDE / 4) + 4];
        const int t = sft + 1;
        r[q][0] = sft < 4 ? wh_funnel4 (d0, d1, sft) : wh_funnel4 (d1, d2, sft - 4);
        r[q][1] = t < 4 ? wh_funnel4 (d0, d1, t) : t < 8 ? wh_funnel4 (d1, d2, t - 4) : d2;
      }
      const uint32_t m = 0x00ff00ffu;
      const uint32_t lo = w00 * (r[0][0] & m) + w10 * (r[0][1] & m) + w01 * (r[1][0] & m) + w11 * (r[1][1] & m) + 0x00200020u;
      const uint32_t hi = w00 * ((r[0][0] >> 8) & m) + w10 * ((r[0][1] >> 8) & m) + w01 * ((r[1][0] >> 8) & m) + w11 * ((r[1][1] >> 8) & m) + 0x00200020u;
      * (uint32_t*)&dst[pl * 64 + (cy + y) * 8 + cx + x] = ((lo >> 6) & m) | (((hi >> 6) & m) << 8);
    }
    WV_LANES_END
  } else {
    WV_LANES_BEGIN (lane)
    for (int i = lane; i < 2 * n; i += 64) {
      const int pl = i >= n, k = i - pl * n, x = k & (cw - 1), y = k >> sh;
      const WH_G uint8_t* p = (const WH_G uint8_t*) (pl ? J.ref[2] : J.ref[1]) + (ptrdiff_t) (ipy + y) * P.rec_stride_c + ipx + x;
      dst[pl * 64 + (cy + y) * 8 + cx + x] = (uint8_t)wh_mc_chroma_w (p[0], p[1], p[P.rec_stride_c], p[P.rec_stride_c + 1], dx, dy);
    }
    WV_LANES_END
  }
}

// ---- motion vector prediction on the 5x6 cache ----------------------------------------------------
// The cache (5 rows x 6 cols of 4x4 blocks, row 0 / col 0 = neighbour MBs) lives in two lane tables: K.mv[i] = packed
// (mvx & 0xffff) | mvy << 16 and K.ref[i] = reference index of cache cell i = (by + 1) * 6 + bx + 1.
typedef struct WhMvCache { WvLaneArr mv, ref; } WhMvCache;
WH_FN int wh_pk_mv (int mx, int my) { return (int) (((unsigned)mx & 0xffffu) | ((unsigned)my << 16)); }
WH_FN int wh_mvx (int pk) { return (int) (int16_t) (pk & 0xffff); }
WH_FN int wh_mvy (int pk) { return pk >> 16; }
WH_FN int wh_cidx (int bx, int by) { return (by + 1) * 6 + bx + 1; }
WH_FN void wh_pred_mv (const WhMvCache& K, int bx, int by, int w, int ref, int* mx, int* my) {
  const int li = (by + 1) * 6 + bx, ti = by * 6 + bx + 1;
  const int lref = WV_LGET (K.ref, li), tref = WV_LGET (K.ref, ti);
  int di = ti + w;
  if (WV_LGET (K.ref, di) == WH_REF_NOT_AVAIL) di = ti - 1;
  const int dref = WV_LGET (K.ref, di);
  const int lmv = WV_LGET (K.mv, li), tmv = WV_LGET (K.mv, ti), dmv = WV_LGET (K.mv, di);
  if (tref == WH_REF_NOT_AVAIL && dref == WH_REF_NOT_AVAIL && lref != WH_REF_NOT_AVAIL) { *mx = wh_mvx (lmv); *my = wh_mvy (lmv); return; }
  const int match = (ref == lref) | ((ref == tref) << 1) | ((ref == dref) << 2);
  if (match == 1) { *mx = wh_mvx (lmv); *my = wh_mvy (lmv); }
  else if (match == 2) { *mx = wh_mvx (tmv); *my = wh_mvy (tmv); }
  else if (match == 4) { *mx = wh_mvx (dmv); *my = wh_mvy (dmv); }
  else { *mx = wh_median3 (wh_mvx (lmv), wh_mvx (tmv), wh_mvx (dmv)); *my = wh_median3 (wh_mvy (lmv), wh_mvy (tmv), wh_mvy (dmv)); }
}
WH_FN void wh_pred_16x8 (const WhMvCache& K, int part, int ref, int* mx, int* my) {
  const int i = part == 0 ? 1 : 18;
  if (ref == WV_LGET (K.ref, i)) { const int v = WV_LGET (K.mv, i); *mx = wh_mvx (v); *my = wh_mvy (v); return; }
  wh_pred_mv (K, 0, part * 2, 4, ref, mx, my);
}
WH_FN void wh_pred_8x16 (const WhMvCache& K, int part, int ref, int* mx, int* my) {
  int idx = 6;
  if (part != 0) { idx = 5; if (WV_LGET (K.ref, 5) == WH_REF_NOT_AVAIL) idx = 2; }
  if (ref == WV_LGET (K.ref, idx)) { const int v = WV_LGET (K.mv, idx); *mx = wh_mvx (v); *my = wh_mvy (v); return; }
  wh_pred_mv (K, part * 2, 0, 2, ref, mx, my);
}
WH_FN void wh_pred_skip_mv (const WhMvCache& K, int* mx, int* my) {
  const int lref = WV_LGET (K.ref, 6), tref = WV_LGET (K.ref, 1);
  if (lref == WH_REF_NOT_AVAIL || tref == WH_REF_NOT_AVAIL || (lref == 0 && WV_LGET (K.mv, 6) == 0) || (tref == 0 && WV_LGET (K.mv, 1) == 0)) { *mx = 0; *my = 0; return; }
  wh_pred_mv (K, 0, 0, 4, 0, mx, my);
}
// write mv/ref into the cache rectangle (bx,by,w,h in 4x4 units)
WH_FN bool wh_cache_in_rect (int cell, int bx, int by, int w, int h) {
  const int r = cell / 6 - 1, c = cell % 6 - 1;
  return cell < 30 && r >= by && r < by + h && c >= bx && c < bx + w;
}
WH_FN void wh_cache_set (WhMvCache& K, int bx, int by, int w, int h, int ref, int mx, int my) {
  WV_LSET_IF (K.ref, lane, wh_cache_in_rect (lane, bx, by, w, h), ref);
  WV_LSET_IF (K.mv, lane, wh_cache_in_rect (lane, bx, by, w, h), wh_pk_mv (mx, my));
}

// ---- partition slots ----------------------------------------------------------------------------------
WH_FN void wh_slot_geom (int slot, int* bx, int* by, int* bw, int* bh) {
  if (slot == 0) { *bx = 0; *by = 0; *bw = 16; *bh = 16; }
  else if (slot < WH_SLOT_16x8) { const int i = slot - WH_SLOT_8x8; *bx = (i & 1) * 8; *by = (i >> 1) * 8; *bw = 8; *bh = 8; }
  else if (slot < WH_SLOT_8x16) { const int i = slot - WH_SLOT_16x8; *bx = 0; *by = i * 8; *bw = 16; *bh = 8; }
  else { const int i = slot - WH_SLOT_8x16; *bx = i * 8; *by = 0; *bw = 8; *bh = 16; }
}
WH_FN void wh_slot_pred (const WhMvCache& K, int slot, int* mx, int* my) {
  if (slot == 0) wh_pred_mv (K, 0, 0, 4, 0, mx, my);
  else if (slot < WH_SLOT_16x8) { const int i = slot - WH_SLOT_8x8; wh_pred_mv (K, (i & 1) * 2, (i >> 1) * 2, 2, 0, mx, my); }
  else if (slot < WH_SLOT_8x16) wh_pred_16x8 (K, slot - WH_SLOT_16x8, 0, mx, my);
  else wh_pred_8x16 (K, slot - WH_SLOT_8x16, 0, mx, my);
}

// ---- one motion search (WelsMotionEstimateSearch) -------------------------------------------------
typedef struct WhMe {
  int bx, by, bw, bh;        // block inside the MB (pixels)
  int mvpx, mvpy;            // predictor (quarter-pel)
  int sad_pred;              // uiSadPred
  int mvx, mvy;              // result (quarter-pel)
  int sad_cost, satd_cost;   // uiSadCost / uiSatdCost
  int satd_raw;              // uSadPredISatd.uiSatd (complexity >= MEDIUM)
} WhMe;

typedef struct WhMeCtx {
  int mbx, mby, lambda, use_satd;
  int minx, miny, maxx, maxy;        // sMvStartMin / sMvStartMax (integer pel)
} WhMeCtx;

// SAD of the block against the reference at integer displacement (imx,imy): from the window when it is inside
WH_FN int wh_cand_sad (const WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, const WhWin& W, const WhMeCtx& C, const WhMe& me, int imx, int imy) {
  const int px = C.mbx * 16 + me.bx + imx, py = C.mby * 16 + me.by + imy;
  if (wh_win_covers (W, px, py, px + me.bw + 3, py + me.bh))
    return wh_sad_win (S, W, me.bx, me.by, me.bw, me.bh, (py - W.y0) * WH_WIN_STRIDE + px - W.x0);
  return wh_sad_global (S, P, J, me.bx, me.by, me.bw, me.bh, px, py);
}

// ---- screen content: what the search of one block does beyond the diamond (PreprocessSliceCoding, encoder_ext.cpp:2700-2765) ----
typedef struct WhSccMe {
  const WH_G WhSccJob* job;
  int method;                  // 0 WelsDiamondSearch, 1 WelsDiamondCrossSearch (16x16), 2 WelsDiamondCrossFeatureSearch (8x8 when the switch is on)
  uint32_t thr;                // uiSadCostThreshold of the block size: the cross / feature searches only run at or above it
  int dir_on, dmx, dmy;        // CheckDirectionalMv: the picture's scroll vector (pfSetScrollingMv == SetScrollingMvToMd), blocks below 16x16
  uint32_t chain;              // pMe->uiSadCost as the previous search that used this SWelsME left it (WhSccJob::chain)
  uint32_t fme_down;           // out: what the feature search took off the cost (pSlice->uiSliceFMECostDown)
} WhSccMe;

// LineFullSearch_c (svc_motion_estimate.cpp:568-613): every position of the column (or the row) through the CO-LOCATED block
// inside the search range, one candidate per lane, 64 at a time; the first minimum wins, and it replaces the search result
// only if it is cheaper.  Row search: neighbouring lanes read overlapping bytes of the same rows (two cache lines per load).
// Column search: lane l would read rows l .. l + bh - 1 -- 64 different lines per load instruction -- so the 64 + bh - 1 rows of
// the column a round needs are staged in LDS first (the search window's storage: the window is given up, whoever needs it next
// reloads it) and every lane takes its rows from there.
WH_FN void wh_line_search (const WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W, const WhMeCtx& C, const WhMe& me, bool vertical, int* best, int* bmx, int* bmy) {
  const int lo = vertical ? C.miny : C.minx, hi = vertical ? C.maxy : C.maxx;
  const int px = C.mbx * 16 + me.bx, py = C.mby * 16 + me.by;
  const WH_G uint8_t* ref = (const WH_G uint8_t*)J.ref[0];
  const int bw = me.bw, bh = me.bh, ex = me.bx, ey = me.by;
  int lbest = 0x7fffffff, lpos = 0;
  if (vertical) { W.x0 = -100000; W.y0 = -100000; }                // (wh_win_covers fails from here on)
  for (int base = lo; base < hi; base += 64) {
    int k, l;
    if (vertical) {
      uint8_t* tile = W.b->win;                                      // rows of 16 bytes
      const int rows = wh_min (64, hi - base) + bh - 1, words = bw >> 2;
      WV_LANES_BEGIN (lane)
      for (int i = lane; i < rows * words; i += 64) {
        const int r = i / words, c = (i - r * words) * 4;
        * (uint32_t*)&tile[r * 16 + c] = wh_ldg4u (ref + (ptrdiff_t) (py + base + r) * P.rec_stride_y + px + c);
      }
      WV_LANES_END
      WV_ARGMIN (k, l, lane, base + lane < hi, ([&] () {
        const int m = base + lane;
        WH_STAT (WH_ST_LINE_SADS);
        int s = 0;
        for (int r = 0; r < bh; ++r)
          for (int c = 0; c < bw; c += 4) s += wh_sad4 (wh_enc4 (S, ex + c, ey + r), * (const uint32_t*)&tile[(lane + r) * 16 + c]);
        return s + wh_mvd_cost (C.lambda, -me.mvpx, m * 4 - me.mvpy); }) ());
    } else {
      WV_ARGMIN (k, l, lane, base + lane < hi, ([&] () {
        const int m = base + lane;
        WH_STAT (WH_ST_LINE_SADS);
        return wh_sad_lane_g (S, ref + (ptrdiff_t)py * P.rec_stride_y + px + m, P.rec_stride_y, ex, ey, bw, bh) + wh_mvd_cost (C.lambda, m * 4 - me.mvpx, -me.mvpy); }) ());
    }
    if (l >= 0 && k < lbest) { lbest = k; lpos = base + l; }
  }
  if (lbest < *best) { *best = lbest; *bmx = vertical ? 0 : lpos; *bmy = vertical ? lpos : 0; }
}

// MotionEstimateFeatureFullSearch / FeatureSearchOne (svc_motion_estimate.cpp:918-1003): the blocks of the reference picture
// whose sample sum equals this block's, in raster order.  The reference walks them one by one with a running best cost and
// stops at the first one below the threshold; as the cost on entry is at or above that threshold, "the first one below the
// threshold, else the first minimum" is the same result, and that is what 64 candidates at a time compute.
WH_FN void wh_feature_search (const WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, const WH_G WhSccJob* Z, const WhMeCtx& C, const WhMe& me, uint32_t thr32,
                              int* best, int* bmx, int* bmy) {
  int feature;
  WV_SUM (feature, lane, (lane < 16 ? wh_sad4 (wh_enc4 (S, me.bx + (lane & 1) * 4, me.by + (lane >> 1)), 0u) : 0));
  if (feature >= Z->fme_list_size) return;
  const int times = (int)Z->fme_times[feature];
  const WH_G uint16_t* loc = (const WH_G uint16_t*)Z->fme_loc + 2 * (size_t)Z->fme_start[feature];
  const WH_G uint8_t* ref = (const WH_G uint8_t*)J.ref[0];
  const int cpx = C.mbx * 16 + me.bx, cpy = C.mby * 16 + me.by, cqx = cpx * 4, cqy = cpy * 4;
  const int min_qx = cqx + C.minx * 4, max_qx = cqx + C.maxx * 4, min_qy = cqy + C.miny * 4, max_qy = cqy + C.maxy * 4;
  const int thr = (int) (thr32 & 0xffffu);          // const uint16_t uiSadCostThresh
  int b = *best;
  for (int base = 0; base < times; base += 64) {
    WvLaneArr cst;
#if defined(WH_EMU)
    memset (&cst, 0, sizeof (cst));
#else
    cst = 0;
#endif
    WV_LSET_IF (cst, lane, true, ([&] () {
      const int i = base + lane;
      if (i >= times) return 0x7fffffff;
      WH_STAT (WH_ST_FME_LISTED);
      const int qx = (int)loc[2 * i], qy = (int)loc[2 * i + 1];
      if (qx > max_qx || qx < min_qx || qy > max_qy || qy < min_qy || qx == cqx || qy == cqy) return 0x7fffffff;
      const int mvdc = wh_mvd_cost (C.lambda, qx - cqx - me.mvpx, qy - cqy - me.mvpy);
      if (mvdc >= b) return 0x7fffffff;
      WH_STAT (WH_ST_FME_SADS);
      const WH_G uint8_t* q = ref + (ptrdiff_t) (qy >> 2) * P.rec_stride_y + (qx >> 2);
      return mvdc + wh_sad_lane_g (S, q, P.rec_stride_y, me.bx, me.by, me.bw, me.bh); }) ());
    int k, l;
    WV_ARGMIN (k, l, lane, WV_LOWN (cst, lane) < thr, lane);
    bool stop = false;
    if (l >= 0) { k = WV_LGET (cst, l); stop = true; }
    else WV_ARGMIN (k, l, lane, WV_LOWN (cst, lane) != 0x7fffffff, WV_LOWN (cst, lane));
    if (l >= 0 && k < b) {
      b = k;
      *bmx = ((int)loc[2 * (base + l)] >> 2) - cpx; *bmy = ((int)loc[2 * (base + l) + 1] >> 2) - cpy;
    }
    if (stop) break;
  }
  *best = b;
}

// SATD of the search result (CalculateSatdCost, complexity >= MEDIUM)
WH_FN void wh_me_satd (WhInterLds& S, const WhWin& W, const WhMeCtx& C, WhMe& me, int bmx, int bmy) {
  const int wo = (C.mby * 16 + me.by + bmy - W.y0) * WH_WIN_STRIDE + C.mbx * 16 + me.bx + bmx - W.x0;
  const int nq = (me.bw >> 2) * (me.bh >> 2) * 4;
  WV_SATD_ROWS (me.satd_raw, lane, lane < nq,
                wh_enc4 (S, me.bx + (lane < nq ? wh_tl_col (lane, me.bw) : 0), me.by + (lane < nq ? wh_tl_row (lane, me.bw) : 0)),
                wh_ld4u (W.b->win, wo + (lane < nq ? wh_tl_row (lane, me.bw) * WH_WIN_STRIDE + wh_tl_col (lane, me.bw) : 0)));
  me.satd_cost = me.satd_raw + wh_mvd_cost (C.lambda, me.mvx - me.mvpx, me.mvy - me.mvpy);
}

WH_FN void wh_motion_search (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W, const WhMeCtx& C, WhMe& me, const WvLaneArr& mvcl, int n_mvc, WhSccMe* Z = nullptr) {
  // initial point (svc_motion_estimate.cpp:222-284); candidates beyond the predictor: packed mvcl[0..n_mvc)
  int bmx = wh_clip3 ((2 + me.mvpx) >> 2, C.minx, C.maxx), bmy = wh_clip3 ((2 + me.mvpy) >> 2, C.miny, C.maxy);
  int best = wh_cand_sad (S, P, J, W, C, me, bmx, bmy) + wh_mvd_cost (C.lambda, bmx * 4 - me.mvpx, bmy * 4 - me.mvpy);
  for (int i = 0; i < n_mvc; ++i) {
    const int cand = WV_LGET (mvcl, i);
    const int cx = wh_clip3 ((2 + wh_mvx (cand)) >> 2, C.minx, C.maxx), cy = wh_clip3 ((2 + wh_mvy (cand)) >> 2, C.miny, C.maxy);
    if (cx != bmx || cy != bmy) {
      const int c = wh_cand_sad (S, P, J, W, C, me, cx, cy) + wh_mvd_cost (C.lambda, cx * 4 - me.mvpx, cy * 4 - me.mvpy);
      if (c < best) { best = c; bmx = cx; bmy = cy; }
    }
  }
  if (Z && Z->dir_on && (Z->dmx | Z->dmy) && Z->dmx >= C.minx && Z->dmx < C.maxx && Z->dmy >= C.miny && Z->dmy < C.maxy) {
    // CheckDirectionalMv: the scroll vector takes over whenever it is cheaper than what this SWelsME held BEFORE this search
    const int c = wh_cand_sad (S, P, J, W, C, me, Z->dmx, Z->dmy) + wh_mvd_cost (C.lambda, Z->dmx * 4 - me.mvpx, Z->dmy * 4 - me.mvpy);
    if ((uint32_t)c < Z->chain) { best = c; bmx = Z->dmx; bmy = Z->dmy; WH_STAT (WH_ST_DIR_TAKEN); }
  }
  const bool diamond = !(best < me.sad_pred);
  WH_STAT_BLK (16, me.bw, me.bh);
  const int bpx = C.mbx * 16 + me.bx, bpy = C.mby * 16 + me.by;
  if (diamond || C.use_satd) wh_win_need (S, P, J, W, bpx + bmx, bpy + bmy, me.bw, me.bh, diamond ? WH_WIN_START : 0);
  if (diamond) {
    // WelsDiamondSearch (svc_motion_estimate.cpp:335-379)
    int dx = bmx * 4 - me.mvpx, dy = bmy * 4 - me.mvpy;
    int wo = (bpy + bmy - W.y0) * WH_WIN_STRIDE + bpx + bmx - W.x0;     // window offset of the current centre
    const int n = (me.bw * me.bh) >> 2;
    // Step `it` probes the four neighbours of a centre at most `it` samples away from the start: with `budget` samples of room at the start
    // the steps below `budget` read inside the window whatever way the walk goes; then the room is looked at again (and the window follows
    // the walk when it is used up)
    int budget = wh_win_room (W, bpx + bmx, bpy + bmy, me.bw, me.bh);
    for (int it = 0; it < 16; ++it) {
      const int cmx = (dx + me.mvpx) >> 2, cmy = (dy + me.mvpy) >> 2;
      if (!(cmx >= C.minx && cmx < C.maxx && cmy >= C.miny && cmy < C.maxy)) break;   // CheckMvInRange fails: nothing changes any more
      if (it >= budget) {
        if (wh_win_room (W, bpx + cmx, bpy + cmy, me.bw, me.bh) < 1) {
          WH_STAT_WIN (4);
          wh_win_need (S, P, J, W, bpx + cmx, bpy + cmy, me.bw, me.bh, WH_WIN_START);
          wo = (bpy + cmy - W.y0) * WH_WIN_STRIDE + bpx + cmx - W.x0;
        }
        budget = it + wh_max (1, wh_win_room (W, bpx + cmx, bpy + cmy, me.bw, me.bh));
      }
#if defined(WH_EMU)
      if (wh_win_room (W, bpx + cmx, bpy + cmy, me.bw, me.bh) < 1 || wo != (bpy + cmy - W.y0) * WH_WIN_STRIDE + bpx + cmx - W.x0) { fprintf (stderr, "emu: diamond step outside the search window\n"); abort(); }
#endif
      // four SADs: up, down, left, right -- two packed 16-bit partial sums per reduction
      WH_STAT_BLK (19, me.bw, me.bh);
      int pud, plr;
      WV_SUM2 (pud, plr, lane,
               (lane < n ? ([&] () { const int r = wh_sl_row (lane, me.bw), c = wh_sl_col (lane, me.bw);
                                     const uint32_t e = wh_enc4 (S, me.bx + c, me.by + r); const int o = wo + r * WH_WIN_STRIDE + c;
                                     return wh_sad4 (e, wh_ld4u (W.b->win, o - WH_WIN_STRIDE)) | (wh_sad4 (e, wh_ld4u (W.b->win, o + WH_WIN_STRIDE)) << 16); }) () : 0),
               (lane < n ? ([&] () { const int r = wh_sl_row (lane, me.bw), c = wh_sl_col (lane, me.bw);
                                     const uint32_t e = wh_enc4 (S, me.bx + c, me.by + r); const int o = wo + r * WH_WIN_STRIDE + c;
                                     return wh_sad4 (e, wh_ld4u (W.b->win, o - 1)) | (wh_sad4 (e, wh_ld4u (W.b->win, o + 1)) << 16); }) () : 0));
      const int c0 = (pud & 0xffff) + wh_mvd_cost (C.lambda, dx, dy - 4);
      const int c1 = (int) ((unsigned)pud >> 16) + wh_mvd_cost (C.lambda, dx, dy + 4);
      const int c2 = (plr & 0xffff) + wh_mvd_cost (C.lambda, dx - 4, dy);
      const int c3 = (int) ((unsigned)plr >> 16) + wh_mvd_cost (C.lambda, dx + 4, dy);
      const int in_cost = best;
      int ix = 0, iy = 0;
      if (c0 < best) { best = c0; ix = 0; iy = 1; }
      if (c1 < best) { best = c1; ix = 0; iy = -1; }
      if (c2 < best) { best = c2; ix = 1; iy = 0; }
      if (c3 < best) { best = c3; ix = -1; iy = 0; }
      if (best == in_cost) break;
      dx -= ix * 4; dy -= iy * 4;
      wo -= ix + iy * WH_WIN_STRIDE;
    }
    bmx = (dx + me.mvpx) >> 2; bmy = (dy + me.mvpy) >> 2;
    if (Z && Z->method >= 1) {
      // WelsDiamondCrossSearch / WelsDiamondCrossFeatureSearch (svc_motion_estimate.cpp:1055-1094)
      if ((uint32_t)best >= Z->thr) {
        WH_STAT (WH_ST_CROSS_V);
        wh_line_search (S, P, J, W, C, me, true, &best, &bmx, &bmy);
        if ((uint32_t)best >= Z->thr) { WH_STAT (WH_ST_CROSS_H); wh_line_search (S, P, J, W, C, me, false, &best, &bmx, &bmy); }
      }
      if (Z->method == 2 && (uint32_t)best >= Z->thr) {
        const int before = best;
        WH_STAT (WH_ST_FME);
        wh_feature_search (S, P, J, Z->job, C, me, Z->thr, &best, &bmx, &bmy);
        if (best < before) WH_STAT (WH_ST_FME_HIT);
        Z->fme_down += (uint32_t) (before - best);
      }
      if (C.use_satd) wh_win_need (S, P, J, W, bpx + bmx, bpy + bmy, me.bw, me.bh, 0);       // (the line searches gave the window up)
    }
  }
  me.mvx = bmx * 4; me.mvy = bmy * 4;
  me.sad_cost = best; me.satd_cost = best; me.satd_raw = 0;
  if (C.use_satd) wh_me_satd (S, W, C, me, bmx, bmy);
}

// WelsMotionEstimateSearchStatic / WelsMotionEstimateSearchScrolled (svc_motion_estimate.cpp:186-218): no search at all -- the
// block takes the given integer vector (zero, or the picture's scroll vector)
WH_FN void wh_me_fixed (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W, const WhMeCtx& C, WhMe& me, int imx, int imy) {
  me.sad_cost = wh_cand_sad (S, P, J, W, C, me, imx, imy) + wh_mvd_cost (C.lambda, imx * 4 - me.mvpx, imy * 4 - me.mvpy);
  me.mvx = imx * 4; me.mvy = imy * 4;
  me.satd_cost = me.sad_cost; me.satd_raw = 0;
  if (C.use_satd) {
    wh_win_need (S, P, J, W, C.mbx * 16 + me.bx + imx, C.mby * 16 + me.by + imy, me.bw, me.bh, 0);
    wh_me_satd (S, W, C, me, imx, imy);
  }
}

// search results per partition slot, one lane table per field
typedef struct WhMeTab { WvLaneArr mv, sad, satd, raw; } WhMeTab;
WH_FN void wh_me_store (WhMeTab& T, int slot, const WhMe& me) {
  WV_LSET (T.mv, slot, wh_pk_mv (me.mvx, me.mvy)); WV_LSET (T.sad, slot, me.sad_cost); WV_LSET (T.satd, slot, me.satd_cost); WV_LSET (T.raw, slot, me.satd_raw);
}
WH_FN void wh_me_fetch (const WhMeTab& T, int slot, WhMe& me) {
  const int v = WV_LGET (T.mv, slot);
  me.mvx = wh_mvx (v); me.mvy = wh_mvy (v); me.sad_cost = WV_LGET (T.sad, slot); me.satd_cost = WV_LGET (T.satd, slot); me.satd_raw = WV_LGET (T.raw, slot);
}

// ---- fractional refinement (MeRefineFracPixel / MeRefineQuarPixel, md.cpp:575-769) -------------------------------
// The search always starts from an integer position, so the candidate set is fixed: the four half-sample neighbours
// (top, bottom, left, right: two h-type and two b-type samples), then the four quarter-sample neighbours of the best
// of those five -- each quarter candidate is the average of two samples of which at least one is already known.
// Everything below is per lane: four horizontally adjacent samples packed in a word, `o` = window offset of the
// lane's integer samples G.

// vertical half samples above (hu, between rows -1 and 0) and below (hd) G
WH_FN void wh_rf_h_pair (const uint8_t* w, int o, uint32_t* hu, uint32_t* hd) {
  uint32_t r[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) r[k] = wh_ld4u (w, o + (k - 3) * WH_WIN_STRIDE);
  int u[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    u[k] = wh_clip255 ((wh_tap6 (WH_BYTE (r[0], k), WH_BYTE (r[1], k), WH_BYTE (r[2], k), WH_BYTE (r[3], k), WH_BYTE (r[4], k), WH_BYTE (r[5], k)) + 16) >> 5);
    d[k] = wh_clip255 ((wh_tap6 (WH_BYTE (r[1], k), WH_BYTE (r[2], k), WH_BYTE (r[3], k), WH_BYTE (r[4], k), WH_BYTE (r[5], k), WH_BYTE (r[6], k)) + 16) >> 5);
  }
  *hu = wh_pack4 (u[0], u[1], u[2], u[3]); *hd = wh_pack4 (d[0], d[1], d[2], d[3]);
}
// unclipped horizontal 6-tap sums at the five positions x-1 .. x+3 of the row at o (bytes o-3 .. o+6)
WH_FN void wh_htaps5 (const uint8_t* w, int o, int* t /*[5]*/) {
  const uint32_t a = wh_ld4u (w, o - 3), b = wh_ld4u (w, o + 1), c = wh_ld4u (w, o + 5);
  int q[10];
#pragma unroll
  for (int k = 0; k < 4; ++k) { q[k] = WH_BYTE (a, k); q[4 + k] = WH_BYTE (b, k); }
  q[8] = WH_BYTE (c, 0); q[9] = WH_BYTE (c, 1);
#pragma unroll
  for (int k = 0; k < 5; ++k) t[k] = wh_tap6 (q[k], q[k + 1], q[k + 2], q[k + 3], q[k + 4], q[k + 5]);
}
// horizontal half samples left (bl, between x-1 and x) and right (br) of G
WH_FN void wh_rf_b_pair (const uint8_t* w, int o, uint32_t* bl, uint32_t* br) {
  int t[5];
  wh_htaps5 (w, o, t);
  int v[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) v[k] = wh_clip255 ((t[k] + 16) >> 5);
  *bl = wh_pack4 (v[0], v[1], v[2], v[3]); *br = wh_pack4 (v[1], v[2], v[3], v[4]);
}
// centre half samples j at (o-1, o): the two that flank a vertical half sample h4 (o)
WH_FN void wh_rf_j_pair_h (const uint8_t* w, int o, uint32_t* jl, uint32_t* jr) {
  int t[6][5];
#pragma unroll
  for (int k = 0; k < 6; ++k) wh_htaps5 (w, o + (k - 2) * WH_WIN_STRIDE, t[k]);
  int v[5];
#pragma unroll
  for (int x = 0; x < 5; ++x) v[x] = wh_clip255 ((wh_tap6 (t[0][x], t[1][x], t[2][x], t[3][x], t[4][x], t[5][x]) + 512) >> 10);
  *jl = wh_pack4 (v[0], v[1], v[2], v[3]); *jr = wh_pack4 (v[1], v[2], v[3], v[4]);
}
// centre half samples j at (o - stride, o): the two that flank a horizontal half sample b4 (o)
WH_FN void wh_rf_j_pair_v (const uint8_t* w, int o, uint32_t* ju, uint32_t* jd) {
  int t[7][4];
#pragma unroll
  for (int k = 0; k < 7; ++k) wh_htaps4 (w, o + (k - 3) * WH_WIN_STRIDE, &t[k][0], &t[k][1], &t[k][2], &t[k][3]);
  int u[4], d[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    u[x] = wh_clip255 ((wh_tap6 (t[0][x], t[1][x], t[2][x], t[3][x], t[4][x], t[5][x]) + 512) >> 10);
    d[x] = wh_clip255 ((wh_tap6 (t[1][x], t[2][x], t[3][x], t[4][x], t[5][x], t[6][x]) + 512) >> 10);
  }
  *ju = wh_pack4 (u[0], u[1], u[2], u[3]); *jd = wh_pack4 (d[0], d[1], d[2], d[3]);
}
// the four quarter-sample candidates (0 top, 1 bottom, 2 left, 3 right) around the half-stage winner hb (-1: the integer position
// itself, 0..3: half candidate hb), from the stage-1 samples G, hu / hd (vertical half samples above / below), bl / br (horizontal
// half samples left / right): each is the average of two samples of which at least one is already known; the centre half samples
// (j) a half-sample winner needs are interpolated once for both of its candidates
WH_FN void wh_rf_quarters (const uint8_t* w, int o, int hb, uint32_t G, uint32_t hu, uint32_t hd, uint32_t bl, uint32_t br, uint32_t* q) {
  uint32_t ja, jb;
  switch (hb) {
  case -1: q[0] = wh_avg4 (G, hu); q[1] = wh_avg4 (G, hd); q[2] = wh_avg4 (G, bl); q[3] = wh_avg4 (G, br); break;
  case 0:                                                    // hu, the h sample of the row above (at o - stride)
    wh_rf_j_pair_h (w, o - WH_WIN_STRIDE, &ja, &jb);
    q[0] = wh_avg4 (wh_ld4u (w, o - WH_WIN_STRIDE), hu); q[1] = wh_avg4 (G, hu); q[2] = wh_avg4 (ja, hu); q[3] = wh_avg4 (jb, hu); break;
  case 1:                                                    // hd
    wh_rf_j_pair_h (w, o, &ja, &jb);
    q[0] = wh_avg4 (G, hd); q[1] = wh_avg4 (wh_ld4u (w, o + WH_WIN_STRIDE), hd); q[2] = wh_avg4 (ja, hd); q[3] = wh_avg4 (jb, hd); break;
  case 2:                                                    // bl, the b sample of the column to the left (at o - 1)
    wh_rf_j_pair_v (w, o - 1, &ja, &jb);
    q[0] = wh_avg4 (ja, bl); q[1] = wh_avg4 (jb, bl); q[2] = wh_avg4 (wh_ld4u (w, o - 1), bl); q[3] = wh_avg4 (G, bl); break;
  default:                                                   // br
    wh_rf_j_pair_v (w, o, &ja, &jb);
    q[0] = wh_avg4 (ja, br); q[1] = wh_avg4 (jb, br); q[2] = wh_avg4 (G, br); q[3] = wh_avg4 (wh_ld4u (w, o + 1), br); break;
  }
}

// Returns through me.mvx/mvy/satd_cost, writes the final luma prediction of the block into S.m.pred_y.
// `sub8`: a 16x8 / 8x16 partition that TryModeMerge made of two 8x8 searches (screen content) keeps BLOCK_8x8 as its block size,
// so the reference scores every candidate on the partition's first 8x8 only (pfMeCost[pMe->uiBlockSize], md.cpp:602-640) while it
// interpolates and copies the whole partition.
// Every candidate's samples are interpolated ONCE and stay in the lane's registers (lane tables): the stage-1 samples feed the quarter
// candidates, and the winner's samples become the prediction without being computed again.
WH_FN void wh_refine_frac (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, WhWin& W, const WhMeCtx& C, WhMe& me, int satd_in_md, int sub8 = 0) {
  const int bpx = C.mbx * 16 + me.bx, bpy = C.mby * 16 + me.by;
  WH_STAT_BLK (22, me.bw, me.bh);
  const int ipx = bpx + (me.mvx >> 2), ipy = bpy + (me.mvy >> 2);          // me.mv is integer-pel here
  wh_win_ensure (S, P, J, W, ipx - 4, ipy - 4, ipx + me.bw + 8, ipy + me.bh + 5);
  const int wo = (ipy - W.y0) * WH_WIN_STRIDE + ipx - W.x0;
  const int nq = (me.bw >> 2) * (me.bh >> 2) * 4, bw = me.bw, ex = me.bx, ey = me.by;
#define WH_RF_ACT (lane < nq && (!sub8 || (bw == 16 ? ((lane >> 2) & 3) < 2 : lane < 16)))
#define WH_RF_ENC wh_enc4 (S, ex + (lane < nq ? wh_tl_col (lane, bw) : 0), ey + (lane < nq ? wh_tl_row (lane, bw) : 0))
#define WH_RF_O (wo + (lane < nq ? wh_tl_row (lane, bw) * WH_WIN_STRIDE + wh_tl_col (lane, bw) : 0))
  const int dmx = me.mvx - me.mvpx, dmy = me.mvy - me.mvpy;
  WvLaneArr vG, vhu, vhd, vbl, vbr, vq0, vq1, vq2, vq3;
#if defined(WH_EMU)
  memset (&vG, 0, sizeof (vG)); memset (&vhu, 0, sizeof (vhu)); memset (&vhd, 0, sizeof (vhd)); memset (&vbl, 0, sizeof (vbl)); memset (&vbr, 0, sizeof (vbr));
  memset (&vq0, 0, sizeof (vq0)); memset (&vq1, 0, sizeof (vq1)); memset (&vq2, 0, sizeof (vq2)); memset (&vq3, 0, sizeof (vq3));
#else
  vG = 0; vhu = 0; vhd = 0; vbl = 0; vbr = 0; vq0 = 0; vq1 = 0; vq2 = 0; vq3 = 0;
#endif
  // integer position + the four half-sample candidates (independent of each other: the SATDs overlap)
  int c_int, c0, c1, c2, c3;
  WV_DECLARE_LANE (lane);                   // one lane id for everything below: the candidates share most of their window loads
  WV_LANE_EVAL (lane, {
    const int o = WH_RF_O;
    uint32_t a, b;
    WV_LOWN (vG, lane) = (int)wh_ld4u (W.b->win, o);
    wh_rf_h_pair (W.b->win, o, &a, &b); WV_LOWN (vhu, lane) = (int)a; WV_LOWN (vhd, lane) = (int)b;
    wh_rf_b_pair (W.b->win, o, &a, &b); WV_LOWN (vbl, lane) = (int)a; WV_LOWN (vbr, lane) = (int)b; });
  if (satd_in_md) c_int = me.satd_raw;                                      // uiSatd of the integer search
  else WV_SATD_ROWS_SHARED (c_int, lane, WH_RF_ACT, WH_RF_ENC, (uint32_t)WV_LOWN (vG, lane));
  WV_SATD_ROWS4_SHARED (c0, c1, c2, c3, lane, WH_RF_ACT, WH_RF_ENC, (uint32_t)WV_LOWN (vhu, lane), (uint32_t)WV_LOWN (vhd, lane), (uint32_t)WV_LOWN (vbl, lane), (uint32_t)WV_LOWN (vbr, lane));
  int best = c_int + wh_mvd_cost (C.lambda, dmx, dmy), hb = -1;
  c0 += wh_mvd_cost (C.lambda, dmx, dmy - 2); if (c0 < best) { best = c0; hb = 0; }
  c1 += wh_mvd_cost (C.lambda, dmx, dmy + 2); if (c1 < best) { best = c1; hb = 1; }
  c2 += wh_mvd_cost (C.lambda, dmx - 2, dmy); if (c2 < best) { best = c2; hb = 2; }
  c3 += wh_mvd_cost (C.lambda, dmx + 2, dmy); if (c3 < best) { best = c3; hb = 3; }
  const int hx = hb == 2 ? -2 : hb == 3 ? 2 : 0, hy = hb == 0 ? -2 : hb == 1 ? 2 : 0;    // winner of the half stage, relative
  // quarter-sample candidates around it
  WV_LANE_EVAL (lane, {
    uint32_t q[4];
    wh_rf_quarters (W.b->win, WH_RF_O, hb, (uint32_t)WV_LOWN (vG, lane), (uint32_t)WV_LOWN (vhu, lane), (uint32_t)WV_LOWN (vhd, lane), (uint32_t)WV_LOWN (vbl, lane), (uint32_t)WV_LOWN (vbr, lane), q);
    WV_LOWN (vq0, lane) = (int)q[0]; WV_LOWN (vq1, lane) = (int)q[1]; WV_LOWN (vq2, lane) = (int)q[2]; WV_LOWN (vq3, lane) = (int)q[3]; });
  WV_SATD_ROWS4_SHARED (c0, c1, c2, c3, lane, WH_RF_ACT, WH_RF_ENC, (uint32_t)WV_LOWN (vq0, lane), (uint32_t)WV_LOWN (vq1, lane), (uint32_t)WV_LOWN (vq2, lane), (uint32_t)WV_LOWN (vq3, lane));
  int qb = -1;
  c0 += wh_mvd_cost (C.lambda, dmx + hx, dmy + hy - 1); if (c0 < best) { best = c0; qb = 0; }
  c1 += wh_mvd_cost (C.lambda, dmx + hx, dmy + hy + 1); if (c1 < best) { best = c1; qb = 1; }
  c2 += wh_mvd_cost (C.lambda, dmx + hx - 1, dmy + hy); if (c2 < best) { best = c2; qb = 2; }
  c3 += wh_mvd_cost (C.lambda, dmx + hx + 1, dmy + hy); if (c3 < best) { best = c3; qb = 3; }
  const int qx = qb == 2 ? -1 : qb == 3 ? 1 : 0, qy = qb == 0 ? -1 : qb == 1 ? 1 : 0;
  me.satd_cost = best;
  // the winner's samples become the prediction
  WV_LANE_EVAL (lane, {
    if (lane < nq) {
      const int v = qb >= 0 ? (qb == 0 ? WV_LOWN (vq0, lane) : qb == 1 ? WV_LOWN (vq1, lane) : qb == 2 ? WV_LOWN (vq2, lane) : WV_LOWN (vq3, lane))
                  : hb >= 0 ? (hb == 0 ? WV_LOWN (vhu, lane) : hb == 1 ? WV_LOWN (vhd, lane) : hb == 2 ? WV_LOWN (vbl, lane) : WV_LOWN (vbr, lane)) : WV_LOWN (vG, lane);
      * (uint32_t*)&S.m.pred_y[(ey + wh_tl_row (lane, bw)) * 16 + ex + wh_tl_col (lane, bw)] = (uint32_t)v;
    } });
  WV_SYNC();
#undef WH_RF_ACT
#undef WH_RF_ENC
#undef WH_RF_O
  me.mvx += hx + qx; me.mvy += hy + qy;
}

// ---- inter luma residual (WelsEncInterY) on S.m.res after wh_dct_luma16; returns cbp luma -----------
WH_FN int wh_enc_inter_y (WhMbLds& S, int qp) {
  wh_quant_blocks (S, 0, 16, qp, qp, S.res, 0);
  // per 8x8: score = sum over its four 4x4 blocks (9 when a level exceeds 1, else the run score); the reference stops
  // adding once an 8x8 reaches 6, which cannot change the two threshold tests below
  int s0, s1, s2, s3;
  WV_QUADSUM4 (s0, s1, s2, s3, lane, (lane < 16 ? WH_Q_SCORE (S.part2[lane]) : 0));          // lanes 0..15 = blocks 0..15 in luma4x4BlkIdx order: a quad is an 8x8
  int cbp = 0;
  if (s0 + s1 + s2 + s3 >= 6) cbp = (s0 >= 4 ? 1 : 0) | (s1 >= 4 ? 2 : 0) | (s2 >= 4 ? 4 : 0) | (s3 >= 4 ? 8 : 0);
  WV_LANES_BEGIN (lane)
  {
    const int b = lane >> 2, on = (cbp >> (b >> 2)) & 1;
    for (int q = 0; q < 4; ++q) { const int k = (lane & 3) * 4 + q; S.lv_luma[b * 16 + k] = S.res[b * 16 + wh_zigzag (k)]; }
    if (lane < 16) {
      int n = 0;
      if ((cbp >> (lane >> 2)) & 1) n = __builtin_popcount (WH_Q_MASK (S.part2[lane]));
      S.nzc[wh_blk_y (lane) * 4 + wh_blk_x (lane)] = (uint8_t)n;
    }
    (void)on;
  }
  WV_LANES_END
  WV_LANES_BEGIN (lane)
  {
    const int b = lane >> 2, on = (cbp >> (b >> 2)) & 1;
    for (int k = 0; k < 4; ++k) {
      const int i = lane * 4 + k, pos = i & 15;
      S.res[i] = on ? (int16_t) (S.res[i] * wh_dq (qp, pos)) : (int16_t)0;
    }
  }
  WV_LANES_END
  return cbp;
}

// quant-to-zero tests of the P_Skip path (WelsTryPYskip / WelsTryPUVskip); operate on copies in S.tmp
WH_FN bool wh_try_py_skip (WhMbLds& S, int qp) {
  wh_quant_blocks (S, 0, 16, qp, qp, S.tmp, 0);
  int bc, u1, u2, u3;                  // sixteen lanes: one DPP row.  Score sum (<= 16 x 9) | number of big blocks << 16
  WV_ROWSUM4 (bc, u1, u2, u3, lane, (lane < 16 ? WH_Q_SCORE (S.part2[lane]) | (WH_Q_BIG (S.part2[lane]) << 16) : 0));      // (the score only counts when no block is big)
  (void)u1; (void)u2; (void)u3;
  return bc < 6;
}
// both chroma planes at once (the reference tests Cb, then Cr: both must pass)
WH_FN bool wh_try_puv_skip (WhMbLds& S, int qpc) {
  // WelsHadamardQuant2x2Skip_c (encode_mb_aux.cpp:226-245)
  const int ff = wh_ff_inter (qpc, 0) << 1, mf = wh_mf (qpc, 0) >> 1;
  const int16_t thr = (int16_t) (((1 << 16) - 1) / mf - ff);
  for (int pl = 0; pl < 2; ++pl) {
    const int16_t* r = &S.res[256 + pl * 64];
    const int16_t s0 = (int16_t) (r[0] + r[32]), s1 = (int16_t) (r[0] - r[32]), s2 = (int16_t) (r[16] + r[48]), s3 = (int16_t) (r[16] - r[48]);
    const int16_t d0 = (int16_t) (s0 + s2), d1 = (int16_t) (s0 - s2), d2 = (int16_t) (s1 + s3), d3 = (int16_t) (s1 - s3);
    if (wh_abs (d0) > thr || wh_abs (d1) > thr || wh_abs (d2) > thr || wh_abs (d3) > thr) return false;
  }
  wh_quant_blocks (S, 256, 8, qpc, qpc, S.tmp, 1);
  int c0, c1, c2, c3;          // per plane (a quad of lanes each): score sum | number of big blocks << 8
  WV_QUADSUM4 (c0, c1, c2, c3, lane, (lane < 8 ? WH_Q_SCORE (S.part2[lane]) | (WH_Q_BIG (S.part2[lane]) << 8) : 0));
  (void)c2; (void)c3;
  return c0 < 7 && c1 < 7;     // (a big block makes the word >= 256)
}

// ---- "cold" inputs of a P macroblock: data no kernel writes while the picture is being coded (source samples,
// previous source picture, the reference picture's MB states).  They come straight from HBM, so a wave starts their
// copy for its NEXT macroblock as soon as the macroblock in hand is complete -- LDS-DMA (no registers held) straight to where the body
// reads them, under the wait for the next macroblock's neighbours.
// PLAIN (here and in wh_inter_mb_body_t): the picture has none of the optional per-picture inputs -- host VAA SADs, the layer's pSadCost array,
// background flags, inter-layer hints, a QP map, GOM rate control, MB ranges, bit counting, a temporal-layer vector shift (common/wh_types.h
// WH_SEQ_PLAIN: the pictures of a session group) -- so none of them is looked at.
// VAR (an integer, shared with wh_inter_mb_body_t): 0 = everything may be there; 1, 2 = PLAIN; 3 = a camera picture of the frame API: the
// inputs a host's pre-processing supplies (VAA SADs, pSadCost, background flags, the vector shift) may be there, the control inputs (inter-layer
// hints, QP map, GOM rate control, MB ranges, bit counting) are not (WH_SEQ_NO_CTRL; candidate, -DWH_FRAME_KERNEL=1)
template <int VAR = 0>
WH_FN void wh_inter_cold_fetch (WhInterLds& S, WhInterStage& G, int lane, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  constexpr bool HOSTIN = VAR == 0 || VAR == 3, CTRL = VAR == 0;
  const int w = P.mb_w, xy = mby * w + mbx;
  // (macroblock-tiled source pictures, WH_SRC_*: luma = 256 consecutive bytes in lane order, both chroma blocks = the 128 behind them: the very
  //  order of m.enc_y / m.enc_c)
  wh_ld_async4 ((const WH_G uint8_t*)J.src[0] + WH_SRC_Y_OFF (w, mbx, mby, 0, 0) + lane * 4, (uint32_t*)S.m.enc_y, lane);
  if (lane < 32) wh_ld_async4 ((const WH_G uint8_t*)J.src[0] + WH_SRC_C_OFF (w, mbx, mby, 0, 0, 0) + lane * 4, (uint32_t*)S.m.enc_c, lane);
  if (P.complexity == 0 && (!HOSTIN || !J.vaa_sad8x8))     // VAA 8x8 SADs (LOW complexity only), unless the host supplies them
    wh_ld_async4 ((const WH_G uint8_t*)J.prev_src_y + WH_SRC_Y_OFF (w, mbx, mby, 0, 0) + lane * 4, G.cold_pv, lane);
  // the reference picture's state of this MB (an I picture's has no motion / SAD: zeros), words 144..179 of nb; lanes 36 / 37: co_mv
  if (lane < 36) { if (J.ref_mbs) wh_ld_async4 ((const WH_G uint32_t*) ((const WH_G WhMbState*)J.ref_mbs + xy) + lane, &S.nb[144], lane); else S.nb[144 + lane] = 0u; }
  if (J.ref_is_p) {
    if (lane >= 36 && lane < 38) {
      const bool ok = lane == 36 ? mbx < P.mb_w - 1 : mby < P.mb_h - 1;
      const WH_G WhMbState* o = (const WH_G WhMbState*)J.ref_mbs + xy + (lane == 36 ? 1 : w);
      if (ok) wh_ld_async4 (&o->p16mv[0], &S.nb[144], lane);
    }
  }
  // pSadCost[0] of the layer's SMB array; the host's four VAA SADs take the place of the previous source picture's first words (cold_pv 0..3)
  if (HOSTIN && lane == 0 && J.sad_cost0) wh_ld_async4 ((CTRL && J.sad_cost0_out && J.dyn_redo && xy == J.mb_begin ? (const WH_G int32_t*)J.sad_cost0_out : (const WH_G int32_t*)J.sad_cost0) + xy, &S.sad_cost0_in, lane);
  if (HOSTIN && lane < 4 && J.vaa_sad8x8) wh_ld_async4 ((const WH_G int32_t*)J.vaa_sad8x8 + xy * 4 + lane, G.cold_pv, lane);
}

typedef struct WhInterCtx {
  int slice_idc, slice_first;        // slice of this MB and its first MB address
  WhWinLds* win;                        // this wave's search windows
  int spec_valid;                       // windows were fetched speculatively (wh_win_speculate) ...
  WhWin spec;                           // ... at this placement
  int* last_mv;                         // out (may be NULL): the slice's most recent final 16x16 vector, packed -- the next guess
} WhInterCtx;

// ---- the P macroblock -----------------------------------------------------------------------------
// VAR: 0 = the general body, 1 = PLAIN (see wh_inter_cold_fetch), 2 = PLAIN and LOW complexity known at compile time (SAD costs: the SATD
// paths of the search, the refinement and the intra test are not compiled in), 3 = the frame API's camera pictures without control inputs
// XWG: the slice is coded by SEVERAL workgroups (hip_backend.hip k_inter_split: a launch of a few pictures spreads every slice over several compute units):
// the neighbours' states and samples are loaded past the caches and this macroblock's are stored write-through (wave.h wh_ld_x32 / wh_st_x); not with
// screen content or the control inputs (their chains through the slice stay inside one workgroup)
template <bool SCC, int VAR = 0, bool XWG = false>
WH_FN void wh_inter_mb_body_t (WhInterLds& S, WhInterStage& G, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, WhInterCtx& X) {
  constexpr bool LOW = VAR == 2, HOSTIN = VAR == 0 || VAR == 3, CTRL = VAR == 0;      // (see wh_inter_cold_fetch)
  WH_PROF_DECL (P);
  WhMbLds& M = S.m;
  const int w = P.mb_w, xy = mby * w + mbx;
  const int slice_idc = X.slice_idc;
  const int avail = wh_mb_avail_in_slice (P, mbx, mby, X.slice_first);
  WhMbCtl ctl;
  if (!CTRL) { ctl.qp_delta = 0; ctl.stale_cbp = 0; ctl.cell12_valid = 0; ctl.pad = 0; ctl.cell12_mv[0] = ctl.cell12_mv[1] = 0; }
  else ctl = wh_mb_ctl (J, xy);
  // GOM-level rate control inside the kernel: the QP of this macroblock's group, settled by the last macroblock of the group before
  // it (wh_gom_close_if_last) -- which this macroblock has waited for (WhPicJob::scc_chain_prev)
  const int qp = (CTRL && J.gom_rc) ? wh_clip3 ((int)wh_ld_wg32 ((const WH_G uint32_t*)& ((const WH_G WhGomRc*)J.gom_rc)->calc_qp), 0, 51) : wh_mb_qp (J, ctl);
  const int qpc = kWhChromaQp[wh_clip3 (qp + P.chroma_qp_offset, 0, 51)];
  const int lambda = kWhLambda[qp];
  const int use_satd = LOW ? 0 : P.complexity > 0;        // pfMdCost == SATD, pfCalculateSatd == CalculateSatdCost
  const bool md_using_sad = !use_satd;          // bMdUsingSad (svc_encode_slice.cpp:699)
  const bool ref_is_p = J.ref_is_p != 0;

  WH_PROF_MARK (P, M, 8);   // kernel arguments, job descriptor, slice lookup
  // ---- batch 1: commit the cold inputs fetched earlier; load neighbour pixels + neighbour MB states (L2-warm: the
  //      neighbours were coded by waves of this very workgroup) ----
  WV_LANES_BEGIN (lane)
  {
    WhTileRegs tr;
    wh_tile_fetch_nb<XWG> (lane, P, J, mbx, mby, &tr);
    // The four neighbours' states by LDS-DMA straight into S.nb (round 6; rounds 1-5: three loads per lane into registers, selects, three LDS
    // stores -- 135 vector instructions of address arithmetic and a division by 36 per load).  Top-left, top and top-right are ONE run of 108
    // words of the picture's state array, which nb[0..107] mirrors; the left state follows in nb[108..143].  A neighbour that does not exist (or
    // belongs to another slice) is not copied: its words keep whatever they held and are never read (TLm / Tm / TRm / Lm are null then).
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int d = lane + 64 * k;
      const bool ok = d < 36 ? (avail & WH_AV_TOPLEFT) != 0 : d < 72 ? (avail & WH_AV_TOP) != 0 : d < 108 ? (avail & WH_AV_TOPRIGHT) != 0 : d < 144 && (avail & WH_AV_LEFT) != 0;
      const int wofs = d < 108 ? (xy - w - 1) * 36 + d : (xy - 1) * 36 + (d - 108);         // word offset inside the state array (36 words per state)
      if (ok) wh_ld_async4_x<XWG> ((const WH_G uint32_t*)J.mbs + wofs, &S.nb[64 * k], lane);
    }
    WH_PROF_SUB (P, M, 3);       /* detail: neighbour loads issued */
    wh_tile_commit_nb (M, lane, &tr);      // (the source samples and the reference picture's state of this MB are in place already: wh_inter_cold_fetch)
  }
  WV_LANES_END
  WV_ASYNC_WAIT();

  WH_PROF_MARK (P, M, 9);   // batch 1 loads
  // ---- neighbour cache (FillNeighborCacheInterWithoutBGD) ----
  const WhMbState* TLm = (avail & WH_AV_TOPLEFT) ? (const WhMbState*)&S.nb[0] : nullptr;
  const WhMbState* Tm = (avail & WH_AV_TOP) ? (const WhMbState*)&S.nb[36] : nullptr;
  const WhMbState* TRm = (avail & WH_AV_TOPRIGHT) ? (const WhMbState*)&S.nb[72] : nullptr;
  const WhMbState* Lm = (avail & WH_AV_LEFT) ? (const WhMbState*)&S.nb[108] : nullptr;
  const WhMbState* Co = (const WhMbState*)&S.nb[144];
  const int tl_type = TLm ? TLm->mb_type : WH_MB_NONE, t_type = Tm ? Tm->mb_type : WH_MB_NONE;
  const int tr_type = TRm ? TRm->mb_type : WH_MB_NONE, l_type = Lm ? Lm->mb_type : WH_MB_NONE;
  const bool l_inter = WH_IS_INTER (l_type), t_inter = WH_IS_INTER (t_type), tl_inter = WH_IS_INTER (tl_type), tr_inter = WH_IS_INTER (tr_type);
  WhMvCache K;
  WhMeTab T;
  WvLaneArr mvcl;
#if defined(WH_EMU)
  memset (&K, 0, sizeof (K)); memset (&T, 0, sizeof (T)); memset (&mvcl, 0, sizeof (mvcl));
#else
  K.mv = 0; K.ref = 0; T.mv = 0; T.sad = 0; T.satd = 0; T.raw = 0; mvcl = 0;
#endif
  // cell (r,c) of the 5x6 cache: row 0 / col 0 come from the neighbour MB states, the rest is this MB (not coded yet;
  // the reference pre-marks 9,11,17,21,23 the same way)
#define WH_CACHE_REF(lane) ([&] () { const int r = (lane) / 6, c = (lane) % 6; \
    if (r == 0 && c == 0) return tl_inter ? (int)TLm->ref_idx[3] : TLm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; \
    if (r == 0 && c == 5) return tr_inter ? (int)TRm->ref_idx[2] : TRm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; \
    if (r == 0) return t_inter ? (int)Tm->ref_idx[2 + ((c - 1) >> 1)] : Tm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; \
    if (c == 0) return l_inter ? (int)Lm->ref_idx[((r - 1) >> 1) * 2 + 1] : Lm ? WH_REF_NOT_IN_LIST : WH_REF_NOT_AVAIL; \
    return WH_REF_NOT_AVAIL; }) ()
#define WH_CACHE_MV(lane) ([&] () { const int r = (lane) / 6, c = (lane) % 6; \
    if (r == 0 && c == 0) return tl_inter ? wh_pk_mv (TLm->mv[15][0], TLm->mv[15][1]) : 0; \
    if (r == 0 && c == 5) return tr_inter ? wh_pk_mv (TRm->mv[12][0], TRm->mv[12][1]) : 0; \
    if (r == 0) return t_inter ? wh_pk_mv (Tm->mv[12 + c - 1][0], Tm->mv[12 + c - 1][1]) : 0; \
    if (c == 0) return l_inter ? wh_pk_mv (Lm->mv[(r - 1) * 4 + 3][0], Lm->mv[(r - 1) * 4 + 3][1]) : 0; \
    return 0; }) ()
  WV_LSET_IF (K.ref, lane, lane < 30, WH_CACHE_REF (lane));
  WV_LSET_IF (K.mv, lane, lane < 30, WH_CACHE_MV (lane));
  if (ctl.cell12_valid) {       // re-encode pass after one that ended as P8x16 (WhMbCtl)
    WV_LSET (K.ref, 12, 0);
    WV_LSET (K.mv, 12, wh_pk_mv (ctl.cell12_mv[0], ctl.cell12_mv[1]));
  }
#undef WH_CACHE_REF
#undef WH_CACHE_MV
  // neighbour SAD / skip context, order of the reference's caches: [0] top-left, [1] top, [2] top-right, [3] left
  const int sadc0 = tl_inter ? TLm->sad_cost[0] : 0, sadc1 = t_inter ? Tm->sad_cost[0] : 0, sadc2 = tr_inter ? TRm->sad_cost[0] : 0, sadc3 = l_inter ? Lm->sad_cost[0] : 0;
  const bool tl_sk = tl_type == WH_MB_PSKIP, t_sk = t_type == WH_MB_PSKIP, tr_sk = tr_type == WH_MB_PSKIP, l_sk = l_type == WH_MB_PSKIP;
  // background detection (pVaaBackgroundMbFlag): this MB's flag and its neighbours'; entry k of the lane table: 0 this MB,
  // 1 left, 2 top, 3 top-right, 4 top-left (only read for neighbours that exist)
  const bool bgd = HOSTIN && J.bgd_flags != nullptr;
  WvLaneArr bgf;
#if defined(WH_EMU)
  memset (&bgf, 0, sizeof (bgf));
#else
  bgf = 0;
#endif
  if (bgd) {
    WV_LSET_IF (bgf, lane, lane < 5, ([&] () { const int off = lane == 0 ? 0 : lane == 1 ? -1 : lane == 2 ? -w : lane == 3 ? -w + 1 : -w - 1;
                                              return xy + off >= 0 ? (int) ((const WH_G int8_t*)J.bgd_flags)[xy + off] : 0; }) ());
  }
  // inter-layer hints of the highest spatial layer (WelsMdInterMbEnhancelayer): base vector candidate and "base MB is intra"
  int il_mv = 0;
  bool il_intra = false;
  if (CTRL && J.il_hint) {
    WvLaneArr ilh;
#if defined(WH_EMU)
    memset (&ilh, 0, sizeof (ilh));
#else
    ilh = 0;
#endif
    WV_LSET_IF (ilh, lane, lane < 3, (int) ((const WH_G int16_t*)J.il_hint)[xy * 4 + lane]);
    il_mv = wh_pk_mv (WV_LGET (ilh, 0), WV_LGET (ilh, 1));
    il_intra = (WV_LGET (ilh, 2) & 1) != 0;
  }
  const bool bg_cur = bgd && WV_LGET (bgf, 0) != 0, bg_l = bgd && WV_LGET (bgf, 1) != 0, bg_t = bgd && WV_LGET (bgf, 2) != 0;
  const bool bg_tr = bgd && WV_LGET (bgf, 3) != 0, bg_tl = bgd && WV_LGET (bgf, 4) != 0;
  // skip context of PredictSadSkip: with background detection a skipped neighbour only counts when it is not background
  // (FillNeighborCacheInterWithBGD, md.cpp:253-372)
  const bool tl_skc = tl_sk && !bg_tl, t_skc = t_sk && !bg_t, tr_skc = tr_sk && !bg_tr, l_skc = l_sk && !bg_l;
  const int sadsk0 = tl_skc ? TLm->skip_sad : 0, sadsk1 = t_skc ? Tm->skip_sad : 0, sadsk2 = tr_skc ? TRm->skip_sad : 0, sadsk3 = l_skc ? Lm->skip_sad : 0;
  const int ref_mb_type = ref_is_p ? Co->mb_type : WH_MB_NONE;
  const bool ref_mb_bg = Co->ref_type == WH_REFTYPE_BACKGROUND;         // a background skip is not MB_TYPE_SKIP where the reference asks for exactly that
  WhMeCtx C;
  C.mbx = mbx; C.mby = mby; C.lambda = lambda; C.use_satd = use_satd;
  C.minx = wh_max (- ((mbx + 1) << 4) + 3, -P.mv_range); C.miny = wh_max (- ((mby + 1) << 4) + 3, -P.mv_range);
  C.maxx = wh_min (((P.mb_w - mbx) << 4) - 3, P.mv_range); C.maxy = wh_min (((P.mb_h - mby) << 4) - 3, P.mv_range);

  WH_PROF_MARK (P, M, 10);  // neighbour cache + context
  // ---- batch 2: reference windows centred on the 16x16 predictor (= the search's initial point) ----
  WhMe me16;
  me16.bx = 0; me16.by = 0; me16.bw = 16; me16.bh = 16;
  wh_pred_mv (K, 0, 0, 4, 0, &me16.mvpx, &me16.mvpy);
  WhWin W;
  W.b = X.win;
  WhI16Cost i16c;
  {
    const int icx = wh_clip3 ((2 + me16.mvpx) >> 2, C.minx, C.maxx), icy = wh_clip3 ((2 + me16.mvpy) >> 2, C.miny, C.maxy);
    const int cx = mbx * 16 + icx, cy = mby * 16 + icy;
    const bool adopt = X.spec_valid && wh_win_room (X.spec, cx, cy, 16, 16) >= WH_WIN_START;
    WH_STAT_WIN (0);
    if (adopt) { W.x0 = X.spec.x0; W.y0 = X.spec.y0; W.cx0 = X.spec.cx0; W.cy0 = X.spec.cy0; WH_STAT_WIN (2); WH_PROF_MARK (P, M, 15); }     // fetched while the wave waited for its neighbours
    else {
      WH_STAT_WIN (3);
      if (X.spec_valid) WV_ASYNC_WAIT();       // (landed long ago: the batch-1 loads were issued after it)
      wh_win_issue_all (P, J, W, cx, cy);
    }
    // the Intra16x16 mode costs need no reference samples: computed while the window loads are in flight
    wh_i16_costs (M, avail, use_satd, lambda, &i16c);
    WV_ASYNC_WAIT();
  }
  WH_PROF_MARK (P, M, 0);   // mvp + batch 2 (window) loads
  int mb_type = WH_MB_P16x16, cbp = 0, cost_luma = 0, cost_skip_mb = 0, sad_cost0 = 0;
  int p16x = 0, p16y = 0;                       // sP16x16Mv
  int skx = 0, sky = 0;
  bool done = false;

  // PredictSadSkip (md.cpp:872-910)
  auto predict_sad_skip = [&] () {
    const int rb = WV_LGET (K.ref, 1), ra = WV_LGET (K.ref, 6);
    int rc = WV_LGET (K.ref, 5);
    const int sb = sadsk1, sa = sadsk3;
    int sc = sadsk2, skip_c = tr_skc;
    if (rc == WH_REF_NOT_AVAIL) { rc = WV_LGET (K.ref, 0); sc = sadsk0; skip_c = tl_skc; }
    if (rb == WH_REF_NOT_AVAIL && rc == WH_REF_NOT_AVAIL && ra != WH_REF_NOT_AVAIL) return sa;
    const int cnt = ((0 == ra) && l_skc) | (((0 == rb) && t_skc) << 1) | (((0 == rc) && skip_c) << 2);
    return cnt == 1 ? sa : cnt == 2 ? sb : cnt == 4 ? sc : wh_median3 (sa, sb, sc);
  };
  const bool try_skip = l_sk || t_sk || tl_sk || tr_sk;
  bool keep_skip = l_sk && t_sk && tr_sk;
  bool b_skip = false;
  const int stale_cbp = ctl.stale_cbp;
  bool bg_coded = false, bg_skip = false, collocated = false;      // bCollocatedPredFlag (WelsMdUpdateBGDInfo)

  // ---- background detection (WelsMdInterJudgeBGDPskip, svc_mode_decision.cpp:216-260) ----
  if (bgd) {
    keep_skip = keep_skip && !bg_l && !bg_t && !bg_tr;
    const int ref_type_raw = (int)Co->ref_type - 1;                // uiRefMbType as the reference picture's buffer holds it (0: never written)
    const bool ref_intra = Co->ref_type != 0 && Co->ref_type != WH_REFTYPE_BACKGROUND && WH_IS_INTRA (ref_type_raw);
    const int ref_qp = Co->ref_qp;
    if (bg_cur && !ref_intra && (ref_qp - qp <= 3 /* DELTA_QP_BGD_THD */ || ref_qp <= 26)) {
      // CheckChromaCost: chroma of the co-located block against the source
      wh_mc_chroma_to (S, P, J, W, mbx, mby, 0, 0, 8, 8, 0, 0, M.pred_c);
      int cb, cr;
      WV_SUM2 (cb, cr, lane, (lane < 16 ? wh_sad4 (* (const uint32_t*)&M.enc_c[lane * 4], * (const uint32_t*)&M.pred_c[lane * 4]) : 0),
               (lane >= 16 && lane < 32 ? wh_sad4 (* (const uint32_t*)&M.enc_c[lane * 4], * (const uint32_t*)&M.pred_c[lane * 4]) : 0));
      const bool too_large = cb > 640 || cr > 640;                 // KNOWN_CHROMA_TOO_LARGE
      const int chroma_sad = cb + cr, pred_skip = predict_sad_skip();
      const bool cannot = (pred_skip > 128 && chroma_sad >= pred_skip) ||       // SMALLEST_INVISIBLE; IsCostLessEqualSkipCost
                          (ref_is_p && ref_mb_type == WH_MB_PSKIP && !ref_mb_bg && Co->skip_sad > 128 && chroma_sad >= Co->skip_sad);
      if (!cannot && !too_large) {
        // WelsMdBackgroundMbEnc (svc_base_layer_md.cpp:1352-1421): prediction = the co-located block (luma always with a zero
        // vector); P_Skip when the skip predictor is zero too, else P16x16 with a zero vector and the usual residual
        int sx, sy;
        wh_pred_skip_mv (K, &sx, &sy);
        bg_coded = true; bg_skip = sx == 0 && sy == 0; collocated = true;
        wh_mc_luma_to (S, P, J, W, mbx, mby, 0, 0, 16, 16, 0, 0, M.pred_y);
        const uint8_t* pl = M.pred_y;
        int sad;
        WV_SUM (sad, lane, wh_sad4 (* (const uint32_t*)&M.enc_y[lane * 4], * (const uint32_t*)&pl[lane * 4]));
        sad_cost0 = sad;
        p16x = 0; p16y = 0;
        if (bg_skip) { b_skip = true; mb_type = WH_MB_PSKIP; skx = 0; sky = 0; cost_luma = 0; cost_skip_mb = 0; done = true; }
        else {
          mb_type = WH_MB_P16x16;
          if (md_using_sad) cost_luma = sad;
          else WV_SATD_ROWS (cost_luma, lane, true, wh_enc4 (S, wh_tl_col (lane, 16), wh_tl_row (lane, 16)), * (const uint32_t*)&M.pred_y[wh_tl_row (lane, 16) * 16 + wh_tl_col (lane, 16)]);
          WV_LANES_BEGIN (lane)
          if (lane < 16) { S.mv_out[lane][0] = 0; S.mv_out[lane][1] = 0; S.mvp_out[lane][0] = (int16_t)me16.mvpx; S.mvp_out[lane][1] = (int16_t)me16.mvpy; }
          WV_LANES_END
          wh_dct_luma16 (M);
          cbp = wh_enc_inter_y (M, qp);
          cbp |= wh_encrec_chroma (M, qpc, 0) << 4;
          wh_idct_luma16 (M);
          wh_idct_chroma (M);
          done = true;
        }
      }
    }
  }

  // ---- screen content: static / scrolled P_Skip (WelsMdInterJudgeSCDPskip, svc_mode_decision.cpp:326-541) ----
  // A macroblock whose four 8x8 blocks the pre-processing found unchanged against the co-located (or the scrolled) block of the
  // reference's SOURCE picture, and whose chroma is identical there too, is predicted with that integer vector without any search:
  // P_Skip when the skip predictor happens to be that vector (and the reference MB's QP is close), P16x16 with residual otherwise.
  const WH_G WhSccJob* Z = nullptr;
  int idc0 = 0, idc1 = 0, idc2 = 0, idc3 = 0, scroll_on = 0, smx = 0, smy = 0;
  uint32_t dch0 = 0, dch1 = 0, dch2 = 0, dch3 = 0, fme_down_mb = 0;      // size-limited slices: the incoming chain, this macroblock's cost-down sum
  bool scd_coded = false;
  if (SCC) {
    Z = (const WH_G WhSccJob*)J.scc;
    const int scd_on = Z->scd_on;
    if (scd_on) {                                                    // SetBlockStaticIdcToMd: part of WelsMdInterJudgeSCDPskip; else all four stay NO_STATIC
      const WH_G uint8_t* ip = (const WH_G uint8_t*)Z->static_idc + (size_t) (2 * mby) * (2 * w) + 2 * mbx;
      idc0 = ip[0]; idc1 = ip[1]; idc2 = ip[2 * w]; idc3 = ip[2 * w + 1];
    }
    const int sflag = Z->scroll_flag;
    smx = Z->scroll_mvx; smy = Z->scroll_mvy;
    scroll_on = sflag && (smx | smy);                                // pfSetScrollingMv == SetScrollingMvToMd (encoder_ext.cpp:2707-2713)
    if (J.dyn_slice && scroll_on && J.scc_chain_prev) {
      // size-limited slices: the chain of the 8x8 searches (below) per macroblock -- take over what the previous macroblock of THIS slice
      // that may search 8x8 blocks left (nothing when it lies before the slice's first macroblock) and pass it on unchanged unless this
      // macroblock gets to the 8x8 searches itself
      const int pv = ((const WH_G int32_t*)J.scc_chain_prev)[xy];
      WH_G uint32_t* cm = (WH_G uint32_t*)Z->chain_mb;
      if (pv >= J.dyn_first) { dch0 = wh_ld_wg32 (cm + 4 * pv); dch1 = wh_ld_wg32 (cm + 4 * pv + 1); dch2 = wh_ld_wg32 (cm + 4 * pv + 2); dch3 = wh_ld_wg32 (cm + 4 * pv + 3); }
      WV_LANES_BEGIN (lane)
      if (lane == 0) { wh_st_wg32 (cm + 4 * xy, dch0); wh_st_wg32 (cm + 4 * xy + 1, dch1); wh_st_wg32 (cm + 4 * xy + 2, dch2); wh_st_wg32 (cm + 4 * xy + 3, dch3); }
      WV_LANES_END
    }
    for (int mode = 0; scd_on && mode < 2 && !done; ++mode) {        // STATIC, SCROLLED
      const int want = mode == 0 ? 1 : 2;                            // COLLOCATED_STATIC / SCROLLED_STATIC
      if (mode == 1 && !sflag) break;
      if (!(idc0 == want && idc1 == want && idc2 == want && idc3 == want)) continue;
      const int ox = mode ? smx : 0, oy = mode ? smy : 0;
      if (mode == 1 && ((mbx << 4) + ox < 0 || (mbx << 4) + ox > ((P.mb_w - 1) << 4) || (mby << 4) + oy < 0 || (mby << 4) + oy > ((P.mb_h - 1) << 4))) continue;   // CheckBorder
      if (!Z->ref_ori_c[0]) continue;
      int cb, cr;
      {
        const WH_G uint8_t* o0 = (const WH_G uint8_t*)Z->ref_ori_c[0];
        const WH_G uint8_t* o1 = (const WH_G uint8_t*)Z->ref_ori_c[1];
        const ptrdiff_t off = (ptrdiff_t) ((mby << 3) + (oy >> 1)) * P.src_stride_c + (mbx << 3) + (ox >> 1);
        WV_SUM2 (cb, cr, lane,
                 (lane < 16 ? wh_sad4 (* (const uint32_t*)&M.enc_c[lane * 4], wh_ldg4u (o0 + off + (ptrdiff_t) ((lane >> 1) & 7) * P.src_stride_c + (lane & 1) * 4)) : 0),
                 (lane >= 16 && lane < 32 ? wh_sad4 (* (const uint32_t*)&M.enc_c[lane * 4], wh_ldg4u (o1 + off + (ptrdiff_t) ((lane >> 1) & 7) * P.src_stride_c + (lane & 1) * 4)) : 0));
      }
      if (cb != 0 || cr != 0) continue;
      // MdInterSCDPskipProcess + SvcMdSCDMbEnc
      const int ref_qp = Co->ref_qp;
      const bool qp_similar = ref_qp - qp <= 5 /* DELTA_QP_SCD_THD */ || ref_qp <= 26;
      int sx, sy;
      wh_pred_skip_mv (K, &sx, &sy);
      const int vx = (int) (int16_t) (ox * 4), vy = (int) (int16_t) (oy * 4);
      const bool as_skip = qp_similar && sx == vx && sy == vy;
      uint8_t* dy = M.pred_y;
      uint8_t* dc = M.pred_c;
      wh_mc_luma_to (S, P, J, W, mbx, mby, 0, 0, 16, 16, vx, vy, dy);
      wh_mc_chroma_to (S, P, J, W, mbx, mby, 0, 0, 8, 8, vx, vy, dc);
      int sad;
      WV_SUM (sad, lane, wh_sad4 (* (const uint32_t*)&M.enc_y[lane * 4], * (const uint32_t*)&dy[lane * 4]));
      sad_cost0 = sad; cost_skip_mb = sad;                           // iCostSkipMb = pSadCost[0]: luma only
      p16x = vx; p16y = vy;
      scd_coded = true;
      if (as_skip) { b_skip = true; mb_type = WH_MB_PSKIP; skx = vx; sky = vy; cost_luma = 0; collocated = vx == 0 && vy == 0; done = true; WH_STAT (mode ? WH_ST_SCROLL_SKIP : WH_ST_STATIC_SKIP); }
      else {
        WH_STAT (WH_ST_SCD_P16);
        mb_type = WH_MB_P16x16;
        collocated = false;                                          // bCollocatedPredFlag keeps WelsMdInterInit's false on this path
        if (md_using_sad) cost_luma = sad;
        else cost_luma = wh_cand_sad (S, P, J, W, C, me16, 0, 0);    // ... against pRefLuma WITHOUT the vector (svc_mode_decision.cpp:461-463)
        WV_LANES_BEGIN (lane)
        if (lane < 16) { S.mv_out[lane][0] = (int16_t)vx; S.mv_out[lane][1] = (int16_t)vy; S.mvp_out[lane][0] = (int16_t)me16.mvpx; S.mvp_out[lane][1] = (int16_t)me16.mvpy; }
        WV_LANES_END
        me16.mvx = vx; me16.mvy = vy;
        wh_dct_luma16 (M);
        cbp = wh_enc_inter_y (M, qp);
        cbp |= wh_encrec_chroma (M, qpc, 0) << 4;
        wh_idct_luma16 (M);
        wh_idct_chroma (M);
        done = true;
      }
    }
  }

  // ---- P_Skip test (WelsMdInterJudgePskip / WelsMdPSkipEnc) ----
  if (!done && ((ref_is_p && ref_mb_type == WH_MB_PSKIP) || try_skip)) {
    const int sad_pred_skip = predict_sad_skip();
    wh_pred_skip_mv (K, &skx, &sky);
    const int nx = (mbx << 4) + (skx >> 2), ny = (mby << 4) + (sky >> 2);
    if (!(nx < -29 || nx > (P.mb_w << 4) + 12 || ny < -29 || ny > (P.mb_h << 4) + 12)) {
      wh_mc_luma_to (S, P, J, W, mbx, mby, 0, 0, 16, 16, skx, sky, M.pred_y);
      WH_PROF_SUB (P, M, 4);     /* detail: P_Skip luma prediction */
      wh_mc_chroma_to (S, P, J, W, mbx, mby, 0, 0, 8, 8, skx, sky, M.pred_c);
      WH_PROF_SUB (P, M, 5);     /* detail: P_Skip chroma prediction */
      int sad_l, sad_c;                       // luma and chroma SAD of the skip prediction in one reduction
      WV_SUM2 (sad_l, sad_c, lane, wh_sad4 (* (const uint32_t*)&S.m.enc_y[lane * 4], * (const uint32_t*)&M.pred_y[lane * 4]),
               (lane < 32 ? wh_sad4 (* (const uint32_t*)&S.m.enc_c[lane * 4], * (const uint32_t*)&M.pred_c[lane * 4]) : 0));
      const int sad_mb = sad_l + sad_c;
      bool ok = sad_mb == 0 || sad_mb < sad_pred_skip || (ref_is_p && ref_mb_type == WH_MB_PSKIP && !ref_mb_bg && sad_mb < Co->skip_sad);
      WH_PROF_SUB (P, M, 6);     /* detail: P_Skip SADs + decision */
      WH_STAT_WIN (25);
      if (!ok) {
        WH_STAT_WIN (26);
        // residual would quantise to nothing?  (WelsDctMb + WelsTryPYskip + WelsTryPUVskip; the transform reads the skip prediction where it lies)
        wh_dct_luma16 (M);
        if (wh_try_py_skip (M, qp)) {
          wh_dct_chroma (M);
          if (wh_try_puv_skip (M, qpc)) ok = true;
        }
      }
      if (ok) {
        b_skip = true;
        if (md_using_sad) cost_luma = sad_l;
        else WV_SATD_ROWS (cost_luma, lane, true, wh_enc4 (S, wh_tl_col (lane, 16), wh_tl_row (lane, 16)), * (const uint32_t*)&M.pred_y[wh_tl_row (lane, 16) * 16 + wh_tl_col (lane, 16)]);
        // pSadCost[0] is only refreshed when bMdUsingSad; otherwise the SMB entry keeps the previous frame's value
        sad_cost0 = md_using_sad ? sad_l : ((HOSTIN && J.sad_cost0) ? (int)S.sad_cost0_in : Co->sad_cost[0]);
        cost_skip_mb = sad_mb;
        p16x = skx; p16y = sky;
      }
    }
  }
  if (b_skip && keep_skip) { mb_type = WH_MB_PSKIP; done = true; }
  WH_PROF_MARK (P, M, 1);   // P_Skip test

  int sad_pred16 = 0;
  if (!done && !b_skip && !il_intra) {
    // PredictSad (md.cpp:826-870)
    int sad_pred;
    {
      const int rb = WV_LGET (K.ref, 1), ra = WV_LGET (K.ref, 6);
      int rc = WV_LGET (K.ref, 5), sc = sadc2;
      if (rc == WH_REF_NOT_AVAIL) { rc = WV_LGET (K.ref, 0); sc = sadc0; }
      if (rb == WH_REF_NOT_AVAIL && rc == WH_REF_NOT_AVAIL && ra != WH_REF_NOT_AVAIL) sad_pred = sadc3;
      else {
        const int cnt = (0 == ra) | ((0 == rb) << 1) | ((0 == rc) << 2);
        sad_pred = cnt == 1 ? sadc3 : cnt == 2 ? sadc1 : cnt == 4 ? sc : wh_median3 (sadc3, sadc1, sc);
      }
      const int v = sad_pred << 6;
      sad_pred = ((v - (v >> 3) + (v >> 5)) + 32) >> 6;
    }
    // ---- P16x16 (WelsMdP16x16): candidates = base (0), left/top P16x16 mv, co-located right/below of the ref ----
    int nm = 1;
    const bool c_l = Lm != nullptr, c_t = Tm != nullptr, c_r = ref_is_p && mbx < P.mb_w - 1, c_b = ref_is_p && mby < P.mb_h - 1;
    const int i_l = 1, i_t = i_l + (c_l ? 1 : 0), i_r = i_t + (c_t ? 1 : 0), i_b = i_r + (c_r ? 1 : 0);
    nm = i_b + (c_b ? 1 : 0);
    WV_LSET (mvcl, 0, il_mv);                   // sMvBase: zero, or twice the base layer's vector (SetMvBaseEnhancelayer)
    if (c_l) WV_LSET (mvcl, i_l, wh_pk_mv (Lm->p16mv[0], Lm->p16mv[1]));
    if (c_t) WV_LSET (mvcl, i_t, wh_pk_mv (Tm->p16mv[0], Tm->p16mv[1]));
    const int msh = HOSTIN ? J.mvc_shift : 0;       // temporal layers: the reference picture's vectors span 2^shift picture intervals
    if (c_r) WV_LSET (mvcl, i_r, wh_pk_mv (S.co_mv[0][0] >> msh, S.co_mv[0][1] >> msh));
    if (c_b) WV_LSET (mvcl, i_b, wh_pk_mv (S.co_mv[1][0] >> msh, S.co_mv[1][1] >> msh));
    me16.sad_pred = sad_pred;
    if (SCC) {
      WhSccMe z;
      z.job = Z; z.method = 1; z.thr = Z->thr16; z.dir_on = 0; z.dmx = 0; z.dmy = 0; z.chain = 0; z.fme_down = 0;     // WelsDiamondCrossSearch
      wh_motion_search (S, P, J, W, C, me16, mvcl, nm, &z);
    } else wh_motion_search (S, P, J, W, C, me16, mvcl, nm);
    p16x = me16.mvx; p16y = me16.mvy;
    cost_luma = me16.satd_cost;
    mb_type = WH_MB_P16x16;
    sad_pred16 = sad_pred;
  }

  WH_PROF_MARK (P, M, 2);   // P16x16 motion search
  // ---- secondary modes (WelsMdInterSecondaryModesEnc) ----
  bool intra = false;
  WhIntraResult ir;
  if (!done && il_intra) {
    // WelsMdSpatialelInterMbIlfmdNoilp, base-layer MB intra (svc_mode_decision.cpp:88-100): no motion search at all -- a skip
    // that is at least as cheap as Intra16x16 stays, anything else becomes intra (I16x16 against I4x4 as in an I slice)
    if (b_skip && cost_luma <= i16c.best_cost) { mb_type = WH_MB_PSKIP; done = true; }
    else { (void)wh_intra_md_enc_p<LOW ? 0 : -1, XWG> (M, P, J, mbx, mby, avail, qp, qpc, 0x7fffffff, &ir, &i16c, stale_cbp); intra = true; done = true; }
  }
  if (!done) {
    // WelsMdFirstIntraMode: I16x16 cost vs the inter/skip cost so far
    if (wh_intra_md_enc_p<LOW ? 0 : -1, XWG> (M, P, J, mbx, mby, avail, qp, qpc, cost_luma, &ir, &i16c, stale_cbp)) { intra = true; done = true; }
  }
  if (!done && b_skip) { mb_type = WH_MB_PSKIP; done = true; }
  WH_PROF_MARK (P, M, 3);   // I16x16 test (+ intra encode when intra wins)
  // the next window guess of this slice: its most recent final 16x16 vector (a race between waves is harmless: any recent vector
  // will do; it changes timing, never a result)
  auto publish_guess = [&] (int fin) {
    if (X.last_mv) {
      WV_LANES_BEGIN (lane)
      if (lane == 0) *X.last_mv = fin;
      WV_LANES_END
    }
  };
  const bool searched = !done;      // the macroblock goes through the partition searches, the refinement and residual coding
  if (done && !intra) publish_guess (mb_type == WH_MB_PSKIP ? wh_pk_mv (skx, sky) : wh_pk_mv (p16x, p16y));     // skip / background / static block

  if (searched) {
    // ---- fine partitions: groups of searches (8x8 x4, 16x8 x2, 8x16 x2), results kept per slot in T ----
    wh_me_store (T, WH_SLOT_16x16, me16);
    WV_LSET (mvcl, 0, il_mv);
    int order0 = -1, order1 = -1, order2 = -1;      // group ids: 0 = 8x8, 1 = 16x8, 2 = 8x16
    bool chain = false;                             // later groups only run when the first one beat the 16x16 cost
    bool merged = false;
    if (SCC) {
      // WelsMdInterFinePartitionVaaOnScreen: unless the pre-processing's 8x8 SADs are flat, four 8x8 searches -- each by the method
      // of its block's static idc -- and, when they win, TryModeMerge: equal vectors side by side or on top of each other make it
      // a 16x8 / 8x16 macroblock whose partitions carry the SUMS of the 8x8 costs
      const int32_t* v8 = (const int32_t*)G.cold_pv;
      const int s8_0 = v8[0], s8_1 = v8[1], s8_2 = v8[2], s8_3 = v8[3];
      const int avg = (s8_0 + s8_1 + s8_2 + s8_3) >> 2;
      const int d0 = (s8_0 >> 6) - (avg >> 6), d1 = (s8_1 >> 6) - (avg >> 6), d2 = (s8_2 >> 6) - (avg >> 6), d3 = (s8_3 >> 6) - (avg >> 6);
      if (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 >= 20) {             // MdInterAnalysisVaaInfo_c != MBVAASIGN_FLAT
        WH_G uint32_t* chn = (WH_G uint32_t*)Z->chain + slice_idc * 4;
        uint32_t down = 0;
        int c = 0;
        for (int i = 0; i < 4; ++i) {
          WhMe m;
          wh_slot_geom (WH_SLOT_8x8 + i, &m.bx, &m.by, &m.bw, &m.bh);
          m.sad_pred = sad_pred16 >> 2;
          wh_slot_pred (K, WH_SLOT_8x8 + i, &m.mvpx, &m.mvpy);
          const int idc = i == 0 ? idc0 : i == 1 ? idc1 : i == 2 ? idc2 : idc3;
          if (idc != 0) WH_STAT (WH_ST_FIXED);
          if (idc == 1) wh_me_fixed (S, P, J, W, C, m, 0, 0);                                          // WelsMotionEstimateSearchStatic
          else if (idc == 2) wh_me_fixed (S, P, J, W, C, m, scroll_on ? smx : 0, scroll_on ? smy : 0); // ...Scrolled (sDirectionalMv)
          else {
            WhSccMe z;
            z.job = Z; z.method = Z->fme ? 2 : 0; z.thr = Z->thr8; z.dir_on = scroll_on; z.dmx = smx; z.dmy = smy; z.fme_down = 0;
            z.chain = !scroll_on ? 0u : !J.dyn_slice ? wh_ld_wg32 (chn + i) : i == 0 ? dch0 : i == 1 ? dch1 : i == 2 ? dch2 : dch3;
            wh_motion_search (S, P, J, W, C, m, mvcl, 1, &z);
            down += z.fme_down;
          }
          if (scroll_on) {           // what the next macroblock of the slice finds in sMe8x8[i].uiSadCost (WH_SEQ_SERIAL: coding order)
            WV_LANES_BEGIN (lane)
            if (lane == 0) wh_st_wg32 (J.dyn_slice ? (WH_G uint32_t*)Z->chain_mb + 4 * xy + i : chn + i, (uint32_t)m.sad_cost);
            WV_LANES_END
          }
          wh_cache_set (K, m.bx >> 2, m.by >> 2, m.bw >> 2, m.bh >> 2, 0, m.mvx, m.mvy);
          wh_me_store (T, WH_SLOT_8x8 + i, m);
          c += m.satd_cost;
        }
        fme_down_mb = down;
        if (down && !J.dyn_slice) {
          WV_LANES_BEGIN (lane)
          if (lane == 0) wh_atomic_add_u32 ((WH_G uint32_t*)Z->fme_cost_down + slice_idc, down);
          WV_LANES_END
        }
        if (c < cost_luma) {
          mb_type = WH_MB_P8x8;
          WH_STAT (WH_ST_P8X8);
          const int v0 = WV_LGET (T.mv, WH_SLOT_8x8), v1 = WV_LGET (T.mv, WH_SLOT_8x8 + 1), v2 = WV_LGET (T.mv, WH_SLOT_8x8 + 2), v3 = WV_LGET (T.mv, WH_SLOT_8x8 + 3);
          const bool same16x8 = v0 == v1 && v2 == v3, same8x16 = v0 == v2 && v1 == v3;
          if (same16x8 != same8x16) {
            merged = true;
            WH_STAT (WH_ST_MERGE);
            mb_type = same16x8 ? WH_MB_P16x8 : WH_MB_P8x16;
            const int first = same16x8 ? WH_SLOT_16x8 : WH_SLOT_8x16, step = same16x8 ? 1 : 2;
            for (int k = 0; k < 2; ++k) {
              const int a = WH_SLOT_8x8 + (same16x8 ? 2 * k : k), b = a + step;
              WhMe m;
              wh_me_fetch (T, a, m);
              m.sad_cost += WV_LGET (T.sad, b); m.satd_cost += WV_LGET (T.satd, b);     // MergeSub16Me: the first block's fields, the costs summed
              wh_me_store (T, first + k, m);
            }
          }
        }
      }
    } else if (!use_satd) {
      // WelsMdInterFinePartitionVaa: partition set chosen from the sign pattern of the four 8x8 SADs
      // between this source MB and the previous source frame (VAACalcSad_c + MdInterAnalysisVaaInfo_c)
      int s8_0, s8_1, s8_2, s8_3;
      if (HOSTIN && J.vaa_sad8x8) { const int32_t* v8 = (const int32_t*)G.cold_pv; s8_0 = v8[0]; s8_1 = v8[1]; s8_2 = v8[2]; s8_3 = v8[3]; }      // the pre-processing's own result
      else {
        int p01, p23;
        WV_SUM2 (p01, p23, lane,
                 (lane < 32 ? (wh_sad4 (* (const uint32_t*)&M.enc_y[lane * 4], G.cold_pv[lane]) << ((lane & 2) ? 16 : 0)) : 0),
                 (lane >= 32 ? (wh_sad4 (* (const uint32_t*)&M.enc_y[lane * 4], G.cold_pv[lane]) << ((lane & 2) ? 16 : 0)) : 0));
        s8_0 = p01 & 0xffff; s8_1 = (int) ((unsigned)p01 >> 16); s8_2 = p23 & 0xffff; s8_3 = (int) ((unsigned)p23 >> 16);
      }
      int sign = 15;
      {
        const int avg = (s8_0 + s8_1 + s8_2 + s8_3) >> 2;
        const int d0 = (s8_0 >> 6) - (avg >> 6), d1 = (s8_1 >> 6) - (avg >> 6), d2 = (s8_2 >> 6) - (avg >> 6), d3 = (s8_3 >> 6) - (avg >> 6);
        if (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 >= 20) sign = ((s8_0 > avg) << 3) | ((s8_1 > avg) << 2) | ((s8_2 > avg) << 1) | (s8_3 > avg);
      }
      if (sign == 3 || sign == 12) order0 = 1;
      else if (sign == 5 || sign == 10) order0 = 2;
      else if (sign == 6 || sign == 9) order0 = 0;
      else if (sign != 15) { order0 = 0; order1 = 1; order2 = 2; chain = true; }
    } else {
      order0 = 0; order1 = 1; order2 = 2; chain = true;     // WelsMdInterFinePartition
    }
    int best_cost = cost_luma;
    for (int oi = 0; oi < 3; ++oi) {
      const int g = oi == 0 ? order0 : oi == 1 ? order1 : order2;
      if (g < 0) break;
      const int first = g == 0 ? WH_SLOT_8x8 : g == 1 ? WH_SLOT_16x8 : WH_SLOT_8x16, cnt = g == 0 ? 4 : 2;
      int c = 0;
      for (int i = 0; i < cnt; ++i) {
        WhMe m;
        wh_slot_geom (first + i, &m.bx, &m.by, &m.bw, &m.bh);
        m.sad_pred = g == 0 ? sad_pred16 >> 2 : sad_pred16 >> 1;
        wh_slot_pred (K, first + i, &m.mvpx, &m.mvpy);
        wh_motion_search (S, P, J, W, C, m, mvcl, 1);
        wh_cache_set (K, m.bx >> 2, m.by >> 2, m.bw >> 2, m.bh >> 2, 0, m.mvx, m.mvy);
        wh_me_store (T, first + i, m);
        c += m.satd_cost;
      }
      if (oi == 0) {
        if (c < best_cost) { best_cost = c; mb_type = g == 0 ? WH_MB_P8x8 : g == 1 ? WH_MB_P16x8 : WH_MB_P8x16; }
        else if (chain) break;
      } else if (c <= best_cost) { best_cost = c; mb_type = g == 1 ? WH_MB_P16x8 : WH_MB_P8x16; }
    }

    WH_PROF_MARK (P, M, 4);   // fine partitions
    // ---- refinement (WelsMdInterMbRefinement) ----
    const int satd_in_md = use_satd;     // bSatdInMdFlag: pfMeCost == pfMdCost == SATD
    int best_sad = 0, best_satd = 0;
    if (mb_type == WH_MB_P8x8) {
      WV_LSET (K.ref, 9, WH_REF_NOT_AVAIL); WV_LSET (K.ref, 21, WH_REF_NOT_AVAIL);
    }
    {
      const int first = mb_type == WH_MB_P16x16 ? WH_SLOT_16x16 : mb_type == WH_MB_P16x8 ? WH_SLOT_16x8 : mb_type == WH_MB_P8x16 ? WH_SLOT_8x16 : WH_SLOT_8x8;
      const int cnt = mb_type == WH_MB_P16x16 ? 1 : mb_type == WH_MB_P8x8 ? 4 : 2;
      for (int i = 0; i < cnt; ++i) {
        WhMe m;
        wh_slot_geom (first + i, &m.bx, &m.by, &m.bw, &m.bh);
        wh_me_fetch (T, first + i, m);
        wh_slot_pred (K, first + i, &m.mvpx, &m.mvpy);
        wh_refine_frac (S, P, J, W, C, m, satd_in_md, merged ? 1 : 0);
        wh_cache_set (K, m.bx >> 2, m.by >> 2, m.bw >> 2, m.bh >> 2, 0, m.mvx, m.mvy);
        {
          const int bx4 = m.bx >> 2, by4 = m.by >> 2, w4 = m.bw >> 2, h4 = m.bh >> 2;
          WV_LANES_BEGIN (lane)
          if (lane < w4 * h4) {
            const int r = (by4 + (w4 == 4 ? lane >> 2 : lane >> 1)) * 4 + bx4 + (lane & (w4 - 1));
            S.mv_out[r][0] = (int16_t)m.mvx; S.mv_out[r][1] = (int16_t)m.mvy; S.mvp_out[r][0] = (int16_t)m.mvpx; S.mvp_out[r][1] = (int16_t)m.mvpy;
          }
          WV_LANES_END
        }
        best_sad += m.sad_cost; best_satd += m.satd_cost;
        wh_mc_chroma_to (S, P, J, W, mbx, mby, m.bx >> 1, m.by >> 1, m.bw >> 1, m.bh >> 1, m.mvx, m.mvy, M.pred_c);
        if (mb_type == WH_MB_P16x16) { me16.mvx = m.mvx; me16.mvy = m.mvy; }
      }
    }
    if (mb_type == WH_MB_P16x16) {
      // iCostSkipMb of a 16x16 MB = SAD of its final prediction (luma + chroma)
      int sl, sc;
      WV_SUM2 (sl, sc, lane, wh_sad4 (* (const uint32_t*)&M.enc_y[lane * 4], * (const uint32_t*)&M.pred_y[lane * 4]),
               (lane < 32 ? wh_sad4 (* (const uint32_t*)&M.enc_c[lane * 4], * (const uint32_t*)&M.pred_c[lane * 4]) : 0));
      cost_skip_mb = sl + sc;
    }
    sad_cost0 = best_sad;
    cost_luma = md_using_sad ? best_sad : best_satd;

    WH_PROF_MARK (P, M, 5);   // fractional refinement + chroma MC
    publish_guess (wh_pk_mv (p16x, p16y));
  }
  // (the prediction is final: nothing below reads a window or the staging area)
  if (searched) {
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
      wh_pred_skip_mv (K, &sx, &sy);
      if (sx == me16.mvx && sy == me16.mvy) { mb_type = WH_MB_PSKIP; skx = sx; sky = sy; }
    }
  }

  WH_PROF_MARK (P, M, 6);   // residual coding
  // ---- store ----
  const bool is_skip = mb_type == WH_MB_PSKIP;
  WH_STAT_WIN (intra ? 10 : (bg_coded || scd_coded) ? 9 : (is_skip && b_skip) ? 8 : is_skip ? 15 : mb_type == WH_MB_P16x16 ? 11 : mb_type == WH_MB_P16x8 ? 12 : mb_type == WH_MB_P8x16 ? 13 : 14);
  WH_G WhMbState* Ms = (WH_G WhMbState*)J.mbs + xy;
  WH_G WhMbRecord* Rs = (WH_G WhMbRecord*)J.records + xy;
  if (intra) {
    WV_LANES_BEGIN (lane)
    if (lane < 16) { wh_st_x<XWG> ((WH_G uint32_t*)&Ms->mv[lane][0], 0u); Rs->mvd[lane][0] = 0; Rs->mvd[lane][1] = 0; }
    if (lane < 2) Rs->mv_tr[lane] = 0;
    if (lane < 4) { wh_st_x<XWG> (&Ms->ref_idx[lane], (int8_t)-1); Rs->ref_idx[lane] = -1; Rs->sub_type[lane] = 0; }
    if (lane == 0) { wh_st_x<XWG> (&Ms->sad_cost[0], 0); wh_st_x<XWG> ((WH_G uint32_t*)&Ms->p16mv[0], (uint32_t)wh_pk_mv (p16x, p16y)); wh_st_x<XWG> (&Ms->skip_sad, 0); if (HOSTIN && J.sad_cost0) (J.sad_cost0_out ? (WH_G int32_t*)J.sad_cost0_out : (WH_G int32_t*)J.sad_cost0)[xy] = 0; }
    WV_LANES_END
    int ibits = 0;
    if (CTRL && J.want_bits) {
      ibits = wh_mb_residual_bits (M, ir.mb_type, ir.cbp, Lm ? Lm->nzc : nullptr, Tm ? Tm->nzc : nullptr) +
              wh_mb_intra_header_bits (M, ir.mb_type, ir.cbp, ir.i16_mode_std, ir.chroma_mode_std, true);
      if (ir.cbp > 0 || ir.mb_type == WH_MB_I16x16) ibits |= WH_BITS_HAS_QP_DELTA;
    }
    wh_store_mb<XWG> (M, P, J, mbx, mby, ir.mb_type, ir.cbp, qp, qpc, ir.i16_mode_std, ir.chroma_mode_std, ir.cost_luma, slice_idc, ibits);
    if (SCC && J.dyn_slice) {
      WV_LANES_BEGIN (lane)
      if (lane == 0) Rs->fme_down = 0;
      WV_LANES_END
    }
    return;
  }
  if (is_skip && b_skip) {
    // decided P_Skip: reconstruction = skip prediction, no residual (WelsRecPskip)
    WV_LANES_BEGIN (lane)
    {
      const int row = lane >> 2, seg = lane & 3;
      * (uint32_t*)&WH_RY (M, seg * 4, row) = * (const uint32_t*)&M.pred_y[row * 16 + seg * 4];
    }
    if (lane < 32) {
      const int pl = lane >> 4, row = (lane >> 1) & 7, half = lane & 1;
      * (uint32_t*)&WH_RC (M, pl, half * 4, row) = * (const uint32_t*)&M.pred_c[pl * 64 + row * 8 + half * 4];
    }
    if (lane < 24) M.nzc[lane] = 0;
    WV_LANES_END
    cbp = 0;
  }
  WV_LANES_BEGIN (lane)
  if (lane < 16) {
    const int mvx = is_skip ? skx : S.mv_out[lane][0], mvy = is_skip ? sky : S.mv_out[lane][1];
    // (one 32-bit store per pair of 16-bit fields: written field by field each pair was two global_store_short)
    const uint32_t mv32 = (uint32_t)wh_pk_mv (mvx, mvy);
    wh_st_x<XWG> ((WH_G uint32_t*)&Ms->mv[lane][0], mv32);
    * (WH_G uint32_t*)&Rs->mvd[lane][0] = is_skip ? 0u : (uint32_t)wh_pk_mv (mvx - S.mvp_out[lane][0], mvy - S.mvp_out[lane][1]);
    if (lane == 3) * (WH_G uint32_t*)&Rs->mv_tr[0] = mv32;
  }
  if (!(cbp & 15) || is_skip) { uint64_t* z = (uint64_t*)M.lv_luma; z[lane] = 0; }
  if (lane == 0) {
    wh_st_x<XWG> ((WH_G uint32_t*)&Ms->ref_idx[0], 0u); * (WH_G uint32_t*)&Rs->ref_idx[0] = 0u; * (WH_G uint32_t*)&Rs->sub_type[0] = 0u;      // four bytes each
    wh_st_x<XWG> (&Ms->sad_cost[0], sad_cost0); wh_st_x<XWG> ((WH_G uint32_t*)&Ms->p16mv[0], (uint32_t)wh_pk_mv (p16x, p16y));
    wh_st_x<XWG> (&Ms->skip_sad, is_skip ? cost_skip_mb : 0);
    if (HOSTIN && J.sad_cost0) (J.sad_cost0_out ? (WH_G int32_t*)J.sad_cost0_out : (WH_G int32_t*)J.sad_cost0)[xy] = sad_cost0;
  }
  WV_LANES_END
  int pbits = 0;
  if (CTRL && J.want_bits && !is_skip) {
    // mb_type (+ four sub_mb_types of P_8x8ref0), the vector differences of the partitions, coded_block_pattern, the residual
    // (svc_set_mb_syn_cavlc.cpp:58-245; one reference picture: no ref_idx)
    const int m16 = mb_type == WH_MB_P16x16, m168 = mb_type == WH_MB_P16x8, m816 = mb_type == WH_MB_P8x16;
    int mvd_bits;
    WV_SUM (mvd_bits, lane, ((lane == 0 || (m168 && lane == 8) || (m816 && lane == 2) || (!m16 && !m168 && !m816 && (lane == 2 || lane == 8 || lane == 10)))
                             ? wh_se_bits_c ((int)S.mv_out[lane & 15][0] - (int)S.mvp_out[lane & 15][0]) + wh_se_bits_c ((int)S.mv_out[lane & 15][1] - (int)S.mvp_out[lane & 15][1]) : 0));
    // (ref_idx_l0 is always 0 here: one bit per partition whenever more than one reference picture is active; P_8x8ref0 carries none)
    const int ref_bits = (J.want_bits & 2) ? (m16 ? 1 : (m168 || m816) ? 2 : 0) : 0;
    pbits = (m16 ? 1 : (m168 || m816) ? 3 : 5 + 4) + ref_bits + mvd_bits + wh_ue_bits (kWhCbpCodeInter[cbp]);
    if (cbp > 0) pbits += wh_mb_residual_bits (M, mb_type, cbp, Lm ? Lm->nzc : nullptr, Tm ? Tm->nzc : nullptr);
    if (cbp > 0) pbits |= WH_BITS_HAS_QP_DELTA;
  }
  wh_store_mb<XWG> (M, P, J, mbx, mby, mb_type, cbp, qp, qpc, 0, 0, cost_luma, slice_idc, pbits);
  if (SCC && J.dyn_slice) {              // (size-limited slices: the host sums the macroblocks the entropy writer really took)
    WV_LANES_BEGIN (lane)
    if (lane == 0) ((WH_G WhMbRecord*)J.records + xy)->fme_down = fme_down_mb;
    WV_LANES_END
  }
  // what the picture keeps for the time it is a reference (WelsMdInterSaveSadAndRefMbType, WelsMdUpdateBGDInfo, both run
  // before the entropy writer): a background skip keeps its own type; pRefMbQp = uiLumaQp unless the MB is an unchanged
  // collocated one (no residual, zero vector, P reference), which inherits the reference's entry.  uiLumaQp at that point is
  // the MB's rate-control QP, except for a skip decided in mode decision (WelsMdInterDecidedPskip, WelsMdBackgroundMbEnc):
  // there it already is the slice's last coded QP -- with a per-MB QP map that is only known in coding order, so the MB
  // leaves a marker that wh_qp_chain_slice resolves.
  const bool decided_skip = is_skip && b_skip;                          // not the P16x16 that WelsMdInterDoubleCheckPskip renames
  if (!bg_coded && !scd_coded) collocated = decided_skip ? (skx == 0 && sky == 0) : ((is_skip || mb_type == WH_MB_P16x16) && cbp == 0 && me16.mvx == 0 && me16.mvy == 0);
  {
    const bool inherit = cbp == 0 && ref_is_p && collocated;
    const bool last_qp = !inherit && decided_skip && CTRL && (J.mb_ctl != nullptr || J.gom_rc != nullptr);
    if (bg_skip || inherit || last_qp) {
      WV_LANES_BEGIN (lane)
      if (lane == 0) {
        if (bg_skip) { wh_st_x<XWG> (&Ms->ref_type, (uint8_t)WH_REFTYPE_BACKGROUND); Rs->bgd_skip = 1; }
        if (inherit) wh_st_x<XWG> (&Ms->ref_qp, (uint8_t)Co->ref_qp);
        else if (last_qp) wh_st_x<XWG> (&Ms->ref_qp, (uint8_t)0xff);                              // WH_REFQP_FROM_CHAIN
      }
      WV_LANES_END
    }
  }
  WH_PROF_MARK (P, M, 7);   // store
}
WH_FN void wh_inter_mb_body (WhInterLds& S, WhInterStage& G, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, WhInterCtx& X) {
  wh_inter_mb_body_t<false> (S, G, P, J, mbx, mby, X);
}

// ---- GOM-level rate control inside the kernel (common/gom_rc.h) ----------------------------------------
// Called for every macroblock after its body.  The last macroblock of a group (in coding order) only starts when every other
// macroblock of the group is complete (its left and top neighbours' chains cover the group's rows), so at this point all records
// of the group are final: walk them in coding order, add the two terms the per-macroblock counts leave out -- ue(mb_skip_run) in
// front of a coded macroblock, se(mb_qp_delta) of one that codes it (WelsSpatialWriteMbSyn, svc_set_mb_syn_cavlc.cpp:260-322) --
// and advance the state exactly like WelsRcMbInfoUpdateGom / WelsRcMbInitGom (ratectl.cpp:1239-1278) would macroblock by
// macroblock.  The macroblocks of the next group wait for this one's done flag and then read the new QP.
WH_FN void wh_gom_close_if_last (const WhSeqParams& P, const WhPicJob& J, int xy) {
  WhGomRc& R = * (WhGomRc*)J.gom_rc;
  const int n = R.n_gom_mb;
  if ((xy + 1) % n != 0 || xy >= R.end_mb) return;          // not the last of its group, or the slice ends with this group
  const int a = xy + 1 - n, b = xy + 1;
  WV_GLOBAL_FENCE();                                       // this wave's own record is among the ones read back
  const int gom_qp = R.calc_qp, p_slice = R.p_slice;
  int skip_run = R.skip_run, last_qp = R.last_qp, bits = 0;
  for (int base = a; base < b; base += 64) {
    WvLaneArr cb, ty;
#ifdef WH_EMU
    memset (&cb, 0, sizeof (cb)); memset (&ty, 0, sizeof (ty));
#else
    cb = 0; ty = 0;
#endif
    WV_LSET_IF (cb, lane, base + lane < b, ((const WH_G WhMbRecord*)J.records + base + lane)->cavlc_bits);
    WV_LSET_IF (ty, lane, base + lane < b, (int) ((const WH_G WhMbRecord*)J.records + base + lane)->mb_type);
    const int cnt = b - base < 64 ? b - base : 64;
    for (int i = 0; i < cnt; ++i) {
      if (WV_LGET (ty, i) == WH_MB_PSKIP) { ++skip_run; continue; }
      const int c = WV_LGET (cb, i);
      bits += c & 0x3fffffff;
      if (p_slice) { bits += wh_ue_bits ((unsigned)skip_run); skip_run = 0; }
      if (c & WH_BITS_HAS_QP_DELTA) { bits += wh_se_bits_c (gom_qp - last_qp); last_qp = gom_qp; }
    }
  }
  wh_gom_next (R, bits);
  R.skip_run = skip_run; R.last_qp = last_qp;
  WV_GLOBAL_FENCE();
}
