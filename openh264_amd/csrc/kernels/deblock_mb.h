// deblock_mb.h -- in-loop deblocking of one macroblock by one wavefront (H.264 8.7).
//
// Reference behaviour restated:
//   codec/encoder/core/src/deblocking.cpp:599-626  DeblockingBSCalc_c (+ :126-230 bS inside / at MB edges)
//   codec/encoder/core/src/deblocking.cpp:357-440  DeblockingInterMb   (edge order, edge QP averaging)
//   codec/encoder/core/src/deblocking.cpp:442-554  DeblockingIntraMb   (bS 4 on MB edges, 3 inside)
//   codec/common/src/deblocking_common.cpp:5-181   DeblockLuma{Lt4,Eq4}_c, DeblockChroma{Lt4,Eq4}_c
// MBs are processed on the same 2:1 diagonals as mode decision: MB (x,y) filters its left edge into
// MB (x-1,y) and its top edge into MB (x,y-1), after (x+1,y-1) has finished its own left edge.
#pragma once
#include "prims.h"

// LDS tile: luma pixel (x,y), x,y in [-4,15] at y[(y+4)*24 + x+4]; chroma (x,y), x in [-4,7], y in [-2,7] -> c[p][(y+2)*12 + x+4]
typedef struct WhDbLds {
  uint8_t y[20 * 24];
  uint8_t c[2][10 * 12];
  uint8_t bs[2][4][4];       // [dir 0=vertical edges,1=horizontal][edge][segment]
  uint32_t st[3 * 36];       // WhMbState copies (36 words each): this MB, left, top
#if defined(WH_PROF)
  uint32_t prof[32];         // phase-profiling accumulators of this wave (WH_PROF_MARK; profiling build only)
#endif
} WhDbLds;
#define WH_DY(S, x, yy) ((S).y[((yy) + 4) * 24 + (x) + 4])
#define WH_DC(S, p, x, yy) ((S).c[p][((yy) + 2) * 12 + (x) + 4])

// one line of a luma edge: pix points at q0, step = distance between p/q samples
WH_FN void wh_db_luma_line (uint8_t* q, int step, int bs, int alpha, int beta, int idx_a) {
  if (bs == 0) return;
  const int p0 = q[-step], p1 = q[-2 * step], p2 = q[-3 * step];
  const int q0 = q[0], q1 = q[step], q2 = q[2 * step];
  const int d = wh_abs (p0 - q0);
  if (!(d < alpha && wh_abs (p1 - p0) < beta && wh_abs (q1 - q0) < beta)) return;
  const bool ap = wh_abs (p2 - p0) < beta, aq = wh_abs (q2 - q0) < beta;
  if (bs < 4) {
    const int tc0 = kWhTc0[idx_a * 3 + bs - 1];
    int tc = tc0;
    if (ap) { q[-2 * step] = (uint8_t) (p1 + wh_clip3 ((p2 + ((p0 + q0 + 1) >> 1) - (p1 * 2)) >> 1, -tc0, tc0)); tc++; }
    if (aq) { q[step] = (uint8_t) (q1 + wh_clip3 ((q2 + ((p0 + q0 + 1) >> 1) - (q1 * 2)) >> 1, -tc0, tc0)); tc++; }
    const int delta = wh_clip3 ((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    q[-step] = wh_clip255 (p0 + delta);
    q[0] = wh_clip255 (q0 - delta);
  } else {
    if (d < ((alpha >> 2) + 2)) {
      if (ap) {
        const int p3 = q[-4 * step];
        q[-step] = (uint8_t) ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        q[-2 * step] = (uint8_t) ((p2 + p1 + p0 + q0 + 2) >> 2);
        q[-3 * step] = (uint8_t) ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      } else {
        q[-step] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
      }
      if (aq) {
        const int q3 = q[3 * step];
        q[0] = (uint8_t) ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        q[step] = (uint8_t) ((p0 + q0 + q1 + q2 + 2) >> 2);
        q[2 * step] = (uint8_t) ((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
      } else {
        q[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
      }
    } else {
      q[-step] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
      q[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
    }
  }
}
WH_FN void wh_db_chroma_line (uint8_t* q, int step, int bs, int alpha, int beta, int idx_a) {
  if (bs == 0) return;
  const int p0 = q[-step], p1 = q[-2 * step], q0 = q[0], q1 = q[step];
  if (!(wh_abs (p0 - q0) < alpha && wh_abs (p1 - p0) < beta && wh_abs (q1 - q0) < beta)) return;
  if (bs < 4) {
    const int tc = kWhTc0[idx_a * 3 + bs - 1] + 1;
    const int delta = wh_clip3 ((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    q[-step] = wh_clip255 (p0 + delta);
    q[0] = wh_clip255 (q0 - delta);
  } else {
    q[-step] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
    q[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
  }
}

// ---- the same filters on samples held in registers ------------------------------------------------------------------
// tc0 comes from one wave-uniform 32-bit load (kWhTc0Packed) instead of a per-lane byte gather inside the dependent chain; the
// bS < 4 filter is select-only; the bS 4 arithmetic is skipped for the whole edge unless some line needs it (`any4`,
// wave-uniform -- only intra MBs have bS 4).  Bit-exact with wh_db_luma_line / wh_db_chroma_line
// (tests/test_frame_parity.py::test_emu_deblock_per_edge).
// `chroma`: the line is a chroma line -- the same filter with the p1 / q1 updates and the strong variant switched off and tc = tc0 + 1
// (DeblockChromaLt4_c / DeblockChromaEq4_c are exactly that subset), so luma and chroma lines share one instruction stream
WH_FN void wh_db_line_px (bool chroma, int bs, int alpha, int beta, int tc3, bool any4, int p3, int& p2, int& p1, int& p0, int& q0, int& q1, int& q2, int q3) {
  const int d = wh_absdiff_px (p0, q0);
  const bool on = bs != 0 && d < alpha && wh_absdiff_px (p1, p0) < beta && wh_absdiff_px (q1, q0) < beta;
  const bool ap = !chroma && wh_absdiff_px (p2, p0) < beta, aq = !chroma && wh_absdiff_px (q2, q0) < beta;
  const int bsn = bs < 1 ? 1 : bs > 3 ? 3 : bs;
  const int tc0 = (tc3 >> ((bsn - 1) * 8)) & 255;
  const int tc = tc0 + (chroma ? 1 : (ap ? 1 : 0) + (aq ? 1 : 0));
  const int avg = (p0 + q0 + 1) >> 1;
  int rp2 = p2, rq2 = q2;
  int rp1 = ap ? p1 + wh_clip3 ((p2 + avg - (p1 * 2)) >> 1, -tc0, tc0) : p1;
  int rq1 = aq ? q1 + wh_clip3 ((q2 + avg - (q1 * 2)) >> 1, -tc0, tc0) : q1;
  const int delta = wh_clip3 ((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
  int rp0 = wh_clip255 (p0 + delta), rq0 = wh_clip255 (q0 - delta);
  if (any4) {
    const bool s4 = bs == 4, strong = !chroma && d < ((alpha >> 2) + 2);
    const int wp0 = (2 * p1 + p0 + q1 + 2) >> 2, wq0 = (2 * q1 + q0 + p1 + 2) >> 2;
    const bool sp = strong && ap, sq = strong && aq;
    const int s_p0 = sp ? (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3 : wp0;
    const int s_p1 = sp ? (p2 + p1 + p0 + q0 + 2) >> 2 : p1;
    const int s_p2 = sp ? (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3 : p2;
    const int s_q0 = sq ? (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3 : wq0;
    const int s_q1 = sq ? (p0 + q0 + q1 + q2 + 2) >> 2 : q1;
    const int s_q2 = sq ? (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3 : q2;
    rp0 = s4 ? s_p0 : rp0; rp1 = s4 ? s_p1 : rp1; rp2 = s4 ? s_p2 : rp2;
    rq0 = s4 ? s_q0 : rq0; rq1 = s4 ? s_q1 : rq1; rq2 = s4 ? s_q2 : rq2;
  }
  if (on) { p2 = rp2; p1 = rp1; p0 = rp0; q0 = rq0; q1 = rq1; q2 = rq2; }
}

// (the two filters by their old names: hip/leaf.hip's per-edge slots)
WH_FN void wh_db_luma_px (int bs, int alpha, int beta, int tc3, bool any4, int p3, int& p2, int& p1, int& p0, int& q0, int& q1, int& q2, int q3) {
  wh_db_line_px (false, bs, alpha, beta, tc3, any4, p3, p2, p1, p0, q0, q1, q2, q3);
}
WH_FN void wh_db_chroma_px (int bs, int alpha, int beta, int tc3, int p1, int& p0, int& q0, int q1) {
  int p2 = 0, q2 = 0, a = p1, b = q1;
  wh_db_line_px (true, bs, alpha, beta, tc3, bs == 4, 0, p2, a, p0, q0, b, q2, 0);
}

// byte offset of the 16-byte row piece of the picture's tiled twin (common/wh_types.h WH_TILE_*) that holds luma sample (x, y) / chroma sample (x, y),
// in 32-bit arithmetic (a tiled picture is far below 4 GB)
WH_FN uint32_t wh_db_tile_y_off (int stride_y, int x, int y) {
  return ((wh_mul_u24 ((uint32_t) ((y + 32) >> 3), (uint32_t) (stride_y >> 4)) + (uint32_t) ((x + 32) >> 4)) << 7) + (uint32_t) (((y + 32) & 7) << 4);
}
WH_FN uint32_t wh_db_tile_c_off (int stride_c, int x, int y) {
  return ((wh_mul_u24 ((uint32_t) ((y + 16) >> 3), (uint32_t) (stride_c >> 3)) + (uint32_t) ((x + 16) >> 3)) << 7) + (uint32_t) (((y + 16) & 7) << 4);
}

WH_FN bool wh_mv_far (const int16_t* a, const int16_t* b) {
  return wh_abs (a[0] - b[0]) >= 4 || wh_abs (a[1] - b[1]) >= 4;
}

// Staging area of a wave for its NEXT macroblock (LDS-DMA, a separate LDS object like WhInterStage): what no wave of this
// kernel writes before that MB is filtered -- the three MB states and the MB's own 16x16 / 8x8 samples.
typedef struct alignas (16) WhDbStage { uint32_t st[128]; uint32_t y[64]; uint32_t c[32]; } WhDbStage;

WH_FN void wh_deblock_cold_fetch (WhDbStage& G, int lane, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  const int w = P.mb_w, xy = mby * w + mbx;
  const WH_G WhMbState* Mg = (const WH_G WhMbState*)J.mbs + xy;
  for (int k = 0; k < 2; ++k) {
    const int i = lane + 64 * k, n = i / 36, wd = i - n * 36;       // state n: this MB, left, top
    if (i < 108 && (n == 0 || (n == 1 ? mbx > 0 : mby > 0))) wh_ld_async4 ((const WH_G uint32_t*) (Mg - (n == 1 ? 1 : n == 2 ? w : 0)) + wd, &G.st[64 * k], lane);
  }
  if (J.rec_blk) {               // the unfiltered samples as mode decision left them: the macroblock's own three lines (WhPicJob::rec_blk), in the staging area's order
    const WH_G uint8_t* b = (const WH_G uint8_t*)J.rec_blk + (size_t)xy * WH_SRC_MB_BYTES;
    wh_ld_async4 (b + lane * 4, G.y, lane);
    if (lane < 32) wh_ld_async4 (b + 256 + lane * 4, G.c, lane);
    return;
  }
  wh_ld_async4 ((const WH_G uint8_t*)J.rec[0] + (ptrdiff_t) (mby * 16 + (lane >> 2)) * P.rec_stride_y + mbx * 16 + (lane & 3) * 4, G.y, lane);
  if (lane < 32) {
    const int pl = lane >> 4, row = (lane >> 1) & 7, half = lane & 1;
    wh_ld_async4 ((const WH_G uint8_t*)J.rec[1 + pl] + (ptrdiff_t) (mby * 8 + row) * P.rec_stride_c + mbx * 8 + half * 4, G.c, lane);
  }
}

// this MB's bottom rows / right columns for the MBs below / to the right; with `left_mod` also the three columns of the
// left MB that the left-edge filter may have changed, inside that MB's bottom rows (they are the top strip of the MB
// below it)
WH_FN void wh_db_publish (WhDbLds& S, uint32_t* etop, uint32_t* eleft, bool left_mod) {
  WV_LANES_BEGIN (lane)
  if (lane < 16) etop[lane] = * (const uint32_t*)&S.y[((lane >> 2) + 12 + 4) * 24 + (lane & 3) * 4 + 4];
  else if (lane < 32) eleft[lane - 16] = * (const uint32_t*)&S.y[(lane - 16 + 4) * 24 + 12 + 4];
  else if (lane < 40) { const int k = lane - 32; etop[16 + k] = * (const uint32_t*)&S.c[k >> 2][(((k >> 1) & 1) + 6 + 2) * 12 + (k & 1) * 4 + 4]; }
  else if (lane < 56) { const int k = lane - 40; eleft[16 + k] = * (const uint32_t*)&S.c[k >> 3][((k & 7) + 2) * 12 + 4 + 4]; }
  else if (left_mod) {
    const int k = lane - 56;                    // 0..3: luma rows 12..15, 4..7: chroma rows 6..7 of both planes
    if (k < 4) (etop - 24)[k * 4 + 3] = * (const uint32_t*)&S.y[(k + 12 + 4) * 24];
    else { const int c = k - 4; (etop - 24)[16 + c * 2 + 1] = * (const uint32_t*)&S.c[c >> 1][((c & 1) + 6 + 2) * 12]; }
  }
  WV_LANES_END
}

// ---- exchange of border strips between the wavefronts of a workgroup ------------------------------------------
// The four rows / columns of samples that a MB shares with the MB below / to its right are handed over through LDS
// instead of through the picture in HBM: a sample just written by a neighbour wave would otherwise be read back through
// a partially written cache line (an HBM fill) on the critical path of every macroblock, and the producer would have to
// drain its stores before it may flag completion.  One entry per MB column holds rows 12..15 (luma) / 6..7 (chroma) of
// the most recently filtered MB of that column, one entry per MB row its columns 12..15 / 4..7.
//   top[col]  : 24 words = luma r(4) x w(4), then per plane r(2) x w(2)
//   left[row] : 32 words = luma row(16), then per plane row(8)
typedef struct WhDbXchg { uint32_t* top; uint32_t* left; int first_row; } WhDbXchg;
WH_HDFN size_t wh_db_xchg_words (int mb_w, int rows) { return (size_t)mb_w * 24 + (size_t)rows * 32; }

// Boundary strength of segment s = l & 3 of edge e = (l >> 2) & 3 in direction dir = l >> 4 (0: vertical edges, 1: horizontal), l = 0 .. 31
// (deblocking.cpp:126-230, 599-626).  M / Nl / Nt: the states of the macroblock, its left and its upper neighbour.
WH_FN int wh_db_bs_lane (const WhMbState* M, const WhMbState* Nl, const WhMbState* Nt, bool left_ok, bool top_ok, bool intra, int type, int l) {
  const int dir = l >> 4, e = (l >> 2) & 3, s = l & 3;
  // block on the q side (inside this MB) and on the p side, raster 4x4 indices
  const int bq = dir == 0 ? s * 4 + e : e * 4 + s;
  int bs = 0;
  if (e == 0) {
    const bool ok = dir == 0 ? left_ok : top_ok;
    if (ok) {
      const WhMbState* N = dir == 0 ? Nl : Nt;
      const int bp = dir == 0 ? s * 4 + 3 : 12 + s;
      if (intra || WH_IS_INTRA (N->mb_type)) bs = 4;
      else if (M->nzc[bq] | N->nzc[bp]) bs = 2;
      else bs = (M->ref_idx[(bq >> 3) * 2 + ((bq & 3) >> 1)] != N->ref_idx[(bp >> 3) * 2 + ((bp & 3) >> 1)]) || wh_mv_far (M->mv[bq], N->mv[bp]);
    }
  } else if (intra) {
    bs = 3;
  } else if (type != WH_MB_PSKIP) {
    const int bp = dir == 0 ? bq - 1 : bq - 4;
    if (M->nzc[bq] | M->nzc[bp]) bs = 2;
    else if (type != WH_MB_P16x16) bs = wh_mv_far (M->mv[bq], M->mv[bp]);
  }
  return bs;
}

// ---- one direction's edges of one macroblock -------------------------------------------------------------------------
// What they are filtered with (uniform per macroblock): index 0 = the macroblock's outer edge (edge QP averaged with the neighbour's,
// deblocking.cpp:357-440), 1 = its inner edges; c = chroma.  bsw<e> = the four boundary strengths of edge e as one word.
typedef struct WhDbDir {
  int alpha0, beta0, tc30, alpha1, beta1, tc31, alphac0, betac0, tc3c0, alphac1, betac1, tc3c1;
  uint32_t bsw0, bsw1, bsw2, bsw3;
  bool act_l0, act_l1, act_l2, act_l3, act_c0, act_c1;      // edge e has luma / chroma lines to filter (the chroma line's second edge is the macroblock's edge 2)
} WhDbDir;
// (wave-uniform) on<e>: some line has edge e to filter; any4<e>: some line of edge e has bS 4
typedef struct WhDbOn { bool on0, on1, on2, on3, any40, any41, any42, any43; } WhDbOn;
WH_FN WhDbDir wh_db_dir_params (const WhDbLds& S, const WhSeqParams& P, int dir, bool outer_ok, const WhMbState* N, int qp, int qpc) {
  WhDbDir D;
  const int eq0 = (qp + (int)N->luma_qp + 1) >> 1, eqc0 = (qpc + (int)N->chroma_qp + 1) >> 1;      // only used when outer_ok
  const int ia0 = wh_clip3 (eq0 + P.alpha_offset, 0, 51), ib0 = wh_clip3 (eq0 + P.beta_offset, 0, 51);
  const int ia1 = wh_clip3 (qp + P.alpha_offset, 0, 51), ib1 = wh_clip3 (qp + P.beta_offset, 0, 51);
  const int iac0 = wh_clip3 (eqc0 + P.alpha_offset, 0, 51), ibc0 = wh_clip3 (eqc0 + P.beta_offset, 0, 51);
  const int iac1 = wh_clip3 (qpc + P.alpha_offset, 0, 51), ibc1 = wh_clip3 (qpc + P.beta_offset, 0, 51);
  D.alpha0 = kWhAlpha[ia0]; D.beta0 = kWhBeta[ib0]; D.tc30 = kWhTc0Packed[ia0];
  D.alpha1 = kWhAlpha[ia1]; D.beta1 = kWhBeta[ib1]; D.tc31 = kWhTc0Packed[ia1];
  D.alphac0 = kWhAlpha[iac0]; D.betac0 = kWhBeta[ibc0]; D.tc3c0 = kWhTc0Packed[iac0];
  D.alphac1 = kWhAlpha[iac1]; D.betac1 = kWhBeta[ibc1]; D.tc3c1 = kWhTc0Packed[iac1];
  D.bsw0 = outer_ok ? * (const uint32_t*)&S.bs[dir][0][0] : 0u; D.bsw1 = * (const uint32_t*)&S.bs[dir][1][0];
  D.bsw2 = * (const uint32_t*)&S.bs[dir][2][0]; D.bsw3 = * (const uint32_t*)&S.bs[dir][3][0];
  D.act_l0 = D.bsw0 != 0 && (D.alpha0 | D.beta0) != 0; D.act_l1 = D.bsw1 != 0 && (D.alpha1 | D.beta1) != 0;
  D.act_l2 = D.bsw2 != 0 && (D.alpha1 | D.beta1) != 0; D.act_l3 = D.bsw3 != 0 && (D.alpha1 | D.beta1) != 0;
  D.act_c0 = D.bsw0 != 0 && (D.alphac0 | D.betac0) != 0; D.act_c1 = D.bsw2 != 0 && (D.alphac1 | D.betac1) != 0;
  return D;
}
WH_FN WhDbOn wh_db_dir_on (const WhDbDir& D) {
  WhDbOn O;
  O.on0 = D.act_l0 || D.act_c0; O.on1 = D.act_l1 || D.act_c1; O.on2 = D.act_l2; O.on3 = D.act_l3;
  O.any40 = (((D.act_l0 ? D.bsw0 : 0u) | (D.act_c0 ? D.bsw0 : 0u)) & 0x04040404u) != 0;
  O.any41 = (((D.act_l1 ? D.bsw1 : 0u) | (D.act_c1 ? D.bsw2 : 0u)) & 0x04040404u) != 0;
  O.any42 = ((D.act_l2 ? D.bsw2 : 0u) & 0x04040404u) != 0;
  O.any43 = ((D.act_l3 ? D.bsw3 : 0u) & 0x04040404u) != 0;
  return O;
}
// One pass per direction: line l owns one line of the tile (luma row / column, l = 0 .. 15; chroma line of one plane, l = 16 .. 31), loads it
// once, filters its four (two) edges one after the other in registers and writes it back: the eight strictly ordered luma edges of a MB cost
// two LDS round trips instead of eight (measured: edge filters 7.6 k -> 5.7 k cycles per MB, the pass 3.15 -> 2.79 ms for 128 1080p pictures).
// A luma line has 20 samples (-4 .. 15); a chroma line's samples -2 .. 7 sit at indices 2 .. 11, so that its two edges fall on the slots of
// the first two luma edges: between [3] | [4] and [7] | [8].  One instruction stream for both (round 5; the chroma lines used to be a second
// branch of their own: a third of the pass's vector instructions).  A chroma line reads a few bytes around it that are not its own (inside
// the tile) and never writes them back.  `D`: the line's macroblock's; `O`: wave-uniform (over all the macroblocks the wave filters at once).
WH_FN void wh_db_filter_line (WhDbLds& S, int l, int dir, const WhDbDir& D, const WhDbOn& O) {
  const bool ch = l >= 16;
  const int pl = (l - 16) >> 3, k = l & 7;
  uint8_t* const Sb = (uint8_t*)&S;
  const int ybase = (int) ((uint8_t*)&S.y[0] - Sb), cbase = (int) ((uint8_t*)&S.c[0][0] - Sb) + pl * 120;
  int px[20];
  if (dir == 0) {
    const uint32_t* wr = (const uint32_t*) (Sb + (ch ? cbase + (k + 2) * 12 : ybase + (l + 4) * 24));
#pragma unroll
    for (int j = 0; j < 5; ++j) { const uint32_t v = wr[j]; px[4 * j] = (int) (v & 255u); px[4 * j + 1] = (int) ((v >> 8) & 255u); px[4 * j + 2] = (int) ((v >> 16) & 255u); px[4 * j + 3] = (int) (v >> 24); }
  } else {
    const uint8_t* col = Sb + (ch ? cbase + k + 4 - 2 * 12 : ybase + l + 4);
    const int st = ch ? 12 : 24;
#pragma unroll
    for (int r = 0; r < 20; ++r) px[r] = col[r * st];
  }
  const int sh = ch ? 8 * (k >> 1) : 8 * (l >> 2);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (!(e == 0 ? O.on0 : e == 1 ? O.on1 : e == 2 ? O.on2 : O.on3)) continue;
    const uint32_t bl = e == 0 ? D.bsw0 : e == 1 ? D.bsw1 : e == 2 ? D.bsw2 : D.bsw3;
    const bool al_on = e == 0 ? D.act_l0 : e == 1 ? D.act_l1 : e == 2 ? D.act_l2 : D.act_l3;
    const bool ac_on = e == 0 ? D.act_c0 : e == 1 ? D.act_c1 : false;
    const uint32_t bc = e == 0 ? D.bsw0 : D.bsw2;                     // the chroma line's second edge is the macroblock's edge 2
    const uint32_t bw = ch ? (ac_on ? bc : 0u) : (al_on ? bl : 0u);
    const int al = ch ? (e == 0 ? D.alphac0 : D.alphac1) : (e == 0 ? D.alpha0 : D.alpha1), be = ch ? (e == 0 ? D.betac0 : D.betac1) : (e == 0 ? D.beta0 : D.beta1);
    const int tc = ch ? (e == 0 ? D.tc3c0 : D.tc3c1) : (e == 0 ? D.tc30 : D.tc31);
    const bool any4 = e == 0 ? O.any40 : e == 1 ? O.any41 : e == 2 ? O.any42 : O.any43;
    wh_db_line_px (ch, (int) ((bw >> sh) & 255u), al, be, tc, any4, px[4 * e], px[4 * e + 1], px[4 * e + 2], px[4 * e + 3], px[4 * e + 4], px[4 * e + 5], px[4 * e + 6], px[4 * e + 7]);
  }
  if (dir == 0) {
    uint32_t* ww = (uint32_t*) (Sb + (ch ? cbase + (k + 2) * 12 : ybase + (l + 4) * 24));
#pragma unroll
    for (int j = 0; j < 5; ++j) if (j < 3 || !ch) ww[j] = (uint32_t)px[4 * j] | ((uint32_t)px[4 * j + 1] << 8) | ((uint32_t)px[4 * j + 2] << 16) | ((uint32_t)px[4 * j + 3] << 24);
  } else {
    uint8_t* col = Sb + (ch ? cbase + k + 4 - 2 * 12 : ybase + l + 4);
    const int st = ch ? 12 : 24;
#pragma unroll
    for (int r = 1; r < 19; ++r) if (!ch || (r >= 3 && r <= 8)) col[r * st] = (uint8_t)px[r];      // (chroma: rows -1 .. 4 hold everything its two edges can change)
  }
}

// Write-back of a macroblock in the middle of its slice (see wh_deblock_mb_body): line l = 0 .. 15 stores one 16-sample luma row (rows -4 .. -1
// from x = 0, rows 0 .. 11 from x = -4), l = 16 .. 31 one 8-sample row of a chroma plane (rows -2 .. -1 from x = 0, rows 0 .. 5 from x = -4), to the
// planar picture and -- `tiles` -- to its tiled twin.
WH_FN void wh_db_store_interior_line (const WhDbLds& S, int l, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, bool tiles) {
  if (l < 16) {
    const int row = l - 4, x0 = l < 4 ? 0 : -4;
    const uint32_t* sp = (const uint32_t*)&WH_DY (S, x0, row);
    const uint32_t w0 = sp[0], w1 = sp[1], w2 = sp[2], w3 = sp[3];
    WH_G uint8_t* d = (WH_G uint8_t*)J.rec[0] + (ptrdiff_t) (mby * 16 + row) * P.rec_stride_y + mbx * 16 + x0;
    wh_stg16_a4 (d, w0, w1, w2, w3);
    if (tiles) {
      const uint32_t tb = wh_db_tile_y_off (P.rec_stride_y, mbx * 16, mby * 16 + row);       // this macroblock's tile column
      WH_G uint8_t* t = (WH_G uint8_t*)J.rec_tiles[0];
      * (WH_G uint32_t*) (t + (x0 == 0 ? tb : tb - 128u + 12u)) = w0;                         // (the left tile column is the previous tile of the row)
      wh_stg12_a4 (t + (x0 == 0 ? tb + 4u : tb), w1, w2, w3);
    }
  } else {
    const int pl = (l - 16) >> 3, k = l & 7, row = k - 2, x0 = k < 2 ? 0 : -4;
    const uint32_t* sp = (const uint32_t*)&WH_DC (S, pl, x0, row);
    const uint32_t w0 = sp[0], w1 = sp[1];
    WH_G uint8_t* d = (WH_G uint8_t*)J.rec[1 + pl] + (ptrdiff_t) (mby * 8 + row) * P.rec_stride_c + mbx * 8 + x0;
    wh_stg8_a4 (d, w0, w1);
    if (tiles) {
      const uint32_t tb = wh_db_tile_c_off (P.rec_stride_c, mbx * 8, mby * 8 + row) + (uint32_t)pl * 8u;
      WH_G uint8_t* t = (WH_G uint8_t*)J.rec_tiles[1];
      * (WH_G uint32_t*) (t + (x0 == 0 ? tb : tb - 128u + 4u)) = w0;
      * (WH_G uint32_t*) (t + (x0 == 0 ? tb + 4u : tb)) = w1;
    }
  }
}

// The neighbours' strips (only now final: the caller has waited for them) and the staged inputs (`G`, landed) into the tile `S`.
// top_lds / left_lds: the strip comes from the exchange buffers (the neighbour is a macroblock of this workgroup), else from the picture.
WH_FN void wh_db_assemble (WhDbLds& S, const WhDbStage& G, const uint32_t* etop, const uint32_t* eleft, bool top_lds, bool left_lds,
                           const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  WV_LANES_BEGIN (lane)
  {
    uint32_t v = 0;
    // lanes 0..15: luma rows -4..-1, words x = 0..12; 16..31: luma rows 0..15, word x = -4;
    // 32..39: chroma rows -2..-1, words x = 0,4 per plane; 40..55: chroma rows 0..7, word x = -4 per plane
    if (lane < 16) {
      const int r = lane >> 2, wd = lane & 3;
      if (top_lds) v = etop[lane];
      else if (mby > 0) v = wh_ld_xwg32 ((const WH_G uint32_t*) ((const WH_G uint8_t*)J.rec[0] + (ptrdiff_t) (mby * 16 + r - 4) * P.rec_stride_y + mbx * 16 + wd * 4));
    } else if (lane < 32) {
      const int row = lane - 16;
      if (left_lds) v = eleft[row];
      else if (mbx > 0) v = wh_ld_xwg32 ((const WH_G uint32_t*) ((const WH_G uint8_t*)J.rec[0] + (ptrdiff_t) (mby * 16 + row) * P.rec_stride_y + mbx * 16 - 4));
    } else if (lane < 40) {
      const int k = lane - 32, pl = k >> 2, r = (k >> 1) & 1, wd = k & 1;
      if (top_lds) v = etop[16 + k];
      else if (mby > 0) v = wh_ld_xwg32 ((const WH_G uint32_t*) ((const WH_G uint8_t*)J.rec[1 + pl] + (ptrdiff_t) (mby * 8 + r - 2) * P.rec_stride_c + mbx * 8 + wd * 4));
    } else if (lane < 56) {
      const int k = lane - 40, pl = k >> 3, row = k & 7;
      if (left_lds) v = eleft[16 + k];
      else if (mbx > 0) v = wh_ld_xwg32 ((const WH_G uint32_t*) ((const WH_G uint8_t*)J.rec[1 + pl] + (ptrdiff_t) (mby * 8 + row) * P.rec_stride_c + mbx * 8 - 4));
    }
    S.st[lane] = G.st[lane];
    if (lane < 44) S.st[64 + lane] = G.st[64 + lane];
    * (uint32_t*)&S.y[((lane >> 2) + 4) * 24 + (lane & 3) * 4 + 4] = G.y[lane];
    if (lane < 32) { const int pl = lane >> 4, row = (lane >> 1) & 7, half = lane & 1; * (uint32_t*)&S.c[pl][(row + 2) * 12 + half * 4 + 4] = G.c[lane]; }
    if (lane < 16) * (uint32_t*)&S.y[(lane >> 2) * 24 + (lane & 3) * 4 + 4] = v;
    else if (lane < 32) * (uint32_t*)&S.y[(lane - 16 + 4) * 24] = v;
    else if (lane < 40) { const int k = lane - 32; * (uint32_t*)&S.c[k >> 2][((k >> 1) & 1) * 12 + (k & 1) * 4 + 4] = v; }
    else if (lane < 56) { const int k = lane - 40; * (uint32_t*)&S.c[k >> 3][((k & 7) + 2) * 12] = v; }
  }
  WV_LANES_END
}

// `G` holds this MB's staged inputs (wh_deblock_cold_fetch, landed); when next_valid the staging area is refilled for
// (next_mbx, next_mby) as soon as it has been emptied.  [first, last) = MB addresses of the workgroup's slice: neighbours
// inside it exchange strips through `E`, the others (another workgroup) through the picture in HBM.
//
// Every sample of the picture is written by exactly ONE macroblock of the slice, so a wave never has to wait for its
// stores before it flags completion: of an MB's 16x16 samples the right four columns belong to the right neighbour and
// the bottom four rows to the MB below (the corner to the latter) whenever that neighbour is inside the slice -- it
// receives them through `E`, possibly filters them, and writes them.  Without such a neighbour the MB writes them itself.
// Returns true when the caller must drain this wave's stores before it flags the MB done inside the workgroup (the MB
// rewrote samples of another slice's MBs that a later MB of this workgroup reads back from the picture).
// `xwg`: a later band (another workgroup) reads samples this MB writes: its stores go through to memory (wh_st_xwg32); strips
// that come from the picture instead of `E` were written that way by the band above and are loaded past the caches.
WH_FN bool wh_deblock_mb_body (WhDbLds& S, WhDbStage& G, const WhDbXchg& E, int first, int last, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby,
                               int next_valid, int next_mbx, int next_mby, bool xwg) {
  WH_PROF_DECL (P);
  const int w = P.mb_w, xy = mby * w + mbx;
  const bool top_lds = mby > 0 && xy - w >= first, left_lds = mbx > 0 && xy - 1 >= first;
  uint32_t* etop = E.top + mbx * 24;
  uint32_t* eleft = E.left + (mby - E.first_row) * 32;
  wh_db_assemble (S, G, etop, eleft, top_lds, left_lds, P, J, mbx, mby);
  if (next_valid) {
    WV_LANES_BEGIN (lane)
    wh_deblock_cold_fetch (G, lane, P, J, next_mbx, next_mby);
    WV_LANES_END
  }

  const WhMbState* M = (const WhMbState*)&S.st[0];
  const WhMbState* Nl = (const WhMbState*)&S.st[36];
  const WhMbState* Nt = (const WhMbState*)&S.st[72];
  const int fidc = (P.deblock_idc != 0);        // 1: do not filter across slice boundaries
  const bool left_ok = mbx > 0 && (!fidc || M->slice_idc == Nl->slice_idc);
  const bool top_ok = mby > 0 && (!fidc || M->slice_idc == Nt->slice_idc);
  const int type = M->mb_type;
  const bool intra = WH_IS_INTRA (type);

  WH_PROF_MARK (P, S, 5);   // neighbour strips + staged inputs assembled in the tile
  // ---- boundary strengths ----
  int any_bs;
  WV_LANES_BEGIN (lane)
  if (lane < 32) S.bs[lane >> 4][(lane >> 2) & 3][lane & 3] = (uint8_t)wh_db_bs_lane (M, Nl, Nt, left_ok, top_ok, intra, type, lane);
  WV_LANES_END
  WV_ANY (any_bs, lane, (lane < 32 && S.bs[lane >> 4][(lane >> 2) & 3][lane & 3] != 0));
  const bool filtered = any_bs != 0;           // nothing to filter: the MB's own samples stay as mode decision left them
  if (filtered) {
  const int qp = M->luma_qp, qpc = M->chroma_qp;
  WH_PROF_MARK (P, S, 6);   // boundary strengths
#if !defined(WH_DB_PER_EDGE)
  // ---- one pass per direction: a lane owns one line (luma row / column, lanes 0..15; chroma line of one plane, lanes
  //      16..31), loads it once, filters its four (two) edges one after the other in registers and writes it back: the
  //      eight strictly ordered luma edges of a MB cost two LDS round trips instead of eight (measured: edge filters 7.6 k ->
  //      5.7 k cycles per MB, the pass 3.15 -> 2.79 ms for 128 1080p pictures) ----
  for (int dir = 0; dir < 2; ++dir) {
    const WhDbDir D = wh_db_dir_params (S, P, dir, dir == 0 ? left_ok : top_ok, dir == 0 ? Nl : Nt, qp, qpc);
    if ((D.bsw0 | D.bsw1 | D.bsw2 | D.bsw3) == 0) continue;
    const WhDbOn O = wh_db_dir_on (D);
    WV_LANES_BEGIN (lane)
    if (lane < 32) wh_db_filter_line (S, lane, dir, D, O);
    WV_LANES_END
  }
#else
  // ---- the plain restatement (test builds, A/B): every edge a lane block of its own, samples addressed in the LDS tile;
  //      vertical edges (dir 0) then horizontal edges (dir 1) ----
  for (int dir = 0; dir < 2; ++dir) {
    for (int e = 0; e < 4; ++e) {
      if (e == 0 && !(dir == 0 ? left_ok : top_ok)) continue;
      int eq = qp, eqc = qpc;
      if (e == 0) {
        const WhMbState* N = dir == 0 ? Nl : Nt;
        eq = (qp + N->luma_qp + 1) >> 1;
        eqc = (qpc + N->chroma_qp + 1) >> 1;
      }
      const int ia = wh_clip3 (eq + P.alpha_offset, 0, 51), ib = wh_clip3 (eq + P.beta_offset, 0, 51);
      const int alpha = kWhAlpha[ia], beta = kWhBeta[ib];
      const int iac = wh_clip3 (eqc + P.alpha_offset, 0, 51), ibc = wh_clip3 (eqc + P.beta_offset, 0, 51);
      const int alphac = kWhAlpha[iac], betac = kWhBeta[ibc];
      WV_LANES_BEGIN (lane)
      if (lane < 16) {
        if (alpha | beta) {
          const int bs = S.bs[dir][e][lane >> 2];
          uint8_t* q = dir == 0 ? &WH_DY (S, e * 4, lane) : &WH_DY (S, lane, e * 4);
          wh_db_luma_line (q, dir == 0 ? 1 : 24, bs, alpha, beta, ia);
        }
      } else if (lane < 32 && (e & 1) == 0) {
        if (alphac | betac) {
          const int pl = (lane - 16) >> 3, k = lane & 7;
          const int bs = S.bs[dir][e][k >> 1];
          uint8_t* q = dir == 0 ? &WH_DC (S, pl, e * 2, k) : &WH_DC (S, pl, k, e * 2);
          wh_db_chroma_line (q, dir == 0 ? 1 : 12, bs, alphac, betac, iac);
        }
      }
      WV_LANES_END
    }
  }
#endif   // WH_DB_PER_EDGE

  }   // filtered

  WH_PROF_MARK (P, S, 7);   // edge filters
  // ---- write-back (see the ownership rule above) ----
  const bool right_in = mbx < w - 1 && xy + 1 < last, below_in = xy + w < last;
  const bool lb_none = xy - 1 + w >= last;        // the left MB has no neighbour below it inside the slice
  const bool own = filtered || J.rec_blk != nullptr;      // the MB's own samples are written: always when the picture does not hold the unfiltered ones (WhPicJob::rec_blk)
  // A macroblock in the middle of its slice -- left, upper, right and lower neighbours all inside the workgroup's range (19 of 20 macroblocks of
  // a 1080p picture) -- owns sixteen 16-sample luma rows (rows -4 .. -1 from x = 0, rows 0 .. 11 from x = -4) and per chroma plane eight
  // 8-sample rows (rows -2 .. -1 from x = 0, rows 0 .. 5 from x = -4): one 16-byte / 8-byte store per row on 32 lanes (round 6).  The general
  // loop below asked every one of 160 words whether it is this macroblock's -- two passes over the lanes with a division by 5 and by 3 each:
  // "write-back + exchange" was 4.0 k of the pass's 19.9 k cycles per macroblock (profiles/r06_phase_cycles_before.txt).
  // The picture's TILED TWIN (what the next picture's search windows are fetched from, common/wh_types.h WH_TILE_*) gets the same samples here
  // (round 6): every sample inside the picture is written by exactly one macroblock of this pass, so the tiling pass behind the border expansion
  // only has the borders left to copy (kernels/tile_pic.h) instead of reading the whole planar picture back and writing it again -- 3.4 MB each way
  // per 1080p picture, 0.31 ms per step of 256.  A luma row piece is one 16-byte tile row (x = 0 .. 15) or the last word of the left tile's row and
  // the first three of this one's (x = -4 .. 11); chroma tile rows hold 8 Cb then 8 Cr samples.
  const bool tiles = J.rec_blk != nullptr && J.rec_tiles[0] != nullptr;
  const bool interior = top_lds && left_lds && own && below_in && right_in && !lb_none && !xwg;
  if (interior) {
    WV_LANES_BEGIN (lane)
    if (lane < 32) wh_db_store_interior_line (S, lane, P, J, mbx, mby, tiles);
    WV_LANES_END
  } else {
  WV_LANES_BEGIN (lane)
  {
    WH_G uint8_t* ry = (WH_G uint8_t*)J.rec[0] + (ptrdiff_t) (mby * 16) * P.rec_stride_y + mbx * 16;
    for (int i = lane; i < 20 * 5; i += 64) {
      const int row = i / 5 - 4, x = (i % 5) * 4 - 4;
      bool wr;
      if (row < 0) wr = x >= 0 && (top_lds ? true : (filtered && top_ok && row >= -3));
      else if (x < 0) wr = left_lds ? (row < 12 || lb_none) : (filtered && left_ok);
      else wr = own && (row < 12 || !below_in) && (x < 12 || !right_in);
      if (wr) {
        WH_G uint32_t* d = (WH_G uint32_t*) (ry + (ptrdiff_t)row * P.rec_stride_y + x);
        const uint32_t v = * (const uint32_t*)&WH_DY (S, x, row);
        if (xwg) wh_st_xwg32 (d, v); else *d = v;
        if (tiles) {          // (the same rule as for the planar word: what another band's workgroup -- another XCD's L2 -- may write again goes through to memory)
          WH_G uint32_t* tp = (WH_G uint32_t*) ((WH_G uint8_t*)J.rec_tiles[0] + wh_db_tile_y_off (P.rec_stride_y, mbx * 16 + x, mby * 16 + row) + (uint32_t) (x & 12));
          if (xwg) wh_st_xwg32 (tp, v); else *tp = v;
        }
      }
    }
    if (lane < 60) {
      const int pl = lane / 30, k = lane % 30, row = k / 3 - 2, x = (k % 3) * 4 - 4;
      bool wr;
      if (row < 0) wr = x >= 0 && (top_lds ? true : (filtered && top_ok && row >= -1));
      else if (x < 0) wr = left_lds ? (row < 6 || lb_none) : (filtered && left_ok);
      else wr = own && (row < 6 || !below_in) && (x < 4 || !right_in);
      if (wr) {
        WH_G uint32_t* d = (WH_G uint32_t*) ((WH_G uint8_t*)J.rec[1 + pl] + (ptrdiff_t) (mby * 8 + row) * P.rec_stride_c + mbx * 8 + x);
        const uint32_t v = * (const uint32_t*)&WH_DC (S, pl, x, row);
        if (xwg) wh_st_xwg32 (d, v); else *d = v;
        if (tiles) {
          WH_G uint32_t* tp = (WH_G uint32_t*) ((WH_G uint8_t*)J.rec_tiles[1] + wh_db_tile_c_off (P.rec_stride_c, mbx * 8 + x, mby * 8 + row) + (uint32_t) (pl * 8 + (x & 4)));
          if (xwg) wh_st_xwg32 (tp, v); else *tp = v;
        }
      }
    }
  }
  WV_LANES_END
  }
  wh_db_publish (S, etop, eleft, filtered && left_ok && left_lds);
  WH_PROF_MARK (P, S, 8);   // write-back + strip exchange
  return filtered && ((left_ok && !left_lds) || (top_ok && !top_lds));
}

// ---- two macroblocks per wavefront (round 6) ---------------------------------------------------------------------------------------------
// The boundary strengths, the edge filters and the interior write-back keep 32 of a wave's 64 lanes busy, and they are most of the pass's vector
// instructions.  Two macroblocks of one 2:1 diagonal -- A = (x, y) and B = (x - 2, y + 1), consecutive in the processing order -- are independent
// (they are what two waves work on at the same time otherwise): lanes 0 .. 31 take A, lanes 32 .. 63 take B, each half on a tile of its own, in ONE
// instruction stream.  What is uniform per macroblock (neighbour availability, edge QPs, alpha / beta / tc0, the strengths' words) is worked out
// for both on the scalar unit and chosen per lane.  Only for pairs of macroblocks in the middle of the band (wh_db_mb_interior: left, upper, right and
// lower neighbours inside it, so both strips come from the exchange buffers and the write-back is the interior one) of a picture whose unfiltered
// samples arrive macroblock by macroblock (WhPicJob::rec_blk); the processing order pairs them up (common/mb_order.h wh_build_db_pair_items).
WH_FN WhDbDir wh_db_dir_sel (bool b, const WhDbDir& A, const WhDbDir& B) {
  WhDbDir D;
  D.alpha0 = b ? B.alpha0 : A.alpha0; D.beta0 = b ? B.beta0 : A.beta0; D.tc30 = b ? B.tc30 : A.tc30;
  D.alpha1 = b ? B.alpha1 : A.alpha1; D.beta1 = b ? B.beta1 : A.beta1; D.tc31 = b ? B.tc31 : A.tc31;
  D.alphac0 = b ? B.alphac0 : A.alphac0; D.betac0 = b ? B.betac0 : A.betac0; D.tc3c0 = b ? B.tc3c0 : A.tc3c0;
  D.alphac1 = b ? B.alphac1 : A.alphac1; D.betac1 = b ? B.betac1 : A.betac1; D.tc3c1 = b ? B.tc3c1 : A.tc3c1;
  D.bsw0 = b ? B.bsw0 : A.bsw0; D.bsw1 = b ? B.bsw1 : A.bsw1; D.bsw2 = b ? B.bsw2 : A.bsw2; D.bsw3 = b ? B.bsw3 : A.bsw3;
  D.act_l0 = b ? B.act_l0 : A.act_l0; D.act_l1 = b ? B.act_l1 : A.act_l1; D.act_l2 = b ? B.act_l2 : A.act_l2; D.act_l3 = b ? B.act_l3 : A.act_l3;
  D.act_c0 = b ? B.act_c0 : A.act_c0; D.act_c1 = b ? B.act_c1 : A.act_c1;
  return D;
}
WH_FN void wh_deblock_pair_body (WhDbLds* S2, WhDbStage* G2, const WhDbXchg& E, const WhSeqParams& P, const WhPicJob& J, int ax, int ay, int bx, int by) {
  WH_PROF_DECL (P);
  WhDbLds& SA = S2[0];
  WhDbLds& SB = S2[1];
  uint32_t* etopA = E.top + ax * 24;
  uint32_t* eleftA = E.left + (ay - E.first_row) * 32;
  uint32_t* etopB = E.top + bx * 24;
  uint32_t* eleftB = E.left + (by - E.first_row) * 32;
  wh_db_assemble (SA, G2[0], etopA, eleftA, true, true, P, J, ax, ay);
  wh_db_assemble (SB, G2[1], etopB, eleftB, true, true, P, J, bx, by);
  const WhMbState* MA = (const WhMbState*)&SA.st[0];
  const WhMbState* MB = (const WhMbState*)&SB.st[0];
  const int fidc = (P.deblock_idc != 0);        // 1: do not filter across slice boundaries
  const bool left_okA = !fidc || MA->slice_idc == ((const WhMbState*)&SA.st[36])->slice_idc, top_okA = !fidc || MA->slice_idc == ((const WhMbState*)&SA.st[72])->slice_idc;
  const bool left_okB = !fidc || MB->slice_idc == ((const WhMbState*)&SB.st[36])->slice_idc, top_okB = !fidc || MB->slice_idc == ((const WhMbState*)&SB.st[72])->slice_idc;
  const int typeA = MA->mb_type, typeB = MB->mb_type;
  const bool intraA = WH_IS_INTRA (typeA), intraB = WH_IS_INTRA (typeB);
  WH_PROF_MARK (P, SA, 5);   // neighbour strips + staged inputs assembled in the tiles
  // ---- boundary strengths of both ----
  WV_LANES_BEGIN (lane)
  {
    const bool b = lane >= 32;
    const int l = lane & 31;
    WhDbLds& S = S2[lane >> 5];
    S.bs[l >> 4][(l >> 2) & 3][l & 3] = (uint8_t)wh_db_bs_lane ((const WhMbState*)&S.st[0], (const WhMbState*)&S.st[36], (const WhMbState*)&S.st[72],
                                                               b ? left_okB : left_okA, b ? top_okB : top_okA, b ? intraB : intraA, b ? typeB : typeA, l);
  }
  WV_LANES_END
  int any_a, any_b;
  WV_ANY (any_a, lane, (lane < 32 && SA.bs[lane >> 4][(lane >> 2) & 3][lane & 3] != 0));
  WV_ANY (any_b, lane, (lane >= 32 && SB.bs[(lane >> 4) & 1][(lane >> 2) & 3][lane & 3] != 0));
  const bool filteredA = any_a != 0, filteredB = any_b != 0;
  WH_PROF_MARK (P, SA, 6);   // boundary strengths
  if (filteredA || filteredB) {
    const int qpA = MA->luma_qp, qpcA = MA->chroma_qp, qpB = MB->luma_qp, qpcB = MB->chroma_qp;
    for (int dir = 0; dir < 2; ++dir) {
      const WhDbDir DA = wh_db_dir_params (SA, P, dir, dir == 0 ? left_okA : top_okA, (const WhMbState*)&SA.st[dir == 0 ? 36 : 72], qpA, qpcA);
      const WhDbDir DB = wh_db_dir_params (SB, P, dir, dir == 0 ? left_okB : top_okB, (const WhMbState*)&SB.st[dir == 0 ? 36 : 72], qpB, qpcB);
      if ((DA.bsw0 | DA.bsw1 | DA.bsw2 | DA.bsw3 | DB.bsw0 | DB.bsw1 | DB.bsw2 | DB.bsw3) == 0) continue;
      const WhDbOn OA = wh_db_dir_on (DA), OB = wh_db_dir_on (DB);
      WhDbOn O;
      O.on0 = OA.on0 || OB.on0; O.on1 = OA.on1 || OB.on1; O.on2 = OA.on2 || OB.on2; O.on3 = OA.on3 || OB.on3;
      O.any40 = OA.any40 || OB.any40; O.any41 = OA.any41 || OB.any41; O.any42 = OA.any42 || OB.any42; O.any43 = OA.any43 || OB.any43;
      WV_LANES_BEGIN (lane)
      {
        const WhDbDir D = wh_db_dir_sel (lane >= 32, DA, DB);
        wh_db_filter_line (S2[lane >> 5], lane & 31, dir, D, O);
      }
      WV_LANES_END
    }
  }
  WH_PROF_MARK (P, SA, 7);   // edge filters
  // ---- write-back: both are interior macroblocks ----
  const bool tiles = J.rec_tiles[0] != nullptr;
  WV_LANES_BEGIN (lane)
  wh_db_store_interior_line (S2[lane >> 5], lane & 31, P, J, lane >= 32 ? bx : ax, lane >= 32 ? by : ay, tiles);
  WV_LANES_END
  wh_db_publish (SA, etopA, eleftA, filteredA && left_okA);
  wh_db_publish (SB, etopB, eleftB, filteredB && left_okB);
  WH_PROF_MARK (P, SA, 8);   // write-back + strip exchange
}
