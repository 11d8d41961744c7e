// bgd_pic.h -- background detection of a source picture: which macroblocks the pre-processing marks as background, from the
// pre-analysis statistics (vaa_pic.h) and the chroma edges of the two source pictures.
//
// Reference: codec/processing/src/backgrounddetection/BackgroundDetection.cpp
//   :117-160  GetOUParameters                      per 16x16 unit: SAD, |sum of differences|, largest difference, the 8x8 blocks' smallest
//                                                   largest-difference and the spread of their sums of differences
//   :162-195  ForegroundBackgroundDivision         the coarse verdict of every unit on its own
//   :197-316  ForegroundDilation* / BackgroundErosion   a unit's verdict revised from its four neighbours' (+ chroma edges)
//   :318-331  UpperOUForegroundCheck               the unit ABOVE loses a background verdict that at most one of its neighbours shares
//   :333-374  ForegroundDilationAndBackgroundErosion    ONE raster pass, in place: a unit reads its left and upper neighbours as revised, its
//                                                   right and lower ones as the coarse pass left them, and then revises the unit above it
// The pass is serial as written, but unit (i, j) only needs (i - 1, j) and (i + 1, j - 1) to be through (the latter because the revision of
// the unit above reads ITS right neighbour's own verdict): the units with the same i + 2 j can go together.  One workgroup walks those
// diagonals (a barrier each); the verdicts live in LDS, the statistics are read where they are needed.  Unit = macroblock (BGD_OU_SIZE 16);
// the picture's width must be a multiple of 16 (else the reference indexes the statistics with two different row lengths -- the caller
// keeps such pictures on the host).
#pragma once
#include "prims.h"
#include "../common/wh_types.h"

#define WH_BGD_Q 128              /* BGD_OU_SIZE * Q_FACTOR */
#define WH_BGD_THD_SAD 512        /* 2 * BGD_OU_SIZE * BGD_OU_SIZE */
#define WH_BGD_THD_ASD_UV 32      /* 4 * BGD_OU_SIZE_UV */

typedef struct WhBgdIn {
  const int32_t* sad8x8;           // [mb][4]  (rows of mb_w macroblocks)
  const int32_t* sd8x8;            // [mb][4]
  const uint8_t* mad8x8;           // [mb][4]
  const uint8_t* cur;              // the two source pictures, macroblock-tiled (WH_SRC_*)
  const uint8_t* ref;
  int w, h;                        // units: (width >> 4) x (height >> 4)
  int mb_w;
} WhBgdIn;
typedef struct WhBgdOu { int sad, sd, mad, min_sub_mad, max_diff_sub_sd; } WhBgdOu;

WH_FN WhBgdOu wh_bgd_ou (const WhBgdIn& I, int i, int j) {
  const int xy = j * I.mb_w + i;
  const WH_G int32_t* s = (const WH_G int32_t*)I.sad8x8 + 4 * xy;
  const WH_G int32_t* d = (const WH_G int32_t*)I.sd8x8 + 4 * xy;
  const WH_G uint8_t* m = (const WH_G uint8_t*)I.mad8x8 + 4 * xy;
  const int d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
  WhBgdOu u;
  u.sad = s[0] + s[1] + s[2] + s[3];
  u.sd = wh_abs (d0 + d1 + d2 + d3);
  u.mad = wh_max (wh_max (m0, m1), wh_max (m2, m3));
  u.min_sub_mad = wh_min (wh_min (m0, m1), wh_min (m2, m3));
  u.max_diff_sub_sd = wh_max (wh_max (d0, d1), wh_max (d2, d3)) - wh_min (wh_min (d0, d1), wh_min (d2, d3));
  return u;
}
WH_FN int wh_bgd_coarse (const WhBgdOu& u) {
  if (u.mad > 63) return 0;
  if ((u.max_diff_sub_sd <= (u.sad >> 3) || u.max_diff_sub_sd <= WH_BGD_Q) && u.sad < (WH_BGD_THD_SAD << 1)) {
    if (u.sad <= WH_BGD_Q) return 1;
    return u.sad < WH_BGD_THD_SAD ? (u.sd < ((u.sad * 3) >> 2)) : ((u.sd << 1) < u.sad);
  }
  return 0;
}
// ForegroundDilation23Luma: f* / m* = the four neighbours' verdicts and largest differences (left, right, up, down)
WH_FN bool wh_bgd_dilation23_luma (const WhBgdOu& u, const int* f, const int* m) {
  if (u.mad > (u.min_sub_mad << 1)) {
    int fg = 0, bg = 0;
    for (int k = 0; k < 4; ++k) { fg = wh_max (fg, (f[k] - 1) & m[k]); bg = wh_max (bg, ((!f[k]) - 1) & m[k]); }
    return (fg > (u.min_sub_mad << 2)) || (u.mad > (bg << 1) && u.mad <= ((fg * 3) >> 1));
  }
  return false;
}
// ForegroundDilation23Chroma: the chroma block's edges towards the foreground neighbours (bit 0 left, 1 right, 2 up, 3 down); V first, then U
WH_FN bool wh_bgd_dilation23_chroma (const WhBgdIn& I, int fg_bits, int i, int j) {
  const size_t blk = (size_t) (j * I.mb_w + i) * WH_SRC_MB_BYTES + 256;
  for (int pl = 1; pl >= 0; --pl) {
    const WH_G uint8_t* c = (const WH_G uint8_t*)I.cur + blk + pl * 64;
    const WH_G uint8_t* r = (const WH_G uint8_t*)I.ref + blk + pl * 64;
    for (int e = 0; e < 4; ++e) {
      if (!(fg_bits & (1 << e))) continue;
      const int first = e == 1 ? 7 : e == 3 ? 56 : 0, step = e < 2 ? 8 : 1;
      int asd = 0;
      for (int k = 0; k < 8; ++k) asd += (int)c[first + k * step] - (int)r[first + k * step];
      if (wh_abs (asd) > WH_BGD_THD_ASD_UV) return true;
    }
  }
  return false;
}
// One unit of the pass: `fl` = the verdicts of all units ([j * w + i]), `mbflag` = the caller-visible flags ([j * mb_w + i])
WH_FN void wh_bgd_step (const WhBgdIn& I, uint8_t* fl, WH_G int8_t* mbflag, int i, int j) {
  const int W = I.w, H = I.h;
  const int il = i > 0 ? i - 1 : i, ir = i < W - 1 ? i + 1 : i, ju = j > 0 ? j - 1 : j, jd = j < H - 1 ? j + 1 : j;
  const WhBgdOu u = wh_bgd_ou (I, i, j);
  const WhBgdOu nl = wh_bgd_ou (I, il, j), nr = wh_bgd_ou (I, ir, j), nu = wh_bgd_ou (I, i, ju), nd = wh_bgd_ou (I, i, jd);
  const int f[4] = {fl[j * W + il], fl[j * W + ir], fl[ju * W + i], fl[jd * W + i]};
  const int m[4] = {nl.mad, nr.mad, nu.mad, nd.mad};
  const int sum = f[0] + f[1] + f[2] + f[3];
  int flag = fl[j * W + i];
  if (flag) {                                      // ForegroundDilation
    if (u.sad > WH_BGD_Q) {
      if (sum <= 1) flag = 0;
      else if (sum <= 3) {
        flag = !wh_bgd_dilation23_luma (u, f, m);
        if (flag) flag = !wh_bgd_dilation23_chroma (I, (!f[0]) | ((!f[1]) << 1) | ((!f[2]) << 2) | ((!f[3]) << 3), i, j);
      }
    }
  } else if (u.max_diff_sub_sd <= WH_BGD_Q) {      // BackgroundErosion
    const int bg_sad = (nl.sad & -f[0]) + (nu.sad & -f[2]) + (nr.sad & -f[1]) + (nd.sad & -f[3]);
    if (u.sad * sum <= ((3 * bg_sad) >> 1)) {
      if (sum == 4) flag = 1;
      else if ((f[0] & f[1]) || (f[2] & f[3])) flag = !wh_bgd_dilation23_luma (u, f, m);
    }
  }
  fl[j * W + i] = (uint8_t)flag;
  // the unit above (UpperOUForegroundCheck): a background verdict that at most one of its neighbours shares is taken back
  if (j > 1 && i > 0 && i < W - 1 && fl[(j - 1) * W + i] == 1) {
    const WhBgdOu a = wh_bgd_ou (I, i, j - 1);
    if (a.sad > WH_BGD_Q && fl[(j - 1) * W + i - 1] + fl[(j - 1) * W + i + 1] + fl[(j - 2) * W + i] + flag <= 1) {
      mbflag[(j - 1) * I.mb_w + i] = 0;
      fl[(j - 1) * W + i] = 0;
    }
  }
  mbflag[j * I.mb_w + i] = (int8_t)flag;
}
// number of diagonals of the pass; the units of diagonal t are (t - 2 j, j)
WH_FN int wh_bgd_steps (int w, int h) { return (w - 1) + 2 * (h - 1) + 1; }
