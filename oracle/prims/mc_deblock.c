/* mc_deblock.c -- TEST INFRASTRUCTURE ONLY (see oracle_prims.h). */
#include "oracle_prims.h"
#include <stdlib.h>

static uint8_t clip255 (int v) { return (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v); }
static int tap6 (int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }

/* codec/common/src/mc.cpp:187-347 McHorVer20/02/22_c, McHorVerXY_c, McLuma_c: half samples by the 6-tap filter
 * (b,h: (x+16)>>5; j: 6-tap over unrounded intermediates, (x+512)>>10), quarter samples by (a+b+1)>>1 */
static int hb (const uint8_t* p) { return clip255 ((tap6 (p[-2], p[-1], p[0], p[1], p[2], p[3]) + 16) >> 5); }
static int hh (const uint8_t* p, int s) { return clip255 ((tap6 (p[-2 * s], p[-s], p[0], p[s], p[2 * s], p[3 * s]) + 16) >> 5); }
static int hj (const uint8_t* p, int s) {
  int v[6];
  for (int k = 0; k < 6; ++k) { const uint8_t* r = p + (k - 2) * s; v[k] = tap6 (r[-2], r[-1], r[0], r[1], r[2], r[3]); }
  return clip255 ((tap6 (v[0], v[1], v[2], v[3], v[4], v[5]) + 512) >> 10);
}
void orc_mc_luma (const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int mvx, int mvy, int w, int h) {
  const int fx = mvx & 3, fy = mvy & 3;        /* src already points at the integer position, as in McLuma_c */
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = src + y * ss + x;
      int v;
      switch (fy * 4 + fx) {
      case 0: v = p[0]; break;
      case 1: v = (p[0] + hb (p) + 1) >> 1; break;
      case 2: v = hb (p); break;
      case 3: v = (p[1] + hb (p) + 1) >> 1; break;
      case 4: v = (p[0] + hh (p, ss) + 1) >> 1; break;
      case 5: v = (hb (p) + hh (p, ss) + 1) >> 1; break;
      case 6: v = (hb (p) + hj (p, ss) + 1) >> 1; break;
      case 7: v = (hb (p) + hh (p + 1, ss) + 1) >> 1; break;
      case 8: v = hh (p, ss); break;
      case 9: v = (hh (p, ss) + hj (p, ss) + 1) >> 1; break;
      case 10: v = hj (p, ss); break;
      case 11: v = (hj (p, ss) + hh (p + 1, ss) + 1) >> 1; break;
      case 12: v = (p[ss] + hh (p, ss) + 1) >> 1; break;
      case 13: v = (hh (p, ss) + hb (p + ss) + 1) >> 1; break;
      case 14: v = (hj (p, ss) + hb (p + ss) + 1) >> 1; break;
      default: v = (hh (p + 1, ss) + hb (p + ss) + 1) >> 1; break;
      }
      dst[y * ds + x] = (uint8_t)v;
    }
}
/* mc.cpp:349-378 McChroma_c: (A p00 + B p01 + C p10 + D p11 + 32) >> 6 with eighth-sample weights (g_kuiABCD :60-93) */
void orc_mc_chroma (const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int mvx, int mvy, int w, int h) {
  const int dx = mvx & 7, dy = mvy & 7;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = src + y * ss + x;
      dst[y * ds + x] = (uint8_t) (((8 - dx) * (8 - dy) * p[0] + dx * (8 - dy) * p[1] + (8 - dx) * dy * p[ss] + dx * dy * p[ss + 1] + 32) >> 6);
    }
}

static int clip3 (int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
/* codec/common/src/deblocking_common.cpp:5-43 DeblockLumaLt4_c (16 lines, tc per 4 lines, tc < 0 = skip) */
void orc_deblock_luma_lt4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta, const int8_t* tc4) {
  const int sx = horizontal ? 1 : stride, sy = horizontal ? stride : 1;
  for (int i = 0; i < 16; ++i, pix += sy) {
    const int tc0 = tc4[i >> 2];
    if (tc0 < 0) continue;
    const int p0 = pix[-sx], p1 = pix[-2 * sx], p2 = pix[-3 * sx], q0 = pix[0], q1 = pix[sx], q2 = pix[2 * sx];
    if (abs (p0 - q0) < alpha && abs (p1 - p0) < beta && abs (q1 - q0) < beta) {
      int tc = tc0;
      if (abs (p2 - p0) < beta) { pix[-2 * sx] = (uint8_t) (p1 + clip3 ((p2 + ((p0 + q0 + 1) >> 1) - (p1 * 2)) >> 1, -tc0, tc0)); tc++; }
      if (abs (q2 - q0) < beta) { pix[sx] = (uint8_t) (q1 + clip3 ((q2 + ((p0 + q0 + 1) >> 1) - (q1 * 2)) >> 1, -tc0, tc0)); tc++; }
      const int d = clip3 ((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
      pix[-sx] = clip255 (p0 + d); pix[0] = clip255 (q0 - d);
    }
  }
}
/* deblocking_common.cpp:44-90 DeblockLumaEq4_c */
void orc_deblock_luma_eq4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta) {
  const int sx = horizontal ? 1 : stride, sy = horizontal ? stride : 1;
  for (int i = 0; i < 16; ++i, pix += sy) {
    const int p0 = pix[-sx], p1 = pix[-2 * sx], p2 = pix[-3 * sx], q0 = pix[0], q1 = pix[sx], q2 = pix[2 * sx];
    const int d = abs (p0 - q0);
    if (d < alpha && abs (p1 - p0) < beta && abs (q1 - q0) < beta) {
      if (d < ((alpha >> 2) + 2)) {
        if (abs (p2 - p0) < beta) {
          const int p3 = pix[-4 * sx];
          pix[-sx] = (uint8_t) ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
          pix[-2 * sx] = (uint8_t) ((p2 + p1 + p0 + q0 + 2) >> 2);
          pix[-3 * sx] = (uint8_t) ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else pix[-sx] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
        if (abs (q2 - q0) < beta) {
          const int q3 = pix[3 * sx];
          pix[0] = (uint8_t) ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
          pix[sx] = (uint8_t) ((p0 + q0 + q1 + q2 + 2) >> 2);
          pix[2 * sx] = (uint8_t) ((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else pix[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
      } else {
        pix[-sx] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2);
        pix[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
      }
    }
  }
}
/* deblocking_common.cpp:92-181 DeblockChromaLt4_c / Eq4_c for one plane (8 lines, tc per 2 lines, tc <= 0 = skip) */
void orc_deblock_chroma_lt4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta, const int8_t* tc4) {
  const int sx = horizontal ? 1 : stride, sy = horizontal ? stride : 1;
  for (int i = 0; i < 8; ++i, pix += sy) {
    const int tc = tc4[i >> 1];
    if (tc <= 0) continue;
    const int p0 = pix[-sx], p1 = pix[-2 * sx], q0 = pix[0], q1 = pix[sx];
    if (abs (p0 - q0) < alpha && abs (p1 - p0) < beta && abs (q1 - q0) < beta) {
      const int d = clip3 ((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
      pix[-sx] = clip255 (p0 + d); pix[0] = clip255 (q0 - d);
    }
  }
}
void orc_deblock_chroma_eq4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta) {
  const int sx = horizontal ? 1 : stride, sy = horizontal ? stride : 1;
  for (int i = 0; i < 8; ++i, pix += sy) {
    const int p0 = pix[-sx], p1 = pix[-2 * sx], q0 = pix[0], q1 = pix[sx];
    if (abs (p0 - q0) < alpha && abs (p1 - p0) < beta && abs (q1 - q0) < beta) {
      pix[-sx] = (uint8_t) ((2 * p1 + p0 + q1 + 2) >> 2); pix[0] = (uint8_t) ((2 * q1 + q0 + p1 + 2) >> 2);
    }
  }
}

/* codec/processing/src/vaacalc/vaacalcfuncs.cpp:254-330 VAACalcSad_c, one macroblock of it */
void orc_vaa_sad8x8 (const uint8_t* cur, const uint8_t* ref, int32_t stride, int32_t* sad4) {
  for (int k = 0; k < 4; ++k) {
    const uint8_t* c = cur + (k >> 1) * 8 * stride + (k & 1) * 8;
    const uint8_t* r = ref + (k >> 1) * 8 * stride + (k & 1) * 8;
    int s = 0;
    for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) s += abs (c[y * stride + x] - r[y * stride + x]);
    sad4[k] = s;
  }
}
