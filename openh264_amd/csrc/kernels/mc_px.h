// mc_px.h -- per-sample H.264 interpolation on a plain byte pointer (any address space); used by the batched leaf
// primitives (hip/prims.hip).  The MB kernel interpolates four samples per lane from its LDS window instead
// (inter_mb.h wh_mc4); both restate codec/common/src/mc.cpp:100-386 (= H.264 8.4.2.2).
#pragma once
#include "prims.h"
// ---- H.264 luma sample interpolation from a byte tile (stride st), integer position p, frac (fx,fy) ----
WH_FN int wh_px_tap6 (int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
WH_FN int wh_mc_b (const uint8_t* p) { return wh_clip255 ((wh_px_tap6 (p[-2], p[-1], p[0], p[1], p[2], p[3]) + 16) >> 5); }
WH_FN int wh_mc_h (const uint8_t* p, int st) { return wh_clip255 ((wh_px_tap6 (p[-2 * st], p[-st], p[0], p[st], p[2 * st], p[3 * st]) + 16) >> 5); }
WH_FN int wh_mc_j (const uint8_t* p, int st) {
  int v[6];
  for (int k = 0; k < 6; ++k) { const uint8_t* r = p + (k - 2) * st; v[k] = wh_px_tap6 (r[-2], r[-1], r[0], r[1], r[2], r[3]); }
  return wh_clip255 ((wh_px_tap6 (v[0], v[1], v[2], v[3], v[4], v[5]) + 512) >> 10);
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
