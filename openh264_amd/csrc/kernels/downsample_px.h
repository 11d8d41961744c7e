// downsample_px.h -- one output sample of each of the reference's down-samplers (codec/processing/src/downsample/downsamplefuncs.cpp):
//   :47-68    DyadicBilinearDownsampler_c           2:1   ((a + b + 1) >> 1 + (c + d + 1) >> 1 + 1) >> 1
//   :70-113   DyadicBilinearQuarter / OneThird      4:1, 3:1: the same 2x2 average of every fourth / third sample
//   :115-181  GeneralBilinearFastDownsampler_c      any ratio, 16/15-bit weights, products truncated (luma)
//   :183-245  GeneralBilinearAccurateDownsampler_c  any ratio, 15-bit weights, 64-bit accumulation (chroma)
// Shared by the kernels (hip/downsample.hip) and by the CPU test build's twin of the picture-level entry point.
#pragma once
#include <stdint.h>
#include <stddef.h>
#if defined(__HIPCC__)
#define WH_DS_FN static __host__ __device__ __forceinline__
#else
#define WH_DS_FN static inline
#endif

// the 2x2 average at `s` (source row pitch `stride`)
WH_DS_FN uint8_t wh_ds_avg2x2 (const uint8_t* s, int stride) {
  const int t1 = (s[0] + s[1] + 1) >> 1, t2 = (s[stride] + s[stride + 1] + 1) >> 1;
  return (uint8_t) ((t1 + t2 + 1) >> 1);
}
// WELS_ROUND ((float)src / (float)dst * (1 << bits)): the scale factors of the general down-samplers
WH_DS_FN int wh_ds_round_scale (int src, int dst, int bits) {
  const float f = (float)src / (float)dst * (float) (1 << bits);
  return (int) (f + (f >= 0 ? 0.5f : -0.5f));
}
// output sample (x, y) of the general down-samplers; last column and last row are nearest-sample copies as in the reference
WH_DS_FN uint8_t wh_ds_general (const uint8_t* src, int src_stride, int dst_w, int dst_h, int x, int y, int scalex, int scaley, int accurate) {
  const int bw = accurate ? 15 : 16, bh = 15;
  const int64_t xinv = ((int64_t)1 << (bw - 1)) + (int64_t)x * scalex, yinv = ((int64_t)1 << (bh - 1)) + (int64_t)y * scaley;
  const int xx = (int) (xinv >> bw), yy = (int) (yinv >> bh);
  const uint8_t* p = src + (size_t)yy * src_stride + xx;
  if (y == dst_h - 1 || x == dst_w - 1) return p[0];
  const uint32_t fu = (uint32_t) (xinv & ((1 << bw) - 1)), fv = (uint32_t) (yinv & ((1 << bh) - 1));
  const uint32_t a = p[0], b = p[1], c = p[src_stride], d = p[src_stride + 1];
  if (accurate) {
    const int64_t k = 1 << 15;
    const int64_t v = ((k - 1 - fu) * (k - 1 - fv) * a + (int64_t)fu * (k - 1 - fv) * b + (k - 1 - fu) * (int64_t)fv * c + (int64_t)fu * fv * d + ((int64_t)1 << 29)) >> 30;
    return (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v);
  }
  const uint32_t sw = 1u << 16, sh = 1u << 15;
  uint32_t v = (((sw - 1 - fu) * (sh - 1 - fv)) >> 16) * a;
  v += ((fu * (sh - 1 - fv)) >> 16) * b;
  v += (((sw - 1 - fu) * fv) >> 16) * c;
  v += ((fu * fv) >> 16) * d;
  v >>= 14;
  v += 1;
  v >>= 1;
  return (uint8_t) (v > 255 ? 255 : v);
}

// ---- which steps CDownsampling::Process (downsample.cpp:144-277) takes from (sw, sh) to (dw, dh) -------------------------------------
// Pictures whose half exceeds 1920x1088 take one step whatever the ratio (2:1, 4:1, 3:1 when it is exactly that, else the general
// filters); everything else halves while the half is still larger than the target and ends with a half or the general filters.
// mode: 0 half, 1 quarter, 2 one third (WELSHIP_DS_* of include/welship.h), -1 general (luma fast, chroma accurate: the C function table)
typedef struct WhDsStage { int mode, sw, sh, dw, dh; } WhDsStage;
WH_DS_FN int wh_ds_plan (int sw, int sh, int dw, int dh, WhDsStage* st /*[8]*/) {
  int n = 0;
  if ((sw >> 1) > 1920 || (sh >> 1) > 1088) {
    const int mode = ((sw >> 1) == dw && (sh >> 1) == dh) ? 0 : ((sw >> 2) == dw && (sh >> 2) == dh) ? 1 : (sw / 3 == dw && sh / 3 == dh) ? 2 : -1;
    st[n].mode = mode; st[n].sw = sw; st[n].sh = sh; st[n].dw = dw; st[n].dh = dh;
    return 1;
  }
  for (;;) {
    const int hw = sw >> 1, hh = sh >> 1;
    if (n >= 8) return -1;
    if (hw == dw && hh == dh) { st[n].mode = 0; st[n].sw = sw; st[n].sh = sh; st[n].dw = dw; st[n].dh = dh; return n + 1; }
    if (hw > dw && hh > dh) { st[n].mode = 0; st[n].sw = sw; st[n].sh = sh; st[n].dw = hw; st[n].dh = hh; ++n; sw = hw; sh = hh; continue; }
    st[n].mode = -1; st[n].sw = sw; st[n].sh = sh; st[n].dw = dw; st[n].dh = dh;
    return n + 1;
  }
}
