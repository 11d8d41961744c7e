// prims.h -- scalar/row-level integer primitives used inside the MB kernels.
//
// Each function restates, in our own form, the arithmetic of one C-fallback primitive of the
// reference (file:line cited per function); widths and wrap-arounds (int16 temporaries) are
// kept identical because the bitstream parity contract depends on them.
#pragma once
#include "wave.h"
#ifndef WH_TABLE
#if defined(WH_EMU)
#define WH_TABLE static const
#else
#define WH_TABLE static __device__ const
#endif
#endif
#include "../common/h264_tables.h"
#include "../common/wh_types.h"

// luma4x4BlkIdx -> (x,y) in 4-pixel units  (svc_base_layer_md.cpp:227-237 g_kiCoordinateIdx4x4X/Y)
WH_FN int wh_blk_x (int b) { return ((b & 1) | ((b >> 1) & 2)); }
WH_FN int wh_blk_y (int b) { return (((b >> 1) & 1) | ((b >> 2) & 2)); }
// zig-zag scan position k -> raster index (encode_mb_aux.cpp:371-386 WelsScan4x4DcAc_c)
WH_FN int wh_zigzag (int k) {
  // 0 1 4 8 5 2 3 6 9 12 13 10 7 11 14 15 packed 4 bits each
  const unsigned long long z = 0xFEB7ADC963258410ULL;
  return (int) ((z >> (k * 4)) & 15);
}

// ---- forward 4-point core transform butterfly (encode_mb_aux.cpp:313-357 WelsDctT4_c) -----------
WH_FN void wh_fdct4 (int d0, int d1, int d2, int d3, int16_t* o0, int16_t* o1, int16_t* o2, int16_t* o3) {
  const int16_t s0 = (int16_t) (d0 + d3), s3 = (int16_t) (d0 - d3);
  const int16_t s1 = (int16_t) (d1 + d2), s2 = (int16_t) (d1 - d2);
  *o0 = (int16_t) (s0 + s1);
  *o2 = (int16_t) (s0 - s1);
  *o1 = (int16_t) ((s3 * 2) + s2);
  *o3 = (int16_t) (s3 - (s2 * 2));
}

// ---- quantisation (encode_mb_aux.cpp:161-224: sign * (((ff + |x|) * mf) >> 16)) -----------------
WH_FN int16_t wh_quant1 (int16_t x, int ff, int mf) {
  const int sign = ((int) x) >> 31;
  const int a = (sign ^ (int) x) - sign;
  const int16_t q = (int16_t) (((ff + a) * mf) >> 16);
  return (int16_t) ((sign ^ (int) q) - sign);
}
// same, also returning the unsigned level (WelsQuantFour4x4Max_c :209-224)
WH_FN int16_t wh_quant1_abs (int16_t x, int ff, int mf, int16_t* absq) {
  const int sign = ((int) x) >> 31;
  const int a = (sign ^ (int) x) - sign;
  const int16_t q = (int16_t) (((ff + a) * mf) >> 16);
  *absq = q;
  return (int16_t) ((sign ^ (int) q) - sign);
}

// The same two with operands from the encoder's own tables (0 <= ff < 2^15, 0 < mf < 2^14: encode_mb_aux.cpp:39-157; |x| <= 2^15), which is
// what the fused kernels pass: the product is below 2^31 and both factors below 2^24, so the full-rate 24-bit multiply gives the same bits.
// (The leaf exports take the CALLER's rows, any int16: they keep the general functions above.)
WH_FN int16_t wh_quant1_t (int16_t x, int ff, int mf) {
  const int sign = ((int) x) >> 31;
  const int a = (sign ^ (int) x) - sign;
  const int16_t q = (int16_t) (wh_mul_u24 ((uint32_t) (ff + a), (uint32_t)mf) >> 16);
  return (int16_t) ((sign ^ (int) q) - sign);
}
WH_FN int16_t wh_quant1_abs_t (int16_t x, int ff, int mf, int16_t* absq) {
  const int sign = ((int) x) >> 31;
  const int a = (sign ^ (int) x) - sign;
  const int16_t q = (int16_t) (wh_mul_u24 ((uint32_t) (ff + a), (uint32_t)mf) >> 16);
  *absq = q;
  return (int16_t) ((sign ^ (int) q) - sign);
}

// ---- inverse 4-point butterflies (decode_mb_aux.cpp:164-199 WelsIDctT4Rec_c) --------------------
// horizontal pass: int16 outputs (the reference keeps iTemp[] in int16)
WH_FN void wh_idct4_h (int16_t c0, int16_t c1, int16_t c2, int16_t c3, int16_t* t0, int16_t* t1, int16_t* t2, int16_t* t3) {
  const int su = c0 + c2, du = c0 - c2;
  const int sd = c1 + (c3 >> 1), dd = (c1 >> 1) - c3;
  *t0 = (int16_t) (su + sd);
  *t1 = (int16_t) (du + dd);
  *t2 = (int16_t) (du - dd);
  *t3 = (int16_t) (su - sd);
}
// vertical pass + rounding; returns the four residuals of one column, top to bottom
WH_FN void wh_idct4_v (int16_t t0, int16_t t1, int16_t t2, int16_t t3, int* r0, int* r1, int* r2, int* r3) {
  const int sl = t0 + t2, dl = t0 - t2;
  const int dr = (t1 >> 1) - t3, sr = t1 + (t3 >> 1);
  *r0 = (sl + sr + 32) >> 6;
  *r1 = (dl + dr + 32) >> 6;
  *r2 = (dl - dr + 32) >> 6;
  *r3 = (sl - sr + 32) >> 6;
}

// ---- 4-point Hadamard used by SATD (sample.cpp:47-96 WelsSampleSatd4x4_c) -----------------------
WH_FN void wh_had4 (int a0, int a1, int a2, int a3, int* o0, int* o1, int* o2, int* o3) {
  const int s0 = a0 + a2, s1 = a1 + a3, s2 = a0 - a2, s3 = a1 - a3;
  *o0 = s0 + s1; *o1 = s2 + s3; *o2 = s2 - s3; *o3 = s0 - s1;
}

// ---- JVT-O079 single-coefficient score of one zig-zag block (encode_mb_aux.cpp:417-436) ---------
WH_FN int wh_single_ctr (const int16_t* lv) {
  int ctr = 0, idx = 15;
  while (idx >= 0 && lv[idx] == 0) --idx;
  while (idx >= 0) {
    --idx;
    int run = idx;
    while (idx >= 0 && lv[idx] == 0) --idx;
    run -= idx;
    ctr += (run == 0) ? 3 : (run <= 2) ? 2 : (run <= 5) ? 1 : 0;   // kiTRunTable {3,2,2,1,1,1,0...}
  }
  return ctr;
}
