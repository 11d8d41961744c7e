/* intra_pred.c -- TEST INFRASTRUCTURE ONLY (see oracle_prims.h).
 * Restates codec/encoder/core/src/get_intra_predictor.cpp:79-613 and
 * codec/common/src/intra_pred_common.cpp:47-77 (H.264 8.3.1.2 / 8.3.3 / 8.3.4). */
#include "oracle_prims.h"
#include <string.h>

static uint8_t clip255 (int v) { return (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v); }
#define F3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)
#define F2(a, b) (((a) + (b) + 1) >> 1)

/* get_intra_predictor.cpp:79-418; modes: 0 V,1 H,2 DC,3 DDL,4 DDR,5 VR,6 HD,7 VL,8 HU,9 DC_L,10 DC_T,11 DC_128,
 * 12 DDL_TOP,13 VL_TOP (the *_TOP forms replace the missing top-right samples by T3) */
void orc_pred_i4x4 (int mode, uint8_t* pred, const uint8_t* ref, int32_t st) {
  int T[8], L[4];
  const int top_ok = !(mode == 1 || mode == 8 || mode == 9 || mode == 11);
  const int left_ok = !(mode == 0 || mode == 3 || mode == 7 || mode == 10 || mode == 11 || mode == 12 || mode == 13);
  const int tl_ok = (mode == 4 || mode == 5 || mode == 6);
  for (int i = 0; i < 8; ++i) T[i] = top_ok ? ((mode == 3 || mode == 7) || i < 4 ? ref[-st + i] : ref[-st + 3]) : 0;
  if (mode == 12 || mode == 13) for (int i = 4; i < 8; ++i) T[i] = ref[-st + 3];
  for (int i = 0; i < 4; ++i) L[i] = left_ok ? ref[i * st - 1] : 0;
  const int TL = tl_ok ? ref[-st - 1] : 0;
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) {
      int v = 128;
      switch (mode) {
      case 0: v = T[x]; break;
      case 1: v = L[y]; break;
      case 2: v = (T[0] + T[1] + T[2] + T[3] + L[0] + L[1] + L[2] + L[3] + 4) >> 3; break;
      case 9: v = (L[0] + L[1] + L[2] + L[3] + 2) >> 2; break;
      case 10: v = (T[0] + T[1] + T[2] + T[3] + 2) >> 2; break;
      case 11: v = 128; break;
      case 3: case 12: v = (x == 3 && y == 3) ? (T[6] + 3 * T[7] + 2) >> 2 : F3 (T[x + y], T[x + y + 1], T[x + y + 2]); break;
      case 4: {
        if (x > y) v = (x - y == 1) ? F3 (TL, T[0], T[1]) : F3 (T[x - y - 2], T[x - y - 1], T[x - y]);
        else if (x < y) v = (y - x == 1) ? F3 (TL, L[0], L[1]) : F3 (L[y - x - 2], L[y - x - 1], L[y - x]);
        else v = F3 (T[0], TL, L[0]);
        break;
      }
      case 5: {
        const int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0 && !(z & 1)) v = i == 0 ? F2 (TL, T[0]) : F2 (T[i - 1], T[i]);
        else if (z >= 0) v = i == 1 ? F3 (TL, T[0], T[1]) : F3 (T[i - 2], T[i - 1], T[i]);
        else if (z == -1) v = F3 (L[0], TL, T[0]);
        else v = (y == 2) ? F3 (L[1], L[0], TL) : F3 (L[2], L[1], L[0]);
        break;
      }
      case 6: {
        const int z = 2 * y - x, j = y - (x >> 1);
        if (z >= 0 && !(z & 1)) v = j == 0 ? F2 (TL, L[0]) : F2 (L[j - 1], L[j]);
        else if (z >= 0) v = j == 1 ? F3 (TL, L[0], L[1]) : F3 (L[j - 2], L[j - 1], L[j]);
        else if (z == -1) v = F3 (L[0], TL, T[0]);
        else v = (x == 2) ? F3 (T[1], T[0], TL) : F3 (T[2], T[1], T[0]);
        break;
      }
      case 7: case 13: { const int i = x + (y >> 1); v = (y & 1) ? F3 (T[i], T[i + 1], T[i + 2]) : F2 (T[i], T[i + 1]); break; }
      case 8: {
        const int z = x + 2 * y, j = y + (x >> 1);
        if (z > 5) v = L[3];
        else if (z == 5) v = (L[2] + 3 * L[3] + 2) >> 2;
        else v = (z & 1) ? F3 (L[j], L[j + 1], L[j + 2]) : F2 (L[j], L[j + 1]);
        break;
      }
      }
      pred[y * 4 + x] = (uint8_t)v;
    }
}

/* modes: 0 V,1 H,2 DC,3 Plane,4 DC_L,5 DC_T,6 DC_128 (intra_pred_common.cpp:47-77, get_intra_predictor.cpp:536-612) */
void orc_pred_i16x16 (int mode, uint8_t* pred, const uint8_t* ref, int32_t st) {
  int st_sum = 0, sl_sum = 0, a = 0, b = 0, c = 0;
  if (mode == 2 || mode == 5) for (int i = 0; i < 16; ++i) st_sum += ref[-st + i];
  if (mode == 2 || mode == 4) for (int i = 0; i < 16; ++i) sl_sum += ref[i * st - 1];
  if (mode == 3) {
    int h = 0, v = 0;
    for (int i = 0; i < 8; ++i) { h += (i + 1) * (ref[-st + 8 + i] - ref[-st + 6 - i]); v += (i + 1) * (ref[(8 + i) * st - 1] - ref[(6 - i) * st - 1]); }
    a = (ref[15 * st - 1] + ref[-st + 15]) << 4; b = (5 * h + 32) >> 6; c = (5 * v + 32) >> 6;
  }
  for (int y = 0; y < 16; ++y)
    for (int x = 0; x < 16; ++x) {
      int v;
      switch (mode) {
      case 0: v = ref[-st + x]; break;
      case 1: v = ref[y * st - 1]; break;
      case 2: v = (st_sum + sl_sum + 16) >> 5; break;
      case 3: v = clip255 ((a + b * (x - 7) + c * (y - 7) + 16) >> 5); break;
      case 4: v = (sl_sum + 8) >> 4; break;
      case 5: v = (st_sum + 8) >> 4; break;
      default: v = 128; break;
      }
      pred[y * 16 + x] = (uint8_t)v;
    }
}

/* modes: 0 DC,1 H,2 V,3 Plane,4 DC_L,5 DC_T,6 DC_128 (get_intra_predictor.cpp:420-534) */
void orc_pred_chroma (int mode, uint8_t* pred, const uint8_t* ref, int32_t st) {
  int t0 = 0, t1 = 0, l0 = 0, l1 = 0, a = 0, b = 0, c = 0;
  if (mode == 0 || mode == 5) for (int i = 0; i < 4; ++i) { t0 += ref[-st + i]; t1 += ref[-st + 4 + i]; }
  if (mode == 0 || mode == 4) for (int i = 0; i < 4; ++i) { l0 += ref[i * st - 1]; l1 += ref[(4 + i) * st - 1]; }
  if (mode == 3) {
    int h = 0, v = 0;
    for (int i = 0; i < 4; ++i) { h += (i + 1) * (ref[-st + 4 + i] - ref[-st + 2 - i]); v += (i + 1) * (ref[(4 + i) * st - 1] - ref[(2 - i) * st - 1]); }
    a = (ref[7 * st - 1] + ref[-st + 7]) << 4; b = (17 * h + 16) >> 5; c = (17 * v + 16) >> 5;
  }
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      int v;
      switch (mode) {
      case 0: v = y < 4 ? (x < 4 ? (t0 + l0 + 4) >> 3 : (t1 + 2) >> 2) : (x < 4 ? (l1 + 2) >> 2 : (t1 + l1 + 4) >> 3); break;
      case 1: v = ref[y * st - 1]; break;
      case 2: v = ref[-st + x]; break;
      case 3: v = clip255 ((a + b * (x - 3) + c * (y - 3) + 16) >> 5); break;
      case 4: v = y < 4 ? (l0 + 2) >> 2 : (l1 + 2) >> 2; break;
      case 5: v = x < 4 ? (t0 + 2) >> 2 : (t1 + 2) >> 2; break;
      default: v = 128; break;
      }
      pred[y * 8 + x] = (uint8_t)v;
    }
}
