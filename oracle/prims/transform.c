/* transform.c -- TEST INFRASTRUCTURE ONLY (see oracle_prims.h). */
#include "oracle_prims.h"
#include <stdlib.h>
#include "tables.inc"

static int pc (int i) { return ((i >> 2) & 1) + (i & 1); }          /* position class of raster index i */

/* codec/encoder/core/src/encode_mb_aux.cpp:313-357 WelsDctT4_c: rows then columns, int16 temporaries */
void orc_dct4x4 (int16_t* d, const uint8_t* p1, int32_t s1, const uint8_t* p2, int32_t s2) {
  int16_t t[16];
  for (int y = 0; y < 4; ++y) {
    const int16_t a0 = p1[y * s1 + 0] - p2[y * s2 + 0], a1 = p1[y * s1 + 1] - p2[y * s2 + 1];
    const int16_t a2 = p1[y * s1 + 2] - p2[y * s2 + 2], a3 = p1[y * s1 + 3] - p2[y * s2 + 3];
    const int16_t s0 = a0 + a3, s3 = a0 - a3, q1 = a1 + a2, q2 = a1 - a2;
    t[y * 4 + 0] = s0 + q1; t[y * 4 + 2] = s0 - q1; t[y * 4 + 1] = (s3 * 2) + q2; t[y * 4 + 3] = s3 - (q2 * 2);
  }
  for (int x = 0; x < 4; ++x) {
    const int16_t s0 = t[x] + t[12 + x], s3 = t[x] - t[12 + x], q1 = t[4 + x] + t[8 + x], q2 = t[4 + x] - t[8 + x];
    d[x] = s0 + q1; d[8 + x] = s0 - q1; d[4 + x] = (s3 * 2) + q2; d[12 + x] = s3 - (q2 * 2);
  }
}

/* encode_mb_aux.cpp:280-311 WelsHadamardT4Dc_c: gathers the 16 block DCs (blocks in luma4x4BlkIdx order,
 * 16 coefficients apart) into raster order, 4x4 Hadamard, (x+1)>>1, clip to int16 */
void orc_hadamard_t4_dc (int16_t* out, const int16_t* dct) {
  int32_t p[16];
  for (int i = 0; i < 16; i += 4) {
    const int idx = ((i & 8) << 4) + ((i & 4) << 3);
    const int32_t s0 = dct[idx] + dct[idx + 80], s3 = dct[idx] - dct[idx + 80];
    const int32_t s1 = dct[idx + 16] + dct[idx + 64], s2 = dct[idx + 16] - dct[idx + 64];
    p[i] = s0 + s1; p[i + 2] = s0 - s1; p[i + 1] = s3 + s2; p[i + 3] = s3 - s2;
  }
  for (int i = 0; i < 4; ++i) {
    const int32_t s0 = p[i] + p[i + 12], s3 = p[i] - p[i + 12], s1 = p[i + 4] + p[i + 8], s2 = p[i + 4] - p[i + 8];
    int32_t v[4] = { (s0 + s1 + 1) >> 1, (s3 + s2 + 1) >> 1, (s0 - s1 + 1) >> 1, (s3 - s2 + 1) >> 1 };
    for (int k = 0; k < 4; ++k) { if (v[k] < -32768) v[k] = -32768; if (v[k] > 32767) v[k] = 32767; }
    out[i] = (int16_t)v[0]; out[i + 4] = (int16_t)v[1]; out[i + 8] = (int16_t)v[2]; out[i + 12] = (int16_t)v[3];
  }
}

/* encode_mb_aux.cpp:161-178 WelsQuant4x4_c: sign * (((ff + |x|) * mf) >> 16); tables :39-157 */
static int16_t q1 (int16_t x, int ff, int mf) {
  const int sign = ((int)x) >> 31, a = (sign ^ (int)x) - sign;
  const int q = ((ff + a) * mf) >> 16;
  return (int16_t) ((sign ^ q) - sign);
}
void orc_quant4x4 (int16_t* d, int qp, int intra) {
  for (int i = 0; i < 16; ++i) d[i] = q1 (d[i], kQuantFF[(qp + (intra ? 6 : 0)) * 3 + pc (i)], kQuantMF[qp * 3 + pc (i)]);
}
/* encode_mb_aux.cpp:209-224 WelsQuantFour4x4Max_c (one block of it) */
int32_t orc_quant4x4_max (int16_t* d, int qp, int intra) {
  int16_t mx = 0;
  for (int i = 0; i < 16; ++i) {
    const int sign = ((int)d[i]) >> 31, a = (sign ^ (int)d[i]) - sign;
    const int16_t q = (int16_t) (((kQuantFF[(qp + (intra ? 6 : 0)) * 3 + pc (i)] + a) * kQuantMF[qp * 3 + pc (i)]) >> 16);
    if (mx < q) mx = q;
    d[i] = (int16_t) ((sign ^ (int)q) - sign);
  }
  return mx;
}
/* the rows of g_kiQuantInterFF[qp (+ 6 for intra)] / g_kiQuantMF[qp] (encode_mb_aux.cpp:39-157) as the encoder hands them to the quantiser
 * slots: eight entries, positions 0..7 of a 4x4 block (the second half of the block repeats them) */
void orc_quant_rows (int qp, int intra, int16_t* ff8, int16_t* mf8) {
  for (int i = 0; i < 8; ++i) { ff8[i] = kQuantFF[(qp + (intra ? 6 : 0)) * 3 + pc (i)]; mf8[i] = kQuantMF[qp * 3 + pc (i)]; }
}
/* encode_mb_aux.cpp:180-192 WelsQuant4x4Dc_c */
void orc_quant4x4_dc (int16_t* d, int16_t ff, int16_t mf) { for (int i = 0; i < 16; ++i) d[i] = q1 (d[i], ff, mf); }

/* encode_mb_aux.cpp:247-277 WelsHadamardQuant2x2_c */
int32_t orc_hadamard_quant2x2 (int16_t* rs, int16_t ff, int16_t mf, int16_t* dct, int16_t* blk) {
  const int16_t s0 = rs[0] + rs[32], s1 = rs[0] - rs[32], s2 = rs[16] + rs[48], s3 = rs[16] - rs[48];
  rs[0] = rs[16] = rs[32] = rs[48] = 0;
  dct[0] = q1 ((int16_t) (s0 + s2), ff, mf); dct[1] = q1 ((int16_t) (s0 - s2), ff, mf);
  dct[2] = q1 ((int16_t) (s1 + s3), ff, mf); dct[3] = q1 ((int16_t) (s1 - s3), ff, mf);
  int n = 0;
  for (int i = 0; i < 4; ++i) { blk[i] = dct[i]; n += blk[i] != 0; }
  return n;
}
/* encode_mb_aux.cpp:226-245 WelsHadamardQuant2x2Skip_c */
int32_t orc_hadamard_quant2x2_skip (const int16_t* rs, int16_t ff, int16_t mf) {
  const int16_t thr = (int16_t) (((1 << 16) - 1) / mf - ff);
  const int16_t s0 = rs[0] + rs[32], s1 = rs[0] - rs[32], s2 = rs[16] + rs[48], s3 = rs[16] - rs[48];
  const int16_t d0 = s0 + s2, d1 = s0 - s2, d2 = s1 + s3, d3 = s1 - s3;
  return abs (d0) > thr || abs (d1) > thr || abs (d2) > thr || abs (d3) > thr;
}

static const uint8_t kZig[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
/* encode_mb_aux.cpp:371-386 WelsScan4x4DcAc_c */
void orc_scan4x4_dcac (int16_t* lv, const int16_t* d) { for (int k = 0; k < 16; ++k) lv[k] = d[kZig[k]]; }
/* encode_mb_aux.cpp:388-401 WelsScan4x4Ac_c */
void orc_scan4x4_ac (int16_t* lv, const int16_t* d) { for (int k = 0; k < 15; ++k) lv[k] = d[kZig[k + 1]]; lv[15] = 0; }
/* encode_mb_aux.cpp:417-436 WelsCalculateSingleCtr4x4_c (JVT-O079) */
int32_t orc_single_ctr4x4 (const int16_t* lv) {
  static const int run_tab[16] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int ctr = 0, idx = 15;
  while (idx >= 0 && lv[idx] == 0) --idx;
  while (idx >= 0) {
    --idx;
    int run = idx;
    while (idx >= 0 && lv[idx] == 0) --idx;
    run -= idx;
    ctr += run_tab[run];
  }
  return ctr;
}
/* encode_mb_aux.cpp:438-451 WelsGetNoneZeroCount_c */
int32_t orc_nonzero_count (const int16_t* lv) { int n = 0; for (int i = 0; i < 16; ++i) n += lv[i] != 0; return n; }

/* codec/encoder/core/src/decode_mb_aux.cpp:139-145 WelsDequant4x4_c: in-place multiply in int16 */
void orc_dequant4x4 (int16_t* r, int qp) { for (int i = 0; i < 16; ++i) r[i] = (int16_t) (r[i] * kDequant[qp * 3 + pc (i)]); }

/* decode_mb_aux.cpp:40-125: qp < 12: WelsIHadamard4x4Dc + WelsDequantLumaDc4x4; else WelsDequantIHadamard4x4_c (mf >> 2)
 * (dispatch as in svc_encode_mb.cpp:99-105) */
void orc_dequant_ihadamard4x4 (int16_t* r, int qp) {
  int16_t t[4];
  for (int i = 0; i < 16; i += 4) {
    t[0] = r[i] + r[i + 2]; t[1] = r[i] - r[i + 2]; t[2] = r[i + 1] - r[i + 3]; t[3] = r[i + 1] + r[i + 3];
    r[i] = t[0] + t[3]; r[i + 1] = t[1] + t[2]; r[i + 2] = t[1] - t[2]; r[i + 3] = t[0] - t[3];
  }
  if (qp < 12) {
    for (int i = 0; i < 4; ++i) {
      t[0] = r[i] + r[i + 8]; t[1] = r[i] - r[i + 8]; t[2] = r[i + 4] - r[i + 12]; t[3] = r[i + 4] + r[i + 12];
      r[i] = t[0] + t[3]; r[i + 4] = t[1] + t[2]; r[i + 8] = t[1] - t[2]; r[i + 12] = t[0] - t[3];
    }
    const int dq = kDequant[(qp % 6) * 3], qf0 = qp / 6, qf1 = 2 - qf0, qf0s = 1 << (1 - qf0);
    for (int i = 0; i < 16; ++i) r[i] = (int16_t) ((r[i] * dq + qf0s) >> qf1);
  } else {
    const int mf = kDequant[qp * 3] >> 2;
    for (int i = 0; i < 4; ++i) {
      t[0] = r[i] + r[i + 8]; t[1] = r[i] - r[i + 8]; t[2] = r[i + 4] - r[i + 12]; t[3] = r[i + 4] + r[i + 12];
      r[i] = (int16_t) ((t[0] + t[3]) * mf); r[i + 4] = (int16_t) ((t[1] + t[2]) * mf);
      r[i + 8] = (int16_t) ((t[1] - t[2]) * mf); r[i + 12] = (int16_t) ((t[0] - t[3]) * mf);
    }
  }
}
/* decode_mb_aux.cpp:127-137 WelsDequantIHadamard2x2Dc */
void orc_dequant_ihadamard2x2_dc (int16_t* d, int qp) {
  const int mf = kDequant[qp * 3];
  const int16_t su = d[0] + d[2], du = d[0] - d[2], sd = d[1] + d[3], dd = d[1] - d[3];
  d[0] = (int16_t) (((su + sd) * mf) >> 1); d[1] = (int16_t) (((su - sd) * mf) >> 1);
  d[2] = (int16_t) (((du + dd) * mf) >> 1); d[3] = (int16_t) (((du - dd) * mf) >> 1);
}
/* decode_mb_aux.cpp:164-199 WelsIDctT4Rec_c: int16 row pass, (x+32)>>6, add prediction, clip */
static uint8_t clip255 (int v) { return (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v); }
void orc_idct4x4_rec (uint8_t* rec, int32_t rs, const uint8_t* pred, int32_t ps, const int16_t* d) {
  int16_t t[16];
  for (int i = 0; i < 4; ++i) {
    const int su = d[i * 4] + d[i * 4 + 2], du = d[i * 4] - d[i * 4 + 2];
    const int sd = d[i * 4 + 1] + (d[i * 4 + 3] >> 1), dd = (d[i * 4 + 1] >> 1) - d[i * 4 + 3];
    t[i * 4] = (int16_t) (su + sd); t[i * 4 + 1] = (int16_t) (du + dd); t[i * 4 + 2] = (int16_t) (du - dd); t[i * 4 + 3] = (int16_t) (su - sd);
  }
  for (int i = 0; i < 4; ++i) {
    const int sl = t[i] + t[8 + i], dl = t[i] - t[8 + i], dr = (t[4 + i] >> 1) - t[12 + i], sr = t[4 + i] + (t[12 + i] >> 1);
    rec[i] = clip255 (pred[i] + ((sl + sr + 32) >> 6));
    rec[rs + i] = clip255 (pred[ps + i] + ((dl + dr + 32) >> 6));
    rec[2 * rs + i] = clip255 (pred[2 * ps + i] + ((dl - dr + 32) >> 6));
    rec[3 * rs + i] = clip255 (pred[3 * ps + i] + ((sl - sr + 32) >> 6));
  }
}
