/* batch.c -- TEST INFRASTRUCTURE ONLY (see oracle_prims.h).
 * Loops of the pinned single-case functions over arrays of cases, so that the GPU tier can compare EVERY computed case of a 10^5 .. 10^6
 * draw (one ctypes call per draw, not per case).  Nothing here restates an algorithm: each entry calls the function of the same name
 * without the _batch suffix (tests/test_oracle_prims.py pins those against the reference's _c functions). */
#include "oracle_prims.h"
#include <string.h>

void orc_sad_batch (int blk, int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = orc_sad (blk, p1 + o1[i], s1, p2 + o2[i], s2);
}
void orc_satd_batch (int blk, int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = orc_satd (blk, p1 + o1[i], s1, p2 + o2[i], s2);
}
void orc_sad_four_batch (int blk, int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int32_t* out) {
  for (int i = 0; i < n; ++i) orc_sad_four (blk, p1 + o1[i], s1, p2 + o2[i], s2, out + 4 * i);
}
void orc_dct4x4_batch (int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int16_t* out) {
  for (int i = 0; i < n; ++i) orc_dct4x4 (out + 16 * i, p1 + o1[i], s1, p2 + o2[i], s2);
}
/* quantisation in place, then both scans, the single-coefficient score and the non-zero count of the DC+AC scan */
void orc_quant_scan_batch (int n, int16_t* io, const uint8_t* qp, int intra, int16_t* mx, int16_t* zz, int16_t* za, int32_t* ctr, int32_t* nz) {
  for (int i = 0; i < n; ++i) {
    mx[i] = (int16_t)orc_quant4x4_max (io + 16 * i, qp[i], intra);
    orc_scan4x4_dcac (zz + 16 * i, io + 16 * i);
    orc_scan4x4_ac (za + 16 * i, io + 16 * i);
    ctr[i] = orc_single_ctr4x4 (zz + 16 * i);
    nz[i] = orc_nonzero_count (zz + 16 * i);
  }
}
void orc_dequant_idct_rec_batch (int n, const int16_t* lev, const uint8_t* qp, const uint8_t* pred, uint8_t* rec, int16_t* deq) {
  for (int i = 0; i < n; ++i) {
    memcpy (deq + 16 * i, lev + 16 * i, 32);
    orc_dequant4x4 (deq + 16 * i, qp[i]);
    orc_idct4x4_rec (rec + 16 * i, 4, pred + 16 * i, 4, deq + 16 * i);
  }
}
