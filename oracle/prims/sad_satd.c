/* sad_satd.c -- TEST INFRASTRUCTURE ONLY (see oracle_prims.h). */
#include "oracle_prims.h"
#include <stdlib.h>

static const int kBw[7] = {16, 16, 8, 8, 4, 8, 4};
static const int kBh[7] = {16, 8, 16, 8, 4, 4, 8};

/* codec/common/src/sad_common.cpp:44-120  WelsSampleSad{4x4,...,16x16}_c: sum |a-b| */
int32_t orc_sad (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb) {
  int32_t s = 0;
  for (int y = 0; y < kBh[blk]; ++y, a += sa, b += sb)
    for (int x = 0; x < kBw[blk]; ++x) s += abs ((int)a[x] - (int)b[x]);
  return s;
}

/* codec/common/src/sad_common.cpp:122-165  WelsSampleSadFour*_c: SADs against the reference block
 * displaced by one row up, one row down, one column left, one column right (that order). */
void orc_sad_four (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb, int32_t* out4) {
  out4[0] = orc_sad (blk, a, sa, b - sb, sb);
  out4[1] = orc_sad (blk, a, sa, b + sb, sb);
  out4[2] = orc_sad (blk, a, sa, b - 1, sb);
  out4[3] = orc_sad (blk, a, sa, b + 1, sb);
}

/* codec/encoder/core/src/sample.cpp:47-96  WelsSampleSatd4x4_c: 4x4 Hadamard of the difference,
 * sum of magnitudes, (sum + 1) >> 1 */
static int32_t satd4x4 (const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb) {
  int m[4][4], s = 0;
  for (int y = 0; y < 4; ++y) {
    const int d0 = a[y * sa + 0] - b[y * sb + 0], d1 = a[y * sa + 1] - b[y * sb + 1];
    const int d2 = a[y * sa + 2] - b[y * sb + 2], d3 = a[y * sa + 3] - b[y * sb + 3];
    const int s0 = d0 + d2, s1 = d1 + d3, s2 = d0 - d2, s3 = d1 - d3;
    m[y][0] = s0 + s1; m[y][1] = s2 + s3; m[y][2] = s2 - s3; m[y][3] = s0 - s1;
  }
  for (int x = 0; x < 4; ++x) {
    const int s0 = m[0][x] + m[2][x], s1 = m[1][x] + m[3][x], s2 = m[0][x] - m[2][x], s3 = m[1][x] - m[3][x];
    s += abs (s0 + s1) + abs (s2 + s3) + abs (s2 - s3) + abs (s0 - s1);
  }
  return (s + 1) >> 1;
}

/* codec/encoder/core/src/sample.cpp:97-156  the larger sizes are sums of 4x4 SATDs (each rounded) */
int32_t orc_satd (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb) {
  int32_t s = 0;
  for (int y = 0; y < kBh[blk]; y += 4)
    for (int x = 0; x < kBw[blk]; x += 4) s += satd4x4 (a + y * sa + x, sa, b + y * sb + x, sb);
  return s;
}
