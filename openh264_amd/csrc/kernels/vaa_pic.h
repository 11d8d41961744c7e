// vaa_pic.h -- the pre-analysis statistics of a source picture against a previous source picture.
//
// Reference: codec/processing/src/vaacalc/vaacalcfuncs.cpp
//   :254-336  VAACalcSad_c        four 8x8 SADs per macroblock (+ the frame's sum)
//   :486-600  VAACalcSadBgd_c     + per 8x8: sum of differences (pSd8x8), largest absolute difference (pMad8x8)
//   :37-147   VAACalcSadSsd_c     + per 16x16: sum, sum of squares of the current picture, sum of squared differences
//   :149-252  VAACalcSadVar_c     + per 16x16: sum, sum of squares
//   :338-484  VAACalcSadSsdBgd_c  all of the above
// selected by CVAACalculation::Process (vaacalculation.cpp:118-157) from iCalcBgd / iCalcSsd / iCalcVar.  Every variant is the same
// walk over the macroblock's 256 sample pairs; which results are stored is the only difference, so here it is ONE function with
// optional outputs.  Both pictures are macroblock-tiled (common/wh_types.h WH_SRC_*): a macroblock's luma is 256 consecutive bytes.
// One THREAD per macroblock: 2 x 16 loads of 16 bytes, ~1.5 k integer operations -- 8160 threads for a 1080p picture, a few
// microseconds; the host adds up the frame SAD from the 8x8 SADs of the macroblocks the reference covers ((w >> 4) x (h >> 4)).
#pragma once
#include "prims.h"

typedef struct WhVaaOut {          // device arrays, [num_mb] macroblocks of the MB-aligned picture; NULL = not wanted
  int32_t* sad8x8;                 // [mb][4]
  int32_t* sd8x8;                  // [mb][4]   pSumOfDiff8x8
  uint8_t* mad8x8;                 // [mb][4]
  int32_t* sum16;                  // [mb]      pSum16x16
  int32_t* sqsum16;                // [mb]      pSumOfSquare16x16
  int32_t* ssd16;                  // [mb]      pSsd16x16
} WhVaaOut;

WH_HDFN void wh_vaa_mb (const WH_G uint8_t* cur, const WH_G uint8_t* ref, int xy, const WhVaaOut& o) {
  const WH_G uint8_t* c = cur + (size_t)xy * WH_SRC_MB_BYTES;
  const WH_G uint8_t* r = ref + (size_t)xy * WH_SRC_MB_BYTES;
  int sad[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0}, mad[4] = {0, 0, 0, 0};
  int sum = 0, sqsum = 0, ssd = 0;
  for (int row = 0; row < 16; ++row) {
    const WhU4 a = wh_ldg16 (c + row * 16), b = wh_ldg16 (r + row * 16);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int blk = (row >> 3) * 2 + (k >> 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = (int) ((aw[k] >> (8 * i)) & 255u), q = (int) ((bw[k] >> (8 * i)) & 255u);
        const int d = p - q, ad = d < 0 ? -d : d;
        sad[blk] += ad; sd[blk] += d; mad[blk] = ad > mad[blk] ? ad : mad[blk];
        sum += p; sqsum += p * p; ssd += d * d;
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (o.sad8x8) ((WH_G int32_t*)o.sad8x8)[xy * 4 + b] = sad[b];
    if (o.sd8x8) ((WH_G int32_t*)o.sd8x8)[xy * 4 + b] = sd[b];
    if (o.mad8x8) ((WH_G uint8_t*)o.mad8x8)[xy * 4 + b] = (uint8_t)mad[b];
  }
  if (o.sum16) ((WH_G int32_t*)o.sum16)[xy] = sum;
  if (o.sqsum16) ((WH_G int32_t*)o.sqsum16)[xy] = sqsum;
  if (o.ssd16) ((WH_G int32_t*)o.ssd16)[xy] = ssd;
}
