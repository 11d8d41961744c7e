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

// `c` / `r`: the macroblock's first sample in the two pictures, `pitch` bytes from one of its rows to the next (16 in a tiled picture)
template <bool ALIGNED>
WH_HDFN void wh_vaa_mb_at (const WH_G uint8_t* c, const WH_G uint8_t* r, size_t pitch, int xy, const WhVaaOut& o) {
  int sad[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0}, mad[4] = {0, 0, 0, 0};
  int sum = 0, sqsum = 0, ssd = 0;
  for (int row = 0; row < 16; ++row) {
    uint32_t aw[4], bw[4];
    if (ALIGNED) {
      const WhU4 a = wh_ldg16 (c + row * pitch), b = wh_ldg16 (r + row * pitch);
      aw[0] = a.x; aw[1] = a.y; aw[2] = a.z; aw[3] = a.w; bw[0] = b.x; bw[1] = b.y; bw[2] = b.z; bw[3] = b.w;
    } else {                                   // any byte address
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const WH_G uint8_t* pc = c + row * pitch + 4 * k;
        const WH_G uint8_t* pr = r + row * pitch + 4 * k;
        aw[k] = (uint32_t)pc[0] | ((uint32_t)pc[1] << 8) | ((uint32_t)pc[2] << 16) | ((uint32_t)pc[3] << 24);
        bw[k] = (uint32_t)pr[0] | ((uint32_t)pr[1] << 8) | ((uint32_t)pr[2] << 16) | ((uint32_t)pr[3] << 24);
      }
    }
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
WH_HDFN void wh_vaa_mb (const WH_G uint8_t* cur, const WH_G uint8_t* ref, int xy, const WhVaaOut& o) {
  wh_vaa_mb_at<true> (cur + (size_t)xy * WH_SRC_MB_BYTES, ref + (size_t)xy * WH_SRC_MB_BYTES, 16, xy, o);
}
// A picture whose width is no multiple of 16, exactly as the C functions walk it: they step from one macroblock row to the next by
// 16 * stride - width (vaacalcfuncs.cpp:46,145-146 and the same lines of the other four variants), so row i begins (width & 15) * i
// samples to the LEFT of the picture's column 0 -- in the stride padding of the line before, and further in that line's samples.  The
// two luma planes are here as the caller has them (`stride` bytes per line, padding bytes included); macroblock (j, i) of the
// (width >> 4) x (height >> 4) the functions cover; results at the MB-aligned picture's index i * mb_w + j like the tiled walk's.
WH_HDFN void wh_vaa_mb_skewed (const WH_G uint8_t* cur, const WH_G uint8_t* ref, int stride, int width, int mb_w, int j, int i, const WhVaaOut& o) {
  const size_t off = (size_t)i * (size_t) (16 * stride - (width & 15)) + (size_t)j * 16;
  wh_vaa_mb_at<false> (cur + off, ref + off, (size_t)stride, i * mb_w + j, o);
}
