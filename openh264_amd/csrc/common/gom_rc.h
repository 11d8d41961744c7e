// gom_rc.h -- the QP recursion of rate control with one slice per picture ("GOM-level QP"), shared by the host (initial state of a
// picture) and the P-picture kernel (which advances it at the end of every group of macroblocks): SURVEY 8(f) 3.
//
// Reference behaviour restated (codec/encoder/core/src/ratectl.cpp):
//   :711-745    RcGomTargetBits        bits the coming group may spend, from the bits left and the pre-analysis' SADs per group
//   :748-775    RcCalculateGomQp       +-1 / +-2 on the slice's QP from how the bits spent compare with the targets so far
//   :1239-1262  WelsRcMbInitGom        at the first macroblock of a group: RcCalculateGomQp (not for the first group), RcGomTargetBits
//   :1264-1278  WelsRcMbInfoUpdateGom  after every macroblock: iFrameBitsSlice / iGomBitsSlice += the bits it took
// A group (GOM) is iNumberMbGom consecutive macroblocks in coding order -- whole macroblock rows whenever the picture width is
// a multiple of 16 (RcInitSequenceParameter, ratectl.cpp:153).
#pragma once
#include <stdint.h>
#include "wh_types.h"
#if defined(__HIPCC__)
#define WH_RC_FN static __host__ __device__ inline
#else
#define WH_RC_FN static inline
#endif

WH_RC_FN int32_t wh_div_round (int64_t x, int64_t y) { return (int32_t) (y == 0 ? x / (y + 1) : (y / 2 + x) / y); }      // WELS_DIV_ROUND(64)

// RcGomTargetBits for group `index` with the bits spent so far in R.frame_bits
WH_RC_FN void wh_gom_target_bits (WhGomRc& R) {
  const int32_t last = R.end_mb / R.n_gom_mb;
  const int32_t left = R.target_bits - R.frame_bits;
  if (left <= 0) { R.gom_target = 0; return; }
  int32_t alloc;
  if (R.index >= last) alloc = left;
  else {
    int32_t sum = 0;
    for (int32_t i = R.index + 1; i <= last; ++i) sum += R.gom_sad[i];
    if (sum == 0) alloc = wh_div_round (left, last - R.index);
    else alloc = wh_div_round ((int64_t)left * R.gom_sad[R.index + 1], sum);
  }
  R.gom_target = alloc;
}

// RcCalculateGomQp at the first macroblock of a group that is not the slice's first
WH_RC_FN void wh_gom_calculate_qp (WhGomRc& R) {
  const int64_t left = (int64_t)R.target_bits - R.frame_bits;
  const int64_t target_left = left + R.gom_bits - R.gom_target;
  if (left <= 0 || target_left <= 0) R.calc_qp += 2;
  else {
    const int64_t ratio = 10000 * left / (target_left + 1);
    if (ratio < 8409) R.calc_qp += 2;
    else if (ratio < 9439) R.calc_qp += 1;
    else if (ratio > 10600) R.calc_qp -= 1;
    else if (ratio > 11900) R.calc_qp -= 2;          // (unreachable after the line above, as in the reference)
  }
  R.calc_qp = R.calc_qp < R.min_qp ? R.min_qp : R.calc_qp > R.max_qp ? R.max_qp : R.calc_qp;
  R.gom_bits = 0;
}

// start of a picture: the state WelsRcMbInitGom leaves at the slice's first macroblock
WH_RC_FN void wh_gom_begin (WhGomRc& R, int32_t global_qp) {
  R.calc_qp = global_qp; R.frame_bits = 0; R.gom_bits = 0; R.index = 0; R.skip_run = 0; R.last_qp = R.slice_qp;
  wh_gom_target_bits (R);
}
// a group is complete and `bits` is what its macroblocks took: the state for the next group's first macroblock
WH_RC_FN void wh_gom_next (WhGomRc& R, int32_t bits) {
  R.frame_bits += bits; R.gom_bits += bits;
  ++R.index;
  wh_gom_calculate_qp (R);
  wh_gom_target_bits (R);
}
