// expand_pic.h -- replicate the reconstructed picture's border (32 luma / 16 chroma pixels) so that
// motion vectors may point outside the picture.
//
// Reference: codec/common/src/expand_pic.cpp:271-350 ExpandPictureLuma_c / ExpandPictureChroma_c,
// called from ExpandReferencingPicture (:388-415) after deblocking (ref_list_mgr_svc.cpp:375).
// One wavefront per padded row; every output pixel is read from its clamped source coordinate, so
// rows are independent (no ordering between the corner / top / side regions as in the C code).
#pragma once
#include "prims.h"

WH_HDFN int wh_expand_num_blocks (const WhSeqParams& P) {
  return (P.mb_h * 16 + 64) + 2 * (P.mb_h * 8 + 32);
}

WH_FN void wh_expand_body (const WhSeqParams& P, const WhPicJob& J, int blk) {
  const int lh = P.mb_h * 16 + 64, ch = P.mb_h * 8 + 32;
  int pl, row, w, h, pad, stride;
  if (blk < lh) { pl = 0; row = blk - 32; w = P.mb_w * 16; h = P.mb_h * 16; pad = 32; stride = P.rec_stride_y; }
  else { const int b = blk - lh; pl = 1 + b / ch; row = b % ch - 16; w = P.mb_w * 8; h = P.mb_h * 8; pad = 16; stride = P.rec_stride_c; }
  WH_G uint8_t* base = (WH_G uint8_t*)J.rec[pl];
  const int sy = row < 0 ? 0 : (row >= h ? h - 1 : row);
  const WH_G uint8_t* srow = base + (ptrdiff_t)sy * stride;
  WH_G uint8_t* drow = base + (ptrdiff_t)row * stride;
  WV_LANES_BEGIN (lane)
  if (row >= 0 && row < h) {
    if (lane < pad) drow[-pad + lane] = srow[0];
    else if (lane < 2 * pad) drow[w + lane - pad] = srow[w - 1];
  } else {
    for (int x = -pad + lane; x < w + pad; x += 64) drow[x] = srow[x < 0 ? 0 : (x >= w ? w - 1 : x)];
  }
  WV_LANES_END
}
