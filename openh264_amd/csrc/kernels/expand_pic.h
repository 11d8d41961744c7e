// expand_pic.h -- replicate the reconstructed picture's border (32 luma / 16 chroma pixels) so that
// motion vectors may point outside the picture.
//
// Reference: codec/common/src/expand_pic.cpp:271-350 ExpandPictureLuma_c / ExpandPictureChroma_c,
// called from ExpandReferencingPicture (:388-415) after deblocking (ref_list_mgr_svc.cpp:375).
//
// The work is a flat list of 32-bit store items per picture, each computed from its clamped source coordinate (so there is
// no ordering between the corner / top / side regions as in the C code): for every plane first the side items (row r,
// dword j of the left or right margin), then the top / bottom items (margin row, dword of the padded width).  A plain
// streaming pass: ~0.3 MB written per 1080p picture.
#pragma once
#include "prims.h"

WH_HDFN void wh_expand_plane_geom (const WhSeqParams& P, int pl, int* w, int* h, int* pad, int* stride) {
  if (pl == 0) { *w = P.mb_w * 16; *h = P.mb_h * 16; *pad = 32; *stride = P.rec_stride_y; }
  else { *w = P.mb_w * 8; *h = P.mb_h * 8; *pad = 16; *stride = P.rec_stride_c; }
}
WH_HDFN int wh_expand_plane_items (int w, int h, int pad) { return h * (pad >> 1) + 2 * pad * ((w + 2 * pad) >> 2); }
WH_HDFN int wh_expand_items (const WhSeqParams& P) {
  return wh_expand_plane_items (P.mb_w * 16, P.mb_h * 16, 32) + 2 * wh_expand_plane_items (P.mb_w * 8, P.mb_h * 8, 16);
}

// item `idx` of the picture whose planes (pixel (0,0)) are rec0..rec2
WH_HDFN void wh_expand_item (const WhSeqParams& P, WH_G uint8_t* rec0, WH_G uint8_t* rec1, WH_G uint8_t* rec2, int idx) {
  int pl = 0, w, h, pad, stride;
  wh_expand_plane_geom (P, 0, &w, &h, &pad, &stride);
  const int n0 = wh_expand_plane_items (w, h, pad);
  if (idx >= n0) {
    idx -= n0;
    wh_expand_plane_geom (P, 1, &w, &h, &pad, &stride);
    const int n1 = wh_expand_plane_items (w, h, pad);
    pl = 1;
    if (idx >= n1) { idx -= n1; pl = 2; }
  }
  WH_G uint8_t* base = pl == 0 ? rec0 : pl == 1 ? rec1 : rec2;
  const int side_items = h * (pad >> 1), per_side = pad >> 2;
  if (idx < side_items) {
    const int r = idx / (2 * per_side), k = idx - r * 2 * per_side, right = k >= per_side, j = k - right * per_side;
    const uint32_t v = 0x01010101u * base[(ptrdiff_t)r * stride + (right ? w - 1 : 0)];
    * (WH_G uint32_t*) (base + (ptrdiff_t)r * stride + (right ? w + 4 * j : -pad + 4 * j)) = v;
    return;
  }
  idx -= side_items;
  const int per_row = (w + 2 * pad) >> 2;
  const int r = idx / per_row, x = -pad + 4 * (idx - r * per_row);
  const bool top = r < pad;
  const WH_G uint8_t* srow = base + (ptrdiff_t) (top ? 0 : h - 1) * stride;
  const uint32_t v = x < 0 ? 0x01010101u * srow[0] : x >= w ? 0x01010101u * srow[w - 1] : * (const WH_G uint32_t*) (srow + x);
  * (WH_G uint32_t*) (base + (ptrdiff_t) (top ? r - pad : h + r - pad) * stride + x) = v;
}
