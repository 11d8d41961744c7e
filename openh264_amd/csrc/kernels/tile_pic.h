// tile_pic.h -- the tiled twin of a reference picture (common/wh_types.h, WH_TILE_*): after deblocking and border
// expansion (the point where the reference encoder calls ExpandReferencingPicture, ref_list_mgr_svc.cpp:375) the whole
// expanded picture is copied once into 128-byte tiles -- luma 16 x 8, chroma 8 x 8 with Cb | Cr side by side in every row --
// from which the P kernel fetches its search windows (inter_mb.h, wh_win_*).  Device-private: the planar picture stays what
// every other pass and the host read.
//
// One item = one 16-byte row of one tile, items in tile order: a wavefront reads 8 picture rows x 128 contiguous bytes and
// writes 1 KB contiguously -- a plain streaming pass (3.4 MB read + written per 1080p picture).
#pragma once
#include "prims.h"

WH_HDFN int wh_tile_items_y (const WhSeqParams& P) { return (P.rec_stride_y >> 4) * ((P.mb_h * 16 + 64) >> 3) * 8; }
WH_HDFN int wh_tile_items_c (const WhSeqParams& P) { return (P.rec_stride_c >> 3) * ((P.mb_h * 8 + 32) >> 3) * 8; }
WH_HDFN int wh_tile_items (const WhSeqParams& P) { return wh_tile_items_y (P) + wh_tile_items_c (P); }

// one 16-byte row of luma tile (tx, ty) / chroma tile (tx, ty) from the planar picture
WH_HDFN void wh_tile_copy_y (const WhSeqParams& P, const WhPicJob& J, int tx, int ty, int row) {
  const int tw = P.rec_stride_y >> 4;
  const WH_G uint8_t* src = (const WH_G uint8_t*)J.rec[0] - (ptrdiff_t)32 * P.rec_stride_y - 32 + (ptrdiff_t) (ty * 8 + row) * P.rec_stride_y + tx * 16;
  wh_stg16 ((WH_G uint8_t*)J.rec_tiles[0] + ((size_t) (ty * tw + tx) * 8 + row) * 16, wh_ldg16 (src));
}
WH_HDFN void wh_tile_copy_c (const WhSeqParams& P, const WhPicJob& J, int tx, int ty, int row) {
  const int tw = P.rec_stride_c >> 3;
  const ptrdiff_t off = - (ptrdiff_t)16 * P.rec_stride_c - 16 + (ptrdiff_t) (ty * 8 + row) * P.rec_stride_c + tx * 8;
  const WH_G uint32_t* cb = (const WH_G uint32_t*) ((const WH_G uint8_t*)J.rec[1] + off);
  const WH_G uint32_t* cr = (const WH_G uint32_t*) ((const WH_G uint8_t*)J.rec[2] + off);
  WhU4 v;
  v.x = cb[0]; v.y = cb[1]; v.z = cr[0]; v.w = cr[1];
  wh_stg16 ((WH_G uint8_t*)J.rec_tiles[1] + ((size_t) (ty * tw + tx) * 8 + row) * 16, v);
}
// the whole expanded picture: pictures that are not deblocked (mode decision wrote the planar picture itself)
WH_HDFN void wh_tile_item (const WhSeqParams& P, const WhPicJob& J, int idx) {
  const int ny = wh_tile_items_y (P);
  if (idx < ny) { const int tw = P.rec_stride_y >> 4, tile = idx >> 3; wh_tile_copy_y (P, J, tile % tw, tile / tw, idx & 7); return; }
  idx -= ny;
  const int tw = P.rec_stride_c >> 3, tile = idx >> 3;
  wh_tile_copy_c (P, J, tile % tw, tile / tw, idx & 7);
}
// A picture that went through the deblocking pass (WhPicJob::rec_blk: every one that is filtered at all) has the tiles INSIDE the picture already --
// that pass writes each sample to both layouts (deblock_mb.h, round 6) -- and only the BORDER tiles are copied here, 9 % of a 1080p picture:
// the tile rows above the picture, those below it, and left and right of it the tile columns outside [2, 2 + mb_w).  Luma tile (tx, ty) starts at
// sample (16 tx - 32, 8 ty - 32): the picture is tile rows [4, 4 + 2 mb_h); chroma tile (tx, ty) at (8 tx - 16, 8 ty - 16): tile rows [2, 2 + mb_h).
WH_HDFN int wh_tile_border_tiles (int tw, int th, int in_w, int in_h) { return tw * (th - in_h) + in_h * (tw - in_w); }
WH_HDFN int wh_tile_border_items_y (const WhSeqParams& P) { return 8 * wh_tile_border_tiles (P.rec_stride_y >> 4, (P.mb_h * 16 + 64) >> 3, P.mb_w, 2 * P.mb_h); }
WH_HDFN int wh_tile_border_items_c (const WhSeqParams& P) { return 8 * wh_tile_border_tiles (P.rec_stride_c >> 3, (P.mb_h * 8 + 32) >> 3, P.mb_w, P.mb_h); }
WH_HDFN int wh_tile_border_items (const WhSeqParams& P) { return wh_tile_border_items_y (P) + wh_tile_border_items_c (P); }
// the t-th border tile of a tw x th grid whose inside is columns [2, 2 + in_w) x rows [top, top + in_h)
WH_HDFN void wh_tile_border_tile (int t, int tw, int th, int top, int in_w, int in_h, int* tx, int* ty) {
  const int above = top * tw, below = (th - top - in_h) * tw;
  if (t < above) { *ty = t / tw; *tx = t - *ty * tw; return; }
  t -= above;
  if (t < below) { const int r = t / tw; *ty = top + in_h + r; *tx = t - r * tw; return; }
  t -= below;
  const int side = tw - in_w, r = t / side, c = t - r * side;
  *ty = top + r; *tx = c < 2 ? c : c + in_w;
}
WH_HDFN void wh_tile_border_item (const WhSeqParams& P, const WhPicJob& J, int idx) {
  const int ny = wh_tile_border_items_y (P);
  int tx, ty;
  if (idx < ny) { wh_tile_border_tile (idx >> 3, P.rec_stride_y >> 4, (P.mb_h * 16 + 64) >> 3, 4, P.mb_w, 2 * P.mb_h, &tx, &ty); wh_tile_copy_y (P, J, tx, ty, idx & 7); return; }
  idx -= ny;
  wh_tile_border_tile (idx >> 3, P.rec_stride_c >> 3, (P.mb_h * 8 + 32) >> 3, 2, P.mb_w, P.mb_h, &tx, &ty);
  wh_tile_copy_c (P, J, tx, ty, idx & 7);
}

// ---- source pictures: planar I420 (as uploaded: tight strides src_stride_y / src_stride_c) -> macroblock tiles (WH_SRC_*) ------------
// One item = 16 bytes of the tiled picture: luma row r of MB xy (items 0..15), two rows of its Cb (16..19) or Cr block (20..23).
WH_HDFN int wh_src_tile_items (const WhSeqParams& P) { return P.mb_w * P.mb_h * 24; }
WH_HDFN void wh_src_tile_item (const WhSeqParams& P, const WH_G uint8_t* planar, WH_G uint8_t* tiled, int idx) {
  const int xy = idx / 24, i = idx - xy * 24, mbx = xy % P.mb_w, mby = xy / P.mb_w;
  const size_t ysz = (size_t)P.src_stride_y * P.mb_h * 16, csz = (size_t)P.src_stride_c * P.mb_h * 8;
  WhU4 v;
  if (i < 16) v = wh_ldg16 (planar + (size_t) (mby * 16 + i) * P.src_stride_y + mbx * 16);
  else {
    const int pl = (i - 16) >> 2, r = ((i - 16) & 3) * 2;
    const WH_G uint32_t* a = (const WH_G uint32_t*) (planar + ysz + pl * csz + (size_t) (mby * 8 + r) * P.src_stride_c + mbx * 8);
    const WH_G uint32_t* b = (const WH_G uint32_t*) ((const WH_G uint8_t*)a + P.src_stride_c);
    v.x = a[0]; v.y = a[1]; v.z = b[0]; v.w = b[1];
  }
  wh_stg16 (tiled + (size_t)idx * 16, v);
}
