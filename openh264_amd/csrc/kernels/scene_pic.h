// scene_pic.h -- scene-change statistic of a source picture against the previous source picture.
//
// Reference: codec/processing/src/scenechangedetection/SceneChangeDetection.h:112-133 (CSceneChangeDetectorVideo::operator()):
// the number of 8x8 luma blocks whose SAD against the co-located block of the previous source picture exceeds
// HIGH_MOTION_BLOCK_THRESHOLD (320).  The reference runs it on its MB-aligned, zero-padded copy of the source, i.e. on
// exactly the padded source planes this engine keeps in HBM (blk8_w x blk8_h = 2 mb_w x 2 mb_h blocks).  The host turns the count into the
// LARGE_CHANGED_SCENE decision (:229-241) and the frame type (encoder.cpp:377-391).
// One wavefront per 16x16 region: lane = (row, 4-pixel segment), the four block SADs are wave reductions, one atomic
// per region.  HBM-bound: two reads of every luma sample, nothing written but one counter.
#pragma once
#include "prims.h"

WH_FN void wh_scene_mb_body (const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  WH_G const uint8_t* cur = (WH_G const uint8_t*)J.src[0];
  WH_G const uint8_t* prv = (WH_G const uint8_t*)J.prev_src_y;
  int s0, s1, s2, s3;
  // (both pictures macroblock-tiled, WH_SRC_*: lane = (row, 4-sample segment) is the byte order of the luma block)
#define WH_SC_SAD(lane) wh_sad4 (* (WH_G const uint32_t*) (cur + WH_SRC_Y_OFF (P.mb_w, mbx, mby, 0, 0) + (lane) * 4), \
                                 * (WH_G const uint32_t*) (prv + WH_SRC_Y_OFF (P.mb_w, mbx, mby, 0, 0) + (lane) * 4))
  // block (by, bx): rows 8*by.., segments 2*bx..  -> lane bit 5 = by, lane bit 1 = bx
  WV_SUM2 (s0, s1, lane, ((lane) & 0x22) == 0x00 ? WH_SC_SAD (lane) : 0, ((lane) & 0x22) == 0x02 ? WH_SC_SAD (lane) : 0);
  WV_SUM2 (s2, s3, lane, ((lane) & 0x22) == 0x20 ? WH_SC_SAD (lane) : 0, ((lane) & 0x22) == 0x22 ? WH_SC_SAD (lane) : 0);
#undef WH_SC_SAD
  const int bx = mbx * 2, by = mby * 2;
  int n = 0;
  if (by < P.blk8_h) { n += (bx < P.blk8_w && s0 > 320) + (bx + 1 < P.blk8_w && s1 > 320); }
  if (by + 1 < P.blk8_h) { n += (bx < P.blk8_w && s2 > 320) + (bx + 1 < P.blk8_w && s3 > 320); }
  WV_LANES_BEGIN (lane)
  if (lane == 0 && n) wh_atomic_add_u32 ((WH_G uint32_t*)J.scene_count, (uint32_t)n);
  WV_LANES_END
}
