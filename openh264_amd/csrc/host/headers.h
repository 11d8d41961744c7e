// headers.h -- SPS / PPS / slice-header syntax writers (host side).
//
// Field choices reproduce what the reference encoder emits for a single-layer AVC stream:
//   codec/encoder/core/src/au_set.cpp:197-262   WelsWriteVUI
//   codec/encoder/core/src/au_set.cpp:264-334   WelsWriteSpsSyntax      (+ :492-563 WelsInitSps, :51-75 level check)
//   codec/encoder/core/src/au_set.cpp:406-474   WelsWritePpsSyntax      (+ :588-645 WelsInitPps)
//   codec/encoder/core/src/svc_encode_slice.cpp:276-346 WelsSliceHeaderWrite
#pragma once
#include "bitwriter.h"

namespace wh {

struct SpsParams {
  int sps_id = 0;
  int profile_idc = 66;
  int level_idc = 0;             // 0 = derive from Table A-1
  bool constraint_set3 = false;  // level 1b signalling for Baseline
  int width = 0, height = 0;     // actual picture size (cropping is derived)
  int mb_w = 0, mb_h = 0;
  int num_ref_frames = 1;
  bool gaps_in_frame_num = false;
  bool frame_cropping = true;
  float frame_rate = 30.f;
  int bitrate = 0;               // 0 = unspecified
};

struct PpsParams {
  int pps_id = 0, sps_id = 0;
  int chroma_qp_offset = 0;
  bool cabac = false;
};

struct SliceHeaderParams {
  int first_mb = 0;
  int slice_type = 2;            // 2 = I, 0 = P   (values the reference writes)
  int pps_id = 0;
  int frame_num = 0;
  bool idr = true;
  int idr_pic_id = 0;
  int nal_ref_idc = 3;
  int slice_qp = 26;
  int disable_deblocking_idc = 0;
  int alpha_offset = 0, beta_offset = 0;
  int num_ref_idx_active = 1;
  bool num_ref_idx_override = false;
};

// Level selection (au_set.cpp:51-75,187-195): first row of Table A-1 that admits the stream.
int select_level_idc (int mb_w, int mb_h, int num_ref_frames, float frame_rate, int bitrate, bool* is_level_1b);

void write_sps_rbsp (std::vector<uint8_t>& rbsp, const SpsParams& p);
void write_pps_rbsp (std::vector<uint8_t>& rbsp, const PpsParams& p);
void write_slice_header (BitWriter& bw, const SliceHeaderParams& h);

}  // namespace wh
