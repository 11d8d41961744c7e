// headers.cpp -- see headers.h
#include "headers.h"
#define WH_TABLE static const
#include "../common/h264_tables.h"

namespace wh {

int select_level_idc (int mb_w, int mb_h, int num_ref_frames, float frame_rate, int bitrate, bool* is_level_1b) {
  const uint32_t pic_mbs = (uint32_t) (mb_w * mb_h);
  int level = 51;
  for (int i = 0; i < 17; ++i) {
    const uint32_t* L = &kWhLevelLimits[i * 6];   // idc, MaxMBPS, MaxFS, MaxDpbMbs, MaxBR, MaxCPB
    if (L[1] < (uint32_t) (pic_mbs * frame_rate)) continue;
    if (L[2] < pic_mbs) continue;
    if ((L[2] << 3) < (uint32_t) (mb_w * mb_w)) continue;
    if ((L[2] << 3) < (uint32_t) (mb_h * mb_h)) continue;
    if (L[3] < (uint32_t)num_ref_frames * pic_mbs) continue;
    if (bitrate != 0 && (int32_t) (L[4] * 1200) < bitrate) continue;
    level = (int)L[0];
    break;
  }
  *is_level_1b = false;
  if (level == 9) { level = 11; *is_level_1b = true; }   // Baseline signals 1b as level 11 + constraint_set3
  return level;
}

void write_sps_rbsp (std::vector<uint8_t>& rbsp, const SpsParams& p) {
  BitWriter bw (&rbsp);
  bw.put (8, (uint32_t)p.profile_idc);
  bw.bit (p.profile_idc == 66);          // constraint_set0
  bw.bit (p.profile_idc <= 77);          // constraint_set1
  bw.bit (0);                            // constraint_set2
  bw.bit (p.constraint_set3);
  bw.put (4, 0);
  bw.put (8, (uint32_t)p.level_idc);
  bw.ue ((uint32_t)p.sps_id);
  bw.ue (15 - 4);                        // log2_max_frame_num_minus4
  bw.ue (2);                             // pic_order_cnt_type
  bw.ue ((uint32_t)p.num_ref_frames);
  bw.bit (p.gaps_in_frame_num);
  bw.ue ((uint32_t) (p.mb_w - 1));
  bw.ue ((uint32_t) (p.mb_h - 1));
  bw.bit (1);                            // frame_mbs_only
  bw.bit (p.level_idc >= 30);            // direct_8x8_inference
  const int aw = p.width & ~1, ah = p.height & ~1;
  const int crop_r = (p.mb_w * 16 - aw) / 2, crop_b = (p.mb_h * 16 - ah) / 2;
  const bool crop = p.frame_cropping && (crop_r > 0 || crop_b > 0);
  bw.bit (crop);
  if (crop) { bw.ue (0); bw.ue ((uint32_t)crop_r); bw.ue (0); bw.ue ((uint32_t)crop_b); }
  bw.bit (1);                            // vui_parameters_present
  bw.bit (0);                            // aspect_ratio_info_present
  bw.bit (0);                            // overscan_info_present
  bw.bit (0);                            // video_signal_type_present
  bw.bit (0);                            // chroma_loc_info_present
  bw.bit (0);                            // timing_info_present
  bw.bit (0);                            // nal_hrd_parameters_present
  bw.bit (0);                            // vcl_hrd_parameters_present
  bw.bit (0);                            // pic_struct_present
  bw.bit (1);                            // bitstream_restriction
  bw.bit (1);                            // motion_vectors_over_pic_boundaries
  bw.ue (0); bw.ue (0);                  // max_bytes_per_pic_denom, max_bits_per_mb_denom
  bw.ue (16); bw.ue (16);                // log2_max_mv_length_horizontal / vertical
  bw.ue (0);                             // max_num_reorder_frames
  bw.ue ((uint32_t)p.num_ref_frames);    // max_dec_frame_buffering
  bw.trailing();
}

void write_pps_rbsp (std::vector<uint8_t>& rbsp, const PpsParams& p) {
  BitWriter bw (&rbsp);
  bw.ue ((uint32_t)p.pps_id);
  bw.ue ((uint32_t)p.sps_id);
  bw.bit (p.cabac);
  bw.bit (0);                            // bottom_field_pic_order_in_frame_present
  bw.ue (0);                             // num_slice_groups_minus1
  bw.ue (0); bw.ue (0);                  // num_ref_idx_l0/l1_default_active_minus1
  bw.bit (0); bw.put (2, 0);             // weighted_pred, weighted_bipred_idc
  bw.se (0); bw.se (0);                  // pic_init_qp/qs - 26
  bw.se (p.chroma_qp_offset);
  bw.bit (1);                            // deblocking_filter_control_present
  bw.bit (0);                            // constrained_intra_pred
  bw.bit (0);                            // redundant_pic_cnt_present
  bw.trailing();
}

void write_slice_header (BitWriter& bw, const SliceHeaderParams& h) {
  bw.ue ((uint32_t)h.first_mb);
  bw.ue ((uint32_t)h.slice_type);
  bw.ue ((uint32_t)h.pps_id);
  bw.put (15, (uint32_t)h.frame_num);
  if (h.idr) bw.ue ((uint32_t)h.idr_pic_id);
  if (h.slice_type == 0) {
    bw.bit (h.num_ref_idx_override);
    if (h.num_ref_idx_override) bw.ue ((uint32_t) (h.num_ref_idx_active - 1));
  }
  if (!h.idr && h.slice_type == 0) {
    bw.bit (1);                          // ref_pic_list_modification_flag_l0
    bw.ue (0); bw.ue (0);                // modification_of_pic_nums_idc 0, abs_diff_pic_num_minus1 0
    bw.ue (3);
  }
  if (h.nal_ref_idc) {
    if (h.idr) { bw.bit (0); bw.bit (0); }   // no_output_of_prior_pics, long_term_reference
    else bw.bit (0);                         // adaptive_ref_pic_marking_mode
  }
  bw.se (h.slice_qp - 26);
  bw.ue ((uint32_t) (h.disable_deblocking_idc == 2 ? 2 : (h.disable_deblocking_idc == 1 ? 1 : 0)));
  if (h.disable_deblocking_idc != 1) { bw.se (h.alpha_offset >> 1); bw.se (h.beta_offset >> 1); }
}

}  // namespace wh
