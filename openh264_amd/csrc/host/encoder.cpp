// encoder.cpp -- the host-side encoder session (mirror of ISVCEncoder for the hot path).
//
// What stays on the host here is what north_star keeps on the host in the reference as well:
// parameter handling, frame-type decision, SPS/PPS/slice headers, CAVLC and NAL packing
// (codec/encoder/plus/src/welsEncoderExt.cpp:175-500, codec/encoder/core/src/encoder_ext.cpp:3441-3960
// WelsEncoderEncodeExt).  Everything per-macroblock runs on the device through wh::Backend.
#include <string.h>
#include <string>
#include <vector>
#include "../../../include/welship.h"
#include "backend.h"
#include "entropy_cavlc.h"
#include "headers.h"

namespace wh {
Backend* create_default_backend (int device, const char** err);   // provided by the HIP lib or the test build
}

static thread_local std::string g_last_error;
static void set_err (const std::string& s) { g_last_error = s; }

namespace {

inline int align_up (int v, int a) { return (v + a - 1) / a * a; }

struct DevPicture {            // one padded reconstruction buffer + its MB state
  uint8_t* base = nullptr;     // allocation
  uint8_t* plane[3] = {nullptr, nullptr, nullptr};   // pixel (0,0)
  WhMbState* mbs = nullptr;
  bool is_p = false;
};

}  // namespace

struct WelsHipEncoder {
  wh::Backend* be = nullptr;
  bool inited = false;
  WelsHipEncParam prm;
  WhSeqParams seq;
  int mb_w = 0, mb_h = 0, num_mb = 0;
  // device buffers
  uint8_t* d_src = nullptr;          // Y | U | V, MB-aligned dims, tight strides
  DevPicture pic[2];
  int cur = 0;
  WhMbRecord* d_records = nullptr;
  WhPicJob* d_job = nullptr;
  size_t rec_alloc_bytes = 0;
  // host staging
  std::vector<uint8_t> h_src;
  std::vector<WhMbRecord> h_records;
  std::vector<uint8_t> bs;           // output bitstream of the current frame
  std::vector<int32_t> nal_len;
  // stream state
  int frame_index = 0;               // frames since the last IDR
  int frame_num = 0;
  int idr_pic_id = 0;
  int sps_counter = 0, pps_counter = 0;   // INCREASING_ID strategy
  int sps_id_in_bs = 0, pps_id_in_bs = 0;
  bool force_idr = false;
  int level_idc = 0;
  bool level_1b = false;
  bool have_recon = false;
};

static int compute_slices (WelsHipEncoder* e) {
  WhSeqParams& s = e->seq;
  const int n = e->prm.uiSliceMode == 0 ? 1 : e->prm.uiSliceNum;
  if (n < 1 || n > WH_MAX_SLICES) return -1;
  s.num_slices = n;
  if (n == 1) { s.slice_first_mb[0] = 0; s.slice_first_mb[1] = e->num_mb; return 0; }
  // SM_FIXEDSLCNUM_SLICE (svc_enc_slice_segment.cpp:118-190 AssignMbMapMultipleSlices + CheckFixedSliceNumMultiSliceSetting):
  // whole MB rows per slice, rows split as evenly as the reference does.
  const int rows = e->mb_h;
  if (n > rows) return -1;
  int first = 0;
  for (int i = 0; i < n; ++i) {
    s.slice_first_mb[i] = first;
    int r = rows / n;
    if (i == n - 1) r = rows - (rows / n) * (n - 1);
    first += r * e->mb_w;
  }
  s.slice_first_mb[n] = e->num_mb;
  return 0;
}

extern "C" {

const char* WelsHipGetLastError (void) { return g_last_error.c_str(); }

int WelsHipCreateEncoder (WelsHipEncoder** pp) {
  if (!pp) return WELSHIP_ERR_INIT_PARA;
  *pp = new WelsHipEncoder();
  memset (& (*pp)->prm, 0, sizeof (WelsHipEncParam));
  return WELSHIP_OK;
}

int WelsHipUninitialize (WelsHipEncoder* e) {
  if (!e) return WELSHIP_ERR_INIT_PARA;
  if (e->be) {
    e->be->sync();
    if (e->d_src) e->be->free (e->d_src);
    for (int i = 0; i < 2; ++i) { if (e->pic[i].base) e->be->free (e->pic[i].base); if (e->pic[i].mbs) e->be->free (e->pic[i].mbs); e->pic[i] = DevPicture(); }
    if (e->d_records) e->be->free (e->d_records);
    if (e->d_job) e->be->free (e->d_job);
    e->d_src = nullptr; e->d_records = nullptr; e->d_job = nullptr;
    delete e->be;
    e->be = nullptr;
  }
  e->inited = false;
  return WELSHIP_OK;
}

void WelsHipDestroyEncoder (WelsHipEncoder* e) {
  if (!e) return;
  WelsHipUninitialize (e);
  delete e;
}

int WelsHipGetDefaultParams (WelsHipEncoder* e, WelsHipEncParam* p) {
  if (!e || !p) return WELSHIP_ERR_INIT_PARA;
  // the defaults of SWelsSvcCodingParam::FillDefault (param_svc.h:132-211) restricted to what we support
  memset (p, 0, sizeof (*p));
  p->iUsageType = 0; p->iRCMode = -1; p->fMaxFrameRate = 60.f;   // MAX_FRAME_RATE
  p->iTemporalLayerNum = 1; p->iSpatialLayerNum = 1; p->iComplexityMode = 0;
  p->uiIntraPeriod = 0; p->eSpsPpsIdStrategy = 1; p->iEntropyCodingModeFlag = 0;
  p->iLoopFilterDisableIdc = 0; p->bEnableFrameCroppingFlag = 1; p->iDLayerQp = 26;   // SVC_QUALITY_BASE_QP
  p->uiSliceMode = 0; p->uiSliceNum = 1;
  return WELSHIP_OK;
}

int WelsHipInitializeExt (WelsHipEncoder* e, const WelsHipEncParam* p) {
  if (!e || !p) return WELSHIP_ERR_INIT_PARA;
  if (e->inited) WelsHipUninitialize (e);
  // ---- validation: same spirit as ParamValidationExt (encoder_ext.cpp:403-680) ----
  if (p->iPicWidth < 16 || p->iPicHeight < 16 || p->iPicWidth > 4096 || p->iPicHeight > 2304) { set_err ("invalid picture size"); return WELSHIP_ERR_INIT_PARA; }
  if (p->fMaxFrameRate <= 0.f) { set_err ("invalid frame rate"); return WELSHIP_ERR_INIT_PARA; }
  if (p->iDLayerQp < 0 || p->iDLayerQp > 51) { set_err ("invalid QP"); return WELSHIP_ERR_INIT_PARA; }
  if (p->iUsageType != 0) { set_err ("only CAMERA_VIDEO_REAL_TIME is supported"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->iRCMode != -1) { set_err ("only RC_OFF_MODE (-1) is supported"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->iTemporalLayerNum != 1 || p->iSpatialLayerNum != 1) { set_err ("only one temporal and one spatial layer supported"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->iEntropyCodingModeFlag != 0) { set_err ("CABAC is not implemented"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->bEnableAdaptiveQuant || p->bEnableBackgroundDetection || p->bEnableSceneChangeDetect || p->bEnableLongTermReference ||
      p->bEnableDenoise || p->bEnableFrameSkip) { set_err ("AQ/BGD/scene-change/LTR/denoise/frame-skip are not supported"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->uiSliceMode != 0 && p->uiSliceMode != 1) { set_err ("slice mode must be 0 or 1"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->iLoopFilterDisableIdc < 0 || p->iLoopFilterDisableIdc > 2) { set_err ("deblocking idc must be 0..2"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->eSpsPpsIdStrategy != 0 && p->eSpsPpsIdStrategy != 1) { set_err ("SpsPpsIdStrategy must be 0 or 1"); return WELSHIP_ERR_UNSUPPORTED; }
  if (p->iComplexityMode < 0 || p->iComplexityMode > 2) { set_err ("invalid complexity mode"); return WELSHIP_ERR_INIT_PARA; }

  const char* berr = nullptr;
  e->be = wh::create_default_backend (p->iDevice, &berr);
  if (!e->be) { set_err (std::string ("no usable device backend: ") + (berr ? berr : "?")); return WELSHIP_ERR_NO_DEVICE; }
  e->prm = *p;
  e->mb_w = (p->iPicWidth + 15) >> 4;
  e->mb_h = (p->iPicHeight + 15) >> 4;
  e->num_mb = e->mb_w * e->mb_h;
  WhSeqParams& s = e->seq;
  memset (&s, 0, sizeof (s));
  s.mb_w = e->mb_w; s.mb_h = e->mb_h;
  s.src_stride_y = e->mb_w * 16; s.src_stride_c = e->mb_w * 8;
  s.rec_stride_y = align_up (e->mb_w * 16 + 64, 64); s.rec_stride_c = s.rec_stride_y / 2;
  s.complexity = p->iComplexityMode;
  s.chroma_qp_offset = 0;
  s.deblock_idc = p->iLoopFilterDisableIdc;
  s.alpha_offset = p->iLoopFilterAlphaC0Offset; s.beta_offset = p->iLoopFilterBetaOffset;
  s.mv_range = 64;
  if (compute_slices (e)) { set_err ("invalid slice number"); delete e->be; e->be = nullptr; return WELSHIP_ERR_INIT_PARA; }

  const size_t ysz = (size_t)s.src_stride_y * e->mb_h * 16, csz = (size_t)s.src_stride_c * e->mb_h * 8;
  e->h_src.assign (ysz + 2 * csz, 0);
  memset (e->h_src.data() + ysz, 0x80, 2 * csz);   // CWelsPreProcess::Padding: luma 0, chroma 0x80
  e->d_src = (uint8_t*)e->be->alloc (ysz + 2 * csz);
  const int rec_h = e->mb_h * 16 + 64;
  const size_t rec_y = (size_t)s.rec_stride_y * rec_h, rec_c = (size_t)s.rec_stride_c * (rec_h / 2);
  e->rec_alloc_bytes = rec_y + 2 * rec_c;
  for (int i = 0; i < 2; ++i) {
    DevPicture& d = e->pic[i];
    d.base = (uint8_t*)e->be->alloc (e->rec_alloc_bytes);
    e->be->fill (d.base, 0, e->rec_alloc_bytes);
    d.plane[0] = d.base + (size_t)32 * s.rec_stride_y + 32;
    d.plane[1] = d.base + rec_y + (size_t)16 * s.rec_stride_c + 16;
    d.plane[2] = d.base + rec_y + rec_c + (size_t)16 * s.rec_stride_c + 16;
    d.mbs = (WhMbState*)e->be->alloc (sizeof (WhMbState) * e->num_mb);
    e->be->fill (d.mbs, 0, sizeof (WhMbState) * e->num_mb);
  }
  e->d_records = (WhMbRecord*)e->be->alloc (sizeof (WhMbRecord) * e->num_mb);
  e->d_job = (WhPicJob*)e->be->alloc (sizeof (WhPicJob));
  e->h_records.resize (e->num_mb);
  e->cur = 0;
  e->frame_index = 0; e->frame_num = 0; e->idr_pic_id = 0; e->sps_counter = 0; e->pps_counter = 0;
  e->force_idr = false; e->have_recon = false;
  // level (au_set.cpp:530-545): the reference feeds iSpatialBitrate even with RC off
  e->level_idc = wh::select_level_idc (e->mb_w, e->mb_h, 1, p->fMaxFrameRate, p->iTargetBitrate, &e->level_1b);
  e->inited = true;
  return WELSHIP_OK;
}

int WelsHipForceIntraFrame (WelsHipEncoder* e, int bIDR) {
  if (!e || !e->inited) return WELSHIP_ERR_INIT_PARA;
  (void)bIDR;
  e->force_idr = true;
  return WELSHIP_OK;
}

const char* WelsHipBackendName (WelsHipEncoder* e) { return (e && e->be) ? e->be->name() : "none"; }

static void stage_source (WelsHipEncoder* e, const WelsHipSourcePicture* src) {
  // WelsMoveMemoryWrapper + Padding (wels_preprocess.cpp:1250-1275,1395-1450): even dims only; the rows/cols
  // that exist only because of MB alignment are luma 0 / chroma 0x80 (h_src is pre-filled that way).
  const WhSeqParams& s = e->seq;
  const int w = e->prm.iPicWidth & ~1, h = e->prm.iPicHeight & ~1;
  uint8_t* y = e->h_src.data();
  uint8_t* u = y + (size_t)s.src_stride_y * e->mb_h * 16;
  uint8_t* v = u + (size_t)s.src_stride_c * e->mb_h * 8;
  for (int r = 0; r < h; ++r) memcpy (y + (size_t)r * s.src_stride_y, src->pData[0] + (size_t)r * src->iStride[0], w);
  for (int r = 0; r < h / 2; ++r) {
    memcpy (u + (size_t)r * s.src_stride_c, src->pData[1] + (size_t)r * src->iStride[1], w / 2);
    memcpy (v + (size_t)r * s.src_stride_c, src->pData[2] + (size_t)r * src->iStride[2], w / 2);
  }
}

int WelsHipEncodeFrame (WelsHipEncoder* e, const WelsHipSourcePicture* src, WelsHipFrameBSInfo* out) {
  if (!e || !e->inited || !src || !out) return WELSHIP_ERR_INIT_PARA;
  if (src->iColorFormat != 23) { set_err ("only videoFormatI420 input"); return WELSHIP_ERR_UNSUPPORTED; }
  if (src->iPicWidth != e->prm.iPicWidth || src->iPicHeight != e->prm.iPicHeight) { set_err ("source size differs from the initialised size"); return WELSHIP_ERR_INIT_PARA; }
  const WhSeqParams& s = e->seq;
  // ---- frame type (encoder_ext.cpp DecideFrameType: IDR at index 0 / intra period / on request) ----
  bool idr = e->force_idr || e->frame_index == 0;
  if (!idr && e->prm.uiIntraPeriod > 0 && (uint32_t)e->frame_index >= e->prm.uiIntraPeriod) idr = true;
  if (idr) { e->frame_index = 0; e->frame_num = 0; e->force_idr = false; }
  const int qp = e->prm.iDLayerQp;

  // ---- device work ----
  stage_source (e, src);
  const size_t ysz = (size_t)s.src_stride_y * e->mb_h * 16, csz = (size_t)s.src_stride_c * e->mb_h * 8;
  e->be->upload (e->d_src, e->h_src.data(), ysz + 2 * csz);
  DevPicture& cur = e->pic[e->cur];
  DevPicture& ref = e->pic[e->cur ^ 1];
  WhPicJob job;
  memset (&job, 0, sizeof (job));
  job.src[0] = e->d_src; job.src[1] = e->d_src + ysz; job.src[2] = e->d_src + ysz + csz;
  for (int i = 0; i < 3; ++i) { job.rec[i] = cur.plane[i]; job.ref[i] = idr ? nullptr : ref.plane[i]; }
  job.records = e->d_records;
  job.mbs = cur.mbs;
  job.ref_mbs = idr ? nullptr : ref.mbs;
  job.qp = qp;
  job.slice_type = idr ? WH_SLICE_I : WH_SLICE_P;
  job.qp_delta = nullptr;
  job.ref_is_p = ref.is_p ? 1 : 0;
  e->be->upload (e->d_job, &job, sizeof (job));
  if (idr) e->be->run_intra (s, e->d_job, 1);
  else e->be->run_inter (s, e->d_job, 1);
  e->be->download (e->h_records.data(), e->d_records, sizeof (WhMbRecord) * e->num_mb);
  if (s.deblock_idc != 1) e->be->run_deblock (s, e->d_job, 1);
  if (e->prm.uiIntraPeriod != 1) e->be->run_expand (s, e->d_job, 1);
  e->be->sync();
  cur.is_p = !idr;
  e->have_recon = true;

  // ---- bitstream ----
  e->bs.clear();
  e->nal_len.clear();
  int n_param_nals = 0;
  if (idr) {
    if (e->prm.eSpsPpsIdStrategy == 1) {          // INCREASING_ID (paraset_strategy.cpp:334-372)
      e->sps_id_in_bs = e->sps_counter % 32; e->pps_id_in_bs = e->pps_counter % 57;
      ++e->sps_counter; ++e->pps_counter;
    } else { e->sps_id_in_bs = 0; e->pps_id_in_bs = 0; }
    e->idr_pic_id = (e->idr_pic_id < 65535) ? e->idr_pic_id + 1 : 0;   // WriteSsvcParaset (encoder_ext.cpp:3122-3136)
    wh::SpsParams sp;
    sp.sps_id = e->sps_id_in_bs; sp.level_idc = e->level_idc; sp.constraint_set3 = e->level_1b;
    sp.width = e->prm.iPicWidth; sp.height = e->prm.iPicHeight; sp.mb_w = e->mb_w; sp.mb_h = e->mb_h;
    sp.num_ref_frames = 1; sp.gaps_in_frame_num = false; sp.frame_cropping = e->prm.bEnableFrameCroppingFlag != 0;
    std::vector<uint8_t> rbsp;
    wh::write_sps_rbsp (rbsp, sp);
    e->nal_len.push_back (wh::append_nal (e->bs, 3, 7, rbsp));
    wh::PpsParams pp;
    pp.pps_id = e->pps_id_in_bs; pp.sps_id = e->sps_id_in_bs;
    rbsp.clear();
    wh::write_pps_rbsp (rbsp, pp);
    e->nal_len.push_back (wh::append_nal (e->bs, 3, 8, rbsp));
    n_param_nals = 2;
  }
  const size_t vcl_start = e->bs.size();
  std::vector<uint8_t> rbsp;
  rbsp.reserve (1 << 16);
  for (int si = 0; si < s.num_slices; ++si) {
    rbsp.clear();
    wh::BitWriter bw (&rbsp);
    wh::SliceHeaderParams sh;
    sh.first_mb = s.slice_first_mb[si];
    sh.slice_type = idr ? 2 : 0;
    sh.pps_id = e->pps_id_in_bs;
    sh.frame_num = e->frame_num;
    sh.idr = idr;
    sh.idr_pic_id = e->idr_pic_id;
    sh.nal_ref_idc = 3;
    sh.slice_qp = qp;
    sh.disable_deblocking_idc = s.deblock_idc;
    sh.alpha_offset = s.alpha_offset; sh.beta_offset = s.beta_offset;
    wh::write_slice_header (bw, sh);
    wh::SliceEntropyState st;
    st.slice_type = idr ? WH_SLICE_I : WH_SLICE_P;
    st.last_qp = qp;
    for (int xy = s.slice_first_mb[si]; xy < s.slice_first_mb[si + 1]; ++xy) {
      const int mbx = xy % e->mb_w, mby = xy / e->mb_w;
      int avail = 0;
      if (mbx > 0 && xy - 1 >= s.slice_first_mb[si]) avail |= wh::WH_AVAIL_LEFT;
      if (mby > 0 && xy - e->mb_w >= s.slice_first_mb[si]) avail |= wh::WH_AVAIL_TOP;
      int dbqp = qp;
      const int rc = wh::write_mb_cavlc (bw, st, e->h_records.data(), e->mb_w, mbx, mby, avail, &dbqp);
      if (rc == -1) { set_err ("CAVLC level escape overflow (re-encode at higher QP not implemented)"); return WELSHIP_ERR_VLC_OVERFLOW; }
      if (rc) { set_err ("bad macroblock record"); return WELSHIP_ERR_UNKNOWN; }
    }
    wh::write_slice_end (bw, st);
    e->nal_len.push_back (wh::append_nal (e->bs, 3, idr ? 5 : 1, rbsp));
  }

  // ---- SFrameBSInfo ----
  memset (out, 0, sizeof (*out));
  int li = 0;
  if (n_param_nals) {
    WelsHipLayerBSInfo& L = out->sLayerInfo[li++];
    L.uiLayerType = WELSHIP_NON_VIDEO_CODING_LAYER; L.eFrameType = WelsHipFrameTypeIDR;
    L.iNalCount = n_param_nals; L.pNalLengthInByte = e->nal_len.data(); L.pBsBuf = e->bs.data();
  }
  WelsHipLayerBSInfo& V = out->sLayerInfo[li++];
  V.uiLayerType = WELSHIP_VIDEO_CODING_LAYER; V.eFrameType = idr ? WelsHipFrameTypeIDR : WelsHipFrameTypeP;
  V.iNalCount = s.num_slices; V.pNalLengthInByte = e->nal_len.data() + n_param_nals; V.pBsBuf = e->bs.data() + vcl_start;
  out->iLayerNum = li;
  out->eFrameType = idr ? WelsHipFrameTypeIDR : WelsHipFrameTypeP;
  out->iFrameSizeInBytes = (int32_t)e->bs.size();
  out->uiTimeStamp = src->uiTimeStamp;

  // ---- advance ----
  ++e->frame_index;
  e->frame_num = (e->frame_num + 1) & 0x7fff;
  e->cur ^= 1;
  return WELSHIP_OK;
}

int WelsHipGetReconFrame (WelsHipEncoder* e, uint8_t* dst, size_t bytes) {
  if (!e || !e->inited || !e->have_recon || !dst) return WELSHIP_ERR_INIT_PARA;
  const int w = e->prm.iPicWidth, h = e->prm.iPicHeight;
  if (bytes < (size_t)w * h * 3 / 2) return WELSHIP_ERR_INIT_PARA;
  const WhSeqParams& s = e->seq;
  std::vector<uint8_t> tmp (e->rec_alloc_bytes);
  const DevPicture& p = e->pic[e->cur ^ 1];     // the picture encoded last
  e->be->download (tmp.data(), p.base, e->rec_alloc_bytes);
  e->be->sync();
  const uint8_t* y = tmp.data() + (p.plane[0] - p.base);
  const uint8_t* u = tmp.data() + (p.plane[1] - p.base);
  const uint8_t* v = tmp.data() + (p.plane[2] - p.base);
  for (int r = 0; r < h; ++r) memcpy (dst + (size_t)r * w, y + (size_t)r * s.rec_stride_y, w);
  uint8_t* du = dst + (size_t)w * h;
  uint8_t* dv = du + (size_t) (w / 2) * (h / 2);
  for (int r = 0; r < h / 2; ++r) {
    memcpy (du + (size_t)r * (w / 2), u + (size_t)r * s.rec_stride_c, w / 2);
    memcpy (dv + (size_t)r * (w / 2), v + (size_t)r * s.rec_stride_c, w / 2);
  }
  return WELSHIP_OK;
}

}  // extern "C"
