// encoder.cpp -- the host-side encoder: session (mirror of ISVCEncoder) and session group.
//
// What stays on the host here is what north_star keeps on the host in the reference as well:
// parameter handling, frame-type decision, SPS/PPS/slice headers, CAVLC and NAL packing
// (codec/encoder/plus/src/welsEncoderExt.cpp:175-500, codec/encoder/core/src/encoder_ext.cpp:3441-3960
// WelsEncoderEncodeExt).  Everything per-macroblock runs on the device through wh::Backend.
//
// A *group* is N independent sessions with identical parameters that advance in lock step: one
// batched launch set per frame step covers all N pictures, which is how a single MI355X is filled
// (independent sessions / simulcast layers / all-IDR frames have no mutual dependency).
#include "encoder_internal.h"
using wh::align_up; using wh::rec_blocks_on; using wh::DevPicture;

namespace wh {
std::string& last_error() { static thread_local std::string e; return e; }
}

#define WH_PIPE_MAX_AHEAD 3            // pipelined groups: the device runs at most this many frame steps ahead of the entropy coder

namespace {


// Stream state + device buffers of one session.  The backend is shared (owned by the caller).
struct SessionCore {
  wh::Backend* be = nullptr;
  WelsHipEncParam prm;
  WhSeqParams seq;
  int mb_w = 0, mb_h = 0, num_mb = 0;
  int ring = 1;                       // number of source slots resident in HBM
  std::vector<uint8_t*> d_src;        // [ring] source pictures, MB-aligned dims, macroblock-tiled (WH_SRC_*)
  uint8_t* d_src_planar = nullptr;    // where an upload lands (Y | U | V, tight strides) before the device rearranges it into its slot
  DevPicture pic[WH_PIPE_MAX_AHEAD + 2];   // reconstruction pictures: two (current / reference); a pipelined group adds one per step it runs ahead (see Pending)
  int nbuf = 2;
  int cur = 0;
  int ref_of (int c) const { return (c + nbuf - 1) % nbuf; }
  int next_of (int c) const { return (c + 1) % nbuf; }
  int last_slot = 0;                  // source slot of the previous frame (VAA reference)
  bool prev_src_dirty = false;        // that slot received a new upload since the frame was begun
  WhMbRecord* d_records = nullptr;
  uint8_t* d_rec_blk = nullptr;       // the unfiltered reconstruction of the picture being coded, macroblock by macroblock (WhPicJob::rec_blk): lives from mode decision to the deblocking pass of the same step
  // packed records (common/compact.h): what a session GROUP copies back instead of the full records
  uint8_t* d_compact = nullptr;
  uint32_t* d_compact_off = nullptr;
  WhHostVec<uint8_t> h_compact;
  WhHostVec<uint32_t> h_compact_off;
  bool use_compact = false;
  uint32_t* d_order = nullptr;
  int32_t* d_bands = nullptr;
  uint32_t* d_dbflags = nullptr;
  uint32_t db_gen = 0;
  size_t rec_alloc_bytes = 0, src_bytes = 0, ysz = 0, csz = 0;
  WhHostVec<uint8_t> h_src;
  WhHostVec<WhMbRecord> h_records;
  std::vector<uint8_t> bs;
  std::vector<int32_t> nal_len;
  int frame_index = 0, frame_num = 0, idr_pic_id = 0;
  int sps_counter = 0, pps_counter = 0, sps_id_in_bs = 0, pps_id_in_bs = 0;
  bool force_idr = false, have_recon = false, cur_idr = true;
  int level_idc = 0;
  bool level_1b = false;
  int n_param_nals = 0;
  size_t vcl_start = 0;
  // per-MB QP offsets of the picture being encoded: all zero (and not passed to the device) unless a macroblock had to
  // be re-encoded after a CAVLC level overflow (svc_encode_slice.cpp:572-576,1863-1867)
  WhMbCtl* d_mb_ctl = nullptr;        // WhPicJob::mb_ctl
  WhHostVec<WhMbCtl> h_mb_ctl;
  bool qp_map_in_use = false;
  int overflow_mb = -1;               // set by finish_frame when it returns WELSHIP_ERR_VLC_OVERFLOW
  int overflow_qp = 0;                // uiLumaQp of that macroblock when the overflow was detected
  int overflow_reencodes = 0;         // statistics
  uint32_t* d_scene = nullptr;        // scene-change statistic of the current source picture (WhPicJob::scene_count)
  uint32_t h_scene = 0;
  bool scene_idr = false;             // LARGE_CHANGED_SCENE seen for the picture about to be encoded
  WhPicJob cur_job;                   // what begin_frame described (re-issued by retry_after_overflow)
  // ---- pipelined session groups (WelsHipGroupEncodeFramesPipelined): the device codes picture k while the host entropy-codes picture
  // k - 1.  The stream state advances when a picture is SUBMITTED; what its entropy coding needs later is kept here.  Second set of
  // packed-record buffers and of the staging buffer (picture k's records / source are written while k - 1's are still being read), third
  // reconstruction picture (a picture whose entropy coding hits a CAVLC overflow is coded again on the device after its successor has
  // already been: its reference must still exist).
  struct Pending { bool valid = false, idr = false; int frame_num = 0, buf = 0; WhPicJob job; };
  std::deque<Pending> pendq;          // submitted, not yet entropy-coded: oldest first
  Pending fin;                        // the one being finished (taken off the queue)
  int depth = 1;                      // buffer sets = steps in flight at most (1 + the steps the device runs ahead)
  bool pipelined = false;
  uint8_t* d_compact_n[WH_PIPE_MAX_AHEAD] = {};        // buffer sets 1 .. (set 0: the members every group has)
  WhHostVec<uint8_t> h_compact_n[WH_PIPE_MAX_AHEAD];
  WhHostVec<uint8_t> h_src_n[WH_PIPE_MAX_AHEAD];
  uint8_t* d_planar_n[WH_PIPE_MAX_AHEAD] = {};         // upload targets (the batch tiling pass of step k reads one while later steps are uploaded)
  int pbuf = 0;                       // which set the picture being submitted uses
  uint8_t* dcompact (int b) const { return b ? d_compact_n[b - 1] : d_compact; }
  uint32_t* dcompact_off (int b) const { return x_doff[b] ? x_doff[b] : d_compact_off; }
  WhHostVec<uint8_t>& hcompact (int b) { return b ? h_compact_n[b - 1] : h_compact; }
  uint32_t* hcompact_off (int b) { return x_hoff[b] ? x_hoff[b] : h_compact_off.data(); }
  WhHostVec<uint8_t>& hsrc (int b) { return b ? h_src_n[b - 1] : h_src; }
  uint8_t* planar (int b) const { return b ? d_planar_n[b - 1] : d_src_planar; }
  // pipelined groups: the offset tables of all sessions are slices of one device / one page-locked host array per buffer set (one copy
  // brings all of them), owned by the group
  uint32_t* x_doff[WH_PIPE_MAX_AHEAD + 1] = {};
  uint32_t* x_hoff[WH_PIPE_MAX_AHEAD + 1] = {};

  static int validate (const WelsHipEncParam* p) {
    // same spirit as ParamValidationExt (encoder_ext.cpp:403-680)
    if (p->iPicWidth < 16 || p->iPicHeight < 16 || p->iPicWidth > 4096 || p->iPicHeight > 2304) { set_err ("invalid picture size"); return WELSHIP_ERR_INIT_PARA; }
    if (p->fMaxFrameRate <= 0.f) { set_err ("invalid frame rate"); return WELSHIP_ERR_INIT_PARA; }
    if (p->iDLayerQp < 0 || p->iDLayerQp > 51) { set_err ("invalid QP"); return WELSHIP_ERR_INIT_PARA; }
    if (p->iComplexityMode < 0 || p->iComplexityMode > 2) { set_err ("invalid complexity mode"); return WELSHIP_ERR_INIT_PARA; }
    if (p->iUsageType != 0) { set_err ("only CAMERA_VIDEO_REAL_TIME is supported"); return WELSHIP_ERR_UNSUPPORTED; }
    if (p->iRCMode != -1) { set_err ("only RC_OFF_MODE (-1) is supported"); return WELSHIP_ERR_UNSUPPORTED; }
    if (p->iTemporalLayerNum != 1 || p->iSpatialLayerNum != 1) { set_err ("only one temporal and one spatial layer supported"); return WELSHIP_ERR_UNSUPPORTED; }
    if (p->iEntropyCodingModeFlag != 0) { set_err ("CABAC is not implemented"); return WELSHIP_ERR_UNSUPPORTED; }
    // bEnableAdaptiveQuant is accepted and ignored exactly like the reference does ("turn off adaptive quant now",
    // ParamValidation, encoder_ext.cpp:300-301); bEnableFrameSkip only acts under rate control (RC is off here);
    // bEnableSceneChangeDetect is implemented (kernels/scene_pic.h + SessionCore::detect_scene_change)
    if (p->bEnableBackgroundDetection || p->bEnableLongTermReference || p->bEnableDenoise) {
      set_err ("background detection / LTR / denoise are not supported"); return WELSHIP_ERR_UNSUPPORTED;
    }
    if (p->uiSliceMode < 0 || p->uiSliceMode > 2) { set_err ("slice mode must be 0 (single), 1 (fixed number) or 2 (raster)"); return WELSHIP_ERR_UNSUPPORTED; }
    if (p->iLoopFilterDisableIdc < 0 || p->iLoopFilterDisableIdc > 2) { set_err ("deblocking idc must be 0..2"); return WELSHIP_ERR_UNSUPPORTED; }
    if (p->iLoopFilterAlphaC0Offset < -6 || p->iLoopFilterAlphaC0Offset > 6 || p->iLoopFilterBetaOffset < -6 || p->iLoopFilterBetaOffset > 6) {
      set_err ("deblocking alpha/beta offsets must be -6..6"); return WELSHIP_ERR_INIT_PARA;     // ParamValidation, encoder_ext.cpp:316-323
    }
    // CONSTANT_ID 0, INCREASING_ID 1; SPS_LISTING 2 and SPS_LISTING_AND_PPS_INCREASING 3 keep finding the session's one SPS /
    // PPS in their lists while the parameters never change (they cannot here), i.e. they write what CONSTANT_ID writes
    // (checked against the reference over 70 IDRs, forced IDRs and EncodeParameterSets); SPS_PPS_LISTING 6 is not implemented
    if (p->eSpsPpsIdStrategy < 0 || p->eSpsPpsIdStrategy > 3) { set_err ("SpsPpsIdStrategy must be 0..3"); return WELSHIP_ERR_UNSUPPORTED; }
    return WELSHIP_OK;
  }

  bool single_slice_mode = true;      // the reference's uiSliceMode ended up as SM_SINGLE_SLICE (requested or fall-back)
  int hdr_deblock_idc = 0;            // disable_deblocking_filter_idc written to the slice headers (seq.deblock_idc drives the device)
  int compute_slices() {
    WhSeqParams& s = seq;
    // "only have one MB, set to single_slice" (ParamValidationExt, encoder_ext.cpp:541-544) comes before everything else
    const bool one_mb = prm.iPicWidth <= 16 && prm.iPicHeight <= 16;
    if (prm.uiSliceMode == 2 && !one_mb) return compute_raster_slices();
    int n = (prm.uiSliceMode == 0 || one_mb) ? 1 : prm.uiSliceNum;
    if (n < 1) return -1;
    // SliceArgumentValidationFixedSliceMode (encoder_ext.cpp:178-255): small pictures fall back to one slice,
    // the slice count is capped, and with RC off every slice gets num_mb / n macroblocks (NOT row aligned),
    // the last one the remainder (CheckFixedSliceNumMultiSliceSetting, svc_enc_slice_segment.cpp:125-150).
    if (n <= 1 || num_mb <= 48) n = 1;
    if (n > 35) n = 35;
    if (n > 1 && num_mb / n <= 0) n = 1;
    s.num_slices = n;
    single_slice_mode = n == 1;
    for (int i = 0; i < n; ++i) s.slice_first_mb[i] = i * (num_mb / n);
    s.slice_first_mb[n] = num_mb;
    return 0;
  }

  // SM_RASTER_SLICE (ParamValidationExt, encoder_ext.cpp:560-612 + CheckRasterMultiSliceSetting,
  // svc_enc_slice_segment.cpp:166-215): uiSliceMbNum[] macroblocks per slice; entry 0 == 0 means one slice per row.
  int compute_raster_slices() {
    WhSeqParams& s = seq;
    const int max_slices = 35;                              // MAX_SLICES_NUM
    int cnt[36];
    int n = 0;
    single_slice_mode = false;
    if (prm.uiSliceMbNum[0] == 0) {                         // row slices: stays SM_RASTER_SLICE even for one row
      if (mb_h > max_slices) return -1;
      n = mb_h;
      for (int i = 0; i < n; ++i) cnt[i] = mb_w;
    } else {
      int total = 0;
      while (n < max_slices && prm.uiSliceMbNum[n] > 0) {
        cnt[n] = (int)prm.uiSliceMbNum[n];
        total += cnt[n];
        ++n;
        if (total >= num_mb) break;
      }
      if (total > num_mb) cnt[n - 1] -= total - num_mb;     // the last one is cut ...
      else if (total < num_mb) {                             // ... or a slice with the rest is appended
        if (n >= max_slices) return -1;
        cnt[n++] = num_mb - total;
      }
      if (n == 1 || num_mb <= 48) { n = 1; cnt[0] = num_mb; single_slice_mode = true; }   // turned into SM_SINGLE_SLICE
    }
    s.num_slices = n;
    int first = 0;
    for (int i = 0; i < n; ++i) { s.slice_first_mb[i] = first; first += cnt[i]; }
    s.slice_first_mb[n] = num_mb;
    return 0;
  }

  int init (wh::Backend* backend, const WelsHipEncParam* p, int ring_slots) {
    be = backend; prm = *p; ring = ring_slots < 2 ? 2 : ring_slots;     // the previous source stays resident (VAA SADs)
    mb_w = (p->iPicWidth + 15) >> 4; mb_h = (p->iPicHeight + 15) >> 4; num_mb = mb_w * mb_h;
    WhSeqParams& s = seq;
    memset (&s, 0, sizeof (s));
    s.mb_w = mb_w; s.mb_h = mb_h;
    s.src_stride_y = mb_w * 16; s.src_stride_c = mb_w * 8;
    s.rec_stride_y = align_up (mb_w * 16 + 64, 64); s.rec_stride_c = s.rec_stride_y / 2;
    s.complexity = p->iComplexityMode;
    s.chroma_qp_offset = 0;
    s.deblock_idc = p->iLoopFilterDisableIdc;
    // the API values are slice_alpha_c0_offset_div2 / slice_beta_offset_div2 (InitDqLayers, encoder_ext.cpp:1104-1105)
    s.alpha_offset = p->iLoopFilterAlphaC0Offset * 2; s.beta_offset = p->iLoopFilterBetaOffset * 2;
    s.mv_range = 64;
    s.blk8_w = mb_w * 2; s.blk8_h = mb_h * 2;      // the reference analyses its MB-aligned, zero-padded copy of the source
    if (compute_slices()) { set_err ("invalid slice number"); return WELSHIP_ERR_INIT_PARA; }
    // InitDqLayers (encoder_ext.cpp:1109-1117): with a single slice (requested, or after the fall-back above)
    // "filter all but slice edges" is signalled and run as idc 0
    // slice threads in the reference: iMultipleThreadIdc = min (threads, slice count); with more than one, idc 0 becomes 2
    // (WelsEncoderApplyLTR ... InitSliceSettings, encoder_ext.cpp:2051-2055).  "auto" (0) depends on the machine the
    // reference runs on and is read as 1 here.
    {
      const int threads = p->iMultipleThreadIdc <= 0 ? 1 : p->iMultipleThreadIdc;
      const int max_slices = single_slice_mode ? 1 : s.num_slices;
      if (std::min (threads, max_slices) != 1 && s.deblock_idc == 0) s.deblock_idc = 2;
    }
    if (single_slice_mode && s.deblock_idc == 2) s.deblock_idc = 0;
    hdr_deblock_idc = s.deblock_idc;
    // A reference quirk reproduced for byte parity: asked for slice threads (iMultipleThreadIdc != 1) it moves the
    // deblocking of idc 1/2 pictures into its slice tasks (bDeblockingParallelFlag, encoder_ext.cpp:459-464,1107-1121,
    // 2773-2783,3870-3879) -- but when the slice count then limits it to one thread after all (a raster-mode picture with
    // a single slice: one macroblock row in row-slice mode) no slice task exists and the picture is never filtered,
    // although its slice header still says idc 2.
    if (p->iMultipleThreadIdc != 1 && !single_slice_mode && s.num_slices == 1 && s.deblock_idc == 2) s.deblock_idc = 1;
    ysz = (size_t)s.src_stride_y * mb_h * 16; csz = (size_t)s.src_stride_c * mb_h * 8; src_bytes = ysz + 2 * csz;
    h_src.assign (src_bytes, 0);
    memset (h_src.data() + ysz, 0x80, 2 * csz);     // CWelsPreProcess::Padding: luma 0, chroma 0x80
    // ---- device memory: everything is allocated first; a failure releases what was taken and fails the call ----
    bool oom = false;
    auto A = [&] (size_t n) { void* p = be->alloc (n); if (!p) oom = true; return p; };
    d_src.assign (ring, nullptr);
    for (int i = 0; i < ring; ++i) d_src[i] = (uint8_t*)A (src_bytes);
    d_src_planar = (uint8_t*)A (src_bytes);
    const int rec_h = mb_h * 16 + 64;
    const size_t rec_y = (size_t)s.rec_stride_y * rec_h, rec_c = (size_t)s.rec_stride_c * (rec_h / 2);
    rec_alloc_bytes = rec_y + 2 * rec_c;
    for (int i = 0; i < 2; ++i) {
      pic[i].base = (uint8_t*)A (DevPicture::alloc_bytes (rec_alloc_bytes + 128));     // 64 guard bytes either side, then the tiled twin
      pic[i].mbs = (WhMbState*)A (sizeof (WhMbState) * num_mb);
    }
    d_records = (WhMbRecord*)A (sizeof (WhMbRecord) * num_mb);
    d_rec_blk = rec_blocks_on() ? (uint8_t*)A ((size_t)WH_SRC_MB_BYTES * num_mb) : nullptr;
    // processing order tables (common/mb_order.h): per slice, whole picture, per deblocking band
    std::vector<uint16_t> order ((size_t)num_mb * 3);
    const int band = 0;            // (rows per band of the per-slice order; banded orders measured slower, profiles/HISTORY.md -- one band)
    for (int i = 0; i < s.num_slices; ++i) wh_build_mb_order (mb_w, s.slice_first_mb[i], s.slice_first_mb[i + 1], order.data() + s.slice_first_mb[i], band);
    wh_build_mb_order (mb_w, 0, num_mb, order.data() + num_mb);
    // deblocking bands: after the slice fall-backs above, i.e. for the idc the device really runs
    WhHostVec<int32_t> bands (3 * (size_t) (mb_h + s.num_slices) + 1);
    const int brows = WH_DB_BAND_ROWS;
    const bool by_slice = true;           // bands confined to slices also with idc 0 (profiles/r03_deblock_bands.txt)
    const int nb = wh_build_db_bands (mb_w, mb_h, s.num_slices, s.slice_first_mb, s.deblock_idc, brows, bands.data(), (int)bands.size(), by_slice);
    if (nb < 1) { set_err ("deblocking band table"); release(); return WELSHIP_ERR_UNKNOWN; }
    for (int b = 0; b < nb; ++b) wh_build_mb_order (mb_w, bands[b], bands[b + 1], order.data() + 2 * (size_t)num_mb + bands[b]);
    WhHostVec<uint32_t> order32 (order.begin(), order.end());      // 32-bit on the device (scalar loads)
    order32.resize ((size_t)num_mb * 4 + 1);                          // + the whole-picture deblocking order as items of one or two macroblocks (common/mb_order.h)
    wh_build_db_pair_items (mb_w, mb_h, wh::db_pair_min_len(), order32.data() + 3 * (size_t)num_mb);
    d_order = (uint32_t*)A (order32.size() * 4);
    d_bands = (int32_t*)A (sizeof (int32_t) * (3 * (size_t)nb + 1 + 4));      // + the one-band table of the whole picture
    d_scene = (uint32_t*)A (64);
    d_dbflags = (uint32_t*)A (sizeof (uint32_t) * num_mb);
    if (oom) { set_err ("out of device memory"); release(); return WELSHIP_ERR_MEMORY; }
    for (int i = 0; i < 2; ++i) {
      DevPicture& d = pic[i];
      be->fill (d.base, 0, DevPicture::alloc_bytes (rec_alloc_bytes + 128));
      d.place_tiles (rec_alloc_bytes + 128, rec_y);
      d.plane[0] = d.base + 64 + (size_t)32 * s.rec_stride_y + 32;
      d.plane[1] = d.base + 64 + rec_y + (size_t)16 * s.rec_stride_c + 16;
      d.plane[2] = d.base + 64 + rec_y + rec_c + (size_t)16 * s.rec_stride_c + 16;
      be->fill (d.mbs, 0, sizeof (WhMbState) * num_mb);
    }
    h_records.resize (num_mb);
    be->pin_host (h_records.data(), sizeof (WhMbRecord) * num_mb);     // D2H target of every frame
    be->pin_host (h_src.data(), src_bytes);                            // H2D source of every frame
    be->upload (d_order, order32.data(), order32.size() * 4);
    { const int32_t whole[4] = {0, num_mb, 0, num_mb}; bands.resize (3 * (size_t)nb + 1); bands.insert (bands.end(), whole, whole + 4); }
    be->upload (d_bands, bands.data(), sizeof (int32_t) * (3 * (size_t)nb + 1 + 4));
    be->fill (d_dbflags, 0, sizeof (uint32_t) * num_mb);
    if (be->sync()) { set_err ("device error while setting up the session"); release(); return WELSHIP_ERR_UNKNOWN; }
    s.mb_order = d_order;
    s.db_num_bands = nb;
    s.db_bands = d_bands;
    s.db_max_mbs = 0; s.db_max_rows = 0;
    for (int b = 0; b < nb; ++b) {
      s.db_max_mbs = std::max (s.db_max_mbs, bands[b + 1] - bands[b]);
      s.db_max_rows = std::max (s.db_max_rows, (bands[b + 1] - 1) / mb_w - bands[b] / mb_w + 1);
    }
    // level (au_set.cpp:530-545): the reference feeds iSpatialBitrate even with RC off
    level_idc = wh::select_level_idc (mb_w, mb_h, 1, p->fMaxFrameRate, p->iTargetBitrate, &level_1b);
    // GetMvMvdRange (encoder_ext.cpp:1508-1532): min (|MinVmv| >> 2, MaxVmv >> 2, 64) of the level just chosen --
    // level 1 allows +63.75 at most, so the integer search stays within 63 samples there (level 1b as well, but a
    // Baseline stream carries 1b as level 1.1 + constraint_set3 and the reference looks up 1.1: au_set.cpp:530-534)
    if (level_idc == 10) s.mv_range = 63;
    return WELSHIP_OK;
  }

  // Groups copy the records back packed (common/compact.h); pictures larger than the packer's workgroup handles keep the
  // full records, as does WELSHIP_COMPACT=0.
  int enable_compact() {
    if (num_mb > WELSHIP_PACKED_MAX_MB) return WELSHIP_OK;
    if (const char* e = getenv ("WELSHIP_COMPACT")) if (atoi (e) == 0) return WELSHIP_OK;
    d_compact = (uint8_t*)be->alloc ((size_t)num_mb * WH_COMPACT_MAX_BYTES);
    d_compact_off = (uint32_t*)be->alloc (sizeof (uint32_t) * ((size_t)num_mb + 1));
    if (!d_compact || !d_compact_off) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    h_compact.resize ((size_t)num_mb * WH_COMPACT_MAX_BYTES);
    h_compact_off.resize ((size_t)num_mb + 1);
    be->pin_host (h_compact.data(), h_compact.size());
    be->pin_host (h_compact_off.data(), sizeof (uint32_t) * h_compact_off.size());
    use_compact = true;
    return WELSHIP_OK;
  }

  // `ahead` more reconstruction pictures and record / staging / upload buffer sets (needs the packed records); before the first picture
  int enable_pipeline (int ahead, uint32_t* const* doff, uint32_t* const* hoff) {
    if (pipelined) return WELSHIP_OK;
    if (ahead < 1 || ahead > WH_PIPE_MAX_AHEAD) return WELSHIP_ERR_INIT_PARA;
    if (!use_compact || frame_index != 0 || have_recon) { set_err ("pipelined groups need packed records and must be switched on before the first picture"); return WELSHIP_ERR_UNSUPPORTED; }
    if (prm.bEnableSceneChangeDetect) { set_err ("pipelined groups: scene-change detection reads a device statistic back before every picture"); return WELSHIP_ERR_UNSUPPORTED; }
    const size_t rec_y = (size_t)seq.rec_stride_y * (mb_h * 16 + 64), rec_c = (size_t)seq.rec_stride_c * ((mb_h * 16 + 64) / 2);
    // The source ring must outlast the steps in flight: when a CAVLC overflow of picture k is found, k+1 .. k+ahead have been tiled into
    // the ring already, and the repeat of k reads its own slot and (LOW complexity: the VAA SADs) the slot of k-1 -- ahead + 2 slots.
    while (ring < ahead + 2) {
      uint8_t* p = (uint8_t*)be->alloc (src_bytes);
      if (!p) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
      d_src.push_back (p);
      ++ring;
    }
    for (int k = 0; k < ahead; ++k) {
      DevPicture& d = pic[2 + k];
      d.base = (uint8_t*)be->alloc (DevPicture::alloc_bytes (rec_alloc_bytes + 128));
      d.mbs = (WhMbState*)be->alloc (sizeof (WhMbState) * num_mb);
      d_compact_n[k] = (uint8_t*)be->alloc ((size_t)num_mb * WH_COMPACT_MAX_BYTES);
      d_planar_n[k] = (uint8_t*)be->alloc (src_bytes);
      if (!d.base || !d.mbs || !d_compact_n[k] || !d_planar_n[k]) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
      be->fill (d.base, 0, DevPicture::alloc_bytes (rec_alloc_bytes + 128));
      d.place_tiles (rec_alloc_bytes + 128, rec_y);
      d.plane[0] = d.base + 64 + (size_t)32 * seq.rec_stride_y + 32;
      d.plane[1] = d.base + 64 + rec_y + (size_t)16 * seq.rec_stride_c + 16;
      d.plane[2] = d.base + 64 + rec_y + rec_c + (size_t)16 * seq.rec_stride_c + 16;
      be->fill (d.mbs, 0, sizeof (WhMbState) * num_mb);
      h_compact_n[k].resize ((size_t)num_mb * WH_COMPACT_MAX_BYTES);
      be->pin_host (h_compact_n[k].data(), h_compact_n[k].size());
      h_src_n[k] = h_src;                     // (keeps the padding values of the MB-alignment area)
      be->pin_host (h_src_n[k].data(), src_bytes);
    }
    for (int b2 = 0; b2 <= ahead; ++b2) { x_doff[b2] = doff[b2]; x_hoff[b2] = hoff[b2]; }
    nbuf = 2 + ahead;
    depth = 1 + ahead;
    pipelined = true;
    return WELSHIP_OK;
  }

  void release() {
    if (!be) return;
    if (pipelined) {
      for (int k = 0; k < WH_PIPE_MAX_AHEAD; ++k) {
        if (!h_compact_n[k].empty()) be->unpin_host (h_compact_n[k].data());
        if (!h_src_n[k].empty()) be->unpin_host (h_src_n[k].data());
        if (d_compact_n[k]) be->free (d_compact_n[k]);
        if (d_planar_n[k]) be->free (d_planar_n[k]);
        d_compact_n[k] = nullptr; d_planar_n[k] = nullptr;
      }
      pipelined = false; nbuf = 2; depth = 1;
      for (int b = 0; b <= WH_PIPE_MAX_AHEAD; ++b) { x_doff[b] = nullptr; x_hoff[b] = nullptr; }
    }
    for (uint8_t* p : d_src) if (p) be->free (p);
    d_src.clear();
    if (d_src_planar) be->free (d_src_planar);
    d_src_planar = nullptr;
    for (int i = 0; i < WH_PIPE_MAX_AHEAD + 2; ++i) { if (pic[i].base) be->free (pic[i].base); if (pic[i].mbs) be->free (pic[i].mbs); pic[i] = DevPicture(); }
    if (d_records) be->free (d_records);
    d_records = nullptr;
    if (d_rec_blk) be->free (d_rec_blk);
    d_rec_blk = nullptr;
    if (d_compact) be->free (d_compact);
    if (d_compact_off) be->free (d_compact_off);
    d_compact = nullptr; d_compact_off = nullptr;
    if (!h_compact.empty()) { be->unpin_host (h_compact.data()); be->unpin_host (h_compact_off.data()); }
    if (!h_records.empty()) be->unpin_host (h_records.data());
    if (!h_src.empty()) be->unpin_host (h_src.data());
    if (d_order) be->free (d_order);
    d_order = nullptr;
    if (d_bands) be->free (d_bands);
    d_bands = nullptr;
    if (d_dbflags) be->free (d_dbflags);
    d_dbflags = nullptr;
    if (d_mb_ctl) be->free (d_mb_ctl);
    d_mb_ctl = nullptr;
    if (d_scene) be->free (d_scene);
    d_scene = nullptr;
    be = nullptr;
  }

  // WelsMoveMemoryWrapper + Padding (wels_preprocess.cpp:1250-1275,1395-1450): even dims only; the rows/cols
  // that exist only because of MB alignment are luma 0 / chroma 0x80 (h_src is pre-filled that way).
  // Two halves so that a group can do the host copies of all its sessions on several threads and then queue the transfers
  // without waiting for any of them: stage_source touches only this session's page-locked staging buffer, issue_upload
  // only queues the DMA (the kernels that read the picture are behind it on the same queue).
  bool upload_pending = false;        // the staging buffer may still be read by a queued transfer
  void stage_source (const WelsHipSourcePicture* src, int buf = 0) {
    const WhSeqParams& s = seq;
    const int w = prm.iPicWidth & ~1, h = prm.iPicHeight & ~1;
    uint8_t* y = hsrc (buf).data();
    uint8_t* u = y + ysz;
    uint8_t* v = u + csz;
    for (int r = 0; r < h; ++r) memcpy (y + (size_t)r * s.src_stride_y, src->pData[0] + (size_t)r * src->iStride[0], w);
    for (int r = 0; r < h / 2; ++r) {
      memcpy (u + (size_t)r * s.src_stride_c, src->pData[1] + (size_t)r * src->iStride[1], w / 2);
      memcpy (v + (size_t)r * s.src_stride_c, src->pData[2] + (size_t)r * src->iStride[2], w / 2);
    }
  }
  void issue_upload (int slot, int buf = 0) {
    if (slot == last_slot) prev_src_dirty = true;
    be->upload (d_src_planar, hsrc (buf).data(), src_bytes);   // (same queue: the next upload waits for this pass)
    be->run_src_tile (seq, d_src_planar, d_src[slot]);
    upload_pending = true;
  }
  // pipelined groups, on a worker thread: staging copy, then the transfer into upload target `buf` on queue q (nothing else: the batch
  // tiling pass runs on the compute queue, so the upload queue holds copy-engine work only and runs under the previous step's kernels)
  void stage_and_upload (const WelsHipSourcePicture* src, int buf, int q) {
    stage_source (src, buf);
    be->upload_on (q, planar (buf), hsrc (buf).data(), src_bytes);
  }
  void upload_source (int slot, const WelsHipSourcePicture* src) {
    if (upload_pending) { be->sync(); upload_pending = false; }      // the previous transfer out of the staging buffer
    stage_source (src);
    issue_upload (slot);
  }

  // Decide the frame type (encoder_ext.cpp DecideFrameType: IDR at index 0 / intra period / on request) and
  // describe the picture to the device.
  // IDR for a reason other than a scene change: first picture, request, intra period (encoder.cpp:377-391)
  bool idr_without_scene_change() const {
    return force_idr || frame_index == 0 || (prm.uiIntraPeriod > 0 && (uint32_t)frame_index >= prm.uiIntraPeriod);
  }
  // The scene-change verdict can only turn the picture into an IDR if nothing else does already and the reference's
  // iFrameIndex -- still the count BEFORE this picture at decision time (InitFrameCoding increments it afterwards,
  // encoder.cpp:281-284) -- has reached 2 * VGOP_SIZE = 16 ("avoid too frequent I frame coding", encoder.cpp:379-382).
  bool scene_check_needed() const { return prm.bEnableSceneChangeDetect && !idr_without_scene_change() && frame_index - 1 >= 16; }
  // Describe the statistic pass for the source picture in `slot` (it is compared with the previous source picture).
  void scene_job (int slot, WhPicJob* job) {
    memset (job, 0, sizeof (*job));
    job->src[0] = d_src[slot];
    job->prev_src_y = d_src[last_slot];
    job->scene_count = d_scene;
  }
  // CSceneChangeDetection::Process (SceneChangeDetection.h:215-241) + GetSceneChangeFlag: LARGE_CHANGED_SCENE only
  void scene_verdict (uint32_t motion_blocks) {
    const int n8 = seq.blk8_w * seq.blk8_h;
    const int thr_large = static_cast<int32_t> (0.85f * n8 + 0.5f + 1e-6);
    scene_idr = (int)motion_blocks >= thr_large;
  }
  // what begin_frame would refuse, without changing anything (a group checks all its sessions before it begins any)
  int begin_frame_check() {
    const bool idr = idr_without_scene_change() || scene_idr;
    // LOW complexity P pictures read the previous SOURCE picture (VAA 8x8 SADs): it must still be resident
    if (!idr && seq.complexity == 0 && prev_src_dirty) {
      set_err ("the previous source picture was overwritten: use at least two source slots and alternate them");
      return WELSHIP_ERR_INIT_PARA;
    }
    return WELSHIP_OK;
  }
  int begin_frame (int slot, WhPicJob* job) {
    if (const int rc = begin_frame_check()) return rc;
    bool idr = idr_without_scene_change() || scene_idr;
    scene_idr = false;
    prev_src_dirty = false;
    if (idr) { frame_index = 0; frame_num = 0; force_idr = false; }
    cur_idr = idr;
    DevPicture& c = pic[cur];
    DevPicture& r = pic[ref_of (cur)];
    memset (job, 0, sizeof (*job));
    job->src[0] = job->src[1] = job->src[2] = d_src[slot];
    for (int i = 0; i < 3; ++i) { job->rec[i] = c.plane[i]; job->ref[i] = idr ? nullptr : r.plane[i]; }
    for (int i = 0; i < 2; ++i) { job->rec_tiles[i] = c.tiles[i]; job->ref_tiles[i] = idr ? nullptr : r.tiles[i]; }
    job->records = d_records;
    job->rec_blk = seq.deblock_idc != 1 ? d_rec_blk : nullptr;
    job->compact = use_compact ? dcompact (pbuf) : nullptr;
    job->compact_off = use_compact ? dcompact_off (pbuf) : nullptr;
    job->mbs = c.mbs;
    job->ref_mbs = idr ? nullptr : r.mbs;
    job->qp = prm.iDLayerQp;
    job->slice_type = idr ? WH_SLICE_I : WH_SLICE_P;
    job->mb_ctl = nullptr;
    job->ref_is_p = r.is_p ? 1 : 0;
    job->prev_src_y = d_src[last_slot];
    last_slot = slot;
    job->db_flags = d_dbflags;
    if (++db_gen == 0) db_gen = 1;
    job->db_gen = db_gen;
    if (qp_map_in_use) { memset (h_mb_ctl.data(), 0, sizeof (WhMbCtl) * h_mb_ctl.size()); qp_map_in_use = false; }
    cur_job = *job;
    return WELSHIP_OK;
  }

  // The macroblock `overflow_mb` cannot be written in Baseline CAVLC at its QP: raise its QP by DELTA_QP (rc.h:77) as
  // UpdateQpForOverflow does and describe the picture again, now with the QP map.  Every other macroblock keeps its
  // QP, so the device reproduces everything up to that macroblock and continues from the re-encoded one exactly like
  // the reference's TRY_REENCODING loop.  Fails once the macroblock's QP has reached 50, as the reference does.
  int retry_after_overflow (WhPicJob* job) {
    if (overflow_mb < 0 || overflow_mb >= num_mb) return WELSHIP_ERR_UNKNOWN;
    if (h_mb_ctl.empty()) { h_mb_ctl.resize (num_mb); memset (h_mb_ctl.data(), 0, sizeof (WhMbCtl) * num_mb); }
    if (!d_mb_ctl) d_mb_ctl = (WhMbCtl*)be->alloc (sizeof (WhMbCtl) * num_mb);
    if (!d_mb_ctl) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    // `pCurMb->uiLumaQp < 50` (svc_encode_slice.cpp:572,1863); beyond that the reference gives up on the frame with
    // cmMallocMemeError (welsEncoderExt.cpp:415-420)
    if (overflow_qp >= 50) { set_err ("bitstream overflow that raising the macroblock QP cannot resolve (reference: cmMallocMemeError)"); return WELSHIP_ERR_MEMORY; }
    // uiLumaQp += DELTA_QP: on top of the QP the macroblock had when the overflow was seen -- for a macroblock without
    // coded residual that is the QP it inherited from the previous one (svc_set_mb_syn_cavlc.cpp:299), not its own
    // ... and neither uiCbp nor the MV cache are re-initialised between the passes (WhMbCtl, common/wh_types.h)
    WhMbCtl& ctl = h_mb_ctl[overflow_mb];
    const WhMbRecord& rec = h_records[overflow_mb];
    ctl.qp_delta = (int8_t) (overflow_qp + 2 - prm.iDLayerQp);
    ctl.stale_cbp = rec.cbp & 0x3f;
    if (rec.mb_type == WH_MB_P8x16) { ctl.cell12_valid = 1; ctl.cell12_mv[0] = rec.mv_tr[0]; ctl.cell12_mv[1] = rec.mv_tr[1]; }
    qp_map_in_use = true;
    ++overflow_reencodes;
    be->upload (d_mb_ctl, h_mb_ctl.data(), sizeof (WhMbCtl) * num_mb);
    if (++db_gen == 0) db_gen = 1;
    cur_job.db_gen = db_gen;
    cur_job.mb_ctl = d_mb_ctl;
    *job = cur_job;
    return WELSHIP_OK;
  }

  // Entropy-code the downloaded records into `bs` and advance the stream state.
  // `packed`: the picture's records are the packed stream in h_compact / h_compact_off (a session group's copy): the writer
  // reads them in place (entropy_cavlc.h MbView); otherwise the full records in h_records.
  int finish_frame (WelsHipFrameBSInfo* out, int64_t ts, bool packed = false) {
    pic[cur].is_p = !cur_idr;
    have_recon = true;
    const int rc = entropy_frame (out, ts, packed, cur_idr, frame_num, h_compact.data(), h_compact_off.data());
    if (rc) return rc;
    ++frame_index;
    frame_num = (frame_num + 1) & 0x7fff;
    cur = next_of (cur);
    return WELSHIP_OK;
  }
  // Pipelined groups: the picture has been handed to the device; the stream state moves on at once (the next picture is begun while
  // this one is still being coded), what the entropy coder will need is queued in `pendq`.
  void submit_advance() {
    Pending pend;
    pend.valid = true; pend.idr = cur_idr; pend.frame_num = frame_num; pend.buf = pbuf; pend.job = cur_job;
    pendq.push_back (pend);
    pic[cur].is_p = !cur_idr;
    have_recon = true;
    ++frame_index;
    frame_num = (frame_num + 1) & 0x7fff;
    cur = next_of (cur);
    pbuf = (pbuf + 1) % depth;
  }
  int finish_pending (WelsHipFrameBSInfo* out) {
    return entropy_frame (out, 0, true, fin.idr, fin.frame_num, hcompact (fin.buf).data(), hcompact_off (fin.buf));
  }
  // Entropy-code one picture: `idr` / `frame_num_` of that picture, its packed records in hc / hoff (or the full records in h_records).
  // Changes nothing of the stream state but the parameter-set ids of an IDR picture (restored on failure).
  int entropy_frame (WelsHipFrameBSInfo* out, int64_t ts, bool packed, bool idr, int frame_num_, const uint8_t* hc, const uint32_t* hoff) {
    const WhSeqParams& s = seq;
    const int qp = prm.iDLayerQp;
    const int saved_ids[5] = {sps_counter, pps_counter, sps_id_in_bs, pps_id_in_bs, idr_pic_id};
    overflow_mb = -1;
    bs.clear();
    nal_len.clear();
    std::vector<long> nal_rbsp_len;
    n_param_nals = 0;
    // iCountBsLen (see below): SEI 128 + SPS slots x 32 + PPS slots x 16 + the picture part; the id strategies that keep
    // lists reserve 32 SPS slots and one PPS slot (paraset_strategy.cpp:404-413), the others 1 and 2 (:203-211)
    const long paraset_bytes = (prm.eSpsPpsIdStrategy >= 2) ? 32 * 32 + 1 * 16 : 1 * 32 + 2 * 16;
    const long bs_capacity = 128 + paraset_bytes + (((3L * mb_w * 16 * mb_h * 16) >> 1) + 800 + 3) / 4 * 4;
    // WelsEncodeNal (nal_encap.cpp:120-131) refuses a NAL unless 1.5x its size still fits into what is left of the frame's
    // output buffer (same iCountBsLen bytes); the reference then fails the frame with cmMallocMemeError.
    auto nal_fits = [&] (const std::vector<uint8_t>& payload) {
      const long need = 4 + (long)payload.size() + 1;
      return bs_capacity - (long)bs.size() >= need + (need >> 1);
    };
    auto fail_frame = [&] () {
      sps_counter = saved_ids[0]; pps_counter = saved_ids[1]; sps_id_in_bs = saved_ids[2]; pps_id_in_bs = saved_ids[3]; idr_pic_id = saved_ids[4];
      set_err ("frame does not fit the bitstream buffer of the reference encoder (cmMallocMemeError there as well)");
      return WELSHIP_ERR_MEMORY;
    };
    std::vector<uint8_t> rbsp;
    if (idr) {
      if (prm.eSpsPpsIdStrategy == 1) {          // INCREASING_ID (paraset_strategy.cpp:334-372)
        sps_id_in_bs = sps_counter % 32; pps_id_in_bs = pps_counter % 57;
        ++sps_counter; ++pps_counter;
      } else { sps_id_in_bs = 0; pps_id_in_bs = 0; }
      idr_pic_id = (idr_pic_id < 65535) ? idr_pic_id + 1 : 0;   // WriteSsvcParaset (encoder_ext.cpp:3122-3136)
      wh::SpsParams sp;
      sp.sps_id = sps_id_in_bs; sp.level_idc = level_idc; sp.constraint_set3 = level_1b;
      sp.width = prm.iPicWidth; sp.height = prm.iPicHeight; sp.mb_w = mb_w; sp.mb_h = mb_h;
      sp.num_ref_frames = 1; sp.gaps_in_frame_num = false; sp.frame_cropping = prm.bEnableFrameCroppingFlag != 0;
      wh::write_sps_rbsp (rbsp, sp);
      if (!nal_fits (rbsp)) return fail_frame();
      nal_len.push_back (wh::append_nal (bs, 3, 7, rbsp));
      nal_rbsp_len.push_back ((long)rbsp.size());
      wh::PpsParams pp;
      pp.pps_id = pps_id_in_bs; pp.sps_id = sps_id_in_bs;
      rbsp.clear();
      wh::write_pps_rbsp (rbsp, pp);
      if (!nal_fits (rbsp)) return fail_frame();
      nal_len.push_back (wh::append_nal (bs, 3, 8, rbsp));
      nal_rbsp_len.push_back ((long)rbsp.size());
      n_param_nals = 2;
    }
    vcl_start = bs.size();
    rbsp.reserve (1 << 16);
    // The reference writes every NAL payload of a frame into one buffer of iCountBsLen bytes (RequestMemorySvc,
    // encoder_ext.cpp:1576-1613: SEI 128 + SPS 32 + PPS 2x16 + picture bytes + 800, 4-aligned) and re-encodes a
    // macroblock at QP+2 when, after writing it, fewer than 800 bytes are left (CheckBitstreamBuffer,
    // svc_set_mb_syn_cavlc.cpp:248-257) -- reproduced here so that streams near the raw picture size stay identical.
    // Its writer flushes 32 bits at a time (golomb_common.h:78-92), hence the 4-byte granularity of the position.
    long frame_pos = 0;
    for (size_t i = 0; i < nal_len.size(); ++i) frame_pos += nal_rbsp_len[i];
    for (int si = 0; si < s.num_slices; ++si) {
      rbsp.clear();
      wh::BitWriter bw (&rbsp);
      wh::SliceHeaderParams sh;
      sh.first_mb = s.slice_first_mb[si];
      sh.slice_type = idr ? 2 : 0;
      sh.pps_id = pps_id_in_bs;
      sh.frame_num = frame_num_;
      sh.idr = idr;
      sh.idr_pic_id = idr_pic_id;
      sh.nal_ref_idc = 3;
      sh.slice_qp = qp;
      sh.disable_deblocking_idc = hdr_deblock_idc;
      sh.alpha_offset = s.alpha_offset; sh.beta_offset = s.beta_offset;
      sh.num_ref_idx_override = !idr; sh.num_ref_idx_active = 1;
      wh::write_slice_header (bw, sh);
      wh::SliceEntropyState st;
      st.slice_type = idr ? WH_SLICE_I : WH_SLICE_P;
      st.last_qp = qp;
      for (int xy = s.slice_first_mb[si]; xy < s.slice_first_mb[si + 1]; ++xy) {
        const int mbx = xy % mb_w, mby = xy / mb_w;
        int avail = 0;
        if (mbx > 0 && xy - 1 >= s.slice_first_mb[si]) avail |= wh::WH_AVAIL_LEFT;
        if (mby > 0 && xy - mb_w >= s.slice_first_mb[si]) avail |= wh::WH_AVAIL_TOP;
        int dbqp = qp;
        const wh::MbView mb = packed ? wh::view_of_packed (hc, hoff, mb_w, xy, avail) : wh::view_of_record (h_records.data(), mb_w, xy, avail);
        const int rc = wh::write_mb_cavlc (bw, st, mb, &dbqp);
        const bool coded = mb.side->mb_type != WH_MB_PSKIP;
        const bool no_room = coded && bs_capacity - (frame_pos + 4 * (long) (bw.bits() / 32)) - 1 < 800;
        if (rc == -1 || (rc == 0 && no_room)) {   // the caller re-encodes the picture with this macroblock's QP raised (retry_after_overflow)
          sps_counter = saved_ids[0]; pps_counter = saved_ids[1]; sps_id_in_bs = saved_ids[2]; pps_id_in_bs = saved_ids[3]; idr_pic_id = saved_ids[4];
          overflow_mb = xy;
          overflow_qp = dbqp;
          if (packed) wh_compact_expand (hc + hoff[xy], hoff[xy + 1] - hoff[xy], &h_records[xy]);   // retry_after_overflow reads it
          set_err ("CAVLC overflow");
          return WELSHIP_ERR_VLC_OVERFLOW;
        }
        if (rc) {
          sps_counter = saved_ids[0]; pps_counter = saved_ids[1]; sps_id_in_bs = saved_ids[2]; pps_id_in_bs = saved_ids[3]; idr_pic_id = saved_ids[4];
          set_err ("bad macroblock record");
          return WELSHIP_ERR_UNKNOWN;
        }
      }
      wh::write_slice_end (bw, st);
      if (!nal_fits (rbsp)) return fail_frame();
      nal_len.push_back (wh::append_nal (bs, 3, idr ? 5 : 1, rbsp));
      frame_pos += (long)rbsp.size();
    }
    if (out) {
      memset (out, 0, sizeof (*out));
      int li = 0;
      if (n_param_nals) {
        WelsHipLayerBSInfo& L = out->sLayerInfo[li++];
        L.uiLayerType = WELSHIP_NON_VIDEO_CODING_LAYER; L.eFrameType = WelsHipFrameTypeIDR;
        L.iNalCount = n_param_nals; L.pNalLengthInByte = nal_len.data(); L.pBsBuf = bs.data();
      }
      WelsHipLayerBSInfo& V = out->sLayerInfo[li++];
      V.uiLayerType = WELSHIP_VIDEO_CODING_LAYER; V.eFrameType = idr ? WelsHipFrameTypeIDR : WelsHipFrameTypeP;
      V.iSubSeqId = idr ? 0 : 3;                  // GetSubSequenceId (encoder_ext.cpp:3110-3124): IDR 0, P of temporal layer 0 -> 3
      V.iNalCount = s.num_slices; V.pNalLengthInByte = nal_len.data() + n_param_nals; V.pBsBuf = bs.data() + vcl_start;
      out->iLayerNum = li;
      out->eFrameType = idr ? WelsHipFrameTypeIDR : WelsHipFrameTypeP;
      out->iFrameSizeInBytes = (int32_t)bs.size();
      out->uiTimeStamp = ts;
    }
    return WELSHIP_OK;
  }

  // WelsEncoderEncodeParameterSets (encoder_ext.cpp:3074-3108): SPS + PPS on their own, through the same id strategy as
  // the parameter sets of an IDR (WelsWriteParameterSets :2867-2960) -- with INCREASING_ID every call takes the next ids
  // and the slices that follow refer to them.
  int encode_parameter_sets (WelsHipFrameBSInfo* out) {
    if (prm.eSpsPpsIdStrategy == 1) {
      sps_id_in_bs = sps_counter % 32; pps_id_in_bs = pps_counter % 57;
      ++sps_counter; ++pps_counter;
    } else { sps_id_in_bs = 0; pps_id_in_bs = 0; }
    bs.clear();
    nal_len.clear();
    std::vector<uint8_t> rbsp;
    wh::SpsParams sp;
    sp.sps_id = sps_id_in_bs; sp.level_idc = level_idc; sp.constraint_set3 = level_1b;
    sp.width = prm.iPicWidth; sp.height = prm.iPicHeight; sp.mb_w = mb_w; sp.mb_h = mb_h;
    sp.num_ref_frames = 1; sp.gaps_in_frame_num = false; sp.frame_cropping = prm.bEnableFrameCroppingFlag != 0;
    wh::write_sps_rbsp (rbsp, sp);
    nal_len.push_back (wh::append_nal (bs, 3, 7, rbsp));
    wh::PpsParams pp;
    pp.pps_id = pps_id_in_bs; pp.sps_id = sps_id_in_bs;
    rbsp.clear();
    wh::write_pps_rbsp (rbsp, pp);
    nal_len.push_back (wh::append_nal (bs, 3, 8, rbsp));
    memset (out, 0, sizeof (*out));
    WelsHipLayerBSInfo& L = out->sLayerInfo[0];
    L.uiLayerType = WELSHIP_NON_VIDEO_CODING_LAYER; L.eFrameType = WelsHipFrameTypeInvalid;
    L.iNalCount = 2; L.pNalLengthInByte = nal_len.data(); L.pBsBuf = bs.data();
    out->iLayerNum = 1; out->eFrameType = WelsHipFrameTypeInvalid; out->iFrameSizeInBytes = (int32_t)bs.size();
    return WELSHIP_OK;
  }

  int copy_recon (uint8_t* dst, size_t bytes) {
    if (!have_recon || !dst) return WELSHIP_ERR_INIT_PARA;
    const int w = prm.iPicWidth, h = prm.iPicHeight;
    if (bytes < (size_t)w * h * 3 / 2) return WELSHIP_ERR_INIT_PARA;
    const WhSeqParams& s = seq;
    WhHostVec<uint8_t> tmp (rec_alloc_bytes + 128);
    const DevPicture& p = pic[ref_of (cur)];     // the picture encoded last
    be->download (tmp.data(), p.base, rec_alloc_bytes + 128);
    be->sync();
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
};

// The sequence parameters of a session group's ordinary step, with the promise that its P pictures carry none of the optional per-picture
// inputs (SessionCore::begin_frame sets none; the re-run after a CAVLC overflow, which brings a QP map, goes through run_device_step).
static WhSeqParams plain_seq (const WhSeqParams& s) { WhSeqParams q = s; if (q.flags == 0) q.flags |= WH_SEQ_PLAIN; return q; }

// Run the device part of one frame step for `n` pictures described by the device array d_jobs.
void run_device_step (wh::Backend* be, const WhSeqParams& s, const WhPicJob* d_jobs, int n, bool idr, bool need_ref, bool qp_map = false) {
  if (idr) be->run_intra (s, d_jobs, n);
  else be->run_inter (s, d_jobs, n);
  if (qp_map && s.deblock_idc != 1) be->run_qp_chain (s, d_jobs, n);
  if (s.deblock_idc != 1) be->run_deblock (s, d_jobs, n);
  if (need_ref) be->run_expand (s, d_jobs, n);
}

// One round of the overflow loop for the picture session `c` is working on: raise the QP of the offending macroblock,
// run the picture again on the device (d_job: a device WhPicJob slot that belongs to this session) and fetch the records.
int reencode_after_overflow (wh::Backend* be, SessionCore& c, WhPicJob* d_job) {
  WhPicJob job;
  const int rc = c.retry_after_overflow (&job);
  if (rc) return rc;
  be->upload (d_job, &job, sizeof (job));
  run_device_step (be, c.seq, d_job, 1, c.cur_idr, c.prm.uiIntraPeriod != 1, true);
  be->download (c.h_records.data(), c.d_records, sizeof (WhMbRecord) * c.num_mb);
  if (be->sync()) { set_err ("device scheduler timed out; the picture was not encoded"); return WELSHIP_ERR_UNKNOWN; }
  return WELSHIP_OK;
}

}  // namespace

struct WelsHipEncoder {
  wh::Backend* be = nullptr;
  bool inited = false;
  SessionCore core;
  WhPicJob* d_job = nullptr;
};

struct WelsHipEncoderGroup {
  wh::Backend* be = nullptr;
  // fn (t, T): worker t of T, on threads created for the call.  (A pool of parked workers was tried and measured 35 % SLOWER
  // end to end on the 2-socket host -- 3.8 k against 5.8 k frames/s with three overlapped groups: fresh threads get placed on
  // idle cores, parked ones stay where they first ran.)
  void parallel (int n_items, const std::function<void (int, int)>& fn) {
    const int T = host_threads < n_items ? host_threads : n_items;
    if (T <= 1) { fn (0, 1); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back ([&fn, t, T] { fn (t, T); });
    for (auto& x : th) x.join();
  }
  std::vector<std::unique_ptr<SessionCore>> sess;
  int queues = 1;                               // sessions are split into `queues` contiguous chunks, one device queue each
  int chunk_first (int q) const { return (int) ((long long)sess.size() * q / queues); }
  int chunk_of (int session) const { int q = 0; while (q + 1 < queues && chunk_first (q + 1) <= session) ++q; return q; }
  WhPicJob* d_jobs = nullptr;
  std::vector<WhPicJob> h_jobs;
  int host_threads = 1;
  // pipelined mode (WelsHipGroupSetPipelined / WelsHipGroupEncodeFramesPipelined): second job array (the device may still read step
  // k - 1's descriptors when step k's are uploaded), page-locked host copies, one spare descriptor for re-runs
  bool pipelined = false;
  bool pipe_failed = false;           // WelsHipGroupSetPipelined failed half way (out of memory): some sessions have their pipeline buffers, the group is unusable
  int depth = 1;                      // buffer sets: 1 + the steps the device may run ahead of the entropy coder
  WhPicJob* d_jobs_n[WH_PIPE_MAX_AHEAD + 1] = {};   // per buffer set ([0] = d_jobs)
  WhPicJob* d_job_aux = nullptr;
  std::vector<WhPicJob> h_jobs_p[WH_PIPE_MAX_AHEAD + 1];
  long step_no = 0;
  int pending = 0;                    // submitted steps whose pictures have not been entropy-coded yet
  uint32_t* d_off_all[WH_PIPE_MAX_AHEAD + 1] = {};  // the sessions' record offset tables, one array per buffer set
  WhHostVec<uint32_t> h_off_all[WH_PIPE_MAX_AHEAD + 1];
  std::vector<void*> dl_ev;           // one event per session: its records have arrived
  void* step_ev[WH_PIPE_MAX_AHEAD + 1] = {};        // per buffer set: the kernels of the step that uses it have run
  void* up_ev[2][WH_PIPE_MAX_AHEAD + 1] = {};       // per upload queue and buffer set: the transfers out of its staging buffers are done
  // thread time the host side of the frame steps has taken so far (WelsHipGroupHostStats): [0] staging copies, [1] entropy coding
  std::mutex stat_mu;
  double host_ms[2] = {0.0, 0.0};
  long host_pics[2] = {0, 0};
  double packed_bytes = 0.0;
  void note_host_time (int what, double ms, int pics) { std::lock_guard<std::mutex> l (stat_mu); host_ms[what] += ms; host_pics[what] += pics; }
  bool step_idr = true;                         // every session codes an IDR this step (all-P otherwise, unless mixed)
  bool mixed = false;                           // sessions disagree (scene changes, forced IDRs): each queue chunk of
  std::vector<int> order;                       //   d_jobs holds its P pictures first, then its IDR pictures;
  std::vector<int> chunk_p;                     //   order[k] = session of job slot k, chunk_p[q] = P pictures of chunk q
};

extern "C" {

const char* WelsHipGetLastError (void) { return wh::last_error().c_str(); }

int WelsHipCreateEncoder (WelsHipEncoder** pp) {
  if (!pp) return WELSHIP_ERR_INIT_PARA;
  *pp = new WelsHipEncoder();
  return WELSHIP_OK;
}

int WelsHipUninitialize (WelsHipEncoder* e) {
  if (!e) return WELSHIP_ERR_INIT_PARA;
  if (e->be) {
    e->be->sync();
    e->core.release();
    if (e->d_job) e->be->free (e->d_job);
    e->d_job = nullptr;
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
  if (!p) return WELSHIP_ERR_INIT_PARA;
  (void)e;
  // the defaults of SWelsSvcCodingParam::FillDefault (param_svc.h:132-211) restricted to what we support
  memset (p, 0, sizeof (*p));
  p->iUsageType = 0; p->iRCMode = -1; p->fMaxFrameRate = 60.f;   // MAX_FRAME_RATE
  p->iTemporalLayerNum = 1; p->iSpatialLayerNum = 1; p->iComplexityMode = 0;
  p->uiIntraPeriod = 0; p->eSpsPpsIdStrategy = 1; p->iEntropyCodingModeFlag = 0;
  p->iLoopFilterDisableIdc = 0; p->bEnableFrameCroppingFlag = 1; p->iDLayerQp = 26;   // SVC_QUALITY_BASE_QP
  p->uiSliceMode = 0; p->uiSliceNum = 1;
  p->iMultipleThreadIdc = 1;
  return WELSHIP_OK;
}

int WelsHipInitializeExt (WelsHipEncoder* e, const WelsHipEncParam* p) {
  if (!e || !p) return WELSHIP_ERR_INIT_PARA;
  if (e->inited) WelsHipUninitialize (e);
  int rc = SessionCore::validate (p);
  if (rc) return rc;
  const char* berr = nullptr;
  e->be = wh::create_default_backend (p->iDevice, &berr);
  if (!e->be) { set_err (std::string ("no usable device backend: ") + (berr ? berr : "?")); return WELSHIP_ERR_NO_DEVICE; }
  e->core = SessionCore();
  rc = e->core.init (e->be, p, 2);
  if (rc) { e->core.release(); delete e->be; e->be = nullptr; return rc; }
  e->d_job = (WhPicJob*)e->be->alloc (sizeof (WhPicJob));
  if (!e->d_job) { set_err ("out of device memory"); e->core.release(); delete e->be; e->be = nullptr; return WELSHIP_ERR_MEMORY; }
  e->inited = true;
  return WELSHIP_OK;
}

int WelsHipForceIntraFrame (WelsHipEncoder* e, int bIDR) {
  // CWelsH264SVCEncoder::ForceIntraFrame (welsEncoderExt.cpp:487-500): bIDR == false is "nothing to do", success
  if (!bIDR) return WELSHIP_OK;
  if (!e || !e->inited) return WELSHIP_ERR_INIT_PARA;
  e->core.force_idr = true;
  return WELSHIP_OK;
}

int WelsHipEncodeParameterSets (WelsHipEncoder* e, WelsHipFrameBSInfo* out) {
  if (!e || !e->inited || !out) return WELSHIP_ERR_INIT_PARA;
  return e->core.encode_parameter_sets (out);
}

// CWelsH264SVCEncoder::SetOption / GetOption (welsEncoderExt.cpp:690-1200,1203-1310), the options that act without RC
int WelsHipSetOption (WelsHipEncoder* e, int id, void* opt) {
  if (!e || !e->inited || !opt) return WELSHIP_ERR_INIT_PARA;
  SessionCore& c = e->core;
  switch (id) {
  case WELSHIP_OPTION_DATAFORMAT:
    if (* (int32_t*)opt == 0) return WELSHIP_ERR_INIT_PARA;
    if (* (int32_t*)opt != 23) { set_err ("only videoFormatI420 input"); return WELSHIP_ERR_UNSUPPORTED; }
    return WELSHIP_OK;
  case WELSHIP_OPTION_IDR_INTERVAL: {        // :716-731: <= -1 means 0; takes effect at the next frame-type decision
    int32_t v = * (int32_t*)opt;
    if (v <= -1) v = 0;
    // an all-IDR session does not expand the borders of its reconstructions (nothing refers to them); leaving that
    // mode, the last picture becomes a reference after all
    if (c.prm.uiIntraPeriod == 1 && v != 1 && c.have_recon) {
      e->be->run_expand (c.seq, e->d_job, 1);
      if (e->be->sync()) { set_err ("device scheduler timed out"); return WELSHIP_ERR_UNKNOWN; }
    }
    c.prm.uiIntraPeriod = (uint32_t)v;
    return WELSHIP_OK;
  }
  case WELSHIP_OPTION_FRAME_RATE: {          // :846-858: clipped to [MIN_FRAME_RATE 1, MAX_FRAME_RATE 60]; nothing reads it while RC is off
    const float f = * (float*)opt;
    if (f <= 0) return WELSHIP_ERR_INIT_PARA;
    c.prm.fMaxFrameRate = f < 1.f ? 1.f : f > 60.f ? 60.f : f;
    return WELSHIP_OK;
  }
  case WELSHIP_OPTION_COMPLEXITY: {          // :1153-1159: stored as is; PreprocessSliceCoding reads it for every picture
    const int32_t v = * (int32_t*)opt;
    c.prm.iComplexityMode = v;
    c.seq.complexity = v;                    // LOW (0) selects the SAD / VAA-gated paths, anything else the SATD paths
    return WELSHIP_OK;
  }
  case WELSHIP_OPTION_TRACE_LEVEL: case WELSHIP_OPTION_TRACE_CALLBACK: case WELSHIP_OPTION_TRACE_CALLBACK_CONTEXT:
    return WELSHIP_OK;
  default:
    set_err ("option not supported by this engine");
    return WELSHIP_ERR_UNSUPPORTED;
  }
}

int WelsHipGetOption (WelsHipEncoder* e, int id, void* opt) {
  if (!e || !e->inited || !opt) return WELSHIP_ERR_INIT_PARA;
  const SessionCore& c = e->core;
  switch (id) {
  case WELSHIP_OPTION_DATAFORMAT: * (int32_t*)opt = 23; return WELSHIP_OK;
  case WELSHIP_OPTION_IDR_INTERVAL: * (int32_t*)opt = (int32_t)c.prm.uiIntraPeriod; return WELSHIP_OK;
  case WELSHIP_OPTION_FRAME_RATE: * (float*)opt = c.prm.fMaxFrameRate; return WELSHIP_OK;
  case WELSHIP_OPTION_COMPLEXITY: * (int32_t*)opt = c.prm.iComplexityMode; return WELSHIP_OK;
  default: return WELSHIP_ERR_INIT_PARA;     // the reference's GetOption: unknown id -> cmInitParaError
  }
}

const char* WelsHipBackendName (WelsHipEncoder* e) { return (e && e->be) ? e->be->name() : "none"; }

int WelsHipEncodeFrame (WelsHipEncoder* e, const WelsHipSourcePicture* src, WelsHipFrameBSInfo* out) {
  if (!e || !e->inited || !src || !out) return WELSHIP_ERR_INIT_PARA;
  SessionCore& c = e->core;
  if (src->iColorFormat != 23) { set_err ("only videoFormatI420 input"); return WELSHIP_ERR_UNSUPPORTED; }
  if (src->iPicWidth != c.prm.iPicWidth || src->iPicHeight != c.prm.iPicHeight) { set_err ("source size differs from the initialised size"); return WELSHIP_ERR_INIT_PARA; }
  const int slot = c.last_slot ^ 1;
  c.upload_source (slot, src);
  WhPicJob job;
  if (c.scene_check_needed()) {      // the statistic decides the frame type, so it is read back before the picture is begun
    c.scene_job (slot, &job);
    e->be->upload (e->d_job, &job, sizeof (job));
    e->be->fill (c.d_scene, 0, 4);
    e->be->run_scene (c.seq, e->d_job, 1);
    e->be->download (&c.h_scene, c.d_scene, 4);
    if (e->be->sync()) { set_err ("device error in the scene-change pass"); return WELSHIP_ERR_UNKNOWN; }
    c.scene_verdict (c.h_scene);
  }
  { const int brc = c.begin_frame (slot, &job); if (brc) return brc; }
  e->be->upload (e->d_job, &job, sizeof (job));
  run_device_step (e->be, c.seq, e->d_job, 1, c.cur_idr, c.prm.uiIntraPeriod != 1);
  e->be->download (c.h_records.data(), c.d_records, sizeof (WhMbRecord) * c.num_mb);
  if (e->be->sync()) { set_err ("device scheduler timed out; the picture was not encoded"); return WELSHIP_ERR_UNKNOWN; }
  c.upload_pending = false;
  int rc = c.finish_frame (out, src->uiTimeStamp);
  while (rc == WELSHIP_ERR_VLC_OVERFLOW) {
    rc = reencode_after_overflow (e->be, c, e->d_job);
    if (rc) return rc;
    rc = c.finish_frame (out, src->uiTimeStamp);
  }
  return rc;
}

int WelsHipGetReconFrame (WelsHipEncoder* e, uint8_t* dst, size_t bytes) {
  if (!e || !e->inited) return WELSHIP_ERR_INIT_PARA;
  return e->core.copy_recon (dst, bytes);
}

// Developer aid: the MB records (WhMbRecord[mb_w*mb_h], csrc/common/wh_types.h) of the last encoded frame.
int WelsHipDebugGetMbRecords (WelsHipEncoder* e, void* dst, size_t bytes) {
  if (!e || !e->inited || !dst) return WELSHIP_ERR_INIT_PARA;
  const size_t n = sizeof (WhMbRecord) * e->core.h_records.size();
  if (bytes < n) return WELSHIP_ERR_INIT_PARA;
  memcpy (dst, e->core.h_records.data(), n);
  return WELSHIP_OK;
}

int WelsHipDebugGetOverflowReencodes (WelsHipEncoder* e) {
  if (!e || !e->inited) return -1;
  return e->core.overflow_reencodes;
}

int WelsHipDebugBuildMbOrder (int mb_w, int first, int last, int band, uint16_t* out) {
  if (mb_w <= 0 || first < 0 || last <= first || !out) return WELSHIP_ERR_INIT_PARA;
  wh_build_mb_order (mb_w, first, last, out, band);
  return WELSHIP_OK;
}

int WelsHipDebugBuildDbPairItems (int mb_w, int mb_h, int min_len, uint32_t* out) {
  if (mb_w <= 0 || mb_h <= 0 || !out) return WELSHIP_ERR_INIT_PARA;
  wh_build_db_pair_items (mb_w, mb_h, min_len, out);
  return WELSHIP_OK;
}

// ---------------------------------------------------------------------------------- session group
int WelsHipGroupCreate (WelsHipEncoderGroup** pp, const WelsHipEncParam* p, int n_sessions, int ring_slots, int host_threads) {
  if (!pp || !p || n_sessions < 1 || n_sessions > 4096) return WELSHIP_ERR_INIT_PARA;
  int rc = SessionCore::validate (p);
  if (rc) return rc;
  const char* berr = nullptr;
  wh::Backend* be = wh::create_default_backend (p->iDevice, &berr);
  if (!be) { set_err (std::string ("no usable device backend: ") + (berr ? berr : "?")); return WELSHIP_ERR_NO_DEVICE; }
  WelsHipEncoderGroup* g = new WelsHipEncoderGroup();
  g->be = be;
  g->host_threads = host_threads < 1 ? 1 : host_threads;
  for (int i = 0; i < n_sessions; ++i) {
    g->sess.emplace_back (new SessionCore());
    rc = g->sess.back()->init (be, p, ring_slots);
    if (!rc) rc = g->sess.back()->enable_compact();
    if (rc) { for (auto& s : g->sess) s->release(); delete be; delete g; return rc; }
  }
  g->d_jobs = (WhPicJob*)be->alloc (sizeof (WhPicJob) * n_sessions);
  if (!g->d_jobs) { set_err ("out of device memory"); for (auto& s : g->sess) s->release(); delete be; delete g; return WELSHIP_ERR_MEMORY; }
  g->h_jobs.resize (n_sessions);
  *pp = g;
  return WELSHIP_OK;
}

void WelsHipGroupDestroy (WelsHipEncoderGroup* g) {
  if (!g) return;
  g->be->sync();
  for (auto& s : g->sess) s->release();
  {   // (whether or not WelsHipGroupSetPipelined got through: a call that failed half way leaves some of these behind; everything here is null-safe)
    for (int b = 0; b <= WH_PIPE_MAX_AHEAD; ++b) {
      if (!g->h_jobs_p[b].empty()) g->be->unpin_host (g->h_jobs_p[b].data());
      if (!g->h_off_all[b].empty()) g->be->unpin_host (g->h_off_all[b].data());
      if (g->d_off_all[b]) g->be->free (g->d_off_all[b]);
      if (b && g->d_jobs_n[b]) g->be->free (g->d_jobs_n[b]);
      g->be->event_destroy (g->step_ev[b]); g->be->event_destroy (g->up_ev[0][b]); g->be->event_destroy (g->up_ev[1][b]);
    }
    for (void* e : g->dl_ev) g->be->event_destroy (e);
    if (g->d_job_aux) g->be->free (g->d_job_aux);
  }
  g->be->free (g->d_jobs);
  delete g->be;
  delete g;
}

int WelsHipGroupUploadSource (WelsHipEncoderGroup* g, int session, int slot, const WelsHipSourcePicture* src) {
  if (!g || session < 0 || session >= (int)g->sess.size() || !src) return WELSHIP_ERR_INIT_PARA;
  SessionCore& c = *g->sess[session];
  if (slot < 0 || slot >= c.ring) return WELSHIP_ERR_INIT_PARA;
  if (src->iPicWidth != c.prm.iPicWidth || src->iPicHeight != c.prm.iPicHeight) return WELSHIP_ERR_INIT_PARA;
  g->be->select_queue (g->chunk_of (session));
  c.upload_source (slot, src);
  return WELSHIP_OK;
}

int WelsHipGroupBegin (WelsHipEncoderGroup* g, int slot) {
  if (!g) return WELSHIP_ERR_INIT_PARA;
  // (a pipelined group's pictures write their records into the buffer set of their step: the synchronous calls would read the wrong one)
  if (g->pipelined) { set_err ("pipelined group: frame steps go through WelsHipGroupEncodeFramesPipelined"); return WELSHIP_ERR_INIT_PARA; }
  if (g->pipe_failed) { set_err ("the group could not be made a pipelined one (out of memory) and is in an inconsistent state: destroy it"); return WELSHIP_ERR_INIT_PARA; }
  const int n = (int)g->sess.size();
  // scene-change statistic for the sessions whose frame type it can still change: one launch, read back before the
  // pictures are begun
  {
    std::vector<int> need;
    for (int i = 0; i < n; ++i) if (g->sess[i]->scene_check_needed()) need.push_back (i);
    if (!need.empty()) {
      g->be->select_queue (0);
      for (size_t k = 0; k < need.size(); ++k) {
        SessionCore& c = *g->sess[need[k]];
        c.scene_job (slot % c.ring, &g->h_jobs[k]);
        g->be->fill (c.d_scene, 0, 4);
      }
      g->be->upload (g->d_jobs, g->h_jobs.data(), sizeof (WhPicJob) * need.size());
      g->be->run_scene (g->sess[0]->seq, g->d_jobs, (int)need.size());
      for (int i : need) g->be->download (&g->sess[i]->h_scene, g->sess[i]->d_scene, 4);
      if (g->be->sync()) { set_err ("device error in the scene-change pass"); return WELSHIP_ERR_UNKNOWN; }
      for (int i : need) g->sess[i]->scene_verdict (g->sess[i]->h_scene);
    }
  }
  std::vector<WhPicJob> jobs (n);
  for (int i = 0; i < n; ++i) { const int rc = g->sess[i]->begin_frame_check(); if (rc) return rc; }   // nothing begun yet: the group stays in step
  for (int i = 0; i < n; ++i) { const int rc = g->sess[i]->begin_frame (slot % g->sess[i]->ring, &jobs[i]); if (rc) return rc; }
  g->step_idr = g->sess[0]->cur_idr;
  g->mixed = false;
  for (int i = 1; i < n; ++i) if (g->sess[i]->cur_idr != g->step_idr) g->mixed = true;
  g->order.resize (n);
  g->chunk_p.assign (g->queues, 0);
  for (int q = 0; q < g->queues; ++q) {
    const int a = g->chunk_first (q), b = g->chunk_first (q + 1);
    int k = a;
    for (int i = a; i < b; ++i) if (!g->sess[i]->cur_idr) g->order[k++] = i;
    g->chunk_p[q] = k - a;
    for (int i = a; i < b; ++i) if (g->sess[i]->cur_idr) g->order[k++] = i;
    for (int j = a; j < b; ++j) g->h_jobs[j] = jobs[g->order[j]];
    g->be->select_queue (q);
    g->be->upload (g->d_jobs + a, g->h_jobs.data() + a, sizeof (WhPicJob) * (b - a));
  }
  return WELSHIP_OK;
}

int WelsHipGroupRunDevice (WelsHipEncoderGroup* g, int wait) {
  if (!g) return WELSHIP_ERR_INIT_PARA;
  SessionCore& c0 = *g->sess[0];
  for (int q = 0; q < g->queues; ++q) {
    const int a = g->chunk_first (q), b = g->chunk_first (q + 1);
    g->be->select_queue (q);
    const int np = g->chunk_p[q], ni = (b - a) - np;
    const WhSeqParams& s = c0.seq;
    if (np) g->be->run_inter (plain_seq (s), g->d_jobs + a, np);
    if (ni) g->be->run_intra (s, g->d_jobs + a + np, ni);
    if (c0.use_compact) g->be->run_compact (s, g->d_jobs + a, b - a);
    if (s.deblock_idc != 1) g->be->run_deblock (s, g->d_jobs + a, b - a);
    if (c0.prm.uiIntraPeriod != 1) g->be->run_expand (s, g->d_jobs + a, b - a);
  }
  if (wait && g->be->sync()) { set_err ("device scheduler timed out"); return WELSHIP_ERR_UNKNOWN; }
  return WELSHIP_OK;
}

int WelsHipGroupFinish (WelsHipEncoderGroup* g, WelsHipFrameBSInfo* outs) {
  if (!g) return WELSHIP_ERR_INIT_PARA;
  const int n = (int)g->sess.size();
  const bool packed = g->sess[0]->use_compact;
  for (int i = 0; i < n; ++i) {
    SessionCore& c = *g->sess[i];
    g->be->select_queue (g->chunk_of (i));
    if (packed) g->be->download (c.h_compact_off.data(), c.d_compact_off, sizeof (uint32_t) * ((size_t)c.num_mb + 1));
    else g->be->download (c.h_records.data(), c.d_records, sizeof (WhMbRecord) * c.num_mb);
  }
  if (g->be->sync()) { set_err ("device scheduler timed out; the step was not encoded"); return WELSHIP_ERR_UNKNOWN; }
  if (packed) {        // the sizes are known now: the packed records themselves
    for (int i = 0; i < n; ++i) {
      SessionCore& c = *g->sess[i];
      const size_t bytes = c.h_compact_off[c.num_mb];
      if (bytes > c.h_compact.size()) { set_err ("corrupt record offsets"); return WELSHIP_ERR_UNKNOWN; }
      g->be->select_queue (g->chunk_of (i));
      g->be->download (c.h_compact.data(), c.d_compact, bytes);
    }
    if (g->be->sync()) { set_err ("device error while copying the records"); return WELSHIP_ERR_UNKNOWN; }
  }
  for (auto& c : g->sess) c->upload_pending = false;
  std::vector<int> rcs (n, 0);
  g->parallel (n, [&] (int t, int T) {
    const auto t0 = std::chrono::steady_clock::now();
    int k = 0;
    for (int i = t; i < n; i += T, ++k) rcs[i] = g->sess[i]->finish_frame (outs ? &outs[i] : nullptr, 0, packed);
    g->note_host_time (1, std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count(), k);
  });
  if (packed) for (auto& c : g->sess) g->packed_bytes += (double)c->h_compact_off[c->num_mb];
  // sessions whose picture hit a CAVLC level overflow are re-encoded one at a time (rare: very low QP on extreme content)
  for (int i = 0; i < n; ++i) {
    SessionCore& c = *g->sess[i];
    while (rcs[i] == WELSHIP_ERR_VLC_OVERFLOW) {
      g->be->select_queue (g->chunk_of (i));
      int slot = i;                     // the job slots are permuted (P pictures first, g->order): session i's own slot
      for (int k = 0; k < (int)g->order.size(); ++k) if (g->order[k] == i) { slot = k; break; }
      rcs[i] = reencode_after_overflow (g->be, c, g->d_jobs + slot);
      if (rcs[i]) break;
      rcs[i] = c.finish_frame (outs ? &outs[i] : nullptr, 0);
    }
  }
  for (int i = 0; i < n; ++i) if (rcs[i]) {
    // the detailed message was recorded on the worker thread that coded the session; leave one the caller can read
    set_err ("session " + std::to_string (i) + (rcs[i] == WELSHIP_ERR_MEMORY ? ": frame does not fit the reference encoder's bitstream buffer (cmMallocMemeError)"
                                                                             : ": entropy coding of the frame failed"));
    return rcs[i];
  }
  return WELSHIP_OK;
}

int WelsHipGroupEncodeFrames (WelsHipEncoderGroup* g, const WelsHipSourcePicture* srcs, WelsHipFrameBSInfo* outs) {
  if (!g || !srcs) return WELSHIP_ERR_INIT_PARA;
  if (g->pending > 0) { set_err ("a pipelined step is pending: finish it first (WelsHipGroupEncodeFramesPipelined with no pictures)"); return WELSHIP_ERR_INIT_PARA; }
  const int n = (int)g->sess.size();
  const int slot = (g->sess[0]->last_slot + 1) % g->sess[0]->ring;    // never the slot of the previous picture
  for (int i = 0; i < n; ++i) {
    const SessionCore& c = *g->sess[i];
    if (srcs[i].iPicWidth != c.prm.iPicWidth || srcs[i].iPicHeight != c.prm.iPicHeight) return WELSHIP_ERR_INIT_PARA;
  }
  // host copies into the sessions' page-locked staging buffers on the entropy threads, then all transfers queued at once
  bool pending = false;
  for (int i = 0; i < n; ++i) pending = pending || g->sess[i]->upload_pending;
  if (pending) { g->be->sync(); for (auto& c : g->sess) c->upload_pending = false; }
  g->parallel (n, [&] (int t, int T) {
    const auto t0 = std::chrono::steady_clock::now();
    int k = 0;
    for (int i = t; i < n; i += T, ++k) g->sess[i]->stage_source (&srcs[i]);
    g->note_host_time (0, std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count(), k);
  });
  for (int i = 0; i < n; ++i) { g->be->select_queue (g->chunk_of (i)); g->sess[i]->issue_upload (slot % g->sess[i]->ring); }
  int rc = WelsHipGroupBegin (g, slot);
  if (rc) return rc;
  WelsHipGroupRunDevice (g, 0);
  return WelsHipGroupFinish (g, outs);
}

// ---- pipelined frame steps: the device codes step k while the host entropy-codes step k - 1 --------------------------------------
// WelsHipGroupEncodeFrames is a chain per step: staging copy -> H2D -> kernels -> D2H -> CAVLC, each waiting for the one before; the
// device idles during the transfers and the host work, the host during the kernels (profiles/r03_e2e_copy_kernel_overlap.txt: copies
// and kernels never overlap).  Here a call SUBMITS step k -- staging, H2D on an upload queue, kernels on the compute queue behind it --
// and then FINISHES step k - 1 while the device works: D2H of its packed records on a third queue, entropy coding on the host threads.
// The stream state of every session advances at submission (SessionCore::submit_advance); a picture's bitstream comes back one call
// late.  Second buffer sets keep step k's data apart from step k - 1's (records, staging, job descriptors), and every session has a
// third reconstruction picture: when the entropy coder finds a CAVLC overflow in step k - 1 (TRY_REENCODING, rare), that picture is
// coded again on the device (its reference, step k - 2, still exists) and then its already submitted successor once more.
// Queues: 0 = kernels, WH_PIPE_UPQ = uploads, WH_PIPE_DLQ = downloads.
static int pipe_queue (int which) {            // upload, download, no second upload queue
  static const int q[3] = {1, 2, -1};          // (measured: queues 30 and 31 share a hardware queue with queue 0 -- their copies waited for its kernels; profiles/r03_pipelined_group.txt)
  return q[which];
}
#define WH_PIPE_UPQ pipe_queue (0)
#define WH_PIPE_DLQ pipe_queue (1)
#define WH_PIPE_UPQ2 pipe_queue (2)
int WelsHipGroupSetPipelined (WelsHipEncoderGroup* g, int ahead) {
  if (!g) return WELSHIP_ERR_INIT_PARA;
  if (ahead <= 0) { if (g->pending) { set_err ("submitted steps are still pending: flush first"); return WELSHIP_ERR_INIT_PARA; } return WELSHIP_OK; }
  if (ahead > WH_PIPE_MAX_AHEAD) { set_err ("pipelined groups: at most 3 steps ahead"); return WELSHIP_ERR_INIT_PARA; }
  if (g->pipelined) { if (ahead == g->depth - 1) return WELSHIP_OK; set_err ("the group is pipelined already, with a different number of steps ahead"); return WELSHIP_ERR_INIT_PARA; }
  if (g->queues != 1) { set_err ("pipelined groups use one compute queue"); return WELSHIP_ERR_UNSUPPORTED; }
  if (g->pipe_failed) { set_err ("an earlier WelsHipGroupSetPipelined failed half way: destroy the group"); return WELSHIP_ERR_INIT_PARA; }
  // from here on a failure leaves buffers behind (WelsHipGroupDestroy frees them) and the group unusable
  struct Guard { WelsHipEncoderGroup* g; bool ok = false; ~Guard() { if (!ok) g->pipe_failed = true; } } guard {g};
  const int n = (int)g->sess.size(), depth = 1 + ahead;
  const size_t off_words = (size_t)g->sess[0]->num_mb + 1;
  for (int b = 0; b < depth; ++b) {
    g->d_off_all[b] = (uint32_t*)g->be->alloc (sizeof (uint32_t) * off_words * n);
    g->h_off_all[b].assign (off_words * n, 0);
    g->be->pin_host (g->h_off_all[b].data(), sizeof (uint32_t) * off_words * n);
    g->d_jobs_n[b] = b ? (WhPicJob*)g->be->alloc (sizeof (WhPicJob) * n) : g->d_jobs;
    if (!g->d_off_all[b] || !g->d_jobs_n[b]) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    g->h_jobs_p[b].assign (n, WhPicJob());
    g->be->pin_host (g->h_jobs_p[b].data(), sizeof (WhPicJob) * n);
    g->step_ev[b] = g->be->event_create(); g->up_ev[0][b] = g->be->event_create(); g->up_ev[1][b] = g->be->event_create();
  }
  for (int i = 0; i < n; ++i) {
    uint32_t* doff[WH_PIPE_MAX_AHEAD + 1];
    uint32_t* hoff[WH_PIPE_MAX_AHEAD + 1];
    for (int b = 0; b < depth; ++b) { doff[b] = g->d_off_all[b] + off_words * i; hoff[b] = g->h_off_all[b].data() + off_words * i; }
    const int rc = g->sess[i]->enable_pipeline (ahead, doff, hoff);
    if (rc) return rc;
  }
  g->d_job_aux = (WhPicJob*)g->be->alloc (sizeof (WhPicJob));
  if (!g->d_job_aux) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
  g->dl_ev.assign (n, nullptr);
  for (int i = 0; i < n; ++i) g->dl_ev[i] = g->be->event_create();
  if (g->be->sync()) { set_err ("device error while setting up the pipelined group"); return WELSHIP_ERR_UNKNOWN; }
  g->depth = depth;
  g->pipelined = true;
  guard.ok = true;
  return WELSHIP_OK;
}

namespace {
// The submitting half of a pipelined call (the calling thread): staging copies and H2D transfers on the worker threads, then the jobs
// and the kernels.  Nothing here waits for the device beyond the transfers out of the staging set it is about to refill (`depth` steps ago).
int pipe_submit (WelsHipEncoderGroup* g, const WelsHipSourcePicture* srcs, double* tm, const std::function<double()>& now) {
  wh::Backend* be = g->be;
  const int n = (int)g->sess.size();
  SessionCore& c0 = *g->sess[0];
  const int slot = (c0.last_slot + 1) % c0.ring;
  const int sb = (int) (g->step_no % g->depth);
  const int upq[2] = {WH_PIPE_UPQ, WH_PIPE_UPQ2 >= 0 ? WH_PIPE_UPQ2 : WH_PIPE_UPQ};
  for (int i = 0; i < n; ++i) if (srcs[i].iPicWidth != g->sess[i]->prm.iPicWidth || srcs[i].iPicHeight != g->sess[i]->prm.iPicHeight) return WELSHIP_ERR_INIT_PARA;
  for (int i = 0; i < n; ++i) { const int rc = g->sess[i]->begin_frame_check(); if (rc) return rc; }      // nothing queued yet: the group stays in step
  if (g->step_no >= g->depth) { be->event_wait (g->up_ev[0][sb]); if (upq[1] != upq[0]) be->event_wait (g->up_ev[1][sb]); }     // staging set sb is free again
  tm[0] = now();
  g->parallel (n, [&] (int t, int T) {
    const auto t0 = std::chrono::steady_clock::now();
    int k = 0;
    for (int i = t; i < n; i += T, ++k) g->sess[i]->stage_and_upload (&srcs[i], sb, upq[i & 1]);
    g->note_host_time (0, std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count(), k);
  });
  be->event_record_on (upq[0], g->up_ev[0][sb]);
  if (upq[1] != upq[0]) be->event_record_on (upq[1], g->up_ev[1][sb]);
  tm[1] = now();
  // describe the pictures (P pictures first, then IDR pictures, as WelsHipGroupBegin orders them)
  std::vector<WhPicJob> jobs (n);
  for (int i = 0; i < n; ++i) {
    SessionCore& c = *g->sess[i];
    if (slot % c.ring == c.last_slot) c.prev_src_dirty = true;
    const int rc = c.begin_frame (slot % c.ring, &jobs[i]);
    if (rc) return rc;
    jobs[i].src[1] = c.planar (sb);                     // tiled into src[0] by the batch pass below
  }
  g->order.resize (n);
  int k = 0;
  for (int i = 0; i < n; ++i) if (!g->sess[i]->cur_idr) g->order[k++] = i;
  const int np = k;
  for (int i = 0; i < n; ++i) if (g->sess[i]->cur_idr) g->order[k++] = i;
  std::vector<WhPicJob>& hj = g->h_jobs_p[sb];
  for (int j = 0; j < n; ++j) hj[j] = jobs[g->order[j]];
  WhPicJob* dj = g->d_jobs_n[sb];
  be->select_queue (0);
  be->queue_wait_event (0, g->up_ev[0][sb]);            // the kernels wait (on the device) for this step's sources
  if (upq[1] != upq[0]) be->queue_wait_event (0, g->up_ev[1][sb]);
  be->upload (dj, hj.data(), sizeof (WhPicJob) * n);
  const WhSeqParams& s = c0.seq;
  be->run_src_tile_jobs (s, dj, n);
  if (np) be->run_inter (plain_seq (s), dj, np);
  if (n - np) be->run_intra (s, dj + np, n - np);
  be->run_compact (s, dj, n);
  if (s.deblock_idc != 1) be->run_deblock (s, dj, n);
  if (c0.prm.uiIntraPeriod != 1) be->run_expand (s, dj, n);
  be->event_record (g->step_ev[sb]);                    // what the download queue will wait for before it copies this step's records
  for (auto& c : g->sess) c->submit_advance();
  ++g->step_no;
  tm[2] = now();
  return WELSHIP_OK;
}

// The finishing half (its own thread while the other half submits): the offset tables of all sessions in one copy, then every session's
// packed records, an event behind each; the entropy threads start on a session as soon as its records have arrived.  Touches only what
// SessionCore::entropy_frame touches (parameter-set ids, bitstream buffers) and `fin` -- nothing the submitting half uses.
int pipe_finish (WelsHipEncoderGroup* g, WelsHipFrameBSInfo* outs, std::vector<int>& rcs, double* tm, const std::function<double()>& now) {
  wh::Backend* be = g->be;
  const int n = (int)g->sess.size();
  const int fb = g->sess[0]->fin.buf;                   // (the sessions advance in lock step)
  const size_t off_words = (size_t)g->sess[0]->num_mb + 1;
  be->queue_wait_event (WH_PIPE_DLQ, g->step_ev[fb]);   // that step's kernels, not the later steps'
  be->download_on (WH_PIPE_DLQ, g->h_off_all[fb].data(), g->d_off_all[fb], sizeof (uint32_t) * off_words * n);
  if (be->sync_queue (WH_PIPE_DLQ) || be->peek_queue_errors (0, WH_PIPE_DLQ)) { set_err ("device scheduler timed out; the step was not encoded"); return WELSHIP_ERR_UNKNOWN; }
  tm[3] = now();
  for (int i = 0; i < n; ++i) {
    SessionCore& c = *g->sess[i];
    const size_t bytes = c.hcompact_off (fb)[c.num_mb];
    if (bytes > c.hcompact (fb).size()) { set_err ("corrupt record offsets"); return WELSHIP_ERR_UNKNOWN; }
    be->download_on (WH_PIPE_DLQ, c.hcompact (fb).data(), c.dcompact (fb), bytes);
    be->event_record_on (WH_PIPE_DLQ, g->dl_ev[i]);
    g->packed_bytes += (double)bytes;
  }
  tm[4] = now();
  // thread t takes sessions t, t + T, ...: in the order their records arrive
  g->parallel (n, [&] (int t, int T) {
    double busy = 0.0;
    int k = 0;
    for (int i = t; i < n; i += T, ++k) {
      be->event_wait (g->dl_ev[i]);
      const auto t0 = std::chrono::steady_clock::now();
      rcs[i] = g->sess[i]->finish_pending (outs ? &outs[i] : nullptr);
      busy += std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count();
    }
    g->note_host_time (1, busy, k);
  });
  tm[5] = now();
  return WELSHIP_OK;
}

// One picture of session c again on the device, alone, with everything waited for: `job` as it was submitted (a fresh filter generation)
int pipe_rerun (WelsHipEncoderGroup* g, SessionCore& c, WhPicJob* job, bool idr) {
  wh::Backend* be = g->be;
  if (++c.db_gen == 0) c.db_gen = 1;
  job->db_gen = c.db_gen;
  job->src[1] = nullptr;                               // (the source is tiled already)
  be->upload (g->d_job_aux, job, sizeof (WhPicJob));
  if (idr) be->run_intra (c.seq, g->d_job_aux, 1); else be->run_inter (c.seq, g->d_job_aux, 1);
  be->run_compact (c.seq, g->d_job_aux, 1);
  if (c.seq.deblock_idc != 1) be->run_deblock (c.seq, g->d_job_aux, 1);
  if (c.prm.uiIntraPeriod != 1) be->run_expand (c.seq, g->d_job_aux, 1);
  if (be->sync()) { set_err ("device scheduler timed out; the picture was not encoded"); return WELSHIP_ERR_UNKNOWN; }
  return WELSHIP_OK;
}
}  // namespace

// srcs != NULL: submit a step with these source pictures.  When as many steps are pending as the group may run ahead, the oldest one is
// finished meanwhile: its bitstreams go to outs[] and *pFinished = 1.  srcs == NULL: finish the oldest pending step (the end of the
// streams: call until *pFinished stays 0).
int WelsHipGroupEncodeFramesPipelined (WelsHipEncoderGroup* g, const WelsHipSourcePicture* srcs, WelsHipFrameBSInfo* outs, int* pFinished) {
  if (!g || !g->pipelined) { set_err ("not a pipelined group (WelsHipGroupSetPipelined)"); return WELSHIP_ERR_INIT_PARA; }
  if (pFinished) *pFinished = 0;
  wh::Backend* be = g->be;
  const int n = (int)g->sess.size();
  const bool finish = srcs ? g->pending >= g->depth - 1 : g->pending > 0;
  // WELSHIP_PIPE_TRACE=1: where the host spends a call (ms since its start), on stderr
  static const bool trace = getenv ("WELSHIP_PIPE_TRACE") && atoi (getenv ("WELSHIP_PIPE_TRACE")) != 0;
  const auto tc0 = std::chrono::steady_clock::now();
  const std::function<double()> now = [&] () { return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - tc0).count(); };
  double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // The cheap checks of the submitting half come before anything is taken off the pending queue: a call that is refused (picture size,
  // a session that cannot begin a frame) leaves the group exactly as it was -- the oldest step is still pending and can be retrieved.
  if (srcs) {
    for (int i = 0; i < n; ++i) if (srcs[i].iPicWidth != g->sess[i]->prm.iPicWidth || srcs[i].iPicHeight != g->sess[i]->prm.iPicHeight) { set_err ("source picture size differs from the session's"); return WELSHIP_ERR_INIT_PARA; }
    for (int i = 0; i < n; ++i) { const int rc = g->sess[i]->begin_frame_check(); if (rc) return rc; }
  }
  if (finish) for (auto& c : g->sess) { c->fin = c->pendq.front(); c->pendq.pop_front(); }
  std::vector<int> rcs (n, 0);
  int frc = WELSHIP_OK, src_rc = WELSHIP_OK;
  std::string ferr;
  std::thread fth;
  if (finish) {
    if (srcs) fth = std::thread ([&] { frc = pipe_finish (g, outs, rcs, tm, now); if (frc) ferr = wh::last_error(); });
    else frc = pipe_finish (g, outs, rcs, tm, now);
    --g->pending;
  }
  const bool submitted = srcs != nullptr && (src_rc = pipe_submit (g, srcs, tm, now)) == WELSHIP_OK;
  if (fth.joinable()) { fth.join(); if (frc) set_err (ferr); }
  if (submitted) ++g->pending;
  if (trace) fprintf (stderr, "welship pipe: submit half: staging set free %.2f, staged + H2D queued %.2f, kernels queued %.2f | finish half: that step done on the device %.2f, copies queued %.2f, entropy-coded %.2f ms\n",
                      tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]);
  // (a failing submit half after the checks above is a device / memory failure: the finished step's results are still reported first, so
  // that its bitstreams and a CAVLC overflow are not lost; the submit error follows)
  if (frc) return frc;
  if (finish) {
    // CAVLC overflow (rare): that picture again with the macroblock's QP raised, then the pictures submitted after it, oldest first -- each
    // predicted from a reconstruction that has just been replaced.  Everything the device has queued is waited for first.
    for (int i = 0; i < n; ++i) {
      if (rcs[i] != WELSHIP_ERR_VLC_OVERFLOW) continue;
      SessionCore& c = *g->sess[i];
      if (be->sync()) { set_err ("device scheduler timed out"); return WELSHIP_ERR_UNKNOWN; }
      be->select_queue (0);
      const bool need_ref = c.prm.uiIntraPeriod != 1;
      while (rcs[i] == WELSHIP_ERR_VLC_OVERFLOW) {
        WhPicJob job;
        std::swap (c.cur_job, c.fin.job);                 // retry_after_overflow edits "the picture being coded"
        rcs[i] = c.retry_after_overflow (&job);
        std::swap (c.cur_job, c.fin.job);
        if (rcs[i]) break;
        job.src[1] = nullptr;
        be->upload (g->d_job_aux, &job, sizeof (job));
        run_device_step (be, c.seq, g->d_job_aux, 1, c.fin.idr, need_ref, true);
        be->download (c.h_records.data(), c.d_records, sizeof (WhMbRecord) * c.num_mb);
        if (be->sync()) { set_err ("device scheduler timed out; the picture was not encoded"); rcs[i] = WELSHIP_ERR_UNKNOWN; break; }
        rcs[i] = c.entropy_frame (outs ? &outs[i] : nullptr, 0, false, c.fin.idr, c.fin.frame_num, nullptr, nullptr);
      }
      if (!c.h_mb_ctl.empty()) memset (c.h_mb_ctl.data(), 0, sizeof (WhMbCtl) * c.h_mb_ctl.size());   // the QP map belonged to that picture only
      c.qp_map_in_use = false;
      for (size_t k = 0; k < c.pendq.size() && rcs[i] == WELSHIP_OK; ++k) rcs[i] = pipe_rerun (g, c, &c.pendq[k].job, c.pendq[k].idr);
      if (!c.pendq.empty()) c.cur_job = c.pendq.back().job;
    }
    for (int i = 0; i < n; ++i) if (rcs[i]) {
      set_err ("session " + std::to_string (i) + (rcs[i] == WELSHIP_ERR_MEMORY ? ": frame does not fit the reference encoder's bitstream buffer (cmMallocMemeError)"
                                                                               : ": entropy coding of the frame failed"));
      return rcs[i];
    }
    if (pFinished) *pFinished = 1;
  }
  if (srcs && src_rc) return src_rc;
  return WELSHIP_OK;
}

int WelsHipGroupGetReconFrame (WelsHipEncoderGroup* g, int session, uint8_t* dst, size_t bytes) {
  if (!g || session < 0 || session >= (int)g->sess.size()) return WELSHIP_ERR_INIT_PARA;
  return g->sess[session]->copy_recon (dst, bytes);
}

const char* WelsHipGroupBackendName (WelsHipEncoderGroup* g) { return g ? g->be->name() : "none"; }

// Advance every session by one frame step on the device only (no D2H, no entropy coding): what the
// hot-path benchmark times.  Stream state (frame type, reference swap) advances exactly as in Finish.
int WelsHipGroupStepDeviceOnly (WelsHipEncoderGroup* g, int slot) {
  int rc = WelsHipGroupBegin (g, slot);
  if (rc) return rc;
  WelsHipGroupRunDevice (g, 0);
  for (auto& s : g->sess) {
    s->pic[s->cur].is_p = !s->cur_idr; s->have_recon = true;
    ++s->frame_index; s->frame_num = (s->frame_num + 1) & 0x7fff; s->cur = s->next_of (s->cur);
  }
  return WELSHIP_OK;
}

// Hot-path benchmark: `warmup` untimed + `steps` timed device-only frame steps, source slot cycling
// ping-pong over the resident ring.  out_ms[0] = total, out_ms[1..3] = mode-decision / deblocking / border-expansion
// passes summed over the timed steps -- HIP events on device queue 0 (with several queues: that queue's own kernels,
// which then overlap the other queues' work; the wall-clock around the call is the throughput measure).
int WelsHipGroupBench (WelsHipEncoderGroup* g, int steps, int warmup, double* out_ms) {
  if (!g || steps < 1 || !out_ms) return WELSHIP_ERR_INIT_PARA;
  wh::Backend* be = g->be;
  const int ring = g->sess[0]->ring;
  const int n = (int)g->sess.size();
  auto slot_of = [&] (int i) { if (ring == 1) return 0; const int period = 2 * (ring - 1); const int k = i % period; return k < ring ? k : period - k; };
  int fi = 0;
  for (int i = 0; i < warmup; ++i) { int rc = WelsHipGroupStepDeviceOnly (g, slot_of (fi++)); if (rc) return rc; }
  if (be->sync()) { set_err ("device scheduler timed out during warm-up"); return WELSHIP_ERR_UNKNOWN; }
  std::vector<void*> ev ((size_t)steps * 4 + 1);
  for (auto& e : ev) e = be->event_create();
  const WhSeqParams& s = g->sess[0]->seq;
  const bool need_ref = g->sess[0]->prm.uiIntraPeriod != 1;
  auto drop_events = [&] { for (auto& e : ev) be->event_destroy (e); };
  for (int i = 0; i < steps; ++i) {
    int rc = WelsHipGroupBegin (g, slot_of (fi++));
    if (rc) { be->sync(); drop_events(); return rc; }
    if (g->mixed) { be->sync(); drop_events(); set_err ("the benchmark entry point needs every session on the same frame type"); return WELSHIP_ERR_UNKNOWN; }
    for (int q = g->queues - 1; q >= 0; --q) {      // queue 0 last: it carries the events
      const int a = g->chunk_first (q), cnt = g->chunk_first (q + 1) - a;
      be->select_queue (q);
      if (q == 0) be->event_record (ev[i * 4 + 0]);
      if (g->step_idr) be->run_intra (s, g->d_jobs + a, cnt); else be->run_inter (plain_seq (s), g->d_jobs + a, cnt);
      if (q == 0) be->event_record (ev[i * 4 + 1]);
      if (s.deblock_idc != 1) be->run_deblock (s, g->d_jobs + a, cnt);
      if (q == 0) be->event_record (ev[i * 4 + 2]);
      if (need_ref) be->run_expand (s, g->d_jobs + a, cnt);
      if (q == 0) be->event_record (ev[i * 4 + 3]);
    }
    for (auto& c : g->sess) { c->pic[c->cur].is_p = !c->cur_idr; c->have_recon = true; ++c->frame_index; c->frame_num = (c->frame_num + 1) & 0x7fff; c->cur = c->next_of (c->cur); }
  }
  be->select_queue (0);
  be->event_record (ev[(size_t)steps * 4]);
  const int timed_out = be->sync();
  out_ms[0] = be->event_elapsed_ms (ev[0], ev[(size_t)steps * 4]);
  out_ms[1] = out_ms[2] = out_ms[3] = 0.0;
  for (int i = 0; i < steps; ++i) {
    out_ms[1] += be->event_elapsed_ms (ev[i * 4 + 0], ev[i * 4 + 1]);
    out_ms[2] += be->event_elapsed_ms (ev[i * 4 + 1], ev[i * 4 + 2]);
    out_ms[3] += be->event_elapsed_ms (ev[i * 4 + 2], ev[i * 4 + 3]);
  }
  drop_events();
  if (timed_out) { set_err ("device scheduler timed out during the benchmark; the timings are invalid"); return WELSHIP_ERR_UNKNOWN; }
  return WELSHIP_OK;
}

// Host share of the complete frame steps so far: the Amdahl term of the end-to-end rate (bench.py reports it).
int WelsHipGroupHostStats (WelsHipEncoderGroup* g, double* out4) {
  if (!g || !out4) return WELSHIP_ERR_INIT_PARA;
  std::lock_guard<std::mutex> l (g->stat_mu);
  out4[0] = g->host_pics[0] ? g->host_ms[0] / (double)g->host_pics[0] : 0.0;
  out4[1] = g->host_pics[1] ? g->host_ms[1] / (double)g->host_pics[1] : 0.0;
  out4[2] = (double)g->host_pics[1];
  out4[3] = g->host_pics[1] ? g->packed_bytes / (double)g->host_pics[1] : 0.0;
  return WELSHIP_OK;
}

// Developer aid: in-kernel phase profiling.  Enable -> subsequent steps accumulate cycle counters; Read copies the
// 32 counters (16 cycle sums + 16 hit counts) and clears them.
int WelsHipGroupProfile (WelsHipEncoderGroup* g, int enable, unsigned long long* out64) {
  if (!g) return WELSHIP_ERR_INIT_PARA;
  wh::Backend* be = g->be;
  unsigned long long*& prof = g->sess[0]->seq.prof;
  const size_t bytes = 2 * 64 * 32 * 8 + 64 + 256 * 8;  // two kernels (mode decision, deblocking) x 64 banks (WH_PROF_MARK) + the wave-lifetime words of k_inter_pool
  if (enable && !prof) { prof = (unsigned long long*)be->alloc (bytes); be->fill (prof, 0, bytes); be->sync(); }
  if (out64 && prof) {
    std::vector<unsigned long long> h (2 * 64 * 32 + 8 + 256);
    be->download (h.data(), prof, bytes); be->sync(); be->fill (prof, 0, bytes); be->sync();
    for (int k = 0; k < 2; ++k)
      for (int i = 0; i < 32; ++i) { out64[k * 32 + i] = 0; for (int b = 0; b < 64; ++b) out64[k * 32 + i] += h[(size_t)k * 2048 + (size_t)b * 32 + i]; }
    // mode-decision launches since the last read, in 100 MHz ticks (spare slots of the deblocking half): [44] first wave's start to
    // the last wave's end, [45] sum of the wave lifetimes, [46] waves
    out64[44] = h[4097] ? h[4097] - ~h[4096] : 0; out64[45] = h[4098]; out64[46] = h[4099];
  }
  if (!enable && prof) { be->free (prof); prof = nullptr; }
  return WELSHIP_OK;
}

}  // extern "C"
