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
#include <string.h>
#include <stddef.h>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <deque>
#include <string>
#include <thread>
#include <chrono>
#include <vector>
#include <queue>
#include <algorithm>
#include <stdlib.h>
#include "../../../include/welship.h"
#include "backend.h"
#include "../common/mb_order.h"
#include "../common/gom_rc.h"
#include "entropy_cavlc.h"
#include "headers.h"
#include "../common/compact.h"

namespace wh {
Backend* create_default_backend (int device, const char** err);   // provided by the HIP lib or the test build
}

#define WH_PIPE_MAX_AHEAD 3            // pipelined groups: the device runs at most this many frame steps ahead of the entropy coder
static thread_local std::string g_last_error;
static void set_err (const std::string& s) { g_last_error = s; }

namespace {

inline int align_up (int v, int a) { return (v + a - 1) / a * a; }

// (the unfiltered reconstruction lives in macroblock-contiguous blocks, WhPicJob::rec_blk; the planar in-place layout of rounds 1-3 measured
//  6.07 x against 4.03 x the algorithmic traffic and lost its switch in round 5: profiles/r04_pmc_traffic_unfiltered_recon_in_blocks.txt)
static bool rec_blocks_on() { return true; }

struct DevPicture {            // one padded reconstruction buffer + its tiled twin (same allocation) + its MB state
  uint8_t* base = nullptr;
  uint8_t* plane[3] = {nullptr, nullptr, nullptr};   // pixel (0,0)
  uint8_t* tiles[2] = {nullptr, nullptr};            // WH_TILE_*: luma, Cb|Cr -- written when the picture becomes a reference (run_expand)
  // the twin lies behind the planar picture: `planar` = bytes of the padded planes incl. the 2 x 64 guard bytes, `rec_y` = luma plane
  void place_tiles (size_t planar, size_t rec_y) { tiles[0] = base + ((planar + 255) & ~ (size_t)255); tiles[1] = tiles[0] + rec_y; }
  static size_t alloc_bytes (size_t planar) { return ((planar + 255) & ~ (size_t)255) + (planar - 128); }
  WhMbState* mbs = nullptr;
  bool is_p = false;
};

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
  std::vector<uint8_t> h_compact;
  std::vector<uint32_t> h_compact_off;
  bool use_compact = false;
  uint32_t* d_order = nullptr;
  int32_t* d_bands = nullptr;
  uint32_t* d_dbflags = nullptr;
  uint32_t db_gen = 0;
  size_t rec_alloc_bytes = 0, src_bytes = 0, ysz = 0, csz = 0;
  std::vector<uint8_t> h_src;
  std::vector<WhMbRecord> h_records;
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
  std::vector<WhMbCtl> h_mb_ctl;
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
  std::vector<uint8_t> h_compact_n[WH_PIPE_MAX_AHEAD];
  std::vector<uint8_t> h_src_n[WH_PIPE_MAX_AHEAD];
  uint8_t* d_planar_n[WH_PIPE_MAX_AHEAD] = {};         // upload targets (the batch tiling pass of step k reads one while later steps are uploaded)
  int pbuf = 0;                       // which set the picture being submitted uses
  uint8_t* dcompact (int b) const { return b ? d_compact_n[b - 1] : d_compact; }
  uint32_t* dcompact_off (int b) const { return x_doff[b] ? x_doff[b] : d_compact_off; }
  std::vector<uint8_t>& hcompact (int b) { return b ? h_compact_n[b - 1] : h_compact; }
  uint32_t* hcompact_off (int b) { return x_hoff[b] ? x_hoff[b] : h_compact_off.data(); }
  std::vector<uint8_t>& hsrc (int b) { return b ? h_src_n[b - 1] : h_src; }
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
    std::vector<int32_t> bands (3 * (size_t) (mb_h + s.num_slices) + 1);
    const int brows = WH_DB_BAND_ROWS;
    const bool by_slice = true;           // bands confined to slices also with idc 0 (profiles/r03_deblock_bands.txt)
    const int nb = wh_build_db_bands (mb_w, mb_h, s.num_slices, s.slice_first_mb, s.deblock_idc, brows, bands.data(), (int)bands.size(), by_slice);
    if (nb < 1) { set_err ("deblocking band table"); release(); return WELSHIP_ERR_UNKNOWN; }
    for (int b = 0; b < nb; ++b) wh_build_mb_order (mb_w, bands[b], bands[b + 1], order.data() + 2 * (size_t)num_mb + bands[b]);
    std::vector<uint32_t> order32 (order.begin(), order.end());      // 32-bit on the device (scalar loads)
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
    std::vector<uint8_t> tmp (rec_alloc_bytes + 128);
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
  std::vector<uint32_t> h_off_all[WH_PIPE_MAX_AHEAD + 1];
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

const char* WelsHipGetLastError (void) { return g_last_error.c_str(); }

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
    if (srcs) fth = std::thread ([&] { frc = pipe_finish (g, outs, rcs, tm, now); if (frc) ferr = g_last_error; });
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

// ---------------------------------------------------------------------------------- explicit frame API (include/welship.h 2b)
// What the SWelsFuncPtrList hooks of the patched reference call: the reference owns the stream (frame types, reference
// lists, rate control, entropy coding); this side owns device twins of its pictures and runs the per-macroblock passes.
//
// All frame contexts of a process that live on the same device share one backend (allocator + queues) and BATCH their
// pictures.  Pictures with identical sequence parameters and type form a KEY (the 720p P pictures of every session; each
// layer size of simulcast sessions); every key has a queue of its own.  A thread that submits a picture while nobody is
// launching for that key becomes its leader, takes everything pending for the key (pictures other sessions' threads
// submitted while the key's previous batch was on the device, or within a short gathering window), issues ONE launch set,
// waits for that queue and wakes the others.  Several sessions therefore cost the device one latency chain per batch
// instead of one per session, and keys never wait for each other (a mixed batch would last as long as its largest
// pictures: measured, 11 instead of 35 frames/s per simulcast session).  A single session is a batch of one.  Pictures with
// GOM-level QP (MB ranges) run on their own.
namespace {

struct FrameLayout {           // processing-order / deblocking-band tables on the device, shared by the contexts that use them
  int mb_w = 0, mb_h = 0, idc = -1;
  std::vector<int32_t> slices;
  uint32_t* d_order = nullptr;
  int32_t* d_bands = nullptr;
  int nb = 0, max_mbs = 0, max_rows = 0;
};

struct FrameItem;
struct FrameLane {             // one launch set in flight: its queue and its job descriptors
  int queue = 0;
  bool busy = false;
  WhPicJob* d_jobs = nullptr;
  int jobs_cap = 0;
  std::vector<WhPicJob> h_jobs;          // page-locked
  void* tail_ev = nullptr;               // marks "the records of this launch set are on the host": what the callers wait for (frame_run_batch)
  std::vector<uint32_t> h_err;           // page-locked: the queue's error words at that point
  // ... and what nobody waits for at once: the launch set's deblocking pass / border expansion ("tail").  Its verdict -- the queue's error words
  // AFTER the expansion -- lands in one of two slots (launch sets alternate), and every context of the launch set is listed until its next call
  // has looked at the slot (WelsHipFrameCtx::check_tail): a time-out inside the tail fails the picture's OWN context at its next call, not
  // whichever launch set synchronises this queue next.
  void* tail_done_ev[2] = {nullptr, nullptr};
  std::vector<uint32_t> h_err_tail;      // page-locked, 2 x 4 words
  unsigned tail_slot = 0;
  std::vector<WelsHipFrameCtx*> tail_ctxs;      // contexts with an unverified tail on this lane's queue
  void fail_tails();                     // an error was seen on this queue: every listed context's reference picture is suspect
};
#define WH_FRAME_LANES 2
struct FrameKey {              // pictures that can share a launch: same sequence parameters, type and passes
  WhSeqParams seq;
  bool is_p = false, qp_map = false, expand = false;
  int queue = 0;                          // where the pictures' inputs are uploaded
  std::vector<FrameItem*> pending;
  // Two launch sets of a key may be on the device at once (each lane a queue of its own): sessions whose host work per picture is
  // long and uneven (screen content: scroll / scene detection, feature lists) do not arrive within one gathering window, and a
  // second batch that had to wait for the first one's whole latency chain halved the rate (16 screen sessions: 352 frames/s)
  FrameLane lane[WH_FRAME_LANES];
  int gathering = 0;                      // leaders currently collecting their batch
  FrameItem* next_leader = nullptr;
  int free_lane() const { for (int i = 0; i < WH_FRAME_LANES; ++i) if (!lane[i].busy) return i; return -1; }
};
struct FrameShared {
  std::mutex mu;
  std::condition_variable cv;
  wh::Backend* be = nullptr;
  int device = 0, users = 0;
  std::vector<std::unique_ptr<FrameLayout>> layouts;
  std::vector<std::unique_ptr<FrameKey>> keys;
  // how long a leader waits for the other threads that have been submitting pictures lately.  Sessions that once end up in
  // different batches stay out of phase for good (each waits for the other's batch); one wait of about a picture's host
  // work merges them, after which they submit together and nobody waits
  int gather_us = 2000;
  std::vector<WelsHipFrameCtx*> ctxs;
  long batches = 0, batched_pictures = 0;
  // WELSHIP_TRACE: per batch size, how many batches and how long they took on the device / spent gathering
  std::vector<long> stat_n; std::vector<double> stat_dev_ms, stat_gather_ms, stat_launch_ms;
  double stat_submit_ms = 0.0;
};

std::mutex g_frame_registry_mu;
std::vector<FrameShared*> g_frame_shared;

}  // namespace

struct WelsHipFrameCtx {
  FrameShared* sh = nullptr;
  wh::Backend* be = nullptr;
  int w = 0, h = 0, mb_w = 0, mb_h = 0, num_mb = 0;
  WhSeqParams seq;
  std::vector<DevPicture> pics;
  size_t rec_alloc_bytes = 0, rec_y = 0, rec_c = 0, ysz = 0, csz = 0, src_bytes = 0;
  uint8_t* d_src = nullptr;              // the source picture being coded, macroblock-tiled (WH_SRC_*): one of src_pool ...
  uint8_t* d_src_planar = nullptr;       // ... and where an upload lands before the device rearranges it
  // Source pictures stay on the device for the pre-analysis of later pictures (WelsHipFrameVaa): a small pool keyed by the caller's
  // luma pointer (the reference rotates a fixed set of SPicture buffers per layer), least recently used slot replaced
  struct SrcSlot { const void* key = nullptr; uint8_t* d = nullptr; uint64_t stamp = 0, luma_sum = 0; };
  std::vector<SrcSlot> src_pool;
  uint64_t src_clock = 0;
  const void* fresh_key = nullptr;       // the picture WelsHipFrameVaa has just uploaded: the FrameEncode that follows finds it resident
  uint8_t* d_vaa_out = nullptr;          // pre-analysis results: sad8x8 | sd8x8 | sum16 | sqsum16 | ssd16 | mad8x8 ([mb] each, 48 bytes per MB)
  std::vector<uint8_t> h_vaa_out;        // page-locked
  int src_find (const void* key) const { for (size_t i = 0; i < src_pool.size(); ++i) if (src_pool[i].key == key) return (int)i; return -1; }
  int src_take (const void* key) {       // the slot of `key`, or the least recently used one (its content is then stale: the caller uploads)
    int k = src_find (key);
    if (k < 0) { k = 0; for (size_t i = 1; i < src_pool.size(); ++i) if (src_pool[i].stamp < src_pool[k].stamp) k = (int)i; }
    src_pool[k].key = key; src_pool[k].stamp = ++src_clock;
    return k;
  }
  // A position-weighted 64-bit sum of the MB-aligned luma area of a host picture: what tells whether the copy a pool slot holds is
  // still what the caller's buffer contains (a buffer's address alone does not: the reference rotates and reuses its pictures)
  uint64_t luma_checksum (const uint8_t* y, int32_t stride) const {
    uint64_t s0 = 0, s1 = 0, k = 1;
    for (int r = 0; r < mb_h * 16; ++r) {
      const uint8_t* row = y + (size_t)r * stride;
      for (int i = 0; i < mb_w * 16; i += 16, k += 2) { uint64_t a, b; memcpy (&a, row + i, 8); memcpy (&b, row + i + 8, 8); s0 += a * k; s1 += b * (k + 0x9E3779B97F4A7C15ull); }
    }
    return s0 ^ (s1 << 1 | s1 >> 63);
  }
  // stage a host picture (MB-aligned area of the caller's planes) into h_src; no device call
  void stage_planes (const uint8_t* const p[3], const int32_t stride[3]) {
    uint8_t* y = h_src.data();
    uint8_t* u = y + ysz;
    uint8_t* v = u + csz;
    for (int r = 0; r < mb_h * 16; ++r) memcpy (y + (size_t)r * seq.src_stride_y, p[0] + (size_t)r * stride[0], (size_t)mb_w * 16);
    for (int r = 0; r < mb_h * 8; ++r) {
      memcpy (u + (size_t)r * seq.src_stride_c, p[1] + (size_t)r * stride[1], (size_t)mb_w * 8);
      memcpy (v + (size_t)r * seq.src_stride_c, p[2] + (size_t)r * stride[2], (size_t)mb_w * 8);
    }
  }
  std::vector<uint8_t> h_src;
  WhMbRecord* d_records = nullptr;
  uint8_t* d_rec_blk = nullptr;       // (SessionCore::d_rec_blk)
  // what the last WelsHipFrameVaa call left on the device for WelsHipFrameBgd: the pair's keys and pool slots, whether the background statistics were computed
  const void* vaa_cur_key = nullptr; const void* vaa_ref_key = nullptr; int vaa_cslot = -1, vaa_rslot = -1, vaa_queue = 0; bool vaa_has_bgd = false;
  uint64_t vaa_cur_sum = 0, vaa_ref_sum = 0;      // what the two slots held when the statistics were made (SrcSlot::luma_sum): a reused slot fails WelsHipFrameBgd
  int8_t* d_bgd_calc = nullptr;       // WelsHipFrameBgd's result (one flag per macroblock)
  std::vector<int8_t> h_bgd_calc;
  uint8_t* d_skew = nullptr;          // pre-analysis of a picture whose width is no multiple of 16: the two luma planes at the caller's stride (WelsHipFrameVaa)
  std::vector<uint8_t> h_skew;
  size_t skew_bytes = 0;
  std::vector<WhMbRecord> h_records;
  // packed records of whole-picture calls (WelsHipFrameJob::bPackedRecords; common/compact.h): device stream + offsets, page-locked host copies.
  // The host copy is brought back in one go up to `compact_est` bytes (a little more than the previous picture's size); the rare rest follows.
  uint8_t* d_compact = nullptr;
  uint32_t* d_compact_off = nullptr;
  std::vector<uint8_t> h_compact;
  std::vector<uint32_t> h_coff;
  size_t compact_est = 0, compact_got = 0;
  WelsHipPackedRecords packed_view = {nullptr, nullptr};
  std::vector<uint8_t> h_pic;
  // page-locked staging for the small per-picture arrays of the caller (pageable copies block on the queue: with the shared
  // lock held that stalls every other session): VAA SADs | pSadCost in | inter-layer hints | background flags, and pSadCost out
  std::vector<uint8_t> h_aux, h_sad_out;
  size_t aux_vaa = 0, aux_sad = 0, aux_il = 0, aux_bgd = 0;
  int h_pic_of = -1;                     // the device picture h_pic holds (copied back with the batch), or -1
  FrameKey* last_key = nullptr;          // the key of this context's last picture, and when it was submitted (FrameShared::gather_us)
  std::chrono::steady_clock::time_point last_submit;
  int queue() const { return last_key ? last_key->queue : 0; }
  int tail_queue = -1;                   // the queue on which this context's last picture is still being deblocked / expanded (frame_run_batch), or -1
  FrameLane* tail_lane = nullptr;        // ... the lane whose verdict slot tail_slot will hold that pass's error words (checked by the next call: check_tail)
  int tail_slot = 0;
  bool tail_failed = false;              // the last picture's deblocking / expansion did not complete: its reconstruction cannot be predicted from
  // whatever touches this context's pictures on another queue comes after that tail (the caller has selected its queue)
  void join_tail (int on_queue) { if (tail_queue >= 0 && tail_queue != on_queue) be->queue_wait (tail_queue); }
  // The verdict of the last picture's deblocking pass / border expansion, at this context's next call (advisor finding, round 4: that pass runs
  // after the callers were released, and its time-out used to fail whichever launch set synchronised the queue next).  Waits for the pass
  // (normally long over: the caller has entropy-coded a picture meanwhile), reads the error words copied out behind it.  Called WITHOUT sh->mu.
  // `intra`: the call codes an I picture, which predicts from nothing -- once the failure has been reported, such a call clears it; until then it is reported to EVERY call
  // on this context (slice tasks of a size-limited picture call concurrently: with a flag that the first caller consumed, the others coded
  // against the bad reference -- advisor finding, round 5).  sh->mu is held throughout except around the wait itself.
  int check_tail (bool intra) {
    std::unique_lock<std::mutex> lock (sh->mu);
    if (tail_lane) {
      FrameLane* TL = tail_lane;
      const int slot = tail_slot;
      void* ev = TL->tail_done_ev[slot];
      lock.unlock();
      be->event_wait (ev);
      lock.lock();
      const uint32_t* e = TL->h_err_tail.data() + 4 * slot;
      if (tail_lane == TL && tail_slot == slot) {      // (an error seen meanwhile on that queue has cleared the list and set tail_failed already; a later tail re-armed the slot)
        if (e[0] | e[1] | e[2] | e[3]) { (void)be->sync_queue (TL->queue); TL->fail_tails(); }
        else { TL->tail_ctxs.erase (std::remove (TL->tail_ctxs.begin(), TL->tail_ctxs.end(), this), TL->tail_ctxs.end()); tail_lane = nullptr; }
      }
    }
    if (!tail_failed) return 0;
    if (intra && tail_reported) { tail_failed = false; tail_reported = false; return 0; }     // the caller has been told and now starts afresh
    tail_reported = true;
    return 1;
  }
  bool tail_reported = false;
  uint32_t* d_dbflags = nullptr;
  uint32_t db_gen = 0;
  WhMbCtl* d_mb_ctl = nullptr;
  std::vector<WhMbCtl> h_mb_ctl;
  int32_t* d_sad_cost0 = nullptr;        // the layer's pSadCost[0] array (persists across pictures)
  int32_t* d_sad_cost0_new = nullptr;    // the copy the picture being coded writes (WhPicJob::sad_cost0_out): size-limited slices swap it in when the picture
                                         // is complete, whole-picture calls when the NEXT picture begins (a bRetry pass reads the previous picture's again)
  bool sad_swap_pending = false;
  int32_t* d_vaa = nullptr;
  int8_t* d_bgd = nullptr;
  int16_t* d_il = nullptr;
  WhPicJob* d_job = nullptr;
  FrameLayout* layout = nullptr;
  // screen-content P pictures (WelsHipFrameJob::pScreen): device copies of the pre-processing's arrays, the reference's SOURCE chroma,
  // the reference picture's feature lists, the per-slice cost chain / feature-search statistics (allocated with the first such picture)
  WhSccJob* d_scc = nullptr;
  uint8_t* d_scc_idc = nullptr;
  uint8_t* d_scc_ori = nullptr;
  uint32_t* d_scc_chain = nullptr;       // [WH_MAX_SLICES][4] chain, then [WH_MAX_SLICES] cost-down sums
  uint32_t* d_scc_order = nullptr;       // WH_SEQ_CHAIN: the picture's processing order [num_mb] | chain_prev [num_mb]
  uint32_t* d_scc_chain_mb = nullptr;    // size-limited slices of such a picture: the chain per macroblock (WhSccJob::chain_mb)
  WhGomRc* d_gom_rc = nullptr;           // GOM-level rate control inside the kernel: inputs + state of the picture in flight
  std::vector<uint8_t> h_gom;            // page-locked staging: WhGomRc | order [num_mb] | dependency [num_mb]
  uint32_t* d_scc_lists = nullptr;       // times[list] | start[list]
  uint16_t* d_scc_loc = nullptr;
  size_t scc_list_cap = 0, scc_loc_cap = 0;
  std::vector<uint8_t> h_scc;            // page-locked staging: idc | source chroma | times | start | locations
  size_t h_scc_cap = 0;
  std::vector<uint8_t> h_scc_small;      // page-locked: the WhSccJob being uploaded | the per-slice statistics coming back (a pageable
                                         // buffer would make the copy synchronous -- under the shared lock, behind the other sessions' kernels)
  WhSccJob& scc_job() { return * (WhSccJob*)h_scc_small.data(); }
  uint32_t* scc_down() { return (uint32_t*) (h_scc_small.data() + 256); }
  bool scc_active = false;               // the picture in flight is a screen-content P picture

  // caller holds sh->mu
  void release_locked() {
    if (!be) return;
    be->sync();
    for (auto& p : pics) { if (p.base) be->free (p.base); if (p.mbs) be->free (p.mbs); }
    pics.clear();
    for (auto& sl : src_pool) if (sl.d) be->free (sl.d);
    src_pool.clear();
    d_src = nullptr;
    if (!h_vaa_out.empty()) be->unpin_host (h_vaa_out.data());
    if (!h_skew.empty()) be->unpin_host (h_skew.data());
    if (!h_bgd_calc.empty()) be->unpin_host (h_bgd_calc.data());
    if (!h_compact.empty()) be->unpin_host (h_compact.data());
    if (!h_coff.empty()) be->unpin_host (h_coff.data());
    if (d_compact) be->free (d_compact);
    if (d_compact_off) be->free (d_compact_off);
    d_compact = nullptr; d_compact_off = nullptr;
    void* ptrs[] = {d_vaa_out, d_bgd_calc, d_skew, d_src_planar, d_records, d_rec_blk, d_dbflags, d_mb_ctl, d_sad_cost0, d_vaa, d_bgd, d_il, d_job, d_scc, d_scc_idc, d_scc_ori, d_scc_chain, d_scc_lists, d_scc_loc, d_scc_order, d_scc_chain_mb, d_gom_rc, d_sad_cost0_new};
    if (!h_gom.empty()) be->unpin_host (h_gom.data());
    if (!h_scc.empty()) be->unpin_host (h_scc.data());
    if (!h_scc_small.empty()) be->unpin_host (h_scc_small.data());
    for (void* p : ptrs) if (p) be->free (p);
    if (!h_records.empty()) be->unpin_host (h_records.data());
    if (!h_src.empty()) be->unpin_host (h_src.data());
    if (!h_pic.empty()) be->unpin_host (h_pic.data());
    if (!h_aux.empty()) be->unpin_host (h_aux.data());
    if (!h_sad_out.empty()) be->unpin_host (h_sad_out.data());
    be = nullptr;
  }

  // (re)select the processing-order and deblocking-band tables for this slice layout / filter mode; caller holds sh->mu
  int set_layout (int n, const int32_t* first, int idc) {
    if (layout && layout->idc == idc && (int)layout->slices.size() == n + 1 && memcmp (layout->slices.data(), first, sizeof (int32_t) * (n + 1)) == 0) return WELSHIP_OK;
    if (n < 1 || n > WH_MAX_SLICES || first[0] != 0 || first[n] != num_mb) { set_err ("invalid slice layout"); return WELSHIP_ERR_INIT_PARA; }
    for (int i = 0; i < n; ++i) if (first[i + 1] <= first[i]) { set_err ("invalid slice layout"); return WELSHIP_ERR_INIT_PARA; }
    FrameLayout* L = nullptr;
    for (auto& up : sh->layouts)
      if (up->mb_w == mb_w && up->mb_h == mb_h && up->idc == idc && (int)up->slices.size() == n + 1 && memcmp (up->slices.data(), first, sizeof (int32_t) * (n + 1)) == 0) { L = up.get(); break; }
    if (!L) {
      std::unique_ptr<FrameLayout> up (new FrameLayout());
      up->mb_w = mb_w; up->mb_h = mb_h; up->idc = idc; up->slices.assign (first, first + n + 1);
      std::vector<uint16_t> order ((size_t)num_mb * 3);
      for (int i = 0; i < n; ++i) wh_build_mb_order (mb_w, first[i], first[i + 1], order.data() + first[i]);
      wh_build_mb_order (mb_w, 0, num_mb, order.data() + num_mb);
      std::vector<int32_t> bands (3 * (size_t) (mb_h + n) + 1);
      const int nb = wh_build_db_bands (mb_w, mb_h, n, first, idc, WH_DB_BAND_ROWS, bands.data(), (int)bands.size());
      if (nb < 1) { set_err ("deblocking band table"); return WELSHIP_ERR_UNKNOWN; }
      for (int b = 0; b < nb; ++b) wh_build_mb_order (mb_w, bands[b], bands[b + 1], order.data() + 2 * (size_t)num_mb + bands[b]);
      up->d_order = (uint32_t*)be->alloc ((size_t)num_mb * 3 * 4);
      up->d_bands = (int32_t*)be->alloc (sizeof (int32_t) * (3 * (size_t)nb + 1 + 4));
      if (!up->d_order || !up->d_bands) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
      std::vector<uint32_t> order32 (order.begin(), order.end());
      be->upload (up->d_order, order32.data(), order32.size() * 4);
      { const int32_t whole[4] = {0, num_mb, 0, num_mb}; bands.resize (3 * (size_t)nb + 1); bands.insert (bands.end(), whole, whole + 4); }
      be->upload (up->d_bands, bands.data(), sizeof (int32_t) * (3 * (size_t)nb + 1 + 4));
      if (be->sync()) { set_err ("device error"); return WELSHIP_ERR_UNKNOWN; }
      up->nb = nb;
      for (int b = 0; b < nb; ++b) {
        up->max_mbs = std::max (up->max_mbs, bands[b + 1] - bands[b]);
        up->max_rows = std::max (up->max_rows, (bands[b + 1] - 1) / mb_w - bands[b] / mb_w + 1);
      }
      L = up.get();
      sh->layouts.push_back (std::move (up));       // tables live as long as the shared device (a handful per resolution)
    }
    layout = L;
    WhSeqParams& s = seq;
    s.num_slices = n;
    for (int i = 0; i < WH_MAX_SLICES + 1; ++i) s.slice_first_mb[i] = i <= n ? first[i] : 0;
    s.mb_order = L->d_order;
    s.db_num_bands = L->nb; s.db_bands = L->d_bands; s.db_max_mbs = L->max_mbs; s.db_max_rows = L->max_rows;
    return WELSHIP_OK;
  }
};

namespace {

struct FrameItem {             // one submitted picture, owned by the submitting thread's stack frame
  WelsHipFrameCtx* c = nullptr;
  bool packed = false;                    // the records come back packed (WelsHipFrameJob::bPackedRecords)
  WhSeqParams seq;
  WhPicJob job;
  bool is_p = false, qp_map = false, expand = false;
  int cur_pic = 0;
  int32_t* sad_dst = nullptr;
  bool done = false;
  int rc = 0;
};

void FrameLane::fail_tails() {
  for (WelsHipFrameCtx* x : tail_ctxs) { x->tail_failed = true; x->tail_lane = nullptr; }
  tail_ctxs.clear();
}

// the leader's work: the key's pending pictures in one launch set on the key's queue.  Called with sh->mu held; releases it
// while the device works.
void frame_run_batch (FrameShared* sh, FrameKey* K, FrameLane* L, std::unique_lock<std::mutex>& lock, std::vector<FrameItem*>& batch) {
  wh::Backend* be = sh->be;
  const auto t_launch0 = std::chrono::steady_clock::now();
  const int n = (int)batch.size();
  be->select_queue (L->queue);
  be->queue_wait (K->queue);        // the pictures' inputs were uploaded on the key's queue (which carries nothing else: a lane never waits for the other lane's kernels)
  for (FrameItem* x : batch) x->c->join_tail (L->queue);      // a context's previous picture may still be in its deblocking pass on the other lane's queue
  if (n > L->jobs_cap) {
    if (L->d_jobs) be->free (L->d_jobs);
    if (!L->h_jobs.empty()) be->unpin_host (L->h_jobs.data());
    L->jobs_cap = std::max (16, 2 * n);
    L->d_jobs = (WhPicJob*)be->alloc (sizeof (WhPicJob) * L->jobs_cap);
    L->h_jobs.assign (L->jobs_cap, WhPicJob());
    be->pin_host (L->h_jobs.data(), sizeof (WhPicJob) * L->jobs_cap);
  }
  int rc_all = L->d_jobs ? WELSHIP_OK : WELSHIP_ERR_MEMORY;
  if (rc_all == WELSHIP_OK) {
    bool any_packed = false;
    for (int i = 0; i < n; ++i) {
      L->h_jobs[i] = batch[i]->job;
      L->h_jobs[i].compact = batch[i]->packed ? batch[i]->c->d_compact : nullptr;
      L->h_jobs[i].compact_off = batch[i]->packed ? batch[i]->c->d_compact_off : nullptr;
      any_packed = any_packed || batch[i]->packed;
    }
    be->upload (L->d_jobs, L->h_jobs.data(), sizeof (WhPicJob) * n);
    const WhSeqParams& s = K->seq;
    if (K->is_p) {
      // camera pictures without control inputs (the usual case: rate control with several slices, or off): the promise the P kernel's
      // frame-API variant needs (WH_SEQ_NO_CTRL, common/wh_types.h)
      bool no_ctrl = s.flags == 0;
      for (int i = 0; i < n && no_ctrl; ++i) {
        const WhPicJob& q = L->h_jobs[i];
        no_ctrl = !q.il_hint && !q.mb_ctl && !q.gom_rc && !q.dyn_slice && !q.want_bits && !q.mb_end && !q.scc;
      }
      WhSeqParams sq = s;
      if (no_ctrl) sq.flags |= WH_SEQ_NO_CTRL;
      be->run_inter (sq, L->d_jobs, n);
    } else be->run_intra (s, L->d_jobs, n);
    if (K->qp_map) be->run_qp_chain (s, L->d_jobs, n);
    if (any_packed) be->run_compact (s, L->d_jobs, n);      // (pictures without a packed stream are left alone: WhPicJob::compact == NULL)
    for (FrameItem* x : batch) {
      WelsHipFrameCtx* c = x->c;
      if (x->packed) {
        c->compact_got = std::min (c->compact_est, c->h_compact.size());
        be->download (c->h_coff.data(), c->d_compact_off, sizeof (uint32_t) * (c->num_mb + 1));
        be->download (c->h_compact.data(), c->d_compact, c->compact_got);
      } else
      be->download (c->h_records.data(), c->d_records, sizeof (WhMbRecord) * c->num_mb);
      // (the reconstruction stays on the device: WelsHipFrameGetPicture fetches it when the caller asks -- PSNR, a frame dump; the
      //  dispatch-table binding's pfHipFetchRecon)
      if (x->sad_dst) be->download (c->h_sad_out.data(), x->job.sad_cost0_out ? x->job.sad_cost0_out : x->job.sad_cost0, sizeof (int32_t) * c->num_mb);
      if (c->scc_active) be->download (c->scc_down(), c->d_scc_chain + 4 * WH_MAX_SLICES, sizeof (uint32_t) * WH_MAX_SLICES);
    }
    // What the callers wait for ends HERE: their entropy coders need the records, nothing else.  The deblocking pass, the border expansion and
    // the tiled twin only matter to whatever touches these pictures next -- the next launch set with one of these contexts, a
    // reconstruction fetch -- and that is ordered behind them on the device (same queue, or WelsHipFrameCtx::join_tail from another one).
    // The host's entropy coding of a picture (4-7 ms for 1080p) thus overlaps the 2.5 ms its filtering takes.  A time-out inside the tail
    // shows up in the next call that synchronises this queue.  WELSHIP_FRAME_TAIL=0: wait for everything, as before round 4.
    const int q = L->queue;
    static const bool tail_env_off = getenv ("WELSHIP_FRAME_TAIL") && atoi (getenv ("WELSHIP_FRAME_TAIL")) == 0;
    const bool tail = !tail_env_off && (s.deblock_idc != 1 || K->expand);
    if (tail) {
      if (!L->tail_ev) { L->tail_ev = be->event_create(); L->h_err.assign (4, 0u); be->pin_host (L->h_err.data(), 16); }
      be->err_snapshot (q, L->h_err.data());
      be->event_record_on (q, L->tail_ev);
    }
    if (s.deblock_idc != 1) be->run_deblock (s, L->d_jobs, n);
    if (K->expand) be->run_expand (s, L->d_jobs, n);
    for (FrameItem* x : batch) x->c->tail_queue = tail && L->tail_ev ? q : -1;
    if (tail && L->tail_ev) {            // the tail's own verdict, for the contexts' next calls (WelsHipFrameCtx::check_tail)
      if (L->h_err_tail.empty()) { L->h_err_tail.assign (8, 0u); be->pin_host (L->h_err_tail.data(), 32); for (void*& e : L->tail_done_ev) e = be->event_create(); }
      const int slot = (int) (++L->tail_slot & 1);
      be->err_snapshot (q, L->h_err_tail.data() + 4 * slot);
      be->event_record_on (q, L->tail_done_ev[slot]);
      for (FrameItem* x : batch) {
        WelsHipFrameCtx* c = x->c;
        if (c->tail_lane && c->tail_lane != L) c->tail_lane->tail_ctxs.erase (std::remove (c->tail_lane->tail_ctxs.begin(), c->tail_lane->tail_ctxs.end(), c), c->tail_lane->tail_ctxs.end());
        if (c->tail_lane != L) L->tail_ctxs.push_back (c);
        c->tail_lane = L; c->tail_slot = slot;
      }
    }
    const double launch_ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t_launch0).count();
    const unsigned swept0 = be->errors_swept();
    lock.unlock();               // other sessions stage and queue their next pictures while the device works
    const auto t_dev0 = std::chrono::steady_clock::now();
    int bad = 0;
    if (tail && L->tail_ev) {
      be->event_wait (L->tail_ev);
      bad = L->h_err[0] != 0;
      if (bad) (void)be->sync_queue (q);         // (reports the time-out and clears the queue's error words: the next launch set starts clean)
    } else bad = be->sync_queue (q);
    if (be->errors_swept() != swept0) bad = 1;       // another thread's sync() found time-outs meanwhile: possibly this launch set's
    const double dev_ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t_dev0).count();
    lock.lock();
    // a packed stream longer than what was brought back with the batch (a picture much larger than the one before it): the rest now
    if (!bad && any_packed) {
      bool more = false;
      for (FrameItem* x : batch) if (x->packed) {
        WelsHipFrameCtx* c = x->c;
        const size_t total = c->h_coff[c->num_mb];
        if (total > c->h_compact.size()) { bad = 1; continue; }                  // (cannot be: the buffer holds the worst case)
        if (total > c->compact_got) {
          if (!more) { be->select_queue (q); more = true; }
          be->download (c->h_compact.data() + c->compact_got, c->d_compact + c->compact_got, total - c->compact_got);
        }
        c->compact_est = std::min (c->h_compact.size(), ((total + total / 4 + 4095) & ~ (size_t)4095) + 65536);
      }
      if (more) { lock.unlock(); if (be->sync_queue (q)) bad = 1; lock.lock(); }
    }
    if (bad) {                           // whatever went wrong on this queue: the pictures still in their tails on it cannot be trusted either
      for (FrameItem* x : batch) { x->c->tail_lane = nullptr; L->tail_ctxs.erase (std::remove (L->tail_ctxs.begin(), L->tail_ctxs.end(), x->c), L->tail_ctxs.end()); }      // (these fail now)
      L->fail_tails();
    }
    if ((int)sh->stat_n.size() <= n) { sh->stat_n.resize (n + 1, 0); sh->stat_dev_ms.resize (n + 1, 0.0); sh->stat_gather_ms.resize (n + 1, 0.0); sh->stat_launch_ms.resize (n + 1, 0.0); }
    ++sh->stat_n[n]; sh->stat_dev_ms[n] += dev_ms; sh->stat_launch_ms[n] += launch_ms;
    if (bad) rc_all = WELSHIP_ERR_UNKNOWN;
    ++sh->batches; sh->batched_pictures += n;
  }
  for (FrameItem* it : batch) {
    it->rc = rc_all;
    if (rc_all == WELSHIP_OK) { it->c->pics[it->cur_pic].is_p = it->is_p; it->c->h_pic_of = -1; }
    it->done = true;
  }
}

FrameKey* frame_find_key (FrameShared* sh, const WhSeqParams& s, bool is_p, bool qp_map, bool expand) {
  for (auto& k : sh->keys)
    if (k->is_p == is_p && k->qp_map == qp_map && k->expand == expand && memcmp (&k->seq, &s, sizeof (WhSeqParams)) == 0) return k.get();
  std::unique_ptr<FrameKey> k (new FrameKey());
  k->seq = s; k->is_p = is_p; k->qp_map = qp_map; k->expand = expand;
  // Queues whose numbers differ modulo 4 are served by different hardware queues (hip_backend.hip, constructor; there are four).  A key's
  // uploads go to hardware queue key % 4, both its launch sets to 3 - key % 4: the kernels of four keys (the layers of a simulcast
  // session) run beside each other, and a key's uploads never queue behind its own kernels.  Measured against the alternative "uploads /
  // launch set 1 / launch set 2 / pre-analysis of every key on hardware queues 0 / 1 / 2 / 3" (profiles/r03_stream_hardware_queues.txt):
  // 8 simulcast sessions 89-92 against 67 frames/s, one-key workloads the same.  Keys beyond the eighth share queues (they only
  // serialise, nothing breaks).
  const int kn = (int) (sh->keys.size() % 8);
  k->queue = kn;                                   // the key's own queue only carries the uploads
  for (int i = 0; i < WH_FRAME_LANES; ++i) k->lane[i].queue = 8 + 8 * i + 4 * (kn / 4) + (3 - kn % 4);
  // The first two keys of a device -- what single-layer sessions have: their I and their P pictures -- get the second launch set on a
  // hardware queue of its own (stream 16 + (2 - kn % 4) % 4 is served by another one than stream 8 + 3 - kn % 4: profiles/r03_stream_hardware_queues.txt).
  // A session that misses its batch by a moment is then coded BESIDE the batch instead of behind it, returns a moment after the others and
  // is back in their batch one picture later; on a shared queue it returned a whole latency chain late, for good (eight 1080p sessions:
  // 211-470 frames/s from run to run on one queue, 398-444 on two; profiles/r05_frame_api_second_launch_set_queue.txt).  Sessions with
  // several layers (more keys) keep both launch sets of a key on one queue: their layers' queues would collide (-4 %).
  if (kn < 2 && sh->keys.size() < 2) k->lane[1].queue = 8 + 8 + (3 - kn % 4 + 3) % 4;
  sh->keys.push_back (std::move (k));
  return sh->keys.back().get();
}

}  // namespace

extern "C" {

int WelsHipFrameCtxCreate (WelsHipFrameCtx** pp, const WelsHipFrameCfg* cfg) {
  if (!pp || !cfg) return WELSHIP_ERR_INIT_PARA;
  if (cfg->iPicWidth < 16 || cfg->iPicHeight < 16 || cfg->iPicWidth > 4096 || cfg->iPicHeight > 2304 || cfg->iNumPictures < 2 || cfg->iNumPictures > 64) {
    set_err ("invalid frame context configuration"); return WELSHIP_ERR_INIT_PARA;
  }
  FrameShared* sh = nullptr;
  {
    std::lock_guard<std::mutex> reg (g_frame_registry_mu);
    for (FrameShared* x : g_frame_shared) if (x->device == cfg->iDevice) sh = x;
    if (!sh) {
      const char* berr = nullptr;
      wh::Backend* be = wh::create_default_backend (cfg->iDevice, &berr);
      if (!be) { set_err (std::string ("no usable device backend: ") + (berr ? berr : "?")); return WELSHIP_ERR_NO_DEVICE; }
      sh = new FrameShared();
      sh->be = be; sh->device = cfg->iDevice;
      g_frame_shared.push_back (sh);
    }
    ++sh->users;
  }
  std::unique_lock<std::mutex> lock (sh->mu);
  wh::Backend* be = sh->be;
  WelsHipFrameCtx* c = new WelsHipFrameCtx();
  c->sh = sh;
  c->be = be;
  c->w = cfg->iPicWidth; c->h = cfg->iPicHeight;
  c->mb_w = (c->w + 15) >> 4; c->mb_h = (c->h + 15) >> 4; c->num_mb = c->mb_w * c->mb_h;
  WhSeqParams& s = c->seq;
  memset (&s, 0, sizeof (s));
  s.mb_w = c->mb_w; s.mb_h = c->mb_h;
  s.src_stride_y = c->mb_w * 16; s.src_stride_c = c->mb_w * 8;
  s.rec_stride_y = align_up (c->mb_w * 16 + 64, 64); s.rec_stride_c = s.rec_stride_y / 2;
  s.blk8_w = c->mb_w * 2; s.blk8_h = c->mb_h * 2;
  c->ysz = (size_t)s.src_stride_y * c->mb_h * 16; c->csz = (size_t)s.src_stride_c * c->mb_h * 8; c->src_bytes = c->ysz + 2 * c->csz;
  const int rec_h = c->mb_h * 16 + 64;
  c->rec_y = (size_t)s.rec_stride_y * rec_h; c->rec_c = (size_t)s.rec_stride_c * (rec_h / 2);
  c->rec_alloc_bytes = c->rec_y + 2 * c->rec_c;
  bool oom = false;
  auto A = [&] (size_t n) { void* p = be->alloc (n); if (!p) oom = true; return p; };
  c->pics.resize (cfg->iNumPictures);
  for (auto& d : c->pics) { d.base = (uint8_t*)A (DevPicture::alloc_bytes (c->rec_alloc_bytes + 128)); d.mbs = (WhMbState*)A (sizeof (WhMbState) * c->num_mb); }
  c->src_pool.resize (std::min (8, std::max (3, cfg->iNumPictures + 1)));
  for (auto& sl : c->src_pool) sl.d = (uint8_t*)A (c->src_bytes);
  c->d_src = c->src_pool[0].d;
  c->d_src_planar = (uint8_t*)A (c->src_bytes);
  c->d_records = (WhMbRecord*)A (sizeof (WhMbRecord) * c->num_mb);
  c->d_rec_blk = rec_blocks_on() ? (uint8_t*)A ((size_t)WH_SRC_MB_BYTES * c->num_mb) : nullptr;
  c->d_dbflags = (uint32_t*)A (sizeof (uint32_t) * c->num_mb);
  c->d_mb_ctl = (WhMbCtl*)A (sizeof (WhMbCtl) * c->num_mb);
  c->d_sad_cost0 = (int32_t*)A (sizeof (int32_t) * c->num_mb);
  c->d_vaa = (int32_t*)A (sizeof (int32_t) * 4 * c->num_mb);
  c->d_bgd = (int8_t*)A ((size_t)c->num_mb + 64);
  c->d_il = (int16_t*)A (sizeof (int16_t) * 4 * c->num_mb);
  c->d_job = (WhPicJob*)A (sizeof (WhPicJob));
  auto fail = [&] (int rc) {
    c->release_locked();
    delete c;
    lock.unlock();
    std::lock_guard<std::mutex> reg (g_frame_registry_mu);
    --sh->users;                      // the shared device stays for the next context (its allocator keeps the slabs)
    return rc;
  };
  if (oom) { set_err ("out of device memory"); return fail (WELSHIP_ERR_MEMORY); }
  for (auto& d : c->pics) {
    be->fill (d.base, 0, DevPicture::alloc_bytes (c->rec_alloc_bytes + 128));
    d.place_tiles (c->rec_alloc_bytes + 128, c->rec_y);
    d.plane[0] = d.base + 64 + (size_t)32 * s.rec_stride_y + 32;
    d.plane[1] = d.base + 64 + c->rec_y + (size_t)16 * s.rec_stride_c + 16;
    d.plane[2] = d.base + 64 + c->rec_y + c->rec_c + (size_t)16 * s.rec_stride_c + 16;
    be->fill (d.mbs, 0, sizeof (WhMbState) * c->num_mb);
  }
  be->fill (c->d_dbflags, 0, sizeof (uint32_t) * c->num_mb);
  be->fill (c->d_sad_cost0, 0, sizeof (int32_t) * c->num_mb);      // WelsMallocz (encoder_ext.cpp:1675-1677)
  be->fill (c->d_records, 0, sizeof (WhMbRecord) * c->num_mb);
  c->h_src.assign (c->src_bytes, 0);
  c->h_records.resize (c->num_mb);
  c->h_mb_ctl.resize (c->num_mb);
  be->pin_host (c->h_records.data(), sizeof (WhMbRecord) * c->num_mb);
  be->pin_host (c->h_src.data(), c->h_src.size());
  c->h_pic.resize (c->rec_alloc_bytes + 128);                 // D2H target of the reconstruction (copied back with every batch)
  be->pin_host (c->h_pic.data(), c->h_pic.size());
  c->aux_vaa = 0; c->aux_sad = (size_t)16 * c->num_mb; c->aux_il = c->aux_sad + (size_t)4 * c->num_mb; c->aux_bgd = c->aux_il + (size_t)8 * c->num_mb;
  c->h_aux.assign (c->aux_bgd + (size_t)c->num_mb + 64, 0);
  c->h_sad_out.assign ((size_t)4 * c->num_mb, 0);
  be->pin_host (c->h_aux.data(), c->h_aux.size());
  be->pin_host (c->h_sad_out.data(), c->h_sad_out.size());
  if (be->sync()) { set_err ("device error while setting up the frame context"); return fail (WELSHIP_ERR_UNKNOWN); }
  sh->ctxs.push_back (c);
  *pp = c;
  return WELSHIP_OK;
}

void WelsHipFrameCtxDestroy (WelsHipFrameCtx* c) {
  if (!c) return;
  FrameShared* sh = c->sh;
  {
    std::unique_lock<std::mutex> lock (sh->mu);
    sh->ctxs.erase (std::remove (sh->ctxs.begin(), sh->ctxs.end(), c), sh->ctxs.end());
    if (c->tail_lane) c->tail_lane->tail_ctxs.erase (std::remove (c->tail_lane->tail_ctxs.begin(), c->tail_lane->tail_ctxs.end(), c), c->tail_lane->tail_ctxs.end());
    c->release_locked();
  }
  delete c;
  std::lock_guard<std::mutex> reg (g_frame_registry_mu);
  if (--sh->users == 0) {
    if ((getenv ("WELSHIP_TRACE") || getenv ("WELSHIP_FRAME_STATS")) && sh->batches) {
      fprintf (stderr, "welship: frame API on device %d: %ld pictures in %ld batches\n", sh->device, sh->batched_pictures, sh->batches);
      for (size_t k = 1; k < sh->stat_n.size(); ++k) if (sh->stat_n[k])
        fprintf (stderr, "welship:   batches of %zu: %ld, issuing %.3f ms, device wait %.3f ms, gathering %.3f ms on average\n", k, sh->stat_n[k],
                 sh->stat_launch_ms[k] / sh->stat_n[k], sh->stat_dev_ms[k] / sh->stat_n[k], sh->stat_gather_ms[k] / sh->stat_n[k]);
      fprintf (stderr, "welship:   submitting (uploads under the lock): %.3f ms per picture\n", sh->stat_submit_ms / sh->batched_pictures);
    }
    for (auto& L : sh->layouts) { sh->be->free (L->d_order); sh->be->free (L->d_bands); }
    for (auto& K : sh->keys) for (FrameLane& L : K->lane) { if (L.d_jobs) sh->be->free (L.d_jobs); if (!L.h_jobs.empty()) sh->be->unpin_host (L.h_jobs.data()); if (L.tail_ev) sh->be->event_destroy (L.tail_ev); if (!L.h_err.empty()) sh->be->unpin_host (L.h_err.data());
      for (void* e : L.tail_done_ev) if (e) sh->be->event_destroy (e); if (!L.h_err_tail.empty()) sh->be->unpin_host (L.h_err_tail.data()); }
    delete sh->be;
    g_frame_shared.erase (std::find (g_frame_shared.begin(), g_frame_shared.end(), sh));
    delete sh;
  }
}

static_assert (sizeof (WelsHipFrameJob) >= WELSHIP_FRAMEJOB_MIN_SIZE && offsetof (WelsHipFrameJob, pbRecordsPacked) + sizeof (int32_t*) == WELSHIP_FRAMEJOB_MIN_SIZE,
               "WelsHipFrameJob: fields are only appended behind pbRecordsPacked (include/welship.h, cbSize)");
int WelsHipFrameEncode (WelsHipFrameCtx* c, const WelsHipFrameJob* j, const void** pp_records) {
  if (!c || !c->be || !j || !pp_records) return WELSHIP_ERR_INIT_PARA;
  if (j->cbSize < WELSHIP_FRAMEJOB_MIN_SIZE || j->cbSize > sizeof (WelsHipFrameJob)) {
    set_err ("WelsHipFrameJob::cbSize is outside [WELSHIP_FRAMEJOB_MIN_SIZE, this library's sizeof (WelsHipFrameJob)]: the caller was built against an include/welship.h this library cannot serve");
    return WELSHIP_ERR_INIT_PARA;
  }
  WelsHipFrameJob older;       // a caller compiled against an older (shorter) header: the fields it does not know are zero = "not used"
  if (j->cbSize != sizeof (WelsHipFrameJob)) { memset (&older, 0, sizeof (older)); memcpy (&older, j, j->cbSize); older.cbSize = (uint32_t)sizeof (older); j = &older; }
  if (j->bPackedRecords && !j->pbRecordsPacked) { set_err ("bPackedRecords without pbRecordsPacked: the caller could not tell which record format it got"); return WELSHIP_ERR_INIT_PARA; }
  if (j->pbRecordsPacked) *j->pbRecordsPacked = 0;
  const int np = (int)c->pics.size();
  const bool is_p = j->eSliceType == 0;
  if (j->iCurPic < 0 || j->iCurPic >= np || (is_p && (j->iRefPic < 0 || j->iRefPic >= np || j->iRefPic == j->iCurPic)) || (!is_p && j->eSliceType != 2)) {
    set_err ("invalid picture indices / slice type"); return WELSHIP_ERR_INIT_PARA;
  }
  if (j->iQp < 0 || j->iQp > 51 || !j->pSrc[0] || !j->pSrc[1] || !j->pSrc[2]) { set_err ("invalid job"); return WELSHIP_ERR_INIT_PARA; }
  // (the library never takes the caller down: the slice table and the plane strides are checked before anything reads through them)
  if (!j->pSliceFirstMb || j->iNumSlices < 1 || j->iNumSlices > WH_MAX_SLICES) { set_err ("invalid slice table"); return WELSHIP_ERR_INIT_PARA; }
  if (j->iSrcStride[0] < c->mb_w * 16 || j->iSrcStride[1] < c->mb_w * 8 || j->iSrcStride[2] < c->mb_w * 8) { set_err ("source strides below the macroblock-aligned picture width"); return WELSHIP_ERR_INIT_PARA; }
  if (is_p && j->pScreen && j->pScreen->pRefOriChroma[0] && j->pScreen->pRefOriChroma[1] && j->pScreen->iRefOriStride < c->mb_w * 8) { set_err ("screen-content job: stride of the reference's source chroma"); return WELSHIP_ERR_INIT_PARA; }
  if (is_p && j->iComplexityMode == 0 && !j->pVaaSad8x8) { set_err ("LOW complexity P pictures need the VAA 8x8 SADs of the pre-processing"); return WELSHIP_ERR_INIT_PARA; }
  const bool ranged = j->iMbEnd > 0;
  // size-limited slices: ranges coded ahead of the entropy writer, one slice per call; the picture-wide passes with a closing call
  const bool dyn = j->iDynSlice > 0;
  const bool dyn_close = dyn && j->iMbBegin == c->num_mb && j->iMbEnd == c->num_mb;
  if (dyn && (!ranged || j->bRetry || j->pGomRc || j->pMbQp || j->iDynSliceFirstMb < 0 || j->iDynSliceFirstMb > j->iMbBegin)) {
    set_err ("size-limited slices: MB ranges of a picture with a frame-constant QP"); return WELSHIP_ERR_INIT_PARA;
  }
  if (dyn && !dyn_close) {      // the slice table of such a picture are its PARTITIONS (one per slice thread; one = the picture): a range stays inside one
    bool inside = false;
    for (int i = 0; i < j->iNumSlices && j->pSliceFirstMb; ++i) inside = inside || (j->iDynSliceFirstMb >= j->pSliceFirstMb[i] && j->iMbEnd <= j->pSliceFirstMb[i + 1]);
    if (!inside) { set_err ("size-limited slices: the MB range crosses a partition of the picture"); return WELSHIP_ERR_INIT_PARA; }
  }
  if (ranged && !dyn_close && (j->iMbBegin < 0 || j->iMbBegin >= j->iMbEnd || j->iMbEnd > c->num_mb)) { set_err ("invalid MB range"); return WELSHIP_ERR_INIT_PARA; }
  const bool retry = j->bRetry != 0;
  if (retry && (ranged || !j->pReencode || j->iNumReencode < 1)) { set_err ("a retry is a whole-picture call with the list of re-encoded macroblocks"); return WELSHIP_ERR_INIT_PARA; }
  // MB ranges (GOM-synchronous rate control, size-limited slices) repeat only the range from the macroblock that overflowed on: the list of
  // macroblocks re-encoded so far travels with every later call of the picture (their QP / leaked state, and the QP_Y chain at the end)
  const bool ranged_reenc = ranged && !retry && j->pReencode && j->iNumReencode > 0;
  // (a retry reuses what the first call of the picture uploaded: source, pre-analysis arrays, screen-content inputs)
  const bool first_part = !retry && (!ranged || (j->iMbBegin == 0 && !j->bRangeAgain)), last_part = !ranged || (dyn ? dyn_close : j->iMbEnd == c->num_mb);
  if (c->check_tail (!is_p)) { set_err ("the deblocking pass / border expansion of this context's previous picture timed out or failed on the device: its reconstruction is unusable (code an IDR picture)"); return WELSHIP_ERR_UNKNOWN; }
  FrameShared* sh = c->sh;
  wh::Backend* be = c->be;
  // host-side staging into this context's own page-locked buffers: outside the shared lock
  // source picture: the MB-aligned area of pEncPic (CWelsPreProcess pads it), tight strides on the device -- unless the pre-analysis
  // of this very picture has put it there already (WelsHipFrameVaa)
  // (the address alone does not identify the content -- the caller may have refilled the buffer since the pre-analysis call, or dropped that
  //  picture: the slot's checksum must be the buffer's)
  bool src_resident = false;
  if (first_part && c->fresh_key != nullptr && c->fresh_key == (const void*)j->pSrc[0]) {
    const int k = c->src_find (c->fresh_key);
    src_resident = k >= 0 && c->src_pool[k].luma_sum == c->luma_checksum (j->pSrc[0], j->iSrcStride[0]);
  }
  if (first_part) c->fresh_key = nullptr;           // one pre-analysis call vouches for one encode call
  if (first_part && !src_resident) c->stage_planes (j->pSrc, j->iSrcStride);
  if (first_part) {
    if (is_p && j->pVaaSad8x8) memcpy (c->h_aux.data() + c->aux_vaa, j->pVaaSad8x8, sizeof (int32_t) * 4 * c->num_mb);
    if (is_p && j->pBgdFlags) memcpy (c->h_aux.data() + c->aux_bgd, j->pBgdFlags, (size_t)c->num_mb);
    if (is_p && j->pIlHint) memcpy (c->h_aux.data() + c->aux_il, j->pIlHint, sizeof (int16_t) * 4 * c->num_mb);
    if (j->pSadCost) memcpy (c->h_aux.data() + c->aux_sad, j->pSadCost, sizeof (int32_t) * c->num_mb);
  }
  // screen content: stage the pre-processing's arrays (page-locked, own to this context)
  const WelsHipScreenInfo* scr = is_p ? j->pScreen : nullptr;
  size_t scc_off_ori = 0, scc_off_times = 0, scc_off_start = 0, scc_off_loc = 0, scc_off_order = 0, scc_lists = 0, scc_entries = 0;
  const bool scc_serial = false;           // (the chained order is the only one: its plain-coding-order fallback lost its switch in round 5)
  const bool scc_scroll = scr && scr->bScrollDetectFlag && (scr->iScrollMvX | scr->iScrollMvY);
  const bool scc_chain = scc_scroll && !scc_serial;
  if (scr) {
    if (!scr->pBlockStaticIdc) { set_err ("screen-content job without the static-block map"); return WELSHIP_ERR_INIT_PARA; }
    const bool fme = scr->bFeatureSearch8x8 != 0;
    if (fme && (!scr->pTimesOfFeatureValue || !scr->pLocationOfFeature || !scr->pLocationPointer || scr->iListSize <= 0 || scr->iLocationEntries < 0)) {
      set_err ("screen-content job: feature search without the reference picture's feature lists"); return WELSHIP_ERR_INIT_PARA;
    }
    scc_lists = fme ? (size_t)scr->iListSize : 0; scc_entries = fme ? (size_t)scr->iLocationEntries : 0;
    scc_off_ori = (size_t)4 * c->num_mb;
    scc_off_times = scc_off_ori + 2 * c->csz;
    scc_off_start = scc_off_times + 4 * scc_lists;
    scc_off_loc = scc_off_start + 4 * scc_lists;
    scc_off_order = (scc_off_loc + 4 * scc_entries + 63) & ~ (size_t)63;
    const size_t need = scc_off_order + 8 * (size_t)c->num_mb + 64;
    if (first_part) {
      if (need > c->h_scc_cap) {
        std::unique_lock<std::mutex> lk (sh->mu);
        if (!c->h_scc.empty()) be->unpin_host (c->h_scc.data());
        c->h_scc.assign (need + need / 4, 0);
        c->h_scc_cap = c->h_scc.size();
        be->pin_host (c->h_scc.data(), c->h_scc.size());
      }
      uint8_t* st = c->h_scc.data();
      memcpy (st, scr->pBlockStaticIdc, (size_t)4 * c->num_mb);
      if (scr->pRefOriChroma[0] && scr->pRefOriChroma[1])
        for (int pl = 0; pl < 2; ++pl)
          for (int r = 0; r < c->mb_h * 8; ++r)
            memcpy (st + scc_off_ori + pl * c->csz + (size_t)r * c->seq.src_stride_c, scr->pRefOriChroma[pl] + (size_t)r * scr->iRefOriStride, (size_t)c->mb_w * 8);
      if (fme) {
        memcpy (st + scc_off_times, scr->pTimesOfFeatureValue, 4 * scc_lists);
        uint32_t* start = (uint32_t*) (st + scc_off_start);
        for (size_t f = 0; f < scc_lists; ++f) {
          const ptrdiff_t d = scr->pLocationOfFeature[f] ? scr->pLocationOfFeature[f] - scr->pLocationPointer : 0;
          if (d < 0 || (d & 1) || (size_t) (d >> 1) + scr->pTimesOfFeatureValue[f] > scc_entries) { set_err ("screen-content job: inconsistent feature lists"); return WELSHIP_ERR_INIT_PARA; }
          start[f] = (uint32_t) (d >> 1);
        }
        memcpy (st + scc_off_loc, scr->pLocationPointer, 4 * scc_entries);
      }
      if (scc_chain) {
        // WH_SEQ_CHAIN (wh_types.h): which macroblocks can search 8x8 blocks at all (MdInterAnalysisVaaInfo_c != MBVAASIGN_FLAT on the
        // pre-analysis' 8x8 SADs, exactly as the kernel evaluates it), the previous such macroblock of each one's slice, and a processing
        // order that respects those edges and the left / top-right ones: Kahn's algorithm, the ready macroblock that comes first in the
        // usual 2:1 order goes next
        if (!j->pVaaSad8x8 || j->iNumSlices < 1 || j->iNumSlices > WH_MAX_SLICES) { set_err ("screen-content job without the pre-analysis SADs"); return WELSHIP_ERR_INIT_PARA; }
        uint32_t* order = (uint32_t*) (st + scc_off_order);
        int32_t* prev = (int32_t*) (st + scc_off_order + 4 * (size_t)c->num_mb);
        std::vector<uint16_t> base (c->num_mb);
        std::vector<int32_t> rank (c->num_mb), indeg (c->num_mb);
        std::vector<int32_t> succ ((size_t)4 * c->num_mb, -1);
        for (int si = 0; si < j->iNumSlices; ++si) {
          const int first = j->pSliceFirstMb[si], last = j->pSliceFirstMb[si + 1];
          if (first < 0 || last > c->num_mb || first >= last) { set_err ("invalid slice layout"); return WELSHIP_ERR_INIT_PARA; }
          wh_build_mb_order (c->mb_w, first, last, base.data() + first);
          for (int t = first; t < last; ++t) rank[base[t]] = t;
          int pv = -1;
          for (int xy = first; xy < last; ++xy) {
            const int32_t* v = j->pVaaSad8x8 + 4 * (size_t)xy;
            const int avg = (v[0] + v[1] + v[2] + v[3]) >> 2;
            const int d0 = (v[0] >> 6) - (avg >> 6), d1 = (v[1] >> 6) - (avg >> 6), d2 = (v[2] >> 6) - (avg >> 6), d3 = (v[3] >> 6) - (avg >> 6);
            const bool nonflat = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 >= 20;
            prev[xy] = nonflat ? pv : -1;
            if (nonflat) pv = xy;
            int da, db;
            wh_mb_deps (c->mb_w, xy, first, &da, &db);
            const int deps[3] = {da, db, prev[xy]};
            indeg[xy] = 0;
            for (int k = 0; k < 3; ++k) if (deps[k] >= first && !(k == 2 && (deps[k] == da || deps[k] == db))) {
              ++indeg[xy];
              int32_t* sl = &succ[(size_t)4 * deps[k]];
              // a macroblock is the left neighbour of one, the top-right / top of at most two (right picture edge), the chain predecessor of one
              int q = 0;
              while (q < 4 && sl[q] >= 0) ++q;
              if (q == 4) { set_err ("internal: order graph"); return WELSHIP_ERR_UNKNOWN; }
              sl[q] = xy;
            }
          }
          std::priority_queue<std::pair<int32_t, int32_t>, std::vector<std::pair<int32_t, int32_t>>, std::greater<std::pair<int32_t, int32_t>>> ready;
          for (int xy = first; xy < last; ++xy) if (indeg[xy] == 0) ready.push ({rank[xy], xy});
          int t = first;
          while (!ready.empty()) {
            const int xy = ready.top().second;
            ready.pop();
            order[t++] = (uint32_t)xy;
            for (int k = 0; k < 4; ++k) { const int n = succ[(size_t)4 * xy + k]; if (n >= 0 && --indeg[n] == 0) ready.push ({rank[n], n}); }
          }
          if (t != last) { set_err ("internal: order graph has a cycle"); return WELSHIP_ERR_UNKNOWN; }
        }
      }
    }
  }
  // GOM-level rate control inside the kernel: the groups become bands of the processing order (2:1 order inside a group), every
  // macroblock of a group waits for the last macroblock of the group before it, which settles the group's QP (kernels/inter_mb.h)
  const WelsHipGomRc* gom = j->pGomRc;
  if (gom) {
    if (ranged || retry || scr || j->pMbQp || j->iNumSlices != 1) { set_err ("GOM-level rate control inside the kernel: a whole single-slice camera-video picture"); return WELSHIP_ERR_INIT_PARA; }
    if (gom->iNumberMbGom < 1 || gom->iNumberMbGom % c->mb_w != 0 || gom->iGomSize < 1 || gom->iGomSize > WH_GOM_MAX || !gom->pGomSad || gom->iEndMbSlice != c->num_mb - 1 ||
        gom->iEndMbSlice / gom->iNumberMbGom >= gom->iGomSize) { set_err ("GOM-level rate control inside the kernel needs groups of whole macroblock rows"); return WELSHIP_ERR_UNSUPPORTED; }
    if (c->h_gom.empty()) {
      std::unique_lock<std::mutex> lk (sh->mu);
      c->h_gom.assign (sizeof (WhGomRc) + 8 * (size_t)c->num_mb + 64, 0);
      be->pin_host (c->h_gom.data(), c->h_gom.size());
    }
    WhGomRc& R = * (WhGomRc*)c->h_gom.data();
    memset (&R, 0, sizeof (R));
    R.n_gom_mb = gom->iNumberMbGom; R.end_mb = gom->iEndMbSlice; R.target_bits = gom->iTargetBitsSlice;
    R.min_qp = gom->iMinFrameQp; R.max_qp = gom->iMaxFrameQp; R.slice_qp = j->iQp; R.p_slice = is_p ? 1 : 0;
    memcpy (R.gom_sad, gom->pGomSad, sizeof (int32_t) * gom->iGomSize);
    wh_gom_begin (R, j->iQp);
    uint32_t* order = (uint32_t*) (c->h_gom.data() + sizeof (WhGomRc));
    int32_t* dep = (int32_t*) (order + c->num_mb);
    std::vector<uint16_t> o16 (c->num_mb);
    wh_build_mb_order (c->mb_w, 0, c->num_mb, o16.data(), gom->iNumberMbGom / c->mb_w);
    for (int i = 0; i < c->num_mb; ++i) { order[i] = o16[i]; dep[i] = i / gom->iNumberMbGom ? (i / gom->iNumberMbGom) * gom->iNumberMbGom - 1 : -1; }
  }
  std::unique_lock<std::mutex> lock (sh->mu);
  // (the per-macroblock control words are built under the lock: the slice tasks of a picture with size-limited slices call concurrently)
  bool qp_map = false;
  if (j->pMbQp) {
    for (int i = 0; i < c->num_mb; ++i) { memset (&c->h_mb_ctl[i], 0, sizeof (WhMbCtl)); c->h_mb_ctl[i].qp_delta = (int8_t) ((int)j->pMbQp[i] - j->iQp); }
    qp_map = true;
  }
  if (retry || ranged_reenc) {
    if (!qp_map) memset (c->h_mb_ctl.data(), 0, sizeof (WhMbCtl) * c->num_mb);
    for (int i = 0; i < j->iNumReencode; ++i) {
      const WelsHipMbReencode& r = j->pReencode[i];
      if (r.iMbXY < 0 || r.iMbXY >= c->num_mb || r.uiLumaQp > 51) { set_err ("invalid re-encode entry"); return WELSHIP_ERR_INIT_PARA; }
      WhMbCtl& ctl = c->h_mb_ctl[r.iMbXY];
      ctl.qp_delta = (int8_t) ((int)r.uiLumaQp - j->iQp);
      ctl.stale_cbp = r.uiStaleCbp & 0x3f;
      ctl.cell12_valid = r.bCell12Valid ? 1 : 0; ctl.cell12_mv[0] = r.iCell12Mv[0]; ctl.cell12_mv[1] = r.iCell12Mv[1];
    }
    qp_map = true;
  }

  const auto t_sub0 = std::chrono::steady_clock::now();
  int rc = c->set_layout (j->iNumSlices, j->pSliceFirstMb, j->iDeblockIdc);
  if (rc) return rc;
  WhSeqParams& s = c->seq;
  s.deblock_idc = j->bDeblock ? j->iDeblockIdc : 1;       // an unfiltered picture (highest temporal layer) keeps the tables of the filtered ones
  s.complexity = j->iComplexityMode;
  s.chroma_qp_offset = j->iChromaQpIndexOffset;
  s.alpha_offset = j->iAlphaOffset; s.beta_offset = j->iBetaOffset;
  s.mv_range = j->iMvRange;
  // screen content: its own kernel variant; a picture with a scroll vector codes the macroblocks of a slice one after the other
  // (the directional-vector test of the 8x8 searches reads what the previous macroblock in CODING order left, WhSccJob::chain)
  s.flags = scr ? (WH_SEQ_SCC | (scc_chain ? WH_SEQ_CHAIN : scc_scroll ? WH_SEQ_SERIAL : 0)) : gom ? WH_SEQ_CHAIN : 0;
  if (ranged) s.flags |= WH_SEQ_RANGED;
  if (gom) qp_map = true;          // the QP changes from group to group: QP_Y of the macroblocks without mb_qp_delta (run_qp_chain)
  // the pictures this one can share a launch with, and the queue they use (MB ranges: queue 0, on their own)
  FrameKey* K = ranged ? nullptr : frame_find_key (sh, s, is_p, qp_map, j->bExpand != 0);
  const int queue = K ? K->queue : 0;
  be->select_queue (queue);
  c->join_tail (queue);                       // the previous picture's tail first: MB ranges run on this queue; a whole picture's uploads wait too (its kernels join
                                              // again in frame_run_batch on their lane's queue) -- no upload may overtake a pass that could still read what it replaces
  if (first_part) {
    const int slot = c->src_take ((const void*)j->pSrc[0]);
    c->d_src = c->src_pool[slot].d;
    if (!src_resident) {
      c->src_pool[slot].luma_sum = c->luma_checksum (c->h_src.data(), s.src_stride_y);       // (of the staged copy: what the slot will hold)
      be->upload (c->d_src_planar, c->h_src.data(), c->src_bytes);
      be->run_src_tile (s, c->d_src_planar, c->d_src);
    }
    if (is_p && j->pVaaSad8x8) be->upload (c->d_vaa, c->h_aux.data() + c->aux_vaa, sizeof (int32_t) * 4 * c->num_mb);
    if (is_p && j->pBgdFlags) be->upload (c->d_bgd, c->h_aux.data() + c->aux_bgd, (size_t)c->num_mb);
    if (is_p && j->pIlHint) be->upload (c->d_il, c->h_aux.data() + c->aux_il, sizeof (int16_t) * 4 * c->num_mb);
    if (c->sad_swap_pending) { std::swap (c->d_sad_cost0, c->d_sad_cost0_new); c->sad_swap_pending = false; }     // the previous picture's array is final now
    if (j->pSadCost) be->upload (c->d_sad_cost0, c->h_aux.data() + c->aux_sad, sizeof (int32_t) * c->num_mb);
    if (++c->db_gen == 0) c->db_gen = 1;
    c->h_pic_of = -1;
  }
  if (retry) { if (++c->db_gen == 0) c->db_gen = 1; c->h_pic_of = -1; }
  if (qp_map && !gom) be->upload (c->d_mb_ctl, c->h_mb_ctl.data(), sizeof (WhMbCtl) * c->num_mb);
  if (gom) {
    if (!c->d_gom_rc) c->d_gom_rc = (WhGomRc*)be->alloc (sizeof (WhGomRc) + 8 * (size_t)c->num_mb + 64);
    if (!c->d_gom_rc) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    be->upload (c->d_gom_rc, c->h_gom.data(), sizeof (WhGomRc) + 8 * (size_t)c->num_mb);
  }
  c->scc_active = scr != nullptr;
  if (scr) {
    bool oom = false;
    auto A = [&] (size_t n) { void* p = be->alloc (n); if (!p) oom = true; return p; };
    if (!c->d_scc) {
      static_assert (sizeof (WhSccJob) <= 256 && sizeof (uint32_t) * WH_MAX_SLICES <= 256, "staging layout");
      c->h_scc_small.assign (512, 0);
      be->pin_host (c->h_scc_small.data(), c->h_scc_small.size());
      c->d_scc = (WhSccJob*)A (sizeof (WhSccJob));
      c->d_scc_idc = (uint8_t*)A ((size_t)4 * c->num_mb + 64);
      c->d_scc_ori = (uint8_t*)A (2 * c->csz + 64);
      c->d_scc_chain = (uint32_t*)A (sizeof (uint32_t) * 5 * WH_MAX_SLICES);
    }
    if (scc_lists > c->scc_list_cap) { if (c->d_scc_lists) be->free (c->d_scc_lists); c->d_scc_lists = (uint32_t*)A (8 * scc_lists + 64); c->scc_list_cap = scc_lists; }
    if (scc_chain && !c->d_scc_order) c->d_scc_order = (uint32_t*)A (8 * (size_t)c->num_mb + 64);
    if (dyn && scc_chain && !c->d_scc_chain_mb) c->d_scc_chain_mb = (uint32_t*)A (16 * (size_t)c->num_mb + 64);
    if (scc_entries > c->scc_loc_cap) { if (c->d_scc_loc) be->free (c->d_scc_loc); c->d_scc_loc = (uint16_t*)A (4 * scc_entries + 64); c->scc_loc_cap = scc_entries; }
    if (oom) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    if (first_part) {
      const uint8_t* st = c->h_scc.data();
      be->upload (c->d_scc_idc, st, (size_t)4 * c->num_mb);
      if (scr->pRefOriChroma[0] && scr->pRefOriChroma[1]) be->upload (c->d_scc_ori, st + scc_off_ori, 2 * c->csz);
      if (scc_lists) { be->upload (c->d_scc_lists, st + scc_off_times, 8 * scc_lists); be->upload (c->d_scc_loc, st + scc_off_loc, 4 * scc_entries); }
      if (scc_chain) be->upload (c->d_scc_order, st + scc_off_order, 8 * (size_t)c->num_mb);
      be->fill (c->d_scc_chain, 0, sizeof (uint32_t) * 5 * WH_MAX_SLICES);
      WhSccJob& z = c->scc_job();
      memset (&z, 0, sizeof (z));
      z.static_idc = c->d_scc_idc;
      const bool have_ori = scr->pRefOriChroma[0] && scr->pRefOriChroma[1];
      z.ref_ori_c[0] = have_ori ? c->d_scc_ori : nullptr; z.ref_ori_c[1] = have_ori ? c->d_scc_ori + c->csz : nullptr;
      z.scroll_flag = scr->bScrollDetectFlag ? 1 : 0; z.scroll_mvx = scr->iScrollMvX; z.scroll_mvy = scr->iScrollMvY;
      z.thr16 = scr->uiSadCostThreshold16x16; z.thr8 = scr->uiSadCostThreshold8x8;
      z.fme = scc_lists ? 1 : 0;
      z.scd_on = scr->bStaticSkipDecision ? 1 : 0;
      z.fme_times = c->d_scc_lists; z.fme_start = c->d_scc_lists ? c->d_scc_lists + scc_lists : nullptr; z.fme_loc = c->d_scc_loc;
      z.fme_list_size = (int32_t)scc_lists;
      z.chain = c->d_scc_chain; z.fme_cost_down = c->d_scc_chain + 4 * WH_MAX_SLICES;
      z.chain_mb = c->d_scc_chain_mb;
      be->upload (c->d_scc, &z, sizeof (z));
    }
    if (retry) be->fill (c->d_scc_chain, 0, sizeof (uint32_t) * 5 * WH_MAX_SLICES);       // the picture is coded again from its first macroblock
  }
  DevPicture& cur = c->pics[j->iCurPic];
  WhPicJob job;
  memset (&job, 0, sizeof (job));
  job.src[0] = job.src[1] = job.src[2] = c->d_src;
  for (int i = 0; i < 3; ++i) { job.rec[i] = cur.plane[i]; job.ref[i] = is_p ? c->pics[j->iRefPic].plane[i] : nullptr; }
  for (int i = 0; i < 2; ++i) { job.rec_tiles[i] = cur.tiles[i]; job.ref_tiles[i] = is_p ? c->pics[j->iRefPic].tiles[i] : nullptr; }
  job.records = c->d_records;
  job.rec_blk = s.deblock_idc != 1 ? c->d_rec_blk : nullptr;
  job.mbs = cur.mbs;
  job.ref_mbs = is_p ? c->pics[j->iRefPic].mbs : nullptr;
  job.qp = j->iQp;
  job.slice_type = is_p ? WH_SLICE_P : WH_SLICE_I;
  job.mb_ctl = (qp_map && !gom) ? c->d_mb_ctl : nullptr;
  job.ref_is_p = is_p && c->pics[j->iRefPic].is_p ? 1 : 0;
  job.want_bits = j->bCountBits ? (1 | (is_p && j->iNumRefIdxL0Active > 1 ? 2 : 0)) : 0;
  job.prev_src_y = nullptr;
  job.db_flags = c->d_dbflags;
  job.db_gen = c->db_gen;
  job.sad_cost0 = c->d_sad_cost0;
  if (dyn) {
    if (j->pSadCost) { set_err ("size-limited slices: single-layer sessions only"); return WELSHIP_ERR_INIT_PARA; }
    if (!c->d_sad_cost0_new) { c->d_sad_cost0_new = (int32_t*)be->alloc (sizeof (int32_t) * c->num_mb); if (c->d_sad_cost0_new) be->fill (c->d_sad_cost0_new, 0, sizeof (int32_t) * c->num_mb); }
    if (!c->d_sad_cost0_new) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    job.sad_cost0_out = c->d_sad_cost0_new;
    job.dyn_redo = j->bDynRedoFirst ? 1 : 0;
  } else if (!ranged) {
    // whole-picture calls: every macroblock writes its entry of the second copy, and a repeat of the picture after a CAVLC overflow (bRetry)
    // starts from the previous picture's entries again, as the reference's TRY_REENCODING does macroblock by macroblock -- not from what
    // the abandoned pass left (a macroblock that was coded in that pass and is a P_Skip above LOW complexity now keeps the OLD entry)
    if (!c->d_sad_cost0_new) { c->d_sad_cost0_new = (int32_t*)be->alloc (sizeof (int32_t) * c->num_mb); if (c->d_sad_cost0_new) be->fill (c->d_sad_cost0_new, 0, sizeof (int32_t) * c->num_mb); }
    if (!c->d_sad_cost0_new) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    job.sad_cost0_out = c->d_sad_cost0_new;
    c->sad_swap_pending = true;
  }
  job.vaa_sad8x8 = is_p && j->pVaaSad8x8 ? c->d_vaa : nullptr;
  job.bgd_flags = is_p && j->pBgdFlags ? c->d_bgd : nullptr;
  job.mvc_shift = j->iMvcShift;
  job.il_hint = is_p && j->pIlHint ? c->d_il : nullptr;
  job.scc = scr ? c->d_scc : nullptr;
  job.scc_order = scc_chain ? c->d_scc_order : nullptr;
  job.scc_chain_prev = scc_chain ? (const int32_t*) (c->d_scc_order + c->num_mb) : nullptr;
  if (gom) {
    job.gom_rc = c->d_gom_rc;
    job.scc_order = (const uint32_t*) ((const uint8_t*)c->d_gom_rc + sizeof (WhGomRc));
    job.scc_chain_prev = (const int32_t*) (job.scc_order + c->num_mb);
    job.want_bits |= 1 | (j->iNumRefIdxL0Active > 1 ? 2 : 0);
  }
  job.mb_begin = ranged ? j->iMbBegin : 0; job.mb_end = ranged ? j->iMbEnd : 0;
  job.dyn_slice = dyn ? j->iDynSlice : 0; job.dyn_first = dyn ? j->iDynSliceFirstMb : 0;

  if (ranged) {
    // GOM-synchronous coding: this MB range now, the picture-wide passes with the last range
    if (!dyn_close) {
      be->upload (c->d_job, &job, sizeof (job));
      if (is_p) be->run_inter (s, c->d_job, 1); else be->run_intra (s, c->d_job, 1);
    }
    if (last_part) {
      job.mb_begin = 0; job.mb_end = 0; be->sync_queue (queue); be->upload (c->d_job, &job, sizeof (job));
      if (qp_map) be->run_qp_chain (s, c->d_job, 1);           // QP_Y for the filter and pRefMbQp of decided skips
      if (s.deblock_idc != 1) be->run_deblock (s, c->d_job, 1);
      if (j->bExpand) be->run_expand (s, c->d_job, 1);
      cur.is_p = is_p;
      if (dyn) std::swap (c->d_sad_cost0, c->d_sad_cost0_new);       // every macroblock of the picture has written its entry of the new copy
      if (j->pSadCost) be->download (j->pSadCost, c->d_sad_cost0, sizeof (int32_t) * c->num_mb);
      if (scr) be->download (c->scc_down(), c->d_scc_chain + 4 * WH_MAX_SLICES, sizeof (uint32_t) * WH_MAX_SLICES);
    }
    if (!dyn_close) be->download (c->h_records.data() + j->iMbBegin, c->d_records + j->iMbBegin, sizeof (WhMbRecord) * (size_t) (j->iMbEnd - j->iMbBegin));
    // wait for the range outside the device-wide lock: other sessions (and the other slice threads of this picture) stage and submit
    // meanwhile -- a picture of ranges is dozens of these round trips (everything above was queued under the lock, in order, on `queue`)
    const unsigned swept0 = be->errors_swept();
    lock.unlock();
    int bad = be->sync_queue (queue);
    if (be->errors_swept() != swept0) bad = 1;       // another thread's sync() found time-outs meanwhile: possibly this range's
    lock.lock();
    c->tail_queue = -1;                       // (this queue was waited for: nothing of the context is in flight)
    if (bad) { set_err ("device scheduler timed out or device error; the picture was not encoded"); return WELSHIP_ERR_UNKNOWN; }
    if (scr && last_part && scr->pSliceFMECostDown) memcpy (scr->pSliceFMECostDown, c->scc_down(), sizeof (uint32_t) * j->iNumSlices);
    *pp_records = c->h_records.data();
    return WELSHIP_OK;
  }

  FrameItem item;
  item.c = c; item.seq = s; item.job = job; item.is_p = is_p; item.qp_map = qp_map; item.expand = j->bExpand != 0;
  item.cur_pic = j->iCurPic; item.sad_dst = j->pSadCost;
  // packed records on request (pictures larger than the packer's workgroup handles keep the full records, as does WELSHIP_COMPACT=0)
  static const bool compact_off_env = getenv ("WELSHIP_COMPACT") && atoi (getenv ("WELSHIP_COMPACT")) == 0;
  if (j->bPackedRecords && c->num_mb <= WELSHIP_PACKED_MAX_MB && !compact_off_env) {
    if (!c->d_compact) {
      c->d_compact = (uint8_t*)be->alloc ((size_t)c->num_mb * WH_COMPACT_MAX_BYTES);
      c->d_compact_off = (uint32_t*)be->alloc (sizeof (uint32_t) * ((size_t)c->num_mb + 1));
      if (!c->d_compact || !c->d_compact_off) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
      c->h_compact.resize ((size_t)c->num_mb * WH_COMPACT_MAX_BYTES);
      c->h_coff.resize ((size_t)c->num_mb + 1);
      be->pin_host (c->h_compact.data(), c->h_compact.size());
      be->pin_host (c->h_coff.data(), sizeof (uint32_t) * c->h_coff.size());
      c->compact_est = c->h_compact.size() / 4;
    }
    item.packed = true;
  }
  c->last_key = K;
  c->last_submit = std::chrono::steady_clock::now();
  sh->stat_submit_ms += std::chrono::duration<double, std::milli> (c->last_submit - t_sub0).count();
  K->pending.push_back (&item);
  if (K->gathering) sh->cv.notify_all();               // a gathering leader may have been waiting for exactly this picture
  for (;;) {
    if (item.done) break;
    const int li = K->free_lane();
    if (li >= 0 && K->gathering == 0 && !K->pending.empty() && (K->next_leader == nullptr || K->next_leader == &item)) {
      // this thread launches for the key: whatever is pending now plus what arrives within the gathering window
      FrameLane* Lk = &K->lane[li];
      Lk->busy = true;
      ++K->gathering;
      K->next_leader = nullptr;
      const auto t_g0 = std::chrono::steady_clock::now();
      if (sh->gather_us > 0 && sh->ctxs.size() > 1) {
        // the contexts whose last picture (within 100 ms) had this key: their next one is probably on its way
        const auto now = std::chrono::steady_clock::now();
        size_t expected = 0;
        for (WelsHipFrameCtx* x : sh->ctxs) if (x->last_key == K && now - x->last_submit < std::chrono::milliseconds (100)) ++expected;
        // (Measured in round 6 and not kept: a key whose last three windows nobody joined launches at once and only tries every eighth time -- eight
        //  four-layer simulcast sessions, whose leaders wait 3 ms of a 10 ms device call for batches that stay at one picture: 126 -> 126 / 108
        //  frames/s; config 5 unchanged.  profiles/r06_config4_layer_split_ab.txt)
        if (K->pending.size() < expected)
          sh->cv.wait_for (lock, std::chrono::microseconds (sh->gather_us), [&] { return K->pending.size() >= expected; });
      }
      std::vector<FrameItem*> batch;
      batch.swap (K->pending);
      K->next_leader = nullptr;                             // (whoever it was is in this batch now)
      --K->gathering;
      const double gather_ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t_g0).count();
      const size_t bn = batch.size();
      frame_run_batch (sh, K, Lk, lock, batch);             // (releases the lock while the device works: the other lane may launch meanwhile)
      if (bn < sh->stat_gather_ms.size()) sh->stat_gather_ms[bn] += gather_ms;
      Lk->busy = false;
      if (!K->pending.empty() && K->gathering == 0 && K->next_leader == nullptr) K->next_leader = K->pending.front();      // pictures that arrived meanwhile: their first submitter goes next
      sh->cv.notify_all();
      continue;
    }
    sh->cv.wait (lock);
  }
  if (item.rc) { set_err ("device scheduler timed out or device error; the picture was not encoded"); return item.rc; }
  lock.unlock();
  if (j->pSadCost) memcpy (j->pSadCost, c->h_sad_out.data(), sizeof (int32_t) * c->num_mb);
  if (scr && scr->pSliceFMECostDown) memcpy (scr->pSliceFMECostDown, c->scc_down(), sizeof (uint32_t) * j->iNumSlices);
  if (item.packed) {
    c->packed_view.pData = c->h_compact.data(); c->packed_view.pOffset = c->h_coff.data();
    *pp_records = &c->packed_view;
    *j->pbRecordsPacked = 1;
  } else *pp_records = c->h_records.data();
  return WELSHIP_OK;
}

int WelsHipFrameVaa (WelsHipFrameCtx* c, const WelsHipVaaJob* j) {
  if (!c || !c->be || !j) return WELSHIP_ERR_INIT_PARA;
  for (int i = 0; i < 3; ++i) if (!j->pCur[i] || !j->pRef[i] || j->iCurStride[i] < (i ? c->mb_w * 8 : c->mb_w * 16) || j->iRefStride[i] < (i ? c->mb_w * 8 : c->mb_w * 16)) {
    set_err ("pre-analysis: planes / strides"); return WELSHIP_ERR_INIT_PARA;
  }
  // A width that is no multiple of 16: the C functions step from one macroblock row to the next by 16 * stride - width, i.e. every row
  // starts (width & 15) samples further left than the one above and runs into the previous line's stride padding and samples
  // (vaacalcfuncs.cpp:46,145-146 ...) -- results that depend on the bytes between the lines.  Such a picture takes the path below that
  // hands the device the two luma planes as the caller has them, padding included (kernels/vaa_pic.h wh_vaa_mb_skewed).
  const bool skewed = (j->iPicWidth & 15) != 0;
  if (skewed && j->iCurStride[0] != j->iRefStride[0]) { set_err ("pre-analysis: a width that is no multiple of 16 needs both pictures at one stride (the C functions take one)"); return WELSHIP_ERR_UNSUPPORTED; }
  const int vw = j->iPicWidth >> 4, vh = j->iPicHeight >> 4;       // the macroblocks the C functions cover
  if (vw < 1 || vh < 1 || vw > c->mb_w || vh > c->mb_h || !j->pSad8x8 || !j->pFrameSad) { set_err ("pre-analysis: picture size / result arrays"); return WELSHIP_ERR_INIT_PARA; }
  // which arrays the selected variant writes (vaacalculation.cpp:118-157)
  const bool bgd = j->bCalcBgd != 0, ssd = j->bCalcSsd != 0, var = !bgd && !ssd && j->bCalcVar != 0;
  const bool want_sd = bgd, want_sum = ssd || var, want_ssd = ssd;
  if ((want_sd && (!j->pSumOfDiff8x8 || !j->pMad8x8)) || (want_sum && (!j->pSum16x16 || !j->pSumOfSquare16x16)) || (want_ssd && !j->pSsd16x16)) {
    set_err ("pre-analysis: a result array of the selected variant is missing"); return WELSHIP_ERR_INIT_PARA;
  }
  FrameShared* sh = c->sh;
  wh::Backend* be = c->be;
  const size_t n = (size_t)c->num_mb;
  const size_t o_sad = 0, o_sd = 16 * n, o_sum = 32 * n, o_sq = 36 * n, o_ssd = 40 * n, o_mad = 44 * n, out_bytes = 48 * n;
  const int vq = (int) ((uintptr_t)c / 64 % 8);
  const int queue = 24 + 4 * (vq / 4) + (3 - vq % 4);          // the pre-analysis queues (keys: 0..7 uploads, 8..23 launch sets)
  if (skewed) {
    const int stride = j->iCurStride[0];
    const size_t plane = (size_t)vh * 16 * (size_t)stride;            // the lines the walk touches: [0, 16 * vh) of each plane, whole lines
    if (c->skew_bytes < 2 * plane) {
      std::unique_lock<std::mutex> lock (sh->mu);
      if (c->d_skew) be->free (c->d_skew);
      if (!c->h_skew.empty()) be->unpin_host (c->h_skew.data());
      c->d_skew = (uint8_t*)be->alloc (2 * plane);
      c->h_skew.assign (2 * plane, 0);
      be->pin_host (c->h_skew.data(), 2 * plane);
      c->skew_bytes = c->d_skew ? 2 * plane : 0;
      if (!c->d_skew) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    }
    memcpy (c->h_skew.data(), j->pCur[0], plane);
    memcpy (c->h_skew.data() + plane, j->pRef[0], plane);
    {
      std::unique_lock<std::mutex> lock (sh->mu);
      if (!c->d_vaa_out) {
        c->d_vaa_out = (uint8_t*)be->alloc (out_bytes);
        c->h_vaa_out.assign (out_bytes, 0);
        be->pin_host (c->h_vaa_out.data(), out_bytes);
        if (!c->d_vaa_out) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
      }
      be->select_queue (queue);
      be->upload (c->d_skew, c->h_skew.data(), 2 * plane);
      uint8_t* o = c->d_vaa_out;
      be->run_vaa_skewed (c->seq, c->d_skew, c->d_skew + plane, stride, j->iPicWidth, j->iPicHeight, (int32_t*) (o + o_sad), want_sd ? (int32_t*) (o + o_sd) : nullptr,
                          want_sd ? o + o_mad : nullptr, want_sum ? (int32_t*) (o + o_sum) : nullptr, want_sum ? (int32_t*) (o + o_sq) : nullptr, want_ssd ? (int32_t*) (o + o_ssd) : nullptr);
      be->download (c->h_vaa_out.data(), o, out_bytes);
    }
    if (be->sync_queue (queue)) { set_err ("device error in the pre-analysis"); return WELSHIP_ERR_UNKNOWN; }
    c->fresh_key = nullptr;              // (the picture itself was not put on the device in the encoder's layout: the encode call uploads it)
    c->vaa_cur_key = nullptr; c->vaa_has_bgd = false;
  } else {
  // Host-side work -- the reference's checksum, staging into this context's own page-locked buffer -- happens OUTSIDE the device-wide lock
  // (other sessions keep submitting meanwhile); the lock covers queue selection and the enqueues only.  The source pool is the context's own.
  // The earlier picture is resident when it was the source of an earlier call AND the caller's buffer still holds what was uploaded then
  // (its slot is refreshed so that it survives the upload below).
  const uint64_t ref_sum = c->luma_checksum (j->pRef[0], j->iRefStride[0]);
  int rslot = c->src_find ((const void*)j->pRef[0]);
  if (rslot >= 0 && c->src_pool[rslot].luma_sum == ref_sum) c->src_pool[rslot].stamp = ++c->src_clock;
  else {
    rslot = c->src_take ((const void*)j->pRef[0]);
    c->src_pool[rslot].luma_sum = ref_sum;
    c->stage_planes (j->pRef, j->iRefStride);
    {
      std::unique_lock<std::mutex> lock (sh->mu);
      be->select_queue (queue);
      be->upload (c->d_src_planar, c->h_src.data(), c->src_bytes);
      be->run_src_tile (c->seq, c->d_src_planar, c->src_pool[rslot].d);
    }
    if (be->sync_queue (queue)) { set_err ("device error in the pre-analysis"); return WELSHIP_ERR_UNKNOWN; }      // (the staging buffer is used again below)
  }
  const int cslot = c->src_take ((const void*)j->pCur[0]);
  if (cslot == rslot) { set_err ("pre-analysis: a picture against itself"); return WELSHIP_ERR_INIT_PARA; }
  c->stage_planes (j->pCur, j->iCurStride);
  c->src_pool[cslot].luma_sum = c->luma_checksum (c->h_src.data(), c->seq.src_stride_y);
  std::unique_lock<std::mutex> lock (sh->mu);
  if (!c->d_vaa_out) {
    c->d_vaa_out = (uint8_t*)be->alloc (out_bytes);
    c->h_vaa_out.assign (out_bytes, 0);
    be->pin_host (c->h_vaa_out.data(), out_bytes);
    if (!c->d_vaa_out) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
  }
  be->select_queue (queue);
  be->upload (c->d_src_planar, c->h_src.data(), c->src_bytes);
  be->run_src_tile (c->seq, c->d_src_planar, c->src_pool[cslot].d);
  uint8_t* o = c->d_vaa_out;
  be->run_vaa (c->seq, c->src_pool[cslot].d, c->src_pool[rslot].d, (int32_t*) (o + o_sad), want_sd ? (int32_t*) (o + o_sd) : nullptr, want_sd ? o + o_mad : nullptr,
               want_sum ? (int32_t*) (o + o_sum) : nullptr, want_sum ? (int32_t*) (o + o_sq) : nullptr, want_ssd ? (int32_t*) (o + o_ssd) : nullptr);
  be->download (c->h_vaa_out.data(), o, out_bytes);
  lock.unlock();
  if (be->sync_queue (queue)) { set_err ("device error in the pre-analysis"); return WELSHIP_ERR_UNKNOWN; }
  c->fresh_key = (const void*)j->pCur[0];
  c->vaa_cur_key = (const void*)j->pCur[0]; c->vaa_ref_key = (const void*)j->pRef[0]; c->vaa_cslot = cslot; c->vaa_rslot = rslot; c->vaa_queue = queue; c->vaa_has_bgd = want_sd;
  c->vaa_cur_sum = c->src_pool[cslot].luma_sum; c->vaa_ref_sum = c->src_pool[rslot].luma_sum;
  }
  // results: the macroblocks the C functions cover, row by row; the frame SAD is their sum
  const uint8_t* h = c->h_vaa_out.data();
  long long frame_sad = 0;
  for (int y = 0; y < vh; ++y) {
    const size_t a = (size_t)y * c->mb_w, d = (size_t)y * vw;        // device rows are mb_w wide, the caller's arrays (w >> 4)
    const int32_t* sad = (const int32_t*) (h + o_sad) + 4 * a;
    memcpy (j->pSad8x8 + 4 * d, sad, sizeof (int32_t) * 4 * vw);
    for (int i = 0; i < 4 * vw; ++i) frame_sad += sad[i];
    if (want_sd) { memcpy (j->pSumOfDiff8x8 + 4 * d, (const int32_t*) (h + o_sd) + 4 * a, sizeof (int32_t) * 4 * vw); memcpy (j->pMad8x8 + 4 * d, h + o_mad + 4 * a, (size_t)4 * vw); }
    if (want_sum) { memcpy (j->pSum16x16 + d, (const int32_t*) (h + o_sum) + a, sizeof (int32_t) * vw); memcpy (j->pSumOfSquare16x16 + d, (const int32_t*) (h + o_sq) + a, sizeof (int32_t) * vw); }
    if (want_ssd) memcpy (j->pSsd16x16 + d, (const int32_t*) (h + o_ssd) + a, sizeof (int32_t) * vw);
  }
  *j->pFrameSad = (int32_t)frame_sad;
  return WELSHIP_OK;
}

int WelsHipFrameBgd (WelsHipFrameCtx* c, const WelsHipBgdJob* j) {
  if (!c || !c->be || !j || !j->pBackgroundMbFlag) return WELSHIP_ERR_INIT_PARA;
  const int uw = j->iPicWidth >> 4, uh = j->iPicHeight >> 4;
  if ((j->iPicWidth & 15) || uw < 1 || uh < 1 || uw > c->mb_w || uh > c->mb_h || (size_t)uw * uh > 65536) { set_err ("background detection: picture size (width a multiple of 16, at most 65536 units)"); return WELSHIP_ERR_UNSUPPORTED; }
  if (!c->vaa_has_bgd || c->vaa_cur_key == nullptr || c->vaa_cur_key != (const void*)j->pCur[0] || c->vaa_ref_key != (const void*)j->pRef[0] || !c->d_vaa_out ||
      c->src_find (c->vaa_cur_key) != c->vaa_cslot || c->src_find (c->vaa_ref_key) != c->vaa_rslot ||
      c->src_pool[c->vaa_cslot].luma_sum != c->vaa_cur_sum || c->src_pool[c->vaa_rslot].luma_sum != c->vaa_ref_sum) {       // (the source ring reuses pointers: the slots' contents must still be this pair's)
    set_err ("background detection: the statistics of this picture pair are not on the device"); return WELSHIP_ERR_UNSUPPORTED;
  }
  wh::Backend* be = c->be;
  const size_t n = (size_t)c->num_mb;
  const uint8_t* o = c->d_vaa_out;                  // (WelsHipFrameVaa's layout)
  {
    std::unique_lock<std::mutex> lock (c->sh->mu);
    if (!c->d_bgd_calc) {
      c->d_bgd_calc = (int8_t*)be->alloc (n);
      c->h_bgd_calc.assign (n, 0);
      be->pin_host (c->h_bgd_calc.data(), n);
      if (!c->d_bgd_calc) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
    }
    be->select_queue (c->vaa_queue);
    be->run_bgd (c->seq, c->src_pool[c->vaa_cslot].d, c->src_pool[c->vaa_rslot].d, (const int32_t*) (o + 0), (const int32_t*) (o + 16 * n), o + 44 * n, uw, uh, c->d_bgd_calc);
    be->download (c->h_bgd_calc.data(), c->d_bgd_calc, n);
  }
  if (be->sync_queue (c->vaa_queue)) { set_err ("device error in the background detection"); return WELSHIP_ERR_UNKNOWN; }
  const int row = (j->iPicWidth + 15) >> 4;
  for (int y = 0; y < uh; ++y) memcpy (j->pBackgroundMbFlag + (size_t)y * row, c->h_bgd_calc.data() + (size_t)y * c->mb_w, (size_t)uw);
  return WELSHIP_OK;
}

int WelsHipFrameGetMbStates (WelsHipFrameCtx* c, int pic, void* dst, size_t bytes) {
  if (!c || !c->be || pic < 0 || pic >= (int)c->pics.size() || !dst || bytes < sizeof (WhMbState) * c->num_mb) return WELSHIP_ERR_INIT_PARA;
  std::unique_lock<std::mutex> lock (c->sh->mu);
  c->be->select_queue (c->queue());
  c->join_tail (c->queue());
  c->be->download (dst, c->pics[pic].mbs, sizeof (WhMbState) * c->num_mb);
  return c->be->sync_queue (c->queue()) ? WELSHIP_ERR_UNKNOWN : WELSHIP_OK;
}

int WelsHipFrameGetPicture (WelsHipFrameCtx* c, int pic, uint8_t* const dst[3], const int32_t stride[3]) {
  if (!c || !c->be || pic < 0 || pic >= (int)c->pics.size() || !dst || !stride) return WELSHIP_ERR_INIT_PARA;
  const WhSeqParams& s = c->seq;
  std::vector<uint8_t>& tmp = c->h_pic;
  const DevPicture& p = c->pics[pic];
  if (c->h_pic_of != pic) {              // not the picture that came back with the last batch (GOM-coded pictures, older pictures)
    std::unique_lock<std::mutex> lock (c->sh->mu);
    c->be->select_queue (c->queue());
    c->join_tail (c->queue());
    c->be->download (tmp.data(), p.base, c->rec_alloc_bytes + 128);
    if (c->be->sync_queue (c->queue())) return WELSHIP_ERR_UNKNOWN;
    c->h_pic_of = pic;
  }
  const uint8_t* y = tmp.data() + (p.plane[0] - p.base);
  const uint8_t* u = tmp.data() + (p.plane[1] - p.base);
  const uint8_t* v = tmp.data() + (p.plane[2] - p.base);
  for (int r = 0; r < c->mb_h * 16; ++r) memcpy (dst[0] + (size_t)r * stride[0], y + (size_t)r * s.rec_stride_y, (size_t)c->mb_w * 16);
  for (int r = 0; r < c->mb_h * 8; ++r) {
    memcpy (dst[1] + (size_t)r * stride[1], u + (size_t)r * s.rec_stride_c, (size_t)c->mb_w * 8);
    memcpy (dst[2] + (size_t)r * stride[2], v + (size_t)r * s.rec_stride_c, (size_t)c->mb_w * 8);
  }
  return WELSHIP_OK;
}

}  // extern "C"
