// emu_backend.cpp -- TEST INFRASTRUCTURE ONLY (never part of libwelship.so).
//
// Compiles the very same macroblock kernel sources (openh264_amd/csrc/kernels/*.h) with -DWH_EMU,
// where a "wavefront" is a 64-iteration loop (kernels/wave.h), and runs them in the same 2:1
// diagonal order the HIP launcher uses.  It exists so that the CPU-only test tier can exercise the
// host side (headers, CAVLC, NAL packing, session logic) and the kernel *logic* against the
// reference bitstream without a GPU.  The product library has no such path and refuses to work
// without an MI355X.
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "../../openh264_amd/csrc/host/backend.h"
#include "../../openh264_amd/csrc/kernels/frame_kernels.h"
#include "../../openh264_amd/csrc/kernels/deblock_mb.h"
#include "../../openh264_amd/csrc/kernels/inter_mb.h"
#include "../../openh264_amd/csrc/kernels/expand_pic.h"
#include "../../openh264_amd/csrc/kernels/tile_pic.h"
#include "../../openh264_amd/csrc/kernels/vaa_pic.h"
#include "../../openh264_amd/csrc/kernels/bgd_pic.h"
#include "../../openh264_amd/csrc/kernels/scene_pic.h"
#include "../../openh264_amd/csrc/common/compact.h"
#include "../../openh264_amd/csrc/kernels/downsample_px.h"
#include "../../include/welship.h"

namespace wh {

class EmuBackend : public Backend {
 public:
  const char* name() const override { return "emu:wave64-on-cpu (test build)"; }
  void* alloc (size_t bytes) override { return ::malloc (bytes ? bytes : 1); }
  void free (void* p) override { ::free (p); }
  void upload (void* dst, const void* src, size_t bytes) override { memcpy (dst, src, bytes); }
  void download (void* dst, const void* src, size_t bytes) override { memcpy (dst, src, bytes); }
  void fill (void* dst, int value, size_t bytes) override { memset (dst, value, bytes); }
  // LDS is not initialised on the GPU: poison the emulated tile so that a kernel relying on stale contents fails here
  static void poison (void* p, size_t n) { memset (p, 0xA5, n); }
  // same per-slice / per-picture dependency order the device kernels walk (common/mb_order.h), one MB at a time
  template <class F> static void for_order (const WhSeqParams& P, int n, bool whole_picture, F f) {
    const int num_mb = P.mb_w * P.mb_h;
    for (int j = 0; j < n; ++j) {
      if (whole_picture) { for (int t = 0; t < num_mb; ++t) { const int xy = P.mb_order[num_mb + t]; f (j, xy % P.mb_w, xy / P.mb_w); } continue; }
      for (int s = 0; s < P.num_slices; ++s)
        for (int t = P.slice_first_mb[s]; t < P.slice_first_mb[s + 1]; ++t) { const int xy = P.mb_order[t]; f (j, xy % P.mb_w, xy / P.mb_w); }
    }
  }
  void run_intra (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    bool any_gom = false;
    for (int j = 0; j < n; ++j) any_gom = any_gom || jobs[j].gom_rc != nullptr;
    // the device picks its kernel variant by the launch's flag (hip_backend.hip run_intra: k_intra_slice_gom for WH_SEQ_CHAIN): the flag must say what the jobs hold
    if (any_gom && !(P.flags & WH_SEQ_CHAIN)) { fprintf (stderr, "emu: an I picture with GOM-level rate control in a launch without WH_SEQ_CHAIN\n"); abort(); }
    if (!any_gom) {
      for_order (P, n, false, [&] (int j, int x, int y) {
        const int xy = y * P.mb_w + x;
        if (jobs[j].mb_end > 0 && (xy < jobs[j].mb_begin || xy >= jobs[j].mb_end)) return;      // GOM-synchronous coding: only this range
        WhMbLds S; poison (&S, sizeof (S)); wh_intra_mb_body<false> (S, P, jobs[j], x, y); });
      return;
    }
    // GOM-level rate control inside the kernel: the picture's own order (groups as bands), checked against everything the device scheduler waits for
    const int num_mb = P.mb_w * P.mb_h;
    for (int j = 0; j < n; ++j) {
      const WhPicJob& J = jobs[j];
      if (!J.gom_rc || P.num_slices != 1 || !J.scc_order || !J.scc_chain_prev) { fprintf (stderr, "emu: GOM-level rate control inside the intra kernel needs a single-slice picture with its own order\n"); abort(); }
      std::vector<char> done_mb ((size_t)num_mb, 0);
      for (int t = 0; t < num_mb; ++t) {
        const int xy = (int)J.scc_order[t];
        int da, db;
        wh_mb_deps (P.mb_w, xy, 0, &da, &db);
        const int dc = J.scc_chain_prev[xy];
        if ((da >= 0 && !done_mb[da]) || (db >= 0 && !done_mb[db]) || (dc >= 0 && !done_mb[dc]) || dc >= xy) { fprintf (stderr, "emu: the I picture's order is not topological at MB %d (deps %d %d %d)\n", xy, da, db, dc); abort(); }
        done_mb[xy] = 1;
        WhMbLds S; poison (&S, sizeof (S));
        wh_intra_mb_body<true> (S, P, J, xy % P.mb_w, xy / P.mb_w);
        wh_gom_close_if_last (P, J, xy);
      }
    }
  }
  void run_inter (const WhSeqParams& Pin, const WhPicJob* jobs, int n) override {
    // WH_SEQ_PLAIN (a session group's step): the body variant that never looks at the optional per-picture inputs (as hip_backend.hip takes it;
    // -DWH_PLAIN_KERNEL=0: the general body) -- and a check that the promise holds
#ifndef WH_PLAIN_KERNEL
#define WH_PLAIN_KERNEL 2
#endif
#ifndef WH_FRAME_KERNEL
#define WH_FRAME_KERNEL 1
#endif
    const bool plain = WH_PLAIN_KERNEL && (Pin.flags & WH_SEQ_PLAIN) != 0, no_ctrl = WH_FRAME_KERNEL && (Pin.flags & WH_SEQ_NO_CTRL) != 0;
    WhSeqParams Pm = Pin;
    Pm.flags &= ~ (WH_SEQ_PLAIN | WH_SEQ_NO_CTRL);
    const WhSeqParams& P = Pm;
    if (Pin.flags & WH_SEQ_PLAIN)
      for (int j = 0; j < n; ++j) {
        const WhPicJob& q = jobs[j];
        if (P.flags || q.vaa_sad8x8 || q.sad_cost0 || q.bgd_flags || q.il_hint || q.mb_ctl || q.gom_rc || q.dyn_slice || q.want_bits || q.mb_end || q.mvc_shift || q.scc) {
          fprintf (stderr, "emu: WH_SEQ_PLAIN on a picture with optional inputs\n"); abort();
        }
      }
    if (Pin.flags & WH_SEQ_NO_CTRL)
      for (int j = 0; j < n; ++j) {
        const WhPicJob& q = jobs[j];
        if (P.flags || q.il_hint || q.mb_ctl || q.gom_rc || q.dyn_slice || q.want_bits || q.mb_end || q.scc) { fprintf (stderr, "emu: WH_SEQ_NO_CTRL on a picture with control inputs\n"); abort(); }
      }
    // one emulated wavefront walks each slice in order like a wave of the device scheduler: a macroblock's cold inputs and
    // its speculative search windows (around the slice's last final vector) are fetched before its body runs
    for (int j = 0; j < n; ++j)
      for (int s = 0; s < P.num_slices; ++s) {
        WhInterLds S;
        WhInterStage G;
        WhWinLds WB;
        poison (&S, sizeof (S)); poison (&G, sizeof (G)); poison (&WB, sizeof (WB));
        const int first = P.slice_first_mb[s], last = P.slice_first_mb[s + 1];
        int last_mv = 0;
        std::vector<char> done_mb ((size_t) (last - first), 0);
        for (int t = first; t < last; ++t) {
          const int xy = (P.flags & WH_SEQ_CHAIN) ? (int)jobs[j].scc_order[t] : (P.flags & WH_SEQ_SERIAL) ? t : (int)P.mb_order[t];
          if (P.flags & WH_SEQ_CHAIN) {      // the picture's own order must respect everything the device scheduler waits for
            int da, db;
            wh_mb_deps (P.mb_w, xy, first, &da, &db);
            const int dc = jobs[j].scc_chain_prev[xy];
            if ((da >= first && !done_mb[da - first]) || (db >= first && !done_mb[db - first]) || (dc >= first && !done_mb[dc - first]) || dc >= xy) {
              fprintf (stderr, "emu: WH_SEQ_CHAIN order is not topological at MB %d (deps %d %d %d)\n", xy, da, db, dc);
              abort();
            }
            done_mb[xy - first] = 1;
          }
          if (jobs[j].mb_end > 0 && (xy < jobs[j].mb_begin || xy >= jobs[j].mb_end)) continue;      // GOM-synchronous coding: only this range
          const int mbx = xy % P.mb_w, mby = xy / P.mb_w;
          // the device scheduler copies a macroblock's cold inputs straight into the wave's tile while it waits for the neighbours: the tile holds
          // nothing else of use by then (poisoned below, after every macroblock)
          for (int lane = 0; lane < 64; ++lane) { if (plain) wh_inter_cold_fetch<1> (S, G, lane, P, jobs[j], mbx, mby); else if (no_ctrl) wh_inter_cold_fetch<3> (S, G, lane, P, jobs[j], mbx, mby); else wh_inter_cold_fetch (S, G, lane, P, jobs[j], mbx, mby); }
          WhInterCtx X;
          X.slice_idc = jobs[j].dyn_slice ? jobs[j].dyn_slice - 1 : s; X.slice_first = jobs[j].dyn_slice ? jobs[j].dyn_first : first;
          X.win = &WB;
          X.spec.b = &WB;
          X.spec_valid = ((t + s + j) % 5) != 0;            // exercise both paths: most macroblocks speculate, every fifth does not
          if (X.spec_valid) wh_win_speculate (P, jobs[j], X.spec, mbx, mby, last_mv);
          X.last_mv = &last_mv;
          if (P.flags & WH_SEQ_SCC) wh_inter_mb_body_t<true> (S, G, P, jobs[j], mbx, mby, X);
          else if (plain && WH_PLAIN_KERNEL == 2 && P.complexity == 0) wh_inter_mb_body_t<false, 2> (S, G, P, jobs[j], mbx, mby, X);
          else if (plain) wh_inter_mb_body_t<false, 1> (S, G, P, jobs[j], mbx, mby, X);
          else if (no_ctrl) wh_inter_mb_body_t<false, 3> (S, G, P, jobs[j], mbx, mby, X);
          else wh_inter_mb_body_t<false> (S, G, P, jobs[j], mbx, mby, X);
          // WELSHIP_EMU_CORRUPT_MB=<xy> (tests/test_hooks_dynslice.py): one level of that macroblock's record is off by one whenever a RANGED
          // call codes it -- what a lost update between the slice tasks' device calls would look like.  The harness must report it.
          if (jobs[j].mb_end > 0) if (const char* cm = getenv ("WELSHIP_EMU_CORRUPT_MB")) if (atoi (cm) == xy) {
            WhMbRecord& R = ((WhMbRecord*)jobs[j].records)[xy];
            int16_t* lv = &R.luma[0][0];
            for (int i = 0; i < 16 * 16; ++i) if (lv[i] > 1 || lv[i] < -1) { lv[i] += lv[i] > 0 ? 1 : -1; break; }
          }
          if (jobs[j].gom_rc) wh_gom_close_if_last (P, jobs[j], xy);
          poison (&S, sizeof (S)); poison (&WB, sizeof (WB)); poison (&G, sizeof (G));      // nothing survives from one macroblock to the next
        }
      }
  }
  void run_deblock (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    // one emulated wavefront per band (band b after band b-1, which satisfies the seam dependencies), with the device
    // scheduler's one-MB look-ahead and its strip exchange: neighbours inside the band through the exchange buffers,
    // neighbours in the previous band through the picture
    // WELSHIP_DB_WHOLE=1: the one-band table of the whole picture (what the device takes for large batches, hip_backend.hip run_deblock)
    const char* we = getenv ("WELSHIP_DB_WHOLE");
    const bool whole = P.deblock_idc == 0 && P.db_bands && we && atoi (we) != 0;
    const int32_t* bands = whole ? WH_DB_WHOLE_TABLE (P) : P.db_bands;
    const int nbands = whole ? 1 : P.db_num_bands;
    for (int j = 0; j < n; ++j)
      for (int s = 0; s < nbands; ++s) {
        WhDbLds S;
        WhDbStage G;
        std::vector<uint32_t> xb (wh_db_xchg_words (P.mb_w, P.mb_h), 0xA5A5A5A5u);
        WhDbXchg E;
        E.top = xb.data(); E.left = xb.data() + (size_t)P.mb_w * 24; E.first_row = 0;
        poison (&S, sizeof (S)); poison (&G, sizeof (G));
        const int first = bands[s], last = bands[s + 1];
        if (whole && jobs[j].rec_blk) {
          // the whole picture in items of one or two macroblocks (common/mb_order.h wh_build_db_pair_items), as the device's k_deblock_pairs
          const uint32_t* items = P.mb_order + 3 * (size_t)P.mb_w * P.mb_h;
          WhDbLds S2[2];
          WhDbStage G2[2];
          for (uint32_t t = 0; t < items[0]; ++t) {
            const int xy = WH_DB_ITEM_Y (items[1 + t]) * P.mb_w + WH_DB_ITEM_X (items[1 + t]), xb = xy + P.mb_w - 2;
            const bool pair = (items[1 + t] & WH_DB_ITEM_PAIR) != 0;
            poison (S2, sizeof (S2)); poison (G2, sizeof (G2));
            for (int lane = 0; lane < 64; ++lane) wh_deblock_cold_fetch (G2[0], lane, P, jobs[j], xy % P.mb_w, xy / P.mb_w);
            if (pair) {
              if (getenv ("WELSHIP_EMU_DB_STATS")) { static long pairs_seen = 0; if ((++pairs_seen & 1023) == 1) fprintf (stderr, "emu: deblocking pair %ld (MB %d + %d)\n", pairs_seen, xy, xb); }
              for (int lane = 0; lane < 64; ++lane) wh_deblock_cold_fetch (G2[1], lane, P, jobs[j], xb % P.mb_w, xb / P.mb_w);
              wh_deblock_pair_body (S2, G2, E, P, jobs[j], xy % P.mb_w, xy / P.mb_w, xb % P.mb_w, xb / P.mb_w);
            } else (void)wh_deblock_mb_body (S2[0], G2[0], E, first, last, P, jobs[j], xy % P.mb_w, xy / P.mb_w, 0, 0, 0, false);
          }
          continue;
        }
        const uint32_t* order = whole ? P.mb_order + P.mb_w * P.mb_h : P.mb_order + 2 * P.mb_w * P.mb_h;
        for (int lane = 0; lane < 64; ++lane) wh_deblock_cold_fetch (G, lane, P, jobs[j], order[first] % P.mb_w, order[first] / P.mb_w);
        for (int t = first; t < last; ++t) {
          const int xy = order[t], xyn = t + 1 < last ? order[t + 1] : 0;
          (void)wh_deblock_mb_body (S, G, E, first, last, P, jobs[j], xy % P.mb_w, xy / P.mb_w, t + 1 < last, xyn % P.mb_w, xyn / P.mb_w, true);
          poison (&S, sizeof (S));
        }
      }
  }
  void run_scene (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for (int j = 0; j < n; ++j)
      for (int y = 0; y < P.mb_h; ++y)
        for (int x = 0; x < P.mb_w; ++x) wh_scene_mb_body (P, jobs[j], x, y);
  }
  void run_qp_chain (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for (int j = 0; j < n; ++j)
      for (int s = 0; s < P.num_slices; ++s) wh_qp_chain_slice (P, jobs[j], P.slice_first_mb[s], P.slice_first_mb[s + 1]);
  }
  void run_expand (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for (int j = 0; j < n; ++j) {
      const int total = wh_expand_items (P);
      for (int i = 0; i < total; ++i) wh_expand_item (P, (uint8_t*)jobs[j].rec[0], (uint8_t*)jobs[j].rec[1], (uint8_t*)jobs[j].rec[2], i);
      if (jobs[j].rec_blk) { for (int i = 0; i < wh_tile_border_items (P); ++i) wh_tile_border_item (P, jobs[j], i); }     // (the deblocking pass wrote the tiles inside the picture)
      else for (int i = 0; i < wh_tile_items (P); ++i) wh_tile_item (P, jobs[j], i);       // the tiled twin (kernels/tile_pic.h), as the device's run_expand
    }
  }
  void run_src_tile (const WhSeqParams& P, const uint8_t* planar, uint8_t* tiled) override {
    for (int i = 0; i < wh_src_tile_items (P); ++i) wh_src_tile_item (P, planar, tiled, i);
  }
  void run_vaa (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, int32_t* sad8x8, int32_t* sd8x8, uint8_t* mad8x8, int32_t* sum16, int32_t* sqsum16, int32_t* ssd16) override {
    const WhVaaOut o = {sad8x8, sd8x8, mad8x8, sum16, sqsum16, ssd16};
    for (int xy = 0; xy < P.mb_w * P.mb_h; ++xy) wh_vaa_mb (cur, ref, xy, o);
  }
  void run_vaa_skewed (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, int stride, int width, int height, int32_t* sad8x8, int32_t* sd8x8, uint8_t* mad8x8,
                       int32_t* sum16, int32_t* sqsum16, int32_t* ssd16) override {
    const WhVaaOut o = {sad8x8, sd8x8, mad8x8, sum16, sqsum16, ssd16};
    for (int i = 0; i < (height >> 4); ++i) for (int j = 0; j < (width >> 4); ++j) wh_vaa_mb_skewed (cur, ref, stride, width, P.mb_w, j, i, o);
  }
  void run_bgd (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, const int32_t* sad8x8, const int32_t* sd8x8, const uint8_t* mad8x8, int units_w, int units_h,
                int8_t* flags) override {
    // the device's schedule: the units of one diagonal (i + 2 j) in any order -- here from the bottom up, the opposite of the raster pass
    const WhBgdIn in = {sad8x8, sd8x8, mad8x8, cur, ref, units_w, units_h, P.mb_w};
    std::vector<uint8_t> fl ((size_t)units_w * units_h);
    for (int k = 0; k < units_w * units_h; ++k) fl[k] = (uint8_t)wh_bgd_coarse (wh_bgd_ou (in, k % units_w, k / units_w));
    for (int t = 0; t < wh_bgd_steps (units_w, units_h); ++t)
      for (int j = units_h - 1; j >= 0; --j) { const int i = t - 2 * j; if (i >= 0 && i < units_w) wh_bgd_step (in, fl.data(), flags, i, j); }
  }
  void run_compact (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for (int j = 0; j < n; ++j) {
      if (!jobs[j].compact || !jobs[j].compact_off) continue;
      uint32_t off = 0;
      for (int xy = 0; xy < P.mb_w * P.mb_h; ++xy) { jobs[j].compact_off[xy] = off; off += wh_compact_pack (&jobs[j].records[xy], jobs[j].compact + off); }
      jobs[j].compact_off[P.mb_w * P.mb_h] = off;
    }
  }
  void select_queue (int) override {}
  int sync() override { return 0; }
  void upload_on (int, void* dst, const void* src, size_t bytes) override { memcpy (dst, src, bytes); }
  void download_on (int, void* dst, const void* src, size_t bytes) override { memcpy (dst, src, bytes); }
  void event_record_on (int, void* ev) override { event_record (ev); }
  // fault injection for the host logic's tests: WELSHIP_EMU_ERR_SNAPSHOT_AT=k makes the k-th copy of a queue's error words (1-based) report a time-out;
  // the words stay set until the queue is synchronised, as on the device
  int snaps_ = 0; uint32_t sticky_err_[64] = {0};
  void err_snapshot (int q, uint32_t* dst) override {
    const char* e = getenv ("WELSHIP_EMU_ERR_SNAPSHOT_AT");
    q = (q < 0 ? 0 : q) % 64;
    if (e && ++snaps_ == atoi (e)) sticky_err_[q] = 1;
    if (dst) { dst[0] = sticky_err_[q]; dst[1] = dst[2] = dst[3] = 0; }
  }
  int sync_queue (int q) override { q = (q < 0 ? 0 : q) % 64; const int bad = sticky_err_[q] != 0; sticky_err_[q] = 0; return bad; }
  void event_wait (void*) override {}
  void queue_wait_event (int, void*) override {}
  void run_src_tile_jobs (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for (int i = 0; i < n; ++i) if (jobs[i].src[1]) run_src_tile (P, jobs[i].src[1], (uint8_t*)jobs[i].src[0]);
  }
  void* event_create() override { return new double (0.0); }
  void event_destroy (void* ev) override { delete (double*)ev; }
  void event_record (void* ev) override { * (double*)ev = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now().time_since_epoch()).count(); }
  float event_elapsed_ms (void* a, void* b) override { return (float) (* (double*)b - * (double*)a); }
};

// WELSHIP_TRACE_DEVICES=1: which device index every backend is asked for (tests of the layer -> device and rank -> device mappings)
Backend* create_default_backend (int device, const char**) {
  if (getenv ("WELSHIP_TRACE_DEVICES")) { fprintf (stderr, "welship emu: backend for device %d\n", device); fflush (stderr); }
  return new EmuBackend();
}

}  // namespace wh

// The picture-level down-sampling entry point of the product library (hip/downsample.hip), for the CPU test tier: the same plan
// (wh_ds_plan) and the same per-sample functions, run as plain loops.
extern "C" int WelsHipDownsamplePicture (int, uint8_t* const pDst[3], const int32_t iDstStride[3], int32_t iDstWidth, int32_t iDstHeight,
                                         const uint8_t* const pSrc[3], const int32_t iSrcStride[3], int32_t iSrcWidth, int32_t iSrcHeight) {
  if (!pDst || !pSrc || !iDstStride || !iSrcStride || iDstWidth < 2 || iDstHeight < 2 || iSrcWidth <= iDstWidth || iSrcHeight <= iDstHeight) return WELSHIP_ERR_INIT_PARA;
  WhDsStage plan[8];
  const int nst = wh_ds_plan (iSrcWidth, iSrcHeight, iDstWidth, iDstHeight, plan);
  if (nst < 1) return WELSHIP_ERR_INIT_PARA;
  for (int pl = 0; pl < 3; ++pl) {
    const int sh_ = pl ? 1 : 0;
    // this plane of the source with one spare row (addressed by the general filter's lower tap, never weighted)
    int w = iSrcWidth >> sh_, h = iSrcHeight >> sh_, stride = (w + 63) & ~63;
    std::vector<uint8_t> cur ((size_t)stride * (h + 1) + 64, 0);
    for (int r = 0; r < h; ++r) memcpy (cur.data() + (size_t)r * stride, pSrc[pl] + (size_t)r * iSrcStride[pl], (size_t)w);
    memcpy (cur.data() + (size_t)h * stride, cur.data() + (size_t) (h - 1) * stride, (size_t)w);
    for (int k = 0; k < nst; ++k) {
      const WhDsStage& g = plan[k];
      const int dw = g.dw >> sh_, dh = g.dh >> sh_, sw = g.sw >> sh_, sh2 = g.sh >> sh_, ds = (dw + 63) & ~63;
      std::vector<uint8_t> out ((size_t)ds * (dh + 1) + 64, 0);
      const int step = g.mode == 0 ? 2 : g.mode == 1 ? 4 : 3;
      const int accurate = pl ? 1 : 0;
      const int scx = wh_ds_round_scale (sw, dw, accurate ? 15 : 16), scy = wh_ds_round_scale (sh2, dh, 15);
      for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x)
          out[(size_t)y * ds + x] = g.mode >= 0 ? wh_ds_avg2x2 (cur.data() + (size_t) (step * y) * stride + step * x, stride)
                                                : wh_ds_general (cur.data(), stride, dw, dh, x, y, scx, scy, accurate);
      memcpy (out.data() + (size_t)dh * ds, out.data() + (size_t) (dh - 1) * ds, (size_t)dw);
      cur.swap (out); stride = ds; w = dw; h = dh;
    }
    for (int r = 0; r < h; ++r) memcpy (pDst[pl] + (size_t)r * iDstStride[pl], cur.data() + (size_t)r * stride, (size_t)w);
  }
  return WELSHIP_OK;
}
