// frame_api.cpp -- the explicit frame API of include/welship.h (2b): what the SWelsFuncPtrList hooks of the patched reference call
// (integration/welship_hooks.cpp).  Split from encoder.cpp in round 6; the session API and the session groups are there.
#include "encoder_internal.h"
using wh::align_up; using wh::rec_blocks_on; using wh::DevPicture;

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
  WhHostVec<uint8_t> h_src;
  WhMbRecord* d_records = nullptr;
  uint8_t* d_rec_blk = nullptr;       // (SessionCore::d_rec_blk)
  // what the last WelsHipFrameVaa call left on the device for WelsHipFrameBgd: the pair's keys and pool slots, whether the background statistics were computed
  const void* vaa_cur_key = nullptr; const void* vaa_ref_key = nullptr; int vaa_cslot = -1, vaa_rslot = -1, vaa_queue = 0; bool vaa_has_bgd = false;
  uint64_t vaa_cur_sum = 0, vaa_ref_sum = 0;      // what the two slots held when the statistics were made (SrcSlot::luma_sum): a reused slot fails WelsHipFrameBgd
  int8_t* d_bgd_calc = nullptr;       // WelsHipFrameBgd's result (one flag per macroblock)
  WhHostVec<int8_t> h_bgd_calc;
  uint8_t* d_skew = nullptr;          // pre-analysis of a picture whose width is no multiple of 16: the two luma planes at the caller's stride (WelsHipFrameVaa)
  WhHostVec<uint8_t> h_skew;
  size_t skew_bytes = 0;
  WhHostVec<WhMbRecord> h_records;
  // packed records of whole-picture calls (WelsHipFrameJob::bPackedRecords; common/compact.h): device stream + offsets, page-locked host copies.
  // The host copy is brought back in one go up to `compact_est` bytes (a little more than the previous picture's size); the rare rest follows.
  uint8_t* d_compact = nullptr;
  uint32_t* d_compact_off = nullptr;
  WhHostVec<uint8_t> h_compact;
  WhHostVec<uint32_t> h_coff;
  size_t compact_est = 0, compact_got = 0;
  WelsHipPackedRecords packed_view = {nullptr, nullptr};
  WhHostVec<uint8_t> h_pic;
  // page-locked staging for the small per-picture arrays of the caller (pageable copies block on the queue: with the shared
  // lock held that stalls every other session): VAA SADs | pSadCost in | inter-layer hints | background flags, and pSadCost out
  WhHostVec<uint8_t> h_aux, h_sad_out;
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
  WhHostVec<WhMbCtl> h_mb_ctl;
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
      WhHostVec<int32_t> bands (3 * (size_t) (mb_h + n) + 1);
      const int nb = wh_build_db_bands (mb_w, mb_h, n, first, idc, WH_DB_BAND_ROWS, bands.data(), (int)bands.size());
      if (nb < 1) { set_err ("deblocking band table"); return WELSHIP_ERR_UNKNOWN; }
      for (int b = 0; b < nb; ++b) wh_build_mb_order (mb_w, bands[b], bands[b + 1], order.data() + 2 * (size_t)num_mb + bands[b]);
      WhHostVec<uint32_t> order32 (order.begin(), order.end());
      order32.resize ((size_t)num_mb * 4 + 1);                        // + the whole-picture deblocking order as items of one or two macroblocks (common/mb_order.h)
      wh_build_db_pair_items (mb_w, mb_h, wh::db_pair_min_len(), order32.data() + 3 * (size_t)num_mb);
      up->d_order = (uint32_t*)be->alloc (order32.size() * 4);
      up->d_bands = (int32_t*)be->alloc (sizeof (int32_t) * (3 * (size_t)nb + 1 + 4));
      if (!up->d_order || !up->d_bands) { set_err ("out of device memory"); return WELSHIP_ERR_MEMORY; }
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
  WhHostVec<uint8_t>& tmp = c->h_pic;
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
