// backend.h -- the narrow interface between the host-side encoder and the device that runs the
// macroblock kernels.  The product implements it with HIP on gfx950 (hip/hip_backend.hip); the
// CPU-only test build implements it with the wave-emulation of the same kernel sources
// (tests/emu/emu_backend.cpp) so that host logic can be tested without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../common/wh_types.h"

namespace wh {

class Backend {
 public:
  virtual ~Backend() {}
  virtual const char* name() const = 0;
  // device memory
  virtual void* alloc (size_t bytes) = 0;
  virtual void free (void* p) = 0;
  virtual void upload (void* dst, const void* src, size_t bytes) = 0;
  virtual void download (void* dst, const void* src, size_t bytes) = 0;
  virtual void fill (void* dst, int value, size_t bytes) = 0;
  // source picture as uploaded (planar I420, tight strides of P) -> the macroblock-tiled layout the kernels read (WH_SRC_*), on the selected queue
  virtual void run_src_tile (const WhSeqParams& P, const uint8_t* planar, uint8_t* tiled) = 0;
  // pre-analysis statistics of the tiled source picture `cur` against `ref` (kernels/vaa_pic.h), every macroblock of the MB-aligned picture
  virtual void run_vaa (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, int32_t* sad8x8, int32_t* sd8x8, uint8_t* mad8x8,
                        int32_t* sum16, int32_t* sqsum16, int32_t* ssd16) = 0;
  // the same for a picture whose width is no multiple of 16 (kernels/vaa_pic.h wh_vaa_mb_skewed): `cur` / `ref` are the two luma planes as the
  // caller has them, `stride` bytes per line; the (width >> 4) x (height >> 4) macroblocks the C functions cover
  virtual void run_vaa_skewed (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, int stride, int width, int height, int32_t* sad8x8, int32_t* sd8x8,
                               uint8_t* mad8x8, int32_t* sum16, int32_t* sqsum16, int32_t* ssd16) = 0;
  // background detection of the picture the statistics above belong to (kernels/bgd_pic.h): units = (width >> 4) x (height >> 4) macroblocks,
  // `flags` [row of mb_w][unit] on the device; the statistics arrays are the ones run_vaa wrote
  virtual void run_bgd (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, const int32_t* sad8x8, const int32_t* sd8x8, const uint8_t* mad8x8,
                        int units_w, int units_h, int8_t* flags) = 0;
  // page-lock a host buffer that is the target of many downloads (best effort; no-op where it does not apply)
  virtual void pin_host (void* p, size_t bytes) { (void)p; (void)bytes; }
  virtual void unpin_host (void* p) { (void)p; }
  // frame-level kernels over `n` pictures that share the sequence parameters P.
  // `jobs` is a DEVICE array of WhPicJob.  All calls are asynchronous on the backend's stream.
  virtual void run_intra (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;     // I pictures: MD + recon
  virtual void run_inter (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;     // P pictures: ME + MD + recon
  virtual void run_scene (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;     // source pictures: scene-change statistic
  virtual void run_qp_chain (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;  // pictures with a QP map: QP_Y chain for the filter
  virtual void run_deblock (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;   // in-loop filter on rec[]
  virtual void run_expand (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;    // make rec[] a reference: replicate its borders (32/16 px), write its tiled twin rec_tiles[]
  virtual void run_compact (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;   // pack records[] into compact / compact_off (common/compact.h)
  // Independent in-order queues (HIP streams): everything issued after select_queue (k) goes to queue k; work on
  // different queues may overlap on the device.  sync() waits for all of them and returns 0, or the number of
  // in-kernel dependency waits that timed out since the last sync (the pictures of that step are then invalid).
  virtual void select_queue (int k) = 0;
  // the selected queue waits (on the device, not the host) for everything issued so far on queue `from`
  virtual void queue_wait (int from) { (void)from; }
  virtual int sync() = 0;
  // the same for queue k only (other queues keep running)
  virtual int sync_queue (int k) { (void)k; return sync(); }
  // Every queue has its own error word, so sync_queue (k) reports queue k's launch set only.  sync() reads and resets all of them:
  // a thread that waits for its own queue outside the callers' lock compares this counter before the launch and after the wait, and
  // fails its pictures when a sync() in between found errors (it may have consumed this queue's verdict).
  virtual unsigned errors_swept() const { return 0; }
  // the dependency-wait time-outs queue k's kernels have reported so far, WITHOUT waiting for the queue (a pipelined caller checks the
  // launch set it is about to read results of while the next one is already running on the same queue); read through queue `via`
  virtual int peek_queue_errors (int k, int via) { (void)k; (void)via; return 0; }
  // ---- calls that name their queue and leave the selected one alone: safe from several host threads at once (a pipelined group stages
  // and uploads the next step's pictures on worker threads while another thread downloads and entropy-codes the previous step's) ----
  virtual void upload_on (int q, void* dst, const void* src, size_t bytes) = 0;
  virtual void download_on (int q, void* dst, const void* src, size_t bytes) = 0;
  // queue q's error words as they are once everything queued so far has run, into `dst` (page-locked, 4 words): lets a caller that waits for an event
  // on q -- not for the whole queue -- see whether the kernels up to here timed out (sync_queue reads and resets the words themselves)
  virtual void err_snapshot (int q, uint32_t* dst) { (void)q; if (dst) dst[0] = 0; }
  virtual void event_record_on (int q, void* ev) = 0;
  virtual void event_wait (void* ev) = 0;                    // the host waits for the event
  virtual void queue_wait_event (int q, void* ev) = 0;       // queue q waits (on the device) for the event
  // the pictures' planar sources (jobs[i].src[1], as uploaded) -> their macroblock-tiled form (jobs[i].src[0]): run_src_tile for a batch, on the selected queue
  virtual void run_src_tile_jobs (const WhSeqParams& P, const WhPicJob* jobs, int n) = 0;
  // timing on the stream the kernels are launched on (HIP events)
  virtual void* event_create() = 0;
  virtual void event_destroy (void* ev) = 0;
  virtual void event_record (void* ev) = 0;
  virtual float event_elapsed_ms (void* a, void* b) = 0;   // waits for b
};

// Implemented by the HIP library only; returns NULL (and sets *err) when no MI355X is usable.
Backend* create_hip_backend (int device, const char** err);

}  // namespace wh
