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
#include "../../openh264_amd/csrc/host/backend.h"
#include "../../openh264_amd/csrc/kernels/frame_kernels.h"
#include "../../openh264_amd/csrc/kernels/deblock_mb.h"
#include "../../openh264_amd/csrc/kernels/inter_mb.h"
#include "../../openh264_amd/csrc/kernels/expand_pic.h"

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
    for_order (P, n, false, [&] (int j, int x, int y) { WhMbLds S; poison (&S, sizeof (S)); wh_intra_mb_body (S, P, jobs[j], x, y); });
  }
  void run_inter (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for_order (P, n, false, [&] (int j, int x, int y) {
      WhInterLds S;
      poison (&S, sizeof (S));
      for (int lane = 0; lane < 64; ++lane) wh_inter_cold_fetch (S, lane, P, jobs[j], x, y);
      WhInterCtx X;
      X.slice_idc = wh_slice_of_mb (P, y * P.mb_w + x); X.slice_first = P.slice_first_mb[X.slice_idc];
      X.next_valid = 0; X.next_mbx = X.next_mby = 0;
      wh_inter_mb_body (S, P, jobs[j], x, y, X);
    });
  }
  void run_deblock (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for_order (P, n, true, [&] (int j, int x, int y) { WhDbLds S; poison (&S, sizeof (S)); wh_deblock_mb_body (S, P, jobs[j], x, y); });
  }
  void run_expand (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    for (int j = 0; j < n; ++j) {
      const int nb = wh_expand_num_blocks (P);
      for (int b = 0; b < nb; ++b) wh_expand_body (P, jobs[j], b);
    }
  }
  void select_queue (int) override {}
  void sync() override {}
  void* event_create() override { return new double (0.0); }
  void event_destroy (void* ev) override { delete (double*)ev; }
  void event_record (void* ev) override { * (double*)ev = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now().time_since_epoch()).count(); }
  float event_elapsed_ms (void* a, void* b) override { return (float) (* (double*)b - * (double*)a); }
};

Backend* create_default_backend (int, const char**) { return new EmuBackend(); }

}  // namespace wh
