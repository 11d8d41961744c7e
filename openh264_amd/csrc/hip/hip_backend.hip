// hip_backend.hip -- gfx950 launchers for the macroblock kernels + the wh::Backend implementation.
//
// Launch geometry: one workgroup = one wavefront (64 lanes) = one macroblock.  A frame-level pass
// walks the 2:1 diagonals of the MB grid (frame_kernels.h); each diagonal is one launch whose grid
// is (MBs on the diagonal) x (pictures in the batch), so independent pictures (all-IDR streams,
// simulcast layers, concurrent sessions) fill the 256 CUs while a single picture only offers
// <= mb_w/2 parallel MBs.  Kernel boundaries provide the inter-MB ordering and visibility; no
// in-kernel spinning, so a launch can never hang the device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include "../host/backend.h"
#include "../kernels/frame_kernels.h"
#include "../kernels/deblock_mb.h"
#include "../kernels/inter_mb.h"
#include "../kernels/expand_pic.h"

namespace {

__global__ __launch_bounds__ (64) void k_intra_diag (WhSeqParams P, const WhPicJob* jobs, int d, int y0) {
  __shared__ WhMbLds S;
  const WhPicJob J = jobs[blockIdx.y];
  const int y = y0 + (int)blockIdx.x, x = d - 2 * y;
  wh_intra_mb_body (S, P, J, x, y);
}
__global__ __launch_bounds__ (64) void k_inter_diag (WhSeqParams P, const WhPicJob* jobs, int d, int y0) {
  __shared__ WhInterLds S;
  const WhPicJob J = jobs[blockIdx.y];
  const int y = y0 + (int)blockIdx.x, x = d - 2 * y;
  wh_inter_mb_body (S, P, J, x, y);
}
__global__ __launch_bounds__ (64) void k_deblock_diag (WhSeqParams P, const WhPicJob* jobs, int d, int y0) {
  __shared__ WhDbLds S;
  const WhPicJob J = jobs[blockIdx.y];
  const int y = y0 + (int)blockIdx.x, x = d - 2 * y;
  wh_deblock_mb_body (S, P, J, x, y);
}
__global__ __launch_bounds__ (64) void k_expand (WhSeqParams P, const WhPicJob* jobs) {
  const WhPicJob J = jobs[blockIdx.y];
  wh_expand_body (P, J, (int)blockIdx.x);
}

#define HIP_CHECK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf (stderr, "welship: HIP error %s at %s:%d\n", hipGetErrorString (_e), __FILE__, __LINE__); abort(); } } while (0)

class HipBackend : public wh::Backend {
 public:
  HipBackend (int dev, const hipDeviceProp_t& prop) : dev_ (dev) {
    HIP_CHECK (hipSetDevice (dev_));
    HIP_CHECK (hipStreamCreateWithFlags (&stream_, hipStreamNonBlocking));
    name_ = std::string ("hip:") + prop.gcnArchName + " " + prop.name;
  }
  ~HipBackend() override { (void)hipSetDevice (dev_); (void)hipStreamSynchronize (stream_); (void)hipStreamDestroy (stream_); }
  const char* name() const override { return name_.c_str(); }
  void* alloc (size_t bytes) override { void* p = nullptr; HIP_CHECK (hipSetDevice (dev_)); HIP_CHECK (hipMalloc (&p, bytes ? bytes : 1)); return p; }
  void free (void* p) override { HIP_CHECK (hipSetDevice (dev_)); HIP_CHECK (hipFree (p)); }
  void upload (void* dst, const void* src, size_t bytes) override { HIP_CHECK (hipMemcpyAsync (dst, src, bytes, hipMemcpyHostToDevice, stream_)); }
  void download (void* dst, const void* src, size_t bytes) override { HIP_CHECK (hipMemcpyAsync (dst, src, bytes, hipMemcpyDeviceToHost, stream_)); }
  void fill (void* dst, int value, size_t bytes) override { HIP_CHECK (hipMemsetAsync (dst, value, bytes, stream_)); }

  template <class K> void diagonals (K kernel, const WhSeqParams& P, const WhPicJob* jobs, int n) {
    const int nd = (P.mb_w - 1) + 2 * (P.mb_h - 1) + 1;
    for (int d = 0; d < nd; ++d) {
      int y0;
      const int cnt = wh_diag_count (P.mb_w, P.mb_h, d, &y0);
      if (cnt <= 0) continue;
      hipLaunchKernelGGL (kernel, dim3 (cnt, n), dim3 (64), 0, stream_, P, jobs, d, y0);
    }
    HIP_CHECK (hipGetLastError());
  }
  void run_intra (const WhSeqParams& P, const WhPicJob* jobs, int n) override { diagonals (k_intra_diag, P, jobs, n); }
  void run_inter (const WhSeqParams& P, const WhPicJob* jobs, int n) override { diagonals (k_inter_diag, P, jobs, n); }
  void run_deblock (const WhSeqParams& P, const WhPicJob* jobs, int n) override { diagonals (k_deblock_diag, P, jobs, n); }
  void run_expand (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    hipLaunchKernelGGL (k_expand, dim3 (wh_expand_num_blocks (P), n), dim3 (64), 0, stream_, P, jobs);
    HIP_CHECK (hipGetLastError());
  }
  void sync() override { HIP_CHECK (hipStreamSynchronize (stream_)); }
  void* event_create() override { hipEvent_t e; HIP_CHECK (hipEventCreate (&e)); return (void*)e; }
  void event_destroy (void* ev) override { HIP_CHECK (hipEventDestroy ((hipEvent_t)ev)); }
  void event_record (void* ev) override { HIP_CHECK (hipEventRecord ((hipEvent_t)ev, stream_)); }
  float event_elapsed_ms (void* a, void* b) override { float ms = 0.f; HIP_CHECK (hipEventSynchronize ((hipEvent_t)b)); HIP_CHECK (hipEventElapsedTime (&ms, (hipEvent_t)a, (hipEvent_t)b)); return ms; }
  hipStream_t stream() const { return stream_; }
 private:
  int dev_;
  hipStream_t stream_ = nullptr;
  std::string name_;
};

}  // namespace

namespace wh {

Backend* create_hip_backend (int device, const char** err) {
  int count = 0;
  if (hipGetDeviceCount (&count) != hipSuccess || count <= 0) { if (err) *err = "no HIP device visible (libwelship needs an MI355X; there is no CPU fallback)"; return nullptr; }
  if (device < 0 || device >= count) { if (err) *err = "HIP device ordinal out of range"; return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties (&prop, device) != hipSuccess) { if (err) *err = "hipGetDeviceProperties failed"; return nullptr; }
  if (std::string (prop.gcnArchName).find ("gfx950") == std::string::npos) { if (err) *err = "device is not gfx950 (this library is built for MI355X only)"; return nullptr; }
  return new HipBackend (device, prop);
}
Backend* create_default_backend (int device, const char** err) { return create_hip_backend (device, err); }

}  // namespace wh
