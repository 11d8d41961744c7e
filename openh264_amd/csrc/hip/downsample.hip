// downsample.hip -- spatial down-sampling of source pictures for simulcast layers (BASELINE config 4).
//
// Reference behaviour restated (codec/processing/src/downsample):
//   downsamplefuncs.cpp:47-68    DyadicBilinearDownsampler_c            2:1, ((a+b+1)>>1 + (c+d+1)>>1 + 1) >> 1
//   downsamplefuncs.cpp:70-91    DyadicBilinearQuarterDownsampler_c     4:1, the same 2x2 average of every fourth sample
//   downsamplefuncs.cpp:93-113   DyadicBilinearOneThirdDownsampler_c    3:1, the same 2x2 average of every third sample
//   downsamplefuncs.cpp:115-181  GeneralBilinearFastDownsampler_c       any ratio, 16/15-bit weights, products truncated (luma)
//   downsamplefuncs.cpp:183-245  GeneralBilinearAccurateDownsampler_c   any ratio, 15-bit weights, 64-bit accumulation (chroma)
//   downsample.cpp:144-277       CDownsampling::Process                 which of them a layer pair uses (the C function table)
// These are the one part of the path that is a plain streaming kernel: no dependency between output samples, every source
// byte read once -- bound by HBM bandwidth.  One launch covers `n` planes (all pictures of a batch) through a plane table.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>
#include "../../../include/welship.h"
#include "../kernels/downsample_px.h"

namespace {

struct DsPlane { const uint8_t* src; uint8_t* dst; };

// 2:1 in both directions: one lane = 8 output samples = two 16-byte source rows
__global__ __launch_bounds__ (256) void k_ds_half (const DsPlane* planes, int src_stride, int dst_stride, int dst_w, int dst_h) {
  const DsPlane pl = planes[blockIdx.z];
  const int x8 = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x8 * 8 >= dst_w) return;
  const uint8_t* s = pl.src + (size_t) (2 * y) * src_stride + x8 * 16;
  uint8_t* d = pl.dst + (size_t)y * dst_stride + x8 * 8;
  if (x8 * 8 + 8 <= dst_w && ((uintptr_t)s & 15) == 0 && ((uintptr_t)d & 7) == 0 && (src_stride & 15) == 0) {
    const uint4 a = * (const uint4*)s, b = * (const uint4*) (s + src_stride);
    const uint32_t ra[4] = {a.x, a.y, a.z, a.w}, rb[4] = {b.x, b.y, b.z, b.w};
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // horizontal pairs: bytes 0,2 of lerp (w, w >> 8) = (p0+p1+1)>>1, (p2+p3+1)>>1; then the two rows
      const uint32_t ha = __builtin_amdgcn_lerp (ra[k], ra[k] >> 8, 0x01010101u), hb = __builtin_amdgcn_lerp (rb[k], rb[k] >> 8, 0x01010101u);
      const uint32_t v = __builtin_amdgcn_lerp (ha, hb, 0x01010101u);
      o[k >> 1] |= ((v & 0xffu) | ((v >> 8) & 0xff00u)) << (16 * (k & 1));
    }
    * (uint2*)d = make_uint2 (o[0], o[1]);
  } else {
    for (int i = 0; i < 8 && x8 * 8 + i < dst_w; ++i) d[i] = wh_ds_avg2x2 (s + 2 * i, src_stride);
  }
}

// 3:1 / 4:1: the 2x2 average at every `step`-th sample, one lane = one output sample
__global__ __launch_bounds__ (256) void k_ds_step (const DsPlane* planes, int src_stride, int dst_stride, int dst_w, int dst_h, int step) {
  const DsPlane pl = planes[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dst_w) return;
  pl.dst[(size_t)y * dst_stride + x] = wh_ds_avg2x2 (pl.src + (size_t) (step * y) * src_stride + step * x, src_stride);
}

// any ratio: `accurate` = GeneralBilinearAccurateDownsampler_c, else GeneralBilinearFastDownsampler_c; last column and last
// row are nearest-sample copies as in the reference
__global__ __launch_bounds__ (256) void k_ds_general (const DsPlane* planes, int src_stride, int dst_stride, int dst_w, int dst_h, int scalex, int scaley, int accurate) {
  const DsPlane pl = planes[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dst_w) return;
  const uint8_t out = wh_ds_general (pl.src, src_stride, dst_w, dst_h, x, y, scalex, scaley, accurate);
  pl.dst[(size_t)y * dst_stride + x] = out;
}

inline int round_scale (int src, int dst, int bits) { return wh_ds_round_scale (src, dst, bits); }

int launch (int mode, const DsPlane* d_planes, int n, int src_stride, int src_w, int src_h, int dst_stride, int dst_w, int dst_h, hipStream_t st) {
  switch (mode) {
  case WELSHIP_DS_HALF:
    hipLaunchKernelGGL (k_ds_half, dim3 ((dst_w + 8 * 256 - 1) / (8 * 256), dst_h, n), dim3 (256), 0, st, d_planes, src_stride, dst_stride, dst_w, dst_h);
    break;
  case WELSHIP_DS_QUARTER: case WELSHIP_DS_ONE_THIRD:
    hipLaunchKernelGGL (k_ds_step, dim3 ((dst_w + 255) / 256, dst_h, n), dim3 (256), 0, st, d_planes, src_stride, dst_stride, dst_w, dst_h, mode == WELSHIP_DS_QUARTER ? 4 : 3);
    break;
  case WELSHIP_DS_GENERAL_FAST: case WELSHIP_DS_GENERAL_ACCURATE: {
    const int acc = mode == WELSHIP_DS_GENERAL_ACCURATE;
    hipLaunchKernelGGL (k_ds_general, dim3 ((dst_w + 255) / 256, dst_h, n), dim3 (256), 0, st, d_planes, src_stride, dst_stride, dst_w, dst_h,
                        round_scale (src_w, dst_w, acc ? 15 : 16), round_scale (src_h, dst_h, 15), acc);
    break;
  }
  default: return WELSHIP_ERR_INIT_PARA;
  }
  return hipGetLastError() == hipSuccess ? WELSHIP_OK : WELSHIP_ERR_UNKNOWN;
}

}  // namespace

extern "C" {

int WelsHipPrimDownsample (int mode, uint8_t* pDst, int32_t iDstStride, int32_t iDstWidth, int32_t iDstHeight,
                           const uint8_t* pSrc, int32_t iSrcStride, int32_t iSrcWidth, int32_t iSrcHeight) {
  int cnt = 0;
  if (hipGetDeviceCount (&cnt) != hipSuccess || cnt <= 0) return WELSHIP_ERR_NO_DEVICE;
  if (!pDst || !pSrc || iDstWidth < 1 || iDstHeight < 1 || iSrcWidth <= iDstWidth || iSrcHeight <= iDstHeight) return WELSHIP_ERR_INIT_PARA;
  // the sample below / right of the last one a filter tap touches must be readable: callers pass padded planes like the
  // reference does (iSrcStride * (iSrcHeight + 1) bytes)
  const size_t sb = (size_t)iSrcStride * (iSrcHeight + 1), db = (size_t)iDstStride * iDstHeight;
  uint8_t *ds = nullptr, *dd = nullptr;
  DsPlane* dp = nullptr;
  int rc = WELSHIP_ERR_MEMORY;
  if (hipMalloc ((void**)&ds, sb + 64) == hipSuccess && hipMalloc ((void**)&dd, db) == hipSuccess && hipMalloc ((void**)&dp, sizeof (DsPlane)) == hipSuccess) {
    (void)hipMemset (ds, 0, sb + 64);
    (void)hipMemcpy (ds, pSrc, (size_t)iSrcStride * iSrcHeight, hipMemcpyHostToDevice);
    (void)hipMemcpy (dd, pDst, db, hipMemcpyHostToDevice);
    const DsPlane pl = {ds, dd};
    (void)hipMemcpy (dp, &pl, sizeof (pl), hipMemcpyHostToDevice);
    rc = launch (mode, dp, 1, iSrcStride, iSrcWidth, iSrcHeight, iDstStride, iDstWidth, iDstHeight, 0);
    if (rc == WELSHIP_OK && hipDeviceSynchronize() != hipSuccess) rc = WELSHIP_ERR_UNKNOWN;
    if (rc == WELSHIP_OK) (void)hipMemcpy (pDst, dd, db, hipMemcpyDeviceToHost);
  }
  if (ds) (void)hipFree (ds);
  if (dd) (void)hipFree (dd);
  if (dp) (void)hipFree (dp);
  return rc;
}

}  // extern "C"

namespace {

// ---- one picture, all three planes, the method CDownsampling::Process (downsample.cpp:144-277) would pick -------------------
// What the dispatch-table binding calls from the reference's pre-processing (integration/welship_hooks.cpp, DownsamplePadding,
// wels_preprocess.cpp:625-675) instead of m_pInterfaceVp->Process (METHOD_DOWNSAMPLE, ..): source planes up, the kernels of the
// cascade, destination planes back.  Per device one context with a queue, page-locked staging and device buffers that grow to
// the largest picture seen; calls on one device serialise on its mutex (a call lasts well under a millisecond).
struct DsContext {
  std::mutex mu;
  hipStream_t st = nullptr;
  uint8_t* h_buf = nullptr; size_t h_cap = 0;       // page-locked: source planes | destination planes
  uint8_t* d_buf = nullptr; size_t d_cap = 0;       // source | two intermediate pictures | destination
  DsPlane* d_planes = nullptr;                       // 3 planes per stage, up to 8 stages
  DsPlane* h_planes = nullptr;
  bool ok = false;
};
std::mutex g_ds_mu;
std::map<int, DsContext*> g_ds_ctx;

inline size_t al256 (size_t v) { return (v + 255) & ~ (size_t)255; }

}  // namespace

extern "C" int WelsHipDownsamplePicture (int iDevice, uint8_t* const pDst[3], const int32_t iDstStride[3], int32_t iDstWidth, int32_t iDstHeight,
                                         const uint8_t* const pSrc[3], const int32_t iSrcStride[3], int32_t iSrcWidth, int32_t iSrcHeight) {
  int cnt = 0;
  if (hipGetDeviceCount (&cnt) != hipSuccess || cnt <= 0) return WELSHIP_ERR_NO_DEVICE;
  if (iDevice < 0 || iDevice >= cnt) return WELSHIP_ERR_NO_DEVICE;
  if (!pDst || !pSrc || !iDstStride || !iSrcStride || iDstWidth < 2 || iDstHeight < 2 || iSrcWidth <= iDstWidth || iSrcHeight <= iDstHeight ||
      iSrcWidth > 8192 || iSrcHeight > 8192) return WELSHIP_ERR_INIT_PARA;
  for (int i = 0; i < 3; ++i) if (!pDst[i] || !pSrc[i] || iSrcStride[i] < (i ? iSrcWidth >> 1 : iSrcWidth) || iDstStride[i] < (i ? iDstWidth >> 1 : iDstWidth)) return WELSHIP_ERR_INIT_PARA;
  DsContext* c = nullptr;
  {
    std::lock_guard<std::mutex> reg (g_ds_mu);
    DsContext*& slot = g_ds_ctx[iDevice];
    if (!slot) slot = new DsContext();
    c = slot;
  }
  std::lock_guard<std::mutex> lock (c->mu);
  if (hipSetDevice (iDevice) != hipSuccess) return WELSHIP_ERR_NO_DEVICE;
  if (!c->ok) {
    // (a call that fails here leaves what it got in the context: the next call continues from there instead of creating another stream)
    if (!c->st && hipStreamCreateWithFlags (&c->st, hipStreamNonBlocking) != hipSuccess) { c->st = nullptr; return WELSHIP_ERR_UNKNOWN; }
    if (!c->d_planes && hipMalloc ((void**)&c->d_planes, sizeof (DsPlane) * 24) != hipSuccess) { c->d_planes = nullptr; (void)hipGetLastError(); return WELSHIP_ERR_MEMORY; }
    if (!c->h_planes && hipHostMalloc ((void**)&c->h_planes, sizeof (DsPlane) * 24) != hipSuccess) { c->h_planes = nullptr; (void)hipGetLastError(); return WELSHIP_ERR_MEMORY; }
    c->ok = true;
  }
  // the stages CDownsampling::Process would take (kernels/downsample_px.h wh_ds_plan)
  typedef WhDsStage Stage;
  WhDsStage plan[8];
  const int nst = wh_ds_plan (iSrcWidth, iSrcHeight, iDstWidth, iDstHeight, plan);
  if (nst < 1) return WELSHIP_ERR_INIT_PARA;
  const std::vector<Stage> stages (plan, plan + nst);
  // buffers: tight strides (64-byte multiples), one spare row per plane (the general filter's lower tap)
  auto pitch = [] (int w) { return (w + 63) & ~63; };
  auto pic_bytes = [&] (int w, int h) { return al256 ((size_t)pitch (w) * (h + 1)) + 2 * al256 ((size_t)pitch (w >> 1) * ((h >> 1) + 1)); };
  const size_t src_b = pic_bytes (iSrcWidth, iSrcHeight), dst_b = pic_bytes (iDstWidth, iDstHeight), tmp_b = pic_bytes (iSrcWidth >> 1, iSrcHeight >> 1);
  const size_t need_d = src_b + 2 * tmp_b + dst_b, need_h = src_b + dst_b;
  if (need_d > c->d_cap) { if (c->d_buf) (void)hipFree (c->d_buf); c->d_buf = nullptr; c->d_cap = 0; if (hipMalloc ((void**)&c->d_buf, need_d) != hipSuccess) return WELSHIP_ERR_MEMORY; c->d_cap = need_d; }
  if (need_h > c->h_cap) { if (c->h_buf) (void)hipHostFree (c->h_buf); c->h_buf = nullptr; c->h_cap = 0; if (hipHostMalloc ((void**)&c->h_buf, need_h) != hipSuccess) return WELSHIP_ERR_MEMORY; c->h_cap = need_h; }
  struct Pic { uint8_t* p[3]; int stride[3]; };
  auto lay = [&] (uint8_t* base, int w, int h) { Pic q; q.stride[0] = pitch (w); q.stride[1] = q.stride[2] = pitch (w >> 1); q.p[0] = base;
                                               q.p[1] = base + al256 ((size_t)q.stride[0] * (h + 1)); q.p[2] = q.p[1] + al256 ((size_t)q.stride[1] * ((h >> 1) + 1)); return q; };
  const Pic hs = lay (c->h_buf, iSrcWidth, iSrcHeight), hd = lay (c->h_buf + src_b, iDstWidth, iDstHeight);
  const Pic ds = lay (c->d_buf, iSrcWidth, iSrcHeight), dd = lay (c->d_buf + src_b + 2 * tmp_b, iDstWidth, iDstHeight);
  for (int i = 0; i < 3; ++i) {
    const int w = i ? iSrcWidth >> 1 : iSrcWidth, h = i ? iSrcHeight >> 1 : iSrcHeight;
    for (int r = 0; r < h; ++r) memcpy (hs.p[i] + (size_t)r * hs.stride[i], pSrc[i] + (size_t)r * iSrcStride[i], (size_t)w);
    memcpy (hs.p[i] + (size_t)h * hs.stride[i], hs.p[i] + (size_t) (h - 1) * hs.stride[i], (size_t)w);        // (the spare row: never weighted, only addressed)
  }
  bool bad = hipMemcpyAsync (c->d_buf, c->h_buf, src_b, hipMemcpyHostToDevice, c->st) != hipSuccess;
  Pic cur = ds;
  int rc = WELSHIP_OK;
  for (size_t k = 0; k < stages.size() && rc == WELSHIP_OK; ++k) {
    const Stage& g = stages[k];
    const bool last = k + 1 == stages.size();
    const Pic out = last ? dd : lay (c->d_buf + src_b + (k & 1) * tmp_b, g.dw, g.dh);
    for (int i = 0; i < 3; ++i) { c->h_planes[3 * k + i].src = cur.p[i]; c->h_planes[3 * k + i].dst = out.p[i]; }
    cur = out;
  }
  bad = bad || hipMemcpyAsync (c->d_planes, c->h_planes, sizeof (DsPlane) * 3 * stages.size(), hipMemcpyHostToDevice, c->st) != hipSuccess;
  cur = ds;
  for (size_t k = 0; k < stages.size() && rc == WELSHIP_OK; ++k) {
    const Stage& g = stages[k];
    const bool last = k + 1 == stages.size();
    const Pic out = last ? dd : lay (c->d_buf + src_b + (k & 1) * tmp_b, g.dw, g.dh);
    if (g.mode >= 0) {
      // luma, then both chroma planes in one launch (same geometry)
      rc = launch (g.mode, c->d_planes + 3 * k, 1, cur.stride[0], g.sw, g.sh, out.stride[0], g.dw, g.dh, c->st);
      if (rc == WELSHIP_OK) rc = launch (g.mode, c->d_planes + 3 * k + 1, 2, cur.stride[1], g.sw >> 1, g.sh >> 1, out.stride[1], g.dw >> 1, g.dh >> 1, c->st);
    } else {
      // pfGeneralRatioLuma = GeneralBilinearFastDownsampler_c, pfGeneralRatioChroma = GeneralBilinearAccurateDownsampler_c (the C table)
      rc = launch (WELSHIP_DS_GENERAL_FAST, c->d_planes + 3 * k, 1, cur.stride[0], g.sw, g.sh, out.stride[0], g.dw, g.dh, c->st);
      if (rc == WELSHIP_OK) rc = launch (WELSHIP_DS_GENERAL_ACCURATE, c->d_planes + 3 * k + 1, 2, cur.stride[1], g.sw >> 1, g.sh >> 1, out.stride[1], g.dw >> 1, g.dh >> 1, c->st);
    }
    cur = out;
  }
  bad = bad || hipMemcpyAsync (c->h_buf + src_b, c->d_buf + src_b + 2 * tmp_b, dst_b, hipMemcpyDeviceToHost, c->st) != hipSuccess;
  if (hipStreamSynchronize (c->st) != hipSuccess || bad) { (void)hipGetLastError(); return WELSHIP_ERR_UNKNOWN; }
  if (rc != WELSHIP_OK) return rc;
  for (int i = 0; i < 3; ++i) {
    const int w = i ? iDstWidth >> 1 : iDstWidth, h = i ? iDstHeight >> 1 : iDstHeight;
    for (int r = 0; r < h; ++r) memcpy (pDst[i] + (size_t)r * iDstStride[i], hd.p[i] + (size_t)r * hd.stride[i], (size_t)w);
  }
  return WELSHIP_OK;
}

extern "C" {

// Throughput of one down-sampling launch over `nPlanes` resident planes (HIP events on the launch stream): pOut[0] = average
// milliseconds per launch, pOut[1] = algorithmic bytes per launch (every source sample the filter touches read once + every
// destination sample written once).
int WelsHipDownsampleBench (int iDevice, int mode, int nPlanes, int iSrcWidth, int iSrcHeight, int iDstWidth, int iDstHeight, int iIters, double* pOut) {
  int cnt = 0;
  if (hipGetDeviceCount (&cnt) != hipSuccess || cnt <= 0) return WELSHIP_ERR_NO_DEVICE;
  if (!pOut || nPlanes < 1 || iIters < 1 || iDevice < 0 || iDevice >= cnt) return WELSHIP_ERR_INIT_PARA;
  if (hipSetDevice (iDevice) != hipSuccess) return WELSHIP_ERR_NO_DEVICE;
  const int ss = (iSrcWidth + 63) & ~63, dsd = (iDstWidth + 63) & ~63;
  const size_t sb = (size_t)ss * (iSrcHeight + 1), db = (size_t)dsd * iDstHeight;
  uint8_t *src = nullptr, *dst = nullptr;
  DsPlane* dp = nullptr;
  int rc = WELSHIP_ERR_MEMORY;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMalloc ((void**)&src, sb * nPlanes) == hipSuccess && hipMalloc ((void**)&dst, db * nPlanes) == hipSuccess && hipMalloc ((void**)&dp, sizeof (DsPlane) * nPlanes) == hipSuccess &&
      hipStreamCreate (&st) == hipSuccess && hipEventCreate (&e0) == hipSuccess && hipEventCreate (&e1) == hipSuccess) {
    std::vector<DsPlane> h (nPlanes);
    for (int i = 0; i < nPlanes; ++i) { h[i].src = src + sb * i; h[i].dst = dst + db * i; }
    (void)hipMemcpy (dp, h.data(), sizeof (DsPlane) * nPlanes, hipMemcpyHostToDevice);
    (void)hipMemset (src, 0x55, sb * nPlanes);
    rc = launch (mode, dp, nPlanes, ss, iSrcWidth, iSrcHeight, dsd, iDstWidth, iDstHeight, st);      // warm-up
    (void)hipEventRecord (e0, st);
    for (int i = 0; i < iIters && rc == WELSHIP_OK; ++i) rc = launch (mode, dp, nPlanes, ss, iSrcWidth, iSrcHeight, dsd, iDstWidth, iDstHeight, st);
    (void)hipEventRecord (e1, st);
    if (hipStreamSynchronize (st) != hipSuccess) rc = WELSHIP_ERR_UNKNOWN;
    float ms = 0.f;
    (void)hipEventElapsedTime (&ms, e0, e1);
    pOut[0] = ms / iIters;
    // source samples touched: the 2x2 footprints (dyadic: all of them; 3:1, 4:1: a 2x2 of every 3x3 / 4x4; general: ~2x2 per output)
    double rd = mode == WELSHIP_DS_HALF ? (double)iDstWidth * 2 * iDstHeight * 2 : (double)iDstWidth * iDstHeight * 4;
    if ((mode == WELSHIP_DS_GENERAL_FAST || mode == WELSHIP_DS_GENERAL_ACCURATE) && rd > (double)iSrcWidth * iSrcHeight) rd = (double)iSrcWidth * iSrcHeight;
    pOut[1] = (rd + (double)iDstWidth * iDstHeight) * nPlanes;
  }
  if (e0) (void)hipEventDestroy (e0);
  if (e1) (void)hipEventDestroy (e1);
  if (st) (void)hipStreamDestroy (st);
  if (src) (void)hipFree (src);
  if (dst) (void)hipFree (dst);
  if (dp) (void)hipFree (dp);
  return rc;
}

}  // extern "C"
