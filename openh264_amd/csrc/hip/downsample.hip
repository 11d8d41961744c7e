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
#include <vector>
#include "../../../include/welship.h"

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
    for (int i = 0; i < 8 && x8 * 8 + i < dst_w; ++i) {
      const int t1 = (s[2 * i] + s[2 * i + 1] + 1) >> 1, t2 = (s[2 * i + src_stride] + s[2 * i + 1 + src_stride] + 1) >> 1;
      d[i] = (uint8_t) ((t1 + t2 + 1) >> 1);
    }
  }
}

// 3:1 / 4:1: the 2x2 average at every `step`-th sample, one lane = one output sample
__global__ __launch_bounds__ (256) void k_ds_step (const DsPlane* planes, int src_stride, int dst_stride, int dst_w, int dst_h, int step) {
  const DsPlane pl = planes[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dst_w) return;
  const uint8_t* s = pl.src + (size_t) (step * y) * src_stride + step * x;
  const int t1 = (s[0] + s[1] + 1) >> 1, t2 = (s[src_stride] + s[src_stride + 1] + 1) >> 1;
  pl.dst[(size_t)y * dst_stride + x] = (uint8_t) ((t1 + t2 + 1) >> 1);
}

// any ratio: `accurate` = GeneralBilinearAccurateDownsampler_c, else GeneralBilinearFastDownsampler_c; last column and last
// row are nearest-sample copies as in the reference
__global__ __launch_bounds__ (256) void k_ds_general (const DsPlane* planes, int src_stride, int dst_stride, int dst_w, int dst_h, int scalex, int scaley, int accurate) {
  const DsPlane pl = planes[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dst_w) return;
  const int bw = accurate ? 15 : 16, bh = 15;
  const int64_t xinv = ((int64_t)1 << (bw - 1)) + (int64_t)x * scalex, yinv = ((int64_t)1 << (bh - 1)) + (int64_t)y * scaley;
  const int xx = (int) (xinv >> bw), yy = (int) (yinv >> bh);
  const uint8_t* p = pl.src + (size_t)yy * src_stride + xx;
  uint8_t out;
  if (y == dst_h - 1 || x == dst_w - 1) out = p[0];
  else {
    const uint32_t fu = (uint32_t) (xinv & ((1 << bw) - 1)), fv = (uint32_t) (yinv & ((1 << bh) - 1));
    const uint32_t a = p[0], b = p[1], c = p[src_stride], d = p[src_stride + 1];
    if (accurate) {
      const int64_t k = 1 << 15;
      int64_t v = ((k - 1 - fu) * (k - 1 - fv) * a + (int64_t)fu * (k - 1 - fv) * b + (k - 1 - fu) * (int64_t)fv * c + (int64_t)fu * fv * d + ((int64_t)1 << 29)) >> 30;
      out = (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v);
    } else {
      const uint32_t sw = 1u << 16, sh = 1u << 15;
      uint32_t v = (((sw - 1 - fu) * (sh - 1 - fv)) >> 16) * a;
      v += ((fu * (sh - 1 - fv)) >> 16) * b;
      v += (((sw - 1 - fu) * fv) >> 16) * c;
      v += ((fu * fv) >> 16) * d;
      v >>= 14;
      v += 1;
      v >>= 1;
      out = (uint8_t) (v > 255 ? 255 : v);
    }
  }
  pl.dst[(size_t)y * dst_stride + x] = out;
}

inline int round_scale (int src, int dst, int bits) {        // WELS_ROUND ((float)src / (float)dst * (1 << bits))
  const float f = (float)src / (float)dst * (float) (1 << bits);
  return (int) (f + (f >= 0 ? 0.5f : -0.5f));
}

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
