// wave.h -- the execution model the macroblock kernels are written in.
//
// One 64-lane wavefront owns one 16x16 macroblock.  Kernel bodies are written as a sequence of
//   * "lane blocks"   WV_LANES_BEGIN(lane) ... WV_LANES_END   -- per-lane (vector) work
//   * uniform code    between lane blocks                      -- wave-uniform scalars & branches
// All cross-lane traffic goes through the wave's LDS tile (struct passed as `S`) or through the
// WV_SUM / WV_MIN reductions.  A lane block never keeps per-lane state alive across its END; that
// discipline is what lets the very same source be compiled
//   * by hipcc for gfx950 (lane = threadIdx.x, END = workgroup barrier on a 1-wave workgroup,
//     reductions = DPP/ds_swizzle butterflies), and
//   * by g++ with -DWH_EMU for the CPU-side test build (lane = loop variable), which the
//     `-m "not gpu"` tests use to exercise the host entropy coder without a GPU.  The emulation
//     build is test infrastructure: the product library never contains or calls it.
#pragma once
#include <stdint.h>

#if defined(WH_EMU)
// -------------------------------------------------------------------------------- CPU emulation
#define WH_FN static inline
#define WH_HDFN static inline
#define WH_CONST static const
#define WV_LANES_BEGIN(lane) for (int lane = 0; lane < 64; ++lane) {
#define WV_LANES_END }
#define WV_SYNC() ((void)0)
#define WV_SUM(dst, lane, expr)                                   \
  do { int _s = 0; for (int lane = 0; lane < 64; ++lane) _s += (int)(expr); (dst) = _s; } while (0)
// minimum of (key) over lanes with `valid`, ties -> lowest lane; dst_lane = that lane or -1
#define WV_ARGMIN(dst_key, dst_lane, lane, valid, key)            \
  do { int _bk = 0x7fffffff, _bl = -1;                            \
       for (int lane = 0; lane < 64; ++lane) if (valid) { int _k = (int)(key); if (_k < _bk) { _bk = _k; _bl = lane; } } \
       (dst_key) = _bk; (dst_lane) = _bl; } while (0)
#define WV_ANY(dst, lane, expr)                                   \
  do { int _a = 0; for (int lane = 0; lane < 64; ++lane) _a |= ((expr) ? 1 : 0); (dst) = _a; } while (0)

#else
// -------------------------------------------------------------------------------- gfx950 device
#include <hip/hip_runtime.h>
#define WH_FN static __device__ __forceinline__
#define WH_HDFN static __host__ __device__ __forceinline__
#define WH_CONST static __device__ const
#define WV_LANES_BEGIN(lane) { const int lane = (int)(threadIdx.x & 63);
#define WV_LANES_END } __syncthreads();
#define WV_SYNC() __syncthreads()

WH_FN int wh_wave_sum_i32 (int v) {
  // 64-lane butterfly; result made uniform (SGPR) with readfirstlane
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor (v, m, 64);
  return __builtin_amdgcn_readfirstlane (v);
}
WH_FN int wh_wave_min_i32 (int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor (v, m, 64); v = o < v ? o : v; }
  return __builtin_amdgcn_readfirstlane (v);
}
#define WV_SUM(dst, lane, expr)                                   \
  do { const int lane = (int)(threadIdx.x & 63); int _v = (int)(expr); (dst) = wh_wave_sum_i32 (_v); } while (0)
#define WV_ARGMIN(dst_key, dst_lane, lane, valid, key)            \
  do { const int lane = (int)(threadIdx.x & 63);                  \
       const bool _ok = (valid);                                  \
       int _k = _ok ? (int)(key) : 0x7fffffff;                    \
       const int _m = wh_wave_min_i32 (_k);                       \
       const unsigned long long _b = __ballot (_ok && _k == _m);  \
       (dst_key) = _m; (dst_lane) = _b ? (int)__builtin_ctzll (_b) : -1; } while (0)
#define WV_ANY(dst, lane, expr)                                   \
  do { const int lane = (int)(threadIdx.x & 63); (dst) = __ballot ((expr)) != 0ULL; } while (0)
#endif

// ---- small integer helpers (host + device) -----------------------------------------------------
WH_FN int wh_abs (int a) { return a < 0 ? -a : a; }
WH_FN int wh_min (int a, int b) { return a < b ? a : b; }
WH_FN int wh_max (int a, int b) { return a > b ? a : b; }
WH_FN int wh_clip3 (int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
WH_FN uint8_t wh_clip255 (int v) { return (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : v)); }
WH_FN int wh_median3 (int a, int b, int c) {
  int mn = wh_min (a, wh_min (b, c)), mx = wh_max (a, wh_max (b, c));
  return a + b + c - mn - mx;
}
