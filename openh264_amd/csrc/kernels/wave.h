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
// A workgroup is ONE wavefront and a wavefront's LDS instructions execute in issue order, so the hand-off between
// lane blocks needs no s_waitcnt / s_barrier: only the compiler must not move LDS accesses across it.
#define WV_SYNC() do { __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                       __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define WV_LANES_END } WV_SYNC();

// 64-lane reductions on the DPP network (no LDS traffic): quad -> half row -> row, then the four row results are
// combined on the scalar unit.  Results are wave-uniform (SGPR).
#define WH_DPP(v, ctrl) __builtin_amdgcn_update_dpp (0, (v), (ctrl), 0xF, 0xF, true)
WH_FN int wh_wave_sum_i32 (int v) {
  v += WH_DPP (v, 0xB1);     // quad_perm [1,0,3,2]
  v += WH_DPP (v, 0x4E);     // quad_perm [2,3,0,1]
  v += WH_DPP (v, 0x141);    // row_half_mirror
  v += WH_DPP (v, 0x140);    // row_mirror
  return __builtin_amdgcn_readlane (v, 0) + __builtin_amdgcn_readlane (v, 16) + __builtin_amdgcn_readlane (v, 32) + __builtin_amdgcn_readlane (v, 48);
}
WH_FN int wh_wave_min_i32 (int v) {
  int o;
  o = WH_DPP (v, 0xB1); v = o < v ? o : v;
  o = WH_DPP (v, 0x4E); v = o < v ? o : v;
  o = WH_DPP (v, 0x141); v = o < v ? o : v;
  o = WH_DPP (v, 0x140); v = o < v ? o : v;
  const int a = __builtin_amdgcn_readlane (v, 0), b = __builtin_amdgcn_readlane (v, 16), c = __builtin_amdgcn_readlane (v, 32), d = __builtin_amdgcn_readlane (v, 48);
  const int ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
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

// ---- optional in-kernel phase profiling (WhSeqParams.prof != NULL): cycles since the previous mark are added
// to counter `id` (64 banks x 32 counters, banked by block id to keep the atomics uncontended) by lane 0.  Costs one uniform branch when disabled.
#if defined(WH_EMU)
#define WH_PROF_DECL(P) ((void)0)
#define WH_PROF_MARK(P, id) ((void)0)
#else
#define WH_PROF_DECL(P) unsigned long long _wh_t0 = (P).prof ? (unsigned long long)__builtin_readcyclecounter() : 0ULL
#define WH_PROF_MARK(P, id) do { if ((P).prof) { const unsigned long long _t = (unsigned long long)__builtin_readcyclecounter(); \
  if ((threadIdx.x & 63) == 0) { unsigned long long* _p = (P).prof + ((blockIdx.x + blockIdx.y * 7u) & 63u) * 32u; atomicAdd (&_p[id], _t - _wh_t0); atomicAdd (&_p[16 + (id)], 1ULL); } _wh_t0 = _t; } } while (0)
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
