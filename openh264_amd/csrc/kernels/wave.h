// wave.h -- the execution model the macroblock kernels are written in.
//
// One 64-lane wavefront owns one 16x16 macroblock.  Kernel bodies are written as a sequence of
//   * "lane blocks"   WV_LANES_BEGIN(lane) ... WV_LANES_END   -- per-lane (vector) work
//   * uniform code    between lane blocks                      -- wave-uniform scalars & branches
// All cross-lane traffic goes through the wave's LDS tile (struct passed as `S`), through the DPP reductions
// (WV_SUM, WV_SUM2, WV_SATD_ROWS, WV_ARGMIN) or through lane tables (WvLaneArr: a wave-uniform table in one VGPR).
// A lane block never keeps per-lane state alive across its END; that discipline is what lets the very same source be
// compiled
//   * by hipcc for gfx950 (lane = threadIdx.x & 63, END = a wavefront-scope fence: a wavefront's LDS instructions
//     execute in order, so no s_waitcnt / s_barrier is needed), and
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
// -DWH_EMU_REVERSE walks the lanes from 63 down: a lane block whose result depends on the order of its lanes (two lanes storing
// to the same LDS word) has no defined outcome on the GPU, and shows up as a difference between the two test builds
#if defined(WH_EMU_REVERSE)
#define WV_LANES_BEGIN(lane) for (int lane = 63; lane >= 0; --lane) {
#else
#define WV_LANES_BEGIN(lane) for (int lane = 0; lane < 64; ++lane) {
#endif
#define WV_LANES_END }
#define WH_UNIFORM_VALUE(x) ((void)0)
#define WV_SYNC() ((void)0)
#define WV_GLOBAL_FENCE() ((void)0)
#define WV_SUM(dst, lane, expr)                                   \
  do { int _s = 0; for (int lane = 0; lane < 64; ++lane) _s += (int)(expr); (dst) = _s; } while (0)
// minimum of (key) over lanes with `valid`, ties -> lowest lane; dst_lane = that lane or -1
#define WV_ARGMIN(dst_key, dst_lane, lane, valid, key)            \
  do { int _bk = 0x7fffffff, _bl = -1;                            \
       for (int lane = 0; lane < 64; ++lane) if (valid) { int _k = (int)(key); if (_k < _bk) { _bk = _k; _bl = lane; } } \
       (dst_key) = _bk; (dst_lane) = _bl; } while (0)
#define WV_ANY(dst, lane, expr)                                   \
  do { int _a = 0; for (int lane = 0; lane < 64; ++lane) _a |= ((expr) ? 1 : 0); (dst) = _a; } while (0)
// two sums in one pass
#define WV_SUM2(dst_a, dst_b, lane, expr_a, expr_b)               \
  do { int _sa = 0, _sb = 0; for (int lane = 0; lane < 64; ++lane) { _sa += (int)(expr_a); _sb += (int)(expr_b); } (dst_a) = _sa; (dst_b) = _sb; } while (0)
// quad sums as a lane table: every lane of quad q (lanes 4 q .. 4 q + 3) of `tab` holds the sum of expr over that quad -- read
// with WV_LGET (tab, 4 * q).  `expr` may have side effects on the LDS tile (it is evaluated once per lane, in lane order).
#define WV_QUADSUM_TAB(tab, lane, expr)                           \
  do { int _t[64]; for (int lane = 0; lane < 64; ++lane) _t[lane] = (int)(expr); \
       for (int _q = 0; _q < 16; ++_q) { const int _s = _t[4 * _q] + _t[4 * _q + 1] + _t[4 * _q + 2] + _t[4 * _q + 3]; \
         for (int _k = 0; _k < 4; ++_k) (tab).v[4 * _q + _k] = _s; } } while (0)
// pair sums as a lane table: both lanes of pair p (lanes 2 p, 2 p + 1) of `tab` hold the sum of expr over the pair (expr: evaluated once per lane, in lane order)
#define WV_PAIRSUM_TAB(tab, lane, expr)                           \
  do { int _t[64]; for (int lane = 0; lane < 64; ++lane) _t[lane] = (int)(expr); \
       for (int _p = 0; _p < 32; ++_p) { (tab).v[2 * _p] = (tab).v[2 * _p + 1] = _t[2 * _p] + _t[2 * _p + 1]; } } while (0)
// dst[lane] = src[idx (lane)] for every lane (idx in 0 .. 63): a lane table read at a per-lane index
#define WV_LSHUF(dst, src, lane, idx)                             \
  do { WvLaneArr _s = (src); for (int lane = 0; lane < 64; ++lane) (dst).v[lane] = _s.v[(idx) & 63]; } while (0)
// four sums at once: d_i = sum of expr over the lanes of quad i (4 i .. 4 i + 3), i = 0..3 -- sixteen lanes, one value each
#define WV_QUADSUM4(d0, d1, d2, d3, lane, expr)                   \
  do { int _q[4] = {0, 0, 0, 0}; for (int lane = 0; lane < 16; ++lane) _q[lane >> 2] += (int)(expr); \
       (d0) = _q[0]; (d1) = _q[1]; (d2) = _q[2]; (d3) = _q[3]; } while (0)
// four sums at once, one per DPP row: d_i = sum of expr over lanes 16 i .. 16 i + 15 -- several small reductions of a phase placed side by
// side on the lanes cost one pass over the DPP network (four steps + four reads) instead of one full wave sum each
#define WV_ROWSUM4(d0, d1, d2, d3, lane, expr)                    \
  do { int _q[4] = {0, 0, 0, 0}; for (int lane = 0; lane < 64; ++lane) _q[lane >> 4] += (int)(expr); \
       (d0) = _q[0]; (d1) = _q[1]; (d2) = _q[2]; (d3) = _q[3]; } while (0)
// SATD over 4x4 blocks held one per lane quad: lane 4q+r supplies row r of block q as four packed source bytes
// (enc4) and four packed prediction bytes (pred4); `active` must be uniform per quad.
// dst = sum over active blocks of (sum|Hadamard4x4(enc - pred)| + 1) >> 1   (sample.cpp:47-96 WelsSampleSatd4x4_c)
#define WV_SATD_ROWS(dst, lane, active, enc4, pred4)              \
  do { int _h[64][4]; bool _a[64];                                \
       for (int lane = 0; lane < 64; ++lane) { _a[lane] = (active); const uint32_t _e = (enc4), _p = (pred4); \
         const int _d0 = (int)(_e & 255) - (int)(_p & 255), _d1 = (int)((_e >> 8) & 255) - (int)((_p >> 8) & 255); \
         const int _d2 = (int)((_e >> 16) & 255) - (int)((_p >> 16) & 255), _d3 = (int)(_e >> 24) - (int)(_p >> 24); \
         const int _s0 = _d0 + _d2, _s1 = _d1 + _d3, _s2 = _d0 - _d2, _s3 = _d1 - _d3; \
         _h[lane][0] = _s0 + _s1; _h[lane][1] = _s2 + _s3; _h[lane][2] = _s2 - _s3; _h[lane][3] = _s0 - _s1; } \
       int _t = 0;                                                \
       for (int _q = 0; _q < 16; ++_q) if (_a[_q * 4]) { int _s = 0; \
         for (int _c = 0; _c < 4; ++_c) { const int _x0 = _h[_q * 4][_c], _x1 = _h[_q * 4 + 1][_c], _x2 = _h[_q * 4 + 2][_c], _x3 = _h[_q * 4 + 3][_c]; \
           const int _u0 = _x0 + _x2, _u1 = _x1 + _x3, _u2 = _x0 - _x2, _u3 = _x1 - _x3; \
           _s += abs (_u0 + _u1) + abs (_u2 + _u3) + abs (_u2 - _u3) + abs (_u0 - _u1); } \
         _t += (_s + 1) >> 1; }                                    \
       (dst) = _t; } while (0)
// the same with one sum per half of the wave (lanes 0..31 / 32..63): two candidates side by side
#define WV_SATD_ROWS_HALVES(dA, dB, lane, active, enc4, pred4)    \
  do { WV_SATD_ROWS (dA, lane, (lane < 32) && (active), enc4, pred4); WV_SATD_ROWS (dB, lane, (lane >= 32) && (active), enc4, pred4); } while (0)
// Variants for several reductions in a row whose per-lane operands overlap: WV_DECLARE_LANE names ONE lane id for all of
// them so that the compiler can share the common loads (each plain macro re-materialises its own opaque lane id).
#define WV_DECLARE_LANE(lane) ((void)0)
#define WV_SATD_ROWS_SHARED(dst, lane, active, enc4, pred4) WV_SATD_ROWS (dst, lane, active, enc4, pred4)
// four candidates against the same source samples in one pass (the device twin shares the reduction over the blocks)
#define WV_SATD_ROWS4_SHARED(d0, d1, d2, d3, lane, active, enc4, p0, p1, p2, p3) \
  do { WV_SATD_ROWS (d0, lane, active, enc4, p0); WV_SATD_ROWS (d1, lane, active, enc4, p1); WV_SATD_ROWS (d2, lane, active, enc4, p2); WV_SATD_ROWS (d3, lane, active, enc4, p3); } while (0)
// per-lane statements under the shared lane id (results kept in lane tables: WV_LOWN (tab, lane) = ...)
#define WV_LANE_EVAL(lane, ...) do { for (int lane = 0; lane < 64; ++lane) { __VA_ARGS__; } } while (0)
// pointers into device global memory (explicit address space on the GPU so that loads are global_*, not flat_*)
#define WH_G
#include <string.h>
#include <stdlib.h>
// A small wave-uniform table kept in ONE vector register: lane i holds entry i.  Uniform code reads entry i with
// v_readlane (a few cycles) instead of an LDS round trip (~100 cycles); lane-parallel code updates many entries at once.
typedef struct WvLaneArr { int v[64]; } WvLaneArr;
#define WV_LGET(a, i) ((a).v[i])
#define WV_LSET(a, i, val) ((a).v[i] = (val))
#define WV_LOWN(a, lane) ((a).v[lane])                      // inside a lane block: this lane's own entry
WH_FN void wh_atomic_add_u32 (uint32_t* p, uint32_t v) { *p += v; }     // device-scope atomic on the GPU
#define WV_LSET_IF(a, lane, cond, val) do { for (int lane = 0; lane < 64; ++lane) if (cond) (a).v[lane] = (val); } while (0)
// asynchronous copy of one 4-byte word per lane from global memory to LDS word `lane` of `lds_base` (GPU: LDS-DMA,
// no register holds the data; complete after WV_ASYNC_WAIT)
WH_FN void wh_ld_async4 (const void* src, uint32_t* lds_base, int lane) { memcpy (&lds_base[lane], src, 4); }
// same with 16 bytes per lane (lds_base 16-byte aligned, lane i fills bytes [16 i, 16 i + 16))
WH_FN void wh_ld_async16 (const void* src, void* lds_base, int lane) { memcpy ((uint8_t*)lds_base + 16 * lane, src, 16); }
#define WV_ASYNC_WAIT() ((void)0)
// One word to / from global memory that ANOTHER workgroup (any XCD) produces / consumes inside the same launch: on the GPU
// a write-through store and a cache-bypassing load (sc0 sc1) -- see the device twins below.
WH_FN void wh_st_xwg32 (uint32_t* p, uint32_t v) { *p = v; }
WH_FN uint32_t wh_ld_xwg32 (const uint32_t* p) { return *p; }
// One word in global memory that another wavefront of the SAME workgroup wrote earlier in this launch (ordered by the
// scheduler's done flags): a vector-memory access on the GPU, never a scalar load (the scalar cache is not coherent with stores)
WH_FN void wh_st_wg32 (uint32_t* p, uint32_t v) { *p = v; }
WH_FN uint32_t wh_ld_wg32 (const uint32_t* p) { return *p; }
// A slice coded by SEVERAL workgroups (hip_backend.hip k_inter_split; X = true): whatever a neighbour macroblock reads back -- the macroblock's state and its
// unfiltered samples -- is stored write-through and loaded past the caches, as wh_st_xwg32 / wh_ld_xwg32 do for the deblocking bands; X = false: plain accesses
template <bool X, class T> WH_FN void wh_st_x (T* p, T v) { *p = v; }
template <bool X> WH_FN uint32_t wh_ld_x32 (const uint32_t* p) { return *p; }
template <bool X> WH_FN void wh_ld_async4_x (const void* src, uint32_t* lds_base, int lane) { memcpy (&lds_base[lane], src, 4); }
// four bytes at any byte offset of a 4-byte aligned LDS array
WH_FN uint32_t wh_ld4u (const uint8_t* base, int off) { uint32_t v; memcpy (&v, base + off, 4); return v; }
// bytes k .. k + 3 (k = 0 .. 3) of the eight bytes lo | hi << 32
WH_FN uint32_t wh_funnel4 (uint32_t lo, uint32_t hi, int k) { return k ? (lo >> (8 * k)) | (hi << (32 - 8 * k)) : lo; }
// sum of absolute differences of four packed bytes
WH_FN int wh_sad4 (uint32_t a, uint32_t b) {
  int s = 0;
  for (int k = 0; k < 4; ++k) { const int d = (int) ((a >> (8 * k)) & 255) - (int) ((b >> (8 * k)) & 255); s += d < 0 ? -d : d; }
  return s;
}
// per-byte (a + b + 1) >> 1
WH_FN uint32_t wh_avg4 (uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int k = 0; k < 4; ++k) r |= ((((a >> (8 * k)) & 255) + ((b >> (8 * k)) & 255) + 1) >> 1) << (8 * k);
  return r;
}

#else
// -------------------------------------------------------------------------------- gfx950 device
#include <hip/hip_runtime.h>
#define WH_FN static __device__ __forceinline__
#define WH_HDFN static __host__ __device__ __forceinline__
#define WH_CONST static __device__ const
// The lane id is re-materialised as an opaque value in every lane block: otherwise the compiler hoists all per-lane
// address arithmetic out of the per-macroblock loop, keeps it live across the whole body and spills it to scratch.
WH_FN int wh_lane_id() { int l = (int)(threadIdx.x & 63); asm volatile ("" : "+v"(l)); return l; }
// a wave-uniform value the compiler must take as it is (kept in a scalar register): a select between two LOADED uniforms is otherwise
// folded into one vector load at a selected address
#define WH_UNIFORM_VALUE(x) asm volatile ("" : "+s"(x))
#define WV_LANES_BEGIN(lane) { const int lane = wh_lane_id();
// A workgroup is ONE wavefront and a wavefront's LDS instructions execute in issue order, so the hand-off between
// lane blocks needs no s_waitcnt / s_barrier: only the compiler must not move LDS accesses across it.
#define WV_SYNC() do { __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                       __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define WV_LANES_END } WV_SYNC();
// this wave's global-memory stores are complete and visible to the workgroup (and to its own later loads) before anything after it
#define WV_GLOBAL_FENCE() __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "workgroup")

// 64-lane reductions on the DPP network (no LDS traffic): quad -> half row -> row, then the four row results are
// combined on the scalar unit.  Results are wave-uniform (SGPR).
#define WH_DPP(v, ctrl) __builtin_amdgcn_update_dpp (0, (v), (ctrl), 0xF, 0xF, true)
// The four row sums are combined on the DPP network too (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: gfx9's
// whole-wave reduction), so the total is ONE v_readlane of lane 63 instead of four reads and three scalar additions -- a macroblock
// makes some eighty of these reductions.
#define WH_DPP_ROWS(v, ctrl, rows) __builtin_amdgcn_update_dpp (0, (v), (ctrl), (rows), 0xF, false)
WH_FN int wh_wave_sum_i32 (int v) {
  v += WH_DPP (v, 0xB1);     // quad_perm [1,0,3,2]
  v += WH_DPP (v, 0x4E);     // quad_perm [2,3,0,1]
  v += WH_DPP (v, 0x141);    // row_half_mirror
  v += WH_DPP (v, 0x140);    // row_mirror
  v += WH_DPP_ROWS (v, 0x142, 0xA);   // row_bcast:15 -> rows 1, 3
  v += WH_DPP_ROWS (v, 0x143, 0xC);   // row_bcast:31 -> rows 2, 3
  return __builtin_amdgcn_readlane (v, 63);
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
  do { const int lane = wh_lane_id(); int _v = (int)(expr); (dst) = wh_wave_sum_i32 (_v); } while (0)
#define WV_ARGMIN(dst_key, dst_lane, lane, valid, key)            \
  do { const int lane = (int)(threadIdx.x & 63);                  \
       const bool _ok = (valid);                                  \
       int _k = _ok ? (int)(key) : 0x7fffffff;                    \
       const int _m = wh_wave_min_i32 (_k);                       \
       const unsigned long long _b = __ballot (_ok && _k == _m);  \
       (dst_key) = _m; (dst_lane) = _b ? (int)__builtin_ctzll (_b) : -1; } while (0)
#define WV_ANY(dst, lane, expr)                                   \
  do { const int lane = (int)(threadIdx.x & 63); (dst) = __ballot ((expr)) != 0ULL; } while (0)
#define WV_SUM2(dst_a, dst_b, lane, expr_a, expr_b)               \
  do { const int lane = wh_lane_id(); int _va = (int)(expr_a), _vb = (int)(expr_b); \
       _va += WH_DPP (_va, 0xB1); _vb += WH_DPP (_vb, 0xB1); _va += WH_DPP (_va, 0x4E); _vb += WH_DPP (_vb, 0x4E); \
       _va += WH_DPP (_va, 0x141); _vb += WH_DPP (_vb, 0x141); _va += WH_DPP (_va, 0x140); _vb += WH_DPP (_vb, 0x140); \
       _va += WH_DPP_ROWS (_va, 0x142, 0xA); _vb += WH_DPP_ROWS (_vb, 0x142, 0xA); _va += WH_DPP_ROWS (_va, 0x143, 0xC); _vb += WH_DPP_ROWS (_vb, 0x143, 0xC); \
       (dst_a) = __builtin_amdgcn_readlane (_va, 63); (dst_b) = __builtin_amdgcn_readlane (_vb, 63); } while (0)
#define WV_QUADSUM_TAB(tab, lane, expr)                           \
  do { const int lane = wh_lane_id(); int _v = (int)(expr);       \
       _v += WH_DPP (_v, 0xB1); _v += WH_DPP (_v, 0x4E); (tab) = _v; } while (0)
#define WV_PAIRSUM_TAB(tab, lane, expr)                           \
  do { const int lane = wh_lane_id(); int _v = (int)(expr);       \
       _v += WH_DPP (_v, 0xB1); (tab) = _v; } while (0)
#define WV_LSHUF(dst, src, lane, idx)                             \
  do { const int lane = wh_lane_id(); (dst) = __builtin_amdgcn_ds_bpermute ((int)(idx) << 2, (src)); } while (0)
#define WV_QUADSUM4(d0, d1, d2, d3, lane, expr)                   \
  do { const int lane = wh_lane_id(); int _v = (int)(expr);       \
       _v += WH_DPP (_v, 0xB1); _v += WH_DPP (_v, 0x4E);          \
       (d0) = __builtin_amdgcn_readlane (_v, 0); (d1) = __builtin_amdgcn_readlane (_v, 4); (d2) = __builtin_amdgcn_readlane (_v, 8); (d3) = __builtin_amdgcn_readlane (_v, 12); } while (0)
#define WV_ROWSUM4(d0, d1, d2, d3, lane, expr)                    \
  do { const int lane = wh_lane_id(); int _v = (int)(expr);       \
       _v += WH_DPP (_v, 0xB1); _v += WH_DPP (_v, 0x4E); _v += WH_DPP (_v, 0x141); _v += WH_DPP (_v, 0x140); \
       (d0) = __builtin_amdgcn_readlane (_v, 0); (d1) = __builtin_amdgcn_readlane (_v, 16); (d2) = __builtin_amdgcn_readlane (_v, 32); (d3) = __builtin_amdgcn_readlane (_v, 48); } while (0)
// SATD over lane quads, entirely in registers: horizontal Hadamard inside the lane, vertical butterflies across the
// quad on DPP quad_perm, then the usual wave sum (see the WH_EMU twin above for the contract).
// Two 16-bit values per register (v_pk_*_i16): every intermediate of a 4x4 Hadamard of byte differences fits (|x| <= 16 * 255).
typedef short wh_s16x2 __attribute__ ((ext_vector_type (2)));
WH_FN wh_s16x2 wh_as_s16x2 (uint32_t v) { return __builtin_bit_cast (wh_s16x2, v); }
WH_FN uint32_t wh_as_u32 (wh_s16x2 v) { return __builtin_bit_cast (uint32_t, v); }
// (x + y, x - y) of a pair: one v_pk_mad_i16 with the halves picked by op_sel
WH_FN wh_s16x2 wh_pk_sumdiff (wh_s16x2 v) { const wh_s16x2 pm = {1, -1}; return v.yy * pm + v.xx; }
// The sum of |Hadamard4x4 (e - p)| of this lane's quad in every lane of the quad (round 6: packed 16-bit arithmetic -- 25 vector
// instructions where the 32-bit version took 55; a P16x16 macroblock makes nine or ten of these passes in its refinement alone).
// Bytes 0,1 and 2,3 go to the 16-bit halves of two registers (v_perm_b32); the horizontal butterflies are two packed add / sub and two
// (x + y, x - y) steps; a vertical butterfly stage is partner + sign * own, sign = -1 in the upper lane of the pair: one DPP move and
// one v_pk_mad_i16 per register; |.| summed by v_sad_u16 against the bias that makes the halves unsigned.
WH_FN int wh_satd_quad (int lane, uint32_t e, uint32_t p) {
  const wh_s16x2 d01 = wh_as_s16x2 (__builtin_amdgcn_perm (0u, e, 0x0c010c00u)) - wh_as_s16x2 (__builtin_amdgcn_perm (0u, p, 0x0c010c00u));
  const wh_s16x2 d23 = wh_as_s16x2 (__builtin_amdgcn_perm (0u, e, 0x0c030c02u)) - wh_as_s16x2 (__builtin_amdgcn_perm (0u, p, 0x0c030c02u));
  wh_s16x2 a03 = wh_pk_sumdiff (d01 + d23);            // (s0 + s1, s0 - s1)
  wh_s16x2 a12 = wh_pk_sumdiff (d01 - d23);            // (s2 + s3, s2 - s3)
  const wh_s16x2 sg2 = wh_as_s16x2 ((lane & 2) ? 0xffffffffu : 0x00010001u), sg1 = wh_as_s16x2 ((lane & 1) ? 0xffffffffu : 0x00010001u);
  a03 = a03 * sg2 + wh_as_s16x2 ((uint32_t)WH_DPP ((int)wh_as_u32 (a03), 0x4E)); a12 = a12 * sg2 + wh_as_s16x2 ((uint32_t)WH_DPP ((int)wh_as_u32 (a12), 0x4E));
  a03 = a03 * sg1 + wh_as_s16x2 ((uint32_t)WH_DPP ((int)wh_as_u32 (a03), 0xB1)); a12 = a12 * sg1 + wh_as_s16x2 ((uint32_t)WH_DPP ((int)wh_as_u32 (a12), 0xB1));
  int s = (int)__builtin_amdgcn_sad_u16 (wh_as_u32 (a03) ^ 0x80008000u, 0x80008000u, 0u);
  s = (int)__builtin_amdgcn_sad_u16 (wh_as_u32 (a12) ^ 0x80008000u, 0x80008000u, (uint32_t)s);
  s += WH_DPP (s, 0xB1);
  s += WH_DPP (s, 0x4E);
  return s;
}
// (Measured in round 6 and not adopted, profiles/r06_ab_satd_reductions_nt_records.txt: the four candidates of a refinement stage reduced in ONE pass -- lane k of
//  every quad keeps candidate k's block sum, two row rotations, the four rows by v_permlane16_swap / v_permlane32_swap, 15 instructions where four separate
//  reductions take 40 -- together with a sum over the row's blocks by row_ror instead of the masked quad stages: MD launch 7.13 -> 7.20 ms, same box, alternating.)
WH_FN int wh_satd_rows (int lane, bool active, uint32_t e, uint32_t p) {
  int s = wh_satd_quad (lane, e, p);
  s = (active && (lane & 3) == 0) ? (s + 1) >> 1 : 0;
  return wh_wave_sum_i32 (s);
}
#define WV_SATD_ROWS(dst, lane, active, enc4, pred4)              \
  do { const int lane = wh_lane_id(); (dst) = wh_satd_rows (lane, (active), (enc4), (pred4)); } while (0)
#define WV_SATD_ROWS_HALVES(dA, dB, lane, active, enc4, pred4)    \
  do { const int lane = wh_lane_id(); int _v = wh_satd_quad (lane, (enc4), (pred4)); _v = (active) ? (_v + 1) >> 1 : 0;     /* every lane of a quad holds its block's sum */ \
       _v += WH_DPP (_v, 0x141); _v += WH_DPP (_v, 0x140);         \
       (dA) = __builtin_amdgcn_readlane (_v, 0) + __builtin_amdgcn_readlane (_v, 16); (dB) = __builtin_amdgcn_readlane (_v, 32) + __builtin_amdgcn_readlane (_v, 48); } while (0)
#define WV_DECLARE_LANE(lane) const int lane = wh_lane_id()
#define WV_SATD_ROWS_SHARED(dst, lane, active, enc4, pred4) do { (dst) = wh_satd_rows (lane, (active), (enc4), (pred4)); } while (0)
#define WV_SATD_ROWS4_SHARED(d0, d1, d2, d3, lane, active, enc4, p0, p1, p2, p3) \
  do { (d0) = wh_satd_rows (lane, (active), (enc4), (p0)); (d1) = wh_satd_rows (lane, (active), (enc4), (p1)); (d2) = wh_satd_rows (lane, (active), (enc4), (p2)); (d3) = wh_satd_rows (lane, (active), (enc4), (p3)); } while (0)
#define WV_LANE_EVAL(lane, ...) do { __VA_ARGS__; } while (0)
typedef int WvLaneArr;
#define WV_LGET(a, i) __builtin_amdgcn_readlane ((a), (i))
#define WV_LSET(a, i, val) do { if ((int)(threadIdx.x & 63) == (i)) (a) = (val); } while (0)
#define WV_LOWN(a, lane) (a)
#define WV_LSET_IF(a, lane, cond, val) do { const int lane = wh_lane_id(); if (cond) (a) = (val); } while (0)
#define WH_G __attribute__ ((address_space (1)))
WH_FN void wh_atomic_add_u32 (WH_G uint32_t* p, uint32_t v) { __hip_atomic_fetch_add (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WH_FN uint32_t wh_ld4u (const uint8_t* base, int off) {
  const uint32_t* w = (const uint32_t*)base + (off >> 2);
  return __builtin_amdgcn_alignbyte (w[1], w[0], (uint32_t)off & 3u);
}
WH_FN void wh_ld_async4 (const WH_G void* src, uint32_t* lds_base, int /*lane*/) {
  __builtin_amdgcn_global_load_lds ((const WH_G uint32_t*)src, (__attribute__ ((address_space (3))) uint32_t*)lds_base, 4, 0, 0);
}
WH_FN void wh_ld_async16 (const WH_G void* src, void* lds_base, int /*lane*/) {
  __builtin_amdgcn_global_load_lds ((const WH_G uint32_t*)src, (__attribute__ ((address_space (3))) uint32_t*)lds_base, 16, 0, 0);
}
#define WV_ASYNC_WAIT() asm volatile ("s_waitcnt vmcnt(0)" ::: "memory")
// Cross-workgroup payload inside one launch (deblocking bands): `global_store_dword ... sc0 sc1` writes through the XCD's
// L2 and `global_load_dword ... sc0 sc1` bypasses L1 and any stale L2 line, so producer and consumer need no agent-scope
// release (buffer_wbl2: writes back EVERY dirty line of the XCD's L2) / acquire (buffer_inv: empties the CU's L1 for all
// its waves) around the hand-off -- only the producer's s_waitcnt vmcnt(0) before its flag store
// (MI355X_MICROARCH.md, inter-workgroup visibility: "sc0 sc1 stores and loads both sides").
WH_FN void wh_st_xwg32 (WH_G uint32_t* p, uint32_t v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
WH_FN uint32_t wh_ld_xwg32 (const WH_G uint32_t* p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
WH_FN void wh_st_wg32 (WH_G uint32_t* p, uint32_t v) { __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
WH_FN uint32_t wh_ld_wg32 (const WH_G uint32_t* p) { return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <bool X, class T> WH_FN void wh_st_x (WH_G T* p, T v) { if (X) __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else *p = v; }
template <bool X> WH_FN uint32_t wh_ld_x32 (const WH_G uint32_t* p) { return X ? __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : *p; }
template <bool X> WH_FN void wh_ld_async4_x (const WH_G void* src, uint32_t* lds_base, int /*lane*/) {
  if (X) __builtin_amdgcn_global_load_lds ((const WH_G uint32_t*)src, (__attribute__ ((address_space (3))) uint32_t*)lds_base, 4, 0, 17);      // sc0 sc1
  else __builtin_amdgcn_global_load_lds ((const WH_G uint32_t*)src, (__attribute__ ((address_space (3))) uint32_t*)lds_base, 4, 0, 0);
}
WH_FN uint32_t wh_funnel4 (uint32_t lo, uint32_t hi, int k) { return __builtin_amdgcn_alignbyte (hi, lo, (uint32_t)k); }
WH_FN int wh_sad4 (uint32_t a, uint32_t b) { return (int)__builtin_amdgcn_sad_u8 (a, b, 0u); }
WH_FN uint32_t wh_avg4 (uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp (a, b, 0x01010101u); }
#endif

// ---- optional in-kernel phase profiling (WhSeqParams.prof != NULL) ----------------------------------------------
// Cycles since the previous mark are accumulated per wave in LDS (`L`.prof[id], count in prof[16 + id]) by lane 0 and
// flushed to the global counters once, when the wave leaves the kernel -- no memory traffic inside the MB loop, so the
// measurement does not perturb the s_waitcnt's it is measuring.  Costs one uniform branch when disabled.
// The accumulators cost every wave 128 bytes of LDS and every mark a branch, so they are compiled in only with -DWH_PROF: the library the
// profiling tools load (openh264_amd/libwelship_prof.so, tools/phase_profile.py builds it), never the product library.
#if defined(WH_EMU) || !defined(WH_PROF)
#define WH_PROF_DECL(P) ((void)0)
#define WH_PROF_MARK(P, L, id) ((void)0)
#define WH_PROF_SUB(P, L, id) ((void)0)
#else
#define WH_PROF_DECL(P) unsigned long long _wh_t0 = (P).prof ? (unsigned long long)__builtin_readcyclecounter() : 0ULL
#define WH_PROF_MARK_RAW(P, L, id) do { if ((P).prof) { const unsigned long long _t = (unsigned long long)__builtin_readcyclecounter(); \
  if ((threadIdx.x & 63) == 0) { (L).prof[id] += (uint32_t) (_t - _wh_t0); (L).prof[16 + (id)] += 1u; } _wh_t0 = _t; } } while (0)
#if defined(WH_PROF_DETAIL)
// Detail build of the P kernel's profile (tools/phase_profile.py --detail): the search .. store phases (2..7) are merged into id 7 and
// 0 / 8 / 10 into id 10, which frees ids for sub-phases of the claim, the neighbour loads and the P_Skip test (WH_PROF_SUB)
#define WH_PROF_MARK(P, L, id) WH_PROF_MARK_RAW (P, L, ((id) >= 2 && (id) <= 7) ? 7 : ((id) == 0 || (id) == 8 || (id) == 10) ? 10 : (id))
#define WH_PROF_SUB(P, L, id) WH_PROF_MARK_RAW (P, L, id)
#else
#define WH_PROF_MARK(P, L, id) WH_PROF_MARK_RAW (P, L, id)
#define WH_PROF_SUB(P, L, id) ((void)0)
#endif
#endif

// 16 bytes moved as one unit (global_load_dwordx4 / ds_read_b128)
typedef struct alignas (16) WhU4 { uint32_t x, y, z, w; } WhU4;
// ... to / from 16-byte aligned device memory
#if defined(WH_EMU)
WH_FN WhU4 wh_ldg16 (const void* p) { WhU4 v; memcpy (&v, p, 16); return v; }
WH_FN void wh_stg16 (void* p, WhU4 v) { memcpy (p, &v, 16); }
typedef WhU4 WhV4;                   // 16 bytes held in a lane's registers across lane blocks
WH_FN WhV4 wh_ldg16v (const void* p) { return wh_ldg16 (p); }
#else
typedef uint32_t wh_u32x4_t __attribute__ ((ext_vector_type (4)));
typedef wh_u32x4_t WhV4;             // (a vector VALUE: a struct variable that lives across a fence is kept in scratch memory)
WH_FN WhV4 wh_ldg16v (const WH_G void* p) { return * (const WH_G wh_u32x4_t*)p; }
WH_HDFN WhU4 wh_ldg16 (const WH_G void* p) { const wh_u32x4_t t = * (const WH_G wh_u32x4_t*)p; WhU4 v; v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w; return v; }
WH_HDFN void wh_stg16 (WH_G void* p, WhU4 v) { wh_u32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; * (WH_G wh_u32x4_t*)p = t; }
#endif

// 16 / 8 bytes to device memory that is only 4-byte aligned (a row piece of a picture at any multiple of four samples): still ONE
// global_store_dwordx4 / _dwordx2 per lane
#if defined(WH_EMU)
WH_FN void wh_stg16_a4 (void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { const uint32_t v[4] = {a, b, c, d}; memcpy (p, v, 16); }
WH_FN void wh_stg8_a4 (void* p, uint32_t a, uint32_t b) { const uint32_t v[2] = {a, b}; memcpy (p, v, 8); }
WH_FN void wh_stg12_a4 (void* p, uint32_t a, uint32_t b, uint32_t c) { const uint32_t v[3] = {a, b, c}; memcpy (p, v, 12); }
#else
typedef uint32_t wh_u32x3_a4_t __attribute__ ((ext_vector_type (3), aligned (4)));
WH_FN void wh_stg12_a4 (WH_G void* p, uint32_t a, uint32_t b, uint32_t c) { const wh_u32x3_a4_t v = {a, b, c}; * (WH_G wh_u32x3_a4_t*)p = v; }
typedef uint32_t wh_u32x4_a4_t __attribute__ ((ext_vector_type (4), aligned (4)));
typedef uint32_t wh_u32x2_a4_t __attribute__ ((ext_vector_type (2), aligned (4)));
WH_FN void wh_stg16_a4 (WH_G void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { const wh_u32x4_a4_t v = {a, b, c, d}; * (WH_G wh_u32x4_a4_t*)p = v; }
WH_FN void wh_stg8_a4 (WH_G void* p, uint32_t a, uint32_t b) { const wh_u32x2_a4_t v = {a, b}; * (WH_G wh_u32x2_a4_t*)p = v; }
#endif

// a * b for 0 <= a, b < 2^24 whose product fits 32 bits: v_mul_u32_u24 issues at the full rate, the general 32-bit multiply
// (v_mul_lo_u32) at a quarter of it -- the quantiser multiplies four coefficients per lane and pass
#if defined(WH_EMU)
WH_FN uint32_t wh_mul_u24 (uint32_t a, uint32_t b) { return a * b; }
#else
WH_FN uint32_t wh_mul_u24 (uint32_t a, uint32_t b) { return __umul24 (a, b); }
#endif

// ---- small integer helpers (host + device) -----------------------------------------------------
WH_FN int wh_abs (int a) { return a < 0 ? -a : a; }
WH_FN int wh_min (int a, int b) { return a < b ? a : b; }
WH_FN int wh_max (int a, int b) { return a > b ? a : b; }
WH_FN int wh_clip3 (int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
#if defined(WH_EMU)
WH_FN uint8_t wh_clip255 (int v) { return (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : v)); }
#else
// The clamp is an explicit v_med3_i32: hipcc (ROCm 7.2) would otherwise fuse "clip255 (x >> n)" pairs into gfx950's
// v_ashr_pk_u8_i32 and OR further bytes into the upper half of its result as if that half were zero -- on the MI355X it
// is not, which corrupts packed pixels (first seen in tests/test_prims_gpu.py::test_motion_compensation).
WH_FN uint8_t wh_clip255 (int v) { int r; asm ("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(255)); return (uint8_t)r; }
#endif
// |a - b| of two samples (both 0 .. 255): one v_sad_u8 on the GPU (the bytes above the samples are zero and add nothing) instead of the
// subtract / negate / maximum the compiler makes of wh_abs (a - b)
#if defined(WH_EMU)
WH_FN int wh_absdiff_px (int a, int b) { return a < b ? b - a : a - b; }
#else
WH_FN int wh_absdiff_px (int a, int b) { return (int)__builtin_amdgcn_sad_u8 ((uint32_t)a, (uint32_t)b, 0u); }
#endif
WH_FN int wh_median3 (int a, int b, int c) {
  int mn = wh_min (a, wh_min (b, c)), mx = wh_max (a, wh_max (b, c));
  return a + b + c - mn - mx;
}
