// leaf.hip -- layer (3c) of the C ABI (include/welship_leaf.h): the leaf primitives with exactly the reference's function-pointer
// signatures, one export per slot of SWelsFuncPtrList / SSampleDealingFunc / SMcFunc / DeblockingFunc
// (codec/encoder/core/inc/wels_func_ptr_def.h:58-188, codec/common/inc/mc.h:40-53).  A call stages the few hundred bytes the
// reference function would touch through one small HBM arena (one host-to-device copy, one wavefront-sized launch, one copy back) and
// runs the device code of the fused macroblock kernels on them (prims_kernels.h and the kernels/ headers) -- nothing is computed on
// the host.  The typedefs have no error channel: without a usable device, or when a launch fails, the call reports on stderr and
// aborts.  Tens of microseconds per call: integration bring-up and parity checking, not the throughput path (welship.h, frame level).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "../../../include/welship.h"
#include "../../../include/welship_leaf.h"
#include "../kernels/frame_kernels.h"
#include "../kernels/inter_mb.h"
#include "../kernels/mc_px.h"
#include "../kernels/deblock_mb.h"

// This file is the second half of prims.hip's translation unit (prims.hip includes it at its end): both layers launch the kernels of
// prims_kernels.h, which are therefore compiled ONCE (rounds 4-5: two translation units, every one of those kernels twice in the code object).
#ifndef WH_PRIMS_TU
#error "leaf.hip is compiled as part of prims.hip"
#endif
namespace {

[[noreturn]] void die (const char* what, hipError_t e) {
  fprintf (stderr, "welship leaf primitive: %s failed: %s -- no CPU fallback, aborting\n", what, hipGetErrorString (e));
  abort();
}
#define HIPCHK(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) die (#x, e_); } while (0)

// ---- staging: a pool of (page-locked host image, device arena, queue), one per concurrent caller ------------------------------
constexpr size_t kArenaBytes = 16 << 10;
struct Arena { uint8_t* h = nullptr; uint8_t* d = nullptr; hipStream_t q = nullptr; };
std::mutex g_mu;
std::vector<Arena*> g_idle;
std::atomic<uint64_t> g_calls {0};

Arena* arena_get() {
  {
    std::lock_guard<std::mutex> l (g_mu);
    if (!g_idle.empty()) { Arena* a = g_idle.back(); g_idle.pop_back(); return a; }
  }
  int n = 0;
  const hipError_t e = hipGetDeviceCount (&n);
  if (e != hipSuccess || n <= 0) die ("hipGetDeviceCount (no usable device)", e != hipSuccess ? e : hipErrorNoDevice);
  Arena* a = new Arena;
  HIPCHK (hipHostMalloc ((void**)&a->h, kArenaBytes, hipHostMallocDefault));
  HIPCHK (hipMalloc ((void**)&a->d, kArenaBytes));
  HIPCHK (hipStreamCreateWithFlags (&a->q, hipStreamNonBlocking));
  return a;
}
void arena_put (Arena* a) { std::lock_guard<std::mutex> l (g_mu); g_idle.push_back (a); }   // kept for the process' lifetime: as many as there were concurrent callers

struct Rect { size_t base, origin, bytes; int pitch; };   // arena offsets of the rectangle's first byte and of the sample the caller's pointer names

// One leaf call: lay the inputs out in the arena image, upload, launch on the arena's queue, bring the output bytes back.
class Call {
  Arena* a_;
  size_t used_ = 0, out0_ = kArenaBytes, out1_ = 0;
  size_t take (size_t bytes) { const size_t o = used_; used_ = (used_ + bytes + 15) & ~ (size_t)15; if (used_ > kArenaBytes) die ("arena layout", hipErrorOutOfMemory); return o; }
 public:
  Call() : a_ (arena_get()) { g_calls.fetch_add (1, std::memory_order_relaxed); }
  ~Call() { arena_put (a_); }
  Call (const Call&) = delete;
  // the w x h samples whose top-left one is p[y0 * stride + x0] (x0 / y0 <= 0: margins before the sample p names); zero-filled when p is null
  Rect rect (const uint8_t* p, int stride, int x0, int y0, int w, int h) {
    Rect r; r.pitch = (w + 15) & ~15; r.bytes = (size_t)r.pitch * h; r.base = take (r.bytes); r.origin = r.base + (size_t) (-y0) * r.pitch + (size_t) (-x0);
    memset (a_->h + r.base, 0, r.bytes);
    if (p) for (int y = 0; y < h; ++y) memcpy (a_->h + r.base + (size_t)y * r.pitch, p + (ptrdiff_t) (y0 + y) * stride + x0, (size_t)w);
    return r;
  }
  // some samples of a rectangle only (the others stay zero): used where the reference function reads a subset of the neighbours
  void fill (const Rect& r, const uint8_t* p, int stride, int x, int y, int w, int h) {
    for (int j = 0; j < h; ++j) memcpy (a_->h + r.origin + (ptrdiff_t) (y + j) * r.pitch + x, p + (ptrdiff_t) (y + j) * stride + x, (size_t)w);
  }
  size_t raw (const void* p, size_t bytes) { const size_t o = take (bytes); if (p) memcpy (a_->h + o, p, bytes); else memset (a_->h + o, 0, bytes); return o; }
  void out (size_t off, size_t bytes) { if (off < out0_) out0_ = off; if (off + bytes > out1_) out1_ = off + bytes; }
  template <class T> T* dev (size_t off) const { return (T*) (a_->d + off); }
  template <class T> T* host (size_t off) const { return (T*) (a_->h + off); }
  hipStream_t queue() const { return a_->q; }
  void upload() { HIPCHK (hipMemcpyAsync (a_->d, a_->h, used_, hipMemcpyHostToDevice, a_->q)); }
  void finish() {
    HIPCHK (hipGetLastError());
    if (out1_ > out0_) HIPCHK (hipMemcpyAsync (a_->h + out0_, a_->d + out0_, out1_ - out0_, hipMemcpyDeviceToHost, a_->q));
    HIPCHK (hipStreamSynchronize (a_->q));
  }
  // rows [y, y + h) x columns [x, x + w) of a rectangle back into the caller's plane
  void store (const Rect& r, uint8_t* p, int stride, int x, int y, int w, int h) const {
    for (int j = 0; j < h; ++j) memcpy (p + (ptrdiff_t) (y + j) * stride + x, a_->h + r.origin + (ptrdiff_t) (y + j) * r.pitch + x, (size_t)w);
  }
};
#define LAUNCH(c, k, g, b, ...) do { (c).upload(); hipLaunchKernelGGL (k, dim3 (g), dim3 (b), 0, (c).queue(), __VA_ARGS__); (c).finish(); } while (0)

const int kW[7] = {16, 16, 8, 8, 4, 8, 4};
const int kH[7] = {16, 8, 16, 8, 4, 4, 8};

// ---- kernels that exist only at this level (the fused kernels have these steps inlined in their lane code) -----------------------
// quantisation with the caller's own FF / MF rows (encode_mb_aux.cpp:161-224); one thread per 4x4 block
__global__ void k_leaf_quant (int16_t* d, const int16_t* ff, const int16_t* mf, int nblk, int dc, int sff, int smf, int16_t* mx) {
  const int b = threadIdx.x;
  if (b >= nblk) return;
  int16_t m = 0;
  for (int k = 0; k < 16; ++k) {
    int16_t a;
    d[b * 16 + k] = wh_quant1_abs (d[b * 16 + k], dc ? sff : ff[k & 7], dc ? smf : mf[k & 7], &a);
    if (m < a) m = a;
  }
  if (mx) mx[b] = m;
}
// pfQuantizationHadamard2x2 / ..Skip (encode_mb_aux.cpp:226-277): int16 arithmetic as there
__global__ void k_leaf_had2x2 (int16_t* rs, int ff, int mf, int16_t* dct, int16_t* blk, int* ret, int skip) {
  if (threadIdx.x) return;
  int16_t s[4], o[4];
  s[0] = (int16_t) (rs[0] + rs[32]); s[1] = (int16_t) (rs[0] - rs[32]); s[2] = (int16_t) (rs[16] + rs[48]); s[3] = (int16_t) (rs[16] - rs[48]);
  o[0] = (int16_t) (s[0] + s[2]); o[1] = (int16_t) (s[0] - s[2]); o[2] = (int16_t) (s[1] + s[3]); o[3] = (int16_t) (s[1] - s[3]);
  if (skip) {
    const int16_t thr = (int16_t) (((1 << 16) - 1) / mf - ff);
    *ret = (wh_abs (o[0]) > thr) || (wh_abs (o[1]) > thr) || (wh_abs (o[2]) > thr) || (wh_abs (o[3]) > thr);
    return;
  }
  rs[0] = rs[16] = rs[32] = rs[48] = 0;
  int n = 0;
  for (int k = 0; k < 4; ++k) { dct[k] = wh_quant1 (o[k], ff, mf); blk[k] = dct[k]; n += blk[k] != 0; }
  *ret = n;
}
// pfTransformHadamard4x4Dc (encode_mb_aux.cpp:280-311)
__global__ void k_leaf_had4x4dc (int16_t* luma_dc, const int16_t* dct) {
  if (threadIdx.x) return;
  int p[16], s[4];
  for (int i = 0; i < 16; i += 4) {
    const int ix = ((i & 8) << 4) + ((i & 4) << 3);
    s[0] = dct[ix] + dct[ix + 80]; s[3] = dct[ix] - dct[ix + 80]; s[1] = dct[ix + 16] + dct[ix + 64]; s[2] = dct[ix + 16] - dct[ix + 64];
    p[i] = s[0] + s[1]; p[i + 2] = s[0] - s[1]; p[i + 1] = s[3] + s[2]; p[i + 3] = s[3] - s[2];
  }
  for (int i = 0; i < 4; ++i) {
    s[0] = p[i] + p[i + 12]; s[3] = p[i] - p[i + 12]; s[1] = p[i + 4] + p[i + 8]; s[2] = p[i + 4] - p[i + 8];
    luma_dc[i] = (int16_t)wh_clip3 ((s[0] + s[1] + 1) >> 1, -32768, 32767);
    luma_dc[i + 8] = (int16_t)wh_clip3 ((s[0] - s[1] + 1) >> 1, -32768, 32767);
    luma_dc[i + 4] = (int16_t)wh_clip3 ((s[3] + s[2] + 1) >> 1, -32768, 32767);
    luma_dc[i + 12] = (int16_t)wh_clip3 ((s[3] - s[2] + 1) >> 1, -32768, 32767);
  }
}
// pfScan4x4 (mode 0), pfScan4x4Ac (1), pfCalculateSingleCtr4x4 (2), pfGetNoneZeroCount (3) (encode_mb_aux.cpp:371-451)
__global__ void k_leaf_scan (int mode, const int16_t* in, int16_t* out, int* ret) {
  if (threadIdx.x) return;
  if (mode == 0) for (int k = 0; k < 16; ++k) out[k] = in[wh_zigzag (k)];
  else if (mode == 1) for (int k = 0; k < 16; ++k) out[k] = k < 15 ? in[wh_zigzag (k + 1)] : (int16_t)0;
  else if (mode == 2) *ret = wh_single_ctr (in);
  else { int n = 0; for (int k = 0; k < 16; ++k) n += in[k] != 0; *ret = n; }
}
// pfDequantization4x4 / Four4x4 (n coefficients, the caller's row of eight factors), pfDequantizationIHadamard4x4 (decode_mb_aux.cpp:107-152)
__global__ void k_leaf_dequant (int16_t* r, const uint16_t* mf8, int n, int ihad, int mf) {
  if (!ihad) { const int i = threadIdx.x; if (i < n) r[i] = (int16_t) (r[i] * mf8[i & 7]); return; }
  if (threadIdx.x) return;
  int16_t t[4];
  for (int i = 0; i < 16; i += 4) {
    t[0] = (int16_t) (r[i] + r[i + 2]); t[1] = (int16_t) (r[i] - r[i + 2]); t[2] = (int16_t) (r[i + 1] - r[i + 3]); t[3] = (int16_t) (r[i + 1] + r[i + 3]);
    r[i] = (int16_t) (t[0] + t[3]); r[i + 1] = (int16_t) (t[1] + t[2]); r[i + 2] = (int16_t) (t[1] - t[2]); r[i + 3] = (int16_t) (t[0] - t[3]);
  }
  for (int i = 0; i < 4; ++i) {
    t[0] = (int16_t) (r[i] + r[i + 8]); t[1] = (int16_t) (r[i] - r[i + 8]); t[2] = (int16_t) (r[i + 4] - r[i + 12]); t[3] = (int16_t) (r[i + 4] + r[i + 12]);
    r[i] = (int16_t) ((t[0] + t[3]) * mf); r[i + 4] = (int16_t) ((t[1] + t[2]) * mf); r[i + 8] = (int16_t) ((t[1] - t[2]) * mf); r[i + 12] = (int16_t) ((t[0] - t[3]) * mf);
  }
}
// pfIDctT4 / pfIDctFourT4 (one thread per 4x4 block) and pfIDctI16x16Dc (decode_mb_aux.cpp:164-233)
__global__ void k_leaf_idct (uint8_t* rec, int rs, const uint8_t* pred, int ps, const int16_t* c, int nblk, int dc16) {
  if (dc16) {
    for (int k = threadIdx.x; k < 256; k += blockDim.x) { const int y = k >> 4, x = k & 15; rec[y * rs + x] = wh_clip255 (pred[y * ps + x] + ((c[ (y & 12) + (x >> 2)] + 32) >> 6)); }
    return;
  }
  const int b = threadIdx.x;
  if (b >= nblk) return;
  const int bx = (b & 1) * 4, by = (b >> 1) * 4;
  const int16_t* d = c + b * 16;
  int16_t t[16];
  for (int y = 0; y < 4; ++y) wh_idct4_h (d[y * 4], d[y * 4 + 1], d[y * 4 + 2], d[y * 4 + 3], &t[y * 4], &t[y * 4 + 1], &t[y * 4 + 2], &t[y * 4 + 3]);
  for (int x = 0; x < 4; ++x) {
    int r[4];
    wh_idct4_v (t[x], t[4 + x], t[8 + x], t[12 + x], &r[0], &r[1], &r[2], &r[3]);
    for (int y = 0; y < 4; ++y) rec[(by + y) * rs + bx + x] = wh_clip255 (pred[(by + y) * ps + bx + x] + r[y]);
  }
}
// pfLumaHalfpelHor / Ver / Cen (mode 0 / 1 / 2; mc.cpp:187-232) and pfSampleAveraging (3; :162-175); one thread per sample
__global__ void k_leaf_halfpel (int mode, const uint8_t* a, int sa, const uint8_t* b, int sb, int w, int h, uint8_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int y = i / w, x = i - y * w;
  const uint8_t* p = a + y * sa + x;
  out[i] = (uint8_t) (mode == 0 ? wh_mc_b (p) : mode == 1 ? wh_mc_h (p, sa) : mode == 2 ? wh_mc_j (p, sa) : (p[0] + b[y * sb + x] + 1) >> 1);
}
// Intra4x4 with the slot's own neighbour handling: avail bit0 left / bit1 top picks the DC flavour, top_rep = the ..Top_c twins
// (the four samples right of the top row are the top row's last one; get_intra_predictor.cpp:164-184, 265-292)
__global__ void k_leaf_pred4 (const uint8_t* ref, int st, int mode, int avail, int top_rep, uint8_t* out) {
  if (threadIdx.x) return;
  uint8_t Eb[16] = {0};
  for (int k = 0; k < 4; ++k) Eb[3 - k] = ref[k * st - 1];
  Eb[4] = ref[-st - 1];
  for (int k = 0; k < 8; ++k) Eb[5 + k] = ref[-st + (top_rep && k > 3 ? 3 : k)];
  WhE13 E;
  for (int k = 0; k < 4; ++k) E.w[k] = (uint32_t)Eb[4 * k] | ((uint32_t)Eb[4 * k + 1] << 8) | ((uint32_t)Eb[4 * k + 2] << 16) | ((uint32_t)Eb[4 * k + 3] << 24);
  const bool l = avail & 1, t = avail & 2;
  int dc = 128;
  if (l && t) dc = (Eb[0] + Eb[1] + Eb[2] + Eb[3] + Eb[5] + Eb[6] + Eb[7] + Eb[8] + 4) >> 3;
  else if (l) dc = (Eb[0] + Eb[1] + Eb[2] + Eb[3] + 2) >> 2;
  else if (t) dc = (Eb[5] + Eb[6] + Eb[7] + Eb[8] + 2) >> 2;
  for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) out[y * 4 + x] = (uint8_t)wh_pred4_px (mode, x, y, E, dc);
}
// the eight edge filters with the caller's alpha / beta / tc (deblocking_common.cpp:5-181) on the register filters of the deblocking
// kernel (deblock_mb.h wh_db_luma_px / wh_db_chroma_px); sx = distance between the samples across the edge, sy = between lines;
// one thread per line and plane
__global__ void k_leaf_deblock (uint8_t* pix0, uint8_t* pix1, int sx, int sy, int chroma, int eq4, int alpha, int beta, int tc4) {
  const int t = threadIdx.x, lines = chroma ? 8 : 16;
  if (t >= lines * (chroma ? 2 : 1)) return;
  const int l = t % lines;
  uint8_t* q = (t < lines ? pix0 : pix1) + l * sy;
  const int tc = (int) (int8_t) ((tc4 >> ((chroma ? l >> 1 : l >> 2) * 8)) & 255);
  if (chroma) {
    int p0 = q[-sx], q0 = q[0];
    const int bs = eq4 ? 4 : tc > 0 ? 1 : 0;
    wh_db_chroma_px (bs, alpha, beta, eq4 ? 0 : (tc - 1) & 255, q[-2 * sx], p0, q0, q[sx]);
    q[-sx] = (uint8_t)p0; q[0] = (uint8_t)q0;
  } else {
    int p2 = q[-3 * sx], p1 = q[-2 * sx], p0 = q[-sx], q0 = q[0], q1 = q[sx], q2 = q[2 * sx];
    const int bs = eq4 ? 4 : tc >= 0 ? 1 : 0;
    wh_db_luma_px (bs, alpha, beta, eq4 ? 0 : tc & 255, eq4 != 0, eq4 ? q[-4 * sx] : 0, p2, p1, p0, q0, q1, q2, eq4 ? q[3 * sx] : 0);
    q[-3 * sx] = (uint8_t)p2; q[-2 * sx] = (uint8_t)p1; q[-sx] = (uint8_t)p0; q[0] = (uint8_t)q0; q[sx] = (uint8_t)q1; q[2 * sx] = (uint8_t)q2;
  }
}

// ---- the families ------------------------------------------------------------------------------------------------------------
// mode 0 SAD, 1 SATD, 2 the four SADs one sample up / down / left / right of pSample2
void sad_family (int mode, int blk, const uint8_t* p1, int s1, const uint8_t* p2, int s2, int32_t* out) {
  Call c;
  const int w = kW[blk], h = kH[blk], m = mode == 2 ? 1 : 0;
  const Rect a = c.rect (p1, s1, 0, 0, w, h), b = c.rect (p2, s2, -m, -m, w + 2 * m, h + 2 * m);
  const int32_t oa = 0, ob = (int32_t) (b.origin - b.base);
  const size_t da = c.raw (&oa, 4), db = c.raw (&ob, 4), dout = c.raw (nullptr, 16);
  c.out (dout, 16);
  if (blk <= 3) LAUNCH (c, k_sad_wave, 1, 64, blk, c.dev<uint8_t> (a.origin), a.pitch, c.dev<int> (da), c.dev<uint8_t> (b.base), b.bytes, b.pitch, c.dev<int> (db), c.dev<int> (dout), mode);
  else LAUNCH (c, k_sad, 1, 256, blk, 1, c.dev<uint8_t> (a.origin), a.pitch, c.dev<int> (da), c.dev<uint8_t> (b.base), b.pitch, c.dev<int> (db), c.dev<int> (dout), mode);
  memcpy (out, c.host<int32_t> (dout), mode == 2 ? 16 : 4);
}
int32_t sad1 (int mode, int blk, const uint8_t* p1, int s1, const uint8_t* p2, int s2) { int32_t v[4]; sad_family (mode, blk, p1, s1, p2, s2, v); return v[0]; }

void dct_family (int16_t* dct, const uint8_t* p1, int s1, const uint8_t* p2, int s2, int nblk) {
  Call c;
  const int e = nblk == 1 ? 4 : 8;
  const Rect a = c.rect (p1, s1, 0, 0, e, e), b = c.rect (p2, s2, 0, 0, e, e);
  const int32_t oa[4] = {0, 4, 4 * a.pitch, 4 * a.pitch + 4}, ob[4] = {0, 4, 4 * b.pitch, 4 * b.pitch + 4};
  const size_t da = c.raw (oa, 16), db = c.raw (ob, 16), dout = c.raw (nullptr, (size_t)nblk * 32);
  c.out (dout, (size_t)nblk * 32);
  LAUNCH (c, k_dct, 1, 256, nblk, c.dev<uint8_t> (a.origin), a.pitch, c.dev<int> (da), c.dev<uint8_t> (b.origin), b.pitch, c.dev<int> (db), c.dev<int16_t> (dout));
  memcpy (dct, c.host<int16_t> (dout), (size_t)nblk * 32);
}
void quant_family (int16_t* dct, const int16_t* ff, const int16_t* mf, int nblk, int dc, int sff, int smf, int16_t* mx) {
  Call c;
  const int16_t z[8] = {0};
  const size_t dd = c.raw (dct, (size_t)nblk * 32), dff = c.raw (ff ? ff : z, 16), dmf = c.raw (mf ? mf : z, 16), dmx = c.raw (nullptr, 8);
  c.out (dd, (size_t)nblk * 32); c.out (dmx, 8);
  LAUNCH (c, k_leaf_quant, 1, 64, c.dev<int16_t> (dd), c.dev<int16_t> (dff), c.dev<int16_t> (dmf), nblk, dc, sff, smf, c.dev<int16_t> (dmx));
  memcpy (dct, c.host<int16_t> (dd), (size_t)nblk * 32);
  if (mx) memcpy (mx, c.host<int16_t> (dmx), (size_t)nblk * 2);
}
int32_t scan_family (int mode, int16_t* in, int16_t* out) {
  Call c;
  const size_t di = c.raw (in, 32), dou = c.raw (nullptr, 32), dr = c.raw (nullptr, 4);
  c.out (dou, 32); c.out (dr, 4);
  LAUNCH (c, k_leaf_scan, 1, 64, mode, c.dev<int16_t> (di), c.dev<int16_t> (dou), c.dev<int> (dr));
  if (out) memcpy (out, c.host<int16_t> (dou), 32);
  return *c.host<int32_t> (dr);
}
void dequant_family (int16_t* res, const uint16_t* mf8, int n, int ihad, int mf) {
  Call c;
  const uint16_t z[8] = {0};
  const size_t dr = c.raw (res, (size_t)n * 2), dm = c.raw (mf8 ? mf8 : z, 16);
  c.out (dr, (size_t)n * 2);
  LAUNCH (c, k_leaf_dequant, 1, 64, c.dev<int16_t> (dr), c.dev<uint16_t> (dm), n, ihad, mf);
  memcpy (res, c.host<int16_t> (dr), (size_t)n * 2);
}
void idct_family (uint8_t* rec, int rs, const uint8_t* pred, int ps, const int16_t* res, int nblk, int dc16) {
  Call c;
  const int e = dc16 ? 16 : nblk == 1 ? 4 : 8;
  const Rect r = c.rect (nullptr, 0, 0, 0, e, e), p = c.rect (pred, ps, 0, 0, e, e);
  const size_t dc = c.raw (res, dc16 ? 32 : (size_t)nblk * 32);
  c.out (r.base, r.bytes);
  LAUNCH (c, k_leaf_idct, 1, 64, c.dev<uint8_t> (r.origin), r.pitch, c.dev<uint8_t> (p.origin), p.pitch, c.dev<int16_t> (dc), nblk, dc16);
  c.store (r, rec, rs, 0, 0, e, e);
}
// pMcLumaFunc / pMcChromaFunc: pSrc names the integer sample position, the vector supplies the fraction (mc.cpp:335-378)
void mc_family (const uint8_t* src, int ss, uint8_t* dst, int ds, int mvx, int mvy, int w, int h, int chroma) {
  Call c;
  const int fx = mvx & (chroma ? 7 : 3), fy = mvy & (chroma ? 7 : 3);
  const int l = chroma ? 0 : fx ? 2 : 0, r = chroma ? (fx || fy) : fx ? 3 : 0, t = chroma ? 0 : fy ? 2 : 0, b = chroma ? (fx || fy) : fy ? 3 : 0;
  const Rect s = c.rect (src, ss, -l, -t, w + l + r, h + t + b);
  const int32_t off = (int32_t) (s.origin - s.base);
  const int16_t mv[2] = { (int16_t)mvx, (int16_t)mvy};
  const size_t dof = c.raw (&off, 4), dmv = c.raw (mv, 4), dout = c.raw (nullptr, (size_t)w * h);
  c.out (dout, (size_t)w * h);
  if (!chroma && (w == 16 || w == 8) && (h == 16 || h == 8))
    LAUNCH (c, k_mc_wave, 1, 64, c.dev<uint8_t> (s.base), s.bytes, s.pitch, c.dev<int> (dof), c.dev<int16_t> (dmv), w, h, c.dev<uint8_t> (dout));
  else
    LAUNCH (c, k_mc, 1, 256, 1, c.dev<uint8_t> (s.base), s.pitch, c.dev<int> (dof), c.dev<int16_t> (dmv), w, h, chroma, c.dev<uint8_t> (dout));
  for (int y = 0; y < h; ++y) memcpy (dst + (ptrdiff_t)y * ds, c.host<uint8_t> (dout) + y * w, (size_t)w);
}
void halfpel_family (int mode, const uint8_t* a, int sa, const uint8_t* b, int sb, uint8_t* dst, int ds, int w, int h) {
  Call c;
  const int mx = (mode == 0 || mode == 2) ? 2 : 0, my = (mode == 1 || mode == 2) ? 2 : 0;
  const Rect ra = c.rect (a, sa, -mx, -my, w + (mx ? 5 : 0), h + (my ? 5 : 0)), rb = c.rect (mode == 3 ? b : nullptr, sb, 0, 0, w, h);
  const size_t dout = c.raw (nullptr, (size_t)w * h);
  c.out (dout, (size_t)w * h);
  LAUNCH (c, k_leaf_halfpel, grid (w * h), 256, mode, c.dev<uint8_t> (ra.origin), ra.pitch, c.dev<uint8_t> (rb.origin), rb.pitch, w, h, c.dev<uint8_t> (dout));
  for (int y = 0; y < h; ++y) memcpy (dst + (ptrdiff_t)y * ds, c.host<uint8_t> (dout) + y * w, (size_t)w);
}
enum { kT = 1, kL = 2, kTL = 4, kTR = 8 };   // the neighbours a predictor reads (only those are fetched from the caller's plane)
void pred4_family (uint8_t* pred, const uint8_t* ref, int st, int mode, int avail, int top_rep, int need) {
  Call c;
  const Rect r = c.rect (nullptr, 0, -1, -1, 9, 5);
  if (need & kT) c.fill (r, ref, st, 0, -1, 4, 1);
  if (need & kTR) c.fill (r, ref, st, 4, -1, 4, 1);
  if (need & kTL) c.fill (r, ref, st, -1, -1, 1, 1);
  if (need & kL) c.fill (r, ref, st, -1, 0, 1, 4);
  const size_t dout = c.raw (nullptr, 16);
  c.out (dout, 16);
  LAUNCH (c, k_leaf_pred4, 1, 64, c.dev<uint8_t> (r.origin), r.pitch, mode, avail, top_rep, c.dev<uint8_t> (dout));
  memcpy (pred, c.host<uint8_t> (dout), 16);
}
// Intra16x16 (chroma = 0) and chroma 8x8 (1) through the macroblock tile code (k_pred_mb predicts both; the other one gets a zero
// tile and its no-neighbour DC mode).  The chroma tile holds the plane twice, 16 columns apart, as k_pred_mb's Cb / Cr.
void predmb_family (uint8_t* pred, const uint8_t* ref, int st, int mode, int chroma, int need) {
  Call c;
  const int e = chroma ? 8 : 16;
  const Rect y = c.rect (nullptr, 0, -1, -1, 17, 17), uv = c.rect (nullptr, 0, -1, -1, 32, 9);
  const Rect& r = chroma ? uv : y;
  if (need & kT) c.fill (r, ref, st, 0, -1, e, 1);
  if (need & kTL) c.fill (r, ref, st, -1, -1, 1, 1);
  if (need & kL) c.fill (r, ref, st, -1, 0, 1, e);
  const int32_t oy = (int32_t) (y.origin - y.base), oc = (int32_t) (uv.origin - uv.base);
  const uint8_t m16 = (uint8_t) (chroma ? 6 /* I16_PRED_DC_128 */ : mode), mc = (uint8_t) (chroma ? mode : 6 /* C_PRED_DC_128 */);
  const size_t doy = c.raw (&oy, 4), doc = c.raw (&oc, 4), dm = c.raw (&m16, 1), dmc = c.raw (&mc, 1), d16 = c.raw (nullptr, 256), dc8 = c.raw (nullptr, 128);
  c.out (d16, 256); c.out (dc8, 128);
  LAUNCH (c, k_pred_mb, 1, 64, c.dev<uint8_t> (y.base), y.pitch, c.dev<int> (doy), c.dev<uint8_t> (uv.base), uv.pitch, c.dev<int> (doc), c.dev<uint8_t> (dm), c.dev<uint8_t> (dmc),
          c.dev<uint8_t> (d16), c.dev<uint8_t> (dc8));
  memcpy (pred, chroma ? c.host<uint8_t> (dc8) : c.host<uint8_t> (d16), chroma ? 64 : 256);
}
// across: 0 = the neighbours of a line are one stride apart ("V": a horizontal edge), 1 = adjacent bytes ("H": a vertical edge)
void deblock_family (uint8_t* p0, uint8_t* p1, int st, int alpha, int beta, const int8_t* tc, int chroma, int eq4, int across_bytes) {
  Call c;
  const int lines = chroma ? 8 : 16, rd = chroma ? 2 : eq4 ? 4 : 3, wr = chroma ? 1 : eq4 ? 3 : 2;   // samples read / possibly changed on each side
  const int x0 = across_bytes ? -rd : 0, y0 = across_bytes ? 0 : -rd, w = across_bytes ? 2 * rd : lines, h = across_bytes ? lines : 2 * rd;
  const Rect a = c.rect (p0, st, x0, y0, w, h), b = c.rect (p1, st, x0, y0, w, h);
  int tc4 = 0;
  if (tc) memcpy (&tc4, tc, 4);
  c.out (a.base, a.bytes); if (p1) c.out (b.base, b.bytes);
  LAUNCH (c, k_leaf_deblock, 1, 64, c.dev<uint8_t> (a.origin), c.dev<uint8_t> (b.origin), across_bytes ? 1 : a.pitch, across_bytes ? a.pitch : 1, chroma, eq4, alpha, beta, tc4);
  const int sx0 = across_bytes ? -wr : 0, sy0 = across_bytes ? 0 : -wr, sw = across_bytes ? 2 * wr : lines, sh = across_bytes ? lines : 2 * wr;
  c.store (a, p0, st, sx0, sy0, sw, sh);
  if (p1) c.store (b, p1, st, sx0, sy0, sw, sh);
}

}  // namespace

// pfCopy* (copy_mb.cpp:48-111): w x h samples from one plane to another.  The device moves them (dwords where both rows allow it are not worth a
// second kernel: these are 16 .. 256 bytes).
__global__ void k_leaf_copy (const uint8_t* src, int sp, int w, int h, uint8_t* dst) {
  for (int i = (int)threadIdx.x; i < w * h; i += (int)blockDim.x) { const int y = i / w, x = i - y * w; dst[i] = src[y * sp + x]; }
}
void copy_family (uint8_t* dst, int ds, const uint8_t* src, int ss, int w, int h) {
  Call c;
  const Rect rs = c.rect (src, ss, 0, 0, w, h);
  const size_t dout = c.raw (nullptr, (size_t)w * h);
  c.out (dout, (size_t)w * h);
  LAUNCH (c, k_leaf_copy, 1, 256, c.dev<uint8_t> (rs.origin), rs.pitch, w, h, c.dev<uint8_t> (dout));
  for (int y = 0; y < h; ++y) memcpy (dst + (ptrdiff_t)y * ds, c.host<uint8_t> (dout) + y * w, (size_t)w);
}
// pfSetMemZero* (copy_mb.cpp:38-46 WelsSetMemZero_c): the zeros come from the device too, a piece of the arena at a time
__global__ void k_leaf_zero (uint32_t* d, int words) { for (int i = (int)threadIdx.x; i < words; i += (int)blockDim.x) d[i] = 0u; }
void zero_family (uint8_t* dst, int size) {
  while (size > 0) {
    Call c;
    const int n = size < (int) (kArenaBytes / 2) ? size : (int) (kArenaBytes / 2);
    const size_t dout = c.raw (dst, (size_t)n);          // (uploaded as it is: only the device's zeros come back)
    c.out (dout, (size_t)n);
    LAUNCH (c, k_leaf_zero, 1, 256, c.dev<uint32_t> (dout), (n + 3) / 4);
    memcpy (dst, c.host<uint8_t> (dout), (size_t)n);
    dst += n; size -= n;
  }
}

extern "C" {

int WelsHipLeafAvailable (void) { int n = 0; return hipGetDeviceCount (&n) == hipSuccess && n > 0 ? WELSHIP_OK : WELSHIP_ERR_NO_DEVICE; }
uint64_t WelsHipLeafCalls (void) { return g_calls.load(); }

#define SAD_SLOT(name, blk) \
  int32_t WelsHipSampleSad##name (uint8_t* a, int32_t sa, uint8_t* b, int32_t sb) { return sad1 (0, blk, a, sa, b, sb); } \
  int32_t WelsHipSampleSatd##name (uint8_t* a, int32_t sa, uint8_t* b, int32_t sb) { return sad1 (1, blk, a, sa, b, sb); } \
  void WelsHipSampleSadFour##name (uint8_t* a, int32_t sa, uint8_t* b, int32_t sb, int32_t* pSad) { sad_family (2, blk, a, sa, b, sb, pSad); }
SAD_SLOT (16x16, 0) SAD_SLOT (16x8, 1) SAD_SLOT (8x16, 2) SAD_SLOT (8x8, 3) SAD_SLOT (4x4, 4) SAD_SLOT (8x4, 5) SAD_SLOT (4x8, 6)

void WelsHipDctT4 (int16_t* pDct, uint8_t* p1, int32_t s1, uint8_t* p2, int32_t s2) { dct_family (pDct, p1, s1, p2, s2, 1); }
void WelsHipDctFourT4 (int16_t* pDct, uint8_t* p1, int32_t s1, uint8_t* p2, int32_t s2) { dct_family (pDct, p1, s1, p2, s2, 4); }
void WelsHipQuant4x4 (int16_t* pDct, const int16_t* pFF, const int16_t* pMF) { quant_family (pDct, pFF, pMF, 1, 0, 0, 0, nullptr); }
void WelsHipQuant4x4Dc (int16_t* pDct, int16_t iFF, int16_t iMF) { quant_family (pDct, nullptr, nullptr, 1, 1, iFF, iMF, nullptr); }
void WelsHipQuantFour4x4 (int16_t* pDct, const int16_t* pFF, const int16_t* pMF) { quant_family (pDct, pFF, pMF, 4, 0, 0, 0, nullptr); }
void WelsHipQuantFour4x4Max (int16_t* pDct, const int16_t* pFF, const int16_t* pMF, int16_t* pMax) { quant_family (pDct, pFF, pMF, 4, 0, 0, 0, pMax); }
static int32_t had2x2 (int16_t* pRes, int ff, int mf, int16_t* pDct, int16_t* pBlock, int skip) {
  Call c;
  const size_t dr = c.raw (pRes, 128), dd = c.raw (nullptr, 8), db = c.raw (nullptr, 8), dret = c.raw (nullptr, 4);
  c.out (dr, 128); c.out (dd, 8); c.out (db, 8); c.out (dret, 4);
  LAUNCH (c, k_leaf_had2x2, 1, 64, c.dev<int16_t> (dr), ff, mf, c.dev<int16_t> (dd), c.dev<int16_t> (db), c.dev<int> (dret), skip);
  if (!skip) { memcpy (pRes, c.host<int16_t> (dr), 128); memcpy (pDct, c.host<int16_t> (dd), 8); memcpy (pBlock, c.host<int16_t> (db), 8); }
  return *c.host<int32_t> (dret);
}
int32_t WelsHipHadamardQuant2x2 (int16_t* pRes, const int16_t kiFF, int16_t iMF, int16_t* pDct, int16_t* pBlock) { return had2x2 (pRes, kiFF, iMF, pDct, pBlock, 0); }
int32_t WelsHipHadamardQuant2x2Skip (int16_t* pRes, int16_t iFF, int16_t iMF) { return had2x2 (pRes, iFF, iMF, nullptr, nullptr, 1); }
void WelsHipHadamardT4Dc (int16_t* pLumaDc, int16_t* pDct) {
  Call c;
  const size_t dd = c.raw (pDct, 512), dl = c.raw (nullptr, 32);
  c.out (dl, 32);
  LAUNCH (c, k_leaf_had4x4dc, 1, 64, c.dev<int16_t> (dl), c.dev<int16_t> (dd));
  memcpy (pLumaDc, c.host<int16_t> (dl), 32);
}
void WelsHipScan4x4DcAc (int16_t* pLevel, int16_t* pDct) { (void)scan_family (0, pDct, pLevel); }
void WelsHipScan4x4Ac (int16_t* pLevel, int16_t* pDct) { (void)scan_family (1, pDct, pLevel); }
int32_t WelsHipCalculateSingleCtr4x4 (int16_t* pDct) { return scan_family (2, pDct, nullptr); }
int32_t WelsHipGetNoneZeroCount (int16_t* pLevel) { return scan_family (3, pLevel, nullptr); }

void WelsHipDequant4x4 (int16_t* pRes, const uint16_t* kpQpTable) { dequant_family (pRes, kpQpTable, 16, 0, 0); }
void WelsHipDequantFour4x4 (int16_t* pRes, const uint16_t* kpQpTable) { dequant_family (pRes, kpQpTable, 64, 0, 0); }
void WelsHipDequantIHadamard4x4 (int16_t* pRes, const uint16_t kuiMF) { dequant_family (pRes, nullptr, 16, 1, kuiMF); }
void WelsHipIDctT4Rec (uint8_t* pRec, int32_t iStride, uint8_t* pPred, int32_t iPredStride, int16_t* pRes) { idct_family (pRec, iStride, pPred, iPredStride, pRes, 1, 0); }
void WelsHipIDctFourT4Rec (uint8_t* pRec, int32_t iStride, uint8_t* pPred, int32_t iPredStride, int16_t* pRes) { idct_family (pRec, iStride, pPred, iPredStride, pRes, 4, 0); }
void WelsHipIDctRecI16x16Dc (uint8_t* pRec, int32_t iStride, uint8_t* pPred, int32_t iPredStride, int16_t* pRes) { idct_family (pRec, iStride, pPred, iPredStride, pRes, 0, 1); }

void WelsHipMcLuma (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int16_t iMvX, int16_t iMvY, int32_t iWidth, int32_t iHeight) { mc_family (pSrc, iSrcStride, pDst, iDstStride, iMvX, iMvY, iWidth, iHeight, 0); }
void WelsHipMcChroma (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int16_t iMvX, int16_t iMvY, int32_t iWidth, int32_t iHeight) { mc_family (pSrc, iSrcStride, pDst, iDstStride, iMvX, iMvY, iWidth, iHeight, 1); }
void WelsHipMcHorVer20 (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int32_t iWidth, int32_t iHeight) { halfpel_family (0, pSrc, iSrcStride, nullptr, 0, pDst, iDstStride, iWidth, iHeight); }
void WelsHipMcHorVer02 (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int32_t iWidth, int32_t iHeight) { halfpel_family (1, pSrc, iSrcStride, nullptr, 0, pDst, iDstStride, iWidth, iHeight); }
void WelsHipMcHorVer22 (const uint8_t* pSrc, int32_t iSrcStride, uint8_t* pDst, int32_t iDstStride, int32_t iWidth, int32_t iHeight) { halfpel_family (2, pSrc, iSrcStride, nullptr, 0, pDst, iDstStride, iWidth, iHeight); }
void WelsHipPixelAvg (uint8_t* pDst, int32_t iDstStride, const uint8_t* pSrcA, int32_t iSrcAStride, const uint8_t* pSrcB, int32_t iSrcBStride, int32_t iWidth, int32_t iHeight) { halfpel_family (3, pSrcA, iSrcAStride, pSrcB, iSrcBStride, pDst, iDstStride, iWidth, iHeight); }

// Intra4x4: (standard mode 0..8, DC flavour, top-right replaced, neighbours read)
#define I4_SLOT(name, mode, avail, top_rep, need) void WelsHipI4x4LumaPred##name (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride) { pred4_family (pPred, pRef, kiStride, mode, avail, top_rep, need); }
I4_SLOT (V, 0, 3, 0, kT) I4_SLOT (H, 1, 3, 0, kL) I4_SLOT (Dc, 2, 3, 0, kT | kL) I4_SLOT (DcLeft, 2, 1, 0, kL) I4_SLOT (DcTop, 2, 2, 0, kT) I4_SLOT (DcNA, 2, 0, 0, 0)
I4_SLOT (DDL, 3, 3, 0, kT | kTR) I4_SLOT (DDLTop, 3, 3, 1, kT) I4_SLOT (DDR, 4, 3, 0, kT | kL | kTL) I4_SLOT (VR, 5, 3, 0, kT | kL | kTL) I4_SLOT (HD, 6, 3, 0, kT | kL | kTL)
I4_SLOT (VL, 7, 3, 0, kT | kTR) I4_SLOT (VLTop, 7, 3, 1, kT) I4_SLOT (HU, 8, 3, 0, kL)
// I16_PRED_* / C_PRED_* numbering of the reference (wels_common_defs.h:330-371) = WH_I16_* / WH_C_* of the kernels
#define I16_SLOT(name, mode, need) void WelsHipI16x16LumaPred##name (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride) { predmb_family (pPred, pRef, kiStride, mode, 0, need); }
I16_SLOT (V, 0, kT) I16_SLOT (H, 1, kL) I16_SLOT (Dc, 2, kT | kL) I16_SLOT (Plane, 3, kT | kL | kTL) I16_SLOT (DcLeft, 4, kL) I16_SLOT (DcTop, 5, kT) I16_SLOT (DcNA, 6, 0)
#define IC_SLOT(name, mode, need) void WelsHipIChromaPred##name (uint8_t* pPred, uint8_t* pRef, const int32_t kiStride) { predmb_family (pPred, pRef, kiStride, mode, 1, need); }
IC_SLOT (Dc, 0, kT | kL) IC_SLOT (H, 1, kL) IC_SLOT (V, 2, kT) IC_SLOT (Plane, 3, kT | kL | kTL) IC_SLOT (DcLeft, 4, kL) IC_SLOT (DcTop, 5, kT) IC_SLOT (DcNA, 6, 0)

void WelsHipDeblockLumaLt4V (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc) { deblock_family (pPix, nullptr, iStride, iAlpha, iBeta, pTc, 0, 0, 0); }
void WelsHipDeblockLumaEq4V (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta) { deblock_family (pPix, nullptr, iStride, iAlpha, iBeta, nullptr, 0, 1, 0); }
void WelsHipDeblockLumaLt4H (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc) { deblock_family (pPix, nullptr, iStride, iAlpha, iBeta, pTc, 0, 0, 1); }
void WelsHipDeblockLumaEq4H (uint8_t* pPix, int32_t iStride, int32_t iAlpha, int32_t iBeta) { deblock_family (pPix, nullptr, iStride, iAlpha, iBeta, nullptr, 0, 1, 1); }
void WelsHipDeblockChromaLt4V (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc) { deblock_family (pPixCb, pPixCr, iStride, iAlpha, iBeta, pTc, 1, 0, 0); }
void WelsHipDeblockChromaEq4V (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta) { deblock_family (pPixCb, pPixCr, iStride, iAlpha, iBeta, nullptr, 1, 1, 0); }
void WelsHipDeblockChromaLt4H (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta, int8_t* pTc) { deblock_family (pPixCb, pPixCr, iStride, iAlpha, iBeta, pTc, 1, 0, 1); }
void WelsHipDeblockChromaEq4H (uint8_t* pPixCb, uint8_t* pPixCr, int32_t iStride, int32_t iAlpha, int32_t iBeta) { deblock_family (pPixCb, pPixCr, iStride, iAlpha, iBeta, nullptr, 1, 1, 1); }


// PCopyFunc pfCopy16x16Aligned / pfCopy16x16NotAligned / pfCopy8x8Aligned / pfCopy16x8NotAligned / pfCopy8x16Aligned / pfCopy4x4 / pfCopy8x4 / pfCopy4x8
// (wels_func_ptr_def.h:238-245; copy_mb.cpp:48-111), PSetMemoryZero pfSetMemZeroSize8 / ..Size64Aligned16 / ..Size64 (:285-287; copy_mb.cpp:38-46)
#define COPY_SLOT(name, w, h) void WelsHipCopy##name (uint8_t* pDst, int32_t iStrideD, uint8_t* pSrc, int32_t iStrideS) { copy_family (pDst, iStrideD, pSrc, iStrideS, w, h); }
COPY_SLOT (4x4, 4, 4) COPY_SLOT (8x4, 8, 4) COPY_SLOT (4x8, 4, 8) COPY_SLOT (8x8, 8, 8) COPY_SLOT (16x8, 16, 8) COPY_SLOT (8x16, 8, 16) COPY_SLOT (16x16, 16, 16)
void WelsHipSetMemZero (void* pDst, int32_t iSize) { zero_family ((uint8_t*)pDst, iSize); }

// PIntraPred*Combined3Func (wels_func_ptr_def.h:129-133, slots :166-176; sample.cpp:153-331): the three cheapest intra modes of a block costed in
// one call.  NULL in the reference's C build (sample.cpp:363-367: only the SIMD builds fill them); here each is the reference's `_c` text over
// the device-backed predictors and costs above -- same order of the candidates, same tie-breaking (`<`), same bytes left in the caller's buffers.
int32_t WelsHipIntra4x4Combined3Satd (uint8_t* pDec, int32_t iDecStride, uint8_t* pEnc, int32_t iEncStride, uint8_t* pDst, int32_t* pBestMode,
                                     int32_t iLambda2, int32_t iLambda1, int32_t iLambda0) {
  uint8_t buf[3][16];
  int32_t best_mode = -1, best = 0x7fffffff, cur;
  WelsHipI4x4LumaPredDc (buf[2], pDec, iDecStride);
  cur = WelsHipSampleSatd4x4 (buf[2], 4, pEnc, iEncStride) + iLambda2; if (cur < best) { best_mode = 2; best = cur; }
  WelsHipI4x4LumaPredH (buf[1], pDec, iDecStride);
  cur = WelsHipSampleSatd4x4 (buf[1], 4, pEnc, iEncStride) + iLambda1; if (cur < best) { best_mode = 1; best = cur; }
  WelsHipI4x4LumaPredV (buf[0], pDec, iDecStride);
  cur = WelsHipSampleSatd4x4 (buf[0], 4, pEnc, iEncStride) + iLambda0; if (cur < best) { best_mode = 0; best = cur; }
  memcpy (pDst, buf[best_mode], 16);
  *pBestMode = best_mode;
  return best;
}
static int32_t combined3_16 (bool satd, uint8_t* pDec, int32_t iDecStride, uint8_t* pEnc, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda, uint8_t* pDst) {
  int32_t best_mode = -1, best = 0x7fffffff, cur;
  auto cost = [&] () { return satd ? WelsHipSampleSatd16x16 (pDst, 16, pEnc, iEncStride) : WelsHipSampleSad16x16 (pDst, 16, pEnc, iEncStride); };
  WelsHipI16x16LumaPredV (pDst, pDec, iDecStride);  cur = cost();               if (cur < best) { best_mode = 0; best = cur; }
  WelsHipI16x16LumaPredH (pDst, pDec, iDecStride);  cur = cost() + iLambda * 2; if (cur < best) { best_mode = 1; best = cur; }
  WelsHipI16x16LumaPredDc (pDst, pDec, iDecStride); cur = cost() + iLambda * 2; if (cur < best) { best_mode = 2; best = cur; }
  *pBestMode = best_mode;
  return best;
}
int32_t WelsHipIntra16x16Combined3Satd (uint8_t* pDec, int32_t iDecStride, uint8_t* pEnc, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda, uint8_t* pDst) {
  return combined3_16 (true, pDec, iDecStride, pEnc, iEncStride, pBestMode, iLambda, pDst);
}
int32_t WelsHipIntra16x16Combined3Sad (uint8_t* pDec, int32_t iDecStride, uint8_t* pEnc, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda, uint8_t* pDst) {
  return combined3_16 (false, pDec, iDecStride, pEnc, iEncStride, pBestMode, iLambda, pDst);
}
static int32_t combined3_8 (bool satd, uint8_t* pDecCb, int32_t iDecStride, uint8_t* pEncCb, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda,
                            uint8_t* pDstChroma, uint8_t* pDecCr, uint8_t* pEncCr) {
  int32_t best_mode = -1, best = 0x7fffffff, cur;
  auto cost = [&] () { return satd ? WelsHipSampleSatd8x8 (pDstChroma, 8, pEncCb, iEncStride) + WelsHipSampleSatd8x8 (pDstChroma + 64, 8, pEncCr, iEncStride)
                                   : WelsHipSampleSad8x8 (pDstChroma, 8, pEncCb, iEncStride) + WelsHipSampleSad8x8 (pDstChroma + 64, 8, pEncCr, iEncStride); };
  WelsHipIChromaPredV (pDstChroma, pDecCb, iDecStride);  WelsHipIChromaPredV (pDstChroma + 64, pDecCr, iDecStride);  cur = cost() + iLambda * 2; if (cur < best) { best_mode = 2; best = cur; }
  WelsHipIChromaPredH (pDstChroma, pDecCb, iDecStride);  WelsHipIChromaPredH (pDstChroma + 64, pDecCr, iDecStride);  cur = cost() + iLambda * 2; if (cur < best) { best_mode = 1; best = cur; }
  WelsHipIChromaPredDc (pDstChroma, pDecCb, iDecStride); WelsHipIChromaPredDc (pDstChroma + 64, pDecCr, iDecStride); cur = cost();               if (cur < best) { best_mode = 0; best = cur; }
  *pBestMode = best_mode;
  return best;
}
int32_t WelsHipIntra8x8Combined3Satd (uint8_t* pDecCb, int32_t iDecStride, uint8_t* pEncCb, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda,
                                     uint8_t* pDstChroma, uint8_t* pDecCr, uint8_t* pEncCr) {
  return combined3_8 (true, pDecCb, iDecStride, pEncCb, iEncStride, pBestMode, iLambda, pDstChroma, pDecCr, pEncCr);
}
int32_t WelsHipIntra8x8Combined3Sad (uint8_t* pDecCb, int32_t iDecStride, uint8_t* pEncCb, int32_t iEncStride, int32_t* pBestMode, int32_t iLambda,
                                    uint8_t* pDstChroma, uint8_t* pDecCr, uint8_t* pEncCr) {
  return combined3_8 (false, pDecCb, iDecStride, pEncCb, iEncStride, pBestMode, iLambda, pDstChroma, pDecCr, pEncCr);
}

}  // extern "C"
