// prims_kernels.h -- the kernels of the leaf-primitive layer (prims.hip: batched over arrays of blocks; leaf.hip: one call of the
// reference's own function-pointer signature each).  Every kernel calls the device functions the fused macroblock kernels use
// (kernels/prims.h, intra_mb.h, inter_mb.h, mc_px.h, deblock_mb.h), so the parity tests of these layers pin the hot path's arithmetic.
// Included inside an unnamed namespace by both translation units.
#pragma once
__device__ const int kBw[7] = {16, 16, 8, 8, 4, 8, 4};
__device__ const int kBh[7] = {16, 8, 16, 8, 4, 4, 8};

__device__ int dev_sad (int blk, const uint8_t* a, int sa, const uint8_t* b, int sb) {
  int s = 0;
  for (int y = 0; y < kBh[blk]; ++y) for (int x = 0; x < kBw[blk]; ++x) s += wh_abs (a[y * sa + x] - b[y * sb + x]);
  return s;
}
__device__ int dev_satd4 (const uint8_t* a, int sa, const uint8_t* b, int sb) {
  int m[16], s = 0;
  for (int y = 0; y < 4; ++y) wh_had4 (a[y * sa] - b[y * sb], a[y * sa + 1] - b[y * sb + 1], a[y * sa + 2] - b[y * sb + 2], a[y * sa + 3] - b[y * sb + 3], &m[y * 4], &m[y * 4 + 1], &m[y * 4 + 2], &m[y * 4 + 3]);
  for (int x = 0; x < 4; ++x) { int o0, o1, o2, o3; wh_had4 (m[x], m[4 + x], m[8 + x], m[12 + x], &o0, &o1, &o2, &o3); s += wh_abs (o0) + wh_abs (o1) + wh_abs (o2) + wh_abs (o3); }
  return (s + 1) >> 1;
}
__global__ void k_sad (int blk, int n, const uint8_t* p1, int s1, const int* o1, const uint8_t* p2, int s2, const int* o2, int* out, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* a = p1 + o1[i];
  const uint8_t* b = p2 + o2[i];
  if (mode == 0) out[i] = dev_sad (blk, a, s1, b, s2);
  else if (mode == 1) { int s = 0; for (int y = 0; y < kBh[blk]; y += 4) for (int x = 0; x < kBw[blk]; x += 4) s += dev_satd4 (a + y * s1 + x, s1, b + y * s2 + x, s2); out[i] = s; }
  else { out[i * 4] = dev_sad (blk, a, s1, b - s2, s2); out[i * 4 + 1] = dev_sad (blk, a, s1, b + s2, s2); out[i * 4 + 2] = dev_sad (blk, a, s1, b - 1, s2); out[i * 4 + 3] = dev_sad (blk, a, s1, b + 1, s2); }
}

// Wave-level variants for the partition sizes the macroblock kernel handles (16x16, 16x8, 8x16, 8x8): one wavefront
// per block, through the very lane primitives of kernels/inter_mb.h (register SATD on DPP, packed-byte SAD, 4-samples-
// per-lane interpolation from an LDS window) -- so the oracle comparison of this layer pins the hot path's arithmetic.
__global__ __launch_bounds__ (64) void k_sad_wave (int blk, const uint8_t* p1, int s1, const int* o1, const uint8_t* p2, size_t b2, int s2, const int* o2, int* out, int mode) {
  __shared__ WhInterLds S;
  __shared__ WhWinLds WB;
  WhWin W; W.x0 = W.y0 = W.cx0 = W.cy0 = 0; W.b = &WB;
  const int i = blockIdx.x, bw = kBw[blk], bh = kBh[blk];
  const uint8_t* a = p1 + o1[i];
  const long base = (long)o2[i] - 8 * s2 - 8;                 // window origin = block position - (8,8)
  WV_LANES_BEGIN (lane)
  for (int k = lane; k < bw * bh; k += 64) S.m.enc_y[(k / bw) * 16 + k % bw] = a[(k / bw) * s1 + k % bw];
  for (int k = lane; k < 32 * 64; k += 64) {              // 32 rows x 64 columns of the plane at the window's row pitch (the block at (8,8) + every tap)
    long ad = base + (long) (k >> 6) * s2 + (k & 63);
    ad = ad < 0 ? 0 : (ad > (long)b2 - 1 ? (long)b2 - 1 : ad);
    WB.win[(k >> 6) * WH_WIN_STRIDE + (k & 63)] = p2[ad];
  }
  WV_LANES_END
  const int wo = 8 * WH_WIN_STRIDE + 8;
  if (mode == 0) {
    const int s = wh_sad_win (S, W, 0, 0, bw, bh, wo);
    if (threadIdx.x == 0) out[i] = s;
  } else if (mode == 1) {
    const int nq = (bw >> 2) * (bh >> 2) * 4;
    int s;
    WV_SATD_ROWS (s, lane, lane < nq, wh_enc4 (S, lane < nq ? wh_tl_col (lane, bw) : 0, lane < nq ? wh_tl_row (lane, bw) : 0),
                  wh_ld4u (WB.win, wo + (lane < nq ? wh_tl_row (lane, bw) * WH_WIN_STRIDE + wh_tl_col (lane, bw) : 0)));
    if (threadIdx.x == 0) out[i] = s;
  } else {
    const int s0 = wh_sad_win (S, W, 0, 0, bw, bh, wo - WH_WIN_STRIDE), s1v = wh_sad_win (S, W, 0, 0, bw, bh, wo + WH_WIN_STRIDE);
    const int s2v = wh_sad_win (S, W, 0, 0, bw, bh, wo - 1), s3 = wh_sad_win (S, W, 0, 0, bw, bh, wo + 1);
    if (threadIdx.x == 0) { out[i * 4] = s0; out[i * 4 + 1] = s1v; out[i * 4 + 2] = s2v; out[i * 4 + 3] = s3; }
  }
}
__global__ __launch_bounds__ (64) void k_mc_wave (const uint8_t* plane, size_t bytes, int st, const int* off, const int16_t* mv, int w, int h, uint8_t* out) {
  __shared__ WhWinLds WB;
  WhWin W; W.x0 = W.y0 = W.cx0 = W.cy0 = 0; W.b = &WB;
  const int i = blockIdx.x;
  const long base = (long)off[i] - 8 * st - 8;
  WV_LANES_BEGIN (lane)
  for (int k = lane; k < 32 * 64; k += 64) {
    long ad = base + (long) (k >> 6) * st + (k & 63);
    ad = ad < 0 ? 0 : (ad > (long)bytes - 1 ? (long)bytes - 1 : ad);
    WB.win[(k >> 6) * WH_WIN_STRIDE + (k & 63)] = plane[ad];
  }
  WV_LANES_END
  const int fx = mv[i * 2] & 3, fy = mv[i * 2 + 1] & 3, wo = 8 * WH_WIN_STRIDE + 8;
  WV_LANES_BEGIN (lane)
  if (lane < (w * h) >> 2) {
    const int r = wh_sl_row (lane, w), c = wh_sl_col (lane, w);
    const uint32_t v = wh_mc4 (WB.win, wo + r * WH_WIN_STRIDE + c, fx, fy);
    uint8_t* d = out + (size_t)i * w * h + r * w + c;
    d[0] = (uint8_t)v; d[1] = (uint8_t) (v >> 8); d[2] = (uint8_t) (v >> 16); d[3] = (uint8_t) (v >> 24);
  }
  WV_LANES_END
}

__global__ void k_dct (int n, const uint8_t* p1, int s1, const int* o1, const uint8_t* p2, int s2, const int* o2, int16_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* a = p1 + o1[i];
  const uint8_t* b = p2 + o2[i];
  int16_t t[16];
  for (int y = 0; y < 4; ++y) wh_fdct4 (a[y * s1] - b[y * s2], a[y * s1 + 1] - b[y * s2 + 1], a[y * s1 + 2] - b[y * s2 + 2], a[y * s1 + 3] - b[y * s2 + 3], &t[y * 4], &t[y * 4 + 1], &t[y * 4 + 2], &t[y * 4 + 3]);
  int16_t* d = out + i * 16;
  for (int x = 0; x < 4; ++x) wh_fdct4 (t[x], t[4 + x], t[8 + x], t[12 + x], &d[x], &d[4 + x], &d[8 + x], &d[12 + x]);
}
// quant (mode 0: pfQuantization4x4, 1: ...Four4x4Max per block), then scan / score / count on the result
__global__ void k_quant (int n, int16_t* io, const uint8_t* qp, int intra, int16_t* maxv, int16_t* zz, int16_t* zz_ac, int* ctr, int* nzc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int16_t* d = io + i * 16;
  const int q = qp[i];
  int16_t mx = 0;
  for (int k = 0; k < 16; ++k) { int16_t a; d[k] = wh_quant1_abs (d[k], kWhQuantFF[(q + (intra ? 6 : 0)) * 3 + WH_POSCLASS (k)], wh_mf (q, k), &a); if (mx < a) mx = a; }
  maxv[i] = mx;
  int16_t lv[16];
  int cnt = 0;
  for (int k = 0; k < 16; ++k) { lv[k] = d[wh_zigzag (k)]; zz[i * 16 + k] = lv[k]; zz_ac[i * 16 + k] = k < 15 ? d[wh_zigzag (k + 1)] : (int16_t)0; cnt += lv[k] != 0; }
  ctr[i] = wh_single_ctr (lv);
  nzc[i] = cnt;
}
__global__ void k_dequant_idct (int n, const int16_t* coef, const uint8_t* qp, const uint8_t* pred, uint8_t* rec, int16_t* deq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int16_t c[16], t[16];
  for (int k = 0; k < 16; ++k) { c[k] = (int16_t) (coef[i * 16 + k] * wh_dq (qp[i], k)); deq[i * 16 + k] = c[k]; }
  for (int y = 0; y < 4; ++y) wh_idct4_h (c[y * 4], c[y * 4 + 1], c[y * 4 + 2], c[y * 4 + 3], &t[y * 4], &t[y * 4 + 1], &t[y * 4 + 2], &t[y * 4 + 3]);
  for (int x = 0; x < 4; ++x) {
    int r0, r1, r2, r3;
    wh_idct4_v (t[x], t[4 + x], t[8 + x], t[12 + x], &r0, &r1, &r2, &r3);
    const uint8_t* p = pred + i * 16;
    uint8_t* o = rec + i * 16;
    o[x] = wh_clip255 (p[x] + r0); o[4 + x] = wh_clip255 (p[4 + x] + r1); o[8 + x] = wh_clip255 (p[8 + x] + r2); o[12 + x] = wh_clip255 (p[12 + x] + r3);
  }
}
// Intra4x4: standard modes 0..8 plus the DC flavours selected by `avail` (bit0 left, bit1 top)
__global__ void k_pred4 (int n, const uint8_t* plane, int st, const int* off, const uint8_t* mode, const uint8_t* avail, uint8_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* ref = plane + off[i];
  uint8_t Eb[16] = {0};
  for (int k = 0; k < 4; ++k) Eb[3 - k] = ref[k * st - 1];
  Eb[4] = ref[-st - 1];
  for (int k = 0; k < 8; ++k) Eb[5 + k] = ref[-st + k];
  WhE13 E;
  for (int k = 0; k < 4; ++k) E.w[k] = (uint32_t)Eb[4 * k] | ((uint32_t)Eb[4 * k + 1] << 8) | ((uint32_t)Eb[4 * k + 2] << 16) | ((uint32_t)Eb[4 * k + 3] << 24);
  const bool l = avail[i] & 1, t = avail[i] & 2;
  int dc = 128;
  if (l && t) dc = (Eb[0] + Eb[1] + Eb[2] + Eb[3] + Eb[5] + Eb[6] + Eb[7] + Eb[8] + 4) >> 3;
  else if (l) dc = (Eb[0] + Eb[1] + Eb[2] + Eb[3] + 2) >> 2;
  else if (t) dc = (Eb[5] + Eb[6] + Eb[7] + Eb[8] + 2) >> 2;
  for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) out[i * 16 + y * 4 + x] = (uint8_t)wh_pred4_px (mode[i], x, y, E, dc);
}
// Intra16x16 / chroma 8x8 predictors through the macroblock tile code path (one wave per block)
__global__ __launch_bounds__ (64) void k_pred_mb (const uint8_t* plane_y, int st_y, const int* off_y, const uint8_t* plane_c, int st_c, const int* off_c,
                                                const uint8_t* mode16, const uint8_t* modec, uint8_t* out16, uint8_t* outc) {
  __shared__ WhMbLds S;
  const int i = blockIdx.x;
  const uint8_t* ry = plane_y + off_y[i];
  const uint8_t* rc = plane_c + off_c[i];          // Cb; Cr taken from the same plane 16 columns to the right
  WV_LANES_BEGIN (lane)
  if (lane < 17) WH_RY (S, lane - 1, -1) = ry[-st_y + lane - 1];
  else if (lane < 33) WH_RY (S, -1, lane - 17) = ry[(lane - 17) * st_y - 1];
  else if (lane < 42) { WH_RC (S, 0, lane - 34, -1) = rc[-st_c + lane - 34]; WH_RC (S, 1, lane - 34, -1) = rc[-st_c + 16 + lane - 34]; }
  else if (lane < 50) { WH_RC (S, 0, -1, lane - 42) = rc[(lane - 42) * st_c - 1]; WH_RC (S, 1, -1, lane - 42) = rc[(lane - 42) * st_c + 16 - 1]; }
  WV_LANES_END
  int sum_t, sum_l, h, v;
  WV_SUM (sum_t, lane, (lane < 16 ? WH_RY (S, lane, -1) : 0));
  WV_SUM (sum_l, lane, (lane < 16 ? WH_RY (S, -1, lane) : 0));
  WV_SUM (h, lane, (lane < 8 ? (lane + 1) * (WH_RY (S, 8 + lane, -1) - WH_RY (S, 6 - lane, -1)) : 0));
  WV_SUM (v, lane, (lane < 8 ? (lane + 1) * (WH_RY (S, -1, 8 + lane) - WH_RY (S, -1, 6 - lane)) : 0));
  const int a = (WH_RY (S, -1, 15) + WH_RY (S, 15, -1)) << 4, b = (5 * h + 32) >> 6, c = (5 * v + 32) >> 6;
  wh_pred_i16 (S, mode16[i], sum_t, sum_l, b, c, a);
  int st[4], sl[4], pa[2], pb[2], pc[2];
  for (int pl = 0; pl < 2; ++pl) {
    WV_SUM (st[pl * 2], lane, (lane < 4 ? WH_RC (S, pl, lane, -1) : 0));
    WV_SUM (st[pl * 2 + 1], lane, (lane < 4 ? WH_RC (S, pl, 4 + lane, -1) : 0));
    WV_SUM (sl[pl * 2], lane, (lane < 4 ? WH_RC (S, pl, -1, lane) : 0));
    WV_SUM (sl[pl * 2 + 1], lane, (lane < 4 ? WH_RC (S, pl, -1, 4 + lane) : 0));
    int hh, vv;
    WV_SUM (hh, lane, (lane < 4 ? (lane + 1) * (WH_RC (S, pl, 4 + lane, -1) - WH_RC (S, pl, 2 - lane, -1)) : 0));
    WV_SUM (vv, lane, (lane < 4 ? (lane + 1) * (WH_RC (S, pl, -1, 4 + lane) - WH_RC (S, pl, -1, 2 - lane)) : 0));
    pa[pl] = (WH_RC (S, pl, -1, 7) + WH_RC (S, pl, 7, -1)) << 4; pb[pl] = (17 * hh + 16) >> 5; pc[pl] = (17 * vv + 16) >> 5;
  }
  wh_pred_chroma (S, modec[i], st, sl, pb, pc, pa);
  WV_LANES_BEGIN (lane)
  for (int k = 0; k < 4; ++k) out16[i * 256 + lane * 4 + k] = S.pred_y[lane * 4 + k];
  for (int k = 0; k < 2; ++k) outc[i * 128 + lane * 2 + k] = S.pred_c[lane * 2 + k];
  WV_LANES_END
}
__global__ void k_mc (int n, const uint8_t* plane, int st, const int* off, const int16_t* mv, int w, int h, int chroma, uint8_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* src = plane + off[i];
  const int mvx = mv[i * 2], mvy = mv[i * 2 + 1];
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x)
    out[(size_t)i * w * h + y * w + x] = chroma ? (uint8_t)wh_mc_chroma_px (src + y * st + x, st, mvx & 7, mvy & 7) : (uint8_t)wh_mc_luma_px (src + y * st + x, st, mvx & 3, mvy & 3);
}
// edge filters: one thread per line; bs per line (0..4), filter parameters from the edge QP index
__global__ void k_deblock (int n_edges, uint8_t* plane, int st, const int* off, int horizontal, int chroma, const uint8_t* bs, const uint8_t* qp_index) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lines = chroma ? 8 : 16;
  if (i >= n_edges * lines) return;
  const int e = i / lines, l = i % lines;
  uint8_t* q = plane + off[e] + (horizontal ? l * st : l);
  const int ia = qp_index[e];
  const int b = bs[e * 4 + (chroma ? l >> 1 : l >> 2)];
  if (!(kWhAlpha[ia] | kWhBeta[ia])) return;
  if (chroma) wh_db_chroma_line (q, horizontal ? 1 : st, b, kWhAlpha[ia], kWhBeta[ia], ia);
  else wh_db_luma_line (q, horizontal ? 1 : st, b, kWhAlpha[ia], kWhBeta[ia], ia);
}
__global__ void k_vaa (int n, const uint8_t* cur, const uint8_t* ref, int st, const int* off, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 4) return;
  const int m = i >> 2, k = i & 3;
  const uint8_t* c = cur + off[m] + (k >> 1) * 8 * st + (k & 1) * 8;
  const uint8_t* r = ref + off[m] + (k >> 1) * 8 * st + (k & 1) * 8;
  int s = 0;
  for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) s += wh_abs (c[y * st + x] - r[y * st + x]);
  out[i] = s;
}

