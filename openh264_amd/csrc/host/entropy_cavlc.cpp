// entropy_cavlc.cpp -- see entropy_cavlc.h
#include "entropy_cavlc.h"
#include <stddef.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "../common/compact.h"
#define WH_TABLE static const
#include "../common/h264_tables.h"

namespace wh {
namespace {

// luma4x4BlkIdx -> raster index of the 4x4 block inside the MB
inline int blk_raster (int b) { return (((b >> 1) & 1) | ((b >> 2) & 2)) * 4 + ((b & 1) | ((b >> 1) & 2)); }

// Bit i set = lv[i] != 0, for the 16 levels of a block (the callers mask it down to the entries their block type has).
inline uint32_t nonzero_mask16 (const int16_t* lv) {
#if defined(__SSE2__)
  const __m128i z = _mm_setzero_si128();
  const __m128i a = _mm_loadu_si128 ((const __m128i*)lv), b = _mm_loadu_si128 ((const __m128i*) (lv + 8));
  return ~(uint32_t)_mm_movemask_epi8 (_mm_packs_epi16 (_mm_cmpeq_epi16 (a, z), _mm_cmpeq_epi16 (b, z))) & 0xffffu;
#else
  uint32_t m = 0;
  for (int i = 0; i < 16; ++i) m |= (uint32_t) (lv[i] != 0) << i;
  return m;
#endif
}

// residual_block_cavlc (7.3.5.3.2 / 9.2).  lv[0..end_idx] zig-zag levels (nullptr: nothing but zeros); nc: 0..16, or 17 for
// ChromaDCLevel.  Returns -1 when a level needs an escape longer than Baseline allows (set_mb_syn_cavlc.cpp:181-184).
int write_block (BitWriter& bw, const int16_t* lv, int end_idx, int nc, bool has_coeff = true) {
  // has_coeff mirrors iCalRunLevelFlag: a block whose total_coeff (nzc) is 0 is written as empty without looking at
  // its level buffer, which may be stale after the encoder's zeroing heuristics (svc_encode_mb.cpp:283-287)
  // Run / level pairs in reverse scan order (CavlcParamCal_c), from the block's non-zero mask instead of a scan per level:
  // run[k] = zeros between level k and the next lower one, the last one's = its own index.
  int16_t level[16];
  uint8_t run[16];
  int total = 0, total_zeros = 0;
  if (has_coeff && lv) {
    uint32_t nz;
    if (end_idx >= 14) nz = nonzero_mask16 (lv) & ((2u << end_idx) - 1u);          // 15 or 16 levels: a 32-byte block
    else { nz = 0; for (int i = 0; i <= end_idx; ++i) nz |= (uint32_t) (lv[i] != 0) << i; }
    if (nz) {
      int i = 31 - __builtin_clz (nz);
      total_zeros = i + 1;
      for (;;) {
        level[total] = lv[i];
        nz &= ~(1u << i);
        if (!nz) { run[total++] = (uint8_t)i; break; }
        const int j = 31 - __builtin_clz (nz);
        run[total++] = (uint8_t) (i - j - 1);
        i = j;
      }
      total_zeros -= total;
    }
  }
  int t1 = 0;
  uint32_t signs = 0;
  for (int k = 0; k < (total > 3 ? 3 : total); ++k) {
    if (level[k] == 1 || level[k] == -1) { ++t1; signs = (signs << 1) | (level[k] < 0 ? 1u : 0u); }
    else break;
  }
  const uint16_t tok = kWhCoeffToken[(kWhNcClass[nc] * 17 + total) * 4 + t1];
  if (total == 0) { bw.put (tok >> 8, tok & 0xff); return 0; }
  bw.put ((tok >> 8) + t1, ((uint32_t) (tok & 0xff) << t1) | signs);

  int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
  for (int k = t1; k < total; ++k) {
    const int val = level[k];
    int code = (val > 0) ? 2 * val - 2 : -2 * val - 1;            // levelCode
    if (k == t1 && t1 < 3) code -= 2;
    int prefix = code >> suffix_len;
    int suffix_size = suffix_len;
    int suffix = code - (prefix << suffix_len);
    if (prefix >= 14 && prefix < 30 && suffix_len == 0) {
      prefix = 14; suffix = code - 14; suffix_size = 4;
    } else if (prefix >= 15) {
      prefix = 15;
      suffix = code - (15 << suffix_len);
      if (suffix >> 11) return -1;
      if (suffix_len == 0) suffix -= 15;
      suffix_size = 12;
    }
    bw.put (prefix + 1 + suffix_size, (1u << suffix_size) | (uint32_t)suffix);
    if (suffix_len == 0) suffix_len = 1;
    const int thr = 3 << (suffix_len - 1);
    if ((val > thr || val < -thr) && suffix_len < 6) ++suffix_len;
  }
  if (total < end_idx + 1) {
    const uint16_t tz = (nc != 17) ? kWhTotalZeros[total * 16 + total_zeros] : kWhTotalZerosChromaDc[total * 4 + total_zeros];
    bw.put (tz >> 8, tz & 0xff);
  }
  int zeros_left = total_zeros;
  for (int k = 0; k + 1 < total && zeros_left > 0; ++k) {
    const int zl = zeros_left > 7 ? 7 : zeros_left;
    const uint16_t rb = kWhRunBefore[zl * 15 + run[k]];
    bw.put (rb >> 8, rb & 0xff);
    zeros_left -= run[k];
  }
  return 0;
}

inline int nc_of (int na, int nb) {   // WELS_NON_ZERO_COUNT_AVERAGE (macros.h:135-139), -1 = unavailable
  int nc = na + nb + 1;
  nc >>= (na != -1 && nb != -1);
  nc += (na == -1 && nb == -1);
  return nc;
}

struct NzcCtx {
  const uint8_t* cur; const uint8_t* left; const uint8_t* top;       // nzc[24] each; left / top nullptr = not available
  // luma: raster block index r (0..15)
  int luma_a (int r) const { return (r & 3) ? cur[r - 1] : (left ? left[r + 3] : -1); }
  int luma_b (int r) const { return (r >> 2) ? cur[r - 4] : (top ? top[r + 12] : -1); }
  // chroma plane p (0/1), raster 2x2 index c
  int chroma_a (int p, int c) const { return (c & 1) ? cur[16 + p * 4 + c - 1] : (left ? left[16 + p * 4 + c + 1] : -1); }
  int chroma_b (int p, int c) const { return (c >> 1) ? cur[16 + p * 4 + c - 2] : (top ? top[16 + p * 4 + c + 2] : -1); }
};

int write_residual (BitWriter& bw, const MbView& mb, const NzcCtx& n) {
  const WhMbRecord& r = *mb.side;
  const int cbp_l = r.cbp & 15, cbp_c = r.cbp >> 4;
  if (r.mb_type == WH_MB_I16x16) {
    if (write_block (bw, mb.block (16), 15, nc_of (n.luma_a (0), n.luma_b (0)))) return -1;
    if (cbp_l) {
      for (int b = 0; b < 16; ++b) {
        const int rr = blk_raster (b);
        if (write_block (bw, mb.block (b), 14, nc_of (n.luma_a (rr), n.luma_b (rr)), r.nzc[rr] > 0)) return -1;
      }
    }
  } else {
    for (int b = 0; b < 16; ++b) {
      if (!(cbp_l & (1 << (b >> 2)))) continue;
      const int rr = blk_raster (b);
      if (write_block (bw, mb.block (b), 15, nc_of (n.luma_a (rr), n.luma_b (rr)), r.nzc[rr] > 0)) return -1;
    }
  }
  if (cbp_c) {
    const int16_t* dc = mb.block (25);
    if (write_block (bw, dc, 3, 17)) return -1;
    if (write_block (bw, dc ? dc + 4 : nullptr, 3, 17)) return -1;
    if (cbp_c & 2) {
      for (int p = 0; p < 2; ++p)
        for (int c = 0; c < 4; ++c)
          if (write_block (bw, mb.block (17 + p * 4 + c), 14, nc_of (n.chroma_a (p, c), n.chroma_b (p, c)), r.nzc[16 + p * 4 + c] > 0)) return -1;
    }
  }
  return 0;
}

const uint8_t kZeroNzc[24] = {0};       // total_coeff of a packed P_Skip macroblock (its record carries none)

}  // namespace

MbView view_of_record (const WhMbRecord* recs, int mb_w, int xy, int avail) {
  MbView v;
  v.side = &recs[xy];
  v.blocks = &recs[xy].luma[0][0];
  if (avail & WH_AVAIL_LEFT) v.nzc_left = recs[xy - 1].nzc;
  if (avail & WH_AVAIL_TOP) v.nzc_top = recs[xy - mb_w].nzc;
  return v;
}

MbView view_of_packed (const uint8_t* packed, const uint32_t* off, int mb_w, int xy, int avail) {
  // (offsets and sizes are multiples of 4: the side information keeps the alignment its int32 members need)
  auto nzc_at = [&] (int k) { return off[k + 1] - off[k] == WH_COMPACT_SKIP_BYTES ? kZeroNzc : packed + off[k] + 4 + offsetof (WhMbRecord, nzc); };
  MbView v;
  v.packed = true;
  const uint8_t* p = packed + off[xy];
  if (off[xy + 1] - off[xy] == WH_COMPACT_SKIP_BYTES) {
    v.side = (const WhMbRecord*)p;               // P_Skip: the writer reads mb_type and nothing else
    v.mask = 0;
  } else {
    memcpy (&v.mask, p, 4);
    v.side = (const WhMbRecord*) (p + 4);
    v.blocks = (const int16_t*) (p + 4 + WH_COMPACT_SIDE);
  }
  if (avail & WH_AVAIL_LEFT) v.nzc_left = nzc_at (xy - 1);
  if (avail & WH_AVAIL_TOP) v.nzc_top = nzc_at (xy - mb_w);
  return v;
}

int write_mb_cavlc (BitWriter& bw, SliceEntropyState& st, const MbView& mb, int* qp_for_deblock) {
  const WhMbRecord& r = *mb.side;
  if (r.mb_type == WH_MB_PSKIP) {
    *qp_for_deblock = st.last_qp;
    ++st.skip_run;
    return 0;
  }
  const bool p_slice = st.slice_type == WH_SLICE_P;
  if (p_slice) { bw.ue ((uint32_t)st.skip_run); st.skip_run = 0; }
  const int off = p_slice ? 5 : 0;
  switch (r.mb_type) {
  case WH_MB_I4x4:
    bw.ue (off + 0);
    for (int b = 0; b < 16; ++b) {
      const int prev = (r.i4_prev_flags >> b) & 1;
      bw.bit (prev);
      if (!prev) bw.put (3, (uint32_t)r.i4_rem[b]);
    }
    bw.ue (r.chroma_mode);
    break;
  case WH_MB_I16x16:
    bw.ue (1 + off + r.i16_mode + ((r.cbp >> 4) << 2) + ((r.cbp & 15) ? 12 : 0));
    bw.ue (r.chroma_mode);
    break;
  case WH_MB_P16x16:
    bw.ue (0);
    if (st.num_ref_idx_l0_active_minus1 > 0) bw.te (st.num_ref_idx_l0_active_minus1, (uint32_t)r.ref_idx[0]);
    bw.se (r.mvd[0][0]); bw.se (r.mvd[0][1]);
    break;
  case WH_MB_P16x8:
    bw.ue (1);
    if (st.num_ref_idx_l0_active_minus1 > 0) { bw.te (st.num_ref_idx_l0_active_minus1, (uint32_t)r.ref_idx[0]); bw.te (st.num_ref_idx_l0_active_minus1, (uint32_t)r.ref_idx[2]); }
    bw.se (r.mvd[0][0]); bw.se (r.mvd[0][1]);
    bw.se (r.mvd[8][0]); bw.se (r.mvd[8][1]);
    break;
  case WH_MB_P8x16:
    bw.ue (2);
    if (st.num_ref_idx_l0_active_minus1 > 0) { bw.te (st.num_ref_idx_l0_active_minus1, (uint32_t)r.ref_idx[0]); bw.te (st.num_ref_idx_l0_active_minus1, (uint32_t)r.ref_idx[1]); }
    bw.se (r.mvd[0][0]); bw.se (r.mvd[0][1]);
    bw.se (r.mvd[2][0]); bw.se (r.mvd[2][1]);
    break;
  case WH_MB_P8x8: {
    const bool all_ref0 = r.ref_idx[0] == 0 && r.ref_idx[1] == 0 && r.ref_idx[2] == 0 && r.ref_idx[3] == 0;
    bw.ue (all_ref0 ? 4 : 3);                    // P_8x8ref0 when every ref_idx is 0
    for (int k = 0; k < 4; ++k) bw.ue (r.sub_type[k]);
    if (st.num_ref_idx_l0_active_minus1 > 0 && !all_ref0)
      for (int k = 0; k < 4; ++k) bw.te (st.num_ref_idx_l0_active_minus1, (uint32_t)r.ref_idx[k]);
    for (int k = 0; k < 4; ++k) {
      const int base = (k >> 1) * 8 + (k & 1) * 2;   // raster index of the 8x8's first 4x4
      static const int kOffs[4][4] = {{0, -1, -1, -1}, {0, 4, -1, -1}, {0, 1, -1, -1}, {0, 1, 4, 5}};
      for (int j = 0; j < 4; ++j) {
        const int o = kOffs[r.sub_type[k]][j];
        if (o < 0) break;
        bw.se (r.mvd[base + o][0]); bw.se (r.mvd[base + o][1]);
      }
    }
    break;
  }
  default:
    return -2;
  }
  if (r.mb_type == WH_MB_I4x4) bw.ue (kWhCbpCodeIntra[r.cbp]);
  else if (r.mb_type != WH_MB_I16x16) bw.ue (kWhCbpCodeInter[r.cbp]);

  if (r.cbp > 0 || r.mb_type == WH_MB_I16x16) {
    bw.se (r.luma_qp - st.last_qp);
    st.last_qp = r.luma_qp;
    *qp_for_deblock = r.luma_qp;
    NzcCtx n;
    n.cur = r.nzc; n.left = mb.nzc_left; n.top = mb.nzc_top;
    if (write_residual (bw, mb, n)) return -1;
  } else {
    *qp_for_deblock = st.last_qp;
  }
  return 0;
}

void write_slice_end (BitWriter& bw, SliceEntropyState& st) {
  if (st.slice_type == WH_SLICE_P && st.skip_run > 0) { bw.ue ((uint32_t)st.skip_run); st.skip_run = 0; }
  bw.trailing();
}

}  // namespace wh
