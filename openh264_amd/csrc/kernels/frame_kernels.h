// frame_kernels.h -- per-macroblock kernel bodies of I pictures (one wavefront = one MB) and the store of a
// macroblock's results.  The __global__ wrappers and the in-kernel scheduling live in hip/hip_backend.hip; the CPU-side
// test build (tests/emu) calls the same bodies in the same dependency order (common/mb_order.h).
//
// Restructures (does not port) the reference's MB loops:
//   svc_encode_slice.cpp:534-599   WelsISliceMdEnc   (I slices)
//   svc_encode_slice.cpp:1807-1899 WelsMdInterMbLoop (P slices, see inter_mb.h)
#pragma once
#include <stddef.h>
#include "intra_mb.h"
#include "cavlc_bits.h"
#include "../common/gom_rc.h"

static_assert (sizeof (WhMbRecord) == 960, "WhMbRecord must be 960 bytes");
static_assert (sizeof (WhMbState) == 144, "WhMbState must be 144 bytes");
static_assert (sizeof (WhMbCtl) == 8, "WhMbCtl must be 8 bytes");
// (wh_store_mb and the P body write these groups of byte / short fields as one 32-bit word each)
static_assert (offsetof (WhMbRecord, mb_type) == 0 && offsetof (WhMbRecord, i16_mode) == 4 && offsetof (WhMbRecord, i4_prev_flags) == 6 && offsetof (WhMbRecord, bgd_skip) % 4 == 0 &&
               offsetof (WhMbRecord, sub_type) % 4 == 0 && offsetof (WhMbRecord, ref_idx) % 4 == 0 && offsetof (WhMbRecord, mvd) % 4 == 0 && offsetof (WhMbRecord, mv_tr) % 4 == 0, "WhMbRecord layout");
static_assert (offsetof (WhMbState, mb_type) == 0 && offsetof (WhMbState, luma_qp) == 1 && offsetof (WhMbState, chroma_qp) == 2 && offsetof (WhMbState, cbp) == 3 &&
               offsetof (WhMbState, slice_idc) == 4 && offsetof (WhMbState, ref_type) == 6 && offsetof (WhMbState, ref_qp) == 7 && offsetof (WhMbState, mv) % 4 == 0 &&
               offsetof (WhMbState, ref_idx) % 4 == 0 && offsetof (WhMbState, p16mv) % 4 == 0 && offsetof (WhMbState, i4_mode) == 8, "WhMbState layout");

// Store the MB's reconstruction, entropy record and neighbour state to HBM.
// X: the slice is coded by several workgroups (wave.h wh_st_x): state and unfiltered samples go through to memory; the record is the host's
template <bool X = false>
WH_FN void wh_store_mb (WhMbLds& S, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby, int mb_type, int cbp,
                        int qp, int qpc, int i16_mode, int chroma_mode, int cost, int slice_idc, int cavlc_bits = 0) {
  const int xy = mby * P.mb_w + mbx;
  // explicit global address space: generic (flat) stores would also count against lgkmcnt and make every later LDS read
  // of the wave wait for them
  WH_G WhMbRecord* R = (WH_G WhMbRecord*)J.records + xy;
  WH_G WhMbState* M = (WH_G WhMbState*)J.mbs + xy;
  WV_LANES_BEGIN (lane)
  if (J.rec_blk) {              // (wave-uniform) the macroblock's own 384 bytes: the lane's luma word is word `lane`, its chroma word 64 + lane
    WH_G uint32_t* b = (WH_G uint32_t*) ((WH_G uint8_t*)J.rec_blk + (size_t)xy * WH_SRC_MB_BYTES);
    wh_st_x<X> (&b[lane], * (const uint32_t*)&WH_RY (S, (lane & 3) * 4, lane >> 2));
    if (lane < 32) wh_st_x<X> (&b[64 + lane], * (const uint32_t*)&WH_RC (S, lane >> 4, (lane & 1) * 4, (lane >> 1) & 7));
  } else {
    {
      const int row = lane >> 2, seg = lane & 3;
      WH_G uint8_t* d = (WH_G uint8_t*)J.rec[0] + (size_t) (mby * 16 + row) * P.rec_stride_y + mbx * 16 + seg * 4;
      * (WH_G uint32_t*)d = * (const uint32_t*)&WH_RY (S, seg * 4, row);
    }
    if (lane < 32) {
      const int pl = lane >> 4, row = (lane >> 1) & 7, half = lane & 1;
      WH_G uint8_t* d = (WH_G uint8_t*) (pl ? J.rec[2] : J.rec[1]) + (size_t) (mby * 8 + row) * P.rec_stride_c + mbx * 8 + half * 4;     // (a select, not an index: the job may live in registers)
      * (WH_G uint32_t*)d = * (const uint32_t*)&WH_RC (S, pl, half * 4, row);
    }
  }
  // coefficient levels: 4 luma + 2 chroma-AC int16 quads per lane
  {
    const uint64_t* s = (const uint64_t*)S.lv_luma;
    WH_G uint64_t* d = (WH_G uint64_t*)&R->luma[0][0];
    d[lane] = s[lane];
  }
  if (lane < 32) {
    const uint64_t* s = (const uint64_t*)S.lv_cac;
    WH_G uint64_t* d = (WH_G uint64_t*)&R->chroma_ac[0][0];
    d[lane] = s[lane];
  } else if (lane < 48) {
    R->luma_dc[lane - 32] = (mb_type == WH_MB_I16x16) ? S.lv_dc[lane - 32] : (int16_t)0;
  } else if (lane < 56) {
    ((WH_G int16_t*)&R->chroma_dc[0][0])[lane - 48] = S.lv_cdc[lane - 48];
  }
  if (lane < 24) { R->nzc[lane] = S.nzc[lane]; wh_st_x<X> (&M->nzc[lane], S.nzc[lane]); }
  if (lane < 16) {
    R->i4_rem[lane] = (mb_type == WH_MB_I4x4) ? S.i4_rem[lane] : (int8_t)0;
    wh_st_x<X> (&M->i4_mode[lane], (mb_type == WH_MB_I4x4) ? S.i4m[((lane >> 2) + 1) * 5 + (lane & 3) + 1] : (int8_t)2);
  }
  if (lane == 0) {
    // (the header fields word by word: field by field they were fifteen byte / short stores with an address each)
    const uint32_t i4p = (mb_type == WH_MB_I4x4) ? (uint32_t)S.i4_prev : 0u;
    * (WH_G uint32_t*)&R->mb_type = (uint32_t) (mb_type & 255) | ((uint32_t) (cbp & 255) << 8) | ((uint32_t) (qp & 255) << 16) | ((uint32_t) (qpc & 255) << 24);
    * (WH_G uint32_t*)&R->i16_mode = (uint32_t) (i16_mode & 255) | ((uint32_t) (chroma_mode & 255) << 8) | (i4p << 16);
    R->cost = cost;
    R->cavlc_bits = cavlc_bits;
    * (WH_G uint32_t*)&R->bgd_skip = 0u;                    // bgd_skip + pad0
    wh_st_x<X> ((WH_G uint32_t*)&M->mb_type, (uint32_t) (mb_type & 255) | ((uint32_t) (qp & 255) << 8) | ((uint32_t) (qpc & 255) << 16) | ((uint32_t) (cbp & 255) << 24));
    // uiRefMbType / pRefMbQp of the picture (read when it is a reference): P pictures store the type, I pictures leave the
    // previous contents of the picture buffer alone (WelsMdInterSaveSadAndRefMbType is a P-slice step); both store the QP
    // (WelsMdUpdateBGDInfo; wh_inter_mb_body overrides it for unchanged collocated macroblocks)
    if (J.slice_type == WH_SLICE_P) wh_st_x<X> ((WH_G uint32_t*)&M->slice_idc, (uint32_t) (slice_idc & 0xffff) | ((uint32_t) ((mb_type + 1) & 255) << 16) | ((uint32_t) (qp & 255) << 24));
    else { wh_st_x<X> (&M->slice_idc, (uint16_t)slice_idc); wh_st_x<X> (&M->ref_qp, (uint8_t)qp); }
  }
  WV_LANES_END
}

// QP of one macroblock: the picture QP, plus the per-MB offset when the host supplies a map (WelsRcMbInitDisable,
// ratectl.cpp, and UpdateQpForOverflow, svc_encode_slice.cpp:526-529: the re-encode after a CAVLC level overflow).
WH_FN WhMbCtl wh_mb_ctl (const WhPicJob& J, int xy) {
  WhMbCtl c;
  uint64_t w = 0;
  if (J.mb_ctl) w = * (const WH_G uint64_t*) ((const WH_G WhMbCtl*)J.mb_ctl + xy);
  c.qp_delta = (int8_t) (w & 0xff); c.stale_cbp = (uint8_t) ((w >> 8) & 0x3f); c.cell12_valid = (uint8_t) ((w >> 16) & 1); c.pad = 0;
  c.cell12_mv[0] = (int16_t) (w >> 32); c.cell12_mv[1] = (int16_t) (w >> 48);
  return c;
}
WH_FN int wh_mb_qp (const WhPicJob& J, const WhMbCtl& c) { return wh_clip3 (J.qp + (int)c.qp_delta, 0, 51); }

// QP_Y as the decoder derives it, for the deblocking filter: a macroblock that codes no mb_qp_delta (P_Skip, or
// cbp == 0 and not Intra16x16) inherits the QP of the previous macroblock of its slice, the first one the slice QP
// (svc_set_mb_syn_cavlc.cpp:232-247,288-300: uiLumaQp = uiLastMbQp).  With one QP per picture that is the identity, so
// this pass only runs for pictures with a per-MB QP map.  One wavefront per slice: 64 states per step are loaded
// lane-parallel, the chain itself is a wave-uniform scan over the lane table.  Size-limited slices (WhPicJob::dyn_slice): [first, last) is a
// partition of the picture and the slices inside it are where the states' slice index changes.
WH_FN void wh_qp_chain_slice (const WhSeqParams& P, const WhPicJob& J, int first, int last) {
  int carry = wh_clip3 (J.qp, 0, 51);
  int cur_slice = -1;
  for (int base = first; base < last; base += 64) {
    WvLaneArr w, o, sl;
#ifdef WH_EMU
    memset (&w, 0, sizeof (w)); memset (&o, 0, sizeof (o)); memset (&sl, 0, sizeof (sl));
#else
    w = 0; o = 0; sl = 0;
#endif
    WV_LSET_IF (w, lane, base + lane < last, (int) * (const WH_G uint32_t*) ((const WH_G WhMbState*)J.mbs + base + lane));
    if (J.dyn_slice) WV_LSET_IF (sl, lane, base + lane < last, (int) ((const WH_G WhMbState*)J.mbs + base + lane)->slice_idc);
    const int n = last - base < 64 ? last - base : 64;
    for (int i = 0; i < n; ++i) {
      const uint32_t v = (uint32_t)WV_LGET (w, i);            // bytes: mb_type, luma_qp, chroma_qp, cbp
      const int type = (int) (v & 0xff), cbp = (int) (v >> 24);
      if (J.dyn_slice) { const int s = WV_LGET (sl, i); if (s != cur_slice) { cur_slice = s; carry = wh_clip3 (J.qp, 0, 51); } }
      if (! (type == WH_MB_PSKIP || (cbp == 0 && type != WH_MB_I16x16))) carry = (int) ((v >> 8) & 0xff);
      WV_LSET (o, i, carry);
    }
    WV_LANES_BEGIN (lane)
    if (base + lane < last) {
      WH_G WhMbState* M = (WH_G WhMbState*)J.mbs + base + lane;
      const int q = WV_LOWN (o, lane);
      M->luma_qp = (uint8_t)q;
      M->chroma_qp = (uint8_t)kWhChromaQp[wh_clip3 (q + P.chroma_qp_offset, 0, 51)];
      if (M->ref_qp == 0xff) M->ref_qp = (uint8_t)q;          // pRefMbQp of a skip decided in mode decision: the last coded QP (inter_mb.h)
    }
    WV_LANES_END
  }
}

// One intra macroblock (I slice).  GOM: the picture may be coded with GOM-level rate control inside the kernel (WH_SEQ_CHAIN launches); a
// compile-time switch because the all-IDR kernel is register-bound and the rate-control path cost it 40 % when it was a run-time test
// (IDR step of 512 720p pictures 17.4 -> 24.4 ms, profiles/r05_intra_kernel_gom_split_ab.txt).
template <bool GOM = true>
WH_FN void wh_intra_mb_body (WhMbLds& S, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  const int xy = mby * P.mb_w + mbx;
  // (size-limited slices: the launch codes ONE slice, which begins at dyn_first -- WhPicJob::dyn_slice)
  const int avail = J.dyn_slice ? wh_mb_avail_in_slice (P, mbx, mby, J.dyn_first) : wh_mb_avail (P, mbx, mby);
  const WhMbCtl ctl = wh_mb_ctl (J, xy);
  // GOM-level rate control inside the kernel (I pictures too since round 5): the QP of this macroblock's group, settled by the last macroblock
  // of the group before it (wh_gom_close_if_last, inter_mb.h), which the scheduler has waited for (WhPicJob::scc_chain_prev)
  const int qp = (GOM && J.gom_rc) ? wh_clip3 ((int)wh_ld_wg32 ((const WH_G uint32_t*)& ((const WH_G WhGomRc*)J.gom_rc)->calc_qp), 0, 51) : wh_mb_qp (J, ctl);
  const int qpc = kWhChromaQp[wh_clip3 (qp + P.chroma_qp_offset, 0, 51)];
  WH_PROF_DECL (P);
  wh_load_mb_tile (S, P, J, mbx, mby);
  WH_PROF_MARK (P, S, 9);    // source + neighbour samples
  WhIntraResult r;
  wh_intra_md_enc (S, P, J, mbx, mby, avail, qp, qpc, &r, ctl.stale_cbp);
  WH_PROF_MARK (P, S, 15);   // (what the decision's own marks leave: its epilogue)
  // intra MBs carry no motion: clear mv/ref so that later P pictures / deblocking see zeros
  WV_LANES_BEGIN (lane)
  WH_G WhMbState* Ms = (WH_G WhMbState*)J.mbs + xy;
  WH_G WhMbRecord* Rs = (WH_G WhMbRecord*)J.records + xy;
  if (lane < 16) { Ms->mv[lane][0] = 0; Ms->mv[lane][1] = 0; Rs->mvd[lane][0] = 0; Rs->mvd[lane][1] = 0; }
  if (lane < 4) { Ms->ref_idx[lane] = -1; Rs->ref_idx[lane] = -1; Rs->sub_type[lane] = 0; Ms->sad_cost[lane] = 0; }
  // pSadCost[0] = 0 (WelsMdIntraSecondaryModesEnc, svc_base_layer_md.cpp:2038) -- in the layer's array too, where a P_Skip of a later
  // picture that is not costed by SAD (complexity above LOW) finds it (found by tools/fuzz_screen.py: an I picture in mid-stream)
  if (lane == 0 && J.sad_cost0) (J.sad_cost0_out ? (WH_G int32_t*)J.sad_cost0_out : (WH_G int32_t*)J.sad_cost0)[xy] = 0;
  WV_LANES_END
  int bits = 0;
  if (J.want_bits) {
    // the neighbours' total_coeff counts (coeff_token contexts): into the reduction scratch of the tile, then counted (cavlc_bits.h)
    const bool al = (avail & WH_AV_LEFT) != 0, at = (avail & WH_AV_TOP) != 0;
    uint8_t* nl = (uint8_t*)S.part;
    uint8_t* nt = (uint8_t*)S.part2;
    WV_LANES_BEGIN (lane)
    if (lane < 24) {
      if (al) nl[lane] = ((const WH_G WhMbState*)J.mbs + xy - 1)->nzc[lane];
      if (at) nt[lane] = ((const WH_G WhMbState*)J.mbs + xy - P.mb_w)->nzc[lane];
    }
    WV_LANES_END
    bits = wh_mb_residual_bits (S, r.mb_type, r.cbp, al ? nl : nullptr, at ? nt : nullptr) +
           wh_mb_intra_header_bits (S, r.mb_type, r.cbp, r.i16_mode_std, r.chroma_mode_std, false);
    if (r.cbp > 0 || r.mb_type == WH_MB_I16x16) bits |= WH_BITS_HAS_QP_DELTA;
  }
  wh_store_mb (S, P, J, mbx, mby, r.mb_type, r.cbp, qp, qpc, r.i16_mode_std, r.chroma_mode_std, r.cost_luma, J.dyn_slice ? J.dyn_slice - 1 : wh_slice_of_mb (P, xy), bits);
  WH_PROF_MARK (P, S, 7);    // store
}

#include "../common/mb_order.h"
