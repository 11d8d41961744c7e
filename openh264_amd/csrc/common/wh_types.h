// wh_types.h -- data layout shared by the HIP kernels, the host encoder and the C ABI.
//
// Everything here is plain C so that it can be included from .hip, .cpp and (mirrored) ctypes.
// Names follow the H.264 / OpenH264 domain: macroblock (MB), slice, picture, level, nzc.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// ---- macroblock types (our own compact enum; the reference uses bit flags,
//      codec/common/inc/wels_common_defs.h:275-300) -------------------------------------------
enum {
  WH_MB_I4x4   = 0,
  WH_MB_I16x16 = 1,
  WH_MB_P16x16 = 2,
  WH_MB_P16x8  = 3,
  WH_MB_P8x16  = 4,
  WH_MB_P8x8   = 5,
  WH_MB_PSKIP  = 6,
  WH_MB_NONE   = 255
};
#define WH_IS_INTRA(t) ((t) <= WH_MB_I16x16)
#define WH_IS_INTER(t) ((t) >= WH_MB_P16x16 && (t) <= WH_MB_PSKIP)
#define WH_REFTYPE_BACKGROUND 64      /* WhMbState::ref_type of an MB_TYPE_BACKGROUND macroblock (coded as P_Skip) */

enum { WH_SLICE_P = 0, WH_SLICE_I = 2 };

// ---- what the device hands to the host entropy coder for one MB --------------------------------
// 960 bytes = 144 B side info + 816 B coefficient levels; the same split the reference keeps in
// SMB + SMbCache + SDCTCoeff (svc_enc_macroblock.h:49-78, mb_cache.h:62-70,106-109).
typedef struct WhMbRecord {
  uint8_t  mb_type;          // WH_MB_*
  uint8_t  cbp;              // coded_block_pattern: luma bits 0..3, chroma (0/1/2) << 4
  uint8_t  luma_qp;          // QP this MB was quantised with
  uint8_t  chroma_qp;
  uint8_t  i16_mode;         // Intra16x16PredMode 0..3 (V,H,DC,Plane)
  uint8_t  chroma_mode;      // intra_chroma_pred_mode 0..3 (DC,H,V,Plane)
  uint16_t i4_prev_flags;    // bit b = prev_intra4x4_pred_mode_flag of luma4x4BlkIdx b
  int8_t   i4_rem[16];       // rem_intra4x4_pred_mode per luma4x4BlkIdx
  uint8_t  sub_type[4];      // sub_mb_type per 8x8 (0:8x8 1:8x4 2:4x8 3:4x4)
  int8_t   ref_idx[4];
  int16_t  mvd[16][2];       // mvd_l0 per 4x4 block in raster order (x,y), quarter-pel
  uint8_t  nzc[24];          // total_coeff: luma raster 0..15, Cb raster 16..19, Cr raster 20..23
  int32_t  cost;             // mode-decision cost of the chosen mode (for rate control)
  int16_t  mv_tr[2];         // inter MBs: final motion vector of the top-right 4x4 block (see WhMbCtl::cell12_mv)
  uint8_t  bgd_skip;         // P_Skip decided by background detection (MB_TYPE_BACKGROUND): the host mirrors VaaBackgroundMbDataUpdate
  uint8_t  pad0[3];
  int32_t  cavlc_bits;       // WhPicJob::want_bits: bits of this macroblock's CAVLC syntax without ue(mb_skip_run) and se(mb_qp_delta)
                             //   (kernels/cavlc_bits.h); | WH_BITS_HAS_QP_DELTA when it codes mb_qp_delta.  Else 0
  uint32_t fme_down;         // screen content with size-limited slices: what this macroblock adds to pSlice->uiSliceFMECostDown (else 0)
  uint8_t  pad[4];
  // coefficient levels in zig-zag order:
  int16_t  luma[16][16];     // per luma4x4BlkIdx; I16x16: entries 0..14 = AC, [15] = 0
  int16_t  luma_dc[16];      // Intra16x16 DC levels
  int16_t  chroma_ac[8][16]; // Cb blkIdx 0..3, Cr 4..7; entries 0..14 = AC
  int16_t  chroma_dc[2][4];
} WhMbRecord;

// ---- per-MB state that lives in HBM across MBs / pictures (neighbour + next-frame context) ----
typedef struct WhMbState {
  uint8_t  mb_type;
  uint8_t  luma_qp;          // QP seen by the deblocking filter (after the "last coded QP" rule)
  uint8_t  chroma_qp;
  uint8_t  cbp;
  uint16_t slice_idc;
  uint8_t  ref_type;         // uiRefMbType of the picture (picture.h:81): WH_MB_* + 1 as P pictures leave it (WH_REFTYPE_BACKGROUND for a
                             //   background-skip MB), 0 = never written; I pictures do not touch it (WelsMdInterSaveSadAndRefMbType)
  uint8_t  ref_qp;           // pRefMbQp of the picture (WelsMdUpdateBGDInfo, svc_mode_decision.cpp:267-282)
  int8_t   i4_mode[16];      // Intra4x4PredMode per 4x4 block, raster; 2 (DC) for non-I4x4 MBs
  uint8_t  nzc[24];          // as in WhMbRecord
  int16_t  mv[16][2];        // raster 4x4, quarter-pel
  int8_t   ref_idx[4];
  int32_t  sad_cost[4];      // pSadCost (md.cpp:826-910 PredictSad)
  int16_t  p16mv[2];         // sP16x16Mv
  int32_t  skip_sad;         // pMbSkipSad of the picture
  uint8_t  pad1[4];
} WhMbState;                 // 144 bytes

// ---- optional per-MB control word (WhPicJob::mb_ctl) ---------------------------------------------
// All zero for a macroblock that is encoded for the first time.  The reference re-encodes a macroblock at QP+2 after a
// CAVLC overflow WITHOUT re-initialising its per-MB state (TRY_REENCODING, svc_encode_slice.cpp:564-576,1845-1867), so
// three things leak from the previous pass into the next one and are reproduced through this word:
typedef struct WhMbCtl {
  int8_t   qp_delta;         // QP of this pass minus the picture QP
  uint8_t  stale_cbp;        // uiCbp left by the previous pass (only cleared in WelsMdIntraInit: an Intra4x4 result ORs onto it)
  uint8_t  cell12_valid;     // a previous pass ended as P8x16: update_P8x16_motion_info (mv_pred.cpp:235-276) then wrote the
  uint8_t  pad;              //   second partition's vector over MV-cache cell 12 (left neighbour, 2nd 4x4 row) with ref 0,
  int16_t  cell12_mv[2];     //   which the 16x8 lower-partition predictor reads as its top-left neighbour
} WhMbCtl;

// ---- screen-content inputs of one P picture (iUsageType == SCREEN_CONTENT_REAL_TIME) ---------------------------------
// What the reference's pre-processing (scene-change / scroll detection) and PreprocessSliceCoding (encoder_ext.cpp:2700-2765)
// hand to mode decision and motion estimation: svc_mode_decision.cpp:293-667, svc_motion_estimate.cpp:380-1097.
typedef struct WhSccJob {
  const uint8_t* static_idc;   // pVaaExt->pVaaBestBlockStaticIdc: one EStaticBlockIdc per 8x8 luma block, [2 * mb_h][2 * mb_w]
                               //   (0 moving, 1 static against the co-located block, 2 static against the scrolled one)
  const uint8_t* ref_ori_c[2]; // chroma planes of pCurDqLayer->pRefOri[0], the SOURCE picture of the reference (stride = src_stride_c)
  int32_t scroll_flag;         // sScrollDetectInfo.bScrollDetectFlag
  int32_t scroll_mvx, scroll_mvy;   // sScrollDetectInfo.iScrollMvX / iScrollMvY (integer samples)
  uint32_t thr16, thr8;        // uiSadCostThreshold[BLOCK_16x16] / [BLOCK_8x8] of the reference picture's SScreenBlockFeatureStorage
  int32_t fme;                 // pfSearchMethod[BLOCK_8x8] == WelsDiamondCrossFeatureSearch for this picture
  const uint32_t* fme_times;   // pTimesOfFeatureValue[fme_list_size]: how many 8x8 blocks of the reference picture have that sample sum
  const uint32_t* fme_start;   // first entry of that sum in fme_loc (in entries)
  const uint16_t* fme_loc;     // pLocationPointer: {x << 2, y << 2} of every block, grouped by sum, raster order inside a group
  int32_t fme_list_size;
  int32_t scd_on;              // pfSCDPSkipDecision == WelsMdInterJudgeSCDPskip (encoder.cpp:205-208: not with HIGH complexity); when it is off
                               //   the macroblock never looks at static_idc either -- SetBlockStaticIdcToMd is part of that function
  uint32_t* chain;             // [num_slices][4]: uiSadCost the slice's SWelsMD keeps in sMe8x8[i] from one macroblock to the next
                               //   (CheckDirectionalMv compares against it BEFORE the search overwrites it, svc_motion_estimate.cpp:385-402)
  uint32_t* fme_cost_down;     // [num_slices]: what the picture adds to pSlice->uiSliceFMECostDown
  uint32_t* chain_mb;          // size-limited slices (WhPicJob::dyn_slice): the same chain kept PER MACROBLOCK, [mb][4] -- what macroblock mb leaves
                               //   for the next one of its slice that may search 8x8 blocks (WhPicJob::scc_chain_prev); a slice can then begin,
                               //   or a range be coded again, at any macroblock.  The cost-down sums travel in the records (WhMbRecord::fme_down)
} WhSccJob;

// ---- rate control with one slice per picture, QP per group of macroblocks: inputs and running state (common/gom_rc.h) -------
#define WH_GOM_MAX 160              /* groups per picture the device state has room for (2304 / 16 rows, one row per group) */

// per-picture inputs (host -> device) and the running state (device), one per picture in flight
typedef struct WhGomRc {
  // inputs
  int32_t n_gom_mb;                 // iNumberMbGom
  int32_t end_mb;                   // pSOverRc->iEndMbSlice
  int32_t target_bits;              // pSOverRc->iTargetBitsSlice
  int32_t min_qp, max_qp;           // pWelsSvcRc->iMinFrameQp / iMaxFrameQp
  int32_t slice_qp;                 // pSlice->uiLastMbQp at the start of the slice (PicInitQp + slice_qp_delta)
  int32_t p_slice;                  // ue(mb_skip_run) exists
  int32_t pad;
  int32_t gom_sad[WH_GOM_MAX];      // pCurrentFrameGomSad of the layer RcGomTargetBits looks at
  // state
  int32_t calc_qp;                  // pSOverRc->iCalculatedQpSlice: the QP of the macroblocks of the current group
  int32_t frame_bits;               // iFrameBitsSlice
  int32_t gom_bits;                 // iGomBitsSlice
  int32_t gom_target;               // iGomTargetBits
  int32_t index;                    // iComplexityIndexSlice: the current group
  int32_t skip_run, last_qp;        // the entropy writer's pSlice->iMbSkipRun and uiLastMbQp after the macroblocks counted so far
  int32_t pad2;
} WhGomRc;

// ---- one picture being encoded (one frame of one session) ---------------------------------------
typedef struct WhPicJob {
  const uint8_t* src[3];     // source picture, dims = mb_w*16 x mb_h*16 (host pads): src[0] = the macroblock-tiled picture (WH_SRC_*); [1] = NULL or the planar picture as uploaded, for run_src_tile_jobs; [2] unused
  uint8_t*       rec[3];     // reconstructed planes (point at pixel (0,0) inside the padded alloc)
  const uint8_t* ref[3];     // reference planes (border-expanded) or NULL for I pictures
  WhMbRecord*    records;    // mb_w*mb_h
  WhMbState*     mbs;        // mb_w*mb_h, this picture
  const WhMbState* ref_mbs;  // previous picture's states (P pictures)
  int32_t        qp;         // picture QP (constant-QP mode) -- per-MB offsets via mb_ctl
  int32_t        slice_type; // WH_SLICE_I / WH_SLICE_P
  const WhMbCtl* mb_ctl;     // optional per-MB control words (QP offsets, re-encode state) or NULL
  int32_t        ref_is_p;   // reference picture was a P picture (co-located MV candidates)
  int32_t        want_bits;  // bit 0: count every macroblock's CAVLC bits (WhMbRecord::cavlc_bits); bit 1: the slice codes ref_idx_l0
                             //   (num_ref_idx_l0_active_minus1 > 0)
  const uint8_t* prev_src_y; // the previous source picture, macroblock-tiled like src[0] (VAA 8x8 SADs of LOW complexity P pictures, scene change)
  uint32_t*      db_flags;   // one word per MB: == db_gen once the MB is deblocked (hand-off between the slices' workgroups)
  uint32_t       db_gen;     // generation of this picture (never 0, changes every frame: the flags need no clearing)
  int32_t        dyn_first;  // size-limited slices (dyn_slice != 0): first macroblock of the slice this launch codes
  uint32_t*      scene_count; // scene-change statistic (kernels/scene_pic.h): zeroed by the host, incremented by the kernel
  // ---- what the reference keeps per LAYER rather than per picture, and what its pre-processing hands to mode decision ----
  int32_t*       sad_cost0;   // pSadCost[0] of every MB (the layer's SMB array, encoder_ext.cpp:900,1675): persists from picture to
                              //   picture whatever the reference picture is; P pictures read and rewrite it, intra macroblocks (I pictures too) zero it
  const int32_t* vaa_sad8x8;  // the host's VAACalcSad result [mb][4] (pVaa->sVaaCalcInfo.pSad8x8), or NULL: computed from prev_src_y
  const int8_t*  bgd_flags;   // pVaa->pVaaBackgroundMbFlag [mb], or NULL: background detection off
  int32_t        mvc_shift;   // sScaleShift (svc_encode_slice.cpp:1652-1655): temporal-layer scaling of the co-located MV candidates
  int32_t        mb_begin;    // only MBs in [mb_begin, mb_end) are coded by this launch (GOM-synchronous rate control); the ones
  int32_t        mb_end;      //   before mb_begin count as done; mb_end == 0 means the whole picture
  int32_t        dyn_slice;   // size-limited slices (SM_SIZELIMITED_SLICE: where a slice ends is only known once its bits are written): 0 = the
                              //   slices are WhSeqParams::slice_first_mb; else 1 + slice_idc of the ONE slice the macroblocks of this launch belong to,
                              //   which begins at dyn_first -- neighbours before dyn_first are another slice's (not available)
  uint8_t*       compact;     // packed records of this picture (common/compact.h), written by the compaction pass, or NULL
  uint32_t*      compact_off; // num_mb + 1 byte offsets into `compact`
  const int16_t* il_hint;     // highest spatial layer of a multi-layer session: what WelsMdInterMbEnhancelayer takes from the layer
                              //   below (svc_mode_decision.cpp:108-150), per MB {sMvBase x, y, flags (bit 0: that MB is intra), 0}; or NULL
  const WhSccJob* scc;        // screen-content P pictures (WhSeqParams::flags & WH_SEQ_SCC): device copy of the inputs above; else NULL
  // WH_SEQ_CHAIN pictures: this picture's own processing order (per slice, like WhSeqParams::mb_order's first section) and, per
  // macroblock, the macroblock it additionally waits for (the previous one of its slice, in coding order, that may search 8x8
  // blocks), or -1
  const uint32_t* scc_order;
  const int32_t* scc_chain_prev;
  // P pictures coded with GOM-level rate control inside the kernel (also WH_SEQ_CHAIN: the groups are bands of the processing order,
  // every macroblock of a group additionally waits for the last macroblock of the group before it): inputs + state, or NULL
  WhGomRc* gom_rc;
  // Size-limited slices code macroblocks AHEAD of the entropy writer and some of them twice (dyn_slice): pSadCost[0] of the layer then has
  // two copies -- sad_cost0 as the previous picture left it (read: a P_Skip above LOW complexity keeps the macroblock's old entry) and
  // sad_cost0_out (written; the context swaps the two when the picture is complete).  One macroblock reads the NEW copy: the first one of
  // a launch that decides it for the second time (dyn_redo: a slice begins with the macroblock the writer took back, or a macroblock is
  // coded again after a CAVLC overflow) -- the reference's first pass has overwritten its entry by then (WelsMdInterSaveSadAndRefMbType).
  int32_t*       sad_cost0_out;
  int32_t        dyn_redo;
  int32_t        pad5;
  // Tiled twins of the border-expanded planes (WH_TILE_*, below): what the search windows of the P kernel are fetched from.
  // [0] luma, [1] Cb and Cr interleaved.  rec_tiles is written by the pass that makes a picture a reference (border expansion,
  // Backend::run_expand), ref_tiles is the reference picture's; both NULL-free whenever rec / ref are.
  uint8_t*       rec_tiles[2];
  const uint8_t* ref_tiles[2];
  // Pictures that get deblocked: mode decision leaves the UNFILTERED reconstruction here, macroblock by macroblock (WH_SRC_MB_BYTES each:
  // 16x16 luma, 8x8 Cb, 8x8 Cr -- three memory lines written whole), instead of in rec[]: the intra predictors of the neighbours and the
  // deblocking pass read it from here, and the deblocking pass writes every sample of rec[] exactly once, filtered or not.  A planar
  // picture costs both passes 32 pieces of 32 different lines per macroblock.  NULL: no deblocking pass follows, rec[] is written directly.
  uint8_t*       rec_blk;
} WhPicJob;

// ---- tiled reference pictures ------------------------------------------------------------------------------------------
// A search window is 64+ columns x 56 rows of the reference at an arbitrary position: in a planar picture every window row is a
// piece of a different 128-byte line (two when it straddles), so a 3.8 KB window cost the fabric 11-12 KB, and the chroma windows
// (32-byte rows) fared worse (profiles/r02_pmc_traffic.json: 12.8x the algorithmic bytes).  The reference picture therefore has
// a second, device-private layout made of 128-byte TILES, one memory line each:
//   luma    16 samples x 8 rows                        -> a 80 x 56 window touches 5 x 7..8 lines  (4.5-5 KB)
//   chroma   8 samples x 8 rows, Cb | Cr per row (16 B) -> a 32 x 32 window (both planes) touches 4 x 4..5 lines (2-2.5 KB)
// covering the whole border-expanded plane (32 / 16 samples each side; strides as in WhSeqParams).  Windows start at tile columns
// (x0 a multiple of 16 / 8), so every 16-byte piece a lane fetches is one row of one tile.
// The SOURCE pictures are kept macroblock by macroblock: 384 bytes per MB -- 16x16 luma row-major, then the 8x8 Cb and the 8x8 Cr block --
// so that a macroblock's source samples are three consecutive memory lines instead of 32 pieces of 32 different ones (a planar picture
// gives every MB row of 16 / 8 bytes a line of its own: 6 KB of line traffic per macroblock incl. the previous picture's luma).  The
// host uploads planar I420 as before; one pass on the device (kernels/tile_pic.h wh_src_tile_item) rearranges it.
#define WH_SRC_MB_BYTES 384
#define WH_SRC_Y_OFF(mb_w, mbx, mby, row, x) ((size_t) ((mby) * (mb_w) + (mbx)) * WH_SRC_MB_BYTES + (size_t) ((row) * 16 + (x)))
#define WH_SRC_C_OFF(mb_w, mbx, mby, pl, row, x) ((size_t) ((mby) * (mb_w) + (mbx)) * WH_SRC_MB_BYTES + (size_t) (256 + (pl) * 64 + (row) * 8 + (x)))
#define WH_TILE_BYTES 128
// byte offset of the 16-byte row piece that holds luma sample (x, y), picture coordinates (x >= -32, y >= -32)
#define WH_TILE_Y_OFF(stride_y, x, y) ((((size_t) (((y) + 32) >> 3) * (size_t) ((stride_y) >> 4) + (size_t) (((x) + 32) >> 4)) << 7) + (size_t) ((((y) + 32) & 7) << 4))
// byte offset of the 16-byte row piece (8 Cb then 8 Cr) that holds chroma sample (x, y) (x >= -16, y >= -16)
#define WH_TILE_C_OFF(stride_c, x, y) ((((size_t) (((y) + 16) >> 3) * (size_t) ((stride_c) >> 3) + (size_t) (((x) + 16) >> 3)) << 7) + (size_t) ((((y) + 16) & 7) << 4))

#define WH_MAX_SLICES 36
#define WH_SEQ_SCC 1                // screen-content mode decision / motion estimation (every picture of the launch has WhPicJob::scc)
#define WH_SEQ_SERIAL 2             // the macroblocks of a slice run one after the other in raster order (fallback of WH_SEQ_CHAIN)
#define WH_SEQ_CHAIN 4              // pictures whose scroll vector is not zero: the reference's directional-vector test of the 8x8 searches
                                    //   reads what the previous macroblock in CODING order that searched 8x8 blocks left (WhSccJob::chain).
                                    //   Only macroblocks whose pre-analysis SADs are not flat can search 8x8 blocks at all (known before the
                                    //   picture starts): each of them waits for the one before it, everything else keeps the 2:1 dependency
                                    //   order -- in a processing order built for the picture that respects both (WhPicJob::scc_order)
#define WH_SEQ_PLAIN 32             // a promise of the host, stripped by the backend: every picture of the launch is a plain camera picture without any optional per-picture
                                    //   input (a session group's step) -- the P kernel then runs a variant that never looks at them (hip_backend.hip WH_PLAIN_KERNEL)
#define WH_SEQ_NO_CTRL 64           // the same kind of promise for the frame API's camera pictures: no inter-layer hints, QP map, GOM rate control, MB ranges or bit
                                    //   counting (what the pre-processing supplies -- VAA SADs, pSadCost, background flags, vector shift -- may be there); candidate: WH_FRAME_KERNEL
#define WH_SEQ_DB_WHOLE 16          // deblocking launch only (set by the backend on its copy): ONE band = the whole picture, in the picture's own 2:1 order (mb_order's second section)
#define WH_SEQ_RANGED 8             // the pictures of the launch code MB ranges (WhPicJob::mb_begin / mb_end, dyn_slice) or carry GOM rate control:
                                    //   the ticket scheduler runs them (k_inter_pool); everything without a flag may take the row scheduler
#define WH_DB_WHOLE_TABLE(P) ((P).db_bands + 3 * (size_t)(P).db_num_bands + 1)
#define WH_DB_BAND_ROWS 24          // a deblocking band (one workgroup) never spans more MB rows than this

// ---- parameters common to every picture of a launch --------------------------------------------
typedef struct WhSeqParams {
  int32_t mb_w, mb_h;
  int32_t src_stride_y, src_stride_c;
  int32_t rec_stride_y, rec_stride_c;   // strides of rec[] and ref[] planes
  int32_t complexity;                   // 0 = LOW (SAD costs, VAA-gated fast I4x4), >=1 = SATD / full
  int32_t chroma_qp_offset;
  int32_t num_slices;                   // slices are contiguous MB ranges
  int32_t slice_first_mb[WH_MAX_SLICES + 1];
  int32_t deblock_idc;                  // 0: all edges, 1: off, 2: not across slice boundaries
  int32_t alpha_offset, beta_offset;
  int32_t mv_range;                     // iMvRange
  int32_t flags;                        // WH_SEQ_*
  int32_t blk8_w, blk8_h;               // picture size in whole 8x8 luma blocks (scene-change statistic)
  unsigned long long* prof;             // optional device array of 64 x 32 cycle counters (phase profiling), or NULL
  const uint32_t* mb_order;             // device table (32-bit entries: a wave-uniform look-up is then a scalar load): [0, num_mb) MB addresses in dependency order per slice (each slice's
                                        // range is [slice_first_mb[s], slice_first_mb[s+1])), [num_mb, 2*num_mb) whole-picture order,
                                        // [2*num_mb, 3*num_mb) per deblocking band (each band's range is [db_bands[b], db_bands[b+1]))
  // Deblocking bands: the MB ranges the deblocking workgroups own (rows of one slice, at most WH_DB_BAND_ROWS of them: a
  // slice is a band unless it is taller, e.g. a single-slice picture is cut into several).  Device table of
  // 3 * db_num_bands + 1 words: [0, n] first MB of band b (and the end of the last), [n+1, 2n] first MB of the slice band b
  // lies in, [2n+1, 3n] end of that slice; then four more words: the same table for ONE band that is the whole picture (WH_DB_WHOLE_TABLE) --
  // what a launch takes when it has about as many pictures as the device has CUs and the filter crosses slice edges anyway (idc 0).
  int32_t db_num_bands, db_max_mbs;     // db_max_mbs / db_max_rows: the largest band (host-side launch geometry)
  const int32_t* db_bands;
  int32_t db_max_rows;
} WhSeqParams;

#ifdef __cplusplus
}
#endif
