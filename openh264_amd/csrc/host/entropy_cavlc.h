// entropy_cavlc.h -- CAVLC macroblock-layer writer (host side; north_star keeps entropy coding on
// the host).  Consumes the macroblock records the GPU produced, full or packed (common/compact.h).
//
// Syntax per ITU-T H.264 7.3.5 / 9.2; mirrors what the reference emits in
//   codec/encoder/core/src/svc_set_mb_syn_cavlc.cpp:260-322  WelsSpatialWriteMbSyn
//   codec/encoder/core/src/svc_set_mb_syn_cavlc.cpp:58-166   WelsSpatialWriteMbPred
//   codec/encoder/core/src/svc_set_mb_syn_cavlc.cpp:168-245  WelsSpatialWriteSubMbPred
//   codec/encoder/core/src/svc_set_mb_syn_cavlc.cpp:324-440  WelsWriteMbResidual
//   codec/encoder/core/src/set_mb_syn_cavlc.cpp:84-232       CavlcParamCal_c / WriteBlockResidualCavlc
#pragma once
#include "bitwriter.h"
#include "../common/wh_types.h"

namespace wh {

struct SliceEntropyState {
  int slice_type = WH_SLICE_I;
  int last_qp = 26;          // QP of the last MB that coded mb_qp_delta (uiLastMbQp)
  int skip_run = 0;          // pending mb_skip_run (P slices)
  int num_ref_idx_l0_active_minus1 = 0;
};

enum { WH_AVAIL_LEFT = 1, WH_AVAIL_TOP = 2 };

// One macroblock as the writer reads it: its side information (the first 144 bytes of a WhMbRecord), the level blocks, and
// the total_coeff arrays of the two neighbours that give the coeff_token context.  Two sources, same bits:
//   * a full WhMbRecord (what a single session copies back): the 26 level blocks follow the side information;
//   * a packed macroblock (common/compact.h, what a session group copies back): only the blocks of `mask` exist, in mask
//     order, and the writer reads them where they lie -- the picture is never expanded into 960-byte records again.
struct MbView {
  const WhMbRecord* side = nullptr;   // mb_type .. cavlc_bits valid; for a packed P_Skip macroblock only the first 8 bytes
  const int16_t* blocks = nullptr;    // full record: &side->luma[0][0]; packed: the first block that was sent
  uint32_t mask = 0xffffffffu;        // packed: blocks that were sent (bit b as in compact.h); full record: all
  bool packed = false;
  const uint8_t* nzc_left = nullptr;  // nzc[24] of the neighbours inside the slice, nullptr = not available
  const uint8_t* nzc_top = nullptr;
  // block b: 0..15 luma[b], 16 luma_dc, 17..24 chroma_ac[b - 17], 25 chroma_dc (2 x 4 levels); nullptr = nothing but zeros
  const int16_t* block (int b) const {
    if (!packed) return blocks + 16 * b;
    if (!((mask >> b) & 1u)) return nullptr;
    return blocks + 16 * __builtin_popcount (mask & ((1u << b) - 1u));
  }
};
// the view of macroblock xy of a picture of full records / of a packed picture (`off`: compact.h's offset table)
MbView view_of_record (const WhMbRecord* recs, int mb_w, int xy, int avail);
MbView view_of_packed (const uint8_t* packed, const uint32_t* off, int mb_w, int xy, int avail);

// Writes one macroblock.  `avail` (in the view: which neighbour arrays are set) tells which neighbours are in the same
// slice.  Returns 0, or -1 on a level-escape overflow
// (ENC_RETURN_VLCOVERFLOWFOUND in the reference: the caller must re-encode the MB at a higher QP).
// On return *qp_for_deblock is the QP the deblocking filter must see for this MB (the "last coded
// QP" rule for skipped / cbp==0 macroblocks, svc_set_mb_syn_cavlc.cpp:265-267,299-301).
int write_mb_cavlc (BitWriter& bw, SliceEntropyState& st, const MbView& mb, int* qp_for_deblock);

// Finish a slice: flush a pending skip run and write rbsp_slice_trailing_bits.
void write_slice_end (BitWriter& bw, SliceEntropyState& st);

}  // namespace wh
