// entropy_cavlc.h -- CAVLC macroblock-layer writer (host side; north_star keeps entropy coding on
// the host).  Consumes the WhMbRecord array the GPU produced.
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

// Writes one macroblock.  `recs` is the picture's record array (neighbour nzc context), `avail`
// tells which neighbours are in the same slice.  Returns 0, or -1 on a level-escape overflow
// (ENC_RETURN_VLCOVERFLOWFOUND in the reference: the caller must re-encode the MB at a higher QP).
// On return *qp_for_deblock is the QP the deblocking filter must see for this MB (the "last coded
// QP" rule for skipped / cbp==0 macroblocks, svc_set_mb_syn_cavlc.cpp:265-267,299-301).
int write_mb_cavlc (BitWriter& bw, SliceEntropyState& st, const WhMbRecord* recs, int mb_w, int mbx, int mby, int avail,
                    int* qp_for_deblock);

// Finish a slice: flush a pending skip run and write rbsp_slice_trailing_bits.
void write_slice_end (BitWriter& bw, SliceEntropyState& st);

}  // namespace wh
