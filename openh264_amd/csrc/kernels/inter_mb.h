// inter_mb.h -- P-slice macroblock kernel (placeholder until the inter path lands).
#pragma once
#include "frame_kernels.h"
typedef WhMbLds WhInterLds;
WH_FN void wh_inter_mb_body (WhInterLds& S, const WhSeqParams& P, const WhPicJob& J, int mbx, int mby) {
  wh_intra_mb_body (S, P, J, mbx, mby);
}
