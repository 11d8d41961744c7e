// mb_order.h -- the order in which the macroblocks of a slice / picture are processed on the device, shared by the
// host (which builds the table once per session) and the kernels (which look up dependencies).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define WH_ORDER_FN static __host__ __device__ inline
#else
#define WH_ORDER_FN static inline
#endif

// ---- dependency order ("wavefront" order) --------------------------------------------------------------
// MB (x,y) depends on (x-1,y), (x,y-1), (x+1,y-1) of its own slice (intra prediction, MV / SAD prediction, skip
// context) -- for deblocking, of the whole picture.  Visiting the MBs of a range [first,last) sorted by
// (x + 2*(y - y_first), y) is a topological order of that graph in which consecutive MBs are independent of each
// other as far as possible (they lie on one 2:1 diagonal), so waves that take MBs from the list in order rarely wait.
// `band` > 0 (an experiment knob, WELSHIP_MB_BAND): the rows are taken in bands of that many rows, each band in its own
// 2:1 diagonal order -- still a topological order (a band only depends on the bands above it); the horizontally adjacent MB
// then follows `band` list entries later instead of one full-height diagonal later, which shortens the L2 reuse distance
// of the overlapping reference windows.
WH_ORDER_FN void wh_build_mb_order (int mb_w, int first, int last, uint16_t* out /* last - first entries */, int band = 0) {
  const int y0 = first / mb_w, y1 = (last - 1) / mb_w;
  if (band <= 0) band = y1 - y0 + 1;
  int n = 0;
  for (int b0 = y0; b0 <= y1; b0 += band) {
    const int b1 = b0 + band - 1 < y1 ? b0 + band - 1 : y1;
    for (int d = 0; d <= (mb_w - 1) + 2 * (b1 - b0); ++d) {
      for (int y = b0; y <= b1; ++y) {
        const int x = d - 2 * (y - b0);
        if (x < 0 || x >= mb_w) continue;
        const int xy = y * mb_w + x;
        if (xy >= first && xy < last) out[n++] = (uint16_t)xy;
      }
    }
  }
}
// The (at most two) MBs whose completion implies that every neighbour MB (x,y) reads is complete: the left one and
// the top-right one (top at the right picture edge); -1 when outside [first, ...).
WH_ORDER_FN void wh_mb_deps (int mb_w, int xy, int first, int* dep_a, int* dep_b) {
  const int mbx = xy % mb_w;
  *dep_a = (mbx > 0 && xy - 1 >= first) ? xy - 1 : -1;
  const int tr = mbx < mb_w - 1 ? xy - mb_w + 1 : xy - mb_w;
  *dep_b = tr >= first ? tr : -1;
}
