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
// `band` > 0: the rows are taken in bands of that many rows, each band in its own 2:1 diagonal order -- still a topological order (a band
// only depends on the bands above it).  Its one user is GOM-level rate control inside the kernel: a group of macroblock rows
// is a band, so that a group is complete before the next one starts.  (Rounds 1-4 also had it as an experiment knob for the L2 reuse
// distance of the search windows, WELSHIP_MB_BAND: no gain, removed in round 5.)
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

// ---- deblocking bands -----------------------------------------------------------------------------------
// The deblocking kernel gives every workgroup a band: a run of MB rows of one slice, at most `max_rows` of them.  Bands hand
// over to each other through the picture (write-through stores) + flag words (hip_backend.hip); every band edge costs a few
// microseconds per macroblock along it, so a slice stays one band unless it is much taller than a workgroup has wavefronts
// (measured on MI355X, 128 four-slice 1080p pictures: 4 bands per picture 3.4 ms, 5 of 14 rows 5.1 ms, 9 of 8 rows 3.7 ms).
// out: 3 * n + 1 words (see WhSeqParams::db_bands); returns n, or -1 when `cap` words are not enough.
// `by_slice` = false (experiment knob, only with idc 0): bands ignore the slice structure and always start at a row.
WH_ORDER_FN int wh_build_db_bands (int mb_w, int mb_h, int num_slices, const int32_t* slice_first, int deblock_idc, int max_rows,
                                   int32_t* out, int cap, bool by_slice = true) {
  const int num_mb = mb_w * mb_h;
  int n = 0;
  // pass 1: count, pass 2: fill (the three sections depend on n)
  for (int pass = 0; pass < 2; ++pass) {
    const int total = n;
    n = 0;
    const bool whole = deblock_idc == 0 && !by_slice;
    const int ns = whole ? 1 : num_slices;
    for (int s = 0; s < ns; ++s) {
      const int first = whole ? 0 : slice_first[s], last = whole ? num_mb : slice_first[s + 1];
      const int r0 = first / mb_w, r1 = (last - 1) / mb_w, span = r1 - r0 + 1;
      const int parts = (span + max_rows - 1) / max_rows, rows = (span + parts - 1) / parts;
      for (int j = 0; j < parts; ++j) {
        int a = (r0 + j * rows) * mb_w, b = (r0 + (j + 1) * rows) * mb_w;
        if (a < first) a = first;
        if (b > last) b = last;
        if (a >= b) continue;
        if (pass == 1) { out[n] = a; out[total + 1 + n] = first; out[2 * total + 1 + n] = last; }
        ++n;
      }
    }
    if (pass == 0 && 3 * n + 1 > cap) return -1;
    if (pass == 1) out[n] = num_mb;
  }
  return n;
}

// ---- deblocking, one band = the whole picture: two macroblocks per wavefront (kernels/deblock_mb.h wh_deblock_pair_body) -----------------
// A macroblock in the middle of the band [first, last): left, upper, right and lower neighbours (and the left neighbour's lower one) inside it.
WH_ORDER_FN bool wh_db_mb_interior (int mb_w, int xy, int first, int last) {
  const int x = xy % mb_w;
  return x > 0 && x < mb_w - 1 && xy - mb_w >= first && xy - 1 >= first && xy + mb_w < last;
}
// The whole-picture order as ITEMS of one or two macroblocks: out[0] = number of items, out[1 + i] = the item's (first) macroblock A as x | y << 12
// (WH_DB_ITEM_X / _Y: no division by the picture's width on the claim path), bit 31 set when the item is a pair -- A = (x, y) and the next macroblock of its 2:1 diagonal, B = (x - 2, y + 1) = A + mb_w - 2.  Items follow
// the diagonals, so an item only depends on earlier items.  Pairs are made of interior macroblocks, and only on diagonals of at least `min_len`
// macroblocks: a shorter diagonal has fewer macroblocks than the workgroup has waves anyway, and a pair takes longer than one macroblock.
// out: mb_w * mb_h + 1 words.
#define WH_DB_ITEM_PAIR 0x80000000u
#define WH_DB_ITEM_X(it) ((int) ((it) & 0xfffu))
#define WH_DB_ITEM_Y(it) ((int) (((it) >> 12) & 0xfffu))
WH_ORDER_FN int wh_build_db_pair_items (int mb_w, int mb_h, int min_len, uint32_t* out) {
  const int num_mb = mb_w * mb_h;
  int n = 0;
  for (int d = 0; d <= (mb_w - 1) + 2 * (mb_h - 1); ++d) {
    const int y0 = d - (mb_w - 1) > 0 ? (d - (mb_w - 1) + 1) / 2 : 0, y1 = d / 2 < mb_h - 1 ? d / 2 : mb_h - 1;     // rows with 0 <= d - 2 y < mb_w
    const int len = y1 - y0 + 1;
    for (int y = y0; y <= y1; ++y) {
      const int xy = y * mb_w + d - 2 * y;
      if (len >= min_len && y < y1 && wh_db_mb_interior (mb_w, xy, 0, num_mb) && wh_db_mb_interior (mb_w, xy + mb_w - 2, 0, num_mb)) { out[1 + n++] = (uint32_t) (d - 2 * y) | ((uint32_t)y << 12) | WH_DB_ITEM_PAIR; ++y; }
      else out[1 + n++] = (uint32_t) (d - 2 * y) | ((uint32_t)y << 12);
    }
  }
  out[0] = (uint32_t)n;
  return n;
}
