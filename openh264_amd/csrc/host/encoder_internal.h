// encoder_internal.h -- what the translation units of the host side share: encoder.cpp (session API, session groups, pipelined groups) and
// frame_api.cpp (the explicit frame API behind the dispatch-table binding).  Not installed: include/welship.h is the interface.
#pragma once
#include <string.h>
#include <stddef.h>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <deque>
#include <string>
#include <thread>
#include <chrono>
#include <vector>
#include <queue>
#include <algorithm>
#include <stdlib.h>
#include "../../../include/welship.h"
#include "backend.h"
#include "../common/mb_order.h"
#include "../common/gom_rc.h"
#include "entropy_cavlc.h"
#include "headers.h"
#include "../common/compact.h"


namespace wh {
Backend* create_default_backend (int device, const char** err);   // provided by the HIP lib or the test build
// the calling thread's last error text (WelsHipGetLastError)
std::string& last_error();
}
static inline void set_err (const std::string& s) { wh::last_error() = s; }

namespace wh {

inline int align_up (int v, int a) { return (v + a - 1) / a * a; }

// (the unfiltered reconstruction lives in macroblock-contiguous blocks, WhPicJob::rec_blk; the planar in-place layout of rounds 1-3 measured
//  6.07 x against 4.03 x the algorithmic traffic and lost its switch in round 5: profiles/r04_pmc_traffic_unfiltered_recon_in_blocks.txt)
inline bool rec_blocks_on() { return true; }
// Whole-picture deblocking, two macroblocks per wavefront (common/mb_order.h wh_build_db_pair_items): the shortest 2:1 diagonal whose macroblocks
// are paired up (measured on the MI355X, 256 1080p pictures: 16 and below 2.07-2.10 ms per step, 32: 2.13-2.15; profiles/r06_deblock_two_macroblocks_per_wave_ab.txt).  WELSHIP_DB_PAIR_MIN (read when a session's tables are built; a value above
// any diagonal's length = no pairs) is the measurement knob.
inline int db_pair_min_len() { const char* e = getenv ("WELSHIP_DB_PAIR_MIN"); const int v = e ? atoi (e) : 16; return v > 2 ? v : 2; }

struct DevPicture {            // one padded reconstruction buffer + its tiled twin (same allocation) + its MB state
  uint8_t* base = nullptr;
  uint8_t* plane[3] = {nullptr, nullptr, nullptr};   // pixel (0,0)
  uint8_t* tiles[2] = {nullptr, nullptr};            // WH_TILE_*: luma, Cb|Cr -- written when the picture becomes a reference (run_expand)
  // the twin lies behind the planar picture: `planar` = bytes of the padded planes incl. the 2 x 64 guard bytes, `rec_y` = luma plane
  void place_tiles (size_t planar, size_t rec_y) { tiles[0] = base + ((planar + 255) & ~ (size_t)255); tiles[1] = tiles[0] + rec_y; }
  static size_t alloc_bytes (size_t planar) { return ((planar + 255) & ~ (size_t)255) + (planar - 128); }
  WhMbState* mbs = nullptr;
  bool is_p = false;
};

}  // namespace wh
