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
#include <new>
#include <queue>
#include <algorithm>
#include <stdlib.h>
#include "../../../include/welship.h"
#include "backend.h"
#include "../common/mb_order.h"
#include "../common/gom_rc.h"
#include "entropy_cavlc.h"

// Host buffers that are the source or the destination of a device copy get pages of their own.  The HIP runtime pins the pages of a pageable buffer for a large
// hipMemcpyAsync -- read-only when the buffer is the copy's SOURCE -- and keeps them pinned until the stream has drained.  Two such buffers that share a page (the tail
// of one and the head of the next on the brk heap: glibc serves even multi-megabyte vectors from there once a process has freed a few large blocks) then let a
// device-to-host copy write into a page the device has mapped read-only, and the runtime aborts the process: "Memory access fault by GPU ... Write access to a
// read-only page" (round 6: seen once in four runs of the GPU tier's first files with one build of the library and never with another -- it only depends on where the
// heap puts the vectors).  Page-aligned, page-padded blocks cannot share a page with anything.
template <class T> struct WhPageAlloc {
  using value_type = T;
  WhPageAlloc() = default;
  template <class U> WhPageAlloc (const WhPageAlloc<U>&) {}
  T* allocate (size_t n) {
    const size_t bytes = (n * sizeof (T) + 4095) & ~ (size_t)4095;
    void* p = aligned_alloc (4096, bytes ? bytes : 4096);
    if (!p) throw std::bad_alloc();
    return (T*)p;
  }
  void deallocate (T* p, size_t) { free (p); }
  template <class U> bool operator== (const WhPageAlloc<U>&) const { return true; }
  template <class U> bool operator!= (const WhPageAlloc<U>&) const { return false; }
};
template <class T> using WhHostVec = std::vector<T, WhPageAlloc<T>>;
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
