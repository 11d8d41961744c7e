// hip_backend.hip -- gfx950 launchers for the macroblock kernels + the wh::Backend implementation.
//
// Launch geometry: one wavefront (64 lanes) processes one macroblock at a time; one workgroup = the wavefronts that
// work on ONE slice of ONE picture (6 or 12 for mode decision, 16 for deblocking); one launch per pass covers every
// slice of every picture of the batch (grid = slices x pictures), so independent pictures (concurrent sessions,
// simulcast layers, all-IDR streams) fill the 256 CUs.  Inside a workgroup the macroblock order and the hand-off
// between neighbouring macroblocks are resolved in the kernel (LDS ticket counter + done flags, see below); between
// the slices of a picture the deblocking kernel synchronises through agent-scope flags in HBM.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <atomic>
#include "../host/backend.h"
#include "../kernels/frame_kernels.h"
#include <mutex>
#include "../kernels/deblock_mb.h"
#include "../kernels/inter_mb.h"
#include "../kernels/expand_pic.h"
#include "../kernels/tile_pic.h"
#include "../kernels/vaa_pic.h"
#include "../kernels/bgd_pic.h"
#include "../kernels/scene_pic.h"
#include "../common/compact.h"

namespace {

// ---- in-kernel scheduling ------------------------------------------------------------------------------------
// One workgroup = one slice (mode decision) or one picture (deblocking) = up to 12/16 wavefronts on ONE compute unit.
// Each wavefront repeatedly takes the next macroblock of the slice's dependency order (an LDS ticket counter), waits
// until the two MBs that gate its neighbourhood are flagged done in LDS, processes the MB, and flags it.  Data passes
// between wavefronts through HBM-backed global memory, which is coherent inside a workgroup (same CU, shared vector
// L1), so the hand-off costs a workgroup-scope release/acquire (s_waitcnt) instead of a kernel boundary.  Tickets are
// handed out in a topological order, so the lowest outstanding ticket can always run: no deadlock; the spin is
// bounded anyway and reports through P.prof-independent error word `err` (host checks it after the step).
#define WH_NUM_QUEUES 32               /* host queues (HIP streams), all created with the backend */
#define WH_ERR_WORDS 4                  /* error dwords per queue: count, block x, block y, awaited index */
#define WH_SEAM_SPIN_LIMIT (1u << 22)   /* waits on flags in device memory (deblocking seams, k_inter_split) */
#define WH_SPIN_LIMIT (1u << 23)       // x ~200 cycles: close to a second.  A wait can legitimately be long when the head of the
                                       // chain waits for another workgroup (deblocking seams) that is not resident yet

// returns false when the wait timed out (err[0] counts, err[1..3] = block x, block y, awaited index of the first one)
__device__ __forceinline__ bool wh_wait_done (const uint32_t* done, int idx, uint32_t* err) {
  if (idx < 0) return true;
  uint32_t spins = 0;
  while (!((__hip_atomic_load (&done[idx >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (idx & 31)) & 1u)) {
    __builtin_amdgcn_s_sleep (4);
    if (++spins > WH_SPIN_LIMIT) {
      if ((threadIdx.x & 63) == 0 && atomicAdd (err, 1u) == 0) { err[1] = blockIdx.x; err[2] = blockIdx.y; err[3] = (uint32_t)idx; }
      return false;
    }
  }
  return true;
}

// where a tile type keeps its profiling accumulators (only the P-frame kernel is instrumented)
__device__ __forceinline__ WhMbLds& wh_prof_holder (WhInterLds& S) { return S.m; }
__device__ __forceinline__ WhMbLds& wh_prof_holder (WhMbLds& S) { return S; }
__device__ __forceinline__ WhMbLds& wh_prof_holder (WhDbLds& S) { return * (WhMbLds*)&S; }     /* never used (PROF = 0) */
#if defined(WH_PROF)
template <class T> __device__ __forceinline__ uint32_t* wh_prof_lds (T& S) { return wh_prof_holder (S).prof; }
#define WH_PROF_ON 1
#else
template <class T> __device__ __forceinline__ uint32_t* wh_prof_lds (T& S) { return nullptr; }      /* (never dereferenced: WH_PROF_ON = 0) */
#define WH_PROF_ON 0
#endif

__device__ __forceinline__ void wh_copy_job (WhPicJob* dst, const WhPicJob* src) {
  for (unsigned i = threadIdx.x; i < sizeof (WhPicJob) / 4; i += blockDim.x) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];      // (a one-wave workgroup has fewer threads than the descriptor has words)
}

#define WH_DEFINE_MB_KERNEL(NAME, LDS_T, BODY, MAX_THREADS, WHOLE_PICTURE, PROF, GOM)                                             \
__global__ __launch_bounds__ (MAX_THREADS) void NAME (WhSeqParams P, const WhPicJob* jobs, uint32_t* err) {             \
  extern __shared__ __align__ (16) uint8_t smem[];                                                                      \
  const int nw = (int)blockDim.x >> 6, lane = (int)threadIdx.x & 63;                                                    \
  const int wave = __builtin_amdgcn_readfirstlane ((int)threadIdx.x >> 6);   /* wave-uniform: keeps S's address scalar */   \
  LDS_T& S = ((LDS_T*)smem)[wave];                                                                                      \
  uint32_t* sched = (uint32_t*) (smem + (size_t)nw * sizeof (LDS_T));       /* [0] ticket counter, [1..] done bits */    \
  const int num_mb = P.mb_w * P.mb_h;                                                                                   \
  const int first = WHOLE_PICTURE ? 0 : P.slice_first_mb[blockIdx.x];                                                   \
  const int n = WHOLE_PICTURE ? num_mb : P.slice_first_mb[blockIdx.x + 1] - first;                                      \
  const uint32_t* order0 = P.mb_order + (WHOLE_PICTURE ? num_mb : first);                                               \
  for (int i = (int)threadIdx.x; i < 1 + ((n + 31) >> 5); i += (int)blockDim.x) sched[i] = 0;                          \
  if (WH_PROF_ON && PROF && P.prof && lane < 32) wh_prof_lds (S)[lane] = 0;                                                           \
  __shared__ WhPicJob Jl;                   /* the job descriptor, read from LDS (lgkmcnt) wherever it is needed */        \
  wh_copy_job (&Jl, &jobs[blockIdx.y]);                                                                                 \
  __syncthreads();                                                                                                      \
  WH_PROF_DECL (P);                                                                                                     \
  const WhPicJob& J = Jl;                                                                                               \
  /* GOM-level rate control inside the kernel (single-slice pictures): the groups are bands of the picture's OWN order, and a     */ \
  /* group's first macroblock waits for the last one of the group before it, which settles its QP (WhPicJob::scc_order / _prev)   */ \
  const uint32_t* order = (GOM && J.gom_rc) ? (const uint32_t*)J.scc_order + first : order0;                                     \
  for (int guard = 0; guard <= n; ++guard) {      /* a wave can never need more than n + 1 tickets */                   \
    int t = 0;                                                                                                          \
    if (lane == 0) t = (int)atomicAdd (&sched[0], 1u);                                                                  \
    t = __builtin_amdgcn_readfirstlane (t);                                                                             \
    if (t >= n) break;                                                                                                  \
    const int xy = order[t];                                                                                            \
    if (PROF) WH_PROF_MARK (P, wh_prof_holder (S), 11);      /* ticket + order lookup */                                 \
    if (J.mb_end > 0) {                   /* GOM-synchronous coding: only [mb_begin, mb_end) in this launch */           \
      if (xy < J.mb_begin) { if (lane == 0) atomicOr (&sched[1 + ((xy - first) >> 5)], 1u << ((xy - first) & 31)); continue; } \
      if (xy >= J.mb_end) continue;                                                                                     \
    }                                                                                                                   \
    int dep_a, dep_b;                                                                                                   \
    wh_mb_deps (P.mb_w, xy, first, &dep_a, &dep_b);                                                                     \
    if (!wh_wait_done (sched + 1, dep_a < 0 ? -1 : dep_a - first, err)) break;     /* give up: the host aborts on err */   \
    if (!wh_wait_done (sched + 1, dep_b < 0 ? -1 : dep_b - first, err)) break;                                          \
    if (GOM && J.gom_rc) { const int dep_c = ((const WH_G int32_t*)J.scc_chain_prev)[xy]; if (!wh_wait_done (sched + 1, dep_c < first ? -1 : dep_c - first, err)) break; } \
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");                                                             \
    if (PROF) WH_PROF_MARK (P, wh_prof_holder (S), 12);      /* dependency wait */                                       \
    BODY (S, P, J, xy % P.mb_w, xy / P.mb_w);                                                                           \
    if (GOM && J.gom_rc) wh_gom_close_if_last (P, J, xy);    /* rate control: the group's last macroblock settles the next group's QP */ \
    if (PROF) WH_PROF_MARK (P, wh_prof_holder (S), 14);      /* the MB itself (sum of the body's own phases) */          \
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");                                                             \
    if (lane == 0) atomicOr (&sched[1 + ((xy - first) >> 5)], 1u << ((xy - first) & 31));                               \
    if (PROF) WH_PROF_MARK (P, wh_prof_holder (S), 13);      /* store drain (release) + done flag */                     \
  }                                                                                                                     \
  if (WH_PROF_ON && PROF && P.prof && lane < 32) atomicAdd (&P.prof[((blockIdx.x + blockIdx.y * 7u) & 63u) * 32u + lane], (unsigned long long)wh_prof_lds (S)[lane]); \
}

WH_DEFINE_MB_KERNEL (k_intra_slice, WhMbLds, wh_intra_mb_body<false>, 1024, 0, 1, false)
WH_DEFINE_MB_KERNEL (k_intra_slice_gom, WhMbLds, wh_intra_mb_body<true>, 1024, 0, 1, true)      /* WH_SEQ_CHAIN launches: GOM-level rate control inside the kernel */

// ---- P pictures: a pool of wavefronts shared by several slices ------------------------------------------------------
// A workgroup owns up to WH_MD_MAX_SLOTS slices (of any pictures of the batch) and as many wavefronts as fit one CU.  A
// free wave takes the next macroblock of the slice with the most macroblocks left, so when one slice of the group is cheap
// (skipped background) or finished, all waves of the CU work on the expensive one: slices cost up to 2-3x their mean on
// real content, and with a fixed six waves per slice the CU idled for half of the launch.  Which slices share a workgroup
// is a table (`groups`): k_md_assign sorts the slices by what they cost in the previous picture of their session and deals
// them out in snake order, heavy with light.  Inside a slice nothing changes: tickets in dependency order (LDS counter),
// done bits in LDS, data through the workgroup-coherent L1/L2.
#define WH_MD_MAX_SLOTS 4
// PLAIN: see inter_mb.h wh_inter_cold_fetch.  The variant has a third fewer instructions (no background-detection, inter-layer, bit-counting,
// rate-control or QP-map code) and codes a session group's pictures 7.4 % faster (MD launch 13.70 -> 12.69 ms, same box:
// profiles/r03_p_kernel_candidates_ab.txt); 0 = every launch takes the general kernel.
#ifndef WH_PLAIN_KERNEL
#define WH_PLAIN_KERNEL 2
#endif
// 2 (the default since round 4: MD launch 10.76 -> 10.45 ms in a same-box A/B, profiles/r04_ab_claim_path_plain2_chroma.txt): additionally a
// variant that knows LOW complexity -- the reference's default -- at compile time (no SATD paths in the search, the refinement and the intra test);
// groups of another complexity take variant 1.  WH_FRAME_KERNEL=1 (the default since round 4): a variant for the frame API's camera pictures
// without control inputs (WH_SEQ_NO_CTRL, promised per launch by the host's frame_run_batch) -- what a session through the dispatch-table binding
// launches unless it has GOM-level rate control, size-limited slices, a QP map or inter-layer hints: one 1080p session 51.5 -> 63.1 frames/s
// (the reference's C path: 61.5; profiles/r04_config5_frame_kernel_ab.txt)
#ifndef WH_FRAME_KERNEL
#define WH_FRAME_KERNEL 1
#endif
// The claim path (a free wave takes its next macroblock) is pure overhead and was a chain of a dozen dependent LDS / memory round
// trips (profiles/r04_p1080p_s256_phase_cycles_v1.txt: 6.6 k of a macroblock's 50 k cycles).  What it needs is therefore kept where a
// wave reaches it without a round trip:
//   * the slots' constants (first macroblock, macroblock count, slice index) in LANE TABLES -- lane sl of a VGPR holds slot sl's value,
//     read with v_readlane;
//   * the tickets left in every slot with ONE LDS access (lane sl reads slot sl's counter);
//   * the processing order through a 64-entry WINDOW per slot in a VGPR: the entries from the wave's last miss on, one per lane.  A wave's
//     tickets of one slot are about a dozen apart (the other waves take the ones in between), so a window serves several claims before
//     the next ticket lies beyond it and one coalesced load refills it;
//   * the job fields the fetch of the next macroblock's inputs reads, copied out of the LDS descriptor in one go (one wait, not one per field).
#ifndef WH_P_WAVES_DEFAULT
#define WH_P_WAVES_DEFAULT 16           /* waves per mode-decision workgroup unless WELSHIP_P_WAVES says otherwise */
#endif
#ifndef WH_MD_ATTR
#define WH_MD_ATTR          /* (A/B builds: an extra function attribute of the mode-decision kernel, e.g. amdgpu_waves_per_eu) */
#endif
template <int MAXT, bool SCC, int VAR = 0>
__global__ WH_MD_ATTR __launch_bounds__ (MAXT) void k_inter_pool (WhSeqParams P, const WhPicJob* jobs, uint32_t* err, const uint16_t* groups, int slots,
                                                       int sched_words, int total_slices, uint32_t* slice_cost) {
  constexpr bool CTRL = VAR == 0, HOSTIN = VAR == 0 || VAR == 3;          // (inter_mb.h wh_inter_cold_fetch)
  extern __shared__ __align__ (16) uint8_t smem[];
  const int nw = (int)blockDim.x >> 6, lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane ((int)threadIdx.x >> 6);
  WhInterLds& S = ((WhInterLds*)smem)[wave];
  __shared__ WhInterStage stage[MAXT / 64];          // separate LDS objects: see WhInterStage
  WhInterStage& G = stage[wave];
  __shared__ WhWinLds winbuf[MAXT / 64];
  uint32_t* sched = (uint32_t*) (smem + (size_t)nw * sizeof (WhInterLds));     // per slot: [0] ticket counter, [1..] done bits
  // The job descriptors of the slots' pictures: a copy in LDS (a uniform field = a ds_read plus a v_readfirstlane).  Measured in round 6 against
  // reading them where they lie through the CONSTANT address space (one scalar load per field, nothing on the vector pipe: 117 vector instructions
  // and 65 LDS instructions fewer in the listing): the scalar loads' latency on the dependent paths costs more than the vector instructions saved --
  // MD launch 7.27 against 7.18 ms, same box, alternating (profiles/r06_ab_job_descriptors_constant_address_space.txt).  -DWH_JOBS_IN_LDS=0 builds that variant.
#ifndef WH_JOBS_IN_LDS
#define WH_JOBS_IN_LDS 1
#endif
#if WH_JOBS_IN_LDS
  __shared__ WhPicJob Jl[WH_MD_MAX_SLOTS];
#define WH_JOB_OF_SLOT(sl) Jl[sl]
#else
  typedef const __attribute__ ((address_space (4))) WhPicJob WhPicJobC;
  const WhPicJobC* const jobs_c = (const WhPicJobC*)jobs;
  __shared__ int slot_pic[WH_MD_MAX_SLOTS];
#define WH_JOB_OF_SLOT(sl) (* (const WhPicJob*) (jobs_c + __builtin_amdgcn_readlane (tab_pic, (sl))))
#endif
  __shared__ int slot_first[WH_MD_MAX_SLOTS], slot_n[WH_MD_MAX_SLOTS], slot_idc[WH_MD_MAX_SLOTS], slot_id[WH_MD_MAX_SLOTS];
  __shared__ int slot_mv[WH_MD_MAX_SLOTS];         // most recent final 16x16 vector of each slot's slice: the window guess (wh_win_speculate)
  for (int i = (int)threadIdx.x; i < slots * sched_words; i += (int)blockDim.x) sched[i] = 0;
  __shared__ uint32_t slot_cost[WH_MD_MAX_SLOTS];
  __shared__ uint32_t waves_left;
  if (threadIdx.x < WH_MD_MAX_SLOTS) slot_cost[threadIdx.x] = 0;
  if (threadIdx.x == 0) waves_left = (uint32_t)nw;
#if WH_PROF_ON
  if (P.prof && lane < 32) S.m.prof[lane] = 0;
#endif
  for (int sl = 0; sl < slots; ++sl) {
    const int k = groups ? (int)groups[blockIdx.x * slots + sl] : (int)blockIdx.x * slots + sl;       // flattened slice id: picture * num_slices + slice
    const bool on = k < total_slices;
    const int pic = on ? k / P.num_slices : 0, idc = on ? k % P.num_slices : 0;
    if (threadIdx.x == 0) {
      slot_first[sl] = P.slice_first_mb[idc]; slot_n[sl] = on ? P.slice_first_mb[idc + 1] - P.slice_first_mb[idc] : 0;
      slot_idc[sl] = idc; slot_id[sl] = on ? k : -1; slot_mv[sl] = 0;
    }
#if WH_JOBS_IN_LDS
    wh_copy_job (&Jl[sl], &jobs[pic]);
#else
    if (threadIdx.x == 0) slot_pic[sl] = pic;
#endif
  }
  if (threadIdx.x == 0) for (int sl = slots; sl < WH_MD_MAX_SLOTS; ++sl) { slot_first[sl] = 0; slot_n[sl] = 0; slot_idc[sl] = 0;
#if !WH_JOBS_IN_LDS
    slot_pic[sl] = 0;
#endif
  }
  __syncthreads();
  WH_PROF_DECL (P);
  const unsigned long long wall0 = P.prof ? wall_clock64() : 0ULL;     // 100 MHz; wave lifetimes against the launch's span (WelsHipGroupProfile)
  // lane tables of the slots' constants (lane sl: slot sl)
  const int tab_first = slot_first[lane & (WH_MD_MAX_SLOTS - 1)], tab_n = slot_n[lane & (WH_MD_MAX_SLOTS - 1)], tab_idc = slot_idc[lane & (WH_MD_MAX_SLOTS - 1)];
#if !WH_JOBS_IN_LDS
  const int tab_pic = slot_pic[lane & (WH_MD_MAX_SLOTS - 1)];
#endif
  WhInterCtx X;
  X.win = &winbuf[wave];
  X.spec_valid = 0;
  X.spec.b = X.win;
  X.last_mv = nullptr;
  uint32_t gone = 0;                      // slots this wave knows to be out of tickets (wave-uniform)
  // cycles / 64 the workgroup spent on each slot's macroblocks: summed in LDS (one add per macroblock), handed to slice_cost by the last wave that
  // leaves.  (Rounds 3-5: four scalar registers per wave, live across the whole body in a kernel that spills scalars.)
  // (slot_cost / waves_left: declared and cleared before the workgroup's barrier above)
  int slot = -1, xy = 0, mbx = 0, mby = 0;        // the macroblock in hand
  int nslot = -1, nxy = 0, nmbx = 0, nmby = 0;    // the wave's next one
  // macroblock address -> (x, y) by a multiplication: ceil (2^32 / mb_w) is exact for addresses below 2^20 and widths below 2^12 (the
  // generic division by a run-time value is a fourteen-instruction fix-up sequence, three times per macroblock)
  const uint32_t w_rcp = P.mb_w > 1 ? 0xffffffffu / (uint32_t)P.mb_w + 1u : 0u;       // (a picture one macroblock wide: 2^32 does not fit -- y is the address itself)
  // order windows: ow<k> lane i = entry ob<k> + i of slot k's processing order (tickets); ob<k> = a ticket no claim can be near: empty
  int ow0 = 0, ow1 = 0, ow2 = 0, ow3 = 0;
  int ob0 = -0x40000000, ob1 = -0x40000000, ob2 = -0x40000000, ob3 = -0x40000000;
  // claim(): the next macroblock for this wave, or nslot = -1 when the workgroup's slices are used up
#define WH_CLAIM()                                                                                                             \
  for (nslot = -1;;) {                                                                                                         \
    int remv = 0;                         /* tickets left in slot `lane` */                                                     \
    if (lane < slots) remv = tab_n - (int)__hip_atomic_load (&sched[lane * sched_words], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    int best = -1, brem = 0;                                                                                                   \
    _Pragma ("unroll") for (int sl = 0; sl < WH_MD_MAX_SLOTS; ++sl) if (sl < slots && !((gone >> sl) & 1u)) {                  \
      const int rem = __builtin_amdgcn_readlane (remv, sl);                                                                    \
      if (rem <= 0) gone |= 1u << sl; else if (rem > brem) { brem = rem; best = sl; }                                          \
    }                                                                                                                          \
    if (best < 0) break;                                                                                                       \
    int tt = 0;                                                                                                                \
    if (lane == 0) tt = (int)atomicAdd (&sched[best * sched_words], 1u);                                                       \
    tt = __builtin_amdgcn_readfirstlane (tt);                                                                                  \
    const int n_ = __builtin_amdgcn_readlane (tab_n, best);                                                                    \
    if (tt >= n_) { gone |= 1u << best; continue; }                                                                            \
    const int first_ = __builtin_amdgcn_readlane (tab_first, best);                                                            \
    int xy_;                                                                                                                   \
    if (SCC && (P.flags & WH_SEQ_SERIAL)) xy_ = first_ + tt;                    /* serial: coding order */                     \
    else {                                                                                                                     \
      int rel = tt - (best == 0 ? ob0 : best == 1 ? ob1 : best == 2 ? ob2 : ob3);                                              \
      if ((unsigned)rel >= 64u) {           /* beyond the slot's window: the 64 entries from this ticket on */                 \
        int v_ = 0;                                                                                                            \
        if (P.flags & WH_SEQ_CHAIN) { const WH_G uint32_t* src_ = (const WH_G uint32_t*)WH_JOB_OF_SLOT (best).scc_order; if (tt + lane < n_) v_ = (int)src_[first_ + tt + lane]; } \
        else { const WH_G uint32_t* src_ = (const WH_G uint32_t*)P.mb_order; if (tt + lane < n_) v_ = (int)src_[first_ + tt + lane]; }   \
        if (best == 0) { ow0 = v_; ob0 = tt; } else if (best == 1) { ow1 = v_; ob1 = tt; } else if (best == 2) { ow2 = v_; ob2 = tt; } else { ow3 = v_; ob3 = tt; } \
        rel = 0;                                                                                                               \
      }                                                                                                                        \
      const int ow_ = best == 0 ? ow0 : best == 1 ? ow1 : best == 2 ? ow2 : ow3;                                               \
      xy_ = __builtin_amdgcn_readlane (ow_, rel);                                                                              \
    }                                                                                                                          \
    const int mb_end_ = CTRL ? WH_JOB_OF_SLOT (best).mb_end : 0;                                                                            \
    if (mb_end_ > 0) {                    /* GOM-synchronous coding: only [mb_begin, mb_end) in this launch */                 \
      if (xy_ < WH_JOB_OF_SLOT (best).mb_begin) { if (lane == 0) atomicOr (&sched[best * sched_words + 1 + ((xy_ - first_) >> 5)], 1u << ((xy_ - first_) & 31)); continue; } \
      if (xy_ >= mb_end_) continue;                                                                                            \
    }                                                                                                                          \
    nslot = best; nxy = xy_;                                                                                                   \
    nmby = P.mb_w > 1 ? (int)__umulhi ((uint32_t)xy_, w_rcp) : xy_; nmbx = xy_ - nmby * P.mb_w;                                \
    break;                                                                                                                     \
  }
  const bool speculate = true;            // (a macroblock's search windows are fetched with its cold inputs, around the slice's last vector)
  // the next macroblock's cold inputs and speculative windows: in flight while the wave waits for the neighbours.  `Jf`: the fields of
  // the slot's job descriptor this reads, taken out of LDS together
#define WH_FETCH_AHEAD()                                                                                                       \
  if (nslot >= 0) {                                                                                                            \
    WhPicJob Jf;                                                                                                               \
    {                                                                                                                          \
      const WhPicJob& Jn = WH_JOB_OF_SLOT (nslot);                                                                                         \
      Jf.src[0] = Jn.src[0]; Jf.prev_src_y = Jn.prev_src_y; Jf.ref_mbs = Jn.ref_mbs; Jf.ref_is_p = Jn.ref_is_p;                \
      Jf.ref_tiles[0] = Jn.ref_tiles[0]; Jf.ref_tiles[1] = Jn.ref_tiles[1];                                                    \
      if (HOSTIN) { Jf.vaa_sad8x8 = Jn.vaa_sad8x8; Jf.sad_cost0 = Jn.sad_cost0; }                                              \
      if (CTRL) { Jf.sad_cost0_out = Jn.sad_cost0_out; Jf.dyn_redo = Jn.dyn_redo; Jf.mb_begin = Jn.mb_begin; }                 \
    }                                                                                                                          \
    const int guess_ = slot_mv[nslot];                                                                                         \
    wh_inter_cold_fetch<VAR> (S, G, lane, P, Jf, nmbx, nmby);                                                                     \
    WH_PROF_SUB (P, S.m, 2);         /* detail: cold inputs issued */                                                          \
    X.spec_valid = 0;                                                                                                          \
    if (speculate) { wh_win_speculate (P, Jf, X.spec, nmbx, nmby, guess_); X.spec_valid = 1; }                                  \
  }
  // (claiming the next macroblock when the body's prediction is final, before residual coding, was measured in round 3: no gain -- and its inputs now land where the macroblock in hand still reads)
  WH_CLAIM()
  WH_FETCH_AHEAD()
  slot = nslot; xy = nxy; mbx = nmbx; mby = nmby;
  while (slot >= 0) {
    const WhPicJob& J = WH_JOB_OF_SLOT (slot);
    const int first = __builtin_amdgcn_readlane (tab_first, slot);
    uint32_t* sc = sched + slot * sched_words;
    WH_PROF_MARK (P, S.m, 11);
    int dep_a = (mbx > 0 && xy - 1 >= first) ? xy - 1 : -1, dep_b;         // (common/mb_order.h wh_mb_deps, with the column already known)
    { const int tr = mbx < P.mb_w - 1 ? xy - P.mb_w + 1 : xy - P.mb_w; dep_b = tr >= first ? tr : -1; }
    if (SCC && (P.flags & WH_SEQ_SERIAL)) { dep_a = xy > first ? xy - 1 : -1; dep_b = -1; }      // the macroblock before it in coding order (WhSccJob::chain)
    if (!wh_wait_done (sc + 1, dep_a < 0 ? -1 : dep_a - first, err)) break;
    if (!wh_wait_done (sc + 1, dep_b < 0 ? -1 : dep_b - first, err)) break;
    if (P.flags & WH_SEQ_CHAIN) {          // ... and the previous macroblock of the slice that may search 8x8 blocks / the last one of the previous group
      const int dep_c = ((const WH_G int32_t*)J.scc_chain_prev)[xy];
      if (!wh_wait_done (sc + 1, dep_c < first ? -1 : dep_c - first, err)) break;
    }
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
    WH_PROF_MARK (P, S.m, 12);
    const uint32_t tc0 = (uint32_t)__builtin_readcyclecounter();
    WV_ASYNC_WAIT();                      /* this MB's cold inputs have landed in the staging area */
    const bool dyn_ = CTRL && J.dyn_slice;
    X.slice_idc = dyn_ ? J.dyn_slice - 1 : __builtin_amdgcn_readlane (tab_idc, slot); X.slice_first = dyn_ ? J.dyn_first : first; X.last_mv = &slot_mv[slot];
    wh_inter_mb_body_t<SCC, VAR> (S, G, P, J, mbx, mby, X);
    if (CTRL && J.gom_rc) wh_gom_close_if_last (P, J, xy);       // rate control: the group's last macroblock settles the next group's QP
    WH_PROF_MARK (P, S.m, 14);
    __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicOr (&sc[1 + ((xy - first) >> 5)], 1u << ((xy - first) & 31));
    WH_PROF_MARK (P, S.m, 13);
    if (slice_cost && lane == 0) atomicAdd (&slot_cost[slot], ((uint32_t)__builtin_readcyclecounter() - tc0) >> 6);
    WH_CLAIM()
    WH_PROF_SUB (P, S.m, 0);       /* detail: slot scan + ticket + order look-up */
    WH_FETCH_AHEAD()
    slot = nslot; xy = nxy; mbx = nmbx; mby = nmby;
  }
#undef WH_CLAIM
#undef WH_FETCH_AHEAD
  if (slice_cost && lane == 0 && atomicSub (&waves_left, 1u) == 1u) {        // the workgroup's last wave: every slot's sum is complete
    for (int sl = 0; sl < slots; ++sl) { const uint32_t c = __hip_atomic_load (&slot_cost[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (slot_id[sl] >= 0 && c) atomicAdd (&slice_cost[slot_id[sl]], c); }
  }
#if WH_PROF_ON
  if (P.prof && lane < 32) atomicAdd (&P.prof[((blockIdx.x * 7u) & 63u) * 32u + lane], (unsigned long long)S.m.prof[lane]);
#endif
  if (P.prof && lane == 0) {
    const unsigned long long wall1 = wall_clock64();
    atomicMax (&P.prof[4096], ~wall0); atomicMax (&P.prof[4097], wall1);
    atomicAdd (&P.prof[4098], wall1 - wall0); atomicAdd (&P.prof[4099], 1ULL);
    atomicMax (&P.prof[4104 + (blockIdx.x & 255u)], wall1);             // when each workgroup's last wave left
  }
}


// ---- P pictures of a launch with FEW slices: every slice on several compute units (round 6) ----------------------------------
// One session through the frame API is one picture per launch: four slices = four workgroups of k_inter_pool on four of 256 CUs, and what the caller waits
// for is a chain of 152 macroblock steps at the latency a macroblock has when sixteen waves share a CU (27 us on the reference's 1080p clip: 4.6 ms per
// picture, profiles/r06_single_session_timeline.txt).  Here a slice's macroblocks are taken by the waves of `parts` workgroups -- few waves per CU, so a
// macroblock runs at close to its unloaded latency -- with the scheduler's words (ticket counter, done bits) in device memory instead of LDS and the data
// that passes between macroblocks (state, unfiltered samples) stored write-through and loaded past the caches (inter_mb.h XWG; the deblocking bands' seams
// work the same way; tools/micro/xcd_stale.hip: a hop costs about a microsecond and never returns a line an earlier read left in the reader's L2).
// Tickets are handed out in a topological order over the whole slice, so a resident wave only ever waits for macroblocks that resident waves hold: no
// assumption on which workgroups are resident, and none on their placement -- blocks b, b + 8, b + 16 ... serve one slice because they share an XCD's L2
// (speed only).  Not for screen content, chained orders or the control inputs (VAR 0): their extra dependencies stay inside one workgroup.
__device__ __forceinline__ bool wh_wait_done_x (const uint32_t* done, int idx, uint32_t* err) {
  if (idx < 0) return true;
  uint32_t spins = 0;
  while (!((__hip_atomic_load (&done[idx >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >> (idx & 31)) & 1u)) {
    __builtin_amdgcn_s_sleep (8);
    if (++spins > WH_SEAM_SPIN_LIMIT) {
      if ((threadIdx.x & 63) == 0 && atomicAdd (err, 1u) == 0) { err[1] = blockIdx.x; err[2] = blockIdx.y; err[3] = 0x40000000u | (uint32_t)idx; }
      return false;
    }
  }
  return true;
}
#define WH_SPLIT_WAVES 6
template <int VAR>
__global__ __launch_bounds__ (WH_SPLIT_WAVES * 64) void k_inter_split (WhSeqParams P, const WhPicJob* jobs, uint32_t* err, uint32_t* xsched, int sched_words, int parts, int total_slices) {
  constexpr bool HOSTIN = VAR == 3;
  extern __shared__ __align__ (16) uint8_t smem[];
  const int lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane ((int)threadIdx.x >> 6);
  // blocks g * 8 * parts + part * 8 + r, part = 0 .. parts - 1, serve slice k = g * 8 + r
  const int per = 8 * parts, g = (int)blockIdx.x / per, rem = (int)blockIdx.x - g * per, k = g * 8 + (rem & 7);
  if (k >= total_slices) return;
  WhInterLds& S = ((WhInterLds*)smem)[wave];
  __shared__ WhInterStage stage[WH_SPLIT_WAVES];
  __shared__ WhWinLds winbuf[WH_SPLIT_WAVES];
  __shared__ WhPicJob Jl;
  __shared__ int slot_mv;
  WhInterStage& G = stage[wave];
  const int pic = k / P.num_slices, idc = k - pic * P.num_slices;
  const int first = P.slice_first_mb[idc], n = P.slice_first_mb[idc + 1] - first;
  uint32_t* sc = xsched + (size_t)k * sched_words;          // [0] ticket counter, [1 ..] done bits: zeroed by the host before the launch
  wh_copy_job (&Jl, &jobs[pic]);
  if (threadIdx.x == 0) slot_mv = 0;
#if WH_PROF_ON
  if (P.prof && lane < 32) S.m.prof[lane] = 0;
#endif
  __syncthreads();
  const WhPicJob& J = Jl;
  const uint32_t* order = P.mb_order + first;
  WhInterCtx X;
  X.win = &winbuf[wave];
  X.spec_valid = 0;
  X.spec.b = X.win;
  X.last_mv = &slot_mv;
  X.slice_idc = idc; X.slice_first = first;
  const uint32_t w_rcp = P.mb_w > 1 ? 0xffffffffu / (uint32_t)P.mb_w + 1u : 0u;
  int xy = -1, mbx = 0, mby = 0;
#define WH_XCLAIM()                                                                                                            \
  {                                                                                                                            \
    int tt = 0;                                                                                                                \
    if (lane == 0) tt = (int)__hip_atomic_fetch_add (&sc[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                  \
    tt = __builtin_amdgcn_readfirstlane (tt);                                                                                  \
    xy = -1;                                                                                                                   \
    if (tt < n) {                                                                                                              \
      xy = (int)order[tt];                                                                                                     \
      mby = P.mb_w > 1 ? (int)__umulhi ((uint32_t)xy, w_rcp) : xy; mbx = xy - mby * P.mb_w;                                    \
      WhPicJob Jf;                                                                                                             \
      Jf.src[0] = J.src[0]; Jf.prev_src_y = J.prev_src_y; Jf.ref_mbs = J.ref_mbs; Jf.ref_is_p = J.ref_is_p;                    \
      Jf.ref_tiles[0] = J.ref_tiles[0]; Jf.ref_tiles[1] = J.ref_tiles[1];                                                      \
      if (HOSTIN) { Jf.vaa_sad8x8 = J.vaa_sad8x8; Jf.sad_cost0 = J.sad_cost0; }                                                \
      const int guess_ = slot_mv;                                                                                              \
      wh_inter_cold_fetch<VAR> (S, G, lane, P, Jf, mbx, mby);                                                                  \
      wh_win_speculate (P, Jf, X.spec, mbx, mby, guess_); X.spec_valid = 1;                                                    \
    }                                                                                                                          \
  }
  WH_XCLAIM()
  while (xy >= 0) {
    const int dep_a = (mbx > 0 && xy - 1 >= first) ? xy - 1 : -1;
    int dep_b;
    { const int tr = mbx < P.mb_w - 1 ? xy - P.mb_w + 1 : xy - P.mb_w; dep_b = tr >= first ? tr : -1; }
    if (!wh_wait_done_x (sc + 1, dep_a < 0 ? -1 : dep_a - first, err)) break;
    if (!wh_wait_done_x (sc + 1, dep_b < 0 ? -1 : dep_b - first, err)) break;
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");       // (nothing of the body moves above the waits; the neighbours' data is loaded past the caches)
    WV_ASYNC_WAIT();
    wh_inter_mb_body_t<false, VAR, true> (S, G, P, J, mbx, mby, X);
    // this wave's write-through stores have left it: then the flag
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_or (&sc[1 + ((xy - first) >> 5)], 1u << ((xy - first) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    WH_XCLAIM()
  }
#undef WH_XCLAIM
#if WH_PROF_ON
  if (P.prof && lane < 32) atomicAdd (&P.prof[((blockIdx.x * 7u) & 63u) * 32u + lane], (unsigned long long)S.m.prof[lane]);
#endif
}

// Deal the slices of a batch out to the mode-decision workgroups: sorted by the cost they had in the previous picture
// (slice_cost, accumulated by k_inter_pool; cleared here for the coming launch), then in snake order over the groups, so
// that every group gets a heavy and a light share.  One workgroup; n <= 4096 slices.
__global__ __launch_bounds__ (1024) void k_md_assign (uint32_t* slice_cost, uint16_t* groups, int n, int n_groups, int slots) {
  __shared__ uint32_t key[4096];
  __shared__ uint16_t val[4096];
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = (int)threadIdx.x; i < m; i += (int)blockDim.x) { key[i] = i < n ? slice_cost[i] : 0u; val[i] = (uint16_t) (i < n ? i : 0xffff); }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) slice_cost[i] = 0;
  // bitonic sort, descending by key; entries beyond n carry key 0 and id 0xffff and end up at the tail (ties: any order)
  for (int k = 2; k <= m; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = (int)threadIdx.x; i < m; i += (int)blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;         // this stretch sorts descending when `up`
          const uint32_t a = key[i], b = key[l];
          const bool sw = up ? (a < b || (a == b && val[i] == 0xffff && val[l] != 0xffff)) : (a > b);
          if (sw) { key[i] = b; key[l] = a; const uint16_t t = val[i]; val[i] = val[l]; val[l] = t; }
        }
      }
      __syncthreads();
    }
  // deal the sorted slices out: the p-th REAL entry (padding carries id 0xffff; it sorts behind every real entry of its key, but a
  // zero-cost slice shares the key) goes to row p / n_groups of the snake.  p = a prefix count over the sorted list: per chunk of 1024
  // entries a ballot inside each wave, the waves' totals scanned by one thread.
  for (int g = (int)threadIdx.x; g < n_groups * slots; g += (int)blockDim.x) groups[g] = 0xffff;
  __shared__ int wtot[17];
  int carry = 0;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nwv = (int)blockDim.x >> 6;
  for (int base = 0; base < m; base += (int)blockDim.x) {
    const int i = base + (int)threadIdx.x;
    const bool valid = i < m && val[i] != 0xffff;
    const unsigned long long b = __ballot (valid);
    const int before = __builtin_popcountll (b & ((1ull << lane) - 1ull));
    __syncthreads();                                   // (wtot of the previous chunk has been read by everybody)
    if (lane == 0) wtot[wave] = __builtin_popcountll (b);
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int w2 = 0; w2 < nwv; ++w2) { const int t = wtot[w2]; wtot[w2] = acc; acc += t; } wtot[16] = acc; }
    __syncthreads();
    if (valid) {
      const int p = carry + wtot[wave] + before;
      const int r = p / n_groups, c = p % n_groups;
      const int g = (r & 1) ? n_groups - 1 - c : c;
      if (r < slots) groups[g * slots + r] = val[i];
    }
    carry += wtot[16];
  }
}

// Deblocking with one workgroup per BAND (WhSeqParams::db_bands: whole rows of one slice -- of the picture when the filter
// crosses slice boundaries, disable_deblocking_filter_idc 0 -- and never more rows than the workgroup has wavefronts, so
// a 2:1 diagonal of the band is filtered in one round).  The MBs along a band's upper edge wait for MBs of the band above
// -- another workgroup, in general on another XCD.  The producer writes the samples a later band reads with write-through
// stores (sc0 sc1), waits for them (vmcnt) and sets a flag word in HBM (J.db_flags[mb] = J.db_gen); the consumer polls the
// flag and loads those samples past its caches (sc0 sc1) -- MI355X_MICROARCH.md, inter-workgroup visibility, "sc0 sc1 stores
// and loads both sides".  No agent-scope fences: a release would write back every dirty line of the XCD's L2 and an acquire
// empty the CU's L1 for all its waves, once per macroblock of every band edge (measured: 3 ms of a 5.4 ms pass).  Only MBs
// that a later band can depend on publish.  The workgroups of one picture have consecutive block ids and band b-1 never waits for band b, so the chain
// cannot deadlock while the earlier workgroup is scheduled; the spin is bounded regardless.
__global__ __launch_bounds__ (1024) void k_deblock_slices (WhSeqParams P, const WhPicJob* jobs, uint32_t* err) {
  extern __shared__ __align__ (16) uint8_t smem[];
  const int nw = (int)blockDim.x >> 6, lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane ((int)threadIdx.x >> 6);
  WhDbLds& S = ((WhDbLds*)smem)[wave];
  WhDbXchg E;                                 // strip exchange between the waves (deblock_mb.h), indexed by absolute MB row
  E.top = (uint32_t*) (smem + (size_t)nw * sizeof (WhDbLds));
  E.left = E.top + (size_t)P.mb_w * 24;
  E.first_row = 0;
  uint32_t* sched = E.left + (size_t)P.mb_h * 32;
  const int w = P.mb_w, num_mb = P.mb_w * P.mb_h;
  const int nb = P.db_num_bands;
  const int first = P.db_bands[blockIdx.x], last = P.db_bands[blockIdx.x + 1], n = last - first;
  // idc 2: nothing is filtered (or needed) across slices -- only the bands of this band's own slice matter
  const bool cross = P.deblock_idc == 0;
  const int sfirst = cross ? 0 : P.db_bands[nb + 1 + blockIdx.x], slast = cross ? num_mb : P.db_bands[2 * nb + 1 + blockIdx.x];
  const uint32_t* order = (P.flags & WH_SEQ_DB_WHOLE) ? P.mb_order + num_mb : P.mb_order + 2 * num_mb + first;
  for (int i = (int)threadIdx.x; i < 1 + ((n + 31) >> 5); i += (int)blockDim.x) sched[i] = 0;
  __shared__ WhPicJob Jl;
  wh_copy_job (&Jl, &jobs[blockIdx.y]);
  __syncthreads();
  const WhPicJob& J = Jl;
  uint32_t* flags = J.db_flags;
  const uint32_t gen = J.db_gen;
  __shared__ WhDbStage stage[16];                 // separate LDS object (see WhInterStage)
  WhDbStage& G = stage[wave];
#if WH_PROF_ON
  if (P.prof && lane < 32) S.prof[lane] = 0;
#endif
  WH_PROF_DECL (P);
  for (int guard = 0; guard <= n; ++guard) {
    // a wave takes a ticket only when it is free (a held ticket could be the one the whole dependency chain is waiting
    // for), starts the loads of that MB's own inputs at once and waits for the neighbours while they are in flight
    int t = 0;
    if (lane == 0) t = (int)atomicAdd (&sched[0], 1u);
    t = __builtin_amdgcn_readfirstlane (t);
    if (t >= n) break;
    const int xy = order[t];
    WH_PROF_MARK (P, S, 0);   // ticket + order
    wh_deblock_cold_fetch (G, lane, P, J, xy % w, xy / w);
    WH_PROF_MARK (P, S, 1);   // own inputs requested
    int dep_a, dep_b;                           // picture-wide dependencies: left, top-right (top at the right edge)
    wh_mb_deps (w, xy, 0, &dep_a, &dep_b);
    bool remote = false, ok = true;
    for (int k = 0; k < 2 && ok; ++k) {
      const int dep = k == 0 ? dep_a : dep_b;
      if (dep < 0) continue;
      if (dep >= first) ok = wh_wait_done (sched + 1, dep - first, err);
      else if (dep >= sfirst) {
        uint32_t spins = 0;
        while (__hip_atomic_load (&flags[dep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != gen) {
          __builtin_amdgcn_s_sleep (8);
          if (++spins > WH_SEAM_SPIN_LIMIT) { if (lane == 0 && atomicAdd (err, 1u) == 0) { err[1] = blockIdx.x; err[2] = blockIdx.y; err[3] = 0x80000000u | (uint32_t)dep; } ok = false; break; }
        }
        remote = true;
      }
    }
    if (!ok) break;
    // no agent-scope acquire: what comes from another band is loaded past the caches (wh_ld_xwg32 in the body)
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
    WH_PROF_MARK (P, S, remote ? 3 : 2);        // neighbours done: 2 = inside the workgroup, 3 = incl. a wait across the seam
    WV_ASYNC_WAIT();                            // this MB's staged inputs have landed
    WH_PROF_MARK (P, S, 4);   // own inputs landed
    // MBs a later band may wait for: its left neighbour (xy + 1), top (xy + w) or top-right consumer (xy + w - 1).  Their
    // stores are write-through (no agent-scope release: buffer_wbl2 would write back every dirty line of the XCD's L2 for
    // each of them); the flag follows once the stores have left the wave
    const bool publish = xy + w + 1 >= last && last < slast;
    const bool drain = wh_deblock_mb_body (S, G, E, first, last, P, J, xy % w, xy / w, 0, 0, 0, publish);
    WH_PROF_MARK (P, S, 11);  // (body total: ids 5..8)
    if (publish) {
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store (&flags[xy], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (drain) {
      __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    }
    // inside the workgroup the hand-off is LDS only (strip exchange + flag, executed in order by the LDS unit): the
    // wave's global stores need not have completed -- no other MB of the slice writes or reads those samples
    WV_SYNC();
    if (lane == 0) atomicOr (&sched[1 + ((xy - first) >> 5)], 1u << ((xy - first) & 31));
    WH_PROF_MARK (P, S, 9);   // publish / release + done flag
  }
  WH_PROF_MARK (P, S, 10);    // idle tail: no ticket left, the slice is still being finished by other waves
  // second half of the profile buffer (the first belongs to the mode-decision kernels)
#if WH_PROF_ON
  if (P.prof && lane < 32) atomicAdd (&P.prof[2048u + ((blockIdx.x + blockIdx.y * 7u) & 63u) * 32u + lane], (unsigned long long)S.prof[lane]);
#endif
}


// Deblocking with one band = the whole picture (large batches, filter across slice edges), TWO macroblocks per wavefront where the processing
// order pairs them up (round 6; kernels/deblock_mb.h wh_deblock_pair_body, common/mb_order.h wh_build_db_pair_items: the order's fourth section).
// A ticket is an item of one or two macroblocks of one 2:1 diagonal; an item only depends on earlier items, everything is inside the workgroup
// (no seams: k_deblock_slices' hand-off between bands does not exist here).  Two tiles and two staging areas per wave.
#ifndef WH_DB_ITEMS_IN_LDS
#define WH_DB_ITEMS_IN_LDS 1
#endif
#define WH_DB_ITEMS_LDS_MAX_MB 9216          /* 36 KB of items at most (1080p: 32 KB) */
__global__ __launch_bounds__ (1024) void k_deblock_pairs (WhSeqParams P, const WhPicJob* jobs, uint32_t* err) {
  extern __shared__ __align__ (16) uint8_t smem[];
  const int nw = (int)blockDim.x >> 6, lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane ((int)threadIdx.x >> 6);
  WhDbLds* S2 = (WhDbLds*)smem + 2 * wave;
  WhDbLds& S = S2[0];
  WhDbXchg E;                                 // strip exchange between the waves (deblock_mb.h), indexed by absolute MB row
  E.top = (uint32_t*) (smem + (size_t)nw * 2 * sizeof (WhDbLds));
  E.left = E.top + (size_t)P.mb_w * 24;
  E.first_row = 0;
  uint32_t* sched = E.left + (size_t)P.mb_h * 32;
  const int w = P.mb_w, num_mb = P.mb_w * P.mb_h;
  const uint32_t* items_g = P.mb_order + 3 * (size_t)num_mb;
  const int n = (int)items_g[0];
  for (int i = (int)threadIdx.x; i < 1 + ((num_mb + 31) >> 5); i += (int)blockDim.x) sched[i] = 0;
  // the item list in LDS when the picture is small enough (WH_DB_ITEMS_LDS_MAX_MB, the launch reserves the room): a wave's claim is then an LDS atomic and
  // an LDS read instead of an LDS atomic and a round trip to the L2 in front of every item's loads
#if WH_DB_ITEMS_IN_LDS
  uint32_t* items_l = sched + 1 + ((num_mb + 31) >> 5);
  const bool items_in_lds = num_mb <= WH_DB_ITEMS_LDS_MAX_MB;
  if (items_in_lds) for (int i = (int)threadIdx.x; i <= n; i += (int)blockDim.x) items_l[i] = items_g[i];
#else
  const bool items_in_lds = false; uint32_t* items_l = nullptr;
#endif
  __shared__ WhPicJob Jl;
  wh_copy_job (&Jl, &jobs[blockIdx.y]);
  __syncthreads();
  const WhPicJob& J = Jl;
  __shared__ WhDbStage stage[16][2];              // separate LDS object (see WhInterStage)
  WhDbStage* G2 = stage[wave];
#if WH_PROF_ON
  if (P.prof && lane < 32) S.prof[lane] = 0;
#endif
  WH_PROF_DECL (P);
  // an item: its first macroblock (x, y), `kind` = 0 none, 1 one macroblock, 2 the pair with (x - 2, y + 1)
  int kind = 0, ax = 0, ay = 0;
#define WH_DB_CLAIM(KIND, X, Y) do {                                                                                     \
    int t_ = 0;                                                                                                           \
    if (lane == 0) t_ = (int)atomicAdd (&sched[0], 1u);                                                                   \
    t_ = __builtin_amdgcn_readfirstlane (t_);                                                                             \
    if (t_ >= n) KIND = 0;                                                                                                \
    else { const uint32_t it_ = WH_DB_ITEMS_IN_LDS && items_in_lds ? (uint32_t)__builtin_amdgcn_readfirstlane ((int)items_l[1 + t_]) : items_g[1 + t_]; KIND = (it_ & WH_DB_ITEM_PAIR) ? 2 : 1; X = WH_DB_ITEM_X (it_); Y = WH_DB_ITEM_Y (it_); } \
    /* (Touching the inputs of the item sixteen tickets further on at this point -- one word per 128-byte line by LDS-DMA into a scrap area, so that they */ \
    /*  are in the L2 when their wave asks for them -- changes nothing: 2.01 against 2.04 ms per step.  The wait in front of a body is the picture's own  */ \
    /*  ramps, where a 2:1 diagonal has fewer items than the workgroup has waves, not the loads: profiles/r06_deblock_two_macroblocks_per_wave_ab.txt)     */ \
  } while (0)
  WH_DB_CLAIM (kind, ax, ay);
  if (kind) { wh_deblock_cold_fetch (G2[0], lane, P, J, ax, ay); if (kind > 1) wh_deblock_cold_fetch (G2[1], lane, P, J, ax - 2, ay + 1); }
  WH_PROF_MARK (P, S, 0);     // ticket + order (+ the first item's inputs requested)
  for (int guard = 0; guard <= n && kind; ++guard) {
    const bool pair = kind > 1;
    const int xy = ay * w + ax, xb = xy + w - 2, bx = ax - 2, by = ay + 1;
    // left and top-right neighbours (top at the right edge) of A; of B: its left one -- its top-right one is A's left
    int dep_a, dep_b;
    wh_mb_deps (w, xy, 0, &dep_a, &dep_b);
    bool ok = wh_wait_done (sched + 1, dep_a, err) && wh_wait_done (sched + 1, dep_b, err);
    if (ok && pair) ok = wh_wait_done (sched + 1, xb - 1, err);
    if (!ok) break;
    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
    WH_PROF_MARK (P, S, 2);   // neighbours done
    WV_ASYNC_WAIT();                            // the staged inputs have landed
    WH_PROF_MARK (P, S, 4);   // own inputs landed
    // (Taking the next item here and requesting its inputs as soon as the staging areas are empty -- so that ticket, order look-up and the loads' way
    //  through the memory system run beside this item's filters -- was measured and lost, as it had with one macroblock per wave: 2.09 -> 2.35 ms per
    //  step of 256 1080p pictures, profiles/r06_deblock_two_macroblocks_per_wave_ab.txt.  A diagonal of a 1080p picture is about thirty items; sixteen
    //  items in hand plus sixteen held reach into the next diagonal, whose items then wait for this one's.)
    if (pair && J.rec_blk) wh_deblock_pair_body (S2, G2, E, P, J, ax, ay, bx, by);
    else {          // (a pair without WhPicJob::rec_blk -- no launch of this library -- is two macroblocks one after the other)
      bool drain = wh_deblock_mb_body (S, G2[0], E, 0, num_mb, P, J, ax, ay, 0, 0, 0, false);
      if (pair) drain |= wh_deblock_mb_body (S2[1], G2[1], E, 0, num_mb, P, J, bx, by, 0, 0, 0, false);
      if (drain) __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
    }
    WH_PROF_MARK (P, S, 11);  // (body total: ids 5..8)
    // the hand-off is LDS only (strip exchange + flags, executed in order by the LDS unit): the wave's global stores need not have completed --
    // no other macroblock of the picture writes or reads those samples
    WV_SYNC();
    if (lane == 0) {
      atomicOr (&sched[1 + (xy >> 5)], 1u << (xy & 31));
      if (pair) atomicOr (&sched[1 + (xb >> 5)], 1u << (xb & 31));
    }
    WH_PROF_MARK (P, S, 9);   // done flags
    // a wave takes a ticket only when it is free, starts the loads of that item's own inputs at once and waits for the neighbours while they are in flight
    WH_DB_CLAIM (kind, ax, ay);
    if (kind) { wh_deblock_cold_fetch (G2[0], lane, P, J, ax, ay); if (kind > 1) wh_deblock_cold_fetch (G2[1], lane, P, J, ax - 2, ay + 1); }
    WH_PROF_MARK (P, S, 0);   // ticket + order + own inputs requested
  }
#undef WH_DB_CLAIM
  WH_PROF_MARK (P, S, 10);    // idle tail: no ticket left, the picture is still being finished by other waves
#if WH_PROF_ON
  if (P.prof && lane < 32) atomicAdd (&P.prof[2048u + ((blockIdx.x + blockIdx.y * 7u) & 63u) * 32u + lane], (unsigned long long)S.prof[lane]);
#endif
}

// ---- record compaction (common/compact.h): one workgroup per picture ------------------------------------------------
// Pass 1: one wavefront per MB tests its 25 coefficient blocks (+ chroma DC) for non-zero levels (ballot over four
// lane-parallel loads), gates them by type / cbp, and leaves mask and size in LDS.  Pass 2: exclusive scan of the sizes.
// Pass 3: one wavefront per MB copies header, side info and the selected blocks to their place.  Reads the records once from
// L2/HBM (they were written by the previous launch), writes ~1/6 of them.
#define WH_CP_MAX_MB 9216          /* 4096 x 2304 / 256 would be 36864: larger pictures keep the full-record path */
__global__ __launch_bounds__ (1024) void k_compact (WhSeqParams P, const WhPicJob* jobs) {
  __shared__ uint16_t s_size[WH_CP_MAX_MB];
  __shared__ uint32_t s_mask[WH_CP_MAX_MB];
  __shared__ uint32_t s_part[1024];
  const WhPicJob J = jobs[blockIdx.x];
  if (!J.compact || !J.compact_off) return;        // (a picture of the batch whose caller takes the full records: uniform for the workgroup)
  const int num_mb = P.mb_w * P.mb_h, lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  const WH_G WhMbRecord* recs = (const WH_G WhMbRecord*)J.records;
  for (int xy = wave; xy < num_mb; xy += nw) {
    const WH_G uint32_t* r = (const WH_G uint32_t*)&recs[xy];
    const uint32_t h0 = r[0];                                        // mb_type, cbp, luma_qp, chroma_qp
    const int mb_type = (int) (h0 & 0xff), cbp = (int) ((h0 >> 8) & 0xff);
    uint32_t mask = 0;
    if (mb_type != WH_MB_PSKIP) {
      const WH_G uint32_t* c = r + 36;                               // 204 coefficient dwords: 25 blocks of 8 + chroma DC (4)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = lane + 64 * k;
        const unsigned long long b = __ballot (d < 204 && c[d] != 0u);
#pragma unroll
        for (int g = 0; g < 8; ++g) if ((b >> (8 * g)) & 0xffull) { const int blk = 8 * k + g; mask |= 1u << (blk < 25 ? blk : 25); }
      }
      mask &= wh_compact_allowed (mb_type, cbp);
    }
    if (lane == 0) { s_mask[xy] = mask; s_size[xy] = (uint16_t)wh_compact_size (mb_type, mask); }
  }
  __syncthreads();
  // exclusive scan: thread t owns the MBs [t * per, (t + 1) * per)
  const int per = (num_mb + (int)blockDim.x - 1) / (int)blockDim.x, t = (int)threadIdx.x;
  uint32_t sum = 0;
  for (int i = t * per; i < (t + 1) * per && i < num_mb; ++i) sum += s_size[i];
  s_part[t] = sum;
  __syncthreads();
  for (int d = 1; d < (int)blockDim.x; d <<= 1) {
    const uint32_t v = t >= d ? s_part[t - d] : 0u;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? s_part[t - 1] : 0u;
  for (int i = t * per; i < (t + 1) * per && i < num_mb; ++i) { J.compact_off[i] = run; run += s_size[i]; }
  if (t == (int)blockDim.x - 1) J.compact_off[num_mb] = s_part[t];
  __syncthreads();
  for (int xy = wave; xy < num_mb; xy += nw) {
    const WH_G uint32_t* r = (const WH_G uint32_t*)&recs[xy];
    WH_G uint32_t* o = (WH_G uint32_t*) ((WH_G uint8_t*)J.compact + J.compact_off[xy]);
    const uint32_t mask = s_mask[xy];
    const uint32_t h0 = r[0];
    if ((h0 & 0xff) == WH_MB_PSKIP) {
      if (lane < 2) o[lane] = r[lane];
      else if (lane == 2) o[2] = r[30];                              // cost (byte 120)
      else if (lane == 3) o[3] = ((const WH_G uint8_t*)r)[WH_COMPACT_SIDE - 16];      // bgd_skip
      continue;
    }
    if (lane == 0) o[0] = mask;
    if (lane < 36) o[1 + lane] = r[lane];
    const WH_G uint32_t* c = r + 36;
    WH_G uint32_t* q = o + 37;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = lane + 64 * k;
      if (d < 204) {
        const int blk = d >> 3 < 25 ? d >> 3 : 25;
        if ((mask >> blk) & 1u) q[8 * __builtin_popcount (mask & ((1u << blk) - 1u)) + (d - 8 * blk)] = c[d];
      }
    }
  }
}

// The same for launches of a few pictures (round 6): one workgroup per picture packed a 1080p picture's 7.8 MB of records in 0.9 ms -- one CU's rate -- which a
// single session through the frame API waited for, picture by picture (profiles/r06_single_session_timeline.txt).  A picture is cut into chunks of
// WH_CP_CHUNK macroblocks, one 256-thread workgroup each, in two launches: k_compact_sizes leaves every macroblock's packed size in its slot of the offset
// table and the chunk's total in `totals`; k_compact_chunks turns the sizes into offsets (the chunks before it: a sum over at most WH_CP_MAX_CHUNKS words;
// its own macroblocks: a scan in LDS) and copies.  The masks are computed twice (the records are L2-warm the second time) instead of being kept.
#define WH_CP_CHUNK 256
#define WH_CP_MAX_CHUNKS ((WH_CP_MAX_MB + WH_CP_CHUNK - 1) / WH_CP_CHUNK)
__device__ __forceinline__ uint32_t wh_cp_mask (const WH_G uint32_t* r, int lane, int mb_type, int cbp) {
  uint32_t mask = 0;
  if (mb_type != WH_MB_PSKIP) {
    const WH_G uint32_t* c = r + 36;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = lane + 64 * k;
      const unsigned long long b = __ballot (d < 204 && c[d] != 0u);
#pragma unroll
      for (int g = 0; g < 8; ++g) if ((b >> (8 * g)) & 0xffull) { const int blk = 8 * k + g; mask |= 1u << (blk < 25 ? blk : 25); }
    }
    mask &= wh_compact_allowed (mb_type, cbp);
  }
  return mask;
}
__global__ __launch_bounds__ (256) void k_compact_sizes (WhSeqParams P, const WhPicJob* jobs, uint32_t* totals) {
  __shared__ uint32_t s_sum;
  const WhPicJob J = jobs[blockIdx.y];
  if (!J.compact || !J.compact_off) return;
  const int num_mb = P.mb_w * P.mb_h, lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  const int a = (int)blockIdx.x * WH_CP_CHUNK, b = a + WH_CP_CHUNK < num_mb ? a + WH_CP_CHUNK : num_mb;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const WH_G WhMbRecord* recs = (const WH_G WhMbRecord*)J.records;
  uint32_t sum = 0;
  for (int xy = a + wave; xy < b; xy += nw) {
    const WH_G uint32_t* r = (const WH_G uint32_t*)&recs[xy];
    const uint32_t h0 = r[0];
    const int mb_type = (int) (h0 & 0xff), cbp = (int) ((h0 >> 8) & 0xff);
    const uint32_t size = wh_compact_size (mb_type, wh_cp_mask (r, lane, mb_type, cbp));
    if (lane == 0) J.compact_off[xy] = size;
    sum += size;
  }
  if (lane == 0) atomicAdd (&s_sum, sum);
  __syncthreads();
  if (threadIdx.x == 0) totals[blockIdx.y * WH_CP_MAX_CHUNKS + blockIdx.x] = s_sum;
}
__global__ __launch_bounds__ (256) void k_compact_chunks (WhSeqParams P, const WhPicJob* jobs, const uint32_t* totals) {
  __shared__ uint32_t s_off[WH_CP_CHUNK];
  const WhPicJob J = jobs[blockIdx.y];
  if (!J.compact || !J.compact_off) return;
  const int num_mb = P.mb_w * P.mb_h, lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nw = (int)blockDim.x >> 6, t = (int)threadIdx.x;
  const int a = (int)blockIdx.x * WH_CP_CHUNK, b = a + WH_CP_CHUNK < num_mb ? a + WH_CP_CHUNK : num_mb;
  uint32_t base = 0;
  for (int c = 0; c < (int)blockIdx.x; ++c) base += totals[blockIdx.y * WH_CP_MAX_CHUNKS + c];
  const uint32_t own = a + t < b ? J.compact_off[a + t] : 0u;      // (this macroblock's size, left by k_compact_sizes)
  s_off[t] = own;
  __syncthreads();
  for (int d = 1; d < WH_CP_CHUNK; d <<= 1) {
    const uint32_t v = t >= d ? s_off[t - d] : 0u;
    __syncthreads();
    s_off[t] += v;
    __syncthreads();
  }
  const uint32_t incl = s_off[t];
  __syncthreads();
  s_off[t] = base + incl - own;
  if (a + t < b) J.compact_off[a + t] = base + incl - own;
  if (a + t == num_mb - 1) J.compact_off[num_mb] = base + incl;
  __syncthreads();
  const WH_G WhMbRecord* recs = (const WH_G WhMbRecord*)J.records;
  for (int xy = a + wave; xy < b; xy += nw) {
    const WH_G uint32_t* r = (const WH_G uint32_t*)&recs[xy];
    WH_G uint32_t* o = (WH_G uint32_t*) ((WH_G uint8_t*)J.compact + s_off[xy - a]);
    const uint32_t h0 = r[0];
    const int mb_type = (int) (h0 & 0xff), cbp = (int) ((h0 >> 8) & 0xff);
    if (mb_type == WH_MB_PSKIP) {
      if (lane < 2) o[lane] = r[lane];
      else if (lane == 2) o[2] = r[30];
      else if (lane == 3) o[3] = ((const WH_G uint8_t*)r)[WH_COMPACT_SIDE - 16];
      continue;
    }
    const uint32_t mask = wh_cp_mask (r, lane, mb_type, cbp);
    if (lane == 0) o[0] = mask;
    if (lane < 36) o[1 + lane] = r[lane];
    const WH_G uint32_t* c = r + 36;
    WH_G uint32_t* q = o + 37;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = lane + 64 * k;
      if (d < 204) {
        const int blk = d >> 3 < 25 ? d >> 3 : 25;
        if ((mask >> blk) & 1u) q[8 * __builtin_popcount (mask & ((1u << blk) - 1u)) + (d - 8 * blk)] = c[d];
      }
    }
  }
}

__global__ __launch_bounds__ (256) void k_expand (WhSeqParams P, const WhPicJob* jobs) {
  const WhPicJob* J = &jobs[blockIdx.y];
  WH_G uint8_t* r0 = (WH_G uint8_t*)J->rec[0];
  WH_G uint8_t* r1 = (WH_G uint8_t*)J->rec[1];
  WH_G uint8_t* r2 = (WH_G uint8_t*)J->rec[2];
  const int total = wh_expand_items (P);
  for (int idx = (int) (blockIdx.x * blockDim.x + threadIdx.x); idx < total; idx += (int) (gridDim.x * blockDim.x)) wh_expand_item (P, r0, r1, r2, idx);
}

// The tiled twin of a picture that has just become a reference (kernels/tile_pic.h): one 16-byte tile row per thread.
// A picture that the deblocking pass has written (WhPicJob::rec_blk) only has its border tiles left: the workgroups beyond those leave at once.
__global__ __launch_bounds__ (256) void k_tile (WhSeqParams P, const WhPicJob* jobs) {
  const bool border_only = jobs[blockIdx.y].rec_blk != nullptr;
  const int idx = (int) (blockIdx.x * blockDim.x + threadIdx.x), items = border_only ? wh_tile_border_items (P) : wh_tile_items (P);
  if ((int) (blockIdx.x * blockDim.x) >= items) return;
  __shared__ WhPicJob Jl;
  wh_copy_job (&Jl, &jobs[blockIdx.y]);
  __syncthreads();
  if (idx >= items) return;
  if (border_only) wh_tile_border_item (P, Jl, idx); else wh_tile_item (P, Jl, idx);
}

// A source picture as uploaded -> macroblock tiles (kernels/tile_pic.h wh_src_tile_item): 16 bytes of the tiled picture per thread.
__global__ __launch_bounds__ (256) void k_src_tile (WhSeqParams P, const uint8_t* planar, uint8_t* tiled) {
  const int idx = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  if (idx < wh_src_tile_items (P)) wh_src_tile_item (P, (const WH_G uint8_t*)planar, (WH_G uint8_t*)tiled, idx);
}

__global__ __launch_bounds__ (256) void k_src_tile_jobs (WhSeqParams P, const WhPicJob* jobs) {
  const int idx = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  const WhPicJob& J = jobs[blockIdx.y];
  if (idx < wh_src_tile_items (P) && J.src[1]) wh_src_tile_item (P, (const WH_G uint8_t*)J.src[1], (WH_G uint8_t*)J.src[0], idx);
}

// Pre-analysis statistics (kernels/vaa_pic.h): one thread per macroblock.
__global__ __launch_bounds__ (64) void k_vaa (int num_mb, const uint8_t* cur, const uint8_t* ref, WhVaaOut o) {
  const int xy = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  if (xy < num_mb) wh_vaa_mb ((const WH_G uint8_t*)cur, (const WH_G uint8_t*)ref, xy, o);
}

__global__ __launch_bounds__ (64) void k_vaa_skewed (int vw, int vh, int mb_w, int stride, int width, const uint8_t* cur, const uint8_t* ref, WhVaaOut o) {
  const int t = (int) (blockIdx.x * blockDim.x + threadIdx.x);
  if (t < vw * vh) wh_vaa_mb_skewed ((const WH_G uint8_t*)cur, (const WH_G uint8_t*)ref, stride, width, mb_w, t % vw, t / vw, o);
}

// Background detection (kernels/bgd_pic.h): ONE workgroup per picture walks the diagonals of the in-place pass, verdicts in LDS.
__global__ __launch_bounds__ (256) void k_bgd (WhBgdIn I, int8_t* mbflag) {
  extern __shared__ __align__ (16) uint8_t bgd_fl[];
  const int n = I.w * I.h;
  for (int k = (int)threadIdx.x; k < n; k += (int)blockDim.x) bgd_fl[k] = (uint8_t)wh_bgd_coarse (wh_bgd_ou (I, k % I.w, k / I.w));
  __syncthreads();
  const int steps = wh_bgd_steps (I.w, I.h);
  for (int t = 0; t < steps; ++t) {
    for (int j = (int)threadIdx.x; j < I.h; j += (int)blockDim.x) {
      const int i = t - 2 * j;
      if (i >= 0 && i < I.w) wh_bgd_step (I, bgd_fl, (WH_G int8_t*)mbflag, i, j);
    }
    __syncthreads();
  }
}

// Scene-change statistic: one wavefront per 16x16 region of the source picture.
__global__ __launch_bounds__ (64) void k_scene (WhSeqParams P, const WhPicJob* jobs) {
  const WhPicJob J = jobs[blockIdx.y];
  wh_scene_mb_body (P, J, (int) (blockIdx.x % (unsigned)P.mb_w), (int) (blockIdx.x / (unsigned)P.mb_w));
}

// QP_Y chain for the deblocking filter of pictures with a per-MB QP map: one wavefront per (slice, picture).
__global__ __launch_bounds__ (64) void k_qp_chain (WhSeqParams P, const WhPicJob* jobs) {
  const WhPicJob J = jobs[blockIdx.y];
  wh_qp_chain_slice (P, J, P.slice_first_mb[blockIdx.x], P.slice_first_mb[blockIdx.x + 1]);
}

// A drop-in library must never take the host application down: a failing HIP call is recorded (first one wins), the
// operation becomes a no-op, and sync() -- which every caller checks -- reports it.  alloc() returns NULL.
#define HIP_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) note_error (_e, #x, __LINE__); } while (0)

class HipBackend : public wh::Backend {
 public:
  HipBackend (int dev, const hipDeviceProp_t& prop) : dev_ (dev), cus_ (prop.multiProcessorCount) {
    HIP_TRY (hipSetDevice (dev_));
    // Every queue the host side can ask for exists from here on (frame API: 8 upload queues + 2 x 8 lane queues; session groups: a
    // handful): streams_ never grows, so a thread may wait on queue k (sync_queue, outside the callers' lock) while another one
    // selects a queue.  Each queue has an error word of its own (WH_ERR_WORDS dwords): a dependency-wait timeout in one launch set
    // must fail that set's pictures and nobody else's.
    // HIP streams share hardware queues (four by default, GPU_MAX_HW_QUEUES): measured with tools/micro/stream_alias.hip, the n-th stream
    // a process creates is served by hardware queue 3 - n % 4 -- except the first four, which get one each in creation order -- and work on
    // two streams of one hardware queue serialises (a copy's barrier packet waits behind the other stream's kernel: profiles/
    // r03_stream_hardware_queues.txt).  Four throw-away streams take the irregular positions, so that for the queues of this backend
    // "k % 4 differs" means "different hardware queue" whatever was created before it; the host side chooses the queues of roles that
    // must overlap accordingly (csrc/host/encoder.cpp: frame_find_key; compute / upload / download of a pipelined group: 0 / 1 / 2).
    for (int k = 0; k < 4; ++k) HIP_TRY (hipStreamCreateWithFlags (&pad_streams_[k], hipStreamNonBlocking));
    streams_.assign (WH_NUM_QUEUES, nullptr);
    for (int k = 0; k < WH_NUM_QUEUES; ++k) HIP_TRY (hipStreamCreateWithFlags (&streams_[k], hipStreamNonBlocking));      // (stream priorities were tried: no effect beyond the queue choice)
    stream_ = streams_[0]; cur_ = 0;
    HIP_TRY (hipMalloc ((void**)&err_, 4 * WH_ERR_WORDS * WH_NUM_QUEUES));
    if (err_) HIP_TRY (hipMemset (err_, 0, 4 * WH_ERR_WORDS * WH_NUM_QUEUES));
    HIP_TRY (hipMalloc ((void**)&cp_totals_, sizeof (uint32_t) * WH_NUM_QUEUES * WH_CP_MAX_CHUNKS * (size_t) (cus_ / 2 + 1)));
    name_ = std::string ("hip:") + prop.gcnArchName + " " + prop.name;
  }
  ~HipBackend() override {
    (void)hipSetDevice (dev_);
    for (hipStream_t st : streams_) if (st) { (void)hipStreamSynchronize (st); (void)hipStreamDestroy (st); }
    for (hipStream_t st : pad_streams_) if (st) (void)hipStreamDestroy (st);
    for (hipEvent_t ev : wait_ev_) if (ev) (void)hipEventDestroy (ev);
    for (auto& sl : slabs_) (void)hipFree (sl.base);
    if (err_) (void)hipFree (err_);
    if (cp_totals_) (void)hipFree (cp_totals_);
    for (uint32_t* p : split_sched_) if (p) (void)hipFree (p);
  }
  bool usable() const { return hip_err_.load (std::memory_order_relaxed) == (int)hipSuccess && stream_ && err_; }
  uint32_t* err_words() const { return err_ + WH_ERR_WORDS * cur_; }         // the selected queue's error word
  const char* name() const override { return name_.c_str(); }
  // Device memory comes from a few large slabs (4 KB granules): many small hipMalloc's end up as many small page-table
  // fragments, and with dozens of planes touched per macroblock the translation misses cost more than the data misses.
  // Inside the slabs: first fit over an address-ordered free list with coalescing, else bump allocation; a new slab is
  // sized by what has been asked for so far (16 MB .. 256 MB), so one small session does not pin a quarter of a GB.
  void* alloc (size_t bytes) override {
    if (hipSetDevice (dev_) != hipSuccess) return nullptr;
    bytes = (bytes + 4095) & ~(size_t)4095;
    if (bytes == 0) bytes = 4096;
    for (auto it = free_.begin(); it != free_.end(); ++it) if (it->second >= bytes) {
      uint8_t* p = (uint8_t*)it->first;
      const size_t rest = it->second - bytes;
      free_.erase (it);
      if (rest) free_[(uintptr_t) (p + bytes)] = rest;
      live_[(uintptr_t)p] = bytes;
      return p;
    }
    if (slabs_.empty() || slabs_.back().used + bytes > slabs_.back().size) {
      size_t want = std::min (std::max (total_asked_, (size_t)16 << 20), (size_t)256 << 20);
      want = std::max (want, bytes);
      void* p = nullptr;
      if (hipMalloc (&p, want) != hipSuccess) {
        (void)hipGetLastError();
        if (want == bytes || hipMalloc (&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }     // out of device memory: the caller fails its call
        want = bytes;
      }
      if (!slabs_.empty() && slabs_.back().size > slabs_.back().used)       // the tail of the previous slab stays usable
        free_[(uintptr_t) (slabs_.back().base + slabs_.back().used)] = slabs_.back().size - slabs_.back().used;
      if (!slabs_.empty()) slabs_.back().used = slabs_.back().size;
      slabs_.push_back (Slab {(uint8_t*)p, want, 0});
    }
    Slab& sl = slabs_.back();
    void* r = sl.base + sl.used;
    sl.used += bytes;
    total_asked_ += bytes;
    live_[(uintptr_t)r] = bytes;
    return r;
  }
  void free (void* p) override {
    if (!p) return;
    auto it = live_.find ((uintptr_t)p);
    if (it == live_.end()) return;
    uintptr_t a = it->first;
    size_t n = it->second;
    live_.erase (it);
    auto nx = free_.lower_bound (a);
    if (nx != free_.end() && a + n == nx->first && same_slab (a, nx->first)) { n += nx->second; nx = free_.erase (nx); }
    if (nx != free_.begin()) { auto pv = std::prev (nx); if (pv->first + pv->second == a && same_slab (pv->first, a)) { pv->second += n; return; } }
    free_[a] = n;
  }
  void upload (void* dst, const void* src, size_t bytes) override { if (dst && usable()) HIP_TRY (hipMemcpyAsync (dst, src, bytes, hipMemcpyHostToDevice, stream_)); else if (!dst) note_null(); }
  void pin_host (void* p, size_t bytes) override { if (hipHostRegister (p, bytes, hipHostRegisterDefault) != hipSuccess) (void)hipGetLastError(); }
  void unpin_host (void* p) override { if (hipHostUnregister (p) != hipSuccess) (void)hipGetLastError(); }
  void download (void* dst, const void* src, size_t bytes) override { if (src && usable()) HIP_TRY (hipMemcpyAsync (dst, src, bytes, hipMemcpyDeviceToHost, stream_)); else if (!src) note_null(); }
  void fill (void* dst, int value, size_t bytes) override { if (dst && usable()) HIP_TRY (hipMemsetAsync (dst, value, bytes, stream_)); else if (!dst) note_null(); }

  // waves per workgroup: bounded by the LDS budget (160 KB per CU), the kernel's register budget and by how many MBs
  // of one slice can be in flight at all (~ min(rows, mb_w / 2))
  // `static_lds`: LDS the kernel declares statically (counts against the 160 KB of a CU as well)
  // `bands` > 0: the grid's x dimension are that many deblocking bands (at most WH_DB_BAND_ROWS rows each) instead of the slices
  template <class K> void mb_pass (K kernel, size_t lds_per_wave, int max_waves, bool whole_picture, const WhSeqParams& P, const WhPicJob* jobs, int n, size_t static_lds = 0, size_t extra_dyn = 0, int bands = 0) {
    const int num_mb = P.mb_w * P.mb_h;
    int max_n = whole_picture ? num_mb : 0, max_rows = whole_picture ? P.mb_h : 0;
    if (bands > 0) { max_rows = P.db_max_rows; max_n = P.db_max_mbs; }
    else if (!whole_picture) for (int s = 0; s < P.num_slices; ++s) {
      const int cnt = P.slice_first_mb[s + 1] - P.slice_first_mb[s];
      if (cnt > max_n) max_n = cnt;
      const int rows = (P.slice_first_mb[s + 1] - 1) / P.mb_w - P.slice_first_mb[s] / P.mb_w + 1;
      if (rows > max_rows) max_rows = rows;
    }
    const size_t sched_bytes = 4 * (size_t) (1 + ((max_n + 31) >> 5)) + extra_dyn;
    int nw = max_waves;
    const int par = std::max (1, std::min (max_rows, (P.mb_w + 1) / 2));
    if (nw > par) nw = par;
    while (nw > 1 && (size_t)nw * lds_per_wave + sched_bytes + static_lds > (size_t)160 * 1024) --nw;
    const size_t lds = (size_t)nw * lds_per_wave + sched_bytes;
    set_dynamic_lds ((const void*)kernel, lds);
    if (trace_) { fprintf (stderr, "welship: launch grid %d x %d, %d waves, %zu B LDS, max_n %d\n", whole_picture ? 1 : P.num_slices, n, nw, lds, max_n); fflush (stderr); }
    hipLaunchKernelGGL (kernel, dim3 (bands > 0 ? bands : whole_picture ? 1 : P.num_slices, n), dim3 (nw * 64), lds, stream_, P, jobs, err_words());
    HIP_TRY (hipGetLastError());
    if (trace_) { HIP_TRY (hipStreamSynchronize (stream_)); fprintf (stderr, "welship: launch done\n"); fflush (stderr); }
  }
  // The dynamic-LDS limit of a kernel is raised once per size it has not had yet, and whether launches are traced is read when the backend is
  // made -- not a driver call and two getenv per launch (round-5 review: they sit on the latency path of a single session).
  void set_dynamic_lds (const void* kernel, size_t lds) {
    for (auto& e : lds_attr_) if (e.first == kernel) { if (lds > e.second) { HIP_TRY (hipFuncSetAttribute (kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); e.second = lds; } return; }
    HIP_TRY (hipFuncSetAttribute (kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_attr_.push_back ({kernel, lds});
  }
  std::vector<std::pair<const void*, size_t>> lds_attr_;
  const bool trace_ = getenv ("WELSHIP_TRACE") != nullptr;
  // I pictures: one workgroup per slice.  Few slices in flight (the latency regime, or an all-intra stream of a few sessions): 16 waves each, every
  // macroblock the wavefront has ready gets a wave.  Enough slices to give every CU two: 12 waves each -- a slice has at most ~17 macroblocks
  // ready at a time and a wave that holds a ticket waits for its neighbours, so sixteen waves idled a quarter of their time (32 k of 141 k
  // cycles per macroblock in the dependency wait).  Measured, IDR step of 256 four-slice 1080p pictures (profiles/r04_intra_waves_per_workgroup.txt):
  // 16 waves 31.3 ms, 12: 24.4 ms, 8 (four workgroups per CU): 25.7 ms, 6: 30.7 ms.  WELSHIP_I_WAVES forces the count.
  void run_intra (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    static const int forced = getenv ("WELSHIP_I_WAVES") ? atoi (getenv ("WELSHIP_I_WAVES")) : 0;
    const int waves = forced > 0 ? std::min (forced, 16) : (P.num_slices * n >= 2 * cus_ ? 12 : 16);
    if (P.flags & WH_SEQ_CHAIN) mb_pass (k_intra_slice_gom, sizeof (WhMbLds), waves, false, P, jobs, n, sizeof (WhPicJob));
    else mb_pass (k_intra_slice, sizeof (WhMbLds), waves, false, P, jobs, n, sizeof (WhPicJob));
  }
  // P pictures: one workgroup per CU-load of slices.  Few slices (latency regime): one slice per workgroup, 12 waves.  Enough
  // slices to fill the chip twice: groups of 2..4 slices share a 12-wave workgroup (k_inter_pool), dealt out by k_md_assign.
  // The group size is slices / CUs ROUNDED UP (round 6; rounds 2-5 rounded down): every workgroup then starts at once instead of a second, partly filled round
  // behind the first -- measured, 1080p pictures of four slices per launch: 80 pictures 7.17 -> 4.47 ms, 96: 7.45 -> 4.49, 160: 7.18 -> 5.60, 224: 10.46 -> 7.12
  // (profiles/r06_md_slots_rounded_up.txt); 64, 128, 192, 256 pictures are the same either way.
  // WELSHIP_MD_SLOTS = 1..4 forces the group size, WELSHIP_P_WAVES the wave count.
  void run_inter (const WhSeqParams& Pin, const WhPicJob* jobs, int n) override {
    const bool plain = WH_PLAIN_KERNEL && (Pin.flags & WH_SEQ_PLAIN) != 0, no_ctrl = WH_FRAME_KERNEL && (Pin.flags & WH_SEQ_NO_CTRL) != 0;
    WhSeqParams Pm = Pin;
    Pm.flags &= ~ (WH_SEQ_PLAIN | WH_SEQ_NO_CTRL);
    const WhSeqParams& P = Pm;
    static const int forced_waves = getenv ("WELSHIP_P_WAVES") ? atoi (getenv ("WELSHIP_P_WAVES")) : 0;
    static const int forced_slots = getenv ("WELSHIP_MD_SLOTS") ? atoi (getenv ("WELSHIP_MD_SLOTS")) : 0;
    const int use_assign = 1;            // slices dealt out by their previous cost (84 -> 94.5 % of the launch span with live waves on real content)
    const int total = P.num_slices * n;
    int slots = forced_slots > 0 ? std::min (forced_slots, WH_MD_MAX_SLOTS) : std::max (1, std::min (WH_MD_MAX_SLOTS, (total + cus_ - 1) / cus_));
    const int groups = (total + slots - 1) / slots;
    int max_n = 0, max_rows = 0;
    for (int s = 0; s < P.num_slices; ++s) {
      max_n = std::max (max_n, P.slice_first_mb[s + 1] - P.slice_first_mb[s]);
      max_rows = std::max (max_rows, (P.slice_first_mb[s + 1] - 1) / P.mb_w - P.slice_first_mb[s] / P.mb_w + 1);
    }
    const int sched_words = 1 + ((max_n + 31) >> 5);
    int nw = forced_waves > 0 ? forced_waves : WH_P_WAVES_DEFAULT;
    nw = std::min (nw, 16);
    int par = std::max (1, std::min (max_rows, (P.mb_w + 1) / 2)) * slots;       // macroblocks that can be in flight at all
    if (P.flags & WH_SEQ_SERIAL) par = 2 * slots;      // one macroblock of a slice at a time; a second wave has the next one's inputs in flight
    if (P.flags & WH_SEQ_SCC) nw = std::min (nw, 6);    // the screen-content variant needs 216 VGPRs: six waves per workgroup, no scratch
    nw = std::min (nw, par);
    // LDS of a workgroup: the waves' tiles (dynamic) + the windows and staging areas of as many waves as the kernel variant is built for
    // (static arrays: 6 / 12 / 14 / 16) + the job descriptors and the scheduler's words.  10.0 KB per wave since round 5 = 16 waves per CU
    // (rounds 1-4: 12.7 KB = 12 waves).
    auto built_for = [] (int w) { return w <= 6 ? 6 : w <= 12 ? 12 : w <= 14 ? 14 : 16; };
    const size_t fixed = WH_MD_MAX_SLOTS * (sizeof (WhPicJob) + 32) + 4 * (size_t)slots * sched_words;
    while (nw > 1 && (size_t)nw * sizeof (WhInterLds) + (size_t)built_for (nw) * (sizeof (WhInterStage) + sizeof (WhWinLds)) + fixed > (size_t)160 * 1024) --nw;
    const size_t lds = (size_t)nw * sizeof (WhInterLds) + 4 * (size_t)slots * sched_words;
    // Few slices in the launch (one or a few sessions through the frame API): every slice on `parts` CUs (k_inter_split), as many waves in all as the slice
    // can have macroblocks in flight (+ a few that hold their next ticket), at most six per CU.  WELSHIP_MD_SPLIT = 0 switches it off, n > 1 forces the parts.
    {
      static const int split_env = getenv ("WELSHIP_MD_SPLIT") ? atoi (getenv ("WELSHIP_MD_SPLIT")) : -1;
      const int variant = (WH_PLAIN_KERNEL == 2 && plain && P.flags == 0 && P.complexity == 0) ? 2 : (WH_PLAIN_KERNEL && plain && P.flags == 0) ? 1 : (WH_FRAME_KERNEL && no_ctrl && P.flags == 0) ? 3 : 0;
      int parts = split_env == 0 ? 1 : std::min (8, cus_ / std::max (1, total));
      if (split_env > 1) parts = std::min (split_env, 8);
      const int par1 = std::max (1, std::min (max_rows, (P.mb_w + 1) / 2));
      parts = std::min (parts, (par1 + 3 + 3) / 4);            // (no more CUs than waves to fill them with four each)
      uint32_t* xs = (variant != 0 && parts >= 2 && forced_slots <= 0) ? split_sched ((size_t)total * sched_words) : nullptr;
      if (xs) {
        const int xw = std::max (2, std::min (WH_SPLIT_WAVES, (par1 + 3 + parts - 1) / parts));
        const size_t xlds = (size_t)xw * sizeof (WhInterLds);
        const int blocks = ((total + 7) / 8) * 8 * parts;
        HIP_TRY (hipMemsetAsync (xs, 0, 4 * (size_t)total * sched_words, stream_));
        auto xlaunch = [&] (auto kernel) {
          set_dynamic_lds ((const void*)kernel, xlds);
          if (trace_) { fprintf (stderr, "welship: MD launch, %d slices on %d CUs each, %d waves per CU\n", total, parts, xw); fflush (stderr); }
          hipLaunchKernelGGL (kernel, dim3 (blocks), dim3 (xw * 64), xlds, stream_, P, jobs, err_words(), xs, sched_words, parts, total);
          HIP_TRY (hipGetLastError());
          if (trace_) { HIP_TRY (hipStreamSynchronize (stream_)); fprintf (stderr, "welship: launch done\n"); fflush (stderr); }
        };
        if (variant == 2) xlaunch (k_inter_split<2>); else if (variant == 1) xlaunch (k_inter_split<1>); else xlaunch (k_inter_split<3>);
        return;
      }
    }
    uint16_t* grp = nullptr;
    uint32_t* cost = nullptr;
    if (slots > 1 && use_assign && total <= 4096 && stream_ == streams_[0]) {      // the cost / assignment buffers belong to queue 0
      if ((size_t)total > md_cap_) {
        if (md_cost_) { free (md_cost_); free (md_groups_); }
        md_cap_ = (size_t)total;
        md_cost_ = (uint32_t*)alloc (4 * md_cap_);
        md_groups_ = (uint16_t*)alloc (2 * (md_cap_ + WH_MD_MAX_SLOTS));
        if (md_cost_) HIP_TRY (hipMemsetAsync (md_cost_, 0, 4 * md_cap_, stream_));
      }
      if (md_cost_ && md_groups_) {
        grp = md_groups_; cost = md_cost_;
        hipLaunchKernelGGL (k_md_assign, dim3 (1), dim3 (1024), 0, stream_, cost, grp, total, groups, slots);
        HIP_TRY (hipGetLastError());
      }
    }
    const WhSeqParams& Pr = P;
    auto launch = [&] (auto kernel) {
      set_dynamic_lds ((const void*)kernel, lds);
      if (trace_) { fprintf (stderr, "welship: MD launch %d groups x %d slots, %d waves, %zu B dynamic LDS\n", groups, slots, nw, lds); fflush (stderr); }
      hipLaunchKernelGGL (kernel, dim3 (groups), dim3 (nw * 64), lds, stream_, Pr, jobs, err_words(), (const uint16_t*)grp, slots, sched_words, total, cost);
      HIP_TRY (hipGetLastError());
      if (trace_) { HIP_TRY (hipStreamSynchronize (stream_)); fprintf (stderr, "welship: launch done\n"); fflush (stderr); }
    };
    if (P.flags & WH_SEQ_SCC) launch (k_inter_pool<384, true>);      // (nw <= 6 above: 249 VGPRs, no scratch; a 12-wave build of this variant spills -- 168 VGPRs + 360 B -- and is not instantiated any more)
#define WH_LAUNCH_POOL(VARIANT) do { if (nw <= 6) launch (k_inter_pool<384, false, VARIANT>); else if (nw <= 12) launch (k_inter_pool<768, false, VARIANT>); \
                                     else if (nw <= 14) launch (k_inter_pool<896, false, VARIANT>); else launch (k_inter_pool<1024, false, VARIANT>); } while (0)
    else if (WH_PLAIN_KERNEL == 2 && plain && P.flags == 0 && P.complexity == 0) WH_LAUNCH_POOL (WH_PLAIN_KERNEL == 2 ? 2 : 0);
    else if (WH_PLAIN_KERNEL && plain && P.flags == 0) WH_LAUNCH_POOL (WH_PLAIN_KERNEL ? 1 : 0);
    else if (WH_FRAME_KERNEL && no_ctrl && P.flags == 0) WH_LAUNCH_POOL (WH_FRAME_KERNEL ? 3 : 0);
    else WH_LAUNCH_POOL (0);
#undef WH_LAUNCH_POOL
  }
  void run_deblock (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    const int db_waves = 12;    // 73 VGPRs: two 12-wave workgroups per CU, one of 16 (measured 3.85 against 4.79 ms per step of 256 pictures; 8: 4.3, 6: 5.1)
    // Enough pictures to give every CU one (and a filter that crosses slice edges anyway): ONE band per picture, 16 waves -- no seams between
    // workgroups at all (measured, 256 four-slice 1080p pictures: 3.17 against 3.88 ms per step; profiles/r03_deblock_bands.txt).
    // With two macroblocks per wavefront (k_deblock_pairs, round 6) one band per picture wins from 96 pictures on (1.92 against 2.10 ms per step; 128: 2.00 against 2.12,
    // 160: 2.09 against 2.79; 64: 1.80 against 1.59 -- profiles/r06_deblock_two_macroblocks_per_wave_ab.txt): 8 x pictures >= 3 x CUs (rounds 3-5: 4 x).
    // WELSHIP_DB_WHOLE = 0 / 1 forces the choice (read per launch: the tests switch it).
    const char* we = getenv ("WELSHIP_DB_WHOLE");
    const bool whole = P.deblock_idc == 0 && P.db_bands && (P.flags & WH_SEQ_DB_WHOLE) == 0 && (we ? atoi (we) != 0 : 8 * n >= 3 * cus_);
    if (whole) {
      WhSeqParams W = P;
      W.flags |= WH_SEQ_DB_WHOLE; W.db_num_bands = 1; W.db_bands = WH_DB_WHOLE_TABLE (P); W.db_max_rows = P.mb_h; W.db_max_mbs = P.mb_w * P.mb_h;
      // ... two macroblocks per wavefront where the order pairs them up (k_deblock_pairs).  WELSHIP_DB_PAIRS=0: the one-macroblock kernel (A/B).
      static const bool pairs = !(getenv ("WELSHIP_DB_PAIRS") && atoi (getenv ("WELSHIP_DB_PAIRS")) == 0);
      // (sixteen waves: the pass's time goes with 1 / waves -- 12: 2.52, 14: 2.32, 16: 2.13 ms per step of 256 1080p pictures)
      const size_t items_lds = (WH_DB_ITEMS_IN_LDS && P.mb_w * P.mb_h <= WH_DB_ITEMS_LDS_MAX_MB) ? 4 * ((size_t)P.mb_w * P.mb_h + 1) : 0;
      if (pairs) mb_pass (k_deblock_pairs, 2 * sizeof (WhDbLds), 16, false, W, jobs, n, 32 * sizeof (WhDbStage) + sizeof (WhPicJob), 4 * wh_db_xchg_words (P.mb_w, P.mb_h) + items_lds, 1);
      else mb_pass (k_deblock_slices, sizeof (WhDbLds), 16, false, W, jobs, n, 16 * sizeof (WhDbStage) + sizeof (WhPicJob), 4 * wh_db_xchg_words (P.mb_w, P.mb_h), 1);
      return;
    }
    mb_pass (k_deblock_slices, sizeof (WhDbLds), db_waves, false, P, jobs, n, 16 * sizeof (WhDbStage) + sizeof (WhPicJob), 4 * wh_db_xchg_words (P.mb_w, P.mb_h), P.db_num_bands);
  }
  void run_scene (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    hipLaunchKernelGGL (k_scene, dim3 (P.mb_w * P.mb_h, n), dim3 (64), 0, stream_, P, jobs);
    HIP_TRY (hipGetLastError());
  }
  void run_qp_chain (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    hipLaunchKernelGGL (k_qp_chain, dim3 (P.num_slices, n), dim3 (64), 0, stream_, P, jobs);
    HIP_TRY (hipGetLastError());
  }
  void run_expand (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    const int blocks = std::max (1, std::min (64, (wh_expand_items (P) + 1023) / 1024));        // ~4 items per thread, at most 64 workgroups per picture
    hipLaunchKernelGGL (k_expand, dim3 (blocks, n), dim3 (256), 0, stream_, P, jobs);
    HIP_TRY (hipGetLastError());
    // ... and, as part of making the picture a reference, its tiled twin (what the next picture's search windows are fetched from)
    hipLaunchKernelGGL (k_tile, dim3 ((wh_tile_items (P) + 255) / 256, n), dim3 (256), 0, stream_, P, jobs);
    HIP_TRY (hipGetLastError());
  }
  void run_src_tile (const WhSeqParams& P, const uint8_t* planar, uint8_t* tiled) override {
    if (!planar || !tiled) { note_null(); return; }
    hipLaunchKernelGGL (k_src_tile, dim3 ((wh_src_tile_items (P) + 255) / 256), dim3 (256), 0, stream_, P, planar, tiled);
    HIP_TRY (hipGetLastError());
  }
  void run_src_tile_jobs (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    if (!jobs || n <= 0) { note_null(); return; }
    hipLaunchKernelGGL (k_src_tile_jobs, dim3 ((wh_src_tile_items (P) + 255) / 256, n), dim3 (256), 0, stream_, P, jobs);
    HIP_TRY (hipGetLastError());
  }
  void upload_on (int q, void* dst, const void* src, size_t bytes) override {
    HIP_TRY (hipSetDevice (dev_));
    HIP_TRY (hipMemcpyAsync (dst, src, bytes, hipMemcpyHostToDevice, streams_[(q < 0 ? 0 : q) % WH_NUM_QUEUES]));
  }
  void download_on (int q, void* dst, const void* src, size_t bytes) override {
    HIP_TRY (hipSetDevice (dev_));
    HIP_TRY (hipMemcpyAsync (dst, src, bytes, hipMemcpyDeviceToHost, streams_[(q < 0 ? 0 : q) % WH_NUM_QUEUES]));
  }
  void err_snapshot (int q, uint32_t* dst) override { if (dst && usable()) { HIP_TRY (hipSetDevice (dev_)); q = (q < 0 ? 0 : q) % WH_NUM_QUEUES; HIP_TRY (hipMemcpyAsync (dst, err_ + WH_ERR_WORDS * q, 4 * WH_ERR_WORDS, hipMemcpyDeviceToHost, streams_[q])); } }
  void event_record_on (int q, void* ev) override { if (ev) { HIP_TRY (hipSetDevice (dev_)); HIP_TRY (hipEventRecord ((hipEvent_t)ev, streams_[(q < 0 ? 0 : q) % WH_NUM_QUEUES])); } }
  void queue_wait_event (int q, void* ev) override { if (ev) { HIP_TRY (hipSetDevice (dev_)); HIP_TRY (hipStreamWaitEvent (streams_[(q < 0 ? 0 : q) % WH_NUM_QUEUES], (hipEvent_t)ev, 0)); } }
  void event_wait (void* ev) override { if (ev) { HIP_TRY (hipSetDevice (dev_)); HIP_TRY (hipEventSynchronize ((hipEvent_t)ev)); } }
  void run_vaa (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, int32_t* sad8x8, int32_t* sd8x8, uint8_t* mad8x8, int32_t* sum16, int32_t* sqsum16, int32_t* ssd16) override {
    if (!cur || !ref) { note_null(); return; }
    const WhVaaOut o = {sad8x8, sd8x8, mad8x8, sum16, sqsum16, ssd16};
    const int num_mb = P.mb_w * P.mb_h;
    hipLaunchKernelGGL (k_vaa, dim3 ((num_mb + 63) / 64), dim3 (64), 0, stream_, num_mb, cur, ref, o);
    HIP_TRY (hipGetLastError());
  }
  void run_vaa_skewed (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, int stride, int width, int height, int32_t* sad8x8, int32_t* sd8x8, uint8_t* mad8x8,
                       int32_t* sum16, int32_t* sqsum16, int32_t* ssd16) override {
    if (!cur || !ref) { note_null(); return; }
    const WhVaaOut o = {sad8x8, sd8x8, mad8x8, sum16, sqsum16, ssd16};
    const int vw = width >> 4, vh = height >> 4;
    hipLaunchKernelGGL (k_vaa_skewed, dim3 ((vw * vh + 63) / 64), dim3 (64), 0, stream_, vw, vh, P.mb_w, stride, width, cur, ref, o);
    HIP_TRY (hipGetLastError());
  }
  void run_bgd (const WhSeqParams& P, const uint8_t* cur, const uint8_t* ref, const int32_t* sad8x8, const int32_t* sd8x8, const uint8_t* mad8x8, int units_w, int units_h,
                int8_t* flags) override {
    if (!cur || !ref || !sad8x8 || !sd8x8 || !mad8x8 || !flags) { note_null(); return; }
    const WhBgdIn in = {sad8x8, sd8x8, mad8x8, cur, ref, units_w, units_h, P.mb_w};
    const size_t lds = (size_t)units_w * units_h;
    set_dynamic_lds ((const void*)k_bgd, lds);
    hipLaunchKernelGGL (k_bgd, dim3 (1), dim3 (256), lds, stream_, in, flags);
    HIP_TRY (hipGetLastError());
  }
  void run_compact (const WhSeqParams& P, const WhPicJob* jobs, int n) override {
    // few pictures: chunks of a picture on as many CUs (k_compact_sizes + k_compact_chunks); a launch that fills the device anyway keeps one workgroup per picture
    const int num_mb = P.mb_w * P.mb_h, chunks = (num_mb + WH_CP_CHUNK - 1) / WH_CP_CHUNK;
    uint32_t* tot = 2 * n <= cus_ && chunks > 1 && chunks <= WH_CP_MAX_CHUNKS ? compact_totals (n) : nullptr;
    if (tot) {
      hipLaunchKernelGGL (k_compact_sizes, dim3 (chunks, n), dim3 (256), 0, stream_, P, jobs, tot);
      hipLaunchKernelGGL (k_compact_chunks, dim3 (chunks, n), dim3 (256), 0, stream_, P, jobs, (const uint32_t*)tot);
    } else
    hipLaunchKernelGGL (k_compact, dim3 (n), dim3 (1024), 0, stream_, P, jobs);
    HIP_TRY (hipGetLastError());
  }
  void queue_wait (int from) override {
    if (from < 0 || from >= (int)streams_.size() || streams_[from] == stream_ || !usable()) return;
    if (wait_ev_.empty()) wait_ev_.assign (16, nullptr);
    hipEvent_t& ev = wait_ev_[wait_next_++ % wait_ev_.size()];
    if (!ev) HIP_TRY (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
    if (!ev) return;
    HIP_TRY (hipEventRecord (ev, streams_[from]));
    HIP_TRY (hipStreamWaitEvent (stream_, ev, 0));
  }
  void select_queue (int k) override {
    HIP_TRY (hipSetDevice (dev_));
    cur_ = k < 0 ? 0 : k % WH_NUM_QUEUES;         // (queues beyond the fixed set share: they only serialise, nothing breaks)
    stream_ = streams_[cur_];
  }
  // 0, the number of in-kernel dependency waits that timed out, or -1 after a HIP error (sticky: the backend is unusable)
  int sync() override {
    HIP_TRY (hipSetDevice (dev_));          // callers may be host threads that never selected a device
    for (hipStream_t st : streams_) if (st) HIP_TRY (hipStreamSynchronize (st));
    const int bad = check_err (0, WH_NUM_QUEUES);
    if (bad > 0) swept_.fetch_add (1, std::memory_order_relaxed);     // may have consumed the verdict of a launch set another thread is waiting for
    return bad;
  }
  unsigned errors_swept() const override { return swept_.load (std::memory_order_relaxed); }
  int peek_queue_errors (int k, int via) override { HIP_TRY (hipSetDevice (dev_)); return check_err (k < 0 ? 0 : k % WH_NUM_QUEUES, 1, streams_[(via < 0 ? 0 : via) % WH_NUM_QUEUES]); }
  // queue k only: its stream, its error word (the other queues' launch sets keep running and keep their own verdicts)
  int sync_queue (int k) override {
    HIP_TRY (hipSetDevice (dev_));
    k = k < 0 ? 0 : k % WH_NUM_QUEUES;
    if (streams_[k]) HIP_TRY (hipStreamSynchronize (streams_[k]));
    return check_err (k, 1, streams_[k]);
  }
  void* event_create() override { hipEvent_t e = nullptr; HIP_TRY (hipEventCreate (&e)); return (void*)e; }
  void event_destroy (void* ev) override { if (ev) HIP_TRY (hipEventDestroy ((hipEvent_t)ev)); }
  void event_record (void* ev) override { if (ev) HIP_TRY (hipEventRecord ((hipEvent_t)ev, stream_)); }
  float event_elapsed_ms (void* a, void* b) override { float ms = 0.f; if (!a || !b) return ms; HIP_TRY (hipEventSynchronize ((hipEvent_t)b)); HIP_TRY (hipEventElapsedTime (&ms, (hipEvent_t)a, (hipEvent_t)b)); return ms; }
  hipStream_t stream() const { return stream_; }
 private:
  // error words of queues [k0, k0 + n): everything on those queues has completed (the callers synchronised them), so the read and
  // the reset race with no kernel that could write them
  // via: the words are read through that queue (a copy on the null stream may wait for other queues' kernels)
  int check_err (int k0, int n, hipStream_t via = nullptr) {
    if (!usable()) return -1;
    uint32_t e[WH_ERR_WORDS * WH_NUM_QUEUES];
    if (via) {
      HIP_TRY (hipMemcpyAsync (e, err_ + WH_ERR_WORDS * k0, 4 * WH_ERR_WORDS * n, hipMemcpyDeviceToHost, via));
      HIP_TRY (hipStreamSynchronize (via));
    } else
    HIP_TRY (hipMemcpy (e, err_ + WH_ERR_WORDS * k0, 4 * WH_ERR_WORDS * n, hipMemcpyDeviceToHost));
    if (!usable()) return -1;
    int total = 0;
    for (int i = 0; i < n; ++i) {
      const uint32_t* w = e + WH_ERR_WORDS * i;
      if (!w[0]) continue;
      total += (int)w[0];
      fprintf (stderr, "welship: queue %d: %u in-kernel dependency waits timed out (first: block %u,%u waiting for MB index %u)\n", k0 + i, w[0], w[1], w[2], w[3]);
      HIP_TRY (hipMemset (err_ + WH_ERR_WORDS * (k0 + i), 0, 4 * WH_ERR_WORDS));
    }
    return total;
  }
  // the chunk totals of run_compact's two launches: one array per queue (launch sets of different queues overlap), for up to cus_ / 2 pictures each;
  // allocated with the backend (run_compact is called from whichever host thread runs a launch set)
  uint32_t* compact_totals (int n) const { return cp_totals_ && n <= cus_ / 2 + 1 ? cp_totals_ + (size_t)cur_ * WH_CP_MAX_CHUNKS * (size_t) (cus_ / 2 + 1) : nullptr; }
  uint32_t* cp_totals_ = nullptr;
  // the scheduler's words of k_inter_split: one array per queue, allocated the first time a queue launches it (any host thread)
  uint32_t* split_sched (size_t words) {
    if (words > kSplitWords) return nullptr;
    std::lock_guard<std::mutex> lk (split_mu_);
    uint32_t*& p = split_sched_[cur_];
    if (!p && hipMalloc ((void**)&p, 4 * kSplitWords) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    return p;
  }
  static constexpr size_t kSplitWords = (size_t)128 * 1160;         // 128 slices of a 4096 x 2304 picture's macroblocks
  uint32_t* split_sched_[WH_NUM_QUEUES] = {};
  std::mutex split_mu_;
  int dev_;
  int cus_;
  std::vector<hipEvent_t> wait_ev_;       // queue_wait: a small ring of events
  size_t wait_next_ = 0;
  hipStream_t stream_ = nullptr;          // the selected queue ...
  int cur_ = 0;                           // ... and its index
  std::vector<hipStream_t> streams_;      // WH_NUM_QUEUES of them, fixed at construction
  hipStream_t pad_streams_[4] = {nullptr, nullptr, nullptr, nullptr};     // never used (see the constructor)
  uint32_t* err_ = nullptr;
  struct Slab { uint8_t* base; size_t size, used; };
  std::vector<Slab> slabs_;
  std::map<uintptr_t, size_t> free_;                     // address -> bytes, coalesced
  std::unordered_map<uintptr_t, size_t> live_;           // what alloc() handed out
  size_t total_asked_ = 0;
  std::atomic<unsigned> swept_ {0};       // how often sync() found (and reset) error words: see Backend::errors_swept
  std::atomic<int> hip_err_ {(int)hipSuccess};       // first HIP error (sticky); read by threads that wait outside the callers' lock
  std::string name_;
  uint32_t* md_cost_ = nullptr;          // per slice of a batch: cost of its previous picture (k_inter_pool -> k_md_assign)
  uint16_t* md_groups_ = nullptr;
  size_t md_cap_ = 0;
  bool same_slab (uintptr_t a, uintptr_t b) const {
    for (const Slab& sl : slabs_) { const uintptr_t lo = (uintptr_t)sl.base, hi = lo + sl.size; if (a >= lo && a < hi) return b >= lo && b < hi; }
    return false;
  }
  void note_error (hipError_t e, const char* what, int line) {
    (void)hipGetLastError();
    int expected = (int)hipSuccess;
    if (!hip_err_.compare_exchange_strong (expected, (int)e)) return;
    fprintf (stderr, "welship: HIP error %s in %s (hip_backend.hip:%d); the session reports failure\n", hipGetErrorString (e), what, line);
  }
  void note_null() { int expected = (int)hipSuccess; if (hip_err_.compare_exchange_strong (expected, (int)hipErrorOutOfMemory)) { fprintf (stderr, "welship: device memory exhausted; the session reports failure\n"); } }
};

}  // namespace

namespace wh {

Backend* create_hip_backend (int device, const char** err) {
  int count = 0;
  if (hipGetDeviceCount (&count) != hipSuccess || count <= 0) { if (err) *err = "no HIP device visible (libwelship needs an MI355X; there is no CPU fallback)"; return nullptr; }
  if (device < 0 || device >= count) { if (err) *err = "HIP device ordinal out of range"; return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties (&prop, device) != hipSuccess) { if (err) *err = "hipGetDeviceProperties failed"; return nullptr; }
  if (std::string (prop.gcnArchName).find ("gfx950") == std::string::npos) { if (err) *err = "device is not gfx950 (this library is built for MI355X only)"; return nullptr; }
  HipBackend* be = new HipBackend (device, prop);
  if (!be->usable()) { delete be; if (err) *err = "HIP stream / memory set-up failed on the device"; return nullptr; }
  return be;
}
// WELSHIP_TRACE_DEVICES=1: which device index every backend is asked for (the layer -> device and rank -> device mappings: bench.py config4_layer_per_gpu)
Backend* create_default_backend (int device, const char** err) {
  if (getenv ("WELSHIP_TRACE_DEVICES")) { fprintf (stderr, "welship: backend for device %d\n", device); fflush (stderr); }
  return create_hip_backend (device, err);
}

}  // namespace wh
