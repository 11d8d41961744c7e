// xcd_stale.hip -- can a load of a word another XCD has just written return the line an earlier read left in the reader's L2?
// Two workgroups (block 0 writes, block 1 reads; blocks 0 and 1 run on different XCDs, printed).  Per 128-byte line:
//   reader loads word 0 (the line is now in the reader's L1 / L2), raises flag A; the writer waits for A, stores word 16 of the same line, raises flag B;
//   the reader waits for B and loads word 16.  Counted: loads that return the old value, per combination of store and load flavour.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_stale tools/micro/xcd_stale.hip ; run: /tmp/xcd_stale
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__ ((address_space (1))) uint32_t gu32;
template <int LM> __device__ __forceinline__ uint32_t ld (uint32_t* p) {
  if (LM == 0) return * (volatile uint32_t*)p;                                                                        // plain (behind an agent-scope acquire)
  if (LM == 1) return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                               // sc1
  return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                                          // sc0 sc1
}
template <int SM, int LM> __global__ void k (uint32_t* data, uint32_t* fa, uint32_t* fb, uint32_t* stale, uint32_t* xcc, int lines, uint32_t gen) {
  const int t = threadIdx.x;
  if (t == 0) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg ((3 << 11) | 20);
  for (int i = t; i < lines; i += blockDim.x) {
    uint32_t* line = data + (size_t)i * 32;
    if (blockIdx.x == 1) {                     // reader
      const uint32_t v0 = ld<LM> (line);       // brings the line in
      __hip_atomic_store (&fa[i], gen + (v0 & 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      int spin = 0;
      while (__hip_atomic_load (&fb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != gen && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep (2);
      if (LM == 0) __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "agent");
      const uint32_t v = ld<LM> (line + 16);
      if (v != gen * 1000u + (uint32_t)i) atomicAdd (&stale[0], 1u);
      if (spin >= (1 << 22)) atomicAdd (&stale[1], 1u);
    } else {                                   // writer
      int spin = 0;
      while (__hip_atomic_load (&fa[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != gen && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep (2);
      if (SM == 0) { line[16] = gen * 1000u + (uint32_t)i; __builtin_amdgcn_fence (__ATOMIC_RELEASE, "agent"); }
      else __hip_atomic_store (&line[16], gen * 1000u + (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store (&fb[i], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
int main() {
  const int lines = 1 << 14;
  uint32_t *data, *fa, *fb, *stale, *xcc;
  hipMalloc (&data, (size_t)lines * 128); hipMalloc (&fa, lines * 4); hipMalloc (&fb, lines * 4); hipMalloc (&stale, 8); hipMalloc (&xcc, 64);
  hipMemset (fa, 0, lines * 4); hipMemset (fb, 0, lines * 4);
  uint32_t gen = 0;
  const char* sn[2] = {"plain store + agent release", "sc0 sc1 store"};
  const char* ln[3] = {"agent acquire + plain load", "sc1 load", "sc0 sc1 load"};
  for (int rep = 0; rep < 3; ++rep) for (int sm = 0; sm < 2; ++sm) for (int lm = 0; lm < 3; ++lm) {
    ++gen;
    hipMemset (stale, 0, 8);
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    hipEventRecord (e0);
#define L(S, M) hipLaunchKernelGGL ((k<S, M>), dim3 (2), dim3 (64), 0, 0, data, fa, fb, stale, xcc, lines, gen)
    if (sm == 0) { if (lm == 0) L (0, 0); else if (lm == 1) L (0, 1); else L (0, 2); } else { if (lm == 0) L (1, 0); else if (lm == 1) L (1, 1); else L (1, 2); }
    hipEventRecord (e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime (&ms, e0, e1);
    uint32_t h[2], x[2]; hipMemcpy (h, stale, 8, hipMemcpyDeviceToHost); hipMemcpy (x, xcc, 8, hipMemcpyDeviceToHost);
    printf ("rep %d  %-28s | %-28s : stale %u of %d (timeouts %u), xcc %u / %u, %.2f us per round trip\n", rep, sn[sm], ln[lm], h[0], lines, h[1], x[0] & 15, x[1] & 15, 1000.f * ms / (lines / 64));
  }
  return 0;
}
