// lds_dma_l2.hip -- does a `global_load_lds` (LDS-DMA) request allocate its line in the XCD's L2 like a plain `global_load`?
// (round-5 review, item 5: the mode-decision kernel fetches its search windows by LDS-DMA and the fabric counters say the L2 keeps
// almost none of them although neighbouring macroblocks' windows overlap by 60-80 %.)
// Every workgroup owns REGION bytes (64 KB: 32 workgroups per XCD x 64 KB = 2 MB of a 4 MB L2; larger than the CU's 32 KB L1) and
// reads them REPS times, once with plain 16-byte loads and once with 16-byte LDS-DMA loads.  If the L2 keeps the lines the fabric sees
// each region about once (TCC_EA0_RDREQ ~ REGION x workgroups / request size); if not, REPS times.  A third pair of kernels writes a
// streaming record (1.5 KB per 7.4 KB read, the kernel's ratio) between the reads, plain and non-temporal, to see what the writes evict.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_l2 lds_dma_l2.hip
// Run:   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -- ./lds_dma_l2      (+ a FETCH_SIZE pass)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REGION (64 * 1024)
#define REPS 64
#define G __attribute__ ((address_space (1)))
#define L __attribute__ ((address_space (3)))

template <bool DMA, int WR>      // WR: 0 no writes, 1 plain stores, 2 non-temporal stores
__global__ __launch_bounds__ (256) void k_reread (const uint8_t* __restrict__ src, uint8_t* __restrict__ out, uint32_t* sink) {
  __shared__ __attribute__ ((aligned (16))) uint8_t lds[4096];          // one 4 KB piece per pass of the workgroup (256 lanes x 16 B)
  const uint8_t* region = src + (size_t)blockIdx.x * REGION;
  uint8_t* wr = out + (size_t)blockIdx.x * (size_t) (REPS * (REGION / 4096) * 1024);
  uint32_t acc = 0;
  const int wave = threadIdx.x >> 6;
  for (int rep = 0; rep < REPS; ++rep)
    for (int p = 0; p < REGION / 4096; ++p) {
      const uint8_t* s = region + p * 4096 + threadIdx.x * 16;
      if (DMA) {
        __builtin_amdgcn_global_load_lds ((const G uint32_t*)s, (L uint32_t*) (lds + wave * 1024), 16, 0, 0);
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
        acc += * (const uint32_t*) (lds + threadIdx.x * 16);
      } else {
        const uint4 v = * (const uint4*)s;
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
      if (WR) {                                                              // 1 KB written per 4 KB read, never read again
        uint32_t* d = (uint32_t*) (wr + (size_t) (rep * (REGION / 4096) + p) * 1024) + threadIdx.x;
        if (WR == 2) __builtin_nontemporal_store (acc, d); else *d = acc;
      }
    }
  if (acc == 0x12345u) *sink = acc;
}

int main() {
  const int wgs = 256;
  uint8_t* buf = nullptr; uint8_t* out = nullptr; uint32_t* sink = nullptr;
  const size_t out_bytes = (size_t)wgs * REPS * (REGION / 4096) * 1024;
  if (hipMalloc ((void**)&buf, (size_t)wgs * REGION) != hipSuccess || hipMalloc ((void**)&out, out_bytes) != hipSuccess || hipMalloc ((void**)&sink, 4) != hipSuccess) { printf ("alloc failed\n"); return 1; }
  hipMemset (buf, 1, (size_t)wgs * REGION);
  hipMemset (out, 0, out_bytes);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate (&a); hipEventCreate (&b);
  const double read_gb = (double)wgs * REGION * REPS * 1e-9;
#define RUN(NAME, ...) do { float best = 1e30f; for (int r = 0; r < 3; ++r) { hipEventRecord (a); hipLaunchKernelGGL ((k_reread<__VA_ARGS__>), dim3 (wgs), dim3 (256), 0, 0, buf, out, sink); \
    hipEventRecord (b); hipEventSynchronize (b); float ms = 0; hipEventElapsedTime (&ms, a, b); if (ms < best) best = ms; } \
    printf ("%-44s %8.3f ms  %8.1f GB/s requested (%.3f GB requested, %.3f GB unique)\n", NAME, best, read_gb / best * 1e3, read_gb, (double)wgs * REGION * 1e-9); } while (0)
  RUN ("plain 16-byte loads", false, 0);
  RUN ("LDS-DMA 16-byte loads", true, 0);
  RUN ("plain loads + plain streaming stores", false, 1);
  RUN ("LDS-DMA loads + plain streaming stores", true, 1);
  RUN ("plain loads + non-temporal stores", false, 2);
  RUN ("LDS-DMA loads + non-temporal stores", true, 2);
  return 0;
}
