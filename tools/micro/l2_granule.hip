// l2_granule.hip -- calibration for the HBM traffic figures (MI355X_MICROARCH.md, "calibrate on a known byte count in your own
// access pattern"): the mode-decision kernel fetches its search windows as 64-byte rows (4 lanes x 16 B) at arbitrary 4-byte
// alignment.  Does a 64-byte piece of a 128-byte line cost the fabric 64 or 128 bytes?  Stream N rows of 64 B out of a buffer far
// larger than L2 + Infinity Cache with row pitch 64 (contiguous), 128 (every other half line), 256, and 64-B rows that straddle two
// lines (pitch 128, offset 96); the time per row says what each row moves.  Build: hipcc --offload-arch=gfx950 -O3 -o l2_granule l2_granule.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

__global__ __launch_bounds__ (256) void k_rows (const uint8_t* __restrict__ src, size_t pitch, size_t offset, size_t rows, uint32_t* sink) {
  // thread t of the grid reads bytes [16 * (t & 3), +16) of row t >> 2
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (size_t i = t; i < rows * 4; i += stride) {
    const uint4 v = * (const uint4*) (src + (i >> 2) * pitch + offset + (i & 3) * 16);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345u) *sink = acc;
}

int main() {
  const size_t bytes = (size_t)6 << 30;         // 6 GiB >> 256 MiB Infinity Cache
  uint8_t* buf = nullptr; uint32_t* sink = nullptr;
  if (hipMalloc ((void**)&buf, bytes) != hipSuccess || hipMalloc ((void**)&sink, 4) != hipSuccess) { printf ("alloc failed\n"); return 1; }
  hipMemset (buf, 1, bytes);
  hipEvent_t a, b; hipEventCreate (&a); hipEventCreate (&b);
  struct { const char* name; size_t pitch, offset; } cases[] = {
    {"pitch 64 (contiguous stream)", 64, 0}, {"pitch 128, first half of every line", 128, 0}, {"pitch 128, straddling (offset 96)", 128, 96},
    {"pitch 256, half of every other line", 256, 0}, {"pitch 2048 (picture rows), aligned", 2048, 0}, {"pitch 2048, straddling (offset 96)", 2048, 96}};
  for (auto& c : cases) {
    const size_t rows = (bytes - 4096) / c.pitch < ((size_t)1 << 25) ? (bytes - 4096) / c.pitch : ((size_t)1 << 25);      // <= 32 M rows = 2 GiB useful
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord (a);
      hipLaunchKernelGGL (k_rows, dim3 (256 * 16), dim3 (256), 0, 0, buf, c.pitch, c.offset, rows, sink);
      hipEventRecord (b); hipEventSynchronize (b);
      float ms = 0; hipEventElapsedTime (&ms, a, b);
      if (ms < best) best = ms;
    }
    printf ("%-42s rows %9zu  %8.3f ms  useful %7.1f GB/s  ns per 1000 rows %.2f\n", c.name, rows, best, 64.0 * rows / best * 1e-6, best * 1e9 / rows * 1e-3);
  }
  return 0;
}
