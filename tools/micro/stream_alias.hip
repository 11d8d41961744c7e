// Which HIP streams share a hardware queue?  32 non-blocking streams created in order (as hip_backend.hip does); for every pair
// (busy, other): a kernel that spins for ~8 ms on `busy`, then a small H2D copy on `other` -- if the copy only completes when the kernel
// has ended, the two streams are served by the same hardware queue (the copy's barrier packet sits behind the kernel).
// hipcc --offload-arch=gfx950 -O2 -o tools/micro/stream_alias tools/micro/stream_alias.hip && tools/micro/stream_alias
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void spin (long long cycles, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (out) *out = 1;
}
int main (int argc, char** argv) {
  const int N = 32;
  std::vector<hipStream_t> st (N);
  for (int k = 0; k < N; ++k) hipStreamCreateWithFlags (&st[k], hipStreamNonBlocking);
  int* d = nullptr; hipMalloc ((void**)&d, 4096);
  int* h = nullptr; hipHostMalloc ((void**)&h, 4096, 0);
  hipLaunchKernelGGL (spin, dim3 (1), dim3 (64), 0, st[0], 1000, d); hipDeviceSynchronize();
  const long long ticks = 800000;            // 100 MHz wall clock: 8 ms
  const int busy_list[] = {0, 1, 2, 8, 16, 24, 30};
  for (int busy : busy_list) {
    printf ("kernel on stream %2d; a copy on stream k waits for it (ms until the copy is done):", busy);
    for (int k = 0; k < N; ++k) {
      if (k == busy) { printf ("  --"); continue; }
      hipLaunchKernelGGL (spin, dim3 (1), dim3 (64), 0, st[busy], ticks, d);
      const auto t0 = std::chrono::steady_clock::now();
      hipMemcpyAsync (d + 64, h, 256, hipMemcpyHostToDevice, st[k]);
      hipStreamSynchronize (st[k]);
      const double ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count();
      hipStreamSynchronize (st[busy]);
      printf (" %s%d", ms > 4.0 ? "*" : " ", k);
    }
    printf ("   (* = blocked)\n");
  }
  // the same for a kernel on `other` instead of a copy
  for (int busy : {0, 8}) {
    printf ("kernel on stream %2d; a KERNEL on stream k waits for it:", busy);
    for (int k = 0; k < N; ++k) {
      if (k == busy) { printf ("  --"); continue; }
      hipLaunchKernelGGL (spin, dim3 (1), dim3 (64), 0, st[busy], ticks, d);
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL (spin, dim3 (1), dim3 (64), 0, st[k], 100, d + 1);
      hipStreamSynchronize (st[k]);
      const double ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now() - t0).count();
      hipStreamSynchronize (st[busy]);
      printf (" %s%d", ms > 4.0 ? "*" : " ", k);
    }
    printf ("   (* = blocked)\n");
  }
  return 0;
}
