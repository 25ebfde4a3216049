#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// each thread writes U consecutive 16-byte vectors; blocks own contiguous runs
template <int U, int NT>
__global__ __launch_bounds__(1024) void w_thread_run(f4* __restrict__ dst, size_t n) {
  const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const size_t per_block = (size_t)blockDim.x * U;
  for (size_t base = (size_t)blockIdx.x * per_block; base < n; base += (size_t)gridDim.x * per_block) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * blockDim.x + threadIdx.x;     // coalesced per instruction
      if (i < n) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
    }
  }
}
// one pass, no loop: grid covers everything, each thread U vectors
template <int U>
__global__ __launch_bounds__(256) void w_flat(f4* __restrict__ dst, size_t n) {
  const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const size_t base = (size_t)blockIdx.x * 256 * U;
#pragma unroll
  for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256 + threadIdx.x; if (i < n) dst[i] = v; }
}
template <typename F> static double best_ms(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;
  f4* a; hipMalloc(&a, bytes); hipMemset(a, 0, bytes);
  for (int threads : {256, 512, 1024})
    for (int grid : {256, 512, 1024, 2048, 8192}) {
      double w1 = best_ms([&] { hipLaunchKernelGGL((w_thread_run<4, 0>), dim3(grid), dim3(threads), 0, 0, a, n); });
      double w2 = best_ms([&] { hipLaunchKernelGGL((w_thread_run<16, 0>), dim3(grid), dim3(threads), 0, 0, a, n); });
      double w3 = best_ms([&] { hipLaunchKernelGGL((w_thread_run<4, 1>), dim3(grid), dim3(threads), 0, 0, a, n); });
      printf("threads %4d grid %5d: U=4 %.0f  U=16 %.0f  U=4 nt %.0f GB/s\n", threads, grid, bytes / w1 / 1e6, bytes / w2 / 1e6, bytes / w3 / 1e6);
    }
  for (int U : {1, 4}) {
    const unsigned grid = (unsigned)((n + 256ull * U - 1) / (256ull * U));
    double w = U == 1 ? best_ms([&] { hipLaunchKernelGGL((w_flat<1>), dim3(grid), dim3(256), 0, 0, a, n); })
                      : best_ms([&] { hipLaunchKernelGGL((w_flat<4>), dim3(grid), dim3(256), 0, 0, a, n); });
    printf("flat grid, U=%d: %.0f GB/s\n", U, bytes / w / 1e6);
  }
  double m = best_ms([&] { hipMemsetAsync(a, 0, bytes, 0); });
  printf("memset %.0f GB/s\n", bytes / m / 1e6);
  return 0;
}
