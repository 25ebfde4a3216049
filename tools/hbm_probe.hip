// Microbenchmark: achievable HBM streaming rates on this chip (the denominators BASELINE.md §3 asks to measure
// rather than take from a datasheet): write-only (what the Kzx fill does), read-only and copy, 16 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void write_k(float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}
__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ src, size_t n, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

template <typename F>
static double best_ms(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / sizeof(float4);   // 4 GiB per buffer: far beyond the 256 MiB Infinity Cache
  float4 *a, *b; float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  for (int bpc : {4, 8, 16, 32}) {
    const int grid = 256 * bpc;
    const double w = best_ms([&] { hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, a, n); });
    const double r = best_ms([&] { hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, n, o); });
    const double c = best_ms([&] { hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n); });
    printf("hbm blocks/CU=%d  write %.0f GB/s  read %.0f GB/s  copy %.0f GB/s (read+write bytes)\n", bpc, bytes / w / 1e6,
           bytes / r / 1e6, 2.0 * bytes / c / 1e6);
  }
  return 0;
}
