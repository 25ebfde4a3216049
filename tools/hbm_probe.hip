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

// Write-only variants, to find what the write path can sustain (the Kzx fill is measured against it):
//   chunk_k   every workgroup owns contiguous CHUNK-byte pieces and walks each front to back (1 KB per wave store,
//             4 KB per workgroup step): long sequential runs per workgroup instead of a grid-strided interleave
//   streams_k the fill's own shape: a workgroup step writes one 4 KB run into each of NS far-apart streams (the L
//             latent matrices), then moves on by 4 KB in all of them
__global__ __launch_bounds__(256) void chunk_k(float4* __restrict__ dst, size_t n, size_t chunk_vec) {
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
  const size_t nchunks = n / chunk_vec;
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    float4* p = dst + c * chunk_vec;
    for (size_t i = threadIdx.x; i < chunk_vec; i += 256) p[i] = v;
  }
}
__global__ __launch_bounds__(256) void streams_k(float4* __restrict__ dst, size_t n, int ns, size_t run_vec) {
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
  const size_t per = n / ns;                     // float4s per stream
  const size_t steps = per / 256;                // 4 KB workgroup steps per stream
  for (size_t st = blockIdx.x * run_vec; st < steps; st += (size_t)gridDim.x * run_vec)
    for (size_t k = 0; k < run_vec && st + k < steps; ++k)
      for (int s = 0; s < ns; ++s) dst[(size_t)s * per + (st + k) * 256 + threadIdx.x] = v;
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
  for (size_t chunk : {(size_t)16 << 10, (size_t)64 << 10, (size_t)1 << 20}) {
    const double w = best_ms([&] { hipLaunchKernelGGL(chunk_k, dim3(256 * 8), dim3(256), 0, 0, a, n, chunk / 16); });
    printf("hbm write, contiguous %zu KB pieces per workgroup: %.0f GB/s\n", chunk >> 10, bytes / w / 1e6);
  }
  for (int ns : {1, 8, 32}) {
    for (size_t run : {(size_t)1, (size_t)8}) {
      const double w = best_ms([&] { hipLaunchKernelGGL(streams_k, dim3(256 * 8), dim3(256), 0, 0, a, n, ns, run); });
      printf("hbm write, %d streams, %zu x 4 KB per stream and workgroup turn: %.0f GB/s\n", ns, run, bytes / w / 1e6);
    }
  }
  {
    const double w = best_ms([&] { hipMemsetAsync(a, 0, bytes, 0); });
    printf("hbm hipMemsetAsync: %.0f GB/s\n", bytes / w / 1e6);
  }
  return 0;
}
