// Calibration of the FETCH_SIZE counter for the access pattern of the wide kernel's A tile: 64-byte row segments (four
// lanes x 16 bytes) at a row stride of 8 KB, the two halves of a 128-byte line read in consecutive steps.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_probe.hip -o /tmp/fetch_probe
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_probe        (then tools/fetch_probe_read.py out)
// Three kernels over the same 4 GiB buffer (far beyond the 256 MiB Infinity Cache), each launched once per run:
//   stream_k    every byte, 16 bytes per lane, lanes contiguous (the pattern the guide's "double FETCH_SIZE" note is about)
//   segments_k  every byte, as 64-byte row segments at 8 KB stride (the A-tile pattern): useful bytes = 4 GiB
//   halves_k    only the first 64 bytes of every 128-byte line, same segment pattern: useful bytes = 2 GiB
// If L2 -> fabric requests are whole 128-byte lines tallied as 64 bytes, all three report 2 GiB; if a 64-byte request
// fetches (and tallies) 64 bytes, segments_k reports 4 GiB and halves_k 2 GiB while stream_k reports 2 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr size_t ROW = 8192;          // bytes per row

__global__ __launch_bounds__(256) void stream_k(const float4* __restrict__ src, size_t n, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}

// a wave owns 16 rows; lane -> (row = lane / 4, 16-byte chunk = lane % 4) of the 64-byte segment at byte offset k
template <int STEP>
__global__ __launch_bounds__(256) void segments_k(const char* __restrict__ src, size_t rows, float* out) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t row = wave * 16 + (lane >> 2);
  if (row >= rows) return;
  const char* p = src + row * ROW + (lane & 3) * 16;
  float s = 0.f;
#pragma unroll 4
  for (size_t k = 0; k < ROW; k += STEP) { const float4 v = *reinterpret_cast<const float4*>(p + k); s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
static double once_ms(F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const size_t bytes = (size_t)4 << 30, rows = bytes / ROW;
  char* a; float* o;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&o, 4);
  (void)hipMemset(a, 0, bytes);
  (void)hipDeviceSynchronize();
  const unsigned grid = (unsigned)(rows / 16 / 4);      // 4 waves per workgroup
  for (int rep = 0; rep < 2; ++rep) {
    const double t0 = once_ms([&] { hipLaunchKernelGGL(stream_k, dim3(256 * 16), dim3(256), 0, 0, (const float4*)a, bytes / 16, o); });
    const double t1 = once_ms([&] { hipLaunchKernelGGL(segments_k<64>, dim3(grid), dim3(256), 0, 0, a, rows, o); });
    const double t2 = once_ms([&] { hipLaunchKernelGGL(segments_k<128>, dim3(grid), dim3(256), 0, 0, a, rows, o); });
    printf("stream %.3f ms (%.0f GB/s)  segments %.3f ms (%.0f GB/s useful)  halves %.3f ms (%.0f GB/s useful)\n", t0,
           bytes / t0 / 1e6, t1, bytes / t1 / 1e6, t2, bytes / 2 / t2 / 1e6);
  }
  return 0;
}
