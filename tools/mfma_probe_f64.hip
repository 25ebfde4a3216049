#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(double* out, int iters) {
  d4 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = d4{0, 0, 0, 0};
  double fa = 1.0 + threadIdx.x * 1e-3, fb = 0.5 + threadIdx.x * 1e-4;
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[a], 0, 0, 0);
  }
  double s = 0;
  for (int a = 0; a < NACC; ++a) for (int g = 0; g < 4; ++g) s += acc[a][g];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int bpc, double* out) {
  const int iters = 20000, grid = 256 * bpc;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NACC>), dim3(grid), dim3(256), 0, 0, out, 10); hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((probe<NACC>), dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double fl = (double)grid * 4 * iters * NACC * 16.0 * 16 * 4 * 2;
  printf("f64 16x16x4 nacc=%d blocks/CU=%d  %.3f ms  %.1f TF  (%.1f cycles/MFMA/SIMD at 2.4GHz)\n", NACC, bpc, best, fl / best / 1e9,
         best * 1e-3 * 2.4e9 / ((double)bpc * iters * NACC));
}
int main() {
  double* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  run<16>(1, out); run<16>(2, out); run<8>(4, out); run<4>(8, out); run<8>(8, out); run<2>(8, out); run<16>(4,out);
  return 0;
}
