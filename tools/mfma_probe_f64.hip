// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 with register-resident operands.
// NACC independent accumulators per wave; DISTINCT = every MFMA of the unrolled body has its own A / B registers (as in
// a GEMM inner loop); with one shared A / B pair the same loop issues ~35 % slower on gfx950 (measured: 49 vs 7x TF).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe_f64.hip -o /tmp/probe_f64 && /tmp/probe_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, bool DISTINCT>
__global__ __launch_bounds__(256) void probe(double* out, int iters) {
  d4 acc[NACC];
  double fa[NACC], fb[NACC];
  for (int a = 0; a < NACC; ++a) {
    acc[a] = d4{0, 0, 0, 0};
    fa[a] = 1.0 + threadIdx.x * 1e-3 + (DISTINCT ? a * 0.01 : 0.0);
    fb[a] = 0.5 + threadIdx.x * 1e-4 + (DISTINCT ? a * 0.02 : 0.0);
  }
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      if (DISTINCT) asm volatile("" : "+v"(fa[a]), "+v"(fb[a]));      // keep the operand registers apart
      acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[a], acc[a], 0, 0, 0);
    }
  }
  double s = 0;
  for (int a = 0; a < NACC; ++a) for (int g = 0; g < 4; ++g) s += acc[a][g];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// GEMM-shaped register use: a 4 x 2 grid of accumulators fed by 4 A and 2 B operands, like one k-step of the 64 x 32
// wave tile of csrc/gemm.hip (operands change every iteration through an opaque register move).
__global__ __launch_bounds__(256) void probe_tile(double* out, int iters) {
  d4 acc[4][2];
  double fa[4], fb[2];
  for (int m = 0; m < 4; ++m) { fa[m] = 1.0 + threadIdx.x * 1e-3 + m * 0.01; for (int n = 0; n < 2; ++n) acc[m][n] = d4{0, 0, 0, 0}; }
  for (int n = 0; n < 2; ++n) fb[n] = 0.5 + threadIdx.x * 1e-4 + n * 0.02;
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(fa[m]));
#pragma unroll
    for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(fb[n]));
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], fb[n], acc[m][n], 0, 0, 0);
  }
  double s = 0;
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int g = 0; g < 4; ++g) s += acc[m][n][g];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
void run_tile(int bpc, double* out) {
  const int iters = 20000, grid = 256 * bpc;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe_tile, dim3(grid), dim3(256), 0, 0, out, 10); (void)hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL(probe_tile, dim3(grid), dim3(256), 0, 0, out, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double fl = (double)grid * 4 * iters * 8 * 16.0 * 16 * 4 * 2;
  printf("f64 16x16x4 tile4x2 distinct-operands blocks/CU=%d  %.3f ms  %.1f TF  (%.1f cycles/MFMA/SIMD at 2.4GHz)\n", bpc, best,
         fl / best / 1e9, best * 1e-3 * 2.4e9 / ((double)bpc * iters * 8));
}

template <int NACC, bool DISTINCT>
void run(int bpc, double* out) {
  const int iters = 20000, grid = 256 * bpc;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NACC, DISTINCT>), dim3(grid), dim3(256), 0, 0, out, 10); (void)hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL((probe<NACC, DISTINCT>), dim3(grid), dim3(256), 0, 0, out, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double fl = (double)grid * 4 * iters * NACC * 16.0 * 16 * 4 * 2;
  printf("f64 16x16x4 nacc=%d %s blocks/CU=%d  %.3f ms  %.1f TF  (%.1f cycles/MFMA/SIMD at 2.4GHz)\n", NACC,
         DISTINCT ? "distinct-operands" : "shared-operands", bpc, best, fl / best / 1e9,
         best * 1e-3 * 2.4e9 / ((double)bpc * iters * NACC));
}
int main() {
  double* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  run<8, true>(1, out); run<8, true>(2, out); run<8, true>(4, out); run<4, true>(8, out); run<16, true>(2, out);
  run<8, false>(4, out); run<16, false>(1, out);
  run_tile(1, out); run_tile(2, out); run_tile(4, out);
  return 0;
}
