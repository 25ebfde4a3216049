// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 with register-resident operands.
// NACC independent accumulators per wave; DISTINCT = every MFMA of the unrolled body has its own A / B registers (as in
// a GEMM inner loop).  Reaches 77.6-77.9 TF = the 78.6 TF datasheet rate at the 2.39 GHz the chip holds (64 shader
// cycles per MFMA per SIMD) from one wave per SIMD upwards.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe_f64.hip -o /tmp/probe_f64 && /tmp/probe_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// In-kernel clock: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz; block 0 stamps both around its loop
// into a buffer nothing else reads (MI355X_MICROARCH.md, DVFS item 6).  A bare fp64 MFMA loop is power-limited: the
// chip holds a lower clock under it than under the GEMM, which stalls more.
__device__ unsigned long long g_stamps[4];

template <int NACC, bool DISTINCT>
__global__ __launch_bounds__(256) void probe(double* out, int iters) {
  d4 acc[NACC];
  double fa[NACC], fb[NACC];
  for (int a = 0; a < NACC; ++a) {
    acc[a] = d4{0, 0, 0, 0};
    fa[a] = 1.0 + threadIdx.x * 1e-3 + (DISTINCT ? a * 0.01 : 0.0);
    fb[a] = 0.5 + threadIdx.x * 1e-4 + (DISTINCT ? a * 0.02 : 0.0);
  }
  const bool stamp = blockIdx.x == 0 && threadIdx.x == 0;
  if (stamp) { g_stamps[0] = __builtin_amdgcn_s_memtime(); g_stamps[1] = __builtin_amdgcn_s_memrealtime(); }
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      // Inline asm with the accumulator tied to VGPRs: with the builtin hipcc keeps the accumulators in AGPRs and
      // copies all eight registers of every tile out and back in each iteration (16 v_accvgpr moves per MFMA), which
      // is what capped the round-2 version of this probe at 49 TF -- vector instructions are not free beside an MFMA.
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[a]) : "v"(fa[a]), "v"(fb[a]));
    }
  }
  if (stamp) { g_stamps[2] = __builtin_amdgcn_s_memtime(); g_stamps[3] = __builtin_amdgcn_s_memrealtime(); }
  double s = 0;
  for (int a = 0; a < NACC; ++a) for (int g = 0; g < 4; ++g) s += acc[a][g];
  out[blockIdx.x * 256 + threadIdx.x] = s;
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
  unsigned long long st[4];
  (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
  const double mhz = (double)(st[2] - st[0]) / (double)(st[3] - st[1]) * 100.0;     // shader cycles per 10 ns tick
  const double cyc = (double)(st[2] - st[0]) / ((double)bpc * iters * NACC);         // block 0's shader cycles per MFMA per SIMD
  printf("f64 16x16x4 nacc=%d %s blocks/CU=%d  %.3f ms  %.1f TF  in-kernel clock %.0f MHz, %.1f shader cycles/MFMA/SIMD -> %.1f TF at 2400 MHz\n",
         NACC, DISTINCT ? "distinct-operands" : "shared-operands", bpc, best, fl / best / 1e9, mhz, cyc,
         fl / best / 1e9 * 2400.0 / mhz);
}
int main() {
  double* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  run<8, true>(1, out); run<8, true>(2, out); run<8, true>(4, out); run<4, true>(8, out); run<16, true>(2, out);
  run<8, false>(4, out); run<16, false>(1, out);
  return 0;
}
