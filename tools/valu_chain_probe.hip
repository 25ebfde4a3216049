// Microbenchmark: what ONE wave pays per instruction in a chain of dependent fp64 vector operations on gfx950, alone
// and with independent operations in between -- the pivot sweep of the diagonal block (csrc/diag128.h factor32) is such
// a chain, 128 columns long, and sits on the critical path of every block column of the factorisation.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_chain_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

constexpr int REP = 64, ITER = 32;

#define CHAIN_BODY(...)                                             \
  for (int it = 0; it < ITER; ++it) {                                  \
    _Pragma("unroll") for (int u = 0; u < REP; ++u) { __VA_ARGS__ } \
  }

__global__ __launch_bounds__(64) void probe(double* out, unsigned long long* cyc) {
  __shared__ double lds[256];
  double x = 1.0 + threadIdx.x * 1e-9, c = 0.999999, d = 1e-7;
  double y0 = 1.0, y1 = 1.1, y2 = 1.2, y3 = 1.3;
  lds[threadIdx.x] = x;
  __syncthreads();
  unsigned long long t0, t1;
  int k = 0;
  // 0: dependent v_fma_f64
  t0 = now();
  CHAIN_BODY(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 1: dependent + 1 independent
  t0 = now();
  CHAIN_BODY(asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(x), "+v"(y0) : "v"(c), "v"(d));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 2: dependent + 2 independent
  t0 = now();
  CHAIN_BODY(asm volatile("v_fma_f64 %0, %0, %3, %4\n\tv_fma_f64 %1, %1, %3, %4\n\tv_fma_f64 %2, %2, %3, %4" : "+v"(x), "+v"(y0), "+v"(y1) : "v"(c), "v"(d));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 3: dependent + 3 independent
  t0 = now();
  CHAIN_BODY(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(x), "+v"(y0), "+v"(y1), "+v"(y2) : "v"(c), "v"(d));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 4: 4 independent chains only (issue rate)
  t0 = now();
  CHAIN_BODY(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(y3), "+v"(y0), "+v"(y1), "+v"(y2) : "v"(c), "v"(d));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = (t1 - t0) / 4; ++k;
  // 5: dependent v_mul_f64
  t0 = now();
  CHAIN_BODY(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 6: dependent v_rsq_f64
  x = 1.0 + threadIdx.x * 1e-3;
  t0 = now();
  CHAIN_BODY(asm volatile("v_rsq_f64 %0, %0" : "+v"(x));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 7: v_rsq_f64 followed by a dependent fma (pair)
  t0 = now();
  CHAIN_BODY(asm volatile("v_rsq_f64 %0, %0\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 8: readlane (2 words) -> fma with the SGPR pair (pair)
  t0 = now();
  CHAIN_BODY(asm volatile("v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %1, 5\n\tv_fma_f64 %2, s[20:21], %3, %2"
                          : : "v"(__double2loint(x)), "v"(__double2hiint(x)), "v"(x), "v"(c) : "s20", "s21");)
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 9: dependent through readlane: x -> sgpr -> fma -> x
  x = 1.0;
  t0 = now();
  CHAIN_BODY({
    int lo = __double2loint(x), hi = __double2hiint(x);
    asm volatile("v_readlane_b32 s20, %1, 5\n\tv_readlane_b32 s21, %2, 5\n\ts_nop 0\n\tv_fma_f64 %0, s[20:21], %3, %4"
                 : "=v"(x) : "v"(lo), "v"(hi), "v"(c), "v"(d) : "s20", "s21");
  })
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 10: dependent through permlane32_swap (2 words) + fma
  t0 = now();
  CHAIN_BODY({
    int lo = __double2loint(x), hi = __double2hiint(x);
    auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    double s = __hiloint2double((int)h2[1], (int)l2[1]);
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "v"(s), "v"(c), "v"(d));
  })
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 11: LDS write -> read round trip, dependent
  t0 = now();
  CHAIN_BODY({
    lds[threadIdx.x] = x;
    asm volatile("" ::: "memory");
    x = lds[threadIdx.x ^ 1] * c;
    asm volatile("" : "+v"(x));
  })
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 12: v_cndmask pair + fma (select on the chain)
  t0 = now();
  CHAIN_BODY({
    int lo = __double2loint(x), hi = __double2hiint(x);
    asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(7), "v"(9) : );
    x = __hiloint2double(hi, lo);
    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
  })
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  // 13: scalar detour: readlane -> v_cmp_gt_f64 (sgpr) -> s_cselect x2 -> v_mov_b64 -> fma   (the d > 0 guard of the sweep)
  x = 1.0;
  t0 = now();
  CHAIN_BODY({
    int lo = __double2loint(x), hi = __double2hiint(x);
    asm volatile("v_readlane_b32 s20, %1, 5\n\tv_readlane_b32 s21, %2, 5\n\ts_nop 1\n\tv_cmp_gt_f64 s[22:23], s[20:21], 0\n\t"
                 "s_and_b64 s[22:23], s[22:23], exec\n\ts_cselect_b32 s21, s21, 0x3ff00000\n\ts_cselect_b32 s20, s20, 0\n\t"
                 "v_mov_b64 %0, s[20:21]\n\tv_fma_f64 %0, %0, %3, %4"
                 : "=&v"(x) : "v"(lo), "v"(hi), "v"(c), "v"(d) : "s20", "s21", "s22", "s23", "scc");
  })
  t1 = now(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
  out[threadIdx.x] = x + y0 + y1 + y2 + y3;
}

int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 64 * sizeof(double));
  (void)hipMalloc(&cyc, 32 * sizeof(unsigned long long));
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc); (void)hipDeviceSynchronize(); }
  unsigned long long h[32];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"dependent v_fma_f64", "dependent fma + 1 independent fma", "dependent fma + 2 independent", "dependent fma + 3 independent",
                         "independent v_fma_f64 (issue)", "dependent v_mul_f64", "dependent v_rsq_f64", "v_rsq_f64 + dependent fma",
                         "2 readlane + fma (not dependent)", "fma -> 2 readlane -> fma (dependent)", "fma -> 2 permlane32_swap -> fma (dependent)",
                         "LDS write -> read -> mul (dependent)", "2 v_cndmask + fma (dependent)", "readlane -> v_cmp -> s_cselect -> v_mov -> fma (dependent)"};
  for (int k = 0; k < 14; ++k) printf("%-60s %7.1f shader cycles per step\n", names[k], (double)h[k] / (REP * ITER));
  return 0;
}
