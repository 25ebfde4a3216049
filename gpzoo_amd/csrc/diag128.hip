// diag128.hip -- Cholesky factor AND inverse of a 128x128 SPD diagonal block, one workgroup per
// matrix, fp64, everything resident in LDS (the "LDS staging of the diagonal panel" of the
// blocked right-looking factorisation, reference call sites gp.py:213/270/360).  The phases live in
// csrc/diag128.h (shared with the one-launch cooperative factorisation, csrc/coop.hip); this file is the
// stand-alone kernel of the launch-per-step path and of the substitution solve (inverse-only mode).
#include "common.h"

namespace gpz {

// Diagnostic build only (-DGPZ_DIAG_STAMPS, tools/diag_stamps.sh): cycle stamps of workgroup 0 at the phase boundaries.
#ifdef GPZ_DIAG_STAMPS
__device__ unsigned long long g_diag_stamps[64];
#define GPZ_STAMP(i)                                                                  \
  do {                                                                                \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                     \
      unsigned long long t_;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      g_diag_stamps[i] = t_;                                                          \
    }                                                                                 \
  } while (0)
#endif

}  // namespace gpz

#include "diag128.h"

namespace gpz {

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void diag128_kernel(double* __restrict__ A, int64_t lda, int64_t stride, int bk,
                                                     double* __restrict__ Dinv, int64_t dinv_stride,
                                                     int32_t* __restrict__ info, int64_t m_real, int factor) {
  extern __shared__ __attribute__((aligned(16))) double S[];  // [128][129] + column buffer [32][32] + 1/diag [32]
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  if (bk < 0) bk = blockIdx.y;   // all diagonal blocks in one launch (inverse-only mode)
  double* Ab = A + (int64_t)b * stride + (int64_t)bk * 128 * (lda + 1);
  GPZ_STAMP(0);
  diag::load_block<256>(S, Ab, lda, tid);
  diag::prepare(S, tid);
  __syncthreads();
  GPZ_STAMP(1);
  diag::factor_block(S, tid, info + b, (int64_t)bk * 128, m_real, factor);
  GPZ_STAMP(18);
  if (factor) diag::store_factor<256>(S, Ab, lda, tid);
  GPZ_STAMP(19);
  diag::inverse_levels(S, tid);
  diag::store_inverse<256>(S, Dinv + (int64_t)b * dinv_stride + (int64_t)bk * 128 * 128, 128, tid);
  GPZ_STAMP(22);
}

#ifdef GPZ_DIAG_STAMPS
extern "C" int gpz_debug_diag_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_diag_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace gpz
