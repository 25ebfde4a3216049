// fused1.h -- stage 1 of the whitened solves with the covariance operand generated in registers (fused1.hip).
#pragma once
#include "common.h"

namespace gpz {

struct Fused1Args {
  const float* Linv; int64_t Mp;              // (L, Mp, Mp) fp32 copy of chol(Kzz)^{-1}, identity padded
  const float* Z; int64_t M;                  // (M, d) inducing inputs
  const float* X; int64_t nreal;              // this chunk's spots (nreal, d)
  int d, kind, L;                             // d in {1, 2}; kind GPZ_KERNEL_RBF / GPZ_KERNEL_MATERN32
  const float* sigma; const float* ell;       // (L,)
  float* Wt; int64_t ncp;                     // out (L, Mp, ncp): Wt = Linv * k(Z, X), zero beyond M rows / nreal columns
  const float* muE;                           // (L, Mp), zero padded
  float* ps_sq; float* ps_mu;                 // out [L][Mp/128][ncp]: per 128-row block, colsum(Wt^2) and muE^T Wt
};

// True when the fused kernel covers this problem (fp32, RBF / Matern-3/2, d <= 2).
bool fused1_supported(int dtype, int kind, int d);
int fused1_launch(const Fused1Args& a, hipStream_t s);

}  // namespace gpz
