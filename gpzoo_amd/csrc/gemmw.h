// gemmw.h -- the two big fp32 products of the forward pass on wide workgroup tiles (gemmw.hip).
#pragma once
#include "common.h"

namespace gpz {

// Stage 1 with the covariance operand generated inside the product: Wt = Linv * k(Z, X) (+ per 128-row block
// colsum(Wt^2) and muE^T Wt).  Kzx is never written.
struct Fused1Args {
  const float* Linv; int64_t Mp;              // (L, Mp, Mp) fp32 copy of chol(Kzz)^{-1}, identity padded
  const float* Z; int64_t M;                  // (M, d) inducing inputs
  const float* X; int64_t nreal;              // this chunk's spots (nreal, d)
  int d, kind, L;                             // d in {1, 2}; kind GPZ_KERNEL_RBF / GPZ_KERNEL_MATERN32
  const float* sigma; const float* ell;       // (L,)
  float* Wt; int64_t ncp;                     // out (L, Mp, ncp): zero beyond M rows / nreal columns
  const float* muE;                           // (L, Mp), zero padded
  float* ps_sq; float* ps_mu;                 // out [L][Mp/128][ncp]
};
// True when the generator covers this problem (fp32, RBF / Matern-3/2, d <= 2).
bool fused1_supported(int dtype, int kind, int d);
int fused1_launch(const Fused1Args& a, hipStream_t s);

// Triangular A (L, Mp, Mp) times dense B (L, Mp, ncp) from memory, fp32:
//   upper = 0: C = A * B with A lower triangular, stored, plus colsum(C^2) and mu^T C per 128-row block   (stage 1)
//   upper = 1: colsum((A * B)^2) per 128-row block with A upper triangular, nothing stored                 (stage 2)
struct WideArgs {
  const float* A; const float* B; int64_t Mp, ncp; int L;
  int upper, store;
  float* C; const float* mu;                  // stage 1 only
  float* ps_sq; float* ps_mu;                 // [L][Mp/128][ncp]; ps_mu stage 1 only
};
bool wide_product_supported(int64_t Mp, int64_t ncp);
int wide_product_launch(const WideArgs& a, hipStream_t s);

}  // namespace gpz
