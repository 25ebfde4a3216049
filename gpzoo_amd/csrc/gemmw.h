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

// Triangular A (L, Mp, Mp) times dense B (L, Mp, ncp) from memory, fp32, on 128 x 256 tiles:
enum WideEpilogue {
  WIDE_STORE_STATS = 0,     // lower A: C = A B stored, plus colsum(C^2) and mu^T C per 128-row block          (forward stage 1)
  WIDE_STATS = 1,           // upper A: colsum((A B)^2) per 128-row block, nothing stored                       (forward stage 2)
  WIDE_STORE = 2,           // upper A: C = A B                                                                  (backward Kbar_x)
  WIDE_STORE_COLSCALE = 3,  // upper A: C[i][j] = colscale[j] (A B)[i][j]                                        (backward Pbar)
  WIDE_WBAR = 4,            // lower A: C[i][j] = (A B)[i][j] + rowvec[i] colvec[j] - aux[i][j] colscale[j]      (backward Wbar)
  WIDE_KBAR = 5,            // DENSE A (upper = 2): C[i][j] = colscale[j] (A B)[i][j] + rowvec[i] colvec[j]      (backward Kbar_x in one product)
  WIDE_ADD_COLSCALE = 6,    // upper A: C[i][j] += colscale[j] (A B)[i][j]; `gate` (a device word): skipped when it is zero
};
struct WideArgs {
  const float* A; const float* B; int64_t Mp, ncp; int L;
  int upper, epilogue;                        // upper: 0 = lower-triangular A, 1 = upper-triangular, 2 = dense
  float* C; const float* mu;                  // mu: WIDE_STORE_STATS
  float* ps_sq; float* ps_mu;                 // [L][Mp/128][ncp]; ps_mu: WIDE_STORE_STATS
  const float* colscale; const float* colvec; // (L, ncp)
  const float* rowvec;                        // (L, Mp)
  const float* aux;                           // (L, Mp, ncp)
  const int32_t* gate;                        // WIDE_ADD_COLSCALE: device word, the launch does nothing when it is zero (or null)
};
bool wide_product_supported(int64_t Mp, int64_t ncp);
int wide_product_launch(const WideArgs& a, hipStream_t s);

// C (L, Mp, Mp) += A (L, Mp, K) * diag(w) * B (L, Mp, K)^T, lower 128-tiles only, fp32: the backward pass's gradient
// accumulation over an N-chunk (A == B: H += W diag(gv2) W^T).  w (L, K): column weights, or null.  The blocks on the
// diagonal are written whole, except for the symmetric product (A == B with weights): there only their 16 x 16
// sub-tiles on and below the diagonal are computed and written (36 of 64; the caller mirrors the triangle).  gate: device word -- when it is zero the launch leaves C alone (the rarely needed correction for
// columns at the variance clamp), or null.
bool wide_nt_supported(int64_t Mp, int64_t K);
// With few tiles (small M, few latents) the k extent is cut into pieces that run side by side and are added up in a fixed
// order; `scratch`: wide_nt_scratch_floats(Mp, K, L) floats (0: never cut), or null: one workgroup per tile walks all of k.
int wide_nt_pieces(int64_t Mp, int64_t K, int L);
size_t wide_nt_scratch_floats(int64_t Mp, int64_t K, int L);
// gm (L, K) with A == B and weights: the launch also forms v[l * v_stride + m] = sum_k A[l][m][k] gm[l][k] (fp64; the tiles
// on the diagonal do it from the fragments they hold) through vpart: wide_nt_vpart_floats(Mp, K, L) floats.
size_t wide_nt_vpart_floats(int64_t Mp, int64_t K, int L);
int wide_nt_launch(const float* A, const float* B, float* C, int64_t Mp, int64_t K, int L, hipStream_t s, float* scratch = nullptr,
                   const float* w = nullptr, const int32_t* gate = nullptr, const float* gm = nullptr, float* vpart = nullptr,
                   double* v = nullptr, int64_t v_stride = 0);

}  // namespace gpz
