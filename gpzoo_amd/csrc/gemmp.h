// gemmp.h -- both forward products on one column panel held in LDS, for few inducing points (gemmp.hip).
#pragma once
#include "common.h"

namespace gpz {

// Wt = Linv * Kzx (optionally stored), colsum(Wt^2), muE^T Wt and colsum((LuT Wt)^2), fp32, Mp <= 512:
struct PanelArgs {
  const float* Linv; const float* LuT;        // (L, Mp, Mp): chol(Kzz)^{-1} (lower) and LuE^T (upper), identity / zero padded
  const float* Kzx;                           // (L, Mp, ncp), zero beyond M rows / the real columns -- or null with Z set:
  const float* Z; const float* X;             // generated operand (panel_generates): (M, d) inducing points, (nreal, d) spots
  const float* sigma; const float* ell;       //   (L,) kernel parameters
  int64_t M, nreal; int kind, d;              //   real rows / columns; GPZ_KERNEL_RBF / GPZ_KERNEL_MATERN32; d in {1, 2}
  float* Wt;                                  // out (L, Mp, ncp), or null: Wt never leaves the chip
  const float* muE;                           // (L, Mp), zero padded
  float* ps1; float* pm1; float* ps2;         // out [L][Mp/128][ncp], per 128-row block: colsum(Wt^2), muE^T Wt, colsum((LuT Wt)^2)
  int64_t Mp, ncp; int L;
};
bool panel_supported(int64_t Mp, int64_t ncp);
// True when the kernel can compute its covariance panel itself (fp32 RBF / Matern-3/2, d <= 2): Kzx is then never written.
bool panel_generates(int kind, int d);
int panel_launch(const PanelArgs& a, hipStream_t s);

}  // namespace gpz
