// gemmp.h -- both forward products on one column panel held in LDS, for few inducing points (gemmp.hip).
#pragma once
#include "common.h"

namespace gpz {

// Wt = Linv * Kzx (optionally stored), colsum(Wt^2), muE^T Wt and colsum((LuT Wt)^2), fp32, Mp <= 512:
struct PanelArgs {
  const float* Linv; const float* LuT;        // (L, Mp, Mp): chol(Kzz)^{-1} (lower) and LuE^T (upper), identity / zero padded
  const float* Kzx;                           // (L, Mp, ncp), zero beyond M rows / the real columns
  float* Wt;                                  // out (L, Mp, ncp), or null: Wt never leaves the chip
  const float* muE;                           // (L, Mp), zero padded
  float* ps1; float* pm1; float* ps2;         // out [L][Mp/128][ncp], per 128-row block: colsum(Wt^2), muE^T Wt, colsum((LuT Wt)^2)
  int64_t Mp, ncp; int L;
};
bool panel_supported(int64_t Mp, int64_t ncp);
int panel_launch(const PanelArgs& a, hipStream_t s);

}  // namespace gpz
