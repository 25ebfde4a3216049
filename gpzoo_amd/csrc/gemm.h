// gemm.h -- batched, triangular-aware MFMA GEMM used by every dense step of the
// path (trailing SYRK of the Cholesky, panel solves, triangular inverse, the
// L^{-1} Kzx and Lu^T Wt products).  All dimensions are multiples of 128.
#pragma once
#include "common.h"

namespace gpz {

enum GemmFlags : int {
  GF_A_LOWER = 1,      // A[i][k] == 0 for k-block > i-block   -> k_end   = (i+1)*128
  GF_A_UPPER = 2,      // A[i][k] == 0 for k-block < i-block   -> k_begin = i*128
  GF_B_LOWER = 4,      // op(B)[k][j] == 0 for k-block < j-block -> k_begin = j*128
  GF_B_UPPER = 8,      // op(B)[k][j] == 0 for k-block > j-block -> k_end   = (j+1)*128
  GF_TILES_LOWER = 16, // only output tiles with i >= j (SYRK / lower x lower)
  GF_B_TRANS = 32,     // B is stored (N,K) row-major: C = A * B^T
  GF_GROUP_COLS = 64,  // schedule: blocks that share a B column panel run together on one XCD
};

enum GemmEpilogue : int {
  EPI_STORE = 0,       // C = alpha * A op(B) + beta * C
  EPI_STORE_STATS = 1, // store, plus per-column sum(v^2) and sum(mu[row] * v) over the tile's rows
  EPI_STATS = 2,       // per-column sum(v^2) only, nothing stored
  EPI_STORE_COLSCALE = 3,  // C[i][j] = alpha * colscale[j] * (A op(B))[i][j]   (NN only; backward's P-bar)
  EPI_WBAR = 4,            // C[i][j] = alpha * acc + rowvec[i] * colvec[j] - aux[i][j] * colscale[j]  (backward's W-bar)
};

template <typename T>
struct GemmParams {
  const T* A = nullptr; const T* B = nullptr; T* C = nullptr;
  int64_t lda = 0, ldb = 0, ldc = 0;
  int64_t sA0 = 0, sA1 = 0, sB0 = 0, sB1 = 0, sC0 = 0, sC1 = 0;  // batch strides (outer, inner), elements
  int nb0 = 1, nb1 = 1;     // outer x inner batch counts
  int mt = 0, nt = 0;       // output tiles of 128 x 128
  int K = 0;                // contraction extent (multiple of 128 when a triangular flag is set)
  int flags = 0;
  int super_cols = 8;       // GF_GROUP_COLS: column tiles per L2 super-tile
  int tiles_per_wg = 1;     // GF_GROUP_COLS: column tiles of one row tile a workgroup computes back to back
  int xcd_contiguous = 1;   // without GF_GROUP_COLS: each XCD takes a contiguous range of the tile order (set by gemm_launch)
  T alpha = 1, beta = 0;
  // column statistics (EPI_STORE_STATS / EPI_STATS): partial sums per (outer batch, row tile, column)
  const T* mu = nullptr; int64_t sMu = 0;  // (outer batch, K) vector, padded with zeros
  T* ps_sq = nullptr;       // [nb0][mt][ncols]
  T* ps_mu = nullptr;       // [nb0][mt][ncols]
  int64_t ncols = 0;        // nt * 128
  const T* colscale = nullptr; int64_t sCs = 0;  // EPI_STORE_COLSCALE / EPI_WBAR: (outer batch, ncols) factors
  const T* colvec = nullptr;                     // EPI_WBAR: (outer batch, ncols), stride sCs
  const T* rowvec = nullptr; int64_t sRv = 0;    // EPI_WBAR: (outer batch, rows)
  const T* aux = nullptr;                        // EPI_WBAR: matrix laid out like C (same ldc / strides)
};

template <typename T>
int gemm_launch(const GemmParams<T>& p, int epilogue, hipStream_t s);

}  // namespace gpz
