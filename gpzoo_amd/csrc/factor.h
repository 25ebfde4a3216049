// factor.h -- batched Cholesky / triangular inverse of padded fp64 matrices (csrc/factor.hip, csrc/coop.hip).
#pragma once
#include "common.h"

namespace gpz {

// One-launch tile dataflow (csrc/coop.hip).  `sync`: coop_sync_words() words of device scratch, zeroed by the call.
bool coop_supported(int64_t Mp, bool inverse);
size_t coop_sync_words(int64_t Mp, int64_t batch);
int factor_coop(double* A, int64_t Mp, int64_t lda, int64_t stride, int64_t batch, int64_t m_real, double* Dinv,
                double* Linv, double* XT, uint32_t* sync, int32_t* info, hipStream_t s, float* Linv32 = nullptr,
                bool sync_cleared = false);

// Which path the entries below take: the one-launch dataflow unless GPZ_FACTOR_PATH=launches (the launch-per-step
// chain of rounds 1-3, kept for comparison and for orders the dataflow's task list cannot hold).
bool factor_use_coop(int64_t Mp, bool inverse);

// In-place Cholesky of `batch` padded (Mp,Mp) fp64 matrices; Dinv receives the inverse of every diagonal 128-block:
// (batch, Mp/128, 128, 128).  sync: coop_sync_words() words, or null (launch-per-step path).
int potrf_padded(double* A, int64_t Mp, int64_t lda, int64_t stride, int64_t batch, int64_t m_real, double* Dinv,
                 int32_t* info, hipStream_t s, bool clear_info = true, uint32_t* sync = nullptr);

// Linv = inverse of the factor in Lc, by recursive doubling over the 128-blocks.  T: batch * Mp * Mp / 2 doubles.
int trtri_padded(const double* Lc, int64_t ldl, int64_t stride_l, const double* Dinv, double* Linv, int64_t Mp,
                 int64_t batch, double* T, hipStream_t s);

// Both: A <- chol(A) in place (pitch Mp), Linv <- inverse of the factor (pitch Mp, zeros above the diagonal).
// T: batch * Mp * Mp doubles of scratch; sync as above (null: launch-per-step path).  Linv32 (nullable): an fp32 copy
// of Linv, written by the one-launch path itself; *wrote32 tells whether it was (the caller casts otherwise).
int factor_invert_padded(double* A, int64_t Mp, int64_t batch, int64_t m_real, double* Dinv, double* Linv, double* T,
                         uint32_t* sync, int32_t* info, hipStream_t s, float* Linv32 = nullptr, bool* wrote32 = nullptr,
                         bool sync_cleared = false);
// Words of `sync` a one-launch factorisation of this shape wants zeroed before it starts (0: the launch chain runs).  A caller
// that zeroes them together with its own flags passes sync_cleared = true and saves the launch.
size_t factor_sync_clear_words(int64_t Mp, int64_t batch, bool with_inverse);

}  // namespace gpz
