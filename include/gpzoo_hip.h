/* gpzoo_hip.h -- C ABI of libgpzoo_hip.so (MI355X / gfx950).
 *
 * The reference (luisdiaz1997/GPzoo) is pure Python/torch and has no FFI of its
 * own; this ABI is what a maintainer binds (ctypes stub in INTEGRATION.md) to
 * replace the torch ops on the SVGP/NSF hot path.  Each entry point cites the
 * reference lines whose arithmetic it replaces (paths under /root/reference).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the
 *    caller unless marked "host"; the library never allocates user-visible
 *    memory; scratch comes from a caller workspace sized by *_workspace_bytes.
 *  - all launches are asynchronous on `stream` (a hipStream_t passed as void*).
 *  - return 0 on success, negative on bad arguments / launch errors
 *    (gpz_last_error() holds the message, per thread).
 *  - dtype: GPZ_F32 = 0, GPZ_F64 = 1.  Matrices are row-major, batched over a
 *    leading latent axis L with an explicit batch stride (in elements).
 *  - non-PD input to a factorisation: LAPACK-style info[b] = k > 0 written to
 *    device memory (order of the first non-positive leading minor); the Python
 *    wrapper turns it into torch.linalg.LinAlgError like gp.py:213/270/360.
 */
#ifndef GPZOO_HIP_H
#define GPZOO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPZ_VERSION 212

enum { GPZ_F32 = 0, GPZ_F64 = 1 };

/* covariance families: every kernel class of gpzoo/kernels.py maps to one */
enum {
  GPZ_KERNEL_RBF = 0,      /* RBF, NSF_RBF, batched_RBF      kernels.py:34-59,106-155  */
  GPZ_KERNEL_MATERN32 = 1, /* batched_Matern32               kernels.py:6-30           */
  GPZ_KERNEL_MGGP_RBF = 2, /* MGGP_RBF, MGGP_NSF_RBF, batched_MGGP_RBF  kernels.py:62-104,158-228 */
  GPZ_KERNEL_DISTANCE = 3  /* plain Euclidean distance (return_distance=True, kernels.py:118,125-126) */
};

/* Hyper-parameters of one covariance family for L independent latents.
 * sigma / lengthscale / group_a: device arrays of L values of dtype `dtype`.
 * group_a is the EFFECTIVE multiplier of the squared group distance
 * (a, a^2 or |a| depending on the class: kernels.py:187 / :222 / :87), NULL
 * unless kind == GPZ_KERNEL_MGGP_RBF.  group_r2 is the (G,G) table of squared
 * distances between group embeddings (dtype `dtype`); group_pow = p/2, the
 * exponent of the denominator (kernels.py:189, :224, :89). */
typedef struct gpz_kernel_desc {
  int32_t kind;
  int32_t n_latent;
  int32_t dtype;
  int32_t n_groups;
  const void* sigma;
  const void* lengthscale;
  const void* group_a;
  const void* group_r2;
  double group_pow;
} gpz_kernel_desc;

int gpz_version(void);
const char* gpz_last_error(void);
/* sha256 (32 hex digits) of the source files this binary was built from ("unknown" for a hand-made build): the Python
 * loader refuses a library whose value differs from the sources lying next to it (a stale binary). */
const char* gpz_source_hash(void);

/* K[l][i][j] = k_l(A_i, B_j) (+ jitter where i == j if jitter != 0).
 * Replaces kernel.forward(X, Z) -- kernels.py:29-30, 57-58, 98-104, 118-130,
 * 146-155, 176-191, 211-228 -- and add_jitter (utilities.py:407-418) fused.
 * A (nA,d), B (nB,d) of k->dtype; gA/gB int64 group ids (MGGP only, else NULL).  An id outside
 * [0, n_groups) -- IndexError in the reference (kernels.py:99-100, 177-178, 209-210) -- is read as group 0
 * here (no out-of-bounds access); gpz_svgp_forward reports it through info (see there).
 * K has dtype out_dtype, row stride ldk, latent stride stride_k (elements).
 * Distances are evaluated by direct differencing in the output precision. */
int gpz_kfill(const gpz_kernel_desc* k, const void* A, int64_t nA, const void* B, int64_t nB,
              int32_t d, const int64_t* gA, const int64_t* gB, void* K, int64_t ldk,
              int64_t stride_k, double jitter, int32_t out_dtype, void* stream);

/* Backward of gpz_kfill: contracts Kbar = dLoss/dK (L,nA,nB; dtype k->dtype, row stride ldk, latent stride
 * stride_k) with dK/d(sigma, lengthscale, group_a) and dK/dA -- what torch autograd sends through
 * kernel.forward in the reference, where kernels are ordinary traced modules (kernels.py:14-30, 42-58,
 * 75-104, 114-130, 139-155, 176-228; e.g. the inline ExactGP of exact_mggp.ipynb).  Outputs fp64:
 * grad_theta (L,4) = d/dsigma, d/dlengthscale, d/dgroup_a (effective multiplier), 0;  grad_A (nA,4), first d
 * columns used, summed over latents (NULL: skipped).  For dLoss/dB call it again with A and B (and the groups)
 * swapped and Kbar transposed.  Matern-3/2: dK/dA at coincident points is 0 (its limit; the reference's
 * autograd yields NaN there).  GPZ_KERNEL_DISTANCE is not differentiable here. */
size_t gpz_kgrad_workspace_bytes(int64_t nA, int32_t n_latent);
int gpz_kgrad(const gpz_kernel_desc* k, const void* A, int64_t nA, const void* B, int64_t nB, int32_t d,
              const int64_t* gA, const int64_t* gB, const void* Kbar, int64_t ldk, int64_t stride_k,
              double* grad_theta, double* grad_A, void* ws, size_t ws_bytes, void* stream);

/* In-place lower Cholesky of `batch` (M,M) matrices stored as `dtype` (the arithmetic is fp64 either way:
 * "factor precision"), zeros written above the diagonal; info[b] as described above.  Replaces
 * torch.linalg.cholesky at gp.py:213, 270, 360.  Default: ONE launch, a left-looking dataflow over 128 x 128 tiles
 * (csrc/coop.hip: diagonal blocks factored in LDS, every tile's sum kept in its owner's registers on
 * v_mfma_f64_16x16x4_f64, hand-offs through flags).  Orders beyond that launch's task list, or
 * GPZ_FACTOR_PATH=launches, take the blocked right-looking chain of launches (csrc/factor.hip: LDS-resident diagonal
 * panel, MFMA panel solve and trailing SYRK/GEMM update).  info[b] = -7 anywhere (a hand-off of the one-launch path
 * timed out) invalidates the WHOLE batch: matrices the aborting workgroups had not reached are left unfactored. */
/* Which factorisation matrices of order M take: 1 = the one-launch tile dataflow (Cholesky and, with_inverse, the
 * triangular inverse in the same launch; csrc/coop.hip), 0 = one launch per step of the blocked algorithm (orders whose
 * task list exceeds the kernel-argument space, or GPZ_FACTOR_PATH=launches in the environment). */
int gpz_factor_path(int64_t M, int32_t with_inverse);
size_t gpz_potrf_workspace_bytes(int64_t M, int64_t batch);
int gpz_potrf_batched(void* A, int32_t dtype, int64_t M, int64_t lda, int64_t stride_a, int64_t batch,
                      int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* X = Lc^{-1} B for `batch` lower-triangular (M,M) factors and (M,N) right-hand sides stored as `dtype`
 * (fp64 arithmetic), X written over B.  Replaces torch.linalg.solve_triangular at gp.py:276 (and each half
 * of cholesky_solve at gp.py:218, 365).  Blocked forward substitution: the 128x128 diagonal blocks are
 * inverted in LDS, then one MFMA GEMM per block row applies X_k = D_k (B_k - sum_{j<k} L_kj X_j) in place.
 * (gpz_svgp_forward does not call this: it forms the explicit inverse once per evaluation and applies it to
 * every N-chunk as one triangular product.) */
size_t gpz_trsm_workspace_bytes(int64_t M, int64_t N, int64_t batch);
int gpz_trsm_lln_batched(const void* Lc, int64_t ldl, int64_t stride_l, void* B, int64_t ldb,
                         int64_t stride_b, int32_t dtype, int64_t M, int64_t N, int64_t batch, void* ws,
                         size_t ws_bytes, void* stream);

/* One evaluation of the SVGP / WSVGP forward pass and (optionally) the
 * closed-form Gaussian ELBO, batched over L latents and tiled over N:
 *   Kzz(+jitter) -> Cholesky -> L^{-1} -> per N-chunk { Kzx fill, Wt = L^{-1} Kzx,
 *   Lu^T Wt, column reductions } -> q(F) mean / scale, KL per latent, ELBO.
 * Replaces WSVGP.forward gp.py:260-306, SVGP.forward gp.py:183-232 (+
 * svgp_forward utilities.py:382-397), MGGP_* gp.py:341-399, whitened_KL
 * utilities.py:27-36, kl_divergence(qU,pU) at utilities.py:481, and the ELBO
 * assembly of mggp_test_exact.ipynb:157-159.
 * dtype = storage/GEMM type of X, Z, mu, Lu_raw, y, mean, scale (k.dtype must
 * match).  Kzz, its Cholesky factor and inverse are carried in fp64 in both modes.
 */
/* gpz_svgp_problem.flags: which kernels the forward pass takes for its two big fp32 products (results agree bit for
 * bit in Wt on every path; 0 = the library's choice: the fill + wide-tile products for M > 512, the panel kernel for
 * M <= 512 where it is the faster one -- see GPZ_SVGP_PANEL_PRODUCTS) */
#define GPZ_SVGP_MATERIALIZE_KZX 1  /* write every Kzx chunk to HBM with the stand-alone fill and run the triangular
                                     * product on it -- the reference's structure, gp.py:255 + :276 */
#define GPZ_SVGP_NARROW_TILES 2     /* the 128 x 128-tile kernel every other precision uses (csrc/gemm.hip) instead of
                                     * the wide-tile one (csrc/gemmw.hip); implies a materialised Kzx */
#define GPZ_SVGP_GENERATE_KZX 4     /* fp32 RBF / Matern-3/2, d <= 2: stage 1 generates its covariance operand inside the
                                     * product (csrc/gemmw.hip) and Kzx is never written */
#define GPZ_SVGP_PANEL_PRODUCTS 8   /* fp32, M <= 512: both products panel by panel in ONE launch (csrc/gemmp.hip): a workgroup
                                     * holds 64 columns x all rows in LDS from the covariance to the column statistics.
                                     * RBF / Matern-3/2 on <= 2-D inputs: the panel is computed inside the kernel and Kzx
                                     * never exists; other kernel families: it is read from the stand-alone fill's Kzx.
                                     * Wt reaches memory only when retained.  Same Wt bits; mean / scale differ from the
                                     * tile path by fp32 rounding (other summation order of the statistics).  The
                                     * library's own choice (flags == 0) in the first case for every M <= 512 and in the
                                     * second for 128 < M <= 512; with this flag wherever it applies; ignored elsewhere
                                     * and next to any of the three flags above. */
/* gpz_svgp_backward only: which form the N-sized work of the pass takes (0 = the library's choice by N / M; the
 * gradients agree to rounding).  ALGEBRA: one weighted symmetric accumulation H += W diag(gv2) W^T per chunk (and, with
 * grad_theta / grad_Z, one dense product for Kbar_x) plus M x M products; CLASSIC: the products autograd would run --
 * Pbar, W Pbar^T (and Wbar, Kbar_x, Kbar_x W^T). */
#define GPZ_SVGP_BACKWARD_ALGEBRA 16
#define GPZ_SVGP_BACKWARD_CLASSIC 32

typedef struct gpz_svgp_problem {
  gpz_kernel_desc k;
  int32_t dtype;
  int32_t whitened;      /* 1: WSVGP (gp.py:260), 0: SVGP (gp.py:183) */
  int32_t d;             /* input dimension */
  int32_t flags;         /* GPZ_SVGP_* bits; 0 = defaults */
  int64_t N, M;
  const void* X;         /* (N,d) */
  const void* Z;         /* (M,d) */
  const int64_t* gX;     /* (N,) MGGP only */
  const int64_t* gZ;     /* (M,) MGGP only */
  const void* mu;        /* (L,M) */
  const void* Lu_raw;    /* (L,M,M) unconstrained: diag is exponentiated */
  double jitter;
  double var_clamp_min;  /* un-whitened only: clamp(cov, min) gp.py:228 (1e-6) / :378 (5e-2) */
  const void* y;         /* (L,N) targets or NULL: no likelihood terms */
  double noise_sd;       /* softplus(noise) of likelihoods.py:33 */
  /* outputs (any may be NULL) */
  void* mean;            /* (L,N) dtype */
  void* scale;           /* (L,N) dtype: sqrt of the predictive variance */
  void* Lu;              /* (L,M,M) dtype: constrained scale_tril of q(U) */
  void* chol;            /* (L,M,M) dtype: Cholesky factor of Kzz + jitter I */
  double* kl;            /* (L,) */
  double* loglik;        /* (L,) sum_n log N(y; mean, s^2) - var / (2 s^2) */
  double* elbo;          /* (1,) sum_l loglik - kl */
  int32_t* info;         /* (L,) potrf info; info[0] = -1: a group id in gX / gZ is outside [0, n_groups)
                          * (LAPACK's "illegal argument"; the Python wrapper raises IndexError like the reference) */
  /* optional cache of everything that depends only on (Z, kernel hyper-parameters, jitter):
   * chol(Kzz), its inverse and sum(log diag).  NULL: recompute every call like the reference
   * (SURVEY §3.3).  Non-NULL (gpz_svgp_factor_cache_bytes bytes, caller owned): filled when
   * bit 0 of factor_cache_valid is 0, reused when it is 1 -- the caller flips the flag and invalidates it when
   * Z / sigma / lengthscale / group parameters / jitter change (SURVEY §8f "next" #3).  Behind the factor the
   * buffer keeps what the two products need of q(U) (LuE^T, muE, un-whitened LuE) as the LAST call prepared it from
   * (mu, Lu_raw); bit 1 (factor_cache_valid == 3) tells gpz_svgp_backward that this is the very (mu, Lu_raw) it is
   * called with -- the backward pass of the forward that wrote the buffer -- so it is not prepared a second time. */
  void* factor_cache;
  int64_t factor_cache_valid;
  /* Optional retained Wt = Linv Kzx of every N-chunk plus its column-sum partials, for training:
   * non-NULL (gpz_svgp_wt_cache_bytes bytes, caller owned) makes gpz_svgp_forward write them there;
   * gpz_svgp_backward with wt_cache_valid != 0 (same inputs, same `chunk`) reads them instead of
   * rebuilding Kzx and repeating the first triangular product (one of its three big GEMMs). */
  void* wt_cache;
  int64_t wt_cache_valid;
} gpz_svgp_problem;

size_t gpz_svgp_factor_cache_bytes(const gpz_svgp_problem* p);
size_t gpz_svgp_wt_cache_bytes(const gpz_svgp_problem* p, int64_t chunk);

size_t gpz_svgp_workspace_bytes(const gpz_svgp_problem* p, int64_t chunk);
/* Which kernels gpz_svgp_forward takes for the two big products of this problem and chunk (0 = automatic chunk):
 * bit 0 the wide-tile fp32 kernels, bit 1 the generated-Kzx stage 1; 4: the panel kernel (both products in one launch);
 * 0: the 128 x 128-tile kernels; -1: bad problem.
 * Unknown bits in `flags` are an error since ABI 210 (the word was `reserved` before 200: zero it). */
int gpz_svgp_forward_path(const gpz_svgp_problem* p, int64_t chunk);
int gpz_svgp_forward(const gpz_svgp_problem* p, int64_t chunk, void* ws, size_t ws_bytes,
                     void* stream);

/* Backward of gpz_svgp_forward: given dLoss/dmean and dLoss/dscale of q(F) (and the forward's
 * scale) writes dLoss/dmu (L,M) and dLoss/dLu_raw (L,M,M) -- what torch autograd produces through
 * gp.py:276-296 / 218-228 for loss.backward() (utilities.py:485) with frozen kernel
 * hyper-parameters (the training mode of Slideseq_NSF_newest_version.ipynb:500-504) -- and,
 * optionally, the gradients w.r.t. sigma / lengthscale / group parameter and Z.  Same problem
 * description as the forward; its output fields are ignored. */
typedef struct gpz_svgp_grads {
  const void* g_mean;    /* (L,N) dtype */
  const void* g_scale;   /* (L,N) dtype */
  const void* scale;     /* (L,N) dtype: forward output */
  void* grad_mu;         /* (L,M) dtype */
  void* grad_Lu_raw;     /* (L,M,M) dtype, zeros above the diagonal */
  /* optional: gradients w.r.t. the kernel hyper-parameters and the inducing inputs -- what
   * autograd sends through kernel.forward, cholesky and the solves (kernels.py, gp.py:213-219,
   * 270-276).  NULL: frozen hyper-parameters. */
  double* grad_theta;    /* (L,4) fp64: d/dsigma, d/dlengthscale, d/dgroup_a (effective), 0 */
  double* grad_Z;        /* (M,4) fp64: first d columns used */
  const void* g_chol;    /* (L,M,M) dtype or NULL: upstream dLoss/dchol (un-whitened: KL(qU||pU) uses it) */
  /* (L,) fp64 or NULL: upstream dLoss/dkl_l of the per-latent KL the forward pass reports (`kl`).  The
   * KL's own gradient -- kl_divergence(qU, pU) through torch's MVN KL and its autograd in the reference
   * (utilities.py:481) -- is then folded into grad_mu, grad_Lu_raw and, with grad_theta / grad_Z, into
   * the factor's gradient, at no extra matrix product: dKL/dLuE = LuE, dKL/dmuE = muE, plus the
   * log-determinant diagonals (whitened: LuE = Lu, muE = mu, the whitened_KL of utilities.py:27-36). */
  const double* g_kl;
  /* gpz_vnngp_backward only: (N,) int64 permutation of the points, or NULL (identity).  The pass lays its per-point
   * records out in this order; along a space-filling curve the points that share an inducing point are neighbours in
   * memory, which is what its fixed-order gather is bound by.  Any permutation gives the same sums up to the order of
   * their terms (the order is fixed by the permutation: results stay bitwise reproducible). */
  const int64_t* point_order;
} gpz_svgp_grads;

size_t gpz_svgp_backward_workspace_bytes(const gpz_svgp_problem* p, int64_t chunk);
int gpz_svgp_backward(const gpz_svgp_problem* p, const gpz_svgp_grads* g, int64_t chunk, void* ws,
                      size_t ws_bytes, void* stream);

/* Monte-Carlo expected Poisson log-likelihood of the NSF factor models and its gradients, fused
 * (SURVEY §8f "next" #2): replaces get_rate + Poisson(V*Z).log_prob(y).mean(0).sum() and their
 * autograd (likelihoods.py:49-53, 74-97, 199-225; utilities.py:610-616) without materialising the
 * (E,D,N) rate.  fp32.  mean, scale (Lt,N): q(F) moments of all factors; eps (E,Lt,N): the
 * standard-normal draws of rsample; W (D,Lt), V (N,): POSITIVE loadings / size factors (after
 * softplus); y (D,N) counts.  Outputs: loglik (2,) fp64: [0] = (1/E) sum_e sum_dn [y log(VZ) - VZ],
 * [1] = sum_dn lgamma(y+1) (0 unless with_lgamma; Poisson.log_prob = [0] - [1]); and
 * d loglik[0] / d{mean, scale, W, V}.  Lt <= 64 factors; E <= 32 samples per call (more samples: one call per group
 * of 32, as gpzoo_amd/ops.py does; y is read once per pass whatever E is).  The three dense products (rate, dW,
 * d exp F) run on MFMA. */
size_t gpz_poisson_nsf_workspace_bytes(int64_t N, int64_t D, int32_t Lt, int32_t E);
int gpz_poisson_nsf(const float* mean, const float* scale, const float* eps, const float* W,
                    const float* V, const float* y, int64_t N, int64_t D, int32_t Lt, int32_t E,
                    int32_t with_lgamma, double* loglik, float* dmean, float* dscale, float* dW,
                    float* dV, void* ws, size_t ws_bytes, void* stream);

/* K nearest rows of Z for every row of X, ascending by (Euclidean distance, index): the neighbour
 * bookkeeping of VNNGP, argsort(cdist(X, Z))[:, :K] (gp.py:31, 64).  idx (N,K) int64.  K <= 32. */
int gpz_knn(const void* X, int64_t N, const void* Z, int64_t M, int32_t d, int32_t K, int32_t dtype,
            int64_t* idx, void* stream);

/* VNNGP.forward (gp.py:21-122), RBF family only (the kernels with return_distance).  Uses the
 * problem's X, Z, kernel, mu, Lu_raw, jitter, var_clamp_min (the reference clamps at 5e-2) and
 * writes mean, scale (L,N), optionally Lu and chol, and info.  idx: (N,K) neighbour lists from
 * gpz_knn, or NULL to compute them here. */
size_t gpz_vnngp_workspace_bytes(const gpz_svgp_problem* p, int32_t K);
/* Hand-off from gpz_vnngp_forward to the gpz_vnngp_backward of the same call: with `factor_cache` pointing at
 * gpz_vnngp_state_bytes(p) bytes (caller owned) the forward leaves Kzz + jitter I, its factor, Lu, S = Lu Lu^T and -- when
 * it evaluates `kl` -- L^{-1}, L^{-1} Lu, L^{-1} mu there; a backward pass given the same buffer with factor_cache_valid =
 * 1 (factor, Lu, S) | 4 (the KL operands) reads them instead of forming them again.  The backward pass uses parts of the
 * buffer as scratch: it is valid for ONE backward.  NULL: every pass forms what it needs in its workspace. */
size_t gpz_vnngp_state_bytes(const gpz_svgp_problem* p);
int gpz_vnngp_forward(const gpz_svgp_problem* p, int32_t K, const int64_t* idx, void* ws,
                      size_t ws_bytes, void* stream);

/* Backward of VNNGP.forward as loss.backward() runs it through the reference's autograd graph
 * (utilities.py:485 over gp.py:21-122): given dLoss/dmean, dLoss/dscale (and, for the kernel
 * hyper-parameters, dLoss/dchol from KL(qU || pU)) writes grad_mu, grad_Lu_raw and, when
 * grad_theta / grad_Z are non-NULL, the gradients w.r.t. (sigma, lengthscale) per latent and Z.
 * The neighbour table is a constant of the graph (argsort has no gradient).  `scale` of
 * gpz_svgp_grads is unused (the variance is recomputed); `g_kl` folds the gradient of the forward's
 * per-latent KL(qU || pU) (problem field `kl`) in, as in gpz_svgp_backward.  The K-sparse terms (the sums over the
 * points that name an inducing point) are formed in a fixed order over the inverted neighbour table -- no atomics on
 * values, bitwise reproducible.  A caller-supplied idx may repeat a neighbour inside a row (checked on the device; those
 * calls add lane after lane); its entries must lie in [0, M).  Limits of the backward pass beyond the forward's:
 * M <= 8192 (two fp64 rows of the padded order in 128 KB of LDS) and N * K < 2^31. */
size_t gpz_vnngp_backward_workspace_bytes(const gpz_svgp_problem* p, int32_t K);
int gpz_vnngp_backward(const gpz_svgp_problem* p, const gpz_svgp_grads* g, int32_t K,
                       const int64_t* idx, void* ws, size_t ws_bytes, void* stream);

/* Moments from a caller-supplied W (L,N,M): WSVGP.forward_precomputed, gp.py:308-322
 * (cov = clamp(sigma^2 - sum W^2, 0) + sum (W Lu)^2, mean = W mu).  sigma (L,), mu (L,M),
 * Lu_raw (L,M,M) -> mean, scale (L,N) and the constrained Lu (L,M,M, may be NULL). */
size_t gpz_wsvgp_precomputed_workspace_bytes(int64_t L, int64_t N, int64_t M, int32_t dtype);
int gpz_wsvgp_precomputed(const void* W, const void* sigma, const void* mu, const void* Lu_raw,
                          int64_t L, int64_t N, int64_t M, int32_t dtype, void* mean, void* scale,
                          void* Lu, void* ws, size_t ws_bytes, void* stream);

/* Backward of gpz_wsvgp_precomputed -- what loss.backward() sends through gp.py:308-322 to mu, Lu and the
 * kernel's sigma (W is the caller's constant): given dLoss/dmean, dLoss/dscale and the forward's scale (L,N)
 * writes grad_mu (L,M), grad_Lu_raw (L,M,M, zeros above the diagonal) and, if non-NULL, grad_sigma (L,) fp64. */
size_t gpz_wsvgp_precomputed_backward_workspace_bytes(int64_t L, int64_t N, int64_t M, int32_t dtype);
int gpz_wsvgp_precomputed_backward(const void* W, const void* sigma, const void* mu, const void* Lu_raw,
                                   int64_t L, int64_t N, int64_t M, int32_t dtype, const void* g_mean,
                                   const void* g_scale, const void* scale, void* grad_mu, void* grad_Lu_raw,
                                   double* grad_sigma, void* ws, size_t ws_bytes, void* stream);

/* Multi-GPU: latent GPs shard across ranks with no data-path collective; the only exchange is
 * the sum of each rank's partial ELBO -- one ncclAllReduce(sum, fp64) over RCCL/xGMI.  The reference has no
 * distributed code to cite; this is the entry SURVEY §8b proposes.  One communicator per process
 * (rank <-> GPU): rank 0 calls gpz_comm_unique_id and hands the 128 bytes to every rank (file, socket, MPI,
 * torch's store, ...), each rank then calls gpz_comm_init with its HIP device current.  `comm` may equally be
 * an ncclComm_t the caller created itself.  gpz_allreduce_sum_f64 sums buf[0..n) in place over the ranks,
 * asynchronously on `stream`.  RCCL is bound at run time (dlopen): the rest of the library works without it. */
int gpz_comm_unique_id(void* id128_host);
int gpz_comm_init(void** comm_out, int32_t world, int32_t rank, const void* id128_host);
int gpz_allreduce_sum_f64(void* comm, double* buf, int64_t n, void* stream);
/* The latent-sharded Poisson NSF step (reference likelihoods.py:49-53, 74-97: rate = softplus(W) @ exp(F) mixes the
 * latents, SURVEY §8e "Caveat"): every rank gathers q(F)'s moments of all latents -- gpz_allgather, `bytes_per_rank`
 * bytes from each rank into recv[rank * bytes_per_rank ...), 2 L N_b s bytes in all against D N_b s for exchanging
 * partial rates -- runs gpz_poisson_nsf on its block of genes, and the gradients w.r.t. the moments return to the
 * latents' owners by gpz_reduce_scatter_sum_f32 (recv = this rank's n_per_rank block of the element-wise sum of every
 * rank's world * n_per_rank values; send and recv may not overlap); gpz_allreduce_sum_f32 sums replicated fp32
 * gradients (the size factors V) in place.  All asynchronous on `stream`. */
int gpz_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int gpz_reduce_scatter_sum_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream);
int gpz_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);
int gpz_comm_destroy(void* comm);

/* Timing hooks used by bench.py: HIP events recorded on `stream` around the
 * dominant kernels of the last gpz_svgp_forward call (roofline.achieved).  These events are the only HIP objects
 * the library ever creates (streams always come from the caller): gpz_profile_enable(1) starts recording,
 * gpz_profile_enable(0) stops and DESTROYS every event, so nothing of the library's outlives it at process exit. */
int gpz_profile_enable(int32_t on);
int gpz_profile_read(double* ms_out, int32_t* counts_out, int32_t n_slots); /* host arrays */

#ifdef __cplusplus
}
#endif
#endif /* GPZOO_HIP_H */
