// svgp.hip -- one SVGP / WSVGP forward pass + closed-form Gaussian ELBO.
//
// Replaces WSVGP.forward (gp.py:260-306), SVGP.forward (gp.py:183-232 with
// svgp_forward utilities.py:382-397), the MGGP variants (gp.py:341-399),
// whitened_KL (utilities.py:27-36), kl_divergence(qU, pU) (utilities.py:481) and
// the ELBO assembly of mggp_test_exact.ipynb:157-159.
//
// With Linv = chol(Kzz + jitter I)^{-1} (fp64, factor.hip) both variants reduce to
//   Wt   = Linv  Kzx                       (lower-triangular x dense, MFMA)
//   P    = LuE^T Wt                        (upper-triangular x dense, MFMA)
//   mean = muE^T Wt,  s1 = colsum(Wt^2),  s2 = colsum(P^2)
// whitened   (gp.py:276-296): LuE = Lu,       muE = mu,      var = max(s^2 - s1, 0) + s2
// un-whitened (gp.py:218-228): LuE = Linv Lu, muE = Linv mu, var = max(s^2 - s1 + s2, clamp)
// because W = Kxz Kzz^{-1} = Wt^T Linv, so W mu = Wt^T (Linv mu), diag(W Kzz W^T) =
// colsum(Wt^2) and diag(W S W^T) = colsum(((Linv Lu)^T Wt)^2).
// N is processed in chunks so Kzx / Wt only ever exist one chunk at a time; P is
// never stored.  All reductions are slab-based (no atomics): bitwise reproducible.
#include "common.h"
#include "gemmw.h"
#include "gemmp.h"
#include "factor.h"
#include "gemm.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace gpz {

struct KgradArgs {
  const void* Kbar; int64_t ld, stride;
  const void* Z; const void* X;
  const int64_t* gZ; const int64_t* gX;
  const void* sigma; const void* ell; const void* ga; const void* gr2;
  double gpow, scalar_scale;
  int64_t M, ncols, Mp;
  int d, G;
  double* acc;
};
int kgrad_launch(int dtype, int kind, const KgradArgs& a, int L, hipStream_t s);

constexpr int NB = 128;

// Schedule of the two big products (gemm.h): strips of `cols` column tiles per (latent, XCD) unit, each workgroup
// computing `tpw` tiles of its row tile back to back (fp32 statistics epilogues only).  Config 3, stage 1 / stage 2:
// 16 x 1 134.5 / 137.0 TF, 32 x 2 135.0 / 138.9, 16 x 2 133.7 / 137.7, 48 x 3 133.9 / 138.8, 64 x 4 96 / 101.
struct ProductSchedule { int cols, tpw; };
// nt = column tiles of the launch.  The strips of a launch get (nearly) equal widths: an XCD takes every 8th unit, so
// with a ragged last strip and a strip count that divides 8 (N_b = 7000: 55 column tiles = 32 + 23) the odd XCDs would
// get all the short units and idle a quarter of the launch (minibatch step 52.4 -> 48.9 ms with 28 + 27).
template <typename T>
static ProductSchedule product_schedule(bool stats_epilogue, int nt) {
  const int tpw = (sizeof(T) == 4 && stats_epilogue) ? 2 : 1;
  const int full = 16 * tpw;
  const int strips = (nt + full - 1) / full;
  int w = (nt + strips * tpw - 1) / (strips * tpw);     // workgroups per row tile and strip
  if (w > 16) w = 16;
  return ProductSchedule{w * tpw, tpw};
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  // fixed-shape tree: lanes -> waves -> block (deterministic)
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;  // valid on thread 0
}

// Constrained scale_tril of q(U) from the raw parameter (gp.py:220/278: strict lower
// triangle kept, diagonal exponentiated), emitted in up to three forms:
//   LuT   (L,Mp,Mp) TG  transposed (upper triangular), zero padded      [whitened: the stage-2 operand]
//   LuD   (L,Mp,Mp) f64 as is (lower triangular), zero padded           [un-whitened: input to Linv * Lu]
//   LuOut (L,M,M)   TIO for MultivariateNormal(scale_tril=...)
// and per-block partial sums of ||Lu||_F^2 and of the raw diagonal (= log diag Lu).
template <typename T>
__global__ __launch_bounds__(256) void lu_prepare_kernel(const T* __restrict__ raw, int64_t M, int64_t Mp,
                                                        T* __restrict__ LuT, double* __restrict__ LuD,
                                                        T* __restrict__ LuOut, double* __restrict__ part,
                                                        T* __restrict__ LuN = nullptr) {
  __shared__ double tile[32][33];
  __shared__ double sh[8];
  const int l = blockIdx.z;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double fro = 0.0, ld = 0.0;
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t i = i0 + rr, j = j0 + tx;
    double v = 0.0;
    if (i < M && j < M && j <= i) {
      const double x = (double)raw[(int64_t)l * M * M + i * M + j];
      if (i == j) { v = exp(x); ld += x; } else v = x;
      fro += v * v;
    }
    tile[rr][tx] = v;
    if (LuOut && i < M && j < M) LuOut[(int64_t)l * M * M + i * M + j] = (T)v;
    if (LuD && i < Mp && j < Mp) LuD[(int64_t)l * Mp * Mp + i * Mp + j] = v;
    if (LuN && i < Mp && j < Mp) LuN[(int64_t)l * Mp * Mp + i * Mp + j] = (T)v;   // padded, lower, not transposed
  }
  __syncthreads();
  if (LuT)
    for (int rr = ty; rr < 32; rr += 8) {
      const int64_t jt = j0 + rr, it = i0 + tx;  // LuT[j][i] = Lu[i][j]
      if (jt < Mp && it < Mp) LuT[(int64_t)l * Mp * Mp + jt * Mp + it] = (T)tile[tx][rr];
    }
  const double f = block_sum(fro, sh);
  const double g = block_sum(ld, sh);
  if (threadIdx.x == 0) {
    const int64_t nb = (int64_t)gridDim.x * gridDim.y, b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    part[((int64_t)l * 2 + 0) * nb + b] = f;
    part[((int64_t)l * 2 + 1) * nb + b] = g;
  }
}

// dst (L,Mp,Mp) T = transpose(src (L,Mp,Mp) f64), plus per-block partial ||src||_F^2.
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const double* __restrict__ src, int64_t Mp,
                                                            T* __restrict__ dst, double* __restrict__ part) {
  __shared__ double tile[32][33];
  __shared__ double sh[8];
  const int l = blockIdx.z;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double fro = 0.0;
  for (int rr = ty; rr < 32; rr += 8) {
    const double v = src[(int64_t)l * Mp * Mp + (i0 + rr) * Mp + j0 + tx];
    tile[rr][tx] = v;
    fro += v * v;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8)
    dst[(int64_t)l * Mp * Mp + (j0 + rr) * Mp + i0 + tx] = (T)tile[tx][rr];
  const double f = block_sum(fro, sh);
  if (threadIdx.x == 0) {
    const int64_t nb = (int64_t)gridDim.x * gridDim.y, b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    part[(int64_t)l * nb + b] = f;
  }
}

template <typename T>
__global__ void cast_kernel(const double* __restrict__ src, T* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (T)src[i];
}

// Cholesky factor out: (L,Mp,Mp) f64 -> (L,M,M) T with zeros above the diagonal; and
// sum of log diag per latent (one block per latent along y == 0 row).
template <typename T>
__global__ __launch_bounds__(256) void chol_out_kernel(const double* __restrict__ Lc, int64_t Mp, int64_t M,
                                                      T* __restrict__ out, double* __restrict__ logdiag) {
  __shared__ double sh[8];
  const int l = blockIdx.y;
  const double* src = Lc + (int64_t)l * Mp * Mp;
  if (out) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < M * M; e += (int64_t)gridDim.x * 256) {
      const int64_t i = e / M, j = e - i * M;
      out[(int64_t)l * M * M + e] = (j <= i) ? (T)src[i * Mp + j] : (T)0;
    }
  }
  if (blockIdx.x == 0) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < M; i += 256) s += log(src[i * (Mp + 1)]);
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) logdiag[l] = t;
  }
}

// muE: whitened -> mu itself; un-whitened -> Linv * mu (fp64 GEMV, one block per 64 rows).
// Writes the padded GEMM-precision vector and per-block partial ||muE||^2.
template <typename T>
__global__ __launch_bounds__(256) void mu_prepare_kernel(const T* __restrict__ mu, int64_t M, int64_t Mp,
                                                        const double* __restrict__ Linv, T* __restrict__ muE,
                                                        double* __restrict__ part) {
  __shared__ double sh[8];
  __shared__ double rowacc[64];
  const int l = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  double sq = 0.0;
  if (!Linv) {
    if (threadIdx.x < 64) {
      const int64_t i = r0 + threadIdx.x;
      const double v = (i < M) ? (double)mu[(int64_t)l * M + i] : 0.0;
      muE[(int64_t)l * Mp + i] = (T)v;
      sq = v * v;
    }
  } else {
    // 4 threads per row, each strides over k; Linv is lower triangular: k <= i
    const int rr = threadIdx.x >> 2, part4 = threadIdx.x & 3;
    const int64_t i = r0 + rr;
    double acc = 0.0;
    const double* Lrow = Linv + (int64_t)l * Mp * Mp + i * Mp;
    const int64_t kmax = (i < M) ? i : -1;
    for (int64_t k = part4; k <= kmax; k += 4) acc += Lrow[k] * (double)mu[(int64_t)l * M + k];
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part4 == 0) rowacc[rr] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
      const double v = rowacc[threadIdx.x];
      muE[(int64_t)l * Mp + r0 + threadIdx.x] = (T)v;
      sq = v * v;
    }
  }
  const double t = block_sum(sq, sh);
  if (threadIdx.x == 0) part[(int64_t)l * gridDim.x + blockIdx.x] = t;
}

// q(F) moments and likelihood terms for one chunk of columns.
template <typename T>
struct FinalizeArgs {
  const T* ps1; const T* pm1; const T* ps2;  // [L][mt][nc]
  const T* sigma;                            // (L,)
  const T* y;                                // (L,N) or null
  T* mean; T* scale;                         // (L,N) or null
  double* part;                              // [L][nfb_total]
  int64_t N, n0, nc, nfb_total, fb0;
  int mt, mt1, whitened;                     // row tiles of ps2, and of ps1/pm1
  double clamp_min, noise_sd;
};

template <typename T>
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs<T> a) {
  __shared__ double sh[8];
  const int l = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = a.n0 + c;
  double term = 0.0;
  if (c < a.nc && n < a.N) {
    T s1 = 0, m1 = 0, s2 = 0;
    for (int i = 0; i < a.mt1; ++i) {
      const int64_t o = ((int64_t)l * a.mt1 + i) * a.nc + c;
      s1 += a.ps1[o]; m1 += a.pm1[o];
    }
    for (int i = 0; i < a.mt; ++i) s2 += a.ps2[((int64_t)l * a.mt + i) * a.nc + c];
    const T sg = a.sigma[l];
    T var;
    if (a.whitened) {
      var = sg * sg - s1;
      var = (var > (T)0 ? var : (T)0) + s2;          // gp.py:286-288
    } else {
      var = sg * sg - s1 + s2;                        // utilities.py:395
      var = var > (T)a.clamp_min ? var : (T)a.clamp_min;  // gp.py:228 / :378
    }
    if (a.mean) a.mean[(int64_t)l * a.N + n] = m1;
    if (a.scale) a.scale[(int64_t)l * a.N + n] = sqrt(var);
    if (a.y) {
      const double s2n = a.noise_sd * a.noise_sd;
      const double r = (double)a.y[(int64_t)l * a.N + n] - (double)m1;
      term = -0.5 * log(6.283185307179586476925 * s2n) - (r * r + (double)var) / (2.0 * s2n);
    }
  }
  const double t = block_sum(term, sh);
  if (threadIdx.x == 0) a.part[(int64_t)l * a.nfb_total + a.fb0 + blockIdx.x] = t;
}

// Final per-latent sums and the scalar ELBO (single block; fixed summation order).
struct ReduceArgs {
  const double* ll_part; int64_t nfb;          // [L][nfb]
  const double* lu_part; int64_t nlu;          // [L][2][nlu]: frob^2, sum raw diag
  const double* fro_part; int64_t nfro;        // [L][nfro] (un-whitened: ||Linv Lu||_F^2) or null
  const double* mu_part; int64_t nmu;          // [L][nmu]
  const double* chol_logdiag;                  // (L,)
  double* kl; double* loglik; double* elbo; double* contrib;
  int L, whitened; int64_t M;
  int has_y;
};

// One block per latent: KL and log-likelihood sums; the per-latent ELBO contribution goes to
// `contrib`, summed in index order by elbo_sum_kernel (fixed order: bitwise reproducible).
__global__ __launch_bounds__(256) void reduce_kernel(ReduceArgs a) {
  __shared__ double sh[8];
  const int l = blockIdx.x;
  double v = 0.0;
  for (int64_t i = threadIdx.x; i < a.nfb; i += 256) v += a.ll_part[(int64_t)l * a.nfb + i];
  const double ll = block_sum(v, sh);
  v = 0.0;
  for (int64_t i = threadIdx.x; i < a.nlu; i += 256) v += a.lu_part[((int64_t)l * 2) * a.nlu + i];
  double fro = block_sum(v, sh);
  v = 0.0;
  for (int64_t i = threadIdx.x; i < a.nlu; i += 256) v += a.lu_part[((int64_t)l * 2 + 1) * a.nlu + i];
  const double logdiag_q = block_sum(v, sh);
  if (a.fro_part) {
    v = 0.0;
    for (int64_t i = threadIdx.x; i < a.nfro; i += 256) v += a.fro_part[(int64_t)l * a.nfro + i];
    fro = block_sum(v, sh);
  }
  v = 0.0;
  for (int64_t i = threadIdx.x; i < a.nmu; i += 256) v += a.mu_part[(int64_t)l * a.nmu + i];
  const double mu2 = block_sum(v, sh);
  if (threadIdx.x == 0) {
    double kl;
    if (a.whitened) kl = 0.5 * (-2.0 * logdiag_q + fro + mu2 - (double)a.M);          // utilities.py:34
    else kl = a.chol_logdiag[l] - logdiag_q + 0.5 * (fro + mu2 - (double)a.M);          // MVN||MVN closed form
    if (a.kl) a.kl[l] = kl;
    if (a.loglik) a.loglik[l] = ll;
    a.contrib[l] = (a.has_y ? ll : 0.0) - kl;
  }
}

__global__ void elbo_sum_kernel(const double* __restrict__ contrib, int L, double* __restrict__ elbo) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && elbo) {
    double t = 0.0;
    for (int l = 0; l < L; ++l) t += contrib[l];
    *elbo = t;
  }
}

struct Plan {
  int64_t L, N, M, Mp, nblk, nc, nchunks, nfb_chunk, nfb_total, nlu, nmu;
  bool f32;
};

static Plan make_plan(const gpz_svgp_problem* p, int64_t chunk) {
  Plan pl;
  pl.L = p->k.n_latent; pl.N = p->N; pl.M = p->M; pl.Mp = pad_up(p->M); pl.nblk = pl.Mp / NB;
  pl.f32 = p->dtype == GPZ_F32;
  const int64_t esz = pl.f32 ? 4 : 8;
  if (chunk <= 0) {
    // auto: Kzx + Wt chunk buffers of ~6 GiB, at least 2048 columns
    chunk = (int64_t)(6.0 * (1ull << 30) / (2.0 * pl.L * pl.Mp * esz));
    if (chunk < 2048) chunk = 2048;
    // the wide-tile fp32 products address a latent's (Mp, chunk) panel through 32-bit byte offsets (gemmw.hip): an
    // automatic chunk never leaves their range (L = 1, M = 2048, N >= 262 144 used to, and silently took the
    // 128 x 128-tile kernels)
    if (pl.f32) {
      const int64_t cap = (((1ll << 31) - 1) / (pl.Mp * 4)) / NB * NB - NB;
      if (cap >= NB && chunk > cap) chunk = cap;
    }
  }
  chunk = pad_up(chunk > pl.N ? pl.N : chunk);
  pl.nc = chunk;
  pl.nchunks = (pl.N + chunk - 1) / chunk;
  pl.nfb_chunk = (chunk + 255) / 256;
  pl.nfb_total = pl.nfb_chunk * pl.nchunks;
  pl.nlu = (pl.Mp / 32) * (pl.Mp / 32);
  pl.nmu = pl.Mp / 64;
  return pl;
}

template <typename T>
struct Buffers {
  double *Kzz, *Dinv, *Linv, *Tmp, *LuD, *LuW;
  uint32_t* fsync;            // tickets and flags of the one-launch factorisation
  T *LinvG, *LuT, *muE, *Kc, *Wc, *ps1, *pm1, *ps2;
  double *ll_part, *lu_part, *fro_part, *mu_part, *chol_logdiag, *contrib;
  size_t bytes;
};

// The factor cache holds what depends only on (Z, kernel hyper-parameters, jitter): the Cholesky
// factor of Kzz, its inverse in fp64 and in GEMM precision, and sum(log diag L) per latent.
// Behind it, in the same buffer, what the two products need of q(U) -- LuE^T, muE and (un-whitened) LuE = Linv Lu in
// fp64 -- as the last call prepared them from (mu, Lu_raw): the backward pass of that very call takes them from here
// (bit 1 of factor_cache_valid) instead of preparing them a second time.
template <typename T>
struct FactorCache { double *Kzz, *Linv, *logdiag, *LuW; T *LinvG, *LuT, *muE; size_t bytes; };
template <typename T>
static FactorCache<T> carve_cache(const Plan& pl, bool whitened, void* mem) {
  FactorCache<T> f;
  Carver c(mem);
  const int64_t mm = pl.L * pl.Mp * pl.Mp;
  f.Kzz = c.take<double>(mm);
  f.Linv = c.take<double>(mm);
  f.LinvG = sizeof(T) == 8 ? reinterpret_cast<T*>(f.Linv) : c.take<T>(mm);
  f.logdiag = c.take<double>(pl.L);
  f.LuT = c.take<T>(mm);
  f.muE = c.take<T>(pl.L * pl.Mp);
  f.LuW = whitened ? nullptr : c.take<double>(mm);
  f.bytes = c.used();
  return f;
}

// Retained Wt (training): every chunk's Wt block and its colsum(Wt^2) partials, one full-width slot per chunk.
template <typename T>
struct WtCache {
  T* base; int64_t wt_slot, ps_slot, nchunks;
  T* wt(int64_t ci) const { return base + ci * wt_slot; }
  T* ps1(int64_t ci) const { return base + nchunks * wt_slot + ci * ps_slot; }
  size_t bytes() const { return sizeof(T) * (size_t)(nchunks * (wt_slot + ps_slot)); }
};
template <typename T>
static WtCache<T> wt_cache_of(const Plan& pl, void* mem) {
  WtCache<T> w;
  w.base = static_cast<T*>(mem); w.nchunks = pl.nchunks;
  w.wt_slot = pl.L * pl.Mp * pl.nc; w.ps_slot = pl.L * pl.nblk * pl.nc;
  return w;
}

template <typename T>
static Buffers<T> carve(const Plan& pl, bool whitened, void* ws) {
  Buffers<T> b;
  Carver c(ws);
  const int64_t mm = pl.L * pl.Mp * pl.Mp;
  b.Kzz = c.take<double>(mm);
  b.Dinv = c.take<double>(pl.L * pl.nblk * NB * NB);
  b.Linv = c.take<double>(mm);
  b.Tmp = c.take<double>(mm);
  b.fsync = c.take<uint32_t>(coop_sync_words(pl.Mp, pl.L));
  b.LuD = whitened ? nullptr : c.take<double>(mm);
  b.LuW = whitened ? nullptr : c.take<double>(mm);
  b.LinvG = sizeof(T) == 8 ? reinterpret_cast<T*>(b.Linv) : c.take<T>(mm);
  b.LuT = c.take<T>(mm);
  b.muE = c.take<T>(pl.L * pl.Mp);
  b.Kc = c.take<T>(pl.L * pl.Mp * pl.nc);
  b.Wc = c.take<T>(pl.L * pl.Mp * pl.nc);
  b.ps1 = c.take<T>(pl.L * pl.nblk * pl.nc);
  b.pm1 = c.take<T>(pl.L * pl.nblk * pl.nc);
  b.ps2 = c.take<T>(pl.L * pl.nblk * pl.nc);
  b.ll_part = c.take<double>(pl.L * pl.nfb_total);
  b.lu_part = c.take<double>(pl.L * 2 * pl.nlu);
  b.fro_part = whitened ? nullptr : c.take<double>(pl.L * pl.nlu);
  b.mu_part = c.take<double>(pl.L * pl.nmu);
  b.chol_logdiag = c.take<double>(pl.L);
  b.contrib = c.take<double>(pl.L);
  b.bytes = c.used();
  return b;
}

// Steps shared by the forward and backward passes: Kzz + jitter I (fp64, identity padded), its
// Cholesky factor and inverse, and q(U)'s parameters in the form the two big products need.
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b, size_t nb) {
  const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for (size_t i = t0; i < na; i += step) a[i] = 0u;
  for (size_t i = t0; i < nb; i += step) b[i] = 0u;
}

// Up to eight buffers zeroed by ONE launch (16-byte stores; sizes in bytes, multiples of 16: every buffer here is carved
// at 256-byte boundaries and padded to 128 elements).  hipMemsetAsync is a launch of its own per buffer -- about 10 us of
// host time each, and the backward pass of a small problem started with six of them.
struct ZeroRanges { void* p[8]; size_t n16[8]; int n; };
__global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroRanges z) {
  const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for (int r = 0; r < z.n; ++r) {
    uint4* q = static_cast<uint4*>(z.p[r]);
    for (size_t i = t0; i < z.n16[r]; i += step) q[i] = uint4{0u, 0u, 0u, 0u};
  }
}
struct Zeroer {
  ZeroRanges z{};
  size_t total = 0;
  void add(void* p, size_t bytes) {
    if (!p || !bytes) return;
    z.p[z.n] = p; z.n16[z.n] = (bytes + 15) / 16; total += z.n16[z.n]; ++z.n;
  }
  int run(hipStream_t s) {
    if (!z.n) return 0;
    const unsigned grid = (unsigned)std::min<size_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(grid), dim3(256), 0, s, z);
    GPZ_LAUNCH_OK();
    return 0;
  }
};

template <typename T>
static int prepare_t(const gpz_svgp_problem* p, const Plan& pl, Buffers<T>& b, hipStream_t s) {
  const bool wh = p->whitened != 0;
  const int64_t L = pl.L, M = pl.M, Mp = pl.Mp, mm = Mp * Mp;
  const int L32 = (int)L;
  // info: 0 = fine, k > 0 = leading minor k not positive-definite (potrf), < 0 = a group id out of range (kfill)
  // ... zeroed in ONE small launch together with the flag words of the one-launch factorisation (each separate memset is a
  // launch of its own, and on a short evaluation -- whose previous `info` check left the queue empty -- every launch of
  // this prelude is a host round trip the GPU waits for)
  const bool will_factor = !(p->factor_cache && (p->factor_cache_valid & 1));
  const size_t nsync = will_factor ? factor_sync_clear_words(Mp, L, true) : 0;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(64, (nsync + (size_t)L) / 1024 + 1)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p->info), (size_t)L, b.fsync, nsync);
  GPZ_LAUNCH_OK();
  // 1. Kzz + jitter I (fp64, identity padded), Cholesky, inverse -- or the caller's cached copy
  if (p->factor_cache) {
    FactorCache<T> f = carve_cache<T>(pl, wh, p->factor_cache);
    b.Kzz = f.Kzz; b.Linv = f.Linv; b.LinvG = f.LinvG; b.chol_logdiag = f.logdiag;
    b.LuT = f.LuT; b.muE = f.muE;
    if (!wh) b.LuW = f.LuW;
  }
  // factor_cache_valid: bit 0 the factor, bit 1 the q(U) operands behind it (set by the backward pass of the call that
  // wrote them: same mu / Lu_raw, same factor)
  const bool qu_cached = p->factor_cache && (p->factor_cache_valid & 3) == 3;
  if (!(p->factor_cache && (p->factor_cache_valid & 1))) {
    if (int rc = kfill_padded(&p->k, p->Z, M, Mp, p->Z, M, Mp, p->d, p->gZ, p->gZ, b.Kzz, Mp, mm, p->jitter, 1,
                              GPZ_F64, s, p->info))
      return rc;
    bool wrote32 = false;     // the one-launch factorisation writes the fp32 copy of the inverse itself
    if (int rc = factor_invert_padded(b.Kzz, Mp, L, M, b.Dinv, b.Linv, b.Tmp, b.fsync, p->info, s,
                                      sizeof(T) == 4 ? reinterpret_cast<float*>(b.LinvG) : nullptr, &wrote32, nsync > 0))
      return rc;
    hipLaunchKernelGGL((chol_out_kernel<T>), dim3(p->chol ? 64 : 1, L32), dim3(256), 0, s, b.Kzz, Mp, M,
                       static_cast<T*>(p->chol), b.chol_logdiag);
    GPZ_LAUNCH_OK();
    if (sizeof(T) == 4 && !wrote32) {
      hipLaunchKernelGGL((cast_kernel<T>), dim3(2048), dim3(256), 0, s, b.Linv, b.LinvG, L * mm);
      GPZ_LAUNCH_OK();
    }
  } else {
    if (p->chol) {
      hipLaunchKernelGGL((chol_out_kernel<T>), dim3(64, L32), dim3(256), 0, s, b.Kzz, Mp, M, static_cast<T*>(p->chol),
                         b.chol_logdiag);
      GPZ_LAUNCH_OK();
    }
  }

  // 2. q(U) parameters in the form the two products need
  const dim3 g32((unsigned)(Mp / 32), (unsigned)(Mp / 32), L32);
  if (qu_cached) return 0;
  if (wh) {
    hipLaunchKernelGGL((lu_prepare_kernel<T>), g32, dim3(256), 0, s, static_cast<const T*>(p->Lu_raw), M, Mp, b.LuT,
                       (double*)nullptr, static_cast<T*>(p->Lu), b.lu_part);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((mu_prepare_kernel<T>), dim3((unsigned)pl.nmu, L32), dim3(256), 0, s,
                       static_cast<const T*>(p->mu), M, Mp, (const double*)nullptr, b.muE, b.mu_part);
    GPZ_LAUNCH_OK();
  } else {
    hipLaunchKernelGGL((lu_prepare_kernel<T>), g32, dim3(256), 0, s, static_cast<const T*>(p->Lu_raw), M, Mp,
                       (T*)nullptr, b.LuD, static_cast<T*>(p->Lu), b.lu_part);
    GPZ_LAUNCH_OK();
    GemmParams<double> g;  // LuW = Linv * Lu  (lower x lower -> lower)
    g.A = b.Linv; g.lda = Mp; g.sA0 = mm;
    g.B = b.LuD; g.ldb = Mp; g.sB0 = mm;
    g.C = b.LuW; g.ldc = Mp; g.sC0 = mm;
    g.nb0 = L32; g.mt = g.nt = (int)pl.nblk; g.K = (int)Mp; g.flags = GF_A_LOWER | GF_B_LOWER | GF_TILES_LOWER;
    GPZ_HIP_OK(hipMemsetAsync(b.LuW, 0, sizeof(double) * L * mm, s));
    if (int rc = gemm_launch(g, EPI_STORE, s)) return rc;
    hipLaunchKernelGGL((transpose_cast_kernel<T>), g32, dim3(256), 0, s, b.LuW, Mp, b.LuT, b.fro_part);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((mu_prepare_kernel<T>), dim3((unsigned)pl.nmu, L32), dim3(256), 0, s,
                       static_cast<const T*>(p->mu), M, Mp, b.Linv, b.muE, b.mu_part);
    GPZ_LAUNCH_OK();
  }

  return 0;
}

// The panel kernel (gemmp.hip: both products of an fp32 chunk with Mp <= 512 in one launch) is the library's choice
// where it is the faster one (evaluations, same box, two tile launches + fill -> panel kernel):
//   * wherever it computes its covariance panel itself (cov.h: RBF / Matern-3/2 on 1-D / 2-D inputs) -- no fill launch, no
//     Kzx: configs[1] (M=512) 2.36 -> 2.27 ms, N=200k M=256 L=32 9.47 -> 8.11 ms, N=100k M=384 L=16 5.71 -> 4.80 ms,
//     N=60k M=128 L=32 1.28 -> 1.22 ms; with Wt retained for the backward pass 2.36 -> 2.26, 9.48 -> 8.63, 5.69 -> 5.13 ms;
//   * with the fill's Kzx as its operand (the other kernel families) for Mp = 256 ... 512 (multi-group RBF, fp32: N=50k M=512
//     L=8 2.38 -> 2.31 ms, N=200k M=256 L=8 2.56 -> 2.41 ms, N=7000 M=500 L=20 1.24 -> 1.17 ms; one 128-block: 0.71 -> 0.76, not taken).
// The choice does not depend on whether Wt is retained: a forward pass gives the same bits either way.  Any of the other
// three flags names a tile path and gets it; GPZ_SVGP_PANEL_PRODUCTS runs the panel kernel wherever it applies;
// GPZ_SVGP_PRODUCTS=panel|tiles overrides the library's choice (A/B timing without rebuilding).
static bool panel_path(const gpz_svgp_problem* p, bool f32, int64_t Mp, int64_t ncp) {
  static const int env = [] {
    const char* e = getenv("GPZ_SVGP_PRODUCTS");
    return !e ? 0 : !strcmp(e, "panel") ? 1 : !strcmp(e, "tiles") ? 2 : 0;
  }();
  if (!f32 || !panel_supported(Mp, ncp)) return false;
  if (p->flags & (GPZ_SVGP_NARROW_TILES | GPZ_SVGP_MATERIALIZE_KZX | GPZ_SVGP_GENERATE_KZX)) return false;
  if (p->flags & GPZ_SVGP_PANEL_PRODUCTS) return true;
  return env == 1 || (env == 0 && (panel_generates(p->k.kind, p->d) || Mp >= 256));
}

template <typename T>
static int svgp_forward_t(const gpz_svgp_problem* p, int64_t chunk, void* ws, size_t ws_bytes, hipStream_t s) {
  const Plan pl = make_plan(p, chunk);
  const bool wh = p->whitened != 0;
  Buffers<T> b = carve<T>(pl, wh, ws);
  GPZ_REQUIRE(ws_bytes >= b.bytes, "gpz_svgp_forward: workspace too small (%zu < %zu)", ws_bytes, b.bytes);
  const int64_t L = pl.L, M = pl.M, Mp = pl.Mp, N = pl.N, mm = Mp * Mp;
  const int L32 = (int)L;

  if (int rc = prepare_t<T>(p, pl, b, s)) return rc;

  // 3. chunks of columns
  const int64_t esz = sizeof(T);
  const WtCache<T> wtc = wt_cache_of<T>(pl, p->wt_cache);
  const bool narrow = (p->flags & GPZ_SVGP_NARROW_TILES) != 0;
  // the generated-operand stage 1 is built and selectable; the materialised fill + wide-tile product is the default
  // because it is the faster one at the benchmark shape
  const bool fused = !narrow && (p->flags & GPZ_SVGP_GENERATE_KZX) && !(p->flags & GPZ_SVGP_MATERIALIZE_KZX) &&
                     fused1_supported(p->dtype, p->k.kind, p->d);
  for (int64_t ci = 0; ci < pl.nchunks; ++ci) {
    const int64_t n0 = ci * pl.nc;
    const int64_t nreal = (N - n0 < pl.nc) ? N - n0 : pl.nc;
    const int64_t ncp = pad_up(nreal);  // columns computed this chunk
    const int nt = (int)(ncp / NB);
    const ProductSchedule sched = product_schedule<T>(true, nt);
    const bool wide = !narrow && pl.f32 && wide_product_supported(Mp, ncp);    // fp32: 128 x 256 tiles (gemmw.hip)
    // fp32, Mp <= 512, where faster or on request: both products on one column panel held in LDS (gemmp.hip)
    const bool panel = panel_path(p, pl.f32, Mp, ncp) && !fused;
    T* const Wc = p->wt_cache ? wtc.wt(ci) : b.Wc;      // retained for the backward pass when asked for
    T* const ps1 = p->wt_cache ? wtc.ps1(ci) : b.ps1;
    if (fused) {
      // Wt = Linv * k(Z, X) with the covariance generated inside the product: Kzx is never written (gemmw.hip)
      if constexpr (sizeof(T) == 4) {
        Fused1Args fa;
        fa.Linv = b.LinvG; fa.Mp = Mp; fa.Z = static_cast<const float*>(p->Z); fa.M = M;
        fa.X = static_cast<const float*>(p->X) + n0 * p->d; fa.nreal = nreal; fa.d = p->d; fa.kind = p->k.kind; fa.L = L32;
        fa.sigma = static_cast<const float*>(p->k.sigma); fa.ell = static_cast<const float*>(p->k.lengthscale);
        fa.Wt = Wc; fa.ncp = ncp; fa.muE = b.muE; fa.ps_sq = ps1; fa.ps_mu = b.pm1;
        prof_begin(PROF_STAGE1, s);
        if (int rc = fused1_launch(fa, s)) return rc;
        prof_end(PROF_STAGE1, s);
      }
    } else {
      // (the panel kernel computes its covariance panel itself where cov.h covers the kernel: no fill, no Kzx)
      static const bool env_panel_fill = [] { const char* e = getenv("GPZ_PANEL_FILL"); return e && !strcmp(e, "1"); }();
      const bool panel_gen = panel && pl.f32 && panel_generates(p->k.kind, p->d) && !env_panel_fill;
      if (!panel_gen) {
        prof_begin(PROF_KFILL, s);
        if (int rc = kfill_padded(&p->k, p->Z, M, Mp, static_cast<const char*>(p->X) + n0 * p->d * esz, nreal, ncp, p->d,
                                  p->gZ, p->gX ? p->gX + n0 : nullptr, b.Kc, ncp, Mp * ncp, 0.0, 0,
                                  pl.f32 ? GPZ_F32 : GPZ_F64, s, p->info))
          return rc;
        prof_end(PROF_KFILL, s);
      }
      prof_begin(PROF_STAGE1, s);
      bool done = false;
      if constexpr (sizeof(T) == 4) {
        if (panel) {     // Wt = Linv * Kzx, its column statistics AND stage 2's, panel by panel; Wt stored only when retained
          PanelArgs pa = {};
          pa.Linv = b.LinvG; pa.LuT = b.LuT; pa.Wt = p->wt_cache ? Wc : nullptr; pa.muE = b.muE;
          if (panel_gen) {
            pa.Z = static_cast<const float*>(p->Z); pa.X = static_cast<const float*>(p->X) + n0 * p->d;
            pa.sigma = static_cast<const float*>(p->k.sigma); pa.ell = static_cast<const float*>(p->k.lengthscale);
            pa.M = M; pa.nreal = nreal; pa.kind = p->k.kind; pa.d = p->d;
          } else {
            pa.Kzx = b.Kc;
          }
          pa.ps1 = ps1; pa.pm1 = b.pm1; pa.ps2 = b.ps2; pa.Mp = Mp; pa.ncp = ncp; pa.L = L32;
          if (int rc = panel_launch(pa, s)) return rc;
          done = true;
        }
      }
      if constexpr (sizeof(T) == 4) {
        if (wide && !done) {      // Wt = Linv * Kzx on the 128 x 256 tile, with colsum(Wt^2) and muE^T Wt
          WideArgs wa = {};
          wa.A = b.LinvG; wa.B = b.Kc; wa.Mp = Mp; wa.ncp = ncp; wa.L = L32; wa.upper = 0; wa.epilogue = WIDE_STORE_STATS;
          wa.C = Wc; wa.mu = b.muE; wa.ps_sq = ps1; wa.ps_mu = b.pm1;
          if (int rc = wide_product_launch(wa, s)) return rc;
          done = true;
        }
      }
      if (!done) {
        GemmParams<T> g1;  // Wt = Linv * Kzx, with colsum(Wt^2) and muE^T Wt
        g1.A = b.LinvG; g1.lda = Mp; g1.sA0 = mm;
        g1.B = b.Kc; g1.ldb = ncp; g1.sB0 = Mp * ncp;
        g1.C = Wc; g1.ldc = ncp; g1.sC0 = Mp * ncp;
        g1.nb0 = L32; g1.mt = (int)pl.nblk; g1.nt = nt; g1.K = (int)Mp; g1.flags = GF_A_LOWER | GF_GROUP_COLS;
        g1.super_cols = sched.cols; g1.tiles_per_wg = sched.tpw; g1.mu = b.muE; g1.sMu = Mp; g1.ps_sq = ps1; g1.ps_mu = b.pm1; g1.ncols = ncp;
        if (int rc = gemm_launch(g1, EPI_STORE_STATS, s)) return rc;
      }
      prof_end(PROF_STAGE1, s);
    }
    prof_begin(PROF_STAGE2, s);
    if (!panel) {
      bool done = false;
      if constexpr (sizeof(T) == 4) {
        if (wide) {      // colsum((LuE^T Wt)^2) on the 128 x 256 tile
          WideArgs wa = {};
          wa.A = b.LuT; wa.B = Wc; wa.Mp = Mp; wa.ncp = ncp; wa.L = L32; wa.upper = 1; wa.epilogue = WIDE_STATS;
          wa.ps_sq = b.ps2;
          if (int rc = wide_product_launch(wa, s)) return rc;
          done = true;
        }
      }
      if (!done) {
        GemmParams<T> g2;  // colsum((LuE^T Wt)^2)
        g2.A = b.LuT; g2.lda = Mp; g2.sA0 = mm;
        g2.B = Wc; g2.ldb = ncp; g2.sB0 = Mp * ncp;
        g2.nb0 = L32; g2.mt = (int)pl.nblk; g2.nt = nt; g2.K = (int)Mp; g2.flags = GF_A_UPPER | GF_GROUP_COLS;
        g2.super_cols = sched.cols; g2.tiles_per_wg = sched.tpw; g2.ps_sq = b.ps2; g2.ncols = ncp;
        if (int rc = gemm_launch(g2, EPI_STATS, s)) return rc;
      }
    }
    prof_end(PROF_STAGE2, s);
    FinalizeArgs<T> f;
    f.ps1 = ps1; f.pm1 = b.pm1; f.ps2 = b.ps2; f.sigma = static_cast<const T*>(p->k.sigma);
    f.y = static_cast<const T*>(p->y); f.mean = static_cast<T*>(p->mean); f.scale = static_cast<T*>(p->scale);
    f.part = b.ll_part; f.N = N; f.n0 = n0; f.nc = ncp; f.nfb_total = pl.nfb_total; f.fb0 = ci * pl.nfb_chunk;
    f.mt = f.mt1 = (int)pl.nblk; f.whitened = wh; f.clamp_min = p->var_clamp_min; f.noise_sd = p->noise_sd;
    prof_begin(PROF_FINAL, s);
    // blocks beyond this chunk's columns still write a zero partial so the slab is fully defined
    hipLaunchKernelGGL((finalize_kernel<T>), dim3((unsigned)pl.nfb_chunk, L32), dim3(256), 0, s, f);
    GPZ_LAUNCH_OK();
    prof_end(PROF_FINAL, s);
  }

  // 4. per-latent KL / log-likelihood and the scalar ELBO
  ReduceArgs r;
  r.ll_part = b.ll_part; r.nfb = pl.nfb_total; r.lu_part = b.lu_part; r.nlu = pl.nlu;
  r.fro_part = b.fro_part; r.nfro = pl.nlu; r.mu_part = b.mu_part; r.nmu = pl.nmu;
  r.chol_logdiag = b.chol_logdiag; r.kl = p->kl; r.loglik = p->loglik; r.elbo = p->elbo;
  r.L = L32; r.whitened = wh; r.M = M; r.has_y = p->y != nullptr;
  r.contrib = b.contrib;
  hipLaunchKernelGGL(reduce_kernel, dim3(L32), dim3(256), 0, s, r);
  GPZ_LAUNCH_OK();
  hipLaunchKernelGGL(elbo_sum_kernel, dim3(1), dim3(64), 0, s, b.contrib, L32, p->elbo);
  GPZ_LAUNCH_OK();
  return 0;
}


// ---- WSVGP.forward_precomputed (gp.py:308-322): moments from a caller-supplied W (L,N,M) ----
// Wt chunk = transpose(W[:, n0:n0+nc, :]) zero padded to (Mp, ncp), plus per-row sum(W^2) and W.mu.
template <typename T>
__global__ __launch_bounds__(256) void w_transpose_kernel(const T* __restrict__ W, int64_t N, int64_t M, int64_t n0,
                                                         int64_t Mp, int64_t ncp, T* __restrict__ Wt) {
  __shared__ T tile[32][33];
  const int l = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.x * 32, m0 = (int64_t)blockIdx.y * 32;   // chunk column / inducing index
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t n = n0 + c0 + rr, m = m0 + tx;
    tile[rr][tx] = (n < N && m < M) ? W[((int64_t)l * N + n) * M + m] : (T)0;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) Wt[((int64_t)l * Mp + m0 + rr) * ncp + c0 + tx] = tile[tx][rr];
}

template <typename T>
__global__ __launch_bounds__(256) void w_rowstats_kernel(const T* __restrict__ W, const T* __restrict__ mu, int64_t N,
                                                        int64_t M, int64_t n0, int64_t ncp, T* __restrict__ ps1,
                                                        T* __restrict__ pm1) {
  const int l = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + wave, n = n0 + c;
  if (c >= ncp) return;
  T s = 0, m = 0;
  if (n < N)
    for (int64_t k = lane; k < M; k += 64) {
      const T w = W[((int64_t)l * N + n) * M + k];
      s = fma(w, w, s);
      m = fma(w, mu[(int64_t)l * M + k], m);
    }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); m += __shfl_down(m, o); }
  if (lane == 0) { ps1[(int64_t)l * ncp + c] = s; pm1[(int64_t)l * ncp + c] = m; }
}

struct PrePlan { int64_t Mp, nblk, nc, nchunks, nfb_chunk, nlu; };
static PrePlan pre_plan(int64_t L, int64_t N, int64_t M, int esz) {
  PrePlan pl;
  pl.Mp = pad_up(M); pl.nblk = pl.Mp / NB;
  int64_t chunk = (int64_t)(2.0 * (1ull << 30) / ((double)L * pl.Mp * esz));
  if (chunk < 1024) chunk = 1024;
  pl.nc = pad_up(chunk > N ? N : chunk);
  pl.nchunks = (N + pl.nc - 1) / pl.nc;
  pl.nfb_chunk = (pl.nc + 255) / 256;
  pl.nlu = (pl.Mp / 32) * (pl.Mp / 32);
  return pl;
}

template <typename T>
struct PreBuffers { T *LuT, *Wc, *ps1, *pm1, *ps2; double *lu_part, *ll_part; size_t bytes; };
template <typename T>
static PreBuffers<T> pre_carve(const PrePlan& pl, int64_t L, void* ws) {
  PreBuffers<T> b;
  Carver c(ws);
  b.LuT = c.take<T>(L * pl.Mp * pl.Mp);
  b.Wc = c.take<T>(L * pl.Mp * pl.nc);
  b.ps1 = c.take<T>(L * pl.nc);
  b.pm1 = c.take<T>(L * pl.nc);
  b.ps2 = c.take<T>(L * pl.nblk * pl.nc);
  b.lu_part = c.take<double>(L * 2 * pl.nlu);
  b.ll_part = c.take<double>(L * pl.nfb_chunk * pl.nchunks);
  b.bytes = c.used();
  return b;
}

template <typename T>
static int precomputed_t(const void* W, const void* sigma, const void* mu, const void* Lu_raw, int64_t L, int64_t N,
                         int64_t M, void* mean, void* scale, void* Lu, void* ws, size_t ws_bytes, hipStream_t s) {
  const PrePlan pl = pre_plan(L, N, M, sizeof(T));
  PreBuffers<T> b = pre_carve<T>(pl, L, ws);
  GPZ_REQUIRE(ws_bytes >= b.bytes, "gpz_wsvgp_precomputed: workspace too small");
  const int L32 = (int)L;
  const int64_t Mp = pl.Mp, mm = Mp * Mp;
  hipLaunchKernelGGL((lu_prepare_kernel<T>), dim3((unsigned)(Mp / 32), (unsigned)(Mp / 32), L32), dim3(256), 0, s,
                     static_cast<const T*>(Lu_raw), M, Mp, b.LuT, (double*)nullptr, static_cast<T*>(Lu), b.lu_part);
  GPZ_LAUNCH_OK();
  for (int64_t ci = 0; ci < pl.nchunks; ++ci) {
    const int64_t n0 = ci * pl.nc;
    const int64_t nreal = (N - n0 < pl.nc) ? N - n0 : pl.nc;
    const int64_t ncp = pad_up(nreal);
    hipLaunchKernelGGL((w_transpose_kernel<T>), dim3((unsigned)(ncp / 32), (unsigned)(Mp / 32), L32), dim3(256), 0, s,
                       static_cast<const T*>(W), N, M, n0, Mp, ncp, b.Wc);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((w_rowstats_kernel<T>), dim3((unsigned)((ncp + 3) / 4), L32), dim3(256), 0, s,
                       static_cast<const T*>(W), static_cast<const T*>(mu), N, M, n0, ncp, b.ps1, b.pm1);
    GPZ_LAUNCH_OK();
    GemmParams<T> g2;
    g2.A = b.LuT; g2.lda = Mp; g2.sA0 = mm;
    g2.B = b.Wc; g2.ldb = ncp; g2.sB0 = Mp * ncp;
    g2.nb0 = L32; g2.mt = (int)pl.nblk; g2.nt = (int)(ncp / NB); g2.K = (int)Mp; g2.flags = GF_A_UPPER | GF_GROUP_COLS;
    const ProductSchedule sched = product_schedule<T>(true, g2.nt);
    g2.super_cols = sched.cols; g2.tiles_per_wg = sched.tpw; g2.ps_sq = b.ps2; g2.ncols = ncp;
    if (int rc = gemm_launch(g2, EPI_STATS, s)) return rc;
    FinalizeArgs<T> f;
    f.ps1 = b.ps1; f.pm1 = b.pm1; f.ps2 = b.ps2; f.sigma = static_cast<const T*>(sigma); f.y = nullptr;
    f.mean = static_cast<T*>(mean); f.scale = static_cast<T*>(scale); f.part = b.ll_part; f.N = N; f.n0 = n0;
    f.nc = ncp; f.nfb_total = pl.nfb_chunk * pl.nchunks; f.fb0 = ci * pl.nfb_chunk; f.mt = (int)pl.nblk; f.mt1 = 1;
    f.whitened = 1; f.clamp_min = 0.0; f.noise_sd = 1.0;
    hipLaunchKernelGGL((finalize_kernel<T>), dim3((unsigned)pl.nfb_chunk, L32), dim3(256), 0, s, f);
    GPZ_LAUNCH_OK();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Backward pass: gradients of a scalar loss w.r.t. mu and the raw Lu (always) and, on request, the kernel
// hyper-parameters and Z, given dLoss/dmean and dLoss/dscale of q(F).  With W = Linv Kzx, P = LuE^T W,
// gv2 = dLoss/dscale / scale (= 2 dLoss/dvar, zero where the variance clamp is active), D = diag(gv2):
//   d/d muE = W gm,   d/d LuE = tril(W (P D)^T) = tril(H LuE)   with   H = W D W^T  (symmetric, M x M)
// whitened: LuE = Lu, muE = mu;  un-whitened: LuE = Linv Lu, muE = Linv mu, so the results are pulled back through
// Linv^T.  Since round 5 the N-sized work of this mode (the big notebooks' training mode, frozen hyper-parameters:
// Slideseq_NSF_newest_version.ipynb:500-504) is ONE weighted symmetric product per chunk, H += W D W^T -- lower tiles
// only, L M^2 N flop -- followed by an M x M product; rounds 1-4 formed Pbar = P D (a second pass over W: L M^2 N flop
// more) and accumulated W Pbar^T.  Forward + backward = 3 triangular-product units instead of 4.
// With the kernel hyper-parameters / Z (`full`): Wbar = LuE Pbar - W diag(gv2 c) + muE gm^T, Kbar_x = Linv^T Wbar feeds
// kgrad.hip, and dLoss/dL = -tril(GL) with GL = sum_chunks Kbar_x W^T = Linv^T Q,
//   Q = sum Wbar W^T = LuE (Pbar W^T) - W diag(gv2 c) W^T + muE (W gm)^T = (Sw - I) H + Hd + muE v^T,
// Sw = LuE LuE^T, Hd = W diag(gv2 (1 - c)) W^T (c = 0 on the columns whose whitened prior term sits at its clamp, gp.py:287;
// zero everywhere else, so Hd is accumulated only in chunks that hold such a column), v = W gm -- M x M algebra in
// fp64 on the same H instead of a third N-sized accumulation; and Kbar_x itself is ONE dense product per chunk,
// [A1 W] diag(gv2) + a3 gm^T with A1 = Linv^T (Sw - I) (see the pass below): 5 units instead of 7 for the all-parameter step.
template <typename T>
__global__ void colscale_kernel(const T* __restrict__ g_scale, const T* __restrict__ scale, int64_t N, int64_t n0,
                                int64_t ncp, int whitened, double clamp_min, T* __restrict__ out,
                                const T* __restrict__ g_mean = nullptr, const T* __restrict__ ps1 = nullptr, int mt = 0,
                                const T* __restrict__ sigma = nullptr, T* __restrict__ out_c = nullptr,
                                T* __restrict__ out_gm = nullptr, T* __restrict__ out_d = nullptr,
                                int32_t* __restrict__ any_d = nullptr) {
  const int l = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncp) return;
  const int64_t n = n0 + c;
  T v = 0, vc = 0, gm = 0;
  if (n < N) {
    const T sc = scale[(int64_t)l * N + n];
    const bool clamped = !whitened && (double)sc * (double)sc <= clamp_min * (1.0 + 1e-6);
    if (sc > (T)0 && !clamped) v = g_scale[(int64_t)l * N + n] / sc;
    vc = v;
    if (out_c && whitened) {           // gp.py:287 clamp(Kxx - sum W^2, min=0): no gradient through a clamped term
      T s1 = 0;
      for (int i = 0; i < mt; ++i) s1 += ps1[((int64_t)l * mt + i) * ncp + c];
      const T sg = sigma[l];
      if (!(sg * sg - s1 > (T)0)) vc = 0;
    }
    if (g_mean) gm = g_mean[(int64_t)l * N + n];
  }
  out[(int64_t)l * ncp + c] = v;
  if (out_c) out_c[(int64_t)l * ncp + c] = vc;
  if (out_gm) out_gm[(int64_t)l * ncp + c] = gm;
  if (out_d) {                         // gv2 (1 - c): the weights of Hd; *any_d says whether the chunk has any
    out_d[(int64_t)l * ncp + c] = v - vc;
    if (v != vc) atomicOr(any_d, 1);
  }
}

// Ws[l][m][c] = W[l][m][c] * w[l][c]   (the weighted operand of H += W diag(w) W^T for the 128 x 128-tile kernel)
template <typename T>
__global__ __launch_bounds__(256) void scale_cols_kernel(const T* __restrict__ W, const T* __restrict__ w, int64_t Mp,
                                                        int64_t ncp, T* __restrict__ out) {
  const int l = blockIdx.z;
  const int64_t m = blockIdx.y, c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncp) return;
  const int64_t o = ((int64_t)l * Mp + m) * ncp + c;
  out[o] = W[o] * w[(int64_t)l * ncp + c];
}

// H's strict upper triangle <- its lower one (the accumulation computes the lower tiles; within a diagonal tile the two
// halves differ by rounding: only one operand carries the weights)
template <typename T>
__global__ __launch_bounds__(256) void mirror_lower_kernel(T* __restrict__ H, int64_t Mp) {
  __shared__ T tile[32][33];
  const int l = blockIdx.z;
  const int64_t bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  T* Hl = H + (int64_t)l * Mp * Mp;
  for (int rr = ty; rr < 32; rr += 8) tile[rr][tx] = Hl[(bi * 32 + rr) * Mp + bj * 32 + tx];
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8)              // element (bj*32 + rr, bi*32 + tx) = H[bi*32 + tx][bj*32 + rr]
    if (bj < bi || tx > rr) Hl[(bj * 32 + rr) * Mp + bi * 32 + tx] = tile[tx][rr];
}

// dst (fp64) = src, or dst += src
template <typename T>
__global__ void widen_kernel(const T* __restrict__ src, double* __restrict__ dst, int64_t n, int add) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (add ? dst[i] : 0.0) + (double)src[i];
}

// A[i][i] -= 1
__global__ void minus_identity_kernel(double* __restrict__ A, int64_t Mp) {
  const int l = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < Mp) A[(int64_t)l * Mp * Mp + i * (Mp + 1)] -= 1.0;
}

// me[l][i] = mu[l][i] in fp64, zero padded (whitened: muE = mu)
template <typename T>
__global__ void mu_widen_kernel(const T* __restrict__ mu, int64_t M, int64_t Mp, double* __restrict__ me) {
  const int l = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < Mp) me[(int64_t)l * Mp + i] = i < M ? (double)mu[(int64_t)l * M + i] : 0.0;
}

// acc[l] += sigma_l * sum_c csc[l][c]   (d var / d sigma through Kxx = sigma^2; one block per latent)
template <typename T>
__global__ __launch_bounds__(256) void sigma_direct_kernel(const T* __restrict__ csc, int64_t ncp,
                                                          const T* __restrict__ sigma, double* __restrict__ acc) {
  __shared__ double sh[8];
  const int l = blockIdx.x;
  double v = 0.0;
  for (int64_t c = threadIdx.x; c < ncp; c += 256) v += (double)csc[(int64_t)l * ncp + c];
  const double t = block_sum(v, sh);
  if (threadIdx.x == 0) acc[l] += (double)sigma[l] * t;
}

// dst = transpose(tril(src)) for (L,Mp,Mp) fp64 (src may hold garbage above the diagonal)
__global__ __launch_bounds__(256) void tril_transpose_kernel(const double* __restrict__ src, int64_t Mp,
                                                            double* __restrict__ dst) {
  __shared__ double tile[32][33];
  const int l = blockIdx.z;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t i = i0 + rr, j = j0 + tx;
    tile[rr][tx] = (j <= i) ? src[(int64_t)l * Mp * Mp + i * Mp + j] : 0.0;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) dst[(int64_t)l * Mp * Mp + (j0 + rr) * Mp + i0 + tx] = tile[tx][rr];
}

// Lbar = -tril(GL) (+ tril(upstream dLoss/dchol)) in fp64
template <typename T, typename TG>
__global__ void lbar_kernel(const TG* __restrict__ GL, int64_t Mp, int64_t M, const T* __restrict__ g_chol,
                            const double* __restrict__ E, double* __restrict__ Lbar,
                            const double* __restrict__ g_kl = nullptr, const double* __restrict__ Lc = nullptr) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  double v = 0.0;
  if (j <= i) {
    v = -(double)GL[(int64_t)l * Mp * Mp + i * Mp + j];
    if (E) v -= E[(int64_t)l * Mp * Mp + i * Mp + j];
    if (g_chol && i < M) v += (double)g_chol[(int64_t)l * M * M + i * M + j];
    if (g_kl && i == j && i < M) v += g_kl[l] / Lc[(int64_t)l * Mp * Mp + i * Mp + i];   // d(sum log diag L)/dL
  }
  Lbar[(int64_t)l * Mp * Mp + i * Mp + j] = v;
}

// dst = (double) tril(src)
template <typename T>
__global__ void tril_to_double_kernel(const T* __restrict__ src, int64_t Mp, double* __restrict__ dst) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  const int64_t o = (int64_t)l * Mp * Mp + i * Mp + j;
  dst[o] = (j <= i) ? (double)src[o] : 0.0;
}

// me[l][j] = (Linv mu)[j] in fp64, one wave per row (coalesced along the row), zero in the padding
template <typename T>
__global__ __launch_bounds__(256) void linv_mu_kernel(const double* __restrict__ Linv, const T* __restrict__ mu, int64_t Mp,
                                                     int64_t M, double* __restrict__ me) {
  const int l = blockIdx.y, lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= Mp) return;
  const double* Lb = Linv + (int64_t)l * Mp * Mp + j * Mp;
  double t = 0.0;
  if (j < M)
    for (int64_t k = lane; k <= j; k += 64) t = fma(Lb[k], (double)mu[(int64_t)l * M + k], t);
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if (lane == 0) me[(int64_t)l * Mp + j] = t;
}

// R[i][j] += u[i] * me[j]   (un-whitened: the muE = Linv mu dependence on the factor).  The version this replaces
// recomputed (Linv mu)[j] in every block with a thread walking row j: 46 ms at M = 3000, L = 20 -- a quarter of the
// all-parameter SVGP training step.
__global__ __launch_bounds__(256) void rank1_update_kernel(double* __restrict__ R, int64_t Mp, int64_t M,
                                                          const double* __restrict__ u, const double* __restrict__ me) {
  const int l = blockIdx.z;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const double mj = me[(int64_t)l * Mp + j];
  for (int64_t i = blockIdx.y; i < M; i += gridDim.y) R[(int64_t)l * Mp * Mp + i * Mp + j] += u[(int64_t)l * Mp + i] * mj;
}

// Phi: keep the lower triangle, halve the diagonal (Cholesky backward, Murray 2016)
__global__ void phi_kernel(double* __restrict__ A, int64_t Mp) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  double& v = A[(int64_t)l * Mp * Mp + i * Mp + j];
  if (j > i) v = 0.0;
  else if (j == i) v *= 0.5;
}

// dst = P + P^T, cast to T
template <typename T>
__global__ __launch_bounds__(256) void sym_cast_kernel(const double* __restrict__ P, int64_t Mp, T* __restrict__ dst) {
  __shared__ double tile[32][33];
  const int l = blockIdx.z;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) tile[rr][tx] = P[(int64_t)l * Mp * Mp + (j0 + rr) * Mp + i0 + tx];  // P[j][i]
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t i = i0 + rr, j = j0 + tx;
    dst[(int64_t)l * Mp * Mp + i * Mp + j] = (T)(P[(int64_t)l * Mp * Mp + i * Mp + j] + tile[tx][rr]);
  }
}

// grad_Z[m][k] = sum_l acc[l][m][k];  grad_theta[l][0..2] = sum_m acc[l][m][4..6] (+ direct sigma term)
__global__ __launch_bounds__(256) void kgrad_finish_kernel(const double* __restrict__ acc, int L, int64_t Mp, int64_t M,
                                                          int d, const double* __restrict__ sig_direct,
                                                          double* __restrict__ grad_Z, double* __restrict__ grad_theta) {
  __shared__ double sh[8];
  if (blockIdx.y == 0) {            // Z rows
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < M && grad_Z)
      for (int k = 0; k < 4; ++k) {
        double t = 0.0;
        if (k < d)
          for (int l = 0; l < L; ++l) t += acc[((int64_t)l * Mp + m) * 8 + k];
        grad_Z[m * 4 + k] = t;
      }
  } else if ((int)blockIdx.x < L && grad_theta) {   // one block per latent
    const int l = blockIdx.x;
    for (int q = 0; q < 3; ++q) {
      double v = 0.0;
      for (int64_t m = threadIdx.x; m < M; m += 256) v += acc[((int64_t)l * Mp + m) * 8 + 4 + q];
      const double t = block_sum(v, sh);
      if (threadIdx.x == 0) grad_theta[l * 4 + q] = t + (q == 0 ? sig_direct[l] : 0.0);
    }
    if (threadIdx.x == 0) grad_theta[l * 4 + 3] = 0.0;
  }
}

// part[l][ci][m] = sum_c Wt[l][m][c] * g_mean[l][n0 + c]   (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ Wt, int64_t Mp, int64_t ncp,
                                                    const T* __restrict__ g_mean, int64_t N, int64_t n0,
                                                    double* __restrict__ part, int64_t nchunks, int64_t ci) {
  const int l = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nreal = (N - n0 < ncp) ? N - n0 : ncp;
  const T* row = Wt + ((int64_t)l * Mp + m) * ncp;
  const T* g = g_mean + (int64_t)l * N + n0;
  double acc = 0.0;
  for (int64_t c = lane; c < nreal; c += 64) acc += (double)row[c] * (double)g[c];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) part[((int64_t)l * nchunks + ci) * Mp + m] = acc;
}

// v[l][m] = sum_ci part[l][ci][m]
__global__ void chunk_sum_kernel(const double* __restrict__ part, int64_t nchunks, int64_t Mp, double* __restrict__ v) {
  const int l = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= Mp) return;
  double t = 0.0;
  for (int64_t ci = 0; ci < nchunks; ++ci) t += part[((int64_t)l * nchunks + ci) * Mp + m];
  v[(int64_t)l * Mp + m] = t;
}

// out[l][a] = sum_{i >= a} Linv[l][i][a] * v[l][i]   (Linv^T v, Linv lower triangular) or v itself
template <typename T>
__global__ void mu_grad_kernel(const double* __restrict__ v, const double* __restrict__ Linv, int64_t Mp, int64_t M,
                               T* __restrict__ out, const double* __restrict__ g_kl_w = nullptr,
                               const T* __restrict__ mu = nullptr) {
  const int l = blockIdx.y;
  const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a >= M) return;
  double t;
  if (!Linv) {
    t = v[(int64_t)l * Mp + a];
    if (g_kl_w) t += g_kl_w[l] * (double)mu[(int64_t)l * M + a];     // whitened KL: d/dmu of |mu|^2 / 2
  } else {
    t = 0.0;
    const double* Lb = Linv + (int64_t)l * Mp * Mp;
    for (int64_t i = a; i < M; ++i) t += Lb[i * Mp + a] * v[(int64_t)l * Mp + i];
  }
  out[(int64_t)l * M + a] = (T)t;
}

// out = Linv^T v with 32 columns x 8 row segments per block (a thread per column alone walks up to M rows serially:
// 0.73 ms at M = 3000, L = 20)
template <typename T>
__global__ __launch_bounds__(256) void mu_grad_linv_kernel(const double* __restrict__ v, const double* __restrict__ Linv,
                                                          int64_t Mp, int64_t M, T* __restrict__ out) {
  __shared__ double sh[8][33];
  const int l = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t a = (int64_t)blockIdx.x * 32 + tx;
  const double* Lb = Linv + (int64_t)l * Mp * Mp;
  double t = 0.0;
  if (a < M)
    for (int64_t i = a + ty; i < M; i += 8) t = fma(Lb[i * Mp + a], v[(int64_t)l * Mp + i], t);
  sh[ty][tx] = t;
  __syncthreads();
  if (ty == 0 && a < M) {
    double r = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) r += sh[q][tx];
    out[(int64_t)l * M + a] = (T)r;
  }
}

// out[l][a] = sum_{i >= a} Linv[l][i][a] * v[l][i]  (Linv^T v) for a < M, zero in the padding; out (L, Mp)
template <typename T>
__global__ __launch_bounds__(256) void linvT_vec_kernel(const double* __restrict__ v, const double* __restrict__ Linv,
                                                       int64_t Mp, int64_t M, T* __restrict__ out) {
  __shared__ double sh[8][33];
  const int l = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t a = (int64_t)blockIdx.x * 32 + tx;
  const double* Lb = Linv + (int64_t)l * Mp * Mp;
  double t = 0.0;
  if (a < M)
    for (int64_t i = a + ty; i < M; i += 8) t = fma(Lb[i * Mp + a], v[(int64_t)l * Mp + i], t);
  sh[ty][tx] = t;
  __syncthreads();
  if (ty == 0 && a < Mp) {
    double r = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) r += sh[q][tx];
    out[(int64_t)l * Mp + a] = (T)r;
  }
}

// zero the strict upper triangle of (L,Mp,Mp)
template <typename T>
__global__ void tril_kernel(T* __restrict__ G, int64_t Mp) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < Mp && j > i) G[(int64_t)l * Mp * Mp + i * Mp + j] = (T)0;
}

// chain rule of the constraint Lu = tril(raw, -1) + diag(exp(diag raw))
template <typename T>
__global__ void lu_grad_kernel(const T* __restrict__ G, int64_t Mp, int64_t M, const T* __restrict__ raw,
                               T* __restrict__ out, const double* __restrict__ g_kl = nullptr, int whitened_kl = 0) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  double g = (double)G[(int64_t)l * Mp * Mp + i * Mp + j];
  const double gk = g_kl ? g_kl[l] : 0.0;
  const double x = (double)raw[(int64_t)l * M * M + i * M + j];
  T v = 0;
  if (j < i) {
    if (whitened_kl) g += gk * x;                 // whitened KL: d/dLu of |Lu|_F^2 / 2 (un-whitened: already in G)
    v = (T)g;
  } else if (j == i) {   // Lu_ii = exp(raw_ii); the KL's -log Lu_ii contributes -g_kl to the raw diagonal
    const double e = exp(x);
    if (whitened_kl) g += gk * e;
    v = (T)(g * e - gk);
  }
  out[(int64_t)l * M * M + i * M + j] = v;
}

// KL(qU || pU) folded into the un-whitened gradients: dKL/dLuE = LuE (lower), dKL/dmuE = muE
template <typename T>
__global__ void kl_add_kernel(T* __restrict__ G, int64_t Mp, const double* __restrict__ LuW,
                              double* __restrict__ mu_sum, const T* __restrict__ muE, const double* __restrict__ g_kl) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  const double gk = g_kl[l];
  if (j <= i) {
    const int64_t o = (int64_t)l * Mp * Mp + i * Mp + j;
    G[o] = (T)((double)G[o] + gk * LuW[o]);
  }
  if (j == 0) mu_sum[(int64_t)l * Mp + i] += gk * (double)muE[(int64_t)l * Mp + i];
}

template <typename T>
struct BwdBuffers {
  T *Pc, *H, *G, *G2, *LinvT, *cs; double *mu_part, *mu_sum;
  float* nt_part;                            // fp32, few tiles: pieces of the A B^T accumulations (gemmw.hip, wide_nt_pieces)
  float* nt_vpart;                           // fp32: per-piece W gm from the same launch
  T* gmc;                                    // (L, nc) dLoss/dmean of the chunk, zero padded
  // kernel / Z gradients only
  T *LuN, *Hd, *GL, *SwI, *csc, *csd, *PS, *A1, *a3; double *D1, *D2, *D3, *D4, *me, *kacc, *sig_direct;
  int32_t* any_d;                            // per chunk: does it hold a column at the whitened clamp (weights of Hd)?
  size_t bytes;
};

template <typename T>
static BwdBuffers<T> carve_bwd(const Plan& pl, bool whitened, bool full, void* ws, size_t offset) {
  BwdBuffers<T> b;
  Carver c(ws);
  c.off = offset;
  const int64_t mm = pl.L * pl.Mp * pl.Mp;
  // Pbar / Kbar_x chunk (full), and the weighted operand of H for the 128 x 128-tile kernel (the wide kernel applies the
  // weights itself)
  b.Pc = c.take<T>(pl.L * pl.Mp * pl.nc);
  b.H = c.take<T>(mm);
  b.G = c.take<T>(mm);
  b.G2 = whitened ? nullptr : c.take<T>(mm);
  b.LinvT = (whitened && !full) ? nullptr : c.take<T>(mm);
  b.cs = c.take<T>(pl.L * pl.nc);
  b.mu_part = c.take<double>(pl.L * pl.nchunks * pl.Mp);
  b.mu_sum = c.take<double>(pl.L * pl.Mp);
  const size_t ntf = sizeof(T) == 4 && wide_nt_supported(pl.Mp, pl.nc) ? wide_nt_scratch_floats(pl.Mp, pl.nc, (int)pl.L) : 0;
  b.nt_part = ntf ? c.take<float>((int64_t)ntf) : nullptr;
  const bool ntw = sizeof(T) == 4 && wide_nt_supported(pl.Mp, pl.nc);
  b.nt_vpart = ntw ? c.take<float>((int64_t)wide_nt_vpart_floats(pl.Mp, pl.nc, (int)pl.L)) : nullptr;
  b.gmc = c.take<T>(pl.L * pl.nc);
  b.LuN = b.Hd = b.GL = b.SwI = b.csc = b.csd = b.PS = b.A1 = b.a3 = nullptr;
  b.D1 = b.D2 = b.D3 = b.D4 = b.me = b.kacc = b.sig_direct = nullptr;
  b.any_d = nullptr;
  if (full) {
    b.LuN = c.take<T>(mm);
    b.Hd = whitened ? c.take<T>(mm) : nullptr;
    b.PS = c.take<T>(mm);
    b.GL = c.take<T>(mm);
    b.SwI = c.take<T>(mm);
    b.A1 = c.take<T>(mm);
    b.a3 = c.take<T>(pl.L * pl.Mp);
    b.csc = c.take<T>(pl.L * pl.nc);
    b.csd = whitened ? c.take<T>(pl.L * pl.nc) : nullptr;
    b.D1 = c.take<double>(mm);
    b.D2 = c.take<double>(mm);
    b.D3 = c.take<double>(mm);
    b.D4 = whitened ? nullptr : c.take<double>(mm);
    b.me = c.take<double>(pl.L * pl.Mp);
    b.kacc = c.take<double>(pl.L * pl.Mp * 8);
    b.sig_direct = c.take<double>(pl.L);
    b.any_d = c.take<int32_t>(pl.nchunks + 1);
  }
  b.bytes = c.used();
  return b;
}

// T-precision helpers of the M x M tail
// R[i][j] += Hd[i][j] + u[i] v[j]   (lower triangle incl. the diagonal tiles; Hd may be null)
template <typename TR, typename T>
__global__ __launch_bounds__(256) void q_finish_kernel(TR* __restrict__ R, const T* __restrict__ Hd, int64_t Mp, int64_t M,
                                                      const double* __restrict__ u, const double* __restrict__ v) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp || (j >> 7) > (i >> 7)) return;
  const int64_t o = (int64_t)l * Mp * Mp + i * Mp + j;
  double r = (double)R[o];
  if (Hd) r += (double)Hd[o];
  if (i < M && j < M) r += u[(int64_t)l * Mp + i] * v[(int64_t)l * Mp + j];
  R[o] = (TR)r;
}
template <typename T>
__global__ void minus_identity_t_kernel(T* __restrict__ A, int64_t Mp) {
  const int l = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < Mp) A[(int64_t)l * Mp * Mp + i * (Mp + 1)] -= (T)1;
}

// Which form the N-sized work takes (see the comment at the head of this section).  The algebra form trades N-sized
// products for M x M ones: per pass, in units of L M^2 N and of 2 L M^3 flop,
//   mu / Lu only      classic 2 (Pbar, W Pbar^T)            algebra 1 (H) + 1/3 (H LuE)                 -> algebra iff N >= 0.75 M
//   all parameters    classic 5 (+ Wbar, Kbar_x, Kbar_x W^T) algebra 3 (H, dense Kbar_x) + 2 (Sw, A1, Q, GL, H LuE) -> iff N >= 2.2 M
//                     (un-whitened fp32 models run those four M x M products in fp64 -- their sigma / lengthscale gradients
//                     are sums with heavy cancellation through Linv: 5e-4 of the fp64 value in fp32 against 1e-4 for the
//                     classic form, measured at N = 3000, M = 1024 -- at half the rate: iff N >= 4.4 M)
// (the notebooks' benchmark model trains Z and the lengthscale on N = 1037 spots with M up to 1000 inducing points:
// classic; the Slide-seq minibatches, N_b = 7000, M = 3000, frozen hyper-parameters: algebra).
static bool backward_algebra(int64_t N, int64_t Mp, bool full, int flags, bool f64_algebra) {
  static const int env = [] { const char* e = getenv("GPZ_SVGP_BACKWARD"); return !e ? 0 : !strcmp(e, "algebra") ? 1 : !strcmp(e, "classic") ? 2 : 0; }();
  if (flags & GPZ_SVGP_BACKWARD_ALGEBRA) return true;
  if (flags & GPZ_SVGP_BACKWARD_CLASSIC) return false;
  if (env) return env == 1;
  return full ? 10 * N >= (f64_algebra ? 44 : 22) * Mp : 4 * N >= 3 * Mp;
}

template <typename T>
static int svgp_backward_t(const gpz_svgp_problem* p, const gpz_svgp_grads* g, int64_t chunk, void* ws, size_t ws_bytes,
                           hipStream_t s) {
  const Plan pl = make_plan(p, chunk);
  const bool wh = p->whitened != 0;
  const bool full = g->grad_theta != nullptr || g->grad_Z != nullptr;   // kernel hyper-parameter / Z gradients
  Buffers<T> b = carve<T>(pl, wh, ws);
  BwdBuffers<T> w = carve_bwd<T>(pl, wh, full, ws, b.bytes);
  GPZ_REQUIRE(ws_bytes >= w.bytes, "gpz_svgp_backward: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
  const int64_t L = pl.L, M = pl.M, Mp = pl.Mp, N = pl.N, mm = Mp * Mp;
  const int L32 = (int)L;
  const bool alg64 = sizeof(T) == 4 && !wh;            // fp32, un-whitened: the M x M products of the algebra form in fp64
  const bool alg = backward_algebra(N, Mp, full, p->flags, alg64);
  gpz_svgp_problem q = *p;          // forward-only outputs are not produced again
  q.chol = nullptr; q.Lu = nullptr;
  if (int rc = prepare_t<T>(&q, pl, b, s)) return rc;
  {
    Zeroer z;
    if (alg) z.add(w.H, sizeof(T) * L * mm);
    z.add(w.G, sizeof(T) * L * mm);          // (its tiles above the diagonal are never written)
    if (!wh) z.add(w.G2, sizeof(T) * L * mm);
    if (full) {
      if (alg && wh) {
        z.add(w.Hd, sizeof(T) * L * mm);
        z.add(w.any_d, sizeof(int32_t) * (pl.nchunks + 1));
      }
      if (!alg) z.add(w.GL, sizeof(T) * L * mm);
      z.add(w.kacc, sizeof(double) * L * Mp * 8);
      z.add(w.sig_direct, sizeof(double) * L);
    }
    if (int rc = z.run(s)) return rc;
  }
  const dim3 g32((unsigned)(Mp / 32), (unsigned)(Mp / 32), L32);
  const dim3 gm((unsigned)((Mp + 255) / 256), (unsigned)Mp, L32);
  auto tgemm = [&](const T* A, const T* B, T* C, int flags) -> int {
    GemmParams<T> d;
    d.A = A; d.lda = Mp; d.sA0 = mm; d.B = B; d.ldb = Mp; d.sB0 = mm; d.C = C; d.ldc = Mp; d.sC0 = mm;
    d.nb0 = L32; d.mt = d.nt = (int)pl.nblk; d.K = (int)Mp; d.flags = flags;
    return gemm_launch(d, EPI_STORE, s);
  };
  if (full) {
    // Lu (lower, not transposed) in GEMM precision and Linv^T
    if (wh) {
      hipLaunchKernelGGL((lu_prepare_kernel<T>), g32, dim3(256), 0, s, static_cast<const T*>(p->Lu_raw), M, Mp,
                         (T*)nullptr, (double*)nullptr, (T*)nullptr, b.lu_part, w.LuN);
    } else {   // LuE = Linv Lu (lower), already formed in fp64 by the preparation step
      hipLaunchKernelGGL((cast_kernel<T>), dim3(2048), dim3(256), 0, s, b.LuW, w.LuN, L * mm);
    }
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((transpose_cast_kernel<T>), g32, dim3(256), 0, s, b.Linv, Mp, w.LinvT, (double*)w.D1);
    GPZ_LAUNCH_OK();
    // muE in fp64 (whitened: mu itself): the rank-1 terms of Kbar_x / Q and, un-whitened, of E below
    if (wh)
      hipLaunchKernelGGL((mu_widen_kernel<T>), dim3((unsigned)((Mp + 255) / 256), L32), dim3(256), 0, s,
                         static_cast<const T*>(p->mu), M, Mp, w.me);
    else
      hipLaunchKernelGGL((linv_mu_kernel<T>), dim3((unsigned)(Mp / 4), L32), dim3(256), 0, s, b.Linv,
                         static_cast<const T*>(p->mu), Mp, M, w.me);
    GPZ_LAUNCH_OK();
    if (alg) {
      // What Kbar_x = Linv^T Wbar needs once per pass.  Wbar = LuE Pbar - W diag(gv2 c) + muE gm^T with Pbar = LuE^T W diag(gv2)
      // is (Sw - I) W D + W (D - Dc) + muE gm^T (Sw = LuE LuE^T, D = diag(gv2), Dc = diag(gv2 c)), so
      //   Kbar_x = [A1 W] D + [Linv^T W] (D - Dc) + a3 gm^T,   A1 = Linv^T (Sw - I) (dense M x M),  a3 = Linv^T muE:
      // ONE dense product per chunk (2 L M^2 N flop) where the classic form runs three triangular ones (Pbar, Wbar, Kbar_x);
      // D - Dc is zero except on columns at the whitened clamp, whose term runs only in chunks that hold one.  The M x M
      // products run in the problem's precision (the classic form accumulates the same quantities in it).
      if (alg64) {
        // Sw - I (kept in D2 for Q) and A1 in fp64, then cast
        GemmParams<double> d;
        auto dg = [&](const double* A, const double* B, double* C, int flags) -> int {
          d.A = A; d.lda = Mp; d.sA0 = mm; d.B = B; d.ldb = Mp; d.sB0 = mm; d.C = C; d.ldc = Mp; d.sC0 = mm;
          d.nb0 = L32; d.mt = d.nt = (int)pl.nblk; d.K = (int)Mp; d.flags = flags;
          return gemm_launch(d, EPI_STORE, s);
        };
        const double* LuE64 = b.LuW;
        if (wh) {
          hipLaunchKernelGGL((widen_kernel<T>), dim3(2048), dim3(256), 0, s, (const T*)w.LuN, w.D3, L * mm, 0);
          GPZ_LAUNCH_OK();
          LuE64 = w.D3;
        }
        if (int rc = dg(LuE64, LuE64, w.D2, GF_A_LOWER | GF_B_UPPER | GF_B_TRANS)) return rc;
        hipLaunchKernelGGL(minus_identity_kernel, dim3((unsigned)((Mp + 255) / 256), L32), dim3(256), 0, s, w.D2, Mp);
        GPZ_LAUNCH_OK();
        hipLaunchKernelGGL(tril_transpose_kernel, g32, dim3(256), 0, s, b.Linv, Mp, w.D1);
        GPZ_LAUNCH_OK();
        if (int rc = dg(w.D1, w.D2, w.D3, GF_A_UPPER)) return rc;
        hipLaunchKernelGGL((cast_kernel<T>), dim3(2048), dim3(256), 0, s, w.D3, w.A1, L * mm);
        GPZ_LAUNCH_OK();
      } else {
      if (int rc = tgemm(w.LuN, w.LuN, w.SwI, GF_A_LOWER | GF_B_UPPER | GF_B_TRANS)) return rc;            // Sw
      hipLaunchKernelGGL((minus_identity_t_kernel<T>), dim3((unsigned)((Mp + 255) / 256), L32), dim3(256), 0, s, w.SwI, Mp);
      GPZ_LAUNCH_OK();                                                                                     // Sw - I (kept for Q)
      if (int rc = tgemm(w.LinvT, w.SwI, w.A1, GF_A_UPPER)) return rc;                                     // A1
      }
      hipLaunchKernelGGL((linvT_vec_kernel<T>), dim3((unsigned)(Mp / 32), L32), dim3(256), 0, s, w.me, b.Linv, Mp, M, w.a3);
      GPZ_LAUNCH_OK();
    }
  }
  const int64_t esz = sizeof(T);
  const bool have_wt = p->wt_cache != nullptr && p->wt_cache_valid != 0;
  const WtCache<T> wtc = wt_cache_of<T>(pl, p->wt_cache);
  for (int64_t ci = 0; ci < pl.nchunks; ++ci) {
    const int64_t n0 = ci * pl.nc;
    const int64_t nreal = (N - n0 < pl.nc) ? N - n0 : pl.nc;
    const int64_t ncp = pad_up(nreal);
    const int nt = (int)(ncp / NB);
    const ProductSchedule sched = product_schedule<T>(true, nt), sched_plain = product_schedule<T>(false, nt);
    const bool wide = !(p->flags & GPZ_SVGP_NARROW_TILES) && pl.f32 && wide_product_supported(Mp, ncp);   // gemmw.hip
    const void* Xc = static_cast<const char*>(p->X) + n0 * p->d * esz;
    T* Wc = b.Wc;
    T* ps1 = b.ps1;
    if (have_wt) {   // the forward pass kept Wt and its column sums: nothing to rebuild
      Wc = wtc.wt(ci);
      ps1 = wtc.ps1(ci);
    } else {
      if (int rc = kfill_padded(&p->k, p->Z, M, Mp, Xc, nreal, ncp, p->d, p->gZ, p->gX ? p->gX + n0 : nullptr, b.Kc, ncp,
                                Mp * ncp, 0.0, 0, pl.f32 ? GPZ_F32 : GPZ_F64, s))
        return rc;
      bool done = false;
      if constexpr (sizeof(T) == 4) {
        if (wide) {      // W = Linv * Kzx on the wide tile (its column statistics come for free)
          WideArgs wa = {};
          wa.A = b.LinvG; wa.B = b.Kc; wa.Mp = Mp; wa.ncp = ncp; wa.L = L32; wa.upper = 0; wa.epilogue = WIDE_STORE_STATS;
          wa.C = b.Wc; wa.mu = b.muE; wa.ps_sq = b.ps1; wa.ps_mu = b.pm1;
          if (int rc = wide_product_launch(wa, s)) return rc;
          done = true;
        }
      }
      if (!done) {
        GemmParams<T> g1;  // W = Linv * Kzx (column sums of W^2 only when the clamp mask is needed)
        g1.A = b.LinvG; g1.lda = Mp; g1.sA0 = mm;
        g1.B = b.Kc; g1.ldb = ncp; g1.sB0 = Mp * ncp;
        g1.C = b.Wc; g1.ldc = ncp; g1.sC0 = Mp * ncp;
        g1.nb0 = L32; g1.mt = (int)pl.nblk; g1.nt = nt; g1.K = (int)Mp; g1.flags = GF_A_LOWER | GF_GROUP_COLS;
        const ProductSchedule s1 = full ? sched : sched_plain;    // the plain-store epilogue has no multi-tile variant
        g1.super_cols = s1.cols; g1.tiles_per_wg = s1.tpw; g1.mu = b.muE; g1.sMu = Mp; g1.ps_sq = b.ps1; g1.ps_mu = b.pm1; g1.ncols = ncp;
        if (int rc = gemm_launch(g1, full ? EPI_STORE_STATS : EPI_STORE, s)) return rc;
      }
    }
    const bool with_hd = alg && full && wh;   // columns at the whitened clamp: their weights gv2 (1 - c) and the chunk's flag
    hipLaunchKernelGGL((colscale_kernel<T>), dim3((unsigned)((ncp + 255) / 256), L32), dim3(256), 0, s,
                       static_cast<const T*>(g->g_scale), static_cast<const T*>(g->scale), N, n0, ncp, (int)wh,
                       p->var_clamp_min, w.cs, static_cast<const T*>(g->g_mean), (const T*)ps1, (int)pl.nblk,
                       static_cast<const T*>(p->k.sigma), w.csc, w.gmc, with_hd ? w.csd : (T*)nullptr,
                       with_hd ? w.any_d + ci : (int32_t*)nullptr);
    GPZ_LAUNCH_OK();
    const bool wide_nt = sizeof(T) == 4 && wide && wide_nt_supported(Mp, ncp);
    // C += A B^T over the chunk (lower tiles) on the 128 x 128-tile kernel
    auto nt_generic = [&](const T* A, const T* B, T* C) -> int {
      GemmParams<T> g3;
      g3.A = A; g3.lda = ncp; g3.sA0 = Mp * ncp;
      g3.B = B; g3.ldb = ncp; g3.sB0 = Mp * ncp;
      g3.C = C; g3.ldc = Mp; g3.sC0 = mm;
      g3.nb0 = L32; g3.mt = g3.nt = (int)pl.nblk; g3.K = (int)ncp; g3.flags = GF_B_TRANS | GF_TILES_LOWER;
      g3.alpha = 1; g3.beta = 1;
      return gemm_launch(g3, EPI_STORE, s);
    };
    bool v_done = false;             // dLoss/dmuE's chunk share W gm formed by the accumulation launch itself
    if (alg) {
      // C += W diag(wts) W^T (lower tiles): the wide kernel weights its B fragments itself (and its diagonal tiles form
      // v = W gm); the 128 x 128-tile kernel gets a weighted copy of W (in the Pbar buffer, free at this point)
      auto syrk = [&](const T* wts, T* C, const int32_t* gate) -> int {
        if (wide_nt) {
          if constexpr (sizeof(T) == 4)
            return wide_nt_launch(Wc, Wc, C, Mp, ncp, L32, s, w.nt_part, wts, gate, gate ? nullptr : w.gmc,
                                  gate ? nullptr : w.nt_vpart, gate ? nullptr : w.mu_part + ci * Mp, pl.nchunks * Mp);
        }
        hipLaunchKernelGGL((scale_cols_kernel<T>), dim3((unsigned)((ncp + 255) / 256), (unsigned)Mp, L32), dim3(256), 0, s,
                           (const T*)Wc, wts, Mp, ncp, w.Pc);
        GPZ_LAUNCH_OK();
        return nt_generic(w.Pc, Wc, C);
      };
      if (int rc = syrk(w.cs, w.H, nullptr)) return rc;                      // H  += W diag(gv2) W^T
      if (with_hd)
        if (int rc = syrk(w.csd, w.Hd, w.any_d + ci)) return rc;             // Hd += W diag(gv2 (1 - c)) W^T, if any
      v_done = wide_nt;
    } else {
      // classic form: Pbar = (LuE^T W) diag(gv2), then G += W Pbar^T
      bool done = false;
      if constexpr (sizeof(T) == 4) {
        if (wide) {
          WideArgs wa = {};
          wa.A = b.LuT; wa.B = Wc; wa.Mp = Mp; wa.ncp = ncp; wa.L = L32; wa.upper = 1; wa.epilogue = WIDE_STORE_COLSCALE;
          wa.C = w.Pc; wa.colscale = w.cs;
          if (int rc = wide_product_launch(wa, s)) return rc;
          done = true;
        }
      }
      if (!done) {
        GemmParams<T> g2;
        g2.A = b.LuT; g2.lda = Mp; g2.sA0 = mm;
        g2.B = Wc; g2.ldb = ncp; g2.sB0 = Mp * ncp;
        g2.C = w.Pc; g2.ldc = ncp; g2.sC0 = Mp * ncp;
        g2.nb0 = L32; g2.mt = (int)pl.nblk; g2.nt = nt; g2.K = (int)Mp; g2.flags = GF_A_UPPER | GF_GROUP_COLS;
        g2.super_cols = sched_plain.cols; g2.colscale = w.cs; g2.sCs = ncp; g2.ncols = ncp;
        if (int rc = gemm_launch(g2, EPI_STORE_COLSCALE, s)) return rc;
      }
      if (wide_nt) {
        if constexpr (sizeof(T) == 4) { if (int rc = wide_nt_launch(Wc, w.Pc, w.G, Mp, ncp, L32, s, w.nt_part)) return rc; }
      } else if (int rc = nt_generic(Wc, w.Pc, w.G)) return rc;
    }
    if (!v_done) {
      hipLaunchKernelGGL((rowdot_kernel<T>), dim3((unsigned)(Mp / 4), L32), dim3(256), 0, s, Wc, Mp, ncp,
                         static_cast<const T*>(g->g_mean), N, n0, w.mu_part, pl.nchunks, ci);
      GPZ_LAUNCH_OK();
    }
    if (full) {
      bool done = false;
      if constexpr (sizeof(T) == 4) {
        if (wide && alg) {   // Kbar_x = [A1 W] diag(gv2) + a3 gm^T in one dense product       (into the Pbar buffer)
          WideArgs wk = {};
          wk.A = w.A1; wk.B = Wc; wk.Mp = Mp; wk.ncp = ncp; wk.L = L32; wk.upper = 2; wk.epilogue = WIDE_KBAR;
          wk.C = w.Pc; wk.colscale = w.cs; wk.colvec = w.gmc; wk.rowvec = w.a3;
          if (int rc = wide_product_launch(wk, s)) return rc;
          if (with_hd) {     // ... += [Linv^T W] diag(gv2 (1 - c)): only in a chunk with a column at the whitened clamp
            WideArgs wc = {};
            wc.A = w.LinvT; wc.B = Wc; wc.Mp = Mp; wc.ncp = ncp; wc.L = L32; wc.upper = 1; wc.epilogue = WIDE_ADD_COLSCALE;
            wc.C = w.Pc; wc.colscale = w.csd; wc.gate = w.any_d + ci;
            if (int rc = wide_product_launch(wc, s)) return rc;
          }
          done = true;
        }
        if (wide && !alg) {  // Wbar = Lu Pbar - W diag(gv2 c) + mu gm^T (into the Kzx buffer), Kbar_x = Linv^T Wbar (Pbar buffer)
          WideArgs wa = {};
          wa.A = w.LuN; wa.B = w.Pc; wa.Mp = Mp; wa.ncp = ncp; wa.L = L32; wa.upper = 0; wa.epilogue = WIDE_WBAR;
          wa.C = b.Kc; wa.colscale = w.csc; wa.colvec = w.gmc; wa.rowvec = b.muE; wa.aux = Wc;
          if (int rc = wide_product_launch(wa, s)) return rc;
          WideArgs wb = {};
          wb.A = w.LinvT; wb.B = b.Kc; wb.Mp = Mp; wb.ncp = ncp; wb.L = L32; wb.upper = 1; wb.epilogue = WIDE_STORE;
          wb.C = w.Pc;
          if (int rc = wide_product_launch(wb, s)) return rc;
          done = true;
        }
      }
      if (!done) {       // the 128 x 128-tile kernels: (Pbar), Wbar, Kbar_x as triangular products
        if (alg) {       // (the algebra form has no Pbar yet)
          GemmParams<T> g2;
          g2.A = b.LuT; g2.lda = Mp; g2.sA0 = mm;
          g2.B = Wc; g2.ldb = ncp; g2.sB0 = Mp * ncp;
          g2.C = w.Pc; g2.ldc = ncp; g2.sC0 = Mp * ncp;
          g2.nb0 = L32; g2.mt = (int)pl.nblk; g2.nt = nt; g2.K = (int)Mp; g2.flags = GF_A_UPPER | GF_GROUP_COLS;
          g2.super_cols = sched_plain.cols; g2.colscale = w.cs; g2.sCs = ncp; g2.ncols = ncp;
          if (int rc = gemm_launch(g2, EPI_STORE_COLSCALE, s)) return rc;
        }
        GemmParams<T> g4;  // Wbar = Lu Pbar - W diag(gv2 c) + mu gm^T            (into the Kzx buffer, no longer needed)
        g4.A = w.LuN; g4.lda = Mp; g4.sA0 = mm;
        g4.B = w.Pc; g4.ldb = ncp; g4.sB0 = Mp * ncp;
        g4.C = b.Kc; g4.ldc = ncp; g4.sC0 = Mp * ncp;
        g4.nb0 = L32; g4.mt = (int)pl.nblk; g4.nt = nt; g4.K = (int)Mp; g4.flags = GF_A_LOWER | GF_GROUP_COLS;
        g4.super_cols = sched_plain.cols; g4.colscale = w.csc; g4.colvec = w.gmc; g4.sCs = ncp; g4.rowvec = b.muE; g4.sRv = Mp;
        g4.aux = Wc; g4.ncols = ncp;
        if (int rc = gemm_launch(g4, EPI_WBAR, s)) return rc;
        GemmParams<T> g5;  // Kbar_x = Linv^T Wbar                                   (into the Pbar buffer)
        g5.A = w.LinvT; g5.lda = Mp; g5.sA0 = mm;
        g5.B = b.Kc; g5.ldb = ncp; g5.sB0 = Mp * ncp;
        g5.C = w.Pc; g5.ldc = ncp; g5.sC0 = Mp * ncp;
        g5.nb0 = L32; g5.mt = (int)pl.nblk; g5.nt = nt; g5.K = (int)Mp; g5.flags = GF_A_UPPER | GF_GROUP_COLS;
        g5.super_cols = sched_plain.cols;
        if (int rc = gemm_launch(g5, EPI_STORE, s)) return rc;
      }
      if (!alg) {        // classic form: GL += Kbar_x W^T (lower tiles); the algebra form gets GL from H below
        if (wide_nt) {
          if constexpr (sizeof(T) == 4) { if (int rc = wide_nt_launch(w.Pc, Wc, w.GL, Mp, ncp, L32, s, w.nt_part)) return rc; }
        } else if (int rc = nt_generic(w.Pc, Wc, w.GL)) return rc;
      }
      // kernel hyper-parameter and Z gradients from Kbar_x
      KgradArgs ka;
      ka.Kbar = w.Pc; ka.ld = ncp; ka.stride = Mp * ncp; ka.Z = p->Z; ka.X = Xc;
      ka.gZ = p->gZ; ka.gX = p->gX ? p->gX + n0 : nullptr;
      ka.sigma = p->k.sigma; ka.ell = p->k.lengthscale; ka.ga = p->k.group_a; ka.gr2 = p->k.group_r2;
      ka.gpow = p->k.group_pow; ka.scalar_scale = 1.0; ka.M = M; ka.ncols = nreal; ka.Mp = Mp; ka.d = p->d;
      ka.G = p->k.n_groups; ka.acc = w.kacc;
      if (int rc = kgrad_launch(p->dtype, p->k.kind, ka, L32, s)) return rc;
      hipLaunchKernelGGL((sigma_direct_kernel<T>), dim3(L32), dim3(256), 0, s, w.csc, ncp,
                         static_cast<const T*>(p->k.sigma), w.sig_direct);
      GPZ_LAUNCH_OK();
    }
  }
  hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((Mp + 255) / 256), L32), dim3(256), 0, s, w.mu_part, pl.nchunks,
                     Mp, w.mu_sum);
  GPZ_LAUNCH_OK();
  auto dgemm = [&](const double* A, const double* B, double* C, int flags) -> int {
    GemmParams<double> d;
    d.A = A; d.lda = Mp; d.sA0 = mm; d.B = B; d.ldb = Mp; d.sB0 = mm; d.C = C; d.ldc = Mp; d.sC0 = mm;
    d.nb0 = L32; d.mt = d.nt = (int)pl.nblk; d.K = (int)Mp; d.flags = flags;
    return gemm_launch(d, EPI_STORE, s);
  };
  if (alg) {
    // dLoss/dLuE = tril(H LuE): H made symmetric, then one M x M product (LuT = LuE^T is the forward's stage-2 operand)
    hipLaunchKernelGGL((mirror_lower_kernel<T>), g32, dim3(256), 0, s, w.H, Mp);
    GPZ_LAUNCH_OK();
    if (int rc = tgemm(w.H, b.LuT, w.G, GF_B_TRANS | GF_B_LOWER | GF_TILES_LOWER)) return rc;
    if (full) {
      // GL = sum_chunks Kbar_x W^T = Linv^T Q,  Q = (Sw - I) H + Hd + muE v^T (v = W gm, before the KL is folded in).  dLoss/dL
      // needs GL's lower triangle only, which needs Q's lower tiles only (Linv^T is upper triangular).
      if (alg64) {
        hipLaunchKernelGGL((widen_kernel<T>), dim3(2048), dim3(256), 0, s, (const T*)w.H, w.D1, L * mm, 0);
        GPZ_LAUNCH_OK();
        if (int rc = dgemm(w.D2, w.D1, w.D3, GF_TILES_LOWER)) return rc;                             // D3 = (Sw - I) H   (lower tiles)
        hipLaunchKernelGGL((q_finish_kernel<double, T>), gm, dim3(256), 0, s, w.D3, (const T*)nullptr, Mp, M,
                           (const double*)w.me, (const double*)w.mu_sum);                            //    + muE v^T
        GPZ_LAUNCH_OK();
        hipLaunchKernelGGL(tril_transpose_kernel, g32, dim3(256), 0, s, b.Linv, Mp, w.D1);
        GPZ_LAUNCH_OK();
        if (int rc = dgemm(w.D1, w.D3, w.D4, GF_A_UPPER | GF_TILES_LOWER)) return rc;                // D4 = GL (lower tiles)
      } else {
        if (int rc = tgemm(w.SwI, w.H, w.PS, GF_TILES_LOWER)) return rc;                             // PS = (Sw - I) H   (lower tiles)
        hipLaunchKernelGGL((q_finish_kernel<T, T>), gm, dim3(256), 0, s, w.PS, wh ? (const T*)w.Hd : (const T*)nullptr, Mp, M,
                           (const double*)w.me, (const double*)w.mu_sum);                            //    + Hd + muE v^T
        GPZ_LAUNCH_OK();
        if (int rc = tgemm(w.LinvT, w.PS, w.GL, GF_A_UPPER | GF_TILES_LOWER)) return rc;             // GL (lower tiles)
      }
    }
  }
  const double* g_kl = g->g_kl;
  if (g_kl && !wh) {
    hipLaunchKernelGGL((kl_add_kernel<T>), gm, dim3(256), 0, s, w.G, Mp, b.LuW, w.mu_sum, b.muE, g_kl);
    GPZ_LAUNCH_OK();
  }
  if (wh)
    hipLaunchKernelGGL((mu_grad_kernel<T>), dim3((unsigned)((M + 255) / 256), L32), dim3(256), 0, s, w.mu_sum,
                       (const double*)nullptr, Mp, M, static_cast<T*>(g->grad_mu), g_kl, static_cast<const T*>(p->mu));
  else
    hipLaunchKernelGGL((mu_grad_linv_kernel<T>), dim3((unsigned)((M + 31) / 32), L32), dim3(256), 0, s, w.mu_sum, b.Linv,
                       Mp, M, static_cast<T*>(g->grad_mu));
  GPZ_LAUNCH_OK();
  T* Gfin = w.G;
  if (!wh) {
    // dLoss/dLu = tril(Linv^T tril(dLoss/dLuE))
    hipLaunchKernelGGL((tril_kernel<T>), gm, dim3(256), 0, s, w.G, Mp);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((transpose_cast_kernel<T>), g32, dim3(256), 0, s, b.Linv, Mp, w.LinvT, b.fro_part);
    GPZ_LAUNCH_OK();
    GemmParams<T> g4;                  // (G2 was zeroed with the other accumulators: its upper tiles are never written)
    g4.A = w.LinvT; g4.lda = Mp; g4.sA0 = mm;
    g4.B = w.G; g4.ldb = Mp; g4.sB0 = mm;
    g4.C = w.G2; g4.ldc = Mp; g4.sC0 = mm;
    g4.nb0 = L32; g4.mt = g4.nt = (int)pl.nblk; g4.K = (int)Mp; g4.flags = GF_A_UPPER | GF_B_LOWER | GF_TILES_LOWER;
    if (int rc = gemm_launch(g4, EPI_STORE, s)) return rc;
    Gfin = w.G2;
  }
  hipLaunchKernelGGL((lu_grad_kernel<T>), dim3((unsigned)((M + 255) / 256), (unsigned)M, L32), dim3(256), 0, s, Gfin, Mp,
                     M, static_cast<const T*>(p->Lu_raw), static_cast<T*>(g->grad_Lu_raw), g_kl, (int)wh);
  GPZ_LAUNCH_OK();
  if (full) {
    // Cholesky backward (Murray 2016): Kbar_zz = Linv^T Phi(L^T Lbar) Linv with Lbar = -tril(GL)
    const double* E = nullptr;
    if (!wh) {
      // muE = Linv mu and LuE = Linv Lu also depend on the factor:
      // E = Linv^T (tril(dLoss/dLuE) LuE^T + dLoss/dmuE muE^T),  Lbar -= tril(E)
      hipLaunchKernelGGL((tril_to_double_kernel<T>), gm, dim3(256), 0, s, w.G, Mp, w.D1);
      GPZ_LAUNCH_OK();
      if (int rc = dgemm(w.D1, b.LuW, w.D2, GF_A_LOWER | GF_B_UPPER | GF_B_TRANS)) return rc;     // D2 = tril(G) LuE^T
      hipLaunchKernelGGL(rank1_update_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)(M < 1024 ? M : 1024), L32),
                         dim3(256), 0, s, w.D2, Mp, M, w.mu_sum, w.me);                           // (me = Linv mu, above)
      GPZ_LAUNCH_OK();
      hipLaunchKernelGGL(tril_transpose_kernel, g32, dim3(256), 0, s, b.Linv, Mp, w.D1);         // D1 = Linv^T
      GPZ_LAUNCH_OK();
      if (int rc = dgemm(w.D1, w.D2, w.D3, GF_A_UPPER)) return rc;                                // D3 = E
      E = w.D3;
    }
    if (alg && alg64)
      hipLaunchKernelGGL((lbar_kernel<T, double>), gm, dim3(256), 0, s, (const double*)w.D4, Mp, M, static_cast<const T*>(g->g_chol), E, w.D2,
                         wh ? nullptr : g_kl, b.Kzz);
    else
      hipLaunchKernelGGL((lbar_kernel<T, T>), gm, dim3(256), 0, s, (const T*)w.GL, Mp, M, static_cast<const T*>(g->g_chol), E, w.D2,
                         wh ? nullptr : g_kl, b.Kzz);
    GPZ_LAUNCH_OK();                                                                              // D2 = Lbar
    hipLaunchKernelGGL(tril_transpose_kernel, g32, dim3(256), 0, s, b.Kzz, Mp, w.D1);            // D1 = L^T
    GPZ_LAUNCH_OK();
    if (int rc = dgemm(w.D1, w.D2, w.D3, GF_A_UPPER | GF_B_LOWER)) return rc;                     // D3 = L^T Lbar
    hipLaunchKernelGGL(phi_kernel, gm, dim3(256), 0, s, w.D3, Mp);
    GPZ_LAUNCH_OK();
    GPZ_HIP_OK(hipMemsetAsync(w.D2, 0, sizeof(double) * L * mm, s));
    if (int rc = dgemm(w.D3, b.Linv, w.D2, GF_A_LOWER | GF_B_LOWER | GF_TILES_LOWER)) return rc;   // D2 = Phi Linv
    hipLaunchKernelGGL(tril_transpose_kernel, g32, dim3(256), 0, s, b.Linv, Mp, w.D1);           // D1 = Linv^T
    GPZ_LAUNCH_OK();
    if (int rc = dgemm(w.D1, w.D2, w.D3, GF_A_UPPER | GF_B_LOWER)) return rc;                     // D3 = P
    hipLaunchKernelGGL((sym_cast_kernel<T>), g32, dim3(256), 0, s, w.D3, Mp, w.PS);               // P + P^T
    GPZ_LAUNCH_OK();
    KgradArgs ka;
    ka.Kbar = w.PS; ka.ld = Mp; ka.stride = mm; ka.Z = p->Z; ka.X = p->Z; ka.gZ = p->gZ; ka.gX = p->gZ;
    ka.sigma = p->k.sigma; ka.ell = p->k.lengthscale; ka.ga = p->k.group_a; ka.gr2 = p->k.group_r2;
    ka.gpow = p->k.group_pow; ka.scalar_scale = 0.5; ka.M = M; ka.ncols = M; ka.Mp = Mp; ka.d = p->d;
    ka.G = p->k.n_groups; ka.acc = w.kacc;
    if (int rc = kgrad_launch(p->dtype, p->k.kind, ka, L32, s)) return rc;
    const unsigned fx = (unsigned)std::max<int64_t>((M + 255) / 256, L);
    hipLaunchKernelGGL(kgrad_finish_kernel, dim3(fx, 2), dim3(256), 0, s, w.kacc, L32, Mp, M, p->d, w.sig_direct,
                       g->grad_Z, g->grad_theta);
    GPZ_LAUNCH_OK();
  }
  return 0;
}

// ---- backward of WSVGP.forward_precomputed (gp.py:308-322 under loss.backward()) ----
// W is the caller's constant; gradients go to mu, the raw Lu and sigma (Kxx = sigma^2):
//   d/dmu = W^T gm,   d/dLu = tril(W^T-chunk (P diag(gv2))^T) with P = Lu^T W^T,   d/dsigma = sigma sum_n gv2_n [unclamped]
template <typename T>
struct PreBwdBuffers { T *Pc, *G, *cs, *csc; double *mu_part, *mu_sum, *sig_direct; size_t bytes; };
template <typename T>
static PreBwdBuffers<T> pre_carve_bwd(const PrePlan& pl, int64_t L, void* ws, size_t offset) {
  PreBwdBuffers<T> b;
  Carver c(ws);
  c.off = offset;
  b.Pc = c.take<T>(L * pl.Mp * pl.nc);
  b.G = c.take<T>(L * pl.Mp * pl.Mp);
  b.cs = c.take<T>(L * pl.nc);
  b.csc = c.take<T>(L * pl.nc);
  b.mu_part = c.take<double>(L * pl.nchunks * pl.Mp);
  b.mu_sum = c.take<double>(L * pl.Mp);
  b.sig_direct = c.take<double>(L);
  b.bytes = c.used();
  return b;
}

__global__ void copy_f64_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

template <typename T>
static int precomputed_backward_t(const void* W, const void* sigma, const void* mu, const void* Lu_raw, int64_t L,
                                  int64_t N, int64_t M, const void* g_mean, const void* g_scale, const void* scale,
                                  void* grad_mu, void* grad_Lu_raw, double* grad_sigma, void* ws, size_t ws_bytes,
                                  hipStream_t s) {
  const PrePlan pl = pre_plan(L, N, M, sizeof(T));
  PreBuffers<T> b = pre_carve<T>(pl, L, ws);
  PreBwdBuffers<T> w = pre_carve_bwd<T>(pl, L, ws, b.bytes);
  GPZ_REQUIRE(ws_bytes >= w.bytes, "gpz_wsvgp_precomputed_backward: workspace too small");
  const int L32 = (int)L;
  const int64_t Mp = pl.Mp, mm = Mp * Mp;
  hipLaunchKernelGGL((lu_prepare_kernel<T>), dim3((unsigned)(Mp / 32), (unsigned)(Mp / 32), L32), dim3(256), 0, s,
                     static_cast<const T*>(Lu_raw), M, Mp, b.LuT, (double*)nullptr, (T*)nullptr, b.lu_part);
  GPZ_LAUNCH_OK();
  GPZ_HIP_OK(hipMemsetAsync(w.G, 0, sizeof(T) * L * mm, s));
  GPZ_HIP_OK(hipMemsetAsync(w.sig_direct, 0, sizeof(double) * L, s));
  for (int64_t ci = 0; ci < pl.nchunks; ++ci) {
    const int64_t n0 = ci * pl.nc;
    const int64_t nreal = (N - n0 < pl.nc) ? N - n0 : pl.nc;
    const int64_t ncp = pad_up(nreal);
    hipLaunchKernelGGL((w_transpose_kernel<T>), dim3((unsigned)(ncp / 32), (unsigned)(Mp / 32), L32), dim3(256), 0, s,
                       static_cast<const T*>(W), N, M, n0, Mp, ncp, b.Wc);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((w_rowstats_kernel<T>), dim3((unsigned)((ncp + 3) / 4), L32), dim3(256), 0, s,
                       static_cast<const T*>(W), static_cast<const T*>(mu), N, M, n0, ncp, b.ps1, b.pm1);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((colscale_kernel<T>), dim3((unsigned)((ncp + 255) / 256), L32), dim3(256), 0, s,
                       static_cast<const T*>(g_scale), static_cast<const T*>(scale), N, n0, ncp, 1, 0.0, w.cs,
                       (const T*)nullptr, (const T*)b.ps1, 1, static_cast<const T*>(sigma), w.csc, (T*)nullptr);
    GPZ_LAUNCH_OK();
    GemmParams<T> g2;  // Pbar = (Lu^T Wt) diag(gv2)
    g2.A = b.LuT; g2.lda = Mp; g2.sA0 = mm;
    g2.B = b.Wc; g2.ldb = ncp; g2.sB0 = Mp * ncp;
    g2.C = w.Pc; g2.ldc = ncp; g2.sC0 = Mp * ncp;
    g2.nb0 = L32; g2.mt = (int)pl.nblk; g2.nt = (int)(ncp / NB); g2.K = (int)Mp; g2.flags = GF_A_UPPER | GF_GROUP_COLS;
    g2.super_cols = product_schedule<T>(false, g2.nt).cols; g2.colscale = w.cs; g2.sCs = ncp; g2.ncols = ncp;
    if (int rc = gemm_launch(g2, EPI_STORE_COLSCALE, s)) return rc;
    GemmParams<T> g3;  // G += Wt Pbar^T  (lower tiles)
    g3.A = b.Wc; g3.lda = ncp; g3.sA0 = Mp * ncp;
    g3.B = w.Pc; g3.ldb = ncp; g3.sB0 = Mp * ncp;
    g3.C = w.G; g3.ldc = Mp; g3.sC0 = mm;
    g3.nb0 = L32; g3.mt = g3.nt = (int)pl.nblk; g3.K = (int)ncp; g3.flags = GF_B_TRANS | GF_TILES_LOWER;
    g3.alpha = 1; g3.beta = 1;
    if (sizeof(T) == 4 && wide_nt_supported(Mp, ncp)) {
      if constexpr (sizeof(T) == 4) { if (int rc = wide_nt_launch(b.Wc, w.Pc, w.G, Mp, ncp, L32, s)) return rc; }
    } else if (int rc = gemm_launch(g3, EPI_STORE, s)) return rc;
    hipLaunchKernelGGL((rowdot_kernel<T>), dim3((unsigned)(Mp / 4), L32), dim3(256), 0, s, b.Wc, Mp, ncp,
                       static_cast<const T*>(g_mean), N, n0, w.mu_part, pl.nchunks, ci);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL((sigma_direct_kernel<T>), dim3(L32), dim3(256), 0, s, w.csc, ncp, static_cast<const T*>(sigma),
                       w.sig_direct);
    GPZ_LAUNCH_OK();
  }
  hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)((Mp + 255) / 256), L32), dim3(256), 0, s, w.mu_part, pl.nchunks,
                     Mp, w.mu_sum);
  GPZ_LAUNCH_OK();
  hipLaunchKernelGGL((mu_grad_kernel<T>), dim3((unsigned)((M + 255) / 256), L32), dim3(256), 0, s, w.mu_sum,
                     (const double*)nullptr, Mp, M, static_cast<T*>(grad_mu), (const double*)nullptr, static_cast<const T*>(mu));
  GPZ_LAUNCH_OK();
  hipLaunchKernelGGL((lu_grad_kernel<T>), dim3((unsigned)((M + 255) / 256), (unsigned)M, L32), dim3(256), 0, s, w.G, Mp, M,
                     static_cast<const T*>(Lu_raw), static_cast<T*>(grad_Lu_raw), (const double*)nullptr, 0);
  GPZ_LAUNCH_OK();
  if (grad_sigma) {
    hipLaunchKernelGGL(copy_f64_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, w.sig_direct, grad_sigma, L32);
    GPZ_LAUNCH_OK();
  }
  return 0;
}

static int check_problem(const gpz_svgp_problem* p) {
  GPZ_REQUIRE(p, "gpz_svgp: null problem");
  GPZ_REQUIRE(p->dtype == GPZ_F32 || p->dtype == GPZ_F64, "gpz_svgp: bad dtype %d", p->dtype);
  GPZ_REQUIRE(p->k.dtype == p->dtype, "gpz_svgp: kernel dtype %d != problem dtype %d", p->k.dtype, p->dtype);
  GPZ_REQUIRE(p->k.n_latent >= 1 && p->N >= 1 && p->M >= 1, "gpz_svgp: bad extents L=%d N=%lld M=%lld",
              p->k.n_latent, (long long)p->N, (long long)p->M);
  GPZ_REQUIRE(p->X && p->Z && p->mu && p->Lu_raw && p->info, "gpz_svgp: null input pointer");
  GPZ_REQUIRE(p->d >= 1 && p->d <= 4, "gpz_svgp: input dimension %d unsupported", p->d);
  if (p->y) GPZ_REQUIRE(p->noise_sd > 0.0, "gpz_svgp: noise_sd must be positive");
  GPZ_REQUIRE((p->flags & ~(GPZ_SVGP_MATERIALIZE_KZX | GPZ_SVGP_NARROW_TILES | GPZ_SVGP_GENERATE_KZX | GPZ_SVGP_PANEL_PRODUCTS |
                             GPZ_SVGP_BACKWARD_ALGEBRA | GPZ_SVGP_BACKWARD_CLASSIC)) == 0,
              "gpz_svgp: unknown bits in flags (0x%x): the field was `reserved` before ABI 210 -- zero it", p->flags);
  return 0;
}

}  // namespace gpz

using namespace gpz;

extern "C" size_t gpz_svgp_workspace_bytes(const gpz_svgp_problem* p, int64_t chunk) {
  if (check_problem(p)) return 0;
  const Plan pl = make_plan(p, chunk);
  return p->dtype == GPZ_F32 ? carve<float>(pl, p->whitened != 0, nullptr).bytes
                             : carve<double>(pl, p->whitened != 0, nullptr).bytes;
}

// Which kernels gpz_svgp_forward takes for the two big products of this problem and chunk: bit 0 the wide-tile fp32
// kernels (gemmw.hip), bit 1 the generated-Kzx stage 1; 4: the panel kernel (gemmp.hip: both products in one launch,
// fp32, Mp <= 512); 0: the 128 x 128-tile kernels (gemm.hip).  -1: bad problem.
extern "C" int gpz_svgp_forward_path(const gpz_svgp_problem* p, int64_t chunk) {
  if (check_problem(p)) return -1;
  const Plan pl = make_plan(p, chunk);
  const bool narrow = (p->flags & GPZ_SVGP_NARROW_TILES) != 0;
  const bool wide = !narrow && pl.f32 && wide_product_supported(pl.Mp, pl.nc);
  const bool fused = !narrow && (p->flags & GPZ_SVGP_GENERATE_KZX) && !(p->flags & GPZ_SVGP_MATERIALIZE_KZX) &&
                     fused1_supported(p->dtype, p->k.kind, p->d);
  const bool panel = !fused && panel_path(p, pl.f32, pl.Mp, pl.nc);
  return panel ? 4 : (wide ? 1 : 0) | (fused ? 2 : 0);
}

extern "C" int gpz_svgp_forward(const gpz_svgp_problem* p, int64_t chunk, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_problem(p)) return rc;
  GPZ_REQUIRE(ws, "gpz_svgp_forward: null workspace");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return p->dtype == GPZ_F32 ? svgp_forward_t<float>(p, chunk, ws, ws_bytes, s)
                             : svgp_forward_t<double>(p, chunk, ws, ws_bytes, s);
}

extern "C" size_t gpz_wsvgp_precomputed_workspace_bytes(int64_t L, int64_t N, int64_t M, int32_t dtype) {
  if (L < 1 || N < 1 || M < 1) return 0;
  const PrePlan pl = pre_plan(L, N, M, dtype == GPZ_F32 ? 4 : 8);
  return dtype == GPZ_F32 ? pre_carve<float>(pl, L, nullptr).bytes : pre_carve<double>(pl, L, nullptr).bytes;
}

extern "C" int gpz_wsvgp_precomputed(const void* W, const void* sigma, const void* mu, const void* Lu_raw, int64_t L,
                                     int64_t N, int64_t M, int32_t dtype, void* mean, void* scale, void* Lu, void* ws,
                                     size_t ws_bytes, void* stream) {
  GPZ_REQUIRE(W && sigma && mu && Lu_raw && mean && scale && ws, "gpz_wsvgp_precomputed: null pointer");
  GPZ_REQUIRE(L >= 1 && N >= 1 && M >= 1, "gpz_wsvgp_precomputed: bad extents");
  GPZ_REQUIRE(dtype == GPZ_F32 || dtype == GPZ_F64, "gpz_wsvgp_precomputed: bad dtype");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == GPZ_F32 ? precomputed_t<float>(W, sigma, mu, Lu_raw, L, N, M, mean, scale, Lu, ws, ws_bytes, s)
                          : precomputed_t<double>(W, sigma, mu, Lu_raw, L, N, M, mean, scale, Lu, ws, ws_bytes, s);
}

extern "C" size_t gpz_wsvgp_precomputed_backward_workspace_bytes(int64_t L, int64_t N, int64_t M, int32_t dtype) {
  if (L < 1 || N < 1 || M < 1) return 0;
  const PrePlan pl = pre_plan(L, N, M, dtype == GPZ_F32 ? 4 : 8);
  return dtype == GPZ_F32 ? pre_carve_bwd<float>(pl, L, nullptr, pre_carve<float>(pl, L, nullptr).bytes).bytes
                          : pre_carve_bwd<double>(pl, L, nullptr, pre_carve<double>(pl, L, nullptr).bytes).bytes;
}

extern "C" int gpz_wsvgp_precomputed_backward(const void* W, const void* sigma, const void* mu, const void* Lu_raw,
                                              int64_t L, int64_t N, int64_t M, int32_t dtype, const void* g_mean,
                                              const void* g_scale, const void* scale, void* grad_mu, void* grad_Lu_raw,
                                              double* grad_sigma, void* ws, size_t ws_bytes, void* stream) {
  GPZ_REQUIRE(W && sigma && mu && Lu_raw && g_mean && g_scale && scale && grad_mu && grad_Lu_raw && ws,
              "gpz_wsvgp_precomputed_backward: null pointer");
  GPZ_REQUIRE(L >= 1 && N >= 1 && M >= 1, "gpz_wsvgp_precomputed_backward: bad extents");
  GPZ_REQUIRE(dtype == GPZ_F32 || dtype == GPZ_F64, "gpz_wsvgp_precomputed_backward: bad dtype");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == GPZ_F32 ? precomputed_backward_t<float>(W, sigma, mu, Lu_raw, L, N, M, g_mean, g_scale, scale, grad_mu,
                                                          grad_Lu_raw, grad_sigma, ws, ws_bytes, s)
                          : precomputed_backward_t<double>(W, sigma, mu, Lu_raw, L, N, M, g_mean, g_scale, scale, grad_mu,
                                                           grad_Lu_raw, grad_sigma, ws, ws_bytes, s);
}

extern "C" size_t gpz_svgp_backward_workspace_bytes(const gpz_svgp_problem* p, int64_t chunk) {
  if (check_problem(p)) return 0;
  const Plan pl = make_plan(p, chunk);
  const bool wh = p->whitened != 0;
  if (p->dtype == GPZ_F32) return carve_bwd<float>(pl, wh, true, nullptr, carve<float>(pl, wh, nullptr).bytes).bytes;
  return carve_bwd<double>(pl, wh, true, nullptr, carve<double>(pl, wh, nullptr).bytes).bytes;
}

extern "C" int gpz_svgp_backward(const gpz_svgp_problem* p, const gpz_svgp_grads* g, int64_t chunk, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (int rc = check_problem(p)) return rc;
  GPZ_REQUIRE(g && g->g_mean && g->g_scale && g->scale && g->grad_mu && g->grad_Lu_raw && ws,
              "gpz_svgp_backward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return p->dtype == GPZ_F32 ? svgp_backward_t<float>(p, g, chunk, ws, ws_bytes, s)
                             : svgp_backward_t<double>(p, g, chunk, ws, ws_bytes, s);
}

extern "C" size_t gpz_svgp_wt_cache_bytes(const gpz_svgp_problem* p, int64_t chunk) {
  if (check_problem(p)) return 0;
  const Plan pl = make_plan(p, chunk);
  return p->dtype == GPZ_F32 ? wt_cache_of<float>(pl, nullptr).bytes() : wt_cache_of<double>(pl, nullptr).bytes();
}

extern "C" size_t gpz_svgp_factor_cache_bytes(const gpz_svgp_problem* p) {
  if (check_problem(p)) return 0;
  const Plan pl = make_plan(p, 0);
  const bool wh = p->whitened != 0;
  return p->dtype == GPZ_F32 ? carve_cache<float>(pl, wh, nullptr).bytes : carve_cache<double>(pl, wh, nullptr).bytes;
}
