// vnngp.hip -- nearest-neighbour variational GP forward (SURVEY.md §8f "next" #4).
//
// Replaces VNNGP.forward (reference gp.py:21-122): for every x_n the K nearest inducing points
// (argsort(cdist(X, Z))[:, :K], gp.py:31,64), the K x K blocks of Kzz + jitter I (jittered once more,
// gp.py:68-77) and of S = Lu Lu^T, W = k_xz[idx] inv(block), and svgp_forward's moments with
// clamp(cov, 5e-2).  The reference gathers L[:, idx] and Lu[:, idx] ((L,N,K,M) tensors) and
// multiplies them out; here Kzz + jitter I and S are formed once per latent (M x M, MFMA) and the
// K x K blocks are gathered from them.
//
//   knn_kernel        thread = datum; Z streamed through LDS; a sorted (distance, index) list of the K
//                     best candidates lives in registers.  Candidates arrive in index order and only a
//                     strictly smaller distance displaces an entry, so ties resolve to the lower index
//                     exactly like a stable ascending argsort: the neighbour lists are bit-exact
//                     bookkeeping given the distances.
//   vnngp_point_kernel thread = (latent, datum): gathers the K x K blocks, Cholesky-solves for W in
//                     fp64 and evaluates mean = W mu[idx], cov = s^2 + W S W^T - W k.  Per-thread
//                     matrices sit in a [element][thread] global scratch so every access is coalesced.
#include "common.h"
#include "gemm.h"

namespace gpz {

int kfill_padded(const gpz_kernel_desc* k, const void* A, int64_t nA, int64_t pA, const void* B, int64_t nB,
                 int64_t pB, int d, const int64_t* gA, const int64_t* gB, void* K, int64_t ldk, int64_t stride,
                 double jitter, int pad_identity, int out_dtype, hipStream_t s);
int potrf_padded(double* A, int64_t Mp, int64_t lda, int64_t stride, int64_t batch, int64_t m_real, double* Dinv,
                 int32_t* info, hipStream_t s);

constexpr int KNN_MAX = 32;

template <typename T, int KM>
__global__ __launch_bounds__(256) void knn_kernel(const T* __restrict__ X, int64_t N, const T* __restrict__ Z, int64_t M,
                                                 int d, int K, int64_t* __restrict__ idx) {
  __shared__ T sz[256 * 4];
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  T x[4] = {0, 0, 0, 0};
  if (n < N)
    for (int k = 0; k < d; ++k) x[k] = X[n * d + k];
  T bd[KM];
  int bi[KM];
#pragma unroll
  for (int j = 0; j < KM; ++j) { bd[j] = (T)INFINITY; bi[j] = -1; }
  for (int64_t m0 = 0; m0 < M; m0 += 256) {
    __syncthreads();
    const int cnt = (int)((M - m0 < 256) ? M - m0 : 256);
    for (int i = threadIdx.x; i < cnt * d; i += 256) sz[i] = Z[m0 * d + i];
    __syncthreads();
    for (int c = 0; c < cnt; ++c) {
      T d2 = 0;
      for (int k = 0; k < d; ++k) { const T df = x[k] - sz[c * d + k]; d2 = fma(df, df, d2); }
      const T dist = sqrt(d2);                       // the reference ranks cdist's distances
      if (dist < bd[KM - 1]) {
        bd[KM - 1] = dist; bi[KM - 1] = (int)(m0 + c);
#pragma unroll
        for (int j = KM - 1; j > 0; --j)
          if (bd[j] < bd[j - 1]) {                   // strict: equal distances keep index order
            const T td = bd[j]; bd[j] = bd[j - 1]; bd[j - 1] = td;
            const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
          }
      }
    }
  }
  if (n < N)
#pragma unroll
    for (int j = 0; j < KM; ++j)
      if (j < K) idx[n * K + j] = bi[j];
}

// The sorted list above holds KM >= K entries; only a list of exactly K entries reproduces
// "the K smallest" (a longer list is a superset and its first K entries are the same).

template <typename T>
struct VnnArgs {
  const T* X; const T* Z; const T* sigma; const T* ell; const T* mu;
  const double* Kzz; const double* S;     // (L,Mp,Mp): Kzz + jitter I (symmetric), Lu Lu^T
  const int64_t* idx;                     // (N,K)
  double* scratch;                        // [(K*K + 2K)][L*N] doubles
  T* mean; T* scale;
  int64_t N, M, Mp;
  int d, K, L;
  double jitter, clamp_min;
};

template <typename T>
__global__ __launch_bounds__(256) void vnngp_point_kernel(VnnArgs<T> a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.L * a.N;
  if (t >= total) return;
  const int l = (int)(t / a.N);
  const int64_t n = t - (int64_t)l * a.N;
  const int K = a.K;
  const int64_t* id = a.idx + n * K;
  double* A = a.scratch + t;                               // A[p*K+q] at A[(p*K+q) * total]
  double* kx = a.scratch + (int64_t)K * K * total + t;     // kx[p] at kx[p * total]
  double* w = kx + (int64_t)K * total;
  const double* Kl = a.Kzz + (int64_t)l * a.Mp * a.Mp;
  const double* Sl = a.S + (int64_t)l * a.Mp * a.Mp;
  const double sg = (double)a.sigma[l], el = (double)a.ell[l];
  const double s2 = sg * sg, c = -0.5 / (el * el);
  for (int p = 0; p < K; ++p) {
    const int64_t ip = id[p];
    double d2 = 0;
    for (int k = 0; k < a.d; ++k) { const double df = (double)a.X[n * a.d + k] - (double)a.Z[ip * a.d + k]; d2 += df * df; }
    kx[p * total] = s2 * exp(c * d2);
    for (int q = 0; q <= p; ++q) A[(int64_t)(p * K + q) * total] = Kl[ip * a.Mp + id[q]] + (p == q ? a.jitter : 0.0);
  }
  // in-place Cholesky of the K x K block (lower), then two triangular solves
  for (int j = 0; j < K; ++j) {
    double dj = A[(int64_t)(j * K + j) * total];
    for (int k = 0; k < j; ++k) { const double v = A[(int64_t)(j * K + k) * total]; dj -= v * v; }
    dj = sqrt(dj > 0.0 ? dj : 1e-300);
    A[(int64_t)(j * K + j) * total] = dj;
    for (int i = j + 1; i < K; ++i) {
      double v = A[(int64_t)(i * K + j) * total];
      for (int k = 0; k < j; ++k) v -= A[(int64_t)(i * K + k) * total] * A[(int64_t)(j * K + k) * total];
      A[(int64_t)(i * K + j) * total] = v / dj;
    }
  }
  for (int i = 0; i < K; ++i) {                   // C u = k
    double v = kx[i * total];
    for (int k = 0; k < i; ++k) v -= A[(int64_t)(i * K + k) * total] * w[k * total];
    w[i * total] = v / A[(int64_t)(i * K + i) * total];
  }
  for (int i = K - 1; i >= 0; --i) {              // C^T w = u
    double v = w[i * total];
    for (int k = i + 1; k < K; ++k) v -= A[(int64_t)(k * K + i) * total] * w[k * total];
    w[i * total] = v / A[(int64_t)(i * K + i) * total];
  }
  double mean = 0.0, wk = 0.0, wsw = 0.0;
  for (int p = 0; p < K; ++p) {
    const double wp = w[p * total];
    mean += wp * (double)a.mu[(int64_t)l * a.M + id[p]];
    wk += wp * kx[p * total];                     // W (Kzz block) W^T = W k
    double row = 0.0;
    for (int q = 0; q < K; ++q) row += Sl[id[p] * a.Mp + id[q]] * w[q * total];
    wsw += wp * row;
  }
  double cov = s2 + wsw - wk;
  if (!(cov > a.clamp_min)) cov = a.clamp_min;
  a.mean[t] = (T)mean;
  a.scale[t] = (T)sqrt(cov);
}

// (L,Mp,Mp) fp64 symmetric copy of the lower triangle (the fill wrote the full matrix already; this
// is for S = Lu Lu^T whose GEMM writes every tile, nothing to do) -- kept for clarity of intent.

template <typename T>
__global__ __launch_bounds__(256) void vnn_lu_kernel(const T* __restrict__ raw, int64_t M, int64_t Mp,
                                                    double* __restrict__ LuD, T* __restrict__ LuOut) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  double v = 0.0;
  if (i < M && j < M && j <= i) {
    const double x = (double)raw[(int64_t)l * M * M + i * M + j];
    v = (i == j) ? exp(x) : x;
  }
  LuD[(int64_t)l * Mp * Mp + i * Mp + j] = v;
  if (LuOut && i < M && j < M) LuOut[(int64_t)l * M * M + i * M + j] = (T)v;
}

template <typename T>
__global__ void vnn_chol_out_kernel(const double* __restrict__ Lc, int64_t Mp, int64_t M, T* __restrict__ out) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  out[(int64_t)l * M * M + i * M + j] = (j <= i) ? (T)Lc[(int64_t)l * Mp * Mp + i * Mp + j] : (T)0;
}

struct VnnPlan { int64_t L, N, M, Mp; int K; size_t bytes; double *Kzz, *Kfac, *Dinv, *LuD, *S, *scratch; int64_t* idx; };

static VnnPlan vnn_plan(const gpz_svgp_problem* p, int K, bool own_idx, void* ws) {
  VnnPlan pl;
  pl.L = p->k.n_latent; pl.N = p->N; pl.M = p->M; pl.Mp = pad_up(p->M); pl.K = K;
  const int64_t mm = pl.L * pl.Mp * pl.Mp;
  Carver c(ws);
  pl.Kzz = c.take<double>(mm);
  pl.Kfac = c.take<double>(mm);
  pl.Dinv = c.take<double>(pl.L * (pl.Mp / 128) * 128 * 128);
  pl.LuD = c.take<double>(mm);
  pl.S = c.take<double>(mm);
  pl.scratch = c.take<double>((int64_t)(K * K + 2 * K) * pl.L * pl.N);
  pl.idx = own_idx ? c.take<int64_t>(pl.N * K) : nullptr;
  pl.bytes = c.used();
  return pl;
}

template <typename T>
static int knn_t(const void* X, int64_t N, const void* Z, int64_t M, int d, int K, int64_t* idx, hipStream_t s) {
  dim3 grid((unsigned)((N + 255) / 256)), block(256);
#define GPZ_KNN(KM) hipLaunchKernelGGL((knn_kernel<T, KM>), grid, block, 0, s, static_cast<const T*>(X), N, \
                                       static_cast<const T*>(Z), M, d, K, idx)
  // the sorted list must hold exactly K entries (see the note above), so KM == K is instantiated per size class
  // by padding with +inf sentinels: a list of KM >= K entries keeps the K smallest in its first K slots
  if (K <= 4) GPZ_KNN(4); else if (K <= 8) GPZ_KNN(8); else if (K <= 16) GPZ_KNN(16); else GPZ_KNN(32);
#undef GPZ_KNN
  GPZ_LAUNCH_OK();
  return 0;
}

template <typename T>
static int vnngp_t(const gpz_svgp_problem* p, int K, const int64_t* idx_in, void* ws, size_t ws_bytes, hipStream_t s) {
  VnnPlan pl = vnn_plan(p, K, idx_in == nullptr, ws);
  GPZ_REQUIRE(ws_bytes >= pl.bytes, "gpz_vnngp_forward: workspace too small");
  const int64_t L = pl.L, M = pl.M, Mp = pl.Mp, N = pl.N, mm = Mp * Mp;
  const int L32 = (int)L;
  const dim3 gm((unsigned)((Mp + 255) / 256), (unsigned)Mp, L32);
  // Kzz + jitter I (fp64, symmetric, identity padded) and its Cholesky factor (for pU)
  if (int rc = kfill_padded(&p->k, p->Z, M, Mp, p->Z, M, Mp, p->d, nullptr, nullptr, pl.Kzz, Mp, mm, p->jitter, 1, GPZ_F64, s))
    return rc;
  GPZ_HIP_OK(hipMemcpyAsync(pl.Kfac, pl.Kzz, sizeof(double) * L * mm, hipMemcpyDeviceToDevice, s));
  if (int rc = potrf_padded(pl.Kfac, Mp, Mp, mm, L, M, pl.Dinv, p->info, s)) return rc;
  if (p->chol) {
    hipLaunchKernelGGL((vnn_chol_out_kernel<T>), dim3((unsigned)((M + 255) / 256), (unsigned)M, L32), dim3(256), 0, s,
                       pl.Kfac, Mp, M, static_cast<T*>(p->chol));
    GPZ_LAUNCH_OK();
  }
  // S = Lu Lu^T on the fp64 MFMA path
  hipLaunchKernelGGL((vnn_lu_kernel<T>), gm, dim3(256), 0, s, static_cast<const T*>(p->Lu_raw), M, Mp, pl.LuD,
                     static_cast<T*>(p->Lu));
  GPZ_LAUNCH_OK();
  GemmParams<double> g;
  g.A = pl.LuD; g.lda = Mp; g.sA0 = mm; g.B = pl.LuD; g.ldb = Mp; g.sB0 = mm; g.C = pl.S; g.ldc = Mp; g.sC0 = mm;
  g.nb0 = L32; g.mt = g.nt = (int)(Mp / 128); g.K = (int)Mp; g.flags = GF_A_LOWER | GF_B_UPPER | GF_B_TRANS;
  if (int rc = gemm_launch(g, EPI_STORE, s)) return rc;
  const int64_t* idx = idx_in;
  if (!idx) {
    if (int rc = knn_t<T>(p->X, N, p->Z, M, p->d, K, pl.idx, s)) return rc;
    idx = pl.idx;
  }
  VnnArgs<T> a;
  a.X = static_cast<const T*>(p->X); a.Z = static_cast<const T*>(p->Z);
  a.sigma = static_cast<const T*>(p->k.sigma); a.ell = static_cast<const T*>(p->k.lengthscale);
  a.mu = static_cast<const T*>(p->mu); a.Kzz = pl.Kzz; a.S = pl.S; a.idx = idx; a.scratch = pl.scratch;
  a.mean = static_cast<T*>(p->mean); a.scale = static_cast<T*>(p->scale);
  a.N = N; a.M = M; a.Mp = Mp; a.d = p->d; a.K = K; a.L = L32; a.jitter = p->jitter; a.clamp_min = p->var_clamp_min;
  hipLaunchKernelGGL((vnngp_point_kernel<T>), dim3((unsigned)((L * N + 255) / 256)), dim3(256), 0, s, a);
  GPZ_LAUNCH_OK();
  return 0;
}

}  // namespace gpz

using namespace gpz;

extern "C" int gpz_knn(const void* X, int64_t N, const void* Z, int64_t M, int32_t d, int32_t K, int32_t dtype,
                       int64_t* idx, void* stream) {
  GPZ_REQUIRE(X && Z && idx, "gpz_knn: null pointer");
  GPZ_REQUIRE(N >= 1 && M >= 1 && d >= 1 && d <= 4, "gpz_knn: bad extents");
  GPZ_REQUIRE(K >= 1 && K <= KNN_MAX && K <= M, "gpz_knn: K=%d unsupported (1..min(%d, M))", K, KNN_MAX);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == GPZ_F32 ? knn_t<float>(X, N, Z, M, d, K, idx, s) : knn_t<double>(X, N, Z, M, d, K, idx, s);
}

static int vnn_check(const gpz_svgp_problem* p, int K) {
  GPZ_REQUIRE(p && p->X && p->Z && p->mu && p->Lu_raw && p->info && p->mean && p->scale, "gpz_vnngp: null pointer");
  GPZ_REQUIRE(p->dtype == GPZ_F32 || p->dtype == GPZ_F64, "gpz_vnngp: bad dtype");
  GPZ_REQUIRE(p->k.kind == GPZ_KERNEL_RBF, "gpz_vnngp: only the RBF family supports return_distance (kernels.py:118-126)");
  GPZ_REQUIRE(p->k.n_latent >= 1 && p->N >= 1 && p->M >= 1 && p->d >= 1 && p->d <= 4, "gpz_vnngp: bad extents");
  GPZ_REQUIRE(K >= 1 && K <= KNN_MAX && K <= p->M, "gpz_vnngp: K=%d unsupported (1..min(%d, M))", K, KNN_MAX);
  return 0;
}

extern "C" size_t gpz_vnngp_workspace_bytes(const gpz_svgp_problem* p, int32_t K) {
  if (vnn_check(p, K)) return 0;
  return vnn_plan(p, K, true, nullptr).bytes;
}

extern "C" int gpz_vnngp_forward(const gpz_svgp_problem* p, int32_t K, const int64_t* idx, void* ws, size_t ws_bytes,
                                 void* stream) {
  if (int rc = vnn_check(p, K)) return rc;
  GPZ_REQUIRE(ws, "gpz_vnngp_forward: null workspace");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return p->dtype == GPZ_F32 ? vnngp_t<float>(p, K, idx, ws, ws_bytes, s) : vnngp_t<double>(p, K, idx, ws, ws_bytes, s);
}
