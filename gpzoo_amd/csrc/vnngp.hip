// vnngp.hip -- nearest-neighbour variational GP, forward and backward (SURVEY.md §8f "next" #4).
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
//   vnngp_point_bwd_kernel  same thread mapping: recomputes the point's solve and leaves per point w,
//                     v = A^{-1} gW, k_xz and the upstream factors as a contiguous record.
//   vnngp_gather_kernel  one wave per (inducing point, latent) walks the entries of the inverted neighbour
//                     table (stable counting sort) in ascending order and forms dLoss/dmu[idx], the rows of
//                     the S and Kzz block gradients and dLoss/dZ through k_xz in that FIXED order: no
//                     atomics on values, bitwise reproducible (a table with a repeated neighbour is summed
//                     lane after lane, vnn_dup_kernel).  The dense tails -- dS -> dLu, Cholesky backward of
//                     dLoss/dchol, the contraction with dKzz/d(sigma, lengthscale, Z) -- reuse the fp64
//                     GEMM and kgrad.hip.
#include "common.h"
#include "factor.h"
#include "gemm.h"

#include <algorithm>

// Timing-only diagnostics of vnngp_gather_kernel (WRONG results by construction; SRC=vnngp tools/ablate_fused.sh):
// -DGPZ_VN_ABL=<bits>  1: no LDS row updates, 2: no record / index loads, 4: rows neither zeroed nor written out.
#ifndef GPZ_VN_ABL
#define GPZ_VN_ABL 0
#endif

namespace gpz {

struct KgradArgs {   // kgrad.hip
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

constexpr int KNN_MAX = 32;

// v*v rounded on its own (x.pow(2) in the reference is a separate op): kept out of fma contraction
__device__ __forceinline__ float __fmul_or(float v) { return __fmul_rn(v, v); }
__device__ __forceinline__ double __fmul_or(double v) { return __dmul_rn(v, v); }

// MM: rank by the distances torch.cdist produces on its matmul-expansion path (the one the reference takes whenever
// either point set has more than 25 rows: kernels.py:118 -> cdist -> _euclidean_dist): with xn = sum_k x_k^2 and
// zn = sum_k z_k^2 (squares rounded, then added in order), d^2 = the dot product of [-2x, xn, 1] and [z, 1, zn]
// accumulated as a k-ordered fma chain from zero (what the CPU GEMM does for this 4..6-long inner dimension; verified
// bit for bit against torch 2.10 fp32 on 2M pairs), clamped at 1e-30, then sqrt.  In fp32 at |x| <= 100 these differ
// from the true distances by up to 0.06, so near-ties order differently than with direct differences:
// the neighbour TABLE is index bookkeeping and has to follow the reference's arithmetic, not the better one.
template <typename T, int KM, bool MM>
__global__ __launch_bounds__(256) void knn_kernel(const T* __restrict__ X, int64_t N, const T* __restrict__ Z, int64_t M,
                                                 int d, int K, int64_t* __restrict__ idx) {
  __shared__ T sz[256 * 4];
  __shared__ T szn[256];
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  T x[4] = {0, 0, 0, 0};
  if (n < N)
    for (int k = 0; k < d; ++k) x[k] = X[n * d + k];
  T xn = 0, xm2[4];
  for (int k = 0; k < 4; ++k) xm2[k] = (T)-2 * x[k];
  if (MM)
    for (int k = 0; k < d; ++k) xn = (k == 0) ? __fmul_or(x[k]) : xn + __fmul_or(x[k]);
  T bd[KM];
  int bi[KM];
#pragma unroll
  for (int j = 0; j < KM; ++j) { bd[j] = (T)INFINITY; bi[j] = -1; }
  for (int64_t m0 = 0; m0 < M; m0 += 256) {
    __syncthreads();
    const int cnt = (int)((M - m0 < 256) ? M - m0 : 256);
    for (int i = threadIdx.x; i < cnt * d; i += 256) sz[i] = Z[m0 * d + i];
    if (MM && (int)threadIdx.x < cnt) {
      T zn = 0;
      for (int k = 0; k < d; ++k) { const T zk = Z[(m0 + threadIdx.x) * d + k]; zn = (k == 0) ? __fmul_or(zk) : zn + __fmul_or(zk); }
      szn[threadIdx.x] = zn;
    }
    __syncthreads();
    for (int c = 0; c < cnt; ++c) {
      T dist;
      if (MM) {
        T acc = 0;
        for (int k = 0; k < d; ++k) acc = fma(xm2[k], sz[c * d + k], acc);
        acc = fma(xn, (T)1, acc);
        acc = fma((T)1, szn[c], acc);
        dist = sqrt(acc > (T)1e-30 ? acc : (T)1e-30);
      } else {
        T d2 = 0;
        for (int k = 0; k < d; ++k) { const T df = x[k] - sz[c * d + k]; d2 = fma(df, df, d2); }
        dist = sqrt(d2);                             // the reference ranks cdist's distances
      }
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

// Per-point forward state left in the thread's scratch columns: A = Cholesky factor of the jittered
// K x K block (lower), kx = k(x, z_idx), w = A^{-1} kx, sw = S_block w.  Returns mean and the unclamped cov.
template <typename T>
__device__ __forceinline__ void vnn_point_solve(const VnnArgs<T>& a, int64_t total, int l, int64_t n,
                                                const int64_t* id, double* A, double* kx, double* w, double* sw,
                                                double& mean, double& cov) {
  const int K = a.K;
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
  double wk = 0.0, wsw = 0.0;
  mean = 0.0;
  for (int p = 0; p < K; ++p) {
    const double wp = w[p * total];
    mean += wp * (double)a.mu[(int64_t)l * a.M + id[p]];
    wk += wp * kx[p * total];                     // W (Kzz block) W^T = W k
    double row = 0.0;
    for (int q = 0; q < K; ++q) row += Sl[id[p] * a.Mp + id[q]] * w[q * total];
    if (sw) sw[p * total] = row;
    wsw += wp * row;
  }
  cov = s2 + wsw - wk;
}

template <typename T>
__global__ __launch_bounds__(256) void vnngp_point_kernel(VnnArgs<T> a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.L * a.N;
  if (t >= total) return;
  const int l = (int)(t / a.N);
  const int64_t n = t - (int64_t)l * a.N;
  const int K = a.K;
  double* A = a.scratch + t;                               // A[p*K+q] at A[(p*K+q) * total]
  double* kx = a.scratch + (int64_t)K * K * total + t;     // kx[p] at kx[p * total]
  double* w = kx + (int64_t)K * total;
  double mean, cov;
  vnn_point_solve<T>(a, total, l, n, a.idx + n * K, A, kx, w, nullptr, mean, cov);
  if (!(cov > a.clamp_min)) cov = a.clamp_min;
  a.mean[t] = (T)mean;
  a.scale[t] = (T)sqrt(cov);
}

template <typename T>
struct VnnBwdArgs {
  VnnArgs<T> f;                 // scratch holds (K*K + 4K + 4) columns here: A, kx, w, sw, v, then gm, gcov, dsigma, dell
  const T* g_mean; const T* g_scale;
  double* gmu;                  // (L,Mp)
  double* gS;                   // (L,Mp,Mp)  T with dLoss/dS = T + T^T
  double* gK;                   // (L,Mp,Mp)  T with T + T^T = 2 sym(dLoss/d(Kzz + jitter I)) from the K x K blocks, or null
  double* kacc;                 // (L,Mp,8)   dz0..3, dsigma, dlengthscale (kgrad.hip layout), or null
  T* rec;                       // [L*N][3K + 2], in the problem's precision: per point w[K], v[K], kx[K], gm, gcov CONTIGUOUS -- what vnngp_gather_kernel
                                // reads per entry (from the [column][point] scratch every value is a 64-byte sector of its own)
  const int32_t* inv;           // (N*K) entries n * 32 + p grouped by the inducing point they name, ascending inside a group
  const int32_t* start;         // (M + 1) group boundaries in inv
  const int32_t* dup;           // one word: non-zero when some point names an inducing point twice (caller-supplied tables
                                // only; gpz_knn's lists are distinct by construction), or null
  // The backward pass works on the points in the caller's `point_order` (position n' holds point order[n']; null:
  // identity): thread n' of the point kernel, record n', entry n' K + p.  Along a space-filling curve the points that name
  // one inducing point sit next to each other, so the gather's reads of their records -- random 128-byte reads in the
  // original order, what bounded it -- become runs of neighbouring records.
  const int64_t* order;
  const int32_t* idxp;          // (N,K) neighbour table in that order, int32
  const T* Xp;                  // (N,d) points in that order
};

template <typename T>
__global__ __launch_bounds__(256) void vnngp_point_bwd_kernel(VnnBwdArgs<T> b) {
  const VnnArgs<T>& a = b.f;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.L * a.N;
  if (t >= total) return;
  const int l = (int)(t / a.N);
  const int64_t np = t - (int64_t)l * a.N;                     // position in the pass's order
  const int64_t n = b.order ? b.order[np] : np;                // the point itself
  const int64_t tn = (int64_t)l * a.N + n;
  const int K = a.K;
  const int64_t* id = a.idx + n * K;
  double* A = a.scratch + t;
  double* kx = a.scratch + (int64_t)K * K * total + t;
  double* w = kx + (int64_t)K * total;
  double* sw = w + (int64_t)K * total;
  double* v = sw + (int64_t)K * total;
  double mean, cov;
  vnn_point_solve<T>(a, total, l, n, id, A, kx, w, sw, mean, cov);
  const double gm = (double)b.g_mean[tn];
  // scale = sqrt(clamp(cov, min)): no gradient through a clamped variance (gp.py:117)
  const double gcov = (cov > a.clamp_min) ? 0.5 * (double)b.g_scale[tn] / sqrt(cov) : 0.0;
  // dLoss/dW with W free, then through W = A^{-1} k:  v = A^{-1} gW,  dk = v - gcov W,  dA = -v W^T
  for (int p = 0; p < K; ++p)
    v[p * total] = gm * (double)a.mu[(int64_t)l * a.M + id[p]] + gcov * (2.0 * sw[p * total] - kx[p * total]);
  for (int i = 0; i < K; ++i) {
    double r = v[i * total];
    for (int k = 0; k < i; ++k) r -= A[(int64_t)(i * K + k) * total] * v[k * total];
    v[i * total] = r / A[(int64_t)(i * K + i) * total];
  }
  for (int i = K - 1; i >= 0; --i) {
    double r = v[i * total];
    for (int k = i + 1; k < K; ++k) r -= A[(int64_t)(k * K + i) * total] * v[k * total];
    v[i * total] = r / A[(int64_t)(i * K + i) * total];
  }
  // What the scatter needs per point stays in its scratch columns (w, v, kx are there already): the sums over the
  // points that name an inducing point are formed by vnngp_gather_kernel in a FIXED order (no atomics: the backward is
  // bitwise reproducible; rounds 1-3 added these contributions with fp64 atomics in arrival order).
  const double sg = (double)a.sigma[l], el = (double)a.ell[l], il2 = 1.0 / (el * el);
  double dsig = gcov * 2.0 * sg, dell = 0.0;
  if (b.kacc)
    for (int p = 0; p < K; ++p) {
      const int64_t ip = id[p];
      const double gk = (v[p * total] - gcov * w[p * total]) * kx[p * total];   // dLoss/dk_p * k_p
      double d2 = 0.0;
      for (int k = 0; k < a.d; ++k) {
        const double df = (double)a.X[n * a.d + k] - (double)a.Z[ip * a.d + k];
        d2 += df * df;
      }
      dsig += gk * 2.0 / sg;
      dell += gk * d2 * il2 / el;
    }
  double* ex = v + (int64_t)K * total;
  ex[0] = gm; ex[total] = gcov; ex[2 * total] = dsig; ex[3 * total] = dell;
  T* rec = b.rec + t * (3 * K + 2);
  for (int p = 0; p < K; ++p) { rec[p] = (T)w[p * total]; rec[K + p] = (T)v[p * total]; rec[2 * K + p] = (T)kx[p * total]; }
  rec[3 * K] = (T)gm; rec[3 * K + 1] = (T)gcov;
}

// ---- the same two kernels with the K x K system in REGISTERS (K <= 16) ----
// The scratch form above moves every element of the K x K factor through global memory several times (K = 10: about a
// thousand coalesced 8-byte accesses per thread, 3 GB per forward at N = 40 000, L = 10: it is bound by exactly that).
// For K <= KT in {8, 12, 16} the system is padded to KT with identity rows (their solution entries are zero) and lives in
// a thread's registers, every loop unrolled with compile-time indices; what is left of the memory traffic is the gathers
// of the two K x K blocks (Kzz + jitter I, S = Lu Lu^T) and, for the backward pass, the per-point vectors the gather
// kernels read afterwards (same scratch columns as above; the factor's columns stay unwritten).
template <int KT> __device__ __forceinline__ constexpr int vtri(int p, int q) { return p * (p + 1) / 2 + q; }

template <typename T, int KT>
__device__ __forceinline__ void vnn_point_solve_reg(const VnnArgs<T>& a, int l, int64_t n, const int64_t* idp,
                                                    double (&A)[KT * (KT + 1) / 2], double (&kx)[KT], double (&w)[KT],
                                                    double (&sw)[KT], int64_t (&id)[KT], double& mean, double& cov) {
  const int K = a.K;
  const double* Kl = a.Kzz + (int64_t)l * a.Mp * a.Mp;
  const double* Sl = a.S + (int64_t)l * a.Mp * a.Mp;
  const double sg = (double)a.sigma[l], el = (double)a.ell[l];
  const double s2 = sg * sg, c = -0.5 / (el * el);
#pragma unroll
  for (int p = 0; p < KT; ++p) id[p] = p < K ? idp[p] : 0;
#pragma unroll
  for (int p = 0; p < KT; ++p) {
    if (p < K) {
      double d2 = 0;
      for (int k = 0; k < a.d; ++k) { const double df = (double)a.X[n * a.d + k] - (double)a.Z[id[p] * a.d + k]; d2 += df * df; }
      kx[p] = s2 * exp(c * d2);
    } else {
      kx[p] = 0.0;
    }
#pragma unroll
    for (int q = 0; q <= p; ++q)
      A[vtri<KT>(p, q)] = p < K ? Kl[id[p] * a.Mp + id[q]] + (p == q ? a.jitter : 0.0) : (p == q ? 1.0 : 0.0);
  }
  // in-place Cholesky (lower), then two triangular solves: the operation order of vnn_point_solve
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    double dj = A[vtri<KT>(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) { const double v = A[vtri<KT>(j, k)]; dj -= v * v; }
    dj = sqrt(dj > 0.0 ? dj : 1e-300);
    A[vtri<KT>(j, j)] = dj;
#pragma unroll
    for (int i = j + 1; i < KT; ++i) {
      double v = A[vtri<KT>(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= A[vtri<KT>(i, k)] * A[vtri<KT>(j, k)];
      A[vtri<KT>(i, j)] = v / dj;
    }
  }
#pragma unroll
  for (int i = 0; i < KT; ++i) {                  // C u = k
    double v = kx[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= A[vtri<KT>(i, k)] * w[k];
    w[i] = v / A[vtri<KT>(i, i)];
  }
#pragma unroll
  for (int i = KT - 1; i >= 0; --i) {             // C^T w = u
    double v = w[i];
#pragma unroll
    for (int k = i + 1; k < KT; ++k) v -= A[vtri<KT>(k, i)] * w[k];
    w[i] = v / A[vtri<KT>(i, i)];
  }
  double wk = 0.0, wsw = 0.0;
  mean = 0.0;
#pragma unroll
  for (int p = 0; p < KT; ++p) {
    double row = 0.0;
    if (p < K) {
      mean += w[p] * (double)a.mu[(int64_t)l * a.M + id[p]];
      wk += w[p] * kx[p];
#pragma unroll
      for (int q = 0; q < KT; ++q)
        if (q < K) row += Sl[id[p] * a.Mp + id[q]] * w[q];
      wsw += w[p] * row;
    }
    sw[p] = row;
  }
  cov = s2 + wsw - wk;
}

template <typename T, int KT>
__global__ __launch_bounds__(256) void vnngp_point_reg_kernel(VnnArgs<T> a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.L * a.N;
  if (t >= total) return;
  const int l = (int)(t / a.N);
  const int64_t n = t - (int64_t)l * a.N;
  double A[KT * (KT + 1) / 2], kx[KT], w[KT], sw[KT], mean, cov;
  int64_t id[KT];
  vnn_point_solve_reg<T, KT>(a, l, n, a.idx + n * a.K, A, kx, w, sw, id, mean, cov);
  if (!(cov > a.clamp_min)) cov = a.clamp_min;
  a.mean[t] = (T)mean;
  a.scale[t] = (T)sqrt(cov);
}

template <typename T, int KT>
__global__ __launch_bounds__(256) void vnngp_point_bwd_reg_kernel(VnnBwdArgs<T> b) {
  const VnnArgs<T>& a = b.f;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)a.L * a.N;
  if (t >= total) return;
  const int l = (int)(t / a.N);
  const int64_t np = t - (int64_t)l * a.N;                     // position in the pass's order
  const int64_t n = b.order ? b.order[np] : np;                // the point itself
  const int64_t tn = (int64_t)l * a.N + n;
  const int K = a.K;
  double A[KT * (KT + 1) / 2], kx[KT], w[KT], sw[KT], v[KT], mean, cov;
  int64_t id[KT];
  vnn_point_solve_reg<T, KT>(a, l, n, a.idx + n * K, A, kx, w, sw, id, mean, cov);
  const double gm = (double)b.g_mean[tn];
  const double gcov = (cov > a.clamp_min) ? 0.5 * (double)b.g_scale[tn] / sqrt(cov) : 0.0;
#pragma unroll
  for (int p = 0; p < KT; ++p)
    v[p] = p < K ? gm * (double)a.mu[(int64_t)l * a.M + id[p]] + gcov * (2.0 * sw[p] - kx[p]) : 0.0;
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    double r = v[i];
#pragma unroll
    for (int k = 0; k < i; ++k) r -= A[vtri<KT>(i, k)] * v[k];
    v[i] = r / A[vtri<KT>(i, i)];
  }
#pragma unroll
  for (int i = KT - 1; i >= 0; --i) {
    double r = v[i];
#pragma unroll
    for (int k = i + 1; k < KT; ++k) r -= A[vtri<KT>(k, i)] * v[k];
    v[i] = r / A[vtri<KT>(i, i)];
  }
  const double sg = (double)a.sigma[l], el = (double)a.ell[l], il2 = 1.0 / (el * el);
  double dsig = gcov * 2.0 * sg, dell = 0.0;
  // the per-point record vnngp_gather_kernel reads, and the two per-latent totals' terms in their scratch columns
  T* rec = b.rec + t * (3 * K + 2);
#pragma unroll
  for (int p = 0; p < KT; ++p)
    if (p < K) {
      rec[p] = (T)w[p]; rec[K + p] = (T)v[p]; rec[2 * K + p] = (T)kx[p];
      if (b.kacc) {
        const double gk = (v[p] - gcov * w[p]) * kx[p];
        double d2 = 0.0;
        for (int k = 0; k < a.d; ++k) {
          const double df = (double)a.X[n * a.d + k] - (double)a.Z[id[p] * a.d + k];
          d2 += df * df;
        }
        dsig += gk * 2.0 / sg;
        dell += gk * d2 * il2 / el;
      }
    }
  rec[3 * K] = (T)gm; rec[3 * K + 1] = (T)gcov;
  double* ex = a.scratch + (int64_t)(K * K + 4 * K) * total + t;
  ex[2 * total] = dsig; ex[3 * total] = dell;
}

// ---- the neighbour table inverted: for every inducing point the (point, slot) entries that name it, in ascending order ----
// A stable counting sort of the N K entries e = n K + p by idx[e], in blocks of VNN_IB consecutive entries:
// per-block histograms (integer atomics: the totals are exact), their running sums over the blocks per inducing point,
// an exclusive scan over the inducing points, then every entry's place = start of its group + entries of its group in
// earlier blocks + entries of its group earlier in its own block.
constexpr int VNN_IB = 1024;

// the neighbour table and the points in the pass's order: idxp[n'][p] = idx[order[n']][p] (int32), Xp[n'] = X[order[n']]
template <typename T>
__global__ __launch_bounds__(256) void vnn_permute_kernel(const int64_t* __restrict__ idx, const T* __restrict__ X,
                                                         const int64_t* __restrict__ order, int64_t N, int K, int d,
                                                         int32_t* __restrict__ idxp, T* __restrict__ Xp) {
  const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (np >= N) return;
  const int64_t n = order ? order[np] : np;
  for (int p = 0; p < K; ++p) idxp[np * K + p] = (int32_t)idx[n * K + p];
  for (int k = 0; k < d; ++k) Xp[np * d + k] = X[n * d + k];
}

__global__ __launch_bounds__(256) void vnn_inv_hist_kernel(const int32_t* __restrict__ idx, int64_t NK, int64_t M,
                                                          int32_t* __restrict__ H) {
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < VNN_IB; i += 256) {
    const int64_t e = b * VNN_IB + i;
    if (e < NK) atomicAdd(H + b * M + idx[e], 1);
  }
}

// one WAVE per inducing point: H[b][m] <- entries of m in blocks < b; count[m] = total.  64 blocks per trip with a
// shuffle scan (a thread per point walked the B = N K / 1024 blocks one dependent load after the other: 95 us at N = 40 000)
__global__ __launch_bounds__(256) void vnn_inv_scan_kernel(int32_t* __restrict__ H, int64_t B, int64_t M,
                                                          int32_t* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  int32_t run = 0;
  for (int64_t b0 = 0; b0 < B; b0 += 64) {
    const int64_t b = b0 + lane;
    const int32_t c = b < B ? H[b * M + m] : 0;
    int32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (b < B) H[b * M + m] = run + inc - c;
    run += __shfl(inc, 63);
  }
  if (lane == 0) count[m] = run;
}

// one block: start[m] = sum of count[0..m), start[M] = N K
__global__ __launch_bounds__(256) void vnn_inv_start_kernel(const int32_t* __restrict__ count, int32_t* __restrict__ start,
                                                           int64_t M) {
  __shared__ int32_t part[256];
  const int64_t per = (M + 255) / 256, lo = threadIdx.x * per, hi = lo + per < M ? lo + per : M;
  int32_t t = 0;
  for (int64_t m = lo; m < hi; ++m) t += count[m];
  part[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int i = 0; i < 256; ++i) { const int32_t c = part[i]; part[i] = run; run += c; }
    start[M] = run;
  }
  __syncthreads();
  int32_t run = part[threadIdx.x];
  for (int64_t m = lo; m < hi; ++m) { start[m] = run; run += count[m]; }
}

__global__ __launch_bounds__(256) void vnn_inv_fill_kernel(const int32_t* __restrict__ idx, int64_t NK, int64_t M,
                                                          const int32_t* __restrict__ H, const int32_t* __restrict__ start,
                                                          int32_t* __restrict__ inv, int K) {
  __shared__ int32_t key[VNN_IB];
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < VNN_IB; i += 256) {
    const int64_t e = b * VNN_IB + i;
    key[i] = e < NK ? idx[e] : -1;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < VNN_IB; i += 256) {
    const int32_t m = key[i];
    if (m < 0) continue;
    int32_t r = 0;
    for (int j = 0; j < i; ++j) r += key[j] == m;
    const int32_t e = (int32_t)(b * VNN_IB + i), n = e / K;
    inv[start[m] + H[b * M + m] + r] = n * 32 + (e - n * K);        // (point, slot) packed: K <= 32
  }
}

// one thread per point: does its neighbour list name an inducing point twice?  (a caller-supplied table may; the
// reference's gathers and its inverse of little_Kzz + jitter I accept that, gp.py:66-77)
__global__ __launch_bounds__(256) void vnn_dup_kernel(const int64_t* __restrict__ idx, int64_t N, int K,
                                                     int32_t* __restrict__ flag) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int64_t* id = idx + n * K;
  bool dup = false;
  for (int p = 1; p < K; ++p) {
    const int64_t ip = id[p];
    for (int q = 0; q < p; ++q) dup |= id[q] == ip;
  }
  if (dup) atomicOr(flag, 1);
}

__device__ __forceinline__ float vnn_readlane(float v, int i) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}
__device__ __forceinline__ double vnn_readlane(double v, int i) {
  const long long bits = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), i);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), i);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// One wave per (inducing point ip, latent l): walks the entries that name ip in ascending order and forms, with the two
// rows resident in LDS, row ip of T_S and T_K, gmu[ip] and the dz part of kacc -- the sums the point kernel used to
// scatter with atomics.  The dsigma / dlengthscale totals of a latent go through vnn_theta_sum_kernel.
template <typename T, bool WIDE, bool KG>       // WIDE: records of more than 64 values (K > 20); KG: kernel / Z gradients too
__global__ __launch_bounds__(64) void vnngp_gather_kernel(VnnBwdArgs<T> b) {
  extern __shared__ double vnn_rows[];       // [2][Mp]
  const VnnArgs<T>& a = b.f;
  const int64_t ip = blockIdx.x;
  const int l = blockIdx.y, lane = threadIdx.x, K = a.K;
  const int64_t total = (int64_t)a.L * a.N;
  double* rowS = vnn_rows;
  double* rowK = vnn_rows + a.Mp;
  if (!(GPZ_VN_ABL & 4))
    for (int64_t j = lane; j < (KG ? 2 : 1) * a.Mp; j += 64) vnn_rows[j] = 0.0;
  __syncthreads();
  const int R = 3 * K + 2;                    // record: w[K], v[K], kx[K], gm, gcov
  const double el = (double)a.ell[l], il2 = 1.0 / (el * el);
  double gmu = 0.0, dz = 0.0;                 // lane 0: gmu; lanes k < d: dz_k
  const int32_t e0 = b.start[ip], e1 = b.start[ip + 1];
  // With pairwise distinct neighbours the lanes q <= p of an entry add into distinct columns of the two rows.  A table that
  // names an inducing point twice (b.dup) makes two lanes meet in one column: those launches add lane after lane, which
  // is the same sum in the order q = 0, 1, ... -- still a fixed order.
  const bool serial = b.dup != nullptr && *b.dup != 0;
  // Eight entries per trip: their loads (entry -> point -> the point's vectors: two dependent levels of global latency) are
  // all issued before the first use, the updates are applied in entry order -- the sums are the same sums in the same
  // order.  One entry per trip was bound by exactly that latency: 400 entries x 2 us per wave, 2.8 ms per backward at
  // N = 40 000, M = 1000, L = 10, K = 10.
  // A point's record (3K + 2 values: w, v, kx, gm, gcov) arrives as ONE load per entry, a value per lane, and the values an
  // update needs are broadcast from the lanes that hold them (w[q] is already in lane q).  Loaded value by value -- seven
  // mostly wave-uniform loads per entry -- the kernel was bound by the rate at which a CU issues vector memory
  // instructions, not by where the records lie: laying them out along a Morton curve changed nothing until this did.
  constexpr int U = 8;
  const double zl = (KG && lane < a.d) ? (double)a.Z[ip * a.d + lane] : 0.0;
  auto pick = [&](T lo, T hi, int i) -> double {          // record value i (wave-uniform) out of the two lane vectors
    if constexpr (!WIDE) return (double)vnn_readlane(lo, i);
    return (double)(i < 64 ? vnn_readlane(lo, i) : vnn_readlane(hi, i - 64));
  };
  // Three trips in flight: the entries (inv) of trip i + 2 and the records of trip i + 1 are requested before trip i's
  // updates run -- entry -> point -> record are two dependent levels of global latency, and with the 16 KB of rows a CU
  // holds ten of these waves: unpipelined, a trip cost both round trips plus its updates.
  struct Ent { int64_t n[U]; int pp[U]; };
  struct Rec { T r0[U], r1[U]; int32_t iq[U]; double xd[U]; };
  auto load_ent = [&](int32_t eb, Ent& E) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t e = eb + u < e1 ? b.inv[eb + u] : -1;       // (point << 5) | slot
      E.n[u] = e < 0 ? 0 : e >> 5;
      E.pp[u] = e < 0 ? -1 : e & 31;
    }
  };
  auto load_rec = [&](const Ent& E, Rec& Q) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const T* rec = b.rec + ((int64_t)l * a.N + ((GPZ_VN_ABL & 2) ? 0 : E.n[u])) * R;
      Q.r0[u] = lane < R ? rec[lane] : (T)0;
      Q.r1[u] = (WIDE && lane + 64 < R) ? rec[lane + 64] : (T)0;
      Q.iq[u] = lane < K ? b.idxp[E.n[u] * K + lane] : 0;
      Q.xd[u] = (KG && lane < a.d) ? (double)b.Xp[E.n[u] * a.d + lane] : 0.0;
    }
  };
  Ent e_cur, e_nxt, e_nx2;
  Rec q_cur, q_nxt;
  load_ent(e0, e_cur);
  load_ent(e0 + U, e_nxt);
  load_rec(e_cur, q_cur);
  for (int32_t eb = e0; eb < e1; eb += U) {
    load_ent(eb + 2 * U, e_nx2);
    load_rec(e_nxt, q_nxt);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e_cur.pp[u] < 0) continue;            // past the group's end (wave-uniform)
      const int p = __builtin_amdgcn_readfirstlane(e_cur.pp[u]);
      const T r0 = q_cur.r0[u], r1 = q_cur.r1[u];
      const double gm = pick(r0, r1, 3 * K), gcov = pick(r0, r1, 3 * K + 1);
      const double wp = pick(r0, r1, p), vp = pick(r0, r1, K + p);
      const double wq = (double)r0;                                         // lane q < K holds w[q]
      const double vq = KG ? (double)__shfl(r0, (K + lane) & 63) : 0.0;     // ... and fetches v[q] from lane K + q
      const int32_t iq = q_cur.iq[u];
      if (GPZ_VN_ABL & 1) {
        gmu += gcov * wq + vq * (double)iq;
      } else if (!serial) {
        if (lane <= p) {                        // one unordered pair (p, q <= p) per lane: distinct columns of the rows
          const int q = lane;
          if (gcov != 0.0) rowS[iq] += (p == q ? 0.5 : 1.0) * gcov * wp * wq;
          if constexpr (KG) rowK[iq] += p == q ? -vp * wq : -(vp * wq + vq * wp);
        }
      } else {
        for (int q = 0; q <= p; ++q)
          if (lane == q) {
            if (gcov != 0.0) rowS[iq] += (p == q ? 0.5 : 1.0) * gcov * wp * wq;
            if constexpr (KG) rowK[iq] += p == q ? -vp * wq : -(vp * wq + vq * wp);
          }
      }
      if (lane == 0) gmu += gm * wp;
      if (KG && lane < a.d) {
        const double kxp = pick(r0, r1, 2 * K + p);
        const double gk = (vp - gcov * wp) * kxp;
        dz += gk * (q_cur.xd[u] - zl) * il2;      // dk/dz = k (x - z) / l^2
      }
      // (two entries of one trip can name the same column through different points: the LDS updates of entry u are
      // complete before entry u + 1 reads the row -- one wave, program order)
    }
    e_cur = e_nxt; e_nxt = e_nx2; q_cur = q_nxt;
  }
  __syncthreads();
  double* gS = b.gS + ((int64_t)l * a.Mp + ip) * a.Mp;
  if (GPZ_VN_ABL & 4) { if (lane == 0) b.gmu[(int64_t)l * a.Mp + ip] = gmu + dz; return; }
  for (int64_t j = lane; j < a.Mp; j += 64) gS[j] = rowS[j];
  if constexpr (KG) {
    double* gK = b.gK + ((int64_t)l * a.Mp + ip) * a.Mp;
    for (int64_t j = lane; j < a.Mp; j += 64) gK[j] = rowK[j];
  }
  if (lane == 0) b.gmu[(int64_t)l * a.Mp + ip] = gmu;
  if (KG && lane < a.d) b.kacc[((int64_t)l * a.Mp + ip) * 8 + lane] = dz;
}

// kacc[l][row 0][4, 5] = sum over the points of a latent of dsigma, dlengthscale (fixed order: 64 strided partial sums per
// latent and quantity, then their tree; one block per latent walked 2 N doubles alone: 0.13 ms at N = 40 000)
constexpr int VNN_TS = 64;
template <typename T>
__global__ __launch_bounds__(256) void vnn_theta_part_kernel(VnnBwdArgs<T> b, double* __restrict__ part) {
  __shared__ double sh[256];
  const VnnArgs<T>& a = b.f;
  const int l = blockIdx.y, sgm = blockIdx.x, K = a.K;
  const int64_t total = (int64_t)a.L * a.N;
  const double* ex = a.scratch + (int64_t)(K * K + 4 * K) * total;
  for (int q = 0; q < 2; ++q) {
    double v = 0.0;
    for (int64_t n = (int64_t)sgm * 256 + threadIdx.x; n < a.N; n += (int64_t)VNN_TS * 256) v += ex[(2 + q) * total + (int64_t)l * a.N + n];
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) part[((int64_t)l * 2 + q) * VNN_TS + sgm] = sh[0];
  }
}
template <typename T>
__global__ __launch_bounds__(64) void vnn_theta_sum_kernel(VnnBwdArgs<T> b, const double* __restrict__ part) {
  const VnnArgs<T>& a = b.f;
  const int l = blockIdx.x, lane = threadIdx.x;
  for (int q = 0; q < 2; ++q) {
    double v = part[((int64_t)l * 2 + q) * VNN_TS + lane];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) b.kacc[(int64_t)l * a.Mp * 8 + 4 + q] = v;
  }
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

struct VnnPlan {
  int64_t L, N, M, Mp; int K; size_t bytes;
  double *Kzz, *Kfac, *Dinv, *LuD, *S, *scratch; int64_t* idx;
  double *Linv, *Tmp, *LuE, *muE;                       // KL(qU || pU): L^{-1}, L^{-1} Lu, L^{-1} mu
  uint32_t* fsync;                                       // tickets and flags of the one-launch Cholesky (csrc/coop.hip)
  double *gmu, *gS, *gK, *kacc, *G, *D1, *D2, *rec; void* PS;  // backward only
  int32_t *inv, *istart, *ihist, *itmp, *idup, *idxp;    // backward only: the inverted neighbour table and its scratch
  void* Xp; double* tpart;
};

// What a forward pass hands to the backward pass of the same call (gpz_svgp_problem.factor_cache, gpz_vnngp_state_bytes):
// Kzz + jitter I, its Cholesky factor and the inverses of its diagonal blocks, Lu and S = Lu Lu^T in fp64, and -- written
// when the forward evaluates KL(qU || pU) -- Linv, LuE = Linv Lu, muE = Linv mu.  A backward pass that finds
// factor_cache_valid bit 0 (the factor, Lu, S) / bit 2 (the KL operands) set reads them instead of forming them again:
// 1.4 ms of a 10 ms forward + backward at N = 40 000, M = 1000, L = 10.
struct VnnState { double *Kzz, *Kfac, *Dinv, *LuD, *S, *Linv, *LuE, *muE; size_t bytes; };
static VnnState vnn_state(const gpz_svgp_problem* p, void* mem) {
  VnnState st;
  const int64_t L = p->k.n_latent, Mp = pad_up(p->M), mm = L * Mp * Mp;
  Carver c(mem);
  st.Kzz = c.take<double>(mm);
  st.Kfac = c.take<double>(mm);
  st.Dinv = c.take<double>(L * (Mp / 128) * 128 * 128);
  st.LuD = c.take<double>(mm);
  st.S = c.take<double>(mm);
  st.Linv = c.take<double>(mm);
  st.LuE = c.take<double>(mm);
  st.muE = c.take<double>(L * Mp);
  st.bytes = c.used();
  return st;
}

// M is a few thousand at most on this path: every mode carves the same (generous) set of M x M buffers
static VnnPlan vnn_plan(const gpz_svgp_problem* p, int K, bool own_idx, void* ws, int bwd = 0) {
  VnnPlan pl;
  pl.L = p->k.n_latent; pl.N = p->N; pl.M = p->M; pl.Mp = pad_up(p->M); pl.K = K;
  const int64_t mm = pl.L * pl.Mp * pl.Mp;
  Carver c(ws);
  if (p->factor_cache) {            // the persistent pieces live in the caller's hand-off buffer
    const VnnState st = vnn_state(p, p->factor_cache);
    pl.Kzz = st.Kzz; pl.Kfac = st.Kfac; pl.Dinv = st.Dinv; pl.LuD = st.LuD; pl.S = st.S;
  } else {
    pl.Kzz = c.take<double>(mm);
    pl.Kfac = c.take<double>(mm);
    pl.Dinv = c.take<double>(pl.L * (pl.Mp / 128) * 128 * 128);
    pl.LuD = c.take<double>(mm);
    pl.S = c.take<double>(mm);
  }
  pl.scratch = c.take<double>((int64_t)(K * K + (bwd ? 4 : 2) * K + (bwd ? 4 : 0)) * pl.L * pl.N);
  pl.idx = own_idx ? c.take<int64_t>(pl.N * K) : nullptr;
  pl.Tmp = c.take<double>(mm / 2 + 64);
  if (p->factor_cache) {
    const VnnState st = vnn_state(p, p->factor_cache);
    pl.Linv = st.Linv; pl.LuE = st.LuE; pl.muE = st.muE;
  } else {
    pl.Linv = c.take<double>(mm);
    pl.LuE = c.take<double>(mm);
    pl.muE = c.take<double>(pl.L * pl.Mp);
  }
  pl.fsync = c.take<uint32_t>(coop_sync_words(pl.Mp, pl.L));
  pl.gmu = pl.gS = pl.gK = pl.kacc = pl.G = pl.D1 = pl.D2 = pl.rec = nullptr;
  pl.PS = nullptr;
  pl.inv = pl.istart = pl.ihist = pl.itmp = pl.idup = pl.idxp = nullptr;
  pl.Xp = nullptr; pl.tpart = nullptr;
  if (bwd) {
    const int64_t NK = pl.N * K, IB = (NK + VNN_IB - 1) / VNN_IB;
    pl.inv = c.take<int32_t>(NK);
    pl.istart = c.take<int32_t>(pl.M + 2);
    pl.itmp = c.take<int32_t>(pl.M + 2);
    pl.idup = c.take<int32_t>(16);
    pl.idxp = c.take<int32_t>(NK);
    pl.Xp = c.take<double>(pl.N * 4);
    pl.tpart = c.take<double>(pl.L * 2 * VNN_TS);
    pl.ihist = c.take<int32_t>(IB * pl.M);
    pl.rec = c.take<double>((int64_t)(3 * K + 2) * pl.L * pl.N);
    pl.gmu = c.take<double>(pl.L * pl.Mp);
    pl.gS = c.take<double>(mm);
    pl.G = c.take<double>(mm);
    pl.gK = c.take<double>(mm);
    pl.kacc = c.take<double>(pl.L * pl.Mp * 8);
    pl.PS = c.take<double>(mm);       // holds T; sized for fp64
    pl.D1 = c.take<double>(mm);
    pl.D2 = c.take<double>(mm);
  }
  pl.bytes = c.used();
  return pl;
}

template <typename T>
static int knn_t(const void* X, int64_t N, const void* Z, int64_t M, int d, int K, int64_t* idx, hipStream_t s) {
  dim3 grid((unsigned)((N + 255) / 256)), block(256);
  // torch.cdist (compute mode "use_mm_for_euclid_dist_if_necessary", the default the reference gets) takes the
  // matmul-expansion path when either side has more than 25 rows and plain differences otherwise
  const bool mm = N > 25 || M > 25;
#define GPZ_KNN(KM)                                                                                                   \
  do {                                                                                                                \
    if (mm) hipLaunchKernelGGL((knn_kernel<T, KM, true>), grid, block, 0, s, static_cast<const T*>(X), N,             \
                               static_cast<const T*>(Z), M, d, K, idx);                                               \
    else hipLaunchKernelGGL((knn_kernel<T, KM, false>), grid, block, 0, s, static_cast<const T*>(X), N,               \
                            static_cast<const T*>(Z), M, d, K, idx);                                                  \
  } while (0)
  // a sorted list of KM >= K entries keeps the K smallest in its first K slots
  if (K <= 4) GPZ_KNN(4); else if (K <= 8) GPZ_KNN(8); else if (K <= 16) GPZ_KNN(16); else GPZ_KNN(32);
#undef GPZ_KNN
  GPZ_LAUNCH_OK();
  return 0;
}

// Kzz + jitter I, its Cholesky factor, S = Lu Lu^T and the neighbour table: shared by forward and backward
template <typename T>
static int vnn_prepare(const gpz_svgp_problem* p, VnnPlan& pl, const int64_t* idx_in, VnnArgs<T>& a, hipStream_t s) {
  const int64_t L = pl.L, M = pl.M, Mp = pl.Mp, N = pl.N, mm = Mp * Mp;
  const int L32 = (int)L;
  const dim3 gm((unsigned)((Mp + 255) / 256), (unsigned)Mp, L32);
  const bool handed = p->factor_cache && (p->factor_cache_valid & 1);      // this call's forward left them in the buffer
  if (handed) {
    GPZ_HIP_OK(hipMemsetAsync(p->info, 0, sizeof(int32_t) * L, s));
  } else {
    if (int rc = kfill_padded(&p->k, p->Z, M, Mp, p->Z, M, Mp, p->d, nullptr, nullptr, pl.Kzz, Mp, mm, p->jitter, 1, GPZ_F64, s))
      return rc;
    GPZ_HIP_OK(hipMemcpyAsync(pl.Kfac, pl.Kzz, sizeof(double) * L * mm, hipMemcpyDeviceToDevice, s));
    if (int rc = potrf_padded(pl.Kfac, Mp, Mp, mm, L, M, pl.Dinv, p->info, s, true, pl.fsync)) return rc;
  }
  if (p->chol) {
    hipLaunchKernelGGL((vnn_chol_out_kernel<T>), dim3((unsigned)((M + 255) / 256), (unsigned)M, L32), dim3(256), 0, s,
                       pl.Kfac, Mp, M, static_cast<T*>(p->chol));
    GPZ_LAUNCH_OK();
  }
  if (!handed) {
    // S = Lu Lu^T on the fp64 MFMA path
    hipLaunchKernelGGL((vnn_lu_kernel<T>), gm, dim3(256), 0, s, static_cast<const T*>(p->Lu_raw), M, Mp, pl.LuD,
                       static_cast<T*>(p->Lu));
    GPZ_LAUNCH_OK();
    GemmParams<double> g;
    g.A = pl.LuD; g.lda = Mp; g.sA0 = mm; g.B = pl.LuD; g.ldb = Mp; g.sB0 = mm; g.C = pl.S; g.ldc = Mp; g.sC0 = mm;
    g.nb0 = L32; g.mt = g.nt = (int)(Mp / 128); g.K = (int)Mp; g.flags = GF_A_LOWER | GF_B_UPPER | GF_B_TRANS;
    if (int rc = gemm_launch(g, EPI_STORE, s)) return rc;
  }
  const int64_t* idx = idx_in;
  if (!idx) {
    if (int rc = knn_t<T>(p->X, N, p->Z, M, p->d, pl.K, pl.idx, s)) return rc;
    idx = pl.idx;
  }
  a.X = static_cast<const T*>(p->X); a.Z = static_cast<const T*>(p->Z);
  a.sigma = static_cast<const T*>(p->k.sigma); a.ell = static_cast<const T*>(p->k.lengthscale);
  a.mu = static_cast<const T*>(p->mu); a.Kzz = pl.Kzz; a.S = pl.S; a.idx = idx; a.scratch = pl.scratch;
  a.mean = static_cast<T*>(p->mean); a.scale = static_cast<T*>(p->scale);
  a.N = N; a.M = M; a.Mp = Mp; a.d = p->d; a.K = pl.K; a.L = L32; a.jitter = p->jitter; a.clamp_min = p->var_clamp_min;
  return 0;
}

// muE = Linv mu (fp64) and, per latent, the un-whitened KL(qU || pU) of utilities.py:481 /
// torch kl.py:442:  sum log diag L - sum log diag Lu + (|Linv Lu|_F^2 + |Linv mu|^2 - M) / 2.
// One wave per row i (coalesced along the row): muE_i, and the row's share of the KL into part[l][i]; a second kernel
// adds the shares of a latent in a fixed order (no atomics: bitwise reproducible).  The one-block-per-latent version
// this replaces walked Linv with a thread per row: 1.45 ms at M=1000, L=10 against 0.03 ms.
template <typename T>
__global__ __launch_bounds__(256) void vnn_kl_rows_kernel(const double* __restrict__ Lc, const double* __restrict__ Linv,
                                                         const double* __restrict__ LuE, const T* __restrict__ raw,
                                                         const T* __restrict__ mu, int64_t M, int64_t Mp,
                                                         double* __restrict__ muE, double* __restrict__ part) {
  const int l = blockIdx.y, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= Mp) return;
  const double* Li = Linv + (int64_t)l * Mp * Mp + i * Mp;
  const double* Le = LuE + (int64_t)l * Mp * Mp + i * Mp;
  double me = 0.0, fro = 0.0;
  if (i < M)
    for (int64_t k = lane; k <= i; k += 64) {
      me = fma(Li[k], (double)mu[(int64_t)l * M + k], me);
      const double v = Le[k];
      fro = fma(v, v, fro);
    }
  for (int o = 32; o > 0; o >>= 1) { me += __shfl_down(me, o); fro += __shfl_down(fro, o); }
  if (lane == 0) {
    muE[(int64_t)l * Mp + i] = me;
    part[(int64_t)l * Mp + i] = (i < M) ? 0.5 * me * me + 0.5 * fro + log(Lc[(int64_t)l * Mp * Mp + i * Mp + i]) -
                                              (double)raw[(int64_t)l * M * M + i * M + i]
                                        : 0.0;
  }
}

__global__ __launch_bounds__(256) void vnn_kl_sum_kernel(const double* __restrict__ part, int64_t M, int64_t Mp,
                                                        double* __restrict__ kl) {
  __shared__ double sh[256];
  const int l = blockIdx.x, tid = threadIdx.x;
  double acc = 0.0;
  for (int64_t i = tid; i < Mp; i += 256) acc += part[(int64_t)l * Mp + i];
  sh[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  if (tid == 0) kl[l] = sh[0] - 0.5 * (double)M;
}

// Linv, LuE = Linv Lu (lower), muE and the per-latent KL (kl may be null: backward only needs the operands)
template <typename T>
static int vnn_kl_prepare(const gpz_svgp_problem* p, VnnPlan& pl, double* kl, hipStream_t s) {
  const int64_t L = pl.L, Mp = pl.Mp, mm = Mp * Mp;
  if (int rc = trtri_padded(pl.Kfac, Mp, mm, pl.Dinv, pl.Linv, Mp, L, pl.Tmp, s)) return rc;
  GPZ_HIP_OK(hipMemsetAsync(pl.LuE, 0, sizeof(double) * L * mm, s));
  GemmParams<double> g;
  g.A = pl.Linv; g.lda = Mp; g.sA0 = mm; g.B = pl.LuD; g.ldb = Mp; g.sB0 = mm; g.C = pl.LuE; g.ldc = Mp; g.sC0 = mm;
  g.nb0 = (int)L; g.mt = g.nt = (int)(Mp / 128); g.K = (int)Mp; g.flags = GF_A_LOWER | GF_B_LOWER | GF_TILES_LOWER;
  if (int rc = gemm_launch(g, EPI_STORE, s)) return rc;
  double* part = pl.Tmp;     // the triangular inverse is done with its scratch (L * Mp * Mp / 2 doubles >= L * Mp)
  hipLaunchKernelGGL((vnn_kl_rows_kernel<T>), dim3((unsigned)(Mp / 4), (unsigned)L), dim3(256), 0, s, pl.Kfac, pl.Linv,
                     pl.LuE, static_cast<const T*>(p->Lu_raw), static_cast<const T*>(p->mu), pl.M, Mp, pl.muE, part);
  GPZ_LAUNCH_OK();
  if (kl) {
    hipLaunchKernelGGL(vnn_kl_sum_kernel, dim3((unsigned)L), dim3(256), 0, s, part, pl.M, Mp, kl);
    GPZ_LAUNCH_OK();
  }
  return 0;
}

template <typename T>
static int vnngp_t(const gpz_svgp_problem* p, int K, const int64_t* idx_in, void* ws, size_t ws_bytes, hipStream_t s) {
  VnnPlan pl = vnn_plan(p, K, idx_in == nullptr, ws);
  GPZ_REQUIRE(ws_bytes >= pl.bytes, "gpz_vnngp_forward: workspace too small");
  VnnArgs<T> a;
  if (int rc = vnn_prepare<T>(p, pl, idx_in, a, s)) return rc;
  {
    const dim3 grid((unsigned)((pl.L * pl.N + 255) / 256));
    if (K <= 8) hipLaunchKernelGGL((vnngp_point_reg_kernel<T, 8>), grid, dim3(256), 0, s, a);
    else if (K <= 12) hipLaunchKernelGGL((vnngp_point_reg_kernel<T, 12>), grid, dim3(256), 0, s, a);
    else if (K <= 16) hipLaunchKernelGGL((vnngp_point_reg_kernel<T, 16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((vnngp_point_kernel<T>), grid, dim3(256), 0, s, a);
  }
  GPZ_LAUNCH_OK();
  if (p->kl)
    if (int rc = vnn_kl_prepare<T>(p, pl, p->kl, s)) return rc;
  return 0;
}

// ---- backward tails -----------------------------------------------------------------------------
template <typename T>
__global__ void vnn_mu_out_kernel(const double* __restrict__ gmu, int64_t Mp, int64_t M, T* __restrict__ out) {
  const int l = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m < M) out[(int64_t)l * M + m] = (T)gmu[(int64_t)l * Mp + m];
}

// chain rule of Lu = tril(raw, -1) + diag(exp(diag raw)) applied to G = dLoss/dLu (fp64, padded)
template <typename T>
__global__ void vnn_lu_grad_kernel(const double* __restrict__ G, int64_t Mp, int64_t M, const T* __restrict__ raw,
                                   T* __restrict__ out, const double* __restrict__ g_kl = nullptr) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const double g = G[(int64_t)l * Mp * Mp + i * Mp + j];
  double v = 0.0;
  if (j < i) v = g;
  else if (j == i)   // Lu_ii = exp(raw_ii); the KL's -log Lu_ii contributes -g_kl to the raw diagonal
    v = g * exp((double)raw[(int64_t)l * M * M + i * M + i]) - (g_kl ? g_kl[l] : 0.0);
  out[(int64_t)l * M * M + i * M + j] = (T)v;
}

// dst (L,Mp,Mp) fp64 = tril(src (L,M,M)), zero padded
template <typename T>
__global__ void vnn_tril_in_kernel(const T* __restrict__ src, int64_t M, int64_t Mp, double* __restrict__ dst) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  dst[(int64_t)l * Mp * Mp + i * Mp + j] = (i < M && j <= i) ? (double)src[(int64_t)l * M * M + i * M + j] : 0.0;
}

// dst = transpose(tril(src)), (L,Mp,Mp) fp64
__global__ __launch_bounds__(256) void vnn_tril_transpose_kernel(const double* __restrict__ src, int64_t Mp,
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

// Phi of the Cholesky backward: keep the lower triangle, halve the diagonal
__global__ void vnn_phi_kernel(double* __restrict__ A, int64_t Mp) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  double& v = A[(int64_t)l * Mp * Mp + i * Mp + j];
  if (j > i) v = 0.0;
  else if (j == i) v *= 0.5;
}

// dst = (P + P^T) [+ (Q + Q^T)] cast to T
template <typename T>
__global__ __launch_bounds__(256) void vnn_sym_cast_kernel(const double* __restrict__ P, const double* __restrict__ Q,
                                                          int64_t Mp, T* __restrict__ dst) {
  __shared__ double tile[32][33];
  const int l = blockIdx.z;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t base = (int64_t)l * Mp * Mp;
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t o = base + (j0 + rr) * Mp + i0 + tx;
    tile[rr][tx] = P[o] + (Q ? Q[o] : 0.0);
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int64_t o = base + (i0 + rr) * Mp + j0 + tx;
    dst[o] = (T)(P[o] + (Q ? Q[o] : 0.0) + tile[tx][rr]);
  }
}

// grad_Z[m][k] = sum_l acc[l][m][k];  grad_theta[l][q] = sum_m acc[l][m][4+q]
__global__ __launch_bounds__(256) void vnn_kgrad_finish_kernel(const double* __restrict__ acc, int L, int64_t Mp, int64_t M,
                                                              int d, double* __restrict__ grad_Z,
                                                              double* __restrict__ grad_theta) {
  __shared__ double sh[256];
  if (blockIdx.y == 0) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < M && grad_Z)
      for (int k = 0; k < 4; ++k) {
        double t = 0.0;
        if (k < d)
          for (int l = 0; l < L; ++l) t += acc[((int64_t)l * Mp + m) * 8 + k];
        grad_Z[m * 4 + k] = t;
      }
  } else if ((int)blockIdx.x < L && grad_theta) {
    const int l = blockIdx.x;
    for (int q = 0; q < 2; ++q) {
      double v = 0.0;
      for (int64_t m = threadIdx.x; m < M; m += 256) v += acc[((int64_t)l * Mp + m) * 8 + 4 + q];
      __syncthreads();
      sh[threadIdx.x] = v;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
      }
      if (threadIdx.x == 0) grad_theta[l * 4 + q] = sh[0];
    }
    if (threadIdx.x == 0) grad_theta[l * 4 + 2] = grad_theta[l * 4 + 3] = 0.0;
  }
}

// ---- KL(qU || pU) folded into the backward pass (g_kl = upstream dLoss/dkl_l) --------------------------------
// G[l] += gk[l] * tril(T[l])                      (dKL/dLu = tril(Linv^T LuE); T = Linv^T LuE on the lower tiles)
__global__ void vnn_kl_addG_kernel(double* __restrict__ G, const double* __restrict__ T, int64_t Mp,
                                   const double* __restrict__ g_kl) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > i || j >= Mp) return;
  const int64_t o = (int64_t)l * Mp * Mp + i * Mp + j;
  G[o] += g_kl[l] * T[o];
}

// gmu[l][a] += gk[l] * sum_{i >= a} Linv[l][i][a] * muE[l][i]        (dKL/dmu = Linv^T Linv mu)
// 32 columns x 8 row segments per block (a thread per column alone left 40 blocks walking up to M rows each).
__global__ __launch_bounds__(256) void vnn_kl_mu_kernel(double* __restrict__ gmu, const double* __restrict__ Linv,
                                                       const double* __restrict__ muE, int64_t Mp, int64_t M,
                                                       const double* __restrict__ g_kl) {
  __shared__ double sh[8][33];
  const int l = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t a = (int64_t)blockIdx.x * 32 + tx;
  const double* Li = Linv + (int64_t)l * Mp * Mp;
  double t = 0.0;
  if (a < M)
    for (int64_t i = a + ty; i < M; i += 8) t = fma(Li[i * Mp + a], muE[(int64_t)l * Mp + i], t);
  sh[ty][tx] = t;
  __syncthreads();
  if (ty == 0 && a < M) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += sh[q][tx];
    gmu[(int64_t)l * Mp + a] += g_kl[l] * v;
  }
}

// Q[l][i][j] += muE[l][i] * muE[l][j]
__global__ void vnn_rank1_kernel(double* __restrict__ Q, const double* __restrict__ muE, int64_t Mp) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  Q[(int64_t)l * Mp * Mp + i * Mp + j] += muE[(int64_t)l * Mp + i] * muE[(int64_t)l * Mp + j];
}

// Lbar[l] += gk[l] * (diag(1 / L_ii) - tril(E[l]))   (the factor's share: log-determinant and LuE / muE dependence)
__global__ void vnn_kl_lbar_kernel(double* __restrict__ Lbar, const double* __restrict__ E, const double* __restrict__ Lc,
                                   int64_t Mp, int64_t M, const double* __restrict__ g_kl) {
  const int l = blockIdx.z;
  const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > i || i >= M) return;
  const int64_t o = (int64_t)l * Mp * Mp + i * Mp + j;
  double v = -E[o];
  if (i == j) v += 1.0 / Lc[o];
  Lbar[o] += g_kl[l] * v;
}

template <typename T>
static int vnngp_backward_t(const gpz_svgp_problem* p, const gpz_svgp_grads* g, int K, const int64_t* idx_in, void* ws,
                            size_t ws_bytes, hipStream_t s) {
  const bool kgrads = g->grad_theta != nullptr || g->grad_Z != nullptr;
  const double* g_kl = g->g_kl;
  const bool with_chol = kgrads && (g->g_chol != nullptr || g_kl != nullptr);
  VnnPlan pl = vnn_plan(p, K, idx_in == nullptr, ws, 1);
  GPZ_REQUIRE(ws_bytes >= pl.bytes, "gpz_vnngp_backward: workspace too small");
  const int64_t L = pl.L, M = pl.M, Mp = pl.Mp, mm = Mp * Mp;
  const int L32 = (int)L;
  gpz_svgp_problem q = *p;           // forward-only outputs are not produced again
  q.chol = nullptr; q.Lu = nullptr; q.kl = nullptr;
  VnnBwdArgs<T> b;
  if (int rc = vnn_prepare<T>(&q, pl, idx_in, b.f, s)) return rc;
  if ((g_kl || with_chol) && !(p->factor_cache && (p->factor_cache_valid & 5) == 5))
    if (int rc = vnn_kl_prepare<T>(p, pl, nullptr, s)) return rc;   // Linv (and, for the KL, LuE = Linv Lu and muE = Linv mu)
  // rows >= M of the accumulators are padding: zero; rows < M are written whole by the gather below
  GPZ_HIP_OK(hipMemsetAsync(pl.gmu, 0, sizeof(double) * L * Mp, s));
  GPZ_HIP_OK(hipMemsetAsync(pl.gS, 0, sizeof(double) * L * mm, s));
  if (kgrads) {
    GPZ_HIP_OK(hipMemsetAsync(pl.gK, 0, sizeof(double) * L * mm, s));
    GPZ_HIP_OK(hipMemsetAsync(pl.kacc, 0, sizeof(double) * L * Mp * 8, s));
  }
  b.g_mean = static_cast<const T*>(g->g_mean); b.g_scale = static_cast<const T*>(g->g_scale);
  b.order = g->point_order; b.idxp = pl.idxp; b.Xp = static_cast<const T*>(pl.Xp);
  hipLaunchKernelGGL((vnn_permute_kernel<T>), dim3((unsigned)((pl.N + 255) / 256)), dim3(256), 0, s, b.f.idx, b.f.X, b.order,
                     pl.N, K, p->d, pl.idxp, static_cast<T*>(pl.Xp));
  GPZ_LAUNCH_OK();
  b.gmu = pl.gmu; b.gS = pl.gS; b.gK = kgrads ? pl.gK : nullptr; b.kacc = kgrads ? pl.kacc : nullptr; b.rec = reinterpret_cast<T*>(pl.rec);
  {
    const dim3 grid((unsigned)((L * pl.N + 255) / 256));
    const int K = b.f.K;
    if (K <= 8) hipLaunchKernelGGL((vnngp_point_bwd_reg_kernel<T, 8>), grid, dim3(256), 0, s, b);
    else if (K <= 12) hipLaunchKernelGGL((vnngp_point_bwd_reg_kernel<T, 12>), grid, dim3(256), 0, s, b);
    else if (K <= 16) hipLaunchKernelGGL((vnngp_point_bwd_reg_kernel<T, 16>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((vnngp_point_bwd_kernel<T>), grid, dim3(256), 0, s, b);
  }
  GPZ_LAUNCH_OK();
  {
    // the neighbour table inverted (stable counting sort of the N K entries by the inducing point they name), then the
    // sums per inducing point in that fixed order
    const int64_t NK = pl.N * K, IB = (NK + VNN_IB - 1) / VNN_IB;
    GPZ_REQUIRE(NK < (1ll << 31) && pl.N < (1ll << 26), "gpz_vnngp_backward: N * K = %lld entries exceed the 32-bit neighbour index", (long long)NK);
    GPZ_REQUIRE(2 * Mp * sizeof(double) <= 128 * 1024, "gpz_vnngp_backward: M = %lld inducing points exceed the gather's LDS rows",
                (long long)M);
    GPZ_HIP_OK(hipMemsetAsync(pl.ihist, 0, sizeof(int32_t) * IB * M, s));
    hipLaunchKernelGGL(vnn_inv_hist_kernel, dim3((unsigned)IB), dim3(256), 0, s, pl.idxp, NK, M, pl.ihist);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL(vnn_inv_scan_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, pl.ihist, IB, M, pl.itmp);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL(vnn_inv_start_kernel, dim3(1), dim3(256), 0, s, pl.itmp, pl.istart, M);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL(vnn_inv_fill_kernel, dim3((unsigned)IB), dim3(256), 0, s, pl.idxp, NK, M, pl.ihist, pl.istart, pl.inv, K);
    GPZ_LAUNCH_OK();
    b.inv = pl.inv; b.start = pl.istart; b.dup = nullptr;
    if (idx_in) {                    // a caller's table may repeat a neighbour; the lists of gpz_knn cannot
      GPZ_HIP_OK(hipMemsetAsync(pl.idup, 0, sizeof(int32_t), s));
      hipLaunchKernelGGL(vnn_dup_kernel, dim3((unsigned)((pl.N + 255) / 256)), dim3(256), 0, s, b.f.idx, pl.N, K, pl.idup);
      GPZ_LAUNCH_OK();
      b.dup = pl.idup;
    }
    const size_t lds = (kgrads ? 2 : 1) * Mp * sizeof(double);      // row of T_S (and of T_K) of the inducing point
    const bool wide_rec = 3 * K + 2 > 64;
    auto launch = [&](auto kern) -> int {
      if (lds > 64 * 1024) GPZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      hipLaunchKernelGGL(kern, dim3((unsigned)M, L32), dim3(64), lds, s, b);
      GPZ_LAUNCH_OK();
      return 0;
    };
    int rc = 0;
    if (kgrads) rc = wide_rec ? launch(vnngp_gather_kernel<T, true, true>) : launch(vnngp_gather_kernel<T, false, true>);
    else rc = wide_rec ? launch(vnngp_gather_kernel<T, true, false>) : launch(vnngp_gather_kernel<T, false, false>);
    if (rc) return rc;
    if (kgrads) {
      hipLaunchKernelGGL((vnn_theta_part_kernel<T>), dim3(VNN_TS, L32), dim3(256), 0, s, b, pl.tpart);
      GPZ_LAUNCH_OK();
      hipLaunchKernelGGL((vnn_theta_sum_kernel<T>), dim3(L32), dim3(64), 0, s, b, (const double*)pl.tpart);
      GPZ_LAUNCH_OK();
    }
  }
  const dim3 g32((unsigned)(Mp / 32), (unsigned)(Mp / 32), L32);
  const dim3 gm((unsigned)((Mp + 255) / 256), (unsigned)Mp, L32);
  auto dgemm = [&](const double* A, const double* B, double* C, int flags, double alpha) -> int {
    GemmParams<double> d;
    d.A = A; d.lda = Mp; d.sA0 = mm; d.B = B; d.ldb = Mp; d.sB0 = mm; d.C = C; d.ldc = Mp; d.sC0 = mm;
    d.nb0 = L32; d.mt = d.nt = (int)(Mp / 128); d.K = (int)Mp; d.flags = flags; d.alpha = alpha;
    return gemm_launch(d, EPI_STORE, s);
  };
  // S = Lu Lu^T, dS = T + T^T symmetric: dLoss/dLu = tril(2 dS Lu)
  hipLaunchKernelGGL((vnn_sym_cast_kernel<double>), g32, dim3(256), 0, s, pl.gS, (const double*)nullptr, Mp, pl.D1);
  GPZ_LAUNCH_OK();
  GPZ_HIP_OK(hipMemsetAsync(pl.G, 0, sizeof(double) * L * mm, s));
  if (int rc = dgemm(pl.D1, pl.LuD, pl.G, GF_B_LOWER | GF_TILES_LOWER, 2.0)) return rc;
  double* const LinvT = pl.S;        // the point kernels are done with S and Kzz: scratch from here on
  double* const Tm = pl.Kzz;
  if (g_kl) {
    hipLaunchKernelGGL(vnn_tril_transpose_kernel, g32, dim3(256), 0, s, pl.Linv, Mp, LinvT);
    GPZ_LAUNCH_OK();
    GPZ_HIP_OK(hipMemsetAsync(Tm, 0, sizeof(double) * L * mm, s));
    if (int rc = dgemm(LinvT, pl.LuE, Tm, GF_A_UPPER | GF_B_LOWER | GF_TILES_LOWER, 1.0)) return rc;   // Linv^T LuE
    hipLaunchKernelGGL(vnn_kl_addG_kernel, gm, dim3(256), 0, s, pl.G, Tm, Mp, g_kl);
    GPZ_LAUNCH_OK();
    hipLaunchKernelGGL(vnn_kl_mu_kernel, dim3((unsigned)((M + 31) / 32), L32), dim3(256), 0, s, pl.gmu, pl.Linv, pl.muE,
                       Mp, M, g_kl);
    GPZ_LAUNCH_OK();
  }
  hipLaunchKernelGGL((vnn_mu_out_kernel<T>), dim3((unsigned)((M + 255) / 256), L32), dim3(256), 0, s, pl.gmu, Mp, M,
                     static_cast<T*>(g->grad_mu));
  GPZ_LAUNCH_OK();
  hipLaunchKernelGGL((vnn_lu_grad_kernel<T>), dim3((unsigned)((M + 255) / 256), (unsigned)M, L32), dim3(256), 0, s, pl.G,
                     Mp, M, static_cast<const T*>(p->Lu_raw), static_cast<T*>(g->grad_Lu_raw), g_kl);
  GPZ_LAUNCH_OK();
  if (!kgrads) return 0;
  const double* P = nullptr;
  if (with_chol) {
    // upstream dLoss/dchol (+ the KL's own dependence on the factor) -> Lbar
    if (g->g_chol) {
      hipLaunchKernelGGL((vnn_tril_in_kernel<T>), gm, dim3(256), 0, s, static_cast<const T*>(g->g_chol), M, Mp, pl.D1);
      GPZ_LAUNCH_OK();
    } else {
      GPZ_HIP_OK(hipMemsetAsync(pl.D1, 0, sizeof(double) * L * mm, s));
    }
    if (g_kl) {
      double* const Q = pl.gS;       // free since G was formed
      if (int rc = dgemm(pl.LuE, pl.LuE, Q, GF_A_LOWER | GF_B_UPPER | GF_B_TRANS, 1.0)) return rc;      // LuE LuE^T
      hipLaunchKernelGGL(vnn_rank1_kernel, gm, dim3(256), 0, s, Q, pl.muE, Mp);
      GPZ_LAUNCH_OK();
      if (int rc = dgemm(LinvT, Q, Tm, GF_A_UPPER, 1.0)) return rc;                                    // E = Linv^T Q
      hipLaunchKernelGGL(vnn_kl_lbar_kernel, gm, dim3(256), 0, s, pl.D1, Tm, pl.Kfac, Mp, M, g_kl);
      GPZ_LAUNCH_OK();
    }
    // Cholesky backward (Murray 2016): P = Linv^T Phi(L^T Lbar) Linv
    hipLaunchKernelGGL(vnn_tril_transpose_kernel, g32, dim3(256), 0, s, pl.Kfac, Mp, pl.G);   // G  = L^T
    GPZ_LAUNCH_OK();
    if (int rc = dgemm(pl.G, pl.D1, pl.D2, GF_A_UPPER | GF_B_LOWER, 1.0)) return rc;          // D2 = L^T Lbar
    hipLaunchKernelGGL(vnn_phi_kernel, gm, dim3(256), 0, s, pl.D2, Mp);
    GPZ_LAUNCH_OK();
    GPZ_HIP_OK(hipMemsetAsync(pl.D1, 0, sizeof(double) * L * mm, s));
    if (int rc = dgemm(pl.D2, pl.Linv, pl.D1, GF_A_LOWER | GF_B_LOWER | GF_TILES_LOWER, 1.0)) return rc;   // D1 = Phi Linv
    hipLaunchKernelGGL(vnn_tril_transpose_kernel, g32, dim3(256), 0, s, pl.Linv, Mp, pl.G);  // G  = Linv^T
    GPZ_LAUNCH_OK();
    if (int rc = dgemm(pl.G, pl.D1, pl.D2, GF_A_UPPER | GF_B_LOWER, 1.0)) return rc;          // D2 = P
    P = pl.D2;
  }
  hipLaunchKernelGGL((vnn_sym_cast_kernel<T>), g32, dim3(256), 0, s, pl.gK, P, Mp, static_cast<T*>(pl.PS));
  GPZ_LAUNCH_OK();
  KgradArgs ka;
  ka.Kbar = pl.PS; ka.ld = Mp; ka.stride = mm; ka.Z = p->Z; ka.X = p->Z; ka.gZ = nullptr; ka.gX = nullptr;
  ka.sigma = p->k.sigma; ka.ell = p->k.lengthscale; ka.ga = nullptr; ka.gr2 = nullptr;
  ka.gpow = 0.0; ka.scalar_scale = 0.5; ka.M = M; ka.ncols = M; ka.Mp = Mp; ka.d = p->d; ka.G = 0; ka.acc = pl.kacc;
  if (int rc = kgrad_launch(p->dtype, GPZ_KERNEL_RBF, ka, L32, s)) return rc;
  const unsigned fx = (unsigned)std::max<int64_t>((M + 255) / 256, L);
  hipLaunchKernelGGL(vnn_kgrad_finish_kernel, dim3(fx, 2), dim3(256), 0, s, pl.kacc, L32, Mp, M, p->d, g->grad_Z,
                     g->grad_theta);
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
  GPZ_REQUIRE(p && p->X && p->Z && p->mu && p->Lu_raw && p->info, "gpz_vnngp: null pointer");
  GPZ_REQUIRE(p->dtype == GPZ_F32 || p->dtype == GPZ_F64, "gpz_vnngp: bad dtype");
  GPZ_REQUIRE(p->k.kind == GPZ_KERNEL_RBF, "gpz_vnngp: only the RBF family supports return_distance (kernels.py:118-126)");
  GPZ_REQUIRE(p->k.n_latent >= 1 && p->N >= 1 && p->M >= 1 && p->d >= 1 && p->d <= 4, "gpz_vnngp: bad extents");
  GPZ_REQUIRE(K >= 1 && K <= KNN_MAX && K <= p->M, "gpz_vnngp: K=%d unsupported (1..min(%d, M))", K, KNN_MAX);
  return 0;
}

extern "C" size_t gpz_vnngp_state_bytes(const gpz_svgp_problem* p) {
  if (!p || p->k.n_latent < 1 || p->M < 1) return 0;
  return vnn_state(p, nullptr).bytes;
}

extern "C" size_t gpz_vnngp_workspace_bytes(const gpz_svgp_problem* p, int32_t K) {
  if (vnn_check(p, K)) return 0;
  return vnn_plan(p, K, true, nullptr).bytes;
}

extern "C" int gpz_vnngp_forward(const gpz_svgp_problem* p, int32_t K, const int64_t* idx, void* ws, size_t ws_bytes,
                                 void* stream) {
  if (int rc = vnn_check(p, K)) return rc;
  GPZ_REQUIRE(ws && p->mean && p->scale, "gpz_vnngp_forward: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return p->dtype == GPZ_F32 ? vnngp_t<float>(p, K, idx, ws, ws_bytes, s) : vnngp_t<double>(p, K, idx, ws, ws_bytes, s);
}

extern "C" size_t gpz_vnngp_backward_workspace_bytes(const gpz_svgp_problem* p, int32_t K) {
  if (vnn_check(p, K)) return 0;
  return vnn_plan(p, K, true, nullptr, 1).bytes;
}

extern "C" int gpz_vnngp_backward(const gpz_svgp_problem* p, const gpz_svgp_grads* g, int32_t K, const int64_t* idx,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (int rc = vnn_check(p, K)) return rc;
  GPZ_REQUIRE(ws && g, "gpz_vnngp_backward: null pointer");
  GPZ_REQUIRE(g->g_mean && g->g_scale && g->grad_mu && g->grad_Lu_raw, "gpz_vnngp_backward: null gradient buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return p->dtype == GPZ_F32 ? vnngp_backward_t<float>(p, g, K, idx, ws, ws_bytes, s)
                             : vnngp_backward_t<double>(p, g, K, idx, ws, ws_bytes, s);
}
