// kgrad.hip -- gradients of the covariance kernels: contraction of dLoss/dK with dK/d(theta, Z).
//
// Backward of kernel.forward (gpzoo/kernels.py: RBF :118-130, Matern-3/2 :14-30, MGGP :75-104,
// :176-228) as torch autograd produces it for the reference, without materialising dK/dtheta:
// one wave owns one (latent, inducing point) row of Kbar, walks its columns coalesced,
// recomputes k(z_m, x_n) from the coordinates and accumulates
//   dz_m     += kbar * dk/dz_m            (d values)
//   dsigma_l += kbar * 2 k / sigma,  dlengthscale_l += kbar * dk/dl,  da_l += kbar * dk/da_eff
// in fp64; the row totals are added to acc[l][m][0..7] by the owning wave only, so repeated launches
// over N-chunks accumulate without atomics (bitwise reproducible).
// The Matern-3/2 derivative w.r.t. z is written in its r -> 0 limit-safe form
// dk/dz = -sigma^2 (3 / l^2) exp(-sqrt3 r / l) (z - x); the reference's autograd returns NaN there
// (sqrt at 0), the true derivative is 0.
#include "common.h"

namespace gpz {

struct KgradArgs {
  const void* Kbar; int64_t ld, stride;
  const void* Z; const void* X;
  const int64_t* gZ; const int64_t* gX;
  const void* sigma; const void* ell; const void* ga; const void* gr2;
  double gpow, scalar_scale;
  int64_t M, ncols, Mp;
  int d, G;
  double* acc;  // (L, Mp, 8): dz0..dz3, dsigma, dlengthscale, da_eff, unused
};

__device__ __forceinline__ float kg_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ double kg_exp(double x) { return exp(x); }
__device__ __forceinline__ float kg_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double kg_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float kg_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ double kg_pow(double x, double y) { return pow(x, y); }

// Per-element arithmetic in the problem's precision T (the reference's autograd differentiates in that precision too;
// row totals in fp64.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void kgrad_kernel(KgradArgs a) {
  const int l = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  if (m >= a.M) return;
  const T* kb = static_cast<const T*>(a.Kbar) + (int64_t)l * a.stride + m * a.ld;
  const T* Zp = static_cast<const T*>(a.Z);
  const T* Xp = static_cast<const T*>(a.X);
  const int d = a.d;
  const T sig = static_cast<const T*>(a.sigma)[l];
  const T ell = static_cast<const T*>(a.ell)[l];
  const T s2 = sig * sig, il2 = (T)1 / (ell * ell);
  T z[4] = {0, 0, 0, 0};
  for (int k = 0; k < d; ++k) z[k] = Zp[m * d + k];
  // ids were range-checked by the forward fill (IndexError there); clamped here so a stray one cannot read out of bounds
  const int gz = (KIND == 2) ? ((uint64_t)a.gZ[m] < (uint64_t)a.G ? (int)a.gZ[m] : 0) : 0;
  const T aeff = (KIND == 2) ? static_cast<const T*>(a.ga)[l] : (T)0;
  const T gpow = (T)a.gpow;
  double dz[4] = {0, 0, 0, 0}, dsig = 0, dell = 0, da = 0;
  // four columns per lane and trip: their loads are independent (the loop is bound by the latency of its small loads
  // otherwise), and their contributions are summed in T before they join the fp64 totals
  constexpr int UN = 4;
  for (int64_t c0 = lane; c0 < a.ncols; c0 += 64 * UN) {
    T g[UN], x[UN][4];
    int gx[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t c = c0 + 64 * u;
      const bool in = c < a.ncols;
      g[u] = in ? kb[c] : (T)0;
#pragma unroll
      for (int k = 0; k < 4; ++k) x[u][k] = (in && k < d) ? Xp[c * d + k] : (T)0;
      gx[u] = (KIND == 2 && in) ? ((uint64_t)a.gX[c] < (uint64_t)a.G ? (int)a.gX[c] : 0) : 0;
    }
    T pz[4] = {0, 0, 0, 0}, psig = 0, pell = 0, pa = 0;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      T diff[4], d2 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { diff[k] = (k < d) ? z[k] - x[u][k] : (T)0; d2 = fma(diff[k], diff[k], d2); }
      T cz;  // dk/dz_m = cz * diff
      if (KIND == 0) {
        const T kv = s2 * kg_exp((T)-0.5 * d2 * il2);
        psig += g[u] * (T)2 * kv / sig;
        pell += g[u] * kv * d2 * il2 / ell;
        cz = -kv * il2;
      } else if (KIND == 1) {
        const T v = (T)1.7320508075688772935 * kg_sqrt(d2) / ell, e = kg_exp(-v);
        psig += g[u] * (T)2 * sig * ((T)1 + v) * e;
        pell += g[u] * s2 * v * v * e / ell;
        cz = -s2 * (T)3 * il2 * e;
      } else {
        const T r2 = static_cast<const T*>(a.gr2)[gz * a.G + gx[u]];
        const T den = aeff * r2 + (T)1;
        const T kv = s2 * kg_exp((T)-0.5 * d2 * il2 / den) * kg_pow(den, -gpow);
        psig += g[u] * (T)2 * kv / sig;
        pell += g[u] * kv * d2 * il2 / (ell * den);
        pa += g[u] * kv * ((T)0.5 * d2 * il2 / (den * den) - gpow / den) * r2;
        cz = -kv * il2 / den;
      }
      const T gc = g[u] * cz;
#pragma unroll
      for (int k = 0; k < 4; ++k) pz[k] = fma(gc, diff[k], pz[k]);
    }
    dsig += (double)psig; dell += (double)pell; da += (double)pa;
#pragma unroll
    for (int k = 0; k < 4; ++k) dz[k] += (double)pz[k];
  }
  double v[7] = {dz[0], dz[1], dz[2], dz[3], dsig * a.scalar_scale, dell * a.scalar_scale, da * a.scalar_scale};
#pragma unroll
  for (int i = 0; i < 7; ++i)
    for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_down(v[i], o);
  if (lane == 0) {
    double* dst = a.acc + ((int64_t)l * a.Mp + m) * 8;
#pragma unroll
    for (int i = 0; i < 7; ++i) dst[i] += v[i];
  }
}

int kgrad_launch(int dtype, int kind, const KgradArgs& a, int L, hipStream_t s) {
  GPZ_REQUIRE(kind >= 0 && kind <= 2, "kgrad: unknown kernel kind %d", kind);
  dim3 grid((unsigned)((a.M + 3) / 4), (unsigned)L), block(256);
#define GPZ_KG(T, K) hipLaunchKernelGGL((kgrad_kernel<T, K>), grid, block, 0, s, a)
  if (dtype == GPZ_F32) { if (kind == 0) GPZ_KG(float, 0); else if (kind == 1) GPZ_KG(float, 1); else GPZ_KG(float, 2); }
  else { if (kind == 0) GPZ_KG(double, 0); else if (kind == 1) GPZ_KG(double, 1); else GPZ_KG(double, 2); }
#undef GPZ_KG
  GPZ_LAUNCH_OK();
  return 0;
}

}  // namespace gpz

// ---- public entry: backward of gpz_kfill ------------------------------------------------------
namespace gpz {
// grad_A[m][k] = sum_l acc[l][m][k];  grad_theta[l][0..2] = sum_m acc[l][m][4..6]
__global__ __launch_bounds__(256) void kgrad_public_finish_kernel(const double* __restrict__ acc, int L, int64_t M, int d,
                                                                  double* __restrict__ grad_A,
                                                                  double* __restrict__ grad_theta) {
  __shared__ double sh[4];
  if (blockIdx.y == 0) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < M && grad_A)
      for (int k = 0; k < 4; ++k) {
        double t = 0.0;
        if (k < d)
          for (int l = 0; l < L; ++l) t += acc[((int64_t)l * M + m) * 8 + k];
        grad_A[m * 4 + k] = t;
      }
  } else if ((int)blockIdx.x < L && grad_theta) {
    const int l = blockIdx.x;
    for (int q = 0; q < 3; ++q) {
      double v = 0.0;
      for (int64_t m = threadIdx.x; m < M; m += 256) v += acc[((int64_t)l * M + m) * 8 + 4 + q];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) grad_theta[l * 4 + q] = sh[0] + sh[1] + sh[2] + sh[3];
    }
    if (threadIdx.x == 0) grad_theta[l * 4 + 3] = 0.0;
  }
}
}  // namespace gpz

extern "C" size_t gpz_kgrad_workspace_bytes(int64_t nA, int32_t n_latent) {
  if (nA < 1 || n_latent < 1) return 0;
  gpz::Carver c(nullptr);
  c.take<double>((int64_t)n_latent * nA * 8);
  return c.used();
}

extern "C" int gpz_kgrad(const gpz_kernel_desc* k, const void* A, int64_t nA, const void* B, int64_t nB, int32_t d,
                         const int64_t* gA, const int64_t* gB, const void* Kbar, int64_t ldk, int64_t stride_k,
                         double* grad_theta, double* grad_A, void* ws, size_t ws_bytes, void* stream) {
  using namespace gpz;
  GPZ_REQUIRE(k && A && B && Kbar && ws, "gpz_kgrad: null pointer");
  GPZ_REQUIRE(nA >= 1 && nB >= 1 && ldk >= nB && d >= 1 && d <= 4, "gpz_kgrad: bad extents nA=%lld nB=%lld ldk=%lld d=%d",
              (long long)nA, (long long)nB, (long long)ldk, d);
  GPZ_REQUIRE(k->kind >= 0 && k->kind <= 2, "gpz_kgrad: kernel kind %d has no parameters to differentiate", k->kind);
  GPZ_REQUIRE(k->n_latent >= 1 && (k->dtype == GPZ_F32 || k->dtype == GPZ_F64), "gpz_kgrad: bad kernel description");
  if (k->kind == GPZ_KERNEL_MGGP_RBF)
    GPZ_REQUIRE(gA && gB && k->group_a && k->group_r2 && k->n_groups >= 1, "gpz_kgrad: MGGP kernel needs groups, group_a, group_r2");
  GPZ_REQUIRE(ws_bytes >= gpz_kgrad_workspace_bytes(nA, k->n_latent), "gpz_kgrad: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = k->n_latent;
  Carver c(ws);
  double* acc = c.take<double>((int64_t)L * nA * 8);
  GPZ_HIP_OK(hipMemsetAsync(acc, 0, sizeof(double) * L * nA * 8, s));
  KgradArgs a;
  a.Kbar = Kbar; a.ld = ldk; a.stride = stride_k; a.Z = A; a.X = B; a.gZ = gA; a.gX = gB;
  a.sigma = k->sigma; a.ell = k->lengthscale; a.ga = k->group_a; a.gr2 = k->group_r2;
  a.gpow = k->group_pow; a.scalar_scale = 1.0; a.M = nA; a.ncols = nB; a.Mp = nA; a.d = d; a.G = k->n_groups; a.acc = acc;
  if (int rc = kgrad_launch(k->dtype, k->kind, a, L, s)) return rc;
  const unsigned fx = (unsigned)((nA + 255) / 256 > L ? (nA + 255) / 256 : L);
  hipLaunchKernelGGL(kgrad_public_finish_kernel, dim3(fx, 2), dim3(256), 0, s, acc, L, nA, d, grad_A, grad_theta);
  GPZ_LAUNCH_OK();
  return 0;
}
