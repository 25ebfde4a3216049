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

template <typename T, int KIND>
__global__ __launch_bounds__(256) void kgrad_kernel(KgradArgs a) {
  const int l = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  if (m >= a.M) return;
  const T* kb = static_cast<const T*>(a.Kbar) + (int64_t)l * a.stride + m * a.ld;
  const T* Zp = static_cast<const T*>(a.Z);
  const T* Xp = static_cast<const T*>(a.X);
  const int d = a.d;
  const double sig = (double)static_cast<const T*>(a.sigma)[l];
  const double ell = (double)static_cast<const T*>(a.ell)[l];
  const double s2 = sig * sig, il2 = 1.0 / (ell * ell);
  double z[4] = {0, 0, 0, 0};
  for (int k = 0; k < d; ++k) z[k] = (double)Zp[m * d + k];
  const int gz = (KIND == 2) ? (int)a.gZ[m] : 0;
  const double aeff = (KIND == 2) ? (double)static_cast<const T*>(a.ga)[l] : 0.0;
  double dz[4] = {0, 0, 0, 0}, dsig = 0, dell = 0, da = 0;
  for (int64_t c = lane; c < a.ncols; c += 64) {
    const double g = (double)kb[c];
    double diff[4] = {0, 0, 0, 0}, d2 = 0;
    for (int k = 0; k < d; ++k) { diff[k] = z[k] - (double)Xp[c * d + k]; d2 += diff[k] * diff[k]; }
    double cz;  // dk/dz_m = cz * diff
    if (KIND == 0) {
      const double kv = s2 * exp(-0.5 * d2 * il2);
      dsig += g * 2.0 * kv / sig;
      dell += g * kv * d2 * il2 / ell;
      cz = -kv * il2;
    } else if (KIND == 1) {
      const double v = 1.7320508075688772935 * sqrt(d2) / ell, e = exp(-v);
      dsig += g * 2.0 * sig * (1.0 + v) * e;
      dell += g * s2 * v * v * e / ell;
      cz = -s2 * 3.0 * il2 * e;
    } else {
      const double r2 = (double)static_cast<const T*>(a.gr2)[gz * a.G + (int)a.gX[c]];
      const double den = aeff * r2 + 1.0;
      const double kv = s2 * exp(-0.5 * d2 * il2 / den) * pow(den, -a.gpow);
      dsig += g * 2.0 * kv / sig;
      dell += g * kv * d2 * il2 / (ell * den);
      da += g * kv * (0.5 * d2 * il2 / (den * den) - a.gpow / den) * r2;
      cz = -kv * il2 / den;
    }
    for (int k = 0; k < d; ++k) dz[k] += g * cz * diff[k];
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
