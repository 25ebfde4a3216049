// cov.h -- fp32 covariance of one (point, point) pair, shared by the stand-alone fill (kfill.hip) and by the
// stage-1 product that generates its Kzx operand itself (gemmw.hip, WB_GEN): one definition, explicit
// operation order, no contraction left to the compiler, so both paths produce the same bits.
//
// Replaces the element-wise part of kernels.py:14-30 (Matern-3/2) and :42-58 / :118-130 (RBF).
#pragma once
#include <hip/hip_runtime.h>

namespace gpz {

// Per-latent constants: amp = sigma^2; RBF: c0 = -0.5 / ell^2 * log2(e); Matern-3/2: c0 = sigma^2 sqrt(3) / ell,
// c1 = sqrt(3) / ell * log2(e) (the exponent goes through v_exp_f32 = 2^x).
struct CovConst { float amp, c0, c1; };

template <int KIND>
__device__ __forceinline__ CovConst cov_const(float sigma, float ell) {
#pragma clang fp contract(off)
  CovConst c;
  c.amp = sigma * sigma;
  if (KIND == 1) {
    const float a = 1.7320508075688772935f / ell;
    c.c0 = c.amp * a;
    c.c1 = a * 1.44269504088896341f;
  } else {
    c.c0 = (-0.5f / (ell * ell)) * 1.44269504088896341f;
    c.c1 = 0.f;
  }
  return c;
}

// Squared distance by direct differencing, coordinates in order (50x more accurate in fp32 than
// the matmul expansion of torch.cdist).
template <int D>
__device__ __forceinline__ float cov_d2(const float* a, const float* b) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) { const float df = a[k] - b[k]; acc = __builtin_fmaf(df, df, acc); }
  return acc;
}

// The part of the evaluation that does not depend on the latent: RBF keeps d^2, Matern takes r = sqrt(d^2)
// (v_sqrt_f32, 1 ulp: the correctly rounded sqrtf costs 15 instructions per element).
template <int KIND>
__device__ __forceinline__ float cov_radial(float d2) {
  return KIND == 1 ? __builtin_amdgcn_sqrtf(d2) : d2;
}

template <int KIND>
__device__ __forceinline__ float cov_value(float s, float amp, float c0, float c1) {
#pragma clang fp contract(off)
  if (KIND == 1) {
    const float lin = __builtin_fmaf(c0, s, amp);                // sigma^2 (1 + sqrt(3) r / ell)
    return lin * __builtin_amdgcn_exp2f(-(c1 * s));              // exp(-sqrt(3) r / ell)
  }
  return amp * __builtin_amdgcn_exp2f(c0 * s);                   // sigma^2 exp(-d^2 / (2 ell^2))
}

}  // namespace gpz
