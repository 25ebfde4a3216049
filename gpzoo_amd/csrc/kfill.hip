// kfill.hip -- pairwise-distance + covariance fill for L latents at once.
//
// Replaces kernel.forward(A, B) of gpzoo/kernels.py (RBF :118-130, NSF_RBF
// :146-155, batched_RBF :42-58, batched_Matern32 :14-30, MGGP_* :75-104,
// :176-228) and the in-place diagonal jitter of utilities.py:407-418.
//
// HBM-write bound: one squared distance per (i,j) pair is shared by all L
// latents; each lane owns VEC consecutive columns (16 B) so every store
// instruction of a wave writes 1 KiB of contiguous K[l][i][j0..].  Per-latent
// constants (sigma^2, exponent scale, MGGP denominators per group pair) live in
// LDS.  Squared distances are formed by direct differencing in the output
// precision (50x more accurate in fp32 than the matmul expansion).
#include "common.h"
#include "cov.h"

#include <type_traits>

namespace gpz {

#ifndef GPZ_KF_TX                // (overridable for timing variants: tools/kfill_variants.sh)
#define GPZ_KF_TX 64
#define GPZ_KF_TY 4
#define GPZ_KF_ROWS 1
#endif
constexpr int KF_TX = GPZ_KF_TX;     // lanes along columns
constexpr int KF_TY = GPZ_KF_TY;     // rows in flight per block
constexpr int KF_ROWS = GPZ_KF_ROWS; // rows per thread.  (Round 4, config-3 chunks: 8 rows 5.58 TB/s, 2 rows 5.77, 1 row 5.83; 256 threads
                                     // along the columns instead of 64 x 4: 5.60-5.65; a "flat" form -- one latent per workgroup, one
                                     // store per thread, the pattern a plain store loop sustains best (tools/write_probe.hip: 6.9 TB/s) --
                                     // recomputes coordinates and distance per store and drops to 1.9-3.1 TB/s: the fill is not a memset.)
constexpr int KF_MAXL = 256;    // latents per launch
constexpr int KF_MAXTAB = 2048; // MGGP (latent, group pair) table entries per launch

struct KfillArgs {
  const void* A; const void* B;
  const int64_t* gA; const int64_t* gB;
  const void* sigma; const void* ell; const void* ga; const void* gr2;
  void* K;
  int64_t nA, nB;      // real extents
  int64_t pA, pB;      // extents written (>= real; the excess is identity / zero padding)
  int64_t ldk, stride;
  double jitter, gpow;
  int d, L, G, pad_identity;
  int32_t* bad;        // set to -1 when a group id is outside [0, G) (nullable)
};

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; using type = float4; };
template <> struct VecOf<double> { static constexpr int N = 2; using type = double2; };

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ double fast_exp(double x) { return exp(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }

// Group id of a point, range checked: the reference indexes the group embedding with it and raises IndexError
// for an id outside [0, n_groups) (kernels.py:99-100, 177-178, 209-210); here the launch flags it and reads group 0.
__device__ __forceinline__ int checked_group(const int64_t* g, int64_t i, int G, int32_t* bad) {
  const int64_t v = g[i];
  if (v < 0 || v >= G) {
    if (bad) atomicMin(bad, -1);
    return 0;
  }
  return (int)v;
}

// KIND: 0 RBF, 1 Matern-3/2, 2 MGGP RBF, 3 plain distance.  VECST: aligned 16-byte stores allowed.
template <typename Tin, typename To, int KIND, bool VECST>
__global__ __launch_bounds__(KF_TX* KF_TY) void kfill_kernel(KfillArgs a) {
  constexpr int VEC = VecOf<To>::N;
  // fp32 RBF / Matern values come from cov.h, the definition the fused stage-1 product shares (bitwise equal)
  constexpr bool F32COV = std::is_same<To, float>::value && (KIND == 0 || KIND == 1);
  __shared__ To s_amp[KF_MAXL];     // sigma^2
  __shared__ To s_coef[KF_MAXL];    // RBF: -0.5/ell^2 ; Matern: sqrt(3)/ell   (F32COV: cov_const's c0)
  __shared__ To s_c1[F32COV ? KF_MAXL : 1];
  __shared__ To s_tab[KIND == 2 ? 2 * KF_MAXTAB : 2];  // MGGP: [l][ga][gb] -> {exp coef, amplitude}

  const int tid = threadIdx.y * KF_TX + threadIdx.x;
  const int L = a.L, G = a.G;
  for (int l = tid; l < L; l += KF_TX * KF_TY) {
    const To s = (To) static_cast<const Tin*>(a.sigma)[l];
    const To e = (To) static_cast<const Tin*>(a.ell)[l];
    if constexpr (F32COV) {
      const CovConst cc = cov_const<KIND>((float)s, (float)e);
      s_amp[l] = cc.amp; s_coef[l] = cc.c0; s_c1[l] = cc.c1;
    } else {
      s_amp[l] = s * s;
      s_coef[l] = (KIND == 1) ? (To)1.7320508075688772935 / e : (To)-0.5 / (e * e);
    }
  }
  if (KIND == 2) {
    __syncthreads();
    for (int t = tid; t < L * G * G; t += KF_TX * KF_TY) {
      const int l = t / (G * G), g = t - l * G * G;
      const To r2 = (To) static_cast<const Tin*>(a.gr2)[g];
      const To den = (To) static_cast<const Tin*>(a.ga)[l] * r2 + (To)1;
      s_tab[2 * t] = s_coef[l] / den;
      s_tab[2 * t + 1] = s_amp[l] * (To)pow((double)den, -a.gpow);
    }
  }
  __syncthreads();

  const int64_t j0 = ((int64_t)blockIdx.x * KF_TX + threadIdx.x) * VEC;
  if (j0 >= a.pB) return;
  const Tin* Bp = static_cast<const Tin*>(a.B);
  const Tin* Ap = static_cast<const Tin*>(a.A);
  const int d = a.d;

  // coordinates (and groups) of this lane's VEC columns
  To bx[VEC][4];
  int gb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int64_t j = j0 + v;
    gb[v] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) bx[v][k] = (To)0;
    if (j < a.nB) {
      for (int k = 0; k < d; ++k) bx[v][k] = (To)Bp[j * d + k];
      if (KIND == 2) gb[v] = checked_group(a.gB, j, G, a.bad);
    }
  }

  const int64_t i_base = ((int64_t)blockIdx.y * KF_TY + threadIdx.y) * KF_ROWS;
  for (int r = 0; r < KF_ROWS; ++r) {
    const int64_t i = i_base + r;
    if (i >= a.pA) break;
    To ax[4] = {0, 0, 0, 0};
    int gai = 0;
    const bool row_real = i < a.nA;
    if (row_real) {
      for (int k = 0; k < d; ++k) ax[k] = (To)Ap[i * d + k];
      if (KIND == 2) gai = checked_group(a.gA, i, G, a.bad);
    }
    To d2[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      To acc = 0;
      for (int k = 0; k < d; ++k) { const To df = ax[k] - bx[v][k]; acc = fma(df, df, acc); }
      if constexpr (F32COV) d2[v] = cov_radial<KIND>(acc);
      else d2[v] = (KIND == 1 || KIND == 3) ? fast_sqrt(acc) : acc;
    }
    To* Krow = static_cast<To*>(a.K) + i * a.ldk + j0;
    for (int l = 0; l < L; ++l) {
      To out[VEC];
      const To amp = s_amp[l], cf = s_coef[l];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int64_t j = j0 + v;
        To val;
        if constexpr (F32COV) {
          val = cov_value<KIND>(d2[v], amp, cf, s_c1[l]);
        } else if (KIND == 0) {
          val = amp * fast_exp(cf * d2[v]);
        } else if (KIND == 1) {
          const To t = cf * d2[v];
          val = amp * ((To)1 + t) * fast_exp(-t);
        } else if (KIND == 3) {
          val = d2[v];
        } else {
          const int t = 2 * ((l * G + gai) * G + gb[v]);
          val = s_tab[t + 1] * fast_exp(s_tab[t] * d2[v]);
        }
        if (!(row_real && j < a.nB)) val = (a.pad_identity && i == j) ? (To)1 : (To)0;
        else if (i == j) val += (To)a.jitter;
        out[v] = val;
      }
      To* dst = Krow + (int64_t)l * a.stride;
      if (VECST) {
        typename VecOf<To>::type pk;
        if constexpr (VEC == 4) pk = make_float4(out[0], out[1], out[2], out[3]);
        else pk = make_double2(out[0], out[1]);
        *reinterpret_cast<typename VecOf<To>::type*>(dst) = pk;
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (j0 + v < a.pB) dst[v] = out[v];
      }
    }
  }
}

template <typename Tin, typename To>
static int launch_kfill(const gpz_kernel_desc* k, KfillArgs a, hipStream_t s) {
  constexpr int VEC = VecOf<To>::N;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(a.K) % 16 == 0) && ((a.ldk * sizeof(To)) % 16 == 0) &&
                      ((a.stride * sizeof(To)) % 16 == 0) && (a.pB % VEC == 0);
  dim3 block(KF_TX, KF_TY);
  dim3 grid((unsigned)((a.pB + KF_TX * VEC - 1) / (KF_TX * VEC)),
            (unsigned)((a.pA + KF_TY * KF_ROWS - 1) / (KF_TY * KF_ROWS)));
  const int L = k->n_latent, G = k->n_groups;
  int lmax = KF_MAXL;
  if (k->kind == GPZ_KERNEL_MGGP_RBF) {
    GPZ_REQUIRE(G >= 1 && G * G <= KF_MAXTAB, "gpz_kfill: n_groups=%d unsupported (max %d)", G, 45);
    lmax = KF_MAXTAB / (G * G);
    if (lmax > KF_MAXL) lmax = KF_MAXL;
  }
  const size_t esz = sizeof(Tin);
  for (int l0 = 0; l0 < L; l0 += lmax) {
    KfillArgs b = a;
    b.L = (L - l0 < lmax) ? L - l0 : lmax;
    b.sigma = static_cast<const char*>(a.sigma) + l0 * esz;
    b.ell = static_cast<const char*>(a.ell) + l0 * esz;
    if (a.ga) b.ga = static_cast<const char*>(a.ga) + l0 * esz;
    b.K = static_cast<To*>(a.K) + (int64_t)l0 * a.stride;
#define GPZ_KF(KIND)                                                                     \
    if (vec_ok) hipLaunchKernelGGL((kfill_kernel<Tin, To, KIND, true>), grid, block, 0, s, b); \
    else hipLaunchKernelGGL((kfill_kernel<Tin, To, KIND, false>), grid, block, 0, s, b)
    switch (k->kind) {
      case GPZ_KERNEL_RBF: GPZ_KF(0); break;
      case GPZ_KERNEL_MATERN32: GPZ_KF(1); break;
      case GPZ_KERNEL_DISTANCE: GPZ_KF(3); break;
      default: GPZ_KF(2); break;
    }
#undef GPZ_KF
    GPZ_LAUNCH_OK();
  }
  return 0;
}

// Internal entry: padded extents + identity padding (used by the fused forward).
int kfill_padded(const gpz_kernel_desc* k, const void* A, int64_t nA, int64_t pA, const void* B, int64_t nB,
                 int64_t pB, int d, const int64_t* gA, const int64_t* gB, void* K, int64_t ldk, int64_t stride,
                 double jitter, int pad_identity, int out_dtype, hipStream_t s, int32_t* bad_group) {
  GPZ_REQUIRE(k && A && B && K, "gpz_kfill: null pointer");
  GPZ_REQUIRE(d >= 1 && d <= 4, "gpz_kfill: input dimension %d unsupported (1..4)", d);
  GPZ_REQUIRE(k->kind >= 0 && k->kind <= 3, "gpz_kfill: unknown kernel kind %d", k->kind);
  GPZ_REQUIRE(k->n_latent >= 1, "gpz_kfill: n_latent must be >= 1");
  GPZ_REQUIRE(k->dtype == GPZ_F32 || k->dtype == GPZ_F64, "gpz_kfill: bad dtype");
  GPZ_REQUIRE(!(k->dtype == GPZ_F64 && out_dtype == GPZ_F32), "gpz_kfill: fp64 inputs need fp64 output");
  if (k->kind == GPZ_KERNEL_MGGP_RBF)
    GPZ_REQUIRE(gA && gB && k->group_a && k->group_r2, "gpz_kfill: MGGP kernel needs groups, group_a, group_r2");
  if (pA <= 0 || pB <= 0) return 0;
  KfillArgs a;
  a.A = A; a.B = B; a.gA = gA; a.gB = gB;
  a.sigma = k->sigma; a.ell = k->lengthscale; a.ga = k->group_a; a.gr2 = k->group_r2;
  a.K = K; a.nA = nA; a.nB = nB; a.pA = pA; a.pB = pB; a.ldk = ldk; a.stride = stride;
  a.jitter = jitter; a.gpow = k->group_pow; a.d = d; a.L = k->n_latent; a.G = k->n_groups;
  a.pad_identity = pad_identity; a.bad = bad_group;
  if (k->dtype == GPZ_F32 && out_dtype == GPZ_F32) return launch_kfill<float, float>(k, a, s);
  if (k->dtype == GPZ_F32 && out_dtype == GPZ_F64) return launch_kfill<float, double>(k, a, s);
  return launch_kfill<double, double>(k, a, s);
}

}  // namespace gpz

extern "C" int gpz_kfill(const gpz_kernel_desc* k, const void* A, int64_t nA, const void* B, int64_t nB,
                         int32_t d, const int64_t* gA, const int64_t* gB, void* K, int64_t ldk,
                         int64_t stride_k, double jitter, int32_t out_dtype, void* stream) {
  GPZ_REQUIRE(nA >= 0 && nB >= 0 && ldk >= nB, "gpz_kfill: bad extents nA=%lld nB=%lld ldk=%lld",
              (long long)nA, (long long)nB, (long long)ldk);
  return gpz::kfill_padded(k, A, nA, nA, B, nB, nB, d, gA, gB, K, ldk, stride_k, jitter, 0, out_dtype,
                           static_cast<hipStream_t>(stream));
}
