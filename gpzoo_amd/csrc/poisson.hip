// poisson.hip -- Monte-Carlo expected Poisson log-likelihood of the NSF factor models and its
// gradients, fused so the (E, D, N) rate tensor never exists in HBM.
//
// Replaces, for the training step, reference likelihoods.py:49-53 (get_rate: softplus(W) @ exp(F)),
// :74-97 / :199-225 (F = qF.rsample((E,)), pY = Poisson(softplus(V) * Z)) and the caller's
// (pY.log_prob(y)).mean(0).sum() of utilities.py:610-616, together with what autograd computes for
// them in loss.backward():
//   F[e,l,n]   = mean[l,n] + scale[l,n] * eps[e,l,n]            (rsample with the caller's eps)
//   Z[e,d,n]   = sum_l W[d,l] exp(F[e,l,n]),   rate = V[n] Z[e,d,n]
//   loglik[0]  = (1/E) sum_{e,d,n} ( y[d,n] log rate - rate ),   loglik[1] = sum_{d,n} lgamma(y[d,n] + 1)
//   dW[d,l]    = (1/E) sum_{e,n} (y/Z - V) expF,     dexpF[e,l,n] = (1/E) sum_d (y/Z - V) W[d,l]
//   dV[n]      = (1/E) sum_{e,d} (y/V - Z),          dmean = sum_e dexpF expF,  dscale = sum_e dexpF expF eps
// W and V are the POSITIVE (already soft-plussed) factors; the chain through softplus stays in torch.
//
// Work split (all fp32 VALU, fp64 block sums, no atomics -> bitwise reproducible):
//   expf_kernel    expF (E,Lt,N) once (a few MB, L2 resident afterwards)
//   spot_kernel    thread = spot n, block = 256 spots x one slice of the genes: y[d][n] is read
//                  coalesced, W rows come from LDS as broadcasts, exp(F) and the dexpF accumulators
//                  live in registers; emits log-lik, dV and dexpF partial slabs per gene slice
//   gene_kernel    wave = gene d, lanes sweep the spots: recomputes Z from the expF tile staged in LDS
//                  and keeps dW[d][0..Lt) in registers, one wave reduction per gene at the end
//   finish_kernel  sums the gene-slice slabs into dmean, dscale, dV and the scalar
#include "common.h"

namespace gpz {

constexpr int PMAXL = 64;   // factors (spatial + non-spatial): the notebooks' hybrids run L = 20 spatial + T = 19..20
constexpr int PMAXE = 4;    // Monte-Carlo samples handled per launch group

struct PoissonArgs {
  const float* mean; const float* scale; const float* eps;   // (Lt,N), (Lt,N), (E,Lt,N)
  const float* W; const float* V; const float* y;            // (D,Lt) positive, (N,) positive, (D,N)
  float* expF;                                               // (E,Lt,N) scratch
  float* dexp_slab; float* dV_slab; double* ll_slab;          // [S][E][Lt][N], [S][N], [2][S][nblk]
  float* dW; float* dmean; float* dscale; float* dV; double* loglik;
  int64_t N, D;
  int Lt, E, S, with_lgamma;
};

__global__ void expf_kernel(PoissonArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)a.Lt * a.N;
  if (i >= per * a.E) return;
  const int64_t ln = i % per;
  a.expF[i] = __expf(a.mean[ln] + a.scale[ln] * a.eps[i]);
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// grid (ceil(N/256), S): block = 256 spots x gene slice s
template <int LT, int E>
__global__ __launch_bounds__(256) void spot_kernel(PoissonArgs a) {
  constexpr int WCH = 64;                 // genes staged per LDS refill
  __shared__ float sW[WCH][LT];
  __shared__ double sh[8];
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  const bool live = n < a.N;
  const int64_t per = (int64_t)a.Lt * a.N;
  float ef[E][LT], de[E][LT];
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      ef[e][l] = (live && l < a.Lt) ? a.expF[e * per + (int64_t)l * a.N + n] : 0.f;
      de[e][l] = 0.f;
    }
  const float Vn = live ? a.V[n] : 1.f;
  const float inv_e = 1.f / (float)E;
  const float inv_v = __builtin_amdgcn_rcpf(Vn);
  const int64_t dper = (a.D + a.S - 1) / a.S;
  const int64_t d_lo = s * dper, d_hi = (d_lo + dper < a.D) ? d_lo + dper : a.D;
  double ll = 0.0, lg = 0.0;
  float dv = 0.f, llf = 0.f, llg = 0.f;   // fp32 partials over one 64-gene refill, folded into fp64
  for (int64_t d0 = d_lo; d0 < d_hi; d0 += WCH) {
    __syncthreads();
    for (int i = threadIdx.x; i < WCH * LT; i += 256) {
      const int r = i / LT, l = i - r * LT;
      sW[r][l] = (d0 + r < d_hi && l < a.Lt) ? a.W[(d0 + r) * a.Lt + l] : 0.f;
    }
    __syncthreads();
    const int rows = (int)((d_hi - d0 < WCH) ? d_hi - d0 : WCH);
    constexpr int RB = 8;                 // y rows fetched ahead of their use (hides the global-load latency)
    for (int r0 = 0; r0 < rows; r0 += RB) {
      float yb[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) yb[k] = (live && r0 + k < rows) ? a.y[(d0 + r0 + k) * a.N + n] : 0.f;
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + k;
        if (r < rows) {
          const float yv = yb[k];
          float lsum = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            float z = 0.f;
#pragma unroll
            for (int l = 0; l < LT; ++l) z = fmaf(sW[r][l], ef[e][l], z);
            const float rate = Vn * z;
            lsum += yv * __logf(rate) - rate;
            const float g = (yv * __builtin_amdgcn_rcpf(z) - Vn) * inv_e;
            dv += (yv * inv_v - z) * inv_e;
#pragma unroll
            for (int l = 0; l < LT; ++l) de[e][l] = fmaf(g, sW[r][l], de[e][l]);
          }
          llf += lsum;
          if (a.with_lgamma) llg += lgammaf(yv + 1.f);
        }
      }
    }
    ll += (double)(llf * inv_e);
    lg += (double)llg;
    llf = 0.f; llg = 0.f;
  }
  if (live) {
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
      for (int l = 0; l < LT; ++l)
        if (l < a.Lt) a.dexp_slab[((int64_t)s * E + e) * per + (int64_t)l * a.N + n] = de[e][l];
    a.dV_slab[(int64_t)s * a.N + n] = dv;
  }
  const double t = block_sum_d(live ? ll : 0.0, sh);
  const double tg = block_sum_d(live ? lg : 0.0, sh);
  if (threadIdx.x == 0) {
    a.ll_slab[(int64_t)s * gridDim.x + blockIdx.x] = t;
    a.ll_slab[((int64_t)a.S + s) * gridDim.x + blockIdx.x] = tg;
  }
}

// grid (ceil(D/(4*GPW))): each wave owns GPW genes (4, or 2 above 32 factors: w and acc are 2 x GPW x LT
// registers); lanes sweep the spots in tiles of 64 staged in LDS
template <int LT, int E, int GPW>
__global__ __launch_bounds__(256) void gene_kernel(PoissonArgs a) {
  __shared__ float sF[E][LT][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t dbase = ((int64_t)blockIdx.x * 4 + wave) * GPW;
  const int64_t per = (int64_t)a.Lt * a.N;
  float w[GPW][LT], acc[GPW][LT];
#pragma unroll
  for (int gq = 0; gq < GPW; ++gq)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      w[gq][l] = (dbase + gq < a.D && l < a.Lt) ? a.W[(dbase + gq) * a.Lt + l] : 0.f;
      acc[gq][l] = 0.f;
    }
  const float inv_e = 1.f / (float)E;
  for (int64_t n0 = 0; n0 < a.N; n0 += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < E * LT * 64; i += 256) {
      const int c = i & 63, el = i >> 6, e = el / LT, l = el - e * LT;
      sF[e][l][c] = (l < a.Lt && n0 + c < a.N) ? a.expF[e * per + (int64_t)l * a.N + n0 + c] : 0.f;
    }
    __syncthreads();
    const int64_t n = n0 + lane;
    if (n < a.N) {
      const float Vn = a.V[n];
      float yv[GPW];
#pragma unroll
      for (int gq = 0; gq < GPW; ++gq) yv[gq] = (dbase + gq < a.D) ? a.y[(dbase + gq) * a.N + n] : 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float f[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) f[l] = sF[e][l][lane];
#pragma unroll
        for (int gq = 0; gq < GPW; ++gq) {
          float z = 0.f;
#pragma unroll
          for (int l = 0; l < LT; ++l) z = fmaf(w[gq][l], f[l], z);
          const float g = (dbase + gq < a.D) ? (yv[gq] * __builtin_amdgcn_rcpf(z) - Vn) * inv_e : 0.f;
#pragma unroll
          for (int l = 0; l < LT; ++l) acc[gq][l] = fmaf(g, f[l], acc[gq][l]);
        }
      }
    }
  }
#pragma unroll
  for (int gq = 0; gq < GPW; ++gq)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      float v = acc[gq][l];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if (lane == 0 && dbase + gq < a.D && l < a.Lt) a.dW[(dbase + gq) * a.Lt + l] = v;
    }
}

// dmean, dscale (Lt,N) and dV (N) from the gene-slice slabs; log-lik total
__global__ __launch_bounds__(256) void poisson_finish_kernel(PoissonArgs a, int nblk_spot) {
  __shared__ double sh[8];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)a.Lt * a.N;
  if (i < per) {
    float dm = 0.f, ds = 0.f;
    for (int e = 0; e < a.E; ++e) {
      float dx = 0.f;
      for (int s = 0; s < a.S; ++s) dx += a.dexp_slab[((int64_t)s * a.E + e) * per + i];
      const float dF = dx * a.expF[e * per + i];
      dm += dF;
      ds += dF * a.eps[e * per + i];
    }
    a.dmean[i] = dm;
    a.dscale[i] = ds;
  }
  if (i < a.N) {
    float dv = 0.f;
    for (int s = 0; s < a.S; ++s) dv += a.dV_slab[(int64_t)s * a.N + i];
    a.dV[i] = dv;
  }
  if (blockIdx.x == 0) {
    double v = 0.0, vg = 0.0;
    for (int j = threadIdx.x; j < a.S * nblk_spot; j += 256) { v += a.ll_slab[j]; vg += a.ll_slab[a.S * nblk_spot + j]; }
    const double t = block_sum_d(v, sh);
    const double tg = block_sum_d(vg, sh);
    if (threadIdx.x == 0) { a.loglik[0] = t; a.loglik[1] = tg; }
  }
}

struct PoissonPlan { int S; int64_t nblk; size_t bytes; float *expF, *dexp, *dVs; double* ll; };

static PoissonPlan poisson_plan(int64_t N, int64_t D, int Lt, int E, void* ws) {
  PoissonPlan pl;
  pl.nblk = (N + 255) / 256;
  // enough (spot block, gene slice) workgroups to fill 256 CUs a few times over
  int64_t S = (2048 + pl.nblk - 1) / pl.nblk;
  if (S < 1) S = 1;
  if (S > 64) S = 64;
  if (S > (D + 63) / 64) S = (D + 63) / 64;
  pl.S = (int)S;
  Carver c(ws);
  pl.expF = c.take<float>((int64_t)E * Lt * N);
  pl.dexp = c.take<float>((int64_t)pl.S * E * Lt * N);
  pl.dVs = c.take<float>((int64_t)pl.S * N);
  pl.ll = c.take<double>((int64_t)2 * pl.S * pl.nblk);
  pl.bytes = c.used();
  return pl;
}

}  // namespace gpz

using namespace gpz;

extern "C" size_t gpz_poisson_nsf_workspace_bytes(int64_t N, int64_t D, int32_t Lt, int32_t E) {
  if (N < 1 || D < 1 || Lt < 1 || Lt > PMAXL || E < 1 || E > PMAXE) return 0;
  return poisson_plan(N, D, Lt, E, nullptr).bytes;
}

extern "C" int gpz_poisson_nsf(const float* mean, const float* scale, const float* eps, const float* W, const float* V,
                               const float* y, int64_t N, int64_t D, int32_t Lt, int32_t E, int32_t with_lgamma,
                               double* loglik, float* dmean, float* dscale, float* dW, float* dV, void* ws,
                               size_t ws_bytes, void* stream) {
  GPZ_REQUIRE(mean && scale && eps && W && V && y && loglik && dmean && dscale && dW && dV && ws,
              "gpz_poisson_nsf: null pointer");
  GPZ_REQUIRE(N >= 1 && D >= 1, "gpz_poisson_nsf: bad extents");
  GPZ_REQUIRE(Lt >= 1 && Lt <= PMAXL, "gpz_poisson_nsf: %d factors unsupported (1..%d)", Lt, PMAXL);
  GPZ_REQUIRE(E >= 1 && E <= PMAXE, "gpz_poisson_nsf: %d samples per call unsupported (1..%d)", E, PMAXE);
  GPZ_REQUIRE(E * ((Lt + 7) / 8 * 8) <= 64, "gpz_poisson_nsf: E * factors = %d x %d exceeds the register budget (64): "
              "call once per group of samples", E, Lt);
  PoissonPlan pl = poisson_plan(N, D, Lt, E, ws);
  GPZ_REQUIRE(ws_bytes >= pl.bytes, "gpz_poisson_nsf: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PoissonArgs a;
  a.mean = mean; a.scale = scale; a.eps = eps; a.W = W; a.V = V; a.y = y;
  a.expF = pl.expF; a.dexp_slab = pl.dexp; a.dV_slab = pl.dVs; a.ll_slab = pl.ll;
  a.dW = dW; a.dmean = dmean; a.dscale = dscale; a.dV = dV; a.loglik = loglik;
  a.N = N; a.D = D; a.Lt = Lt; a.E = E; a.S = pl.S; a.with_lgamma = with_lgamma;
  const int64_t tot = (int64_t)E * Lt * N;
  hipLaunchKernelGGL(expf_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a);
  GPZ_LAUNCH_OK();
  const int gpw = Lt <= 32 ? 4 : 2;
  const dim3 gs((unsigned)pl.nblk, (unsigned)pl.S), gg((unsigned)((D + 4 * gpw - 1) / (4 * gpw)));
#define GPZ_PO(LT, EE)                                                                       \
  do {                                                                                       \
    hipLaunchKernelGGL((spot_kernel<LT, EE>), gs, dim3(256), 0, s, a);                       \
    hipLaunchKernelGGL((gene_kernel<LT, EE, (LT <= 32 ? 4 : 2)>), gg, dim3(256), 0, s, a);   \
  } while (0)
#define GPZ_POE(LT)                                                            \
  do {                                                                         \
    if (E == 1) GPZ_PO(LT, 1); else if (E == 2) GPZ_PO(LT, 2); else if (E == 3) GPZ_PO(LT, 3); else GPZ_PO(LT, 4); \
  } while (0)
#define GPZ_POE2(LT) do { if (E == 1) GPZ_PO(LT, 1); else GPZ_PO(LT, 2); } while (0)
  if (Lt <= 8) GPZ_POE(8); else if (Lt <= 16) GPZ_POE(16); else if (Lt <= 24) GPZ_POE2(24); else if (Lt <= 32) GPZ_POE2(32);
  else if (Lt <= 48) GPZ_PO(48, 1); else GPZ_PO(64, 1);
#undef GPZ_POE2
#undef GPZ_POE
#undef GPZ_PO
  GPZ_LAUNCH_OK();
  const int64_t per = (int64_t)Lt * N;
  hipLaunchKernelGGL(poisson_finish_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, a, (int)pl.nblk);
  GPZ_LAUNCH_OK();
  return 0;
}
