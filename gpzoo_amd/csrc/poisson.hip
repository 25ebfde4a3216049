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
// Work split (fp32, the three dense products on MFMA, fp64 block sums, no atomics -> bitwise reproducible):
//   expf_kernel       expF (E,Lt,N) once (a few MB, L2 resident afterwards)
//   spot_mfma_kernel  pass A: rate tiles with genes as rows -> log-lik, dV and dexpF = W^T G partial slabs
//   gene_mfma_kernel  pass B: the transposed tiles (spots as rows) -> dW = G expF^T partial slabs
//   finish_kernel     sums the slabs into dmean, dscale, dV, dW and the scalar
#include "common.h"

namespace gpz {

constexpr int PMAXL = 64;   // factors (spatial + non-spatial): the notebooks' hybrids run L = 20 spatial + T = 19..20
constexpr int PMAXE = 32;   // Monte-Carlo samples per call (the reference's benchmark notebooks run E = 20)
constexpr int PEG = 4;      // ... of which pass B holds this many exp(F) tiles in LDS at a time (a "sample group")

struct PoissonArgs {
  const float* mean; const float* scale; const float* eps;   // (Lt,N), (Lt,N), (E,Lt,N)
  const float* W; const float* V; const float* y;            // (D,Lt) positive, (N,) positive, (D,N)
  float* expF;                                               // (E,Lt,N) scratch
  float* dexp_slab; float* dV_slab; double* ll_slab;          // [SD][E][Lt][N], [SV][N], [S][nblk * E]
  double* lg_slab; int nlg;                                   // partial sums of lgamma(y + 1), one per workgroup of lgamma_sum_kernel
  float* dW_slab;                                             // [SN][D][Lt]
  float* dW; float* dmean; float* dscale; float* dV; double* loglik;
  int64_t N, D;
  int Lt, E, S, with_lgamma;
  int SD, SV, SN;                                             // slab counts: dexpF, dV, dW
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Both passes put their three dense products -- the rate Z = W expF, dexpF = W^T G and dW = G expF^T with
// G = (y / Z - V) / E -- on v_mfma_f32_16x16x4_f32 and keep the element-wise part (log, reciprocal, the Poisson
// terms) on the accumulator registers in between; nothing but y, W and expF is read and only the gradients are
// written.  The 16 x 16 result of one product is the operand of the next without leaving registers: register g of
// lane (r, q) holds element (row 4q + g, column r), which is exactly what k-step g of an MFMA that sums over the ROW
// index wants as its B operand (k slot q <-> row 4q + g; the A operand is read with the same permutation).  dexpF
// sums G over its gene index and dW over its spot index, so pass A computes Z with genes as rows and pass B computes
// the transposed tile with spots as rows -- each pass then has the index it reduces on the register axis and no
// tile is ever transposed through LDS.  K = factors is padded to 8 (KS k-steps of 4, a template parameter: operand
// fragments live in registers), outputs with factors as rows to 16.
//
// Pass A  grid (E * ceil(N/64), S), 4 waves: the workgroup owns sample e = blockIdx.x % E (consecutive workgroups share
//         a y tile through L2), wave gs every 4th 16-gene group of gene slice s for the 64 spots of the block.  Column sub-tile c of its 16 x 64 tile is the spots 4r + c, so a lane's four
//         accumulator tiles hold four CONSECUTIVE spots of a gene row: y arrives as one 16-byte load per row (256
//         contiguous bytes per gene row and wave).  Emits log-lik, dV and dexpF partial slabs (no atomics).
// Factor counts of the form 16 j + 1 ... 16 j + 4 (KS = 4 j + 1 k-steps; the notebooks' NSF models run 20 factors) pad
// their last 1 .. 4 factor rows to a whole 16-row MFMA tile in the gradient products -- 16 of 52 MFMAs per 16 x 64 tile, 12
// of their 16 rows zeros.  Pass B forms those rows on the vector pipe instead (TAIL: 64 multiply-adds per lane and tile
// as packed ones, partial sums combined across the lane groups once at the end): 0.478 -> 0.42 ms at D = 17 702, N_b = 7000,
// E = 3.  The same in pass A was measured and NOT kept: that pass already issues 4.3 vector instructions per MFMA (log,
// reciprocal and the Poisson terms per element), the extra ones cost more issue time than the 16 MFMAs they replace
// (0.60 -> 0.86 ms with the operands single-buffered to make room for them, 0.72 ms for the single-buffering alone).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void spot_mfma_kernel(PoissonArgs a, int GS) {
  constexpr int LT16 = (KS + 3) / 4;          // 16-row tiles of the factor axis
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int e = blockIdx.x % a.E, gs = wave;
  const int s = blockIdx.y;
  const int64_t n0 = (int64_t)(blockIdx.x / a.E) * 64;
  const int64_t per = (int64_t)a.Lt * a.N;
  const int64_t dper = (a.D + a.S - 1) / a.S;
  const int64_t d_lo = s * dper, d_hi = (d_lo + dper < a.D) ? d_lo + dper : a.D;
  const float inv_e = 1.f / (float)a.E;
  const bool vec = (a.N & 3) == 0;            // rows of y / the slabs start 16-byte aligned
  const bool tile_full = n0 + 64 <= a.N;      // (wave-uniform) every spot of the tile exists
  // this lane's four spots, their V, and the exp(F) operand of the rate product: B[k = factor 4s + q][column = spot 4r + c]
  float vn[4], invv[4], bz[KS][4];
  bool nok[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t n = n0 + 4 * r + c;
    nok[c] = n < a.N;
    vn[c] = nok[c] ? a.V[n] : 1.f;
    invv[c] = __builtin_amdgcn_rcpf(vn[c]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int l = 4 * ks + q;
      bz[ks][c] = (nok[c] && l < a.Lt) ? a.expF[e * per + (int64_t)l * a.N + n] : 0.f;
    }
  }
  f32x4 dacc[LT16][4];
#pragma unroll
  for (int lt = 0; lt < LT16; ++lt)
#pragma unroll
    for (int c = 0; c < 4; ++c) dacc[lt][c] = f32x4{0, 0, 0, 0};
  float dv[4] = {0.f, 0.f, 0.f, 0.f};
  double ll = 0.0;
  // One 16-gene group's operands: W as the A operand of the rate product (rows = genes r), y rows 4q + g (four
  // consecutive spots per lane), and W again as the A operand of dexpF (k slot q <-> gene 4q + g, rows = factors).
  // The next group's loads are issued before the current group is computed (the loop is otherwise bound by the
  // latency of these L2 reads: two waves per SIMD cannot cover it).
  struct Group { float wa[KS]; f32x4 yv[4]; float wt[4][LT16]; };
  // Interior groups (all 16 genes and all 64 spots exist, rows 16-byte aligned) load without a branch: every address is
  // valid, only the padded factor slots (l >= Lt) are zeroed by a select on a clamped index.  The general form guards
  // every access (each guard is an exec-mask branch: they were a third of the loop's scalar instructions).
  auto load_group = [&](int64_t d0, Group& G) __attribute__((always_inline)) {
    if (d0 + 16 <= d_hi && tile_full && vec) {
      const float* wr = a.W + (d0 + r) * a.Lt;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int l = 4 * ks + q;
        const float v = wr[l < a.Lt ? l : a.Lt - 1];
        G.wa[ks] = l < a.Lt ? v : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t dg = d0 + 4 * q + g;
        G.yv[g] = *reinterpret_cast<const f32x4*>(a.y + dg * a.N + n0 + 4 * r);
        const float* wg = a.W + dg * a.Lt;
#pragma unroll
        for (int lt = 0; lt < LT16; ++lt) {
          const int l = 16 * lt + r;
          const float v = wg[l < a.Lt ? l : a.Lt - 1];
          G.wt[g][lt] = l < a.Lt ? v : 0.f;
        }
      }
      return;
    }
    const int64_t d = d0 + r;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int l = 4 * ks + q;
      G.wa[ks] = (d < d_hi && l < a.Lt) ? a.W[d * a.Lt + l] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t dg = d0 + 4 * q + g;
      G.yv[g] = f32x4{0, 0, 0, 0};
      if (dg < d_hi) {
        const float* yp = a.y + dg * a.N + n0 + 4 * r;
        if (vec && nok[3]) G.yv[g] = *reinterpret_cast<const f32x4*>(yp);
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c) if (nok[c]) G.yv[g][c] = yp[c];
        }
      }
#pragma unroll
      for (int lt = 0; lt < LT16; ++lt) {
        const int l = 16 * lt + r;
        G.wt[g][lt] = (dg < d_hi && l < a.Lt) ? a.W[dg * a.Lt + l] : 0.f;
      }
    }
  };
  Group cur, nxt;
  const int64_t dstep = 16 * GS;
  int64_t d0 = d_lo + 16 * gs;
  if (d0 < d_hi) load_group(d0, cur);
  for (; d0 < d_hi; d0 += dstep) {
    if (d0 + dstep < d_hi) load_group(d0 + dstep, nxt);
    // rate tile Z[gene][spot]
    f32x4 z[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      z[c] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) z[c] = mfma4(cur.wa[ks], bz[ks][c], z[c]);
    }
    // element-wise: the Poisson terms, and G (gene rows on the register axis) in place of Z.  The common factor 1 / E
    // of G and dV is applied once at the end (to dexpF and dV).  Interior tiles take the branch-free, mask-free form.
    float llf = 0.f;
    if (d0 + 16 <= d_hi && tile_full) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float zz = z[c][g], y1 = cur.yv[g][c];
          const float rate = vn[c] * zz;
          llf += __builtin_fmaf(y1 * 0.69314718055994530942f, __builtin_amdgcn_logf(rate), -rate);
          dv[c] += __builtin_fmaf(y1, invv[c], -zz);
          z[c][g] = __builtin_fmaf(y1, __builtin_amdgcn_rcpf(zz), -vn[c]);
        }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bool dok = d0 + 4 * q + g < d_hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = dok && nok[c];
          const float zz = ok ? z[c][g] : 1.f, y1 = cur.yv[g][c];
          const float rate = vn[c] * zz;
          llf += ok ? __builtin_fmaf(y1 * 0.69314718055994530942f, __builtin_amdgcn_logf(rate), -rate) : 0.f;
          dv[c] += ok ? __builtin_fmaf(y1, invv[c], -zz) : 0.f;
          z[c][g] = ok ? __builtin_fmaf(y1, __builtin_amdgcn_rcpf(zz), -vn[c]) : 0.f;
        }
      }
    }
    ll += (double)llf;
    // dexpF[factor][spot] += sum over the group's genes: k-step g pairs G's register g (gene 4q + g) with
    // A[row = factor 16 lt + r][k slot q] = W[gene 4q + g][factor 16 lt + r]
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int lt = 0; lt < LT16; ++lt)
#pragma unroll
        for (int c = 0; c < 4; ++c) dacc[lt][c] = mfma4(cur.wt[g][lt], z[c][g], dacc[lt][c]);
    cur = nxt;
  }
  // slabs: dexpF per (slice, gene sub-group), dV per wave; the finish kernel sums them in a fixed order
  const int64_t slab = (int64_t)s * GS + gs;
#pragma unroll
  for (int lt = 0; lt < LT16; ++lt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int l = 16 * lt + 4 * q + g;
      if (l < a.Lt) {
        float* dp = a.dexp_slab + ((slab * a.E + e) * a.Lt + l) * a.N + n0 + 4 * r;
        if (vec && nok[3])
          *reinterpret_cast<f32x4*>(dp) = f32x4{dacc[lt][0][g] * inv_e, dacc[lt][1][g] * inv_e, dacc[lt][2][g] * inv_e, dacc[lt][3][g] * inv_e};
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c) if (nok[c]) dp[c] = dacc[lt][c][g] * inv_e;
        }
      }
    }
  const int nw = a.E * GS;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = dv[c] * inv_e;
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    if (q == 0 && nok[c]) a.dV_slab[((int64_t)s * nw + e * GS + gs) * a.N + n0 + 4 * r + c] = v;
  }
  const double t = block_sum_d(ll * (double)inv_e, sh);
  if (threadIdx.x == 0) a.ll_slab[(int64_t)s * gridDim.x + blockIdx.x] = t;
}

// sum over (gene, spot) of lgamma(y + 1) = log(y!), the parameter-free term of the Poisson log-density: its own pass
// over y (one more read of the counts: 0.1 ms at Slide-seq size) instead of a branch per element inside pass A, where the
// library function's inlined code (the non-table case) made the loop seven vector instructions per MFMA (rocprofv3 PMC,
// round 4: pass A 0.755 -> 0.646 ms without it; summed in pass B instead, where a y tile serves all samples, it cost
// 0.18 ms).  A table for integer y < 256, lgammaf for anything else; one partial sum per workgroup.
__global__ __launch_bounds__(256) void lgamma_sum_kernel(PoissonArgs a) {
  __shared__ double sh[8];
  __shared__ float lfact[256];
  for (int i = threadIdx.x; i < 256; i += 256) lfact[i] = lgammaf((float)i + 1.f);
  __syncthreads();
  const int64_t tot = a.D * a.N;
  auto term = [&](float y1) {
    const int yi = (int)y1;
    return (y1 == (float)yi && yi >= 0 && yi < 256) ? lfact[yi] : lgammaf(y1 + 1.f);
  };
  double lg = 0.0;
  float part = 0.f;
  int cnt = 0;
  const int64_t tot4 = ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) ? tot / 4 : 0;     // 16-byte loads when aligned
  const f32x4* y4 = reinterpret_cast<const f32x4*>(a.y);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot4; i += (int64_t)a.nlg * 256) {
    const f32x4 v = y4[i];
    part += (term(v[0]) + term(v[1])) + (term(v[2]) + term(v[3]));
    if (++cnt == 16) { lg += (double)part; part = 0.f; cnt = 0; }      // fp32 partial sums of at most 64 terms
  }
  for (int64_t i = 4 * tot4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)a.nlg * 256) part += term(a.y[i]);
  lg += (double)part;
  const double tg = block_sum_d(lg, sh);
  if (threadIdx.x == 0) a.lg_slab[blockIdx.x] = tg;
}

// Pass B  grid (ceil(D/64), SN), 4 waves: wave w owns the 16 genes 64 b + 16 w and sweeps spot slice sn in tiles of 64
//         (exp(F) and V of the tile staged in LDS).  The tile is computed TRANSPOSED, spots as rows: row 4q + g of
//         column sub-tile c is spot 16q + 4c + g, so again a lane's accumulators hold consecutive spots of one gene and
//         y arrives as four 16-byte loads per lane (256 contiguous bytes per gene).  dW^T[factor][gene] stays in 4 * LT16
//         registers for the whole sweep.
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void gene_mfma_kernel(PoissonArgs a, int SN) {
  constexpr bool TAIL = (KS % 4) == 1;                               // the last 1 .. 4 factors on the vector pipe (see pass A)
  constexpr int LTM = TAIL ? KS / 4 : (KS + 3) / 4, LTA = LTM > 0 ? LTM : 1, L0 = 16 * LTM;
  constexpr int LT16 = (KS + 3) / 4, LP = 16 * LT16, PF = 68;      // factor rows staged (zero padded), row pitch
  extern __shared__ float smem_p[];
  // Samples go through LDS in groups of EG <= PEG: the spot tile's y stays in registers across the groups, so y is
  // read once per pass for ANY number of samples (E = 20 in the reference's benchmarks).
  const int EG = a.E < PEG ? a.E : PEG, NG = (a.E + EG - 1) / EG;
  const int FB = EG * LP * PF;                  // one buffer of exp(F): [EG][LP][PF]
  float* sF = smem_p;                           // [2][EG][LP][PF]
  float* sV = smem_p + 2 * FB;                  // [2][64]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int64_t d0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  const int64_t dcol = d0 + r;                  // this lane's gene (a column of the transposed tile)
  const bool dok = dcol < a.D;
  const bool d_full = d0 + 16 <= a.D;           // (wave-uniform) every gene of this wave exists
  const int64_t per = (int64_t)a.Lt * a.N;
  const int64_t ntiles = (a.N + 63) / 64, tper = (ntiles + SN - 1) / SN;
  const int64_t t_lo = blockIdx.y * tper, t_hi = (t_lo + tper < ntiles) ? t_lo + tper : ntiles;
  const float inv_e = 1.f / (float)a.E;
  const bool vec = (a.N & 3) == 0;
  float wb[KS];                                 // B[k = factor 4s + q][column = gene r]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int l = 4 * ks + q;
    wb[ks] = (dok && l < a.Lt) ? a.W[dcol * a.Lt + l] : 0.f;
  }
  f32x4 dwt[LTA];
#pragma unroll
  for (int lt = 0; lt < LTA; ++lt) dwt[lt] = f32x4{0, 0, 0, 0};
  f32x2 tw[4];                                  // TAIL: [factor L0 + t]: two partial sums over this lane's spots of gene r
#pragma unroll
  for (int t = 0; t < 4; ++t) tw[t] = f32x2{0.f, 0.f};
  const int arow = 16 * (r >> 2) + (r & 3);     // A row r of sub-tile c is spot 16 (r >> 2) + 4c + (r & 3) of the tile
  // exp(F) and V tiles arrive by LDS-DMA (one 256-byte row of 64 spots per wave instruction, no registers), double
  // buffered: (tile, sample group) s + 1 travels while s is computed.  Reads past the end of a row's valid spots return
  // the next row's (finite) values or, past the array, zero: those spots are masked below.  The padded factor rows are
  // zeroed once.
  typedef __attribute__((address_space(3))) void lds_void;
  for (int i = threadIdx.x; i < 2 * FB; i += 256) sF[i] = 0.f;
  __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t f_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.expF), 0, (int)((int64_t)a.E * per * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.V), 0, (int)(a.N * sizeof(float)), 0x00020000);
#endif
  auto stage = [&](int64_t tt, int gi, int buf) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int64_t n0 = tt * 64;
    const int e0 = gi * EG, ne = (a.E - e0 < EG) ? a.E - e0 : EG;
    // (rows dealt to the waves per sample: a flat index over (sample, factor) needs an integer division per row, and
    // scalar division is ~40 instructions: they were most of this kernel's 3.4 scalar instructions per MFMA)
    for (int el = 0; el < ne; ++el)
      for (int l = wave; l < a.Lt; l += 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(f_rsrc, (lds_void*)(sF + buf * FB + (el * LP + l) * PF), 4, lane * 4,
                                                 (int)(((e0 + el) * per + (int64_t)l * a.N + n0) * 4), 0, 0);
    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(sV + buf * 64), 4, lane * 4, (int)(n0 * 4), 0, 0);
#endif
  };
  auto load_y = [&](int64_t tt, f32x4 (&yv)[4]) __attribute__((always_inline)) {
    const int64_t n0 = tt * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t n = n0 + 16 * q + 4 * c;
      yv[c] = f32x4{0, 0, 0, 0};
      if (dok) {
        const float* yp = a.y + dcol * a.N + n;
        if (vec && n + 3 < a.N) yv[c] = *reinterpret_cast<const f32x4*>(yp);
        else {
#pragma unroll
          for (int g = 0; g < 4; ++g) if (n + g < a.N) yv[c][g] = yp[g];
        }
      }
    }
  };
  f32x4 yv[4], yn[4];
  if (t_lo < t_hi) { stage(t_lo, 0, 0); load_y(t_lo, yv); }
  int buf = 0;
  for (int64_t tt = t_lo; tt < t_hi; ++tt) {
    const int64_t n0 = tt * 64;
    for (int gi = 0; gi < NG; ++gi, buf ^= 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                          // (tile, group) has landed; every wave is done with the other buffer
      if (gi + 1 < NG) stage(tt, gi + 1, buf ^ 1);
      else if (tt + 1 < t_hi) { stage(tt + 1, 0, buf ^ 1); load_y(tt + 1, yn); }
      f32x4 vv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) vv[c] = *reinterpret_cast<const f32x4*>(sV + buf * 64 + 16 * q + 4 * c);
      const int ne = (a.E - gi * EG < EG) ? a.E - gi * EG : EG;
      for (int e = 0; e < ne; ++e) {
        const float* fe = sF + buf * FB + e * LP * PF;
        f32x4 z[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          z[c] = f32x4{0, 0, 0, 0};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) z[c] = mfma4(fe[(4 * ks + q) * PF + arow + 4 * c], wb[ks], z[c]);
        }
        // G^T in place of Z^T (its common factor 1 / E is applied to dW at the end); interior tiles without masks
        if (d_full && n0 + 64 <= a.N) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) z[c][g] = __builtin_fmaf(yv[c][g], __builtin_amdgcn_rcpf(z[c][g]), -vv[c][g]);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const bool ok = dok && n0 + 16 * q + 4 * c + g < a.N;
              const float zz = ok ? z[c][g] : 1.f;
              z[c][g] = ok ? __builtin_fmaf(yv[c][g], __builtin_amdgcn_rcpf(zz), -vv[c][g]) : 0.f;
            }
        }
        // dW^T[factor][gene] += sum over the tile's spots: k-step (c, g) pairs G^T's register g of sub-tile c (spot
        // 16q + 4c + g) with A[row = factor 16 lt + r][k slot q] = expF[factor][that spot]
#pragma unroll
        for (int lt = 0; lt < LTM; ++lt)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(fe + (16 * lt + r) * PF + 16 * q + 4 * c);
#pragma unroll
            for (int g = 0; g < 4; ++g) dwt[lt] = mfma4(fa[g], z[c][g], dwt[lt]);
          }
        if constexpr (TAIL) {
          // the last four factors: G^T's registers (spots 16q + 4c + g of gene r) times exp(F)[L0 + t][those spots]
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4 fv = *reinterpret_cast<const f32x4*>(fe + (L0 + t) * PF + 16 * q + 4 * c);
              tw[t] = pkfma(f32x2{z[c][0], z[c][1]}, f32x2{fv[0], fv[1]}, tw[t]);
              tw[t] = pkfma(f32x2{z[c][2], z[c][3]}, f32x2{fv[2], fv[3]}, tw[t]);
            }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) yv[c] = yn[c];
  }
  if constexpr (TAIL) {       // factor L0 + t of gene r: sum the spot groups (lanes r, r + 16, ...), lanes q == 0 write
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = (tw[t][0] + tw[t][1]) * inv_e;
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      if (q == 0 && dok && L0 + t < a.Lt) a.dW_slab[((int64_t)blockIdx.y * a.D + dcol) * a.Lt + L0 + t] = v;
    }
  }
  if (dok) {
#pragma unroll
    for (int lt = 0; lt < LTM; ++lt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int l = 16 * lt + 4 * q + g;
        if (l < a.Lt) a.dW_slab[((int64_t)blockIdx.y * a.D + dcol) * a.Lt + l] = dwt[lt][g] * inv_e;
      }
  }
}

// dmean, dscale (Lt,N), dV (N) and dW (D,Lt) from the slabs (fixed summation order); log-lik total
__global__ __launch_bounds__(256) void poisson_finish_kernel(PoissonArgs a, int nblk_spot) {
  __shared__ double sh[8];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)a.Lt * a.N;
  if (i < per) {
    float dm = 0.f, ds = 0.f;
    for (int e = 0; e < a.E; ++e) {
      float dx = 0.f;
      for (int s = 0; s < a.SD; ++s) dx += a.dexp_slab[((int64_t)s * a.E + e) * per + i];
      const float dF = dx * a.expF[e * per + i];
      dm += dF;
      ds += dF * a.eps[e * per + i];
    }
    a.dmean[i] = dm;
    a.dscale[i] = ds;
  }
  if (i < a.N) {
    float dv = 0.f;
    for (int s = 0; s < a.SV; ++s) dv += a.dV_slab[(int64_t)s * a.N + i];
    a.dV[i] = dv;
  }
  const int64_t dl = a.D * a.Lt;
  for (int64_t j = i; j < dl; j += (int64_t)gridDim.x * 256) {
    float w = 0.f;
    for (int s = 0; s < a.SN; ++s) w += a.dW_slab[(int64_t)s * dl + j];
    a.dW[j] = w;
  }
  if (blockIdx.x == 0) {
    double v = 0.0, vg = 0.0;
    for (int j = threadIdx.x; j < a.S * nblk_spot; j += 256) v += a.ll_slab[j];
    for (int j = threadIdx.x; j < a.nlg; j += 256) vg += a.lg_slab[j];
    const double t = block_sum_d(v, sh);
    const double tg = block_sum_d(vg, sh);
    if (threadIdx.x == 0) { a.loglik[0] = t; a.loglik[1] = tg; }
  }
}

struct PoissonPlan { int S, GS, SN; int64_t nblk; size_t bytes; float *expF, *dexp, *dVs, *dWs; double *ll, *lg; int nlg; };

static PoissonPlan poisson_plan(int64_t N, int64_t D, int Lt, int E, void* ws) {
  PoissonPlan pl;
  pl.nblk = (N + 63) / 64;                     // spot tiles of pass A
  pl.GS = 4;                                   // 16-gene sub-groups (waves) per workgroup
  // pass A: enough (spot tile, gene slice) workgroups to fill 256 CUs a few times over, slices of >= 8 gene groups per wave
  int64_t S = (1536 + pl.nblk * E - 1) / (pl.nblk * E);
  const int64_t smax = (D + 128 * pl.GS - 1) / (128 * pl.GS);
  if (S > smax) S = smax;
  if (S > 32) S = 32;
  if (S < 1) S = 1;
  pl.S = (int)S;
  // pass B: gene blocks of 64 x spot slices
  const int64_t gblk = (D + 63) / 64;
  int64_t SN = (1024 + gblk - 1) / gblk;
  if (SN > (N + 511) / 512) SN = (N + 511) / 512;
  if (SN > 16) SN = 16;
  if (SN < 1) SN = 1;
  pl.SN = (int)SN;
  Carver c(ws);
  pl.expF = c.take<float>((int64_t)E * Lt * N);
  pl.dexp = c.take<float>((int64_t)pl.S * pl.GS * E * Lt * N);
  pl.dVs = c.take<float>((int64_t)pl.S * E * pl.GS * N);
  pl.dWs = c.take<float>((int64_t)pl.SN * D * Lt);
  pl.ll = c.take<double>((int64_t)pl.S * pl.nblk * E);
  pl.nlg = 4096;
  pl.lg = c.take<double>(pl.nlg);
  pl.bytes = c.used();
  return pl;
}

}  // namespace gpz

using namespace gpz;

extern "C" size_t gpz_poisson_nsf_workspace_bytes(int64_t N, int64_t D, int32_t Lt, int32_t E) {
  if (N < 1 || D < 1 || Lt < 1 || Lt > PMAXL || E < 1 || E > PMAXE) return 0;
  return poisson_plan(N, D, Lt, E, nullptr).bytes;
}

template <int KS>
static int poisson_passes(const PoissonArgs& a, const PoissonPlan& pl, hipStream_t s) {
  constexpr int LP = 16 * ((KS + 3) / 4);
  const size_t lds = sizeof(float) * 2 * ((size_t)(a.E < PEG ? a.E : PEG) * LP * 68 + 64);
  // The opt-in above 64 KB is per kernel function and device and is set ONCE, so it is set to what the kernel can need
  // at most (a full sample group), not to this call's size: a later call with more samples must still fit under it.
  constexpr size_t lds_max = sizeof(float) * 2 * ((size_t)PEG * LP * 68 + 64);
  static_assert(lds_max <= 160 * 1024, "gene_mfma_kernel: LDS of the largest sample group");
  if (lds_max > 64 * 1024) {
    static bool set[64] = {};
    int dev = 0;
    GPZ_HIP_OK(hipGetDevice(&dev));
    if (!set[dev & 63]) {
      GPZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(gene_mfma_kernel<KS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
      set[dev & 63] = true;
    }
  }
  hipLaunchKernelGGL((spot_mfma_kernel<KS>), dim3((unsigned)(pl.nblk * a.E), (unsigned)pl.S), dim3(64 * pl.GS), 0, s, a, pl.GS);
  GPZ_LAUNCH_OK();
  if (a.with_lgamma) hipLaunchKernelGGL(lgamma_sum_kernel, dim3((unsigned)a.nlg), dim3(256), 0, s, a);
  else GPZ_HIP_OK(hipMemsetAsync(a.lg_slab, 0, sizeof(double) * a.nlg, s));
  GPZ_LAUNCH_OK();
  hipLaunchKernelGGL((gene_mfma_kernel<KS>), dim3((unsigned)((a.D + 63) / 64), (unsigned)pl.SN), dim3(256), lds, s, a, pl.SN);
  GPZ_LAUNCH_OK();
  return 0;
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
  // gene_mfma_kernel addresses exp(F) through a buffer descriptor with 32-bit byte extents and offsets
  GPZ_REQUIRE((int64_t)E * Lt * N * 4 < (1ll << 31),
              "gpz_poisson_nsf: E * Lt * N = %lld elements of exp(F) exceed the 2 GiB a call can address: split N",
              (long long)((int64_t)E * Lt * N));
  PoissonPlan pl = poisson_plan(N, D, Lt, E, ws);
  GPZ_REQUIRE(ws_bytes >= pl.bytes, "gpz_poisson_nsf: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PoissonArgs a;
  a.mean = mean; a.scale = scale; a.eps = eps; a.W = W; a.V = V; a.y = y;
  a.expF = pl.expF; a.dexp_slab = pl.dexp; a.dV_slab = pl.dVs; a.ll_slab = pl.ll; a.dW_slab = pl.dWs;
  a.lg_slab = pl.lg; a.nlg = pl.nlg;
  a.dW = dW; a.dmean = dmean; a.dscale = dscale; a.dV = dV; a.loglik = loglik;
  a.N = N; a.D = D; a.Lt = Lt; a.E = E; a.S = pl.S; a.with_lgamma = with_lgamma;
  a.SD = pl.S * pl.GS; a.SV = pl.S * E * pl.GS; a.SN = pl.SN;
  const int64_t tot = (int64_t)E * Lt * N;
  hipLaunchKernelGGL(expf_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a);
  GPZ_LAUNCH_OK();
  // KS k-steps of 4 factors (operand fragments live in registers: a template parameter); up to 40 factors -- the
  // notebooks' hybrids run 20 spatial + 19..20 non-spatial -- exactly, beyond that padded to a multiple of 8
  int rc;
  switch ((Lt + 3) / 4) {
    case 1: rc = poisson_passes<1>(a, pl, s); break;
    case 2: rc = poisson_passes<2>(a, pl, s); break;
    case 3: rc = poisson_passes<3>(a, pl, s); break;
    case 4: rc = poisson_passes<4>(a, pl, s); break;
    case 5: rc = poisson_passes<5>(a, pl, s); break;
    case 6: rc = poisson_passes<6>(a, pl, s); break;
    case 7: rc = poisson_passes<7>(a, pl, s); break;
    case 8: rc = poisson_passes<8>(a, pl, s); break;
    case 9: rc = poisson_passes<9>(a, pl, s); break;
    case 10: rc = poisson_passes<10>(a, pl, s); break;
    case 11: case 12: rc = poisson_passes<12>(a, pl, s); break;
    case 13: case 14: rc = poisson_passes<14>(a, pl, s); break;
    default: rc = poisson_passes<16>(a, pl, s); break;
  }
  if (rc) return rc;
  const int64_t per = (int64_t)Lt * N;
  hipLaunchKernelGGL(poisson_finish_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, a, (int)(pl.nblk * E));
  GPZ_LAUNCH_OK();
  return 0;
}
