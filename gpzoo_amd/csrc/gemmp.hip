// gemmp.hip -- the covariance fill and both forward products on ONE column panel held in LDS, for few inducing points
// (Mp <= 512, fp32).
//
//   fill     Kzx = k(Z, X)        (gp.py:255; kernels.py forward)        computed in the kernel where cov.h covers the family
//   stage 1  Wt = Linv * Kzx      (gp.py:276)                             A lower triangular, B = the Kzx panel
//   stage 2  colsum((LuE^T Wt)^2) (gp.py:280-296 / utilities.py:382-397)  A upper triangular, B = the Wt panel
//
// At M = 512 a row tile of the wide-tile kernels (gemmw.hip) has 8 .. 32 steps: its epilogue (store of the Wt tile, 9 %),
// the operand traffic of a B panel that every row tile reads again (6 %) and the barrier of every step are first-order
// terms there (configs[1] stage 1 at 0.74 of the fp32 MFMA peak).  Here a workgroup owns a whole
// panel -- all Mp rows of 64 columns of one latent -- from the covariance to the column statistics:
//   * the panel sits in LDS as [column][k] (k contiguous: one ds_read_b128 is a lane's four k values of a 16-deep chunk),
//     the four 16-byte slots of every chunk XOR-permuted per column so that ds_read_b128's lane groups are conflict-free;
//   * Mp / 32 waves (16 at Mp = 512); wave w owns the 32 rows of one row block for BOTH stages: its A fragments come
//     straight from L2 into registers (no other wave needs them: nothing is staged, and there is NO barrier inside a stage
//     -- the panel is read-only), one 16-deep chunk ahead of the MFMAs that use them;
//   * stage 1's result stays in the accumulators (32 rows x 64 columns per wave), gives colsum(Wt^2) and muE^T Wt from
//     there, and overwrites the Kzx panel in LDS (all waves are through stage 1 by then) as stage 2's B operand; Wt
//     reaches memory only when the caller retains it for the backward pass (copied out of LDS by waves that are idle);
//   * row block r has r + 1 units of k in stage 1 and nb - r in stage 2; the blocks are dealt to the waves so that every
//     SIMD carries the same sum in both stages and its longest waves are of similar length (row_block_of below);
//   * the waves of the upper half of the row blocks have the short stage 2: they prepare the NEXT panel meanwhile, into the
//     registers their dead accumulators free -- computing its covariance values (cov.h: the fill kernel's own arithmetic,
//     same bits; no fill launch, no Kzx in memory) or, for other kernel families, loading them from the fill's Kzx;
//   * the latents of a launch are dealt to the XCDs (blocks b, b + 8, ... share one): an XCD's workgroups stream the same
//     Linv / LuE^T (1 MB each at Mp = 512) out of its 4 MB L2.
// k order, MFMA order and the values of Wt are those of gemmw.hip / gemm.hip (lane group q owns k = 4q .. 4q+3 of a 16-deep
// chunk, chunks ascending from k = 0, zero sub-tiles of the diagonal block skipped): the retained Wt is bitwise the same.
// Measured: configs[1] (N=50k, M=512, L=8) 1.69 ms for fill + both products = 0.79 of the fp32 MFMA
// peak, evaluation 2.32 -> 2.18 ms against fill + two tile launches; M = 256 / 384: -20 % / -23 %.
#include "gemmp.h"

#include "cov.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

// Timing-only diagnostics (WRONG results): -DGPZ_P_ABL=4 builds the kernel without its A-fragment loads, 8 without its B-fragment reads; at run time
// GPZ_PANEL_DBG=1 loads the Kzx panel once per workgroup, =2 deals the row blocks to the waves in order (no SIMD pairing),
// =8 gives every wave the k range of the middle row block (equal durations), =16 swaps the A operands of the two stages,
// =32 deals the row blocks as the first version did (s, 7 - s, 8 + s, 15 - s), =64 no raised priority for the fetching waves.
#ifndef GPZ_P_ABL
#define GPZ_P_ABL 0
#endif

namespace gpz {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PanelParams {
  const float* Linv; const float* LuT;      // (L, Mp, Mp)
  const float* Kzx;                         // (L, Mp, ncp); GEN: unused
  const float* Z; const float* X;           // GEN: (M, D) inducing points, (nreal, D) this chunk's spots
  const float* sigma; const float* ell;     // GEN: (L,)
  int M, nreal;                             // GEN: real rows / columns (the rest of the panel is zero, as the fill writes it)
  float* Wt;                                // (L, Mp, ncp) or null
  const float* muE;                         // (L, Mp)
  float* ps1; float* pm1; float* ps2;       // [L][Mp / 128][ncp]
  int Mp, ncp, L, npan, units;              // panels per latent, units = L * npan
  unsigned long long* stamps;               // debug (gpz_debug_panel_stamps): workgroup 0's phase times, [panel][16 waves][8]
  int dbg;                                  // timing diagnostics (GPZ_PANEL_DBG, listed above; wrong results)
};

constexpr int P_TN = 64;                    // columns of a panel

// which 32-row block wave w takes.  A workgroup's waves go to the SIMDs in a cyclic order of period 4, so waves w, w + 4,
// w + 8, w + 12 share one; block r has r + 1 units of k in stage 1 and nb - r in stage 2.  The blocks are dealt so that every
// SIMD carries the same sum in both stages AND its two longest waves are of nearly equal length: the longest wave of a SIMD
// ends the stage alone, and a wave alone does not keep the matrix pipe full (its own loads and waits show).
//   16 blocks: {0, 3, 12, 15}, {1, 2, 13, 14}, {4, 5, 10, 11}, {6, 7, 8, 9}: 34 units each, at most 3 of them alone
//              (s, 7 - s, 8 + s, 15 - s, the first dealing, left 7 alone on SIMD 0: configs[1] +1.3 %, M = 384 +4.4 %;
//              {0, 1, 14, 15}, {2, 3, 12, 13}, ... leaves 1 alone but two waves for 26 of the 34: no better than the first)
//   12 blocks: {0, 5, 11}, {1, 6, 10}, {3, 4, 9}, {2, 7, 8}: 19 / 20 / 19 / 20 units (s, 7 - s, 8 + s: 18 ... 21)
template <int NB> __device__ __forceinline__ int row_block_of(int w) {
  if constexpr (NB == 16) { constexpr int t[16] = {0, 1, 4, 6, 3, 2, 5, 7, 12, 13, 10, 8, 15, 14, 11, 9}; return t[w]; }
  else if constexpr (NB == 12) { constexpr int t[12] = {0, 1, 3, 2, 5, 6, 4, 7, 11, 10, 9, 8}; return t[w]; }
  else if constexpr (NB == 8) return w < 4 ? w : 11 - w;                                //  s, 7 - s (two workgroups per CU)
  else return w;
}

// GEN: 0 the Kzx panel is read from memory (the stand-alone fill wrote it); 1 + 2 * KIND + (D - 1): the fetching waves
// compute it themselves (cov.h: the fill's own arithmetic, same bits) -- fp32 RBF (KIND 0) / Matern-3/2 (KIND 1) on 1-D / 2-D
// inputs -- and Kzx never exists.
template <int NB, bool STORE, int GEN>
__global__ __launch_bounds__(64 * NB) void panel_kernel(const PanelParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // Panel image: column c at c * PK, k contiguous, the four 16-byte slots of every 16-deep chunk XOR-permuted by column
  // (slot s of column c sits at s ^ g(c), g = [0, 2, 3, 1][(c >> 2) & 3]).  PK = 16 * odd puts columns c .. c + 3 on the
  // four quarters of the 256-byte bank row; with the permutation every lane group of a ds_read_b128 -- lanes {0-3, 12-15,
  // 20-27} etc.: columns r and k slots q mixed -- covers sixteen distinct 16-byte bank slots (no conflict).
  constexpr int Mp = 32 * NB, PK = Mp + 16;
  constexpr int MI = 2;                          // 16-row sub-tiles of a wave's 32 rows
  float* const P = smem;                         // [64][PK]: Kzx panel, then Wt panel
  float* const red = smem + P_TN * PK;           // [3][NB][64]: per-wave column statistics
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int rb = (p.dbg & 2) ? wave : (p.dbg & 32) ? (NB == 16 ? (wave < 4 ? wave : wave < 8 ? 11 - wave : wave < 12 ? wave : 27 - wave) : NB == 12 ? (wave < 4 ? wave : wave < 8 ? 11 - wave : wave) : row_block_of<NB>(wave)) : row_block_of<NB>(wave);
  constexpr int NT = 64 * NB;
  auto gperm = [](int c) { const int t = (c >> 2) & 3; return (((t >> 1) ^ t) & 1) << 1 | (t >> 1); };

  // units (latent, panel) in latent-major order; XCD x takes the contiguous range [x U / 8, (x + 1) U / 8)
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int u_lo = (int)((int64_t)x * p.units / 8), u_hi = (int)((int64_t)(x + 1) * p.units / 8);
  if (u_lo + j >= u_hi) return;

  // this lane's A rows: row block rb, sub-tile mi, row r; its k offset 4q inside a chunk
  int a_voff[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) a_voff[mi] = ((rb * 32 + mi * 16 + r) * Mp + 4 * q) * 4;
  // B fragment: column ni * 16 + r, k = k0 + 4q .. 4q + 3 (slot q of the chunk)
  const float* const bfrag = P + r * PK + ((q ^ gperm(r)) << 2);
  const int kd = ((p.dbg & 8) ? NB / 2 - 1 : rb) * 32;        // first k of the wave's diagonal block

  int n_it = 0;
  auto stamp = [&](int ph) __attribute__((always_inline)) {
    if (p.stamps && blockIdx.x == 0 && n_it < 16 && lane == 0)
      p.stamps[(n_it * 16 + wave) * 8 + ph] = __builtin_readcyclecounter();
  };

  using std::integral_constant;
  using i0 = integral_constant<int, 0>;
  using i1 = integral_constant<int, 1>;

  // The Kzx panel travels memory -> registers -> LDS (transposed).  It is fetched by the waves of the upper half of the row
  // blocks: their stage 2 is the short one, so they are through with it (accumulators and fragments dead: 64 free
  // registers) in the first half of the workgroup's stage 2 and would only wait at its barrier -- the next panel's HBM
  // latency passes there.  Fetching thread ft takes columns 4 c4 .. 4 c4 + 3 of the k rows k_of, k_of + 2 NB, ...: one
  // per-lane offset, the row step in the scalar offset.
  constexpr int NR = 16;                         // float4 per fetching thread: Mp * 64 * 4 bytes / (NT / 2 threads)
  constexpr int RP = 2 * NB;                     // k rows per pass of the NT / 2 fetching threads
  f32x4 pv[NR];
  const bool fetcher = rb >= NB / 2;
  const int ft = (rb - NB / 2) * 64 + lane;
  const int pc4 = ft & 15, pk_of = ft >> 4;
  auto panel_fetch = [&](int l, int64_t col0) __attribute__((always_inline)) {
    if constexpr (GEN == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
      const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.Kzx + (int64_t)l * Mp * p.ncp + col0), 0, (Mp - 1) * p.ncp * 4 + P_TN * 4, 0x00020000);
      const int voff = (pk_of * p.ncp + 4 * pc4) * 4, sstep = RP * p.ncp * 4;
#pragma unroll
      for (int i = 0; i < NR; ++i)
        pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kr, voff, i * sstep, 0));
#endif
    } else {
      // Computed: a fetching wave takes sixteen groups of four consecutive k (inducing points: wave-uniform)
      // for all 64 columns (one spot per lane, in registers) -- a lane's four values are one 16-byte LDS write.  Padded rows
      // and columns are exactly zero, as the stand-alone fill writes them.
      constexpr int KIND = (GEN - 1) >> 1, D = ((GEN - 1) & 1) + 1;
      const CovConst cc = cov_const<KIND>(p.sigma[l], p.ell[l]);
      const int64_t n = col0 + lane;
      const bool real = n < p.nreal;
      float xc[D];
#pragma unroll
      for (int k = 0; k < D; ++k) xc[k] = real ? p.X[n * D + k] : 0.f;
      // the wave's 64 inducing points: lane 4 i + e holds point k(i, e) (one vector load per coordinate), handed to all lanes
      // by v_readlane when its turn comes (64 scalar loads would take 128 SGPRs at D = 2: they spill)
      const int fw = rb - NB / 2;
      const int kl = 4 * (fw + (NB / 2) * (lane >> 2)) + (lane & 3);
      float zl[D];
#pragma unroll
      for (int kk = 0; kk < D; ++kk) zl[kk] = kl < p.M ? p.Z[kl * D + kk] : 0.f;
      // padding as a factor (1 or 0, exact; the values are finite and non-negative: 0 * v = +0 as the fill writes it) rather
      // than 64 lane-mask predicates, which would live in 128 SGPRs
      const float rowf = kl < p.M ? 1.f : 0.f, colf = real ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float z[D];
#pragma unroll
          for (int kk = 0; kk < D; ++kk) z[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zl[kk]), 4 * i + e));
          const float keep = colf * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rowf), 4 * i + e));
          pv[i][e] = keep * cov_value<KIND>(cov_radial<KIND>(cov_d2<D>(z, xc)), cc.amp, cc.c0, cc.c1);
        }
      }
    }
  };
  auto panel_write = [&]() __attribute__((always_inline)) {
    if constexpr (GEN == 0) {
      int kof = pk_of;
      asm volatile("" : "+v"(kof));              // per panel: keeps the store addresses from being hoisted out of the unit loop (and spilled)
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int k = i * RP + kof;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * pc4 + e;
          P[c * PK + (k & ~15) + ((((k >> 2) & 3) ^ gperm(c)) << 2) + (k & 3)] = pv[i][e];
        }
      }
    } else {
      int base = lane * PK;
      asm volatile("" : "+v"(base));
      const int gs = gperm(lane), fw = rb - NB / 2;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int kq = fw + (NB / 2) * i;        // group of four k: slot kq & 3 of chunk kq >> 2
        *reinterpret_cast<f32x4*>(P + base + ((kq >> 2) << 4) + (((kq & 3) ^ gs) << 2)) = pv[i];
      }
    }
  };

  f32x4 acc[MI][4];
  f32x4 fa0[MI], fb0[4], fa1[MI], fb1[4];
  // One 16-deep chunk: MFMAs over the sub-tiles LO .. HI of the wave's 32 rows with the fragments in (fa, fb)
  auto mma = [&](const f32x4 (&fa)[MI], const f32x4 (&fb)[4], auto lo_c, auto hi_c) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int mi = LO; mi <= HI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi][jj], fb[ni][jj], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // A fragments straight from L2: WHICH = 1 Linv, 2 LuE^T of latent l
  auto load_a = [&](auto which_c, int l, f32x4 (&fa)[MI], int k0, auto lo_c, auto hi_c) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    if (GPZ_P_ABL & 4) return;
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(((decltype(which_c)::value == 1) != ((p.dbg & 16) != 0) ? p.Linv : p.LuT) + (int64_t)l * Mp * Mp), 0, Mp * Mp * 4, 0x00020000);
#pragma unroll
    for (int mi = LO; mi <= HI; ++mi)
      fa[mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ar, a_voff[mi], k0 * 4, 0));
#endif
  };
  auto load_b = [&](f32x4 (&fb)[4], int k0) __attribute__((always_inline)) {
    if (GPZ_P_ABL & 8) return;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fb[ni] = *reinterpret_cast<const f32x4*>(bfrag + ni * 16 * PK + k0);
  };
  // the fragments of the NEXT chunk are requested before this chunk's MFMAs and stay there (hipcc otherwise sinks the
  // loads to their first use and the wave waits out their latency with the matrix pipe empty)
  auto load = [&](auto which_c, int l, f32x4 (&fa)[MI], f32x4 (&fb)[4], int k0, auto lo_c, auto hi_c) __attribute__((always_inline)) {
    load_a(which_c, l, fa, k0, lo_c, hi_c);
    load_b(fb, k0);
    __builtin_amdgcn_sched_barrier(0);
  };
  const i1 a1s{};
  const integral_constant<int, 2> a2s{};

  int u = u_lo + j;
  int l = u / p.npan, pn = u - l * p.npan;
#pragma unroll 1
  for (;;) {
    const int64_t col0 = (int64_t)pn * P_TN;
    const int un = u + per;
    stamp(0);
    if (u == u_lo + j) {                         // the workgroup's first panel; later ones are on their way
      if (fetcher) panel_fetch(l, col0);
      else {
#pragma unroll
        for (int i = 0; i < NR; ++i) pv[i] = f32x4{0, 0, 0, 0};
      }
    }
    load_a(a1s, l, fa0, 0, i0{}, i1{});          // stage 1's first A chunk
    __builtin_amdgcn_sched_barrier(0);
    if (fetcher && (!(p.dbg & 1) || u == u_lo + j)) panel_write();
    stamp(1);
    __syncthreads();
    stamp(2);

    // ---------------- stage 1: rows of block rb, k = 0 .. kd + 31 (the last 32: the diagonal block) ----------------
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};
    load_b(fb0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int k = 0; k < kd; k += 32) {           // full chunks, two per trip (kd is a multiple of 32)
      load(a1s, l, fa1, fb1, k + 16, i0{}, i1{});
      mma(fa0, fb0, i0{}, i1{});
      load(a1s, l, fa0, fb0, k + 32, i0{}, i1{});   // k + 32 <= kd: at most the diagonal block's first chunk
      mma(fa1, fb1, i0{}, i1{});
    }
    // diagonal block: chunk v holds k = kd + 16 v ..: rows of sub-tiles mi < v are zero there
    load(a1s, l, fa1, fb1, kd + 16, i1{}, i1{});
    mma(fa0, fb0, i0{}, i1{});
    load_a(a2s, l, fa0, kd, i0{}, i0{});         // stage 2's first A chunk sets out
    __builtin_amdgcn_sched_barrier(0);
    mma(fa1, fb1, i1{}, i1{});
    stamp(3);

    // column statistics of this wave's 32 rows, from the accumulators
    {
      float ssq[4] = {0.f, 0.f, 0.f, 0.f}, smu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(p.muE + (int64_t)l * Mp + rb * 32 + mi * 16 + 4 * q);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float v = acc[mi][ni][g];
            ssq[ni] = __builtin_fmaf(v, v, ssq[ni]);
            smu[ni] = __builtin_fmaf(m4[g], v, smu[ni]);
          }
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        ssq[ni] += __shfl_xor(ssq[ni], 16); ssq[ni] += __shfl_xor(ssq[ni], 32);
        smu[ni] += __shfl_xor(smu[ni], 16); smu[ni] += __shfl_xor(smu[ni], 32);
        if (q == 0) {
          red[(0 * NB + rb) * 64 + ni * 16 + r] = ssq[ni];
          red[(1 * NB + rb) * 64 + ni * 16 + r] = smu[ni];
        }
      }
    }
    __syncthreads();                             // every wave is through with the Kzx panel
    // Wt block -> LDS (stage 2's B operand): lane (r, q) holds rows 4q .. 4q+3 of column ni * 16 + r: one 16-byte write
    {
      int wbase = r * PK + rb * 32 + ((q ^ gperm(r)) << 2);
      asm volatile("" : "+v"(wbase));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(P + wbase + ni * 16 * PK + mi * 16) = acc[mi][ni];
    }
    __syncthreads();
    if (tid < 64) {                              // per 128-row block, as the tile kernels write them (finalize_kernel sums the blocks)
#pragma unroll
      for (int b = 0; b < NB / 4; ++b) {
        const int64_t o = ((int64_t)l * (NB / 4) + b) * p.ncp + col0 + tid;
        float s1 = 0.f, m1 = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) { s1 += red[(0 * NB + 4 * b + h) * 64 + tid]; m1 += red[(1 * NB + 4 * b + h) * 64 + tid]; }
        p.ps1[o] = s1;
        p.pm1[o] = m1;
      }
    }

    stamp(4);
    // Two workgroups per CU (8 blocks): the fetching waves -- short stage 2, and the next panel to prepare behind it -- go
    // first (N=200k, M=256, L=32: 7.25 -> 7.17 ms per evaluation, three alternating runs; with one workgroup per CU, 12 and
    // 16 blocks, the same priority changes nothing or costs 0.5 %)
    if constexpr (NB == 8) { if (fetcher && !(p.dbg & 64)) __builtin_amdgcn_s_setprio(3); }
    // ---------------- stage 2: rows of block rb of LuE^T Wt, k = kd .. Mp - 1 (the first 32: the diagonal block) ----------------
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};
    // diagonal block: chunk v holds k = kd + 16 v ..: rows of sub-tiles mi > v are zero there
    load_b(fb0, kd);
    __builtin_amdgcn_sched_barrier(0);
    load(a2s, l, fa1, fb1, kd + 16, i0{}, i1{});
    mma(fa0, fb0, i0{}, i0{});
    if (kd + 32 < Mp) {
      load(a2s, l, fa0, fb0, kd + 32, i0{}, i1{});
      mma(fa1, fb1, i0{}, i1{});
#pragma unroll 1
      for (int k = kd + 32; k < Mp; k += 32) {
        load(a2s, l, fa1, fb1, k + 16, i0{}, i1{});
        mma(fa0, fb0, i0{}, i1{});
        if (k + 32 < Mp) load(a2s, l, fa0, fb0, k + 32, i0{}, i1{});
        mma(fa1, fb1, i0{}, i1{});
      }
    } else {
      mma(fa1, fb1, i0{}, i1{});
    }
    stamp(5);
    {
      float ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) ssq[ni] = __builtin_fmaf(acc[mi][ni][g], acc[mi][ni][g], ssq[ni]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        ssq[ni] += __shfl_xor(ssq[ni], 16); ssq[ni] += __shfl_xor(ssq[ni], 32);
        if (q == 0) red[(2 * NB + rb) * 64 + ni * 16 + r] = ssq[ni];
      }
    }
    stamp(6);
    if constexpr (NB == 8) __builtin_amdgcn_s_setprio(0);
    if (fetcher && un < u_hi && !(p.dbg & 1)) {  // this wave's accumulators are dead: the next panel sets out
      const int ln = un / p.npan;
      panel_fetch(ln, (int64_t)(un - ln * p.npan) * P_TN);
    } else {                                     // (tells the register allocator that nothing of pv lives through the stages)
#pragma unroll
      for (int i = 0; i < NR; ++i) pv[i] = f32x4{0, 0, 0, 0};
    }
    if constexpr (STORE) {
      // Wt retained for the backward pass: the same early finishers copy the Wt panel from LDS to memory while the
      // fetch is in flight -- lane = column (whole 256-byte row segments per store), wave = 64 rows; the 16-byte reads
      // down a column of the permuted image are conflict-free too
      if (fetcher) {
        const int fw = rb - NB / 2;
        const float* const src = P + lane * PK + fw * 64;
        const int gs = gperm(lane);
        float* const dst = p.Wt + ((int64_t)l * Mp + fw * 64) * p.ncp + col0 + lane;
#pragma unroll 4
        for (int kk = 0; kk < 64; kk += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + (kk & ~15) + ((((kk >> 2) & 3) ^ gs) << 2));
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[(int64_t)(kk + e) * p.ncp] = v[e];
        }
      }
    }
    __syncthreads();                             // ... and every wave is through with the Wt panel
    stamp(7);
    ++n_it;
    if (tid < 64) {
#pragma unroll
      for (int b = 0; b < NB / 4; ++b) {
        float s2 = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) s2 += red[(2 * NB + 4 * b + h) * 64 + tid];
        p.ps2[((int64_t)l * (NB / 4) + b) * p.ncp + col0 + tid] = s2;
      }
    }
    if (un >= u_hi) break;
    u = un; l = un / p.npan; pn = un - l * p.npan;
  }
}

static unsigned long long* g_panel_stamps = nullptr;

bool panel_generates(int kind, int d) {
  return (kind == GPZ_KERNEL_RBF || kind == GPZ_KERNEL_MATERN32) && (d == 1 || d == 2);
}

bool panel_supported(int64_t Mp, int64_t ncp) {
  return Mp >= 128 && Mp <= 512 && Mp % 128 == 0 && ncp % P_TN == 0 && Mp * ncp * 4 < (1ll << 31);
}

template <int NB, int GEN>
static int launch_t(const PanelParams& p, bool store, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)P_TN * (p.Mp + 16) + 3 * NB * 64);
  // Dynamic LDS above 64 KB is an opt-in per kernel function AND per device (a process may drive several: the attribute set
  // on the first one does not reach the others), as is the CU count the persistent grid is sized by.
  struct PerDevice { bool attr = false; int cus = 0; };
  static PerDevice seen[64];
  static std::mutex mu;
  int dev = 0;
  GPZ_HIP_OK(hipGetDevice(&dev));
  int cus = 256;
  {
    std::lock_guard<std::mutex> lock(mu);
    PerDevice& d = seen[dev & 63];
    if (!d.attr) {
      GPZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&panel_kernel<NB, false, GEN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GPZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&panel_kernel<NB, true, GEN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
      d.cus = n;
      d.attr = true;
    }
    cus = d.cus;
  }
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  int wgs = cus * (per_cu > 2 ? 2 : per_cu);
  wgs -= wgs % 8;
  if (store) hipLaunchKernelGGL((panel_kernel<NB, true, GEN>), dim3(wgs), dim3(64 * NB), lds, s, p);
  else hipLaunchKernelGGL((panel_kernel<NB, false, GEN>), dim3(wgs), dim3(64 * NB), lds, s, p);
  GPZ_LAUNCH_OK();
  return 0;
}

int panel_launch(const PanelArgs& a, hipStream_t s) {
  GPZ_REQUIRE(panel_supported(a.Mp, a.ncp), "panel_launch: unsupported shape Mp=%lld ncp=%lld", (long long)a.Mp, (long long)a.ncp);
  PanelParams p = {};
  p.Linv = a.Linv; p.LuT = a.LuT; p.Kzx = a.Kzx; p.Wt = a.Wt; p.muE = a.muE;
  p.ps1 = a.ps1; p.pm1 = a.pm1; p.ps2 = a.ps2;
  p.stamps = g_panel_stamps;
  static const int dbg = [] { const char* e = getenv("GPZ_PANEL_DBG"); return e ? atoi(e) : 0; }();
  p.dbg = dbg;
  p.Mp = (int)a.Mp; p.ncp = (int)a.ncp; p.L = a.L; p.npan = (int)(a.ncp / P_TN); p.units = a.L * p.npan;
  const bool store = a.Wt != nullptr;
  int gen = 0;
  if (a.Z) {       // generated operand
    GPZ_REQUIRE(panel_generates(a.kind, a.d) && a.X && a.sigma && a.ell && a.M >= 1 && a.M <= a.Mp && a.nreal >= 1 && a.nreal <= a.ncp,
                "panel_launch: bad generator arguments (kind=%d d=%d)", a.kind, a.d);
    gen = 1 + 2 * (a.kind == GPZ_KERNEL_MATERN32 ? 1 : 0) + (a.d - 1);
    p.Z = a.Z; p.X = a.X; p.sigma = a.sigma; p.ell = a.ell; p.M = (int)a.M; p.nreal = (int)a.nreal;
  } else {
    GPZ_REQUIRE(a.Kzx, "panel_launch: neither a Kzx buffer nor generator arguments");
  }
#define GPZ_PANEL_CASE(NBV)                                  \
  case NBV:                                                  \
    switch (gen) {                                           \
      case 1: return launch_t<NBV, 1>(p, store, s);          \
      case 2: return launch_t<NBV, 2>(p, store, s);          \
      case 3: return launch_t<NBV, 3>(p, store, s);          \
      case 4: return launch_t<NBV, 4>(p, store, s);          \
      default: return launch_t<NBV, 0>(p, store, s);         \
    }
  switch (a.Mp / 32) {
    GPZ_PANEL_CASE(4)
    GPZ_PANEL_CASE(8)
    GPZ_PANEL_CASE(12)
    default:
    GPZ_PANEL_CASE(16)
  }
#undef GPZ_PANEL_CASE
}

}  // namespace gpz

// Debug: device buffer of 16 * 16 * 8 uint64 that workgroup 0 of the next panel launches fills with its phase times
// (shader clock; tools/panel_trace.py), or null to stop.
extern "C" void gpz_debug_panel_stamps(unsigned long long* buf) { gpz::g_panel_stamps = buf; }
