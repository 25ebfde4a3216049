// diag128.h -- Cholesky factor AND inverse of a 128x128 SPD diagonal block held in LDS, as device functions that a
// workgroup of NTH = 256 or 512 threads calls phase by phase (reference call sites gp.py:213/270/360).  Used by
// diag128_kernel (csrc/diag128.hip: one launch per block column of the launch-per-step factorisation) and by the
// one-launch cooperative factorisation (csrc/coop.hip), where the block arrives from the accumulators of its tile.
//
// The block is processed as a 4x4 grid of 32x32 sub-blocks:
//   * a 32x32 diagonal sub-block is factored and inverted by ONE wave with the rows held in
//     registers; the pivot column / the row being eliminated is broadcast with v_readlane, so the
//     sweep needs no LDS traffic and no barrier (wavefront-level reductions);
//   * the sub-panel solve  L[a][s] = A[a][s] inv(L[s][s])^T  and the in-block trailing update
//     A[a][b] -= L[a][s] L[b][s]^T  run on v_mfma_f64_16x16x4_f64 with operands read from LDS;
//   * the inverse of the whole block is assembled by recursive doubling (32 -> 64 -> 128), again on
//     MFMA; the accumulator tile of C*A is fed straight back as the B operand of -B*(C*A) (the f64
//     C/D layout row = (lane>>4) + 4*reg makes register g of a tile exactly k-step g's operand).
// Storage: S[128][129] doubles.  L lives row-major in the lower triangle; the inverse X is kept
// transposed and shifted one column right, X[i][c] at S[c][i+1], which is free space.  A compact
// [32][32] column buffer and the reciprocal diagonal of the sub-block in flight follow S.
// Only waves 0..3 compute; with NTH = 512 waves 4..7 share the copies and meet the barriers.
#pragma once
#include "common.h"

#include <type_traits>

#ifndef GPZ_STAMP
#define GPZ_STAMP(i) do {} while (0)
#endif

namespace gpz {
namespace diag {

constexpr size_t LDS_DOUBLES = (size_t)128 * 129 + 32 * 32 + 32;   // S + column buffer + reciprocal diagonal

constexpr int DP = 129;  // LDS pitch
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ d4 mma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// X[i][c], i >= c, of the inverse
__device__ __forceinline__ double& XT(double* S, int c, int i) { return S[c * DP + i + 1]; }

// A zero the compiler cannot see through, held in a VGPR: added to a uniform LDS address it keeps the
// broadcast loads in vector registers (otherwise every loaded value is moved to an SGPR pair with
// v_readlane and the 500 live multipliers spill).
__device__ __forceinline__ int vzero() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

// sqrt(d) and 1/sqrt(d) in fp64 from the hardware seed (v_rsq_f64, relative error ~2^-23) by two coupled Goldschmidt
// steps  r = 1/2 - g h;  g += g r;  h += h r  (g -> sqrt(d), h -> 1/(2 sqrt(d)); quadratic: 2^-23 -> 2^-45 -> < 2^-53).
// Both results come out of ONE chain of 6 dependent operations; Newton steps for 1/sqrt(d) followed by a corrected
// product for sqrt(d) took 13, and this chain sits on the critical path of every column of the factorisation.
__device__ __forceinline__ void sqrt_rsqrt(double d, double& sq, double& rs) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
  }
  sq = g;
  rs = h + h;
}

// One wave: right-looking Cholesky of the 32x32 block at offset o.  Row i lives in lanes i and i + 32: the low lane holds
// its even columns, the high lane its odd ones (round 5; rounds 2-4 kept the whole row in both and did every update
// twice).  tools/valu_chain_probe.hip: on this chip a wave pays ~5 cycles for every vector instruction it issues and 8.5
// for a dependent one, and independent work does NOT hide in a dependent chain's shadow -- a column costs the SUM of what
// the wave issues, so the sweep is written for few instructions, not for a short chain:
//   * per column the chain is: pivot (v_readlane), reciprocal square root (v_rsq_f64 + two coupled Goldschmidt steps),
//     the multipliers l_ij = a_ij r in the half that holds column j, their store to the column buffer CB[j][.];
//   * ALL updates by column j -- of column j + 1 as of the others -- are made in the next column's body with this row's
//     multiplier and the columns' multipliers read back from CB (one plain and a few broadcast LDS reads: the half that
//     did not compute l_ij gets it that way too, nothing is exchanged between the halves in registers); they are needed
//     only when that column's reciprocal square root is through, ~100 cycles after the reads were issued;
//   * the one value the next pivot needs at once, a_(j+1,j+1) - l_(j+1,j)^2, is formed separately from two v_readlane.
// Every element sees the updates of columns 0, 1, 2, ... in that order with the same operands as in the one-row-per-lane
// form of rounds 2-4: the factor is bit for bit the same.  No test of the pivot on the way (compare, select and branch
// through scalar registers were 40 cycles a column): a non-positive pivot turns into NaN and spreads, and the diagonal
// is looked at once behind the sweep.  RI[j] = 1 / l_jj for the inverse.
__device__ __forceinline__ void factor32(double* S, double* CBu, double* RI, int o, int lane, int32_t* info,
                                         int64_t gbase, int64_t m_real) {
  const int zz = vzero();          // also keeps the 32 lane-compare masks from being hoisted out of the caller's loop
  const int i = (lane & 31) + zz;
  const int h = lane >> 5;         // this lane holds columns 2 m + h of row i in a[m]
  const double* CB = CBu + zz + h;
  double a[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) a[m] = S[(o + i) * DP + o + 2 * m + h];
  double d = bcast(a[0], 0);
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    // the updates by column j - 1 of the columns c = 2 m + h >= j: m >= j / 2 in both halves (for odd j the low half's
    // register (j - 1) / 2 is column j - 1, spent: updating it is harmless)
    // (the reads are issued first, the reciprocal square root runs while they travel, the updates follow it)
    double lprev = 0.0, cb[16];
    if (j >= 1) {
      lprev = CBu[(j - 1) * 32 + i];
#pragma unroll
      for (int m = j >> 1; m < 16; ++m) cb[m] = CB[(j - 1) * 32 + 2 * m];
    }
    double sq, r;
    sqrt_rsqrt(d, sq, r);
    __builtin_amdgcn_sched_barrier(0);
    if (j >= 1) {
#pragma unroll
      for (int m = j >> 1; m < 16; ++m) a[m] = fma(-lprev, cb[m], a[m]);
    }
    const double lj = (i == j) ? sq : a[j >> 1] * r;       // meaningful in the half (j & 1); rows i < j carry garbage
    if (h == (j & 1)) {                                      // ... that is never read back
      CBu[j * 32 + i] = lj;
      if (i >= j) S[(o + i) * DP + o + j] = lj;
    }
    // RI[j] doubles as the "column j is in LDS" flag for the wave that inverts this sub-block behind us (it was zeroed
    // before the sweep; 1 / l_jj > 0).  Stored LAST, as a release at WAVEFRONT scope: that only keeps the compiler from
    // moving the stores above behind it -- the hardware executes a wave's LDS operations in order -- whereas a
    // workgroup-scope release would make this wave drain its LDS queue in every column of the pivot chain.
    __hip_atomic_store(&RI[j], r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WAVEFRONT);   // wave-uniform
    if (j + 1 < 32) {
      // the next pivot: a_(j+1,j+1) as it stands (updated through column j - 1) minus l_(j+1,j)^2
      const double bc = bcast(lj, j + 1 + 32 * (j & 1));
      const double t = fma(-bc, bc, a[(j + 1) >> 1]);
      d = bcast(t, j + 1 + 32 * ((j + 1) & 1));
    }
    // pin the updates to this column: LLVM otherwise sinks each one to the column that consumes it and keeps
    // all the broadcast values alive until then (spills)
#pragma unroll
    for (int m = j >> 1; m < 16; ++m) asm volatile("" : "+v"(a[m]));
    __builtin_amdgcn_sched_barrier(0);
  }
  // LAPACK's info: the first column whose pivot was not positive.  sqrt / rsqrt of such a pivot is NaN (d < 0, d = NaN) or
  // becomes one in the first Goldschmidt product (d = 0: 0 * inf), so l_jj > 0 fails for it -- and, the NaN spreading
  // down and right, possibly for later columns, never for earlier ones.  (A wave's LDS operations execute in order: the
  // read below sees the sweep's stores.)
  const double lii = S[(o + i) * DP + o + i];
  const unsigned long long bad = __ballot(!(lii > 0.0)) & 0xffffffffull;
  if (bad != 0ull) {
    const int jb = __builtin_ctzll(bad);
    if (lane == 0 && gbase + jb < m_real) atomicCAS(info, 0, (int)(gbase + jb + 1));
  }
}

// Inverse-only mode: the column buffer and reciprocal diagonal of an already factored block.
__device__ __forceinline__ void stage32(const double* S, double* CB, double* RI, int o, int lane) {
  const int i = lane & 31;
  if (lane < 32) {
    for (int k = 0; k <= i; ++k) CB[k * 32 + i] = S[(o + i) * DP + o + k];
    RI[i] = 1.0 / S[(o + i) * DP + o + i];
  }
}

// One wave: inverse of the factored 32x32 lower-triangular block by column-oriented forward
// substitution; lane c owns column c of the inverse, x_ii = acc_ii / l_ii, then acc_m -= l_m,ii x_ii
// (independent FMAs fed by broadcast reads of column ii).
__device__ __forceinline__ void invert32(double* S, const double* CBu, const double* RIu, int o, int lane) {
  const int z = vzero();
  const int c = (lane & 31) + z;
  const double* CB = CBu + z;
  const double* RI = RIu + z;
  double acc[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) acc[m] = (m == c) ? 1.0 : 0.0;
  // column ii of L (the multipliers of step ii) is fetched one step ahead into the other half of cb[][]: the factor is
  // complete before this loop, so only the order of the loads decides whether each step waits for an LDS round trip
  double cb[2][32], ri[2];
#pragma unroll
  for (int m = 1; m < 32; ++m) cb[0][m] = CB[m];
  ri[0] = RI[0];
#pragma unroll
  for (int ii = 0; ii < 32; ++ii) {
    if (ii + 1 < 32) {
#pragma unroll
      for (int m = ii + 2; m < 32; ++m) cb[(ii + 1) & 1][m] = CB[(ii + 1) * 32 + m];
      ri[(ii + 1) & 1] = RI[ii + 1];
    }
    const double x = acc[ii] * ri[ii & 1];
    if (lane < 32 && ii >= c) XT(S, o + c, o + ii) = x;
#pragma unroll
    for (int m = ii + 1; m < 32; ++m) acc[m] = fma(-cb[ii & 1][m], x, acc[m]);
#pragma unroll
    for (int m = ii + 1; m < 32; ++m) asm volatile("" : "+v"(acc[m]));
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The value the lanes of half `from` (0: lanes 0-31, 1: lanes 32-63) hold, in both halves: one v_permlane32_swap per word
// (with the same register as both operands it leaves the low half's copy in one result and the high half's in the other).
__device__ __forceinline__ double from_half(double v, int from) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h2[from], (int)l2[from]);
}

// invert32 run by a SECOND wave while the first one is still factoring the sub-block: step ii needs column ii of L
// and 1 / l_ii only, which factor32 publishes column by column (RI[ii] turns non-zero last).  It has to keep up with a
// sweep of ~350 cycles a column (round 5), so, like the sweep, it is written for few instructions: column c of the
// inverse lives in lanes c and c + 32, its even rows in the low lane and its odd rows in the high one (half the
// multiply-adds and half the LDS reads of a lane owning the whole column; x_ii goes from the half that formed it to the
// other by v_permlane32_swap), and a step's flag and multipliers come in ONE LDS round trip -- the flag's read is issued
// first and a wave's LDS reads execute in order, so multipliers read behind a flag that turns out set are the
// published ones (waiting for the flag and reading then was two round trips a step).
__device__ __forceinline__ void invert32_follow(double* S, const double* CBu, double* RIu, int o, int lane) {
  const int z = vzero();
  const int c = (lane & 31) + z;
  const int h = lane >> 5;                       // this lane holds rows 2 m + h of column c in acc[m]
  const double* CB = CBu + z + h;
  double* RI = RIu + z;
  double acc[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[m] = (2 * m + h == c) ? 1.0 : 0.0;
#pragma unroll
  for (int ii = 0; ii < 32; ++ii) {
    // rows 2 m + h > ii: m >= (ii + 1) / 2 in both halves (for even ii the low half's register ii / 2 is row ii itself:
    // spent once x is formed, updating it is harmless)
    double ri, cb[16];
    for (;;) {
      asm volatile("" ::: "memory");
      ri = RI[ii];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int m = (ii + 1) >> 1; m < 16; ++m) cb[m] = CB[ii * 32 + 2 * m];
      asm volatile("" ::: "memory");
      if (ri != 0.0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    const double x = from_half(acc[ii >> 1] * ri, ii & 1);
    if (h == (ii & 1) && ii >= c) XT(S, o + c, o + ii) = x;
#pragma unroll
    for (int m = (ii + 1) >> 1; m < 16; ++m) acc[m] = fma(-cb[m], x, acc[m]);
#pragma unroll
    for (int m = (ii + 1) >> 1; m < 16; ++m) asm volatile("" : "+v"(acc[m]));
    __builtin_amdgcn_sched_barrier(0);
  }
}
// ---- phases ---------------------------------------------------------------------------------------------------------
// global -> S: the whole block with independent 16-byte loads in flight (a rolled load -> LDS loop pays one memory
// round trip per element: 64 x ~0.7 us); the strict upper triangle is zeroed on the way in
template <int NTH>
__device__ __forceinline__ void load_block(double* S, const double* Ab, int64_t lda, int tid) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int NIT = 8192 / NTH;
  d2 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e2 = tid + NTH * it, i = e2 >> 6, j = (e2 & 63) * 2;
    v[it] = *reinterpret_cast<const d2*>(Ab + (int64_t)i * lda + j);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e2 = tid + NTH * it, i = e2 >> 6, j = (e2 & 63) * 2;
    S[i * DP + j] = (j <= i) ? v[it][0] : 0.0;
    S[i * DP + j + 1] = (j + 1 <= i) ? v[it][1] : 0.0;
  }
}

// the shifted column of the transposed inverse and the "column is there" flags; a barrier must follow
__device__ __forceinline__ void prepare(double* S, int tid) {
  double* RI = S + 128 * DP + 32 * 32;
  if (tid < 128) S[tid * DP + 128] = 0.0;
  if (tid < 32) RI[tid] = 0.0;
}

// Factor (factor != 0) the block in S and invert its four 32x32 diagonal sub-blocks; every thread of the workgroup
// calls it (barriers inside).  info: first non-positive pivot of this matrix (real rows only), LAPACK style.
// NW: waves that take part in the sub-panel and trailing products (4, or 8 when the workgroup has them: these phases
// are chains of dependent LDS reads and MFMAs, and a second wave per SIMD runs in the first one's latency).
template <int NW = 4>
__device__ __forceinline__ void factor_block(double* S, int tid, int32_t* info, int64_t gbase, int64_t m_real, int factor) {
  double* CB = S + 128 * DP;
  double* RI = CB + 32 * 32;
  const int lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // tiles t = first, first + stride, ... < last of the trailing update of the step whose sub-block starts at o (R0 = o + 32):
  // two tiles of a wave at a time, advanced together as independent MFMA chains; the loop condition makes both valid, so
  // no MFMA sits under a branch; a possible last single tile follows
  auto trail_tiles = [&](int o, int R0, int first, int stride, int last) __attribute__((always_inline)) {
    auto tile_of = [&](int t, int& i0, int& j0) __attribute__((always_inline)) {
      int ta = 0, rem = t;
      while (rem > ta) { rem -= ta + 1; ++ta; }
      i0 = R0 + 16 * ta; j0 = R0 + 16 * rem;
    };
    int t = first;
    for (; t + stride < last; t += 2 * stride) {
      int ia, ja, ib, jb;
      tile_of(t, ia, ja);
      tile_of(t + stride, ib, jb);
      d4 acc0, acc1;
#pragma unroll
      for (int g = 0; g < 4; ++g) { acc0[g] = S[(ia + q + 4 * g) * DP + ja + r]; acc1[g] = S[(ib + q + 4 * g) * DP + jb + r]; }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = o + 4 * kk + q;
        acc0 = mma(-S[(ia + r) * DP + k], S[(ja + r) * DP + k], acc0);
        acc1 = mma(-S[(ib + r) * DP + k], S[(jb + r) * DP + k], acc1);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) { S[(ia + q + 4 * g) * DP + ja + r] = acc0[g]; S[(ib + q + 4 * g) * DP + jb + r] = acc1[g]; }
    }
    if (t < last) {
      int i0, j0;
      tile_of(t, i0, j0);
      d4 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = S[(i0 + q + 4 * g) * DP + j0 + r];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = o + 4 * kk + q;
        acc = mma(-S[(i0 + r) * DP + k], S[(j0 + r) * DP + k], acc);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) S[(i0 + q + 4 * g) * DP + j0 + r] = acc[g];
    }
  };
  // the tiles t >= 3 of the step before sub-block s, by the waves 2, 3, 6, 7 of SIMDs 2 and 3 (eight-wave workgroups)
  auto trail_rest = [&](int s) __attribute__((always_inline)) {
    if (NW != 8 || s < 1 || s > 2 || (w & 3) < 2) return;
    constexpr int NH = 4;
    const int hi = (w & 1) + 2 * (w >> 2);
    const int R0 = 32 * s, n16 = (128 - R0) / 16;
    trail_tiles(R0 - 32, R0, 3 + hi, NH, n16 * (n16 + 1) / 2);
  };
  for (int s = 0; s < 4; ++s) {
    const int o = 32 * s;
    if (factor) {
      // wave 0 factors, wave 1 inverts one column behind it.  (Letting the idle waves 2 and 3 write the finished block
      // column of L back meanwhile was measured: the final write-back shrinks by 3.5 k cycles and factor32 grows by as
      // much from the LDS contention.)
      if (w == 0) {
        factor32(S, CB, RI, o, lane, info, gbase + o, m_real);
        GPZ_STAMP(2 + 4 * s);
      } else if (w == 1) {
#ifndef GPZ_DIAG_NOFOLLOW
        invert32_follow(S, CB, RI, o, lane);
#endif
      } else {
        trail_rest(s);
      }
    } else if (w == 0) {
      stage32(S, CB, RI, o, lane);
      invert32(S, CB, RI, o, lane);
    }
    __syncthreads();
    if (factor && w == 2 && lane < 32) RI[lane] = 0.0;   // flags of the next sub-block (two barriers away)
    GPZ_STAMP(3 + 4 * s);
    if (!factor || s == 3) continue;
    const int R0 = o + 32;
    // ---- sub-panel: rows R0..127, columns o..o+31, in place: L = A * inv(Lss)^T ----
    {
      const int ntiles = ((128 - R0) / 16) * 2;
      d4 acc[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) acc[u] = d4{0, 0, 0, 0};
      // every wave owns up to ceil(ntiles / NW) tiles (t = w, w + NW, ...): advanced together k-step by k-step, as
      // independent MFMA chains, in a body specialised on that count (MFMAs under a run-time condition make hipcc shuffle
      // the accumulators through VGPR copies: measured 2x slower)
      auto panel = [&](auto nu_c) __attribute__((always_inline)) {
        constexpr int NU = decltype(nu_c)::value;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int k = 4 * kk + q;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int t = w + NW * u;
            const int i0 = R0 + 16 * (t >> 1), j = 16 * (t & 1) + r;
            const double av = S[(i0 + r) * DP + o + k];
            const double bv = (j >= k) ? XT(S, o + k, o + j) : 0.0;   // inv(Lss)[j][k]
            acc[u] = mma(av, bv, acc[u]);
          }
        }
      };
      const int mine = w < NW ? (ntiles - w + NW - 1) / NW : 0;     // tiles t = w, w + NW, ... < ntiles (wave-uniform)
      if (mine == 3) panel(std::integral_constant<int, 3>{});
      else if (mine == 2) panel(std::integral_constant<int, 2>{});
      else if (mine == 1) panel(std::integral_constant<int, 1>{});
      __syncthreads();
      if (w < NW) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int t = w + NW * u;
          if (t < ntiles) {
            const int i0 = R0 + 16 * (t >> 1), j0 = 16 * (t & 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) S[(i0 + q + 4 * g) * DP + o + j0 + r] = acc[u][g];
          }
        }
      }
      __syncthreads();
    }
    GPZ_STAMP(4 + 4 * s);
    // ---- in-block trailing update: A[a][b] -= L[a][s] L[b][s]^T, 16x16 tiles with a >= b ----
    // Only the three tiles of the NEXT diagonal sub-block (t = 0, 1, 2) stand between this step and the next sweep; they
    // are done here, the others by the waves of SIMDs 2 and 3 WHILE that sweep runs on SIMDs 0 and 1 (trail_rest at the
    // top of the loop: the sweep touches nothing but its sub-block, the column buffer and the strictly upper triangle;
    // the barrier that ends it is the one these tiles are needed behind).
    // (With four waves there are two left for that, and the 18 other tiles of the first step take them longer than the
    // sweep lasts: measured no gain -- all tiles stay here then.)
    if (NW == 8) {
      if (w >= 2 && w < 5) trail_tiles(o, R0, w - 2, 3, 3);
    } else if (w < NW) {
      const int n16 = (128 - R0) / 16;
      trail_tiles(o, R0, w, NW, n16 * (n16 + 1) / 2);
    }
    __syncthreads();
    GPZ_STAMP(5 + 4 * s);
  }
}

// S -> global: the factor, zeros above the diagonal.  8 LDS values are read before their 8 stores.
template <int NTH>
__device__ __forceinline__ void store_factor(const double* S, double* Ab, int64_t lda, int tid) {
#pragma unroll 1
  for (int e0 = tid; e0 < 128 * 128; e0 += NTH * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + NTH * u, i = e >> 7, j = e & 127;
      v[u] = (j <= i) ? S[i * DP + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + NTH * u;
      Ab[(int64_t)(e >> 7) * lda + (e & 127)] = v[u];
    }
  }
}

// The 32 -> 64 -> 128 levels of the block's inverse (the four 32x32 diagonal inverses are in place); every thread calls
// it (barriers inside); the inverse is complete behind the last barrier.
__device__ __forceinline__ void inverse_levels(double* S, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // ---- level 1: 32 -> 64.  wave = (pair p, column tile jt of the left block) ----
  if (w < 4) {
    const int p = w >> 1, jt = w & 1;
    const int oA = 64 * p, oB = oA + 32;
    const int j = 16 * jt + r;
    d4 T[2], X[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      T[it] = d4{0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = 4 * kk + q;
        const double av = S[(oB + 16 * it + r) * DP + oA + k];                 // C = L[2p+1][2p]
        const double bv = (k >= j) ? XT(S, oA + j, oA + k) : 0.0;              // A^-1[k][j]
        T[it] = mma(av, bv, T[it]);
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      X[it] = d4{0, 0, 0, 0};
      const int i = 16 * it + r;
#pragma unroll
      for (int kt = 0; kt <= it; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k = 16 * kt + 4 * g + q;
          const double av = (k <= i) ? XT(S, oB + k, oB + i) : 0.0;            // B^-1[i][k]
          X[it] = mma(-av, T[kt][g], X[it]);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int g = 0; g < 4; ++g) XT(S, oA + j, oB + 16 * it + q + 4 * g) = X[it][g];
  }
  __syncthreads();
  GPZ_STAMP(20);
  // ---- level 2: 64 -> 128.  wave = column tile jt of the left 64 columns ----
  if (w < 4) {
    const int jt = w, j = 16 * jt + r;
    d4 T[4], X[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) { T[kt] = d4{0, 0, 0, 0}; X[kt] = d4{0, 0, 0, 0}; }
    // the four tiles of T (and then of X) advance together: independent MFMA chains in flight, B operand shared
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = 4 * kk + q;
      const double bv = (k >= j) ? XT(S, j, k) : 0.0;                          // A^-1[k][j], A = X[0:64][0:64]
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) T[kt] = mma(S[(64 + 16 * kt + r) * DP + k], bv, T[kt]);   // C = L[64:128][0:64]
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = 16 * kt + 4 * g + q;
#pragma unroll
        for (int it = kt; it < 4; ++it) {
          const int i = 16 * it + r;
          const double av = (k <= i) ? XT(S, 64 + k, 64 + i) : 0.0;            // B^-1[i][k], B = X[64:][64:]
          X[it] = mma(-av, T[kt][g], X[it]);
        }
      }
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int g = 0; g < 4; ++g) XT(S, j, 64 + 16 * it + q + 4 * g) = X[it][g];
  }
  __syncthreads();
  GPZ_STAMP(21);
}

// S -> global: the inverse X (lower triangular, zeros above), rows of length 128 at pitch ldd
template <int NTH>
__device__ __forceinline__ void store_inverse(const double* Sc, double* Db, int64_t ldd, int tid) {
  double* S = const_cast<double*>(Sc);
#pragma unroll 1
  for (int e0 = tid; e0 < 128 * 128; e0 += NTH * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + NTH * u, i = e >> 7, c = e & 127;
      v[u] = (c <= i) ? XT(S, c, i) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + NTH * u;
      Db[(int64_t)(e >> 7) * ldd + (e & 127)] = v[u];
    }
  }
}

// S -> global: the TRANSPOSE of the inverse (upper triangular, zeros below), rows of length 128 at pitch ldd
template <int NTH>
__device__ __forceinline__ void store_inverse_t(const double* Sc, double* Dt, int64_t ldd, int tid) {
  double* S = const_cast<double*>(Sc);
#pragma unroll 1
  for (int e0 = tid; e0 < 128 * 128; e0 += NTH * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + NTH * u, c = e >> 7, i = e & 127;
      v[u] = (c <= i) ? XT(S, c, i) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + NTH * u;
      Dt[(int64_t)(e >> 7) * ldd + (e & 127)] = v[u];
    }
  }
}

}  // namespace diag
}  // namespace gpz
