// gemm.hip -- 128x128-tile batched GEMM on the CDNA4 matrix cores.
//
// fp32: v_mfma_f32_16x16x4_f32, fp64: v_mfma_f64_16x16x4_f64 (exact IEEE FMA
// chains, so results are deterministic and match a plain fp32/fp64 reference to
// rounding).  256 threads = 4 waves in a 2x2 grid, 64x64 outputs per wave held
// in 16 accumulator tiles.  A and (transposed) B tiles are staged [row][k] in LDS
// and fetched with one 16-byte ds_read per lane covering VEC consecutive k; the
// non-transposed B tile is staged [k][n].  Because the MFMA sums its four k
// slots, lane group q may own k = VEC*q .. VEC*q+VEC-1 as long as A and B agree,
// which is what makes the wide read legal.  Global->LDS staging is double
// buffered through registers with one barrier per k-tile.
//
// Triangular structure (L^{-1} and Lu^T are triangular, SYRK only needs the lower
// tiles) is expressed as a per-tile k-range in units of the 128-block, so no
// flop is spent on known-zero blocks.
#include "gemm.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace gpz {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Mma;
template <> struct Mma<float> {
  using acc_t = f32x4;
  using vec_t = f32x4;  // 16 bytes
  static constexpr int VEC = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of the 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int crow(int q, int g) { return 4 * q + g; }
};
template <> struct Mma<double> {
  using acc_t = f64x4;
  using vec_t = f64x2;
  static constexpr int VEC = 2;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // f64 is the exception: row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int crow(int q, int g) { return q + 4 * g; }
};


// LDS bytes of one instantiation (two buffers of an A and a B tile)
template <typename T, int KV, bool BT>
constexpr size_t gemm_lds_bytes() {
  constexpr int VEC = 16 / sizeof(T), BK = 4 * VEC * KV, LDR = BK + VEC, LDN = 128 + (VEC == 4 ? 4 : 8);
  return sizeof(T) * 2 * (128 * LDR + (BT ? 128 * LDR : BK * LDN));
}

// KV = number of 64-byte k-chunks per staged tile (2 in production: 32-deep f32 / 16-deep f64 tiles,
// 70 / 74 KB of LDS for the two buffers; with one chunk the one-tile prefetch distance of the fp64
// kernel was shorter than the memory latency).
// NI = 16-column sub-tiles per wave: 4 gives 4 waves (2x2) of 64x64, 2 gives 8 waves (2x4) of
// 64x32 (production).  fp64 MFMAs only reach their rate with >= 3 waves per SIMD (36 TF with one wave,
// 49 TF with three or more, measured), which the 128 accumulator registers of a 64x64 fp64 wave tile
// rule out; 64x32 halves them and leaves 4 waves per SIMD in both precisions.
// PIPE: the full-range part of the k-loop runs software-pipelined across the tile barrier (see below).
template <typename T, int KV, int NI, bool BT, int EPI, bool PIPE>
__global__ __launch_bounds__(1024 / NI, NI == 4 ? 3 : 4) void gemm128_kernel(const GemmParams<T> p) {
  constexpr int WN = 8 / NI;                // waves along N
  constexpr int NT = 128 * WN;              // threads: 2 x WN waves
  using M = Mma<T>;
  using vec_t = typename M::vec_t;
  using acc_t = typename M::acc_t;
  constexpr int VEC = M::VEC;
  constexpr int BK = 4 * VEC * KV;          // k-depth of a staged tile
  constexpr int LDR = BK + VEC;             // [row][k] tile row: 16-byte pad (2-way on 16-B reads)
  constexpr int LDN = 128 + (VEC == 4 ? 4 : 8);  // [k][n] tile row (elements)
  constexpr int A_ELEMS = 128 * LDR;
  constexpr int B_ELEMS = BT ? 128 * LDR : BK * LDN;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* const smem = reinterpret_cast<T*>(smem_raw);
  auto sA = [&](int buf) -> T* { return smem + buf * (A_ELEMS + B_ELEMS); };
  auto sB = [&](int buf) -> T* { return smem + buf * (A_ELEMS + B_ELEMS) + A_ELEMS; };

  // ---------------- tile decode ----------------
  int b0, b1, ti, tj;
  {
    const int bid = blockIdx.x;
    if (p.flags & GF_GROUP_COLS) {
      // Super-tiles for L2 reuse: one XCD (blocks b, b+8, ... share an XCD's L2) works through
      // (latent, strip of SUPER_COLS column tiles) units; inside a unit, row tiles go longest
      // k-range first and the SUPER_COLS blocks of one row tile are dispatched together, so they
      // stream the same A panel in lock-step while the unit's B panels stay L2 resident.
      const int SUPER_COLS = p.super_cols;
      const int x = bid & 7, s = bid >> 3;
      const int strips = (p.nt + SUPER_COLS - 1) / SUPER_COLS;
      const int per_unit = p.mt * SUPER_COLS;
      const int unit = (s / per_unit) * 8 + x, within = s % per_unit;
      if (unit >= p.nb0 * strips) return;
      b0 = unit / strips; b1 = 0;
      const int ii = within / SUPER_COLS;
      tj = (unit - b0 * strips) * SUPER_COLS + within % SUPER_COLS;
      if (tj >= p.nt) return;
      ti = (p.flags & GF_A_LOWER) ? p.mt - 1 - ii : ii;   // longest k-range first
    } else {
      const int per = (p.flags & GF_TILES_LOWER) ? p.mt * (p.mt + 1) / 2 : p.mt * p.nt;
      const int b = bid / per;
      int t = bid - b * per;
      b0 = b / p.nb1; b1 = b - b0 * p.nb1;
      if (p.flags & GF_TILES_LOWER) {
        int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        ti = i; tj = t - i * (i + 1) / 2;
      } else {
        ti = t / p.nt; tj = t - ti * p.nt;
      }
    }
  }
  int k_begin = 0, k_end = p.K;
  if (p.flags & GF_A_LOWER) k_end = min(k_end, (ti + 1) * 128);
  if (p.flags & GF_A_UPPER) k_begin = max(k_begin, ti * 128);
  if (p.flags & GF_B_LOWER) k_begin = max(k_begin, tj * 128);
  if (p.flags & GF_B_UPPER) k_end = min(k_end, (tj + 1) * 128);

  const T* Ag = p.A + b0 * p.sA0 + b1 * p.sA1 + (int64_t)ti * 128 * p.lda;
  const T* Bg = p.B + b0 * p.sB0 + b1 * p.sB1 + (BT ? (int64_t)tj * 128 * p.ldb : (int64_t)tj * 128);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // Waves 0..WN-1 sit on different SIMDs than waves WN..2WN-1 in every workgroup, and with a triangular A
  // one row half of the tile has less MFMA work than the other (zero skipping below).  Alternating which
  // half a wave takes from tile to tile keeps the four MFMA pipes of a CU equally loaded.
  const int wm = (wave / WN) ^ ((p.flags & (GF_A_LOWER | GF_A_UPPER)) ? ((ti ^ tj) & 1) : 0), wn = wave % WN;
  const int r = lane & 15, q = lane >> 4;

  // ---------------- staging maps (16 bytes per thread per load) ----------------
  // [row][k] tiles (A, and B when BT): VPR vectors per row, 256/VPR rows per pass
  constexpr int VPR = BK / VEC;               // 4 (KV=1) / 8 (KV=2)
  constexpr int RROWS = NT / VPR;             // rows per pass
  constexpr int RP = 128 / RROWS;             // passes
  const int ra_row = tid / VPR, ra_vc = (tid % VPR) * VEC;
  // [k][n] tile: 128/VEC vectors per row
  constexpr int NV = 128 / VEC;               // 32 / 64 threads per row
  constexpr int BROWS = NT / NV;              // rows per pass
  constexpr int NP = BK / BROWS;              // passes
  static_assert(NP == RP, "A and B staging use the same number of passes");
  const int rb_row = tid / NV, rb_vc = (tid % NV) * VEC;

  // running global pointers of this thread's 16-byte pieces of each tile
  const T* pa[RP];
  const T* pb[RP];
#pragma unroll
  for (int h = 0; h < RP; ++h) {
    pa[h] = Ag + (int64_t)(ra_row + RROWS * h) * p.lda + k_begin + ra_vc;
    pb[h] = BT ? Bg + (int64_t)(ra_row + RROWS * h) * p.ldb + k_begin + ra_vc
               : Bg + (int64_t)(k_begin + rb_row + BROWS * h) * p.ldb + rb_vc;
  }
  const int64_t b_step = BT ? (int64_t)BK : (int64_t)BK * p.ldb;
  // Global -> register -> LDS staging, PFD tiles ahead (tile i lives in register set i % PFD).  Two tiles ahead
  // (fp32 has the registers for it) measured the same as one: the prefetch distance is not what the MFMA pipes
  // wait for.  Kept at one.
  constexpr int PFD = 1;
  vec_t ga[PFD][RP], gb[PFD][RP];
  auto gload_set = [&](auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int h = 0; h < RP; ++h) {
      ga[S][h] = *reinterpret_cast<const vec_t*>(pa[h]);
      gb[S][h] = *reinterpret_cast<const vec_t*>(pb[h]);
      pa[h] += BK;
      pb[h] += b_step;
    }
  };
  auto sstore_set = [&](int buf, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int h = 0; h < RP; ++h)
      *reinterpret_cast<vec_t*>(sA(buf) + (ra_row + RROWS * h) * LDR + ra_vc) = ga[S][h];
    if (BT) {
#pragma unroll
      for (int h = 0; h < RP; ++h)
        *reinterpret_cast<vec_t*>(sB(buf) + (ra_row + RROWS * h) * LDR + ra_vc) = gb[S][h];
    } else {
#pragma unroll
      for (int h = 0; h < RP; ++h)
        *reinterpret_cast<vec_t*>(sB(buf) + (rb_row + BROWS * h) * LDN + rb_vc) = gb[S][h];
    }
  };
  using std::integral_constant;

  acc_t acc[4][NI];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = acc_t{0, 0, 0, 0};

  // One staged tile of MFMAs over the 16-row sub-tiles MI_LO..MI_HI of this wave (compile-time range).
  // In each 64-byte k-chunk lane (r, q) owns k = VEC*q + j, j < VEC.
  auto compute = [&](int buf, auto lo_c, auto hi_c) __attribute__((always_inline)) {
    constexpr int MI_LO = decltype(lo_c)::value, MI_HI = decltype(hi_c)::value;
#pragma unroll
    for (int kc = 0; kc < KV; ++kc) {
      const int ko = kc * 4 * VEC;
      vec_t fa[4];
#pragma unroll
      for (int mi = MI_LO; mi <= MI_HI; ++mi)
        fa[mi] = *reinterpret_cast<const vec_t*>(sA(buf) + (wm * 64 + mi * 16 + r) * LDR + ko + q * VEC);
      if (BT) {
        vec_t fb[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          fb[ni] = *reinterpret_cast<const vec_t*>(sB(buf) + (wn * 16 * NI + ni * 16 + r) * LDR + ko + q * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
          for (int mi = MI_LO; mi <= MI_HI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = M::mma(fa[mi][j], fb[ni][j], acc[mi][ni]);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          T fb[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) fb[ni] = sB(buf)[(ko + q * VEC + j) * LDN + wn * 16 * NI + ni * 16 + r];
#pragma unroll
          for (int mi = MI_LO; mi <= MI_HI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = M::mma(fa[mi][j], fb[ni], acc[mi][ni]);
        }
      }
    }
  };

  // Zero skipping in the diagonal 128-block of a triangular A, without branching around MFMAs (hipcc
  // shuffles all accumulators through VGPRs when they are individually conditional): the k-loop is a
  // sequence of straight-line loops.
  //   * a wave whose 64 rows x BK k lie entirely in the zero triangle stages and syncs but issues no
  //     MFMA: a prefix (A upper, lower-half waves) or a suffix (A lower, upper-half waves) of PT tiles;
  //   * inside the wave's own 64 x 64 diagonal sub-block the 16-row sub-tiles drop out one by one: PT
  //     tiles in four phases with a compile-time sub-tile range (mi >= u for A lower, mi <= u for A upper).
  constexpr int PT = 64 / BK;               // staged tiles per 64 k
  const int nk = (k_end - k_begin) / BK;
  const int wm_s = __builtin_amdgcn_readfirstlane(wm);
  int n_pre = 0, n_post = 0;
  bool part_lo = false, part_hi = false;
  if ((p.flags & GF_A_LOWER) && k_end == (ti + 1) * 128 && k_begin <= ti * 128) { part_lo = true; n_post = wm_s == 0 ? PT : 0; }
  if ((p.flags & GF_A_UPPER) && k_begin == ti * 128 && k_end >= (ti + 1) * 128) { part_hi = true; n_pre = wm_s == 1 ? PT : 0; }
  if (nk > 0) {
    gload_set(integral_constant<int, 0>{});
    sstore_set(0, integral_constant<int, 0>{});
    if (PFD == 2 && nk > 1) gload_set(integral_constant<int, PFD - 1>{});
  }
  __syncthreads();
  int t = 0;
  // Iteration t: fetch tile t + PFD, run tile t (when `run`), stage tile t + 1 into the other LDS buffer.
  // Every region below starts at an even t and has an even length (k-ranges are multiples of 128 = 4 BK at
  // most... PT tiles per 64 k), so the parity of t -- the LDS buffer and, with two register sets, which set
  // is fetched and which is staged -- is a compile-time constant of each iteration: straight-line code, and
  // the compiler can count exactly how many loads may still be in flight at each LDS store.
  static_assert(PT % 2 == 0, "staged tiles come in pairs");
  auto iteration = [&](auto par_c, auto run_c, auto lo_c, auto hi_c) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;                  // t & 1
    constexpr int FS = PFD == 1 ? 0 : P, SS = PFD == 1 ? 0 : 1 - P;
    if (t + PFD < nk) gload_set(integral_constant<int, FS>{});
    if constexpr (decltype(run_c)::value) compute(P, lo_c, hi_c);
    if (t + 1 < nk) sstore_set(P ^ 1, integral_constant<int, SS>{});
    __syncthreads();
    ++t;
  };
  using no_run = integral_constant<bool, false>;
  using run = integral_constant<bool, true>;
  using i0 = integral_constant<int, 0>;
  using i1 = integral_constant<int, 1>;
  using i3 = integral_constant<int, 3>;
  // tile u of the diagonal sub-block covers k in [u*BK, (u+1)*BK): it needs the 16-row sub-tiles
  // mi >= u*BK/16 when A is lower triangular, mi <= ((u+1)*BK-1)/16 when it is upper triangular
  for (int u = 0; u < n_pre; u += 2) { iteration(i0{}, no_run{}, i0{}, i3{}); iteration(i1{}, no_run{}, i0{}, i3{}); }
  if (part_hi) {
    auto phases = [&](auto self, auto u_c) __attribute__((always_inline)) -> void {
      constexpr int U = decltype(u_c)::value;
      iteration(integral_constant<int, U & 1>{}, run{}, i0{}, integral_constant<int, ((U + 1) * BK - 1) / 16>{});
      if constexpr (U + 1 < PT) self(self, integral_constant<int, U + 1>{});
    };
    phases(phases, i0{});
  }
  const int t_main_end = nk - n_post - (part_lo ? PT : 0);
  if constexpr (PIPE) {
    // Software pipeline across the barrier.  A staged tile is two 64-byte k-chunks; the fragments of chunk 0 of
    // tile t+1 are fetched from LDS right AFTER the barrier that publishes the tile, and the MFMAs of chunk 1 of
    // tile t -- operands already in registers -- are issued behind those reads, so the pipe has 32 MFMAs (1024
    // cycles) to chew on while the LDS round trip of the new tile is in flight; chunk 1's fragments are fetched
    // in front of chunk 0's MFMAs the same way.  Without this each wave (and, in lock-step, its partner of the
    // same workgroup on the same SIMD) starts every tile with all its fragment reads and an exposed LDS latency,
    // and waits again on the operand reads the compiler places just in time mid-tile.
    static_assert(KV == 2, "the pipelined loop is written for two k-chunks per staged tile");
    struct Frag { vec_t a[4]; vec_t bv[NI]; T bs[VEC][NI]; };
    auto frag_read = [&](int buf, auto kc_c, Frag& f) __attribute__((always_inline)) {
      constexpr int ko = decltype(kc_c)::value * 4 * VEC;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        f.a[mi] = *reinterpret_cast<const vec_t*>(sA(buf) + (wm * 64 + mi * 16 + r) * LDR + ko + q * VEC);
      if constexpr (BT) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          f.bv[ni] = *reinterpret_cast<const vec_t*>(sB(buf) + (wn * 16 * NI + ni * 16 + r) * LDR + ko + q * VEC);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) f.bs[j][ni] = sB(buf)[(ko + q * VEC + j) * LDN + wn * 16 * NI + ni * 16 + r];
      }
    };
    auto frag_mma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if constexpr (BT) acc[mi][ni] = M::mma(f.a[mi][j], f.bv[ni][j], acc[mi][ni]);
            else acc[mi][ni] = M::mma(f.a[mi][j], f.bs[j][ni], acc[mi][ni]);
          }
    };
    Frag f0, f1;
    if (t < t_main_end) frag_read(0, i0{}, f0);       // t is even here: buffer 0
    auto piter = [&](auto par_c) __attribute__((always_inline)) {
      constexpr int P = decltype(par_c)::value;
      if (t + 1 < nk) gload_set(i0{});
      frag_read(P, i1{}, f1);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(f0);
      if (t + 1 < nk) sstore_set(P ^ 1, i0{});
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < t_main_end) frag_read(P ^ 1, i0{}, f0);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(f1);
      ++t;
    };
    while (t < t_main_end) { piter(i0{}); piter(i1{}); }
  } else {
    while (t < t_main_end) { iteration(i0{}, run{}, i0{}, i3{}); iteration(i1{}, run{}, i0{}, i3{}); }
  }
  if (part_lo) {
    auto phases = [&](auto self, auto u_c) __attribute__((always_inline)) -> void {
      constexpr int U = decltype(u_c)::value;
      iteration(integral_constant<int, U & 1>{}, run{}, integral_constant<int, (U * BK) / 16>{}, i3{});
      if constexpr (U + 1 < PT) self(self, integral_constant<int, U + 1>{});
    };
    phases(phases, i0{});
  }
  while (t < nk) { iteration(i0{}, no_run{}, i0{}, i3{}); iteration(i1{}, no_run{}, i0{}, i3{}); }

  // ---------------- epilogue ----------------
  const int64_t crow0 = (int64_t)ti * 128 + wm * 64;
  const int64_t ccol0 = (int64_t)tj * 128 + wn * 16 * NI;
  if (EPI == EPI_STORE_STATS || EPI == EPI_STATS) {
    // column sums over this block's 128 rows: registers -> lane groups -> the two wm waves
    T* red = smem;  // [2 stats][2 wm][128 cols]; all tile reads are behind the loop's last barrier
    const T* mu = (EPI == EPI_STORE_STATS) ? p.mu + b0 * p.sMu + crow0 : nullptr;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      T ssq = 0, smu = 0;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const T v = p.alpha * acc[mi][ni][g];
          ssq = fma(v, v, ssq);
          if (EPI == EPI_STORE_STATS) smu = fma(mu[mi * 16 + M::crow(q, g)], v, smu);
        }
      ssq += __shfl_xor(ssq, 16); ssq += __shfl_xor(ssq, 32);
      if (EPI == EPI_STORE_STATS) { smu += __shfl_xor(smu, 16); smu += __shfl_xor(smu, 32); }
      if (q == 0) {
        red[wm * 128 + wn * 16 * NI + ni * 16 + r] = ssq;
        if (EPI == EPI_STORE_STATS) red[256 + wm * 128 + wn * 16 * NI + ni * 16 + r] = smu;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int64_t o = ((int64_t)b0 * p.mt + ti) * p.ncols + (int64_t)tj * 128 + tid;
      p.ps_sq[o] = red[tid] + red[128 + tid];
      if (EPI == EPI_STORE_STATS) p.ps_mu[o] = red[256 + tid] + red[384 + tid];
    }
  }
  if (EPI == EPI_STORE_STATS) __syncthreads();   // the strips below reuse the reduction scratch
  if (EPI != EPI_STATS) {
    T* Cg = p.C + b0 * p.sC0 + b1 * p.sC1;
    if (EPI == EPI_STORE && BT && p.beta != (T)0) {
      // read-modify-write (trailing updates): straight from the accumulator layout.  All loads of a batch
      // are issued before its first store: written element by element, the compiler must assume every store
      // aliases the next load and the tile pays one memory round trip per element (32 x ~0.8 us in fp64)
      // instead of one per 16-row sub-tile.
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        T cin[NI][4];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            cin[ni][g] = Cg[(crow0 + mi * 16 + M::crow(q, g)) * p.ldc + ccol0 + ni * 16 + r];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            Cg[(crow0 + mi * 16 + M::crow(q, g)) * p.ldc + ccol0 + ni * 16 + r] =
                p.alpha * acc[mi][ni][g] + p.beta * cin[ni][g];
      }
    } else {
      // pure store: transpose each wave's tile through its private LDS strip so every store
      // instruction writes whole row segments (16 B per lane) instead of 64-byte pieces
      constexpr int WC = 16 * NI;                 // columns per wave
      constexpr int MPP = VEC == 4 ? 2 : 1;       // 16-row sub-tiles per pass (LDS budget)
      constexpr int LDE = WC + VEC;               // strip row stride (elements), keeps 16-B alignment
      T* strip = smem + wave * (MPP * 16 * LDE);
      T cs[NI];                                   // alpha, times the column factor when asked for
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        cs[ni] = (EPI == EPI_STORE_COLSCALE) ? p.alpha * p.colscale[b0 * p.sCs + ccol0 + ni * 16 + r] : p.alpha;
#pragma unroll
      for (int pass = 0; pass < 4 / MPP; ++pass) {
#pragma unroll
        for (int mm = 0; mm < MPP; ++mm)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              strip[(mm * 16 + M::crow(q, g)) * LDE + ni * 16 + r] = cs[ni] * acc[pass * MPP + mm][ni][g];
        // the strip is private to this wave and a wave's LDS operations complete in order: no block barrier
        __builtin_amdgcn_wave_barrier();
        constexpr int LPR = WC / VEC;             // lanes per row of the strip
        constexpr int RPI = 64 / LPR;             // rows per wave instruction
        constexpr int NIT = MPP * 16 / RPI;
        const int c4 = (lane % LPR) * VEC;
        vec_t w[NIT], sc, cv;
        T rv[NIT];
        if (EPI == EPI_WBAR) {   // every operand of the pass is loaded before its first store (stores may alias)
          sc = *reinterpret_cast<const vec_t*>(p.colscale + b0 * p.sCs + ccol0 + c4);
          cv = *reinterpret_cast<const vec_t*>(p.colvec + b0 * p.sCs + ccol0 + c4);
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int64_t grow = crow0 + pass * MPP * 16 + it * RPI + lane / LPR;
            w[it] = *reinterpret_cast<const vec_t*>(p.aux + b0 * p.sC0 + b1 * p.sC1 + grow * p.ldc + ccol0 + c4);
            rv[it] = p.rowvec[b0 * p.sRv + grow];
          }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int row = it * RPI + lane / LPR;
          vec_t v = *reinterpret_cast<const vec_t*>(strip + row * LDE + c4);
          const int64_t grow = crow0 + pass * MPP * 16 + row;
          if (EPI == EPI_WBAR) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = v[e] + rv[it] * cv[e] - w[it][e] * sc[e];
          }
          *reinterpret_cast<vec_t*>(Cg + grow * p.ldc + ccol0 + c4) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

template <typename T>
int gemm_launch(const GemmParams<T>& p, int epilogue, hipStream_t s) {
  GPZ_REQUIRE(p.mt > 0 && p.nt > 0 && p.nb0 > 0 && p.nb1 > 0, "gemm: empty problem");
  GPZ_REQUIRE(p.K % 128 == 0, "gemm: K=%d is not a multiple of 128", p.K);
  GPZ_REQUIRE(p.lda % 4 == 0 && p.ldb % 4 == 0, "gemm: leading dimensions must be multiples of 4");
  int64_t nblocks;
  if (p.flags & GF_GROUP_COLS) {
    GPZ_REQUIRE(p.nb1 == 1 && !(p.flags & GF_TILES_LOWER), "gemm: GROUP_COLS needs a flat batch and a full tile grid");
    const int sc = p.super_cols;
    GPZ_REQUIRE(sc >= 1, "gemm: super_cols must be >= 1");
    const int64_t units = (int64_t)p.nb0 * ((p.nt + sc - 1) / sc);
    nblocks = (units + 7) / 8 * 8 * p.mt * sc;
  } else {
    const int64_t per = (p.flags & GF_TILES_LOWER) ? (int64_t)p.mt * (p.mt + 1) / 2 : (int64_t)p.mt * p.nt;
    nblocks = per * p.nb0 * p.nb1;
  }
  GPZ_REQUIRE(nblocks < (1ll << 31), "gemm: grid too large");
  const bool bt = (p.flags & GF_B_TRANS) != 0;
  if (epilogue != EPI_STORE) GPZ_REQUIRE(!bt, "gemm: stats / column-scale epilogues are NN only");
  if (p.beta != (T)0) GPZ_REQUIRE(bt && epilogue == EPI_STORE, "gemm: accumulation (beta != 0) is built for C += A * B^T only");
  if (epilogue == EPI_STORE_COLSCALE) GPZ_REQUIRE(p.colscale && p.beta == (T)0, "gemm: column-scale epilogue needs factors and beta = 0");
  // Tile configuration (KV 64-byte k-chunks per staged tile, NI 16-column sub-tiles per wave): both
  // precisions run 8 waves of 64x32 on two-chunk tiles (32-deep fp32, 16-deep fp64), 4 waves per SIMD.
  // Measured against 4 waves of 64x64 on one-chunk tiles in fp32: +0.6 % at M=2048, +4 % at M=512.
  auto run = [&](auto kv_c, auto ni_c, auto pipe_c) -> int {
    constexpr int KV = decltype(kv_c)::value, NI = decltype(ni_c)::value;
    dim3 grid((unsigned)nblocks), block(1024 / NI);
    auto launch = [&](auto kernel, size_t lds) -> int {
      // dynamic LDS above 64 KB is an opt-in per kernel FUNCTION and device: every variant has the same pointer
      // type, so the record is keyed on the pointer value (a handful of variants: linear search)
      struct Seen { const void* fn; int dev; };
      static Seen seen[256];
      static int n_seen = 0;
      static std::mutex mu;
      int dev = 0;
      GPZ_HIP_OK(hipGetDevice(&dev));
      if (lds > 64 * 1024) {
        const void* fn = reinterpret_cast<const void*>(kernel);
        std::lock_guard<std::mutex> lock(mu);
        bool have = false;
        for (int i = 0; i < n_seen; ++i) have = have || (seen[i].fn == fn && seen[i].dev == dev);
        if (!have) {
          GPZ_HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          if (n_seen < 256) seen[n_seen++] = Seen{fn, dev};
        }
      }
      hipLaunchKernelGGL(kernel, grid, block, lds, s, p);
      GPZ_LAUNCH_OK();
      return 0;
    };
    constexpr bool PIPE = decltype(pipe_c)::value;
    if (epilogue == EPI_STORE)
      return bt ? launch(gemm128_kernel<T, KV, NI, true, EPI_STORE, PIPE>, gemm_lds_bytes<T, KV, true>())
                : launch(gemm128_kernel<T, KV, NI, false, EPI_STORE, PIPE>, gemm_lds_bytes<T, KV, false>());
    if (epilogue == EPI_WBAR) {
      GPZ_REQUIRE(p.colscale && p.colvec && p.rowvec && p.aux && p.beta == (T)0, "gemm: W-bar epilogue needs its operands");
      return launch(gemm128_kernel<T, KV, NI, false, EPI_WBAR, PIPE>, gemm_lds_bytes<T, KV, false>());
    }
    if (epilogue == EPI_STORE_COLSCALE)
      return launch(gemm128_kernel<T, KV, NI, false, EPI_STORE_COLSCALE, PIPE>, gemm_lds_bytes<T, KV, false>());
    if (epilogue == EPI_STORE_STATS)
      return launch(gemm128_kernel<T, KV, NI, false, EPI_STORE_STATS, PIPE>, gemm_lds_bytes<T, KV, false>());
    return launch(gemm128_kernel<T, KV, NI, false, EPI_STATS, PIPE>, gemm_lds_bytes<T, KV, false>());
  };
  using std::integral_constant;
  // the pipelined k-loop needs two fragment sets in registers: fits the 128-VGPR budget of 4 waves per SIMD in fp32,
  // not in fp64 (accumulators and fragments are twice as wide)
  static const int pipe_mode = [] { const char* e = getenv("GPZ_GEMM_PIPE"); return e ? atoi(e) : 1; }();
  if constexpr (sizeof(T) == 4) {
    if (pipe_mode) return run(integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<bool, true>{});
  }
  return run(integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<bool, false>{});
}

template int gemm_launch<float>(const GemmParams<float>&, int, hipStream_t);
template int gemm_launch<double>(const GemmParams<double>&, int, hipStream_t);

}  // namespace gpz
