// gemm.hip -- 128x128-tile batched GEMM on the CDNA4 matrix cores.
//
// fp32: v_mfma_f32_16x16x4_f32, fp64: v_mfma_f64_16x16x4_f64 (exact IEEE FMA
// chains, so results are deterministic and match a plain fp32/fp64 reference to
// rounding).  512 threads = 8 waves in a 2x4 grid, 64x32 outputs per wave held
// in 8 accumulator tiles.  A and (transposed) B tiles are staged [row][k] in LDS
// and fetched with one 16-byte ds_read per lane covering VEC consecutive k; the
// non-transposed B tile is staged [k][n].  Because the MFMA sums its four k
// slots, lane group q may own k = VEC*q .. VEC*q+VEC-1 as long as A and B agree,
// which is what makes the wide read legal.  Tiles reach LDS through registers (global load -> ds_write),
// double buffered, one barrier per k-tile; the kernel template also carries an LDS-DMA staging form (STG = 1), which
// the two big fp64 products of the forward pass use (gemm_launch), and 64 x 64 wave tiles (NI = 4) that were measured
// in round 2 and are not instantiated.  The two big fp32 products of the forward pass run on
// csrc/gemmw.hip instead.
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


// Timing-only diagnostics (wrong results!): -DGPZ_ABL=<bits> drops parts of the fp32 k-loop to see what the MFMA pipes
// wait for -- 1: no global loads after the first tile, 2: no LDS staging stores / no LDS-DMA, 4: no per-tile barrier.
// tools/ablate_gemm.sh builds such variants next to the real library.
#ifndef GPZ_ABL
#define GPZ_ABL 0
#endif
#ifndef GPZ_SCHED
#define GPZ_SCHED 0
#endif

// ---- LDS geometry ----------------------------------------------------------------------------------------------
// A staged tile is 128 rows x BK of A and BK x 128 of op(B), BK = two 64-byte k-chunks (32 fp32 / 16 fp64), so a
// [row][k] tile row is exactly 128 bytes = eight 16-byte chunks in both precisions.
//
// STG = 0 (default): register staging into padded images (16-byte row pad / 4-8 element pad: 2-way on the wide reads).
// STG = 1: tiles are filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write, no wait for load
// data before an LDS store; 16 fewer VGPRs).  Timing-only builds show that staging costs the MFMA pipes ~10 % of their
// time in this kernel; LDS-DMA moves the same bytes over the same L2 -> LDS path and measures the same as register
// staging (gemm_launch), so the cost is the movement, not the instructions.  One wave instruction writes 1 KB of LDS
// contiguously (lane i -> base + 16 i), so the images are lane-linear and every bank-conflict measure sits on the
// per-lane SOURCE address or between 1-KB pieces:
//   [row][k] tiles (A; B when stored (N,K)): unpadded, 8 rows per piece, chunk c of row w stored at slot c ^ (w & 7)
//     -- the 16 lanes of a ds_read_b128 group then cover all 64 banks exactly once (conflict free; the padded image of
//     the register-staged variant is 2-way);
//   [k][n] tile (B stored (K,N)): a k-row is 128 elements = 512 B / 1 KB, a piece holds 2 / 1 rows, and pieces sit
//     32 / 64 bytes apart so that the two k-rows a 32-lane read group touches (k and k + VEC) fall on disjoint banks.
template <typename T, bool BT, int STG>
struct GemmLds {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int BK = 8 * VEC;
  static constexpr int LDR = STG ? BK : BK + VEC;                        // [row][k] row pitch (elements)
  static constexpr int LDN = 128 + (VEC == 4 ? 4 : 8);                   // STG 0: [k][n] row pitch
  static constexpr int RPP = (1024 / (int)sizeof(T)) / 128;             // STG 1: k-rows per 1-KB piece of a [k][n] tile
  static constexpr int PIECE = 1024 / (int)sizeof(T) + (sizeof(T) == 4 ? 8 : 8);   // piece pitch: 1 KB + 32 B / 64 B
  static constexpr int A_ELEMS = 128 * LDR;
  static constexpr int B_ELEMS = BT ? 128 * LDR : (STG ? (BK / RPP) * PIECE : BK * LDN);
  static constexpr size_t bytes = sizeof(T) * 2 * (A_ELEMS + B_ELEMS);  // two buffers
};

// 8 waves (2 x 4) of 64 x 32 outputs, 4 waves per SIMD, 2 workgroups per CU in both precisions.  fp64 MFMAs only
// reach their rate with >= 3 waves per SIMD, which the 128 accumulator registers of a 64 x 64 fp64 wave tile rule out.
// NI = 16-column sub-tiles per wave: 2 -> 8 waves (2 x 4) of 64 x 32, four waves per SIMD; 4 -> 4 waves (2 x 2) of
// 64 x 64, two waves per SIMD (one per workgroup: no two waves of a SIMD share a barrier) and a third less LDS read
// traffic per flop.
// MULTI: a workgroup may compute several column tiles back to back (GemmParams::tiles_per_wg); a separate instantiation
// because the outer loop costs registers (the fp64 kernels would spill).
template <typename T, int NI, bool BT, int EPI, int STG, bool MULTI = false>
__global__ __launch_bounds__(1024 / NI) __attribute__((amdgpu_waves_per_eu(8 / NI, 8 / NI))) void gemm128_kernel(const GemmParams<T> p) {
  constexpr int WN = 8 / NI;                // waves along N
  constexpr int NT = 128 * WN;
  using M = Mma<T>;
  using vec_t = typename M::vec_t;
  using acc_t = typename M::acc_t;
  using G = GemmLds<T, BT, STG>;
  constexpr int VEC = M::VEC;
  constexpr int KV = 2;                     // 64-byte k-chunks per staged tile
  constexpr int BK = G::BK;
  constexpr int LDR = G::LDR, LDN = G::LDN, RPP = G::RPP, PIECE = G::PIECE;
  constexpr int A_ELEMS = G::A_ELEMS, B_ELEMS = G::B_ELEMS;
  constexpr bool ABL = GPZ_ABL != 0 && sizeof(T) == 4;
  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  T* const smem = reinterpret_cast<T*>(smem_raw);
  auto sA = [&](int buf) -> T* { return smem + buf * (A_ELEMS + B_ELEMS); };
  auto sB = [&](int buf) -> T* { return smem + buf * (A_ELEMS + B_ELEMS) + A_ELEMS; };

  // ---------------- tile decode ----------------
  int b0, b1, ti, tj;
  int ntile = 1, tj_step = 0;   // column tiles this workgroup computes, and their distance in tiles
  {
    const int bid = blockIdx.x;
    if (p.flags & GF_GROUP_COLS) {
      // Super-tiles for L2 reuse: one XCD (blocks b, b+8, ... share an XCD's L2) works through
      // (latent, strip of SUPER_COLS column tiles) units; inside a unit, row tiles go longest
      // k-range first and the SUPER_COLS blocks of one row tile are dispatched together, so they
      // stream the same A panel in lock-step while the unit's B panels stay L2 resident.
      // With tiles_per_wg = TPW > 1 a workgroup walks TPW column tiles of its row tile one after the other (columns w,
      // w + W, ... of the strip, W = SUPER_COLS / TPW workgroups per row tile): same A panel, same k-range, and the
      // first loads of the next tile are in flight while the epilogue of the current one runs.
      const int SUPER_COLS = p.super_cols;
      const int W = MULTI ? SUPER_COLS / p.tiles_per_wg : SUPER_COLS;
      const int x = bid & 7, s = bid >> 3;
      const int strips = (p.nt + SUPER_COLS - 1) / SUPER_COLS;
      const int per_unit = p.mt * W;
      const int unit = (s / per_unit) * 8 + x, within = s % per_unit;
      if (unit >= p.nb0 * strips) return;
      b0 = unit / strips; b1 = 0;
      const int ii = within / W;
      tj = (unit - b0 * strips) * SUPER_COLS + within % W;
      if (tj >= p.nt) return;
      if (MULTI) {
        tj_step = W;
        ntile = min(p.tiles_per_wg, (p.nt - tj + W - 1) / W);
      }
      ti = (p.flags & GF_A_LOWER) ? p.mt - 1 - ii : ii;   // longest k-range first
    } else {
      // Blocks b, b + 8, ... run on one XCD (round-robin dispatch): give XCD x a CONTIGUOUS range of the logical tile
      // order (batch-major, then row-major tiles), so the tiles that share an operand panel meet in one L2 instead of
      // pulling it into all eight.
      int lid = bid;
      if (p.xcd_contiguous) {
        const int nb = (int)gridDim.x, q = nb >> 3, rem = nb & 7, x = bid & 7;
        lid = x * q + min(x, rem) + (bid >> 3);
      }
      const int per = (p.flags & GF_TILES_LOWER) ? p.mt * (p.mt + 1) / 2 : p.mt * p.nt;
      const int b = lid / per;
      int t = lid - b * per;
      b0 = b / p.nb1; b1 = b - b0 * p.nb1;
      if (p.flags & GF_TILES_LOWER) {
        int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        ti = i; tj = t - i * (i + 1) / 2;
      } else if (p.flags & (GF_B_LOWER | GF_B_UPPER)) {
        // the k-range shrinks with the column tile: column-major, longest range first (greedy placement then ends on
        // the short tiles instead of starting on them)
        tj = t / p.mt; ti = t - tj * p.mt;
        if (p.flags & GF_B_UPPER) tj = p.nt - 1 - tj;
      } else {
        ti = t / p.nt; tj = t - ti * p.nt;
        if (p.flags & GF_A_LOWER) ti = p.mt - 1 - ti;     // longest k-range first
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

  // ---------------- staging: this thread's / wave's share of a tile ----------------
  // STG 0: two 16-byte vectors of A and of B per thread, through registers.
  // STG 1: two 1-KB pieces of A and of B per wave, by LDS-DMA; the lane's source address carries the swizzle.
  constexpr int RP = NI;                    // staging passes per thread (STG 0) / 1-KB pieces per wave (STG 1), per operand
  const T* pa[RP];
  const T* pb[RP];
  const int64_t b_step = BT ? (int64_t)BK : (int64_t)BK * p.ldb;
  // STG 1: wave-uniform tile bases (advanced by scalar adds) + constant 32-bit lane offsets: no vector ALU work per tile
  const T* a_base = Ag + k_begin;
  const T* b_base = BT ? Bg + k_begin : Bg + (int64_t)k_begin * p.ldb;
  uint32_t a_off[RP], b_off[RP];
  // STG 0 maps
  constexpr int VPR = BK / VEC, RROWS = NT / VPR;     // [row][k]: 8 vectors per row, 64 rows per pass
  constexpr int NV = 128 / VEC, BROWS = NT / NV;      // [k][n]: 32 / 64 vectors per row
  static_assert(128 / RROWS == RP && BK / BROWS == RP, "staging passes per tile");
  const int ra_row = tid / VPR, ra_vc = (tid % VPR) * VEC;
  const int rb_row = tid / NV, rb_vc = (tid % NV) * VEC;
  // STG 1 maps: piece RP * wave + h; [row][k]: lane -> (row lane / 8, slot lane % 8 holding chunk slot ^ row);
  // [k][n]: lane -> (k-row lane / CH, chunk lane % CH)
  constexpr int CH = 128 / VEC;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int h = 0; h < RP; ++h) {
    if (STG) {
      const int piece = RP * wave_u + h;
      const int prow = piece * 8 + (lane >> 3), pch = ((lane & 7) ^ (lane >> 3)) * VEC;
      a_off[h] = (uint32_t)(prow * (int)p.lda + pch);
      b_off[h] = BT ? (uint32_t)(prow * (int)p.ldb + pch) : (uint32_t)((piece * RPP + lane / CH) * (int)p.ldb + (lane % CH) * VEC);
      pa[h] = pb[h] = nullptr;
    } else {
      pa[h] = Ag + (int64_t)(ra_row + RROWS * h) * p.lda + k_begin + ra_vc;
      pb[h] = BT ? Bg + (int64_t)(ra_row + RROWS * h) * p.ldb + k_begin + ra_vc
                 : Bg + (int64_t)(k_begin + rb_row + BROWS * h) * p.ldb + rb_vc;
    }
  }
  bool abl_started = false;          // diagnostics only (GPZ_ABL): the prologue's first tile is always staged
  vec_t ga[RP], gb[RP];
  // STG 0: global -> registers.  STG 1: global -> LDS buffer `buf`, asynchronously (tracked by vmcnt).
  auto stage_load = [&](int buf) __attribute__((always_inline)) {
    if (ABL && (GPZ_ABL & (STG ? 2 : 1)) && abl_started) return;
#pragma unroll
    for (int h = 0; h < RP; ++h) {
      if constexpr (STG != 0) {
#if defined(__HIP_DEVICE_COMPILE__)   // a gfx950 builtin: the host pass of this single-source file only needs the kernel's stub
        typedef __attribute__((address_space(3))) void lds_void;
        const int piece = RP * wave_u + h;
        __builtin_amdgcn_global_load_lds(a_base + a_off[h], (lds_void*)(sA(buf) + piece * (1024 / (int)sizeof(T))), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b_base + b_off[h], (lds_void*)(sB(buf) + piece * (BT ? 1024 / (int)sizeof(T) : PIECE)), 16, 0, 0);
#endif
      } else {
        ga[h] = *reinterpret_cast<const vec_t*>(pa[h]);
        gb[h] = *reinterpret_cast<const vec_t*>(pb[h]);
        pa[h] += BK;
        pb[h] += b_step;
      }
    }
    if (STG) { a_base += BK; b_base += b_step; }
  };
  // STG 0: registers -> LDS buffer `buf`.  STG 1: wait for this wave's pieces to have landed.
  auto stage_commit = [&](int buf) __attribute__((always_inline)) {
    if constexpr (STG != 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (ABL && (GPZ_ABL & 2) && abl_started) return;
#pragma unroll
      for (int h = 0; h < RP; ++h)
        *reinterpret_cast<vec_t*>(sA(buf) + (ra_row + RROWS * h) * LDR + ra_vc) = ga[h];
      if (BT) {
#pragma unroll
        for (int h = 0; h < RP; ++h)
          *reinterpret_cast<vec_t*>(sB(buf) + (ra_row + RROWS * h) * LDR + ra_vc) = gb[h];
      } else {
#pragma unroll
        for (int h = 0; h < RP; ++h)
          *reinterpret_cast<vec_t*>(sB(buf) + (rb_row + BROWS * h) * LDN + rb_vc) = gb[h];
      }
    }
  };
  using std::integral_constant;

  acc_t acc[4][NI];

  // fragment addresses (elements): lane (r, q) owns k = VEC * q + j, j < VEC, of each 64-byte k-chunk
  //   STG 1 [row][k]: row w = base + r, chunk kc * 4 + q at slot (kc * 4 + q) ^ (w & 7) = ((q ^ (r & 7)) ^ (4 kc))
  const int fr_a = STG ? (wm * 64 + r) * LDR + ((q ^ (r & 7)) * VEC) : (wm * 64 + r) * LDR + q * VEC;
  const int fr_bt = STG ? (wn * 16 * NI + r) * LDR + ((q ^ (r & 7)) * VEC) : (wn * 16 * NI + r) * LDR + q * VEC;
  //   [k][n]: STG 1 k-row k lives in piece k / RPP at row k % RPP; q * VEC is a multiple of RPP
  const int fr_bn = STG ? (q * VEC / RPP) * PIECE + wn * 16 * NI + r : (q * VEC) * LDN + wn * 16 * NI + r;

  // One staged tile of MFMAs over the 16-row sub-tiles MI_LO..MI_HI of this wave (compile-time range).  All fragment
  // reads of the tile are issued up front (16 LDS instructions, 48 registers) and the MFMAs follow behind counted
  // waits, so the LDS round trip is paid once per tile and overlaps the first MFMAs instead of recurring in front of
  // every k-step (left to itself hipcc emits read - wait - 8 MFMAs eight times per tile for this loop).
  // The range may differ between the tile's two k-chunks (lo_c/hi_c: chunk 0, lo1_c/hi1_c: chunk 1): in the diagonal
  // block of a triangular A the zero boundary moves by one 16-row sub-tile per 16 k, i.e. per fp32 chunk.
  auto compute = [&](int buf, auto lo_c, auto hi_c, auto lo1_c, auto hi1_c, auto mid) __attribute__((always_inline)) {
    constexpr int LO[2] = {decltype(lo_c)::value, decltype(lo1_c)::value}, HI[2] = {decltype(hi_c)::value, decltype(hi1_c)::value};
    constexpr int MI_LO = LO[0] < LO[1] ? LO[0] : LO[1], MI_HI = HI[0] > HI[1] ? HI[0] : HI[1];
    vec_t fa[KV][4];
    vec_t fbv[KV][NI];
    T fbs[KV][VEC][NI];
#pragma unroll
    for (int kc = 0; kc < KV; ++kc) {
      const int ko = STG ? 0 : kc * 4 * VEC;                 // STG 1: the chunk index is folded into the slot (xor)
      const int fa_off = STG ? (fr_a ^ (kc * 4 * VEC)) : fr_a + ko;
#pragma unroll
      for (int mi = MI_LO; mi <= MI_HI; ++mi)
        if (mi >= LO[kc] && mi <= HI[kc]) fa[kc][mi] = *reinterpret_cast<const vec_t*>(sA(buf) + fa_off + mi * 16 * LDR);
      if (BT) {
        const int fb_off = STG ? (fr_bt ^ (kc * 4 * VEC)) : fr_bt + ko;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          fbv[kc][ni] = *reinterpret_cast<const vec_t*>(sB(buf) + fb_off + ni * 16 * LDR);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int kk = kc * 4 * VEC + j;                   // k-row of lane group q = 0
          const int off = STG ? (kk / RPP) * PIECE + (kk % RPP) * 128 : kk * LDN;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) fbs[kc][j][ni] = sB(buf)[fr_bn + off + ni * 16];
        }
      }
    }
#pragma unroll
    for (int kc = 0; kc < KV; ++kc) {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int mi = MI_LO; mi <= MI_HI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (mi < LO[kc] || mi > HI[kc]) continue;        // folded at compile time: the loops are fully unrolled
            if constexpr (BT) acc[mi][ni] = M::mma(fa[kc][mi][j], fbv[kc][ni][j], acc[mi][ni]);
            else acc[mi][ni] = M::mma(fa[kc][mi][j], fbs[kc][j][ni], acc[mi][ni]);
          }
      if (kc == 0) mid();                                    // GPZ_SCHED 5: the next tile's LDS stores, mid-tile
    }
    // Instruction order of the tile (GPZ_SCHED, timing experiments): 0 = hipcc's own, 1 = every read first, 2 = the
    // first k-step's reads, then three bursts behind the first MFMAs, 3 = first k-step's reads, then one read per MFMA,
    // 4 = all A fragments + the first B operand, then one B read per k-step, 5 = hipcc's own order with the staging
    // stores of the next tile placed between the two k-chunks instead of behind the last MFMA
    {
      constexpr int NMI = MI_HI - MI_LO + 1;
      constexpr int FIRST = NMI + (BT ? NI : NI / 2);       // A fragments of chunk 0 + the first B operand (b32 pairs merged)
      constexpr int REST = KV * NMI + (BT ? KV * NI : KV * VEC * NI) - FIRST;   // upper bound (b32 pairs may merge)
      if constexpr (GPZ_SCHED == 1) {
        __builtin_amdgcn_sched_group_barrier(0x100, 32, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 128, 0);
      } else if constexpr (GPZ_SCHED == 2) {
        __builtin_amdgcn_sched_group_barrier(0x100, FIRST, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (REST + 2) / 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (REST + 2) / 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, REST, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 128, 0);
      } else if constexpr (GPZ_SCHED == 3) {
        __builtin_amdgcn_sched_group_barrier(0x100, FIRST, 0);
#define GPZ_S3(i) if constexpr (REST > i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        GPZ_S3(0) GPZ_S3(1) GPZ_S3(2) GPZ_S3(3) GPZ_S3(4) GPZ_S3(5) GPZ_S3(6) GPZ_S3(7) GPZ_S3(8) GPZ_S3(9) GPZ_S3(10) GPZ_S3(11)
        GPZ_S3(12) GPZ_S3(13) GPZ_S3(14) GPZ_S3(15) GPZ_S3(16) GPZ_S3(17) GPZ_S3(18) GPZ_S3(19)
#undef GPZ_S3
        __builtin_amdgcn_sched_group_barrier(0x008, 128, 0);
      } else if constexpr (GPZ_SCHED == 4) {
        constexpr int NA = KV * NMI;                         // every A fragment of the tile
        constexpr int NB = BT ? KV * NI : KV * VEC * NI / 2; // B read instructions (b32 pairs merged)
        __builtin_amdgcn_sched_group_barrier(0x100, FIRST, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NA - NMI, 0);
#define GPZ_S4(i) if constexpr (NB > i) { __builtin_amdgcn_sched_group_barrier(0x008, 7, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        GPZ_S4(1) GPZ_S4(2) GPZ_S4(3) GPZ_S4(4) GPZ_S4(5) GPZ_S4(6) GPZ_S4(7) GPZ_S4(8) GPZ_S4(9) GPZ_S4(10) GPZ_S4(11)
        GPZ_S4(12) GPZ_S4(13) GPZ_S4(14) GPZ_S4(15)
#undef GPZ_S4
        __builtin_amdgcn_sched_group_barrier(0x008, 128, 0);
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
  if (nk > 0) stage_load(0);
  int t = 0;
  for (int ct = 0;; ++ct) {       // column tiles of this workgroup
  const int tjc = tj + ct * tj_step;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = acc_t{0, 0, 0, 0};
  if (nk > 0) stage_commit(0);
  __syncthreads();
  abl_started = true;
  t = 0;
  // Iteration t: start fetching tile t + 1 (STG 1: straight into the other LDS buffer, which every wave finished
  // reading before the barrier that ended iteration t - 1), run tile t (when `run`), then make tile t + 1 visible
  // (STG 0: registers -> LDS; STG 1: wait for this wave's pieces) and meet at the barrier.  Every region below starts at an
  // even t and has an even length (PT tiles per 64 k), so the LDS buffer is a compile-time constant of each iteration.
  static_assert(PT % 2 == 0, "staged tiles come in pairs");
  auto iteration = [&](auto par_c, auto run_c, auto lo_c, auto hi_c, auto lo1_c, auto hi1_c) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;                  // t & 1
    if (t + 1 < nk) stage_load(P ^ 1);
    if constexpr (decltype(run_c)::value && GPZ_SCHED == 5) {
      compute(P, lo_c, hi_c, lo1_c, hi1_c, [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) stage_commit(P ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      if constexpr (decltype(run_c)::value) compute(P, lo_c, hi_c, lo1_c, hi1_c, [] {});
      if (t + 1 < nk) stage_commit(P ^ 1);
    }
    if (!(ABL && (GPZ_ABL & 4) && decltype(run_c)::value)) __syncthreads();
    ++t;
  };
  using no_run = integral_constant<bool, false>;
  using run = integral_constant<bool, true>;
  using i0 = integral_constant<int, 0>;
  using i1 = integral_constant<int, 1>;
  using i3 = integral_constant<int, 3>;
  // chunk c (c = 0, 1) of tile u of the diagonal sub-block covers k in [u*BK + c*BK/2, u*BK + (c+1)*BK/2): it needs the
  // 16-row sub-tiles mi >= (first k)/16 when A is lower triangular, mi <= (last k)/16 when it is upper triangular
  constexpr int HK = BK / 2;                 // k per chunk
  for (int u = 0; u < n_pre; u += 2) { iteration(i0{}, no_run{}, i0{}, i3{}, i0{}, i3{}); iteration(i1{}, no_run{}, i0{}, i3{}, i0{}, i3{}); }
  if (part_hi) {
    auto phases = [&](auto self, auto u_c) __attribute__((always_inline)) -> void {
      constexpr int U = decltype(u_c)::value;
      iteration(integral_constant<int, U & 1>{}, run{}, i0{}, integral_constant<int, (U * BK + HK - 1) / 16>{}, i0{},
                integral_constant<int, ((U + 1) * BK - 1) / 16>{});
      if constexpr (U + 1 < PT) self(self, integral_constant<int, U + 1>{});
    };
    phases(phases, i0{});
  }
  const int t_main_end = nk - n_post - (part_lo ? PT : 0);
  while (t < t_main_end) { iteration(i0{}, run{}, i0{}, i3{}, i0{}, i3{}); iteration(i1{}, run{}, i0{}, i3{}, i0{}, i3{}); }
  if (part_lo) {
    auto phases = [&](auto self, auto u_c) __attribute__((always_inline)) -> void {
      constexpr int U = decltype(u_c)::value;
      iteration(integral_constant<int, U & 1>{}, run{}, integral_constant<int, (U * BK) / 16>{}, i3{},
                integral_constant<int, (U * BK + HK) / 16>{}, i3{});
      if constexpr (U + 1 < PT) self(self, integral_constant<int, U + 1>{});
    };
    phases(phases, i0{});
  }
  while (t < nk) { iteration(i0{}, no_run{}, i0{}, i3{}, i0{}, i3{}); iteration(i1{}, no_run{}, i0{}, i3{}, i0{}, i3{}); }

  // next column tile: operands back to k_begin, B over by tj_step tiles; with register staging its first staged tile
  // travels while the epilogue below runs (LDS buffer 0 is only written after the barrier that ends the epilogue)
  const bool more = MULTI && ct + 1 < ntile;
  if (more && nk > 0) {
    const int64_t a_back = (int64_t)nk * BK;
    const int64_t b_fwd = (BT ? (int64_t)tj_step * 128 * p.ldb : (int64_t)tj_step * 128) - (int64_t)nk * b_step;
    if (STG) { a_base -= a_back; b_base += b_fwd; }
    else {
#pragma unroll
      for (int h = 0; h < RP; ++h) { pa[h] -= a_back; pb[h] += b_fwd; }
    }
    abl_started = false;
    if (STG == 0) stage_load(0);
  }
  // ---------------- epilogue ----------------
  const int64_t crow0 = (int64_t)ti * 128 + wm * 64;
  const int64_t ccol0 = (int64_t)tjc * 128 + wn * 16 * NI;
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
      const int64_t o = ((int64_t)b0 * p.mt + ti) * p.ncols + (int64_t)tjc * 128 + tid;
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
  if (!MULTI || !more) break;
  __syncthreads();                 // every wave is done with the epilogue's LDS scratch (it aliases tile buffer 0)
  if (STG != 0 && nk > 0) stage_load(0);   // LDS-DMA writes LDS directly: only now
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
    const int sc = p.super_cols, tpw = p.tiles_per_wg;
    GPZ_REQUIRE(sc >= 1, "gemm: super_cols must be >= 1");
    GPZ_REQUIRE(tpw >= 1 && sc % tpw == 0, "gemm: tiles_per_wg=%d must divide super_cols=%d", tpw, sc);
    if (tpw > 1) {
      GPZ_REQUIRE(!(p.flags & (GF_B_LOWER | GF_B_UPPER)), "gemm: tiles_per_wg > 1 needs a dense B");
      GPZ_REQUIRE(sizeof(T) == 4 && (epilogue == EPI_STORE_STATS || epilogue == EPI_STATS),
                  "gemm: tiles_per_wg > 1 is built for the fp32 statistics epilogues only");
    }
    const int64_t units = (int64_t)p.nb0 * ((p.nt + sc - 1) / sc);
    nblocks = (units + 7) / 8 * 8 * p.mt * (sc / tpw);
  } else {
    const int64_t per = (p.flags & GF_TILES_LOWER) ? (int64_t)p.mt * (p.mt + 1) / 2 : (int64_t)p.mt * p.nt;
    nblocks = per * p.nb0 * p.nb1;
  }
  GPZ_REQUIRE(nblocks < (1ll << 31), "gemm: grid too large");
  const bool bt = (p.flags & GF_B_TRANS) != 0;
  if (epilogue != EPI_STORE) GPZ_REQUIRE(!bt, "gemm: stats / column-scale epilogues are NN only");
  if (p.beta != (T)0) GPZ_REQUIRE(bt && epilogue == EPI_STORE, "gemm: accumulation (beta != 0) is built for C += A * B^T only");
  if (epilogue == EPI_STORE_COLSCALE) GPZ_REQUIRE(p.colscale && p.beta == (T)0, "gemm: column-scale epilogue needs factors and beta = 0");
  // Staging variant (GPZ_GEMM_STG): 0 = through registers (default), 1 = LDS-DMA.  Wave tile (GPZ_GEMM_NI, fp32): 2 = 64 x 32
  // (default), 4 = 64 x 64.  Measured on config 3 (stage 1 / stage 2 TF): registers 132.6 / 136.2, LDS-DMA 126.4 / 130.8
  // with hipcc's own instruction order and 131.7 / 136.2 with the order pinned (GPZ_SCHED=4); 64 x 64 wave tiles 129-131 /
  // 131-133 either way.  The staging data movement itself, not the instructions that carry it, is what costs the MFMA
  // pipes their ~10 % (timing-only builds without any staging: 144 / 149), so the simpler, longer-proven variant ships.
  const bool multi = (p.flags & GF_GROUP_COLS) && p.tiles_per_wg > 1;
  GemmParams<T> pp = p;
  pp.xcd_contiguous = 1;
  auto run = [&](auto stg_c, auto ni_c) -> int {
    constexpr int STG = decltype(stg_c)::value, NI = decltype(ni_c)::value;
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
      hipLaunchKernelGGL(kernel, grid, block, lds, s, pp);
      GPZ_LAUNCH_OK();
      return 0;
    };
    if (epilogue == EPI_STORE)
      return bt ? launch(gemm128_kernel<T, NI, true, EPI_STORE, STG>, GemmLds<T, true, STG>::bytes)
                : launch(gemm128_kernel<T, NI, false, EPI_STORE, STG>, GemmLds<T, false, STG>::bytes);
    if (epilogue == EPI_WBAR) {
      GPZ_REQUIRE(p.colscale && p.colvec && p.rowvec && p.aux && p.beta == (T)0, "gemm: W-bar epilogue needs its operands");
      return launch(gemm128_kernel<T, NI, false, EPI_WBAR, STG>, GemmLds<T, false, STG>::bytes);
    }
    if (epilogue == EPI_STORE_COLSCALE)
      return launch(gemm128_kernel<T, NI, false, EPI_STORE_COLSCALE, STG>, GemmLds<T, false, STG>::bytes);
    if constexpr (sizeof(T) == 4 && NI == 2 && STG == 0) {
      if (multi && epilogue == EPI_STORE_STATS)
        return launch(gemm128_kernel<T, NI, false, EPI_STORE_STATS, STG, true>, GemmLds<T, false, STG>::bytes);
      if (multi && epilogue == EPI_STATS)
        return launch(gemm128_kernel<T, NI, false, EPI_STATS, STG, true>, GemmLds<T, false, STG>::bytes);
    }
    GPZ_REQUIRE(!multi, "gemm: tiles_per_wg > 1 is built for the fp32 statistics epilogues (register staging, 64 x 32 wave tiles)");
    if (epilogue == EPI_STORE_STATS)
      return launch(gemm128_kernel<T, NI, false, EPI_STORE_STATS, STG>, GemmLds<T, false, STG>::bytes);
    return launch(gemm128_kernel<T, NI, false, EPI_STATS, STG>, GemmLds<T, false, STG>::bytes);
  };
  using std::integral_constant;
  // one wave tile ships (64 x 32) and, but for one case, one staging form: through registers (the alternatives measured
  // slower or equal, see above)
  // ... except the two big fp64 products of the forward pass (the epilogues with column statistics), which take their
  // tiles by LDS-DMA: N=100 000, M=2048, L=8 fp64 (configs[4] per GPU) stage 1 67.5 -> 68.7 TF, stage 2 70.3 -> 70.8,
  // the evaluation 103.0 -> 101.9 ms.  (In fp32 the same switch lost 4 %: there the MFMAs are half as long and the
  // register-staged form's instruction order matters more; the fp32 products have their own kernel now, gemmw.hip.)
  if constexpr (sizeof(T) == 8) {
    if (epilogue == EPI_STORE_STATS || epilogue == EPI_STATS) return run(integral_constant<int, 1>{}, integral_constant<int, 2>{});
  }
  return run(integral_constant<int, 0>{}, integral_constant<int, 2>{});
}

template int gemm_launch<float>(const GemmParams<float>&, int, hipStream_t);
template int gemm_launch<double>(const GemmParams<double>&, int, hipStream_t);

}  // namespace gpz
