// gemmw.hip -- the two big fp32 products of the forward pass on wide workgroup tiles.
//
//   stage 1  Wt = Linv * Kzx      (gp.py:255 + :276: Kzx = kernel(Z, X); solve_triangular(L, Kzx))  A lower triangular
//   stage 2  colsum((LuE^T Wt)^2) (gp.py:280-296 / utilities.py:382-397)                            A upper triangular
//
// One kernel template, two instantiations of its tile:
//   * B from memory (stage 2; stage 1 of every kernel by default): 128 x 256 tile, 8 waves side by side.  Every wave of
//     the workgroup owns the same 128 rows, so all of them see the same k range of the triangular operand: no wave idles
//     while another finishes (a 256 x 128 tile of two row-waves, the first form of this kernel, left one of them idle for
//     8 steps per tile -- a quarter of all wave-steps at M = 512: configs[1] stage 1 118 -> 123 TF, stage 2 128 -> 131.5;
//     config 3 142.3 -> 142.6 / 150.4 -> 151.1, same box, both forms timed back to back).  In long launches without a
//     store epilogue (stage 2 at config 3) a workgroup runs TWO row tiles of its column tile, the pi-th longest and the
//     pi-th shortest k range, so every workgroup of the launch executes the same number of steps, and the second tile's
//     first operand tiles are fetched during the first tile's last step; elsewhere one tile, longest k range first.
//   * B GENERATED (stage 1, fp32 RBF / Matern-3/2 on 1-D / 2-D inputs): 512 x 128 tile, 16 waves, whose 16 x 128 slice of
//     Kzx is computed by the workgroup itself, ONCE, from the Z block in LDS and each lane's own columns -- Kzx is never
//     written to HBM (the reference and the two-kernel path move 52 GB of it per evaluation at N=200k, M=2048, L=32).
//     On gfx950 the f32 MFMA runs at the vector rate and vector instructions do NOT issue in its shadow (measured:
//     every VALU instruction added to the loop costs its own ~4-8 issue cycles of MFMA time, pinned interleaving or
//     not), so the covariance arithmetic is a tax proportional to (values per workgroup step) / (MFMAs per step)
//     = 16 / rows of the tile: generated per wave in registers for a 128-row wave tile it cost 19 %, shared through
//     LDS by 512 rows it costs 4 % -- to which a 1024-thread workgroup adds 3.5 % of barrier time (all four waves of a
//     SIMD belong to it and meet at every step), so this path stays behind the fill + product from memory and is
//     selectable, not the default.
//
// Common structure: 8 (16) waves, wave tile 128 x 32 (16 accumulator tiles of v_mfma_f32_16x16x4_f32), k staged 16 deep.
// Both operand tiles reach LDS by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write, no vector address
// arithmetic: descriptor + constant per-lane offset + scalar offset); A sits [row][16 k] with its four 16-byte chunks
// permuted per row so that the ds_read_b128 fragment reads are conflict-free, B sits [k][n] in 1-KB pieces 16 / 32
// bytes apart so that the four k rows one ds_read_b32 touches fall on disjoint banks.  Double buffered, one barrier per
// 16-deep step, <= 128 VGPRs; 50 KB of LDS and two workgroups per CU (from memory), 84 KB and one (generated).  The DMA
// is never what a step waits for (a timing build that issues the loads and never waits for them runs at the same
// speed); what the loads cost (4-5 %: the build without them) is their traffic.  Triangular A: a wave skips the steps in
// which its 128 rows are zero and, inside its diagonal 128-block, the 16-row sub-tiles that are -- as straight-line
// phases with compile-time ranges (hipcc copies accumulators around MFMAs that sit under run-time branches).
//
// Values, k order and MFMA order equal those of kfill.hip + gemm128_kernel (cov.h is shared; lane group q owns
// k = 4q .. 4q+3 of a 16-deep chunk in both), so Wt is bitwise the same on either path.
#include "gemmw.h"

#include "cov.h"


#include <cmath>
#include <mutex>
#include <type_traits>
#include <utility>

// Timing-only diagnostics (WRONG results by construction; tools/ablate_fused.sh builds them next to the real library):
// -DGPZ_W_ABL=<bits>  1: no covariance arithmetic (the generated B tile holds a coordinate), 2: no tile loads after the
// first, 4: no epilogue (statistics, Wt store), 8: no per-step barrier, 16: tile loads issued but never waited for.
#ifndef GPZ_W_ABL
#define GPZ_W_ABL 0
#endif

namespace gpz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { WB_MEM = 0, WB_GEN = 1 };
enum { WA_DENSE = 0, WA_LOWER = 1, WA_UPPER = 2 };
enum { WE_STORE_STATS = 0,      // C = A B, plus colsum(C^2) and mu^T C per 128-row block        (forward stage 1)
       WE_STATS = 1,            // colsum((A B)^2) per 128-row block only                         (forward stage 2)
       WE_STORE = 2,            // C = A B                                                        (backward: Kbar_x = Linv^T Wbar)
       WE_STORE_COLSCALE = 3,   // C[i][j] = colscale[j] (A B)[i][j]                              (backward: Pbar)
       WE_WBAR = 4,             // C[i][j] = (A B)[i][j] + rowvec[i] colvec[j] - aux[i][j] colscale[j]   (backward: Wbar)
       WE_KBAR = 5,             // C[i][j] = colscale[j] (A B)[i][j] + rowvec[i] colvec[j]               (backward: Kbar_x, dense A)
       WE_ADD_COLSCALE = 6 };   // C[i][j] += colscale[j] (A B)[i][j]                                    (its clamp correction)

struct WParams {
  const float* A; int64_t lda, sA0;         // (L, Mp, Mp)
  const float* B; int64_t ldb, sB0;         // WB_MEM: (L, Mp, ncols) row-major
  const float* Z; int64_t MD;               // WB_GEN: (M, D), MD = M * D
  const float* X; int64_t nreal;            //         (nreal, D)
  const float* sigma; const float* ell;     //         (L,)
  float* C; int64_t ldc, sC0;               // WE_STORE_STATS: (L, Mp, ncols)
  const float* mu; int64_t sMu;             // WE_STORE_STATS: (L, Mp)
  float* ps_sq; float* ps_mu; int64_t ncols;  // [L][nblk][ncols]
  const float* colscale; const float* colvec; int64_t sCs;   // WE_STORE_COLSCALE / WE_WBAR: (L, ncols) vectors
  const float* rowvec; int64_t sRv;           // WE_WBAR: (L, Mp)
  const float* aux;                           // WE_WBAR: laid out like C
  int64_t M;                                // real rows (the rest is padding)
  int L, nblk, mtw, nt, W, strips;          // latents, 128-blocks, row tiles, column tiles, strip width, strips
  int pair, colmajor;                       // WB_MEM: two row tiles per workgroup; dispatch order column tile by column tile
  const int32_t* gate;                      // device word: the launch does nothing when it is zero (null: always runs)
};

constexpr int W_BK = 16;

template <int TM, int TN, int BSRC, int D>
struct WLds {
  static constexpr int A_ELEMS = TM * W_BK;
  static constexpr int RPP = 256 / TN;                        // k rows per 1-KB piece of the B tile
  // floats between pieces: the k rows j, j + 4, j + 8, j + 12 one ds_read_b32 touches must fall 16 banks apart
  static constexpr int PITCH = 256 + (RPP == 2 ? 8 : RPP == 1 ? 4 : 16);
  static constexpr int B_ELEMS = (W_BK / RPP) * PITCH;
  static constexpr int STAGE = A_ELEMS + B_ELEMS;
  static constexpr int ZB = 128 * D;                          // WB_GEN: one 128-point block of Z per buffer
  static constexpr size_t bytes = sizeof(float) * (2 * STAGE + (BSRC == WB_GEN ? 2 * ZB : 0));
};

template <int TM, int TN, int BSRC, int ATRI, int EPI, int KIND, int D>
__global__ __launch_bounds__(64 * (TM / 128) * (TN / 32)) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemmw_kernel(const WParams p) {
  constexpr int BK = W_BK;
  if (p.gate && *p.gate == 0) return;
  constexpr int WMW = TM / 128, WNW = TN / 32;          // waves along rows / columns
  constexpr int NW = WMW * WNW;                         // waves: 8 (128 x 256, from memory) or 16 (512 x 128, generated)
  static_assert(NW == 8 || NW == 16, "8 or 16 waves of 128 x 32");
  using G = WLds<TM, TN, BSRC, D>;
  constexpr int RPP = G::RPP, PITCH = G::PITCH;
  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  float* const smem = reinterpret_cast<float*>(smem_raw);
  auto sA = [&](int buf) -> float* { return smem + buf * G::STAGE; };
  auto sB = [&](int buf) -> float* { return smem + buf * G::STAGE + G::A_ELEMS; };
  float* const sZ = smem + 2 * G::STAGE;       // WB_GEN: [2][ZB], inducing points of the 128-blocks (parity of the block)
  auto addrB = [](int k, int n) { return (k / RPP) * PITCH + (k % RPP) * TN + n; };

  // ---------------- workgroup decode ----------------
  // Blocks b, b + 8, ... run on one XCD (round-robin dispatch) and share its L2.
  // WB_GEN: the only operand in memory is A.  A unit = (row tile, latent, strip of W column tiles): its W workgroups
  //   stream one Linv row panel together; units go longest k range first, each level spread evenly over the XCDs.
  // WB_MEM: a unit = (latent, strip of W column tiles).  Short launches: inside it row tiles go longest k range first and
  //   the W workgroups of a row tile are dispatched together, so they walk the same A panel in lock-step over the strip's
  //   B panels.  Long launches (p.pair, p.colmajor; the host side decides): a workgroup takes TWO row tiles of one column
  //   tile, the pi-th longest and the pi-th shortest k range (a middle tile of an odd count alone) -- every workgroup of
  //   the launch then runs the same number of steps and the second tile's first operand tiles travel during the first
  //   tile's last step -- and the pairs of one column tile are dispatched together: they read the same B panel.
  int tj, b0, leg_ti[2], nlegs = 1;
  {
    const int bid = blockIdx.x;
    const int x = bid & 7, s = bid >> 3;
    if (BSRC == WB_GEN) {
      const int ug = s / p.W, within = s - ug * p.W;
      const int u = ug * 8 + x;
      const int per_level = p.L * p.strips;
      if (u >= p.mtw * per_level) return;
      const int level = u / per_level, rem = u - level * per_level;
      leg_ti[0] = leg_ti[1] = (ATRI == WA_LOWER) ? p.mtw - 1 - level : level;
      b0 = rem / p.strips;
      tj = (rem - b0 * p.strips) * p.W + within;
    } else {
      const int per_unit = (p.pair ? (p.mtw + 1) >> 1 : p.mtw) * p.W;
      const int u = (s / per_unit) * 8 + x, within = s % per_unit;
      if (u >= p.L * p.strips) return;
      b0 = u / p.strips;
      const int nrow = p.pair ? (p.mtw + 1) >> 1 : p.mtw;
      const int pi = p.colmajor ? within % nrow : within / p.W, far = p.pair ? p.mtw - 1 - pi : pi;
      // strips of equal width up to one column tile (an XCD's units then carry equal work): strip i is [i nt / strips, ...)
      const int strip = u - b0 * p.strips, c0 = strip * p.nt / p.strips, c1 = (strip + 1) * p.nt / p.strips;
      tj = c0 + (p.colmajor ? within / nrow : within % p.W);
      if (tj >= c1) return;
      // the longer k range first in one workgroup, last in the one that most likely shares its CU (an XCD's 32 CUs take
      // 32 consecutive workgroups of its sequence each): their epilogues and thin diagonal steps then do not coincide.
      // (Measured against always-long-first, always-short-first and the order that re-reads the B rows just read:
      // all within run-to-run noise, 140.1 - 141.4 TF on stage 1 of config 3.)
      const bool long_first = ((s >> 5) & 1) == 0;
      leg_ti[0] = ((ATRI == WA_LOWER) == long_first) ? far : pi;
      leg_ti[1] = ((ATRI == WA_LOWER) == long_first) ? pi : far;
      nlegs = far == pi ? 1 : 2;
      if (!p.pair) leg_ti[0] = leg_ti[1] = (ATRI == WA_LOWER) ? p.mtw - 1 - pi : pi;    // single tiles, longest k range first
    }
    if (tj >= p.nt) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WNW;
  const int r = lane & 15, q = lane >> 4;
  const int Mp = p.nblk * 128;

  // ---------------- staging: LDS-DMA ----------------
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr int NPA = TM / (16 * NW);           // 1-KB pieces (16 rows x 64 B) of the A tile per wave
#if defined(__HIP_DEVICE_COMPILE__)   // gfx950 builtins: the host pass of this single-source file only needs the kernel's stub
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A + b0 * p.sA0), 0, (int)(p.lda * Mp * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(BSRC == WB_MEM ? p.B + b0 * p.sB0 : p.Z), 0,
      (int)((BSRC == WB_MEM ? p.ldb * Mp : p.MD) * sizeof(float)), 0x00020000);   // reads past the end return zero
#endif
  auto k_begin_of = [&](int ti) { return (ATRI == WA_UPPER) ? TM * ti : 0; };
  auto k_end_of = [&](int ti) { return (ATRI == WA_LOWER) ? min(TM * (ti + 1), Mp) : Mp; };
  int a_soff[NPA], b_soff = 0;
  // chunk c of row w sits at slot c ^ g(w), g = [0, 2, 3, 1][(w >> 2) & 3]: the 16 lanes of a ds_read_b128 group
  // (rows r, chunk q) then cover sixteen distinct 16-byte slots of the 256-byte bank row
  auto gperm = [](int w) { const int t = (w >> 2) & 3; return (((t >> 1) ^ t) & 1) << 1 | (t >> 1); };
  const int a_voff = ((lane >> 2) * (int)p.lda + (((lane & 3) ^ gperm(lane >> 2)) * 4)) * (int)sizeof(float);
  // B from memory: wave w fetches the 1-KB pieces w, w + NW, ... of the step (TN = 256: a piece is one k row, two per wave)
  const int b_voff = ((lane / (TN / 4)) * (int)p.ldb + (lane % (TN / 4)) * 4) * (int)sizeof(float);
  bool abl_first = true;
  auto stage_load = [&](int buf) __attribute__((always_inline)) {
    if ((GPZ_W_ABL & 2) && !abl_first) return;
    abl_first = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int h = 0; h < NPA; ++h) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(sA(buf) + (NPA * wave + h) * 256), 16, a_voff, a_soff[h], 0, 0);
      a_soff[h] += BK * (int)sizeof(float);
    }
    if constexpr (BSRC == WB_MEM) {
      constexpr int NPB = (W_BK / RPP) / NW;      // 1-KB pieces of the B tile per wave: 1 (TN = 128) or 2 (TN = 256)
      static_assert(BSRC != WB_MEM || (NPB * NW * RPP == W_BK && NPB >= 1), "whole B pieces per wave");
#pragma unroll
      for (int h = 0; h < NPB; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void*)(sB(buf) + (wave + NW * h) * PITCH), 16, b_voff,
                                                 b_soff + h * NW * RPP * (int)p.ldb * (int)sizeof(float), 0, 0);
      b_soff += BK * (int)p.ldb * (int)sizeof(float);
    }
#endif
  };
  // point the DMA at the first step of row tile ti and start it (into buffer 0)
  auto tile_begin = [&](int ti) __attribute__((always_inline)) {
    const int kb = k_begin_of(ti);
#pragma unroll
    for (int h = 0; h < NPA; ++h) {
      int row = ti * TM + (NPA * wave + h) * 16;
      while (row >= Mp) row -= 128;             // rows past the matrix: re-read valid ones (their waves are inactive)
      a_soff[h] = (row * (int)p.lda + kb) * (int)sizeof(float);
    }
    b_soff = ((kb + RPP * wave) * (int)p.ldb + tj * TN) * (int)sizeof(float);
    stage_load(0);
  };

  // ---------------- WB_GEN: the B tile is this workgroup's own slice of Kzx ----------------
  // Inducing points: one 128-point block per LDS buffer (waves 0 .. 2D-1, one coordinate per lane), fetched one block
  // ahead.  Thread (wave w, lane l) computes rows w and w + 8 of the 16-deep step at columns l, l + 64, ...: its z is a
  // wave-uniform LDS broadcast, its x sits in registers for the whole tile.
  constexpr int NXC = TN / 64;                  // columns per lane
  float xc[NXC][D], ampn[NXC], c0n[NXC];
  CovConst cc = {0.f, 0.f, 0.f};
  if constexpr (BSRC == WB_GEN) {
    cc = cov_const<KIND>(p.sigma[b0], p.ell[b0]);
#pragma unroll
    for (int h = 0; h < NXC; ++h) {
      const int64_t n = (int64_t)tj * TN + h * 64 + lane;
      const bool real = n < p.nreal;
#pragma unroll
      for (int k = 0; k < D; ++k) xc[h][k] = real ? p.X[n * D + k] : 0.f;
      ampn[h] = real ? cc.amp : 0.f;            // padded columns: exactly zero, as the stand-alone fill writes them
      c0n[h] = (KIND == 1 && !real) ? 0.f : cc.c0;   // (Matern: c0 carries the amplitude too)
    }
  }
  auto z_load = [&](int blk) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (wave < 2 * D)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void*)(sZ + (blk & 1) * G::ZB + wave * 64), 4, lane * 4,
                                               (blk * G::ZB + wave * 64) * 4, 0, 0);
#endif
  };
  auto b_generate = [&](int buf, int step) __attribute__((always_inline)) {   // step: index of the 16-deep step from k = 0
#pragma unroll
    for (int i = 0; i < BK / NW; ++i) {
      const int kk = wave + NW * i, kg = step * BK + kk;
      const float* zp = sZ + ((kg >> 7) & 1) * G::ZB + (kg & 127) * D;
      float z[D];
#pragma unroll
      for (int k = 0; k < D; ++k) z[k] = zp[k];
#pragma unroll
      for (int h = 0; h < NXC; ++h) {
        const float v = (GPZ_W_ABL & 1) ? z[0] + xc[h][0]
                                        : cov_value<KIND>(cov_radial<KIND>(cov_d2<D>(z, xc[h])), ampn[h], c0n[h], cc.c1);
        sB(buf)[addrB(kk, h * 64 + lane)] = v;
      }
    }
  };

  f32x4 acc[8][2];

  // per row tile ("leg") of this workgroup: which 128-row block the wave takes, its k range, its fragment address
  int wm = 0, nk = 0, db = 0, n_a = 0, fr_a = 0, step0 = 0, t = 0, next_ti = -1;
  bool active = false;
  // fragment addresses (floats): lane (r, q) owns k = 4q .. 4q+3 of the 16-deep step
  int fr_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fr_b[j] = addrB(4 * q + j, wn * 32 + r);

  // One step of MFMAs over the 16-row sub-tiles LO .. HI (compile-time); the A fragments come in two halves of four
  // sub-tiles (16 registers live instead of 32).
  auto mma_step = [&](int buf, auto lo_c, auto hi_c) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    float bf[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[j][ni] = sB(buf)[fr_b[j] + ni * 16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 fa[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (half * 4 + m >= LO && half * 4 + m <= HI)
          fa[m] = *reinterpret_cast<const f32x4*>(sA(buf) + fr_a + (half * 4 + m) * 16 * BK);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            if (half * 4 + m >= LO && half * 4 + m <= HI)
              acc[half * 4 + m][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][j], bf[j][ni], acc[half * 4 + m][ni], 0, 0, 0);
    }
  };

  // ---------------- k loop ----------------
  // Step t: start the DMA of step t + 1 into the other buffer (and, WB_GEN, compute its B tile there), run step t, wait
  // for the DMA, barrier.  Every region below starts at an even t and has an even length (8 steps per 128-block), so the
  // buffer is a compile-time constant per step; no MFMA sits under a run-time branch.
  using std::integral_constant;
  auto step = [&](auto par_c, auto run_c, auto lo_c, auto hi_c) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;
    if (t + 1 < nk) {
      stage_load(P ^ 1);
      if constexpr (BSRC == WB_GEN) {
        const int st = step0 + t;
        if ((st & 7) == 0 && (st >> 3) + 1 < p.nblk) z_load((st >> 3) + 1);   // first step of a 128-block: fetch the next block
        b_generate(P ^ 1, st + 1);
      }
    } else if (next_ti >= 0) {
      // last step of this row tile (it reads buffer 1; buffer 0 is free since the previous step's barrier): the first
      // operand tiles of the workgroup's next row tile start now and are waited for with this step's own DMA wait, so
      // nothing after the epilogue has to wait on the memory counter (which would also wait for the epilogue's stores)
      tile_begin(next_ti);
    }
    if constexpr (decltype(run_c)::value) mma_step(P, lo_c, hi_c);
    if (GPZ_W_ABL & 16) {        // timing only: the DMA is issued but never waited for (what a deeper prefetch could hide at most)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(GPZ_W_ABL & 8)) __syncthreads();
    }
    ++t;
  };
  using no_run = integral_constant<bool, false>;
  using run = integral_constant<bool, true>;
  using i0 = integral_constant<int, 0>;
  using i1 = integral_constant<int, 1>;
  using i7 = integral_constant<int, 7>;
  auto idle_until = [&](int end) __attribute__((always_inline)) {
    while (t < end) { step(i0{}, no_run{}, i0{}, i7{}); step(i1{}, no_run{}, i0{}, i7{}); }
  };
  auto full_until = [&](int end) __attribute__((always_inline)) {
    while (t < end) { step(i0{}, run{}, i0{}, i7{}); step(i1{}, run{}, i0{}, i7{}); }
  };
  // diagonal 128-block: step u covers k = 16u .. 16u+15 of it.  A lower triangular: the rows above sub-tile u are zero
  // there (mi >= u); A upper triangular: the rows below it are (mi <= u).
  auto diagonal = [&](auto self, auto u_c) __attribute__((always_inline)) -> void {
    constexpr int U = decltype(u_c)::value;
    if constexpr (ATRI == WA_LOWER) step(integral_constant<int, U & 1>{}, run{}, integral_constant<int, U>{}, i7{});
    else step(integral_constant<int, U & 1>{}, run{}, i0{}, integral_constant<int, U>{});
    if constexpr (U + 1 < 8) self(self, integral_constant<int, U + 1>{});
  };

  tile_begin(leg_ti[0]);
#pragma unroll 1
  for (int leg = 0; leg < nlegs; ++leg) {
    const int ti = leg_ti[leg];
    // which 128-row block of the tile a wave takes alternates from tile to tile (blocks carry different amounts of work)
    wm = WMW == 1 ? 0 : (wave / WNW) ^ ((ti ^ tj) & 1);
    const int k_begin = k_begin_of(ti);
    nk = (k_end_of(ti) - k_begin) / BK;
    db = WMW * ti + wm;                         // this wave's 128-row block
    // a block count that is no multiple of WMW leaves the last row tile partly empty, a column count that is no multiple
    // of TN the last column tile (its operand reads past a row's end land in the next row or, past the matrix, return zero)
    active = db < p.nblk && (int64_t)tj * TN + wn * 32 < p.ncols;
    // steps of this wave: LOWER  [0, n_a) full, [n_a, n_a + 8) diagonal block, rest idle;
    //                     UPPER  [0, n_a) idle, [n_a, n_a + 8) diagonal block, rest full;   an inactive wave idles throughout
    n_a = 8 * (ATRI == WA_LOWER ? db : wm);
    fr_a = (wm * 128 + r) * BK + ((q ^ gperm(r)) * 4);
    step0 = k_begin / BK;
    t = 0;
    next_ti = leg + 1 < nlegs ? leg_ti[leg + 1] : -1;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};
    if constexpr (BSRC == WB_GEN) {
      z_load(step0 >> 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      b_generate(0, step0);
    }
    // The first operand tiles have landed (leg 0: wait here; later legs: they were waited for in the previous leg's last
    // step) and every wave is through the previous leg's epilogue, whose strips lie in buffer 1.  A bare barrier behind
    // the LDS counter: __syncthreads() would also drain the epilogue's global stores.
    if (leg == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) {
      idle_until(nk);
    } else if constexpr (ATRI == WA_DENSE) {
      full_until(nk);
    } else if constexpr (ATRI == WA_LOWER) {
      full_until(n_a);
      diagonal(diagonal, i0{});
      idle_until(nk);
    } else {
      idle_until(n_a);
      diagonal(diagonal, i0{});
      full_until(nk);
    }

    // ---------------- epilogue ----------------
    if (!active) continue;
    if (GPZ_W_ABL & 4) {         // keep the accumulators alive, store (practically) nothing
      float sum = 0.f;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) sum += acc[mi][ni][g];
      if (sum == 12345.678f) p.ps_sq[0] = sum;
      continue;
    }
    const int64_t row0 = (int64_t)db * 128;
    const int64_t ccol0 = (int64_t)tj * TN + wn * 32;
    // WB_GEN: rows >= M are padding -- Linv is the identity there, so they picked up k(0, x); the stand-alone path has zeros
    if (BSRC == WB_GEN && row0 + 128 > p.M) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (row0 + mi * 16 + 4 * q + g >= p.M) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni][g] = 0.f;
          }
    }
    // column statistics over this wave's 128 rows = one 128-row block: registers -> lane groups, no workgroup step
    if constexpr (EPI == WE_STORE_STATS || EPI == WE_STATS) {
      float ssq[2] = {0.f, 0.f}, smu[2] = {0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        f32x4 m4 = {0, 0, 0, 0};
        if constexpr (EPI == WE_STORE_STATS) m4 = *reinterpret_cast<const f32x4*>(p.mu + b0 * p.sMu + row0 + 4 * q + mi * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const float v = acc[mi][ni][g];
            ssq[ni] = __builtin_fmaf(v, v, ssq[ni]);
            if constexpr (EPI == WE_STORE_STATS) smu[ni] = __builtin_fmaf(m4[g], v, smu[ni]);
          }
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        ssq[ni] += __shfl_xor(ssq[ni], 16); ssq[ni] += __shfl_xor(ssq[ni], 32);
        if constexpr (EPI == WE_STORE_STATS) { smu[ni] += __shfl_xor(smu[ni], 16); smu[ni] += __shfl_xor(smu[ni], 32); }
        if (q == 0) {
          const int64_t o = ((int64_t)b0 * p.nblk + db) * p.ncols + ccol0 + ni * 16 + r;
          p.ps_sq[o] = ssq[ni];
          if constexpr (EPI == WE_STORE_STATS) p.ps_mu[o] = smu[ni];
        }
      }
    }
    // output tile: through a wave-private LDS strip of one 16-row sub-tile (the strips lie in tile buffer 1: every read of
    // it is behind the loop's last barrier, and the next leg's DMA has buffer 0) so each store instruction writes eight
    // whole 128-byte row segments
    if constexpr (EPI != WE_STATS) {
      constexpr int LDE = 36;
      static_assert(NW * 16 * LDE <= G::STAGE, "the epilogue strips fit tile buffer 1");
      float* strip = sA(1) + wave * (16 * LDE);
      float* Cg = p.C + b0 * p.sC0 + row0 * p.ldc + ccol0;
      int srow = lane >> 3;
      asm volatile("" : "+v"(srow));          // per leg: keeps the sixteen row addresses from being hoisted (and spilled)
      const int c4 = (lane & 7) * 4;
      float cs[2] = {1.f, 1.f};
      if constexpr (EPI == WE_STORE_COLSCALE) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) cs[ni] = p.colscale[b0 * p.sCs + ccol0 + ni * 16 + r];
      }
      f32x4 sc = {0, 0, 0, 0}, cv = {0, 0, 0, 0};
      if constexpr (EPI == WE_WBAR || EPI == WE_KBAR || EPI == WE_ADD_COLSCALE)
        sc = *reinterpret_cast<const f32x4*>(p.colscale + b0 * p.sCs + ccol0 + c4);
      if constexpr (EPI == WE_WBAR || EPI == WE_KBAR)
        cv = *reinterpret_cast<const f32x4*>(p.colvec + b0 * p.sCs + ccol0 + c4);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) strip[(4 * q + g) * LDE + ni * 16 + r] = cs[ni] * acc[mi][ni][g];
        __builtin_amdgcn_wave_barrier();
        f32x4 w[2];
        float rv[2];
        if constexpr (EPI == WE_WBAR || EPI == WE_KBAR || EPI == WE_ADD_COLSCALE) {   // every operand of the pass is loaded before its first store (stores may alias)
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int64_t row = mi * 16 + it * 8 + srow;
            if constexpr (EPI == WE_WBAR) w[it] = *reinterpret_cast<const f32x4*>(p.aux + b0 * p.sC0 + (row0 + row) * p.ldc + ccol0 + c4);
            if constexpr (EPI == WE_ADD_COLSCALE) w[it] = *reinterpret_cast<const f32x4*>(Cg + (int64_t)row * p.ldc + c4);
            if constexpr (EPI != WE_ADD_COLSCALE) rv[it] = p.rowvec[b0 * p.sRv + row0 + row];
          }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + srow;
          f32x4 v = *reinterpret_cast<const f32x4*>(strip + row * LDE + c4);
          if constexpr (EPI == WE_WBAR) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] + rv[it] * cv[e] - w[it][e] * sc[e];
          }
          if constexpr (EPI == WE_KBAR) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + rv[it] * cv[e];
          }
          if constexpr (EPI == WE_ADD_COLSCALE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = w[it][e] + v[e] * sc[e];
          }
          *reinterpret_cast<f32x4*>(Cg + (int64_t)(mi * 16 + row) * p.ldc + c4) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// ---------------- C += A diag(w) B^T over a long k (the backward pass's M x M gradient accumulation) ----------------
// H += W diag(gv2) W^T (gp.py:276-296 under autograd; svgp.hip explains why the backward pass needs only this one
// N-sized product for mu and Lu): A, B (L, Mp, K) row-major with K = the N-chunk, w (L, K) column weights or null,
// C (L, Mp, Mp), lower tiles only.  Same wave tiling and LDS-DMA staging as above; both operands are [row][16 k] images
// (B's fragments are ds_read_b128 too), every step runs all sub-tiles, and the epilogue is a read-modify-write of the
// 256 x 128 tile through the wave-private strips.  Tiles of one matrix take a contiguous range of blocks on ONE XCD
// (blocks b, b + 8, ... share an XCD): its workgroups walk k together and share the operand panels in that L2.
struct NTParams {
  const float* A; const float* B; float* C;
  int64_t ld, sAB, ldc, sC;                 // operand leading dimension (= K) / batch stride, C's
  int K, nblk, mt, L, T;                    // k extent, 128-blocks, 256-row tiles, matrices, tiles per matrix
  // few tiles (small M, few latents): the k extent is cut into S pieces of Ks (a multiple of 32) and a workgroup takes one
  // (tile, piece); its result goes to part[piece] (laid out like C, plain store) and nt_reduce_kernel adds the pieces to C
  int S, Ks;
  float* part; int64_t sPart;
  const float* w; int64_t sW;               // column weights (L, K): C += A diag(w) B^T (applied to B's fragments), or null
  const int32_t* gate;                      // device word: the launch does nothing when it is zero (null: always runs)
  // A gm (WTS only): per (piece, matrix) the product of A's rows with the vector gm (L, K), formed by the tiles on the
  // diagonal from the A fragments they hold anyway (dLoss/dmuE = W gm: a separate pass over W was a third of this
  // kernel's time at configs[1]); vpart [S][L][Mp], or null
  const float* gm; float* vpart;
};

// TM = 256: 8 waves, two 128-row blocks per tile; TM = 128 (few blocks: at four 128-blocks the 256-row tile computes 12
// block-slots for the 10 blocks of the lower triangle, at two blocks 4 for 3): 4 waves, one block per tile.
template <int TM, bool WTS>
__global__ __launch_bounds__(TM * 2) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemmw_nt_kernel(const NTParams p) {
  constexpr int BK = W_BK, TN = 128;
  if (p.gate && *p.gate == 0) return;
  constexpr int NW = TM / 32;                    // waves: (TM / 128) row blocks x 4 column strips
  constexpr int A_ELEMS = TM * BK, B_ELEMS = TN * BK, STAGE = A_ELEMS + B_ELEMS;
  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  float* const smem = reinterpret_cast<float*>(smem_raw);
  auto sA = [&](int buf) -> float* { return smem + buf * STAGE; };
  auto sB = [&](int buf) -> float* { return smem + buf * STAGE + A_ELEMS; };
  // block -> (matrix, tile): XCD x takes the contiguous range [x n / 8, (x + 1) n / 8) of the (matrix-major) tile order
  int b0, ti, tj, piece;
  {
    const int nb = (int)gridDim.x, qn = nb >> 3, rem = nb & 7, x = blockIdx.x & 7;
    const int lid0 = x * qn + min(x, rem) + (int)(blockIdx.x >> 3);
    // (matrix, piece of k, tile), tile fastest: the tiles of one (matrix, piece) are dispatched together on one XCD and walk
    // the same k range -- every row block of it is read by several of them (at four 128-blocks: 4 times) and only the first
    // read comes from HBM.  (Piece fastest, the first form, put the S pieces of ONE tile side by side: disjoint k ranges,
    // nothing shared, every operand byte fetched 4 times over: 3.3 GB per launch at configs[1] for 0.82 GB of Wt.)
    const int grp = lid0 / p.T;
    const int t = lid0 - grp * p.T;
    b0 = grp / p.S;
    piece = grp - b0 * p.S;
    if constexpr (TM == 256) {
      // row tile ti holds column tiles 0 .. 2 ti + 1 (the last row of an odd block count one fewer): ti (ti + 1) tiles precede it
      int i = (int)((sqrtf(4.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while ((i + 1) * (i + 2) <= t) ++i;
      while (i * (i + 1) > t) --i;
      ti = i; tj = t - i * (i + 1);
    } else {
      // row block ti holds column blocks 0 .. ti: ti (ti + 1) / 2 tiles precede it
      int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while ((i + 1) * (i + 2) / 2 <= t) ++i;
      while (i * (i + 1) / 2 > t) --i;
      ti = i; tj = t - i * (i + 1) / 2;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  const int Mp = p.nblk * 128;
  const int db = (TM / 128) * ti + wm;               // this wave's 128-row block
  const bool active = db < p.nblk && tj <= db;       // blocks above the diagonal (and past the matrix) are not computed
  // a diagonal tile of A diag(w) A^T: the B tile IS the A tile -- it is fetched once and B's fragments are read from A's image
  const bool same = TM == 128 && p.A == p.B && tj == ti;
  typedef __attribute__((address_space(3))) void lds_void;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A + b0 * p.sAB), 0, (int)(p.ld * Mp * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.B + b0 * p.sAB), 0, (int)(p.ld * Mp * sizeof(float)), 0x00020000);
#endif
  auto gperm = [](int w) { const int t = (w >> 2) & 3; return (((t >> 1) ^ t) & 1) << 1 | (t >> 1); };
  const int voff = ((lane >> 2) * (int)p.ld + (((lane & 3) ^ gperm(lane >> 2)) * 4)) * (int)sizeof(float);
  int a_soff[2], b_soff;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int row = ti * TM + (2 * wave + h) * 16;
    while (row >= Mp) row -= 128;
    a_soff[h] = (row * (int)p.ld + piece * p.Ks) * (int)sizeof(float);
  }
  constexpr int NPB = 8 / NW;                    // 16-row pieces of the B tile per wave: 1 (8 waves) or 2 (4 waves)
  b_soff = ((tj * TN + wave * NPB * 16) * (int)p.ld + piece * p.Ks) * (int)sizeof(float);
  auto stage_load = [&](int buf) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(sA(buf) + (2 * wave + h) * 256), 16, voff, a_soff[h], 0, 0);
      a_soff[h] += BK * (int)sizeof(float);
    }
    if (!same) {
#pragma unroll
      for (int h = 0; h < NPB; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void*)(sB(buf) + (wave * NPB + h) * 256), 16, voff,
                                                 b_soff + h * 16 * (int)p.ld * (int)sizeof(float), 0, 0);
      b_soff += BK * (int)sizeof(float);
    }
#endif
  };
  // the step's 16 column weights: lane group q owns k = 4q .. 4q+3 of a 16-deep step, so one 16-byte load per lane and
  // step.  ONE register set: a step's weights are consumed at its top (B's fragments are scaled as they are read) and
  // the next step's are requested right behind that -- the wait for the tile loads at the end of the step covers them.
  const float* wk = WTS ? p.w + b0 * p.sW + (int64_t)piece * p.Ks + 4 * (lane >> 4) : nullptr;
  f32x4 wv;
  auto w_load = [&]() __attribute__((always_inline)) {
    if constexpr (WTS) { wv = *reinterpret_cast<const f32x4*>(wk); wk += BK; }
  };
  const int fr_a = (wm * 128 + r) * BK + ((q ^ gperm(r)) * 4);
  const int nk = (min(p.K, (piece + 1) * p.Ks) - piece * p.Ks) / BK;      // even: K is a multiple of 128, Ks of 32
  const bool split = p.S > 1;
  // A block ON the diagonal of the symmetric product A diag(w) A^T (WTS, A == B): only its lower triangle is wanted (the
  // caller mirrors it), so the wave computes the 16 x 16 sub-tiles (mi, c) with mi >= c only -- 36 of the block's 64.
  // Wave wn takes the column sub-tiles wn and 7 - wn (8 - wn and wn + 1 row sub-tiles: 9 for every wave, against 16 for a
  // full 128 x 32 strip) as a code path of its own with compile-time ranges (no MFMA under a run-time branch); the steps and
  // barriers are those of the other waves.  It also forms its rows' share of A gm (sub-tiles 2 wn, 2 wn + 1) from the A
  // fragments it reads anyway: no separate pass over A for dLoss/dmuE.
  const bool diag = WTS && p.A == p.B && db == tj && db < p.nblk;      // wave-uniform
  if (diag) {
    auto diag_path = [&](auto wn_c) __attribute__((always_inline)) {
      constexpr int C0 = decltype(wn_c)::value, C1 = 7 - C0, N0 = 8 - C0, N1 = 8 - C1;
      const bool rd = p.gm != nullptr;
      const float* gk = rd ? p.gm + b0 * p.sW + (int64_t)piece * p.Ks + 4 * (lane >> 4) : nullptr;
      f32x4 gv = f32x4{0, 0, 0, 0};
      float pv0 = 0.f, pv1 = 0.f;
      f32x4 a0[N0], a1[N1];
#pragma unroll
      for (int i = 0; i < N0; ++i) a0[i] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < N1; ++i) a1[i] = f32x4{0, 0, 0, 0};
      const int fb_off = r * BK + ((q ^ gperm(r)) * 4) + (same ? -A_ELEMS : 0);
      auto step = [&](int buf, bool more) __attribute__((always_inline)) {
        f32x4 f0 = *reinterpret_cast<const f32x4*>(sB(buf) + fb_off + C0 * 16 * BK);
        f32x4 f1 = *reinterpret_cast<const f32x4*>(sB(buf) + fb_off + C1 * 16 * BK);
        f0 *= wv; f1 *= wv;
        if (more) w_load();
#pragma unroll
        for (int mi = C0; mi < 8; ++mi) {
          const f32x4 fa = *reinterpret_cast<const f32x4*>(sA(buf) + fr_a + mi * 16 * BK);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a0[mi - C0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], f0[j], a0[mi - C0], 0, 0, 0);
            if (mi >= C1) a1[mi >= C1 ? mi - C1 : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], f1[j], a1[mi >= C1 ? mi - C1 : 0], 0, 0, 0);
          }
          if (mi == 2 * C0 || mi == 2 * C0 + 1) {
            if (rd) {
              const f32x4 t = fa * gv;
              if (mi == 2 * C0) pv0 += (t[0] + t[1]) + (t[2] + t[3]); else pv1 += (t[0] + t[1]) + (t[2] + t[3]);
            }
          }
        }
        if (more && rd) { gv = *reinterpret_cast<const f32x4*>(gk); gk += BK; }
      };
      if (rd) { gv = *reinterpret_cast<const f32x4*>(gk); gk += BK; }
      stage_load(0);
      w_load();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int t = 0; t < nk; t += 2) {
        if (t + 1 < nk) stage_load(1);
        step(0, t + 1 < nk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < nk) stage_load(0);
        step(1, t + 2 < nk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (rd) {                    // rows r + 16 mi of the block: sum the four k-slot groups, lanes 0..15 write
        float* vp = p.vpart + ((int64_t)piece * p.L + b0) * Mp + db * 128;
        pv0 += __shfl_xor(pv0, 16); pv0 += __shfl_xor(pv0, 32);
        pv1 += __shfl_xor(pv1, 16); pv1 += __shfl_xor(pv1, 32);
        if (q == 0) { vp[(2 * C0) * 16 + r] = pv0; vp[(2 * C0 + 1) * 16 + r] = pv1; }
      }
      // accumulator (row 4q + g, column r) of sub-tile (mi, c): 64-byte row segments, once per tile
      float* Cg = (split ? p.part + piece * p.sPart : p.C) + b0 * p.sC + (int64_t)db * 128 * p.ldc + (int64_t)tj * TN;
      auto put = [&](const f32x4& a, int mi, int c) __attribute__((always_inline)) {
        float* o = Cg + (int64_t)(mi * 16 + 4 * q) * p.ldc + c * 16 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) o[g * p.ldc] = split ? a[g] : o[g * p.ldc] + a[g];
      };
#pragma unroll
      for (int mi = C0; mi < 8; ++mi) put(a0[mi - C0], mi, C0);
#pragma unroll
      for (int mi = C1; mi < 8; ++mi) put(a1[mi - C1], mi, C1);
    };
    switch (wn) {
      case 0: diag_path(std::integral_constant<int, 0>{}); break;
      case 1: diag_path(std::integral_constant<int, 1>{}); break;
      case 2: diag_path(std::integral_constant<int, 2>{}); break;
      default: diag_path(std::integral_constant<int, 3>{}); break;
    }
    return;
  }
  f32x4 acc[8][2];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};
  const int fr_b = (wn * 32 + r) * BK + ((q ^ gperm(r)) * 4) + (same ? -A_ELEMS : 0);
  auto mma_step = [&](int buf, bool more) __attribute__((always_inline)) {
    f32x4 fb[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      fb[ni] = *reinterpret_cast<const f32x4*>(sB(buf) + fr_b + ni * 16 * BK);
      if constexpr (WTS) fb[ni] *= wv;
    }
    if (more) w_load();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 fa[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) fa[m] = *reinterpret_cast<const f32x4*>(sA(buf) + fr_a + (half * 4 + m) * 16 * BK);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[half * 4 + m][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][j], fb[ni][j], acc[half * 4 + m][ni], 0, 0, 0);
    }
  };
  stage_load(0);
  w_load();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // (the wave-uniform `active` test sits outside the loops: no MFMA under a run-time branch)
  if (active) {
    for (int t = 0; t < nk; t += 2) {
      if (t + 1 < nk) stage_load(1);
      mma_step(0, t + 1 < nk);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 2 < nk) stage_load(0);
      mma_step(1, t + 2 < nk);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    for (int t = 0; t < nk; t += 2) {
      if (t + 1 < nk) stage_load(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 2 < nk) stage_load(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    return;
  }
  // read-modify-write of the tile (all loads of a pass are issued before its first store) -- or, with the k extent cut
  // into pieces, a plain store of this piece's partial tile
  constexpr int LDE = 36;
  float* strip = smem + wave * (32 * LDE);
  float* Cg = (split ? p.part + piece * p.sPart : p.C) + b0 * p.sC + (int64_t)db * 128 * p.ldc + (int64_t)tj * TN + wn * 32;
  const int srow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) strip[(mm * 16 + 4 * q + g) * LDE + ni * 16 + r] = acc[pass * 2 + mm][ni][g];
    __builtin_amdgcn_wave_barrier();
    f32x4 cin[4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
      cin[it] = split ? f32x4{0, 0, 0, 0} : *reinterpret_cast<const f32x4*>(Cg + (int64_t)(pass * 32 + it * 8 + srow) * p.ldc + c4);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + srow;
      const f32x4 v = *reinterpret_cast<const f32x4*>(strip + row * LDE + c4);
      *reinterpret_cast<f32x4*>(Cg + (int64_t)(pass * 32 + row) * p.ldc + c4) = cin[it] + v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// C += part[0] + part[1] + ... (ascending: reproducible) over the 128-blocks the product computes (column block <= row block)
__global__ __launch_bounds__(256) void nt_reduce_kernel(float* __restrict__ C, const float* __restrict__ part, int S, int64_t sPart,
                                                        int Mp, int64_t total4, const int32_t* __restrict__ gate, int sym) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= total4 || (gate && *gate == 0)) return;
  const int64_t e = i4 * 4;
  const int col = (int)(e % Mp), row = (int)((e / Mp) % Mp);
  // (symmetric product: the blocks on the diagonal hold their lower 16 x 16 sub-tiles only)
  if ((col >> 7) > (row >> 7) || (sym && (col >> 4) > (row >> 4))) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(C + e);
  for (int s = 0; s < S; ++s) v += *reinterpret_cast<const f32x4*>(part + s * sPart + e);
  *reinterpret_cast<f32x4*>(C + e) = v;
}

// v[l][m] = sum over the pieces (ascending) of vpart[piece][l][m], in fp64
__global__ __launch_bounds__(256) void nt_vsum_kernel(const float* __restrict__ vpart, int S, int64_t LM, double* __restrict__ v,
                                                      int64_t Mp, int64_t v_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= LM) return;
  double t = 0.0;
  for (int s = 0; s < S; ++s) t += (double)vpart[s * LM + i];
  v[(i / Mp) * v_stride + i % Mp] = t;
}

// ---------------- host side ----------------
template <typename K>
static int launch_wide(K kernel, size_t lds, const WParams& p, int64_t nblocks, hipStream_t s, int threads = 512) {
  // dynamic LDS above 64 KB is an opt-in per kernel function and device (a handful of instantiations: linear search)
  struct Seen { const void* fn; int dev; };
  static Seen seen[64];
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
      if (n_seen < 64) seen[n_seen++] = Seen{fn, dev};
    }
  }
  GPZ_REQUIRE(nblocks > 0 && nblocks < (1ll << 31), "wide product: bad grid");
  hipLaunchKernelGGL(kernel, dim3((unsigned)nblocks), dim3((unsigned)threads), lds, s, p);
  GPZ_LAUNCH_OK();
  return 0;
}

bool fused1_supported(int dtype, int kind, int d) {
  return dtype == GPZ_F32 && (kind == GPZ_KERNEL_RBF || kind == GPZ_KERNEL_MATERN32) && (d == 1 || d == 2);
}

int fused1_launch(const Fused1Args& a, hipStream_t s) {
  GPZ_REQUIRE(fused1_supported(GPZ_F32, a.kind, a.d), "fused stage 1: kind=%d d=%d unsupported", a.kind, a.d);
  GPZ_REQUIRE(a.Mp % 128 == 0 && a.ncp % 128 == 0 && a.Mp > 0 && a.ncp > 0 && a.L > 0, "fused stage 1: bad extents");
  GPZ_REQUIRE(a.Mp * a.Mp * 4 < (1ll << 31) && a.M * a.d * 4 < (1ll << 31), "fused stage 1: M too large");
  constexpr int TM = 512, TN = 128;           // 16 waves (1024 threads), one workgroup per CU
  WParams p = {};
  p.A = a.Linv; p.lda = a.Mp; p.sA0 = a.Mp * a.Mp;
  p.Z = a.Z; p.MD = a.M * a.d; p.M = a.M;
  p.X = a.X; p.nreal = a.nreal;
  p.sigma = a.sigma; p.ell = a.ell;
  p.C = a.Wt; p.ldc = a.ncp; p.sC0 = a.Mp * a.ncp;
  p.mu = a.muE; p.sMu = a.Mp;
  p.ps_sq = a.ps_sq; p.ps_mu = a.ps_mu; p.ncols = a.ncp;
  p.L = a.L; p.nblk = (int)(a.Mp / 128); p.mtw = (int)((a.Mp + TM - 1) / TM); p.nt = (int)(a.ncp / TN);
  // strips of W column tiles: wide enough to share a row panel in L2 (one XCD holds 64 workgroups), numerous enough that
  // every level of row tiles gives each XCD work
  int strips = (p.nt + 31) / 32;              // one XCD holds 32 of these workgroups
  const int want = (32 + p.mtw * p.L - 1) / (p.mtw * p.L);
  if (strips < want) strips = want < p.nt ? want : p.nt;
  p.W = (p.nt + strips - 1) / strips;
  p.strips = (p.nt + p.W - 1) / p.W;
  const int64_t units = (int64_t)p.mtw * p.L * p.strips;
  const int64_t nblocks = (units + 7) / 8 * 8 * p.W;
#define GPZ_W1(KIND, D) return launch_wide(gemmw_kernel<TM, TN, WB_GEN, WA_LOWER, WE_STORE_STATS, KIND, D>, \
                                           WLds<TM, TN, WB_GEN, D>::bytes, p, nblocks, s, 1024)
  if (a.kind == GPZ_KERNEL_MATERN32) { if (a.d == 2) GPZ_W1(1, 2); GPZ_W1(1, 1); }
  if (a.d == 2) GPZ_W1(0, 2);
  GPZ_W1(0, 1);
#undef GPZ_W1
}

bool wide_product_supported(int64_t Mp, int64_t ncp) {
  return Mp % 128 == 0 && ncp % 128 == 0 && Mp * Mp * 4 < (1ll << 31) && Mp * ncp * 4 < (1ll << 31);
}

template <int TM, int TN>
static int wide_product_launch_t(const WideArgs& a, hipStream_t s) {
  WParams p = {};
  p.A = a.A; p.lda = a.Mp; p.sA0 = a.Mp * a.Mp;
  p.B = a.B; p.ldb = a.ncp; p.sB0 = a.Mp * a.ncp;
  p.C = a.C; p.ldc = a.ncp; p.sC0 = a.Mp * a.ncp;
  p.mu = a.mu; p.sMu = a.Mp;
  p.ps_sq = a.ps_sq; p.ps_mu = a.ps_mu; p.ncols = a.ncp; p.M = a.Mp;
  p.colscale = a.colscale; p.colvec = a.colvec; p.sCs = a.ncp; p.rowvec = a.rowvec; p.sRv = a.Mp; p.aux = a.aux;
  p.gate = a.gate;
  p.L = a.L; p.nblk = (int)(a.Mp / 128); p.mtw = (int)((a.Mp + TM - 1) / TM); p.nt = (int)((a.ncp + TN - 1) / TN);
  // strips of (nearly) equal width, at most 4096 columns: an XCD takes every 8th (latent, strip) unit.  Measured at
  // config 3 with 128-column tiles (evaluation ms, stage 1 / stage 2 TF): 8 tiles 391.0, 141.1 / 145.7 (L2 -> fabric
  // 23.2 GB per stage-1 launch); 16 tiles 386.7, 142.8 / 147.4 (17.8 GB); 24: 386.1; 32: 384.2, 143.7 / 148.5; 48: 385.8
  constexpr int WMAX = 4096 / TN;
  int strips = (p.nt + WMAX - 1) / WMAX;
  // ... but at least 8 units where the problem has that many column tiles: with fewer units than XCDs part of the chip
  // sits idle (N=3000, M=3000, L=4 as one 24-column strip per latent = 4 units: 61 TF; two strips = 8 units: 122 TF;
  // four strips = 16 units: 108 TF, narrower strips share each A panel among fewer workgroups)
  const int want = (8 + p.L - 1) / p.L;
  if (strips < want) strips = want < p.nt ? want : p.nt;
  p.strips = strips;
  p.W = (p.nt + strips - 1) / strips;        // the widest strip (the kernel deals the column tiles out evenly)
  const int64_t units = (int64_t)p.L * p.strips;
  // Two schedules.  From 4 rounds of 512 resident workgroups up (a config-3 chunk: 24; configs[1]: 6): a workgroup takes TWO
  // row tiles (equal work everywhere, the second tile's operands prefetched) and the workgroups of a unit are dispatched
  // column tile by column tile, so the row tiles of a column walk its B panel together in L2 and no tail of long tiles is
  // left.  Same box, paired column-major against single tiles row tile by row tile, stage 1 / stage 2 TF: config-3 chunks
  // 139.7 / 149.6 -> 141.5 / 150.7 (L2 -> fabric reads 20.9 -> 17.5 GB per stage-1 launch); configs[1] 120.1 / 126.8 ->
  // 122.2 / 130.3; N=5000, M=512, L=64 114.0 / 122.8 -> 124.7 / 134.8; N=13000, M=2048, L=32 136.5 / 147.6 -> 139.7 / 149.4;
  // N_b=7000, M=3000, L=20 129.5 / 139.0 -> 130.9 / 138.7.  (Column-major WITHOUT the pairing leaves the long tiles of the
  // last columns running alone: 121 TF at config 3.  And equal strips matter more than either: with a short last strip the
  // XCDs carry unequal work, which had made the pairs look 4 % WORSE than single tiles at N_b=7000.)  Below that a launch
  // has too few pairs to fill the chip (N=3000, M=3000, L=4: 576 pairs, 122 -> 98 TF): single tiles, longest k range first.
  const int64_t paired = (units + 7) / 8 * 8 * ((p.mtw + 1) / 2) * p.W;
  p.pair = p.colmajor = paired >= 4 * 512;
  const int64_t nblocks = p.pair ? paired : (units + 7) / 8 * 8 * p.mtw * p.W;
  constexpr size_t lds = WLds<TM, TN, WB_MEM, 1>::bytes;
#define GPZ_WM(ATRI, EPI) return launch_wide(gemmw_kernel<TM, TN, WB_MEM, ATRI, EPI, 0, 1>, lds, p, nblocks, s)
  switch (a.epilogue) {
    case WIDE_STORE_STATS:
      GPZ_REQUIRE(!a.upper && a.C && a.mu && a.ps_sq && a.ps_mu, "wide product: store + statistics is the lower-triangular product and needs C, mu and both slabs");
      GPZ_WM(WA_LOWER, WE_STORE_STATS);
    case WIDE_STATS:
      GPZ_REQUIRE(a.upper && a.ps_sq, "wide product: statistics only is the upper-triangular product");
      GPZ_WM(WA_UPPER, WE_STATS);
    case WIDE_STORE:
      GPZ_REQUIRE(a.upper && a.C, "wide product: the plain store is built for the upper-triangular product");
      GPZ_WM(WA_UPPER, WE_STORE);
    case WIDE_STORE_COLSCALE:
      GPZ_REQUIRE(a.upper && a.C && a.colscale, "wide product: the column-scaled store is the upper-triangular product and needs the factors");
      GPZ_WM(WA_UPPER, WE_STORE_COLSCALE);
    case WIDE_WBAR:
      GPZ_REQUIRE(!a.upper && a.C && a.colscale && a.colvec && a.rowvec && a.aux, "wide product: the W-bar epilogue needs its operands");
      GPZ_WM(WA_LOWER, WE_WBAR);
    case WIDE_KBAR:
      GPZ_REQUIRE(a.upper == 2 && a.C && a.colscale && a.colvec && a.rowvec, "wide product: the K-bar epilogue is the dense product and needs its vectors");
      GPZ_WM(WA_DENSE, WE_KBAR);
    case WIDE_ADD_COLSCALE:
      GPZ_REQUIRE(a.upper == 1 && a.C && a.colscale, "wide product: the accumulating column-scaled store is the upper-triangular product");
      GPZ_WM(WA_UPPER, WE_ADD_COLSCALE);
    default: break;
  }
#undef GPZ_WM
  GPZ_REQUIRE(false, "wide product: unknown epilogue %d", a.epilogue);
}

int wide_product_launch(const WideArgs& a, hipStream_t s) {
  GPZ_REQUIRE(wide_product_supported(a.Mp, a.ncp) && a.L > 0, "wide product: bad extents");
  return wide_product_launch_t<128, 256>(a, s);
}

bool wide_nt_supported(int64_t Mp, int64_t K) {
  return Mp % 128 == 0 && K % 128 == 0 && K > 0 && Mp * K * 4 < (1ll << 31);
}

// Pieces of the k extent for a launch with few tiles: at Mp = 512, L = 8 (configs[1]) there are 48 tiles for 256 CUs and one
// workgroup walked all 50 048 columns (6.0 ms per launch: 60 % of a forward + backward there); cut into pieces the launch
// fills the chip.  1: no cut (enough tiles, or a short k).
static bool nt_small_tiles(int64_t nblk) { return nblk <= 4; }      // one 128-row block per tile (gemmw_nt_kernel<128>)
static int64_t nt_tiles(int64_t nblk) {
  if (nt_small_tiles(nblk)) return nblk * (nblk + 1) / 2;
  const int64_t mt = (nblk + 1) / 2;
  return mt * (mt + 1) - (nblk & 1);             // row tile ti holds min(2 ti + 2, nblk) column tiles
}

int wide_nt_pieces(int64_t Mp, int64_t K, int L) {
  const int64_t tiles = nt_tiles(Mp / 128) * L;
  if (tiles >= 384) {
    // Enough tiles to fill the chip -- but they all run equally long (same k extent), so the launch takes WHOLE rounds of
    // the 512 resident workgroups: 2304 tiles (config 3: 16 blocks, L = 32) are 4.5 rounds and run as long as 5.  Cutting k
    // in two makes them 9.0: pick the cut (<= 4 pieces, >= 128 steps each) whose last round is fullest, charging each extra
    // piece 1.5 % for its partial tiles' trip through memory.
    const double r = (double)tiles / 512.0;
    int best = 1;
    double best_cost = 1e30;
    for (int S = 1; S <= 4; ++S) {
      if (S > 1 && K / S < 2048) break;
      const double rounds = r * S, cost = std::ceil(rounds - 1e-9) / rounds * (1.0 + 0.015 * (S - 1));
      if (cost < best_cost - 1e-9) { best_cost = cost; best = S; }
    }
    return best;
  }
  // one round of resident workgroups, not a second round with a few stragglers: two 8-wave workgroups per CU, or four of
  // the 4-wave ones (one 128-block per tile)
  int64_t S = (nt_small_tiles(Mp / 128) ? 1024 : 512) / tiles;
  S = std::min<int64_t>(S, K / 1024);           // at least 64 steps per piece
  S = std::min<int64_t>(S, 16);
  return (int)std::max<int64_t>(S, 1);
}

size_t wide_nt_vpart_floats(int64_t Mp, int64_t K, int L) { return (size_t)wide_nt_pieces(Mp, K, L) * L * Mp; }

size_t wide_nt_scratch_floats(int64_t Mp, int64_t K, int L) {
  const int S = wide_nt_pieces(Mp, K, L);
  return S > 1 ? (size_t)S * L * Mp * Mp : 0;
}

int wide_nt_launch(const float* A, const float* B, float* C, int64_t Mp, int64_t K, int L, hipStream_t s, float* scratch,
                   const float* w, const int32_t* gate, const float* gm, float* vpart, double* v, int64_t v_stride) {
  GPZ_REQUIRE(A && B && C && wide_nt_supported(Mp, K) && L > 0, "wide A B^T: bad arguments");
  GPZ_REQUIRE(!gm || (w && vpart && v && A == B), "wide A B^T: the row products A gm need weights, A == B and their buffers");
  NTParams p;
  p.w = w; p.sW = K; p.gate = gate; p.gm = gm; p.vpart = vpart;
  p.A = A; p.B = B; p.C = C; p.ld = K; p.sAB = Mp * K; p.ldc = Mp; p.sC = Mp * Mp;
  p.K = (int)K; p.nblk = (int)(Mp / 128); p.mt = (p.nblk + 1) / 2; p.L = L;
  const bool small = nt_small_tiles(p.nblk);
  p.T = (int)nt_tiles(p.nblk);
  int S = scratch ? wide_nt_pieces(Mp, K, L) : 1;
  p.Ks = (int)(((K + S - 1) / S + 31) / 32 * 32);
  S = (int)((K + p.Ks - 1) / p.Ks);
  p.S = S; p.part = scratch; p.sPart = (int64_t)L * Mp * Mp;
  const int64_t nblocks = (int64_t)p.T * L * S;
  GPZ_REQUIRE(nblocks > 0 && nblocks < (1ll << 31), "wide A B^T: bad grid");
  if (small) {
    constexpr size_t lds = sizeof(float) * 2 * (128 + 128) * W_BK;
    if (w) hipLaunchKernelGGL((gemmw_nt_kernel<128, true>), dim3((unsigned)nblocks), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((gemmw_nt_kernel<128, false>), dim3((unsigned)nblocks), dim3(256), lds, s, p);
  } else {
    constexpr size_t lds = sizeof(float) * 2 * (256 + 128) * W_BK;
    if (w) hipLaunchKernelGGL((gemmw_nt_kernel<256, true>), dim3((unsigned)nblocks), dim3(512), lds, s, p);
    else hipLaunchKernelGGL((gemmw_nt_kernel<256, false>), dim3((unsigned)nblocks), dim3(512), lds, s, p);
  }
  GPZ_LAUNCH_OK();
  if (S > 1) {
    const int64_t total4 = (int64_t)L * Mp * Mp / 4;
    hipLaunchKernelGGL(nt_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, C, scratch, S, p.sPart, (int)Mp, total4, gate, (int)(w != nullptr && A == B));
    GPZ_LAUNCH_OK();
  }
  if (gm) {
    const int64_t LM = (int64_t)L * Mp;
    hipLaunchKernelGGL(nt_vsum_kernel, dim3((unsigned)((LM + 255) / 256)), dim3(256), 0, s, vpart, S, LM, v, Mp, v_stride);
    GPZ_LAUNCH_OK();
  }
  return 0;
}

}  // namespace gpz
