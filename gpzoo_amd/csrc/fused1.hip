// fused1.hip -- Wt = Linv * k(Z, X) with the covariance operand generated in registers.
//
// Stage 1 of gp.py:255 + :276 (Kzx = kernel(Z, X); Wt = solve_triangular(L, Kzx)) for the fp32 RBF / Matern-3/2
// kernels on 1-D / 2-D inputs.  The reference -- and this library's own two-kernel path (kfill.hip + gemm.hip) --
// materialises Kzx: 52 GB written and read back per evaluation at N=200k, M=2048, L=32.  Here the B operand of
// every v_mfma_f32_16x16x4_f32 is one covariance value per lane, k(z_k, x_n) for the lane's (k, n) slot, computed
// from the lane's own column coordinates (registers, fixed for the whole tile) and four inducing points per
// 16-deep k-chunk (a 256-byte Z tile staged next to the A tile, read as LDS broadcasts).  Kzx never exists: no
// fill kernel, no B tile in LDS, no B fragment reads; the ~10 VALU / 2 transcendental instructions per value issue
// in the shadow of the 8 MFMAs that consume it (one value feeds the eight 16-row sub-tiles of a 128-row wave tile).
//
// Workgroup: 256 x 128 output tile, 8 waves as 2 (rows) x 4 (columns), wave tile 128 x 32 = 16 accumulator tiles.
// A (Linv, lower triangular) is staged [row][k] 32 deep through registers into an XOR-swizzled LDS image
// (conflict-free ds_read_b128 fragments), double buffered, one barrier per staged tile.  <= 128 VGPRs and 65 KB
// of LDS: two workgroups per CU, four waves per SIMD.
//
// Values, k order and MFMA order are those of the two-kernel path (cov.h is shared with kfill.hip; lane group q owns
// k = 4q .. 4q+3 of a chunk in both), so Wt is bitwise what kfill + gemm128_kernel produce.
#include "fused1.h"

#include "cov.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

// Timing-only diagnostics (WRONG results by construction; tools/ablate_fused.sh builds them next to the real library):
// -DGPZ_F1_ABL=<bits>  1: no covariance arithmetic (the B operand is a coordinate), 2: no tile loads after the first,
// 4: no epilogue (statistics, Wt store), 8: no per-tile barrier.  -DGPZ_F1_WPE=<n>: waves per SIMD the kernel is compiled for.
#ifndef GPZ_F1_ABL
#define GPZ_F1_ABL 0
#endif
#ifndef GPZ_F1_WPE
#define GPZ_F1_WPE 4
#endif
#ifndef GPZ_F1_SCHED       // 1: pin the MFMA / covariance-arithmetic interleave of a full chunk (default), 0: hipcc's own order
#define GPZ_F1_SCHED 1
#endif

namespace gpz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Fused1Params {
  const float* A; int64_t lda, sA0;
  const float* Z; int64_t MD;               // M * D
  const float* X; int64_t nreal;
  const float* sigma; const float* ell;
  float* C; int64_t ldc, sC0;
  const float* mu; int64_t sMu;
  float* ps_sq; float* ps_mu; int64_t ncols;
  int64_t M;
  int L, nblk, mt2, nt, W, strips;
};

constexpr int F1_BK = 32;                    // staged k depth: two 16-deep chunks
constexpr int F1_TM = 256;                   // rows per workgroup tile
constexpr int F1_A_ELEMS = F1_TM * F1_BK;    // 32 KB per buffer
constexpr int F1_ZB = 256;                   // floats per Z block buffer (128 inducing points x 2 coordinates)
constexpr size_t F1_LDS = sizeof(float) * 2 * (F1_A_ELEMS + F1_ZB);

template <int KIND, int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(GPZ_F1_WPE, GPZ_F1_WPE))) void fused_stage1_kernel(const Fused1Params p) {
  constexpr int BK = F1_BK;
  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  float* const smem = reinterpret_cast<float*>(smem_raw);
  auto sA = [&](int buf) -> float* { return smem + buf * F1_A_ELEMS; };
  float* const sZ = smem + 2 * F1_A_ELEMS;   // [2][F1_ZB]: inducing points of the 128-blocks b (parity b & 1)

  // ---------------- tile decode ----------------
  // Blocks b, b + 8, ... run on one XCD (round-robin dispatch).  A unit = (row tile, latent, strip of W column tiles): its
  // W workgroups stream the same Linv row panel (<= 2 MB) through that XCD's L2 together.  Units are ordered longest
  // k-range first (row tiles descending), every level's units spread evenly over the XCDs.
  int ti, tj, b0;
  {
    const int bid = blockIdx.x;
    const int x = bid & 7, s = bid >> 3;
    const int ug = s / p.W, within = s - ug * p.W;
    const int u = ug * 8 + x;
    const int per_level = p.L * p.strips;
    if (u >= p.mt2 * per_level) return;
    const int level = u / per_level, rem = u - level * per_level;
    ti = p.mt2 - 1 - level;
    b0 = rem / p.strips;
    tj = (rem - b0 * p.strips) * p.W + within;
    if (tj >= p.nt) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // which 128-row half a wave takes alternates from tile to tile: the upper half skips its last four staged tiles
  const int wm = (wave >> 2) ^ ((ti ^ tj) & 1), wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  const int Mp = p.nblk * 128;
  const int k_end = min(F1_TM * (ti + 1), Mp);
  const int nk = k_end / BK;
  const int db = 2 * ti + wm;                 // this wave's 128-row block: Linv[db][k-block] = 0 for k-block > db
  const bool active = db < p.nblk;            // an odd block count leaves the last tile's upper half empty
  const int n_full = active ? 4 * db : 0;     // staged tiles left of the diagonal block

  // ---------------- this lane's columns ----------------
  const CovConst cc = cov_const<KIND>(p.sigma[b0], p.ell[b0]);
  float xc[2][D], ampn[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = (int64_t)tj * 128 + wn * 32 + ni * 16 + r;
    const bool real = n < p.nreal;
#pragma unroll
    for (int k = 0; k < D; ++k) xc[ni][k] = real ? p.X[n * D + k] : 0.f;
    ampn[ni] = real ? cc.amp : 0.f;           // padded columns: exactly zero, as the stand-alone fill writes them
  }

  // ---------------- staging (LDS-DMA: buffer_load ... lds, no staging registers, no ds_write) ----------------
  // A tile image: [256 rows][32 k] floats, unpadded; 16-byte chunk c of row w sits at slot c ^ (w & 7), which makes the
  // ds_read_b128 fragment reads conflict-free.  Wave w fills the 1-KB pieces 4w .. 4w+3 (8 rows each); lane i of a
  // buffer_load_dwordx4 ... lds lands at piece base + 16 i, so the swizzle sits on the lane's SOURCE address.  The buffer
  // form splits that address into a descriptor (this latent's Linv), one constant per-lane byte offset (VGPR) and a
  // wave-uniform byte offset (SGPR) advanced by scalar adds: no vector address arithmetic in the loop.
  typedef __attribute__((address_space(3))) void lds_void;
#if defined(__HIP_DEVICE_COMPILE__)   // gfx950 builtins: the host pass of this single-source file only needs the kernel's stub
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A + b0 * p.sA0), 0, (int)(p.lda * Mp * sizeof(float)), 0x00020000);
  // reads past the end of Z return zero: padded inducing rows may hold any finite point, their Wt rows are masked below
  const __amdgpu_buffer_rsrc_t z_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.Z), 0, (int)(p.MD * sizeof(float)), 0x00020000);
#endif
  int a_soff[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    int row = ti * F1_TM + (4 * wave + h) * 8;
    if (row >= Mp) row -= 128;                // empty upper half: re-read the lower half's rows (never used)
    a_soff[h] = row * (int)p.lda * (int)sizeof(float);
  }
  const int a_voff = ((lane >> 3) * (int)p.lda + ((lane & 7) ^ (lane >> 3)) * 4) * (int)sizeof(float);
  bool abl_first = true;
  auto stage_load = [&](int buf) __attribute__((always_inline)) {
    if ((GPZ_F1_ABL & 2) && !abl_first) return;
    abl_first = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(sA(buf) + (4 * wave + h) * 256), 16, a_voff, a_soff[h], 0, 0);
      a_soff[h] += BK * (int)sizeof(float);
    }
#endif
  };
  // Inducing points: one 128-point block (128 D floats) per LDS buffer, fetched a whole block ahead by waves 0 .. 2D-1
  // (one coordinate per lane).
  auto z_load = [&](int blk) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (wave < 2 * D)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rsrc, (lds_void*)(sZ + (blk & 1) * F1_ZB + wave * 64), 4, lane * 4,
                                               (blk * 128 * D + wave * 64) * 4, 0, 0);
#endif
  };

  f32x4 acc[8][2];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};

  // fragment addresses (floats): lane (r, q) owns k = 4q .. 4q+3 of each 16-deep chunk
  const int fr_a = (wm * 128 + r) * BK + ((q ^ (r & 7)) * 4);
  const int fr_z = 4 * q * D;

  // The eight covariance values (4 k-slots x 2 column sub-tiles) of 16-deep chunk s of the k range, from the Z block in LDS.
  auto cov_chunk = [&](int s, float (&bv)[4][2]) __attribute__((always_inline)) {
    const float* zp = sZ + ((s >> 3) & 1) * F1_ZB + (s & 7) * 16 * D + fr_z;
    float zz[4][D];
    if constexpr (D == 2) {
      const f32x4 z01 = *reinterpret_cast<const f32x4*>(zp);
      const f32x4 z23 = *reinterpret_cast<const f32x4*>(zp + 4);
      zz[0][0] = z01[0]; zz[0][1] = z01[1]; zz[1][0] = z01[2]; zz[1][1] = z01[3];
      zz[2][0] = z23[0]; zz[2][1] = z23[1]; zz[3][0] = z23[2]; zz[3][1] = z23[3];
    } else {
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(zp);
#pragma unroll
      for (int j = 0; j < 4; ++j) zz[j][0] = z0[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        bv[j][ni] = (GPZ_F1_ABL & 1) ? zz[j][0] + xc[ni][0]
                                     : cov_value<KIND>(cov_radial<KIND>(cov_d2<D>(zz[j], xc[ni])), ampn[ni], cc.c0, cc.c1);
  };

  // One chunk of MFMAs over the 16-row sub-tiles mi >= LO (compile-time: inside the diagonal block the zero boundary moves
  // one sub-tile per chunk); the A fragments come in two halves of four sub-tiles (16 registers live instead of 32).
  auto mma_chunk = [&](int buf, int kc, auto lo_c, const float (&bv)[4][2]) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_c)::value;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 fa[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (half * 4 + m >= LO) fa[m] = *reinterpret_cast<const f32x4*>(sA(buf) + (fr_a ^ (kc * 16)) + (half * 4 + m) * 16 * BK);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            if (half * 4 + m >= LO)
              acc[half * 4 + m][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][j], bv[j][ni], acc[half * 4 + m][ni], 0, 0, 0);
    }
  };
  // The instruction order of a full chunk: its 64 MFMAs in 16 groups of four, each followed by its share of the NEXT
  // chunk's covariance arithmetic (8 values = 64 plain + 16 transcendental instructions) -- an MFMA occupies the SIMD's
  // vector issue for 8 of its 32 cycles, so five short instructions per four MFMAs ride in cycles the matrix pipe leaves
  // free.  Left to itself hipcc puts the arithmetic in one block at the head of the chunk; since all waves of a CU run
  // this loop in lock-step between barriers, those blocks then coincide on every wave and the matrix pipes idle.
  auto interleave = [&]() __attribute__((always_inline)) {
#if GPZ_F1_SCHED
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
#if GPZ_F1_SCHED == 2     // the tile's DMA instructions spread over the chunk instead of issued back to back at its head
      if (g % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
    }
#endif
  };

  // ---------------- k loop ----------------
  // Iteration t: start the DMA of tile t + 1 into the other buffer, run tile t, wait for the DMA, barrier.  Every region
  // below starts at an even t and has an even length, so the buffer is a compile-time constant per iteration.  No MFMA
  // sits under a run-time branch (hipcc would shuffle the accumulators through copies): a wave's k loop is n_full full
  // tiles, four diagonal-block tiles with compile-time sub-tile ranges, then (upper half) idle tiles.
  // The covariance values run one chunk ahead of the MFMAs that consume them: chunk s + 1's are computed among chunk s's
  // MFMAs (bvA / bvB alternate; bvA crosses the barrier in registers).
  using std::integral_constant;
  stage_load(0);
  z_load(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float bvA[4][2], bvB[4][2];
  if (active) cov_chunk(0, bvA);
  int t = 0;
  auto iteration = [&](auto par_c, auto run_c, auto lo0_c, auto lo1_c, auto more_c) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;
    constexpr bool RUN = decltype(run_c)::value, MORE = decltype(more_c)::value;   // MORE: this wave runs tile t + 1 too
    if (RUN && MORE) {
      stage_load(P ^ 1);                      // a wave that runs tile t + 1 is not at the end of the k range
    } else if (t + 1 < nk) {
      stage_load(P ^ 1);
    }
    if ((t & 3) == 0 && 4 * ((t >> 2) + 1) < nk) z_load((t >> 2) + 1);      // first tile of a 128-block: fetch the next block
    if constexpr (RUN) {
      cov_chunk(2 * t + 1, bvB);
      mma_chunk(P, 0, lo0_c, bvA);
      if constexpr (decltype(lo0_c)::value == 0) interleave();
      if constexpr (MORE) cov_chunk(2 * t + 2, bvA);
      mma_chunk(P, 1, lo1_c, bvB);
      if constexpr (MORE && decltype(lo1_c)::value == 0) interleave();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(GPZ_F1_ABL & 8)) __syncthreads();
    ++t;
  };
  using no_run = integral_constant<bool, false>;
  using run = integral_constant<bool, true>;
  using yes = integral_constant<bool, true>;
  using no = integral_constant<bool, false>;
  using i0 = integral_constant<int, 0>;
  using i1 = integral_constant<int, 1>;
  while (t < n_full) { iteration(i0{}, run{}, i0{}, i0{}, yes{}); iteration(i1{}, run{}, i0{}, i0{}, yes{}); }
  if (active) {
    // diagonal block: tile u, chunk c covers k = 32u + 16c ..: rows below sub-tile 2u + c are zero there
    iteration(i0{}, run{}, integral_constant<int, 0>{}, integral_constant<int, 1>{}, yes{});
    iteration(i1{}, run{}, integral_constant<int, 2>{}, integral_constant<int, 3>{}, yes{});
    iteration(i0{}, run{}, integral_constant<int, 4>{}, integral_constant<int, 5>{}, yes{});
    iteration(i1{}, run{}, integral_constant<int, 6>{}, integral_constant<int, 7>{}, no{});
  }
  while (t < nk) { iteration(i0{}, no_run{}, i0{}, i0{}, no{}); iteration(i1{}, no_run{}, i0{}, i0{}, no{}); }

  // ---------------- epilogue ----------------
  if (!active) return;
  if (GPZ_F1_ABL & 4) {        // keep the accumulators alive, store (practically) nothing
    float sum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) sum += acc[mi][ni][g];
    if (sum == 12345.678f) p.ps_sq[0] = sum;
    return;
  }
  const int64_t row0 = (int64_t)db * 128;
  const int64_t ccol0 = (int64_t)tj * 128 + wn * 32;
  // rows >= M are padding: Linv is the identity there, so they picked up k(0, x); the stand-alone path has zeros
  if (row0 + 128 > p.M) {
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
  {
    const float* mu = p.mu + b0 * p.sMu + row0 + 4 * q;
    float ssq[2] = {0.f, 0.f}, smu[2] = {0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(mu + mi * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float v = acc[mi][ni][g];
          ssq[ni] = __builtin_fmaf(v, v, ssq[ni]);
          smu[ni] = __builtin_fmaf(m4[g], v, smu[ni]);
        }
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      ssq[ni] += __shfl_xor(ssq[ni], 16); ssq[ni] += __shfl_xor(ssq[ni], 32);
      smu[ni] += __shfl_xor(smu[ni], 16); smu[ni] += __shfl_xor(smu[ni], 32);
      if (q == 0) {
        const int64_t o = ((int64_t)b0 * p.nblk + db) * p.ncols + ccol0 + ni * 16 + r;
        p.ps_sq[o] = ssq[ni];
        p.ps_mu[o] = smu[ni];
      }
    }
  }
  // Wt tile: through a wave-private LDS strip (it aliases the tile buffers: every read of them is behind the loop's last
  // barrier) so each store instruction writes eight whole 128-byte row segments
  {
    constexpr int LDE = 36;
    float* strip = smem + wave * (32 * LDE);
    float* Cg = p.C + b0 * p.sC0 + row0 * p.ldc + ccol0;
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
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + srow;
        const f32x4 v = *reinterpret_cast<const f32x4*>(strip + row * LDE + c4);
        *reinterpret_cast<f32x4*>(Cg + (int64_t)(pass * 32 + row) * p.ldc + c4) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

bool fused1_supported(int dtype, int kind, int d) {
  return dtype == GPZ_F32 && (kind == GPZ_KERNEL_RBF || kind == GPZ_KERNEL_MATERN32) && (d == 1 || d == 2);
}

template <int KIND, int D>
static int launch_one(const Fused1Params& p, int64_t nblocks, hipStream_t s) {
  // dynamic LDS above 64 KB is an opt-in per kernel function and device
  static std::mutex mu;
  static bool seen[64] = {};
  int dev = 0;
  GPZ_HIP_OK(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 64 && !seen[dev]) {
      GPZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_stage1_kernel<KIND, D>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)F1_LDS));
      seen[dev] = true;
    }
  }
  hipLaunchKernelGGL((fused_stage1_kernel<KIND, D>), dim3((unsigned)nblocks), dim3(512), F1_LDS, s, p);
  GPZ_LAUNCH_OK();
  return 0;
}

int fused1_launch(const Fused1Args& a, hipStream_t s) {
  GPZ_REQUIRE(fused1_supported(GPZ_F32, a.kind, a.d), "fused stage 1: kind=%d d=%d unsupported", a.kind, a.d);
  GPZ_REQUIRE(a.Mp % 128 == 0 && a.ncp % 128 == 0 && a.Mp > 0 && a.ncp > 0 && a.L > 0, "fused stage 1: bad extents");
  GPZ_REQUIRE(a.Mp * a.Mp * 4 < (1ll << 31) && a.M * a.d * 4 < (1ll << 31), "fused stage 1: M too large");
  Fused1Params p;
  p.A = a.Linv; p.lda = a.Mp; p.sA0 = a.Mp * a.Mp;
  p.Z = a.Z; p.MD = a.M * a.d; p.M = a.M;
  p.X = a.X; p.nreal = a.nreal;
  p.sigma = a.sigma; p.ell = a.ell;
  p.C = a.Wt; p.ldc = a.ncp; p.sC0 = a.Mp * a.ncp;
  p.mu = a.muE; p.sMu = a.Mp;
  p.ps_sq = a.ps_sq; p.ps_mu = a.ps_mu; p.ncols = a.ncp;
  p.L = a.L; p.nblk = (int)(a.Mp / 128); p.mt2 = (p.nblk + 1) / 2; p.nt = (int)(a.ncp / 128);
  // strips of W column tiles: wide enough to share a row panel in L2 (one XCD holds 64 workgroups), numerous enough that
  // every level of row tiles gives each XCD work
  int strips = (p.nt + 63) / 64;
  const int want = (32 + p.mt2 * p.L - 1) / (p.mt2 * p.L);
  if (strips < want) strips = want < p.nt ? want : p.nt;
  p.W = (p.nt + strips - 1) / strips;
  if (const char* e = getenv("GPZ_F1_W")) { const int w = atoi(e); if (w >= 1) p.W = w < p.nt ? w : p.nt; }   // diagnostics
  p.strips = (p.nt + p.W - 1) / p.W;
  const int64_t units = (int64_t)p.mt2 * p.L * p.strips;
  const int64_t nblocks = (units + 7) / 8 * 8 * p.W;
  GPZ_REQUIRE(nblocks < (1ll << 31), "fused stage 1: grid too large");
#define GPZ_F1(KIND, D) return launch_one<KIND, D>(p, nblocks, s)
  if (a.kind == GPZ_KERNEL_MATERN32) { if (a.d == 2) GPZ_F1(1, 2); GPZ_F1(1, 1); }
  if (a.d == 2) GPZ_F1(0, 2);
  GPZ_F1(0, 1);
#undef GPZ_F1
}

}  // namespace gpz
