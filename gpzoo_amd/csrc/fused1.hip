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
constexpr int F1_ZT = 64;                    // floats reserved per Z tile (32 points x 2 coordinates)
constexpr size_t F1_LDS = sizeof(float) * 2 * (F1_A_ELEMS + F1_ZT);

template <int KIND, int D, int STG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void fused_stage1_kernel(const Fused1Params p) {
  constexpr int BK = F1_BK;
  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  float* const smem = reinterpret_cast<float*>(smem_raw);
  auto sA = [&](int buf) -> float* { return smem + buf * F1_A_ELEMS; };
  auto sZ = [&](int buf) -> float* { return smem + 2 * F1_A_ELEMS + buf * F1_ZT; };

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

  // ---------------- staging ----------------
  // A tile image: [256 rows][32 k] floats, unpadded; 16-byte chunk c of row w sits at slot c ^ (w & 7), which makes the
  // ds_read_b128 fragment reads conflict-free.
  // STG 0: through registers -- thread -> (row tid / 8 + 64 h, chunk tid % 8).
  // STG 1: LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) -- wave w fills the 1-KB pieces 4w .. 4w+3
  //   (8 rows each), lane i lands at piece base + 16 i, so the swizzle sits on the lane's SOURCE address.
  // Wave-uniform row bases advance by scalar adds; the lane's share of the address is one constant 32-bit offset.
  const int ra = STG ? (lane >> 3) : (tid >> 3), ch = STG ? ((lane & 7) ^ (lane >> 3)) : (tid & 7);
  const float* a_base[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    int row = STG ? ti * F1_TM + (4 * wave + h) * 8 : ti * F1_TM + 64 * h;
    if (row >= Mp) row -= 128;                // empty upper half: re-read the lower half's rows (never used)
    a_base[h] = p.A + b0 * p.sA0 + (int64_t)row * p.lda;
  }
  const uint32_t a_off = (uint32_t)(ra * (int)p.lda + ch * 4);
  const int st_off = ra * BK + ((ch ^ (ra & 7)) * 4);
  // the Z tile (32 points): wave 0, one coordinate per lane; indices past the end re-read the last coordinate -- padded
  // inducing rows may hold any finite point, their Wt rows are masked below
  int zk = 0;
  const int zlast = (int)p.MD - 1;
  f32x4 ga[STG ? 1 : 4];
  float gz = 0.f;
  auto stage_load = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if constexpr (STG != 0) {
#if defined(__HIP_DEVICE_COMPILE__)   // a gfx950 builtin: the host pass of this single-source file only needs the kernel's stub
        typedef __attribute__((address_space(3))) void lds_void;
        __builtin_amdgcn_global_load_lds(a_base[h] + a_off, (lds_void*)(sA(buf) + (4 * wave + h) * 256), 16, 0, 0);
#endif
      } else {
        ga[h] = *reinterpret_cast<const f32x4*>(a_base[h] + a_off);
      }
      a_base[h] += BK;
    }
    if (wave == 0) {
      if (D == 2 || lane < 32) {
        const int zi = min(zk + lane, zlast);
        if constexpr (STG != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
          typedef __attribute__((address_space(3))) void lds_void;
          __builtin_amdgcn_global_load_lds(p.Z + zi, (lds_void*)sZ(buf), 4, 0, 0);
#endif
        } else {
          gz = p.Z[zi];
        }
      }
    }
    zk += BK * D;
  };
  auto stage_commit = [&](int buf) __attribute__((always_inline)) {
    if constexpr (STG != 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h) *reinterpret_cast<f32x4*>(sA(buf) + st_off + h * 64 * BK) = ga[h];
      if (wave == 0 && (D == 2 || lane < 32)) sZ(buf)[lane] = gz;
    }
  };

  f32x4 acc[8][2];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0, 0, 0, 0};

  // fragment addresses (floats): lane (r, q) owns k = 4q .. 4q+3 of each 16-deep chunk
  const int fr_a = (wm * 128 + r) * BK + ((q ^ (r & 7)) * 4);
  const int fr_z = 4 * q * D;

  // One staged tile: per chunk the A fragments of the sub-tiles mi >= LO (compile-time: inside the diagonal block the
  // zero boundary moves one 16-row sub-tile per chunk), four inducing points, then per k-slot two covariance values
  // and their MFMAs.
  auto compute = [&](int buf, auto lo0_c, auto lo1_c) __attribute__((always_inline)) {
    constexpr int LO[2] = {decltype(lo0_c)::value, decltype(lo1_c)::value};
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      float zz[4][D];
      if constexpr (D == 2) {
        const f32x4 z01 = *reinterpret_cast<const f32x4*>(sZ(buf) + fr_z + kc * 16 * D);
        const f32x4 z23 = *reinterpret_cast<const f32x4*>(sZ(buf) + fr_z + kc * 16 * D + 4);
        zz[0][0] = z01[0]; zz[0][1] = z01[1]; zz[1][0] = z01[2]; zz[1][1] = z01[3];
        zz[2][0] = z23[0]; zz[2][1] = z23[1]; zz[3][0] = z23[2]; zz[3][1] = z23[3];
      } else {
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(sZ(buf) + fr_z + kc * 16 * D);
#pragma unroll
        for (int j = 0; j < 4; ++j) zz[j][0] = z0[j];
      }
      float bv[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          bv[j][ni] = cov_value<KIND>(cov_radial<KIND>(cov_d2<D>(zz[j], xc[ni])), ampn[ni], cc.c0, cc.c1);
      // the A fragments in two halves of four sub-tiles (16 registers live instead of 32)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        constexpr int dummy = 0; (void)dummy;
        f32x4 fa[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (half * 4 + m >= LO[kc]) fa[m] = *reinterpret_cast<const f32x4*>(sA(buf) + (fr_a ^ (kc * 16)) + (half * 4 + m) * 16 * BK);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              if (half * 4 + m >= LO[kc])
                acc[half * 4 + m][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][j], bv[j][ni], acc[half * 4 + m][ni], 0, 0, 0);
      }
    }
  };

  // ---------------- k loop ----------------
  // Iteration t: fetch tile t + 1 into registers, run tile t, store tile t + 1 to the other LDS buffer, barrier.  Every
  // region below starts at an even t and has an even length, so the buffer is a compile-time constant per iteration.
  // No MFMA sits under a run-time branch (hipcc would shuffle the accumulators through copies): a wave's k loop is
  // n_full full tiles, four diagonal-block tiles with compile-time sub-tile ranges, then (upper half) idle tiles.
  using std::integral_constant;
  stage_load(0);
  stage_commit(0);
  __syncthreads();
  int t = 0;
  auto iteration = [&](auto par_c, auto run_c, auto lo0_c, auto lo1_c) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;
    if (t + 1 < nk) stage_load(P ^ 1);
    if constexpr (decltype(run_c)::value) compute(P, lo0_c, lo1_c);
    if (t + 1 < nk) stage_commit(P ^ 1);
    __syncthreads();
    ++t;
  };
  using no_run = integral_constant<bool, false>;
  using run = integral_constant<bool, true>;
  using i0 = integral_constant<int, 0>;
  using i1 = integral_constant<int, 1>;
  while (t < n_full) { iteration(i0{}, run{}, i0{}, i0{}); iteration(i1{}, run{}, i0{}, i0{}); }
  if (active) {
    // diagonal block: tile u, chunk c covers k = 32u + 16c ..: rows below sub-tile 2u + c are zero there
    iteration(i0{}, run{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    iteration(i1{}, run{}, integral_constant<int, 2>{}, integral_constant<int, 3>{});
    iteration(i0{}, run{}, integral_constant<int, 4>{}, integral_constant<int, 5>{});
    iteration(i1{}, run{}, integral_constant<int, 6>{}, integral_constant<int, 7>{});
  }
  while (t < nk) { iteration(i0{}, no_run{}, i0{}, i0{}); iteration(i1{}, no_run{}, i0{}, i0{}); }

  // ---------------- epilogue ----------------
  if (!active) return;
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

template <int KIND, int D, int STG>
static int launch_one(const Fused1Params& p, int64_t nblocks, hipStream_t s) {
  // dynamic LDS above 64 KB is an opt-in per kernel function and device
  static std::mutex mu;
  static bool seen[64] = {};
  int dev = 0;
  GPZ_HIP_OK(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 64 && !seen[dev]) {
      GPZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_stage1_kernel<KIND, D, STG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)F1_LDS));
      seen[dev] = true;
    }
  }
  hipLaunchKernelGGL((fused_stage1_kernel<KIND, D, STG>), dim3((unsigned)nblocks), dim3(512), F1_LDS, s, p);
  GPZ_LAUNCH_OK();
  return 0;
}

int fused1_launch(const Fused1Args& a, hipStream_t s) {
  GPZ_REQUIRE(fused1_supported(GPZ_F32, a.kind, a.d), "fused stage 1: kind=%d d=%d unsupported", a.kind, a.d);
  GPZ_REQUIRE(a.Mp % 128 == 0 && a.ncp % 128 == 0 && a.Mp > 0 && a.ncp > 0 && a.L > 0, "fused stage 1: bad extents");
  GPZ_REQUIRE(a.Mp * a.Mp < (1ll << 31), "fused stage 1: M too large");
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
  p.strips = (p.nt + p.W - 1) / p.W;
  const int64_t units = (int64_t)p.mt2 * p.L * p.strips;
  const int64_t nblocks = (units + 7) / 8 * 8 * p.W;
  GPZ_REQUIRE(nblocks < (1ll << 31), "fused stage 1: grid too large");
  static const int stg = [] { const char* e = getenv("GPZ_F1_STG"); return e ? atoi(e) : 1; }();
#define GPZ_F1(KIND, D) return stg ? launch_one<KIND, D, 1>(p, nblocks, s) : launch_one<KIND, D, 0>(p, nblocks, s)
  if (a.kind == GPZ_KERNEL_MATERN32) { if (a.d == 2) GPZ_F1(1, 2); GPZ_F1(1, 1); }
  if (a.d == 2) GPZ_F1(0, 2);
  GPZ_F1(0, 1);
#undef GPZ_F1
}

}  // namespace gpz
