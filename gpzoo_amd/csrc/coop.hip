// coop.hip -- batched Cholesky factorisation AND triangular inverse in ONE launch: a left-looking tile dataflow.
//
// Replaces, for the fused forward / backward passes and the stand-alone entries, the launch-per-step chain of
// csrc/factor.hip (diagonal block -> panel -> column update -> diagonal block -> panel -> SYRK, then four levels of two
// GEMM launches for the inverse): there every step of the chain is its own latency-bound launch and nothing overlaps.
// Reference call sites: torch.linalg.cholesky at gp.py:213, 270, 360; the solves it feeds at gp.py:218, 276, 365.
//
// Work units are 128 x 128 fp64 tiles, each written exactly ONCE by the workgroup that owns it and read by others only
// behind its flag (so no workgroup ever holds a stale copy of a line, whatever XCD it runs on):
//   C(i,j), i > j :  L[i][j] = (A[i][j] - sum_{k<j} L[i][k] L[j][k]^T) inv(L[j][j])^T
//   C(j,j)        :  L[j][j] = chol(A[j][j] - sum_{k<j} L[j][k] L[j][k]^T), plus its inverse (csrc/diag128.h, in LDS)
//   X(i,j), i > j :  Linv[i][j] = -inv(L[i][i]) sum_{k=j}^{i-1} L[i][k] Linv[k][j]          (forward substitution)
// The sums run on v_mfma_f64_16x16x4_f64 with the accumulators of a tile resident in registers for its whole k range
// (no read-modify-write of a trailing matrix: every tile is read as an operand and written once); both operands of
// every product are [row][k] images (Linv's transpose is kept beside it for that), staged by LDS-DMA, double buffered.
//
// Scheduling: the workgroups of a matrix's cluster claim tiles from ONE ordered list with an atomic ticket.  The list
// is a linear extension of the tile DAG, so the earliest unfinished tile is always held by a running workgroup and has
// all its inputs: the launch makes progress with ANY number of resident workgroups (no co-residency assumption, no
// grid barrier).  A claimed tile accumulates the k-blocks whose inputs are there and polls for the rest, which is what
// overlaps the diagonal-block chain with the bulk of the updates.  The order itself comes from a host-side model of
// exactly this execution (coop_order: greedy by remaining critical path among the tiles that would not block;
// tools/coop_sched_sim.py is the same model in Python) and travels as a kernel argument.
//
// Hand-offs follow the agent-scope protocol of the CDNA4 guide: producer plain stores -> every wave s_waitcnt vmcnt(0)
// -> barrier -> one lane: release fence, s_waitcnt, relaxed agent flag store; consumer one wave polls relaxed (sc1)
// loads -> one lane acquire fence, s_waitcnt -> barrier -> plain loads.  A poll that sees nothing for five seconds
// raises the launch's abort word, which every poll loop reads: all workgroups then leave and info reports -7.
#include "common.h"
#include "diag128.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace gpz {

constexpr int CO_THREADS = 512;
constexpr int CO_MAX_TASKS = 1536;                 // 3 KB of kernel arguments: orders up to nblk = 38 with the inverse
constexpr int CO_SYNC_HEAD = 32;                   // words in front of a matrix's flags (ticket on its own line)
constexpr size_t CO_TILE_DOUBLES = 4 * 2 * 128 * 16;                       // four stages of an A and a B image
constexpr size_t CO_SH_OFF = sizeof(double) * diag::LDS_DOUBLES;           // shared scalars behind the larger role
constexpr size_t CO_LDS_BYTES = CO_SH_OFF + 64;
static_assert(sizeof(double) * CO_TILE_DOUBLES <= CO_SH_OFF, "tile buffers alias the diagonal block's storage");
static_assert(sizeof(double) * 128 * 130 <= CO_SH_OFF, "the tile image aliases them too");

struct CoopOrder { uint16_t t[CO_MAX_TASKS]; };    // kind << 12 | i << 6 | j; kind 0: C(i,j), 1: X(i,j), 2: C(j,j-1) + C(j,j), 3: the early part of C(j,j)

struct CoopParams {
  double* A; int64_t lda, sA;          // (batch) padded matrices, factored in place
  double* Dinv; int64_t sD;            // (batch, nblk, 128, 128) inverses of the diagonal blocks
  double* Linv; int64_t sL;            // (batch, Mp, Mp) inverse of the factor (pitch Mp), or null: factor only
  double* XT; int64_t sX;              // (batch, Mp, Mp) scratch: transpose of Linv
  float* Linv32;                       // (batch, Mp, Mp) fp32 copy of Linv written beside it (stride sL), or null
  int32_t* info; int64_t m_real;
  uint32_t* sync; int64_t sS;          // per matrix: [0] ticket, then flags of C tiles, then flags of X tiles (X(j,j):
                                       // the inverse of diagonal block j is stored, ahead of C(j,j): everything of it is)
  uint32_t* abort_word;
  int nblk, batch, ntasks, gmax;
  unsigned long long timeout_ticks;    // of the 100 MHz s_memrealtime clock
  unsigned long long* trace;           // diagnostics (gpz_debug_coop_trace): 8 words per (matrix, ticket), or null
  const uint32_t* debug_mute;          // diagnostics (gpz_debug_coop_mute): a flag that is never stored, or null
};

namespace {
typedef double d2v __attribute__((ext_vector_type(2)));
using diag::d4;
using diag::DP;

__device__ __forceinline__ unsigned long long realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) uint32_t guint;

__device__ __forceinline__ uint32_t flag_load(const guint* f) {
  return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Ctx {
  int tid, lane, wave, wm, wn, r, q;
  int fr_a, fr_b;                  // fragment offsets (doubles) inside an A / B image
  const CoopParams* p;
};

// Dynamic LDS, named in every function that touches it (a pointer carried through Ctx would reach the noinline task
// functions as a generic pointer and every fragment read would become a flat_load):
//   tiles: [stage][A | B][128 rows][16 k] doubles, chunk c of row w at slot c ^ (w & 7); aliased by the diagonal block's S
//   scalars behind the larger of the two: [0] ticket, [1] poll result
__device__ __forceinline__ double* lds_tiles() {
  extern __shared__ __attribute__((aligned(1024))) char co_smem[];
  return reinterpret_cast<double*>(co_smem);
}
typedef __attribute__((address_space(3))) volatile int lds_vint;
__device__ __forceinline__ lds_vint* lds_scalars() {
  extern __shared__ __attribute__((aligned(1024))) char co_smem[];
  return (lds_vint*)(co_smem + CO_SH_OFF);
}

// Everything a consumer of this flag reads was stored WRITE-THROUGH (sc1: store_rows / store_block_inverse below), so
// there is no release fence (it would write back the whole XCD's dirty L2 lines: measured 8-10 us per publish with a
// 128 KB tile freshly stored); every storing wave drains its stores, then ONE lane stores the flag.
__device__ __forceinline__ void publish(const Ctx& c, guint* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // (debug_mute: tests/test_hip_linalg.py withholds one flag to see the launch give up cleanly instead of hanging)
  if (c.tid == 0 && (const uint32_t*)flag != c.p->debug_mute) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Number of leading k-blocks of [kb0, kb1) whose inputs are published (>= 1 after spinning), or -1: abort.  Wave 0
// polls, one block per lane; what it has seen is visible to every wave behind the barrier.
template <typename Ready>
__device__ __forceinline__ int wait_ready(const Ctx& c, int kb0, int kb1, Ready ready) {
  __syncthreads();                 // the previous result has been read; every wave is done with the buffers
  if (c.wave == 0) {
    int n;
    unsigned long long t0 = 0;
    const unsigned long long tw0 = c.p->trace ? realtime() : 0ull;
    for (int spin = 0;; ++spin) {
      const bool ok = (kb0 + c.lane < kb1) ? ready(kb0 + c.lane) : false;
      const unsigned long long m = __ballot(ok);
      n = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
      if (n >= 1) break;
      if ((spin & 15) == 15) {
        if (flag_load((const guint*)c.p->abort_word) != 0u) { n = -1; break; }
        const unsigned long long now = realtime();
        if (t0 == 0) t0 = now;
        else if (now - t0 > c.p->timeout_ticks) {
          if (c.lane == 0) __hip_atomic_store(c.p->abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          n = -1;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(4);
    }
    if (c.lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_scalars()[1] = n;
      if (c.p->trace) lds_scalars()[2] = lds_scalars()[2] + (int)(realtime() - tw0);
    }
  }
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(lds_scalars()[1]);
}

// acc += A * B^T over the 16-wide k-tiles [t0, t1) (multiples of 8: k-blocks are 8 tiles), A and B [128 rows][k] with
// pitches lda / ldb, k-tile t at column 16 t.  Every wave must be done with the tile buffers on entry (a barrier
// since their last use); they are free again behind the barrier that follows the call's last use by the caller.
__device__ __forceinline__ void run_tiles(const Ctx& c, d4 (&acc)[4][2], const double* Ag, int64_t lda, const double* Bg,
                                          int64_t ldb, int t0, int t1) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) void lds_void;
  const int wave_u = __builtin_amdgcn_readfirstlane(c.wave);
  const int fr_b0 = c.fr_b, fr_b1 = c.fr_b + 256;       // rows of the B image the wave's two 16-column sub-tiles read
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int piece = 2 * wave_u + h;
    const int prow = piece * 8 + (c.lane >> 3), pch = ((c.lane & 7) ^ (c.lane >> 3)) * 2;
    a_off[h] = (uint32_t)(prow * (int)lda + pch);
    b_off[h] = (uint32_t)(prow * (int)ldb + pch);
  }
  const double* a_base = Ag + (int64_t)t0 * 16;
  const double* b_base = Bg + (int64_t)t0 * 16;
  auto issue = [&](int buf) __attribute__((always_inline)) {
    double* sA = lds_tiles() + buf * 4096;
    double* sB = sA + 2048;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int piece = 2 * wave_u + h;
      __builtin_amdgcn_global_load_lds(a_base + a_off[h], (lds_void*)(sA + piece * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(b_base + b_off[h], (lds_void*)(sB + piece * 128), 16, 0, 0);
    }
    a_base += 16;
    b_base += 16;
  };
  typedef d2v FragA[2][4];
  typedef d2v FragB[2][2];
  auto load_frags = [&](int stage, FragA& fa, FragB& fb) __attribute__((always_inline)) {
    const double* sA = lds_tiles() + stage * 4096;
    const double* sB = sA + 2048;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[kc][mi] = *reinterpret_cast<const d2v*>(sA + (c.fr_a ^ (kc * 8)) + mi * 256);
      fb[kc][0] = *reinterpret_cast<const d2v*>(sB + (fr_b0 ^ (kc * 8)));
      fb[kc][1] = *reinterpret_cast<const d2v*>(sB + (fr_b1 ^ (kc * 8)));
    }
  };
  auto mfmas = [&](int, const FragA& fa, const FragB& fb) __attribute__((always_inline)) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = diag::mma(fa[kc][mi][j], fb[kc][ni][j], acc[mi][ni]);
  };
  // Four stages.  Tile t + 3 is requested while tile t is consumed (with one workgroup per CU nobody else hides the
  // latency of a request that misses L2), and the fragments of tile t + 1 are read from LDS into a second register set
  // while tile t's MFMAs run: behind a barrier the matrix pipes start at once instead of waiting for eight waves'
  // fragment reads to drain through the LDS pipe.  A wave's loads complete in order, so "at most 8 / 4 / 0 of mine
  // outstanding" says which tile has landed (4 loads per wave and tile).
  FragA fa0, fa1;
  FragB fb0, fb1;
  issue(0);
  issue(1);
  issue(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  load_frags(0, fa0, fb0);
  auto step = [&](int t, int stage, const FragA& fa, const FragB& fb, FragA& fan, FragB& fbn) __attribute__((always_inline)) {
    const int rem = t1 - t;        // tiles left, this one included
    if (rem >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // tile t + 1 has landed (t + 2 may be in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();               // ... for everybody; everybody has read tile t - 1's stage into registers
    if (rem > 3) issue((stage + 3) & 3);
    if (rem >= 2) load_frags((stage + 1) & 3, fan, fbn);
    mfmas(t, fa, fb);
  };
  for (int t = t0; t < t1; t += 4) {
    step(t, 0, fa0, fb0, fa1, fb1);
    step(t + 1, 1, fa1, fb1, fa0, fb0);
    step(t + 2, 2, fa0, fb0, fa1, fb1);
    step(t + 3, 3, fa1, fb1, fa0, fb0);
  }
#endif
}

__device__ __forceinline__ void zero_acc(d4 (&acc)[4][2]) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0, 0, 0, 0};
}

// ---- the 128 x 128 tile as an LDS image (pitch 130: rows stay 16-byte aligned), aliasing the tile buffers ---------------
constexpr int TP = 130;

// image[row][col] (TRANS: image[col][row]) = scale * acc, element (mi, ni, g) of a wave at row 64 wm + 16 mi + q + 4 g and
// column 32 wn + 16 ni + r -- or, REMAP, at column 16 (ni ? 7 - wn : wn) + r, which is how multiply_image deals the
// column sub-tiles.  Every wave must be behind its last read of the tile buffers / the previous image (a barrier); a
// barrier must follow before the image is read.
template <bool TRANS, bool REMAP>
__device__ __forceinline__ void stage_image(const Ctx& c, const d4 (&acc)[4][2], double scale) {
  double* T = lds_tiles();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = c.wm * 64 + 16 * mi + c.q + 4 * g;
        const int col = REMAP ? 16 * (ni ? 7 - c.wn : c.wn) + c.r : c.wn * 32 + 16 * ni + c.r;
        T[TRANS ? col * TP + row : row * TP + col] = scale * acc[mi][ni][g];
      }
}

// image -> global as whole 1-KB rows, 16 bytes per lane, written through (sc1) so that a consumer behind the tile's flag
// finds them without a release fence on this side.  `zeros`: a second tile cleared in the same sweep (plain stores).
__device__ __forceinline__ void store_rows(const Ctx& c, double* dst, int64_t ld, double* zeros) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const double* T = lds_tiles();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)((127 * ld + 128) * 8), 0x00020000);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int e = c.tid + CO_THREADS * it, n = e >> 6, c2 = (e & 63) * 2;
    const u32x4 v = *reinterpret_cast<const u32x4*>(T + n * TP + c2);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)((n * ld + c2) * 8), 0, 16);   // aux 16 = sc1
    if (zeros) *reinterpret_cast<d2v*>(zeros + (int64_t)n * ld + c2) = d2v{0.0, 0.0};
  }
#endif
}

// image -> global as fp32 rows (the GEMM-precision copy of the inverse the fp32 products read: written here, one pass
// over HBM less than a cast kernel behind the launch); `zeros`: a second fp32 tile cleared in the same sweep.
__device__ __forceinline__ void store_rows_f32(const Ctx& c, float* dst, int64_t ld, float* zeros) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const double* T = lds_tiles();
#pragma unroll 4
  for (int it = 0; it < 8; ++it) {
    const int e = c.tid + CO_THREADS * it, n = e >> 5, c4 = (e & 31) * 4;
    const d2v a = *reinterpret_cast<const d2v*>(T + n * TP + c4), b = *reinterpret_cast<const d2v*>(T + n * TP + c4 + 2);
    *reinterpret_cast<f4v*>(dst + (int64_t)n * ld + c4) = f4v{(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
    if (zeros) *reinterpret_cast<f4v*>(zeros + (int64_t)n * ld + c4) = f4v{0.f, 0.f, 0.f, 0.f};
  }
}

// The diagonal block's inverse as fp32 rows (lower triangular, zeros above)
__device__ __forceinline__ void store_block_inverse_f32(const Ctx& c, float* dst, int64_t ld) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const double* S = lds_tiles();
#pragma unroll 4
  for (int it = 0; it < 8; ++it) {
    const int e = c.tid + CO_THREADS * it, row = e >> 5, c4 = (e & 31) * 4;
    f4v v;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (c4 + u <= row) ? (float)S[(c4 + u) * DP + row + 1] : 0.f;
    *reinterpret_cast<f4v*>(dst + (int64_t)row * ld + c4) = v;
  }
}

// The inverse X of the diagonal block in S (csrc/diag128.h: X[i][c] at S[c][i + 1]) -> global, lower triangular with
// zeros above (TRANS: its transpose), 16-byte write-through stores.
template <bool TRANS>
__device__ __forceinline__ void store_block_inverse(const Ctx& c, double* dst, int64_t ld) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const double* S = lds_tiles();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)((127 * ld + 128) * 8), 0x00020000);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int e = c.tid + CO_THREADS * it, row = e >> 6, c2 = (e & 63) * 2;
    d2v v;
    if (TRANS) {   // dst[row = c][col = i] = X[i][c]
      v[0] = (row <= c2) ? S[row * DP + c2 + 1] : 0.0;
      v[1] = (row <= c2 + 1) ? S[row * DP + c2 + 2] : 0.0;
    } else {       // dst[row = i][col = c] = X[i][c]
      v[0] = (c2 <= row) ? S[c2 * DP + row + 1] : 0.0;
      v[1] = (c2 + 1 <= row) ? S[(c2 + 1) * DP + row + 1] : 0.0;
    }
    u32x4 u;
    __builtin_memcpy(&u, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, (int)((row * ld + c2) * 8), 0, 16);
  }
#endif
}

// acc(row, col) = sum_k image[row][k] * G[col][k], K = 128, G a lower-triangular 128 x 128 block in global memory
// (pitch ldg; zeros above its diagonal): the multiplication by the inverse of a diagonal block, with the tile that is
// multiplied never leaving the CU -- it is the LDS image -- and G's fragments loaded straight into registers (all 24
// 16-byte loads of a lane are issued before the first MFMA; they hit L2 / the Infinity Cache).  No barrier inside: the
// image is read-only, the waves run free.  A wave owns rows 64 wm .. + 63 and the 16-column sub-tiles c = wn and 7 - wn;
// sub-tile c needs k < 16 (c + 1) only, so every wave has 9 of 16 sub-tile steps and the four SIMDs equal work.
__device__ __forceinline__ void multiply_image(const Ctx& c, d4 (&acc)[4][2], const double* G, int64_t ldg) {
  const int wave_u = __builtin_amdgcn_readfirstlane(c.wave);
  const int wm_u = wave_u >> 2, c0 = wave_u & 3, c1 = 7 - c0;
  const double* T = lds_tiles() + (wm_u * 64 + c.r) * TP + 2 * c.q;
  const double* g0 = G + (int64_t)(16 * c0 + c.r) * ldg + 2 * c.q;
  const double* g1 = G + (int64_t)(16 * c1 + c.r) * ldg + 2 * c.q;
  d2v b0[4][2], b1[8][2];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      b1[kt][kc] = *reinterpret_cast<const d2v*>(g1 + 16 * kt + 8 * kc);
      if (kt < 4) b0[kt][kc] = *reinterpret_cast<const d2v*>(g0 + 16 * kt + 8 * kc);
    }
  zero_acc(acc);
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    if (kt > c1) continue;         // wave-uniform
    d2v a[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) a[mi][kc] = *reinterpret_cast<const d2v*>(T + mi * 16 * TP + 16 * kt + 8 * kc);
    if (kt < 4 && kt <= c0) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[mi][0] = diag::mma(a[mi][kc][j], b0[kt < 4 ? kt : 0][kc][j], acc[mi][0]);
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi][1] = diag::mma(a[mi][kc][j], b1[kt][kc][j], acc[mi][1]);
  }
}

// acc(row, col) = sum_k image[row][k] * image[col][k], K = 128, lower sub-tiles only: the update a freshly computed
// tile L[j][j-1] -- still the LDS image -- contributes to the diagonal tile below it, without a trip through memory.
// A wave owns rows 64 WM .. + 63 and the column sub-tiles WN and 7 - WN (the layout multiply_image leaves); sub-tile
// (row 4 WM + mi, column c) is computed iff row >= c: 9 of 16 on every SIMD.  WM, WN are constants so that the skipped
// products are absent from the code rather than branched around.  No barrier inside: the image is read-only.
template <int WM, int WN>
__device__ __forceinline__ void syrk_image_w(const Ctx& c, d4 (&acc)[4][2]) {
  constexpr int c0 = WN, c1 = 7 - WN;
  const double* Ta = lds_tiles() + (WM * 64 + c.r) * TP + 2 * c.q;
  const double* Tb0 = lds_tiles() + (16 * c0 + c.r) * TP + 2 * c.q;
  const double* Tb1 = lds_tiles() + (16 * c1 + c.r) * TP + 2 * c.q;
  zero_acc(acc);
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    d2v a[4][2], b0[2], b1[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        if (4 * WM + mi >= c0) a[mi][kc] = *reinterpret_cast<const d2v*>(Ta + mi * 16 * TP + 16 * kt + 8 * kc);
      b0[kc] = *reinterpret_cast<const d2v*>(Tb0 + 16 * kt + 8 * kc);
      if (4 * WM + 3 >= c1) b1[kc] = *reinterpret_cast<const d2v*>(Tb1 + 16 * kt + 8 * kc);
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (4 * WM + mi >= c0) acc[mi][0] = diag::mma(a[mi][kc][j], b0[kc][j], acc[mi][0]);
          if (4 * WM + mi >= c1) acc[mi][1] = diag::mma(a[mi][kc][j], b1[kc][j], acc[mi][1]);
        }
  }
}

__device__ __forceinline__ void syrk_image(const Ctx& c, d4 (&acc)[4][2]) {
  switch (__builtin_amdgcn_readfirstlane(c.wave)) {         // wave = 4 wm + wn
    case 0: syrk_image_w<0, 0>(c, acc); break;
    case 1: syrk_image_w<0, 1>(c, acc); break;
    case 2: syrk_image_w<0, 2>(c, acc); break;
    case 3: syrk_image_w<0, 3>(c, acc); break;
    case 4: syrk_image_w<1, 0>(c, acc); break;
    case 5: syrk_image_w<1, 1>(c, acc); break;
    case 6: syrk_image_w<1, 2>(c, acc); break;
    default: syrk_image_w<1, 3>(c, acc); break;
  }
}

// One matrix of the batch.  The pointers travel to the noinline task functions through memory; stored as generic
// pointers they would come back as flat_ accesses, so the struct keeps integers and the accessors rebuild global ones.
struct Mat {
  uintptr_t Ab_, Db_, Lb_, Xb_, L32_, fC_, fX_, info_;
  int64_t lda, ldl;
  int nblk;
  __device__ __forceinline__ double* Ab() const { return (double*)(gdouble*)Ab_; }
  __device__ __forceinline__ double* Db() const { return (double*)(gdouble*)Db_; }
  __device__ __forceinline__ double* Lb() const { return (double*)(gdouble*)Lb_; }
  __device__ __forceinline__ double* Xb() const { return (double*)(gdouble*)Xb_; }
  __device__ __forceinline__ float* L32() const { return (float*)(__attribute__((address_space(1))) float*)L32_; }
  __device__ __forceinline__ guint* fC() const { return (guint*)fC_; }
  __device__ __forceinline__ guint* fX() const { return (guint*)fX_; }
  __device__ __forceinline__ int32_t* info() const { return (int32_t*)(guint*)info_; }
};

// element (mi, ni, g) of a wave's accumulators: row0 + 16 mi + 4 g, col0 + 16 ni
#define CO_ROW0 (c.wm * 64 + c.q)
#define CO_COL0 (c.wn * 32 + c.r)

// Sum over the k-blocks [0, nkb) of A-block * B-block^T as they become available; false on abort.
template <typename Ready>
__device__ __forceinline__ bool accumulate(const Ctx& c, d4 (&acc)[4][2], const double* Ag, int64_t lda, const double* Bg,
                                           int64_t ldb, int nkb, Ready ready) {
  for (int kb = 0; kb < nkb;) {
    const int n = wait_ready(c, kb, nkb, ready);
    if (n < 0) return false;
    const unsigned long long tr0 = (c.p->trace && c.tid == 0) ? realtime() : 0ull;
    const unsigned long long tc0 = (c.p->trace && c.tid == 0) ? __builtin_readcyclecounter() : 0ull;
    run_tiles(c, acc, Ag, lda, Bg, ldb, kb * 8, (kb + n) * 8);
    if (c.p->trace && c.tid == 0) {
      lds_scalars()[3] = lds_scalars()[3] + (int)(realtime() - tr0);
      lds_scalars()[4] = lds_scalars()[4] + 1;
      lds_scalars()[5] = lds_scalars()[5] + (int)(__builtin_readcyclecounter() - tc0);
    }
    kb += n;
  }
  return true;
}

// The diagonal block in S (LDS): factor, invert, store, announce.  Shared by the two tasks that produce tile (j, j).
__device__ __forceinline__ void finish_diag(const Ctx& c, const Mat& m, int j, double* Ct) {
  const int nblk = m.nblk;
  double* S = lds_tiles();
  const bool trc = c.p->trace != nullptr && c.tid == 0;
  const unsigned long long td0 = trc ? realtime() : 0ull;
  diag::prepare(S, c.tid);
  __syncthreads();
  diag::factor_block<8>(S, c.tid, m.info(), (int64_t)j * 128, c.p->m_real, 1);
  const unsigned long long td1 = trc ? realtime() : 0ull;
  diag::inverse_levels(S, c.tid);
  if (trc) { lds_scalars()[4] = (int)(td1 - td0); lds_scalars()[5] = (int)(realtime() - td1); }   // (diagnostics: factor, inverse)
  // What the tiles below and beside this one wait for is the block's INVERSE (they multiply by it; nobody reads L[j][j]
  // inside the launch): it is stored and announced first, under a flag of its own (the unused diagonal entry of the X
  // flags), and the factor and the copies of the inverse that only later tasks / the caller read follow behind it --
  // four 128 KB store sweeps and their drain off the chain of diagonal blocks.
  store_block_inverse<false>(c, m.Db() + (int64_t)j * 128 * 128, 128);
  publish(c, m.fX() + j * nblk + j);
  diag::store_factor<CO_THREADS>(S, Ct, m.lda, c.tid);
  if (m.Lb()) {
    store_block_inverse<false>(c, m.Lb() + (int64_t)j * 128 * (m.ldl + 1), m.ldl);
    store_block_inverse<true>(c, m.Xb() + (int64_t)j * 128 * (m.ldl + 1), m.ldl);
    if (m.L32_) store_block_inverse_f32(c, m.L32() + (int64_t)j * 128 * (m.ldl + 1), m.ldl);
  }
  publish(c, m.fC() + j * nblk + j);
}

// ---------------- Cholesky tile (j, j): sum, factor + invert in LDS, publish the inverse ----------------
__device__ __attribute__((noinline)) bool task_chol_diag(const Ctx& c_in, const Mat& m_in, int j) {
  const Ctx c = c_in;               // private copies: values reached through a reference are reloaded behind every
  const Mat m = m_in;               // "memory"-clobbering wait in the k-loop
  const int nblk = m.nblk;
  d4 acc[4][2];
  zero_acc(acc);
  const double* Ag = m.Ab() + (int64_t)j * 128 * m.lda;
  const guint* fr = m.fC() + j * nblk;
  if (!accumulate(c, acc, Ag, m.lda, Ag, m.lda, j, [&](int k) { return flag_load(fr + k) != 0u; })) return false;
  double* Ct = m.Ab() + (int64_t)j * 128 * (m.lda + 1);
  __syncthreads();                 // the tile buffers alias S
  double* S = lds_tiles();
  const int row0 = CO_ROW0, col0 = CO_COL0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    double cin[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) cin[ni][g] = Ct[(int64_t)(row0 + 16 * mi + 4 * g) * m.lda + col0 + 16 * ni];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mm = row0 + 16 * mi + 4 * g, n = col0 + 16 * ni;
        S[mm * DP + n] = (n <= mm) ? cin[ni][g] - acc[mi][ni][g] : 0.0;
      }
  }
  finish_diag(c, m, j, Ct);
  return true;
}

// ---------------- the part of tile (j, j), j >= 2, that does not wait for the chain ----------------
// A[j][j] <- A[j][j] - sum_{k < j - 1} L[j][k] L[j][k]^T, in place, written through; flag: entry (j - 1, j) of the C flags.
__device__ __attribute__((noinline)) bool task_chol_presum(const Ctx& c_in, const Mat& m_in, int j) {
  const Ctx c = c_in;
  const Mat m = m_in;
  const int nblk = m.nblk;
  d4 acc[4][2];
  zero_acc(acc);
  const double* Aj = m.Ab() + (int64_t)j * 128 * m.lda;
  const guint* fj = m.fC() + j * nblk;
  if (!accumulate(c, acc, Aj, m.lda, Aj, m.lda, j - 1, [&](int k) { return flag_load(fj + k) != 0u; })) return false;
  double* Cd = m.Ab() + (int64_t)j * 128 * (m.lda + 1);
  const int row0 = CO_ROW0, col0 = CO_COL0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        double* e = Cd + (int64_t)(row0 + 16 * mi + 4 * g) * m.lda + col0 + 16 * ni;
        __hip_atomic_store(e, *e - acc[mi][ni][g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
  publish(c, m.fC() + (j - 1) * nblk + j);
  return true;
}

// ---------------- Cholesky tiles (j, j - 1) AND (j, j), j >= 1, by one workgroup ----------------
// The chain of the factorisation is  D(j-1) -> L[j][j-1] = T inv(L[j-1][j-1])^T -> D(j) = chol(A[j][j] - ... - L[j][j-1] L[j][j-1]^T).
// As two tasks the second step ended in a store sweep, its drain and a flag, and the third began by polling that flag
// and pulling the tile back through L2 for a 128 x 128 x 128 product of which half is not needed (16 us on one CU).  Here
// the tile stays where it was computed -- the LDS image -- and its symmetric update comes from there, lower sub-tiles
// only (9 us); its stores drain behind those MFMAs and the flag for the other users of L[j][j-1] goes up afterwards.
// What does not depend on the chain is done before it arrives: T's sum over k < j - 1 stays in the accumulators; the
// diagonal tile's, over the same k, is a task of its own (task_chol_presum: one workgroup doing both in turn had the
// second sum start when the first one's last block -- which arrives one column ahead of the chain -- was in).
__device__ __attribute__((noinline)) bool task_chol_fused(const Ctx& c_in, const Mat& m_in, int j) {
  const Ctx c = c_in;
  const Mat m = m_in;
  const int nblk = m.nblk, jm = j - 1;
  d4 acc[4][2];
  const guint *fj = m.fC() + j * nblk, *fjm = m.fC() + jm * nblk;
  const double* Aj = m.Ab() + (int64_t)j * 128 * m.lda;
  double* Cd = m.Ab() + (int64_t)j * 128 * (m.lda + 1);
  double* Co = m.Ab() + (int64_t)j * 128 * m.lda + (int64_t)jm * 128;
  const int row0 = CO_ROW0, col0 = CO_COL0;
  zero_acc(acc);
  if (!accumulate(c, acc, Aj, m.lda, m.Ab() + (int64_t)jm * 128 * m.lda, m.lda, jm,
                  [&](int k) { return flag_load(fj + k) != 0u && flag_load(fjm + k) != 0u; }))
    return false;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[mi][ni][g] = Co[(int64_t)(row0 + 16 * mi + 4 * g) * m.lda + col0 + 16 * ni] - acc[mi][ni][g];
  __syncthreads();                 // every wave is behind its last fragment read of the tile buffers
  stage_image<false, false>(c, acc, 1.0);
  const guint* fd = m.fX() + jm * nblk + jm;      // "inv(L[j-1][j-1]) is stored"
  const int n = wait_ready(c, 0, 1, [&](int) { return flag_load(fd) != 0u; });   // (its barriers publish the image)
  if (n < 0) return false;
  multiply_image(c, acc, m.Db() + (int64_t)jm * 128 * 128, 128);
  __syncthreads();                 // the image has been read by everybody
  stage_image<false, true>(c, acc, 1.0);
  __syncthreads();
  store_rows(c, Co, m.lda, nullptr);
  syrk_image(c, acc);              // acc = L[j][j-1] L[j][j-1]^T, lower sub-tiles, while the rows drain
  publish(c, m.fC() + j * nblk + jm);   // (its barrier: every wave is done with the image, which S now overwrites)
  // S = (A[j][j] - earlier sums) - acc.  For j >= 2 the earlier sums were subtracted in place by task_chol_presum, long
  // ago as a rule (its flag: the unused entry (j - 1, j) of the C flags); its stores were written through, read them past
  // the L1.  Element (mi, ni, g) of syrk_image: row 64 wm + 16 mi + q + 4 g, column 16 (ni ? 7 - wn : wn) + r.
  if (jm > 0) {
    const guint* fp = m.fC() + jm * nblk + j;
    if (wait_ready(c, 0, 1, [&](int) { return flag_load(fp) != 0u; }) < 0) return false;
  }
  double* S = lds_tiles();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    double cin[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        cin[ni][g] = __hip_atomic_load(Cd + (int64_t)(row0 + 16 * mi + 4 * g) * m.lda + 16 * (ni ? 7 - c.wn : c.wn) + c.r,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mm = row0 + 16 * mi + 4 * g, nn = 16 * (ni ? 7 - c.wn : c.wn) + c.r;
        S[mm * DP + nn] = (nn <= mm) ? cin[ni][g] - acc[mi][ni][g] : 0.0;
      }
  }
  finish_diag(c, m, j, Cd);
  return true;
}

// ---------------- Cholesky tile (i, j), i > j ----------------
__device__ __attribute__((noinline)) bool task_chol_off(const Ctx& c_in, const Mat& m_in, int i, int j) {
  const Ctx c = c_in;               // private copies: values reached through a reference are reloaded behind every
  const Mat m = m_in;               // "memory"-clobbering wait in the k-loop
  const int nblk = m.nblk;
  d4 acc[4][2];
  zero_acc(acc);
  const guint *fi = m.fC() + i * nblk, *fj = m.fC() + j * nblk;
  if (!accumulate(c, acc, m.Ab() + (int64_t)i * 128 * m.lda, m.lda, m.Ab() + (int64_t)j * 128 * m.lda, m.lda, j,
                  [&](int k) { return flag_load(fi + k) != 0u && flag_load(fj + k) != 0u; }))
    return false;
  // T = A[i][j] - sum becomes an LDS image, is multiplied by inv(L[j][j])^T there and leaves as whole rows
  const bool trc = c.p->trace != nullptr && c.tid == 0;
  const unsigned long long ts0 = trc ? realtime() : 0ull;
  double* Ct = m.Ab() + (int64_t)i * 128 * m.lda + (int64_t)j * 128;
  const int row0 = CO_ROW0, col0 = CO_COL0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[mi][ni][g] = Ct[(int64_t)(row0 + 16 * mi + 4 * g) * m.lda + col0 + 16 * ni] - acc[mi][ni][g];
  __syncthreads();                 // every wave is behind its last fragment read of the tile buffers
  stage_image<false, false>(c, acc, 1.0);
  const guint* fd = m.fX() + j * nblk + j;        // "inv(L[j][j]) is stored" (task_chol_diag)
  const int n = wait_ready(c, 0, 1, [&](int) { return flag_load(fd) != 0u; });   // (its barriers publish the image)
  if (n < 0) return false;
  const unsigned long long ts1 = trc ? realtime() : 0ull;
  multiply_image(c, acc, m.Db() + (int64_t)j * 128 * 128, 128);
  const unsigned long long ts2 = trc ? realtime() : 0ull;
  if (trc) { lds_scalars()[4] = (int)(ts1 - ts0); lds_scalars()[5] = (int)(ts2 - ts1); }
  __syncthreads();                 // the image has been read by everybody
  stage_image<false, true>(c, acc, 1.0);
  __syncthreads();
  store_rows(c, Ct, m.lda, nullptr);
  publish(c, m.fC() + i * nblk + j);
  return true;
}

// ---------------- inverse tile (i, j), i > j ----------------
__device__ __attribute__((noinline)) bool task_inverse(const Ctx& c_in, const Mat& m_in, int i, int j) {
  const Ctx c = c_in;               // private copies: values reached through a reference are reloaded behind every
  const Mat m = m_in;               // "memory"-clobbering wait in the k-loop
  const int nblk = m.nblk;
  const int64_t ldl = m.ldl;
  d4 acc[4][2];
  zero_acc(acc);
  const guint *fi = m.fC() + i * nblk, *fjj = m.fC() + j * nblk + j, *fx = m.fX() + j;
  // A: L[i][j + kb]; B: Linv[j + kb][j]^T
  if (!accumulate(c, acc, m.Ab() + (int64_t)i * 128 * m.lda + (int64_t)j * 128, m.lda,
                  m.Xb() + (int64_t)j * 128 * ldl + (int64_t)j * 128, ldl, i - j, [&](int kk) {
                    const int k = j + kk;
                    return flag_load(fi + k) != 0u && flag_load(kk == 0 ? fjj : fx + k * nblk) != 0u;
                  }))
    return false;
  // With S the sum: Linv[i][j] = -inv(L[i][i]) S.  The image is S^T, the product image * inv(L[i][i])^T = (inv S)^T
  // comes out transposed: as it stands it is Linv[i][j]^T (kept for the tiles below this one), transposed it is Linv[i][j].
  const bool trc = c.p->trace != nullptr && c.tid == 0;
  const unsigned long long ts0 = trc ? realtime() : 0ull;
  __syncthreads();
  stage_image<true, false>(c, acc, -1.0);
  const guint* fd = m.fX() + i * nblk + i;        // "inv(L[i][i]) is stored"
  const int n = wait_ready(c, 0, 1, [&](int) { return flag_load(fd) != 0u; });
  if (n < 0) return false;
  const unsigned long long ts1 = trc ? realtime() : 0ull;
  multiply_image(c, acc, m.Db() + (int64_t)i * 128 * 128, 128);
  const unsigned long long ts2 = trc ? realtime() : 0ull;
  if (trc) { lds_scalars()[4] = (int)(ts1 - ts0); lds_scalars()[5] = (int)(ts2 - ts1); }
  double* Tt = m.Xb() + (int64_t)j * 128 * ldl + (int64_t)i * 128;
  double* Lt = m.Lb() + (int64_t)i * 128 * ldl + (int64_t)j * 128;
  double* Ut = m.Lb() + (int64_t)j * 128 * ldl + (int64_t)i * 128;   // the mirrored block above the diagonal: zeros
  __syncthreads();
  stage_image<false, true>(c, acc, 1.0);
  __syncthreads();
  store_rows(c, Tt, ldl, Ut);
  __syncthreads();
  stage_image<true, true>(c, acc, 1.0);
  __syncthreads();
  store_rows(c, Lt, ldl, nullptr);
  if (m.L32_)
    store_rows_f32(c, m.L32() + (int64_t)i * 128 * ldl + (int64_t)j * 128, ldl, m.L32() + (int64_t)j * 128 * ldl + (int64_t)i * 128);
  publish(c, m.fX() + i * nblk + j);
  return true;
}

// One matrix: claim tiles until the list is exhausted.  Returns false on abort.
__device__ __forceinline__ bool run_matrix(const Ctx& c, const CoopOrder& ord, int b) {
  const CoopParams& p = *c.p;
  Mat m;
  m.nblk = p.nblk;
  m.Ab_ = (uintptr_t)(p.A + (int64_t)b * p.sA);
  m.Db_ = (uintptr_t)(p.Dinv + (int64_t)b * p.sD);
  m.Lb_ = p.Linv ? (uintptr_t)(p.Linv + (int64_t)b * p.sL) : 0;
  m.Xb_ = p.Linv ? (uintptr_t)(p.XT + (int64_t)b * p.sX) : 0;
  m.L32_ = p.Linv32 ? (uintptr_t)(p.Linv32 + (int64_t)b * p.sL) : 0;
  m.lda = p.lda; m.ldl = (int64_t)p.nblk * 128;
  uint32_t* sync = p.sync + (int64_t)b * p.sS;
  m.fC_ = (uintptr_t)(sync + CO_SYNC_HEAD);
  m.fX_ = (uintptr_t)(sync + CO_SYNC_HEAD + p.nblk * p.nblk);
  m.info_ = (uintptr_t)(p.info + b);
  for (;;) {
    __syncthreads();
    if (c.tid == 0) lds_scalars()[0] = (int)__hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(lds_scalars()[0]);
    if (ticket >= p.ntasks) return true;
    const unsigned code = ord.t[ticket];
    const int kind = code >> 12, i = (code >> 6) & 63, j = code & 63;
    unsigned long long* tr = p.trace ? p.trace + ((int64_t)b * p.ntasks + ticket) * 8 : nullptr;
    if (tr && c.tid == 0) { tr[0] = code; tr[1] = blockIdx.x; tr[2] = realtime(); }
    lds_scalars()[2] = 0;            // the task's time in polls (diagnostics)
    lds_scalars()[3] = 0;            // ... in the accumulation loops, and their number
    lds_scalars()[4] = 0;
    lds_scalars()[5] = 0;
    bool ok;
    if (kind == 2) ok = task_chol_fused(c, m, j);
    else if (kind == 3) ok = task_chol_presum(c, m, j);
    else if (kind != 0) ok = task_inverse(c, m, i, j);
    else if (i == j) ok = task_chol_diag(c, m, j);
    else ok = task_chol_off(c, m, i, j);
    if (tr && c.tid == 0) { tr[3] = realtime(); tr[4] = (unsigned long long)lds_scalars()[2];
                             tr[5] = (unsigned long long)lds_scalars()[3]; tr[6] = (unsigned long long)lds_scalars()[4];
                             tr[7] = (unsigned long long)lds_scalars()[5]; }
    if (!ok) {
      if (c.tid == 0) atomicCAS(m.info(), 0, -7);
      return false;
    }
  }
}
}  // namespace

__global__ __launch_bounds__(CO_THREADS) void coop_factor_kernel(const CoopParams p, const CoopOrder ord) {
  Ctx c;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6;
  c.wm = c.wave >> 2; c.wn = c.wave & 3; c.r = c.lane & 15; c.q = c.lane >> 4;
  c.fr_a = (c.wm * 64 + c.r) * 16 + ((c.q ^ (c.r & 7)) * 2);
  c.fr_b = (c.wn * 32 + c.r) * 16 + ((c.q ^ (c.r & 7)) * 2);
  c.p = &p;
  // Blocks b, b + 8, ... share an XCD (observed round-robin dispatch; used for locality only, never for correctness):
  // numbering the workgroups XCD by XCD and cutting that sequence into one contiguous run per matrix gives every matrix
  // the same number of workgroups (+-1) whatever the batch size, almost all of them inside one XCD.
  const int bid = blockIdx.x, nwg = (int)gridDim.x, per_xcd = nwg >> 3;
  const int v = (bid & 7) * per_xcd + (bid >> 3);
  const int nclus = min(p.batch, nwg);
  const int cl = (int)(((long long)v * nclus) / nwg);
  for (int b = cl; b < p.batch; b += nclus)
    if (!run_matrix(c, ord, b)) return;
  // Out of work: help whichever matrix still has unclaimed tiles (any workgroup may join any matrix's list).
  for (;;) {
    __syncthreads();
    if (c.wave == 0) {
      int found = -1;
      for (int base = 0; base < p.batch && found < 0; base += 64) {
        const int mb = base + c.lane;
        const int idx = (cl + 1 + mb) % p.batch;
        const bool has = mb < p.batch && flag_load((const guint*)(p.sync + (int64_t)idx * p.sS)) < (uint32_t)p.ntasks;
        const unsigned long long mk = __ballot(has);
        if (mk != 0ull) found = (cl + 1 + base + __builtin_ctzll(mk)) % p.batch;
      }
      if (c.lane == 0) lds_scalars()[0] = found;
    }
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(lds_scalars()[0]);
    if (b < 0) return;
    if (!run_matrix(c, ord, b)) return;
  }
}

// ---- the order: a host-side model of the execution above -------------------------------------------------------------
namespace {
struct OrderKey { int nblk, G, inv, fused; bool operator<(const OrderKey& o) const { return std::tie(nblk, G, inv, fused) < std::tie(o.nblk, o.G, o.inv, o.fused); } };

// Rough phase times (us) on one CU: a 128^3 fp64 accumulation step, the multiply by a diagonal block's inverse with its
// park / reload, the symmetric update from the LDS image, the diagonal block up to the flag of its inverse and the
// stores behind that flag, flag latency, a claim.  Only their ratios matter for the order.
constexpr double T_OP = 16.0, T_P = 15.0, T_S = 9.0, T_D = 47.0, T_DT = 8.0, T_FLAG = 2.0, T_CLAIM = 1.0;

// What a step waits for: tiles C(i,j) = i * nblk + j, X(i,j) = nblk^2 + i * nblk + j (a diagonal C tile: everything of
// it), and INV(j) = 2 nblk^2 + j: the inverse of diagonal block j, announced ahead of the rest of C(j,j).
struct Step { int dep[2]; double dur; };

// kind 0: C(i,j); 1: X(i,j); 2: the fused pair C(j,j-1) + C(j,j); 3: PRE(j), the early sums of C(j,j) -- named by the
// unused tile id (j-1, j)
static void steps_of(int kind, int i, int j, int nblk, std::vector<Step>& st) {
  const int X0 = nblk * nblk, I0 = 2 * X0;
  st.clear();
  if (kind == 3) {
    for (int k = 0; k < j - 1; ++k) st.push_back({{j * nblk + k, -1}, T_OP});
    st.push_back({{-1, -1}, 3.0});
  } else if (kind == 2) {
    for (int k = 0; k < j - 1; ++k) st.push_back({{j * nblk + k, (j - 1) * nblk + k}, T_OP});
    st.push_back({{I0 + j - 1, -1}, T_P});       // [size - 3]: the tile C(j,j-1) exists behind this step
    st.push_back({{-1, -1}, T_S});
    st.push_back({{j >= 2 ? (j - 1) * nblk + j : -1, -1}, T_D});
  } else if (kind == 0) {
    for (int k = 0; k < j; ++k) st.push_back({{i * nblk + k, i != j ? j * nblk + k : -1}, T_OP});
    if (i == j) st.push_back({{-1, -1}, T_D});
    else st.push_back({{I0 + j, -1}, T_P});
  } else {
    for (int k = j; k < i; ++k) st.push_back({{i * nblk + k, k > j ? X0 + k * nblk + j : j * nblk + j}, T_OP});
    st.push_back({{I0 + i, -1}, T_P});
  }
}

static CoopOrder build_order(int nblk, int G, bool inv, bool fused, int* ntasks_out) {
  const int X0 = nblk * nblk, I0 = 2 * X0, NT = I0 + nblk;
  // tasks are named by the tile they finish LAST: the fused task of column j - 1 / row j is C(j,j)
  std::vector<int> kind(NT, -1), prod(NT, -1);     // of a task id; of a tile / INV id: the task that makes it
  std::vector<double> avail(NT, 0.0), bl(NT, 0.0);
  std::vector<char> claimed(NT, 0);
  std::vector<int> all;
  for (int j = 0; j < nblk; ++j)
    for (int i = j; i < nblk; ++i) {
      const int t = i * nblk + j;
      if (fused && i == j + 1) { prod[t] = i * nblk + i; continue; }          // made by the fused task of row i
      if (fused && i == j && j >= 2) {                                          // its early sums, listed ahead of it
        const int pre = (j - 1) * nblk + j;
        kind[pre] = 3; prod[pre] = pre;
        all.push_back(pre);
      }
      kind[t] = (fused && i == j && j >= 1) ? 2 : 0;
      prod[t] = t;
      all.push_back(t);
      if (i == j) prod[I0 + j] = t;
    }
  if (inv)
    for (int j = 0; j < nblk; ++j)
      for (int i = j + 1; i < nblk; ++i) { kind[X0 + i * nblk + j] = 1; prod[X0 + i * nblk + j] = X0 + i * nblk + j; all.push_back(X0 + i * nblk + j); }
  auto ij_of = [&](int t, int& i, int& j) { const int u = t % X0; i = u / nblk; j = u % nblk; };
  // bottom levels: only a task's last steps sit on a chain (the earlier ones are accumulated ahead of time)
  std::vector<std::vector<int>> succ(NT);
  std::vector<Step> st;
  std::vector<double> last(NT, 0.0);
  for (int t : all) {
    int i, j;
    ij_of(t, i, j);
    steps_of(kind[t], i, j, nblk, st);
    const size_t nchain = kind[t] == 2 ? 3 : 2;
    for (size_t u = st.size() > nchain ? st.size() - nchain : 0; u < st.size(); ++u) last[t] += st[u].dur;
    for (const Step& sp : st)
      for (int d : sp.dep)
        if (d >= 0) succ[prod[d]].push_back(t);
  }
  // every dependency of a task precedes it in (column-major C, then column-major X) order -- the fused task of row i,
  // listed in column i, after column i - 1 whose tiles it needs: reverse sweep
  for (auto it = all.rbegin(); it != all.rend(); ++it) {
    double m = 0.0;
    for (int sc : succ[*it]) m = std::max(m, bl[sc]);
    bl[*it] = last[*it] + m;
  }
  std::vector<double> wfree(G, 0.0);
  std::vector<int> remaining = all;
  CoopOrder ord{};
  int n = 0;
  const double slack = 6.0;
  while (!remaining.empty()) {
    const int w = (int)(std::min_element(wfree.begin(), wfree.end()) - wfree.begin());
    const double t0 = wfree[w];
    int best = -1;
    double best_key0 = 0, best_key1 = 0, best_fin = 0, best_mid = 0;
    for (size_t ci = 0; ci < remaining.size(); ++ci) {
      const int c = remaining[ci];
      int i, j;
      ij_of(c, i, j);
      steps_of(kind[c], i, j, nblk, st);
      bool ok = true;
      double t = t0 + T_CLAIM, work = 0.0, mid = 0.0;
      for (size_t u = 0; u < st.size(); ++u) {
        const Step& sp = st[u];
        for (int d : sp.dep)
          if (d >= 0) {
            if (!claimed[prod[d]]) { ok = false; break; }
            t = std::max(t, avail[d] + T_FLAG);
          }
        if (!ok) break;
        t += sp.dur;
        work += sp.dur;
        if (kind[c] == 2 && u + 3 == st.size()) mid = t;
      }
      if (!ok) continue;
      const double blocked = t - t0 - T_CLAIM - work;
      const double k0 = blocked <= slack ? 0.0 : 1.0, k1 = blocked <= slack ? -bl[c] : blocked;
      if (best < 0 || k0 < best_key0 || (k0 == best_key0 && k1 < best_key1)) {
        best = (int)ci; best_key0 = k0; best_key1 = k1; best_fin = t; best_mid = mid;
      }
    }
    const int c = remaining[best];
    int i, j;
    ij_of(c, i, j);
    ord.t[n++] = (uint16_t)((kind[c] << 12) | ((kind[c] == 3 ? j : i) << 6) | j);
    claimed[c] = 1;
    if ((kind[c] == 0 || kind[c] == 2) && i == j) {          // a diagonal block: its inverse first, the rest behind it
      avail[I0 + j] = best_fin;
      avail[c] = best_fin + T_DT;
      if (kind[c] == 2) avail[i * nblk + i - 1] = best_mid + T_S;       // (the flag goes up behind the symmetric update)
      wfree[w] = best_fin + T_DT;
    } else {
      avail[c] = best_fin;
      wfree[w] = best_fin;
    }
    remaining.erase(remaining.begin() + best);
  }
  *ntasks_out = n;
  return ord;
}

static const CoopOrder& cached_order(int nblk, int G, bool inv, bool fused, int* ntasks) {
  static std::map<OrderKey, std::pair<CoopOrder, int>> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const OrderKey key{nblk, G, inv ? 1 : 0, fused ? 1 : 0};
  auto it = cache.find(key);
  if (it == cache.end()) {
    int n = 0;
    CoopOrder o = build_order(nblk, G, inv, fused, &n);
    it = cache.emplace(key, std::make_pair(o, n)).first;
  }
  *ntasks = it->second.second;
  return it->second.first;
}
}  // namespace

static unsigned long long* g_coop_trace = nullptr;
static int g_coop_mute_i = -1, g_coop_mute_j = -1;      // diagnostics: the Cholesky tile of matrix 0 whose flag is withheld

bool coop_supported(int64_t Mp, bool inverse) {
  const int64_t nblk = Mp / 128;
  const int64_t ntasks = nblk * (nblk + 1) / 2 + (inverse ? nblk * (nblk - 1) / 2 : 0);
  return Mp % 128 == 0 && nblk >= 1 && nblk <= 63 && ntasks <= CO_MAX_TASKS;
}

// words of device memory factor_coop needs at `sync` (zeroed by factor_coop itself)
size_t coop_sync_words(int64_t Mp, int64_t batch) {
  const int64_t nblk = Mp / 128;
  return (size_t)(batch * (CO_SYNC_HEAD + 2 * nblk * nblk) + 32);
}

// In-place Cholesky of `batch` padded (Mp,Mp) fp64 matrices (identity-padded beyond m_real), the inverses of the
// diagonal blocks in Dinv and -- when Linv is given -- the inverse of the factor in Linv (pitch Mp; its blocks above
// the diagonal are NOT written) with XT (batch, Mp, Mp) as scratch.  info as potrf_padded; -7: a hand-off timed out.
int factor_coop(double* A, int64_t Mp, int64_t lda, int64_t stride, int64_t batch, int64_t m_real, double* Dinv,
                double* Linv, double* XT, uint32_t* sync, int32_t* info, hipStream_t s, float* Linv32, bool sync_cleared) {
  GPZ_REQUIRE(coop_supported(Mp, Linv != nullptr), "factor_coop: order %lld not supported", (long long)Mp);
  GPZ_REQUIRE(!Linv || XT, "factor_coop: the inverse needs its transposed scratch");
  GPZ_REQUIRE(lda * 128 * 8 < (1ll << 31), "factor_coop: leading dimension too large for 32-bit lane offsets");
  const int nblk = (int)(Mp / 128);
  CoopParams p;
  p.A = A; p.lda = lda; p.sA = stride;
  p.Dinv = Dinv; p.sD = (int64_t)nblk * 128 * 128;
  p.Linv = Linv; p.sL = Mp * Mp;
  p.XT = XT; p.sX = Mp * Mp;
  p.Linv32 = Linv ? Linv32 : nullptr;
  p.info = info; p.m_real = m_real;
  p.sS = CO_SYNC_HEAD + 2 * nblk * nblk;
  p.sync = sync + 32; p.abort_word = sync;
  p.nblk = nblk; p.batch = (int)batch;
  p.gmax = std::max(1, std::min(32, 3 * nblk / 2));
  // 5 s of the 100 MHz clock: far beyond any wait a healthy launch sees, even beside another process's kernels
  // (GPZ_COOP_TIMEOUT_MS: the tests' give-up case does not want to wait that long)
  static const unsigned long long timeout_ticks = [] {
    const char* e = std::getenv("GPZ_COOP_TIMEOUT_MS");
    const long ms = e ? atol(e) : 5000;
    return (unsigned long long)(ms > 0 ? ms : 5000) * 100000ull;
  }();
  p.timeout_ticks = timeout_ticks;
  p.trace = g_coop_trace;
  p.debug_mute = (g_coop_mute_i >= 0 && g_coop_mute_i < nblk && g_coop_mute_j >= 0 && g_coop_mute_j <= g_coop_mute_i)
                     ? p.sync + CO_SYNC_HEAD + g_coop_mute_i * nblk + g_coop_mute_j : nullptr;
  const int nclus = (int)std::min<int64_t>(batch, 256);
  const int nwg = std::min(256, (nclus * p.gmax + 7) / 8 * 8);
  const int G = std::max(1, nwg / nclus);
  // GPZ_COOP_UNFUSED: tiles (j, j-1) and (j, j) as two tasks, as in round 4 (A/B timing)
  static const bool fused = std::getenv("GPZ_COOP_UNFUSED") == nullptr;
  const CoopOrder& ord = cached_order(nblk, G, Linv != nullptr, fused, &p.ntasks);
  static bool attr_set[64] = {};
  int dev = 0;
  GPZ_HIP_OK(hipGetDevice(&dev));
  if (!attr_set[dev & 63]) {
    GPZ_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(coop_factor_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)CO_LDS_BYTES));
    attr_set[dev & 63] = true;
  }
  if (!sync_cleared) GPZ_HIP_OK(hipMemsetAsync(sync, 0, sizeof(uint32_t) * coop_sync_words(Mp, batch), s));
  hipLaunchKernelGGL(coop_factor_kernel, dim3(nwg), dim3(CO_THREADS), CO_LDS_BYTES, s, p, ord);
  GPZ_LAUNCH_OK();
  return 0;
}

}  // namespace gpz

// Diagnostics: a device buffer of 8 * batch * ntasks words that the following factor_coop launches fill with, per
// (matrix, ticket): task code, workgroup, claim time, end time, time spent polling (100 MHz ticks).  null: off.
// Diagnostics: the following factor_coop launches never publish Cholesky tile (i, j) of matrix 0, so everything that
// needs it waits until the launch's timeout, raises the abort word and leaves with info = -7.  (-1, -1): off.
extern "C" int gpz_debug_coop_mute(int i, int j) {
  gpz::g_coop_mute_i = i;
  gpz::g_coop_mute_j = j;
  return 0;
}

// Diagnostics (host only, no GPU call): the claim order factor_coop would launch with for `nblk` block columns and G
// workgroups per matrix, as task codes kind << 12 | i << 6 | j (kind 0: C(i,j), 1: X(i,j), 2: C(j,j-1) + C(j,j), 3: the early part of C(j,j)); returns
// their number.  tests/test_abi.py checks that every order is a linear extension of the tile DAG -- the property the
// launch's progress rests on.
extern "C" int gpz_debug_coop_order(int nblk, int G, int inverse, int fused, uint16_t* out, int cap) {
  if (nblk < 1 || nblk > 63 || G < 1 || !gpz::coop_supported((int64_t)nblk * 128, inverse != 0)) return -1;
  int n = 0;
  const gpz::CoopOrder& o = gpz::cached_order(nblk, G, inverse != 0, fused != 0, &n);
  if (n > cap) return -1;
  for (int k = 0; k < n; ++k) out[k] = o.t[k];
  return n;
}

extern "C" int gpz_debug_coop_trace(void* buf) {
  gpz::g_coop_trace = static_cast<unsigned long long*>(buf);
  return 0;
}
