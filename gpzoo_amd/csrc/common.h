// common.h -- shared host/device helpers for libgpzoo_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/gpzoo_hip.h"

namespace gpz {

// Every internal matrix dimension is padded to a multiple of this (the GEMM
// block tile), so the MFMA kernels never see a ragged edge.
constexpr int PAD = 128;

inline int64_t pad_up(int64_t v, int64_t m = PAD) { return (v + m - 1) / m * m; }

void set_error(const char* fmt, ...);

#define GPZ_HIP_OK(expr)                                                          \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      gpz::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -2;                                                                  \
    }                                                                             \
  } while (0)

#define GPZ_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      gpz::set_error(__VA_ARGS__);        \
      return -1;                          \
    }                                     \
  } while (0)

#define GPZ_LAUNCH_OK()                   \
  GPZ_HIP_OK(hipGetLastError())

// Carves a caller-provided workspace into 256-byte aligned pieces.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    off = (off + 255) & ~size_t(255);
    T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return r;
  }
  size_t used() const { return (off + 255) & ~size_t(255); }
};

// Covariance fill with padded extents (csrc/kfill.hip): rows/columns beyond the real extents are written as
// zero (identity on the diagonal when pad_identity).  `bad_group`: device int32 that is set to -1 when a group
// id of a multi-group kernel lies outside [0, n_groups) (the id is then read as 0: no out-of-bounds access).
int kfill_padded(const gpz_kernel_desc* k, const void* A, int64_t nA, int64_t pA, const void* B, int64_t nB,
                 int64_t pB, int d, const int64_t* gA, const int64_t* gB, void* K, int64_t ldk, int64_t stride,
                 double jitter, int pad_identity, int out_dtype, hipStream_t s, int32_t* bad_group = nullptr);

// ---- profiling slots (HIP events around the dominant kernels) -------------
enum ProfSlot { PROF_KFILL = 0, PROF_STAGE1 = 1, PROF_STAGE2 = 2, PROF_POTRF_TRAIL = 3,
                PROF_POTRF_ALL = 4, PROF_TRTRI = 5, PROF_FINAL = 6, PROF_NSLOTS = 8 };
bool prof_enabled();
void prof_begin(int slot, hipStream_t s);
void prof_end(int slot, hipStream_t s);

}  // namespace gpz
