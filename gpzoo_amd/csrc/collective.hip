// collective.hip -- the exchanges of the sharded evaluations: a sum of fp64 scalars over the ranks (Gaussian ELBO), and
// for the latent-sharded Poisson NSF step (reference likelihoods.py:49-53, 74-97: the rate mixes latents) an all-gather
// of q(F)'s moments with the matching reduce-scatter / all-reduce of their gradients.
//
// Latent GPs shard across the GPUs of a node with no data-path collective; what remains is
// one all-reduce of the per-rank ELBO (1-3 doubles) per evaluation -- RCCL over xGMI, latency bound.  The
// Python layer uses torch.distributed (backend "nccl" == RCCL) for it; these entry points give a plain-C
// client of the ABI the same exchange without Python: one communicator per process (rank <-> GPU), created
// from a 128-byte id that rank 0 generates and the caller distributes by whatever means it has (file, MPI,
// socket, torch's store).
//
// RCCL is bound at run time (dlopen of librccl.so.1 -- the copy the process already holds when PyTorch is
// loaded, /opt/rocm's otherwise), so single-GPU users of the library do not need it at all.
#include "common.h"

#include <cstdio>

#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace gpz {
namespace {

struct RcclId { char internal[128]; };          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;                          // ncclComm_t
constexpr int kRcclFloat64 = 8;                  // ncclFloat64
constexpr int kRcclFloat32 = 7;                  // ncclFloat32
constexpr int kRcclUint8 = 1;                    // ncclUint8
constexpr int kRcclSum = 0;                      // ncclSum

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_why[256] = "not tried";               // why RCCL is unavailable (dlerror() is cleared by every call: kept here)

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names)                       // a copy already in the process (PyTorch's) wins
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h)
    for (const char* n : names) {
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      const char* e = dlerror();                    // the failing LOAD attempt's reason, read exactly once
      snprintf(g_rccl_why, sizeof(g_rccl_why), "library missing: %s", e ? e : "dlopen failed");
    }
  if (!h) return;
  Rccl r;
  r.handle = h;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
  r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (r.GetUniqueId && r.CommInitRank && r.AllReduce && r.AllGather && r.ReduceScatter && r.CommDestroy) g_rccl = r;
  else snprintf(g_rccl_why, sizeof(g_rccl_why), "symbol missing: librccl was loaded but lacks ncclGetUniqueId / "
                "ncclCommInitRank / ncclAllReduce / ncclAllGather / ncclReduceScatter / ncclCommDestroy");
}

int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  GPZ_REQUIRE(g_rccl.handle, "RCCL is not available (%s)", g_rccl_why);
  return 0;
}

#define GPZ_RCCL_OK(expr)                                                                          \
  do {                                                                                             \
    const int _r = (expr);                                                                         \
    if (_r != 0) {                                                                                 \
      set_error("%s -> RCCL error %d (%s)", #expr, _r, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
      return -3;                                                                                   \
    }                                                                                              \
  } while (0)

}  // namespace
}  // namespace gpz

using namespace gpz;

extern "C" int gpz_comm_unique_id(void* id128_host) {
  GPZ_REQUIRE(id128_host, "gpz_comm_unique_id: null pointer");
  if (int rc = need_rccl()) return rc;
  RcclId id;
  GPZ_RCCL_OK(g_rccl.GetUniqueId(&id));
  memcpy(id128_host, id.internal, sizeof(id.internal));
  return 0;
}

extern "C" int gpz_comm_init(void** comm_out, int32_t world, int32_t rank, const void* id128_host) {
  GPZ_REQUIRE(comm_out && id128_host, "gpz_comm_init: null pointer");
  GPZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "gpz_comm_init: bad rank %d of %d", rank, world);
  if (int rc = need_rccl()) return rc;
  RcclId id;
  memcpy(id.internal, id128_host, sizeof(id.internal));
  RcclComm c = nullptr;
  GPZ_RCCL_OK(g_rccl.CommInitRank(&c, world, id, rank));     // binds the calling thread's current HIP device
  *comm_out = c;
  return 0;
}

extern "C" int gpz_allreduce_sum_f64(void* comm, double* buf, int64_t n, void* stream) {
  GPZ_REQUIRE(comm && buf && n >= 1, "gpz_allreduce_sum_f64: bad arguments");
  if (int rc = need_rccl()) return rc;
  GPZ_RCCL_OK(g_rccl.AllReduce(buf, buf, (size_t)n, kRcclFloat64, kRcclSum, comm, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int gpz_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
  GPZ_REQUIRE(comm && buf && n >= 1, "gpz_allreduce_sum_f32: bad arguments");
  if (int rc = need_rccl()) return rc;
  GPZ_RCCL_OK(g_rccl.AllReduce(buf, buf, (size_t)n, kRcclFloat32, kRcclSum, comm, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int gpz_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  GPZ_REQUIRE(comm && send && recv && bytes_per_rank >= 1, "gpz_allgather: bad arguments");
  if (int rc = need_rccl()) return rc;
  GPZ_RCCL_OK(g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, kRcclUint8, comm, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int gpz_reduce_scatter_sum_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream) {
  GPZ_REQUIRE(comm && send && recv && n_per_rank >= 1, "gpz_reduce_scatter_sum_f32: bad arguments");
  if (int rc = need_rccl()) return rc;
  GPZ_RCCL_OK(g_rccl.ReduceScatter(send, recv, (size_t)n_per_rank, kRcclFloat32, kRcclSum, comm,
                                   static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int gpz_comm_destroy(void* comm) {
  if (!comm) return 0;
  if (int rc = need_rccl()) return rc;
  GPZ_RCCL_OK(g_rccl.CommDestroy(comm));
  return 0;
}
