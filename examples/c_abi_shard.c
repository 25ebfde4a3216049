/* c_abi_shard.c -- the multi-GPU exchange of the sharded evaluation from plain C (no Python, no torch): one rank per
 * process and GPU; rank 0 draws the RCCL id (gpz_comm_unique_id) and hands it to the others through a file; every rank
 * evaluates ITS block of latents with gpz_svgp_forward (latents shard without any data-path collective)
 * and the partial ELBOs are summed with gpz_allreduce_sum_f64 -- one fp64 scalar over RCCL / xGMI.
 *
 *   c_abi_shard <world> <rank> <id-file>          e.g. on an 8-GPU node:  for r in 0..7: c_abi_shard 8 $r /tmp/gpz.id &
 *
 * Run as `c_abi_shard 1 0 /tmp/gpz.id` it is a one-rank job (what tests/test_hip_cabi.py executes on the one-GPU box):
 * the communicator is created, the all-reduce runs on the device scalar, and the sum equals the local ELBO.
 *
 *   gcc -std=c11 examples/c_abi_shard.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Lgpzoo_amd -lgpzoo_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gpzoo_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_shard
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "gpzoo_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_GPZ(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "gpz rc=%d: %s\n", r_, gpz_last_error()); return 3; } } while (0)

static uint64_t lcg_state;
static double lcg(void) {
  lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(lcg_state >> 11) / 9007199254740992.0;
}
static void* to_device(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s world rank id-file\n", argv[0]); return 1; }
  const int world = atoi(argv[1]), rank = atoi(argv[2]);
  const char* id_file = argv[3];
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  CHECK_HIP(hipSetDevice(rank % ndev));

  /* the 128-byte communicator id: rank 0 writes it, the others wait for the file */
  char id[128];
  if (rank == 0) {
    CHECK_GPZ(gpz_comm_unique_id(id));
    char tmp[512];
    snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 4;
    fclose(f);
    rename(tmp, id_file);
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); ++tries) usleep(100000);
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 4;
    fclose(f);
  }
  void* comm = NULL;
  CHECK_GPZ(gpz_comm_init(&comm, world, rank, id));

  /* the model: Ltot latents, this rank owns [l0, l0 + L); X, Z replicated (same stream on every rank) */
  const int64_t N = 1200, M = 160, d = 2;
  const int Ltot = 2 * world, L = 2, l0 = 2 * rank;
  lcg_state = 0x9E3779B97F4A7C15ULL;
  double* X = malloc(sizeof(double) * N * d); double* Z = malloc(sizeof(double) * M * d);
  for (int64_t i = 0; i < N * d; ++i) X[i] = 20.0 * lcg() - 10.0;
  for (int64_t i = 0; i < M * d; ++i) Z[i] = 20.0 * lcg() - 10.0;
  double* mu = malloc(sizeof(double) * L * M); double* Lu = malloc(sizeof(double) * L * M * M);
  double* y = malloc(sizeof(double) * L * N); double sigma[2], ell[2];
  for (int l = 0; l < Ltot; ++l) {            /* per-latent arrays are drawn for every latent; a rank keeps only its own */
    const int mine = l >= l0 && l < l0 + L, j = l - l0;
    for (int64_t i = 0; i < M; ++i) { const double v = lcg() - 0.5; if (mine) mu[j * M + i] = v; }
    for (int64_t i = 0; i < M * M; ++i) { const double v = 0.1 * (lcg() - 0.5); if (mine) Lu[j * M * M + i] = v; }
    for (int64_t i = 0; i < N; ++i) { const double v = 2.0 * lcg() - 1.0; if (mine) y[j * N + i] = v; }
    if (mine) { sigma[j] = 0.8 + 0.1 * l; ell[j] = 2.0 + 0.5 * l; }
  }
  gpz_svgp_problem p;
  memset(&p, 0, sizeof p);
  p.k.kind = GPZ_KERNEL_RBF; p.k.n_latent = L; p.k.dtype = GPZ_F64;
  p.k.sigma = to_device(sigma, sizeof sigma); p.k.lengthscale = to_device(ell, sizeof ell);
  p.dtype = GPZ_F64; p.whitened = 1; p.d = (int32_t)d; p.N = N; p.M = M;
  p.X = to_device(X, sizeof(double) * N * d); p.Z = to_device(Z, sizeof(double) * M * d);
  p.mu = to_device(mu, sizeof(double) * L * M); p.Lu_raw = to_device(Lu, sizeof(double) * L * M * M);
  p.jitter = 1e-2; p.var_clamp_min = 1e-6; p.y = to_device(y, sizeof(double) * L * N); p.noise_sd = 0.5;
  double* scal = NULL; int32_t* info = NULL;
  CHECK_HIP(hipMalloc((void**)&scal, sizeof(double) * (2 * L + 1)));
  CHECK_HIP(hipMalloc((void**)&info, sizeof(int32_t) * L));
  p.kl = scal; p.loglik = scal + L; p.elbo = scal + 2 * L; p.info = info;
  const size_t wsb = gpz_svgp_workspace_bytes(&p, 0);
  void* ws = NULL;
  CHECK_HIP(hipMalloc(&ws, wsb));
  hipStream_t s;
  CHECK_HIP(hipStreamCreate(&s));
  CHECK_GPZ(gpz_svgp_forward(&p, 0, ws, wsb, s));
  double local = 0.0;
  CHECK_HIP(hipMemcpyAsync(&local, p.elbo, sizeof local, hipMemcpyDeviceToHost, s));
  CHECK_GPZ(gpz_allreduce_sum_f64(comm, p.elbo, 1, s));           /* the one exchange: sum of the partial ELBOs */
  double total = 0.0;
  CHECK_HIP(hipMemcpyAsync(&total, p.elbo, sizeof total, hipMemcpyDeviceToHost, s));
  CHECK_HIP(hipStreamSynchronize(s));
  printf("rank %d\nworld %d\nlocal_elbo %.17g\ntotal_elbo %.17g\n", rank, world, local, total);
  CHECK_GPZ(gpz_comm_destroy(comm));
  return 0;
}
