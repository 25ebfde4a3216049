/* c_abi_demo.c -- libgpzoo_hip.so from plain C: no Python, no torch types, HIP runtime calls only for
 * memory.  Builds a small whitened SVGP problem (Matern-3/2, L latents) from a fixed linear-congruential
 * stream, evaluates the closed-form Gaussian ELBO with gpz_svgp_forward and prints it with the first
 * moments, one "key value" pair per line.  tests/test_hip_cabi.py regenerates the same inputs in numpy,
 * runs them through the Python mirror and compares.
 *
 *   gcc -std=c11 examples/c_abi_demo.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Lgpzoo_amd -lgpzoo_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gpzoo_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpzoo_hip.h"

static uint64_t lcg_state = 0x2545F4914F6CDD1DULL;
static double lcg(void) { /* uniform in [0,1): the test reproduces this stream bit for bit */
  lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(lcg_state >> 11) / 9007199254740992.0;
}

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_GPZ(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "gpz rc=%d: %s\n", r_, gpz_last_error()); return 3; } } while (0)

static void* to_device(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  const int64_t N = 1500, M = 200, L = 3;
  const int d = 2;
  double* X = malloc(sizeof(double) * N * d); double* Z = malloc(sizeof(double) * M * d);
  double* mu = malloc(sizeof(double) * L * M); double* Lu = malloc(sizeof(double) * L * M * M);
  double* y = malloc(sizeof(double) * L * N);
  const double sigma[3] = {0.8, 1.0, 1.2}, ell[3] = {2.0, 3.0, 4.0};
  for (int64_t i = 0; i < N * d; ++i) X[i] = 20.0 * lcg() - 10.0;
  for (int64_t i = 0; i < M * d; ++i) Z[i] = 20.0 * lcg() - 10.0;
  for (int64_t i = 0; i < L * M; ++i) mu[i] = lcg() - 0.5;
  for (int64_t i = 0; i < L * M * M; ++i) Lu[i] = 0.1 * (lcg() - 0.5);
  for (int64_t i = 0; i < L * N; ++i) y[i] = 2.0 * lcg() - 1.0;

  printf("version %d\n", gpz_version());
  gpz_svgp_problem p = {0};
  p.k.kind = GPZ_KERNEL_MATERN32; p.k.n_latent = (int32_t)L; p.k.dtype = GPZ_F64;
  p.k.sigma = to_device(sigma, sizeof sigma); p.k.lengthscale = to_device(ell, sizeof ell);
  p.dtype = GPZ_F64; p.whitened = 1; p.d = d; p.N = N; p.M = M;
  p.X = to_device(X, sizeof(double) * N * d); p.Z = to_device(Z, sizeof(double) * M * d);
  p.mu = to_device(mu, sizeof(double) * L * M); p.Lu_raw = to_device(Lu, sizeof(double) * L * M * M);
  p.y = to_device(y, sizeof(double) * L * N);
  p.jitter = 1e-2; p.var_clamp_min = 1e-6; p.noise_sd = 0.5;
  double *mean, *scale, *scal; int32_t* info;
  CHECK_HIP(hipMalloc((void**)&mean, sizeof(double) * L * N)); CHECK_HIP(hipMalloc((void**)&scale, sizeof(double) * L * N));
  CHECK_HIP(hipMalloc((void**)&scal, sizeof(double) * (2 * L + 1))); CHECK_HIP(hipMalloc((void**)&info, sizeof(int32_t) * L));
  p.mean = mean; p.scale = scale; p.kl = scal; p.loglik = scal + L; p.elbo = scal + 2 * L; p.info = info;
  if (!p.X || !p.Z || !p.mu || !p.Lu_raw || !p.y || !p.k.sigma || !p.k.lengthscale) { fprintf(stderr, "alloc failed\n"); return 2; }

  const size_t ws_bytes = gpz_svgp_workspace_bytes(&p, 512);
  if (ws_bytes == 0) { fprintf(stderr, "workspace query: %s\n", gpz_last_error()); return 3; }
  void* ws = NULL;
  CHECK_HIP(hipMalloc(&ws, ws_bytes));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_GPZ(gpz_svgp_forward(&p, 512, ws, ws_bytes, stream));     /* three ragged chunks of spots */
  CHECK_HIP(hipStreamSynchronize(stream));

  double h_scal[7], h_mean[4], h_scale[4]; int32_t h_info[3];
  CHECK_HIP(hipMemcpy(h_scal, scal, sizeof(double) * (2 * L + 1), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_mean, mean, sizeof h_mean, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_scale, scale, sizeof h_scale, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_info, info, sizeof h_info, hipMemcpyDeviceToHost));
  printf("elbo %.17g\n", h_scal[2 * L]);
  for (int l = 0; l < L; ++l) printf("kl%d %.17g\nloglik%d %.17g\ninfo%d %d\n", l, h_scal[l], l, h_scal[L + l], l, h_info[l]);
  for (int i = 0; i < 4; ++i) printf("mean%d %.17g\nscale%d %.17g\n", i, h_mean[i], i, h_scale[i]);

  /* error path: a bad argument is reported through the return code and gpz_last_error, not a crash */
  p.M = 0;
  printf("bad_rc %d\n", gpz_svgp_forward(&p, 512, ws, ws_bytes, stream));
  printf("bad_msg %s\n", gpz_last_error());
  return 0;
}
