// Microbenchmark: what would fp32 emulated by split-bf16 operands sustain on gfx950?
// One emulated k16 step of a 64x64 wave tile = 6 products (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) per 32x32 tile
// = 24 v_mfma_f32_32x32x16_bf16, fed by 12 16-byte fragment reads (3 pieces each of 2 A and 2 B sub-tiles).
// V=0: operands in registers.  V=1: + LDS fragment reads.  V=2: V1 + one block barrier per two k16 steps.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe_bf16.hip -o /tmp/probe_bf16 && /tmp/probe_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2][6][128 * 24];   // [buf][piece of A or B][128 rows x (16 + 8 pad)]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;                               // 2 x 4 waves of 64 x 32
  for (int i = tid; i < 2 * 6 * 128 * 24; i += 512) (&smem[0][0][0])[i] = (__bf16)(0.01f * ((i * 7 + 3) % 11));
  __syncthreads();
  f32x16 acc[2];
  for (int a = 0; a < 2; ++a) for (int g = 0; g < 16; ++g) acc[a][g] = 0;
  const int r32 = lane & 31, h = lane >> 5;
  bf16x8 fa[2][3], fb[3];
  for (int p = 0; p < 3; ++p) { for (int e = 0; e < 8; ++e) { fa[0][p][e] = (__bf16)(1.f + p); fa[1][p][e] = (__bf16)(0.5f + p); fb[p][e] = (__bf16)(0.25f * p + 1.f); } }
  for (int t = 0; t < iters; ++t) {
    const int buf = t & 1;
    if (V >= 1) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          fa[mi][p] = *reinterpret_cast<const bf16x8*>(&smem[buf][p][(wm * 64 + mi * 32 + r32) * 24 + h * 8]);
        fb[p] = *reinterpret_cast<const bf16x8*>(&smem[buf][3 + p][(wn * 32 + r32) * 24 + h * 8]);
      }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[0], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[1], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[0], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[2], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[1], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][2], fb[0], acc[mi], 0, 0, 0);
    }
    if (V >= 2 && (t & 1)) __syncthreads();
  }
  float s = 0;
  for (int a = 0; a < 2; ++a) for (int g = 0; g < 16; ++g) s += acc[a][g];
  out[blockIdx.x * 512 + tid] = s;
}

template <int V>
static void run(const char* name) {
  float* out; hipMalloc(&out, sizeof(float) * 512 * 512);
  const int iters = 20000, blocks = 512;                 // 2 blocks of 8 waves per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(512), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_flops = (double)blocks * 8 * iters * 12 * 32.0 * 32 * 16 * 2;
  printf("%-28s %8.2f ms  bf16 MFMA %7.1f TF  -> fp32-equivalent (6 products) %6.1f TF\n", name, ms, mfma_flops / ms / 1e9,
         mfma_flops / 6 / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0>("registers only");
  run<1>("+ LDS fragment reads");
  run<2>("+ barrier per 32 k");
  return 0;
}
