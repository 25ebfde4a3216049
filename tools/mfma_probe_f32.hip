// Microbenchmark: where does the fp32 16x16x4 MFMA loop lose cycles?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V=0: pure MFMA, operands in registers. V=1: + LDS fragment reads per k-tile. V=2: V1 + barrier per k-tile
template <int V, int SHAPE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (128 * 20 + 16 * 132)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 15, q = lane >> 4;
  for (int i = tid; i < 2 * (128 * 20 + 16 * 132); i += 256) smem[i] = (float)((i * 7 + 3) % 11) * 0.01f;
  __syncthreads();
  if (SHAPE == 16) {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    f32x4 fa[4]; float fb[4][4];
    for (int a = 0; a < 4; ++a) { fa[a] = f32x4{1.f + lane, 2.f, 3.f, 4.f}; for (int j = 0; j < 4; ++j) fb[j][a] = 0.5f + a + j; }
    for (int t = 0; t < iters; ++t) {
      const float* sA = smem + (t & 1) * (128 * 20 + 16 * 132);
      const float* sB = sA + 128 * 20;
      if (V >= 1) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = *reinterpret_cast<const f32x4*>(sA + (wm * 64 + mi * 16 + r) * 20 + q * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (V >= 1) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) fb[j][ni] = sB[(q * 4 + j) * 132 + wn * 64 + ni * 16 + r];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi][j], fb[j][ni], acc[mi][ni], 0, 0, 0);
      }
      if (V >= 2) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int g = 0; g < 4; ++g) s += acc[a][b][g];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    // 32x32x2: wave tile 64x64 = 2x2 tiles, A frag: lane l -> row l&31, k = l>>5
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) acc[a][b][g] = 0;
    const int r32 = lane & 31, h = lane >> 5;
    f32x4 fa[2][2]; float fb[8][2];
    for (int a = 0; a < 2; ++a) { fa[a][0] = f32x4{1.f + lane, 2.f, 3.f, 4.f}; fa[a][1] = fa[a][0]; }
    for (int j = 0; j < 8; ++j) { fb[j][0] = j; fb[j][1] = j + 1; }
    for (int t = 0; t < iters; ++t) {
      const float* sA = smem + (t & 1) * (128 * 20 + 16 * 132);
      const float* sB = sA + 128 * 20;
      if (V >= 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            fa[mi][kk] = *reinterpret_cast<const f32x4*>(sA + (wm * 64 + mi * 32 + r32) * 20 + kk * 8 + h * 4);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (V >= 1) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[kk * 4 + j][ni] = sB[(kk * 8 + h * 4 + j) * 132 + wn * 64 + ni * 32 + r32];
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][kk][j], fb[kk * 4 + j][ni], acc[mi][ni], 0, 0, 0);
        }
      if (V >= 2) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int g = 0; g < 16; ++g) s += acc[a][b][g];
    out[blockIdx.x * 256 + tid] = s;
  }
}

template <int V, int SHAPE>
void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<V, SHAPE>), dim3(grid), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<V, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  // flops per wave per iter: 64x64x16x2
  double fl = (double)grid * 4 * iters * 64.0 * 64 * 16 * 2;
  printf("%-28s shape=%d blocks/CU=%d  %.3f ms  %.1f TF\n", name, SHAPE, blocks_per_cu, best, fl / best / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int b : {1, 2, 3, 4}) {
    run<0, 16>("pure mfma", b, out);
    run<1, 16>("mfma + lds frag reads", b, out);
    run<2, 16>("mfma + lds + barrier", b, out);
    run<0, 32>("pure mfma", b, out);
    run<1, 32>("mfma + lds frag reads", b, out);
    run<2, 32>("mfma + lds + barrier", b, out);
  }
  return 0;
}
