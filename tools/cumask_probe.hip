// cumask_probe.hip -- does a CU-masked stream (hipExtStreamCreateWithCUMask) keep compute units free on this box, how do
// mask bits map to (XCC, SE, CU), and does a 134 KB-LDS workgroup of a second stream start while a two-workgroups-per-CU
// kernel (74 KB LDS each, the trailing SYRK's footprint) fills the rest?   hipcc --offload-arch=gfx950 -O3 ... && ./a.out
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Rec { uint32_t hw, xcc; uint64_t t0, t1; };

__device__ __forceinline__ uint32_t hw_id() { return __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }

// spins `ticks` of the 100 MHz wall clock; dynamic LDS only sets the footprint
__global__ void spin_kernel(Rec* out, int ticks) {
  extern __shared__ char lds[];
  const uint64_t t0 = wall_clock64();
  if (threadIdx.x == 0) lds[0] = 1;
  while ((int64_t)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    Rec r; r.hw = hw_id(); r.xcc = xcc_id(); r.t0 = t0; r.t1 = wall_clock64();
    out[blockIdx.x] = r;
  }
}

static uint32_t cu_key(const Rec& r) {   // (xcc, se, sh, cu)
  const uint32_t cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7;
  return (r.xcc << 12) | (se << 8) | (sh << 4) | cu;
}

int main() {
  hipDeviceProp_t prop;
  OK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s  CUs %d\n", prop.name, ncu);
  OK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  const int words = (ncu + 31) / 32;
  Rec* d;
  const int NWG = 8192;
  OK(hipMalloc(&d, sizeof(Rec) * NWG));
  std::vector<Rec> h(NWG);

  auto distinct = [&](hipStream_t s, const char* what) {
    OK(hipMemset(d, 0, sizeof(Rec) * NWG));
    hipLaunchKernelGGL(spin_kernel, dim3(4096), dim3(512), 74 * 1024, s, d, 500);
    OK(hipStreamSynchronize(s));
    OK(hipMemcpy(h.data(), d, sizeof(Rec) * 4096, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus;
    int per_xcc[16] = {};
    for (int i = 0; i < 4096; ++i) cus.insert(cu_key(h[i]));
    for (uint32_t k : cus) per_xcc[(k >> 12) & 15]++;
    printf("%-40s distinct CUs %3zu  per XCC:", what, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    return cus;
  };

  hipStream_t plain;
  OK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  auto all = distinct(plain, "unmasked stream");

  // masks: first 224 bits; all but every 8th bit; all but the last 4 bits of each 32
  struct M { const char* name; std::vector<uint32_t> w; };
  std::vector<M> masks;
  { M m{"mask: bits [0,224)", std::vector<uint32_t>(words, 0)}; for (int b = 0; b < 224 && b < ncu; ++b) m.w[b / 32] |= 1u << (b % 32); masks.push_back(m); }
  { M m{"mask: all but bits [0,32)", std::vector<uint32_t>(words, 0xffffffffu)}; m.w[0] = 0; masks.push_back(m); }
  { M m{"mask: all but bits 32k..32k+3", std::vector<uint32_t>(words, 0xfffffff0u)}; masks.push_back(m); }
  std::vector<hipStream_t> ms;
  for (auto& m : masks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.w.size(), m.w.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", m.name, hipGetErrorString(e)); ms.push_back(nullptr); continue; }
    ms.push_back(s);
    auto got = distinct(s, m.name);
    std::set<uint32_t> miss;
    for (uint32_t k : all) if (!got.count(k)) miss.insert(k);
    printf("    unused CUs (xcc.se.sh.cu):");
    int n = 0;
    for (uint32_t k : miss) { if (n++ < 40) printf(" %u.%u.%u.%u", (k >> 12) & 15, (k >> 8) & 7, (k >> 4) & 1, k & 15); }
    printf("\n");
  }

  // overlap: hog on stream H (two 74 KB workgroups per CU, 8192 x 30 us), then a 134 KB-LDS kernel of 32 workgroups on
  // the plain stream 100 us later; when do the 32 start relative to the hog's first start / last end?
  auto overlap = [&](hipStream_t hog, const char* what) {
    Rec* d2;
    OK(hipMalloc(&d2, sizeof(Rec) * 64));
    OK(hipMemset(d, 0, sizeof(Rec) * NWG));
    OK(hipDeviceSynchronize());
    hipLaunchKernelGGL(spin_kernel, dim3(NWG), dim3(512), 74 * 1024, hog, d, 3000);           // 30 us each
    hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 134 * 1024, plain, d2, 5000);        // 50 us each
    OK(hipDeviceSynchronize());
    std::vector<Rec> hb(32);
    OK(hipMemcpy(h.data(), d, sizeof(Rec) * NWG, hipMemcpyDeviceToHost));
    OK(hipMemcpy(hb.data(), d2, sizeof(Rec) * 32, hipMemcpyDeviceToHost));
    uint64_t h0 = ~0ull, h1 = 0, b0 = ~0ull, b1 = 0, blast = 0;
    for (int i = 0; i < NWG; ++i) { if (h[i].t0 < h0) h0 = h[i].t0; if (h[i].t1 > h1) h1 = h[i].t1; }
    for (int i = 0; i < 32; ++i) { if (hb[i].t0 < b0) b0 = hb[i].t0; if (hb[i].t0 > blast) blast = hb[i].t0; if (hb[i].t1 > b1) b1 = hb[i].t1; }
    printf("%-40s hog %.0f us; big-LDS kernel: first start +%.0f us, last start +%.0f us, end +%.0f us (after hog start)\n", what,
           (h1 - h0) / 100.0, ((double)b0 - (double)h0) / 100.0, ((double)blast - (double)h0) / 100.0, ((double)b1 - (double)h0) / 100.0);
    OK(hipFree(d2));
  };
  overlap(plain, "hog on the SAME plain stream (serial)");
  hipStream_t plain2;
  OK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
  overlap(plain2, "hog on a second unmasked stream");
  hipStream_t lo, hi;
  int pl, ph;
  OK(hipDeviceGetStreamPriorityRange(&pl, &ph));
  OK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, pl));
  overlap(lo, "hog on a low-priority stream");
  for (size_t i = 0; i < ms.size(); ++i)
    if (ms[i]) overlap(ms[i], masks[i].name);
  return 0;
}
