// What does a v_mad_u64_u32 cost on gfx950 at full occupancy: dependent chain, two interleaved chains, MAD + carry add.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/mad_probe.hip -o /tmp/mad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define MAC1(lo, x, y) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y) : "vcc")
#define MAC2(lo, hi, x, y) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc")
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t x, uint32_t y, int iters) {
  uint64_t a = threadIdx.x, b = blockIdx.x, c = 3, d = 4;
  uint32_t h = 0, h2 = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 32; u++) {
      if (MODE == 0) { MAC1(a, x, y); MAC1(a, y, x); }                    // one dependent chain: 2 MADs
      if (MODE == 1) { MAC1(a, x, y); MAC1(b, y, x); }                    // two independent chains: 2 MADs
      if (MODE == 2) { MAC2(a, h, x, y); MAC2(a, h, y, x); }              // one chain, MAD + addc: 2 MADs
      if (MODE == 3) { MAC2(a, h, x, y); MAC2(b, h2, y, x); }             // two chains, MAD + addc
      if (MODE == 4) { MAC1(a, x, y); MAC1(b, y, x); MAC1(c, x, x); MAC1(d, y, y); }  // four chains: 4 MADs
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a ^ b ^ c ^ d) ^ h ^ h2;
}
template <int MODE>
static void run(const char* name, int mads_per_u) {
  const int blocks = 2048, threads = 256, iters = 200;
  uint32_t* out; hipMalloc(&out, blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 67890u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  double mads = (double)blocks * threads * iters * 32 * mads_per_u;
  double waves_mads = mads / 64;
  double cyc = ms * 1e-3 * 2.4e9 * 1024 / waves_mads;   // SIMD-cycles per wave-level MAD, assuming 1024 SIMDs at 2.4 GHz
  printf("%-34s %.3f ms  %.2f T lane-MAD/s  %.2f cycles per wave-MAD\n", name, ms, mads / ms / 1e9, cyc);
  hipFree(out);
}
int main() {
  run<0>("1 chain, MAD only", 2);
  run<1>("2 chains, MAD only", 2);
  run<4>("4 chains, MAD only", 4);
  run<2>("1 chain, MAD + carry add", 2);
  run<3>("2 chains, MAD + carry add", 2);
  return 0;
}
