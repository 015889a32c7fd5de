// tools/experiments/mul_probe.hip -- how fast do the library's lazy-limb Montgomery products (fq28.hpp: 14 x 28 bits, fs29.hpp: 9 x 29 bits) run as a
// function of waves per SIMD, and does splitting a column's MAC chain over two accumulators help?  Every lane runs a dependent chain of products
// (x <- x * y), as the group law does; reports ns per product per wave and the implied share of the v_mad_u64_u32 issue peak.
//   hipcc -O3 --offload-arch=gfx950 -I zkdl_amd/csrc tools/experiments/mul_probe.hip -o tools/experiments/mul_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "fq28.hpp"
#include "fs29.hpp"
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void mac(uint64_t& acc, uint32_t x, uint32_t y) { acc += (uint64_t)x * y; }
__device__ __forceinline__ void mac_c(uint64_t& acc, uint32_t x, uint32_t y) { acc += (uint64_t)x * y; }   // constants may go to SGPRs / literals
// two accumulators per column: even / odd partial products (the MACs as inline asm: the compiler re-associates the plain C form back into one chain)
__device__ __forceinline__ Fq fq_mul_split(const Fq& a, const Fq& b) {
  constexpr int N = 14;
  uint32_t m[N];
  Fq t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint64_t acc2 = 0;
#pragma unroll
    for (int i = 0; i <= k; i++) {
      if (i & 1) mac(acc2, a.v[i], b.v[k - i]);
      else mac(acc, a.v[i], b.v[k - i]);
    }
#pragma unroll
    for (int i = 0; i < k; i++) {
      if (i & 1) mac(acc, m[i], fq28::P(k - i));
      else mac(acc2, m[i], fq28::P(k - i));
    }
    asm("" : "+v"(acc2));   // keep the second chain a chain of its own
    acc += acc2;
    m[k] = ((uint32_t)acc * fq28::INV) & fq28::MASK;
    acc += (uint64_t)m[k] * fq28::P(0);
    acc >>= 28;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
    uint64_t acc2 = 0;
#pragma unroll
    for (int i = k - N + 1; i < N; i++) {
      if (i & 1) mac(acc2, a.v[i], b.v[k - i]);
      else mac(acc, a.v[i], b.v[k - i]);
    }
#pragma unroll
    for (int i = k - N + 1; i < N; i++) {
      if (i & 1) mac(acc, m[i], fq28::P(k - i));
      else mac(acc2, m[i], fq28::P(k - i));
    }
    asm("" : "+v"(acc2));   // keep the second chain a chain of its own
    acc += acc2;
    t.v[k - N] = (uint32_t)acc & fq28::MASK;
    acc >>= 28;
  }
  t.v[N - 1] = (uint32_t)acc;
  return t;
}

template <int VAR, int CHAINS>
__global__ void __launch_bounds__(256) k_fq(uint32_t* out, uint32_t seed, int iters) {
  Fq x[CHAINS], y;
#pragma unroll
  for (int i = 0; i < 14; i++) y.v[i] = (seed * (i + 3) + threadIdx.x) & fq28::MASK;
#pragma unroll
  for (int c = 0; c < CHAINS; c++)
#pragma unroll
    for (int i = 0; i < 14; i++) x[c].v[i] = (seed * (i + 7 + c) + threadIdx.x * 5) & fq28::MASK;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = VAR == 0 ? fq_mul_inl(x[c], y) : fq_mul_split(x[c], y);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++)
#pragma unroll
    for (int i = 0; i < 14; i++) s ^= x[c].v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_fs(uint32_t* out, uint32_t seed, int iters) {
  Fs x[CHAINS], y;
#pragma unroll
  for (int i = 0; i < 9; i++) y.v[i] = (seed * (i + 3) + threadIdx.x) & fs29::MASK;
#pragma unroll
  for (int c = 0; c < CHAINS; c++)
#pragma unroll
    for (int i = 0; i < 9; i++) x[c].v[i] = (seed * (i + 7 + c) + threadIdx.x * 5) & fs29::MASK;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = fs_mul(x[c], y);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++)
#pragma unroll
    for (int i = 0; i < 9; i++) s ^= x[c].v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// raw dependent / independent v_mad_u64_u32 chains
template <int CHAINS>
__global__ void __launch_bounds__(256) k_mad(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
  uint64_t d[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) d[c] = c + 1;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[c]) : "v"(a), "v"(b) : "vcc");
  }
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += d[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

template <class K>
static int run(const char* name, K kern, uint32_t* out, int waves_per_simd, int iters, double units_per_iter, double macs_per_unit) {
  // 256-thread blocks = 4 waves = one per SIMD of a CU; waves_per_simd blocks per CU
  const int blocks = 256 * waves_per_simd;
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, a, b));
  const double units_per_simd = (double)waves_per_simd * iters * units_per_iter;   // wave-level units executed by one SIMD
  const double ns_per_unit = ms * 1e6 / units_per_simd;
  const double tmacs = 1024.0 * 64.0 * units_per_simd * macs_per_unit / (ms * 1e-3) / 1e12;
  printf("%-28s waves/SIMD %d: %8.1f ns per wave-level unit per SIMD, %6.2f T lane-MAC/s (%.0f %% of 34.0)\n", name, waves_per_simd, ns_per_unit, tmacs, tmacs / 34.0 * 100);
  return 0;
}

int main() {
  uint32_t* out;
  CHK(hipMalloc(&out, 256 * 8 * 256 * 4));
  for (int w : {1, 2, 4, 8}) {
    run("v_mad_u64_u32 1 chain", k_mad<1>, out, w, 2000, 16, 1);
    run("v_mad_u64_u32 2 chains", k_mad<2>, out, w, 2000, 32, 1);
    run("v_mad_u64_u32 4 chains", k_mad<4>, out, w, 2000, 64, 1);
  }
  for (int w : {1, 2, 3, 4}) {
    run("fq_mul_inl, 1 chain", k_fq<0, 1>, out, w, 400, 1, 392);
    run("fq_mul_split, 1 chain", k_fq<1, 1>, out, w, 400, 1, 392);
    run("fq_mul_inl, 2 chains", k_fq<0, 2>, out, w, 400, 2, 392);
    run("fq_mul_split, 2 chains", k_fq<1, 2>, out, w, 400, 2, 392);
  }
  for (int w : {1, 2, 4, 8}) {
    run("fs_mul, 1 chain", k_fs<1>, out, w, 1000, 1, 153);
    run("fs_mul, 2 chains", k_fs<2>, out, w, 1000, 2, 153);
  }
  return 0;
}
