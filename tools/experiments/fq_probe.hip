// Microbenchmark: 32-bit-limb Fp Montgomery product (field.hpp) vs a 28-bit-limb variant whose MACs need no carry add.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkdl_amd/csrc -I tools/experiments tools/experiments/fq_probe.hip -o /tmp/fq_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "field.hpp"
using namespace zk;

struct Fq { uint32_t v[14]; };
__device__ __constant__ uint32_t PQ[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};
static constexpr uint32_t INVQ = 0xffcfffdu, M28 = 0xfffffffu;

__device__ __forceinline__ Fq fq_mul(const Fq& a, const Fq& b) {
  uint32_t m[14];
  Fq t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 14; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * PQ[k - i];
    m[k] = ((uint32_t)acc * INVQ) & M28;
    acc += (uint64_t)m[k] * PQ[0];
    acc >>= 28;
  }
#pragma unroll
  for (int k = 14; k < 27; k++) {
#pragma unroll
    for (int i = k - 13; i < 14; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * PQ[k - i];
    t.v[k - 14] = (uint32_t)acc & M28;
    acc >>= 28;
  }
  t.v[13] = (uint32_t)acc;
  return t;
}

__global__ void __launch_bounds__(256) k_fp(const Fp* in, Fp* out, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  Fp a = in[i], b = in[i + 1];
  for (int it = 0; it < iters; it++) { a = fp_mul(a, b); b = fp_mul(b, a); }
  out[i] = fp_add(a, b);
}
__global__ void __launch_bounds__(256) k_fq(const Fq* in, Fq* out, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = in[i], b = in[i + 1];
  for (int it = 0; it < iters; it++) { a = fq_mul(a, b); b = fq_mul(b, a); }
#pragma unroll
  for (int l = 0; l < 14; l++) a.v[l] += b.v[l];
  out[i] = a;
}
int main() {
  const int blocks = 4096, threads = 256, iters = 200, n = blocks * threads + 1;
  Fp *ip, *op; Fq *iq, *oq;
  hipMalloc(&ip, n * sizeof(Fp)); hipMalloc(&op, n * sizeof(Fp)); hipMalloc(&iq, n * sizeof(Fq)); hipMalloc(&oq, n * sizeof(Fq));
  hipMemset(ip, 0x11, n * sizeof(Fp)); hipMemset(iq, 0x05, n * sizeof(Fq));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    float ms;
    hipEventRecord(e0); hipLaunchKernelGGL(k_fp, dim3(blocks), dim3(threads), 0, 0, ip, op, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("fp32-limb: %.3f ms  -> %.1f G products/s\n", ms, 2.0 * iters * blocks * threads / ms / 1e6);
    hipEventRecord(e0); hipLaunchKernelGGL(k_fq, dim3(blocks), dim3(threads), 0, 0, iq, oq, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("fq28-limb: %.3f ms  -> %.1f G products/s\n", ms, 2.0 * iters * blocks * threads / ms / 1e6);
  }
  return 0;
}
