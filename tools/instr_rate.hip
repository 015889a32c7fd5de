// tools/instr_rate.hip -- measured issue rates of the integer / fp64 instructions a big-integer
// multiply can be built from, on gfx950.  One kernel per instruction: NITER x 16 independent
// instructions per wave, 8 waves per SIMD, every CU busy; reports cycles per wave-instruction per
// SIMD (2 = full FP32 rate on the SIMD-32).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define NITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(name, decl, body, sink)                                                     \
  __global__ void __launch_bounds__(512) name(uint32_t* out, uint32_t seed) {              \
    decl;                                                                                  \
    for (int it = 0; it < NITER; it++) {                                                   \
      body body body body body body body body body body body body body body body body      \
    }                                                                                      \
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;                                     \
  }

#define INIT32 uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c0 = 1, c1 = 2, c2 = 3, c3 = 4
#define INIT64 uint32_t a = seed + threadIdx.x, b = seed * 3 + 1; uint64_t d0 = 1, d1 = 2, d2 = 3, d3 = 4

KERNEL(k_mad_u64_u32, INIT64,
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(a), "v"(b) : "vcc");
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d1) : "v"(a), "v"(b) : "vcc");
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d2) : "v"(a), "v"(b) : "vcc");
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d3) : "v"(a), "v"(b) : "vcc");,
  (uint32_t)(d0 + d1 + d2 + d3))
KERNEL(k_mul_lo_u32, INIT32,
  asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(c0) : "v"(a));
  asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(c1) : "v"(a));
  asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(c2) : "v"(a));
  asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(c3) : "v"(a));,
  c0 + c1 + c2 + c3)
KERNEL(k_mul_hi_u32, INIT32,
  asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(c0) : "v"(a));
  asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(c1) : "v"(a));
  asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(c2) : "v"(a));
  asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(c3) : "v"(a));,
  c0 + c1 + c2 + c3)
KERNEL(k_mad_u32_u24, INIT32,
  asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
  asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
  asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
  asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));,
  c0 + c1 + c2 + c3)
KERNEL(k_mul_hi_u32_u24, INIT32,
  asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(c0) : "v"(a));
  asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(c1) : "v"(a));
  asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(c2) : "v"(a));
  asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(c3) : "v"(a));,
  c0 + c1 + c2 + c3)
KERNEL(k_add_co_u32, INIT32,
  asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(c0) : "v"(a) : "vcc");
  asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(c1) : "v"(a) : "vcc");
  asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(c2) : "v"(a) : "vcc");
  asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(c3) : "v"(a) : "vcc");,
  c0 + c1 + c2 + c3)
KERNEL(k_add_u32, INIT32,
  asm volatile("v_add_u32 %0, %1, %0" : "+v"(c0) : "v"(a));
  asm volatile("v_add_u32 %0, %1, %0" : "+v"(c1) : "v"(a));
  asm volatile("v_add_u32 %0, %1, %0" : "+v"(c2) : "v"(a));
  asm volatile("v_add_u32 %0, %1, %0" : "+v"(c3) : "v"(a));,
  c0 + c1 + c2 + c3)
KERNEL(k_lshl_add_u64, INIT64,
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d0) : "v"(d1));
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d1) : "v"(d2));
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d2) : "v"(d3));
  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d3) : "v"(d0));,
  (uint32_t)(d0 + d1 + d2 + d3) + a + b)
KERNEL(k_mov_b32, INIT32,
  asm volatile("v_mov_b32 %0, %1" : "+v"(c0) : "v"(a));
  asm volatile("v_mov_b32 %0, %1" : "+v"(c1) : "v"(b));
  asm volatile("v_mov_b32 %0, %1" : "+v"(c2) : "v"(a));
  asm volatile("v_mov_b32 %0, %1" : "+v"(c3) : "v"(b));,
  c0 + c1 + c2 + c3)
KERNEL(k_add3_u32, INIT32,
  asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
  asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
  asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
  asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));,
  c0 + c1 + c2 + c3)

#define INITF64 double a = (double)(seed + threadIdx.x), b = 1.0000001, e0 = 1, e1 = 2, e2 = 3, e3 = 4
KERNEL(k_fma_f64, INITF64,
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(e0) : "v"(a), "v"(b));
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(e1) : "v"(a), "v"(b));
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(e2) : "v"(a), "v"(b));
  asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(e3) : "v"(a), "v"(b));,
  (uint32_t)(e0 + e1 + e2 + e3))
KERNEL(k_add_f64, INITF64,
  asm volatile("v_add_f64 %0, %1, %0" : "+v"(e0) : "v"(a));
  asm volatile("v_add_f64 %0, %1, %0" : "+v"(e1) : "v"(a));
  asm volatile("v_add_f64 %0, %1, %0" : "+v"(e2) : "v"(a));
  asm volatile("v_add_f64 %0, %1, %0" : "+v"(e3) : "v"(a));,
  (uint32_t)(e0 + e1 + e2 + e3) + (uint32_t)b)
#define INITF32 float a = (float)(seed + threadIdx.x), b = 1.0000001f, f0 = 1, f1 = 2, f2 = 3, f3 = 4
KERNEL(k_fma_f32, INITF32,
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f0) : "v"(a), "v"(b));
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f1) : "v"(a), "v"(b));
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f2) : "v"(a), "v"(b));
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f3) : "v"(a), "v"(b));,
  (uint32_t)(f0 + f1 + f2 + f3))

template <class K>
int run(const char* name, K kern, uint32_t* out, int blocks) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, 12345u);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, 12345u);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, a, b));
  // waves per SIMD = blocks*8 waves / (256 CUs * 4 SIMDs)
  double waves_per_simd = (double)blocks * 8 / 1024.0;
  double instr_per_simd = waves_per_simd * NITER * 16.0 * 4.0;
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-18s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz nominal)  %7.2f T lane-ops/s\n", name, ms, cycles / instr_per_simd,
         (double)blocks * 512 * NITER * 64.0 / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  uint32_t* out;
  int blocks = 256 * 4;  // 4 blocks x 8 waves per CU = 8 waves/SIMD
  CHK(hipMalloc(&out, (size_t)blocks * 512 * 4));
  run("v_fma_f32", k_fma_f32, out, blocks);
  run("v_add_u32", k_add_u32, out, blocks);
  run("v_mov_b32", k_mov_b32, out, blocks);
  run("v_add3_u32", k_add3_u32, out, blocks);
  run("v_add(c)_co_u32", k_add_co_u32, out, blocks);
  run("v_lshl_add_u64", k_lshl_add_u64, out, blocks);
  run("v_mad_u32_u24", k_mad_u32_u24, out, blocks);
  run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, out, blocks);
  run("v_mul_lo_u32", k_mul_lo_u32, out, blocks);
  run("v_mul_hi_u32", k_mul_hi_u32, out, blocks);
  run("v_mad_u64_u32", k_mad_u64_u32, out, blocks);
  run("v_fma_f64", k_fma_f64, out, blocks);
  run("v_add_f64", k_add_f64, out, blocks);
  return 0;
}
