// tools/experiments/lds_atomic_probe.hip -- LDS atomic-add rates on gfx950: ds_add_u64 against ds_add_u32 (random bins, no return value), the
// operation k_binb_words is made of (50 ds_add_u64 per 32-bit word).   hipcc --offload-arch=gfx950 -O3 lds_atomic_probe.hip -o lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <class T, int LANES_SAME>
__global__ void __launch_bounds__(1024) k_probe(uint32_t iters, uint32_t nbins, T* out) {
  extern __shared__ unsigned long long lds[];
  T* H = reinterpret_cast<T*>(lds);
  for (uint32_t i = threadIdx.x; i < nbins; i += blockDim.x) H[i] = 0;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (uint32_t it = 0; it < iters; it++) {
    x = x * 1664525u + 1013904223u;
    uint32_t b = (x >> 8) % nbins;
    if (LANES_SAME > 1) b = __shfl(b, (threadIdx.x & 63u) / LANES_SAME * LANES_SAME, 64);  // groups of lanes hit one bin (byte values repeat)
    atomicAdd(&H[b], (T)(x | 1u));
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = H[0] + H[nbins - 1];
}

template <class T, int S>
static void run(const char* name, uint32_t nbins) {
  const uint32_t grid = 256, iters = 4096;
  T* out;
  hipMalloc(&out, grid * sizeof(T));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const size_t lds = (size_t)nbins * sizeof(T);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<T, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_probe<T, S><<<grid, 1024, lds>>>(16, nbins, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_probe<T, S><<<grid, 1024, lds>>>(iters, nbins, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double lane_ops = (double)grid * 1024 * iters;
  printf("%-34s bins %6u: %8.3f ms  %7.1f G lane-atomics/s  (%5.2f clk per lane-atomic per CU at 2.1 GHz)\n", name, nbins, ms, lane_ops / ms / 1e6,
         ms * 1e-3 * 2.1e9 / ((double)1024 * iters));
  hipFree(out);
}

int main() {
  run<unsigned long long, 1>("ds_add_u64, every lane its own bin", 12800);
  run<unsigned int, 1>("ds_add_u32, every lane its own bin", 12800);
  run<unsigned int, 1>("ds_add_u32, every lane its own bin", 25600);
  run<unsigned long long, 4>("ds_add_u64, 4 lanes share a bin", 12800);
  run<unsigned int, 4>("ds_add_u32, 4 lanes share a bin", 12800);
  run<unsigned long long, 16>("ds_add_u64, 16 lanes share a bin", 12800);
  run<unsigned int, 16>("ds_add_u32, 16 lanes share a bin", 12800);
  return 0;
}
