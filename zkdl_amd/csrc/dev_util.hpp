// zkdl_amd/csrc/dev_util.hpp -- wave64 / workgroup reductions over field elements (device only).
#pragma once
#include "field.hpp"
#include "g1.hpp"

namespace zk {

// wave64 shuffle ladder: after the call lane 0 holds the sum of all 64 lanes
__device__ __forceinline__ Fr wave_sum_fr(Fr x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Fr y;
#pragma unroll
    for (int i = 0; i < 8; i++) y.v[i] = __shfl_down(x.v[i], off, 64);
    x = fr_add(x, y);
  }
  return x;
}
// workgroup sum (blockDim.x a multiple of 64, at most 1024); `lds` holds blockDim.x/64 elements.
// Thread 0 returns the total.
__device__ __forceinline__ Fr block_sum_fr(Fr x, Fr* lds) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  x = wave_sum_fr(x);
  if (nw == 1) return x;
  __syncthreads();  // protect lds reuse across consecutive calls
  if (lane == 0) lds[wave] = x;
  __syncthreads();
  if (wave == 0) {
    x = lane < nw ? lds[lane] : fr_zero();
    x = wave_sum_fr(x);
  }
  return x;
}

}  // namespace zk
