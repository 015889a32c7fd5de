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

// three workgroup sums at once (the coefficient triple of a sumcheck round): the three shuffle ladders interleave,
// and there is ONE LDS stage and two barriers instead of three of each.  `lds` holds 3 * blockDim.x/64 elements.
// Thread 0 returns the totals in x0, x1, x2.
__device__ __forceinline__ void block_sum3_fr(Fr& x0, Fr& x1, Fr& x2, Fr* lds) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Fr y0, y1, y2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      y0.v[i] = __shfl_down(x0.v[i], off, 64);
      y1.v[i] = __shfl_down(x1.v[i], off, 64);
      y2.v[i] = __shfl_down(x2.v[i], off, 64);
    }
    x0 = fr_add(x0, y0);
    x1 = fr_add(x1, y1);
    x2 = fr_add(x2, y2);
  }
  if (nw == 1) return;
  __syncthreads();  // protect lds reuse across consecutive calls
  if (lane == 0) {
    lds[3 * wave + 0] = x0;
    lds[3 * wave + 1] = x1;
    lds[3 * wave + 2] = x2;
  }
  __syncthreads();
  if (wave == 0) {
    x0 = lane < nw ? lds[3 * lane + 0] : fr_zero();
    x1 = lane < nw ? lds[3 * lane + 1] : fr_zero();
    x2 = lane < nw ? lds[3 * lane + 2] : fr_zero();
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {  // nw <= 16
      Fr y0, y1, y2;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        y0.v[i] = __shfl_down(x0.v[i], off, 64);
        y1.v[i] = __shfl_down(x1.v[i], off, 64);
        y2.v[i] = __shfl_down(x2.v[i], off, 64);
      }
      x0 = fr_add(x0, y0);
      x1 = fr_add(x1, y1);
      x2 = fr_add(x2, y2);
    }
  }
}

}  // namespace zk
