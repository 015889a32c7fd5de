// zkdl_amd/csrc/memops.hip -- device copies / fills as ordinary kernels on the context's stream.
//
// A block released to the context's allocator is handed out again immediately; that is safe only if
// everything that touches device memory is ordered on the context's stream the way kernels are.  So the
// library issues no copy-engine work: device-to-device copies, fills and the small pinned-host ->
// device uploads (challenge vectors, MSM job tables) are kernels, and the host-facing transfers
// (zk_memcpy_h2d / d2h) drain the stream before a synchronous copy.
#include "ctx.hpp"

namespace zk {

__device__ __forceinline__ void k_copy16(uint32_t zk_nblk, const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)zk_nblk * blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ void k_copy4(uint32_t zk_nblk, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)zk_nblk * blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ void k_zero4(uint32_t zk_nblk, uint32_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)zk_nblk * blockDim.x) dst[i] = 0u;
}

// src may be device memory or pinned (device-visible) host memory; sizes are multiples of 4 bytes
int dev_copy(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return ZK_OK;
  if (bytes & 3u) return ctx->fail(ZK_ERR_ARG, "dev_copy: size must be a multiple of 4 bytes");
  if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15u) == 0) {
    size_t n = bytes / 16;
    ZK_LAUNCHB(ctx, "copy", 2ull * bytes, 256, k_copy16, dim3(stream_grid(n)), 0, (const uint4*)src, (uint4*)dst, n);
  } else {
    size_t n = bytes / 4;
    ZK_LAUNCHB(ctx, "copy", 2ull * bytes, 256, k_copy4, dim3(stream_grid(n)), 0, (const uint32_t*)src, (uint32_t*)dst, n);
  }
  return ZK_OK;
}
int dev_zero(zk_ctx* ctx, void* dst, size_t bytes) {
  if (bytes == 0) return ZK_OK;
  if (bytes & 3u) return ctx->fail(ZK_ERR_ARG, "dev_zero: size must be a multiple of 4 bytes");
  size_t n = bytes / 4;
  ZK_LAUNCHB(ctx, "copy", bytes, 256, k_zero4, dim3(stream_grid(n)), 0, (uint32_t*)dst, n);
  return ZK_OK;
}
// blocking host <-> device transfers: drain the stream first, then a synchronous copy
int host_to_dev(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return ZK_OK;
  ctx->join_lanes();
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ZK_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return ZK_OK;
}
int dev_to_host(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return ZK_OK;
  ctx->join_lanes();
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ZK_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return ZK_OK;
}

}  // namespace zk
