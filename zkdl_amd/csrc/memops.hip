// zkdl_amd/csrc/memops.hip -- device copies / fills as ordinary kernels on the context's stream.
//
// A block released to the context's allocator is handed out again immediately; that is safe only if
// everything that touches device memory is ordered on the context's stream the way kernels are.  So the
// library issues no copy-engine work: device-to-device copies, fills and the small pinned-host ->
// device uploads (challenge vectors, MSM job tables) are kernels, and the host-facing transfers
// (zk_memcpy_h2d / d2h) are stream-ordered copies on the context's stream followed by a stream wait.
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
// blocking host <-> device transfers, ordered on the context's stream like everything else (never on the legacy
// NULL stream: the CU-masked G1 streams are "blocking" streams in HIP's sense -- hipExtStreamCreateWithCUMask has
// no flags argument -- so NULL-stream work would serialise with a sealed proof's G1 phase)
int host_to_dev(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return ZK_OK;
  ctx->join_lanes();
  ZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZK_OK;
}
int dev_to_host(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return ZK_OK;
  ctx->join_lanes();
  ZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZK_OK;
}

}  // namespace zk
