// zkdl_amd/csrc/dist_rccl.hip -- the one collective of the multi-GPU path, natively: an RCCL all-reduce (SUM, uint64) of the per-rank
// run aggregate over xGMI, so that a C++ host (zkdl_amd/host/demo --gpus N) needs no torch.distributed.
//
// north_star: "independent proof batches shard across the 8 GPUs of one node with an RCCL all-reduce over xGMI only for the final
// sumcheck aggregate".  One process per GPU, every rank proves its own batches (no data-path collective); each proof leaves its
// digest as 8 zero-extended 64-bit lanes (zk_transcript_seal / zk_transcript_digest_dev), a rank sums the lanes of its proofs on the
// device, and ONE ncclAllReduce(ncclUint64, ncclSum) of those 8 lanes closes the run -- exact for < 2^32 ranks x proofs; every rank
// then carry-propagates and reduces mod r (zk_dist_lanes_to_fr).  The reference has no multi-GPU code at all (README.md:135).
//
// librccl is opened with dlopen at zk_dist_init, not linked: a single-GPU user of libzkdl_amd.so never loads it.
// Rendezvous: rank 0 calls zk_dist_unique_id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any means the launcher
// has (a file, an environment variable, MPI, a torch store ...); zk_dist_init is ncclCommInitRank on the context's device.
#include <dlfcn.h>
#include <string.h>

#include "layers.hpp"

namespace {
// the few declarations of <rccl/rccl.h> this file needs (kept local so that the build does not depend on the header's location)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclUint64 = 5, ncclSum = 0 };
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // ROCm's own librccl (soname librccl.so.1), the one built against the HIP runtime this library is linked with.  A process may carry another
    // copy (PyTorch wheels bundle a librccl.so bound to their own bundled HIP runtime): its streams and ours are not interchangeable, so it is
    // deliberately NOT picked up through the global namespace.
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (r.so) {
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
      r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.so, "ncclAllReduce"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
      r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.so, "ncclCommCount"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
      if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
        dlclose(r.so);
        r.so = nullptr;
      }
    }
  }
  return r.so ? &r : nullptr;
}
}  // namespace

struct zk_dist {
  zk_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int zk_dist_unique_id(uint8_t* id128) {
  if (!id128) return ZK_ERR_ARG;
  Rccl* r = rccl();
  if (!r) return ZK_ERR_DEVICE;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != 0) return ZK_ERR_DEVICE;
  memcpy(id128, id.internal, 128);
  return ZK_OK;
}

int zk_dist_init(zk_ctx* ctx, int rank, int world, const uint8_t* id128, zk_dist** out) {
  if (!ctx) return ZK_ERR_ARG;
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return ctx->fail(ZK_ERR_ARG, "zk_dist_init: bad argument");
  *out = nullptr;
  Rccl* r = rccl();
  if (!r) return ctx->fail(ZK_ERR_DEVICE, "zk_dist_init: librccl could not be loaded");
  (void)hipSetDevice(ctx->device);
  zk_dist* d = new zk_dist();
  d->ctx = ctx;
  d->rank = rank;
  d->world = world;
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  const ncclResult_t rc = r->CommInitRank(&d->comm, world, id, rank);
  if (rc != 0) {
    delete d;
    return ctx->fail(ZK_ERR_DEVICE, std::string("ncclCommInitRank failed: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?"));
  }
  *out = d;
  return ZK_OK;
}

int zk_dist_world_size(const zk_dist* d) {
  if (!d) return 0;
  int n = d->world;
  Rccl* r = rccl();
  if (r && r->CommCount) (void)r->CommCount(d->comm, &n);  // as RCCL reports it
  return n;
}

// In-place SUM all-reduce of n 64-bit lanes (device memory), enqueued on the context's main stream behind everything the context has
// submitted so far -- including sealed proofs' G1 phases, whose digests the lanes usually are; returns after the stream has drained.
int zk_dist_allreduce_u64(zk_dist* d, uint64_t* lanes_d, size_t n) {
  if (!d || !d->ctx) return ZK_ERR_ARG;
  zk_ctx* ctx = d->ctx;
  if (!lanes_d || (((uintptr_t)lanes_d) & 7u)) return ctx->fail(ZK_ERR_ARG, "zk_dist_allreduce_u64: null or misaligned pointer");
  (void)hipSetDevice(ctx->device);
  ZK_TRY(zk_ctx_sync(ctx));  // settles recorded work, flushes pending G1 batches, joins every stream into the main stream
  Rccl* r = rccl();
  if (!r) return ctx->fail(ZK_ERR_DEVICE, "librccl could not be loaded");
  const ncclResult_t rc = r->AllReduce(lanes_d, lanes_d, n, ncclUint64, ncclSum, d->comm, ctx->main_stream);
  if (rc != 0) return ctx->fail(ZK_ERR_DEVICE, std::string("ncclAllReduce failed: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?"));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->main_stream));
  return ZK_OK;
}

// the device-side sum of `rows` digests (8 lanes each, one row per proof, as zk_transcript_seal leaves them) into out8_d[8]
}  // extern "C"

namespace zk {
__global__ void __launch_bounds__(64) k_sum_rows(const uint64_t* __restrict__ rows, size_t n, uint64_t* __restrict__ out8) {
  const uint32_t lane = threadIdx.x & 7u, part = threadIdx.x >> 3;  // 8 partial sums per lane
  uint64_t s = 0;
  for (size_t i = part; i < n; i += 8) s += rows[i * 8 + lane];
  __shared__ uint64_t sm[64];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 8) {
    uint64_t t = 0;
    for (int p = 0; p < 8; p++) t += sm[p * 8 + threadIdx.x];
    out8[threadIdx.x] = t;
  }
}
}  // namespace zk

extern "C" {

int zk_dist_sum_rows(zk_ctx* ctx, const uint64_t* rows_d, size_t n, uint64_t* out8_d) {
  if (!ctx) return ZK_ERR_ARG;
  if (!rows_d || !out8_d) return ctx->fail(ZK_ERR_ARG, "zk_dist_sum_rows: null pointer");
  (void)hipSetDevice(ctx->device);
  ZK_TRY(zk_ctx_sync(ctx));  // every sealed proof's digest row is final
  ZK_LAUNCH(ctx, "digest_row_sum", 64ull * n, zk::k_sum_rows, dim3(1), dim3(64), 0, rows_d, n, out8_d);
  return ZK_OK;
}

// 8 summed 64-bit lanes (host) -> the field element sum_j lane_j 2^(32 j) mod r, as raw little-endian limbs
int zk_dist_lanes_to_fr(const uint64_t* lanes8_h, zk_fr* out_h) {
  if (!lanes8_h || !out_h) return ZK_ERR_ARG;
  // carry-propagate into 10 limbs, then reduce by repeated conditional subtraction of r shifted: the value is < 2^96 * 2^224, so fold the
  // top limbs with 2^256 = R mod r: v = lo + hi * (2^256 mod r)
  unsigned __int128 acc = 0;
  uint32_t limbs[11];
  for (int j = 0; j < 11; j++) limbs[j] = 0;
  for (int j = 0; j < 8; j++) {
    acc += (unsigned __int128)lanes8_h[j];
    limbs[j] = (uint32_t)acc;
    acc >>= 32;
  }
  limbs[8] = (uint32_t)acc;
  limbs[9] = (uint32_t)(acc >> 32);
  limbs[10] = (uint32_t)(acc >> 64);
  zk::Fr lo, hi = zk::fr_zero();
  for (int j = 0; j < 8; j++) lo.v[j] = limbs[j];
  hi.v[0] = limbs[8];
  hi.v[1] = limbs[9];
  hi.v[2] = limbs[10];
  // lo may be >= r (at most a few multiples): reduce
  zk::Fr rmod;
  for (int i = 0; i < 8; i++) rmod.v[i] = zk::FrP::P(i);
  while (zk::l_gte(lo, rmod)) {
    uint32_t bw = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t t = (uint64_t)lo.v[i] - rmod.v[i] - bw;
      lo.v[i] = (uint32_t)t;
      bw = (uint32_t)((t >> 32) & 1u);
    }
  }
  // hi * 2^256 mod r: 2^256 mod r is the Montgomery one; the Montgomery product of hi with R^2 gives hi * R = hi * 2^256 mod r
  zk::Fr r2;
  for (int i = 0; i < 8; i++) r2.v[i] = zk::FrP::R2(i);
  const zk::Fr top = zk::fr_mul(hi, r2);
  const zk::Fr v = zk::fr_add(lo, top);
  memcpy(out_h, &v, sizeof v);
  return ZK_OK;
}

int zk_dist_destroy(zk_dist* d) {
  if (!d) return ZK_ERR_ARG;
  Rccl* r = rccl();
  if (r && d->comm) (void)r->CommDestroy(d->comm);
  delete d;
  return ZK_OK;
}

}  // extern "C"
