// zkdl_amd/csrc/launch.hpp -- launch plumbing: immediate launches, and RECORDED launches that are
// executed later, horizontally batched across layers.
//
// Why: one layer's proof is a dependent chain of ~45 launches, most of them far too small to fill 256 CUs
// (a round on 2^12 elements, a one-workgroup tail, an eq table).  The reference runs the layers one after
// the other; worker streams ("lanes") overlap a few chains but HIP streams share a handful of hardware
// queues.  All challenges are known up front, so the layers' proofs are independent: zk_fc_prove /
// zk_relu_prove therefore only RECORD their launches (one chain per call), and when the transcript is
// flushed or sealed the chains are zipped -- launches of the same kernel at the heads of different chains
// become ONE launch whose blockIdx.z selects the job -- which divides the launch count by the number of
// layers and gives every launch several layers' worth of blocks.
//
// Kernels on this path are written as __device__ bodies  void body(uint32_t nblk, args...)  where nblk is
// the LOGICAL grid size in x (a batched launch is as wide as its widest job).  k_one<N, body> launches one
// immediately; k_many<N, body> reads {nblk, args...} of job blockIdx.z from a device table.
#pragma once
#include <functional>
#include <tuple>

namespace zk {

template <class... A>
struct Pack;
template <>
struct Pack<> {};
template <class H, class... T>
struct Pack<H, T...> {
  H head;
  Pack<T...> tail;
};
inline Pack<> make_pack() { return Pack<>{}; }
template <class H, class... T>
inline Pack<H, T...> make_pack(const H& h, const T&... t) {
  return Pack<H, T...>{h, make_pack(t...)};
}
template <auto Body, class... Done>
__device__ __forceinline__ void pack_apply(uint32_t nblk, const Pack<>&, const Done&... d) {
  Body(nblk, d...);
}
template <auto Body, class H, class... T, class... Done>
__device__ __forceinline__ void pack_apply(uint32_t nblk, const Pack<H, T...>& p, const Done&... d) {
  pack_apply<Body>(nblk, p.tail, d..., p.head);
}

template <class... A>
struct JobRec {
  uint32_t nblk;
  uint32_t pad_;
  Pack<A...> args;
};

template <int N, auto Body, class... A>
__global__ void __launch_bounds__(N) k_one(A... a) {
  Body(gridDim.x, a...);
}
template <int N, auto Body, class... A>
__global__ void __launch_bounds__(N) k_many(const JobRec<A...>* __restrict__ jobs) {
  const JobRec<A...>& j = jobs[blockIdx.z];  // uniform address: scalar loads
  if (blockIdx.x >= j.nblk) return;          // the launch is as wide as its widest job
  pack_apply<Body>(j.nblk, j.args);
}

struct KernelDesc {
  int (*launch_many)(hipStream_t st, dim3 grid, size_t shmem, const void* jobs_dev);
  uint32_t job_size;
};
struct RecOp {
  const KernelDesc* k = nullptr;  // nullptr: opaque (not batchable), run through `opaque`
  uint32_t gx = 1, gy = 1;
  size_t shmem = 0;
  const char* name = "";
  uint64_t bytes = 0;
  size_t args_off = 0;  // JobRec image inside Recorder::args
  std::function<int(hipStream_t)> opaque;
};
struct RecChain {
  uint32_t group = 0;  // chains of one group are zipped together and run on that group's stream
  bool early = false;  // scheduled before the group's other chains; an event behind the last early launch lets a consumer start
                       // before the rest of the Fr phase has run (the opening's scalar preparation: all the G1 phase needs)
  std::vector<RecOp> ops;
};
struct Recorder {
  std::vector<RecChain> chains;            // independent dependency chains, in recording order
  std::vector<uint8_t> args;               // JobRec images of all recorded launches
  std::vector<uint8_t> blob;               // host image of the device blob: small uploads, then the job tables
  uint8_t* blob_dev = nullptr;
  size_t blob_cap = 0;
  std::vector<void*> deferred_free;
  std::vector<void*> held_free;            // temporaries of an UNJOINED execution: released when the owner knows the lanes are done
  bool empty() const { return chains.empty(); }
};
static constexpr size_t REC_BLOB_CAP = 8u << 20;
// groups (= streams) the layers' independent pieces are spread over, balanced on the 18M-parameter MLP:
// FC sumcheck + Z(u) + ReLU partial_me's | FC opening + ReLU Hadamard sumcheck | ReLU magnitude bits | ReLU remainder bits
enum { REC_GROUP_FC_A = 0, REC_GROUP_FC_OPEN = 1, REC_GROUP_RELU_MAG = 2, REC_GROUP_RELU_REM = 3 };

int rec_begin_chain(zk_ctx* ctx, Recorder* r, uint32_t group = 0);  // sets ctx->rec; allocates the blob on first use
void rec_split(zk_ctx* ctx, uint32_t group, bool early = false);  // recording: what follows is independent of what came before
void rec_end_chain(zk_ctx* ctx);                           // clears ctx->rec
void rec_rollback(zk_ctx* ctx, Recorder* r, size_t keep_chains);  // a recording prove call failed: drop its chains
// zip + launch on ctx->stream; join: home waits for the groups, temporaries released.  early_out (optional): receives one event per
// group that had early chains, recorded behind the last early launch on that group's stream (events from ctx->get_event(); the caller
// returns them to ctx->event_pool once it has enqueued its waits)
int rec_execute(zk_ctx* ctx, Recorder* r, bool join = true, std::vector<hipEvent_t>* early_out = nullptr);
int rec_merge(zk_ctx* ctx, Recorder* dst, Recorder* src);  // src's chains join dst's: one execution zips both
void rec_release_held(zk_ctx* ctx, Recorder* r);           // after an unjoined execution, once every group stream is known to be ordered before the caller
void* rec_blob_put(zk_ctx* ctx, const void* src, size_t bytes);  // -> DEVICE address inside the blob
int rec_opaque(zk_ctx* ctx, const char* name, uint64_t bytes, std::function<int(hipStream_t)> fn);

// RAII: record the enclosed launches as one chain of `r` (no-op when batching is off)
struct RecScope {
  zk_ctx* ctx;
  bool on = false;
  int rc = ZK_OK;
  RecScope(zk_ctx* c, Recorder* r, uint32_t group) : ctx(c) {
    if (!c->batch_layers || c->rec || c->stream != c->main_stream) return;
    rc = rec_begin_chain(c, r, group);
    on = rc == ZK_OK;
  }
  ~RecScope() {
    if (on) rec_end_chain(ctx);
  }
};

template <int N, auto Body, class F = decltype(Body)>
struct Kern;
template <int N, auto Body, class... A>
struct Kern<N, Body, void (*)(uint32_t, A...)> {
  using Job = JobRec<A...>;
  static int launch_many(hipStream_t st, dim3 grid, size_t shmem, const void* jobs_dev) {
    hipLaunchKernelGGL((k_many<N, Body, A...>), grid, dim3(N), shmem, st, (const Job*)jobs_dev);
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
  // opt in to more than 64 KB of dynamic LDS (both launch forms)
  static void set_max_lds(size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_one<N, Body, A...>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_many<N, Body, A...>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  }
  static const KernelDesc* desc() {
    static const KernelDesc d{&launch_many, (uint32_t)sizeof(Job)};
    return &d;
  }
  static int launch(zk_ctx* ctx, const char* name, uint64_t bytes, dim3 grid, size_t shmem, A... a) {
    if (ctx->rec) {
      Recorder* r = ctx->rec;
      RecOp op;
      op.k = desc();
      op.gx = grid.x;
      op.gy = grid.y;
      op.shmem = shmem;
      op.name = name;
      op.bytes = bytes;
      op.args_off = r->args.size();
      Job j{grid.x, 0u, make_pack(a...)};
      const uint8_t* src = reinterpret_cast<const uint8_t*>(&j);
      r->args.insert(r->args.end(), src, src + sizeof(Job));
      r->chains.back().ops.push_back(std::move(op));
      return ZK_OK;
    }
    {
      ProfScope ps(ctx, name, bytes);
      hipLaunchKernelGGL((k_one<N, Body, A...>), grid, dim3(N), shmem, ctx->stream, a...);
    }
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return ctx->fail(ZK_ERR_DEVICE, std::string("launch of ") + name + " failed: " + hipGetErrorString(le));
    return ZK_OK;
  }
};

// batchable launch of a __device__ body with N threads per block; `bytes` = algorithmic HBM bytes
#define ZK_LAUNCHB(ctx, name, bytes, N, body, grid, shmem, ...)                                          \
  do {                                                                                                   \
    int _rc = zk::Kern<N, &body>::launch((ctx), (name), (uint64_t)(bytes), dim3(grid), (shmem), __VA_ARGS__); \
    if (_rc != ZK_OK) return _rc;                                                                        \
  } while (0)

// plain __global__ kernel on the context's stream (recorded as an opaque, unbatched step while recording)
#define ZK_LAUNCH(ctx, name, bytes, kernel, grid, block, shmem, ...)                                     \
  do {                                                                                                   \
    if ((ctx)->rec) {                                                                                    \
      auto _args = std::make_tuple(__VA_ARGS__);                                                         \
      dim3 _g = dim3(grid), _b = dim3(block);                                                                        \
      size_t _sh = (shmem);                                                                              \
      int _rc = zk::rec_opaque((ctx), (name), (uint64_t)(bytes), [=](hipStream_t _st) {                  \
        std::apply([&](auto... _a) { hipLaunchKernelGGL(kernel, _g, _b, _sh, _st, _a...); }, _args);     \
        return hipGetLastError() == hipSuccess ? 0 : 1;                                                  \
      });                                                                                                \
      if (_rc != ZK_OK) return _rc;                                                                      \
      break;                                                                                             \
    }                                                                                                    \
    {                                                                                                    \
      zk::ProfScope _ps((ctx), (name), (uint64_t)(bytes));                                               \
      hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);                        \
    }                                                                                                    \
    hipError_t _le = hipGetLastError();                                                                  \
    if (_le != hipSuccess) return (ctx)->fail(ZK_ERR_DEVICE, std::string("launch of ") + (name) + " failed: " + hipGetErrorString(_le)); \
  } while (0)

}  // namespace zk
