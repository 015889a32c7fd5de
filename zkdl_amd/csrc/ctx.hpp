// zkdl_amd/csrc/ctx.hpp -- context, stream-ordered temporaries, launch + profiling plumbing.
//
// The reference allocates with cudaMalloc/cudaFree around every operator and calls
// cudaDeviceSynchronize() after every launch (fr-tensor.cu:92-127).  Here every operation is
// enqueued on one HIP stream, all device memory comes from the context's caching allocator (no
// driver call in steady state), and the host only waits when a value has to cross to it.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/zkdl_amd.h"
#include "field.hpp"
#include "g1.hpp"
#include "stage_ring.hpp"

struct zk_prof_entry {
  const char* name;
  double ms = 0;
  uint64_t launches = 0;
  uint64_t bytes = 0;
};

namespace zk {
struct Recorder;
}

struct zk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  // RNG (counter-based): element i of draw d of seed s is a pure function of (s, d, i)
  uint64_t rng_seed = 0;
  uint64_t rng_draw = 0;
  // MSM configuration
  int msm_mode = ZK_MSM_PIPPENGER;
  uint32_t msm_window = 0;
  bool flag_q16 = true;     // keep an int16 copy of quantised weights for the prover's contractions of W
  bool relu_packed = true;  // ReLU aux proofs from packed bits (relu_packed.hip) instead of one-Fr-per-bit tensors
  bool pipeline_fr = true;     // sealed proofs: do not join the Fr groups on the main stream (next proof's groups run right behind)
  bool derive_rounds = false;  // me_open: T_1 .. T_{k-1} from the folding relation behind the MSM batch instead of as MSMs (msm.hip);
                               // off by default: -31 % msm_accum but a 13 ms scalar-multiplication tail, 2 % net
  bool full_tables = true;  // full-width MSM jobs over generator sets that were prepared with zk_gens_prepare_full take the bucket-free path
  bool relu_bytes = true;   // ... through byte histograms (rounds 0-4 / 0-3 without a table) rather than the three-round table form
  // timing
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool prof_on = false;
  struct Pending {
    const char* name;
    uint64_t bytes;
    hipEvent_t a, b;
    hipStream_t stream;
  };
  std::vector<Pending> pending;
  hipEvent_t prof_base = nullptr;  // recorded when profiling is switched on: origin of the ZKDL_PROF_TRACE timeline
  std::vector<hipEvent_t> event_pool;
  std::map<std::string, zk_prof_entry> prof;
  // pinned staging ring for small host->device uploads (challenge vectors, broadcast scalars):
  // the caller's host buffer may die right after the call returns, the staged copy may not
  uint8_t* stage = nullptr;
  zk::StageRing stage_ring;  // offsets + current segment (stage_ring.hpp)
  static constexpr size_t STAGE_SEGS = 8;
  std::vector<hipEvent_t> stage_events[STAGE_SEGS];
  size_t stage_marked[STAGE_SEGS] = {};
  void* stage_put(const void* src, size_t bytes);  // returns pinned address holding a copy

  // ---- device memory: a size-class caching allocator over plain hipMalloc blocks.
  // A block may be handed out again as soon as it is released, PROVIDED every later use is ordered
  // behind every earlier use.  That holds per stream, so every stream ("lane", below) has its own pool:
  // a block is only ever re-issued by the pool that first issued it, i.e. to work on the same stream
  // (or on the main stream after that lane has been joined).  Blocks are never returned to the driver
  // before the context dies (288 GB of HBM), so steady-state proving makes no allocator calls into the
  // runtime at all.  (hipMallocAsync/hipFreeAsync were dropped: on ROCm 7.2 the stream-ordered pool was
  // observed to hand a still-live block out a second time.)
  struct Pool {
    std::map<size_t, std::vector<void*>> free_lists;  // class size -> idle blocks
  };
  struct Block {
    size_t cls;
    Pool* owner;
    bool idle;
  };
  std::unordered_map<void*, Block> blocks;  // every block ever allocated
  Pool main_pool;
  Pool* cur_pool = &main_pool;
  size_t pool_bytes = 0;
  uint64_t pool_double_release = 0;
  static size_t pool_class(size_t bytes) {
    if (bytes < 256) return 256;
    size_t p = 256;
    while (p < bytes) p <<= 1;  // smallest power of two >= bytes
    size_t q = p >> 3;          // 4 classes per octave: waste <= 19 %
    for (size_t c = (p >> 1) + q; c < p; c += q)
      if (c >= bytes) return c;
    return p;
  }
  bool rec_defer_free(void* p);
  void* pool_alloc(size_t bytes) {
    size_t cls = pool_class(bytes);
    auto& fl = cur_pool->free_lists[cls];
    if (!fl.empty()) {
      void* p = fl.back();
      fl.pop_back();
      blocks[p].idle = false;
      return p;
    }
    void* p = nullptr;
    static const bool trace = getenv("ZKDL_TRACE_ALLOC") != nullptr;
    if (trace) fprintf(stderr, "[zkdl] pool miss: %zu B (class %zu) pool %p, %zu MiB held\n", bytes, cls, (void*)cur_pool, pool_bytes >> 20);
    if (hipMalloc(&p, cls) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    blocks[p] = Block{cls, cur_pool, false};
    pool_bytes += cls;
    return p;
  }
  void pool_release(void* p) {
    if (!p) return;
    if (rec_defer_free(p)) return;  // recording: the block stays live until the recorded launches have run
    auto it = blocks.find(p);
    if (it == blocks.end()) return;  // not ours (never happens for library-made pointers)
    if (it->second.idle) {  // releasing twice would hand one block to two owners: refuse and remember
      ++pool_double_release;
      last_error = "internal error: device block released twice";
      return;
    }
    it->second.idle = true;
    it->second.owner->free_lists[it->second.cls].push_back(p);
  }
  void pool_destroy() {
    for (auto& kv : blocks) (void)hipFree(kv.first);
    blocks.clear();
    main_pool.free_lists.clear();
    for (auto& l : lanes) l.pool.free_lists.clear();
    for (auto& g : g1_lanes) g.pool.free_lists.clear();
    for (auto& g : tail_lanes) g.pool.free_lists.clear();
    g1_full.pool.free_lists.clear();
  }

  // ---- lanes: worker streams for independent per-layer proofs (zk_fc_prove / zk_relu_prove).
  // All challenges are known up front, so the layers' sumchecks are independent; running them on a few
  // streams overlaps their many small, latency-bound launches.  A lane scope forks from the main stream
  // (event wait), swaps ctx->stream / cur_pool, and records a `done` event when it ends; every other
  // entry point first makes the main stream wait on all dirty lanes (join_lanes).
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    Pool pool;
    bool dirty = false;
  };
  std::vector<Lane> lanes;
  // G1 streams: a sealed transcript runs its MSM batch, canonicalisation and digest on one of these
  // (round-robin), so the latency-bound tail of one proof's G1 phase overlaps the next one's accumulation
  std::vector<Lane> g1_lanes;
  uint32_t next_g1 = 0;
  // tail streams: the latency-bound end of a sealed proof's G1 phase (derived round commitments, the canonicalising inversion,
  // the digest: a handful of waves for milliseconds) runs here, behind an event of the G1 stream, so that the G1 streams -- which
  // carry the chip-filling MSM kernels -- are free for the next proof's batch.  Empty: everything stays on the G1 stream.
  std::vector<Lane> tail_lanes;
  uint32_t next_tail = 0;
  hipStream_t main_stream = nullptr;
  uint32_t* range_flag = nullptr;      // pinned host word a device-side check raises (msm.hip: k_bits_guard); zk_ctx_sync reports and clears it
  bool api_join = true;                // immediate tensor calls settle the context first (capi.hip: CHECK_CTX); off: main-stream order only
  hipStream_t copy_stream = nullptr;   // zk_transcript_read_sealed: device-to-host copies that wait for nothing but their own transcript
  hipEvent_t fork_ev = nullptr;
  hipEvent_t pre_ev = nullptr;   // main stream at the start of a seal(): what the early-started G1 phase waits for besides the opening chains
  bool early_g1 = true;          // sealed proofs: the G1 phase starts behind the opening's scalar preparation, not behind the whole Fr phase
  uint32_t g1_batch = 1;         // sealed proofs whose G1 phases run as ONE batch (1: every proof on its own)
  uint32_t g1_active = 0;        // G1 streams in use, round-robin (0 = all that were created); "g1_streams"
  Lane g1_full;                  // a G1 stream WITHOUT the CU mask: a proof with nothing to overlap wants the whole chip for its accumulation ("g1_mask" = 0)
  bool g1_masked = true;
  std::vector<zk_transcript*> g1_pending;  // sealed, Fr phase submitted, G1 phase waiting for partners
  int flush_g1_pending();        // layers.hip
  uint32_t layer_serial = 0;  // next zk_fc / zk_relu serial number (their lane hint)
  bool lanes_enabled = true;
  // main stream waits for the worker lanes (NOT for the G1 stream: a sealed proof keeps running there
  // under the next proof's Fr phase; whoever touches its transcript waits on that transcript's own event)
  void join_lanes() {
    for (auto& l : lanes)
      if (l.dirty) {
        (void)hipStreamWaitEvent(main_stream, l.done, 0);
        l.dirty = false;
      }
  }
  // `st` additionally waits for everything enqueued on the G1 streams so far
  void wait_g1(hipStream_t st) {
    for (auto& g : g1_lanes)
      if (g.dirty) (void)hipStreamWaitEvent(st, g.done, 0);
    for (auto& g : tail_lanes)
      if (g.dirty) (void)hipStreamWaitEvent(st, g.done, 0);
    if (g1_full.dirty) (void)hipStreamWaitEvent(st, g1_full.done, 0);
  }
  void join_all() {
    join_lanes();
    wait_g1(main_stream);
    for (auto& g : g1_lanes) g.dirty = false;
    for (auto& g : tail_lanes) g.dirty = false;
    g1_full.dirty = false;
  }

  int fail(int code, const std::string& msg) {
    last_error = msg;
    return code;
  }
  hipEvent_t get_event() {
    if (!event_pool.empty()) {
      hipEvent_t e = event_pool.back();
      event_pool.pop_back();
      return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  void prof_flush();

  // ---- deferred, horizontally batched execution of the layers' Fr work (launch.hpp / recorder)
  // While `rec` is set, every launch made through ZK_LAUNCH is RECORDED into the current chain instead
  // of being enqueued, temporaries are kept alive, and small host uploads go into the recorder's blob.
  // `pending` lists the recorders that hold unexecuted chains; settle() runs them (then joins the lanes)
  // so that every other entry point observes plain stream order.
  // kernels that opted in to > 64 KiB of dynamic LDS on THIS context's device (the attribute is per device)
  std::set<const void*> lds_opted;
  bool lds_first_use(const void* key) { return lds_opted.insert(key).second; }
  zk::Recorder* rec = nullptr;
  std::vector<zk::Recorder*> pending_recs;
  bool batch_layers = true;
  int settle();
};

namespace zk {

#define ZK_HIP(ctx, expr)                                                                        \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return (ctx)->fail(ZK_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));      \
  } while (0)

// Timing-experiment switches (skip the recorded Fr work, the MSM batch or the derived round commitments: the transcript is then
// garbage) exist only in builds made with `make EXPERIMENTS=1` (-DZKDL_EXPERIMENTS); the shipping library ignores the variables.
#ifdef ZKDL_EXPERIMENTS
inline bool experiment_on(const char* name) { return getenv(name) != nullptr; }
inline const char* experiment_env(const char* name) { return getenv(name); }
#else
constexpr bool experiment_on(const char*) { return false; }
// A/B switches that select between value-preserving code paths or scheduling knobs (stream counts, priorities, kernel variants): the shipping
// library always takes the default; tools/ab_*.sh run the EXPERIMENTS build (ZKDL_AMD_LIB=zkdl_amd/lib/variants/lib_exp.so)
constexpr const char* experiment_env(const char*) { return nullptr; }
#endif

#define ZK_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != ZK_OK) return _rc; \
  } while (0)

// Scope that brackets one kernel family launch with HIP events when profiling is enabled.
struct ProfScope {
  zk_ctx* ctx;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  uint64_t bytes;
  ProfScope(zk_ctx* c, const char* n, uint64_t by) : ctx(c), name(n), bytes(by) {
    if (ctx->prof_on) {
      a = ctx->get_event();
      b = ctx->get_event();
      if (a) (void)hipEventRecord(a, ctx->stream);
    }
  }
  ~ProfScope() {
    if (a && b) {
      (void)hipEventRecord(b, ctx->stream);
      ctx->pending.push_back({name, bytes, a, b, ctx->stream});
    }
  }
};

// RAII: run the enclosed work on a worker lane (no-op when lanes are disabled or unavailable).  `hint`
// is the layer object's serial number: a layer always proves on the same lane, so that lane's pool sees
// the same allocation sizes every proof and stays warm, and successive proofs of one layer simply queue
// behind each other in stream order.
struct LaneScope {
  zk_ctx* ctx;
  zk_ctx::Lane* lane = nullptr;
  LaneScope(zk_ctx* c, uint32_t hint) : ctx(c) {
    if (hint == ~0u || !c->lanes_enabled || c->lanes.empty() || c->stream != c->main_stream) return;
    lane = &c->lanes[hint % c->lanes.size()];
    (void)hipEventRecord(c->fork_ev, c->main_stream);
    (void)hipStreamWaitEvent(lane->stream, c->fork_ev, 0);
    c->stream = lane->stream;
    c->cur_pool = &lane->pool;
  }
  ~LaneScope() {
    if (!lane) return;
    (void)hipEventRecord(lane->done, lane->stream);
    lane->dirty = true;
    ctx->stream = ctx->main_stream;
    ctx->cur_pool = &ctx->main_pool;
  }
};

}  // namespace zk
#include "launch.hpp"
namespace zk {

// Persistent device buffers and temporaries both come from the context's caching allocator.
inline hipError_t dev_alloc(zk_ctx* ctx, void** p, size_t bytes) {
  *p = ctx->pool_alloc(bytes);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline void dev_free(zk_ctx* ctx, void* p) {
  if (ctx) ctx->pool_release(p);
}

// Temporary device buffer (released back to the context's allocator; reuse is stream-ordered).
template <class T>
struct DBuf {
  zk_ctx* ctx = nullptr;
  T* p = nullptr;
  size_t n = 0;
  bool borrowed = false;  // points into memory owned by someone else (the recorder's blob)
  DBuf() = default;
  DBuf(zk_ctx* c, size_t count) { alloc(c, count); }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : ctx(o.ctx), p(o.p), n(o.n), borrowed(o.borrowed) { o.p = nullptr; }
  DBuf& operator=(DBuf&& o) noexcept {
    release();
    ctx = o.ctx; p = o.p; n = o.n; borrowed = o.borrowed;
    o.p = nullptr;
    return *this;
  }
  void borrow(zk_ctx* c, T* ptr, size_t count) {
    release();
    ctx = c; p = ptr; n = count; borrowed = true;
  }
  bool alloc(zk_ctx* c, size_t count) {
    release();
    ctx = c;
    n = count;
    borrowed = false;
    size_t bytes = (count ? count : 1) * sizeof(T);
    p = static_cast<T*>(c->pool_alloc(bytes));
    return p != nullptr;
  }
  void release() {
    if (p && !borrowed) ctx->pool_release(p);
    p = nullptr;
    borrowed = false;
  }
  ~DBuf() { release(); }
  T* get() const { return p; }
  explicit operator bool() const { return p != nullptr; }
};

#define ZK_ALLOC(ctx, buf, T, count)                                                          \
  zk::DBuf<T> buf;                                                                            \
  if (!buf.alloc((ctx), (count))) return (ctx)->fail(ZK_ERR_ALLOC, "device allocation failed: " #buf)

inline uint32_t ceil_log2(uint32_t num) {  // proof.cu:13-30 (0 -> 0, 1 -> 0)
  if (num == 0) return 0;
  num--;
  uint32_t r = 0;
  while (num > 0) {
    num >>= 1;
    r++;
  }
  return r;
}

// grid for streaming kernels: enough blocks to fill 256 CUs several times over, grid-stride beyond
inline uint32_t stream_grid(uint64_t n, uint32_t block = 256, uint32_t max_blocks = 256 * 16) {
  uint64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (uint32_t)g;
}

// ---- counter-based RNG shared by host (random_vec) and device (FrTensor::random)
ZK_HD uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
ZK_HD Fr rng_fr(uint64_t seed, uint64_t draw, uint64_t idx) {
  Fr r;
  uint64_t base = mix64(seed ^ mix64(draw * 0xD1342543DE82EF95ull + 1)) + idx * 4;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint64_t w = mix64(base + q);
    r.v[2 * q] = (uint32_t)w;
    r.v[2 * q + 1] = (uint32_t)(w >> 32);
  }
  r.v[7] %= 1944954707u;  // fr-tensor.cu:346 / proof.cu:9: keeps the value below r
  return r;
}

// memops.hip
int dev_copy(zk_ctx* ctx, void* dst, const void* src, size_t bytes);
int dev_zero(zk_ctx* ctx, void* dst, size_t bytes);
int host_to_dev(zk_ctx* ctx, void* dst, const void* src, size_t bytes);
int dev_to_host(zk_ctx* ctx, void* dst, const void* src, size_t bytes);

// ---- internal (device-pointer) entry points shared between translation units
// fr_kernels.hip
int fr_binary(zk_ctx* ctx, int op, const Fr* a, const Fr* b, Fr* out, uint32_t n);
int fr_scalar(zk_ctx* ctx, int op, const Fr* a, const Fr& x, Fr* out, uint32_t n);
int fr_unary(zk_ctx* ctx, int op, const Fr* a, Fr* out, uint32_t n);
int fr_sum_dev(zk_ctx* ctx, const Fr* a, uint32_t n, Fr* dst_d);
int fr_me_dev(zk_ctx* ctx, const Fr* t, uint32_t n, const Fr* u_d, uint32_t k, Fr* dst_d);
int fr_partial_me_dev(zk_ctx* ctx, const Fr* t, uint32_t n, const Fr* u_d, uint32_t k, uint32_t window, Fr* out_d);
int fr_partial_me_q16_dev(zk_ctx* ctx, const int16_t* q, uint32_t n, const Fr* u_d, uint32_t k, uint32_t window, Fr* out_d);
int fr_quant16_dev(zk_ctx* ctx, const Fr* t, uint32_t n, int16_t* q, uint32_t* overflow_d);
int fr_fold_dev(zk_ctx* ctx, const Fr* in, Fr* out, const Fr* x_d, uint32_t in_size, uint32_t out_size);
uint32_t fr_partial_me_size(uint32_t n, uint32_t k, uint32_t window);
int fr_split_dev(zk_ctx* ctx, const Fr* t, uint32_t n, uint32_t window, Fr* o0, Fr* o1);
int fr_random_dev(zk_ctx* ctx, Fr* out, uint32_t n);
int fr_random_int_dev(zk_ctx* ctx, Fr* out, uint32_t n, uint32_t num_bits);
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_NEG = 3, OP_MONT = 4, OP_UNMONT = 5 };
// sumcheck.hip
int sumcheck_ip_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, const Fr* u_d, uint32_t k, Fr* proof_d);
int sumcheck_hp_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d, const uint8_t* b_bits = nullptr);
int sumcheck_bin_dev(zk_ctx* ctx, const Fr* a, uint32_t n, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d);
// g1_kernels.hip
int g1_sum_dev(zk_ctx* ctx, const G1J* a, uint32_t n, G1J* dst_d);
int g1_me_dev(zk_ctx* ctx, const G1J* t, uint32_t n, const Fr* u_d, uint32_t k, G1J* dst_d);
int g1_to_affine_dev(zk_ctx* ctx, const G1J* a, uint32_t n, G1A* out, uint8_t* inf);
// upload a host challenge vector (k may be 0)
inline int upload_fr(zk_ctx* ctx, DBuf<Fr>& buf, const zk_fr* h, uint32_t k) {
  if (ctx->rec && k) {  // recording: the vector travels in the recorder's blob, uploaded once per batch
    void* d = rec_blob_put(ctx, h, (size_t)k * sizeof(Fr));
    if (!d) return ctx->fail(ZK_ERR_ALLOC, "recorder blob exhausted");
    buf.borrow(ctx, static_cast<Fr*>(d), k);
    return ZK_OK;
  }
  if (!buf.alloc(ctx, k ? k : 1)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: challenges");
  if (k) {
    void* st = ctx->stage_put(h, (size_t)k * sizeof(Fr));
    if (!st) return ctx->fail(ZK_ERR_ALLOC, "pinned staging exhausted");
    ZK_TRY(dev_copy(ctx, buf.get(), st, (size_t)k * sizeof(Fr)));  // kernel reading the pinned staging ring
  }
  return ZK_OK;
}

}  // namespace zk
