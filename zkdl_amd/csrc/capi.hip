// zkdl_amd/csrc/capi.hip -- the extern "C" surface declared in include/zkdl_amd.h.
// Thin: argument checks, host<->device staging, then the device-pointer entry points of the
// other translation units.  Nothing here (or anywhere in the library) falls back to a CPU path:
// without a usable HIP device zk_ctx_create fails and no other call can be made.
#include <string.h>

#include <string>

#include "layers.hpp"

using namespace zk;

static_assert(sizeof(zk_fr) == sizeof(Fr) && sizeof(zk_fp) == sizeof(Fp), "POD layout");
static_assert(sizeof(zk_g1a) == sizeof(G1A) && sizeof(zk_g1j) == sizeof(G1J), "POD layout");

// every entry point except the lane-parallel provers first joins the worker lanes
// every entry point except the prove calls first runs recorded-but-unexecuted layer work and joins the
// worker lanes, so callers observe plain stream order
// An immediate tensor call first SETTLES the context: pending recordings are executed and the main stream waits for every worker stream, so the call may
// read whatever recorded work wrote.  With the flag "api_join" off (zk_set_flag) it does neither: the call is launched on the main stream behind
// earlier main-stream work only -- for callers that know their operands do not come from recorded work still in flight (zkdl_amd/shard.py's streamed
// phases: static tensors and host challenges in, buffers that only LATER recorded launches read out; those fork from the main stream, i.e. behind it).
#define CHECK_CTX(ctx)                                  \
  if (!(ctx)) return ZK_ERR_ARG;                        \
  do {                                                  \
    int _src = (ctx)->api_join ? (ctx)->settle() : ZK_OK; \
    if (_src != ZK_OK) return _src;                     \
  } while (0)
#define CHECK_CTX_NOJOIN(ctx) \
  if (!(ctx)) return ZK_ERR_ARG
#define CHECK_PTR(ctx, p)                                                                  \
  if (!(p) || (((uintptr_t)(p)) & 15u)) return (ctx)->fail(ZK_ERR_ARG, "null or misaligned pointer: " #p)
#define CHECK_HOST(ctx, p) \
  if (!(p)) return (ctx)->fail(ZK_ERR_ARG, "null host pointer: " #p)

static inline const Fr* F(const zk_fr* p) { return reinterpret_cast<const Fr*>(p); }
static inline Fr* F(zk_fr* p) { return reinterpret_cast<Fr*>(p); }
static inline const G1J* J(const zk_g1j* p) { return reinterpret_cast<const G1J*>(p); }
static inline G1J* J(zk_g1j* p) { return reinterpret_cast<G1J*>(p); }
static inline const G1A* A(const zk_g1a* p) { return reinterpret_cast<const G1A*>(p); }
static inline G1A* A(zk_g1a* p) { return reinterpret_cast<G1A*>(p); }
static inline Fr load_fr(const zk_fr* h) {
  Fr r;
  memcpy(&r, h, sizeof(Fr));
  return r;
}

void* zk_ctx::stage_put(const void* src, size_t bytes) {
  // The ring is cut into STAGE_SEGS segments.  Leaving a segment records an event on every stream that can
  // have consumed it (main + lanes); re-entering it STAGE_SEGS-1 segments later waits on those events, which
  // by then have long completed, so steady-state proving never blocks the host here.
  const size_t need = (bytes + 63) & ~(size_t)63;
  size_t at = 0, left = 0;
  bool rolled = false;
  if (!stage_ring.place(need, &at, &rolled, &left)) return nullptr;
  if (rolled) {
    auto& evs = stage_events[left];
    size_t e = 0;
    auto mark = [&](hipStream_t st) {
      if (e == evs.size()) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
        evs.push_back(ev);
      }
      return hipEventRecord(evs[e++], st) == hipSuccess;
    };
    if (!mark(main_stream)) return nullptr;
    for (auto& l : lanes)
      if (!mark(l.stream)) return nullptr;
    for (auto& g : g1_lanes)
      if (!mark(g.stream)) return nullptr;
    for (auto& g : tail_lanes)
      if (!mark(g.stream)) return nullptr;
    stage_marked[left] = e;
    const size_t seg = stage_ring.seg;
    for (size_t q = 0; q < stage_marked[seg]; q++)
      if (hipEventSynchronize(stage_events[seg][q]) != hipSuccess) return nullptr;
    stage_marked[seg] = 0;
  }
  void* dst = stage + at;
  memcpy(dst, src, bytes);
  return dst;
}

void zk_ctx::prof_flush() {
  if (pending.empty()) return;
  join_all();
  (void)hipStreamSynchronize(main_stream);
  // ZKDL_PROF_TRACE=<file>: additionally append one line per launch (name, stream index, start and duration
  // in microseconds since profiling was enabled, algorithmic bytes) -- a timeline without an external profiler
  FILE* trace = nullptr;
  if (const char* path = getenv("ZKDL_PROF_TRACE")) trace = fopen(path, "a");
  auto stream_index = [&](hipStream_t st) {
    if (st == main_stream) return 0;
    for (size_t i = 0; i < lanes.size(); i++)
      if (lanes[i].stream == st) return (int)i + 1;
    for (size_t i = 0; i < g1_lanes.size(); i++)
      if (g1_lanes[i].stream == st) return 90 + (int)i;
    for (size_t i = 0; i < tail_lanes.size(); i++)
      if (tail_lanes[i].stream == st) return 80 + (int)i;
    return -1;
  };
  for (auto& p : pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto& e = prof[p.name];
      e.name = p.name;
      e.ms += ms;
      e.launches += 1;
      e.bytes += p.bytes;
      float t0ms = 0;
      if (trace && prof_base && hipEventElapsedTime(&t0ms, prof_base, p.a) == hipSuccess)
        fprintf(trace, "%s,%d,%.2f,%.2f,%llu\n", p.name, stream_index(p.stream), t0ms * 1e3, ms * 1e3, (unsigned long long)p.bytes);
    }
    event_pool.push_back(p.a);
    event_pool.push_back(p.b);
  }
  if (trace) fclose(trace);
  pending.clear();
}

extern "C" {

const char* zk_version(void) { return "zkdl_amd 0.1 (gfx950)"; }

int zk_ctx_create(int device, zk_ctx** out) {
  if (!out) return ZK_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZK_ERR_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return ZK_ERR_DEVICE;
  zk_ctx* ctx = new zk_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ctx->t0) != hipSuccess ||
      hipEventCreate(&ctx->t1) != hipSuccess || hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->pre_ev, hipEventDisableTiming) != hipSuccess) {
    delete ctx;
    return ZK_ERR_DEVICE;
  }
  ctx->main_stream = ctx->stream;
  {
    int nl = 8;
    if (const char* e = experiment_env("ZKDL_LANES")) nl = atoi(e);
    if (nl < 0) nl = 0;
    if (nl > 16) nl = 16;
    ctx->lanes.resize((size_t)nl);
    // ZKDL_LANE_PRIO (experiment): 1 = worker lanes at the highest stream priority (their launches are small and latency-bound; the G1 stream carries the bulk)
    int lane_prio = 0;
    if (const char* e = experiment_env("ZKDL_LANE_PRIO")) lane_prio = atoi(e);
    int lp_least = 0, lp_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&lp_least, &lp_greatest);
    for (auto& l : ctx->lanes)
      if ((lane_prio ? hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, lane_prio > 0 ? lp_greatest : lp_least) : hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking)) != hipSuccess ||
          hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) {
        ctx->lanes.clear();  // no lanes: everything stays on the main stream
        break;
      }
    // G1 stream priority: the MSM batch is bulk work that should fill the gaps the latency-bound Fr chains
    // leave, not delay them (ZKDL_G1_PRIO = -1 low, 0 normal (default; low priority with more than the default number of hardware queues stalled the pipeline), 1 high)
    int g1_prio = 0;
    if (const char* e = experiment_env("ZKDL_G1_PRIO")) g1_prio = atoi(e);
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    int prio = g1_prio < 0 ? least : g1_prio > 0 ? greatest : (least + greatest) / 2;
    {
      int ng = 2;  // a sealed proof's G1 phase is ~7 ms long (derived round commitments on the 28-bit-limb field); 3 or 4 streams measure slower
      if (const char* e = experiment_env("ZKDL_G1_STREAMS")) ng = atoi(e);
      ng = ng < 0 ? 0 : ng > 8 ? 8 : ng;
      ctx->g1_lanes.resize((size_t)ng);
      // The G1 streams are confined to every other compute unit: the bulk MSM accumulation (a whole-chip launch
      // of ~1.7 ms) then cannot take the chip away from the next proof's latency-bound Fr chains.  Measured on
      // the 18M MLP: 7.06 -> 6.77 ms/proof at batch 256, 5.68 -> 4.18 at batch 1; neutral on the width-4096 model.
      // Irregular masks (45 %, 55 %, 60 %) were much slower than none.  ZKDL_G1_CU_PERCENT=100 turns it off.
      int cu_pct = 50;
      if (const char* e = experiment_env("ZKDL_G1_CU_PERCENT")) cu_pct = atoi(e);
      std::vector<uint32_t> cu_mask;
      if (cu_pct > 0 && cu_pct < 100) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
          const int ncu = prop.multiProcessorCount;
          cu_mask.assign((size_t)(ncu + 31) / 32, 0u);
          for (int i = 0; i < ncu; i++)
            if ((i * cu_pct) / 100 != ((i + 1) * cu_pct) / 100) cu_mask[(size_t)i / 32] |= 1u << (i % 32);
        }
      }
      for (auto& g : ctx->g1_lanes) {
        bool ok = !cu_mask.empty() && hipExtStreamCreateWithCUMask(&g.stream, (uint32_t)cu_mask.size(), cu_mask.data()) == hipSuccess;
        if (!ok) {
          (void)hipGetLastError();
          ok = hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, prio) == hipSuccess;
        }
        if (!ok || hipEventCreateWithFlags(&g.done, hipEventDisableTiming) != hipSuccess) {
          ctx->g1_lanes.clear();  // sealing then degrades to the synchronous path
          break;
        }
      }
    }
  }
  if (!ctx->g1_lanes.empty()) {  // the G1 stream over ALL compute units ("g1_mask" = 0: single proofs, nothing to share the chip with).  Created through the
    // CU-mask call with a full mask: such a stream gets a hardware queue of its own -- an ordinary stream shares one with the Fr lanes, and a proof's
    // G1 phase then measured 6.3 ms instead of 5.1 ms on the half-chip stream
    hipDeviceProp_t prop;
    bool ok = false;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      std::vector<uint32_t> full((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
      ok = hipExtStreamCreateWithCUMask(&ctx->g1_full.stream, (uint32_t)full.size(), full.data()) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      ok = hipStreamCreateWithFlags(&ctx->g1_full.stream, hipStreamNonBlocking) == hipSuccess;
    }
    if (!ok || hipEventCreateWithFlags(&ctx->g1_full.done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      ctx->g1_full.stream = nullptr;
    }
  }
  {
    int nt = 0;  // ZKDL_TAIL_STREAMS: streams for the latency-bound tail of a sealed proof's G1 phase (0 = keep it on the G1 stream)
    if (const char* e = experiment_env("ZKDL_TAIL_STREAMS")) nt = atoi(e);
    nt = nt < 0 ? 0 : nt > 8 ? 8 : nt;
    if (ctx->g1_lanes.empty()) nt = 0;
    ctx->tail_lanes.resize((size_t)nt);
    // created with a CU mask (all CUs) like the G1 streams: what matters is that such a stream gets a hardware queue of its own --
    // a long low-occupancy kernel on an ordinary stream sits at the head of a hardware queue it shares with the Fr lanes
    // (ordinary tail streams: 6.9 ms/proof instead of 4.9)
    hipDeviceProp_t prop;
    std::vector<uint32_t> full;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) full.assign((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
    for (auto& t : ctx->tail_lanes) {
      bool ok = !full.empty() && hipExtStreamCreateWithCUMask(&t.stream, (uint32_t)full.size(), full.data()) == hipSuccess;
      if (!ok) {
        (void)hipGetLastError();
        ok = hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking) == hipSuccess;
      }
      if (!ok || hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) {
        ctx->tail_lanes.clear();
        break;
      }
    }
  }
  ctx->stage_ring.cap = 16u << 20;
  ctx->stage_ring.segs = zk_ctx::STAGE_SEGS;
  if (hipHostMalloc((void**)&ctx->stage, ctx->stage_ring.cap, hipHostMallocDefault) != hipSuccess) {
    delete ctx;
    return ZK_ERR_ALLOC;
  }
  if (hipHostMalloc((void**)&ctx->range_flag, 64, hipHostMallocDefault) == hipSuccess) *ctx->range_flag = 0;
  else ctx->range_flag = nullptr;
  *out = ctx;
  return ZK_OK;
}
int zk_ctx_destroy(zk_ctx* ctx) {
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  (void)ctx->flush_g1_pending();
  ctx->join_all();
  // drain the streams this context owns, one by one.  (Not hipDeviceSynchronize: the device may be shared with other contexts or with torch -- and
  // a device-wide synchronise at this point was observed to hang after long test sessions where the per-stream ones return at once.)
  (void)hipStreamSynchronize(ctx->main_stream);
  for (auto& l : ctx->lanes)
    if (l.stream) (void)hipStreamSynchronize(l.stream);
  for (auto& g : ctx->g1_lanes)
    if (g.stream) (void)hipStreamSynchronize(g.stream);
  for (auto& g : ctx->tail_lanes)
    if (g.stream) (void)hipStreamSynchronize(g.stream);
  if (ctx->g1_full.stream) (void)hipStreamSynchronize(ctx->g1_full.stream);
  ctx->prof_flush();
  for (auto& l : ctx->lanes) {
    if (l.stream) (void)hipStreamDestroy(l.stream);
    if (l.done) (void)hipEventDestroy(l.done);
  }
  for (auto& g : ctx->g1_lanes) {
    if (g.stream) (void)hipStreamDestroy(g.stream);
    if (g.done) (void)hipEventDestroy(g.done);
  }
  for (auto& g : ctx->tail_lanes) {
    if (g.stream) (void)hipStreamDestroy(g.stream);
    if (g.done) (void)hipEventDestroy(g.done);
  }
  if (ctx->g1_full.stream) (void)hipStreamDestroy(ctx->g1_full.stream);
  if (ctx->g1_full.done) (void)hipEventDestroy(ctx->g1_full.done);
  (void)hipEventDestroy(ctx->fork_ev);
  (void)hipEventDestroy(ctx->pre_ev);
  if (ctx->prof_base) (void)hipEventDestroy(ctx->prof_base);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  (void)hipEventDestroy(ctx->t0);
  (void)hipEventDestroy(ctx->t1);
  for (auto& evs : ctx->stage_events)
    for (auto e : evs) (void)hipEventDestroy(e);
  (void)hipHostFree(ctx->stage);
  ctx->pool_destroy();
  if (ctx->range_flag) (void)hipHostFree(ctx->range_flag);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  (void)hipStreamDestroy(ctx->main_stream);
  delete ctx;
  return ZK_OK;
}
int zk_ctx_sync(zk_ctx* ctx) {
  CHECK_CTX(ctx);
  ZK_TRY(ctx->flush_g1_pending());
  ctx->join_all();
  ZK_HIP(ctx, hipStreamSynchronize(ctx->main_stream));
  if (ctx->range_flag && *(volatile uint32_t*)ctx->range_flag) {
    *ctx->range_flag = 0;
    return ctx->fail(ZK_ERR_RANGE, "a scalar was longer than the bound its caller stated (zk_act_commit max_bits): the commitments of that batch are wrong");
  }
  return ZK_OK;
}
const char* zk_last_error(const zk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }
void* zk_ctx_stream(zk_ctx* ctx) { return ctx ? (void*)ctx->main_stream : nullptr; }

int zk_malloc(zk_ctx* ctx, size_t bytes, void** out_d) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, out_d);
  *out_d = ctx->pool_alloc(bytes ? bytes : 16);
  if (!*out_d) return ctx->fail(ZK_ERR_ALLOC, "device allocation of " + std::to_string(bytes) + " bytes failed");
  return ZK_OK;
}
int zk_free(zk_ctx* ctx, void* p) {
  CHECK_CTX(ctx);
  ctx->pool_release(p);
  return ZK_OK;
}
int zk_memcpy_h2d(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  CHECK_CTX(ctx);
  if (!bytes) return ZK_OK;
  return host_to_dev(ctx, dst, src, bytes);
}
int zk_memcpy_d2h(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  CHECK_CTX(ctx);
  if (!bytes) return ZK_OK;
  return dev_to_host(ctx, dst, src, bytes);
}
int zk_memcpy_d2d(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  CHECK_CTX(ctx);
  return dev_copy(ctx, dst, src, bytes);
}
int zk_memset_zero(zk_ctx* ctx, void* dst, size_t bytes) {
  CHECK_CTX(ctx);
  return dev_zero(ctx, dst, bytes);
}

int zk_timer_start(zk_ctx* ctx) {
  CHECK_CTX(ctx);
  ZK_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
  return ZK_OK;
}
int zk_timer_stop_ms(zk_ctx* ctx, float* ms) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, ms);
  ZK_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
  ZK_HIP(ctx, hipEventSynchronize(ctx->t1));
  ZK_HIP(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return ZK_OK;
}
int zk_prof_enable(zk_ctx* ctx, int on) {
  CHECK_CTX(ctx);
  ctx->prof_flush();
  ctx->prof_on = on != 0;
  if (ctx->prof_on) {
    if (!ctx->prof_base && hipEventCreate(&ctx->prof_base) != hipSuccess) ctx->prof_base = nullptr;
    if (ctx->prof_base) (void)hipEventRecord(ctx->prof_base, ctx->main_stream);
  }
  return ZK_OK;
}
int zk_prof_reset(zk_ctx* ctx) {
  CHECK_CTX(ctx);
  ctx->prof_flush();
  ctx->prof.clear();
  return ZK_OK;
}
int zk_prof_read(zk_ctx* ctx, const char** names, float* ms, uint64_t* launches, uint64_t* bytes, int cap, int* n) {
  CHECK_CTX(ctx);
  ctx->prof_flush();
  int i = 0;
  for (auto& kv : ctx->prof) {
    if (i >= cap) break;
    if (names) names[i] = kv.second.name;
    if (ms) ms[i] = (float)kv.second.ms;
    if (launches) launches[i] = kv.second.launches;
    if (bytes) bytes[i] = kv.second.bytes;
    i++;
  }
  if (n) *n = i;
  return ZK_OK;
}

// ---------------------------------------------------------------- RNG
int zk_rng_seed(zk_ctx* ctx, uint64_t seed) {
  CHECK_CTX(ctx);
  ctx->rng_seed = seed;
  ctx->rng_draw = 0;
  return ZK_OK;
}
int zk_random_vec(zk_ctx* ctx, uint32_t len, zk_fr* out_h) {
  CHECK_CTX_NOJOIN(ctx);  // host-only: no reason to make the main stream wait for the worker lanes
  if (len) CHECK_HOST(ctx, out_h);
  uint64_t draw = ctx->rng_draw++;
  for (uint32_t i = 0; i < len; i++) {
    Fr v = rng_fr(ctx->rng_seed, draw, i);
    memcpy(&out_h[i], &v, sizeof(Fr));
  }
  return ZK_OK;
}
int zk_fr_random(zk_ctx* ctx, zk_fr* out_d, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) CHECK_PTR(ctx, out_d);
  return fr_random_dev(ctx, F(out_d), n);
}
int zk_fr_random_int(zk_ctx* ctx, zk_fr* out_d, uint32_t n, uint32_t num_bits) {
  CHECK_CTX(ctx);
  if (n) CHECK_PTR(ctx, out_d);
  return fr_random_int_dev(ctx, F(out_d), n, num_bits);
}
uint32_t zk_ceil_log2(uint32_t num) { return ceil_log2(num); }

// ---------------------------------------------------------------- Fr tensor
#define FR_BIN(name, op)                                                                        \
  int name(zk_ctx* ctx, const zk_fr* a, const zk_fr* b, zk_fr* out, uint32_t n) {               \
    CHECK_CTX(ctx);                                                                             \
    if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, b); CHECK_PTR(ctx, out); }                       \
    return fr_binary(ctx, op, F(a), F(b), F(out), n);                                           \
  }
FR_BIN(zk_fr_add, OP_ADD)
FR_BIN(zk_fr_sub, OP_SUB)
FR_BIN(zk_fr_mul, OP_MUL)
#define FR_SCALAR(name, op)                                                                     \
  int name(zk_ctx* ctx, const zk_fr* a, const zk_fr* x_h, zk_fr* out, uint32_t n) {             \
    CHECK_CTX(ctx);                                                                             \
    CHECK_HOST(ctx, x_h);                                                                       \
    if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }                                          \
    return fr_scalar(ctx, op, F(a), load_fr(x_h), F(out), n);                                   \
  }
FR_SCALAR(zk_fr_add_scalar, OP_ADD)
FR_SCALAR(zk_fr_sub_scalar, OP_SUB)
FR_SCALAR(zk_fr_mul_scalar, OP_MUL)
#define FR_UN(name, op)                                                                         \
  int name(zk_ctx* ctx, const zk_fr* a, zk_fr* out, uint32_t n) {                               \
    CHECK_CTX(ctx);                                                                             \
    if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }                                          \
    return fr_unary(ctx, op, F(a), F(out), n);                                                  \
  }
FR_UN(zk_fr_neg, OP_NEG)
FR_UN(zk_fr_mont, OP_MONT)
FR_UN(zk_fr_unmont, OP_UNMONT)

int zk_fr_sum(zk_ctx* ctx, const zk_fr* a, uint32_t n, zk_fr* out_h) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, out_h);
  if (n) CHECK_PTR(ctx, a);
  ZK_ALLOC(ctx, dst, Fr, 1);
  ZK_TRY(fr_sum_dev(ctx, F(a), n, dst.get()));
  return zk_memcpy_d2h(ctx, out_h, dst.get(), sizeof(Fr));
}
int zk_fr_me(zk_ctx* ctx, const zk_fr* t, uint32_t n, const zk_fr* u_h, uint32_t k, zk_fr* out_h) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, out_h);
  if (n) CHECK_PTR(ctx, t);
  if (k) CHECK_HOST(ctx, u_h);
  DBuf<Fr> u;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  ZK_ALLOC(ctx, dst, Fr, 1);
  ZK_TRY(fr_me_dev(ctx, F(t), n, u.get(), k, dst.get()));
  return zk_memcpy_d2h(ctx, out_h, dst.get(), sizeof(Fr));
}
uint32_t zk_fr_partial_me_size(uint32_t n, uint32_t k, uint32_t window) { return window ? fr_partial_me_size(n, k, window) : 0; }
int zk_fr_partial_me(zk_ctx* ctx, const zk_fr* t, uint32_t n, const zk_fr* u_h, uint32_t k, uint32_t window, zk_fr* out) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, t); CHECK_PTR(ctx, out); }
  if (k) CHECK_HOST(ctx, u_h);
  DBuf<Fr> u;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  return fr_partial_me_dev(ctx, F(t), n, u.get(), k, window, F(out));
}
int zk_fr_split(zk_ctx* ctx, const zk_fr* t, uint32_t n, uint32_t window, zk_fr* o0, zk_fr* o1) {
  CHECK_CTX(ctx);
  CHECK_PTR(ctx, t); CHECK_PTR(ctx, o0); CHECK_PTR(ctx, o1);
  return fr_split_dev(ctx, F(t), n, window, F(o0), F(o1));
}

// ---------------------------------------------------------------- sumchecks
int zk_sumcheck_ip(zk_ctx* ctx, const zk_fr* a, const zk_fr* b, uint32_t n, const zk_fr* u_h, uint32_t k, zk_fr* proof_h) {
  CHECK_CTX(ctx);
  CHECK_PTR(ctx, a); CHECK_PTR(ctx, b); CHECK_HOST(ctx, proof_h);
  if (k) CHECK_HOST(ctx, u_h);
  DBuf<Fr> u;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  ZK_ALLOC(ctx, proof, Fr, 3 * (size_t)k + 2);
  ZK_TRY(sumcheck_ip_dev(ctx, F(a), F(b), n, u.get(), k, proof.get()));
  return zk_memcpy_d2h(ctx, proof_h, proof.get(), (3 * (size_t)k + 2) * sizeof(Fr));
}
int zk_sumcheck_hadamard(zk_ctx* ctx, const zk_fr* a, const zk_fr* b, uint32_t n, const zk_fr* u_h, const zk_fr* v_h, uint32_t k, zk_fr* proof_h) {
  CHECK_CTX(ctx);
  CHECK_PTR(ctx, a); CHECK_PTR(ctx, b); CHECK_HOST(ctx, proof_h);
  if (k) { CHECK_HOST(ctx, u_h); CHECK_HOST(ctx, v_h); }
  DBuf<Fr> u, v;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  ZK_TRY(upload_fr(ctx, v, v_h, k));
  ZK_ALLOC(ctx, proof, Fr, 3 * (size_t)k + 2);
  ZK_TRY(sumcheck_hp_dev(ctx, F(a), F(b), n, u.get(), v.get(), k, proof.get()));
  return zk_memcpy_d2h(ctx, proof_h, proof.get(), (3 * (size_t)k + 2) * sizeof(Fr));
}
int zk_sumcheck_binary(zk_ctx* ctx, const zk_fr* a, uint32_t n, const zk_fr* u_h, const zk_fr* v_h, uint32_t k, zk_fr* proof_h) {
  CHECK_CTX(ctx);
  CHECK_PTR(ctx, a); CHECK_HOST(ctx, proof_h);
  if (k) { CHECK_HOST(ctx, u_h); CHECK_HOST(ctx, v_h); }
  DBuf<Fr> u, v;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  ZK_TRY(upload_fr(ctx, v, v_h, k));
  ZK_ALLOC(ctx, proof, Fr, 3 * (size_t)k + 1);
  ZK_TRY(sumcheck_bin_dev(ctx, F(a), n, u.get(), v.get(), k, proof.get()));
  return zk_memcpy_d2h(ctx, proof_h, proof.get(), (3 * (size_t)k + 1) * sizeof(Fr));
}

// ---------------------------------------------------------------- G1 tensor
int zk_g1_fill(zk_ctx* ctx, zk_g1j* out, uint32_t n, const zk_g1j* g_h) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, g_h);
  if (n) CHECK_PTR(ctx, out);
  G1J g;
  memcpy(&g, g_h, sizeof g);
  return g1_fill_dev(ctx, J(out), n, g);
}
int zk_g1a_fill(zk_ctx* ctx, zk_g1a* out, uint32_t n, const zk_g1a* g_h) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, g_h);
  if (n) CHECK_PTR(ctx, out);
  G1A g;
  memcpy(&g, g_h, sizeof g);
  return g1a_fill_dev(ctx, A(out), n, g);
}
int zk_g1_from_affine(zk_ctx* ctx, const zk_g1a* a, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }
  return g1_from_affine_dev(ctx, A(a), J(out), n);
}
int zk_g1_neg(zk_ctx* ctx, const zk_g1j* a, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }
  return g1_neg_dev(ctx, J(a), J(out), n);
}
int zk_g1a_neg(zk_ctx* ctx, const zk_g1a* a, zk_g1a* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }
  return g1a_neg_dev(ctx, A(a), A(out), n);
}
int zk_g1_add(zk_ctx* ctx, const zk_g1j* a, const zk_g1j* b, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, b); CHECK_PTR(ctx, out); }
  return g1_addsub_dev(ctx, false, J(a), J(b), J(out), n);
}
int zk_g1_sub(zk_ctx* ctx, const zk_g1j* a, const zk_g1j* b, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, b); CHECK_PTR(ctx, out); }
  return g1_addsub_dev(ctx, true, J(a), J(b), J(out), n);
}
int zk_g1_madd(zk_ctx* ctx, const zk_g1j* a, const zk_g1a* b, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, b); CHECK_PTR(ctx, out); }
  return g1_maddsub_dev(ctx, false, J(a), A(b), J(out), n);
}
int zk_g1_msub(zk_ctx* ctx, const zk_g1j* a, const zk_g1a* b, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, b); CHECK_PTR(ctx, out); }
  return g1_maddsub_dev(ctx, true, J(a), A(b), J(out), n);
}
int zk_g1_add_bcast(zk_ctx* ctx, const zk_g1j* a, const zk_g1j* x_h, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, x_h);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }
  G1J x;
  memcpy(&x, x_h, sizeof x);
  return g1_addsub_bcast_dev(ctx, false, J(a), x, J(out), n);
}
int zk_g1_sub_bcast(zk_ctx* ctx, const zk_g1j* a, const zk_g1j* x_h, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, x_h);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }
  G1J x;
  memcpy(&x, x_h, sizeof x);
  return g1_addsub_bcast_dev(ctx, true, J(a), x, J(out), n);
}
int zk_g1_madd_bcast(zk_ctx* ctx, const zk_g1j* a, const zk_g1a* x_h, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, x_h);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }
  G1A x;
  memcpy(&x, x_h, sizeof x);
  return g1_maddsub_bcast_dev(ctx, false, J(a), x, J(out), n);
}
int zk_g1_msub_bcast(zk_ctx* ctx, const zk_g1j* a, const zk_g1a* x_h, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, x_h);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); }
  G1A x;
  memcpy(&x, x_h, sizeof x);
  return g1_maddsub_bcast_dev(ctx, true, J(a), x, J(out), n);
}
int zk_g1_sum(zk_ctx* ctx, const zk_g1j* a, uint32_t n, zk_g1j* out_h) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, out_h);
  if (n) CHECK_PTR(ctx, a);
  ZK_ALLOC(ctx, dst, G1J, 1);
  ZK_TRY(g1_sum_dev(ctx, J(a), n, dst.get()));
  return zk_memcpy_d2h(ctx, out_h, dst.get(), sizeof(G1J));
}
int zk_g1_mul(zk_ctx* ctx, const zk_g1j* g, const zk_fr* s, zk_g1j* out, uint32_t n) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, g); CHECK_PTR(ctx, s); CHECK_PTR(ctx, out); }
  return g1_mul_dev(ctx, J(g), n, F(s), n, J(out));
}
int zk_g1_mul_bcast(zk_ctx* ctx, const zk_g1j* g, uint32_t n, const zk_fr* s, uint32_t total, zk_g1j* out) {
  CHECK_CTX(ctx);
  if (n == 0 || total % n != 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // g1-tensor.cu:448
  if (total) { CHECK_PTR(ctx, g); CHECK_PTR(ctx, s); CHECK_PTR(ctx, out); }
  return g1_mul_dev(ctx, J(g), n, F(s), total, J(out));
}
int zk_g1_me(zk_ctx* ctx, const zk_g1j* t, uint32_t n, const zk_fr* u_h, uint32_t k, zk_g1j* out_h) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, out_h);
  if (n) CHECK_PTR(ctx, t);
  if (k) CHECK_HOST(ctx, u_h);
  DBuf<Fr> u;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  ZK_ALLOC(ctx, dst, G1J, 1);
  ZK_TRY(g1_me_dev(ctx, J(t), n, u.get(), k, dst.get()));
  return zk_memcpy_d2h(ctx, out_h, dst.get(), sizeof(G1J));
}
int zk_g1_to_affine(zk_ctx* ctx, const zk_g1j* a, uint32_t n, zk_g1a* out, uint8_t* inf) {
  CHECK_CTX(ctx);
  if (n) { CHECK_PTR(ctx, a); CHECK_PTR(ctx, out); CHECK_HOST(ctx, inf); }
  return g1_to_affine_dev(ctx, J(a), n, A(out), inf);
}

// ---------------------------------------------------------------- commitment
int zk_set_msm_mode(zk_ctx* ctx, int mode) {
  CHECK_CTX(ctx);
  if (mode != ZK_MSM_LITERAL && mode != ZK_MSM_PIPPENGER) return ctx->fail(ZK_ERR_ARG, "unknown MSM mode");
  ctx->msm_mode = mode;
  return ZK_OK;
}
int zk_set_flag(zk_ctx* ctx, const char* name, int value) {
  CHECK_CTX_NOJOIN(ctx);
  CHECK_HOST(ctx, name);
  if (!strcmp(name, "lanes") || !strcmp(name, "batch_layers")) {  // these change how recorded work runs: settle what is pending first
    int src = ctx->settle();
    if (src != ZK_OK) return src;
  }
  if (!strcmp(name, "relu_packed")) ctx->relu_packed = value != 0;
  else if (!strcmp(name, "relu_bytes")) ctx->relu_bytes = value != 0;
  else if (!strcmp(name, "derive_rounds")) ctx->derive_rounds = value != 0;
  else if (!strcmp(name, "pipeline_fr")) ctx->pipeline_fr = value != 0;
  else if (!strcmp(name, "early_g1")) ctx->early_g1 = value != 0;
  else if (!strcmp(name, "api_join")) ctx->api_join = value != 0;
  else if (!strcmp(name, "full_tables")) ctx->full_tables = value != 0;
  else if (!strcmp(name, "g1_streams")) ctx->g1_active = value < 0 ? 0u : (uint32_t)value;
  else if (!strcmp(name, "g1_mask")) ctx->g1_masked = value != 0;
  else if (!strcmp(name, "g1_batch")) {
    ZK_TRY(ctx->flush_g1_pending());
    ctx->g1_batch = value < 1 ? 1u : value > 8 ? 8u : (uint32_t)value;
  }
  else if (!strcmp(name, "lanes")) ctx->lanes_enabled = value != 0;
  else if (!strcmp(name, "batch_layers")) ctx->batch_layers = value != 0;
  else if (!strcmp(name, "weights_q16")) ctx->flag_q16 = value != 0;
  else return ctx->fail(ZK_ERR_ARG, std::string("unknown flag: ") + name);
  return ZK_OK;
}
int zk_set_msm_window(zk_ctx* ctx, uint32_t w) {
  CHECK_CTX(ctx);
  // 2 .. 13: the counting sort keeps one LDS counter per bucket (2^(c-1) + 1 of them); wider windows used to be clamped silently
  if (w == 1 || w > 13) return ctx->fail(ZK_ERR_ARG, "MSM window must be 0 (auto) or 2..13 bits");
  ctx->msm_window = w;
  return ZK_OK;
}
int zk_gens_create(zk_ctx* ctx, const zk_g1j* gens, uint32_t n, zk_gens** out) {
  CHECK_CTX(ctx);
  CHECK_PTR(ctx, gens); CHECK_HOST(ctx, out);
  return gens_create(ctx, J(gens), n, true, out);
}
int zk_gens_prepare_full(zk_ctx* ctx, zk_gens* g, uint32_t window_bits) {
  CHECK_CTX(ctx);
  if (!g) return ctx->fail(ZK_ERR_ARG, "null zk_gens");
  return gens_prepare_full(ctx, g, window_bits);
}
int zk_gens_destroy(zk_gens* g) {
  if (!g) return ZK_ERR_ARG;
  if (g->ctx) {
    (void)g->ctx->settle();
    (void)g->ctx->flush_g1_pending();  // sealed proofs still waiting for G1-batch partners hold pointers into these tables: run them now
    g->ctx->join_all();  // a sealed proof may still be reading the window tables on the G1 stream
  }
  gens_destroy(g);
  return ZK_OK;
}
uint32_t zk_gens_size(const zk_gens* g) { return g ? g->n : 0; }
const zk_g1j* zk_gens_points(const zk_gens* g) { return g ? reinterpret_cast<const zk_g1j*>(g->pts) : nullptr; }
int zk_commit(zk_ctx* ctx, zk_gens* gens, const zk_fr* t, uint32_t tn, zk_g1j* out) {
  CHECK_CTX(ctx);
  if (!gens) return ctx->fail(ZK_ERR_ARG, "null zk_gens");
  CHECK_PTR(ctx, t); CHECK_PTR(ctx, out);
  return commit_dev(ctx, gens, F(t), tn, J(out));
}
int zk_me_open(zk_ctx* ctx, zk_gens* gens, const zk_fr* t, const zk_fr* u_h, uint32_t k, zk_g1j* proof_h, zk_fr* ret_h) {
  CHECK_CTX(ctx);
  if (!gens) return ctx->fail(ZK_ERR_ARG, "null zk_gens");
  CHECK_PTR(ctx, t); CHECK_HOST(ctx, proof_h); CHECK_HOST(ctx, ret_h);
  if (k) CHECK_HOST(ctx, u_h);
  DBuf<Fr> u;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  ZK_ALLOC(ctx, proof, G1J, 3 * (size_t)k + 1);
  ZK_ALLOC(ctx, ret, Fr, 1);
  ZK_TRY(me_open_dev(ctx, gens, F(t), u.get(), k, proof.get(), ret.get(), nullptr));
  ZK_TRY(zk_memcpy_d2h(ctx, proof_h, proof.get(), (3 * (size_t)k + 1) * sizeof(G1J)));
  return zk_memcpy_d2h(ctx, ret_h, ret.get(), sizeof(Fr));
}
int zk_open(zk_ctx* ctx, zk_gens* gens, const zk_fr* t, uint32_t tn, const zk_g1j* com, uint32_t m, const zk_fr* u_h, uint32_t k, zk_g1j* com_eval_h,
            zk_g1j* proof_h, zk_fr* ret_h) {
  CHECK_CTX(ctx);
  if (!gens) return ctx->fail(ZK_ERR_ARG, "null zk_gens");
  CHECK_PTR(ctx, t); CHECK_PTR(ctx, com);
  CHECK_HOST(ctx, com_eval_h); CHECK_HOST(ctx, proof_h); CHECK_HOST(ctx, ret_h);
  if (k) CHECK_HOST(ctx, u_h);
  uint32_t k_hi = ceil_log2(m);
  if (m == 0 || k_hi > k) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  uint32_t k_lo = k - k_hi;
  DBuf<Fr> u;
  ZK_TRY(upload_fr(ctx, u, u_h, k));
  ZK_ALLOC(ctx, proof, G1J, 3 * (size_t)k_lo + 2);
  ZK_ALLOC(ctx, ret, Fr, 1);
  zk_gens* cg = nullptr;  // ad-hoc handle over the caller's commitment tensor (zkFC keeps a persistent one)
  ZK_TRY(gens_create(ctx, J(com), m, true, &cg));
  int rc = open_dev(ctx, gens, F(t), tn, cg, u.get(), k, proof.get(), proof.get() + 1, ret.get(), nullptr);
  if (rc == ZK_OK) rc = zk_memcpy_d2h(ctx, com_eval_h, proof.get(), sizeof(G1J));
  if (rc == ZK_OK) rc = zk_memcpy_d2h(ctx, proof_h, proof.get() + 1, (3 * (size_t)k_lo + 1) * sizeof(G1J));
  if (rc == ZK_OK) rc = zk_memcpy_d2h(ctx, ret_h, ret.get(), sizeof(Fr));
  gens_destroy(cg);
  return rc;
}
int zk_msm_batch(zk_ctx* ctx, zk_gens* gens, const zk_fr* scalars, uint32_t q, zk_g1j* out) {
  CHECK_CTX(ctx);
  if (!gens) return ctx->fail(ZK_ERR_ARG, "null zk_gens");
  if (q) { CHECK_PTR(ctx, scalars); CHECK_PTR(ctx, out); }
  return msm_batch_dev(ctx, gens, F(scalars), q, /*do_unmont=*/false, /*probe_bits=*/false, J(out));
}

// ---------------------------------------------------------------- layers
int zk_float_to_fr(zk_ctx* ctx, const float* fs, zk_fr* out, uint32_t rows, uint32_t rows_p, uint32_t cols, uint32_t cols_p) {
  CHECK_CTX(ctx);
  CHECK_PTR(ctx, out);
  if (!fs) return ctx->fail(ZK_ERR_ARG, "null pointer: fs_d");
  return float_to_fr_dev(ctx, fs, F(out), rows, rows_p, cols, cols_p);
}
int zk_fc_create(zk_ctx* ctx, uint32_t in, uint32_t out, const zk_fr* w, zk_gens* gens, zk_fc** out_fc) {
  CHECK_CTX(ctx);
  if (!gens) return ctx->fail(ZK_ERR_ARG, "null zk_gens");
  CHECK_PTR(ctx, w); CHECK_HOST(ctx, out_fc);
  return fc_create(ctx, in, out, F(w), gens, out_fc);
}
int zk_fc_from_float(zk_ctx* ctx, uint32_t in, uint32_t out, const float* w, zk_gens* gens, zk_fc** out_fc) {
  CHECK_CTX(ctx);
  if (!gens) return ctx->fail(ZK_ERR_ARG, "null zk_gens");
  CHECK_HOST(ctx, out_fc);
  if (!w) return ctx->fail(ZK_ERR_ARG, "null pointer: w_d");
  if (in == 0 || out == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  uint32_t in_p = 1u << ceil_log2(in), out_p = 1u << ceil_log2(out);  // zkfc.cu:92-93
  ZK_ALLOC(ctx, wq, Fr, (size_t)in_p * out_p);
  ZK_TRY(float_to_fr_dev(ctx, w, wq.get(), in, in_p, out, out_p));
  ZK_TRY(fr_unary(ctx, OP_MONT, wq.get(), wq.get(), in_p * out_p));  // weights.mont(), zkfc.cu:99
  return fc_create(ctx, in_p, out_p, wq.get(), gens, out_fc);
}
int zk_fc_destroy(zk_fc* fc) {
  if (!fc) return ZK_ERR_ARG;
  if (fc->ctx) {
    (void)fc->ctx->settle();
    (void)fc->ctx->flush_g1_pending();  // deferred G1 phases read the commitment points' tables
    fc->ctx->join_all();
  }
  if (fc->ctx) dev_free(fc->ctx, fc->weights);
  if (fc->ctx) dev_free(fc->ctx, fc->wq);
  gens_destroy(fc->com);
  delete fc;
  return ZK_OK;
}
uint32_t zk_fc_in(const zk_fc* fc) { return fc ? fc->in : 0; }
uint32_t zk_fc_out(const zk_fc* fc) { return fc ? fc->out : 0; }
uint32_t zk_fc_com_size(const zk_fc* fc) { return fc ? fc->m : 0; }
const zk_fr* zk_fc_weights(const zk_fc* fc) { return fc ? reinterpret_cast<const zk_fr*>(fc->weights) : nullptr; }
const zk_g1j* zk_fc_com(const zk_fc* fc) { return fc && fc->com ? reinterpret_cast<const zk_g1j*>(fc->com->pts) : nullptr; }
int zk_fc_load_input(zk_ctx* ctx, uint32_t batch, uint32_t in_dim, const float* x, zk_fr* out) {
  CHECK_CTX(ctx);
  CHECK_PTR(ctx, out);
  if (!x) return ctx->fail(ZK_ERR_ARG, "null pointer: x_d");
  if (batch == 0 || in_dim == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  return float_to_fr_dev(ctx, x, F(out), batch, 1u << ceil_log2(batch), in_dim, 1u << ceil_log2(in_dim));  // zkfc.cu:108-111
}
int zk_fc_forward(zk_ctx* ctx, const zk_fc* fc, const zk_fr* x, uint32_t xn, zk_fr* z) {
  CHECK_CTX(ctx);
  if (!fc) return ctx->fail(ZK_ERR_ARG, "null zk_fc");
  CHECK_PTR(ctx, x); CHECK_PTR(ctx, z);
  return fc_forward(ctx, fc, F(x), xn, F(z));
}
int zk_fc_prove(zk_ctx* ctx, zk_fc* fc, const zk_fr* x, uint32_t xn, const zk_fr* z, uint32_t zn, zk_gens* gens, const zk_fr* u_bs_h,
                const zk_fr* u_in_h, const zk_fr* u_out_h, zk_transcript* tr) {
  CHECK_CTX_NOJOIN(ctx);
  if (!fc || !tr || !gens) return ctx->fail(ZK_ERR_ARG, "null zk_fc / zk_gens / zk_transcript");
  CHECK_PTR(ctx, x); CHECK_PTR(ctx, z);
  return fc_prove(ctx, fc, F(x), xn, F(z), zn, gens, u_bs_h, u_in_h, u_out_h, tr);
}
int zk_relu_create(zk_ctx* ctx, zk_relu** out) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, out);
  *out = new zk_relu();
  (*out)->serial = ctx->layer_serial++;
  (*out)->ctx = ctx;
  return ZK_OK;
}
int zk_relu_destroy(zk_relu* relu) {
  if (!relu) return ZK_ERR_ARG;
  if (relu->ctx) (void)relu->ctx->settle();  // recorded layer work may still reference the aux tensors
  relu->free_all();
  delete relu;
  return ZK_OK;
}
int zk_relu_forward(zk_ctx* ctx, zk_relu* relu, const zk_fr* x, uint32_t n, zk_fr* z) {
  CHECK_CTX(ctx);
  if (!relu) return ctx->fail(ZK_ERR_ARG, "null zk_relu");
  if (n) { CHECK_PTR(ctx, x); CHECK_PTR(ctx, z); }
  return relu_forward(ctx, relu, F(x), n, F(z));
}
const zk_fr* zk_relu_sign(const zk_relu* r) { return r ? reinterpret_cast<const zk_fr*>(r->sign) : nullptr; }
// the bit tensors are materialised (from the packed words) the first time they are asked for
const zk_fr* zk_relu_mag_bin(const zk_relu* r) {
  if (!r || !r->ctx || r->ctx->settle() != ZK_OK || relu_materialise_bins(r->ctx, const_cast<zk_relu*>(r)) != ZK_OK) return nullptr;
  return reinterpret_cast<const zk_fr*>(r->mag_bin);
}
const zk_fr* zk_relu_rem_bin(const zk_relu* r) {
  if (!r || !r->ctx || r->ctx->settle() != ZK_OK || relu_materialise_bins(r->ctx, const_cast<zk_relu*>(r)) != ZK_OK) return nullptr;
  return reinterpret_cast<const zk_fr*>(r->rem_bin);
}
int zk_relu_prove(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x, const zk_fr* z, uint32_t n, const zk_fr* u_z, const zk_fr* v_z, const zk_fr* u_r,
                  const zk_fr* v_r, const zk_fr* u_rec, const zk_fr* u_hp, const zk_fr* v_hp, zk_transcript* tr) {
  CHECK_CTX_NOJOIN(ctx);
  if (!relu || !tr) return ctx->fail(ZK_ERR_ARG, "null zk_relu / zk_transcript");
  CHECK_PTR(ctx, x); CHECK_PTR(ctx, z);
  return relu_prove(ctx, relu, F(x), F(z), n, u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp, tr);
}

int zk_relu_set_activation(zk_relu* relu, uint32_t kind, uint32_t shift) {
  if (!relu || !relu->ctx) return ZK_ERR_ARG;
  if (kind > 1 || shift > 31) return relu->ctx->fail(ZK_ERR_ARG, "zk_relu_set_activation: kind 0 (ReLU) or 1 (LeakyReLU, negative slope 2^-shift, shift <= 31)");
  relu->act_kind = kind;
  relu->act_shift = kind ? shift : 0;
  return ZK_OK;
}
int zk_leaky_prove(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x, const zk_fr* z, uint32_t n, const zk_fr* u_z, const zk_fr* v_z, const zk_fr* u_r, const zk_fr* v_r,
                   const zk_fr* u_rec, const zk_fr* u_hp, const zk_fr* v_hp, const zk_fr* u_lk, const zk_fr* v_lk, zk_transcript* tr) {
  CHECK_CTX_NOJOIN(ctx);
  if (!relu || !tr) return ctx->fail(ZK_ERR_ARG, "null zk_relu / zk_transcript");
  CHECK_PTR(ctx, x); CHECK_PTR(ctx, z);
  const zk_fr* ch[9] = {u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp, u_lk, v_lk};
  return leaky_prove(ctx, relu, F(x), F(z), n, ch, tr);
}

// ---------------------------------------------------------------- transcript
int zk_transcript_create(zk_ctx* ctx, zk_transcript** out) {
  CHECK_CTX(ctx);
  CHECK_HOST(ctx, out);
  *out = new zk_transcript();
  (*out)->ctx = ctx;
  return ZK_OK;
}
int zk_transcript_destroy(zk_transcript* tr) {
  if (!tr) return ZK_ERR_ARG;
  tr->ctx->join_lanes();
  (void)tr->flush();
  zk::rec_release_held(tr->ctx, &tr->rec);
  if (tr->sealed_ev) (void)hipEventDestroy(tr->sealed_ev);
  dev_free(tr->ctx, tr->fr_d);
  dev_free(tr->ctx, tr->g1_d);
  delete tr;
  return ZK_OK;
}
int zk_transcript_reset(zk_transcript* tr) {
  if (!tr) return ZK_ERR_ARG;
  int rc = ZK_OK;
  tr->unseal();                                // a sealed G1 phase may still be reading the buffers
  // pending MSMs own scratch that must be released in stream order, and recorded-but-unexecuted chains (LITERAL
  // back end, ReLU-only proofs: no MSM jobs) would otherwise be zipped into the next proof's launches
  if (!tr->queue.jobs.empty() || !tr->rec.empty()) rc = tr->flush();
  tr->n_fr = tr->n_g1 = tr->n_g1_final = 0;
  return rc;
}
int zk_transcript_flush(zk_transcript* tr) {
  if (!tr) return ZK_ERR_ARG;
  tr->ctx->join_lanes();
  return tr->flush();
}
int zk_transcript_seal(zk_transcript* tr, uint64_t* out8_d) {
  if (!tr) return ZK_ERR_ARG;
  if (out8_d && (((uintptr_t)out8_d) & 7u)) return tr->ctx->fail(ZK_ERR_ARG, "misaligned pointer: out8_d");
  return tr->seal(out8_d);
}
int zk_transcript_seal_pair(zk_transcript* a, uint64_t* a_out8_d, zk_transcript* b, uint64_t* b_out8_d) {
  if (!a || !b) return ZK_ERR_ARG;
  if ((a_out8_d && (((uintptr_t)a_out8_d) & 7u)) || (b_out8_d && (((uintptr_t)b_out8_d) & 7u))) return a->ctx->fail(ZK_ERR_ARG, "misaligned pointer: out8_d");
  return a->seal_pair(b, a_out8_d, b_out8_d);
}
void* zk_transcript_stream(zk_transcript* tr) { return tr ? (void*)(tr->sealed_stream ? tr->sealed_stream : tr->ctx->main_stream) : nullptr; }
// the numbers of values appended so far: host-side counters, known when a prove call returns (no flush, no wait)
int zk_transcript_sizes(const zk_transcript* tr, size_t* n_fr, size_t* n_g1) {
  if (!tr) return ZK_ERR_ARG;
  if (n_fr) *n_fr = tr->n_fr;
  if (n_g1) *n_g1 = tr->n_g1;
  return ZK_OK;
}
int zk_transcript_counts(zk_transcript* tr, size_t* n_fr, size_t* n_g1) {
  if (!tr) return ZK_ERR_ARG;
  tr->ctx->join_lanes();
  ZK_TRY(tr->flush());
  ZK_HIP(tr->ctx, hipStreamSynchronize(tr->ctx->stream));
  if (n_fr) *n_fr = tr->n_fr;
  if (n_g1) *n_g1 = tr->n_g1;
  return ZK_OK;
}
int zk_transcript_read(zk_transcript* tr, zk_fr* fr_h, size_t fr_cap, zk_g1j* g1_h, size_t g1_cap) {
  if (!tr) return ZK_ERR_ARG;
  zk_ctx* ctx = tr->ctx;
  ctx->join_lanes();
  if (fr_cap < tr->n_fr || g1_cap < tr->n_g1) return ctx->fail(ZK_ERR_ARG, "transcript buffers too small");
  ZK_TRY(tr->flush());
  if (tr->n_fr) ZK_TRY(zk_memcpy_d2h(ctx, fr_h, tr->fr_d, tr->n_fr * sizeof(Fr)));
  if (tr->n_g1) ZK_TRY(zk_memcpy_d2h(ctx, g1_h, tr->g1_d, tr->n_g1 * sizeof(G1J)));
  return zk_ctx_sync(ctx);
}
// A SEALED transcript read without synchronising anything else: the host waits for this transcript's own G1 phase (which waited for its Fr phase), and
// the values travel on a private copy stream.  zk_transcript_read / read_affine join every worker stream and synchronise the context -- with several
// proofs in flight that is a wait for ALL of them (zkdl_amd/shard.py's software pipeline gained nothing from it).  G1 values are already canonical
// (x, y, Z = ONE; infinity {0, ONE, 0}) behind a sealed phase: the affine form is their first two coordinates.
int zk_transcript_read_sealed(zk_transcript* tr, zk_fr* fr_h, size_t fr_cap, zk_g1a* g1_h, uint8_t* inf_h, size_t g1_cap) {
  if (!tr) return ZK_ERR_ARG;
  zk_ctx* ctx = tr->ctx;
  (void)hipSetDevice(ctx->device);
  if (fr_cap < tr->n_fr || g1_cap < tr->n_g1) return ctx->fail(ZK_ERR_ARG, "transcript buffers too small");
  if ((tr->n_fr && !fr_h) || (tr->n_g1 && (!g1_h || !inf_h))) return ctx->fail(ZK_ERR_ARG, "zk_transcript_read_sealed: null buffer");
  if (ctx->msm_mode != ZK_MSM_PIPPENGER) {   // the literal back end leaves Jacobian representatives: the ordinary reads (they normalise)
    std::vector<zk_g1j> scratch(tr->n_g1 ? tr->n_g1 : 1);
    ZK_TRY(zk_transcript_read(tr, fr_h, fr_cap, scratch.data(), scratch.size()));
    return tr->n_g1 ? zk_transcript_read_affine(tr, g1_h, inf_h, g1_cap) : ZK_OK;
  }
  if (tr->sealed && tr->g1_deferred) ZK_TRY(ctx->flush_g1_pending());   // its G1 phase was still waiting for partners
  if (tr->sealed && tr->sealed_ev) {
    ZK_HIP(ctx, hipEventSynchronize(tr->sealed_ev));
  } else {   // not sealed, or sealed synchronously (no G1 stream): finish it the ordinary way, then the same copies
    ctx->join_lanes();
    ZK_TRY(tr->flush());
    ZK_TRY(zk_ctx_sync(ctx));
  }
  if (!ctx->copy_stream) ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  std::vector<G1J> jac(tr->n_g1);
  if (tr->n_fr) ZK_HIP(ctx, hipMemcpyAsync(fr_h, tr->fr_d, tr->n_fr * sizeof(Fr), hipMemcpyDeviceToHost, ctx->copy_stream));
  if (tr->n_g1) ZK_HIP(ctx, hipMemcpyAsync(jac.data(), tr->g1_d, tr->n_g1 * sizeof(G1J), hipMemcpyDeviceToHost, ctx->copy_stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
  for (size_t i = 0; i < tr->n_g1; i++) {
    const bool inf = l_is_zero(jac[i].z);
    G1A a;
    a.x = inf ? fp_zero() : jac[i].x;
    a.y = inf ? fp_zero() : jac[i].y;
    memcpy(&g1_h[i], &a, sizeof a);
    inf_h[i] = inf ? 1 : 0;
  }
  return ZK_OK;
}
int zk_transcript_read_affine(zk_transcript* tr, zk_g1a* g1_h, uint8_t* inf_h, size_t cap) {
  if (!tr) return ZK_ERR_ARG;
  zk_ctx* ctx = tr->ctx;
  ctx->join_lanes();
  if (cap < tr->n_g1) return ctx->fail(ZK_ERR_ARG, "transcript buffers too small");
  ZK_TRY(tr->flush());
  if (!tr->n_g1) return zk_ctx_sync(ctx);
  ZK_ALLOC(ctx, aff, G1A, tr->n_g1);
  ZK_ALLOC(ctx, inf, uint8_t, tr->n_g1);
  ZK_TRY(g1_to_affine_dev(ctx, tr->g1_d, (uint32_t)tr->n_g1, aff.get(), inf.get()));
  ZK_TRY(zk_memcpy_d2h(ctx, g1_h, aff.get(), tr->n_g1 * sizeof(G1A)));
  return zk_memcpy_d2h(ctx, inf_h, inf.get(), tr->n_g1);
}
int zk_transcript_serialize(zk_transcript* tr, void* buf_h, size_t cap, size_t* len) {
  if (!tr) return ZK_ERR_ARG;
  zk_ctx* ctx = tr->ctx;
  CHECK_HOST(ctx, len);
  ZK_TRY(tr->flush());
  const size_t need = 8 + 16 + tr->n_fr * sizeof(Fr) + tr->n_g1 * (sizeof(G1A) + 1);
  *len = need;
  if (!buf_h || cap < need) return ctx->fail(ZK_ERR_ARG, "transcript buffer too small");
  uint8_t* p = static_cast<uint8_t*>(buf_h);
  memcpy(p, "ZKDLTR01", 8);
  const uint64_t nf = tr->n_fr, ng = tr->n_g1;
  memcpy(p + 8, &nf, 8);
  memcpy(p + 16, &ng, 8);
  p += 24;
  if (tr->n_fr) ZK_TRY(zk_memcpy_d2h(ctx, p, tr->fr_d, tr->n_fr * sizeof(Fr)));
  p += tr->n_fr * sizeof(Fr);
  if (tr->n_g1) {
    ZK_ALLOC(ctx, aff, G1A, tr->n_g1);
    ZK_ALLOC(ctx, inf, uint8_t, tr->n_g1);
    ZK_TRY(g1_to_affine_dev(ctx, tr->g1_d, (uint32_t)tr->n_g1, aff.get(), inf.get()));
    ZK_TRY(zk_memcpy_d2h(ctx, p, aff.get(), tr->n_g1 * sizeof(G1A)));
    ZK_TRY(zk_memcpy_d2h(ctx, p + tr->n_g1 * sizeof(G1A), inf.get(), tr->n_g1));
  }
  return zk_ctx_sync(ctx);
}
int zk_transcript_digest(zk_transcript* tr, zk_fr* out_h) {
  if (!tr) return ZK_ERR_ARG;
  zk_ctx* ctx = tr->ctx;
  ctx->join_lanes();
  CHECK_HOST(ctx, out_h);
  ZK_TRY(tr->flush());
  ZK_ALLOC(ctx, d, Fr, 1);
  ZK_TRY(transcript_digest_dev(ctx, tr->fr_d, tr->n_fr, tr->g1_d, tr->n_g1, d.get(), nullptr));
  return zk_memcpy_d2h(ctx, out_h, d.get(), sizeof(Fr));
}
int zk_transcript_digest_dev(zk_transcript* tr, uint64_t* out8_d) {
  if (!tr) return ZK_ERR_ARG;
  zk_ctx* ctx = tr->ctx;
  ctx->join_lanes();
  if (!out8_d || (((uintptr_t)out8_d) & 7u)) return ctx->fail(ZK_ERR_ARG, "null or misaligned pointer: out8_d");
  ZK_TRY(tr->flush());
  return transcript_digest_dev(ctx, tr->fr_d, tr->n_fr, tr->g1_d, tr->n_g1, nullptr, out8_d);
}

}  // extern "C"
