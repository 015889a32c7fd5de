// zkdl_amd/csrc/layers.hip -- zkFC / zkReLU (forward + prove) and the proof transcript for gfx950
// (reference: zkfc.cu, zkrelu.cu, demo.cu:23-38,124-138).
#include "layers.hpp"

#include "dev_util.hpp"

namespace zk {

// ------------------------------------------------------------------ quantiser (zkfc.cu:63-88)
// v = round-half-away(|x * 2^16|) as uint32; negative -> r - v; zero padding; NOT Montgomery.
__global__ void __launch_bounds__(256) k_float_to_fr(const float* __restrict__ fs, Fr* __restrict__ frs, uint32_t rows, uint32_t rows_p, uint32_t cols, uint32_t cols_p) {
  const uint64_t total = (uint64_t)rows_p * cols_p;
  for (uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t d0 = (uint32_t)(tid / cols_p), d1 = (uint32_t)(tid - (uint64_t)d0 * cols_p);
    Fr r = fr_zero();
    if (d0 < rows && d1 < cols) {
      float x = fs[(uint64_t)d0 * cols + d1] * 65536.0f;
      float ax = roundf(fabsf(x));
      uint32_t v = ax >= 4294967296.0f ? 0xffffffffu : (uint32_t)ax;  // cast is UB in the reference beyond 2^32 (App. D-8): clamp
      r.v[0] = v;
      if (copysignf(1.0f, x) < 0.0f) r = fr_sub(fr_zero(), r);
    }
    frs[tid] = r;
  }
}
int float_to_fr_dev(zk_ctx* ctx, const float* fs, Fr* out, uint32_t rows, uint32_t rows_p, uint32_t cols, uint32_t cols_p) {
  if (rows > rows_p || cols > cols_p) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  uint64_t total = (uint64_t)rows_p * cols_p;
  if (total == 0) return ZK_OK;
  ZK_LAUNCH(ctx, "float_to_fr", 32ull * total + 4ull * rows * cols, k_float_to_fr, dim3(stream_grid(total)), dim3(256), 0, fs, out, rows, rows_p, cols, cols_p);
  return ZK_OK;
}

// ------------------------------------------------------------------ field matmul (zkfc.cu:6-47)
// Z[B x out] = X[B x in] * W[in x out] over Fr (Montgomery product + modular add).  255-bit modular
// products have no MFMA mapping; this is integer-VALU work.  16x16 output tile per workgroup, the
// two operand tiles staged through LDS (2 x 8 KiB).
#define ZK_TILE 16
__global__ void __launch_bounds__(256) k_fr_matmul(const Fr* __restrict__ A, const Fr* __restrict__ B, Fr* __restrict__ C, uint32_t rowsA, uint32_t colsA, uint32_t colsB) {
  __shared__ Fr At[ZK_TILE][ZK_TILE];
  __shared__ Fr Bt[ZK_TILE][ZK_TILE];
  const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
  const uint32_t row = blockIdx.y * ZK_TILE + ty, col = blockIdx.x * ZK_TILE + tx;
  Fr sum = fr_zero();
  for (uint32_t t0 = 0; t0 < colsA; t0 += ZK_TILE) {
    At[ty][tx] = (row < rowsA && t0 + tx < colsA) ? A[(uint64_t)row * colsA + t0 + tx] : fr_zero();
    Bt[ty][tx] = (t0 + ty < colsA && col < colsB) ? B[(uint64_t)(t0 + ty) * colsB + col] : fr_zero();
    __syncthreads();
#pragma unroll 1
    for (uint32_t kk = 0; kk < ZK_TILE; kk++) sum = fr_add(sum, fr_mul(At[ty][kk], Bt[kk][tx]));
    __syncthreads();
  }
  if (row < rowsA && col < colsB) C[(uint64_t)row * colsB + col] = sum;
}

// ---- integer forward pass of a quantised layer.  Activations are small integers (|x| < 2^33: 16-bit-scaled
// inputs, 32-bit rescaled ReLU outputs) and weights fit int16, so  Z = X W  is exact in int64
// (2^33 * 2^15 * in <= 2^62 for in <= 2^14) and equals the field product the reference computes with one
// Montgomery multiplication per term (zkfc.cu:6-47): ~5 instructions per term instead of ~350.
__global__ void __launch_bounds__(256) k_fr_to_i64(const Fr* __restrict__ x, int64_t* __restrict__ out, uint32_t n, uint32_t* __restrict__ overflow) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    Fr v = fr_unmont(x[i]);
    Fr nv = fr_neg(v);
    uint32_t hv = 0, hn = 0;
#pragma unroll
    for (int l = 2; l < 8; l++) {
      hv |= v.v[l];
      hn |= nv.v[l];
    }
    int64_t r = 0;
    if (hv == 0 && v.v[1] < 2u) r = (int64_t)(((uint64_t)v.v[1] << 32) | v.v[0]);
    else if (hn == 0 && nv.v[1] < 2u) r = -(int64_t)(((uint64_t)nv.v[1] << 32) | nv.v[0]);
    else atomicOr(overflow, 1u);
    out[i] = r;
  }
}
__global__ void __launch_bounds__(256) k_i64_to_fr(const int64_t* __restrict__ v, Fr* __restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int64_t x = v[i];
    const uint64_t mag = x < 0 ? (uint64_t)(-x) : (uint64_t)x;
    Fr m = fr_zero();
    m.v[0] = (uint32_t)mag;
    m.v[1] = (uint32_t)(mag >> 32);
    m = fr_mont(m);
    out[i] = x < 0 ? fr_neg(m) : m;
  }
}
// C[rowsA x colsB] = A[rowsA x colsA] (int64) * B[colsA x colsB] (int16); 64 x 64 tile per workgroup, 4 x 4 per lane
#define ZK_GT 64
#define ZK_GK 16
__global__ void __launch_bounds__(256) k_gemm_i64_i16(const int64_t* __restrict__ A, const int16_t* __restrict__ B, int64_t* __restrict__ C, uint32_t rowsA, uint32_t colsA,
                                                       uint32_t colsB) {
  __shared__ int64_t At[ZK_GK][ZK_GT + 1];
  __shared__ int32_t Bt[ZK_GK][ZK_GT];
  const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
  const uint32_t row0 = blockIdx.y * ZK_GT, col0 = blockIdx.x * ZK_GT;
  int64_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0;
  for (uint32_t k0 = 0; k0 < colsA; k0 += ZK_GK) {
    for (uint32_t e = threadIdx.x; e < ZK_GT * ZK_GK; e += 256) {
      uint32_t r = e / ZK_GK, k = e % ZK_GK;  // A tile: consecutive lanes walk k (contiguous in memory)
      At[k][r] = (row0 + r < rowsA && k0 + k < colsA) ? A[(uint64_t)(row0 + r) * colsA + k0 + k] : 0;
      uint32_t kb = e / ZK_GT, cb = e % ZK_GT;  // B tile: consecutive lanes walk the columns
      Bt[kb][cb] = (k0 + kb < colsA && col0 + cb < colsB) ? (int32_t)B[(uint64_t)(k0 + kb) * colsB + col0 + cb] : 0;
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t k = 0; k < ZK_GK; k++) {
      int64_t a[4];
      int64_t b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = At[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = (int64_t)Bt[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t r = row0 + ty * 4 + i, cc = col0 + tx * 4 + j;
      if (r < rowsA && cc < colsB) C[(uint64_t)r * colsB + cc] = acc[i][j];
    }
}

// ------------------------------------------------------------------ ReLU (zkrelu.cu:3-41)
// Writes the reference's aux tensors (one Fr per bit) AND the same bits packed (32 magnitude bits
// per element in mag_packed, 16 remainder bits in rem_packed) for the bit-packed prover path.
__global__ void __launch_bounds__(256) k_relu(const Fr* __restrict__ X, Fr* __restrict__ Z, Fr* __restrict__ sign, uint32_t* __restrict__ mag_packed,
                                               uint16_t* __restrict__ rem_packed, uint8_t* __restrict__ sign_packed, uint32_t n, uint32_t* __restrict__ out_of_range,
                                               uint32_t leaky, uint32_t shift, Fr* __restrict__ low, Fr* __restrict__ nsign) {
  const Fr one = fr_one(), zero = fr_zero();
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
    Fr x = fr_unmont(X[g]);
    // positive window: x <= 2^47 - 1 ; negative window: x >= r - 2^47
    Fr pos_max = fr_zero();
    pos_max.v[0] = 4294967295u;
    pos_max.v[1] = 32767u;
    Fr neg_min;
    neg_min.v[0] = 1u; neg_min.v[1] = 4294934527u; neg_min.v[2] = 4294859774u; neg_min.v[3] = 1404937218u;
    neg_min.v[4] = 161601541u; neg_min.v[5] = 859428872u; neg_min.v[6] = 698187080u; neg_min.v[7] = 1944954707u;
    uint64_t mag = 0;
    bool positive = false;
    if (l_gte(pos_max, x)) {
      positive = true;
      mag = (uint64_t)x.v[0] | ((uint64_t)x.v[1] << 32);
    } else if (l_gte(x, neg_min)) {
      Fr two47 = fr_zero();
      two47.v[1] = 32768u;
      Fr y = fr_add(x, two47);
      mag = (uint64_t)y.v[0] | ((uint64_t)y.v[1] << 32);
    } else {
      atomicAdd(out_of_range, 1u);  // undefined in the reference (mag uninitialised); deterministic here: mag = 0, sign = 0
    }
    const bool rem_sign = (mag & 32768ull) != 0;
    const uint32_t rem_mag = (uint32_t)(mag & 32767ull);
    const int32_t rem = rem_sign ? ((int32_t)rem_mag - (1 << 15)) : (int32_t)rem_mag;
    const uint32_t mag_rescaled = (uint32_t)((mag - (uint64_t)(int64_t)rem) >> 16);
    sign[g] = positive ? one : zero;
    mag_packed[g] = mag_rescaled;
    rem_packed[g] = (uint16_t)(rem_mag | (rem_sign ? 0x8000u : 0u));
    sign_packed[g] = positive ? 1 : 0;
    Fr mr = fr_zero();
    mr.v[0] = mag_rescaled;
    Z[g] = positive ? fr_mont(mr) : zero;  // mont(mag_rescaled) * sign with sign in {ONE, 0}
    if (leaky) {  // extension: for a negative input mag_rescaled = 2^31 + q (offset encoding of q = round(x / 2^16) < 0, zkrelu.cu:19-22), so
                  // floor(q / 2^shift) = (mag_rescaled >> shift) - 2^(31 - shift): magnitude bits shift .. 31 minus a constant
      Fr lo = fr_zero();
      lo.v[0] = shift < 32u ? (mag_rescaled >> shift) : 0u;
      lo = fr_mont(lo);
      low[g] = lo;
      nsign[g] = positive ? zero : one;
      if (!positive) {
        Fr cst = fr_zero();
        cst.v[0] = shift < 32u ? (1u << (31u - shift)) : 0u;
        Z[g] = fr_sub(lo, fr_mont(cst));
      }
    }
  }
}

// the reference's one-Fr-per-bit aux tensors (zkrelu.cu:29-37) from the packed words: out[g*NB + i] = bit i of word g
template <int NB, class WordT>
__global__ void __launch_bounds__(256) k_relu_unpack(const WordT* __restrict__ packed, Fr* __restrict__ out, uint64_t total) {
  const Fr one = fr_one(), zero = fr_zero();
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t word = packed[t / NB];
    out[t] = ((word >> (uint32_t)(t % NB)) & 1u) ? one : zero;
  }
}

// ------------------------------------------------------------------ transcript digest
// sum of all Fr entries + every G1 entry's affine x, y folded in as 224-bit chunks (each < r)
__global__ void __launch_bounds__(256) k_digest(const Fr* __restrict__ fr, uint32_t n_fr, const G1A* __restrict__ aff, const uint8_t* __restrict__ inf, uint32_t n_g1,
                                                 Fr* __restrict__ out_fr, uint64_t* __restrict__ out8) {
  __shared__ Fr lds[4];
  Fr acc = fr_zero();
  for (uint32_t i = threadIdx.x; i < n_fr; i += blockDim.x) acc = fr_add(acc, fr[i]);
  for (uint32_t i = threadIdx.x; i < 4 * n_g1; i += blockDim.x) {
    uint32_t p = i >> 2, cidx = i & 3u;
    if (inf[p]) continue;
    const uint32_t* src = cidx < 2 ? aff[p].x.v : aff[p].y.v;
    uint32_t lo = (cidx & 1u) ? 7u : 0u, cnt = (cidx & 1u) ? 5u : 7u;
    Fr e = fr_zero();
#pragma unroll
    for (uint32_t j = 0; j < 7; j++) e.v[j] = j < cnt ? src[lo + j] : 0u;
    acc = fr_add(acc, e);
  }
  acc = block_sum_fr(acc, lds);
  if (threadIdx.x == 0) {
    if (out_fr) *out_fr = acc;
    if (out8)
      for (int j = 0; j < 8; j++) out8[j] = acc.v[j];
  }
}
int transcript_digest_dev(zk_ctx* ctx, const Fr* fr, size_t n_fr, const G1J* g1, size_t n_g1, Fr* out_fr_d, uint64_t* out8_d) {
  ZK_ALLOC(ctx, aff, G1A, n_g1 ? n_g1 : 1);
  ZK_ALLOC(ctx, inf, uint8_t, n_g1 ? n_g1 : 1);
  if (n_g1) ZK_TRY(g1_to_affine_dev(ctx, g1, (uint32_t)n_g1, aff.get(), inf.get()));
  ZK_LAUNCH(ctx, "transcript_digest", 32ull * n_fr + 96ull * n_g1, k_digest, dim3(1), dim3(256), 0, fr, (uint32_t)n_fr, (const G1A*)aff.get(), (const uint8_t*)inf.get(),
            (uint32_t)n_g1, out_fr_d, out8_d);
  return ZK_OK;
}

}  // namespace zk

using namespace zk;

// ------------------------------------------------------------------ transcript
int zk_transcript::ensure(size_t add_fr, size_t add_g1) {
  unseal();
  if (n_fr + add_fr > cap_fr || n_g1 + add_g1 > cap_g1) {
    ZK_TRY(rec_execute(ctx, &rec));  // recorded chains hold pointers into the old buffers
    ctx->join_lanes();               // lanes may still be writing them
  }
  if (n_fr + add_fr > cap_fr) {
    size_t want = n_fr + add_fr;
    (void)want;
    size_t nc = cap_fr ? cap_fr : 8192;
    while (nc < n_fr + add_fr) nc *= 2;
    Fr* p = nullptr;
    ZK_HIP(ctx, dev_alloc(ctx, (void**)&p, nc * sizeof(Fr)));
    if (n_fr) ZK_TRY(dev_copy(ctx, p, fr_d, n_fr * sizeof(Fr)));
    dev_free(ctx, fr_d);
    fr_d = p;
    cap_fr = nc;
  }
  if (n_g1 + add_g1 > cap_g1) {
    ZK_TRY(flush());  // queued MSMs write into g1_d: settle them before the buffer moves
    size_t nc = cap_g1 ? cap_g1 : 1024;
    while (nc < n_g1 + add_g1) nc *= 2;
    G1J* p = nullptr;
    ZK_HIP(ctx, dev_alloc(ctx, (void**)&p, nc * sizeof(G1J)));
    if (n_g1) ZK_TRY(dev_copy(ctx, p, g1_d, n_g1 * sizeof(G1J)));
    dev_free(ctx, g1_d);
    g1_d = p;
    cap_g1 = nc;
  }
  return ZK_OK;
}

void zk_transcript::unseal() {
  if (!sealed) return;
  if (g1_deferred) (void)ctx->flush_g1_pending();  // its G1 phase was still waiting for partners
  (void)hipStreamWaitEvent(ctx->main_stream, sealed_ev, 0);
  sealed = false;
  deferred_keep.clear();  // owned by worker-lane pools; lanes fork from the main stream, i.e. behind this wait
  rec_release_held(ctx, &rec);  // temporaries of an unjoined Fr phase: the sealed G1 phase waited for every group stream
}

int zk_transcript::flush() {
  unseal();
  ZK_TRY(rec_execute(ctx, &rec));  // the layers' Fr work, zipped across layers (launch.hpp)
  ctx->join_lanes();
  ZK_TRY(msm_flush(ctx, &queue));
  if (n_g1 > n_g1_final && ctx->msm_mode == ZK_MSM_PIPPENGER) ZK_TRY(g1_canonicalise_dev(ctx, g1_d + n_g1_final, (uint32_t)(n_g1 - n_g1_final)));
  n_g1_final = n_g1;
  return ZK_OK;
}

// The G1 phase of a proof (MSM batch -> canonical outputs -> digest) as one asynchronous unit on the G1
// stream.  The scalar-prep kernels of the queued MSMs ran on worker lanes, so the G1 stream first waits on
// the main stream and on every dirty lane; the main stream does not wait for the G1 stream, so the next
// proof's Fr phase (worker lanes) runs under this proof's MSM tail.  Scratch comes from the G1 stream's own
// pool; the prepared scalars (worker-lane pools) are held until unseal().
static int g1_phase_run(zk_ctx* ctx, const std::vector<zk_transcript*>& trs);
int zk_transcript::seal(uint64_t* out8_d) {
  if (ctx->g1_lanes.empty() || !ctx->lanes_enabled || ctx->stream != ctx->main_stream) {
    sealed_stream = nullptr;
    ZK_TRY(flush());  // no G1 stream: same result, synchronously ordered on the main stream
    return out8_d ? transcript_digest_dev(ctx, fr_d, n_fr, g1_d, n_g1, nullptr, out8_d) : ZK_OK;
  }
  unseal();
  // Fr phase on the main stream + group lanes; the G1 stream waits for it below.  With "pipeline_fr" the groups are NOT joined on
  // the main stream: the next proof's groups queue behind this proof's, stream by stream, so a short group no longer idles until
  // the longest one is done (the temporaries then stay with the recorder until unseal()).
  std::vector<hipEvent_t> early;
  (void)hipEventRecord(ctx->pre_ev, ctx->main_stream);  // everything the main stream held before this proof's Fr phase (buffer moves of ensure())
  int rc = rec_execute(ctx, &rec, /*join=*/!ctx->pipeline_fr, &early);
  if (rc != ZK_OK) {
    for (hipEvent_t e : early) ctx->event_pool.push_back(e);
    return rc;
  }
  // the G1 phase may start behind the early chains only if EVERY queued MSM / derive job took its scalars from one of them
  return submit_g1(out8_d, &early, !early.empty() && !queue.late_inputs);
}

// after the Fr phase has been submitted: run the G1 phase now, or -- "g1_batch" > 1 -- leave it with the context until that many
// sealed proofs are waiting (or until anyone needs this one).  The events are taken over.
int zk_transcript::submit_g1(uint64_t* out8_d, const std::vector<hipEvent_t>* early, bool early_ok) {
  pending_out8 = out8_d;
  if (early) pending_early = *early;
  else pending_early.clear();
  pending_early_ok = early_ok;
  if (ctx->g1_batch > 1) {
    g1_deferred = true;
    sealed = true;
    ctx->g1_pending.push_back(this);
    return ctx->g1_pending.size() >= ctx->g1_batch ? ctx->flush_g1_pending() : ZK_OK;
  }
  return g1_phase_run(ctx, {this});
}

// Two proofs recorded on two transcripts, submitted together: their Fr chains are zipped into the same launches (a proof's Fr phase
// is ~80 mostly latency-bound launches; two proofs share them), then each proof's G1 phase goes to its own G1 stream as in seal().
// Same values as two seal() calls.  The temporaries of both Fr phases stay with THIS transcript's recorder until it is unsealed.
int zk_transcript::seal_pair(zk_transcript* other, uint64_t* out8_d, uint64_t* other_out8_d) {
  if (!other || other == this || other->ctx != ctx) return ctx->fail(ZK_ERR_ARG, "seal_pair: needs two transcripts of one context");
  if (ctx->g1_lanes.empty() || !ctx->lanes_enabled || ctx->stream != ctx->main_stream) {
    ZK_TRY(seal(out8_d));
    return other->seal(other_out8_d);
  }
  unseal();
  other->unseal();
  ZK_TRY(rec_merge(ctx, &rec, &other->rec));
  (void)hipEventRecord(ctx->pre_ev, ctx->main_stream);
  std::vector<hipEvent_t> early, early2;
  int rce = rec_execute(ctx, &rec, /*join=*/!ctx->pipeline_fr, &early);
  const bool merged = other->rec.chains.empty();
  if (rce == ZK_OK) rce = rec_execute(ctx, &other->rec, /*join=*/!ctx->pipeline_fr, &early2);  // only when there was nothing to merge into
  if (rce != ZK_OK) {
    for (hipEvent_t e : early) ctx->event_pool.push_back(e);
    for (hipEvent_t e : early2) ctx->event_pool.push_back(e);
    return rce;
  }
  // both G1 phases are always submitted: a failure of the first must not leave `other` with its Fr phase executed, its MSM queue
  // intact and no seal (the first error is what the caller sees; a failed transcript must be reset before reuse).
  // Merged execution: the events behind the zipped opening chains cover both proofs; this transcript carries them.
  const bool ok1 = !early.empty() && !queue.late_inputs && !other->queue.late_inputs;
  const bool same_batch = ctx->g1_batch > 1 && ctx->g1_pending.size() + 2 <= ctx->g1_batch;  // both land in one G1 batch: its waits cover both
  const int rc = submit_g1(out8_d, &early, ok1);
  const int rc2 = other->submit_g1(other_out8_d, &early2, merged ? (ok1 && same_batch) : (!early2.empty() && !other->queue.late_inputs));
  return rc != ZK_OK ? rc : rc2;
}

// The G1 phase of one or several sealed proofs as ONE batch on a G1 stream: all their MSM jobs in one sort / accumulate / combine /
// reduce sequence, all their derived round commitments in one launch pair, one canonicalisation, then a digest each.  The chip-filling
// accumulation scales with the number of proofs; everything behind it is a handful of waves per proof running dependent chains of
// field products for ~3 ms (standalone: combine 0.36, row/column sums 0.22, weighted sums 0.33, derivation 1.8, canonical form 0.5),
// whose latency is the same for one proof or for several -- so batching N proofs divides that part of the chain by N ("g1_batch").
//
// pending_early (per transcript, optional): events behind the opening chains' scalar preparation (rec_execute).  The MSM batch depends
// on nothing else of the Fr phase, so with them the G1 stream starts while the proofs' sumchecks are still running and only the digests
// -- which read every Fr value -- wait for the whole Fr phase.  Without them (nothing recorded, immediate mode) the G1 stream waits for
// the Fr phase first.
static int g1_phase_run(zk_ctx* ctx, const std::vector<zk_transcript*>& trs) {
  if (trs.empty()) return ZK_OK;
  const size_t n_g1s = ctx->g1_active && ctx->g1_active < ctx->g1_lanes.size() ? ctx->g1_active : ctx->g1_lanes.size();
  zk_ctx::Lane& g = (!ctx->g1_masked && ctx->g1_full.stream) ? ctx->g1_full : ctx->g1_lanes[ctx->next_g1++ % n_g1s];
  for (zk_transcript* t : trs) {
    t->sealed_stream = g.stream;
    if (!t->sealed_ev && hipEventCreateWithFlags(&t->sealed_ev, hipEventDisableTiming) != hipSuccess) return ctx->fail(ZK_ERR_DEVICE, "hipEventCreate failed");
  }
  auto wait_fr_phase = [&](hipStream_t st) {
    (void)hipEventRecord(ctx->fork_ev, ctx->main_stream);
    (void)hipStreamWaitEvent(st, ctx->fork_ev, 0);
    for (auto& l : ctx->lanes)
      if (l.dirty) (void)hipStreamWaitEvent(st, l.done, 0);  // stays dirty: the main stream has not joined it
  };
  bool early_start = ctx->early_g1;
  for (zk_transcript* t : trs)
    if (!t->pending_early_ok) early_start = false;
  if (early_start) {
    (void)hipStreamWaitEvent(g.stream, ctx->pre_ev, 0);  // recorded at the latest seal(): main-stream order covers the earlier ones
    for (zk_transcript* t : trs)
      for (hipEvent_t e : t->pending_early) (void)hipStreamWaitEvent(g.stream, e, 0);
  } else {
    wait_fr_phase(g.stream);
  }
  ctx->stream = g.stream;
  ctx->cur_pool = &g.pool;
  int rc = ZK_OK;
  static const bool skip_g1 = experiment_on("ZKDL_EXPERIMENT_SKIP_G1");  // timing experiments only: the G1 values are garbage then
  // one MSM batch: the jobs of every proof (they must share the window width; they do unless zk_set_msm_window changed in between)
  uint32_t c = 0;
  bool same_c = true;
  size_t njobs = 0, nder = 0;
  for (zk_transcript* t : trs) {
    if (t->queue.jobs.empty()) continue;
    if (c == 0) c = t->queue.c;
    else if (t->queue.c != c) same_c = false;
    njobs += t->queue.jobs.size();
    nder += t->queue.derives.size();
  }
  if (!skip_g1 && njobs) {
    if (same_c && trs.size() > 1) {
      std::vector<MsmJobHost> jobs;
      jobs.reserve(njobs);
      for (zk_transcript* t : trs) jobs.insert(jobs.end(), t->queue.jobs.begin(), t->queue.jobs.end());
      rc = msm_run_jobs(ctx, jobs, c);
    } else {
      for (zk_transcript* t : trs)
        if (rc == ZK_OK && !t->queue.jobs.empty()) rc = msm_run_jobs(ctx, t->queue.jobs, t->queue.c);
    }
  }
  // the latency-bound tail (derived round commitments, canonicalisation, digest) moves to a tail stream when there is one
  zk_ctx::Lane* tail = ctx->tail_lanes.empty() ? nullptr : &ctx->tail_lanes[ctx->next_tail++ % ctx->tail_lanes.size()];
  hipStream_t fin = g.stream;
  if (tail) {
    (void)hipEventRecord(g.done, g.stream);
    g.dirty = true;
    (void)hipStreamWaitEvent(tail->stream, g.done, 0);
    ctx->stream = tail->stream;
    ctx->cur_pool = &tail->pool;
    fin = tail->stream;
    for (zk_transcript* t : trs) t->sealed_stream = tail->stream;
  }
  if (rc == ZK_OK && !skip_g1 && nder) {
    if (trs.size() > 1) {
      std::vector<DeriveJobHost> derives;
      derives.reserve(nder);
      for (zk_transcript* t : trs) derives.insert(derives.end(), t->queue.derives.begin(), t->queue.derives.end());
      rc = msm_run_derives(ctx, derives);
    } else {
      rc = msm_run_derives(ctx, trs[0]->queue.derives);
    }
  }
  if (rc == ZK_OK && ctx->msm_mode == ZK_MSM_PIPPENGER) {
    CanonRanges cr;
    cr.cnt = 0;
    for (zk_transcript* t : trs)
      if (t->n_g1 > t->n_g1_final) {
        cr.p[cr.cnt] = t->g1_d + t->n_g1_final;
        cr.n[cr.cnt] = (uint32_t)(t->n_g1 - t->n_g1_final);
        cr.cnt++;
      }
    if (cr.cnt == 1) rc = g1_canonicalise_dev(ctx, cr.p[0], cr.n[0]);
    else if (cr.cnt > 1) rc = g1_canonicalise_multi_dev(ctx, cr);
  }
  if (early_start) wait_fr_phase(fin);  // the digests read every Fr value; unseal() (which waits for sealed_ev) must imply the Fr phase as well
  for (zk_transcript* t : trs)
    if (rc == ZK_OK && t->pending_out8) rc = transcript_digest_dev(ctx, t->fr_d, t->n_fr, t->g1_d, t->n_g1, nullptr, t->pending_out8);
  for (zk_transcript* t : trs) (void)hipEventRecord(t->sealed_ev, fin);
  if (tail) {
    (void)hipEventRecord(tail->done, tail->stream);
    tail->dirty = true;
  } else {
    (void)hipEventRecord(g.done, g.stream);
    g.dirty = true;
  }
  ctx->stream = ctx->main_stream;
  ctx->cur_pool = &ctx->main_pool;
  for (zk_transcript* t : trs) {
    t->sealed = true;
    t->g1_deferred = false;
    t->n_g1_final = t->n_g1;
    t->queue.jobs.clear();
    t->queue.derives.clear();
    t->queue.c = 0;
    t->queue.late_inputs = false;
    for (auto& b : t->queue.keep) t->deferred_keep.push_back(std::move(b));
    t->queue.keep.clear();
    for (hipEvent_t e : t->pending_early) ctx->event_pool.push_back(e);
    t->pending_early.clear();
    t->pending_early_ok = false;
    t->pending_out8 = nullptr;
  }
  return rc;
}

int zk_transcript::g1_phase(uint64_t* out8_d, const std::vector<hipEvent_t>* early) {
  pending_out8 = out8_d;
  if (early) pending_early = *early;
  else pending_early.clear();
  pending_early_ok = early && !early->empty();
  return g1_phase_run(ctx, {this});
}

// sealed proofs whose G1 phase is waiting for partners ("g1_batch" > 1): run them now, as one batch
int zk_ctx::flush_g1_pending() {
  if (g1_pending.empty()) return ZK_OK;
  std::vector<zk_transcript*> list;
  list.swap(g1_pending);
  return g1_phase_run(this, list);
}

// ------------------------------------------------------------------ zkFC
namespace zk {

int fc_create(zk_ctx* ctx, uint32_t in, uint32_t out, const Fr* weights_d, zk_gens* gens, zk_fc** out_fc) {
  const uint32_t n_gens = gens->n;
  if (in == 0 || out == 0 || n_gens == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  uint64_t wn = (uint64_t)in * out;
  if (wn > 0xffffffffull || wn % n_gens != 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // commitment.cu:31 via zkfc.cu:102
  zk_fc* fc = new zk_fc();
  fc->ctx = ctx;
  fc->serial = ctx->layer_serial++;
  fc->in = in;
  fc->out = out;
  fc->m = (uint32_t)(wn / n_gens);
  fc->n_gens = n_gens;
  if (dev_alloc(ctx, (void**)&fc->weights, wn * sizeof(Fr)) != hipSuccess || gens_create(ctx, nullptr, fc->m, false, &fc->com) != ZK_OK) {
    zk_fc_destroy(fc);
    return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: zkFC");
  }
  ZK_TRY(dev_copy(ctx, fc->weights, weights_d, wn * sizeof(Fr)));
  int rc = commit_dev(ctx, gens, fc->weights, (uint32_t)wn, fc->com->pts);
  if (rc != ZK_OK) {
    zk_fc_destroy(fc);
    return rc;
  }
  // int16 copy for the prover's two contractions of W (setup time: one host round trip for the range flag)
  if (ctx->flag_q16 && dev_alloc(ctx, (void**)&fc->wq, wn * sizeof(int16_t)) == hipSuccess) {
    ZK_ALLOC(ctx, ovf, uint32_t, 1);
    uint32_t h_ovf = 1;
    rc = dev_zero(ctx, ovf.get(), 4);
    if (rc == ZK_OK) rc = fr_quant16_dev(ctx, fc->weights, (uint32_t)wn, fc->wq, ovf.get());
    if (rc == ZK_OK) rc = dev_to_host(ctx, &h_ovf, ovf.get(), 4);
    if (rc != ZK_OK || h_ovf) {  // not a quantised layer: the Fr path serves it
      dev_free(ctx, fc->wq);
      fc->wq = nullptr;
    }
    (void)hipGetLastError();
  }
  *out_fc = fc;
  return ZK_OK;
}

int fc_forward(zk_ctx* ctx, const zk_fc* fc, const Fr* x, uint32_t xn, Fr* z) {
  if (xn % fc->in != 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // zkfc.cu:118
  uint32_t batch = xn / fc->in;
  if (batch == 0) return ZK_OK;
  if (fc->wq && batch == 1 && fc->in <= 16384 && (fc->in & (fc->in - 1)) == 0)  // one row of arbitrary field elements: no range check, no host round trip
    return fr_vec_q16_dev(ctx, x, fc->in, fc->wq, fc->out, z);
  if (fc->wq && fc->in <= 16384) {  // quantised layer: exact int64 GEMM when the activations are small integers too
    ZK_ALLOC(ctx, xi, int64_t, xn);
    ZK_ALLOC(ctx, ovf, uint32_t, 1);
    uint32_t h_ovf = 1;
    ZK_TRY(dev_zero(ctx, ovf.get(), 4));
    ZK_LAUNCH(ctx, "fc_forward_int", 40ull * xn, k_fr_to_i64, dim3(stream_grid(xn)), dim3(256), 0, x, xi.get(), xn, ovf.get());
    ZK_TRY(dev_to_host(ctx, &h_ovf, ovf.get(), 4));
    if (!h_ovf) {
      const uint64_t zn = (uint64_t)batch * fc->out;
      ZK_ALLOC(ctx, zi, int64_t, zn);
      dim3 g((fc->out + ZK_GT - 1) / ZK_GT, (batch + ZK_GT - 1) / ZK_GT);
      ZK_LAUNCH(ctx, "fc_forward_int", 8ull * xn + 2ull * fc->in * fc->out + 8ull * zn, k_gemm_i64_i16, g, dim3(256), 0, (const int64_t*)xi.get(), (const int16_t*)fc->wq,
                zi.get(), batch, fc->in, fc->out);
      ZK_LAUNCH(ctx, "fc_forward_int", 40ull * zn, k_i64_to_fr, dim3(stream_grid(zn)), dim3(256), 0, (const int64_t*)zi.get(), z, (uint32_t)zn);
      return ZK_OK;
    }
  }
  dim3 grid((fc->out + ZK_TILE - 1) / ZK_TILE, (batch + ZK_TILE - 1) / ZK_TILE);
  uint64_t bytes = 32ull * ((uint64_t)xn + (uint64_t)fc->in * fc->out + (uint64_t)batch * fc->out);
  ZK_LAUNCH(ctx, "fc_matmul", bytes, k_fr_matmul, grid, dim3(256), 0, x, (const Fr*)fc->weights, z, batch, fc->in, fc->out);
  return ZK_OK;
}

// A prove call that fails part-way (allocation, launch) must leave the transcript as it found it: the slots it
// reserved, the chains it recorded and the MSM jobs it queued are dropped again.
struct ProveGuard {
  zk_ctx* ctx;
  zk_transcript* tr;
  size_t n_fr, n_g1, chains, jobs, keep, derives;
  ProveGuard(zk_ctx* c, zk_transcript* t)
      : ctx(c), tr(t), n_fr(t->n_fr), n_g1(t->n_g1), chains(t->rec.chains.size()), jobs(t->queue.jobs.size()), keep(t->queue.keep.size()), derives(t->queue.derives.size()) {}
  int finish(int rc) {
    if (rc == ZK_OK) return rc;
    rec_rollback(ctx, &tr->rec, chains);
    if (tr->queue.jobs.size() > jobs) tr->queue.jobs.resize(jobs);
    if (tr->queue.derives.size() > derives) tr->queue.derives.resize(derives);
    while (tr->queue.keep.size() > keep) tr->queue.keep.pop_back();
    tr->n_fr = n_fr;
    tr->n_g1 = n_g1;
    return rc;
  }
};

static int fc_prove_body(zk_ctx* ctx, zk_fc* fc, const Fr* x, uint32_t xn, const Fr* z, uint32_t zn, zk_gens* gens,
                         const zk_fr* u_bs_h, const zk_fr* u_in_h, const zk_fr* u_out_h, zk_transcript* tr);
// zkFC::prove, zkfc.cu:128-145
int fc_prove(zk_ctx* ctx, zk_fc* fc, const Fr* x, uint32_t xn, const Fr* z, uint32_t zn, zk_gens* gens,
             const zk_fr* u_bs_h, const zk_fr* u_in_h, const zk_fr* u_out_h, zk_transcript* tr) {
  ProveGuard guard(ctx, tr);
  return guard.finish(fc_prove_body(ctx, fc, x, xn, z, zn, gens, u_bs_h, u_in_h, u_out_h, tr));
}
static int fc_prove_body(zk_ctx* ctx, zk_fc* fc, const Fr* x, uint32_t xn, const Fr* z, uint32_t zn, zk_gens* gens,
                         const zk_fr* u_bs_h, const zk_fr* u_in_h, const zk_fr* u_out_h, zk_transcript* tr) {
  const uint32_t n_gens = gens->n;
  if (xn % fc->in != 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions 1");
  const uint32_t batch = xn / fc->in;
  const uint32_t k_bs = ceil_log2(batch), k_in = ceil_log2(fc->in), k_out = ceil_log2(fc->out);
  const uint32_t wn = fc->in * fc->out;
  const uint32_t k_lo = ceil_log2(n_gens), k_hi = ceil_log2(fc->m);
  if (k_out + k_in < k_hi || k_out + k_in - k_hi != k_lo) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  uint32_t asz = fr_partial_me_size(xn, k_bs, fc->in), bsz = fr_partial_me_size(wn, k_out, 1);
  if (asz != bsz) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  // reserve this call's transcript slots on the main timeline, then run the layer on a worker lane
  const size_t add_fr = 3 * k_in + 2 + 1 + 1, add_g1 = 1 + 3 * k_lo + 1;
  ZK_TRY(tr->ensure(add_fr, add_g1));
  Fr* const fr_slot = tr->fr_d + tr->n_fr;
  G1J* const g1_slot = tr->g1_d + tr->n_g1;
  tr->n_fr += add_fr;
  tr->n_g1 += add_g1;
  RecScope recording(ctx, &tr->rec, REC_GROUP_FC_A);  // batching on: this call only records its launches
  if (recording.rc != ZK_OK) return recording.rc;
  LaneScope lane(ctx, recording.on ? ~0u : fc->serial);  // declared before every temporary: they are released while the lane is current

  // all challenges of this layer in one device array: [u_out | u_in | u_bs]  (so u_out||u_in is contiguous)
  std::vector<zk_fr> ch(k_out + k_in + k_bs + 1);
  for (uint32_t i = 0; i < k_out; i++) ch[i] = u_out_h[i];
  for (uint32_t i = 0; i < k_in; i++) ch[k_out + i] = u_in_h[i];
  for (uint32_t i = 0; i < k_bs; i++) ch[k_out + k_in + i] = u_bs_h[i];
  DBuf<Fr> chd;
  ZK_TRY(upload_fr(ctx, chd, ch.data(), k_out + k_in + k_bs));
  const Fr* u_out = chd.get();
  const Fr* u_in = chd.get() + k_out;
  const Fr* u_bs = chd.get() + k_out + k_in;
  // u_Z = u_out || u_bs needs its own contiguous copy
  DBuf<Fr> uz;
  {
    std::vector<zk_fr> z_(k_out + k_bs + 1);
    for (uint32_t i = 0; i < k_out; i++) z_[i] = u_out_h[i];
    for (uint32_t i = 0; i < k_bs; i++) z_[k_out + i] = u_bs_h[i];
    ZK_TRY(upload_fr(ctx, uz, z_.data(), k_out + k_bs));
  }
  // a = X.partial_me(u_bs, in) ; b = W.partial_me(u_out, 1)
  ZK_ALLOC(ctx, a, Fr, asz);
  ZK_ALLOC(ctx, b, Fr, bsz);
  // three independent pieces (recorded as separate chains): the inner-product sumcheck, Z(u_Z), the opening
  ZK_TRY(fr_partial_me_dev(ctx, x, xn, u_bs, k_bs, fc->in, a.get()));
  if (fc->wq && k_out > 0) ZK_TRY(fr_partial_me_q16_dev(ctx, fc->wq, wn, u_out, k_out, 1, b.get()));
  else ZK_TRY(fr_partial_me_dev(ctx, fc->weights, wn, u_out, k_out, 1, b.get()));
  ZK_TRY(sumcheck_ip_dev(ctx, a.get(), b.get(), asz, u_in, k_in, fr_slot));
  // Z(u_Z)
  rec_split(ctx, REC_GROUP_FC_A);
  ZK_TRY(fr_me_dev(ctx, z, zn, uz.get(), k_out + k_bs, fr_slot + 3 * k_in + 2));
  // generators.open(weights, com, u_out || u_in)
  rec_split(ctx, REC_GROUP_FC_OPEN, /*early=*/true);  // all the G1 phase needs from the Fr phase: it may start behind these launches
  ZK_TRY(open_dev(ctx, gens, fc->weights, wn, fc->com, u_out, k_out + k_in, g1_slot, g1_slot + 1, fr_slot + 3 * k_in + 3, &tr->queue, fc->wq));
  return ZK_OK;
}

// ------------------------------------------------------------------ zkReLU
int relu_forward(zk_ctx* ctx, zk_relu* relu, const Fr* x, uint32_t n, Fr* z) {
  if (n == 0) return ZK_OK;
  if (relu->cap < n) {  // reset_ptrs, zkrelu.cu:54-63
    relu->free_all();
    bool ok = dev_alloc(ctx, (void**)&relu->sign, (size_t)n * sizeof(Fr)) == hipSuccess && dev_alloc(ctx, (void**)&relu->mag_packed, (size_t)n * 4) == hipSuccess &&
              dev_alloc(ctx, (void**)&relu->rem_packed, (size_t)n * 2) == hipSuccess && dev_alloc(ctx, (void**)&relu->sign_packed, (size_t)n) == hipSuccess &&
              dev_alloc(ctx, (void**)&relu->oor, 16) == hipSuccess;
    if (!ok) {
      relu->free_all();
      return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: zkReLU aux tensors");
    }
    relu->cap = n;
  }
  if (relu->act_kind == 1 && !relu->low) {
    if (dev_alloc(ctx, (void**)&relu->low, (size_t)relu->cap * sizeof(Fr)) != hipSuccess || dev_alloc(ctx, (void**)&relu->nsign, (size_t)relu->cap * sizeof(Fr)) != hipSuccess)
      return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: leaky aux tensors");
  }
  relu->n = n;
  relu->bins_valid = false;  // mag_bin / rem_bin (48 field elements per input) are materialised on demand
  ZK_TRY(dev_zero(ctx, relu->oor, 4));
  ZK_LAUNCH(ctx, "relu_forward", 32ull * 3 * n, k_relu, dim3(stream_grid(n)), dim3(256), 0, x, z, relu->sign, relu->mag_packed, relu->rem_packed, relu->sign_packed, n, relu->oor,
            relu->act_kind == 1 ? 1u : 0u, relu->act_shift, relu->low, relu->nsign);
  uint32_t oor = 0;
  ZK_TRY(dev_to_host(ctx, &oor, relu->oor, 4));
  if (oor) return ctx->fail(ZK_ERR_RANGE, "zkReLU: " + std::to_string(oor) + " input(s) outside the +-2^47 window (undefined in the reference)");
  return ZK_OK;
}

// mag_bin (32 N) and rem_bin (16 N) as the reference stores them: built from the packed words the first time
// something asks for them (the unpacked prover path, zk_relu_mag_bin / zk_relu_rem_bin)
int relu_materialise_bins(zk_ctx* ctx, zk_relu* relu) {
  if (relu->bins_valid || relu->n == 0) return ZK_OK;
  const uint32_t n = relu->n;
  if (relu->bins_cap < n) {
    dev_free(ctx, relu->mag_bin);
    dev_free(ctx, relu->rem_bin);
    relu->mag_bin = relu->rem_bin = nullptr;
    relu->bins_cap = 0;
    if (dev_alloc(ctx, (void**)&relu->mag_bin, (size_t)n * 32 * sizeof(Fr)) != hipSuccess || dev_alloc(ctx, (void**)&relu->rem_bin, (size_t)n * 16 * sizeof(Fr)) != hipSuccess)
      return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: zkReLU bit tensors");
    relu->bins_cap = n;
  }
  ZK_LAUNCH(ctx, "relu_forward", 32ull * 32 * n, (k_relu_unpack<32, uint32_t>), dim3(stream_grid((uint64_t)n * 32)), dim3(256), 0, (const uint32_t*)relu->mag_packed, relu->mag_bin, (uint64_t)n * 32);
  ZK_LAUNCH(ctx, "relu_forward", 32ull * 16 * n, (k_relu_unpack<16, uint16_t>), dim3(stream_grid((uint64_t)n * 16)), dim3(256), 0, (const uint16_t*)relu->rem_packed, relu->rem_bin, (uint64_t)n * 16);
  relu->bins_valid = true;
  return ZK_OK;
}

static int relu_prove_body(zk_ctx* ctx, const zk_relu* relu, const Fr* x, const Fr* z, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h,
                           const zk_fr* v_r_h, const zk_fr* u_rec_h, const zk_fr* u_hp_h, const zk_fr* v_hp_h, zk_transcript* tr);
// zkReLU::prove, zkrelu.cu:79-100 (Q = 32, R = 16)
int relu_prove(zk_ctx* ctx, const zk_relu* relu, const Fr* x, const Fr* z, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h,
               const zk_fr* v_r_h, const zk_fr* u_rec_h, const zk_fr* u_hp_h, const zk_fr* v_hp_h, zk_transcript* tr) {
  ProveGuard guard(ctx, tr);
  return guard.finish(relu_prove_body(ctx, relu, x, z, n, u_z_h, v_z_h, u_r_h, v_r_h, u_rec_h, u_hp_h, v_hp_h, tr));
}
static int relu_prove_body(zk_ctx* ctx, const zk_relu* relu, const Fr* x, const Fr* z, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h,
                           const zk_fr* v_r_h, const zk_fr* u_rec_h, const zk_fr* u_hp_h, const zk_fr* v_hp_h, zk_transcript* tr) {
  (void)z;
  if (relu->n != n || n == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // zkrelu.cu:81
  if ((uint64_t)n * 32 > 0xffffffffull) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t k = ceil_log2(n);
  const size_t add_fr = 3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 3 * k + 2;
  if (!ctx->relu_packed) ZK_TRY(relu_materialise_bins(ctx, const_cast<zk_relu*>(relu)));  // before recording starts
  ZK_TRY(tr->ensure(add_fr, 0));
  Fr* out = tr->fr_d + tr->n_fr;
  tr->n_fr += add_fr;
  RecScope recording(ctx, &tr->rec, REC_GROUP_RELU_MAG);
  if (recording.rc != ZK_OK) return recording.rc;
  LaneScope lane(ctx, recording.on ? ~0u : relu->serial);  // declared before every temporary

  const uint32_t lens[7] = {k + 5, k + 5, k + 4, k + 4, k, k, k};
  const zk_fr* hs[7] = {u_z_h, v_z_h, u_r_h, v_r_h, u_rec_h, u_hp_h, v_hp_h};
  std::vector<zk_fr> ch;
  uint32_t offs[7], tot = 0;
  for (int i = 0; i < 7; i++) {
    offs[i] = tot;
    for (uint32_t j = 0; j < lens[i]; j++) ch.push_back(hs[i][j]);
    tot += lens[i];
  }
  ch.push_back(zk_fr{});
  DBuf<Fr> chd;
  ZK_TRY(upload_fr(ctx, chd, ch.data(), tot));
  const Fr* c = chd.get();
  if (ctx->relu_packed && ctx->relu_bytes) {
    // byte-histogram form: one pass over the packed words serves the bit sumcheck's first rounds AND the partial_me
    ZK_TRY(binb_sumcheck_pme_dev(ctx, relu->mag_packed, n, 32, c + offs[0], c + offs[1], k + 5, out, c + offs[4], out + 3 * (k + 5) + 1));
    out += 3 * (k + 5) + 1 + 32;
    rec_split(ctx, REC_GROUP_RELU_REM);
    ZK_TRY(binb_sumcheck_pme_dev(ctx, relu->rem_packed, n, 16, c + offs[2], c + offs[3], k + 4, out, c + offs[4], out + 3 * (k + 4) + 1));
    out += 3 * (k + 4) + 1 + 16;
  } else if (ctx->relu_packed) {
    ZK_TRY(binp_sumcheck_dev(ctx, relu->mag_packed, n, 32, c + offs[0], c + offs[1], k + 5, out));
    out += 3 * (k + 5) + 1;
    rec_split(ctx, REC_GROUP_FC_A);  // the five pieces are independent: separate chains, spread over the groups
    ZK_TRY(binp_partial_me_dev(ctx, relu->mag_packed, n, 32, c + offs[4], k, out));
    out += 32;
    rec_split(ctx, REC_GROUP_RELU_REM);
    ZK_TRY(binp_sumcheck_dev(ctx, relu->rem_packed, n, 16, c + offs[2], c + offs[3], k + 4, out));
    out += 3 * (k + 4) + 1;
    rec_split(ctx, REC_GROUP_FC_A);
    ZK_TRY(binp_partial_me_dev(ctx, relu->rem_packed, n, 16, c + offs[4], k, out));
    out += 16;
  } else {
    ZK_TRY(sumcheck_bin_dev(ctx, relu->mag_bin, n * 32, c + offs[0], c + offs[1], k + 5, out));
    out += 3 * (k + 5) + 1;
    ZK_TRY(fr_partial_me_dev(ctx, relu->mag_bin, n * 32, c + offs[4], k, 32, out));
    out += 32;
    ZK_TRY(sumcheck_bin_dev(ctx, relu->rem_bin, n * 16, c + offs[2], c + offs[3], k + 4, out));
    out += 3 * (k + 4) + 1;
    ZK_TRY(fr_partial_me_dev(ctx, relu->rem_bin, n * 16, c + offs[4], k, 16, out));
    out += 16;
  }
  {
    static const int hp_group = [] { const char* e = experiment_env("ZKDL_HP_GROUP"); return e ? atoi(e) : (int)REC_GROUP_FC_OPEN; }();  // A/B: which stream the Hadamard chains share
    rec_split(ctx, (uint32_t)hp_group);
  }
  ZK_TRY(sumcheck_hp_dev(ctx, x, relu->sign, n, c + offs[5], c + offs[6], k, out, ctx->relu_packed ? relu->sign_packed : nullptr));
  return ZK_OK;
}

// LeakyReLU (extension): zkReLU::prove's transcript, then hadamard_product_sumcheck(low, nsign, u_lk, v_lk) -- the negative branch's
// magnitude (low o nsign)~(u_lk), proved the way the reference proves (Z o sign)~(u_hp) (proof.cu:110-150).  ch_h: the reference's
// seven vectors followed by u_lk, v_lk (k each).
int leaky_prove(zk_ctx* ctx, const zk_relu* relu, const Fr* x, const Fr* z, uint32_t n, const zk_fr* const ch_h[9], zk_transcript* tr) {
  if (relu->act_kind != 1 || !relu->low) return ctx->fail(ZK_ERR_ARG, "zk_leaky_prove: the layer's last forward pass was not a LeakyReLU one");
  ProveGuard guard(ctx, tr);
  int rc = relu_prove_body(ctx, relu, x, z, n, ch_h[0], ch_h[1], ch_h[2], ch_h[3], ch_h[4], ch_h[5], ch_h[6], tr);
  if (rc != ZK_OK) return guard.finish(rc);
  const uint32_t k = ceil_log2(n);
  const size_t add_fr = 3 * k + 2;
  rc = tr->ensure(add_fr, 0);
  if (rc != ZK_OK) return guard.finish(rc);
  Fr* out = tr->fr_d + tr->n_fr;
  tr->n_fr += add_fr;
  auto body = [&]() -> int {
    RecScope recording(ctx, &tr->rec, REC_GROUP_FC_A);
    if (recording.rc != ZK_OK) return recording.rc;
    LaneScope lane(ctx, recording.on ? ~0u : relu->serial);
    std::vector<zk_fr> ch(2 * k + 1);
    for (uint32_t j = 0; j < k; j++) {
      ch[j] = ch_h[7][j];
      ch[k + j] = ch_h[8][j];
    }
    DBuf<Fr> chd;
    ZK_TRY(upload_fr(ctx, chd, ch.data(), 2 * k));
    return sumcheck_hp_dev(ctx, relu->low, relu->nsign, n, chd.get(), chd.get() + k, k, out);
  };
  return guard.finish(body());
}

}  // namespace zk
