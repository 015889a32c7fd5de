// zkdl_amd/csrc/commit.hip -- Commitment::commit / me_open / open for gfx950
// (reference: commitment.cu).  Two interchangeable back ends selected by zk_set_msm_mode():
//
//  * ZK_MSM_LITERAL (this file): the reference's op sequence -- one 256-step double-and-add per
//    (generator, scalar) pair, thread-strided row accumulation + 32..1 ladder for commit
//    (commitment.cu:3-41, with the INTENDED one-block-per-row launch), five
//    scalar-muls per thread per round and three reference-shaped sums for me_open
//    (commitment.cu:43-81).  Jacobian limbs equal the oracle's bit for bit; it is the
//    "reference algorithm on MI355X" baseline.
//  * ZK_MSM_PIPPENGER (msm.hip): bucket MSMs over shared bases and the flattened opening; same
//    group elements, compared in affine-canonical form.
#include "layers.hpp"
#include "dev_util.hpp"

namespace zk {

int commit_pippenger_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, uint32_t tn, G1J* out);
int me_open_flat_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, const Fr* u_d, uint32_t k, G1J* proof_d, Fr* ret_d, MsmQueue* queue, const G1J* com_eval_d);
int g1_me_msm_dev(zk_ctx* ctx, zk_gens* pts, const Fr* u_d, uint32_t k, G1J* dst_d, MsmQueue* queue);

int gens_create(zk_ctx* ctx, const G1J* gens_d, uint32_t n, bool copy, zk_gens** out) {
  if (n == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  zk_gens* g = new zk_gens();
  g->ctx = ctx;
  g->n = n;
  if (dev_alloc(ctx, (void**)&g->pts, (size_t)n * sizeof(G1J)) != hipSuccess) {
    delete g;
    return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: generators");
  }
  if (copy) ZK_TRY(dev_copy(ctx, g->pts, gens_d, (size_t)n * sizeof(G1J)));
  *out = g;
  return ZK_OK;
}
void gens_destroy(zk_gens* g) {
  if (!g) return;
  for (auto& kv : g->tables) {
    dev_free(g->ctx, kv.second.pts);
    dev_free(g->ctx, kv.second.inf);
  }
  dev_free(g->ctx, g->full.pts);
  dev_free(g->ctx, g->pts);
  delete g;
}

// one 64-thread block per row: acc = sum_{i = l, l+64, ...} gens[i] * unmont(t[row*n+i]); ladder.
__global__ void __launch_bounds__(64) k_commit_literal(const G1J* __restrict__ gens, uint32_t n, const Fr* __restrict__ t, G1J* __restrict__ out) {
  __shared__ G1J s[64];
  const uint32_t l = threadIdx.x;
  const uint64_t row = blockIdx.x;
  G1J acc = g1_zero();
  if (l < n) {
    acc = g1_mul_naive(gens[l], fr_unmont(t[row * n + l]));
    for (uint32_t i = l + 64; i < n; i += 64) acc = g1_add(acc, g1_mul_naive(gens[i], fr_unmont(t[row * n + i])));
  }
  s[l] = acc;
  __syncthreads();
  for (uint32_t st = 32; st > 0; st >>= 1) {
    if (l < st) s[l] = g1_add(s[l], s[l + st]);
    __syncthreads();
  }
  if (l == 0) out[row] = s[0];
}

int commit_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, uint32_t tn, G1J* out) {
  const uint32_t n = g->n;
  const G1J* gens = g->pts;
  if (n == 0 || tn % n != 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // commitment.cu:31
  uint32_t m = tn / n;
  if (m == 0) return ZK_OK;
  if (ctx->msm_mode == ZK_MSM_PIPPENGER) return commit_pippenger_dev(ctx, g, t, tn, out);
  ZK_LAUNCH(ctx, "commit_literal", 32ull * tn + 144ull * (n + m), k_commit_literal, dim3(m), dim3(64), 0, gens, n, t, out);
  return ZK_OK;
}

// me_open_step, commitment.cu:43-60
__global__ void __launch_bounds__(64) k_me_open_step(const Fr* __restrict__ s, const G1J* __restrict__ g, const Fr* __restrict__ up, Fr* __restrict__ ns,
                                                      G1J* __restrict__ ng, G1J* __restrict__ t, G1J* __restrict__ t0, G1J* __restrict__ t1, uint32_t new_size) {
  uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= new_size) return;
  const Fr u = *up;
  const Fr uu = fr_unmont(u);
  const uint32_t i0 = 2 * gid, i1 = 2 * gid + 1;
  const Fr s0 = s[i0], s1 = s[i1];
  const G1J g0 = g[i0], g1 = g[i1];
  ns[gid] = fr_add(s0, fr_mul(u, fr_sub(s1, s0)));
  ng[gid] = g1_add(g1, g1_mul_naive(g1_add(g0, g1_neg(g1)), uu));
  t[gid] = g1_add(g1_mul_naive(g0, s0), g1_mul_naive(g1, s1));
  t0[gid] = g1_mul_naive(g1, s0);
  t1[gid] = g1_mul_naive(g0, s1);
}

static int me_open_literal_dev(zk_ctx* ctx, const Fr* t, const G1J* gens, uint32_t n, const Fr* u_d, uint32_t k, G1J* proof_d, Fr* ret_d) {
  uint32_t half = n / 2 + 1;
  ZK_ALLOC(ctx, sbuf, Fr, 2 * (size_t)half);
  ZK_ALLOC(ctx, gbuf, G1J, 2 * (size_t)half);
  ZK_ALLOC(ctx, tmp, G1J, 3 * (size_t)half);
  const Fr* s = t;
  const G1J* g = gens;
  uint32_t sz = n;
  for (uint32_t r = 0; r < k; r++) {
    uint32_t ns = sz / 2;
    if (ns == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
    Fr* so = sbuf.get() + (r & 1) * half;
    G1J* go = gbuf.get() + (r & 1) * half;
    G1J *T = tmp.get(), *T0 = tmp.get() + half, *T1 = tmp.get() + 2 * half;
    ZK_LAUNCH(ctx, "me_open_step", 352ull * sz, k_me_open_step, dim3((ns + 63) / 64), dim3(64), 0, s, g, u_d + r, so, go, T, T0, T1, ns);
    ZK_TRY(g1_sum_dev(ctx, T, ns, proof_d + 3 * r + 0));
    ZK_TRY(g1_sum_dev(ctx, T0, ns, proof_d + 3 * r + 1));
    ZK_TRY(g1_sum_dev(ctx, T1, ns, proof_d + 3 * r + 2));
    s = so;
    g = go;
    sz = ns;
  }
  ZK_TRY(dev_copy(ctx, proof_d + 3 * k, g, sizeof(G1J)));  // generators(0)
  ZK_TRY(dev_copy(ctx, ret_d, s, sizeof(Fr)));            // t(0)
  return ZK_OK;
}

// Commitment::me_open, commitment.cu:62-81
int me_open_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, const Fr* u_d, uint32_t k, G1J* proof_d, Fr* ret_d, MsmQueue* queue, const G1J* com_eval_d) {
  const uint32_t n = g->n;
  if (n == 0 || k > 31) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if ((n >> k) == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // a round would see an empty tensor
  // the flattened form needs the full binary tree the prover always has (n == 2^k, commitment.cu:89);
  // ragged sizes, which only the bare me_open API can produce, keep the reference's op sequence
  if (ctx->msm_mode == ZK_MSM_PIPPENGER && k <= 19 && n == (1u << k)) return me_open_flat_dev(ctx, g, t, u_d, k, proof_d, ret_d, queue, com_eval_d);
  return me_open_literal_dev(ctx, t, g->pts, n, u_d, k, proof_d, ret_d);
}

// Commitment::open, commitment.cu:83-92.  u = u_lo || u_hi with |u_hi| = ceilLog2(m).
int open_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, uint32_t tn, zk_gens* com, const Fr* u_d, uint32_t k, G1J* com_eval_d, G1J* proof_d, Fr* ret_d, MsmQueue* queue,
             const int16_t* tq) {
  const uint32_t n = g->n, m = com->n;
  uint32_t k_hi = ceil_log2(m);
  if (k_hi > k) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  uint32_t k_lo = k - k_hi;
  const Fr* u_lo = u_d;
  const Fr* u_hi = u_d + k_lo;
  // com(u_hi): computed by the reference and dropped (commitment.cu:88); kept as timed work and
  // recorded in the transcript.  With digit tables of the GENERATORS at hand (zk_gens_prepare_full) it is computed as the commitment of the
  // row-folded tensor instead of as a multilinear evaluation over the commitment points: com_row = sum_j g_j unmont(t[row][j]) (commit_dev),
  // so sum_row eq(u_hi, row) com_row = sum_j g_j unmont(folded_j) -- the same group element, one more MSM over the generators' tables, and no
  // bucket-method job (over the commitment points' window tables) left in a proof.
  const bool com_by_fold = ctx->msm_mode == ZK_MSM_PIPPENGER && queue && ctx->full_tables && g->full.pts && !ctx->msm_window;
  if (com_by_fold) {
    if (m != (1u << k_hi)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  } else if (ctx->msm_mode == ZK_MSM_PIPPENGER) {
    ZK_TRY(g1_me_msm_dev(ctx, com, u_hi, k_hi, com_eval_d, queue));
  } else {
    ZK_TRY(g1_me_dev(ctx, com->pts, m, u_hi, k_hi, com_eval_d));
  }
  if (k_lo > 31 || n != (1u << k_lo)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // commitment.cu:89
  uint32_t fsz = fr_partial_me_size(tn, k_hi, n);
  if (fsz != n) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  ZK_ALLOC(ctx, folded, Fr, n);
  if (tq && k_hi > 0) ZK_TRY(fr_partial_me_q16_dev(ctx, tq, tn, u_hi, k_hi, n, folded.get()));
  else ZK_TRY(fr_partial_me_dev(ctx, t, tn, u_hi, k_hi, n, folded.get()));
  if (com_by_fold) ZK_TRY(msm_enqueue(ctx, g, folded.get(), 1, /*do_unmont=*/true, /*probe_bits=*/false, com_eval_d, queue));
  // com_eval_d (this call's com(u_hi), a slot of the same deferred batch) lets the derived-rounds form obtain T_0 = [R] com(u_hi) as well
  return me_open_dev(ctx, g, folded.get(), u_lo, k_lo, proof_d, ret_d, queue, ctx->msm_mode == ZK_MSM_PIPPENGER ? com_eval_d : nullptr);
}

}  // namespace zk
