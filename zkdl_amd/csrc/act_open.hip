// zkdl_amd/csrc/act_open.hip -- activation commitments inside the proof: ONE opening per hidden layer ("commitment of
// activations / proof compression across layers", the reference's stated future work, README.md:131-135).
//
// The reference's layer proofs leave the activation-side evaluations as open claims: zkFC_i ends with a claim on Z_i at
// p1 = u_out || u_bs, zkReLU_i with a claim on Z_i at p2 = v_hp, zkFC_{i+1} with a claim on A_i at p3 = u_in || u_bs (verify_mlp lists
// them).  With Z_i and A_i committed row-wise over layer i's generators (zk_commit), round 2 closed each claim with its own opening:
// three openings per hidden layer, run one at a time through the synchronous API (210 ms for 21 openings at full size).  Here the
// three claims of a layer are first REDUCED to one, then opened once, and everything is recorded into the proof's own batch:
//
//   * the three claims are one inner product over the concatenated tensor T = [Z_i ; A_i] (2N entries, N = B_p * out_p):
//         y1 + alpha y2 + alpha^2 y3  =  sum_x T[x] E[x],    E = [ eq(p1, .) + alpha eq(p2, .) ; alpha^2 eq(p3, .) ],
//     a statement the ordinary inner-product sumcheck proves (proof.cu:55-108: k + 1 rounds), ending in T~(r') and E~(r') at the
//     round challenges r' = (r, r_top).  E~(r') is public arithmetic; T~(r') = (1 - r_top) Z~(r) + r_top A~(r) is a random linear
//     combination of the two tensors at ONE point;
//   * the commitment of T is the concatenation [com(Z_i) ; com(A_i)] of the two row-wise commitments (same generators), so T~(r')
//     is opened by ONE Commitment::open (commitment.cu:83-92) at r' -- the row challenges simply include r_top;
//   * the prover knows the tensor, so com(u_hi) is computed as the commitment of the row-folded tensor (one MSM over the layer's
//     FIXED generators, whose window tables exist) instead of a multilinear evaluation over the per-proof commitment points: the same
//     group element by linearity, and the whole opening then consists of MSMs over the generators -- jobs of the transcript's deferred
//     batch, accumulated in the same launches as the weight openings.
//
// Per hidden layer: one (k + 1)-round sumcheck over 2^(k+1) entries and one opening instead of three openings; the verifier checks the
// sumcheck, evaluates E~(r') itself, and checks the opening against [com(Z_i) ; com(A_i)] with the checks of zk_verify_open.
#include <string.h>

#include <vector>

#include "layers.hpp"

#include "dev_util.hpp"

namespace zk {

// factor tables of the three points: lo[q][j] = eq(p_q[0..c), j), hi[q][j] = eq(p_q[c..k), j)
__device__ __forceinline__ void k_act_eq_tables(uint32_t zk_nblk, const Fr* __restrict__ p, uint32_t k, uint32_t c, Fr* __restrict__ lo, Fr* __restrict__ hi) {
  const uint32_t nlo = 1u << c, nhi = 1u << (k - c);
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 3 * (nlo + nhi); idx += zk_nblk * blockDim.x) {
    const uint32_t q = idx / (nlo + nhi), j = idx % (nlo + nhi);
    const Fr* pq = p + (size_t)q * k;
    const bool low = j < nlo;
    const uint32_t jj = low ? j : j - nlo, nv = low ? c : k - c;
    const Fr* u = low ? pq : pq + c;
    Fr one = fr_one(), acc = one;
    for (uint32_t b = 0; b < nv; b++) {
      const Fr ub = u[b];
      acc = fr_mul(acc, ((jj >> b) & 1u) ? ub : fr_sub(one, ub));
    }
    if (low) lo[(size_t)q * nlo + jj] = acc;
    else hi[(size_t)q * nhi + jj] = acc;
  }
}
// T = [Z ; A], E = [eq(p1) + alpha eq(p2) ; alpha^2 eq(p3)]
__device__ __forceinline__ void k_act_concat(uint32_t zk_nblk, const Fr* __restrict__ Z, const Fr* __restrict__ A, uint32_t n, const Fr* __restrict__ lo, const Fr* __restrict__ hi,
                                              uint32_t c, uint32_t k, Fr alpha, Fr alpha2, Fr* __restrict__ T, Fr* __restrict__ E) {
  const uint32_t nlo = 1u << c, nhi = 1u << (k - c), mask = nlo - 1u;
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < 2 * n; x += zk_nblk * blockDim.x) {
    if (x < n) {
      T[x] = Z[x];
      const Fr e1 = fr_mul(lo[x & mask], hi[x >> c]);
      const Fr e2 = fr_mul(lo[nlo + (x & mask)], hi[nhi + (x >> c)]);
      E[x] = fr_add(e1, fr_mul(alpha, e2));
    } else {
      const uint32_t y = x - n;
      T[x] = A[y];
      E[x] = fr_mul(alpha2, fr_mul(lo[2 * nlo + (y & mask)], hi[2 * nhi + (y >> c)]));
    }
  }
}

// Commitment::open for a tensor the prover holds together with its row-wise commitment: com(u_hi) as the commitment of the row-folded
// tensor (an MSM over the generators), then me_open (commitment.cu:43-81); every MSM goes to `queue`
static int open_known_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, uint32_t tn, const Fr* u_d, uint32_t k, G1J* com_eval_d, G1J* proof_d, Fr* ret_d, MsmQueue* queue) {
  const uint32_t n = g->n;
  if (n == 0 || tn % n) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t m = tn / n, k_hi = ceil_log2(m);
  if (k_hi > k || m != (1u << k_hi)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t k_lo = k - k_hi;
  if (k_lo > 31 || n != (1u << k_lo)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // commitment.cu:89
  ZK_ALLOC(ctx, folded, Fr, n);
  ZK_TRY(fr_partial_me_dev(ctx, t, tn, u_d + k_lo, k_hi, n, folded.get()));
  ZK_TRY(msm_enqueue(ctx, g, folded.get(), 1, /*do_unmont=*/true, /*probe_bits=*/false, com_eval_d, queue));
  return me_open_dev(ctx, g, folded.get(), u_d, k_lo, proof_d, ret_d, queue, com_eval_d);
}

// pts_h = p1 | p2 | p3 (k challenges each), fold_h: the k + 1 round challenges.  Appends Fr: [sumcheck 3 (k + 1) + 2][t(0)], G1: [com(u_hi)][3 k_lo + 1].
// fs_seed32 != nullptr (zk_act_open_fs): alpha = zk_fs_derive(fs_seed32, layer, 2, 0) and every ROUND challenge of the reduction -- hence the opening
// point r' -- is a hash of the rounds before it, computed on the device (fs_rounds.hip); alpha_h / fold_h are then ignored.
static int act_open_body(zk_ctx* ctx, zk_gens* gens, const Fr* z, const Fr* a, uint32_t n, const zk_fr* pts_h, const zk_fr* alpha_h, const zk_fr* fold_h, zk_transcript* tr,
                         const uint8_t* fs_seed32 = nullptr, uint32_t layer = 0) {
  const uint32_t n_gens = gens->n;
  const uint32_t k = ceil_log2(n);
  if (n == 0 || n != (1u << k) || k + 1 > 31 || (2ull * n) % n_gens) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t k_lo = ceil_log2(n_gens);
  if (n_gens != (1u << k_lo) || k_lo > k) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const size_t add_fr = 3 * (size_t)(k + 1) + 2 + 1, add_g1 = 1 + 3 * (size_t)k_lo + 1;
  ZK_TRY(tr->ensure(add_fr, add_g1));
  Fr* const fr_slot = tr->fr_d + tr->n_fr;
  G1J* const g1_slot = tr->g1_d + tr->n_g1;
  tr->n_fr += add_fr;
  tr->n_g1 += add_g1;
  RecScope recording(ctx, &tr->rec, REC_GROUP_FC_OPEN);
  if (recording.rc != ZK_OK) return recording.rc;
  // ONE chain (the sumcheck and the opening both read T), marked early: it feeds the proof's MSM batch, and a sealed proof's G1 phase
  // starts behind the early chains only (layers.hip: g1_phase_run)
  rec_split(ctx, REC_GROUP_FC_OPEN, /*early=*/true);
  LaneScope lane(ctx, recording.on ? ~0u : 0u);
  // [p1 | p2 | p3 | fold | hash state (sound mode)]
  std::vector<zk_fr> ch(3 * (size_t)k + k + 1 + (fs_seed32 ? 1 : 0), zk_fr{});
  for (size_t i = 0; i < 3 * (size_t)k; i++) ch[i] = pts_h[i];
  zk_fr alpha_fs;
  if (fs_seed32) {
    ZK_TRY(zk_fs_derive(fs_seed32, layer, 2, 0, 1, &alpha_fs));
    alpha_h = &alpha_fs;
    fs_state0_slot(fs_seed32, layer, 2, 0, &ch[3 * (size_t)k + k + 1]);
  } else {
    for (uint32_t i = 0; i <= k; i++) ch[3 * (size_t)k + i] = fold_h[i];
  }
  DBuf<Fr> chd;
  ZK_TRY(upload_fr(ctx, chd, ch.data(), (uint32_t)ch.size()));
  Fr* fold_d = chd.get() + 3 * (size_t)k;
  const uint32_t c = k < 10 ? k : 10;
  ZK_ALLOC(ctx, lo, Fr, (size_t)3 << c);
  ZK_ALLOC(ctx, hi, Fr, (size_t)3 << (k - c));
  ZK_ALLOC(ctx, T, Fr, 2ull * n);
  ZK_ALLOC(ctx, E, Fr, 2ull * n);
  const uint32_t ntab = 3u * ((1u << c) + (1u << (k - c)));
  ZK_LAUNCHB(ctx, "eq_table", 32ull * ntab, 256, k_act_eq_tables, dim3((ntab + 255) / 256), 0, (const Fr*)chd.get(), k, c, lo.get(), hi.get());
  Fr alpha, alpha2;
  memcpy(&alpha, alpha_h, sizeof alpha);
  alpha2 = fr_mul(alpha, alpha);
  ZK_LAUNCHB(ctx, "act_concat", 32ull * 6 * n, 256, k_act_concat, dim3(stream_grid(2ull * n)), 0, z, a, n, (const Fr*)lo.get(), (const Fr*)hi.get(), c, k, alpha, alpha2, T.get(),
             E.get());
  if (fs_seed32) ZK_TRY(sumcheck_ip_fs_dev(ctx, T.get(), E.get(), 2 * n, fold_d, k + 1, fr_slot));
  else ZK_TRY(sumcheck_ip_dev(ctx, T.get(), E.get(), 2 * n, fold_d, k + 1, fr_slot));
  // the opening of T at r' = the round challenges
  return open_known_dev(ctx, gens, T.get(), 2 * n, fold_d, k + 1, g1_slot, g1_slot + 1, fr_slot + 3 * (size_t)(k + 1) + 2, &tr->queue);
}

}  // namespace zk

using namespace zk;

namespace {
struct ActGuard {
  zk_ctx* ctx;
  zk_transcript* tr;
  size_t n_fr, n_g1, chains, jobs, keep, derives;
  ActGuard(zk_ctx* c, zk_transcript* t)
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
inline Fr ldf(const zk_fr* p) {
  Fr r;
  memcpy(&r, p, sizeof r);
  return r;
}
// eq(p, r) = prod_j (p_j r_j + (1 - p_j)(1 - r_j))
Fr eq_eval(const zk_fr* p, const zk_fr* r, uint32_t k) {
  const Fr one = fr_one();
  Fr acc = one;
  for (uint32_t j = 0; j < k; j++) {
    const Fr pj = ldf(p + j), rj = ldf(r + j);
    acc = fr_mul(acc, fr_add(fr_mul(pj, rj), fr_mul(fr_sub(one, pj), fr_sub(one, rj))));
  }
  return acc;
}
}  // namespace

extern "C" {

int zk_act_open(zk_ctx* ctx, zk_gens* gens, const zk_fr* z_d, const zk_fr* a_d, uint32_t n, const zk_fr* pts_h, const zk_fr* alpha_h, const zk_fr* fold_h, zk_transcript* tr) {
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (!gens || !z_d || !a_d || !pts_h || !alpha_h || !fold_h || !tr) return ctx->fail(ZK_ERR_ARG, "zk_act_open: null argument");
  if ((((uintptr_t)z_d) | ((uintptr_t)a_d)) & 15u) return ctx->fail(ZK_ERR_ARG, "misaligned device pointer");
  ActGuard guard(ctx, tr);
  return guard.finish(act_open_body(ctx, gens, reinterpret_cast<const Fr*>(z_d), reinterpret_cast<const Fr*>(a_d), n, pts_h, alpha_h, fold_h, tr));
}

// Row-wise commitment of a tensor (Commitment::commit's intended row MSM, commitment.cu:29-41, as zk_commit computes it) DEFERRED into a transcript: the
// tn / n_gens commitment points are appended to tr's G1 list and their MSMs join the transcript's batch -- on the digit tables when the generator set
// has them (a zero digit costs nothing there, so short scalars such as activations pay for their non-zero windows only), and a sealed transcript runs them
// under other proofs' work.  This is what makes the activation commitments part of the proving pipeline instead of a synchronous setup call per tensor.
// max_bits: 0, or the caller's bound on the bit length of the CENTRED values (min(v, r - v)): only the windows such scalars can touch are walked -- a 48-bit
// activation is 5 of 22 twelve-bit windows; a longer scalar raises the context's range flag (zk_ctx_sync returns ZK_ERR_RANGE).
int zk_act_commit(zk_ctx* ctx, zk_gens* gens, const zk_fr* t_d, uint32_t tn, uint32_t max_bits, zk_transcript* tr) {
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (!gens || !t_d || !tr) return ctx->fail(ZK_ERR_ARG, "zk_act_commit: null argument");
  if (((uintptr_t)t_d) & 15u) return ctx->fail(ZK_ERR_ARG, "misaligned device pointer");
  const uint32_t n = gens->n;
  if (n == 0 || tn == 0 || tn % n) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");   // commitment.cu:31
  if (max_bits > 254) return ctx->fail(ZK_ERR_ARG, "zk_act_commit: max_bits 0 .. 254");
  const uint32_t m = tn / n;
  ActGuard guard(ctx, tr);
  auto body = [&]() -> int {
    ZK_TRY(tr->ensure(0, m));
    G1J* const slot = tr->g1_d + tr->n_g1;
    tr->n_g1 += m;
    RecScope recording(ctx, &tr->rec, REC_GROUP_FC_OPEN);
    if (recording.rc != ZK_OK) return recording.rc;
    rec_split(ctx, REC_GROUP_FC_OPEN, /*early=*/true);   // scalar preparation only: the G1 phase may start behind it
    LaneScope lane(ctx, recording.on ? ~0u : 0u);
    return msm_enqueue(ctx, gens, reinterpret_cast<const Fr*>(t_d), m, /*do_unmont=*/true, /*probe_bits=*/false, slot, &tr->queue, nullptr, nullptr, max_bits);
  };
  return guard.finish(body());
}

int zk_act_open_fs(zk_ctx* ctx, zk_gens* gens, const zk_fr* z_d, const zk_fr* a_d, uint32_t n, const zk_fr* pts_h, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (!gens || !z_d || !a_d || !pts_h || !seed32 || !tr) return ctx->fail(ZK_ERR_ARG, "zk_act_open_fs: null argument");
  if ((((uintptr_t)z_d) | ((uintptr_t)a_d)) & 15u) return ctx->fail(ZK_ERR_ARG, "misaligned device pointer");
  ActGuard guard(ctx, tr);
  return guard.finish(act_open_body(ctx, gens, reinterpret_cast<const Fr*>(z_d), reinterpret_cast<const Fr*>(a_d), n, pts_h, nullptr, nullptr, tr, seed32, layer));
}

// The verifier's side of zk_act_open.  com_d: the 2m points [com(Z) ; com(A)] (device); y_h: the three claimed evaluations Z~(p1), Z~(p2),
// A~(p3) as the layer proofs state them; fr_h: [sumcheck 3 (k + 1) + 2][t(0)]; g1_h / inf_h: [com(u_hi)][3 k_lo + 1] affine-canonical.
int zk_verify_act_open(zk_ctx* ctx, zk_gens* gens, const zk_g1j* com_d, uint32_t m2, uint32_t k, const zk_fr* pts_h, const zk_fr* alpha_h, const zk_fr* fold_h,
                       const zk_fr* y_h, const zk_fr* fr_h, size_t n_fr, const zk_g1a* g1_h, const uint8_t* inf_h, size_t n_g1, int* ok) {
  if (!ctx) return ZK_ERR_ARG;
  if (!gens || !com_d || !pts_h || !alpha_h || !fold_h || !y_h || !fr_h || !g1_h || !inf_h || !ok) return ctx->fail(ZK_ERR_ARG, "zk_verify_act_open: null argument");
  *ok = 0;
  if (n_fr != 3 * (size_t)(k + 1) + 3) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  auto fail = [&](const std::string& what) {
    ctx->last_error = "verification failed: " + what;
    return ZK_OK;
  };
  {  // canonical encodings only (see verify.hip: first_unreduced)
    Fr rm;
    for (int i = 0; i < 8; i++) rm.v[i] = FrP::P(i);
    for (size_t i = 0; i < n_fr; i++)
      if (l_gte(ldf(fr_h + i), rm)) return fail("transcript value " + std::to_string(i) + " is not a canonical field element (>= r)");
    for (size_t i = 0; i < 3; i++)
      if (l_gte(ldf(y_h + i), rm)) return fail("claimed evaluation " + std::to_string(i) + " is not a canonical field element (>= r)");
  }
  const Fr alpha = ldf(alpha_h), alpha2 = fr_mul(alpha, alpha);
  // sum_x T[x] E[x] = y1 + alpha y2 + alpha^2 y3   (proof.cu:55-96)
  Fr claim = fr_add(ldf(y_h), fr_add(fr_mul(alpha, ldf(y_h + 1)), fr_mul(alpha2, ldf(y_h + 2))));
  for (uint32_t r = 0; r <= k; r++) {
    const Fr c0 = ldf(fr_h + 3 * r), c1 = ldf(fr_h + 3 * r + 1), c2 = ldf(fr_h + 3 * r + 2);
    if (!l_eq(fr_add(c0, fr_add(c0, fr_add(c1, c2))), claim)) return fail("claim-reduction sumcheck, round " + std::to_string(r) + ": p(0) + p(1) != claim");
    const Fr x = ldf(fold_h + r);
    claim = fr_add(c0, fr_mul(x, fr_add(c1, fr_mul(x, c2))));
  }
  const Fr fa = ldf(fr_h + 3 * (size_t)(k + 1)), fb = ldf(fr_h + 3 * (size_t)(k + 1) + 1), t0 = ldf(fr_h + 3 * (size_t)(k + 1) + 2);
  if (!l_eq(fr_mul(fa, fb), claim)) return fail("claim-reduction sumcheck: T~ * E~ != final claim");
  // E~(r') is public: (1 - r_top) (eq(p1, r) + alpha eq(p2, r)) + r_top alpha^2 eq(p3, r)
  const Fr rtop = ldf(fold_h + k);
  const Fr e = fr_add(fr_mul(fr_sub(fr_one(), rtop), fr_add(eq_eval(pts_h, fold_h, k), fr_mul(alpha, eq_eval(pts_h + k, fold_h, k)))),
                      fr_mul(rtop, fr_mul(alpha2, eq_eval(pts_h + 2 * (size_t)k, fold_h, k))));
  if (!l_eq(fb, e)) return fail("claim-reduction sumcheck: E~(r') is not the evaluation of the public weight table");
  if (!l_eq(fa, t0)) return fail("the opened value differs from the sumcheck's T~(r')");
  return zk_verify_open(ctx, gens, com_d, m2, fold_h, k + 1, reinterpret_cast<const zk_fr*>(&t0), g1_h, inf_h, n_g1, ok);
}

}  // extern "C"
