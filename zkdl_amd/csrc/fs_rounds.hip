// zkdl_amd/csrc/fs_rounds.hip -- ROUND-BY-ROUND Fiat-Shamir provers ("sound mode").
//
// The reference draws every challenge from std::random_device before any round runs (proof.cu:3-11, zkfc.cu:135-137,
// zkrelu.cu:85-98): its sumchecks are provers for an INTERACTIVE verifier whose coins happen to be pre-drawn.  The throughput path
// (layers.hip) keeps exactly that interface -- challenge vectors are arguments -- which is what bit-parity with the reference is
// defined on, and what lets a proof's 15 calls and the rounds inside them be fused and zipped.  It is not a non-interactive
// argument: a prover who knows x_r before sending round r's message is not bound by it.
//
// Here round r's fold challenge is  x_r = H(state_r),  state_{r+1} = SHA-256(state_r | round r's three coefficients)  with
// state_0 = SHA-256("zkdl-amd/fs-rounds/v1" | statement digest | layer | kind | which sumcheck): every fold challenge depends on
// every message the prover sent before it (sha256.hpp; the verifier re-derives them from the serialised proof with the same
// function, zk_fs_round_challenges).  Consequences for the kernels:
//   * the fused round of sumcheck.hip (coefficients of round r AND the fold by x_r in one pass) cannot be used: x_r does not exist
//     until the coefficients are final.  k_fs_round fuses the other way round -- the fold by x_{r-1} and the coefficients of
//     round r -- and a one-workgroup kernel (k_fs_hash) reduces the workgroups' partials, writes the round's message, hashes and
//     leaves x_r in device memory for the next launch.  No host round trip: a sumcheck is still one asynchronous chain, and the
//     chains of different layers are still zipped by the recorder.
//   * the last <= 10 rounds run in one workgroup with the table in LDS (k_fs_tail), one lane hashing between rounds.
//   * the bit sumchecks of zkReLU take ALL their in-word rounds (5 / 4) from the packed words, staged by what each message needs
//     (relu_packed.hip: fsb_sumcheck_head_dev): one weighted byte histogram that depends on the weight point only -- statement-derived
//     -- serves rounds 0-2 (one workgroup, hashing between them); a cross-term histogram behind x_2 serves round 3; the half-word
//     products behind x_3 serve round 4; the generic rounds here start from one entry per word.  (The throughput mode's single pass
//     presumes x_3 and x_4 before their rounds.)
//   * the opening's point is u_out || u_in and u_in are the inner-product sumcheck's round challenges, so the opening is recorded
//     BEHIND the sumcheck in the same chain; its own folding uses the point's coordinates (commitment.cu:43-81), no new coins.
// Challenges that are evaluation points fixed before a sub-protocol starts (u_bs, u_out, u_z, u_r, u_rec, u_hp) stay
// statement-derived (zk_fs_derive).  The transcript has the same layout as the reference-parity mode's, and -- fed the challenges
// this mode derives -- the CPU oracle reproduces it bit for bit (tests/test_gpu_fs_rounds.py).
#include <string.h>

#include <vector>

#include "layers.hpp"

#include "dev_util.hpp"
#include "sha256.hpp"

namespace zk {

enum { FSC_IP = 0, FSC_HP = 1, FSC_BIN = 2 };
static constexpr uint32_t FS_TAIL_N = 1024;
static constexpr uint32_t FS_MAX_BLOCKS = 1024;

__device__ __forceinline__ Fr fs_eq_product(const Fr* __restrict__ u, uint32_t nvars, uint32_t j) {
  Fr one = fr_one(), acc = one;
  for (uint32_t b = 0; b < nvars; b++) {
    Fr ub = u[b];
    acc = fr_mul(acc, ((j >> b) & 1u) ? ub : fr_sub(one, ub));
  }
  return acc;
}
// weight tables of all big rounds (same layout as sumcheck.hip's k_sc_prep): B[j] = eq(w[s..k), j), A[aoff(r) + j] = eq(w[r+1..s), j),
// aoff(r) = 2^s - 2^(s-r); the weight of pair h in round r is A_r[h mod 2^(s-1-r)] * B[h >> (s-1-r)]
__device__ __forceinline__ void k_fs_prep(uint32_t zk_nblk, const Fr* __restrict__ w, uint32_t k, uint32_t s, Fr* __restrict__ A, Fr* __restrict__ B) {
  const uint32_t nA = (1u << s) - 1u, nB = 1u << (k - s);
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nA + nB; idx += zk_nblk * blockDim.x) {
    if (idx < nA) {
      uint32_t rem = (1u << s) - idx;
      uint32_t r = s - (32u - __clz(rem - 1u));
      uint32_t j = idx - ((1u << s) - (1u << (s - r)));
      A[idx] = fs_eq_product(w + r + 1, s - 1 - r, j);
    } else {
      uint32_t j = idx - nA;
      B[j] = fs_eq_product(w + s, k - s, j);
    }
  }
}

// element e of the CURRENT table: the previous table folded by x (Fr_me_step's rules, fr-tensor.cu:399-409: a missing partner
// reads as zero), or the previous table itself in the first round
__device__ __forceinline__ Fr fs_cur_elem(const Fr* __restrict__ prev, uint32_t prev_size, bool first, const Fr& x, uint64_t e) {
  if (first) return e < prev_size ? prev[e] : fr_zero();
  const uint64_t e0 = 2 * e, e1 = e0 + 1;
  if (e0 >= prev_size) return fr_zero();
  const Fr p0 = prev[e0];
  const Fr p1 = e1 < prev_size ? prev[e1] : fr_zero();
  return fr_add(p0, fr_mul(x, fr_sub(p1, p0)));
}

// One big round: fold the previous table(s) by x_{r-1} (not in round 0), write the current table(s), accumulate round r's three
// coefficients into per-workgroup partials.  npairs = ceil(cur_size / 2).
template <int MODE>
__device__ __forceinline__ void k_fs_round(uint32_t zk_nblk, const Fr* __restrict__ a_prev, const Fr* __restrict__ b_prev, Fr* __restrict__ a_cur, Fr* __restrict__ b_cur,
                                            const Fr* __restrict__ fold_prev, const Fr* __restrict__ A, uint32_t c, const Fr* __restrict__ B, uint32_t prev_size,
                                            uint32_t cur_size, uint32_t npairs, Fr* __restrict__ partial, uint32_t b_is_bits) {
  __shared__ Fr lds[12];
  const bool first = fold_prev == nullptr;
  const Fr x = first ? fr_zero() : *fold_prev;
  const uint32_t mask = (1u << c) - 1u;
  Fr s0 = fr_zero(), s1 = fr_zero(), s2 = fr_zero();
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += zk_nblk * blockDim.x) {
    const uint64_t e0 = 2ull * g, e1 = e0 + 1;
    const Fr a0 = fs_cur_elem(a_prev, prev_size, first, x, e0), a1 = e1 < cur_size ? fs_cur_elem(a_prev, prev_size, first, x, e1) : fr_zero();
    if (!first) {
      a_cur[e0] = a0;
      if (e1 < cur_size) a_cur[e1] = a1;
    }
    if (MODE == FSC_BIN && l_is_zero(a0) && l_is_zero(a1)) continue;  // nothing to add (bit tables are mostly zero in their high positions)
    const Fr da = fr_sub(a1, a0);
    Fr c0, c1, c2;
    if (MODE == FSC_BIN) {
      const Fr w = fr_mul(A[g & mask], B[g >> c]);
      const Fr a0w = fr_mul(a0, w), daw = fr_mul(da, w);
      c0 = fr_sub(fr_mul(a0w, a0), a0w);          // w (a0^2 - a0)          proof.cu:152-163
      c1 = fr_sub(fr_dbl(fr_mul(a0w, da)), daw);  // w (2 a0 da - da)
      c2 = fr_mul(daw, da);                       // w da^2
    } else {
      const Fr b0 = fs_cur_elem(b_prev, prev_size, first, x, e0), b1 = e1 < cur_size ? fs_cur_elem(b_prev, prev_size, first, x, e1) : fr_zero();
      if (!first) {
        b_cur[e0] = b0;
        if (e1 < cur_size) b_cur[e1] = b1;
      }
      const Fr db = fr_sub(b1, b0);
      if (MODE == FSC_HP) {
        const Fr w = fr_mul(A[g & mask], B[g >> c]);
        const Fr a0w = fr_mul(a0, w), daw = fr_mul(da, w);
        if (first && b_is_bits) {
          // round 0 of zkReLU's Hadamard sumchecks: the second operand is a 0 / 1 tensor (sign, 1 - sign), so its three products are selections
          const bool z0 = l_is_zero(b0), z1 = l_is_zero(b1);
          c0 = z0 ? fr_zero() : a0w;
          c2 = z0 == z1 ? fr_zero() : (z1 ? fr_neg(daw) : daw);
          c1 = fr_sub(fr_sub(z1 ? fr_zero() : fr_add(a0w, daw), c0), c2);
        } else {
          c0 = fr_mul(a0w, b0);                     // proof.cu:110-137
          c2 = fr_mul(daw, db);
          c1 = fr_sub(fr_sub(fr_mul(fr_add(a0w, daw), b1), c0), c2);
        }
      } else {
        c0 = fr_mul(a0, b0);                      // proof.cu:55-70
        c2 = fr_mul(da, db);
        c1 = fr_sub(fr_sub(fr_mul(a1, b1), c0), c2);
      }
    }
    s0 = fr_add(s0, c0);
    s1 = fr_add(s1, c1);
    s2 = fr_add(s2, c2);
  }
  block_sum3_fr(s0, s1, s2, lds);
  if (threadIdx.x == 0) {
    partial[(uint64_t)blockIdx.x * 3 + 0] = s0;
    partial[(uint64_t)blockIdx.x * 3 + 1] = s1;
    partial[(uint64_t)blockIdx.x * 3 + 2] = s2;
  }
}

// one workgroup: the round's message = the sum of the workgroups' partials; then the hash step leaves x_r in *fold_dst
__device__ __forceinline__ void k_fs_hash(uint32_t zk_nblk, const Fr* __restrict__ partial, uint32_t nblocks, Fr* __restrict__ msg, FsState* __restrict__ state,
                                           Fr* __restrict__ fold_dst) {
  __shared__ Fr lds[12];
  Fr s0 = fr_zero(), s1 = fr_zero(), s2 = fr_zero();
  for (uint32_t i = threadIdx.x; i < nblocks; i += blockDim.x) {
    s0 = fr_add(s0, partial[(uint64_t)i * 3 + 0]);
    s1 = fr_add(s1, partial[(uint64_t)i * 3 + 1]);
    s2 = fr_add(s2, partial[(uint64_t)i * 3 + 2]);
  }
  block_sum3_fr(s0, s1, s2, lds);
  if (threadIdx.x == 0) {
    msg[0] = s0;
    msg[1] = s1;
    msg[2] = s2;
    FsState st = *state;
    *fold_dst = fs_round_step(st, s0, s1, s2);
    *state = st;
  }
}

// the remaining rounds r0 .. k-1 in one workgroup, table(s) in LDS; between a round's message and its fold one lane hashes.
// prev: the table before the pending fold by fold[r0 - 1] (r0 > 0), or the input table (r0 == 0); the folded table has <= FS_TAIL_N entries.
template <int MODE>
__device__ __forceinline__ void k_fs_tail(uint32_t zk_nblk, const Fr* __restrict__ a_prev, const Fr* __restrict__ b_prev, uint32_t prev_size, Fr* __restrict__ fold,
                                           const Fr* __restrict__ w, uint32_t k, uint32_t r0, Fr* __restrict__ proof, FsState* __restrict__ state) {
  extern __shared__ Fr sm[];
  Fr* ta = sm;
  Fr* tb = sm + FS_TAIL_N;
  Fr* E = sm + (MODE == FSC_BIN ? FS_TAIL_N : 2 * FS_TAIL_N);
  Fr* red = E + FS_TAIL_N / 2;  // 24 reduction slots
  Fr* xs = red + 24;            // the round challenge, broadcast
  const uint32_t tid = threadIdx.x;
  const bool first = r0 == 0;
  const Fr xp = first ? fr_zero() : fold[r0 - 1];
  uint32_t sz = first ? prev_size : (prev_size + 1) / 2;
  for (uint32_t i = tid; i < sz; i += blockDim.x) {
    ta[i] = fs_cur_elem(a_prev, prev_size, first, xp, i);
    if (MODE != FSC_BIN) tb[i] = fs_cur_elem(b_prev, prev_size, first, xp, i);
  }
  if (MODE != FSC_IP && r0 < k) {
    const uint32_t nv = k - r0 - 1;
    for (uint32_t j = tid; j < (1u << nv); j += blockDim.x) E[j] = fs_eq_product(w + r0 + 1, nv, j);
  }
  FsState st;
  if (tid == 0) st = *state;
  __syncthreads();
  for (uint32_t r = r0; r < k; r++) {
    const uint32_t osz = (sz + 1) / 2;
    Fr s0 = fr_zero(), s1 = fr_zero(), s2 = fr_zero();
    const uint32_t g = tid;
    const bool act = g < osz;
    Fr a0 = fr_zero(), a1 = fr_zero(), b0 = fr_zero(), b1 = fr_zero();
    if (act) {
      const uint32_t g0 = 2 * g, g1 = 2 * g + 1;
      a0 = ta[g0];
      a1 = g1 < sz ? ta[g1] : fr_zero();
      const Fr da = fr_sub(a1, a0);
      Fr c0, c1, c2;
      if (MODE == FSC_BIN) {
        c0 = fr_sub(fr_sqr(a0), a0);
        c1 = fr_sub(fr_mul(fr_dbl(a0), da), da);
        c2 = fr_sqr(da);
      } else {
        b0 = tb[g0];
        b1 = g1 < sz ? tb[g1] : fr_zero();
        const Fr db = fr_sub(b1, b0);
        c0 = fr_mul(a0, b0);
        c2 = fr_mul(da, db);
        c1 = fr_sub(fr_sub(fr_mul(a1, b1), c0), c2);
      }
      if (MODE != FSC_IP) {
        const Fr wgt = E[g];
        c0 = fr_mul(c0, wgt);
        c1 = fr_mul(c1, wgt);
        c2 = fr_mul(c2, wgt);
      }
      s0 = c0; s1 = c1; s2 = c2;
    }
    Fr ne = fr_zero();
    const uint32_t nE = (MODE != FSC_IP && r + 1 < k) ? (1u << (k - r - 2)) : 0u;
    if (g < nE) ne = fr_add(E[2 * g], E[2 * g + 1]);
    block_sum3_fr(s0, s1, s2, red);
    if (tid == 0) {
      proof[3 * r + 0] = s0;
      proof[3 * r + 1] = s1;
      proof[3 * r + 2] = s2;
      const Fr x = fs_round_step(st, s0, s1, s2);
      fold[r] = x;
      xs[0] = x;
    }
    __syncthreads();  // the challenge is visible; all reads of ta / tb / E of this round are done
    const Fr x = xs[0];
    if (act) {
      ta[g] = fr_add(a0, fr_mul(x, fr_sub(a1, a0)));  // a missing partner is zero: a0 - x a0
      if (MODE != FSC_BIN) tb[g] = fr_add(b0, fr_mul(x, fr_sub(b1, b0)));
    }
    if (g < nE) E[g] = ne;
    __syncthreads();
    sz = osz;
  }
  if (tid == 0) {
    proof[3 * k] = ta[0];
    if (MODE != FSC_BIN) proof[3 * k + 1] = tb[0];
    *state = st;
  }
}

// fold_d: k slots the round challenges are WRITTEN to; weight_d: the eq point (static), nullptr for the inner-product form
template <int MODE>
static int run_sumcheck_fs(zk_ctx* ctx, const Fr* a_in, const Fr* b_in, uint32_t n, FsState* state_d, Fr* fold_d, const Fr* weight_d, uint32_t k, Fr* proof_d,
                           bool b_is_bits = false) {
  const bool two = MODE != FSC_BIN;
  std::vector<uint32_t> sizes(k + 2);
  sizes[0] = n;
  for (uint32_t r = 0; r <= k; r++) sizes[r + 1] = (sizes[r] + 1) / 2;
  uint32_t nbig = 0;
  while (nbig < k && sizes[nbig] > FS_TAIL_N) nbig++;  // rounds whose table does not fit the tail kernel
  const uint32_t half = (n + 1) / 2;
  DBuf<Fr> abuf, bbuf, partial, tabA, tabB;
  uint32_t s = 0;
  if (nbig) {
    if (!abuf.alloc(ctx, (size_t)half + (half + 1) / 2 + 4) || (two && !bbuf.alloc(ctx, (size_t)half + (half + 1) / 2 + 4)) || !partial.alloc(ctx, 2 * 3 * FS_MAX_BLOCKS))
      return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck tables");
    if (MODE != FSC_IP) {
      s = nbig;
      if (!tabA.alloc(ctx, (size_t)1 << s) || !tabB.alloc(ctx, (size_t)1 << (k - s))) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck weights");
      const uint32_t tot = (1u << s) + (1u << (k - s));
      ZK_LAUNCHB(ctx, "sumcheck_prep", 32ull * tot, 256, k_fs_prep, dim3(stream_grid(tot, 256, 256)), 0, weight_d, k, s, tabA.get(), tabB.get());
    }
  }
  Fr* a_slots[2] = {abuf.get(), abuf.get() ? abuf.get() + half + 2 : nullptr};
  Fr* b_slots[2] = {bbuf.get(), bbuf.get() ? bbuf.get() + half + 2 : nullptr};
  const char* tag = MODE == FSC_IP ? "fs_ip_round" : MODE == FSC_HP ? "fs_hp_round" : "fs_bin_round";
  const Fr* a_prev = a_in;
  const Fr* b_prev = b_in;
  uint32_t prev_size = n;
  for (uint32_t r = 0; r < nbig; r++) {
    const uint32_t cur_size = sizes[r], npairs = sizes[r + 1];
    Fr* ac = r == 0 ? nullptr : a_slots[(r + 1) & 1];  // T_1 (ceil(n/2) entries) takes the large slot
    Fr* bc = r == 0 ? nullptr : b_slots[(r + 1) & 1];
    const uint32_t grid = stream_grid(npairs, 256, FS_MAX_BLOCKS);
    Fr* part = partial.get() + (r & 1) * 3 * FS_MAX_BLOCKS;
    const uint32_t c = MODE != FSC_IP ? s - 1 - r : 0;
    const Fr* A = MODE != FSC_IP ? tabA.get() + ((1u << s) - (1u << (s - r))) : nullptr;
    const uint64_t bytes = (two ? 2ull : 1ull) * 32ull * ((uint64_t)prev_size + (r ? cur_size : 0));
    ZK_LAUNCHB(ctx, tag, bytes, 256, k_fs_round<MODE>, dim3(grid), 0, a_prev, b_prev, ac, bc, (const Fr*)(r ? fold_d + (r - 1) : nullptr), A, c, (const Fr*)tabB.get(), prev_size,
               cur_size, npairs, part, b_is_bits ? 1u : 0u);
    ZK_LAUNCHB(ctx, "fs_hash", 96ull * grid, 256, k_fs_hash, dim3(1), 0, (const Fr*)part, grid, proof_d + 3 * r, state_d, fold_d + r);
    if (r > 0) {
      a_prev = ac;
      b_prev = bc;
      prev_size = cur_size;
    }
  }
  // the tail: input = the table before the pending fold by x_{nbig-1} (nbig > 0), or the input table
  const size_t lds = ((MODE == FSC_BIN ? 1 : 2) * FS_TAIL_N + FS_TAIL_N / 2 + 24 + 1) * sizeof(Fr);
  if (ctx->lds_first_use(zk::Kern<512, &k_fs_tail<MODE>>::desc())) {
    zk::Kern<512, &k_fs_tail<MODE>>::set_max_lds(lds);
    (void)hipGetLastError();
  }
  const char* ttag = MODE == FSC_IP ? "fs_ip_tail" : MODE == FSC_HP ? "fs_hp_tail" : "fs_bin_tail";
  ZK_LAUNCHB(ctx, ttag, (two ? 2ull : 1ull) * 32ull * prev_size, 512, k_fs_tail<MODE>, dim3(1), lds, a_prev, b_prev, prev_size, fold_d, weight_d, k, nbig, proof_d, state_d);
  return ZK_OK;
}

// state_0 of one sumcheck (host): SHA-256("zkdl-amd/fs-rounds/v1" | seed | layer | kind | which), integers little-endian
FsState fs_state0(const uint8_t seed32[32], uint32_t layer, uint32_t kind, uint32_t which) {
  static const char tag[] = "zkdl-amd/fs-rounds/v1";
  uint8_t msg[21 + 32 + 12];
  memcpy(msg, tag, 21);
  memcpy(msg + 21, seed32, 32);
  const uint32_t w[3] = {layer, kind, which};
  for (int q = 0; q < 3; q++)
    for (int b = 0; b < 4; b++) msg[53 + 4 * q + b] = (uint8_t)(w[q] >> (8 * b));
  uint8_t d[32];
  (void)zk_sha256(msg, sizeof msg, d);
  FsState st;
  for (int i = 0; i < 8; i++) st.h[i] = ((uint32_t)d[4 * i] << 24) | ((uint32_t)d[4 * i + 1] << 16) | ((uint32_t)d[4 * i + 2] << 8) | d[4 * i + 3];
  return st;
}

void fs_state0_slot(const uint8_t seed32[32], uint32_t layer, uint32_t kind, uint32_t which, zk_fr* slot) {
  const FsState st = fs_state0(seed32, layer, kind, which);
  for (int i = 0; i < 8; i++) slot->v[i] = st.h[i];
}
int sumcheck_ip_fs_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, Fr* fold_state_d, uint32_t k, Fr* proof_d) {
  if (n == 0 || k > 31 || n <= ((1u << k) / 2) || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // proof.cu:98-108
  return run_sumcheck_fs<FSC_IP>(ctx, a, b, n, reinterpret_cast<FsState*>(fold_state_d + k), fold_state_d, nullptr, k, proof_d);
}

struct FsGuard {  // the same roll-back as layers.hip's ProveGuard
  zk_ctx* ctx;
  zk_transcript* tr;
  size_t n_fr, n_g1, chains, jobs, keep, derives;
  FsGuard(zk_ctx* c, zk_transcript* t)
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

static void put_state(std::vector<zk_fr>& ch, const FsState& st) {
  zk_fr slot;
  for (int i = 0; i < 8; i++) slot.v[i] = st.h[i];
  ch.push_back(slot);
}

// zkFC::prove (zkfc.cu:128-145) with round-by-round challenges.  Transcript layout as fc_prove's.
static int fc_prove_fs_body(zk_ctx* ctx, zk_fc* fc, const Fr* x, uint32_t xn, const Fr* z, uint32_t zn, zk_gens* gens, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  const uint32_t n_gens = gens->n;
  if (xn % fc->in != 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions 1");
  const uint32_t batch = xn / fc->in;
  const uint32_t k_bs = ceil_log2(batch), k_in = ceil_log2(fc->in), k_out = ceil_log2(fc->out);
  const uint32_t wn = fc->in * fc->out;
  const uint32_t k_lo = ceil_log2(n_gens), k_hi = ceil_log2(fc->m);
  if (k_out + k_in < k_hi || k_out + k_in - k_hi != k_lo) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  uint32_t asz = fr_partial_me_size(xn, k_bs, fc->in), bsz = fr_partial_me_size(wn, k_out, 1);
  if (asz != bsz) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const size_t add_fr = 3 * k_in + 2 + 1 + 1, add_g1 = 1 + 3 * k_lo + 1;
  ZK_TRY(tr->ensure(add_fr, add_g1));
  Fr* const fr_slot = tr->fr_d + tr->n_fr;
  G1J* const g1_slot = tr->g1_d + tr->n_g1;
  tr->n_fr += add_fr;
  tr->n_g1 += add_g1;
  RecScope recording(ctx, &tr->rec, REC_GROUP_FC_A);
  if (recording.rc != ZK_OK) return recording.rc;
  LaneScope lane(ctx, recording.on ? ~0u : fc->serial);

  // [u_out | u_in (written by the sumcheck) | u_bs | state]; u_out, u_bs: statement-derived evaluation points (zk_fs_derive, vec 2 / 0)
  std::vector<zk_fr> ch(k_out + k_in + k_bs);
  if (k_out) ZK_TRY(zk_fs_derive(seed32, layer, 0, 2, k_out, ch.data()));
  for (uint32_t i = 0; i < k_in; i++) ch[k_out + i] = zk_fr{};
  if (k_bs) ZK_TRY(zk_fs_derive(seed32, layer, 0, 0, k_bs, ch.data() + k_out + k_in));
  put_state(ch, fs_state0(seed32, layer, 0, 0));
  DBuf<Fr> chd;
  ZK_TRY(upload_fr(ctx, chd, ch.data(), (uint32_t)ch.size()));
  Fr* u_out = chd.get();
  Fr* u_in = chd.get() + k_out;
  Fr* u_bs = chd.get() + k_out + k_in;
  FsState* state = reinterpret_cast<FsState*>(chd.get() + k_out + k_in + k_bs);
  DBuf<Fr> uz;
  {
    std::vector<zk_fr> z_(k_out + k_bs + 1);
    for (uint32_t i = 0; i < k_out; i++) z_[i] = ch[i];
    for (uint32_t i = 0; i < k_bs; i++) z_[k_out + i] = ch[k_out + k_in + i];
    ZK_TRY(upload_fr(ctx, uz, z_.data(), k_out + k_bs));
  }
  ZK_ALLOC(ctx, a, Fr, asz);
  ZK_ALLOC(ctx, b, Fr, bsz);
  ZK_TRY(fr_partial_me_dev(ctx, x, xn, u_bs, k_bs, fc->in, a.get()));
  if (fc->wq && k_out > 0) ZK_TRY(fr_partial_me_q16_dev(ctx, fc->wq, wn, u_out, k_out, 1, b.get()));
  else ZK_TRY(fr_partial_me_dev(ctx, fc->weights, wn, u_out, k_out, 1, b.get()));
  if (asz == 0 || k_in > 31 || asz <= ((1u << k_in) / 2) || asz > (1u << k_in)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // proof.cu:98-108
  ZK_TRY(run_sumcheck_fs<FSC_IP>(ctx, a.get(), b.get(), asz, state, u_in, nullptr, k_in, fr_slot));
  // the opening at u_out || u_in: behind the sumcheck, in the SAME chain (u_in are its round challenges)
  ZK_TRY(open_dev(ctx, gens, fc->weights, wn, fc->com, u_out, k_out + k_in, g1_slot, g1_slot + 1, fr_slot + 3 * k_in + 3, &tr->queue, fc->wq));
  // Z(u_out || u_bs): independent of the rounds
  rec_split(ctx, REC_GROUP_FC_OPEN);
  ZK_TRY(fr_me_dev(ctx, z, zn, uz.get(), k_out + k_bs, fr_slot + 3 * k_in + 2));
  return ZK_OK;
}
int fc_prove_fs(zk_ctx* ctx, zk_fc* fc, const Fr* x, uint32_t xn, const Fr* z, uint32_t zn, zk_gens* gens, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  FsGuard guard(ctx, tr);
  return guard.finish(fc_prove_fs_body(ctx, fc, x, xn, z, zn, gens, seed32, layer, tr));
}

// zkReLU::prove (zkrelu.cu:79-100) with round-by-round challenges, on the one-field-element-per-bit tensors
static int relu_prove_fs_body(zk_ctx* ctx, const zk_relu* relu, const Fr* x, uint32_t n, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  if (relu->n != n || n == 0) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");  // zkrelu.cu:81
  if ((uint64_t)n * 32 > 0xffffffffull) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t k = ceil_log2(n);
  const size_t add_fr = 3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 3 * k + 2;
  const bool packed = ctx->relu_packed && relu->mag_packed && relu->rem_packed;
  if (!packed) ZK_TRY(relu_materialise_bins(ctx, const_cast<zk_relu*>(relu)));  // before recording starts
  ZK_TRY(tr->ensure(add_fr, 0));
  Fr* out = tr->fr_d + tr->n_fr;
  tr->n_fr += add_fr;
  RecScope recording(ctx, &tr->rec, REC_GROUP_RELU_MAG);
  if (recording.rc != ZK_OK) return recording.rc;
  LaneScope lane(ctx, recording.on ? ~0u : relu->serial);

  // the reference's draw order (zkrelu.cu:85-98): u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp; the v's are the round challenges
  const uint32_t lens[7] = {k + 5, k + 5, k + 4, k + 4, k, k, k};
  const bool derived[7] = {true, false, true, false, true, true, false};
  std::vector<zk_fr> ch;
  uint32_t offs[7], tot = 0;
  for (int i = 0; i < 7; i++) {
    offs[i] = tot;
    ch.resize(tot + lens[i], zk_fr{});
    if (derived[i] && lens[i]) ZK_TRY(zk_fs_derive(seed32, layer, 1, (uint32_t)i, lens[i], ch.data() + tot));
    tot += lens[i];
  }
  for (uint32_t which = 1; which <= 3; which++) put_state(ch, fs_state0(seed32, layer, 1, which));
  DBuf<Fr> chd;
  ZK_TRY(upload_fr(ctx, chd, ch.data(), (uint32_t)ch.size()));
  Fr* c = chd.get();
  FsState* states = reinterpret_cast<FsState*>(c + tot);
  DBuf<Fr> t3_mag, t3_rem;  // declared behind the lane scope: released into the recording's deferred frees
  if (packed) {
    // the in-word rounds from the packed words (weighted byte histograms, the first taken before any challenge exists), the rest on a table of
    // one entry per word
    uint32_t R = 0;
    ZK_TRY(fsb_sumcheck_head_dev(ctx, relu->mag_packed, n, 32, c + offs[0], k + 5, states + 0, c + offs[1], out, c + offs[4], out + 3 * (k + 5) + 1, t3_mag, &R));
    ZK_TRY(run_sumcheck_fs<FSC_BIN>(ctx, t3_mag.get(), nullptr, n, states + 0, c + offs[1] + R, c + offs[0] + R, k + 5 - R, out + 3 * R));
    out += 3 * (k + 5) + 1 + 32;
    rec_split(ctx, REC_GROUP_RELU_REM);
    ZK_TRY(fsb_sumcheck_head_dev(ctx, relu->rem_packed, n, 16, c + offs[2], k + 4, states + 1, c + offs[3], out, c + offs[4], out + 3 * (k + 4) + 1, t3_rem, &R));
    ZK_TRY(run_sumcheck_fs<FSC_BIN>(ctx, t3_rem.get(), nullptr, n, states + 1, c + offs[3] + R, c + offs[2] + R, k + 4 - R, out + 3 * R));
    out += 3 * (k + 4) + 1 + 16;
  } else {
    ZK_TRY(run_sumcheck_fs<FSC_BIN>(ctx, relu->mag_bin, nullptr, n * 32, states + 0, c + offs[1], c + offs[0], k + 5, out));
    out += 3 * (k + 5) + 1;
    ZK_TRY(fr_partial_me_dev(ctx, relu->mag_bin, n * 32, c + offs[4], k, 32, out));
    out += 32;
    rec_split(ctx, REC_GROUP_RELU_REM);
    ZK_TRY(run_sumcheck_fs<FSC_BIN>(ctx, relu->rem_bin, nullptr, n * 16, states + 1, c + offs[3], c + offs[2], k + 4, out));
    out += 3 * (k + 4) + 1;
    ZK_TRY(fr_partial_me_dev(ctx, relu->rem_bin, n * 16, c + offs[4], k, 16, out));
    out += 16;
  }
  rec_split(ctx, REC_GROUP_FC_OPEN);
  if (k > 0 && n <= (1u << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions 3");
  ZK_TRY(run_sumcheck_fs<FSC_HP>(ctx, x, relu->sign, n, states + 2, c + offs[6], c + offs[5], k, out, true));
  return ZK_OK;
}
int relu_prove_fs(zk_ctx* ctx, const zk_relu* relu, const Fr* x, uint32_t n, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  FsGuard guard(ctx, tr);
  return guard.finish(relu_prove_fs_body(ctx, relu, x, n, seed32, layer, tr));
}
// zk_leaky_prove with round-by-round challenges: zkReLU's transcript, then hadamard_product_sumcheck(low, 1 - sign, u_lk, v_lk) with u_lk
// statement-derived (vec 7 of the layer's draw order) and v_lk the hash chain "which = 4"
int leaky_prove_fs(zk_ctx* ctx, const zk_relu* relu, const Fr* x, uint32_t n, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  if (relu->act_kind != 1 || !relu->low) return ctx->fail(ZK_ERR_ARG, "zk_leaky_prove_fs: the layer's last forward pass was not a LeakyReLU one");
  FsGuard guard(ctx, tr);
  int rc = relu_prove_fs_body(ctx, relu, x, n, seed32, layer, tr);
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
    std::vector<zk_fr> ch(2 * k, zk_fr{});
    if (k) ZK_TRY(zk_fs_derive(seed32, layer, 1, 7, k, ch.data()));
    put_state(ch, fs_state0(seed32, layer, 1, 4));
    DBuf<Fr> chd;
    ZK_TRY(upload_fr(ctx, chd, ch.data(), (uint32_t)ch.size()));
    Fr* c = chd.get();
    return run_sumcheck_fs<FSC_HP>(ctx, relu->low, relu->nsign, n, reinterpret_cast<FsState*>(c + 2 * k), c + k, c, k, out, true);
  };
  return guard.finish(body());
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_fc_prove_fs(zk_ctx* ctx, zk_fc* fc, const zk_fr* x_d, uint32_t xn, const zk_fr* z_d, uint32_t zn, zk_gens* gens, const uint8_t* seed32, uint32_t layer,
                   zk_transcript* tr) {
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (!fc || !tr || !gens || !seed32 || !x_d || !z_d) return ctx->fail(ZK_ERR_ARG, "zk_fc_prove_fs: null argument");
  if ((((uintptr_t)x_d) | ((uintptr_t)z_d)) & 15u) return ctx->fail(ZK_ERR_ARG, "misaligned device pointer");
  return fc_prove_fs(ctx, fc, reinterpret_cast<const Fr*>(x_d), xn, reinterpret_cast<const Fr*>(z_d), zn, gens, seed32, layer, tr);
}

int zk_relu_prove_fs(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x_d, const zk_fr* z_d, uint32_t n, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  (void)z_d;
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (!relu || !tr || !seed32 || !x_d) return ctx->fail(ZK_ERR_ARG, "zk_relu_prove_fs: null argument");
  if (((uintptr_t)x_d) & 15u) return ctx->fail(ZK_ERR_ARG, "misaligned device pointer");
  return relu_prove_fs(ctx, relu, reinterpret_cast<const Fr*>(x_d), n, seed32, layer, tr);
}

int zk_leaky_prove_fs(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x_d, const zk_fr* z_d, uint32_t n, const uint8_t* seed32, uint32_t layer, zk_transcript* tr) {
  (void)z_d;
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (!relu || !x_d || !seed32 || !tr) return ctx->fail(ZK_ERR_ARG, "zk_leaky_prove_fs: null argument");
  if (((uintptr_t)x_d) & 15u) return ctx->fail(ZK_ERR_ARG, "misaligned device pointer");
  return leaky_prove_fs(ctx, relu, reinterpret_cast<const Fr*>(x_d), n, seed32, layer, tr);
}

// The verifier's side of the round challenges: coeffs_h holds the k rounds' coefficient triples as the proof carries them.
int zk_fs_round_challenges(const uint8_t* seed32, uint32_t layer, uint32_t kind, uint32_t which, const zk_fr* coeffs_h, uint32_t k, zk_fr* out_h) {
  if (!seed32 || (k && (!coeffs_h || !out_h))) return ZK_ERR_ARG;
  FsState st = fs_state0(seed32, layer, kind, which);
  Fr rm;
  for (int i = 0; i < 8; i++) rm.v[i] = FrP::P(i);
  for (uint32_t r = 0; r < k; r++) {
    Fr c[3];
    memcpy(c, coeffs_h + 3 * r, sizeof c);
    // the limbs are hashed as stored: only the canonical encoding of a coefficient may enter the chain (c and c + r would be two transcripts for one element)
    if (l_gte(c[0], rm) || l_gte(c[1], rm) || l_gte(c[2], rm)) return ZK_ERR_RANGE;
    const Fr x = fs_round_step(st, c[0], c[1], c[2]);
    memcpy(out_h + r, &x, sizeof x);
  }
  return ZK_OK;
}

}  // extern "C"
