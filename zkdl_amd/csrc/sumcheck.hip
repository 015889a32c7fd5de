// zkdl_amd/csrc/sumcheck.hip -- inner-product / Hadamard / binary sumcheck provers for gfx950
// (reference: proof.cu:55-200).
//
// The reference runs, per round, a step kernel that materialises three coefficient tensors, then
// reduces each (sum, or a full log-depth MLE chain of launches), then folds the operand tables with
// two more launches (proof.cu:72-96,110-137,165-188): ~(6 + 3 log n) launches, ~6x the table in
// HBM traffic, a malloc/free pair per temporary and a device sync after every launch.  Here:
//
//  * ONE fused kernel per "big" round reads the live table(s) once and (a) accumulates the three
//    round coefficients -- plain sums for the inner-product variant, eq(u[r+1:], g)-weighted sums
//    for the Hadamard / binary variants -- through a wave64 shuffle ladder + one LDS stage into
//    per-workgroup partials, (b) writes the folded table(s), and (c) in workgroup 0, reduces the
//    PREVIOUS round's partials into the transcript (the kernel boundary is the grid-wide barrier,
//    so no atomics or flags are needed).
//  * Once the table fits LDS (<= 1024 elements) ONE single-workgroup kernel finishes all remaining
//    rounds with the tables resident in LDS (k_sc_tail).
//  * The eq weights of round r factor as E_r[g] = A_r[g mod 2^c_r] * B[g >> c_r] with a FIXED split
//    of the challenge vector: B = eq(u[s:]) never changes and A_r = eq(u[r+1:s)) shrinks by one
//    variable per round; every A_r and B is produced by one prep launch (k_sc_prep).  The tail
//    kernel keeps its own weight table in LDS and halves it each round (E'[g] = E[2g] + E[2g+1]).
//
// A length-2^24 binary sumcheck is 1 + 14 + 1 launches instead of the reference's ~2000.  Fr results
// are canonical, so this re-association is bit-identical to the reference's order.
#include "ctx.hpp"
#include "dev_util.hpp"

namespace zk {

enum { SC_IP = 0, SC_HP = 1, SC_BIN = 2 };
static constexpr uint32_t SC_TAIL_N = 1024;   // table length handled by the LDS-resident tail kernel
static constexpr uint32_t SC_TAIL_LOG = 10;
static constexpr uint32_t SC_MAX_BLOCKS = 1024;

__device__ __forceinline__ Fr eq_product(const Fr* __restrict__ u, uint32_t nvars, uint32_t j) {
  Fr one = fr_one(), acc = one;
  for (uint32_t b = 0; b < nvars; b++) {
    Fr ub = u[b];
    acc = fr_mul(acc, ((j >> b) & 1u) ? ub : fr_sub(one, ub));
  }
  return acc;
}

// Weight tables of all big rounds in one launch.  w = weight challenges (u), k variables, split s:
//   B[j]            = eq(w[s .. k), j)                      j < 2^(k-s)
//   A[aoff(r) + j]  = eq(w[r+1 .. s), j)                    j < 2^(s-1-r),  r = 0 .. s-1,  aoff(r) = 2^s - 2^(s-r)
__global__ void __launch_bounds__(256) k_sc_prep(const Fr* __restrict__ w, uint32_t k, uint32_t s, Fr* __restrict__ A, Fr* __restrict__ B) {
  const uint32_t nA = (1u << s) - 1u, nB = 1u << (k - s);
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nA + nB; idx += gridDim.x * blockDim.x) {
    if (idx < nA) {
      // level r holds 2^(s-1-r) entries at offset 2^s - 2^(s-r):  2^(s-r-1) < 2^s - idx <= 2^(s-r)
      uint32_t rem = (1u << s) - idx;  // >= 2
      uint32_t r = s - (32u - __clz(rem - 1u));
      uint32_t j = idx - ((1u << s) - (1u << (s - r)));
      A[idx] = eq_product(w + r + 1, s - 1 - r, j);
    } else {
      uint32_t j = idx - nA;
      B[j] = eq_product(w + s, k - s, j);
    }
  }
}

// One big sumcheck round (see file header).  x = fold challenge of this round, read from device memory.
template <int MODE>
__global__ void __launch_bounds__(256) k_sc_round(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ a_out, Fr* __restrict__ b_out,
                                                   const Fr* __restrict__ xp, const Fr* __restrict__ A, const Fr* __restrict__ B, uint32_t c,
                                                   uint32_t in_size, uint32_t out_size, Fr* __restrict__ partial,
                                                   const Fr* __restrict__ prev_partial, uint32_t prev_blocks, Fr* __restrict__ prev_dst) {
  __shared__ Fr lds[4];
  const Fr x = *xp;
  const uint32_t mask = (1u << c) - 1u;
  Fr s0 = fr_zero(), s1 = fr_zero(), s2 = fr_zero();
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < out_size; g += gridDim.x * blockDim.x) {
    const uint32_t g0 = 2 * g, g1 = 2 * g + 1;
    Fr a0 = a[g0];  // g0 < in_size always holds for g < out_size
    Fr a1 = g1 < in_size ? a[g1] : fr_zero();
    Fr da = fr_sub(a1, a0);
    Fr c0, c1, c2;
    if (MODE == SC_BIN) {
      c0 = fr_sub(fr_sqr(a0), a0);
      c1 = fr_sub(fr_mul(fr_dbl(a0), da), da);
      c2 = fr_sqr(da);
    } else {
      Fr b0 = b[g0];
      Fr b1 = g1 < in_size ? b[g1] : fr_zero();
      Fr db = fr_sub(b1, b0);
      c0 = fr_mul(a0, b0);
      c1 = fr_add(fr_mul(a0, db), fr_mul(b0, da));
      c2 = fr_mul(da, db);
      b_out[g] = g1 < in_size ? fr_add(b0, fr_mul(x, db)) : fr_sub(b0, fr_mul(x, b0));  // Fr_me_step incl. odd tail
    }
    a_out[g] = g1 < in_size ? fr_add(a0, fr_mul(x, da)) : fr_sub(a0, fr_mul(x, a0));
    if (MODE != SC_IP) {
      Fr w = fr_mul(A[g & mask], B[g >> c]);
      c0 = fr_mul(c0, w);
      c1 = fr_mul(c1, w);
      c2 = fr_mul(c2, w);
    }
    s0 = fr_add(s0, c0);
    s1 = fr_add(s1, c1);
    s2 = fr_add(s2, c2);
  }
  s0 = block_sum_fr(s0, lds);
  s1 = block_sum_fr(s1, lds);
  s2 = block_sum_fr(s2, lds);
  if (threadIdx.x == 0) {
    partial[3 * blockIdx.x + 0] = s0;
    partial[3 * blockIdx.x + 1] = s1;
    partial[3 * blockIdx.x + 2] = s2;
  }
  // workgroup 0 also closes the previous round: its partials were completed by the previous launch
  if (blockIdx.x == 0 && prev_blocks) {
    for (uint32_t q = 0; q < 3; q++) {
      Fr acc = fr_zero();
      for (uint32_t i = threadIdx.x; i < prev_blocks; i += blockDim.x) acc = fr_add(acc, prev_partial[3 * i + q]);
      acc = block_sum_fr(acc, lds);
      if (threadIdx.x == 0) prev_dst[q] = acc;
    }
  }
}

// All remaining rounds with the table(s) resident in LDS.  n <= SC_TAIL_N, rounds r0 .. k-1.
// fold = fold challenges (all k), w = weight challenges (all k) or nullptr for SC_IP.
// Also closes the last big round (prev_*), and writes the final a(0) [, b(0)].
template <int MODE>
__global__ void __launch_bounds__(512) k_sc_tail(const Fr* __restrict__ a_in, const Fr* __restrict__ b_in, uint32_t n, const Fr* __restrict__ fold,
                                                  const Fr* __restrict__ w, uint32_t k, uint32_t r0, Fr* __restrict__ proof,
                                                  const Fr* __restrict__ prev_partial, uint32_t prev_blocks, Fr* __restrict__ prev_dst) {
  extern __shared__ Fr sm[];
  Fr* ta = sm;                                              // SC_TAIL_N
  Fr* tb = sm + SC_TAIL_N;                                  // SC_TAIL_N (unused for SC_BIN)
  Fr* E = sm + (MODE == SC_BIN ? SC_TAIL_N : 2 * SC_TAIL_N);  // SC_TAIL_N / 2 weights
  Fr* red = E + SC_TAIL_N / 2;                              // 8 reduction slots
  const uint32_t tid = threadIdx.x;
  if (prev_blocks) {
    for (uint32_t q = 0; q < 3; q++) {
      Fr acc = fr_zero();
      for (uint32_t i = tid; i < prev_blocks; i += blockDim.x) acc = fr_add(acc, prev_partial[3 * i + q]);
      acc = block_sum_fr(acc, red);
      if (tid == 0) prev_dst[q] = acc;
    }
  }
  for (uint32_t i = tid; i < n; i += blockDim.x) {
    ta[i] = a_in[i];
    if (MODE != SC_BIN) tb[i] = b_in[i];
  }
  uint32_t sz = n;
  if (MODE != SC_IP && r0 < k) {
    // weights of round r0 over the remaining variables w[r0+1 .. k)
    const uint32_t nv = k - r0 - 1;
    for (uint32_t j = tid; j < (1u << nv); j += blockDim.x) E[j] = eq_product(w + r0 + 1, nv, j);
  }
  __syncthreads();
  for (uint32_t r = r0; r < k; r++) {
    const uint32_t osz = (sz + 1) / 2;
    const Fr x = fold[r];
    Fr s0 = fr_zero(), s1 = fr_zero(), s2 = fr_zero();
    // with <= 512 outputs every lane owns at most one pair: read, barrier, then fold in place
    const uint32_t g = tid;
    Fr na = fr_zero(), nb = fr_zero();
    const bool act = g < osz;
    if (act) {
      const uint32_t g0 = 2 * g, g1 = 2 * g + 1;
      Fr a0 = ta[g0];
      Fr a1 = g1 < sz ? ta[g1] : fr_zero();
      Fr da = fr_sub(a1, a0);
      Fr c0, c1, c2;
      if (MODE == SC_BIN) {
        c0 = fr_sub(fr_sqr(a0), a0);
        c1 = fr_sub(fr_mul(fr_dbl(a0), da), da);
        c2 = fr_sqr(da);
      } else {
        Fr b0 = tb[g0];
        Fr b1 = g1 < sz ? tb[g1] : fr_zero();
        Fr db = fr_sub(b1, b0);
        c0 = fr_mul(a0, b0);
        c1 = fr_add(fr_mul(a0, db), fr_mul(b0, da));
        c2 = fr_mul(da, db);
        nb = g1 < sz ? fr_add(b0, fr_mul(x, db)) : fr_sub(b0, fr_mul(x, b0));
      }
      na = g1 < sz ? fr_add(a0, fr_mul(x, da)) : fr_sub(a0, fr_mul(x, a0));
      if (MODE != SC_IP) {
        Fr wgt = E[g];
        c0 = fr_mul(c0, wgt);
        c1 = fr_mul(c1, wgt);
        c2 = fr_mul(c2, wgt);
      }
      s0 = c0; s1 = c1; s2 = c2;
    }
    // next round's weights: sum out the variable that pairs g: E'[j] = E[2j] + E[2j+1]
    Fr ne = fr_zero();
    const uint32_t nE = (MODE != SC_IP && r + 1 < k) ? (1u << (k - r - 2)) : 0u;
    if (g < nE) ne = fr_add(E[2 * g], E[2 * g + 1]);
    s0 = block_sum_fr(s0, red);
    s1 = block_sum_fr(s1, red);
    s2 = block_sum_fr(s2, red);
    if (tid == 0) {
      proof[3 * r + 0] = s0;
      proof[3 * r + 1] = s1;
      proof[3 * r + 2] = s2;
    }
    __syncthreads();  // all reads of ta/tb/E of this round are done
    if (act) {
      ta[g] = na;
      if (MODE != SC_BIN) tb[g] = nb;
    }
    if (g < nE) E[g] = ne;
    __syncthreads();
    sz = osz;
  }
  if (tid == 0) {  // proof.cu:75-79,115-119,168-171
    proof[3 * k] = ta[0];
    if (MODE != SC_BIN) proof[3 * k + 1] = tb[0];
  }
}

template <int MODE>
static int run_sumcheck(zk_ctx* ctx, const Fr* a_in, const Fr* b_in, uint32_t n, const Fr* fold_d, const Fr* weight_d, uint32_t k, Fr* proof_d) {
  const bool two = (MODE != SC_BIN);
  // big rounds: while the live table is longer than SC_TAIL_N
  uint32_t nbig = 0;
  {
    uint32_t sz = n;
    while (sz > SC_TAIL_N && nbig < k) {
      sz = (sz + 1) / 2;
      nbig++;
    }
  }
  const uint32_t half = (n + 1) / 2;
  DBuf<Fr> abuf, bbuf, partial, tabA, tabB;
  uint32_t s = 0;
  if (nbig) {
    if (!abuf.alloc(ctx, (size_t)half + (half + 1) / 2 + 2) || (two && !bbuf.alloc(ctx, (size_t)half + (half + 1) / 2 + 2)) || !partial.alloc(ctx, 2 * 3 * SC_MAX_BLOCKS))
      return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck tables");
    if (MODE != SC_IP) {
      // fixed split: every big round r (< nbig) needs c_r = s - 1 - r >= 0
      s = nbig;
      if (s > k) s = k;
      if (!tabA.alloc(ctx, (size_t)1 << s) || !tabB.alloc(ctx, (size_t)1 << (k - s))) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck weights");
      uint32_t tot = (1u << s) + (1u << (k - s));
      ZK_LAUNCH(ctx, "sumcheck_prep", 32ull * tot, k_sc_prep, dim3(stream_grid(tot, 256, 256)), dim3(256), 0, weight_d, k, s, tabA.get(), tabB.get());
    }
  }
  const Fr* a = a_in;
  const Fr* b = b_in;
  uint32_t sz = n;
  Fr* a_slots[2] = {abuf.get(), abuf.get() ? abuf.get() + half + 1 : nullptr};
  Fr* b_slots[2] = {bbuf.get(), bbuf.get() ? bbuf.get() + half + 1 : nullptr};
  const char* tag = MODE == SC_IP ? "sumcheck_ip_round" : MODE == SC_HP ? "sumcheck_hp_round" : "sumcheck_bin_round";
  const Fr* prev_partial = nullptr;
  uint32_t prev_blocks = 0;
  Fr* prev_dst = nullptr;
  for (uint32_t r = 0; r < nbig; r++) {
    uint32_t osz = (sz + 1) / 2;
    uint32_t c = MODE == SC_IP ? 0 : s - 1 - r;
    const Fr* Ar = MODE == SC_IP ? nullptr : tabA.get() + ((1u << s) - (1u << (s - r)));
    uint32_t grid = stream_grid(osz, 256, SC_MAX_BLOCKS);
    Fr* ao = a_slots[r & 1];
    Fr* bo = b_slots[r & 1];
    Fr* part = partial.get() + (r & 1) * 3 * SC_MAX_BLOCKS;
    uint64_t bytes = (two ? 2ull : 1ull) * 32ull * ((uint64_t)sz + osz);
    ZK_LAUNCH(ctx, tag, bytes, k_sc_round<MODE>, dim3(grid), dim3(256), 0, a, b, ao, bo, fold_d + r, Ar, (const Fr*)tabB.get(), c, sz, osz, part, prev_partial, prev_blocks,
              prev_dst);
    prev_partial = part;
    prev_blocks = grid;
    prev_dst = proof_d + 3 * r;
    a = ao;
    b = bo;
    sz = osz;
  }
  size_t lds = ((MODE == SC_BIN ? 1 : 2) * SC_TAIL_N + SC_TAIL_N / 2 + 8) * sizeof(Fr);
  static bool lds_attr_set = false;  // > 64 KiB of dynamic LDS needs the opt-in on some runtimes
  if (!lds_attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sc_tail<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    lds_attr_set = true;
  }
  const char* ttag = MODE == SC_IP ? "sumcheck_ip_tail" : MODE == SC_HP ? "sumcheck_hp_tail" : "sumcheck_bin_tail";
  ZK_LAUNCH(ctx, ttag, (two ? 2ull : 1ull) * 32ull * sz, k_sc_tail<MODE>, dim3(1), dim3(512), lds, a, b, sz, fold_d, weight_d, k, nbig, proof_d, prev_partial, prev_blocks, prev_dst);
  return ZK_OK;
}

// inner_product_sumcheck guards: proof.cu:98-108
int sumcheck_ip_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, const Fr* u_d, uint32_t k, Fr* proof_d) {
  if (n == 0 || k > 31 || n <= ((1u << k) / 2) || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  return run_sumcheck<SC_IP>(ctx, a, b, n, u_d, nullptr, k, proof_d);
}
// hadamard_product_sumcheck guards: proof.cu:139-150
int sumcheck_hp_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d) {
  if (n == 0 || k > 31 || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions 4");
  if (k > 0 && n <= (1u << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions 3");
  return run_sumcheck<SC_HP>(ctx, a, b, n, v_d, u_d, k, proof_d);
}
// binary_sumcheck guards: proof.cu:190-200
int sumcheck_bin_dev(zk_ctx* ctx, const Fr* a, uint32_t n, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d) {
  if (n == 0 || k > 31 || n <= ((1u << k) / 2) || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  return run_sumcheck<SC_BIN>(ctx, a, nullptr, n, v_d, u_d, k, proof_d);
}

}  // namespace zk
