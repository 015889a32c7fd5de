// zkdl_amd/csrc/sumcheck.hip -- inner-product / Hadamard / binary sumcheck provers for gfx950
// (reference: proof.cu:55-200).
//
// Round structure.  In round r the reference launches a step kernel that materialises three
// coefficient tensors, reduces each (sum, or a full log-depth MLE chain), then folds the operand
// tables with two more launches (proof.cu:72-96,110-137,165-188).  Here one fused kernel per
// round reads the live table(s) ONCE and (a) accumulates the three round coefficients -- plain
// sums for the inner-product variant, eq(u[r+1:], g)-weighted sums for the Hadamard/binary
// variants -- through a wave64 shuffle ladder + one LDS stage into per-block partials, and
// (b) writes the folded table(s).  A second tiny launch reduces the partials into the transcript.
// Fr arithmetic is canonical, so the re-association is bit-identical to the reference order.
//
// The eq weights of round r are E_r[g] = A[g mod 2^c] * B[g >> c] with c = min(10, #vars left):
// two small tables rebuilt per round from the challenge suffix (k_eq_table in fr_kernels.hip is
// reused through sc_eq_tables below), so no table as large as the operand is ever stored.
#include "ctx.hpp"
#include "dev_util.hpp"

namespace zk {

__global__ void __launch_bounds__(256) k_sc_eq(const Fr* __restrict__ u, uint32_t c, Fr* __restrict__ E) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (1u << c)) return;
  Fr one = fr_one();
  Fr acc = one;
  for (uint32_t b = 0; b < c; b++) {
    Fr ub = u[b];
    acc = fr_mul(acc, ((j >> b) & 1u) ? ub : fr_sub(one, ub));
  }
  E[j] = acc;
}

enum { SC_IP = 0, SC_HP = 1, SC_BIN = 2 };

// One sumcheck round.  in_size = live table length, out_size = (in_size+1)/2.
//   MODE SC_IP : coefficients of a(t) b(t), unweighted          (Fr_ip_sc_step + 3x sum)
//   MODE SC_HP : same coefficients, weighted by A[g & mask] B[g >> c]   (Fr_ip_sc_step + 3x MLE)
//   MODE SC_BIN: coefficients of a(t)^2 - a(t), weighted         (Fr_bin_sc_step + 3x MLE)
// x = fold challenge of this round (u[r] for IP, v[r] otherwise), read from device memory.
// partial[3*blockIdx.x + {0,1,2}] receives the block's partial coefficient sums.
template <int MODE>
__global__ void __launch_bounds__(256) k_sc_round(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ a_out, Fr* __restrict__ b_out,
                                                   const Fr* __restrict__ xp, const Fr* __restrict__ A, const Fr* __restrict__ B, uint32_t c,
                                                   uint32_t in_size, uint32_t out_size, Fr* __restrict__ partial) {
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
      // fold b exactly as Fr_me_step does (odd tail: b0 - x*b0)
      b_out[g] = g1 < in_size ? fr_add(b0, fr_mul(x, db)) : fr_sub(b0, fr_mul(x, b0));
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
}

// dst[0..2] = column sums of partial[nblk][3]
__global__ void __launch_bounds__(256) k_sc_finish(const Fr* __restrict__ partial, uint32_t nblk, Fr* __restrict__ dst) {
  __shared__ Fr lds[4];
  for (uint32_t q = 0; q < 3; q++) {
    Fr acc = fr_zero();
    for (uint32_t i = threadIdx.x; i < nblk; i += blockDim.x) acc = fr_add(acc, partial[3 * i + q]);
    acc = block_sum_fr(acc, lds);
    if (threadIdx.x == 0) dst[q] = acc;
  }
}

template <int MODE>
static int run_sumcheck(zk_ctx* ctx, const Fr* a_in, const Fr* b_in, uint32_t n, const Fr* fold_d, const Fr* weight_d, uint32_t k, Fr* proof_d) {
  // tables ping-pong between two halves of one scratch buffer
  uint32_t half = (n + 1) / 2;
  const bool two = (MODE != SC_BIN);
  ZK_ALLOC(ctx, abuf, Fr, (size_t)half + (half + 1) / 2 + 2);
  DBuf<Fr> bbuf;
  if (two && !bbuf.alloc(ctx, (size_t)half + (half + 1) / 2 + 2)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck b");
  const uint32_t max_blocks = 1024;
  ZK_ALLOC(ctx, partial, Fr, 3 * max_blocks);
  ZK_ALLOC(ctx, tabA, Fr, 1024);
  ZK_ALLOC(ctx, tabB, Fr, ((size_t)1 << (k > 11 ? k - 11 : 0)) + 2);

  const Fr* a = a_in;
  const Fr* b = b_in;
  uint32_t sz = n;
  Fr* a_slots[2] = {abuf.get(), abuf.get() + half + 1};
  Fr* b_slots[2] = {two ? bbuf.get() : nullptr, two ? bbuf.get() + half + 1 : nullptr};
  const char* tag = MODE == SC_IP ? "sumcheck_ip_round" : MODE == SC_HP ? "sumcheck_hp_round" : "sumcheck_bin_round";
  for (uint32_t r = 0; r < k; r++) {
    uint32_t osz = (sz + 1) / 2;
    uint32_t c = 0;
    if (MODE != SC_IP) {
      // weights over the remaining variables u[r+1 .. k): low c of them in A, the rest in B
      uint32_t rem = k - r - 1;
      c = rem < 10 ? rem : 10;
      uint32_t hb = rem - c;
      ZK_LAUNCH(ctx, "eq_table", 32ull << c, k_sc_eq, dim3(((1u << c) + 255) / 256), dim3(256), 0, weight_d + r + 1, c, tabA.get());
      ZK_LAUNCH(ctx, "eq_table", 32ull << hb, k_sc_eq, dim3(((1u << hb) + 255) / 256), dim3(256), 0, weight_d + r + 1 + c, hb, tabB.get());
    }
    uint32_t grid = stream_grid(osz, 256, max_blocks);
    Fr* ao = a_slots[r & 1];
    Fr* bo = b_slots[r & 1];
    uint64_t bytes = (two ? 2ull : 1ull) * 32ull * ((uint64_t)sz + osz);
    ZK_LAUNCH(ctx, tag, bytes, k_sc_round<MODE>, dim3(grid), dim3(256), 0, a, b, ao, bo, fold_d + r, (const Fr*)tabA.get(), (const Fr*)tabB.get(), c, sz, osz,
              partial.get());
    ZK_LAUNCH(ctx, "sumcheck_finish", 96ull * grid, k_sc_finish, dim3(1), dim3(256), 0, (const Fr*)partial.get(), grid, proof_d + 3 * r);
    a = ao;
    b = bo;
    sz = osz;
  }
  // tail: a(0) [, b(0)]  (proof.cu:75-79,115-119,168-171)
  ZK_HIP(ctx, hipMemcpyAsync(proof_d + 3 * k, a, sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
  if (two) ZK_HIP(ctx, hipMemcpyAsync(proof_d + 3 * k + 1, b, sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
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
