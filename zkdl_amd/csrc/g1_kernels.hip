// zkdl_amd/csrc/g1_kernels.hip -- G1TensorAffine / G1TensorJacobian kernels for gfx950
// (reference: g1-tensor.cu).  These are the operator-level kernels that keep the reference's
// association order, so their Jacobian limbs match it bit for bit; the re-associated fast paths
// (bucket MSM, flattened opening) live in msm.hip.
//
// Work per element is ~10^4 integer MACs against 144-288 B of traffic: every kernel here is bound by
// integer VALU throughput (v_mad_u64_u32), not HBM.  One wave64 per workgroup, as in the reference
// (G1NumThread = 64, g1-tensor.cuh:11), which also makes the sum tree below identical to its.
#include "ctx.hpp"
#include "dev_util.hpp"
#include "fpinv.hpp"

namespace zk {

static constexpr uint32_t G1T = 64;
static inline dim3 g1_grid(uint32_t n) { return dim3((n + G1T - 1) / G1T); }

// g1-tensor.cu:60-65,128-133
template <class P>
__global__ void __launch_bounds__(64) k_g1_fill(P* out, P g, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = g;
}
// g1-tensor.cu:142-147
__global__ void __launch_bounds__(64) k_g1_from_affine(const G1A* a, G1J* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1J r;
  r.x = a[i].x;
  r.y = a[i].y;
  r.z = fp_one();
  out[i] = r;
}
// g1-tensor.cu:93-98,169-174
__global__ void __launch_bounds__(64) k_g1_neg(const G1J* a, G1J* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = g1_neg(a[i]);
}
__global__ void __launch_bounds__(64) k_g1a_neg(const G1A* a, G1A* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = g1a_neg(a[i]);
}
// g1-tensor.cu:184-210,276-302.  SUB negates the second operand first, as the reference does.
template <bool SUB>
__global__ void __launch_bounds__(64) k_g1_add(const G1J* a, const G1J* b, G1J* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1J y = b[i];
  if (SUB) y = g1_neg(y);
  out[i] = g1_add(a[i], y);
}
template <bool SUB>
__global__ void __launch_bounds__(64) k_g1_add_bcast(const G1J* a, G1J x, G1J* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (SUB) x = g1_neg(x);
  out[i] = g1_add(a[i], x);
}
template <bool SUB>
__global__ void __launch_bounds__(64) k_g1_madd(const G1J* a, const G1A* b, G1J* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1A y = b[i];
  if (SUB) y = g1a_neg(y);
  out[i] = g1_add_mixed(a[i], y);
}
template <bool SUB>
__global__ void __launch_bounds__(64) k_g1_madd_bcast(const G1J* a, G1A x, G1J* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (SUB) x = g1a_neg(x);
  out[i] = g1_add_mixed(a[i], x);
}

int g1_fill_dev(zk_ctx* ctx, G1J* out, uint32_t n, const G1J& g) {
  if (n) ZK_LAUNCH(ctx, "g1_elementwise", 144ull * n, k_g1_fill<G1J>, g1_grid(n), dim3(G1T), 0, out, g, n);
  return ZK_OK;
}
int g1a_fill_dev(zk_ctx* ctx, G1A* out, uint32_t n, const G1A& g) {
  if (n) ZK_LAUNCH(ctx, "g1_elementwise", 96ull * n, k_g1_fill<G1A>, g1_grid(n), dim3(G1T), 0, out, g, n);
  return ZK_OK;
}
int g1_from_affine_dev(zk_ctx* ctx, const G1A* a, G1J* out, uint32_t n) {
  if (n) ZK_LAUNCH(ctx, "g1_elementwise", 240ull * n, k_g1_from_affine, g1_grid(n), dim3(G1T), 0, a, out, n);
  return ZK_OK;
}
int g1_neg_dev(zk_ctx* ctx, const G1J* a, G1J* out, uint32_t n) {
  if (n) ZK_LAUNCH(ctx, "g1_elementwise", 288ull * n, k_g1_neg, g1_grid(n), dim3(G1T), 0, a, out, n);
  return ZK_OK;
}
int g1a_neg_dev(zk_ctx* ctx, const G1A* a, G1A* out, uint32_t n) {
  if (n) ZK_LAUNCH(ctx, "g1_elementwise", 192ull * n, k_g1a_neg, g1_grid(n), dim3(G1T), 0, a, out, n);
  return ZK_OK;
}
int g1_addsub_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1J* b, G1J* out, uint32_t n) {
  if (!n) return ZK_OK;
  if (sub) ZK_LAUNCH(ctx, "g1_elementwise", 432ull * n, k_g1_add<true>, g1_grid(n), dim3(G1T), 0, a, b, out, n);
  else ZK_LAUNCH(ctx, "g1_elementwise", 432ull * n, k_g1_add<false>, g1_grid(n), dim3(G1T), 0, a, b, out, n);
  return ZK_OK;
}
int g1_maddsub_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1A* b, G1J* out, uint32_t n) {
  if (!n) return ZK_OK;
  if (sub) ZK_LAUNCH(ctx, "g1_elementwise", 384ull * n, k_g1_madd<true>, g1_grid(n), dim3(G1T), 0, a, b, out, n);
  else ZK_LAUNCH(ctx, "g1_elementwise", 384ull * n, k_g1_madd<false>, g1_grid(n), dim3(G1T), 0, a, b, out, n);
  return ZK_OK;
}
int g1_addsub_bcast_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1J& x, G1J* out, uint32_t n) {
  if (!n) return ZK_OK;
  if (sub) ZK_LAUNCH(ctx, "g1_elementwise", 288ull * n, k_g1_add_bcast<true>, g1_grid(n), dim3(G1T), 0, a, x, out, n);
  else ZK_LAUNCH(ctx, "g1_elementwise", 288ull * n, k_g1_add_bcast<false>, g1_grid(n), dim3(G1T), 0, a, x, out, n);
  return ZK_OK;
}
int g1_maddsub_bcast_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1A& x, G1J* out, uint32_t n) {
  if (!n) return ZK_OK;
  if (sub) ZK_LAUNCH(ctx, "g1_elementwise", 288ull * n, k_g1_madd_bcast<true>, g1_grid(n), dim3(G1T), 0, a, x, out, n);
  else ZK_LAUNCH(ctx, "g1_elementwise", 288ull * n, k_g1_madd_bcast<false>, g1_grid(n), dim3(G1T), 0, a, x, out, n);
  return ZK_OK;
}

// ------------------------------------------------------------------ sum (g1-tensor.cu:368-420)
// One level of the reference's tree: a 64-thread block covers 128 inputs, s[t] = in[i] (+ in[i+64]),
// then the 32..1 ladder; grid = ceil(cur/64), so trailing blocks emit the canonical infinity.
__global__ void __launch_bounds__(64) k_g1_sum_level(const G1J* __restrict__ in, G1J* __restrict__ out, uint32_t n) {
  __shared__ G1J s[64];
  const uint32_t t = threadIdx.x;
  const uint32_t i = blockIdx.x * 128 + t;
  G1J v = i < n ? in[i] : g1_zero();
  if (i + 64 < n) v = g1_add(v, in[i + 64]);
  s[t] = v;
  __syncthreads();
  for (uint32_t st = 32; st > 0; st >>= 1) {
    if (t < st) s[t] = g1_add(s[t], s[t + st]);
    __syncthreads();
  }
  if (t == 0) out[blockIdx.x] = s[0];
}
int g1_sum_dev(zk_ctx* ctx, const G1J* a, uint32_t n, G1J* dst_d) {
  if (n == 0) {
    G1J z = g1_zero();
    return g1_fill_dev(ctx, dst_d, 1, z);
  }
  if (n == 1) {
    return dev_copy(ctx, dst_d, a, sizeof(G1J));
  }
  uint32_t first = (n + 63) / 64;
  ZK_ALLOC(ctx, buf0, G1J, first);
  ZK_ALLOC(ctx, buf1, G1J, (first + 63) / 64 + 1);
  const G1J* cur = a;
  uint32_t cur_n = n;
  G1J* outs[2] = {buf0.get(), buf1.get()};
  int which = 0;
  while (cur_n > 1) {
    uint32_t grid = (cur_n + 63) / 64;
    G1J* out = grid == 1 ? dst_d : outs[which];
    ZK_LAUNCH(ctx, "g1_sum", 144ull * cur_n, k_g1_sum_level, dim3(grid), dim3(64), 0, cur, out, cur_n);
    cur = out;
    cur_n = grid;
    which ^= 1;
  }
  return ZK_OK;
}

// ------------------------------------------------------------------ scalar-mul (g1-tensor.cu:433-461)
__global__ void __launch_bounds__(64) k_g1_mul(const G1J* __restrict__ g, const Fr* __restrict__ s, G1J* __restrict__ out, uint32_t n_g, uint32_t total) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  out[i] = g1_mul_naive(g[i % n_g], s[i]);
}
int g1_mul_dev(zk_ctx* ctx, const G1J* g, uint32_t n_g, const Fr* s, uint32_t total, G1J* out) {
  if (total) ZK_LAUNCH(ctx, "g1_scalar_mul", 320ull * total, k_g1_mul, g1_grid(total), dim3(G1T), 0, g, s, out, n_g, total);
  return ZK_OK;
}

// ------------------------------------------------------------------ G1 multilinear evaluation
// G1_me_step, g1-tensor.cu:463-475: out = P0 + unmont(x) * (P1 - P0); odd tail P0 - unmont(x)*P0
__global__ void __launch_bounds__(64) k_g1_me_step(const G1J* __restrict__ in, G1J* __restrict__ out, const Fr* __restrict__ xp, uint32_t in_size, uint32_t out_size) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= out_size) return;
  Fr xu = fr_unmont(*xp);
  uint32_t g0 = 2 * g, g1 = 2 * g + 1;
  G1J r;
  if (g1 < in_size) r = g1_add(in[g0], g1_mul_naive(g1_add(in[g1], g1_neg(in[g0])), xu));
  else if (g0 < in_size) r = g1_add(in[g0], g1_neg(g1_mul_naive(in[g0], xu)));
  else r = g1_zero();
  out[g] = r;
}
// literal fold chain (g1-tensor.cu:477-491); the MSM form used by the fast path is in msm.hip
int g1_me_dev(zk_ctx* ctx, const G1J* t, uint32_t n, const Fr* u_d, uint32_t k, G1J* dst_d) {
  if (n == 0 || k > 31 || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (k > 0 && n <= (1u << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (k == 0) {
    return dev_copy(ctx, dst_d, t, sizeof(G1J));
  }
  uint32_t half = (n + 1) / 2;
  ZK_ALLOC(ctx, buf0, G1J, half);
  ZK_ALLOC(ctx, buf1, G1J, (half + 1) / 2 + 1);
  G1J* outs[2] = {buf0.get(), buf1.get()};
  const G1J* cur = t;
  uint32_t sz = n;
  for (uint32_t r = 0; r < k; r++) {
    uint32_t osz = (sz + 1) / 2;
    G1J* out = (r + 1 == k) ? dst_d : outs[r & 1];
    ZK_LAUNCH(ctx, "g1_me_step", 144ull * ((uint64_t)sz + osz), k_g1_me_step, g1_grid(osz), dim3(G1T), 0, cur, out, u_d + r, sz, osz);
    cur = out;
    sz = osz;
  }
  return ZK_OK;
}

// ------------------------------------------------------------------ affine-canonical form
__global__ void __launch_bounds__(64) k_g1_to_affine(const G1J* __restrict__ a, uint32_t n, G1A* __restrict__ out, uint8_t* __restrict__ inf) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1J p = a[i];
  G1A r;
  if (l_is_zero(p.z)) {
    r.x = fp_zero();
    r.y = fp_zero();
    inf[i] = 1;
  } else if (l_eq(p.z, fp_one())) {  // already canonical (every MSM result is): nothing to invert
    r.x = p.x;
    r.y = p.y;
    inf[i] = 0;
  } else {
    Fp zi = fp_inv(p.z);
    Fp zi2 = fp_sqr(zi);
    r.x = fp_mul(p.x, zi2);
    r.y = fp_mul(p.y, fp_mul(zi2, zi));
    inf[i] = 0;
  }
  out[i] = r;
}
int g1_to_affine_dev(zk_ctx* ctx, const G1J* a, uint32_t n, G1A* out, uint8_t* inf) {
  if (n) ZK_LAUNCH(ctx, "g1_to_affine", 241ull * n, k_g1_to_affine, g1_grid(n), dim3(G1T), 0, a, n, out, inf);
  return ZK_OK;
}

}  // namespace zk
