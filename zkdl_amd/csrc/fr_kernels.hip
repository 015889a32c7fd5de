// zkdl_amd/csrc/fr_kernels.hip -- FrTensor kernels for gfx950 (reference: fr-tensor.cu).
//
// Layout: AoS, 32 B per element, 16-B aligned, so a lane moves one element with two dwordx4
// accesses and a wave covers 2 KiB contiguous.  All of these kernels are HBM-streaming
// (algorithmic bytes per launch are passed to ZK_LAUNCH and documented in DESIGN.md).
//
// Multilinear evaluation / partial evaluation do not replay the reference's log2(n) fold launches
// (fr-tensor.cu:411-443).  Fr results are canonical, so any algebraically equal order is
// bit-identical; here a fold over c index bits is ONE pass that contracts the table with the
// 2^c-entry eq-weight table of those c challenges (built on device by k_eq_table).
#include "ctx.hpp"
#include "dev_util.hpp"

namespace zk {

// ------------------------------------------------------------------ elementwise
template <int OP>
__device__ __forceinline__ void k_fr_binary(uint32_t zk_nblk, const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += zk_nblk * blockDim.x) {
    Fr x = a[i], y = b[i];
    out[i] = OP == OP_ADD ? fr_add(x, y) : OP == OP_SUB ? fr_sub(x, y) : fr_mul(x, y);
  }
}
template <int OP>
__device__ __forceinline__ void k_fr_scalar(uint32_t zk_nblk, const Fr* __restrict__ a, Fr y, Fr* __restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += zk_nblk * blockDim.x) {
    Fr x = a[i];
    out[i] = OP == OP_ADD ? fr_add(x, y) : OP == OP_SUB ? fr_sub(x, y) : fr_mul(x, y);
  }
}
template <int OP>
__device__ __forceinline__ void k_fr_unary(uint32_t zk_nblk, const Fr* __restrict__ a, Fr* __restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += zk_nblk * blockDim.x) {
    Fr x = a[i];
    out[i] = OP == OP_NEG ? fr_neg(x) : OP == OP_MONT ? fr_mont(x) : fr_unmont(x);
  }
}

int fr_binary(zk_ctx* ctx, int op, const Fr* a, const Fr* b, Fr* out, uint32_t n) {
  if (n == 0) return ZK_OK;
  dim3 g(stream_grid(n)), bl(256);
  uint64_t bytes = 96ull * n;
  switch (op) {
    case OP_ADD: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_binary<OP_ADD>, g, 0, a, b, out, n); break;
    case OP_SUB: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_binary<OP_SUB>, g, 0, a, b, out, n); break;
    case OP_MUL: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_binary<OP_MUL>, g, 0, a, b, out, n); break;
    default: return ctx->fail(ZK_ERR_ARG, "fr_binary: bad op");
  }
  return ZK_OK;
}
int fr_scalar(zk_ctx* ctx, int op, const Fr* a, const Fr& x, Fr* out, uint32_t n) {
  if (n == 0) return ZK_OK;
  dim3 g(stream_grid(n)), bl(256);
  uint64_t bytes = 64ull * n;
  switch (op) {
    case OP_ADD: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_scalar<OP_ADD>, g, 0, a, x, out, n); break;
    case OP_SUB: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_scalar<OP_SUB>, g, 0, a, x, out, n); break;
    case OP_MUL: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_scalar<OP_MUL>, g, 0, a, x, out, n); break;
    default: return ctx->fail(ZK_ERR_ARG, "fr_scalar: bad op");
  }
  return ZK_OK;
}
int fr_unary(zk_ctx* ctx, int op, const Fr* a, Fr* out, uint32_t n) {
  if (n == 0) return ZK_OK;
  dim3 g(stream_grid(n)), bl(256);
  uint64_t bytes = 64ull * n;
  switch (op) {
    case OP_NEG: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_unary<OP_NEG>, g, 0, a, out, n); break;
    case OP_MONT: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_unary<OP_MONT>, g, 0, a, out, n); break;
    case OP_UNMONT: ZK_LAUNCHB(ctx, "fr_elementwise", bytes, 256, k_fr_unary<OP_UNMONT>, g, 0, a, out, n); break;
    default: return ctx->fail(ZK_ERR_ARG, "fr_unary: bad op");
  }
  return ZK_OK;
}

// ------------------------------------------------------------------ sum (fr-tensor.cu:240-292)
// grid-stride accumulate -> wave64 shuffle ladder -> LDS across the 4 waves -> one partial per block
__device__ __forceinline__ void k_fr_sum(uint32_t zk_nblk, const Fr* __restrict__ a, uint32_t n, Fr* __restrict__ partial) {
  __shared__ Fr lds[4];
  Fr acc = fr_zero();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += zk_nblk * blockDim.x) acc = fr_add(acc, a[i]);
  acc = block_sum_fr(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
int fr_sum_dev(zk_ctx* ctx, const Fr* a, uint32_t n, Fr* dst_d) {
  if (n == 0) {
    return dev_zero(ctx, dst_d, sizeof(Fr));
  }
  uint32_t g = stream_grid(n, 256, 1024);
  if (g == 1) {
    ZK_LAUNCHB(ctx, "fr_sum", 32ull * n, 256, k_fr_sum, dim3(1), 0, a, n, dst_d);
    return ZK_OK;
  }
  ZK_ALLOC(ctx, part, Fr, g);
  ZK_LAUNCHB(ctx, "fr_sum", 32ull * n, 256, k_fr_sum, dim3(g), 0, a, n, part.get());
  ZK_LAUNCHB(ctx, "fr_sum", 32ull * g, 256, k_fr_sum, dim3(1), 0, (const Fr*)part.get(), g, dst_d);
  return ZK_OK;
}

// ------------------------------------------------------------------ eq-weight tables and contractions
// E[j] = prod_b (bit_b(j) ? u[b] : 1 - u[b]), j < 2^c: the weights of a c-variable multilinear fold
// (LSB-first, as Fr_me_step folds bit 0 first, fr-tensor.cu:399-409).
__device__ __forceinline__ void k_eq_table(uint32_t zk_nblk, const Fr* __restrict__ u, uint32_t c, Fr* __restrict__ E) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (1u << c)) return;
  Fr one = fr_one();
  Fr acc = one;
  for (uint32_t b = 0; b < c; b++) {
    Fr ub = u[b];
    Fr f = ((j >> b) & 1u) ? ub : fr_sub(one, ub);
    acc = fr_mul(acc, f);
  }
  E[j] = acc;
}

// out[h*W + w] = sum_{m<M} E[m] * t[(h*M + m)*W + w]   (zero beyond n); one thread per output.
__device__ __forceinline__ void k_colsum(uint32_t zk_nblk, const Fr* __restrict__ t, uint32_t n, const Fr* __restrict__ E, uint32_t M, uint32_t W,
                                                 uint32_t total_out, Fr* __restrict__ out) {
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total_out; idx += zk_nblk * blockDim.x) {
    uint32_t h = idx / W, w = idx - h * W;
    Fr acc = fr_zero();
    uint64_t base = (uint64_t)h * M * W + w;
    for (uint32_t m = 0; m < M; m++) {
      uint64_t src = base + (uint64_t)m * W;
      if (src < n) acc = fr_add(acc, fr_mul(E[m], t[src]));
    }
    out[idx] = acc;
  }
}
// W == 1, M >= 64: one block per output h, lanes stride the contiguous row.
__device__ __forceinline__ void k_dot_block(uint32_t zk_nblk, const Fr* __restrict__ t, uint32_t n, const Fr* __restrict__ E, uint32_t M, Fr* __restrict__ out) {
  __shared__ Fr lds[4];
  uint32_t h = blockIdx.x;
  uint64_t base = (uint64_t)h * M;
  Fr acc = fr_zero();
  for (uint32_t m = threadIdx.x; m < M; m += blockDim.x) {
    uint64_t src = base + m;
    if (src < n) acc = fr_add(acc, fr_mul(E[m], t[src]));
  }
  acc = block_sum_fr(acc, lds);
  if (threadIdx.x == 0) out[h] = acc;
}

static int eq_table(zk_ctx* ctx, const Fr* u_d, uint32_t c, Fr* E) {
  uint32_t M = 1u << c;
  ZK_LAUNCHB(ctx, "eq_table", 32ull * M, 256, k_eq_table, dim3((M + 255) / 256), 0, u_d, c, E);
  return ZK_OK;
}

// fold the lowest c window-index bits of t (n elements, window W) into out (H*W elements)
static int contract(zk_ctx* ctx, const Fr* t, uint32_t n, const Fr* u_d, uint32_t c, uint32_t W, uint32_t H, Fr* out) {
  uint32_t M = 1u << c;
  ZK_ALLOC(ctx, E, Fr, M);
  ZK_TRY(eq_table(ctx, u_d, c, E.get()));
  uint64_t bytes = 32ull * ((uint64_t)n + (uint64_t)H * W);
  if (W == 1 && M >= 64) {
    ZK_LAUNCHB(ctx, "fr_contract", bytes, 256, k_dot_block, dim3(H), 0, t, n, (const Fr*)E.get(), M, out);
  } else {
    uint32_t total = H * W;
    ZK_LAUNCHB(ctx, "fr_contract", bytes, 256, k_colsum, dim3(stream_grid(total)), 0, t, n, (const Fr*)E.get(), M, W, total, out);
  }
  return ZK_OK;
}

// FrTensor::operator()(u) (fr-tensor.cu:295-300 guard, :411-418 semantics); result to dst_d
int fr_me_dev(zk_ctx* ctx, const Fr* t, uint32_t n, const Fr* u_d, uint32_t k, Fr* dst_d) {
  if (n == 0 || k > 31) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (n <= ((1u << k) / 2) || n > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (k == 0) {
    return dev_copy(ctx, dst_d, t, sizeof(Fr));
  }
  const Fr* cur = t;
  uint32_t cur_n = n, kk = k, off = 0;
  DBuf<Fr> tmp[2];
  int which = 0;
  while (kk > 0) {
    uint32_t c = kk < 10 ? kk : 10;
    uint32_t M = 1u << c;
    uint32_t H = (cur_n + M - 1) / M;
    Fr* out = dst_d;
    if (kk != c) {
      if (!tmp[which].alloc(ctx, H)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: me temp");
      out = tmp[which].get();
    }
    ZK_TRY(contract(ctx, cur, cur_n, u_d + off, c, 1, H, out));
    cur = out;
    cur_n = H;
    off += c;
    kk -= c;
    which ^= 1;
  }
  return ZK_OK;
}

uint32_t fr_partial_me_size(uint32_t n, uint32_t k, uint32_t window) {
  uint64_t sz = n;
  for (uint32_t r = 0; r < k; r++) sz = (uint64_t)window * ((sz + 2ull * window - 1) / (2ull * window));
  return (uint32_t)sz;
}

// FrTensor::partial_me (fr-tensor.cu:370-374 guard, :434-443 semantics); k == 0 is the identity
// (UB in the reference)
int fr_partial_me_dev(zk_ctx* ctx, const Fr* t, uint32_t n, const Fr* u_d, uint32_t k, uint32_t W, Fr* out_d) {
  if (W == 0 || k > 31) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (k == 0) {
    return dev_copy(ctx, out_d, t, (size_t)n * sizeof(Fr));
  }
  if ((uint64_t)n <= (uint64_t)W * (1ull << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const Fr* cur = t;
  uint32_t cur_n = n, kk = k, off = 0;
  uint32_t nw = (uint32_t)(((uint64_t)n + W - 1) / W);
  DBuf<Fr> tmp[2];
  int which = 0;
  while (kk > 0) {
    uint32_t cmax = (W == 1) ? 10 : 5;
    uint32_t c = kk < cmax ? kk : cmax;
    uint32_t M = 1u << c;
    uint32_t H = (nw + M - 1) / M;
    Fr* out = out_d;
    if (kk != c) {
      if (!tmp[which].alloc(ctx, (size_t)H * W)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: partial_me temp");
      out = tmp[which].get();
    }
    ZK_TRY(contract(ctx, cur, cur_n, u_d + off, c, W, H, out));
    cur = out;
    cur_n = H * W;
    nw = H;
    off += c;
    kk -= c;
    which ^= 1;
  }
  return ZK_OK;
}

// Fr_me_step with the challenge read from device memory (fr-tensor.cu:399-409)
__device__ __forceinline__ void k_fr_fold(uint32_t zk_nblk, const Fr* __restrict__ in, Fr* __restrict__ out, const Fr* __restrict__ xp, uint32_t in_size, uint32_t out_size) {
  Fr x = *xp;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < out_size; g += zk_nblk * blockDim.x) {
    uint32_t g0 = 2 * g, g1 = 2 * g + 1;
    Fr r;
    if (g1 < in_size) {
      Fr a0 = in[g0], a1 = in[g1];
      r = fr_add(a0, fr_mul(x, fr_sub(a1, a0)));
    } else if (g0 < in_size) {
      Fr a0 = in[g0];
      r = fr_sub(a0, fr_mul(x, a0));
    } else {
      r = fr_zero();
    }
    out[g] = r;
  }
}
int fr_fold_dev(zk_ctx* ctx, const Fr* in, Fr* out, const Fr* x_d, uint32_t in_size, uint32_t out_size) {
  if (out_size == 0) return ZK_OK;
  ZK_LAUNCHB(ctx, "fr_fold", 32ull * ((uint64_t)in_size + out_size), 256, k_fr_fold, dim3(stream_grid(out_size)), 0, in, out, x_d, in_size, out_size);
  return ZK_OK;
}

// ------------------------------------------------------------------ split (fr-tensor.cu:376-397)
__device__ __forceinline__ void k_fr_split(uint32_t zk_nblk, const Fr* __restrict__ in, Fr* __restrict__ o0, Fr* __restrict__ o1, uint32_t in_size, uint32_t out_size, uint32_t w) {
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < out_size; g += zk_nblk * blockDim.x) {
    uint32_t wid = g / w, idx = g - wid * w;
    uint64_t g0 = 2ull * wid * w + idx, g1 = (2ull * wid + 1) * w + idx;
    o0[g] = g0 < in_size ? in[g0] : fr_zero();
    o1[g] = g1 < in_size ? in[g1] : fr_zero();
  }
}
int fr_split_dev(zk_ctx* ctx, const Fr* t, uint32_t n, uint32_t window, Fr* o0, Fr* o1) {
  if (window < 1 || window >= n) return ctx->fail(ZK_ERR_SHAPE, "Invalid window size.");
  uint32_t out_size = (n + 1) / 2;
  ZK_LAUNCHB(ctx, "fr_split", 64ull * out_size + 32ull * n, 256, k_fr_split, dim3(stream_grid(out_size)), 0, t, o0, o1, n, out_size, window);
  return ZK_OK;
}

// ------------------------------------------------------------------ random (fr-tensor.cu:302-368)
__device__ __forceinline__ void k_fr_random(uint32_t zk_nblk, Fr* out, uint32_t n, uint64_t seed, uint64_t draw) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += zk_nblk * blockDim.x) out[i] = rng_fr(seed, draw, i);
}
// uniform num_bits-bit integer minus 2^(num_bits-1), as a field element (fr-tensor.cu:302-314)
__device__ __forceinline__ void k_fr_random_int(uint32_t zk_nblk, Fr* out, uint32_t n, uint32_t num_bits, uint64_t seed, uint64_t draw) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += zk_nblk * blockDim.x) {
    Fr r = rng_fr(seed, draw, i);
    Fr v = fr_zero(), h = fr_zero();
    v.v[0] = num_bits >= 32 ? r.v[0] : (r.v[0] & ((1u << num_bits) - 1u));
    h.v[0] = 1u << (num_bits - 1);
    out[i] = fr_sub(v, h);
  }
}
int fr_random_dev(zk_ctx* ctx, Fr* out, uint32_t n) {
  uint64_t draw = ctx->rng_draw++;
  if (n) ZK_LAUNCHB(ctx, "fr_random", 32ull * n, 256, k_fr_random, dim3(stream_grid(n)), 0, out, n, ctx->rng_seed, draw);
  return ZK_OK;
}
int fr_random_int_dev(zk_ctx* ctx, Fr* out, uint32_t n, uint32_t num_bits) {
  if (num_bits == 0 || num_bits > 32) return ctx->fail(ZK_ERR_ARG, "random_int: num_bits must be in 1..32");
  uint64_t draw = ctx->rng_draw++;
  if (n) ZK_LAUNCHB(ctx, "fr_random", 32ull * n, 256, k_fr_random_int, dim3(stream_grid(n)), 0, out, n, num_bits, ctx->rng_seed, draw);
  return ZK_OK;
}

}  // namespace zk
