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
#include "fs29.hpp"

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
// A term is NOT a Montgomery product plus a modular addition (~360 instructions in the 32-bit-limb form): both operands go to nine 29-bit limbs and their
// 81 partial products land in 64-bit columns that live across the whole loop (fs29.hpp: fs_mul_raw_acc64; carries every sixth term), and the sum is
// reduced ONCE per output (fs_cols_to_fr: T / 2^256 mod r, canonical -- the value sum_m fr_mul(E[m], t[..]) has, so the transcript is bit-identical).
// ~120 instructions per term; the row loop keeps six 32-byte loads in flight.  E[m] is uniform over the wave: scalar loads, unpacked on the scalar unit.
__device__ __forceinline__ void k_colsum(uint32_t zk_nblk, const Fr* __restrict__ t, uint32_t n, const Fr* __restrict__ E, uint32_t M, uint32_t W,
                                                 uint32_t total_out, Fr* __restrict__ out) {
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total_out; idx += zk_nblk * blockDim.x) {
    const uint32_t h = idx / W, w = idx - h * W;
    const uint64_t base = (uint64_t)h * M * W + w;
    uint64_t acc[18];
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0;
    uint32_t m = 0;
    for (; m + 6 <= M && base + (uint64_t)(m + 5) * W < n; m += 6) {
      Fr v[6];
#pragma unroll
      for (int q = 0; q < 6; q++) v[q] = t[base + (uint64_t)(m + q) * W];
#pragma unroll
      for (int q = 0; q < 6; q++) fs_mul_raw_acc64(acc, fs_unpack(E[m + q]), fs_unpack(v[q]));
      fs_cols_norm64(acc);
    }
    for (; m < M; m++) {  // at most five terms in range (rows beyond n are zero)
      const uint64_t src = base + (uint64_t)m * W;
      if (src < n) fs_mul_raw_acc64(acc, fs_unpack(E[m]), fs_unpack(t[src]));
    }
    fs_cols_norm64(acc);
    out[idx] = fs_cols_to_fr(acc);
  }
}
// W == 1, M >= 64: one WAVE per output h (four per workgroup); lanes stride the contiguous row, four loads in flight; each lane keeps its own 64-bit
// columns, the wave adds them as plain integers by shuffles, and lane 0 reduces once (no LDS stage, no barriers).
__device__ __forceinline__ void k_dot_block(uint32_t zk_nblk, const Fr* __restrict__ t, uint32_t n, const Fr* __restrict__ E, uint32_t M, Fr* __restrict__ out,
                                                    uint32_t H) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (h >= H) return;
  const uint64_t base = (uint64_t)h * M;
  uint64_t acc[18];
#pragma unroll
  for (int k = 0; k < 18; k++) acc[k] = 0;
  uint32_t m = lane;
  for (; m + 192 < M && base + m + 192 < n; m += 256) {
    Fr v[4], e[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      v[q] = t[base + m + 64 * q];
      e[q] = E[m + 64 * q];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) fs_mul_raw_acc64(acc, fs_unpack(e[q]), fs_unpack(v[q]));
    fs_cols_norm64(acc);
  }
  for (; m < M; m += 64) {  // at most three terms per lane
    const uint64_t src = base + m;
    if (src < n) fs_mul_raw_acc64(acc, fs_unpack(E[m]), fs_unpack(t[src]));
  }
  fs_cols_norm64(acc);  // columns < 2^29 (the top one: the lane's whole sum / 2^493): 64 lanes add without overflow
#pragma unroll
  for (int k = 0; k < 18; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off, 64);
  }
  if (lane == 0) out[h] = fs_cols_to_fr(acc);
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
  if (W == 1 && M >= 128) {
    ZK_LAUNCHB(ctx, "fr_contract", bytes, 256, k_dot_block, dim3((H + 3) / 4), 0, t, n, (const Fr*)E.get(), M, out, H);
  } else {
    uint32_t total = H * W;
    ZK_LAUNCHB(ctx, "fr_contract", bytes, 256, k_colsum, dim3(stream_grid(total)), 0, t, n, (const Fr*)E.get(), M, W, total, out);
  }
  return ZK_OK;
}

// ------------------------------------------------------------------ contractions of a QUANTISED tensor
// A layer's weights are small signed integers (|w| < 2^15 after zkfc.cu:80-88's scaling); the prover reads
// them twice per proof (W.partial_me(u_out, 1) and the row fold of Commitment::open), and a full Montgomery
// product per 32-byte element caps that at ~3.3 TB/s of issue rate.  With the weights also kept as int16,
// E[m] * w is an 8-limb x 1-limb product accumulated UNREDUCED in 9 limbs (sum of <= 2^11 terms < 2^283), and
// only each output is reduced mod r: ~40 instructions and 2 bytes per element instead of ~330 and 32.
// The value is the same field element: sum_m E[m] * w_m mod r with E in Montgomery form, i.e. exactly what
// Fr_partial_me computes on the Montgomery-form weights.
struct Acc9 {
  uint32_t v[9];
};
__device__ __forceinline__ Acc9 acc9_zero() {
  Acc9 a;
#pragma unroll
  for (int i = 0; i < 9; i++) a.v[i] = 0;
  return a;
}
// T = lo + hi * 2^256  ->  T mod r:  lo < 2^256 < 5r by repeated subtraction, hi * 2^256 = mont(hi)
__device__ __forceinline__ Fr acc9_reduce(const Acc9& a) {
  Fr lo;
#pragma unroll
  for (int i = 0; i < 8; i++) lo.v[i] = a.v[i];
#pragma unroll
  for (int it = 0; it < 4; it++) lo = f_reduce_once<FrP>(lo);
  Fr hi = fr_zero();
  hi.v[0] = a.v[8];
  return fr_add(lo, fr_mont(hi));
}
// ---- the int16 contractions, version 2: no per-element sign handling, no carries, and the weight table kept in registers.
//   * signs: with u = w + 2^15 (an unsigned 16-bit number),  sum_m E[m] w_m = sum_m E[m] u_m - 2^15 T,  T = sum_m E[m] -- one integer per
//     launch shape (the same for every output).  The correction enters as the non-negative constant  C = 2^15 (M r - T)  (a multiple of r
//     more), so an output is  (sum_m E[m] u_m + C) mod r.
//   * carries: E[m]_i * u < 2^48, so a limb's products are summed as they are in a 64-bit COLUMN (one v_mad_u64_u32 each; <= 2^16 terms);
//     columns add across lanes as plain integers and are carried once per output.
//   * k_dot_q16: a wave takes DOT_ROWS consecutive rows and every lane two adjacent columns per step, whose table entries stay in registers
//     for all the rows: one dword of weights and 16 MACs per row and step, the 32-byte table entries read once per DOT_ROWS rows.
// ~10 instructions per weight instead of ~50 (conditional negation of the table entry + a carry chain per limb).
struct Col8 {
  uint64_t c[8];
};
__device__ __forceinline__ void col8_zero(Col8& a) {
#pragma unroll
  for (int i = 0; i < 8; i++) a.c[i] = 0;
}
__device__ __forceinline__ void col8_mac(Col8& a, const Fr& e, uint32_t u) {
#pragma unroll
  for (int i = 0; i < 8; i++) a.c[i] += (uint64_t)e.v[i] * u;
}
// all lanes receive the wave's column sums
__device__ __forceinline__ void col8_wave_sum(Col8& a) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t lo = __shfl_xor((uint32_t)a.c[i], off, 64), hi = __shfl_xor((uint32_t)(a.c[i] >> 32), off, 64);
      a.c[i] += ((uint64_t)hi << 32) | lo;
    }
  }
}
// columns (sum < 2^288) -> the 9-limb integer
__device__ __forceinline__ Acc9 col8_carry(const Col8& a) {
  Acc9 r;
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t lo = (a.c[i] & 0xffffffffull) + (carry & 0xffffffffull);
    r.v[i] = (uint32_t)lo;
    carry = (a.c[i] >> 32) + (carry >> 32) + (lo >> 32);
  }
  r.v[8] = (uint32_t)carry;
  return r;
}
__device__ __forceinline__ Acc9 acc9_add(const Acc9& a, const Acc9& b) {
  Acc9 r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t co;
    r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co);
    c = co;
  }
  return r;
}
// C = 2^15 (M r - T), T = the integer sum of the M table entries (< M r), M = 2^c
__device__ __forceinline__ Acc9 q16_offset_const(const Acc9& T, uint32_t c) {
  Acc9 mr;  // r << c
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t lo = i < 8 ? FrP::P(i) : 0u, below = i ? FrP::P(i - 1) : 0u;
    mr.v[i] = c ? ((lo << c) | (below >> (32 - c))) : lo;
  }
  Acc9 d;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t bo;
    d.v[i] = __builtin_subc(mr.v[i], T.v[i], bw, &bo);
    bw = bo;
  }
  Acc9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (d.v[i] << 15) | (i ? d.v[i - 1] >> 17 : 0u);
  return r;
}
// the offset constant of one launch shape (one wave): C = 2^15 (M r - sum_m E[m])
__device__ __forceinline__ void k_q16_offset(uint32_t zk_nblk, const Fr* __restrict__ E, uint32_t M, uint32_t c, Acc9* __restrict__ out) {
  if (blockIdx.x) return;
  Col8 t;
  col8_zero(t);
  for (uint32_t m = threadIdx.x; m < M; m += 64) col8_mac(t, E[m], 1u);
  col8_wave_sum(t);
  if (threadIdx.x == 0) *out = q16_offset_const(col8_carry(t), c);
}
// out[h*W + w] = sum_{m<M} E[m] * q[(h*M + m)*W + w]; one thread per output; E[m] is uniform across the wave
__device__ __forceinline__ void k_colsum_q16(uint32_t zk_nblk, const int16_t* __restrict__ q, uint32_t n, const Fr* __restrict__ E, uint32_t M, const Acc9* __restrict__ Cp, uint32_t W,
                                                     uint32_t total_out, Fr* __restrict__ out) {
  const Acc9 C = *Cp;
  // E[m] is the same for every lane: read it through the constant address space, i.e. as SCALAR loads (the table was written by an earlier launch) --
  // as 32 bytes per lane and weight from L1 the loop ran at the L1 read bandwidth (2 KB per wave and step), 3.5x below its MAC rate
  const __attribute__((address_space(4))) uint32_t* Ec = (const __attribute__((address_space(4))) uint32_t*)E;
  const __attribute__((address_space(1))) int16_t* qg = (const __attribute__((address_space(1))) int16_t*)q;
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total_out; idx += zk_nblk * blockDim.x) {
    const uint32_t h = idx / W, w = idx - h * W;
    Col8 acc;
    col8_zero(acc);
    const uint64_t base = (uint64_t)h * M * W + w;
    for (uint32_t m = 0; m < M; m++) {
      const uint64_t src = base + (uint64_t)m * W;
      const uint32_t u = src < n ? (uint32_t)((int32_t)qg[src] + 32768) : 32768u;   // a missing element is the weight 0
      Fr e;
#pragma unroll
      for (int l = 0; l < 8; l++) e.v[l] = Ec[8u * m + l];
      col8_mac(acc, e, u);
    }
    out[idx] = acc9_reduce(acc9_add(col8_carry(acc), C));
  }
}
// W == 1: out[h] = sum_{m<M} E[m] q[h*M + m], M = 2^c a multiple of 128.  One wave per DOT_ROWS consecutive rows: a lane's two table entries of a
// step (64 bytes from L1) serve 2 x DOT_ROWS weights -- at four rows the kernel ran at the L1 read bandwidth, not at the MAC rate
static constexpr uint32_t DOT_ROWS = 8;
__device__ __forceinline__ void k_dot_q16(uint32_t zk_nblk, const int16_t* __restrict__ q, uint32_t n, const Fr* __restrict__ E, uint32_t M, const Acc9* __restrict__ Cp,
                                                  Fr* __restrict__ out, uint32_t H) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t h0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * DOT_ROWS;
  if (h0 >= H) return;
  Col8 acc[DOT_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < DOT_ROWS; r++) col8_zero(acc[r]);
  for (uint32_t m = 2 * lane; m < M; m += 128) {
    const Fr e0 = E[m], e1 = E[m + 1];
#pragma unroll
    for (uint32_t r = 0; r < DOT_ROWS; r++) {
      const uint64_t src = (uint64_t)(h0 + r) * M + m;   // even: the two int16 are one aligned dword
      uint32_t pair = 0u;                                 // two zero weights
      if (h0 + r < H) {
        if (src + 1 < n) pair = *reinterpret_cast<const uint32_t*>(q + src);
        else if (src < n) pair = (uint32_t)(uint16_t)q[src];
      }
      const uint32_t u0 = ((pair & 0xffffu) ^ 0x8000u), u1 = ((pair >> 16) ^ 0x8000u);   // w + 2^15 for a two's complement int16
      col8_mac(acc[r], e0, u0);
      col8_mac(acc[r], e1, u1);
    }
  }
  const Acc9 C = *Cp;
#pragma unroll
  for (uint32_t r = 0; r < DOT_ROWS; r++) {
    col8_wave_sum(acc[r]);
    if (lane == r && h0 + r < H) out[h0 + r] = acc9_reduce(acc9_add(col8_carry(acc[r]), C));
  }
}

// w (Montgomery) -> int16 if |w| < 2^15, else *overflow is set
__device__ __forceinline__ void k_quant16(uint32_t zk_nblk, const Fr* __restrict__ t, uint32_t n, int16_t* __restrict__ q, uint32_t* __restrict__ overflow) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += zk_nblk * blockDim.x) {
    Fr v = fr_unmont(t[i]);
    Fr nv = fr_neg(v);
    uint32_t hi_v = 0, hi_n = 0;
#pragma unroll
    for (int l = 1; l < 8; l++) {
      hi_v |= v.v[l];
      hi_n |= nv.v[l];
    }
    int32_t r = 0;
    if (hi_v == 0 && v.v[0] < 32768u) r = (int32_t)v.v[0];
    else if (hi_n == 0 && nv.v[0] <= 32768u) r = -(int32_t)nv.v[0];
    else atomicOr(overflow, 1u);
    q[i] = (int16_t)r;
  }
}
int fr_quant16_dev(zk_ctx* ctx, const Fr* t, uint32_t n, int16_t* q, uint32_t* overflow_d) {
  ZK_LAUNCHB(ctx, "fr_quant16", 34ull * n, 256, k_quant16, dim3(stream_grid(n)), 0, t, n, q, overflow_d);
  return ZK_OK;
}

// out[w] = sum_{m<M} x[m] * q[m*W + w]: a row vector of ARBITRARY field elements times the int16 copy of a quantised [M][W] matrix (M = 2^c <= 2^14:
// a column of <= 2^14 products < 2^48 fits 64 bits).  The batch-1 forward pass of a layer on a folded activation vector (zkdl_amd/shard.py: the
// owner's stand-in Z = a W) -- through the Montgomery matmul that was 4 M field products on a grid shaped for batches, ~1.5 ms per layer.
// Two launches: (w, slice of VQ_SLICE rows) -> 64-bit columns (x[m] uniform per wave: scalar loads), then per w the slices' sum, the offset, the reduction.
static constexpr uint32_t VQ_SLICE = 64;
__device__ __forceinline__ void k_vec_q16_part(uint32_t zk_nblk, const int16_t* __restrict__ q, const Fr* __restrict__ x, uint32_t M, uint32_t W, Col8* __restrict__ part) {
  const uint32_t S = (M + VQ_SLICE - 1) / VQ_SLICE;
  const __attribute__((address_space(4))) uint32_t* xc = (const __attribute__((address_space(4))) uint32_t*)x;
  const __attribute__((address_space(1))) int16_t* qg = (const __attribute__((address_space(1))) int16_t*)q;
  const uint32_t wblocks = (W + blockDim.x - 1) / blockDim.x;
  for (uint32_t blk = blockIdx.x; blk < wblocks * S; blk += zk_nblk) {
    const uint32_t sl = blk / wblocks, w = (blk - sl * wblocks) * blockDim.x + threadIdx.x;
    if (w >= W) continue;
    Col8 acc;
    col8_zero(acc);
    const uint32_t m1 = min(M, (sl + 1) * VQ_SLICE);
    for (uint32_t m = sl * VQ_SLICE; m < m1; m++) {
      const uint32_t u = (uint32_t)((int32_t)qg[(uint64_t)m * W + w] + 32768);
      Fr e;
#pragma unroll
      for (int l = 0; l < 8; l++) e.v[l] = xc[8u * m + l];
      col8_mac(acc, e, u);
    }
    part[(uint64_t)sl * W + w] = acc;
  }
}
__device__ __forceinline__ void k_vec_q16_fin(uint32_t zk_nblk, const Col8* __restrict__ part, uint32_t S, uint32_t W, const Acc9* __restrict__ Cp, Fr* __restrict__ out) {
  const Acc9 C = *Cp;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < W; w += zk_nblk * blockDim.x) {
    Col8 acc;
    col8_zero(acc);
    for (uint32_t sl = 0; sl < S; sl++) {
      const Col8 p = part[(uint64_t)sl * W + w];
#pragma unroll
      for (int i = 0; i < 8; i++) acc.c[i] += p.c[i];
    }
    out[w] = acc9_reduce(acc9_add(col8_carry(acc), C));
  }
}
int fr_vec_q16_dev(zk_ctx* ctx, const Fr* x, uint32_t M, const int16_t* q, uint32_t W, Fr* out) {
  const uint32_t c = ceil_log2(M);
  if (M == 0 || W == 0 || M != (1u << c) || c > 14) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t S = (M + VQ_SLICE - 1) / VQ_SLICE;
  ZK_ALLOC(ctx, C, Acc9, 1);
  ZK_ALLOC(ctx, part, Col8, (size_t)S * W);
  ZK_LAUNCHB(ctx, "eq_table", 32ull * M, 64, k_q16_offset, dim3(1), 0, x, M, c, C.get());
  ZK_LAUNCHB(ctx, "fr_contract_q16", 2ull * M * W + 32ull * M + 64ull * S * W, 64, k_vec_q16_part, dim3(stream_grid((uint64_t)S * W, 64, 4096)), 0, q, x, M, W, part.get());
  ZK_LAUNCHB(ctx, "fr_contract_q16", 64ull * S * W + 32ull * W, 256, k_vec_q16_fin, dim3(stream_grid(W)), 0, (const Col8*)part.get(), S, W, (const Acc9*)C.get(), out);
  return ZK_OK;
}

// FrTensor::partial_me over the int16 copy of a quantised tensor: the first level (all the traffic) reads the
// int16s, the remaining levels fold its Fr output exactly as fr_partial_me_dev does
int fr_partial_me_q16_dev(zk_ctx* ctx, const int16_t* q, uint32_t n, const Fr* u_d, uint32_t k, uint32_t W, Fr* out_d) {
  if (W == 0 || k == 0 || k > 31) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if ((uint64_t)n <= (uint64_t)W * (1ull << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t nw = (uint32_t)(((uint64_t)n + W - 1) / W);
  const uint32_t cmax = (W == 1) ? 10 : 6;
  const uint32_t c = k < cmax ? k : cmax;
  const uint32_t M = 1u << c;
  const uint32_t H = (nw + M - 1) / M;
  DBuf<Fr> tmp;
  Fr* out = out_d;
  if (k != c) {
    if (!tmp.alloc(ctx, (size_t)H * W)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: partial_me temp");
    out = tmp.get();
  }
  ZK_ALLOC(ctx, E, Fr, M);
  ZK_TRY(eq_table(ctx, u_d, c, E.get()));
  ZK_ALLOC(ctx, C, Acc9, 1);
  ZK_LAUNCHB(ctx, "eq_table", 32ull * M, 64, k_q16_offset, dim3(1), 0, (const Fr*)E.get(), M, c, C.get());
  const uint64_t bytes = 2ull * n + 32ull * ((uint64_t)H * W + M);
  if (W == 1 && M >= 128) {
    ZK_LAUNCHB(ctx, "fr_contract_q16", bytes, 256, k_dot_q16, dim3((H + 4 * DOT_ROWS - 1) / (4 * DOT_ROWS)), 0, q, n, (const Fr*)E.get(), M, (const Acc9*)C.get(), out, H);
  } else {
    uint32_t total = H * W;
    ZK_LAUNCHB(ctx, "fr_contract_q16", bytes, 256, k_colsum_q16, dim3(stream_grid(total)), 0, q, n, (const Fr*)E.get(), M, (const Acc9*)C.get(), W, total, out);
  }
  if (k == c) return ZK_OK;
  return fr_partial_me_dev(ctx, out, H * W, u_d + c, k - c, W, out_d);
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
