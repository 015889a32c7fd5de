// zkdl_amd/csrc/msm.hip -- bucket-method (Pippenger) multi-scalar multiplication over SHARED bases
// for gfx950, and the commitment/opening paths built on it (ZK_MSM_PIPPENGER back end).
//
// What the reference does: one 256-step double-and-add per (generator, scalar) pair, then sums
// (g1-tensor.cu:422-445, commitment.cu:3-81) -- m*n independent 19-us-latency chains for commit and
// five chained scalar-muls per thread per round for me_open.  What this file does instead:
//
//  * Window tables.  Every MSM on the path runs over a FIXED generator set (a layer's generators,
//    or a layer's commitment points), so 2^(c*w) * G_j is precomputed once per set, in affine form
//    (zk_gens, built lazily).  All windows of an MSM then share ONE bucket set and there is no
//    Horner doubling chain on the critical path.
//  * Signed digits.  A scalar s < r is first folded to (sign, min(s, r-s)) so the quantised weights
//    (|w| < 2^13, half of them "negative" = r - |w|) need one or two windows, then recoded to digits
//    in [-2^(c-1), 2^(c-1)) by the add-a-constant trick (no inter-window carries): 2^(c-1) buckets.
//  * Per (MSM, window) workgroup: digit histogram + exclusive scan + scatter in LDS -> items grouped by
//    bucket (k_msm_sort).  Per (MSM, bucket) lane: mixed additions of its items (k_msm_accum).
//  * Bucket reduction without a running-sum chain: sum_b b*S_b = sum_i 2^i * (sum of buckets whose
//    index has bit i), i.e. c independent tree sums (k_msm_bits, LDS tree) + a c-step Horner
//    (k_msm_horner).
//  * Flattened opening.  Every value Commitment::me_open pushes (T, T0, T1 per round and the final
//    folded generator) is a linear combination of the ORIGINAL generators with scalars computable in
//    Fr alone (derivation in DESIGN.md "Flattened opening"), so all 3k+1 of them are ONE batched MSM
//    instead of k dependent rounds of scalar-muls.
//
// Same group elements as the reference, different Jacobian representatives: results are emitted in
// canonical form (affine x, y with Z = ONE; infinity as {0, ONE, 0}) and compared affine-canonically.
// All of this is integer-VALU bound (v_mad_u64_u32), not HBM bound; bases stay L2/MALL resident.
#include "layers.hpp"

#include "dev_util.hpp"

namespace zk {

// ------------------------------------------------------------------ window tables
// jac[w*n + j] = 2^(c*w) * gens[j]
__global__ void __launch_bounds__(64) k_table_chain(const G1J* __restrict__ gens, uint32_t n, uint32_t c, uint32_t W, G1J* __restrict__ jac) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  G1J P = gens[j];
  for (uint32_t w = 0; w < W; w++) {
    jac[(uint64_t)w * n + j] = P;
    if (w + 1 < W)
      for (uint32_t i = 0; i < c; i++) P = g1_double(P);
  }
}

int gens_table(zk_ctx* ctx, zk_gens* g, uint32_t c, uint32_t W, const G1Table** out) {
  auto it = g->tables.find(c);
  if (it != g->tables.end() && it->second.W >= W) {
    *out = &it->second;
    return ZK_OK;
  }
  if (it != g->tables.end()) {
    (void)hipFree(it->second.pts);
    (void)hipFree(it->second.inf);
    g->tables.erase(it);
  }
  G1Table T;
  T.n = g->n;
  T.c = c;
  T.W = W;
  size_t total = (size_t)W * g->n;
  if (hipMalloc((void**)&T.pts, total * sizeof(G1A)) != hipSuccess || hipMalloc((void**)&T.inf, total) != hipSuccess)
    return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: MSM window table");
  {
    ZK_ALLOC(ctx, jac, G1J, total);
    ZK_LAUNCH(ctx, "msm_table_build", 144ull * total, k_table_chain, dim3((g->n + 63) / 64), dim3(64), 0, (const G1J*)g->pts, g->n, c, W, jac.get());
    ZK_TRY(g1_to_affine_dev(ctx, jac.get(), (uint32_t)total, T.pts, T.inf));
  }
  g->tables[c] = T;
  *out = &g->tables[c];
  return ZK_OK;
}

// ------------------------------------------------------------------ scalar preparation
// mag = min(s, r - s) with the sign in bit 255 (mag < 2^254); *maxbits = max bit length of mag.
__global__ void __launch_bounds__(256) k_scalar_prep(const Fr* __restrict__ s, Fr* __restrict__ out, uint64_t total, int do_unmont, uint32_t* __restrict__ maxbits) {
  uint32_t mx = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    Fr v = s[i];
    if (do_unmont) v = fr_unmont(v);
    Fr nv = fr_neg(v);  // r - v (0 for v == 0)
    bool neg = !l_gte(nv, v);  // r - v < v
    Fr m = neg ? nv : v;
    uint32_t len = 0;
#pragma unroll
    for (int l = 0; l < 8; l++)
      if (m.v[l]) len = 32u * l + (32u - __clz(m.v[l]));
    mx = len > mx ? len : mx;
    if (neg) m.v[7] |= 0x80000000u;
    out[i] = m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if ((threadIdx.x & 63u) == 0 && mx) atomicMax(maxbits, mx);
}

// K = sum_{w < Wk} 2^(c*w + c - 1): adding it makes every c-bit window of mag + K equal to
// (signed digit + 2^(c-1)), with no carry dependence between windows.
struct RecodeK {
  uint32_t v[9];
};
static RecodeK make_recode_k(uint32_t c, uint32_t W) {
  RecodeK k;
  for (int i = 0; i < 9; i++) k.v[i] = 0;
  for (uint32_t w = 0; w < W; w++) {
    uint32_t bit = c * w + c - 1;
    if (bit < 288) k.v[bit >> 5] |= 1u << (bit & 31);
  }
  return k;
}
// signed digit of window w and the point sign; returns bucket index |d| (0 = skip)
__device__ __forceinline__ uint32_t recode_digit(const Fr& mag, const RecodeK& K, uint32_t c, uint32_t w, bool& negative) {
  uint32_t t[10];
  uint32_t carry = 0;
  bool sneg = (mag.v[7] >> 31) != 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t m = i == 7 ? (mag.v[7] & 0x7fffffffu) : mag.v[i];
    uint32_t co;
    t[i] = __builtin_addc(m, K.v[i], carry, &co);
    carry = co;
  }
  t[8] = K.v[8] + carry;
  t[9] = 0;
  uint32_t bit = c * w, li = bit >> 5, sh = bit & 31;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    lo = (li == (uint32_t)i) ? t[i] : lo;
    hi = (li == (uint32_t)i) ? t[i + 1] : hi;
  }
  uint64_t both = ((uint64_t)hi << 32) | lo;
  uint32_t dp = (uint32_t)(both >> sh) & ((1u << c) - 1u);
  int32_t d = (int32_t)dp - (int32_t)(1u << (c - 1));
  negative = (d < 0) != sneg;
  return (uint32_t)(d < 0 ? -d : d);
}

// ------------------------------------------------------------------ sort items of one (MSM, window) by bucket
// grid (W, q); LDS: hist[B+1] + 256 scan slots.  idx[(q*W + w)*n + pos] = j | sign<<31, grouped by
// bucket; off[(q*W + w)*(B+2) + b] = first position of bucket b (b = 0..B+1).
__global__ void __launch_bounds__(256) k_msm_sort(const Fr* __restrict__ mags, uint32_t n, uint32_t W, uint32_t c, RecodeK K, const uint8_t* __restrict__ base_inf,
                                                   uint32_t* __restrict__ idx, uint32_t* __restrict__ off) {
  extern __shared__ uint32_t lds[];
  const uint32_t B = 1u << (c - 1);
  uint32_t* hist = lds;            // B + 2
  uint32_t* part = lds + (B + 2);  // 256
  const uint32_t w = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
  const Fr* row = mags + (uint64_t)q * n;
  for (uint32_t b = tid; b < B + 2; b += 256) hist[b] = 0;
  __syncthreads();
  for (uint32_t j = tid; j < n; j += 256) {
    bool neg;
    uint32_t b = base_inf[j] ? 0u : recode_digit(row[j], K, c, w, neg);
    atomicAdd(&hist[b], 1u);
  }
  __syncthreads();
  // exclusive scan of hist[0..B+1)
  const uint32_t per = (B + 2 + 255) / 256;
  uint32_t s = 0;
  for (uint32_t i = 0; i < per; i++) {
    uint32_t b = tid * per + i;
    if (b < B + 2) s += hist[b];
  }
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0;
  uint32_t* o = off + ((uint64_t)q * W + w) * (B + 2);
  for (uint32_t i = 0; i < per; i++) {
    uint32_t b = tid * per + i;
    if (b < B + 2) {
      uint32_t h = hist[b];
      hist[b] = run;  // becomes the scatter cursor
      o[b] = run;
      run += h;
    }
  }
  __syncthreads();
  uint32_t* out = idx + ((uint64_t)q * W + w) * n;
  for (uint32_t j = tid; j < n; j += 256) {
    bool neg;
    uint32_t b = base_inf[j] ? 0u : recode_digit(row[j], K, c, w, neg);
    uint32_t pos = atomicAdd(&hist[b], 1u);
    out[pos] = j | (neg ? 0x80000000u : 0u);
  }
}

// one lane per (MSM q, bucket b in 1..B): mixed additions of every item of that bucket in all windows
__global__ void __launch_bounds__(64) k_msm_accum(const G1A* __restrict__ table, uint32_t n, uint32_t W, uint32_t c, const uint32_t* __restrict__ idx,
                                                   const uint32_t* __restrict__ off, G1J* __restrict__ buckets) {
  const uint32_t B = 1u << (c - 1);
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const uint32_t q = blockIdx.y;
  if (b > B) return;
  G1J acc = g1_zero();
  for (uint32_t w = 0; w < W; w++) {
    const uint32_t* o = off + ((uint64_t)q * W + w) * (B + 2);
    const uint32_t* items = idx + ((uint64_t)q * W + w) * n;
    const G1A* tw = table + (uint64_t)w * n;
    uint32_t p0 = o[b], p1 = o[b + 1];
    for (uint32_t p = p0; p < p1; p++) {
      uint32_t it = items[p];
      G1A P = tw[it & 0x7fffffffu];
      if (it >> 31) P.y = fp_neg(P.y);
      acc = g1_add_mixed(acc, P);
    }
  }
  buckets[(uint64_t)q * B + (b - 1)] = acc;
}

// U[q*c + i] = sum of buckets b (1..B) whose index has bit i set; grid (c, q), LDS tree
template <int T>
__global__ void __launch_bounds__(T) k_msm_bits(const G1J* __restrict__ buckets, uint32_t c, G1J* __restrict__ U) {
  __shared__ G1J s[T];
  const uint32_t B = 1u << (c - 1);
  const uint32_t i = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
  const G1J* bk = buckets + (uint64_t)q * B;
  G1J acc = g1_zero();
  for (uint32_t b = tid + 1; b <= B; b += T)
    if ((b >> i) & 1u) acc = g1_add(acc, bk[b - 1]);
  s[tid] = acc;
  __syncthreads();
  for (uint32_t st = T / 2; st > 0; st >>= 1) {
    if (tid < st) s[tid] = g1_add(s[tid], s[tid + st]);
    __syncthreads();
  }
  if (tid == 0) U[(uint64_t)q * c + i] = s[0];
}

// out[q] = sum_i 2^i U[q][i], emitted in canonical form (affine x, y, Z = ONE; infinity {0, ONE, 0})
__global__ void __launch_bounds__(64) k_msm_horner(const G1J* __restrict__ U, uint32_t c, uint32_t q_total, G1J* __restrict__ out) {
  uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= q_total) return;
  const G1J* u = U + (uint64_t)q * c;
  G1J acc = u[c - 1];
  for (int i = (int)c - 2; i >= 0; i--) acc = g1_add(g1_double(acc), u[i]);
  G1J r = g1_zero();
  if (!l_is_zero(acc.z)) {
    Fp zi = fp_inv(acc.z), zi2 = fp_sqr(zi);
    r.x = fp_mul(acc.x, zi2);
    r.y = fp_mul(acc.y, fp_mul(zi2, zi));
    r.z = fp_one();
  }
  out[q] = r;
}

// q MSMs over the table's n bases; mags = prepared scalars (k_scalar_prep); W windows in use
static int msm_run(zk_ctx* ctx, const G1Table& T, const Fr* mags, uint32_t q, uint32_t W, G1J* out) {
  const uint32_t n = T.n, c = T.c;
  if (W > T.W) return ctx->fail(ZK_ERR_ARG, "MSM: table has too few windows");
  const uint32_t B = 1u << (c - 1);
  RecodeK K = make_recode_k(c, W);
  ZK_ALLOC(ctx, idx, uint32_t, (size_t)q * W * n);
  ZK_ALLOC(ctx, off, uint32_t, (size_t)q * W * (B + 2));
  ZK_ALLOC(ctx, buckets, G1J, (size_t)q * B);
  ZK_ALLOC(ctx, U, G1J, (size_t)q * c);
  size_t lds = (size_t)(B + 2 + 256) * 4;
  uint64_t pairs = (uint64_t)q * n;
  ZK_LAUNCH(ctx, "msm_sort", 32ull * pairs * W, k_msm_sort, dim3(W, q), dim3(256), lds, mags, n, W, c, K, (const uint8_t*)T.inf, idx.get(), off.get());
  ZK_LAUNCH(ctx, "msm_accum", 100ull * pairs * W, k_msm_accum, dim3((B + 63) / 64, q), dim3(64), 0, (const G1A*)T.pts, n, W, c, (const uint32_t*)idx.get(),
            (const uint32_t*)off.get(), buckets.get());
  if (B >= 1024) ZK_LAUNCH(ctx, "msm_bucket_reduce", 144ull * q * B, k_msm_bits<256>, dim3(c, q), dim3(256), 0, (const G1J*)buckets.get(), c, U.get());
  else ZK_LAUNCH(ctx, "msm_bucket_reduce", 144ull * q * B, k_msm_bits<64>, dim3(c, q), dim3(64), 0, (const G1J*)buckets.get(), c, U.get());
  ZK_LAUNCH(ctx, "msm_horner", 144ull * q * c, k_msm_horner, dim3((q + 63) / 64), dim3(64), 0, (const G1J*)U.get(), c, q, out);
  return ZK_OK;
}

static uint32_t pick_window(zk_ctx* ctx, uint32_t n, uint32_t bits) {
  if (ctx->msm_window) return ctx->msm_window < 2 ? 2 : (ctx->msm_window > 13 ? 13 : ctx->msm_window);
  if (bits <= 24) {
    // short scalars (quantised weights): few windows, small bucket sets
    uint32_t c = 7;
    return c;
  }
  uint32_t lg = ceil_log2(n);
  uint32_t c = lg + 1;
  if (c < 8) c = 8;
  if (c > 13) c = 13;
  return c;
}

// general entry: scalars are Fr values (Montgomery form if do_unmont, else raw canonical integers)
int msm_batch_dev(zk_ctx* ctx, zk_gens* g, const Fr* scalars, uint32_t q, bool do_unmont, bool probe_bits, G1J* out) {
  const uint32_t n = g->n;
  if (q == 0) return ZK_OK;
  uint64_t total = (uint64_t)q * n;
  ZK_ALLOC(ctx, mags, Fr, total);
  ZK_ALLOC(ctx, maxbits, uint32_t, 1);
  ZK_HIP(ctx, hipMemsetAsync(maxbits.get(), 0, 4, ctx->stream));
  ZK_LAUNCH(ctx, "msm_scalar_prep", 64ull * total, k_scalar_prep, dim3(stream_grid(total)), dim3(256), 0, scalars, mags.get(), total, do_unmont ? 1 : 0, maxbits.get());
  uint32_t bits = 254;
  if (probe_bits) {  // setup-time paths only (commit): one host round trip buys 10x fewer windows
    ZK_HIP(ctx, hipMemcpyAsync(&bits, maxbits.get(), 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  uint32_t c = pick_window(ctx, n, bits);
  uint32_t W = (bits + 2 + c - 1) / c;  // +2 bits of head-room so mag + K never spills past the top window
  if (W == 0) W = 1;
  const G1Table* T = nullptr;
  ZK_TRY(gens_table(ctx, g, c, W, &T));
  return msm_run(ctx, *T, mags.get(), q, W, out);
}

// Commitment::commit, intended row semantics: out[row] = sum_j gens[j] * unmont(t[row*n + j])
int commit_pippenger_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, uint32_t tn, G1J* out) {
  return msm_batch_dev(ctx, g, t, tn / g->n, /*do_unmont=*/true, /*probe_bits=*/true, out);
}

// ------------------------------------------------------------------ flattened opening
// C[(2^r - 1) + l] = prod_{t<r} (bit_t(l) ? 1 - u_t : u_t)   (Montgomery form), r = 0..k
__global__ void __launch_bounds__(256) k_mo_coef(const Fr* __restrict__ u, uint32_t k, Fr* __restrict__ C) {
  uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t total = (2u << k) - 1u;
  if (idx >= total) return;
  uint32_t r = 31u - __clz(idx + 1u);
  uint32_t l = idx + 1u - (1u << r);
  Fr one = fr_one(), acc = one;
  for (uint32_t t = 0; t < r; t++) {
    Fr ut = u[t];
    acc = fr_mul(acc, ((l >> t) & 1u) ? fr_sub(one, ut) : ut);
  }
  C[idx] = acc;
}
// scal[(3r + e)*n + i] for rounds r < k, e in {T, T0, T1}; scal[3k*n + i] = unmont(C_k[i]).
// S holds the folded scalar tensors back to back: S_r at offset n*(2 - 2^(1-r)) = 2n - (2n >> r).
__global__ void __launch_bounds__(256) k_mo_scalars(const Fr* __restrict__ C, const Fr* __restrict__ S, uint32_t n, uint32_t k, Fr* __restrict__ scal) {
  uint64_t total = (uint64_t)(k + 1) * n;
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t r = (uint32_t)(g / n), i = (uint32_t)(g - (uint64_t)r * n);
    if (r == k) {
      scal[(uint64_t)3 * k * n + i] = fr_unmont(C[(1u << k) - 1u + i]);
      continue;
    }
    uint32_t j = i >> r, l = i & ((1u << r) - 1u);
    Fr cf = C[(1u << r) - 1u + l];
    const Fr* Sr = S + (2ull * n - ((2ull * n) >> r));
    Fr zero = fr_zero();
    // mont_mul(C~, s^) = C * s^ mod r as an integer: the MSM scalar of original generator i
    scal[(uint64_t)(3 * r + 0) * n + i] = fr_mul(cf, Sr[j]);
    scal[(uint64_t)(3 * r + 1) * n + i] = (j & 1u) ? fr_mul(cf, Sr[j - 1]) : zero;  // T0 = sum g_{2j+1} * s_{2j}
    scal[(uint64_t)(3 * r + 2) * n + i] = (j & 1u) ? zero : fr_mul(cf, Sr[j + 1]);  // T1 = sum g_{2j} * s_{2j+1}
  }
}

int me_open_flat_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, const Fr* u_d, uint32_t k, G1J* proof_d, Fr* ret_d) {
  const uint32_t n = g->n;
  if (k > 20 || n != (1u << k)) {
    // the flattened form needs a full binary tree; odd sizes take the literal path
    return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions (flattened me_open needs n == 2^k)");
  }
  // folded scalar tensors S_0 .. S_k (Fr only, tiny)
  ZK_ALLOC(ctx, S, Fr, 2ull * n + 2);
  ZK_HIP(ctx, hipMemcpyAsync(S.get(), t, (size_t)n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
  for (uint32_t r = 0; r < k; r++) {
    Fr* cur = S.get() + (2ull * n - ((2ull * n) >> r));
    Fr* nxt = S.get() + (2ull * n - ((2ull * n) >> (r + 1)));
    ZK_TRY(fr_fold_dev(ctx, cur, nxt, u_d + r, n >> r, n >> (r + 1)));
  }
  ZK_HIP(ctx, hipMemcpyAsync(ret_d, S.get() + (2ull * n - 2), sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));  // t(0) = S_k[0]
  uint32_t ncoef = (2u << k) - 1u;
  ZK_ALLOC(ctx, C, Fr, ncoef);
  ZK_LAUNCH(ctx, "me_open_scalars", 32ull * ncoef, k_mo_coef, dim3((ncoef + 255) / 256), dim3(256), 0, u_d, k, C.get());
  uint32_t q = 3 * k + 1;
  ZK_ALLOC(ctx, scal, Fr, (size_t)q * n);
  ZK_LAUNCH(ctx, "me_open_scalars", 32ull * q * n, k_mo_scalars, dim3(stream_grid((uint64_t)(k + 1) * n)), dim3(256), 0, (const Fr*)C.get(), (const Fr*)S.get(), n, k,
            scal.get());
  return msm_batch_dev(ctx, g, scal.get(), q, /*do_unmont=*/false, /*probe_bits=*/false, proof_d);
}

// com(u): sum_i eq(unmont u, i) * P_i over the (precomputed) commitment points -- the MSM form of
// G1TensorJacobian::operator()(u) (g1-tensor.cu:463-491)
__global__ void __launch_bounds__(256) k_eq_scalars(const Fr* __restrict__ u, uint32_t k, uint32_t m, Fr* __restrict__ out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  Fr one = fr_one(), acc = one;
  for (uint32_t b = 0; b < k; b++) {
    Fr ub = u[b];
    acc = fr_mul(acc, ((j >> b) & 1u) ? ub : fr_sub(one, ub));
  }
  out[j] = fr_unmont(acc);
}
int g1_me_msm_dev(zk_ctx* ctx, zk_gens* pts, const Fr* u_d, uint32_t k, G1J* dst_d) {
  const uint32_t m = pts->n;
  if (m == 0 || k > 31 || m > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (k > 0 && m <= (1u << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  ZK_ALLOC(ctx, scal, Fr, m);
  ZK_LAUNCH(ctx, "me_open_scalars", 32ull * m, k_eq_scalars, dim3((m + 255) / 256), dim3(256), 0, u_d, k, m, scal.get());
  return msm_batch_dev(ctx, pts, scal.get(), 1, /*do_unmont=*/false, /*probe_bits=*/false, dst_d);
}

}  // namespace zk
