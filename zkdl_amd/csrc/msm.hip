// zkdl_amd/csrc/msm.hip -- bucket-method (Pippenger) multi-scalar multiplication over SHARED bases
// for gfx950, and the commitment/opening paths built on it (ZK_MSM_PIPPENGER back end).
//
// What the reference does: one 256-step double-and-add per (generator, scalar) pair, then sums
// (g1-tensor.cu:422-445, commitment.cu:3-81) -- m*n independent ~10-ms chains for commit and five
// chained scalar-muls per thread per round for me_open.  What this file does instead:
//
//  * Window tables.  Every MSM on the path runs over a FIXED point set (a layer's generators, or a
//    layer's commitment points), so 2^(c*w) * G_j is precomputed once per set in affine form (zk_gens,
//    built lazily).  All windows of an MSM then share ONE bucket set: no Horner doubling chain over
//    windows on the critical path.
//  * Signed digits.  A scalar s < r is folded to (sign, min(s, r-s)) -- so quantised weights
//    (|w| < 2^13, half of them stored as r - |w|) need two 7-bit windows instead of 37 -- then recoded
//    to digits in [-2^(c-1), 2^(c-1)) by adding one constant (no inter-window carries): 2^(c-1) buckets.
//  * Skew-proof accumulation.  A single G1 mixed add has ~28 us latency on one wave, so the critical
//    path is what matters.  Per MSM, ONE workgroup counting-sorts all (point, window) items by bucket
//    in LDS (k_msm_sort); then each LANE takes a fixed-size chunk of L consecutive items of the sorted
//    list (k_msm_accum: every lane does exactly L mixed adds whatever the digit distribution -- a
//    partially filled top window or repeated scalars cannot overload a lane); buckets that straddle
//    chunks are stitched from per-chunk head/tail partials (k_msm_combine, four lanes per bucket).
//  * Bucket reduction without a running-sum chain and without idle lanes: sum_b b*S_b with b = hi*2^L + lo is
//    sum_lo lo*C_lo + 2^L * sum_hi hi*R_hi over row sums R and column sums C (k_msm_rowcol: 2B additions,
//    four lanes per sum), then the two small weighted sums bit-slice by bit-slice plus a doubling + 4-level
//    tree per MSM (k_msm_weighted).
//  * Deferred batching.  Nothing in the protocol consumes a G1 value (the reference drops them all),
//    so zkFC::prove only ENQUEUES its MSMs as jobs; every MSM of a whole proof (all layers, ~270 of
//    them over 16 different tables) is executed by one sort/accum/combine/bits/horner sequence when the
//    transcript is read, digested or flushed.
//  * Flattened opening.  Every value Commitment::me_open pushes (T, T0, T1 per round, the final folded
//    generator) is a linear combination of the ORIGINAL generators with scalars computable in Fr alone
//    (DESIGN.md "Flattened opening"): 3k+1 independent MSMs instead of k dependent rounds.
//
// Same group elements as the reference, different Jacobian representatives: results are emitted in
// canonical form (affine x, y with Z = ONE; infinity as {0, ONE, 0}) and compared affine-canonically.
// All of this is integer-VALU bound (v_mad_u64_u32), not HBM bound; tables stay L2/MALL resident.
#include "layers.hpp"

#include "dev_util.hpp"
#include "fq28.hpp"
#include "g1q.hpp"

namespace zk {

#ifndef ZK_MSM_CHUNK
#define ZK_MSM_CHUNK 16
#endif
static constexpr uint32_t MSM_CHUNK = ZK_MSM_CHUNK;  // items per lane in k_msm_accum
#ifndef ZK_ACCUM_INLINE
#define ZK_ACCUM_INLINE 1
#endif
#ifndef ZK_ACCUM_WAVES
#define ZK_ACCUM_WAVES 2
#endif
static constexpr bool ACCUM_INLINE = ZK_ACCUM_INLINE;

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

// window tables hold x * 2^392 mod p (the Montgomery domain of fq28.hpp) in the 12 x 32-bit layout: eight modular doublings
// of the library's x * 2^384 mod p
__global__ void __launch_bounds__(256) k_table_nform(G1A* __restrict__ pts, uint64_t total) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  G1A p = pts[i];
#pragma unroll 1
  for (int d = 0; d < 8; d++) {
    p.x = fp_dbl(p.x);
    p.y = fp_dbl(p.y);
  }
  pts[i] = p;
}

int gens_table(zk_ctx* ctx, zk_gens* g, uint32_t c, uint32_t W, const G1Table** out) {
  auto it = g->tables.find(c);
  if (it != g->tables.end() && it->second.W >= W) {
    *out = &it->second;
    return ZK_OK;
  }
  G1Table T;
  T.n = g->n;
  T.c = c;
  T.W = W;
  size_t total = (size_t)W * g->n;
  if (dev_alloc(ctx, (void**)&T.pts, total * sizeof(G1A)) != hipSuccess || dev_alloc(ctx, (void**)&T.inf, total) != hipSuccess)
    return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: MSM window table");
  {
    ZK_ALLOC(ctx, jac, G1J, total);
    ZK_LAUNCH(ctx, "msm_table_build", 144ull * total, k_table_chain, dim3((g->n + 63) / 64), dim3(64), 0, (const G1J*)g->pts, g->n, c, W, jac.get());
    ZK_TRY(g1_to_affine_dev(ctx, jac.get(), (uint32_t)total, T.pts, T.inf));
    ZK_LAUNCH(ctx, "msm_table_nform", 192ull * total, k_table_nform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, T.pts, (uint64_t)total);
  }
  if (it != g->tables.end()) {
    // a wider table replaces a narrower one IN PLACE: queued jobs hold a pointer to this map node and
    // pick up the new (superset) arrays when they run; the old arrays go once the stream has drained
    // (queued jobs have not been launched yet; launched ones are ordered before these frees, including a
    // sealed batch still running on the G1 stream)
    ctx->wait_g1(ctx->stream);
    dev_free(ctx, it->second.pts);
    dev_free(ctx, it->second.inf);
    it->second = T;
    *out = &it->second;
    return ZK_OK;
  }
  g->tables[c] = T;
  *out = &g->tables[c];
  return ZK_OK;
}

// ------------------------------------------------------------------ scalar preparation
// mag = min(s, r - s) with the sign in bit 255 (mag < 2^254); *maxbits = max bit length of mag.
__device__ __forceinline__ void k_scalar_prep(uint32_t zk_nblk, const Fr* __restrict__ s, Fr* __restrict__ out, uint64_t total, int do_unmont, uint32_t* __restrict__ maxbits) {
  uint32_t mx = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)zk_nblk * blockDim.x) {
    Fr v = s[i];
    if (do_unmont) v = fr_unmont(v);
    Fr nv = fr_neg(v);         // r - v (0 for v == 0)
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

// a caller-stated bound on the scalars' bit length (zk_act_commit) is checked, not trusted: a longer scalar raises the context's sticky range flag (pinned
// host word), which zk_ctx_sync turns into ZK_ERR_RANGE
__device__ __forceinline__ void k_bits_guard(uint32_t zk_nblk, const uint32_t* __restrict__ maxbits, uint32_t bound, uint32_t* __restrict__ flag) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && *maxbits > bound) {
    __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// K = sum_{w < W} 2^(c*w + c - 1): every c-bit window of mag + K equals (signed digit + 2^(c-1)).
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
struct Recoded {
  uint32_t t[10];
  bool sneg;
};
__device__ __forceinline__ Recoded recode_prepare(const Fr& mag, const RecodeK& K) {
  Recoded r;
  uint32_t carry = 0;
  r.sneg = (mag.v[7] >> 31) != 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t m = i == 7 ? (mag.v[7] & 0x7fffffffu) : mag.v[i];
    uint32_t co;
    r.t[i] = __builtin_addc(m, K.v[i], carry, &co);
    carry = co;
  }
  r.t[8] = K.v[8] + carry;
  r.t[9] = 0;
  return r;
}
// bucket index |d| of window w (0 = skip) and the sign of the point to add
__device__ __forceinline__ uint32_t recode_digit(const Recoded& r, uint32_t c, uint32_t w, bool& negative) {
  uint32_t bit = c * w, li = bit >> 5, sh = bit & 31;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    lo = (li == (uint32_t)i) ? r.t[i] : lo;
    hi = (li == (uint32_t)i) ? r.t[i + 1] : hi;
  }
  uint64_t both = ((uint64_t)hi << 32) | lo;
  uint32_t dp = (uint32_t)(both >> sh) & ((1u << c) - 1u);
  int32_t d = (int32_t)dp - (int32_t)(1u << (c - 1));
  negative = (d < 0) != r.sneg;
  return (uint32_t)(d < 0 ? -d : d);
}

// ------------------------------------------------------------------ bucket coordinates: extended Jacobian (XYZZ)
// Inside the MSM pipeline a point is (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 (infinity: ZZ = 0):
// a mixed addition costs 8M + 2S instead of madd-2007-bl's 7M + 4S and a full addition 12M + 2S instead of
// 11M + 5S (EFD "xyzz": madd-2008-s, add-2008-s, dbl-2008-s-1).  Buckets, chunk partials and row/column sums
// never leave the pipeline, so their coordinates live in the 14 x 28-bit lazily reduced form of fq28.hpp (a product is
// 392 single-instruction MACs instead of 288 MAC + carry pairs; sums and differences are carry-free).  The result
// goes back to Jacobian as (X*ZZ*ZZZ^2, Y*ZZ^3*ZZZ^2, ZZ*ZZZ), converted to the library's canonical 12 x 32-bit
// Montgomery form on the way out -- the same group element, and the canonical (affine) form the transcript keeps
// is unaffected.
//
// Bounds (units of p; "n" = normalised limbs).  Stored X < 18, Y < 6, ZZ, ZZZ < 2, all n.  Every product below has
// operand values whose product is < 2^11 and limbs < 2^28 x 2^32 or < 2^29.6 x 2^29.6 (fq28.hpp), so it returns n, < 2.
struct alignas(16) G1X {
  Fq x, y, zz, zzz;
};
__device__ __forceinline__ G1X g1x_zero() {
  G1X r;
  r.x = fq_zero();
  r.y = fq_one();
  r.zz = fq_zero();
  r.zzz = fq_zero();
  return r;
}
__device__ __forceinline__ bool g1x_is_zero(const G1X& a) { return fq_all_zero(a.zz); }
__device__ __forceinline__ G1X g1x_double(const G1X& a) {
  if (g1x_is_zero(a)) return a;
  Fq u = fq_add(a.y, a.y);                 // < 12, limbs < 2^29
  Fq v = fq_sqr(u);
  Fq w = fq_mul(u, v);
  Fq s_ = fq_mul(a.x, v);
  Fq xx = fq_sqr(a.x);
  Fq m = fq_add(fq_add(xx, xx), xx);       // < 6, limbs < 2^29.6
  G1X o;
  o.x = fq_norm(fq_sub3<16>(fq_sqr(m), fq_zero(), s_));                          // m^2 - 2s: < 2 + 16
  o.y = fq_norm(fq_sub<4>(fq_mul(m, fq_sub<36>(s_, o.x)), fq_mul(w, a.y)));      // < 2 + 4
  o.zz = fq_mul(v, a.zz);
  o.zzz = fq_mul(w, a.zzz);
  return o;
}
// INL: the products inlined into the caller (the hot loop of k_msm_accum: no argument shuffling around ten calls) or
// real calls (everywhere else: code size)
template <bool INL>
__device__ __forceinline__ Fq fq_mul_t(const Fq& a, const Fq& b) {
  if constexpr (INL) return fq_mul_inl(a, b);
  else return fq_mul(a, b);
}
template <bool INL>
__device__ __forceinline__ Fq fq_sqr_t(const Fq& a) {
  if constexpr (INL) return fq_sqr_inl(a);
  else return fq_sqr(a);
}
// shared tail of the two additions: (u1, s1) the first point and (pd, r) = (u2 - u1, s2 - s1), pp = pd^2, zz/zzz the
// products of the inputs' ZZ / ZZZ
template <bool INL = false>
__device__ __forceinline__ G1X g1x_add_tail(const Fq& u1, const Fq& s1, const Fq& pd, const Fq& r, const Fq& pp, const Fq& zz, const Fq& zzz) {
  Fq ppp = fq_mul_t<INL>(pd, pp);
  Fq q = fq_mul_t<INL>(u1, pp);
  G1X o;
  o.x = fq_norm(fq_sub3<16>(fq_sqr_t<INL>(r), ppp, q));                                          // r^2 - ppp - 2q: < 2 + 16 = 18
  o.y = fq_norm(fq_sub<4>(fq_mul_t<INL>(r, fq_sub<36>(q, o.x)), fq_mul_t<INL>(s1, ppp)));        // < 2 + 4 = 6
  o.zz = fq_mul_t<INL>(zz, pp);
  o.zzz = fq_mul_t<INL>(zzz, ppp);
  return o;
}
// b: affine table point, n-form, normalised, < 1
template <bool INL = false>
__device__ __forceinline__ G1X g1x_madd(const G1X& a, const Fq& bx, const Fq& by) {
  if (g1x_is_zero(a)) {
    G1X o;
    o.x = bx;
    o.y = by;
    o.zz = fq_one();
    o.zzz = fq_one();
    return o;
  }
  Fq u2 = fq_mul_t<INL>(bx, a.zz);
  Fq s2 = fq_mul_t<INL>(by, a.zzz);
  Fq pd = fq_sub<36>(u2, a.x);   // < 2 + 36, limbs < 2^29.6
  Fq r = fq_sub<16>(s2, a.y);    // < 2 + 16
  Fq pp = fq_sqr_t<INL>(pd);     // 38^2 p/R' + 1 < 1.6
  if (fq_is_zero_mod_p(pp)) return fq_is_zero_mod_p(fq_sqr(r)) ? g1x_double(a) : g1x_zero();
  return g1x_add_tail<INL>(a.x, a.y, pd, r, pp, a.zz, a.zzz);
}
__device__ __forceinline__ G1X g1x_add(const G1X& a, const G1X& b) {
  if (g1x_is_zero(a)) return b;
  if (g1x_is_zero(b)) return a;
  Fq u1 = fq_mul(a.x, b.zz);
  Fq u2 = fq_mul(b.x, a.zz);
  Fq s1 = fq_mul(a.y, b.zzz);
  Fq s2 = fq_mul(b.y, a.zzz);
  Fq pd = fq_sub<4>(u2, u1);
  Fq r = fq_sub<4>(s2, s1);
  Fq pp = fq_sqr(pd);
  if (fq_is_zero_mod_p(pp)) return fq_is_zero_mod_p(fq_sqr(r)) ? g1x_double(a) : g1x_zero();
  return g1x_add_tail(u1, s1, pd, r, pp, fq_mul(a.zz, b.zz), fq_mul(a.zzz, b.zzz));
}
__device__ __forceinline__ G1J g1x_to_jacobian(const G1X& a) {
  if (g1x_is_zero(a)) return g1_zero();
  Fq z = fq_mul(a.zz, a.zzz);                 // Z = ZZ*ZZZ: Z^2 = ZZ^2 ZZZ^2, Z^3 = ZZ^3 ZZZ^3
  Fq t = fq_mul(a.zz, fq_sqr(a.zzz));         // ZZ ZZZ^2
  G1J o;
  o.x = fq_to_fp(fq_mul(a.x, t));                       // x Z^2 = X/ZZ * ZZ^2 ZZZ^2
  o.y = fq_to_fp(fq_mul(a.y, fq_mul(t, fq_sqr(a.zz))));  // y Z^3 = Y/ZZZ * ZZ^3 ZZZ^3
  o.z = fq_to_fp(z);
  return o;
}
__device__ __forceinline__ G1X g1x_shfl_xor(const G1X& p, int mask) {
  G1X r;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    r.x.v[i] = __shfl_xor(p.x.v[i], mask, 64);
    r.y.v[i] = __shfl_xor(p.y.v[i], mask, 64);
    r.zz.v[i] = __shfl_xor(p.zz.v[i], mask, 64);
    r.zzz.v[i] = __shfl_xor(p.zzz.v[i], mask, 64);
  }
  return r;
}

// ------------------------------------------------------------------ job table
struct MsmJob {          // one MSM: out = sum_j scalar_j * base_j
  const G1A* table;      // [W][n] affine window table of the base set
  const uint8_t* inf;    // [n] base-at-infinity flags
  const Fr* mags;        // [n] prepared scalars (k_scalar_prep)
  G1J* out;
  uint32_t n, W;
  uint64_t item_off;     // first slot of this job in the sorted-item array
  uint64_t chunk_off;    // first slot of this job in the head/tail partial arrays
};

// ------------------------------------------------------------------ sort all (point, window) items of one MSM by bucket
// grid: one workgroup per job; LDS: hist[B+2] + 1024 scan slots.
// idx[item_off + pos] = j | w << 20 | sign << 31, grouped by bucket; off[job*(B+2) + b] = first pos of bucket b.
__global__ void __launch_bounds__(1024) k_msm_sort(const MsmJob* __restrict__ jobs, uint32_t c, RecodeK K, uint32_t* __restrict__ idx, uint32_t* __restrict__ off) {
  extern __shared__ uint32_t lds[];
  const uint32_t B = 1u << (c - 1);
  uint32_t* hist = lds;            // B + 2
  uint32_t* part = lds + (B + 2);  // 1024
  const MsmJob job = jobs[blockIdx.x];
  const uint32_t tid = threadIdx.x, n = job.n, W = job.W;
  for (uint32_t b = tid; b < B + 2; b += 1024) hist[b] = 0;
  __syncthreads();
  for (uint32_t j = tid; j < n; j += 1024) {
    if (job.inf[j]) {
      atomicAdd(&hist[0], W);
      continue;
    }
    Recoded r = recode_prepare(job.mags[j], K);
    for (uint32_t w = 0; w < W; w++) {
      bool neg;
      atomicAdd(&hist[recode_digit(r, c, w, neg)], 1u);
    }
  }
  __syncthreads();
  const uint32_t per = (B + 2 + 1023) / 1024;
  uint32_t s = 0;
  for (uint32_t i = 0; i < per; i++) {
    uint32_t b = tid * per + i;
    if (b < B + 2) s += hist[b];
  }
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0;
  uint32_t* o = off + (uint64_t)blockIdx.x * (B + 2);
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
  uint32_t* out = idx + job.item_off;
  for (uint32_t j = tid; j < n; j += 1024) {
    if (job.inf[j]) continue;  // bucket-0 slots are never read
    Recoded r = recode_prepare(job.mags[j], K);
    for (uint32_t w = 0; w < W; w++) {
      bool neg;
      uint32_t b = recode_digit(r, c, w, neg);
      if (b == 0) continue;
      uint32_t pos = atomicAdd(&hist[b], 1u);
      out[pos] = j | (w << 20) | (neg ? 0x80000000u : 0u);
    }
  }
}

// Each lane adds MSM_CHUNK consecutive items of the bucket-sorted list.  A bucket wholly inside the
// chunk is stored to buckets[]; a bucket that started earlier goes to head[chunk], one that continues
// later to tail[chunk].  grid (chunk blocks, jobs).
// Stitching inside the wave: with ~n / 2^(c-1) items per bucket and 16 per chunk almost every straddling bucket spans exactly two
// NEIGHBOURING chunks, i.e. two neighbouring lanes.  When both are lanes of this wave the second lane leaves its head piece in LDS and the
// first adds it to its own last run and stores the complete bucket -- one more addition per wave (+9 % on sixteen mixed additions) instead
// of a pass over every bucket of every MSM with two to four lanes each (k_msm_combine: 0.42 of this kernel's time on 4096-generator
// layers, 0.12 of the issue peak); k_msm_combine keeps the buckets that cross a wave boundary or span three chunks or more.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZK_ACCUM_WAVES, ZK_ACCUM_WAVES))) k_msm_accum(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t chunk, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ off,
                                                   G1X* __restrict__ buckets, G1X* __restrict__ head, G1X* __restrict__ tail) {
  const uint32_t B = 1u << (c - 1);
  const uint32_t q = blockIdx.y;
  const MsmJob job = jobs[q];
  const uint32_t* o = off + (uint64_t)q * (B + 2);
  const uint32_t base = o[1], total = o[B + 1];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t cs64 = (uint64_t)base + (uint64_t)t * chunk;
  if (cs64 >= total) return;
  const uint32_t cs = (uint32_t)cs64;
  const uint32_t ce = cs + chunk < total ? cs + chunk : total;
  // bucket containing item cs: largest b in [1, B] with o[b] <= cs
  uint32_t lo = 1, hi = B;
  while (lo < hi) {
    uint32_t mid = (lo + hi + 1) >> 1;
    if (o[mid] <= cs) lo = mid;
    else hi = mid - 1;
  }
  uint32_t b = lo;
  uint32_t bend = o[b + 1];
  uint32_t run_start = cs;
  const uint32_t* items = idx + job.item_off;
  G1X acc = g1x_zero();
  G1X* bk = buckets + (uint64_t)q * B;
  __shared__ G1X stitch[64];
  // this lane's FIRST run is the second half of a bucket that began in the previous lane's chunk and ends in this one
  const bool give = threadIdx.x > 0 && o[b] < cs && o[b] + chunk >= cs && bend <= ce;
  for (uint32_t p = cs; p < ce; p++) {
    while (p >= bend) {  // bucket b ends here (possibly followed by empty buckets)
      if (p > run_start) {
        if (run_start == o[b]) bk[b - 1] = acc;                     // complete: started and ended inside this chunk
        else if (give) stitch[threadIdx.x] = acc;                    // started in the previous lane's chunk: that lane completes it
        else head[job.chunk_off + t] = acc;                          // started in an earlier chunk
        acc = g1x_zero();
      }
      b++;
      bend = o[b + 1];
      run_start = p;
    }
    uint32_t it = items[p];
    uint32_t j = it & 0xfffffu, w = (it >> 20) & 0x7ffu;
    G1A P;
    {
      const __attribute__((address_space(1))) uint32_t* src = (const __attribute__((address_space(1))) uint32_t*)&job.table[(uint64_t)w * job.n + j];
#pragma unroll
      for (int l = 0; l < 12; l++) {
        P.x.v[l] = src[l];
        P.y.v[l] = src[12 + l];
      }
    }
    if (it >> 31) P.y = fp_neg(P.y);
    acc = g1x_madd<ACCUM_INLINE>(acc, fq_unpack(P.x), fq_unpack(P.y));
  }
  // last run of the chunk: [run_start, ce) of bucket b
  bool starts = run_start == o[b], ends = ce == bend;
  // this lane's LAST run is the first half of a bucket that ends in the next lane's chunk (the mirror image of `give` there)
  const bool take = !ends && threadIdx.x < 63u && o[b] >= cs && (uint64_t)bend <= (uint64_t)cs + 2ull * chunk;
  if (starts && ends) bk[b - 1] = acc;
  else if (take) {}                                           // completed below
  else if (run_start == cs) {                                 // reaches back to the chunk start (spans the whole chunk or came from before)
    if (give && ends) stitch[threadIdx.x] = acc;
    else head[job.chunk_off + t] = acc;
  } else tail[job.chunk_off + t] = acc;                       // starts inside, continues into the next chunk
  __syncthreads();                                            // (one wave per workgroup: lanes that left early do not take part)
  if (take) bk[b - 1] = g1x_add(acc, stitch[threadIdx.x + 1]);
}

// LPB lanes per (job, bucket): empty -> infinity; split across chunks -> sum of the chunk partials, each lane
// adding every LPB-th partial and log2(LPB) butterfly steps joining them.  (The top window of a scalar < r/2 has one
// or two bits of entropy, so for some window widths half of all points of an MSM land in one bucket: a single
// lane would walk ~100 additions there.)
template <int LPB>  // lanes per bucket: 2 when buckets span few chunks, 4 when a near-empty top window makes hot buckets
__global__ void __launch_bounds__(64) k_msm_combine(const MsmJob* __restrict__ jobs, uint32_t c, uint32_t chunk, const uint32_t* __restrict__ off, G1X* __restrict__ buckets,
                                                     const G1X* __restrict__ head, const G1X* __restrict__ tail) {
  const uint32_t B = 1u << (c - 1);
  const uint32_t q = blockIdx.y;
  const uint32_t sub = threadIdx.x % LPB;
  // Workgroups reach the XCDs (and the compute units inside one) round-robin in launch order, and the crowded buckets of every MSM -- the few magnitudes the
  // partial top window can take, up to half of an MSM's points in ONE bucket -- sit in the same block of every job: with a block count that is a multiple of 8
  // all the long chains of a batch landed on four compute units of one XCD (measured, 2^24 pairs at c = 11 / 12: 28-48 ms of a 95-120 ms batch in this kernel
  // against 6 ms at c = 10, whose top window is five bits wide).  Rotating the block order by the job index spreads them over the chip.
  const uint32_t xb = (blockIdx.x + q * 13u) % gridDim.x;
  const uint32_t b = xb * (64u / LPB) + (threadIdx.x / LPB) + 1;
  const bool live = b <= B;
  const MsmJob job = jobs[q];
  const uint32_t* o = off + (uint64_t)q * (B + 2);
  const uint32_t base = o[1];
  const uint32_t s = live ? o[b] : 0, e = live ? o[b + 1] : 0;
  G1X acc = g1x_zero();
  bool store = live && s == e;  // empty bucket: infinity
  bool pieces = false;          // this bucket's chunk partials are added here (then the butterfly below has something to join)
  if (live && s != e) {
    const uint32_t t0 = (s - base) / chunk, t1 = (e - 1 - base) / chunk;
    // (t0 == t1: complete inside one chunk; two neighbouring chunks of one wave: stitched there -- both already stored by k_msm_accum)
    if (t0 != t1 && !(t1 == t0 + 1 && (t0 & 63u) != 63u)) {
      store = true;
      pieces = true;
      // chunk t0 holds the bucket's first piece: a tail unless the bucket starts exactly at the chunk start
      if (sub == 0) acc = ((s - base) == t0 * chunk) ? head[job.chunk_off + t0] : tail[job.chunk_off + t0];
      for (uint32_t t = t0 + 1 + sub; t <= t1; t += LPB) acc = g1x_add(acc, head[job.chunk_off + t]);
    }
  }
  // with the stitching inside k_msm_accum most waves have no bucket left to join: the butterfly (LPB / 2 .. LPB full additions per lane, whatever the
  // operands) is skipped wave by wave
  if (__builtin_amdgcn_ballot_w64(pieces) != 0ull) {
#pragma unroll
    for (int m = 1; m < LPB; m <<= 1) acc = g1x_add(acc, g1x_shfl_xor(acc, m));
  }
  if (store && sub == 0) buckets[(uint64_t)q * B + (b - 1)] = acc;
}

// ---- bucket reduction  out[q] = sum_{b=1..B} b * S_b  without a running-sum chain and without idle lanes.
// Write b = hi * 2^LB + lo.  Then  sum_b b*S_b = sum_lo lo*C_lo + 2^LB * sum_hi hi*R_hi  with the row sums
// R_hi = sum_lo S_b and column sums C_lo = sum_hi S_b: 2B additions instead of the (c/2)*B of summing one
// bit-slice of the bucket index at a time over all buckets.  The two small weighted sums (2^LB and B/2^LB + 1
// terms) are then taken bit-slice by bit-slice: T_j, j < c, and out = sum_j 2^j T_j.
// Both kernels keep every lane busy with a short sequential run of additions and use few waves, so they
// share the machine with the next proof's Fr work instead of parking thousands of barrier-bound waves on it.
struct ReduceShape {
  uint32_t LB, nlo, nhi, nsums;  // sums 0..nhi-1: rows, nhi..nhi+nlo-1: columns
};
static ReduceShape reduce_shape(uint32_t c) {
  ReduceShape s;
  s.LB = c / 2;  // c = 10: 32 columns x 17 rows
  s.nlo = 1u << s.LB;
  s.nhi = ((1u << (c - 1)) >> s.LB) + 1u;
  s.nsums = s.nhi + s.nlo;
  return s;
}
// RC[q*nsums + s]: four lanes per sum, each adds every fourth term, then two butterfly steps.  grid (blocks, jobs)
__global__ void __launch_bounds__(64) k_msm_rowcol(const G1X* __restrict__ buckets, uint32_t c, ReduceShape sh, G1X* __restrict__ RC) {
  const uint32_t B = 1u << (c - 1);
  const uint32_t q = blockIdx.y;
  const uint32_t lane = blockIdx.x * 64u + threadIdx.x;
  const uint32_t s = lane >> 2, t = lane & 3u;
  const G1X* bk = buckets + (uint64_t)q * B;
  G1X acc = g1x_zero();
  if (s < sh.nhi) {  // row hi = s
    for (uint32_t lo = t; lo < sh.nlo; lo += 4) {
      uint32_t b = s * sh.nlo + lo;
      if (b >= 1 && b <= B) acc = g1x_add(acc, bk[b - 1]);
    }
  } else if (s < sh.nsums) {  // column lo = s - nhi
    const uint32_t lo = s - sh.nhi;
    for (uint32_t hi = t; hi < sh.nhi; hi += 4) {
      uint32_t b = hi * sh.nlo + lo;
      if (b >= 1 && b <= B) acc = g1x_add(acc, bk[b - 1]);
    }
  }
  acc = g1x_add(acc, g1x_shfl_xor(acc, 1));
  acc = g1x_add(acc, g1x_shfl_xor(acc, 2));
  if (s < sh.nsums && t == 0) RC[(uint64_t)q * sh.nsums + s] = acc;
}
// one wave per job: lanes 4j..4j+3 build slice T_j, lane j then doubles it j times, and a 4-level tree adds them
__global__ void __launch_bounds__(64) k_msm_weighted(const MsmJob* __restrict__ jobs, const G1X* __restrict__ RC, uint32_t c, ReduceShape sh) {
  __shared__ G1X T[16];
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  const uint32_t j = lane >> 2, t = lane & 3u;
  const G1X* rc = RC + (uint64_t)q * sh.nsums;
  G1X acc = g1x_zero();
  // the m-th index with bit i set is  ((m >> i) << (i + 1)) | 1 << i | (m & (2^i - 1)):  every iteration adds
  if (j < sh.LB) {  // column weights: bit j of lo
    for (uint32_t m = t; m < sh.nlo / 2; m += 4) {
      uint32_t lo = ((m >> j) << (j + 1)) | (1u << j) | (m & ((1u << j) - 1u));
      acc = g1x_add(acc, rc[sh.nhi + lo]);
    }
  } else if (j < c) {  // row weights: bit j - LB of hi (hi <= nhi - 1, a power of two)
    const uint32_t i = j - sh.LB;
    for (uint32_t m = t; m < sh.nhi - 1; m += 4) {
      uint32_t hi = ((m >> i) << (i + 1)) | (1u << i) | (m & ((1u << i) - 1u));
      if (hi < sh.nhi) acc = g1x_add(acc, rc[hi]);
    }
  }
  acc = g1x_add(acc, g1x_shfl_xor(acc, 1));
  acc = g1x_add(acc, g1x_shfl_xor(acc, 2));
  if (t == 0 && j < 16) T[j] = j < c ? acc : g1x_zero();
  __syncthreads();
  if (lane < 16) {
    acc = T[lane];
    const uint32_t nd = lane < c ? lane : 0;  // lanes >= c hold infinity
    for (uint32_t d = 0; d < nd; d++) acc = g1x_double(acc);
    T[lane] = acc;
  }
  __syncthreads();
  for (uint32_t st = 8; st > 0; st >>= 1) {
    if (lane < st) T[lane] = g1x_add(T[lane], T[lane + st]);
    __syncthreads();
  }
  if (lane == 0) *jobs[q].out = g1x_to_jacobian(T[0]);
}

// canonical form in place: affine x, y with Z = ONE; infinity as {0, ONE, 0}
__global__ void __launch_bounds__(64) k_g1_canonicalise(G1J* __restrict__ pts, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1J p = pts[i];
  if (!l_is_zero(p.z) && l_eq(p.z, fp_one())) return;
  pts[i] = g1q_canonical(g1q_from_jacobian(p));  // one inversion per point: the binary GCD of fpinv.hpp (no field products on the chain)
}
int g1_canonicalise_dev(zk_ctx* ctx, G1J* pts, uint32_t n) {
  if (n) ZK_LAUNCH(ctx, "g1_canonicalise", 288ull * n, k_g1_canonicalise, dim3((n + 63) / 64), dim3(64), 0, pts, n);
  return ZK_OK;
}

// the same over up to 8 ranges in ONE launch (the G1 phases of several proofs run as one batch: the inversion's latency is paid once)
__global__ void __launch_bounds__(64) k_g1_canonicalise_multi(CanonRanges r) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  G1J* pts = nullptr;
  for (uint32_t q = 0; q < r.cnt; q++) {
    if (pts) break;
    if (i < r.n[q]) pts = r.p[q];
    else i -= r.n[q];
  }
  if (!pts) return;
  G1J p = pts[i];
  if (!l_is_zero(p.z) && l_eq(p.z, fp_one())) return;
  pts[i] = g1q_canonical(g1q_from_jacobian(p));
}
int g1_canonicalise_multi_dev(zk_ctx* ctx, const CanonRanges& r) {
  uint32_t n = 0;
  for (uint32_t q = 0; q < r.cnt; q++) n += r.n[q];
  if (n) ZK_LAUNCH(ctx, "g1_canonicalise", 288ull * n, k_g1_canonicalise_multi, dim3((n + 63) / 64), dim3(64), 0, r);
  return ZK_OK;
}

uint32_t msm_pick_window(zk_ctx* ctx, uint32_t bits, uint32_t n) {
  if (ctx->msm_window) return ctx->msm_window;  // zk_set_msm_window admits 2 .. 13 only
  if (bits <= 24) return 7;                     // short scalars (quantised weights): two 7-bit windows
  // full width: the width that leaves 40-50 items per bucket (n W / 2^(c-1)): fewer windows to accumulate while the bucket reduction (2 x 2^(c-1) additions per
  // MSM) stays a tenth of the accumulation and most buckets still end inside one 64-item chunk.  2^24 pairs (profiles/r04_msm_sweep_phases.txt):
  //   4096 generators: c = 10 / 11 / 12 / 13: 72.5 / 76.3 / 68.1 / 68.9 ms;   2048 generators: c = 10 / 11 / 12: 78.2 / 73.7 / 76.0 ms
  return n >= 8192u ? 13 : n >= 4096u ? 12 : n >= 2048u ? 11 : 10;
}
uint32_t msm_num_windows(uint32_t bits, uint32_t c) {
  uint32_t W = (bits + 2 + c - 1) / c;  // +2 bits of head-room so mag + K never spills past the top window
  return W ? W : 1;
}

// ------------------------------------------------------------------ the bucket-free path: full digit tables
// The generators of a layer never change, and 288 GB of HBM hold a lot of table: with EVERY signed digit of every window precomputed
// (d * 2^(c w) * G_j for d = 1 .. 2^(c-1): at c = 12, 22 windows x 2048 entries x 128 B = 5.8 MB per generator, 11.8 GB per 2048-generator
// layer; an entry is 96 bytes in a 128-byte slot, so that a read touches one 128-byte line -- unpadded, the entries straddled lines and the
// accumulation fetched 2.4x its algorithmic bytes) an MSM is nothing but mixed additions of table entries: no buckets, so no sort, no stitching of chunk partials, no row / column /
// weighted bucket reduction -- and 22 windows instead of 26.  Per proof of the 18M-parameter MLP: 4.3 M mixed additions + 0.27 M additions of
// lane partials, against 5.1 M + ~0.7 M additions in three latency-bound reduction launches on the bucket path.  The table reads are random
// one-line accesses into gigabytes: ~0.6 GB of HBM traffic per proof, which the accumulation's arithmetic hides.
//   k_fulltable_build  one lane per (generator, window) row: 2^(c-1) - 1 chained mixed additions, put into affine form 64 at a time with one
//                      shared inversion (Montgomery's trick); setup, ~0.1 s per layer
//   k_msmf_accum       each lane adds MSM_CHUNK consecutive (point, window) items of one job (point-major: a lane recodes a scalar once)
//   k_msmf_reduce      one workgroup per job sums the lanes' partials (a few chained additions per lane, then a shuffle / LDS tree)
struct FullJob {
  const G1AE* tab;
  const uint8_t* inf;
  const Fr* mags;
  G1J* out;
  uint32_t n, W, nnz;
  uint32_t pat_r, pat_par;
  uint64_t part_off;
  uint32_t Wuse;   // windows 0 .. Wuse-1 are walked (scalars known to be short: zk_act_commit); the table's row stride stays W
};
static constexpr uint32_t FT_SEG = 64;  // points put into affine form per shared inversion
#ifndef ZK_FT_SUBS
#define ZK_FT_SUBS 1
#endif
static constexpr uint32_t FT_SUBS = ZK_FT_SUBS;  // chunks of MSM_CHUNK items a lane of k_msmf_accum adds into its one partial
__global__ void __launch_bounds__(64) k_fulltable_build(const G1A* __restrict__ win, const uint8_t* __restrict__ inf, uint32_t n, uint32_t W, uint32_t half,
                                                         G1AE* __restrict__ out, G1X* __restrict__ scratch, Fq* __restrict__ prefix) {
  const uint32_t row = blockIdx.x * 64u + threadIdx.x;  // row = j * W + w
  if (row >= n * W) return;
  const uint32_t j = row / W, w = row % W;
  if (inf[(size_t)w * n + j]) return;  // an infinite base: its entries are never read
  const G1A B = win[(size_t)w * n + j];
  const Fq bx = fq_unpack(B.x), by = fq_unpack(B.y);
  G1X* sc = scratch + (size_t)row * FT_SEG;
  Fq* pr = prefix + (size_t)row * FT_SEG;
  G1AE* dst = out + (size_t)row * half;
  G1X acc = g1x_zero();
  for (uint32_t d0 = 0; d0 < half; d0 += FT_SEG) {
    const uint32_t cnt = half - d0 < FT_SEG ? half - d0 : FT_SEG;
    Fq run = fq_one();
#pragma unroll 1
    for (uint32_t i = 0; i < cnt; i++) {
      acc = g1x_madd(acc, bx, by);  // (d0 + i + 1) * B
      sc[i] = acc;
      run = fq_mul(run, fq_mul(acc.zz, acc.zzz));
      pr[i] = run;
    }
    Fq inv = fq_inv(fq_norm(run));
#pragma unroll 1
    for (uint32_t ii = cnt; ii > 0; ii--) {
      const uint32_t i = ii - 1;
      const G1X p = sc[i];
      const Fq it = i ? fq_mul(inv, pr[i - 1]) : inv;  // 1 / (zz zzz) of point i
      inv = fq_mul(inv, fq_mul(p.zz, p.zzz));
      const Fq x = fq_mul(p.x, fq_mul(it, p.zzz));      // X / ZZ
      const Fq y = fq_mul(p.y, fq_mul(it, p.zz));       // Y / ZZZ
      // canonical, in the pipeline's domain: out of it (x 2^384, fully reduced) and back in by eight modular doublings (as k_table_nform)
      Fp cx = fq_to_fp(x), cy = fq_to_fp(y);
#pragma unroll 1
      for (int t = 0; t < 8; t++) {
        cx = fp_dbl(cx);
        cy = fp_dbl(cy);
      }
      G1A o;
      o.x = cx;
      o.y = cy;
      dst[d0 + i].p = o;
    }
  }
}

// i-th index of { x : ((x >> r) & 1) == par }
__device__ __forceinline__ uint32_t pat_index(uint32_t t, uint32_t r, uint32_t par) {
  if (r >= 32u) return t;
  return ((t >> r) << (r + 1)) | (par << r) | (t & ((1u << r) - 1u));
}
// Pointers that come out of a job record are generic to the compiler (flat loads, which can only be waited for all at once); the tables,
// scalars and flags live in global memory, and saying so gives global loads with counted waits -- i.e. a prefetch that really overlaps.
#define ZK_GLOBAL __attribute__((address_space(1)))
template <bool INL, int AHEAD>
__device__ __forceinline__ void msmf_accum_body(const FullJob* __restrict__ jobs, uint32_t c, const RecodeK& K, G1X* __restrict__ partial) {
  const FullJob job = jobs[blockIdx.y];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t items = (uint64_t)job.nnz * job.Wuse;
  if ((uint64_t)t * (MSM_CHUNK * FT_SUBS) >= items) return;
  const uint32_t half_log = c - 1;
  const ZK_GLOBAL G1AE* tab = (const ZK_GLOBAL G1AE*)job.tab;
  const ZK_GLOBAL Fr* mags = (const ZK_GLOBAL Fr*)job.mags;
  const ZK_GLOBAL uint8_t* inf = (const ZK_GLOBAL uint8_t*)job.inf;
  G1X acc = g1x_zero();
  // FT_SUBS consecutive chunks of MSM_CHUNK items go into ONE partial.  Measured at 2 (round 4): k_msmf_reduce 0.36 -> 0.27 ms per proof (half the partials), this
  // kernel 0.78 -> 0.95 (half the lanes: the last round of waves is half empty on the G1 stream's half of the chip); kept at 1
#pragma unroll 1
  for (uint32_t sub = 0; sub < FT_SUBS; sub++) {
  const uint64_t e0 = ((uint64_t)t * FT_SUBS + sub) * MSM_CHUNK;
  if (e0 >= items) break;
  const uint64_t e1 = e0 + MSM_CHUNK < items ? e0 + MSM_CHUNK : items;
  const uint32_t pt0 = (uint32_t)(e0 / job.Wuse), w0 = (uint32_t)(e0 - (uint64_t)pt0 * job.Wuse);
  // pass 1 (integer work only): the chunk's table offsets; bit 63 = negate, ~0 = nothing to add (digit 0, infinite base).  A chunk of 16
  // point-major items touches at most two points when a point has >= 16 windows (every width up to 16 bits): both scalars are requested at
  // once -- one wait instead of a dependent load + wait per item
  uint64_t offs[MSM_CHUNK];
  if (job.Wuse >= MSM_CHUNK) {
    const uint32_t split = job.Wuse - w0;  // items [0, split) belong to point pt0, the rest to pt0 + 1
    const bool two = e0 + split < e1;
    const uint32_t iA = pat_index(pt0, job.pat_r, job.pat_par), iB = two ? pat_index(pt0 + 1, job.pat_r, job.pat_par) : iA;
    const uint8_t fA = inf[iA], fB = inf[iB];
    Fr mA, mB;
#pragma unroll
    for (int l = 0; l < 8; l++) {
      mA.v[l] = mags[iA].v[l];
      mB.v[l] = mags[iB].v[l];
    }
    const Recoded rA = recode_prepare(mA, K), rB = recode_prepare(mB, K);
#pragma unroll
    for (uint32_t s_ = 0; s_ < MSM_CHUNK; s_++) {
      uint64_t o = ~0ull;
      if (e0 + s_ < e1) {
        const bool second = s_ >= split;
        const uint32_t w = second ? s_ - split : w0 + s_;
        const uint32_t i = second ? iB : iA;
        if (!(second ? fB : fA)) {
          bool nA, nB;
          const uint32_t dA = recode_digit(rA, c, w, nA), dB = recode_digit(rB, c, w, nB);
          const uint32_t d = second ? dB : dA;
          const bool neg = second ? nB : nA;
          if (d) o = (((((uint64_t)i * job.W + w)) << half_log) + (d - 1)) | (neg ? (1ull << 63) : 0ull);
        }
      }
      offs[s_] = o;
    }
  } else {
    Recoded rc;
    uint32_t i = 0, pt = pt0, w = w0;
    bool live = false, fresh = true;
#pragma unroll 1
    for (uint32_t s_ = 0; s_ < MSM_CHUNK; s_++) {
      uint64_t o = ~0ull;
      if (e0 + s_ < e1) {
        if (fresh) {
          i = pat_index(pt, job.pat_r, job.pat_par);
          live = !inf[i];
          if (live) {
            Fr m;
#pragma unroll
            for (int l = 0; l < 8; l++) m.v[l] = mags[i].v[l];
            rc = recode_prepare(m, K);
          }
          fresh = false;
        }
        if (live) {
          bool neg;
          const uint32_t d = recode_digit(rc, c, w, neg);
          if (d) o = (((((uint64_t)i * job.W + w)) << half_log) + (d - 1)) | (neg ? (1ull << 63) : 0ull);
        }
        if (++w == job.Wuse) {
          w = 0;
          pt++;
          fresh = true;
        }
      }
      offs[s_] = o;
    }
  }
  // pass 2: the additions, each table entry requested AHEAD items early (random one-line reads from gigabytes: HBM latency, not L2)
  auto fetch = [&](uint64_t o) {
    G1A r;
    const ZK_GLOBAL uint32_t* src = (const ZK_GLOBAL uint32_t*)&tab[o & ~(1ull << 63)];
#pragma unroll
    for (int l = 0; l < 12; l++) {
      r.x.v[l] = src[l];
      r.y.v[l] = src[12 + l];
    }
    return r;
  };
  G1A q[AHEAD];
#pragma unroll
  for (int a = 0; a < AHEAD; a++)
    if (a < (int)MSM_CHUNK && offs[a] != ~0ull) q[a] = fetch(offs[a]);
#pragma unroll
  for (uint32_t s_ = 0; s_ < MSM_CHUNK; s_++) {
    G1A P = q[s_ % AHEAD];
    const uint64_t o = offs[s_];
    if (s_ + AHEAD < MSM_CHUNK && offs[s_ + AHEAD] != ~0ull) q[s_ % AHEAD] = fetch(offs[s_ + AHEAD]);
    if (o != ~0ull) {
      if (o >> 63) P.y = fp_neg(P.y);
      acc = g1x_madd<INL>(acc, fq_unpack(P.x), fq_unpack(P.y));
    }
  }
  }  // sub
  partial[job.part_off + t] = acc;
}
// variants (ZKDL_FT_VARIANT, experiments): 0 = inlined products, 2 waves/SIMD, 1 entry ahead (default); 1 = two entries ahead; 2 = products as calls, 3 waves;
// 3 = products as calls, 4 waves
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_msmf_accum(const FullJob* __restrict__ jobs, uint32_t c, RecodeK K, G1X* __restrict__ partial) {
  msmf_accum_body<true, 1>(jobs, c, K, partial);
}
#ifdef ZKDL_EXPERIMENTS
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_msmf_accum_v1(const FullJob* __restrict__ jobs, uint32_t c, RecodeK K, G1X* __restrict__ partial) {
  msmf_accum_body<true, 2>(jobs, c, K, partial);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_msmf_accum_v2(const FullJob* __restrict__ jobs, uint32_t c, RecodeK K, G1X* __restrict__ partial) {
  msmf_accum_body<false, 1>(jobs, c, K, partial);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_msmf_accum_v3(const FullJob* __restrict__ jobs, uint32_t c, RecodeK K, G1X* __restrict__ partial) {
  msmf_accum_body<false, 2>(jobs, c, K, partial);
}
#endif
__global__ void __launch_bounds__(256) k_msmf_reduce(const FullJob* __restrict__ jobs, const G1X* __restrict__ partial) {
  __shared__ G1X sm[4];
  const FullJob job = jobs[blockIdx.x];
  const uint64_t items = (uint64_t)job.nnz * job.Wuse;
  const uint32_t np = (uint32_t)((items + MSM_CHUNK * FT_SUBS - 1) / (MSM_CHUNK * FT_SUBS));
  G1X acc = g1x_zero();
  for (uint32_t p = threadIdx.x; p < np; p += 256) acc = g1x_add(acc, partial[job.part_off + p]);
#pragma unroll 1
  for (int m = 1; m < 64; m <<= 1) acc = g1x_add(acc, g1x_shfl_xor(acc, m));
  if ((threadIdx.x & 63u) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    acc = g1x_add(g1x_add(sm[0], sm[1]), g1x_add(sm[2], sm[3]));
    *job.out = g1x_to_jacobian(acc);
  }
}

int gens_prepare_full(zk_ctx* ctx, zk_gens* g, uint32_t c) {
  if (c < 4 || c > 14) return ctx->fail(ZK_ERR_ARG, "zk_gens_prepare_full: window width 4 .. 14");
  if (g->full.pts && g->full.c == c) return ZK_OK;
  if (g->full.pts) {
    ZK_TRY(ctx->flush_g1_pending());  // deferred G1 phases (g1_batch > 1) hold pointers to the table about to be rebuilt
    ctx->wait_g1(ctx->stream);
    dev_free(ctx, g->full.pts);
    g->full = G1FullTable();
  }
  const uint32_t W = msm_num_windows(254, c), half = 1u << (c - 1);
  const G1Table* T = nullptr;
  ZK_TRY(gens_table(ctx, g, c, W, &T));
  const size_t rows = (size_t)g->n * W, total = rows * half;
  G1FullTable F;
  F.n = g->n;
  F.c = c;
  F.W = W;
  F.inf = T->inf;
  if (dev_alloc(ctx, (void**)&F.pts, total * sizeof(G1AE)) != hipSuccess) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: full digit table");
  {
    ZK_ALLOC(ctx, scratch, G1X, rows * FT_SEG);
    ZK_ALLOC(ctx, prefix, Fq, rows * FT_SEG);
    ZK_LAUNCH(ctx, "msm_fulltable_build", (uint64_t)sizeof(G1AE) * total, k_fulltable_build, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (const G1A*)T->pts, (const uint8_t*)T->inf,
              g->n, W, half, F.pts, scratch.get(), prefix.get());
  }
  g->full = F;
  return ZK_OK;
}

static int msm_run_full_jobs(zk_ctx* ctx, const std::vector<MsmJobHost>& hjobs) {
  const uint32_t nj = (uint32_t)hjobs.size();
  if (nj == 0) return ZK_OK;
  if (nj > 65535) return ctx->fail(ZK_ERR_ARG, "MSM: more than 65535 jobs in one batch");
  std::vector<FullJob> jobs(nj);
  uint64_t parts = 0, items_all = 0;
  uint32_t max_lanes = 0;
  const uint32_t c = hjobs[0].F->c, W = hjobs[0].F->W;
  for (uint32_t i = 0; i < nj; i++) {
    const MsmJobHost& h = hjobs[i];
    if (h.F->c != c || h.F->W != W) return ctx->fail(ZK_ERR_ARG, "MSM: full tables of different window widths in one batch");
    FullJob& j = jobs[i];
    j.tab = h.F->pts;
    j.inf = h.F->inf;
    j.mags = h.mags;
    j.out = h.out;
    j.n = h.F->n;
    j.W = W;
    j.pat_r = h.pat_r == 0xff ? 32u : h.pat_r;
    j.pat_par = h.pat_par;
    j.nnz = h.pat_r == 0xff ? j.n : j.n / 2;
    j.part_off = parts;
    j.Wuse = h.Wuse && h.Wuse < W ? h.Wuse : W;
    const uint64_t items = (uint64_t)j.nnz * j.Wuse;
    const uint32_t lanes = (uint32_t)((items + MSM_CHUNK * FT_SUBS - 1) / (MSM_CHUNK * FT_SUBS));
    parts += lanes;
    items_all += items;
    max_lanes = lanes > max_lanes ? lanes : max_lanes;
  }
  RecodeK K = make_recode_k(c, W);
  ZK_ALLOC(ctx, djobs, FullJob, nj);
  {
    const size_t bytes = (size_t)nj * sizeof(FullJob);
    void* st = bytes <= (1u << 20) ? ctx->stage_put(jobs.data(), bytes) : nullptr;
    if (st) ZK_TRY(dev_copy(ctx, djobs.get(), st, (bytes + 15) & ~(size_t)15));
    else ZK_TRY(host_to_dev(ctx, djobs.get(), jobs.data(), bytes));
  }
  ZK_ALLOC(ctx, partial, G1X, parts);
  const dim3 grid((max_lanes + 63) / 64, nj);
#ifdef ZKDL_EXPERIMENTS
  static const int variant = [] { const char* e = getenv("ZKDL_FT_VARIANT"); return e ? atoi(e) : 0; }();
  if (variant == 1) ZK_LAUNCH(ctx, "msm_table_accum", 100ull * items_all, k_msmf_accum_v1, grid, dim3(64), 0, (const FullJob*)djobs.get(), c, K, partial.get());
  else if (variant == 2) ZK_LAUNCH(ctx, "msm_table_accum", 100ull * items_all, k_msmf_accum_v2, grid, dim3(64), 0, (const FullJob*)djobs.get(), c, K, partial.get());
  else if (variant == 3) ZK_LAUNCH(ctx, "msm_table_accum", 100ull * items_all, k_msmf_accum_v3, grid, dim3(64), 0, (const FullJob*)djobs.get(), c, K, partial.get());
  else
#endif
  ZK_LAUNCH(ctx, "msm_table_accum", 100ull * items_all, k_msmf_accum, grid, dim3(64), 0, (const FullJob*)djobs.get(), c, K, partial.get());
  ZK_LAUNCH(ctx, "msm_table_sum", (uint64_t)sizeof(G1X) * parts, k_msmf_reduce, dim3(nj), dim3(256), 0, (const FullJob*)djobs.get(), (const G1X*)partial.get());
  return ZK_OK;
}

// run a list of MSM jobs that share the window width c (their tables were built with it)
static int msm_run_bucket_jobs(zk_ctx* ctx, const std::vector<MsmJobHost>& hjobs, uint32_t c);
int msm_run_jobs(zk_ctx* ctx, const std::vector<MsmJobHost>& all, uint32_t c) {
  bool any_full = false;
  for (const auto& h : all) any_full = any_full || h.F != nullptr;
  if (!any_full) return msm_run_bucket_jobs(ctx, all, c);
  std::vector<MsmJobHost> full, bucket;
  for (const auto& h : all) (h.F ? full : bucket).push_back(h);
  ZK_TRY(msm_run_full_jobs(ctx, full));
  return msm_run_bucket_jobs(ctx, bucket, c);
}
static int msm_run_bucket_jobs(zk_ctx* ctx, const std::vector<MsmJobHost>& hjobs, uint32_t c) {
  const uint32_t nj = (uint32_t)hjobs.size();
  if (nj == 0) return ZK_OK;
  const uint32_t B = 1u << (c - 1);
  std::vector<MsmJob> jobs(nj);
  uint64_t items = 0, chunks = 0, pairs = 0;
  uint32_t max_chunks = 0, Wmax = 0;
  // items per lane: MSM_CHUNK (16: the critical path of a proof-sized batch), doubled while the batch still fills the chip several times over.  A bucket
  // holds n W / 2^(c-1) items (all windows share the buckets: the tables carry 2^(c w)), 104-208 on 2048-4096-generator layers: at 16 per chunk a bucket is
  // 7-13 pieces for k_msm_combine to add (0.42 of the accumulation's time on the depth-64 model), at 64 two to four, half of which the wave stitches itself
  uint64_t all_items = 0;
  for (uint32_t i = 0; i < nj; i++) all_items += (uint64_t)hjobs[i].T->n * hjobs[i].W;
  uint32_t chunk = MSM_CHUNK;
  while (chunk < 64u && all_items / (2ull * chunk) >= (1ull << 20)) chunk *= 2;
  for (uint32_t i = 0; i < nj; i++) {
    const MsmJobHost& h = hjobs[i];
    if (h.W > h.T->W || h.T->c != c) return ctx->fail(ZK_ERR_ARG, "MSM: job/table window mismatch");
    MsmJob& j = jobs[i];
    j.table = h.T->pts;
    j.inf = h.T->inf;
    j.mags = h.mags;
    j.out = h.out;
    j.n = h.T->n;
    j.W = h.W;
    j.item_off = items;
    j.chunk_off = chunks;
    uint64_t ni = (uint64_t)j.n * j.W;
    uint32_t nc = (uint32_t)((ni + chunk - 1) / chunk);
    items += ni;
    chunks += nc;
    pairs += j.n;
    max_chunks = nc > max_chunks ? nc : max_chunks;
    Wmax = j.W > Wmax ? j.W : Wmax;
    if (j.n >= (1u << 20) || j.W >= (1u << 11)) return ctx->fail(ZK_ERR_ARG, "MSM: base set too large for the packed item format");
  }
  if (nj > 65535) return ctx->fail(ZK_ERR_ARG, "MSM: more than 65535 jobs in one batch");
  RecodeK K = make_recode_k(c, Wmax);
  ZK_ALLOC(ctx, djobs, MsmJob, nj);
  {
    // job table through the pinned staging ring (large batches: chunked synchronous copy)
    size_t bytes = (size_t)nj * sizeof(MsmJob);
    void* st = bytes <= (1u << 20) ? ctx->stage_put(jobs.data(), bytes) : nullptr;
    if (st) ZK_TRY(dev_copy(ctx, djobs.get(), st, bytes));
    else ZK_TRY(host_to_dev(ctx, djobs.get(), jobs.data(), bytes));
  }
  ZK_ALLOC(ctx, idx, uint32_t, items);
  ZK_ALLOC(ctx, off, uint32_t, (size_t)nj * (B + 2));
  ZK_ALLOC(ctx, buckets, G1X, (size_t)nj * B);
  ZK_ALLOC(ctx, head, G1X, chunks);
  ZK_ALLOC(ctx, tail, G1X, chunks);
  ZK_ALLOC(ctx, RC, G1X, (size_t)nj * reduce_shape(c).nsums);
  size_t lds = (size_t)(B + 2 + 1024) * 4;
  ZK_LAUNCH(ctx, "msm_sort", 32ull * pairs + 4ull * items, k_msm_sort, dim3(nj), dim3(1024), lds, (const MsmJob*)djobs.get(), c, K, idx.get(), off.get());
  ZK_LAUNCH(ctx, "msm_accum", 100ull * items, k_msm_accum, dim3((max_chunks + 63) / 64, nj), dim3(64), 0, (const MsmJob*)djobs.get(), c, chunk, (const uint32_t*)idx.get(),
            (const uint32_t*)off.get(), buckets.get(), head.get(), tail.get());
  {
    // Full-width scalars: the top window holds 256 - c*(Wmax-1) bits, of which a folded scalar (< 2^254) fills two
    // fewer; at <= 2 bits of entropy half of an MSM's points share a handful of buckets -> more lanes per bucket.
    // (Short-scalar batches -- commitments of quantised weights -- keep four lanes: their top window is partial too.)
    const bool full = c * Wmax >= 254u;
    const bool hot = !full || 256u - c * (Wmax - 1) <= 4u || chunks > 12ull * nj * B;  // or simply many chunks per bucket
    if (hot)
      ZK_LAUNCH(ctx, "msm_combine", (uint64_t)sizeof(G1X) * nj * B, k_msm_combine<4>, dim3((B + 15) / 16, nj), dim3(64), 0, (const MsmJob*)djobs.get(), c, chunk, (const uint32_t*)off.get(),
                buckets.get(), (const G1X*)head.get(), (const G1X*)tail.get());
    else
      ZK_LAUNCH(ctx, "msm_combine", (uint64_t)sizeof(G1X) * nj * B, k_msm_combine<2>, dim3((B + 31) / 32, nj), dim3(64), 0, (const MsmJob*)djobs.get(), c, chunk, (const uint32_t*)off.get(),
                buckets.get(), (const G1X*)head.get(), (const G1X*)tail.get());
  }
  const ReduceShape sh = reduce_shape(c);
  ZK_LAUNCH(ctx, "msm_bucket_reduce", (uint64_t)sizeof(G1X) * nj * B, k_msm_rowcol, dim3((sh.nsums * 4 + 63) / 64, nj), dim3(64), 0, (const G1X*)buckets.get(), c, sh, RC.get());
  ZK_LAUNCH(ctx, "msm_horner", (uint64_t)sizeof(G1X) * nj * sh.nsums, k_msm_weighted, dim3(nj), dim3(64), 0, (const MsmJob*)djobs.get(), (const G1X*)RC.get(), c, sh);
  return ZK_OK;
}

// prepare q scalar rows over the generator set g and append them as jobs (out[i] receives MSM i)
int msm_enqueue(zk_ctx* ctx, zk_gens* g, const Fr* scalars, uint32_t q, bool do_unmont, bool probe_bits, G1J* out, MsmQueue* queue, const uint32_t* out_idx,
                const uint16_t* pattern, uint32_t bits_bound) {
  const uint32_t n = g->n;
  if (q == 0) return ZK_OK;
  if (ctx->rec && !ctx->rec->chains.empty() && !ctx->rec->chains.back().early) queue->late_inputs = true;
  uint64_t total = (uint64_t)q * n;
  DBuf<Fr> mags;
  if (!mags.alloc(ctx, total)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: MSM scalars");
  ZK_ALLOC(ctx, maxbits, uint32_t, 1);
  ZK_TRY(dev_zero(ctx, maxbits.get(), 4));
  ZK_LAUNCHB(ctx, "msm_scalar_prep", 64ull * total, 256, k_scalar_prep, dim3(stream_grid(total)), 0, scalars, mags.get(), total, do_unmont ? 1 : 0, maxbits.get());
  uint32_t bits = 254;
  if (probe_bits) {  // setup-time paths only (commit): one host round trip buys 10x fewer windows
    ZK_TRY(dev_to_host(ctx, &bits, maxbits.get(), 4));
  }
  if (bits_bound && bits_bound < 254 && !probe_bits) {  // the caller states a bound (no host round trip): use it, and have the device check it
    bits = bits_bound;
    if (ctx->range_flag) ZK_LAUNCHB(ctx, "msm_scalar_prep", 8, 64, k_bits_guard, dim3(1), 0, (const uint32_t*)maxbits.get(), bits_bound, ctx->range_flag);
  }
  if (!probe_bits && ctx->full_tables && g->full.pts && !ctx->msm_window) {  // full-width scalars over a prepared generator set: the bucket-free path
    for (uint32_t i = 0; i < q; i++) {
      MsmJobHost h{nullptr, mags.get() + (uint64_t)i * n, out + (out_idx ? out_idx[i] : i), g->full.W};
      h.F = &g->full;
      h.Wuse = msm_num_windows(bits, g->full.c);   // short scalars: their low windows only
      if (pattern && pattern[i] != 0x1fe && n >= 2) {
        h.pat_r = (uint8_t)(pattern[i] >> 1);
        h.pat_par = (uint8_t)(pattern[i] & 1u);
      }
      queue->jobs.push_back(h);
    }
    queue->keep.push_back(std::move(mags));
    return ZK_OK;
  }
  uint32_t c = msm_pick_window(ctx, bits, n);
  if (!ctx->msm_window && bits > 24 && queue->c >= 10) c = queue->c;  // one window width per deferred batch: the first full-width job decides
  uint32_t W = msm_num_windows(bits, c);
  if (queue->c != 0 && queue->c != c) {  // one window width per batch
    if (ctx->stream != ctx->main_stream) return ctx->fail(ZK_ERR_ARG, "MSM: window width changed inside a deferred batch");
    ZK_TRY(msm_flush(ctx, queue));
  }
  queue->c = c;
  const G1Table* T = nullptr;
  ZK_TRY(gens_table(ctx, g, c, W, &T));
  for (uint32_t i = 0; i < q; i++) queue->jobs.push_back(MsmJobHost{T, mags.get() + (uint64_t)i * n, out + (out_idx ? out_idx[i] : i), W});
  queue->keep.push_back(std::move(mags));
  return ZK_OK;
}

int msm_flush(zk_ctx* ctx, MsmQueue* queue) {  // on the current stream; releases the prepared scalars
  if (queue->jobs.empty()) return ZK_OK;
  int rc = msm_run_jobs(ctx, queue->jobs, queue->c);
  if (rc == ZK_OK) rc = msm_run_derives(ctx, queue->derives);
  queue->jobs.clear();
  queue->derives.clear();
  queue->keep.clear();  // stream-ordered frees: safe behind the kernels just enqueued
  queue->c = 0;
  queue->late_inputs = false;
  return rc;
}

// immediate (non-deferred) batch: q MSMs over g, canonical results in out
int msm_batch_dev(zk_ctx* ctx, zk_gens* g, const Fr* scalars, uint32_t q, bool do_unmont, bool probe_bits, G1J* out) {
  MsmQueue queue;
  ZK_TRY(msm_enqueue(ctx, g, scalars, q, do_unmont, probe_bits, out, &queue));
  ZK_TRY(msm_flush(ctx, &queue));
  return g1_canonicalise_dev(ctx, out, q);
}

// Commitment::commit, intended row semantics: out[row] = sum_j gens[j] * unmont(t[row*n + j])
int commit_pippenger_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, uint32_t tn, G1J* out) {
  return msm_batch_dev(ctx, g, t, tn / g->n, /*do_unmont=*/true, /*probe_bits=*/true, out);
}

// ------------------------------------------------------------------ flattened opening
// C[(2^r - 1) + l] = prod_{t<r} (bit_t(l) ? 1 - u_t : u_t)   (Montgomery form), r = 0..k
__device__ __forceinline__ void k_mo_coef(uint32_t zk_nblk, const Fr* __restrict__ u, uint32_t k, Fr* __restrict__ C) {
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
// S holds the folded scalar tensors back to back: S_r at offset 2n - (2n >> r).
// compact = 1 (derived rounds): rows are [T_0][T0_0, T1_0][T0_1, T1_1] .. [final]: T_r, r >= 1, is not an MSM;
// compact = 2: T_0 is derived as well (from com(u_hi)): rows [T0_0, T1_0] .. [final]
__device__ __forceinline__ void k_mo_scalars(uint32_t zk_nblk, const Fr* __restrict__ C, const Fr* __restrict__ S, uint32_t n, uint32_t k, Fr* __restrict__ scal, uint32_t compact) {
  uint64_t total = (uint64_t)(k + 1) * n;
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)zk_nblk * blockDim.x) {
    uint32_t r = (uint32_t)(g / n), i = (uint32_t)(g - (uint64_t)r * n);
    if (r == k) {
      scal[(uint64_t)(compact == 2 ? 2 * k : compact ? 2 * k + 1 : 3 * k) * n + i] = fr_unmont(C[(1u << k) - 1u + i]);
      continue;
    }
    uint32_t j = i >> r, l = i & ((1u << r) - 1u);
    Fr cf = C[(1u << r) - 1u + l];
    const Fr* Sr = S + (2ull * n - ((2ull * n) >> r));
    Fr zero = fr_zero();
    // mont_mul(C~, s^) = C * s^ mod r as an integer: the MSM scalar of original generator i
    if (!compact || (r == 0 && compact == 1)) scal[(uint64_t)(3 * r + 0) * n + i] = fr_mul(cf, Sr[j]);
    const uint32_t r0 = compact == 2 ? 2 * r : compact ? 2 * r + 1 : 3 * r + 1;
    scal[(uint64_t)r0 * n + i] = (j & 1u) ? fr_mul(cf, Sr[j - 1]) : zero;        // T0 = sum g_{2j+1} * s_{2j}
    scal[(uint64_t)(r0 + 1) * n + i] = (j & 1u) ? zero : fr_mul(cf, Sr[j + 1]);  // T1 = sum g_{2j} * s_{2j+1}
  }
}

// ---- derived round commitments.  T_r is the commitment of round r's scalars under round r's generators, so the verifier's
// folding relation  T_{r+1} = a_r T_r + b_r T0_r + c_r T1_r  (a = u(1-u), b = (1-u)^2, c = u^2, u = unmont(challenge)) holds for the
// values the reference computes (zk_verify_fc checks exactly this).  Unrolled,
//     T_r = (prod_{i<r} a_i) T_0 + sum_{j<r} (prod_{j<i<r} a_i) (b_j T0_j + c_j T1_j),
// a combination of 2r + 1 points that the MSM batch produces anyway.  So only T_0, the T0_r / T1_r and the final generator run
// as MSMs (2k + 2 of the 3k + 1: the T_r are the FULL-size ones, 42 % of all (point, window) items of a proof), and T_1 .. T_{k-1}
// are k^2 - 1 scalar multiplications behind the batch: about 4 ms of latency on a handful of waves, i.e. a longer G1 phase under the
// following proofs' Fr phases, for a fifth less vector work per proof.  Same group elements as the reference's T_r.
// coef[r^2 + s], r0 <= r < k: s = 0 -> the base point; s = 1 + 2j -> T0_j; s = 2 + 2j -> T1_j (j < r); raw integers.
// Base point: T_0 (r0 = 1), or -- inside Commitment::open, where com(u_hi) is known -- com(u_hi) itself (r0 = 0): T_0 = [R] com(u_hi)
// (the opening multiplies by the Montgomery limbs of scalars whose VALUES commit to com(u_hi); zk_verify_fc checks the same), so the
// base coefficient is (prod a_i) * R mod r, which is the Montgomery form of the product as it stands, and T_0 is one more derived point.
__device__ __forceinline__ void k_mo_derive_coef(uint32_t zk_nblk, const Fr* __restrict__ u, uint32_t k, uint32_t r0, Fr* __restrict__ coef) {
  const uint32_t total = k * k;
  for (uint32_t idx = r0 + blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += zk_nblk * blockDim.x) {
    uint32_t r = 0;
    while ((r + 1) * (r + 1) <= idx) r++;
    const uint32_t s = idx - r * r;
    const Fr one = fr_one();
    Fr acc = one;
    uint32_t from = 0;  // product of a_i over from <= i < r
    if (s > 0) {
      const uint32_t j = (s - 1) >> 1;
      const Fr uj = u[j], nj = fr_sub(one, uj);
      acc = (s & 1u) ? fr_mul(nj, nj) : fr_mul(uj, uj);
      from = j + 1;
    }
    for (uint32_t i = from; i < r; i++) {
      const Fr ui = u[i];
      acc = fr_mul(acc, fr_mul(ui, fr_sub(one, ui)));
    }
    coef[idx] = (s == 0 && r0 == 0) ? acc : fr_unmont(acc);
  }
}
struct DeriveJob {
  G1J* proof;
  const G1J* base;   // the point the s = 0 coefficients multiply: proof[0] (T_0) or com(u_hi)
  const Fr* coef;
  uint32_t k, r0, lane_off;
};
// ---- scalar * point for the derived round commitments: ONE lane's latency per product, so the chain is what counts.
//  * GLV split.  G1 has the endomorphism phi(x, y) = (beta x, y) = lambda * (x, y) with lambda^2 + lambda + 1 = 0 mod r and
//    lambda = z^2 - 1 only 128 bits long (z the curve parameter), so k = k1 + k2 lambda with k2 = floor(k / lambda) and
//    k1 = k mod lambda both < 2^128 and  k P = k1 P + k2 phi(P):  two lanes run a 128-bit product each (of P and of phi(P), which
//    costs one field product) and the even lane adds the two halves -- half the doublings on the critical path.
//  * Signed 4-bit fixed window: s + sum_i 8 * 16^i has the nibbles (d_i + 8) with d_i in [-8, 7] and s = sum_i d_i 16^i (no carries
//    between digits: the same recoding as the bucket MSM's), so the lane keeps P .. 8P in a private slice of `tab` (global,
//    L2-resident) and runs 32 x (4 doublings + 1 addition).
//  * The group law runs on the 28-bit-limb field (g1q.hpp).
// k = k1 + k2 * lambda, 0 <= k1 < lambda, k2 <= lambda + 1 (k < r = lambda^2 + lambda + 1)
__device__ __forceinline__ void glv_split(const Fr& k, uint32_t (&k1)[4], uint32_t (&k2)[4]) {
  constexpr uint32_t LAM[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};               // lambda
  constexpr uint32_t MU[5] = {0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x1u};          // floor(2^256 / lambda)
  // q = (k * mu) >> 256: floor(k / lambda) or one less
  uint32_t prod[13];
#pragma unroll
  for (int i = 0; i < 13; i++) prod[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      uint64_t t = (uint64_t)k.v[i] * MU[j] + prod[i + j] + c;
      prod[i + j] = (uint32_t)t;
      c = t >> 32;
    }
    prod[i + 5] = (uint32_t)c;
  }
  uint32_t q[5];
#pragma unroll
  for (int i = 0; i < 5; i++) q[i] = prod[8 + i];
  // rem = k - q * lambda (< 2 lambda < 2^129): five limbs are enough
  uint32_t ql[5];
#pragma unroll
  for (int i = 0; i < 5; i++) ql[i] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i + j < 5) {
        uint64_t t = (uint64_t)q[i] * LAM[j] + ql[i + j] + c;
        ql[i + j] = (uint32_t)t;
        c = t >> 32;
      }
    }
    if (i == 0) ql[4] = (uint32_t)c;  // the only row whose carry still lands inside the five limbs
  }
  uint32_t rem[5], bw = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    uint32_t bo;
    rem[i] = __builtin_subc(k.v[i], ql[i], bw, &bo);
    bw = bo;
  }
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {  // at most one correction is ever needed (the estimate is off by <= 1); two are harmless
    uint32_t d[5], b2 = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      uint32_t bo;
      d[i] = __builtin_subc(rem[i], i < 4 ? LAM[i] : 0u, b2, &bo);
      b2 = bo;
    }
    if (b2 == 0) {  // rem >= lambda
#pragma unroll
      for (int i = 0; i < 5; i++) rem[i] = d[i];
      uint32_t c = 1;
#pragma unroll
      for (int i = 0; i < 5; i++) {
        uint32_t co;
        q[i] = __builtin_addc(q[i], 0u, c, &co);
        c = co;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    k1[i] = rem[i];
    k2[i] = q[i];
  }
}
__device__ __forceinline__ G1Q g1q_mul128_w4(const G1Q& P, const uint32_t (&k)[4], G1Q* __restrict__ tab) {
  // t = k + K, K = 0x888...8 (32 nibbles): 4 limbs + a carry digit
  uint32_t t[5];
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t co;
    t[i] = __builtin_addc(k[i], 0x88888888u, carry, &co);
    carry = co;
  }
  t[4] = carry;  // digit 32: 0 or 1 (its "+8" was not added: plain digit)
  tab[0] = P;
  G1Q acc = g1q_double(P);
  tab[1] = acc;
#pragma unroll 1
  for (int i = 2; i < 8; i++) {
    acc = g1q_add(acc, P);
    tab[i] = acc;
  }
  acc = t[4] ? P : g1q_zero();
#pragma unroll 1
  for (int w = 3; w >= 0; w--) {
    uint32_t limb = t[0];
#pragma unroll
    for (int i = 1; i < 4; i++) limb = (w == i) ? t[i] : limb;
#pragma unroll 1
    for (int nb = 7; nb >= 0; nb--) {
#pragma unroll 1
      for (int d4 = 0; d4 < 4; d4++) acc = g1q_double(acc);
      const int d = (int)((limb >> (4 * nb)) & 15u) - 8;
      if (d != 0) {
        G1Q q = tab[(d < 0 ? -d : d) - 1];
        if (d < 0) q = g1q_neg(q);
        acc = g1q_add(acc, q);
      }
    }
  }
  return acc;
}
// two lanes per (job, r, s): the halves k1 * P and k2 * phi(P) of coef * point; tmp[pair] = their sum
__global__ void __launch_bounds__(64) k_derive_mul(const DeriveJob* __restrict__ jobs, uint32_t nj, uint32_t total, G1Q* __restrict__ tmp, G1Q* __restrict__ tabs) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pair = lane >> 1, half = lane & 1u;
  const bool live = pair < total;
  G1Q acc = g1q_zero();
  if (live) {
    uint32_t q = 0;
    while (q + 1 < nj && jobs[q + 1].lane_off <= pair) q++;
    const DeriveJob job = jobs[q];
    const uint32_t idx = pair - job.lane_off + job.r0;
    uint32_t r = 0;
    while ((r + 1) * (r + 1) <= idx) r++;
    const uint32_t s = idx - r * r;
    const G1J P = s == 0 ? *job.base : job.proof[3u * ((s - 1) >> 1) + ((s & 1u) ? 1u : 2u)];
    uint32_t k1[4], k2[4];
    glv_split(job.coef[idx], k1, k2);
    G1Q Q = g1q_from_jacobian(P);
    if (half) {
      constexpr uint32_t beta[14] = {37886809u, 200165479u, 30609410u, 74842500u, 80498822u, 209116160u, 179210523u,
                                     180816594u, 266565572u, 239503032u, 99423592u, 150315392u, 2841631u, 53668u};  // beta * 2^392 mod p
      Fq bq;
#pragma unroll
      for (int i = 0; i < 14; i++) bq.v[i] = beta[i];
      Q.x = fq_mul(Q.x, bq);  // phi(X, Y, Z) = (beta X, Y, Z)
#pragma unroll
      for (int i = 0; i < 4; i++) k1[i] = k2[i];
    }
    acc = g1q_mul128_w4(Q, k1, tabs + (size_t)lane * 8);
  }
  G1Q other;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    other.x.v[i] = __shfl_xor(acc.x.v[i], 1, 64);
    other.y.v[i] = __shfl_xor(acc.y.v[i], 1, 64);
    other.z.v[i] = __shfl_xor(acc.z.v[i], 1, 64);
  }
  if (live && !half) tmp[pair] = g1q_add(acc, other);
}
// four lanes per (job, r): proof[3r] = sum of its 2r + 1 products, each lane adding every fourth one, then two butterfly steps
// (a single lane would chain up to 41 additions behind the 128-bit products)
__global__ void __launch_bounds__(64) k_derive_sum(const DeriveJob* __restrict__ jobs, uint32_t nj, uint32_t nseg, const uint32_t* __restrict__ seg_job, const G1Q* __restrict__ tmp) {
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = lane >> 2, sub = lane & 3u;
  const bool live = t < nseg;
  G1Q acc = g1q_zero();
  uint32_t q = 0, r = 0;
  if (live) {
    q = seg_job[2 * t];
    r = seg_job[2 * t + 1];
    const DeriveJob job = jobs[q];
    const G1Q* p = tmp + job.lane_off + (r * r - job.r0);
#pragma unroll 1
    for (uint32_t s = sub; s < 2 * r + 1; s += 4) acc = g1q_add(acc, p[s]);
  }
#pragma unroll 1
  for (int m = 1; m < 4; m <<= 1) {
    G1Q other;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      other.x.v[i] = __shfl_xor(acc.x.v[i], m, 64);
      other.y.v[i] = __shfl_xor(acc.y.v[i], m, 64);
      other.z.v[i] = __shfl_xor(acc.z.v[i], m, 64);
    }
    acc = g1q_add(acc, other);
  }
  if (live && sub == 0) jobs[q].proof[3 * r] = g1q_to_jacobian(acc);
}
int msm_run_derives(zk_ctx* ctx, const std::vector<DeriveJobHost>& derives) {
  if (derives.empty()) return ZK_OK;
  static const bool skip = experiment_on("ZKDL_EXPERIMENT_SKIP_DERIVE");  // timing experiments only: T_1.. stay unset
  if (skip) return ZK_OK;
  std::vector<DeriveJob> jobs;
  std::vector<uint32_t> seg;
  uint32_t total = 0;
  for (const auto& d : derives) {
    const uint32_t r0 = d.base ? 0u : 1u;
    if (d.k <= r0) continue;
    for (uint32_t r = r0; r < d.k; r++) {
      seg.push_back((uint32_t)jobs.size());
      seg.push_back(r);
    }
    jobs.push_back(DeriveJob{d.proof, d.base ? d.base : d.proof, d.coef, d.k, r0, total});
    total += d.k * d.k - r0;
  }
  if (jobs.empty()) return ZK_OK;
  const uint32_t nj = (uint32_t)jobs.size(), nseg = (uint32_t)seg.size() / 2;
  ZK_ALLOC(ctx, djobs, DeriveJob, nj);
  ZK_ALLOC(ctx, dseg, uint32_t, seg.size());
  ZK_ALLOC(ctx, tmp, G1Q, total);
  ZK_ALLOC(ctx, tabs, G1Q, (size_t)total * 16);
  {
    void* st = ctx->stage_put(jobs.data(), jobs.size() * sizeof(DeriveJob));
    if (!st) return ctx->fail(ZK_ERR_ALLOC, "pinned staging exhausted");
    ZK_TRY(dev_copy(ctx, djobs.get(), st, (jobs.size() * sizeof(DeriveJob) + 15) & ~(size_t)15));
    st = ctx->stage_put(seg.data(), seg.size() * 4);
    if (!st) return ctx->fail(ZK_ERR_ALLOC, "pinned staging exhausted");
    ZK_TRY(dev_copy(ctx, dseg.get(), st, (seg.size() * 4 + 15) & ~(size_t)15));
  }
  ZK_LAUNCH(ctx, "msm_derive", 320ull * total, k_derive_mul, dim3((2 * total + 63) / 64), dim3(64), 0, (const DeriveJob*)djobs.get(), nj, total, tmp.get(), tabs.get());
  ZK_LAUNCH(ctx, "msm_derive", 144ull * total, k_derive_sum, dim3((4 * nseg + 63) / 64), dim3(64), 0, (const DeriveJob*)djobs.get(), nj, nseg, (const uint32_t*)dseg.get(), (const G1Q*)tmp.get());
  return ZK_OK;
}

// all k folds of the opening's scalar vector in ONE workgroup (the vector lives in LDS): S_0 = t, S_{r+1}[j] = S_r[2j] + u_r (S_r[2j+1] - S_r[2j]);
// S_r is written to S + 2n - (2n >> r), and t(0) = S_k[0] to ret.  Replaces k launches of k_fr_fold on vectors of 2048 .. 1 elements: on the
// opening's stream those were ~0.6 ms of launch latency per proof for a few microseconds of work.
__device__ __forceinline__ void k_mo_fold_all(uint32_t zk_nblk, const Fr* __restrict__ t, const Fr* __restrict__ u, uint32_t n, uint32_t k, Fr* __restrict__ S, Fr* __restrict__ ret) {
  extern __shared__ Fr sm[];  // two buffers of n elements
  Fr* cur = sm;
  Fr* nxt = sm + n;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const Fr v = t[i];
    cur[i] = v;
    S[i] = v;
  }
  __syncthreads();
  uint32_t sz = n;
  for (uint32_t r = 0; r < k; r++) {
    const uint32_t osz = sz >> 1;
    const Fr x = u[r];
    Fr* dst = S + (2ull * n - ((2ull * n) >> (r + 1)));
    for (uint32_t j = threadIdx.x; j < osz; j += blockDim.x) {
      const Fr a0 = cur[2 * j], a1 = cur[2 * j + 1];
      const Fr v = fr_add(a0, fr_mul(x, fr_sub(a1, a0)));
      nxt[j] = v;
      dst[j] = v;
    }
    __syncthreads();
    Fr* tmp = cur;  // ping-pong between the two LDS buffers
    cur = nxt;
    nxt = tmp;
    sz = osz;
  }
  if (threadIdx.x == 0) *ret = cur[0];
}

int me_open_flat_dev(zk_ctx* ctx, zk_gens* g, const Fr* t, const Fr* u_d, uint32_t k, G1J* proof_d, Fr* ret_d, MsmQueue* queue, const G1J* com_eval_d) {
  const uint32_t n = g->n;
  if (k > 19 || n != (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions (flattened me_open needs n == 2^k)");
  // folded scalar tensors S_0 .. S_k (Fr only, tiny)
  ZK_ALLOC(ctx, S, Fr, 2ull * n + 2);
  if (n <= 2048 && k >= 1) {
    const size_t lds = (size_t)2 * n * sizeof(Fr);
    if (ctx->lds_first_use(zk::Kern<1024, &k_mo_fold_all>::desc())) {
      zk::Kern<1024, &k_mo_fold_all>::set_max_lds(128u << 10);
      (void)hipGetLastError();
    }
    ZK_LAUNCHB(ctx, "fr_fold", 32ull * 3 * n, 1024, k_mo_fold_all, dim3(1), lds, t, u_d, n, k, S.get(), ret_d);
  } else {
    ZK_TRY(dev_copy(ctx, S.get(), t, (size_t)n * sizeof(Fr)));
    for (uint32_t r = 0; r < k; r++) {
      Fr* cur = S.get() + (2ull * n - ((2ull * n) >> r));
      Fr* nxt = S.get() + (2ull * n - ((2ull * n) >> (r + 1)));
      ZK_TRY(fr_fold_dev(ctx, cur, nxt, u_d + r, n >> r, n >> (r + 1)));
    }
    ZK_TRY(dev_copy(ctx, ret_d, S.get() + (2ull * n - 2), sizeof(Fr)));  // t(0) = S_k[0]
  }
  uint32_t ncoef = (2u << k) - 1u;
  ZK_ALLOC(ctx, C, Fr, ncoef);
  ZK_LAUNCHB(ctx, "me_open_scalars", 32ull * ncoef, 256, k_mo_coef, dim3((ncoef + 255) / 256), 0, u_d, k, C.get());
  const bool derive = queue && ctx->derive_rounds && (k >= 2 || (com_eval_d && k >= 1));
  const bool from_com = derive && com_eval_d != nullptr;  // T_0 too: [R] com(u_hi)
  uint32_t q = from_com ? 2 * k + 1 : derive ? 2 * k + 2 : 3 * k + 1;
  ZK_ALLOC(ctx, scal, Fr, (size_t)q * n);
  ZK_LAUNCHB(ctx, "me_open_scalars", 32ull * q * n, 256, k_mo_scalars, dim3(stream_grid((uint64_t)(k + 1) * n)), 0, (const Fr*)C.get(), (const Fr*)S.get(), n, k,
            scal.get(), from_com ? 2u : derive ? 1u : 0u);
  if (derive) {
    // deferred batch: [T_0,] every T0_r / T1_r and the final generator as MSMs; the T_r derived behind the batch
    std::vector<uint32_t> rows;
    std::vector<uint16_t> pat;  // T0_r only has scalars where bit r of the generator index is set, T1_r where it is clear (k_mo_scalars)
    if (!from_com) {
      rows.push_back(0);
      pat.push_back(0x1fe);
    }
    for (uint32_t r = 0; r < k; r++) {
      rows.push_back(3 * r + 1);
      pat.push_back((uint16_t)((r << 1) | 1u));
      rows.push_back(3 * r + 2);
      pat.push_back((uint16_t)((r << 1) | 0u));
    }
    rows.push_back(3 * k);
    pat.push_back(0x1fe);
    ZK_TRY(msm_enqueue(ctx, g, scal.get(), q, /*do_unmont=*/false, /*probe_bits=*/false, proof_d, queue, rows.data(), pat.data()));
    DBuf<Fr> coef;
    if (!coef.alloc(ctx, (size_t)k * k)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: derive coefficients");
    ZK_LAUNCHB(ctx, "me_open_scalars", 32ull * k * k, 64, k_mo_derive_coef, dim3((k * k + 63) / 64), 0, u_d, k, from_com ? 0u : 1u, coef.get());
    queue->derives.push_back(DeriveJobHost{proof_d, coef.get(), k, from_com ? com_eval_d : nullptr});
    if (ctx->rec && !ctx->rec->chains.empty() && !ctx->rec->chains.back().early) queue->late_inputs = true;
    queue->keep.push_back(std::move(coef));
    return ZK_OK;
  }
  {
    std::vector<uint16_t> pat;  // rows 3r, 3r + 1, 3r + 2 = T_r (dense), T0_r, T1_r; row 3k the final generator
    for (uint32_t r = 0; r < k; r++) {
      pat.push_back(0x1fe);
      pat.push_back((uint16_t)((r << 1) | 1u));
      pat.push_back((uint16_t)((r << 1) | 0u));
    }
    pat.push_back(0x1fe);
    if (queue) return msm_enqueue(ctx, g, scal.get(), q, /*do_unmont=*/false, /*probe_bits=*/false, proof_d, queue, nullptr, pat.data());
    MsmQueue own;
    ZK_TRY(msm_enqueue(ctx, g, scal.get(), q, /*do_unmont=*/false, /*probe_bits=*/false, proof_d, &own, nullptr, pat.data()));
    ZK_TRY(msm_flush(ctx, &own));
    return g1_canonicalise_dev(ctx, proof_d, q);
  }
}

// com(u): sum_i eq(unmont u, i) * P_i over the (precomputed) commitment points -- the MSM form of
// G1TensorJacobian::operator()(u) (g1-tensor.cu:463-491)
__device__ __forceinline__ void k_eq_scalars(uint32_t zk_nblk, const Fr* __restrict__ u, uint32_t k, uint32_t m, Fr* __restrict__ out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  Fr one = fr_one(), acc = one;
  for (uint32_t b = 0; b < k; b++) {
    Fr ub = u[b];
    acc = fr_mul(acc, ((j >> b) & 1u) ? ub : fr_sub(one, ub));
  }
  out[j] = fr_unmont(acc);
}
int g1_me_msm_dev(zk_ctx* ctx, zk_gens* pts, const Fr* u_d, uint32_t k, G1J* dst_d, MsmQueue* queue) {
  const uint32_t m = pts->n;
  if (m == 0 || k > 31 || m > (1u << k)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (k > 0 && m <= (1u << (k - 1))) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  ZK_ALLOC(ctx, scal, Fr, m);
  ZK_LAUNCHB(ctx, "me_open_scalars", 32ull * m, 256, k_eq_scalars, dim3((m + 255) / 256), 0, u_d, k, m, scal.get());
  if (queue) return msm_enqueue(ctx, pts, scal.get(), 1, /*do_unmont=*/false, /*probe_bits=*/false, dst_d, queue);
  return msm_batch_dev(ctx, pts, scal.get(), 1, /*do_unmont=*/false, /*probe_bits=*/false, dst_d);
}

}  // namespace zk
