// zkdl_amd/csrc/relu_packed.hip -- the ReLU auxiliary proofs computed from PACKED bits.
//
// zkReLU stores its bit decompositions as one 32-byte field element per bit (zkrelu.cu:29-37) and
// binary_sumcheck streams those tensors -- 48*N elements, ~95 % of all Fr traffic of a proof
// (SURVEY.md App. B).  The same bits are kept packed by k_relu (one uint32 / uint16 per element), and
// the first three rounds of binary_sumcheck depend on the table only through small bit groups:
//
//   round 0: a pair is 2 bits (b0,b1):  c0 = 0, c2 = b0 xor b1, c1 = -c2
//   round 1: a pair is 4 bits: both operands take one of the 4 values {0, 1-v0, v0, 1}  -> 16-entry tables
//   round 2: a pair is 8 bits                                                           -> 256-entry tables
//   and the folded table after round 2 is V3[byte] (256 entries).
//
// The eq weight of pair h = (element g, position i') factors as Elo_r[i'] * Ehi[g] with Ehi = eq(u[log nb:], g)
// the SAME for all three rounds.  So one pass over the packed words computes, per element,
//   F_r,q(word) = sum_i' (Elo_r[i'] * T_r,q)[bit group i']        -- LDS lookups of position-prescaled tables + adds
// and accumulates Ehi[g] * F_r,q: 8 Fr products per element instead of 8 per PAIR x 28 pairs.  The pass also
// writes the round-3 table (N*nb/8 elements) for the generic rounds (sumcheck.hip).  HBM traffic of the three
// rounds drops from 32*(S + S/2 + ... ) bytes to nb/8 bytes read + 4*nb bytes written per element.
// FrTensor::partial_me(., nb) over the same tensor is likewise a masked sum of eq weights.
//
// Results are the same field elements (Fr is canonical): bit-identical to the reference's transcript.
#include "layers.hpp"

#include "dev_util.hpp"

namespace zk {

__device__ __forceinline__ Fr eq_prod(const Fr* __restrict__ u, uint32_t nvars, uint32_t j) {
  Fr one = fr_one(), acc = one;
  for (uint32_t b = 0; b < nvars; b++) {
    Fr ub = u[b];
    acc = fr_mul(acc, ((j >> b) & 1u) ? ub : fr_sub(one, ub));
  }
  return acc;
}
// value of 2^t consecutive bits `bits` after folding with v[0..t): the multilinear extension of the bits at v
__device__ __forceinline__ Fr fold_bits(uint32_t bits, uint32_t t, const Fr* __restrict__ v) {
  Fr acc = fr_zero();
  for (uint32_t i = 0; i < (1u << t); i++)
    if ((bits >> i) & 1u) acc = fr_add(acc, eq_prod(v, t, i));
  return acc;
}
__device__ __forceinline__ void bin_coefs(const Fr& a0, const Fr& a1, Fr& c0, Fr& c1, Fr& c2) {  // Fr_bin_sc_step, proof.cu:152-163
  Fr d = fr_sub(a1, a0);
  c0 = fr_sub(fr_sqr(a0), a0);
  c1 = fr_sub(fr_mul(fr_dbl(a0), d), d);
  c2 = fr_sqr(d);
}

// table layout (in Fr elements), P_r = nb >> (r+1) pairs per element in round r
// Compact tables: of a pair's three coefficients, c0 = a0^2 - a0 depends on the LOW half of the bit group only
// and p(1) = c0 + c1 + c2 = a1^2 - a1 on the HIGH half only.  So per position the prescaled tables are
//   C2[group] (16 / 256 entries), C0[low half] (4 / 16), P1[high half] (4 / 16)
// and c1 is recovered per element as sum P1 - sum C0 - sum C2: 1616 entries (51 KB) instead of 3728 (119 KB) for
// 32-bit words -- three workgroups of 256 lanes per CU instead of one of 512.
struct PackedLayout {
  uint32_t lognb, P0, P1, P2;
  uint32_t off_e0, off_t1, off_t2, off_v3, total;
};
static constexpr uint32_t T1_STRIDE = 16 + 4 + 4;      // per position: C2[16] | C0[4] | P1[4]
static constexpr uint32_t T2_STRIDE = 256 + 16 + 16;   // per position: C2[256] | C0[16] | P1[16]
static PackedLayout packed_layout(uint32_t nb) {
  PackedLayout L;
  L.lognb = nb == 32 ? 5 : 4;
  L.P0 = nb / 2; L.P1 = nb / 4; L.P2 = nb / 8;
  L.off_e0 = 0;
  L.off_t1 = 16;
  L.off_t2 = L.off_t1 + L.P1 * T1_STRIDE;
  L.off_v3 = L.off_t2 + L.P2 * T2_STRIDE;
  L.total = L.off_v3 + 256;
  return L;
}

// u = weight challenges, v = fold challenges (k each).  tab: PackedLayout; AH/BH: two-level Ehi tables.
__device__ __forceinline__ void k_binp_prep(uint32_t zk_nblk, const Fr* __restrict__ u, const Fr* __restrict__ v, uint32_t k, PackedLayout L, uint32_t ch, Fr* __restrict__ tab,
                                                    Fr* __restrict__ AH, Fr* __restrict__ BH) {
  const uint32_t nAH = 1u << ch, nBH = 1u << (k - L.lognb - ch);
  const uint32_t total = L.total + nAH + nBH;
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += zk_nblk * blockDim.x) {
    if (idx < L.off_t1) {  // Elo_0[i'] = eq(u[1..lognb), i')
      tab[idx] = idx < L.P0 ? eq_prod(u + 1, L.lognb - 1, idx) : fr_zero();
    } else if (idx < L.off_t2) {  // round 1, position i' = t / 24: weights eq(u[2..lognb), i')
      uint32_t t = idx - L.off_t1, ip = t / T1_STRIDE, e = t % T1_STRIDE;
      Fr wgt = eq_prod(u + 2, L.lognb - 2, ip);
      Fr val;
      if (e < 16) {  // C2[nibble]
        Fr a0 = fold_bits(e & 3u, 1, v), a1 = fold_bits(e >> 2, 1, v);
        Fr c0, c1, c2;
        bin_coefs(a0, a1, c0, c1, c2);
        val = c2;
      } else {       // C0[low 2 bits] / P1[high 2 bits]: x^2 - x of the half's folded value
        Fr x = fold_bits((e - 16) & 3u, 1, v);
        val = fr_sub(fr_sqr(x), x);
      }
      tab[idx] = fr_mul(wgt, val);
    } else if (idx < L.off_v3) {  // round 2, position i' = t / 288: weights eq(u[3..lognb), i')
      uint32_t t = idx - L.off_t2, ip = t / T2_STRIDE, e = t % T2_STRIDE;
      Fr wgt = eq_prod(u + 3, L.lognb - 3, ip);
      Fr val;
      if (e < 256) {  // C2[byte]
        Fr a0 = fold_bits(e & 15u, 2, v), a1 = fold_bits(e >> 4, 2, v);
        Fr c0, c1, c2;
        bin_coefs(a0, a1, c0, c1, c2);
        val = c2;
      } else {        // C0[low nibble] / P1[high nibble]
        Fr x = fold_bits((e - 256) & 15u, 2, v);
        val = fr_sub(fr_sqr(x), x);
      }
      tab[idx] = fr_mul(wgt, val);
    } else if (idx < L.total) {  // V3[byte]: the table after three folds
      tab[idx] = fold_bits(idx - L.off_v3, 3, v);
    } else if (idx < L.total + nAH) {
      AH[idx - L.total] = eq_prod(u + L.lognb, ch, idx - L.total);
    } else {
      BH[idx - L.total - nAH] = eq_prod(u + L.lognb + ch, k - L.lognb - ch, idx - L.total - nAH);
    }
  }
}

// Rounds 0..2 of binary_sumcheck from packed bits.  partial[blk*7 + {0: c2 of round 0; 1..3: round 1; 4..6: round 2}]
template <int NB, class WordT>
__device__ __forceinline__ void k_binp_rounds(uint32_t zk_nblk, const WordT* __restrict__ packed, uint32_t n, PackedLayout L, const Fr* __restrict__ tab_g, const Fr* __restrict__ AH,
                                                      const Fr* __restrict__ BH, uint32_t ch, Fr* __restrict__ a3, Fr* __restrict__ partial) {
  extern __shared__ Fr tab[];  // L.total entries + 16 reduction slots
  Fr* red = tab + L.total;
  for (uint32_t i = threadIdx.x; i < L.total; i += blockDim.x) tab[i] = tab_g[i];
  __syncthreads();
  const Fr* E0 = tab + L.off_e0;
  const Fr* T1 = tab + L.off_t1;
  const Fr* T2 = tab + L.off_t2;
  const Fr* V3 = tab + L.off_v3;
  const uint32_t mh = (1u << ch) - 1u;
  Fr acc[7];
#pragma unroll
  for (int i = 0; i < 7; i++) acc[i] = fr_zero();
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += zk_nblk * blockDim.x) {
    const uint32_t word = packed[g];
    const Fr w = fr_mul(AH[g & mh], BH[g >> ch]);
    Fr f[7];
#pragma unroll
    for (int i = 0; i < 7; i++) f[i] = fr_zero();
    const uint32_t x = word ^ (word >> 1);  // bit 2i' = b_{2i'} xor b_{2i'+1}
#pragma unroll 4
    for (uint32_t ip = 0; ip < NB / 2; ip++)
      if ((x >> (2 * ip)) & 1u) f[0] = fr_add(f[0], E0[ip]);
    Fr p1a = fr_zero(), p1b = fr_zero();  // sums of p(1) over the positions: c1 = p(1) - c0 - c2
#pragma unroll 2
    for (uint32_t ip = 0; ip < NB / 4; ip++) {
      const uint32_t nib = (word >> (4 * ip)) & 15u;
      const Fr* t = T1 + ip * T1_STRIDE;
      f[3] = fr_add(f[3], t[nib]);
      f[1] = fr_add(f[1], t[16 + (nib & 3u)]);
      p1a = fr_add(p1a, t[20 + (nib >> 2)]);
    }
    f[2] = fr_sub(fr_sub(p1a, f[1]), f[3]);
#pragma unroll
    for (uint32_t ip = 0; ip < NB / 8; ip++) {
      const uint32_t byte = (word >> (8 * ip)) & 255u;
      const Fr* t = T2 + ip * T2_STRIDE;
      f[6] = fr_add(f[6], t[byte]);
      f[4] = fr_add(f[4], t[256 + (byte & 15u)]);
      p1b = fr_add(p1b, t[272 + (byte >> 4)]);
      a3[(uint64_t)g * (NB / 8) + ip] = V3[byte];
    }
    f[5] = fr_sub(fr_sub(p1b, f[4]), f[6]);
#pragma unroll
    for (int i = 0; i < 7; i++) acc[i] = fr_add(acc[i], fr_mul(w, f[i]));
  }
  block_sum3_fr(acc[1], acc[2], acc[3], red);
  block_sum3_fr(acc[4], acc[5], acc[6], red);
  Fr s0 = block_sum_fr(acc[0], red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 7] = s0;
#pragma unroll
    for (int i = 1; i < 7; i++) partial[blockIdx.x * 7 + i] = acc[i];
  }
}
// proof[0..8] of the three packed rounds
__device__ __forceinline__ void k_binp_finish(uint32_t zk_nblk, const Fr* __restrict__ partial, uint32_t nblk, Fr* __restrict__ proof) {
  __shared__ Fr lds[4];
  for (uint32_t q = 0; q < 7; q++) {
    Fr acc = fr_zero();
    for (uint32_t i = threadIdx.x; i < nblk; i += blockDim.x) acc = fr_add(acc, partial[i * 7 + q]);
    acc = block_sum_fr(acc, lds);
    if (threadIdx.x == 0) {
      if (q == 0) {
        proof[0] = fr_zero();
        proof[1] = fr_neg(acc);
        proof[2] = acc;
      } else {
        proof[2 + q] = acc;
      }
    }
  }
}

// out[i] = sum_g eq(u_rec, g) * bit_i(word_g): FrTensor::partial_me(u_rec, nb) of the bit tensor.
// grid (blocks, nb/8): blockIdx.y selects a group of 8 bit positions.  partial[(blk*NB/8 + grp)*8 + j]
template <int NB, class WordT>
__device__ __forceinline__ void k_binp_partial_me(uint32_t zk_nblk, const WordT* __restrict__ packed, uint32_t n, const Fr* __restrict__ A, const Fr* __restrict__ B, uint32_t c,
                                                          Fr* __restrict__ partial) {
  __shared__ Fr lds[12];
  const uint32_t grp = blockIdx.y, mask = (1u << c) - 1u;
  Fr acc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) acc[j] = fr_zero();
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += zk_nblk * blockDim.x) {
    const uint32_t bits = ((uint32_t)packed[g] >> (8 * grp)) & 255u;
    if (!bits) continue;
    const Fr w = fr_mul(A[g & mask], B[g >> c]);
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((bits >> j) & 1u) acc[j] = fr_add(acc[j], w);
  }
  block_sum3_fr(acc[0], acc[1], acc[2], lds);
  block_sum3_fr(acc[3], acc[4], acc[5], lds);
  Fr z = fr_zero();
  block_sum3_fr(acc[6], acc[7], z, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 8; j++) partial[((uint64_t)blockIdx.x * (NB / 8) + grp) * 8 + j] = acc[j];
  }
}
__device__ __forceinline__ void k_binp_partial_me_finish(uint32_t zk_nblk, const Fr* __restrict__ partial, uint32_t nblk, uint32_t nb, Fr* __restrict__ out) {
  __shared__ Fr lds[4];
  for (uint32_t i = 0; i < nb; i++) {
    Fr acc = fr_zero();
    for (uint32_t b = threadIdx.x; b < nblk; b += blockDim.x) acc = fr_add(acc, partial[(uint64_t)b * nb + i]);
    acc = block_sum_fr(acc, lds);
    if (threadIdx.x == 0) out[i] = acc;
  }
}
__device__ __forceinline__ void k_eq_pair_tables(uint32_t zk_nblk, const Fr* __restrict__ u, uint32_t k, uint32_t c, Fr* __restrict__ A, Fr* __restrict__ B) {
  const uint32_t nA = 1u << c, nB = 1u << (k - c);
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nA + nB; idx += zk_nblk * blockDim.x) {
    if (idx < nA) A[idx] = eq_prod(u, c, idx);
    else B[idx - nA] = eq_prod(u + c, k - c, idx - nA);
  }
}

// binary_sumcheck(bit tensor of n elements x nb bits, u, v) with k = ceilLog2(n) + log2(nb) challenges each.
// packed: uint32 words (nb = 32) or uint16 words (nb = 16).  proof_d receives 3k+1 elements.
int binp_sumcheck_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d) {
  if (nb != 32 && nb != 16) return ctx->fail(ZK_ERR_ARG, "packed sumcheck: nb must be 16 or 32");
  PackedLayout L = packed_layout(nb);
  if (n == 0 || k != ceil_log2(n) + L.lognb) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t kh = k - L.lognb;
  const uint32_t ch = kh < 10 ? kh : 10;
  ZK_ALLOC(ctx, tab, Fr, L.total);
  ZK_ALLOC(ctx, AH, Fr, (size_t)1 << ch);
  ZK_ALLOC(ctx, BH, Fr, (size_t)1 << (kh - ch));
  const uint32_t nprep = L.total + (1u << ch) + (1u << (kh - ch));
  ZK_LAUNCHB(ctx, "binp_prep", 32ull * nprep, 256, k_binp_prep, dim3(stream_grid(nprep, 256, 64)), 0, u_d, v_d, k, L, ch, tab.get(), AH.get(), BH.get());
  const uint32_t a3n = n * (nb / 8);
  ZK_ALLOC(ctx, a3, Fr, a3n);
  const uint32_t grid = stream_grid(n, 256, 1536);
  ZK_ALLOC(ctx, partial, Fr, (size_t)grid * 7);
  const size_t lds = (size_t)(L.total + 16) * sizeof(Fr);
  uint64_t bytes = (uint64_t)n * (nb / 8) + 32ull * a3n;
  if (nb == 32) {
    if (ctx->lds_first_use(zk::Kern<256, &k_binp_rounds<32, uint32_t>>::desc())) {
      zk::Kern<256, &k_binp_rounds<32, uint32_t>>::set_max_lds(lds);
      (void)hipGetLastError();
    }
    ZK_LAUNCHB(ctx, "binp_rounds", bytes, 256, (k_binp_rounds<32, uint32_t>), dim3(grid), lds, (const uint32_t*)packed, n, L, (const Fr*)tab.get(), (const Fr*)AH.get(),
              (const Fr*)BH.get(), ch, a3.get(), partial.get());
  } else {
    if (ctx->lds_first_use(zk::Kern<256, &k_binp_rounds<16, uint16_t>>::desc())) {
      zk::Kern<256, &k_binp_rounds<16, uint16_t>>::set_max_lds(lds);
      (void)hipGetLastError();
    }
    ZK_LAUNCHB(ctx, "binp_rounds", bytes, 256, (k_binp_rounds<16, uint16_t>), dim3(grid), lds, (const uint16_t*)packed, n, L, (const Fr*)tab.get(), (const Fr*)AH.get(),
              (const Fr*)BH.get(), ch, a3.get(), partial.get());
  }
  ZK_LAUNCHB(ctx, "binp_finish", 32ull * 7 * grid, 256, k_binp_finish, dim3(1), 0, (const Fr*)partial.get(), grid, proof_d);
  // rounds 3 .. k-1 and the final a(0): generic path on the 8x smaller table
  return sumcheck_bin_dev(ctx, a3.get(), a3n, u_d + 3, v_d + 3, k - 3, proof_d + 9);
}

// FrTensor::partial_me(u, nb) of the bit tensor: nb outputs
int binp_partial_me_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, uint32_t k, Fr* out_d) {
  if (nb != 32 && nb != 16) return ctx->fail(ZK_ERR_ARG, "packed partial_me: nb must be 16 or 32");
  if (n == 0 || k != ceil_log2(n)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t c = k < 10 ? k : 10;
  ZK_ALLOC(ctx, A, Fr, (size_t)1 << c);
  ZK_ALLOC(ctx, B, Fr, (size_t)1 << (k - c));
  const uint32_t nt = (1u << c) + (1u << (k - c));
  ZK_LAUNCHB(ctx, "binp_prep", 32ull * nt, 256, k_eq_pair_tables, dim3(stream_grid(nt, 256, 64)), 0, u_d, k, c, A.get(), B.get());
  const uint32_t grid = stream_grid(n, 256, 64);
  ZK_ALLOC(ctx, partial, Fr, (size_t)grid * nb);
  if (nb == 32)
    ZK_LAUNCHB(ctx, "binp_partial_me", 4ull * n, 256, (k_binp_partial_me<32, uint32_t>), dim3(grid, 4), 0, (const uint32_t*)packed, n, (const Fr*)A.get(), (const Fr*)B.get(), c,
              partial.get());
  else
    ZK_LAUNCHB(ctx, "binp_partial_me", 2ull * n, 256, (k_binp_partial_me<16, uint16_t>), dim3(grid, 2), 0, (const uint16_t*)packed, n, (const Fr*)A.get(), (const Fr*)B.get(), c,
              partial.get());
  ZK_LAUNCHB(ctx, "binp_finish", 32ull * grid * nb, 256, k_binp_partial_me_finish, dim3(1), 0, (const Fr*)partial.get(), grid, nb, out_d);
  return ZK_OK;
}

}  // namespace zk
