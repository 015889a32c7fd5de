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
#include "sha256.hpp"

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

// ====================================================================================================
// Byte-histogram form of the packed rounds (default; the table form above stays as the A/B reference).
//
// Every coefficient sum of rounds 0..2 -- and c0, p(1) and the squares of round 3 (pairs of bytes) and of round 4
// (pairs of byte pairs, 32-bit words only) -- has the shape  sum_g w_g * sum_p T_p[byte_p(word_g)]  with
// w_g = eq(u[log nb:], g): a table lookup per byte position, weighted per word.  Swapping the sums,
//     sum_g w_g T_p[byte_p(g)] = sum_e T_p[e] * H[p][e],     H[p][e] = sum_{g : byte_p(g) = e} w_g,
// so ONE weighted histogram over (byte position, byte value) serves all of them, and the per-word work is one
// product (w_g) plus integer adds: the Montgomery residue of w_g is cut into five 51-bit limbs and added into
// 64-bit LDS counters with ds_add_u64 (a residue sum is congruent to the residue of the sum; 8192 words per
// workgroup keep every counter below 2^64).  Zero bytes are skipped: every table vanishes at 0.
// What is not a function of single bytes are the cross terms of the squares,
//     X_j(g) = V3[byte_2j(g)] * V3[byte_2j+1(g)]                  (round 3: (a1 - a0)^2; round 4: a0^2, a1^2)
//     A(g)   = a0(g) * a1(g)                                      (round 4: (a1 - a0)^2)
// X_j goes through a second histogram keyed by the LOW byte, H'_j[e] = sum_g [byte_2j = e] w_g V3[byte_2j+1]
// (one product per pair, then sum_e V3[e] H'_j[e]); A is accumulated directly (two products per word).
// Per 32-bit word: 6 products instead of 7 + 6 * 3.5 (three generic rounds), and the first table written to HBM
// has ONE element per word (S/32) instead of four (S/8).  16-bit words: 2 products, table of S/16.
// Each workgroup stores its counters into a slab of its own; k_binb_reduce adds the slabs (64-bit sums with a carry word),
// k_binb_finish normalises each bin to a field element and contracts it with the per-bin factors k_binb_prep tabulated.
struct BytesLayout {
  uint32_t lognb, npos, R;      // byte positions per word; rounds handled here (4 for 16-bit words, 5 for 32-bit)
  uint32_t nH, nHP, NF;         // histogram bins (npos*256), cross-term bins ((npos/2)*256), factors per bin
};
static BytesLayout bytes_layout(uint32_t nb) {
  BytesLayout L;
  L.lognb = nb == 32 ? 5 : 4;
  L.npos = nb / 8;
  L.R = nb == 32 ? 5 : 4;
  L.nH = L.npos * 256;
  L.nHP = (L.npos / 2) * 256;
  L.NF = nb == 32 ? 13 : 10;
  return L;
}
static constexpr uint32_t BINB_LIMBS_LDS = 5;   // 51-bit limbs in LDS
#ifndef ZK_BINB_WORDS_PER_WG
#define ZK_BINB_WORDS_PER_WG 8192
#endif
// words per workgroup = additions a counter can take: <= 8192 (13 bits of headroom above the 51-bit limbs: 8192 (2^51 - 1) < 2^64).  8192 instead of 4096
// halves the slabs a pass flushes and k_binb_reduce adds: same-box 2.94 -> 2.84 ms per proof (K = 60), steady state 2.88 -> 2.76 (round 4)
static constexpr uint32_t BINB_WORDS_PER_WG = ZK_BINB_WORDS_PER_WG;

// all NF factors of histogram bin (p, e) -- k_binb_finish says how the sums combine into the round coefficients:
//   f[0]          round 0: sum over the byte's four bit pairs of Elo_0 * (b0 xor b1)                (c2; c0 = 0, c1 = -c2)
//   f[1..3]       round 1: sum over the two nibbles of Elo_1 * {a0^2 - a0, a1^2 - a1, (a1 - a0)^2}   (c0, p(1), c2)
//   f[4..6]       round 2: Elo_2[p] * the same three of the byte's nibble pair
//   f[7..9]       round 3: bytes (2j, 2j+1) pair up: Elo_3[j] * {[p even] (V3^2 - V3), [p odd] (V3^2 - V3), V3^2}
//   f[10..12]     round 4 (32-bit words): a0 = (1 - v3) V3[b0] + v3 V3[b1], a1 likewise from bytes 2, 3; with c = 1 - v3 or v3:
//                 {[p < 2] (c^2 V3^2 - c V3), [p >= 2] the same, c^2 V3^2}   (c0, p(1) and the squares, all without cross terms)
// elo: this job's position weights in LDS: Elo_0[16] | Elo_1[8] | Elo_2[4] | Elo_3[2]; eqv: eq(v[0..3), i) for i < 8
__device__ __forceinline__ void binb_factors(uint32_t p, uint32_t e, const Fr* __restrict__ elo, const Fr* __restrict__ eqv, const Fr& v0, const Fr& v1, const Fr& v3,
                                                     uint32_t NF, Fr* __restrict__ f) {
  const Fr zero = fr_zero(), one = fr_one();
  {  // round 0
    Fr acc = zero;
    const uint32_t x = e ^ (e >> 1);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
      if ((x >> (2 * i)) & 1u) acc = fr_add(acc, elo[4 * p + i]);
    f[0] = acc;
  }
  {  // round 1: a pair of bit pairs; a bit pair (x0, x1) folds to x0 (1 - v0) + x1 v0
    Fr c0 = zero, p1 = zero, c2 = zero;
    const Fr nv0 = fr_sub(one, v0);
    for (uint32_t i = 0; i < 2; i++) {
      const uint32_t nib = (e >> (4 * i)) & 15u;
      if (!nib) continue;
      auto fold1 = [&](uint32_t bits) { return bits == 0 ? zero : bits == 1 ? nv0 : bits == 2 ? v0 : one; };
      const Fr a0 = fold1(nib & 3u), a1 = fold1(nib >> 2), w = elo[16 + 2 * p + i], d = fr_sub(a1, a0);
      c0 = fr_add(c0, fr_mul(w, fr_sub(fr_sqr(a0), a0)));
      p1 = fr_add(p1, fr_mul(w, fr_sub(fr_sqr(a1), a1)));
      c2 = fr_add(c2, fr_mul(w, fr_sqr(d)));
    }
    f[1] = c0; f[2] = p1; f[3] = c2;
  }
  // folds of the nibbles with (v0, v1) and of the byte with (v0, v1, v2): sums of eq weights over the set bits
  Fr lo4 = zero, hi4 = zero, t = zero;
  {
    const Fr nv1 = fr_sub(one, v1);
    const Fr e4[4] = {fr_mul(fr_sub(one, v0), nv1), fr_mul(v0, nv1), fr_mul(fr_sub(one, v0), v1), fr_mul(v0, v1)};
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      if ((e >> i) & 1u) lo4 = fr_add(lo4, e4[i]);
      if ((e >> (4 + i)) & 1u) hi4 = fr_add(hi4, e4[i]);
    }
#pragma unroll
    for (uint32_t i = 0; i < 8; i++)
      if ((e >> i) & 1u) t = fr_add(t, eqv[i]);
  }
  {  // round 2
    const Fr w = elo[24 + p], d = fr_sub(hi4, lo4);
    f[4] = fr_mul(w, fr_sub(fr_sqr(lo4), lo4));
    f[5] = fr_mul(w, fr_sub(fr_sqr(hi4), hi4));
    f[6] = fr_mul(w, fr_sqr(d));
  }
  const Fr sq = fr_sqr(t);
  {  // round 3
    const Fr w = elo[28 + (p >> 1)];
    const Fr q = fr_mul(w, fr_sub(sq, t));
    f[7] = (p & 1u) ? zero : q;
    f[8] = (p & 1u) ? q : zero;
    f[9] = fr_mul(w, sq);
  }
  if (NF > 10) {  // round 4
    const Fr c = (p & 1u) ? v3 : fr_sub(one, v3);
    const Fr c2sq = fr_mul(fr_sqr(c), sq);
    const Fr q = fr_sub(c2sq, fr_mul(c, t));
    f[10] = p < 2 ? q : zero;
    f[11] = p >= 2 ? q : zero;
    f[12] = c2sq;
  }
}

// tables for k_binb_words ([V3 | VA = (1 - v3) V3 | VB = v3 V3], 768 entries), the two-level word weights and the per-bin
// factors -- one launch; workgroups [0, nH / 256) do the factors (one bin per lane), the rest everything else
__device__ __forceinline__ void k_binb_prep(uint32_t zk_nblk, const Fr* __restrict__ u, const Fr* __restrict__ v, uint32_t k, BytesLayout L, uint32_t ch, Fr* __restrict__ tab,
                                                    Fr* __restrict__ AH, Fr* __restrict__ BH, Fr* __restrict__ fac, const Fr* __restrict__ u_rec, Fr* __restrict__ A2,
                                                    Fr* __restrict__ B2) {
  __shared__ Fr elo[30 + 8];
  const uint32_t nfb = L.nH / 256;  // factor workgroups (blockDim.x = 256)
  if (threadIdx.x < 38) {           // every workgroup needs eqv; the factor workgroups also the position weights
    const uint32_t i = threadIdx.x;
    Fr r;
    if (i < 16) r = i < L.npos * 4 ? eq_prod(u + 1, L.lognb - 1, i) : fr_zero();
    else if (i < 24) r = (i - 16) < L.npos * 2 ? eq_prod(u + 2, L.lognb - 2, i - 16) : fr_zero();
    else if (i < 28) r = (i - 24) < L.npos ? eq_prod(u + 3, L.lognb - 3, i - 24) : fr_zero();
    else if (i < 30) r = (i - 28) < (L.npos + 1) / 2 ? eq_prod(u + 4, L.lognb - 4, i - 28) : fr_zero();
    else r = eq_prod(v, 3, i - 30);
    elo[i] = r;
  }
  __syncthreads();
  if (blockIdx.x < nfb) {
    const uint32_t bin = blockIdx.x * 256 + threadIdx.x;
    Fr f[13];
    binb_factors(bin >> 8, bin & 255u, elo, elo + 30, v[0], v[1], v[3], L.NF, f);
    for (uint32_t q = 0; q < L.NF; q++) fac[(size_t)bin * L.NF + q] = f[q];
    return;
  }
  const uint32_t nAH = 1u << ch, nBH = 1u << (k - L.lognb - ch);
  // (u_rec != nullptr: the two-level eq(u_rec, .) tables of the partial_me histogram as well -- they were a launch of their own at the head of the chain)
  const uint32_t o1 = 768, o2 = o1 + nAH, o3 = o2 + nBH, o4 = u_rec ? o3 + nAH : o3, total = u_rec ? o4 + nBH : o3;
  for (uint32_t idx = (blockIdx.x - nfb) * blockDim.x + threadIdx.x; idx < total; idx += (zk_nblk - nfb) * blockDim.x) {
    if (idx < o1) {
      const uint32_t e = idx & 255u, which = idx >> 8;
      Fr t = fr_zero();
#pragma unroll
      for (uint32_t i = 0; i < 8; i++)
        if ((e >> i) & 1u) t = fr_add(t, elo[30 + i]);
      tab[idx] = which == 0 ? t : which == 1 ? fr_mul(fr_sub(fr_one(), v[3]), t) : fr_mul(v[3], t);
    } else if (idx < o2) {
      AH[idx - o1] = eq_prod(u + L.lognb, ch, idx - o1);
    } else if (idx < o3) {
      BH[idx - o2] = eq_prod(u + L.lognb + ch, k - L.lognb - ch, idx - o2);
    } else if (idx < o4) {
      A2[idx - o3] = eq_prod(u_rec, ch, idx - o3);
    } else {
      B2[idx - o4] = eq_prod(u_rec + ch, k - L.lognb - ch, idx - o4);
    }
  }
}

// five 51-bit limbs of a 255-bit residue
__device__ __forceinline__ void binb_limbs51(const Fr& x, unsigned long long* l) {
  const unsigned long long q0 = (unsigned long long)x.v[0] | ((unsigned long long)x.v[1] << 32), q1 = (unsigned long long)x.v[2] | ((unsigned long long)x.v[3] << 32),
                           q2 = (unsigned long long)x.v[4] | ((unsigned long long)x.v[5] << 32), q3 = (unsigned long long)x.v[6] | ((unsigned long long)x.v[7] << 32);
  const unsigned long long M = (1ull << 51) - 1ull;
  l[0] = q0 & M;
  l[1] = ((q0 >> 51) | (q1 << 13)) & M;
  l[2] = ((q1 >> 38) | (q2 << 26)) & M;
  l[3] = ((q2 >> 25) | (q3 << 39)) & M;
  l[4] = q3 >> 12;
}
__device__ __forceinline__ void binb_hist_add(unsigned long long* __restrict__ bin, const Fr& x) {
  unsigned long long l[5];
  binb_limbs51(x, l);
#pragma unroll
  for (int j = 0; j < 5; j++) atomicAdd(bin + j, l[j]);   // ds_add_u64, result unused
}

// The pass over the packed words.  out[g] = the table entry of word g after R folds (all rounds handled here);
// partialA[blk] = this workgroup's sum of w a0 a1 (32-bit words); slabs[blk] = this workgroup's histogram counters.
// A2 / B2 (two-level eq(u_rec, g) tables, split c2) add a third histogram H2[p][e] = sum_{g : byte_p(g) = e} eq(u_rec, g): the
// multilinear evaluations of the nb bit planes (FrTensor::partial_me(u_rec, nb), zkrelu.cu:88,93) are its bit marginals, so
// the same pass over the words serves the partial_me that follows each bit sumcheck.
template <int NB, class WordT>
__device__ __forceinline__ void k_binb_words(uint32_t zk_nblk, const WordT* __restrict__ packed, uint32_t n, const Fr* __restrict__ tab_g, const Fr* __restrict__ AH,
                                                     const Fr* __restrict__ BH, uint32_t ch, const Fr* __restrict__ v, Fr* __restrict__ out, Fr* __restrict__ partialA,
                                                     unsigned long long* __restrict__ slabs, const Fr* __restrict__ A2, const Fr* __restrict__ B2, uint32_t c2) {
  constexpr uint32_t NPOS = NB / 8, NH = NPOS * 256, NHP = (NPOS / 2) * 256, NBINS = NH + NHP + NH;
  extern __shared__ unsigned long long lds64[];
  unsigned long long* H = lds64;                                   // NBINS * 5 counters
  Fr* tab = reinterpret_cast<Fr*>(lds64 + NBINS * BINB_LIMBS_LDS);   // 768 table entries (16-byte aligned: NBINS * 40 B)
  Fr* red = tab + 768;                                             // 16 reduction slots
  for (uint32_t i = threadIdx.x; i < NBINS * BINB_LIMBS_LDS; i += blockDim.x) H[i] = 0ull;
  for (uint32_t i = threadIdx.x; i < 768; i += blockDim.x) tab[i] = tab_g[i];
  __syncthreads();
  const Fr* V3 = tab;
  const Fr* VA = tab + 256;
  const Fr* VB = tab + 512;
  const uint32_t mh = (1u << ch) - 1u, m2 = (1u << c2) - 1u;
  Fr accA = fr_zero();
  Fr v4 = fr_zero();
  if (NB == 32) v4 = v[4];
  // a contiguous range of words per workgroup: at most BINB_WORDS_PER_WG (the host sizes the grid), so that no counter overflows
  const uint32_t per = (n + zk_nblk - 1) / zk_nblk;
  const uint32_t g_end = min(n, (blockIdx.x + 1) * per);
  for (uint32_t g = blockIdx.x * per + threadIdx.x; g < g_end; g += blockDim.x) {
    const uint32_t word = packed[g];
    if (word == 0u) {
      out[g] = fr_zero();
      continue;
    }
    const Fr w = fr_mul(AH[g & mh], BH[g >> ch]);
    {
      unsigned long long l[5];
      binb_limbs51(w, l);
#pragma unroll
      for (uint32_t p = 0; p < NPOS; p++) {
        const uint32_t b = (word >> (8 * p)) & 255u;
        if (b) {
          unsigned long long* bin = H + (p * 256 + b) * BINB_LIMBS_LDS;
#pragma unroll
          for (int j = 0; j < 5; j++) atomicAdd(bin + j, l[j]);
        }
      }
    }
    {
      const Fr w2 = fr_mul(A2[g & m2], B2[g >> c2]);
      unsigned long long l[5];
      binb_limbs51(w2, l);
#pragma unroll
      for (uint32_t p = 0; p < NPOS; p++) {
        const uint32_t b = (word >> (8 * p)) & 255u;
        if (b) {
          unsigned long long* bin = H + (NH + NHP + p * 256 + b) * BINB_LIMBS_LDS;
#pragma unroll
          for (int j = 0; j < 5; j++) atomicAdd(bin + j, l[j]);
        }
      }
    }
    const uint32_t b0 = word & 255u, b1 = (word >> 8) & 255u;
    if (b0 && b1) binb_hist_add(H + (NH + b0) * BINB_LIMBS_LDS, fr_mul(w, V3[b1]));
    const Fr a0 = fr_add(VA[b0], VB[b1]);
    if (NB == 32) {
      const uint32_t b2 = (word >> 16) & 255u, b3 = word >> 24;
      if (b2 | b3) {
        if (b2 && b3) binb_hist_add(H + (NH + 256 + b2) * BINB_LIMBS_LDS, fr_mul(w, V3[b3]));
        const Fr a1 = fr_add(VA[b2], VB[b3]);
        if (b0 | b1) accA = fr_add(accA, fr_mul(fr_mul(w, a0), a1));
        out[g] = fr_add(a0, fr_mul(v4, fr_sub(a1, a0)));
      } else {
        out[g] = fr_sub(a0, fr_mul(v4, a0));   // a1 = 0
      }
    } else {
      out[g] = a0;
    }
  }
  if (NB == 32) {
    accA = block_sum_fr(accA, red);
    if (threadIdx.x == 0) partialA[blockIdx.x] = accA;
  }
  __syncthreads();
  // flush: this workgroup's counters, as they are, into its own slab (plain coalesced stores; k_binb_reduce adds the slabs)
  unsigned long long* dst = slabs + (size_t)blockIdx.x * NBINS * BINB_LIMBS_LDS;
  for (uint32_t i = threadIdx.x; i < NBINS * BINB_LIMBS_LDS; i += blockDim.x) dst[i] = H[i];
}

// sums[i] = (low 64 bits, carries) of sum over the slabs of counter i.  Eight lanes share a counter (each adds every eighth slab, three shuffle steps join
// them): the launch is a chain of dependent-latency loads on the bit sumchecks' streams (one lane per counter walked all 56 slabs of a 453 888-word
// tensor: 65 us), not bandwidth (6 MB).
static constexpr uint32_t BINB_REDUCE_LANES = 8;
__device__ __forceinline__ void k_binb_reduce(uint32_t zk_nblk, const unsigned long long* __restrict__ slabs, uint32_t nslab, uint32_t ncount, unsigned long long* __restrict__ sums) {
  const uint32_t sub = threadIdx.x % BINB_REDUCE_LANES, per_blk = blockDim.x / BINB_REDUCE_LANES;
  for (uint32_t base = blockIdx.x * per_blk; base < ncount; base += zk_nblk * per_blk) {  // uniform over the workgroup: every lane takes part in the shuffles
    const uint32_t i = base + threadIdx.x / BINB_REDUCE_LANES;
    unsigned long long lo = 0, hi = 0;
    if (i < ncount) {
      for (uint32_t sl = sub; sl < nslab; sl += BINB_REDUCE_LANES) {
        const unsigned long long x = slabs[(size_t)sl * ncount + i];
        lo += x;
        hi += lo < x ? 1ull : 0ull;
      }
    }
#pragma unroll
    for (uint32_t m = 1; m < BINB_REDUCE_LANES; m <<= 1) {
      const unsigned long long olo = __shfl_xor(lo, m, 64), ohi = __shfl_xor(hi, m, 64);
      lo += olo;
      hi += ohi + (lo < olo ? 1ull : 0ull);
    }
    if (i < ncount && sub == 0) {
      sums[2 * (size_t)i] = lo;
      sums[2 * (size_t)i + 1] = hi;
    }
  }
}

// sum_j (lo_j + 2^64 hi_j) 2^(51 j)  (five (lo, hi) pairs, hi_j < 2^32)  ->  field element
__device__ __forceinline__ Fr binb_bin_to_fr(const unsigned long long* __restrict__ pairs) {
  uint32_t x[11];
#pragma unroll
  for (int i = 0; i < 11; i++) x[i] = 0;
#pragma unroll
  for (int j = 0; j < (int)BINB_LIMBS_LDS; j++) {
    const unsigned long long lo = pairs[2 * j], hi = pairs[2 * j + 1];
    // the 96-bit value hi:lo shifted left by 51 j bits, added into x
    const int bit = 51 * j, wi = bit >> 5, sh = bit & 31;
    const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32), w2 = (uint32_t)hi;
    uint32_t add[4];
    add[0] = w0 << sh;
    add[1] = sh ? ((w1 << sh) | (w0 >> (32 - sh))) : w1;
    add[2] = sh ? ((w2 << sh) | (w1 >> (32 - sh))) : w2;
    add[3] = sh ? (w2 >> (32 - sh)) : 0u;
    uint32_t cy = 0;
#pragma unroll
    for (int t = 0; t < 11; t++) {
      if (t < wi) continue;
      const uint32_t a = (t - wi) < 4 ? add[t - wi] : 0u;
      uint32_t co;
      x[t] = __builtin_addc(x[t], a, cy, &co);
      cy = co;
    }
  }
  // x < 2^(204 + 96): lo (256 bits) + hi * 2^256 with hi < 2^44:  lo mod r by conditional subtraction (2^256 < 3 r), hi * R mod r = mont(hi)
  Fr lo, hi = fr_zero();
#pragma unroll
  for (int i = 0; i < 8; i++) lo.v[i] = x[i];
  hi.v[0] = x[8];
  hi.v[1] = x[9];
  lo = f_reduce_once<FrP>(f_reduce_once<FrP>(lo));
  if ((x[8] | x[9]) == 0u) return lo;
  return fr_add(lo, fr_mont(hi));
}

// One workgroup of 1024 lanes (one histogram bin per lane): normalise every bin, contract with the factors, combine into
// proof[0 .. 3R).
//   S_q = sum_bins fac[bin][q] * H[bin]      SX_j = sum_e V3[e] * H'_j[e]      A = sum of the workgroups' partialA
//   round 0: (0, -S_0, S_0)      round 1: c0 = S_1, p1 = S_2, c2 = S_3      round 2: S_4, S_5, S_6
//   round 3: c0 = S_7, p1 = S_8, c2 = S_9 - 2 sum_j eq(u[4..lognb), j) SX_j
//   round 4: c0 = S_10 + m SX_0, p1 = S_11 + m SX_1, c2 = S_12 + m (SX_0 + SX_1) - 2 A,   m = 2 (1 - v3) v3
//   and c1 = p1 - c0 - c2 everywhere
__device__ __forceinline__ void k_binb_finish(uint32_t zk_nblk, const unsigned long long* __restrict__ sums, const Fr* __restrict__ fac, const Fr* __restrict__ tab,
                                                      const Fr* __restrict__ partialA, uint32_t nA, BytesLayout L, const Fr* __restrict__ u, const Fr* __restrict__ v,
                                                      Fr* __restrict__ proof, Fr* __restrict__ pme_out) {
  __shared__ Fr red[48];
  __shared__ Fr S[16];
  const uint32_t tid = threadIdx.x;
  const bool own = tid < L.nH;      // nH <= blockDim.x = 1024
  Fr h = fr_zero();
  if (own) h = binb_bin_to_fr(sums + (size_t)tid * BINB_LIMBS_LDS * 2);
  const bool live = own && !l_is_zero(h);
  for (uint32_t q0 = 0; q0 < L.NF; q0 += 3) {
    Fr a0 = fr_zero(), a1 = fr_zero(), a2 = fr_zero();
    if (live) {
      const Fr* f = fac + (size_t)tid * L.NF + q0;
      a0 = fr_mul(f[0], h);
      if (q0 + 1 < L.NF) a1 = fr_mul(f[1], h);
      if (q0 + 2 < L.NF) a2 = fr_mul(f[2], h);
    }
    block_sum3_fr(a0, a1, a2, red);
    if (tid == 0) {
      S[q0] = a0;
      S[q0 + 1] = a1;
      S[q0 + 2] = a2;
    }
  }
  Fr sx0 = fr_zero(), sx1 = fr_zero(), A = fr_zero();
  if (tid < L.nHP) {
    const Fr hp = binb_bin_to_fr(sums + (size_t)(L.nH + tid) * BINB_LIMBS_LDS * 2);
    if (!l_is_zero(hp)) {
      const Fr t = fr_mul(tab[tid & 255u], hp);
      if (tid < 256) sx0 = t;
      else sx1 = t;
    }
  }
  for (uint32_t i = tid; i < nA; i += blockDim.x) A = fr_add(A, partialA[i]);
  block_sum3_fr(sx0, sx1, A, red);
  __syncthreads();
  if (tid == 0) {
    const Fr zero = fr_zero();
    auto put = [&](uint32_t r, const Fr& c0, const Fr& p1, const Fr& c2) {
      proof[3 * r + 0] = c0;
      proof[3 * r + 1] = fr_sub(fr_sub(p1, c0), c2);
      proof[3 * r + 2] = c2;
    };
    put(0, zero, zero, S[0]);
    put(1, S[1], S[2], S[3]);
    put(2, S[4], S[5], S[6]);
    Fr cross;
    if (L.lognb == 5) {
      const Fr u4 = u[4];
      cross = fr_add(fr_mul(fr_sub(fr_one(), u4), sx0), fr_mul(u4, sx1));
    } else {
      cross = sx0;
    }
    put(3, S[7], S[8], fr_sub(S[9], fr_dbl(cross)));
    if (L.lognb == 5) {
      const Fr m = fr_dbl(fr_mul(fr_sub(fr_one(), v[3]), v[3]));
      const Fr m0 = fr_mul(m, sx0), m1 = fr_mul(m, sx1);
      put(4, fr_add(S[10], m0), fr_add(S[11], m1), fr_sub(fr_add(S[12], fr_add(m0, m1)), fr_dbl(A)));
    }
  }
  // partial_me(u_rec, nb): out[8p + j] = sum over the bytes e with bit j set of H2[p][e]; lanes [256p, 256p + 256) own position p
  Fr h2 = fr_zero();
  if (own) h2 = binb_bin_to_fr(sums + (size_t)(L.nH + L.nHP + tid) * BINB_LIMBS_LDS * 2);
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  for (uint32_t j = 0; j < 8; j++) {
    Fr x = ((tid >> j) & 1u) ? h2 : fr_zero();   // the byte value of bin tid is tid & 255
    x = wave_sum_fr(x);
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    if (lane == 0 && (wave & 3u) == 0 && tid < L.nH) pme_out[8 * (tid >> 8) + j] = fr_add(fr_add(red[wave], red[wave + 1]), fr_add(red[wave + 2], red[wave + 3]));
  }
}

int sumcheck_bin_dev(zk_ctx* ctx, const Fr* a, uint32_t n, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d);
// binary_sumcheck of the bit tensor AND its partial_me(u_rec, nb) (pme_d receives nb elements; u_rec_d holds k - log2(nb) challenges)
int binb_sumcheck_pme_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d, const Fr* u_rec_d, Fr* pme_d) {
  if (nb != 32 && nb != 16) return ctx->fail(ZK_ERR_ARG, "packed sumcheck: nb must be 16 or 32");
  const BytesLayout L = bytes_layout(nb);
  if (n == 0 || k != ceil_log2(n) + L.lognb) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  const uint32_t kh = k - L.lognb;
  const uint32_t ch = kh < 10 ? kh : 10;
  const uint32_t nbins = L.nH + L.nHP + L.nH, ncount = nbins * BINB_LIMBS_LDS;
  ZK_ALLOC(ctx, A2, Fr, (size_t)1 << ch);
  ZK_ALLOC(ctx, B2, Fr, (size_t)1 << (kh - ch));
  ZK_ALLOC(ctx, tab, Fr, 768);
  ZK_ALLOC(ctx, AH, Fr, (size_t)1 << ch);
  ZK_ALLOC(ctx, BH, Fr, (size_t)1 << (kh - ch));
  ZK_ALLOC(ctx, fac, Fr, (size_t)L.nH * L.NF);
  const uint32_t nrest = 768 + 2 * ((1u << ch) + (1u << (kh - ch)));   // the tables, the word weights and the partial_me weights: one launch
  ZK_LAUNCHB(ctx, "binp_prep", 32ull * (nrest + L.nH * L.NF), 256, k_binb_prep, dim3(L.nH / 256 + stream_grid(nrest, 256, 16)), 0, u_d, v_d, k, L, ch, tab.get(), AH.get(), BH.get(),
            fac.get(), u_rec_d, A2.get(), B2.get());
  // one workgroup (1024 lanes) per BINB_WORDS_PER_WG words: few enough that no 64-bit counter can overflow, many enough to amortise
  // the workgroup's histogram set-up and flush
  const uint32_t grid = (n + BINB_WORDS_PER_WG - 1) / BINB_WORDS_PER_WG;
  ZK_ALLOC(ctx, out, Fr, n);
  ZK_ALLOC(ctx, partialA, Fr, grid);
  ZK_ALLOC(ctx, slabs, unsigned long long, (size_t)grid * ncount);
  ZK_ALLOC(ctx, sums, unsigned long long, (size_t)ncount * 2);
  const size_t lds = (size_t)ncount * 8 + (768 + 16) * sizeof(Fr);
  // algorithmic traffic: the packed words in, one field element per word out
  const uint64_t bytes = (uint64_t)n * (nb / 8) + 32ull * n;
  if (nb == 32) {
    if (ctx->lds_first_use(zk::Kern<1024, &k_binb_words<32, uint32_t>>::desc())) {
      zk::Kern<1024, &k_binb_words<32, uint32_t>>::set_max_lds(lds);
      (void)hipGetLastError();
    }
    ZK_LAUNCHB(ctx, "binp_words", bytes, 1024, (k_binb_words<32, uint32_t>), dim3(grid), lds, (const uint32_t*)packed, n, (const Fr*)tab.get(), (const Fr*)AH.get(), (const Fr*)BH.get(), ch,
              v_d, out.get(), partialA.get(), slabs.get(), (const Fr*)A2.get(), (const Fr*)B2.get(), ch);
  } else {
    if (ctx->lds_first_use(zk::Kern<1024, &k_binb_words<16, uint16_t>>::desc())) {
      zk::Kern<1024, &k_binb_words<16, uint16_t>>::set_max_lds(lds);
      (void)hipGetLastError();
    }
    ZK_LAUNCHB(ctx, "binp_words", bytes, 1024, (k_binb_words<16, uint16_t>), dim3(grid), lds, (const uint16_t*)packed, n, (const Fr*)tab.get(), (const Fr*)AH.get(), (const Fr*)BH.get(), ch,
              v_d, out.get(), partialA.get(), slabs.get(), (const Fr*)A2.get(), (const Fr*)B2.get(), ch);
  }
  ZK_LAUNCHB(ctx, "binp_finish", 8ull * grid * ncount, 256, k_binb_reduce, dim3(stream_grid(ncount * BINB_REDUCE_LANES, 256, 128)), 0, (const unsigned long long*)slabs.get(), grid, ncount, sums.get());
  ZK_LAUNCHB(ctx, "binp_finish", 16ull * ncount + 32ull * L.nH * L.NF, 1024, k_binb_finish, dim3(1), 0, (const unsigned long long*)sums.get(), (const Fr*)fac.get(),
            (const Fr*)tab.get(), (const Fr*)partialA.get(), nb == 32 ? grid : 0u, L, u_d, v_d, proof_d, pme_d);
  // rounds R .. k-1 and the final a(0): generic path on the table of one element per word
  return sumcheck_bin_dev(ctx, out.get(), n, u_d + L.R, v_d + L.R, k - L.R, proof_d + 3 * L.R);
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

// ====================================================================================================
// Sound mode (fs_rounds.hip): the in-word rounds of a bit sumcheck when round r's fold challenge is a hash of the messages of
// rounds < r (rounds 0..2 here, rounds 3 and 4 further down).  The weighted byte histogram H[p][e] = sum_{g : byte_p(g) = e} eq(u[log nb:], g) does not depend on any fold
// challenge, so it is taken ONCE, before the first round; each round's three coefficient sums are then  sum_bins f_r[bin] * H[bin]
// with factors that need only x_0 .. x_{r-1} (binb_factors' rounds 0..2) -- one workgroup computes the three messages one after the
// other, hashing between them, and leaves x_0..x_2, the table V3[byte] (a byte's eight bits folded by them) and, from the second
// histogram H2 (weights eq(u_rec, g)), the partial_me that follows the sumcheck.
template <int NB, class WordT>
__device__ __forceinline__ void k_fsb_hist(uint32_t zk_nblk, const WordT* __restrict__ packed, uint32_t n, const Fr* __restrict__ AH, const Fr* __restrict__ BH, uint32_t ch,
                                                   const Fr* __restrict__ A2, const Fr* __restrict__ B2, uint32_t c2, unsigned long long* __restrict__ slabs) {
  constexpr uint32_t NPOS = NB / 8, NH = NPOS * 256, NBINS = 2 * NH;
  extern __shared__ unsigned long long lds64[];
  unsigned long long* H = lds64;
  for (uint32_t i = threadIdx.x; i < NBINS * BINB_LIMBS_LDS; i += blockDim.x) H[i] = 0ull;
  __syncthreads();
  const uint32_t mh = (1u << ch) - 1u, m2 = (1u << c2) - 1u;
  const uint32_t per = (n + zk_nblk - 1) / zk_nblk;   // <= BINB_WORDS_PER_WG (host): no counter overflows
  const uint32_t g_end = min(n, (blockIdx.x + 1) * per);
  for (uint32_t g = blockIdx.x * per + threadIdx.x; g < g_end; g += blockDim.x) {
    const uint32_t word = packed[g];
    if (word == 0u) continue;
    unsigned long long l[5], l2[5];
    binb_limbs51(fr_mul(AH[g & mh], BH[g >> ch]), l);
    binb_limbs51(fr_mul(A2[g & m2], B2[g >> c2]), l2);
#pragma unroll
    for (uint32_t p = 0; p < NPOS; p++) {
      const uint32_t b = (word >> (8 * p)) & 255u;
      if (b) {
        unsigned long long* bin = H + (p * 256 + b) * BINB_LIMBS_LDS;
        unsigned long long* bin2 = bin + NH * BINB_LIMBS_LDS;
#pragma unroll
        for (int j = 0; j < 5; j++) atomicAdd(bin + j, l[j]);
#pragma unroll
        for (int j = 0; j < 5; j++) atomicAdd(bin2 + j, l2[j]);
      }
    }
  }
  __syncthreads();
  unsigned long long* dst = slabs + (size_t)blockIdx.x * NBINS * BINB_LIMBS_LDS;
  for (uint32_t i = threadIdx.x; i < NBINS * BINB_LIMBS_LDS; i += blockDim.x) dst[i] = H[i];
}

// One workgroup of 1024 lanes, one histogram bin (p, e) per lane.  proof[0..9): the messages of rounds 0..2; fold[0..3): x_0..x_2;
// v3tab[256]; pme_out[nb]; *state advanced by three rounds.
__device__ __forceinline__ void k_fsb_rounds(uint32_t zk_nblk, const unsigned long long* __restrict__ sums, BytesLayout L, const Fr* __restrict__ u, Fr* __restrict__ fold,
                                                     FsState* __restrict__ state, Fr* __restrict__ proof, Fr* __restrict__ pme_out, Fr* __restrict__ v3tab, Fr* __restrict__ hfr) {
  __shared__ Fr red[48];
  __shared__ Fr elo[28];
  __shared__ Fr xs[3];
  __shared__ Fr eqv[8];
  const uint32_t tid = threadIdx.x;
  if (tid < 28) {  // position weights of the three rounds: Elo_0[16] | Elo_1[8] | Elo_2[4]  (k_binb_prep)
    Fr r;
    if (tid < 16) r = tid < L.npos * 4 ? eq_prod(u + 1, L.lognb - 1, tid) : fr_zero();
    else if (tid < 24) r = (tid - 16) < L.npos * 2 ? eq_prod(u + 2, L.lognb - 2, tid - 16) : fr_zero();
    else r = (tid - 24) < L.npos ? eq_prod(u + 3, L.lognb - 3, tid - 24) : fr_zero();
    elo[tid] = r;
  }
  const bool own = tid < L.nH;  // nH <= blockDim.x = 1024
  const uint32_t p = tid >> 8, e = tid & 255u;
  Fr h = fr_zero();
  if (own) {
    h = binb_bin_to_fr(sums + (size_t)tid * BINB_LIMBS_LDS * 2);
    hfr[tid] = h;  // rounds 3 and 4 (k_fsb_round3 / k_fsb_round4) contract the same histogram with their own factors
  }
  const bool live = own && !l_is_zero(h);
  FsState st;
  if (tid == 0) st = *state;
  __syncthreads();
  const Fr zero = fr_zero(), one = fr_one();
  auto send = [&](uint32_t r, const Fr& c0, const Fr& p1, const Fr& c2) {  // lane 0: the round's message, then the hash step
    const Fr c1 = fr_sub(fr_sub(p1, c0), c2);
    proof[3 * r + 0] = c0;
    proof[3 * r + 1] = c1;
    proof[3 * r + 2] = c2;
    const Fr x = fs_round_step(st, c0, c1, c2);
    fold[r] = x;
    xs[r] = x;
  };
  {  // round 0: a pair is two bits: c0 = 0, c2 = sum of the weights of the pairs that differ, c1 = -c2
    Fr a = zero, z1 = zero, z2 = zero;
    if (live) {
      const uint32_t x = e ^ (e >> 1);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++)
        if ((x >> (2 * i)) & 1u) a = fr_add(a, elo[4 * p + i]);
      a = fr_mul(a, h);
    }
    block_sum3_fr(a, z1, z2, red);
    if (tid == 0) send(0, zero, zero, a);
  }
  __syncthreads();
  const Fr v0 = xs[0], nv0 = fr_sub(one, v0);
  {  // round 1: a pair is two bit pairs, each folded by x_0
    Fr c0 = zero, p1 = zero, c2 = zero;
    if (live) {
      for (uint32_t i = 0; i < 2; i++) {
        const uint32_t nib = (e >> (4 * i)) & 15u;
        if (!nib) continue;
        auto fold1 = [&](uint32_t bits) { return bits == 0 ? zero : bits == 1 ? nv0 : bits == 2 ? v0 : one; };
        const Fr a0 = fold1(nib & 3u), a1 = fold1(nib >> 2), w = elo[16 + 2 * p + i], d = fr_sub(a1, a0);
        c0 = fr_add(c0, fr_mul(w, fr_sub(fr_sqr(a0), a0)));
        p1 = fr_add(p1, fr_mul(w, fr_sub(fr_sqr(a1), a1)));
        c2 = fr_add(c2, fr_mul(w, fr_sqr(d)));
      }
      c0 = fr_mul(c0, h);
      p1 = fr_mul(p1, h);
      c2 = fr_mul(c2, h);
    }
    block_sum3_fr(c0, p1, c2, red);
    if (tid == 0) send(1, c0, p1, c2);
  }
  __syncthreads();
  const Fr v1 = xs[1], nv1 = fr_sub(one, v1);
  const Fr e4[4] = {fr_mul(nv0, nv1), fr_mul(v0, nv1), fr_mul(nv0, v1), fr_mul(v0, v1)};
  {  // round 2: a pair is the byte's two nibbles, each folded by (x_0, x_1)
    Fr c0 = zero, p1 = zero, c2 = zero;
    if (live) {
      Fr lo4 = zero, hi4 = zero;
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        if ((e >> i) & 1u) lo4 = fr_add(lo4, e4[i]);
        if ((e >> (4 + i)) & 1u) hi4 = fr_add(hi4, e4[i]);
      }
      const Fr w = fr_mul(elo[24 + p], h), d = fr_sub(hi4, lo4);
      c0 = fr_mul(w, fr_sub(fr_sqr(lo4), lo4));
      p1 = fr_mul(w, fr_sub(fr_sqr(hi4), hi4));
      c2 = fr_mul(w, fr_sqr(d));
    }
    block_sum3_fr(c0, p1, c2, red);
    if (tid == 0) {
      send(2, c0, p1, c2);
      *state = st;
    }
  }
  __syncthreads();
  if (tid < 8) {
    const Fr v2 = xs[2];
    eqv[tid] = fr_mul(e4[tid & 3u], (tid & 4u) ? v2 : fr_sub(one, v2));
  }
  __syncthreads();
  if (tid < 256) {
    Fr t = zero;
#pragma unroll
    for (uint32_t i = 0; i < 8; i++)
      if ((tid >> i) & 1u) t = fr_add(t, eqv[i]);
    v3tab[tid] = t;
  }
  // partial_me(u_rec, nb) from H2: as in k_binb_finish
  Fr h2 = fr_zero();
  if (own) h2 = binb_bin_to_fr(sums + (size_t)(L.nH + tid) * BINB_LIMBS_LDS * 2);
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  for (uint32_t j = 0; j < 8; j++) {
    Fr x = ((tid >> j) & 1u) ? h2 : fr_zero();
    x = wave_sum_fr(x);
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    if (lane == 0 && (wave & 3u) == 0 && tid < L.nH) pme_out[8 * (tid >> 8) + j] = fr_add(fr_add(red[wave], red[wave + 1]), fr_add(red[wave + 2], red[wave + 3]));
  }
}

// ---- rounds 3 and 4, still inside a word.  Their coefficient sums are, again, contractions of H with per-bin factors (now functions of
// V3, i.e. of x_0..x_2, and for round 4 of x_3) PLUS the terms that are not functions of single bytes (k_binb_finish):
//   round 3: c2 -= 2 sum_j eq(u[4..lognb), j) SX_j,   SX_j = sum_g w_g V3[byte_2j(g)] V3[byte_2j+1(g)]
//   round 4 (32-bit words): c0 += m SX_0, p1 += m SX_1, c2 += m (SX_0 + SX_1) - 2 A,   m = 2 (1 - x_3) x_3,   A = sum_g w_g a0(g) a1(g)
// SX needs V3 (a pass over the words behind round 2: k_fsb_cross, a histogram keyed by the low byte of each byte pair), A needs x_3 (a
// pass behind round 3: k_fsb_prodA).  The table the generic rounds start from has ONE entry per word (k_fsb_out, behind the last of these).
template <int NB, class WordT>
__device__ __forceinline__ void k_fsb_cross(uint32_t zk_nblk, const WordT* __restrict__ packed, uint32_t n, const Fr* __restrict__ AH, const Fr* __restrict__ BH, uint32_t ch,
                                                    const Fr* __restrict__ v3tab, unsigned long long* __restrict__ slabs) {
  constexpr uint32_t NPAIR = NB / 16, NBINS = NPAIR * 256;
  extern __shared__ unsigned long long lds64[];
  unsigned long long* H = lds64;
  Fr* V3 = reinterpret_cast<Fr*>(lds64 + NBINS * BINB_LIMBS_LDS);
  for (uint32_t i = threadIdx.x; i < NBINS * BINB_LIMBS_LDS; i += blockDim.x) H[i] = 0ull;
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) V3[i] = v3tab[i];
  __syncthreads();
  const uint32_t mh = (1u << ch) - 1u;
  const uint32_t per = (n + zk_nblk - 1) / zk_nblk;
  const uint32_t g_end = min(n, (blockIdx.x + 1) * per);
  for (uint32_t g = blockIdx.x * per + threadIdx.x; g < g_end; g += blockDim.x) {
    const uint32_t word = packed[g];
    bool any = false;
#pragma unroll
    for (uint32_t j = 0; j < NPAIR; j++) any = any || (((word >> (16 * j)) & 255u) && ((word >> (16 * j + 8)) & 255u));
    if (!any) continue;
    const Fr w = fr_mul(AH[g & mh], BH[g >> ch]);
#pragma unroll
    for (uint32_t j = 0; j < NPAIR; j++) {
      const uint32_t b0 = (word >> (16 * j)) & 255u, b1 = (word >> (16 * j + 8)) & 255u;
      if (b0 && b1) binb_hist_add(H + (j * 256 + b0) * BINB_LIMBS_LDS, fr_mul(w, V3[b1]));
    }
  }
  __syncthreads();
  unsigned long long* dst = slabs + (size_t)blockIdx.x * NBINS * BINB_LIMBS_LDS;
  for (uint32_t i = threadIdx.x; i < NBINS * BINB_LIMBS_LDS; i += blockDim.x) dst[i] = H[i];
}

// One workgroup of 1024 lanes: round 3's message from the stored histogram and the cross-term sums, the hash step, and what x_3 unlocks:
// tabAB = [VA = (1 - x_3) V3 | VB = x_3 V3] (512 entries) and sx[0..2) for round 4
__device__ __forceinline__ void k_fsb_round3(uint32_t zk_nblk, const Fr* __restrict__ hfr, const unsigned long long* __restrict__ xsums, BytesLayout L, const Fr* __restrict__ u,
                                                     const Fr* __restrict__ v3tab, Fr* __restrict__ fold, FsState* __restrict__ state, Fr* __restrict__ proof, Fr* __restrict__ tabAB,
                                                     Fr* __restrict__ sx_out) {
  __shared__ Fr red[48];
  __shared__ Fr xs[1];
  const uint32_t tid = threadIdx.x;
  const bool own = tid < L.nH;
  const uint32_t p = tid >> 8, e = tid & 255u;
  const Fr zero = fr_zero(), one = fr_one();
  Fr c0 = zero, p1 = zero, c2 = zero;
  const Fr t = v3tab[e];
  if (own) {
    const Fr h = hfr[tid];
    if (!l_is_zero(h) && e) {
      const Fr w = fr_mul(eq_prod(u + 4, L.lognb - 4, p >> 1), h), sq = fr_sqr(t);
      const Fr q = fr_mul(w, fr_sub(sq, t));
      if (p & 1u) p1 = q;
      else c0 = q;
      c2 = fr_mul(w, sq);
    }
  }
  block_sum3_fr(c0, p1, c2, red);
  Fr sx0 = zero, sx1 = zero, z2 = zero;
  if (tid < L.nHP) {
    const Fr hp = binb_bin_to_fr(xsums + (size_t)tid * BINB_LIMBS_LDS * 2);
    if (!l_is_zero(hp)) {
      const Fr v = fr_mul(t, hp);   // t = V3[tid & 255]
      if (tid < 256) sx0 = v;
      else sx1 = v;
    }
  }
  block_sum3_fr(sx0, sx1, z2, red);
  if (tid == 0) {
    Fr cross = sx0;
    if (L.lognb == 5) {
      const Fr u4 = u[4];
      cross = fr_add(fr_mul(fr_sub(one, u4), sx0), fr_mul(u4, sx1));
    }
    c2 = fr_sub(c2, fr_dbl(cross));
    const Fr c1 = fr_sub(fr_sub(p1, c0), c2);
    proof[0] = c0;
    proof[1] = c1;
    proof[2] = c2;
    FsState st = *state;
    const Fr x = fs_round_step(st, c0, c1, c2);
    *state = st;
    fold[0] = x;
    xs[0] = x;
    sx_out[0] = sx0;
    sx_out[1] = sx1;
  }
  __syncthreads();
  if (tid < 512) {
    const Fr x3 = xs[0];
    tabAB[tid] = fr_mul(tid < 256 ? fr_sub(one, x3) : x3, t);
  }
}

// A = sum_g w_g a0(g) a1(g) over 32-bit words, a0 = VA[b0] + VB[b1], a1 = VA[b2] + VB[b3]: per-workgroup partials
__device__ __forceinline__ void k_fsb_prodA(uint32_t zk_nblk, const uint32_t* __restrict__ packed, uint32_t n, const Fr* __restrict__ AH, const Fr* __restrict__ BH, uint32_t ch,
                                                    const Fr* __restrict__ tabAB, Fr* __restrict__ partialA) {
  __shared__ Fr tab[512];
  __shared__ Fr red[16];
  for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = tabAB[i];
  __syncthreads();
  const uint32_t mh = (1u << ch) - 1u;
  Fr acc = fr_zero();
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += zk_nblk * blockDim.x) {
    const uint32_t word = packed[g];
    if ((word & 0xffffu) == 0u || (word >> 16) == 0u) continue;
    const Fr a0 = fr_add(tab[word & 255u], tab[256 + ((word >> 8) & 255u)]), a1 = fr_add(tab[(word >> 16) & 255u], tab[256 + (word >> 24)]);
    acc = fr_add(acc, fr_mul(fr_mul(fr_mul(AH[g & mh], BH[g >> ch]), a0), a1));
  }
  acc = block_sum_fr(acc, red);
  if (threadIdx.x == 0) partialA[blockIdx.x] = acc;
}

// One workgroup of 1024 lanes (32-bit words): round 4's message and x_4
__device__ __forceinline__ void k_fsb_round4(uint32_t zk_nblk, const Fr* __restrict__ hfr, const Fr* __restrict__ v3tab, const Fr* __restrict__ x3p, const Fr* __restrict__ sx,
                                                     const Fr* __restrict__ partialA, uint32_t nA, Fr* __restrict__ fold, FsState* __restrict__ state, Fr* __restrict__ proof) {
  __shared__ Fr red[48];
  const uint32_t tid = threadIdx.x;
  const uint32_t p = tid >> 8, e = tid & 255u;
  const Fr zero = fr_zero(), one = fr_one();
  const Fr v3 = x3p[0];
  Fr c0 = zero, p1 = zero, c2 = zero;
  {
    const Fr h = hfr[tid];   // nH = 1024 = blockDim.x
    if (!l_is_zero(h) && e) {
      const Fr t = v3tab[e], c = (p & 1u) ? v3 : fr_sub(one, v3);
      const Fr c2sq = fr_mul(fr_mul(fr_sqr(c), fr_sqr(t)), h), q = fr_sub(c2sq, fr_mul(fr_mul(c, t), h));
      if (p < 2) c0 = q;
      else p1 = q;
      c2 = c2sq;
    }
  }
  block_sum3_fr(c0, p1, c2, red);
  Fr A = zero, z1 = zero, z2 = zero;
  for (uint32_t i = tid; i < nA; i += blockDim.x) A = fr_add(A, partialA[i]);
  block_sum3_fr(A, z1, z2, red);
  if (tid == 0) {
    const Fr m = fr_dbl(fr_mul(fr_sub(one, v3), v3));
    const Fr m0 = fr_mul(m, sx[0]), m1 = fr_mul(m, sx[1]);
    c0 = fr_add(c0, m0);
    p1 = fr_add(p1, m1);
    c2 = fr_sub(fr_add(c2, fr_add(m0, m1)), fr_dbl(A));
    const Fr c1 = fr_sub(fr_sub(p1, c0), c2);
    proof[0] = c0;
    proof[1] = c1;
    proof[2] = c2;
    FsState st = *state;
    fold[0] = fs_round_step(st, c0, c1, c2);
    *state = st;
  }
}

// the table the generic rounds start from, one entry per word: a0 + x_4 (a1 - a0) (32-bit words; x4p = the challenge of round 4), a0 (16-bit words)
template <int NB, class WordT>
__device__ __forceinline__ void k_fsb_out(uint32_t zk_nblk, const WordT* __restrict__ packed, uint32_t n, const Fr* __restrict__ tabAB, const Fr* __restrict__ x4p, Fr* __restrict__ out) {
  __shared__ Fr tab[512];
  for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = tabAB[i];
  __syncthreads();
  Fr x4 = fr_zero();
  if (NB == 32) x4 = x4p[0];
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += zk_nblk * blockDim.x) {
    const uint32_t word = packed[g];
    if (word == 0u) {
      out[g] = fr_zero();
      continue;
    }
    const Fr a0 = fr_add(tab[word & 255u], tab[256 + ((word >> 8) & 255u)]);
    if (NB == 32) {
      const Fr a1 = fr_add(tab[(word >> 16) & 255u], tab[256 + (word >> 24)]);
      out[g] = fr_add(a0, fr_mul(x4, fr_sub(a1, a0)));
    } else {
      out[g] = a0;
    }
  }
}

// The in-word rounds of a bit sumcheck in the sound mode: messages into proof_d[0 .. 3R), x_0..x_{R-1} into fold_d, *state_d advanced (R = 5 for
// 32-bit words, 4 for 16-bit words: *rounds_done), partial_me(u_rec, nb) into pme_d, and the table of round R (one entry per word) into t
int fsb_sumcheck_head_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, uint32_t k, void* state_d, Fr* fold_d, Fr* proof_d, const Fr* u_rec_d, Fr* pme_d,
                          DBuf<Fr>& t, uint32_t* rounds_done) {
  if (nb != 32 && nb != 16) return ctx->fail(ZK_ERR_ARG, "packed sumcheck: nb must be 16 or 32");
  const BytesLayout L = bytes_layout(nb);
  if (n == 0 || k != ceil_log2(n) + L.lognb) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  FsState* state = reinterpret_cast<FsState*>(state_d);
  const uint32_t kh = k - L.lognb;
  const uint32_t ch = kh < 10 ? kh : 10;
  const uint32_t nbins = 2 * L.nH, ncount = nbins * BINB_LIMBS_LDS, xcount = L.nHP * BINB_LIMBS_LDS;
  ZK_ALLOC(ctx, AH, Fr, (size_t)1 << ch);
  ZK_ALLOC(ctx, BH, Fr, (size_t)1 << (kh - ch));
  ZK_ALLOC(ctx, A2, Fr, (size_t)1 << ch);
  ZK_ALLOC(ctx, B2, Fr, (size_t)1 << (kh - ch));
  const uint32_t nt = (1u << ch) + (1u << (kh - ch));
  ZK_LAUNCHB(ctx, "binp_prep", 32ull * nt, 256, k_eq_pair_tables, dim3(stream_grid(nt, 256, 16)), 0, u_d + L.lognb, kh, ch, AH.get(), BH.get());
  ZK_LAUNCHB(ctx, "binp_prep", 32ull * nt, 256, k_eq_pair_tables, dim3(stream_grid(nt, 256, 16)), 0, u_rec_d, kh, ch, A2.get(), B2.get());
  const uint32_t grid = (n + BINB_WORDS_PER_WG - 1) / BINB_WORDS_PER_WG;
  ZK_ALLOC(ctx, slabs, unsigned long long, (size_t)grid * ncount);
  ZK_ALLOC(ctx, sums, unsigned long long, (size_t)ncount * 2);
  ZK_ALLOC(ctx, xslabs, unsigned long long, (size_t)grid * xcount);
  ZK_ALLOC(ctx, xsums, unsigned long long, (size_t)xcount * 2);
  ZK_ALLOC(ctx, v3tab, Fr, 256);
  ZK_ALLOC(ctx, hfr, Fr, 1024);
  ZK_ALLOC(ctx, tabAB, Fr, 512);
  ZK_ALLOC(ctx, sx, Fr, 2);
  const size_t lds = (size_t)ncount * 8, xlds = (size_t)xcount * 8 + 256 * sizeof(Fr);
  const uint64_t bytes = (uint64_t)n * (nb / 8);
  const uint32_t* p32 = (const uint32_t*)packed;
  const uint16_t* p16 = (const uint16_t*)packed;
  // ---- before any challenge: the histograms
  if (nb == 32) {
    if (ctx->lds_first_use(zk::Kern<1024, &k_fsb_hist<32, uint32_t>>::desc())) {
      zk::Kern<1024, &k_fsb_hist<32, uint32_t>>::set_max_lds(lds);
      (void)hipGetLastError();
    }
    ZK_LAUNCHB(ctx, "fs_bit_hist", bytes, 1024, (k_fsb_hist<32, uint32_t>), dim3(grid), lds, p32, n, (const Fr*)AH.get(), (const Fr*)BH.get(), ch, (const Fr*)A2.get(), (const Fr*)B2.get(), ch,
              slabs.get());
  } else {
    if (ctx->lds_first_use(zk::Kern<1024, &k_fsb_hist<16, uint16_t>>::desc())) {
      zk::Kern<1024, &k_fsb_hist<16, uint16_t>>::set_max_lds(lds);
      (void)hipGetLastError();
    }
    ZK_LAUNCHB(ctx, "fs_bit_hist", bytes, 1024, (k_fsb_hist<16, uint16_t>), dim3(grid), lds, p16, n, (const Fr*)AH.get(), (const Fr*)BH.get(), ch, (const Fr*)A2.get(), (const Fr*)B2.get(), ch,
              slabs.get());
  }
  ZK_LAUNCHB(ctx, "binp_finish", 8ull * grid * ncount, 256, k_binb_reduce, dim3(stream_grid(ncount * BINB_REDUCE_LANES, 256, 128)), 0, (const unsigned long long*)slabs.get(), grid, ncount, sums.get());
  // ---- rounds 0, 1, 2 (one workgroup, hashing between them); leaves x_0..x_2, V3 and the normalised histogram
  ZK_LAUNCHB(ctx, "fs_bit_rounds", 16ull * ncount, 1024, k_fsb_rounds, dim3(1), 0, (const unsigned long long*)sums.get(), L, u_d, fold_d, state, proof_d, pme_d, v3tab.get(), hfr.get());
  // ---- round 3: the byte pairs' cross terms need V3
  if (nb == 32) {
    ZK_LAUNCHB(ctx, "fs_bit_hist", bytes, 1024, (k_fsb_cross<32, uint32_t>), dim3(grid), xlds, p32, n, (const Fr*)AH.get(), (const Fr*)BH.get(), ch, (const Fr*)v3tab.get(), xslabs.get());
  } else {
    ZK_LAUNCHB(ctx, "fs_bit_hist", bytes, 1024, (k_fsb_cross<16, uint16_t>), dim3(grid), xlds, p16, n, (const Fr*)AH.get(), (const Fr*)BH.get(), ch, (const Fr*)v3tab.get(), xslabs.get());
  }
  ZK_LAUNCHB(ctx, "binp_finish", 8ull * grid * xcount, 256, k_binb_reduce, dim3(stream_grid(xcount * BINB_REDUCE_LANES, 256, 128)), 0, (const unsigned long long*)xslabs.get(), grid, xcount, xsums.get());
  ZK_LAUNCHB(ctx, "fs_bit_rounds", 16ull * xcount + 32ull * L.nH, 1024, k_fsb_round3, dim3(1), 0, (const Fr*)hfr.get(), (const unsigned long long*)xsums.get(), L, u_d, (const Fr*)v3tab.get(), fold_d + 3,
            state, proof_d + 9, tabAB.get(), sx.get());
  // ---- round 4 (32-bit words): the product of a word's two halves needs x_3
  const uint32_t gridA = stream_grid(n, 256, 1024);
  ZK_ALLOC(ctx, partialA, Fr, gridA);
  if (nb == 32) {
    ZK_LAUNCHB(ctx, "fs_bit_hist", bytes, 256, k_fsb_prodA, dim3(gridA), 0, p32, n, (const Fr*)AH.get(), (const Fr*)BH.get(), ch, (const Fr*)tabAB.get(), partialA.get());
    ZK_LAUNCHB(ctx, "fs_bit_rounds", 32ull * (L.nH + gridA), 1024, k_fsb_round4, dim3(1), 0, (const Fr*)hfr.get(), (const Fr*)v3tab.get(), (const Fr*)(fold_d + 3), (const Fr*)sx.get(),
              (const Fr*)partialA.get(), gridA, fold_d + 4, state, proof_d + 12);
  }
  // ---- the table of round R: one entry per word
  if (!t.alloc(ctx, n)) return ctx->fail(ZK_ERR_ALLOC, "device allocation failed: sumcheck tables");
  if (nb == 32)
    ZK_LAUNCHB(ctx, "fs_bit_expand", bytes + 32ull * n, 256, (k_fsb_out<32, uint32_t>), dim3(stream_grid(n, 256, 2048)), 0, p32, n, (const Fr*)tabAB.get(), (const Fr*)(fold_d + 4), t.get());
  else
    ZK_LAUNCHB(ctx, "fs_bit_expand", bytes + 32ull * n, 256, (k_fsb_out<16, uint16_t>), dim3(stream_grid(n, 256, 2048)), 0, p16, n, (const Fr*)tabAB.get(), (const Fr*)nullptr, t.get());
  *rounds_done = L.R;
  return ZK_OK;
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
