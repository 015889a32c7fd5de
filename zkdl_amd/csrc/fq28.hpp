// zkdl_amd/csrc/fq28.hpp -- the base field of bls12-381 in 14 limbs of 28 bits, lazily reduced: the arithmetic the
// bucket pipeline of the MSM (msm.hip) runs on.
//
// Why a second representation.  On gfx950 a 32x32->64 multiply-accumulate is ONE instruction (v_mad_u64_u32) but its
// carry out of the 64-bit accumulator costs a second one (v_addc_co_u32): with saturated 32-bit limbs a 381-bit
// Montgomery product is 288 MACs + 288 carry adds + a conditional subtraction (~670 VALU instructions,
// field.hpp:f_mul_dev), and the pipeline is bound by VALU issue slots.  With 28-bit limbs a column
// of 14 + 14 partial products (each < 2^60) fits a 64-bit accumulator with room to spare, so a MAC is the single
// v_mad_u64_u32 and the column shift is one v_lshrrev_b64: 392 MACs + ~70 other instructions.  The four spare bits per
// limb and the eleven spare bits of R = 2^392 over p also make additions and subtractions carry-free (one v_add /
// v_sub + v_add per limb) and remove every conditional subtraction.
//
// Conventions
//   value(a) = sum_i a.v[i] * 2^(28 i); "normalised": v[i] < 2^28 for i < 13 (v[13] holds the rest).
//   Montgomery domain R' = 2^392: the element x is stored as x * 2^392 mod p ("n-form"); the rest of the library keeps
//   x * 2^384 mod p in 12 x 32 bits ("o-form", field.hpp).  n-form = o-form * 2^8: window tables are converted once
//   when they are built (msm.hip:k_table_nform) and MSM results are converted back by fq_to_fp (one product).
//   Values are only bounded, not reduced: every product returns a normalised value < 2p provided
//   value(a) * value(b) < 2^11 p^2 and  14 * max_limb(a) * max_limb(b) + 2^60 < 2^64;  the callers (g1x_* in msm.hip)
//   state the bound of each operand.  Nothing here is on the reference's path (it has no MSM); the reference's field
//   arithmetic these values must agree with is bls12-381.cu:259-330 (mod p add/sub/mul), through the canonical results
//   the parity tests compare.
#pragma once
#include "field.hpp"
#include "fpinv.hpp"

namespace zk {

struct alignas(8) Fq {
  uint32_t v[14];
};

namespace fq28 {
constexpr int N = 14;
constexpr uint32_t MASK = (1u << 28) - 1u;
// 28-bit limb i of a 12 x 32-bit constant
template <class GET>
ZK_HD constexpr uint32_t limb_of(GET get, int i) {
  int bit = 28 * i, li = bit >> 5, sh = bit & 31;
  uint64_t lo = get(li), hi = li + 1 < 12 ? get(li + 1) : 0;
  return (uint32_t)((((hi << 32) | lo) >> sh) & MASK);
}
ZK_HD constexpr uint32_t P(int i) { return limb_of([](int j) constexpr { return FpP::P(j); }, i); }
ZK_HD constexpr uint32_t C384(int i) { return limb_of([](int j) constexpr { return FpP::ONE(j); }, i); }  // 2^384 mod p
constexpr uint32_t INV = FpP::INV & MASK;                                                                 // -1/p mod 2^28
// 2^392 mod p: the n-form of 1
ZK_HD constexpr uint32_t ONE(int i) {
  constexpr uint32_t o[14] = {55049400u, 226492416u, 176409u, 13493970u, 208802528u, 138027152u, 3630751u,
                              228652862u, 162569026u, 19495856u, 85318031u, 1231484u, 106560762u, 22394u};
  return o[i];
}
// K*p arranged for carry-free subtraction of NSUB normalised values: limb i < 13 carries an extra NSUB * 2^28 borrowed
// from limb i+1, so that  a.v[i] + C.v[i] - b1.v[i] - ... never goes negative.  Needs K*p's top limb - NSUB to cover the
// subtrahends' top limbs: K >= (sum of the subtrahends' value bounds, in units of p) + 1 is enough (p's top limb is ~10^5).
struct SubConst {
  uint32_t v[14];
};
ZK_HD constexpr SubConst sub_const(uint32_t K, uint32_t nsub) {
  SubConst c{};
  uint64_t carry = 0;
  for (int i = 0; i < 14; i++) {
    uint64_t t = (uint64_t)P(i) * K + carry;
    c.v[i] = i < 13 ? (uint32_t)(t & MASK) : (uint32_t)t;
    carry = t >> 28;
  }
  for (int i = 0; i < 13; i++) {
    c.v[i] += nsub << 28;
    c.v[i + 1] -= nsub;
  }
  return c;
}
}  // namespace fq28

typedef uint32_t zk_q4 __attribute__((ext_vector_type(4)));
typedef uint32_t zk_q2 __attribute__((ext_vector_type(2)));

ZK_HD Fq fq_zero() {
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = 0;
  return r;
}
ZK_HD Fq fq_one() {
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = fq28::ONE(i);
  return r;
}
// every limb zero (the explicit infinity marker; NOT a test for "is a multiple of p")
ZK_HD bool fq_all_zero(const Fq& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) acc |= a.v[i];
  return acc == 0;
}
// a normalised, < 2p:  a == 0 (mod p)  <=>  a is 0 or p
ZK_HD bool fq_is_zero_mod_p(const Fq& a) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ fq28::P(i);
  }
  return z == 0 || e == 0;
}

// ---- Montgomery product, product scanning with a 64-bit column accumulator and no carries.
// returns (a*b + m*p) / 2^392, normalised, < a*b/2^392 + p.
ZK_HD Fq fq_mul_inl(const Fq& a, const Fq& b) {
  constexpr int N = 14;
  uint32_t m[N];
  Fq t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fq28::P(k - i);
    m[k] = ((uint32_t)acc * fq28::INV) & fq28::MASK;
    acc += (uint64_t)m[k] * fq28::P(0);  // low 28 bits are now zero
    acc >>= 28;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * fq28::P(k - i);
    t.v[k - N] = (uint32_t)acc & fq28::MASK;
    acc >>= 28;
  }
  t.v[N - 1] = (uint32_t)acc;
  return t;
}
// square: the cross products once, against the doubled operand.  Needs 7 * 2 * max_limb(a)^2 < 2^63.
ZK_HD Fq fq_sqr_inl(const Fq& a) {
  constexpr int N = 14;
  uint32_t m[N], d[N];
#pragma unroll
  for (int i = 0; i < N; i++) d[i] = a.v[i] << 1;
  Fq t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) acc += (uint64_t)d[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fq28::P(k - i);
    m[k] = ((uint32_t)acc * fq28::INV) & fq28::MASK;
    acc += (uint64_t)m[k] * fq28::P(0);
    acc >>= 28;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; 2 * i < k; i++) acc += (uint64_t)d[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
    for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * fq28::P(k - i);
    t.v[k - N] = (uint32_t)acc & fq28::MASK;
    acc >>= 28;
  }
  t.v[N - 1] = (uint32_t)acc;
  return t;
}
// Real device calls (a mixed addition holds ten of them); operands travel in VGPRs as vector arguments, see
// field.hpp:fp_mul_regs.
static ZK_HD_CALL Fq fq_mul_regs(zk_q4 a0, zk_q4 a1, zk_q4 a2, zk_q2 a3, zk_q4 b0, zk_q4 b1, zk_q4 b2, zk_q2 b3) {
  Fq a, b;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a.v[i] = a0[i]; a.v[4 + i] = a1[i]; a.v[8 + i] = a2[i];
    b.v[i] = b0[i]; b.v[4 + i] = b1[i]; b.v[8 + i] = b2[i];
  }
  a.v[12] = a3[0]; a.v[13] = a3[1];
  b.v[12] = b3[0]; b.v[13] = b3[1];
  return fq_mul_inl(a, b);
}
static ZK_HD_CALL Fq fq_sqr_regs(zk_q4 a0, zk_q4 a1, zk_q4 a2, zk_q2 a3) {
  Fq a;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a.v[i] = a0[i]; a.v[4 + i] = a1[i]; a.v[8 + i] = a2[i];
  }
  a.v[12] = a3[0]; a.v[13] = a3[1];
  return fq_sqr_inl(a);
}
ZK_HD Fq fq_mul(const Fq& a, const Fq& b) {
  zk_q4 a0 = {a.v[0], a.v[1], a.v[2], a.v[3]}, a1 = {a.v[4], a.v[5], a.v[6], a.v[7]}, a2 = {a.v[8], a.v[9], a.v[10], a.v[11]};
  zk_q4 b0 = {b.v[0], b.v[1], b.v[2], b.v[3]}, b1 = {b.v[4], b.v[5], b.v[6], b.v[7]}, b2 = {b.v[8], b.v[9], b.v[10], b.v[11]};
  zk_q2 a3 = {a.v[12], a.v[13]}, b3 = {b.v[12], b.v[13]};
  return fq_mul_regs(a0, a1, a2, a3, b0, b1, b2, b3);
}
ZK_HD Fq fq_sqr(const Fq& a) {
  zk_q4 a0 = {a.v[0], a.v[1], a.v[2], a.v[3]}, a1 = {a.v[4], a.v[5], a.v[6], a.v[7]}, a2 = {a.v[8], a.v[9], a.v[10], a.v[11]};
  zk_q2 a3 = {a.v[12], a.v[13]};
  return fq_sqr_regs(a0, a1, a2, a3);
}

// limb-wise sum: limbs and value add up
ZK_HD Fq fq_add(const Fq& a, const Fq& b) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a + K p - b,  b normalised with value(b) <= (K - 1) p:  value < value(a) + K p,  limbs < max_limb(a) + 2^29
template <uint32_t K>
ZK_HD Fq fq_sub(const Fq& a, const Fq& b) {
  constexpr fq28::SubConst C = fq28::sub_const(K, 1);
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = a.v[i] - b.v[i] + C.v[i];
  return r;
}
// a + K p - b - c - c  (b, c normalised, value(b) + 2 value(c) <= (K - 1) p): limbs < max_limb(a) + 2^30
template <uint32_t K>
ZK_HD Fq fq_sub3(const Fq& a, const Fq& b, const Fq& c) {
  constexpr fq28::SubConst C = fq28::sub_const(K, 3);
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = a.v[i] - b.v[i] - (c.v[i] << 1) + C.v[i];
  return r;
}
// a + K p - b - c  (b, c normalised, value(b) + value(c) <= (K - 1) p)
template <uint32_t K>
ZK_HD Fq fq_sub2(const Fq& a, const Fq& b, const Fq& c) {
  constexpr fq28::SubConst C = fq28::sub_const(K, 2);
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = a.v[i] - b.v[i] - c.v[i] + C.v[i];
  return r;
}
// a + K p - 8 c  (c normalised, 8 value(c) <= (K - 1) p): limbs < max_limb(a) + 2^31 + 2^28
template <uint32_t K>
ZK_HD Fq fq_sub8(const Fq& a, const Fq& c) {
  constexpr fq28::SubConst C = fq28::sub_const(K, 8);
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = a.v[i] - (c.v[i] << 3) + C.v[i];
  return r;
}
// carry the limbs back under 2^28 (value unchanged)
ZK_HD Fq fq_norm(const Fq& a) {
  Fq r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 13; i++) {
    uint32_t t = a.v[i] + c;
    r.v[i] = t & fq28::MASK;
    c = t >> 28;
  }
  r.v[13] = a.v[13] + c;
  return r;
}

// 12 x 32 bits (any value < 2^384) -> 14 x 28 bits, same integer
ZK_HD Fq fq_unpack(const Fp& a) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const int bit = 28 * i, li = bit >> 5, sh = bit & 31;
    uint32_t lo = a.v[li];
    uint32_t v;
    if (sh == 0) v = lo;
    else if (sh <= 4 || li + 1 >= 12) v = lo >> sh;
    else v = (lo >> sh) | (a.v[li + 1 < 12 ? li + 1 : 11] << (32 - sh));
    r.v[i] = v & fq28::MASK;
  }
  return r;
}
// normalised, value < 2^384 -> 12 x 32 bits
ZK_HD Fp fq_pack(const Fq& a) {
  Fp r;
#pragma unroll
  for (int j = 0; j < 12; j++) {
    const int bit = 32 * j, li = bit / 28, off = bit - 28 * li;  // word j starts off bits into limb li
    uint32_t w = a.v[li] >> off;
    if (li + 1 < 14) w |= a.v[li + 1] << (28 - off);
    if (li + 2 < 14 && 56 - off < 32) w |= a.v[li + 2] << (56 - off);
    r.v[j] = w;
  }
  return r;
}
// n-form, any bounded value -> canonical o-form (x 2^384 mod p, fully reduced): the way out of the pipeline
ZK_HD Fp fq_to_fp(const Fq& a) {
  Fq c;
#pragma unroll
  for (int i = 0; i < 14; i++) c.v[i] = fq28::C384(i);
  return f_reduce_once<FpP>(fq_pack(fq_mul(a, c)));  // a * 2^384 / 2^392 = a / 2^8, < 2p
}

// canonical o-form (x 2^384 mod p, 12 x 32 bits) -> n-form: the way into the pipeline for values that do not come from a window table
ZK_HD Fq fq_from_fp(const Fp& a) {
  constexpr uint32_t c400[14] = {135160473u, 55574580u, 246491222u, 233514649u, 211322480u, 82798231u, 118064104u,
                                 172912617u, 65577043u, 248456990u, 203477795u, 116089733u, 75436678u, 87770u};  // 2^400 mod p
  Fq c;
#pragma unroll
  for (int i = 0; i < 14; i++) c.v[i] = c400[i];
  return fq_mul(fq_unpack(a), c);  // a 2^400 / 2^392 = a 2^8
}
// the inverse of a bounded n-form value (normalised, < 2^5 p), n-form, < 2p:  through the canonical 12 x 32-bit form and the binary GCD of
// fpinv.hpp (two conversion products + ~33 k scalar-width instructions, against the ~290 k of the Fermat chain below).  0 -> 0.
ZK_HD Fq fq_inv(const Fq& a) { return fq_from_fp(fp_inv_bingcd(fq_to_fp(a))); }
// a^(p-2), a normalised and < 2^5 p (Fermat inverse; the independent cross-check of fq_inv; backs the affine-canonical form, see field.hpp:fp_inv), by 4-bit fixed windows:
// a^1 .. a^15 in `tab` (entry i at tab[i * stride]: the caller's scratch -- LDS on the device, where every lane of a wave reads the same
// entry index because the exponent is a constant), then 95 x (four squarings + one table product): 380 + 91 + 14 products instead of the
// 381 + 229 of the bitwise ladder.  The inversion is ONE lane's latency chain behind every proof's G1 phase.
ZK_HD Fq fq_inv_fermat(const Fq& a, Fq* tab, int stride) {
  tab[1 * stride] = a;
#pragma unroll 1
  for (int i = 2; i < 16; i++) tab[i * stride] = fq_mul(tab[(i - 1) * stride], a);
  // exponent p - 2, most significant nibble first; p's low limb is 0xffffaaab, so p - 2 only changes limb 0 (and its lowest nibble: b -> 9)
  Fq acc = tab[1 * stride];  // the top nibble of the 384-bit exponent is 1 (p = 0x1a01...)
#pragma unroll 1
  for (int i = 94; i >= 0; i--) {
    uint32_t limb = FpP::P(0) - 2u;
#pragma unroll
    for (int j = 1; j < 12; j++) limb = ((i >> 3) == j) ? FpP::P(j) : limb;
    const uint32_t d = (limb >> (4 * (i & 7))) & 15u;
    acc = fq_sqr(fq_sqr(fq_sqr(fq_sqr(acc))));
    if (d) acc = fq_mul(acc, tab[d * stride]);
  }
  return acc;
}

}  // namespace zk
