// zkdl_amd/csrc/fs29.hpp -- the scalar field of bls12-381 in 9 limbs of 29 bits, for the inner loops of the big sumcheck
// rounds (sumcheck.hip: k_sc_round29).
//
// Same reason as fq28.hpp: with saturated 32-bit limbs every 32x32->64 multiply-accumulate needs a second instruction for
// the carry out of the 64-bit accumulator, and these kernels are bound by VALU issue slots.  With 29-bit limbs a column of
// 9 + 9 partial products (each < 2^58) fits the accumulator, so a MAC is the single v_mad_u64_u32.  Two more savings come
// with it:
//   * sums and differences are carry-free (limb-wise), and
//   * the three round coefficients are SUMS of products, so the products that feed them are accumulated UNREDUCED
//     (81 MACs each, fs_mul_raw_acc) as 18 x 29-bit columns per lane; lanes, waves and the workgroup add columns as plain
//     integers, and one lane per coefficient turns the 522+-bit total into a canonical Fr with three ordinary Montgomery
//     products (fs_cols_to_fr).  Only the products whose results are operands again (the weighted operands, the fold) pay
//     for a Montgomery reduction.
//
// Conventions
//   value(a) = sum_i a.v[i] * 2^(29 i); "normalised": v[i] < 2^29 (v[8] holds the rest).
//   The Montgomery domain stays R = 2^256 (the tables in HBM are the library's canonical 8 x 32-bit Montgomery form):
//   fs_mul reduces by eight full limbs and one 24-bit step.  r/R = 0.453, so a product of values A r and B r returns a
//   value < (0.453 A B + 1) r: values stay bounded because every chain here is two products deep and starts from canonical
//   (< r) table entries; the callers state the bounds.  Everything is checked bit for bit against the reference's
//   arithmetic (bls12-381.cu:662-760) through the transcripts the parity tests compare, and on the CPU against big-integer
//   arithmetic (tests/test_fs29.py).
#pragma once
#include "field.hpp"

namespace zk {

struct Fs {
  uint32_t v[9];
};

namespace fs29 {
constexpr uint32_t MASK = (1u << 29) - 1u;
template <class GET>
ZK_HD constexpr uint32_t limb_of(GET get, int i) {
  int bit = 29 * i, li = bit >> 5, sh = bit & 31;
  uint64_t lo = get(li), hi = li + 1 < 8 ? get(li + 1) : 0;
  return (uint32_t)((((hi << 32) | lo) >> sh) & MASK);
}
ZK_HD constexpr uint32_t R(int i) { return limb_of([](int j) constexpr { return FrP::P(j); }, i); }      // the modulus r
ZK_HD constexpr uint32_t ONE(int i) { return limb_of([](int j) constexpr { return FrP::ONE(j); }, i); }  // 2^256 mod r
static_assert(FrP::INV == 0xffffffffu, "m = -t mod 2^29 below relies on r = 1 mod 2^32");
static_assert(R(0) == 1u, "r = 1 mod 2^29");
// K*r arranged for carry-free subtraction of NSUB normalised values (see fq28.hpp:sub_const); K >= (sum of the
// subtrahends' bounds in units of r) + 1
struct SubConst {
  uint32_t v[9];
};
ZK_HD constexpr SubConst sub_const(uint32_t K, uint32_t nsub) {
  SubConst c{};
  uint64_t carry = 0;
  for (int i = 0; i < 9; i++) {
    uint64_t t = (uint64_t)R(i) * K + carry;
    c.v[i] = i < 8 ? (uint32_t)(t & MASK) : (uint32_t)t;
    carry = t >> 29;
  }
  for (int i = 0; i < 8; i++) {
    c.v[i] += nsub << 29;
    c.v[i + 1] -= nsub;
  }
  return c;
}
}  // namespace fs29

ZK_HD Fs fs_zero() {
  Fs r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = 0;
  return r;
}
// 8 x 32 bits -> 9 x 29 bits, same integer
ZK_HD Fs fs_unpack(const Fr& a) {
  Fs r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, li = bit >> 5, sh = bit & 31;
    uint32_t v = a.v[li] >> sh;
    if (sh > 3 && li + 1 < 8) v |= a.v[li + 1] << (32 - sh);
    r.v[i] = v & fs29::MASK;
  }
  return r;
}
// normalised, value < 2^257 -> 8 x 32 bits + bit 256
ZK_HD Fr fs_pack(const Fs& a, uint32_t& top) {
  Fr r;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int bit = 32 * j, li = bit / 29, off = bit - 29 * li;
    uint32_t w = a.v[li] >> off;
    if (li + 1 < 9) w |= a.v[li + 1] << (29 - off);
    if (li + 2 < 9 && 58 - off < 32) w |= a.v[li + 2] << (58 - off);
    r.v[j] = w;
  }
  top = a.v[8] >> 24;  // bit 256 = bit 24 of limb 8 (232 + 24)
  return r;
}
// limb-wise sum
ZK_HD Fs fs_add(const Fs& a, const Fs& b) {
  Fs r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a + K r - b, b normalised with value(b) <= (K - 1) r
template <uint32_t K>
ZK_HD Fs fs_sub(const Fs& a, const Fs& b) {
  constexpr fs29::SubConst C = fs29::sub_const(K, 1);
  Fs r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i] + C.v[i];
  return r;
}
// carry the limbs back under 2^29 (value unchanged)
ZK_HD Fs fs_norm(const Fs& a) {
  Fs r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t t = a.v[i] + c;
    r.v[i] = t & fs29::MASK;
    c = t >> 29;
  }
  r.v[8] = a.v[8] + c;
  return r;
}

// Montgomery product (a*b + M r) / 2^256, normalised, < a*b/2^256 + r.  Limbs of a and b < 2^30 (9 * 2^60 + 9 * 2^58 < 2^64).
ZK_HD Fs fs_mul(const Fs& a, const Fs& b) {
  uint32_t m[9], col[10];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fs29::R(k - i);
    m[k] = (0u - (uint32_t)acc) & fs29::MASK;  // -1/r = -1 mod 2^29
    acc += m[k];                               // r's low limb is 1: the low 29 bits are now zero
    acc >>= 29;
  }
  {  // the ninth step only clears 24 bits: 8 * 29 + 24 = 256
#pragma unroll
    for (int i = 0; i <= 8; i++) acc += (uint64_t)a.v[i] * b.v[8 - i];
#pragma unroll
    for (int i = 0; i < 8; i++) acc += (uint64_t)m[i] * fs29::R(8 - i);
    m[8] = (0u - (uint32_t)acc) & 0xffffffu;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(m[8]));  // keep the 24-bit range from the compiler: it would select v_mul_u32_u24 / v_mul_hi_u32_u24 for m[8] * r's top limb
#endif
    acc += m[8];
    col[0] = (uint32_t)acc & fs29::MASK;
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i <= 8; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 8; i <= 8; i++) acc += (uint64_t)m[i] * fs29::R(k - i);
    col[k - 8] = (uint32_t)acc & fs29::MASK;
    acc >>= 29;
  }
  col[9] = (uint32_t)acc;
  Fs t;  // drop the 24 zero bits
#pragma unroll
  for (int j = 0; j < 8; j++) t.v[j] = ((col[j] >> 24) | (col[j + 1] << 5)) & fs29::MASK;
  t.v[8] = (col[8] >> 24) | (col[9] << 5);
  return t;
}

// acc[0..17] += the eighteen 29-bit columns of the integer a * b (a, b normalised).  Seven calls fit between two
// fs_cols_norm.
ZK_HD void fs_mul_raw_acc(uint32_t* acc, const Fs& a, const Fs& b) {
  uint64_t t = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) t += (uint64_t)a.v[i] * b.v[k - i];
    acc[k] += (uint32_t)t & fs29::MASK;
    t >>= 29;
  }
  acc[17] += (uint32_t)t;
}
// The same into 64-bit column accumulators that live ACROSS products: every MAC lands in its column as it is (no cut into 29-bit pieces per product:
// 3 instructions per column per product saved), and the seventeen columns are independent chains.  With normalised operands a column takes at most
// nine products < 2^58 per call: six calls fit between two fs_cols_norm64 (6 * 9 * 2^58 + the carry of a normalised neighbour < 2^64).
ZK_HD void fs_mul_raw_acc64(uint64_t* acc, const Fs& a, const Fs& b) {
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) acc[k] += (uint64_t)a.v[i] * b.v[k - i];
  }
}
// carry the 64-bit columns back under 2^29 (acc[17] keeps the rest)
ZK_HD void fs_cols_norm64(uint64_t* acc) {
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
    const uint64_t t = acc[k] + c;
    acc[k] = t & fs29::MASK;
    c = t >> 29;
  }
  acc[17] += c;
}
// carry the columns back under 2^29 (acc[17] keeps the rest)
ZK_HD void fs_cols_norm(uint32_t* acc) {
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
    uint32_t t = acc[k] + c;
    acc[k] = t & fs29::MASK;
    c = t >> 29;
  }
  acc[17] += c;
}
// T = sum_k cols[k] 2^(29 k) (any 64-bit column values, T < 2^696)  ->  T / 2^256 mod r, canonical.
// T is cut into 232-bit chunks c_j (each a canonical Fr: 232 < 254 bits) and  T/R = sum_j c_j * (2^(232 j) / R)  is three
// ordinary Montgomery products  fr_mul(c_j, 2^(232 j) mod r).
ZK_HD Fr fs_cols_to_fr(const uint64_t* cols) {
  uint32_t limb[24];
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 24; k++) {
    uint64_t t = c + (k < 18 ? cols[k] : 0);
    limb[k] = (uint32_t)t & fs29::MASK;
    c = t >> 29;
  }
  Fr acc = fr_zero();
#pragma unroll
  for (int j = 0; j < 3; j++) {
    Fs ch;
#pragma unroll
    for (int i = 0; i < 8; i++) ch.v[i] = limb[8 * j + i];
    ch.v[8] = 0;
    uint32_t top;
    Fr cj = fs_pack(ch, top);
    Fr kj = fr_zero();
    if (j == 0) kj.v[0] = 1u;
    if (j == 1) kj.v[7] = 0x100u;  // 2^232
    if (j == 2) {                  // 2^464 mod r
      constexpr uint32_t k2[8] = {0x528ebd8du, 0x6ebc8427u, 0x2e5747bau, 0x8a38d0a4u, 0xb00d5ba7u, 0xf98b22b4u, 0x247c4690u, 0x37402f0eu};
#pragma unroll
      for (int i = 0; i < 8; i++) kj.v[i] = k2[i];
    }
    acc = fr_add(acc, fr_mul(cj, kj));
  }
  return acc;
}
// normalised value < 4r -> canonical Fr
ZK_HD Fr fs_to_fr(const Fs& a) {
  uint32_t top;
  Fr t = fs_pack(a, top);
  // subtract 2r, then r, when it fits (nine-word compare: the value can have bit 256 set)
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    Fr d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t bo;
      // limb i of r << (1 - pass)
      const uint32_t ri = pass == 0 ? ((FrP::P(i) << 1) | (i ? FrP::P(i - 1) >> 31 : 0u)) : FrP::P(i);
      d.v[i] = __builtin_subc(t.v[i], ri, bw, &bo);
      bw = bo;
    }
    const uint32_t rtop = pass == 0 ? (FrP::P(7) >> 31) : 0u;  // bit 256 of 2r (zero: r < 2^255)
    const bool ge = top > rtop || (top == rtop && bw == 0);
    if (ge) {
      top = top - rtop - bw;
#pragma unroll
      for (int i = 0; i < 8; i++) t.v[i] = d.v[i];
    }
  }
  return t;
}

}  // namespace zk
