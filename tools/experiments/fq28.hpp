// zkdl_amd/csrc/fq28.hpp -- bls12-381 Fp on 14 limbs of 28 bits, for use INSIDE the MSM pipeline (device only).
//
// With 32-bit limbs a column of the Montgomery product overflows 64 bits after two partial products, so every
// v_mad_u64_u32 needs a carry add behind it (field.hpp: 288 MACs + 288 carry adds, ~700 instructions per Fp
// product), and on gfx950 a carry add issues as slowly as the MAC (tools/mad_probe.hip: 5.0 cycles per
// wave-MAD alone, 9.2 with its carry add).  With 28-bit limbs a column of <= 28 products of < 2^56 fits 64 bits:
// a MAC is ONE instruction.  392 MACs + ~80 shifts/masks = ~470 instructions, 29 % faster per product
// (tools/fq_probe.hip: 77.9 vs 60.5 G products/s).  Additions and subtractions cost more here (carries travel by
// shift and mask), but a mixed addition is 10 products against ~7 additions.
//
// Values are kept canonical ([0, p), limbs < 2^28), Montgomery radix R' = 2^392.  Nothing in this form leaves the
// MSM: window tables are converted when they are built (x*2^384 -> x*2^392 is a multiplication by 2^8), results
// are converted back before they are written (the reference's 32-bit Montgomery form).
#pragma once
#include "field.hpp"
#include "g1.hpp"

namespace zk {

typedef uint32_t fq_u32x4 __attribute__((ext_vector_type(4)));
struct Fq {
  uint32_t v[14];
};
static constexpr uint32_t FQ_M28 = 0xfffffffu;
static constexpr uint32_t FQ_INV = 0xffcfffdu;  // -p^-1 mod 2^28
__device__ __forceinline__ constexpr uint32_t fq_p(int i) {
  constexpr uint32_t p[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                              0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x1a011u};
  return p[i];
}
__device__ __forceinline__ constexpr uint32_t fq_one_limb(int i) {  // 2^392 mod p
  constexpr uint32_t o[14] = {0x347fcb8u, 0xd800000u, 0x2b119u, 0xcde6d2u, 0xc7212e0u, 0x83a2090u, 0x37669fu,
                              0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x12ca7cu, 0x659fcfau, 0x577au};
  return o[i];
}
__device__ __forceinline__ Fq fq_zero() {
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = 0;
  return r;
}
__device__ __forceinline__ Fq fq_one() {
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = fq_one_limb(i);
  return r;
}
__device__ __forceinline__ bool fq_is_zero(const Fq& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) acc |= a.v[i];
  return acc == 0;
}
// t in [0, 2p) with normalised limbs -> [0, p)
__device__ __forceinline__ Fq fq_reduce_once(const Fq& t) {
  Fq d;
  int32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    int32_t x = (int32_t)t.v[i] - (int32_t)fq_p(i) - bw;
    d.v[i] = (uint32_t)x & FQ_M28;
    bw = (x >> 31) & 1;
  }
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = bw ? t.v[i] : d.v[i];
  return r;
}
__device__ __forceinline__ Fq fq_add(const Fq& a, const Fq& b) {
  Fq s;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    uint32_t x = a.v[i] + b.v[i] + c;
    s.v[i] = i < 13 ? (x & FQ_M28) : x;
    c = x >> 28;
  }
  return fq_reduce_once(s);
}
__device__ __forceinline__ Fq fq_dbl(const Fq& a) { return fq_add(a, a); }
__device__ __forceinline__ Fq fq_sub(const Fq& a, const Fq& b) {
  Fq d;
  int32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    int32_t x = (int32_t)a.v[i] - (int32_t)b.v[i] - bw;
    d.v[i] = (uint32_t)x & FQ_M28;
    bw = (x >> 31) & 1;
  }
  // a < b: the limbs hold a - b + 2^392; adding p and dropping the carry out of the top limb gives a - b + p
  Fq e;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    uint32_t x = d.v[i] + fq_p(i) + c;
    e.v[i] = x & FQ_M28;
    c = x >> 28;
  }
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) r.v[i] = bw ? e.v[i] : d.v[i];
  return r;
}
__device__ __forceinline__ Fq fq_neg(const Fq& a) { return fq_sub(fq_zero(), a); }

// Montgomery product, product scanning with ONE 64-bit column accumulator (no carry adds)
__device__ __forceinline__ Fq fq_mul_inl(const Fq& a, const Fq& b) {
  uint32_t m[14];
  Fq t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 14; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * fq_p(k - i);
    m[k] = ((uint32_t)acc * FQ_INV) & FQ_M28;
    acc += (uint64_t)m[k] * fq_p(0);
    acc >>= 28;
  }
#pragma unroll
  for (int k = 14; k < 27; k++) {
#pragma unroll
    for (int i = k - 13; i < 14; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * fq_p(k - i);
    t.v[k - 14] = (uint32_t)acc & FQ_M28;
    acc >>= 28;
  }
  t.v[13] = (uint32_t)acc;
  return fq_reduce_once(t);
}
// a real device CALL with the 28 operand limbs in VGPRs (cf. fp_mul_regs in field.hpp)
static __device__ __attribute__((noinline)) Fq fq_mul_regs(fq_u32x4 a0, fq_u32x4 a1, fq_u32x4 a2, fq_u32x4 a3, fq_u32x4 b0, fq_u32x4 b1, fq_u32x4 b2, fq_u32x4 b3) {
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
__device__ __forceinline__ Fq fq_mul(const Fq& a, const Fq& b) {
  fq_u32x4 a0 = {a.v[0], a.v[1], a.v[2], a.v[3]}, a1 = {a.v[4], a.v[5], a.v[6], a.v[7]}, a2 = {a.v[8], a.v[9], a.v[10], a.v[11]}, a3 = {a.v[12], a.v[13], 0u, 0u};
  fq_u32x4 b0 = {b.v[0], b.v[1], b.v[2], b.v[3]}, b1 = {b.v[4], b.v[5], b.v[6], b.v[7]}, b2 = {b.v[8], b.v[9], b.v[10], b.v[11]}, b3 = {b.v[12], b.v[13], 0u, 0u};
  return fq_mul_regs(a0, a1, a2, a3, b0, b1, b2, b3);
}
__device__ __forceinline__ Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }

// ---- conversions: the same residue, 32-bit Montgomery form (x * 2^384) <-> 28-bit form (x * 2^392)
__device__ __forceinline__ Fq fq_from_fp(const Fp& a32) {
  Fp c;  // 2^8 in 32-bit Montgomery form: fp_mul(a, c) = a * 2^8 mod p
  constexpr uint32_t cl[12] = {55049400u, 433586176u, 1830814385u, 316673246u, 2718502002u, 929472387u, 3658445792u, 162569026u, 2400360379u, 2814711257u, 4244242732u, 91727449u};
#pragma unroll
  for (int i = 0; i < 12; i++) c.v[i] = cl[i];
  Fp t = fp_mul(a32, c);
  Fq r;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    const int bit = 28 * i, w = bit >> 5, sh = bit & 31;
    uint32_t lo = t.v[w] >> sh;
    uint32_t hi = (sh > 4 && w + 1 < 12) ? (t.v[w + 1] << (32 - sh)) : 0u;
    r.v[i] = (lo | hi) & FQ_M28;
  }
  return r;
}
__device__ __forceinline__ Fp fq_to_fp(const Fq& a) {
  Fp t;
#pragma unroll
  for (int j = 0; j < 12; j++) {  // bits [32j, 32j+32) of the 392-bit limb string (value < p < 2^381)
    const int bit = 32 * j, l = bit / 28, sh = bit % 28;
    uint32_t x = a.v[l] >> sh;                         // 28 - sh bits
    if (l + 1 < 14) x |= a.v[l + 1] << (28 - sh);      // next 28 bits
    if (sh > 24 && l + 2 < 14) x |= a.v[l + 2] << (56 - sh);
    t.v[j] = x;
  }
  Fp d = fp_zero();  // 2^-8 in 32-bit Montgomery form is 2^376
  d.v[11] = 16777216u;
  return fp_mul(t, d);
}

// ---- G1 in extended Jacobian (XYZZ) coordinates over Fq (formulas as msm.hip's 32-bit G1X)
struct alignas(16) G1Aq {
  Fq x, y;
};
struct alignas(16) G1Xq {
  Fq x, y, zz, zzz;
};
__device__ __forceinline__ G1Xq g1xq_zero() {
  G1Xq r;
  r.x = fq_zero();
  r.y = fq_one();
  r.zz = fq_zero();
  r.zzz = fq_zero();
  return r;
}
__device__ __forceinline__ G1Xq g1xq_double(const G1Xq& a) {
  if (fq_is_zero(a.zz)) return a;
  Fq u = fq_dbl(a.y);
  Fq v = fq_sqr(u);
  Fq w = fq_mul(u, v);
  Fq s_ = fq_mul(a.x, v);
  Fq xx = fq_sqr(a.x);
  Fq m = fq_add(fq_dbl(xx), xx);
  G1Xq o;
  o.x = fq_sub(fq_sub(fq_sqr(m), s_), s_);
  o.y = fq_sub(fq_mul(m, fq_sub(s_, o.x)), fq_mul(w, a.y));
  o.zz = fq_mul(v, a.zz);
  o.zzz = fq_mul(w, a.zzz);
  return o;
}
__device__ __forceinline__ G1Xq g1xq_madd(const G1Xq& a, const G1Aq& b) {
  if (fq_is_zero(a.zz)) {
    G1Xq o;
    o.x = b.x;
    o.y = b.y;
    o.zz = fq_one();
    o.zzz = fq_one();
    return o;
  }
  Fq u2 = fq_mul(b.x, a.zz);
  Fq s2 = fq_mul(b.y, a.zzz);
  Fq pd = fq_sub(u2, a.x);
  Fq r = fq_sub(s2, a.y);
  if (fq_is_zero(pd)) return fq_is_zero(r) ? g1xq_double(a) : g1xq_zero();
  Fq pp = fq_sqr(pd);
  Fq ppp = fq_mul(pd, pp);
  Fq q = fq_mul(a.x, pp);
  G1Xq o;
  o.x = fq_sub(fq_sub(fq_sub(fq_sqr(r), ppp), q), q);
  o.y = fq_sub(fq_mul(r, fq_sub(q, o.x)), fq_mul(a.y, ppp));
  o.zz = fq_mul(a.zz, pp);
  o.zzz = fq_mul(a.zzz, ppp);
  return o;
}
__device__ __forceinline__ G1Xq g1xq_add(const G1Xq& a, const G1Xq& b) {
  if (fq_is_zero(a.zz)) return b;
  if (fq_is_zero(b.zz)) return a;
  Fq u1 = fq_mul(a.x, b.zz);
  Fq u2 = fq_mul(b.x, a.zz);
  Fq s1 = fq_mul(a.y, b.zzz);
  Fq s2 = fq_mul(b.y, a.zzz);
  Fq pd = fq_sub(u2, u1);
  Fq r = fq_sub(s2, s1);
  if (fq_is_zero(pd)) return fq_is_zero(r) ? g1xq_double(a) : g1xq_zero();
  Fq pp = fq_sqr(pd);
  Fq ppp = fq_mul(pd, pp);
  Fq q = fq_mul(u1, pp);
  G1Xq o;
  o.x = fq_sub(fq_sub(fq_sub(fq_sqr(r), ppp), q), q);
  o.y = fq_sub(fq_mul(r, fq_sub(q, o.x)), fq_mul(s1, ppp));
  o.zz = fq_mul(fq_mul(a.zz, b.zz), pp);
  o.zzz = fq_mul(fq_mul(a.zzz, b.zzz), ppp);
  return o;
}
__device__ __forceinline__ G1J g1xq_to_jacobian(const G1Xq& a) {
  if (fq_is_zero(a.zz)) return g1_zero();
  Fq z = fq_mul(a.zz, a.zzz);
  Fq t = fq_mul(a.zz, fq_sqr(a.zzz));
  G1J o;
  o.x = fq_to_fp(fq_mul(a.x, t));
  o.y = fq_to_fp(fq_mul(a.y, fq_mul(t, fq_sqr(a.zz))));
  o.z = fq_to_fp(z);
  return o;
}
__device__ __forceinline__ G1Xq g1xq_shfl_xor(const G1Xq& p, int mask) {
  G1Xq r;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    r.x.v[i] = __shfl_xor(p.x.v[i], mask, 64);
    r.y.v[i] = __shfl_xor(p.y.v[i], mask, 64);
    r.zz.v[i] = __shfl_xor(p.zz.v[i], mask, 64);
    r.zzz.v[i] = __shfl_xor(p.zzz.v[i], mask, 64);
  }
  return r;
}

}  // namespace zk
