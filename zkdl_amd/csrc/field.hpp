// zkdl_amd/csrc/field.hpp -- BLS12-381 Fr / Fp arithmetic for gfx950 (and for the library's own
// host-side challenge bookkeeping: every function is __host__ __device__).
//
// Representation is the reference's (bls12-381.cuh:116-124,218-226): little-endian 32-bit limbs,
// Montgomery form, R = 2^256 for Fr and 2^384 for Fp, every result fully reduced -- so any
// algebraically equal evaluation order gives bit-identical limbs.  The code is written for CDNA4:
// fully unrolled limb loops (no runtime-indexed limb arrays, which would spill to scratch),
// 32x32->64 products that hipcc lowers to v_mad_u64_u32, and v_add_co/v_addc_co carry chains via
// __builtin_addc/__builtin_subc.  No PTX-style emulation layer, no dual code path.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
// Montgomery products and the G1 group law are real device CALLS, not inlined bodies: one Fp product
// is ~10^3 instructions, a Jacobian add holds 16 of them, and a kernel with a few inlined group
// operations would overflow the 64 KB instruction cache (and take minutes to compile).
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_HD_CALL __host__ __device__ __attribute__((noinline))
#else
#define ZK_HD_CALL __host__ __device__ inline
#endif
#else
#define ZK_HD inline
#define ZK_D inline
#define ZK_HD_CALL inline
#endif

namespace zk {

template <int N>
struct alignas(16) Limbs {
  uint32_t v[N];
};
using Fr = Limbs<8>;   // byte-compatible with Fr_t (fr-tensor.cuh:13)
using Fp = Limbs<12>;  // byte-compatible with Fp_t (g1-tensor.cuh:10)

// Field parameters.  Limb values: bls12-381.cu:3-11; INV = -p^-1 mod 2^32: bls12-381.cuh:119,221.
struct FrP {
  static constexpr int N = 8;
  static constexpr uint32_t INV = 0xffffffffu;
  ZK_HD static constexpr uint32_t P(int i) {
    constexpr uint32_t p[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    return p[i];
  }
  ZK_HD static constexpr uint32_t ONE(int i) {  // R mod r
    constexpr uint32_t o[8] = {4294967294u, 1u, 215042u, 1485092858u, 3971764213u, 2576109551u, 2898593135u, 405057881u};
    return o[i];
  }
  ZK_HD static constexpr uint32_t R2(int i) {  // R^2 mod r
    constexpr uint32_t o[8] = {4092763245u, 3382307216u, 2274516003u, 728559051u, 1918122383u, 97719446u, 2673475345u, 122214873u};
    return o[i];
  }
};
struct FpP {
  static constexpr int N = 12;
  static constexpr uint32_t INV = 4294770685u;
  ZK_HD static constexpr uint32_t P(int i) {
    constexpr uint32_t p[12] = {4294945451u, 3120496639u, 2975072255u, 514588670u, 4138792484u, 1731252896u,
                                4085584575u, 1685539716u, 1129032919u, 1260103606u, 964683418u, 436277738u};
    return p[i];
  }
  ZK_HD static constexpr uint32_t ONE(int i) {
    constexpr uint32_t o[12] = {196605u, 1980301312u, 3289120770u, 3958636555u, 1405573306u, 1598593111u,
                                1884444485u, 2010011731u, 2723605613u, 1543969431u, 4202751123u, 368467651u};
    return o[i];
  }
  ZK_HD static constexpr uint32_t R2(int i) {
    constexpr uint32_t o[12] = {473175878u, 4108263220u, 164693233u, 175564454u, 1284880085u, 2380613484u,
                                2476573632u, 1743489193u, 3038352685u, 2591637125u, 2462770090u, 295210981u};
    return o[i];
  }
};

template <class F>
using El = Limbs<F::N>;

template <class F>
ZK_HD El<F> f_zero() {
  El<F> r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = 0;
  return r;
}
template <class F>
ZK_HD El<F> f_one() {
  El<F> r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = F::ONE(i);
  return r;
}
template <int N>
ZK_HD bool l_is_zero(const Limbs<N>& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; i++) acc |= a.v[i];
  return acc == 0;
}
template <int N>
ZK_HD bool l_eq(const Limbs<N>& a, const Limbs<N>& b) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; i++) acc |= a.v[i] ^ b.v[i];
  return acc == 0;
}
// a >= b, most significant limb first (bls12-381.cu:245-253): computed as "a - b does not borrow"
template <int N>
ZK_HD bool l_gte(const Limbs<N>& a, const Limbs<N>& b) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t bo;
    (void)__builtin_subc(a.v[i], b.v[i], bw, &bo);
    bw = bo;
  }
  return bw == 0;
}

// r = t - p if t >= p else t  (the closing step of add / double / mul in the reference)
template <class F>
ZK_HD El<F> f_reduce_once(const El<F>& t) {
  El<F> d;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) {
    uint32_t bo;
    d.v[i] = __builtin_subc(t.v[i], F::P(i), bw, &bo);
    bw = bo;
  }
  El<F> r;
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = bw ? t.v[i] : d.v[i];
  return r;
}

// bls12-381.cu:296-300 / :703-707
template <class F>
ZK_HD El<F> f_add(const El<F>& a, const El<F>& b) {
  El<F> s;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) {
    uint32_t co;
    s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co);
    c = co;
  }
  return f_reduce_once<F>(s);
}
// bls12-381.cu:289-293 / :696-700
template <class F>
ZK_HD El<F> f_sub(const El<F>& a, const El<F>& b) {
  El<F> d;
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) {
    uint32_t bo;
    d.v[i] = __builtin_subc(a.v[i], b.v[i], bw, &bo);
    bw = bo;
  }
  // a < b  <=>  the subtraction borrowed: add p back (wrapping)
  El<F> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < F::N; i++) {
    uint32_t co;
    r.v[i] = __builtin_addc(d.v[i], bw ? F::P(i) : 0u, c, &co);
    c = co;
  }
  return r;
}
template <class F>
ZK_HD El<F> f_neg(const El<F>& a) {
  return f_sub<F>(f_zero<F>(), a);
}
// bls12-381.cu:550-556 / :957-963
template <class F>
ZK_HD El<F> f_double(const El<F>& a) {
  El<F> s;
#pragma unroll
  for (int i = F::N - 1; i >= 1; i--) s.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 31);
  s.v[0] = a.v[0] << 1;
  return f_reduce_once<F>(s);
}

// CIOS Montgomery product a*b*R^-1 mod p, one conditional subtract (bls12-381.cu:499-530 /
// :906-937).  Kept in the general form (t[N], t[N+1] overflow limbs) so non-canonical operands
// give the reference's limbs too.
template <class F>
ZK_HD El<F> f_mul(const El<F>& a, const El<F>& b) {
  constexpr int N = F::N;
  uint32_t t[N + 2];
#pragma unroll
  for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
      t[j] = (uint32_t)s;
      c = s >> 32;
    }
    uint64_t s = (uint64_t)t[N] + c;
    t[N] = (uint32_t)s;
    t[N + 1] = (uint32_t)(s >> 32);
    uint32_t m = t[0] * F::INV;
    s = (uint64_t)m * F::P(0) + t[0];
    c = s >> 32;
#pragma unroll
    for (int j = 1; j < N; j++) {
      s = (uint64_t)m * F::P(j) + t[j] + c;
      t[j - 1] = (uint32_t)s;
      c = s >> 32;
    }
    s = (uint64_t)t[N] + c;
    t[N - 1] = (uint32_t)s;
    t[N] = t[N + 1] + (uint32_t)(s >> 32);
  }
  El<F> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = t[i];
  return f_reduce_once<F>(r);
}
#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 Montgomery product: product scanning (FIPS) with a 96-bit column accumulator.
// Measured on MI355X (tools/instr_rate.hip): v_mad_u64_u32 issues every ~4.6 cycles per wave, the
// same as a carry add, a 64-bit add or a VOP3 move-class op -- so the cost of a big-integer multiply
// is its INSTRUCTION COUNT, not its multiplies.  Operand-scanning CIOS compiles to 1 MAD + 1 64-bit
// add + ~2.5 moves per partial product; here every partial product is exactly
//     v_mad_u64_u32  acc[63:0], vcc, x, y, acc[63:0]      (64-bit accumulate, carry out in vcc)
//     v_addc_co_u32  acc[95:64], vcc, 0, acc[95:64], vcc
// and a column shift is three moves.  The value computed is (a*b + m*p) / R with the same m as
// CIOS, so limbs (including the discarded overflow limb) are identical to the reference's.
#define ZK_MAC_VV(lo, hi, x, y) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc")
#define ZK_MAC_VS(lo, hi, x, sy) \
  asm("v_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "s"(sy) : "vcc")

template <class F>
__device__ __forceinline__ El<F> f_mul_dev(const El<F>& a, const El<F>& b) {
  constexpr int N = F::N;
  uint32_t m[N];
  El<F> t;
  uint64_t lo = 0;
  uint32_t hi = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) ZK_MAC_VV(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) ZK_MAC_VS(lo, hi, m[i], F::P(k - i));
    m[k] = (uint32_t)lo * F::INV;
    ZK_MAC_VS(lo, hi, m[k], F::P(0));  // clears the low word
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
  }
#pragma unroll
  for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
    for (int i = k - N + 1; i < N; i++) ZK_MAC_VV(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
    for (int i = k - N + 1; i < N; i++) ZK_MAC_VS(lo, hi, m[i], F::P(k - i));
    t.v[k - N] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
  }
  t.v[N - 1] = (uint32_t)lo;  // (lo >> 32) is the overflow limb the reference drops as well
  return f_reduce_once<F>(t);
}
#endif

template <class F>
ZK_HD El<F> f_sqr(const El<F>& a) {
  return f_mul<F>(a, a);
}
// bls12-381.cu:585-593
template <class F>
ZK_HD El<F> f_mont(const El<F>& a) {
  El<F> r2;
#pragma unroll
  for (int i = 0; i < F::N; i++) r2.v[i] = F::R2(i);
  return f_mul<F>(a, r2);
}
template <class F>
ZK_HD El<F> f_unmont(const El<F>& a) {
  El<F> one = f_zero<F>();
  one.v[0] = 1;
  return f_mul<F>(a, one);
}

// Short names used throughout the kernels.
ZK_HD Fr fr_add(const Fr& a, const Fr& b) { return f_add<FrP>(a, b); }
ZK_HD Fr fr_sub(const Fr& a, const Fr& b) { return f_sub<FrP>(a, b); }
// Fr products are inlined into the streaming kernels (~330 instructions each).
ZK_HD Fr fr_mul(const Fr& a, const Fr& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return f_mul_dev<FrP>(a, b);
#else
  return f_mul<FrP>(a, b);
#endif
}
ZK_HD Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }
ZK_HD Fr fr_neg(const Fr& a) { return f_neg<FrP>(a); }
ZK_HD Fr fr_dbl(const Fr& a) { return f_double<FrP>(a); }
ZK_HD Fr fr_mont(const Fr& a) {
  Fr r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.v[i] = FrP::R2(i);
  return fr_mul(a, r2);
}
ZK_HD Fr fr_unmont(const Fr& a) {
  Fr one = f_zero<FrP>();
  one.v[0] = 1;
  return fr_mul(a, one);
}
ZK_HD Fr fr_zero() { return f_zero<FrP>(); }
ZK_HD Fr fr_one() { return f_one<FrP>(); }
ZK_HD Fp fp_add(const Fp& a, const Fp& b) { return f_add<FpP>(a, b); }
ZK_HD Fp fp_sub(const Fp& a, const Fp& b) { return f_sub<FpP>(a, b); }
// The Fp product is a real device CALL (a Jacobian add holds 16 of them) whose 24 operand limbs
// travel in VGPRs: clang passes non-aggregate vector arguments directly, whereas two 48-byte structs
// would exceed its 16-register aggregate budget and go through scratch memory.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t zk_u32x4 __attribute__((ext_vector_type(4)));
static __device__ __attribute__((noinline)) Fp fp_mul_regs(zk_u32x4 a0, zk_u32x4 a1, zk_u32x4 a2, zk_u32x4 b0, zk_u32x4 b1, zk_u32x4 b2) {
  Fp a, b;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a.v[i] = a0[i]; a.v[4 + i] = a1[i]; a.v[8 + i] = a2[i];
    b.v[i] = b0[i]; b.v[4 + i] = b1[i]; b.v[8 + i] = b2[i];
  }
  return f_mul_dev<FpP>(a, b);
}
#endif
ZK_HD Fp fp_mul(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  zk_u32x4 a0 = {a.v[0], a.v[1], a.v[2], a.v[3]}, a1 = {a.v[4], a.v[5], a.v[6], a.v[7]}, a2 = {a.v[8], a.v[9], a.v[10], a.v[11]};
  zk_u32x4 b0 = {b.v[0], b.v[1], b.v[2], b.v[3]}, b1 = {b.v[4], b.v[5], b.v[6], b.v[7]}, b2 = {b.v[8], b.v[9], b.v[10], b.v[11]};
  return fp_mul_regs(a0, a1, a2, b0, b1, b2);
#else
  return f_mul<FpP>(a, b);
#endif
}
ZK_HD Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }
ZK_HD Fp fp_neg(const Fp& a) { return f_neg<FpP>(a); }
ZK_HD Fp fp_dbl(const Fp& a) { return f_double<FpP>(a); }
ZK_HD Fp fp_zero() { return f_zero<FpP>(); }
ZK_HD Fp fp_one() { return f_one<FpP>(); }

// a^(p-2) in the Montgomery domain (Fermat inverse): 380 dependent squarings.  Kept as the independent cross-check of fp_inv
// (fpinv.hpp: binary GCD, ~9x fewer instructions), which is what backs the affine-canonical form that defines G1 parity.
ZK_HD Fp fp_inv_fermat(const Fp& a) {
  Fp acc = fp_one();
  // exponent p - 2, MSB first; p's low limb is 0xffffaaab so p-2 only changes limb 0
  for (int i = 12 * 32 - 1; i >= 0; i--) {
    uint32_t limb = FpP::P(i >> 5);
    if ((i >> 5) == 0) limb -= 2u;
    acc = fp_sqr(acc);
    if ((limb >> (i & 31)) & 1u) acc = fp_mul(acc, a);
  }
  return acc;
}

}  // namespace zk
