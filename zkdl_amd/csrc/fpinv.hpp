// zkdl_amd/csrc/fpinv.hpp -- modular inversion in the base field by an optimised binary GCD, for the affine-canonical form.
//
// The canonical form of a G1 point (x = X/Z^2, y = Y/Z^3: the build's definition of G1 parity, the reference never normalises) needs one
// inversion per point.  The Fermat inverse a^(p-2) is a chain of 380 dependent squarings + ~100 products on ONE lane (~0.5 ms: the tail of
// every proof's G1 phase, g1_canonicalise); nothing can shorten that chain.  The binary extended GCD needs no field products at all: the
// variant below (T. Pornin, "Optimized Binary GCD for Modular Inversion", 2020) runs the 2 * 381 - 1 = 761 halving steps in 25 groups of 31,
// each group on 64-bit APPROXIMATIONS of (a, b) -- their low 31 bits, which decide every parity test exactly, and their top 33 bits, which
// decide the comparisons -- while recording the group's effect as a 2 x 2 matrix of 32-bit factors, and then applies the matrix once to the
// full-width values (a, b) and, modulo p, to the Bezout coefficients (u, v).  Per group: 31 steps of ~25 scalar-width instructions and ten
// 12-limb multiply-accumulate rows, ~1.3 k instructions, ~33 k for an inversion against ~290 k for the Fermat chain.  Every lane runs the
// same 25 x 31 steps whatever its input (selects, no data-dependent branches).
//
// Invariants (m = p, y the input):  a = y u (mod m),  b = y v (mod m).  Start a = y, b = m, u = 1, v = 0; one step is
//     a even:            a <- a / 2,        u <- u / 2
//     a odd (a >= b):    a <- (a - b) / 2,  u <- (u - v) / 2        (with (a, u) <-> (b, v) first when a < b)
// and after 761 steps a = 0, b = gcd(y, m) = 1, v = 1 / y.  A group's factors satisfy  2^31 a' = f0 a + g0 b,  2^31 b' = f1 a + g1 b  with
// |f0| + |g0| <= 2^31, |f1| + |g1| <= 2^31; a comparison decided on the approximations may be wrong when a and b agree in their top 33 bits,
// in which case a' comes out negative and is negated together with its factors (the paper's correction), and the step count still suffices.
// The inverse of 0 is 0 (the Fermat chain's value).  The result is the unique inverse in [0, p): bit-identical to a^(p-2).
#pragma once
#include "field.hpp"

namespace zk {
namespace fpinv {

constexpr int N = 12;  // 32-bit limbs of p

// r (N + 2 limbs, two's complement) = sf * (x * |f|) + sg * (y * |g|),  x, y < 2^384 unsigned, |f|, |g| <= 2^31 with signs sf, sg
ZK_HD void lin2(uint32_t (&r)[N + 2], const uint32_t (&x)[N], int64_t f, const uint32_t (&y)[N], int64_t g) {
  const uint32_t fa = (uint32_t)(f < 0 ? -f : f), ga = (uint32_t)(g < 0 ? -g : g);  // magnitudes <= 2^31 fit 32 bits
  uint32_t p[N + 2], q[N + 2];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    c += (uint64_t)x[i] * fa;
    p[i] = (uint32_t)c;
    c >>= 32;
  }
  p[N] = (uint32_t)c;
  p[N + 1] = 0;
  c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    c += (uint64_t)y[i] * ga;
    q[i] = (uint32_t)c;
    c >>= 32;
  }
  q[N] = (uint32_t)c;
  q[N + 1] = 0;
  // conditional negation (two's complement) of each product, then the sum
  const uint32_t mf = f < 0 ? 0xffffffffu : 0u, mg = g < 0 ? 0xffffffffu : 0u;
  uint32_t cf = mf & 1u, cg = mg & 1u, cs = 0;
#pragma unroll
  for (int i = 0; i < N + 2; i++) {
    uint32_t co;
    const uint32_t pi = __builtin_addc(p[i] ^ mf, 0u, cf, &co);
    cf = co;
    const uint32_t qi = __builtin_addc(q[i] ^ mg, 0u, cg, &co);
    cg = co;
    r[i] = __builtin_addc(pi, qi, cs, &co);
    cs = co;
  }
}
// |x / 2^31| for a two's complement (N + 2)-limb x that is divisible by 2^31 and whose quotient fits N limbs in magnitude; returns the sign
ZK_HD bool shr31_abs(uint32_t (&out)[N], const uint32_t (&x)[N + 2]) {
  const bool neg = (x[N + 1] >> 31) != 0;
  const uint32_t m = neg ? 0xffffffffu : 0u;
  uint32_t t[N + 1];
#pragma unroll
  for (int i = 0; i <= N; i++) t[i] = (x[i] >> 31) | (x[i + 1] << 1);
  uint32_t c = m & 1u;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t co;
    out[i] = __builtin_addc(t[i] ^ m, 0u, c, &co);
    c = co;
  }
  return neg;
}
// (x + q p) / 2^31 reduced into [0, p), x two's complement (N + 2 limbs) with |x| <= 2^31 p, q chosen so that the sum is divisible by 2^31
ZK_HD void redc31(uint32_t (&out)[N], const uint32_t (&x)[N + 2]) {
  const uint32_t q = (x[0] * FpP::INV) & 0x7fffffffu;  // INV = -1/p mod 2^32
  uint32_t s[N + 2];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    c += (uint64_t)FpP::P(i) * q + x[i];
    s[i] = (uint32_t)c;
    c >>= 32;
  }
  c += x[N];
  s[N] = (uint32_t)c;
  c >>= 32;
  s[N + 1] = x[N + 1] + (uint32_t)c;
  // t = s >> 31 (arithmetic): in (-p, 2p), N + 1 limbs incl. the sign limb
  uint32_t t[N + 1];
#pragma unroll
  for (int i = 0; i <= N; i++) t[i] = (s[i] >> 31) | (s[i + 1] << 1);
  // negative: add p
  const uint32_t neg = (t[N] >> 31) ? 0xffffffffu : 0u;
  uint32_t ca = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t co;
    t[i] = __builtin_addc(t[i], FpP::P(i) & neg, ca, &co);
    ca = co;
  }
  t[N] += ca;  // now >= 0: t[N] is 0 or 1... (< 2p < 2^382: t[N] == 0)
  // >= p: subtract p
  uint32_t d[N], bw = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint32_t bo;
    d[i] = __builtin_subc(t[i], FpP::P(i), bw, &bo);
    bw = bo;
  }
#pragma unroll
  for (int i = 0; i < N; i++) out[i] = bw ? t[i] : d[i];
}

}  // namespace fpinv

// y^-1 mod p for the INTEGER y < p given as 12 x 32-bit limbs (no Montgomery factor involved); 0 -> 0
ZK_HD Fp fp_inv_int(const Fp& y) {
  using namespace fpinv;
  uint32_t a[N], b[N], u[N], v[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    a[i] = y.v[i];
    b[i] = FpP::P(i);
    u[i] = i == 0 ? 1u : 0u;
    v[i] = 0u;
  }
#pragma unroll 1
  for (int it = 0; it < 25; it++) {
    // ---- 64-bit approximations: the 64 bits below the top bit of max(a, b) (the two low limbs as they are once both fit 64 bits)
    uint32_t ah = a[1], am = a[0], al = 0, bh = b[1], bm = b[0], bl = 0;
    bool found = false;
#pragma unroll
    for (int i = N - 1; i >= 2; i--) {
      const bool take = !found && ((a[i] | b[i]) != 0u);
      if (take) {
        ah = a[i]; am = a[i - 1]; al = a[i - 2];
        bh = b[i]; bm = b[i - 1]; bl = b[i - 2];
      }
      found = found || take;
    }
    const uint32_t top = ah | bh;
    const int s = (found && top) ? __builtin_clz(top) : 0;
    uint64_t ta = ((uint64_t)ah << 32) | am, tb = ((uint64_t)bh << 32) | bm;
    if (s) {
      ta = (ta << s) | (al >> (32 - s));
      tb = (tb << s) | (bl >> (32 - s));
    }
    uint64_t xa = (ta & ~0x7fffffffull) | (a[0] & 0x7fffffffu), xb = (tb & ~0x7fffffffull) | (b[0] & 0x7fffffffu);
    // ---- 31 steps on the approximations
    int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
    for (int j = 0; j < 31; j++) {
      const uint64_t odd = 0ull - (xa & 1ull);
      const uint64_t sw = odd & (xa < xb ? ~0ull : 0ull);
      const uint64_t tx = (xa ^ xb) & sw;
      xa ^= tx;
      xb ^= tx;
      const int64_t tf = (f0 ^ f1) & (int64_t)sw, tg = (g0 ^ g1) & (int64_t)sw;
      f0 ^= tf; f1 ^= tf;
      g0 ^= tg; g1 ^= tg;
      xa -= xb & odd;
      f0 -= f1 & (int64_t)odd;
      g0 -= g1 & (int64_t)odd;
      xa >>= 1;
      f1 <<= 1;
      g1 <<= 1;
    }
    // ---- apply the factors to (a, b) ...
    uint32_t ra[N + 2], rb[N + 2];
    lin2(ra, a, f0, b, g0);
    lin2(rb, a, f1, b, g1);
    const bool na = shr31_abs(a, ra), nb = shr31_abs(b, rb);
    if (na) { f0 = -f0; g0 = -g0; }
    if (nb) { f1 = -f1; g1 = -g1; }
    // ---- ... and, modulo p, to (u, v)
    uint32_t ru[N + 2], rv[N + 2];
    lin2(ru, u, f0, v, g0);
    lin2(rv, u, f1, v, g1);
    redc31(u, ru);
    redc31(v, rv);
  }
  // a = 0, b = gcd: 1 for y != 0 (p is prime), p for y = 0
  bool b_is_one = b[0] == 1u;
#pragma unroll
  for (int i = 1; i < N; i++) b_is_one = b_is_one && b[i] == 0u;
  Fp r;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = b_is_one ? v[i] : 0u;
  return r;
}

// the inverse in the library's Montgomery domain (R = 2^384): a = x R  ->  x^-1 R.   (x R)^-1 = x^-1 R^-1 as integers mod p; one Montgomery
// product by R^3 mod p restores the factor:  (x^-1 R^-1)(R^3) / R = x^-1 R.
ZK_HD Fp fp_inv_bingcd(const Fp& a) {
  constexpr uint32_t r3[12] = {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au,
                               0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u};  // 2^1152 mod p
  Fp c;
#pragma unroll
  for (int i = 0; i < 12; i++) c.v[i] = r3[i];
  return fp_mul(fp_inv_int(a), c);
}

ZK_HD Fp fp_inv(const Fp& a) { return fp_inv_bingcd(a); }

}  // namespace zk
