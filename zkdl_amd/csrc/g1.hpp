// zkdl_amd/csrc/g1.hpp -- BLS12-381 G1 group law in Jacobian coordinates for gfx950.
//
// Formulas and special-case branches follow what the reference executes:
// dbl-2009-l (bls12-381.cu:1332-1357), add-2007-bl (:1400-1435), madd-2007-bl (:1360-1397), so a
// kernel that keeps the reference's association order reproduces its (X, Y, Z) limbs exactly;
// kernels that re-associate (Pippenger, flattened openings) are compared in affine-canonical
// form instead.  Infinity is Z == 0; the canonical infinity is {0, ONE, 0} (bls12-381.cuh:417).
#pragma once
#include "field.hpp"

namespace zk {

struct alignas(16) G1A {  // byte-compatible with G1Affine_t
  Fp x, y;
};
struct alignas(16) G1J {  // byte-compatible with G1Jacobian_t
  Fp x, y, z;
};

ZK_HD G1J g1_zero() {
  G1J r;
  r.x = fp_zero();
  r.y = fp_one();
  r.z = fp_zero();
  return r;
}
ZK_HD bool g1_is_inf(const G1J& a) { return l_is_zero(a.z); }

ZK_HD G1J g1_double(const G1J& q) {
  if (l_is_zero(q.z)) return q;
  Fp a = fp_sqr(q.x);
  Fp b = fp_sqr(q.y);
  Fp c = fp_sqr(b);
  Fp d = fp_add(q.x, b);
  d = fp_sqr(d);
  d = fp_sub(fp_sub(d, a), c);
  d = fp_dbl(d);
  Fp e = fp_add(fp_dbl(a), a);
  Fp f = fp_sqr(e);
  G1J o;
  o.z = fp_dbl(fp_mul(q.y, q.z));
  o.x = fp_sub(fp_sub(f, d), d);
  c = fp_dbl(fp_dbl(fp_dbl(c)));
  o.y = fp_sub(fp_mul(fp_sub(d, o.x), e), c);
  return o;
}

ZK_HD G1J g1_add(const G1J& a, const G1J& b) {
  if (l_is_zero(a.z)) return b;
  if (l_is_zero(b.z)) return a;
  Fp z1z1 = fp_sqr(a.z);
  Fp z2z2 = fp_sqr(b.z);
  Fp u1 = fp_mul(a.x, z2z2);
  Fp u2 = fp_mul(b.x, z1z1);
  Fp s1 = fp_mul(fp_mul(a.y, b.z), z2z2);
  Fp s2 = fp_mul(fp_mul(b.y, a.z), z1z1);
  if (l_eq(u1, u2) && l_eq(s1, s2)) return g1_double(a);
  Fp h = fp_sub(u2, u1);
  Fp i = fp_sqr(fp_dbl(h));
  Fp j = fp_mul(h, i);
  Fp r = fp_dbl(fp_sub(s2, s1));
  Fp v = fp_mul(u1, i);
  G1J o;
  o.x = fp_sub(fp_sub(fp_sub(fp_sqr(r), j), v), v);
  o.y = fp_mul(fp_sub(v, o.x), r);
  s1 = fp_dbl(fp_mul(s1, j));
  o.y = fp_sub(o.y, s1);
  o.z = fp_sqr(fp_add(a.z, b.z));
  o.z = fp_sub(fp_sub(o.z, z1z1), z2z2);
  o.z = fp_mul(o.z, h);
  return o;
}

ZK_HD G1J g1_add_mixed(const G1J& a, const G1A& b) {
  if (l_is_zero(a.z)) {
    G1J o;
    o.x = b.x;
    o.y = b.y;
    o.z = fp_one();
    return o;
  }
  Fp z1z1 = fp_sqr(a.z);
  Fp u2 = fp_mul(b.x, z1z1);
  Fp s2 = fp_mul(fp_mul(b.y, a.z), z1z1);
  if (l_eq(a.x, u2) && l_eq(a.y, s2)) return g1_double(a);
  Fp h = fp_sub(u2, a.x);
  Fp hh = fp_sqr(h);
  Fp i = fp_dbl(fp_dbl(hh));
  Fp j = fp_mul(h, i);
  Fp r = fp_dbl(fp_sub(s2, a.y));
  Fp v = fp_mul(a.x, i);
  G1J o;
  o.x = fp_sub(fp_sub(fp_sqr(r), j), fp_dbl(v));
  j = fp_dbl(fp_mul(a.y, j));
  o.y = fp_sub(fp_mul(fp_sub(v, o.x), r), j);
  o.z = fp_add(a.z, h);
  o.z = fp_sub(fp_sub(fp_sqr(o.z), z1z1), hh);
  return o;
}

// g1-tensor.cu:10-20
ZK_HD G1J g1_neg(const G1J& a) {
  G1J r = a;
  r.y = fp_neg(a.y);
  return r;
}
ZK_HD G1A g1a_neg(const G1A& a) {
  G1A r = a;
  r.y = fp_neg(a.y);
  return r;
}

// g1-tensor.cu:422-430: LSB-first double-and-add over all 256 raw limb bits of x.
// limb w of x without a runtime-indexed register array (that would go to scratch on gfx950)
ZK_HD uint32_t fr_limb(const Fr& x, uint32_t w) {
  uint32_t r = x.v[0];
#pragma unroll
  for (uint32_t i = 1; i < 8; i++) r = (w == i) ? x.v[i] : r;
  return r;
}

ZK_HD G1J g1_mul_naive(G1J a, const Fr& x) {
  G1J out = g1_zero();
#pragma unroll 1
  for (uint32_t w = 0; w < 8; w++) {
    uint32_t limb = fr_limb(x, w);
#pragma unroll 1
    for (uint32_t b = 0; b < 32; b++) {
      if ((limb >> b) & 1u) out = g1_add(out, a);
      a = g1_double(a);
    }
  }
  return out;
}

}  // namespace zk
