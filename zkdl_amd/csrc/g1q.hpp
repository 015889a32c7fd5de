// zkdl_amd/csrc/g1q.hpp -- the Jacobian group law on the 14 x 28-bit lazily reduced base field of fq28.hpp, for the
// latency-bound scalar multiplications behind the derived round commitments (msm.hip: k_derive_mul / k_derive_sum) and the
// affine-canonical form: those kernels are ONE lane's chain of ~4000 dependent field products, and a product is 500
// instructions here against 950 issue slots (MAC + carry pairs and the wait states between them) in field.hpp's form.
// Formulas: dbl-2009-l and add-2007-bl, as g1.hpp (bls12-381.cu:1332-1357, :1400-1435); the results leave through
// g1q_to_jacobian / canonical form, so only the group element matters.
//
// Bounds (units of p; every stored coordinate has normalised limbs):  after a doubling X < 31, Y < 19, Z < 4; after an
// addition X < 9, Y < 7, Z < 2; an input converted from the library's form < 2.  Every product below has operand values whose
// product is < 2^11 and limbs within fq28.hpp's column bound.
#pragma once
#include "fq28.hpp"
#include "g1.hpp"

namespace zk {

struct alignas(8) G1Q {
  Fq x, y, z;
};
ZK_HD G1Q g1q_zero() {
  G1Q r;
  r.x = fq_zero();
  r.y = fq_one();
  r.z = fq_zero();
  return r;
}
ZK_HD bool g1q_is_zero(const G1Q& a) { return fq_all_zero(a.z); }
ZK_HD G1Q g1q_from_jacobian(const G1J& a) {
  if (l_is_zero(a.z)) return g1q_zero();
  G1Q r;
  r.x = fq_from_fp(a.x);
  r.y = fq_from_fp(a.y);
  r.z = fq_from_fp(a.z);
  return r;
}
ZK_HD G1J g1q_to_jacobian(const G1Q& a) {
  if (g1q_is_zero(a)) return g1_zero();
  G1J r;
  r.x = fq_to_fp(a.x);
  r.y = fq_to_fp(a.y);
  r.z = fq_to_fp(a.z);
  return r;
}
ZK_HD G1Q g1q_double(const G1Q& q) {
  if (g1q_is_zero(q)) return q;
  const Fq a = fq_sqr(q.x), b = fq_sqr(q.y), c = fq_sqr(b);
  const Fq d0 = fq_norm(fq_sub2<5>(fq_sqr(fq_add(q.x, b)), a, c));  // (x + b)^2 - a - c: < 7
  const Fq d = fq_norm(fq_add(d0, d0));                              // < 14
  const Fq e = fq_add(fq_add(a, a), a);                              // < 6, limbs < 2^29.6
  G1Q o;
  o.x = fq_norm(fq_sub3<29>(fq_sqr(e), fq_zero(), d));               // e^2 - 2d: < 31
  o.y = fq_norm(fq_sub8<17>(fq_mul(e, fq_sub<32>(d, o.x)), c));      // e (d - x3) - 8c: < 19
  const Fq yz = fq_mul(q.y, q.z);
  o.z = fq_norm(fq_add(yz, yz));                                     // < 4
  return o;
}
ZK_HD G1Q g1q_add(const G1Q& a, const G1Q& b) {
  if (g1q_is_zero(a)) return b;
  if (g1q_is_zero(b)) return a;
  const Fq z1z1 = fq_sqr(a.z), z2z2 = fq_sqr(b.z);
  const Fq u1 = fq_mul(a.x, z2z2), u2 = fq_mul(b.x, z1z1);
  const Fq s1 = fq_mul(fq_mul(a.y, b.z), z2z2), s2 = fq_mul(fq_mul(b.y, a.z), z1z1);
  const Fq h = fq_norm(fq_sub<3>(u2, u1));                           // < 5
  const Fq i = fq_sqr(fq_add(h, h));
  const Fq rh = fq_norm(fq_sub<3>(s2, s1));
  const Fq r = fq_add(rh, rh);                                       // < 10, limbs < 2^29
  const Fq rr = fq_sqr(r);
  if (fq_is_zero_mod_p(i)) return fq_is_zero_mod_p(rr) ? g1q_double(a) : g1q_zero();
  const Fq j = fq_mul(h, i), v = fq_mul(u1, i);
  G1Q o;
  o.x = fq_norm(fq_sub3<7>(rr, j, v));                               // r^2 - j - 2v: < 9
  o.y = fq_norm(fq_sub3<5>(fq_mul(r, fq_sub<10>(v, o.x)), fq_zero(), fq_mul(s1, j)));  // r (v - x3) - 2 s1 j: < 7
  o.z = fq_mul(fq_sub2<5>(fq_sqr(fq_add(a.z, b.z)), z1z1, z2z2), h);  // ((z1 + z2)^2 - z1z1 - z2z2) h: < 2
  return o;
}
ZK_HD G1Q g1q_neg(const G1Q& a) {
  G1Q r = a;
  r.y = fq_norm(fq_sub<20>(fq_zero(), a.y));  // < 20
  return r;
}
// (x, y, 1) canonical, fully reduced, in the library's form; infinity as {0, ONE, 0}
ZK_HD G1J g1q_canonical(const G1Q& a) {
  if (g1q_is_zero(a)) return g1_zero();
  const Fq zi = fq_inv(a.z), zi2 = fq_sqr(zi);
  G1J r;
  r.x = fq_to_fp(fq_mul(a.x, zi2));
  r.y = fq_to_fp(fq_mul(a.y, fq_mul(zi2, zi)));
  r.z = fp_one();
  return r;
}

}  // namespace zk
