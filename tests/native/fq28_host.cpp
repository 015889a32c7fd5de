// Host-side driver for tests/test_fq28.py: the 28-bit-limb base-field arithmetic (zkdl_amd/csrc/fq28.hpp) and the Jacobian group law on
// it (g1q.hpp), compiled for the CPU, on operands read from stdin (hex, the library's 12 x 32-bit Montgomery form); results canonical.
//   mul a b      -> to_fp(from_fp(a) * from_fp(b))            sqr a -> ...      inv a -> to_fp(inv(from_fp(a)))   (invf: the Fermat chain)
//   inv32 a / inv32f a -> fp_inv (binary GCD, fpinv.hpp) / fp_inv_fermat on the 12 x 32-bit form;  invint y -> y^-1 mod p of the plain integer
//   lazy a b     -> to_fp( (a + b) * norm(a - b + 3p) )      (unreduced operands)
//   dbl x y z / add x1 y1 z1 x2 y2 z2 / mul4 x y z k(hex, < 2^128)  -> canonical affine x y (Jacobian in, the library's form)
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include "../../zkdl_amd/csrc/g1q.hpp"
using namespace zk;

static Fp parse(const std::string& h) {
  Fp r = fp_zero();
  std::string s = h;
  while (s.size() < 96) s = "0" + s;
  for (int i = 0; i < 12; i++) r.v[i] = (uint32_t)strtoul(s.substr(96 - 8 * (i + 1), 8).c_str(), nullptr, 16);
  return r;
}
static void show(const Fp& a) {
  for (int i = 11; i >= 0; i--) printf("%08x", a.v[i]);
  printf("\n");
}
static G1J point() {
  std::string x, y, z;
  std::cin >> x >> y >> z;
  G1J p;
  p.x = parse(x); p.y = parse(y); p.z = parse(z);
  return p;
}
static void show_point(const G1Q& q) {
  G1J c = g1q_canonical(q);
  show(c.x); show(c.y); show(c.z);
}
int main() {
  std::string op;
  while (std::cin >> op) {
    if (op == "mul" || op == "lazy") {
      std::string a, b; std::cin >> a >> b;
      Fq x = fq_from_fp(parse(a)), y = fq_from_fp(parse(b));
      if (op == "mul") show(fq_to_fp(fq_mul(x, y)));
      else show(fq_to_fp(fq_mul(fq_add(x, y), fq_norm(fq_sub<3>(x, y)))));
    } else if (op == "sqr" || op == "inv") {
      std::string a; std::cin >> a;
      Fq x = fq_from_fp(parse(a));
      show(fq_to_fp(op == "sqr" ? fq_sqr(x) : fq_inv(x)));
    } else if (op == "invf") {       // the Fermat chains (a^(p-2)): the independent cross-check of the binary-GCD inverse
      std::string a; std::cin >> a;
      Fq tab[16];
      show(fq_to_fp(fq_inv_fermat(fq_from_fp(parse(a)), tab, 1)));
    } else if (op == "inv32" || op == "inv32f" || op == "invint") {   // 12 x 32-bit Montgomery form: fp_inv (binary GCD) / fp_inv_fermat; invint: plain integer inverse
      std::string a; std::cin >> a;
      show(op == "inv32" ? fp_inv(parse(a)) : op == "inv32f" ? fp_inv_fermat(parse(a)) : fp_inv_int(parse(a)));
    } else if (op == "dbl") {
      show_point(g1q_double(g1q_from_jacobian(point())));
    } else if (op == "add") {
      G1Q a = g1q_from_jacobian(point()), b = g1q_from_jacobian(point());
      show_point(g1q_add(a, b));
    } else if (op == "chain") {  // ((2P + Q) doubled 5 times - Q) + P: the value ranges of the scalar-multiplication loop
      G1Q p = g1q_from_jacobian(point()), q = g1q_from_jacobian(point());
      G1Q a = g1q_add(g1q_double(p), q);
      for (int i = 0; i < 5; i++) a = g1q_double(a);
      a = g1q_add(g1q_add(a, g1q_neg(q)), p);
      show_point(a);
    }
  }
  return 0;
}
