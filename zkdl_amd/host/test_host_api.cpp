// zkdl_amd/host/test_host_api.cpp -- exercises the C++ mirror of the reference API on the GPU
// (run by tests/test_gpu_host_api.py).  Checks algebraic identities and the reference's error strings.
#include <cstring>
#include <iostream>

#include "zkdl.hpp"

static int fails = 0;
#define EXPECT(cond)                                                       \
  do {                                                                     \
    if (!(cond)) { std::cerr << "FAIL " << __LINE__ << ": " #cond "\n"; ++fails; } \
  } while (0)
static bool eq(const Fr_t& a, const Fr_t& b) { return memcmp(&a, &b, sizeof a) == 0; }
template <class F> static bool throws(F f, const char* msg) {
  try { f(); } catch (const std::runtime_error& e) { return std::string(e.what()) == msg; }
  return false;
}

int main() {
  zkdl::seed(42);
  const uint n = 1 << 10;
  FrTensor a = FrTensor::random(n), b = FrTensor::random(n);
  FrTensor c = a + b;
  FrTensor d = c - b;
  EXPECT(eq(d(5), a(5)) && eq(d(n - 1), a(n - 1)));
  FrTensor e = a;                       // deep copy
  e += b;
  EXPECT(eq(e(7), c(7)) && !eq(a(7), c(7)));
  Fr_t one_raw = {{1, 0, 0, 0, 0, 0, 0, 0}};
  FrTensor ones(n);
  zk_memset_zero(zkdl::ctx(), ones.data(), sizeof(Fr_t) * n);
  ones += one_raw;
  ones.mont();
  EXPECT(eq((a * ones)(3), a(3)));                                   // Montgomery 1 is the identity
  EXPECT(eq(ones.unmont().sum(), Fr_t{{n, 0, 0, 0, 0, 0, 0, 0}}));
  {  // the reference's member name (bls12-381.cuh:120): reference-style code reads .val[]
    Fr_t s = ones.sum();
    EXPECT(s.val[0] == n && s.val[7] == 0 && &s.val[3] == &s.v[3]);
    Fp_t one = G1_ONE;
    EXPECT(one.val[0] == 196605u && one.val[11] == G1Jacobian_generator.z.val[11]);
  }
  // sumcheck: round 0 opens the inner product
  auto u = random_vec(10);
  auto pr = inner_product_sumcheck(a, b, u);
  EXPECT(pr.size() == 32);
  FrTensor ab = a * b;
  // <a,b> in Montgomery form = sum(a*b); c0 + (c0+c1+c2) must equal it
  FrTensor chk(3, pr.data());
  Fr_t lhs = (FrTensor(1, &pr[0]) + FrTensor(1, &pr[0]) + FrTensor(1, &pr[1]) + FrTensor(1, &pr[2]))(0);
  EXPECT(eq(lhs, ab.sum()));
  EXPECT(throws([&] { a + FrTensor(3); }, "Incompatible dimensions"));
  EXPECT(throws([&] { a(random_vec(3)); }, "Incompatible dimensions"));
  EXPECT(throws([&] { a.split(n); }, "Invalid window size."));
  EXPECT(throws([&] { hadamard_product_sumcheck(a, b, random_vec(10), random_vec(9)); }, "Incompatible dimensions 1"));
  // G1 / Commitment
  Commitment gens(16, G1Jacobian_generator);
  gens *= FrTensor::random(16);
  FrTensor w = FrTensor::random_int(64, 12).mont();
  G1TensorJacobian com = gens.commit(w);
  EXPECT(com.size == 4);
  G1TensorJacobian com2 = gens.commit(w + w);                        // linearity: commit(2w) = 2 commit(w)
  G1TensorJacobian dbl = com + com;
  G1Affine_t x1, x2;
  uint8_t* inf = zkdl::dalloc<uint8_t>(16);
  G1Affine_t* aff = zkdl::dalloc<G1Affine_t>(4);
  zk_g1_to_affine(zkdl::ctx(), com2.data(), 4, aff, inf);
  zk_memcpy_d2h(zkdl::ctx(), &x1, aff + 1, sizeof x1);
  zk_g1_to_affine(zkdl::ctx(), dbl.data(), 4, aff, inf);
  zk_memcpy_d2h(zkdl::ctx(), &x2, aff + 1, sizeof x2);
  EXPECT(memcmp(&x1, &x2, sizeof x1) == 0);
  std::vector<Fr_t> uo = random_vec(6);
  Fr_t ret = gens.open(w, com, uo);
  (void)ret;
  EXPECT(throws([&] { gens.commit(FrTensor(7)); }, "Incompatible dimensions"));
  EXPECT(throws([&] { gens *= FrTensor(3); }, "Incompatible dimensions 01"));
  // layers
  zkFC fc(8, 8, w, [&]() -> const Commitment& { static Commitment g8(8, G1Jacobian_generator); return g8; }());
  EXPECT(fc.inputSize == 8 && fc.outputSize == 8);
  FrTensor X = FrTensor::random_int(16, 10).mont();
  FrTensor Z = fc(X);
  zkReLU relu;
  FrTensor A = relu(Z);
  EXPECT(A.size == Z.size);
  relu.prove(Z, A);
  Commitment g8(8, G1Jacobian_generator);
  zkFC fc2(8, 8, w, g8);
  fc2.prove(X, Z, g8);
  size_t nf = 0, ng = 0;
  zk_transcript_counts(zkdl::transcript(), &nf, &ng);
  EXPECT(nf > 0 && ng == 1 + 3 * 3 + 1);
  EXPECT(throws([&] { zkFC bad(8, 4, w, g8); }, "Incompatible dimensions"));
  std::cout << (fails ? "host api FAILED" : "host api ok") << " (" << nf << " Fr, " << ng << " G1 transcript values)" << std::endl;
  return fails ? 1 : 0;
}
