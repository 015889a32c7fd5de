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

static int run();
int main() {
  try {
    return run();
  } catch (const std::exception& e) {
    std::cerr << "host api FAILED: uncaught exception: " << e.what() << "\n";
    return 2;
  }
}
static int run() {
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
  {  // the friend free functions with the reference's iterator signatures (fr-tensor.cuh:120-126, g1-tensor.cuh:170)
    EXPECT(eq(Fr_me(a, u.begin(), u.end()), a(u)));
    EXPECT(eq(Fr_me(a, u.begin(), u.begin()), a(0)));                       // no challenges: t(0), fr-tensor.cu:414
    {  // fewer challenges than log2(size): k folds of adjacent pairs, element 0 -- the multilinear evaluation of the first 2^k entries
      std::vector<Fr_t> host(128);
      zk_memcpy_d2h(zkdl::ctx(), host.data(), a.data(), sizeof(Fr_t) * 128);
      FrTensor head(128, host.data());
      EXPECT(eq(Fr_me(a, u.begin() + 3, u.end()), head(std::vector<Fr_t>(u.begin() + 3, u.end()))));
    }
    FrTensor pm = Fr_partial_me(a, u.begin(), u.begin() + 4, 8), pm2 = a.partial_me(std::vector<Fr_t>(u.begin(), u.begin() + 4), 8);
    EXPECT(pm.size == pm2.size && eq(pm(0), pm2(0)) && eq(pm(pm.size - 1), pm2(pm2.size - 1)));
    FrTensor idt = Fr_partial_me(a, u.begin(), u.begin(), 8);                // App. D-3: the identity
    EXPECT(idt.size == a.size && eq(idt(17), a(17)));
    std::vector<Fr_t> p1{one_raw}, p2 = inner_product_sumcheck(a, b, u);     // appends behind what is there
    Fr_ip_sc(a, b, u.begin(), u.end(), p1);
    EXPECT(p1.size() == 1 + p2.size() && memcmp(&p1[1], p2.data(), p2.size() * sizeof(Fr_t)) == 0);
    std::vector<Fr_t> p0;
    Fr_ip_sc(a, b, u.begin(), u.begin(), p0);
    EXPECT(p0.size() == 2 && eq(p0[0], a(0)) && eq(p0[1], b(0)));
    auto v = random_vec(10);
    std::vector<Fr_t> h1, h2 = hadamard_product_sumcheck(a, b, u, v);
    Fr_hp_sc(a, b, u.begin(), u.end(), v.begin(), v.end(), h1);
    EXPECT(h1.size() == h2.size() && memcmp(h1.data(), h2.data(), h2.size() * sizeof(Fr_t)) == 0);
    EXPECT(throws([&] { Fr_hp_sc(a, b, u.begin(), u.end(), v.begin(), v.end() - 1, h1); }, "Incompatible dimensions 6"));
    EXPECT(throws([&] { Fr_hp_sc(a, FrTensor(3), u.begin(), u.end(), v.begin(), v.end(), h1); }, "Incompatible dimensions 5"));
    std::vector<Fr_t> b1, b2 = binary_sumcheck(a, u, v);
    Fr_bin_sc(a, u.begin(), u.end(), v.begin(), v.end(), b1);
    EXPECT(b1.size() == b2.size() && memcmp(b1.data(), b2.data(), b2.size() * sizeof(Fr_t)) == 0);
    std::vector<Fr_t> b0;
    Fr_bin_sc(a, u.begin(), u.begin(), v.begin(), v.begin(), b0);
    EXPECT(b0.size() == 1 && eq(b0[0], a(0)));
    std::vector<Fr_t> ug = random_vec(4);
    G1Jacobian_t m1 = G1_me(gens, ug.begin(), ug.end()), m2 = gens(ug), m0 = G1_me(gens, ug.begin(), ug.begin()), g0 = gens(0u);
    EXPECT(memcmp(&m1, &m2, sizeof m1) == 0 && memcmp(&m0, &g0, sizeof m0) == 0);
  }
  {  // zkdl::ProofPipeline: sealed, rotating, batched proofs give the same transcripts as synchronous ones -- the run aggregate is the sum of the digests
    const int K = 7;
    auto body = [&] { relu.prove(Z, A); fc2.prove(X, Z, g8); };
    FrTensor acc(1);
    zk_memset_zero(zkdl::ctx(), acc.data(), sizeof(Fr_t));
    for (int k = 0; k < K; k++) {
      zk_transcript_reset(zkdl::transcript());
      zkdl::seed(1000 + k);
      body();
      Fr_t dg;
      zk_transcript_digest(zkdl::transcript(), &dg);
      acc += dg;
    }
    for (uint g1b : {1u, 3u}) {
      zkdl::ProofPipeline pipe(K, 4, g1b);
      for (int k = 0; k < K; k++) { zkdl::seed(1000 + k); pipe.submit(body); }
      EXPECT(zkdl::transcript() == zkdl::default_transcript());          // the pipeline's transcript is current only inside submit()
      Fr_t agg = pipe.finish();
      EXPECT(eq(agg, acc(0)));
      EXPECT(throws([&] { pipe.submit(body); }, "zkdl::ProofPipeline: capacity exhausted"));
      pipe.rewind(2);
      for (int k = 0; k < 2; k++) { zkdl::seed(1000 + k); pipe.submit(body); }
      size_t f2 = 0, g2 = 0;
      zk_transcript_counts(pipe.transcript_of(1), &f2, &g2);
      EXPECT(f2 == nf && g2 == ng);
    }
    zk_transcript_reset(zkdl::transcript());
    zkdl::seed(5);
    body();
    zk_transcript_counts(zkdl::transcript(), &nf, &ng);
  }
  std::cout << (fails ? "host api FAILED" : "host api ok") << " (" << nf << " Fr, " << ng << " G1 transcript values)" << std::endl;
  return fails ? 1 : 0;
}
