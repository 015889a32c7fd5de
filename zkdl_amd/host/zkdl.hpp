// zkdl_amd/host/zkdl.hpp -- the reference's C++ API rebuilt on the C ABI (include/zkdl_amd.h).
//
// Same class names, member signatures, ownership rules and exception strings as
// fr-tensor.cuh:62-133, g1-tensor.cuh:65-174, proof.cuh:13-34, commitment.cuh:8-23,
// zkfc.cuh:16-33, zkrelu.cuh:10-20, timer.hpp:6-22, so demo.cu-style drivers compile against it
// unchanged.  Differences, all additive:
//   * every operation runs on libzkdl_amd.so's HIP stream; nothing here touches the device itself;
//   * random_vec / FrTensor::random draw from a seedable counter-based generator (zkdl::seed());
//     unseeded it is seeded from std::random_device, like the reference;
//   * the proof values the reference computes and drops are appended to zkdl::transcript().
#pragma once
#include <cstdint>
#include <iomanip>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zkdl_amd.h"

typedef zk_fr Fr_t;          // limbs reachable as .val[] (bls12-381.cuh:120) and as .v[] (the C ABI's spelling)
typedef zk_fp Fp_t;
typedef zk_g1a G1Affine_t;
typedef zk_g1j G1Jacobian_t;
typedef unsigned int uint;

namespace zkdl {
inline zk_ctx*& ctx_slot() { static zk_ctx* c = nullptr; return c; }
inline bool& seeded() { static bool s = false; return s; }
inline zk_ctx* ctx() {
  zk_ctx*& c = ctx_slot();
  if (!c) {
    int dev = 0;
    if (const char* e = getenv("ZKDL_DEVICE")) dev = atoi(e);
    if (zk_ctx_create(dev, &c) != ZK_OK) throw std::runtime_error("zkdl: no usable HIP device (the proving path has no CPU fallback)");
  }
  return c;
}
inline void seed(uint64_t s) { zk_rng_seed(ctx(), s); seeded() = true; }
inline void ensure_seeded() {
  if (!seeded()) {
    if (const char* e = getenv("ZKDL_SEED")) seed(strtoull(e, nullptr, 0));
    else { std::random_device rd; seed(((uint64_t)rd() << 32) | rd()); }   // proof.cu:5, fr-tensor.cu:351-359
  }
}
// status -> the reference's exceptions
inline void check(int rc, const char* what = "Incompatible dimensions") {
  if (rc == ZK_OK) return;
  if (rc == ZK_ERR_SHAPE) throw std::runtime_error(what);
  throw std::runtime_error(std::string("zkdl: ") + zk_last_error(ctx()));
}
// the transcript prove calls append to: the process-wide default one, unless a ProofPipeline (below) has made one of its own current
inline zk_transcript*& current_transcript() { static zk_transcript* t = nullptr; return t; }
inline zk_transcript* default_transcript() {
  static zk_transcript* t = nullptr;
  if (!t) check(zk_transcript_create(ctx(), &t));
  return t;
}
inline zk_transcript* transcript() { return current_transcript() ? current_transcript() : default_transcript(); }
template <class T> inline T* dalloc(size_t n) {
  void* p = nullptr;
  check(zk_malloc(ctx(), (n ? n : 1) * sizeof(T), &p));
  return static_cast<T*>(p);
}
}  // namespace zkdl

// g1-tensor.cuh:28-63
const Fp_t G1_generator_x_mont = {{4250078230u, 1555269520u, 2574712821u, 2014837863u, 339452353u, 357537223u, 4090554183u, 4037962445u, 568063040u, 3989728972u, 2651585397u, 302085953u}};
const Fp_t G1_generator_y_mont = {{216474225u, 3131872213u, 2031680910u, 2351063834u, 1460086222u, 3713621779u, 1346392468u, 1370249257u, 2902481344u, 236751935u, 1342743146u, 196886268u}};
const Fp_t G1_ONE = {{196605u, 1980301312u, 3289120770u, 3958636555u, 1405573306u, 1598593111u, 1884444485u, 2010011731u, 2723605613u, 1543969431u, 4202751123u, 368467651u}};
const G1Affine_t G1Affine_generator{G1_generator_x_mont, G1_generator_y_mont};
const G1Jacobian_t G1Jacobian_generator{G1_generator_x_mont, G1_generator_y_mont, G1_ONE};

// stream printers: fr-tensor.cu:5-13, g1-tensor.cu:22-42
inline std::ostream& operator<<(std::ostream& os, const Fr_t& x) {
  os << "0x" << std::hex;
  for (uint i = 8; i > 0; --i) os << std::setfill('0') << std::setw(8) << x.v[i - 1];
  return os << std::dec << std::setw(0) << std::setfill(' ');
}
inline std::ostream& operator<<(std::ostream& os, const Fp_t& x) {
  os << "0x" << std::hex;
  for (uint i = 12; i > 0; --i) os << std::setfill('0') << std::setw(8) << x.v[i - 1];
  return os << std::dec << std::setw(0) << std::setfill(' ');
}
inline std::ostream& operator<<(std::ostream& os, const G1Affine_t& g) { return os << "(" << g.x << ", " << g.y << ")"; }
inline std::ostream& operator<<(std::ostream& os, const G1Jacobian_t& g) { return os << "(" << g.x << ", " << g.y << ", " << g.z << ")"; }

class G1TensorAffine;
class G1TensorJacobian;
class Commitment;
class zkFC;
class zkReLU;

// ------------------------------------------------------------------ FrTensor (fr-tensor.cuh:62-133)
class FrTensor {
 private:
  Fr_t* gpu_data;

 public:
  const uint size;
  FrTensor(uint size) : gpu_data(zkdl::dalloc<Fr_t>(size)), size(size) {}
  FrTensor(uint size, const Fr_t* cpu_data) : gpu_data(zkdl::dalloc<Fr_t>(size)), size(size) {
    zkdl::check(zk_memcpy_h2d(zkdl::ctx(), gpu_data, cpu_data, sizeof(Fr_t) * size));
  }
  FrTensor(const FrTensor& t) : gpu_data(zkdl::dalloc<Fr_t>(t.size)), size(t.size) {   // deep copy, fr-tensor.cu:103-107
    zkdl::check(zk_memcpy_d2d(zkdl::ctx(), gpu_data, t.gpu_data, sizeof(Fr_t) * size));
  }
  ~FrTensor() { zk_free(zkdl::ctx_slot(), gpu_data); gpu_data = nullptr; }
  Fr_t* data() { return gpu_data; }                 // additive: device pointer for ABI-level callers
  const Fr_t* data() const { return gpu_data; }

  Fr_t operator()(uint idx) const {
    Fr_t out;
    zkdl::check(zk_memcpy_d2h(zkdl::ctx(), &out, gpu_data + idx, sizeof(Fr_t)));
    return out;
  }
#define ZKDL_FR_BIN(OP, FN)                                                                     \
  FrTensor operator OP(const FrTensor& t) const {                                               \
    if (size != t.size) throw std::runtime_error("Incompatible dimensions");                    \
    FrTensor out(size);                                                                         \
    zkdl::check(FN(zkdl::ctx(), gpu_data, t.gpu_data, out.gpu_data, size));                     \
    return out;                                                                                 \
  }                                                                                             \
  FrTensor& operator OP##=(const FrTensor& t) {                                                 \
    if (size != t.size) throw std::runtime_error("Incompatible dimensions");                    \
    zkdl::check(FN(zkdl::ctx(), gpu_data, t.gpu_data, gpu_data, size));                         \
    return *this;                                                                               \
  }                                                                                             \
  FrTensor operator OP(const Fr_t& x) const {                                                   \
    FrTensor out(size);                                                                         \
    zkdl::check(FN##_scalar(zkdl::ctx(), gpu_data, &x, out.gpu_data, size));                    \
    return out;                                                                                 \
  }                                                                                             \
  FrTensor& operator OP##=(const Fr_t& x) {                                                     \
    zkdl::check(FN##_scalar(zkdl::ctx(), gpu_data, &x, gpu_data, size));                        \
    return *this;                                                                               \
  }
  ZKDL_FR_BIN(+, zk_fr_add)
  ZKDL_FR_BIN(-, zk_fr_sub)
  ZKDL_FR_BIN(*, zk_fr_mul)
#undef ZKDL_FR_BIN
  FrTensor operator-() const {
    FrTensor out(size);
    zkdl::check(zk_fr_neg(zkdl::ctx(), gpu_data, out.gpu_data, size));
    return out;
  }
  FrTensor& mont() { zkdl::check(zk_fr_mont(zkdl::ctx(), gpu_data, gpu_data, size)); return *this; }
  FrTensor& unmont() { zkdl::check(zk_fr_unmont(zkdl::ctx(), gpu_data, gpu_data, size)); return *this; }
  Fr_t sum() const {
    Fr_t out;
    zkdl::check(zk_fr_sum(zkdl::ctx(), gpu_data, size, &out));
    return out;
  }
  Fr_t operator()(const std::vector<Fr_t>& u) const {   // fr-tensor.cu:295-300
    Fr_t out;
    zkdl::check(zk_fr_me(zkdl::ctx(), gpu_data, size, u.data(), (uint32_t)u.size(), &out));
    return out;
  }
  std::pair<FrTensor, FrTensor> split(uint window_size) const {   // fr-tensor.cu:389-397
    if (window_size < 1 || window_size >= size) throw std::runtime_error("Invalid window size.");
    uint out_size = (size + 1) / 2;
    std::pair<FrTensor, FrTensor> out{out_size, out_size};
    zkdl::check(zk_fr_split(zkdl::ctx(), gpu_data, size, window_size, out.first.gpu_data, out.second.gpu_data), "Invalid window size.");
    return out;
  }
  FrTensor partial_me(std::vector<Fr_t> u, uint window_size) const {   // fr-tensor.cu:370-374
    uint n = zk_fr_partial_me_size(size, (uint32_t)u.size(), window_size);
    if (!u.empty() && (uint64_t)size <= (uint64_t)window_size * (1ull << (u.size() - 1))) throw std::runtime_error("Incompatible dimensions");
    FrTensor out(n);
    zkdl::check(zk_fr_partial_me(zkdl::ctx(), gpu_data, size, u.data(), (uint32_t)u.size(), window_size, out.gpu_data));
    return out;
  }
  static FrTensor random_int(uint size, uint num_bits) {   // fr-tensor.cu:316-335
    zkdl::ensure_seeded();
    FrTensor out(size);
    zkdl::check(zk_fr_random_int(zkdl::ctx(), out.gpu_data, size, num_bits));
    return out;
  }
  static FrTensor random(uint size) {   // fr-tensor.cu:349-368
    zkdl::ensure_seeded();
    FrTensor out(size);
    zkdl::check(zk_fr_random(zkdl::ctx(), out.gpu_data, size));
    return out;
  }
  friend class G1TensorAffine;
  friend class G1TensorJacobian;
  friend class Commitment;
  friend class zkFC;
  friend class zkReLU;
  friend std::vector<Fr_t> inner_product_sumcheck(const FrTensor& a, const FrTensor& b, std::vector<Fr_t> u);
  friend std::vector<Fr_t> hadamard_product_sumcheck(const FrTensor& a, const FrTensor& b, std::vector<Fr_t> u, std::vector<Fr_t> v);
  friend std::vector<Fr_t> binary_sumcheck(const FrTensor& a, std::vector<Fr_t> u, std::vector<Fr_t> v);
};
inline std::ostream& operator<<(std::ostream& os, const FrTensor& A) {   // fr-tensor.cu:445-451 (one bulk copy instead of size copies)
  std::vector<Fr_t> h(A.size);
  zkdl::check(zk_memcpy_d2h(zkdl::ctx(), h.data(), A.data(), sizeof(Fr_t) * A.size));
  os << '[';
  for (uint i = 0; i + 1 < A.size; ++i) os << h[i] << '\n';
  if (A.size) os << h[A.size - 1];
  return os << ']';
}

// the friend free functions with the reference's iterator signatures (fr-tensor.cuh:120-126; bodies fr-tensor.cu:411-443).  The reference
// recurses launch by launch; here each is one call into the C ABI (same field elements).
inline Fr_t Fr_me(const FrTensor& t, std::vector<Fr_t>::const_iterator begin, std::vector<Fr_t>::const_iterator end) {
  if (begin >= end) return t(0);                                    // fr-tensor.cu:414
  // fewer challenges than log2(size): the recursion folds adjacent pairs k times and returns element 0, which depends on the first 2^k entries only
  const size_t k = (size_t)(end - begin);
  const uint n = (k < 32 && t.size > (1u << k)) ? (1u << k) : t.size;
  std::vector<Fr_t> u(begin, end);
  Fr_t out;
  zkdl::check(zk_fr_me(zkdl::ctx(), t.data(), n, u.data(), (uint32_t)u.size(), &out));
  return out;
}
inline FrTensor Fr_partial_me(const FrTensor& t, std::vector<Fr_t>::const_iterator begin, std::vector<Fr_t>::const_iterator end, uint window_size) {
  if (begin >= end) return t;                                       // fr-tensor.cu:436 (App. D-3: no challenges = the identity)
  return t.partial_me(std::vector<Fr_t>(begin, end), window_size);
}

// ------------------------------------------------------------------ proof.cuh:13-34
namespace zkdl {
// additive: when set, every challenge random_vec hands out is also appended here, in draw order (demo --dump:
// lets a checker replay the proof with the same challenges; the reference draws them and forgets them, proof.cu:3-11)
inline std::vector<Fr_t>*& challenge_log() { static std::vector<Fr_t>* p = nullptr; return p; }
// additive: hash-derived challenges.  After fs_begin(statement digest), fs_call(layer, kind) before a prove call makes that
// call's random_vec draws zk_fs_derive(seed, layer, kind, 0), (.., 1), ... in the reference's draw order; a verifier that
// knows the statement re-derives them (include/zkdl_amd.h: zk_fs_derive).  fs_end() returns to the seeded generator.
struct FsState {
  bool on = false;
  uint8_t seed[32] = {};
  uint32_t layer = 0, kind = 0, vec = 0;
};
inline FsState& fs() { static FsState s; return s; }
inline void fs_begin(const uint8_t seed32[32]) { FsState& f = fs(); f.on = true; for (int i = 0; i < 32; i++) f.seed[i] = seed32[i]; f.vec = 0; }
inline void fs_call(uint32_t layer, uint32_t kind) { FsState& f = fs(); f.layer = layer; f.kind = kind; f.vec = 0; }
inline void fs_end() { fs().on = false; }
}  // namespace zkdl
inline std::vector<Fr_t> random_vec(uint len) {
  std::vector<Fr_t> out(len);
  if (zkdl::fs().on) {
    zkdl::FsState& f = zkdl::fs();
    zkdl::check(zk_fs_derive(f.seed, f.layer, f.kind, f.vec++, len, out.data()));
    if (zkdl::challenge_log()) zkdl::challenge_log()->insert(zkdl::challenge_log()->end(), out.begin(), out.end());
    return out;
  }
  zkdl::ensure_seeded();
  zkdl::check(zk_random_vec(zkdl::ctx(), len, out.data()));
  if (zkdl::challenge_log()) zkdl::challenge_log()->insert(zkdl::challenge_log()->end(), out.begin(), out.end());
  return out;
}
inline uint ceilLog2(uint num) { return zk_ceil_log2(num); }
template <typename T>
std::vector<T> concatenate(const std::vector<std::vector<T>>& vecs) {
  std::vector<T> result;
  for (const auto& v : vecs) result.insert(result.end(), v.begin(), v.end());
  return result;
}
inline std::vector<Fr_t> inner_product_sumcheck(const FrTensor& a, const FrTensor& b, std::vector<Fr_t> u) {
  if (a.size != b.size) throw std::runtime_error("Incompatible dimensions");
  std::vector<Fr_t> proof(3 * u.size() + 2);
  zkdl::check(zk_sumcheck_ip(zkdl::ctx(), a.gpu_data, b.gpu_data, a.size, u.data(), (uint32_t)u.size(), proof.data()));
  return proof;
}
inline std::vector<Fr_t> hadamard_product_sumcheck(const FrTensor& a, const FrTensor& b, std::vector<Fr_t> u, std::vector<Fr_t> v) {
  if (u.size() != v.size()) throw std::runtime_error("Incompatible dimensions 1");
  if (a.size != b.size) throw std::runtime_error("Incompatible dimensions 2");
  std::vector<Fr_t> proof(3 * u.size() + 2);
  zkdl::check(zk_sumcheck_hadamard(zkdl::ctx(), a.gpu_data, b.gpu_data, a.size, u.data(), v.data(), (uint32_t)u.size(), proof.data()), "Incompatible dimensions 3");
  return proof;
}
inline std::vector<Fr_t> binary_sumcheck(const FrTensor& a, std::vector<Fr_t> u, std::vector<Fr_t> v) {
  if (u.size() != v.size()) throw std::runtime_error("Incompatible dimensions");
  std::vector<Fr_t> proof(3 * u.size() + 1);
  zkdl::check(zk_sumcheck_binary(zkdl::ctx(), a.gpu_data, a.size, u.data(), v.data(), (uint32_t)u.size(), proof.data()));
  return proof;
}

// proof.cu:72-96, 110-137, 165-188: the recursive provers behind the three wrappers above, appending to `proof`.  Unlike the wrappers they make no
// size checks against the number of challenges (the recursion simply runs end - begin rounds); the C ABI needs a.size <= 2^rounds (a shorter
// tensor is zero-padded, which is what the reference's bounds-checked kernels compute), so a longer one is rejected here.
inline void Fr_ip_sc(const FrTensor& a, const FrTensor& b, std::vector<Fr_t>::const_iterator begin, std::vector<Fr_t>::const_iterator end, std::vector<Fr_t>& proof) {
  if (a.size != b.size) throw std::runtime_error("Incompatible dimensions");
  if (begin >= end) { proof.push_back(a(0)); proof.push_back(b(0)); return; }
  auto p = inner_product_sumcheck(a, b, std::vector<Fr_t>(begin, end));
  proof.insert(proof.end(), p.begin(), p.end());
}
inline void Fr_hp_sc(const FrTensor& a, const FrTensor& b, std::vector<Fr_t>::const_iterator u_begin, std::vector<Fr_t>::const_iterator u_end,
                     std::vector<Fr_t>::const_iterator v_begin, std::vector<Fr_t>::const_iterator v_end, std::vector<Fr_t>& proof) {
  if (a.size != b.size) throw std::runtime_error("Incompatible dimensions 5");
  if (v_end - v_begin != u_end - u_begin) throw std::runtime_error("Incompatible dimensions 6");
  if (v_begin >= v_end) { proof.push_back(a(0)); proof.push_back(b(0)); return; }
  auto p = hadamard_product_sumcheck(a, b, std::vector<Fr_t>(u_begin, u_end), std::vector<Fr_t>(v_begin, v_end));
  proof.insert(proof.end(), p.begin(), p.end());
}
inline void Fr_bin_sc(const FrTensor& a, std::vector<Fr_t>::const_iterator u_begin, std::vector<Fr_t>::const_iterator u_end, std::vector<Fr_t>::const_iterator v_begin,
                      std::vector<Fr_t>::const_iterator v_end, std::vector<Fr_t>& proof) {
  if (v_end - v_begin != u_end - u_begin) throw std::runtime_error("Incompatible dimensions 6");
  if (v_begin >= v_end) { proof.push_back(a(0)); return; }
  auto p = binary_sumcheck(a, std::vector<Fr_t>(u_begin, u_end), std::vector<Fr_t>(v_begin, v_end));
  proof.insert(proof.end(), p.begin(), p.end());
}

// ------------------------------------------------------------------ G1 tensors (g1-tensor.cuh:65-174)
class G1Tensor {
 public:
  const uint size;
  G1Tensor(uint size) : size(size) {}
};

class G1TensorAffine : public G1Tensor {
 protected:
  G1Affine_t* gpu_data;

 public:
  G1TensorAffine(const G1TensorAffine& t) : G1Tensor(t.size), gpu_data(zkdl::dalloc<G1Affine_t>(t.size)) {
    zkdl::check(zk_memcpy_d2d(zkdl::ctx(), gpu_data, t.gpu_data, sizeof(G1Affine_t) * size));
  }
  G1TensorAffine(uint size) : G1Tensor(size), gpu_data(zkdl::dalloc<G1Affine_t>(size)) {}
  G1TensorAffine(uint size, const G1Affine_t& g) : G1Tensor(size), gpu_data(zkdl::dalloc<G1Affine_t>(size)) {
    zkdl::check(zk_g1a_fill(zkdl::ctx(), gpu_data, size, &g));
  }
  G1TensorAffine(uint size, const G1Affine_t* cpu_data) : G1Tensor(size), gpu_data(zkdl::dalloc<G1Affine_t>(size)) {
    zkdl::check(zk_memcpy_h2d(zkdl::ctx(), gpu_data, cpu_data, sizeof(G1Affine_t) * size));
  }
  ~G1TensorAffine() { zk_free(zkdl::ctx_slot(), gpu_data); gpu_data = nullptr; }
  G1Affine_t operator()(uint idx) const {
    G1Affine_t out;
    zkdl::check(zk_memcpy_d2h(zkdl::ctx(), &out, gpu_data + idx, sizeof(G1Affine_t)));
    return out;
  }
  G1TensorAffine operator-() const {
    G1TensorAffine out(size);
    zkdl::check(zk_g1a_neg(zkdl::ctx(), gpu_data, out.gpu_data, size));
    return out;
  }
  friend class G1TensorJacobian;
};

class G1TensorJacobian : public G1Tensor {
 protected:
  G1Jacobian_t* gpu_data;
  uint64_t version_ = 0;   // bumped by every mutating operator (Commitment caches its MSM tables against it)

 public:
  G1TensorJacobian(const G1TensorJacobian& t) : G1Tensor(t.size), gpu_data(zkdl::dalloc<G1Jacobian_t>(t.size)) {
    zkdl::check(zk_memcpy_d2d(zkdl::ctx(), gpu_data, t.gpu_data, sizeof(G1Jacobian_t) * size));
  }
  G1TensorJacobian(uint size) : G1Tensor(size), gpu_data(zkdl::dalloc<G1Jacobian_t>(size)) {}
  G1TensorJacobian(uint size, const G1Jacobian_t& g) : G1Tensor(size), gpu_data(zkdl::dalloc<G1Jacobian_t>(size)) {
    zkdl::check(zk_g1_fill(zkdl::ctx(), gpu_data, size, &g));
  }
  G1TensorJacobian(uint size, const G1Jacobian_t* cpu_data) : G1Tensor(size), gpu_data(zkdl::dalloc<G1Jacobian_t>(size)) {
    zkdl::check(zk_memcpy_h2d(zkdl::ctx(), gpu_data, cpu_data, sizeof(G1Jacobian_t) * size));
  }
  G1TensorJacobian(const G1TensorAffine& a) : G1Tensor(a.size), gpu_data(zkdl::dalloc<G1Jacobian_t>(a.size)) {
    zkdl::check(zk_g1_from_affine(zkdl::ctx(), a.gpu_data, gpu_data, size));
  }
  virtual ~G1TensorJacobian() { zk_free(zkdl::ctx_slot(), gpu_data); gpu_data = nullptr; }
  G1Jacobian_t* data() { return gpu_data; }
  const G1Jacobian_t* data() const { return gpu_data; }

  G1Jacobian_t operator()(uint idx) const {
    G1Jacobian_t out;
    zkdl::check(zk_memcpy_d2h(zkdl::ctx(), &out, gpu_data + idx, sizeof(G1Jacobian_t)));
    return out;
  }
  G1TensorJacobian operator-() const {
    G1TensorJacobian out(size);
    zkdl::check(zk_g1_neg(zkdl::ctx(), gpu_data, out.gpu_data, size));
    return out;
  }
#define ZKDL_G1_OPS(OP, JJ, JA, JS, AS)                                                          \
  G1TensorJacobian operator OP(const G1TensorJacobian& t) const {                                \
    if (size != t.size) throw std::runtime_error("Incompatible dimensions");                     \
    G1TensorJacobian out(size);                                                                  \
    zkdl::check(JJ(zkdl::ctx(), gpu_data, t.gpu_data, out.gpu_data, size));                      \
    return out;                                                                                  \
  }                                                                                              \
  G1TensorJacobian operator OP(const G1TensorAffine& t) const {                                  \
    if (size != t.size) throw std::runtime_error("Incompatible dimensions");                     \
    G1TensorJacobian out(size);                                                                  \
    zkdl::check(JA(zkdl::ctx(), gpu_data, t.gpu_data, out.gpu_data, size));                      \
    return out;                                                                                  \
  }                                                                                              \
  G1TensorJacobian operator OP(const G1Jacobian_t& x) const {                                    \
    G1TensorJacobian out(size);                                                                  \
    zkdl::check(JS(zkdl::ctx(), gpu_data, &x, out.gpu_data, size));                              \
    return out;                                                                                  \
  }                                                                                              \
  G1TensorJacobian operator OP(const G1Affine_t& x) const {                                      \
    G1TensorJacobian out(size);                                                                  \
    zkdl::check(AS(zkdl::ctx(), gpu_data, &x, out.gpu_data, size));                              \
    return out;                                                                                  \
  }                                                                                              \
  G1TensorJacobian& operator OP##=(const G1TensorJacobian& t) {                                  \
    if (size != t.size) throw std::runtime_error("Incompatible dimensions");                     \
    ++version_;                                                                                  \
    zkdl::check(JJ(zkdl::ctx(), gpu_data, t.gpu_data, gpu_data, size));                          \
    return *this;                                                                                \
  }                                                                                              \
  G1TensorJacobian& operator OP##=(const G1TensorAffine& t) {                                    \
    if (size != t.size) throw std::runtime_error("Incompatible dimensions");                     \
    ++version_;                                                                                  \
    zkdl::check(JA(zkdl::ctx(), gpu_data, t.gpu_data, gpu_data, size));                          \
    return *this;                                                                                \
  }                                                                                              \
  G1TensorJacobian& operator OP##=(const G1Jacobian_t& x) {                                      \
    ++version_;                                                                                  \
    zkdl::check(JS(zkdl::ctx(), gpu_data, &x, gpu_data, size));                                  \
    return *this;                                                                                \
  }                                                                                              \
  G1TensorJacobian& operator OP##=(const G1Affine_t& x) {                                        \
    ++version_;                                                                                  \
    zkdl::check(AS(zkdl::ctx(), gpu_data, &x, gpu_data, size));                                  \
    return *this;                                                                                \
  }
  ZKDL_G1_OPS(+, zk_g1_add, zk_g1_madd, zk_g1_add_bcast, zk_g1_madd_bcast)
  ZKDL_G1_OPS(-, zk_g1_sub, zk_g1_msub, zk_g1_sub_bcast, zk_g1_msub_bcast)
#undef ZKDL_G1_OPS
  G1Jacobian_t sum() const {
    G1Jacobian_t out;
    zkdl::check(zk_g1_sum(zkdl::ctx(), gpu_data, size, &out));
    return out;
  }
  G1TensorJacobian operator*(const FrTensor& s) const {   // g1-tensor.cu:447-454
    if (s.size % size != 0) throw std::runtime_error("Incompatible dimensions");
    G1TensorJacobian out(s.size);
    zkdl::check(zk_g1_mul_bcast(zkdl::ctx(), gpu_data, size, s.gpu_data, s.size, out.gpu_data));
    return out;
  }
  G1TensorJacobian& operator*=(const FrTensor& s) {   // g1-tensor.cu:456-461
    if (size != s.size) throw std::runtime_error("Incompatible dimensions 01");
    ++version_;
    zkdl::check(zk_g1_mul(zkdl::ctx(), gpu_data, s.gpu_data, gpu_data, size));
    return *this;
  }
  G1Jacobian_t operator()(const std::vector<Fr_t>& u) const {   // g1-tensor.cu:486-491
    G1Jacobian_t out;
    zkdl::check(zk_g1_me(zkdl::ctx(), gpu_data, size, u.data(), (uint32_t)u.size(), &out));
    return out;
  }
  friend class G1TensorAffine;
  friend class Commitment;
  friend class zkFC;
};

// g1-tensor.cuh:170, g1-tensor.cu:477-484
inline G1Jacobian_t G1_me(const G1TensorJacobian& t, std::vector<Fr_t>::const_iterator begin, std::vector<Fr_t>::const_iterator end) {
  if (begin >= end) return t(0);
  const size_t k = (size_t)(end - begin);
  const uint n = (k < 32 && t.size > (1u << k)) ? (1u << k) : t.size;   // as Fr_me: element 0 after k folds of adjacent pairs
  std::vector<Fr_t> u(begin, end);
  G1Jacobian_t out;
  zkdl::check(zk_g1_me(zkdl::ctx(), t.data(), n, u.data(), (uint32_t)u.size(), &out));
  return out;
}

// ------------------------------------------------------------------ Commitment (commitment.cuh:8-23)
class Commitment : public G1TensorJacobian {
  mutable zk_gens* handle_ = nullptr;
  mutable uint64_t handle_version_ = ~0ull;

 public:
  using G1TensorJacobian::G1TensorJacobian;
  using G1TensorJacobian::operator+;
  using G1TensorJacobian::operator-;
  using G1TensorJacobian::operator*;
  using G1TensorJacobian::operator*=;
  Commitment(const Commitment& c) : G1TensorJacobian(c) {}
  ~Commitment() override { if (handle_) zk_gens_destroy(handle_); }
  // additive: precompute every signed digit of every window of these generators (zk_gens_prepare_full): full-width MSMs over them -- the
  // openings of zkFC::prove -- then run without buckets.  Setup, like the commitments; the points must not be mutated afterwards.
  void prepare_full(uint window_bits = 12) const { zkdl::check(zk_gens_prepare_full(zkdl::ctx(), handle(), window_bits)); }
  // generator handle (snapshot of the current points + their MSM window tables), rebuilt after mutation
  zk_gens* handle() const {
    if (!handle_ || handle_version_ != version_) {
      if (handle_) zk_gens_destroy(handle_);
      handle_ = nullptr;
      zkdl::check(zk_gens_create(zkdl::ctx(), gpu_data, size, &handle_));
      handle_version_ = version_;
    }
    return handle_;
  }
  G1TensorJacobian commit(const FrTensor& t) const {   // commitment.cu:29-41
    if (t.size % size != 0) throw std::runtime_error("Incompatible dimensions");
    G1TensorJacobian out(t.size / size);
    zkdl::check(zk_commit(zkdl::ctx(), handle(), t.gpu_data, t.size, out.gpu_data));
    return out;
  }
  Fr_t open(const FrTensor& t, const G1TensorJacobian& c, const std::vector<Fr_t>& u) const {   // commitment.cu:83-92
    uint k_hi = ceilLog2(c.size);
    if (k_hi > u.size()) throw std::runtime_error("Incompatible dimensions");
    std::vector<G1Jacobian_t> proof(3 * (u.size() - k_hi) + 1);
    G1Jacobian_t com_eval;
    Fr_t ret;
    zkdl::check(zk_open(zkdl::ctx(), handle(), t.gpu_data, t.size, c.gpu_data, c.size, u.data(), (uint32_t)u.size(), &com_eval, proof.data(), &ret));
    return ret;
  }
  static Fr_t me_open(const FrTensor& t, const Commitment& generators, std::vector<Fr_t>::const_iterator begin, std::vector<Fr_t>::const_iterator end,
                      std::vector<G1Jacobian_t>& proof) {   // commitment.cu:62-81
    if (t.size != generators.size) throw std::runtime_error("Incompatible dimensions");
    std::vector<Fr_t> u(begin, end);
    std::vector<G1Jacobian_t> p(3 * u.size() + 1);
    Fr_t ret;
    zkdl::check(zk_me_open(zkdl::ctx(), generators.handle(), t.gpu_data, u.data(), (uint32_t)u.size(), p.data(), &ret));
    proof.insert(proof.end(), p.begin(), p.end());
    return ret;
  }
};

// ------------------------------------------------------------------ zkFC (zkfc.cuh:16-33)
class zkFC {
 private:
  zk_fc* h_ = nullptr;
  explicit zkFC(zk_fc* h) : h_(h), inputSize(zk_fc_in(h)), outputSize(zk_fc_out(h)) {}

 public:
  const uint inputSize;
  const uint outputSize;
  zkFC(uint input_size, uint output_size, const FrTensor& t, const Commitment& generators) : inputSize(input_size), outputSize(output_size) {
    if (t.size != input_size * output_size) throw std::runtime_error("Incompatible dimensions");   // zkfc.cu:103
    zkdl::check(zk_fc_create(zkdl::ctx(), input_size, output_size, t.gpu_data, generators.handle(), &h_));
  }
  zkFC(const zkFC&) = delete;   // the reference deep-copies on vector growth (no move ctor); here vectors move
  zkFC(zkFC&& o) noexcept : h_(o.h_), inputSize(o.inputSize), outputSize(o.outputSize) { o.h_ = nullptr; }
  ~zkFC() { if (h_) zk_fc_destroy(h_); }
  uint commitment_size() const { return zk_fc_com_size(h_); }                 // additive: the public commitment `com` (zkfc.cuh:22)
  const G1Jacobian_t* commitment_points() const { return zk_fc_com(h_); }     // device pointer
  FrTensor operator()(const FrTensor& X) const {   // zkfc.cu:117-126
    if (X.size % inputSize != 0) throw std::runtime_error("Incompatible dimensions");
    FrTensor out(X.size / inputSize * outputSize);
    zkdl::check(zk_fc_forward(zkdl::ctx(), h_, X.gpu_data, X.size, out.gpu_data));
    return out;
  }
  void prove(const FrTensor& X, const FrTensor& Z, Commitment& generators) const {   // zkfc.cu:128-145
    if (X.size % inputSize != 0) throw std::runtime_error("Incompatible dimensions 1");
    uint batchSize = X.size / inputSize;
    auto u_bs = random_vec(ceilLog2(batchSize));          // draw order: zkfc.cu:135-137
    auto u_in_dim = random_vec(ceilLog2(inputSize));
    auto u_out_dim = random_vec(ceilLog2(outputSize));
    zkdl::check(zk_fc_prove(zkdl::ctx(), h_, X.gpu_data, X.size, Z.gpu_data, Z.size, generators.handle(), u_bs.data(), u_in_dim.data(), u_out_dim.data(),
                            zkdl::transcript()));
  }
  // additive: the round-by-round Fiat-Shamir prover (include/zkdl_amd.h: zk_fc_prove_fs).  u_bs and u_out_dim are derived from the statement
  // digest, every round challenge of the inner-product sumcheck (u_in_dim) from a hash of the rounds before it, on the device.
  void prove_fs(const FrTensor& X, const FrTensor& Z, Commitment& generators, const uint8_t seed32[32], uint layer) const {
    if (X.size % inputSize != 0) throw std::runtime_error("Incompatible dimensions 1");
    zkdl::check(zk_fc_prove_fs(zkdl::ctx(), h_, X.gpu_data, X.size, Z.gpu_data, Z.size, generators.handle(), seed32, layer, zkdl::transcript()));
  }
  // additive (the reference has no verifier): check this layer's slice of a serialised proof against the layer's public
  // commitment; challenges in zkfc.cu:135-137's order.  Returns the first failed check's message, empty when it verifies.
  std::string verify(uint batch_size, const Commitment& generators, bool z_is_raw, const std::vector<Fr_t>& u_bs, const std::vector<Fr_t>& u_in_dim,
                     const std::vector<Fr_t>& u_out_dim, const Fr_t* fr, size_t n_fr, const G1Affine_t* g1, const uint8_t* inf, size_t n_g1,
                     const Fr_t* expect_z = nullptr, const Fr_t* expect_x = nullptr) const {
    int ok = 0;
    zkdl::check(zk_verify_fc(zkdl::ctx(), generators.handle(), zk_fc_com(h_), zk_fc_com_size(h_), batch_size, inputSize, outputSize, z_is_raw ? 1 : 0, u_bs.data(),
                             u_in_dim.data(), u_out_dim.data(), fr, n_fr, g1, inf, n_g1, expect_z, expect_x, nullptr, nullptr, &ok));
    return ok ? std::string() : std::string(zk_last_error(zkdl::ctx()));
  }
  static zkFC from_float_gpu_ptr(uint input_size, uint output_size, float* float_gpu_ptr, const Commitment& generators) {   // zkfc.cu:90-100
    zk_fc* h = nullptr;
    zkdl::check(zk_fc_from_float(zkdl::ctx(), input_size, output_size, float_gpu_ptr, generators.handle(), &h));
    return zkFC(h);
  }
  static FrTensor load_float_gpu_input(uint batch_size, uint input_dim, float* input_ptr) {   // zkfc.cu:106-115
    FrTensor t((1u << ceilLog2(batch_size)) * (1u << ceilLog2(input_dim)));
    zkdl::check(zk_fc_load_input(zkdl::ctx(), batch_size, input_dim, input_ptr, t.gpu_data));
    return t;
  }
};

// ------------------------------------------------------------------ zkReLU (zkrelu.cuh:10-20)
class zkReLU {
 protected:
  zk_relu* h_ = nullptr;
  uint n_ = 0;
  uint kind_ = 0, shift_ = 0;   // additive: 0 = ReLU (the reference's only activation), 1 = LeakyReLU with negative slope 2^-shift (0: identity)

 public:
  zkReLU() { zkdl::check(zk_relu_create(zkdl::ctx(), &h_)); }
  zkReLU(const zkReLU&) = delete;   // the reference never copies it (raw owning pointers, demo.cu:103)
  zkReLU(zkReLU&& o) noexcept : h_(o.h_), n_(o.n_), kind_(o.kind_), shift_(o.shift_) { o.h_ = nullptr; }
  // additive: the activation this layer applies from its next operator() on
  void set_activation(uint kind, uint shift) {
    zkdl::check(zk_relu_set_activation(h_, kind, shift));
    kind_ = kind;
    shift_ = kind ? shift : 0;
  }
  bool leaky() const { return kind_ == 1; }
  ~zkReLU() { if (h_) zk_relu_destroy(h_); }
  FrTensor operator()(const FrTensor& X) {   // zkrelu.cu:44-51
    FrTensor out(X.size);
    zkdl::check(zk_relu_forward(zkdl::ctx(), h_, X.gpu_data, X.size, out.gpu_data));
    n_ = X.size;
    return out;
  }
  void prove(const FrTensor& X, const FrTensor& Z) {   // zkrelu.cu:79-100
    if (X.size != Z.size) throw std::runtime_error("Incompatible dimensions");
    uint log_size = ceilLog2(X.size);
    auto u_z_bin = random_vec(log_size + 5), v_z_bin = random_vec(log_size + 5);
    auto u_r_bin = random_vec(log_size + 4), v_r_bin = random_vec(log_size + 4);
    auto u_recover = random_vec(log_size);
    auto u_hp = random_vec(log_size), v_hp = random_vec(log_size);
    if (kind_ == 1) {   // additive: one more Hadamard sumcheck for the negative branch (zk_leaky_prove)
      auto u_lk = random_vec(log_size), v_lk = random_vec(log_size);
      zkdl::check(zk_leaky_prove(zkdl::ctx(), h_, X.gpu_data, Z.gpu_data, X.size, u_z_bin.data(), v_z_bin.data(), u_r_bin.data(), v_r_bin.data(), u_recover.data(),
                                 u_hp.data(), v_hp.data(), u_lk.data(), v_lk.data(), zkdl::transcript()));
      return;
    }
    zkdl::check(zk_relu_prove(zkdl::ctx(), h_, X.gpu_data, Z.gpu_data, X.size, u_z_bin.data(), v_z_bin.data(), u_r_bin.data(), v_r_bin.data(), u_recover.data(),
                              u_hp.data(), v_hp.data(), zkdl::transcript()));
  }
  // additive: round-by-round Fiat-Shamir (zk_relu_prove_fs / zk_leaky_prove_fs): v_z, v_r, v_hp (and v_lk) are hashes of the rounds before them
  void prove_fs(const FrTensor& X, const FrTensor& Z, const uint8_t seed32[32], uint layer) {
    if (X.size != Z.size) throw std::runtime_error("Incompatible dimensions");
    if (kind_ == 1) {   // one more Hadamard sumcheck, its round challenges the hash chain which = 4 (zk_leaky_prove_fs)
      zkdl::check(zk_leaky_prove_fs(zkdl::ctx(), h_, X.gpu_data, Z.gpu_data, X.size, seed32, layer, zkdl::transcript()));
      return;
    }
    zkdl::check(zk_relu_prove_fs(zkdl::ctx(), h_, X.gpu_data, Z.gpu_data, X.size, seed32, layer, zkdl::transcript()));
  }
  // additive: the verifier's checks of one zkReLU::prove slice (challenge vectors in zkrelu.cu:85-98's order; u_recover enters no
  // check); claims5 receives the five evaluations the reference leaves unbound
  static std::string verify(uint n, const std::vector<std::vector<Fr_t>>& ch, const Fr_t* fr, size_t n_fr, Fr_t* claims5 = nullptr) {
    if (ch.size() != 7) throw std::runtime_error("Incompatible dimensions");
    int ok = 0;
    zkdl::check(zk_verify_relu(zkdl::ctx(), n, ch[0].data(), ch[1].data(), ch[2].data(), ch[3].data(), ch[5].data(), ch[6].data(), fr, n_fr, claims5, &ok));
    return ok ? std::string() : std::string(zk_last_error(zkdl::ctx()));
  }
  // the same for a LeakyReLU slice (nine vectors: the seven above, then u_lk, v_lk); claims8 = the ReLU's five + (low o nsign)~, low~, nsign~
  static std::string verify_leaky(uint n, const std::vector<std::vector<Fr_t>>& ch, const Fr_t* fr, size_t n_fr, Fr_t* claims8 = nullptr) {
    if (ch.size() != 9) throw std::runtime_error("Incompatible dimensions");
    int ok = 0;
    zkdl::check(zk_verify_leaky(zkdl::ctx(), n, ch[0].data(), ch[1].data(), ch[2].data(), ch[3].data(), ch[5].data(), ch[6].data(), ch[7].data(), ch[8].data(), fr, n_fr,
                                claims8, &ok));
    return ok ? std::string() : std::string(zk_last_error(zkdl::ctx()));
  }
};

// ------------------------------------------------------------------ additive: the pipelined prover (what bench.py's timed loop does, in C++)
// The reference's driver proves ONE statement and waits (demo.cu:124-140).  A prover that serves a stream of statements keeps several
// proofs in flight: every proof is recorded into its own transcript and SEALED (zk_transcript_seal) -- its G1 phase (MSM batch, derived round
// commitments, canonical form, digest) then runs on the library's G1 stream under the Fr phases of the following proofs; `g1_batch` sealed
// proofs share one G1 batch; `inflight` transcripts rotate.  Every proof leaves its digest (8 x u64 lanes) in its own row of a device
// buffer; finish() waits for all of them, sums the rows on the device (zk_dist_sum_rows), optionally all-reduces the sum over RCCL
// (zk_dist_allreduce_u64: independent proof batches per GPU, ONE collective per run) and returns the aggregate as a field element.
//   zkdl::ProofPipeline pipe(K);                       // throughput configuration: 8 in flight, 3 per G1 batch, derived round commitments
//   for (k = 0; k < K; k++) pipe.submit([&] { fcs[L-1].prove(...); for (i = L-2 .. 0) { relus[i].prove(...); fcs[i].prove(...); } });
//   Fr_t aggregate = pipe.finish();
// The prove calls inside submit() are the reference's, unchanged: they append to zkdl::transcript(), which is the pipeline's current
// transcript while the body runs.  Pure scheduling: every transcript holds the same values as a synchronous proof of the same statement
// with the same challenges (tests/test_gpu_host_api.py compares the run aggregate with bench.py's).
namespace zkdl {
class ProofPipeline {
  std::vector<zk_transcript*> trs_;
  uint64_t* rows_ = nullptr;   // [capacity][8] on the device
  size_t alloc_, cap_, n_ = 0, drain_ = 0;
  uint g1_batch_;
  bool derive_, drained_ = false;

 public:
  // capacity = proofs until finish(); the last `drain` of them (default: one G1 batch) take the short G1 form -- every round commitment an MSM,
  // G1 phases unbatched -- because they have no later proof to hide the longer derived form under
  explicit ProofPipeline(size_t capacity, uint inflight = 8, uint g1_batch = 3, bool derive_rounds = true, int drain = -1)
      : alloc_(capacity ? capacity : 1), cap_(alloc_), g1_batch_(g1_batch ? g1_batch : 1), derive_(derive_rounds) {
    drain_ = derive_ ? (drain < 0 ? g1_batch_ : (size_t)drain) : 0;
    trs_.resize(inflight < 2 ? 2 : inflight);
    for (auto& t : trs_) check(zk_transcript_create(ctx(), &t));
    rows_ = dalloc<uint64_t>(8 * cap_);
    std::vector<uint64_t> z(8 * cap_, 0);
    check(zk_memcpy_h2d(ctx(), rows_, z.data(), z.size() * sizeof(uint64_t)));
    configure(false);
  }
  ProofPipeline(const ProofPipeline&) = delete;
  ~ProofPipeline() {
    current_transcript() = nullptr;
    for (auto t : trs_) if (t) zk_transcript_destroy(t);
    zk_free(ctx_slot(), rows_);
    zk_set_flag(ctx(), "derive_rounds", 0);   // back to the library's defaults (synchronous use)
    zk_set_flag(ctx(), "g1_batch", 1);
    zk_set_flag(ctx(), "g1_streams", 0);
  }
  void configure(bool drain_form) {
    check(zk_set_flag(ctx(), "derive_rounds", derive_ && !drain_form ? 1 : 0));
    check(zk_set_flag(ctx(), "g1_batch", drain_form ? 1 : (int)g1_batch_));
    check(zk_set_flag(ctx(), "g1_streams", g1_batch_ > 1 ? 1 : 0));   // batched G1 phases want ONE G1 stream
  }
  size_t submitted() const { return n_; }
  size_t capacity() const { return cap_; }
  // the transcript the k-th submitted proof went to (valid until `inflight` more proofs have been submitted)
  zk_transcript* transcript_of(size_t k) const { return trs_[k % trs_.size()]; }
  const uint64_t* rows() const { return rows_; }
  template <class Body> void submit(Body&& body) {
    if (n_ >= cap_) throw std::runtime_error("zkdl::ProofPipeline: capacity exhausted");
    if (!drained_ && drain_ && n_ + drain_ >= cap_) { configure(true); drained_ = true; }
    zk_transcript* tr = trs_[n_ % trs_.size()];
    check(zk_transcript_reset(tr));          // waits for this transcript's own earlier G1 phase only
    current_transcript() = tr;
    try { body(); } catch (...) { current_transcript() = nullptr; throw; }
    current_transcript() = nullptr;
    check(zk_transcript_seal(tr, rows_ + 8 * n_));
    ++n_;
  }
  // waits for every sealed proof, sums the digest rows on the device, all-reduces the sum over `dist` (if given), returns the aggregate
  Fr_t finish(zk_dist* dist = nullptr) {
    uint64_t* agg = dalloc<uint64_t>(8);
    uint64_t lanes[8];
    check(zk_dist_sum_rows(ctx(), rows_, n_, agg));
    if (dist) check(zk_dist_allreduce_u64(dist, agg, 8));
    check(zk_memcpy_d2h(ctx(), lanes, agg, sizeof lanes));
    zk_free(ctx(), agg);
    Fr_t out;
    zk_dist_lanes_to_fr(lanes, &out);
    return out;
  }
  // start over with the same transcripts (the next run's rows overwrite this one's)
  void rewind(size_t capacity = 0) {
    check(zk_ctx_sync(ctx()));
    if (capacity > alloc_) throw std::runtime_error("zkdl::ProofPipeline: rewind beyond the allocated rows");
    if (capacity) cap_ = capacity;
    n_ = 0;
    drained_ = false;
    configure(false);
  }
};
}  // namespace zkdl

// ------------------------------------------------------------------ Timer (timer.hpp:6-22)
#include <chrono>
class Timer {
  std::chrono::high_resolution_clock::time_point startTime;
  std::chrono::duration<double> totalTime{0.0};
  bool isRunning = false;

 public:
  void start() { if (!isRunning) { startTime = std::chrono::high_resolution_clock::now(); isRunning = true; } }
  void stop() {
    if (isRunning) {
      totalTime += std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::high_resolution_clock::now() - startTime);
      isRunning = false;
    }
  }
  void reset() { isRunning = false; totalTime = std::chrono::duration<double>(0.0); }
  double getTotalTime() const { return totalTime.count(); }
};
