// zkdl_amd/csrc/verify.hip -- the verifier's side of zkFC::prove / zkReLU::prove, and hash-derived challenges.
//
// The reference computes every proof value and drops it (zkfc.cu:139-144, zkrelu.cu:91-99) and draws its challenges from
// std::random_device before any round runs (proof.cu:3-11, zkfc.cu:135-137, zkrelu.cu:85-98); nothing in it verifies.
// Here the values are kept (zk_transcript); this file checks them WITHOUT the witness:
//
//   zk_verify_fc    inner-product sumcheck recurrence and final claim (proof.cu:55-96), the opening's scalar against the
//                   sumcheck's b~ value, and Commitment::open / me_open (commitment.cu:43-92) against the layer's public
//                   commitment: com(u_hi) recomputed from the commitment points, T_0 = [R] com(u_hi), the folding relation
//                   T_{r+1} = u(1-u) T_r + (1-u)^2 T0_r + u^2 T1_r of every round, T_last' = g_final * t(0), and g_final
//                   recomputed from the generators.  Inputs: generators, commitment points, challenges, the proof slice.
//   zk_verify_relu  both binary sumchecks (proof.cu:152-188: claim 0, eq-weighted recurrence, a^2 - a at the end) and the
//                   Hadamard sumcheck (proof.cu:110-137).  The reference's ReLU protocol binds none of its final
//                   evaluations (bit tensors, sign, Z, the 48 partial_me values) to a commitment -- activation
//                   commitments are its stated future work (README) -- so these come back to the caller as open claims.
//
// Field work (a few hundred products) runs on the host with the same field.hpp code the kernels use; the two
// multilinear evaluations over G1 (commitment rows, generators) and the 3k+2 scalar multiplications run on the GPU through
// the LITERAL kernels (g1_me_dev, g1_mul_dev: the reference's op sequence), i.e. not through the bucket-MSM code that
// produced the proof.
//
// zk_fs_derive: challenges as SHA-256 outputs in the reference's distribution (8 uniform limbs, top limb mod 1944954707).
// They are derived PER PROVE CALL from a caller-supplied statement digest and the call's (layer, kind, vector) tags, not per
// round: like the reference's pre-drawn coins they do not depend on the call's own round messages, which is what lets a
// proof's 15 calls run as independent, zipped chains -- and also means this is a reproducible public-coin
// schedule, not a sound non-interactive argument by itself (a round-by-round transform would have to serialise every round
// behind a hash of the previous one and gives up the fused rounds).
#include <string.h>

#include <string>
#include <vector>

#include "layers.hpp"

using namespace zk;

// ------------------------------------------------------------------ SHA-256 (FIPS 180-4), host only
namespace {
struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len = 0;
  size_t fill = 0;
  Sha256() {
    static const uint32_t init[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h, init, sizeof h);
  }
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u,
        0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du,
        0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu,
        0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u,
        0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    len += n;
    while (n) {
      size_t take = 64 - fill < n ? 64 - fill : n;
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void finish(uint8_t out[32]) {
    uint64_t bits = len * 8;
    uint8_t pad = 0x80;
    update(&pad, 1);
    uint8_t z = 0;
    while (fill != 56) update(&z, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; i++) {
      out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
    }
  }
};

// ---- host field helpers (Montgomery domain; fr_mul is the CIOS host path of field.hpp)
inline Fr ld(const zk_fr* p) {
  Fr r;
  memcpy(&r, p, sizeof r);
  return r;
}
inline G1A lda(const zk_g1a* p) {  // caller buffers need not be 16-byte aligned (G1A is)
  G1A r;
  memcpy(&r, p, sizeof r);
  return r;
}
// Proof values are field elements in their ONE canonical encoding (Montgomery limbs below r).  c and c + r are the same element but different
// bytes: hashed as stored (sound mode, sha256.hpp: fs_round_step) they would give different challenges, and fr_add / l_eq have no defined
// behaviour on unreduced inputs.  Index of the first value >= r, or -1.
inline long first_unreduced(const zk_fr* p, size_t n) {
  Fr rm;
  for (int i = 0; i < 8; i++) rm.v[i] = FrP::P(i);
  for (size_t i = 0; i < n; i++)
    if (l_gte(ld(p + i), rm)) return (long)i;
  return -1;
}
inline Fr poly2(const Fr& c0, const Fr& c1, const Fr& c2, const Fr& x) { return fr_add(c0, fr_mul(x, fr_add(c1, fr_mul(x, c2)))); }
inline bool feq(const Fr& a, const Fr& b) { return l_eq(a, b); }

struct Fail {
  zk_ctx* ctx;
  int* ok;
  int operator()(const std::string& what) const {
    ctx->last_error = "verification failed: " + what;
    *ok = 0;
    return ZK_OK;
  }
};

// affine transcript point (x, y, inf) == Jacobian point computed by the verifier?   X = x Z^2, Y = y Z^3
bool same_point(const G1J& j, const G1A& a, bool a_inf) {
  const bool j_inf = l_is_zero(j.z);
  if (j_inf || a_inf) return j_inf && a_inf;
  const Fp z2 = fp_sqr(j.z);
  return l_eq(j.x, fp_mul(a.x, z2)) && l_eq(j.y, fp_mul(a.y, fp_mul(z2, j.z)));
}
// A prover-supplied affine point must be a canonical encoding of a curve point: both coordinates below p and y^2 = x^3 + 4
// (Montgomery form; the a = 0 group law never reads b, so an off-curve point would be processed silently otherwise).
// Infinity is encoded as all-zero coordinates + flag.  Membership in the prime-order subgroup is checked by the caller ([r] P = O).
bool valid_encoding(const G1A& a, bool inf) {
  if (inf) return l_is_zero(a.x) && l_is_zero(a.y);
  Fp pm;
  for (int i = 0; i < 12; i++) pm.v[i] = FpP::P(i);
  if (l_gte(a.x, pm) || l_gte(a.y, pm)) return false;
  Fp four = fp_one();
  four = fp_dbl(fp_dbl(four));
  return l_eq(fp_sqr(a.y), fp_add(fp_mul(fp_sqr(a.x), a.x), four));
}
G1J from_affine(const G1A& a, bool inf) {
  G1J r = g1_zero();
  if (!inf) {
    r.x = a.x;
    r.y = a.y;
    r.z = fp_one();
  }
  return r;
}
}  // namespace

namespace {
// Commitment::open / me_open (commitment.cu:43-92) against a public commitment: g1 = [com(u_hi)][T, T0, T1 per round][g_final].
// u: k = k_lo + k_hi challenges, the first k_lo fold the generator index, the last k_hi the commitment rows (commitment.cu:85-86);
// t0: the opened value (Montgomery form) = the committed tensor's multilinear evaluation at u.
int verify_opening(zk_ctx* ctx, const Fail& fail, zk_gens* gens, const zk_g1j* com_d, uint32_t m, const zk_fr* u_h, uint32_t k, const Fr& t0, const zk_g1a* g1_h,
                   const uint8_t* inf_h, bool* passed) {
  *passed = false;
  const uint32_t n_gens = gens->n;
  const uint32_t k_lo = ceil_log2(n_gens), k_hi = ceil_log2(m);
  // me_open folds the generators the other way round from the scalars, g' = g1 + u (g0 - g1) (commitment.cu:54-56): the final
  // generator is the multilinear evaluation of the generators at 1 - u
  std::vector<zk_fr> u(k + k_lo + 1);
  for (uint32_t i = 0; i < k; i++) u[i] = u_h[i];
  for (uint32_t i = 0; i < k_lo; i++) {
    const Fr c = fr_sub(fr_one(), ld(&u[i]));
    memcpy(&u[k + i], &c, sizeof c);
  }
  DBuf<Fr> ud;
  ZK_TRY(upload_fr(ctx, ud, u.data(), k + k_lo));
  ZK_ALLOC(ctx, evals, G1J, 2);
  ZK_TRY(g1_me_dev(ctx, reinterpret_cast<const G1J*>(com_d), m, ud.get() + k_lo, k_hi, evals.get()));          // com(u_hi)
  ZK_TRY(g1_me_dev(ctx, gens->pts, n_gens, ud.get() + k, k_lo, evals.get() + 1));                               // generators(1 - u_lo)
  // the 3 k_lo + 2 scalar multiplications of the folding relation, as one batched launch of the literal kernel:
  //   [R] com_eval | per round: u(1-u) T_r, (1-u)^2 T0_r, u^2 T1_r | t(0) g_final      (scalars = raw integers)
  const size_t nmul = 3 * (size_t)k_lo + 2;
  for (size_t i = 0; i < nmul; i++)
    if (!valid_encoding(lda(&g1_h[i]), inf_h[i] != 0)) return fail("transcript point " + std::to_string(i) + " is not a canonical encoding of a point on y^2 = x^3 + 4");
  // the second half of the batch is [r] P for every transcript point: all must be the identity (prime-order subgroup)
  std::vector<G1J> pts(2 * nmul);
  std::vector<Fr> sc(2 * nmul);
  auto tp = [&](size_t i) { return from_affine(lda(&g1_h[i]), inf_h[i] != 0); };
  pts[0] = tp(0);
  sc[0] = fr_one();  // raw limbs of the Montgomery one = R mod r
  const Fr one = fr_one();
  for (uint32_t r = 0; r < k_lo; r++) {
    const Fr ur = ld(&u[r]), nur = fr_sub(one, ur);
    pts[1 + 3 * r] = tp(1 + 3 * r);
    pts[2 + 3 * r] = tp(2 + 3 * r);
    pts[3 + 3 * r] = tp(3 + 3 * r);
    sc[1 + 3 * r] = fr_unmont(fr_mul(ur, nur));
    sc[2 + 3 * r] = fr_unmont(fr_mul(nur, nur));
    sc[3 + 3 * r] = fr_unmont(fr_mul(ur, ur));
  }
  pts[nmul - 1] = tp(3 * (size_t)k_lo + 1);
  sc[nmul - 1] = t0;  // me_open multiplies by the RAW limbs of the (Montgomery-form) scalars (commitment.cu:57-59)
  {
    Fr rmod;
    for (int i = 0; i < 8; i++) rmod.v[i] = FrP::P(i);  // the scalar multiplication runs on raw limbs (g1-tensor.cu:422-430): k = r is fine
    for (size_t i = 0; i < nmul; i++) {
      pts[nmul + i] = tp(i);
      sc[nmul + i] = rmod;
    }
  }
  ZK_ALLOC(ctx, pd, G1J, 2 * nmul);
  ZK_ALLOC(ctx, sd, Fr, 2 * nmul);
  ZK_ALLOC(ctx, od, G1J, 2 * nmul);
  ZK_TRY(host_to_dev(ctx, pd.get(), pts.data(), 2 * nmul * sizeof(G1J)));
  ZK_TRY(host_to_dev(ctx, sd.get(), sc.data(), 2 * nmul * sizeof(Fr)));
  ZK_TRY(g1_mul_dev(ctx, pd.get(), (uint32_t)(2 * nmul), sd.get(), (uint32_t)(2 * nmul), od.get()));
  std::vector<G1J> prod(2 * nmul);
  G1J ev[2];
  ZK_TRY(dev_to_host(ctx, prod.data(), od.get(), 2 * nmul * sizeof(G1J)));
  ZK_TRY(dev_to_host(ctx, ev, evals.get(), sizeof ev));
  for (size_t i = 0; i < nmul; i++)
    if (!l_is_zero(prod[nmul + i].z)) return fail("transcript point " + std::to_string(i) + " is not in the prime-order subgroup ([r] P != O)");
  if (!same_point(ev[0], lda(&g1_h[0]), inf_h[0] != 0)) return fail("com(u_hi) is not the multilinear evaluation of the commitment");
  if (!same_point(ev[1], lda(&g1_h[3 * (size_t)k_lo + 1]), inf_h[3 * (size_t)k_lo + 1] != 0))
    return fail("the final generator is not the multilinear evaluation of the generators");
  G1J cur = prod[0];  // commitment of the round's scalars under the round's generators, as it must be
  for (uint32_t r = 0; r < k_lo; r++) {
    if (!same_point(cur, lda(&g1_h[1 + 3 * r]), inf_h[1 + 3 * r] != 0))
      return fail(r == 0 ? std::string("T_0 does not commit to the row-folded tensor") : "me_open round " + std::to_string(r - 1) + ": folding relation violated");
    cur = g1_add(g1_add(prod[1 + 3 * r], prod[2 + 3 * r]), prod[3 + 3 * r]);
  }
  {
    const G1J want = prod[nmul - 1];  // t(0) * g_final
    const bool ci = l_is_zero(cur.z), wi = l_is_zero(want.z);
    bool same = ci && wi;
    if (!ci && !wi) {
      const Fp z1 = fp_sqr(cur.z), z2 = fp_sqr(want.z);
      same = l_eq(fp_mul(cur.x, z2), fp_mul(want.x, z1)) && l_eq(fp_mul(cur.y, fp_mul(z2, want.z)), fp_mul(want.y, fp_mul(z1, cur.z)));
    }
    if (!same) return fail(k_lo ? "me_open: last folding relation / final opening t(0) * g_final violated" : "opening of a single generator violated");
  }
  *passed = true;
  return ZK_OK;
}
}  // namespace

extern "C" {

int zk_sha256(const void* data, size_t len, uint8_t* out32) {
  if ((!data && len) || !out32) return ZK_ERR_ARG;
  Sha256 s;
  s.update(data, len);
  s.finish(out32);
  return ZK_OK;
}

// out[i] = limbs of SHA-256("zkdl-amd/fs/v1" | seed | layer | kind | vec | i)  (integers little-endian u32; digest bytes read as
// 8 little-endian u32 limbs; top limb reduced mod 1944954707 as random_vec does, proof.cu:9)
int zk_fs_derive(const uint8_t* seed32, uint32_t layer, uint32_t kind, uint32_t vec, uint32_t len, zk_fr* out_h) {
  if (!seed32 || (len && !out_h)) return ZK_ERR_ARG;
  static const char tag[] = "zkdl-amd/fs/v1";
  for (uint32_t i = 0; i < len; i++) {
    Sha256 s;
    s.update(tag, sizeof tag - 1);
    s.update(seed32, 32);
    const uint32_t w[4] = {layer, kind, vec, i};
    uint8_t le[16];
    for (int q = 0; q < 4; q++)
      for (int b = 0; b < 4; b++) le[4 * q + b] = (uint8_t)(w[q] >> (8 * b));
    s.update(le, 16);
    uint8_t d[32];
    s.finish(d);
    zk_fr r;
    for (int l = 0; l < 8; l++) r.v[l] = (uint32_t)d[4 * l] | ((uint32_t)d[4 * l + 1] << 8) | ((uint32_t)d[4 * l + 2] << 16) | ((uint32_t)d[4 * l + 3] << 24);
    r.v[7] %= 1944954707u;
    out_h[i] = r;
  }
  return ZK_OK;
}

int zk_verify_fc(zk_ctx* ctx, zk_gens* gens, const zk_g1j* com_d, uint32_t m, uint32_t batch, uint32_t in, uint32_t out, int z_is_raw, const zk_fr* u_bs_h,
                 const zk_fr* u_in_h, const zk_fr* u_out_h, const zk_fr* fr_h, size_t n_fr, const zk_g1a* g1_h, const uint8_t* inf_h, size_t n_g1, const zk_fr* expect_z_h,
                 const zk_fr* expect_x_h, zk_fr* claim_z_h, zk_fr* claim_x_h, int* ok) {
  if (!ctx) return ZK_ERR_ARG;
  {
    int src = ctx->settle();
    if (src != ZK_OK) return src;
  }
  if (!gens || !com_d || !fr_h || !g1_h || !inf_h || !ok) return ctx->fail(ZK_ERR_ARG, "zk_verify_fc: null argument");
  *ok = 0;
  const uint32_t n_gens = gens->n;
  const uint32_t k_bs = ceil_log2(batch), k_in = ceil_log2(in), k_out = ceil_log2(out), k_lo = ceil_log2(n_gens), k_hi = ceil_log2(m);
  if (in == 0 || out == 0 || batch == 0 || m == 0 || n_gens == 0 || (uint64_t)in * out != (uint64_t)m * n_gens || k_in + k_out != k_lo + k_hi)
    return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (n_fr != 3 * (size_t)k_in + 4 || n_g1 != 3 * (size_t)k_lo + 2 || n_gens != (1u << k_lo)) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if ((k_bs && !u_bs_h) || (k_in && !u_in_h) || (k_out && !u_out_h)) return ctx->fail(ZK_ERR_ARG, "zk_verify_fc: null challenge vector");
  Fail fail{ctx, ok};
  (void)k_bs;
  if (long bad = first_unreduced(fr_h, n_fr); bad >= 0) return fail("transcript value " + std::to_string(bad) + " is not a canonical field element (>= r)");
  // ---- inner-product sumcheck (proof.cu:55-96): Z~(u_out || u_bs) = <X~(., u_bs), W~(u_out, .)>
  const Fr zclaim = ld(fr_h + 3 * k_in + 2);
  if (claim_z_h) memcpy(claim_z_h, &zclaim, sizeof zclaim);
  if (expect_z_h && !feq(zclaim, ld(expect_z_h))) return fail("Z~(u_out || u_bs) differs from the expected evaluation of the layer's output");
  // the last layer is proved on the un-Montgomery'd output (demo.cu:119): the transcript then carries the evaluation of the
  // raw-integer tensor, and the sumcheck's claim is its Montgomery form
  Fr claim = z_is_raw ? fr_mont(zclaim) : zclaim;
  for (uint32_t r = 0; r < k_in; r++) {
    const Fr c0 = ld(fr_h + 3 * r), c1 = ld(fr_h + 3 * r + 1), c2 = ld(fr_h + 3 * r + 2);
    if (!feq(fr_add(c0, fr_add(c0, fr_add(c1, c2))), claim)) return fail("inner-product sumcheck, round " + std::to_string(r) + ": p(0) + p(1) != claim");
    claim = poly2(c0, c1, c2, ld(u_in_h + r));
  }
  const Fr fa = ld(fr_h + 3 * k_in), fb = ld(fr_h + 3 * k_in + 1), t0 = ld(fr_h + 3 * k_in + 3);
  if (!feq(fr_mul(fa, fb), claim)) return fail("inner-product sumcheck: a~ * b~ != final claim");
  if (!feq(fb, t0)) return fail("the opening's scalar differs from the sumcheck's W~(u_out || u_in)");
  if (claim_x_h) memcpy(claim_x_h, &fa, sizeof fa);
  if (expect_x_h && !feq(fa, ld(expect_x_h))) return fail("X~(u_in || u_bs) differs from the expected evaluation of the layer's input");
  // ---- the opening (commitment.cu:43-92) of W at u = u_out || u_in
  {
    std::vector<zk_fr> u(k_out + k_in + 1);
    for (uint32_t i = 0; i < k_out; i++) u[i] = u_out_h[i];
    for (uint32_t i = 0; i < k_in; i++) u[k_out + i] = u_in_h[i];
    bool passed = false;
    ZK_TRY(verify_opening(ctx, fail, gens, com_d, m, u.data(), k_out + k_in, t0, g1_h, inf_h, &passed));
    if (!passed) return ZK_OK;  // *ok = 0 and the message were set by fail
  }
  *ok = 1;
  return ZK_OK;
}

int zk_verify_open(zk_ctx* ctx, zk_gens* gens, const zk_g1j* com_d, uint32_t m, const zk_fr* u_h, uint32_t k, const zk_fr* value_h, const zk_g1a* g1_h,
                   const uint8_t* inf_h, size_t n_g1, int* ok) {
  if (!ctx) return ZK_ERR_ARG;
  {
    int src = ctx->settle();
    if (src != ZK_OK) return src;
  }
  if (!gens || !com_d || !value_h || !g1_h || !inf_h || !ok || (k && !u_h)) return ctx->fail(ZK_ERR_ARG, "zk_verify_open: null argument");
  *ok = 0;
  const uint32_t k_lo = ceil_log2(gens->n), k_hi = ceil_log2(m);
  if (m == 0 || k != k_lo + k_hi || gens->n != (1u << k_lo) || n_g1 != 3 * (size_t)k_lo + 2) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  Fail fail{ctx, ok};
  if (first_unreduced(value_h, 1) >= 0) return fail("the opened value is not a canonical field element (>= r)");
  bool passed = false;
  ZK_TRY(verify_opening(ctx, fail, gens, com_d, m, u_h, k, ld(value_h), g1_h, inf_h, &passed));
  if (passed) *ok = 1;
  return ZK_OK;
}

int zk_verify_relu(zk_ctx* ctx, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h, const zk_fr* v_r_h, const zk_fr* u_hp_h, const zk_fr* v_hp_h,
                   const zk_fr* fr_h, size_t n_fr, zk_fr* claims_h, int* ok) {
  if (!ctx) return ZK_ERR_ARG;
  if (!fr_h || !ok || n == 0) return ctx->fail(ZK_ERR_ARG, "zk_verify_relu: null argument");
  *ok = 0;
  const uint32_t k = ceil_log2(n);
  if (n_fr != 3 * (size_t)(k + 5) + 1 + 32 + 3 * (size_t)(k + 4) + 1 + 16 + 3 * (size_t)k + 2) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (!u_z_h || !v_z_h || !u_r_h || !v_r_h || (k && (!u_hp_h || !v_hp_h))) return ctx->fail(ZK_ERR_ARG, "zk_verify_relu: null challenge vector");
  Fail fail{ctx, ok};
  if (long bad = first_unreduced(fr_h, n_fr); bad >= 0) return fail("transcript value " + std::to_string(bad) + " is not a canonical field element (>= r)");
  const Fr one = fr_one();
  size_t off = 0;
  Fr finals[2];
  // ---- the two binary sumchecks (proof.cu:152-188): sum_g eq(u, g) (a_g^2 - a_g) = 0
  for (int which = 0; which < 2; which++) {
    const uint32_t kk = k + (which == 0 ? 5 : 4);
    const zk_fr* u = which == 0 ? u_z_h : u_r_h;
    const zk_fr* v = which == 0 ? v_z_h : v_r_h;
    const char* name = which == 0 ? "magnitude-bit" : "remainder-bit";
    Fr claim = fr_zero();
    for (uint32_t r = 0; r < kk; r++) {
      const Fr c0 = ld(fr_h + off + 3 * r), c1 = ld(fr_h + off + 3 * r + 1), c2 = ld(fr_h + off + 3 * r + 2);
      const Fr ur = ld(u + r), p1 = fr_add(c0, fr_add(c1, c2));
      if (!feq(fr_add(fr_mul(fr_sub(one, ur), c0), fr_mul(ur, p1)), claim))
        return fail(std::string(name) + " sumcheck, round " + std::to_string(r) + ": (1 - u) p(0) + u p(1) != claim");
      claim = poly2(c0, c1, c2, ld(v + r));
    }
    const Fr fa = ld(fr_h + off + 3 * kk);
    if (!feq(fr_sub(fr_mul(fa, fa), fa), claim)) return fail(std::string(name) + " sumcheck: a~^2 - a~ != final claim");
    finals[which] = fa;
    off += 3 * (size_t)kk + 1 + (which == 0 ? 32 : 16);
  }
  // ---- Hadamard sumcheck (proof.cu:110-137): (Z o sign)~(u_hp); its starting claim is defined by round 0
  Fr claim = fr_zero(), claim0 = fr_zero();
  for (uint32_t r = 0; r < k; r++) {
    const Fr c0 = ld(fr_h + off + 3 * r), c1 = ld(fr_h + off + 3 * r + 1), c2 = ld(fr_h + off + 3 * r + 2);
    const Fr ur = ld(u_hp_h + r), p1 = fr_add(c0, fr_add(c1, c2));
    const Fr lhs = fr_add(fr_mul(fr_sub(one, ur), c0), fr_mul(ur, p1));
    if (r == 0) claim0 = lhs;
    else if (!feq(lhs, claim)) return fail("Hadamard sumcheck, round " + std::to_string(r) + ": (1 - u) p(0) + u p(1) != claim");
    claim = poly2(c0, c1, c2, ld(v_hp_h + r));
  }
  const Fr ha = ld(fr_h + off + 3 * k), hb = ld(fr_h + off + 3 * k + 1);
  if (k == 0) claim0 = claim = fr_mul(ha, hb);
  if (!feq(fr_mul(ha, hb), claim)) return fail("Hadamard sumcheck: Z~ * sign~ != final claim");
  if (claims_h) {  // open claims: mag_bin~(v_z), rem_bin~(v_r), (Z o sign)~(u_hp), Z~(v_hp), sign~(v_hp)
    const Fr outv[5] = {finals[0], finals[1], claim0, ha, hb};
    memcpy(claims_h, outv, sizeof outv);
  }
  *ok = 1;
  return ZK_OK;
}

// LeakyReLU (extension): the ReLU checks on the first part of the slice, then the extra Hadamard sumcheck (low o nsign)~(u_lk).
// claims_h[0..5) as zk_verify_relu; claims_h[5..8) = (low o nsign)~(u_lk), low~(v_lk), nsign~(v_lk) -- open, like the ReLU's.
int zk_verify_leaky(zk_ctx* ctx, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h, const zk_fr* v_r_h, const zk_fr* u_hp_h, const zk_fr* v_hp_h,
                    const zk_fr* u_lk_h, const zk_fr* v_lk_h, const zk_fr* fr_h, size_t n_fr, zk_fr* claims_h, int* ok) {
  if (!ctx) return ZK_ERR_ARG;
  if (!fr_h || !ok || n == 0) return ctx->fail(ZK_ERR_ARG, "zk_verify_leaky: null argument");
  *ok = 0;
  const uint32_t k = ceil_log2(n);
  const size_t relu_len = 3 * (size_t)(k + 5) + 1 + 32 + 3 * (size_t)(k + 4) + 1 + 16 + 3 * (size_t)k + 2;
  if (n_fr != relu_len + 3 * (size_t)k + 2) return ctx->fail(ZK_ERR_SHAPE, "Incompatible dimensions");
  if (k && (!u_lk_h || !v_lk_h)) return ctx->fail(ZK_ERR_ARG, "zk_verify_leaky: null challenge vector");
  if (long bad = first_unreduced(fr_h, n_fr); bad >= 0) {
    ctx->last_error = "verification failed: transcript value " + std::to_string(bad) + " is not a canonical field element (>= r)";
    return ZK_OK;
  }
  int ok1 = 0;
  ZK_TRY(zk_verify_relu(ctx, n, u_z_h, v_z_h, u_r_h, v_r_h, u_hp_h, v_hp_h, fr_h, relu_len, claims_h, &ok1));
  if (!ok1) return ZK_OK;
  Fail fail{ctx, ok};
  const Fr one = fr_one();
  const size_t off = relu_len;
  Fr claim = fr_zero(), claim0 = fr_zero();
  for (uint32_t r = 0; r < k; r++) {
    const Fr c0 = ld(fr_h + off + 3 * r), c1 = ld(fr_h + off + 3 * r + 1), c2 = ld(fr_h + off + 3 * r + 2);
    const Fr ur = ld(u_lk_h + r), p1 = fr_add(c0, fr_add(c1, c2));
    const Fr lhs = fr_add(fr_mul(fr_sub(one, ur), c0), fr_mul(ur, p1));
    if (r == 0) claim0 = lhs;
    else if (!feq(lhs, claim)) return fail("negative-branch Hadamard sumcheck, round " + std::to_string(r) + ": (1 - u) p(0) + u p(1) != claim");
    claim = poly2(c0, c1, c2, ld(v_lk_h + r));
  }
  const Fr ha = ld(fr_h + off + 3 * k), hb = ld(fr_h + off + 3 * k + 1);
  if (k == 0) claim0 = claim = fr_mul(ha, hb);
  if (!feq(fr_mul(ha, hb), claim)) return fail("negative-branch Hadamard sumcheck: low~ * nsign~ != final claim");
  if (claims_h) {
    const Fr outv[3] = {claim0, ha, hb};
    memcpy(claims_h + 5, outv, sizeof outv);
  }
  *ok = 1;
  return ZK_OK;
}

}  // extern "C"
