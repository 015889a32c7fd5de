/*
 * oracle/zkdl_oracle.c -- TEST INFRASTRUCTURE ONLY (see zkdl_oracle.h).
 *
 * CPU restatement of the reference algorithm for the proving hot path.  Written from the
 * reference's behaviour, not its text: one generic N-limb Montgomery core instead of the
 * generated per-field copies, loops instead of kernels + recursion.  Every function names the
 * reference lines it restates.  Arithmetic is 32-bit-limb, little-endian, Montgomery form with
 * R = 2^256 (Fr) / 2^384 (Fp), exactly as bls12-381.cuh:116-124,218-226.
 */
#include "zkdl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ constants */
/* bls12-381.cu:3-11 (public BLS12-381 parameters in the reference's limb order) */
static const zo_fr FR_ONE = {{4294967294u, 1u, 215042u, 1485092858u, 3971764213u, 2576109551u, 2898593135u, 405057881u}};
static const zo_fr FR_P = {{1u, 4294967295u, 4294859774u, 1404937218u, 161601541u, 859428872u, 698187080u, 1944954707u}};
static const zo_fr FR_R2 = {{4092763245u, 3382307216u, 2274516003u, 728559051u, 1918122383u, 97719446u, 2673475345u, 122214873u}};
static const zo_fr FR_0 = {{0, 0, 0, 0, 0, 0, 0, 0}};
#define FR_INV 4294967295u /* bls12-381.cuh:119 */
static const zo_fp FP_ONE = {{196605u, 1980301312u, 3289120770u, 3958636555u, 1405573306u, 1598593111u, 1884444485u, 2010011731u, 2723605613u, 1543969431u, 4202751123u, 368467651u}};
static const zo_fp FP_P = {{4294945451u, 3120496639u, 2975072255u, 514588670u, 4138792484u, 1731252896u, 4085584575u, 1685539716u, 1129032919u, 1260103606u, 964683418u, 436277738u}};
static const zo_fp FP_R2 = {{473175878u, 4108263220u, 164693233u, 175564454u, 1284880085u, 2380613484u, 2476573632u, 1743489193u, 3038352685u, 2591637125u, 2462770090u, 295210981u}};
static const zo_fp FP_0 = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
#define FP_INV 4294770685u /* bls12-381.cuh:221 */
/* g1-tensor.cuh:29-58 */
static const zo_fp G1_GEN_X = {{4250078230u, 1555269520u, 2574712821u, 2014837863u, 339452353u, 357537223u, 4090554183u, 4037962445u, 568063040u, 3989728972u, 2651585397u, 302085953u}};
static const zo_fp G1_GEN_Y = {{216474225u, 3131872213u, 2031680910u, 2351063834u, 1460086222u, 3713621779u, 1346392468u, 1370249257u, 2902481344u, 236751935u, 1342743146u, 196886268u}};

/* ------------------------------------------------------------------ N-limb core */
/* gte: bls12-381.cu:245-253 / :628-636 (most-significant limb first; equal => true) */
static inline int l_gte(const uint32_t* a, const uint32_t* b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static inline int l_eq(const uint32_t* a, const uint32_t* b, int n) { return memcmp(a, b, 4 * (size_t)n) == 0; }
static inline int l_is_zero(const uint32_t* a, int n) {
  uint32_t acc = 0;
  for (int i = 0; i < n; i++) acc |= a[i];
  return acc == 0;
}
/* wrapping add/sub: bls12-381.cu:268-285 / :675-692 */
static inline void l_add_(uint32_t* r, const uint32_t* a, const uint32_t* b, int n) {
  uint64_t c = 0;
  for (int i = 0; i < n; i++) { c += (uint64_t)a[i] + b[i]; r[i] = (uint32_t)c; c >>= 32; }
}
static inline void l_sub_(uint32_t* r, const uint32_t* a, const uint32_t* b, int n) {
  uint64_t bw = 0;
  for (int i = 0; i < n; i++) { uint64_t d = (uint64_t)a[i] - b[i] - bw; r[i] = (uint32_t)d; bw = (d >> 32) & 1; }
}
/* modular add/sub: bls12-381.cu:289-300 / :696-707 */
static inline void l_mod_add(uint32_t* r, const uint32_t* a, const uint32_t* b, const uint32_t* p, int n) {
  uint32_t t[12];
  l_add_(t, a, b, n);
  if (l_gte(t, p, n)) l_sub_(t, t, p, n);
  memcpy(r, t, 4 * (size_t)n);
}
static inline void l_mod_sub(uint32_t* r, const uint32_t* a, const uint32_t* b, const uint32_t* p, int n) {
  uint32_t t[12];
  int ge = l_gte(a, b, n);
  l_sub_(t, a, b, n);
  if (!ge) l_add_(t, t, p, n);
  memcpy(r, t, 4 * (size_t)n);
}
/* double: bls12-381.cu:550-556 / :957-963 */
static inline void l_mod_double(uint32_t* r, const uint32_t* a, const uint32_t* p, int n) {
  uint32_t t[12];
  for (int i = n - 1; i >= 1; i--) t[i] = (a[i] << 1) | (a[i - 1] >> 31);
  t[0] = a[0] << 1;
  if (l_gte(t, p, n)) l_sub_(t, t, p, n);
  memcpy(r, t, 4 * (size_t)n);
}
/* CIOS Montgomery product: bls12-381.cu:499-530 / :906-937 */
static inline void l_mont_mul(uint32_t* r, const uint32_t* a, const uint32_t* b, const uint32_t* p, uint32_t inv, int n) {
  uint32_t t[14];
  memset(t, 0, sizeof t);
  for (int i = 0; i < n; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < n; j++) {
      uint64_t s = (uint64_t)a[j] * b[i] + t[j] + carry;
      t[j] = (uint32_t)s; carry = s >> 32;
    }
    uint64_t s = (uint64_t)t[n] + carry;
    t[n] = (uint32_t)s; t[n + 1] = (uint32_t)(s >> 32);
    uint32_t m = inv * t[0];
    s = (uint64_t)m * p[0] + t[0];
    carry = s >> 32;
    for (int j = 1; j < n; j++) {
      s = (uint64_t)m * p[j] + t[j] + carry;
      t[j - 1] = (uint32_t)s; carry = s >> 32;
    }
    s = (uint64_t)t[n] + carry;
    t[n - 1] = (uint32_t)s;
    t[n] = t[n + 1] + (uint32_t)(s >> 32);
  }
  if (l_gte(t, p, n)) l_sub_(t, t, p, n);
  memcpy(r, t, 4 * (size_t)n);
}

/* ------------------------------------------------------------------ Fr / Fp */
#define FR(x) ((x)->v)
void zo_fr_add(const zo_fr* a, const zo_fr* b, zo_fr* r) { l_mod_add(r->v, a->v, b->v, FR_P.v, 8); }
void zo_fr_sub(const zo_fr* a, const zo_fr* b, zo_fr* r) { l_mod_sub(r->v, a->v, b->v, FR_P.v, 8); }
void zo_fr_mul(const zo_fr* a, const zo_fr* b, zo_fr* r) { l_mont_mul(r->v, a->v, b->v, FR_P.v, FR_INV, 8); }
void zo_fr_sqr(const zo_fr* a, zo_fr* r) { zo_fr_mul(a, a, r); }                  /* bls12-381.cu:544-546 */
void zo_fr_double(const zo_fr* a, zo_fr* r) { l_mod_double(r->v, a->v, FR_P.v, 8); }
void zo_fr_mont(const zo_fr* a, zo_fr* r) { zo_fr_mul(a, &FR_R2, r); }            /* bls12-381.cu:585-587 */
void zo_fr_unmont(const zo_fr* a, zo_fr* r) {                                     /* bls12-381.cu:589-593 */
  zo_fr one = FR_0; one.v[0] = 1; zo_fr_mul(a, &one, r);
}
void zo_fp_add(const zo_fp* a, const zo_fp* b, zo_fp* r) { l_mod_add(r->v, a->v, b->v, FP_P.v, 12); }
void zo_fp_sub(const zo_fp* a, const zo_fp* b, zo_fp* r) { l_mod_sub(r->v, a->v, b->v, FP_P.v, 12); }
void zo_fp_mul(const zo_fp* a, const zo_fp* b, zo_fp* r) { l_mont_mul(r->v, a->v, b->v, FP_P.v, FP_INV, 12); }
void zo_fp_sqr(const zo_fp* a, zo_fp* r) { zo_fp_mul(a, a, r); }
void zo_fp_double(const zo_fp* a, zo_fp* r) { l_mod_double(r->v, a->v, FP_P.v, 12); }
void zo_fp_mont(const zo_fp* a, zo_fp* r) { zo_fp_mul(a, &FP_R2, r); }
void zo_fp_unmont(const zo_fp* a, zo_fp* r) { zo_fp one = FP_0; one.v[0] = 1; zo_fp_mul(a, &one, r); }

void zo_constants(zo_fr* fr_one, zo_fr* fr_p, zo_fr* fr_r2, zo_fp* fp_one, zo_fp* fp_p, zo_fp* fp_r2) {
  *fr_one = FR_ONE; *fr_p = FR_P; *fr_r2 = FR_R2; *fp_one = FP_ONE; *fp_p = FP_P; *fp_r2 = FP_R2;
}
void zo_g1_generator(zo_g1j* g) { g->x = G1_GEN_X; g->y = G1_GEN_Y; g->z = FP_ONE; }

/* value-style helpers keep the curve formulas readable */
static inline zo_fr fr_add(zo_fr a, zo_fr b) { zo_fr r; zo_fr_add(&a, &b, &r); return r; }
static inline zo_fr fr_sub(zo_fr a, zo_fr b) { zo_fr r; zo_fr_sub(&a, &b, &r); return r; }
static inline zo_fr fr_mul(zo_fr a, zo_fr b) { zo_fr r; zo_fr_mul(&a, &b, &r); return r; }
static inline zo_fr fr_dbl(zo_fr a) { zo_fr r; zo_fr_double(&a, &r); return r; }
static inline zo_fr fr_unmont(zo_fr a) { zo_fr r; zo_fr_unmont(&a, &r); return r; }
static inline zo_fr fr_mont(zo_fr a) { zo_fr r; zo_fr_mont(&a, &r); return r; }
static inline zo_fp fp_add(zo_fp a, zo_fp b) { zo_fp r; zo_fp_add(&a, &b, &r); return r; }
static inline zo_fp fp_sub(zo_fp a, zo_fp b) { zo_fp r; zo_fp_sub(&a, &b, &r); return r; }
static inline zo_fp fp_mul(zo_fp a, zo_fp b) { zo_fp r; zo_fp_mul(&a, &b, &r); return r; }
static inline zo_fp fp_sqr(zo_fp a) { zo_fp r; zo_fp_mul(&a, &a, &r); return r; }
static inline zo_fp fp_dbl(zo_fp a) { zo_fp r; zo_fp_double(&a, &r); return r; }
static inline int fp_eq(zo_fp a, zo_fp b) { return l_eq(a.v, b.v, 12); }
static inline int fp_is_zero(zo_fp a) { return l_is_zero(a.v, 12); }

/* ------------------------------------------------------------------ G1 (Jacobian) */
static const zo_g1j G1_ZERO_ = {{{0}}, {{196605u, 1980301312u, 3289120770u, 3958636555u, 1405573306u, 1598593111u, 1884444485u, 2010011731u, 2723605613u, 1543969431u, 4202751123u, 368467651u}}, {{0}}};
/* bls12-381.cuh:417 G1Affine_ZERO = {0, ONE, 0} */

/* dbl-2009-l, bls12-381.cu:1332-1357 */
static zo_g1j g1_double(zo_g1j q) {
  if (fp_is_zero(q.z)) return q;
  zo_fp a = fp_sqr(q.x), b = fp_sqr(q.y), c = fp_sqr(b);
  zo_fp d = fp_add(q.x, b);
  d = fp_sqr(d); d = fp_sub(fp_sub(d, a), c); d = fp_dbl(d);
  zo_fp e = fp_add(fp_dbl(a), a);
  zo_fp f = fp_sqr(e);
  zo_g1j o;
  o.z = fp_dbl(fp_mul(q.y, q.z));
  o.x = fp_sub(fp_sub(f, d), d);
  c = fp_dbl(fp_dbl(fp_dbl(c)));
  o.y = fp_sub(fp_mul(fp_sub(d, o.x), e), c);
  return o;
}
/* add-2007-bl, bls12-381.cu:1400-1435 */
static zo_g1j g1_add(zo_g1j a, zo_g1j b) {
  if (fp_is_zero(a.z)) return b;
  if (fp_is_zero(b.z)) return a;
  zo_fp z1z1 = fp_sqr(a.z), z2z2 = fp_sqr(b.z);
  zo_fp u1 = fp_mul(a.x, z2z2), u2 = fp_mul(b.x, z1z1);
  zo_fp s1 = fp_mul(fp_mul(a.y, b.z), z2z2), s2 = fp_mul(fp_mul(b.y, a.z), z1z1);
  if (fp_eq(u1, u2) && fp_eq(s1, s2)) return g1_double(a);
  zo_fp h = fp_sub(u2, u1);
  zo_fp i = fp_sqr(fp_dbl(h));
  zo_fp j = fp_mul(h, i);
  zo_fp r = fp_dbl(fp_sub(s2, s1));
  zo_fp v = fp_mul(u1, i);
  zo_g1j o;
  o.x = fp_sub(fp_sub(fp_sub(fp_sqr(r), j), v), v);
  o.y = fp_mul(fp_sub(v, o.x), r);
  s1 = fp_dbl(fp_mul(s1, j));
  o.y = fp_sub(o.y, s1);
  o.z = fp_sqr(fp_add(a.z, b.z));
  o.z = fp_sub(fp_sub(o.z, z1z1), z2z2);
  o.z = fp_mul(o.z, h);
  return o;
}
/* madd-2007-bl, bls12-381.cu:1360-1397 */
static zo_g1j g1_add_mixed(zo_g1j a, zo_g1a b) {
  if (fp_is_zero(a.z)) { zo_g1j o; o.x = b.x; o.y = b.y; o.z = FP_ONE; return o; }
  zo_fp z1z1 = fp_sqr(a.z);
  zo_fp u2 = fp_mul(b.x, z1z1);
  zo_fp s2 = fp_mul(fp_mul(b.y, a.z), z1z1);
  if (fp_eq(a.x, u2) && fp_eq(a.y, s2)) return g1_double(a);
  zo_fp h = fp_sub(u2, a.x);
  zo_fp hh = fp_sqr(h);
  zo_fp i = fp_dbl(fp_dbl(hh));
  zo_fp j = fp_mul(h, i);
  zo_fp r = fp_dbl(fp_sub(s2, a.y));
  zo_fp v = fp_mul(a.x, i);
  zo_g1j o;
  o.x = fp_sub(fp_sub(fp_sqr(r), j), fp_dbl(v));
  j = fp_dbl(fp_mul(a.y, j));
  o.y = fp_sub(fp_mul(fp_sub(v, o.x), r), j);
  o.z = fp_add(a.z, h);
  o.z = fp_sub(fp_sub(fp_sqr(o.z), z1z1), hh);
  return o;
}
/* g1-tensor.cu:10-20 */
static zo_g1j g1_neg(zo_g1j a) { a.y = fp_sub(FP_0, a.y); return a; }
static zo_g1a g1a_neg(zo_g1a a) { a.y = fp_sub(FP_0, a.y); return a; }
/* g1-tensor.cu:422-430: LSB-first double-and-add over all 256 raw limb bits */
static zo_g1j g1_mul(zo_g1j a, zo_fr x) {
  zo_g1j out = G1_ZERO_;
  for (unsigned i = 0; i < 256; ++i) {
    if ((x.v[i / 32] >> (i % 32)) & 1u) out = g1_add(out, a);
    a = g1_double(a);
  }
  return out;
}
void zo_g1_double(const zo_g1j* a, zo_g1j* r) { *r = g1_double(*a); }
void zo_g1_add(const zo_g1j* a, const zo_g1j* b, zo_g1j* r) { *r = g1_add(*a, *b); }
void zo_g1_add_mixed(const zo_g1j* a, const zo_g1a* b, zo_g1j* r) { *r = g1_add_mixed(*a, *b); }
void zo_g1_neg(const zo_g1j* a, zo_g1j* r) { *r = g1_neg(*a); }
void zo_g1_mul(const zo_g1j* a, const zo_fr* x, zo_g1j* r) { *r = g1_mul(*a, *x); }

/* affine-canonical form; Fp inverse by Fermat (a^(p-2)) in the Montgomery domain */
static zo_fp fp_inv(zo_fp a) {
  uint32_t e[12], two[12] = {2};
  l_sub_(e, FP_P.v, two, 12);
  zo_fp acc = FP_ONE;
  for (int i = 12 * 32 - 1; i >= 0; i--) {
    acc = fp_sqr(acc);
    if ((e[i / 32] >> (i % 32)) & 1u) acc = fp_mul(acc, a);
  }
  return acc;
}
void zo_g1_to_affine(const zo_g1j* a, size_t n, zo_g1a* out, uint8_t* inf) {
#pragma omp parallel for schedule(dynamic, 8)
  for (size_t i = 0; i < n; i++) {
    if (fp_is_zero(a[i].z)) { inf[i] = 1; out[i].x = FP_0; out[i].y = FP_0; continue; }
    inf[i] = 0;
    zo_fp zi = fp_inv(a[i].z), zi2 = fp_sqr(zi);
    out[i].x = fp_mul(a[i].x, zi2);
    out[i].y = fp_mul(a[i].y, fp_mul(zi2, zi));
  }
}

/* ------------------------------------------------------------------ Fr tensor ops */
/* fr-tensor.cu:20-87 (kernel bodies) + :122-238 (operators) */
#define VLOOP(expr) do { _Pragma("omp parallel for if (n > 4096)") for (size_t i = 0; i < n; i++) { expr; } } while (0)
void zo_fr_vadd(const zo_fr* a, const zo_fr* b, zo_fr* out, size_t n) { VLOOP(out[i] = fr_add(a[i], b[i])); }
void zo_fr_vsub(const zo_fr* a, const zo_fr* b, zo_fr* out, size_t n) { VLOOP(out[i] = fr_sub(a[i], b[i])); }
void zo_fr_vmul(const zo_fr* a, const zo_fr* b, zo_fr* out, size_t n) { VLOOP(out[i] = fr_mul(a[i], b[i])); }
void zo_fr_vneg(const zo_fr* a, zo_fr* out, size_t n) { VLOOP(out[i] = fr_sub(FR_0, a[i])); }
void zo_fr_vadd_s(const zo_fr* a, const zo_fr* x, zo_fr* out, size_t n) { VLOOP(out[i] = fr_add(a[i], *x)); }
void zo_fr_vsub_s(const zo_fr* a, const zo_fr* x, zo_fr* out, size_t n) { VLOOP(out[i] = fr_sub(a[i], *x)); }
void zo_fr_vmul_s(const zo_fr* a, const zo_fr* x, zo_fr* out, size_t n) { VLOOP(out[i] = fr_mul(a[i], *x)); }
void zo_fr_vmont(const zo_fr* a, zo_fr* out, size_t n) { VLOOP(out[i] = fr_mont(a[i])); }
void zo_fr_vunmont(const zo_fr* a, zo_fr* out, size_t n) { VLOOP(out[i] = fr_unmont(a[i])); }

/* fr-tensor.cu:240-292.  Fr results are canonical, so the tree shape does not matter. */
void zo_fr_sum(const zo_fr* a, size_t n, zo_fr* out) {
  zo_fr acc = FR_0;
  for (size_t i = 0; i < n; i++) acc = fr_add(acc, a[i]);
  *out = acc;
}

/* fr-tensor.cu:399-409 */
static void fr_me_step(const zo_fr* in, zo_fr* out, zo_fr x, size_t in_size, size_t out_size) {
#pragma omp parallel for if (out_size > 4096)
  for (size_t g = 0; g < out_size; g++) {
    size_t g0 = 2 * g, g1 = 2 * g + 1;
    if (g1 < in_size) out[g] = fr_add(in[g0], fr_mul(x, fr_sub(in[g1], in[g0])));
    else if (g0 < in_size) out[g] = fr_sub(in[g0], fr_mul(x, in[g0]));
    else out[g] = FR_0;
  }
}
/* fr-tensor.cu:295-300 (guard) + :411-418 (Fr_me) */
int zo_fr_me(const zo_fr* t, size_t n, const zo_fr* u, size_t k, zo_fr* out) {
  if (n == 0) return -1;
  if (n <= (((size_t)1 << k) / 2) || n > ((size_t)1 << k)) return -1;
  zo_fr* cur = (zo_fr*)malloc(n * sizeof(zo_fr));
  zo_fr* nxt = (zo_fr*)malloc(((n + 1) / 2) * sizeof(zo_fr));
  memcpy(cur, t, n * sizeof(zo_fr));
  size_t sz = n;
  for (size_t r = 0; r < k; r++) {
    size_t osz = (sz + 1) / 2;
    fr_me_step(cur, nxt, u[r], sz, osz);
    zo_fr* tmp = cur; cur = nxt; nxt = tmp; sz = osz;
  }
  *out = cur[0];
  free(cur); free(nxt);
  return 0;
}
/* fr-tensor.cu:420-432 */
static void fr_partial_me_step(const zo_fr* in, zo_fr* out, zo_fr x, size_t in_size, size_t out_size, size_t w) {
#pragma omp parallel for if (out_size > 4096)
  for (size_t g = 0; g < out_size; g++) {
    size_t wid = g / w, idx = g % w;
    size_t g0 = 2 * wid * w + idx, g1 = (2 * wid + 1) * w + idx;
    if (g1 < in_size) out[g] = fr_add(in[g0], fr_mul(x, fr_sub(in[g1], in[g0])));
    else if (g0 < in_size) out[g] = fr_sub(in[g0], fr_mul(x, in[g0]));
    else out[g] = FR_0;
  }
}
size_t zo_fr_partial_me_size(size_t n, size_t k, size_t window) {
  size_t sz = n;
  for (size_t r = 0; r < k; r++) sz = window * ((sz + 2 * window - 1) / (2 * window));
  return sz;
}
/* fr-tensor.cu:370-374 (guard; k==0 is UB there -- defined here as identity)
 * + :434-443 (Fr_partial_me) */
int zo_fr_partial_me(const zo_fr* t, size_t n, const zo_fr* u, size_t k, size_t window, zo_fr* out) {
  if (window == 0) return -1;
  if (k == 0) { memcpy(out, t, n * sizeof(zo_fr)); return 0; }
  if (n <= window * ((size_t)1 << (k - 1))) return -1;
  zo_fr* cur = (zo_fr*)malloc(n * sizeof(zo_fr));
  memcpy(cur, t, n * sizeof(zo_fr));
  size_t sz = n;
  for (size_t r = 0; r < k; r++) {
    size_t osz = window * ((sz + 2 * window - 1) / (2 * window));
    zo_fr* nxt = (zo_fr*)malloc(osz * sizeof(zo_fr));
    fr_partial_me_step(cur, nxt, u[r], sz, osz, window);
    free(cur); cur = nxt; sz = osz;
  }
  memcpy(out, cur, sz * sizeof(zo_fr));
  free(cur);
  return 0;
}
/* fr-tensor.cu:376-397 */
int zo_fr_split(const zo_fr* t, size_t n, size_t window, zo_fr* out0, zo_fr* out1) {
  if (window < 1 || window >= n) return -1;
  size_t out_size = (n + 1) / 2;
  for (size_t g = 0; g < out_size; g++) {
    size_t wid = g / window, idx = g % window;
    size_t g0 = 2 * wid * window + idx, g1 = (2 * wid + 1) * window + idx;
    out0[g] = (g0 < n) ? t[g0] : FR_0;
    out1[g] = (g1 < n) ? t[g1] : FR_0;
  }
  return 0;
}

/* ------------------------------------------------------------------ sumchecks */
/* proof.cu:13-30 */
uint32_t zo_ceil_log2(uint32_t num) {
  if (num == 0) return 0;
  num--;
  uint32_t result = 0;
  while (num > 0) { num >>= 1; result++; }
  return result;
}
/* proof.cu:55-70 */
static void fr_ip_sc_step(const zo_fr* a, const zo_fr* b, zo_fr* o0, zo_fr* o1, zo_fr* o2, size_t in_size, size_t out_size) {
#pragma omp parallel for if (out_size > 4096)
  for (size_t g = 0; g < out_size; g++) {
    size_t g0 = 2 * g, g1 = 2 * g + 1;
    zo_fr a0 = (g0 < in_size) ? a[g0] : FR_0, b0 = (g0 < in_size) ? b[g0] : FR_0;
    zo_fr a1 = (g1 < in_size) ? a[g1] : FR_0, b1 = (g1 < in_size) ? b[g1] : FR_0;
    o0[g] = fr_mul(a0, b0);
    o1[g] = fr_add(fr_mul(a0, fr_sub(b1, b0)), fr_mul(b0, fr_sub(a1, a0)));
    o2[g] = fr_mul(fr_sub(a1, a0), fr_sub(b1, b0));
  }
}
/* proof.cu:72-108 */
int zo_inner_product_sumcheck(const zo_fr* a_in, const zo_fr* b_in, size_t n, const zo_fr* u, size_t k, zo_fr* proof) {
  if (n == 0 || n <= (((size_t)1 << k) / 2) || n > ((size_t)1 << k)) return -1;
  zo_fr *a = malloc(n * sizeof(zo_fr)), *b = malloc(n * sizeof(zo_fr));
  size_t half = (n + 1) / 2;
  zo_fr *o0 = malloc(half * sizeof(zo_fr)), *o1 = malloc(half * sizeof(zo_fr)), *o2 = malloc(half * sizeof(zo_fr));
  zo_fr *an = malloc(half * sizeof(zo_fr)), *bn = malloc(half * sizeof(zo_fr));
  memcpy(a, a_in, n * sizeof(zo_fr)); memcpy(b, b_in, n * sizeof(zo_fr));
  size_t sz = n, p = 0;
  for (size_t r = 0; r < k; r++) {
    size_t osz = (sz + 1) / 2;
    fr_ip_sc_step(a, b, o0, o1, o2, sz, osz);
    zo_fr_sum(o0, osz, &proof[p++]); zo_fr_sum(o1, osz, &proof[p++]); zo_fr_sum(o2, osz, &proof[p++]);
    fr_me_step(a, an, u[r], sz, osz);
    fr_me_step(b, bn, u[r], sz, osz);
    memcpy(a, an, osz * sizeof(zo_fr)); memcpy(b, bn, osz * sizeof(zo_fr));
    sz = osz;
  }
  proof[p++] = a[0]; proof[p++] = b[0];
  free(a); free(b); free(o0); free(o1); free(o2); free(an); free(bn);
  return 0;
}
/* proof.cu:110-150 */
int zo_hadamard_product_sumcheck(const zo_fr* a_in, const zo_fr* b_in, size_t n, const zo_fr* u, const zo_fr* v, size_t k, zo_fr* proof) {
  if (n == 0 || n > ((size_t)1 << k)) return -1;
  if (k > 0 && n <= ((size_t)1 << (k - 1))) return -1;
  zo_fr *a = malloc(n * sizeof(zo_fr)), *b = malloc(n * sizeof(zo_fr));
  size_t half = (n + 1) / 2;
  zo_fr *o0 = malloc(half * sizeof(zo_fr)), *o1 = malloc(half * sizeof(zo_fr)), *o2 = malloc(half * sizeof(zo_fr));
  zo_fr *an = malloc(half * sizeof(zo_fr)), *bn = malloc(half * sizeof(zo_fr));
  memcpy(a, a_in, n * sizeof(zo_fr)); memcpy(b, b_in, n * sizeof(zo_fr));
  size_t sz = n, p = 0;
  int rc = 0;
  for (size_t r = 0; r < k && rc == 0; r++) {
    size_t osz = (sz + 1) / 2;
    fr_ip_sc_step(a, b, o0, o1, o2, sz, osz);
    rc |= zo_fr_me(o0, osz, u + r + 1, k - r - 1, &proof[p++]);
    rc |= zo_fr_me(o1, osz, u + r + 1, k - r - 1, &proof[p++]);
    rc |= zo_fr_me(o2, osz, u + r + 1, k - r - 1, &proof[p++]);
    fr_me_step(a, an, v[r], sz, osz);
    fr_me_step(b, bn, v[r], sz, osz);
    memcpy(a, an, osz * sizeof(zo_fr)); memcpy(b, bn, osz * sizeof(zo_fr));
    sz = osz;
  }
  proof[p++] = a[0]; proof[p++] = b[0];
  free(a); free(b); free(o0); free(o1); free(o2); free(an); free(bn);
  return rc;
}
/* proof.cu:152-163 */
static void fr_bin_sc_step(const zo_fr* a, zo_fr* o0, zo_fr* o1, zo_fr* o2, size_t in_size, size_t out_size) {
#pragma omp parallel for if (out_size > 4096)
  for (size_t g = 0; g < out_size; g++) {
    zo_fr a0 = (2 * g < in_size) ? a[2 * g] : FR_0;
    zo_fr a1 = (2 * g + 1 < in_size) ? a[2 * g + 1] : FR_0;
    o0[g] = fr_sub(fr_mul(a0, a0), a0);
    zo_fr d = fr_sub(a1, a0);
    o1[g] = fr_sub(fr_mul(fr_dbl(a0), d), d);
    o2[g] = fr_mul(d, d);
  }
}
/* proof.cu:165-200 */
int zo_binary_sumcheck(const zo_fr* a_in, size_t n, const zo_fr* u, const zo_fr* v, size_t k, zo_fr* proof) {
  if (n == 0 || n <= (((size_t)1 << k) / 2) || n > ((size_t)1 << k)) return -1;
  zo_fr* a = malloc(n * sizeof(zo_fr));
  size_t half = (n + 1) / 2;
  zo_fr *o0 = malloc(half * sizeof(zo_fr)), *o1 = malloc(half * sizeof(zo_fr)), *o2 = malloc(half * sizeof(zo_fr));
  zo_fr* an = malloc(half * sizeof(zo_fr));
  memcpy(a, a_in, n * sizeof(zo_fr));
  size_t sz = n, p = 0;
  int rc = 0;
  for (size_t r = 0; r < k && rc == 0; r++) {
    size_t osz = (sz + 1) / 2;
    fr_bin_sc_step(a, o0, o1, o2, sz, osz);
    rc |= zo_fr_me(o0, osz, u + r + 1, k - r - 1, &proof[p++]);
    rc |= zo_fr_me(o1, osz, u + r + 1, k - r - 1, &proof[p++]);
    rc |= zo_fr_me(o2, osz, u + r + 1, k - r - 1, &proof[p++]);
    fr_me_step(a, an, v[r], sz, osz);
    memcpy(a, an, osz * sizeof(zo_fr));
    sz = osz;
  }
  proof[p++] = a[0];
  free(a); free(o0); free(o1); free(o2); free(an);
  return rc;
}

/* ------------------------------------------------------------------ G1 tensor ops */
/* g1-tensor.cu:169-366 (kernel bodies; sub = add of the negated operand) */
#define GLOOP(expr) do { _Pragma("omp parallel for schedule(dynamic, 16) if (n > 64)") for (size_t i = 0; i < n; i++) { expr; } } while (0)
void zo_g1_vadd(const zo_g1j* a, const zo_g1j* b, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add(a[i], b[i])); }
void zo_g1_vsub(const zo_g1j* a, const zo_g1j* b, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add(a[i], g1_neg(b[i]))); }
void zo_g1_vneg(const zo_g1j* a, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_neg(a[i])); }
void zo_g1_vadd_s(const zo_g1j* a, const zo_g1j* x, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add(a[i], *x)); }
void zo_g1_vsub_s(const zo_g1j* a, const zo_g1j* x, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add(a[i], g1_neg(*x))); }
void zo_g1_vmadd(const zo_g1j* a, const zo_g1a* b, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add_mixed(a[i], b[i])); }
void zo_g1_vmsub(const zo_g1j* a, const zo_g1a* b, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add_mixed(a[i], g1a_neg(b[i]))); }
void zo_g1_vmadd_s(const zo_g1j* a, const zo_g1a* x, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add_mixed(a[i], *x)); }
void zo_g1_vmsub_s(const zo_g1j* a, const zo_g1a* x, zo_g1j* out, size_t n) { GLOOP(out[i] = g1_add_mixed(a[i], g1a_neg(*x))); }
/* g1-tensor.cu:142-147 */
void zo_g1_from_affine(const zo_g1a* a, zo_g1j* out, size_t n) {
  for (size_t i = 0; i < n; i++) { out[i].x = a[i].x; out[i].y = a[i].y; out[i].z = FP_ONE; }
}
/* g1-tensor.cu:368-420 with G1NumThread = 64 (g1-tensor.cuh:11): blocks of 64 threads over 128
 * inputs, grid = ceil(cur/64) (over-provisioned: trailing blocks emit ZERO), 32..1 ladder. */
void zo_g1_sum(const zo_g1j* a, size_t n, zo_g1j* out) {
  if (n == 0) { *out = G1_ZERO_; return; }
  zo_g1j* in = malloc(n * sizeof(zo_g1j));
  zo_g1j* o = malloc(((n + 1) / 2 + 1) * sizeof(zo_g1j));
  memcpy(in, a, n * sizeof(zo_g1j));
  size_t cur = n;
  while (cur > 1) {
    size_t grid = (cur + 63) / 64;
#pragma omp parallel for schedule(dynamic, 1) if (grid > 1)
    for (size_t b = 0; b < grid; b++) {
      zo_g1j s[64];
      for (size_t t = 0; t < 64; t++) {
        size_t i = b * 128 + t;
        s[t] = (i < cur) ? in[i] : G1_ZERO_;
        if (i + 64 < cur) s[t] = g1_add(s[t], in[i + 64]);
      }
      for (size_t st = 32; st > 0; st >>= 1)
        for (size_t t = 0; t < st; t++) s[t] = g1_add(s[t], s[t + st]);
      o[b] = s[0];
    }
    zo_g1j* tmp = in; in = o; o = tmp;
    cur = grid;
  }
  *out = in[0];
  free(in); free(o);
}
/* g1-tensor.cu:433-445 */
void zo_g1_vmul(const zo_g1j* g, const zo_fr* s, zo_g1j* out, size_t n) {
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t i = 0; i < n; i++) out[i] = g1_mul(g[i], s[i]);
}
void zo_g1_vmul_bcast(const zo_g1j* g, size_t n, const zo_fr* s, size_t total, zo_g1j* out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t i = 0; i < total; i++) out[i] = g1_mul(g[i % n], s[i]);
}
/* g1-tensor.cu:463-475 */
static void g1_me_step(const zo_g1j* in, zo_g1j* out, zo_fr x, size_t in_size, size_t out_size) {
  zo_fr xu = fr_unmont(x);
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t g = 0; g < out_size; g++) {
    size_t g0 = 2 * g, g1 = 2 * g + 1;
    if (g1 < in_size) out[g] = g1_add(in[g0], g1_mul(g1_add(in[g1], g1_neg(in[g0])), xu));
    else if (g0 < in_size) out[g] = g1_add(in[g0], g1_neg(g1_mul(in[g0], xu)));
    else out[g] = G1_ZERO_;
  }
}
/* g1-tensor.cu:477-491 */
int zo_g1_me(const zo_g1j* t, size_t n, const zo_fr* u, size_t k, zo_g1j* out) {
  if (n == 0 || n > ((size_t)1 << k)) return -1;
  if (k > 0 && n <= ((size_t)1 << (k - 1))) return -1;
  zo_g1j* cur = malloc(n * sizeof(zo_g1j));
  zo_g1j* nxt = malloc(((n + 1) / 2) * sizeof(zo_g1j));
  memcpy(cur, t, n * sizeof(zo_g1j));
  size_t sz = n;
  for (size_t r = 0; r < k; r++) {
    size_t osz = (sz + 1) / 2;
    g1_me_step(cur, nxt, u[r], sz, osz);
    zo_g1j* tmp = cur; cur = nxt; nxt = tmp; sz = osz;
  }
  *out = cur[0];
  free(cur); free(nxt);
  return 0;
}

/* ------------------------------------------------------------------ commitment */
/* commitment.cu:29-41 with the INTENDED launch of sum_axis_n_optimized (one 64-thread block per
 * row, commitment.cu:3-27): thread-strided accumulate, then the 32..1 ladder. */
int zo_commit(const zo_g1j* gens, size_t n, const zo_fr* t, size_t tn, zo_g1j* out) {
  if (n == 0 || tn % n != 0) return -1;
  size_t m = tn / n;
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < m; row++) {
    zo_g1j s[64];
    for (size_t l = 0; l < 64; l++) {
      if (l >= n) { s[l] = G1_ZERO_; continue; }
      zo_g1j acc = g1_mul(gens[l], fr_unmont(t[row * n + l]));
      for (size_t i = l + 64; i < n; i += 64) acc = g1_add(acc, g1_mul(gens[i], fr_unmont(t[row * n + i])));
      s[l] = acc;
    }
    for (size_t st = 32; st > 0; st >>= 1)
      for (size_t l = 0; l < st; l++) s[l] = g1_add(s[l], s[l + st]);
    out[row] = s[0];
  }
  return 0;
}
/* commitment.cu:3-27 exactly as launched at :38 (grid ceil(m/64) x 64 threads, gid = one row per
 * THREAD): thread (b, l) reads row gid=64b+l at columns l, l+64, ...; the block ladder then mixes
 * the 64 rows of block b; only local_id 0 stores, to arr_out[gid] = out[64b].  Columns beyond the
 * row are read from the NEXT row's memory (arr[gid*n + local_id] with local_id >= n never occurs
 * for n >= 64; for n < 64 the read runs into the following rows -- reproduced literally, guarded
 * at the end of the buffer where the reference reads out of bounds). */
int zo_commit_as_launched(const zo_g1j* gens, size_t n, const zo_fr* t, size_t tn, zo_g1j* out, uint8_t* written) {
  if (n == 0 || tn % n != 0) return -1;
  size_t m = tn / n;
  memset(written, 0, m);
  size_t grid = (m + 63) / 64;
  for (size_t b = 0; b < grid; b++) {
    zo_g1j s[64];
    int oob = 0;
    for (size_t l = 0; l < 64; l++) {
      size_t gid = b * 64 + l;
      s[l] = G1_ZERO_;
      if (gid >= m) { oob = 1; continue; } /* thread returned before the barrier: UB in the reference */
      size_t idx = gid * n + l;
      if (idx >= tn) { oob = 1; continue; }
      zo_g1j acc = g1_mul(gens[idx % n], fr_unmont(t[idx]));
      for (size_t i = l + 64; i < n; i += 64) acc = g1_add(acc, g1_mul(gens[(gid * n + i) % n], fr_unmont(t[gid * n + i])));
      s[l] = acc;
    }
    (void)oob;
    for (size_t st = 32; st > 0; st >>= 1)
      for (size_t l = 0; l < st; l++) s[l] = g1_add(s[l], s[l + st]);
    out[b * 64] = s[0];
    written[b * 64] = 1;
  }
  return 0;
}
/* commitment.cu:43-60 (me_open_step) + :62-81 (me_open) */
int zo_me_open(const zo_fr* t_in, const zo_g1j* gens_in, size_t n, const zo_fr* u, size_t k, zo_g1j* proof, zo_fr* ret) {
  if (n == 0) return -1;
  zo_fr* t = malloc(n * sizeof(zo_fr));
  zo_g1j* g = malloc(n * sizeof(zo_g1j));
  memcpy(t, t_in, n * sizeof(zo_fr)); memcpy(g, gens_in, n * sizeof(zo_g1j));
  size_t sz = n, p = 0;
  for (size_t r = 0; r < k; r++) {
    size_t ns = sz / 2;
    if (ns == 0) { free(t); free(g); return -1; }
    zo_fr* nt = malloc(ns * sizeof(zo_fr));
    zo_g1j *ng = malloc(ns * sizeof(zo_g1j)), *tmp = malloc(ns * sizeof(zo_g1j)), *tmp0 = malloc(ns * sizeof(zo_g1j)), *tmp1 = malloc(ns * sizeof(zo_g1j));
    zo_fr uu = fr_unmont(u[r]);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < ns; i++) {
      size_t g0 = 2 * i, g1 = 2 * i + 1;
      nt[i] = fr_add(t[g0], fr_mul(u[r], fr_sub(t[g1], t[g0])));
      ng[i] = g1_add(g[g1], g1_mul(g1_add(g[g0], g1_neg(g[g1])), uu));
      tmp[i] = g1_add(g1_mul(g[g0], t[g0]), g1_mul(g[g1], t[g1]));
      tmp0[i] = g1_mul(g[g1], t[g0]);
      tmp1[i] = g1_mul(g[g0], t[g1]);
    }
    zo_g1_sum(tmp, ns, &proof[p++]);
    zo_g1_sum(tmp0, ns, &proof[p++]);
    zo_g1_sum(tmp1, ns, &proof[p++]);
    free(t); free(g); free(tmp); free(tmp0); free(tmp1);
    t = nt; g = ng; sz = ns;
  }
  proof[p++] = g[0];
  *ret = t[0];
  free(t); free(g);
  return 0;
}
/* commitment.cu:83-92 */
int zo_open(const zo_g1j* gens, size_t n, const zo_fr* t, size_t tn, const zo_g1j* com, size_t m,
            const zo_fr* u, size_t k, zo_g1j* com_eval, zo_g1j* proof, zo_fr* ret) {
  size_t k_hi = zo_ceil_log2((uint32_t)m);
  if (k_hi > k) return -1;
  size_t k_lo = k - k_hi;
  const zo_fr* u_lo = u;
  const zo_fr* u_hi = u + k_lo;
  if (zo_g1_me(com, m, u_hi, k_hi, com_eval) != 0) return -1;
  if (n != ((size_t)1 << k_lo)) return -1;
  size_t fsz = zo_fr_partial_me_size(tn, k_hi, n);
  zo_fr* folded = malloc((fsz ? fsz : 1) * sizeof(zo_fr));
  int rc = zo_fr_partial_me(t, tn, u_hi, k_hi, n, folded);
  if (rc == 0 && fsz != n) rc = -1;
  if (rc == 0) rc = zo_me_open(folded, gens, n, u_lo, k_lo, proof, ret);
  free(folded);
  return rc;
}

/* ------------------------------------------------------------------ layers */
/* zkfc.cu:63-78 (float_to_Fr) + :80-88 (kernel).  CUDA round() = half away from zero = roundf. */
static zo_fr float_to_fr(float x) {
  x = x * (float)(1 << 16);
  float abs_x = roundf(fabsf(x));
  int negative = copysignf(1.0f, x) < 0;
  uint32_t ra = (abs_x >= 4294967296.0f) ? 0xFFFFFFFFu : (uint32_t)abs_x; /* UB in the reference; clamp (App. D-8) */
  zo_fr v = FR_0; v.v[0] = ra;
  return negative ? fr_sub(FR_0, v) : v;
}
void zo_float_to_fr(const float* fs, zo_fr* frs, uint32_t fs_rows, uint32_t frs_rows, uint32_t fs_cols, uint32_t frs_cols) {
  size_t total = (size_t)frs_rows * frs_cols;
#pragma omp parallel for if (total > 4096)
  for (size_t tid = 0; tid < total; tid++) {
    size_t d0 = tid / frs_cols, d1 = tid % frs_cols;
    frs[tid] = (d0 < fs_rows && d1 < fs_cols) ? float_to_fr(fs[d0 * fs_cols + d1]) : FR_0;
  }
}
/* zkfc.cu:6-47,117-126: Z[B x out] = X[B x in] * W[in x out], Montgomery mul + modular add */
void zo_fc_forward(const zo_fr* X, const zo_fr* W, uint32_t batch, uint32_t in, uint32_t out, zo_fr* Z) {
#pragma omp parallel for collapse(2) schedule(static)
  for (uint32_t b = 0; b < batch; b++)
    for (uint32_t o = 0; o < out; o++) {
      zo_fr s = FR_0;
      for (uint32_t kk = 0; kk < in; kk++) s = fr_add(s, fr_mul(X[(size_t)b * in + kk], W[(size_t)kk * out + o]));
      Z[(size_t)b * out + o] = s;
    }
}
/* zkrelu.cu:3-41.  Out-of-window inputs are UB in the reference (mag uninitialised, sign not
 * written); here they yield mag = 0, sign = 0 so runs stay deterministic. */
void zo_relu_forward(const zo_fr* X, size_t n, zo_fr* Z, zo_fr* sign, zo_fr* mag_bin, zo_fr* rem_bin) {
  static const zo_fr POS_MAX = {{4294967295u, 32767u, 0, 0, 0, 0, 0, 0}};
  static const zo_fr NEG_MIN = {{1u, 4294934527u, 4294859774u, 1404937218u, 161601541u, 859428872u, 698187080u, 1944954707u}};
  static const zo_fr TWO47 = {{0u, 32768u, 0, 0, 0, 0, 0, 0}};
#pragma omp parallel for if (n > 1024)
  for (size_t g = 0; g < n; g++) {
    zo_fr x = fr_unmont(X[g]);
    uint64_t mag = 0;
    if (l_gte(POS_MAX.v, x.v, 8)) {
      sign[g] = FR_ONE;
      mag = (uint64_t)x.v[0] | ((uint64_t)x.v[1] << 32);
    } else if (l_gte(x.v, NEG_MIN.v, 8)) {
      sign[g] = FR_0;
      zo_fr y = fr_add(x, TWO47);
      mag = (uint64_t)y.v[0] | ((uint64_t)y.v[1] << 32);
    } else {
      sign[g] = FR_0;
    }
    int rem_sign = (mag & 32768ull) != 0;
    uint32_t rem_mag = (uint32_t)(mag & 32767ull);
    int32_t rem = rem_sign ? ((int32_t)rem_mag - (1 << 15)) : (int32_t)rem_mag;
    uint32_t mag_rescaled = (uint32_t)((mag - (uint64_t)(int64_t)rem) >> 16);
    for (unsigned i = 0; i < 32; i++) mag_bin[g * 32 + i] = ((mag_rescaled >> i) & 1u) ? FR_ONE : FR_0;
    for (unsigned i = 0; i < 15; i++) rem_bin[g * 16 + i] = ((rem_mag >> i) & 1u) ? FR_ONE : FR_0;
    rem_bin[g * 16 + 15] = rem_sign ? FR_ONE : FR_0;
    zo_fr mr = FR_0; mr.v[0] = mag_rescaled;
    Z[g] = fr_mul(fr_mont(mr), sign[g]);
  }
}

/* zkfc.cu:128-145 */
int zo_fc_prove(const zo_fr* X, size_t xn, const zo_fr* Z, size_t zn, const zo_fr* W, uint32_t in, uint32_t out,
                const zo_g1j* gens, size_t n_gens, const zo_g1j* com, size_t m,
                const zo_fr* u_bs, size_t k_bs, const zo_fr* u_in, size_t k_in, const zo_fr* u_out, size_t k_out,
                zo_fr* fr_out, size_t* n_fr, zo_g1j* g1_out, size_t* n_g1) {
  if (in == 0 || xn % in != 0) return -1;
  size_t batch = xn / in;
  if (k_bs != zo_ceil_log2((uint32_t)batch) || k_in != zo_ceil_log2(in) || k_out != zo_ceil_log2(out)) return -2;
  size_t wn = (size_t)in * out;
  int rc = 0;
  size_t asz = zo_fr_partial_me_size(xn, k_bs, in), bsz = zo_fr_partial_me_size(wn, k_out, 1);
  zo_fr *a = malloc(asz * sizeof(zo_fr)), *b = malloc(bsz * sizeof(zo_fr));
  rc |= zo_fr_partial_me(X, xn, u_bs, k_bs, in, a);
  rc |= zo_fr_partial_me(W, wn, u_out, k_out, 1, b);
  size_t pf = 0, pg = 0;
  if (rc == 0 && asz != bsz) rc = -1;
  if (rc == 0) rc = zo_inner_product_sumcheck(a, b, asz, u_in, k_in, fr_out + pf);
  pf += 3 * k_in + 2;
  free(a); free(b);
  if (rc) return rc;
  /* Z(u_Z), u_Z = u_out || u_bs */
  zo_fr* uz = malloc((k_out + k_bs + 1) * sizeof(zo_fr));
  memcpy(uz, u_out, k_out * sizeof(zo_fr)); memcpy(uz + k_out, u_bs, k_bs * sizeof(zo_fr));
  rc = zo_fr_me(Z, zn, uz, k_out + k_bs, &fr_out[pf++]);
  free(uz);
  if (rc) return rc;
  /* generators.open(weights, com, u_out || u_in) */
  zo_fr* uo = malloc((k_out + k_in + 1) * sizeof(zo_fr));
  memcpy(uo, u_out, k_out * sizeof(zo_fr)); memcpy(uo + k_out, u_in, k_in * sizeof(zo_fr));
  size_t k_lo = zo_ceil_log2((uint32_t)n_gens);
  rc = zo_open(gens, n_gens, W, wn, com, m, uo, k_out + k_in, &g1_out[0], &g1_out[1], &fr_out[pf++]);
  pg = 1 + 3 * k_lo + 1;
  free(uo);
  *n_fr = pf; *n_g1 = pg;
  return rc;
}
/* zkrelu.cu:74-100 (Q = 32, R = 16) */
int zo_relu_prove(const zo_fr* X, const zo_fr* Z, size_t n, const zo_fr* sign, const zo_fr* mag_bin, const zo_fr* rem_bin,
                  const zo_fr* u_z, const zo_fr* v_z, const zo_fr* u_r, const zo_fr* v_r, const zo_fr* u_rec,
                  const zo_fr* u_hp, const zo_fr* v_hp, zo_fr* fr_out, size_t* n_fr) {
  (void)Z;
  size_t k = zo_ceil_log2((uint32_t)n), p = 0;
  int rc = zo_binary_sumcheck(mag_bin, n * 32, u_z, v_z, k + 5, fr_out + p); p += 3 * (k + 5) + 1;
  if (rc) return rc;
  if (zo_fr_partial_me_size(n * 32, k, 32) != 32) return -1;
  rc = zo_fr_partial_me(mag_bin, n * 32, u_rec, k, 32, fr_out + p); p += 32;
  if (rc) return rc;
  rc = zo_binary_sumcheck(rem_bin, n * 16, u_r, v_r, k + 4, fr_out + p); p += 3 * (k + 4) + 1;
  if (rc) return rc;
  rc = zo_fr_partial_me(rem_bin, n * 16, u_rec, k, 16, fr_out + p); p += 16;
  if (rc) return rc;
  rc = zo_hadamard_product_sumcheck(X, sign, n, u_hp, v_hp, k, fr_out + p); p += 3 * k + 2;
  *n_fr = p;
  return rc;
}

/* ---- LeakyReLU with negative slope 2^-shift (shift 0: identity, i.e. signed requantisation) -- an EXTENSION: the reference's only
 * activation is zkReLU (zkrelu.cu:11-41).  Same decomposition as zo_relu_forward (sign, 32 magnitude bits of the rescaled value,
 * 16 remainder bits).  For a negative input the reference's kernel decomposes x + 2^47 (zkrelu.cu:19-22), so the "magnitude" it
 * stores is the OFFSET encoding mag_rescaled = 2^31 + q of the requantised value q = round(x / 2^16) < 0, and
 *     floor(q / 2^shift) = (mag_rescaled >> shift) - 2^(31 - shift)
 * exactly (arithmetic shift, rounding toward -infinity): the magnitude bits shift .. 31 minus a constant, so no further bit tensor
 * is needed.  low[g] = mont(mag_rescaled >> shift), nsign[g] = 1 - sign[g], and the negative branch outputs low - 2^(31 - shift). */
void zo_leaky_forward(const zo_fr* X, size_t n, uint32_t shift, zo_fr* Z, zo_fr* sign, zo_fr* mag_bin, zo_fr* rem_bin, zo_fr* low, zo_fr* nsign) {
  zo_relu_forward(X, n, Z, sign, mag_bin, rem_bin);
#pragma omp parallel for if (n > 1024)
  for (size_t g = 0; g < n; g++) {
    uint32_t m = 0;
    for (unsigned i = 0; i < 32; i++)
      if (mag_bin[g * 32 + i].v[0] | mag_bin[g * 32 + i].v[7]) m |= 1u << i;   /* FR_ONE is non-zero in limbs 0 and 7 */
    zo_fr lo = FR_0; lo.v[0] = shift < 32 ? (m >> shift) : 0u;
    low[g] = fr_mont(lo);
    int positive = (sign[g].v[0] | sign[g].v[7]) != 0;
    nsign[g] = positive ? FR_0 : FR_ONE;
    if (!positive) {
      zo_fr c = FR_0; c.v[0] = shift < 32 ? (1u << (31 - shift)) : 0u;
      Z[g] = fr_sub(low[g], fr_mont(c));
    }
  }
}
/* zkReLU::prove's transcript (zkrelu.cu:74-100) followed by hadamard_product_sumcheck(low, nsign, u_lk, v_lk) (proof.cu:110-150):
 * the negative branch's magnitude (low o nsign)~(u_lk), in the style of the reference's (Z o sign) sumcheck */
int zo_leaky_prove(const zo_fr* X, const zo_fr* Z, size_t n, const zo_fr* sign, const zo_fr* mag_bin, const zo_fr* rem_bin, const zo_fr* low, const zo_fr* nsign,
                   const zo_fr* u_z, const zo_fr* v_z, const zo_fr* u_r, const zo_fr* v_r, const zo_fr* u_rec,
                   const zo_fr* u_hp, const zo_fr* v_hp, const zo_fr* u_lk, const zo_fr* v_lk, zo_fr* fr_out, size_t* n_fr) {
  size_t p = 0;
  int rc = zo_relu_prove(X, Z, n, sign, mag_bin, rem_bin, u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp, fr_out, &p);
  if (rc) return rc;
  size_t k = zo_ceil_log2((uint32_t)n);
  rc = zo_hadamard_product_sumcheck(low, nsign, n, u_lk, v_lk, k, fr_out + p); p += 3 * k + 2;
  *n_fr = p;
  return rc;
}

int zo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
