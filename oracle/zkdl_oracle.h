/*
 * oracle/zkdl_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the zkDL prover hot path, function by function, each citing the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.  The product
 * (zkdl_amd/csrc, libzkdl_amd.so) never includes, links or calls anything in oracle/.
 *
 * Parity pin (see DESIGN.md "Oracle"): (1) limb-for-limb agreement with the reference's own
 * compiled arithmetic (oracle/_ref, built from /root/reference/bls12-381.cu) on random + edge
 * operands, frozen as tests/golden/prims.json; (2) an independent big-integer model of the
 * field/curve in tests/ (Python ints, affine group law); (3) protocol invariants
 * (SURVEY.md App. E).  The reference ships no tests or fixtures of its own.
 */
#ifndef ZKDL_ORACLE_H
#define ZKDL_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* byte-compatible with Fr_t / Fp_t / G1Affine_t / G1Jacobian_t (bls12-381.cuh:120,222,419-430) */
typedef struct { uint32_t v[8]; } zo_fr;
typedef struct { uint32_t v[12]; } zo_fp;
typedef struct { zo_fp x, y; } zo_g1a;
typedef struct { zo_fp x, y, z; } zo_g1j;

/* ---- primitives (bls12-381.cu) ---- */
void zo_fr_add(const zo_fr* a, const zo_fr* b, zo_fr* r);
void zo_fr_sub(const zo_fr* a, const zo_fr* b, zo_fr* r);
void zo_fr_mul(const zo_fr* a, const zo_fr* b, zo_fr* r);
void zo_fr_sqr(const zo_fr* a, zo_fr* r);
void zo_fr_double(const zo_fr* a, zo_fr* r);
void zo_fr_mont(const zo_fr* a, zo_fr* r);
void zo_fr_unmont(const zo_fr* a, zo_fr* r);
void zo_fp_add(const zo_fp* a, const zo_fp* b, zo_fp* r);
void zo_fp_sub(const zo_fp* a, const zo_fp* b, zo_fp* r);
void zo_fp_mul(const zo_fp* a, const zo_fp* b, zo_fp* r);
void zo_fp_sqr(const zo_fp* a, zo_fp* r);
void zo_fp_double(const zo_fp* a, zo_fp* r);
void zo_fp_mont(const zo_fp* a, zo_fp* r);
void zo_fp_unmont(const zo_fp* a, zo_fp* r);
void zo_g1_double(const zo_g1j* a, zo_g1j* r);
void zo_g1_add(const zo_g1j* a, const zo_g1j* b, zo_g1j* r);
void zo_g1_add_mixed(const zo_g1j* a, const zo_g1a* b, zo_g1j* r);
void zo_g1_neg(const zo_g1j* a, zo_g1j* r);
void zo_g1_mul(const zo_g1j* a, const zo_fr* x, zo_g1j* r); /* g1-tensor.cu:422-430 */
void zo_constants(zo_fr* fr_one, zo_fr* fr_p, zo_fr* fr_r2, zo_fp* fp_one, zo_fp* fp_p, zo_fp* fp_r2);
void zo_g1_generator(zo_g1j* g); /* g1-tensor.cuh:29-63 */
/* affine-canonical form (the build's definition of G1 parity): x=X/Z^2, y=Y/Z^3 in
 * Montgomery limbs; *inf = 1 and x=y=0 when Z==0. Not in the reference (it never normalises). */
void zo_g1_to_affine(const zo_g1j* a, size_t n, zo_g1a* out, uint8_t* inf);

/* ---- Fr tensor ops (fr-tensor.cu) ---- */
void zo_fr_vadd(const zo_fr* a, const zo_fr* b, zo_fr* out, size_t n);
void zo_fr_vsub(const zo_fr* a, const zo_fr* b, zo_fr* out, size_t n);
void zo_fr_vmul(const zo_fr* a, const zo_fr* b, zo_fr* out, size_t n);
void zo_fr_vneg(const zo_fr* a, zo_fr* out, size_t n);
void zo_fr_vadd_s(const zo_fr* a, const zo_fr* x, zo_fr* out, size_t n);
void zo_fr_vsub_s(const zo_fr* a, const zo_fr* x, zo_fr* out, size_t n);
void zo_fr_vmul_s(const zo_fr* a, const zo_fr* x, zo_fr* out, size_t n);
void zo_fr_vmont(const zo_fr* a, zo_fr* out, size_t n);
void zo_fr_vunmont(const zo_fr* a, zo_fr* out, size_t n);
void zo_fr_sum(const zo_fr* a, size_t n, zo_fr* out);
/* returns 0, or -1 for the reference's "Incompatible dimensions" throw */
int zo_fr_me(const zo_fr* t, size_t n, const zo_fr* u, size_t k, zo_fr* out);
/* out must hold zo_fr_partial_me_size() elements */
size_t zo_fr_partial_me_size(size_t n, size_t k, size_t window);
int zo_fr_partial_me(const zo_fr* t, size_t n, const zo_fr* u, size_t k, size_t window, zo_fr* out);
int zo_fr_split(const zo_fr* t, size_t n, size_t window, zo_fr* out0, zo_fr* out1);

/* ---- sumchecks (proof.cu) ; proof arrays sized by caller: ip 3k+2, hp 3k+2, bin 3k+1 ---- */
int zo_inner_product_sumcheck(const zo_fr* a, const zo_fr* b, size_t n, const zo_fr* u, size_t k, zo_fr* proof);
int zo_hadamard_product_sumcheck(const zo_fr* a, const zo_fr* b, size_t n, const zo_fr* u, const zo_fr* v, size_t k, zo_fr* proof);
int zo_binary_sumcheck(const zo_fr* a, size_t n, const zo_fr* u, const zo_fr* v, size_t k, zo_fr* proof);
uint32_t zo_ceil_log2(uint32_t num);

/* ---- G1 tensor ops (g1-tensor.cu) ---- */
void zo_g1_vadd(const zo_g1j* a, const zo_g1j* b, zo_g1j* out, size_t n);
void zo_g1_vsub(const zo_g1j* a, const zo_g1j* b, zo_g1j* out, size_t n);
void zo_g1_vneg(const zo_g1j* a, zo_g1j* out, size_t n);
void zo_g1_vadd_s(const zo_g1j* a, const zo_g1j* x, zo_g1j* out, size_t n);
void zo_g1_vsub_s(const zo_g1j* a, const zo_g1j* x, zo_g1j* out, size_t n);
void zo_g1_vmadd(const zo_g1j* a, const zo_g1a* b, zo_g1j* out, size_t n);
void zo_g1_vmsub(const zo_g1j* a, const zo_g1a* b, zo_g1j* out, size_t n);
void zo_g1_vmadd_s(const zo_g1j* a, const zo_g1a* x, zo_g1j* out, size_t n);
void zo_g1_vmsub_s(const zo_g1j* a, const zo_g1a* x, zo_g1j* out, size_t n);
void zo_g1_from_affine(const zo_g1a* a, zo_g1j* out, size_t n);
void zo_g1_sum(const zo_g1j* a, size_t n, zo_g1j* out);           /* reference-shaped tree */
void zo_g1_vmul(const zo_g1j* g, const zo_fr* s, zo_g1j* out, size_t n); /* elementwise_mul */
void zo_g1_vmul_bcast(const zo_g1j* g, size_t n, const zo_fr* s, size_t total, zo_g1j* out);
int zo_g1_me(const zo_g1j* t, size_t n, const zo_fr* u, size_t k, zo_g1j* out);

/* ---- commitment (commitment.cu) ---- */
/* intended row-MSM semantics: out[row] = sum_j G[j]*unmont(t[row*n+j]) */
int zo_commit(const zo_g1j* gens, size_t n, const zo_fr* t, size_t tn, zo_g1j* out);
/* literal as-launched semantics (one THREAD per row): writes only out[64*b]; written[i]=1 there */
int zo_commit_as_launched(const zo_g1j* gens, size_t n, const zo_fr* t, size_t tn, zo_g1j* out, uint8_t* written);
/* proof holds 3k+1 points; returns t(0) in *ret */
int zo_me_open(const zo_fr* t, const zo_g1j* gens, size_t n, const zo_fr* u, size_t k, zo_g1j* proof, zo_fr* ret);
/* open: com_eval = com(u_hi) (the value the reference computes and drops); proof 3*log2(n)+1 pts */
int zo_open(const zo_g1j* gens, size_t n, const zo_fr* t, size_t tn, const zo_g1j* com, size_t m,
            const zo_fr* u, size_t k, zo_g1j* com_eval, zo_g1j* proof, zo_fr* ret);

/* ---- layers (zkfc.cu, zkrelu.cu) ---- */
void zo_float_to_fr(const float* fs, zo_fr* frs, uint32_t fs_rows, uint32_t frs_rows, uint32_t fs_cols, uint32_t frs_cols);
void zo_fc_forward(const zo_fr* X, const zo_fr* W, uint32_t batch, uint32_t in, uint32_t out, zo_fr* Z);
void zo_relu_forward(const zo_fr* X, size_t n, zo_fr* Z, zo_fr* sign, zo_fr* mag_bin, zo_fr* rem_bin);

/* Transcript of one zkFC::prove (order: SURVEY.md App. A.6):
 *   fr_out: [ip sumcheck 3*log(in)+2][Z(u_Z) 1][me_open return 1]
 *   g1_out: [com(u_hi) 1][me_open proof 3*log(n_gens)+1]                      */
int zo_fc_prove(const zo_fr* X, size_t xn, const zo_fr* Z, size_t zn, const zo_fr* W, uint32_t in, uint32_t out,
                const zo_g1j* gens, size_t n_gens, const zo_g1j* com, size_t m,
                const zo_fr* u_bs, size_t k_bs, const zo_fr* u_in, size_t k_in, const zo_fr* u_out, size_t k_out,
                zo_fr* fr_out, size_t* n_fr, zo_g1j* g1_out, size_t* n_g1);
/* Transcript of one zkReLU::prove:
 *   fr_out: [bin_sc(mag) 3(k+5)+1][partial_me(mag,32) 32][bin_sc(rem) 3(k+4)+1][partial_me(rem,16) 16][hp_sc 3k+2] */
int zo_relu_prove(const zo_fr* X, const zo_fr* Z, size_t n, const zo_fr* sign, const zo_fr* mag_bin, const zo_fr* rem_bin,
                  const zo_fr* u_z, const zo_fr* v_z, const zo_fr* u_r, const zo_fr* v_r, const zo_fr* u_rec,
                  const zo_fr* u_hp, const zo_fr* v_hp, zo_fr* fr_out, size_t* n_fr);
/* extension (no reference counterpart): LeakyReLU with slope 2^-shift on the negative branch; see zkdl_oracle.c */
void zo_leaky_forward(const zo_fr* X, size_t n, uint32_t shift, zo_fr* Z, zo_fr* sign, zo_fr* mag_bin, zo_fr* rem_bin, zo_fr* low, zo_fr* nsign);
int zo_leaky_prove(const zo_fr* X, const zo_fr* Z, size_t n, const zo_fr* sign, const zo_fr* mag_bin, const zo_fr* rem_bin, const zo_fr* low, const zo_fr* nsign,
                   const zo_fr* u_z, const zo_fr* v_z, const zo_fr* u_r, const zo_fr* v_r, const zo_fr* u_rec,
                   const zo_fr* u_hp, const zo_fr* v_hp, const zo_fr* u_lk, const zo_fr* v_lk, zo_fr* fr_out, size_t* n_fr);

int zo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
