/*
 * oracle/ref_wrap.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Pointer-based C entry points onto the REFERENCE'S OWN field/curve arithmetic,
 * i.e. /root/reference/bls12-381.cu compiled where it lies (see oracle/Makefile,
 * target _ref/libzkdl_ref.so).  Nothing from the reference is copied here: this
 * file only declares the prototypes of the functions that object file defines
 * (bls12-381.cuh:116-124 Fr, :218-226 Fp, :419-439 G1) and forwards to them.
 *
 * Used by tests/ and oracle/gen_golden.py in the build container to (a) validate
 * the restatement in zkdl_oracle.c limb-for-limb and (b) emit tests/golden/*.json.
 * /root/reference does not exist on the GPU box; the prebuilt .so travels instead.
 */
#include <stdint.h>

typedef struct { uint32_t val[8]; } S;    /* blstrs__scalar__Scalar, bls12-381.cuh:120 */
typedef struct { uint32_t val[12]; } F;   /* blstrs__fp__Fp, bls12-381.cuh:222 */
typedef struct { F x, y; } A;             /* blstrs__g1__G1Affine_affine, bls12-381.cuh:421 */
typedef struct { F x, y, z; } J;          /* blstrs__g1__G1Affine_jacobian, bls12-381.cuh:426 */

extern S blstrs__scalar__Scalar_add(S, S);
extern S blstrs__scalar__Scalar_sub(S, S);
extern S blstrs__scalar__Scalar_mul(S, S);
extern S blstrs__scalar__Scalar_sqr(S);
extern S blstrs__scalar__Scalar_double(S);
extern S blstrs__scalar__Scalar_mont(S);
extern S blstrs__scalar__Scalar_unmont(S);
extern F blstrs__fp__Fp_add(F, F);
extern F blstrs__fp__Fp_sub(F, F);
extern F blstrs__fp__Fp_mul(F, F);
extern F blstrs__fp__Fp_sqr(F);
extern F blstrs__fp__Fp_double(F);
extern F blstrs__fp__Fp_mont(F);
extern F blstrs__fp__Fp_unmont(F);
extern J blstrs__g1__G1Affine_double(J);
extern J blstrs__g1__G1Affine_add(J, J);
extern J blstrs__g1__G1Affine_add_mixed(J, A);
extern const S blstrs__scalar__Scalar_ONE, blstrs__scalar__Scalar_P, blstrs__scalar__Scalar_R2;
extern const F blstrs__fp__Fp_ONE, blstrs__fp__Fp_P, blstrs__fp__Fp_R2;

#define EXPORT __attribute__((visibility("default")))

EXPORT void ref_fr_add(const S* a, const S* b, S* r) { *r = blstrs__scalar__Scalar_add(*a, *b); }
EXPORT void ref_fr_sub(const S* a, const S* b, S* r) { *r = blstrs__scalar__Scalar_sub(*a, *b); }
EXPORT void ref_fr_mul(const S* a, const S* b, S* r) { *r = blstrs__scalar__Scalar_mul(*a, *b); }
EXPORT void ref_fr_sqr(const S* a, S* r) { *r = blstrs__scalar__Scalar_sqr(*a); }
EXPORT void ref_fr_double(const S* a, S* r) { *r = blstrs__scalar__Scalar_double(*a); }
EXPORT void ref_fr_mont(const S* a, S* r) { *r = blstrs__scalar__Scalar_mont(*a); }
EXPORT void ref_fr_unmont(const S* a, S* r) { *r = blstrs__scalar__Scalar_unmont(*a); }
EXPORT void ref_fp_add(const F* a, const F* b, F* r) { *r = blstrs__fp__Fp_add(*a, *b); }
EXPORT void ref_fp_sub(const F* a, const F* b, F* r) { *r = blstrs__fp__Fp_sub(*a, *b); }
EXPORT void ref_fp_mul(const F* a, const F* b, F* r) { *r = blstrs__fp__Fp_mul(*a, *b); }
EXPORT void ref_fp_sqr(const F* a, F* r) { *r = blstrs__fp__Fp_sqr(*a); }
EXPORT void ref_fp_double(const F* a, F* r) { *r = blstrs__fp__Fp_double(*a); }
EXPORT void ref_fp_mont(const F* a, F* r) { *r = blstrs__fp__Fp_mont(*a); }
EXPORT void ref_fp_unmont(const F* a, F* r) { *r = blstrs__fp__Fp_unmont(*a); }
EXPORT void ref_g1_double(const J* a, J* r) { *r = blstrs__g1__G1Affine_double(*a); }
EXPORT void ref_g1_add(const J* a, const J* b, J* r) { *r = blstrs__g1__G1Affine_add(*a, *b); }
EXPORT void ref_g1_add_mixed(const J* a, const A* b, J* r) { *r = blstrs__g1__G1Affine_add_mixed(*a, *b); }

/* constants as the reference's object file holds them (bls12-381.cu:3-11) */
EXPORT void ref_constants(S* fr_one, S* fr_p, S* fr_r2, F* fp_one, F* fp_p, F* fp_r2) {
  *fr_one = blstrs__scalar__Scalar_ONE; *fr_p = blstrs__scalar__Scalar_P; *fr_r2 = blstrs__scalar__Scalar_R2;
  *fp_one = blstrs__fp__Fp_ONE; *fp_p = blstrs__fp__Fp_P; *fp_r2 = blstrs__fp__Fp_R2;
}
