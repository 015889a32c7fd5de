/*
 * include/zkdl_amd.h -- C ABI of libzkdl_amd.so, the MI355X (gfx950) implementation of the zkDL
 * proving hot path.  Plain pointers and sizes only; every call returns an int status and never
 * throws.  The reference (SafeAILab/zkDL) has no FFI layer: its boundary is the C++ class API of
 * fr-tensor.cuh / g1-tensor.cuh / proof.cuh / commitment.cuh / zkfc.cuh / zkrelu.cuh.  Each entry
 * point below names the reference interface it replaces; zkdl_amd/host/ *.hpp rebuilds those C++
 * classes (same names, argument meaning and error behaviour) on top of this ABI, and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *  - POD element types are byte-compatible with the reference's Fr_t / Fp_t / G1Affine_t /
 *    G1Jacobian_t (bls12-381.cuh:120,222,419-430): little-endian uint32 limbs, Montgomery form.
 *  - Pointers named *_d are DEVICE pointers (from zk_malloc or any hipMalloc'd memory), 16-byte
 *    aligned; pointers named *_h are HOST pointers.  Challenge vectors are host arrays, LSB-first
 *    (u[0] folds index bit 0), exactly like the reference's vector<Fr_t> arguments.
 *  - All work is enqueued on the context's HIP stream.  Calls that return values to the host
 *    synchronise that stream; calls that only write device memory do not.
 *  - Shape violations return ZK_ERR_SHAPE where the reference throws
 *    std::runtime_error("Incompatible dimensions ...").
 */
#ifndef ZKDL_AMD_H
#define ZKDL_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The limb array is reachable as .v (this ABI) and as .val (the reference's member name, bls12-381.cuh:120,222),
 * so reference-style code such as zkrelu.cu:3-9 (`x.val[0]`) compiles against the C++ mirror unchanged. */
typedef struct { union { uint32_t v[8]; uint32_t val[8]; }; } zk_fr;      /* Fr_t   (fr-tensor.cuh:13) */
typedef struct { union { uint32_t v[12]; uint32_t val[12]; }; } zk_fp;    /* Fp_t   (g1-tensor.cuh:10) */
typedef struct { zk_fp x, y; } zk_g1a;          /* G1Affine_t    (fr-tensor.cuh:14) */
typedef struct { zk_fp x, y, z; } zk_g1j;       /* G1Jacobian_t  (fr-tensor.cuh:15) */

typedef struct zk_ctx zk_ctx;                   /* device + stream + pools + RNG */
typedef struct zk_gens zk_gens;                 /* Commitment generators (commitment.cuh:8-23) + MSM tables */
typedef struct zk_fc zk_fc;                     /* zkFC   (zkfc.cuh:16-33)   */
typedef struct zk_relu zk_relu;                 /* zkReLU (zkrelu.cuh:10-20) */
typedef struct zk_transcript zk_transcript;     /* proof values the reference computes and drops */

enum {
  ZK_OK = 0,
  ZK_ERR_SHAPE = 1,   /* reference: throw std::runtime_error("Incompatible dimensions...") */
  ZK_ERR_ALLOC = 2,   /* reference: silent nullptr from cudaMalloc (fr-tensor.cu:94)        */
  ZK_ERR_DEVICE = 3,  /* reference: ignored CUDA error, printed at exit (demo.cu:141)        */
  ZK_ERR_ARG = 4,     /* null / misaligned pointer, bad enum                                */
  ZK_ERR_RANGE = 5    /* value outside the domain the reference leaves undefined (App. D)   */
};

/* ---------------------------------------------------------------- context, memory */
int zk_ctx_create(int device, zk_ctx** out);
int zk_ctx_destroy(zk_ctx* ctx);
int zk_ctx_sync(zk_ctx* ctx);
const char* zk_last_error(const zk_ctx* ctx);     /* message of the last non-OK status */
const char* zk_version(void);
void* zk_ctx_stream(zk_ctx* ctx);                 /* the hipStream_t, for callers that interleave */
/* cudaMalloc / cudaFree / cudaMemcpy in FrTensor and G1Tensor ctors (fr-tensor.cu:92-120) */
int zk_malloc(zk_ctx* ctx, size_t bytes, void** out_d);
int zk_free(zk_ctx* ctx, void* ptr_d);
int zk_memcpy_h2d(zk_ctx* ctx, void* dst_d, const void* src_h, size_t bytes);
int zk_memcpy_d2h(zk_ctx* ctx, void* dst_h, const void* src_d, size_t bytes);   /* synchronises */
int zk_memcpy_d2d(zk_ctx* ctx, void* dst_d, const void* src_d, size_t bytes);
int zk_memset_zero(zk_ctx* ctx, void* dst_d, size_t bytes);

/* HIP-event timing on the context's stream (bench.py roofline leg) */
int zk_timer_start(zk_ctx* ctx);
int zk_timer_stop_ms(zk_ctx* ctx, float* ms_out);  /* synchronises */
/* per-kernel-family device time accumulated with HIP events when enabled (debug/bench only) */
int zk_prof_enable(zk_ctx* ctx, int on);
int zk_prof_reset(zk_ctx* ctx);
/* writes up to cap entries; names are static strings; returns count in *n */
int zk_prof_read(zk_ctx* ctx, const char** names, float* ms, uint64_t* launches, uint64_t* bytes, int cap, int* n);

/* ---------------------------------------------------------------- challenges / RNG */
/* random_vec (proof.cu:3-11) and FrTensor::random (fr-tensor.cu:337-368) draw 8 uniform limbs with
 * the top limb reduced modulo 1944954707.  Here the stream is a seeded counter-based generator so
 * runs are reproducible; callers may instead pass their own challenge arrays everywhere. */
int zk_rng_seed(zk_ctx* ctx, uint64_t seed);
int zk_random_vec(zk_ctx* ctx, uint32_t len, zk_fr* out_h);              /* random_vec */
int zk_fr_random(zk_ctx* ctx, zk_fr* out_d, uint32_t n);                  /* FrTensor::random */
int zk_fr_random_int(zk_ctx* ctx, zk_fr* out_d, uint32_t n, uint32_t num_bits); /* FrTensor::random_int */
uint32_t zk_ceil_log2(uint32_t num);                                       /* ceilLog2, proof.cu:13-30 */

/* ---------------------------------------------------------------- Fr tensor (fr-tensor.cuh:62-133) */
int zk_fr_add(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* b_d, zk_fr* out_d, uint32_t n);      /* operator+ / += */
int zk_fr_sub(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* b_d, zk_fr* out_d, uint32_t n);      /* operator- / -= */
int zk_fr_mul(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* b_d, zk_fr* out_d, uint32_t n);      /* operator* / *= */
int zk_fr_neg(zk_ctx* ctx, const zk_fr* a_d, zk_fr* out_d, uint32_t n);                        /* unary -        */
int zk_fr_add_scalar(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* x_h, zk_fr* out_d, uint32_t n);
int zk_fr_sub_scalar(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* x_h, zk_fr* out_d, uint32_t n);
int zk_fr_mul_scalar(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* x_h, zk_fr* out_d, uint32_t n);
int zk_fr_mont(zk_ctx* ctx, const zk_fr* a_d, zk_fr* out_d, uint32_t n);                       /* mont()   */
int zk_fr_unmont(zk_ctx* ctx, const zk_fr* a_d, zk_fr* out_d, uint32_t n);                     /* unmont() */
int zk_fr_sum(zk_ctx* ctx, const zk_fr* a_d, uint32_t n, zk_fr* out_h);                        /* sum()    */
/* FrTensor::operator()(const vector<Fr_t>&): multilinear evaluation; size must be in (2^(k-1), 2^k] */
int zk_fr_me(zk_ctx* ctx, const zk_fr* t_d, uint32_t n, const zk_fr* u_h, uint32_t k, zk_fr* out_h);
/* FrTensor::partial_me: *out_n receives the result length; out_d must hold zk_fr_partial_me_size() */
uint32_t zk_fr_partial_me_size(uint32_t n, uint32_t k, uint32_t window);
int zk_fr_partial_me(zk_ctx* ctx, const zk_fr* t_d, uint32_t n, const zk_fr* u_h, uint32_t k, uint32_t window, zk_fr* out_d);
/* FrTensor::split (fr-tensor.cu:389-397); outputs hold (n+1)/2 elements each */
int zk_fr_split(zk_ctx* ctx, const zk_fr* t_d, uint32_t n, uint32_t window, zk_fr* out0_d, zk_fr* out1_d);

/* ---------------------------------------------------------------- sumchecks (proof.cuh:13-34) */
/* proof_h receives 3k+2 (inner product, Hadamard) or 3k+1 (binary) elements */
int zk_sumcheck_ip(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* b_d, uint32_t n, const zk_fr* u_h, uint32_t k, zk_fr* proof_h);
int zk_sumcheck_hadamard(zk_ctx* ctx, const zk_fr* a_d, const zk_fr* b_d, uint32_t n, const zk_fr* u_h, const zk_fr* v_h, uint32_t k, zk_fr* proof_h);
int zk_sumcheck_binary(zk_ctx* ctx, const zk_fr* a_d, uint32_t n, const zk_fr* u_h, const zk_fr* v_h, uint32_t k, zk_fr* proof_h);

/* ---------------------------------------------------------------- G1 tensor (g1-tensor.cuh:76-174) */
int zk_g1_fill(zk_ctx* ctx, zk_g1j* out_d, uint32_t n, const zk_g1j* g_h);                   /* ctor(size, g)      */
int zk_g1a_fill(zk_ctx* ctx, zk_g1a* out_d, uint32_t n, const zk_g1a* g_h);
int zk_g1_from_affine(zk_ctx* ctx, const zk_g1a* a_d, zk_g1j* out_d, uint32_t n);            /* ctor(G1TensorAffine) */
int zk_g1_neg(zk_ctx* ctx, const zk_g1j* a_d, zk_g1j* out_d, uint32_t n);                    /* unary -            */
int zk_g1a_neg(zk_ctx* ctx, const zk_g1a* a_d, zk_g1a* out_d, uint32_t n);
int zk_g1_add(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1j* b_d, zk_g1j* out_d, uint32_t n);  /* + / += (Jacobian)   */
int zk_g1_sub(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1j* b_d, zk_g1j* out_d, uint32_t n);
int zk_g1_madd(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1a* b_d, zk_g1j* out_d, uint32_t n); /* + / += (affine)     */
int zk_g1_msub(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1a* b_d, zk_g1j* out_d, uint32_t n);
int zk_g1_add_bcast(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1j* x_h, zk_g1j* out_d, uint32_t n);
int zk_g1_sub_bcast(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1j* x_h, zk_g1j* out_d, uint32_t n);
int zk_g1_madd_bcast(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1a* x_h, zk_g1j* out_d, uint32_t n);
int zk_g1_msub_bcast(zk_ctx* ctx, const zk_g1j* a_d, const zk_g1a* x_h, zk_g1j* out_d, uint32_t n);
/* G1TensorJacobian::sum: reduction tree shaped like the reference's (g1-tensor.cu:368-420), so the
 * Jacobian limbs match it bit for bit */
int zk_g1_sum(zk_ctx* ctx, const zk_g1j* a_d, uint32_t n, zk_g1j* out_h);
/* operator*= (elementwise) and operator* (generators broadcast over m rows): 256-step scalar-muls
 * on the raw limb bits, as g1-tensor.cu:422-461 */
int zk_g1_mul(zk_ctx* ctx, const zk_g1j* g_d, const zk_fr* s_d, zk_g1j* out_d, uint32_t n);
int zk_g1_mul_bcast(zk_ctx* ctx, const zk_g1j* g_d, uint32_t n, const zk_fr* s_d, uint32_t total, zk_g1j* out_d);
/* G1TensorJacobian::operator()(const vector<Fr_t>&) */
int zk_g1_me(zk_ctx* ctx, const zk_g1j* t_d, uint32_t n, const zk_fr* u_h, uint32_t k, zk_g1j* out_h);
/* affine-canonical form (x = X/Z^2, y = Y/Z^3, Montgomery limbs; inf flag, x = y = 0 when Z == 0):
 * the build's definition of G1 parity -- the reference never normalises */
int zk_g1_to_affine(zk_ctx* ctx, const zk_g1j* a_d, uint32_t n, zk_g1a* out_d, uint8_t* inf_d);

/* ---------------------------------------------------------------- commitment (commitment.cuh:8-23) */
/* MSM algorithm selection for commit/open. LITERAL = the reference's op sequence (one 256-step
 * double-and-add per pair, reference-shaped sums); PIPPENGER = bucket method (same group elements,
 * different Jacobian representatives). */
enum { ZK_MSM_LITERAL = 0, ZK_MSM_PIPPENGER = 1 };
int zk_set_msm_mode(zk_ctx* ctx, int mode);
int zk_set_msm_window(zk_ctx* ctx, uint32_t window_bits);   /* 0 = automatic; else 2 .. 13 (ZK_ERR_ARG otherwise) */
/* Scheduling / representation switches, on by default unless noted; none of them changes a transcript value:
 *   "relu_packed"   ReLU aux sumchecks and partial_me read the packed bits instead of the one-Fr-per-bit tensors
 *   "relu_bytes"    ... and take rounds 0-4 (32-bit words) / 0-3 (16-bit) from weighted byte histograms instead of the
 *                   three-round lookup-table form
 *   "derive_rounds" (default 0) the opening's round commitments T_1 .. T_{k-1} are derived from T_0, T0_j, T1_j by the folding
 *                   relation (k^2 - 1 scalar multiplications behind the MSM batch) instead of run as k - 1 full-size MSMs
 *   "batch_layers"  zk_fc_prove / zk_relu_prove record their launches; flush/seal runs them batched across layers
 *   "lanes"         worker streams (the recorder's groups, the G1 streams of zk_transcript_seal); off = one stream
 *   "g1_batch"      (default 1) the G1 phases of that many consecutively sealed proofs run as ONE batch (one MSM batch, one derivation,
 *                   one canonicalisation): the latency-bound launches behind the accumulation are shared; a sealed proof then waits for
 *                   its partners (or for anyone touching its transcript, or zk_ctx_sync).  Throughput configuration of bench.py and zkdl::ProofPipeline: 3.
 *   "g1_mask"       (default 1) sealed proofs' G1 phases run on streams confined to every other compute unit, so that the chip-filling accumulation
 *                   leaves room for the next proofs' Fr chains; 0 = an unmasked stream: what a single proof with nothing in flight wants (latency)
 *   "g1_streams"    (default 0 = all) how many of the context's G1 streams sealed proofs rotate over
 *   "early_g1"      a sealed proof's G1 phase starts behind the openings' scalar preparation (all it needs) instead of behind the whole
 *                   Fr phase; only the digest waits for the rest
 *   "api_join"      (default 1) an immediate tensor call (zk_fr_*, zk_fc_forward, ...) first executes pending recordings and makes the main stream wait
 *                   for every worker stream, so it may read what recorded work wrote; 0: it is launched behind earlier main-stream work only -- for
 *                   callers whose operands do not come from recorded work still in flight (zkdl_amd/shard.py: the streamed phases)
 *   "weights_q16"   layers created while set keep an int16 copy of quantised weights (|w| < 2^15) for the
 *                   prover's two contractions of W and the integer forward pass */
int zk_set_flag(zk_ctx* ctx, const char* name, int value);
/* Commitment object (commitment.cuh:8-23): a generator tensor.  The handle keeps its own copy of the
 * n Jacobian generators and, for the PIPPENGER back end, lazily builds affine window tables
 * 2^(c*w) * G_j once (all later MSMs over these generators share them). */
int zk_gens_create(zk_ctx* ctx, const zk_g1j* gens_d, uint32_t n, zk_gens** out);
int zk_gens_destroy(zk_gens* g);
/* Optional setup step for generator sets that will see many full-width MSMs (the openings of zk_fc_prove, zk_act_open, zk_msm_batch): precompute
 * EVERY signed digit of every window, d * 2^(c w) * G_j for d = 1 .. 2^(c-1) (c = window_bits, 4 .. 14; 22 windows at c = 12, one 128-byte slot per entry: 5.8 MB per
 * generator, 11.8 GB for 2048 generators -- sized for 288 GB of HBM).  Full-width MSMs over this handle then run without buckets: mixed
 * additions of table entries and one sum of lane partials per MSM (no sort, no bucket reduction, 22 instead of 26 windows).  Same group
 * elements.  Commitments of quantised tensors (short scalars) keep the bucket method.  The flag "full_tables" (default 1) turns the path off
 * without freeing the tables. */
int zk_gens_prepare_full(zk_ctx* ctx, zk_gens* g, uint32_t window_bits);
uint32_t zk_gens_size(const zk_gens* g);
const zk_g1j* zk_gens_points(const zk_gens* g);   /* device pointer to the n Jacobian generators */
/* Commitment::commit with the INTENDED row semantics:
 * out[row] = sum_j gens[j] * unmont(t[row*n + j]);  tn % n == 0, out holds tn/n points */
int zk_commit(zk_ctx* ctx, zk_gens* gens, const zk_fr* t_d, uint32_t tn, zk_g1j* out_d);
/* Commitment::me_open: t holds n = zk_gens_size scalars; proof_h receives 3k+1 points, *ret_h = t(0) */
int zk_me_open(zk_ctx* ctx, zk_gens* gens, const zk_fr* t_d, const zk_fr* u_h, uint32_t k, zk_g1j* proof_h, zk_fr* ret_h);
/* Commitment::open: com_eval_h = com(u_hi) (computed and dropped by the reference) */
int zk_open(zk_ctx* ctx, zk_gens* gens, const zk_fr* t_d, uint32_t tn, const zk_g1j* com_d, uint32_t m,
            const zk_fr* u_h, uint32_t k, zk_g1j* com_eval_h, zk_g1j* proof_h, zk_fr* ret_h);
/* MSM throughput probe for the "G1 MSM Mop/s" metric: q independent MSMs over the handle's n
 * generators, scalars_d holds q*n canonical Fr values used as raw integers; out_d receives q points */
int zk_msm_batch(zk_ctx* ctx, zk_gens* gens, const zk_fr* scalars_d, uint32_t q, zk_g1j* out_d);

/* ---------------------------------------------------------------- layers */
/* float_to_Fr_kernel (zkfc.cu:80-88): quantise rows x cols floats (device) into a zero-padded
 * rows_p x cols_p Fr tensor, 16 fractional bits, NOT yet Montgomery */
int zk_float_to_fr(zk_ctx* ctx, const float* fs_d, zk_fr* out_d, uint32_t rows, uint32_t rows_p, uint32_t cols, uint32_t cols_p);

/* zkFC (zkfc.cuh:16-33).  weights_d: in*out Montgomery elements, row-major [in][out]; the object
 * keeps its own copy and commits to it with gens (constructor, zkfc.cu:102-104). */
int zk_fc_create(zk_ctx* ctx, uint32_t in, uint32_t out, const zk_fr* weights_d, zk_gens* gens, zk_fc** out_fc);
/* zkFC::from_float_gpu_ptr (zkfc.cu:90-100): pads to powers of two, quantises, mont(), commits */
int zk_fc_from_float(zk_ctx* ctx, uint32_t in, uint32_t out, const float* w_d, zk_gens* gens, zk_fc** out_fc);
int zk_fc_destroy(zk_fc* fc);
uint32_t zk_fc_in(const zk_fc* fc);
uint32_t zk_fc_out(const zk_fc* fc);
uint32_t zk_fc_com_size(const zk_fc* fc);
const zk_fr* zk_fc_weights(const zk_fc* fc);     /* device pointer */
const zk_g1j* zk_fc_com(const zk_fc* fc);        /* device pointer, com_size points */
/* zkFC::load_float_gpu_input (zkfc.cu:106-115) */
int zk_fc_load_input(zk_ctx* ctx, uint32_t batch, uint32_t in_dim, const float* x_d, zk_fr* out_d);
/* zkFC::operator() (zkfc.cu:117-126): Z[B x out] = X[B x in] * W; xn % in == 0 */
int zk_fc_forward(zk_ctx* ctx, const zk_fc* fc, const zk_fr* x_d, uint32_t xn, zk_fr* z_d);
/* zkFC::prove (zkfc.cu:128-145) with the three challenge vectors made explicit
 * (k_bs = ceilLog2(batch), k_in = ceilLog2(in), k_out = ceilLog2(out)); appends to tr:
 *   Fr: [ip sumcheck 3*k_in+2][Z(u_out||u_bs)][me_open return]   G1: [com(u_hi)][me_open 3*log2(n_gens)+1] */
int zk_fc_prove(zk_ctx* ctx, zk_fc* fc, const zk_fr* x_d, uint32_t xn, const zk_fr* z_d, uint32_t zn,
                zk_gens* gens, const zk_fr* u_bs_h, const zk_fr* u_in_h, const zk_fr* u_out_h, zk_transcript* tr);

/* zkReLU (zkrelu.cuh:10-20): owns sign / mag_bin / rem_bin produced by the forward pass */
int zk_relu_create(zk_ctx* ctx, zk_relu** out_relu);
int zk_relu_destroy(zk_relu* relu);
/* zkReLU::operator() (zkrelu.cu:44-51): returns ZK_ERR_RANGE (after completing) if any input lies
 * outside the +-2^47 window where the reference's kernel is undefined */
int zk_relu_forward(zk_ctx* ctx, zk_relu* relu, const zk_fr* x_d, uint32_t n, zk_fr* z_d);
/* aux tensors as the reference stores them (one Fr per bit); device pointers valid until the next forward */
const zk_fr* zk_relu_sign(const zk_relu* relu);
const zk_fr* zk_relu_mag_bin(const zk_relu* relu);
const zk_fr* zk_relu_rem_bin(const zk_relu* relu);
/* zkReLU::prove (zkrelu.cu:79-100); k = ceilLog2(n); challenge lengths k+5,k+5,k+4,k+4,k,k,k; appends
 *   Fr: [bin(mag) 3(k+5)+1][partial_me(mag,32) 32][bin(rem) 3(k+4)+1][partial_me(rem,16) 16][hadamard 3k+2] */
int zk_relu_prove(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x_d, const zk_fr* z_d, uint32_t n,
                  const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h, const zk_fr* v_r_h, const zk_fr* u_rec_h,
                  const zk_fr* u_hp_h, const zk_fr* v_hp_h, zk_transcript* tr);

/* ---- activation kinds beyond the reference's ReLU (an extension: zkrelu.cu:11-41 is the reference's only activation).
 * kind 0 = ReLU (default), 1 = LeakyReLU with negative slope 2^-shift (shift 0: identity, i.e. a signed requantisation -- what a Linear layer
 * followed directly by another Linear layer needs).  Same decomposition as ReLU (sign, the 32 magnitude bits of the rescaled value, 16
 * remainder bits: zkrelu.cu:11-41).  For a negative input the reference's kernel decomposes x + 2^47, i.e. mag_rescaled = 2^31 + q is the offset
 * encoding of the requantised value q < 0, and floor(q / 2^shift) = (mag_rescaled >> shift) - 2^(31 - shift): magnitude bits shift..31 minus a
 * constant, so the bit tensors the prover already has cover it.  Takes effect at the next zk_relu_forward. */
int zk_relu_set_activation(zk_relu* relu, uint32_t kind, uint32_t shift);
/* zk_relu_prove's transcript followed by hadamard_product_sumcheck(low, 1 - sign, u_lk, v_lk) (proof.cu:110-150; 3k+2 more Fr values), low =
 * mont(mag_rescaled >> shift): the negative branch's magnitude (low o (1 - sign))~(u_lk), in the style of the reference's (Z o sign)~(u_hp).
 * A = mont(mag_rescaled) o sign + (low - 2^(31 - shift)) o (1 - sign). */
int zk_leaky_prove(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x_d, const zk_fr* z_d, uint32_t n,
                   const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h, const zk_fr* v_r_h, const zk_fr* u_rec_h,
                   const zk_fr* u_hp_h, const zk_fr* v_hp_h, const zk_fr* u_lk_h, const zk_fr* v_lk_h, zk_transcript* tr);
/* zk_verify_relu's checks plus the extra Hadamard sumcheck; claims_h[0..5) as zk_verify_relu, claims_h[5..8) = (low o nsign)~(u_lk), low~(v_lk),
 * nsign~(v_lk) */
int zk_verify_leaky(zk_ctx* ctx, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h, const zk_fr* v_r_h, const zk_fr* u_hp_h,
                    const zk_fr* v_hp_h, const zk_fr* u_lk_h, const zk_fr* v_lk_h, const zk_fr* fr_h, size_t n_fr, zk_fr* claims_h, int* ok);

/* ---------------------------------------------------------------- transcript */
int zk_transcript_create(zk_ctx* ctx, zk_transcript** out);
int zk_transcript_destroy(zk_transcript* tr);
int zk_transcript_reset(zk_transcript* tr);
/* With the PIPPENGER back end zk_fc_prove only ENQUEUES its MSMs; they run, all layers together, when
 * the transcript is flushed -- explicitly here, or implicitly by counts / read / digest / reset. */
int zk_transcript_flush(zk_transcript* tr);
/* synchronises; counts of values appended so far */
int zk_transcript_counts(zk_transcript* tr, size_t* n_fr, size_t* n_g1);
int zk_transcript_read(zk_transcript* tr, zk_fr* fr_h, size_t fr_cap, zk_g1j* g1_h, size_t g1_cap);
/* G1 entries in affine-canonical form (inf flags as bytes) */
int zk_transcript_read_affine(zk_transcript* tr, zk_g1a* g1_h, uint8_t* inf_h, size_t cap);
/* Fr checksum of the transcript: sum of all Fr entries plus, per G1 entry, its affine x and y limbs
 * folded into Fr -- the per-proof aggregate that ranks all-reduce in the multi-GPU bench */
int zk_transcript_digest(zk_transcript* tr, zk_fr* out_h);
/* the same digest left on the DEVICE as 8 uint64 lanes (one per 32-bit limb, zero-extended), enqueued
 * on the context's stream without a host round trip: ranks all-reduce these lanes with ncclSum on
 * uint64 (exact for < 2^32 ranks), then carry-propagate and reduce mod r locally */
int zk_transcript_digest_dev(zk_transcript* tr, uint64_t* out8_d);
/* Serialised transcript (the proof values the reference computes and drops, zkfc.cu:139-144): synchronises and
 * writes   "ZKDLTR01" | u64 n_fr | u64 n_g1 | n_fr x 32 B (Fr, Montgomery limbs, little endian, as Fr_t)
 *          | n_g1 x 96 B (affine x, y: Fp Montgomery limbs, little endian; zeros for infinity) | n_g1 x 1 B (infinity flags)
 * into buf_h (capacity cap bytes); *len receives the size (also when buf_h is NULL or too small: ZK_ERR_ARG then,
 * so a caller can size the buffer with a first call). */
int zk_transcript_serialize(zk_transcript* tr, void* buf_h, size_t cap, size_t* len);
/* Finish the proof ASYNCHRONOUSLY: the deferred MSM batch, the canonicalisation of its outputs and (when
 * out8_d is not NULL) the digest_dev above run on one of the context's G1 streams (zk_transcript_stream
 * tells which), behind the Fr work already enqueued, and the main stream does NOT wait for them.  A caller proving a stream of
 * batches alternates two transcripts: the G1 phase of proof i then runs under the Fr phase of proof i+1.
 * Any later call on the SAME transcript (read, digest, counts, reset, flush, destroy), and zk_ctx_sync,
 * waits for the sealed phase first; work that consumes out8_d must be ordered on zk_transcript_stream(tr).
 * Without worker lanes this degrades to flush + digest_dev on the main stream (same values). */
int zk_transcript_seal(zk_transcript* tr, uint64_t* out8_d);
/* Read a SEALED transcript (Fr values; G1 values affine-canonical + infinity flags) waiting for nothing but that transcript's own sealed phase: the other
 * proofs in flight keep running (zk_transcript_read / read_affine synchronise the whole context).  The transcript stays sealed. */
int zk_transcript_sizes(const zk_transcript* tr, size_t* n_fr, size_t* n_g1);   /* counts without flushing or waiting (zk_transcript_counts does both) */
int zk_transcript_read_sealed(zk_transcript* tr, zk_fr* fr_h, size_t fr_cap, zk_g1a* g1_h, uint8_t* inf_h, size_t g1_cap);
/* Two proofs submitted together: same values as zk_transcript_seal(a, ..) followed by zk_transcript_seal(b, ..), but the recorded
 * Fr work of both runs as ONE set of batched launches (each launch carries the jobs of both proofs), then each proof's G1 phase
 * goes to its own G1 stream.  No reference counterpart (the reference proves one batch at a time, synchronously); throughput
 * only: a proof's latency grows by its partner's Fr phase. */
int zk_transcript_seal_pair(zk_transcript* a, uint64_t* a_out8_d, zk_transcript* b, uint64_t* b_out8_d);
/* the hipStream_t the last zk_transcript_seal of this transcript was enqueued on (the main stream if it ran
 * synchronously or was never sealed) */
void* zk_transcript_stream(zk_transcript* tr);

/* ---------------------------------------------------------------- verifier, hash-derived challenges
 * The reference computes every proof value and drops it (zkfc.cu:139-144, zkrelu.cu:91-99); nothing in it verifies.
 * These entry points check a proof slice WITHOUT the witness.  Proof values are passed as the serialised transcript
 * holds them: Fr in Montgomery limbs, G1 in affine-canonical form + infinity flags (zk_transcript_serialize /
 * zk_transcript_read_affine).  The call returns ZK_OK when it could run; *ok = 1 if every check passed, 0 otherwise
 * (zk_last_error then names the first failed check).
 *
 * zk_verify_fc -- one zkFC::prove (zkfc.cu:128-145): fr_h = [ip sumcheck 3 k_in + 2][Z~(u_out||u_bs)][t(0)],
 * g1_h = [com(u_hi)][me_open 3 k_lo + 1].  Checks the inner-product sumcheck recurrence and final product
 * (proof.cu:55-96), t(0) = W~(u_out||u_in), and the opening against the PUBLIC commitment (commitment.cu:43-92):
 * com(u_hi) and the final generator are recomputed from com_d / gens, T_0 = [R] com(u_hi), every round's folding
 * relation, and T_last' = t(0) * g_final.  z_is_raw: the layer was proved on the un-Montgomery'd output (the last layer,
 * demo.cu:119,128).  expect_z_h / expect_x_h (optional): evaluations of the layer's public output / input the transcript's
 * claims must equal; claim_z_h / claim_x_h (optional) receive the two claims Z~(u_out||u_bs), X~(u_in||u_bs). */
int zk_verify_fc(zk_ctx* ctx, zk_gens* gens, const zk_g1j* com_d, uint32_t m, uint32_t batch, uint32_t in, uint32_t out, int z_is_raw,
                 const zk_fr* u_bs_h, const zk_fr* u_in_h, const zk_fr* u_out_h, const zk_fr* fr_h, size_t n_fr, const zk_g1a* g1_h, const uint8_t* inf_h,
                 size_t n_g1, const zk_fr* expect_z_h, const zk_fr* expect_x_h, zk_fr* claim_z_h, zk_fr* claim_x_h, int* ok);
/* zk_verify_open -- one Commitment::open (commitment.cu:83-92) on its own: g1_h = [com(u_hi)][me_open 3 k_lo + 1] as zk_open returns them
 * (affine-canonical), value_h the opened value t(0) (Montgomery form) = the committed tensor's multilinear evaluation at u (k = k_lo + k_hi
 * challenges: generator bits first, then commitment rows).  The checks are zk_verify_fc's opening checks; used for ACTIVATION commitments
 * (the reference's stated future work: commit Z_i / A_i with zk_commit, open them at the points the layer proofs leave as open claims). */
int zk_verify_open(zk_ctx* ctx, zk_gens* gens, const zk_g1j* com_d, uint32_t m, const zk_fr* u_h, uint32_t k, const zk_fr* value_h,
                   const zk_g1a* g1_h, const uint8_t* inf_h, size_t n_g1, int* ok);
/* zk_verify_relu -- one zkReLU::prove (zkrelu.cu:79-100) over n elements: fr_h as zk_relu_prove appends it.  Checks both
 * binary sumchecks (proof.cu:152-188) and the Hadamard sumcheck (proof.cu:110-137).  The reference binds none of the
 * ReLU's final evaluations to a commitment (activation commitments are its stated future work), so they are returned as
 * open claims: claims_h[0..5) = mag_bin~(v_z), rem_bin~(v_r), (Z o sign)~(u_hp), Z~(v_hp), sign~(v_hp). */
int zk_verify_relu(zk_ctx* ctx, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h, const zk_fr* v_r_h, const zk_fr* u_hp_h,
                   const zk_fr* v_hp_h, const zk_fr* fr_h, size_t n_fr, zk_fr* claims_h, int* ok);
/* Challenges as hash outputs in random_vec's distribution (proof.cu:3-11): out_h[i] = SHA-256("zkdl-amd/fs/v1" | seed32 |
 * layer | kind | vec | i) read as 8 little-endian limbs, top limb mod 1944954707.  One derivation per prove call and
 * challenge vector (kind 0 = zkFC, 1 = zkReLU; vec = position in the reference's draw order, zkfc.cu:135-137 /
 * zkrelu.cu:85-98), seeded by a digest of the public statement: reproducible by the verifier, independent of the call's
 * own round messages exactly like the reference's pre-drawn coins. */
int zk_fs_derive(const uint8_t* seed32, uint32_t layer, uint32_t kind, uint32_t vec, uint32_t len, zk_fr* out_h);
int zk_sha256(const void* data, size_t len, uint8_t* out32);

/* ---------------------------------------------------------------- diagnostics
 * The device field functions every kernel is built from, applied element-wise (one element per lane), so that known-answer vectors reach each of them
 * directly.  which: 0 = Fr (8 x u32 per element), 1 = Fp (12 x u32) -- field.hpp, the reference's blstrs__scalar__Scalar_* / blstrs__fp__Fp_*
 * (bls12-381.cu:499-593, 652-963); 2 = Fr through the 9 x 29-bit lazily reduced form of the big sumcheck rounds (fs29.hpp; op 5..7: the UNREDUCED
 * product summed as columns), 3 = Fp through the 14 x 28-bit form of the MSM pipeline (fq28.hpp; no mont / unmont).  op: 0 add, 1 sub, 2 mul, 3 sqr,
 * 4 double, 5 mont, 6 unmont, 7 neg.  b_d may be NULL for unary ops.  Results are canonical.  Nothing on the proving path calls this. */
int zk_field_selftest(zk_ctx* ctx, int which, int op, const void* a_d, const void* b_d, void* out_d, uint32_t n);

/* ---------------------------------------------------------------- activation commitments inside the proof
 * The reference's "commitment of activations / proof compression" future work (README.md:131-135).  With Z_i and A_i committed row-wise over
 * layer i's generators (zk_commit), the three activation-side claims of a hidden layer -- Z_i~(p1), p1 = u_out || u_bs (zkFC_i); Z_i~(p2), p2 = v_hp
 * (zkReLU_i); A_i~(p3), p3 = u_in || u_bs (zkFC_{i+1}) -- are reduced to ONE evaluation of the concatenated tensor T = [Z_i ; A_i] by an inner-product
 * sumcheck against the public table E = [eq(p1,.) + alpha eq(p2,.) ; alpha^2 eq(p3,.)] (k + 1 rounds, k = log2 n), and T~(r') is opened ONCE against
 * the concatenated commitment [com(Z_i) ; com(A_i)].  n = B_p * out_p entries per tensor (a power of two, a multiple of the generator count);
 * pts_h = p1 | p2 | p3 (k each), alpha_h one challenge, fold_h the k + 1 round challenges (= the opening point).  Appends
 *   Fr: [sumcheck 3 (k + 1) + 2][t(0)]   G1: [com(u_hi)][me_open 3 log2(n_gens) + 1]
 * to tr; the opening's MSMs join the transcript's deferred batch (com(u_hi) is computed as the commitment of the row-folded tensor: the prover
 * holds the tensor, and the generators' window tables exist).  One sumcheck and one opening per hidden layer instead of three openings. */
int zk_act_open(zk_ctx* ctx, zk_gens* gens, const zk_fr* z_d, const zk_fr* a_d, uint32_t n, const zk_fr* pts_h, const zk_fr* alpha_h, const zk_fr* fold_h,
                zk_transcript* tr);
/* Commitment::commit (the intended row MSM, as zk_commit) deferred into a transcript: appends the tn / n_gens points to tr's G1 list; the MSMs join the
 * transcript's batch (digit tables when present), so a sealed transcript computes them under other proofs' work.  Used for the per-input activation
 * commitments com(Z_i), com(A_i) (zkdl_amd/proof.py: prove_with_activation_commitments_stream).  max_bits: 0 (full width), or a bound on the bit length of
 * the centred values min(v, r - v): only the windows such scalars can touch are walked; the device checks the bound, and zk_ctx_sync returns ZK_ERR_RANGE if
 * a scalar was longer (zkReLU's inputs are < 2^47 in magnitude by zk_relu_forward's own range check, its outputs < 2^32). */
int zk_act_commit(zk_ctx* ctx, zk_gens* gens, const zk_fr* t_d, uint32_t tn, uint32_t max_bits, zk_transcript* tr);
/* The same with the reduction's challenges bound to the prover's messages (the sound mode of the layer proofs, zk_*_prove_fs): alpha =
 * zk_fs_derive(seed32, layer, 2, 0, 1) and every ROUND challenge -- hence the opening point r' -- is the hash chain of zk_fs_round_challenges with
 * kind = 2, which = 0 over the reduction's own round messages, computed on the device.  seed32 must absorb everything alpha may depend on: the
 * statement, the activation commitments, the three claimed evaluations and their points (zkdl_amd/proof.py: act_layer_seed).  Same transcript layout;
 * the verifier re-derives alpha and fold and calls zk_verify_act_open. */
int zk_act_open_fs(zk_ctx* ctx, zk_gens* gens, const zk_fr* z_d, const zk_fr* a_d, uint32_t n, const zk_fr* pts_h, const uint8_t* seed32, uint32_t layer,
                   zk_transcript* tr);
/* The verifier's side: com_d = the 2m points [com(Z_i) ; com(A_i)] (device), y_h = the three claimed evaluations as the layer proofs state them,
 * fr_h / g1_h / inf_h the slice zk_act_open appended (G1 affine-canonical).  Checks the sumcheck, E~(r') (public arithmetic), T~(r') == t(0), and the
 * opening with the checks of zk_verify_open. */
int zk_verify_act_open(zk_ctx* ctx, zk_gens* gens, const zk_g1j* com_d, uint32_t m2, uint32_t k, const zk_fr* pts_h, const zk_fr* alpha_h, const zk_fr* fold_h,
                       const zk_fr* y_h, const zk_fr* fr_h, size_t n_fr, const zk_g1a* g1_h, const uint8_t* inf_h, size_t n_g1, int* ok);

/* ---------------------------------------------------------------- round-by-round Fiat-Shamir ("sound mode")
 * zk_fc_prove / zk_relu_prove take every challenge as an argument, like the reference's pre-drawn coins (proof.cu:3-11): that is the
 * reference-parity interface, and with zk_fs_derive a reproducible public-coin schedule -- but a prover who knows round r's challenge
 * before sending round r's message is not bound by it.  The *_fs entry points derive the challenges themselves:
 *   evaluation points fixed before a sub-protocol starts (zkFC: u_bs, u_out; zkReLU: u_z, u_r, u_rec, u_hp) = zk_fs_derive(seed32, layer, kind, vec, .)
 *   with vec = the position in the reference's draw order (zkfc.cu:135-137, zkrelu.cu:85-98);
 *   every sumcheck ROUND challenge (zkFC: u_in; zkReLU: v_z, v_r, v_hp)  x_r = H(state_r),  state_{r+1} = SHA-256(state_r | c0 | c1 | c2) over the
 *   round's three coefficients as the transcript holds them (32 bytes each), state_0 = SHA-256("zkdl-amd/fs-rounds/v1" | seed32 | layer | kind |
 *   which), which = 0 (zkFC inner product), 1 / 2 / 3 (zkReLU magnitude bits, remainder bits, Hadamard); integers as 4 little-endian bytes;
 *   x_r = the digest read as 8 little-endian uint32 limbs, top limb mod 1944954707 (random_vec's distribution).
 * The hash runs on the device between a round's message and its fold (no host round trip); the opening's point u_out || u_in exists only after
 * the sumcheck, so it is proved behind it.  seed32: a digest of the public statement (zkdl_amd/proof.py: statement_digest).  The transcript
 * has the layout of zk_fc_prove / zk_relu_prove; zk_fs_round_challenges re-derives the round challenges from it (coeffs_h: the k coefficient
 * triples), after which zk_verify_fc / zk_verify_relu check the proof as before.  Fed the derived challenges, zk_fc_prove / zk_relu_prove
 * (and the CPU oracle) produce the same transcript bit for bit.  Coefficients are hashed as stored, so only canonical encodings (< r) are accepted:
 * zk_fs_round_challenges returns ZK_ERR_RANGE otherwise, and every zk_verify_* rejects a proof holding a field element >= r. */
int zk_fc_prove_fs(zk_ctx* ctx, zk_fc* fc, const zk_fr* x_d, uint32_t xn, const zk_fr* z_d, uint32_t zn, zk_gens* gens, const uint8_t* seed32, uint32_t layer,
                   zk_transcript* tr);
int zk_relu_prove_fs(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x_d, const zk_fr* z_d, uint32_t n, const uint8_t* seed32, uint32_t layer, zk_transcript* tr);
/* zk_leaky_prove (above) in this mode: zkReLU's round-by-round transcript, then the extra Hadamard sumcheck with u_lk = zk_fs_derive(seed32, layer, 1, 7, .)
 * and v_lk the hash chain which = 4. */
int zk_leaky_prove_fs(zk_ctx* ctx, const zk_relu* relu, const zk_fr* x_d, const zk_fr* z_d, uint32_t n, const uint8_t* seed32, uint32_t layer, zk_transcript* tr);
int zk_fs_round_challenges(const uint8_t* seed32, uint32_t layer, uint32_t kind, uint32_t which, const zk_fr* coeffs_h, uint32_t k, zk_fr* out_h);

/* ---------------------------------------------------------------- multi-GPU: the run aggregate, natively over RCCL / xGMI
 * One process per GPU, independent proof batches per rank, no data-path collective (north_star); the only exchange is ONE all-reduce of the
 * per-rank aggregate -- the sum of the digests (8 zero-extended 64-bit lanes each, zk_transcript_seal / zk_transcript_digest_dev) of the rank's
 * proofs.  These entry points let a C / C++ host do that without torch.distributed (bench.py uses torch.distributed's RCCL backend for the same
 * all-reduce).  librccl is dlopen'ed by zk_dist_unique_id / zk_dist_init; nothing else in the library needs it.  The reference has no multi-GPU path.
 *   rank 0: zk_dist_unique_id(id) and hand the 128 bytes to the other ranks (file, environment, MPI ...); every rank: zk_dist_init(ctx, rank, world, id, &d);
 *   per run: zk_dist_sum_rows(ctx, rows_d, n_proofs, agg_d); zk_dist_allreduce_u64(d, agg_d, 8); copy agg to the host; zk_dist_lanes_to_fr. */
typedef struct zk_dist zk_dist;
int zk_dist_unique_id(uint8_t* id128);
int zk_dist_init(zk_ctx* ctx, int rank, int world, const uint8_t* id128, zk_dist** out);
int zk_dist_world_size(const zk_dist* d);                                   /* as RCCL reports it */
int zk_dist_sum_rows(zk_ctx* ctx, const uint64_t* rows_d, size_t n, uint64_t* out8_d);   /* waits for every sealed proof first */
int zk_dist_allreduce_u64(zk_dist* d, uint64_t* lanes_d, size_t n);           /* in place, ncclSum on ncclUint64; returns when done */
int zk_dist_lanes_to_fr(const uint64_t* lanes8_h, zk_fr* out_h);              /* sum_j lane_j 2^(32 j) mod r, raw limbs */
int zk_dist_destroy(zk_dist* d);

#ifdef __cplusplus
}
#endif
#endif
