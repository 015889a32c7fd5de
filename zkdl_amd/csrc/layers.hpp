// zkdl_amd/csrc/layers.hpp -- opaque objects behind the C ABI (zk_fc, zk_relu, zk_transcript).
#pragma once
#include "ctx.hpp"

namespace zk {
struct G1Table {  // affine window table: pts[w*n + j] = 2^(c*w) * base_j ; inf[...] marks infinite bases
  uint32_t n = 0, c = 0, W = 0;
  G1A* pts = nullptr;
  uint8_t* inf = nullptr;
};
}  // namespace zk

namespace zk {
struct alignas(128) G1AE {  // one table entry: 96 bytes of affine point in a 128-byte slot, so that a (random) read touches exactly one 128-byte line
  G1A p;
  uint32_t pad[8];
};
struct G1FullTable {  // every signed digit of every window precomputed: pts[((j * W + w) << (c - 1)) + (d - 1)] = d * 2^(c w) * base_j, d = 1 .. 2^(c-1)
  uint32_t n = 0, c = 0, W = 0;
  G1AE* pts = nullptr;      // affine, the bucket pipeline's Montgomery domain (fq28.hpp); 128 B x n x W x 2^(c-1)
  const uint8_t* inf = nullptr;  // [n] base-at-infinity flags (the window table's)
};
struct MsmJobHost {  // one pending MSM: *out = sum_j mags[j] * base_j over table T (W windows in use)
  const G1Table* T;
  const Fr* mags;
  G1J* out;
  uint32_t W;
  const G1FullTable* F = nullptr;  // not null: the job runs on the bucket-free path (msm.hip: k_msmf_*)
  uint32_t Wuse = 0;                  // bucket-free path: windows to walk (0 = all W): scalars with a stated bound (zk_act_commit)
  uint8_t pat_r = 0xff, pat_par = 0;  // scalars known to vanish outside { i : ((i >> pat_r) & 1) == pat_par } (0xff: dense) -- T0_r / T1_r of the flattened opening
};
struct DeriveJobHost {  // T_1 .. T_{k-1} of one me_open, derived from T_0 and the T0_j / T1_j behind the MSM batch (msm.hip)
  G1J* proof;            // the me_open's 3k+1 transcript slots
  const Fr* coef;        // k^2 raw scalars (k_mo_derive_coef)
  uint32_t k;
  const G1J* base;       // com(u_hi) when T_0 is derived too (Commitment::open), else nullptr (T_0 is an MSM)
};
struct MsmQueue {  // MSMs enqueued by the provers, executed together by msm_flush()
  std::vector<MsmJobHost> jobs;
  std::vector<DeriveJobHost> derives;
  std::vector<DBuf<Fr>> keep;  // prepared scalars, alive until the flush
  uint32_t c = 0;
  // a job's scalars were prepared by a recorded chain that is NOT marked early (zk_fc_prove_fs: the opening sits behind the whole hashed
  // sumcheck): a sealed proof's G1 phase must then wait for the whole Fr phase, not only for the early chains' events
  bool late_inputs = false;
};
}  // namespace zk

struct zk_gens {  // Commitment (commitment.cuh:8-23): generator tensor + lazily built MSM tables
  zk_ctx* ctx = nullptr;
  uint32_t n = 0;
  zk::G1J* pts = nullptr;
  std::map<uint32_t, zk::G1Table> tables;  // keyed by window width c
  zk::G1FullTable full;                    // optional (zk_gens_prepare_full): all digit multiples, for full-width scalars
};

struct zk_fc {  // zkFC, zkfc.cuh:16-33: padded weights (Montgomery, [in][out]) + their commitment
  zk_ctx* ctx = nullptr;
  uint32_t in = 0, out = 0, m = 0, n_gens = 0;
  zk::Fr* weights = nullptr;
  int16_t* wq = nullptr;   // the same weights as int16 when every |w| < 2^15 (quantised layers), else nullptr
  zk_gens* com = nullptr;  // the m commitment points, wrapped so com(u) can run as a table MSM
  uint32_t serial = 0;     // lane hint
};

struct zk_relu {  // zkReLU, zkrelu.cuh:10-20: aux tensors of the last forward pass
  zk_ctx* ctx = nullptr;
  uint32_t n = 0, cap = 0, serial = 0, bins_cap = 0;
  bool bins_valid = false;  // mag_bin / rem_bin hold the last forward pass's bits
  zk::Fr *sign = nullptr, *mag_bin = nullptr, *rem_bin = nullptr;
  uint32_t* mag_packed = nullptr;  // the same bits, 32 per element
  uint16_t* rem_packed = nullptr;  // 15 remainder bits + sign bit
  uint8_t* sign_packed = nullptr;
  uint32_t* oor = nullptr;
  // activation kind (extension; the reference's only activation is ReLU): 0 = ReLU, 1 = LeakyReLU with negative slope 2^-act_shift
  // (act_shift 0: identity, a signed requantisation).  low = mont(mag_rescaled >> act_shift), nsign = 1 - sign: written by the forward
  // pass of a leaky layer, consumed by its extra Hadamard sumcheck
  uint32_t act_kind = 0, act_shift = 0;
  zk::Fr *low = nullptr, *nsign = nullptr;
  void free_all() {
    zk::dev_free(ctx, sign); zk::dev_free(ctx, mag_bin); zk::dev_free(ctx, rem_bin);
    zk::dev_free(ctx, mag_packed); zk::dev_free(ctx, rem_packed); zk::dev_free(ctx, sign_packed); zk::dev_free(ctx, oor);
    zk::dev_free(ctx, low); zk::dev_free(ctx, nsign);
    low = nsign = nullptr;
    sign = mag_bin = rem_bin = nullptr;
    mag_packed = nullptr; rem_packed = nullptr; sign_packed = nullptr; oor = nullptr;
    cap = n = bins_cap = 0;
    bins_valid = false;
  }
};

struct zk_transcript {  // proof values in the order of SURVEY.md App. A.6, resident on the device
  zk_ctx* ctx = nullptr;
  zk::Fr* fr_d = nullptr;
  zk::G1J* g1_d = nullptr;
  size_t n_fr = 0, n_g1 = 0, cap_fr = 0, cap_g1 = 0;
  zk::MsmQueue queue;      // deferred MSMs whose results land in g1_d
  zk::Recorder rec;        // recorded (not yet executed) Fr work of the prove calls since the last flush
  size_t n_g1_final = 0;   // G1 entries already flushed + canonicalised
  int ensure(size_t add_fr, size_t add_g1);
  std::vector<zk::DBuf<zk::Fr>> deferred_keep;  // prepared scalars of a sealed batch, released when it is unsealed
  bool sealed = false;              // the G1 phase is (possibly still) running on the G1 stream
  hipEvent_t sealed_ev = nullptr;   // recorded behind it
  hipStream_t sealed_stream = nullptr;  // where the last seal ran (nullptr: main stream)
  bool g1_deferred = false;             // sealed, Fr phase submitted, G1 phase waiting in ctx->g1_pending for partners ("g1_batch")
  uint64_t* pending_out8 = nullptr;     // where that G1 phase leaves the digest lanes
  std::vector<hipEvent_t> pending_early;  // events behind the openings' scalar preparation (the G1 phase may start behind them)
  bool pending_early_ok = false;        // this proof's MSM inputs are complete behind the batch's early events (its own or a merged partner's)
  int submit_g1(uint64_t* out8_d, const std::vector<hipEvent_t>* early, bool early_ok);  // run the G1 phase now or leave it with ctx->g1_pending
  int flush();                      // run pending MSMs on the main stream, canonicalise their outputs
  int seal(uint64_t* out8_d);       // the same plus the digest, on the G1 stream, with no main-stream wait
  int seal_pair(zk_transcript* other, uint64_t* out8_d, uint64_t* other_out8_d);  // two proofs: ONE zipped Fr phase, then each one's G1 phase
  int g1_phase(uint64_t* out8_d, const std::vector<hipEvent_t>* early = nullptr);   // the part of seal() after the Fr phase
  void unseal();                    // main stream waits for a sealed G1 phase; its scalars are released
};

namespace zk {
int transcript_digest_dev(zk_ctx* ctx, const Fr* fr, size_t n_fr, const G1J* g1, size_t n_g1, Fr* out_fr_d, uint64_t* out8_d);
int binp_sumcheck_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d);
int binp_partial_me_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, uint32_t k, Fr* out_d);
int binb_sumcheck_pme_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, const Fr* v_d, uint32_t k, Fr* proof_d, const Fr* u_rec_d, Fr* pme_d);
// sound mode: the in-word rounds (5 for 32-bit words, 4 for 16-bit words) of the bit sumcheck from the packed words, hashing between the rounds
// (state_d: FsState*); t receives the table of the next round, one entry per word
int fsb_sumcheck_head_dev(zk_ctx* ctx, const void* packed, uint32_t n, uint32_t nb, const Fr* u_d, uint32_t k, void* state_d, Fr* fold_d, Fr* proof_d, const Fr* u_rec_d, Fr* pme_d,
                          DBuf<Fr>& t, uint32_t* rounds_done);
// sound mode, inner-product form for callers outside fs_rounds.hip (act_open.hip): fold_state_d[0..k) RECEIVES the round challenges, slot [k]
// holds the running hash state (fs_state0_slot writes its initial value on the host, before the upload)
int sumcheck_ip_fs_dev(zk_ctx* ctx, const Fr* a, const Fr* b, uint32_t n, Fr* fold_state_d, uint32_t k, Fr* proof_d);
void fs_state0_slot(const uint8_t seed32[32], uint32_t layer, uint32_t kind, uint32_t which, zk_fr* slot);
int fr_vec_q16_dev(zk_ctx* ctx, const Fr* x, uint32_t M, const int16_t* q, uint32_t W, Fr* out);
int float_to_fr_dev(zk_ctx* ctx, const float* fs, Fr* out, uint32_t rows, uint32_t rows_p, uint32_t cols, uint32_t cols_p);
int gens_create(zk_ctx* ctx, const G1J* gens_d, uint32_t n, bool copy, zk_gens** out);
void gens_destroy(zk_gens* g);
int gens_table(zk_ctx* ctx, zk_gens* g, uint32_t c, uint32_t W, const G1Table** out);
int msm_batch_dev(zk_ctx* ctx, zk_gens* g, const Fr* scalars, uint32_t q, bool do_unmont, bool probe_bits, G1J* out);
// pattern (optional, q entries): (r << 1) | parity per row, 0x1fe = dense -- see MsmJobHost::pat_r
// bits_bound (optional): the caller's bound on the scalars' centred bit length; fewer windows are walked, and the device raises the context's range flag if a scalar is longer
int msm_enqueue(zk_ctx* ctx, zk_gens* g, const Fr* scalars, uint32_t q, bool do_unmont, bool probe_bits, G1J* out, MsmQueue* queue, const uint32_t* out_idx = nullptr,
                const uint16_t* pattern = nullptr, uint32_t bits_bound = 0);
int gens_prepare_full(zk_ctx* ctx, zk_gens* g, uint32_t c);
int msm_run_derives(zk_ctx* ctx, const std::vector<DeriveJobHost>& derives);
int msm_flush(zk_ctx* ctx, MsmQueue* queue);
int msm_run_jobs(zk_ctx* ctx, const std::vector<MsmJobHost>& hjobs, uint32_t c);
int g1_canonicalise_dev(zk_ctx* ctx, G1J* pts, uint32_t n);
struct CanonRanges {  // up to 8 ranges of points put into canonical form by ONE launch
  G1J* p[8];
  uint32_t n[8];
  uint32_t cnt;
};
int g1_canonicalise_multi_dev(zk_ctx* ctx, const CanonRanges& r);
int commit_dev(zk_ctx* ctx, zk_gens* gens, const Fr* t, uint32_t tn, G1J* out);
// queue == nullptr: run now; otherwise (PIPPENGER back end) the MSMs are deferred to msm_flush(queue)
int me_open_dev(zk_ctx* ctx, zk_gens* gens, const Fr* t, const Fr* u_d, uint32_t k, G1J* proof_d, Fr* ret_d, MsmQueue* queue, const G1J* com_eval_d = nullptr);
int open_dev(zk_ctx* ctx, zk_gens* gens, const Fr* t, uint32_t tn, zk_gens* com, const Fr* u_d, uint32_t k, G1J* com_eval_d, G1J* proof_d, Fr* ret_d, MsmQueue* queue,
             const int16_t* tq = nullptr);  // tq: optional int16 copy of t (quantised weights)
int fc_create(zk_ctx* ctx, uint32_t in, uint32_t out, const Fr* weights_d, zk_gens* gens, zk_fc** out_fc);
int fc_forward(zk_ctx* ctx, const zk_fc* fc, const Fr* x, uint32_t xn, Fr* z);
int fc_prove(zk_ctx* ctx, zk_fc* fc, const Fr* x, uint32_t xn, const Fr* z, uint32_t zn, zk_gens* gens,
             const zk_fr* u_bs_h, const zk_fr* u_in_h, const zk_fr* u_out_h, zk_transcript* tr);
int relu_forward(zk_ctx* ctx, zk_relu* relu, const Fr* x, uint32_t n, Fr* z);
int relu_materialise_bins(zk_ctx* ctx, zk_relu* relu);
int leaky_prove(zk_ctx* ctx, const zk_relu* relu, const Fr* x, const Fr* z, uint32_t n, const zk_fr* const ch_h[9], zk_transcript* tr);
int relu_prove(zk_ctx* ctx, const zk_relu* relu, const Fr* x, const Fr* z, uint32_t n, const zk_fr* u_z_h, const zk_fr* v_z_h, const zk_fr* u_r_h,
               const zk_fr* v_r_h, const zk_fr* u_rec_h, const zk_fr* u_hp_h, const zk_fr* v_hp_h, zk_transcript* tr);
// g1_kernels.hip
int g1_fill_dev(zk_ctx* ctx, G1J* out, uint32_t n, const G1J& g);
int g1a_fill_dev(zk_ctx* ctx, G1A* out, uint32_t n, const G1A& g);
int g1_from_affine_dev(zk_ctx* ctx, const G1A* a, G1J* out, uint32_t n);
int g1_neg_dev(zk_ctx* ctx, const G1J* a, G1J* out, uint32_t n);
int g1a_neg_dev(zk_ctx* ctx, const G1A* a, G1A* out, uint32_t n);
int g1_addsub_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1J* b, G1J* out, uint32_t n);
int g1_maddsub_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1A* b, G1J* out, uint32_t n);
int g1_addsub_bcast_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1J& x, G1J* out, uint32_t n);
int g1_maddsub_bcast_dev(zk_ctx* ctx, bool sub, const G1J* a, const G1A& x, G1J* out, uint32_t n);
int g1_mul_dev(zk_ctx* ctx, const G1J* g, uint32_t n_g, const Fr* s, uint32_t total, G1J* out);
}  // namespace zk
