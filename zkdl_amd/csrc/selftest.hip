// zkdl_amd/csrc/selftest.hip -- zk_field_selftest: the device field functions the kernels are built from, one element per lane, so that the golden
// vectors taken from the reference's own compiled arithmetic (tests/golden/prims.json: blstrs__scalar__Scalar_* bls12-381.cu:499-593 and
// blstrs__fp__Fp_* bls12-381.cu:652-963) reach EVERY one of them on the GPU directly -- the Fp functions otherwise only through the group law, f_double /
// f_sqr only through a + a / a * a, and the lazily reduced 29- / 28-bit-limb forms (fs29.hpp, fq28.hpp) only through whole sumchecks and MSMs.
// A diagnostic entry point: nothing on the proving path calls it.
#include "ctx.hpp"
#include "dev_util.hpp"
#include "fq28.hpp"
#include "fs29.hpp"

namespace zk {

enum { ST_ADD = 0, ST_SUB = 1, ST_MUL = 2, ST_SQR = 3, ST_DOUBLE = 4, ST_MONT = 5, ST_UNMONT = 6, ST_NEG = 7 };

template <class F>
__device__ __forceinline__ El<F> st_plain(int op, const El<F>& x, const El<F>& y) {
  switch (op) {
    case ST_ADD: return f_add<F>(x, y);
    case ST_SUB: return f_sub<F>(x, y);
    case ST_MUL: return f_mul<F>(x, y);
    case ST_SQR: return f_sqr<F>(x);
    case ST_DOUBLE: return f_double<F>(x);
    case ST_MONT: return f_mont<F>(x);
    case ST_UNMONT: return f_unmont<F>(x);
    default: return f_neg<F>(x);
  }
}
// which = 0: Fr, field.hpp   1: Fp, field.hpp   2: Fr through the 9 x 29-bit form   3: Fp through the 14 x 28-bit form (n-form and back)
__global__ void __launch_bounds__(64) k_field_selftest(int which, int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (which == 0 || which == 2) {
    Fr x, y;
    for (int j = 0; j < 8; j++) {
      x.v[j] = a[8 * (size_t)i + j];
      y.v[j] = b ? b[8 * (size_t)i + j] : 0u;
    }
    Fr r;
    if (which == 0) {
      r = st_plain<FrP>(op, x, y);
    } else {
      const Fs X = fs_unpack(x), Y = fs_unpack(y);
      Fs t;
      switch (op) {
        case ST_ADD: t = fs_norm(fs_add(X, Y)); break;                 // < 2
        case ST_SUB: t = fs_norm(fs_sub<2>(X, Y)); break;              // < 3
        case ST_MUL: t = fs_mul(X, Y); break;                          // < 1.5
        case ST_SQR: t = fs_mul(X, X); break;
        case ST_DOUBLE: t = fs_norm(fs_add(X, X)); break;
        default: {                                                     // the unreduced product, summed as columns: fs_mul_raw_acc + fs_cols_to_fr
          uint32_t acc[18];
          for (int k = 0; k < 18; k++) acc[k] = 0;
          fs_mul_raw_acc(acc, X, Y);
          fs_cols_norm(acc);
          uint64_t cols[18];
          for (int k = 0; k < 18; k++) cols[k] = acc[k];
          r = fs_cols_to_fr(cols);
          for (int j = 0; j < 8; j++) out[8 * (size_t)i + j] = r.v[j];
          return;
        }
      }
      r = fs_to_fr(t);
    }
    for (int j = 0; j < 8; j++) out[8 * (size_t)i + j] = r.v[j];
  } else {
    Fp x, y;
    for (int j = 0; j < 12; j++) {
      x.v[j] = a[12 * (size_t)i + j];
      y.v[j] = b ? b[12 * (size_t)i + j] : 0u;
    }
    Fp r;
    if (which == 1) {
      r = st_plain<FpP>(op, x, y);
    } else {
      const Fq X = fq_from_fp(x), Y = fq_from_fp(y);                   // normalised, < 2p
      Fq t;
      switch (op) {
        case ST_ADD: t = fq_add(X, Y); break;
        case ST_SUB: t = fq_sub<3>(X, Y); break;
        case ST_MUL: t = fq_mul(X, Y); break;
        case ST_SQR: t = fq_sqr(X); break;
        case ST_DOUBLE: t = fq_add(X, X); break;
        default: t = fq_sub<3>(fq_zero(), X); break;                   // ST_NEG
      }
      r = fq_to_fp(fq_norm(t));
    }
    for (int j = 0; j < 12; j++) out[12 * (size_t)i + j] = r.v[j];
  }
}

}  // namespace zk

extern "C" int zk_field_selftest(zk_ctx* ctx, int which, int op, const void* a_d, const void* b_d, void* out_d, uint32_t n) {
  if (!ctx) return ZK_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  if (which < 0 || which > 3 || op < 0 || op > 7) return ctx->fail(ZK_ERR_ARG, "zk_field_selftest: which 0..3, op 0..7");
  const bool binary = op <= zk::ST_MUL || (which == 2 && op >= zk::ST_MONT);
  if ((which >= 2 && (op == zk::ST_MONT || op == zk::ST_UNMONT)) && which == 3) return ctx->fail(ZK_ERR_ARG, "zk_field_selftest: no mont / unmont in the 28-bit form");
  if (!a_d || !out_d || (binary && !b_d)) return ctx->fail(ZK_ERR_ARG, "zk_field_selftest: null pointer");
  if (n == 0) return ZK_OK;
  int rc = ctx->settle();
  if (rc != ZK_OK) return rc;
  hipLaunchKernelGGL(zk::k_field_selftest, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, which, op, static_cast<const uint32_t*>(a_d), static_cast<const uint32_t*>(b_d),
                     static_cast<uint32_t*>(out_d), n);
  if (hipGetLastError() != hipSuccess) return ctx->fail(ZK_ERR_DEVICE, "zk_field_selftest: launch failed");
  return ZK_OK;
}
