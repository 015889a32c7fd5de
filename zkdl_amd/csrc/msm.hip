// zkdl_amd/csrc/msm.hip -- bucket-method MSM back end (placeholder until the Pippenger kernels land).
#include "ctx.hpp"

namespace zk {
int commit_pippenger_dev(zk_ctx* ctx, const G1J*, uint32_t, const Fr*, uint32_t, G1J*) { return ctx->fail(ZK_ERR_ARG, "ZK_MSM_PIPPENGER back end not built"); }
int me_open_flat_dev(zk_ctx* ctx, const Fr*, const G1J*, uint32_t, const Fr*, uint32_t, G1J*, Fr*) { return ctx->fail(ZK_ERR_ARG, "ZK_MSM_PIPPENGER back end not built"); }
int g1_me_msm_dev(zk_ctx* ctx, const G1J*, uint32_t, const Fr*, uint32_t, G1J*) { return ctx->fail(ZK_ERR_ARG, "ZK_MSM_PIPPENGER back end not built"); }
}  // namespace zk
