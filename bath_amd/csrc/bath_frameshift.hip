// placeholder until the frameshift kernels land (replaced later in this round)
#include "bath_common.hpp"
extern "C" int bath_hip_fsprofile_convert(bath_hip_ctx *ctx, const bath_fs_profile *, bath_hip_fsprofile **) { ctx->set_error("fs not built yet"); return BATH_EINVAL; }
extern "C" void bath_hip_fsprofile_destroy(bath_hip_fsprofile *) {}
extern "C" int bath_hip_fs3_forward_parser(bath_hip_ctx *ctx, const bath_hip_fsprofile *, const bath_hip_seqs *, int, float *, float *, const int64_t *) { ctx->set_error("fs not built yet"); return BATH_EINVAL; }
extern "C" int bath_hip_fs3_backward_parser(bath_hip_ctx *ctx, const bath_hip_fsprofile *, const bath_hip_seqs *, int, float *, float *, const int64_t *) { ctx->set_error("fs not built yet"); return BATH_EINVAL; }
extern "C" int bath_hip_fs5_envelopes(bath_hip_ctx *ctx, const bath_hip_fsprofile *, const bath_hip_seqs *, int, int, bath_fs5_result *, float *, const int64_t *, float *, const int64_t *) { ctx->set_error("fs not built yet"); return BATH_EINVAL; }
