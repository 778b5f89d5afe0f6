// Behaviour-intent encoder (EncoderRNN, nova/behavior_net.py:6-22) for rollout inference.
//
// One wave = 16 (env, entity) rows of one agent-net; 4 waves per workgroup share the LDS-staged
// weights.  The whole chain Linear+ReLU -> 10 GRU steps -> Linear -> softmax -> soft update runs in
// registers in the D layout; HBM sees the window once (L*d floats per row), h0/hL and the latent.
#include "api_util.h"
#include "gru_tile.h"
#include "enc_body.h"

namespace iplan {

__global__ __launch_bounds__(256) void enc_fwd_kernel(IplanEncFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_enc[ENC_LDS_FLOATS];
    enc_fwd_block(a, (int)blockIdx.y, (int)blockIdx.x * 4 + wave_id(), s_enc);
}

}  // namespace iplan

extern "C" int iplan_enc_fwd(const IplanEncFwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a) return fail(IPLAN_EINVAL, "iplan_enc_fwd: null args");
    if (a->d < 1 || a->d > 16 || a->Z < 1 || a->Z > 16 || a->L < 1 || a->n_nets < 1 || a->B < 1 || a->N < 1)
        return fail(IPLAN_EINVAL, "iplan_enc_fwd: unsupported dims d=%d Z=%d L=%d", a->d, a->Z, a->L);
    if (!a->x || !a->h0 || !a->hL || !a->latent_out || !a->params)
        return fail(IPLAN_EINVAL, "iplan_enc_fwd: null tensor pointer");
    const int rows = a->B * a->N;
    hipLaunchKernelGGL(enc_fwd_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)a->n_nets), dim3(256), 0,
                       (hipStream_t)stream, *a);
    return check_launch("iplan_enc_fwd");
}
