// Behaviour-intent encoder (EncoderRNN, nova/behavior_net.py:6-22) for rollout inference.
//
// One wave = 16 (env, entity) rows of one agent-net; 4 waves per workgroup share the LDS-staged
// weights.  The whole chain Linear+ReLU -> 10 GRU steps -> Linear -> softmax -> soft update runs in
// registers in the D layout; HBM sees the window once (L*d floats per row), h0/hL and the latent.
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int ER = 32;         // encoder_rnn_dim
constexpr int ELD = ER + 4;    // padded LDS leading dimension

__global__ __launch_bounds__(256) void enc_fwd_kernel(IplanEncFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_lin[ER * 20];
    __shared__ __attribute__((aligned(16))) float s_wih[3 * ER * ELD];
    __shared__ __attribute__((aligned(16))) float s_whh[3 * ER * ELD];
    __shared__ __attribute__((aligned(16))) float s_out[16 * ELD];
    __shared__ __attribute__((aligned(16))) float s_blin[ER], s_bih[3 * ER], s_bhh[3 * ER], s_bout[16];

    const int net = (int)blockIdx.y;
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    stage_matrix(s_lin, 20, ER, P + a.off[IPLAN_ENC_LIN_W], ER, a.d);
    stage_matrix(s_wih, ELD, 3 * ER, P + a.off[IPLAN_ENC_WIH], 3 * ER, ER);
    stage_matrix(s_whh, ELD, 3 * ER, P + a.off[IPLAN_ENC_WHH], 3 * ER, ER);
    stage_matrix(s_out, ELD, 16, P + a.off[IPLAN_ENC_OUT_W], a.Z, ER);
    stage_vector(s_blin, ER, P + a.off[IPLAN_ENC_LIN_B], ER);
    stage_vector(s_bih, 3 * ER, P + a.off[IPLAN_ENC_BIH], 3 * ER);
    stage_vector(s_bhh, 3 * ER, P + a.off[IPLAN_ENC_BHH], 3 * ER);
    stage_vector(s_bout, 16, P + a.off[IPLAN_ENC_OUT_B], a.Z);
    __syncthreads();

    const int l = lane_id(), g = l >> 4;
    const int rows = a.B * a.N;
    const int row = ((int)blockIdx.x * 4 + wave_id()) * 16 + (l & 15);
    const bool valid = row < rows;
    const int b = valid ? row / a.N : 0, i = valid ? row % a.N : 0;
    const int64_t xs_i = a.x_s_i ? a.x_s_i : (int64_t)a.L * a.d, xs_t = a.x_s_t ? a.x_s_t : (int64_t)a.d;
    const float* xrow = a.x + (int64_t)net * a.x_s_net + (int64_t)b * a.x_s_b + (int64_t)i * xs_i;
    f32x4 h[2];
    {
        const float* hrow = a.h0 + (int64_t)net * a.h0_s_net + (int64_t)b * a.h0_s_b + (int64_t)i * ER;
        h[0] = vload(hrow, valid, ER, 0);
        h[1] = vload(hrow, valid, ER, 1);
    }
    for (int t = 0; t < a.L; ++t) {
        f32x4 x[1];
        x[0] = vload(xrow + t * xs_t, valid, a.d, 0);
        f32x4 u[2];
        u[0] = relu4(dense_tile<1>(s_lin, 20, 0, x, bfrag_lds(s_blin, 0)));
        u[1] = relu4(dense_tile<1>(s_lin, 20, 16, x, bfrag_lds(s_blin, 1)));
        gru_step_lds<2, 2>(s_wih, ELD, s_whh, ELD, s_bih, s_bhh, u, h, nullptr);
    }
    {
        float* hrow = a.hL + (int64_t)net * a.hL_s_net + (int64_t)b * a.hL_s_b + (int64_t)i * ER;
        vstore(hrow, valid, ER, 0, h[0]);
        vstore(hrow, valid, ER, 1, h[1]);
    }
    // latent = softmax(W_out h + b) over the Z real entries of the single output tile
    f32x4 lg = dense_tile<2>(s_out, ELD, 0, h, bfrag_lds(s_bout, 0));
    float m = -INFINITY;
    for (int q = 0; q < 4; ++q)
        if (4 * g + q < a.Z) m = fmaxf(m, lg[q]);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    f32x4 e;
    float ssum = 0.f;
    for (int q = 0; q < 4; ++q) {
        e[q] = (4 * g + q < a.Z) ? expf(lg[q] - m) : 0.f;
        ssum += e[q];
    }
    ssum = group_sum(ssum);
    f32x4 lat;
    for (int q = 0; q < 4; ++q) lat[q] = e[q] / ssum;
    if (a.prev_latent) {
        const float* prow = a.prev_latent + (int64_t)net * a.pl_s_net + (int64_t)b * a.pl_s_b + (int64_t)i * a.Z;
        const f32x4 pv = vload(prow, valid, a.Z, 0);
        for (int q = 0; q < 4; ++q) lat[q] = a.one_minus_c * pv[q] + lat[q] * a.c;   // stable_behavior_policy.py:118
    }
    float* lrow = a.latent_out + (int64_t)net * a.lo_s_net + (int64_t)b * a.lo_s_b + (int64_t)i * a.Z;
    vstore(lrow, valid, a.Z, 0, lat);
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
