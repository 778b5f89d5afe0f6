// Rollout encoder body (EncoderRNN, nova/behavior_net.py:6-22 + the soft update), shared by enc_fwd_kernel (behavior.hip) and the
// fused GAT + encoder launch of the rollout (gat.hip): one wave = 16 (env, entity) rows of one agent-net, the workgroup's waves
// share the LDS-staged weights.
#pragma once
#include "gru_tile.h"

namespace iplan {

constexpr int ER = 32;         // encoder_rnn_dim
constexpr int ELD = ER + 4;    // padded LDS leading dimension
constexpr int ENC_LDS_FLOATS = ER * 20 + 2 * 3 * ER * ELD + 16 * ELD + ER + 2 * 3 * ER + 16;     // 33.6 KB

// `lds`: >= ENC_LDS_FLOATS floats, 16-byte aligned; `tile` = the wave's 16-row tile index; all threads of the block call it.
// COH: the new latent is stored device-coherently (read by other workgroups of the same launch: gat.hip, gat_enc_ac_fwd_kernel)
template <bool COH = false>
__device__ __forceinline__ void enc_fwd_block(const IplanEncFwdArgs& a, int net, int tile, float* __restrict__ lds) {
    float* s_lin = lds;
    float* s_wih = s_lin + ER * 20;
    float* s_whh = s_wih + 3 * ER * ELD;
    float* s_out = s_whh + 3 * ER * ELD;
    float* s_blin = s_out + 16 * ELD;
    float* s_bih = s_blin + ER;
    float* s_bhh = s_bih + 3 * ER;
    float* s_bout = s_bhh + 3 * ER;
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
    const int row = tile * 16 + (l & 15);
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
    vstore_c<COH>(lrow, valid, a.Z, 0, lat);
}

}  // namespace iplan
