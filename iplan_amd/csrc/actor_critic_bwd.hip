// Backward of the recurrent actor / critic (R_Actor.evaluate_actions / R_Critic.forward under
// loss.backward(), learners/ippo_learner.py:202,216) for all agents and both nets in one launch
// sequence:
//   1. ac_bwd_tail_kernel   one wave per 16-row tile, everything in the D layout of wave_tile.h:
//        head^T -> LN3' -> GRU step' -> W_ih^T -> LN2' -> ReLU' -> fc2^T -> LN1' -> ReLU'
//      emits the row-level pre-activation gradients (dz1, dz2, GRU gates, head) and per-tile
//      LayerNorm-parameter partial sums.  Weight gradients of the 64-wide layers are then plain
//      dY^T X contractions (wgrad.hip).
//   2. ac_fc1_wgrad_kernel  the one big contraction G[m][c] = sum_r dz1[r][m] * xhat[r][c]
//      (M = 64, F = 2485 at Highway chaotic, 22 950 rows per agent) on MFMA, with the normalised
//      feature row xhat gathered straight from the episode-buffer fields exactly like the forward.
//   3. ac_fc1_finalize_kernel  uses LN(F)'s affine structure so that ONE contraction serves three
//      gradients:  dW1 = gamma*G + beta*S,  dgamma = sum_m W1*G,  dbeta = sum_m W1*S  (S = db1).
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int BM = IPLAN_AC_HIDDEN;    // 64
constexpr int BT = BM / 16;            // 4 tiles

template <int KT>
__device__ __forceinline__ f32x4 dense_tile_t(const float* __restrict__ W, int ld, int rows, int cols, int o0,
                                              const f32x4 (&x)[KT], f32x4 acc) {
    // acc += (W^T x)[o0 .. o0+15],  W row-major [rows x cols], x has `rows` entries (KT tiles)
    for (int T = 0; T < KT; ++T) acc = mma_block(wfrag_t(W, ld, rows, cols, o0, 16 * T), x[T], acc);
    return acc;
}

// LayerNorm backward on a 64-wide per-chain vector.  dy -> dx (in place); dgam/dbet accumulate.
__device__ __forceinline__ void ln_bwd_tiles(f32x4 (&dy)[BT], const f32x4 (&xhat)[BT], const float* __restrict__ gamma,
                                             float rstd, f32x4 (&dgam)[BT], f32x4 (&dbet)[BT]) {
    float s1 = 0.f, s2 = 0.f;
    f32x4 dxh[BT];
    for (int t = 0; t < BT; ++t) {
        const f32x4 gm = bfrag(gamma, BM, t);
        for (int q = 0; q < 4; ++q) {
            dgam[t][q] = fmaf(dy[t][q], xhat[t][q], dgam[t][q]);
            dbet[t][q] += dy[t][q];
            dxh[t][q] = dy[t][q] * gm[q];
            s1 += dxh[t][q];
            s2 = fmaf(dxh[t][q], xhat[t][q], s2);
        }
    }
    const float m1 = group_sum(s1) * (1.0f / BM), m2 = group_sum(s2) * (1.0f / BM);
    for (int t = 0; t < BT; ++t)
        for (int q = 0; q < 4; ++q) dy[t][q] = rstd * (dxh[t][q] - m1 - xhat[t][q] * m2);
}

// sum over the 16 chains of a wave tile (lanes n = l & 15); every lane ends with the total
__device__ __forceinline__ float chain_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

__global__ __launch_bounds__(256) void ac_bwd_tail_kernel(IplanAcBwdArgs a) {
    const IplanAcFwdArgs& fa = a.fwd;
    const int net = (int)blockIdx.y;
    const int which = fa.which == 2 ? (int)blockIdx.z : fa.which;
    const IplanAcNet& nw = which ? fa.critic : fa.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    const IplanAcFeatures& ft = fa.feat;
    const int l = lane_id(), n = l & 15, g = l >> 4;
    const int tile = (int)blockIdx.x * 4 + wave_id();
    const int tiles = (fa.rows + 15) / 16;
    if (tile >= tiles) return;
    const int r = tile * 16 + n;
    const bool valid = r < fa.rows;
    const int64_t pr = valid ? (int64_t)(r / ft.T) * ft.T_phys + (r % ft.T) : 0;
    const int64_t orow = (int64_t)net * fa.rows + (valid ? r : 0);
    const float* sv = fa.saved + (((int64_t)which * fa.n_agents + net) * fa.rows + (valid ? r : 0)) * IPLAN_AC_SAVE_FLOATS;
    float* ds = a.dsave + (((int64_t)which * fa.n_agents + net) * fa.rows + (valid ? r : 0)) * IPLAN_AC_DSAVE_FLOATS;
    const int n_out = nw.n_out;

    f32x4 a1[BT], a2[BT], gr[BT], gz[BT], gn[BT], ghn[BT], hnew[BT], f3[BT];
    for (int t = 0; t < BT; ++t) {
        a1[t] = vload(sv, valid, BM, t);
        a2[t] = vload(sv + 2 * BM, valid, BM, t);
        gr[t] = vload(sv + 4 * BM, valid, BM, t);
        gz[t] = vload(sv + 5 * BM, valid, BM, t);
        gn[t] = vload(sv + 6 * BM, valid, BM, t);
        ghn[t] = vload(sv + 7 * BM, valid, BM, t);
        hnew[t] = vload(sv + 8 * BM, valid, BM, t);
        f3[t] = vload(sv + 9 * BM, valid, BM, t);
    }
    float mu1 = 0.f, rs1 = 0.f, mu2 = 0.f, rs2 = 0.f, mu3 = 0.f, rs3 = 0.f;
    if (valid) {
        const float* st = sv + 10 * BM;
        mu1 = st[2]; rs1 = st[3]; mu2 = st[4]; rs2 = st[5]; mu3 = st[6]; rs3 = st[7];
    }

    // ---- head gradient (D layout, one tile of n_out <= 16 entries)
    f32x4 dhead[1];
    dhead[0] = splat4(0.f);
    if (which == 1) {
        if (valid && g == 0) dhead[0][0] = a.g_values[orow];
    } else {
        // recompute the masked categorical exactly as the forward does
        const f32x4 lg = dense_tile_g_bwd: ;
    }
}

}  // namespace iplan
