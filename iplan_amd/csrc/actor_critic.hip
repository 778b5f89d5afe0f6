// Recurrent actor / critic forward (R_Actor, R_Critic) for all agents in one launch, fused with
// the observation assembly of DcntrlMAC._build_inputs[_ippo].
//
// A 16-row tile of one (agent, actor|critic) net is owned by `ksplit` waves of a 512-thread
// workgroup.  The F-wide input row (F = 2485 at Highway chaotic) is gathered straight from the
// episode-buffer fields (history || attention latent || behaviour latent per entity, one-hots
// synthesised in registers), LayerNorm statistics are two-pass over L2-resident data, the
// normalised features feed v_mfma_f32_16x16x4_f32 as the B operand with fc1.weight fragments as A.
// With ksplit = 8 (rollout: 32 rows per agent) the contraction is split across the waves and
// reduced through LDS in a fixed order; with ksplit = 1 (PPO: 22 950 rows per agent) every wave
// streams its own tile.  The 64-wide tail (LN, fc2, LN, GRU step, LN, head, masked log-softmax,
// sampling, entropy) stays in registers of one wave.
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int AM = IPLAN_AC_HIDDEN;    // 64
constexpr int AT = AM / 16;            // 4 tiles

struct FeatCursor {
    // Streams features c0, c0+1, c0+2, c0+3 of one row (c0 % 4 == 0).
    const IplanAcFeatures* f;
    int net;
    int64_t pr;
    int W, NW;
    int last;
    __device__ __forceinline__ float at(int c) const {
        if (c < NW) {
            const int i = c / W, k = c - i * W;
            if (k < f->w[0]) return f->src[0][(int64_t)net * f->s_net[0] + pr * f->s_row[0] + (int64_t)i * f->w[0] + k];
            if (k < f->w[0] + f->w[1])
                return f->src[1][(int64_t)net * f->s_net[1] + pr * f->s_row[1] + (int64_t)i * f->w[1] + (k - f->w[0])];
            return f->src[2][(int64_t)net * f->s_net[2] + pr * f->s_row[2] + (int64_t)i * f->w[2] + (k - f->w[0] - f->w[1])];
        }
        c -= NW;
        if (c < f->n_actions) return c == last ? 1.0f : 0.0f;
        c -= f->n_actions;
        if (c < f->n_id) return c == net ? 1.0f : 0.0f;
        return 0.0f;
    }
};

__global__ __launch_bounds__(512) void ac_fwd_kernel(IplanAcFwdArgs a) {
    __shared__ float s_red[8][16];
    __shared__ __attribute__((aligned(16))) f32x4 s_acc[8][AT][64];

    const int net = (int)blockIdx.y;
    const int which = a.which == 2 ? (int)blockIdx.z : a.which;       // 0 actor, 1 critic
    const IplanAcNet& nw = which ? a.critic : a.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    const IplanAcFeatures& ft = a.feat;
    const int l = lane_id(), w = wave_id(), n = l & 15, g = l >> 4;
    const int ks = a.ksplit;
    const int groups = 8 / ks;
    const int part = w % ks;
    const int tile = (int)blockIdx.x * groups + w / ks;
    const int r = tile * 16 + n;
    const bool valid = r < a.rows;
    const int64_t pr = valid ? (int64_t)(r / ft.T) * ft.T_phys + (r % ft.T) : 0;
    const int W = ft.w[0] + ft.w[1] + ft.w[2];
    const int F = ft.N * W + ft.n_actions + ft.n_id;
    const int KT = (F + 15) / 16;
    const int T_lo = (int)((int64_t)KT * part / ks), T_hi = (int)((int64_t)KT * (part + 1) / ks);

    FeatCursor cur;
    cur.f = &ft; cur.net = net; cur.pr = pr; cur.W = W; cur.NW = ft.N * W;
    cur.last = (valid && ft.n_actions > 0 && ft.last_action) ? ft.last_action[(int64_t)net * ft.la_s_net + pr * ft.la_s_row] : -1;

    // ---- LayerNorm(F) statistics, two passes (mean, then centred second moment)
    float s = 0.f;
    if (valid)
        for (int T = T_lo; T < T_hi; ++T)
            for (int q = 0; q < 4; ++q) { const int c = 16 * T + 4 * g + q; if (c < F) s += cur.at(c); }
    s = group_sum(s);
    if (g == 0) s_red[w][n] = s;
    __syncthreads();
    float mu = 0.f;
    for (int p2 = 0; p2 < ks; ++p2) mu += s_red[(w / ks) * ks + p2][n];
    mu /= (float)F;
    __syncthreads();
    float v2 = 0.f;
    if (valid)
        for (int T = T_lo; T < T_hi; ++T)
            for (int q = 0; q < 4; ++q) { const int c = 16 * T + 4 * g + q; if (c < F) { const float d = cur.at(c) - mu; v2 = fmaf(d, d, v2); } }
    v2 = group_sum(v2);
    if (g == 0) s_red[w][n] = v2;
    __syncthreads();
    float var = 0.f;
    for (int p2 = 0; p2 < ks; ++p2) var += s_red[(w / ks) * ks + p2][n];
    const float rstd = 1.0f / sqrtf(var / (float)F + 1e-5f);

    // ---- fc1 contraction over this wave's share of F
    const float* fnw = P + nw.off[IPLAN_AC_FN_W];
    const float* fnb = P + nw.off[IPLAN_AC_FN_B];
    const float* W1 = P + nw.off[IPLAN_AC_FC1_W];
    f32x4 acc[AT];
    for (int t = 0; t < AT; ++t) acc[t] = splat4(0.f);
    for (int T = T_lo; T < T_hi; ++T) {
        f32x4 x = splat4(0.f);
        for (int q = 0; q < 4; ++q) {
            const int c = 16 * T + 4 * g + q;
            if (valid && c < F) x[q] = (cur.at(c) - mu) * rstd * fnw[c] + fnb[c];
        }
        for (int t = 0; t < AT; ++t) acc[t] = mma_block(wfrag(W1, F, AM, F, 16 * t, 16 * T), x, acc[t]);
    }
    if (ks > 1) {
        for (int t = 0; t < AT; ++t) s_acc[w][t][l] = acc[t];
        __syncthreads();
        if (part == 0) {
            for (int t = 0; t < AT; ++t) {
                f32x4 sum = s_acc[w][t][l];
                for (int p2 = 1; p2 < ks; ++p2) sum += s_acc[w + p2][t][l];
                acc[t] = sum;
            }
        }
    }
    if (part != 0) return;

    // ---- 64-wide tail, one wave per row tile
    float* sv = a.saved ? a.saved + (((int64_t)which * a.n_agents + net) * a.rows + (valid ? r : 0)) * IPLAN_AC_SAVE_FLOATS : nullptr;
    float mu1, rs1, mu2, rs2, mu3, rs3;
    f32x4 f[AT];
    for (int t = 0; t < AT; ++t) {
        f[t] = relu4(acc[t] + bfrag(P + nw.off[IPLAN_AC_FC1_B], AM, t));
        if (sv) vstore(sv, valid, AM, t, f[t]);                       // a1
    }
    layer_norm_tiles<AT>(f, P + nw.off[IPLAN_AC_LN1_W], P + nw.off[IPLAN_AC_LN1_B], &mu1, &rs1);
    if (sv) for (int t = 0; t < AT; ++t) vstore(sv + AM, valid, AM, t, f[t]);   // f1
    f32x4 f2[AT];
    for (int t = 0; t < AT; ++t) {
        f2[t] = relu4(dense_tile_g<AT>(P + nw.off[IPLAN_AC_FC2_W], AM, AM, AM, 16 * t, f, bfrag(P + nw.off[IPLAN_AC_FC2_B], AM, t)));
        if (sv) vstore(sv + 2 * AM, valid, AM, t, f2[t]);             // a2
    }
    layer_norm_tiles<AT>(f2, P + nw.off[IPLAN_AC_LN2_W], P + nw.off[IPLAN_AC_LN2_B], &mu2, &rs2);
    if (sv) for (int t = 0; t < AT; ++t) vstore(sv + 3 * AM, valid, AM, t, f2[t]);  // f2
    // GRU step (rnn.py:24-27) from the stored hidden state
    const float* hsrc = which ? a.h_critic : a.h_actor;
    const float* hrow = hsrc + (int64_t)net * a.hs_net + pr * a.hs_row;
    f32x4 h[AT], hnew[AT];
    for (int t = 0; t < AT; ++t) h[t] = vload(hrow, valid, AM, t);
    {
        const float* Wi = P + nw.off[IPLAN_AC_WIH];
        const float* Wh = P + nw.off[IPLAN_AC_WHH];
        const float* bi = P + nw.off[IPLAN_AC_BIH];
        const float* bh = P + nw.off[IPLAN_AC_BHH];
        for (int t = 0; t < AT; ++t) {
            f32x4 prr = bfrag(bi, 3 * AM, t) + bfrag(bh, 3 * AM, t);
            f32x4 pz = bfrag(bi, 3 * AM, AT + t) + bfrag(bh, 3 * AM, AT + t);
            f32x4 gn = bfrag(bi, 3 * AM, 2 * AT + t);
            f32x4 hn = bfrag(bh, 3 * AM, 2 * AT + t);
            prr = dense_tile_g<AT>(Wi, AM, 3 * AM, AM, 16 * t, f2, prr);
            prr = dense_tile_g<AT>(Wh, AM, 3 * AM, AM, 16 * t, h, prr);
            pz = dense_tile_g<AT>(Wi, AM, 3 * AM, AM, AM + 16 * t, f2, pz);
            pz = dense_tile_g<AT>(Wh, AM, 3 * AM, AM, AM + 16 * t, h, pz);
            gn = dense_tile_g<AT>(Wi, AM, 3 * AM, AM, 2 * AM + 16 * t, f2, gn);
            hn = dense_tile_g<AT>(Wh, AM, 3 * AM, AM, 2 * AM + 16 * t, h, hn);
            const GruGates o = gru_gates(prr, pz, gn, hn, h[t]);
            hnew[t] = o.h;
            if (sv) {
                vstore(sv + 4 * AM, valid, AM, t, o.r);
                vstore(sv + 5 * AM, valid, AM, t, o.z);
                vstore(sv + 6 * AM, valid, AM, t, o.n);
                vstore(sv + 7 * AM, valid, AM, t, o.hn);
                vstore(sv + 8 * AM, valid, AM, t, o.h);
            }
        }
    }
    float* hout = which ? a.h_critic_out : a.h_actor_out;
    if (hout) {
        float* orow = hout + ((int64_t)net * a.rows + (valid ? r : 0)) * AM;
        for (int t = 0; t < AT; ++t) vstore(orow, valid, AM, t, hnew[t]);
    }
    layer_norm_tiles<AT>(hnew, P + nw.off[IPLAN_AC_LN3_W], P + nw.off[IPLAN_AC_LN3_B], &mu3, &rs3);
    if (sv) {
        for (int t = 0; t < AT; ++t) vstore(sv + 9 * AM, valid, AM, t, hnew[t]);      // f3
        if (valid && g == 0) {
            float* st = sv + 10 * AM;
            st[0] = mu; st[1] = rstd; st[2] = mu1; st[3] = rs1; st[4] = mu2; st[5] = rs2; st[6] = mu3; st[7] = rs3;
        }
    }
    // ---- head
    const int n_out = nw.n_out;
    const f32x4 lg = dense_tile_g<AT>(P + nw.off[IPLAN_AC_HEAD_W], AM, n_out, AM, 0, hnew, bfrag(P + nw.off[IPLAN_AC_HEAD_B], n_out, 0));
    const int64_t orow = (int64_t)net * a.rows + (valid ? r : 0);
    if (which == 1) {
        if (valid && g == 0 && a.values) a.values[orow] = lg[0];
        return;
    }
    // masked categorical (distributions.py:64-68, act.py:81-83,159-164)
    f32x4 x;
    float m = -INFINITY;
    for (int q = 0; q < 4; ++q) {
        const int idx = 4 * g + q;
        x[q] = lg[q];
        if (idx < n_out) {
            if (a.avail && valid && a.avail[(int64_t)net * a.av_s_net + pr * a.av_s_row + idx] == 0) x[q] = -1e10f;
            m = fmaxf(m, x[q]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    f32x4 e;
    float se = 0.f;
    for (int q = 0; q < 4; ++q) { e[q] = (4 * g + q < n_out) ? expf(x[q] - m) : 0.f; se += e[q]; }
    se = group_sum(se);
    const float lse = m + logf(se);
    f32x4 lp, pb;
    for (int q = 0; q < 4; ++q) { lp[q] = x[q] - lse; pb[q] = e[q] / se; }
    int action = 0;
    if (a.mode == 2) {
        action = valid ? (int)a.actions_in[(int64_t)net * a.act_s_net + pr * a.act_s_row] : 0;
    } else {
        // argmax of probs (mode 0) or of probs / q (mode 1: torch.multinomial's exponential race)
        f32x4 key;
        float best = -INFINITY;
        for (int q = 0; q < 4; ++q) {
            const int idx = 4 * g + q;
            key[q] = -INFINITY;
            if (idx < n_out) {
                key[q] = pb[q];
                if (a.mode == 1) key[q] = pb[q] / a.q_noise[orow * n_out + idx];
                best = fmaxf(best, key[q]);
            }
        }
        best = fmaxf(best, __shfl_xor(best, 16));
        best = fmaxf(best, __shfl_xor(best, 32));
        int cand = 1 << 30;
        for (int q = 3; q >= 0; --q)
            if (4 * g + q < n_out && key[q] == best) cand = 4 * g + q;
        int o = __shfl_xor(cand, 16); cand = o < cand ? o : cand;
        o = __shfl_xor(cand, 32); cand = o < cand ? o : cand;
        action = cand;
        if (valid && g == 0 && a.actions_out) a.actions_out[orow] = (int64_t)action;
    }
    float sel = 0.f, ent = 0.f;
    for (int q = 0; q < 4; ++q) {
        const int idx = 4 * g + q;
        if (idx < n_out) {
            if (idx == action) sel += lp[q];
            ent -= pb[q] * lp[q];
            if (a.probs && valid) a.probs[orow * n_out + idx] = pb[q];
        }
    }
    sel = group_sum(sel);
    ent = group_sum(ent);
    if (valid && g == 0) {
        if (a.logp) a.logp[orow] = sel;
        if (a.entropy) a.entropy[orow] = ent;
    }
}

}  // namespace iplan

extern "C" int iplan_ac_fwd(const IplanAcFwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a) return fail(IPLAN_EINVAL, "iplan_ac_fwd: null args");
    if (a->ksplit != 1 && a->ksplit != 8) return fail(IPLAN_EINVAL, "iplan_ac_fwd: ksplit must be 1 or 8 (got %d)", a->ksplit);
    if (a->which < 0 || a->which > 2 || a->n_agents < 1 || a->rows < 1)
        return fail(IPLAN_EINVAL, "iplan_ac_fwd: bad which/n_agents/rows");
    if (a->feat.T < 1 || a->feat.T_phys < a->feat.T) return fail(IPLAN_EINVAL, "iplan_ac_fwd: bad T/T_phys");
    if (a->which != 1 && (a->actor.n_out < 1 || a->actor.n_out > 16))
        return fail(IPLAN_EINVAL, "iplan_ac_fwd: n_actions=%d outside [1,16]", a->actor.n_out);
    if (a->which != 1 && a->mode == 1 && !a->q_noise) return fail(IPLAN_EINVAL, "iplan_ac_fwd: mode 1 needs q_noise");
    if (a->which != 1 && a->mode == 2 && !a->actions_in) return fail(IPLAN_EINVAL, "iplan_ac_fwd: mode 2 needs actions_in");
    const int tiles = (a->rows + 15) / 16;
    const int groups = 8 / a->ksplit;
    dim3 grid((unsigned)((tiles + groups - 1) / groups), (unsigned)a->n_agents, a->which == 2 ? 2u : 1u);
    hipLaunchKernelGGL(ac_fwd_kernel, grid, dim3(512), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_fwd");
}
