// Backward of the recurrent actor / critic (R_Actor.evaluate_actions / R_Critic.forward under
// loss.backward(), learners/ippo_learner.py:202,216) for all agents and both nets per launch:
//   1. ac_bwd_tail_kernel   one wave per 16-row tile, everything in the D layout of wave_tile.h:
//        head^T -> LN3' -> GRU step' -> W_ih^T -> LN2' -> ReLU' -> fc2^T -> LN1' -> ReLU'
//      emits the row-level pre-activation gradients (dz1, dz2, GRU gates, head) and per-tile
//      LayerNorm-parameter partial sums.  Weight gradients of the 64-wide layers are then plain
//      dY^T X contractions (wgrad.hip).
//   2. ac_fc1_wgrad_kernel  the one big contraction G[m][c] = sum_r dz1[r][m] * xhat[r][c]
//      (M = 64, F = 2485 at Highway chaotic, 22 950 rows per agent) on MFMA, with the normalised
//      feature row xhat gathered straight from the episode-buffer fields exactly like the forward
//      (the [rows, F] matrix is never materialised).
//   3. ac_fc1_finalize_kernel  uses LN(F)'s affine structure so that ONE contraction serves three
//      gradients:  dW1 = gamma*G + beta*S,  dgamma = sum_m W1*G,  dbeta = sum_m W1*S  (S = db1).
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int BM = IPLAN_AC_HIDDEN;    // 64
constexpr int BT = BM / 16;            // 4 tiles

// LayerNorm backward on a 64-wide per-chain vector.  dy -> dx (in place); dgam/dbet accumulate.
__device__ __forceinline__ void ln_bwd_tiles(f32x4 (&dy)[BT], const f32x4 (&xhat)[BT], const float* __restrict__ gamma,
                                             float rstd, f32x4 (&dgam)[BT], f32x4 (&dbet)[BT]) {
    float s1 = 0.f, s2 = 0.f;
    f32x4 dxh[BT];
    for (int t = 0; t < BT; ++t) {
        const f32x4 gm = bfrag_a(gamma, t);
        for (int q = 0; q < 4; ++q) {
            dgam[t][q] = dy[t][q] * xhat[t][q];
            dbet[t][q] = dy[t][q];
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

__device__ __forceinline__ void store_ln_part(float* __restrict__ dst, const f32x4 (&dgam)[BT], const f32x4 (&dbet)[BT]) {
    // dst: [gamma(64) | beta(64)] of this tile; lanes with n == 0 write their 4-element slices
    const int l = lane_id(), n = l & 15, g = l >> 4;
    for (int t = 0; t < BT; ++t)
        for (int q = 0; q < 4; ++q) {
            const float sg = chain_sum(dgam[t][q]), sb = chain_sum(dbet[t][q]);
            if (n == 0) {
                dst[16 * t + 4 * g + q] = sg;
                dst[BM + 16 * t + 4 * g + q] = sb;
            }
        }
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
    const int64_t srow = ((int64_t)which * fa.n_agents + net) * fa.rows + (valid ? r : 0);
    const float* sv = fa.saved + srow * IPLAN_AC_SAVE_FLOATS;
    float* ds = a.dsave + srow * IPLAN_AC_DSAVE_FLOATS;
    float* lnp = a.ln_part + (((int64_t)which * fa.n_agents + net) * tiles + tile) * IPLAN_AC_LNPART_FLOATS;
    const int n_out = nw.n_out;

    f32x4 f3[BT], hnew[BT];
    for (int t = 0; t < BT; ++t) {
        hnew[t] = vload(sv + 8 * BM, valid, BM, t);
        f3[t] = vload(sv + 9 * BM, valid, BM, t);
    }
    float mu1 = 0.f, rs1 = 0.f, mu2 = 0.f, rs2 = 0.f, mu3 = 0.f, rs3 = 0.f;
    if (valid) {
        const float* st = sv + 10 * BM;
        mu1 = st[2]; rs1 = st[3]; mu2 = st[4]; rs2 = st[5]; mu3 = st[6]; rs3 = st[7];
    }

    // ---- head gradient (D layout: lane (n,g) holds entries 4g..4g+3 of the n_out <= 16 outputs)
    f32x4 dhead[1];
    dhead[0] = splat4(0.f);
    if (which == 1) {
        if (valid && g == 0) dhead[0][0] = a.g_values[orow];
    } else {
        // recompute the masked categorical exactly as the forward does (distributions.py:64-68)
        const f32x4 lg = dense_tile_ga<BT>(P + nw.off[IPLAN_AC_HEAD_W], BM, n_out, 0, f3, bfrag(P + nw.off[IPLAN_AC_HEAD_B], n_out, 0));
        f32x4 x;
        bool masked[4];
        float m = -INFINITY;
        for (int q = 0; q < 4; ++q) {
            const int idx = 4 * g + q;
            x[q] = lg[q];
            masked[q] = false;
            if (idx < n_out) {
                if (fa.avail && valid && fa.avail[(int64_t)net * fa.av_s_net + pr * fa.av_s_row + idx] == 0) { x[q] = -1e10f; masked[q] = true; }
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
        float ent = 0.f;
        for (int q = 0; q < 4; ++q) {
            lp[q] = x[q] - lse;
            pb[q] = e[q] / se;
            if (4 * g + q < n_out) ent -= pb[q] * lp[q];
        }
        ent = group_sum(ent);
        const int action = valid ? (int)fa.actions_in[(int64_t)net * fa.act_s_net + pr * fa.act_s_row] : 0;
        const float glp = valid ? a.g_logp[orow] : 0.f;
        const float gen = valid ? (a.g_entropy ? a.g_entropy[orow] : a.g_entropy_const) : 0.f;
        for (int q = 0; q < 4; ++q) {
            const int idx = 4 * g + q;
            float d = 0.f;
            if (idx < n_out && !masked[q]) {                 // the in-place mask assignment cuts the gradient
                d = glp * ((idx == action ? 1.0f : 0.0f) - pb[q]);
                d -= gen * pb[q] * (lp[q] + ent);            // d(-sum p log p)/dlogit_k = -p_k (log p_k + H)
            }
            dhead[0][q] = d;
        }
    }
    vstore(ds + 6 * BM, valid, 16, 0, dhead[0]);

    f32x4 dgam[BT], dbet[BT];
    // ---- f3 = LN3(hnew)
    f32x4 d[BT];
    for (int t = 0; t < BT; ++t) d[t] = dense_tile_gt<1>(P + nw.off[IPLAN_AC_HEAD_W], BM, n_out, BM, 16 * t, dhead, splat4(0.f));
    {
        f32x4 xh[BT];
        for (int t = 0; t < BT; ++t)
            for (int q = 0; q < 4; ++q) xh[t][q] = (hnew[t][q] - mu3) * rs3;
        ln_bwd_tiles(d, xh, P + nw.off[IPLAN_AC_LN3_W], rs3, dgam, dbet);
        store_ln_part(lnp, dgam, dbet);
    }
    // ---- GRU step
    f32x4 dg[3 * BT];                                         // [dr | dz | dn_i] tiles
    {
        const float* hsrc = which ? fa.h_critic : fa.h_actor;
        const float* hrow = hsrc + (int64_t)net * fa.hs_net + pr * fa.hs_row;
        for (int t = 0; t < BT; ++t) {
            const f32x4 gr = vload(sv + 4 * BM, valid, BM, t), gz = vload(sv + 5 * BM, valid, BM, t);
            const f32x4 gn = vload(sv + 6 * BM, valid, BM, t), ghn = vload(sv + 7 * BM, valid, BM, t);
            const f32x4 hin = vload(hrow, valid, BM, t);
            const GruGrads o = gru_gates_bwd(d[t], gr, gz, gn, ghn, hin);
            dg[t] = o.dr;
            dg[BT + t] = o.dz;
            dg[2 * BT + t] = o.dni;
            vstore(ds + 2 * BM, valid, BM, t, o.dr);
            vstore(ds + 3 * BM, valid, BM, t, o.dz);
            vstore(ds + 4 * BM, valid, BM, t, o.dni);
            vstore(ds + 5 * BM, valid, BM, t, o.dnh);
        }
    }
    for (int t = 0; t < BT; ++t) d[t] = dense_tile_gta<3 * BT>(P + nw.off[IPLAN_AC_WIH], BM, 16 * t, dg, splat4(0.f));
    // ---- f2 = LN2(a2), a2 = ReLU(fc2(f1))
    {
        f32x4 xh[BT], a2[BT];
        for (int t = 0; t < BT; ++t) {
            a2[t] = vload(sv + 2 * BM, valid, BM, t);
            for (int q = 0; q < 4; ++q) xh[t][q] = (a2[t][q] - mu2) * rs2;
        }
        ln_bwd_tiles(d, xh, P + nw.off[IPLAN_AC_LN2_W], rs2, dgam, dbet);
        store_ln_part(lnp + 2 * BM, dgam, dbet);
        for (int t = 0; t < BT; ++t) {
            for (int q = 0; q < 4; ++q) d[t][q] = a2[t][q] > 0.f ? d[t][q] : 0.f;
            vstore(ds + BM, valid, BM, t, d[t]);                  // dz2
        }
    }
    {
        f32x4 df1[BT];
        for (int t = 0; t < BT; ++t) df1[t] = dense_tile_gta<BT>(P + nw.off[IPLAN_AC_FC2_W], BM, 16 * t, d, splat4(0.f));
        for (int t = 0; t < BT; ++t) d[t] = df1[t];
    }
    // ---- f1 = LN1(a1), a1 = ReLU(fc1(LN_F(x)))
    {
        f32x4 xh[BT], a1[BT];
        for (int t = 0; t < BT; ++t) {
            a1[t] = vload(sv, valid, BM, t);
            for (int q = 0; q < 4; ++q) xh[t][q] = (a1[t][q] - mu1) * rs1;
        }
        ln_bwd_tiles(d, xh, P + nw.off[IPLAN_AC_LN1_W], rs1, dgam, dbet);
        store_ln_part(lnp + 4 * BM, dgam, dbet);
        for (int t = 0; t < BT; ++t) {
            for (int q = 0; q < 4; ++q) d[t][q] = a1[t][q] > 0.f ? d[t][q] : 0.f;
            vstore(ds, valid, BM, t, d[t]);                       // dz1
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G[m][c] = sum_r dz1[r][m] * xhat[r][c]   (xhat = (x - mu_r) * rstd_r, LN(F) without its affine part)
// grid: (column group of 64, row chunk, which * n_agents + net); one wave per workgroup.
struct ColRef {
    int kind;                 // 0 source load, 1 last-action one-hot, 2 constant, 3 padding (xhat = 0)
    const float* base;
    int64_t s_row;
    int idx;
    float cval;
};

__global__ __launch_bounds__(64) void ac_fc1_wgrad_kernel(IplanAcBwdArgs a) {
    const IplanAcFwdArgs& fa = a.fwd;
    const IplanAcFeatures& ft = fa.feat;
    const int nz = (int)blockIdx.z;
    const int which_i = nz / fa.n_agents, net = nz % fa.n_agents;
    const int which = fa.which == 2 ? which_i : fa.which;
    const int l = lane_id(), i = l & 15, g = l >> 4;
    const int W = ft.w[0] + ft.w[1] + ft.w[2];
    const int NW = ft.N * W;
    const int F = NW + ft.n_actions + ft.n_id;
    const int Fpad = (F + 63) / 64 * 64;
    const int c0 = (int)blockIdx.x * 64;
    const int chunk = (int)blockIdx.y;
    const int64_t r_lo = (int64_t)chunk * a.fc1_chunk_rows;
    const int64_t r_hi = r_lo + a.fc1_chunk_rows < fa.rows ? r_lo + a.fc1_chunk_rows : fa.rows;

    ColRef col[4];
    for (int u = 0; u < 4; ++u) {
        int c = c0 + 16 * u + i;
        ColRef& cr = col[u];
        cr.base = nullptr; cr.s_row = 0; cr.idx = 0; cr.cval = 0.f;
        if (c >= F) { cr.kind = 3; continue; }
        if (c < NW) {
            const int e = c / W;
            int k = c - e * W, s = 0;
            if (k >= ft.w[0]) { k -= ft.w[0]; s = 1; if (k >= ft.w[1]) { k -= ft.w[1]; s = 2; } }
            cr.kind = 0;
            cr.base = ft.src[s] + (int64_t)net * ft.s_net[s] + (int64_t)e * ft.w[s] + k;
            cr.s_row = ft.s_row[s];
            continue;
        }
        c -= NW;
        if (c < ft.n_actions) { cr.kind = 1; cr.idx = c; continue; }
        c -= ft.n_actions;
        cr.kind = 2;
        cr.cval = (c == net) ? 1.0f : 0.0f;
    }
    const int64_t sbase = ((int64_t)which * fa.n_agents + net) * fa.rows;
    f32x4 acc[BT][4];
    for (int t = 0; t < BT; ++t)
        for (int u = 0; u < 4; ++u) acc[t][u] = splat4(0.f);
    for (int64_t rb = r_lo; rb < r_hi; rb += 16) {
        for (int s = 0; s < 4; ++s) {
            const int64_t r = rb + 4 * s + g;
            const bool rv = r < r_hi;
            float av[BT], bv[4];
            float mu = 0.f, rstd = 0.f;
            int64_t pr = 0;
            if (rv) {
                const float* st = fa.saved + (sbase + r) * IPLAN_AC_SAVE_FLOATS + 10 * BM;
                mu = st[0]; rstd = st[1];
                pr = (r / ft.T) * ft.T_phys + (r % ft.T);
                const float* dz = a.dsave + (sbase + r) * IPLAN_AC_DSAVE_FLOATS;
                for (int t = 0; t < BT; ++t) av[t] = dz[16 * t + i];
            } else {
                for (int t = 0; t < BT; ++t) av[t] = 0.f;
            }
            int last = -1;
            for (int u = 0; u < 4; ++u) {
                float x = 0.f;
                const ColRef& cr = col[u];
                if (rv && cr.kind != 3) {
                    if (cr.kind == 0) x = cr.base[pr * cr.s_row];
                    else if (cr.kind == 1) {
                        if (last == -1) {
                            if (ft.last_action) last = ft.last_action[(int64_t)net * ft.la_s_net + pr * ft.la_s_row];
                            else if (ft.last_action64) last = (int)ft.last_action64[(int64_t)net * ft.la64_s_net + pr * ft.la64_s_row];
                        }
                        x = (cr.idx == last) ? 1.0f : 0.0f;
                    } else x = cr.cval;
                    x = (x - mu) * rstd;
                }
                bv[u] = x;
            }
            for (int t = 0; t < BT; ++t)
                for (int u = 0; u < 4; ++u) acc[t][u] = mfma4(av[t], bv[u], acc[t][u]);
        }
    }
    float* part = a.g_part + (((int64_t)which_i * fa.n_agents + net) * a.fc1_chunks + chunk) * (int64_t)BM * Fpad;
    for (int t = 0; t < BT; ++t)
        for (int q = 0; q < 4; ++q) {
            const int m = 16 * t + 4 * g + q;
            for (int u = 0; u < 4; ++u) part[(int64_t)m * Fpad + c0 + 16 * u + i] = acc[t][u][q];
        }
}

// grid: (ceil(F/256), n_agents, n_which); thread per feature column
__global__ __launch_bounds__(256) void ac_fc1_finalize_kernel(IplanAcBwdArgs a) {
    const IplanAcFwdArgs& fa = a.fwd;
    const IplanAcFeatures& ft = fa.feat;
    const int net = (int)blockIdx.y, which_i = (int)blockIdx.z;
    const int which = fa.which == 2 ? which_i : fa.which;
    const IplanAcNet& nw = which ? fa.critic : fa.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    float* __restrict__ G = (which ? a.critic_grad : a.actor_grad) + (int64_t)net * (which ? a.critic_grad_s_net : a.actor_grad_s_net);
    const int W = ft.w[0] + ft.w[1] + ft.w[2];
    const int F = ft.N * W + ft.n_actions + ft.n_id;
    const int Fpad = (F + 63) / 64 * 64;
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c >= F) return;
    const float gam = P[nw.off[IPLAN_AC_FN_W] + c], bet = P[nw.off[IPLAN_AC_FN_B] + c];
    const float* __restrict__ W1 = P + nw.off[IPLAN_AC_FC1_W];
    const float* __restrict__ S = G + nw.off[IPLAN_AC_FC1_B];
    const float* __restrict__ part = a.g_part + ((int64_t)which_i * fa.n_agents + net) * a.fc1_chunks * (int64_t)BM * Fpad;
    float dgam = 0.f, dbet = 0.f;
    for (int m = 0; m < BM; ++m) {
        float gsum = 0.f;
        for (int k = 0; k < a.fc1_chunks; ++k) gsum += part[((int64_t)k * BM + m) * Fpad + c];
        const float w = W1[(int64_t)m * F + c], s = S[m];
        G[nw.off[IPLAN_AC_FC1_W] + (int64_t)m * F + c] = fmaf(gam, gsum, bet * s);
        dgam = fmaf(w, gsum, dgam);
        dbet = fmaf(w, s, dbet);
    }
    G[nw.off[IPLAN_AC_FN_W] + c] = dgam;
    G[nw.off[IPLAN_AC_FN_B] + c] = dbet;
}

static int check_bwd_args(const IplanAcBwdArgs* a, const char* what) {
    if (!a) return fail(IPLAN_EINVAL, "%s: null args", what);
    const IplanAcFwdArgs& f = a->fwd;
    if (f.which < 0 || f.which > 2 || f.n_agents < 1 || f.rows < 1 || !f.saved || !a->dsave)
        return fail(IPLAN_EINVAL, "%s: bad which/n_agents/rows or missing saved/dsave", what);
    if (f.which != 1 && (f.mode != 2 || !f.actions_in || !a->g_logp))
        return fail(IPLAN_EINVAL, "%s: actor backward needs mode 2, actions_in and g_logp", what);
    if (f.which != 0 && !a->g_values) return fail(IPLAN_EINVAL, "%s: critic backward needs g_values", what);
    return IPLAN_OK;
}

}  // namespace iplan

extern "C" int iplan_ac_bwd_tail(const IplanAcBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_bwd_args(a, "iplan_ac_bwd_tail")) return rc;
    if (!a->ln_part) return fail(IPLAN_EINVAL, "iplan_ac_bwd_tail: ln_part missing");
    const int tiles = (a->fwd.rows + 15) / 16;
    dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)a->fwd.n_agents, a->fwd.which == 2 ? 2u : 1u);
    hipLaunchKernelGGL(ac_bwd_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_bwd_tail");
}

extern "C" int iplan_ac_bwd_fc1(const IplanAcBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_bwd_args(a, "iplan_ac_bwd_fc1")) return rc;
    if (!a->g_part || a->fc1_chunk_rows < 16 || (a->fc1_chunk_rows & 15) ||
        (int64_t)a->fc1_chunks * a->fc1_chunk_rows < a->fwd.rows)
        return fail(IPLAN_EINVAL, "iplan_ac_bwd_fc1: bad chunking (%d chunks x %d rows for %d rows)", a->fc1_chunks,
                    a->fc1_chunk_rows, a->fwd.rows);
    const IplanAcFeatures& ft = a->fwd.feat;
    const int F = ft.N * (ft.w[0] + ft.w[1] + ft.w[2]) + ft.n_actions + ft.n_id;
    const unsigned nw = a->fwd.which == 2 ? 2u : 1u;
    dim3 grid((unsigned)((F + 63) / 64), (unsigned)a->fc1_chunks, nw * (unsigned)a->fwd.n_agents);
    hipLaunchKernelGGL(ac_fc1_wgrad_kernel, grid, dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_bwd_fc1");
}

extern "C" int iplan_ac_bwd_fc1_finalize(const IplanAcBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_bwd_args(a, "iplan_ac_bwd_fc1_finalize")) return rc;
    if (!a->g_part || (a->fwd.which != 1 && !a->actor_grad) || (a->fwd.which != 0 && !a->critic_grad))
        return fail(IPLAN_EINVAL, "iplan_ac_bwd_fc1_finalize: missing g_part / gradient arenas");
    const IplanAcFeatures& ft = a->fwd.feat;
    const int F = ft.N * (ft.w[0] + ft.w[1] + ft.w[2]) + ft.n_actions + ft.n_id;
    dim3 grid((unsigned)((F + 255) / 256), (unsigned)a->fwd.n_agents, a->fwd.which == 2 ? 2u : 1u);
    hipLaunchKernelGGL(ac_fc1_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_bwd_fc1_finalize");
}
