// Backward of the recurrent actor / critic (R_Actor.evaluate_actions / R_Critic.forward under
// loss.backward(), learners/ippo_learner.py:202,216) for all agents and both nets per launch:
//   1. ac_bwd_tail_kernel   one wave per 16-row tile, everything in the D layout of wave_tile.h:
//        head^T -> LN3' -> GRU step' -> W_ih^T -> LN2' -> ReLU' -> fc2^T -> LN1' -> ReLU'
//      emits the row-level pre-activation gradients (dz1, dz2, GRU gates, head) and per-tile
//      LayerNorm-parameter partial sums.  Weight gradients of the 64-wide layers are then plain
//      dY^T X contractions (wgrad.hip).
//   2. the one big contraction G[m][c] = sum_r dz1[r][m] * xhat[r][c]: in the full-batch PPO epochs
//      ac_fc1_split_wgrad_kernel (ac_fc1_split.hip: split-bf16, packed xhat, actor and critic in one pass); otherwise
//      ac_fc1_wgrad_kernel here
//      (M = 64, F = 2485 at Highway chaotic, 22 950 rows per agent) on MFMA, with the normalised
//      feature row xhat gathered straight from the episode-buffer fields exactly like the forward
//      (the [rows, F] matrix is never materialised).
//   3. ac_fc1_finalize_kernel  uses LN(F)'s affine structure so that ONE contraction serves three
//      gradients:  dW1 = gamma*G + beta*S,  dgamma = sum_m W1*G,  dbeta = sum_m W1*S  (S = db1).
#include "api_util.h"
#include "gru_tile.h"
#include "ac_kmap.h"

namespace iplan {

constexpr int BM = IPLAN_AC_HIDDEN;    // 64
constexpr int BT = BM / 16;            // 4 tiles

// LayerNorm backward on a 64-wide per-chain vector.  dy -> dx (in place); dgam/dbet accumulate.
// (gamma: 64 floats in LDS)
__device__ __forceinline__ void ln_bwd_tiles(f32x4 (&dy)[BT], const f32x4 (&xhat)[BT], const float* __restrict__ gamma,
                                             float rstd, f32x4 (&dgam)[BT], f32x4 (&dbet)[BT]) {
    float s1 = 0.f, s2 = 0.f;
    f32x4 dxh[BT];
    for (int t = 0; t < BT; ++t) {
        const f32x4 gm = bfrag_lds(gamma, t);
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

// The three backward-data products of a tile (head^T, W_ih^T, fc2^T) read their weights TRANSPOSED from LDS: a workgroup of
// BWD_TAIL_WAVES waves stages  W_ih^T [64][192], fc2^T [64][64], head^T [64][16]  (76 KB: two workgroups per CU, four waves per
// SIMD at <= 128 registers) so that an A fragment is one ds_read_b128.  Read from global memory they were 4 dword loads per
// fragment from 4 rows of the row-major weight (68 KB per tile through L1/L2, each layer's loads in front of its MFMA chain):
// the launch sat at 14 % MFMA-busy with 42 % of its wave-cycles waiting for an instruction's operands
// (profiles/r04d_pmc_ppo_train.txt).  Summation order per accumulator as before (k-tile outer, sub-step inner).
constexpr int BWD_TAIL_WAVES = 8;
constexpr int BT_LDI = 3 * BM + 8, BT_LDF = BM + 8, BT_LDH = 16 + 8;      // row strides (floats): conflict-free b128 fragments
struct AcBwdTailShared {
    __attribute__((aligned(16))) float wih_t[BM * BT_LDI];                // W_ih^T:  [input feature][gate row]
    __attribute__((aligned(16))) float fc2_t[BM * BT_LDF];                // fc2^T
    __attribute__((aligned(16))) float head_t[BM * BT_LDH];               // head^T, rows >= n_out zero
    __attribute__((aligned(16))) float gam[3][BM];                        // LN3 / LN2 / LN1 weight
};

__global__ __launch_bounds__(64 * BWD_TAIL_WAVES, 4) void ac_bwd_tail_kernel(IplanAcBwdArgs a) {
    __shared__ AcBwdTailShared sh;
    const IplanAcFwdArgs& fa = a.fwd;
    const bool act_tanh = fa.act_tanh != 0;                     // (uniform) MLPBase activation of the forward pass
    const int net = (int)blockIdx.y;
    const int which = fa.which == 2 ? (int)blockIdx.z : fa.which;
    const IplanAcNet& nw = which ? fa.critic : fa.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    const IplanAcFeatures& ft = fa.feat;
    const int l = lane_id(), n = l & 15, g = l >> 4;
    const int tile = (int)blockIdx.x * BWD_TAIL_WAVES + wave_id();
    const int tiles = (fa.rows + 15) / 16;
    {
        // 16-byte chunks of the row-major weights (coalesced), scattered into the transposed LDS images
        const float* Wi = P + nw.off[IPLAN_AC_WIH];
        const float* W2 = P + nw.off[IPLAN_AC_FC2_W];
        const float* Wh = P + nw.off[IPLAN_AC_HEAD_W];
        constexpr int NT = 64 * BWD_TAIL_WAVES;
        for (int c = (int)threadIdx.x; c < (3 * BM + BM + 16) * 16; c += NT) {
            const int r = c >> 4, c4 = c & 15;
            if (r < 3 * BM) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Wi + r * BM + 4 * c4);
                for (int q = 0; q < 4; ++q) sh.wih_t[(4 * c4 + q) * BT_LDI + r] = v[q];
            } else if (r < 4 * BM) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(W2 + (r - 3 * BM) * BM + 4 * c4);
                for (int q = 0; q < 4; ++q) sh.fc2_t[(4 * c4 + q) * BT_LDF + (r - 3 * BM)] = v[q];
            } else {
                const int k = r - 4 * BM;
                f32x4 v = splat4(0.f);
                if (k < nw.n_out) v = *reinterpret_cast<const f32x4*>(Wh + k * BM + 4 * c4);
                for (int q = 0; q < 4; ++q) sh.head_t[(4 * c4 + q) * BT_LDH + k] = v[q];
            }
        }
        const int gsrc[3] = {IPLAN_AC_LN3_W, IPLAN_AC_LN2_W, IPLAN_AC_LN1_W};
        for (int i = (int)threadIdx.x; i < 3 * BM; i += NT) sh.gam[i / BM][i % BM] = P[nw.off[gsrc[i / BM]] + i % BM];
    }
    __syncthreads();
    if (tile >= tiles) return;
    const int o4[BT] = {0, 16, 32, 48};
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
    for (int t = 0; t < BT; ++t) d[t] = splat4(0.f);
    dense_multi<BT, 1>(sh.head_t, BT_LDH, o4, 0, dhead, d);
    {
        f32x4 xh[BT];
        for (int t = 0; t < BT; ++t)
            for (int q = 0; q < 4; ++q) xh[t][q] = (hnew[t][q] - mu3) * rs3;
        ln_bwd_tiles(d, xh, sh.gam[0], rs3, dgam, dbet);
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
    for (int t = 0; t < BT; ++t) d[t] = splat4(0.f);
    dense_multi<BT, 3 * BT>(sh.wih_t, BT_LDI, o4, 0, dg, d);
    // ---- f2 = LN2(a2), a2 = ReLU(fc2(f1))
    {
        f32x4 xh[BT], a2[BT];
        for (int t = 0; t < BT; ++t) {
            a2[t] = vload(sv + 2 * BM, valid, BM, t);
            for (int q = 0; q < 4; ++q) xh[t][q] = (a2[t][q] - mu2) * rs2;
        }
        ln_bwd_tiles(d, xh, sh.gam[1], rs2, dgam, dbet);
        store_ln_part(lnp + 2 * BM, dgam, dbet);
        for (int t = 0; t < BT; ++t) {
            for (int q = 0; q < 4; ++q) d[t][q] = act_tanh ? d[t][q] * (1.0f - a2[t][q] * a2[t][q]) : (a2[t][q] > 0.f ? d[t][q] : 0.f);
            vstore(ds + BM, valid, BM, t, d[t]);                  // dz2
        }
    }
    {
        f32x4 df1[BT];
        for (int t = 0; t < BT; ++t) df1[t] = splat4(0.f);
        dense_multi<BT, BT>(sh.fc2_t, BT_LDF, o4, 0, d, df1);
        for (int t = 0; t < BT; ++t) d[t] = df1[t];
    }
    // ---- f1 = LN1(a1), a1 = ReLU(fc1(LN_F(x)))
    {
        f32x4 xh[BT], a1[BT];
        for (int t = 0; t < BT; ++t) {
            a1[t] = vload(sv, valid, BM, t);
            for (int q = 0; q < 4; ++q) xh[t][q] = (a1[t][q] - mu1) * rs1;
        }
        ln_bwd_tiles(d, xh, sh.gam[2], rs1, dgam, dbet);
        store_ln_part(lnp + 4 * BM, dgam, dbet);
        for (int t = 0; t < BT; ++t) {
            for (int q = 0; q < 4; ++q) d[t][q] = act_tanh ? d[t][q] * (1.0f - a1[t][q] * a1[t][q]) : (a1[t][q] > 0.f ? d[t][q] : 0.f);
            vstore(ds, valid, BM, t, d[t]);                       // dz1
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G[m][kb] = sum_r dz1[r][m] * xhat[r][kb]   (xhat = (x - mu_r) * rstd_r, LN(F) without its affine part), with the
// feature axis in the SOURCE-MAJOR K order of ac_kmap.h.  Same data path as wgrad.hip's wide jobs: a wave owns all
// 4 o-tiles x up to FC1_KG k-tiles (128 accumulator registers, two waves per SIMD), both operands are loaded from
// global memory directly in MFMA operand order (sub-step s: lane (i, g) reads dz1[row 4s+g][16t+i] and feature
// 16u+i of row 4s+g from its source field), addressing is one uniform base per field + 32-bit byte offsets advanced
// incrementally, and the registers of a sub-step are reloaded for the next block as soon as its MFMAs are issued.
// dz1 is re-read once per k-group (21 groups at F = 2485), the feature fields exactly once.
// A wave job never straddles two source blocks of the K order (each block's k-tiles are cut into groups of at most
// FC1_KG), so a job reads ONE field: one base, one row offset per sub-step, no per-tile selection.
// grid: (k-group, row chunk, which * n_agents + net); one wave per workgroup.
// FC1_KG = 8 (128 accumulator registers, 244 in all: TWO waves per SIMD) against 12 (one wave per SIMD), same box: ac_backward
// 2.31 -> 1.99 ms, PPO train() 61.8 -> 57.5 ms; 10 / 6 / 5 / 4 k-tiles: 2.44 / 2.04 / 2.15 / 2.07 ms (profiles/r02g_notes.md).
// The caller sizes the row chunks so that k-groups x chunks x nets = 2048 waves.
#ifndef AC_FC1_KG
#define AC_FC1_KG 8
#endif
constexpr int FC1_KG = AC_FC1_KG;

struct Fc1Group { int blk, T0, nkt; };
// k-group `gid` of the launch -> (source block, first k-tile, tiles); returns the number of groups when gid < 0
__host__ __device__ inline int fc1_group(const int (&kt0)[5], int gid, Fc1Group* out) {
    int n = 0;
    for (int b = 0; b < 4; ++b) {
        const int tiles = kt0[b + 1] - kt0[b];
        const int ng = (tiles + FC1_KG - 1) / FC1_KG;
        if (gid >= n && gid < n + ng && out) {
            const int j = gid - n;
            out->blk = b;
            out->T0 = kt0[b] + (int)(((int64_t)j * tiles) / ng);
            out->nkt = kt0[b] + (int)(((int64_t)(j + 1) * tiles) / ng) - out->T0;
        }
        n += ng;
    }
    return n;
}

__global__ __launch_bounds__(64) void ac_fc1_wgrad_kernel(IplanAcBwdArgs a) {
    const IplanAcFwdArgs& fa = a.fwd;
    const IplanAcFeatures& ft = fa.feat;
    const int nz = (int)blockIdx.z;
    const int which_i = nz / fa.n_agents, net = nz % fa.n_agents;
    const int which = fa.which == 2 ? which_i : fa.which;
    const int l = lane_id(), i = l & 15, g = l >> 4;
    const KMap km = make_kmap(ft);
    const int KT = km.kt0[4], Kpad = KT * 16;
    Fc1Group grp;
    grp.blk = 0; grp.T0 = 0; grp.nkt = 0;
    fc1_group(km.kt0, (int)blockIdx.x, &grp);
    const int T0 = grp.T0, nkt = grp.nkt, blk = grp.blk;
    const int chunk = (int)blockIdx.y;
    const int64_t r_lo = (int64_t)chunk * a.fc1_chunk_rows;
    const int64_t r_hi = r_lo + a.fc1_chunk_rows < fa.rows ? r_lo + a.fc1_chunk_rows : fa.rows;
    if (r_lo >= r_hi || nkt <= 0) return;
    const int n_rows = (int)(r_hi - r_lo);
    const int64_t sbase = ((int64_t)which * fa.n_agents + net) * fa.rows;
    const bool onehot = blk == 3;
    const int sk = onehot ? 0 : blk;                           // source field of this job

    // uniform bases; everything per lane is a 32-bit byte offset from them
    const char* __restrict__ dzb = reinterpret_cast<const char*>(a.dsave + (sbase + r_lo) * IPLAN_AC_DSAVE_FLOATS);
    const char* __restrict__ stb = reinterpret_cast<const char*>(fa.saved + (sbase + r_lo) * IPLAN_AC_SAVE_FLOATS + 10 * BM);
    const bool la32 = ft.n_actions > 0 && ft.last_action != nullptr, la64 = ft.n_actions > 0 && !la32 && ft.last_action64 != nullptr;
    const char* __restrict__ fb =                              // the field the job's feature operand comes from
        onehot ? (la32 ? reinterpret_cast<const char*>(ft.last_action + (int64_t)net * ft.la_s_net)
                       : (la64 ? reinterpret_cast<const char*>(ft.last_action64 + (int64_t)net * ft.la64_s_net) : nullptr))
               : reinterpret_cast<const char*>(ft.src[sk] + (int64_t)net * ft.s_net[sk]);
    const int64_t f_row = onehot ? (la32 ? 4 * ft.la_s_row : 8 * ft.la64_s_row) : 4 * ft.s_row[sk];   // bytes per physical row

    // this lane's entry of every k-tile: byte offset inside a source row (one-hot block: the index itself).  Entries
    // past the block's end are clamped: they only feed partial columns that the finalize kernel never reads.
    uint32_t foff[FC1_KG];
#pragma unroll
    for (int u = 0; u < FC1_KG; ++u) {
        const int f = imin(16 * (T0 + imin(u, nkt - 1) - km.kt0[blk]) + i, km.len[blk] - 1);
        foff[u] = onehot ? (uint32_t)f : 4u * (uint32_t)f;
    }
    // row cursors, one per sub-step s (rows 4s + g of the block): row inside the chunk, step inside the episode and the
    // offset of the physical row in the field; a block = 16 rows = adv_ep whole episodes + adv_t steps
    const int adv_ep = 16 / ft.T, adv_t = 16 % ft.T;
    const uint32_t f_adv = (uint32_t)(((int64_t)adv_ep * ft.T_phys + adv_t) * f_row);
    const uint32_t f_wrap = (uint32_t)((int64_t)(ft.T_phys - ft.T) * f_row);
    int rrel[4], tt[4];
    uint32_t fo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int64_t r = r_lo + 4 * s + g;
        const int64_t ep = r / ft.T;
        rrel[s] = 4 * s + g;
        tt[s] = (int)(r - ep * ft.T);
        fo[s] = (uint32_t)((ep * ft.T_phys + tt[s]) * f_row);
    }
    f32x4 acc[BT][FC1_KG];
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int u = 0; u < FC1_KG; ++u) acc[t][u] = splat4(0.f);
    float av[4][BT], bv[4][FC1_KG];
    // load sub-step s of the next block (tail: its rows may lie past the chunk's end)
    auto fetch = [&](int s, bool tail) {
        const bool rv = !tail || rrel[s] < n_rows;
        uint32_t of = fo[s];
        uint32_t od = (uint32_t)rrel[s] * (uint32_t)(4 * IPLAN_AC_DSAVE_FLOATS), os = (uint32_t)rrel[s] * (uint32_t)(4 * IPLAN_AC_SAVE_FLOATS);
        if (tail) { of = rv ? of : 0u; od = rv ? od : 0u; os = rv ? os : 0u; }
        rrel[s] += 16;
        tt[s] += adv_t;
        fo[s] += f_adv;
        if (tt[s] >= ft.T) { tt[s] -= ft.T; fo[s] += f_wrap; }
        const float mu = *reinterpret_cast<const float*>(stb + os), rstd = *reinterpret_cast<const float*>(stb + (os + 4u));
        const float nm = -mu * rstd;
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const float v = *reinterpret_cast<const float*>(dzb + (od + (uint32_t)(64 * t + 4 * i)));
            av[s][t] = rv ? v : 0.f;
        }
        if (!onehot) {
#pragma unroll
            for (int u = 0; u < FC1_KG; ++u) bv[s][u] = fmaf(*reinterpret_cast<const float*>(fb + (of + foff[u])), rstd, nm);
        } else {
            int last = -1;
            if (la32) last = *reinterpret_cast<const int32_t*>(fb + of);
            else if (la64) last = (int)*reinterpret_cast<const int64_t*>(fb + of);
#pragma unroll
            for (int u = 0; u < FC1_KG; ++u) {
                const int idx = (int)foff[u];
                const float x = idx < km.n_actions ? (idx == last ? 1.0f : 0.0f) : (idx - km.n_actions == net ? 1.0f : 0.0f);
                bv[s][u] = fmaf(x, rstd, nm);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) fetch(s, 16 > n_rows);
    for (int rb = 0; rb < n_rows; rb += 16) {
        const bool tail = rb + 32 > n_rows;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < BT; ++t)
#pragma unroll
                for (int u = 0; u < FC1_KG; ++u) acc[t][u] = mfma4(av[s][t], bv[s][u], acc[t][u]);
            fetch(s, tail);                                  // rolling prefetch of the next block's sub-step s
        }
    }
    float* part = a.g_part + (((int64_t)which_i * fa.n_agents + net) * a.fc1_chunks + chunk) * (int64_t)BM * Kpad;
#pragma unroll
    for (int t = 0; t < BT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = 16 * t + 4 * g + q;
#pragma unroll
            for (int u = 0; u < FC1_KG; ++u)
                if (u < nkt) part[(int64_t)m * Kpad + (T0 + u) * 16 + i] = acc[t][u][q];
        }
}

// grid: (KT, n_agents, n_which), 256 threads = the 16 K-order positions of k-tile T x 16 groups of output rows (m = mg, mg + 16,
// mg + 32, mg + 48): adjacent threads read adjacent partial-tile entries, every (m, k) pair sums its row chunks in chunk order,
// and the column sums over m (dgamma, dbeta) meet in LDS in a fixed order.  (The first form -- one thread per feature column
// looping over all 64 rows x chunks, 100 workgroups -- took 0.15 - 0.28 ms for 32 - 64 MB of partial tiles.)
__global__ __launch_bounds__(256) void ac_fc1_finalize_kernel(IplanAcBwdArgs a) {
    __shared__ float s_g[16][17], s_b[16][17];
    const IplanAcFwdArgs& fa = a.fwd;
    const IplanAcFeatures& ft = fa.feat;
    const int T = (int)blockIdx.x, net = (int)blockIdx.y, which_i = (int)blockIdx.z;
    const int which = fa.which == 2 ? which_i : fa.which;
    const IplanAcNet& nw = which ? fa.critic : fa.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    float* __restrict__ G = (which ? a.critic_grad : a.actor_grad) + (int64_t)net * (which ? a.critic_grad_s_net : a.actor_grad_s_net);
    const KMap km = make_kmap(ft);
    const int F = km.NW + km.n_actions + km.n_id;
    const int Kpad = km.kt0[4] * 16;
    const int j = (int)threadIdx.x & 15, mg = (int)threadIdx.x >> 4;
    const KTile kt = ktile_at(km, T, 4 * (j >> 2));
    const bool live = (j & 3) < kt.nv;
    const int c = live ? kt.c[j & 3] : 0;
    const float gam = live ? P[nw.off[IPLAN_AC_FN_W] + c] : 0.f, bet = live ? P[nw.off[IPLAN_AC_FN_B] + c] : 0.f;
    const float* __restrict__ W1 = P + nw.off[IPLAN_AC_FC1_W];
    const float* __restrict__ S = G + nw.off[IPLAN_AC_FC1_B];
    const float* __restrict__ part = a.g_part + ((int64_t)which_i * fa.n_agents + net) * a.fc1_chunks * (int64_t)BM * Kpad + T * 16 + j;
    float dgam = 0.f, dbet = 0.f;
    for (int mi = 0; mi < 4; ++mi) {
        const int m = mg + 16 * mi;
        float gsum = 0.f;
        for (int k = 0; k < a.fc1_chunks; ++k) gsum += part[((int64_t)k * BM + m) * Kpad];
        if (live) {
            const float w = W1[(int64_t)m * F + c], sm = S[m];
            G[nw.off[IPLAN_AC_FC1_W] + (int64_t)m * F + c] = fmaf(gam, gsum, bet * sm);
            dgam = fmaf(w, gsum, dgam);
            dbet = fmaf(w, sm, dbet);
        }
    }
    s_g[mg][j] = dgam;
    s_b[mg][j] = dbet;
    __syncthreads();
    if (mg == 0 && live) {
        float sg = 0.f, sb = 0.f;
        for (int q = 0; q < 16; ++q) { sg += s_g[q][j]; sb += s_b[q][j]; }
        G[nw.off[IPLAN_AC_FN_W] + c] = sg;
        G[nw.off[IPLAN_AC_FN_B] + c] = sb;
    }
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

// padded length of the source-major K order (multiple of 16): N*w_s rounded up per source, then the one-hots
extern "C" int iplan_ac_kpad(const IplanAcFeatures* ft) {
    int t = 0;
    for (int s = 0; s < 3; ++s) t += (ft->N * ft->w[s] + 15) / 16;
    t += (ft->n_actions + ft->n_id + 15) / 16;
    return t * 16;
}

extern "C" int iplan_ac_bwd_tail(const IplanAcBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_bwd_args(a, "iplan_ac_bwd_tail")) return rc;
    if (!a->ln_part) return fail(IPLAN_EINVAL, "iplan_ac_bwd_tail: ln_part missing");
    const int tiles = (a->fwd.rows + 15) / 16;
    dim3 grid((unsigned)((tiles + BWD_TAIL_WAVES - 1) / BWD_TAIL_WAVES), (unsigned)a->fwd.n_agents, a->fwd.which == 2 ? 2u : 1u);
    hipLaunchKernelGGL(ac_bwd_tail_kernel, grid, dim3(64 * BWD_TAIL_WAVES), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_bwd_tail");
}

extern "C" int iplan_ac_fc1_groups(const IplanAcFeatures* ft) {
    using namespace iplan;
    if (!ft) return 0;
    int kt0[5], t = 0;
    for (int s = 0; s < 3; ++s) { kt0[s] = t; t += (ft->N * ft->w[s] + 15) / 16; }
    kt0[3] = t;
    t += (ft->n_actions + ft->n_id + 15) / 16;
    kt0[4] = t;
    return fc1_group(kt0, -1, nullptr);
}

extern "C" int iplan_ac_bwd_fc1(const IplanAcBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_bwd_args(a, "iplan_ac_bwd_fc1")) return rc;
    if (!a->g_part || a->fc1_chunk_rows < 16 || (a->fc1_chunk_rows & 15) ||
        (int64_t)a->fc1_chunks * a->fc1_chunk_rows < a->fwd.rows)
        return fail(IPLAN_EINVAL, "iplan_ac_bwd_fc1: bad chunking (%d chunks x %d rows for %d rows)", a->fc1_chunks,
                    a->fc1_chunk_rows, a->fwd.rows);
    const unsigned nw = a->fwd.which == 2 ? 2u : 1u;
    dim3 grid((unsigned)iplan_ac_fc1_groups(&a->fwd.feat), (unsigned)a->fc1_chunks, nw * (unsigned)a->fwd.n_agents);
    hipLaunchKernelGGL(ac_fc1_wgrad_kernel, grid, dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_bwd_fc1");
}

extern "C" int iplan_ac_bwd_fc1_finalize(const IplanAcBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_bwd_args(a, "iplan_ac_bwd_fc1_finalize")) return rc;
    if (!a->g_part || (a->fwd.which != 1 && !a->actor_grad) || (a->fwd.which != 0 && !a->critic_grad))
        return fail(IPLAN_EINVAL, "iplan_ac_bwd_fc1_finalize: missing g_part / gradient arenas");
    dim3 grid((unsigned)(iplan_ac_kpad(&a->fwd.feat) / 16), (unsigned)a->fwd.n_agents, a->fwd.which == 2 ? 2u : 1u);
    hipLaunchKernelGGL(ac_fc1_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_bwd_fc1_finalize");
}
