// Body of the recurrent actor / critic forward (R_Actor, R_Critic; see actor_critic.hip), as a device function so that the
// rollout's fused launch (gat.hip: the GAT scenes, the behaviour encoder's tiles and the NEXT step's action selection in one
// grid) can run it as trailing workgroups.  actor_critic.hip wraps it into ac_fwd_kernel<RT, PRE>.
#pragma once
#include "api_util.h"
#include "gru_tile.h"
#include "ac_kmap.h"

namespace iplan {

// MLPBase activation (utils/mappo_utils/mlp.py:10: [nn.Tanh(), nn.ReLU()][use_ReLU]); `tanh` is wave-uniform
__device__ __forceinline__ f32x4 ac_act4(f32x4 v, bool tanh) {
    f32x4 r;
    for (int q = 0; q < 4; ++q) r[q] = tanh ? tanh_f(v[q]) : (v[q] > 0.0f ? v[q] : 0.0f);
    return r;
}

constexpr int AM = IPLAN_AC_HIDDEN;    // 64
constexpr int AT = AM / 16;            // 4 tiles
#ifndef AC_RT
#define AC_RT 2                       // row tiles per wave in the streaming (PPO) variant
#endif


// LDS of one workgroup of the forward (a struct, so that the fused launch can overlay it with the GAT scene's)
template <int RT, int NW = 8>          // NW: waves per workgroup
struct AcShared {
#ifdef AC_NO_STAGED_TAIL                // A/B builds (scripts/build_variants.sh)
    static constexpr bool STAGED_BUILD = false, STAGED2 = false;
#else
    static constexpr bool STAGED_BUILD = RT == 1, STAGED2 = RT == 2;
#endif
    static constexpr int TLDW = AM + 8;
    static constexpr int TW_ROWS = STAGED2 ? AM + 3 * AM + 16 + 3 * AM : (STAGED_BUILD ? AM + 3 * AM + 16 : 0);
    static constexpr int TP_FLOATS = 912;
    float red[NW][16], red2[NW][16], cc[NW][2][AM];
    float hand[4];                        // [0]: the tail's hand-over flag (1 = this workgroup runs the tail)
    __attribute__((aligned(16))) f32x4 acc[RT == 1 ? 8 : 1][AT][64];
    __attribute__((aligned(16))) float tw[TW_ROWS ? TW_ROWS * TLDW : 4];
    __attribute__((aligned(16))) f32x4 gh[STAGED_BUILD ? 3 * AT : 1][64];
    __attribute__((aligned(16))) float tp[TW_ROWS ? TP_FLOATS : 4];
};

struct AcGrid {                         // the workgroup's position in the (tiles x ksplit_wg, n_agents, actor|critic) grid
    int bx, by, bz, gx, gy, gz;
};

// Fused rollout launch: the workgroups in front of the actor/critic ones (GAT scenes, encoder tiles) count themselves into
// sync[0] when their outputs are written; the actor/critic workgroups wait for n_prod of them.  sync[1] counts the consumers
// that are past the wait -- the last of n_cons zeroes both counters for the next launch (stream order keeps launches apart);
// sync[2] is set when a wait gave up (a poll limit instead of a hang: the launch then finishes with stale inputs and the host
// raises).  Forward progress: workgroups are dispatched in index order, so every producer is resident or done before the first
// consumer occupies a slot.
// No cache maintenance on either side (an agent-scope release / acquire pair is an L2 write-back / invalidate per workgroup --
// measured: the launch 65 us longer, polling with acquire loads included): the few values that cross -- the new attention
// and behaviour latents -- are written with device-coherent stores (vstore_c<true>) and read with device-coherent loads
// (coh_load16_untracked in the operand ring, kfeat<true> for the ragged tiles), the producers drain their stores (vmcnt 0)
// before they count themselves in, and the consumers issue the loads after they saw the count.  A wait that gives up leaves the
// counters unusable (late producers still count): sync[2] stays set and the host refuses to go on (ops.check_fused_sync).
struct AcProducers {
    int32_t* sync;
    int n_prod, n_cons;
};

__device__ __forceinline__ void ac_signal_producer_done(int32_t* sync) {
#ifdef IPLAN_HOST_EMULATION
    __syncthreads();
    if (threadIdx.x == 0) sync[0] += 1;
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every wave: its coherent stores have completed
    __syncthreads();
    if (threadIdx.x == 0) {
#if IPLAN_FUSED_FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

__device__ __forceinline__ void ac_wait_producers(const AcProducers& pr) {
#ifdef IPLAN_HOST_EMULATION
    // (the emulator runs the workgroups of a launch in index order: the producers are done)
    if (threadIdx.x == 0) {
        if (pr.sync[0] < pr.n_prod) pr.sync[2] = 1;
        if (++pr.sync[1] == pr.n_cons) { pr.sync[0] = 0; pr.sync[1] = 0; }
    }
    __syncthreads();
#else
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(pr.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pr.n_prod) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { __hip_atomic_store(pr.sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
#if IPLAN_FUSED_FENCES
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        if (__hip_atomic_fetch_add(pr.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pr.n_cons - 1) {
            __hip_atomic_store(pr.sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pr.sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
#endif
}

template <int RT, bool PRE, bool FUSED, int NW = 8>
__device__ __forceinline__ void ac_fwd_body(const IplanAcFwdArgs& a, const AcGrid& gp, AcShared<RT, NW>& sh, const AcProducers& prod) {
    auto& s_red = sh.red;
    auto& s_red2 = sh.red2;
    auto& s_cc = sh.cc;
    auto& s_acc = sh.acc;                                                                  // (cross-wave K reduction: rollout shape only)
    // Rollout shape (RT == 1, the 8 waves of a workgroup share ONE 16-row tile): the 64-wide tail runs in a single wave and
    // was a chain of ~10 dependent stages, each waiting ~1 us for its weight fragments from L2 (15 us of a 52 us launch).
    // All 8 waves therefore stage fc2 / GRU W_ih / head weights into LDS at kernel entry (the loads land during the fc1
    // contraction) and share out the 12 tiles of  W_hh h + b_hh  (it does not depend on fc1); the tail wave then reads
    // everything from LDS.  The streaming instantiation (RT == 2, one wave per tile pair) keeps reading L1/L2.
    constexpr int TLDW = AM + 8;                                                  // conflict-free ds_read_b128 fragments
    // The streaming instantiation (RT == 2, one wave per tile pair, 8 independent waves) stages the same weights PLUS W_hh
    // (464 rows, 131 KiB: one workgroup per CU either way at 199 registers) -- its tail read 133 KB of weight fragments per
    // row tile from L1/L2, each fragment load followed by its wait: 112 us of a wave's 357 us (phase clocks, profiles/r02d_notes.md).
#ifdef AC_NO_STAGED_TAIL                // A/B builds (scripts/build_variants.sh)
    constexpr bool STAGED_BUILD = false, STAGED2 = false;
#else
    constexpr bool STAGED_BUILD = RT == 1, STAGED2 = RT == 2;
#endif
    constexpr int TW_ROWS = AcShared<RT, NW>::TW_ROWS;
    static_assert(TLDW == AcShared<RT, NW>::TLDW, "tail weight row stride");
    auto& s_tw = sh.tw;                                                                    // fc2 | W_ih | head rows (| W_hh)
    auto& s_gh = sh.gh;
    // the tail's parameter VECTORS (biases, LayerNorm gamma / beta): 26 of them per row tile, each a dependent ~1 us L2 round
    // trip when read where it is used (37 of the 50 us a row tile's tail took in the streaming form)
    constexpr int TP_FC1B = 0, TP_LN1W = 64, TP_LN1B = 128, TP_FC2B = 192, TP_LN2W = 256, TP_LN2B = 320, TP_BIH = 384, TP_BHH = 576,
                  TP_LN3W = 768, TP_LN3B = 832, TP_HEADB = 896, TP_FLOATS = 912;
    static_assert(TP_FLOATS == AcShared<RT, NW>::TP_FLOATS, "tail parameter vectors");
    auto& s_tp = sh.tp;

    // XCD-aware placement (speed only, any placement is correct): workgroup b runs on XCD b % 8 and every (net, actor|critic)
    // pair has its own 0.64 MB of fc1 weights.  In launch order a pair's row tiles land on all 8 XCDs and every XCD's 4 MB
    // L2 sees all 10 pairs (6.4 MB: the fragments come from the Infinity Cache instead, profiles/r02d: 20 MB fetched per
    // rollout launch for 9.6 MB of unique bytes).  The remap hands XCD k a CONTIGUOUS range of the pair-major work list.
    int bx = gp.bx, by = gp.by, bz = gp.bz;
#ifndef AC_NO_XCD_REMAP
    {
        const int X = gp.gx, G = X * gp.gy * gp.gz;
        const int b = bx + X * (by + gp.gy * bz);
        const int k = b & 7, slot = b >> 3, q8 = G >> 3, r8 = G & 7;
        const int item = k * q8 + (k < r8 ? k : r8) + slot;                        // bijective: XCD k owns q8 (+1 if k < r8) items
        bx = item % X;
        const int pair = item / X;
        by = pair % gp.gy;
        bz = pair / gp.gy;
    }
#endif
    const int net = by;
    const int which = a.which == 2 ? bz : a.which;                    // 0 actor, 1 critic
    const IplanAcNet& nw = which ? a.critic : a.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    const IplanAcFeatures& ft = a.feat;
    const int l = lane_id(), w = uniform_i(wave_id()), n = l & 15, g = l >> 4;
    const int ks = RT == 1 ? a.ksplit : 1;
    const int groups = NW / ks;
    const int part = w % ks;
    // Rollout shape, K split over KW WORKGROUPS as well (a.ksplit_wg): 2 row tiles x 10 nets are 20 workgroups on 256 CUs and
    // the F-wide contraction is a latency chain of ~20 k-tiles per wave; with KW = 4 every wave owns ~5 and the partial sums
    // of a (tile, net) unit meet in global memory -- the LAST of its KW workgroups to arrive (atomic ticket) adds them in
    // workgroup order and runs the tail.  Fixed summation order, so the result does not depend on who arrives last.
    const int KW = (RT == 1 && !PRE && a.ksplit_wg > 1 && ks == 8) ? a.ksplit_wg : 1;
    const int pw = bx % KW;
    bx /= KW;
    const int kpart = part * KW + pw;
    const bool clk = !FUSED && a.phase_clocks && bx == 0 && by == 0 && bz == 0 && threadIdx.x == 0;
    if (clk) a.phase_clocks[0] = IPLAN_CLOCK();
    // fused launch (diagnostics): 8 clocks per actor/critic workgroup -- entry, before / after the producers' wait, contraction
    // done, partial sums exchanged (last arrival only), end of the tail
    int64_t* fclk = (FUSED && a.phase_clocks && threadIdx.x == 0) ? a.phase_clocks + 8 * (gp.bx + gp.gx * (gp.by + gp.gy * gp.bz)) : nullptr;
    if (fclk) fclk[0] = IPLAN_CLOCK();
    const KMap km = make_kmap(ft);
    const int F = km.NW + km.n_actions + km.n_id;
    const int KT = km.kt0[4];
    // pre-packed fc1 operands (iplan_ac_pack_fc1): fragment-major weights, k-ordered LayerNorm(F) parameters
    const float* __restrict__ pkw = which ? a.packed_critic : a.packed_actor;
    if (pkw) pkw += (int64_t)net * a.packed_s_net;
    const float* __restrict__ pkg = pkw ? pkw + (int64_t)KT * 1024 : nullptr;
    const float* __restrict__ pkb = pkw ? pkg + (int64_t)KT * 16 : nullptr;
    const float* __restrict__ pkc = pkw ? pkb + (int64_t)KT * 16 : nullptr;      // (W gamma)[64] | (W beta)[64]
    // k-tiles are dealt round-robin to the ks cooperating waves (tile T belongs to wave T % ks): the slow tiles
    // (the gathered history block) are spread evenly instead of landing on one straggler wave
    const int T_lo = kpart, T_hi = KT, T_st = ks * KW;

    int rr[RT], last[RT];
    bool vld[RT];
    int64_t prr[RT];
    const float* src[RT][3];
    for (int t = 0; t < RT; ++t) {
        const int tile = (bx * groups + w / ks) * RT + t;
        rr[t] = tile * 16 + n;
        vld[t] = rr[t] < a.rows;
        prr[t] = vld[t] ? (int64_t)(rr[t] / ft.T) * ft.T_phys + (rr[t] % ft.T) : 0;
        for (int s = 0; s < 3; ++s)
            src[t][s] = ft.w[s] > 0 ? ft.src[s] + (int64_t)net * ft.s_net[s] + prr[t] * ft.s_row[s] : nullptr;
        last[t] = -1;
        if (vld[t] && ft.n_actions > 0) {
            if (ft.last_action) last[t] = ft.last_action[(int64_t)net * ft.la_s_net + prr[t] * ft.la_s_row];
            else if (ft.last_action64) last[t] = (int)ft.last_action64[(int64_t)net * ft.la64_s_net + prr[t] * ft.la64_s_row];
        }
    }

    const bool staged = STAGED_BUILD && ks == 8 && groups == 1 && a.saved == nullptr;    // (training launches keep the gate records)
    if (STAGED_BUILD && staged) {
        // (a) 4352 16-byte chunks of [fc2.weight 64x64 | rnn.weight_ih 192x64 | head n_out x 64 (zero padded to 16 rows)]
        const float* Wsrc[3] = {P + nw.off[IPLAN_AC_FC2_W], P + nw.off[IPLAN_AC_WIH], P + nw.off[IPLAN_AC_HEAD_W]};
        for (int c = (int)threadIdx.x; c < (AM + 3 * AM + 16) * 16; c += 64 * NW) {
            const int r = c >> 4, c4 = c & 15;
            f32x4 v = splat4(0.f);
            if (r < AM) v = *reinterpret_cast<const f32x4*>(Wsrc[0] + r * AM + 4 * c4);
            else if (r < 4 * AM) v = *reinterpret_cast<const f32x4*>(Wsrc[1] + (r - AM) * AM + 4 * c4);
            else if (r - 4 * AM < nw.n_out) v = *reinterpret_cast<const f32x4*>(Wsrc[2] + (r - 4 * AM) * AM + 4 * c4);
            *reinterpret_cast<f32x4*>(&s_tw[r * TLDW + 4 * c4]) = v;
        }
        // (b) gh tile t = W_hh[16 t .. 16 t + 15] h + b_hh: tiles w and w + 8 of the 12
        const float* hsrc0 = which ? a.h_critic : a.h_actor;
        const float* hrow0 = hsrc0 + (int64_t)net * a.hs_net + prr[0] * a.hs_row;
        f32x4 h0[AT];
        for (int t = 0; t < AT; ++t) h0[t] = vload(hrow0, vld[0], AM, t);
        for (int t = w; t < 3 * AT; t += NW)
            s_gh[t][l] = dense_tile_ga<AT>(P + nw.off[IPLAN_AC_WHH], AM, 3 * AM, 16 * t, h0, bfrag_a(P + nw.off[IPLAN_AC_BHH], t));
    }

    if (STAGED2) {      // 7424 16-byte chunks: [fc2 64 | rnn.weight_ih 192 | head n_out (zero padded to 16) | rnn.weight_hh 192] x 64
        const float* Wsrc[4] = {P + nw.off[IPLAN_AC_FC2_W], P + nw.off[IPLAN_AC_WIH], P + nw.off[IPLAN_AC_HEAD_W], P + nw.off[IPLAN_AC_WHH]};
        for (int c = (int)threadIdx.x; c < TW_ROWS * 16; c += 64 * NW) {
            const int r = c >> 4, c4 = c & 15;
            f32x4 v = splat4(0.f);
            if (r < AM) v = *reinterpret_cast<const f32x4*>(Wsrc[0] + r * AM + 4 * c4);
            else if (r < 4 * AM) v = *reinterpret_cast<const f32x4*>(Wsrc[1] + (r - AM) * AM + 4 * c4);
            else if (r < 4 * AM + 16) { if (r - 4 * AM < nw.n_out) v = *reinterpret_cast<const f32x4*>(Wsrc[2] + (r - 4 * AM) * AM + 4 * c4); }
            else v = *reinterpret_cast<const f32x4*>(Wsrc[3] + (r - 4 * AM - 16) * AM + 4 * c4);
            *reinterpret_cast<f32x4*>(&s_tw[r * TLDW + 4 * c4]) = v;
        }
    }

    const bool lds_tail = STAGED2 || (STAGED_BUILD && staged);
    if (TW_ROWS && lds_tail) {
        const int tp_off[11] = {TP_FC1B, TP_LN1W, TP_LN1B, TP_FC2B, TP_LN2W, TP_LN2B, TP_BIH, TP_BHH, TP_LN3W, TP_LN3B, TP_HEADB};
        const int tp_src[11] = {IPLAN_AC_FC1_B, IPLAN_AC_LN1_W, IPLAN_AC_LN1_B, IPLAN_AC_FC2_B, IPLAN_AC_LN2_W, IPLAN_AC_LN2_B, IPLAN_AC_BIH,
                                IPLAN_AC_BHH, IPLAN_AC_LN3_W, IPLAN_AC_LN3_B, IPLAN_AC_HEAD_B};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const int len = k == 10 ? 16 : ((k == 6 || k == 7) ? 3 * AM : AM), have = k == 10 ? nw.n_out : len;
            for (int i = (int)threadIdx.x; i < len; i += 64 * NW) s_tp[tp_off[k] + i] = i < have ? P[nw.off[tp_src[k]] + i] : 0.f;
        }
    }

    // ---- LayerNorm(F) statistics, two passes (mean, then centred second moment) over L2-resident rows
    // Rollout shape (a few row tiles per net, K split over the 8 waves): ONE pass instead -- the contraction accumulates
    //   P1 = W (gamma o x) on MFMA,  W gamma and W beta (lane-local dot products with the same W fragments),  sum x,  sum x^2
    // and  fc1(LN(x)) = rstd (P1 - mu W gamma) + W beta  is finished after the cross-wave reduction (no statistics pre-pass
    // over the row: it cost a third of the launch, all of it load latency).
    const bool fold = !PRE && RT == 1 && ks > 1 && a.ln_stats_mode == 0;
    float mu[RT], rstd[RT];
    float fsx = 0.f, fsxx = 0.f;
    float c1a[AT], c2a[AT];                                   // lane-local partial (W gamma)[16 oo + n], (W beta)[16 oo + n]
    for (int o = 0; o < AT; ++o) { c1a[o] = 0.f; c2a[o] = 0.f; }
    if (fold) {
        for (int t = 0; t < RT; ++t) { mu[t] = 0.f; rstd[t] = 1.f; }
    } else if (PRE || a.ln_stats_mode == 2) {
        for (int t = 0; t < RT; ++t) {
            mu[t] = 0.f; rstd[t] = 0.f;
            if (vld[t]) {
                const float* st = a.ln_stats + (int64_t)net * a.ln_stats_s_net + prr[t] * 2;
                mu[t] = st[0]; rstd[t] = st[1];
            }
        }
    } else {
        float s[RT];
        for (int t = 0; t < RT; ++t) s[t] = 0.f;
        for (int T = T_lo; T < T_hi; T += T_st) {
            const KTile kt = ktile(km, T);
            for (int t = 0; t < RT; ++t) {
                const f32x4 x = kfeat<FUSED>(km, kt, src[t], vld[t], last[t], net);
                s[t] += (x[0] + x[1]) + (x[2] + x[3]);
            }
        }
        for (int t = 0; t < RT; ++t) s[t] = group_sum(s[t]);
        if (ks > 1) {
            if (g == 0) s_red[w][n] = s[0];
            __syncthreads();
            float m = 0.f;
            for (int p2 = 0; p2 < ks; ++p2) m += s_red[(w / ks) * ks + p2][n];
            s[0] = m;
            __syncthreads();
        }
        for (int t = 0; t < RT; ++t) mu[t] = s[t] / (float)F;
        float v2[RT];
        for (int t = 0; t < RT; ++t) v2[t] = 0.f;
        for (int T = T_lo; T < T_hi; T += T_st) {
            const KTile kt = ktile(km, T);
            for (int t = 0; t < RT; ++t) {
                const f32x4 x = kfeat<FUSED>(km, kt, src[t], vld[t], last[t], net);
                for (int q = 0; q < 4; ++q)
                    if (q < kt.nv) { const float d = x[q] - mu[t]; v2[t] = fmaf(d, d, v2[t]); }
            }
        }
        for (int t = 0; t < RT; ++t) v2[t] = group_sum(v2[t]);
        if (ks > 1) {
            if (g == 0) s_red[w][n] = v2[0];
            __syncthreads();
            float m = 0.f;
            for (int p2 = 0; p2 < ks; ++p2) m += s_red[(w / ks) * ks + p2][n];
            v2[0] = m;
        }
        for (int t = 0; t < RT; ++t) {
            rstd[t] = 1.0f / sqrtf(v2[t] / (float)F + 1e-5f);
            if (a.ln_stats_mode == 1 && vld[t] && g == 0 && part == 0) {
                float* st = a.ln_stats + (int64_t)net * a.ln_stats_s_net + prr[t] * 2;
                st[0] = mu[t]; st[1] = rstd[t];
            }
        }
    }

    if (clk) a.phase_clocks[1] = IPLAN_CLOCK();
    // ---- fc1 contraction over this wave's share of K
    const float* fnw = P + nw.off[IPLAN_AC_FN_W];
    const float* fnb = P + nw.off[IPLAN_AC_FN_B];
    const float* W1 = P + nw.off[IPLAN_AC_FC1_W];
    f32x4 accs[RT][AT];
    for (int t = 0; t < RT; ++t)
        for (int o = 0; o < AT; ++o) accs[t][o] = splat4(0.f);
    // Streaming form (a whole row's K = F in ONE wave: compute_returns' pass over every stored step, the minibatch / fp32 epochs):
    // F = 2485 is 622 dependent fp32 accumulations per output -- one rounding each, 2.8e-6 of the pre-activation's scale against
    // 1e-6 for a blocked contraction (VERDICT r5 "weak" 1).  The k-tiles therefore accumulate into a GROUP accumulator that is added
    // to the running sum every AC_GROUP_ROUNDS ring rounds and at the end of every source block: chains of <= 24 MFMAs + <= 30 adds.
    // (The rollout form splits K over 8 waves x 4 workgroups: <= 5 k-tiles per chain already.)
#ifdef AC_NO_GROUPS                     // A/B builds: one accumulation chain over the whole row (the form until round 6)
    constexpr bool GROUPED = false;
#else
    constexpr bool GROUPED = !PRE && RT > 1;
#endif
    f32x4 blk[GROUPED ? RT : 1][GROUPED ? AT : 1];
    if (GROUPED)
        for (int t = 0; t < RT; ++t)
            for (int o = 0; o < AT; ++o) blk[GROUPED ? t : 0][GROUPED ? o : 0] = splat4(0.f);
    auto flush_group = [&]() {
        if constexpr (GROUPED)
            for (int t = 0; t < RT; ++t)
                for (int o = 0; o < AT; ++o) { accs[t][o] += blk[t][o]; blk[t][o] = splat4(0.f); }
    };
    auto acc_mma = [&](const f32x4& wfrag, const f32x4& xfrag, int t, int oo) {
        if constexpr (GROUPED) blk[t][oo] = mma_block(wfrag, xfrag, blk[t][oo]);
        else accs[t][oo] = mma_block(wfrag, xfrag, accs[t][oo]);
    };
    // Software pipeline: the operands of k-tile T + PF are requested while k-tile T's MFMAs issue.  The weight rows
    // (636 KB per net) and the feature rows come from L2 / HBM with ~1-2 us latency; un-pipelined, every k-tile paid
    // that latency in full (profiles/: 2.1 us per k-tile in the rollout variant).
    struct KOps {
        KTile kt;
        f32x4 gm, bt, x[RT], wf[AT];
    };
    auto kload = [&](int T, KOps& o) {
        o.kt = ktile(km, T);
        for (int t = 0; t < RT; ++t) o.x[t] = kfeat<FUSED>(km, o.kt, src[t], vld[t], last[t], net);
        if (pkw) {
            o.gm = *reinterpret_cast<const f32x4*>(pkg + T * 16 + 4 * g);
            o.bt = *reinterpret_cast<const f32x4*>(pkb + T * 16 + 4 * g);
            for (int oo = 0; oo < AT; ++oo) o.wf[oo] = *reinterpret_cast<const f32x4*>(pkw + ((int64_t)(T * AT + oo) * 64 + l) * 4);
            return;
        }
        o.gm = kcols(o.kt, fnw);
        o.bt = kcols(o.kt, fnb);
        for (int oo = 0; oo < AT; ++oo) o.wf[oo] = kcols(o.kt, W1 + (int64_t)(16 * oo + n) * F);
    };
    // folded variant of the normalisation: B operand = gamma o x (plus the [gamma beta] columns), statistics on the side
    auto fold_mma = [&](const f32x4& x, const f32x4& gm, const f32x4& bt, const f32x4 (&wf)[AT], int nv) {
        f32x4 xg, gz, bz;
        for (int q = 0; q < 4; ++q) {
            const bool ok = q < nv;
            const float xv = (vld[0] && ok) ? x[q] : 0.f;
            xg[q] = ok ? xv * gm[q] : 0.f;
            fsx += xv;
            fsxx = fmaf(xv, xv, fsxx);
            gz[q] = ok ? gm[q] : 0.f;
            bz[q] = ok ? bt[q] : 0.f;
        }
        for (int oo = 0; oo < AT; ++oo) {
            accs[0][oo] = mma_block(wf[oo], xg, accs[0][oo]);
            if (!pkc)                                          // (packed operands carry W gamma, W beta precomputed)
                for (int q = 0; q < 4; ++q) {                  // wf[oo] = W[16 oo + n][4g .. 4g+3] of this k-tile
                    c1a[oo] = fmaf(wf[oo][q], gz[q], c1a[oo]);
                    c2a[oo] = fmaf(wf[oo][q], bz[q], c2a[oo]);
                }
        }
    };
    auto kmma = [&](const KOps& o) {
        if (fold) { fold_mma(o.x[0], o.gm, o.bt, o.wf, o.kt.nv); return; }
        f32x4 xn[RT];
        for (int t = 0; t < RT; ++t)
            for (int q = 0; q < 4; ++q)
                xn[t][q] = (vld[t] && q < o.kt.nv) ? (o.x[t][q] - mu[t]) * rstd[t] * o.gm[q] + o.bt[q] : 0.f;
        for (int oo = 0; oo < AT; ++oo)
            for (int t = 0; t < RT; ++t) acc_mma(o.wf[oo], xn[t], t, oo);
    };
    // Fast tiles: whole 16-column tiles inside a source block whose width is a multiple of 4 (attention 32,
    // behaviour 8: 137 of the 157 k-tiles at Highway chaotic).  Their operand fetch has NO per-lane control flow --
    // one unaligned 16-byte load per operand -- so the ring below really keeps PF k-tiles of loads in flight (with
    // divergent branches around the loads the compiler has to drain vmcnt at every join).
    struct FOps {
        f32x4 gm, bt, x[RT], wf[AT];
    };
    auto fload = [&](int T, int s, FOps& o) {
        const int f0 = 16 * (T - km.kt0[s]) + 4 * g;
        // (fused launch, latent blocks: device-coherent and FIRST in the slot -- the fragment loads below are what the compiler waits for)
        for (int t = 0; t < RT; ++t) o.x[t] = (FUSED && s > 0) ? coh_load16_untracked(src[t][s] + f0) : ldu4(src[t][s] + f0);
        if (pkw) {                                          // one contiguous 1 KiB block per fragment, 64 B for gamma / beta
            o.gm = *reinterpret_cast<const f32x4*>(pkg + T * 16 + 4 * g);
            o.bt = *reinterpret_cast<const f32x4*>(pkb + T * 16 + 4 * g);
            for (int oo = 0; oo < AT; ++oo) o.wf[oo] = *reinterpret_cast<const f32x4*>(pkw + ((int64_t)(T * AT + oo) * 64 + l) * 4);
            return;
        }
        const int wS = km.w[s];
        const int e = f0 / wS;
        const int c0 = e * km.W + km.off[s] + (f0 - e * wS);
        o.gm = ldu4(fnw + c0);
        o.bt = ldu4(fnb + c0);
        for (int oo = 0; oo < AT; ++oo) o.wf[oo] = ldu4(W1 + (int64_t)(16 * oo + n) * F + c0);
    };
    auto fmma = [&](const FOps& o) {
        // fused launch: o.x may come from coh_load16_untracked, which the compiler's wait counts do not see -- tie it to the LAST tracked
        // load of the same ring slot (loads return in order), so that its first use cannot be scheduled in front of that load's wait
        f32x4 ox[RT];
        for (int t = 0; t < RT; ++t) ox[t] = FUSED ? after_load(o.x[t], o.wf[AT - 1][3]) : o.x[t];
        if (fold) { fold_mma(ox[0], o.gm, o.bt, o.wf, 4); return; }
        f32x4 xn[RT];
        for (int t = 0; t < RT; ++t)
            for (int q = 0; q < 4; ++q) xn[t][q] = vld[t] ? (ox[t][q] - mu[t]) * rstd[t] * o.gm[q] + o.bt[q] : 0.f;
        for (int oo = 0; oo < AT; ++oo)
            for (int t = 0; t < RT; ++t) acc_mma(o.wf[oo], xn[t], t, oo);
    };
#ifndef AC_PF                           // ring depth of the streaming form: 2 slots of 32 registers (3 until round 6: the third slot's
#define AC_PF 2                         // registers now hold the group accumulator -- a 512-thread workgroup caps a wave at 256)
#endif
#ifndef AC_GROUP_ROUNDS
#define AC_GROUP_ROUNDS 3
#endif
#ifndef AC_PF1                          // ring depth of the rollout form (one row tile per wave: 28 registers per slot)
#define AC_PF1 3
#endif
    constexpr int PF = RT == 1 ? AC_PF1 : AC_PF;
    if (PRE) {
        for (int t = 0; t < RT; ++t) {
            const float* zrow = a.fc1_pre + (((int64_t)which * a.n_agents + net) * a.rows + (vld[t] ? rr[t] : 0)) * AM;
            for (int o = 0; o < AT; ++o) accs[t][o] = vload_a(zrow, vld[t], o);
            // (small batches: the K loop of the split contraction ran in fc1_pre_parts workgroups -- their parts, in order)
            const int64_t zpart = 2 * (int64_t)a.n_agents * a.rows * AM;
            for (int pp = 1; pp < a.fc1_pre_parts; ++pp)
                for (int o = 0; o < AT; ++o) accs[t][o] += vload_a(zrow + pp * zpart, vld[t], o);
        }
    }
    for (int s = 0; s < (PRE ? 0 : 4); ++s) {
        // fused rollout launch (gat.hip: gat_enc_ac_fwd_kernel): blocks 1 / 2 -- the attention and behaviour latents -- are written
        // by the GAT scenes and encoder tiles of THIS launch; everything above (tail weights, W_hh h, the history block) ran beside them
        if (FUSED && s == 1) {
            if (fclk) fclk[1] = IPLAN_CLOCK();
            ac_wait_producers(prod);
            if (fclk) fclk[2] = IPLAN_CLOCK();
        }
        // this wave's tiles of block s: T in [b_lo, b_hi) with T % T_st == part; the fast ones are below f_hi
        // (km's arrays are indexed by the loop counter and live in scratch: what comes back is a VGPR, and loop bounds in
        // VGPRs make every branch below a divergent one -- exec-masked loads, wait counters drained at each join.  Through
        // SGPRs the ring keeps its PF k-tiles of loads in flight.)
        const int kt0s = uniform_i(km.kt0[s]);
        const int b_end = imin(T_hi, uniform_i(km.kt0[s + 1]));
        const int b_lo = kt0s + ((kpart - kt0s) % T_st + T_st) % T_st;              // first owned tile of the block
        if (b_lo >= b_end) continue;
        int f_hi = b_lo;
        if (s < 3 && (uniform_i(km.w[s]) & 3) == 0 && uniform_i(ft.w[s]) > 0) f_hi = imin(b_end, kt0s + uniform_i(km.len[s]) / 16);
        int T_slow = b_lo;
        if (f_hi > b_lo) {
            FOps ring[PF];
            int T = b_lo;
            // Steady state: while a whole round of PF tiles AND their PF successors exist, every load of the round is
            // unconditional -- only then does the number of loads in flight not depend on the path taken, and only then can
            // the compiler wait for the OLDEST ring slot alone (vmcnt(2 slots)) instead of draining everything (vmcnt(0))
            // at the top of each round.  The ragged end of the block runs through the guarded round below.
            if (b_lo + (2 * PF - 1) * T_st < f_hi) {
                for (int i = 0; i < PF; ++i) fload(b_lo + i * T_st, s, ring[i]);
                IPLAN_SCHED_FENCE();
                int rounds = 0;
                for (; T + (2 * PF - 1) * T_st < f_hi; T += PF * T_st) {
                    for (int i = 0; i < PF; ++i) {
                        fmma(ring[i]);
                        IPLAN_SCHED_FENCE();
                        fload(T + (i + PF) * T_st, s, ring[i]);
                        IPLAN_SCHED_FENCE();
                    }
                    if (GROUPED && ++rounds == AC_GROUP_ROUNDS) { flush_group(); rounds = 0; }
                }
            } else {
                for (int i = 0; i < PF; ++i)
                    if (b_lo + i * T_st < f_hi) fload(b_lo + i * T_st, s, ring[i]);
            }
            IPLAN_SCHED_FENCE();                            // keep the prefetches where they are: the scheduler would
            for (; T < f_hi; T += PF * T_st) {              // otherwise sink every load next to its use
                for (int i = 0; i < PF; ++i) {
                    if (T + i * T_st < f_hi) {
                        fmma(ring[i]);
                        IPLAN_SCHED_FENCE();
                        if (T + (i + PF) * T_st < f_hi) fload(T + (i + PF) * T_st, s, ring[i]);
                        IPLAN_SCHED_FENCE();
                    }
                }
            }
            T_slow = b_lo + ((f_hi - b_lo + T_st - 1) / T_st) * T_st;              // first owned tile at or past f_hi
#if !defined(IPLAN_HOST_EMULATION)
            // every ring slot has been consumed here (its wait covered the slot's untracked load: fmma); the explicit drain makes that hold
            // on EVERY control-flow path the compiler emitted, feasible or not, before the ring's registers are reused
            // (scripts/check_untracked_load_waits.py walks the paths) -- nothing is in flight at this point, it costs nothing
            if (FUSED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
        for (int T = T_slow; T < b_end; T += T_st) {        // ragged / gathered / one-hot tiles
            KOps o;
            kload(T, o);
            kmma(o);
        }
        flush_group();
    }
    if (fclk) fclk[3] = IPLAN_CLOCK();
    if (ks > 1) {
        for (int t = 0; t < AT; ++t) s_acc[w][t][l] = accs[0][t];
        if (fold) {                                           // W gamma, W beta and the row statistics of the 8 waves: parked with the
            fsx = group_sum(fsx);                             // partial tiles, ONE barrier for both (it sits on the launch's critical path)
            fsxx = group_sum(fsxx);
            if (g == 0) { s_red[w][n] = fsx; s_red2[w][n] = fsxx; }
            if (!pkc)
                for (int t = 0; t < AT; ++t) {
                    const float c1 = group_sum(c1a[t]), c2 = group_sum(c2a[t]);
                    if (g == 0) { s_cc[w][0][16 * t + n] = c1; s_cc[w][1][16 * t + n] = c2; }
                }
        }
        __syncthreads();
        if (part == 0) {
            for (int t = 0; t < AT; ++t) {
                f32x4 sum = s_acc[w][t][l];
                for (int p2 = 1; p2 < ks; ++p2) sum += s_acc[w + p2][t][l];
                accs[0][t] = sum;
            }
        }
        if (fold) {
            if (part == 0) {
                float sx = 0.f, sxx = 0.f;
                for (int p2 = 0; p2 < ks; ++p2) { sx += s_red[w + p2][n]; sxx += s_red2[w + p2][n]; }
                if (KW > 1) {
                    const int unit = (bz * gp.gy + by) * (gp.gx / KW) + bx;
                    float* slot = a.ks_scratch + ((int64_t)unit * KW + pw) * IPLAN_AC_KS_SLOT_FLOATS;
                    // our partial out, a ticket, the others' partials in.  The workgroups of a unit may sit on different XCDs, i.e.
                    // behind different L2s: the 4 KiB that cross are written and read device-coherently (wave_tile.h: coh_store /
                    // coh_load) and the stores are drained before the ticket is taken -- round 3's form, an agent-scope release /
                    // acquire pair around plain accesses, is an L2 write-back and an L2 invalidate per workgroup on the launch's
                    // critical path (IPLAN_FUSED_FENCES=1 builds keep it)
                    for (int t = 0; t < AT; ++t) coh_store16(slot + (t * 64 + l) * 4, accs[0][t]);
                    if (g == 0) { coh_store(slot + AT * 256 + n, sx); coh_store(slot + AT * 256 + 16 + n, sxx); }
                    int ticket = 0;
#ifdef IPLAN_HOST_EMULATION
                    IPLAN_WAVE_SYNC();
                    if (l == 0) { ticket = a.ks_count[unit]; a.ks_count[unit] = ticket + 1 == KW ? 0 : ticket + 1; s_red[0][0] = (float)ticket; }
                    IPLAN_WAVE_SYNC();
                    ticket = (int)s_red[0][0];
#else
#if IPLAN_FUSED_FENCES
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (l == 0) ticket = __hip_atomic_fetch_add(a.ks_count + unit, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (l == 0) ticket = __hip_atomic_fetch_add(a.ks_count + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                    ticket = __builtin_amdgcn_readfirstlane(ticket);
#endif
                    if (ticket != KW - 1) {                       // (the tail belongs to the last arrival)
                        if (STAGED_BUILD && staged) {             // the other waves wait at the tail's hand-over barrier: send them home
                            if (l == 0) sh.hand[0] = 0.f;
#ifndef IPLAN_HOST_EMULATION
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                        }
                        return;
                    }
#ifndef IPLAN_HOST_EMULATION
                    if (l == 0) __hip_atomic_store(a.ks_count + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#if IPLAN_FUSED_FENCES
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
#endif
                    sx = 0.f; sxx = 0.f;
                    for (int t = 0; t < AT; ++t) accs[0][t] = splat4(0.f);
                    for (int q2 = 0; q2 < KW; ++q2) {                // (one load instruction per tile; the slot's statistics -- loaded
                        const float* sl = a.ks_scratch + ((int64_t)unit * KW + q2) * IPLAN_AC_KS_SLOT_FLOATS;   // after them -- carry the wait)
                        f32x4 pt[AT];
                        for (int t = 0; t < AT; ++t) pt[t] = coh_load16_untracked(sl + (t * 64 + l) * 4);
                        const float psx = coh_load(sl + AT * 256 + n), psxx = coh_load(sl + AT * 256 + 16 + n);
                        for (int t = 0; t < AT; ++t) accs[0][t] += after_load(pt[t], psxx);
                        sx += psx;
                        sxx += psxx;
                    }
                }
                mu[0] = sx / (float)F;
                rstd[0] = 1.0f / sqrtf(fmaxf(sxx / (float)F - mu[0] * mu[0], 0.f) + 1e-5f);
                for (int t = 0; t < AT; ++t)
                    for (int q = 0; q < 4; ++q) {
                        const int o = 16 * t + 4 * g + q;
                        float c1 = 0.f, c2 = 0.f;
                        if (pkc) { c1 = pkc[o]; c2 = pkc[AM + o]; }
                        else for (int p2 = 0; p2 < ks; ++p2) { c1 += s_cc[w + p2][0][o]; c2 += s_cc[w + p2][1][o]; }
                        accs[0][t][q] = rstd[0] * (accs[0][t][q] - mu[0] * c1) + c2;
                    }
            }
        }
    }
    if (clk) a.phase_clocks[2] = IPLAN_CLOCK();
    if (fclk) fclk[4] = IPLAN_CLOCK();
    if (STAGED2) __syncthreads();                               // the staged tail weights (every wave gets here: ks == 1)
    // Rollout shape with the tail staged in LDS: the 64-wide tail is the launch's critical path (one row tile per workgroup, 11 us as a
    // one-wave chain: scripts/dev/fused_step_clocks.py) -- FOUR waves run it, one 16-wide output tile each (fc2, then the three GRU
    // gate tiles of the same hidden units), everything lane-local is computed redundantly, and the tiles meet twice in LDS.
    const bool ptail = STAGED_BUILD && staged;
    if (STAGED_BUILD && ptail) {
        if (part == 0) {
            for (int t = 0; t < AT; ++t) s_acc[0][t][l] = accs[0][t];
            if (l == 0) sh.hand[0] = 1.f;
        }
        __syncthreads();
        if (sh.hand[0] == 0.f || w >= AT) return;
        for (int t = 0; t < AT; ++t) accs[0][t] = s_acc[0][t][l];
    } else if (part != 0) return;

  const bool act_tanh = a.act_tanh != 0;                       // (uniform) MLPBase activation: tanh instead of ReLU (args.use_ReLU off)
  for (int rt = 0; rt < RT; ++rt) {
    const int r = rr[rt];
    const bool valid = vld[rt];
    const int64_t pr = prr[rt];
    const float mu_ = mu[rt], rstd_ = rstd[rt];
    f32x4 acc[AT];
    for (int t = 0; t < AT; ++t) acc[t] = accs[rt][t];
    if (RT > 1 && (r - n) >= a.rows) break;                       // whole row tile beyond the end
    // ---- 64-wide tail, one wave per row tile
    float* sv = a.saved ? a.saved + (((int64_t)which * a.n_agents + net) * a.rows + (valid ? r : 0)) * IPLAN_AC_SAVE_FLOATS : nullptr;
    float mu1 = 0.f, rs1 = 0.f, mu2 = 0.f, rs2 = 0.f, mu3, rs3;
    // tile t of a parameter vector: LDS copy (tail staged) or the arena (two loads in two branches, never a mixed pointer)
    auto pv = [&](int tp, int which_vec, int t) { return (TW_ROWS && lds_tail) ? bfrag_a(s_tp + tp, t) : bfrag_a(P + nw.off[which_vec], t); };
    // the row's stored hidden state is fetched first: it lands while fc1's epilogue, fc2 and two LayerNorms run
    const float* hsrc = which ? a.h_critic : a.h_actor;
    const float* hrow = hsrc + (int64_t)net * a.hs_net + pr * a.hs_row;
    f32x4 h[AT], hnew[AT];
    f32x4 f[AT], gmv[AT], btv[AT];
    if (STAGED_BUILD && ptail) {
        // ---- four waves, wave w owns output tile w (sv == nullptr here: nothing is recorded in the rollout shape)
        const f32x4 hw = vload(hrow, valid, AM, w);
        for (int t = 0; t < AT; ++t) f[t] = ac_act4(acc[t] + bfrag_a(s_tp + TP_FC1B, t), act_tanh);
        for (int t = 0; t < AT; ++t) { gmv[t] = bfrag_a(s_tp + TP_LN1W, t); btv[t] = bfrag_a(s_tp + TP_LN1B, t); }
        layer_norm_tiles_f<AT>(f, gmv, btv, &mu1, &rs1);
        s_acc[RT == 1 ? 1 : 0][w][l] = ac_act4(dense_tile<AT>(s_tw, TLDW, 16 * w, f, bfrag_a(s_tp + TP_FC2B, w)), act_tanh);
        __syncthreads();
        f32x4 f2p[AT];
        for (int t = 0; t < AT; ++t) f2p[t] = s_acc[RT == 1 ? 1 : 0][t][l];
        for (int t = 0; t < AT; ++t) { gmv[t] = bfrag_a(s_tp + TP_LN2W, t); btv[t] = bfrag_a(s_tp + TP_LN2B, t); }
        layer_norm_tiles_f<AT>(f2p, gmv, btv, &mu2, &rs2);
        const float* sWi = s_tw + AM * TLDW;
        const f32x4 pr_s = dense_tile<AT>(sWi, TLDW, 16 * w, f2p, bfrag_a(s_tp + TP_BIH, w) + s_gh[w][l]);
        const f32x4 pz_s = dense_tile<AT>(sWi, TLDW, AM + 16 * w, f2p, bfrag_a(s_tp + TP_BIH, AT + w) + s_gh[AT + w][l]);
        const f32x4 gn_s = dense_tile<AT>(sWi, TLDW, 2 * AM + 16 * w, f2p, bfrag_a(s_tp + TP_BIH, 2 * AT + w));
        s_acc[RT == 1 ? 2 : 0][w][l] = gru_gates(pr_s, pz_s, gn_s, s_gh[2 * AT + w][l], hw).h;
        __syncthreads();
        if (w != 0) return;                                           // LayerNorm, head and the draw: one wave
        for (int t = 0; t < AT; ++t) hnew[t] = s_acc[RT == 1 ? 2 : 0][t][l];
    } else {
    for (int t = 0; t < AT; ++t) h[t] = vload(hrow, valid, AM, t);
    for (int t = 0; t < AT; ++t) {
        f[t] = ac_act4(acc[t] + pv(TP_FC1B, IPLAN_AC_FC1_B, t), act_tanh);
        if (sv) vstore(sv, valid, AM, t, f[t]);                       // a1
    }
    for (int t = 0; t < AT; ++t) { gmv[t] = pv(TP_LN1W, IPLAN_AC_LN1_W, t); btv[t] = pv(TP_LN1B, IPLAN_AC_LN1_B, t); }
    layer_norm_tiles_f<AT>(f, gmv, btv, &mu1, &rs1);
    if (sv) for (int t = 0; t < AT; ++t) vstore(sv + AM, valid, AM, t, f[t]);   // f1
    f32x4 f2[AT];
    for (int t = 0; t < AT; ++t) {
        if (lds_tail) f2[t] = ac_act4(dense_tile<AT>(s_tw, TLDW, 16 * t, f, pv(TP_FC2B, IPLAN_AC_FC2_B, t)), act_tanh);
        else f2[t] = ac_act4(dense_tile_ga<AT>(P + nw.off[IPLAN_AC_FC2_W], AM, AM, 16 * t, f, bfrag_a(P + nw.off[IPLAN_AC_FC2_B], t)), act_tanh);
        if (sv) vstore(sv + 2 * AM, valid, AM, t, f2[t]);             // a2
    }
    for (int t = 0; t < AT; ++t) { gmv[t] = pv(TP_LN2W, IPLAN_AC_LN2_W, t); btv[t] = pv(TP_LN2B, IPLAN_AC_LN2_B, t); }
    layer_norm_tiles_f<AT>(f2, gmv, btv, &mu2, &rs2);
    if (sv) for (int t = 0; t < AT; ++t) vstore(sv + 3 * AM, valid, AM, t, f2[t]);  // f2
    // GRU step (rnn.py:24-27) from the stored hidden state
    {
        const float* Wi = P + nw.off[IPLAN_AC_WIH];
        const float* Wh = P + nw.off[IPLAN_AC_WHH];
        for (int t = 0; t < AT; ++t) {
            if (STAGED_BUILD && staged) {                    // W_ih from LDS, W_hh h + b_hh precomputed by the 8 waves
                const float* sWi = s_tw + AM * TLDW;
                const f32x4 pr_s = dense_tile<AT>(sWi, TLDW, 16 * t, f2, bfrag_a(s_tp + TP_BIH, t) + s_gh[t][l]);
                const f32x4 pz_s = dense_tile<AT>(sWi, TLDW, AM + 16 * t, f2, bfrag_a(s_tp + TP_BIH, AT + t) + s_gh[AT + t][l]);
                const f32x4 gn_s = dense_tile<AT>(sWi, TLDW, 2 * AM + 16 * t, f2, bfrag_a(s_tp + TP_BIH, 2 * AT + t));
                hnew[t] = gru_gates(pr_s, pz_s, gn_s, s_gh[2 * AT + t][l], h[t]).h;
                continue;
            }
            f32x4 prr = pv(TP_BIH, IPLAN_AC_BIH, t) + pv(TP_BHH, IPLAN_AC_BHH, t);
            f32x4 pz = pv(TP_BIH, IPLAN_AC_BIH, AT + t) + pv(TP_BHH, IPLAN_AC_BHH, AT + t);
            f32x4 gn = pv(TP_BIH, IPLAN_AC_BIH, 2 * AT + t);
            f32x4 hn = pv(TP_BHH, IPLAN_AC_BHH, 2 * AT + t);
            if (STAGED2) {
                const float* sWi = s_tw + AM * TLDW;
                const float* sWh = s_tw + (4 * AM + 16) * TLDW;
                prr = dense_tile<AT>(sWh, TLDW, 16 * t, h, dense_tile<AT>(sWi, TLDW, 16 * t, f2, prr));
                pz = dense_tile<AT>(sWh, TLDW, AM + 16 * t, h, dense_tile<AT>(sWi, TLDW, AM + 16 * t, f2, pz));
                gn = dense_tile<AT>(sWi, TLDW, 2 * AM + 16 * t, f2, gn);
                hn = dense_tile<AT>(sWh, TLDW, 2 * AM + 16 * t, h, hn);
            } else {
            prr = dense_tile_ga<AT>(Wi, AM, 3 * AM, 16 * t, f2, prr);
            prr = dense_tile_ga<AT>(Wh, AM, 3 * AM, 16 * t, h, prr);
            pz = dense_tile_ga<AT>(Wi, AM, 3 * AM, AM + 16 * t, f2, pz);
            pz = dense_tile_ga<AT>(Wh, AM, 3 * AM, AM + 16 * t, h, pz);
            gn = dense_tile_ga<AT>(Wi, AM, 3 * AM, 2 * AM + 16 * t, f2, gn);
            hn = dense_tile_ga<AT>(Wh, AM, 3 * AM, 2 * AM + 16 * t, h, hn);
            }
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
    }   // (one-wave tail)
    float* hout = which ? a.h_critic_out : a.h_actor_out;
    if (hout) {
        float* orow = a.ho_s_row ? hout + (int64_t)net * a.ho_s_net + (int64_t)(valid ? r : 0) * a.ho_s_row
                                 : hout + ((int64_t)net * a.rows + (valid ? r : 0)) * AM;
        for (int t = 0; t < AT; ++t) vstore(orow, valid, AM, t, hnew[t]);
    }
    for (int t = 0; t < AT; ++t) { gmv[t] = pv(TP_LN3W, IPLAN_AC_LN3_W, t); btv[t] = pv(TP_LN3B, IPLAN_AC_LN3_B, t); }
    layer_norm_tiles_f<AT>(hnew, gmv, btv, &mu3, &rs3);
    if (sv) {
        for (int t = 0; t < AT; ++t) vstore(sv + 9 * AM, valid, AM, t, hnew[t]);      // f3
        if (valid && g == 0) {
            float* st = sv + 10 * AM;
            st[0] = mu_; st[1] = rstd_; st[2] = mu1; st[3] = rs1; st[4] = mu2; st[5] = rs2; st[6] = mu3; st[7] = rs3;
        }
    }
    // ---- head
    const int n_out = nw.n_out;
    const f32x4 lg = lds_tail
        ? dense_tile<AT>(s_tw + 4 * AM * TLDW, TLDW, 0, hnew, bfrag_a(s_tp + TP_HEADB, 0))
        : dense_tile_ga<AT>(P + nw.off[IPLAN_AC_HEAD_W], AM, n_out, 0, hnew, bfrag(P + nw.off[IPLAN_AC_HEAD_B], n_out, 0));
    const int64_t orow = (int64_t)net * a.rows + (valid ? r : 0);
    if (which == 1) {
        if (valid && g == 0 && a.values) a.values[orow] = lg[0];
        continue;
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
        if (valid && g == 0 && a.actions_out) {
            if (a.ao_s_row) a.actions_out[(int64_t)net * a.ao_s_net + (int64_t)r * a.ao_s_row] = (int64_t)action;
            else a.actions_out[orow] = (int64_t)action;
        }
        if (valid && a.onehot_out)
            for (int q = 0; q < 4; ++q)
                if (4 * g + q < n_out) a.onehot_out[(int64_t)net * a.oh_s_net + (int64_t)r * a.oh_s_row + 4 * g + q] = (4 * g + q == action) ? 1.0f : 0.0f;
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
  }   // row tiles
    if (clk) a.phase_clocks[3] = IPLAN_CLOCK();
    if (fclk) fclk[5] = IPLAN_CLOCK();
}

}  // namespace iplan
