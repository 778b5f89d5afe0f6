// fc1 of the recurrent actor / critic for the PPO epochs, on the bf16 matrix cores (include/iplan_hip.h, "fc1 of the PPO
// epochs"; learners/ippo_learner.py:190-221 evaluates the same stored rows ppo_epoch times, only the weights move).
//
// The fp32 forms (actor_critic.hip, actor_critic_bwd.hip) contract the F = 2485 features on v_mfma_f32_16x16x4_f32 -- exact,
// but that instruction runs at the fp32 vector rate (64 FLOP/clk/SIMD): 0.49 ms of matrix-pipe time per pass over the
// 114 750 x 2 x 64 x 2485 products of a config-3 epoch, and they reach about a third of it.  Here every operand is split
// into three bf16 pieces in registers (x = p0 + p1 + p2 exactly, wave_tile.h) and the six largest piece products are
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (1024 FLOP/clk/SIMD): 6/16 of the matrix time for a result that agrees
// with the fp32 contraction to fp32 round-off.  What makes that rate reachable:
//   * xhat is gathered and normalised ONCE per train() into fragment-major fp32 arrays (a wave's operand of one MFMA step
//     is one contiguous 2 KiB block) -- the fp32 kernels re-gather it from the episode-buffer fields in every pass;
//   * the actor and the critic of an agent see the same rows: one pass over xhat feeds both (M = 128 outputs);
//   * the operand every wave of a workgroup shares (forward: the weight pieces; gradient: the dz1 pieces) goes through
//     LDS, double-buffered, one barrier per K step of 32.
#include <cstdlib>

#include "api_util.h"
#include "ac_kmap.h"

namespace iplan {

constexpr int SM = IPLAN_AC_HIDDEN;            // 64
constexpr int S_OT = 2 * SM / 16;              // 8 output tiles: actor 0..3 | critic 4..7
constexpr int S_FRAG = S_OT * 3 * 64;          // bf16x8 entries of the LDS-shared operand per K step (24 KiB)

__device__ __forceinline__ int split_kt(const IplanAcFeatures& ft) {
    int t = 0;
    for (int s = 0; s < 3; ++s) t += (ft.N * ft.w[s] + 15) / 16;
    return t + (ft.n_actions + ft.n_id + 15) / 16;
}

#ifndef AC_SPLIT_ABL
#define AC_SPLIT_ABL 0                    // timing ablations of the forward kernel (results WRONG): 1 no piece split, 2 no xhat loads
#endif                                    // in the loop, 3 no weight staging / barrier in the loop, 4 three products instead of six
// The six piece products of one K = 32 step, smallest first, for NA x NB accumulators: every product is issued for all
// accumulators before the next one (independent MFMAs back to back -- a chain of six dependent ones per accumulator leaves
// the matrix pipe waiting for its own result).
template <int NA, int NB>
__device__ __forceinline__ void mma6_rr(const bf16x8 (&a)[NA][3], const Bf3 (&b)[NB], f32x4* (&acc)[NA][NB]) {
#if AC_SPLIT_ABL == 4
#define IPLAN_MMA6_PRODUCTS X(1, p0) X(0, p1) X(0, p0)
#else
#define IPLAN_MMA6_PRODUCTS X(2, p0) X(0, p2) X(1, p1) X(1, p0) X(0, p1) X(0, p0)
#endif
#define X(AP, BP)                                                                      \
    _Pragma("unroll") for (int i = 0; i < NA; ++i)                                     \
        _Pragma("unroll") for (int j = 0; j < NB; ++j) *acc[i][j] = mfma_bf16(a[i][AP], b[j].BP, *acc[i][j]);
    IPLAN_MMA6_PRODUCTS
#undef X
#undef IPLAN_MMA6_PRODUCTS
}

// ---- xhat fragments ------------------------------------------------------------------------------------------------------
// grid (ceil(rows / 32), n_agents), 128 threads: wave w gathers row tile 2 b + w exactly like the forward (ac_kmap.h), the
// 32 x 16 block of every k-tile is turned through LDS for the row-major-K copy.
__global__ __launch_bounds__(128) void ac_xhat_pack_kernel(IplanAcXhatArgs a) {
    __shared__ float s_x[32][17];
    const IplanAcFeatures& ft = a.feat;
    const int net = (int)blockIdx.y, rb = (int)blockIdx.x, RB = (int)gridDim.x;
    const int l = lane_id(), w = uniform_i(wave_id()), n = l & 15, g = l >> 4;
    const KMap km = make_kmap(ft);
    const int KT = km.kt0[4], KS = (KT + 1) / 2;
    const int tile = 2 * rb + w;
    const int r = tile * 16 + n;
    const bool vld = r < a.rows;
    const int64_t pr = vld ? (int64_t)(r / ft.T) * ft.T_phys + (r % ft.T) : 0;
    const float* src[3];
    for (int s = 0; s < 3; ++s) src[s] = ft.w[s] > 0 ? ft.src[s] + (int64_t)net * ft.s_net[s] + pr * ft.s_row[s] : nullptr;
    int last = -1;
    if (vld && ft.n_actions > 0) {
        if (ft.last_action) last = ft.last_action[(int64_t)net * ft.la_s_net + pr * ft.la_s_row];
        else if (ft.last_action64) last = (int)ft.last_action64[(int64_t)net * ft.la64_s_net + pr * ft.la64_s_row];
    }
    float mu = 0.f, rstd = 0.f;
    if (vld) {
        const float* st = a.ln_stats + (int64_t)net * a.ln_stats_s_net + pr * 2;
        mu = st[0]; rstd = st[1];
    }
    float* __restrict__ xf = a.xf + (((int64_t)net * 2 * RB + tile) * KS) * 512 + l * 8;
    float* __restrict__ xb = a.xb + (((int64_t)net * RB + rb) * KT) * 512;
    const int t = (int)threadIdx.x, bf = t & 15, bg = (t >> 4) & 3, bh = t >> 6;
    for (int T = 0; T < 2 * KS; ++T) {
        f32x4 xh = splat4(0.f);
        if (T < KT) {
            const KTile kt = ktile(km, T);
            const f32x4 x = kfeat(km, kt, src, vld, last, net);
            for (int q = 0; q < 4; ++q) xh[q] = (vld && q < kt.nv) ? (x[q] - mu) * rstd : 0.f;
        }
        *reinterpret_cast<f32x4*>(xf + (int64_t)(T >> 1) * 512 + (T & 1) * 4) = xh;
        if (T < KT) {
            for (int q = 0; q < 4; ++q) s_x[16 * w + n][4 * g + q] = xh[q];
            __syncthreads();
            f32x4 o;
            for (int j = 0; j < 4; ++j) o[j] = s_x[8 * bg + 4 * bh + j][bf];
            *reinterpret_cast<f32x4*>(xb + (int64_t)T * 512 + (16 * bg + bf) * 8 + 4 * bh) = o;
            __syncthreads();
        }
    }
}

// ---- weight pieces: (W o gamma) in A-fragment order, three bf16 pieces ------------------------------------------------
// block (ks, net, which), 256 threads = 4 output tiles x 64 lanes
__device__ __forceinline__ void ac_fc1_wsplit_block(const IplanAcFc1SplitArgs& a, int ks, int net, int which) {
    const IplanAcNet& nw = which ? a.critic : a.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    const KMap km = make_kmap(a.feat);
    const int F = km.NW + km.n_actions + km.n_id, KT = km.kt0[4], KS = (KT + 1) / 2;
    const int t = (int)threadIdx.x, ot = t >> 6, lane = t & 63, m = lane & 15, g = lane >> 4;
    const float* __restrict__ Wrow = P + nw.off[IPLAN_AC_FC1_W] + (int64_t)(16 * ot + m) * F;
    const float* __restrict__ gam = P + nw.off[IPLAN_AC_FN_W];
    f32x4 v[2];
    for (int hf = 0; hf < 2; ++hf) {
        v[hf] = splat4(0.f);
        const int T = 2 * ks + hf;
        if (T < KT) {
            const KTile kt = ktile_at(km, T, 4 * g);
            for (int q = 0; q < 4; ++q)
                if (q < kt.nv) v[hf][q] = Wrow[kt.c[q]] * gam[kt.c[q]];
        }
    }
    const Bf3 s = split_bf3(v[0], v[1]);
    bf16x8* dst = reinterpret_cast<bf16x8*>(a.wsplit) + (((int64_t)net * KS + ks) * 2 + which) * (4 * 3 * 64) + ot * 192 + lane;
    dst[0] = s.p0;
    dst[64] = s.p1;
    dst[128] = s.p2;
}

// (W beta)[o]: block (net, which, 8-row group bz), 256 threads = 8 output rows x 32 threads striding the feature axis together
__device__ __forceinline__ void ac_fc1_wbeta_block(const IplanAcFc1SplitArgs& a, int net, int which, int bz, float (&s_p)[8][32]) {
    const IplanAcNet& nw = which ? a.critic : a.actor;
    const float* __restrict__ P = nw.params + (int64_t)net * nw.params_s_net;
    const IplanAcFeatures& ft = a.feat;
    const int F = ft.N * (ft.w[0] + ft.w[1] + ft.w[2]) + ft.n_actions + ft.n_id;
    const int part = (int)threadIdx.x & 31, ro = (int)threadIdx.x >> 5, o = bz * 8 + ro;
    float c = 0.f;
    for (int k = part; k < F; k += 32) c = fmaf(P[nw.off[IPLAN_AC_FC1_W] + (int64_t)o * F + k], P[nw.off[IPLAN_AC_FN_B] + k], c);
    s_p[ro][part] = c;
    __syncthreads();
    if (part == 0) {
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += s_p[ro][k];
        a.wbeta[((int64_t)which * a.n_agents + net) * SM + o] = s;
    }
}

// Both weight preparations of an epoch as ONE launch (they are independent, and each was a 20 - 25 us link of the chain between an
// epoch's optimiser step and its forward pass): blocks [0, KS n_agents 2) split the weights, the rest compute W beta.
__global__ __launch_bounds__(256) void ac_fc1_wprep_kernel(IplanAcFc1SplitArgs a, int KS) {
    __shared__ float s_p[8][32];
    const int b = (int)blockIdx.x, n_ws = KS * a.n_agents * 2;
    if (b < n_ws) {
        ac_fc1_wsplit_block(a, b % KS, (b / KS) % a.n_agents, b / (KS * a.n_agents));
    } else {
        const int j = b - n_ws;
        ac_fc1_wbeta_block(a, j % a.n_agents, (j / a.n_agents) & 1, j / (2 * a.n_agents), s_p);
    }
}

// ---- forward: z1[which][net][row][o] = sum_k (W o gamma)[o][k] xhat[row][k] + (W beta)[o] ---------------------------
// grid (ceil(row tiles / (SF_RT NW)), n_agents), NW waves: wave w owns SF_RT consecutive row tiles and all 8 output tiles
// (SF_RT x 32 accumulator registers: 128 at the adopted 8 waves x 4 tiles); the 24 KiB of weight pieces of a K step are staged
// once per workgroup.
#ifndef AC_SPLIT_RT
#define AC_SPLIT_RT 4
#endif
constexpr int SF_RT = AC_SPLIT_RT;                // row tiles (forward) / k-tiles (gradient) per wave
#ifndef AC_SPLIT_NW
#define AC_SPLIT_NW 8                     // waves per workgroup of the two contraction kernels
#endif
// RT = row tiles per wave: RT for the full buffer; 1 for small batches (a data-parallel rank's 2 880 rows: 4 row tiles per wave
// leave 30 workgroups on the chip, each one long K chain -- with one tile per wave there are four times as many)
template <int NW, int RT>
__global__ __launch_bounds__(64 * NW) void ac_fc1_split_fwd_kernel(IplanAcFc1SplitArgs a) {
    constexpr int NTH = 64 * NW, WST = S_FRAG / NTH;
    __shared__ __attribute__((aligned(16))) bf16x8 s_w[2][S_FRAG];
    // XCD-aware placement as in ac_fwd_kernel: an agent's workgroups (they share its 1.9 MB of weight pieces) are dealt to
    // one XCD's L2 instead of all eight
    int bx = (int)blockIdx.x, net = (int)blockIdx.y;
    {
        const int X = (int)gridDim.x, G = X * (int)gridDim.y;
        const int b = bx + X * net;
        const int k = b & 7, slot = b >> 3, q8 = G >> 3, r8 = G & 7;
        const int item = k * q8 + (k < r8 ? k : r8) + slot;
        bx = item % X;
        net = item / X;
    }
    const int l = lane_id(), w = uniform_i(wave_id()), n = l & 15, g = l >> 4;
    const int KT = split_kt(a.feat), KS = (KT + 1) / 2;
    const int tiles = (a.rows + 15) / 16, tiles_alloc = 2 * ((a.rows + 31) / 32);
    int tile[RT];
    const float* __restrict__ xp[RT];
    for (int t = 0; t < RT; ++t) {
        tile[t] = (bx * NW + w) * RT + t;
        xp[t] = a.xf + (((int64_t)net * tiles_alloc + imin(tile[t], tiles_alloc - 1)) * KS) * 512 + l * 8;
    }
    const bf16x8* __restrict__ wsrc = reinterpret_cast<const bf16x8*>(a.wsplit) + (int64_t)net * KS * S_FRAG;
    const int tx = (int)threadIdx.x;
    f32x4 acc[RT][S_OT];
    for (int t = 0; t < RT; ++t)
        for (int o = 0; o < S_OT; ++o) acc[t][o] = splat4(0.f);
    bf16x8 wst[WST];
    f32x4 xr[RT][2];
    // K-step range of this workgroup: all of it, or part blockIdx.z of a.kparts (small batches)
    const int kparts = a.kparts > 1 ? a.kparts : 1, kp = (int)blockIdx.z;
    const int ks_lo = (int)((int64_t)KS * kp / kparts), ks_hi = (int)((int64_t)KS * (kp + 1) / kparts);
    for (int i = 0; i < WST; ++i) wst[i] = wsrc[(int64_t)ks_lo * S_FRAG + tx + NTH * i];
    for (int t = 0; t < RT; ++t) {
        xr[t][0] = *reinterpret_cast<const f32x4*>(xp[t] + (int64_t)ks_lo * 512);
        xr[t][1] = *reinterpret_cast<const f32x4*>(xp[t] + (int64_t)ks_lo * 512 + 4);
    }
    for (int i = 0; i < WST; ++i) s_w[ks_lo & 1][tx + NTH * i] = wst[i];
    __syncthreads();
    for (int ks = ks_lo; ks < ks_hi; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < ks_hi;
        Bf3 xs[RT];
#if AC_SPLIT_ABL == 1
        for (int t = 0; t < RT; ++t) {
            xs[t].p0 = __builtin_bit_cast(bf16x8, xr[t][0]);
            xs[t].p1 = __builtin_bit_cast(bf16x8, xr[t][1]);
            xs[t].p2 = __builtin_bit_cast(bf16x8, xr[t][0] + xr[t][1]);
        }
#else
        for (int t = 0; t < RT; ++t) xs[t] = split_bf3(xr[t][0], xr[t][1]);
#endif
        if (more) {                                              // next step's operands: in flight under this step's MFMAs
#if AC_SPLIT_ABL != 3
            for (int i = 0; i < WST; ++i) wst[i] = wsrc[(int64_t)(ks + 1) * S_FRAG + tx + NTH * i];
#endif
#if AC_SPLIT_ABL != 2
            for (int t = 0; t < RT; ++t) {
                xr[t][0] = *reinterpret_cast<const f32x4*>(xp[t] + (int64_t)(ks + 1) * 512);
                xr[t][1] = *reinterpret_cast<const f32x4*>(xp[t] + (int64_t)(ks + 1) * 512 + 4);
            }
#endif
        }
#pragma unroll
        for (int o = 0; o < S_OT; o += 2) {
            const bf16x8* sw = &s_w[AC_SPLIT_ABL == 3 ? 0 : cur][o * 192 + l];
            const bf16x8 wp[2][3] = {{sw[0], sw[64], sw[128]}, {sw[192], sw[256], sw[320]}};
            f32x4* ap[2][RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) { ap[0][t] = &acc[t][o]; ap[1][t] = &acc[t][o + 1]; }
            mma6_rr<2, RT>(wp, xs, ap);
        }
#if AC_SPLIT_ABL != 3
        if (more)
            for (int i = 0; i < WST; ++i) s_w[cur ^ 1][tx + NTH * i] = wst[i];
        __syncthreads();
#endif
    }
    float* __restrict__ zout = a.z1 + (int64_t)kp * 2 * a.n_agents * a.rows * SM;          // part kp of the output
    for (int o = 0; o < S_OT; ++o) {
        const int which = o >> 2, oc = 16 * (o & 3) + 4 * g;
        f32x4 c = splat4(0.f);
        if (kp == 0) c = *reinterpret_cast<const f32x4*>(a.wbeta + ((int64_t)which * a.n_agents + net) * SM + oc);
        for (int t = 0; t < RT; ++t) {
            const int r = tile[t] * 16 + n;
            if (tile[t] < tiles && r < a.rows)
                *reinterpret_cast<f32x4*>(zout + (((int64_t)which * a.n_agents + net) * a.rows + r) * SM + oc) = acc[t][o] + c;
        }
    }
}

// ---- weight gradient: G[which][net][m][k] = sum_rows dz1[row][m] xhat[row][k] ----------------------------------------
// grid (ceil(KT / (SW_NF NW)), row chunks, n_agents), NW waves: wave w owns SW_NF consecutive k-tiles and all 8 output tiles of
// (actor | critic); per 32-row step the dz1 pieces (every wave loads, splits and stages 8 / NW output tiles) are shared through LDS.
// Partial tiles per row chunk in iplan_ac_bwd_fc1's g_part layout, summed in chunk order by iplan_ac_bwd_fc1_finalize.
constexpr int SW_NF = AC_SPLIT_RT;
template <int NW>
__global__ __launch_bounds__(64 * NW) void ac_fc1_split_wgrad_kernel(IplanAcBwdArgs a) {
    constexpr int MS = S_OT / NW;                                  // output tiles of dz1 a wave stages per step
    __shared__ __attribute__((aligned(16))) bf16x8 s_a[2][S_FRAG];
    const IplanAcFwdArgs& fa = a.fwd;
    const int net = (int)blockIdx.z, chunk = (int)blockIdx.y;
    const int l = lane_id(), w = uniform_i(wave_id()), i = l & 15, g = l >> 4;
    const int KT = split_kt(fa.feat), Kpad = KT * 16;
    const int RB = (fa.rows + 31) / 32;
    const int b_lo = chunk * (a.fc1_chunk_rows / 32), b_hi = imin(RB, b_lo + a.fc1_chunk_rows / 32);
    const int steps = imax(0, b_hi - b_lo);                       // (an empty chunk still writes its zero partial tiles)
    // staging role: output tiles MS w + e = (which, 16 columns) of dz1, this lane's 8 rows 8 g + j of the step
    const float* __restrict__ dz[MS];
    for (int e = 0; e < MS; ++e) {
        const int mt = MS * w + e;
        dz[e] = a.dsave + (((int64_t)(mt >> 2) * fa.n_agents + net) * fa.rows) * IPLAN_AC_DSAVE_FLOATS + 16 * (mt & 3) + i;
    }
    auto load_dz = [&](int b, f32x4 (&v)[MS][2]) {
#pragma unroll
        for (int e = 0; e < MS; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 32 * b + 8 * g + j;
                v[e][j >> 2][j & 3] = r < fa.rows ? dz[e][(int64_t)r * IPLAN_AC_DSAVE_FLOATS] : 0.f;
            }
    };
    auto stage_dz = [&](const f32x4 (&v)[MS][2], bf16x8* buf) {
#pragma unroll
        for (int e = 0; e < MS; ++e) {
            const Bf3 s = split_bf3(v[e][0], v[e][1]);
            bf16x8* d = buf + (MS * w + e) * 192 + l;
            d[0] = s.p0; d[64] = s.p1; d[128] = s.p2;
        }
    };
    int T[SW_NF];
    const float* __restrict__ xp[SW_NF];
    for (int u = 0; u < SW_NF; ++u) {
        T[u] = ((int)blockIdx.x * NW + w) * SW_NF + u;
        xp[u] = a.xb + (((int64_t)net * RB + imin(b_lo, RB - 1)) * KT + imin(T[u], KT - 1)) * 512 + l * 8;
    }
    f32x4 acc[S_OT][SW_NF];
    for (int m = 0; m < S_OT; ++m)
        for (int u = 0; u < SW_NF; ++u) acc[m][u] = splat4(0.f);
    f32x4 dr[MS][2], xr[SW_NF][2];
    if (steps > 0) {
        load_dz(b_lo, dr);
        for (int u = 0; u < SW_NF; ++u) {
            xr[u][0] = *reinterpret_cast<const f32x4*>(xp[u]);
            xr[u][1] = *reinterpret_cast<const f32x4*>(xp[u] + 4);
        }
        stage_dz(dr, &s_a[0][0]);
    }
    __syncthreads();
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
        const bool more = st + 1 < steps;
        Bf3 xs[SW_NF];
        for (int u = 0; u < SW_NF; ++u) xs[u] = split_bf3(xr[u][0], xr[u][1]);
        if (more) {
            load_dz(b_lo + st + 1, dr);
            for (int u = 0; u < SW_NF; ++u) {
                xr[u][0] = *reinterpret_cast<const f32x4*>(xp[u] + (int64_t)(st + 1) * KT * 512);
                xr[u][1] = *reinterpret_cast<const f32x4*>(xp[u] + (int64_t)(st + 1) * KT * 512 + 4);
            }
        }
#pragma unroll
        for (int m = 0; m < S_OT; m += 2) {
            const bf16x8* sa = &s_a[cur][m * 192 + l];
            const bf16x8 dp[2][3] = {{sa[0], sa[64], sa[128]}, {sa[192], sa[256], sa[320]}};
            f32x4* ap[2][SW_NF];
#pragma unroll
            for (int u = 0; u < SW_NF; ++u) { ap[0][u] = &acc[m][u]; ap[1][u] = &acc[m + 1][u]; }
            mma6_rr<2, SW_NF>(dp, xs, ap);
        }
        if (more) stage_dz(dr, &s_a[cur ^ 1][0]);
        __syncthreads();
    }
    for (int m = 0; m < S_OT; ++m) {
        float* __restrict__ part = a.g_part + ((((int64_t)(m >> 2) * fa.n_agents + net) * a.fc1_chunks + chunk) * SM + 16 * (m & 3) + 4 * g) * (int64_t)Kpad;
        for (int u = 0; u < SW_NF; ++u)
            if (T[u] < KT)
                for (int q = 0; q < 4; ++q) part[(int64_t)q * Kpad + T[u] * 16 + i] = acc[m][u][q];
    }
}

}  // namespace iplan

extern "C" int64_t iplan_ac_xhat_floats(const IplanAcFeatures* ft, int32_t rows, int32_t which) {
    if (!ft || rows < 1) return 0;
    int kt = 0;
    for (int s = 0; s < 3; ++s) kt += (ft->N * ft->w[s] + 15) / 16;
    kt += (ft->n_actions + ft->n_id + 15) / 16;
    const int64_t rb = (rows + 31) / 32;
    return which == 0 ? 2 * rb * ((kt + 1) / 2) * 512 : rb * kt * 512;
}

extern "C" int iplan_ac_xhat_pack(const IplanAcXhatArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_agents < 1 || a->rows < 1 || !a->ln_stats || !a->xf || !a->xb || !aligned16(a->xf) || !aligned16(a->xb))
        return fail(IPLAN_EINVAL, "iplan_ac_xhat_pack: bad arguments");
    if (a->feat.T < 1 || a->feat.T_phys < a->feat.T) return fail(IPLAN_EINVAL, "iplan_ac_xhat_pack: bad T/T_phys");
    hipLaunchKernelGGL(ac_xhat_pack_kernel, dim3((unsigned)((a->rows + 31) / 32), (unsigned)a->n_agents), dim3(128), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_xhat_pack");
}

// small batches: one row tile per wave (IPLAN_AC_SPLIT_SHAPE=full / small forces a shape: tests run both at one size)
static bool split_small_shape(int n_agents, int rows) {
    const char* shape = getenv("IPLAN_AC_SPLIT_SHAPE");
    const int tiles = (rows + 15) / 16;
    return shape ? shape[0] == 's' : tiles * n_agents <= 2 * 256 * AC_SPLIT_NW;
}

// K parts of the small-batch forward: as many as bring its workgroup count to about two per CU, at most 4 (the consumer re-reads
// every part; a part may come out empty when there are fewer K steps than parts -- it then holds zeros / W beta alone)
extern "C" int iplan_ac_fc1_split_parts(int32_t n_agents, int32_t rows) {
    if (n_agents < 1 || rows < 1 || !split_small_shape(n_agents, rows)) return 1;
    const char* e = getenv("IPLAN_AC_SPLIT_KPARTS");
    if (e && e[0]) return iplan::imax(1, iplan::imin(8, atoi(e)));
    const int wgs = ((rows + 15) / 16 + AC_SPLIT_NW - 1) / AC_SPLIT_NW * n_agents;
    return iplan::imax(1, iplan::imin(4, 512 / iplan::imax(wgs, 1)));
}

extern "C" int iplan_ac_fc1_split_fwd(const IplanAcFc1SplitArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_agents < 1 || a->rows < 1 || !a->xf || !a->wsplit || !a->wbeta || !a->z1 || !a->actor.params || !a->critic.params ||
        !aligned16(a->wsplit) || !aligned16(a->wbeta) || !aligned16(a->z1))
        return fail(IPLAN_EINVAL, "iplan_ac_fc1_split_fwd: bad arguments");
    int kt = 0;
    for (int s = 0; s < 3; ++s) kt += (a->feat.N * a->feat.w[s] + 15) / 16;
    kt += (a->feat.n_actions + a->feat.n_id + 15) / 16;
    const int KS = (kt + 1) / 2, tiles = (a->rows + 15) / 16;
    hipLaunchKernelGGL(ac_fc1_wprep_kernel, dim3((unsigned)(KS * a->n_agents * 2 + a->n_agents * 2 * (SM / 8))), dim3(256), 0, (hipStream_t)stream, *a, KS);
    constexpr int TPW = AC_SPLIT_NW * SF_RT;                       // row tiles per workgroup
    const bool small = split_small_shape(a->n_agents, a->rows);
    const int kparts = a->kparts > 1 ? a->kparts : 1;
    if (kparts > 1 && (!small || kparts > iplan_ac_fc1_split_parts(a->n_agents, a->rows)))
        return fail(IPLAN_EINVAL, "iplan_ac_fc1_split_fwd: kparts = %d, but iplan_ac_fc1_split_parts() allows %d at this size", kparts,
                    iplan_ac_fc1_split_parts(a->n_agents, a->rows));
    if (small) {
        hipLaunchKernelGGL((ac_fc1_split_fwd_kernel<AC_SPLIT_NW, 1>), dim3((unsigned)((tiles + AC_SPLIT_NW - 1) / AC_SPLIT_NW), (unsigned)a->n_agents, (unsigned)kparts),
                           dim3(64 * AC_SPLIT_NW), 0, (hipStream_t)stream, *a);
        return check_launch("iplan_ac_fc1_split_fwd");
    }
    hipLaunchKernelGGL((ac_fc1_split_fwd_kernel<AC_SPLIT_NW, SF_RT>), dim3((unsigned)((tiles + TPW - 1) / TPW), (unsigned)a->n_agents), dim3(64 * AC_SPLIT_NW), 0,
                       (hipStream_t)stream, *a);
    return check_launch("iplan_ac_fc1_split_fwd");
}

// Row chunking of iplan_ac_bwd_fc1_split: as many chunks as give every CU its resident workgroups once (one 8-wave or
// three 4-wave workgroups of ~150 registers per lane), chunk_rows a multiple of the 32-row step.
extern "C" int iplan_ac_fc1_split_chunks(const IplanAcFeatures* ft, int32_t n_agents, int32_t rows, int32_t* chunk_rows) {
    using namespace iplan;
    if (!ft || !chunk_rows || n_agents < 1 || rows < 1) return 0;
    int kt = 0;
    for (int s = 0; s < 3; ++s) kt += (ft->N * ft->w[s] + 15) / 16;
    kt += (ft->n_actions + ft->n_id + 15) / 16;
    constexpr int KPW = AC_SPLIT_NW * SW_NF;
    const int kblocks = (kt + KPW - 1) / KPW;
    const int slots = 256 * (AC_SPLIT_NW == 8 ? 1 : (AC_SPLIT_RT == 2 ? 3 : 2));
    const int want = imax(1, slots / (kblocks * n_agents));
    const int cr = imax(32, ((rows + want - 1) / want + 31) / 32 * 32);
    *chunk_rows = cr;
    return (rows + cr - 1) / cr;
}

extern "C" int iplan_ac_bwd_fc1_split(const IplanAcBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || !a->xb || !a->dsave || !a->g_part || a->fwd.which != 2 || a->fwd.n_agents < 1 || a->fwd.rows < 1)
        return fail(IPLAN_EINVAL, "iplan_ac_bwd_fc1_split: needs xb, dsave, g_part and a which = 2 forward");
    if (a->fc1_chunk_rows < 32 || (a->fc1_chunk_rows & 31) || (int64_t)a->fc1_chunks * a->fc1_chunk_rows < a->fwd.rows)
        return fail(IPLAN_EINVAL, "iplan_ac_bwd_fc1_split: bad chunking (%d chunks x %d rows for %d rows)", a->fc1_chunks,
                    a->fc1_chunk_rows, a->fwd.rows);
    int kt = 0;
    for (int s = 0; s < 3; ++s) kt += (a->fwd.feat.N * a->fwd.feat.w[s] + 15) / 16;
    kt += (a->fwd.feat.n_actions + a->fwd.feat.n_id + 15) / 16;
    constexpr int KPW = AC_SPLIT_NW * SW_NF;                       // k-tiles per workgroup
    hipLaunchKernelGGL(ac_fc1_split_wgrad_kernel<AC_SPLIT_NW>, dim3((unsigned)((kt + KPW - 1) / KPW), (unsigned)a->fc1_chunks, (unsigned)a->fwd.n_agents),
                       dim3(64 * AC_SPLIT_NW), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_bwd_fc1_split");
}
