// GAT-RNN instant-incentive encoder: forward pass of GAT_Net (nova/GAT_Net.py:41-142) for every
// (net, env) scene in one launch.
//
// One 512-thread workgroup (8 waves) owns one scene = N <= 64 entities of one (agent-net, env):
//   phase 1  node encode + per-node projections (MFMA): h = ReLU(W_enc x + b); the bi-GRU input
//            projection is separable, W_ih [h_i ; h_j] = W_a h_i + W_b h_j, so the
//            [N-1, B*N, 2H] pair tensor the reference materialises (GAT_Net.py:57-75, 55 % of its
//            CPU time) never exists: W_a h_i stays in registers of the wave that owns chain i,
//            W_b h_j goes to LDS.  q, k, v also go to LDS.
//   phase 2  hard-attention bi-GRU: wave (dir, tile) runs 16 ego chains for N-1 steps; W_hh lives in
//            registers as MFMA A fragments, the hidden state never leaves the D layout, gates are
//            lane-local.  At step s every chain needs W_b h_j for j = s + [s >= i], i.e. one of two
//            LDS rows -> near-broadcast reads.  Each step's contribution to the 2 hard logits is
//            reduced over the 4 lane groups and parked in LDS.
//   phase 3  per 16-ego tile (one wave each): scaled dot-product scores and the gated aggregation of v as two
//            small MFMA products with the ego as the column of the D layout (the softmax over the N-1
//            neighbours and the gumbel-softmax gate, tau = 0.01, are then lane-local).
//   phase 4  output GRUCell (MFMA) -> new attention latent.
// HBM traffic per scene is the compulsory obs + h_prev + noise + out (~1.4 MB per net at cfg3);
// everything else lives in the 150 KB of LDS / registers.  Training launches additionally stream
// the activations the backward pass needs.
#include "api_util.h"
#include "wave_tile.h"
#include "enc_body.h"
#include "ac_fwd_body.h"

namespace iplan {

constexpr int GH = IPLAN_GAT_HIDDEN;   // H == A == 32
constexpr int NP = IPLAN_MAX_ENTITIES; // 64
constexpr int BST = 100;               // padded row stride (floats) of the W_b h_j table
constexpr int QST = 33;                // padded row stride of q/k/v/x tables
#ifndef IPLAN_GAT_BF3
#define IPLAN_GAT_BF3 1                // recurrence on the bf16 matrix cores (fp32-exact split); 0 = fp32 MFMA (A/B builds)
#endif

struct GatShared {                      // LDS of one scene (157 KB)
    float B[2][NP][BST];
    float q[NP][QST], k[NP][QST], v[NP][QST], x[NP][QST], x1[NP][QST];      // x / x1: the two directions' partial aggregates (phase 3)
    float pl[2][NP][NP][2];
    float sm[2][4][16][2];              // phase 3: (running maximum, partial denominator) of an ego's softmax, per direction wave
};

// one scene = workgroup `block` of the launch (512 threads); COH: the new latent is stored device-coherently (read by other
// workgroups of the same launch: gat_enc_ac_fwd_kernel)
// FOLD: the gates' exp2 constants folded into the recurrence's operands (below; launches that do not store the gate record)
template <bool COH = false, bool FOLD = false>
__device__ __forceinline__ void gat_fwd_block(const IplanGatFwdArgs& a, int block, GatShared& sh) {
    auto& s_B = sh.B;
    auto& s_q = sh.q;
    auto& s_k = sh.k;
    auto& s_v = sh.v;
    auto& s_x = sh.x;
    auto& s_x1 = sh.x1;
    auto& s_sm = sh.sm;
    auto& s_pl = sh.pl;

    const int net = block / a.B;
    const int b = block % a.B;
    const int N = a.N;
    const int D = a.d0 + a.d1;
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    const int l = lane_id(), w = wave_id();
    const int n = l & 15, g = l >> 4;
    const int tile = w & 3, dir = w >> 2;
    const int node = 16 * tile + n;
    const bool tile_live = 16 * tile < N;
    const bool valid = node < N;
    const int64_t sb = (int64_t)net * a.B + b;
    const IplanGatSaved& sv = a.saved;
#ifdef GAT_P3_CLOCKS                   // profiling build (scripts/build_variants.sh): 6 more clocks inside phase 3
    constexpr int CLK_STRIDE = 12;
#define GAT_SUBCLK(i) do { if (clk && threadIdx.x == 0) clk[i] = IPLAN_CLOCK(); } while (0)
#else
    constexpr int CLK_STRIDE = 5;
#define GAT_SUBCLK(i) do {} while (0)
#endif
    int64_t* clk = a.phase_clocks ? a.phase_clocks + (int64_t)block * CLK_STRIDE : nullptr;
    if (clk && threadIdx.x == 0) clk[0] = IPLAN_CLOCK();

    // FOLD (the launcher picks it when the gate record is not stored): the gates' exp2 constants -- sigmoid(x) = rcp(1 + exp2(-log2e x)), tanh(x) = 1 - 2 rcp(exp2(2 log2e x) + 1)
    // -- are folded into the recurrence's OPERANDS once per launch (x-projection, W_b h_j rows, b_hn, and W_hh's rows before their bf16
    // split), so the pre-activations arrive scaled and the 48 multiplies per wave-step in front of v_exp are gone (gru_gates_folded).  The
    // training form stores hn for the backward pass and keeps the plain operands.
    constexpr float GATE_RZ = -1.4426950408889634f, GATE_N = 2.8853900817779268f;
    constexpr bool fold = IPLAN_GAT_BF3 && FOLD;
    const float* bih = P + a.off[dir ? IPLAN_GAT_R_BIH : IPLAN_GAT_F_BIH];
    const float* bhh = P + a.off[dir ? IPLAN_GAT_R_BHH : IPLAN_GAT_F_BHH];

    // ---------------------------------------------------------------- phase 1: node projections
    f32x4 areg[6];
    for (int t = 0; t < 6; ++t) areg[t] = splat4(0.f);
    if (tile_live) {
        f32x4 h[2];
        {
            f32x4 acc0 = bfrag(P + a.off[IPLAN_GAT_ENC_B], GH, 0);
            f32x4 acc1 = bfrag(P + a.off[IPLAN_GAT_ENC_B], GH, 1);
            const float* r0 = a.src0 + (int64_t)net * a.src0_s_net + (int64_t)b * a.src0_s_b + (int64_t)node * a.d0;
            const float* r1 = a.d1 > 0 ? a.src1 + (int64_t)net * a.src1_s_net + (int64_t)b * a.src1_s_b + (int64_t)node * a.d1 : nullptr;
            const float* Wenc = P + a.off[IPLAN_GAT_ENC_W];
            const int KT = (D + 15) / 16;
            for (int T = 0; T < KT; ++T) {
                f32x4 x = splat4(0.f);
                if (valid) {
                    for (int q = 0; q < 4; ++q) {
                        const int c = 16 * T + 4 * g + q;
                        if (c < a.d0) x[q] = r0[c];
                        else if (c < D) x[q] = r1[c - a.d0];
                    }
                }
                acc0 = mma_block(wfrag(Wenc, D, GH, D, 0, 16 * T), x, acc0);
                acc1 = mma_block(wfrag(Wenc, D, GH, D, 16, 16 * T), x, acc1);
            }
            h[0] = relu4(acc0);
            h[1] = relu4(acc1);
        }
        if (sv.h_enc && dir == 0) {
            float* row = sv.h_enc + (sb * N + node) * GH;
            vstore(row, valid, GH, 0, h[0]);
            vstore(row, valid, GH, 1, h[1]);
        }
        const float* Wih = P + a.off[dir ? IPLAN_GAT_R_WIH : IPLAN_GAT_F_WIH];   // [3H][2H]
        for (int t = 0; t < 6; ++t) {
            f32x4 ac = bfrag(bih, 3 * GH, t);
            if (t < 4) ac += bfrag(bhh, 3 * GH, t);          // r,z gates: both biases sit outside r*(.)
            f32x4 bc = splat4(0.f);
            for (int T = 0; T < 2; ++T) {
                ac = mma_block(wfrag_a(Wih, 2 * GH, 3 * GH, 16 * t, 16 * T), h[T], ac);
                bc = mma_block(wfrag_a(Wih, 2 * GH, 3 * GH, 16 * t, GH + 16 * T), h[T], bc);
            }
            if (fold) { const float cs = t < 4 ? GATE_RZ : GATE_N; ac *= cs; bc *= cs; }
            areg[t] = ac;
            if (valid) *reinterpret_cast<f32x4*>(&s_B[dir][node][16 * t + 4 * g]) = bc;
        }
        if (dir == 0) {
            for (int which = 0; which < 2; ++which) {
                const float* Wm = P + a.off[which ? IPLAN_GAT_K_W : IPLAN_GAT_Q_W];
                float (*dst)[QST] = which ? s_k : s_q;
                for (int t = 0; t < 2; ++t) {
                    f32x4 ac = splat4(0.f);
                    for (int T = 0; T < 2; ++T) ac = mma_block(wfrag_a(Wm, GH, GH, 16 * t, 16 * T), h[T], ac);
                    if (valid)
                        for (int q = 0; q < 4; ++q) dst[node][16 * t + 4 * g + q] = ac[q];
                    if (sv.qkv) vstore(sv.qkv + ((sb * N + node) * 3 + which) * GH, valid, GH, t, ac);
                }
            }
        } else {
            const float* Wm = P + a.off[IPLAN_GAT_V_W];
            for (int t = 0; t < 2; ++t) {
                f32x4 ac = bfrag(P + a.off[IPLAN_GAT_V_B], GH, t);
                for (int T = 0; T < 2; ++T) ac = mma_block(wfrag_a(Wm, GH, GH, 16 * t, 16 * T), h[T], ac);
                ac = relu4(ac);
                if (valid)
                    for (int q = 0; q < 4; ++q) s_v[node][16 * t + 4 * g + q] = ac[q];
                if (sv.qkv) vstore(sv.qkv + ((sb * N + node) * 3 + 2) * GH, valid, GH, t, ac);
            }
        }
    }
    __syncthreads();
    if (clk && threadIdx.x == 0) clk[1] = IPLAN_CLOCK();

    // ---------------------------------------------------------------- phase 2: hard-attention bi-GRU
    if (tile_live) {
        const float* Whh = P + a.off[dir ? IPLAN_GAT_R_WHH : IPLAN_GAT_F_WHH];   // [3H][H]
        const float* Wh = P + a.off[IPLAN_GAT_HARD_W];                           // [2][2H]
        const f32x4 bhn0 = bfrag(bhh, 3 * GH, 4) * (fold ? GATE_N : 1.0f), bhn1 = bfrag(bhh, 3 * GH, 5) * (fold ? GATE_N : 1.0f);
        // hard-attention logits as a 7th MFMA chain: A = hard_encoding.weight[:, dir*H:(dir+1)*H] (2 real rows),
        // B = the hidden state -> class c of chain n lands in lane (n, g = 0), register c.  The chain runs one
        // step behind (it contracts the SAME h operand the recurrent chains use), so it rides along in the
        // round-robin issue order instead of costing cross-lane reductions on the critical path.
#if IPLAN_GAT_BF3
        // W_hh h on the bf16 matrix cores, fp32-exact (wave_tile.h, split-bf16): the three pieces of every weight
        // fragment are loop invariants in registers, the hidden state is split once per step.  42 K=32 MFMAs per step
        // (7 chains x 6 piece products) that run BESIDE the gate arithmetic of the SIMD's other wave, instead of 56
        // fp32 MFMAs that take the VALU's issue time (1 792 of a step's 2 700 cycles).  (Fetching the pieces in front of the
        // barrier that ends phase 1 was measured: phases 1 / 2 +1.4 / +1.8 us, 247 registers -- not kept.)
        Bf3 whh[6], wl;
        for (int t = 0; t < 6; ++t) whh[t] = wfrag_bf3_scaled(Whh, GH, 3 * GH, 16 * t, 0, fold ? (t < 4 ? GATE_RZ : GATE_N) : 1.0f);
        wl = wfrag_bf3(Wh + dir * GH, 2 * GH, 2, 0, 0);
#else
        f32x4 whh[6][2];
        for (int t = 0; t < 6; ++t)
            for (int T = 0; T < 2; ++T) whh[t][T] = wfrag_a(Whh, GH, 3 * GH, 16 * t, 16 * T);
        f32x4 wl[2];
        for (int T = 0; T < 2; ++T) wl[T] = wfrag(Wh, 2 * GH, 2, 2 * GH, 0, dir * GH + 16 * T);
#endif
        f32x4 h0 = splat4(0.f), h1 = splat4(0.f);
        auto brow = [&](int it) -> const float* {
            const int s = dir ? (N - 2 - it) : it;
            int j = s + (s >= node ? 1 : 0);
            if (j > N - 1) j = N - 1;
            return &s_B[dir][j][4 * g];
        };
        f32x4 bc[6], bn[6];
        {
            const float* Bj = brow(0);
            for (int t = 0; t < 6; ++t) bc[t] = *reinterpret_cast<const f32x4*>(Bj + 16 * t);
        }
        int s_prev = 0;
        for (int it = 0; it < N - 1; ++it) {
            const int s = dir ? (N - 2 - it) : it;
            if (it + 1 < N - 1) {                                   // next step's W_b h_j rows: issued now, consumed next iteration
                const float* Bj = brow(it + 1);
                for (int t = 0; t < 6; ++t) bn[t] = *reinterpret_cast<const f32x4*>(Bj + 16 * t);
            }
            f32x4 acc[7];
            acc[0] = areg[0]; acc[1] = areg[1]; acc[2] = areg[2]; acc[3] = areg[3];
            acc[4] = bhn0; acc[5] = bhn1; acc[6] = splat4(0.f);
            // 7 independent accumulator chains issued round-robin: consecutive MFMAs never depend on each other
#if IPLAN_GAT_BF3
            {
                // smallest piece products first (the accumulator is fp32).  MFMA time and VALU / transcendental time ADD on a
                // SIMD -- across its two waves as well as inside one (ablations and a software-pipelined variant that issued
                // tile 0's gates between tile 1's MFMAs: profiles/r02g_notes.md) -- so the plain order is kept.
                const Bf3 hs = split_bf3(h0, h1);
#define GAT_BF3_ROUND(WP, HP)                                                  \
    for (int c = 0; c < 6; ++c) acc[c] = mfma_bf16(whh[c].WP, hs.HP, acc[c]);  \
    acc[6] = mfma_bf16(wl.WP, hs.HP, acc[6]);
                GAT_BF3_ROUND(p2, p0) GAT_BF3_ROUND(p0, p2) GAT_BF3_ROUND(p1, p1)
                GAT_BF3_ROUND(p1, p0) GAT_BF3_ROUND(p0, p1) GAT_BF3_ROUND(p0, p0)
#undef GAT_BF3_ROUND
            }
#else
            for (int T = 0; T < 2; ++T) {
                const f32x4 hb = T ? h1 : h0;
                for (int q = 0; q < 4; ++q) {
                    for (int c = 0; c < 6; ++c) acc[c] = mfma4(whh[c][T][q], hb[q], acc[c]);
                    acc[6] = mfma4(wl[T][q], hb[q], acc[6]);
                }
            }
#endif
            if (it > 0 && g == 0 && valid) {                        // logits of the previous step
                s_pl[dir][node][s_prev][0] = acc[6][0];
                s_pl[dir][node][s_prev][1] = acc[6][1];
            }
            const f32x4 pr0 = acc[0] + bc[0], pr1 = acc[1] + bc[1], pz0 = acc[2] + bc[2], pz1 = acc[3] + bc[3];
            const f32x4 gn0 = areg[4] + bc[4], gn1 = areg[5] + bc[5];
            GruGates o0, o1;
            if constexpr (fold) {
                o0 = gru_gates_folded(pr0, pz0, gn0, acc[4], h0);
                o1 = gru_gates_folded(pr1, pz1, gn1, acc[5], h1);
            } else {
                o0 = gru_gates(pr0, pz0, gn0, acc[4], h0);
                o1 = gru_gates(pr1, pz1, gn1, acc[5], h1);
            }
            h0 = o0.h;
            h1 = o1.h;
            if (sv.gru) {
                // [net][dir][scene][ego tile][step][group: h h r r z z n n][16 chains][16 columns]: one 1 KiB block per store
                // instruction (gat_bwd.hip reads it back the same way; it recomputes hn = W_hn h_prev + b_hn -- a fifth of the record)
                float* blk = sv.gru + ((((((int64_t)net * 2 + dir) * a.B + b) * ((N + 15) / 16) + tile) * (N - 1) + s) * 8) * 256 + n * 16 + 4 * g;
                if (valid) {
                    *reinterpret_cast<f32x4*>(blk) = o0.h;             *reinterpret_cast<f32x4*>(blk + 256) = o1.h;
                    *reinterpret_cast<f32x4*>(blk + 2 * 256) = o0.r;   *reinterpret_cast<f32x4*>(blk + 3 * 256) = o1.r;
                    *reinterpret_cast<f32x4*>(blk + 4 * 256) = o0.z;   *reinterpret_cast<f32x4*>(blk + 5 * 256) = o1.z;
                    *reinterpret_cast<f32x4*>(blk + 6 * 256) = o0.n;   *reinterpret_cast<f32x4*>(blk + 7 * 256) = o1.n;
                }
            }
            s_prev = s;
            for (int t = 0; t < 6; ++t) bc[t] = bn[t];
        }
        {   // logits of the last step
            f32x4 la = splat4(0.f);
#if IPLAN_GAT_BF3
            const Bf3 hs = split_bf3(h0, h1);
            la = mfma_bf16(wl.p2, hs.p0, la); la = mfma_bf16(wl.p0, hs.p2, la); la = mfma_bf16(wl.p1, hs.p1, la);
            la = mfma_bf16(wl.p1, hs.p0, la); la = mfma_bf16(wl.p0, hs.p1, la); la = mfma_bf16(wl.p0, hs.p0, la);
#else
            la = mma_block(wl[0], h0, la);
            la = mma_block(wl[1], h1, la);
#endif
            if (g == 0 && valid) {
                s_pl[dir][node][s_prev][0] = la[0];
                s_pl[dir][node][s_prev][1] = la[1];
            }
        }
    }
    __syncthreads();
    if (clk && threadIdx.x == 0) clk[2] = IPLAN_CLOCK();

    // ---------------------------------------------------------------- phase 3: gated soft attention
    // Wave `tile` (dir 0) owns 16 egos; everything is a tiny GEMM in the D layout with the ego as the COLUMN, so
    // the softmax over an ego's neighbours is lane-local plus a reduction over the 4 lane groups:
    //   S^T[j][i] = k_j . q_i            A = k rows (LDS), B = q rows of the ego tile (LDS)        32 MFMAs
    //   w[i][j]   = softmax_j(S) * gumbel-gate(i, j)      lane (ego n, g) holds j = 16T + 4g + q
    //   x[i][c]   = sum_j w[i][j] v[j][c]  A = w straight from these registers, B = v rows (LDS)    32 MFMAs
    // (phase 4's recurrent operand: requested here, it lands while phase 3 runs)
    f32x4 hp4[2];
    {
        const float* hrow4 = a.h_prev + (int64_t)net * a.h_s_net + (int64_t)b * a.h_s_b + (int64_t)imin(node, N - 1) * GH + 4 * g;
        for (int T = 0; T < 2; ++T) hp4[T] = *(const IPLAN_GLOBAL_AS f32x4*)(hrow4 + 16 * T);
    }
    // Both waves of an ego tile work (round 4; the direction-1 wave used to idle here for 9.4 us): wave `dir` takes the neighbour tiles
    // T = 2 dir, 2 dir + 1 (j in [32 dir, 32 dir + 32)) -- its share of the scores, of the gumbel gates (half of the phase: 16 pairs per
    // lane with two exponentials and four LDS reads each) and of the aggregation.  The two halves of an ego's softmax meet once in LDS
    // (running maximum + partial denominator, combined in direction order), the two partial aggregates in phase 4.
    const int T0 = 2 * dir;
    f32x4 sc[2], aw[2];
    float gz0[2][4], gz1[2][4];
    float m_own = -INFINITY, den_own = 0.f;
    const int i = node;                                      // this lane's ego (column)
    const int NT = (N + 15) / 16;
    if (tile_live) {
        const float* noise = a.noise + sb * N * (N - 1) * 2;
        // gumbel noise of the lane's 8 (ego, neighbour) pairs: issued first, consumed after the score GEMM
        for (int T = 0; T < 2; ++T)
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * (T0 + T) + 4 * g + q;
                const bool ok = valid && j < N && j != i;
                const int sidx = j - (j > i ? 1 : 0);
                const float* gz = noise + ((int64_t)i * (N - 1) + (ok ? sidx : 0)) * 2;
                gz0[T][q] = ok ? gz[0] : 0.f;
                gz1[T][q] = ok ? gz[1] : 0.f;
            }
        GAT_SUBCLK(5);
        for (int T = 0; T < 2; ++T) sc[T] = splat4(0.f);
        for (int ks = 0; ks < GH / 4; ++ks) {
            const float qv = s_q[node][4 * ks + g];
            for (int T = 0; T < 2; ++T)
                if (T0 + T < NT) sc[T] = mfma4(s_k[16 * (T0 + T) + n][4 * ks + g], qv, sc[T]);
        }
        GAT_SUBCLK(6);
        for (int T = 0; T < 2; ++T)
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * (T0 + T) + 4 * g + q;
                const bool ok = j < N && j != i;
                sc[T][q] = ok ? sc[T][q] / 5.656854249492381f : -INFINITY;     // / sqrt(attention_dim)  (GAT_Net.py:126)
                m_own = fmaxf(m_own, sc[T][q]);
            }
        m_own = fmaxf(m_own, __shfl_xor(m_own, 16));
        m_own = fmaxf(m_own, __shfl_xor(m_own, 32));
        for (int T = 0; T < 2; ++T)
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * (T0 + T) + 4 * g + q;
                const float e = (j < N && j != i && m_own > -INFINITY) ? expf(sc[T][q] - m_own) : 0.f;
                aw[T][q] = e;
                den_own += e;
            }
        den_own = group_sum(den_own);
        if (g == 0) { s_sm[dir][tile][n][0] = m_own; s_sm[dir][tile][n][1] = den_own; }
    }
    __syncthreads();
    if (tile_live) {
        // softmax over ALL neighbours of the ego: m = max of the halves, den = den_0 e^(m_0 - m) + den_1 e^(m_1 - m) (direction order)
        const float m0 = s_sm[0][tile][n][0], d0 = s_sm[0][tile][n][1], m1 = s_sm[1][tile][n][0], d1 = s_sm[1][tile][n][1];
        const float m = fmaxf(m0, m1);
        const float den = (m0 > -INFINITY ? d0 * expf(m0 - m) : 0.f) + (m1 > -INFINITY ? d1 * expf(m1 - m) : 0.f);
        const float resc = m_own > -INFINITY ? expf(m_own - m) : 0.f;
        const float hb0 = P[a.off[IPLAN_GAT_HARD_B]], hb1 = P[a.off[IPLAN_GAT_HARD_B] + 1];
        GAT_SUBCLK(7);
        for (int T = 0; T < 2; ++T)
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * (T0 + T) + 4 * g + q;
                const bool ok = valid && j < N && j != i;
                const int sidx = j - (j > i ? 1 : 0);
                const float soft = aw[T][q] * resc / den;
                float hard = 0.f;
                if (ok) {
                    const float l0 = hb0 + s_pl[0][i][sidx][0] + s_pl[1][i][sidx][0];
                    const float l1 = hb1 + s_pl[0][i][sidx][1] + s_pl[1][i][sidx][1];
                    const float y0 = (l0 + gz0[T][q]) / a.tau, y1 = (l1 + gz1[T][q]) / a.tau;   // gumbel_softmax, GAT_Net.py:93
                    const float mm = fmaxf(y0, y1);
                    const float e0 = expf(y0 - mm), e1 = expf(y1 - mm);
                    hard = e1 / (e0 + e1);
                    if (sv.soft) sv.soft[(sb * N + i) * (N - 1) + sidx] = soft;
                    if (sv.hard) sv.hard[(sb * N + i) * (N - 1) + sidx] = hard;
                }
                aw[T][q] = ok ? soft * hard : 0.f;            // no renormalisation (GAT_Net.py:132)
            }
        GAT_SUBCLK(8);
        float (*dstx)[QST] = dir ? s_x1 : s_x;               // this direction's partial aggregate; phase 4 adds the two
        for (int ct = 0; ct < 2; ++ct) {
            f32x4 xa = splat4(0.f);
            for (int T = 0; T < 2; ++T)
                if (T0 + T < NT)
                    for (int q = 0; q < 4; ++q) {
                        const int j = 16 * (T0 + T) + 4 * g + q;
                        xa = mfma4(aw[T][q], s_v[j < N ? j : 0][16 * ct + n], xa);
                    }
            // D layout: lane (c = n, g) holds x[ego 16 tile + 4g + q][16 ct + n]
            for (int q = 0; q < 4; ++q) {
                const int e = 16 * tile + 4 * g + q;
                if (e < N) dstx[e][16 * ct + n] = xa[q];
            }
        }
    }
    GAT_SUBCLK(9);
    // (phase 4's weight fragments -- output tile `dir` of the GRUCell -- are requested before the barrier: they land while the slower
    // waves of the scene finish phase 3)
    f32x4 w4i[3][2], w4c[3][2];
    {
        const float* Wi = P + a.off[IPLAN_GAT_C_WIH];
        const float* Wc = P + a.off[IPLAN_GAT_C_WHH];
        for (int gt = 0; gt < 3; ++gt)
            for (int T = 0; T < 2; ++T) {
                w4i[gt][T] = wfrag_a(Wi, GH, 3 * GH, gt * GH + 16 * dir, 16 * T);
                w4c[gt][T] = wfrag_a(Wc, GH, 3 * GH, gt * GH + 16 * dir, 16 * T);
            }
    }
    __syncthreads();
    if (clk && threadIdx.x == 0) clk[3] = IPLAN_CLOCK();

    // ---------------------------------------------------------------- phase 4: output GRUCell
    if (tile_live) {
        const int t = dir;                                      // wave (dir,tile) produces output tile `dir`
        f32x4 x[2], hp[2];
        for (int T = 0; T < 2; ++T) {
            x[T] = splat4(0.f);
            if (valid)
                for (int q = 0; q < 4; ++q) x[T][q] = s_x[node][16 * T + 4 * g + q] + s_x1[node][16 * T + 4 * g + q];
            if (sv.x && dir == 0) vstore(sv.x + (sb * N + node) * GH, valid, GH, T, x[T]);
            for (int q = 0; q < 4; ++q) hp[T][q] = valid ? hp4[T][q] : 0.f;
        }
        const float* bi = P + a.off[IPLAN_GAT_C_BIH];
        const float* bc = P + a.off[IPLAN_GAT_C_BHH];
        f32x4 pr = bfrag(bi, 3 * GH, t) + bfrag(bc, 3 * GH, t);
        f32x4 pz = bfrag(bi, 3 * GH, 2 + t) + bfrag(bc, 3 * GH, 2 + t);
        f32x4 gn = bfrag(bi, 3 * GH, 4 + t);
        f32x4 hn = bfrag(bc, 3 * GH, 4 + t);
        for (int T = 0; T < 2; ++T) {
            pr = mma_block(w4i[0][T], x[T], pr);
            pr = mma_block(w4c[0][T], hp[T], pr);
            pz = mma_block(w4i[1][T], x[T], pz);
            pz = mma_block(w4c[1][T], hp[T], pz);
            gn = mma_block(w4i[2][T], x[T], gn);
            hn = mma_block(w4c[2][T], hp[T], hn);
        }
        const GruGates o = gru_gates(pr, pz, gn, hn, hp[t]);
        float* orow = a.out + (int64_t)net * a.out_s_net + (int64_t)b * a.out_s_b + (int64_t)node * GH;
        vstore_c<COH>(orow, valid, GH, t, o.h);
        if (sv.cell) {
            float* row = sv.cell + (sb * N + node) * (4 * GH);
            vstore(row, valid, GH, t, o.r);
            vstore(row + GH, valid, GH, t, o.z);
            vstore(row + 2 * GH, valid, GH, t, o.n);
            vstore(row + 3 * GH, valid, GH, t, o.hn);
        }
    }
    if (clk) {
        __syncthreads();
        if (threadIdx.x == 0) clk[4] = IPLAN_CLOCK();
    }
}

template <bool FOLD>
__global__ __launch_bounds__(512) void gat_fwd_kernel(IplanGatFwdArgs a) {
    __shared__ __attribute__((aligned(16))) GatShared sh;
    gat_fwd_block<false, FOLD>(a, (int)blockIdx.x, sh);
}

// The rollout's vector step: GAT_latent_update and the behaviour encoder's latent_update read the PREVIOUS latents, are
// independent of each other and were two launches on two streams -- whose workgroups the hardware dispatched in either order:
// when the encoder's 140 small workgroups got onto the CUs first, GAT's 160 whole-CU workgroups waited for them (139 us
// instead of 108 us in the kernel trace), and every step paid two cross-stream event round trips.  One launch instead: the
// first n_nets * B workgroups are the GAT scenes (dispatched first, one CU each), the following ones run the encoder, eight
// 16-row tiles each, on the CUs that are left (its LDS is the head of the same allocation).
template <bool FOLD>
__global__ __launch_bounds__(512) void gat_enc_fwd_kernel(IplanGatFwdArgs a, IplanEncFwdArgs e, int n_gat, int enc_blocks_per_net) {
    __shared__ __attribute__((aligned(16))) GatShared sh;
    static_assert(sizeof(GatShared) >= sizeof(float) * ENC_LDS_FLOATS, "encoder LDS must fit into the scene's");
    const int block = (int)blockIdx.x;
    if (block < n_gat) {
        gat_fwd_block<false, FOLD>(a, block, sh);
    } else {
        const int j = block - n_gat, net = j / enc_blocks_per_net, tb = j - net * enc_blocks_per_net;
        enc_fwd_block(e, net, tb * 8 + wave_id(), reinterpret_cast<float*>(&sh));
    }
}

// ... and the NEXT step's action selection behind both (runners/ippo_parallel_runner.py:166-268: select_actions of step t + 1
// reads exactly what the latent updates of step t write, and nothing sits between them -- the environment steps AFTER the
// action selection).  The actor/critic workgroups are the last ones of the grid: they stage their tail weights, compute
// W_hh h and the history block of the fc1 contraction while the scenes run, wait until every scene and encoder workgroup has
// counted itself into sync[0] (ac_fwd_body.h), and go on with the latent blocks.  One launch boundary per vector step instead of
// two, and the action selection's prologue off the critical path.
union GatAcShared {
    GatShared gat;
    AcShared<1> ac;
};

template <bool FOLD>
__global__ __launch_bounds__(512) void gat_enc_ac_fwd_kernel(IplanGatFwdArgs a, IplanEncFwdArgs e, IplanAcFwdArgs c, int n_gat, int enc_blocks_per_net,
                                                             int n_enc, int ac_gx, int32_t* sync) {
    __shared__ __attribute__((aligned(16))) GatAcShared sh;
    static_assert(sizeof(GatShared) >= sizeof(float) * ENC_LDS_FLOATS, "encoder LDS must fit into the scene's");
    const int block = (int)blockIdx.x;
    if (block < n_gat) {
        gat_fwd_block<true, FOLD>(a, block, sh.gat);
        ac_signal_producer_done(sync);
    } else if (block < n_gat + n_enc) {
        const int j = block - n_gat, net = j / enc_blocks_per_net, tb = j - net * enc_blocks_per_net;
        enc_fwd_block<true>(e, net, tb * 8 + wave_id(), reinterpret_cast<float*>(&sh));
        ac_signal_producer_done(sync);
    } else {
        const int j = block - n_gat - n_enc, gy = c.n_agents, gz = c.which == 2 ? 2 : 1;
        const AcGrid gp = {j % ac_gx, (j / ac_gx) % gy, j / (ac_gx * gy), ac_gx, gy, gz};
        ac_fwd_body<1, false, true>(c, gp, sh.ac, AcProducers{sync, n_gat + n_enc, ac_gx * gy * gz});
    }
}

}  // namespace iplan

static int check_gat(const IplanGatFwdArgs* a, const char* what);

extern "C" int iplan_gat_enc_fwd(const IplanGatFwdArgs* a, const IplanEncFwdArgs* e, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_gat(a, "iplan_gat_enc_fwd")) return rc;
    if (!e) return fail(IPLAN_EINVAL, "iplan_gat_enc_fwd: null encoder args");
    if (e->d < 1 || e->d > 16 || e->Z < 1 || e->Z > 16 || e->L < 1 || e->n_nets < 1 || e->B < 1 || e->N < 1)
        return fail(IPLAN_EINVAL, "iplan_gat_enc_fwd: unsupported encoder dims d=%d Z=%d L=%d", e->d, e->Z, e->L);
    if (!e->x || !e->h0 || !e->hL || !e->latent_out || !e->params)
        return fail(IPLAN_EINVAL, "iplan_gat_enc_fwd: null encoder tensor pointer");
    const int n_gat = a->n_nets * a->B, per_net = (e->B * e->N + 127) / 128;
    if (a->saved.gru) hipLaunchKernelGGL(gat_enc_fwd_kernel<false>, dim3((unsigned)(n_gat + per_net * e->n_nets)), dim3(512), 0, (hipStream_t)stream, *a, *e, n_gat, per_net);
    else hipLaunchKernelGGL(gat_enc_fwd_kernel<true>, dim3((unsigned)(n_gat + per_net * e->n_nets)), dim3(512), 0, (hipStream_t)stream, *a, *e, n_gat, per_net);
    return check_launch("iplan_gat_enc_fwd");
}

extern "C" int iplan_gat_enc_ac_fwd(const IplanGatFwdArgs* a, const IplanEncFwdArgs* e, const IplanAcFwdArgs* c, int32_t* sync, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_gat(a, "iplan_gat_enc_ac_fwd")) return rc;
    if (!e || !c || !sync) return fail(IPLAN_EINVAL, "iplan_gat_enc_ac_fwd: null encoder / actor-critic args or sync");
    if (e->d < 1 || e->d > 16 || e->Z < 1 || e->Z > 16 || e->L < 1 || e->n_nets < 1 || e->B < 1 || e->N < 1)
        return fail(IPLAN_EINVAL, "iplan_gat_enc_ac_fwd: unsupported encoder dims d=%d Z=%d L=%d", e->d, e->Z, e->L);
    if (!e->x || !e->h0 || !e->hL || !e->latent_out || !e->params)
        return fail(IPLAN_EINVAL, "iplan_gat_enc_ac_fwd: null encoder tensor pointer");
    if (int rc = ac_fwd_check(c)) return rc;
    if (c->ksplit != 8 || c->saved || c->fc1_pre || c->ln_stats_mode != 0)
        return fail(IPLAN_EINVAL, "iplan_gat_enc_ac_fwd: the actor/critic part must be a rollout-shaped launch (ksplit 8, one-pass "
                                  "LayerNorm statistics, nothing saved)");
    const int n_gat = a->n_nets * a->B, per_net = (e->B * e->N + 127) / 128, n_enc = per_net * e->n_nets;
    const int tiles = (c->rows + 15) / 16, ac_gx = tiles * (c->ksplit_wg > 1 ? c->ksplit_wg : 1);
    const int n_ac = ac_gx * c->n_agents * (c->which == 2 ? 2 : 1);
    if (a->saved.gru) hipLaunchKernelGGL(gat_enc_ac_fwd_kernel<false>, dim3((unsigned)(n_gat + n_enc + n_ac)), dim3(512), 0, (hipStream_t)stream, *a, *e, *c, n_gat, per_net, n_enc, ac_gx, sync);
    else hipLaunchKernelGGL(gat_enc_ac_fwd_kernel<true>, dim3((unsigned)(n_gat + n_enc + n_ac)), dim3(512), 0, (hipStream_t)stream, *a, *e, *c, n_gat, per_net, n_enc, ac_gx, sync);
    return check_launch("iplan_gat_enc_ac_fwd");
}

static int check_gat(const IplanGatFwdArgs* a, const char* what) {
    using namespace iplan;
    if (!a) return fail(IPLAN_EINVAL, "%s: null args", what);
    if (a->N < 2 || a->N > IPLAN_MAX_ENTITIES)
        return fail(IPLAN_EINVAL, "%s: N=%d outside [2,%d]", what, a->N, IPLAN_MAX_ENTITIES);
    if (a->n_nets < 1 || a->B < 1 || a->d0 < 1 || a->d1 < 0)
        return fail(IPLAN_EINVAL, "%s: bad dims n_nets=%d B=%d d0=%d d1=%d", what, a->n_nets, a->B, a->d0, a->d1);
    if (!a->src0 || (a->d1 > 0 && !a->src1) || !a->h_prev || !a->out || !a->noise || !a->params)
        return fail(IPLAN_EINVAL, "%s: null tensor pointer", what);
    if (!aligned16(a->h_prev) || !aligned16(a->out) || (a->h_s_net & 3) || (a->h_s_b & 3) ||
        (a->out_s_net & 3) || (a->out_s_b & 3))
        return fail(IPLAN_EALIGN, "%s: h_prev/out must be 16-byte aligned with strides %% 4 == 0", what);
    return IPLAN_OK;
}

extern "C" int iplan_gat_fwd(const IplanGatFwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_gat(a, "iplan_gat_fwd")) return rc;
    if (a->saved.gru) hipLaunchKernelGGL(gat_fwd_kernel<false>, dim3((unsigned)(a->n_nets * a->B)), dim3(512), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(gat_fwd_kernel<true>, dim3((unsigned)(a->n_nets * a->B)), dim3(512), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_gat_fwd");
}
