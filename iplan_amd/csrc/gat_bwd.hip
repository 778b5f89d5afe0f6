// Backward of the GAT-RNN encoder (autograd of GAT_Net.forward, nova/GAT_Net.py:41-142, under
// Prediction_policy.learn's loss.backward(), nova/prediction_policy.py:228) -- one 512-thread
// workgroup per (net, env) scene, same ownership as the forward kernel (gat.hip):
//   A  GRUCell'            rows of the scene, MFMA with transposed weight fragments -> dx
//   B  per ego  (wave):    t_s = dx_i . v_j ; softmax' ; gumbel-gate' (x 1/tau) ; dq_i
//   C  per node (wave):    the scatter side dk_j, dv_j as gathers over the N-1 egos that see j
//   D  BPTT of the bidirectional pair GRU: wave (dir, tile) walks its 16 ego chains backwards,
//      W_hh^T in registers as MFMA A fragments, gate math lane-local, the hard-gate gradient
//      injected per step; dW_hh / db_hh accumulate in-kernel (12 register tiles, operands turned through
//      LDS); the input-side gate gradients [dr dz dn_i] go to a scratch laid out for phase E
//   E  per node: d(W_b h_j) = gather-sum of the pair-step gradients of the egos that saw j
//   F  node projections': dh_enc = W_a^T da + W_b^T db + W_q^T dq + W_k^T dk + W_v^T dv, ReLU'
// Node-level pre-activation gradients leave the kernel (their weight gradients are dY^T X contractions,
// wgrad.hip); the pair-level W_hh gradient is accumulated here -- streaming the pair-step gradients out
// and back in for it was a third of the kernel pair's HBM traffic.
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int BH = IPLAN_GAT_HIDDEN;   // H == A == 32
constexpr int BNP = IPLAN_MAX_ENTITIES;
constexpr int BQS = 33;
constexpr int DY = IPLAN_GAT_NODE_DY;
constexpr int DY_DA = 32, DY_DB = 128, DY_DIR = 192, DY_DQ = 416, DY_DK = 448, DY_DV = 480, DY_CELL = 512;

__device__ __forceinline__ float chain_sum16(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

__global__ __launch_bounds__(512) void gat_bwd_kernel(IplanGatBwdArgs a) {
    __shared__ float s_q[BNP][BQS];
    __shared__ float s_k[BNP][BQS];
    __shared__ float s_v[BNP][BQS];
    __shared__ float s_dx[BNP][BQS];
    // d score and soft*hard per (ego, step) for phases B / C; phase D re-uses the 32 KB for W_hh^T of both directions
    // ([H][3H + 8]: conflict-free ds_read_b128 fragments) -- in registers they put the step loop 24 VGPRs over budget
    __shared__ __attribute__((aligned(16))) float s_sw[2][BNP][BNP];
    float (*s_ds)[BNP] = s_sw[0];
    float (*s_w)[BNP] = s_sw[1];
    constexpr int WLD = 3 * BH + 8;
    static_assert(2 * BH * WLD <= 2 * BNP * BNP, "W_hh^T copies must fit the score buffers");
    __shared__ float s_dd[BNP][BNP];
    __shared__ __attribute__((aligned(16))) f32x4 s_acc[4][2][64];
    __shared__ float s_red[8];
    __shared__ __attribute__((aligned(16))) float s_bhn[2][BH];
    __shared__ __attribute__((aligned(16))) float s_turn[8][8][256];       // per wave: 6 gate-gradient tiles + 2 h_prev tiles (phase D)

    const IplanGatFwdArgs& f = a.fwd;
    const IplanGatSaved& sv = f.saved;
    const int net = (int)blockIdx.x / f.B;
    const int b = (int)blockIdx.x % f.B;
    const int N = f.N;
    const float* __restrict__ P = f.params + (int64_t)net * f.params_s_net;
    const int l = lane_id(), w = wave_id();
    const int n = l & 15, g = l >> 4;
    const int tile = w & 3, dir = w >> 2;
    const int node = 16 * tile + n;
    const bool tile_live = 16 * tile < N;
    const bool valid = node < N;
    const int64_t sb = (int64_t)net * f.B + b;
    float* __restrict__ ndy = a.node_dy + sb * N * DY;
    float* __restrict__ hpart = a.hard_part + sb * IPLAN_GAT_HARD_PART;

    // optional phase clocks of workgroup 0 (slots 8..14 of the forward's profiling buffer: entry, A, B, C, D, E, F)
    int64_t* __restrict__ clk = (f.phase_clocks && blockIdx.x == 0 && threadIdx.x == 0) ? f.phase_clocks + 8 : nullptr;
    if (clk) clk[0] = IPLAN_CLOCK();
    // q, k, v of the scene -> LDS
    for (int idx = (int)threadIdx.x; idx < N * 3 * BH; idx += (int)blockDim.x) {
        const int nd = idx / (3 * BH), c = idx - nd * 3 * BH;
        const float val = sv.qkv[(sb * N + nd) * 3 * BH + c];
        if (c < BH) s_q[nd][c] = val;
        else if (c < 2 * BH) s_k[nd][c - BH] = val;
        else s_v[nd][c - 2 * BH] = val;
    }
    // ---------------------------------------------------------------- A: output GRUCell backward
    if (tile_live && dir == 0) {
        const float* crow = sv.cell + (sb * N + node) * (4 * BH);
        const float* hrow = f.h_prev + (int64_t)net * f.h_s_net + (int64_t)b * f.h_s_b + (int64_t)node * BH;
        const float* grow = a.g_out + (int64_t)net * a.g_s_net + (int64_t)b * a.g_s_b + (int64_t)node * BH;
        float* drow = ndy + (int64_t)node * DY + DY_CELL;
        f32x4 dgc[6];
        for (int T = 0; T < 2; ++T) {
            const GruGrads o = gru_gates_bwd(vload(grow, valid, BH, T), vload(crow, valid, BH, T), vload(crow + BH, valid, BH, T),
                                             vload(crow + 2 * BH, valid, BH, T), vload(crow + 3 * BH, valid, BH, T),
                                             vload(hrow, valid, BH, T));
            dgc[T] = o.dr; dgc[2 + T] = o.dz; dgc[4 + T] = o.dni;
            vstore(drow, valid, BH, T, o.dr);
            vstore(drow + BH, valid, BH, T, o.dz);
            vstore(drow + 2 * BH, valid, BH, T, o.dni);
            vstore(drow + 3 * BH, valid, BH, T, o.dnh);
        }
        const float* Wi = P + f.off[IPLAN_GAT_C_WIH];
        for (int T = 0; T < 2; ++T) {
            const f32x4 dx = dense_tile_gt<6>(Wi, BH, 3 * BH, BH, 16 * T, dgc, splat4(0.f));
            if (valid)
                for (int q = 0; q < 4; ++q) s_dx[node][16 * T + 4 * g + q] = dx[q];
        }
    }
    __syncthreads();
    if (clk) clk[1] = IPLAN_CLOCK();

    // ---------------------------------------------------------------- B: attention backward per ego
    {
        float ddsum = 0.f;
        for (int i = w; i < N; i += 8) {
            const int s = l;
            const bool live = s < N - 1;
            const int j = live ? s + (s >= i ? 1 : 0) : 0;
            float soft = 0.f, hard = 0.f;
            if (live) {
                soft = sv.soft[(sb * N + i) * (N - 1) + s];
                hard = sv.hard[(sb * N + i) * (N - 1) + s];
            }
            float t = 0.f;
            for (int c = 0; c < BH; ++c) t = fmaf(s_dx[i][c], s_v[j][c], t);
            const float dsoft = t * hard, dhard = t * soft;
            const float dot = wave_sum(live ? soft * dsoft : 0.f);
            const float ds = live ? soft * (dsoft - dot) / 5.656854249492381f : 0.f;     // d score, incl. 1/sqrt(A)
            const float dd = live ? dhard * hard * (1.0f - hard) / f.tau : 0.f;          // d(l1 - l0)
            if (live) {
                s_ds[i][s] = ds;
                s_w[i][s] = soft * hard;
                s_dd[i][s] = dd;
            }
            ddsum += wave_sum(dd);
            const int c = l & 31, hf = l >> 5;
            float acc = 0.f;
            for (int it = 0; 2 * it < N - 1; ++it) {
                const int s2 = 2 * it + hf;
                const float dsv = __shfl(ds, s2);
                const int j2 = s2 < N - 1 ? s2 + (s2 >= i ? 1 : 0) : 0;
                acc = fmaf(dsv, s_k[j2][c], acc);
            }
            acc += __shfl_xor(acc, 32);
            if (l < 32) ndy[(int64_t)i * DY + DY_DQ + c] = acc;
        }
        if (l == 0) s_red[w] = ddsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += s_red[k];
        hpart[8 * BH] = t;
    }

    if (clk) clk[2] = IPLAN_CLOCK();
    // ---------------------------------------------------------------- C: dk_j, dv_j per node
    for (int j = w; j < N; j += 8) {
        const int c = l & 31, hf = l >> 5;
        float ak = 0.f, av = 0.f;
        for (int i = hf; i < N; i += 2) {
            if (i == j) continue;
            const int s = j - (j > i ? 1 : 0);
            ak = fmaf(s_ds[i][s], s_q[i][c], ak);
            av = fmaf(s_w[i][s], s_dx[i][c], av);
        }
        ak += __shfl_xor(ak, 32);
        av += __shfl_xor(av, 32);
        if (l < 32) {
            ndy[(int64_t)j * DY + DY_DK + c] = ak;
            ndy[(int64_t)j * DY + DY_DV + c] = s_v[j][c] > 0.f ? av : 0.f;       // v = ReLU(.)
        }
    }

    __syncthreads();                                                // phases B / C are done with the score buffers
    {
        float* swT = &s_sw[0][0][0];
        stage_matrix_t(swT, WLD, BH, P + f.off[IPLAN_GAT_F_WHH], 3 * BH, BH);
        stage_matrix_t(swT + BH * WLD, WLD, BH, P + f.off[IPLAN_GAT_R_WHH], 3 * BH, BH);
        if (threadIdx.x < 2 * BH)                                  // b_hn of both directions (the recomputed gh_n's bias)
            s_bhn[threadIdx.x >> 5][threadIdx.x & 31] = P[f.off[(threadIdx.x >> 5) ? IPLAN_GAT_R_BHH : IPLAN_GAT_F_BHH] + 2 * BH + (threadIdx.x & 31)];
    }
    __syncthreads();
    if (clk) clk[3] = IPLAN_CLOCK();
    // ---------------------------------------------------------------- D: BPTT through the pair GRU
    // dgru: kernel-private scratch, [net][dir][scene][ego tile][step][class 2][3H].  A pair step (ego i, step s) belongs to neighbour
    // j = s + [s >= i], so the 16 egos of a tile feed at most TWO neighbours per step -- class 0: egos i <= s -> j = s + 1; class 1:
    // egos i > s -> j = s -- and d(W_b h_j) is a SUM over the egos that see j.  Phase D therefore adds a step's [dr dz dn_i] over the
    // tile's chains per class (one more MFMA against a 0 / 1 operand on the tiles it parks for dW_hh anyway) and stores two
    // 3H-float rows per wave-step; phase E adds at most eight rows per (node, direction).  Until round 5 every pair step's 3H
    // floats went out to a neighbour-major scratch and came back in phase E: 2 x 2.3 MB per scene of a kernel that runs AT the
    // HBM bandwidth (3.8 MB of gate records in + the scratch, 4.5 TB/s over phases D + E: profiles/r05_notes.md).
    const int NT_live = (N + 15) / 16;
    float* __restrict__ dgp = a.dgru + ((((int64_t)net * 2 + dir) * f.B + b) * NT_live + tile) * (int64_t)(N - 1) * (2 * 3 * BH);
    if (tile_live) {
        const float* swT = &s_sw[0][0][0] + dir * (BH * WLD);                      // W_hh^T [H][3H] of this direction
        const float* Wh = P + f.off[IPLAN_GAT_HARD_W];                              // [2][2H]
        f32x4 wdiff[2];
        for (int T = 0; T < 2; ++T) wdiff[T] = bfrag(Wh + 2 * BH + dir * BH, BH, T) - bfrag(Wh + dir * BH, BH, T);
        // The forward's record of a pair step (h_s, r, z, n, hn) and of the step the forward came from (h of it = h_prev).
        // Fetched one step ahead with plain 16-byte loads -- the former predicated, alignment-checked loads each ended in a
        // full wait: twelve serial L2 / HBM round trips per step (8.6 us per step, profiles/r02e_notes.md).  A lane without a
        // chain (the ragged last tile) is a CLONE of the scene's last node: it loads, computes and stores exactly what that
        // node's lane does (same values to the same addresses), so nothing in the step loop is predicated; only the sums
        // over the 16 chains below leave the clones out.
        // Record layout (gat.hip writes it): [net][dir][scene][ego tile][step][group: h h r r z z n n][16 chains][16 columns] (REC = 8 groups x 256 floats;
        // gh_n = W_hn h_prev + b_hn is recomputed below, not stored) --
        // a wave's load of one 16-column group of its 16 chains is ONE contiguous 1 KiB block.  (Chain-major rows of 5H floats until
        // round 5: every load instruction was 16 separate 64-byte pieces, and the BPTT sat at the memory pipeline's request rate, not
        // at its latency -- touching the lines three steps ahead made it SLOWER, 345 -> 397 us: profiles/r05_notes.md.)
        const int cnode = imin(node, N - 1);
        const int NT = (N + 15) / 16;
        constexpr int REC = 8 * 256;                                                // floats of a tile's record of one step (h r z n)
        const float* gbase = sv.gru + (((((int64_t)net * 2 + dir) * f.B + b) * NT + tile) * (int64_t)(N - 1)) * REC + (cnode & 15) * 16 + 4 * g;
        float (*turn)[256] = s_turn[w];
        // a parked tile is [16 chains][16 columns], chain r's columns rotated by 4 (r >> 1) (conflict-free ds_write_b128 /
        // ds_read_b32, as in beh_enc_bwd_kernel); pick = operand element [chain 4 s + g][column n]
        auto park = [&](int slot, f32x4 v) { *reinterpret_cast<f32x4*>(&turn[slot][n * 16 + ((4 * g + 4 * (n >> 1)) & 15)]) = v; };
        auto pick = [&](int slot, int s4) { const int r = 4 * s4 + g; return turn[slot][r * 16 + ((n + 4 * (r >> 1)) & 15)]; };
        f32x4 aW[6][2], bNh[2];                                     // dW_hh [3H x H] in 12 register tiles; lane-local sums of dn_h
        for (int t = 0; t < 6; ++t) { aW[t][0] = splat4(0.f); aW[t][1] = splat4(0.f); }   // (db_hh's r, z parts = the da sums below)
        bNh[0] = splat4(0.f); bNh[1] = splat4(0.f);
        struct PairIn {
            f32x4 hs[2], r[2], z[2], nn[2], hp[2];
            float dd;
            bool has_prev;
        };
        auto load_step = [&](int it, PairIn& o, bool first) {
            const int s = dir ? it : (N - 2 - it);                 // reverse of the forward visiting order
            const int sp = dir ? s + 1 : s - 1;                    // the step the forward came from
            o.has_prev = sp >= 0 && sp <= N - 2;
            const float* row = gbase + (int64_t)s * REC;
            const float* prow = gbase + (int64_t)(o.has_prev ? sp : s) * REC;
            for (int T = 0; T < 2; ++T) {
                if (first) o.hs[T] = *reinterpret_cast<const f32x4*>(row + 256 * T);     // later: the previous step's hp
                o.r[T] = *reinterpret_cast<const f32x4*>(row + 256 * (2 + T));
                o.z[T] = *reinterpret_cast<const f32x4*>(row + 256 * (4 + T));
                o.nn[T] = *reinterpret_cast<const f32x4*>(row + 256 * (6 + T));
                o.hp[T] = *reinterpret_cast<const f32x4*>(prow + 256 * T);
            }
            o.dd = s_dd[cnode][s];
        };
        f32x4 dh[2], da[6], hacc[2];
        for (int T = 0; T < 2; ++T) { dh[T] = splat4(0.f); hacc[T] = splat4(0.f); }
        for (int t = 0; t < 6; ++t) da[t] = splat4(0.f);
        PairIn cur;
        load_step(0, cur, true);
        for (int it = 0; it < N - 1; ++it) {
            const int s = dir ? it : (N - 2 - it);
            const float dd = cur.dd;
            f32x4 dgh[6];
            f32x4 dhd[2];
            GruGrads o2[2];
            // the forward's gh_n = W_hn h_prev + b_hn is not in the record (a fifth of its bytes): recomputed from the previous state
            // with W_hh^T's LDS copy read transposed (16 fp32 MFMAs per step)
            f32x4 hpm[2], hnr[2];
            for (int T = 0; T < 2; ++T) hpm[T] = zero_unless(cur.has_prev, cur.hp[T]);
            for (int T = 0; T < 2; ++T) {
                f32x4 acc = *reinterpret_cast<const f32x4*>(&s_bhn[dir][16 * T + 4 * g]);
                for (int Tk = 0; Tk < 2; ++Tk) {
                    f32x4 wf;
                    for (int q = 0; q < 4; ++q) wf[q] = swT[(16 * Tk + 4 * g + q) * WLD + 2 * BH + 16 * T + n];
                    acc = mma_block(wf, hpm[Tk], acc);
                }
                hnr[T] = acc;
            }
            for (int T = 0; T < 2; ++T) {
                const f32x4 hs = cur.hs[T];
                const f32x4 hp = hpm[T];
                f32x4 dht;
                for (int q = 0; q < 4; ++q) {
                    dht[q] = fmaf(wdiff[T][q], dd, dh[T][q]);
                    hacc[T][q] = fmaf(dd, hs[q], hacc[T][q]);
                }
                o2[T] = gru_gates_bwd(dht, cur.r[T], cur.z[T], cur.nn[T], hnr[T], hp);
                da[T] += o2[T].dr; da[2 + T] += o2[T].dz; da[4 + T] += o2[T].dni;
                dgh[T] = o2[T].dr; dgh[2 + T] = o2[T].dz; dgh[4 + T] = o2[T].dnh;
                dhd[T] = o2[T].dh_direct;
            }
            // dW_hh += [dr dz dn_h]^T h_prev over the tile's 16 chains (clones contribute zero): operands turned through LDS
            for (int t = 0; t < 6; ++t) park(t, dgh[t]);
            bNh[0] += dgh[4]; bNh[1] += dgh[5];
            for (int T = 0; T < 2; ++T) park(6 + T, zero_unless(valid && cur.has_prev, cur.hp[T]));
            IPLAN_WAVE_SYNC();
            IPLAN_SCHED_FENCE();
            // next step's record under this step's 48 MFMAs; its h is this step's h_prev (the BPTT walks the forward's order
            // backwards), already in registers
            for (int T = 0; T < 2; ++T) cur.hs[T] = cur.hp[T];
            load_step(it + 1 < N - 1 ? it + 1 : it, cur, false);
            IPLAN_SCHED_FENCE();
            for (int T = 0; T < 2; ++T) {
                f32x4 acc = dhd[T];
                for (int t = 0; t < 6; ++t) acc = mma_block(wfrag_lds(swT, WLD, 16 * T, 16 * t), dgh[t], acc);
                dh[T] = acc;
            }
            // class sums of [dr dz dn_i] over the tile's chains (header of this phase): B operand = the 0 / 1 class membership of chain
            // 4 s4 + g for class n (n < 2), A = the parked gate-gradient tile -> lane (n = class, g) ends with the sums of columns 4g ..
            for (int s4 = 0; s4 < 4; ++s4) {
                const float h0 = pick(6, s4), h1 = pick(7, s4);
                for (int t = 0; t < 6; ++t) {
                    const float av = pick(t, s4);
                    aW[t][0] = mfma4(av, h0, aW[t][0]);
                    aW[t][1] = mfma4(av, h1, aW[t][1]);
                }
            }
            // (two gate tiles at a time, in passes of their own: with all six sums live beside the dW_hh accumulators the loop spilled)
            auto cmask = [&](int s4) {
                const int ego = 16 * tile + 4 * s4 + g;
                return (ego < N && n < 2 && (n == 0 ? ego <= s : ego > s)) ? 1.0f : 0.0f;
            };
            float* crow = dgp + ((int64_t)s * 2 + (n & 1)) * (3 * BH) + 4 * g;
            auto class_rows = [&](int t0) {
                f32x4 c0 = splat4(0.f), c1 = splat4(0.f);
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float m = cmask(s4);
                    c0 = mfma4(pick(t0, s4), m, c0);
                    c1 = mfma4(pick(t0 + 1, s4), m, c1);
                }
                if (n < 2) {
                    *reinterpret_cast<f32x4*>(crow + 16 * t0) = c0;
                    *reinterpret_cast<f32x4*>(crow + 16 * (t0 + 1)) = c1;
                }
            };
            class_rows(0);                                                              // dr
            class_rows(2);                                                              // dz
            IPLAN_WAVE_SYNC();
            park(4, o2[0].dni);                                                         // dn_i takes dn_h's slots
            park(5, o2[1].dni);
            IPLAN_WAVE_SYNC();
            class_rows(4);
            IPLAN_WAVE_SYNC();
        }
        {   // this wave's partial of dW_hh / db_hh (reduced over scenes and tiles by gat_whh_grad_kernel, fixed order)
            float* part = a.whh_part + (((sb * 2 + dir) * 4 + tile) * (int64_t)IPLAN_GAT_WHH_PART);
            for (int t = 0; t < 6; ++t)
                for (int T = 0; T < 2; ++T)
                    for (int q = 0; q < 4; ++q) part[(16 * t + 4 * g + q) * BH + 16 * T + n] = aW[t][T][q];
            for (int t = 0; t < 6; ++t)
                for (int q = 0; q < 4; ++q) {
                    const float v = t < 4 ? da[t][q] : bNh[t - 4][q];       // sum over steps of dr | dz | dn_h of this lane's chain
                    const float sum = chain_sum16(valid ? v : 0.f);
                    if (n == 0) part[3 * BH * BH + 16 * t + 4 * g + q] = sum;
                }
        }
        float* arow = ndy + (int64_t)node * DY + DY_DA + dir * DY_DIR;
        for (int t = 0; t < 6; ++t) vstore(arow, valid, 3 * BH, t, da[t]);
        for (int T = 0; T < 2; ++T)
            for (int q = 0; q < 4; ++q) {
                const float sum = chain_sum16(valid ? hacc[T][q] : 0.f);
                if (n == 0) hpart[(dir * 4 + tile) * BH + 16 * T + 4 * g + q] = sum;
            }
    } else {
        if (l < BH) hpart[(dir * 4 + tile) * BH + l] = 0.f;
    }
    __syncthreads();
    if (clk) clk[4] = IPLAN_CLOCK();

    // ---------------------------------------------------------------- E: d(W_b h_j) per node = the class rows of the steps that see it
    // node j is seen at step s = j by the egos i > j (class 1) and at step s = j - 1 by the egos i < j (class 0): at most two rows per
    // live ego tile, added in tile order (fixed summation order)
    for (int p = w; p < 2 * N; p += 8) {
        const int j = p >> 1, d2 = p & 1;
        const float* base = a.dgru + ((((int64_t)net * 2 + d2) * f.B + b) * NT_live) * (int64_t)(N - 1) * (2 * 3 * BH);
        const int c1 = 64 + (l & 31);                              // (lanes >= 32 repeat lanes < 32)
        // all (at most 16) loads first, branch-free (a row that does not exist is replaced by an existing one and masked), then the sums
        float v0[8], v1[8];
        const bool hasA = j >= 1, hasB = j <= N - 2;
        for (int tl = 0; tl < 4; ++tl) {
            const float* tb = base + (int64_t)imin(tl, NT_live - 1) * (N - 1) * (2 * 3 * BH);
            const float* rowA = tb + ((int64_t)(hasA ? j - 1 : 0) * 2 + 0) * (3 * BH);
            const float* rowB = tb + ((int64_t)(hasB ? j : 0) * 2 + 1) * (3 * BH);
            v0[2 * tl] = rowA[l];     v1[2 * tl] = rowA[c1];
            v0[2 * tl + 1] = rowB[l]; v1[2 * tl + 1] = rowB[c1];
        }
        float a0 = 0.f, a1 = 0.f;
        for (int tl = 0; tl < 4; ++tl) {
            const bool live = tl < NT_live;
            a0 += (live && hasA) ? v0[2 * tl] : 0.f;     a1 += (live && hasA) ? v1[2 * tl] : 0.f;
            a0 += (live && hasB) ? v0[2 * tl + 1] : 0.f; a1 += (live && hasB) ? v1[2 * tl + 1] : 0.f;
        }
        float* brow = ndy + (int64_t)j * DY + DY_DB + d2 * DY_DIR;
        brow[l] = a0;
        if (l < 32) brow[64 + l] = a1;
    }
    __syncthreads();
    if (clk) clk[5] = IPLAN_CLOCK();

    // ---------------------------------------------------------------- F: node projections backward
    if (tile_live) {
        const float* nrow = ndy + (int64_t)node * DY;
        const float* Wih = P + f.off[dir ? IPLAN_GAT_R_WIH : IPLAN_GAT_F_WIH];      // [3H][2H]
        f32x4 da[6], db[6];
        for (int t = 0; t < 6; ++t) {
            da[t] = vload(nrow + DY_DA + dir * DY_DIR, valid, 3 * BH, t);
            db[t] = vload(nrow + DY_DB + dir * DY_DIR, valid, 3 * BH, t);
        }
        f32x4 part[2];
        for (int T = 0; T < 2; ++T) {
            f32x4 acc = splat4(0.f);
            for (int t = 0; t < 6; ++t) {
                acc = mma_block(wfrag_t(Wih, 2 * BH, 3 * BH, 2 * BH, 16 * T, 16 * t), da[t], acc);
                acc = mma_block(wfrag_t(Wih, 2 * BH, 3 * BH, 2 * BH, BH + 16 * T, 16 * t), db[t], acc);
            }
            part[T] = acc;
        }
        if (dir == 0) {
            f32x4 dq[2], dk[2];
            for (int t = 0; t < 2; ++t) { dq[t] = vload(nrow + DY_DQ, valid, BH, t); dk[t] = vload(nrow + DY_DK, valid, BH, t); }
            for (int T = 0; T < 2; ++T) {
                part[T] = dense_tile_gt<2>(P + f.off[IPLAN_GAT_Q_W], BH, BH, BH, 16 * T, dq, part[T]);
                part[T] = dense_tile_gt<2>(P + f.off[IPLAN_GAT_K_W], BH, BH, BH, 16 * T, dk, part[T]);
            }
        } else {
            f32x4 dv[2];
            for (int t = 0; t < 2; ++t) dv[t] = vload(nrow + DY_DV, valid, BH, t);
            for (int T = 0; T < 2; ++T) part[T] = dense_tile_gt<2>(P + f.off[IPLAN_GAT_V_W], BH, BH, BH, 16 * T, dv, part[T]);
            s_acc[tile][0][l] = part[0];
            s_acc[tile][1][l] = part[1];
        }
        __syncthreads();
        if (dir == 0) {
            const float* hrow = sv.h_enc + (sb * N + node) * BH;
            float* erow = ndy + (int64_t)node * DY;
            for (int T = 0; T < 2; ++T) {
                const f32x4 he = vload(hrow, valid, BH, T);
                f32x4 tot = part[T] + s_acc[tile][T][l];
                for (int q = 0; q < 4; ++q) tot[q] = he[q] > 0.f ? tot[q] : 0.f;
                vstore(erow, valid, BH, T, tot);
            }
        }
    } else {
        __syncthreads();
    }
    if (clk) clk[6] = IPLAN_CLOCK();
}

// hard_bi_GRU.weight_hh_l0{,_reverse} / bias_hh gradients <- sum of the wave partials over scenes and live tiles, fixed order.
// grid: (ceil(IPLAN_GAT_WHH_PART / 64), n_nets * 2); 512 threads = 8 waves x 64 consecutive elements: wave w adds the partial blocks
// p = w, w + 8, ... (p = scene * tiles + tile) into eight interleaved running sums (64 loads in flight per element instead of one
// dependent chain over all B * tiles blocks: 405 -> ~40 us at config 5's B = 256), the eight waves' sums meet in LDS in wave order.
__global__ __launch_bounds__(512) void gat_whh_grad_kernel(IplanGatBwdArgs a) {
    __shared__ float s_sum[8][64];
    const IplanGatFwdArgs& f = a.fwd;
    const int l = lane_id(), w = wave_id();
    const int e = (int)blockIdx.x * 64 + l;
    const int net = (int)blockIdx.y >> 1, dir = (int)blockIdx.y & 1;
    const int tiles = (f.N + 15) / 16, P = f.B * tiles;
    const int ec = e < IPLAN_GAT_WHH_PART ? e : IPLAN_GAT_WHH_PART - 1;
    auto part = [&](int p) {
        const int b = p / tiles, t = p - b * tiles;
        return a.whh_part[((((int64_t)net * f.B + b) * 2 + dir) * 4 + t) * (int64_t)IPLAN_GAT_WHH_PART + ec];
    };
    float acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    int p = w;
    for (; p + 56 < P; p += 64) {
        float v[8];
        for (int k = 0; k < 8; ++k) v[k] = part(p + 8 * k);
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    for (int k = 0; p < P; p += 8, ++k) acc[k & 7] += part(p);
    s_sum[w][l] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w != 0 || e >= 3 * BH * BH + 3 * BH) return;
    float sum = 0.f;
    for (int k = 0; k < 8; ++k) sum += s_sum[k][l];
    float* g = a.grad + (int64_t)net * a.grad_s_net;
    if (e < 3 * BH * BH) g[f.off[dir ? IPLAN_GAT_R_WHH : IPLAN_GAT_F_WHH] + e] = sum;
    else g[f.off[dir ? IPLAN_GAT_R_BHH : IPLAN_GAT_F_BHH] + (e - 3 * BH * BH)] = sum;
}

}  // namespace iplan

extern "C" int iplan_gat_bwd(const IplanGatBwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a) return fail(IPLAN_EINVAL, "iplan_gat_bwd: null args");
    const IplanGatFwdArgs& f = a->fwd;
    if (f.N < 2 || f.N > IPLAN_MAX_ENTITIES || f.n_nets < 1 || f.B < 1)
        return fail(IPLAN_EINVAL, "iplan_gat_bwd: bad dims");
    const IplanGatSaved& s = f.saved;
    if (!s.h_enc || !s.gru || !s.qkv || !s.soft || !s.hard || !s.x || !s.cell)
        return fail(IPLAN_EINVAL, "iplan_gat_bwd: the forward launch did not save its activations");
    if (!a->g_out || !a->dgru || !a->node_dy || !a->hard_part || !a->whh_part || !a->grad || !f.h_prev || !f.params)
        return fail(IPLAN_EINVAL, "iplan_gat_bwd: null tensor pointer");
    if (!aligned16(a->g_out) || (a->g_s_net & 3) || (a->g_s_b & 3) || !aligned16(a->dgru) || !aligned16(a->node_dy))
        return fail(IPLAN_EALIGN, "iplan_gat_bwd: g_out / dgru / node_dy must be 16-byte aligned");
    hipLaunchKernelGGL(gat_bwd_kernel, dim3((unsigned)(f.n_nets * f.B)), dim3(512), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(gat_whh_grad_kernel, dim3((unsigned)((IPLAN_GAT_WHH_PART + 63) / 64), (unsigned)(f.n_nets * 2)), dim3(512), 0,
                       (hipStream_t)stream, *a);
    return check_launch("iplan_gat_bwd");
}
