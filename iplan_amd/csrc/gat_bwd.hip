// Backward of the GAT-RNN encoder (autograd of GAT_Net.forward, nova/GAT_Net.py:41-142, under
// Prediction_policy.learn's loss.backward(), nova/prediction_policy.py:228) -- one 512-thread
// workgroup per (net, env) scene, same ownership as the forward kernel (gat.hip):
//   A  GRUCell'            rows of the scene, MFMA with transposed weight fragments -> dx
//   B  per ego  (wave):    t_s = dx_i . v_j ; softmax' ; gumbel-gate' (x 1/tau) ; dq_i
//   C  per node (wave):    the scatter side dk_j, dv_j as gathers over the N-1 egos that see j
//   D  BPTT of the bidirectional pair GRU: wave (dir, tile) walks its 16 ego chains backwards,
//      W_hh^T in registers as MFMA A fragments, gate math lane-local, the hard-gate gradient
//      injected per step; the per-step gate gradients stream to HBM (they are the dY operand of
//      the W_hh weight-gradient contraction, wgrad.hip)
//   E  per node: d(W_b h_j) = gather-sum of the pair-step gradients of the egos that saw j
//   F  node projections': dh_enc = W_a^T da + W_b^T db + W_q^T dq + W_k^T dk + W_v^T dv, ReLU'
// Only row-level pre-activation gradients leave the kernel; every weight gradient is a dY^T X
// contraction over them (node level for everything except W_hh).
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
    __shared__ float s_ds[BNP][BNP];
    __shared__ float s_w[BNP][BNP];
    __shared__ float s_dd[BNP][BNP];
    __shared__ __attribute__((aligned(16))) f32x4 s_acc[4][2][64];
    __shared__ float s_red[8];

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

    if (clk) clk[3] = IPLAN_CLOCK();
    // ---------------------------------------------------------------- D: BPTT through the pair GRU
    float* __restrict__ dgru_base = a.dgru + ((((int64_t)net * 2 + dir) * f.B + b) * N) * (int64_t)(N - 1) * (4 * BH);
    if (tile_live) {
        const float* Whh = P + f.off[dir ? IPLAN_GAT_R_WHH : IPLAN_GAT_F_WHH];     // [3H][H]
        f32x4 whT[2][6];
        for (int T = 0; T < 2; ++T)
            for (int t = 0; t < 6; ++t) whT[T][t] = wfrag_t(Whh, BH, 3 * BH, BH, 16 * T, 16 * t);
        const float* Wh = P + f.off[IPLAN_GAT_HARD_W];                              // [2][2H]
        f32x4 wdiff[2];
        for (int T = 0; T < 2; ++T) wdiff[T] = bfrag(Wh + 2 * BH + dir * BH, BH, T) - bfrag(Wh + dir * BH, BH, T);
        // The forward's record of a pair step (h_s, r, z, n, hn) and of the step the forward came from (h of it = h_prev).
        // Fetched one step ahead with plain 16-byte loads -- the former predicated, alignment-checked loads each ended in a
        // full wait: twelve serial L2 / HBM round trips per step (8.6 us per step, profiles/r02e_notes.md).  A lane without a
        // chain (the ragged last tile) is a CLONE of the scene's last node: it loads, computes and stores exactly what that
        // node's lane does (same values to the same addresses), so nothing in the step loop is predicated; only the sums
        // over the 16 chains below leave the clones out.
        const int cnode = imin(node, N - 1);
        const float* gbase = sv.gru + ((((int64_t)net * 2 + dir) * f.B + b) * N + cnode) * (int64_t)(N - 1) * (5 * BH) + 4 * g;
        float* dbase = dgru_base + (int64_t)cnode * (N - 1) * (4 * BH) + 4 * g;
        struct PairIn {
            f32x4 hs[2], r[2], z[2], nn[2], hn[2], hp[2];
            float dd;
            bool has_prev;
        };
        auto load_step = [&](int it, PairIn& o, bool first) {
            const int s = dir ? it : (N - 2 - it);                 // reverse of the forward visiting order
            const int sp = dir ? s + 1 : s - 1;                    // the step the forward came from
            o.has_prev = sp >= 0 && sp <= N - 2;
            const float* row = gbase + (int64_t)s * (5 * BH);
            const float* prow = gbase + (int64_t)(o.has_prev ? sp : s) * (5 * BH);
            for (int T = 0; T < 2; ++T) {
                if (first) o.hs[T] = *reinterpret_cast<const f32x4*>(row + 16 * T);      // later: the previous step's hp
                o.r[T] = *reinterpret_cast<const f32x4*>(row + BH + 16 * T);
                o.z[T] = *reinterpret_cast<const f32x4*>(row + 2 * BH + 16 * T);
                o.nn[T] = *reinterpret_cast<const f32x4*>(row + 3 * BH + 16 * T);
                o.hn[T] = *reinterpret_cast<const f32x4*>(row + 4 * BH + 16 * T);
                o.hp[T] = *reinterpret_cast<const f32x4*>(prow + 16 * T);
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
            for (int T = 0; T < 2; ++T) {
                const f32x4 hs = cur.hs[T];
                const f32x4 hp = zero_unless(cur.has_prev, cur.hp[T]);
                f32x4 dht;
                for (int q = 0; q < 4; ++q) {
                    dht[q] = fmaf(wdiff[T][q], dd, dh[T][q]);
                    hacc[T][q] = fmaf(dd, hs[q], hacc[T][q]);
                }
                o2[T] = gru_gates_bwd(dht, cur.r[T], cur.z[T], cur.nn[T], cur.hn[T], hp);
                da[T] += o2[T].dr; da[2 + T] += o2[T].dz; da[4 + T] += o2[T].dni;
                dgh[T] = o2[T].dr; dgh[2 + T] = o2[T].dz; dgh[4 + T] = o2[T].dnh;
                dhd[T] = o2[T].dh_direct;
            }
            float* drow = dbase + (int64_t)s * (4 * BH);
            for (int T = 0; T < 2; ++T) {
                *reinterpret_cast<f32x4*>(drow + 16 * T) = o2[T].dr;
                *reinterpret_cast<f32x4*>(drow + BH + 16 * T) = o2[T].dz;
                *reinterpret_cast<f32x4*>(drow + 2 * BH + 16 * T) = o2[T].dni;
                *reinterpret_cast<f32x4*>(drow + 3 * BH + 16 * T) = o2[T].dnh;
            }
            IPLAN_SCHED_FENCE();
            // next step's record under this step's 48 MFMAs; its h is this step's h_prev (the BPTT walks the forward's order
            // backwards), already in registers
            for (int T = 0; T < 2; ++T) cur.hs[T] = cur.hp[T];
            load_step(it + 1 < N - 1 ? it + 1 : it, cur, false);
            IPLAN_SCHED_FENCE();
            for (int T = 0; T < 2; ++T) {
                f32x4 acc = dhd[T];
                for (int t = 0; t < 6; ++t) acc = mma_block(whT[T][t], dgh[t], acc);
                dh[T] = acc;
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

    // ---------------------------------------------------------------- E: d(W_b h_j) gather per node
    // node j is seen by ego i != j at pair step s = j - (j > i); the egos are enumerated as i = u + (u >= j), u = 0 .. N-2 (no
    // branch), 18 rows fetched per batch before any of them is added (the former load-add-load chain was one L2 round trip
    // per row: 400 us per scene), summed in the same order as before
    for (int p = w; p < 2 * N; p += 8) {
        const int j = p >> 1, d2 = p & 1;
        const float* base = a.dgru + ((((int64_t)net * 2 + d2) * f.B + b) * N) * (int64_t)(N - 1) * (4 * BH) + l;
        float a0 = 0.f, a1 = 0.f;
        constexpr int UB = 18;
        for (int u0 = 0; u0 < N - 1; u0 += UB) {
            float v0[UB], v1[UB];
            for (int k = 0; k < UB; ++k) {
                const int u = imin(u0 + k, N - 2);
                const int i = u + (u >= j ? 1 : 0);
                const int s = i < j ? j - 1 : j;
                const float* row = base + ((int64_t)i * (N - 1) + s) * (4 * BH);
                v0[k] = row[0];
                v1[k] = row[64];                                   // (lanes >= 32 read the dnh columns: unused)
            }
            for (int k = 0; k < UB; ++k)
                if (u0 + k < N - 1) { a0 += v0[k]; a1 += v1[k]; }
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
    if (!a->g_out || !a->dgru || !a->node_dy || !a->hard_part || !f.h_prev || !f.params)
        return fail(IPLAN_EINVAL, "iplan_gat_bwd: null tensor pointer");
    if (!aligned16(a->g_out) || (a->g_s_net & 3) || (a->g_s_b & 3) || !aligned16(a->dgru) || !aligned16(a->node_dy))
        return fail(IPLAN_EALIGN, "iplan_gat_bwd: g_out / dgru / node_dy must be 16-byte aligned");
    hipLaunchKernelGGL(gat_bwd_kernel, dim3((unsigned)(f.n_nets * f.B)), dim3(512), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_gat_bwd");
}
