// Behavior_policy.learn (soft update = iPLAN), nova/stable_behavior_policy.py:161-279 -- forward and
// BPTT of the whole episode in two persistent launches.
//
// A chain is one (env, entity) row of one agent-net.  Chains never interact (the only coupling is the
// mask normaliser of the loss, which depends on the mask alone), so a wave owns 16 chains for the
// entire episode: for every window j < T-1-L it runs
//      decoder:  y_t, h^d = Linear(Dropout(tanh(GRU64(ReLU(Linear([x_t || latent])), h^d))))   t < L
//      encoder:  h^e = GRU32(ReLU(Linear(x_t)), h^e)  t < L;  latent <- (1-c) latent + c softmax(Linear(h^e))
// with both hidden states and the latent carried from window to window in registers (D layout of
// wave_tile.h), weights resident in LDS (140 KB), ~5 100 MFMAs per window.  The forward streams the
// activations BPTT needs; the backward walks the 2 x 790 GRU steps in reverse and streams the
// row-level pre-activation gradients that wgrad.hip contracts into weight gradients.
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int DHd = 64, DT = 4;          // decoder_rnn_dim
constexpr int EHd = 32, ET = 2;          // encoder_rnn_dim
constexpr int DLD = DHd + 4;             // LDS leading dims (ld % 4 == 0, bank-staggered)
constexpr int ELDB = EHd + 4;
constexpr int SVD = IPLAN_BEH_SAVE_DEC, SVE = IPLAN_BEH_SAVE_ENC, SVL = IPLAN_BEH_SAVE_LAT;
constexpr int DSD = IPLAN_BEH_DSAVE_DEC, DSE = IPLAN_BEH_DSAVE_ENC, DSL = IPLAN_BEH_DSAVE_LAT;
constexpr float BEPS = 1e-10f;
// saved_dec / dsave_dec / saved_enc / dsave_enc column offsets (include/iplan_hip.h)
constexpr int SD_X = 0, SD_LAT = 16, SD_U = 32, SD_R = 96, SD_Z = 160, SD_N = 224, SD_HN = 288, SD_H = 352, SD_A = 416, SD_Y = 480;
constexpr int DD_DY = 0, DD_DU = 16, DD_DR = 80, DD_DZ = 144, DD_DNI = 208, DD_DNH = 272;
constexpr int SE_U = 0, SE_R = 32, SE_Z = 64, SE_N = 96, SE_HN = 128, SE_H = 160;
constexpr int DE_DU = 0, DE_DR = 32, DE_DZ = 64, DE_DNI = 96, DE_DNH = 128;

__device__ __forceinline__ float chain_sum_b(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// window geometry: soft update = sliding windows (stride 1, right-aligned zero padding), hard update = blocks
__device__ __forceinline__ int beh_windows(const IplanBehArgs& a) { return a.hard ? a.T / a.L - 1 : a.T - 1 - a.L; }
__device__ __forceinline__ int beh_x_step(const IplanBehArgs& a, int j, int t) { return a.hard ? j * a.L + t : j - (a.L - 1) + t; }
__device__ __forceinline__ int beh_y_step(const IplanBehArgs& a, int j, int t) { return a.hard ? (j + 1) * a.L + t : j + 1 + t; }
__device__ __forceinline__ int beh_m_step(const IplanBehArgs& a, int j, int t) { return a.hard ? j * a.L + t : j + 1 + t; }

// counter-based Bernoulli(1-p) keep flag (used when no mask tensor is injected): same value in the
// forward and the backward launch for the same (seed, element index)
__device__ __forceinline__ float keep_flag(uint64_t seed, uint64_t idx, float p) {
    // murmur3-style 32-bit finaliser over (element index, seed): two 32-bit multiplies per flag
    uint32_t x = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0x9E3779B9u) ^ (uint32_t)seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    x ^= (uint32_t)(seed >> 32);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f : 0.0f;
}

__device__ __forceinline__ f32x4 keep_tile(const IplanBehArgs& a, int net, int j, int row, int t, int T, bool valid, int rows) {
    const int l = lane_id(), g = l >> 4;
    const int64_t base = ((((int64_t)net * beh_windows(a) + j) * rows + row) * a.L + t) * DHd + 16 * T + 4 * g;
    f32x4 k = splat4(0.f);
    if (!valid) return k;
    if (a.keep) {
        for (int q = 0; q < 4; ++q) k[q] = (float)a.keep[base + q];
    } else if (a.drop_p > 0.f) {
        for (int q = 0; q < 4; ++q) k[q] = keep_flag(a.seed, (uint64_t)(base + q), a.drop_p);
    } else {
        k = splat4(1.0f);
    }
    return k;
}

// sum of the mask over the window's target steps (all envs), times N * d  (mask_over_next_traj.sum())
__device__ __forceinline__ float window_mask_sum(const IplanBehArgs& a, int net, int j) {
    float s = 0.f;
    // hard update: ONE normaliser over all windows (nova/behavior_policy.py:185-187)
    const int span = a.hard ? beh_windows(a) * a.L : a.L;
    const int first = a.hard ? 0 : j + 1;
    const int cnt = a.E * span;
    for (int i = lane_id(); i < cnt; i += 64) {
        const int e = i / span, t = i - e * span;
        s += a.mask[((int64_t)net * a.E + e) * a.T + first + t];
    }
    return wave_sum(s) * (float)(a.N * a.d);
}

// x_t of window j for this lane's chain: history[e, j-(L-1)+t] or zeros (right-aligned window)
__device__ __forceinline__ f32x4 window_x(const IplanBehArgs& a, const float* __restrict__ hrow, int j, int t, bool valid) {
    const int st = beh_x_step(a, j, t);
    return vload(hrow + (int64_t)(st < 0 ? 0 : st) * a.h_s_t, valid && st >= 0, a.d, 0);
}

__global__ __launch_bounds__(256) void beh_fwd_kernel(IplanBehArgs a) {
    IPLAN_DYN_LDS(smem);
    float* s_dwih = smem;                                   // [192][DLD]
    float* s_dwhh = s_dwih + 3 * DHd * DLD;                 // [192][DLD]
    float* s_dlin = s_dwhh + 3 * DHd * DLD;                 // [64][20]   W_lin[:, :d]
    float* s_dlinz = s_dlin + DHd * 20;                     // [64][20]   W_lin[:, d:d+Z]
    float* s_dout = s_dlinz + DHd * 20;                     // [16][DLD]
    float* s_ewih = s_dout + 16 * DLD;                      // [96][ELDB]
    float* s_ewhh = s_ewih + 3 * EHd * ELDB;
    float* s_elin = s_ewhh + 3 * EHd * ELDB;                // [32][20]
    float* s_eout = s_elin + EHd * 20;                      // [16][ELDB]
    float* s_db = s_eout + 16 * ELDB;                       // dec biases: lin 64 | ih 192 | hh 192 | out 16
    float* s_eb = s_db + 64 + 192 + 192 + 16;               // enc biases: lin 32 | ih 96 | hh 96 | out 16

    const int net = (int)blockIdx.y;
    const float* __restrict__ PD = a.dec_params + (int64_t)net * a.dec_s_net;
    const float* __restrict__ PE = a.enc_params + (int64_t)net * a.enc_s_net;
    const int din = a.d + a.Z;
    stage_matrix(s_dwih, DLD, 3 * DHd, PD + a.dec_off[IPLAN_DEC_WIH], 3 * DHd, DHd);
    stage_matrix(s_dwhh, DLD, 3 * DHd, PD + a.dec_off[IPLAN_DEC_WHH], 3 * DHd, DHd);
    {   // split the input Linear by source: columns [0, d) act on x_t, [d, d+Z) on the latent
        const float* Wl = PD + a.dec_off[IPLAN_DEC_LIN_W];
        for (int idx = (int)threadIdx.x; idx < DHd * 20; idx += (int)blockDim.x) {
            const int m = idx / 20, c = idx - m * 20;
            s_dlin[idx] = c < a.d ? Wl[(int64_t)m * din + c] : 0.f;
            s_dlinz[idx] = c < a.Z ? Wl[(int64_t)m * din + a.d + c] : 0.f;
        }
    }
    stage_matrix(s_dout, DLD, 16, PD + a.dec_off[IPLAN_DEC_OUT_W], a.d, DHd);
    stage_matrix(s_ewih, ELDB, 3 * EHd, PE + a.enc_off[IPLAN_ENC_WIH], 3 * EHd, EHd);
    stage_matrix(s_ewhh, ELDB, 3 * EHd, PE + a.enc_off[IPLAN_ENC_WHH], 3 * EHd, EHd);
    stage_matrix(s_elin, 20, EHd, PE + a.enc_off[IPLAN_ENC_LIN_W], EHd, a.d);
    stage_matrix(s_eout, ELDB, 16, PE + a.enc_off[IPLAN_ENC_OUT_W], a.Z, EHd);
    stage_vector(s_db, 64, PD + a.dec_off[IPLAN_DEC_LIN_B], 64);
    stage_vector(s_db + 64, 192, PD + a.dec_off[IPLAN_DEC_BIH], 192);
    stage_vector(s_db + 256, 192, PD + a.dec_off[IPLAN_DEC_BHH], 192);
    stage_vector(s_db + 448, 16, PD + a.dec_off[IPLAN_DEC_OUT_B], a.d);
    stage_vector(s_eb, 32, PE + a.enc_off[IPLAN_ENC_LIN_B], 32);
    stage_vector(s_eb + 32, 96, PE + a.enc_off[IPLAN_ENC_BIH], 96);
    stage_vector(s_eb + 128, 96, PE + a.enc_off[IPLAN_ENC_BHH], 96);
    stage_vector(s_eb + 224, 16, PE + a.enc_off[IPLAN_ENC_OUT_B], a.Z);
    __syncthreads();

    const int l = lane_id(), n = l & 15, g = l >> 4;
    const int rows = a.E * a.N;
    const int tiles = (rows + 15) / 16;
    const int tile = (int)blockIdx.x * 4 + wave_id();
    if (tile >= tiles) return;
    const int row = tile * 16 + n;
    const bool valid = row < rows;
    const int e = valid ? row / a.N : 0, ent = valid ? row % a.N : 0;
    const int J = beh_windows(a);
    const float* __restrict__ hrow = a.hist ? a.hist + (int64_t)net * a.h_s_net + (int64_t)e * a.h_s_e + (int64_t)ent * a.d : nullptr;
    const float* __restrict__ mrow = a.mask ? a.mask + ((int64_t)net * a.E + e) * a.T : nullptr;
    const int64_t grow = (int64_t)net * rows + (valid ? row : 0);
    const float inv_keep = 1.0f / (1.0f - a.drop_p);

    f32x4 hd[DT], he[ET], lat;
    for (int t = 0; t < DT; ++t) hd[t] = splat4(0.f);
    for (int t = 0; t < ET; ++t) he[t] = splat4(0.f);
    lat = splat4(0.f);
    const bool dec_only = a.win != nullptr;                  // Behavior_Latent_Decoder.forward on an explicit window
    if (dec_only) {
        lat = vload(a.lat_in + grow * a.Z, valid, a.Z, 0);
        for (int t = 0; t < DT; ++t) hd[t] = vload(a.hd_in + grow * DHd, valid, DHd, t);
    }
    float beh = 0.f, stab = 0.f;
    for (int j = 0; j < J; ++j) {
        const float scale = dec_only ? 0.f : (float)(a.d * a.N) / (window_mask_sum(a, net, j) + BEPS);
        float err = 0.f;
        f32x4 zproj[DT], lat1[1];
        lat1[0] = lat;
        for (int T = 0; T < DT; ++T) zproj[T] = dense_tile<1>(s_dlinz, 20, 16 * T, lat1, bfrag_lds(s_db, T));
        for (int t = 0; t < a.L; ++t) {
            const f32x4 xt = dec_only ? vload(a.win + (grow * a.L + t) * a.d, valid, a.d, 0) : window_x(a, hrow, j, t, valid);
            // ---- decoder step (behavior_net.py:39-45, 55-69)
            float* sd = a.saved_dec + ((grow * J + j) * a.L + t) * SVD;
            f32x4 x1[1];
            x1[0] = xt;
            vstore(sd + SD_X, valid, 16, 0, xt);
            vstore(sd + SD_LAT, valid, 16, 0, lat);
            f32x4 u[DT];
            for (int T = 0; T < DT; ++T) {
                // Linear([x_t || latent]) = W[:, :d] x_t + (W[:, d:] latent + b)  -- the latent part is per window
                u[T] = relu4(dense_tile<1>(s_dlin, 20, 16 * T, x1, zproj[T]));
                vstore(sd + SD_U, valid, DHd, T, u[T]);
            }
            GruGates kg[DT];
            gru_step_lds<DT, DT>(s_dwih, DLD, s_dwhh, DLD, s_db + 64, s_db + 256, u, hd, kg);
            f32x4 act[DT];
            for (int T = 0; T < DT; ++T) {
                vstore(sd + SD_R, valid, DHd, T, kg[T].r);
                vstore(sd + SD_Z, valid, DHd, T, kg[T].z);
                vstore(sd + SD_N, valid, DHd, T, kg[T].n);
                vstore(sd + SD_HN, valid, DHd, T, kg[T].hn);
                vstore(sd + SD_H, valid, DHd, T, hd[T]);
                const f32x4 km = keep_tile(a, net, j, row, t, T, valid, rows);
                for (int q = 0; q < 4; ++q) act[T][q] = tanh_f(hd[T][q]) * (km[q] * inv_keep);
                vstore(sd + SD_A, valid, DHd, T, act[T]);
            }
            const f32x4 y = dense_tile<DT>(s_dout, DLD, 0, act, bfrag_lds(s_db + 448, 0));
            vstore(sd + SD_Y, valid, 16, 0, y);
            if (dec_only) {
                vstore(a.pred_out + (grow * a.L + t) * a.d, valid, a.d, 0, y);
                continue;
            }
            // masked L1 against the next window, stability vs the current one (:226, 233-240)
            const f32x4 nx = vload(hrow + (int64_t)beh_y_step(a, j, t) * a.h_s_t, valid, a.d, 0);
            const float m = valid ? mrow[beh_m_step(a, j, t)] : 0.f;
            float d2 = 0.f;
            for (int q = 0; q < 4; ++q) {
                if (4 * g + q < a.d) {
                    err += fabsf(nx[q] - y[q]) * m;
                    const float df = xt[q] - y[q];
                    d2 = fmaf(df, df, d2);
                }
            }
            d2 = group_sum(d2);
            if (valid && g == 0) stab += fmaxf(sqrtf(d2) - a.thres, 0.f);
            // ---- encoder step (behavior_net.py:17-22)
            float* se = a.saved_enc + ((grow * J + j) * a.L + t) * SVE;
            f32x4 ue[ET];
            for (int T = 0; T < ET; ++T) {
                ue[T] = relu4(dense_tile<1>(s_elin, 20, 16 * T, x1, bfrag_lds(s_eb, T)));
                vstore(se + SE_U, valid, EHd, T, ue[T]);
            }
            GruGates ke[ET];
            gru_step_lds<ET, ET>(s_ewih, ELDB, s_ewhh, ELDB, s_eb + 32, s_eb + 128, ue, he, ke);
            for (int T = 0; T < ET; ++T) {
                vstore(se + SE_R, valid, EHd, T, ke[T].r);
                vstore(se + SE_Z, valid, EHd, T, ke[T].z);
                vstore(se + SE_N, valid, EHd, T, ke[T].n);
                vstore(se + SE_HN, valid, EHd, T, ke[T].hn);
                vstore(se + SE_H, valid, EHd, T, he[T]);
            }
        }
        if (dec_only) {
            for (int t = 0; t < DT; ++t) vstore(a.hd_out + grow * DHd, valid, DHd, t, hd[t]);
            return;
        }
        beh = fmaf(err, scale, beh);
        // latent head + soft update (:223-230)
        const f32x4 lg = dense_tile<ET>(s_eout, ELDB, 0, he, bfrag_lds(s_eb + 224, 0));
        float mx = -INFINITY;
        for (int q = 0; q < 4; ++q)
            if (4 * g + q < a.Z) mx = fmaxf(mx, lg[q]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        f32x4 ex;
        float ss = 0.f;
        for (int q = 0; q < 4; ++q) { ex[q] = (4 * g + q < a.Z) ? expf(lg[q] - mx) : 0.f; ss += ex[q]; }
        ss = group_sum(ss);
        f32x4 nl;
        for (int q = 0; q < 4; ++q) nl[q] = ex[q] / ss;
        vstore(a.saved_lat + (grow * J + j) * SVL, valid, 16, 0, nl);
        for (int q = 0; q < 4; ++q) lat[q] = a.hard ? nl[q] : (1.0f - a.coef) * lat[q] + nl[q] * a.coef;
    }
    beh = chain_sum_b(group_sum(beh)) / (a.hard ? 1.0f : (float)J);
    stab = chain_sum_b(group_sum(stab)) / (float)a.E / (float)a.L / (float)J;
    if (l == 0) {
        a.loss_part[((int64_t)net * tiles + tile) * 2] = beh;
        a.loss_part[((int64_t)net * tiles + tile) * 2 + 1] = stab;
    }
}

__global__ __launch_bounds__(64) void beh_loss_kernel(IplanBehArgs a) {
    const int net = (int)blockIdx.x;
    const int tiles = (a.E * a.N + 15) / 16;
    float b = 0.f, s = 0.f;
    for (int i = lane_id(); i < tiles; i += 64) {
        b += a.loss_part[((int64_t)net * tiles + i) * 2];
        s += a.loss_part[((int64_t)net * tiles + i) * 2 + 1];
    }
    b = wave_sum(b);
    s = wave_sum(s);
    if (lane_id() == 0) {
        a.loss[net * 2] = b;
        a.loss[net * 2 + 1] = s;
    }
}

__global__ __launch_bounds__(256) void beh_bwd_kernel(IplanBehArgs a) {
    IPLAN_DYN_LDS(smem);
    constexpr int TLD = 3 * DHd + 8;                        // 200: ld % 16 == 8 -> conflict-free ds_read_b128 fragments
    constexpr int TLE = 3 * EHd + 8;                        // 104
    float* s_dwihT = smem;                                  // [64][196]   W_ih^T
    float* s_dwhhT = s_dwihT + DHd * TLD;                   // [64][196]
    float* s_doutT = s_dwhhT + DHd * TLD;                   // [64][20]    W_out^T (cols = d)
    float* s_dlatT = s_doutT + DHd * 20;                    // [16][DLD]   W_lin[:, d:d+Z]^T
    float* s_ewihT = s_dlatT + 16 * DLD;                    // [32][100]
    float* s_ewhhT = s_ewihT + EHd * TLE;                   // [32][100]
    float* s_eoutT = s_ewhhT + EHd * TLE;                   // [32][20]    W_out_enc^T (cols = Z)

    const int net = (int)blockIdx.y;
    const float* __restrict__ PD = a.dec_params + (int64_t)net * a.dec_s_net;
    const float* __restrict__ PE = a.enc_params + (int64_t)net * a.enc_s_net;
    const int din = a.d + a.Z;
    stage_matrix_t(s_dwihT, TLD, DHd, PD + a.dec_off[IPLAN_DEC_WIH], 3 * DHd, DHd);
    stage_matrix_t(s_dwhhT, TLD, DHd, PD + a.dec_off[IPLAN_DEC_WHH], 3 * DHd, DHd);
    stage_matrix_t(s_doutT, 20, DHd, PD + a.dec_off[IPLAN_DEC_OUT_W], a.d, DHd);
    {   // s_dlatT[z][m] = W_lin[m][d + z]
        const float* Wl = PD + a.dec_off[IPLAN_DEC_LIN_W];
        for (int idx = (int)threadIdx.x; idx < 16 * DLD; idx += (int)blockDim.x) {
            const int z = idx / DLD, m = idx - z * DLD;
            s_dlatT[idx] = (z < a.Z && m < DHd) ? Wl[(int64_t)m * din + a.d + z] : 0.f;
        }
    }
    stage_matrix_t(s_ewihT, TLE, EHd, PE + a.enc_off[IPLAN_ENC_WIH], 3 * EHd, EHd);
    stage_matrix_t(s_ewhhT, TLE, EHd, PE + a.enc_off[IPLAN_ENC_WHH], 3 * EHd, EHd);
    stage_matrix_t(s_eoutT, 20, EHd, PE + a.enc_off[IPLAN_ENC_OUT_W], a.Z, EHd);
    __syncthreads();

    const int l = lane_id(), n = l & 15, g = l >> 4;
    const int rows = a.E * a.N;
    const int tiles = (rows + 15) / 16;
    const int tile = (int)blockIdx.x * 4 + wave_id();
    if (tile >= tiles) return;
    const int row = tile * 16 + n;
    const bool valid = row < rows;
    const int e = valid ? row / a.N : 0, ent = valid ? row % a.N : 0;
    const int J = beh_windows(a);
    const float* __restrict__ hrow = a.hist + (int64_t)net * a.h_s_net + (int64_t)e * a.h_s_e + (int64_t)ent * a.d;
    const float* __restrict__ mrow = a.mask + ((int64_t)net * a.E + e) * a.T;
    const int64_t grow = (int64_t)net * rows + (valid ? row : 0);
    const float inv_keep = 1.0f / (1.0f - a.drop_p);

    // Everything one backward step reads from HBM (the forward's record of that step, the loss target, the mask
    // and the dropout flags).  The NEXT step's record is fetched while the current step computes: with one wave
    // per SIMD nothing else hides the ~2 us HBM latency, and the un-prefetched kernel spent half its cycles in
    // s_waitcnt (profiles/).
    struct StepIn {
        f32x4 e_r[ET], e_z[ET], e_n[ET], e_hn[ET], e_hp[ET];
        f32x4 d_r[DT], d_z[DT], d_n[DT], d_hn[DT], d_hp[DT];
        f32x4 d_y, d_nx;
        float m;
    };
    auto load_step = [&](int j, int t, StepIn& o) {
        const int64_t step = (grow * J + j) * a.L + t;
        const bool first = (j == 0 && t == 0);
        const float* se = a.saved_enc + step * SVE;
        const float* sd = a.saved_dec + step * SVD;
        for (int T = 0; T < ET; ++T) {
            o.e_r[T] = vload(se + SE_R, valid, EHd, T);
            o.e_z[T] = vload(se + SE_Z, valid, EHd, T);
            o.e_n[T] = vload(se + SE_N, valid, EHd, T);
            o.e_hn[T] = vload(se + SE_HN, valid, EHd, T);
            o.e_hp[T] = vload(se - SVE + SE_H, valid && !first, EHd, T);
        }
        for (int T = 0; T < DT; ++T) {
            o.d_r[T] = vload(sd + SD_R, valid, DHd, T);
            o.d_z[T] = vload(sd + SD_Z, valid, DHd, T);
            o.d_n[T] = vload(sd + SD_N, valid, DHd, T);
            o.d_hn[T] = vload(sd + SD_HN, valid, DHd, T);
            o.d_hp[T] = vload(sd - SVD + SD_H, valid && !first, DHd, T);
        }
        o.d_y = vload(sd + SD_Y, valid, 16, 0);
        o.d_nx = vload(hrow + (int64_t)beh_y_step(a, j, t) * a.h_s_t, valid, a.d, 0);
        o.m = valid ? mrow[beh_m_step(a, j, t)] : 0.f;
    };

    f32x4 dhd[DT], dhe[ET], dlat;
    for (int t = 0; t < DT; ++t) dhd[t] = splat4(0.f);
    for (int t = 0; t < ET; ++t) dhe[t] = splat4(0.f);
    dlat = splat4(0.f);
    StepIn cur;
    load_step(J - 1, a.L - 1, cur);
    f32x4 hcur[DT];                                     // decoder h of the current step (= h_prev of the step just done)
    for (int T = 0; T < DT; ++T) hcur[T] = vload(a.saved_dec + ((grow * J + (J - 1)) * a.L + (a.L - 1)) * SVD + SD_H, valid, DHd, T);
    for (int j = J - 1; j >= 0; --j) {
        const float scale = (float)(a.d * a.N) / (window_mask_sum(a, net, j) + BEPS) / (a.hard ? 1.0f : (float)J);
        // ---- soft update + latent head backward
        f32x4 dlog[1];
        {
            const f32x4 nl = vload(a.saved_lat + (grow * J + j) * SVL, valid, 16, 0);
            float s = 0.f;
            f32x4 dnew;
            const float cn = a.hard ? 1.0f : a.coef, ck = a.hard ? 0.0f : 1.0f - a.coef;
            for (int q = 0; q < 4; ++q) { dnew[q] = cn * dlat[q]; s = fmaf(nl[q], dnew[q], s); }
            s = group_sum(s);
            for (int q = 0; q < 4; ++q) {
                dlog[0][q] = nl[q] * (dnew[q] - s);
                dlat[q] *= ck;
            }
            vstore(a.dsave_lat + (grow * J + j) * DSL, valid, 16, 0, dlog[0]);
            for (int T = 0; T < ET; ++T) dhe[T] = dense_tile<1>(s_eoutT, 20, 16 * T, dlog, dhe[T]);
        }
        for (int t = a.L - 1; t >= 0; --t) {
            const int64_t step = (grow * J + j) * a.L + t;
            float* de = a.dsave_enc + step * DSE;
            float* dd_ = a.dsave_dec + step * DSD;
            // ReLU inputs of the two input Linears: needed only after the first MFMA products, loaded here so their
            // latency hides behind part A
            f32x4 eu[ET], du_[DT];
            for (int T = 0; T < ET; ++T) eu[T] = vload(a.saved_enc + step * SVE + SE_U, valid, EHd, T);
            for (int T = 0; T < DT; ++T) du_[T] = vload(a.saved_dec + step * SVD + SD_U, valid, DHd, T);
            // ---- part A: everything lane-local that consumes the step's record
            f32x4 eg[4 * ET], edd[ET];                     // encoder [dr | dz | dn_i | dn_h], direct path
            for (int T = 0; T < ET; ++T) {
                const GruGrads o = gru_gates_bwd(dhe[T], cur.e_r[T], cur.e_z[T], cur.e_n[T], cur.e_hn[T], cur.e_hp[T]);
                vstore(de + DE_DR, valid, EHd, T, o.dr);
                vstore(de + DE_DZ, valid, EHd, T, o.dz);
                vstore(de + DE_DNI, valid, EHd, T, o.dni);
                vstore(de + DE_DNH, valid, EHd, T, o.dnh);
                eg[T] = o.dr; eg[ET + T] = o.dz; eg[2 * ET + T] = o.dni; eg[3 * ET + T] = o.dnh;
                edd[T] = o.dh_direct;
            }
            f32x4 dy[1];
            for (int q = 0; q < 4; ++q) {
                float v = 0.f;
                if (valid && 4 * g + q < a.d) {
                    const float er = cur.d_nx[q] - cur.d_y[q];
                    v = -((er > 0.f) ? 1.0f : (er < 0.f ? -1.0f : 0.0f)) * cur.m * scale;
                }
                dy[0][q] = v;
            }
            vstore(dd_ + DD_DY, valid, 16, 0, dy[0]);
            f32x4 dg[4 * DT], ddir[DT];                    // decoder [dr | dz | dn_i | dn_h]
            for (int T = 0; T < DT; ++T) {
                const f32x4 da = dense_tile<1>(s_doutT, 20, 16 * T, dy, splat4(0.f));
                const f32x4 km = keep_tile(a, net, j, row, t, T, valid, rows);
                f32x4 dht;
                for (int q = 0; q < 4; ++q) {
                    const float th = tanh_f(hcur[T][q]);
                    dht[q] = fmaf(da[q] * km[q] * inv_keep, 1.0f - th * th, dhd[T][q]);
                }
                const GruGrads o = gru_gates_bwd(dht, cur.d_r[T], cur.d_z[T], cur.d_n[T], cur.d_hn[T], cur.d_hp[T]);
                vstore(dd_ + DD_DR, valid, DHd, T, o.dr);
                vstore(dd_ + DD_DZ, valid, DHd, T, o.dz);
                vstore(dd_ + DD_DNI, valid, DHd, T, o.dni);
                vstore(dd_ + DD_DNH, valid, DHd, T, o.dnh);
                dg[T] = o.dr; dg[DT + T] = o.dz; dg[2 * DT + T] = o.dni; dg[3 * DT + T] = o.dnh;
                ddir[T] = o.dh_direct;
                hcur[T] = cur.d_hp[T];                     // h_{t-1}: the next step's "current" hidden state
            }
            // ---- the record is consumed: fetch the next step's while the transposed-weight products below run
            IPLAN_SCHED_FENCE();
            if (t > 0) load_step(j, t - 1, cur);
            else if (j > 0) load_step(j - 1, a.L - 1, cur);
            IPLAN_SCHED_FENCE();
            // ---- part B: backward-data products on MFMA (all output tiles of a product share the B operand)
            {
                const int oe[ET] = {0, 16};
                f32x4 du[ET];
                for (int T = 0; T < ET; ++T) du[T] = splat4(0.f);
                dense_multi<ET, 3 * ET>(s_ewihT, TLE, oe, 0, eg, du);                       // W_ih^T [dr dz dn_i]
                for (int T = 0; T < ET; ++T) {
                    f32x4 dup;
                    for (int q = 0; q < 4; ++q) dup[q] = eu[T][q] > 0.f ? du[T][q] : 0.f;
                    vstore(de + DE_DU, valid, EHd, T, dup);
                    dhe[T] = edd[T];
                }
                dense_multi<ET, 2 * ET>(s_ewhhT, TLE, oe, 0, eg, dhe);                      // W_hh^T [dr dz | dn_h]
                dense_multi<ET, ET>(s_ewhhT, TLE, oe, 2 * EHd, eg + 3 * ET, dhe);
            }
            f32x4 dup[DT];
            {
                const int od[DT] = {0, 16, 32, 48};
                f32x4 du[DT];
                for (int T = 0; T < DT; ++T) du[T] = splat4(0.f);
                dense_multi<DT, 3 * DT>(s_dwihT, TLD, od, 0, dg, du);
                for (int T = 0; T < DT; ++T) {
                    for (int q = 0; q < 4; ++q) dup[T][q] = du_[T][q] > 0.f ? du[T][q] : 0.f;
                    vstore(dd_ + DD_DU, valid, DHd, T, dup[T]);
                    dhd[T] = ddir[T];
                }
                IPLAN_SCHED_FENCE();
                dense_multi<DT, 2 * DT>(s_dwhhT, TLD, od, 0, dg, dhd);
                dense_multi<DT, DT>(s_dwhhT, TLD, od, 2 * DHd, dg + 3 * DT, dhd);
                IPLAN_SCHED_FENCE();
            }
            dlat = dense_tile<DT>(s_dlatT, DLD, 0, dup, dlat);       // through the tiled latent input
        }
    }
}

static int check_beh(const IplanBehArgs* a, const char* what) {
    if (!a) return fail(IPLAN_EINVAL, "%s: null args", what);
    if (a->n_nets < 1 || a->E < 1 || a->N < 1 || a->L < 1 || (a->hard ? a->T / a->L - 1 : a->T - 1 - a->L) < 1 || a->d < 1 || a->Z < 1 ||
        a->d + a->Z > 16 || a->Z > 16)
        return fail(IPLAN_EINVAL, "%s: unsupported dims E=%d N=%d T=%d L=%d d=%d Z=%d", what, a->E, a->N, a->T, a->L, a->d, a->Z);
    if (a->win) {
        if (!a->lat_in || !a->hd_in || !a->pred_out || !a->hd_out || a->T != a->L + 2 || !a->saved_dec)
            return fail(IPLAN_EINVAL, "%s: single-window decoder mode needs lat_in, hd_in, pred_out, hd_out, saved_dec and T == L + 2", what);
    } else if (!a->hist || !a->mask || !a->saved_enc || !a->saved_lat) {
        return fail(IPLAN_EINVAL, "%s: null tensor pointer", what);
    }
    if (!a->enc_params || !a->dec_params || !a->saved_dec)
        return fail(IPLAN_EINVAL, "%s: null tensor pointer", what);
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return fail(IPLAN_EINVAL, "%s: dropout p=%f", what, a->drop_p);
    return IPLAN_OK;
}

}  // namespace iplan

extern "C" int iplan_beh_fwd(const IplanBehArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_beh(a, "iplan_beh_fwd")) return rc;
    if (!a->win && (!a->loss_part || !a->loss)) return fail(IPLAN_EINVAL, "iplan_beh_fwd: loss buffers missing");
    const int tiles = (a->E * a->N + 15) / 16;
    const size_t lds = sizeof(float) * (2 * 3 * DHd * DLD + 2 * DHd * 20 + 16 * DLD + 2 * 3 * EHd * ELDB + EHd * 20 + 16 * ELDB +
                                        (64 + 192 + 192 + 16) + (32 + 96 + 96 + 16));
#ifndef IPLAN_HOST_EMULATION
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(beh_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
    hipLaunchKernelGGL(beh_fwd_kernel, dim3((unsigned)((tiles + 3) / 4), (unsigned)a->n_nets), dim3(256), lds,
                       (hipStream_t)stream, *a);
    if (!a->win) hipLaunchKernelGGL(beh_loss_kernel, dim3((unsigned)a->n_nets), dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_beh_fwd");
}

extern "C" int iplan_beh_bwd(const IplanBehArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_beh(a, "iplan_beh_bwd")) return rc;
    if (!a->dsave_dec || !a->dsave_enc || !a->dsave_lat) return fail(IPLAN_EINVAL, "iplan_beh_bwd: dsave buffers missing");
    const int tiles = (a->E * a->N + 15) / 16;
    const size_t lds = sizeof(float) * (2 * DHd * (3 * DHd + 8) + DHd * 20 + 16 * DLD + 2 * EHd * (3 * EHd + 8) + EHd * 20);
#ifndef IPLAN_HOST_EMULATION
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(beh_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
    hipLaunchKernelGGL(beh_bwd_kernel, dim3((unsigned)((tiles + 3) / 4), (unsigned)a->n_nets), dim3(256), lds,
                       (hipStream_t)stream, *a);
    return check_launch("iplan_beh_bwd");
}
