// Behavior_policy.learn (soft update = iPLAN), nova/stable_behavior_policy.py:161-279 -- forward and
// BPTT of the whole episode in two persistent launches.
//
// A chain is one (env, entity) row of one agent-net.  Chains never interact (the only coupling is the
// mask normaliser of the loss, which depends on the mask alone), so a wave owns 16 chains for the
// entire episode; for every window j it runs
//      encoder:  h^e = GRU32(ReLU(Linear(x_t)), h^e)  t < L;  latent <- (1-c) latent + c softmax(Linear(h^e))
//      decoder:  y_t, h^d = Linear(Dropout(tanh(GRU64(ReLU(Linear([x_t || latent])), h^d))))   t < L
// The data flow between the two is one-way and tiny: the decoder only consumes the per-window latent, the
// encoder's backward only consumes the decoder's per-window d(loss)/d(latent).  So the two chains are four
// kernels, each with only its own weights in LDS:
//   beh_enc_fwd  (all windows; light, several waves per SIMD)  -> latents per window, encoder records
//   beh_dec_fwd  (416 MFMAs per step, 126 KB of LDS weights) -> loss, decoder records
//   beh_dec_bwd  BPTT of the decoder; streams row gradients for wgrad.hip and d(loss)/d(latent_j)
//                In both decoder kernels a 16-chain tile is shared by TWO waves of the same SIMD (waves w, w + 4
//                of a 512-thread block): each computes the gates of two of the four hidden tiles and the halves
//                meet through LDS once per step (forward: the new h halves; backward: partial W^T products).
//   beh_enc_bwd  BPTT of the encoder with its weight gradients accumulated IN the kernel (H = 32: the
//                24 accumulator tiles fit in registers; operands are turned through LDS each step)
// hidden states and the latent are carried from window to window in registers (D layout of wave_tile.h).
#define IPLAN_SPLIT_PAIRS           // wave_tile.h: the bf16 split two values at a time (this file's kernels gain by it: learn 15.4 -> 15.0 ms)
#include <cstdlib>

#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int DHd = 64, DT = 4;          // decoder_rnn_dim
constexpr int EHd = 32, ET = 2;          // encoder_rnn_dim
constexpr int DLD = DHd + 8;             // LDS leading dims: ld % 16 == 8 -> conflict-free ds_read_b128 fragment reads
constexpr int ELDB = EHd + 8;
constexpr int SVD = IPLAN_BEH_SAVE_DEC, SVE = IPLAN_BEH_SAVE_ENC, SVL = IPLAN_BEH_SAVE_LAT;
constexpr int DSD = IPLAN_BEH_DSAVE_DEC, DSL = IPLAN_BEH_DSAVE_LAT;
constexpr float BEPS = 1e-10f;
constexpr int DEC_FWD_BIAS = 64 + 192 + 192 + 16;
// saved_dec / dsave_dec / saved_enc column offsets (include/iplan_hip.h)
constexpr int SD_X = 0, SD_U = 32, SD_R = 96, SD_Z = 160, SD_N = 224, SD_HN = 288, SD_H = 352, SD_A = 416, SD_Y = 480;
constexpr int DD_DY = 0, DD_DU = 16, DD_DR = 80, DD_DZ = 144, DD_DNI = 208, DD_DNH = 272;
constexpr int SE_U = 0, SE_R = 32, SE_Z = 64, SE_N = 96, SE_HN = 128, SE_H = 160;

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

// counter-based Bernoulli(1-p) keep flags (used when no mask tensor is injected): same values in the forward
// and the backward launch for the same (seed, element index).  One 32-bit murmur3-style finaliser yields two
// 16-bit uniforms, so a lane's 4 flags of a tile cost two hashes (integer multiplies are quarter rate and do
// not overlap with the MFMA pipe).
__device__ __forceinline__ uint32_t keep_hash(uint64_t seed, uint64_t idx) {
    uint32_t x = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0x9E3779B9u) ^ (uint32_t)seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    x ^= (uint32_t)(seed >> 32);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}
__device__ __forceinline__ f32x4 keep_flags4(uint64_t seed, uint64_t base, float p) {
    const uint32_t thr = (uint32_t)(p * 65536.0f);          // keep iff u16 >= p * 2^16
    const uint32_t h0 = keep_hash(seed, base), h1 = keep_hash(seed, base + 2);
    f32x4 k;
    k[0] = (h0 & 0xFFFFu) >= thr ? 1.0f : 0.0f;
    k[1] = (h0 >> 16) >= thr ? 1.0f : 0.0f;
    k[2] = (h1 & 0xFFFFu) >= thr ? 1.0f : 0.0f;
    k[3] = (h1 >> 16) >= thr ? 1.0f : 0.0f;
    return k;
}

__device__ __forceinline__ f32x4 keep_tile(const IplanBehArgs& a, int net, int j, int row, int t, int T, bool valid, int rows) {
    const int l = lane_id(), g = l >> 4;
    const int64_t base = ((((int64_t)net * beh_windows(a) + j) * rows + row) * a.L + t) * DHd + 16 * T + 4 * g;
    f32x4 k = splat4(0.f);
    if (!valid) return k;
    if (a.keep) {
        for (int q = 0; q < 4; ++q) k[q] = (float)a.keep[base + q];
    } else if (a.drop_p > 0.f) {
        k = keep_flags4(a.seed, (uint64_t)base, a.drop_p);
    } else {
        k = splat4(1.0f);
    }
    return k;
}

// sum of the mask over the window's target steps (all envs), times N * d  (mask_over_next_traj.sum())
__device__ __forceinline__ float window_mask_sum(const IplanBehArgs& a, int net, int j) {
    if (a.win_norm) return a.win_norm[(int64_t)net * beh_windows(a) + j] * (float)(a.N * a.d);   // all data-parallel ranks' envs
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

// entries 4g .. 4g+3 of a dim-wide row behind a per-lane pointer (any alignment), indices clamped to the row, NOT masked
__device__ __forceinline__ f32x4 ld_row_raw(const float* __restrict__ row, int dim, int g) {
    f32x4 v;
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * g + q;
        v[q] = row[i < dim ? i : dim - 1];
    }
    return v;
}
// ------------------------------------------------------------------------------------------------------------
// shared prologue of the four kernels
struct BehChain {
    int net, rows, tiles, tile, row, e, ent, J, n, g;
    bool valid;
    const float* hrow;
    const float* mrow;
    int64_t grow;
};
__device__ __forceinline__ bool beh_chain(const IplanBehArgs& a, BehChain& c) {
    const int l = lane_id();
    c.n = l & 15;
    c.g = l >> 4;
    c.net = (int)blockIdx.y;
    c.rows = a.E * a.N;
    c.tiles = (c.rows + 15) / 16;
    c.tile = (int)blockIdx.x * 4 + (wave_id() & 3);      // the decoder kernels run two waves per tile (waves w, w + 4)
    c.row = c.tile * 16 + c.n;
    c.valid = c.tile < c.tiles && c.row < c.rows;
    c.e = c.valid ? c.row / a.N : 0;
    c.ent = c.valid ? c.row % a.N : 0;
    c.J = beh_windows(a);
    c.hrow = a.hist ? a.hist + (int64_t)c.net * a.h_s_net + (int64_t)c.e * a.h_s_e + (int64_t)c.ent * a.d : nullptr;
    c.mrow = a.mask ? a.mask + ((int64_t)c.net * a.E + c.e) * a.T : nullptr;
    c.grow = (int64_t)c.net * c.rows + (c.valid ? c.row : 0);
    return c.tile < c.tiles;
}

// ------------------------------------------------------------------------------------------------------------
// encoder forward over the whole episode: records per step (saved_enc) and per window (saved_lat)
// BF3 (default since round 4; IPLAN_ENC_FP32=1 selects the fp32-MFMA form): the GRU32's two contractions W_ih u, W_hh h in the
// fp32-exact split-bf16 form (wave_tile.h) with ALL weight pieces register-resident -- 2 x 6 row tiles x one K = 32 chunk = 144
// registers; u and h are split once per step (the 8 D-layout values a lane holds of a 32-vector are the 8 K slots of its lane
// group).  72 bf16 MFMAs (~17 cycles each) replace 96 fp32 MFMAs (32 cycles each, and they take the VALU's issue slots): the
// shape that took the GAT recurrence from 156 to 115 us (gat.hip).  The 5-wide input Linear stays on fp32 MFMA (8 per step).
template <bool BF3>
__global__ __launch_bounds__(256) void beh_enc_fwd_kernel(IplanBehArgs a) {
    __shared__ __attribute__((aligned(16))) float s_wih[BF3 ? 16 : 3 * EHd * ELDB];
    __shared__ __attribute__((aligned(16))) float s_whh[BF3 ? 16 : 3 * EHd * ELDB];
    __shared__ __attribute__((aligned(16))) float s_lin[EHd * 24];
    __shared__ __attribute__((aligned(16))) float s_out[16 * ELDB];
    __shared__ __attribute__((aligned(16))) float s_b[32 + 96 + 96 + 16];
    const float* __restrict__ PE = a.enc_params + (int64_t)blockIdx.y * a.enc_s_net;
    if (!BF3) {
        stage_matrix(s_wih, ELDB, 3 * EHd, PE + a.enc_off[IPLAN_ENC_WIH], 3 * EHd, EHd);
        stage_matrix(s_whh, ELDB, 3 * EHd, PE + a.enc_off[IPLAN_ENC_WHH], 3 * EHd, EHd);
    }
    stage_matrix(s_lin, 24, EHd, PE + a.enc_off[IPLAN_ENC_LIN_W], EHd, a.d);
    stage_matrix(s_out, ELDB, 16, PE + a.enc_off[IPLAN_ENC_OUT_W], a.Z, EHd);
    stage_vector(s_b, 32, PE + a.enc_off[IPLAN_ENC_LIN_B], 32);
    stage_vector(s_b + 32, 96, PE + a.enc_off[IPLAN_ENC_BIH], 96);
    stage_vector(s_b + 128, 96, PE + a.enc_off[IPLAN_ENC_BHH], 96);
    stage_vector(s_b + 224, 16, PE + a.enc_off[IPLAN_ENC_OUT_B], a.Z);
    __syncthreads();
    BehChain c;
    if (!beh_chain(a, c)) return;
    const bool valid = c.valid;
    const int g = c.g, J = c.J, l = lane_id();
    const int j_lo = imax(a.fwd_j_lo, 0), j_hi = a.fwd_j_hi > 0 ? imin(a.fwd_j_hi, J) : J;
    float* carry = a.enc_carry ? a.enc_carry + ((int64_t)c.net * c.tiles + c.tile) * 768 : nullptr;
    f32x4 he[ET], lat = splat4(0.f);
    for (int t = 0; t < ET; ++t) he[t] = splat4(0.f);
    if (j_lo > 0 && carry) {                                // hidden state and latent handed over by the previous piece
        for (int t = 0; t < ET; ++t) he[t] = *reinterpret_cast<const f32x4*>(carry + 256 * t + 4 * l);
        lat = *reinterpret_cast<const f32x4*>(carry + 512 + 4 * l);
    }
    // x_t is fetched one step ahead (raw, from a clamped row) and masked where it is consumed: the load has a whole GRU step
    // to land instead of being waited for at the top of every one of the 790 steps
    auto x_fetch = [&](int j, int t, bool& has) {
        const int st = beh_x_step(a, j, t);
        has = st >= 0;
        return ld_row_raw(c.hrow + (int64_t)(st < 0 ? 0 : st) * a.h_s_t, a.d, g);
    };
    bool x_has = false;
    f32x4 x_raw = x_fetch(j_lo, 0, x_has);
    const int64_t ecgs = (int64_t)J * a.L * 256;                                  // floats between the record's 16-column groups
    float* enc_rec = a.saved_enc + ((int64_t)c.net * c.tiles + (c.tile < c.tiles ? c.tile : 0)) * (SVE / 16) * ecgs + 16 * c.n;
    // split-bf16 form: weight pieces of row tile c (c = 0, 1: r; 2, 3: z; 4, 5: n -- PyTorch's gate order) and the gate biases
    Bf3 wih[BF3 ? 6 : 1], whh[BF3 ? 6 : 1];
    f32x4 b_rz[4], b_in[2], b_hn[2];
    if (BF3) {
        for (int t = 0; t < 6; ++t) {
            wih[t] = wfrag_bf3(PE + a.enc_off[IPLAN_ENC_WIH], EHd, 3 * EHd, 16 * t, 0);
            whh[t] = wfrag_bf3(PE + a.enc_off[IPLAN_ENC_WHH], EHd, 3 * EHd, 16 * t, 0);
        }
        for (int t = 0; t < 4; ++t) b_rz[t] = bfrag_lds(s_b + 32, t) + bfrag_lds(s_b + 128, t);
        for (int t = 0; t < 2; ++t) { b_in[t] = bfrag_lds(s_b + 32, 4 + t); b_hn[t] = bfrag_lds(s_b + 128, 4 + t); }
    }
    for (int j = j_lo; j < j_hi; ++j) {
        float* sl = a.saved_lat + (c.grow * J + j) * SVL;
        vstore_a(sl + 16, valid, 0, lat);                      // the latent the decoder uses in window j
        for (int t = 0; t < a.L; ++t) {
            f32x4 x1[1];
            for (int q = 0; q < 4; ++q) x1[0][q] = keep_if(valid && x_has && 4 * g + q < a.d, x_raw[q]);
            {
                const bool last_t = t + 1 == a.L, more = j + 1 < j_hi;
                x_raw = x_fetch(last_t && more ? j + 1 : j, last_t ? (more ? 0 : t) : t + 1, x_has);
            }
            // the encoder record is column-grouped like the decoder's ([tile][16-column group][step][chain][16], include/iplan_hip.h):
            // a store of one column group of the wave's 16 chains is one contiguous 1 KiB block
            float* se = enc_rec + ((int64_t)j * a.L + t) * 256;
            f32x4 ue[ET];
            for (int T = 0; T < ET; ++T) {
                ue[T] = relu4(dense_tile<1>(s_lin, 24, 16 * T, x1, bfrag_lds(s_b, T)));
                vstore_a(se + ((SE_U >> 4) + T) * ecgs, valid, 0, ue[T]);
            }
            GruGates ke[ET];
            if (BF3) {
                const Bf3 us = split_bf3(ue[0], ue[1]), hs = split_bf3(he[0], he[1]);
                f32x4 acc[6] = {b_rz[0], b_rz[1], b_rz[2], b_rz[3], b_in[0], b_in[1]}, ahn[2] = {b_hn[0], b_hn[1]};
                // smallest piece products first (fp32 accumulators), the chains issued round-robin; the input half first: it
                // does not wait for the previous step's gates
#define ENC_BF3_ROUND(WP, XP)                                                          \
    for (int cc = 0; cc < 6; ++cc) acc[cc] = mfma_bf16(wih[cc].WP, us.XP, acc[cc]);
                ENC_BF3_ROUND(p2, p0) ENC_BF3_ROUND(p0, p2) ENC_BF3_ROUND(p1, p1) ENC_BF3_ROUND(p1, p0) ENC_BF3_ROUND(p0, p1) ENC_BF3_ROUND(p0, p0)
#undef ENC_BF3_ROUND
#define ENC_BF3_ROUND(WP, XP)                                                          \
    for (int cc = 0; cc < 4; ++cc) acc[cc] = mfma_bf16(whh[cc].WP, hs.XP, acc[cc]);    \
    for (int cc = 0; cc < 2; ++cc) ahn[cc] = mfma_bf16(whh[4 + cc].WP, hs.XP, ahn[cc]);
                ENC_BF3_ROUND(p2, p0) ENC_BF3_ROUND(p0, p2) ENC_BF3_ROUND(p1, p1) ENC_BF3_ROUND(p1, p0) ENC_BF3_ROUND(p0, p1) ENC_BF3_ROUND(p0, p0)
#undef ENC_BF3_ROUND
                for (int T = 0; T < ET; ++T) {
                    ke[T] = gru_gates(acc[T], acc[2 + T], acc[4 + T], ahn[T], he[T]);
                    he[T] = ke[T].h;
                }
            } else {
                gru_step_lds<ET, ET>(s_wih, ELDB, s_whh, ELDB, s_b + 32, s_b + 128, ue, he, ke);
            }
            for (int T = 0; T < ET; ++T) {
                vstore_a(se + ((SE_R >> 4) + T) * ecgs, valid, 0, ke[T].r);
                vstore_a(se + ((SE_Z >> 4) + T) * ecgs, valid, 0, ke[T].z);
                vstore_a(se + ((SE_N >> 4) + T) * ecgs, valid, 0, ke[T].n);
                vstore_a(se + ((SE_HN >> 4) + T) * ecgs, valid, 0, ke[T].hn);
                vstore_a(se + ((SE_H >> 4) + T) * ecgs, valid, 0, he[T]);
            }
        }
        // latent head + soft / hard update (stable_behavior_policy.py:223-230, behavior_policy.py:174-176)
        const f32x4 lg = dense_tile<ET>(s_out, ELDB, 0, he, bfrag_lds(s_b + 224, 0));
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
        vstore_a(sl, valid, 0, nl);
        for (int q = 0; q < 4; ++q) lat[q] = a.hard ? nl[q] : (1.0f - a.coef) * lat[q] + nl[q] * a.coef;
    }
    if (j_hi < J && carry) {
        for (int t = 0; t < ET; ++t) *reinterpret_cast<f32x4*>(carry + 256 * t + 4 * l) = he[t];
        *reinterpret_cast<f32x4*>(carry + 512 + 4 * l) = lat;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Decoder kernels: geometry and addressing shared by the forward and the BPTT.
//
// QUARTER SPLIT.  A 16-chain tile's GRU64 step is 416 MFMAs; with the weights in LDS (one copy per workgroup) the number of
// tiles a CU can host decides how evenly the 550 tiles of config 3 spread over the chip.  A tile is shared by FOUR waves,
// one per SIMD: wave q owns hidden tile q (its r, z, n gate rows: 3 of the 12 row tiles of W_ih / W_hh), needs the whole u
// and h vectors as B operands and hands its 16 new hidden units to the other three through LDS once per step.  A
// 768-thread workgroup = 3 tiles x 4 quarters, so a SIMD runs three quarter-waves = 0.75 tile-steps per step (the former
// half split, 4 tiles x 2 halves on 138 CUs, put a whole tile-step on every SIMD), and 5 x 37 = 185 workgroups leave 71
// CUs to the encoder / weight-gradient kernels that run beside.
//
// ADDRESSING.  Everything a lane touches is a wave-uniform 64-bit base (SGPRs) plus a 32-bit lane byte offset: the records
// of a tile are [16 chains][J*L steps][cols], so the per-step advance is one scalar add and the column is an immediate.
// FULL = all 16 chains of the tile exist (every tile when rows % 16 == 0): no predication on any load / store.
constexpr int DEC_TILES = 3;
constexpr int DEC_THREADS = 256 * DEC_TILES;


// per-wave view of one chain tile of the decoder kernels
struct DecTile {
    int net, rows, tiles, tile, J;
    bool live, full, valid;
    int n, g, row, e, ent;
    uint32_t rec_lane;            // byte offset of this lane's chain inside the tile's block of a per-step record, per float column: x steps*cols*4
    const char* hist;             // uniform: a.hist + net * h_s_net
    uint32_t hist_lane;           // (e * h_s_e + ent * d) * 4
    const char* mask;             // uniform: a.mask + net * E * T
    uint32_t mask_lane;           // e * T * 4
    int64_t grow0;                // net * rows + tile * 16  (first chain of the tile)
    int64_t trow0;                // (net * tiles + tile) * 16: first chain slot of the tile in the column-grouped decoder records
};
__device__ __forceinline__ void dec_tile(const IplanBehArgs& a, DecTile& c, int tile) {
    const int l = lane_id();
    c.n = l & 15;
    c.g = l >> 4;
    c.net = (int)blockIdx.y;
    c.rows = a.E * a.N;
    c.tiles = (c.rows + 15) / 16;
    c.tile = tile;
    c.live = tile < c.tiles;
    c.full = c.live && tile * 16 + 15 < c.rows;
    c.row = tile * 16 + c.n;
    c.valid = c.live && c.row < c.rows;
    const int rr = c.valid ? c.row : 0;
    c.e = rr / a.N;
    c.ent = rr - c.e * a.N;
    c.J = beh_windows(a);
    c.hist = reinterpret_cast<const char*>(a.hist ? a.hist + (int64_t)c.net * a.h_s_net : nullptr);
    c.hist_lane = (uint32_t)(((int64_t)c.e * a.h_s_e + (int64_t)c.ent * a.d) * 4);
    c.mask = reinterpret_cast<const char*>(a.mask ? a.mask + (int64_t)c.net * a.E * a.T : nullptr);
    c.mask_lane = (uint32_t)(c.e * a.T * 4);
    c.grow0 = (int64_t)c.net * c.rows + (int64_t)(c.live ? tile : 0) * 16;
    c.trow0 = ((int64_t)c.net * c.tiles + (int64_t)(c.live ? tile : 0)) * 16;
}

// Decoder records are COLUMN-GROUPED (include/iplan_hip.h, IPLAN_BEH_SAVE_DEC): [net][chain tile][16-column group][step][chain][16].
// A lane's 16-byte slice of column group cg of step s:  tile base (uniform)  +  cg * cgs  +  s * 1024  +  n * 64 + g * 16  bytes,
// cgs = steps per chain * 1024: every store / load instruction of a wave is ONE contiguous 1 KiB block (the chain-major layout
// of rounds 1-2 made it 16 segments of 64 bytes 1.5 MB apart: 3.3 instead of 5.6 TB/s of stores, scripts/ubench/record_store.hip).
#define REC_CG(col) ((uint32_t)((col) >> 4))

// Loads are BRANCH-FREE: a lane that has nothing to fetch reads a clamped (in-bounds) address and the result is replaced by
// zeros with a select.  A load under `if (lane condition)` becomes an exec-masked block, and the compiler's wait-count
// placement around those blocks put `s_waitcnt vmcnt(0)` right behind the record prefetch of the BPTT step -- every step
// then sat out a full HBM round trip plus the drain of its own row-gradient stores (profiles/r02d_notes.md).
template <bool FULL>
__device__ __forceinline__ f32x4 ld4(const char* __restrict__ base, uint32_t off, bool valid) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + ((FULL || valid) ? off : 0u));
    return FULL ? v : zero_unless(valid, v);
}
// fetch only (clamped address, NO masking): for prefetches whose values are masked where they are consumed -- an AND
// right behind the load would pull the wait for it up to the load
template <bool FULL>
__device__ __forceinline__ f32x4 ld4_raw(const char* __restrict__ base, uint32_t off, bool valid) {
    return *reinterpret_cast<const f32x4*>(base + ((FULL || valid) ? off : 0u));
}
template <bool FULL>
__device__ __forceinline__ f32x4 ld_row_raw(const char* __restrict__ base, uint32_t off, bool valid, int dim, int g) {
    const uint32_t o = (FULL || valid) ? off : 0u;
    f32x4 v;
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * g + q;
        v[q] = *reinterpret_cast<const float*>(base + (o + 4u * (uint32_t)(i < dim ? i : dim - 1)));
    }
    return v;
}
template <bool FULL>
__device__ __forceinline__ void st4(char* __restrict__ base, uint32_t off, bool valid, f32x4 v) {
    if (FULL || valid) *reinterpret_cast<f32x4*>(base + off) = v;
}
// the lane's (up to 4) entries 4g .. 4g+3 of a dim-wide row that is NOT 16-byte aligned (history rows: d = 5 floats).
// Predicated variant (loads under lane conditions).  The decoder FORWARD keeps it: its history loads have a whole MFMA phase
// between issue and use either way, and the clamped-index / mask arithmetic of the branch-free form cost it 5 % (same-box A/B,
// profiles/r02d_notes.md).
template <bool FULL>
__device__ __forceinline__ f32x4 ld_row_br(const char* __restrict__ base, uint32_t off, bool valid, int dim, int g) {
    f32x4 v = splat4(0.f);
    if (FULL || valid) {
        const float* p = reinterpret_cast<const float*>(base + off) + 4 * g;
        for (int q = 0; q < 4; ++q)
            if (4 * g + q < dim) v[q] = p[q];
    }
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// decoder forward over the whole episode (+ masked-L1 loss and the stability statistic); also serves
// Behavior_Latent_Decoder.forward on one explicit window (a.win != NULL)
constexpr int XF_H = 0, XF_U = 4, XF_Y = 8, XF_SLOTS = 11;      // exchange slots of a tile: new h | next u | partial y (quarters 1..3)

// (FULL, Q) are template parameters: with the quarter a compile-time constant every LDS fragment address is an immediate and
// the per-tile register arrays are indexed statically; a wave runs exactly one of the instantiations.
template <bool FULL, int Q>
__device__ __forceinline__ void dec_fwd_body(const IplanBehArgs& a, const DecTile& c, const float* __restrict__ s_wih,
                                             const float* __restrict__ s_whh, const float* __restrict__ s_lin,
                                             const float* __restrict__ s_out, const float* __restrict__ s_b, float* __restrict__ xch,
                                             int* __restrict__ tcnt) {
    constexpr int q = Q;
    // Per-tile rendezvous through an LDS counter (IPLAN_TILE_SYNC) or the workgroup barrier: same-box A/B, kernels alone --
    // BPTT 8.19 -> 6.96 ms with the rendezvous (its four quarters are symmetric and three tiles no longer wait for each
    // other), forward 5.89 -> 6.13 ms (quarter 0 carries the output / loss work of a step; the other three then spin on the
    // counter and take issue slots from the quarter-0 waves of the other tiles on their SIMD): barrier here, rendezvous there.
#ifndef DEC_FWD_TILE_SYNC
#define DEC_FWD_TILE_SYNC 0
#endif
#ifndef DEC_BWD_TILE_SYNC
#define DEC_BWD_TILE_SYNC 1
#endif
    int sync_n = 0;
    auto tile_sync = [&]() {
        if (DEC_FWD_TILE_SYNC) { sync_n += 4; IPLAN_TILE_SYNC(tcnt, sync_n); }
        else IPLAN_LDS_BARRIER();
    };
    const bool valid = c.valid;
    const int l = lane_id(), g = c.g, J = c.J, net = c.net;
    const float inv_keep = 1.0f / (1.0f - a.drop_p);
    const bool dec_only = a.win != nullptr;
    const int Lw = a.L;
    auto put = [&](int slot, f32x4 v) { *reinterpret_cast<f32x4*>(xch + slot * 256 + 4 * l) = v; };
    auto get = [&](int slot) { return *reinterpret_cast<const f32x4*>(xch + slot * 256 + 4 * l); };
    const int j_lo = dec_only ? 0 : imax(a.fwd_j_lo, 0), j_hi = (!dec_only && a.fwd_j_hi > 0) ? imin(a.fwd_j_hi, J) : J;
    // record bases of this tile (uniform) and lane offsets; a step's record starts at step * cols * 4 bytes
    const int64_t steps_per_chain = (int64_t)J * Lw;
    char* sd_base = reinterpret_cast<char*>(a.saved_dec + c.trow0 * steps_per_chain * SVD);
    const uint32_t sd_lane = 64u * (uint32_t)c.n + 16u * (uint32_t)g, cgs = (uint32_t)(steps_per_chain * 1024);
    const char* sl_base = reinterpret_cast<const char*>(a.saved_lat ? a.saved_lat + c.grow0 * J * SVL : nullptr);
    const uint32_t sl_lane = (uint32_t)((int64_t)c.n * J * SVL * 4);
    float* carry = (a.dec_carry && c.live) ? a.dec_carry + ((int64_t)net * c.tiles + c.tile) * 1024 : nullptr;
    const int64_t grow = c.grow0 + c.n;                      // this lane's chain (single-window mode addressing)

    // The Linear's input row [x_t || latent] as one tile: lane (n, g) holds columns 4g .. 4g+3 (x in [0, d), latent in [d, d+Z))
    auto latent_shifted = [&](int j) {
        f32x4 v = splat4(0.f);
        if (FULL || valid) {
            const float* lp = dec_only ? a.lat_in + grow * a.Z
                                       : reinterpret_cast<const float*>(sl_base + sl_lane + (uint32_t)j * (uint32_t)(SVL * 4)) + 16;
            for (int k = 0; k < 4; ++k) {
                const int z = 4 * g + k - a.d;
                if (z >= 0 && z < a.Z) v[k] = lp[z];
            }
        }
        return v;
    };
    auto x_of = [&](int j, int t) {
        if (dec_only) return ld_row_br<FULL>(reinterpret_cast<const char*>(a.win), (uint32_t)((grow * Lw + t) * a.d * 4), valid, a.d, g);
        const int st = beh_x_step(a, j, t);
        if (st < 0) return splat4(0.f);
        return ld_row_br<FULL>(c.hist, c.hist_lane + (uint32_t)((int64_t)st * a.h_s_t * 4), valid, a.d, g);
    };
    auto u_own = [&](f32x4 xin) {                            // own tile of ReLU(Linear([x || latent]))
        f32x4 x1[1];
        x1[0] = xin;
        return relu4(dense_tile<1>(s_lin, 24, 16 * q, x1, bfrag_lds(s_b, q)));
    };

    f32x4 h[DT], u[DT];
    for (int i = 0; i < DT; ++i) h[i] = dec_only ? ld4<FULL>(reinterpret_cast<const char*>(a.hd_in), (uint32_t)(grow * DHd * 4) + 64u * i + 16u * g, valid)
                                                 : splat4(0.f);
    if (j_lo > 0 && carry)
        for (int i = 0; i < DT; ++i) h[i] = *reinterpret_cast<const f32x4*>(carry + 256 * i + 4 * l);
    // first step's u: every quarter computes its tile, all four are exchanged
    f32x4 latsh = latent_shifted(j_lo);
    f32x4 xin = x_of(j_lo, 0) + latsh;
    put(XF_U + q, u_own(xin));
    tile_sync();
    for (int i = 0; i < DT; ++i) u[i] = get(XF_U + i);
    tile_sync();

    float beh = 0.f, stab = 0.f;
    const int rows3[3] = {16 * q, DHd + 16 * q, 2 * DHd + 16 * q};                       // r, z, n gate rows of hidden tile q
    for (int j = j_lo; j < j_hi; ++j) {
        const float scale = (dec_only || q) ? 0.f : (float)(a.d * a.N) / (window_mask_sum(a, net, j) + BEPS);
        float err = 0.f;
        f32x4 latsh_next = latsh;
        if (j + 1 < j_hi) latsh_next = latent_shifted(j + 1);
        for (int t = 0; t < Lw; ++t) {
            const uint32_t so = sd_lane + (uint32_t)(((int64_t)j * Lw + t) * 1024);           // this step's record
            // next step's Linear input (the first step of the next window uses that window's latent); loads issued early
            const bool last_t = t + 1 == Lw;
            const bool has_next = !last_t || j + 1 < j_hi;
            f32x4 xin_next = splat4(0.f);
            if (has_next) xin_next = x_of(last_t ? j + 1 : j, last_t ? 0 : t + 1) + (last_t ? latsh_next : latsh);
            f32x4 nx = splat4(0.f);
            float m = 0.f;
            if (q == 0 && !dec_only) {
                nx = ld_row_br<FULL>(c.hist, c.hist_lane + (uint32_t)((int64_t)beh_y_step(a, j, t) * a.h_s_t * 4), valid, a.d, g);
                if (FULL || valid) m = *reinterpret_cast<const float*>(c.mask + c.mask_lane + 4u * (uint32_t)beh_m_step(a, j, t));
            }
            if (q == 0) st4<FULL>(sd_base, so + cgs * REC_CG(SD_X), valid, xin);
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_U + 16 * q), valid, u[q]);
            // gates of hidden tile q
            f32x4 ai[3], ah[3];
            ai[0] = bfrag_lds(s_b + 64, q) + bfrag_lds(s_b + 256, q);
            ai[1] = bfrag_lds(s_b + 64, DT + q) + bfrag_lds(s_b + 256, DT + q);
            ai[2] = bfrag_lds(s_b + 64, 2 * DT + q);
            dense_multi<3, DT>(s_wih, DLD, rows3, 0, u, ai);                                 // W_ih u: pre_r, pre_z, gi_n
            ah[0] = ai[0];
            ah[1] = ai[1];
            ah[2] = bfrag_lds(s_b + 256, 2 * DT + q);
            dense_multi<3, DT>(s_whh, DLD, rows3, 0, h, ah);                                 // + W_hh h: pre_r, pre_z, gh_n
            const GruGates o = gru_gates(ah[0], ah[1], ai[2], ah[2], h[q]);
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_R + 16 * q), valid, o.r);
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_Z + 16 * q), valid, o.z);
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_N + 16 * q), valid, o.n);
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_HN + 16 * q), valid, o.hn);
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_H + 16 * q), valid, o.h);
            const f32x4 km = keep_tile(a, net, j, c.row, t, q, FULL || valid, c.rows);
            f32x4 act[1];
            for (int k = 0; k < 4; ++k) act[0][k] = tanh_f(o.h[k]) * (km[k] * inv_keep);
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_A + 16 * q), valid, act[0]);
            // y = W_out act + b: every quarter contracts its own tile, quarter 0 adds the partials up
            const f32x4 yp = dense_tile_k<1>(s_out, DLD, 0, 16 * q, act, q ? splat4(0.f) : bfrag_lds(s_b + 448, 0));
            put(XF_H + q, o.h);
            if (q) put(XF_Y + q - 1, yp);
            if (has_next) put(XF_U + q, u_own(xin_next));
            tile_sync();                                     // (record stores stay in flight across the rendezvous)
            for (int i = 0; i < DT; ++i) h[i] = (i == q) ? o.h : get(XF_H + i);      // (q is a constant: no select)
            if (has_next)
                for (int i = 0; i < DT; ++i) u[i] = get(XF_U + i);
            f32x4 y = yp;
            if (q == 0) y = (yp + get(XF_Y)) + (get(XF_Y + 1) + get(XF_Y + 2));
            tile_sync();
            const f32x4 xt = xin;                            // columns >= d hold the latent: masked out below
            xin = xin_next;
            if (q) continue;                                 // the rest of the step (output, loss terms) is quarter 0's
            st4<FULL>(sd_base, so + cgs * REC_CG(SD_Y), valid, y);
            if (dec_only) {
                if (FULL || valid) {
                    float* po = a.pred_out + (grow * Lw + t) * a.d + 4 * g;
                    for (int k = 0; k < 4; ++k)
                        if (4 * g + k < a.d) po[k] = y[k];
                }
                continue;
            }
            // masked L1 against the target window, stability vs the current one (stable_behavior_policy.py:226, 233-240)
            float d2 = 0.f;
            for (int k = 0; k < 4; ++k) {
                if (4 * g + k < a.d) {
                    err += fabsf(nx[k] - y[k]) * m;
                    const float df = xt[k] - y[k];
                    d2 = fmaf(df, df, d2);
                }
            }
            d2 = group_sum(d2);
            if ((FULL || valid) && g == 0) stab += fmaxf(sqrtf(d2) - a.thres, 0.f);
        }
        latsh = latsh_next;
        if (dec_only) {
            st4<FULL>(reinterpret_cast<char*>(a.hd_out), (uint32_t)(grow * DHd * 4) + 64u * q + 16u * g, valid, h[q]);
            return;
        }
        beh = fmaf(err, scale, beh);
    }
    if (j_hi < J && carry) *reinterpret_cast<f32x4*>(carry + 256 * q + 4 * l) = h[q];
    beh = chain_sum_b(group_sum(beh)) / (a.hard ? 1.0f : (float)J);
    stab = chain_sum_b(group_sum(stab)) / (float)a.E / (float)a.L / (float)J;
    if (l == 0 && q == 0 && c.live) {
        float* lp = a.loss_part + ((int64_t)net * c.tiles + c.tile) * 2;
        lp[0] = j_lo > 0 ? lp[0] + beh : beh;             // pieces accumulate
        lp[1] = j_lo > 0 ? lp[1] + stab : stab;
    }
}

__global__ __launch_bounds__(DEC_THREADS) void beh_dec_fwd_kernel(IplanBehArgs a) {
    IPLAN_DYN_LDS(smem);
    float* s_wih = smem;                                    // [192][DLD]
    float* s_whh = s_wih + 3 * DHd * DLD;                   // [192][DLD]
    float* s_lin = s_whh + 3 * DHd * DLD;                   // [64][24]   W_lin, columns [0, d + Z)
    float* s_out = s_lin + DHd * 24;                        // [16][DLD]
    float* s_b = s_out + 16 * DLD;                          // lin 64 | ih 192 | hh 192 | out 16
    float* s_xch = s_b + DEC_FWD_BIAS;                      // [DEC_TILES][XF_SLOTS][256]
    int* s_tcnt = reinterpret_cast<int*>(s_xch + DEC_TILES * XF_SLOTS * 256);           // [DEC_TILES] rendezvous counters (+ pad to 16)
    if (threadIdx.x < 16) s_tcnt[threadIdx.x] = 0;
    const float* __restrict__ PD = a.dec_params + (int64_t)blockIdx.y * a.dec_s_net;
    stage_matrix(s_wih, DLD, 3 * DHd, PD + a.dec_off[IPLAN_DEC_WIH], 3 * DHd, DHd);
    stage_matrix(s_whh, DLD, 3 * DHd, PD + a.dec_off[IPLAN_DEC_WHH], 3 * DHd, DHd);
    stage_matrix(s_lin, 24, DHd, PD + a.dec_off[IPLAN_DEC_LIN_W], DHd, a.d + a.Z);
    stage_matrix(s_out, DLD, 16, PD + a.dec_off[IPLAN_DEC_OUT_W], a.d, DHd);
    stage_vector(s_b, 64, PD + a.dec_off[IPLAN_DEC_LIN_B], 64);
    stage_vector(s_b + 64, 192, PD + a.dec_off[IPLAN_DEC_BIH], 192);
    stage_vector(s_b + 256, 192, PD + a.dec_off[IPLAN_DEC_BHH], 192);
    stage_vector(s_b + 448, 16, PD + a.dec_off[IPLAN_DEC_OUT_B], a.d);
    __syncthreads();
    // (rotating the quarters over the SIMDs per tile -- quarter 0 carries a tile's extra work -- measured 2 % slower)
    const int w = uniform_i(wave_id()), q = w & 3, ts = w >> 2;
    DecTile c;
    dec_tile(a, c, (int)blockIdx.x * DEC_TILES + ts);       // waves without a tile still take part in the block barriers
    float* xch = s_xch + ts * (XF_SLOTS * 256);
#define IPLAN_DEC_FWD(F, QQ) dec_fwd_body<F, QQ>(a, c, s_wih, s_whh, s_lin, s_out, s_b, xch, s_tcnt + ts)
    if (c.full) { if (q == 0) IPLAN_DEC_FWD(true, 0); else if (q == 1) IPLAN_DEC_FWD(true, 1); else if (q == 2) IPLAN_DEC_FWD(true, 2); else IPLAN_DEC_FWD(true, 3); }
    else { if (q == 0) IPLAN_DEC_FWD(false, 0); else if (q == 1) IPLAN_DEC_FWD(false, 1); else if (q == 2) IPLAN_DEC_FWD(false, 2); else IPLAN_DEC_FWD(false, 3); }
#undef IPLAN_DEC_FWD
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

// ------------------------------------------------------------------------------------------------------------
// decoder BPTT: row gradients for wgrad.hip (dsave_dec) and d(loss)/d(latent_j) per window (dsave_lat).
// Quarter split as in the forward: wave Q owns hidden tile Q -- its gate gradients [dr dz dn_i dn_h] (lane-local from the
// forward's record), which it parks in LDS for the other three, and output tile Q of the two backward-data products
//      du = W_ih^T [dr dz dn_i],    dh_prev = W_hh^T [dr dz dn_h] + z * dh        (K = 192: 12 k-tiles each),
// two accumulator chains issued alternately.  The NEXT step's record is fetched between the lane-local part and the MFMA part.
constexpr int XB_SLOTS = 16;                                 // exchange slots of a tile: gate * 4 + quarter, gate = dr | dz | dn_i | dn_h

template <bool FULL, int Q>
__device__ __forceinline__ void dec_bwd_body(const IplanBehArgs& a, const DecTile& c, const float* __restrict__ s_wihT,
                                             const float* __restrict__ s_whhT, const float* __restrict__ s_outT,
                                             const float* __restrict__ s_latT, float* __restrict__ xch, int* __restrict__ tcnt) {
    constexpr int TLD = 3 * DHd + 8;
    int sync_n = 0;                     // rendezvous of this tile's four quarter-waves (see the forward body)
    auto tile_sync = [&]() {
        if (DEC_BWD_TILE_SYNC) { sync_n += 4; IPLAN_TILE_SYNC(tcnt, sync_n); }
        else IPLAN_LDS_BARRIER();
    };
    const bool valid = c.valid;
    const int l = lane_id(), g = c.g, J = c.J, net = c.net, Lw = a.L;
    const float inv_keep = 1.0f / (1.0f - a.drop_p);
    auto put = [&](int slot, f32x4 v) { *reinterpret_cast<f32x4*>(xch + slot * 256 + 4 * l) = v; };
    auto get = [&](int slot) { return *reinterpret_cast<const f32x4*>(xch + slot * 256 + 4 * l); };
    const int64_t steps_per_chain = (int64_t)J * Lw;
    const char* sd_base = reinterpret_cast<const char*>(a.saved_dec + c.trow0 * steps_per_chain * SVD);
    const uint32_t sd_lane = 64u * (uint32_t)c.n + 16u * (uint32_t)g, cgs = (uint32_t)(steps_per_chain * 1024);
    char* dd_base = reinterpret_cast<char*>(a.dsave_dec + c.trow0 * steps_per_chain * DSD);
    const uint32_t dd_lane = sd_lane;
    char* dl_base = reinterpret_cast<char*>(a.dsave_lat + c.grow0 * J * DSL);
    const uint32_t dl_lane = (uint32_t)((int64_t)c.n * J * DSL * 4) + 16u * (uint32_t)g;

    // The forward's record of a step (own hidden tile), the loss target and the mask
    struct StepIn {
        f32x4 r, z, n, hn, hp, u, y, nx, xc;
        float m;
        bool first, has_xc;
    };
    // straight-line fetch: no lane- or step-dependent branch around a load, no masking (mask_step does that at the top of
    // the step that consumes the record, one MFMA phase later)
    auto load_step = [&](int j, int t, StepIn& o) {
        const uint32_t so = sd_lane + (uint32_t)(((int64_t)j * Lw + t) * 1024);
        o.first = (j == 0 && t == 0);
        o.r = ld4_raw<FULL>(sd_base, so + cgs * REC_CG(SD_R + 16 * Q), valid);
        o.z = ld4_raw<FULL>(sd_base, so + cgs * REC_CG(SD_Z + 16 * Q), valid);
        o.n = ld4_raw<FULL>(sd_base, so + cgs * REC_CG(SD_N + 16 * Q), valid);
        o.hn = ld4_raw<FULL>(sd_base, so + cgs * REC_CG(SD_HN + 16 * Q), valid);
        o.u = ld4_raw<FULL>(sd_base, so + cgs * REC_CG(SD_U + 16 * Q), valid);
        o.hp = ld4_raw<FULL>(sd_base, (o.first ? so : so - 1024u) + cgs * REC_CG(SD_H + 16 * Q), valid);   // h_{-1} = 0
        o.y = ld4_raw<FULL>(sd_base, so + cgs * REC_CG(SD_Y), valid);
        o.nx = ld_row_raw<FULL>(c.hist, c.hist_lane + (uint32_t)((int64_t)beh_y_step(a, j, t) * a.h_s_t * 4), valid, a.d, g);
        o.m = *reinterpret_cast<const float*>(c.mask + c.mask_lane + 4u * (uint32_t)beh_m_step(a, j, t));
        o.has_xc = false;
        o.xc = splat4(0.f);
        if (a.penalty != 0.f) {                             // the stability term compares the prediction with the CURRENT window's step
            const int st = beh_x_step(a, j, t);
            o.has_xc = st >= 0;
            o.xc = ld_row_raw<FULL>(c.hist, c.hist_lane + (uint32_t)((int64_t)(st < 0 ? 0 : st) * a.h_s_t * 4), valid, a.d, g);
        }
    };
    auto mask_step = [&](StepIn& o) {
        o.hp = zero_unless(!o.first, o.hp);
        for (int k = 0; k < 4; ++k) o.xc[k] = keep_if(o.has_xc && 4 * g + k < a.d, o.xc[k]);
        if (!FULL) {
            o.r = zero_unless(valid, o.r); o.z = zero_unless(valid, o.z); o.n = zero_unless(valid, o.n);
            o.hn = zero_unless(valid, o.hn); o.u = zero_unless(valid, o.u); o.hp = zero_unless(valid, o.hp);
            o.y = zero_unless(valid, o.y); o.nx = zero_unless(valid, o.nx); o.xc = zero_unless(valid, o.xc);
            o.m = keep_if(valid, o.m);
        }
    };
    // stability term (stable_behavior_policy.py:238-246): penalty / J * sum max(||x_t - y_t|| - thres, 0) / (E L)
    const float pen = a.penalty / (float)J / (float)(a.E_norm > 0 ? a.E_norm : a.E) / (float)Lw;
    // window range of this launch (the BPTT may run in pieces, see bwd_j_lo / bwd_j_hi in the header)
    const int j_hi = a.bwd_j_hi > 0 ? imin(a.bwd_j_hi, J) : J, j_lo = imax(a.bwd_j_lo, 0);
    float* carry = (a.dec_carry && c.live) ? a.dec_carry + ((int64_t)net * c.tiles + c.tile) * 1024 + 256 * Q : nullptr;
    f32x4 dhd = (j_hi < J && carry) ? *reinterpret_cast<const f32x4*>(carry + 4 * l) : splat4(0.f);
    StepIn cur;
    load_step(j_hi - 1, Lw - 1, cur);
    f32x4 hcur = ld4<FULL>(sd_base, sd_lane + (uint32_t)((((int64_t)(j_hi - 1)) * Lw + (Lw - 1)) * 1024) + cgs * REC_CG(SD_H + 16 * Q), valid);
    for (int j = j_hi - 1; j >= j_lo; --j) {
        const float scale = (float)(a.d * a.N) / (window_mask_sum(a, net, j) + BEPS) / (a.hard ? 1.0f : (float)J);
        f32x4 dlat = splat4(0.f);                           // d(loss)/d(latent_j) through this window's decoder inputs (own share)
        for (int t = Lw - 1; t >= 0; --t) {
            const uint32_t dof = dd_lane + (uint32_t)(((int64_t)j * Lw + t) * 1024);
            // ---- part A: lane-local, consumes the step's record
            mask_step(cur);
            f32x4 dy[1];
            for (int k = 0; k < 4; ++k) {
                float v = 0.f;
                if ((FULL || valid) && 4 * g + k < a.d) {
                    const float er = cur.nx[k] - cur.y[k];
                    v = -((er > 0.f) ? 1.0f : (er < 0.f ? -1.0f : 0.0f)) * cur.m * scale;
                }
                dy[0][k] = v;
            }
            if (a.penalty != 0.f) {                         // d/dy of max(||x - y||_2 - thres, 0): -(x - y) / ||x - y|| where active
                float d2 = 0.f;
                f32x4 df;
                for (int k = 0; k < 4; ++k) {
                    df[k] = (4 * g + k < a.d) ? cur.xc[k] - cur.y[k] : 0.f;
                    d2 = fmaf(df[k], df[k], d2);
                }
                const float nrm = sqrtf(group_sum(d2));
                if ((FULL || valid) && nrm > a.thres)
                    for (int k = 0; k < 4; ++k) dy[0][k] -= pen * df[k] / nrm;
            }
#ifndef BWD_ABL
#define BWD_ABL 0                        // timing ablations (scripts/build_variants.sh): 1 no dd stores, 2 no record prefetch,
#endif                                   // 3 no part-B MFMAs, 4 no second barrier, 5 no first barrier (results are WRONG under them)
            if (Q == 0 && BWD_ABL != 1) st4<FULL>(dd_base, dof + cgs * REC_CG(DD_DY), valid, dy[0]);
            const f32x4 da = dense_tile<1>(s_outT, 24, 16 * Q, dy, splat4(0.f));
            const f32x4 km = keep_tile(a, net, j, c.row, t, Q, FULL || valid, c.rows);
            f32x4 dht;
            for (int k = 0; k < 4; ++k) {
                const float th = tanh_f(hcur[k]);
                dht[k] = fmaf(da[k] * km[k] * inv_keep, 1.0f - th * th, dhd[k]);
            }
            const GruGrads o = gru_gates_bwd(dht, cur.r, cur.z, cur.n, cur.hn, cur.hp);
            if (BWD_ABL != 1) {
                st4<FULL>(dd_base, dof + cgs * REC_CG(DD_DR + 16 * Q), valid, o.dr);
                st4<FULL>(dd_base, dof + cgs * REC_CG(DD_DZ + 16 * Q), valid, o.dz);
                st4<FULL>(dd_base, dof + cgs * REC_CG(DD_DNI + 16 * Q), valid, o.dni);
                st4<FULL>(dd_base, dof + cgs * REC_CG(DD_DNH + 16 * Q), valid, o.dnh);
            }
            put(0 * 4 + Q, o.dr);
            put(1 * 4 + Q, o.dz);
            put(2 * 4 + Q, o.dni);
            put(3 * 4 + Q, o.dnh);
            const f32x4 u_own = cur.u;
            hcur = cur.hp;                                  // h_{t-1}: the next step's "current" hidden state
            IPLAN_SCHED_FENCE();
            if (BWD_ABL != 2) {                              // the last step re-reads its own record (no branch around the loads)
                const bool wrap = t == 0, more = j > j_lo;
                load_step(wrap && more ? j - 1 : j, wrap ? (more ? Lw - 1 : 0) : t - 1, cur);
            }
            if (BWD_ABL != 5) tile_sync();                   // (the record prefetch and the row-gradient stores stay in flight)
            // ---- part B: output tile Q of the two backward-data products over all 12 gate k-tiles
            f32x4 du = splat4(0.f), pd = splat4(0.f);
            // k-tile kt = gate * 4 + T (columns gate * 64 + 16 T): B operands [dr dz dn_i] for W_ih^T, [dr dz dn_h] for W_hh^T,
            // the own tile from registers, the others from the exchange slots; operands of k-tile kt + 1 are read from LDS
            // while k-tile kt's MFMAs issue
            auto b_ih = [&](int kt) { const int gate = kt >> 2, T = kt & 3;
                                      return (T == Q) ? (gate == 0 ? o.dr : (gate == 1 ? o.dz : o.dni)) : get(gate * 4 + T); };
            auto b_hn = [&](int T) { return (T == Q) ? o.dnh : get(3 * 4 + T); };
            f32x4 fa = wfrag_lds(s_wihT, TLD, 16 * Q, 0), fb = wfrag_lds(s_whhT, TLD, 16 * Q, 0);
            f32x4 bi = b_ih(0), bh = bi;
#pragma unroll
            for (int kt = 0; kt < 3 * DT; ++kt) {
                f32x4 fan = fa, fbn = fb, bin = bi, bhn = bh;
                if (kt + 1 < 3 * DT) {
                    fan = wfrag_lds(s_wihT, TLD, 16 * Q, 16 * (kt + 1));
                    fbn = wfrag_lds(s_whhT, TLD, 16 * Q, 16 * (kt + 1));
                    bin = b_ih(kt + 1);
                    bhn = (kt + 1) < 2 * DT ? bin : b_hn((kt + 1) & 3);
                }
                if (BWD_ABL != 3)
                    for (int k = 0; k < 4; ++k) {
                        du = mfma4(fa[k], bi[k], du);
                        pd = mfma4(fb[k], bh[k], pd);
                    }
                else { du += fa * bi; pd += fb * bh; }
                fa = fan; fb = fbn; bi = bin; bh = bhn;
            }
            f32x4 dup[1];
            for (int k = 0; k < 4; ++k) dup[0][k] = u_own[k] > 0.f ? du[k] : 0.f;
            if (BWD_ABL != 1) st4<FULL>(dd_base, dof + cgs * REC_CG(DD_DU + 16 * Q), valid, dup[0]);
            dhd = o.dh_direct + pd;
            dlat = dense_tile_k<1>(s_latT, DLD, 0, 16 * Q, dup, dlat);                    // through the tiled latent input
            if (BWD_ABL != 4) tile_sync();
        }
        // d(loss)/d(latent_j): sum of the four quarters' shares
        if (Q) put(Q - 1, dlat);
        tile_sync();
        if (Q == 0) st4<FULL>(dl_base, dl_lane + (uint32_t)j * (uint32_t)(DSL * 4), valid, (dlat + get(0)) + (get(1) + get(2)));
        tile_sync();
    }
    if (j_lo > 0 && carry) *reinterpret_cast<f32x4*>(carry + 4 * l) = dhd;
}

__global__ __launch_bounds__(DEC_THREADS) void beh_dec_bwd_kernel(IplanBehArgs a) {
    IPLAN_DYN_LDS(smem);
    constexpr int TLD = 3 * DHd + 8;                        // 200: ld % 16 == 8 -> conflict-free ds_read_b128 fragments
    float* s_wihT = smem;                                   // [64][200]   W_ih^T
    float* s_whhT = s_wihT + DHd * TLD;                     // [64][200]
    float* s_outT = s_whhT + DHd * TLD;                     // [64][24]    W_out^T (cols = d)
    float* s_latT = s_outT + DHd * 24;                      // [16][DLD]   W_lin[:, d:d+Z]^T
    float* s_xch = s_latT + 16 * DLD;                       // [DEC_TILES][XB_SLOTS][256]
    int* s_tcnt = reinterpret_cast<int*>(s_xch + DEC_TILES * XB_SLOTS * 256);           // [DEC_TILES] rendezvous counters (+ pad to 16)
    if (threadIdx.x < 16) s_tcnt[threadIdx.x] = 0;
    const float* __restrict__ PD = a.dec_params + (int64_t)blockIdx.y * a.dec_s_net;
    const int din = a.d + a.Z;
    stage_matrix_t(s_wihT, TLD, DHd, PD + a.dec_off[IPLAN_DEC_WIH], 3 * DHd, DHd);
    stage_matrix_t(s_whhT, TLD, DHd, PD + a.dec_off[IPLAN_DEC_WHH], 3 * DHd, DHd);
    stage_matrix_t(s_outT, 24, DHd, PD + a.dec_off[IPLAN_DEC_OUT_W], a.d, DHd);
    {   // s_latT[z][m] = W_lin[m][d + z]
        const float* Wl = PD + a.dec_off[IPLAN_DEC_LIN_W];
        for (int idx = (int)threadIdx.x; idx < 16 * DLD; idx += (int)blockDim.x) {
            const int z = idx / DLD, m = idx - z * DLD;
            s_latT[idx] = (z < a.Z && m < DHd) ? Wl[(int64_t)m * din + a.d + z] : 0.f;
        }
    }
    __syncthreads();
    // (rotating the quarters over the SIMDs per tile -- quarter 0 carries a tile's extra work -- measured 2 % slower)
    const int w = uniform_i(wave_id()), q = w & 3, ts = w >> 2;
    DecTile c;
    dec_tile(a, c, (int)blockIdx.x * DEC_TILES + ts);       // waves without a tile still take part in the block barriers
    float* xch = s_xch + ts * (XB_SLOTS * 256);
#define IPLAN_DEC_BWD(F, QQ) dec_bwd_body<F, QQ>(a, c, s_wihT, s_whhT, s_outT, s_latT, xch, s_tcnt + ts)
    if (c.full) { if (q == 0) IPLAN_DEC_BWD(true, 0); else if (q == 1) IPLAN_DEC_BWD(true, 1); else if (q == 2) IPLAN_DEC_BWD(true, 2); else IPLAN_DEC_BWD(true, 3); }
    else { if (q == 0) IPLAN_DEC_BWD(false, 0); else if (q == 1) IPLAN_DEC_BWD(false, 1); else if (q == 2) IPLAN_DEC_BWD(false, 2); else IPLAN_DEC_BWD(false, 3); }
#undef IPLAN_DEC_BWD
}

// ------------------------------------------------------------------------------------------------------------
// encoder BPTT with in-kernel weight gradients.  Per step the wave turns its 16-chain tiles of
// [dr dz dn_i dn_h | du | u | h_prev | x] through LDS into MFMA operand order and accumulates
//   dW_ih += [dr dz dn_i]^T u,  dW_hh += [dr dz dn_h]^T h_prev,  dW_lin += du^T x   (per window: dW_out += dlogit^T h_L)
// in 28 register tiles; bias gradients are lane-local sums.  One partial per wave goes to enc_part, reduced by
// beh_enc_grad_kernel in tile order (fixed summation order, no atomics).
constexpr int EP_WIH = 0, EP_WHH = 3072, EP_BIH = 6144, EP_BHH = 6240, EP_LINW = 6336, EP_LINB = 6848, EP_OUTW = 6880, EP_OUTB = 7392;
static_assert(IPLAN_BEH_ENC_PART == 7408, "enc_part layout");

// BF3 (default since round 4; IPLAN_ENC_FP32=1 selects the fp32-MFMA form): the two backward-data products of a step,
//      du = W_ih^T [dr dz dn_i],    dh_prev = W_hh^T [dr dz dn_h] + z * dh        (K = 96: three K = 32 chunks each),
// in the fp32-exact split-bf16 form: 72 bf16 MFMAs instead of 96 fp32 ones.  The kernel already holds 28 weight-gradient
// accumulator tiles (112 registers) plus the bias sums, so the weights' pieces are NOT register-resident here: they are staged
// once per workgroup in LDS in fragment order ([matrix][out tile][chunk][piece][lane] bf16x8: 36 KB) and read as 16 bytes per
// lane -- 36 ds_read_b128 per step for one wave per SIMD, nowhere near the LDS rate.  The gate gradients are split once per
// step ([dr], [dz] serve both products).  The weight-gradient accumulation stays on fp32 MFMA (operands turned through LDS).
constexpr int EB_FRAGS = 2 * 2 * 3;                          // matrix (ih | hh) x out tile x chunk
template <bool BF3>
__global__ __launch_bounds__(256) void beh_enc_bwd_kernel(IplanBehArgs a) {
    constexpr int TLE = 3 * EHd + 8;                        // 104
    __shared__ __attribute__((aligned(16))) float s_wihT[BF3 ? 16 : EHd * TLE];
    __shared__ __attribute__((aligned(16))) float s_whhT[BF3 ? 16 : EHd * TLE];
    __shared__ __attribute__((aligned(16))) bf16x8 s_wp[BF3 ? EB_FRAGS * 3 * 64 : 1];
    __shared__ __attribute__((aligned(16))) float s_outT[EHd * 24];
    __shared__ __attribute__((aligned(16))) float s_turn[4][15][256];      // per wave: 8 dg + 2 du + 2 u + 2 hp + 1 x tiles
    const float* __restrict__ PE = a.enc_params + (int64_t)blockIdx.y * a.enc_s_net;
    if (BF3) {
        for (int f = wave_id(); f < EB_FRAGS; f += 4) {      // fragment f = (matrix * 2 + T) * 3 + ch
            const int mtx = f / 6, T = (f / 3) & 1, ch = f % 3;
            const Bf3 w = wfrag_t_bf3(PE + a.enc_off[mtx ? IPLAN_ENC_WHH : IPLAN_ENC_WIH], EHd, 16 * T, 32 * ch);
            s_wp[(f * 3 + 0) * 64 + lane_id()] = w.p0;
            s_wp[(f * 3 + 1) * 64 + lane_id()] = w.p1;
            s_wp[(f * 3 + 2) * 64 + lane_id()] = w.p2;
        }
    } else {
        stage_matrix_t(s_wihT, TLE, EHd, PE + a.enc_off[IPLAN_ENC_WIH], 3 * EHd, EHd);
        stage_matrix_t(s_whhT, TLE, EHd, PE + a.enc_off[IPLAN_ENC_WHH], 3 * EHd, EHd);
    }
    stage_matrix_t(s_outT, 24, EHd, PE + a.enc_off[IPLAN_ENC_OUT_W], a.Z, EHd);
    __syncthreads();
    BehChain c;
    const bool live = beh_chain(a, c);
    const bool valid = c.valid;
    const int l = lane_id(), n = c.n, g = c.g, J = c.J;
    float (*turn)[256] = s_turn[wave_id()];                  // private to the wave: the turns below need no cross-wave barrier --
                                                             // a wave's LDS accesses execute in order (IPLAN_WAVE_SYNC: emulator rendezvous)
    const float cn = a.hard ? 1.0f : a.coef, ck = a.hard ? 0.0f : 1.0f - a.coef;

    f32x4 aWih[6][2], aWhh[6][2], aLin[2], aOut[2];
    f32x4 bG[8], bU[2], bO;
    for (int t = 0; t < 6; ++t)
        for (int u = 0; u < 2; ++u) { aWih[t][u] = splat4(0.f); aWhh[t][u] = splat4(0.f); }
    for (int u = 0; u < 2; ++u) { aLin[u] = splat4(0.f); aOut[u] = splat4(0.f); bU[u] = splat4(0.f); }
    for (int t = 0; t < 8; ++t) bG[t] = splat4(0.f);
    bO = splat4(0.f);
    // A parked tile is [16 chains][16 columns] with the columns of chain r rotated by 4 * (r >> 1): the ds_write_b128 of a
    // group of 8 lanes (chains 2k, 2k + 1 at one g) then covers all 32 banks once, and the ds_read_b32 of a group of 32
    // lanes (rows 4s + {0, 1} or {2, 3}: one rotation, banks [0, 16) and [16, 32)) is conflict free as well.  (Unrotated,
    // the stores were 4-way conflicted: SQ_LDS_BANK_CONFLICT 27 M cycles against 5 M LDS instructions, profiles/r01g_pmc_*.)
    auto park = [&](int slot, f32x4 v) { *reinterpret_cast<f32x4*>(&turn[slot][n * 16 + ((4 * g + 4 * (n >> 1)) & 15)]) = v; };
    // operand element [chain 4s + g][column i] of a parked tile (i = lane & 15)
    auto pick = [&](int slot, int s) { const int r = 4 * s + g; return turn[slot][r * 16 + ((n + 4 * (r >> 1)) & 15)]; };

    // window range of this launch (pieces, top down: see bwd_j_lo / bwd_j_hi in the header); d(loss)/d(h), d(loss)/d(latent)
    // cross the pieces in enc_carry, the weight-gradient partials accumulate in enc_part
    const int j_hi = a.bwd_j_hi > 0 ? imin(a.bwd_j_hi, J) : J, j_lo = imax(a.bwd_j_lo, 0);
    float* carry = (a.enc_carry && live) ? a.enc_carry + ((int64_t)c.net * c.tiles + c.tile) * 768 : nullptr;
    f32x4 dhe[ET], dlat = splat4(0.f);
    for (int t = 0; t < ET; ++t) dhe[t] = splat4(0.f);
    if (j_hi < J && carry) {
        for (int t = 0; t < ET; ++t) dhe[t] = *reinterpret_cast<const f32x4*>(carry + 256 * t + 4 * l);
        dlat = *reinterpret_cast<const f32x4*>(carry + 512 + 4 * l);
    }
    // The forward's record of a step.  Loads are branch-free (see ld4): a lane without a chain reads chain 0 of its net.
    struct EncIn {
        f32x4 r[ET], z[ET], n[ET], hn[ET], u[ET], hp[ET], x;
        bool first, has_x;
    };
    const bool ok = valid && live;
    const int64_t ecgs = (int64_t)J * a.L * 256;                               // floats between the record's 16-column groups
    // this lane's slice of its chain's first record (column-grouped: [tile][16-column group][step][chain][16])
    const float* rec0 = a.saved_enc + ((int64_t)c.net * c.tiles + (c.tile < c.tiles ? c.tile : 0)) * (SVE / 16) * ecgs + 16 * c.n + 4 * g;
    // fetch only: the values are masked where they are consumed (mask_step) -- an AND behind the load would pull the wait for
    // it up to the load
    auto load_step = [&](int j, int t, EncIn& o) {
        const int64_t step = (int64_t)j * a.L + t;
        o.first = step == 0;
        const float* se = rec0 + step * 256;
        const float* sp = o.first ? se : se - 256;                             // h_{-1} = 0
        for (int T = 0; T < ET; ++T) {
            o.hp[T] = *reinterpret_cast<const f32x4*>(sp + ((SE_H >> 4) + T) * ecgs);
            o.u[T] = *reinterpret_cast<const f32x4*>(se + ((SE_U >> 4) + T) * ecgs);
            o.r[T] = *reinterpret_cast<const f32x4*>(se + ((SE_R >> 4) + T) * ecgs);
            o.z[T] = *reinterpret_cast<const f32x4*>(se + ((SE_Z >> 4) + T) * ecgs);
            o.n[T] = *reinterpret_cast<const f32x4*>(se + ((SE_N >> 4) + T) * ecgs);
            o.hn[T] = *reinterpret_cast<const f32x4*>(se + ((SE_HN >> 4) + T) * ecgs);
        }
        const int st = beh_x_step(a, j, t);
        o.has_x = st >= 0;
        o.x = ld_row_raw(c.hrow + (int64_t)(st < 0 ? 0 : st) * a.h_s_t, a.d, g);
    };
    auto mask_step = [&](EncIn& o) {
        for (int T = 0; T < ET; ++T) {
            o.hp[T] = zero_unless(ok && !o.first, o.hp[T]);
            o.u[T] = zero_unless(ok, o.u[T]);
            o.r[T] = zero_unless(ok, o.r[T]);
            o.z[T] = zero_unless(ok, o.z[T]);
            o.n[T] = zero_unless(ok, o.n[T]);
            o.hn[T] = zero_unless(ok, o.hn[T]);
        }
        for (int q = 0; q < 4; ++q) o.x[q] = keep_if(ok && o.has_x && 4 * g + q < a.d, o.x[q]);
    };
    // what a window's head backward reads (new latent, d loss / d latent from the decoder, h after the window's last step):
    // fetched while the PREVIOUS window's last step runs, like the step records
    struct WinIn {
        f32x4 nl, dl, hL[ET];
    };
    auto load_win = [&](int j, WinIn& o) {
        o.nl = *reinterpret_cast<const f32x4*>(a.saved_lat + (c.grow * J + j) * SVL + 4 * g);
        o.dl = *reinterpret_cast<const f32x4*>(a.dsave_lat + (c.grow * J + j) * DSL + 4 * g);
        for (int T = 0; T < ET; ++T) o.hL[T] = *reinterpret_cast<const f32x4*>(rec0 + ((int64_t)j * a.L + (a.L - 1)) * 256 + ((SE_H >> 4) + T) * ecgs);
    };
    EncIn cur;
    WinIn win;
    load_step(j_hi - 1, a.L - 1, cur);
    load_win(j_hi - 1, win);
    for (int j = j_hi - 1; j >= j_lo; --j) {
        // ---- latent update + head backward (dlat = d(loss)/d(latent_{j+1}) on entry)
        f32x4 dlog[1], hL[ET];
        {
            const f32x4 nl = zero_unless(ok, win.nl);
            float s = 0.f;
            f32x4 dnew;
            for (int q = 0; q < 4; ++q) { dnew[q] = cn * dlat[q]; s = fmaf(nl[q], dnew[q], s); }
            s = group_sum(s);
            const f32x4 dl_dec = zero_unless(ok, win.dl);
            for (int q = 0; q < 4; ++q) {
                dlog[0][q] = nl[q] * (dnew[q] - s);
                dlat[q] = ck * dlat[q] + dl_dec[q];          // now d(loss)/d(latent_j)
                bO[q] += dlog[0][q];
            }
            for (int T = 0; T < ET; ++T) {
                hL[T] = zero_unless(ok, win.hL[T]);
                dhe[T] = dense_tile<1>(s_outT, 24, 16 * T, dlog, dhe[T]);
            }
            // dW_out += dlogit^T h_L
            park(0, dlog[0]); park(1, hL[0]); park(2, hL[1]);
            IPLAN_WAVE_SYNC();
            for (int s4 = 0; s4 < 4; ++s4) {
                const float av = pick(0, s4);
                aOut[0] = mfma4(av, pick(1, s4), aOut[0]);
                aOut[1] = mfma4(av, pick(2, s4), aOut[1]);
            }
            IPLAN_WAVE_SYNC();
        }
        for (int t = a.L - 1; t >= 0; --t) {
            f32x4 dg[4 * ET], dd[ET], u[ET], hp[ET];
            mask_step(cur);
            for (int T = 0; T < ET; ++T) {
                hp[T] = cur.hp[T];
                u[T] = cur.u[T];
                const GruGrads o = gru_gates_bwd(dhe[T], cur.r[T], cur.z[T], cur.n[T], cur.hn[T], hp[T]);
                dg[T] = o.dr; dg[ET + T] = o.dz; dg[2 * ET + T] = o.dni; dg[3 * ET + T] = o.dnh;
                dd[T] = o.dh_direct;
            }
            for (int k = 0; k < 8; ++k) bG[k] += dg[k];
            const int oe[ET] = {0, 16};
            f32x4 du[ET], dup[ET];
            for (int T = 0; T < ET; ++T) { du[T] = splat4(0.f); dhe[T] = dd[T]; }
            if (BF3) {
                // chunk ch of the gate-gradient vector: [dr] | [dz] | [dn_i] (input side) or [dn_h] (recurrent side)
                for (int ch = 0; ch < 3; ++ch) {
                    const Bf3 gi = split_bf3(dg[2 * ch], dg[2 * ch + 1]);
                    const Bf3 gh = ch < 2 ? gi : split_bf3(dg[3 * ET], dg[3 * ET + 1]);
                    Bf3 wi[ET], wh[ET];
                    for (int T = 0; T < ET; ++T) {
                        const bf16x8* fi = s_wp + ((0 * 2 + T) * 3 + ch) * 3 * 64 + l;
                        const bf16x8* fh = s_wp + ((1 * 2 + T) * 3 + ch) * 3 * 64 + l;
                        wi[T].p0 = fi[0]; wi[T].p1 = fi[64]; wi[T].p2 = fi[128];
                        wh[T].p0 = fh[0]; wh[T].p1 = fh[64]; wh[T].p2 = fh[128];
                    }
#define ENCB_ROUND(WP, XP)                                                           \
    for (int T = 0; T < ET; ++T) {                                                   \
        du[T] = mfma_bf16(wi[T].WP, gi.XP, du[T]);                                   \
        dhe[T] = mfma_bf16(wh[T].WP, gh.XP, dhe[T]);                                 \
    }
                    ENCB_ROUND(p2, p0) ENCB_ROUND(p0, p2) ENCB_ROUND(p1, p1) ENCB_ROUND(p1, p0) ENCB_ROUND(p0, p1) ENCB_ROUND(p0, p0)
#undef ENCB_ROUND
                }
            } else {
                dense_multi<ET, 3 * ET>(s_wihT, TLE, oe, 0, dg, du);                      // W_ih^T [dr dz dn_i]
                dense_multi<ET, 2 * ET>(s_whhT, TLE, oe, 0, dg, dhe);                     // W_hh^T [dr dz | dn_h]
                dense_multi<ET, ET>(s_whhT, TLE, oe, 2 * EHd, dg + 3 * ET, dhe);
            }
            for (int T = 0; T < ET; ++T) {
                for (int q = 0; q < 4; ++q) dup[T][q] = u[T][q] > 0.f ? du[T][q] : 0.f;
                bU[T] += dup[T];
            }
            // ---- weight gradients of this step
            for (int k = 0; k < 8; ++k) park(k, dg[k]);
            park(8, dup[0]); park(9, dup[1]); park(10, u[0]); park(11, u[1]); park(12, hp[0]); park(13, hp[1]); park(14, cur.x);
            IPLAN_SCHED_FENCE();
            {   // the next step's record is fetched under this step's 104 weight-gradient MFMAs (the last step re-reads its own)
                const bool wrap = t == 0, more = j > j_lo;
                load_step(wrap && more ? j - 1 : j, wrap ? (more ? a.L - 1 : 0) : t - 1, cur);
                if (wrap) load_win(more ? j - 1 : j, win);
            }
            IPLAN_WAVE_SYNC();
            for (int s4 = 0; s4 < 4; ++s4) {
                const float u0 = pick(10, s4), u1 = pick(11, s4), h0 = pick(12, s4), h1 = pick(13, s4), xv = pick(14, s4);
                for (int o = 0; o < 6; ++o) {
                    const float ai = pick(o, s4);                                          // [dr dz dn_i] tiles 0..5
                    const float ah = o < 4 ? ai : pick(o + 2, s4);                         // [dr dz dn_h]: tiles 0..3, 6, 7
                    aWih[o][0] = mfma4(ai, u0, aWih[o][0]);
                    aWih[o][1] = mfma4(ai, u1, aWih[o][1]);
                    aWhh[o][0] = mfma4(ah, h0, aWhh[o][0]);
                    aWhh[o][1] = mfma4(ah, h1, aWhh[o][1]);
                }
                aLin[0] = mfma4(pick(8, s4), xv, aLin[0]);
                aLin[1] = mfma4(pick(9, s4), xv, aLin[1]);
            }
            IPLAN_WAVE_SYNC();
        }
    }
    if (!live) return;
    if (j_lo > 0 && carry) {
        for (int t = 0; t < ET; ++t) *reinterpret_cast<f32x4*>(carry + 256 * t + 4 * l) = dhe[t];
        *reinterpret_cast<f32x4*>(carry + 512 + 4 * l) = dlat;
    }
    const bool add = j_hi < J;                              // later pieces add to the first piece's partials
    // ---- one partial per wave: accumulator tiles are in D layout (lane (col j = n, rows 4g + q))
    float* part = a.enc_part + ((int64_t)c.net * c.tiles + c.tile) * IPLAN_BEH_ENC_PART;
    auto put_part = [&](int idx, float v) { part[idx] = add ? part[idx] + v : v; };
    for (int o = 0; o < 6; ++o)
        for (int u = 0; u < 2; ++u)
            for (int q = 0; q < 4; ++q) {
                const int orow = 16 * o + 4 * g + q, col = 16 * u + n;
                put_part(EP_WIH + orow * EHd + col, aWih[o][u][q]);
                put_part(EP_WHH + orow * EHd + col, aWhh[o][u][q]);
            }
    for (int u = 0; u < 2; ++u)
        for (int q = 0; q < 4; ++q) {
            put_part(EP_LINW + (16 * u + 4 * g + q) * 16 + n, aLin[u][q]);                    // [32][16]: row = du index, col = x index
            put_part(EP_OUTW + (4 * g + q) * EHd + 16 * u + n, aOut[u][q]);                   // [16][32]: row = logit index, col = h index
        }
    // bias gradients: sums over the 16 chains of lane-local sums
    for (int k = 0; k < 8; ++k)
        for (int q = 0; q < 4; ++q) {
            const float sum = chain_sum_b(bG[k][q]);
            if (n == 0) {
                const int gate = k >> 1, T = k & 1, idx = 16 * T + 4 * g + q;              // k = 2 * gate + T
                if (gate < 3) put_part(EP_BIH + gate * EHd + idx, sum);                       // b_ih: dr dz dn_i
                if (gate < 2) put_part(EP_BHH + gate * EHd + idx, sum);                       // b_hh: dr dz ...
                if (gate == 3) put_part(EP_BHH + 2 * EHd + idx, sum);                         //       ... dn_h
            }
        }
    for (int T = 0; T < 2; ++T)
        for (int q = 0; q < 4; ++q) {
            const float sum = chain_sum_b(bU[T][q]);
            if (n == 0) put_part(EP_LINB + 16 * T + 4 * g + q, sum);
        }
    for (int q = 0; q < 4; ++q) {
        const float sum = chain_sum_b(bO[q]);
        if (n == 0) put_part(EP_OUTB + 4 * g + q, sum);
    }
}

// encoder gradient arena <- sum over the wave partials, in tile order.  grid: (ceil(P_enc / 256), n_nets)
__global__ __launch_bounds__(256) void beh_enc_grad_kernel(IplanBehArgs a) {
    const int net = (int)blockIdx.y;
    const int tiles = (a.E * a.N + 15) / 16;
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    // destination tensors in state_dict order with their partial offsets / shapes
    const int sizes[8] = {EHd * a.d, EHd, 3 * EHd * EHd, 3 * EHd * EHd, 3 * EHd, 3 * EHd, a.Z * EHd, a.Z};
    int rem = idx, which = -1;
    for (int k = 0; k < 8; ++k) {
        if (rem < sizes[k]) { which = k; break; }
        rem -= sizes[k];
    }
    if (which < 0) return;
    int src;
    switch (which) {
        case IPLAN_ENC_LIN_W: src = EP_LINW + (rem / a.d) * 16 + rem % a.d; break;
        case IPLAN_ENC_LIN_B: src = EP_LINB + rem; break;
        case IPLAN_ENC_WIH: src = EP_WIH + rem; break;
        case IPLAN_ENC_WHH: src = EP_WHH + rem; break;
        case IPLAN_ENC_BIH: src = EP_BIH + rem; break;
        case IPLAN_ENC_BHH: src = EP_BHH + rem; break;
        case IPLAN_ENC_OUT_W: src = EP_OUTW + rem; break;
        default: src = EP_OUTB + rem; break;
    }
    const float* part = a.enc_part + (int64_t)net * tiles * IPLAN_BEH_ENC_PART + src;
    float s = 0.f;
    for (int t = 0; t < tiles; ++t) s += part[(int64_t)t * IPLAN_BEH_ENC_PART];
    float* dst = a.enc_grad + (int64_t)net * a.enc_grad_s_net + a.enc_off[which] + rem;
    *dst = a.enc_grad_beta != 0.f ? fmaf(a.enc_grad_beta, *dst, s) : s;
}

// ------------------------------------------------------------------------------------------------------------
// Decoder forward, SECOND FORM (round 3; the default where it applies, IPLAN_DEC_FWD_V1=1 selects the first form: see iplan_beh_fwd).
//
// The first form is bound by fp32-MFMA + VALU issue (v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate and takes the VALU's
// slots: 416 of them per tile-step) and by LDS operand delivery (a weight fragment read from LDS serves 16 chains).  Here
//  * every contraction of the GRU runs in the fp32-exact split-bf16 form (wave_tile.h) on the bf16 matrix cores, and EVERY weight
//    piece is a loop invariant in registers -- possible because the work of a chain tile is split over EIGHT waves by
//    (hidden quarter q) x (matrix): wave B_q owns the rows (r_q, z_q, n_q) of W_hh (72 piece registers), wave A_q the same rows
//    of W_ih; a 512-thread workgroup = 4 B-waves + 4 A-waves, B_q and A_q on the same SIMD, and carries D2_TILES = 3 chain tiles
//    that every wave walks one after the other (three independent chains of work per wave);
//  * the input projection  gi_t = W_ih u_t + b  does not depend on the hidden state: the A-waves run AHEAD of the recurrence
//    (a 2-slot ring per (tile, quarter) in LDS, the result lands as the initial accumulator of B_q's chain), so the recurrent
//    critical path of a step is 36 bf16 MFMAs + the gate arithmetic + one exchange of the new hidden quarter's bf16 pieces;
//  * nothing is lock-stepped: every hand-off (u pieces among the A-waves, h pieces among the B-waves, gi from A_q to B_q, output
//    partials from the B-waves to the A-wave that owns the tile's output / loss work) is a monotonic counter in LDS, double
//    buffered so that no second rendezvous is needed.  While B_q waits it sleeps and A_q has the SIMD.
// Same arithmetic results as the first form up to fp32 round-off of the split products (< 2^-26 per product); same records.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int D2_TILES = 3, D2_THREADS = 512, D2_LAG = 2;
constexpr int D2_GI = D2_TILES * 4 * 2 * 3 * 256;          // floats: [tile][q][slot][gate][64 lanes x f32x4]
constexpr int D2_HX = D2_TILES * 2 * 2 * 3 * 256;          //         [tile][parity][k-chunk][piece][64 lanes x bf16x8]
constexpr int D2_UX = D2_HX;
constexpr int D2_YX = D2_TILES * 2 * 4 * 128;              //         [tile][parity][q][32 lanes x f32x4]  (outputs 0..7: d <= 8)
constexpr int D2_CNT_INTS = 20;                            // hcnt | ucnt | (unused x 2) | gcnt[4] | bcnt[4] | ycnt[4] | rcnt[D2_TILES] | pad
constexpr int D2_MAX_WINDOWS = 512;                        // per-window loss normalisers of a launch (floats)
constexpr int D2_LDS_FLOATS = D2_GI + D2_HX + D2_UX + D2_YX + D2_CNT_INTS + D2_MAX_WINDOWS;

typedef int i32x4 __attribute__((ext_vector_type(4)));
#ifdef IPLAN_HOST_EMULATION
__device__ inline void d2_signal(int* c) { IPLAN_WAVE_SYNC(); if (lane_id() == 0) *c += 1; }
__device__ inline void d2_wait(const int* c, int target) { while (*reinterpret_cast<const volatile int*>(c) < target) iplan_emu::yield_(); }
// four counters at once (16-byte aligned): every one whose bit is set in `live` has reached `target`
__device__ inline void d2_wait4(const int* c, int target, int live) {
    for (;;) {
        bool ok = true;
        for (int i = 0; i < 4; ++i) ok = ok && (!((live >> i) & 1) || reinterpret_cast<const volatile int*>(c)[i] >= target);
        if (ok) return;
        iplan_emu::yield_();
    }
}
#else
__device__ __forceinline__ void d2_signal(int* c) {        // (a wave's LDS operations execute in order: the data is there before the count)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63u) == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void d2_wait(const int* c, int target) {
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// four counters with ONE 16-byte LDS read per poll (the counters are dwords bumped by LDS atomics: a 128-bit read sees each of them
// whole): every one whose bit is set in `live` has reached `target`.  Costs what the single-counter poll cost.
__device__ __forceinline__ void d2_wait4(const int* c, int target, int live) {
    const i32x4 dead = {(live & 1) ? 0 : 0x7fffffff, (live & 2) ? 0 : 0x7fffffff, (live & 4) ? 0 : 0x7fffffff, (live & 8) ? 0 : 0x7fffffff};
    for (;;) {
        const i32x4 v = *reinterpret_cast<const volatile i32x4*>(c) | dead;
        const int lo = min(min(v[0], v[1]), min(v[2], v[3]));
        if (__builtin_amdgcn_readfirstlane(lo) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
#endif

// the three bf16 pieces of a lane's 4 values of ONE 16-tile (half of a K = 32 operand): x = p0 + p1 + p2 exactly
__device__ __forceinline__ void split_bf3_half(f32x4 v, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
    u32x2 q0, q1, q2;                                        // two values per conversion (wave_tile.h: split_bf3_pair)
    for (int k = 0; k < 2; ++k) {
        f32x2 w;
        w[0] = v[2 * k];
        w[1] = v[2 * k + 1];
        uint32_t a, b, c;
        split_bf3_pair(w, a, b, c);
        q0[k] = a; q1[k] = b; q2[k] = c;
    }
    p0 = __builtin_bit_cast(bf16x4, q0);
    p1 = __builtin_bit_cast(bf16x4, q1);
    p2 = __builtin_bit_cast(bf16x4, q2);
}
// 36 MFMAs: acc[gate] += W[gate][chunk] . x[chunk] for 3 gate tiles x 2 k-chunks x 6 piece products, smallest products first,
// the three accumulator chains issued round-robin
__device__ __forceinline__ void d2_contract(const Bf3 (&W)[3][2], const Bf3 (&x)[2], f32x4 (&acc)[3]) {
#define D2_ROUND(WP, XP)                     \
    for (int cc = 0; cc < 2; ++cc)           \
        for (int gt = 0; gt < 3; ++gt) acc[gt] = mfma_bf16(W[gt][cc].WP, x[cc].XP, acc[gt]);
    D2_ROUND(p2, p0) D2_ROUND(p0, p2) D2_ROUND(p1, p1) D2_ROUND(p1, p0) D2_ROUND(p0, p1) D2_ROUND(p0, p0)
#undef D2_ROUND
}

#ifndef D2_ABL
#define D2_ABL 0                        // timing ablations (results WRONG): 1 no B-wave record stores, 2 no A-wave record stores
#endif
// -DD2_CLOCKS: cycle accounting of wave B_0 / A_0 of workgroup (0, 0) per step segment (diagnostic builds only; read back with
// iplan_debug_d2_clocks).  s_memtime waits for the wave's LDS operations, so the instrumented kernel is slower than the plain one.
#if defined(D2_CLOCKS) && !defined(IPLAN_HOST_EMULATION)
__device__ long long g_d2_clk[32];
#define D2_CLK_DECL(n) long long clk_acc[n]; for (int ci = 0; ci < n; ++ci) clk_acc[ci] = 0; long long clk_last = __builtin_amdgcn_s_memtime(); const bool clk_on = blockIdx.x == 0 && blockIdx.y == 0 && q == 0
#define D2_CLK(i) do { if (clk_on) { const long long now = __builtin_amdgcn_s_memtime(); clk_acc[i] += now - clk_last; clk_last = now; } } while (0)
#define D2_CLK_OUT(base, n) do { if (clk_on && l == 0) for (int ci = 0; ci < n; ++ci) g_d2_clk[base + ci] = clk_acc[ci]; } while (0)
#else
#define D2_CLK_DECL(n) do {} while (0)
#define D2_CLK(i) do {} while (0)
#define D2_CLK_OUT(base, n) do {} while (0)
#endif

// Global loads of the A-waves.  vmcnt counts loads AND stores, in order, and the compiler's wait-count pass gives up on exact
// counts as soon as control flow lies between a load and its use (here: the spin loops of the LDS counters and the owner's
// conditional output work): the loop-carried prefetch of the next step's inputs came back as `s_waitcnt vmcnt(0)` at the top of
// every step -- a full drain, the wave's own record stores of the previous step included (5.3 of the kernel's 5.5 ms were
// spent like that).  So these loads are issued from inline asm (the compiler then inserts no wait for them), and the code
// waits by hand: D2_VM_WINDOW is smaller than the number of vector-memory operations any A-wave issues between a prefetch and
// its use WHEN ALL THREE TILES EXIST (>= 19: the wave that owns no tile's output; owners 29-32), so `vmcnt(D2_VM_WINDOW)` leaves
// the latest stores and prefetches in flight and still covers the loads about to be consumed; workgroups with fewer tiles
// (the last one of a net) drain completely.  NO compiler-visible load may be issued while such
// loads are in flight (its computed wait count would not include them): the A-waves' step loop has none.
constexpr int D2_VM_WINDOW = 16;
#ifdef IPLAN_HOST_EMULATION
__device__ inline void d2_gload(float& dst, const char* sbase, uint32_t voff) { dst = *reinterpret_cast<const float*>(sbase + voff); }
__device__ inline const char* d2_ubase(const char* p) { return p; }
template <int WINDOW> __device__ inline void d2_landed(float (&)[4], float (&)[4]) {}
template <int WINDOW> __device__ inline void d2_landed(float (&)[4], float (&)[4], float&) {}
__device__ inline void d2_drain_vm() {}
#else
__device__ __forceinline__ void d2_gload(float& dst, const char* sbase, uint32_t voff) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// a wave-uniform base address the compiler keeps in VGPRs (it cannot prove it uniform) -> SGPR pair.  v_readfirstlane is a VALU
// write of an SGPR; a vector-memory instruction that reads it as its address needs 5 wait states behind it, and nothing pads
// inside or around inline asm (guide 5.7): the s_nop stands between the pair and the loads that use it.
__device__ __forceinline__ const char* d2_ubase(const char* p) {
    const uint64_t b = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const char* u = reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    asm volatile("s_nop 4" : "+s"(u));
    return u;
}
template <int WINDOW> __device__ __forceinline__ void d2_landed(float (&a)[4], float (&b)[4]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(WINDOW) : "memory");
}
template <int WINDOW> __device__ __forceinline__ void d2_landed(float (&a)[4], float (&b)[4], float& m) {
    asm volatile("s_waitcnt vmcnt(%9)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(m) : "n"(WINDOW) : "memory");
}
__device__ __forceinline__ void d2_drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// what every wave of the workgroup knows (uniform unless noted)
struct D2Ctx {
    float *s_gi, *s_hx, *s_ux, *s_yx, *s_scale;
    int *hcnt, *ucnt, *ycnt, *rcnt, *gcnt, *bcnt;            // gcnt / bcnt: this wave's quarter
    int q, l, n, g, net, J, Lw, j_lo, j_hi, steps, YL, n_live;
    int64_t steps_per_chain;
    uint32_t cgs;                 // byte distance of the record's 16-column groups: steps per chain * 1024
    const float* PD;
};
// piece exchange slots: writer = the quarter that owns 16-tile q of the vector (half of k-chunk q >> 1), reader = everyone
__device__ __forceinline__ void d2_px_write(const D2Ctx& x, float* base, int k, int par, f32x4 v) {
    bf16x4 p0, p1, p2;
    split_bf3_half(v, p0, p1, p2);
    char* slot = reinterpret_cast<char*>(base + ((k * 2 + par) * 2 + (x.q >> 1)) * 3 * 256) + 16 * x.l + 8 * (x.q & 1);
    *reinterpret_cast<bf16x4*>(slot) = p0;
    *reinterpret_cast<bf16x4*>(slot + 1024) = p1;
    *reinterpret_cast<bf16x4*>(slot + 2048) = p2;
}
__device__ __forceinline__ void d2_px_read(const D2Ctx& x, const float* base, int k, int par, Bf3 (&v)[2]) {
    for (int cc = 0; cc < 2; ++cc) {
        const char* slot = reinterpret_cast<const char*>(base + ((k * 2 + par) * 2 + cc) * 3 * 256) + 16 * x.l;
        v[cc].p0 = *reinterpret_cast<const bf16x8*>(slot);
        v[cc].p1 = *reinterpret_cast<const bf16x8*>(slot + 1024);
        v[cc].p2 = *reinterpret_cast<const bf16x8*>(slot + 2048);
    }
}
__device__ __forceinline__ f32x4* d2_gi_slot(const D2Ctx& x, int k, int slot, int gate) {
    return reinterpret_cast<f32x4*>(x.s_gi + (((k * 4 + x.q) * 2 + slot) * 3 + gate) * 256) + x.l;
}
__device__ __forceinline__ char* d2_sd_base(const IplanBehArgs& a, const D2Ctx& x, const DecTile& ct) {
    return reinterpret_cast<char*>(a.saved_dec + ct.trow0 * x.steps_per_chain * SVD);
}
__device__ __forceinline__ uint32_t d2_sd_off(const D2Ctx& x, int step) {
    return 64u * (uint32_t)x.n + 16u * (uint32_t)x.g + (uint32_t)((int64_t)step * 1024);
}

// ---- B_q: the recurrence.  FAST = all three tiles exist and are full: no predication, no branch around any memory operation
// (a branch around a load or a store makes the compiler drain the wait counters at its join).
// SAVE_ACT: store the output layer's input (compile-time: no branch around the store; see IplanBehArgs.fwd_skip_act)
template <bool FAST, bool SAVE_ACT>
__device__ __forceinline__ void d2_recurrent(const IplanBehArgs& a, const D2Ctx& x, const DecTile (&c)[D2_TILES]) {
    const int q = x.q, l = x.l, g = x.g, net = x.net, Lw = x.Lw;
    (void)g;
    const float* Whh = x.PD + a.dec_off[IPLAN_DEC_WHH];
    Bf3 W[3][2];
    for (int gt = 0; gt < 3; ++gt)
        for (int cc = 0; cc < 2; ++cc) W[gt][cc] = wfrag_bf3(Whh, DHd, 3 * DHd, gt * DHd + 16 * q, 32 * cc);
    const f32x4 bhn = bfrag_a(x.PD + a.dec_off[IPLAN_DEC_BHH], 2 * DT + q);
    const f32x4 wout = wfrag(x.PD + a.dec_off[IPLAN_DEC_OUT_W], DHd, a.d, DHd, 0, 16 * q);        // W_out[m][16 q + 4 g ..]: rows = outputs
    const f32x4 bout = q ? splat4(0.f) : bfrag(x.PD + a.dec_off[IPLAN_DEC_OUT_B], a.d, 0);
    const float inv_keep = 1.0f / (1.0f - a.drop_p);
    f32x4 hq[D2_TILES];
#pragma unroll
    for (int k = 0; k < D2_TILES; ++k) {
        hq[k] = splat4(0.f);
        if (!FAST && !c[k].live) continue;
        if (x.j_lo > 0 && a.dec_carry)
            hq[k] = *reinterpret_cast<const f32x4*>(a.dec_carry + ((int64_t)net * c[k].tiles + c[k].tile) * 1024 + 256 * q + 4 * l);
        d2_px_write(x, x.s_hx, k, 1, hq[k]);                                         // h_{-1}: parity of step -1
    }
    d2_signal(x.hcnt);
    int live_mask = 0;                                       // tiles whose owner counts consumed outputs (rcnt[k])
#pragma unroll
    for (int k = 0; k < D2_TILES; ++k) live_mask |= (FAST || c[k].live) ? (1 << k) : 0;
    D2_CLK_DECL(8);
    int j = x.j_lo, t = 0;
    for (int s = 0; s < x.steps; ++s) {
        const uint32_t so = d2_sd_off(x, j * Lw + t);
        D2_CLK(7);
        d2_wait(x.hcnt, 4 * (s + 1));
        d2_wait(x.gcnt, s + 1);
        D2_CLK(0);
        f32x4 acc[D2_TILES][3], gin[D2_TILES];
#pragma unroll
        for (int k = 0; k < D2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            Bf3 H[2];
            d2_px_read(x, x.s_hx, k, (s + 1) & 1, H);
            acc[k][0] = *d2_gi_slot(x, k, s & 1, 0);
            acc[k][1] = *d2_gi_slot(x, k, s & 1, 1);
            gin[k] = *d2_gi_slot(x, k, s & 1, 2);
            acc[k][2] = bhn;
            d2_contract(W, H, acc[k]);
        }
        D2_CLK(1);
        d2_signal(x.bcnt);                                                           // the ring slots may be refilled
        GruGates o[D2_TILES];
#pragma unroll
        for (int k = 0; k < D2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            o[k] = gru_gates(acc[k][0], acc[k][1], gin[k], acc[k][2], hq[k]);
            hq[k] = o[k].h;
            d2_px_write(x, x.s_hx, k, s & 1, o[k].h);
        }
        d2_signal(x.hcnt);
        D2_CLK(2);
        f32x4 yp[D2_TILES];
#pragma unroll
        for (int k = 0; k < D2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            const bool valid = FAST || c[k].valid;
            char* sdb = d2_sd_base(a, x, c[k]);
            if (!(D2_ABL & 1)) {
                st4<FAST>(sdb, so + x.cgs * REC_CG(SD_R + 16 * q), valid, o[k].r);
                st4<FAST>(sdb, so + x.cgs * REC_CG(SD_Z + 16 * q), valid, o[k].z);
                st4<FAST>(sdb, so + x.cgs * REC_CG(SD_N + 16 * q), valid, o[k].n);
                st4<FAST>(sdb, so + x.cgs * REC_CG(SD_HN + 16 * q), valid, o[k].hn);
                st4<FAST>(sdb, so + x.cgs * REC_CG(SD_H + 16 * q), valid, o[k].h);
            }
            const f32x4 km = keep_tile(a, net, j, c[k].row, t, q, valid, c[k].rows);
            f32x4 act;
            for (int i = 0; i < 4; ++i) act[i] = tanh_f(o[k].h[i]) * (km[i] * inv_keep);
            if (SAVE_ACT && !(D2_ABL & 1)) st4<FAST>(sdb, so + x.cgs * REC_CG(SD_A + 16 * q), valid, act);
            yp[k] = mma_block(wout, act, bout);                                      // own share of y = W_out act + b
        }
        D2_CLK(3);
        // the output slots of step s - 2 were consumed -- by EVERY owner: one counter per tile (round 6).  The owners counted into one
        // until then and this wait read n_live (s - 1); but behind their main loop the owners finish the last D2_LAG steps WITHOUT the
        // per-step rendezvous of the A-waves (ucnt), one of them could be two outputs ahead of another, the sum was reached, the slot of
        // step s - 2 overwritten with step s's share and the slower owner stored a wrong y for the last-but-two step of the launch's last
        // window (seen at tile 108 -- the two-tile workgroup at a net's end --, t = 7, once in ~10 000 forward passes)
        if (s >= 2) d2_wait4(x.rcnt, s - 1, live_mask);
        D2_CLK(4);
#pragma unroll
        for (int k = 0; k < D2_TILES; ++k)
            if ((FAST || c[k].live) && l < x.YL) *(reinterpret_cast<f32x4*>(x.s_yx + ((k * 2 + (s & 1)) * 4 + q) * 128) + l) = yp[k];
        d2_signal(x.ycnt + q);                                                       // (per quarter: see finish_y)
        D2_CLK(5);
        if (++t == Lw) { t = 0; ++j; }
    }
    D2_CLK_OUT(0, 8);
#pragma unroll
    for (int k = 0; k < D2_TILES; ++k)
        if ((FAST || c[k].live) && x.j_hi < x.J && a.dec_carry)
            *reinterpret_cast<f32x4*>(a.dec_carry + ((int64_t)net * c[k].tiles + c[k].tile) * 1024 + 256 * q + 4 * l) = hq[k];
}

// ---- A_q: input projection, running ahead of the recurrence; A_k (k < D2_TILES) also owns tile k's output / loss work.
// Q0 = the wave that stores the Linear's input row of the record (compile-time: no branch around that store).
template <bool FAST, bool Q0>
__device__ __forceinline__ void d2_input(const IplanBehArgs& a, const D2Ctx& x, const DecTile (&c)[D2_TILES]) {
    const int q = x.q, l = x.l, n = x.n, g = x.g, Lw = x.Lw, J = x.J, din = a.d + a.Z;
    const float* Wih = x.PD + a.dec_off[IPLAN_DEC_WIH];
    Bf3 W[3][2];
    for (int gt = 0; gt < 3; ++gt)
        for (int cc = 0; cc < 2; ++cc) W[gt][cc] = wfrag_bf3(Wih, DHd, 3 * DHd, gt * DHd + 16 * q, 32 * cc);
    const float* bih = x.PD + a.dec_off[IPLAN_DEC_BIH];
    const float* bhh = x.PD + a.dec_off[IPLAN_DEC_BHH];
    const f32x4 b_r = bfrag_a(bih, q) + bfrag_a(bhh, q), b_z = bfrag_a(bih, DT + q) + bfrag_a(bhh, DT + q), b_n = bfrag_a(bih, 2 * DT + q);
    const f32x4 wlin = wfrag(x.PD + a.dec_off[IPLAN_DEC_LIN_W], din, DHd, din, 16 * q, 0);        // W_lin[16 q + m][4 g ..] (cols < d + Z)
    const f32x4 blin = bfrag_a(x.PD + a.dec_off[IPLAN_DEC_LIN_B], q);
    // Everything read from global memory is fetched ONE STEP AHEAD and branch-free (consumed where it is loaded, every tile of
    // every step would sit out an L2 / HBM round trip and the recurrence would starve on gi: first measurement of this kernel,
    // 6.1 ms; with a uniform branch around the latent's fetch the compiler drained vmcnt at every join and the "prefetch"
    // was waited for on the spot: 5.3 ms).  A lane's four entries of the Linear's input tile [x_t (d) | latent_j (Z) | 0] come
    // from two rows; every address is a UNIFORM base that moves with the step / the window (scalar arithmetic only) plus a
    // per-lane byte offset that never changes (clamped into the row), and what a lane has no use for is removed by
    // loop-invariant bit masks where the value is consumed.
    uint32_t hl[D2_TILES], lv[D2_TILES], vm[D2_TILES], cx[4], cz[4], mx[4], ml[4];
    const char* latb[D2_TILES];
    for (int i = 0; i < 4; ++i) {
        const int col = 4 * g + i, zc = col - a.d;
        mx[i] = col < a.d ? 0xFFFFFFFFu : 0u;
        ml[i] = (zc >= 0 && zc < a.Z) ? 0xFFFFFFFFu : 0u;
        cx[i] = 4u * (uint32_t)(col < a.d ? col : a.d - 1);
        cz[i] = 4u * (uint32_t)(16 + (zc < 0 ? 0 : (zc < a.Z ? zc : a.Z - 1)));
    }
#pragma unroll
    for (int k = 0; k < D2_TILES; ++k) {
        const bool valid = FAST || c[k].valid;
        vm[k] = valid ? 0xFFFFFFFFu : 0u;
        latb[k] = reinterpret_cast<const char*>(a.saved_lat + c[k].grow0 * J * SVL);
        hl[k] = valid ? c[k].hist_lane : 0u;
        lv[k] = (uint32_t)(((int64_t)(valid ? n : 0) * J) * SVL * 4);
    }
    auto bits = [](float v) { return __builtin_bit_cast(uint32_t, v); };
    auto x_fetch4 = [&](float (&v)[4], uint32_t lane_off, int st) {   // raw entries of history step max(st, 0) at the lane's offsets
        const char* hb = d2_ubase(c[0].hist + (int64_t)(st < 0 ? 0 : st) * a.h_s_t * 4);
        for (int i = 0; i < 4; ++i) d2_gload(v[i], hb, lane_off + cx[i]);
    };
    auto lat_fetch4 = [&](float (&v)[4], int k, int jj) {
        const char* lb = d2_ubase(latb[k] + (int64_t)jj * (SVL * 4));
        for (int i = 0; i < 4; ++i) d2_gload(v[i], lb, lv[k] + cz[i]);
    };
    auto x_keep = [&](const float (&v)[4], int st, uint32_t vmask) {     // zero what is not a real x entry of a real chain at a real step
        const uint32_t sm = st >= 0 ? vmask : 0u;
        f32x4 r;
        for (int i = 0; i < 4; ++i) r[i] = __builtin_bit_cast(float, bits(v[i]) & mx[i] & sm);
        return r;
    };
    // output / loss bookkeeping of the tile this wave owns (A_k <-> tile k)
    const int ko = q < D2_TILES ? q : 0;
    DecTile co = c[0];                                       // (selected, not indexed: a run-time index would put c[] into scratch)
    if (q == 1) co = c[1];
    if (q == 2) co = c[D2_TILES - 1];
    const bool owner = q < D2_TILES && (FAST || co.live);
    const bool ovalid = FAST || co.valid;
    const uint32_t xoo = q == 1 ? hl[1] : (q == 2 ? hl[D2_TILES - 1] : hl[0]);
    const uint32_t vmo = ovalid ? 0xFFFFFFFFu : 0u;
    float beh = 0.f, stab = 0.f, err = 0.f;
    struct YIn { float nx[4], xt[4], m, scale; };
    auto y_fetch = [&](YIn& o, int jj, int tt) {             // targets / mask / window normaliser of step (jj, tt)
        x_fetch4(o.nx, xoo, beh_y_step(a, jj, tt));
        x_fetch4(o.xt, xoo, beh_x_step(a, jj, tt));
        d2_gload(o.m, d2_ubase(co.mask), (ovalid ? co.mask_lane : 0u) + 4u * (uint32_t)beh_m_step(a, jj, tt));
        o.scale = x.s_scale[jj - x.j_lo];
    };
    auto finish_y = [&](int s, int jj, int tt, YIn& in) {
        d2_landed<FAST ? D2_VM_WINDOW : 0>(in.nx, in.xt, in.m);
        // ONE counter per quarter (round 6).  Until then the four B-waves counted into one and the owner waited for 4 (s + 1) -- but
        // the B-waves meet only mid-step (hcnt), not where they publish their share of y: a wave stalled in its record stores of step s
        // could be a step behind one that had already published step s + 1, the SUM still read 4 (s + 1), and the owner added the stalled
        // quarter's STALE slot (step s - 2's share): one (tile, step) of y, the loss and everything downstream off by ~1e-6 of a
        // gradient's max, once in ~300 forward passes at config 3 (profiles/r06_notes.md section 9; scripts/dev/beh_race_hunt.py)
        d2_wait4(x.ycnt, s + 1, 15);
        f32x4 yq[4];
        for (int i = 0; i < 4; ++i) {
            yq[i] = splat4(0.f);
            if (l < x.YL) yq[i] = *(reinterpret_cast<const f32x4*>(x.s_yx + ((ko * 2 + (s & 1)) * 4 + i) * 128) + l);
        }
        d2_signal(x.rcnt + ko);                              // (this owner's tile: the B-waves wait for every live tile's count)
        const f32x4 y = (yq[0] + yq[1]) + (yq[2] + yq[3]);
        st4<FAST>(d2_sd_base(a, x, co), d2_sd_off(x, jj * Lw + tt) + x.cgs * REC_CG(SD_Y), ovalid, y);
        const f32x4 nx = x_keep(in.nx, beh_y_step(a, jj, tt), vmo), xt = x_keep(in.xt, beh_x_step(a, jj, tt), vmo);
        const float m = __builtin_bit_cast(float, bits(in.m) & vmo);
        float d2 = 0.f;
        for (int i = 0; i < 4; ++i) {                        // (nx, xt and the mask are zero beyond the d real columns: y is not)
            const float yi = __builtin_bit_cast(float, bits(y[i]) & mx[i]);
            err += fabsf(nx[i] - yi) * m;
            const float df = xt[i] - yi;
            d2 = fmaf(df, df, d2);
        }
        d2 = group_sum(d2);
        stab += (ovalid && g == 0) ? fmaxf(sqrtf(d2) - a.thres, 0.f) : 0.f;
        if (tt == Lw - 1) { beh = fmaf(err, in.scale, beh); err = 0.f; }
    };
    float xr[D2_TILES][4], lr[D2_TILES][4];
    d2_drain_vm();                                           // every compiler-visible load (the weights) has landed: see d2_gload
#pragma unroll
    for (int k = 0; k < D2_TILES; ++k) {
        x_fetch4(xr[k], hl[k], beh_x_step(a, x.j_lo, 0));
        lat_fetch4(lr[k], k, x.j_lo);
    }
    YIn yin;
    y_fetch(yin, x.j_lo, 0);
    int j = x.j_lo, t = 0, jy = x.j_lo, ty = 0;               // (j, t) of step s; (jy, ty) of the next step whose output gets finished
    D2_CLK_DECL(8);
    for (int s = 0; s < x.steps; ++s) {
        int jn = j, tn = t + 1;                              // the step fetched now (clamped to the launch's last step)
        if (tn == Lw) { tn = 0; ++jn; }
        if (s + 1 >= x.steps) { jn = j; tn = t; }
        const uint32_t so = d2_sd_off(x, j * Lw + t);
        f32x4 uk[D2_TILES], xk[D2_TILES];                    // record entries: stored BEHIND the step's gi hand-off (a store that
        D2_CLK(7);                                           // waits for room in the memory pipeline must not hold up the recurrence)
#pragma unroll
        for (int k = 0; k < D2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            f32x4 xin;                                       // [x_t | latent_j | 0]: the two sources occupy disjoint columns
            d2_landed<FAST ? D2_VM_WINDOW : 0>(xr[k], lr[k]);
            {
                const uint32_t sm = beh_x_step(a, j, t) >= 0 ? vm[k] : 0u;
                for (int i = 0; i < 4; ++i) xin[i] = __builtin_bit_cast(float, (bits(xr[k][i]) & mx[i] & sm) | (bits(lr[k][i]) & ml[i] & vm[k]));
            }
            x_fetch4(xr[k], hl[k], beh_x_step(a, jn, tn));
            lat_fetch4(lr[k], k, jn);                        // (every step, not only at a window's end: no branch around a load)
            const f32x4 u = relu4(mma_block(wlin, xin, blin));
            uk[k] = u;
            if (Q0) xk[k] = xin;
            d2_px_write(x, x.s_ux, k, s & 1, u);
        }
        d2_signal(x.ucnt);
        D2_CLK(0);
        d2_wait(x.ucnt, 4 * (s + 1));
        D2_CLK(1);
        if (s >= 2) d2_wait(x.bcnt, s - 1);                                          // B_q has taken gi(s - 2) out of this ring slot
        D2_CLK(3);                                                                   // (long ago: checked before the products so that
#pragma unroll                                                                       //  each tile's result leaves the registers at once)
        for (int k = 0; k < D2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            Bf3 U[2];
            d2_px_read(x, x.s_ux, k, s & 1, U);
            f32x4 acc[3] = {b_r, b_z, b_n};
            d2_contract(W, U, acc);
            *d2_gi_slot(x, k, s & 1, 0) = acc[0];
            *d2_gi_slot(x, k, s & 1, 1) = acc[1];
            *d2_gi_slot(x, k, s & 1, 2) = acc[2];
        }
        D2_CLK(2);
        d2_signal(x.gcnt);
        D2_CLK(4);
        if (!(D2_ABL & 2)) {
#pragma unroll
            for (int k = 0; k < D2_TILES; ++k) {
                if (!FAST && !c[k].live) continue;
                char* sdb = d2_sd_base(a, x, c[k]);
                if (Q0) st4<FAST>(sdb, so + x.cgs * REC_CG(SD_X), FAST || c[k].valid, xk[k]);
                st4<FAST>(sdb, so + x.cgs * REC_CG(SD_U + 16 * q), FAST || c[k].valid, uk[k]);
            }
        }
        if (owner && s >= D2_LAG) {
            finish_y(s - D2_LAG, jy, ty, yin);
            if (++ty == Lw) { ty = 0; ++jy; }
            y_fetch(yin, jy, ty);                            // (s - D2_LAG + 1 < steps: always a step of this launch)
        }
        D2_CLK(5);
        if (++t == Lw) { t = 0; ++j; }
    }
    D2_CLK_OUT(8, 8);
    if (!owner) return;
    for (int s = imax(x.steps - D2_LAG, 0); s < x.steps; ++s) {
        d2_drain_vm();                                       // (the tail: fewer operations behind the fetch than D2_VM_WINDOW)
        finish_y(s, jy, ty, yin);
        if (s + 1 < x.steps) {
            if (++ty == Lw) { ty = 0; ++jy; }
            y_fetch(yin, jy, ty);
        }
    }
    d2_drain_vm();
    beh = chain_sum_b(group_sum(beh)) / (a.hard ? 1.0f : (float)J);
    stab = chain_sum_b(group_sum(stab)) / (float)a.E / (float)a.L / (float)J;
    if (l == 0) {
        float* lp = a.loss_part + ((int64_t)x.net * co.tiles + co.tile) * 2;
        lp[0] = x.j_lo > 0 ? lp[0] + beh : beh;             // pieces accumulate
        lp[1] = x.j_lo > 0 ? lp[1] + stab : stab;
    }
}

__global__ __launch_bounds__(D2_THREADS, 2) void beh_dec_fwd2_kernel(IplanBehArgs a) {
    IPLAN_DYN_LDS(smem);
    D2Ctx x;
    x.s_gi = smem;
    x.s_hx = x.s_gi + D2_GI;
    x.s_ux = x.s_hx + D2_HX;
    x.s_yx = x.s_ux + D2_UX;
    int* s_cnt = reinterpret_cast<int*>(x.s_yx + D2_YX);
    x.s_scale = reinterpret_cast<float*>(s_cnt + D2_CNT_INTS);
    if (threadIdx.x < D2_CNT_INTS) s_cnt[threadIdx.x] = 0;
    const int w = uniform_i(wave_id()), role = w >> 2;                                // role 0: recurrent wave B_q, 1: input wave A_q
    x.q = w & 3;
    x.l = lane_id(); x.n = x.l & 15; x.g = x.l >> 4; x.net = (int)blockIdx.y;
    x.PD = a.dec_params + (int64_t)x.net * a.dec_s_net;
    x.Lw = a.L;
    DecTile c[D2_TILES];
#pragma unroll
    for (int k = 0; k < D2_TILES; ++k) dec_tile(a, c[k], (int)blockIdx.x * D2_TILES + k);
    x.J = c[0].J;
    x.j_lo = imax(a.fwd_j_lo, 0);
    x.j_hi = a.fwd_j_hi > 0 ? imin(a.fwd_j_hi, x.J) : x.J;
    x.steps = (x.j_hi - x.j_lo) * x.Lw;
    x.steps_per_chain = (int64_t)x.J * x.Lw;
    x.cgs = (uint32_t)(x.steps_per_chain * 1024);
    x.YL = 16 * ((a.d + 3) / 4);                                                      // lanes that hold real outputs
    x.hcnt = s_cnt + 0; x.ucnt = s_cnt + 1; x.ycnt = s_cnt + 12; x.rcnt = s_cnt + 16; x.gcnt = s_cnt + 4 + x.q; x.bcnt = s_cnt + 8 + x.q;
    x.n_live = 0;
    bool fast = true;
#pragma unroll
    for (int k = 0; k < D2_TILES; ++k) { x.n_live += c[k].live ? 1 : 0; fast = fast && c[k].full; }
    // the loss normaliser of every window of this launch (the mask alone decides it): wave w takes windows w, w + 8, ...
    for (int jj = x.j_lo + w; jj < x.j_hi; jj += D2_THREADS / 64) {
        const float sc = (float)(a.d * a.N) / (window_mask_sum(a, x.net, jj) + BEPS);
        if (x.l == 0) x.s_scale[jj - x.j_lo] = sc;
    }
    __syncthreads();
    const bool fastu = uniform_i(fast ? 1 : 0) != 0;
    if (role == 0) {
        if (a.fwd_skip_act) { if (fastu) d2_recurrent<true, false>(a, x, c); else d2_recurrent<false, false>(a, x, c); }
        else { if (fastu) d2_recurrent<true, true>(a, x, c); else d2_recurrent<false, true>(a, x, c); }
    } else if (x.q == 0) {
        if (fastu) d2_input<true, true>(a, x, c); else d2_input<false, true>(a, x, c);
    } else {
        if (fastu) d2_input<true, false>(a, x, c); else d2_input<false, false>(a, x, c);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Decoder BPTT, SECOND FORM (round 4; the default where it applies, IPLAN_DEC_BWD_V1=1 selects the first form: see iplan_beh_bwd).
//
// Same reasoning as the forward's second form: the first form is bound by fp32-MFMA + VALU issue (104 v_mfma_f32_16x16x4_f32
// per quarter-step, 3 quarter-waves per SIMD).  Here the two backward-data products of a step,
//      du = W_ih^T [dr dz dn_i]          dh_prev = W_hh^T [dr dz dn_h] + z * dh          (K = 192 = six K = 32 chunks each),
// run in the fp32-exact split-bf16 form with ALL weight pieces register-resident: a chain tile's step is split over EIGHT waves by
// (hidden quarter q) x (matrix) -- wave B_q owns output rows 16q .. 16q+15 of W_hh^T (72 piece registers), the gate backward of
// its 16 hidden units and the recurrence  d loss / d h;  wave A_q owns the same rows of W_ih^T, the ReLU / Linear backward of its
// 16 units of u and the step's share of d loss / d latent.  A 512-thread workgroup = 4 B-waves + 4 A-waves carries B2_TILES = 3
// chain tiles that every wave walks one after the other.
//  * the gate gradients of a step cross the tile as bf16 PIECES ([dr | dz | dn_i | dn_h]: 8 K-chunks x 3 pieces x 1 KiB per
//    tile, written once by the B-wave that owns the units, read by the 3 other B-waves and the 4 A-waves);
//  * nothing on the recurrent path needs du: the A-waves consume the pieces behind the B-waves and never hold them up;
//  * hand-offs are monotonic LDS counters (gcnt: pieces of step s written, 4 per step; rcnt: pieces of step s read, 8 per step --
//    a B-wave overwrites the slots only when rcnt says every reader is done), no workgroup barrier inside the episode;
//  * record loads are issued one step ahead, branch-free, masked where they are consumed (as in the first form).
// Same arithmetic results as the first form up to fp32 round-off of the split products; same records.
constexpr int B2_TILES = 3, B2_THREADS = 512;
constexpr int B2_GX = B2_TILES * 8 * 3 * 256;               // floats: [tile][chunk = gate * 2 + half][piece][64 lanes x bf16x8]
constexpr int B2_LX = 2 * B2_TILES * 4 * 256;               //         [window parity][tile][q][64 lanes x f32x4]  d loss / d latent shares
constexpr int B2_DY = 4 * B2_TILES * 256;                   //         [step & 3][tile][64 lanes x f32x4]  loss gradient of a step (published 2 ahead)
constexpr int B2_TURN = 8 * 2 * 256;                        //         [wave][2][16 x 16]  operand tiles turned into MFMA A / B order (thin weight gradients)
constexpr int B2_CNT_INTS = 16;                             // gcnt | rcnt | lcnt | ocnt | dcnt[B2_TILES] | pad
constexpr int B2_LDS_FLOATS = B2_GX + B2_LX + B2_DY + B2_TURN + B2_CNT_INTS + D2_MAX_WINDOWS;
// thin weight-gradient partials of a workgroup (IplanBehArgs.dec_thin_part)
constexpr int TP_WOUT = 0, TP_BOUT = 1024, TP_WLIN = 1040, TP_BLIN = 2064;
static_assert(IPLAN_BEH_DEC_THIN_PART >= TP_BLIN + 64 && IPLAN_BEH_DEC_THIN_PART % 16 == 0, "dec_thin_part layout");

struct B2Ctx {
    float *s_gx, *s_lx, *s_dy, *s_turn, *s_scale;                // s_turn: this wave's two 16 x 16 turn tiles
    int *gcnt, *rcnt, *lcnt, *ocnt, *dcnt;
    int q, l, n, g, net, J, Lw, j_lo, j_hi, steps, n_live;
    int64_t steps_per_chain;
    uint32_t cgs;
    const float* PD;
};
// pieces of the own 16 units of gate `gate` (0 dr, 1 dz, 2 dn_i, 3 dn_h) of tile k: half (q & 1) of chunk gate * 2 + (q >> 1)
__device__ __forceinline__ void b2_g_write(const B2Ctx& x, int k, int gate, f32x4 v) {
    bf16x4 p0, p1, p2;
    split_bf3_half(v, p0, p1, p2);
    char* slot = reinterpret_cast<char*>(x.s_gx + ((k * 8 + gate * 2 + (x.q >> 1)) * 3) * 256) + 16 * x.l + 8 * (x.q & 1);
    *reinterpret_cast<bf16x4*>(slot) = p0;
    *reinterpret_cast<bf16x4*>(slot + 1024) = p1;
    *reinterpret_cast<bf16x4*>(slot + 2048) = p2;
}
__device__ __forceinline__ Bf3 b2_g_read(const B2Ctx& x, int k, int chunk) {
    const char* slot = reinterpret_cast<const char*>(x.s_gx + ((k * 8 + chunk) * 3) * 256) + 16 * x.l;
    Bf3 v;
    v.p0 = *reinterpret_cast<const bf16x8*>(slot);
    v.p1 = *reinterpret_cast<const bf16x8*>(slot + 1024);
    v.p2 = *reinterpret_cast<const bf16x8*>(slot + 2048);
    return v;
}
// acc += W^T[own 16 rows][chunk] . g[chunk]: six piece products, smallest first
__device__ __forceinline__ f32x4 b2_chunk(const Bf3& W, const Bf3& G, f32x4 acc) {
    acc = mfma_bf16(W.p2, G.p0, acc); acc = mfma_bf16(W.p0, G.p2, acc); acc = mfma_bf16(W.p1, G.p1, acc);
    acc = mfma_bf16(W.p1, G.p0, acc); acc = mfma_bf16(W.p0, G.p1, acc); acc = mfma_bf16(W.p0, G.p0, acc);
    return acc;
}

// Thin weight gradients in the kernel (IplanBehArgs.dec_thin_part): dW[o][k] += sum over the tile's 16 chains of P[chain][o] Q[chain][k]
// for two per-chain tiles P, Q held in the D layout (lane (n = chain, g): columns 4g .. 4g+3) -- both are turned through a
// wave-private LDS tile into MFMA operand order (A[o][chain], B[chain][k]: the contraction runs over the chains), 4 fp32 MFMAs.
// Parked tiles use the encoder BPTT's rotation (conflict-free ds_write_b128 / ds_read_b32, see beh_enc_bwd_kernel).
__device__ __forceinline__ f32x4 b2_outer_acc(const B2Ctx& x, f32x4 P, f32x4 Q, f32x4 acc) {
    const int n = x.n, g = x.g;
    float* t0 = x.s_turn;
    float* t1 = x.s_turn + 256;
    *reinterpret_cast<f32x4*>(&t0[n * 16 + ((4 * g + 4 * (n >> 1)) & 15)]) = P;
    *reinterpret_cast<f32x4*>(&t1[n * 16 + ((4 * g + 4 * (n >> 1)) & 15)]) = Q;
    IPLAN_WAVE_SYNC();
    for (int s4 = 0; s4 < 4; ++s4) {
        const int r = 4 * s4 + g, o = r * 16 + ((n + 4 * (r >> 1)) & 15);
        acc = mfma4(t0[o], t1[o], acc);
    }
    IPLAN_WAVE_SYNC();
    return acc;
}
// partial tile (D layout: rows 4g + r, column n) and bias sums into the workgroup's slot; `add`: a later window-range piece
__device__ __forceinline__ void b2_store_thin(float* part, bool add, f32x4 acc, int row0, int ld, int col0, int n, int g) {
    for (int r = 0; r < 4; ++r) {
        float* p = part + (row0 + 4 * g + r) * ld + col0 + n;
        *p = add ? *p + acc[r] : acc[r];
    }
}

// ---- B_q: gate backward of the own 16 hidden units + the recurrence.  FAST = all three tiles exist and are full.
// Register budget (two waves per SIMD: 256): 72 weight-piece registers + the prefetched record of the NEXT step for three tiles
// (r, z, n, hn, h_prev: 60) + the carried d loss / d h and tanh' inputs (24).  The loss gradient dy of a step needs three more
// record fields (y, target, mask) and does not depend on the recurrence: the A-wave that owns a tile computes it one step AHEAD
// and hands it over through LDS (b2_input), which keeps those fields out of this wave's registers.
template <bool FAST>
__device__ __forceinline__ void b2_recurrent(const IplanBehArgs& a, const B2Ctx& x, const DecTile (&c)[B2_TILES]) {
    const int q = x.q, l = x.l, g = x.g, net = x.net, Lw = x.Lw, J = x.J;
    const float* Whh = x.PD + a.dec_off[IPLAN_DEC_WHH];
    Bf3 W[6];
    for (int ch = 0; ch < 6; ++ch) W[ch] = wfrag_t_bf3(Whh, DHd, 16 * q, 32 * ch);
    // W_out^T rows 16q .. (hidden units) x K = the d outputs: lane (m, g) holds W_out[4g + r][16q + m]
    f32x4 woutT;
    {
        const float* Wo = x.PD + a.dec_off[IPLAN_DEC_OUT_W];
        for (int r = 0; r < 4; ++r) woutT[r] = (4 * g + r < a.d) ? Wo[(int64_t)(4 * g + r) * DHd + 16 * q + (l & 15)] : 0.f;
    }
    const float inv_keep = 1.0f / (1.0f - a.drop_p);
    const uint32_t sd_lane = 64u * (uint32_t)x.n + 16u * (uint32_t)g;
    struct StepIn {
        f32x4 r, z, n, hn, hp;
    };
    const char* sdb[B2_TILES];
    char* ddb[B2_TILES];
    f32x4 dhd[B2_TILES], hcur[B2_TILES], dhdir[B2_TILES];
    StepIn cur[B2_TILES];
    const bool thin = a.dec_thin_part != nullptr;               // (uniform) out.weight / out.bias gradients accumulated here
    f32x4 gWout = splat4(0.f), gbout = splat4(0.f);              // (one accumulator for the three tiles: this wave is at its register budget)
    auto load_step = [&](int k, int j, int t, StepIn& o) {       // straight-line fetch, masked where it is consumed
        const bool valid = FAST || c[k].valid;
        const uint32_t so = sd_lane + (uint32_t)(((int64_t)j * Lw + t) * 1024);
        const bool first = (j == 0 && t == 0);
        o.r = ld4_raw<FAST>(sdb[k], so + x.cgs * REC_CG(SD_R + 16 * q), valid);
        o.z = ld4_raw<FAST>(sdb[k], so + x.cgs * REC_CG(SD_Z + 16 * q), valid);
        o.n = ld4_raw<FAST>(sdb[k], so + x.cgs * REC_CG(SD_N + 16 * q), valid);
        o.hn = ld4_raw<FAST>(sdb[k], so + x.cgs * REC_CG(SD_HN + 16 * q), valid);
        o.hp = ld4_raw<FAST>(sdb[k], (first ? so : so - 1024u) + x.cgs * REC_CG(SD_H + 16 * q), valid);
    };
#pragma unroll
    for (int k = 0; k < B2_TILES; ++k) {
        sdb[k] = reinterpret_cast<const char*>(a.saved_dec + c[k].trow0 * x.steps_per_chain * SVD);
        ddb[k] = reinterpret_cast<char*>(a.dsave_dec + c[k].trow0 * x.steps_per_chain * DSD);
        dhd[k] = splat4(0.f);
        hcur[k] = splat4(0.f);
        dhdir[k] = splat4(0.f);
        if (!FAST && !c[k].live) continue;
        if (x.j_hi < J && a.dec_carry)
            dhd[k] = *reinterpret_cast<const f32x4*>(a.dec_carry + ((int64_t)net * c[k].tiles + c[k].tile) * 1024 + 256 * q + 4 * l);
        load_step(k, x.j_hi - 1, Lw - 1, cur[k]);
        hcur[k] = ld4<FAST>(sdb[k], sd_lane + (uint32_t)((((int64_t)(x.j_hi - 1)) * Lw + (Lw - 1)) * 1024) + x.cgs * REC_CG(SD_H + 16 * q),
                            FAST || c[k].valid);
    }
    int j = x.j_hi - 1, t = Lw - 1;
    for (int s = 0; s < x.steps; ++s) {
        const bool first = (j == 0 && t == 0);
        const uint32_t dof = sd_lane + (uint32_t)(((int64_t)j * Lw + t) * 1024);
        int jn = j, tn = t - 1;                                   // the step fetched now (the last one re-reads itself: no branch)
        if (tn < 0) { tn = Lw - 1; --jn; }
        if (s + 1 >= x.steps) { jn = j; tn = t; }
        d2_wait(x.rcnt, 8 * s);                                   // every reader is done with step s - 1's pieces: the slots are free
        // dy of step s is there (published two steps ahead).  ONE COUNTER PER TILE: a single counter over the three owners
        // reaches its target at step 0 with (2, 1, 0) publications -- two owners through dy(0), dy(1) and the third not started
        // (cold start) -- and this wave would read an unwritten slot; from step 1 on the rcnt wait above already implies
        // that every owner has finished its previous iteration.  [found on the GPU as a rare run-to-run difference of the
        // gradients: scripts/dev/beh_repro.py; the fibre emulator schedules the owners first]
#pragma unroll
        for (int k = 0; k < B2_TILES; ++k)
            if (FAST || c[k].live) d2_wait(x.dcnt + k, s + 1);
        // ---- lane-local: output / tanh / dropout backward, gate backward of the own units; pieces published tile by tile
#pragma unroll
        for (int k = 0; k < B2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            const bool valid = FAST || c[k].valid;
            StepIn in = cur[k];
            in.hp = zero_unless(!first, in.hp);
            if (!FAST) {
                in.r = zero_unless(valid, in.r); in.z = zero_unless(valid, in.z); in.n = zero_unless(valid, in.n);
                in.hn = zero_unless(valid, in.hn); in.hp = zero_unless(valid, in.hp);
            }
            const f32x4 dy = *(reinterpret_cast<const f32x4*>(x.s_dy + ((s & 3) * B2_TILES + k) * 256) + l);
            const f32x4 da = mma_block(woutT, dy, splat4(0.f));
            const f32x4 km = keep_tile(a, net, j, c[k].row, t, q, valid, c[k].rows);
            f32x4 dht, act;
            for (int i = 0; i < 4; ++i) {
                const float th = tanh_f(hcur[k][i]);
                act[i] = th * (km[i] * inv_keep);                  // the output layer's input, as the forward formed it
                dht[i] = fmaf(da[i] * km[i] * inv_keep, 1.0f - th * th, dhd[k][i]);
            }
            if (thin) {                                            // d out.weight[:, own units] += dy^T act  (dy is zero for absent chains)
                gWout = b2_outer_acc(x, dy, act, gWout);
                if (q == 0) gbout += dy;
            }
            const GruGrads o = gru_gates_bwd(dht, in.r, in.z, in.n, in.hn, in.hp);
            st4<FAST>(ddb[k], dof + x.cgs * REC_CG(DD_DR + 16 * q), valid, o.dr);
            st4<FAST>(ddb[k], dof + x.cgs * REC_CG(DD_DZ + 16 * q), valid, o.dz);
            st4<FAST>(ddb[k], dof + x.cgs * REC_CG(DD_DNI + 16 * q), valid, o.dni);
            st4<FAST>(ddb[k], dof + x.cgs * REC_CG(DD_DNH + 16 * q), valid, o.dnh);
            b2_g_write(x, k, 0, o.dr);
            b2_g_write(x, k, 1, o.dz);
            b2_g_write(x, k, 2, o.dni);
            b2_g_write(x, k, 3, o.dnh);
            dhdir[k] = o.dh_direct;
            hcur[k] = in.hp;                                      // h_{t-1}: the next step's "current" hidden state
            load_step(k, jn, tn, cur[k]);                         // next step's record: in flight across the exchange and the MFMAs
        }
        d2_signal(x.gcnt);
        d2_wait(x.gcnt, 4 * (s + 1));
        // ---- own output rows of W_hh^T [dr dz dn_h] (pieces read chunk by chunk)
#pragma unroll
        for (int k = 0; k < B2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            f32x4 pd = dhdir[k];
            for (int ch = 0; ch < 6; ++ch) pd = b2_chunk(W[ch], b2_g_read(x, k, ch < 4 ? ch : ch + 2), pd);
            dhd[k] = pd;
        }
        d2_signal(x.rcnt);                                         // (waits for this wave's LDS reads, not for its MFMAs)
        if (--t < 0) { t = Lw - 1; --j; }
    }
#pragma unroll
    for (int k = 0; k < B2_TILES; ++k)
        if ((FAST || c[k].live) && x.j_lo > 0 && a.dec_carry)
            *reinterpret_cast<f32x4*>(a.dec_carry + ((int64_t)net * c[k].tiles + c[k].tile) * 1024 + 256 * q + 4 * l) = dhd[k];
    if (thin) {
        float* part = a.dec_thin_part + ((int64_t)net * gridDim.x + blockIdx.x) * IPLAN_BEH_DEC_THIN_PART;
        const bool add = x.j_hi < J;
        b2_store_thin(part + TP_WOUT, add, gWout, 0, DHd, 16 * q, x.n, g);          // [output 4g + r][unit 16q + n]
        if (q == 0)
            for (int r = 0; r < 4; ++r) {
                const float sum = chain_sum_b(gbout[r]);
                if (x.n == 0) part[TP_BOUT + 4 * g + r] = add ? part[TP_BOUT + 4 * g + r] + sum : sum;
            }
    }
}

// ---- A_q: du = W_ih^T [dr dz dn_i] for the own 16 units of u, ReLU backward, the step's share of d loss / d latent.
// A_k (k < B2_TILES) owns tile k's bookkeeping: it computes the loss gradient dy of every step ONE STEP AHEAD of the recurrence
// (dy depends on the forward's record alone) and publishes it for the four B-waves through a 2-slot LDS ring, stores it for the
// weight-gradient contraction, and adds up the four quarters' shares of d loss / d latent_j at every window's end.
template <bool FAST>
__device__ __forceinline__ void b2_input(const IplanBehArgs& a, const B2Ctx& x, const DecTile (&c)[B2_TILES]) {
    const int q = x.q, l = x.l, g = x.g, Lw = x.Lw, J = x.J, din = a.d + a.Z;
    const float* Wih = x.PD + a.dec_off[IPLAN_DEC_WIH];
    Bf3 W[6];
    for (int ch = 0; ch < 6; ++ch) W[ch] = wfrag_t_bf3(Wih, DHd, 16 * q, 32 * ch);
    // (W_lin[:, d : d + Z])^T rows z x K = the own 16 units: lane (m = z, g) holds W_lin[16q + 4g + r][d + z]
    f32x4 wlatT;
    {
        const float* Wl = x.PD + a.dec_off[IPLAN_DEC_LIN_W];
        const int z = l & 15;
        for (int r = 0; r < 4; ++r) wlatT[r] = z < a.Z ? Wl[(int64_t)(16 * q + 4 * g + r) * din + a.d + z] : 0.f;
    }
    const uint32_t sd_lane = 64u * (uint32_t)x.n + 16u * (uint32_t)g;
    const char* sdb[B2_TILES];
    char* ddb[B2_TILES];
    f32x4 u[B2_TILES], xin[B2_TILES], dlat[B2_TILES];
    const bool thin = a.dec_thin_part != nullptr;               // (uniform) linear.weight / linear.bias gradients accumulated here
    f32x4 gWlin[B2_TILES], gblin = splat4(0.f);
    for (int k = 0; k < B2_TILES; ++k) gWlin[k] = splat4(0.f);
#pragma unroll
    for (int k = 0; k < B2_TILES; ++k) {
        sdb[k] = reinterpret_cast<const char*>(a.saved_dec + c[k].trow0 * x.steps_per_chain * SVD);
        ddb[k] = reinterpret_cast<char*>(a.dsave_dec + c[k].trow0 * x.steps_per_chain * DSD);
        dlat[k] = splat4(0.f);
        u[k] = xin[k] = splat4(0.f);
        if (!FAST && !c[k].live) continue;
        const uint32_t so = sd_lane + (uint32_t)((((int64_t)(x.j_hi - 1)) * Lw + (Lw - 1)) * 1024);
        u[k] = ld4_raw<FAST>(sdb[k], so + x.cgs * REC_CG(SD_U + 16 * q), FAST || c[k].valid);
        xin[k] = ld4_raw<FAST>(sdb[k], so + x.cgs * REC_CG(SD_X), FAST || c[k].valid);     // the Linear's input row [x_t || latent_j]
    }
    // the tile this wave keeps the books of (selected, not indexed: a run-time index would put c[] into scratch)
    DecTile co = c[0];
    if (q == 1) co = c[1];
    if (q == 2) co = c[B2_TILES - 1];
    const int ko = q < B2_TILES ? q : 0;
    const bool owner = q < B2_TILES && (FAST || co.live);
    const bool ovalid = FAST || co.valid;
    const char* sdo = reinterpret_cast<const char*>(a.saved_dec + co.trow0 * x.steps_per_chain * SVD);
    char* ddo = reinterpret_cast<char*>(a.dsave_dec + co.trow0 * x.steps_per_chain * DSD);
    char* dl_base = reinterpret_cast<char*>(a.dsave_lat + co.grow0 * J * DSL);
    const uint32_t dl_lane = (uint32_t)((int64_t)co.n * J * DSL * 4) + 16u * (uint32_t)g;
    const float pen = a.penalty / (float)J / (float)(a.E_norm > 0 ? a.E_norm : a.E) / (float)Lw;
    struct YIn {
        f32x4 y, nx, xc;
        float m;
    };
    auto y_fetch = [&](YIn& o, int jj, int tt) {               // what dy of step (jj, tt) needs; masked where it is consumed
        const uint32_t so = sd_lane + (uint32_t)(((int64_t)jj * Lw + tt) * 1024);
        o.y = ld4_raw<FAST>(sdo, so + x.cgs * REC_CG(SD_Y), ovalid);
        o.nx = ld_row_raw<FAST>(co.hist, co.hist_lane + (uint32_t)((int64_t)beh_y_step(a, jj, tt) * a.h_s_t * 4), ovalid, a.d, g);
        o.m = *reinterpret_cast<const float*>(co.mask + (ovalid ? co.mask_lane : 0u) + 4u * (uint32_t)beh_m_step(a, jj, tt));
        o.xc = splat4(0.f);
        if (a.penalty != 0.f) {
            const int st = beh_x_step(a, jj, tt);
            o.xc = ld_row_raw<FAST>(co.hist, co.hist_lane + (uint32_t)((int64_t)(st < 0 ? 0 : st) * a.h_s_t * 4), ovalid, a.d, g);
        }
    };
    // dy of step (jj, tt) = index sidx of this launch: into ring slot sidx & 1, and into the row-gradient record
    auto publish_dy = [&](const YIn& in, int jj, int tt, int sidx) {
        const float scale = x.s_scale[jj - x.j_lo];
        f32x4 dy;
        for (int i = 0; i < 4; ++i) {
            float v = 0.f;
            if (ovalid && 4 * g + i < a.d) {
                const float er = in.nx[i] - in.y[i];
                v = -((er > 0.f) ? 1.0f : (er < 0.f ? -1.0f : 0.0f)) * in.m * scale;
            }
            dy[i] = v;
        }
        if (a.penalty != 0.f) {                                // d/dy of max(||x - y||_2 - thres, 0): -(x - y) / ||x - y|| where active
            const bool has_xc = beh_x_step(a, jj, tt) >= 0;
            float d2 = 0.f;
            f32x4 df;
            for (int i = 0; i < 4; ++i) {
                df[i] = (ovalid && 4 * g + i < a.d) ? (has_xc ? in.xc[i] : 0.f) - in.y[i] : 0.f;
                d2 = fmaf(df[i], df[i], d2);
            }
            const float nrm = sqrtf(group_sum(d2));
            if (ovalid && nrm > a.thres)
                for (int i = 0; i < 4; ++i) dy[i] -= pen * df[i] / nrm;
        }
        *(reinterpret_cast<f32x4*>(x.s_dy + ((sidx & 3) * B2_TILES + ko) * 256) + l) = dy;
        d2_signal(x.dcnt + ko);
        if (!thin) st4<FAST>(ddo, sd_lane + (uint32_t)(((int64_t)jj * Lw + tt) * 1024) + x.cgs * REC_CG(DD_DY), ovalid, dy);
    };
    int j = x.j_hi - 1, t = Lw - 1, w = 0;                          // w: windows finished in this launch
    // (jy, ty): the step whose dy is published next; the ring runs TWO steps ahead of the recurrence, fed at the END of an
    // iteration -- the fields' loads then have a whole iteration to land, and the hand-off is never what a B-wave waits for
    int jy = j, ty = t, sy = 0;
    auto y_prev = [&]() { if (--ty < 0) { ty = Lw - 1; --jy; } };
    YIn yin;
    if (owner) {
        y_fetch(yin, jy, ty);
        publish_dy(yin, jy, ty, sy++);
        if (x.steps > 1) {
            y_prev();
            y_fetch(yin, jy, ty);
            publish_dy(yin, jy, ty, sy++);
        }
        if (x.steps > 2) {
            y_prev();
            y_fetch(yin, jy, ty);                                  // step 2's fields, published at the end of iteration 0
        }
    }
    for (int s = 0; s < x.steps; ++s) {
        const uint32_t dof = sd_lane + (uint32_t)(((int64_t)j * Lw + t) * 1024);
        int jn = j, tn = t - 1;
        if (tn < 0) { tn = Lw - 1; --jn; }
        if (s + 1 >= x.steps) { jn = j; tn = t; }
        d2_wait(x.gcnt, 4 * (s + 1));
        f32x4 duk[B2_TILES];
#pragma unroll
        for (int k = 0; k < B2_TILES; ++k) {
            duk[k] = splat4(0.f);
            if (!FAST && !c[k].live) continue;
            for (int ch = 0; ch < 6; ++ch) duk[k] = b2_chunk(W[ch], b2_g_read(x, k, ch), duk[k]);
        }
        d2_signal(x.rcnt);
#pragma unroll
        for (int k = 0; k < B2_TILES; ++k) {
            if (!FAST && !c[k].live) continue;
            const bool valid = FAST || c[k].valid;
            const f32x4 du = duk[k];
            const f32x4 uu = FAST ? u[k] : zero_unless(valid, u[k]);
            f32x4 dup;
            for (int i = 0; i < 4; ++i) dup[i] = uu[i] > 0.f ? du[i] : 0.f;
            if (thin) {                                            // d linear.weight[own units, :] += dup^T [x || latent]  (dup is zero for absent chains)
                gWlin[k] = b2_outer_acc(x, dup, xin[k], gWlin[k]);
                gblin += dup;
            } else {
                st4<FAST>(ddb[k], dof + x.cgs * REC_CG(DD_DU + 16 * q), valid, dup);
            }
            dlat[k] = mma_block(wlatT, dup, dlat[k]);
            const uint32_t sn = sd_lane + (uint32_t)(((int64_t)jn * Lw + tn) * 1024);
            u[k] = ld4_raw<FAST>(sdb[k], sn + x.cgs * REC_CG(SD_U + 16 * q), valid);
            xin[k] = ld4_raw<FAST>(sdb[k], sn + x.cgs * REC_CG(SD_X), valid);
        }
        if (owner && s + 2 < x.steps) {
            // dy of step s + 2 into the slot of step s - 2 (every B-wave passed that step long ago); then the fields of step s + 3
            publish_dy(yin, jy, ty, sy++);
            if (s + 3 < x.steps) {
                y_prev();
                y_fetch(yin, jy, ty);
            }
        }
        if (t == 0) {
            // ---- window j done: d loss / d latent_j = the four quarters' shares (double buffered by window parity; a slot
            // is free once the owners have read window w - 2)
            if (w >= 2) d2_wait(x.ocnt, x.n_live * (w - 1));
            float* lx = x.s_lx + (w & 1) * (B2_TILES * 4 * 256);
#pragma unroll
            for (int k = 0; k < B2_TILES; ++k) {
                if (!FAST && !c[k].live) continue;
                *reinterpret_cast<f32x4*>(lx + (k * 4 + q) * 256 + 4 * l) = dlat[k];
                dlat[k] = splat4(0.f);
            }
            d2_signal(x.lcnt);
            if (owner) {
                d2_wait(x.lcnt, 4 * (w + 1));
                const float* p = lx + ko * 4 * 256 + 4 * l;
                const f32x4 sum = (*reinterpret_cast<const f32x4*>(p) + *reinterpret_cast<const f32x4*>(p + 256)) +
                                  (*reinterpret_cast<const f32x4*>(p + 512) + *reinterpret_cast<const f32x4*>(p + 768));
                d2_signal(x.ocnt);
                st4<FAST>(dl_base, dl_lane + (uint32_t)j * (uint32_t)(DSL * 4), ovalid, sum);
            }
            ++w;
        }
        if (--t < 0) { t = Lw - 1; --j; }
    }
    if (thin) {
        float* part = a.dec_thin_part + ((int64_t)x.net * gridDim.x + blockIdx.x) * IPLAN_BEH_DEC_THIN_PART;
        const bool add = x.j_hi < J;
        b2_store_thin(part + TP_WLIN, add, (gWlin[0] + gWlin[1]) + gWlin[B2_TILES - 1], 16 * q, 16, 0, x.n, g);    // [unit 16q + 4g + r][input n]
        for (int r = 0; r < 4; ++r) {
            const float sum = chain_sum_b(gblin[r]);
            if (x.n == 0) part[TP_BLIN + 16 * q + 4 * g + r] = add ? part[TP_BLIN + 16 * q + 4 * g + r] + sum : sum;
        }
    }
}

// decoder gradient arena <- sum over the workgroups' thin partials, in workgroup order.  grid: (ceil(P / 256), n_nets)
__global__ __launch_bounds__(256) void beh_dec_thin_grad_kernel(IplanBehArgs a, int n_wg) {
    const int net = (int)blockIdx.y, din = a.d + a.Z;
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int sizes[4] = {a.d * DHd, a.d, DHd * din, DHd};
    const int which_of[4] = {IPLAN_DEC_OUT_W, IPLAN_DEC_OUT_B, IPLAN_DEC_LIN_W, IPLAN_DEC_LIN_B};
    int rem = idx, k = -1;
    for (int i = 0; i < 4; ++i) {
        if (rem < sizes[i]) { k = i; break; }
        rem -= sizes[i];
    }
    if (k < 0) return;
    int src;
    switch (k) {
        case 0: src = TP_WOUT + rem; break;                                  // [o][64] rows o < d
        case 1: src = TP_BOUT + rem; break;
        case 2: src = TP_WLIN + (rem / din) * 16 + rem % din; break;        // [unit][16] columns < d + Z
        default: src = TP_BLIN + rem; break;
    }
    const float* part = a.dec_thin_part + (int64_t)net * n_wg * IPLAN_BEH_DEC_THIN_PART + src;
    float s = 0.f;
    for (int w = 0; w < n_wg; ++w) s += part[(int64_t)w * IPLAN_BEH_DEC_THIN_PART];
    float* dst = a.dec_grad + (int64_t)net * a.dec_grad_s_net + a.dec_off[which_of[k]] + rem;
    *dst = a.dec_grad_beta != 0.f ? fmaf(a.dec_grad_beta, *dst, s) : s;
}

__global__ __launch_bounds__(B2_THREADS, 2) void beh_dec_bwd2_kernel(IplanBehArgs a) {
    IPLAN_DYN_LDS(smem);
    B2Ctx x;
    x.s_gx = smem;
    x.s_lx = x.s_gx + B2_GX;
    x.s_dy = x.s_lx + B2_LX;
    x.s_turn = x.s_dy + B2_DY + uniform_i(wave_id()) * 512;
    int* s_cnt = reinterpret_cast<int*>(x.s_dy + B2_DY + B2_TURN);
    x.s_scale = reinterpret_cast<float*>(s_cnt + B2_CNT_INTS);
    if (threadIdx.x < B2_CNT_INTS) s_cnt[threadIdx.x] = 0;
    const int w = uniform_i(wave_id()), role = w >> 2;                                // role 0: recurrent wave B_q, 1: input-side wave A_q
    x.q = w & 3;
    x.l = lane_id(); x.n = x.l & 15; x.g = x.l >> 4; x.net = (int)blockIdx.y;
    x.PD = a.dec_params + (int64_t)x.net * a.dec_s_net;
    x.Lw = a.L;
    DecTile c[B2_TILES];
#pragma unroll
    for (int k = 0; k < B2_TILES; ++k) dec_tile(a, c[k], (int)blockIdx.x * B2_TILES + k);
    x.J = c[0].J;
    x.j_lo = imax(a.bwd_j_lo, 0);
    x.j_hi = a.bwd_j_hi > 0 ? imin(a.bwd_j_hi, x.J) : x.J;
    x.steps = (x.j_hi - x.j_lo) * x.Lw;
    x.steps_per_chain = (int64_t)x.J * x.Lw;
    x.cgs = (uint32_t)(x.steps_per_chain * 1024);
    x.gcnt = s_cnt + 0; x.rcnt = s_cnt + 1; x.lcnt = s_cnt + 2; x.ocnt = s_cnt + 3; x.dcnt = s_cnt + 4;
    x.n_live = 0;
    bool fast = true;
#pragma unroll
    for (int k = 0; k < B2_TILES; ++k) { x.n_live += c[k].live ? 1 : 0; fast = fast && c[k].full; }
    // the loss scale of every window of this launch: d N / (sum of the window's mask + eps) / J  (wave w: windows w, w + 8, ...)
    for (int jj = x.j_lo + w; jj < x.j_hi; jj += B2_THREADS / 64) {
        const float sc = (float)(a.d * a.N) / (window_mask_sum(a, x.net, jj) + BEPS) / (a.hard ? 1.0f : (float)x.J);
        if (x.l == 0) x.s_scale[jj - x.j_lo] = sc;
    }
    __syncthreads();
    const bool fastu = uniform_i(fast ? 1 : 0) != 0;
    if (role == 0) {
        if (fastu) b2_recurrent<true>(a, x, c); else b2_recurrent<false>(a, x, c);
    } else {
        if (fastu) b2_input<true>(a, x, c); else b2_input<false>(a, x, c);
    }
}

static int check_beh(const IplanBehArgs* a, const char* what) {
    if (!a) return fail(IPLAN_EINVAL, "%s: null args", what);
    if (a->n_nets < 1 || a->E < 1 || a->N < 1 || a->L < 1 || (a->hard ? a->T / a->L - 1 : a->T - 1 - a->L) < 1 || a->d < 1 || a->Z < 1 ||
        a->d > 16 || a->Z > 16 || a->d + a->Z > 16)
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
    if ((a->fwd_j_lo > 0 || a->fwd_j_hi > 0) && (a->win || !a->enc_carry || !a->dec_carry || a->fwd_j_lo < 0 ||
                                                 (a->fwd_j_hi > 0 && a->fwd_j_hi <= a->fwd_j_lo)))
        return fail(IPLAN_EINVAL, "iplan_beh_fwd: a window range needs enc_carry, dec_carry and 0 <= fwd_j_lo < fwd_j_hi");
    const int tiles = (a->E * a->N + 15) / 16;
    const dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)a->n_nets);                       // encoder: 4 tiles per workgroup
    const dim3 dgrid((unsigned)((tiles + DEC_TILES - 1) / DEC_TILES), (unsigned)a->n_nets);  // decoder: 3 tiles x 4 quarter-waves
    const int ph = a->win ? 2 : a->fwd_phase;
    if (ph == 0 || ph == 1) {
        if (getenv("IPLAN_ENC_FP32") == nullptr) hipLaunchKernelGGL(beh_enc_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL(beh_enc_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    }
    // The second form (split-bf16, register-resident weights, input projection ahead of the recurrence) is the DEFAULT where it
    // applies (not in single-window mode, d <= 8, at most D2_MAX_WINDOWS windows); IPLAN_DEC_FWD_V1=1 selects the first form.
    // History: on the chain-major records of rounds 1-2 both forms ended up behind the record stores (6.0 vs 5.9 ms per pass: the
    // second form computes in 3.5 ms, the 13.8 GB of records took 3.1 - 4.1 ms as 64-byte pieces, and a wave that waits for room
    // in the store path issues nothing else, so the two added up; profiles/r03b_notes.md).  With the records column-grouped
    // (1 KiB blocks, 2.5 ms by the same probe) the second form runs 4.8 ms against the first form's 6.0 ms, behaviour learn
    // 20.2 -> 18.6 ms, cycle 311-314 -> 304 ms on one box (profiles/r03c_notes.md, call r3ac).
    const bool v2 = !a->win && a->d <= 8 && (a->hard ? a->T / a->L - 1 : a->T - 1 - a->L) <= D2_MAX_WINDOWS && getenv("IPLAN_DEC_FWD_V1") == nullptr;
    if ((ph == 0 || ph == 2) && v2) {
        const dim3 grid2((unsigned)((tiles + D2_TILES - 1) / D2_TILES), (unsigned)a->n_nets);
        const size_t lds = sizeof(float) * D2_LDS_FLOATS;
#ifndef IPLAN_HOST_EMULATION
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(beh_dec_fwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
        hipLaunchKernelGGL(beh_dec_fwd2_kernel, grid2, dim3(D2_THREADS), lds, (hipStream_t)stream, *a);
    } else if (ph == 0 || ph == 2) {
        const size_t lds = sizeof(float) * (2 * 3 * DHd * DLD + DHd * 24 + 16 * DLD + DEC_FWD_BIAS + DEC_TILES * XF_SLOTS * 256 + 16);
#ifndef IPLAN_HOST_EMULATION
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(beh_dec_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
        hipLaunchKernelGGL(beh_dec_fwd_kernel, dgrid, dim3(DEC_THREADS), lds, (hipStream_t)stream, *a);
    }
    if (!a->win && (ph == 0 || ph == 3)) hipLaunchKernelGGL(beh_loss_kernel, dim3((unsigned)a->n_nets), dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_beh_fwd");
}

#if defined(D2_CLOCKS) && !defined(IPLAN_HOST_EMULATION)
extern "C" int iplan_debug_d2_clocks(long long* out) {     // diagnostic builds only: 16 accumulated segment clocks of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(iplan::g_d2_clk), 16 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int iplan_beh_bwd(const IplanBehArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_beh(a, "iplan_beh_bwd")) return rc;
    if (a->win) return fail(IPLAN_EINVAL, "iplan_beh_bwd: not available in single-window decoder mode");
    if (!a->dsave_dec || !a->dsave_lat || !a->enc_part || !a->enc_grad)
        return fail(IPLAN_EINVAL, "iplan_beh_bwd: dsave_dec / dsave_lat / enc_part / enc_grad missing");
    if ((a->bwd_j_lo > 0 || a->bwd_j_hi > 0) &&
        (a->bwd_phase == 0 || (a->bwd_phase == 1 && !a->dec_carry) || (a->bwd_phase == 2 && !a->enc_carry) || a->bwd_j_lo < 0 ||
         (a->bwd_j_hi > 0 && a->bwd_j_hi <= a->bwd_j_lo)))
        return fail(IPLAN_EINVAL, "iplan_beh_bwd: a window range needs bwd_phase 1 (+ dec_carry) or 2 (+ enc_carry) and 0 <= bwd_j_lo < bwd_j_hi");
    const int tiles = (a->E * a->N + 15) / 16;
    const dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)a->n_nets);
    const dim3 dgrid((unsigned)((tiles + DEC_TILES - 1) / DEC_TILES), (unsigned)a->n_nets);
    const size_t lds = sizeof(float) * (2 * DHd * (3 * DHd + 8) + DHd * 24 + 16 * DLD + DEC_TILES * XB_SLOTS * 256 + 16);
#ifndef IPLAN_HOST_EMULATION
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(beh_dec_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
    // The second form (split-bf16 backward-data products, register-resident weight pieces, role-split waves) is the DEFAULT
    // where it applies (at most D2_MAX_WINDOWS windows per launch); IPLAN_DEC_BWD_V1=1 selects the first form.
    const int jn_hi = a->bwd_j_hi > 0 ? a->bwd_j_hi : (a->hard ? a->T / a->L - 1 : a->T - 1 - a->L);
    const bool v2 = jn_hi - (a->bwd_j_lo > 0 ? a->bwd_j_lo : 0) <= D2_MAX_WINDOWS && getenv("IPLAN_DEC_BWD_V1") == nullptr;
    if (a->bwd_phase != 2 && v2) {
        const dim3 grid2((unsigned)((tiles + B2_TILES - 1) / B2_TILES), (unsigned)a->n_nets);
        const size_t lds2 = sizeof(float) * B2_LDS_FLOATS;
#ifndef IPLAN_HOST_EMULATION
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(beh_dec_bwd2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
#endif
        hipLaunchKernelGGL(beh_dec_bwd2_kernel, grid2, dim3(B2_THREADS), lds2, (hipStream_t)stream, *a);
        if (a->dec_thin_part && a->bwd_j_lo <= 0) {          // the last (or only) piece: reduce the thin weight-gradient partials
            if (!a->dec_grad) return fail(IPLAN_EINVAL, "iplan_beh_bwd: dec_thin_part needs dec_grad");
            const int p_thin = a->d * DHd + a->d + DHd * (a->d + a->Z) + DHd;
            hipLaunchKernelGGL(beh_dec_thin_grad_kernel, dim3((unsigned)((p_thin + 255) / 256), (unsigned)a->n_nets), dim3(256), 0,
                               (hipStream_t)stream, *a, (int)grid2.x);
        }
    } else if (a->bwd_phase != 2) {
        if (a->dec_thin_part) return fail(IPLAN_EINVAL, "iplan_beh_bwd: dec_thin_part is only supported by the decoder BPTT's second form");
        hipLaunchKernelGGL(beh_dec_bwd_kernel, dgrid, dim3(DEC_THREADS), lds, (hipStream_t)stream, *a);
    }
    if (a->bwd_phase != 1) {
        if (getenv("IPLAN_ENC_FP32") == nullptr) hipLaunchKernelGGL(beh_enc_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL(beh_enc_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, *a);
        if (a->bwd_j_lo <= 0) {                             // the last (or only) piece: reduce the wave partials
            const int p_enc = 32 * a->d + 32 + 2 * 96 * 32 + 2 * 96 + a->Z * 32 + a->Z;
            hipLaunchKernelGGL(beh_enc_grad_kernel, dim3((unsigned)((p_enc + 255) / 256), (unsigned)a->n_nets), dim3(256), 0,
                               (hipStream_t)stream, *a);
        }
    }
    return check_launch("iplan_beh_bwd");
}
