// LDS-resident weights + GRU / dense steps on wave tiles (see wave_tile.h for the layout).
#pragma once
#include "wave_tile.h"

namespace iplan {

// Cooperative copy of a row-major [rows x cols] global matrix into LDS as [rows_pad x ld]
// (zero padded; ld % 4 == 0, ld >= cols rounded up to 16).  All threads of the block call it.
__device__ __forceinline__ void stage_matrix(float* __restrict__ dst, int ld, int rows_pad,
                                             const float* __restrict__ src, int rows, int cols) {
    const int total = rows_pad * ld;
    for (int idx = (int)threadIdx.x; idx < total; idx += (int)blockDim.x) {
        const int r = idx / ld, c = idx - r * ld;
        dst[idx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.0f;
    }
}

// Transposed staging: dst[c][r] = src[r][c]  (dst is [cols_pad x ld], zero padded).
__device__ __forceinline__ void stage_matrix_t(float* __restrict__ dst, int ld, int cols_pad,
                                               const float* __restrict__ src, int rows, int cols) {
    const int total = cols_pad * ld;
    for (int idx = (int)threadIdx.x; idx < total; idx += (int)blockDim.x) {
        const int c = idx / ld, r = idx - c * ld;
        dst[idx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.0f;
    }
}

__device__ __forceinline__ void stage_vector(float* __restrict__ dst, int n_pad,
                                             const float* __restrict__ src, int n) {
    for (int idx = (int)threadIdx.x; idx < n_pad; idx += (int)blockDim.x) dst[idx] = idx < n ? src[idx] : 0.0f;
}

// A fragment from an LDS-staged matrix: lane (m,g) reads W[o0+m][k0+4g .. +3] (one ds_read_b128).
__device__ __forceinline__ f32x4 wfrag_lds(const float* __restrict__ sW, int ld, int o0, int k0) {
    const int l = lane_id();
    return *reinterpret_cast<const f32x4*>(sW + (o0 + (l & 15)) * ld + k0 + 4 * (l >> 4));
}

// Bias tile from LDS: lane (n,g) reads b[16t+4g .. +3].
__device__ __forceinline__ f32x4 bfrag_lds(const float* __restrict__ sb, int t) {
    return *reinterpret_cast<const f32x4*>(sb + 16 * t + 4 * (lane_id() >> 4));
}

// y[t'] (+)= W[16t'.., :] x  for one output tile, weights in LDS, KT input tiles.
template <int KT>
__device__ __forceinline__ f32x4 dense_tile(const float* __restrict__ sW, int ld, int o0,
                                            const f32x4 (&x)[KT], f32x4 acc) {
    for (int T = 0; T < KT; ++T) acc = mma_block(wfrag_lds(sW, ld, o0, 16 * T), x[T], acc);
    return acc;
}

// Same with the weights read straight from global memory (L1/L2 resident; row-major [rows x cols]).
template <int KT>
__device__ __forceinline__ f32x4 dense_tile_g(const float* __restrict__ W, int ld, int rows, int cols, int o0,
                                              const f32x4 (&x)[KT], f32x4 acc) {
    for (int T = 0; T < KT; ++T) acc = mma_block(wfrag(W, ld, rows, cols, o0, 16 * T), x[T], acc);
    return acc;
}

// Aligned fast paths (see wfrag_a / wfrag_ta): arena tensors are 16-byte aligned and the 32/64-wide layers have
// ld % 4 == 0 and whole 16-column tiles.
template <int KT>
__device__ __forceinline__ f32x4 dense_tile_ga(const float* __restrict__ W, int ld, int rows, int o0,
                                               const f32x4 (&x)[KT], f32x4 acc) {
    f32x4 wf[KT];
    for (int T = 0; T < KT; ++T) wf[T] = wfrag_a(W, ld, rows, o0, 16 * T);
    for (int T = 0; T < KT; ++T) acc = mma_block(wf[T], x[T], acc);
    return acc;
}
template <int KT>
__device__ __forceinline__ f32x4 dense_tile_gta(const float* __restrict__ W, int ld, int o0, const f32x4 (&x)[KT], f32x4 acc) {
    f32x4 wf[KT];
    for (int T = 0; T < KT; ++T) wf[T] = wfrag_ta(W, ld, o0, 16 * T);
    for (int T = 0; T < KT; ++T) acc = mma_block(wf[T], x[T], acc);
    return acc;
}

// acc += (W^T x)[o0 .. o0+15]: W row-major [rows x cols], x has `rows` entries in KT tiles
// (backward-data of a dense layer: dx = W^T dy).
template <int KT>
__device__ __forceinline__ f32x4 dense_tile_gt(const float* __restrict__ W, int ld, int rows, int cols, int o0,
                                               const f32x4 (&x)[KT], f32x4 acc) {
    for (int T = 0; T < KT; ++T) acc = mma_block(wfrag_t(W, ld, rows, cols, o0, 16 * T), x[T], acc);
    return acc;
}

// GRU step backward, lane-local (PyTorch gate equations, see gru_gates()).  Given dh = dL/dh_new and
// the saved gate activations, returns the pre-activation gradients and the direct path to h_prev.
struct GruGrads {
    f32x4 dr, dz, dni, dnh, dh_direct;   // dni = d/d(gi_n), dnh = d/d(gh_n) = dni * r
};
__device__ __forceinline__ GruGrads gru_gates_bwd(f32x4 dh, f32x4 r, f32x4 z, f32x4 n, f32x4 hn, f32x4 h_prev) {
    GruGrads o;
    for (int q = 0; q < 4; ++q) {
        const float dn = dh[q] * (1.0f - z[q]);
        const float dzg = dh[q] * (h_prev[q] - n[q]);
        o.dni[q] = dn * (1.0f - n[q] * n[q]);
        const float dr = o.dni[q] * hn[q];
        o.dnh[q] = o.dni[q] * r[q];
        o.dr[q] = dr * r[q] * (1.0f - r[q]);
        o.dz[q] = dzg * z[q] * (1.0f - z[q]);
        o.dh_direct[q] = dh[q] * z[q];
    }
    return o;
}

// Same as dense_tile with the input tiles taken from x[0..KT) against weight columns k0 + 16*T
// (lets the backward contract [dr | dz | dn_h] out of a [dr | dz | dn_i | dn_h] register array).
template <int KT>
__device__ __forceinline__ f32x4 dense_tile_k(const float* __restrict__ sW, int ld, int o0, int k0,
                                              const f32x4* __restrict__ x, f32x4 acc) {
    for (int T = 0; T < KT; ++T) acc = mma_block(wfrag_lds(sW, ld, o0, k0 + 16 * T), x[T], acc);
    return acc;
}

// NA output tiles that share one B operand (x, KT tiles), weights in LDS: acc[a] += W[o0[a].., k0..] x.
// The NA accumulator chains are issued round-robin (consecutive MFMAs never depend on each other: the
// 16x16x4 fp32 MFMA has 40 cycles of dependent latency against a 32-cycle issue) and the next k-tile's fragments
// are read from LDS while the current k-tile's MFMAs issue (LDS latency off the critical path).
template <int NA, int KT>
__device__ __forceinline__ void dense_multi(const float* __restrict__ sW, int ld, const int (&o0)[NA], int k0,
                                            const f32x4* __restrict__ x, f32x4 (&acc)[NA]) {
    f32x4 f[NA], fn[NA];
    for (int a = 0; a < NA; ++a) f[a] = wfrag_lds(sW, ld, o0[a], k0);
    for (int T = 0; T < KT; ++T) {
        if (T + 1 < KT)
            for (int a = 0; a < NA; ++a) fn[a] = wfrag_lds(sW, ld, o0[a], k0 + 16 * (T + 1));
        for (int q = 0; q < 4; ++q)
            for (int a = 0; a < NA; ++a) acc[a] = mfma4(f[a][q], x[T][q], acc[a]);
        if (T + 1 < KT)
            for (int a = 0; a < NA; ++a) f[a] = fn[a];
    }
}

// One GRU step with LDS-resident weights.  HT = H/16 hidden tiles, XT = input tiles.
// sWih: [3H x ldi], sWhh: [3H x ldh], sbih/sbhh: [3H].  h is updated in place; when `keep` is
// non-null the gate activations (r, z, n, hn) of every tile are returned for the backward pass.
template <int HT, int XT>
__device__ __forceinline__ void gru_step_lds(const float* __restrict__ sWih, int ldi,
                                             const float* __restrict__ sWhh, int ldh,
                                             const float* __restrict__ sbih, const float* __restrict__ sbhh,
                                             const f32x4 (&x)[XT], f32x4 (&h)[HT], GruGates* keep) {
    constexpr int H = 16 * HT;
    f32x4 hnew[HT];
    for (int t = 0; t < HT; ++t) {
        const int rows[3] = {16 * t, H + 16 * t, 2 * H + 16 * t};            // r, z, n gate rows of this hidden tile
        f32x4 ai[3], ah[3];
        ai[0] = bfrag_lds(sbih, t) + bfrag_lds(sbhh, t);
        ai[1] = bfrag_lds(sbih, HT + t) + bfrag_lds(sbhh, HT + t);
        ai[2] = bfrag_lds(sbih, 2 * HT + t);
        dense_multi<3, XT>(sWih, ldi, rows, 0, x, ai);                        // W_ih x: pre_r, pre_z, gi_n
        ah[0] = ai[0];
        ah[1] = ai[1];
        ah[2] = bfrag_lds(sbhh, 2 * HT + t);
        dense_multi<3, HT>(sWhh, ldh, rows, 0, h, ah);                        // + W_hh h: pre_r, pre_z, gh_n
        const GruGates o = gru_gates(ah[0], ah[1], ai[2], ah[2], h[t]);
        hnew[t] = o.h;
        if (keep) keep[t] = o;
        if (HT > 2) IPLAN_SCHED_FENCE();                  // 64-wide GRU: keep the next tile's fragment reads below this tile
    }
    for (int t = 0; t < HT; ++t) h[t] = hnew[t];
}

// LayerNorm over a per-chain vector of DT*16 real features (all tiles fully populated) held in
// D layout: statistics reduce over the 4 lane groups of the chain.  eps = 1e-5, biased variance.
template <int DT>
__device__ __forceinline__ void layer_norm_tiles(f32x4 (&v)[DT], const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* mean_out, float* rstd_out) {
    constexpr float inv = 1.0f / (16 * DT);
    float s = 0.f;
    for (int t = 0; t < DT; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    const float mu = group_sum(s) * inv;
    float q = 0.f;
    for (int t = 0; t < DT; ++t)
        for (int k = 0; k < 4; ++k) { const float d = v[t][k] - mu; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(group_sum(q) * inv + 1e-5f);
    for (int t = 0; t < DT; ++t) {
        const f32x4 gm = bfrag_a(gamma, t), bt = bfrag_a(beta, t);
        for (int k = 0; k < 4; ++k) v[t][k] = (v[t][k] - mu) * rstd * gm[k] + bt[k];
    }
    if (mean_out) *mean_out = mu;
    if (rstd_out) *rstd_out = rstd;
}

// same, with gamma / beta handed over as fragments (callers that hold the parameter vectors in LDS)
template <int DT>
__device__ __forceinline__ void layer_norm_tiles_f(f32x4 (&v)[DT], const f32x4 (&gm)[DT], const f32x4 (&bt)[DT], float* mean_out, float* rstd_out) {
    constexpr float inv = 1.0f / (16 * DT);
    float s = 0.f;
    for (int t = 0; t < DT; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
    const float mu = group_sum(s) * inv;
    float q = 0.f;
    for (int t = 0; t < DT; ++t)
        for (int k = 0; k < 4; ++k) { const float d = v[t][k] - mu; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(group_sum(q) * inv + 1e-5f);
    for (int t = 0; t < DT; ++t)
        for (int k = 0; k < 4; ++k) v[t][k] = (v[t][k] - mu) * rstd * gm[t][k] + bt[t][k];
    if (mean_out) *mean_out = mu;
    if (rstd_out) *rstd_out = rstd;
}

}  // namespace iplan
