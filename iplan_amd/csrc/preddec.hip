// Trajectory-prediction decoder (Prediction_Decoder / DecoderRNN, nova/prediction_net.py:6-63) with
// the masked-L1 loss of Prediction_policy.learn (nova/prediction_policy.py:223-225), forward and
// backward.  One wave = 16 (sample, entity) rows of one agent-net; the whole autoregressive chain
//    y_p, h = Linear(Dropout(tanh(GRU(ReLU(Linear(y_{p-1})), h)))),  h_0 = GAT output,  p < pred_length
// runs in registers in the D layout (wave_tile.h); the forward streams the per-step activations the
// backward needs, the backward streams the row-level pre-activation gradients that wgrad.hip
// contracts into weight gradients, and hands d(loss)/d(h_0) to the GAT backward.
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int PH = 32;            // attention_dim == decoder hidden
constexpr int PLD = PH + 4;
constexpr int PSV = IPLAN_PDEC_SAVE;
constexpr int PDS = IPLAN_PDEC_DSAVE;
constexpr float PEPS = 1e-10f;    // EPS of nova/prediction_policy.py:12

__device__ __forceinline__ float chain_sum_p(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

__global__ __launch_bounds__(256) void pdec_fwd_kernel(IplanPdecArgs a) {
    __shared__ __attribute__((aligned(16))) float s_lin[PH * 20];
    __shared__ __attribute__((aligned(16))) float s_wih[3 * PH * PLD];
    __shared__ __attribute__((aligned(16))) float s_whh[3 * PH * PLD];
    __shared__ __attribute__((aligned(16))) float s_out[16 * PLD];
    __shared__ __attribute__((aligned(16))) float s_blin[PH], s_bih[3 * PH], s_bhh[3 * PH], s_bout[16];

    const int net = (int)blockIdx.y;
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    stage_matrix(s_lin, 20, PH, P + a.off[IPLAN_DEC_LIN_W], PH, a.d);
    stage_matrix(s_wih, PLD, 3 * PH, P + a.off[IPLAN_DEC_WIH], 3 * PH, PH);
    stage_matrix(s_whh, PLD, 3 * PH, P + a.off[IPLAN_DEC_WHH], 3 * PH, PH);
    stage_matrix(s_out, PLD, 16, P + a.off[IPLAN_DEC_OUT_W], a.d, PH);
    stage_vector(s_blin, PH, P + a.off[IPLAN_DEC_LIN_B], PH);
    stage_vector(s_bih, 3 * PH, P + a.off[IPLAN_DEC_BIH], 3 * PH);
    stage_vector(s_bhh, 3 * PH, P + a.off[IPLAN_DEC_BHH], 3 * PH);
    stage_vector(s_bout, 16, P + a.off[IPLAN_DEC_OUT_B], a.d);
    __syncthreads();

    const int l = lane_id(), n = l & 15, g = l >> 4;
    const int tile = (int)blockIdx.x * 4 + wave_id();
    const int tiles = (a.rows + 15) / 16;
    if (tile >= tiles) return;
    const int row = tile * 16 + n;
    const bool valid = row < a.rows;
    const int64_t gr = (int64_t)net * a.rows + (valid ? row : 0);
    const float m = valid ? a.mask[(int64_t)net * (a.rows / a.N) + row / a.N] : 0.f;
    const float inv_keep = 1.0f / (1.0f - a.drop_p);
    f32x4 x[1], h[2];
    x[0] = vload(a.x0 + gr * a.d, valid, a.d, 0);
    h[0] = vload(a.h0 + gr * PH, valid, PH, 0);
    h[1] = vload(a.h0 + gr * PH, valid, PH, 1);
    float err = 0.f;
    for (int p = 0; p < a.P; ++p) {
        float* sv = a.saved + (gr * a.P + p) * PSV;
        vstore(sv, valid, 16, 0, x[0]);
        f32x4 u[2];
        u[0] = relu4(dense_tile<1>(s_lin, 20, 0, x, bfrag_lds(s_blin, 0)));
        u[1] = relu4(dense_tile<1>(s_lin, 20, 16, x, bfrag_lds(s_blin, 1)));
        GruGates keep[2];
        gru_step_lds<2, 2>(s_wih, PLD, s_whh, PLD, s_bih, s_bhh, u, h, keep);
        f32x4 act[2];
        for (int T = 0; T < 2; ++T) {
            vstore(sv + 16, valid, PH, T, u[T]);
            vstore(sv + 48, valid, PH, T, keep[T].r);
            vstore(sv + 80, valid, PH, T, keep[T].z);
            vstore(sv + 112, valid, PH, T, keep[T].n);
            vstore(sv + 144, valid, PH, T, keep[T].hn);
            vstore(sv + 176, valid, PH, T, h[T]);
            f32x4 km = splat4(1.0f);
            if (a.keep) {
                km = vload(a.keep + (((int64_t)net * a.P + p) * a.rows + (valid ? row : 0)) * PH, valid, PH, T);
                for (int q = 0; q < 4; ++q) km[q] *= inv_keep;
            }
            for (int q = 0; q < 4; ++q) act[T][q] = tanh_f(h[T][q]) * km[q];
            vstore(sv + 208, valid, PH, T, act[T]);
        }
        const f32x4 y = dense_tile<2>(s_out, PLD, 0, act, bfrag_lds(s_bout, 0));
        vstore(sv + 240, valid, 16, 0, y);
        const float* trow = a.target + (gr * a.P + p) * a.d;
        float* prow = a.pred + (gr * a.P + p) * a.d;
        const f32x4 tg = vload(trow, valid, a.d, 0);
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * g + q;
            if (valid && c < a.d) {
                prow[c] = y[q];
                err += fabsf(tg[q] - y[q]) * m;
            }
        }
        x[0] = (a.teacher && a.teacher[net * a.P + p]) ? tg : y;
    }
    err = chain_sum_p(group_sum(err));
    if (l == 0) a.loss_part[(int64_t)net * tiles + tile] = err;
}

// loss[net] = sum(loss_part) / (sum(mask) * N * P * d + EPS) * d * P
__global__ __launch_bounds__(64) void pdec_loss_kernel(IplanPdecArgs a) {
    const int net = (int)blockIdx.x;
    const int S = a.rows / a.N, tiles = (a.rows + 15) / 16;
    float ms = 0.f, es = 0.f;
    for (int i = lane_id(); i < S; i += 64) ms += a.mask[(int64_t)net * S + i];
    for (int i = lane_id(); i < tiles; i += 64) es += a.loss_part[(int64_t)net * tiles + i];
    ms = a.mask_sum ? a.mask_sum[net] : wave_sum(ms);
    es = wave_sum(es);
    if (lane_id() == 0) a.loss[net] = es / (ms * (float)(a.N * a.P * a.d) + PEPS) * (float)(a.d * a.P);
}

__global__ __launch_bounds__(256) void pdec_bwd_kernel(IplanPdecArgs a) {
    const int net = (int)blockIdx.y;
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    const int l = lane_id(), n = l & 15, g = l >> 4;
    const int tile = (int)blockIdx.x * 4 + wave_id();
    const int tiles = (a.rows + 15) / 16;
    if (tile >= tiles) return;
    const int row = tile * 16 + n;
    const bool valid = row < a.rows;
    const int64_t gr = (int64_t)net * a.rows + (valid ? row : 0);
    const int S = a.rows / a.N;
    float ms = 0.f;
    for (int i = l; i < S; i += 64) ms += a.mask[(int64_t)net * S + i];
    ms = a.mask_sum ? a.mask_sum[net] : wave_sum(ms);
    const float scale = (float)(a.d * a.P) / (ms * (float)(a.N * a.P * a.d) + PEPS);
    const float m = valid ? a.mask[(int64_t)net * S + row / a.N] : 0.f;
    const float inv_keep = 1.0f / (1.0f - a.drop_p);

    const float* Wih = P + a.off[IPLAN_DEC_WIH];
    const float* Whh = P + a.off[IPLAN_DEC_WHH];
    const float* Wout = P + a.off[IPLAN_DEC_OUT_W];    // [d][32]
    const float* Wlin = P + a.off[IPLAN_DEC_LIN_W];    // [32][d]
    f32x4 wihT[2][6], whhT[2][6];
    for (int T = 0; T < 2; ++T)
        for (int t = 0; t < 6; ++t) {
            wihT[T][t] = wfrag_t(Wih, PH, 3 * PH, PH, 16 * T, 16 * t);
            whhT[T][t] = wfrag_t(Whh, PH, 3 * PH, PH, 16 * T, 16 * t);
        }
    f32x4 dh[2], dxn;
    dh[0] = splat4(0.f); dh[1] = splat4(0.f); dxn = splat4(0.f);
    for (int p = a.P - 1; p >= 0; --p) {
        const float* sv = a.saved + (gr * a.P + p) * PSV;
        float* ds = a.dsave + (gr * a.P + p) * PDS;
        const f32x4 y = vload(sv + 240, valid, 16, 0);
        const f32x4 tg = vload(a.target + (gr * a.P + p) * a.d, valid, a.d, 0);
        const bool feeds_next = (p < a.P - 1) && !(a.teacher && a.teacher[net * a.P + p]);
        f32x4 dy[1];
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * g + q;
            float v = 0.f;
            if (valid && c < a.d) {
                const float e = tg[q] - y[q];
                v = -((e > 0.f) ? 1.0f : (e < 0.f ? -1.0f : 0.0f)) * m * scale;      // d|e|/dy, sign(0) = 0
                if (feeds_next) v += dxn[q];
            }
            dy[0][q] = v;
        }
        vstore(ds, valid, 16, 0, dy[0]);
        const float* hprow = p > 0 ? (sv - PSV + 176) : (a.h0 + gr * PH);
        f32x4 dgi[6], dgh[6], dhd[2];
        for (int T = 0; T < 2; ++T) {
            const f32x4 da = dense_tile_gt<1>(Wout, PH, a.d, PH, 16 * T, dy, splat4(0.f));
            const f32x4 hs = vload(sv + 176, valid, PH, T);
            f32x4 km = splat4(1.0f);
            if (a.keep) {
                km = vload(a.keep + (((int64_t)net * a.P + p) * a.rows + (valid ? row : 0)) * PH, valid, PH, T);
                for (int q = 0; q < 4; ++q) km[q] *= inv_keep;
            }
            f32x4 dht;
            for (int q = 0; q < 4; ++q) {
                const float th = tanh_f(hs[q]);
                dht[q] = fmaf(da[q] * km[q], 1.0f - th * th, dh[T][q]);
            }
            const GruGrads o = gru_gates_bwd(dht, vload(sv + 48, valid, PH, T), vload(sv + 80, valid, PH, T),
                                             vload(sv + 112, valid, PH, T), vload(sv + 144, valid, PH, T),
                                             vload(hprow, valid, PH, T));
            vstore(ds + 48, valid, PH, T, o.dr);
            vstore(ds + 80, valid, PH, T, o.dz);
            vstore(ds + 112, valid, PH, T, o.dni);
            vstore(ds + 144, valid, PH, T, o.dnh);
            dgi[T] = o.dr; dgi[2 + T] = o.dz; dgi[4 + T] = o.dni;
            dgh[T] = o.dr; dgh[2 + T] = o.dz; dgh[4 + T] = o.dnh;
            dhd[T] = o.dh_direct;
        }
        f32x4 du[2];
        for (int T = 0; T < 2; ++T) {
            f32x4 acc = splat4(0.f), acch = dhd[T];
            for (int t = 0; t < 6; ++t) {
                acc = mma_block(wihT[T][t], dgi[t], acc);
                acch = mma_block(whhT[T][t], dgh[t], acch);
            }
            const f32x4 u = vload(sv + 16, valid, PH, T);
            for (int q = 0; q < 4; ++q) du[T][q] = u[q] > 0.f ? acc[q] : 0.f;
            vstore(ds + 16, valid, PH, T, du[T]);
            dh[T] = acch;
        }
        dxn = dense_tile_gt<2>(Wlin, a.d, PH, a.d, 0, du, splat4(0.f));
    }
    vstore(a.g_h0 + gr * PH, valid, PH, 0, dh[0]);
    vstore(a.g_h0 + gr * PH, valid, PH, 1, dh[1]);
}

static int check_pdec(const IplanPdecArgs* a, const char* what) {
    if (!a) return fail(IPLAN_EINVAL, "%s: null args", what);
    if (a->n_nets < 1 || a->rows < 1 || a->P < 1 || a->d < 1 || a->d > 16 || a->N < 1 || a->rows % a->N)
        return fail(IPLAN_EINVAL, "%s: unsupported dims rows=%d N=%d P=%d d=%d", what, a->rows, a->N, a->P, a->d);
    if (!a->x0 || !a->h0 || !a->target || !a->mask || !a->params || !a->saved)
        return fail(IPLAN_EINVAL, "%s: null tensor pointer", what);
    return IPLAN_OK;
}

}  // namespace iplan

extern "C" int iplan_pdec_fwd(const IplanPdecArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_pdec(a, "iplan_pdec_fwd")) return rc;
    if (!a->pred || !a->loss_part || !a->loss) return fail(IPLAN_EINVAL, "iplan_pdec_fwd: pred / loss buffers missing");
    const int tiles = (a->rows + 15) / 16;
    hipLaunchKernelGGL(pdec_fwd_kernel, dim3((unsigned)((tiles + 3) / 4), (unsigned)a->n_nets), dim3(256), 0,
                       (hipStream_t)stream, *a);
    hipLaunchKernelGGL(pdec_loss_kernel, dim3((unsigned)a->n_nets), dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_pdec_fwd");
}

extern "C" int iplan_pdec_bwd(const IplanPdecArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_pdec(a, "iplan_pdec_bwd")) return rc;
    if (!a->dsave || !a->g_h0) return fail(IPLAN_EINVAL, "iplan_pdec_bwd: dsave / g_h0 missing");
    const int tiles = (a->rows + 15) / 16;
    hipLaunchKernelGGL(pdec_bwd_kernel, dim3((unsigned)((tiles + 3) / 4), (unsigned)a->n_nets), dim3(256), 0,
                       (hipStream_t)stream, *a);
    return check_launch("iplan_pdec_bwd");
}
