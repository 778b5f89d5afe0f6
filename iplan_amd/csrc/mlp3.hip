// Three-layer perceptron  y = W3 tanh(W2 tanh(W1 x + b1) + b2) + b3  (optionally softmax(y)) for stacked nets, forward and
// backward-data: the FC behaviour ablation of iPLAN (nova/behavior_FC_net.py:6-37 Encoder_3FC / Decoder_3FC, used by
// nova/behavior_FC_policy.py).  Same wave-tile idiom as the recurrent kernels: a wave owns 16 rows, every vector lives
// in the MFMA D layout, the weight fragments come straight from L1/L2 (the nets are a few KB), tanh / softmax are
// lane-local.  Weight gradients are iplan_wgrad contractions over the saved activations and the row gradients
// this kernel emits.  Dims: K0, H, O <= 64, H % 16 == 0.
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

constexpr int MT = 4;          // up to 4 tiles (64) per vector

struct MlpDims { int kt, ht, ot; };
__device__ __forceinline__ MlpDims mlp_dims(const IplanMlp3Args& a) {
    MlpDims d;
    d.kt = (a.K0 + 15) / 16; d.ht = a.H / 16; d.ot = (a.O + 15) / 16;
    return d;
}

// y[t] = act(W[16t.., :] x + b) for the `nt` output tiles; W is [rows x cols] row-major
__device__ __forceinline__ void dense_g(const float* __restrict__ W, const float* __restrict__ b, int rows, int cols, int nt, int kt,
                                        const f32x4 (&x)[MT], f32x4 (&y)[MT]) {
    for (int t = 0; t < MT; ++t) {
        y[t] = splat4(0.f);
        if (t < nt) {
            f32x4 acc = bfrag(b, rows, t);
            for (int T = 0; T < MT; ++T)
                if (T < kt) acc = mma_block(wfrag(W, cols, rows, cols, 16 * t, 16 * T), x[T], acc);
            y[t] = acc;
        }
    }
}
// y[t] = (W^T x)[16t..] for the `nt` output tiles (W [rows x cols], x has `rows` entries in kt tiles)
__device__ __forceinline__ void dense_gt(const float* __restrict__ W, int rows, int cols, int nt, int kt, const f32x4 (&x)[MT],
                                         f32x4 (&y)[MT]) {
    for (int t = 0; t < MT; ++t) {
        y[t] = splat4(0.f);
        if (t < nt) {
            f32x4 acc = splat4(0.f);
            for (int T = 0; T < MT; ++T)
                if (T < kt) acc = mma_block(wfrag_t(W, cols, rows, cols, 16 * t, 16 * T), x[T], acc);
            y[t] = acc;
        }
    }
}

// grid: (ceil(rows / 64), n_nets); block 256 = 4 waves x 16 rows
__global__ __launch_bounds__(256) void mlp3_fwd_kernel(IplanMlp3Args a) {
    const int net = (int)blockIdx.y, l = lane_id(), n = l & 15, g = l >> 4;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave_id()) * 16 + n;
    const bool valid = row < a.rows;
    const int64_t gr = (int64_t)net * a.rows + (valid ? row : 0);
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    const MlpDims d = mlp_dims(a);
    f32x4 x[MT], y1[MT], y2[MT], y3[MT];
    for (int t = 0; t < MT; ++t) x[t] = t < d.kt ? vload(a.x + gr * a.K0, valid, a.K0, t) : splat4(0.f);
    dense_g(P + a.off[0], P + a.off[1], a.H, a.K0, d.ht, d.kt, x, y1);
    for (int t = 0; t < MT; ++t)
        for (int q = 0; q < 4; ++q) y1[t][q] = t < d.ht ? tanh_f(y1[t][q]) : 0.f;
    dense_g(P + a.off[2], P + a.off[3], a.H, a.H, d.ht, d.ht, y1, y2);
    for (int t = 0; t < MT; ++t)
        for (int q = 0; q < 4; ++q) y2[t][q] = t < d.ht ? tanh_f(y2[t][q]) : 0.f;
    dense_g(P + a.off[4], P + a.off[5], a.O, a.H, d.ot, d.ht, y2, y3);
    if (a.softmax) {                                            // over the O outputs of the row: lane-local + 4 lane groups
        float m = -INFINITY;
        for (int t = 0; t < MT; ++t)
            for (int q = 0; q < 4; ++q)
                if (16 * t + 4 * g + q < a.O) m = fmaxf(m, y3[t][q]);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float s = 0.f;
        for (int t = 0; t < MT; ++t)
            for (int q = 0; q < 4; ++q) {
                y3[t][q] = (16 * t + 4 * g + q < a.O) ? expf(y3[t][q] - m) : 0.f;
                s += y3[t][q];
            }
        s = group_sum(s);
        for (int t = 0; t < MT; ++t)
            for (int q = 0; q < 4; ++q) y3[t][q] /= s;
    }
    for (int t = 0; t < MT; ++t) {
        if (t < d.ht && a.saved) {
            vstore(a.saved + gr * 2 * a.H, valid, a.H, t, y1[t]);
            vstore(a.saved + gr * 2 * a.H + a.H, valid, a.H, t, y2[t]);
        }
        if (t < d.ot) vstore(a.out + gr * a.O, valid, a.O, t, y3[t]);
    }
    if (a.target) {                                             // sum |target - y| of this wave's rows (L1 loss numerator)
        float e = 0.f;
        for (int t = 0; t < MT; ++t)
            if (t < d.ot) {
                const f32x4 tg = vload(a.target + gr * a.O, valid, a.O, t);
                for (int q = 0; q < 4; ++q)
                    if (valid && 16 * t + 4 * g + q < a.O) e += fabsf(tg[q] - y3[t][q]);
            }
        e = wave_sum(e);
        if (l == 0) a.loss_part[(int64_t)net * gridDim.x * 4 + blockIdx.x * 4 + wave_id()] = e;
    }
}

// backward-data.  d(out) per row = g_out (if given) or -sign(target - out) * g_scale (the L1 loss); emits the row gradients
// dsave = [dy1pre (H) | dy2pre (H) | dy3 (O16)] for iplan_wgrad and, if asked, dx [rows, K0].
__global__ __launch_bounds__(256) void mlp3_bwd_kernel(IplanMlp3Args a) {
    const int net = (int)blockIdx.y, l = lane_id(), n = l & 15, g = l >> 4;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave_id()) * 16 + n;
    const bool valid = row < a.rows;
    const int64_t gr = (int64_t)net * a.rows + (valid ? row : 0);
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    const MlpDims d = mlp_dims(a);
    const int O16 = 16 * d.ot, DS = 2 * a.H + O16;
    f32x4 y1[MT], y2[MT], out[MT], dy[MT];
    for (int t = 0; t < MT; ++t) {
        y1[t] = t < d.ht ? vload(a.saved + gr * 2 * a.H, valid, a.H, t) : splat4(0.f);
        y2[t] = t < d.ht ? vload(a.saved + gr * 2 * a.H + a.H, valid, a.H, t) : splat4(0.f);
        out[t] = t < d.ot ? vload(a.out + gr * a.O, valid, a.O, t) : splat4(0.f);
        dy[t] = splat4(0.f);
        if (t < d.ot) {
            if (a.g_out) dy[t] = vload(a.g_out + gr * a.O, valid, a.O, t);
            else {
                const f32x4 tg = vload(a.target + gr * a.O, valid, a.O, t);
                for (int q = 0; q < 4; ++q) {
                    const float er = tg[q] - out[t][q];
                    dy[t][q] = (valid && 16 * t + 4 * g + q < a.O) ? -((er > 0.f) ? 1.0f : (er < 0.f ? -1.0f : 0.0f)) * a.g_scale : 0.f;
                }
            }
        }
    }
    if (a.softmax) {                                            // d logits = p o (d p - sum p d p)
        float s = 0.f;
        for (int t = 0; t < MT; ++t)
            for (int q = 0; q < 4; ++q) s = fmaf(out[t][q], dy[t][q], s);
        s = group_sum(s);
        for (int t = 0; t < MT; ++t)
            for (int q = 0; q < 4; ++q) dy[t][q] = out[t][q] * (dy[t][q] - s);
    }
    float* ds = a.dsave + gr * DS;
    f32x4 d2[MT], d1[MT], dx[MT];
    dense_gt(P + a.off[4], a.O, a.H, d.ht, d.ot, dy, d2);
    for (int t = 0; t < MT; ++t)
        for (int q = 0; q < 4; ++q) d2[t][q] *= 1.0f - y2[t][q] * y2[t][q];
    dense_gt(P + a.off[2], a.H, a.H, d.ht, d.ht, d2, d1);
    for (int t = 0; t < MT; ++t)
        for (int q = 0; q < 4; ++q) d1[t][q] *= 1.0f - y1[t][q] * y1[t][q];
    for (int t = 0; t < MT; ++t) {
        if (t < d.ht) {
            vstore(ds, valid, a.H, t, d1[t]);
            vstore(ds + a.H, valid, a.H, t, d2[t]);
        }
        if (t < d.ot) vstore(ds + 2 * a.H, valid, O16, t, dy[t]);
    }
    if (a.dx) {
        dense_gt(P + a.off[0], a.H, a.K0, d.kt, d.ht, d1, dx);
        for (int t = 0; t < MT; ++t)
            if (t < d.kt) vstore(a.dx + gr * a.K0, valid, a.K0, t, dx[t]);
    }
}

static int check_mlp(const IplanMlp3Args* a, const char* what, bool bwd) {
    if (!a) return fail(IPLAN_EINVAL, "%s: null args", what);
    if (a->n_nets < 1 || a->rows < 1 || a->K0 < 1 || a->K0 > 64 || a->H < 16 || a->H > 64 || (a->H & 15) || a->O < 1 || a->O > 64)
        return fail(IPLAN_EINVAL, "%s: unsupported dims rows=%lld K0=%d H=%d O=%d", what, (long long)a->rows, a->K0, a->H, a->O);
    if (!a->x || !a->params || !a->out) return fail(IPLAN_EINVAL, "%s: null tensor pointer", what);
    if (bwd && (!a->saved || !a->dsave || (!a->g_out && !a->target))) return fail(IPLAN_EINVAL, "%s: saved / dsave / g_out|target missing", what);
    if (!bwd && a->target && !a->loss_part) return fail(IPLAN_EINVAL, "%s: loss_part missing", what);
    return IPLAN_OK;
}

}  // namespace iplan

extern "C" int iplan_mlp3_fwd(const IplanMlp3Args* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_mlp(a, "iplan_mlp3_fwd", false)) return rc;
    hipLaunchKernelGGL(mlp3_fwd_kernel, dim3((unsigned)((a->rows + 63) / 64), (unsigned)a->n_nets), dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_mlp3_fwd");
}

extern "C" int iplan_mlp3_bwd(const IplanMlp3Args* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_mlp(a, "iplan_mlp3_bwd", true)) return rc;
    hipLaunchKernelGGL(mlp3_bwd_kernel, dim3((unsigned)((a->rows + 63) / 64), (unsigned)a->n_nets), dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_mlp3_bwd");
}
