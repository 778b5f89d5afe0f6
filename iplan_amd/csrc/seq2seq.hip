// nova/Seq2Seq.py:41-70 -- Seq2Seq.forward on the wave-tile GRU idiom: one wave owns 16 rows (N*V chains) for the whole
// call; the encoder GRU stack consumes the T_in input steps from a zero state, then pred_length autoregressive decoder
// steps feed the previous prediction (or the teacher's location) back in.  Hidden states of all layers stay in registers
// in the D layout; weights are read as MFMA A fragments straight from L1/L2 (the whole model is < 200 KB).
// The reference never calls this module on its training path: it is built for API completeness
// (same constructor / forward / state_dict).  Round 5: a saving form of the forward and the backward (seq2seq_bwd_kernel below), so
// that a caller who does train it gets gradients; neither is tuned -- one wave per 16 rows, weights from L1 / L2.
#include "api_util.h"
#include "gru_tile.h"

namespace iplan {

// one GRU layer step: x = `xt` input tiles of real width `in_dim`, weights [3H, in_dim] / [3H, H] in global memory
template <int HT>
__device__ __forceinline__ void gru_layer_step(const float* __restrict__ Wih, const float* __restrict__ Whh,
                                               const float* __restrict__ bih, const float* __restrict__ bhh, int in_dim,
                                               const f32x4 (&x)[4], int xt, f32x4 (&h)[HT], float* __restrict__ rec = nullptr,
                                               bool valid = false) {
    constexpr int H = 16 * HT;
    f32x4 hnew[HT];
    for (int t = 0; t < HT; ++t) {
        f32x4 pr = bfrag(bih, 3 * H, t) + bfrag(bhh, 3 * H, t);
        f32x4 pz = bfrag(bih, 3 * H, HT + t) + bfrag(bhh, 3 * H, HT + t);
        f32x4 gi = bfrag(bih, 3 * H, 2 * HT + t), gh = bfrag(bhh, 3 * H, 2 * HT + t);
        for (int T = 0; T < xt; ++T) {
            pr = mma_block(wfrag(Wih, in_dim, 3 * H, in_dim, 16 * t, 16 * T), x[T], pr);
            pz = mma_block(wfrag(Wih, in_dim, 3 * H, in_dim, H + 16 * t, 16 * T), x[T], pz);
            gi = mma_block(wfrag(Wih, in_dim, 3 * H, in_dim, 2 * H + 16 * t, 16 * T), x[T], gi);
        }
        for (int T = 0; T < HT; ++T) {
            pr = mma_block(wfrag_a(Whh, H, 3 * H, 16 * t, 16 * T), h[T], pr);
            pz = mma_block(wfrag_a(Whh, H, 3 * H, H + 16 * t, 16 * T), h[T], pz);
            gh = mma_block(wfrag_a(Whh, H, 3 * H, 2 * H + 16 * t, 16 * T), h[T], gh);
        }
        const GruGates o = gru_gates(pr, pz, gi, gh, h[t]);
        hnew[t] = o.h;
        if (rec) {                                           // h_prev | r | z | n | gh_n | h_new of this row (the backward's record)
            vstore(rec, valid, H, t, h[t]);
            vstore(rec + H, valid, H, t, o.r);
            vstore(rec + 2 * H, valid, H, t, o.z);
            vstore(rec + 3 * H, valid, H, t, o.n);
            vstore(rec + 4 * H, valid, H, t, o.hn);
            vstore(rec + 5 * H, valid, H, t, o.h);
        }
    }
    for (int t = 0; t < HT; ++t) h[t] = hnew[t];
}

template <int HT>
__global__ __launch_bounds__(256) void seq2seq_fwd_kernel(IplanSeq2SeqArgs a) {
    constexpr int H = 16 * HT;
    const int l = lane_id(), g = l >> 4;
    const int row = ((int)blockIdx.x * 4 + wave_id()) * 16 + (l & 15);
    const bool valid = row < a.rows;
    const float* __restrict__ P = a.params;
    f32x4 h[IPLAN_S2S_MAX_LAYERS][HT];
    for (int L = 0; L < IPLAN_S2S_MAX_LAYERS; ++L)
        for (int t = 0; t < HT; ++t) h[L][t] = splat4(0.f);
    const int xt_in = (a.In + 15) / 16;
    // ---- encoder: (N*V, T, C) -> hidden (L, N*V, H)
    for (int t = 0; t < a.T_in; ++t) {
        f32x4 x[4];
        for (int T = 0; T < 4; ++T) x[T] = T < xt_in ? vload(a.x + ((int64_t)row * a.T_in + t) * a.In, valid, a.In, T) : splat4(0.f);
#pragma unroll
        for (int L = 0; L < IPLAN_S2S_MAX_LAYERS; ++L) {                                  // (unrolled: h[L] is indexed statically)
            if (L >= a.layers) break;
            const int in_dim = L ? H : a.In;
            gru_layer_step<HT>(P + a.enc_off[4 * L], P + a.enc_off[4 * L + 1], P + a.enc_off[4 * L + 2], P + a.enc_off[4 * L + 3],
                               in_dim, x, L ? HT : xt_in, h[L],
                               a.save ? a.save + (((int64_t)t * a.layers + L) * a.rows + (valid ? row : 0)) * (6 * H) : nullptr, valid);
            for (int T = 0; T < 4; ++T) x[T] = T < HT ? h[L][T] : splat4(0.f);           // the layer above reads this layer's output
        }
    }
    // ---- decoder: pred_length autoregressive steps from last_location
    const float inv_keep = a.keep ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    f32x4 yin = vload(a.last + (int64_t)row * a.O, valid, a.O, 0);
    for (int t = 0; t < a.P; ++t) {
        f32x4 x[4];
        x[0] = yin;
        for (int T = 1; T < 4; ++T) x[T] = splat4(0.f);
#pragma unroll
        for (int L = 0; L < IPLAN_S2S_MAX_LAYERS; ++L) {
            if (L >= a.layers) break;
            const int in_dim = L ? H : a.O;
            gru_layer_step<HT>(P + a.dec_off[4 * L], P + a.dec_off[4 * L + 1], P + a.dec_off[4 * L + 2], P + a.dec_off[4 * L + 3],
                               in_dim, x, L ? HT : 1, h[L],
                               a.save ? a.save + (((int64_t)(a.T_in + t) * a.layers + L) * a.rows + (valid ? row : 0)) * (6 * H) : nullptr, valid);
            for (int T = 0; T < 4; ++T) x[T] = T < HT ? h[L][T] : splat4(0.f);
        }
        f32x4 act[HT];
        for (int T = 0; T < HT; ++T) {
            f32x4 km = splat4(1.0f);
            if (a.keep) km = vload(a.keep + ((int64_t)t * a.rows + row) * H, valid, H, T);
            for (int q = 0; q < 4; ++q) act[T][q] = tanh_f(x[T][q]) * (km[q] * inv_keep);   // x = the top layer's new hidden state
        }
        if (a.save) {                                        // the step's input and the output layer's input
            float* dr = a.save + (int64_t)(a.T_in + a.P) * a.layers * a.rows * (6 * H) + ((int64_t)t * a.rows + (valid ? row : 0)) * (16 + H);
            vstore(dr, valid, 16, 0, yin);
            for (int T = 0; T < HT; ++T) vstore(dr + 16, valid, H, T, act[T]);
        }
        const f32x4 y = dense_tile_g<HT>(P + a.lin_off[0], H, a.O, H, 0, act, bfrag(P + a.lin_off[1], a.O, 0));
        vstore(a.out + ((int64_t)row * a.P + t) * a.O, valid, a.O, 0, y);
        yin = y;
        if (a.teacher && a.coins && a.coins[t]) yin = vload(a.teacher + ((int64_t)row * a.P + t) * a.O, valid, a.O, 0);
        if (!valid) yin = splat4(0.f);
        for (int q = 0; q < 4; ++q)
            if (4 * g + q >= a.O) yin[q] = 0.f;
    }
    if (a.hidden_out) {
#pragma unroll
        for (int L = 0; L < IPLAN_S2S_MAX_LAYERS; ++L) {
            if (L >= a.layers) break;
            for (int T = 0; T < HT; ++T) vstore(a.hidden_out + ((int64_t)L * a.rows + row) * H, valid, H, T, h[L][T]);
        }
    }
}

// Backward: one wave per 16 rows walks the decoder steps and then the encoder steps in reverse.  dh[L] carries dLoss/d h_L across
// time (the decoder starts from the encoder's final state, so the same carry runs through both stacks); within a step the
// gradient goes from the output layer down the stack; a decoder step's input gradient is handed to the previous step's output
// when that output was what it was fed (no teacher forcing at that step).  Only data gradients are propagated here: every GRU
// step's [dr dz dn_i dn_h] and every d out row go to `dsave`, the weight gradients are contractions over them (iplan_wgrad).
// The carry lives in LDS, lane-private slots [wave][layer][tile][lane] (no barrier: a lane only ever touches its own): the layer
// loop is then a real loop over a.layers and one layer's temporaries (36 tiles) are all the registers hold.  (First form: dh of 4
// layers in registers under an unrolled layer loop -- 512 registers + 272 bytes of scratch per lane at H = 64, and wrong encoder
// gradients for layers 0 / 1 at 4 x 64 on the GPU while the host build of the same source was right.)
template <int HT>
__global__ __launch_bounds__(256) void seq2seq_bwd_kernel(IplanSeq2SeqBwdArgs b) {
    constexpr int H = 16 * HT;
    IPLAN_DYN_LDS(s_raw);                                    // [4 waves][layers][HT][64 lanes] f32x4
    const IplanSeq2SeqArgs& a = b.fwd;
    const int l = lane_id(), g = l >> 4;
    const int row = ((int)blockIdx.x * 4 + wave_id()) * 16 + (l & 15);
    const bool valid = row < a.rows;
    const int64_t vrow = valid ? row : 0;
    const float* __restrict__ P = a.params;
    const float inv_keep = a.keep ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const int n_steps = a.T_in + a.P, top = a.layers - 1;
    const float* __restrict__ keep = a.keep;
    float* __restrict__ ddec = b.dsave + (int64_t)n_steps * a.layers * a.rows * (4 * H);
    f32x4* __restrict__ s_dh = reinterpret_cast<f32x4*>(s_raw) + (int64_t)wave_id() * a.layers * HT * 64 + l;   // + (L * HT + t) * 64
    for (int i = 0; i < a.layers * HT; ++i) s_dh[i * 64] = splat4(0.f);
    f32x4 dx[HT];                                            // the input gradient of the layer processed last
    for (int t = 0; t < HT; ++t) dx[t] = splat4(0.f);
    for (int tau = n_steps - 1; tau >= 0; --tau) {
        const bool dec = tau >= a.T_in;
        const int t = tau - a.T_in;                          // decoder step
        f32x4 dyv[1] = {splat4(0.f)};
        if (dec) {
            // d out[t]: the loss's own gradient + the next step's input gradient (dx of its layer 0) where that input was this output
            f32x4 dy = vload(b.g_out + (vrow * a.P + t) * a.O, valid, a.O, 0);
            const bool fed_back = t + 1 < a.P && !(a.teacher && a.coins && a.coins[t]);
            if (fed_back) dy += dx[0];
            for (int q = 0; q < 4; ++q)
                if (4 * g + q >= a.O || !valid) dy[q] = 0.f;
            vstore(ddec + ((int64_t)t * a.rows + vrow) * 16, valid, 16, 0, dy);
            dyv[0] = dy;
        }
        for (int L = top; L >= 0; --L) {
            const float* rec = a.save + (((int64_t)tau * a.layers + L) * a.rows + vrow) * (6 * H);
            float* dsv = b.dsave + (((int64_t)tau * a.layers + L) * a.rows + vrow) * (4 * H);
            const int64_t* off = dec ? a.dec_off : a.enc_off;
            const float* Wih = P + off[4 * L];
            const float* Whh = P + off[4 * L + 1];
            const int in_dim = L ? H : (dec ? a.O : a.In);
            f32x4 dgi[3 * HT], dgh[3 * HT], direct[HT];
            for (int T = 0; T < HT; ++T) {
                f32x4 dhT = s_dh[(L * HT + T) * 64];         // through time
                if (L != top) dhT += dx[T];                  // the layer above's input is this layer's output
                else if (dec) {                              // output layer: y = W act + b, act = tanh(h_top) * keep / (1 - p)
                    const f32x4 dact = dense_tile_gt<1>(P + a.lin_off[0], H, a.O, H, 16 * T, dyv, splat4(0.f));
                    const f32x4 hn = vload(rec + 5 * H, valid, H, T);
                    f32x4 km = splat4(1.0f);
                    if (keep) km = vload(keep + ((int64_t)t * a.rows + vrow) * H, valid, H, T);
                    for (int q = 0; q < 4; ++q) {
                        const float th = tanh_f(hn[q]);
                        dhT[q] += dact[q] * (km[q] * inv_keep) * (1.0f - th * th);
                    }
                }
                const GruGrads o = gru_gates_bwd(dhT, vload(rec + H, valid, H, T), vload(rec + 2 * H, valid, H, T), vload(rec + 3 * H, valid, H, T),
                                                 vload(rec + 4 * H, valid, H, T), vload(rec, valid, H, T));
                dgi[T] = o.dr; dgi[HT + T] = o.dz; dgi[2 * HT + T] = o.dni;
                dgh[T] = o.dr; dgh[HT + T] = o.dz; dgh[2 * HT + T] = o.dnh;
                direct[T] = o.dh_direct;
                vstore(dsv, valid, H, T, o.dr);
                vstore(dsv + H, valid, H, T, o.dz);
                vstore(dsv + 2 * H, valid, H, T, o.dni);
                vstore(dsv + 3 * H, valid, H, T, o.dnh);
            }
            for (int T = 0; T < HT; ++T) s_dh[(L * HT + T) * 64] = dense_tile_gt<3 * HT>(Whh, H, 3 * H, H, 16 * T, dgh, direct[T]);   // -> d/d h_prev
            // the layer's input gradient: all H columns for an upper layer, the O <= 16 fed-back columns for the decoder's first, none
            // for the encoder's first (data)
            const int xt_out = L ? HT : (dec ? 1 : 0);
            for (int T = 0; T < HT; ++T)
                dx[T] = T < xt_out ? dense_tile_gt<3 * HT>(Wih, in_dim, 3 * H, in_dim, 16 * T, dgi, splat4(0.f)) : splat4(0.f);
        }
    }
}

}  // namespace iplan

extern "C" int iplan_seq2seq_bwd(const IplanSeq2SeqBwdArgs* b, iplan_stream_t stream) {
    using namespace iplan;
    if (!b) return fail(IPLAN_EINVAL, "iplan_seq2seq_bwd: null args");
    const IplanSeq2SeqArgs* a = &b->fwd;
    if (a->rows < 1 || a->T_in < 1 || a->In < 1 || a->In > 64 || a->layers < 1 || a->layers > IPLAN_S2S_MAX_LAYERS || a->P < 1 ||
        a->O < 1 || a->O > 16 || (a->H != 32 && a->H != 64))
        return fail(IPLAN_EINVAL, "iplan_seq2seq_bwd: unsupported dims");
    if (!a->save || !a->params || !b->g_out || !b->dsave) return fail(IPLAN_EINVAL, "iplan_seq2seq_bwd: the forward launch did not save its record, or null tensor pointer");
    const dim3 grid((unsigned)((a->rows + 63) / 64));
    const size_t lds = (size_t)4 * a->layers * (a->H / 16) * 64 * 16;      // the carry: <= 64 KiB
    if (a->H == 32) hipLaunchKernelGGL(seq2seq_bwd_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, *b);
    else hipLaunchKernelGGL(seq2seq_bwd_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, *b);
    return check_launch("iplan_seq2seq_bwd");
}

extern "C" int iplan_seq2seq_fwd(const IplanSeq2SeqArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a) return fail(IPLAN_EINVAL, "iplan_seq2seq_fwd: null args");
    if (a->rows < 1 || a->T_in < 1 || a->In < 1 || a->In > 64 || a->layers < 1 || a->layers > IPLAN_S2S_MAX_LAYERS || a->P < 1 ||
        a->O < 1 || a->O > 16 || (a->H != 32 && a->H != 64))
        return fail(IPLAN_EINVAL, "iplan_seq2seq_fwd: unsupported dims rows=%d T=%d In=%d H=%d layers=%d P=%d O=%d (In <= 64, O <= 16, H in {32, 64})",
                    a->rows, a->T_in, a->In, a->H, a->layers, a->P, a->O);
    if (!a->x || !a->last || !a->params || !a->out) return fail(IPLAN_EINVAL, "iplan_seq2seq_fwd: null tensor pointer");
    if (a->keep && (a->drop_p < 0.f || a->drop_p >= 1.f)) return fail(IPLAN_EINVAL, "iplan_seq2seq_fwd: dropout p=%f", a->drop_p);
    const dim3 grid((unsigned)((a->rows + 63) / 64));
    if (a->H == 32) hipLaunchKernelGGL(seq2seq_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(seq2seq_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_seq2seq_fwd");
}
