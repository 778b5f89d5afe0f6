// Recurrent actor / critic forward (R_Actor, R_Critic) for all agents in one launch, fused with
// the observation assembly of DcntrlMAC._build_inputs[_ippo].
//
// A 16-row tile of one (agent, actor|critic) net is owned by `ksplit` waves of a 512-thread
// workgroup.  The F-wide input row (F = 2485 at Highway chaotic) is gathered straight from the
// episode-buffer fields (history || attention latent || behaviour latent per entity, one-hots
// synthesised in registers), LayerNorm statistics are two-pass over L2-resident data, the
// normalised features feed v_mfma_f32_16x16x4_f32 as the B operand with fc1.weight fragments as A.
// With ksplit = 8 (rollout: 32 rows per agent) the contraction is split across the waves and
// reduced through LDS in a fixed order; with ksplit = 1 (PPO: 22 950 rows per agent) every wave
// streams its own tile.  The 64-wide tail (LN, fc2, LN, GRU step, LN, head, masked log-softmax,
// sampling, entropy) stays in registers of one wave.
#include "ac_fwd_body.h"

namespace iplan {

// NW = waves per workgroup.  The PPO epoch's tail launch (PRE: fc1 pre-activations given, one wave per pair of row tiles, the
// 64-wide weights staged in LDS -- one workgroup per CU) runs 16 waves behind one staging when 8-wave workgroups would need more
// than one round over the chip: a tile pair's tail is a chain of dependent stages (LDS fragment -> MFMA chain -> LayerNorm -> ...),
// and with 2 waves per SIMD the SIMDs sat 1.2 waves deep on average (SQ_WAVE_CYCLES, profiles/r04d_pmc_ppo_train.txt).  128
// registers per lane are enough for it.  Small batches (a data-parallel rank's 2 880 rows: 120 workgroups) keep 8 waves -- one
// round either way, and more CUs share it.
template <int RT, bool PRE, int NW>
__global__ __launch_bounds__(64 * NW) void ac_fwd_kernel(IplanAcFwdArgs a) {
    __shared__ __attribute__((aligned(16))) AcShared<RT, NW> sh;
    const AcGrid gp = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z};
    ac_fwd_body<RT, PRE, false, NW>(a, gp, sh, AcProducers{nullptr, 0, 0});
}

// fc1.weight / feature_norm.{weight, bias} -> the forward kernels' K order and MFMA fragment order (see the header)
__global__ __launch_bounds__(256) void ac_pack_fc1_kernel(IplanAcPackArgs a) {
    const int T = (int)blockIdx.x, net = (int)blockIdx.y;
    const KMap km = make_kmap(a.feat);
    const int F = km.NW + km.n_actions + km.n_id, KT = km.kt0[4];
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    float* __restrict__ out = a.packed + (int64_t)net * a.packed_s_net;
    for (int idx = (int)threadIdx.x; idx < 1024; idx += (int)blockDim.x) {
        const int q = idx & 3, lane = (idx >> 2) & 63, oo = idx >> 8;
        const int n = lane & 15, g = lane >> 4;
        const KTile kt = ktile_at(km, T, 4 * g);
        out[(int64_t)T * 1024 + idx] = q < kt.nv ? P[a.off_w1 + (int64_t)(16 * oo + n) * F + kt.c[q]] : 0.f;
    }
    if (threadIdx.x < 16) {
        const int g = (int)threadIdx.x >> 2, q = (int)threadIdx.x & 3;
        const KTile kt = ktile_at(km, T, 4 * g);
        out[(int64_t)KT * 1024 + T * 16 + threadIdx.x] = q < kt.nv ? P[a.off_fn_w + kt.c[q]] : 0.f;
        out[(int64_t)KT * 1040 + T * 16 + threadIdx.x] = q < kt.nv ? P[a.off_fn_b + kt.c[q]] : 0.f;
    }
}

// (W gamma)[o], (W beta)[o] of fc1 / feature_norm for the folded LayerNorm(F) of the rollout forward
// (fc1(LN(x)) = rstd (W (gamma o x) - mu W gamma) + W beta).  grid (n_nets, 8): a workgroup owns 8 output rows, 32 threads
// per row stride the K axis together (coalesced), partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void ac_pack_wgamma_kernel(IplanAcPackArgs a) {
    __shared__ float s_p[2][8][32];
    const int net = (int)blockIdx.x;
    const KMap km = make_kmap(a.feat);
    const int F = km.NW + km.n_actions + km.n_id, KT = km.kt0[4];
    const float* __restrict__ P = a.params + (int64_t)net * a.params_s_net;
    const int part = (int)threadIdx.x & 31, ro = (int)threadIdx.x >> 5, o = (int)blockIdx.y * 8 + ro;
    float c1 = 0.f, c2 = 0.f;
    for (int c = part; c < F; c += 32) {
        const float wv = P[a.off_w1 + (int64_t)o * F + c];
        c1 = fmaf(wv, P[a.off_fn_w + c], c1);
        c2 = fmaf(wv, P[a.off_fn_b + c], c2);
    }
    s_p[0][ro][part] = c1;
    s_p[1][ro][part] = c2;
    __syncthreads();
    if (part < 2) {                                           // part 0: W gamma, part 1: W beta
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += s_p[part][ro][k];
        a.packed[(int64_t)net * a.packed_s_net + (int64_t)KT * 1056 + part * AM + o] = s;
    }
}

}  // namespace iplan

extern "C" int64_t iplan_ac_packed_floats(const IplanAcFeatures* ft) {
    if (!ft) return 0;
    int t = 0;
    for (int s = 0; s < 3; ++s) t += (ft->N * ft->w[s] + 15) / 16;
    t += (ft->n_actions + ft->n_id + 15) / 16;
    return (int64_t)t * (1024 + 32) + 2 * 64;              // + W gamma, W beta
}

extern "C" int iplan_ac_pack_fc1(const IplanAcPackArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_nets < 1 || !a->params || !a->packed || a->packed_s_net < iplan_ac_packed_floats(&a->feat))
        return fail(IPLAN_EINVAL, "iplan_ac_pack_fc1: bad arguments");
    const int KT = (int)((iplan_ac_packed_floats(&a->feat) - 128) / (1024 + 32));
    if (a->parts < 0 || a->parts > 2) return fail(IPLAN_EINVAL, "iplan_ac_pack_fc1: parts must be 0, 1 or 2");
    if (a->parts != 2) hipLaunchKernelGGL(ac_pack_fc1_kernel, dim3((unsigned)KT, (unsigned)a->n_nets), dim3(256), 0, (hipStream_t)stream, *a);
    if (a->parts != 1) hipLaunchKernelGGL(ac_pack_wgamma_kernel, dim3((unsigned)a->n_nets, AM / 8), dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ac_pack_fc1");
}

// argument checks of iplan_ac_fwd, shared with the fused rollout launch (gat.hip: iplan_gat_enc_ac_fwd)
__attribute__((visibility("hidden"))) int iplan::ac_fwd_check(const IplanAcFwdArgs* a) {
    if (!a) return fail(IPLAN_EINVAL, "iplan_ac_fwd: null args");
    if (a->ksplit != 1 && a->ksplit != 8) return fail(IPLAN_EINVAL, "iplan_ac_fwd: ksplit must be 1 or 8 (got %d)", a->ksplit);
    if (a->which < 0 || a->which > 2 || a->n_agents < 1 || a->rows < 1)
        return fail(IPLAN_EINVAL, "iplan_ac_fwd: bad which/n_agents/rows");
    if (a->feat.T < 1 || a->feat.T_phys < a->feat.T) return fail(IPLAN_EINVAL, "iplan_ac_fwd: bad T/T_phys");
    if (a->which != 1 && (a->actor.n_out < 1 || a->actor.n_out > 16))
        return fail(IPLAN_EINVAL, "iplan_ac_fwd: n_actions=%d outside [1,16]", a->actor.n_out);
    if (a->which != 1 && a->mode == 1 && !a->q_noise) return fail(IPLAN_EINVAL, "iplan_ac_fwd: mode 1 needs q_noise");
    if (a->which != 1 && a->mode == 2 && !a->actions_in) return fail(IPLAN_EINVAL, "iplan_ac_fwd: mode 2 needs actions_in");
    if (a->fc1_pre && (a->ksplit != 1 || a->ln_stats_mode != 2 || !a->ln_stats || !iplan::aligned16(a->fc1_pre)))
        return fail(IPLAN_EINVAL, "iplan_ac_fwd: fc1_pre needs ksplit 1 and stored LayerNorm statistics (ln_stats_mode 2)");
    if (a->ksplit_wg > 1 && (a->ksplit != 8 || a->ln_stats_mode != 0 || !a->ks_scratch || !a->ks_count || a->saved ||
                             (a->which != 1 && !a->packed_actor) || (a->which != 0 && !a->packed_critic) || a->ksplit_wg > 8))
        return fail(IPLAN_EINVAL, "iplan_ac_fwd: ksplit_wg needs the rollout shape (ksplit 8, folded LayerNorm statistics, packed fc1 operands incl. "
                                  "W gamma / W beta, no saved activations), ks_scratch and ks_count");
    return IPLAN_OK;
}

extern "C" int iplan_ac_fwd(const IplanAcFwdArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = ac_fwd_check(a)) return rc;
    const int tiles = (a->rows + 15) / 16;
    const unsigned nz = a->which == 2 ? 2u : 1u;
    if (a->ksplit == 8) {
        dim3 grid((unsigned)(tiles * (a->ksplit_wg > 1 ? a->ksplit_wg : 1)), (unsigned)a->n_agents, nz);
        hipLaunchKernelGGL((ac_fwd_kernel<1, false, 8>), grid, dim3(512), 0, (hipStream_t)stream, *a);
    } else {
        const int wg8 = (tiles + 8 * AC_RT - 1) / (8 * AC_RT);                                    // 8 waves x AC_RT row tiles per workgroup
        const char* env = getenv("IPLAN_AC_PRE_WAVES");                                          // (A/B knob: 8 or 16)
        const bool wide = a->fc1_pre && (env ? atoi(env) == 16 : (int64_t)wg8 * a->n_agents * nz > 256);
        dim3 grid((unsigned)(wide ? (wg8 + 1) / 2 : wg8), (unsigned)a->n_agents, nz);
        if (wide) hipLaunchKernelGGL((ac_fwd_kernel<AC_RT, true, 16>), grid, dim3(1024), 0, (hipStream_t)stream, *a);
        else if (a->fc1_pre) hipLaunchKernelGGL((ac_fwd_kernel<AC_RT, true, 8>), grid, dim3(512), 0, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL((ac_fwd_kernel<AC_RT, false, 8>), grid, dim3(512), 0, (hipStream_t)stream, *a);
    }
    return check_launch("iplan_ac_fwd");
}
