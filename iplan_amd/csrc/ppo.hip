// PPO bookkeeping kernels of IPPOLearner.train (learners/ippo_learner.py:227-317): GAE returns +
// advantage normalisation, and the clipped policy / clipped-Huber value losses with their
// gradients w.r.t. log-prob and value.  One 1024-thread workgroup per agent, fixed summation order
// (a row-strided partial per thread, then a tree over the workgroup) -> reproducible.
#include "api_util.h"
#include "wave_tile.h"

namespace iplan {

__device__ __forceinline__ float block_sum_1024(float v, float* s_part) {
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) s_part[wave_id()] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[w];
    return t;
}

// compute_returns (ippo_learner.py:344-365) + advantage normalisation (:273-279)
__global__ __launch_bounds__(1024) void ppo_prepare_kernel(IplanPpoPrepareArgs a) {
    __shared__ float s_part[16];
    const int net = (int)blockIdx.x;
    const int T = a.T, T1 = a.T + 1, bs = a.bs;
    const float* __restrict__ val = a.values + (int64_t)net * bs * T1;
    float* __restrict__ ret = a.returns + (int64_t)net * bs * T;
    float* __restrict__ adv = a.adv + (int64_t)net * bs * T;
    float* __restrict__ msk = a.mask + (int64_t)net * bs * T;
    float* __restrict__ vpr = a.value_preds + (int64_t)net * bs * T;
    for (int b = (int)threadIdx.x; b < bs; b += (int)blockDim.x) {
        const float* rw = a.reward + (int64_t)net * a.rw_s_net + (int64_t)b * a.rw_s_ep;
        const uint8_t* tm = a.terminated + (int64_t)net * a.tm_s_net + (int64_t)b * a.tm_s_ep;
        float gae = 0.f, disc = val[(int64_t)b * T1 + T];     // (no_gae: bootstrapped discounted return, :360-362)
        for (int t = T - 1; t >= 0; --t) {
            const float m1 = 1.0f - (float)tm[(int64_t)(t + 1) * a.tm_s_t];
            const float v0 = val[(int64_t)b * T1 + t], v1 = val[(int64_t)b * T1 + t + 1];
            const float delta = rw[(int64_t)t * a.rw_s_t] + a.gamma * v1 * m1 - v0;
            gae = delta + a.gamma * a.lam * m1 * gae;
            disc = disc * a.gamma * m1 + rw[(int64_t)t * a.rw_s_t];
            const float r = a.no_gae ? disc : gae + v0;
            const float m0 = 1.0f - (float)tm[(int64_t)t * a.tm_s_t];
            ret[(int64_t)b * T + t] = r;
            msk[(int64_t)b * T + t] = m0;
            vpr[(int64_t)b * T + t] = v0;
            adv[(int64_t)b * T + t] = m0 == 0.0f ? 0.0f : r - v0;
        }
    }
    if (a.skip_norm) return;                                // the caller normalises over all data-parallel ranks
    __syncthreads();
    const int n = bs * T;
    float s = 0.f;
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) s += adv[i];
    const float mean = block_sum_1024(s, s_part) / (float)n;
    float q = 0.f;
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) { const float d = adv[i] - mean; q = fmaf(d, d, q); }
    const float var = block_sum_1024(q, s_part) / (float)(n - 1);           // th.std_mean: unbiased
    const float inv = 1.0f / (sqrtf(var) + 1e-5f);
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) adv[i] = (adv[i] - mean) * inv;
}

// advantage normalisation over the rows of all data-parallel ranks (see the header): partial sums per rank
__global__ __launch_bounds__(1024) void adv_norm_kernel(IplanAdvNormArgs a) {
    __shared__ float s_part[16];
    const int net = (int)blockIdx.x;
    float* __restrict__ adv = a.adv + (int64_t)net * a.row_stride;
    const int n = a.n;
    if (a.phase == 0) {
        float s = 0.f;
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) s += adv[i];
        s = block_sum_1024(s, s_part);
        if (threadIdx.x == 0) a.sum[net] = s;
        return;
    }
    const float mean = a.sum[net] / a.count;
    if (a.phase == 1) {
        float q = 0.f;
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) { const float d = adv[i] - mean; q = fmaf(d, d, q); }
        q = block_sum_1024(q, s_part);
        if (threadIdx.x == 0) a.sqdev[net] = q;
        return;
    }
    const float inv = 1.0f / (sqrtf(a.sqdev[net] / (a.count - 1.0f)) + 1e-5f);           // th.std_mean: unbiased
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) adv[i] = (adv[i] - mean) * inv;
}

__device__ __forceinline__ float huber_q(float e, float d) {      // util.py:33-36 (one-sided)
    const float ae = fabsf(e);
    return (ae <= d ? e * e * 0.5f : 0.f) + (e > d ? d * (ae - d * 0.5f) : 0.f);
}
__device__ __forceinline__ float huber_dq(float e, float d) {
    return (fabsf(e) <= d ? e : 0.f) + (e > d ? d : 0.f);
}

// ppo_update losses (ippo_learner.py:185-197, 128-159) and their gradients per row
__global__ __launch_bounds__(1024) void ppo_loss_kernel(IplanPpoLossArgs a) {
    __shared__ float s_part[16];
    const int net = (int)blockIdx.x;
    const int64_t o = (int64_t)net * a.row_stride;
    const float* __restrict__ logp = a.logp + (int64_t)net * a.rows;
    const float* __restrict__ val = a.values + (int64_t)net * a.rows;
    const float* __restrict__ ent = a.entropy + (int64_t)net * a.rows;
    float* __restrict__ g_lp = a.g_logp + (int64_t)net * a.rows;
    float* __restrict__ g_v = a.g_values + (int64_t)net * a.rows;
    const int n = a.rows;
    // the rows of this workgroup: all of them, or range blockIdx.y of n_parts (then the denominator comes from the caller)
    const int P = a.n_parts > 1 ? a.n_parts : 1, part = (int)blockIdx.y;
    const int per = (n + P - 1) / P, r0 = part * per, r1 = r0 + per < n ? r0 + per : n;
    float msum;
    if (P > 1) {
        msum = a.mask_sum[net];
    } else {
        float sm = 0.f;
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) sm += a.mask[o + i];
        const float msum_own = block_sum_1024(sm, s_part);
        msum = a.mask_sum ? a.mask_sum[net] : msum_own;
    }
    const float nrow = a.row_count > 0.f ? a.row_count : (float)n;
    const bool mse = a.flags & IPLAN_PPO_MSE, no_vclip = a.flags & IPLAN_PPO_NO_VCLIP;
    const bool v_mean = a.flags & IPLAN_PPO_VALUE_MEAN, p_mean = a.flags & IPLAN_PPO_POLICY_MEAN;
    float pol = 0.f, vls = 0.f, rat = 0.f, en = 0.f;
    for (int i = r0 + (int)threadIdx.x; i < r1; i += (int)blockDim.x) {
        const float m = a.mask[o + i], ad = a.adv[o + i];
        const float ratio = expf(logp[i] - a.old_logp[o + i]);
        const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = ratio * ad, s2 = rc * ad;
        const float wp = p_mean ? 1.0f / nrow : m / msum, wv = v_mean ? 1.0f / nrow : m / msum;   // row weights of the two losses
        pol -= fminf(s1, s2) * wp;
        const bool inside = ratio >= lo && ratio <= hi;
        // th.min ties split evenly; inside the clip range both branches carry d/dratio = adv
        float dr = 0.f;
        if (s1 < s2) dr = ad;
        else if (s1 == s2) dr = inside ? ad : 0.5f * ad;
        g_lp[i] = -wp * dr * ratio;
        rat += ratio;
        en += ent[i];
        const float v = val[i], vp = a.value_preds[o + i], rt = a.returns[o + i];
        const float dv = v - vp;
        const float vc = vp + fminf(fmaxf(dv, -a.clip), a.clip);
        const float e1 = rt - v, e2 = rt - vc;
        const float h1 = mse ? 0.5f * e1 * e1 : huber_q(e1, a.huber_delta), h2 = mse ? 0.5f * e2 * e2 : huber_q(e2, a.huber_delta);
        const float d1 = -(mse ? e1 : huber_dq(e1, a.huber_delta));
        const float d2 = (dv >= -a.clip && dv <= a.clip) ? -(mse ? e2 : huber_dq(e2, a.huber_delta)) : 0.f;
        vls += (no_vclip ? h1 : fmaxf(h1, h2)) * wv;
        const float dvl = (no_vclip || h1 > h2) ? d1 : (h2 > h1 ? d2 : 0.5f * (d1 + d2));
        g_v[i] = a.value_loss_coef * wv * dvl;
    }
    pol = block_sum_1024(pol, s_part);
    vls = block_sum_1024(vls, s_part);
    rat = block_sum_1024(rat, s_part);
    en = block_sum_1024(en, s_part);
    if (threadIdx.x == 0) {
        float* st = a.stats + ((int64_t)net * P + part) * 8;
        st[0] = pol;                   // policy_loss (row weights applied above)
        st[1] = vls;                   // value_loss
        st[2] = rat / (float)n;        // imp_weights.mean()
        st[3] = en / (float)n;         // dist_entropy (unmasked mean, act.py:164)
        st[4] = part == 0 ? msum : 0.f;   // (P > 1: every entry is this range's SHARE -- the caller adds the P of them)
    }
}

}  // namespace iplan

extern "C" int iplan_ppo_prepare(const IplanPpoPrepareArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_agents < 1 || a->bs < 1 || a->T < 1 || a->bs * a->T < 2 || !a->reward || !a->terminated ||
        !a->values || !a->returns || !a->adv || !a->mask || !a->value_preds)
        return fail(IPLAN_EINVAL, "iplan_ppo_prepare: bad arguments");
    hipLaunchKernelGGL(ppo_prepare_kernel, dim3((unsigned)a->n_agents), dim3(1024), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ppo_prepare");
}

extern "C" int iplan_ppo_loss(const IplanPpoLossArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_agents < 1 || a->rows < 1 || !a->logp || !a->entropy || !a->values || !a->old_logp || !a->adv ||
        !a->value_preds || !a->returns || !a->mask || !a->g_logp || !a->g_values || !a->stats)
        return fail(IPLAN_EINVAL, "iplan_ppo_loss: bad arguments");
    if (a->n_parts > 1 && (!a->mask_sum || a->n_parts > 1024))
        return fail(IPLAN_EINVAL, "iplan_ppo_loss: n_parts > 1 needs mask_sum (and at most 1024 parts)");
    hipLaunchKernelGGL(ppo_loss_kernel, dim3((unsigned)a->n_agents, (unsigned)(a->n_parts > 1 ? a->n_parts : 1)), dim3(1024), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ppo_loss");
}

extern "C" int iplan_ppo_adv_norm(const IplanAdvNormArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_agents < 1 || a->n < 1 || !a->adv || !a->sum || !a->sqdev || a->phase < 0 || a->phase > 2 || !(a->count >= 2.0f))
        return fail(IPLAN_EINVAL, "iplan_ppo_adv_norm: bad arguments");
    hipLaunchKernelGGL(adv_norm_kernel, dim3((unsigned)a->n_agents), dim3(1024), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_ppo_adv_norm");
}
