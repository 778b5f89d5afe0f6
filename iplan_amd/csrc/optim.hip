// Flat-arena optimiser kernels: L2 norm of a gradient slice, and clip + Adam fused.
// (clip_grad_norm_ + torch.optim.Adam of learners/ippo_learner.py:204-221,
//  nova/prediction_policy.py:231-241, nova/stable_behavior_policy.py:252-262.)
// Pure streaming fp32: bound by HBM (4 reads + 3 writes of the slice); at these sizes (<= 1 MB per
// net) it is launch-latency bound, which is why all nets go through one launch.
#include "api_util.h"
#include "wave_tile.h"

namespace iplan {

// out[net*out_stride + slot] = sum(g[net*g_stride + off .. + n)^2); one block per net, fixed order.
// Latency bound (one workgroup streams the whole slice): 16-byte loads, four of them in flight per thread -- with one dword
// per thread and iteration the 190 000-float actor slice took 55 us, twice per PPO epoch.
__global__ __launch_bounds__(1024) void sqnorm_kernel(const float* __restrict__ g, int64_t g_stride, int64_t off,
                                                      int64_t n, float* __restrict__ out, int out_stride, int slot) {
    __shared__ float s_part[16];
    const float* p = g + (int64_t)blockIdx.x * g_stride + off;
    float acc = 0.f;
    int64_t done = 0;
    if (aligned16(p)) {
        const f32x4* __restrict__ p4 = reinterpret_cast<const f32x4*>(p);
        const int64_t n4 = n >> 2, B = blockDim.x;
        f32x4 a4[4];
        for (int u = 0; u < 4; ++u) a4[u] = splat4(0.f);
        int64_t i = threadIdx.x;
        for (; i + 3 * B < n4; i += 4 * B) {
            const f32x4 v0 = p4[i], v1 = p4[i + B], v2 = p4[i + 2 * B], v3 = p4[i + 3 * B];
            for (int q = 0; q < 4; ++q) {
                a4[0][q] = fmaf(v0[q], v0[q], a4[0][q]);
                a4[1][q] = fmaf(v1[q], v1[q], a4[1][q]);
                a4[2][q] = fmaf(v2[q], v2[q], a4[2][q]);
                a4[3][q] = fmaf(v3[q], v3[q], a4[3][q]);
            }
        }
        for (; i < n4; i += B) {
            const f32x4 v = p4[i];
            for (int q = 0; q < 4; ++q) a4[0][q] = fmaf(v[q], v[q], a4[0][q]);
        }
        for (int u = 0; u < 4; ++u) acc += (a4[u][0] + a4[u][1]) + (a4[u][2] + a4[u][3]);
        done = n4 << 2;
    }
    for (int64_t i = done + threadIdx.x; i < n; i += blockDim.x) acc = fmaf(p[i], p[i], acc);
    acc = wave_sum(acc);
    if (lane_id() == 0) s_part[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[w];
        out[(int64_t)blockIdx.x * out_stride + slot] = t;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(IplanAdamArgs a) {
    const int net = (int)blockIdx.y;
    float scale = 1.0f;
    if (a.sqnorm) {
        const float nrm = sqrtf(a.sqnorm[(int64_t)net * a.sqnorm_stride + a.sqnorm_slot]);
        scale = fminf(1.0f, a.max_norm / (nrm + 1e-6f));               // clip_grad_norm_ semantics
    }
    const int64_t base = (int64_t)net * a.stride + a.off;
    const float bc1 = a.bc1[net], bc2s = a.bc2_sqrt[net];
    const float step_size = a.lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        float g = a.grad[base + i] * scale;
        if (a.write_clipped) a.grad[base + i] = g;
        if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, a.param[base + i], g);
        const float m = a.exp_avg[base + i] * a.beta1 + (1.0f - a.beta1) * g;
        const float v = a.exp_avg_sq[base + i] * a.beta2 + (1.0f - a.beta2) * g * g;
        a.exp_avg[base + i] = m;
        a.exp_avg_sq[base + i] = v;
        const float denom = sqrtf(v) / bc2s + a.eps;
        a.param[base + i] -= step_size * (m / denom);
    }
}

// Gumbel noise: two uniforms per 64-bit mix of (seed, counter) -- splitmix64's finaliser on a Weyl sequence, the generator
// behind java.util.SplittableRandom (passes BigCrush) --, 23 random bits each, centred so that u is never 0 or 1.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float gumbel_of(uint32_t bits) {
    const float u = ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f);      // 23 bits: k + 0.5 is exact, u in [2^-24, 1 - 2^-24]
    return -logf(-logf(u));
}
__global__ __launch_bounds__(256) void gumbel_kernel(float* __restrict__ out, int64_t n4, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t a = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(2 * i + 1)), b = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(2 * i + 2));
        f32x4 v;
        v[0] = gumbel_of((uint32_t)a); v[1] = gumbel_of((uint32_t)(a >> 32));
        v[2] = gumbel_of((uint32_t)b); v[3] = gumbel_of((uint32_t)(b >> 32));
        *reinterpret_cast<f32x4*>(out + 4 * i) = v;
    }
}

}  // namespace iplan

extern "C" int iplan_gumbel_noise(float* out, int64_t n, uint64_t seed, iplan_stream_t stream) {
    using namespace iplan;
    if (!out || n < 0 || (n & 3) || !aligned16(out)) return fail(IPLAN_EINVAL, "iplan_gumbel_noise: out must be 16-byte aligned and n a multiple of 4");
    const int64_t n4 = n / 4;
    if (n4 == 0) return IPLAN_OK;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipLaunchKernelGGL(gumbel_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, n4, seed);
    return check_launch("iplan_gumbel_noise");
}

extern "C" int iplan_grad_sqnorm(const float* grad, int64_t stride, int64_t off, int64_t n, int32_t n_nets,
                                 float* out, int32_t out_stride, int32_t slot, iplan_stream_t stream) {
    using namespace iplan;
    if (!grad || !out || n < 0 || n_nets < 1) return fail(IPLAN_EINVAL, "iplan_grad_sqnorm: bad arguments");
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)n_nets), dim3(1024), 0, (hipStream_t)stream, grad, stride, off, n,
                       out, (int)out_stride, (int)slot);
    return check_launch("iplan_grad_sqnorm");
}

extern "C" int iplan_adam_step(const IplanAdamArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || a->n_nets < 1 || a->n_nets > IPLAN_MAX_NETS)
        return fail(IPLAN_EINVAL, "iplan_adam_step: bad arguments");
    const unsigned blocks = (unsigned)((a->n + 255) / 256 > 1024 ? 1024 : (a->n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks ? blocks : 1, (unsigned)a->n_nets), dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_adam_step");
}
