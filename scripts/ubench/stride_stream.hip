// Does a wave that streams 16 far-apart records per step (chain-major layout) reach the same bandwidth as one
// streaming 16 adjacent records (tile-step-major)?  552 x 2 waves like the decoder kernels, 26 x 16 B loads per lane per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(const float* __restrict__ src, float* __restrict__ dst, int steps, long chain_stride, long step_stride,
                                             long tile_stride, int do_store) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, n = l & 15, g = l >> 4;
    const int tile = blockIdx.x * 4 + (w & 3), hf = w >> 2;
    const float* p = src + tile * tile_stride + n * chain_stride + 4 * g + 32 * hf * 4;
    float* q = dst + tile * tile_stride + n * chain_stride + 4 * g + 32 * hf * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 cur[13];
    for (int k = 0; k < 13; ++k) cur[k] = *reinterpret_cast<const f32x4*>(p + 16 * k);
    for (int s = 0; s < steps; ++s) {
        f32x4 nxt[13];
        const float* pn = p + (s + 1 < steps ? (s + 1) : s) * step_stride;
        for (int k = 0; k < 13; ++k) nxt[k] = *reinterpret_cast<const f32x4*>(pn + 16 * k);
        for (int k = 0; k < 13; ++k) acc += cur[k];
        // some ALU time per step
        for (int r = 0; r < 200; ++r) acc = acc * 1.0001f + 0.5f;
        if (do_store)
            for (int k = 0; k < 6; ++k) *reinterpret_cast<f32x4*>(q + s * step_stride + 16 * k) = acc;
        for (int k = 0; k < 13; ++k) cur[k] = nxt[k];
    }
    if (acc[0] == 123.456f) dst[0] = acc[1];
}

int main() {
    const int tiles = 552, steps = 790, rec = 496;
    const size_t total = (size_t)tiles * 16 * steps * rec;
    float *src, *dst;
    hipMalloc(&src, total * 4); hipMalloc(&dst, total * 4);
    hipMemset(src, 0, total * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int st = 0; st < 2; ++st)
        for (int mode = 0; mode < 2; ++mode) {
            // mode 0: chain-major [tile][chain][step][rec]; mode 1: tile-step-major [tile][step][chain][rec]
            const long chain_stride = mode ? rec : (long)steps * rec, step_stride = mode ? 16L * rec : rec, tile_stride = 16L * steps * rec;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                probe<<<138, 512>>>(src, dst, steps, chain_stride, step_stride, tile_stride, st);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double gb = 552.0 * 2 * 64 * steps * (13 * 16 + st * 6 * 16) / 1e9;
            printf("%s %-16s: %.3f ms  %.2f TB/s\n", st ? "load+store" : "load only ", mode ? "tile-step-major" : "chain-major", ms, gb / ms);
        }
    return 0;
}
