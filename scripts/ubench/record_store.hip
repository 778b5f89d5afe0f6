// Store-pattern probe for the behaviour-learning records (round 3): how fast can 184 workgroups x 8 waves write 13.8 GB when
// every 16-byte-per-lane store instruction lands as
//   mode 0: 16 segments of 64 B, one per chain, chains 1.5 MB apart        (records [chain][step][496]: the round-2 layout)
//   mode 1: 16 segments of 64 B, chains 1 984 B apart                      (records [step][chain][496])
//   mode 2: one contiguous 1 KiB block                                      (records [tile][step][column group][chain][16])
//   mode 3: one contiguous 1 KiB block, column groups STEPS KiB apart      (records [tile][column group][step][chain][16]: rows of
//           one column group contiguous over (step, chain) -- what a weight-gradient contraction would stream)
// Build: hipcc --offload-arch=gfx950 -O3 record_store.hip -o record_store ; run: ./record_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int COLS = 496, STEPS = 790, TILES_PER_WG = 3, GROUPS = COLS / 16;   // 31 column groups of 16 floats

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(float* __restrict__ rec, int tiles) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, n = l & 15, g = l >> 4;
    for (int s = 0; s < STEPS; ++s)
        for (int k = 0; k < TILES_PER_WG; ++k) {
            const int tile = blockIdx.x * TILES_PER_WG + k;
            if (tile >= tiles) continue;
            // the 31 column groups of a tile-step are shared out over the 8 waves
            for (int cg = w; cg < GROUPS; cg += 8) {
                size_t off;
                if (MODE == 0) off = ((size_t)(tile * 16 + n) * STEPS + s) * COLS + cg * 16 + 4 * g;
                else if (MODE == 1) off = ((size_t)((size_t)tile * STEPS + s) * 16 + n) * COLS + cg * 16 + 4 * g;
                else if (MODE == 2) off = ((((size_t)tile * STEPS + s) * GROUPS + cg) * 16 + n) * 16 + 4 * g;
                else off = ((((size_t)tile * GROUPS + cg) * STEPS + s) * 16 + n) * 16 + 4 * g;
                f32x4 v = {(float)s, (float)k, (float)cg, (float)l};
                *reinterpret_cast<f32x4*>(rec + off) = v;
            }
        }
}

template <int MODE>
__global__ __launch_bounds__(512) void load_kernel(const float* __restrict__ rec, int tiles, float* out) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, n = l & 15, g = l >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < STEPS; ++s)
        for (int k = 0; k < TILES_PER_WG; ++k) {
            const int tile = blockIdx.x * TILES_PER_WG + k;
            if (tile >= tiles) continue;
            for (int cg = w; cg < GROUPS; cg += 8) {
                size_t off;
                if (MODE == 0) off = ((size_t)(tile * 16 + n) * STEPS + s) * COLS + cg * 16 + 4 * g;
                else if (MODE == 1) off = ((size_t)((size_t)tile * STEPS + s) * 16 + n) * COLS + cg * 16 + 4 * g;
                else if (MODE == 2) off = ((((size_t)tile * STEPS + s) * GROUPS + cg) * 16 + n) * 16 + 4 * g;
                else off = ((((size_t)tile * GROUPS + cg) * STEPS + s) * 16 + n) * 16 + 4 * g;
                acc += *reinterpret_cast<const f32x4*>(rec + off);
            }
        }
    if (acc[0] == 12345.f) out[0] = acc[1];
}

template <int MODE> void run(float* rec, int tiles, double bytes, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = (tiles + TILES_PER_WG - 1) / TILES_PER_WG;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(store_kernel<MODE>, dim3(wgs), dim3(512), 0, 0, rec, tiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (pass) printf("mode %d  store %7.3f ms  %6.2f TB/s", MODE, ms, bytes / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(load_kernel<MODE>, dim3(wgs), dim3(512), 0, 0, rec, tiles, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass) printf("   load %7.3f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
    }
}

int main() {
    const int tiles = 550;
    const size_t floats = (size_t)tiles * 16 * STEPS * COLS;
    float *rec, *out;
    hipMalloc(&rec, floats * 4); hipMalloc(&out, 16);
    const double bytes = (double)floats * 4;
    printf("%.2f GB, %d workgroups\n", bytes / 1e9, (tiles + 2) / 3);
    run<0>(rec, tiles, bytes, out); run<1>(rec, tiles, bytes, out); run<2>(rec, tiles, bytes, out); run<3>(rec, tiles, bytes, out);
    return 0;
}
