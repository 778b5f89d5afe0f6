// Can VALU / transcendental work of one wave overlap with MFMAs of another wave on the same SIMD (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// mode bit 0: waves with (wave_id < 4) run MFMAs; bit 1: waves >= 4 (or all if WG = 256 and bit0 clear) run VALU work
// kind: 0 = fma chain (full rate), 1 = exp2 (transcendental), 2 = integer mul_lo
template <int kind>
__global__ void probe(float* out, int iters, int mode, int same_wave) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    f32x4 acc[4];
    for (int a = 0; a < 4; ++a) acc[a] = {0.f, 0.f, 0.f, 0.f};
    float bv = 0.5f + l * 0.01f;
    float v[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = bv + i; u[i] = l * 7 + i; }
    const bool do_m = same_wave ? (mode & 1) : ((mode & 1) && w < 4);
    const bool do_v = same_wave ? (mode & 2) : ((mode & 2) && (w >= 4 || blockDim.x == 256));
    for (int it = 0; it < iters; ++it) {
        if (do_m)
            for (int q = 0; q < 4; ++q)
                for (int a = 0; a < 4; ++a) acc[a] = MFMA(bv, bv + q, acc[a]);
        if (do_v) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (kind == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);
                    else if (kind == 1) v[i] = __builtin_amdgcn_exp2f(v[i] * 0.001f);
                    else u[i] = u[i] * 2654435761u + 12345u;
                }
        }
    }
    float r = 0.f;
    for (int a = 0; a < 4; ++a) r += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    for (int i = 0; i < 8; ++i) r += v[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int kind>
float run(int threads, int iters, int mode, int same) {
    float* out;
    hipMalloc(&out, sizeof(float) * threads * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<kind><<<256, threads>>>(out, iters, mode, same);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<kind><<<256, threads>>>(out, iters, mode, same);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

int main() {
    const int it = 20000;   // per iteration: 16 MFMAs (512 pipe cycles) and/or 32 VALU ops
    const char* kn[3] = {"fma", "exp2", "mul_lo_u32"};
#define ROW(K) printf("%-10s  two waves/SIMD: mfma only %.3f  valu only %.3f  both (different waves) %.3f | one wave: mfma %.3f valu %.3f both (same wave) %.3f ms\n", kn[K], \
               run<K>(512, it, 1, 0), run<K>(512, it, 2, 0), run<K>(512, it, 3, 0), run<K>(256, it, 1, 1), run<K>(256, it, 2, 1), run<K>(256, it, 3, 1));
    ROW(0) ROW(1) ROW(2)
    return 0;
}
