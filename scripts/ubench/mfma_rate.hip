// Issue-rate probe for v_mfma_f32_16x16x4_f32 on gfx950: cycles per MFMA for a few dependency / occupancy patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NACC, bool LDSOP>
__global__ void probe(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float s[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += blockDim.x) s[i] = 0.001f * (i % 97);
    __syncthreads();
    const int l = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = {0.f, 0.f, 0.f, 0.f};
    float bv = 0.5f + l * 0.01f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (LDSOP) {
            f32x4 f[NACC];
            for (int a = 0; a < NACC; ++a) f[a] = *reinterpret_cast<const f32x4*>(s + (16 * (a & 3) + (l & 15)) * 72 + 16 * (it & 3) + 4 * (l >> 4));
            for (int q = 0; q < 4; ++q)
                for (int a = 0; a < NACC; ++a) acc[a] = MFMA(f[a][q], bv, acc[a]);
        } else {
            for (int q = 0; q < 4; ++q)
                for (int a = 0; a < NACC; ++a) acc[a] = MFMA(bv, bv + q, acc[a]);
        }
    }
    long long t1 = clock64();
    float r = 0.f;
    for (int a = 0; a < NACC; ++a) r += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool LDSOP>
void run(const char* name, int threads, int blocks, int iters) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC, LDSOP><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC, LDSOP><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * NACC;
    printf("%-34s threads/WG %4d blocks %4d: %8.2f clock64 ticks / MFMA / wave, kernel %.3f ms -> %.1f ns / MFMA / wave\n", name, threads, blocks,
           (double)h / n, ms, ms * 1e6 / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 20000;
    run<1, false>("dependent chain (1 acc)", 256, 256, it);
    run<2, false>("2 acc round robin", 256, 256, it);
    run<4, false>("4 acc round robin", 256, 256, it);
    run<4, false>("4 acc, 1 wave/CU", 64, 256, it);
    run<4, false>("4 acc, 2 waves/SIMD", 512, 256, it);
    run<4, true>("4 acc + LDS frags", 256, 256, it);
    run<3, true>("3 acc + LDS frags", 256, 256, it);
    run<4, true>("4 acc + LDS frags, 2 waves/SIMD", 512, 256, it);
    run<4, false>("4 acc, 138 blocks", 256, 138, it);
    return 0;
}
