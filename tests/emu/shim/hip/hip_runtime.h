// TEST INFRASTRUCTURE ONLY -- a host-side functional emulator of the tiny slice of the HIP
// programming model the iplan_amd kernels use, so that the *unmodified* kernel sources under
// iplan_amd/csrc/ can be compiled with the host clang++ and exercised by the CPU test-suite
// (index math, MFMA fragment layouts, LDS hand-offs, barrier placement) without a GPU.
//
// It is reached only through `-I tests/emu/shim` (this file shadows <hip/hip_runtime.h>); the
// product library libiplan_hip.so is built by hipcc against the real header and never sees it.
// Nothing in iplan_amd/ loads the emulator build: tests/emu/emu_lib.py is the only loader.
//
// Model: a kernel launch runs its workgroups one after another; the threads of a workgroup are
// ucontext fibers scheduled round-robin on the calling OS thread.  Collectives (__syncthreads,
// __shfl*, MFMA) are rendezvous points: a fiber yields until every participant has arrived.
// Wave64; MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md §3
// (v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15,row=4*(l>>4)+reg).
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define IPLAN_HOST_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
inline hipError_t hipGetLastError() { return hipSuccess; }
typedef uint8_t __emu_u8;
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

namespace iplan_emu {

struct Wave {
    int nlanes = 0, arrived = 0, gen = 0;
    float fa[64], fb[64];
    int ia[64];
    float fa8[64][8], fb8[64][8];   // operands of the K = 32 bf16 MFMA (8 k-slots per lane)
};

// Context switch.  glibc's swapcontext saves / restores the signal mask with a system call on every switch (half of the
// emulator's CPU time was spent in the kernel); on x86-64 a fiber switch here is the callee-saved registers and the stack
// pointer (tests/emu/emu_runtime.cpp), elsewhere ucontext.
#if defined(__x86_64__)
#define IPLAN_EMU_FAST_SWITCH 1
struct Ctx {
    void* sp = nullptr;
};
extern "C" void iplan_emu_switch(Ctx* from, Ctx* to);
inline void switch_ctx(Ctx* from, Ctx* to) { iplan_emu_switch(from, to); }
#else
typedef ucontext_t Ctx;
inline void switch_ctx(Ctx* from, Ctx* to) { swapcontext(from, to); }
#endif

struct Fiber {
    Ctx ctx;
    std::vector<char> stack;
    dim3 tid;
    int lane = 0, wave = 0;
    bool done = false;
};

struct Block {
    dim3 bid, bdim, gdim;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int alive = 0, arrived = 0, gen = 0;
    Ctx sched;
    std::function<void()> body;
};

extern Block* g_block;
extern Fiber* g_cur;

inline void yield_() { switch_ctx(&g_cur->ctx, &g_block->sched); }

inline void block_sync() {
    Block* b = g_block;
    int gen = b->gen;
    if (++b->arrived >= b->alive) { b->arrived = 0; b->gen++; return; }
    while (b->gen == gen) yield_();
}

inline void wave_sync() {
    Wave& w = g_block->waves[g_cur->wave];
    int gen = w.gen;
    if (++w.arrived >= w.nlanes) { w.arrived = 0; w.gen++; return; }
    while (w.gen == gen) yield_();
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body);
float* dyn_lds();   // 160 KiB scratch standing in for a kernel's dynamic LDS (workgroups run one at a time)

}  // namespace iplan_emu

#define threadIdx (iplan_emu::g_cur->tid)
#define blockIdx (iplan_emu::g_block->bid)
#define blockDim (iplan_emu::g_block->bdim)
#define gridDim (iplan_emu::g_block->gdim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    iplan_emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { iplan_emu::block_sync(); }

inline float __shfl(float v, int src_lane) {
    auto& w = iplan_emu::g_block->waves[iplan_emu::g_cur->wave];
    w.fa[iplan_emu::g_cur->lane] = v;
    iplan_emu::wave_sync();
    float r = w.fa[src_lane & 63];
    iplan_emu::wave_sync();
    return r;
}
inline int __shfl(int v, int src_lane) {
    auto& w = iplan_emu::g_block->waves[iplan_emu::g_cur->wave];
    w.ia[iplan_emu::g_cur->lane] = v;
    iplan_emu::wave_sync();
    int r = w.ia[src_lane & 63];
    iplan_emu::wave_sync();
    return r;
}
inline float __shfl_xor(float v, int mask) { return __shfl(v, iplan_emu::g_cur->lane ^ mask); }
inline int __shfl_xor(int v, int mask) { return __shfl(v, iplan_emu::g_cur->lane ^ mask); }
inline float __shfl_down(float v, int d) { return __shfl(v, iplan_emu::g_cur->lane + d < 64 ? iplan_emu::g_cur->lane + d : iplan_emu::g_cur->lane); }

typedef float __emu_f32x4 __attribute__((ext_vector_type(4)));
inline __emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, __emu_f32x4 c, int, int, int) {
    auto& w = iplan_emu::g_block->waves[iplan_emu::g_cur->wave];
    int l = iplan_emu::g_cur->lane;
    w.fa[l] = a;
    w.fb[l] = b;
    iplan_emu::wave_sync();
    __emu_f32x4 d = c;
    int col = l & 15;
    for (int reg = 0; reg < 4; ++reg) {
        int row = 4 * (l >> 4) + reg;
        float acc = c[reg];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[row + 16 * k], w.fb[col + 16 * k], acc);
        d[reg] = acc;
    }
    iplan_emu::wave_sync();
    return d;
}

// v_mfma_f32_16x16x32_bf16: lane (i = l & 15, g = l >> 4) holds A[i][8g .. 8g+7] / B[8g .. 8g+7][i]; C/D as above.  The 32
// products of an output element are exact in fp32; they are summed here in double and rounded once with C (the hardware's
// internal summation order is not documented -- the GPU parity tests measure the real thing).
typedef __bf16 __emu_bf16x8 __attribute__((ext_vector_type(8)));
inline __emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(__emu_bf16x8 a, __emu_bf16x8 b, __emu_f32x4 c, int, int, int) {
    auto& w = iplan_emu::g_block->waves[iplan_emu::g_cur->wave];
    int l = iplan_emu::g_cur->lane;
    for (int j = 0; j < 8; ++j) { w.fa8[l][j] = (float)a[j]; w.fb8[l][j] = (float)b[j]; }
    iplan_emu::wave_sync();
    __emu_f32x4 d = c;
    int col = l & 15;
    for (int reg = 0; reg < 4; ++reg) {
        int row = 4 * (l >> 4) + reg;
        double acc = (double)c[reg];
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) acc += (double)w.fa8[row + 16 * g][j] * (double)w.fb8[col + 16 * g][j];
        d[reg] = (float)acc;
    }
    iplan_emu::wave_sync();
    return d;
}

inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline float __expf(float x) { return expf(x); }
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __logf(float x) { return logf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
