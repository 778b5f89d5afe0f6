// Wave-tile engine: the building blocks every iplan_amd kernel is written in.
//
// One wavefront (64 lanes) owns a tile of 16 independent "chains" (rows of a batch: an entity, an
// (env,entity) pair, a PPO sample ...).  A per-chain vector of dimension Dm (multiple of 16) lives
// in registers in the *D layout* of v_mfma_f32_16x16x4_f32:
//
//      lane l = (n = l & 15, g = l >> 4)  holds  vec_n[16*t + 4*g + r]   in  v[t][r],  t < Dm/16, r < 4
//
// With that layout a dense layer y = W x (+ b) is, per 16-output tile t', a chain of MFMAs
//      acc = mfma(A = W[16t'+m][16T+4g+r]  (lane m=l&15,g),  B = x[T][r],  acc)      T < In/16, r < 4
// because the hardware contracts over the 4 lane-groups g and the K index may be visited in any
// order as long as A and B agree.  The result lands in the D layout again (col = chain n,
// row = 4g+reg), so GRU gates, activations, LayerNorm terms and the next layer's B operand are
// all lane-local: no transposes, no LDS round trip between layers or between time steps.
// The A fragment of a row-major weight matrix is 4 consecutive floats -> one 16-byte load.
//
// All results are fp32: v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain (guide §3), which is what the 1e-5 parity contract
// with the reference's fp32 CPU path needs; contractions whose weights are loop invariants in registers may run in the
// fp32-exact split-bf16 form on the bf16 matrix cores instead (split_bf3 / mfma_bf16 below).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// dynamic LDS of a kernel (the host-emulated test build has no `extern __shared__`)
#ifdef IPLAN_HOST_EMULATION
#define IPLAN_DYN_LDS(name) float* name = iplan_emu::dyn_lds()
#else
#define IPLAN_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) float name[]
#endif

// Scheduling fence: keeps the compiler from hoisting the next tile's LDS fragment reads above the
// current tile's MFMA chain (which otherwise blows the register budget in the 64-wide GRU kernels).
#ifdef IPLAN_HOST_EMULATION
#define IPLAN_CLOCK() ((int64_t)0)
#else
#define IPLAN_CLOCK() ((int64_t)wall_clock64())
#endif
#ifdef IPLAN_HOST_EMULATION
#define IPLAN_SCHED_FENCE() do {} while (0)
#define IPLAN_WAVE_SYNC() iplan_emu::wave_sync()         /* lanes are fibres: LDS hand-offs inside a wave need a rendezvous */
#else
#define IPLAN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define IPLAN_WAVE_SYNC() __builtin_amdgcn_wave_barrier()  /* a wave's LDS accesses execute in order: nothing to wait for */
#endif

// Workgroup barrier that orders LDS traffic ONLY: `__syncthreads()` is fence + barrier, and the fence drains vmcnt -- every
// global load / store in flight (the BPTT kernels prefetch the next step's record and stream their row gradients) is waited
// for at every barrier, i.e. the HBM latency the prefetch was meant to hide is paid in full once or twice per step
// (ablations in profiles/r02d_notes.md: the decoder BPTT spent 45 % of its time on its loads, 26 % on its stores).  The step
// loops exchange data between waves through LDS alone, so they wait for their own LDS operations and nothing else.
#ifdef IPLAN_HOST_EMULATION
#define IPLAN_LDS_BARRIER() __syncthreads()
#else
#define IPLAN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

namespace iplan {

// Rendezvous of the `n_waves` waves that share ONE chain tile (the decoder's four quarter-waves), through a counter in LDS
// instead of s_barrier: a workgroup barrier would also tie the workgroup's OTHER tiles to this tile's pace although they
// exchange nothing (three tiles per workgroup share one copy of the weights, and gfx950 has no named barriers).  Each wave
// adds 1 after its own LDS writes have been issued (a wave's LDS operations execute in order) and spins until the count
// reaches `target` (callers advance it by n_waves per rendezvous; the counter only grows, so no reset race).
#ifdef IPLAN_HOST_EMULATION
#define IPLAN_TILE_SYNC(cnt, target) __syncthreads()            /* (every wave of the block calls it the same number of times) */
#else
__device__ __forceinline__ void iplan_tile_sync(int* cnt, int target) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63u) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
#define IPLAN_TILE_SYNC(cnt, target) iplan_tile_sync((cnt), (target))
#endif

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- split-bf16 contraction: an fp32 product out of the bf16 matrix cores ----------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate on gfx950 (64 FLOP/clk/SIMD, MI355X_MICROARCH.md "Matrix cores")
// and shares the VALU's issue time; v_mfma_f32_16x16x32_bf16 is 16x as fast and runs beside the VALU.  An fp32 value is
// EXACTLY the sum of three bf16 pieces (3 x 8 significand bits, round-to-nearest residuals: x = p0 + p1 + p2), a
// bf16 x bf16 product is exact in fp32, and the accumulator is fp32 -- so
//      a . b  =  a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)  +  O(2^-26 |a||b|)
// six bf16 MFMAs (K = 32 each) reproduce the fp32 contraction to below fp32 round-off (the dropped terms are a quarter
// of an fp32 half-ulp of the product).  Used in the recurrences, where the weights' pieces are loop invariants in
// registers and the hidden state is split once per step (5.5 VALU operations per value).
// K slots: a lane group g of the K = 32 instruction holds 8 k-values; slot j < 4 = element 4g + j of the first 16-tile,
// slot j >= 4 = element 4g + j - 4 of the second one, i.e. exactly the 8 D-layout values a lane holds of a 32-vector --
// A and B agree by construction, no data movement between steps.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Bf3 {
    bf16x8 p0, p1, p2;
};
__device__ __forceinline__ Bf3 split_bf3_scalar(f32x4 lo, f32x4 hi) {
    Bf3 s;
    for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? lo[j] : hi[j - 4];
        const __bf16 b0 = (__bf16)v;                    // v_cvt_pk_bf16_f32: round to nearest even
        const float r1 = v - (float)b0;                 // exact
        const __bf16 b1 = (__bf16)r1;
        const float r2 = r1 - (float)b1;                // exact, fits 8 bits
        s.p0[j] = b0;
        s.p1[j] = b1;
        s.p2[j] = (__bf16)r2;
    }
    return s;
}
// The same split two values at a time: ONE v_cvt_pk_bf16_f32 rounds a pair and IS the packed piece; the pair's bf16 values come back as
// floats by a shift and a mask, the residuals by one packed subtraction (v_pk_add_f32) -- 36 instead of the 60 VALU instructions the
// value-at-a-time form compiles to per K = 32 operand (it converts every value alone, and once more to pack the pieces).  Same
// roundings, exact subtractions: bit-identical pieces.  Which form is FASTER depends on the instruction stream around it (round 5,
// same-box A/Bs of whole libraries, profiles/r05_notes.md): the behaviour kernels gain (learn 15.4 -> 15.0 ms), the wide weight-gradient
// contraction LOSES 11 % although its loop shrinks from 2 012 to 1 699 instructions (packed fp32 VALU beside a dense MFMA stream), the GAT
// recurrence does not move; with plain subtractions instead of the packed one nobody gains.  So a source file opts in
// (#define IPLAN_SPLIT_PAIRS in front of its includes: behavior_learn.hip) and everybody else keeps the form above.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pack2(f32x2 v) {                      // round to nearest even, v[0] in the low half
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 bf16_unpack2(uint32_t u) {
    f32x2 r;
    r[0] = __builtin_bit_cast(float, u << 16);
    r[1] = __builtin_bit_cast(float, u & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void split_bf3_pair(f32x2 v, uint32_t& q0, uint32_t& q1, uint32_t& q2) {
    q0 = bf16_pack2(v);
    const f32x2 r1 = v - bf16_unpack2(q0);              // exact
    q1 = bf16_pack2(r1);
    const f32x2 r2 = r1 - bf16_unpack2(q1);             // exact, fits 8 bits
    q2 = bf16_pack2(r2);
}
__device__ __forceinline__ Bf3 split_bf3_pairs(f32x4 lo, f32x4 hi) {
    u32x4 q0, q1, q2;
    for (int k = 0; k < 4; ++k) {
        f32x2 v;
        v[0] = k < 2 ? lo[2 * k] : hi[2 * k - 4];
        v[1] = k < 2 ? lo[2 * k + 1] : hi[2 * k - 3];
        uint32_t a, b, c;
        split_bf3_pair(v, a, b, c);
        q0[k] = a; q1[k] = b; q2[k] = c;
    }
    Bf3 s;
    s.p0 = __builtin_bit_cast(bf16x8, q0);
    s.p1 = __builtin_bit_cast(bf16x8, q1);
    s.p2 = __builtin_bit_cast(bf16x8, q2);
    return s;
}
__device__ __forceinline__ Bf3 split_bf3(f32x4 lo, f32x4 hi) {
#ifdef IPLAN_SPLIT_PAIRS
    return split_bf3_pairs(lo, hi);
#else
    return split_bf3_scalar(lo, hi);
#endif
}
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// A fragment (three pieces) of rows o0..o0+15, columns k0..k0+31 of a row-major matrix in the slot order above; rows
// past `rows` read as zero.  16-byte aligned base, ld % 4 == 0.
__device__ __forceinline__ Bf3 wfrag_bf3(const float* __restrict__ W, int ld, int rows, int o0, int k0) {
    const int l = lane_id();
    const int r = o0 + (l & 15);
    const int rc = r < rows ? r : rows - 1;
    const float* p = W + (size_t)rc * ld + k0 + 4 * (l >> 4);
    f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 16);
    if (r >= rows) lo = hi = f32x4{0.f, 0.f, 0.f, 0.f};
    return split_bf3(lo, hi);
}

// ... of c * W (c = 1: the same pieces as wfrag_bf3)
__device__ __forceinline__ Bf3 wfrag_bf3_scaled(const float* __restrict__ W, int ld, int rows, int o0, int k0, float c) {
    const int l = lane_id();
    const int r = o0 + (l & 15);
    const int rc = r < rows ? r : rows - 1;
    const float* p = W + (size_t)rc * ld + k0 + 4 * (l >> 4);
    f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 16);
    if (r >= rows) lo = hi = f32x4{0.f, 0.f, 0.f, 0.f};
    return split_bf3(lo * c, hi * c);
}

// Same for the TRANSPOSE of a row-major matrix (A operand of y = W^T x): output rows o0..o0+15 of W^T = columns of W, K slots
// k0..k0+31 = rows of W (full tiles only; strided loads, done once per launch).
__device__ __forceinline__ Bf3 wfrag_t_bf3(const float* __restrict__ W, int ld, int o0, int k0) {
    const int l = lane_id();
    const float* p = W + (size_t)(k0 + 4 * (l >> 4)) * ld + o0 + (l & 15);
    f32x4 lo, hi;
    for (int j = 0; j < 4; ++j) { lo[j] = p[(size_t)j * ld]; hi[j] = p[(size_t)(16 + j) * ld]; }
    return split_bf3(lo, hi);
}

__device__ __forceinline__ f32x4 splat4(float v) {
    f32x4 r = {v, v, v, v};
    return r;
}

// Gate non-linearities.  Default: the hardware transcendental pipe (v_exp_f32 = 2^x, v_rcp_f32, both
// ~1 ulp), 4-5 VALU instructions per value instead of the ~25-40 of libm's expf/tanhf -- the gate math
// is otherwise co-dominant with the MFMA chain in every GRU step.  Absolute error <= ~2e-7, the same
// size as fp32 round-off of the reference itself (parity tests run on these).  -DIPLAN_EXACT_GATES
// restores libm.
#ifdef IPLAN_EXACT_GATES
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_f(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.8853900817779268f * x) + 1.0f);
}
#endif

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    f32x4 r;
    for (int q = 0; q < 4; ++q) r[q] = v[q] > 0.0f ? v[q] : 0.0f;
    return r;
}

// ---- A-operand fragments ----------------------------------------------------------------------
// Fragment of a row-major [rows x cols] matrix (leading dimension ld): lane (m,g) gets
// W[o0+m][k0+4g .. k0+4g+3]; out-of-range elements read as 0 (this is how odd dims are padded).
__device__ __forceinline__ f32x4 wfrag(const float* __restrict__ W, int ld, int rows, int cols,
                                       int o0, int k0) {
    const int l = lane_id();
    const int r = o0 + (l & 15);
    const int c = k0 + 4 * (l >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
        const float* p = W + (size_t)r * ld + c;
        if (c + 3 < cols && ((ld & 3) == 0) && ((((size_t)p) & 15) == 0)) {
            v = *reinterpret_cast<const f32x4*>(p);
        } else {
            for (int q = 0; q < 4; ++q)
                if (c + q < cols) v[q] = p[q];
        }
    }
    return v;
}

// Fast path of wfrag for the common case -- base 16-byte aligned, ld % 4 == 0, the 16 columns k0..k0+15 inside the
// matrix: ONE unconditional 16-byte load (rows past the end are clamped and zeroed by a select), so the compiler is
// free to batch many fragment loads instead of serialising load -> wait -> MFMA through alignment branches.
__device__ __forceinline__ f32x4 wfrag_a(const float* __restrict__ W, int ld, int rows, int o0, int k0) {
    const int l = lane_id();
    const int r = o0 + (l & 15);
    const int rc = r < rows ? r : rows - 1;
    f32x4 v = *reinterpret_cast<const f32x4*>(W + (size_t)rc * ld + k0 + 4 * (l >> 4));
    if (r >= rows) v = f32x4{0.f, 0.f, 0.f, 0.f};
    return v;
}

// Same for the transposed fragment (full tiles only: rows k0..k0+15 and columns o0..o0+15 exist).
__device__ __forceinline__ f32x4 wfrag_ta(const float* __restrict__ W, int ld, int o0, int k0) {
    const int l = lane_id();
    const float* p = W + (size_t)(k0 + 4 * (l >> 4)) * ld + o0 + (l & 15);
    f32x4 v;
    v[0] = p[0]; v[1] = p[ld]; v[2] = p[2 * ld]; v[3] = p[3 * ld];
    return v;
}

// Bias tile when 16 t + 15 < dim and b is 16-byte aligned.
__device__ __forceinline__ f32x4 bfrag_a(const float* __restrict__ b, int t) {
    return *reinterpret_cast<const f32x4*>(b + 16 * t + 4 * (lane_id() >> 4));
}

// Fragment of the TRANSPOSE of a row-major [rows x cols] matrix: lane (m,g) gets
// W[k0+4g+q][o0+m], q<4  (A operand of  y = W^T x).
__device__ __forceinline__ f32x4 wfrag_t(const float* __restrict__ W, int ld, int rows, int cols,
                                         int o0, int k0) {
    const int l = lane_id();
    const int c = o0 + (l & 15);
    const int r = k0 + 4 * (l >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < cols)
        for (int q = 0; q < 4; ++q)
            if (r + q < rows) v[q] = W[(size_t)(r + q) * ld + c];
    return v;
}

// ---- D-layout vectors ---------------------------------------------------------------------------
// Bias / per-feature vector in D layout: lane (n,g) gets b[16t+4g .. +3] (same for all chains).
__device__ __forceinline__ f32x4 bfrag(const float* __restrict__ b, int dim, int t) {
    const int c = 16 * t + 4 * (lane_id() >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 4; ++q)
        if (c + q < dim) v[q] = b[c + q];
    return v;
}

// Pointers that went through a local array (indexed at run time -> scratch) come back as GENERIC pointers: loads through
// them are flat_load, which counts on vmcnt AND lgkmcnt and forces `s_waitcnt vmcnt(0) lgkmcnt(0)` before any use -- a
// software-pipelined ring of loads is then drained completely every round.  Everything the kernels read is global memory.
#ifdef IPLAN_HOST_EMULATION
#define IPLAN_GLOBAL_AS
#else
#define IPLAN_GLOBAL_AS __attribute__((address_space(1)))
#endif
template <class T>
__device__ __forceinline__ const IPLAN_GLOBAL_AS T* as_global(const T* p) { return (const IPLAN_GLOBAL_AS T*)p; }

// Zeroing a loaded value is a bitwise AND with an all-ones / zero lane mask, NOT `cond ? loaded : 0`: the compiler turns a
// select whose operand is a load back into a branch around the load (and then waits for it on the spot).
__device__ __forceinline__ float keep_if(bool ok, float v) {
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & (ok ? 0xFFFFFFFFu : 0u));
}
__device__ __forceinline__ f32x4 zero_unless(bool ok, f32x4 v) {
    const uint32_t m = ok ? 0xFFFFFFFFu : 0u;
    f32x4 r;
    for (int q = 0; q < 4; ++q) r[q] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, (float)v[q]) & m);
    return r;
}
// A wave-uniform value the compiler cannot prove uniform (anything derived from threadIdx.x, like the wave index): through
// an SGPR.  Branches and loop bounds on it become scalar branches -- with a "divergent" condition every load inside turns
// into an exec-masked block and the wait counters are drained at each join.
__device__ __forceinline__ int uniform_i(int v) {
#ifdef IPLAN_HOST_EMULATION
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// Load tile t of a per-chain vector from a row pointer (one row per chain); `valid` = chain exists.
__device__ __forceinline__ f32x4 vload(const float* __restrict__ row, bool valid, int dim, int t) {
    const int c = 16 * t + 4 * (lane_id() >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        if (c + 3 < dim && ((((size_t)(row + c)) & 15) == 0)) {
            v = *reinterpret_cast<const f32x4*>(row + c);
        } else {
            for (int q = 0; q < 4; ++q)
                if (c + q < dim) v[q] = row[c + q];
        }
    }
    return v;
}

__device__ __forceinline__ void vstore(float* __restrict__ row, bool valid, int dim, int t, f32x4 v) {
    const int c = 16 * t + 4 * (lane_id() >> 4);
    if (valid) {
        if (c + 3 < dim && ((((size_t)(row + c)) & 15) == 0)) {
            *reinterpret_cast<f32x4*>(row + c) = v;
        } else {
            for (int q = 0; q < 4; ++q)
                if (c + q < dim) row[c + q] = v[q];
        }
    }
}

// Device-coherent accesses (agent scope, relaxed): a value written by one workgroup of a launch and read by another of the SAME
// launch -- possibly behind another XCD's L2 -- goes around the non-coherent caches (sc1 stores write through, sc1 loads do not
// hit stale lines), so that neither side needs a cache write-back / invalidate (the fused rollout launch, ac_fwd_body.h).
#ifndef IPLAN_FUSED_FENCES
#define IPLAN_FUSED_FENCES 0          // 1 (A/B builds): plain stores / loads + one agent-scope release / acquire fence per workgroup
#endif
__device__ __forceinline__ void coh_store(float* p, float v) {
#if defined(IPLAN_HOST_EMULATION) || IPLAN_FUSED_FENCES
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ float coh_load(const float* p) {
#if defined(IPLAN_HOST_EMULATION) || IPLAN_FUSED_FENCES
    return *p;
#else
    return __hip_atomic_load(as_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// 16 bytes as ONE load instruction that the compiler does not take for an atomic: it gives every returning atomic (which is what
// coh_load is to it) a full `s_waitcnt vmcnt(0)`, which turned an operand ring of four coh_load's per k-tile into one memory
// round trip per k-tile whatever its depth (profiles/r04_notes.md).  The instruction is invisible to the compiler's wait-count bookkeeping, so the CALLER guarantees the
// wait: loads return in order, hence the value is there once a compiler-visible load issued AFTER this one has been waited for
// (ac_fwd_body.h: the weight fragments of the same ring slot); the compiler's own counts can only come out too strict, never
// too loose.  16-byte aligned address.
__device__ __forceinline__ f32x4 coh_load16_untracked(const float* p) {
#if defined(IPLAN_HOST_EMULATION) || IPLAN_FUSED_FENCES
    return *reinterpret_cast<const f32x4*>(p);
#else
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
#endif
}
// ... and its store counterpart (16-byte aligned address; the caller drains vmcnt before it publishes)
__device__ __forceinline__ void coh_store16(float* p, f32x4 v) {
#if defined(IPLAN_HOST_EMULATION) || IPLAN_FUSED_FENCES
    *reinterpret_cast<f32x4*>(p) = v;
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#endif
}
// make `v` (loaded by coh_load16_untracked) usable only after `dep` -- a compiler-visible load issued after it -- has arrived
__device__ __forceinline__ f32x4 after_load(f32x4 v, float dep) {
#if !defined(IPLAN_HOST_EMULATION)
    asm volatile("" : "+v"(v) : "v"(dep));
#endif
    return v;
}
// vstore with device-coherent stores (COH) or plain ones
template <bool COH>
__device__ __forceinline__ void vstore_c(float* __restrict__ row, bool valid, int dim, int t, f32x4 v) {
    if (!COH) { vstore(row, valid, dim, t, v); return; }
    const int c = 16 * t + 4 * (lane_id() >> 4);
    if (valid) {
        if (c + 3 < dim && ((((size_t)(row + c)) & 15) == 0)) {
            coh_store16(row + c, v);
        } else {
            for (int q = 0; q < 4; ++q)
                if (c + q < dim) coh_store(row + c + q, v[q]);
        }
    }
}

// Aligned whole-tile variants (row 16-byte aligned, tile t entirely inside the vector): one predicated 16-byte
// access, none of the per-lane alignment / tail branching of vload / vstore.
__device__ __forceinline__ f32x4 vload_a(const float* __restrict__ row, bool valid, int t) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid) v = *reinterpret_cast<const f32x4*>(row + 16 * t + 4 * (lane_id() >> 4));
    return v;
}
__device__ __forceinline__ void vstore_a(float* __restrict__ row, bool valid, int t, f32x4 v) {
    if (valid) *reinterpret_cast<f32x4*>(row + 16 * t + 4 * (lane_id() >> 4)) = v;
}

// acc += W[o0.., k0..k0+15] . x   (one 16x16 weight block against tile T of the input vector)
__device__ __forceinline__ f32x4 mma_block(f32x4 w, f32x4 x, f32x4 acc) {
    acc = mfma4(w[0], x[0], acc);
    acc = mfma4(w[1], x[1], acc);
    acc = mfma4(w[2], x[2], acc);
    acc = mfma4(w[3], x[3], acc);
    return acc;
}

// Sum over the 4 lane-groups g (lanes n, n+16, n+32, n+48): every lane ends with the total.
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// Sum / max over all 64 lanes.
__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// GRU gate math on one 4-element slice (PyTorch gate order r,z,n; b_hn inside the r* term).
struct GruGates {
    f32x4 r, z, n, hn, h;
};
__device__ __forceinline__ GruGates gru_gates(f32x4 pre_r, f32x4 pre_z, f32x4 gi_n, f32x4 gh_n, f32x4 h_prev) {
    GruGates o;
    for (int q = 0; q < 4; ++q) {
        o.r[q] = sigmoid_f(pre_r[q]);
        o.z[q] = sigmoid_f(pre_z[q]);
        o.hn[q] = gh_n[q];
        o.n[q] = tanh_f(gi_n[q] + o.r[q] * gh_n[q]);
        o.h[q] = (1.0f - o.z[q]) * o.n[q] + o.z[q] * h_prev[q];
    }
    return o;
}

// The same step with the gates' exp2 constants already folded into the pre-activations: pre_r, pre_z = -log2e x (r / z gate), gi_n and
// gh_n = 2 log2e x (both parts of the n gate; o.hn is the SCALED value -- callers that store it use gru_gates).
__device__ __forceinline__ GruGates gru_gates_folded(f32x4 pre_r, f32x4 pre_z, f32x4 gi_n, f32x4 gh_n, f32x4 h_prev) {
    GruGates o;
    for (int q = 0; q < 4; ++q) {
#ifdef IPLAN_EXACT_GATES
        o.r[q] = 1.0f / (1.0f + exp2f(pre_r[q]));
        o.z[q] = 1.0f / (1.0f + exp2f(pre_z[q]));
        o.n[q] = 1.0f - 2.0f / (exp2f(gi_n[q] + o.r[q] * gh_n[q]) + 1.0f);
#else
        o.r[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_r[q]));
        o.z[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre_z[q]));
        o.n[q] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(gi_n[q] + o.r[q] * gh_n[q]) + 1.0f);
#endif
        o.hn[q] = gh_n[q];
        o.h[q] = (1.0f - o.z[q]) * o.n[q] + o.z[q] * h_prev[q];
    }
    return o;
}

}  // namespace iplan
