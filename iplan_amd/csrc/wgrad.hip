// Weight-gradient contraction  dW[o][k] = sum_rows dY[row][o] * X[row][k],  db[o] = sum_rows dY[row][o]
// for every dense / GRU weight on the path, all nets and up to IPLAN_WGRAD_MAX problems per launch.
//
// The backward kernels (GAT, behaviour/prediction GRUs, actor/critic tail) store the row-level
// pre-activation gradients dY and keep the forward activations X; this kernel is the one real dense
// contraction of the backward pass and runs on v_mfma_f32_16x16x4_f32:
//      D[i = o][j = k] += A[i][kk] * B[kk][j],   A[i][kk] = dY[row kk][o0+i],  B[kk][j] = X[row kk][k0+j]
// i.e. both operands are read in their natural row-major layout (16 consecutive floats of a row per
// 16-lane group, 4 rows per MFMA) -- no transposes.  Rows are indexed (outer, inner) so that a
// recurrent weight's X operand (the previous hidden state) is the saved hidden sequence shifted by
// one inner step, with an optional initial-state row.
// Reduction order is fixed: a wave owns a contiguous row range ("virtual chunk"), partial tiles go
// to the workspace and a second kernel adds the chunks in index order -> bitwise reproducible, no
// atomics.
//
// The contraction streams every row once per wave job, so it is HBM-bound by the bytes its jobs REQUEST: waves
// that read the same rows drift apart by far more than the L2 a streaming wave can count on, i.e. sharing
// through the cache does not happen (round 4: the two wide jobs of a GRU -- same dY rows -- as the two waves of one workgroup,
// i.e. on one CU and at one pace: wide kernel 2.84 -> 2.79 ms, cycle unchanged; not kept).  Hence two job shapes:
//   wide   12 x 4 tiles (192 accumulator registers, one wave per SIMD): a GRU weight [192 x 64] is ONE job --
//          its dY rows and its X rows are read exactly once.  Since round 3 these jobs run on the bf16 matrix cores
//          (wgrad_partial_bf16_kernel below: fp32-exact split-bf16 products, 32-row blocks); the fp32 kernel keeps the
//          problems with an initial-state operand (x0) -- there the 192 MFMAs of a 16-row block (2.6 us) cover the
//          latency of the next block's loads
//   narrow  4 x 4 tiles, three waves per SIMD: the small heads (a few MFMAs per block) are latency
//          bound and get their parallelism from loads in flight and several waves per SIMD instead
#include "api_util.h"
#include "wave_tile.h"

namespace iplan {

constexpr int WG_TK = 4;            // k-tiles per wave job (both shapes)
#ifndef IPLAN_WG_TO_WIDE
#define IPLAN_WG_TO_WIDE 12
#endif
#ifndef IPLAN_WG_WIDE_RB
#define IPLAN_WG_WIDE_RB 1
#endif
#ifndef IPLAN_WG_WIDE_ROUNDS_MAX
#define IPLAN_WG_WIDE_ROUNDS_MAX 1024
#endif
#ifndef IPLAN_WG_WIDE_SLOTS
#define IPLAN_WG_WIDE_SLOTS 1024    // wave slots one round of wide jobs fills (1024 SIMDs x resident wide waves per SIMD)
#endif
#ifndef IPLAN_WG_MIN_ROWS
#define IPLAN_WG_MIN_ROWS 64        // shortest row chunk of a wave job.  256 until round 4: a data-parallel rank's 2 880 PPO rows were then
#endif                              // 12 chunks per problem -- a handful of long waves per launch; 64 (45 chunks): rank-of-8 step 41.9 -> 40.3 ms,
                                    // config 3 unchanged (272 ms); 32: no better at 2 880 rows, worse at 22 950 (profiles/r04_notes.md)
constexpr int WG_TO_WIDE = IPLAN_WG_TO_WIDE;      // o-tiles per wave job: wide shape (problems with more than 8 o-tiles) ...
constexpr int WG_TO_NARROW = 4;     // ... and narrow shape

struct WgradGeom {
    int OT, KT, to, n_og, n_kg, jobs;
    int64_t rows;
    int vchunks, vrows;                      // virtual chunks and rows per virtual chunk (multiple of 16)
    int64_t part_floats;                     // floats per virtual chunk: OT*16 * (KT*16 + 1)
};

// chunks_wide: row chunks per net of the wide problems of this launch (the narrow ones use IPLAN_WGRAD_MAX_CHUNKS)
__host__ __device__ inline WgradGeom wgrad_geom(const IplanWgradProblem& p, int chunks_wide) {
    WgradGeom g;
    g.OT = (p.O + 15) / 16;
    g.KT = (p.K + 15) / 16;
    g.to = g.OT > 2 * WG_TO_NARROW ? WG_TO_WIDE : WG_TO_NARROW;        // wide from 9 o-tiles on (at most 25 % idle accumulators)
    g.n_og = (g.OT + g.to - 1) / g.to;
    g.n_kg = g.KT > 0 ? (g.KT + WG_TK - 1) / WG_TK : 1;
    g.jobs = g.n_og * g.n_kg;
    g.rows = (int64_t)p.n_outer * p.n_inner;
    // thin problems (one tile along O or K: a few MFMAs per 16 rows) are latency bound and their partial tiles are tiny:
    // 8x more, shorter row chunks give them the waves in flight that hide the load latency
    const bool thin = g.to != WG_TO_WIDE && (g.KT <= 1 || g.OT == 1);
    const int target = g.to == WG_TO_WIDE ? chunks_wide : (thin ? 8 * IPLAN_WGRAD_MAX_CHUNKS : IPLAN_WGRAD_MAX_CHUNKS);
    int64_t vr = (g.rows + target - 1) / target;
    if (vr < IPLAN_WG_MIN_ROWS) vr = IPLAN_WG_MIN_ROWS;
    vr = (vr + 15) / 16 * 16;
    g.vrows = (int)vr;
    g.vchunks = (int)((g.rows + vr - 1) / vr);
    if (g.vchunks < 1) g.vchunks = 1;
    g.part_floats = (int64_t)g.OT * 16 * (g.KT * 16 + 1);
    return g;
}

__device__ __forceinline__ int ocol(const IplanWgradProblem& p, int o) {
    return o < p.seg_split ? p.seg_c0 + o : p.seg_c1 + (o - p.seg_split);
}
// element offset of column c inside its row: plain, or column-grouped (16-column groups cg_stride elements apart)
__device__ __forceinline__ uint32_t colmap(int c, int cg_stride) {
    return cg_stride ? (uint32_t)(c >> 4) * (uint32_t)cg_stride + (uint32_t)(c & 15) : (uint32_t)c;
}
// rows in front of an outer index's first row that the shifted X operand may read in place (IplanWgradProblem.x_pre_valid)
__device__ __forceinline__ int x_pre_rows(const IplanWgradProblem& p) { return (p.x_pre_valid && p.x_shift < 0) ? -p.x_shift : 0; }

// jobs of one shape in a launch
constexpr int WG_MAX_JOBS = 64;
struct WgradJobs {
    int n;
    int pj[WG_MAX_JOBS];                     // problem << 8 | job  (dwords: scalar loads straight from the kernel arguments)
};

// grid: (job of this shape, virtual chunk, net); block: 64 (one wave).
//
// Data path: the operands are loaded from global memory DIRECTLY in MFMA operand order -- for sub-step s of a
// 16-row block lane (i, g) fetches dY[row 4s+g][o0+i] and X[row 4s+g][k0+i], i.e. the 16 lanes of a group read
// the 64 contiguous bytes of a row tile and a load instruction covers 4 rows.  No LDS, no transposes, no barriers.
// Prefetch is rolling: as soon as the MFMAs of a sub-step are issued its operand registers are reloaded with the
// same sub-step RB blocks ahead.  Bias gradients are lane-local column sums (VALU), reduced across the four
// lane groups at the end.
template <int TO, int TK, int RB>
__global__ __launch_bounds__(64) void wgrad_partial_kernel(IplanWgradArgs a, WgradJobs jl, int chunks_wide) {
    const int pj = jl.pj[blockIdx.x], pi = pj >> 8, job = pj & 255, net = (int)blockIdx.z;
    const IplanWgradProblem& p = a.p[pi];
    const WgradGeom gm = wgrad_geom(p, chunks_wide);
    const int vc = (int)blockIdx.y;
    if (vc >= gm.vchunks) return;
    const int og = job / gm.n_kg, kg = job % gm.n_kg;
    const int l = lane_id(), i = l & 15, g = l >> 4;
    const int ot0 = og * (TO == 1 ? WG_TO_NARROW : TO), kt0 = kg * WG_TK;        // (job origin: the geometry's tile counts)
    const int not_ = imin(TO, gm.OT - ot0), nkt = gm.KT > 0 ? imin(TK, gm.KT - kt0) : 0;
    const bool want_bias = (kg == 0);

    f32x4 acc[TO][TK];
    float bsum[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        bsum[t] = 0.f;
#pragma unroll
        for (int u = 0; u < TK; ++u) acc[t][u] = splat4(0.f);
    }
    const int64_t r_lo = (int64_t)vc * gm.vrows;
    const int64_t r_hi = gm.rows < r_lo + gm.vrows ? gm.rows : r_lo + gm.vrows;
    // Addressing: one uniform base per operand (the chunk's first outer index) + 32-bit per-lane BYTE offsets, so a
    // load costs one v_add_u32 and a global_load with scalar base.  The lane's column of every operand tile is
    // clamped into the matrix: an out-of-range column only feeds accumulator rows / columns that the reduction never
    // reads (element (o, k) of the product depends on column o of dY and column k of X alone).
    const int o_lo = (int)(r_lo / p.n_inner);
    const char* __restrict__ abase = reinterpret_cast<const char*>(p.dy + (int64_t)net * p.dy_s_net + (int64_t)o_lo * p.dy_s_outer);
    // (pre = rows in front of inner index 0 that exist in memory: the base moves back by them, the offsets forward, so that the
    // 32-bit byte offsets stay non-negative)
    const int pre = x_pre_rows(p);
    const char* __restrict__ xbase = p.x ? reinterpret_cast<const char*>(p.x + (int64_t)net * p.x_s_net + (int64_t)o_lo * p.x_s_outer - (int64_t)pre * p.x_s_inner) : nullptr;
    const float* __restrict__ x0 = p.x0 ? p.x0 + (int64_t)net * p.x0_s_net + (int64_t)o_lo * p.x0_s_outer : nullptr;
    uint32_t acolb[TO], bcolb[TK];
#pragma unroll
    for (int t = 0; t < TO; ++t) acolb[t] = 4u * colmap(ocol(p, imin((ot0 + imin(t, not_ - 1)) * 16 + i, p.O - 1)), p.dy_cg_stride);
#pragma unroll
    for (int u = 0; u < TK; ++u) bcolb[u] = nkt > 0 ? 4u * colmap(p.x_col0 + imin((kt0 + imin(u, nkt - 1)) * 16 + i, p.K - 1), p.x_cg_stride) : 0u;
    // cursors of the rows this lane loads next, one per (block slot j, sub-step s): rows 16j + 4s + g of the block,
    // advanced by 16 * RB rows per reload without divisions
    int ri[RB][4], orel[RB][4];
    uint32_t aoff[RB][4], xoff[RB][4];
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t r = r_lo + 16 * j + 4 * s + g;
            const int ro = (int)(r / p.n_inner);
            ri[j][s] = (int)(r - (int64_t)ro * p.n_inner);
            orel[j][s] = ro - o_lo;
            aoff[j][s] = (uint32_t)(4 * ((int64_t)orel[j][s] * p.dy_s_outer + (int64_t)ri[j][s] * p.dy_s_inner));
            xoff[j][s] = (uint32_t)(4 * ((int64_t)orel[j][s] * p.x_s_outer + (int64_t)(ri[j][s] + p.x_shift + pre) * p.x_s_inner));
        }
    const int adv_o = (16 * RB) / p.n_inner, adv_i = (16 * RB) % p.n_inner;
    const uint32_t a_adv = (uint32_t)(4 * ((int64_t)adv_o * p.dy_s_outer + (int64_t)adv_i * p.dy_s_inner));
    const uint32_t x_adv = (uint32_t)(4 * ((int64_t)adv_o * p.x_s_outer + (int64_t)adv_i * p.x_s_inner));
    const uint32_t a_wrap = (uint32_t)(4 * (p.dy_s_outer - (int64_t)p.n_inner * p.dy_s_inner));
    const uint32_t x_wrap = (uint32_t)(4 * (p.x_s_outer - (int64_t)p.n_inner * p.x_s_inner));
    const bool shifted = p.x_shift != 0;

    float av[RB][4][TO], bv[RB][4][TK];
    // rb = first row of the block that goes to slot j; tail = the block may reach past r_hi (uniform)
    auto fetch = [&](int64_t rb, int j, int s, bool tail) {
        const int inner = ri[j][s], outer_rel = orel[j][s];
        uint32_t ao = aoff[j][s], xo = xoff[j][s];
        ri[j][s] += adv_i;
        orel[j][s] += adv_o;
        aoff[j][s] += a_adv;
        xoff[j][s] += x_adv;
        if (ri[j][s] >= p.n_inner) { ri[j][s] -= p.n_inner; orel[j][s] += 1; aoff[j][s] += a_wrap; xoff[j][s] += x_wrap; }
        const bool rv = !tail || rb + 4 * s + g < r_hi;
        const bool inr = !shifted || (unsigned)(inner + p.x_shift + pre) < (unsigned)(p.n_inner + pre);
        if (tail) { ao = rv ? ao : 0u; }
        if (tail || shifted) { xo = (rv && inr) ? xo : 0u; }
#pragma unroll
        for (int t = 0; t < TO; ++t) av[j][s][t] = *reinterpret_cast<const float*>(abase + (ao + acolb[t]));
        if (nkt > 0) {
#pragma unroll
            for (int u = 0; u < TK; ++u) bv[j][s][u] = *reinterpret_cast<const float*>(xbase + (xo + bcolb[u]));
        } else {
#pragma unroll
            for (int u = 0; u < TK; ++u) bv[j][s][u] = 0.f;
        }
        if (tail && !rv) {
#pragma unroll
            for (int t = 0; t < TO; ++t) av[j][s][t] = 0.f;
#pragma unroll
            for (int u = 0; u < TK; ++u) bv[j][s][u] = 0.f;
        }
        if (shifted && rv && !inr) {                         // the recurrent operand's step before the first one
#pragma unroll
            for (int u = 0; u < TK; ++u) bv[j][s][u] = x0 ? x0[(int64_t)outer_rel * p.x0_s_outer + (bcolb[u] >> 2) - p.x_col0] : 0.f;
        }
    };
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) fetch(r_lo + 16 * j, j, s, r_lo + 16 * (j + 1) > r_hi);
    for (int64_t rb = r_lo; rb < r_hi; rb += 16 * RB) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int64_t nb = rb + 16 * (RB + j);           // the block that takes over slot j
            const bool tail = nb + 16 > r_hi;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // all TO x TK products, unguarded: tiles past the job's edge repeat the last valid tile (clamped
                // columns) and their accumulators are simply never written out -- no branches between the MFMAs
#pragma unroll
                for (int t = 0; t < TO; ++t) {
#pragma unroll
                    for (int u = 0; u < TK; ++u) acc[t][u] = mfma4(av[j][s][t], bv[j][s][u], acc[t][u]);
                    bsum[t] += av[j][s][t];
                }
                // rolling prefetch: the registers of this sub-step are free again
                fetch(nb, j, s, tail);
            }
        }
    }
    // partial tile layout: part[vc][o (OT*16)][KT*16 + 1]; D layout: lane (col j = i, row = 4g + q)
    float* __restrict__ part = a.workspace + p.ws_off + ((int64_t)net * gm.vchunks + vc) * gm.part_floats;
    const int ldp = gm.KT * 16 + 1;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (t >= not_) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = (ot0 + t) * 16 + 4 * g + q;
#pragma unroll
            for (int u = 0; u < TK; ++u)
                if (u < nkt) part[(int64_t)o * ldp + (kt0 + u) * 16 + i] = acc[t][u][q];
        }
        if (want_bias) {
            float b = bsum[t];
            b += __shfl_xor(b, 16);
            b += __shfl_xor(b, 32);
            if (g == 0) part[(int64_t)((ot0 + t) * 16 + i) * ldp + gm.KT * 16] = b;
        }
    }
}

// The wide jobs on the bf16 matrix cores: same job shape, row chunks, addressing and partial-tile layout as the fp32 kernel
// above, but a block is 32 rows (the K of v_mfma_f32_16x16x32_bf16: lane (i, g) holds rows 8 g + j, j < 8, of column i of
// every operand tile), every loaded value is split into its three bf16 pieces in registers (x = p0 + p1 + p2 exactly) and the
// six largest piece products are accumulated in fp32 -- the fp32 contraction to fp32 round-off at 6/16 of its matrix-pipe
// time (a [192 x 64] GRU weight over 1.4 M rows per net is 2.2 ms of v_mfma_f32_16x16x4_f32 issue, the floor of the fp32 form).
// Rolling prefetch per operand tile: as soon as a tile's raw values are split its registers are reloaded with the next block.
// NO control flow around the loads (the compiler drains the memory counter at every join behind one: a branch per loaded row for
// the recurrent operand's initial-state rows made this kernel 5x slower than the fp32 form): masks are bitwise ANDs, and
// problems WITH an initial-state operand (x0) stay on the fp32 kernel (iplan_wgrad routes them; a second always-executed load
// per X value for those rare rows cost more than the matrix time it saved: behaviour learn with in-line pieces 21.7 -> 28 ms).
template <int TO, int TK>
__global__ __launch_bounds__(64) void wgrad_partial_bf16_kernel(IplanWgradArgs a, WgradJobs jl, int chunks_wide) {
    const int pj = jl.pj[blockIdx.x], pi = pj >> 8, job = pj & 255, net = (int)blockIdx.z;
    const IplanWgradProblem& p = a.p[pi];
    const WgradGeom gm = wgrad_geom(p, chunks_wide);
    const int vc = (int)blockIdx.y;
    if (vc >= gm.vchunks) return;
    const int og = job / gm.n_kg, kg = job % gm.n_kg;
    const int l = lane_id(), i = l & 15, g = l >> 4;
    const int ot0 = og * TO, kt0 = kg * WG_TK;
    const int not_ = imin(TO, gm.OT - ot0), nkt = gm.KT > 0 ? imin(TK, gm.KT - kt0) : 0;
    const bool want_bias = (kg == 0);

    f32x4 acc[TO][TK];
    float bsum[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        bsum[t] = 0.f;
#pragma unroll
        for (int u = 0; u < TK; ++u) acc[t][u] = splat4(0.f);
    }
    const int64_t r_lo = (int64_t)vc * gm.vrows;
    const int64_t r_hi = gm.rows < r_lo + gm.vrows ? gm.rows : r_lo + gm.vrows;
    const int o_lo = (int)(r_lo / p.n_inner);
    const char* __restrict__ abase = reinterpret_cast<const char*>(p.dy + (int64_t)net * p.dy_s_net + (int64_t)o_lo * p.dy_s_outer);
    // (a bias-only problem has no X: its B loads read dY's first row instead and are masked to zero)
    const bool has_x = nkt > 0 && p.x != nullptr;
    const int pre = x_pre_rows(p);
    const char* __restrict__ xbase = has_x ? reinterpret_cast<const char*>(p.x + (int64_t)net * p.x_s_net + (int64_t)o_lo * p.x_s_outer - (int64_t)pre * p.x_s_inner) : abase;
    uint32_t acolb[TO], bcolb[TK];
#pragma unroll
    for (int t = 0; t < TO; ++t) acolb[t] = 4u * colmap(ocol(p, imin((ot0 + imin(t, not_ - 1)) * 16 + i, p.O - 1)), p.dy_cg_stride);
#pragma unroll
    for (int u = 0; u < TK; ++u) {
        const int kc = has_x ? imin((kt0 + imin(u, nkt - 1)) * 16 + i, p.K - 1) : 0;
        bcolb[u] = has_x ? 4u * colmap(p.x_col0 + kc, p.x_cg_stride) : 0u;
    }
    // cursors of this lane's 8 rows of the block being LOADED (rows 8 g + j), advanced by 32 rows per block
    int ri[8];
    uint32_t aoff[8], xoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t r = r_lo + 8 * g + j;
        const int ro = (int)(r / p.n_inner);
        ri[j] = (int)(r - (int64_t)ro * p.n_inner);
        aoff[j] = (uint32_t)(4 * ((int64_t)(ro - o_lo) * p.dy_s_outer + (int64_t)ri[j] * p.dy_s_inner));
        xoff[j] = (uint32_t)(4 * ((int64_t)(ro - o_lo) * p.x_s_outer + (int64_t)(ri[j] + p.x_shift + pre) * p.x_s_inner));
    }
    const int adv_o = 32 / p.n_inner, adv_i = 32 % p.n_inner;
    const uint32_t a_adv = (uint32_t)(4 * ((int64_t)adv_o * p.dy_s_outer + (int64_t)adv_i * p.dy_s_inner));
    const uint32_t x_adv = (uint32_t)(4 * ((int64_t)adv_o * p.x_s_outer + (int64_t)adv_i * p.x_s_inner));
    const uint32_t a_wrap = (uint32_t)(4 * (p.dy_s_outer - (int64_t)p.n_inner * p.dy_s_inner));
    const uint32_t x_wrap = (uint32_t)(4 * (p.x_s_outer - (int64_t)p.n_inner * p.x_s_inner));
    const bool shifted = p.x_shift != 0;

    float ar[TO][8], br[TK][8];
    // offsets / masks of the block the raw registers are (re)loaded with, fixed by `advance` before the block's loads are issued
    uint32_t ao[8], xo[8];
    bool rvj[8], xvj[8];
    auto advance = [&](int64_t rb, bool tail) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool rv = !tail || rb + 8 * g + j < r_hi;
            const bool inr = !shifted || (unsigned)(ri[j] + p.x_shift + pre) < (unsigned)(p.n_inner + pre);
            ao[j] = rv ? aoff[j] : 0u;
            xo[j] = (rv && inr) ? xoff[j] : 0u;
            rvj[j] = rv;
            xvj[j] = rv && inr && has_x;                     // (the step before a chain's first one reads zeros: no x0 here)
            ri[j] += adv_i;
            aoff[j] += a_adv;
            xoff[j] += x_adv;
            if (ri[j] >= p.n_inner) { ri[j] -= p.n_inner; aoff[j] += a_wrap; xoff[j] += x_wrap; }
        }
    };
    auto load_b = [&](int u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            br[u][j] = keep_if(xvj[j], *reinterpret_cast<const float*>(xbase + (xo[j] + bcolb[u])));
        }
    };
    auto load_a = [&](int t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ar[t][j] = keep_if(rvj[j], *reinterpret_cast<const float*>(abase + (ao[j] + acolb[t])));
    };
    auto split8 = [&](const float (&v)[8]) {
        f32x4 lo, hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = v[q]; hi[q] = v[4 + q]; }
        return split_bf3(lo, hi);
    };
    advance(r_lo, r_lo + 32 > r_hi);
#pragma unroll
    for (int u = 0; u < TK; ++u) load_b(u);
#pragma unroll
    for (int t = 0; t < TO; ++t) load_a(t);
    for (int64_t rb = r_lo; rb < r_hi; rb += 32) {
        const int64_t nb = rb + 32;                          // the block whose loads are issued during this one
        Bf3 bp[TK];
#pragma unroll
        for (int u = 0; u < TK; ++u) bp[u] = split8(br[u]);
        advance(nb, nb + 32 > r_hi);
#pragma unroll
        for (int u = 0; u < TK; ++u) load_b(u);
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            const Bf3 ap = split8(ar[t]);
            float bs = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) bs += ar[t][j];
            bsum[t] += bs;
            load_a(t);
            // all TK products of the tile, smallest piece products first; tiles past the job's edge repeat the last valid one
            // (clamped columns) and are never written out
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = mfma_bf16(ap.p2, bp[u].p0, acc[t][u]);
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = mfma_bf16(ap.p0, bp[u].p2, acc[t][u]);
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = mfma_bf16(ap.p1, bp[u].p1, acc[t][u]);
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = mfma_bf16(ap.p1, bp[u].p0, acc[t][u]);
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = mfma_bf16(ap.p0, bp[u].p1, acc[t][u]);
#pragma unroll
            for (int u = 0; u < TK; ++u) acc[t][u] = mfma_bf16(ap.p0, bp[u].p0, acc[t][u]);
        }
    }
    float* __restrict__ part = a.workspace + p.ws_off + ((int64_t)net * gm.vchunks + vc) * gm.part_floats;
    const int ldp = gm.KT * 16 + 1;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (t >= not_) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = (ot0 + t) * 16 + 4 * g + q;
#pragma unroll
            for (int u = 0; u < TK; ++u)
                if (u < nkt) part[(int64_t)o * ldp + (kt0 + u) * 16 + i] = acc[t][u][q];
        }
        if (want_bias) {
            float b = bsum[t];
            b += __shfl_xor(b, 16);
            b += __shfl_xor(b, 32);
            if (g == 0) part[(int64_t)((ot0 + t) * 16 + i) * ldp + gm.KT * 16] = b;
        }
    }
}

// Two wide jobs that read the SAME dY rows through the same first 8 o-tiles -- the two weights of a GRU: dW_ih from [dr dz dn_i] and
// the step's input, dW_hh from [dr dz dn_h] and the previous hidden state -- as the two waves of ONE workgroup that walk the rows
// in step and fetch the shared [dr dz] tiles ONCE: wave w loads and splits shared tiles 4w .. 4w+3, publishes their bf16 pieces in
// LDS (double buffered, one barrier per 32-row block) and both waves take all eight from there; tiles 8 .. 11 (dn_i / dn_h) and
// the X operand are each wave's own.  The contraction is HBM-bound by the bytes its waves request (header), and sharing through the
// caches does not happen (round 4: the same two waves WITHOUT the exchange, 2.84 -> 2.79 ms), so this is a quarter of the bytes
// gone: 4 (128 + 64 + 64 + 64 + 64) instead of 4 x 2 x (192 + 64) per row.  Same pieces, same products, same accumulation order
// per output element as wgrad_partial_bf16_kernel<12, 4>: bit-identical partial tiles, same reduction kernel.
struct WgradPairShared {
    bf16x8 pc[2][8][3][64];                  // [block parity][shared o-tile][piece][lane]: 48 KiB
};
__global__ __launch_bounds__(128) void wgrad_pair_bf16_kernel(IplanWgradArgs a, WgradJobs jl, int chunks_wide) {
    constexpr int TO = 12, TK = WG_TK, TS = 8, TL = 8;       // o-tiles of a job, shared ones, tiles whose raw values a wave loads
    __shared__ __attribute__((aligned(16))) WgradPairShared sh;
    const int w = uniform_i(wave_id());                      // (an SGPR: the problem's fields then come by scalar loads, as in the unpaired kernel)
    const int pi = jl.pj[2 * blockIdx.x + w] >> 8, pi_other = jl.pj[2 * blockIdx.x + (w ^ 1)] >> 8, net = (int)blockIdx.z;
    const IplanWgradProblem& p = a.p[pi];
    const WgradGeom gm = wgrad_geom(p, chunks_wide);         // (the pair has ONE geometry: same O, K and rows -- iplan_wgrad checks)
    const int vc = (int)blockIdx.y;
    if (vc >= gm.vchunks) return;                            // (both waves alike)
    const int l = lane_id(), i = l & 15, g = l >> 4;

    f32x4 acc[TO][TK];
    float bsum[TL];
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int u = 0; u < TK; ++u) acc[t][u] = splat4(0.f);
#pragma unroll
    for (int s = 0; s < TL; ++s) bsum[s] = 0.f;
    const int64_t r_lo = (int64_t)vc * gm.vrows;
    const int64_t r_hi = gm.rows < r_lo + gm.vrows ? gm.rows : r_lo + gm.vrows;
    const int o_lo = (int)(r_lo / p.n_inner);
    const char* __restrict__ abase = reinterpret_cast<const char*>(p.dy + (int64_t)net * p.dy_s_net + (int64_t)o_lo * p.dy_s_outer);
    const int pre = x_pre_rows(p);
    const char* __restrict__ xbase = reinterpret_cast<const char*>(p.x + (int64_t)net * p.x_s_net + (int64_t)o_lo * p.x_s_outer - (int64_t)pre * p.x_s_inner);
    // the tiles this wave loads: slots 0 .. 3 = shared tiles 4w .. 4w+3, slots 4 .. 7 = its own tiles 8 .. 11
    uint32_t acolb[TL], bcolb[TK];
#pragma unroll
    for (int s = 0; s < TL; ++s) {
        const int t = s < 4 ? 4 * w + s : 4 + s;
        acolb[s] = 4u * colmap(ocol(p, t * 16 + i), p.dy_cg_stride);
    }
#pragma unroll
    for (int u = 0; u < TK; ++u) bcolb[u] = 4u * colmap(p.x_col0 + u * 16 + i, p.x_cg_stride);
    int ri[8];
    uint32_t aoff[8], xoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t r = r_lo + 8 * g + j;
        const int ro = (int)(r / p.n_inner);
        ri[j] = (int)(r - (int64_t)ro * p.n_inner);
        aoff[j] = (uint32_t)(4 * ((int64_t)(ro - o_lo) * p.dy_s_outer + (int64_t)ri[j] * p.dy_s_inner));
        xoff[j] = (uint32_t)(4 * ((int64_t)(ro - o_lo) * p.x_s_outer + (int64_t)(ri[j] + p.x_shift + pre) * p.x_s_inner));
    }
    const int adv_o = 32 / p.n_inner, adv_i = 32 % p.n_inner;
    const uint32_t a_adv = (uint32_t)(4 * ((int64_t)adv_o * p.dy_s_outer + (int64_t)adv_i * p.dy_s_inner));
    const uint32_t x_adv = (uint32_t)(4 * ((int64_t)adv_o * p.x_s_outer + (int64_t)adv_i * p.x_s_inner));
    const uint32_t a_wrap = (uint32_t)(4 * (p.dy_s_outer - (int64_t)p.n_inner * p.dy_s_inner));
    const uint32_t x_wrap = (uint32_t)(4 * (p.x_s_outer - (int64_t)p.n_inner * p.x_s_inner));
    const bool shifted = p.x_shift != 0;

    float ar[TL][8], br[TK][8];
    uint32_t ao[8], xo[8];
    bool rvj[8], xvj[8];
    auto advance = [&](int64_t rb, bool tail) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool rv = !tail || rb + 8 * g + j < r_hi;
            const bool inr = !shifted || (unsigned)(ri[j] + p.x_shift + pre) < (unsigned)(p.n_inner + pre);
            ao[j] = rv ? aoff[j] : 0u;
            xo[j] = (rv && inr) ? xoff[j] : 0u;
            rvj[j] = rv;
            xvj[j] = rv && inr;
            ri[j] += adv_i;
            aoff[j] += a_adv;
            xoff[j] += x_adv;
            if (ri[j] >= p.n_inner) { ri[j] -= p.n_inner; aoff[j] += a_wrap; xoff[j] += x_wrap; }
        }
    };
    // Loads are RAW (rows past the chunk's end / steps in front of a chain's first one read offset 0 -- mapped memory, any value);
    // the masks of the block a register set was loaded for are applied where the values are consumed, one block later (rvc / xvc).
    // With the mask at the load (keep_if(load), as the unpaired kernel writes it) the compiler ran every load of this kernel through
    // one temporary register behind `s_waitcnt vmcnt(0)` -- 96 serial memory round trips per block.
    bool rvc[8], xvc[8];
    auto load_b = [&](int u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) br[u][j] = *reinterpret_cast<const float*>(xbase + (xo[j] + bcolb[u]));
    };
    auto load_a = [&](int s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ar[s][j] = *reinterpret_cast<const float*>(abase + (ao[j] + acolb[s]));
    };
    auto split8 = [&](const float (&v)[8], const bool (&ok)[8], float* colsum) {
        f32x4 lo, hi;
        float bs = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float m = keep_if(ok[q], v[q]);
            bs += m;
            if (q < 4) lo[q] = m; else hi[q - 4] = m;
        }
        if (colsum) *colsum += bs;
        return split_bf3(lo, hi);
    };
    auto products = [&](const Bf3& ap, const Bf3 (&bp)[TK], f32x4 (&c)[TK]) {      // smallest piece products first (as the unpaired kernel)
#pragma unroll
        for (int u = 0; u < TK; ++u) c[u] = mfma_bf16(ap.p2, bp[u].p0, c[u]);
#pragma unroll
        for (int u = 0; u < TK; ++u) c[u] = mfma_bf16(ap.p0, bp[u].p2, c[u]);
#pragma unroll
        for (int u = 0; u < TK; ++u) c[u] = mfma_bf16(ap.p1, bp[u].p1, c[u]);
#pragma unroll
        for (int u = 0; u < TK; ++u) c[u] = mfma_bf16(ap.p1, bp[u].p0, c[u]);
#pragma unroll
        for (int u = 0; u < TK; ++u) c[u] = mfma_bf16(ap.p0, bp[u].p1, c[u]);
#pragma unroll
        for (int u = 0; u < TK; ++u) c[u] = mfma_bf16(ap.p0, bp[u].p0, c[u]);
    };
    advance(r_lo, r_lo + 32 > r_hi);
#pragma unroll
    for (int u = 0; u < TK; ++u) load_b(u);
#pragma unroll
    for (int s = 0; s < TL; ++s) load_a(s);
    int par = 0;
    for (int64_t rb = r_lo; rb < r_hi; rb += 32, par ^= 1) {
        const int64_t nb = rb + 32;
        Bf3 bp[TK];
#pragma unroll
        for (int j = 0; j < 8; ++j) { rvc[j] = rvj[j]; xvc[j] = xvj[j]; }          // masks of the block in the registers
#pragma unroll
        for (int u = 0; u < TK; ++u) bp[u] = split8(br[u], xvc, nullptr);
        advance(nb, nb + 32 > r_hi);
#pragma unroll
        for (int u = 0; u < TK; ++u) load_b(u);
        // this wave's four shared tiles: split, publish, reload
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const Bf3 ap = split8(ar[s], rvc, &bsum[s]);
            load_a(s);
            sh.pc[par][4 * w + s][0][l] = ap.p0;
            sh.pc[par][4 * w + s][1][l] = ap.p1;
            sh.pc[par][4 * w + s][2][l] = ap.p2;
        }
        // its own tiles 8 .. 11 in front of the barrier (their products do not wait for the other wave)
#pragma unroll
        for (int s = 4; s < TL; ++s) {
            const Bf3 ap = split8(ar[s], rvc, &bsum[s]);
            load_a(s);
            products(ap, bp, acc[4 + s]);
        }
        IPLAN_LDS_BARRIER();                 // LDS traffic only (__syncthreads would drain the rolling prefetch: 2.9 -> 14 ms, measured);
                                             // one barrier per block: the parity buffers make the second one unnecessary
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            Bf3 ap;
            ap.p0 = sh.pc[par][t][0][l];
            ap.p1 = sh.pc[par][t][1][l];
            ap.p2 = sh.pc[par][t][2][l];
            products(ap, bp, acc[t]);
        }
    }
    const int ldp = gm.KT * 16 + 1;
    float* __restrict__ part = a.workspace + p.ws_off + ((int64_t)net * gm.vchunks + vc) * gm.part_floats;
    float* __restrict__ part_other = a.workspace + a.p[pi_other].ws_off + ((int64_t)net * gm.vchunks + vc) * gm.part_floats;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = t * 16 + 4 * g + q;
#pragma unroll
            for (int u = 0; u < TK; ++u) part[(int64_t)o * ldp + u * 16 + i] = acc[t][u][q];
        }
    }
    // bias columns: the column sums of a shared tile were taken by the wave that loaded it and go to BOTH problems' partial tiles
#pragma unroll
    for (int s = 0; s < TL; ++s) {
        const int t = s < 4 ? 4 * w + s : 4 + s;
        float b = bsum[s];
        b += __shfl_xor(b, 16);
        b += __shfl_xor(b, 32);
        if (g == 0) {
            part[(int64_t)(t * 16 + i) * ldp + gm.KT * 16] = b;
            if (s < 4) part_other[(int64_t)(t * 16 + i) * ldp + gm.KT * 16] = b;
        }
    }
}

// grid: (ceil(O*(K+1)/256), problem * n_nets + net)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(IplanWgradArgs a, int chunks_wide) {
    const int pi = (int)blockIdx.y / a.n_nets, net = (int)blockIdx.y % a.n_nets;
    const IplanWgradProblem& p = a.p[pi];
    const WgradGeom gm = wgrad_geom(p, chunks_wide);
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int K1 = p.K + 1;
    if (idx >= p.O * K1) return;
    const int o = idx / K1, k = idx - o * K1;
    const bool is_bias = (k == p.K);
    if (is_bias ? (p.db_off < 0) : (p.dw_off < 0)) return;
    const int ldp = gm.KT * 16 + 1;
    const float* __restrict__ part = a.workspace + p.ws_off + (int64_t)net * gm.vchunks * gm.part_floats +
                                     (int64_t)o * ldp + (is_bias ? gm.KT * 16 : k);
    // fixed order: four interleaved running sums (independent loads in flight), combined at the end
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    int vc = 0;
    for (; vc + 4 <= gm.vchunks; vc += 4) {
        const float v0 = part[(int64_t)vc * gm.part_floats], v1 = part[(int64_t)(vc + 1) * gm.part_floats];
        const float v2 = part[(int64_t)(vc + 2) * gm.part_floats], v3 = part[(int64_t)(vc + 3) * gm.part_floats];
        s4[0] += v0; s4[1] += v1; s4[2] += v2; s4[3] += v3;
    }
    for (; vc < gm.vchunks; ++vc) s4[vc & 3] += part[(int64_t)vc * gm.part_floats];
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    s *= p.scale;
    float* dst = a.grad + (int64_t)net * a.grad_s_net +
                 (is_bias ? p.db_off + o : p.dw_off + (int64_t)o * p.dw_ld + p.dw_col0 + k);
    *dst = p.beta != 0.f ? p.beta * (*dst) + s : s;
}

}  // namespace iplan

extern "C" size_t iplan_wgrad_workspace_floats(const IplanWgradArgs* a) {
    using namespace iplan;
    if (!a) return 0;
    size_t total = 0;
    for (int i = 0; i < a->n_problems; ++i) {
        const WgradGeom g = wgrad_geom(a->p[i], IPLAN_WGRAD_MAX_CHUNKS);     // upper bound over the chunk targets
        total += (size_t)g.part_floats * g.vchunks * a->n_nets;
    }
    return total;
}

extern "C" int iplan_wgrad(IplanWgradArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_problems < 1 || a->n_problems > IPLAN_WGRAD_MAX || a->n_nets < 1 || !a->grad || !a->workspace)
        return fail(IPLAN_EINVAL, "iplan_wgrad: bad arguments");
    int wide_jobs = 0;
    for (int i = 0; i < a->n_problems; ++i) {
        IplanWgradProblem& p = a->p[i];
        if (!p.dy || p.O < 1 || p.O > 1024 || p.K < 0 || p.K > 1024 || (p.K > 0 && !p.x) || p.n_outer < 1 || p.n_inner < 1)
            return fail(IPLAN_EINVAL, "iplan_wgrad: problem %d has unsupported dims O=%d K=%d", i, p.O, p.K);
        if ((p.dy_cg_stride || p.x_cg_stride) && p.x0)
            return fail(IPLAN_EINVAL, "iplan_wgrad: problem %d: an initial-state operand (x0) cannot be combined with column-grouped operands", i);
        if (p.dy_cg_stride < 0 || p.x_cg_stride < 0 || (p.x_pre_valid && p.x0))
            return fail(IPLAN_EINVAL, "iplan_wgrad: problem %d: bad cg strides / x_pre_valid with x0", i);
        const WgradGeom g = wgrad_geom(p, IPLAN_WGRAD_MAX_CHUNKS);
        if (g.to == WG_TO_WIDE) wide_jobs += g.jobs;
    }
    // Wide jobs run one wave per SIMD: trim their row chunks so that the waves fill whole rounds of the 1024 SIMDs.
    int chunks_wide = IPLAN_WGRAD_MAX_CHUNKS;
    if (wide_jobs > 0) {
        const int per_chunk = wide_jobs * a->n_nets;
        const int rounds = imin(IPLAN_WG_WIDE_ROUNDS_MAX, imax(1, per_chunk * IPLAN_WGRAD_MAX_CHUNKS / IPLAN_WG_WIDE_SLOTS));
        chunks_wide = imax(1, imin(IPLAN_WGRAD_MAX_CHUNKS, rounds * IPLAN_WG_WIDE_SLOTS / per_chunk));
    }
    int64_t off = 0;
    int max_elems = 1;
    // job shapes: wide (12 x 4 tiles) for the big weights; for the rest 4 x 4, or the exact thin shapes 4 x 1 (K <= 16) and
    // 1 x 4 (O <= 16): the thin jobs are latency / issue bound, so they load and multiply only the tiles they own
    enum { J_WIDE = 0, J_SQUARE, J_THIN_K, J_THIN_O, J_KINDS };
    WgradJobs jl[J_KINDS];
    int vcs[J_KINDS];
    for (int k = 0; k < J_KINDS; ++k) { jl[k].n = 0; vcs[k] = 1; }
    for (int i = 0; i < a->n_problems; ++i) {
        IplanWgradProblem& p = a->p[i];
        const WgradGeom g = wgrad_geom(p, chunks_wide);
        p.ws_off = off;
        off += g.part_floats * g.vchunks * a->n_nets;
        if (p.O * (p.K + 1) > max_elems) max_elems = p.O * (p.K + 1);
        // the kernel addresses a row chunk with 32-bit byte offsets from the chunk's first outer index
        const int64_t outers = g.vrows / p.n_inner + 2;
        const int64_t span_dy = 4 * (outers * llabs(p.dy_s_outer) + (int64_t)(p.n_inner + 1) * llabs(p.dy_s_inner) + 2048 + 65ll * p.dy_cg_stride);
        const int64_t span_x = 4 * (outers * llabs(p.x_s_outer) + (int64_t)(p.n_inner + 1 + llabs(p.x_shift)) * llabs(p.x_s_inner) + 2048 + 65ll * p.x_cg_stride);
        if (p.dy_s_outer < 0 || p.dy_s_inner < 0 || p.x_s_outer < 0 || p.x_s_inner < 0 || span_dy >= (1ll << 32) || span_x >= (1ll << 32))
            return fail(IPLAN_EINVAL, "iplan_wgrad: problem %d: a row chunk spans more than 4 GiB (or negative strides)", i);
        const int kind = g.to == WG_TO_WIDE ? J_WIDE : (g.KT <= 1 ? J_THIN_K : (g.OT == 1 ? J_THIN_O : J_SQUARE));
        if (g.vchunks > vcs[kind]) vcs[kind] = g.vchunks;
        for (int j = 0; j < g.jobs; ++j) {
            if (jl[kind].n >= WG_MAX_JOBS) return fail(IPLAN_EINVAL, "iplan_wgrad: more than %d tile jobs of one shape", WG_MAX_JOBS);
            jl[kind].pj[jl[kind].n++] = (i << 8) | j;
        }
    }
    if (off > a->workspace_floats)
        return fail(IPLAN_EINVAL, "iplan_wgrad: workspace too small (%lld floats needed, %lld given)", (long long)off,
                    (long long)a->workspace_floats);
    // Pairs among the wide jobs (wgrad_pair_bf16_kernel): two single-job problems [192 x 64] on the same dY rows whose first 128
    // output columns are the same dY columns -- the two weights of a 64-wide GRU.  IPLAN_WG_NO_PAIR=1: A/B knob.
    WgradJobs jpair;
    jpair.n = 0;
#ifndef IPLAN_WG_WIDE_FP32
    if (jl[J_WIDE].n >= 2 && getenv("IPLAN_WG_NO_PAIR") == nullptr) {
        auto pairable = [&](int i) {
            const IplanWgradProblem& p = a->p[i];
            return p.O == 192 && p.K == 64 && !p.x0 && p.x && p.seg_split >= 128 && p.seg_split <= 192;
        };
        bool used[IPLAN_WGRAD_MAX] = {};
        WgradJobs rest;
        rest.n = 0;
        for (int k = 0; k < jl[J_WIDE].n; ++k) {
            const int i = jl[J_WIDE].pj[k] >> 8;
            if (used[i]) continue;
            int mate = -1;
            if (pairable(i))
                for (int k2 = k + 1; k2 < jl[J_WIDE].n && mate < 0; ++k2) {
                    const int j = jl[J_WIDE].pj[k2] >> 8;
                    const IplanWgradProblem &p = a->p[i], &q = a->p[j];
                    if (!used[j] && j != i && pairable(j) && p.dy == q.dy && p.dy_s_net == q.dy_s_net && p.dy_s_outer == q.dy_s_outer &&
                        p.dy_s_inner == q.dy_s_inner && p.dy_cg_stride == q.dy_cg_stride && p.n_outer == q.n_outer && p.n_inner == q.n_inner &&
                        p.seg_c0 == q.seg_c0)
                        mate = j;
                }
            if (mate >= 0) {
                used[i] = used[mate] = true;
                jpair.pj[jpair.n++] = i << 8;
                jpair.pj[jpair.n++] = mate << 8;
            } else if (!used[i]) {
                used[i] = true;
                rest.pj[rest.n++] = jl[J_WIDE].pj[k];
            }
        }
        jl[J_WIDE] = rest;
    }
#endif
    // The wide jobs first (one long-lived 372-register wave per SIMD: nothing else gets onto the chip while they run), the
    // thin / square ones -- many short waves -- behind them.  In the training loop this call is the decoder update that
    // Behavior_policy.learn defers beside the next rollout: with the thin kernels in front (1.6 ms) the wide one was still
    // running when the rollout's first GAT launches arrived and they queued behind it (in-situ gat_enc_fwd mean 143-145 us);
    // with the wide one in front it has retired by then (116 us).  Cycle time unchanged (356.2-358.2 ms either way, same box).
#define IPLAN_WGRAD_LAUNCH(KIND, TO_, TK_, RB_)                                                                              \
    if (jl[KIND].n)                                                                                                          \
        hipLaunchKernelGGL((wgrad_partial_kernel<TO_, TK_, RB_>), dim3((unsigned)jl[KIND].n, (unsigned)vcs[KIND], (unsigned)a->n_nets), \
                           dim3(64), 0, (hipStream_t)stream, *a, jl[KIND], chunks_wide);
#ifdef IPLAN_WG_WIDE_FP32                                    // A/B builds: the fp32 MFMA form of the wide jobs
    IPLAN_WGRAD_LAUNCH(J_WIDE, WG_TO_WIDE, WG_TK, IPLAN_WG_WIDE_RB)
#else
    if (getenv("IPLAN_WG_DEBUG")) fprintf(stderr, "iplan_wgrad: %d paired GRU jobs, %d single wide jobs, %d row chunks\n", jpair.n, jl[J_WIDE].n, vcs[J_WIDE]);
    if (jpair.n)
        hipLaunchKernelGGL(wgrad_pair_bf16_kernel, dim3((unsigned)(jpair.n / 2), (unsigned)vcs[J_WIDE], (unsigned)a->n_nets), dim3(128), 0,
                           (hipStream_t)stream, *a, jpair, chunks_wide);
    if (jl[J_WIDE].n) {
        bool any_x0 = false;
        for (int k = 0; k < jl[J_WIDE].n; ++k) any_x0 = any_x0 || a->p[jl[J_WIDE].pj[k] >> 8].x0 != nullptr;
        if (any_x0) { IPLAN_WGRAD_LAUNCH(J_WIDE, WG_TO_WIDE, WG_TK, IPLAN_WG_WIDE_RB) }
        else
            hipLaunchKernelGGL((wgrad_partial_bf16_kernel<WG_TO_WIDE, WG_TK>), dim3((unsigned)jl[J_WIDE].n, (unsigned)vcs[J_WIDE], (unsigned)a->n_nets),
                               dim3(64), 0, (hipStream_t)stream, *a, jl[J_WIDE], chunks_wide);
    }
#endif
    IPLAN_WGRAD_LAUNCH(J_THIN_K, WG_TO_NARROW, 1, 2)
    IPLAN_WGRAD_LAUNCH(J_THIN_O, 1, WG_TK, 2)
    IPLAN_WGRAD_LAUNCH(J_SQUARE, WG_TO_NARROW, WG_TK, 1)
#undef IPLAN_WGRAD_LAUNCH
    const unsigned z = (unsigned)(a->n_problems * a->n_nets);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((max_elems + 255) / 256), z), dim3(256), 0,
                       (hipStream_t)stream, *a, chunks_wide);
    return check_launch("iplan_wgrad");
}
