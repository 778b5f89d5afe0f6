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
#include "api_util.h"
#include "wave_tile.h"

namespace iplan {

constexpr int WG_TO = 4;   // o-tiles per wave job
constexpr int WG_TK = 4;   // k-tiles per wave job

struct WgradGeom {
    int OT, KT, n_og, n_kg, jobs, sub;      // sub = row sub-chunks per workgroup (4 waves / jobs)
    int64_t rows;
    int vchunks, vrows;                      // virtual chunks and rows per virtual chunk (multiple of 16)
    int64_t part_floats;                     // floats per virtual chunk: OT*16 * (KT*16 + 1)
};

__host__ __device__ inline WgradGeom wgrad_geom(const IplanWgradProblem& p) {
    WgradGeom g;
    g.OT = (p.O + 15) / 16;
    g.KT = (p.K + 15) / 16;
    g.n_og = (g.OT + WG_TO - 1) / WG_TO;
    g.n_kg = g.KT > 0 ? (g.KT + WG_TK - 1) / WG_TK : 1;
    g.jobs = g.n_og * g.n_kg;
    g.rows = (int64_t)p.n_outer * p.n_inner;
    int64_t vr = (g.rows + IPLAN_WGRAD_MAX_CHUNKS - 1) / IPLAN_WGRAD_MAX_CHUNKS;
    if (vr < 256) vr = 256;
    vr = (vr + 15) / 16 * 16;
    g.vrows = (int)vr;
    g.vchunks = (int)((g.rows + vr - 1) / vr);
    if (g.vchunks < 1) g.vchunks = 1;
    g.sub = 1;
    g.part_floats = (int64_t)g.OT * 16 * (g.KT * 16 + 1);
    return g;
}

__device__ __forceinline__ int ocol(const IplanWgradProblem& p, int o) {
    return o < p.seg_split ? p.seg_c0 + o : p.seg_c1 + (o - p.seg_split);
}

// grid: (problem * max_jobs + job, virtual chunk, net); block: 64 (one wave).  The (problem, job) index varies
// fastest so that the waves that read the same rows (other tile jobs of a problem, and the other problems of a
// launch -- a GRU's W_ih / W_hh share their dY rows) are dispatched together and meet in L2 / the Infinity Cache.
//
// Data path per 16-row block: every lane fetches ONE 16-byte piece of each operand tile (row = lane/4, 4
// consecutive columns: 64 contiguous bytes per row, whole cache lines across neighbouring tiles), the 16x16 tiles
// are parked in LDS and read back in MFMA operand order (lane (i, g), sub-step s: element [4s+g][i]; with a
// 16-float row pitch the two 32-lane halves of a ds_read_b32 hit disjoint banks).  The next block's global loads
// are in flight while the current block's 64 MFMAs run.
__global__ __launch_bounds__(64) void wgrad_partial_kernel(IplanWgradArgs a, int max_jobs) {
    __shared__ __attribute__((aligned(16))) float s_t[2][WG_TO + WG_TK][256];
    const int pi = (int)blockIdx.x / max_jobs, job = (int)blockIdx.x % max_jobs, net = (int)blockIdx.z;
    const IplanWgradProblem& p = a.p[pi];
    const WgradGeom gm = wgrad_geom(p);
    const int vc = (int)blockIdx.y;
    if (vc >= gm.vchunks || job >= gm.jobs) return;
    const int og = job / gm.n_kg, kg = job % gm.n_kg;
    const int l = lane_id(), i = l & 15, g = l >> 4;       // MFMA role
    const int lr = l >> 2, lc = 4 * (l & 3);               // loader role: row of the block, first of 4 columns
    const int ot0 = og * WG_TO, kt0 = kg * WG_TK;
    const int not_ = imin(WG_TO, gm.OT - ot0), nkt = gm.KT > 0 ? imin(WG_TK, gm.KT - kt0) : 0;
    const bool want_bias = (kg == 0);

    const float* __restrict__ dy = p.dy + (int64_t)net * p.dy_s_net;
    const float* __restrict__ x = p.x ? p.x + (int64_t)net * p.x_s_net : nullptr;
    const float* __restrict__ x0 = p.x0 ? p.x0 + (int64_t)net * p.x0_s_net : nullptr;

    f32x4 acc[WG_TO][WG_TK];
    f32x4 bacc[WG_TO];
    for (int t = 0; t < WG_TO; ++t) {
        bacc[t] = splat4(0.f);
        for (int u = 0; u < WG_TK; ++u) acc[t][u] = splat4(0.f);
    }
    // loader-side column bookkeeping: this lane's 4 columns of every tile
    int acol[WG_TO], anv[WG_TO], bcol[WG_TK], bnv[WG_TK];
    bool avec[WG_TO], bvec[WG_TK];
    const bool dy_al = ((p.dy_s_outer | p.dy_s_inner | p.dy_s_net) & 3) == 0 && ((((size_t)p.dy) & 15) == 0);
    const bool x_al = ((p.x_s_outer | p.x_s_inner | p.x_s_net | p.x0_s_outer | p.x0_s_net) & 3) == 0 &&
                      ((((size_t)p.x) & 15) == 0) && ((((size_t)p.x0) & 15) == 0);
    for (int t = 0; t < WG_TO; ++t) {
        const int o = (ot0 + t) * 16 + lc;
        const int rem = p.O - o;
        anv[t] = t < not_ ? (rem >= 4 ? 4 : (rem > 0 ? rem : 0)) : 0;
        acol[t] = anv[t] > 0 ? ocol(p, o) : 0;
        avec[t] = anv[t] == 4 && ocol(p, o + 3) == acol[t] + 3 && dy_al && (acol[t] & 3) == 0;
    }
    for (int u = 0; u < WG_TK; ++u) {
        const int k = (kt0 + u) * 16 + lc;
        const int rem = p.K - k;
        bnv[u] = u < nkt ? (rem >= 4 ? 4 : (rem > 0 ? rem : 0)) : 0;
        bcol[u] = p.x_col0 + k;
        bvec[u] = bnv[u] == 4 && x_al && (bcol[u] & 3) == 0;
    }
    const int64_t r_lo = (int64_t)vc * gm.vrows;
    const int64_t r_hi = gm.rows < r_lo + gm.vrows ? gm.rows : r_lo + gm.vrows;
    // (outer, inner) of the row this lane loads, advanced by 16 rows per block (no division in the loop)
    int o_s, i_s;
    {
        const int64_t r = r_lo + lr;
        o_s = (int)(r / p.n_inner);
        i_s = (int)(r - (int64_t)o_s * p.n_inner);
    }
    const int adv_o = 16 / p.n_inner, adv_i = 16 % p.n_inner;

    struct Regs { f32x4 av[WG_TO]; f32x4 bv[WG_TK]; };
    auto fetch = [&](int64_t rb, Regs& o) {
        const int64_t r = rb + lr;
        const bool rv = r < r_hi;
        const int64_t outer = o_s;
        const int inner = i_s;
        o_s += adv_o;
        i_s += adv_i;
        if (i_s >= p.n_inner) { i_s -= p.n_inner; o_s += 1; }
        const float* dyr = dy + outer * p.dy_s_outer + (int64_t)inner * p.dy_s_inner;
        for (int t = 0; t < WG_TO; ++t) {
            f32x4 v = splat4(0.f);
            if (rv && anv[t] > 0) {
                if (avec[t]) v = *reinterpret_cast<const f32x4*>(dyr + acol[t]);
                else for (int q = 0; q < 4; ++q) if (q < anv[t]) v[q] = dyr[ocol(p, (ot0 + t) * 16 + lc + q)];
            }
            o.av[t] = v;
        }
        const float* xr = nullptr;
        if (rv && nkt > 0) {
            const int xi = inner + p.x_shift;
            if (xi >= 0 && xi < p.n_inner) xr = x + outer * p.x_s_outer + (int64_t)xi * p.x_s_inner;
            else if (x0) xr = x0 + outer * p.x0_s_outer - p.x_col0;
        }
        for (int u = 0; u < WG_TK; ++u) {
            f32x4 v = splat4(0.f);
            if (xr && bnv[u] > 0) {
                if (bvec[u]) v = *reinterpret_cast<const f32x4*>(xr + bcol[u]);
                else for (int q = 0; q < 4; ++q) if (q < bnv[u]) v[q] = xr[bcol[u] + q];
            }
            o.bv[u] = v;
        }
    };
    auto park = [&](const Regs& o, int buf) {
        for (int t = 0; t < WG_TO; ++t)
            if (t < not_) *reinterpret_cast<f32x4*>(&s_t[buf][t][lr * 16 + lc]) = o.av[t];
        for (int u = 0; u < WG_TK; ++u)
            if (u < nkt) *reinterpret_cast<f32x4*>(&s_t[buf][WG_TO + u][lr * 16 + lc]) = o.bv[u];
    };
    auto contract = [&](int buf) {
        for (int s = 0; s < 4; ++s) {
            const int e = (4 * s + g) * 16 + i;
            float av[WG_TO], bv[WG_TK];
            for (int t = 0; t < WG_TO; ++t) av[t] = t < not_ ? s_t[buf][t][e] : 0.f;
            for (int u = 0; u < WG_TK; ++u) bv[u] = u < nkt ? s_t[buf][WG_TO + u][e] : 0.f;
            for (int t = 0; t < WG_TO; ++t) {
                if (t < not_) {
                    for (int u = 0; u < WG_TK; ++u)
                        if (u < nkt) acc[t][u] = mfma4(av[t], bv[u], acc[t][u]);
                    if (want_bias) bacc[t] = mfma4(av[t], 1.0f, bacc[t]);
                }
            }
        }
    };
    Regs rg;
    fetch(r_lo, rg);
    int buf = 0;
    for (int64_t rb = r_lo; rb < r_hi; rb += 16) {
        park(rg, buf);
        if (rb + 16 < r_hi) fetch(rb + 16, rg);            // next block's global loads fly during the MFMAs below
        __syncthreads();                                    // one wave per workgroup: orders the LDS hand-off
        contract(buf);
        buf ^= 1;
    }
    // partial tile layout: part[vc][o (OT*16)][KT*16 + 1]; D layout: lane (col j = i, row = 4g + q)
    float* __restrict__ part = a.workspace + p.ws_off + ((int64_t)net * gm.vchunks + vc) * gm.part_floats;
    const int ldp = gm.KT * 16 + 1;
    for (int t = 0; t < WG_TO; ++t) {
        if (t >= not_) continue;
        for (int q = 0; q < 4; ++q) {
            const int o = (ot0 + t) * 16 + 4 * g + q;
            for (int u = 0; u < WG_TK; ++u)
                if (u < nkt) part[(int64_t)o * ldp + (kt0 + u) * 16 + i] = acc[t][u][q];
            if (want_bias && i == 0) part[(int64_t)o * ldp + gm.KT * 16] = bacc[t][q];
        }
    }
}

// grid: (ceil(O*(K+1)/256), problem * n_nets + net)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(IplanWgradArgs a) {
    const int pi = (int)blockIdx.y / a.n_nets, net = (int)blockIdx.y % a.n_nets;
    const IplanWgradProblem& p = a.p[pi];
    const WgradGeom gm = wgrad_geom(p);
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int K1 = p.K + 1;
    if (idx >= p.O * K1) return;
    const int o = idx / K1, k = idx - o * K1;
    const bool is_bias = (k == p.K);
    if (is_bias ? (p.db_off < 0) : (p.dw_off < 0)) return;
    const int ldp = gm.KT * 16 + 1;
    const float* __restrict__ part = a.workspace + p.ws_off + (int64_t)net * gm.vchunks * gm.part_floats +
                                     (int64_t)o * ldp + (is_bias ? gm.KT * 16 : k);
    float s = 0.f;
    for (int vc = 0; vc < gm.vchunks; ++vc) s += part[(int64_t)vc * gm.part_floats];
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
        const WgradGeom g = wgrad_geom(a->p[i]);
        total += (size_t)g.part_floats * g.vchunks * a->n_nets;
    }
    return total;
}

extern "C" int iplan_wgrad(IplanWgradArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->n_problems < 1 || a->n_problems > IPLAN_WGRAD_MAX || a->n_nets < 1 || !a->grad || !a->workspace)
        return fail(IPLAN_EINVAL, "iplan_wgrad: bad arguments");
    int64_t off = 0;
    int max_vc = 1, max_jobs = 1, max_elems = 1;
    for (int i = 0; i < a->n_problems; ++i) {
        IplanWgradProblem& p = a->p[i];
        if (!p.dy || p.O < 1 || p.O > 1024 || p.K < 0 || p.K > 1024 || (p.K > 0 && !p.x) || p.n_outer < 1 || p.n_inner < 1)
            return fail(IPLAN_EINVAL, "iplan_wgrad: problem %d has unsupported dims O=%d K=%d", i, p.O, p.K);
        const WgradGeom g = wgrad_geom(p);
        p.ws_off = off;
        off += g.part_floats * g.vchunks * a->n_nets;
        if (g.vchunks > max_vc) max_vc = g.vchunks;
        if (g.jobs > max_jobs) max_jobs = g.jobs;
        if (p.O * (p.K + 1) > max_elems) max_elems = p.O * (p.K + 1);
    }
    if (off > a->workspace_floats)
        return fail(IPLAN_EINVAL, "iplan_wgrad: workspace too small (%lld floats needed, %lld given)", (long long)off,
                    (long long)a->workspace_floats);
    const unsigned z = (unsigned)(a->n_problems * a->n_nets);
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3((unsigned)(a->n_problems * max_jobs), (unsigned)max_vc, (unsigned)a->n_nets),
                       dim3(64), 0, (hipStream_t)stream, *a, max_jobs);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((max_elems + 255) / 256), z), dim3(256), 0,
                       (hipStream_t)stream, *a);
    return check_launch("iplan_wgrad");
}
