// Feature addressing shared by the actor/critic forward and the fc1 weight-gradient kernels.
#pragma once
#include "iplan_hip.h"
#include "wave_tile.h"

namespace iplan {

// ---- K order of the fc1 contraction ---------------------------------------------------------------
// The reference lays the F input features out entity-major ([hist_i || att_i || beh_i] per entity,
// then the one-hots).  An MFMA contraction may visit K in any order as long as A (weights) and B
// (features) agree, so the kernels use a SOURCE-major order: block s = the row's contiguous vector of
// source s (N * w_s floats, padded to a multiple of 16), then one block for the one-hots.  Feature
// loads are then plain 16-byte vector loads from the episode-buffer fields (no per-element index
// math), and the matching weight / LayerNorm columns are  e*W + off_s + k  -- 4 consecutive columns
// whenever w_s % 4 == 0 (attention 32, behaviour 8), per-element otherwise (history 5).
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4 ldu4(const float* __restrict__ p) { return *(const IPLAN_GLOBAL_AS f32x4_u*)p; }

struct KMap {
    int kt0[5];        // first k-tile of block 0..3, and the total
    int len[4];        // valid entries of each block
    int w[3], off[3];
    int W, NW, n_actions, n_id;
};

__device__ __forceinline__ KMap make_kmap(const IplanAcFeatures& ft) {
    KMap k;
    k.W = ft.w[0] + ft.w[1] + ft.w[2];
    k.NW = ft.N * k.W;
    k.n_actions = ft.n_actions;
    k.n_id = ft.n_id;
    int t = 0, off = 0;
    for (int s = 0; s < 3; ++s) {
        k.w[s] = ft.w[s];
        k.off[s] = off;
        off += ft.w[s];
        k.len[s] = ft.N * ft.w[s];
        k.kt0[s] = t;
        t += (k.len[s] + 15) / 16;
    }
    k.len[3] = ft.n_actions + ft.n_id;
    k.kt0[3] = t;
    t += (k.len[3] + 15) / 16;
    k.kt0[4] = t;
    return k;
}

struct KTile {
    int s;             // block
    int f0;            // first in-block index of this lane's 4 entries
    int nv;            // how many of the 4 are real (0..4)
    int c[4];          // their feature columns in the reference's layout
    bool contig;       // c[q] == c[0] + q
};

// c4 = which 4 of the tile's 16 columns this lane handles (MFMA role: 4 * (lane >> 4); loader role: 4 * (lane & 3))
__device__ __forceinline__ KTile ktile_at(const KMap& k, int T, int c4) {
    KTile o;
    o.s = T >= k.kt0[3] ? 3 : (T >= k.kt0[2] ? 2 : (T >= k.kt0[1] ? 1 : 0));
    o.f0 = 16 * (T - k.kt0[o.s]) + c4;
    const int rem = k.len[o.s] - o.f0;
    o.nv = rem >= 4 ? 4 : (rem > 0 ? rem : 0);
    if (o.s == 3) {
        for (int q = 0; q < 4; ++q) o.c[q] = k.NW + o.f0 + q;
        o.contig = true;
    } else {
        const int w = k.w[o.s];
        if ((w & 3) == 0) {
            const int e = o.f0 / w;
            const int c0 = e * k.W + k.off[o.s] + (o.f0 - e * w);
            for (int q = 0; q < 4; ++q) o.c[q] = c0 + q;
            o.contig = true;
        } else {
            for (int q = 0; q < 4; ++q) {
                const int f = o.f0 + q, e = f / w;
                o.c[q] = e * k.W + k.off[o.s] + (f - e * w);
            }
            o.contig = false;
        }
    }
    return o;
}

__device__ __forceinline__ KTile ktile(const KMap& k, int T) { return ktile_at(k, T, 4 * (lane_id() >> 4)); }

// K-order index of feature column c of the reference layout (inverse of KTile::c)
__device__ __forceinline__ int korder_of_column(const KMap& k, int c) {
    if (c >= k.NW) return k.kt0[3] * 16 + (c - k.NW);
    const int e = c / k.W;
    int r = c - e * k.W, s = 0;
    if (r >= k.w[0]) { r -= k.w[0]; s = 1; if (r >= k.w[1]) { r -= k.w[1]; s = 2; } }
    return k.kt0[s] * 16 + e * k.w[s] + r;
}

// this lane's 4 raw features of one row for a k-tile; COH: blocks 1 and 2 (the latents) are read device-coherently -- in the
// fused rollout launch other workgroups of the same launch write them (ac_fwd_body.h)
template <bool COH = false>
__device__ __forceinline__ f32x4 kfeat(const KMap& k, const KTile& kt, const float* const (&src)[3], bool valid, int last, int net) {
    f32x4 v = splat4(0.f);
    if (!valid || kt.nv == 0) return v;
    if (kt.s < 3) {
        const float* p = src[kt.s] + kt.f0;
        if (COH && kt.s > 0) { for (int q = 0; q < 4; ++q) if (q < kt.nv) v[q] = coh_load(p + q); }
        else if (kt.nv == 4) v = ldu4(p);
        else for (int q = 0; q < 4; ++q) if (q < kt.nv) v[q] = as_global(p)[q];
    } else {
        for (int q = 0; q < 4; ++q) {
            const int idx = kt.f0 + q;
            if (q < kt.nv) v[q] = idx < k.n_actions ? (idx == last ? 1.0f : 0.0f) : (idx - k.n_actions == net ? 1.0f : 0.0f);
        }
    }
    return v;
}

// 4 entries of a per-feature vector (LayerNorm gamma / beta, or one row of fc1.weight) at the tile's columns
__device__ __forceinline__ f32x4 kcols(const KTile& kt, const float* __restrict__ vec) {
    f32x4 v = splat4(0.f);
    if (kt.nv == 4 && kt.contig) return ldu4(vec + kt.c[0]);
    for (int q = 0; q < 4; ++q) if (q < kt.nv) v[q] = as_global(vec)[kt.c[q]];
    return v;
}


}  // namespace iplan
