// The id -> slot history of the Highway observations on the device (reference: observation_wrapper.py:68-141,
// observersation_state_history_wrapper.obs_history_create / obs_history_output / obs_single_history_output).
//
// The reference walks threads x agents x observed rows in Python every environment step; iplan_amd/observation_wrapper.py is the
// vectorised numpy form (15 sequential row rounds of ~15 small-array calls each: 1.1 ms per step at 32 envs on the GPU box, 70 % of
// a vector step of the device-resident runner).  Here the state lives in HBM next to the episode container and a step is ONE
// launch fed by ONE host -> device copy of the raw observations (K x nA x obs_num x (1 + d) floats, 58 KB at config 3):
//
//     slot_id [K, nA, N] int32 (-1 = free)    n_slots [K, nA]    win [K, nA, N, L, d]  (right-aligned last L entries per slot)
//
// One wave per (thread k, registered agent index): it owns that pair's slots, pulls the pair's window block into LDS, replays the
// rows of every agent whose ego id resolves to it in the NUMPY WRAPPER's order (rows outer, agents inner -- slot numbers depend
// on the order of first appearance; zero entries after all rows).  That equals the reference's order (agent outer: agent i's
// rows, then agent i's zero entries, then agent i + 1) whenever the ego ids of a thread are unique, which is every case the
// reference itself survives: with two agents on one ego id its obs_history_output raises IndexError (the duplicate-id tests
// compare this kernel with the numpy class only).  It appends a zero entry to every known slot an agent did not observe, and
// writes the block back together with the single-step view the GAT kernel reads (straight into the episode container).  Values
// are copied, never computed: the windows are bit-identical to the numpy wrapper's after its float32 cast.
#include "api_util.h"
#include "wave_tile.h"

namespace iplan {

__device__ __forceinline__ int wave_min_i(int v) {
    for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(64) void obs_history_kernel(IplanObsHistArgs a) {
    IPLAN_DYN_LDS(s_win);                                           // [N][L * d]
    const int nA = a.nA, N = a.N, d = a.d, LD = a.L * a.d, W = 1 + a.d;
    const int k = (int)blockIdx.x / nA, idx = (int)blockIdx.x % nA;
    const int l = (int)threadIdx.x;
    const int64_t st = (int64_t)k * nA + idx;
    float* __restrict__ gw = a.win + st * N * LD;
    for (int i = l; i < N * LD; i += 64) s_win[i] = gw[i];
    int my_id = l < N ? a.slot_id[st * N + l] : -1;                 // lane = slot
    int n_slots = a.n_slots[st];
    uint32_t seen = 0;                                              // bit a2 of lane `slot`: agent a2 observed this slot in this step
    __syncthreads();
    // agents of this thread whose ego id resolves to this state: position of the id in the registered list (list.index)
    uint32_t mine = 0;
    for (int a2 = 0; a2 < nA; ++a2) {
        const int ego = (int)a.obs[((int64_t)(k * nA + a2) * a.obs_num) * W];
        int pos = -1;
        for (int p = nA - 1; p >= 0; --p)
            if (a.agent_ids[k * nA + p] == ego) pos = p;
        if (pos < 0 && idx == 0 && l == 0) atomicOr(a.err, 1);      // "x is not in list" in the reference
        if (pos == idx) mine |= 1u << a2;
    }
    for (int j = 0; j < a.obs_num; ++j) {
        for (int a2 = 0; a2 < nA; ++a2) {
            if (!((mine >> a2) & 1u)) continue;
            const float* __restrict__ row = a.obs + ((int64_t)(k * nA + a2) * a.obs_num + j) * W;
            bool present = false;
            for (int c = 0; c < W; ++c) present = present || row[c] != 0.0f;
            if (!present) continue;                                 // an all-zero row is padding (:76)
            const int vid = (int)row[0];
            const int hit = wave_min_i((l < N && my_id == vid) ? l : 64);
            const int slot = hit < 64 ? hit : n_slots;
            if (slot >= N) {                                        // more than max_vehicle_num vehicles seen by one agent
                if (l == 0) atomicOr(a.err, 2);
                continue;
            }
            if (hit == 64) {
                if (l == slot) my_id = vid;
                n_slots += 1;
            }
            // append: the slot's L entries move up by one, the new one goes last (deque(maxlen) semantics of the window view)
            float* __restrict__ sw = s_win + slot * LD;
            for (int base = 0; base < LD; base += 64) {
                const int i = base + l;
                float nv = 0.f;
                if (i < LD) nv = i < LD - d ? sw[i + d] : row[1 + i - (LD - d)];
                IPLAN_WAVE_SYNC();
                if (i < LD) sw[i] = nv;
                IPLAN_WAVE_SYNC();
            }
            if (l == slot) seen |= 1u << a2;
        }
    }
    // known ids an agent did not observe in this step: a zero entry each (:92-96), agent after agent
    for (int a2 = 0; a2 < nA; ++a2) {
        if (!((mine >> a2) & 1u)) continue;
        if (l < n_slots && !((seen >> a2) & 1u)) {
            float* __restrict__ sw = s_win + l * LD;
            for (int i = 0; i < LD - d; ++i) sw[i] = sw[i + d];
            for (int i = LD - d; i < LD; ++i) sw[i] = 0.f;
        }
    }
    __syncthreads();
    for (int i = l; i < N * LD; i += 64) gw[i] = s_win[i];
    if (l < N) a.slot_id[st * N + l] = my_id;
    if (l == 0) a.n_slots[st] = n_slots;
    if (a.single) {
        float* __restrict__ so = a.single + (int64_t)k * a.single_s_k + (int64_t)idx * a.single_s_a;
        for (int i = l; i < N * d; i += 64) so[i] = s_win[(i / d) * LD + (LD - d) + i % d];
    }
}

}  // namespace iplan

extern "C" int iplan_obs_history_step(const IplanObsHistArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (!a || a->K < 1 || a->nA < 1 || a->nA > 32 || a->N < 1 || a->N > 64 || a->L < 1 || a->d < 1 || a->obs_num < 1 || !a->obs ||
        !a->agent_ids || !a->slot_id || !a->n_slots || !a->win || !a->err)
        return fail(IPLAN_EINVAL, "iplan_obs_history_step: bad arguments (N <= 64, n_agents <= 32)");
    const size_t lds = sizeof(float) * (size_t)a->N * a->L * a->d;
    if (lds > 64 * 1024) return fail(IPLAN_EINVAL, "iplan_obs_history_step: a (thread, agent) window block of %zu bytes does not fit", lds);
    hipLaunchKernelGGL(obs_history_kernel, dim3((unsigned)(a->K * a->nA)), dim3(64), lds, (hipStream_t)stream, *a);
    return check_launch("iplan_obs_history_step");
}
