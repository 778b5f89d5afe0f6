// One-shot peer-to-peer sum all-reduce over xGMI (include/iplan_hip.h, IplanP2pArgs): every rank copies its buffer into a
// staging half its peers can read, tells them so through a flag word in THEIR memory, waits for their flags in its own memory
// and sums all staging buffers in rank order.  No ring, no intermediate hops: for the path's 0.3-4 MB gradient arenas the
// collective is one local copy, one remote 4-byte store per peer and one pass of remote reads.
#include <cstdlib>
#include <cstring>

#include "api_util.h"

namespace iplan {

#ifdef IPLAN_HOST_EMULATION
#define P2P_RELEASE_FENCE() do {} while (0)
#define P2P_ACQUIRE_FENCE() do {} while (0)
#define P2P_STORE_FLAG(p, v) (*(p) = (v))
#define P2P_LOAD_FLAG(p) (*(p))
#define P2P_SLEEP() do {} while (0)
#else
// system scope: the flag lives in another device's memory (store) / is written by another device (load)
#define P2P_RELEASE_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "")
#define P2P_ACQUIRE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "")
#define P2P_STORE_FLAG(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM)
#define P2P_LOAD_FLAG(p) __hip_atomic_load((p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
#define P2P_SLEEP() __builtin_amdgcn_s_sleep(8)
#endif

typedef float p2p_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void p2p_copy_kernel(IplanP2pArgs a) {
    const int64_t n4 = a.count / 4;
    const p2p_f4* __restrict__ src = reinterpret_cast<const p2p_f4*>(a.data);
    p2p_f4* __restrict__ dst = reinterpret_cast<p2p_f4*>(a.stage[a.rank] + (int64_t)(a.seq & 1u) * a.capacity);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// after the copy kernel has completed (stream order): one flag store per rank, this rank's own included
__global__ __launch_bounds__(64) void p2p_flag_kernel(IplanP2pArgs a) {
    P2P_RELEASE_FENCE();
    const int p = (int)threadIdx.x;
    if (p < a.world) P2P_STORE_FLAG(a.flags[p] + a.rank, a.seq);
}

__global__ __launch_bounds__(256) void p2p_reduce_kernel(IplanP2pArgs a) {
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if ((int)threadIdx.x < a.world) {                       // lane p waits for rank p's flag (sequence numbers only grow)
        const uint32_t* f = a.flags[a.rank] + threadIdx.x;
        int64_t polls = 0;
        while ((int32_t)(P2P_LOAD_FLAG(f) - a.seq) < 0) {
            if (a.spin_limit > 0 && ++polls > a.spin_limit) { s_bad = 1; break; }
            P2P_SLEEP();
        }
    }
    __syncthreads();
    if (s_bad) {
        if (threadIdx.x == 0 && a.error) *a.error = 1;
        return;
    }
    P2P_ACQUIRE_FENCE();                                    // every workgroup: its own caches may hold the half's previous contents
    const int64_t n4 = a.count / 4;
    const int64_t half = (int64_t)(a.seq & 1u) * a.capacity;
    p2p_f4* __restrict__ out = reinterpret_cast<p2p_f4*>(a.data);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        p2p_f4 s = reinterpret_cast<const p2p_f4*>(a.stage[0] + half)[i];
        for (int p = 1; p < a.world; ++p) s += reinterpret_cast<const p2p_f4*>(a.stage[p] + half)[i];
        out[i] = s;
    }
}

static int check_p2p(const IplanP2pArgs* a, const char* what) {
    if (!a) return fail(IPLAN_EINVAL, "%s: null args", what);
    if (a->world < 1 || a->world > IPLAN_P2P_MAX_RANKS || a->rank < 0 || a->rank >= a->world)
        return fail(IPLAN_EINVAL, "%s: world=%d rank=%d", what, a->world, a->rank);
    if (a->count < 0 || a->count > a->capacity || (a->count & 3) || (a->capacity & 3) || a->seq == 0)
        return fail(IPLAN_EINVAL, "%s: count=%lld capacity=%lld (multiples of 4, count <= capacity), seq=%u (>= 1)", what,
                    (long long)a->count, (long long)a->capacity, a->seq);
    if (!a->data || !aligned16(a->data)) return fail(IPLAN_EALIGN, "%s: data must be a 16-byte aligned device pointer", what);
    for (int p = 0; p < a->world; ++p)
        if (!a->stage[p] || !a->flags[p] || !aligned16(a->stage[p])) return fail(IPLAN_EINVAL, "%s: stage / flags of rank %d missing", what, p);
    return IPLAN_OK;
}

}  // namespace iplan

extern "C" int iplan_p2p_publish(const IplanP2pArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_p2p(a, "iplan_p2p_publish")) return rc;
    const int64_t n4 = a->count / 4;
    const unsigned blocks = (unsigned)(n4 < 256 ? 1 : (n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256);
    if (n4 > 0) hipLaunchKernelGGL(p2p_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(p2p_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_p2p_publish");
}

extern "C" int iplan_p2p_reduce(const IplanP2pArgs* a, iplan_stream_t stream) {
    using namespace iplan;
    if (int rc = check_p2p(a, "iplan_p2p_reduce")) return rc;
    const int64_t n4 = a->count / 4;
    // few enough workgroups that all are resident: every one of them polls the flags
    const unsigned blocks = (unsigned)(n4 < 256 ? 1 : (n4 + 255) / 256 > 512 ? 512 : (n4 + 255) / 256);
    hipLaunchKernelGGL(p2p_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return check_launch("iplan_p2p_reduce");
}

#ifdef IPLAN_HOST_EMULATION
// host build (tests/emu): "device" memory is host memory and a handle is the pointer itself (one process)
extern "C" int iplan_p2p_alloc(size_t bytes, void** p) { if (!p) return IPLAN_EINVAL; *p = calloc(1, bytes); return *p ? IPLAN_OK : IPLAN_EHIP; }
extern "C" int iplan_p2p_free(void* p) { free(p); return IPLAN_OK; }
extern "C" int iplan_p2p_export(void* p, IplanIpcHandle* h) { if (!p || !h) return IPLAN_EINVAL; memset(h, 0, sizeof(*h)); memcpy(h->bytes, &p, sizeof(p)); return IPLAN_OK; }
extern "C" int iplan_p2p_open(const IplanIpcHandle* h, void** p) { if (!p || !h) return IPLAN_EINVAL; memcpy(p, h->bytes, sizeof(*p)); return IPLAN_OK; }
extern "C" int iplan_p2p_close(void*) { return IPLAN_OK; }
#else
static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(IplanIpcHandle), "IplanIpcHandle too small");
extern "C" int iplan_p2p_alloc(size_t bytes, void** p) {
    using namespace iplan;
    if (!p || bytes == 0) return fail(IPLAN_EINVAL, "iplan_p2p_alloc: bad arguments");
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? IPLAN_OK : fail(IPLAN_EHIP, "iplan_p2p_alloc: %s", hipGetErrorString(e));
}
extern "C" int iplan_p2p_free(void* p) {
    using namespace iplan;
    hipError_t e = hipFree(p);
    return e == hipSuccess ? IPLAN_OK : fail(IPLAN_EHIP, "iplan_p2p_free: %s", hipGetErrorString(e));
}
extern "C" int iplan_p2p_export(void* p, IplanIpcHandle* h) {
    using namespace iplan;
    if (!p || !h) return fail(IPLAN_EINVAL, "iplan_p2p_export: null argument");
    hipIpcMemHandle_t ih;
    hipError_t e = hipIpcGetMemHandle(&ih, p);
    if (e != hipSuccess) return fail(IPLAN_EHIP, "iplan_p2p_export: %s", hipGetErrorString(e));
    memset(h, 0, sizeof(*h));
    memcpy(h->bytes, &ih, sizeof(ih));
    return IPLAN_OK;
}
extern "C" int iplan_p2p_open(const IplanIpcHandle* h, void** p) {
    using namespace iplan;
    if (!p || !h) return fail(IPLAN_EINVAL, "iplan_p2p_open: null argument");
    hipIpcMemHandle_t ih;
    memcpy(&ih, h->bytes, sizeof(ih));
    hipError_t e = hipIpcOpenMemHandle(p, ih, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? IPLAN_OK : fail(IPLAN_EHIP, "iplan_p2p_open: %s", hipGetErrorString(e));
}
extern "C" int iplan_p2p_close(void* p) {
    using namespace iplan;
    hipError_t e = hipIpcCloseMemHandle(p);
    return e == hipSuccess ? IPLAN_OK : fail(IPLAN_EHIP, "iplan_p2p_close: %s", hipGetErrorString(e));
}
#endif
