// Error reporting + launch checking shared by the C-ABI entry points.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "iplan_hip.h"

namespace iplan {

char* error_buffer();   // thread-local, defined in api.cpp

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(IPLAN_EHIP, "%s: %s", what, hipGetErrorString(e));
    return IPLAN_OK;
}

int ac_fwd_check(const IplanAcFwdArgs* a);   // argument checks of iplan_ac_fwd (actor_critic.hip), also used by iplan_gat_enc_ac_fwd

__host__ __device__ inline bool aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

}  // namespace iplan
