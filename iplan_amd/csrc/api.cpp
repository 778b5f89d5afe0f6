// C-ABI bookkeeping: version + thread-local error string.
#include <cstring>

#include "api_util.h"

namespace iplan {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace iplan

extern "C" const char* iplan_last_error(void) { return iplan::error_buffer(); }
extern "C" int iplan_version(void) { return 100; }

extern "C" size_t iplan_sizeof(const char* name) {
    if (!name) return 0;
#define IPLAN_SZ(T) if (!strcmp(name, #T)) return sizeof(T);
    IPLAN_SZ(IplanGatSaved) IPLAN_SZ(IplanGatFwdArgs) IPLAN_SZ(IplanGatBwdArgs) IPLAN_SZ(IplanEncFwdArgs) IPLAN_SZ(IplanAcNet)
    IPLAN_SZ(IplanAcFeatures) IPLAN_SZ(IplanAcFwdArgs) IPLAN_SZ(IplanAcBwdArgs) IPLAN_SZ(IplanAdamArgs) IPLAN_SZ(IplanWgradProblem)
    IPLAN_SZ(IplanWgradArgs) IPLAN_SZ(IplanPpoPrepareArgs) IPLAN_SZ(IplanPpoLossArgs) IPLAN_SZ(IplanPdecArgs) IPLAN_SZ(IplanBehArgs) IPLAN_SZ(IplanMlp3Args) IPLAN_SZ(IplanAdvNormArgs) IPLAN_SZ(IplanSeq2SeqArgs) IPLAN_SZ(IplanAcPackArgs) IPLAN_SZ(IplanP2pArgs) IPLAN_SZ(IplanIpcHandle) IPLAN_SZ(IplanAcXhatArgs) IPLAN_SZ(IplanAcFc1SplitArgs) IPLAN_SZ(IplanObsHistArgs) IPLAN_SZ(IplanSeq2SeqBwdArgs)
#undef IPLAN_SZ
    return 0;
}
