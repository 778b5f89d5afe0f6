// C-ABI bookkeeping: version + thread-local error string.
#include "api_util.h"

namespace iplan {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace iplan

extern "C" const char* iplan_last_error(void) { return iplan::error_buffer(); }
extern "C" int iplan_version(void) { return 100; }
