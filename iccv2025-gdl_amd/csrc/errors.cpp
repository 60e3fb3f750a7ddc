// errors.cpp -- thread-local error state of libgdl_hip.
#include <stdlib.h>
#include <stdarg.h>

#include "common.h"

namespace gdl {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return GDL_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return GDL_ERR_HIP;
}
const char* last_error() { return g_err; }
const char* tune_env(const char* name) {
    static int on = -1;
    if (on < 0) {
        const char* t = getenv("GDL_TUNING");
        on = (t && atoi(t) != 0) ? 1 : 0;
    }
    return on ? getenv(name) : nullptr;
}
}  // namespace gdl
