// Thread-local error message and version of libgat_mi355.so.
#include <stdarg.h>

#include "common.h"

namespace gatmi {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gatmi

extern "C" const char *gatmi_last_error(void) { return gatmi::g_err; }
extern "C" int gatmi_version(void) { return 100; }
