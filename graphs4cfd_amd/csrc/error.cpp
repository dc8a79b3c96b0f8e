#include "g4c_common.h"

namespace g4c {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int visible_devices() {
    static const int n = [] {
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; }
        return c;
    }();
    return n;
}
}  // namespace g4c

extern "C" int g4c_version(void) { return 1; }
extern "C" const char *g4c_last_error(void) { return g4c::g_err; }
