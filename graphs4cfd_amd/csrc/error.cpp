#include "g4c_common.h"
#include <atomic>

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

int cu_count() {
    constexpr int MAXD = 64;
    static std::atomic<int> cached[MAXD];          // zero-initialised; a benign race stores the same value twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    int n = (dev >= 0 && dev < MAXD) ? cached[dev].load(std::memory_order_relaxed) : 0;
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    if (dev >= 0 && dev < MAXD) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}
}  // namespace g4c

extern "C" int g4c_version(void) { return 1; }
extern "C" int g4c_device_info(const void *device_ptr, int32_t *device, int32_t *cu_count) {
    G4C_REQUIRE(device_ptr && device && cu_count, G4C_EINVAL, "g4c_device_info: null pointer");
    g4c::DeviceGuard on_device(device_ptr);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); g4c::set_error("g4c_device_info: hipGetDevice failed"); return G4C_ELAUNCH; }
    *device = dev;
    *cu_count = g4c::cu_count();
    return G4C_OK;
}
extern "C" const char *g4c_last_error(void) { return g4c::g_err; }
