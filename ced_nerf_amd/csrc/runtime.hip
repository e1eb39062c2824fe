// Library-wide state: version and the per-thread error string of the C ABI (include/cednerf_hip.h).
#include "ced_common.hpp"

namespace ced {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace ced

extern "C" int ced_version(void) { return 1; }

extern "C" int64_t ced_wall_clock_khz(void)
{
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
    return khz;
}

extern "C" const char *ced_last_error_string(void) { return ced::g_error; }
