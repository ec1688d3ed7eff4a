// Error reporting + version for libthinktwice_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/thinktwice_hip.h"

namespace tt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tt

extern "C" const char* tt_last_error(void) { return tt::g_err; }

// Which kernel the last tt_conv2d_fwd of this thread launched (measurement aid: bench.py groups its per-launch HIP-event
// times by kernel so that the dominant kernel's roofline can be set beside rocprofv3's per-kernel average).
namespace tt { thread_local char g_conv_kernel[96] = ""; }
extern "C" const char* tt_conv_last_kernel(void) { return tt::g_conv_kernel; }
extern "C" int tt_version(void) { return 100; }
