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
// Measurement aid (tools/conv_trace.py): while set, every workgroup of the LDS-DMA conv kernel writes four wall-clock stamps (10 ns
// ticks: entry, first K tile landed, K loop done, epilogue done) at stamps[blockIdx.x * 4].  Null (the default) in the product.
namespace tt { long long* g_conv_trace = nullptr; }
extern "C" int tt_conv_set_trace(void* stamps_or_null) {
    tt::g_conv_trace = static_cast<long long*>(stamps_or_null);
    return 0;
}
extern "C" int tt_version(void) { return 100; }
