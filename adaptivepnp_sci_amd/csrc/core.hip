// Library identity and the thread-local error string of the C ABI.
#include "host_common.hpp"
#include <cstring>

namespace scipnp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace scipnp

extern "C" {
const char* scipnp_version(void) { return "scipnp 0.6.0 (round 6)"; }
const char* scipnp_last_error(void) { return scipnp::g_err; }
const char* scipnp_arch(void) { return "gfx950"; }
}
