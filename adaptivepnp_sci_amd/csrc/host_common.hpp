// Host-only helpers of the C ABI (no HIP header): the error string, argument checks.  Included by common.hpp for the device
// sources and directly by the pure-host sources (core.hip, host_rng.hip, host_pack.hip), which `make asan` builds with the
// host compiler under AddressSanitizer + UndefinedBehaviorSanitizer.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/scipnp.h"

namespace scipnp {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define SCIPNP_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::scipnp::fail(SCIPNP_EINVAL, __VA_ARGS__); } while (0)
#define SCIPNP_ALIGNED(p) \
    do { if (!::scipnp::aligned16(p)) return ::scipnp::fail(SCIPNP_EALIGN, #p " is not 16-byte aligned"); } while (0)

}  // namespace scipnp
