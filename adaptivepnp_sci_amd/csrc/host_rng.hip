// HOST function: NumPy's legacy Gaussian stream, bit for bit.
//
// The reference draws the FastDVDnet finetune noise with  np.random.normal(0, 5/255, shape)  from the GLOBAL legacy
// NumPy RNG (utils/utils_image.py:183-192): 6.3 M float64 values at 512x512x8, ~65 ms inside NumPy -- which holds the GIL
// for the whole call, so no Python thread (and hence no kernel launch) makes progress meanwhile.  This restatement of the
// published algorithm (numpy/random/src/legacy/legacy-distributions.c `legacy_gauss`, src/mt19937/mt19937.h: MT19937,
// 53-bit doubles from two 32-bit draws, Marsaglia polar method with one cached deviate) takes and returns the generator
// state in the form of np.random.get_state() / set_state(), is called through ctypes (GIL released) and reproduces the
// stream exactly (tests/test_host_rng.py compares against np.random.normal for many states and sizes).
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "host_common.hpp"

namespace {

constexpr int MT_N = 624, MT_M = 397;

inline void mt_refill(uint32_t* mt) {
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int kk = 0;
    uint32_t y;
    for (; kk < MT_N - MT_M; ++kk) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; kk < MT_N - 1; ++kk) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (mt[MT_N - 1] & UPPER) | (mt[0] & LOWER);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
}

struct Mt {
    uint32_t* key;
    int pos;
    inline uint32_t next32() {
        if (pos == MT_N) { mt_refill(key); pos = 0; }
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    inline double next_double() {
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

}  // namespace

extern "C" int scipnp_host_legacy_normal(uint32_t* key, int* pos, int* has_gauss, double* cached_gaussian, double loc,
                                         double scale, double* out, size_t n) {
    SCIPNP_REQUIRE(key && pos && has_gauss && cached_gaussian && (out || !n), "null pointer");
    SCIPNP_REQUIRE(*pos >= 0 && *pos <= MT_N, "generator position %d outside the MT19937 state (0..624)", *pos);
    Mt g{key, *pos};
    int have = *has_gauss;
    double cache = *cached_gaussian;
    for (size_t i = 0; i < n; ++i) {
        double v;
        if (have) {
            v = cache;
            have = 0;
            cache = 0.0;
        } else {
            double f, x1, x2, r2;
            do {
                x1 = 2.0 * g.next_double() - 1.0;
                x2 = 2.0 * g.next_double() - 1.0;
                r2 = x1 * x1 + x2 * x2;
            } while (r2 >= 1.0 || r2 == 0.0);
            f = std::sqrt(-2.0 * std::log(r2) / r2);
            cache = f * x1;
            have = 1;
            v = f * x2;
        }
        out[i] = loc + scale * v;
    }
    *pos = g.pos;
    *has_gauss = have;
    *cached_gaussian = cache;
    return SCIPNP_OK;
}
