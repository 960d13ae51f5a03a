// HOST functions: the weight packers a host calls on host memory (the engines themselves pack on the device,
// scipnp_pack_conv3x3_device / _split_device).  Pure host code -- no kernel, no HIP call -- so that `make asan` can build
// it with the host compiler under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_asan.py).
//   fp32 layout    [Cin/8][9 taps][CoutP][8] floats + CoutP bias floats           (csrc/conv.hip reads it)
//   split layout   [Cin/8][9 taps][hi | lo'][CoutP][8] fp16 + CoutP bias floats   (csrc/conv_split.hip reads it)
#include <cmath>
#include <cstring>

#include "host_common.hpp"

namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int round_up_s(int v, int m) { return round_up(v, m); }
constexpr float CS_LO_SCALE = 2048.f;      // lo' = fp16((v - hi) * 2^11), csrc/conv_split.hip

// fp32 -> (hi, lo') halves on the host (round-to-nearest-even through _Float16)
inline void split_host(float v, _Float16* hi, _Float16* lo) {
    const _Float16 h = (_Float16)v;
    *hi = h;
    *lo = (_Float16)((v - (float)h) * CS_LO_SCALE);
}

}  // namespace

using namespace scipnp;

extern "C" {

size_t scipnp_conv3x3_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    const int CoutP = round_up(Cout, 32);
    return (size_t)(Cin / 8) * 9 * CoutP * 8 + CoutP;
}

int scipnp_pack_conv3x3_weights(const float* w, const float* bias, const float* bn_scale, const float* bn_shift,
                                int Cin_real, int Cout_real, int Cin, int Cout, float* packed) {
    SCIPNP_REQUIRE(w && packed, "null pointer");
    SCIPNP_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin_real > 0 && Cout_real > 0 && Cin_real <= Cin && Cout_real <= Cout,
                   "bad channel counts Cin_real=%d Cout_real=%d Cin=%d Cout=%d", Cin_real, Cout_real, Cin, Cout);
    const int CoutP = round_up(Cout, 32);
    const size_t nw = (size_t)(Cin / 8) * 9 * CoutP * 8;
    for (size_t i = 0; i < nw + CoutP; ++i) packed[i] = 0.f;
    for (int co = 0; co < Cout_real; ++co) {
        const float sc = bn_scale ? bn_scale[co] : 1.f;
        for (int ci = 0; ci < Cin_real; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                const float v = w[((size_t)co * Cin_real + ci) * 9 + tap];
                packed[(((size_t)(ci / 8) * 9 + tap) * CoutP + co) * 8 + (ci % 8)] = bn_scale ? v * sc : v;
            }
        float bv = bias ? bias[co] : 0.f;
        if (bn_scale) bv = bv * sc;
        if (bn_shift) bv = bv + bn_shift[co];
        packed[nw + co] = bv;
    }
    return SCIPNP_OK;
}

size_t scipnp_conv3x3_split_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    const int CoutP = round_up_s(Cout, 32);
    return (size_t)(Cin / 8) * 9 * 2 * CoutP * 16 + (size_t)CoutP * 4;
}

int scipnp_pack_conv3x3_split_bn(const float* w, const float* bias, const float* bn_scale, const float* bn_shift,
                                 int Cin_real, int Cout_real, int Cin, int Cout, void* packed) {
    SCIPNP_REQUIRE(w && packed, "null pointer");
    SCIPNP_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin_real > 0 && Cout_real > 0 && Cin_real <= Cin && Cout_real <= Cout,
                   "bad channel counts");
    const int CoutP = round_up_s(Cout, 32);
    const size_t nw_bytes = (size_t)(Cin / 8) * 9 * 2 * CoutP * 16;
    memset(packed, 0, nw_bytes + (size_t)CoutP * 4);
    _Float16* p = (_Float16*)packed;
    for (int co = 0; co < Cout_real; ++co) {
        const float sc = bn_scale ? bn_scale[co] : 1.f;
        for (int ci = 0; ci < Cin_real; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                _Float16 hi, lo;
                float wv_ = w[((size_t)co * Cin_real + ci) * 9 + tap];
                if (bn_scale) wv_ = wv_ * sc;
                if (!(fabsf(wv_) < 31.9f)) return fail(SCIPNP_EINVAL, "split-fp16 conv needs |w| < 31.9 (got %g)", (double)wv_);
                split_host(wv_, &hi, &lo);
                const size_t base = ((size_t)(ci / 8) * 9 + tap) * 2;
                p[((base + 0) * CoutP + co) * 8 + (ci % 8)] = hi;
                p[((base + 1) * CoutP + co) * 8 + (ci % 8)] = lo;
            }
    }
    float* b = (float*)((char*)packed + nw_bytes);
    for (int co = 0; co < Cout_real; ++co) {
        float bv = bias ? bias[co] : 0.f;
        if (bn_scale) bv = bv * bn_scale[co];
        if (bn_shift) bv = bv + bn_shift[co];
        b[co] = bv;
    }
    return SCIPNP_OK;
}

int scipnp_pack_conv3x3_split(const float* w, const float* bias, int Cin_real, int Cout_real, int Cin, int Cout,
                              void* packed) {
    return scipnp_pack_conv3x3_split_bn(w, bias, nullptr, nullptr, Cin_real, Cout_real, Cin, Cout, packed);
}

}  // extern "C"
