// Shared pieces of the fp32 Winograd F(4x4,3x3) kernels (conv_wino4.hip: 4-wave workgroups, positions split over two waves;
// conv_wino4x.hip: 6-wave workgroups, positions split over three waves): slab / halo-tile geometry, the argument block, the
// packed-fp32 transform operations, clock stamps, reciprocal division.
#pragma once
#include "common.hpp"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace scipnp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int W4_SLAB = 2 * 9 * 64 * 4;               // floats per k-step slab (18432 B)
constexpr int W4_PIECES = W4_SLAB / 256;              // 1 KiB LDS-DMA pieces per slab (18)
constexpr int W4_TW = 64, W4_TH = 8;                  // output pixels per workgroup
constexpr int W4_TWP = W4_TW + 2, W4_THP = W4_TH + 2; // halo tile
constexpr int W4_THREADS = 256;
constexpr int W4_FLAG_NT = 1 << 20;                          // Wino4Args.flags, set by the entry point: whole-line stores with the nt hint
constexpr long long W4_NT_BYTES = 128ll << 20;              // ... for launches that write at least this much
constexpr int W4_RSL = W4_TWP + W4_TWP / 16;          // slots per halo row: pixel x sits in slot x + (x >> 4) (one padding slot per 16)
constexpr int W4_UNITS = W4_THP * W4_RSL * 2;         // 16-byte units (4 channels of a pixel) of the halo tile: [hf][row][slot]
constexpr int W4_RAW_PIECES = (W4_UNITS + 63) / 64;   // 1 KiB LDS-DMA pieces per raw tile (21; the last one partly padding)
constexpr int W4_RAW = W4_RAW_PIECES * 256;           // floats per raw buffer
constexpr int W4_IN_ITERS = (W4_RAW_PIECES + 3) / 4;  // raw pieces per wave and group (waves 1..3 fetch their fifth piece twice)
constexpr int W4_DMA_ITERS = (W4_PIECES + 3) / 4;     // U pieces per wave and slab (waves 2, 3 fetch their fourth piece twice)
constexpr size_t W4_LDS_BYTES = (2 * (size_t)W4_RAW + 2 * (size_t)W4_SLAB) * sizeof(float);
static_assert(W4_LDS_BYTES >= 4 * 16 * 64 * 16, "the epilogue's exchange buffer lives in the loop's LDS");
static_assert(2 * W4_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

struct Wino4Args {
    const float* in;
    const float* wpk;        // [2*CGin k-steps][CoutP/32][4608] + bias[CoutP]
    float* out;
    const float* residual;
    const float* mask_src;
    int CGin, CGout, NCB;    // NCB = CoutP / 32
    int H, W;
    int ntx, nty;
    unsigned m_ncb, m_ntx, m_nty;   // floor(2^32 / d) of the three divisors of the block index (w4_div below), set by w4_geometry
    unsigned total_units;           // (tile, output-channel block) units of the launch; the grid of the classic form, walked by the persistent one
    int flags;
    unsigned long long* dbg; // STAMP instantiation (DIAG bit6) only: 128 words per workgroup, see scipnp_conv3x3_c8w4_stamped
};

// (host pass: only parsed -- the kernel body never runs there)
__host__ __device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return a * b + c;
#endif
}
__host__ __device__ __forceinline__ f32x4 pk_fma(f32x4 a, f32x4 b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return a * b + c;
#endif
}
// a - b on a float2 as ONE v_pk_add_f32 (see conv_wino.hip: the compiler selects two v_sub_f32)
__host__ __device__ __forceinline__ f32x2 psub4(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - b;
#endif
}
// c * x + y on a float2 as ONE v_pk_fma_f32, c an inline constant (the compiler scalarises a <2 x float> fma whose result is
// only ever read element by element -- the MFMA operands -- into two v_fma_f32, and every vector instruction is matrix time)
#define W4_PK_FMA_CONST(NAME, LIT)                                                                     \
    __host__ __device__ __forceinline__ f32x2 NAME(f32x2 x, f32x2 y) {                                 \
        f32x2 r = x * (float)(LIT) + y;                                                                \
        W4_DEVICE_ASM("v_pk_fma_f32 %0, %1, " #LIT ", %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(y)); \
        return r;                                                                                      \
    }
#if defined(__HIP_DEVICE_COMPILE__)
#define W4_DEVICE_ASM(...) asm(__VA_ARGS__)
#else
#define W4_DEVICE_ASM(...) (void)0
#endif
W4_PK_FMA_CONST(fma_p4, 4.0)
W4_PK_FMA_CONST(fma_m4, -4.0)
W4_PK_FMA_CONST(fma_p2, 2.0)
W4_PK_FMA_CONST(fma_m2, -2.0)
// k * x + y with k (both halves the same value) in a scalar register pair: -5 is not an inline constant
__host__ __device__ __forceinline__ f32x2 fma_k(f32x2 x, f32x2 y, f32x2 k) {
    f32x2 r = x * k + y;
    W4_DEVICE_ASM("v_pk_fma_f32 %0, %1, %3, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(y), "s"(k));
    return r;
}
__host__ __device__ __forceinline__ f32x2 padd(f32x2 a, f32x2 b) {
    f32x2 r = a + b;
    W4_DEVICE_ASM("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// a - b on a float4 as two v_pk_add_f32
__host__ __device__ __forceinline__ f32x4 psub4(f32x4 a, f32x4 b) {
    const f32x2 lo = psub4(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = psub4(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

template <typename T>
__host__ __device__ __forceinline__ T splat(float v);
template <>
__host__ __device__ __forceinline__ f32x2 splat<f32x2>(float v) { return f32x2{v, v}; }
template <>
__host__ __device__ __forceinline__ f32x4 splat<f32x4>(float v) { return f32x4{v, v, v, v}; }

// one clock stamp of wave 0, written with a SCALAR store (no vmcnt traffic: the K loop's waits count vector memory operations)
#if defined(__HIP_DEVICE_COMPILE__)
#define W4_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if constexpr ((DIAG & 64) != 0) {                                                                           \
            if (wvu == 0) {                                                                                         \
                unsigned long long t_;                                                                              \
                const unsigned long long* p_ = stamp_base + (slot);                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : "=&s"(t_) : "s"(p_) : "memory"); \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
#else
#define W4_STAMP(slot) (void)0
#endif

template <int... I, typename F>
__host__ __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// ONE packed operation (K = 0..5) of half a 1-D input transform: LO: rows 0..2 of B^T x from (x0..x4) = i0..i4; else rows 3..5
// from (x1..x5) = i0..i4.  ta, tb carry the two intermediates between the operations of one half.
template <bool LO, int K>
__host__ __device__ __forceinline__ void half_op(const f32x2 i0, const f32x2 i1, const f32x2 i2, const f32x2 i3, const f32x2 i4,
                                                 f32x2& o0, f32x2& o1, f32x2& o2, f32x2& ta, f32x2& tb, const f32x2 m5) {
    if constexpr (LO) {                 // 4x0 - 5x2 + x4 | (x4 - 4x2) + (x3 - 4x1) | (x4 - 4x2) - (x3 - 4x1)
        if constexpr (K == 0) ta = fma_k(i2, i4, m5);
        if constexpr (K == 1) o0 = fma_p4(i0, ta);
        if constexpr (K == 2) ta = fma_m4(i2, i4);
        if constexpr (K == 3) tb = fma_m4(i1, i3);
        if constexpr (K == 4) o1 = padd(ta, tb);
        if constexpr (K == 5) o2 = psub4(ta, tb);
    } else {                            // (x4 - x2) + 2(x3 - x1) | (x4 - x2) - 2(x3 - x1) | 4x1 - 5x3 + x5
        if constexpr (K == 0) ta = psub4(i3, i1);
        if constexpr (K == 1) tb = psub4(i2, i0);
        if constexpr (K == 2) o0 = fma_p2(tb, ta);
        if constexpr (K == 3) o1 = fma_m2(tb, ta);
        if constexpr (K == 4) ta = fma_k(i2, i4, m5);
        if constexpr (K == 5) o2 = fma_p4(i0, ta);
    }
}

// x / d and x % d by a host-made reciprocal m = floor(2^32 / d) (0xFFFFFFFF for d = 1): q = mulhi(x, m) is the quotient or one
// short of it, one correction makes it exact for every 32-bit x -- six scalar instructions where the compiler's division by a
// run-time value takes some thirty-five (three of them open every workgroup's life)
__host__ __device__ __forceinline__ unsigned w4_div(unsigned x, unsigned d, unsigned m, unsigned& rem) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned q = __umulhi(x, m);
#else
    unsigned q = (unsigned)(((unsigned long long)x * m) >> 32);
#endif
    unsigned r = x - q * d;
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}
static inline unsigned w4_magic(int d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)d); }

}  // namespace scipnp
