// Shared helpers for the scipnp HIP sources (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
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

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SCIPNP_EHIP, "%s: %s", what, hipGetErrorString(e));
    return SCIPNP_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define SCIPNP_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::scipnp::fail(SCIPNP_EINVAL, __VA_ARGS__); } while (0)
#define SCIPNP_ALIGNED(p) \
    do { if (!::scipnp::aligned16(p)) return ::scipnp::fail(SCIPNP_EALIGN, #p " is not 16-byte aligned"); } while (0)

constexpr int WAVE = 64;

// Sums of nB addends in the exact orders PyTorch's CPU `torch.sum(dim)` uses (cascade_sum in
// ATen/native/cpu/SumKernel.cpp, 8-lane float vectors; verified against torch 2.10 for 1 <= nB <= 48),
// so that the projection is bit-identical to the reference's expressions.  Both are fully unrolled
// over MAXB with predicates, so `term(i)` only ever sees compile-time indices (register arrays stay
// in registers).
//
// (1) reduced dim NOT contiguous (e.g. `torch.sum(Phiall[...,ib], dim=2)`, dvp...:72): four
//     accumulators over the full groups of four (element i -> accumulator i%4), the remainder appended
//     to accumulator 0, then ((a0+a1)+a2)+a3.
template <int MAXB, typename F>
__device__ __forceinline__ float torch_strided_sum(int nB, F term) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int n4 = nB & ~3;
#pragma unroll
    for (int i = 0; i + 3 < MAXB; i += 4) {
        if (i < n4) {
            a0 = a0 + term(i);
            a1 = a1 + term(i + 1);
            a2 = a2 + term(i + 2);
            a3 = a3 + term(i + 3);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXB; ++i)
        if (i >= n4 && i < nB) a0 = a0 + term(i);
    return ((a0 + a1) + a2) + a3;
}

// (2) reduced dim contiguous (e.g. `torch.sum(x*Phi, dim=2)` on the fresh (M,N,B) product inside A_(),
//     utilspy.py:33): for nB >= 8 the row is cut into 8-float vectors; lane sums L[k] = sum_j v[8j+k]
//     (sequential in j for up to four vectors); result = (sum of the tail elements past the last full
//     vector, sequential) then + L[0], + L[1], ... + L[7].  For nB < 8 the scalar path (1) applies.
//     Valid for nB <= 39 (MAXB <= 32 here).
template <int MAXB, typename F>
__device__ __forceinline__ float torch_contig_sum(int nB, F term) {
    if (nB < 8) return torch_strided_sum<MAXB>(nB, term);
    const int nv = nB >> 3;
    float f = 0.f;
#pragma unroll
    for (int i = 0; i < MAXB; ++i)
        if (i >= 8 * nv && i < nB) f = f + term(i);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float L = 0.f;
#pragma unroll
        for (int j = 0; j < MAXB / 8; ++j)
            if (j < nv) L = L + term(8 * j + k);
        f = f + L;
    }
    return f;
}

// block-wide sum of one double per thread (block size a multiple of 64, <= 1024; `tid` is the
// linear thread id); result valid in thread 0.  Fixed tree + fixed wave order: deterministic.
__device__ __forceinline__ double block_sum_double(double v, double* lds /* >= 16 doubles */, int tid, int nthreads) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (tid == 0) {
        const int nw = (nthreads + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += lds[i];
    }
    __syncthreads();
    return r;
}

// ---- split-fp16 operand format of conv_split.hip: v = hi + lo' * 2^-11
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
constexpr float SPLIT_LO_SCALE = 2048.f;
__device__ __forceinline__ void split8_store(const float (&v)[8], char* hi_ptr, char* lo_ptr) {
    half8_t h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        h[e] = hh;
        l[e] = (_Float16)((v[e] - (float)hh) * SPLIT_LO_SCALE);
    }
    *(half8_t*)hi_ptr = h;
    *(half8_t*)lo_ptr = l;
}

}  // namespace scipnp
