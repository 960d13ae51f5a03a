// Micro-benchmarks that MEASURE the ceilings the rooflines are quoted against ("vendor figures; builder to
// confirm with a measured stream / MFMA microbenchmark and report both"): a register-resident MFMA loop on random
// operands (fp16 32x32x16, fp16 16x16x32, fp32 32x32x2) and an HBM stream (read-only reduction, copy).  Diagnostic
// entries of the library, driven by tools/peaks_bench.py; not on the reconstruction path.
#include "common.hpp"

namespace scipnp {

typedef float pk_f32x16 __attribute__((ext_vector_type(16)));
typedef float pk_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float hash_unit(unsigned x) {            // deterministic pseudo-random value in [-1, 1)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

// mode 0: v_mfma_f32_32x32x16_f16, 1: v_mfma_f32_16x16x32_f16, 2: v_mfma_f32_32x32x2_f32; 4 independent accumulator
// chains per wave so that the matrix pipe never waits on a dependency
__global__ void __launch_bounds__(256)
mfma_peak_kernel(float* __restrict__ out, int iters, int mode, unsigned seed) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned id = (blockIdx.x * blockDim.x + threadIdx.x) * 16u + seed;
    pk_f16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)hash_unit(id + e); b[e] = (_Float16)hash_unit(id + 8 + e); }
    const float af = hash_unit(id + 3), bf = hash_unit(id + 5);
    float sink = 0.f;
    if (mode == 0) {
        pk_f32x16 acc[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][15];
    } else if (mode == 1) {
        pk_f32x4 acc[8] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) sink += acc[j][0] + acc[j][3];
    } else {
        pk_f32x16 acc[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
#endif
}

// How much vector-ALU issue a matrix instruction hides: per MFMA, NV independent v_add_f32 of the same wave between it and
// the next MFMA (16 accumulator chains, operands in registers).  F32 = v_mfma_f32_16x16x4_f32 (the Winograd kernel's),
// else v_mfma_f32_32x32x16_f16; both occupy the matrix pipe for 32 cycles.  The kernel records its own cycle count
// (s_memtime) per wave, so the result does not depend on the clock the part holds.
template <int NV, bool F32, int KIND = 0>
__global__ void __launch_bounds__(256)
mfma_valu_kernel(float* __restrict__ out, unsigned long long* __restrict__ cycles, int iters, unsigned seed) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned id = (blockIdx.x * blockDim.x + threadIdx.x) * 16u + seed;
    pk_f16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)hash_unit(id + e); b[e] = (_Float16)hash_unit(id + 8 + e); }
    const float af = hash_unit(id + 3), bf = hash_unit(id + 5);
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = hash_unit(id + 20 + k);
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    pk_f32x2 xp[4], xq = {bf, af};
#pragma unroll
    for (int k = 0; k < 4; ++k) xp[k] = pk_f32x2{x[k], x[k + 4]};
    __shared__ float lds_buf[256 * 4 + 64];
    lds_buf[threadIdx.x] = af;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(size_t)(lds_buf) + (threadIdx.x & 63) * 16;
    pk_f32x4 ld[2] = {};
    pk_f32x4 acc4[16] = {};
    pk_f32x16 acc16[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (F32) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc4[j], 0, 0, 0);
            else acc16[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc16[j & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k & 7]) : "v"(bf));
                else if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[k & 3]) : "v"(xq));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[k & 1]) : "v"(lds_addr), "i"(0) : "memory");
            }
            if (KIND == 2 && NV > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sink += acc4[j][0] + acc4[j][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) sink += acc16[j][0] + acc16[j][15];
#pragma unroll
    for (int k = 0; k < 8; ++k) sink += x[k];
    sink += xp[0][0] + xp[1][1] + xp[2][0] + xp[3][1] + ld[0][0] + ld[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
#endif
}

// HBM stream: mode 0 read-only (sum into one value per thread), mode 1 copy; grid-stride over float4
__global__ void __launch_bounds__(256)
stream_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4, int mode, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = in[i];
        if (mode) out[i] = v;
        else acc += v.x + v.y + v.z + v.w;
    }
    if (!mode) sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

/* launches `blocks` workgroups of 4 waves, each wave issuing iters x (4 or 8) MFMAs; flop per launch =
 * blocks*4*iters*4*32768 (modes 0, 1: 8*16384) or *4*4096 (mode 2).  out: blocks*256 floats. */
int scipnp_bench_mfma(float* out, int blocks, int iters, int mode, scipnp_stream_t s) {
    SCIPNP_REQUIRE(out && blocks > 0 && iters > 0 && mode >= 0 && mode <= 2, "bad arguments");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out, iters, mode, 12345u);
    return launch_status("mfma_peak_kernel");
}

/* `blocks` workgroups of 4 waves, each wave issuing iters x 16 MFMAs with nv (0, 1, 2, 4, 6, 8) v_add_f32 after each;
 * f32 = 1: v_mfma_f32_16x16x4_f32 (2: the fillers are v_pk_add_f32, 3: ds_read_b128 drained once per MFMA), 0: v_mfma_f32_32x32x16_f16.  cycles: blocks*4 words, s_memtime ticks of each wave's loop. */
int scipnp_bench_mfma_valu(float* out, unsigned long long* cycles, int blocks, int iters, int nv, int f32, scipnp_stream_t s) {
    SCIPNP_REQUIRE(out && cycles && blocks > 0 && iters > 0, "bad arguments");
    const dim3 g(blocks), b(256);
    hipStream_t st = (hipStream_t)s;
#define SCIPNP_MV(NV)                                                                                              \
    case NV:                                                                                                       \
        if (f32 == 1) hipLaunchKernelGGL((mfma_valu_kernel<NV, true>), g, b, 0, st, out, cycles, iters, 4321u);    \
        else if (f32 == 2) hipLaunchKernelGGL((mfma_valu_kernel<NV, true, 1>), g, b, 0, st, out, cycles, iters, 4321u); \
        else if (f32 == 3) hipLaunchKernelGGL((mfma_valu_kernel<NV, true, 2>), g, b, 0, st, out, cycles, iters, 4321u); \
        else hipLaunchKernelGGL((mfma_valu_kernel<NV, false>), g, b, 0, st, out, cycles, iters, 4321u);            \
        break;
    switch (nv) {
        SCIPNP_MV(0) SCIPNP_MV(1) SCIPNP_MV(2) SCIPNP_MV(4) SCIPNP_MV(6) SCIPNP_MV(8)
        default: return fail(SCIPNP_EINVAL, "nv must be 0, 1, 2, 4, 6 or 8");
    }
#undef SCIPNP_MV
    return launch_status("mfma_valu_kernel");
}

/* mode 0: read n floats of `in` (sink: blocks*256 floats); mode 1: copy n floats in -> out.  n % 4 == 0. */
int scipnp_bench_stream(const float* in, float* out, size_t n, int mode, int blocks, float* sink, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && n > 0 && n % 4 == 0 && blocks > 0 && (mode ? out != nullptr : sink != nullptr), "bad arguments");
    SCIPNP_ALIGNED(in);
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float4*)in, (float4*)out, n / 4, mode,
                       sink);
    return launch_status("stream_kernel");
}

}  // extern "C"
