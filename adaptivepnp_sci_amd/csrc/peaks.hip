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

/* mode 0: read n floats of `in` (sink: blocks*256 floats); mode 1: copy n floats in -> out.  n % 4 == 0. */
int scipnp_bench_stream(const float* in, float* out, size_t n, int mode, int blocks, float* sink, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && n > 0 && n % 4 == 0 && blocks > 0 && (mode ? out != nullptr : sink != nullptr), "bad arguments");
    SCIPNP_ALIGNED(in);
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float4*)in, (float4*)out, n / 4, mode,
                       sink);
    return launch_status("stream_kernel");
}

}  // extern "C"
