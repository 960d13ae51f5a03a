// Micro-benchmarks that MEASURE the ceilings the rooflines are quoted against ("vendor figures; builder to
// confirm with a measured stream / MFMA microbenchmark and report both"): a register-resident MFMA loop on random
// operands (fp16 32x32x16, fp16 16x16x32, fp32 32x32x2) and an HBM stream (read-only reduction, copy).  Diagnostic
// entries of the library, driven by tools/peaks_bench.py; not on the reconstruction path.
#include "common.hpp"
#include "../../include/scipnp_diag.h"

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

// Issue patterns of v_mfma_f32_16x16x4_f32 accumulation chains (round 3): 16 accumulators per wave, 32 MFMAs per round --
// every accumulator is used twice -- with the second use DIST MFMAs behind the first (DIST = 1: back to back on the same
// accumulator; 2: the pairing (a0, a1, a0, a1), (a2, a3, a2, a3), ... of a kernel that multiplies two k-steps per position
// pair; 4, 8, 16: wider interleaves).  Cycles per wave by s_memtime.
template <int DIST>
__global__ void __launch_bounds__(256)
mfma_dep_kernel(float* __restrict__ out, unsigned long long* __restrict__ cycles, int iters, unsigned seed) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned id = (blockIdx.x * blockDim.x + threadIdx.x) * 16u + seed;
    const float af = hash_unit(id + 3), bf = hash_unit(id + 5), cf = hash_unit(id + 7), df = hash_unit(id + 9);
    pk_f32x4 acc[16] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int base = 0; base < 16; base += DIST) {
#pragma unroll
            for (int k = 0; k < DIST; ++k) acc[base + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[base + k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < DIST; ++k) acc[base + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(cf, df, acc[base + k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sink += acc[j][0] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
#endif
}

// VGPR-bank experiment for v_mfma_f32_16x16x4_f32 (round 3): the same 32 MFMAs per round on 16 accumulators (second use 16
// instructions behind the first), operands in EXPLICIT registers.  VAR 0: one A and one B register in different banks (register
// index mod 4); 1: the same bank; 2: four A / four B registers, A and B of every instruction in the SAME bank (u[nu], v[nu] of
// two float4 fragments -- what a kernel gets that multiplies element nu of one fragment by element nu of another); 3: the same
// registers rotated so that A and B never share a bank.  Pure assembly loops: the compiler allocates nothing in them.
#define SCIPNP_BANK_BODY(NAME, BODY)                                                                               \
    __global__ void __launch_bounds__(256) NAME(float* __restrict__ out, unsigned long long* __restrict__ cycles, int iters) { \
        unsigned long long t0 = 0, t1 = 0;                                                                        \
        float sink = 0.f;                                                                                         \
        (void)iters;                                                                                              \
        SCIPNP_BANK_ASM(BODY)                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = sink;                                                        \
        if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;              \
    }
#if defined(__HIP_DEVICE_COMPILE__)
#define SCIPNP_BANK_ASM(BODY)                                                                                      \
    t0 = __builtin_amdgcn_s_memtime();                                                                             \
    asm volatile(BODY : "=v"(sink) : "s"(iters) : "s20", "scc", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");                                         \
    t1 = __builtin_amdgcn_s_memtime();
#else
#define SCIPNP_BANK_ASM(BODY)
#endif
#define SCIPNP_BANK_0 \
        "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 0.5\n v_mov_b32 v66, 2.0\n v_mov_b32 v67, -1.0\n" \
        "v_mov_b32 v68, 0.5\n v_mov_b32 v69, 1.0\n v_mov_b32 v70, -0.5\n v_mov_b32 v71, 2.0\n" \
        "v_mov_b32 v0, 0\n" \
        "v_mov_b32 v1, 0\n" \
        "v_mov_b32 v2, 0\n" \
        "v_mov_b32 v3, 0\n" \
        "v_mov_b32 v4, 0\n" \
        "v_mov_b32 v5, 0\n" \
        "v_mov_b32 v6, 0\n" \
        "v_mov_b32 v7, 0\n" \
        "v_mov_b32 v8, 0\n" \
        "v_mov_b32 v9, 0\n" \
        "v_mov_b32 v10, 0\n" \
        "v_mov_b32 v11, 0\n" \
        "v_mov_b32 v12, 0\n" \
        "v_mov_b32 v13, 0\n" \
        "v_mov_b32 v14, 0\n" \
        "v_mov_b32 v15, 0\n" \
        "v_mov_b32 v16, 0\n" \
        "v_mov_b32 v17, 0\n" \
        "v_mov_b32 v18, 0\n" \
        "v_mov_b32 v19, 0\n" \
        "v_mov_b32 v20, 0\n" \
        "v_mov_b32 v21, 0\n" \
        "v_mov_b32 v22, 0\n" \
        "v_mov_b32 v23, 0\n" \
        "v_mov_b32 v24, 0\n" \
        "v_mov_b32 v25, 0\n" \
        "v_mov_b32 v26, 0\n" \
        "v_mov_b32 v27, 0\n" \
        "v_mov_b32 v28, 0\n" \
        "v_mov_b32 v29, 0\n" \
        "v_mov_b32 v30, 0\n" \
        "v_mov_b32 v31, 0\n" \
        "v_mov_b32 v32, 0\n" \
        "v_mov_b32 v33, 0\n" \
        "v_mov_b32 v34, 0\n" \
        "v_mov_b32 v35, 0\n" \
        "v_mov_b32 v36, 0\n" \
        "v_mov_b32 v37, 0\n" \
        "v_mov_b32 v38, 0\n" \
        "v_mov_b32 v39, 0\n" \
        "v_mov_b32 v40, 0\n" \
        "v_mov_b32 v41, 0\n" \
        "v_mov_b32 v42, 0\n" \
        "v_mov_b32 v43, 0\n" \
        "v_mov_b32 v44, 0\n" \
        "v_mov_b32 v45, 0\n" \
        "v_mov_b32 v46, 0\n" \
        "v_mov_b32 v47, 0\n" \
        "v_mov_b32 v48, 0\n" \
        "v_mov_b32 v49, 0\n" \
        "v_mov_b32 v50, 0\n" \
        "v_mov_b32 v51, 0\n" \
        "v_mov_b32 v52, 0\n" \
        "v_mov_b32 v53, 0\n" \
        "v_mov_b32 v54, 0\n" \
        "v_mov_b32 v55, 0\n" \
        "v_mov_b32 v56, 0\n" \
        "v_mov_b32 v57, 0\n" \
        "v_mov_b32 v58, 0\n" \
        "v_mov_b32 v59, 0\n" \
        "v_mov_b32 v60, 0\n" \
        "v_mov_b32 v61, 0\n" \
        "v_mov_b32 v62, 0\n" \
        "v_mov_b32 v63, 0\n" \
        "s_mov_b32 s20, %1\n" \
        "1:\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v64, v69, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v64, v69, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v64, v69, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v64, v69, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v64, v69, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v64, v69, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v64, v69, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v64, v69, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v64, v69, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v64, v69, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v64, v69, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v64, v69, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v64, v69, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v64, v69, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v64, v69, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v64, v69, v[60:63]\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v64, v69, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v64, v69, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v64, v69, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v64, v69, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v64, v69, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v64, v69, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v64, v69, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v64, v69, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v64, v69, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v64, v69, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v64, v69, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v64, v69, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v64, v69, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v64, v69, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v64, v69, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v64, v69, v[60:63]\n" \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
        "v_add_f32 %0, v0, v63\n"
SCIPNP_BANK_BODY(mfma_bank_kernel_0, SCIPNP_BANK_0)
#define SCIPNP_BANK_1 \
        "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 0.5\n v_mov_b32 v66, 2.0\n v_mov_b32 v67, -1.0\n" \
        "v_mov_b32 v68, 0.5\n v_mov_b32 v69, 1.0\n v_mov_b32 v70, -0.5\n v_mov_b32 v71, 2.0\n" \
        "v_mov_b32 v0, 0\n" \
        "v_mov_b32 v1, 0\n" \
        "v_mov_b32 v2, 0\n" \
        "v_mov_b32 v3, 0\n" \
        "v_mov_b32 v4, 0\n" \
        "v_mov_b32 v5, 0\n" \
        "v_mov_b32 v6, 0\n" \
        "v_mov_b32 v7, 0\n" \
        "v_mov_b32 v8, 0\n" \
        "v_mov_b32 v9, 0\n" \
        "v_mov_b32 v10, 0\n" \
        "v_mov_b32 v11, 0\n" \
        "v_mov_b32 v12, 0\n" \
        "v_mov_b32 v13, 0\n" \
        "v_mov_b32 v14, 0\n" \
        "v_mov_b32 v15, 0\n" \
        "v_mov_b32 v16, 0\n" \
        "v_mov_b32 v17, 0\n" \
        "v_mov_b32 v18, 0\n" \
        "v_mov_b32 v19, 0\n" \
        "v_mov_b32 v20, 0\n" \
        "v_mov_b32 v21, 0\n" \
        "v_mov_b32 v22, 0\n" \
        "v_mov_b32 v23, 0\n" \
        "v_mov_b32 v24, 0\n" \
        "v_mov_b32 v25, 0\n" \
        "v_mov_b32 v26, 0\n" \
        "v_mov_b32 v27, 0\n" \
        "v_mov_b32 v28, 0\n" \
        "v_mov_b32 v29, 0\n" \
        "v_mov_b32 v30, 0\n" \
        "v_mov_b32 v31, 0\n" \
        "v_mov_b32 v32, 0\n" \
        "v_mov_b32 v33, 0\n" \
        "v_mov_b32 v34, 0\n" \
        "v_mov_b32 v35, 0\n" \
        "v_mov_b32 v36, 0\n" \
        "v_mov_b32 v37, 0\n" \
        "v_mov_b32 v38, 0\n" \
        "v_mov_b32 v39, 0\n" \
        "v_mov_b32 v40, 0\n" \
        "v_mov_b32 v41, 0\n" \
        "v_mov_b32 v42, 0\n" \
        "v_mov_b32 v43, 0\n" \
        "v_mov_b32 v44, 0\n" \
        "v_mov_b32 v45, 0\n" \
        "v_mov_b32 v46, 0\n" \
        "v_mov_b32 v47, 0\n" \
        "v_mov_b32 v48, 0\n" \
        "v_mov_b32 v49, 0\n" \
        "v_mov_b32 v50, 0\n" \
        "v_mov_b32 v51, 0\n" \
        "v_mov_b32 v52, 0\n" \
        "v_mov_b32 v53, 0\n" \
        "v_mov_b32 v54, 0\n" \
        "v_mov_b32 v55, 0\n" \
        "v_mov_b32 v56, 0\n" \
        "v_mov_b32 v57, 0\n" \
        "v_mov_b32 v58, 0\n" \
        "v_mov_b32 v59, 0\n" \
        "v_mov_b32 v60, 0\n" \
        "v_mov_b32 v61, 0\n" \
        "v_mov_b32 v62, 0\n" \
        "v_mov_b32 v63, 0\n" \
        "s_mov_b32 s20, %1\n" \
        "1:\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v64, v68, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v64, v68, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v64, v68, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v64, v68, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v64, v68, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v64, v68, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v64, v68, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v64, v68, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v64, v68, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v64, v68, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v64, v68, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v64, v68, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v64, v68, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v64, v68, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v64, v68, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v64, v68, v[60:63]\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v64, v68, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v64, v68, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v64, v68, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v64, v68, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v64, v68, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v64, v68, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v64, v68, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v64, v68, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v64, v68, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v64, v68, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v64, v68, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v64, v68, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v64, v68, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v64, v68, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v64, v68, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v64, v68, v[60:63]\n" \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
        "v_add_f32 %0, v0, v63\n"
SCIPNP_BANK_BODY(mfma_bank_kernel_1, SCIPNP_BANK_1)
#define SCIPNP_BANK_2 \
        "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 0.5\n v_mov_b32 v66, 2.0\n v_mov_b32 v67, -1.0\n" \
        "v_mov_b32 v68, 0.5\n v_mov_b32 v69, 1.0\n v_mov_b32 v70, -0.5\n v_mov_b32 v71, 2.0\n" \
        "v_mov_b32 v0, 0\n" \
        "v_mov_b32 v1, 0\n" \
        "v_mov_b32 v2, 0\n" \
        "v_mov_b32 v3, 0\n" \
        "v_mov_b32 v4, 0\n" \
        "v_mov_b32 v5, 0\n" \
        "v_mov_b32 v6, 0\n" \
        "v_mov_b32 v7, 0\n" \
        "v_mov_b32 v8, 0\n" \
        "v_mov_b32 v9, 0\n" \
        "v_mov_b32 v10, 0\n" \
        "v_mov_b32 v11, 0\n" \
        "v_mov_b32 v12, 0\n" \
        "v_mov_b32 v13, 0\n" \
        "v_mov_b32 v14, 0\n" \
        "v_mov_b32 v15, 0\n" \
        "v_mov_b32 v16, 0\n" \
        "v_mov_b32 v17, 0\n" \
        "v_mov_b32 v18, 0\n" \
        "v_mov_b32 v19, 0\n" \
        "v_mov_b32 v20, 0\n" \
        "v_mov_b32 v21, 0\n" \
        "v_mov_b32 v22, 0\n" \
        "v_mov_b32 v23, 0\n" \
        "v_mov_b32 v24, 0\n" \
        "v_mov_b32 v25, 0\n" \
        "v_mov_b32 v26, 0\n" \
        "v_mov_b32 v27, 0\n" \
        "v_mov_b32 v28, 0\n" \
        "v_mov_b32 v29, 0\n" \
        "v_mov_b32 v30, 0\n" \
        "v_mov_b32 v31, 0\n" \
        "v_mov_b32 v32, 0\n" \
        "v_mov_b32 v33, 0\n" \
        "v_mov_b32 v34, 0\n" \
        "v_mov_b32 v35, 0\n" \
        "v_mov_b32 v36, 0\n" \
        "v_mov_b32 v37, 0\n" \
        "v_mov_b32 v38, 0\n" \
        "v_mov_b32 v39, 0\n" \
        "v_mov_b32 v40, 0\n" \
        "v_mov_b32 v41, 0\n" \
        "v_mov_b32 v42, 0\n" \
        "v_mov_b32 v43, 0\n" \
        "v_mov_b32 v44, 0\n" \
        "v_mov_b32 v45, 0\n" \
        "v_mov_b32 v46, 0\n" \
        "v_mov_b32 v47, 0\n" \
        "v_mov_b32 v48, 0\n" \
        "v_mov_b32 v49, 0\n" \
        "v_mov_b32 v50, 0\n" \
        "v_mov_b32 v51, 0\n" \
        "v_mov_b32 v52, 0\n" \
        "v_mov_b32 v53, 0\n" \
        "v_mov_b32 v54, 0\n" \
        "v_mov_b32 v55, 0\n" \
        "v_mov_b32 v56, 0\n" \
        "v_mov_b32 v57, 0\n" \
        "v_mov_b32 v58, 0\n" \
        "v_mov_b32 v59, 0\n" \
        "v_mov_b32 v60, 0\n" \
        "v_mov_b32 v61, 0\n" \
        "v_mov_b32 v62, 0\n" \
        "v_mov_b32 v63, 0\n" \
        "s_mov_b32 s20, %1\n" \
        "1:\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v64, v68, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v65, v69, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v66, v70, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v67, v71, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v64, v68, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v65, v69, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v66, v70, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v67, v71, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v64, v68, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v65, v69, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v66, v70, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v67, v71, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v64, v68, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v65, v69, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v66, v70, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v67, v71, v[60:63]\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v65, v69, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v66, v70, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v67, v71, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v64, v68, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v65, v69, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v66, v70, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v67, v71, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v64, v68, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v65, v69, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v66, v70, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v67, v71, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v64, v68, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v65, v69, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v66, v70, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v67, v71, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v64, v68, v[60:63]\n" \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
        "v_add_f32 %0, v0, v63\n"
SCIPNP_BANK_BODY(mfma_bank_kernel_2, SCIPNP_BANK_2)
#define SCIPNP_BANK_3 \
        "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 0.5\n v_mov_b32 v66, 2.0\n v_mov_b32 v67, -1.0\n" \
        "v_mov_b32 v68, 0.5\n v_mov_b32 v69, 1.0\n v_mov_b32 v70, -0.5\n v_mov_b32 v71, 2.0\n" \
        "v_mov_b32 v0, 0\n" \
        "v_mov_b32 v1, 0\n" \
        "v_mov_b32 v2, 0\n" \
        "v_mov_b32 v3, 0\n" \
        "v_mov_b32 v4, 0\n" \
        "v_mov_b32 v5, 0\n" \
        "v_mov_b32 v6, 0\n" \
        "v_mov_b32 v7, 0\n" \
        "v_mov_b32 v8, 0\n" \
        "v_mov_b32 v9, 0\n" \
        "v_mov_b32 v10, 0\n" \
        "v_mov_b32 v11, 0\n" \
        "v_mov_b32 v12, 0\n" \
        "v_mov_b32 v13, 0\n" \
        "v_mov_b32 v14, 0\n" \
        "v_mov_b32 v15, 0\n" \
        "v_mov_b32 v16, 0\n" \
        "v_mov_b32 v17, 0\n" \
        "v_mov_b32 v18, 0\n" \
        "v_mov_b32 v19, 0\n" \
        "v_mov_b32 v20, 0\n" \
        "v_mov_b32 v21, 0\n" \
        "v_mov_b32 v22, 0\n" \
        "v_mov_b32 v23, 0\n" \
        "v_mov_b32 v24, 0\n" \
        "v_mov_b32 v25, 0\n" \
        "v_mov_b32 v26, 0\n" \
        "v_mov_b32 v27, 0\n" \
        "v_mov_b32 v28, 0\n" \
        "v_mov_b32 v29, 0\n" \
        "v_mov_b32 v30, 0\n" \
        "v_mov_b32 v31, 0\n" \
        "v_mov_b32 v32, 0\n" \
        "v_mov_b32 v33, 0\n" \
        "v_mov_b32 v34, 0\n" \
        "v_mov_b32 v35, 0\n" \
        "v_mov_b32 v36, 0\n" \
        "v_mov_b32 v37, 0\n" \
        "v_mov_b32 v38, 0\n" \
        "v_mov_b32 v39, 0\n" \
        "v_mov_b32 v40, 0\n" \
        "v_mov_b32 v41, 0\n" \
        "v_mov_b32 v42, 0\n" \
        "v_mov_b32 v43, 0\n" \
        "v_mov_b32 v44, 0\n" \
        "v_mov_b32 v45, 0\n" \
        "v_mov_b32 v46, 0\n" \
        "v_mov_b32 v47, 0\n" \
        "v_mov_b32 v48, 0\n" \
        "v_mov_b32 v49, 0\n" \
        "v_mov_b32 v50, 0\n" \
        "v_mov_b32 v51, 0\n" \
        "v_mov_b32 v52, 0\n" \
        "v_mov_b32 v53, 0\n" \
        "v_mov_b32 v54, 0\n" \
        "v_mov_b32 v55, 0\n" \
        "v_mov_b32 v56, 0\n" \
        "v_mov_b32 v57, 0\n" \
        "v_mov_b32 v58, 0\n" \
        "v_mov_b32 v59, 0\n" \
        "v_mov_b32 v60, 0\n" \
        "v_mov_b32 v61, 0\n" \
        "v_mov_b32 v62, 0\n" \
        "v_mov_b32 v63, 0\n" \
        "s_mov_b32 s20, %1\n" \
        "1:\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v64, v69, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v65, v70, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v66, v71, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v67, v68, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v64, v69, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v65, v70, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v66, v71, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v67, v68, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v64, v69, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v65, v70, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v66, v71, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v67, v68, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v64, v69, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v65, v70, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v66, v71, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v67, v68, v[60:63]\n" \
        "v_mfma_f32_16x16x4_f32 v[0:3], v65, v70, v[0:3]\n" \
        "v_mfma_f32_16x16x4_f32 v[4:7], v66, v71, v[4:7]\n" \
        "v_mfma_f32_16x16x4_f32 v[8:11], v67, v68, v[8:11]\n" \
        "v_mfma_f32_16x16x4_f32 v[12:15], v64, v69, v[12:15]\n" \
        "v_mfma_f32_16x16x4_f32 v[16:19], v65, v70, v[16:19]\n" \
        "v_mfma_f32_16x16x4_f32 v[20:23], v66, v71, v[20:23]\n" \
        "v_mfma_f32_16x16x4_f32 v[24:27], v67, v68, v[24:27]\n" \
        "v_mfma_f32_16x16x4_f32 v[28:31], v64, v69, v[28:31]\n" \
        "v_mfma_f32_16x16x4_f32 v[32:35], v65, v70, v[32:35]\n" \
        "v_mfma_f32_16x16x4_f32 v[36:39], v66, v71, v[36:39]\n" \
        "v_mfma_f32_16x16x4_f32 v[40:43], v67, v68, v[40:43]\n" \
        "v_mfma_f32_16x16x4_f32 v[44:47], v64, v69, v[44:47]\n" \
        "v_mfma_f32_16x16x4_f32 v[48:51], v65, v70, v[48:51]\n" \
        "v_mfma_f32_16x16x4_f32 v[52:55], v66, v71, v[52:55]\n" \
        "v_mfma_f32_16x16x4_f32 v[56:59], v67, v68, v[56:59]\n" \
        "v_mfma_f32_16x16x4_f32 v[60:63], v64, v69, v[60:63]\n" \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
        "v_add_f32 %0, v0, v63\n"
SCIPNP_BANK_BODY(mfma_bank_kernel_3, SCIPNP_BANK_3)

// HBM stream: mode 0 read-only (sum into one value per thread), mode 1 copy, mode 2 write-only (a value made from the index: what a
// layer that writes many more bytes than it reads is bounded by), mode 3 write-only with the nt hint; grid-stride over float4
__global__ void __launch_bounds__(256)
stream_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4, int mode, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    if (mode >= 2) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const float f = (float)(unsigned)i;
            const float4 v = make_float4(f, f + 1.f, f + 2.f, f + 3.f);
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            if (mode == 3) __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, (f32x4_t*)(out + i));
            else out[i] = v;
        }
        return;
    }
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

/* `blocks` workgroups of 4 waves, each wave issuing iters x 32 v_mfma_f32_16x16x4_f32 on 16 accumulators, the second use of an
 * accumulator `dist` (1, 2, 4, 8, 16) MFMAs behind the first.  cycles: blocks*4 words (s_memtime ticks per wave). */
int scipnp_bench_mfma_dep(float* out, unsigned long long* cycles, int blocks, int iters, int dist, scipnp_stream_t s) {
    SCIPNP_REQUIRE(out && cycles && blocks > 0 && iters > 0, "bad arguments");
    const dim3 g(blocks), b(256);
    hipStream_t st = (hipStream_t)s;
    switch (dist) {
        case 1: hipLaunchKernelGGL((mfma_dep_kernel<1>), g, b, 0, st, out, cycles, iters, 99u); break;
        case 2: hipLaunchKernelGGL((mfma_dep_kernel<2>), g, b, 0, st, out, cycles, iters, 99u); break;
        case 4: hipLaunchKernelGGL((mfma_dep_kernel<4>), g, b, 0, st, out, cycles, iters, 99u); break;
        case 8: hipLaunchKernelGGL((mfma_dep_kernel<8>), g, b, 0, st, out, cycles, iters, 99u); break;
        case 16: hipLaunchKernelGGL((mfma_dep_kernel<16>), g, b, 0, st, out, cycles, iters, 99u); break;
        default: return fail(SCIPNP_EINVAL, "dist must be 1, 2, 4, 8 or 16");
    }
    return launch_status("mfma_dep_kernel");
}

/* VGPR-bank experiment: `blocks` workgroups of 4 waves, iters x 32 v_mfma_f32_16x16x4_f32 per wave, operand registers by `var`
 * (0 different banks, 1 same bank, 2 same-bank float4 fragments, 3 rotated fragments); cycles: blocks*4 words of s_memtime ticks */
int scipnp_bench_mfma_bank(float* out, unsigned long long* cycles, int blocks, int iters, int var, scipnp_stream_t s) {
    SCIPNP_REQUIRE(out && cycles && blocks > 0 && iters > 0, "bad arguments");
    const dim3 g(blocks), b(256);
    hipStream_t st = (hipStream_t)s;
    switch (var) {
        case 0: hipLaunchKernelGGL(mfma_bank_kernel_0, g, b, 0, st, out, cycles, iters); break;
        case 1: hipLaunchKernelGGL(mfma_bank_kernel_1, g, b, 0, st, out, cycles, iters); break;
        case 2: hipLaunchKernelGGL(mfma_bank_kernel_2, g, b, 0, st, out, cycles, iters); break;
        case 3: hipLaunchKernelGGL(mfma_bank_kernel_3, g, b, 0, st, out, cycles, iters); break;
        default: return fail(SCIPNP_EINVAL, "var must be 0..3");
    }
    return launch_status("mfma_bank_kernel");
}

/* mode 0: read n floats of `in` (sink: blocks*256 floats); mode 1: copy n floats in -> out; mode 2 / 3: write n floats of `out`
 * (plain / nt stores; `in` is not read but must be a valid pointer).  n % 4 == 0. */
int scipnp_bench_stream(const float* in, float* out, size_t n, int mode, int blocks, float* sink, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && n > 0 && n % 4 == 0 && blocks > 0 && (mode ? out != nullptr : sink != nullptr), "bad arguments");
    SCIPNP_ALIGNED(in);
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float4*)in, (float4*)out, n / 4, mode,
                       sink);
    return launch_status("stream_kernel");
}

}  // extern "C"
