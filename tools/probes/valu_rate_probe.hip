// GPU box: issue cost of the vector instructions the TV kernels are made of (clocks per wave instruction, one wave per SIMD
// and four waves per SIMD): s_memtime around 8 independent chains of 256 instructions each.
//   hipcc --offload-arch=gfx950 -O3 -o build/variants/valu_rate_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define REP 256

template <int OP>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* clk, float seed) {
    float a[CHAINS];
    double d[CHAINS];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CHAINS];
    unsigned long long mask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1), m2[2] = {0, 0};
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; p[i] = f2{a[i], a[i] + 1}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % CHAINS]));
            if (OP == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) % CHAINS]));
            if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed));
            if (OP == 8) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 1) % CHAINS]));
            if (OP == 9) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(seed) : "vcc");
            if (OP == 10) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) % CHAINS]));
            if (OP == 12) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) % CHAINS]));
            if (OP == 13) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 14) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(seed) : "vcc");
            if (OP == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "s"(mask));
            if (OP == 16) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2[i & 1]) : "v"(a[i]), "v"(seed));
            if (OP == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 18) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 19) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 20) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]), "s"(mask));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)(m2[0] + m2[1]);
    for (int i = 0; i < CHAINS; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, unsigned long long* clk) {
    for (int waves_per_simd : {1}) {
        const int threads = 256, blocks = 256 * waves_per_simd;       // 4 waves per block = one per SIMD; blocks per CU = waves_per_simd
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, clk, 1.5f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, clk, 1.5f);
        hipDeviceSynchronize();
        unsigned long long c = 0;
        hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%-16s %d wave(s)/SIMD: %6.2f clocks per instruction of one wave\n", name, waves_per_simd, (double)c / (CHAINS * REP));
    }
}

int main() {
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    hipMalloc(&clk, 8);
    run<0>("v_add_f32", out, clk);
    run<6>("v_fma_f32", out, clk);
    run<5>("v_pk_fma_f32", out, clk);
    run<11>("v_pk_mul_f32", out, clk);
    run<1>("v_cvt_f64_f32", out, clk);
    run<2>("v_add_f64", out, clk);
    run<12>("v_fma_f64", out, clk);
    run<3>("v_sqrt_f32", out, clk);
    run<4>("v_rcp_f32", out, clk);
    run<7>("v_cndmask_b32", out, clk);
    run<8>("v_mov_b32_dpp", out, clk);
    run<9>("v_div_scale_f32", out, clk);
    run<10>("v_div_fixup_f32", out, clk);
    run<13>("v_min3_u32", out, clk);
    run<14>("v_cmp_lt_f32", out, clk);
    run<15>("v_cndmask_e64 sgpr", out, clk);
    run<20>("v_cndmask_e64 2 vgpr", out, clk);
    run<16>("v_cmp_e64 sgpr", out, clk);
    run<17>("v_add_u32", out, clk);
    run<18>("v_lshl_add_u32", out, clk);
    run<19>("v_mul_f32", out, clk);
    return 0;
}
