// GPU box: what a barrier among the workgroups of ONE XCD costs (the ADMM-TV iteration as one persistent launch per
// solve, the 64 workgroups of a Bayer plane placed on one XCD and synchronised through that XCD's L2 instead of a kernel boundary).
//   hipcc --offload-arch=gfx950 -O3 -o build/variants/xcd_barrier_probe tools/probes/xcd_barrier_probe.hip
// Every workgroup reads its XCC id (s_getreg HW_REG_XCC_ID), takes a slot on its XCD (an atomic counter per XCD) and then runs ITERS
// rounds of:  a little work (WORK dependent v_fma per thread, to open the arrival skew a real phase has)  ->  stores of its 2 KB of
// "state" (plain stores, drained with s_waitcnt vmcnt(0))  ->  barrier among the PER_XCD workgroups of its XCD (one monotonic counter
// per XCD in device memory, agent-scope atomic add by lane 0, sc1-load poll with s_sleep, bounded)  ->  sc1 loads of a neighbour
// workgroup's state written before the barrier (checked: a stale value is counted).  Workgroups beyond PER_XCD on an XCD exit at once.
// Reported: microseconds per round with the barrier, without it (the work + stores + loads alone), the difference = the barrier,
// for 32 and 64 workgroups per XCD on 8 and on 4 XCDs; stale reads; rounds that hit the poll bound (must be 0).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Ctl {
    unsigned slot[8];            // next free slot per XCD
    unsigned pad0[24];
    unsigned arrive[8][32];      // one counter per XCD, on a line of its own
    unsigned stale, bound_hit;
};

template <bool BARRIER>
__global__ void __launch_bounds__(256) probe(Ctl* ctl, float* state, unsigned long long* clk, int per_xcd, int xcds, int iters, int work) {
    __shared__ int s_slot, s_xcc;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        s_xcc = (int)xcc;
        s_slot = (xcc < (unsigned)xcds) ? (int)atomicAdd(&ctl->slot[xcc], 1u) : per_xcd;
    }
    __syncthreads();
    const int xcc = s_xcc, slot = s_slot;
    if (slot >= per_xcd) return;                                   // surplus workgroups of this XCD (and the XCDs not taking part)
    float* mine = state + ((size_t)xcc * 64 + slot) * 512;         // 2 KB per workgroup
    const float* nb = state + ((size_t)xcc * 64 + (slot + 1) % per_xcd) * 512;
    float acc = 1.0f + tid * 1e-6f;
    unsigned stale = 0, bound = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 1; it <= iters; ++it) {
        for (int k = 0; k < work + ((slot * 7 + it) & 15) * 8; ++k) acc = __builtin_fmaf(acc, 0.999f, 0.001f);     // skewed arrival
        __builtin_nontemporal_store((float)it + acc * 0.f, mine + tid);
        __builtin_nontemporal_store((float)it, mine + 256 + tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (BARRIER) {
            if (tid == 0) {
                __hip_atomic_fetch_add(&ctl->arrive[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)it * (unsigned)per_xcd;
                unsigned polls = 0;
                while (__hip_atomic_load(&ctl->arrive[xcc][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > 200000u) { bound = 1; break; }
                }
            }
            __syncthreads();
            // the neighbour's state of THIS round: L1-bypassing loads (the XCD's L2 is the point of coherence between its CUs)
            const float v = __hip_atomic_load(nb + 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stale += (v != (float)it);
            acc += v * 1e-9f;
        } else {
            const float v = __hip_atomic_load(nb + 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += v * 1e-9f;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) clk[xcc * 64 + slot] = t1 - t0;
    if (stale) atomicAdd(&ctl->stale, stale);
    if (bound && tid == 0) atomicAdd(&ctl->bound_hit, 1u);
    if (acc == 12345.678f) state[0] = acc;
}

int main() {
    Ctl* ctl; float* state; unsigned long long* clk;
    CHECK(hipMalloc(&ctl, sizeof(Ctl))); CHECK(hipMalloc(&state, 8 * 64 * 512 * 4)); CHECK(hipMalloc(&clk, 8 * 64 * 8));
    const int iters = 200;
    for (int work : {200, 4000})
        for (int xcds : {8, 4})
            for (int per : {32, 64}) {
                double us[2] = {0, 0};
                unsigned stale = 0, bound = 0;
                int counted = 0;
                for (int bar = 0; bar < 2; ++bar) {
                    for (int rep = 0; rep < 3; ++rep) {
                        CHECK(hipMemset(ctl, 0, sizeof(Ctl))); CHECK(hipMemset(clk, 0, 8 * 64 * 8));
                        // 4 x the slots needed: the dispatcher deals workgroups round-robin over the XCDs, the surplus exits at once
                        const int blocks = 8 * per * 4;
                        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                        CHECK(hipEventRecord(e0));
                        if (bar) hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(256), 0, 0, ctl, state, clk, per, xcds, iters, work);
                        else hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(256), 0, 0, ctl, state, clk, per, xcds, iters, work);
                        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                        us[bar] = ms * 1e3 / iters;
                        Ctl h; CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
                        if (bar) { stale = h.stale; bound = h.bound_hit; }
                        std::vector<unsigned long long> c(8 * 64); CHECK(hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost));
                        counted = (int)std::count_if(c.begin(), c.end(), [](unsigned long long v) { return v != 0; });
                    }
                }
                printf("work %4d fma | %d XCDs x %2d workgroups (%3d took part): %6.2f us per round with the XCD-local barrier, %6.2f without -> barrier %5.2f us; "
                       "stale reads %u, poll bound hit by %u workgroups\n", work, xcds, per, counted, us[1], us[0], us[1] - us[0], stale, bound);
            }
    return 0;
}
