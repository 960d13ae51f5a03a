// Shared helpers for the scipnp HIP sources (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/scipnp.h"

#include "host_common.hpp"

namespace scipnp {

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SCIPNP_EHIP, "%s: %s", what, hipGetErrorString(e));
    return SCIPNP_OK;
}

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

// The same two orders with the frame count known only at run time, `term(i)` a memory load; T is float or float4 (four
// independent sums).  Both follow ATen's cascade_sum (SumKernel.cpp, torch 2.10: row_sum = multi_row_sum over groups of
// four with level step 16):
//   strided: element i feeds accumulator i % 4; after every 16 groups of four the accumulators are flushed into the next
//            level (and that one after 16 flushes, ...); leftovers of the last partial run stay in level 0; the levels are
//            added to level 0 in order, the tail elements (past the last full group of four) to accumulator 0, result
//            ((a0+a1)+a2)+a3.  Below 64 addends no flush ever happens and this is the plain four-accumulator sum of (1).
//   contiguous: the same over 8-float VECTORS per lane (the flush needs 64 vectors = 512 addends: not restated, hence
//            TORCH_SUM_RT_MAX), then the lane sums are added to the tail sum in lane order, as in (2).
constexpr int TORCH_SUM_RT_MAX = 511;
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float f4_add(float a, float b) { return a + b; }

template <typename T, typename F>
__device__ __forceinline__ T torch_strided_sum_rt(int nB, T zero, F term) {
    T a0 = zero, a1 = zero, a2 = zero, a3 = zero;           // level 0
    T u0 = zero, u1 = zero, u2 = zero, u3 = zero;           // level 1 (level 2 would need 256 groups = 1024 addends)
    const int ngrp = nB >> 2;
    int g = 0;
    for (; g + 16 <= ngrp; g += 16) {
        for (int j = 0; j < 16; ++j) {
            const int i = 4 * (g + j);
            a0 = f4_add(a0, term(i));
            a1 = f4_add(a1, term(i + 1));
            a2 = f4_add(a2, term(i + 2));
            a3 = f4_add(a3, term(i + 3));
        }
        u0 = f4_add(u0, a0); u1 = f4_add(u1, a1); u2 = f4_add(u2, a2); u3 = f4_add(u3, a3);
        a0 = zero; a1 = zero; a2 = zero; a3 = zero;
    }
    for (; g < ngrp; ++g) {
        const int i = 4 * g;
        a0 = f4_add(a0, term(i));
        a1 = f4_add(a1, term(i + 1));
        a2 = f4_add(a2, term(i + 2));
        a3 = f4_add(a3, term(i + 3));
    }
    if (ngrp >= 16) { a0 = f4_add(a0, u0); a1 = f4_add(a1, u1); a2 = f4_add(a2, u2); a3 = f4_add(a3, u3); }
    for (int i = 4 * ngrp; i < nB; ++i) a0 = f4_add(a0, term(i));
    return f4_add(f4_add(f4_add(a0, a1), a2), a3);
}

template <typename T, typename F>
__device__ __forceinline__ T torch_contig_sum_rt(int nB, T zero, F term) {
    if (nB < 8) return torch_strided_sum_rt(nB, zero, term);
    const int nv = nB >> 3, nq = nv >> 2;
    T f = zero;
    for (int i = 8 * nv; i < nB; ++i) f = f4_add(f, term(i));
    for (int k = 0; k < 8; ++k) {
        T c0 = zero, c1 = zero, c2 = zero, c3 = zero;
        for (int i = 0; i < nq; ++i) {
            c0 = f4_add(c0, term(8 * (4 * i) + k));
            c1 = f4_add(c1, term(8 * (4 * i + 1) + k));
            c2 = f4_add(c2, term(8 * (4 * i + 2) + k));
            c3 = f4_add(c3, term(8 * (4 * i + 3) + k));
        }
        for (int j = 4 * nq; j < nv; ++j) c0 = f4_add(c0, term(8 * j + k));
        f = f4_add(f, f4_add(f4_add(f4_add(c0, c1), c2), c3));
    }
    return f;
}

// range-guard word of the split-fp16 kernels (csrc/conv_split.hip): binds `w` for the calling thread's following launches
// and returns the previous binding (nullptr = the process-wide word)
int* exchange_overflow_word(int* w);
struct OverflowScope {                 // binds a word for one call's launches when given one; restores the thread's binding
    int* prev = nullptr;
    bool on;
    explicit OverflowScope(int* w) : on(w != nullptr) { if (on) prev = exchange_overflow_word(w); }
    ~OverflowScope() { if (on) exchange_overflow_word(prev); }
    OverflowScope(const OverflowScope&) = delete;
    OverflowScope& operator=(const OverflowScope&) = delete;
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one of these per launch site sets it
// once per device the site is used on; racing first calls from several host threads both set it (idempotent).
struct LdsAttrOnce {
    std::atomic<unsigned long long> done{0};
    int ensure(const void* fn, size_t bytes, const char* what) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(SCIPNP_EHIP, "hipGetDevice (%s)", what);
        const unsigned long long bit = 1ull << dev;
        if (done.load(std::memory_order_acquire) & bit) return SCIPNP_OK;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(SCIPNP_EHIP, "hipFuncSetAttribute(%s, %zu B LDS): %s", what, bytes, hipGetErrorString(e));
        done.fetch_or(bit, std::memory_order_release);
        return SCIPNP_OK;
    }
};

// tv.hip: can the banded one-launch Chambolle kernel take planes of M x N with n_iter iterations?
bool tv_band_fits(int M, int N, int n_iter);
// tv.hip: the CANDIDATE form of the banded kernel (one launch, nothing recomputed, no communication between bands): `out` of
// every iteration and every band's partial energy sums go to the workspace; the consumer evaluates the stop test per channel
// (tv_band_stop_test, by one full wave) and reads theta_raw = cand[(stop - 1) * C*M*N + i]
struct TvCandidates {
    const float* cand;      // [n_iter - 1][C][M][N]
    const double* part;     // [C][nbands][n_iter][2]
    int32_t* stop;          // [C] scratch for tv_stop_test_launch
    int nbands, n_iter;
    size_t MN;
};
bool tv_candidates_fit(int M, int N, int n_iter);
void tv_candidate_ptrs(int M, int N, int C, int n_iter, void* workspace, TvCandidates* out);
int tv_band_candidates(const float* x, const float* b, float coef, int M, int N, int C, float weight, float eps, int n_iter,
                       void* workspace, size_t workspace_bytes, hipStream_t st);
int tv_stop_test_launch(const TvCandidates& cd, int C, float weight, float eps, hipStream_t st);
double tv_scalar_as_double(float f);     // weight / eps: the shortest decimal that round-trips the float (tv.hip as_double)

// A/B switches of the laboratory (the round-2 schedules of the TV kernels, the general dual-update + projection kernel, band /
// batch geometries): the PRODUCT library reads no environment variable -- every choice of form it offers is an argument
// (scipnp.h) or a field of adaptivepnp_sci_amd.config.Config.  `make tvvariant NAME=lab TVFLAGS=-DSCIPNP_LAB_SWITCHES` builds a
// library in which lab_switch("SCIPNP_...") looks the variable up (tools/gpu_round5_tv_*.sh run their A/B rows on that build).
inline const char* lab_switch(const char* name) {
#ifdef SCIPNP_LAB_SWITCHES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// State handed from one launch of the ADMM-TV iteration to the next (the band kernel's candidates and partial sums; theta / b / x
// of the fused dual update + projection) is stored NON-TEMPORALLY: the consumer runs on all eight XCDs, whose L2s are not coherent
// with each other, so these lines have to reach the memory side before the next launch can read them -- written through as they
// are produced instead of in the end-of-kernel write-back, the iteration at 256x256x8 takes 22.4 us instead of 24.4
// (profiles/r05zd_*; -DSCIPNP_TV_NT=0 builds the plain stores, =2 also loads the hand-off state non-temporally).
#ifndef SCIPNP_TV_NT
#define SCIPNP_TV_NT 1
#endif
template <typename T>
__device__ __forceinline__ void tv_handoff_store(T* p, T v) {
#if SCIPNP_TV_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename T>
__device__ __forceinline__ T tv_handoff_load(const T* p) {
#if SCIPNP_TV_NT >= 2
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// a double moved between lanes by DPP (two 32-bit VALU moves; lanes without a source read 0): CTRL 0x100 + n = row_shl:n (lane l
// reads lane l + n of its row of 16), 0x110 + n = row_shr:n -- the ds_bpermute pairs that __shfl_down(double) compiles to go
// through the LDS pipe (8 waves reducing 8 doubles each measured 4100 clocks in the round-5 band kernel)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum of one double per lane over the wave, the same value in every lane (fixed order: rows of 16 by a shift tree, then the rows)
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0x111>(v);
    v += dpp_f64<0x112>(v);
    v += dpp_f64<0x114>(v);
    v += dpp_f64<0x118>(v);
    const double r0 = readlane_f64(v, 15), r1 = readlane_f64(v, 31), r2 = readlane_f64(v, 47), r3 = readlane_f64(v, 63);
    return (r0 + r1) + (r2 + r3);
}

// skimage's stop test of one channel from the partial sums of its bands (pc[band][it][2] = sum d^2, sum |grad|): executed by
// ONE FULL WAVE (all 64 lanes call it); lane l sums bands l, l+64, ... in order, then a fixed shuffle tree -- the same sums in
// the same order wherever it is evaluated -- and every lane returns the iteration whose `out` skimage keeps.
// float32 array sums (held exactly: rounded once to float) then double arithmetic, as NumPy 1.x does.
__device__ __forceinline__ int tv_band_stop_test(const double* __restrict__ pc, int nbands, int n_iter, size_t MN, double weight,
                                                 double eps) {
    constexpr int MAXIT = 4;                             // n_iter - 1 <= 4 (the banded kernels' halo)
    const int lane = threadIdx.x & 63;
    double s1[MAXIT], s2[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        s1[it] = 0.0;
        s2[it] = 0.0;
        if (it < n_iter - 1)
            for (int k = lane; k < nbands; k += 64) {
                s1[it] += pc[(size_t)k * 2 * n_iter + 2 * it];
                s2[it] += pc[(size_t)k * 2 * n_iter + 2 * it + 1];
            }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        for (int off = 32; off > 0; off >>= 1) {
            s1[it] += __shfl_down(s1[it], off, 64);
            s2[it] += __shfl_down(s2[it], off, 64);
        }
    double E0 = 0.0, Eprev = 0.0;
    int stop_at = n_iter - 1;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if (it >= n_iter - 1 || stop_at != n_iter - 1) continue;
        double E = (double)(float)s1[it];
        E += weight * (double)(float)s2[it];
        E /= (double)MN;
        if (it == 0) { E0 = E; Eprev = E; }
        else if (fabs(Eprev - E) < eps * E0) stop_at = it;
        else Eprev = E;
    }
    return __shfl(stop_at, 0, 64);                       // (lane 0 holds the complete sums)
}

// The same test for EIGHT channels at once when a channel has at most 8 bands: lane = 8 * (channel of the wave) + band; `pc` is
// the lane's channel (nullptr: none).  The shuffle tree inside the 8-lane segments adds exactly the terms the 64-lane tree of
// tv_band_stop_test adds that are not zeros, in the same order -- bit-identical sums; lanes with band 0 return their channel's result.
// In two halves, so that a caller can put its own loads between the operand loads and the arithmetic (round 5:
// pm_dual_project_spec_kernel).  inv_mn != 0: M*N is a power of two and E / (double)MN is the exact scaling E * inv_mn.
constexpr int TV_STOP_MAXIT = 4;
__device__ __forceinline__ void tv_band_stop_load8(const double* __restrict__ pc, int nbands, int n_iter,
                                                   double (&s1)[TV_STOP_MAXIT], double (&s2)[TV_STOP_MAXIT]) {
    const int k = threadIdx.x & 7;
#pragma unroll
    for (int it = 0; it < TV_STOP_MAXIT; ++it) {
        const bool on = pc != nullptr && it < n_iter - 1 && k < nbands;
        s1[it] = on ? pc[(size_t)k * 2 * n_iter + 2 * it] : 0.0;
        s2[it] = on ? pc[(size_t)k * 2 * n_iter + 2 * it + 1] : 0.0;
    }
}
// ... for exactly 5 TV iterations (4 energy pairs per band), from an address that is valid for EVERY lane (`pc_valid`: the lane's
// own band where it has one, any band of the workspace otherwise; `lane_on` says which): four unconditional 16-byte loads, no
// branch -- the predicated form above compiles to a load / wait / select block per iteration (four round trips)
__device__ __forceinline__ void tv_band_stop_load8_full(const double* __restrict__ pc_valid, bool lane_on,
                                                        double (&s1)[TV_STOP_MAXIT], double (&s2)[TV_STOP_MAXIT]) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v[TV_STOP_MAXIT];
#pragma unroll
    for (int it = 0; it < TV_STOP_MAXIT; ++it) v[it] = *(const d2*)(pc_valid + 2 * it);
#pragma unroll
    for (int it = 0; it < TV_STOP_MAXIT; ++it) {
        s1[it] = lane_on ? v[it].x : 0.0;
        s2[it] = lane_on ? v[it].y : 0.0;
    }
}
__device__ __forceinline__ int tv_band_stop_finish8(double (&s1)[TV_STOP_MAXIT], double (&s2)[TV_STOP_MAXIT], int n_iter, size_t MN,
                                                    double weight, double eps, double inv_mn = 0.0) {
    // v[l] += v[l + 4], v[l + 2], v[l + 1] by DPP row shifts: the lanes with band 0 (l = 0, 8 of a row of 16) add exactly the terms
    // of their own 8-lane segment in the order of __shfl_down(., off, 8) -- the other lanes hold nothing that is used
#pragma unroll
    for (int it = 0; it < TV_STOP_MAXIT; ++it) {
        s1[it] += dpp_f64<0x104>(s1[it]);
        s2[it] += dpp_f64<0x104>(s2[it]);
    }
#pragma unroll
    for (int it = 0; it < TV_STOP_MAXIT; ++it) {
        s1[it] += dpp_f64<0x102>(s1[it]);
        s2[it] += dpp_f64<0x102>(s2[it]);
    }
#pragma unroll
    for (int it = 0; it < TV_STOP_MAXIT; ++it) {
        s1[it] += dpp_f64<0x101>(s1[it]);
        s2[it] += dpp_f64<0x101>(s2[it]);
    }
    double E0 = 0.0, Eprev = 0.0;
    int stop_at = n_iter - 1;
#pragma unroll
    for (int it = 0; it < TV_STOP_MAXIT; ++it) {
        if (it >= n_iter - 1 || stop_at != n_iter - 1) continue;
        double E = (double)(float)s1[it];
        E += weight * (double)(float)s2[it];
        if (inv_mn != 0.0) E *= inv_mn;
        else E /= (double)MN;
        if (it == 0) { E0 = E; Eprev = E; }
        else if (fabs(Eprev - E) < eps * E0) stop_at = it;
        else Eprev = E;
    }
    return stop_at;
}
__device__ __forceinline__ int tv_band_stop_test8(const double* __restrict__ pc, int nbands, int n_iter, size_t MN, double weight,
                                                  double eps) {
    double s1[TV_STOP_MAXIT], s2[TV_STOP_MAXIT];
    tv_band_stop_load8(pc, nbands, n_iter, s1, s2);
    return tv_band_stop_finish8(s1, s2, n_iter, MN, weight, eps);
}
// sci_ops.hip: scipnp_pm_dual_update with theta_raw selected per channel (sel != nullptr: theta_raw is the candidate base,
// element i of channel c = i / MN comes from theta_raw[(sel[c] - 1) * img + i]) and scipnp_pm_dual_project doing the stop test
// of the candidate form itself (cd != nullptr; weight / eps of the TV step as doubles)
int pm_dual_update_sel(const float* theta_raw, const int32_t* sel, const float* x, float* theta, float* b, const float* orig,
                       double* sse_part, int which, float sign, int M, int N, int B, int* nblocks, hipStream_t st, int units = 1);
void dual_project_shape(long long Q, int B, int nfill, bool vec_ok, int* VEC, int* CH, unsigned* grid, int units = 1);
int pm_dual_project_sel(const float* theta_raw, const TvCandidates* cd, double tv_weight, double tv_eps, float* x, float* theta,
                        float* b, const float* Phi, const float* y, const float* Phisum, const float* orig, double* sse_part,
                        int nfill, int M, int N, int B, int mode, float c0, float c1, hipStream_t st, int units = 1);
// tv.hip: Chambolle TV + ADMM dual update of planes up to 128 x 128 in one launch (used by iterate.hip)
bool tv_plane_dual_fits(int M, int N, int C, int nfill, bool want_sse);
int tv_plane_dual(const float* x, float* b, float coef, float* theta, int M, int N, int C, float weight, float eps,
                  int n_iter_max, const float* orig, double* sse_part, int which, float sign, int nfill, hipStream_t st);

// block-wide sum of one double per thread (block size a multiple of 64, <= 1024; `tid` is the
// linear thread id); result valid in thread 0.  Fixed tree + fixed wave order: deterministic.
__device__ __forceinline__ double block_sum_double(double v, double* lds /* >= 16 doubles */, int tid, int nthreads) {
    v = wave_sum_f64(v);                  // (DPP row shifts + readlane: 12 ds_bpermute through the LDS pipe before round 5)
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
