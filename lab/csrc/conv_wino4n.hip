// fp32 Winograd F(4x4,3x3) convolution with 16-output-channel workgroups, THREE INDEPENDENT workgroups per CU (round 5,
// LABORATORY: libscipnp_diag.so only) -- the prototype the round-4 review asked for.
//
// conv_wino4.hip: workgroup = 4 waves = 2 tile rows x 2 halves of the transformed patch, 32 output channels: 144 accumulator
// registers per wave (242 VGPRs), 78 KB of LDS -> two workgroups per CU.  Here the workgroup keeps its shape but computes ONE
// 16-channel half: 18 positions x 16 channels = 72 accumulator registers (<= 168 VGPRs: three waves per SIMD), U slabs of 9 KB
// per k-step (double-buffered) and a SINGLE raw-tile buffer (21.5 KB: the tile of group g+1 is requested at the start of k-step
// (g, 0), when the column pass of group g has long finished, and has that k-step to land) = 39 KB of LDS -> THREE workgroups
// per CU, each with its own barriers (unlike the 12-wave workgroup of conv_wino4x.hip, whose waves run in lockstep).
// What it costs: the input transform and the raw tile are now shared by 16 output channels instead of 32 -- twice the packed
// vector operations, patch reads and raw-tile requests per MFMA; U requests per MFMA unchanged.
// Same products, same accumulation order, same association of the output transform's sums as scipnp_conv3x3_c8w4:
// BIT-IDENTICAL results (tests/test_gpu_ops.py).  Weights: scipnp_repack_wino4n re-lays the F(4x4) packing into 16-channel slabs.
#include "wino4_common.hpp"
#ifdef SCIPNP_DIAG_BUILD
#include "../scipnp_lab.h"

namespace scipnp {

constexpr int WN_SLAB = 2 * 9 * 64 * 2;                 // floats per k-step slab of 16 output channels: [xh][v][lane][2] (9216 B)
constexpr int WN_PIECES = WN_SLAB / 256;                // 9 LDS-DMA pieces of 1 KiB
constexpr int WN_DMA_ITERS = (WN_PIECES + 3) / 4;       // U pieces per wave and slab (3; some waves fetch one twice)
constexpr int WN_IN_ITERS = (W4_RAW_PIECES + 3) / 4;    // raw pieces per wave and group (6)
constexpr size_t WN_LDS_BYTES = ((size_t)W4_RAW + 2 * (size_t)WN_SLAB) * sizeof(float);
static_assert(WN_LDS_BYTES >= 4 * 8 * 64 * 16, "the epilogue's exchange buffer lives in the loop's LDS");
static_assert(3 * WN_LDS_BYTES <= 160 * 1024, "three workgroups per CU");
static_assert(WN_DMA_ITERS + WN_IN_ITERS <= 9, "a k-step has nine U vectors to hang its requests on");

// DIAG (timing experiments only, wrong results): bit0 no transform, 1 no raw staging, 2 no U DMA, 3 no barriers, 4 no MFMAs;
// bit6: s_memtime stamps of wave 0 (results unchanged)
template <int DIAG = 0>
__global__ void __launch_bounds__(W4_THREADS, 3)
conv3x3_c8wn_kernel(const Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem_wn[];
    float* const raw_lds = smem_wn;                     // [RAW]
    float* const u_lds = smem_wn + W4_RAW;              // [2][SLAB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long* const stamp_base = (DIAG & 64) ? a.dbg + (size_t)blockIdx.x * 128 : nullptr;
    (void)stamp_base;
    W4_STAMP(0);
    const int tg = wvu >> 1, xh = wvu & 1;
    const int tn = lane & 15, q = lane >> 4;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const float* w_g = nullptr;                         // advanced by NCB*SLAB per k-step (NCB = 16-channel blocks here)
    const float* in_g = nullptr;
    unsigned in_off[WN_IN_ITERS];
    const size_t w_step = (size_t)a.NCB * WN_SLAB;
    const unsigned plane_bytes = (unsigned)(HW * 32);
    (void)plane_bytes;

    auto issue_raw_piece = [&](int k) {
        if (DIAG & 2) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, plane_bytes, 0x00020000);
        int pc = wvu + 4 * k;
        if (pc >= W4_RAW_PIECES) pc -= 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (__attribute__((address_space(3))) void*)((char*)raw_lds + 1024 * pc), 16, in_off[k], 0, 0, 0);
#endif
    };
    auto issue_u_piece = [&](float* dst, int k) {
        if (DIAG & 4) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, WN_SLAB * 4, 0x00020000);
        int pc = wvu + 4 * k;
        if (pc >= WN_PIECES) pc -= 4;
        if (pc >= WN_PIECES) pc -= 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16,
                                                 (unsigned)(1024 * pc + 16 * lane), 0, 0, 0);
#endif
    };
    auto raw_done = [&](bool last) { if (!last) in_g += HW * 8; };
    auto u_done = [&](bool last) { if (!last) w_g += w_step; };

    int split, n, x0, y0;
    {
        unsigned lin = blockIdx.x;
        const unsigned total = a.total_units;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
        unsigned r_split, r_bx, r_by;
        unsigned t = w4_div(lin, (unsigned)a.NCB, a.m_ncb, r_split);
        split = (int)r_split;
        w_g = a.wpk + (size_t)split * WN_SLAB;
#pragma unroll
        for (int k = 0; k < WN_DMA_ITERS; ++k) issue_u_piece(u_lds, k);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        w_g += w_step;
        t = w4_div(t, (unsigned)a.ntx, a.m_ntx, r_bx);
        n = (int)w4_div(t, (unsigned)a.nty, a.m_nty, r_by);
        x0 = (int)r_bx * W4_TW;
        y0 = (int)r_by * W4_TH;
#pragma unroll
        for (int k = 0; k < WN_IN_ITERS; ++k) {
            int pc = wvu + 4 * k;
            if (pc >= W4_RAW_PIECES) pc -= 4;
            const int u = pc * 64 + lane;
            const int hf = u >= W4_UNITS / 2 ? 1 : 0;
            const int v = u - hf * (W4_UNITS / 2);
            static_assert(W4_RSL == 70 && W4_UNITS / 2 < 1259, "the reciprocal 937 / 2^16 is exact for v < 1259 only");
            const int r = (int)(((unsigned)v * 937u) >> 16), sl = v - r * W4_RSL;
            const int g17 = (int)(((unsigned)sl * 241u) >> 12);
            const int c = sl - g17;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            const bool ok = (u < W4_UNITS) & (sl - 17 * g17 != 16) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            const unsigned off = (unsigned)((gy * W + gx) * 32 + 16 * hf);
            in_off[k] = ok ? off : 0xFFFFFF00u;
        }
        in_g = a.in + (size_t)n * a.CGin * HW * 8;
#pragma unroll
        for (int k = 0; k < WN_IN_ITERS; ++k) issue_raw_piece(k);
        raw_done(a.CGin <= 1);
    }

    f32x4 acc[3][6];                                    // [own row xi - 3 xh][nu]
    const int CG = a.CGin;
    const int b_row = (((q >> 1) * W4_THP + 4 * tg + xh) * W4_RSL) * 4 + (q & 1) * 2;   // wave xh reads the patch rows xh .. xh + 4
    const int b_off0 = b_row + (4 * tn + (tn >> 2)) * 4;
    const int b_off1 = b_row + (4 * tn + ((tn + 1) >> 2)) * 4;
    const int a_off = xh * (9 * 128) + lane * 2;

    // the two MFMAs of one U vector -- positions (row, 2np), (row, 2np+1), k-step J -- with two packed vector operations behind
    // each (ops(P, i)) and this vector's LDS-DMA request behind the first
    auto duo = [&](const f32x2 u, const f32x2 (&Vr)[6], f32x4 (&ac)[6], auto NP, auto J, auto P, auto&& ops, auto&& dma) {
        constexpr int np = decltype(NP)::value, j = decltype(J)::value;
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
        if (!(DIAG & 16)) ac[2 * np] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[0], Vr[2 * np][j], ac[2 * np], 0, 0, 0);
        dma(P);
        ops(P, C0{}); ops(P, C1{});
        if (!(DIAG & 16)) ac[2 * np + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[1], Vr[2 * np + 1][j], ac[2 * np + 1], 0, 0, 0);
        ops(P, C2{}); ops(P, C3{});
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
    };

    auto k_loop = [&](auto LO) {
        constexpr bool lo = decltype(LO)::value;
        f32x2 T[3][6], V[3][6];
        f32x2 ta = {0.f, 0.f}, tb = {0.f, 0.f};
        const f32x2 m5 = {-5.f, -5.f};
        if constexpr ((DIAG & 63) != 0) {
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) T[x][nu] = V[x][nu] = f32x2{(float)lane, 1.f};
        }
        auto load_col = [&](int c, f32x2 (&x)[5]) {
            const int bo = (c < 4 ? b_off0 : b_off1) + c * 4;
#pragma unroll
            for (int r = 0; r < 5; ++r) x[r] = *(const f32x2*)(raw_lds + bo + r * (W4_RSL * 4));
        };
        auto col_op = [&](const f32x2 (&x)[5], f32x2& o0, f32x2& o1, f32x2& o2, auto K) {
            if (DIAG & 1) return;
            half_op<lo, decltype(K)::value>(x[0], x[1], x[2], x[3], x[4], o0, o1, o2, ta, tb, m5);
        };
        auto row_op = [&](auto R, auto K) {
            if (DIAG & 1) return;
            constexpr int r = decltype(R)::value, k = decltype(K)::value;
            if constexpr (k < 6) half_op<true, k>(T[r][0], T[r][1], T[r][2], T[r][3], T[r][4], V[r][0], V[r][1], V[r][2], ta, tb, m5);
            else half_op<false, k - 6>(T[r][1], T[r][2], T[r][3], T[r][4], T[r][5], V[r][3], V[r][4], V[r][5], ta, tb, m5);
        };
        auto u_vec = [&](const float* ucur, int pos) { return *(const f32x2*)(ucur + a_off + pos * 128); };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) acc[x][nu] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // U of k-step 0 and the first raw tile
            W4_STAMP(1);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                f32x2 x[5];
                load_col(c, x);
                static_for<6>([&](auto K) { col_op(x, T[0][c], T[1][c], T[2][c], K); });
            }
            W4_STAMP(2);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has read the tile: the buffer is free
        }
        for (int g = 0; g < CG; ++g) {
            // ---- k-step (g, 0), row by row: U of k-step 2g+1 -> buffer 1, raw tile of group g+1 -> the one raw buffer
            const bool more_raw = g + 1 < CG;
            f32x2 af[2];
            af[0] = u_vec(u_lds, 0);
            static_for<12>([&](auto K) { row_op(I0{}, K); });
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            static_for<9>([&](auto P) {
                constexpr int pos = decltype(P)::value, x = pos / 3, np = pos % 3;
                if constexpr (pos + 1 < 9) af[(pos + 1) % 2] = u_vec(u_lds, pos + 1);
                duo(af[pos % 2], V[x], acc[x], std::integral_constant<int, np>{}, I0{}, P, [&](auto Q, auto I) {
                    constexpr int qx = decltype(Q)::value / 3, qn = decltype(Q)::value % 3;
                    if constexpr (qx < 2) row_op(std::integral_constant<int, qx + 1>{}, std::integral_constant<int, 4 * qn + decltype(I)::value>{});
                }, [&](auto Q) {
                    constexpr int qq = decltype(Q)::value;
                    if constexpr (qq < WN_DMA_ITERS) issue_u_piece(u_lds + WN_SLAB, qq);
                    else if constexpr (qq - WN_DMA_ITERS < WN_IN_ITERS) { if (more_raw) issue_raw_piece(qq - WN_DMA_ITERS); }
                });
            });
            u_done(2 * g + 2 >= 2 * CG);
            raw_done(g + 2 >= CG);
            if (g < 24) W4_STAMP(8 + 4 * g);
            if (DIAG & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (g < 24) W4_STAMP(9 + 4 * g);
            // ---- k-step (g, 1), column pair by column pair, with the column pass of group g+1: U of k-step 2g+2 -> buffer 0
            const bool more_u = g + 1 < CG;
            const float* const ub = u_lds + WN_SLAB;
            af[0] = u_vec(ub, 0);
            f32x2 xa[5], xb[5];
            load_col(0, xa);
            load_col(1, xb);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            auto col_ops4 = [&](auto Q, auto I) {
                constexpr int k = 4 * decltype(Q)::value + decltype(I)::value, c = k / 6;
                if constexpr (c % 2 == 0) col_op(xa, T[0][c], T[1][c], T[2][c], std::integral_constant<int, k % 6>{});
                else col_op(xb, T[0][c], T[1][c], T[2][c], std::integral_constant<int, k % 6>{});
                if constexpr (k % 6 == 5 && c + 2 < 6) load_col(c + 2, c % 2 == 0 ? xa : xb);
            };
            static_for<9>([&](auto P) {
                constexpr int pos = decltype(P)::value, np = pos / 3, x = pos % 3;
                if constexpr (pos + 1 < 9) af[(pos + 1) % 2] = u_vec(ub, pos + 1);
                duo(af[pos % 2], V[x], acc[x], std::integral_constant<int, np>{}, I1{}, P, [&](auto Q, auto I) {
                    if constexpr (decltype(Q)::value > 0) col_ops4(std::integral_constant<int, decltype(Q)::value - 1>{}, I);
                }, [&](auto Q) {
                    if constexpr (decltype(Q)::value < WN_DMA_ITERS) { if (more_u) issue_u_piece(u_lds, decltype(Q)::value); }
                });
            });
            u_done(2 * g + 3 >= 2 * CG);
            static_for<4>([&](auto I) { col_ops4(std::integral_constant<int, 8>{}, I); });
            if (g < 24) W4_STAMP(10 + 4 * g);
            if (DIAG & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (g < 24) W4_STAMP(11 + 4 * g);
        }
    };
    if (xh == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    W4_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- output transform, exchange of the partial tiles through LDS, stores (the classic epilogue of conv_wino4.hip for one half)
    float* const xbuf = smem_wn;                        // [wave 4][slot 8][lane 64][4]
    f32x4 keep[2][4];
    {
        f32x4 R[3][4];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const f32x4 m0 = acc[x][0], m1 = acc[x][1], m2 = acc[x][2], m3 = acc[x][3], m4 = acc[x][4], m5 = acc[x][5];
            const f32x4 s1 = m1 + m2, d1 = psub4(m1, m2), s2 = m3 + m4, d2 = psub4(m3, m4);
            R[x][0] = (m0 + s1) + s2;
            R[x][1] = pk_fma(splat<f32x4>(2.f), d2, d1);
            R[x][2] = pk_fma(splat<f32x4>(4.f), s2, s1);
            R[x][3] = pk_fma(splat<f32x4>(8.f), d2, d1) + m5;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 P[4];
            if (xh == 0) {
                const f32x4 s = R[1][j] + R[2][j], d = psub4(R[1][j], R[2][j]);
                P[0] = R[0][j] + s; P[1] = d; P[2] = s; P[3] = d;
            } else {
                const f32x4 s = R[0][j] + R[1][j], d = psub4(R[0][j], R[1][j]);
                P[0] = s; P[1] = d * 2.f; P[2] = s * 4.f; P[3] = pk_fma(splat<f32x4>(8.f), d, R[2][j]);
            }
#pragma unroll
            for (int il = 0; il < 2; ++il) {
                keep[il][j] = xh == 0 ? P[il] : P[2 + il];
                *(f32x4*)(xbuf + ((wvu * 8 + il * 4 + j) * 64 + lane) * 4) = xh == 0 ? P[2 + il] : P[il];
            }
        }
    }
    __syncthreads();
    W4_STAMP(4);
    const float* bias = a.wpk + (size_t)2 * a.CGin * w_step;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
    (void)relu; (void)add_res; (void)mask; (void)bias;
    const int cog0 = split * 2;                             // this lane's c8 group: cog0 + (q >> 1)
    if (cog0 < a.CGout) {
        const bool lane_ok = cog0 + (q >> 1) < a.CGout;
        const f32x4 bs = *(const f32x4*)(bias + (cog0 + (q >> 1)) * 8 + 4 * (q & 1));
        f32x4 v[2][4];
        unsigned off[2][4];
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = y0 + 4 * tg + 2 * xh + il, x = x0 + 4 * tn + j;
                const f32x4 other = *(const f32x4*)(xbuf + (((wvu ^ 1) * 8 + il * 4 + j) * 64 + lane) * 4);
                v[il][j] = (keep[il][j] + other) + bs;
                off[il][j] = (lane_ok && y < H && x < W) ? (unsigned)((y * W + x) * 32 + 16 * (q & 1)) + (unsigned)(q >> 1) * plane_bytes
                                                        : 0x80000000u;
            }
#if defined(__HIP_DEVICE_COMPILE__)
        const size_t half0 = ((size_t)n * a.CGout + cog0) * HW * 8;
        if (add_res) {
            auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[il][j] = v[il][j] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off[il][j], 0, 0));
        }
        if (relu) {
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[il][j][e] = fmaxf(v[il][j][e], 0.f);
        }
        if (mask) {
            auto r_m = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask_src + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 fw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_m, off[il][j], 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[il][j][e] = (fw[e] > 0.f) ? v[il][j][e] : 0.f;
                }
        }
        auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[il][j]), r_out, off[il][j], 0, 0);
#endif
    }
    W4_STAMP(5);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr ((DIAG & 64) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4_STAMP(6);
        if (tid == 0) a.dbg[(size_t)blockIdx.x * 128 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                                                           (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
        asm volatile("s_dcache_wb" ::: "memory");
    }
#endif
}

// the F(4x4) packing of scipnp_pack_conv3x3_wino4 ([k-step][Cout/32][xh][v][lane][4] + bias) re-laid into 16-channel slabs
// [k-step][Cout/16][xh][v][lane][2] + bias: element e of the destination = element 2 (block & 1) + e of the 32-channel vector
__global__ void repack_wino4n_kernel(const float* __restrict__ src, float* __restrict__ dst, int CGin, int NCB32) {
    const size_t total = (size_t)2 * CGin * NCB32 * 2 * WN_SLAB;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int el = (int)(i % WN_SLAB);
        const size_t sl = i / WN_SLAB;
        const int cb16 = (int)(sl % (2 * NCB32)), ks = (int)(sl / (2 * NCB32));
        const int e = el & 1, rest = el >> 1;                                   // rest = (xh * 9 + v) * 64 + lane
        dst[i] = src[((size_t)ks * NCB32 + (cb16 >> 1)) * W4_SLAB + (size_t)rest * 4 + 2 * (cb16 & 1) + e];
    }
    if (i < (size_t)NCB32 * 32) dst[total + i] = src[(size_t)2 * CGin * NCB32 * W4_SLAB + i];
}

static int wn_fill_args(Wino4Args& a, const float* in, const float* packed, float* out, const float* residual, const float* mask_src,
                        int n, int Cin, int Cout, int h, int w, int flags) {
    SCIPNP_REQUIRE(in && packed && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad shape");
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed); SCIPNP_ALIGNED(out);
    SCIPNP_REQUIRE(!(flags & (4 | 8 | 0x200)), "the 16-channel F(4x4,3x3) kernel is stride 1, 8-row workgroups, plain store");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    a.in = in; a.wpk = packed; a.out = out; a.residual = residual; a.mask_src = mask_src; a.dbg = nullptr;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = 2 * ((Cout + 31) / 32);      // 16-channel blocks (of the 32-padded packing)
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W4_TH - 1) / W4_TH;
    a.m_ncb = w4_magic(a.NCB); a.m_ntx = w4_magic(a.ntx); a.m_nty = w4_magic(a.nty);
    a.flags = flags;
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    a.total_units = (unsigned)total;
    return SCIPNP_OK;
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

size_t scipnp_conv3x3_wino4n_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    const int NCB32 = (Cout + 31) / 32;
    return (size_t)2 * (Cin / 8) * NCB32 * 2 * WN_SLAB + (size_t)NCB32 * 32;
}

int scipnp_repack_wino4n(const float* packed_wino4, float* packed_wino4n, int Cin, int Cout, scipnp_stream_t s) {
    SCIPNP_REQUIRE(packed_wino4 && packed_wino4n && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad arguments");
    const int NCB32 = (Cout + 31) / 32;
    const size_t total = (size_t)2 * (Cin / 8) * NCB32 * 2 * WN_SLAB;
    hipLaunchKernelGGL(repack_wino4n_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, packed_wino4,
                       packed_wino4n, Cin / 8, NCB32);
    return launch_status("repack_wino4n_kernel");
}

int scipnp_conv3x3_c8wn(const float* in, const float* packed_wino4n, float* out, const float* residual, const float* mask_src,
                        int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    Wino4Args a;
    if (int rc = wn_fill_args(a, in, packed_wino4n, out, residual, mask_src, n, Cin, Cout, h, w, flags)) return rc;
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_c8wn_kernel<0>, WN_LDS_BYTES, "conv3x3_c8wn")) return rc;
    hipLaunchKernelGGL((conv3x3_c8wn_kernel<0>), dim3(a.total_units), dim3(W4_THREADS), WN_LDS_BYTES, (hipStream_t)s, a);
    return launch_status("conv3x3_c8wn_kernel");
}

int scipnp_conv3x3_c8wn_stamped(const float* in, const float* packed_wino4n, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s) {
    SCIPNP_REQUIRE(stamps, "null pointer");
    Wino4Args a;
    if (int rc = wn_fill_args(a, in, packed_wino4n, out, nullptr, nullptr, n, Cin, Cout, h, w, flags & 1)) return rc;
    a.dbg = stamps;
    // SCIPNP_WN_WGS_PER_CU = 1 / 2: pad the LDS request so that fewer workgroups are resident (what a workgroup's pace is worth alone)
    const char* e = getenv("SCIPNP_WN_WGS_PER_CU");
    const int per = e ? atoi(e) : 3;
    const size_t lds_req = per == 1 ? (size_t)160 * 1024 : per == 2 ? (size_t)80 * 1024 : WN_LDS_BYTES;
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_c8wn_kernel<64>, (size_t)160 * 1024, "conv3x3_c8wn stamped")) return rc;
    hipLaunchKernelGGL((conv3x3_c8wn_kernel<64>), dim3(a.total_units), dim3(W4_THREADS), lds_req, (hipStream_t)s, a);
    return launch_status("conv3x3_c8wn_kernel<stamped>");
}

int scipnp_conv3x3_c8wn_diag(const float* in, const float* packed_wino4n, float* out, int n, int Cin, int Cout, int h, int w,
                             int flags, int diag, scipnp_stream_t s) {
    Wino4Args a;
    if (int rc = wn_fill_args(a, in, packed_wino4n, out, nullptr, nullptr, n, Cin, Cout, h, w, flags & 1)) return rc;
#define WN_DIAG_CASE(D)                                                                                                    \
    case D: {                                                                                                              \
        static LdsAttrOnce attr;                                                                                           \
        if (int rc = attr.ensure((const void*)conv3x3_c8wn_kernel<D>, WN_LDS_BYTES, "conv3x3_c8wn diag")) return rc;      \
        hipLaunchKernelGGL((conv3x3_c8wn_kernel<D>), dim3(a.total_units), dim3(W4_THREADS), WN_LDS_BYTES, (hipStream_t)s, a); \
        break;                                                                                                             \
    }
    switch (diag) {
        WN_DIAG_CASE(1) WN_DIAG_CASE(2) WN_DIAG_CASE(4) WN_DIAG_CASE(8) WN_DIAG_CASE(16) WN_DIAG_CASE(6) WN_DIAG_CASE(7) WN_DIAG_CASE(15)
        default: SCIPNP_REQUIRE(false, "diag mask %d has no instantiation", diag);
    }
#undef WN_DIAG_CASE
    return launch_status("conv3x3_c8wn_kernel<diag>");
}

}  // extern "C"
#endif  /* SCIPNP_DIAG_BUILD */
