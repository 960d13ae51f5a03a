// fp32 Winograd F(4x4,3x3) convolution, THREE waves per SIMD (round 5).
//
// conv_wino4.hip splits the 36 Winograd positions of a tile over two waves (144 accumulator registers each, 242 VGPRs): two
// waves per SIMD, and its stamps (profiles/r04g_wino4_stamps.txt) show what that costs -- a wave alone on its SIMD feeds the
// matrix pipe 57 % of the cycles (every MFMA -> vector operation -> MFMA turn-around is its own to pay), two waves under the
// older-first arbiter 77 %, and a workgroup spends 40 % of its life at its boundaries with the partner in one of those states.
// This kernel keeps everything that sets the traffic and the vector work per MFMA -- 8 x 64 output pixels x 32 output channels
// per workgroup, the same halo tiles, the same k-step slabs IN THE SAME PACKING (scipnp_pack_conv3x3_wino4), the same LDS
// footprint, two workgroups per CU -- and splits the positions over THREE waves by rows of the transformed patch:
//
//     third A: rows xi = 1, 2      B^T rows (0,-4,-4,1,1,0), (0,4,-4,-1,1,0)   share  a = d4 - 4 d2,  b = d3 - 4 d1   (patch rows 1..4)
//     third B: rows xi = 3, 4      B^T rows (0,-2,-1,2,1,0), (0,2,-1,-2,1,0)   share  c = d4 - d2,    e = d3 - d1     (patch rows 1..4)
//     third C: rows xi = 0, 5      B^T rows (4,0,-5,0,1,0),  (0,4,0,-5,0,1)                                          (patch rows 0..5)
//
// four packed operations per patch column for every third (the consecutive split {0,1},{2,3},{4,5} would need 5 / 6 / 5), then
// the full row pass of the two own rows (12 each): 48 packed operations per 48 MFMAs and channel group -- the 1 : 1 of the
// two-wave kernel.  A wave holds 12 positions x 2 output-channel halves = 96 accumulator registers; with the transform's
// registers that is 168: three waves per SIMD, i.e. workgroup = 3 NTG waves = (tile row) x (third).
//
// Every product, every accumulation order and every rounding of the two-wave kernel is kept (same column / row operations, same
// group and k-step order, the output transform's sums associated the same way), so the results are BIT-IDENTICAL to
// scipnp_conv3x3_c8w4 -- tests/test_gpu_ops.py compares them with torch.equal.  LABORATORY code: built into libscipnp_diag.so only.
//
// Output transform: rows xi own to a wave give R[xi][j] (the row pass of the output transform), and the column pass needs only
//     sA = R1 + R2, dA = R1 - R2      sB = R3 + R4, dB = R3 - R4      R0, R5
//     Y0 = (R0 + sA) + sB    Y1 = dA + 2 dB    Y2 = 4 sB + sA    Y3 = fma(8, dB, R5) + dA
// thirds A and B leave their (s, d) in LDS; A finishes output row 1, B row 2, C rows 0 and 3.
#include "wino4_common.hpp"
#ifdef SCIPNP_DIAG_BUILD
#include "../scipnp_lab.h"
#endif

namespace scipnp {

#ifndef W6_WPE
#define W6_WPE 3
#endif
#ifndef W6_NTG
#define W6_NTG 4                                         // tile rows per workgroup: 4 -> 12 waves, ONE workgroup per CU (2 -> 6 waves)
#endif
// Six-wave workgroups do NOT give three waves per SIMD: the dispatcher reserves ceil(6 / 4) = 2 wave slots on every SIMD for a
// workgroup, so a second one does not fit into the 3 slots 168 registers leave (measured: one resident workgroup per CU,
// 336 us against 253 us; profiles/r05a_wino6_*).  Twelve waves = 4 tile rows x 3 thirds fill the slots exactly -- one workgroup
// per CU, 16 x 64 output pixels x 32 channels, and the U slab of a k-step now serves 64 tiles instead of 32.
constexpr int W6_WAVES = 3 * W6_NTG;
constexpr int W6_THREADS = 64 * W6_WAVES;
constexpr int W6_TH = 4 * W6_NTG, W6_THP = W6_TH + 2;                     // output rows per workgroup, halo rows
constexpr int W6_UNITS = W6_THP * W4_RSL * 2;                             // 16-byte units of the halo tile: [hf][row][slot]
constexpr int W6_RAW_PIECES = (W6_UNITS + 63) / 64;
constexpr int W6_RAW = W6_RAW_PIECES * 256;                               // floats per raw buffer
constexpr int W6_IN_ITERS = (W6_RAW_PIECES + W6_WAVES - 1) / W6_WAVES;    // raw pieces per wave and group (some waves fetch one twice)
constexpr int W6_DMA_ITERS = (W4_PIECES + W6_WAVES - 1) / W6_WAVES;       // U pieces per wave and slab (likewise)
constexpr size_t W6_LOOP_BYTES = (2 * (size_t)W6_RAW + 2 * (size_t)W4_SLAB) * sizeof(float);
constexpr size_t W6_XBUF_BYTES = (size_t)W6_NTG * 2 * 16 * 64 * 16;       // the epilogue's exchange buffer, in the loop's LDS
constexpr size_t W6_LDS_BYTES = W6_LOOP_BYTES > W6_XBUF_BYTES ? W6_LOOP_BYTES : W6_XBUF_BYTES;
static_assert(W6_LDS_BYTES <= 160 * 1024, "LDS of a CU");
static_assert(W6_DMA_ITERS + W6_IN_ITERS <= 7, "request placement of k-step 1");

// position rows of a third, and where the packer (pack_wino4_kernel: [xh][vector v][lane][4], v = 3 row + np for k-step 0,
// 3 np + row for k-step 1) put the vector (xi, np) of k-step j
__host__ __device__ constexpr int w6_row(int xt, int r) { return xt == 0 ? 1 + r : xt == 1 ? 3 + r : 5 * r; }
__host__ __device__ constexpr int w6_uvec(int xt, int r, int np, int j) {
    const int xi = w6_row(xt, r), xh = xi / 3, own = xi % 3;
    return (xh * 9 + (j ? 3 * np + own : 3 * own + np)) * 256;            // floats
}

// DIAG (timing experiments only, wrong results): bit0 no transform, 1 no raw staging, 2 no U DMA, 3 no barriers, 4 no MFMAs,
// 5 no epilogue; bit6: s_memtime stamps of wave 0 (results unchanged)
template <int TAG, int DIAG = 0>
__global__ void __launch_bounds__(W6_THREADS, W6_WPE)
conv3x3_c8w6_kernel(const Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem_w6[];
    float* const raw_lds = smem_w6;                    // [2][RAW]
    float* const u_lds = smem_w6 + 2 * W6_RAW;         // [2][SLAB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long* const stamp_base = (DIAG & 64) ? a.dbg + (size_t)blockIdx.x * 128 : nullptr;
    (void)stamp_base;
    W4_STAMP(0);
    const int tg = (int)(((unsigned)wvu * 11u) >> 5), xt = wvu - 3 * tg;  // tile row of the workgroup (wvu / 3), third of the transformed rows
    const int tn = lane & 15, q = lane >> 4;             // tile along x, channel pair
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    const float* w_g = nullptr;                          // advanced by NCB*SLAB per k-step
    const float* in_g = nullptr;                         // advanced by HW*8 per group
    unsigned in_off[W6_IN_ITERS];
    const size_t w_step = (size_t)a.NCB * W4_SLAB;
    const unsigned plane_bytes = (unsigned)(HW * 32);
    (void)plane_bytes;

    auto issue_raw_piece = [&](float* dst, int k) {
        if (DIAG & 2) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, plane_bytes, 0x00020000);
        int pc = wvu + W6_WAVES * k;
        if (pc >= W6_RAW_PIECES) pc -= W6_WAVES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16, in_off[k], 0, 0, 0);
#endif
    };
    auto issue_u_piece = [&](float* dst, int k) {
        if (DIAG & 4) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, W4_SLAB * 4, 0x00020000);
        int pc = wvu + W6_WAVES * k;
        if (pc >= W4_PIECES) pc -= W6_WAVES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16,
                                                 (unsigned)(1024 * pc + 16 * lane), 0, 0, 0);
#endif
    };
    // `last`: no further group / k-step exists -- the pointer stays and the same data is fetched again (unused), so that every
    // step issues the same number of memory operations and the vmcnt waits below are constants
    auto raw_done = [&](bool last) { if (!last) in_g += HW * 8; };
    auto u_done = [&](bool last) { if (!last) w_g += w_step; };

    // ---- the unit: XCD-aware order (as conv_wino4.hip), U of k-step 0 requested before anything else is known
    int split, n, x0, y0;
    {
        unsigned lin = blockIdx.x;
        const unsigned total = a.total_units;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
        unsigned r_split, r_bx, r_by;
        unsigned t = w4_div(lin, (unsigned)a.NCB, a.m_ncb, r_split);
        split = (int)r_split;
        w_g = a.wpk + (size_t)split * W4_SLAB;
#pragma unroll
        for (int k = 0; k < W6_DMA_ITERS; ++k) issue_u_piece(u_lds, k);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        w_g += w_step;
        t = w4_div(t, (unsigned)a.ntx, a.m_ntx, r_bx);
        n = (int)w4_div(t, (unsigned)a.nty, a.m_nty, r_by);
        x0 = (int)r_bx * W4_TW;
        y0 = (int)r_by * W6_TH;
#pragma unroll
        for (int k = 0; k < W6_IN_ITERS; ++k) {
            int pc = wvu + W6_WAVES * k;
            if (pc >= W6_RAW_PIECES) pc -= W6_WAVES;
            const int u = pc * 64 + lane;
            const int hf = u >= W6_UNITS / 2 ? 1 : 0;
            const int v = u - hf * (W6_UNITS / 2);
            static_assert(W4_RSL == 70 && W6_UNITS / 2 + 64 < 43000, "the reciprocal 3745 / 2^18 is exact for v < 43690");
            const int r = (int)(((unsigned)v * 3745u) >> 18), sl = v - r * W4_RSL;
            const int g17 = (int)(((unsigned)sl * 241u) >> 12);
            const int c = sl - g17;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            const bool ok = (u < W6_UNITS) & (sl - 17 * g17 != 16) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            const unsigned off = (unsigned)((gy * W + gx) * 32 + 16 * hf);
            in_off[k] = ok ? off : 0xFFFFFF00u;
        }
        in_g = a.in + (size_t)n * a.CGin * HW * 8;
#pragma unroll
        for (int k = 0; k < W6_IN_ITERS; ++k) issue_raw_piece(raw_lds, k);
        raw_done(a.CGin <= 1);
#pragma unroll
        for (int k = 0; k < W6_IN_ITERS; ++k) issue_raw_piece(raw_lds + W6_RAW, k);
        raw_done(a.CGin <= 2);
    }

    f32x4 acc[2][6][2];                                 // [own row][nu][co half]
    const int CG = a.CGin;

    // per-lane LDS offsets (floats): patch of tile (tg, tn), channel pair q (half-pixel plane q >> 1, 8 bytes (q & 1) of the unit)
    const int b_plane = ((q >> 1) * W6_THP + 4 * tg) * W4_RSL * 4 + (q & 1) * 2;
    const int b_off0 = b_plane + (4 * tn + (tn >> 2)) * 4;             // columns 0..3
    const int b_off1 = b_plane + (4 * tn + ((tn + 1) >> 2)) * 4;       // columns 4, 5
    const int a_off = lane * 4;

    // the four MFMAs of one U vector -- positions (row, 2np), (row, 2np+1) x the two output-channel halves, k-step J -- with one
    // packed vector operation behind each (ops(P, i)) and this vector's share of the k-step's LDS-DMA requests behind the first
    auto quad = [&](const f32x4 u, const f32x2 (&Vr)[6], f32x4 (&ac)[6][2], auto NP, auto J, auto P, auto&& ops, auto&& dma) {
        constexpr int np = decltype(NP)::value, j = decltype(J)::value;
        const float b0 = Vr[2 * np][j], b1 = Vr[2 * np + 1][j];
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
        if (!(DIAG & 16)) ac[2 * np][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[0], b0, ac[2 * np][0], 0, 0, 0);
        dma(P);
        ops(P, C0{});
        if (!(DIAG & 16)) ac[2 * np + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[1], b1, ac[2 * np + 1][0], 0, 0, 0);
        ops(P, C1{});
        if (!(DIAG & 16)) ac[2 * np][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[2], b0, ac[2 * np][1], 0, 0, 0);
        ops(P, C2{});
        if (!(DIAG & 16)) ac[2 * np + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[3], b1, ac[2 * np + 1][1], 0, 0, 0);
        ops(P, C3{});
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
    };

    auto k_loop = [&](auto XT) {
        constexpr int XTc = decltype(XT)::value;
        constexpr int NR = XTc == 2 ? 6 : 4;                     // patch rows a column pass reads
        constexpr int R0 = XTc == 2 ? 0 : 1;                     // the first of them
        f32x2 T[2][6], V[2][6];
        f32x2 ta = {0.f, 0.f}, tb = {0.f, 0.f};
        const f32x2 m5 = {-5.f, -5.f};
        if constexpr ((DIAG & 63) != 0) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) T[x][nu] = V[x][nu] = f32x2{(float)lane, 1.f};
        }
        auto load_col = [&](const float* rawp, int c, f32x2 (&x)[NR]) {
            const int bo = (c < 4 ? b_off0 : b_off1) + c * 4 + R0 * (W4_RSL * 4);
#pragma unroll
            for (int r = 0; r < NR; ++r) x[r] = *(const f32x2*)(rawp + bo + r * (W4_RSL * 4));
        };
        // operation K (0..3) of the column pass of one patch column -> own two rows of B^T d (the same operations, in the same
        // association, as half_op of wino4_common.hpp forms for these rows)
        auto col_op = [&](const f32x2 (&x)[NR], f32x2& o0, f32x2& o1, auto K) {
            if (DIAG & 1) return;
            constexpr int k = decltype(K)::value;
            if constexpr (XTc == 0) {             // x[] = d1..d4: (d4 - 4 d2) +- (d3 - 4 d1)
                if constexpr (k == 0) ta = fma_m4(x[1], x[3]);
                if constexpr (k == 1) tb = fma_m4(x[0], x[2]);
                if constexpr (k == 2) o0 = padd(ta, tb);
                if constexpr (k == 3) o1 = psub4(ta, tb);
            } else if constexpr (XTc == 1) {      // (d4 - d2) +- 2 (d3 - d1)
                if constexpr (k == 0) ta = psub4(x[3], x[1]);
                if constexpr (k == 1) tb = psub4(x[2], x[0]);
                if constexpr (k == 2) o0 = fma_p2(tb, ta);
                if constexpr (k == 3) o1 = fma_m2(tb, ta);
            } else {                              // x[] = d0..d5: 4 d0 - 5 d2 + d4 | 4 d1 - 5 d3 + d5
                if constexpr (k == 0) ta = fma_k(x[2], x[4], m5);
                if constexpr (k == 1) tb = fma_k(x[3], x[5], m5);
                if constexpr (k == 2) o0 = fma_p4(x[0], ta);
                if constexpr (k == 3) o1 = fma_p4(x[1], tb);
            }
        };
        // operation K (0..11) of the row pass of own row r: T[r][0..5] -> V[r][0..5]
        auto row_op = [&](auto R, auto K) {
            if (DIAG & 1) return;
            constexpr int r = decltype(R)::value, k = decltype(K)::value;
            if constexpr (k < 6) half_op<true, k>(T[r][0], T[r][1], T[r][2], T[r][3], T[r][4], V[r][0], V[r][1], V[r][2], ta, tb, m5);
            else half_op<false, k - 6>(T[r][1], T[r][2], T[r][3], T[r][4], T[r][5], V[r][3], V[r][4], V[r][5], ta, tb, m5);
        };
        auto u_vec = [&](const float* ucur, auto R, auto NP, auto J) {
            return *(const f32x4*)(ucur + a_off + w6_uvec(XTc, decltype(R)::value, decltype(NP)::value, decltype(J)::value));
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

        {   // head of the unit: U of k-step 0 and the raw tiles of groups 0 and 1 are requested; column pass of group 0
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int nu = 0; nu < 6; ++nu)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[x][nu][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(W6_IN_ITERS) : "memory");     // all but the second tile's requests
            W4_STAMP(1);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                f32x2 x[NR];
                load_col(raw_lds, c, x);
                static_for<4>([&](auto K) { col_op(x, T[0][c], T[1][c], K); });
            }
            W4_STAMP(2);
        }
        for (int g = 0; g < CG; ++g) {
            float* const rcur = raw_lds + (g & 1) * W6_RAW;          // group g (its column pass is done): free behind the next barrier
            float* const rnext = raw_lds + ((g & 1) ^ 1) * W6_RAW;   // group g+1, requested one k-step ago
            // ---- k-step (g, 0), row by row; U of k-step 2g+1 -> its buffer
            f32x4 af[2];                                             // the current U vector and the next one
            af[0] = u_vec(u_lds, I0{}, I0{}, I0{});
            static_for<12>([&](auto K) { row_op(I0{}, K); });
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            static_for<6>([&](auto P) {
                constexpr int pos = decltype(P)::value, x = pos / 3, np = pos % 3;
                if constexpr (pos + 1 < 6)
                    af[(pos + 1) % 2] = u_vec(u_lds, std::integral_constant<int, (pos + 1) / 3>{}, std::integral_constant<int, (pos + 1) % 3>{}, I0{});
                quad(af[pos % 2], V[x], acc[x], std::integral_constant<int, np>{}, I0{}, P, [&](auto Q, auto I) {
                    // vectors 0..2 (the MFMAs of row 0) carry the twelve operations of the row pass of row 1
                    if constexpr (decltype(Q)::value < 3) row_op(I1{}, std::integral_constant<int, 4 * decltype(Q)::value + decltype(I)::value>{});
                }, [&](auto Q) {
                    if constexpr (decltype(Q)::value < W6_DMA_ITERS) issue_u_piece(u_lds + W4_SLAB, decltype(Q)::value);
                });
            });
            u_done(2 * g + 2 >= 2 * CG);
            if (g < 24) W4_STAMP(8 + 4 * g);
            if (DIAG & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (g < 24) W4_STAMP(9 + 4 * g);
            // ---- k-step (g, 1), column pair by column pair, with the column pass of group g+1: U of k-step 2g+2 -> the buffer of
            // k-step 2g, raw tile of group g+2 -> the buffer of group g
            const bool more_u = g + 1 < CG, more_raw = g + 2 < CG;
            const float* const ub = u_lds + W4_SLAB;
            af[0] = u_vec(ub, I0{}, I0{}, I1{});
            f32x2 xc[NR];                                            // own patch rows of the current column
            load_col(rnext, 0, xc);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            // 24 column-pass operations (column c: 4c .. 4c+3), four under each U vector from the second on, the last four behind the
            // last vector.  One register set for the patch column: operations 0, 1 consume the rows 1..4 (thirds A, B) / 2..5 (third C),
            // so the next column's values are requested right behind them (C: its rows 0, 1 behind operation 3) and have two MFMAs
            // and the next vector's first to arrive.  The results overwrite the V entries the k-step has finished with: T[.][c] is
            // written behind vector c + 1, row 0's entry dead since vector c or earlier, row 1's read last by that vector's MFMAs.
            auto load_rows = [&](int c, auto LO, auto HI) {
                const int bo = (c < 4 ? b_off0 : b_off1) + c * 4 + R0 * (W4_RSL * 4);
#pragma unroll
                for (int r = decltype(LO)::value; r < decltype(HI)::value; ++r) xc[r] = *(const f32x2*)(rnext + bo + r * (W4_RSL * 4));
            };
            auto col_ops4 = [&](auto Q, auto I) {
                constexpr int c = decltype(Q)::value, k = decltype(I)::value;
                col_op(xc, T[0][c], T[1][c], I);
                if constexpr (c + 1 < 6) {
                    if constexpr (XTc != 2) { if constexpr (k == 1) load_rows(c + 1, I0{}, std::integral_constant<int, NR>{}); }
                    else {
                        if constexpr (k == 1) load_rows(c + 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{});
                        if constexpr (k == 3) load_rows(c + 1, I0{}, std::integral_constant<int, 2>{});
                    }
                }
            };
            static_for<6>([&](auto P) {
                constexpr int pos = decltype(P)::value, np = pos / 2, x = pos % 2;
                if constexpr (pos + 1 < 6)
                    af[(pos + 1) % 2] = u_vec(ub, std::integral_constant<int, (pos + 1) % 2>{}, std::integral_constant<int, (pos + 1) / 2>{}, I1{});
                quad(af[pos % 2], V[x], acc[x], std::integral_constant<int, np>{}, I1{}, P, [&](auto Q, auto I) {
                    if constexpr (decltype(Q)::value > 0) col_ops4(std::integral_constant<int, decltype(Q)::value - 1>{}, I);
                }, [&](auto Q) {
                    // the U pieces first, then the raw pieces (the wait at the end of the k-step counts on that order): request i of
                    // the DMA_ITERS + IN_ITERS behind vector i; with seven of them the first vector carries two
                    constexpr int qq = decltype(Q)::value, NREQ = W6_DMA_ITERS + W6_IN_ITERS;
                    constexpr int first = (NREQ == 7 && qq > 0) ? qq + 1 : qq, cnt = (NREQ == 7 && qq == 0) ? 2 : 1;
                    static_for<cnt>([&](auto E) {
                        constexpr int i = first + decltype(E)::value;
                        if constexpr (i < W6_DMA_ITERS) { if (more_u) issue_u_piece(u_lds, i); }
                        else if constexpr (i < NREQ) { if (more_raw) issue_raw_piece(rcur, i - W6_DMA_ITERS); }
                    });
                });
            });
            u_done(2 * g + 3 >= 2 * CG);
            raw_done(g + 3 >= CG);
            if (!more_raw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            static_for<4>([&](auto I) { col_ops4(std::integral_constant<int, 5>{}, I); });
            if (g < 24) W4_STAMP(10 + 4 * g);
            if (DIAG & 8) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W6_IN_ITERS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W6_IN_ITERS) : "memory");   // U of k-step 2g+2
            if (g < 24) W4_STAMP(11 + 4 * g);
        }
    };
    if (xt == 0) k_loop(std::integral_constant<int, 0>{});
    else if (xt == 1) k_loop(std::integral_constant<int, 1>{});
    else k_loop(std::integral_constant<int, 2>{});
    W4_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the re-fetched last slab must not land in the exchange buffer)
    __syncthreads();
    if (DIAG & 32) return;

    // ---- output transform.  Own rows: R[r][j] = sum_nu M[r][nu] A^T[j][nu]; thirds A, B: s = R[0] + R[1], d = R[0] - R[1] -> LDS
    float* const xbuf = smem_w6;                        // [tg 2][third A, B][slot 16 = 8 which + 2 j + h][lane 64][4]
    f32x4 p[2][4][2];                                   // [s | d, or R0 | R5][j][h]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 R[2][4];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const f32x4 m0 = acc[x][0][h], m1 = acc[x][1][h], m2 = acc[x][2][h], m3 = acc[x][3][h], m4 = acc[x][4][h], m5 = acc[x][5][h];
            const f32x4 s1 = m1 + m2, d1 = psub4(m1, m2), s2 = m3 + m4, d2 = psub4(m3, m4);
            R[x][0] = (m0 + s1) + s2;
            R[x][1] = pk_fma(splat<f32x4>(2.f), d2, d1);
            R[x][2] = pk_fma(splat<f32x4>(4.f), s2, s1);
            R[x][3] = pk_fma(splat<f32x4>(8.f), d2, d1) + m5;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (xt == 2) { p[0][j][h] = R[0][j]; p[1][j][h] = R[1][j]; }
            else { p[0][j][h] = R[0][j] + R[1][j]; p[1][j][h] = psub4(R[0][j], R[1][j]); }
        }
    }
    if (xt != 2) {
#pragma unroll
        for (int wh = 0; wh < 2; ++wh)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *(f32x4*)(xbuf + ((((tg * 2 + xt) * 16) + wh * 8 + j * 2 + h) * 64 + lane) * 4) = p[wh][j][h];
    }
    __syncthreads();
    W4_STAMP(4);
    auto other = [&](int third, int wh, int j, int h) {
        return *(const f32x4*)(xbuf + ((((tg * 2 + third) * 16) + wh * 8 + j * 2 + h) * 64 + lane) * 4);
    };
    const float* bias = a.wpk + (size_t)2 * a.CGin * w_step;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
    // one output row of the tile: v[j][h] = the row's values before bias; lane: tile (tg, tn), channels 32 split + 16 h + 4 q + r
    auto store_row = [&](int i, f32x4 (&v)[4][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cog0 = split * 4 + h * 2;                    // this lane's group: cog0 + (q >> 1)
            if (cog0 >= a.CGout) continue;                         // wave-uniform
            const bool lane_ok = cog0 + (q >> 1) < a.CGout;
            const f32x4 bs = *(const f32x4*)(bias + (cog0 + (q >> 1)) * 8 + 4 * (q & 1));
            unsigned off[4];
            const int y = y0 + 4 * tg + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + 4 * tn + j;
                v[j][h] = v[j][h] + bs;
                off[j] = (lane_ok && y < H && x < W) ? (unsigned)((y * W + x) * 32 + 16 * (q & 1)) + (unsigned)(q >> 1) * plane_bytes
                                                    : 0x80000000u;
            }
#if defined(__HIP_DEVICE_COMPILE__)
            const size_t half0 = ((size_t)n * a.CGout + cog0) * HW * 8;
            if (add_res) {
                auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j][h] = v[j][h] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off[j], 0, 0));
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][h][e] = fmaxf(v[j][h][e], 0.f);
            }
            if (mask) {
                auto r_m = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask_src + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 fw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_m, off[j], 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][h][e] = (fw[e] > 0.f) ? v[j][h][e] : 0.f;
                }
            }
            auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j][h]), r_out, off[j], 0, 0);
#endif
        }
    };
    if (xt == 0) {                                      // Y1 = dA + 2 dB
        f32x4 v[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[j][h] = p[1][j][h] + other(1, 1, j, h) * 2.f;
        store_row(1, v);
    } else if (xt == 1) {                               // Y2 = 4 sB + sA
        f32x4 v[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[j][h] = p[0][j][h] * 4.f + other(0, 0, j, h);
        store_row(2, v);
    } else {                                            // Y0 = (R0 + sA) + sB, Y3 = fma(8, dB, R5) + dA
        f32x4 v[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[j][h] = (p[0][j][h] + other(0, 0, j, h)) + other(1, 0, j, h);
        store_row(0, v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[j][h] = pk_fma(splat<f32x4>(8.f), other(1, 1, j, h), p[1][j][h]) + other(0, 1, j, h);
        store_row(3, v);
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

static int w6_fill_args(Wino4Args& a, const float* in, const float* packed_wino4, float* out, const float* residual,
                        const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags) {
    SCIPNP_REQUIRE(in && packed_wino4 && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0,
                   "bad shape n=%d Cin=%d Cout=%d h=%d w=%d (channels must be multiples of 8)", n, Cin, Cout, h, w);
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed_wino4); SCIPNP_ALIGNED(out);
    if (residual) SCIPNP_ALIGNED(residual);
    if (mask_src) SCIPNP_ALIGNED(mask_src);
    SCIPNP_REQUIRE(!(flags & (4 | 8 | 0x200)), "the six-wave F(4x4,3x3) kernel is stride 1, 8-row workgroups, plain store");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    a.in = in; a.wpk = packed_wino4; a.out = out; a.residual = residual; a.mask_src = mask_src; a.dbg = nullptr;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = (Cout + 31) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W6_TH - 1) / W6_TH;
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

#ifdef SCIPNP_DIAG_BUILD   /* ---- laboratory entries only (libscipnp_diag.so, include/scipnp_diag.h): measured slower than the product's conv_wino4.hip */

int scipnp_conv3x3_c8w6(const float* in, const float* packed_wino4, float* out, const float* residual, const float* mask_src,
                        int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    Wino4Args a;
    if (int rc = w6_fill_args(a, in, packed_wino4, out, residual, mask_src, n, Cin, Cout, h, w, flags)) return rc;
    const int tag = (flags & 0x100) ? 1 : 0;
    const void* fns[2] = {(const void*)conv3x3_c8w6_kernel<0>, (const void*)conv3x3_c8w6_kernel<1>};
    static LdsAttrOnce attr[2];
    if (int rc = attr[tag].ensure(fns[tag], W6_LDS_BYTES, "conv3x3_c8w6")) return rc;
    const dim3 grid(a.total_units), block(W6_THREADS);
    if (tag) hipLaunchKernelGGL((conv3x3_c8w6_kernel<1>), grid, block, W6_LDS_BYTES, (hipStream_t)s, a);
    else hipLaunchKernelGGL((conv3x3_c8w6_kernel<0>), grid, block, W6_LDS_BYTES, (hipStream_t)s, a);
    return launch_status("conv3x3_c8w6_kernel");
}


/* stamped instantiation: the slots of scipnp_conv3x3_c8w4_stamped (tools/probes/wino4_stamps.py reads both) */
int scipnp_conv3x3_c8w6_stamped(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s) {
    SCIPNP_REQUIRE(stamps, "null pointer");
    Wino4Args a;
    if (int rc = w6_fill_args(a, in, packed_wino4, out, nullptr, nullptr, n, Cin, Cout, h, w, flags & 1)) return rc;
    a.dbg = stamps;
    const dim3 grid(a.total_units), block(W6_THREADS);
    const size_t lds_req = getenv("SCIPNP_W4_ONE_PER_CU") ? (size_t)160 * 1024 : W6_LDS_BYTES;
#define W6_STAMP_CASE(D)                                                                                                     \
    case D: {                                                                                                                \
        static LdsAttrOnce attr;                                                                                             \
        if (int rc = attr.ensure((const void*)conv3x3_c8w6_kernel<0, 64 | D>, (size_t)160 * 1024, "conv3x3_c8w6 stamped")) return rc; \
        hipLaunchKernelGGL((conv3x3_c8w6_kernel<0, 64 | D>), grid, block, lds_req, (hipStream_t)s, a);                       \
        break;                                                                                                               \
    }
    switch ((flags >> 12) & 7) {
        W6_STAMP_CASE(0) W6_STAMP_CASE(1) W6_STAMP_CASE(6) W6_STAMP_CASE(7)
        default: SCIPNP_REQUIRE(false, "no stamped build for that mask");
    }
#undef W6_STAMP_CASE
    return launch_status("conv3x3_c8w6_kernel<stamped>");
}

/* the same kernel with parts switched off (timing only, WRONG results): diag bits as scipnp_conv3x3_c8w4_diag */
int scipnp_conv3x3_c8w6_diag(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                             int flags, int diag, scipnp_stream_t s) {
    Wino4Args a;
    if (int rc = w6_fill_args(a, in, packed_wino4, out, nullptr, nullptr, n, Cin, Cout, h, w, flags & 1)) return rc;
    const dim3 grid(a.total_units), block(W6_THREADS);
#define W6_DIAG_CASE(D)                                                                                                    \
    case D: {                                                                                                              \
        static LdsAttrOnce attr;                                                                                           \
        if (int rc = attr.ensure((const void*)conv3x3_c8w6_kernel<0, D>, W6_LDS_BYTES, "conv3x3_c8w6 diag")) return rc;   \
        hipLaunchKernelGGL((conv3x3_c8w6_kernel<0, D>), grid, block, W6_LDS_BYTES, (hipStream_t)s, a);                     \
        break;                                                                                                             \
    }
    switch (diag) {
        W6_DIAG_CASE(1) W6_DIAG_CASE(2) W6_DIAG_CASE(4) W6_DIAG_CASE(8) W6_DIAG_CASE(16) W6_DIAG_CASE(6) W6_DIAG_CASE(7)
        W6_DIAG_CASE(15) W6_DIAG_CASE(48) W6_DIAG_CASE(49)
        default: SCIPNP_REQUIRE(false, "diag mask %d has no instantiation", diag);
    }
#undef W6_DIAG_CASE
    return launch_status("conv3x3_c8w6_kernel<diag>");
}

#endif  /* SCIPNP_DIAG_BUILD */

}  // extern "C"
