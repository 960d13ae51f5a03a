// 3x3 / stride-1 / pad-1 convolution in fp32 arithmetic as Winograd F(4x4, 3x3) on the CDNA4 matrix cores.
//
//   Y(4x4) = A^T [ (G g G^T) (.) (B^T d B) ] A          per 4x4 output tile, 6x6 input patch d, 3x3 filter g
//
// 36 products per 16 outputs and channel pair: 2.25 multiply-adds per output against 4 for F(2x2,3x3) (conv_wino.hip) and 9
// for the direct form (conv.hip) -- 1.78x fewer v_mfma_f32_16x16x4_f32 than the F(2x2) kernel, every product still an exact
// fp32 product accumulated in fp32.  Interpolation points 0, +-1, +-2, inf (the matrices of Lavin & Gray); the transforms are
// small-integer combinations, the filter transform U = G g G^T is done once per weight update, in double
// (scipnp_pack_conv3x3_wino4).  Through the 12 layers of FFDNet the result differs from float64 by 3e-7 rel-L2 (F(2x2): 1.9e-7,
// direct fp32: 1.1e-7; tests/test_wino4_numerics.py restates the algorithm in NumPy).
//
// GEMM view, per Winograd position p = (xi, nu) of 36:   M_p[co][tile] = sum_ci U_p[co][ci] * V_p[ci][tile]
//   A = U_p   16 (co) x 4 (ci)     lane l holds A[l & 15][l >> 4]
//   B = V_p    4 (ci) x 16 (tile)  lane l holds B[l >> 4][l & 15]
//   D         16 (co) x 16 (tile)  lane l holds D[4*(l >> 4) + r][l & 15], r = 0..3
// As in conv_wino.hip a lane owns ONE tile and, per 8-channel group, the channel pair {2q, 2q+1}, q = l >> 4, and transforms
// its own patch in registers with packed fp32 math.  All 36 positions of 16 tiles x 32 output channels would be 288
// accumulator registers, so the positions are split over TWO waves: wave xh owns the rows xi = 3 xh .. 3 xh + 2 of the
// transformed patch (18 positions x 2 output-channel halves = 36 accumulators, 144 VGPRs) -- it needs only its three rows of the
// column pass B^T d (6 instead of 12 operations per column, nothing computed twice) and the full row pass of those rows.  Both
// waves form partial output tiles  sum_{xi in own rows} A^T[:, xi] (M[xi, :] A)  and exchange half of them through LDS once,
// after the K loop (wave 0 finishes output rows 0, 1 of the tile, wave 1 rows 2, 3).
//
// Workgroup = 4 waves = (tile row tg in 0..1) x (xh in 0..1): 8 rows x 64 columns of output pixels x 32 output channels; two
// workgroups per CU.  K loop over input channel groups of 8 = two k-steps j (k-step j multiplies the channels 2q + j).  LDS:
//   raw halo tile (10 x 66 pixels x 8 channels), two buffers, filled by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
//     registers, pixels outside the image fetched past the descriptor's range = zeros) one group ahead: units of 16 bytes =
//     4 channels of a pixel, laid out [hf][row][slot], pixel x in slot x + (x >> 4) (one padding slot per 16 pixels): a DMA
//     instruction fetches 64 consecutive pixels' halves (16 cache lines), and the 8-byte reads of a patch column (pixel 4tn + c
//     of the 16 tiles of a wave, channel pair q) fall on 16 different bank groups (one 2-way conflict in columns 4, 5);
//   U slabs per K-STEP (36 positions x 32 co x 4 ci = 18 KiB, laid out [xh][vector v][lane][4] by the packer: one 16-byte read
//     feeds four MFMAs: positions (xi, 2np), (xi, 2np+1) x the two output-channel halves), two buffers, filled by LDS-DMA one
//     k-step ahead.
// One barrier per k-step (36 MFMAs per wave): a whole group's slab pair double-buffered (74 KiB) plus the tiles would not leave
// room for two workgroups per CU.
#include "wino4_common.hpp"
#ifdef SCIPNP_DIAG_BUILD
#include "../../include/scipnp_diag.h"
#endif

namespace scipnp {


// DIAG (timing experiments only, wrong results): bit0 no transform, 1 no raw staging, 2 no U DMA, 3 no barriers, 4 no MFMAs, 5 no epilogue
// PERSIST (round 4 experiment, NOT instantiated by the library): a grid of two workgroups per CU, each walking the units
// blockIdx.x, blockIdx.x + gridDim.x, ...: the next unit's first U slab and raw tiles are requested from inside the current
// unit's epilogue -- after the partial tiles have been exchanged through LDS, before the stores -- so their latency would run
// under the store phase instead of opening the next unit's life.  Correct (the F(4x4) tests pass on it), and slower: 330 us
// against 253 us on the FFDNet body layer, with or without a start stagger of half the grid (profiles/r04k_persist_stagger.txt).
// The unit loop keeps ~15 more values alive than the 256-register budget of two waves per SIMD holds next to the 144
// accumulators: 92 spilled registers, scratch accesses that share the vmcnt queue with the LDS-DMA requests (the compiler's
// own vmcnt(0) in front of a spill at the head of the K loop waits for every request in flight), and the two workgroups of a CU
// run in phase.  Same wall as round 3's persistent form; the classic form (one unit per workgroup: the loop below runs once) is
// the product.
// LINES (round 5, the product's plain-store epilogue): the partial tiles of both waves of a tile row go through LDS as a tile IMAGE
// and leave in whole 128-byte lines -- the per-lane stores of the classic epilogue (LINES = false) touch 64 lines with 16 bytes
// each per instruction and cost the layer 15 us (profiles/r05b_wino4_store_ablate.txt).
template <int TAG, int DIAG = 0, bool SHUF = false, bool PERSIST = false, bool LINES = (!SHUF && !PERSIST)>
__global__ void __launch_bounds__(W4_THREADS, 2)
conv3x3_c8w4_kernel(const Wino4Args a) {
    static_assert(!(PERSIST && SHUF), "the PixelShuffle epilogue assembles its tile in LDS: no requests may be in flight there");

    extern __shared__ __attribute__((aligned(16))) float smem_w4[];
    float* const raw_lds = smem_w4;                    // [2][RAW]
    float* const u_lds = smem_w4 + 2 * W4_RAW;         // [2][SLAB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long* const stamp_base = (DIAG & 64) ? a.dbg + (size_t)blockIdx.x * 128 : nullptr;
    (void)stamp_base;
    W4_STAMP(0);
    const int tg = wvu >> 1, xh = wvu & 1;             // tile row of the workgroup, half of the transformed rows
    const int tn = lane & 15, q = lane >> 4;           // tile along x, channel pair
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    // ---- per-unit state (a unit = one 8 x 64-pixel tile x one 32-channel output block; the classic form runs exactly one)
    int split = 0, n = 0, x0 = 0, y0 = 0;
    const float* w_g = nullptr;                                                 // advanced by NCB*SLAB per k-step
    const float* in_g = nullptr;                                                // advanced by HW*8 per group
    unsigned in_off[W4_IN_ITERS];
    const size_t w_step = (size_t)a.NCB * W4_SLAB;
    const unsigned plane_bytes = (unsigned)(HW * 32);
    (void)w_g; (void)plane_bytes; (void)in_g; (void)wvu;

    // `last`: no further group / k-step exists -- the pointer stays and the same data is fetched again (unused), so that
    // every step issues the same number of memory operations and the vmcnt waits below are constants
    // one 1 KiB LDS-DMA piece of the raw tile / of the U slab.  A wave's buffer_load ... lds takes some 40 cycles of its issue
    // time (tools/probes/wino4_stamps.py, one workgroup per CU: a k-step with its eleven requests in a row at its head is 450
    // cycles longer), so inside the K loop the pieces go out one at a time behind the first MFMA of a U vector.
    auto issue_raw_piece = [&](float* dst, int k) {
        if (DIAG & 2) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, plane_bytes, 0x00020000);
        int pc = wvu + 4 * k;
        if (pc >= W4_RAW_PIECES) pc -= 4;
        // (requested with the nt hint the tiles' halo rows are no longer served from L2: DDnet 13.6 -> 14.8 ms, profiles/r05zz_w4_load_nt.txt)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16, in_off[k], 0, 0, 0);
#endif
    };
    auto issue_u_piece = [&](float* dst, int k) {
        if (DIAG & 4) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, W4_SLAB * 4, 0x00020000);
        int pc = wvu + 4 * k;
        if (pc >= W4_PIECES) pc -= 4;                                           // (waves 2, 3: their fifth piece is their fourth again)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16,
                                                 (unsigned)(1024 * pc + 16 * lane), 0, 0, 0);
#endif
    };
    // `last`: no further group / k-step exists -- the pointer stays and the same data is fetched again (unused), so that
    // every step issues the same number of memory operations and the vmcnt waits below are constants
    auto raw_done = [&](bool last) { if (!last) in_g += HW * 8; };
    auto u_done = [&](bool last) { if (!last) w_g += w_step; };
    auto issue_raw = [&](float* dst, bool last) {
#pragma unroll
        for (int k = 0; k < W4_IN_ITERS; ++k) issue_raw_piece(dst, k);
        raw_done(last);
    };

    // ---- a unit's opening requests: U slab of k-step 0, raw tiles of groups 0 and 1 (in this order: the wait at the head of the
    // K loop -- all but the W4_IN_ITERS youngest requests, plus whatever stores follow -- then covers the slab and the first tile)
    auto begin_unit = [&](unsigned unit) {
        // XCD-aware order (as conv_wino.hip): one XCD works through a contiguous run of (tile, co-block) pairs, the co-blocks of
        // a tile adjacent, so the input tile is fetched from HBM once
        unsigned lin = unit;
        {
            const unsigned total = a.total_units;
            if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
        }
        unsigned r_split, r_bx, r_by;
        unsigned t = w4_div(lin, (unsigned)a.NCB, a.m_ncb, r_split);
        split = (int)r_split;
        w_g = a.wpk + (size_t)split * W4_SLAB;
        // the U slab of k-step 0 is requested HERE, before anything else is known about the unit: its addresses need only the
        // output-channel split, and its latency then runs under the address arithmetic of the raw tile below
#pragma unroll
        for (int k = 0; k < W4_DMA_ITERS; ++k) issue_u_piece(u_lds, k);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        w_g += w_step;
        t = w4_div(t, (unsigned)a.ntx, a.m_ntx, r_bx);
        n = (int)w4_div(t, (unsigned)a.nty, a.m_nty, r_by);
        x0 = (int)r_bx * W4_TW;
        y0 = (int)r_by * W4_TH;
        // staging plan of the raw tile: LDS unit u = 64 * piece + lane = (hf * THP + r) * RSL + slot -> pixel (r, c) of the halo tile,
        // channels 4hf .. 4hf+3; wave w fetches the pieces w, w + 4, ...  Pixels outside the image (and the padding units of the last
        // piece) get an offset past the buffer descriptor's range and arrive as zeros.
#pragma unroll
        for (int k = 0; k < W4_IN_ITERS; ++k) {
            int pc = wvu + 4 * k;
            if (pc >= W4_RAW_PIECES) pc -= 4;
            const int u = pc * 64 + lane;
            const int hf = u >= W4_UNITS / 2 ? 1 : 0;
            const int v = u - hf * (W4_UNITS / 2);
            // v / 70 and sl / 17 as multiply-shifts (exact for v < 1259, sl < 70: W4_RSL = 70 slots per row, 17 per 16 pixels)
            static_assert(W4_RSL == 70 && W4_UNITS / 2 < 1259, "the reciprocal 937 / 2^16 is exact for v < 1259 only");
            const int r = (int)(((unsigned)v * 937u) >> 16), sl = v - r * W4_RSL;
            const int g17 = (int)(((unsigned)sl * 241u) >> 12);                      // sl / 17
            const int c = sl - g17;                                                  // 16 (sl / 17) + sl % 17; sl % 17 == 16: a padding slot
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            // (bitwise, not short-circuit: a guarded offset compiles to an exec-masked branch per piece)
            const bool ok = (u < W4_UNITS) & (sl - 17 * g17 != 16) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            const unsigned off = (unsigned)((gy * W + gx) * 32 + 16 * hf);
            in_off[k] = ok ? off : 0xFFFFFF00u;
            if constexpr ((DIAG & 256) != 0) {
                // timing only (WRONG results): every piece = both halves of 32 consecutive pixels of a row, 1 KB contiguous (8
                // cache lines instead of 16) -- what a request costs as a function of the lines it touches
                const int row = pc >> 1, px = (pc & 1) * 32 + (lane >> 1);
                const int gy2 = y0 - 1 + (row < W4_THP ? row : 0), gx2 = x0 - 1 + px;
                const bool ok2 = ((unsigned)gy2 < (unsigned)H) & ((unsigned)gx2 < (unsigned)W);
                in_off[k] = ok2 ? (unsigned)((gy2 * W + gx2) * 32 + 16 * (lane & 1)) : 0xFFFFFF00u;
            }
        }
        in_g = a.in + (size_t)n * a.CGin * HW * 8;
        issue_raw(raw_lds, a.CGin <= 1);
        issue_raw(raw_lds + W4_RAW, a.CGin <= 2);
    };

    f32x4 acc[3][6][2];                                 // [own row xi - 3 xh][nu][co half]; zeroed behind the first requests
    // DIAG bit 7 (timing only, WRONG results): the same accumulator registers as nine 32x32 blocks, multiplied by
    // v_mfma_f32_32x32x2_f32 -- half as many matrix instructions of twice the length, half the U operands
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc16[3][3];
    (void)acc16;

    // per-lane LDS offsets (floats): patch of tile (tg, tn), channel pair q (half-pixel plane q >> 1, 8 bytes (q & 1) of the
    // unit); U vectors of half xh
    // (persistent form: re-derived from the lane number at the head of every unit, see loop_offsets(), so that they do not stay
    // live through the epilogue, whose output transform needs every register)
    int b_off0 = 0, b_off1 = 0, a_off = 0;
    auto loop_offsets = [&]() {
        int l = lane;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (PERSIST) asm volatile("" : "+v"(l));                                   // opaque: a fresh value every unit
#endif
        const int tn_ = l & 15, q_ = l >> 4;
        const int b_row = (((q_ >> 1) * W4_THP + 4 * tg + xh) * W4_RSL) * 4 + (q_ & 1) * 2;  // wave xh reads the patch rows xh .. xh + 4
        b_off0 = b_row + (4 * tn_ + (tn_ >> 2)) * 4;                                         // columns 0..3
        b_off1 = b_row + (4 * tn_ + ((tn_ + 1) >> 2)) * 4;                                   // columns 4, 5
        a_off = xh * (9 * 256) + l * 4;                                                      // + v * 256
    };

    const int CG = a.CGin;
    // one vector of the k-step slab, stored by the packer in the order the k-step walks its accumulators (see pack_wino4_kernel)
    auto u_vec = [&](const float* ucur, int pos) { return *(const f32x4*)(ucur + a_off + pos * 256); };
    // the four MFMAs of one U vector -- positions (x, 2np), (x, 2np+1) x the two output-channel halves, k-step J -- and the four
    // packed vector operations of the input transform that go with it (ops(P, i), P = the vector's number in the k-step).
    // W4_VBLK = how the vector operations are placed: 1 = one behind each MFMA, 4 = four behind each vector's MFMAs, 12 = twelve
    // behind every third vector.  Two waves on a SIMD interleave best 1 : 1 (1.7 cycles of the matrix pipe per packed operation
    // against 3.2 in blocks of eight, tools/probes/mfma_valu_coissue.py), but a wave's own vector instruction waits for its own
    // MFMA to finish (46 cycles per MFMA + operation for a wave alone on its SIMD), and the arbiter serves the OLDER wave first:
    // the older workgroup of a CU runs like a lone one and the younger fills the gaps (tools/probes/wino4_stamps.py), so what
    // counts is the lone wave's pace -- fewer, larger blocks.
#ifndef W4_VBLK
#define W4_VBLK 1
#endif
    auto quad = [&](const f32x4 u, const f32x2 (&Vr)[6], f32x4 (&ac)[6][2], auto NP, auto J, auto P, auto&& ops, auto&& dma) {
        constexpr int np = decltype(NP)::value, j = decltype(J)::value, pos = decltype(P)::value;
        const float b0 = Vr[2 * np][j], b1 = Vr[2 * np + 1][j];
        auto op = [&](auto Q, auto I) { ops(Q, I); };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
        if constexpr ((DIAG & 128) != 0) {
            f32x16& a16 = acc16[pos % 3][pos / 3 % 3];
            a16 = __builtin_amdgcn_mfma_f32_32x32x2f32(u[0], b0, a16, 0, 0, 0);
            dma(P);
            op(P, C0{}); op(P, C1{});
            a16 = __builtin_amdgcn_mfma_f32_32x32x2f32(u[2], b1, a16, 0, 0, 0);
            op(P, C2{}); op(P, C3{});
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            return;
        }
        if (!(DIAG & 16)) ac[2 * np][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[0], b0, ac[2 * np][0], 0, 0, 0);
        dma(P);                                            // (this vector's share of the k-step's LDS-DMA requests)
        if constexpr (W4_VBLK == 1) op(P, C0{});
        if (!(DIAG & 16)) ac[2 * np + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[1], b1, ac[2 * np + 1][0], 0, 0, 0);
        if constexpr (W4_VBLK == 1) op(P, C1{});
        if (!(DIAG & 16)) ac[2 * np][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[2], b0, ac[2 * np][1], 0, 0, 0);
        if constexpr (W4_VBLK == 1) op(P, C2{});
        if (!(DIAG & 16)) ac[2 * np + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[3], b1, ac[2 * np + 1][1], 0, 0, 0);
        if constexpr (W4_VBLK == 1) op(P, C3{});
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr (W4_VBLK == 4) { op(P, C0{}); op(P, C1{}); op(P, C2{}); op(P, C3{}); }
        if constexpr (W4_VBLK == 12 && pos % 3 == 2) {
            static_for<3>([&](auto QQ) {
                using Q = std::integral_constant<int, pos - 2 + decltype(QQ)::value>;
                op(Q{}, C0{}); op(Q{}, C1{}); op(Q{}, C2{}); op(Q{}, C3{});
            });
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (W4_VBLK != 1) __builtin_amdgcn_sched_barrier(0);
#endif
    };

    // The K loop for one transform flavour (LO: this wave owns the rows 0..2 of the transformed patch, else the rows 3..5).
    //   T  = own rows of the column pass B^T d of the NEXT group's patch, formed during k-step (g, 1), one patch-column pair per
    //        third of its MFMAs (the k-step walks its accumulators column-pair by column-pair, so the V entries it has finished
    //        with make room);
    //   V  = own rows of B^T d B, formed from T by the row pass during k-step (g, 0), one row per third of its MFMAs (that k-step
    //        walks row by row; only the first row's pass runs ahead of the MFMAs).
    auto k_loop = [&](auto LO, int stores_behind) {
        constexpr bool lo = decltype(LO)::value;
        f32x2 T[3][6], V[3][6];
        f32x2 ta = {0.f, 0.f}, tb = {0.f, 0.f};
        const f32x2 m5 = {-5.f, -5.f};
        if constexpr (DIAG != 0) {
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) T[x][nu] = V[x][nu] = f32x2{(float)lane, 1.f};
        }
        // own five rows of patch column c of the tile in rawp
        auto load_col = [&](const float* rawp, int c, f32x2 (&x)[5]) {
            const int bo = (c < 4 ? b_off0 : b_off1) + c * 4;
#pragma unroll
            for (int r = 0; r < 5; ++r) x[r] = *(const f32x2*)(rawp + bo + r * (W4_RSL * 4));
        };
        // operation K (0..5) of the column pass of one patch column: x = own five patch rows -> own three rows of B^T d
        auto col_op = [&](const f32x2 (&x)[5], f32x2& o0, f32x2& o1, f32x2& o2, auto K) {
            if (DIAG & 1) return;
            half_op<lo, decltype(K)::value>(x[0], x[1], x[2], x[3], x[4], o0, o1, o2, ta, tb, m5);
        };
        // operation K (0..11) of the row pass of own row r: T[r][0..5] -> V[r][0..5]
        auto row_op = [&](auto R, auto K) {
            if (DIAG & 1) return;
            constexpr int r = decltype(R)::value, k = decltype(K)::value;
            if constexpr (k < 6) half_op<true, k>(T[r][0], T[r][1], T[r][2], T[r][3], T[r][4], V[r][0], V[r][1], V[r][2], ta, tb, m5);
            else half_op<false, k - 6>(T[r][1], T[r][2], T[r][3], T[r][4], T[r][5], V[r][3], V[r][4], V[r][5], ta, tb, m5);
        };

        {   // head of a unit: begin_unit() has requested U of k-step 0 and the raw tiles of groups 0 and 1; column pass of group 0
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int nu = 0; nu < 6; ++nu)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[x][nu][h] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr ((DIAG & 128) != 0) {
#pragma unroll
                for (int x = 0; x < 3; ++x)
#pragma unroll
                    for (int y = 0; y < 3; ++y)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc16[x][y][r] = 0.f;
            }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            // landed by now: U of k-step 0 and the first raw tile -- everything but the W4_IN_ITERS requests of the second tile and
            // (persistent form) the previous unit's output stores, which were issued behind them: vmcnt counts in issue order
            if (stores_behind == 16) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(W4_IN_ITERS + 16) : "memory");
            else if (stores_behind == 8) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(W4_IN_ITERS + 8) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(W4_IN_ITERS) : "memory");
            W4_STAMP(1);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                f32x2 x[5];
                load_col(raw_lds, c, x);
                static_for<6>([&](auto K) { col_op(x, T[0][c], T[1][c], T[2][c], K); });
            }
            W4_STAMP(2);
        }
        for (int g = 0; g < CG; ++g) {
            float* const rcur = raw_lds + (g & 1) * W4_RAW;          // held group g (its column pass is done): free behind the next barrier
            float* const rnext = raw_lds + ((g & 1) ^ 1) * W4_RAW;   // group g+1, requested one k-step ago
            // ---- k-step (g, 0), row by row.  Behind the last barrier every wave has finished k-step 2g-1: U of k-step 2g+1 -> its buffer
            f32x4 af[3];
            af[0] = u_vec(u_lds, 0);
            af[1] = u_vec(u_lds, 1);
            static_for<12>([&](auto K) { row_op(std::integral_constant<int, 0>{}, K); });
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            static_for<9>([&](auto P) {
                constexpr int pos = decltype(P)::value, x = pos / 3, np = pos % 3;
                if constexpr (pos + 2 < 9) af[(pos + 2) % 3] = u_vec(u_lds, pos + 2);
                quad(af[pos % 3], V[x], acc[x], std::integral_constant<int, np>{}, std::integral_constant<int, 0>{}, P, [&](auto Q, auto I) {
                    // vector Q = 3 x' + np' of the k-step carries the operations 4 np' .. 4 np' + 3 of the row pass of row x' + 1
                    constexpr int qx = decltype(Q)::value / 3, qn = decltype(Q)::value % 3;
                    if constexpr (qx < 2) row_op(std::integral_constant<int, qx + 1>{}, std::integral_constant<int, 4 * qn + decltype(I)::value>{});
                }, [&](auto Q) {
                    if constexpr (decltype(Q)::value < W4_DMA_ITERS) issue_u_piece(u_lds + W4_SLAB, decltype(Q)::value);
                });
            });
            u_done(2 * g + 2 >= 2 * CG);
            // bare s_barrier (__syncthreads() is a fence too and would wait for every LDS-DMA in flight, whatever the count).  Landed
            // by now: the raw tile of group g+1 and U of k-step 2g+1 (every wave's own pieces); own LDS reads are done.
            if (g < 24) W4_STAMP(8 + 4 * g);
            if (DIAG & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (g < 24) W4_STAMP(9 + 4 * g);
            // ---- k-step (g, 1), column pair by column pair, with the column pass of group g+1: U of k-step 2g+2 -> the buffer of
            // k-step 2g, raw tile of group g+2 -> the buffer of group g (two k-steps ahead of its use)
            // (the last groups request nothing past the end: a re-fetched tile would only be waited for before the output transform)
            const bool more_u = g + 1 < CG, more_raw = g + 2 < CG;
            const float* const ub = u_lds + W4_SLAB;
            af[0] = u_vec(ub, 0);
            af[1] = u_vec(ub, 1);
            f32x2 xa[5], xb[5];                                      // own rows of the even / odd patch column of the current pair
            load_col(rnext, 0, xa);
            load_col(rnext, 1, xb);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            // 36 column-pass operations (column c: operations 6c .. 6c+5, even columns from xa, odd ones from xb), four under each U
            // vector from the SECOND vector on -- the first vector's MFMAs cover the latency of the first patch reads -- and the last
            // four behind the last vector.  T is free since the row passes of k-step (g, 0).
            auto col_ops4 = [&](auto Q, auto I) {                                          // operation 4Q + I of the 36
                constexpr int k = 4 * decltype(Q)::value + decltype(I)::value, c = k / 6;
                if constexpr (c % 2 == 0) col_op(xa, T[0][c], T[1][c], T[2][c], std::integral_constant<int, k % 6>{});
                else col_op(xb, T[0][c], T[1][c], T[2][c], std::integral_constant<int, k % 6>{});
                // the next pair's patch values: xa is free after operation 6c+5 of an even column, xb after that of an odd one
                if constexpr (k % 6 == 5 && c + 2 < 6) load_col(rnext, c + 2, c % 2 == 0 ? xa : xb);
            };
            static_for<9>([&](auto P) {
                constexpr int pos = decltype(P)::value, np = pos / 3, x = pos % 3;
                if constexpr (pos + 2 < 9) af[(pos + 2) % 3] = u_vec(ub, pos + 2);
                quad(af[pos % 3], V[x], acc[x], std::integral_constant<int, np>{}, std::integral_constant<int, 1>{}, P, [&](auto Q, auto I) {
                    if constexpr (W4_VBLK == 12) col_ops4(Q, I);                            // (whole column pairs behind their third of the MFMAs)
                    else if constexpr (decltype(Q)::value > 0) col_ops4(std::integral_constant<int, decltype(Q)::value - 1>{}, I);
                }, [&](auto Q) {
                    // the U pieces first (two per vector), then the raw pieces: the wait at the end of the k-step counts on that order
                    constexpr int qq = decltype(Q)::value;
                    if constexpr (2 * qq < W4_DMA_ITERS) { if (more_u) issue_u_piece(u_lds, 2 * qq); }
                    if constexpr (2 * qq + 1 < W4_DMA_ITERS) { if (more_u) issue_u_piece(u_lds, 2 * qq + 1); }
                    constexpr int r0 = (W4_DMA_ITERS + 1) / 2;                              // first vector that carries a raw piece
                    if constexpr (qq >= r0 && qq - r0 < W4_IN_ITERS) { if (more_raw) issue_raw_piece(rcur, qq - r0); }
                });
            });
            static_assert((W4_DMA_ITERS + 1) / 2 + W4_IN_ITERS <= 9, "a k-step has nine U vectors to hang its requests on");
            u_done(2 * g + 3 >= 2 * CG);
            raw_done(g + 3 >= CG);
            // (no raw requests behind the U pieces in the last two groups: the U pieces are then the youngest requests)
            if (!more_raw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (W4_VBLK != 12) static_for<4>([&](auto I) { col_ops4(std::integral_constant<int, 8>{}, I); });
            if (g < 24) W4_STAMP(10 + 4 * g);
            if (DIAG & 8) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W4_IN_ITERS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W4_IN_ITERS) : "memory");   // U of k-step 2g+2
            if (g < 24) W4_STAMP(11 + 4 * g);
        }
    };
    unsigned unit = blockIdx.x;
    begin_unit(unit);
    int stores_behind = 0;              // output stores of the previous unit issued behind this unit's opening requests (persistent form)
    for (;;) {
        loop_offsets();
        if (xh == 0) k_loop(std::true_type{}, stores_behind);
        else k_loop(std::false_type{}, stores_behind);
        W4_STAMP(3);
        if constexpr ((DIAG & 128) != 0) {                  // (keeps the blocks alive into the output transform)
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int y = 0; y < 3; ++y)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[x][2 * y + (r >> 3)][(r >> 2) & 1][r & 3] = acc16[x][y][r];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the re-fetched last slab must not land in the exchange buffer)
        __syncthreads();

        // ---- output transform.  Own rows xi: R[xi][j] = sum_nu M[xi][nu] A^T[j][nu], then the partial tile P[i][j] = sum_xi A^T[i][xi] R[xi][j];
        // wave xh finishes the output rows 2xh, 2xh+1 and hands the other two to its partner through LDS.
        // lane: tile (tg, tn), channels 32*split + 16*h + 4*q + r.
        if (DIAG & 32) return;
        if constexpr (LINES) {
            // image [xh 2][grp 2][row 8][tile 16][36 floats: 4 pixels x 8 channels + 4 of padding (conflict-free 16-byte writes)]: per
            // output-channel half h one round -- every wave writes its partial tile P[0..3][j] (all four output rows), then thread
            // (unit = (grp, row, tile), chunk c of 8) adds the two waves' partials, bias, residual / ReLU / mask, and stores 16 bytes:
            // eight consecutive lanes cover one 128-byte line = 4 pixels x 8 channels, a store instruction eight whole lines.
            // (keep + other) of the classic epilogue is LO + HI for the rows 0, 1 and HI + LO for 2, 3: the same sums.
            constexpr int TILE_F = 36, IMG_F = 2 * 8 * 16 * TILE_F;
            static_assert((size_t)2 * IMG_F * 4 <= W4_LDS_BYTES, "two images in the loop's LDS");
            float* const img = smem_w4;
            const float* bias = a.wpk + (size_t)2 * a.CGin * w_step;
            const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
            (void)relu; (void)add_res; (void)mask; (void)bias;
            bool wrote = false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cog0 = split * 4 + h * 2;
                if (cog0 >= a.CGout) continue;                       // workgroup-uniform
                if (wrote) __syncthreads();                          // the previous round's reads are done
                wrote = true;
                f32x4 R[3][4];
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const f32x4 m0 = acc[x][0][h], m1 = acc[x][1][h], m2 = acc[x][2][h], m3 = acc[x][3][h], m4 = acc[x][4][h], m5 = acc[x][5][h];
                    const f32x4 s1 = m1 + m2, d1 = psub4(m1, m2), s2 = m3 + m4, d2 = psub4(m3, m4);
                    R[x][0] = (m0 + s1) + s2;
                    R[x][1] = pk_fma(splat<f32x4>(2.f), d2, d1);
                    R[x][2] = pk_fma(splat<f32x4>(4.f), s2, s1);
                    R[x][3] = pk_fma(splat<f32x4>(8.f), d2, d1) + m5;
                }
                float* const dst = img + xh * IMG_F + (((q >> 1) * 8 + 4 * tg) * 16 + tn) * TILE_F + 4 * (q & 1);
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
                    for (int i = 0; i < 4; ++i) *(f32x4*)(dst + i * (16 * TILE_F) + j * 8) = P[i];
                }
                __syncthreads();
                if (h == 0) W4_STAMP(4);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int gr = 0; gr < 2; ++gr) {
                    const int cog = cog0 + gr;
                    if (cog >= a.CGout) continue;                    // workgroup-uniform
                    const int c = tid & 7;
                    const f32x4 bs = *(const f32x4*)(bias + cog * 8 + 4 * (c & 1));
                    const size_t plane = ((size_t)n * a.CGout + cog) * HW * 8;
                    auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + plane), 0, plane_bytes, 0x00020000);
                    f32x4 v[4];
                    unsigned off[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int unit = (tid >> 3) + 32 * it, tile = unit & 15, row = unit >> 4;   // row 0..7
                        const float* src = img + ((gr * 8 + row) * 16 + tile) * TILE_F + 4 * c;
                        v[it] = (*(const f32x4*)src + *(const f32x4*)(src + IMG_F)) + bs;
                        const int y = y0 + row, x = x0 + 4 * tile + (c >> 1);
                        off[it] = (y < H && x < W) ? (unsigned)((y * W + x) * 32 + 16 * (c & 1)) : 0x80000000u;
                    }
                    if (add_res) {
                        auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual + plane), 0, plane_bytes, 0x00020000);
#pragma unroll
                        for (int it = 0; it < 4; ++it)
                            v[it] = v[it] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off[it], 0, 0));
                    }
                    if (relu) {
#pragma unroll
                        for (int it = 0; it < 4; ++it)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[it][e] = fmaxf(v[it][e], 0.f);
                    }
                    if (mask) {
                        auto r_m = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask_src + plane), 0, plane_bytes, 0x00020000);
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const f32x4 fw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_m, off[it], 0, 0));
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[it][e] = (fw[e] > 0.f) ? v[it][e] : 0.f;
                        }
                    }
                    // Cache hints on the stores.  With the nt hint the ISOLATED body layer takes 227 - 230 us instead of 246 - 249
                    // (profiles/r05d_wino4_store_hints.txt), but inside a network pass, where every launch reads what the previous
                    // one wrote, it depends on the size: outputs that the 256 MB Infinity Cache can hold next to the layer's input are
                    // served from there when the stores allocate (256x256x16 tile, 50 MB per launch: 1.31 -> 1.36 ms per iteration
                    // with nt), at 100 MB per launch it is even, larger outputs leave faster without allocating (1024x1024x8: 10.53
                    // -> 10.34 ms; DDnet 13.81 -> 13.56 ms, FastDVDnet 5.50 -> 5.43 ms; profiles/r05zz_w4nt_*).  The entry point
                    // sets W4_FLAG_NT for launches whose output is 128 MB or more; DIAG bits 10 / 11 force nt / sc1 (laboratory).
#if defined(W4_STORE_AUX)                            /* variant build (make w4variant W4FLAGS=-DW4_STORE_AUX=n): cache-policy bits of every whole-line store */
                    constexpr int AUX = W4_STORE_AUX;
#elif defined(W4_STORE_NT)                           /* variant build: every whole-line store with the nt hint */
                    constexpr int AUX = 2;
#else
                    constexpr int AUX = (DIAG & 1024) ? 2 : (DIAG & 2048) ? 16 : 0;
#endif
#if defined(W4_STORE_AUX)
                    if (false) {
#else
                    if (AUX == 0 && (a.flags & W4_FLAG_NT)) {            // (workgroup-uniform)
#endif
#pragma unroll
                        for (int it = 0; it < 4; ++it)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[it]), r_out, off[it], 0, 2);
                    } else {
#pragma unroll
                        for (int it = 0; it < 4; ++it)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[it]), r_out, off[it], 0, AUX);
                    }
                }
#endif
            }
            break;                                                   // (LINES excludes the persistent form: one unit per workgroup)
        }
        float* const xbuf = smem_w4;                        // [wave 4][slot 16][lane 64][4]
        f32x4 keep[2][4][2];                                // [row 2xh + il][j][h]
    #pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 R[3][4];
    #pragma unroll
            for (int x = 0; x < 3; ++x) {
                const f32x4 m0 = acc[x][0][h], m1 = acc[x][1][h], m2 = acc[x][2][h], m3 = acc[x][3][h], m4 = acc[x][4][h], m5 = acc[x][5][h];
                const f32x4 s1 = m1 + m2, d1 = psub4(m1, m2), s2 = m3 + m4, d2 = psub4(m3, m4);
                R[x][0] = (m0 + s1) + s2;
                R[x][1] = pk_fma(splat<f32x4>(2.f), d2, d1);
                R[x][2] = pk_fma(splat<f32x4>(4.f), s2, s1);
                R[x][3] = pk_fma(splat<f32x4>(8.f), d2, d1) + m5;
            }
            auto finish = [&](auto LO) {
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 P[4];
                    if constexpr (decltype(LO)::value) {        // rows xi = 0, 1, 2 of A^T: (1,0,0,0) (1,1,1,1) (1,-1,1,-1)
                        const f32x4 s = R[1][j] + R[2][j], d = psub4(R[1][j], R[2][j]);
                        P[0] = R[0][j] + s; P[1] = d; P[2] = s; P[3] = d;
                    } else {                                    // rows xi = 3, 4, 5: (1,2,4,8) (1,-2,4,-8) (0,0,0,1)
                        const f32x4 s = R[0][j] + R[1][j], d = psub4(R[0][j], R[1][j]);
                        P[0] = s; P[1] = d * 2.f; P[2] = s * 4.f; P[3] = pk_fma(splat<f32x4>(8.f), d, R[2][j]);
                    }
                    constexpr int KEEP = decltype(LO)::value ? 0 : 2, SEND = decltype(LO)::value ? 2 : 0;
    #pragma unroll
                    for (int il = 0; il < 2; ++il) {
                        keep[il][j][h] = P[KEEP + il];
                        *(f32x4*)(xbuf + ((wvu * 16 + (il * 4 + j) * 2 + h) * 64 + lane) * 4) = P[SEND + il];
                    }
                }
            };
            if (xh == 0) finish(std::true_type{});
            else finish(std::false_type{});
        }
        __syncthreads();
        W4_STAMP(4);
        const float* bias = a.wpk + (size_t)2 * a.CGin * w_step;
        const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
        (void)relu; (void)add_res; (void)mask; (void)bias;
        if constexpr (SHUF) {
            // PixelShuffle(2) folded into the store (flags bit3, as scipnp_conv3x3_c8w): conv channel 4c + 2dy + dx -> channel c of pixel
            // (2y + dy, 2x + dx); the 32 conv channels of this workgroup are the 8 channels of output group `split`, and a lane's four
            // values are the 2x2 sub-pixels of ONE output channel c = 4h + q.  The shuffled 16 x 128-pixel tile is assembled in LDS (the
            // exchange area is free once every wave has read its partner's rows) and leaves in whole 128-byte lines: one thread = four
            // consecutive pixels x 8 channels, skip tensor (same layout) added there.  out = relu?(conv + bias + res).
            constexpr int ROW = 128 * 8 + 32 * 4;               // floats per shuffled row: 4 floats of padding per 4 pixels
            static_assert(16 * ROW * 4 <= W4_LDS_BYTES, "shuffled tile");
            float* const tile = smem_w4;
            f32x4 v[2][2][4];
    #pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 bs = *(const f32x4*)(bias + (split * 4 + h * 2 + (q >> 1)) * 8 + 4 * (q & 1));
    #pragma unroll
                for (int il = 0; il < 2; ++il)
    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[h][il][j] = (keep[il][j][h] + *(const f32x4*)(xbuf + (((wvu ^ 1) * 16 + (il * 4 + j) * 2 + h) * 64 + lane) * 4)) + bs;
            }
            __syncthreads();                                    // partner rows read: the area becomes the shuffled tile
    #pragma unroll
            for (int h = 0; h < 2; ++h)
    #pragma unroll
                for (int il = 0; il < 2; ++il)
    #pragma unroll
                    for (int j = 0; j < 4; ++j)
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int yl = 2 * (4 * tg + 2 * xh + il) + (e >> 1), xl = 2 * (4 * tn + j) + (e & 1);
                            tile[yl * ROW + xl * 8 + (xl >> 2) * 4 + 4 * h + q] = v[h][il][j][e];
                        }
            __syncthreads();
    #if defined(__HIP_DEVICE_COMPILE__)
            const int H2 = 2 * H, W2 = 2 * W;
            const size_t plane = ((size_t)n * a.NCB + split) * (size_t)H2 * W2 * 8;               // floats; NCB = Cout/32 output groups
            const unsigned pbytes = (unsigned)((size_t)H2 * W2 * 32);
            auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + plane), 0, pbytes, 0x00020000);
            auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)((add_res ? a.residual : a.out) + plane), 0, pbytes, 0x00020000);
            // 16 rows x 32 groups of 4 pixels = 512 lines of 128 bytes, 4096 pieces of 16 bytes: piece s * 256 + tid -> line (piece >> 3),
            // chunk (tid & 7): the eight lanes of a line write it in ONE store instruction (round 5; until then a thread wrote its own
            // 128 bytes in eight instructions, each of which touched 64 lines with 16 bytes -- the pattern the LINES epilogue removed
            // from the plain store)
            const int chunk = tid & 7;
    #pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 px[8];
                unsigned off[8];
    #pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int line = (half * 8 + s) * 32 + (tid >> 3), yl = line >> 5, xg = line & 31;
                    const int y2 = 2 * y0 + yl, x2 = 2 * x0 + 4 * xg + (chunk >> 1);
                    px[s] = *(const f32x4*)(tile + yl * ROW + xg * 36 + 4 * chunk);
                    off[s] = (y2 < H2 && x2 < W2) ? (unsigned)(((size_t)y2 * W2 + x2) * 32 + 16 * (chunk & 1)) : 0x80000000u;
                }
                if (add_res) {
    #pragma unroll
                    for (int s = 0; s < 8; ++s)
                        px[s] = px[s] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off[s], 0, 0));
                }
                if (relu) {
    #pragma unroll
                    for (int s = 0; s < 8; ++s)
    #pragma unroll
                        for (int e = 0; e < 4; ++e) px[s][e] = fmaxf(px[s][e], 0.f);
                }
                if (a.flags & W4_FLAG_NT) {                      // (outputs of 128 MB or more: see the whole-line epilogue)
    #pragma unroll
                    for (int s = 0; s < 8; ++s)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, px[s]), r_out, off[s], 0, 2);
                } else {
    #pragma unroll
                    for (int s = 0; s < 8; ++s)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, px[s]), r_out, off[s], 0, 0);
                }
            }
    #endif
            return;
        }
        // ---- plain store.  Everything that READS memory comes first (partner rows from LDS, bias, residual / mask source), then -- in
        // the persistent form -- the next unit's opening requests, then the 16 stores of this unit: the requests' latency runs
        // under the store phase, and no load of this epilogue ever queues behind them (vmcnt counts in issue order).
        f32x4 v[2][2][4];
        unsigned off[2][2][4];
        bool h_on[2];
        const float* outp[2];
        int le = lane;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (PERSIST) asm volatile("" : "+v"(le));        // (tile / channel-pair numbers re-derived here: not live through the K loop)
#endif
        const int tn = le & 15, q = le >> 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cog0 = split * 4 + h * 2;                    // this lane's group: cog0 + (q >> 1)
            h_on[h] = cog0 < a.CGout;                              // wave-uniform
            outp[h] = a.out + ((size_t)n * a.CGout + cog0) * HW * 8;
            if (!h_on[h]) continue;
            const bool lane_ok = cog0 + (q >> 1) < a.CGout;
            const f32x4 bs = *(const f32x4*)(bias + (cog0 + (q >> 1)) * 8 + 4 * (q & 1));          // bias holds CoutP entries
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = y0 + 4 * tg + 2 * xh + il, x = x0 + 4 * tn + j;
                    const f32x4 other = *(const f32x4*)(xbuf + (((wvu ^ 1) * 16 + (il * 4 + j) * 2 + h) * 64 + lane) * 4);
                    v[h][il][j] = (keep[il][j][h] + other) + bs;
                    off[h][il][j] = (lane_ok && y < H && x < W)
                                        ? (unsigned)((y * W + x) * 32 + 16 * (q & 1)) + (unsigned)(q >> 1) * plane_bytes
                                        : 0x80000000u;
                    if constexpr ((DIAG & 512) != 0)    // timing only (WRONG results): every store instruction writes 1 KB of contiguous memory
                        off[h][il][j] = (unsigned)(((y0 + 4 * tg + 2 * xh + il) * W + x0) * 32 + j * 1024 + 16 * lane) % (2 * plane_bytes - 16);
                }
#if defined(__HIP_DEVICE_COMPILE__)
            const size_t half0 = ((size_t)n * a.CGout + cog0) * HW * 8;                            // floats
            if (add_res) {
                auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
                for (int il = 0; il < 2; ++il)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[h][il][j] = v[h][il][j] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off[h][il][j], 0, 0));
            }
            if (relu) {
#pragma unroll
                for (int il = 0; il < 2; ++il)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[h][il][j][e] = fmaxf(v[h][il][j][e], 0.f);
            }
            if (mask) {   // ReLU backward: pass the gradient where the forward activation was > 0
                auto r_m = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask_src + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
                for (int il = 0; il < 2; ++il)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 fw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_m, off[h][il][j], 0, 0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[h][il][j][e] = (fw[e] > 0.f) ? v[h][il][j][e] : 0.f;
                    }
            }
#endif
        }
        bool more = false;
        const unsigned next = unit + gridDim.x;
        if constexpr (PERSIST) {
            more = next < a.total_units;
            if (more) {
                __syncthreads();                                   // every wave has read its partner's rows: the LDS is free again
                begin_unit(next);                                  // (overwrites split, n, x0, y0: the stores below use outp / off)
            }
        }
        stores_behind = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!h_on[h]) continue;
            auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp[h], 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[h][il][j]), r_out, off[h][il][j], 0, 0);
            stores_behind += 8;
        }
#endif
        if (!more) break;
        unit = next;
    }
    W4_STAMP(5);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr ((DIAG & 64) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4_STAMP(6);                                       // stores acknowledged
        if (tid == 0) a.dbg[(size_t)blockIdx.x * 128 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                                                           (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));      // XCC_ID, HW_ID
        asm volatile("s_dcache_wb" ::: "memory");
    }
#endif
}

// U = G g G^T from the fp32 direct packing [cig][tap][CoutP][8]; one thread per slab element
__device__ __forceinline__ void pack_wino4_body(const float* __restrict__ pk, float* __restrict__ out, int CGin, int CoutP, size_t i) {
    const size_t NCB = CoutP / 32;
    const size_t total = (size_t)2 * CGin * NCB * W4_SLAB;
    if (i < total) {
        // slab element index: [xh 2][v 9][lane 64 = q*16 + tn][e 4] -> U_p[co = 32 cb + 16 (e >> 1) + tn][ci = 2 q + j],
        // p = (xi = 3 xh + own row, nu = 2 np + (e & 1)), (own row, np) = (v / 3, v % 3) for j = 0 and (v % 3, v / 3) for j = 1;
        // slab index = (2 cig + j) * NCB + cb
        const int el = (int)(i % W4_SLAB);
        const size_t sl = i / W4_SLAB;
        const int cb = (int)(sl % NCB), ks = (int)(sl / NCB);
        const int cig = ks >> 1, j = ks & 1;
        const int e = el & 3, tnl = (el >> 2) & 15, ql = (el >> 6) & 3, vv = (el >> 8) % 9, xhh = (el >> 8) / 9;
        // vector order = the order the k-step walks its accumulators: j = 0 row by row, j = 1 column pair by column pair
        const int xl = j ? vv % 3 : vv / 3, npp = j ? vv / 3 : vv % 3;
        const int xi = 3 * xhh + xl, nu = 2 * npp + (e & 1), h = e >> 1;
        const int co = cb * 32 + h * 16 + tnl, ci = 2 * ql + j;
        const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
        double u = 0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                u += G[xi][ky] * G[nu][kx] * (double)pk[(((size_t)cig * 9 + ky * 3 + kx) * CoutP + co) * 8 + ci];
        out[i] = (float)u;
    }
    if (i < (size_t)CoutP) out[total + i] = pk[(size_t)CGin * 9 * CoutP * 8 + i];       // bias
}
__global__ void pack_wino4_kernel(const float* __restrict__ pk, float* __restrict__ out, int CGin, int CoutP) {
    pack_wino4_body(pk, out, CGin, CoutP, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// ... of MANY layers in one launch (blockIdx.y = the job): see pack_device_multi_kernel (finetune.hip)
constexpr int PACK4_MULTI_MAX = 32;
struct PackWino4Jobs {
    const float* pk[PACK4_MULTI_MAX];
    float* out[PACK4_MULTI_MAX];
    int cgin[PACK4_MULTI_MAX], coutp[PACK4_MULTI_MAX];
};
__global__ void pack_wino4_multi_kernel(const PackWino4Jobs j) {
    const int q = blockIdx.y;
    pack_wino4_body(j.pk[q], j.out[q], j.cgin[q], j.coutp[q], (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

static inline int round_up_w4(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace scipnp

using namespace scipnp;

extern "C" {

#ifndef SCIPNP_DIAG_BUILD   /* ---- product entries (libscipnp.so) */

size_t scipnp_conv3x3_wino4_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    const int CoutP = round_up_w4(Cout, 32);
    return (size_t)2 * (Cin / 8) * (CoutP / 32) * W4_SLAB + CoutP;
}

int scipnp_pack_conv3x3_wino4_multi(int n, const float* const* packed_f32, float* const* packed_wino4, const int* Cin,
                                    const int* Cout, scipnp_stream_t s) {
    SCIPNP_REQUIRE(n >= 0 && (n == 0 || (packed_f32 && packed_wino4 && Cin && Cout)), "bad arguments");
    for (int base = 0; base < n; base += PACK4_MULTI_MAX) {
        const int m = n - base < PACK4_MULTI_MAX ? n - base : PACK4_MULTI_MAX;
        PackWino4Jobs j = {};
        size_t most = 0;
        for (int q = 0; q < m; ++q) {
            const int g = base + q;
            SCIPNP_REQUIRE(packed_f32[g] && packed_wino4[g] && Cin[g] > 0 && Cout[g] > 0 && Cin[g] % 8 == 0 && Cout[g] % 8 == 0,
                           "bad arguments in job %d", g);
            j.pk[q] = packed_f32[g]; j.out[q] = packed_wino4[g];
            j.cgin[q] = Cin[g] / 8; j.coutp[q] = round_up_w4(Cout[g], 32);
            const size_t total = (size_t)2 * j.cgin[q] * (j.coutp[q] / 32) * W4_SLAB;
            most = total > most ? total : most;
        }
        hipLaunchKernelGGL(pack_wino4_multi_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)m), dim3(256), 0, (hipStream_t)s, j);
    }
    return launch_status("pack_wino4_multi_kernel");
}

int scipnp_pack_conv3x3_wino4(const float* packed_f32, float* packed_wino4, int Cin, int Cout, scipnp_stream_t s) {
    SCIPNP_REQUIRE(packed_f32 && packed_wino4, "null pointer");
    SCIPNP_REQUIRE(Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad channel counts Cin=%d Cout=%d", Cin, Cout);
    SCIPNP_ALIGNED(packed_f32); SCIPNP_ALIGNED(packed_wino4);
    const int CoutP = round_up_w4(Cout, 32);
    const size_t total = (size_t)2 * (Cin / 8) * (CoutP / 32) * W4_SLAB;
    hipLaunchKernelGGL(pack_wino4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, packed_f32,
                       packed_wino4, Cin / 8, CoutP);
    return launch_status("pack_wino4_kernel");
}

int scipnp_conv3x3_c8w4(const float* in, const float* packed_wino4, float* out, const float* residual, const float* mask_src,
                        int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_wino4 && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0,
                   "bad shape n=%d Cin=%d Cout=%d h=%d w=%d (channels must be multiples of 8)", n, Cin, Cout, h, w);
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed_wino4); SCIPNP_ALIGNED(out);
    if (residual) SCIPNP_ALIGNED(residual);
    if (mask_src) SCIPNP_ALIGNED(mask_src);
    SCIPNP_REQUIRE(!(flags & (4 | 0x200)), "the F(4x4,3x3) kernel is stride 1, 8-row workgroups");
    SCIPNP_REQUIRE(!(flags & 8) || (Cout % 32 == 0 && !(flags & 16)), "PixelShuffle store: Cout must be a multiple of 32, no ReLU-mask epilogue");
    SCIPNP_REQUIRE(!(flags & 8) || (long long)h * w * 4 * 32 < (1ll << 30), "shuffled plane too large for 32-bit buffer offsets");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    Wino4Args a;
    a.in = in; a.wpk = packed_wino4; a.out = out; a.residual = residual; a.mask_src = mask_src; a.dbg = nullptr;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w4(Cout, 32) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W4_TH - 1) / W4_TH;
    a.m_ncb = w4_magic(a.NCB); a.m_ntx = w4_magic(a.ntx); a.m_nty = w4_magic(a.nty);
    a.flags = flags;
    if ((long long)n * Cout * h * w * 4 >= W4_NT_BYTES) a.flags |= W4_FLAG_NT;      // (see the store epilogue)
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    a.total_units = (unsigned)total;
    const int tag = (flags & 0x100) ? 1 : 0;
    const void* fns[2] = {(const void*)conv3x3_c8w4_kernel<0>, (const void*)conv3x3_c8w4_kernel<1>};
    static LdsAttrOnce attr[2];
    if (int rc = attr[tag].ensure(fns[tag], W4_LDS_BYTES, "conv3x3_c8w4")) return rc;
    const dim3 grid((unsigned)total), block(W4_THREADS);
    if (flags & 8) {                                            // PixelShuffle(2) store (UpBlocks of FastDVDnet / DDnet)
        static LdsAttrOnce shuf_attr;
        if (int rc = shuf_attr.ensure((const void*)conv3x3_c8w4_kernel<0, 0, true>, W4_LDS_BYTES, "conv3x3_c8w4 shuffle")) return rc;
        hipLaunchKernelGGL((conv3x3_c8w4_kernel<0, 0, true>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);
        return launch_status("conv3x3_c8w4_kernel<shuffle>");
    }
    if (tag) hipLaunchKernelGGL((conv3x3_c8w4_kernel<1>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);
    else hipLaunchKernelGGL((conv3x3_c8w4_kernel<0>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);
    return launch_status("conv3x3_c8w4_kernel");
}

/* The whole FFDNet pass with the layers that have an F(4x4,3x3) packing (packed_wino4[l] != NULL; NULL array: none) on
 * scipnp_conv3x3_c8w4 and the others on scipnp_conv3x3_c8w. */
int scipnp_ffdnet_forward_c8w4(const float* in_c8, float* out_c8, const float* const* packed_wino, const float* const* packed_wino4,
                               int nb, int nc, float* scratch0, float* scratch1, int B, int M, int N, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in_c8 && out_c8 && packed_wino && scratch0 && scratch1, "null pointer");
    SCIPNP_REQUIRE(nb >= 2 && nc % 8 == 0 && nc > 0, "bad network shape nb=%d nc=%d", nb, nc);
    float* buf[2] = {scratch0, scratch1};
    auto layer = [&](int l, const float* in, float* out, int cin, int cout, int flags) {
        if (packed_wino4 && packed_wino4[l])
            return scipnp_conv3x3_c8w4(in, packed_wino4[l], out, nullptr, nullptr, B, cin, cout, M, N, flags, s);
        return scipnp_conv3x3_c8w(in, packed_wino[l], out, nullptr, nullptr, B, cin, cout, M, N, flags, s);
    };
    int rc = layer(0, in_c8, buf[0], 16, nc, 1 | 0x100);
    if (rc) return rc;
    int cur = 0;
    for (int l = 1; l < nb - 1; ++l) {
        rc = layer(l, buf[cur], buf[cur ^ 1], nc, nc, 1);
        if (rc) return rc;
        cur ^= 1;
    }
    return layer(nb - 1, buf[cur], out_c8, nc, 16, 0x100);
}

#else   /* ---- SCIPNP_DIAG_BUILD: the laboratory entries (libscipnp_diag.so, include/scipnp_diag.h); the product library holds none */

/* DIAGNOSTIC instantiation with s_memtime stamps of wave 0 of every workgroup (128 words per workgroup; tools/probes/wino4_stamps.py):
 * [0] entry, [1] first tiles / slab in LDS, [2] first column pass done, [8 + 4g + {0, 1, 2, 3}] k-step (g, 0) MFMAs issued | its
 * barrier passed | k-step (g, 1) MFMAs issued | its barrier passed (g < 24), [3] loop left, [4] partial tiles exchanged,
 * [5] stores issued, [6] stores acknowledged, [7] XCC_ID << 32 | HW_ID.  Results are the product kernel's. */
int scipnp_conv3x3_c8w4_stamped(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_wino4 && out && stamps, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad shape");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    Wino4Args a;
    a.in = in; a.wpk = packed_wino4; a.out = out; a.residual = nullptr; a.mask_src = nullptr; a.dbg = stamps;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w4(Cout, 32) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W4_TH - 1) / W4_TH;
    a.m_ncb = w4_magic(a.NCB); a.m_ntx = w4_magic(a.ntx); a.m_nty = w4_magic(a.nty);
    a.flags = flags & 1;
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    a.total_units = (unsigned)total;
    const dim3 grid((unsigned)total), block(W4_THREADS);
    // flags bits 12.. select a stamped build with parts switched off (diag bits 0..2 of scipnp_conv3x3_c8w4_diag; wrong results)
#define W4_STAMP_CASE(D)                                                                                                     \
    case D: {                                                                                                                \
        static LdsAttrOnce attr;                                                                                             \
        if (int rc = attr.ensure((const void*)conv3x3_c8w4_kernel<0, 64 | D>, (size_t)160 * 1024, "conv3x3_c8w4 stamped")) return rc; \
        hipLaunchKernelGGL((conv3x3_c8w4_kernel<0, 64 | D>), grid, block, lds_req, (hipStream_t)s, a);                       \
        break;                                                                                                               \
    }
    // (experiment: SCIPNP_W4_ONE_PER_CU=1 asks for the whole LDS, i.e. a single resident workgroup per CU)
    const size_t lds_req = getenv("SCIPNP_W4_ONE_PER_CU") ? (size_t)160 * 1024 : W4_LDS_BYTES;
    switch ((flags >> 12) & 7) {
        W4_STAMP_CASE(0) W4_STAMP_CASE(1) W4_STAMP_CASE(2) W4_STAMP_CASE(4) W4_STAMP_CASE(6) W4_STAMP_CASE(7)
        default: SCIPNP_REQUIRE(false, "no stamped build for that mask");
    }
#undef W4_STAMP_CASE
    return launch_status("conv3x3_c8w4_kernel<stamped>");
}

/* diagnostic: the same kernel with parts switched off (timing only, WRONG results) -- tools/probes/wino4_ablate.py.
 * diag: bit0 no input transform, bit1 no raw-tile staging, bit2 no U LDS-DMA, bit3 no barriers in the K loop, bit4 no MFMAs,
 * bit5 no output transform / stores, bit7 the matrix work as v_mfma_f32_32x32x2_f32 on the same registers (half the instructions,
 * twice their length, half the U operands), bit8 every raw-tile request fetches 1 KB of contiguous memory, bit9 every output store instruction writes 1 KB of contiguous memory */
int scipnp_conv3x3_c8w4_diag(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                             int flags, int diag, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_wino4 && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad shape");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    Wino4Args a;
    a.in = in; a.wpk = packed_wino4; a.out = out; a.residual = nullptr; a.mask_src = nullptr; a.dbg = nullptr;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w4(Cout, 32) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W4_TH - 1) / W4_TH;
    a.m_ncb = w4_magic(a.NCB); a.m_ntx = w4_magic(a.ntx); a.m_nty = w4_magic(a.nty);
    a.flags = flags & 1;
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    a.total_units = (unsigned)total;
    const dim3 grid((unsigned)total), block(W4_THREADS);
#define W4_DIAG_CASE(D)                                                                                                    \
    case D: {                                                                                                              \
        static LdsAttrOnce attr;                                                                                           \
        if (int rc = attr.ensure((const void*)conv3x3_c8w4_kernel<0, D>, W4_LDS_BYTES, "conv3x3_c8w4 diag")) return rc;   \
        hipLaunchKernelGGL((conv3x3_c8w4_kernel<0, D>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);                     \
        break;                                                                                                             \
    }
    if (diag == 4096) {                                     // the classic per-lane store epilogue (LINES = false), results unchanged
        static LdsAttrOnce attr;
        if (int rc = attr.ensure((const void*)conv3x3_c8w4_kernel<0, 0, false, false, false>, W4_LDS_BYTES, "conv3x3_c8w4 classic stores")) return rc;
        hipLaunchKernelGGL((conv3x3_c8w4_kernel<0, 0, false, false, false>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);
        return launch_status("conv3x3_c8w4_kernel<classic stores>");
    }
    if (diag == 8192 || diag == 8193) {                     // the persistent form (PERSIST = true; results unchanged): 2 (8192) or 1 (8193) workgroups per CU walk the units
        static LdsAttrOnce attr;
        if (int rc = attr.ensure((const void*)conv3x3_c8w4_kernel<0, 0, false, true>, (size_t)160 * 1024, "conv3x3_c8w4 persistent")) return rc;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const unsigned per = diag == 8192 ? 2u : 1u;
        const unsigned g = (unsigned)std::min<long long>(total, (long long)per * cus);
        hipLaunchKernelGGL((conv3x3_c8w4_kernel<0, 0, false, true>), dim3(g), block, per == 1 ? (size_t)160 * 1024 : W4_LDS_BYTES, (hipStream_t)s, a);
        return launch_status("conv3x3_c8w4_kernel<persistent>");
    }
    switch (diag) {
        W4_DIAG_CASE(1) W4_DIAG_CASE(2) W4_DIAG_CASE(4) W4_DIAG_CASE(8) W4_DIAG_CASE(16) W4_DIAG_CASE(32) W4_DIAG_CASE(6)
        W4_DIAG_CASE(7) W4_DIAG_CASE(15) W4_DIAG_CASE(39) W4_DIAG_CASE(47) W4_DIAG_CASE(48) W4_DIAG_CASE(49) W4_DIAG_CASE(55)
        W4_DIAG_CASE(63) W4_DIAG_CASE(3) W4_DIAG_CASE(5) W4_DIAG_CASE(9) W4_DIAG_CASE(10) W4_DIAG_CASE(12) W4_DIAG_CASE(14)
        W4_DIAG_CASE(128) W4_DIAG_CASE(256) W4_DIAG_CASE(512) W4_DIAG_CASE(1024) W4_DIAG_CASE(2048)
        default: SCIPNP_REQUIRE(false, "diag mask %d has no instantiation", diag);
    }
#undef W4_DIAG_CASE
    return launch_status("conv3x3_c8w4_kernel<diag>");
}

#endif  /* SCIPNP_DIAG_BUILD */

}  // extern "C"
