// fp32 Winograd F(4x4,3x3) convolution with PRODUCER / CONSUMER waves (round 5): one transform and one raw tile for 64 output channels.
//
// What rounds 4 and 5 measured on conv_wino4.hip and its prototypes: on this part the fp32 MFMA issues on the
// SIMD's vector port, so a wave's packed transform operations, its LDS-DMA requests (40 - 100 issue cycles each) and its MFMAs
// exclude each other, and the launch is their SUM; more resident waves only overlap the workgroup boundaries.  The sum shrinks
// when fewer transform operations and requests are issued PER MFMA, i.e. when more output channels share a tile's transform --
// which the accumulators forbid inside one wave (144 registers for 18 positions x 32 channels).  Hence wave specialisation:
//
//   workgroup = 12 waves = one per wave slot of a CU at 168 VGPRs (3 per SIMD), one workgroup per CU, one barrier per k-step
//   8 CONSUMER waves  (tile row tg) x (position half xh) x (channel block cb of two): 36 MFMAs per k-step, operands from LDS only
//                     (U as in conv_wino4.hip: one 16-byte vector feeds four MFMAs; V: one dword per position), 144 accumulators
//   4 PRODUCER waves  (tile row tg) x (position half xh): every LDS-DMA request of the workgroup (raw tiles, the U slabs of BOTH
//                     channel blocks: 36 KB per k-step) and the input transform, ONCE for both blocks, written to LDS per k-step
//
// A SIMD holds two consumers and one producer (waves w, w + 4, w + 8): its vector port carries 72 packed operations and 24 requests
// per channel group beside 2 x 72 MFMAs, against 144 + 28 beside the same MFMAs in conv_wino4.hip.  Unit = 8 x 64 output pixels x
// 64 output channels (Cout must be a multiple of 64: FastDVDnet's 64 -> 64 and 128 -> 128 layers, the FFDNet-gray body).
// Same packed weights (scipnp_pack_conv3x3_wino4), same products, same accumulation order, same sums in the output transform as
// scipnp_conv3x3_c8w4: BIT-IDENTICAL results.
//
// LDS (150 KB): raw halo tiles [2] (as conv_wino4.hip), U slabs [2 k-step buffers][2 channel blocks] (72 KB), V [2 k-step buffers]
// [tg, xh][9 position pairs][64 lanes][2] (36 KB).  Schedule, k-step s = 2g + j (every wave passes ONE barrier per k-step, so the roles
// cannot fall out of step):
//   consumers   MFMAs of k-step s on U[s & 1], V[j]
//   producers   requests: U of k-step s + 1; at j = 0 the raw tile of group g + 2
//               j = 0: V(g, 1) from their registers -> V[1]; column pass of group g + 1 (its tile landed two barriers ago)
//               j = 1: row pass of group g + 1 -> registers; V(g + 1, 0) -> V[0]
//               s_waitcnt vmcnt(0) before the barrier: whatever a k-step requested has landed when the next one starts
#include "wino4_common.hpp"

namespace scipnp {

constexpr int WP_WAVES = 12, WP_THREADS = 64 * WP_WAVES;
constexpr int WP_U = 2 * W4_SLAB;                            // floats of one k-step's U (two channel blocks)
constexpr int WP_V = 4 * 18 * 64;                            // floats of one k-step's V
constexpr int WP_U_ITERS = (W4_PIECES + 3) / 4;                  // U pieces per CONSUMER and k-step: its own channel block's slab over the block's four waves (5; some fetched twice)
constexpr int WP_IN_ITERS = (W4_RAW_PIECES + 3) / 4;         // raw pieces per producer and group (6; some fetched twice)
constexpr size_t WP_LDS_BYTES = (2 * (size_t)W4_RAW + 2 * (size_t)WP_U + 2 * (size_t)WP_V) * sizeof(float);
constexpr int WP_TILE_F = 36, WP_IMG_F = 2 * 8 * 16 * WP_TILE_F;      // the epilogue's tile image (as conv_wino4.hip, LINES)
static_assert((size_t)2 * 2 * WP_IMG_F * 4 <= WP_LDS_BYTES, "two images per channel block in the loop's LDS");
static_assert(WP_LDS_BYTES <= 160 * 1024, "LDS of a CU");

template <int DIAG = 0>
__global__ void __launch_bounds__(WP_THREADS, 3)
conv3x3_c8wp_kernel(const Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem_wp[];
    float* const raw_lds = smem_wp;                          // [2][RAW]
    float* const u_lds = smem_wp + 2 * W4_RAW;               // [2][2][SLAB]
    float* const v_lds = u_lds + 2 * WP_U;                   // [2][4][18][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wvu >= 8;
    const unsigned long long* const stamp_base = (DIAG & 64) ? a.dbg + (size_t)blockIdx.x * 128 : nullptr;
    (void)stamp_base;
#if defined(__HIP_DEVICE_COMPILE__)
#define WP_STAMP(wave, slot)                                                                                        \
    do {                                                                                                            \
        if constexpr ((DIAG & 64) != 0) {                                                                           \
            if (wvu == (wave)) {                                                                                    \
                unsigned long long t_;                                                                              \
                const unsigned long long* p_ = stamp_base + (slot);                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : "=&s"(t_) : "s"(p_) : "memory"); \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
#else
#define WP_STAMP(wave, slot) (void)0
#endif
    WP_STAMP(0, 0);
    // consumers: wave = (tg, xh) * 2 + cb; producers: wave - 8 = (tg, xh)
    const int role = producer ? wvu - 8 : wvu >> 1;          // (tg, xh) pair 0..3
    const int cb = wvu & 1;
    const int tg = role >> 1, xh = role & 1;
    const int tn = lane & 15, q = lane >> 4;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int CG = a.CGin;
    const unsigned plane_bytes = (unsigned)(HW * 32);
    (void)plane_bytes;

    // ---- the unit: (tile, pair of channel blocks), pairs of a tile adjacent, XCD-aware order as conv_wino4.hip
    int pair, n, x0, y0;
    {
        unsigned lin = blockIdx.x;
        const unsigned total = a.total_units;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
        unsigned r_pair, r_bx, r_by;
        unsigned t = w4_div(lin, (unsigned)(a.NCB >> 1), a.m_ncb, r_pair);
        pair = (int)r_pair;
        t = w4_div(t, (unsigned)a.ntx, a.m_ntx, r_bx);
        n = (int)w4_div(t, (unsigned)a.nty, a.m_nty, r_by);
        x0 = (int)r_bx * W4_TW;
        y0 = (int)r_by * W4_TH;
    }
    const size_t w_step = (size_t)a.NCB * W4_SLAB;          // floats per k-step of the packing
    const int split = 2 * pair + cb;                        // this consumer's 32-channel block

    f32x4 acc[3][6][2];
    f32x2 T[3][6] = {}, V[3][6] = {};                        // producers only
    unsigned in_off[WP_IN_ITERS];
    const float* w_g = a.wpk + (size_t)(2 * pair) * W4_SLAB; // producers: U of the next k-step to request (both blocks: contiguous)
    const float* in_g = a.in + (size_t)n * a.CGin * HW * 8;  // producers: raw tile of the next group to request
    const int pw = wvu - 8;                                  // producer number 0..3
    (void)pw;

    auto issue_raw_piece = [&](float* dst, int k) {
        if (DIAG & 2) return;                                // (timing experiment: no raw staging, wrong results)
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, plane_bytes, 0x00020000);
        int pc = pw + 4 * k;
        if (pc >= W4_RAW_PIECES) pc -= 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16, in_off[k], 0, 0, 0);
#endif
    };
    // one 1 KiB piece of this wave's channel block's slab (consumers: piece role + 4k of the 18; the prologue's producers fetch both
    // blocks' first slab: piece pw + 4k of the 36).  A wave's LDS-DMA requests execute one after the other -- measured here: nine in
    // a row cost a producer 2000 cycles, 13 B/cycle for the CU -- so the requests are spread over ALL twelve waves, a few per wave
    // and k-step with MFMAs or transform work between them (profiles/r05f_winop4_stamps_*.txt)
    auto issue_u_piece = [&](const float* src, float* dst, int pc) {
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, WP_U * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16,
                                                 (unsigned)(1024 * pc + 16 * lane), 0, 0, 0);
#endif
    };
    // per-lane LDS offsets (floats)
    const int b_row = (((q >> 1) * W4_THP + 4 * tg + xh) * W4_RSL) * 4 + (q & 1) * 2;    // producers: patch rows xh .. xh + 4
    const int b_off0 = b_row + (4 * tn + (tn >> 2)) * 4, b_off1 = b_row + (4 * tn + ((tn + 1) >> 2)) * 4;
    const int a_off = cb * W4_SLAB + xh * (9 * 256) + lane * 4;                           // consumers: U vectors
    const int v_off = role * (9 * 128) + lane * 2;                                        // both: V[role][pair r * 3 + np][lane][2]
    const f32x2 m5 = {-5.f, -5.f};

    // producers: own rows of the column pass of the tile in rawp -> T; row pass T -> V (registers).  The row pass is written op-major:
    // operation k of all three rows and both halves back to back -- six independent packed operations per step.  A producer shares
    // its SIMD with two consumers whose fp32 MFMAs hold the vector port 32 cycles at a time: a DEPENDENT chain gets one operation in
    // per MFMA boundary (the 36 operations of a row pass took 2800 cycles; op-major 2200).
    auto column_pass = [&](const float* rawp, auto LO) {
        constexpr bool lo = decltype(LO)::value;
        if (DIAG & 4) return;                                // (timing experiment: no transform)
        // (column by column: with all 30 patch values requested up front and the operations op-major the even k-steps got SLOWER,
        // 33.6 -> 36.3 hundred cycles -- the burst delays the consumers' own operand reads)
        f32x2 ta = {0.f, 0.f}, tb = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int bo = (c < 4 ? b_off0 : b_off1) + c * 4;
            f32x2 x[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) x[r] = *(const f32x2*)(rawp + bo + r * (W4_RSL * 4));
            static_for<6>([&](auto K) { half_op<lo, decltype(K)::value>(x[0], x[1], x[2], x[3], x[4], T[0][c], T[1][c], T[2][c], ta, tb, m5); });
        }
    };
    auto row_pass = [&]() {
        if (DIAG & 4) return;
        f32x2 tr[3][2][2];
        static_for<6>([&](auto K) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                half_op<true, decltype(K)::value>(T[r][0], T[r][1], T[r][2], T[r][3], T[r][4], V[r][0], V[r][1], V[r][2], tr[r][0][0], tr[r][0][1], m5);
                half_op<false, decltype(K)::value>(T[r][1], T[r][2], T[r][3], T[r][4], T[r][5], V[r][3], V[r][4], V[r][5], tr[r][1][0], tr[r][1][1], m5);
            }
        });
    };
    auto store_v = [&](int j) {                              // V(., j) of the held group -> V[j], one 8-byte store per position pair
        if (DIAG & 1) return;                                // (timing experiment: no V stores)
        float* dst = v_lds + j * WP_V + v_off;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int np = 0; np < 3; ++np) *(f32x2*)(dst + (r * 3 + np) * 128) = f32x2{V[r][2 * np][j], V[r][2 * np + 1][j]};
    };

    if (producer) {
        // The producers' few vector operations must not queue behind the consumers' MFMAs: at equal priority the SIMD's arbiter serves
        // the two consumers' back-to-back MFMAs first and a producer's packed operation gets a slot every other MFMA -- its 36
        // operations of a k-step took 2800 cycles, longer than the k-step's 2304 cycles of matrix work, and the consumers waited at
        // every barrier (profiles/r05f_winop4_stamps_prio0.txt).  Priority 3: they issue at once and cost the pipe their 4 cycles each.
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_setprio(3);
#endif
        // ---- opening requests: U of k-step 0, raw tiles of groups 0 and 1
#pragma unroll
        for (int k = 0; k < (2 * W4_PIECES) / 4; ++k) issue_u_piece(w_g, u_lds, pw + 4 * k);
#pragma unroll
        for (int k = 0; k < WP_IN_ITERS; ++k) {
            int pc = pw + 4 * k;
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
            in_off[k] = ok ? (unsigned)((gy * W + gx) * 32 + 16 * hf) : 0xFFFFFF00u;
        }
#pragma unroll
        for (int k = 0; k < WP_IN_ITERS; ++k) issue_raw_piece(raw_lds, k);
        if (CG > 1) {
            in_g += HW * 8;
#pragma unroll
            for (int k = 0; k < WP_IN_ITERS; ++k) issue_raw_piece(raw_lds + W4_RAW, k);
        }
        if (CG > 2) in_g += HW * 8;                          // -> group 2
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // P1: the opening tiles and slab are in LDS
        WP_STAMP(8, 56);
        if (xh == 0) column_pass(raw_lds, std::true_type{});
        else column_pass(raw_lds, std::false_type{});
        row_pass();
        store_v(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // P2: V(0, 0) is in LDS
        WP_STAMP(8, 57);
        // ---- the K loop: 2 CG k-steps, ONE barrier per k-step in EITHER role's loop.  The two roles are two separate straight
        // paths through the kernel (the producers return below): where the paths met, the register allocator kept the consumers'
        // 144 accumulators and the producers' 72 transform registers live together (81 spilled registers)
        for (int g = 0; g < CG; ++g) {
            static_for<2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const int s = 2 * g + j;
                if constexpr (j == 0) {
                    if (g + 2 < CG) {                        // raw tile of group g + 2 -> the buffer of group g (column pass done): first half
#pragma unroll
                        for (int k = 0; k < WP_IN_ITERS / 2; ++k) issue_raw_piece(raw_lds + (g & 1) * W4_RAW, k);
                    }
                    store_v(1);                              // V(g, 1), held since the row pass of group g
                    if (g + 1 < CG) {
                        const float* rn = raw_lds + ((g + 1) & 1) * W4_RAW;
                        if (xh == 0) column_pass(rn, std::true_type{});
                        else column_pass(rn, std::false_type{});
                    }
                } else {
                    if (g + 2 < CG) {                        // ... second half (landed at this k-step's barrier, read two k-steps later)
#pragma unroll
                        for (int k = WP_IN_ITERS / 2; k < WP_IN_ITERS; ++k) issue_raw_piece(raw_lds + (g & 1) * W4_RAW, k);
                        in_g += HW * 8;
                    }
                    if (g + 1 < CG) {
                        row_pass();
                        store_v(0);                          // V(g + 1, 0)
                    }
                }
                if (s < 24) WP_STAMP(8, 64 + 2 * s);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (s < 24) WP_STAMP(8, 65 + 2 * s);
            });
        }
        // the epilogue's three barriers (see below): the producers have nothing else to do there
        __syncthreads();
        __syncthreads();
        __syncthreads();
        return;
    }
    // ================================================================ consumers
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[x][nu][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_barrier" ::: "memory");                                      // P1
    asm volatile("s_barrier" ::: "memory");                                      // P2
    WP_STAMP(0, 1);
    const float* wc_g = a.wpk + (size_t)split * W4_SLAB + w_step;                // this block's slab of the NEXT k-step
    // The k-step walks its accumulators row by row (j = 0) or column pair by column pair (j = 1), the order the packer laid the U
    // vectors out in.  Vector i of k-step j sits in operand slot (i + j) & 1; the LAST vector of a k-step is loaded but multiplied
    // only behind the barrier, after the next k-step's first operands have been requested: its four MFMAs run while those arrive
    // (both consumers of a SIMD pass the barrier together -- without this the pipe idles for an LDS round trip every k-step).
    f32x4 af[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};     // (slot 1 = the "held vector" of a k-step -1: zeros)
    f32x2 bv[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    auto mfma4 = [&](auto X, auto NP, const f32x4 u, const f32x2 b) {
        constexpr int x = decltype(X)::value, np = decltype(NP)::value;
        if (DIAG & 16) return;
        acc[x][2 * np][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[0], b[0], acc[x][2 * np][0], 0, 0, 0);
        acc[x][2 * np + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[1], b[1], acc[x][2 * np + 1][0], 0, 0, 0);
        acc[x][2 * np][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[2], b[0], acc[x][2 * np][1], 0, 0, 0);
        acc[x][2 * np + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[3], b[1], acc[x][2 * np + 1][1], 0, 0, 0);
    };
    for (int g = 0; g < CG; ++g) {
        static_for<2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int s = 2 * g + j;
            const float* const ucur = u_lds + (s & 1) * WP_U + a_off;
            const float* const vcur = v_lds + j * WP_V + v_off;
            float* const unext = u_lds + ((s + 1) & 1) * WP_U + cb * W4_SLAB;
            const bool more_u = s + 1 < 2 * CG;
            // (row, np) of vector i: j = 0: (i / 3, i % 3); j = 1: (i % 3, i / 3); its V pair: row * 3 + np
            af[j & 1] = *(const f32x4*)(ucur);
            bv[j & 1] = *(const f32x2*)(vcur);
            // the previous k-step's last vector (k-step j ^ 1: vector 8 = (2, 2) either way), held in slot (8 + (j ^ 1)) & 1
            mfma4(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, af[(j ^ 1) & 1], bv[(j ^ 1) & 1]);
            static_for<8>([&](auto P) {
                constexpr int pos = decltype(P)::value;
                constexpr int nx = pos + 1;
                constexpr int r1 = j == 0 ? nx / 3 : nx % 3, np1 = j == 0 ? nx % 3 : nx / 3;
                af[(nx + j) & 1] = *(const f32x4*)(ucur + nx * 256);
                bv[(nx + j) & 1] = *(const f32x2*)(vcur + (r1 * 3 + np1) * 128);
                constexpr int x = j == 0 ? pos / 3 : pos % 3, np = j == 0 ? pos % 3 : pos / 3;
                mfma4(std::integral_constant<int, x>{}, std::integral_constant<int, np>{}, af[(pos + j) & 1], bv[(pos + j) & 1]);
                if constexpr (pos < WP_U_ITERS) {              // one request of the next k-step's slab behind each of the first vectors
                    if (more_u && !(DIAG & 8)) {                 // (DIAG bit 3, timing experiment: no U requests)
                        int pc = role + 4 * pos;
                        if (pc >= W4_PIECES) pc -= 4;
#if defined(__HIP_DEVICE_COMPILE__)
                        // (the lane's offset 16 * lane rides in the register that already addresses its U vectors: a_off * 4 =
                        // KB + 16 * lane with the wave-uniform KB taken off the descriptor's base; the piece's 1 KiB offset is scalar)
                        const unsigned KB = (unsigned)((cb * W4_SLAB + xh * (9 * 256)) * 4);
                        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)wc_g - KB), 0, KB + W4_SLAB * 4, 0x00020000);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)((char*)unext + 1024 * pc), 16,
                                                                 (unsigned)(a_off * 4), (unsigned)(1024 * pc), 0, 0);
#endif
                    }
                }
            });
            if (more_u) wc_g += w_step;
            if (s < 24) WP_STAMP(0, 8 + 2 * s);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s < 24) WP_STAMP(0, 9 + 2 * s);
        });
    }
    // the very last vector (k-step 2 CG - 1, j = 1: slot (8 + 1) & 1)
    mfma4(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, af[1], bv[1]);
    WP_STAMP(0, 3);

    // ---- epilogue: the consumers' partial tiles through LDS as tile images (two per channel block), whole-line stores -- the LINES
    // epilogue of conv_wino4.hip, per channel block; the producers only keep the barrier count
    float* const img = smem_wp + cb * (2 * WP_IMG_F);
    const float* bias = a.wpk + (size_t)2 * a.CGin * w_step;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
    (void)relu; (void)add_res; (void)mask; (void)bias;
    const int ctid = role * 64 + lane;                       // consumers: thread 0..255 of the channel block's four waves
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1) __syncthreads();                         // round 0's reads are done
        {
            f32x4 R[3][4];
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const f32x4 m0 = acc[x][0][h], m1 = acc[x][1][h], m2 = acc[x][2][h], m3 = acc[x][3][h], m4 = acc[x][4][h], m5_ = acc[x][5][h];
                const f32x4 s1 = m1 + m2, d1 = psub4(m1, m2), s2 = m3 + m4, d2 = psub4(m3, m4);
                R[x][0] = (m0 + s1) + s2;
                R[x][1] = pk_fma(splat<f32x4>(2.f), d2, d1);
                R[x][2] = pk_fma(splat<f32x4>(4.f), s2, s1);
                R[x][3] = pk_fma(splat<f32x4>(8.f), d2, d1) + m5_;
            }
            float* const dst = img + xh * WP_IMG_F + (((q >> 1) * 8 + 4 * tg) * 16 + tn) * WP_TILE_F + 4 * (q & 1);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                f32x4 P[4];
                if (xh == 0) {
                    const f32x4 s_ = R[1][jj] + R[2][jj], d = psub4(R[1][jj], R[2][jj]);
                    P[0] = R[0][jj] + s_; P[1] = d; P[2] = s_; P[3] = d;
                } else {
                    const f32x4 s_ = R[0][jj] + R[1][jj], d = psub4(R[0][jj], R[1][jj]);
                    P[0] = s_; P[1] = d * 2.f; P[2] = s_ * 4.f; P[3] = pk_fma(splat<f32x4>(8.f), d, R[2][jj]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(dst + i * (16 * WP_TILE_F) + jj * 8) = P[i];
            }
        }
        __syncthreads();
        if (h == 0) WP_STAMP(0, 4);
#if defined(__HIP_DEVICE_COMPILE__)
        {
#pragma unroll
            for (int gr = 0; gr < 2; ++gr) {
                const int cog = split * 4 + h * 2 + gr;
                if (cog >= a.CGout) continue;
                const int c = ctid & 7;
                const f32x4 bs = *(const f32x4*)(bias + cog * 8 + 4 * (c & 1));
                const size_t plane = ((size_t)n * a.CGout + cog) * HW * 8;
                auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + plane), 0, plane_bytes, 0x00020000);
                f32x4 v[4];
                unsigned off[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int unit = (ctid >> 3) + 32 * it, tile = unit & 15, row = unit >> 4;
                    const float* src = img + ((gr * 8 + row) * 16 + tile) * WP_TILE_F + 4 * c;
                    v[it] = (*(const f32x4*)src + *(const f32x4*)(src + WP_IMG_F)) + bs;
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
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[it]), r_out, off[it], 0, 0);
            }
        }
#endif
    }
    WP_STAMP(0, 5);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr ((DIAG & 64) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WP_STAMP(0, 6);
        if (tid == 0) a.dbg[(size_t)blockIdx.x * 128 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                                                           (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
        asm volatile("s_dcache_wb" ::: "memory");
    }
#endif
}

}  // namespace scipnp

using namespace scipnp;

static int wp_fill_args(Wino4Args& a, const float* in, const float* packed_wino4, float* out, const float* residual,
                        const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags);

extern "C" {

#ifdef SCIPNP_DIAG_BUILD   /* LABORATORY only (libscipnp_diag.so): measured at parity with conv_wino4.hip inside a network pass */
#include "../scipnp_lab.h"
/* stamped instantiation: slots in the header comment of the kernel's WP_STAMP uses -- wave 0 (a consumer): [0] entry,
 * [1] P2 passed, [8 + 2s] MFMAs of k-step s issued, [9 + 2s] its barrier passed, [3] loop left, [4] first image written, [5] stores
 * issued, [6] acknowledged, [7] XCC_ID << 32 | HW_ID; wave 8 (a producer): [56] P1 passed, [57] P2 passed, [64 + 2s] work of k-step
 * s issued, [65 + 2s] its barrier passed */
int scipnp_conv3x3_c8wp_stamped(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s) {
    SCIPNP_REQUIRE(stamps, "null pointer");
    Wino4Args a;
    if (int rc = wp_fill_args(a, in, packed_wino4, out, nullptr, nullptr, n, Cin, Cout, h, w, flags & 1)) return rc;
    a.dbg = stamps;
    // flags bits 12.. : timing experiments (wrong results): 1 no V stores, 2 no raw staging, 4 no transform
#define WP_STAMP_CASE(D)                                                                                                  \
    case D: {                                                                                                             \
        static LdsAttrOnce attr;                                                                                          \
        if (int rc = attr.ensure((const void*)conv3x3_c8wp_kernel<64 | D>, WP_LDS_BYTES, "conv3x3_c8wp stamped")) return rc; \
        hipLaunchKernelGGL((conv3x3_c8wp_kernel<64 | D>), dim3(a.total_units), dim3(WP_THREADS), WP_LDS_BYTES, (hipStream_t)s, a); \
        break;                                                                                                            \
    }
    switch ((flags >> 12) & 15) {
        WP_STAMP_CASE(0) WP_STAMP_CASE(1) WP_STAMP_CASE(2) WP_STAMP_CASE(4) WP_STAMP_CASE(6) WP_STAMP_CASE(7) WP_STAMP_CASE(8) WP_STAMP_CASE(10)
        default: SCIPNP_REQUIRE(false, "no stamped build for that mask");
    }
#undef WP_STAMP_CASE
    return launch_status("conv3x3_c8wp_kernel<stamped>");
}

/* scipnp_conv3x3_c8w4's convolution for layers whose Cout is a multiple of 64, on the producer / consumer kernel above: the same
 * packed_wino4 buffer, flags bit0 ReLU, bit1 residual, bit4 ReLU-backward mask, bit8 ignored (no tag instantiation); stride 1, plain
 * store.  Bit-identical to scipnp_conv3x3_c8w4. */
int scipnp_conv3x3_c8wp(const float* in, const float* packed_wino4, float* out, const float* residual, const float* mask_src,
                        int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    Wino4Args a;
    if (int rc = wp_fill_args(a, in, packed_wino4, out, residual, mask_src, n, Cin, Cout, h, w, flags)) return rc;
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_c8wp_kernel<0>, WP_LDS_BYTES, "conv3x3_c8wp")) return rc;
    hipLaunchKernelGGL((conv3x3_c8wp_kernel<0>), dim3(a.total_units), dim3(WP_THREADS), WP_LDS_BYTES, (hipStream_t)s, a);
    return launch_status("conv3x3_c8wp_kernel");
}
#endif

}  // extern "C"

static int wp_fill_args(Wino4Args& a, const float* in, const float* packed_wino4, float* out, const float* residual,
                        const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags) {
    SCIPNP_REQUIRE(in && packed_wino4 && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 64 == 0,
                   "bad shape n=%d Cin=%d Cout=%d h=%d w=%d (Cin a multiple of 8, Cout a multiple of 64)", n, Cin, Cout, h, w);
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed_wino4); SCIPNP_ALIGNED(out);
    if (residual) SCIPNP_ALIGNED(residual);
    if (mask_src) SCIPNP_ALIGNED(mask_src);
    SCIPNP_REQUIRE(!(flags & (4 | 8 | 0x200)), "the producer / consumer F(4x4,3x3) kernel is stride 1, 8-row workgroups, plain store");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    a.in = in; a.wpk = packed_wino4; a.out = out; a.residual = residual; a.mask_src = mask_src; a.dbg = nullptr;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = Cout / 32;
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W4_TH - 1) / W4_TH;
    a.m_ncb = w4_magic(a.NCB / 2); a.m_ntx = w4_magic(a.ntx); a.m_nty = w4_magic(a.nty);
    a.flags = flags;
    const long long total = (long long)a.ntx * a.nty * n * (a.NCB / 2);
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    a.total_units = (unsigned)total;
    return SCIPNP_OK;
}
