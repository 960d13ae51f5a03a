// fp32 weight gradient of a 3x3 / pad 1 / stride 1 convolution in the Winograd F(4x4,3x3) domain
// (reference packages/ffdnet/test_ffdnet_ipol.py:296, packages/fastdvdnet/test_fastdvdnet.py:449 `loss.backward()`):
//   forward   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 4x4 output tile (A^T 4x6, G 6x3, B^T 6x6, points 0, +-1, +-2, inf)
//   gradient  dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
// 36 fp32 products per 16 outputs and channel pair: 2.25 per output against the 4 of the F(2x2) form (csrc/wgrad_wino.hip) and
// the 9 of the direct form (csrc/finetune.hip), accumulated on v_mfma_f32_32x32x2_f32 with the TILES as the K dimension:
//   dU_p[co][ci] = sum_t dM_p[co][t] V_p[ci][t],   p = 6 xi + nu.
// The price is rounding: the transforms carry factors up to 8 (A) and 5 (B^T), rel-L2 3e-6 against float64 where F(2x2) gives
// 5e-7 (tools/probes/wgrad4_numerics_sim.py) -- far inside the FFDNet trainer's gradient gate; G^T . G is applied once, in
// double, by the finish kernel.
//
// Workgroup = 12 waves, one per CU, 32 output x 32 input channels, persistent over chunks of 8 tiles (32 x 4 output pixels).
// LDS: the transformed chunk T = dM | V, [36 positions][8 tiles][32 channels] each (72 KB), and one raw buffer (43 KB) holding
// the chunk's 6 x 34 input pixels and 4 x 32 output-gradient pixels of the 4 + 4 channel groups as they lie in memory.
// Per chunk, two phases separated by workgroup barriers:
//   A  waves 8..11 (one per SIMD) read their (tile, channel pair)'s 4x4 / 6x6 raw values from LDS, transform them (80 / 144
//      packed operations) and write T;
//   B  waves 0..7 (two per SIMD) multiply: wave c <-> positions 9 (c & 3) .. + 8 and tiles 4 (c >> 2) .. + 3 (the K dimension
//      split in two, so that two waves interleave their MFMAs on every SIMD; 9 x 16 accumulator registers each; the halves are
//      added through LDS at the end).  Meanwhile waves 8..11 write the NEXT chunk's raw values, which they requested a chunk
//      ago, from registers to LDS, and request the chunk after it: 44 pieces of 64 lanes x 16 bytes, one image row of one channel
//      group each -- whole cache lines.
// How it got here (profiles/r04m_wgrad4_ablate.txt; 96 -> 96 on 8 x 256 x 256, reduction included):
//   * every (tile, channel pair) lane loading its own raw values into registers, T double-buffered, producers beside the
//     multiplying waves: 391-408 us.  A wave-load of 8 bytes per lane touches 16 cache lines for 512 bytes; the L1 takes 3.2 k
//     cycles per chunk for the 52 of them, the producers' vector instructions only issue when the MFMA stream of the same SIMD
//     pauses, and requesting two chunks ahead makes it worse (475 us: the lines are evicted before their neighbours' loads);
//   * raw values by LDS-DMA (no registers): right result, 386 us -- a CU's LDS-DMA sustained ~13 bytes per cycle here;
//   * this form: 382 us against 441 us of the F(2x2) kernel.  Phase A 1.7 k cycles (bound by the LDS: 53 KB read, 74 KB
//     written), phase B 2.2 k (the MFMAs' own time); the raw staging adds 0.8 k for the loads and 0.8 k for the LDS stores --
//     moving a chunk's 44 KB through the vector registers of a SIMD takes that SIMD's issue from its MFMA waves.
// A 32 x 32 block is what 36 positions' accumulators leave room for (9 x 16 registers per multiplying wave); it needs 12.7 bytes
// of LDS traffic per matrix cycle where the F(2x2) kernel's 96 x 32 block needs 5.3 -- the smaller product count is spent on
// data movement.  The (co block, ci block) workgroups of a slab walk the same chunks; whole slabs are placed on one XCD, where
// they meet in its L2 (TCC hit rate 19 % -> 60 %, HBM fetch 693 -> 330 MB per launch).
#include "common.hpp"
#ifdef SCIPNP_DIAG_BUILD
#include "../../include/scipnp_diag.h"
#endif
#include <type_traits>

// the laboratory build (libscipnp_diag.so links libscipnp.so) gets its own kernel symbols
#ifdef SCIPNP_DIAG_BUILD
#define conv3x3_wgrad_wino4_kernel conv3x3_wgrad_wino4_diag_kernel
#define wgrad_wino4_sum_kernel wgrad_wino4_diag_sum_kernel
#define wgrad_wino4_finish_kernel wgrad_wino4_diag_finish_kernel
#endif

namespace scipnp {

typedef float w4g_f32x16 __attribute__((ext_vector_type(16)));
typedef float w4g_f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4G_T = 8;                                    // tiles per chunk, along x
constexpr int W4G_POS = 36;
constexpr int W4G_CONS = 8;                                 // multiplying waves: 4 position groups x 2 tile halves
constexpr int W4G_PPW = 9;                                  // positions per multiplying wave
constexpr int W4G_THREADS = 64 * (W4G_CONS + 4);
constexpr int W4G_M_FLOATS = W4G_POS * W4G_T * 32;          // dM [p][tile][co]; V [p][tile][ci] behind it
constexpr int W4G_T_FLOATS = 2 * W4G_M_FLOATS;
// raw buffers, in units of 16 bytes (half a pixel of one channel group).  Input: 6 rows x 4 channel groups, 70 units each (34
// pixels + 1 of padding: row stride 1120 B = 96 mod 128, so that the four channel groups of a wave's 8-byte reads fall on
// different banks); output gradient: 4 rows x 4 channel groups, 66 units each (32 pixels + 1: 1056 B = 32 mod 128)
constexpr int W4G_VROW = 70, W4G_VUNITS = 24 * W4G_VROW, W4G_VPIECES = (W4G_VUNITS + 63) / 64;      // 1680, 27 (last: 16 lanes)
constexpr int W4G_DROW = 66, W4G_DUNITS = 16 * W4G_DROW, W4G_DPIECES = (W4G_DUNITS + 63) / 64;      // 1056, 17 (last: 32 lanes)
constexpr int W4G_RAW_BYTES = (W4G_VUNITS + W4G_DUNITS) * 16;                                        // 43776
constexpr size_t W4G_XCH_BYTES = (size_t)4 * W4G_PPW * 16 * 64 * sizeof(float);                       // the final exchange of the K halves
constexpr size_t W4G_LOOP_BYTES = (size_t)W4G_T_FLOATS * sizeof(float) + (size_t)W4G_RAW_BYTES;      // 117504
constexpr size_t W4G_LDS_BYTES = W4G_LOOP_BYTES > W4G_XCH_BYTES ? W4G_LOOP_BYTES : W4G_XCH_BYTES;   // 147456
static_assert(W4G_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// 1-D transforms on channel pairs.  A (6x4) = (A^T)^T:  m = A y;  B^T (6x6):  v = B^T d
__device__ __forceinline__ w4g_f32x2 w4g_fma(float k, const w4g_f32x2 x, const w4g_f32x2 y) {      // k x + y, one rounding
    return __builtin_elementwise_fma((w4g_f32x2)(k), x, y);
}
__device__ __forceinline__ void w4g_a(const w4g_f32x2 y0, const w4g_f32x2 y1, const w4g_f32x2 y2, const w4g_f32x2 y3,
                                      w4g_f32x2* m) {
    const w4g_f32x2 s02 = y0 + y2, s13 = y1 + y3;
    const w4g_f32x2 e = w4g_fma(4.f, y2, y0), o = w4g_fma(4.f, y3, y1);
    m[0] = y0;
    m[1] = s02 + s13;
    m[2] = s02 - s13;
    m[3] = w4g_fma(2.f, o, e);
    m[4] = w4g_fma(-2.f, o, e);
    m[5] = y3;
}
__device__ __forceinline__ void w4g_bt(const w4g_f32x2* d, w4g_f32x2* v) {
    const w4g_f32x2 a = w4g_fma(-4.f, d[2], d[4]), b = w4g_fma(-4.f, d[1], d[3]);
    const w4g_f32x2 c = d[4] - d[2], e = d[3] - d[1];
    v[0] = w4g_fma(-5.f, d[2], w4g_fma(4.f, d[0], d[4]));
    v[1] = a + b;
    v[2] = a - b;
    v[3] = w4g_fma(2.f, e, c);
    v[4] = w4g_fma(-2.f, e, c);
    v[5] = w4g_fma(-5.f, d[3], w4g_fma(4.f, d[1], d[5]));
}

// grid = nslab * ncob * ncib workgroups.  act: [n][CGin][h][w][8], dz: [n][CGout][h][w][8];
// slabs[slab][p 36][coP = 32 ncob][ciP = 32 ncib]
// dbg_arg (laboratory build only; timing runs, results meaningless): 1 no MFMAs, 2 no raw loads / stores, 4 no transform, 8 placement
// probe (HW_ID of every wave), 32 loads but no LDS stores of them, 64 stores but no loads
__global__ void __launch_bounds__(W4G_THREADS)
conv3x3_wgrad_wino4_kernel(const float* __restrict__ act, const float* __restrict__ dz, float* __restrict__ slabs, int n_img,
                           int CGin, int CGout, int H, int W, int nslab, int ncob, int ncib, int dbg_arg) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef SCIPNP_DIAG_BUILD
    const int dbg = dbg_arg;                                       // timing-only ablations (tools/probes/wgrad4_ablate.py)
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem_w4g[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: the role branches below are uniform
    const int li = lane & 31, lh = lane >> 5;
    if (dbg & 8) {                                                 // placement probe: HW_ID of every wave
        if (lane == 0) ((unsigned*)slabs)[blockIdx.x * 12 + wave] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        return;
    }
    // workgroup -> (slab, co block, ci block): consecutive workgroup ids go round the XCDs, so with nslab % 8 == 0 the blocks
    // of a slab get ids that are congruent mod 8
    const int nblk = ncob * ncib;
    int slab, blk;
    {
        // consecutive workgroup ids go round the 8 XCDs.  Each XCD gets `local` whole slabs (all blocks of a slab on one XCD:
        // they fetch the same chunks and meet in its L2); the workgroups left over form the remaining slabs across XCDs.
        const int id = blockIdx.x;
        const int local = (int)(gridDim.x >> 3) / nblk;            // whole slabs per XCD
        if (id < 8 * local * nblk) {
            const int xcd = id & 7, j = id >> 3;
            slab = xcd * local + j / nblk;
            blk = j % nblk;
        } else {
            const int r = id - 8 * local * nblk;
            slab = 8 * local + r / nblk;
            blk = r % nblk;
        }
    }
    const int cob = blk / ncib, cib = blk - cob * ncib;
    const size_t HW = (size_t)H * W;
    const int tiles_x = (W + 3) / 4, tiles_y = (H + 3) / 4;
    const int chunks_x = (tiles_x + W4G_T - 1) / W4G_T;
    const int chunks = n_img * tiles_y * chunks_x;
    const int first = slab, step = nslab;
    const int coP = 32 * ncob, ciP = 32 * ncib;
    float* const Tbuf = smem_w4g;
    char* const raw0 = (char*)(smem_w4g + W4G_T_FLOATS);
    if (first >= chunks) {                                         // a slab without chunks is zeros
        if (wave < 4) {
            const int pg = wave;
#pragma unroll
            for (int i = 0; i < W4G_PPW; ++i) {
                float* out = slabs + (((size_t)slab * W4G_POS + W4G_PPW * pg + i) * coP + cob * 32) * ciP + cib * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * ciP] = 0.f;
            }
        }
        return;
    }

    if (wave < W4G_CONS) {
        // ------------------------------------------------------------ waves 0..7: MFMAs (phase B)
        w4g_f32x16 acc[W4G_PPW];
#pragma unroll
        for (int i = 0; i < W4G_PPW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // MFMA operands: A[row = co][k = tile], B[k = tile][col = ci]; lane (li, lh) supplies row / column li of K index lh
        const int pg = wave & 3, h = wave >> 2;
        const int a_lane = W4G_PPW * pg * W4G_T * 32 + (4 * h + lh) * 32 + li;        // + i * T*32 + (2 ks) * 32
        const int b_lane = W4G_M_FLOATS + a_lane;
        for (int chunk = first; chunk < chunks; chunk += step) {
            __syncthreads();                                       // X: T is free
            __syncthreads();                                       // Y: T holds `chunk`
            if (!(dbg & 1)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < W4G_PPW; ++i) {
                        const float av = Tbuf[a_lane + i * W4G_T * 32 + 2 * ks * 32];
                        const float bv = Tbuf[b_lane + i * W4G_T * 32 + 2 * ks * 32];
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                    }
            }
        }
        // the two tile halves of a position group are added through LDS (everything in it is dead now): h = 1 writes, h = 0 adds
        // (fixed order: deterministic) and stores.  [pg][i][r][lane]: conflict-free 4-byte accesses
        float* const xch = smem_w4g + (size_t)pg * W4G_PPW * 16 * 64 + lane;
        __syncthreads();                                           // Z1: the last chunk's T has been read by everybody
        if (h == 1) {
#pragma unroll
            for (int i = 0; i < W4G_PPW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(i * 16 + r) * 64] = acc[i][r];
        }
        __syncthreads();                                           // Z2
        if (h == 0) {
            // C[row = co_local][col = ci_local]: row = (r&3) + 8*(r>>2) + 4*lh, col = li.  A running pointer (rows +1 +1 +1 +5 ...):
            // sixteen row offsets held at once would not fit beside the accumulators
            float* out = slabs + (((size_t)slab * W4G_POS + W4G_PPW * pg) * coP + cob * 32 + 4 * lh) * ciP + cib * 32 + li;
            const size_t pos_stride = (size_t)coP * ciP - (size_t)27 * ciP;       // from row 27 of a position to row 0 of the next
#pragma unroll
            for (int i = 0; i < W4G_PPW; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    *out = acc[i][r] + xch[(i * 16 + r) * 64];
                    out += (r == 15) ? pos_stride : ((r & 3) == 3 ? (size_t)5 * ciP : (size_t)ciP);
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------- waves 8..11: transforms (phase A), raw staging (phase B)
    // The chunk's raw values travel memory -> registers -> LDS in pieces of 64 lanes x 16 bytes (one image row of one channel
    // group, whole cache lines): loaded in phase B of chunk k - 2, written to LDS in phase B of chunk k - 1 (a whole chunk of
    // latency cover, one raw buffer), transformed in phase A of chunk k.  Wave 8 + PW owns the pieces G0 .. G0 + NPC - 1 of the 44
    // (input 0..26, gradient 27..43); the two gradient-transform waves, whose transform needs fewer registers, carry 14 each.
    auto produce = [&](auto pw_c) {
        constexpr int PW = decltype(pw_c)::value;                  // 0, 1: dM of tiles 0..3 / 4..7;  2, 3: V
        constexpr bool DZ = PW < 2;
        constexpr int NPC = PW < 2 ? 14 : 8;
        constexpr int G0 = PW == 0 ? 0 : PW == 1 ? 14 : PW == 2 ? 28 : 36;
        static_assert(2 * 14 + 2 * 8 == W4G_VPIECES + W4G_DPIECES && W4G_VPIECES - 1 >= 14 && W4G_VPIECES < 28, "piece plan: wave 9 owns the input's last and the gradient's first piece");
        static_assert(W4G_LOOP_BYTES + 1024 <= W4G_LDS_BYTES, "the last piece's overrun");
        constexpr unsigned OOB = 0x80000000u;                      // >= num_records of both descriptors
        // the lane's unit of each piece: offset from the chunk origin (input: from the origin shifted by (-1, -1), carried by
        // the descriptor's base) and four flags -- off the image when the chunk is in the first tile row / last tile row / first
        // chunk column / last chunk column
        unsigned pre[NPC];
        unsigned long long flags = 0;
        const int ty_last = tiles_y - 1, cx_last = chunks_x - 1;
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
            constexpr int dummy = 0; (void)dummy;
            const int g = G0 + k;
            const bool isv = g < W4G_VPIECES;
            const int u = 64 * (isv ? g : g - W4G_VPIECES) + lane;
            const int rowu = isv ? W4G_VROW : W4G_DROW;
            const int rc = u / rowu, x = u - rc * rowu, px = x >> 1, row = rc >> 2;
            const int cg = (isv ? cib : cob) * 4 + (rc & 3);
            const bool ok = (u < (isv ? W4G_VUNITS : W4G_DUNITS)) & (px < (isv ? 34 : 32)) & (cg < (isv ? CGin : CGout));
            pre[k] = ok ? (unsigned)((size_t)cg * HW * 32 + ((size_t)row * W + px) * 32 + 16 * (x & 1)) : OOB;
            const int sh = isv ? -1 : 0;
            const unsigned f = (row + sh < 0 ? 1u : 0u) | (4 * ty_last + row + sh >= H ? 2u : 0u) | (px + sh < 0 ? 4u : 0u) |
                               (4 * W4G_T * cx_last + px + sh >= W ? 8u : 0u);
            flags |= (unsigned long long)f << (4 * k);
        }
        const unsigned dz_bytes = (unsigned)((size_t)n_img * CGout * HW * 32), act_bytes = (unsigned)((size_t)n_img * CGin * HW * 32);
        // the bytes in front of the input tensor are only ever addressed by units of the first tile row / chunk column of image 0,
        // which are masked
        const auto rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)act - ((size_t)W + 1) * 32), 0,
                                                            act_bytes + (unsigned)((W + 1) * 32), 0x00020000);
        const auto rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, dz_bytes, 0x00020000);
        // chunk -> (image n, tile row ty, chunk column cx) by a walker that advances `step` chunks with carries
        const int per_img = tiles_y * chunks_x;
        const int d_n = step / per_img, d_ty = (step - d_n * per_img) / chunks_x, d_cx = step - d_n * per_img - d_ty * chunks_x;
        int w_n = first / per_img, w_ty = (first - w_n * per_img) / chunks_x, w_cx = first - w_n * per_img - w_ty * chunks_x;
        auto advance = [&]() {
            w_cx += d_cx;
            if (w_cx >= chunks_x) { w_cx -= chunks_x; ++w_ty; }
            w_ty += d_ty;
            if (w_ty >= tiles_y) { w_ty -= tiles_y; ++w_n; }
            w_n += d_n;
        };
        typedef unsigned w4g_u32x4 __attribute__((ext_vector_type(4)));
        w4g_u32x4 st[NPC];                                         // the pieces of the chunk after next, on their way
        if (dbg & 64)
#pragma unroll
            for (int k = 0; k < NPC; ++k) st[k] = (w4g_u32x4)(0u);
        auto load = [&]() {                                        // the walker's chunk -> st
            if (dbg & (2 | 64)) return;
            const int cx = w_cx, ty = w_ty, n = w_n;
            const long long org = ((long long)4 * ty * W + (long long)4 * W4G_T * cx) * 32;
            const unsigned so_v = (unsigned)((long long)n * CGin * (long long)HW * 32 + org);
            const unsigned so_d = (unsigned)((long long)n * CGout * (long long)HW * 32 + org);
            const unsigned long long m4 = (ty == 0 ? 1u : 0u) | (ty == ty_last ? 2u : 0u) | (cx == 0 ? 4u : 0u) | (cx == cx_last ? 8u : 0u);
            const unsigned long long bad = flags & (m4 * 0x1111111111111111ull);          // m4 == 0: an interior chunk
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
                const unsigned off = ((bad >> (4 * k)) & 15u) ? OOB : pre[k];
                st[k] = (G0 + k < W4G_VPIECES) ? __builtin_amdgcn_raw_buffer_load_b128(rs_v, off, so_v, 0)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, so_d, 0);
            }
        };
        auto store = [&]() {                                       // st -> the raw buffer
            if (dbg & (2 | 32)) return;
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
                constexpr int dummy = 0; (void)dummy;
                const int g = G0 + k;
                const bool isv = g < W4G_VPIECES;
                const int pc = isv ? g : g - W4G_VPIECES;
                char* dst = raw0 + (isv ? 0 : 16 * W4G_VUNITS) + 1024 * pc + 16 * lane;
                // (no lane mask -- it would be a vector compare in the MFMA phase: the input region's last piece runs 48 units into
                //  the gradient region with zeros, which the same wave's next piece, the gradient region's first, overwrites in
                //  program order; the gradient region's last piece runs 512 bytes past the buffer, inside the allocation)
                *(w4g_u32x4*)dst = st[k];
            }
        };
        auto transform = [&]() {
            constexpr int ROWS = DZ ? 4 : 6;                           // 4x4 output-gradient tile / 6x6 input patch
            constexpr int RSTRIDE = 4 * (DZ ? W4G_DROW : W4G_VROW) * 16;   // bytes between image rows of a channel group in the raw buffer
            const int tl = 4 * (PW & 1) + (lane >> 4), q = lane & 15;  // tile of the chunk, channel pair of the 32-block
            const int r_off = (DZ ? 16 * W4G_VUNITS : 0) + ((q >> 2) * (DZ ? W4G_DROW : W4G_VROW) + 8 * tl) * 16 + 8 * (q & 3);
            const int w_off = (DZ ? 0 : W4G_M_FLOATS) + tl * 32 + 2 * q;
            constexpr int w_step = W4G_T * 32;
            const char* src = raw0 + r_off;
            float* dst = Tbuf + w_off;
            // every LDS read first (column by column, the order the first pass consumes them in), then the arithmetic: one
            // exposed LDS latency per chunk instead of one per batch the scheduler forms
            w4g_f32x2 rw[ROWS][ROWS];
#pragma unroll
            for (int c = 0; c < ROWS; ++c)
#pragma unroll
                for (int r = 0; r < ROWS; ++r) rw[r][c] = *(const w4g_f32x2*)(src + r * RSTRIDE + c * 32);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DZ) {     // A dY A^T: columns first (4 -> 6 rows), then each of the 6 rows (4 -> 6 columns)
                w4g_f32x2 rr[6][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    w4g_f32x2 m[6];
                    w4g_a(rw[0][c], rw[1][c], rw[2][c], rw[3][c], m);
#pragma unroll
                    for (int xi = 0; xi < 6; ++xi) rr[xi][c] = m[xi];
                }
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) {
                    w4g_f32x2 m[6];
                    w4g_a(rr[xi][0], rr[xi][1], rr[xi][2], rr[xi][3], m);
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) *(w4g_f32x2*)(dst + (xi * 6 + nu) * w_step) = m[nu];
                }
            } else {                // B^T d B
                w4g_f32x2 tt[6][6];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    w4g_f32x2 d[6], v[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) d[r] = rw[r][c];
                    w4g_bt(d, v);
#pragma unroll
                    for (int xi = 0; xi < 6; ++xi) tt[xi][c] = v[xi];
                }
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) {
                    w4g_f32x2 v[6];
                    w4g_bt(tt[xi], v);
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) *(w4g_f32x2*)(dst + (xi * 6 + nu) * w_step) = v[nu];
                }
            }
        };
        load();                                                    // chunk `first`
        store();                                                   // (waits for it)
        if (first + step < chunks) {
            advance();
            load();                                                // first + step, in flight
        }
        // bare barriers (this wave's LDS operations done, then s_barrier): __syncthreads() is a fence and would also wait for the
        // global loads in flight, which are meant to stay in flight for a whole chunk
        for (int chunk = first; chunk < chunks; chunk += step) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // X: T is free, the raw buffer holds `chunk`
            if (!(dbg & 4)) transform();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // Y: T holds `chunk`, the raw buffer is free
            if (chunk + step < chunks) {
                store();                                           // chunk + step (requested a chunk ago)
                if (chunk + 2 * step < chunks) {
                    advance();
                    load();                                        // chunk + 2 step
                }
            }
        }
        __syncthreads();                                           // Z1
        __syncthreads();                                           // Z2
    };
    if (wave == W4G_CONS) produce(std::integral_constant<int, 0>{});
    else if (wave == W4G_CONS + 1) produce(std::integral_constant<int, 1>{});
    else if (wave == W4G_CONS + 2) produce(std::integral_constant<int, 2>{});
    else produce(std::integral_constant<int, 3>{});
#endif
}

// sum over slabs in fixed order, one thread per (position, co, ci): the sums replace slab 0 in place (every thread reads
// and writes only its own element)
__device__ __forceinline__ void w4g_sum_body(float* __restrict__ slabs, int nslab, size_t stride, size_t idx) {
    if (idx >= stride) return;
    float* pp = slabs + idx;
    // four interleaved partial sums (slab k -> accumulator k % 4), then ((s0+s1)+s2)+s3, as wgrad_reduce_kernel
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nslab; k += 4) {
        s0 += pp[(size_t)k * stride];
        s1 += pp[(size_t)(k + 1) * stride];
        s2 += pp[(size_t)(k + 2) * stride];
        s3 += pp[(size_t)(k + 3) * stride];
    }
    for (; k < nslab; ++k) s0 += pp[(size_t)k * stride];
    pp[0] = ((s0 + s1) + s2) + s3;
}

// dW[co][ci][ky][kx] (OIHW, real channel counts) = G^T S G from the summed slab, in double
__device__ __forceinline__ void w4g_finish_body(const float* __restrict__ sums, float* __restrict__ dW, int Cin_real, int Cout_real, int coP,
                                                int ciP, int idx) {
    if (idx >= Cout_real * Cin_real) return;
    const int ci = idx % Cin_real, co = idx / Cin_real;                   // ci fastest: coalesced reads
    const size_t pstride = (size_t)coP * ciP;
    const float* base = sums + (size_t)co * ciP + ci;
    const double G[6][3] = {{0.25, 0, 0},
                            {-1.0 / 6, -1.0 / 6, -1.0 / 6},
                            {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6},
                            {1.0 / 24, -1.0 / 12, 1.0 / 6},
                            {0, 0, 1}};
    double t[6][3];                                                       // S G
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
        double u[6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) u[nu] = (double)base[(xi * 6 + nu) * pstride];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            double s = 0.0;
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) s += G[nu][kx] * u[nu];
            t[xi][kx] = s;
        }
    }
    float* out = dW + ((size_t)co * Cin_real + ci) * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            double s = 0.0;
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) s += G[xi][ky] * t[xi][kx];
            out[ky * 3 + kx] = (float)s;
        }
}

__global__ void __launch_bounds__(256) wgrad_wino4_sum_kernel(float* __restrict__ slabs, int nslab, size_t stride) {
    w4g_sum_body(slabs, nslab, stride, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256)
wgrad_wino4_finish_kernel(const float* __restrict__ sums, float* __restrict__ dW, int Cin_real, int Cout_real, int coP, int ciP) {
    w4g_finish_body(sums, dW, Cin_real, Cout_real, coP, ciP, blockIdx.x * blockDim.x + threadIdx.x);
}
#ifndef SCIPNP_DIAG_BUILD
// the slab reductions / back-transforms of MANY layers in one launch each (blockIdx.y = the layer): a trainer that keeps every
// layer's slabs until the end of its backward pass finishes them all with two launches instead of two per layer
constexpr int W4G_MULTI_MAX = 32;
struct W4gFinishJobs {
    float* ws[W4G_MULTI_MAX];
    float* dW[W4G_MULTI_MAX];
    int nslab[W4G_MULTI_MAX], cin_real[W4G_MULTI_MAX], cout_real[W4G_MULTI_MAX], coP[W4G_MULTI_MAX], ciP[W4G_MULTI_MAX];
};
__global__ void __launch_bounds__(256) wgrad_wino4_sum_multi_kernel(const W4gFinishJobs j) {
    const int q = blockIdx.y;
    w4g_sum_body(j.ws[q], j.nslab[q], (size_t)W4G_POS * j.coP[q] * j.ciP[q], (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) wgrad_wino4_finish_multi_kernel(const W4gFinishJobs j) {
    const int q = blockIdx.y;
    w4g_finish_body(j.ws[q], j.dW[q], j.cin_real[q], j.cout_real[q], j.coP[q], j.ciP[q], blockIdx.x * blockDim.x + threadIdx.x);
}
#endif

static inline int w4g_round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace scipnp

using namespace scipnp;

static int w4g_launch(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n, int Cin_real,
                      int Cout_real, int Cin, int Cout, int h, int w, int dbg, scipnp_stream_t s) {
    SCIPNP_REQUIRE(act_c8 && dz_c8 && workspace, "null pointer");      // (dW == NULL: the slabs only, see ..._finish_multi)
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin_real > 0 &&
                   Cout_real > 0 && Cin_real <= Cin && Cout_real <= Cout && nslab > 0, "bad shape");
    // (the input descriptor spans the tensor plus (w + 1) pixels: masked lanes carry the offset 0x80000000, which must stay
    // past its range)
    SCIPNP_REQUIRE((long long)n * Cin * h * w * 4 + (long long)(w + 1) * 32 <= (1ll << 31) && (long long)n * Cout * h * w * 4 < (1ll << 31),
                   "tensors of 2 GiB or more: use scipnp_conv3x3_wgrad");
    SCIPNP_ALIGNED(act_c8); SCIPNP_ALIGNED(dz_c8);
    const int ciP = w4g_round_up(Cin, 32), coP = w4g_round_up(Cout, 32);
    const int ncib = ciP / 32, ncob = coP / 32;
    SCIPNP_REQUIRE((long long)nslab * ncib * ncob <= 1 << 20, "too many workgroups");
    hipStream_t st = (hipStream_t)s;
    static LdsAttrOnce attr;
    if (int rc_ = attr.ensure((const void*)conv3x3_wgrad_wino4_kernel, W4G_LDS_BYTES, "wgrad_wino4")) return rc_;
    hipLaunchKernelGGL(conv3x3_wgrad_wino4_kernel, dim3((unsigned)(nslab * ncob * ncib)), dim3(W4G_THREADS), W4G_LDS_BYTES, st,
                       act_c8, dz_c8, workspace, n, Cin / 8, Cout / 8, h, w, nslab, ncob, ncib, dbg);
    int rc = launch_status("conv3x3_wgrad_wino4_kernel");
    if (rc || dW == nullptr) return rc;
    const size_t per_slab = (size_t)W4G_POS * coP * ciP;
    hipLaunchKernelGGL(wgrad_wino4_sum_kernel, dim3((unsigned)((per_slab + 255) / 256)), dim3(256), 0, st, workspace, nslab,
                       per_slab);
    const int total = Cout_real * Cin_real;
    hipLaunchKernelGGL(wgrad_wino4_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, dW, Cin_real, Cout_real,
                       coP, ciP);
    return launch_status("wgrad_wino4_sum / finish kernels");
}

extern "C" {

#ifndef SCIPNP_DIAG_BUILD   /* ---- product entries (libscipnp.so) */

size_t scipnp_conv3x3_wgrad_wino4_workspace_floats(int Cin, int Cout, int nslab) {
    if (Cin <= 0 || Cout <= 0 || nslab <= 0) return 0;
    return (size_t)nslab * W4G_POS * w4g_round_up(Cout, 32) * w4g_round_up(Cin, 32);
}

int scipnp_conv3x3_wgrad_wino4(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                               int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, scipnp_stream_t s) {
    return w4g_launch(act_c8, dz_c8, dW, workspace, nslab, n, Cin_real, Cout_real, Cin, Cout, h, w, 0, s);
}

int scipnp_conv3x3_wgrad_wino4_finish_multi(int n, float* const* workspace, float* const* dW, const int* nslab, const int* Cin_real,
                                            const int* Cout_real, const int* Cin, const int* Cout, scipnp_stream_t s) {
    SCIPNP_REQUIRE(n >= 0 && (n == 0 || (workspace && dW && nslab && Cin_real && Cout_real && Cin && Cout)), "bad arguments");
    for (int base = 0; base < n; base += W4G_MULTI_MAX) {
        const int m = n - base < W4G_MULTI_MAX ? n - base : W4G_MULTI_MAX;
        W4gFinishJobs j = {};
        size_t most_sum = 0;
        int most_fin = 0;
        for (int q = 0; q < m; ++q) {
            const int g = base + q;
            SCIPNP_REQUIRE(workspace[g] && dW[g] && nslab[g] > 0 && Cin[g] > 0 && Cout[g] > 0 && Cin[g] % 8 == 0 && Cout[g] % 8 == 0 &&
                           Cin_real[g] > 0 && Cout_real[g] > 0 && Cin_real[g] <= Cin[g] && Cout_real[g] <= Cout[g], "bad arguments in job %d", g);
            j.ws[q] = workspace[g]; j.dW[q] = dW[g]; j.nslab[q] = nslab[g]; j.cin_real[q] = Cin_real[g]; j.cout_real[q] = Cout_real[g];
            j.ciP[q] = w4g_round_up(Cin[g], 32); j.coP[q] = w4g_round_up(Cout[g], 32);
            const size_t per_slab = (size_t)W4G_POS * j.coP[q] * j.ciP[q];
            most_sum = per_slab > most_sum ? per_slab : most_sum;
            most_fin = Cout_real[g] * Cin_real[g] > most_fin ? Cout_real[g] * Cin_real[g] : most_fin;
        }
        hipLaunchKernelGGL(wgrad_wino4_sum_multi_kernel, dim3((unsigned)((most_sum + 255) / 256), (unsigned)m), dim3(256), 0,
                           (hipStream_t)s, j);
        hipLaunchKernelGGL(wgrad_wino4_finish_multi_kernel, dim3((unsigned)((most_fin + 255) / 256), (unsigned)m), dim3(256), 0,
                           (hipStream_t)s, j);
    }
    return launch_status("wgrad_wino4 sum / finish multi kernels");
}

#else   /* ---- SCIPNP_DIAG_BUILD: the laboratory entry (libscipnp_diag.so, include/scipnp_diag.h) */

/* the product kernel with its timing-only ablation switches (see the kernel's comment); workspace as the product entry's */
int scipnp_diag_conv3x3_wgrad_wino4(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                                    int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, int dbg, scipnp_stream_t s) {
    return w4g_launch(act_c8, dz_c8, dW, workspace, nslab, n, Cin_real, Cout_real, Cin, Cout, h, w, dbg, s);
}

#endif

}  // extern "C"
