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
#include "common.hpp"
#include <type_traits>

namespace scipnp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int W4_SLAB = 2 * 9 * 64 * 4;               // floats per k-step slab (18432 B)
constexpr int W4_PIECES = W4_SLAB / 256;              // 1 KiB LDS-DMA pieces per slab (18)
constexpr int W4_TW = 64, W4_TH = 8;                  // output pixels per workgroup
constexpr int W4_TWP = W4_TW + 2, W4_THP = W4_TH + 2; // halo tile
constexpr int W4_THREADS = 256;
constexpr int W4_RSL = W4_TWP + W4_TWP / 16;          // slots per halo row: pixel x sits in slot x + (x >> 4) (one padding slot per 16)
constexpr int W4_UNITS = W4_THP * W4_RSL * 2;         // 16-byte units (4 channels of a pixel) of the halo tile: [hf][row][slot]
constexpr int W4_RAW_PIECES = (W4_UNITS + 63) / 64;   // 1 KiB LDS-DMA pieces per raw tile (21; the last one partly padding)
constexpr int W4_RAW = W4_RAW_PIECES * 256;           // floats per raw buffer
constexpr int W4_IN_ITERS = (W4_RAW_PIECES + 3) / 4;  // raw pieces per wave and group (waves 1..3 fetch their fifth piece twice)
constexpr int W4_DMA_ITERS = (W4_PIECES + 3) / 4;     // U pieces per wave and slab (waves 2, 3 fetch their fourth piece twice)
constexpr size_t W4_LDS_BYTES = (2 * (size_t)W4_RAW + 2 * (size_t)W4_SLAB) * sizeof(float);
static_assert(W4_LDS_BYTES >= 4 * 16 * 64 * 16, "the epilogue's exchange buffer lives in the loop's LDS");
static_assert(2 * W4_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

struct Wino4Args {
    const float* in;
    const float* wpk;        // [2*CGin k-steps][CoutP/32][4608] + bias[CoutP]
    float* out;
    const float* residual;
    const float* mask_src;
    int CGin, CGout, NCB;    // NCB = CoutP / 32
    int H, W;
    int ntx, nty;
    int flags;
};

// (host pass: only parsed -- the kernel body never runs there)
__host__ __device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return a * b + c;
#endif
}
__host__ __device__ __forceinline__ f32x4 pk_fma(f32x4 a, f32x4 b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(a, b, c);
#else
    return a * b + c;
#endif
}
// a - b on a float2 as ONE v_pk_add_f32 (see conv_wino.hip: the compiler selects two v_sub_f32)
__host__ __device__ __forceinline__ f32x2 psub4(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - b;
#endif
}
// a - b on a float4 as two v_pk_add_f32
__host__ __device__ __forceinline__ f32x4 psub4(f32x4 a, f32x4 b) {
    const f32x2 lo = psub4(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = psub4(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

template <typename T>
__host__ __device__ __forceinline__ T splat(float v);
template <>
__host__ __device__ __forceinline__ f32x2 splat<f32x2>(float v) { return f32x2{v, v}; }
template <>
__host__ __device__ __forceinline__ f32x4 splat<f32x4>(float v) { return f32x4{v, v, v, v}; }

// rows 0..2 of B^T x for x = (x0 .. x4)  (x5 does not enter):  4x0 - 5x2 + x4 | -4x1 - 4x2 + x3 + x4 | 4x1 - 4x2 - x3 + x4
__host__ __device__ __forceinline__ void bt_lo(const f32x2 x0, const f32x2 x1, const f32x2 x2, const f32x2 x3, const f32x2 x4, f32x2& o0,
                                      f32x2& o1, f32x2& o2) {
    const f32x2 c4 = splat<f32x2>(4.f), m4 = splat<f32x2>(-4.f), m5 = splat<f32x2>(-5.f);
    o0 = pk_fma(c4, x0, pk_fma(m5, x2, x4));
    const f32x2 a = pk_fma(m4, x2, x4), b = pk_fma(m4, x1, x3);
    o1 = a + b;
    o2 = psub4(a, b);
}
// rows 3..5 of B^T x for x = (x1 .. x5)  (x0 does not enter):  -2x1 - x2 + 2x3 + x4 | 2x1 - x2 - 2x3 + x4 | 4x1 - 5x3 + x5
__host__ __device__ __forceinline__ void bt_hi(const f32x2 x1, const f32x2 x2, const f32x2 x3, const f32x2 x4, const f32x2 x5, f32x2& o3,
                                      f32x2& o4, f32x2& o5) {
    const f32x2 c4 = splat<f32x2>(4.f), c2 = splat<f32x2>(2.f), m2 = splat<f32x2>(-2.f), m5 = splat<f32x2>(-5.f);
    const f32x2 c = psub4(x4, x2), e = psub4(x3, x1);
    o3 = pk_fma(c2, e, c);
    o4 = pk_fma(m2, e, c);
    o5 = pk_fma(c4, x1, pk_fma(m5, x3, x5));
}

// TAG only changes the symbol name (1 = first / last layer of a network) so profiler statistics of the body layers stay clean
// DIAG (timing experiments only, wrong results): bit0 no transform, 1 no raw staging, 2 no U DMA, 3 no barriers, 4 no MFMAs, 5 no epilogue
template <int TAG, int DIAG = 0>
__global__ void __launch_bounds__(W4_THREADS, 2)
conv3x3_c8w4_kernel(const Wino4Args a) {

    extern __shared__ __attribute__((aligned(16))) float smem_w4[];
    float* const raw_lds = smem_w4;                    // [2][RAW]
    float* const u_lds = smem_w4 + 2 * W4_RAW;         // [2][SLAB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wvu >> 1, xh = wvu & 1;             // tile row of the workgroup, half of the transformed rows
    const int tn = lane & 15, q = lane >> 4;           // tile along x, channel pair
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    // XCD-aware order (as conv_wino.hip): one XCD works through a contiguous run of (tile, co-block) pairs, the co-blocks of
    // a tile adjacent, so the input tile is fetched from HBM once
    unsigned lin = blockIdx.x;
    {
        const unsigned total = gridDim.x;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    }
    const int split = lin % a.NCB;
    unsigned t = lin / a.NCB;
    const int bx = t % a.ntx;
    t /= a.ntx;
    const int by = t % a.nty, n = t / a.nty;
    const int x0 = bx * W4_TW, y0 = by * W4_TH;

    // ---- staging plan of the raw tile: LDS unit u = 64 * piece + lane = (hf * THP + r) * TWP + slot -> pixel (r, c) of the halo
    // tile, channels 4hf .. 4hf+3; wave w fetches the pieces w, w + 4, ...  Pixels outside the image (and the padding units of the
    // last piece) get an offset past the buffer descriptor's range and arrive as zeros.
    unsigned in_off[W4_IN_ITERS];
#pragma unroll
    for (int k = 0; k < W4_IN_ITERS; ++k) {
        int pc = wvu + 4 * k;
        if (pc >= W4_RAW_PIECES) pc -= 4;
        const int u = pc * 64 + lane;
        const int hf = u >= W4_UNITS / 2 ? 1 : 0;
        const int v = u - hf * (W4_UNITS / 2);
        const int r = v / W4_RSL, sl = v - r * W4_RSL;
        const int c = 16 * (sl / 17) + sl % 17;                                 // (sl % 17 == 16: a padding slot)
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        in_off[k] = (u < W4_UNITS && sl % 17 != 16 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)((gy * W + gx) * 32 + 16 * hf) : 0xFFFFFF00u;
    }
    const float* in_g = a.in + (size_t)n * a.CGin * HW * 8;                     // advanced by HW*8 per group
    const float* w_g = a.wpk + (size_t)split * W4_SLAB;                         // advanced by NCB*SLAB per k-step
    const size_t w_step = (size_t)a.NCB * W4_SLAB;
    const unsigned plane_bytes = (unsigned)(HW * 32);
    (void)w_g; (void)plane_bytes; (void)in_g; (void)wvu;

    // `last`: no further group / k-step exists -- the pointer stays and the same data is fetched again (unused), so that
    // every step issues the same number of memory operations and the vmcnt waits below are constants
    auto issue_raw = [&](float* dst, bool last) {
        if (DIAG & 2) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < W4_IN_ITERS; ++k) {
            int pc = wvu + 4 * k;
            if (pc >= W4_RAW_PIECES) pc -= 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_in, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16,
                                                     in_off[k], 0, 0, 0);
        }
#endif
        if (!last) in_g += HW * 8;
    };
    auto issue_u = [&](float* dst, bool last) {
        if (DIAG & 4) return;
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, W4_SLAB * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < W4_DMA_ITERS; ++k) {
            int pc = wvu + 4 * k;
            if (pc >= W4_PIECES) pc -= 4;                                       // (waves 2, 3: their fifth piece is their fourth again)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)((char*)dst + 1024 * pc), 16,
                                                     (unsigned)(1024 * pc + 16 * lane), 0, 0, 0);
        }
#endif
        if (!last) w_g += w_step;
    };

    f32x4 acc[3][6][2];                                 // [own row xi - 3 xh][nu][co half]
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[x][nu][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS offsets (floats): patch of tile (tg, tn), channel pair q (half-pixel plane q >> 1, 8 bytes (q & 1) of the
    // unit); U vectors of half xh
    const int b_row = (((q >> 1) * W4_THP + 4 * tg + xh) * W4_RSL) * 4 + (q & 1) * 2;        // wave xh reads the patch rows xh .. xh + 4
    const int b_off0 = b_row + (4 * tn + (tn >> 2)) * 4, b_off1 = b_row + (4 * tn + ((tn + 1) >> 2)) * 4;   // columns 0..3 | 4, 5
    const int a_off = xh * (9 * 256) + lane * 4;                                             // + v * 256

    const int CG = a.CGin;
    f32x2 V[3][6];                                      // own three rows of B^T d B, x the channel pair
    if constexpr (DIAG != 0) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) V[x][nu] = f32x2{(float)lane, 1.f};
    }
    // B^T d B, own rows: column pass one patch column at a time (five 8-byte reads), then the row pass in place
    auto transform_half = [&](const float* rawp, auto LO) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int bo = (c < 4 ? b_off0 : b_off1) + c * 4;
            f32x2 x[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) x[r] = *(const f32x2*)(rawp + bo + r * (W4_RSL * 4));
            if constexpr (decltype(LO)::value) bt_lo(x[0], x[1], x[2], x[3], x[4], V[0][c], V[1][c], V[2][c]);
            else bt_hi(x[0], x[1], x[2], x[3], x[4], V[0][c], V[1][c], V[2][c]);
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            f32x2 o[6];
            bt_lo(V[x][0], V[x][1], V[x][2], V[x][3], V[x][4], o[0], o[1], o[2]);
            bt_hi(V[x][1], V[x][2], V[x][3], V[x][4], V[x][5], o[3], o[4], o[5]);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) V[x][nu] = o[nu];
        }
    };
    auto transform = [&](const float* rawp) {
        if (DIAG & 1) return;
        if (xh == 0) transform_half(rawp, std::true_type{});
        else transform_half(rawp, std::false_type{});
    };
    // the 36 MFMAs of one k-step: nine U vectors, each the four fragments (nu = 2np, 2np+1) x (half 0, 1) of one own row
    auto mfma_step = [&](const float* ucur, int j) {
        if (DIAG & 16) return;
        f32x4 af[3];
        af[0] = *(const f32x4*)(ucur + a_off);
        af[1] = *(const f32x4*)(ucur + a_off + 256);
#pragma unroll
        for (int v = 0; v < 9; ++v) {
            if (v + 2 < 9) af[(v + 2) % 3] = *(const f32x4*)(ucur + a_off + (v + 2) * 256);
            const f32x4 u = af[v % 3];
            const int x = v / 3, np = v % 3;
            const float b0 = j ? V[x][2 * np][1] : V[x][2 * np][0], b1 = j ? V[x][2 * np + 1][1] : V[x][2 * np + 1][0];
            acc[x][2 * np][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[0], b0, acc[x][2 * np][0], 0, 0, 0);
            acc[x][2 * np + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[1], b1, acc[x][2 * np + 1][0], 0, 0, 0);
            acc[x][2 * np][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[2], b0, acc[x][2 * np][1], 0, 0, 0);
            acc[x][2 * np + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[3], b1, acc[x][2 * np + 1][1], 0, 0, 0);
        }
    };

    {   // prologue: raw tile of group 0, U of k-step 0
        issue_raw(raw_lds, CG <= 1);
        issue_u(u_lds, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int g = 0; g < CG; ++g) {
        float* const rcur = raw_lds + (g & 1) * W4_RAW;
        float* const rnext = raw_lds + ((g & 1) ^ 1) * W4_RAW;
        // ---- k-step (g, 0).  Behind the last barrier every wave has finished k-step 2g-1 and the transform of group g-1:
        // U of k-step 2g+1 -> its buffer, raw tile of group g+1 -> the other tile buffer (two k-steps ahead of its use)
        issue_u(u_lds + W4_SLAB, 2 * g + 2 >= 2 * CG);
        issue_raw(rnext, g + 2 >= CG);
        transform(rcur);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        mfma_step(u_lds, 0);
        // bare s_barrier: __syncthreads() is a fence too and would wait for the raw tile's LDS-DMA issued above.  What must have
        // landed is U of k-step 2g+1 (every wave's own pieces; the raw pieces were issued behind them); own LDS reads are done.
        if (DIAG & 8) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W4_IN_ITERS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(W4_IN_ITERS) : "memory");
        // ---- k-step (g, 1): U of k-step 2g+2 -> the buffer of k-step 2g
        issue_u(u_lds, 2 * g + 3 >= 2 * CG);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        mfma_step(u_lds + W4_SLAB, 1);
        if (DIAG & 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // raw tile of group g+1, U of k-step 2g+2
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the re-fetched last slab must not land in the exchange buffer)
    __syncthreads();

    // ---- output transform.  Own rows xi: R[xi][j] = sum_nu M[xi][nu] A^T[j][nu], then the partial tile P[i][j] = sum_xi A^T[i][xi] R[xi][j];
    // wave xh finishes the output rows 2xh, 2xh+1 and hands the other two to its partner through LDS.
    // lane: tile (tg, tn), channels 32*split + 16*h + 4*q + r.
    if (DIAG & 32) return;
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
    const float* bias = a.wpk + (size_t)2 * a.CGin * w_step;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
    (void)relu; (void)add_res; (void)mask; (void)bias;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cog0 = split * 4 + h * 2;                    // this lane's group: cog0 + (q >> 1)
        if (cog0 >= a.CGout) continue;                         // wave-uniform
        const bool lane_ok = cog0 + (q >> 1) < a.CGout;
        const f32x4 bs = *(const f32x4*)(bias + (cog0 + (q >> 1)) * 8 + 4 * (q & 1));          // bias holds CoutP entries
        f32x4 v[2][4];
        unsigned off[2][4];
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = y0 + 4 * tg + 2 * xh + il, x = x0 + 4 * tn + j;
                const f32x4 other = *(const f32x4*)(xbuf + (((wvu ^ 1) * 16 + (il * 4 + j) * 2 + h) * 64 + lane) * 4);
                v[il][j] = (keep[il][j][h] + other) + bs;
                off[il][j] = (lane_ok && y < H && x < W)
                                 ? (unsigned)((y * W + x) * 32 + 16 * (q & 1)) + (unsigned)(q >> 1) * plane_bytes
                                 : 0x80000000u;
            }
#if defined(__HIP_DEVICE_COMPILE__)
        const size_t half0 = ((size_t)n * a.CGout + cog0) * HW * 8;                            // floats
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
        if (mask) {   // ReLU backward: pass the gradient where the forward activation was > 0
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
}

// U = G g G^T from the fp32 direct packing [cig][tap][CoutP][8]; one thread per slab element
__global__ void pack_wino4_kernel(const float* __restrict__ pk, float* __restrict__ out, int CGin, int CoutP) {
    const size_t NCB = CoutP / 32;
    const size_t total = (size_t)2 * CGin * NCB * W4_SLAB;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        // slab element index: [xh 2][v 9][lane 64 = q*16 + tn][e 4] -> U_p[co = 32 cb + 16 (e >> 1) + tn][ci = 2 q + j],
        // p = (xi = 3 xh + v / 3, nu = 2 (v % 3) + (e & 1)); slab index = (2 cig + j) * NCB + cb
        const int el = (int)(i % W4_SLAB);
        const size_t sl = i / W4_SLAB;
        const int cb = (int)(sl % NCB), ks = (int)(sl / NCB);
        const int cig = ks >> 1, j = ks & 1;
        const int e = el & 3, tnl = (el >> 2) & 15, ql = (el >> 6) & 3, vv = (el >> 8) % 9, xhh = (el >> 8) / 9;
        const int xi = 3 * xhh + vv / 3, nu = 2 * (vv % 3) + (e & 1), h = e >> 1;
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

static inline int round_up_w4(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace scipnp

using namespace scipnp;

extern "C" {

size_t scipnp_conv3x3_wino4_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    const int CoutP = round_up_w4(Cout, 32);
    return (size_t)2 * (Cin / 8) * (CoutP / 32) * W4_SLAB + CoutP;
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
    SCIPNP_REQUIRE(!(flags & (4 | 8 | 0x200)), "the F(4x4,3x3) kernel is stride 1, plain store, 8-row workgroups");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    Wino4Args a;
    a.in = in; a.wpk = packed_wino4; a.out = out; a.residual = residual; a.mask_src = mask_src;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w4(Cout, 32) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W4_TH - 1) / W4_TH;
    a.flags = flags;
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    const int tag = (flags & 0x100) ? 1 : 0;
    const void* fns[2] = {(const void*)conv3x3_c8w4_kernel<0>, (const void*)conv3x3_c8w4_kernel<1>};
    static LdsAttrOnce attr[2];
    if (int rc = attr[tag].ensure(fns[tag], W4_LDS_BYTES, "conv3x3_c8w4")) return rc;
    const dim3 grid((unsigned)total), block(W4_THREADS);
    if (tag) hipLaunchKernelGGL((conv3x3_c8w4_kernel<1>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);
    else hipLaunchKernelGGL((conv3x3_c8w4_kernel<0>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);
    return launch_status("conv3x3_c8w4_kernel");
}

/* diagnostic: the same kernel with parts switched off (timing only, WRONG results) -- tools/probes/wino4_ablate.py.
 * diag: bit0 no input transform, bit1 no raw-tile staging, bit2 no U LDS-DMA, bit3 no barriers in the K loop, bit4 no MFMAs,
 * bit5 no output transform / stores */
int scipnp_conv3x3_c8w4_diag(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                             int flags, int diag, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_wino4 && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad shape");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    Wino4Args a;
    a.in = in; a.wpk = packed_wino4; a.out = out; a.residual = nullptr; a.mask_src = nullptr;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w4(Cout, 32) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + W4_TW - 1) / W4_TW; a.nty = (h + W4_TH - 1) / W4_TH;
    a.flags = flags & 1;
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    const dim3 grid((unsigned)total), block(W4_THREADS);
#define W4_DIAG_CASE(D)                                                                                                    \
    case D: {                                                                                                              \
        static LdsAttrOnce attr;                                                                                           \
        if (int rc = attr.ensure((const void*)conv3x3_c8w4_kernel<0, D>, W4_LDS_BYTES, "conv3x3_c8w4 diag")) return rc;   \
        hipLaunchKernelGGL((conv3x3_c8w4_kernel<0, D>), grid, block, W4_LDS_BYTES, (hipStream_t)s, a);                     \
        break;                                                                                                             \
    }
    switch (diag) {
        W4_DIAG_CASE(1) W4_DIAG_CASE(2) W4_DIAG_CASE(4) W4_DIAG_CASE(8) W4_DIAG_CASE(16) W4_DIAG_CASE(32) W4_DIAG_CASE(6)
        W4_DIAG_CASE(7) W4_DIAG_CASE(15) W4_DIAG_CASE(39) W4_DIAG_CASE(47) W4_DIAG_CASE(48) W4_DIAG_CASE(49) W4_DIAG_CASE(55)
        W4_DIAG_CASE(63) W4_DIAG_CASE(3) W4_DIAG_CASE(5) W4_DIAG_CASE(9) W4_DIAG_CASE(10) W4_DIAG_CASE(12) W4_DIAG_CASE(14)
        default: SCIPNP_REQUIRE(false, "diag mask %d has no instantiation", diag);
    }
#undef W4_DIAG_CASE
    return launch_status("conv3x3_c8w4_kernel<diag>");
}

}  // extern "C"
