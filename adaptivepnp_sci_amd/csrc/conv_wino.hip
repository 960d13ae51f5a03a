// 3x3 / stride-1 / pad-1 convolution in fp32 arithmetic as Winograd F(2x2, 3x3) on the CDNA4 matrix cores.
//
//   Y(2x2) = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 products per output tile and channel pair instead of 36: 2.25x fewer multiply-adds than the direct form of
// conv.hip, every one of them an exact fp32 product accumulated in fp32 on v_mfma_f32_16x16x4_f32 (the fp32 MFMA runs
// at the fp32 VECTOR rate on gfx950, so the input / output transforms -- O(C) adds per tile against O(C^2) products --
// cost a few per cent; this is the same trade the reference's cuDNN makes for fp32 3x3 convolutions).  The filter
// transform U = G g G^T is done once per weight update, in double, by scipnp_pack_conv3x3_wino.
//
// GEMM view, per Winograd position p = (xi, nu) of 16:   M_p[co][tile] = sum_ci U_p[co][ci] * V_p[ci][tile]
//   A = U_p   16 (co) x 4 (ci)     lane l holds A[l & 15][l >> 4]
//   B = V_p    4 (ci) x 16 (tile)  lane l holds B[l >> 4][l & 15]
//   D         16 (co) x 16 (tile)  lane l holds D[4*(l >> 4) + r][l & 15], r = 0..3
// A lane therefore owns ONE tile and, per 8-channel group, the channel pair {2q, 2q+1}, q = l >> 4: it reads its own
// 4x4 patch of those two channels (16 ds_read_b64), transforms it in registers (32 adds per channel) and feeds the 16
// positions x 2 output-channel halves x 2 k-steps = 64 MFMAs of the group; the 32 accumulators (128 VGPRs) hold all 16
// positions of 32 output channels x 16 tiles, so the output transform happens once, after the K loop.
//
// Work decomposition: workgroup = 8 waves = 16 rows x 32 columns of output pixels x 32 output channels; wave w owns the
// tile row w (output rows 2w, 2w+1; 16 tiles along x).  Output-channel blocks of 32 are separate workgroups placed next
// to each other on ONE XCD (its L2 serves the re-read of the input tile).  K loop over input channel groups of 8; per
// group, double-buffered in LDS:
//   raw input halo tile (18 x 34 pixels x 8 channels) -> four channel-pair planes [q][18][34][2] (+2 floats between
//     planes: the 32 lanes of a ds_read_b64 group then touch 64 distinct banks), global -> registers -> ds_write_b64;
//   U slab (16 positions x 32 co x 8 ci = 16 KiB, laid out [p][co/16][ci/4][co%16][ci%4] by the packer so that the
//     fragment reads are conflict-free) by LDS-DMA (buffer_load_dwordx4 ... lds), no staging registers.
#include "common.hpp"

namespace scipnp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WN_TW = 32, WN_TH = 16;                 // output pixels per workgroup (columns, rows)
constexpr int WN_THREADS = 512;
constexpr int WN_TWP = WN_TW + 2, WN_THP = WN_TH + 2; // input halo tile
constexpr int WN_PLANE = WN_THP * WN_TWP * 2 + 2;     // floats per channel-pair plane (+2: bank spread)
constexpr int WN_RAW = 4 * WN_PLANE;                  // 4904 floats = 19616 B (16-B multiple)
constexpr int WN_SLAB = 16 * 32 * 8;                  // 4096 floats = 16 KiB
constexpr int WN_STAGE = WN_RAW + WN_SLAB;
constexpr size_t WN_LDS_BYTES = 2 * (size_t)WN_STAGE * sizeof(float);
constexpr int WN_UNITS = WN_THP * WN_TWP * 2;         // 16-byte half pixels of the halo tile
constexpr int WN_IN_ITERS = (WN_UNITS + WN_THREADS - 1) / WN_THREADS;

struct WinoArgs {
    const float* in;
    const float* wpk;        // [CGin][CoutP/32][4096] + bias[CoutP]
    float* out;
    const float* residual;
    const float* mask_src;
    int CGin, CGout, NCB;    // NCB = CoutP / 32
    int H, W;
    int ntx, nty;
    int flags;
};

// TAG only changes the symbol name (1 = network head layer) so profiler statistics of the body layers stay clean.
template <int TAG>
__global__ void __launch_bounds__(WN_THREADS, 2)
conv3x3_c8w_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int tn = lane & 15, q = lane >> 4;           // tile along x, channel pair
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; one XCD works through a contiguous run of
    // (tile, co-block) pairs with the co-blocks of a tile adjacent, so the input tile is fetched from HBM once
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
    const int x0 = bx * WN_TW, y0 = by * WN_TH;

    // ---- staging plan of the raw tile: unit e = tid + 512k -> pixel (r, c) of the halo tile, 16-byte half `hf`
    int in_off[WN_IN_ITERS], lds_off[WN_IN_ITERS];
#pragma unroll
    for (int k = 0; k < WN_IN_ITERS; ++k) {
        const int e = tid + k * WN_THREADS;
        in_off[k] = -1;
        lds_off[k] = -1;
        if (e < WN_UNITS) {
            const int pix = e >> 1, hf = e & 1;
            const int r = pix / WN_TWP, c = pix - r * WN_TWP;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            lds_off[k] = (2 * hf) * WN_PLANE + pix * 2;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) in_off[k] = (gy * W + gx) * 8 + 4 * hf;
        }
    }
    const float* in_g = a.in + (size_t)n * a.CGin * HW * 8;                  // advanced by HW*8 per group
    const float* w_g = a.wpk + (size_t)split * WN_SLAB;                      // advanced by NCB*4096 per group
    const size_t w_step = (size_t)a.NCB * WN_SLAB;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    (void)wvu; (void)w_g;

    f32x4 st_in[WN_IN_ITERS];
    auto issue_loads = [&](float* stage) {
#pragma unroll
        for (int k = 0; k < WN_IN_ITERS; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in_off[k] >= 0) v = *(const f32x4*)(in_g + in_off[k]);
            st_in[k] = v;
        }
        in_g += HW * 8;
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins: keep them out of the host pass that only emits the launch stub
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, WN_SLAB * 4, 0x00020000);
        char* ub = (char*)(stage + WN_RAW);
#pragma unroll
        for (int k = 0; k < 2; ++k)        // 1024 16-byte units / 512 threads
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_w, (__attribute__((address_space(3))) void*)(ub + 16 * (wvu * 64 + k * WN_THREADS)), 16,
                (unsigned)(16 * (tid + k * WN_THREADS)), 0, 0, 0);
#endif
        w_g += w_step;
    };
    auto write_lds = [&](float* stage) {
#pragma unroll
        for (int k = 0; k < WN_IN_ITERS; ++k)
            if (lds_off[k] >= 0) {
                *(f32x2*)(stage + lds_off[k]) = f32x2{st_in[k][0], st_in[k][1]};
                *(f32x2*)(stage + lds_off[k] + WN_PLANE) = f32x2{st_in[k][2], st_in[k][3]};
            }
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_loads(smem_w);
    write_lds(smem_w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-lane LDS offsets (floats)
    const int b_off = q * WN_PLANE + ((2 * wv) * WN_TWP + 2 * tn) * 2;              // + (dy*TWP + dx)*2
    const int a_off = WN_RAW + (((q >> 1) * 16 + tn) * 4) + 2 * (q & 1);            // + ((p*2 + h)*2)*64

    for (int cig = 0; cig < a.CGin; ++cig) {
        float* buf = smem_w + (cig & 1) * WN_STAGE;
        float* nxt = smem_w + ((cig + 1) & 1) * WN_STAGE;
        const bool more = (cig + 1 < a.CGin);
        if (more) issue_loads(nxt);
        // ---- input transform  V = B^T d B  of this lane's patch, two channels at once
        f32x2 d[4][4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) d[dy][dx] = *(const f32x2*)(buf + b_off + (dy * WN_TWP + dx) * 2);
        f32x2 V[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x2 t0 = d[0][c] - d[2][c], t1 = d[1][c] + d[2][c], t2 = d[2][c] - d[1][c], t3 = d[1][c] - d[3][c];
            d[0][c] = t0; d[1][c] = t1; d[2][c] = t2; d[3][c] = t3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            V[r][0] = d[r][0] - d[r][2];
            V[r][1] = d[r][1] + d[r][2];
            V[r][2] = d[r][2] - d[r][1];
            V[r][3] = d[r][1] - d[r][3];
        }
        // ---- 16 positions x 2 output-channel halves x 2 k-steps
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            f32x2 af[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) af[h] = *(const f32x2*)(buf + a_off + (p * 2 + h) * 128);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[p][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[h][j], V[p >> 2][p & 3][j], acc[p][h], 0, 0, 0);
            // the other stage was last read before the previous barrier: fill it once the loads have had time to land
            if (p == 5 && more) write_lds(nxt);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next group's U slab (LDS-DMA) has landed
        __syncthreads();
    }

    // ---- output transform  Y = A^T M A, bias, epilogue; lane: tile (wv, tn), channels 32*split + 16*h + 4*q + r
    const float* bias = a.wpk + (size_t)a.CGin * w_step;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cog = split * 4 + h * 2 + (q >> 1);
        if (cog >= a.CGout) continue;
        const f32x4 bs = *(const f32x4*)(bias + cog * 8 + 4 * (q & 1));
        f32x4 tm[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            tm[xi][0] = (acc[xi * 4 + 0][h] + acc[xi * 4 + 1][h]) + acc[xi * 4 + 2][h];
            tm[xi][1] = (acc[xi * 4 + 1][h] - acc[xi * 4 + 2][h]) - acc[xi * 4 + 3][h];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = y0 + 2 * wv + i, x = x0 + 2 * tn + j;
                if (y >= H || x >= W) continue;
                f32x4 v = (i == 0) ? (tm[0][j] + tm[1][j]) + tm[2][j] : (tm[1][j] - tm[2][j]) - tm[3][j];
                v = v + bs;
                const size_t o = (((size_t)n * a.CGout + cog) * H + y) * (size_t)W * 8 + (size_t)x * 8 + 4 * (q & 1);
                if (add_res) v = v + *(const f32x4*)(a.residual + o);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (mask) {   // ReLU backward: pass the gradient where the forward activation was > 0
                    const f32x4 fw = *(const f32x4*)(a.mask_src + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (fw[e] > 0.f) ? v[e] : 0.f;
                }
                *(f32x4*)(a.out + o) = v;
            }
    }
}

// U = G g G^T from the fp32 direct packing [cig][tap][CoutP][8]; one thread per (cig, co-block, p, co32, ci8) element
__global__ void pack_wino_kernel(const float* __restrict__ pk, float* __restrict__ out, int CGin, int CoutP) {
    const size_t NCB = CoutP / 32;
    const size_t total = (size_t)CGin * NCB * WN_SLAB;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        // slab element index: [p 16][h 2][qh 2][c16 16][ci%4 4]
        const int e = (int)(i % WN_SLAB);
        const size_t sl = i / WN_SLAB;
        const int cb = (int)(sl % NCB), cig = (int)(sl / NCB);
        const int c4 = e & 3, c16 = (e >> 2) & 15, qh = (e >> 6) & 1, h = (e >> 7) & 1, p = e >> 8;
        const int co = cb * 32 + h * 16 + c16, ci = qh * 4 + c4;
        const int xi = p >> 2, nu = p & 3;
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        double u = 0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                u += G[xi][ky] * G[nu][kx] * (double)pk[(((size_t)cig * 9 + ky * 3 + kx) * CoutP + co) * 8 + ci];
        out[i] = (float)u;
    }
    if (i < (size_t)CoutP) out[total + i] = pk[(size_t)CGin * 9 * CoutP * 8 + i];       // bias
}

static inline int round_up_w(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace scipnp

using namespace scipnp;

extern "C" {

size_t scipnp_conv3x3_wino_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    const int CoutP = round_up_w(Cout, 32);
    return (size_t)(Cin / 8) * (CoutP / 32) * WN_SLAB + CoutP;
}

int scipnp_pack_conv3x3_wino(const float* packed_f32, float* packed_wino, int Cin, int Cout, scipnp_stream_t s) {
    SCIPNP_REQUIRE(packed_f32 && packed_wino, "null pointer");
    SCIPNP_REQUIRE(Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad channel counts Cin=%d Cout=%d", Cin, Cout);
    SCIPNP_ALIGNED(packed_f32); SCIPNP_ALIGNED(packed_wino);
    const int CoutP = round_up_w(Cout, 32);
    const size_t total = (size_t)(Cin / 8) * (CoutP / 32) * WN_SLAB;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, packed_f32,
                       packed_wino, Cin / 8, CoutP);
    return launch_status("pack_wino_kernel");
}

int scipnp_conv3x3_c8w(const float* in, const float* packed_wino, float* out, const float* residual, const float* mask_src,
                       int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_wino && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0,
                   "bad shape n=%d Cin=%d Cout=%d h=%d w=%d (channels must be multiples of 8)", n, Cin, Cout, h, w);
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed_wino); SCIPNP_ALIGNED(out);
    if (residual) SCIPNP_ALIGNED(residual);
    if (mask_src) SCIPNP_ALIGNED(mask_src);
    SCIPNP_REQUIRE(!(flags & (4 | 8)), "the Winograd kernel is stride 1 without pixel shuffle");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 8 < (1ll << 31), "image too large for 32-bit tile offsets");
    WinoArgs a;
    a.in = in; a.wpk = packed_wino; a.out = out; a.residual = residual; a.mask_src = mask_src;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w(Cout, 32) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + WN_TW - 1) / WN_TW; a.nty = (h + WN_TH - 1) / WN_TH;
    a.flags = flags;
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    static bool attr_set[2] = {false, false};
    const int tag = (flags & 0x100) ? 1 : 0;
    const void* fn = tag ? (const void*)conv3x3_c8w_kernel<1> : (const void*)conv3x3_c8w_kernel<0>;
    if (!attr_set[tag]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WN_LDS_BYTES);
        if (e != hipSuccess) return fail(SCIPNP_EHIP, "hipFuncSetAttribute(conv3x3_c8w, %zu B LDS): %s", WN_LDS_BYTES,
                                         hipGetErrorString(e));
        attr_set[tag] = true;
    }
    if (tag) hipLaunchKernelGGL((conv3x3_c8w_kernel<1>), dim3((unsigned)total), dim3(WN_THREADS), WN_LDS_BYTES, (hipStream_t)s, a);
    else hipLaunchKernelGGL((conv3x3_c8w_kernel<0>), dim3((unsigned)total), dim3(WN_THREADS), WN_LDS_BYTES, (hipStream_t)s, a);
    return launch_status("conv3x3_c8w_kernel");
}

}  // extern "C"
