// 3x3 convolution (zero pad 1, stride 1 or 2) as an implicit GEMM on the CDNA4 matrix cores with
// fp32 operands: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate -- the only MFMA
// precision that meets the 1e-5 per-iterate parity bar of the PnP loop out of the box; plain bf16/fp16
// operands fail it by 2-3 orders of magnitude).
//
// Data layout ("c8"): activations [n][C/8][h][w][8] -- 8-channel groups innermost.  With
//   A = weights  [co (32 rows)][k]      lane l holds A[l&31][l>>5]
//   B = pixels   [k][px (32 cols)]      lane l holds B[l>>5][l&31]
// and k running over the channels of one group, one ds_read_b128 per lane yields four MFMA k-steps
// for both operands (lane half h reads channels 4h..4h+3), and the 32x32 accumulator of a lane
// holds, for each of its four row groups g, the four consecutive output channels 8g+4h..8g+4h+3 of
// its pixel: the epilogue is one 16-byte store per lane and group, 1 KiB contiguous per wave --
// i.e. the output is written directly in the c8 layout the next layer reads.
//
// Work decomposition: workgroup = 4 waves = output tile 8 rows x 32 columns x (32*COB) channels;
// wave w owns rows 2w, 2w+1 (2 pixel blocks x COB channel blocks = 2*COB accumulators).  K loop =
// channel groups of the input; per group the input halo tile and the 9 x (32*COB) x 8 weight slab go
// L2/HBM -> registers -> LDS, double-buffered: the loads for group g+1 are issued before the MFMAs of
// group g and written to the other LDS buffer after the third tap, so the only thing left at the end
// of a group is one barrier.  The kernel is MFMA-bound by construction: per group and wave
// 9*4*2*COB MFMAs of 64 cycles against 2+COB ds_read_b128 per tap.
//
// Epilogue modes: bias (+ residual) (+ ReLU) in c8; or PixelShuffle(2) folded into the store
// (channel co = 4c + 2dy + dx goes to pixel (2y+dy, 2x+dx), channel c; residual in that layout).
#include "common.hpp"

namespace scipnp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CV_TH = 8, CV_TW = 32;
constexpr int CV_THREADS = 256;

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int COB, int STRIDE>
struct ConvCfg {
    static constexpr int TWP = (CV_TW - 1) * STRIDE + 3;   // input tile columns (34 / 65)
    static constexpr int THP = (CV_TH - 1) * STRIDE + 3;   // input tile rows    (10 / 17)
    static constexpr int IN_FLOATS = THP * TWP * 8;
    static constexpr int IN_VEC = IN_FLOATS / 4;
    static constexpr int IN_ITERS = (IN_VEC + CV_THREADS - 1) / CV_THREADS;
    static constexpr int COUTP = 32 * COB;
    static constexpr int W_FLOATS = 9 * COUTP * 8;
    static constexpr int W_VEC = W_FLOATS / 4;
    static constexpr int W_ITERS = (W_VEC + CV_THREADS - 1) / CV_THREADS;
    static constexpr int STAGE = IN_FLOATS + W_FLOATS;
    static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * sizeof(float);
    static constexpr int WAVES_PER_SIMD = (LDS_BYTES * 2 <= 160 * 1024 && COB <= 3) ? 2 : 1;
};

struct ConvArgs {
    const float* in;
    const float* wpk;
    float* out;
    const float* residual;
    const float* mask_src;   // ReLU-backward mask source (forward activation), flag bit 4
    int CGin, CGout, CoutP_total, nsplit;
    int H, W;        // input size
    int Ho, Wo;      // conv output size (before any pixel shuffle)
    int flags;
};

// TAG only changes the kernel's symbol name (0 = body/tail layers, 1 = network head layer with its
// short K) so that per-kernel profiler statistics of the body layers are not diluted by the head.
template <int COB, int TAG, int STRIDE, int SHUF>
__global__ void __launch_bounds__(CV_THREADS, (ConvCfg<COB, STRIDE>::WAVES_PER_SIMD))
conv3x3_c8_kernel(const ConvArgs a) {
    using Cfg = ConvCfg<COB, STRIDE>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware order (round 6, as the Winograd kernels): workgroups are dealt round-robin to the 8 XCDs in dispatch order; remapped, one
    // XCD works through a contiguous run of (tile, output-channel split) pairs -- the splits of a tile adjacent, tiles row by row -- so
    // the input tile a split re-reads and the halo rows of the tile below are found in that XCD's L2
    int x0, y0, n, split;
    {
        const unsigned ntx = gridDim.x, nty = gridDim.y, nz = gridDim.z, total = ntx * nty * nz;
        unsigned lin = (blockIdx.z * nty + blockIdx.y) * ntx + blockIdx.x;
        if ((total & 7u) == 0u) lin = (lin & 7u) * (total >> 3) + (lin >> 3);
        const unsigned ns = (unsigned)a.nsplit;
        split = (int)(lin % ns);
        unsigned t = lin / ns;
        x0 = (int)(t % ntx) * CV_TW;
        t /= ntx;
        y0 = (int)(t % nty) * CV_TH;
        n = (int)(t / nty);
    }
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    // ---- staging plan, computed once: per-thread source offsets (floats) within one channel group
    // of the input / one channel-group slab of the weights; -1 = zero fill (outside the image).
    int in_off[Cfg::IN_ITERS];
#pragma unroll
    for (int k = 0; k < Cfg::IN_ITERS; ++k) {
        const int e = tid + k * CV_THREADS;
        in_off[k] = -1;
        if (e < Cfg::IN_VEC) {
            const int pix = e >> 1, half = e & 1;
            const int r = pix / Cfg::TWP, c = pix - r * Cfg::TWP;
            const int gy = y0 * STRIDE - 1 + r, gx = x0 * STRIDE - 1 + c;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) in_off[k] = (gy * W + gx) * 8 + 4 * half;
        }
    }
    int w_off[Cfg::W_ITERS];
#pragma unroll
    for (int k = 0; k < Cfg::W_ITERS; ++k) {
        const int e = tid + k * CV_THREADS;
        w_off[k] = -1;
        if (e < Cfg::W_VEC) {
            const int tap = e / (Cfg::COUTP * 2), rem = e - tap * (Cfg::COUTP * 2);
            w_off[k] = (tap * a.CoutP_total * 2 + rem) * 4;
        }
    }
    const float* in_g = a.in + (size_t)n * a.CGin * HW * 8;              // advanced by HW*8 per group
    const float* w_g = a.wpk + (size_t)split * Cfg::COUTP * 8;           // advanced by 9*CoutP*8 per group
    const size_t w_step = (size_t)9 * a.CoutP_total * 8;

    f32x4 st_in[Cfg::IN_ITERS];
    f32x4 st_w[Cfg::W_ITERS];
    auto issue_loads = [&]() {
#pragma unroll
        for (int k = 0; k < Cfg::IN_ITERS; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in_off[k] >= 0) v = *(const f32x4*)(in_g + in_off[k]);
            st_in[k] = v;
        }
#pragma unroll
        for (int k = 0; k < Cfg::W_ITERS; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w_off[k] >= 0) v = *(const f32x4*)(w_g + w_off[k]);
            st_w[k] = v;
        }
        in_g += HW * 8;
        w_g += w_step;
    };
    auto write_lds = [&](float* buf) {
#pragma unroll
        for (int k = 0; k < Cfg::IN_ITERS; ++k) {
            const int e = tid + k * CV_THREADS;
            if (e < Cfg::IN_VEC) *(f32x4*)(buf + 4 * e) = st_in[k];
        }
#pragma unroll
        for (int k = 0; k < Cfg::W_ITERS; ++k) {
            const int e = tid + k * CV_THREADS;
            if (e < Cfg::W_VEC) *(f32x4*)(buf + Cfg::IN_FLOATS + 4 * e) = st_w[k];
        }
    };

    f32x16 acc[2][COB];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int cb = 0; cb < COB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][cb][r] = 0.f;

    issue_loads();
    write_lds(smem);
    __syncthreads();

    // per-lane LDS offsets (floats)
    const int b_off = ((2 * wv * STRIDE) * Cfg::TWP + li * STRIDE) * 8 + 4 * lh;   // + ((pb*S+ky)*TWP + kx)*8
    const int a_off = Cfg::IN_FLOATS + li * 8 + 4 * lh;                            // + (tap*COUTP + cb*32)*8

    for (int cig = 0; cig < a.CGin; ++cig) {
        const float* buf = smem + (cig & 1) * Cfg::STAGE;
        const bool more = (cig + 1 < a.CGin);
        if (more) issue_loads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            f32x4 bf[2], af[COB];
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
                bf[pb] = *(const f32x4*)(buf + b_off + ((pb * STRIDE + ky) * Cfg::TWP + kx) * 8);
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
                af[cb] = *(const f32x4*)(buf + a_off + (tap * Cfg::COUTP + cb * 32) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int cb = 0; cb < COB; ++cb)
                        acc[pb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][j], bf[pb][j], acc[pb][cb], 0, 0, 0);
            // the other LDS buffer was last read before the previous barrier: fill it as soon as the
            // loads have had a few thousand cycles to land, instead of serialising it before the barrier
            if (tap == 2 && more) write_lds(smem + ((cig + 1) & 1) * Cfg::STAGE);
        }
        __syncthreads();
    }

    // ---- epilogue
    const float* bias = a.wpk + (size_t)a.CGin * 9 * a.CoutP_total * 8;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
    const int Ho = a.Ho, Wo = a.Wo;
    const int x = x0 + li;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int y = y0 + 2 * wv + pb;
        if (y < Ho && x < Wo) {
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cog = (split * COB + cb) * 4 + g;
                    if (cog < a.CGout) {
                        const f32x4 bs = *(const f32x4*)(bias + cog * 8 + 4 * lh);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[pb][cb][4 * g + e] + bs[e];
                        if (!SHUF) {
                            const size_t o = (((size_t)n * a.CGout + cog) * Ho + y) * (size_t)Wo * 8 + (size_t)x * 8 + 4 * lh;
                            if (add_res) {
                                const f32x4 rs = *(const f32x4*)(a.residual + o);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] + rs[e];
                            }
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
                        } else {
                            // PixelShuffle(2): conv channel 8*cog + 4*lh + e  ->  channel c = 2*cog + lh of
                            // pixel (2y + (e>>1), 2x + (e&1));  shuffled tensor [n][CGout/4][2Ho][2Wo][8]
                            const int CGs = a.CGout >> 2;
                            const size_t base = (((size_t)n * CGs + (cog >> 2)) * (2 * Ho) + 2 * y) * (size_t)(2 * Wo) * 8 +
                                                (size_t)(2 * x) * 8 + (cog & 3) * 2 + lh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const size_t o = base + (size_t)(e >> 1) * (2 * Wo) * 8 + (e & 1) * 8;
                                float r = v[e];
                                if (add_res) r = r + a.residual[o];
                                if (relu) r = fmaxf(r, 0.f);
                                a.out[o] = r;
                            }
                        }
                    }
                }
        }
    }
}

template <int COB, int TAG, int STRIDE, int SHUF>
static int launch_conv(const ConvArgs& a, int n, hipStream_t st) {
    using Cfg = ConvCfg<COB, STRIDE>;
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_c8_kernel<COB, TAG, STRIDE, SHUF>, Cfg::LDS_BYTES, "conv3x3_c8")) return rc;
    const dim3 grid((a.Wo + CV_TW - 1) / CV_TW, (a.Ho + CV_TH - 1) / CV_TH, n * a.nsplit);
    hipLaunchKernelGGL((conv3x3_c8_kernel<COB, TAG, STRIDE, SHUF>), grid, dim3(CV_THREADS), Cfg::LDS_BYTES, st, a);
    return launch_status("conv3x3_c8_kernel");
}

template <int STRIDE, int SHUF>
static int dispatch_cob(ConvArgs& a, int n, hipStream_t st) {
    const int CoutP = a.CoutP_total;
    if (CoutP % 96 == 0) {
        a.nsplit = CoutP / 96;
        if (STRIDE == 1 && !SHUF && (a.flags & 0x100)) return launch_conv<3, 1, 1, 0>(a, n, st);
        return launch_conv<3, 0, STRIDE, SHUF>(a, n, st);
    }
    if (CoutP % 128 == 0) { a.nsplit = CoutP / 128; return launch_conv<4, 0, STRIDE, SHUF>(a, n, st); }
    if (CoutP % 64 == 0) { a.nsplit = CoutP / 64; return launch_conv<2, 0, STRIDE, SHUF>(a, n, st); }
    a.nsplit = CoutP / 32;
    return launch_conv<1, 0, STRIDE, SHUF>(a, n, st);
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

/* scipnp_conv3x3_packed_floats / scipnp_pack_conv3x3_weights (host functions): csrc/host_pack.hip */

int scipnp_conv3x3_c8_ex(const float* in, const float* packed_w, float* out, const float* residual,
                         const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_w && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0,
                   "bad shape n=%d Cin=%d Cout=%d h=%d w=%d (channels must be multiples of 8)", n, Cin, Cout, h, w);
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed_w); SCIPNP_ALIGNED(out);
    if (residual) SCIPNP_ALIGNED(residual);
    if (mask_src) SCIPNP_ALIGNED(mask_src);
    const bool stride2 = flags & 4, shuf = flags & 8;
    SCIPNP_REQUIRE(!(stride2 && shuf), "stride-2 and pixel-shuffle epilogue cannot be combined");
    SCIPNP_REQUIRE(!((flags & 16) && shuf), "ReLU-mask epilogue excludes pixel shuffle");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!shuf || Cout % 32 == 0, "pixel-shuffle epilogue needs Cout %% 32 == 0 (got %d)", Cout);
    SCIPNP_REQUIRE((long long)h * w * 8 < (1ll << 31), "image too large for 32-bit tile offsets");
    ConvArgs a;
    a.in = in; a.wpk = packed_w; a.out = out; a.residual = residual; a.mask_src = mask_src;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.CoutP_total = round_up(Cout, 32); a.nsplit = 1;
    a.H = h; a.W = w;
    a.Ho = stride2 ? (h - 1) / 2 + 1 : h;
    a.Wo = stride2 ? (w - 1) / 2 + 1 : w;
    a.flags = flags;
    SCIPNP_REQUIRE((long long)n * (a.CoutP_total / 32) <= 65535, "grid too large");
    hipStream_t st = (hipStream_t)s;
    if (stride2) return dispatch_cob<2, 0>(a, n, st);
    if (shuf) return dispatch_cob<1, 1>(a, n, st);
    return dispatch_cob<1, 0>(a, n, st);
}

int scipnp_conv3x3_c8(const float* in, const float* packed_w, float* out, const float* residual, int n, int Cin,
                      int Cout, int h, int w, int flags, scipnp_stream_t s) {
    // historical form: with flag bit4 (and no bit1) `residual` is the mask source
    if ((flags & 16) && !(flags & 2))
        return scipnp_conv3x3_c8_ex(in, packed_w, out, nullptr, residual, n, Cin, Cout, h, w, flags, s);
    SCIPNP_REQUIRE(!(flags & 16), "mask + residual needs scipnp_conv3x3_c8_ex");
    return scipnp_conv3x3_c8_ex(in, packed_w, out, residual, nullptr, n, Cin, Cout, h, w, flags, s);
}

int scipnp_ffdnet_forward(const float* in_c8, float* out_c8, const float* const* packed, int nb, int nc,
                          float* scratch0, float* scratch1, int B, int M, int N, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in_c8 && out_c8 && packed && scratch0 && scratch1, "null pointer");
    SCIPNP_REQUIRE(nb >= 2 && nc % 8 == 0 && nc > 0, "bad network shape nb=%d nc=%d", nb, nc);
    float* buf[2] = {scratch0, scratch1};
    int rc = scipnp_conv3x3_c8(in_c8, packed[0], buf[0], nullptr, B, 16, nc, M, N, 1 | 0x100, s);
    if (rc) return rc;
    int cur = 0;
    for (int l = 1; l < nb - 1; ++l) {
        rc = scipnp_conv3x3_c8(buf[cur], packed[l], buf[cur ^ 1], nullptr, B, nc, nc, M, N, 1, s);
        if (rc) return rc;
        cur ^= 1;
    }
    return scipnp_conv3x3_c8(buf[cur], packed[nb - 1], out_c8, nullptr, B, nc, 16, M, N, 0, s);
}

}  // extern "C"
