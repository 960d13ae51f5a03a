// FastDVDnet glue kernels (the convolutions themselves are conv.hip): building the 3-frame DenBlock
// input from planar frames with CIRCULAR temporal indexing, and the residual `in1 - x` of a DenBlock.
//   reference: packages/fastdvdnet/models.py:179-198 (torch.cat of frames and noise maps, in1 - x),
//              packages/fastdvdnet/fastdvdnet.py:113-116 (circular window (frameidx + [0..4] - 2) mod N)
#include "common.hpp"

namespace scipnp {

// out c8 [B][2][H][W][8]: group 0 = (f0.rgb, sigma, f1.rgb, sigma), group 1 = (f2.rgb, sigma, 0,0,0,0)
// with f0,f1,f2 = frames (n-1, n, n+1) mod B of the planar input [B][3][H][W].
__global__ void __launch_bounds__(256)
fastdvd_pack_kernel(const float* __restrict__ frames, float* __restrict__ out, int B, int U, size_t HW, float sigma) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;                               // frame t U + u of a unit batch [B][U] (U = 1: frame t)
    if (p >= HW) return;
    const int t = n / U, u = n - t * U;
    const int f0 = ((t + B - 1) % B) * U + u, f2 = ((t + 1) % B) * U + u;
    const float* a = frames + (size_t)f0 * 3 * HW + p;
    const float* b = frames + (size_t)n * 3 * HW + p;
    const float* c = frames + (size_t)f2 * 3 * HW + p;
    float4* d0 = (float4*)(out + (((size_t)n * 2 + 0) * HW + p) * 8);
    float4* d1 = (float4*)(out + (((size_t)n * 2 + 1) * HW + p) * 8);
    d0[0] = make_float4(a[0], a[HW], a[2 * HW], sigma);
    d0[1] = make_float4(b[0], b[HW], b[2 * HW], sigma);
    d1[0] = make_float4(c[0], c[HW], c[2 * HW], sigma);
    d1[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// same, written in the split-fp16 c8s layout [B][2][2 planes][HW][8 fp16] of conv_split.hip
__global__ void __launch_bounds__(256)
fastdvd_pack_c8s_kernel(const float* __restrict__ frames, char* __restrict__ out, int B, int U, size_t HW, float sigma) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= HW) return;
    const int t = n / U, u = n - t * U;
    const int f0 = ((t + B - 1) % B) * U + u, f2 = ((t + 1) % B) * U + u;
    const float* a = frames + (size_t)f0 * 3 * HW + p;
    const float* b = frames + (size_t)n * 3 * HW + p;
    const float* c = frames + (size_t)f2 * 3 * HW + p;
    const float g0[8] = {a[0], a[HW], a[2 * HW], sigma, b[0], b[HW], b[2 * HW], sigma};
    const float g1[8] = {c[0], c[HW], c[2 * HW], sigma, 0.f, 0.f, 0.f, 0.f};
    char* base = out + (size_t)n * 2 * (2 * HW * 16);
    split8_store(g0, base + p * 16, base + HW * 16 + p * 16);
    split8_store(g1, base + 2 * HW * 16 + p * 16, base + 3 * HW * 16 + p * 16);
}

// out[n][c][p] = center[n][c][p] - x_c8[n][0][p][c]   (c < 3)
__global__ void __launch_bounds__(256)
fastdvd_finish_kernel(const float* __restrict__ center, const float* __restrict__ x_c8, float* __restrict__ out,
                      size_t HW) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= HW) return;
    const float4 x = *(const float4*)(x_c8 + ((size_t)n * HW + p) * 8);
    const size_t o = (size_t)n * 3 * HW + p;
    out[o] = center[o] - x.x;
    out[o + HW] = center[o + HW] - x.y;
    out[o + 2 * HW] = center[o + 2 * HW] - x.z;
}

// ------------------------------------------------------------------ backward-pass glue (online finetune)
// zero-insertion upsample: out[n][cg][2y][2x] = in[n][cg][y][x], zero elsewhere (out is H x W).  The gradient of
// a stride-2 conv w.r.t. its input / weights is the stride-1 backward applied to this tensor.
__global__ void __launch_bounds__(256)
upsample_zero_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index over out
    if (i >= total) return;
    const int half = i & 1;
    const size_t pix = i >> 1;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H);
    const size_t img = pix / ((size_t)W * H);                           // n*CG + cg
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(X & 1) && !(Y & 1) && (Y >> 1) < h && (X >> 1) < w)
        v = *(const float4*)(in + ((img * h + (Y >> 1)) * (size_t)w + (X >> 1)) * 8 + 4 * half);
    *(float4*)(out + pix * 8 + 4 * half) = v;
}

// PixelShuffle(2) backward in c8: dconv[n][4c+2dy+dx][y][x] = dshuf[n][c][2y+dy][2x+dx]
// dshuf: [n][Cs/8][2h][2w][8], dconv: [n][4*Cs/8][h][w][8]; thread = (n, conv channel group, y, x)
__global__ void __launch_bounds__(256)
unshuffle_bwd_kernel(const float* __restrict__ dshuf, float* __restrict__ dconv, int CGs, int h, int w, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int cog = (int)((i / ((size_t)w * h)) % (4 * CGs));
    const size_t n = i / ((size_t)w * h * 4 * CGs);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int co = cog * 8 + e;                 // conv channel
        const int c = co >> 2, dy = (co >> 1) & 1, dx = co & 1;
        v[e] = dshuf[(((n * CGs + (c >> 3)) * (2 * h) + 2 * y + dy) * (size_t)(2 * w) + 2 * x + dx) * 8 + (c & 7)];
    }
    float4* d = (float4*)(dconv + i * 8);
    d[0] = make_float4(v[0], v[1], v[2], v[3]);
    d[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// the same two data movements on c8s tensors ([n][CG][2 planes][HW][8 fp16]): pure permutations, applied to the hi and
// the lo' plane alike.  `img` = (n*CG + cg)*2 + plane.
__global__ void __launch_bounds__(256)
upsample_zero_c8s_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int h, int w, int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // pixel index over out (16 B each)
    if (i >= total) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const size_t img = i / ((size_t)W * H);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!(X & 1) && !(Y & 1) && (Y >> 1) < h && (X >> 1) < w) v = in[(img * h + (Y >> 1)) * (size_t)w + (X >> 1)];
    out[i] = v;
}

__global__ void __launch_bounds__(256)
unshuffle_bwd_c8s_kernel(const _Float16* __restrict__ dshuf, _Float16* __restrict__ dconv, int CGs, int h, int w, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (n, conv channel group, plane, y, x)
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const int plane = (int)((i / ((size_t)w * h)) & 1);
    const int cog = (int)((i / ((size_t)w * h * 2)) % (4 * CGs));
    const size_t n = i / ((size_t)w * h * 2 * 4 * CGs);
    half8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int co = cog * 8 + e;
        const int c = co >> 2, dy = (co >> 1) & 1, dx = co & 1;
        v[e] = dshuf[((((n * CGs + (c >> 3)) * 2 + plane) * (2 * h) + 2 * y + dy) * (size_t)(2 * w) + 2 * x + dx) * 8 + (c & 7)];
    }
    *(half8_t*)(dconv + i * 8) = v;
}

// measurement loss on planar RGB frames and its gradient (reference test_fastdvdnet.py:424-431):
//   L = mean_{r,c} ( sum_t Phi_mosaic[r,c,t] * out[t][color(r,c)][r][c] - y_mosaic[r,c] )^2
// thread = one Bayer quad; Phi, y plane-major ([B][4][M][N], [4][M][N]); dout planar, zero off the CFA sites.
__global__ void __launch_bounds__(256)
fastdvd_loss_grad_kernel(const float* __restrict__ out, const float* __restrict__ Phi, const float* __restrict__ y,
                         float* __restrict__ dout, double* __restrict__ loss_part, int M, int N, int B) {
    __shared__ double red[16];
    const size_t plane = (size_t)M * N;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (q < plane) {
        const int m = (int)(q / N), n = (int)(q % N);
        const int W = 2 * N;
        const size_t HW = 4 * plane;
        const int col[4] = {0, 1, 1, 2};
        size_t off[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
            off[ib] = (size_t)col[ib] * HW + (size_t)(2 * m + (ib >> 1)) * W + 2 * n + (ib & 1);
        float up[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < B; ++t)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
                up[ib] = up[ib] + out[(size_t)t * 3 * HW + off[ib]] * Phi[((size_t)t * 4 + ib) * plane + q];
        const float norm = 2.0f / (float)(4 * plane);
        float g[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float d = up[ib] - y[(size_t)ib * plane + q];
            acc += (double)(d * d);
            g[ib] = norm * d;
        }
        for (int t = 0; t < B; ++t) {
            float* dt = dout + (size_t)t * 3 * HW;
            // zero the quad in all three planes, then the four CFA sites
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
                    *(float2*)(dt + (size_t)c * HW + (size_t)(2 * m + dy) * W + 2 * n) = make_float2(0.f, 0.f);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) dt[off[ib]] = g[ib] * Phi[((size_t)t * 4 + ib) * plane + q];
        }
    }
    const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = s;
}

// DenBlock residual backward: out = center - x  ->  dx (c8, 8 channels: 0..2 = -dout, rest 0)
__global__ void __launch_bounds__(256)
fastdvd_finish_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx_c8, size_t HW) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= HW) return;
    const size_t o = (size_t)n * 3 * HW + p;
    float4* d = (float4*)(dx_c8 + ((size_t)n * HW + p) * 8);
    d[0] = make_float4(-dout[o], -dout[o + HW], -dout[o + 2 * HW], 0.f);
    d[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// gradient w.r.t. the planar frames that fed scipnp_fastdvd_pack_triplets (+ the `center -` path):
//   dframes[m][c] = dtin[m+1][c] + dtin[m][4+c] + dtin[m-1][8+c] + extra[m][c]     (indices mod B)
__global__ void __launch_bounds__(256)
fastdvd_unpack_bwd_kernel(const float* __restrict__ dtin, const float* __restrict__ extra, float* __restrict__ dframes,
                          int B, size_t HW) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (p >= HW) return;
    const int nxt = (m + 1) % B, prv = (m + B - 1) % B;
    const float4 a = *(const float4*)(dtin + (((size_t)nxt * 2 + 0) * HW + p) * 8);          // frame m is f0 of window m+1
    const float4 b = *(const float4*)(dtin + (((size_t)m * 2 + 0) * HW + p) * 8 + 4);        // f1 of window m
    const float4 c = *(const float4*)(dtin + (((size_t)prv * 2 + 1) * HW + p) * 8);          // f2 of window m-1
    const size_t o = (size_t)m * 3 * HW + p;
    const float e0 = extra ? extra[o] : 0.f, e1 = extra ? extra[o + HW] : 0.f, e2 = extra ? extra[o + 2 * HW] : 0.f;
    dframes[o] = ((a.x + b.x) + c.x) + e0;
    dframes[o + HW] = ((a.y + b.y) + c.y) + e1;
    dframes[o + 2 * HW] = ((a.z + b.z) + c.z) + e2;
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_fastdvd_pack_triplets_units(const float* frames, float* out_c8, int B, int units, int H, int W, float sigma,
                                       scipnp_stream_t s) {
    SCIPNP_REQUIRE(frames && out_c8 && B > 0 && units > 0 && (long long)B * units <= 65535 && H > 0 && W > 0, "bad arguments");
    SCIPNP_ALIGNED(out_c8);
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(fastdvd_pack_kernel, dim3((unsigned)((HW + 255) / 256), B * units), dim3(256), 0, (hipStream_t)s, frames,
                       out_c8, B, units, HW, sigma);
    return launch_status("fastdvd_pack_kernel");
}

int scipnp_fastdvd_pack_triplets(const float* frames, float* out_c8, int B, int H, int W, float sigma, scipnp_stream_t s) {
    return scipnp_fastdvd_pack_triplets_units(frames, out_c8, B, 1, H, W, sigma, s);
}

int scipnp_fastdvd_pack_triplets_c8s_units(const float* frames, void* out_c8s, int B, int units, int H, int W, float sigma,
                                           scipnp_stream_t s) {
    SCIPNP_REQUIRE(frames && out_c8s && B > 0 && units > 0 && (long long)B * units <= 65535 && H > 0 && W > 0, "bad arguments");
    SCIPNP_ALIGNED(out_c8s);
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(fastdvd_pack_c8s_kernel, dim3((unsigned)((HW + 255) / 256), B * units), dim3(256), 0, (hipStream_t)s, frames,
                       (char*)out_c8s, B, units, HW, sigma);
    return launch_status("fastdvd_pack_c8s_kernel");
}

int scipnp_fastdvd_pack_triplets_c8s(const float* frames, void* out_c8s, int B, int H, int W, float sigma, scipnp_stream_t s) {
    return scipnp_fastdvd_pack_triplets_c8s_units(frames, out_c8s, B, 1, H, W, sigma, s);
}

int scipnp_fastdvd_finish(const float* center, const float* x_c8, float* out, int B, int H, int W,
                          scipnp_stream_t s) {
    SCIPNP_REQUIRE(center && x_c8 && out && B > 0 && B <= 65535 && H > 0 && W > 0, "bad arguments");
    SCIPNP_ALIGNED(x_c8);
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(fastdvd_finish_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)s, center,
                       x_c8, out, HW);
    return launch_status("fastdvd_finish_kernel");
}

int scipnp_upsample_zero_c8(const float* in, float* out, int n, int C, int h, int w, int H, int W, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && out && n > 0 && C % 8 == 0 && h > 0 && w > 0 && H >= 2 * h - 1 && W >= 2 * w - 1 && H <= 2 * h &&
                   W <= 2 * w, "bad arguments");
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(out);
    const size_t total = (size_t)n * (C / 8) * H * W * 2;
    hipLaunchKernelGGL(upsample_zero_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, in, out, h,
                       w, H, W, total);
    return launch_status("upsample_zero_kernel");
}

int scipnp_pixel_shuffle_bwd_c8(const float* dshuf, float* dconv, int n, int Cs, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(dshuf && dconv && n > 0 && Cs % 8 == 0 && h > 0 && w > 0, "bad arguments");
    SCIPNP_ALIGNED(dconv);
    const size_t total = (size_t)n * (4 * Cs / 8) * h * w;
    hipLaunchKernelGGL(unshuffle_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, dshuf,
                       dconv, Cs / 8, h, w, total);
    return launch_status("unshuffle_bwd_kernel");
}

int scipnp_upsample_zero_c8s(const void* in, void* out, int n, int C, int h, int w, int H, int W, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && out && n > 0 && C % 8 == 0 && h > 0 && w > 0 && H >= 2 * h - 1 && W >= 2 * w - 1 && H <= 2 * h &&
                   W <= 2 * w, "bad arguments");
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(out);
    const size_t total = (size_t)n * (C / 8) * 2 * H * W;
    hipLaunchKernelGGL(upsample_zero_c8s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       (const uint4*)in, (uint4*)out, h, w, H, W, total);
    return launch_status("upsample_zero_c8s_kernel");
}

int scipnp_pixel_shuffle_bwd_c8s(const void* dshuf, void* dconv, int n, int Cs, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(dshuf && dconv && n > 0 && Cs % 8 == 0 && h > 0 && w > 0, "bad arguments");
    SCIPNP_ALIGNED(dconv);
    const size_t total = (size_t)n * (4 * Cs / 8) * 2 * h * w;
    hipLaunchKernelGGL(unshuffle_bwd_c8s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       (const _Float16*)dshuf, (_Float16*)dconv, Cs / 8, h, w, total);
    return launch_status("unshuffle_bwd_c8s_kernel");
}

int scipnp_fastdvd_loss_grad(const float* out, const float* Phi, const float* y, float* dout, double* loss_part, int M,
                             int N, int B, int* nblocks, scipnp_stream_t s) {
    SCIPNP_REQUIRE(nblocks && M > 0 && N > 0 && B > 0, "bad arguments");
    const size_t plane = (size_t)M * N;
    const unsigned blocks = (unsigned)((plane + 255) / 256);
    *nblocks = (int)blocks;
    if (loss_part == nullptr) return SCIPNP_OK;
    SCIPNP_REQUIRE(out && Phi && y && dout, "null pointer");
    hipLaunchKernelGGL(fastdvd_loss_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out, Phi, y, dout, loss_part, M,
                       N, B);
    return launch_status("fastdvd_loss_grad_kernel");
}

int scipnp_fastdvd_finish_bwd(const float* dout, float* dx_c8, int B, int H, int W, scipnp_stream_t s) {
    SCIPNP_REQUIRE(dout && dx_c8 && B > 0 && B <= 65535, "bad arguments");
    SCIPNP_ALIGNED(dx_c8);
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(fastdvd_finish_bwd_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)s, dout,
                       dx_c8, HW);
    return launch_status("fastdvd_finish_bwd_kernel");
}

int scipnp_fastdvd_unpack_bwd(const float* dtin_c8, const float* extra, float* dframes, int B, int H, int W,
                              scipnp_stream_t s) {
    SCIPNP_REQUIRE(dtin_c8 && dframes && B > 0 && B <= 65535, "bad arguments");
    SCIPNP_ALIGNED(dtin_c8);
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(fastdvd_unpack_bwd_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)s,
                       dtin_c8, extra, dframes, B, HW);
    return launch_status("fastdvd_unpack_bwd_kernel");
}

}  // extern "C"
