// DDnet (deep demosaicking) glue kernels; the convolutions are conv.hip / conv_split.hip.
//   reference: models/network_demosaicking.py:381-463 (DDnet.forward: channel sum -> mosaic, Bayer planes, gate
//              scalars weight_tensor_in/in2/out, temp1 / temp11 / temp2 DenBlocks), :186-244 and :310-379 (DenBlock:
//              torch.cat of the three frames, `in1 + x`, bilinear x2 + fusion), packages/DDnet/DDnet_test.py:166-216
//              (circular 5-frame window), dvp_linear_inv_2_stage_ADMM_tensor_online.py:192-194, :242-244 (call sites).
#include "common.hpp"

namespace scipnp {

// DenBlock input of evaluation e: channels (frame idx[e][0], idx[e][1], idx[e][2]) x C of the planar source
// [F][C][HW], each multiplied by its gate scalar scale[e][i][c] (no multiply when scale == nullptr, like the second
// stage of the reference), zero-padded to 8*G channels.  Written as fp32 c8 [E][G][HW][8] or split-fp16 c8s.
template <int C>
__global__ void __launch_bounds__(256)
ddnet_gather_kernel(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ scale,
                    float* __restrict__ out_c8, char* __restrict__ out_c8s, size_t HW) {
    constexpr int K = 3 * C, G = (K + 7) / 8;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (p >= HW) return;
    float v[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[g][k] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float* f = src + (size_t)idx[e * 3 + i] * C * HW + p;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int k = i * C + c;
            const float x = f[(size_t)c * HW];
            v[k / 8][k % 8] = scale ? x * scale[(e * 3 + i) * C + c] : x;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (out_c8) {
            float4* d = (float4*)(out_c8 + (((size_t)e * G + g) * HW + p) * 8);
            d[0] = make_float4(v[g][0], v[g][1], v[g][2], v[g][3]);
            d[1] = make_float4(v[g][4], v[g][5], v[g][6], v[g][7]);
        }
        if (out_c8s) {
            char* base = out_c8s + ((size_t)e * G + g) * (2 * HW * 16);
            split8_store(v[g], base + p * 16, base + HW * 16 + p * 16);
        }
    }
}

// out[e][c][p] = in1[e][c][p] + x_c8[e][0][p][c],  c < Cout, with in1 = the (scaled) centre frame of the triplet,
// broadcast over the output channels when the source has one channel (temp1: mosaic -> 3 channels); in1 = 0 when
// src == nullptr (plain c8 -> planar copy of the fusion block's output).
__global__ void __launch_bounds__(256)
ddnet_finish_kernel(const float* __restrict__ src, const int* __restrict__ idx, const float* __restrict__ scale,
                    const float* __restrict__ x_c8, float* __restrict__ out, int C, int Cout, size_t HW) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (p >= HW) return;
    const float4 x = *(const float4*)(x_c8 + ((size_t)e * HW + p) * 8);
    const float xv[4] = {x.x, x.y, x.z, x.w};
    for (int c = 0; c < Cout; ++c) {
        float r = xv[c];
        if (src) {
            const int cs = (C == 1) ? 0 : c;
            float in1 = src[((size_t)idx[e * 3 + 1] * C + cs) * HW + p];
            if (scale) in1 = in1 * scale[(e * 3 + 1) * C + cs];
            r = in1 + r;
        }
        out[((size_t)e * Cout + c) * HW + p] = r;
    }
}

// nn.UpsamplingBilinear2d(scale_factor=2) (align_corners=True) of planar [E][4][h][w], written as the 8-channel
// (4 real) c8 / c8s input of the fusion block at [E][1][2h][2w].  Index / weight arithmetic as PyTorch's CPU kernel:
// src = i*(in-1)/(out-1) in fp32, i0 = floor, l1 = src - i0, l0 = 1 - l1, value = l0h*(l0w*a + l1w*b) + l1h*(l0w*c + l1w*d).
__global__ void __launch_bounds__(256)
bilinear_up2_kernel(const float* __restrict__ in, float* __restrict__ out_c8, char* __restrict__ out_c8s, int h, int w) {
    const int H = 2 * h, W = 2 * w;
    const size_t HW = (size_t)H * W;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (p >= HW) return;
    const int Y = (int)(p / W), X = (int)(p % W);
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float ry = sy * (float)Y, rx = sx * (float)X;
    const int y0 = min((int)ry, h - 1), x0 = min((int)rx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly1 = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* s = in + ((size_t)e * 4 + c) * h * w;
        const float a = s[(size_t)y0 * w + x0], b = s[(size_t)y0 * w + x1];
        const float cc = s[(size_t)y1 * w + x0], d = s[(size_t)y1 * w + x1];
        v[c] = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * cc + lx1 * d);
    }
    if (out_c8) {
        float4* dst = (float4*)(out_c8 + ((size_t)e * HW + p) * 8);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (out_c8s) {
        char* base = out_c8s + (size_t)e * (2 * HW * 16);
        split8_store(v, base + p * 16, base + HW * 16 + p * 16);
    }
}

// x_out = a3[0]*x_out1 + a3[1]*x_out2 (per-channel gates), both branches stacked in one planar tensor [2B][3][HW]
__global__ void __launch_bounds__(256)
ddnet_mix_kernel(const float* __restrict__ o, const float* __restrict__ a3, float* __restrict__ out, int B, size_t HW) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= HW) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float u = a3[c] * o[((size_t)n * 3 + c) * HW + p];
        const float v = a3[3 + c] * o[((size_t)(B + n) * 3 + c) * HW + p];
        out[((size_t)n * 3 + c) * HW + p] = u + v;
    }
}

// v = x + coef*b on the plane-major state [B][4][M][N] -> the Bayer planes (temp11 input) and the mosaic [B][H][W]
// (temp1 input).  reference dvp...:168-171 (xb_all, x_bayer) + network_demosaicking.py:425-437.
__global__ void __launch_bounds__(256)
pm_ddnet_inputs_kernel(const float* __restrict__ x, const float* __restrict__ b, float coef,
                       float* __restrict__ planes, float* __restrict__ mosaic, int M, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    const int t = blockIdx.z;
    if (n >= N) return;
    const size_t plane = (size_t)M * N;
    const int W = 2 * N;
    float v[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
        const size_t o = ((size_t)t * 4 + ib) * plane + (size_t)m * N + n;
        v[ib] = b ? (x[o] + coef * b[o]) : x[o];
        planes[o] = v[ib];
    }
    float* mt = mosaic + (size_t)t * 4 * plane;
    *(float2*)(mt + (size_t)(2 * m) * W + 2 * n) = make_float2(v[0], v[1]);
    *(float2*)(mt + (size_t)(2 * m + 1) * W + 2 * n) = make_float2(v[2], v[3]);
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_ddnet_gather(const float* src, const int* idx, const float* scale, float* out_c8, void* out_c8s, int E, int C,
                        int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(src && idx && (out_c8 || out_c8s) && E > 0 && E <= 65535 && h > 0 && w > 0, "bad arguments");
    SCIPNP_REQUIRE(C == 1 || C == 3 || C == 4, "channels per frame must be 1, 3 or 4 (got %d)", C);
    if (out_c8) SCIPNP_ALIGNED(out_c8);
    if (out_c8s) SCIPNP_ALIGNED(out_c8s);
    const size_t HW = (size_t)h * w;
    const dim3 grid((unsigned)((HW + 255) / 256), E), block(256);
    hipStream_t st = (hipStream_t)s;
    if (C == 1) hipLaunchKernelGGL(ddnet_gather_kernel<1>, grid, block, 0, st, src, idx, scale, out_c8, (char*)out_c8s, HW);
    else if (C == 3) hipLaunchKernelGGL(ddnet_gather_kernel<3>, grid, block, 0, st, src, idx, scale, out_c8, (char*)out_c8s, HW);
    else hipLaunchKernelGGL(ddnet_gather_kernel<4>, grid, block, 0, st, src, idx, scale, out_c8, (char*)out_c8s, HW);
    return launch_status("ddnet_gather_kernel");
}

int scipnp_ddnet_finish(const float* src, const int* idx, const float* scale, const float* x_c8, float* out, int E, int C,
                        int Cout, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x_c8 && out && E > 0 && E <= 65535 && h > 0 && w > 0, "bad arguments");
    SCIPNP_REQUIRE(Cout >= 1 && Cout <= 4 && (!src || (idx && (C == 1 || C == Cout))), "bad channel counts C=%d Cout=%d", C, Cout);
    SCIPNP_ALIGNED(x_c8);
    const size_t HW = (size_t)h * w;
    hipLaunchKernelGGL(ddnet_finish_kernel, dim3((unsigned)((HW + 255) / 256), E), dim3(256), 0, (hipStream_t)s, src, idx,
                       scale, x_c8, out, C, Cout, HW);
    return launch_status("ddnet_finish_kernel");
}

int scipnp_bilinear_up2_c8(const float* in, float* out_c8, void* out_c8s, int E, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && (out_c8 || out_c8s) && E > 0 && E <= 65535 && h > 0 && w > 0, "bad arguments");
    if (out_c8) SCIPNP_ALIGNED(out_c8);
    if (out_c8s) SCIPNP_ALIGNED(out_c8s);
    const size_t HW = (size_t)h * w * 4;
    hipLaunchKernelGGL(bilinear_up2_kernel, dim3((unsigned)((HW + 255) / 256), E), dim3(256), 0, (hipStream_t)s, in, out_c8,
                       (char*)out_c8s, h, w);
    return launch_status("bilinear_up2_kernel");
}

int scipnp_ddnet_mix(const float* branches, const float* gates, float* out, int B, int H, int W, scipnp_stream_t s) {
    SCIPNP_REQUIRE(branches && gates && out && B > 0 && B <= 65535 && H > 0 && W > 0, "bad arguments");
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(ddnet_mix_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)s, branches, gates,
                       out, B, HW);
    return launch_status("ddnet_mix_kernel");
}

int scipnp_pm_ddnet_inputs(const float* x, const float* b, float coef, float* planes, float* mosaic, int M, int N, int B,
                           scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && planes && mosaic && M > 0 && N > 0 && B > 0 && M <= 65535 && B <= 65535, "bad arguments");
    SCIPNP_ALIGNED(mosaic);
    hipLaunchKernelGGL(pm_ddnet_inputs_kernel, dim3((N + 255) / 256, M, B), dim3(256), 0, (hipStream_t)s, x, b, coef, planes,
                       mosaic, M, N);
    return launch_status("pm_ddnet_inputs_kernel");
}

}  // extern "C"
