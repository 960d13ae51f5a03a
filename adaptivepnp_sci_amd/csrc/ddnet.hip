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


// ===================================================================== online finetune of the demosaicker
// packages/DDnet/DDnet_test.py:248-296 (`args.dm_update`): loss = MSE(input CFA-site cube, CFA samples of the network's output)
// over F*3*H*W elements -- at a pixel only the channel of its CFA colour differs from zero on both sides -- so
//   d loss / d out[f][c][p] = (2 / (3 F H W)) * (out[f][c][p] - mosaic[f][p])   where c is the CFA colour of p, else 0.
// RGGB: (even row, even col) R = 0, (even, odd) and (odd, even) G = 1, (odd, odd) B = 2 (DDnet_test.py:208-216).
__global__ void __launch_bounds__(256)
ddnet_loss_grad_kernel(const float* __restrict__ out, const float* __restrict__ mosaic, float* __restrict__ dout,
                       double* __restrict__ part, int H, int W, float scale) {
    __shared__ double red[16];
    const size_t HW = (size_t)H * W;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    double acc = 0.0;
    if (p < HW) {
        const int y = (int)(p / W), x = (int)(p % W);
        const int c = (y & 1) + (x & 1);
        const float d = out[((size_t)n * 3 + c) * HW + p] - mosaic[(size_t)n * HW + p];
#pragma unroll
        for (int k = 0; k < 3; ++k) dout[((size_t)n * 3 + k) * HW + p] = (k == c) ? scale * d : 0.f;
        acc = (double)(d * d);
    }
    const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = s;
}

// backward of ddnet_mix_kernel: d_s2[n][c] = a3[c] dout[n][c], d_s2[B + n][c] = a3[3 + c] dout[n][c]; partial sums of
// d a3[br][c] = sum dout[n][c] * s2[br B + n][c] in part[(br * 3 + c)][n * gridDim.x + block]
__global__ void __launch_bounds__(256)
ddnet_mix_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ s2, const float* __restrict__ a3,
                     float* __restrict__ d_s2, double* __restrict__ part, int B, size_t HW) {
    __shared__ double red[16];
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    if (p < HW) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = dout[((size_t)n * 3 + c) * HW + p];
            const size_t ia = ((size_t)n * 3 + c) * HW + p, ib = ((size_t)(B + n) * 3 + c) * HW + p;
            d_s2[ia] = a3[c] * g;
            d_s2[ib] = a3[3 + c] * g;
            acc[c] = (double)(g * s2[ia]);
            acc[3 + c] = (double)(g * s2[ib]);
        }
    }
    const size_t cols = (size_t)B * gridDim.x, col = (size_t)n * gridDim.x + blockIdx.x;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double s = block_sum_double(acc[k], red, threadIdx.x, blockDim.x);
        if (threadIdx.x == 0) part[(size_t)k * cols + col] = s;
        __syncthreads();
    }
}

// backward of ddnet_finish_kernel's `+ x_c8` path: the planar gradient [E][Cout][HW] as the 8-channel c8 gradient of the block's tail
__global__ void __launch_bounds__(256)
ddnet_finish_bwd_kernel(const float* __restrict__ d_out, float* __restrict__ d_x8, int Cout, size_t HW) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (p >= HW) return;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cout; ++c) v[c] = d_out[((size_t)e * Cout + c) * HW + p];
    float4* d = (float4*)(d_x8 + ((size_t)e * HW + p) * 8);
    d[0] = make_float4(v[0], v[1], v[2], v[3]);
    d[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// backward of ddnet_gather_kernel (+ the `in1 +` path of ddnet_finish_kernel): g = d t_in[e][i*C + c] (+ d_center for the centre
// frame i = 1: the gradient at the block's OUTPUT, summed over its Cd channels when the source has one channel);
//   gate gradients   part[(j*3 + i)*C + c][n*gridDim.x + block] = partial sum of g * src[idx[e][i]][c],  e = j*Bn + n
//   source gradients d_src[idx[e][i]][c] = g * scale  (every source frame is referenced once: the second stage)
template <int C>
__global__ void __launch_bounds__(256)
ddnet_gather_bwd_kernel(const float* __restrict__ d_tin, const float* __restrict__ d_center, int Cd, const float* __restrict__ src,
                        const int* __restrict__ idx, const float* __restrict__ scale, float* __restrict__ d_src,
                        double* __restrict__ part, int Bn, size_t HW) {
    __shared__ double red[16];
    constexpr int K = 3 * C, G = (K + 7) / 8;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    if (p < HW) {
        float g[G][8];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            const float4* s4 = (const float4*)(d_tin + (((size_t)e * G + gg) * HW + p) * 8);
            const float4 a = s4[0], b = s4[1];
            g[gg][0] = a.x; g[gg][1] = a.y; g[gg][2] = a.z; g[gg][3] = a.w;
            g[gg][4] = b.x; g[gg][5] = b.y; g[gg][6] = b.z; g[gg][7] = b.w;
        }
        float ctr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) ctr[c] = 0.f;
        if (d_center) {
            if (C == 1) {
                for (int cd = 0; cd < Cd; ++cd) ctr[0] = ctr[0] + d_center[((size_t)e * Cd + cd) * HW + p];
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) ctr[c] = d_center[((size_t)e * Cd + c) * HW + p];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const size_t fo = (size_t)idx[e * 3 + i] * C * HW + p;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int k = i * C + c;
                float gv = g[k / 8][k % 8];
                if (i == 1) gv = gv + ctr[c];
                if (part) acc[k] = (double)(gv * src[fo + (size_t)c * HW]);
                if (d_src) d_src[fo + (size_t)c * HW] = scale ? gv * scale[(e * 3 + i) * C + c] : gv;
            }
        }
    }
    if (part) {
        const int j = e / Bn, n = e - j * Bn;
        const size_t cols = (size_t)Bn * gridDim.x, col = (size_t)n * gridDim.x + blockIdx.x;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double s = block_sum_double(acc[k], red, threadIdx.x, blockDim.x);
            if (threadIdx.x == 0) part[((size_t)j * K + k) * cols + col] = s;
            __syncthreads();
        }
    }
}

// adjoint of bilinear_up2_kernel: d_in[e][c][y][x] = sum over the output pixels (Y, X) whose interpolation reads (y, x) of their
// weight times d_up[e][(Y, X)][c]; the weights are recomputed exactly as the forward kernel forms them
__global__ void __launch_bounds__(256)
bilinear_up2_bwd_kernel(const float* __restrict__ d_up, float* __restrict__ d_in, int h, int w) {
    const int H = 2 * h, W = 2 * w;
    const size_t hw = (size_t)h * w;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (p >= hw) return;
    const int y = (int)(p / w), x = (int)(p % w);
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const int Ylo = (sy > 0.f) ? max(0, (int)floorf((float)(y - 1) / sy) - 1) : 0;
    const int Yhi = (sy > 0.f) ? min(H - 1, (int)ceilf((float)(y + 1) / sy) + 1) : H - 1;
    const int Xlo = (sx > 0.f) ? max(0, (int)floorf((float)(x - 1) / sx) - 1) : 0;
    const int Xhi = (sx > 0.f) ? min(W - 1, (int)ceilf((float)(x + 1) / sx) + 1) : W - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int Y = Ylo; Y <= Yhi; ++Y) {
        const float ry = sy * (float)Y;
        const int y0 = min((int)ry, h - 1), y1 = min(y0 + 1, h - 1);
        const float ly1 = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), ly0 = 1.f - ly1;
        const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
        if (wy == 0.f) continue;
        for (int X = Xlo; X <= Xhi; ++X) {
            const float rx = sx * (float)X;
            const int x0 = min((int)rx, w - 1), x1 = min(x0 + 1, w - 1);
            const float lx1 = fminf(fmaxf(rx - (float)x0, 0.f), 1.f), lx0 = 1.f - lx1;
            const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
            if (wx == 0.f) continue;
            const float4 gq = *(const float4*)(d_up + ((size_t)e * H * W + (size_t)Y * W + X) * 8);
            const float wgt = wy * wx;
            acc[0] += wgt * gq.x; acc[1] += wgt * gq.y; acc[2] += wgt * gq.z; acc[3] += wgt * gq.w;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) d_in[((size_t)e * 4 + c) * hw + p] = acc[c];
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


/* ---- online finetune of the demosaicker (DDnet_test.py:248-296): loss gradient and the adjoints of the glue kernels above */
int scipnp_ddnet_loss_grad(const float* out, const float* mosaic, float* dout, double* loss_part, int H, int W, int B,
                           int* nblocks, scipnp_stream_t s) {
    SCIPNP_REQUIRE(nblocks && H > 0 && W > 0 && B > 0 && B <= 65535, "bad arguments");
    const size_t HW = (size_t)H * W;
    const unsigned gx = (unsigned)((HW + 255) / 256);
    *nblocks = (int)(gx * B);
    if (!out) return SCIPNP_OK;                                  // size query
    SCIPNP_REQUIRE(mosaic && dout && loss_part, "null pointer");
    const float scale = (float)(2.0 / (3.0 * (double)B * (double)HW));
    hipLaunchKernelGGL(ddnet_loss_grad_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)s, out, mosaic, dout, loss_part, H, W, scale);
    return launch_status("ddnet_loss_grad_kernel");
}

int scipnp_ddnet_mix_bwd(const float* dout, const float* branches, const float* gates, float* d_branches, double* part, int B,
                         int H, int W, int* ncols, scipnp_stream_t s) {
    SCIPNP_REQUIRE(ncols && B > 0 && B <= 65535 && H > 0 && W > 0, "bad arguments");
    const size_t HW = (size_t)H * W;
    const unsigned gx = (unsigned)((HW + 255) / 256);
    *ncols = (int)(gx * B);
    if (!dout) return SCIPNP_OK;                                 // size query: part holds 6 rows of *ncols doubles
    SCIPNP_REQUIRE(branches && gates && d_branches && part, "null pointer");
    hipLaunchKernelGGL(ddnet_mix_bwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)s, dout, branches, gates, d_branches, part, B, HW);
    return launch_status("ddnet_mix_bwd_kernel");
}

int scipnp_ddnet_finish_bwd(const float* d_out, float* d_x8, int E, int Cout, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(d_out && d_x8 && E > 0 && E <= 65535 && Cout >= 1 && Cout <= 8 && h > 0 && w > 0, "bad arguments");
    SCIPNP_ALIGNED(d_x8);
    const size_t HW = (size_t)h * w;
    hipLaunchKernelGGL(ddnet_finish_bwd_kernel, dim3((unsigned)((HW + 255) / 256), E), dim3(256), 0, (hipStream_t)s, d_out, d_x8, Cout, HW);
    return launch_status("ddnet_finish_bwd_kernel");
}

int scipnp_ddnet_gather_bwd(const float* d_tin_c8, const float* d_center, int Cd, const float* src, const int* idx,
                            const float* scale, float* d_src, double* part, int E, int Bn, int C, int h, int w, int* ncols,
                            scipnp_stream_t s) {
    SCIPNP_REQUIRE(ncols && E > 0 && E <= 65535 && Bn > 0 && E % Bn == 0 && h > 0 && w > 0, "bad arguments");
    SCIPNP_REQUIRE(C == 1 || C == 3 || C == 4, "channels per frame must be 1, 3 or 4 (got %d)", C);
    const size_t HW = (size_t)h * w;
    const unsigned gx = (unsigned)((HW + 255) / 256);
    *ncols = (int)(gx * Bn);
    if (!d_tin_c8) return SCIPNP_OK;                             // size query: part holds (E / Bn) * 3 * C rows of *ncols doubles
    SCIPNP_REQUIRE(src && idx && (d_src || part), "null pointer");
    SCIPNP_REQUIRE(!d_center || (Cd >= 1 && Cd <= 4 && (C == 1 || Cd == C)), "bad centre-gradient channel count %d", Cd);
    SCIPNP_ALIGNED(d_tin_c8);
    const dim3 grid(gx, E), block(256);
    hipStream_t st = (hipStream_t)s;
    if (C == 1) hipLaunchKernelGGL(ddnet_gather_bwd_kernel<1>, grid, block, 0, st, d_tin_c8, d_center, Cd, src, idx, scale, d_src, part, Bn, HW);
    else if (C == 3) hipLaunchKernelGGL(ddnet_gather_bwd_kernel<3>, grid, block, 0, st, d_tin_c8, d_center, Cd, src, idx, scale, d_src, part, Bn, HW);
    else hipLaunchKernelGGL(ddnet_gather_bwd_kernel<4>, grid, block, 0, st, d_tin_c8, d_center, Cd, src, idx, scale, d_src, part, Bn, HW);
    return launch_status("ddnet_gather_bwd_kernel");
}

int scipnp_bilinear_up2_bwd_c8(const float* d_up_c8, float* d_in, int E, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(d_up_c8 && d_in && E > 0 && E <= 65535 && h > 0 && w > 0, "bad arguments");
    SCIPNP_ALIGNED(d_up_c8);
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(bilinear_up2_bwd_kernel, dim3((unsigned)((hw + 255) / 256), E), dim3(256), 0, (hipStream_t)s, d_up_c8, d_in, h, w);
    return launch_status("bilinear_up2_bwd_kernel");
}

}  // extern "C"
