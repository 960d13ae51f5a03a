// Layout kernels behind the stand-alone denoiser plug-ins (adaptivepnp_sci_amd/denoisers.py): the solver fuses these
// steps into its pre/post kernels, code written against the reference's plug-in API goes through here.
//   FFDNet.forward's input assembly   models/network_ffdnet.py:54-64  (replicate pad to even size, pixel-unshuffle,
//                                                                      sigma map as the last input channel)
//   FFDNet.forward's output assembly  models/network_ffdnet.py:66-69  (pixel-shuffle, crop)
//   test_ddnet's channel sum          packages/DDnet/DDnet_test.py:166-216 (the network sees the mosaic only)
#include "common.hpp"
#include <cstddef>

namespace scipnp {

// one thread per (n, group, pixel of the half-resolution grid): writes the 8 channels of its c8 group
__global__ void ffdnet_pack_input_kernel(const float* __restrict__ x, float sigma, float* __restrict__ out_c8, int C, int H, int W, int h, int w, int CG,
                                         size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int px = (int)(i % ((size_t)h * w));
    const size_t r = i / ((size_t)h * w);
    const int g = (int)(r % CG), n = (int)(r / CG);
    const int yy = px / w, xx = px - yy * w;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g * 8 + e;                       // unshuffled channel c*4 + dy*2 + dx, then sigma, then zeros
        float val = 0.f;
        if (ch < 4 * C) {
            const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
            const int sy = min(2 * yy + dy, H - 1), sx = min(2 * xx + dx, W - 1);        // replicate pad (:57-59)
            val = x[(((size_t)n * C + c) * H + sy) * W + sx];
        } else if (ch == 4 * C) {
            val = sigma;
        }
        v[e] = val;
    }
    float4* o = (float4*)(out_c8 + i * 8);
    o[0] = make_float4(v[0], v[1], v[2], v[3]);
    o[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// one thread per output pixel (n, c, y, x): y[n][c][2i+dy][2j+dx] = net_out[n][c*4 + dy*2 + dx][i][j]
__global__ void ffdnet_unpack_output_kernel(const float* __restrict__ out_c8, float* __restrict__ y, int C, int H, int W, int h,
                                            int w, int CG, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % W);
    size_t r = i / W;
    const int yy = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), n = (int)(r / C);
    const int ch = c * 4 + (yy & 1) * 2 + (xx & 1);
    y[i] = out_c8[((((size_t)n * CG + (ch >> 3)) * h + (yy >> 1)) * w + (xx >> 1)) * 8 + (ch & 7)];
}

// (H,W,3,B) -> (H,W,B): sum over the colour axis (one non-zero term per pixel for a CFA-sampled cube)
__global__ void cube_sum3_kernel(const float* __restrict__ cube, float* __restrict__ out, int B, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t p = i / B;
    const int t = (int)(i - p * B);
    const float* s = cube + p * 3 * B + t;
    out[i] = (s[0] + s[B]) + s[2 * B];                   // torch.sum(dim=2) order for 3 addends
}

// out = -in (the one-stage CNN branches run the shared pre / post kernels on -b: x + 1*(-b) = x - b exactly)
__global__ void negate_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -in[i];
}

// FastDVDnet finetune input (test_fastdvdnet.py:359 with utils_image.py:183-192): v + float32(float64(v) + noise)
__global__ void noisy_input_kernel(const float* __restrict__ v, const double* __restrict__ noise, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v[i] + (float)((double)v[i] + noise[i]);
}

// out[r] = sum of row r of a [rows][n] table of fp64 partial sums, sequential per lane then a fixed shuffle / LDS tree
// (one workgroup per row; deterministic)
__global__ void __launch_bounds__(256)
sum_rows_f64_kernel(const double* __restrict__ part, double* __restrict__ out, int n) {
    __shared__ double red[4];
    const double* p = part + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_ffdnet_pack_input(const float* x, float sigma, float* out_c8, int n, int C, int H, int W, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && out_c8, "null pointer");
    SCIPNP_REQUIRE(n > 0 && C > 0 && H > 0 && W > 0, "bad shape n=%d C=%d H=%d W=%d", n, C, H, W);
    SCIPNP_ALIGNED(out_c8);
    const int h = (H + 1) / 2, w = (W + 1) / 2, CG = (4 * C + 1 + 7) / 8;
    const size_t total = (size_t)n * CG * h * w;
    hipLaunchKernelGGL(ffdnet_pack_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, sigma,
                       out_c8, C, H, W, h, w, CG, total);
    return launch_status("ffdnet_pack_input_kernel");
}

int scipnp_ffdnet_unpack_output(const float* out_c8, float* y, int n, int C, int H, int W, scipnp_stream_t s) {
    SCIPNP_REQUIRE(out_c8 && y, "null pointer");
    SCIPNP_REQUIRE(n > 0 && C > 0 && H > 0 && W > 0, "bad shape n=%d C=%d H=%d W=%d", n, C, H, W);
    const int h = (H + 1) / 2, w = (W + 1) / 2, CG = (4 * C + 7) / 8;
    const size_t total = (size_t)n * C * H * W;
    hipLaunchKernelGGL(ffdnet_unpack_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, out_c8,
                       y, C, H, W, h, w, CG, total);
    return launch_status("ffdnet_unpack_output_kernel");
}

int scipnp_cube_sum3(const float* cube, float* out, int H, int W, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(cube && out, "null pointer");
    SCIPNP_REQUIRE(H > 0 && W > 0 && B > 0, "bad shape");
    const size_t total = (size_t)H * W * B;
    hipLaunchKernelGGL(cube_sum3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, cube, out, B,
                       total);
    return launch_status("cube_sum3_kernel");
}

int scipnp_negate(const float* in, float* out, size_t n, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && out && n > 0, "null pointer or empty");
    hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, in, out, n);
    return launch_status("negate_kernel");
}

int scipnp_fastdvd_noisy_input(const float* v, const double* noise, float* out, size_t n, scipnp_stream_t s) {
    SCIPNP_REQUIRE(v && noise && out && n > 0, "null pointer or empty");
    hipLaunchKernelGGL(noisy_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, v, noise, out, n);
    return launch_status("noisy_input_kernel");
}

int scipnp_sum_rows_f64(const double* part, double* out, int rows, int n, scipnp_stream_t s) {
    SCIPNP_REQUIRE(part && out && rows > 0 && n > 0, "null pointer or empty");
    hipLaunchKernelGGL(sum_rows_f64_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)s, part, out, n);
    return launch_status("sum_rows_f64_kernel");
}

}  // extern "C"
