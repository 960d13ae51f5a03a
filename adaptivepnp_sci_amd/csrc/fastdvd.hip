// FastDVDnet glue kernels (the convolutions themselves are conv.hip): building the 3-frame DenBlock
// input from planar frames with CIRCULAR temporal indexing, and the residual `in1 - x` of a DenBlock.
//   reference: packages/fastdvdnet/models.py:179-198 (torch.cat of frames and noise maps, in1 - x),
//              packages/fastdvdnet/fastdvdnet.py:113-116 (circular window (frameidx + [0..4] - 2) mod N)
#include "common.hpp"

namespace scipnp {

// out c8 [B][2][H][W][8]: group 0 = (f0.rgb, sigma, f1.rgb, sigma), group 1 = (f2.rgb, sigma, 0,0,0,0)
// with f0,f1,f2 = frames (n-1, n, n+1) mod B of the planar input [B][3][H][W].
__global__ void __launch_bounds__(256)
fastdvd_pack_kernel(const float* __restrict__ frames, float* __restrict__ out, int B, size_t HW, float sigma) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= HW) return;
    const int f0 = (n + B - 1) % B, f2 = (n + 1) % B;
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

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_fastdvd_pack_triplets(const float* frames, float* out_c8, int B, int H, int W, float sigma,
                                 scipnp_stream_t s) {
    SCIPNP_REQUIRE(frames && out_c8 && B > 0 && B <= 65535 && H > 0 && W > 0, "bad arguments");
    SCIPNP_ALIGNED(out_c8);
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(fastdvd_pack_kernel, dim3((unsigned)((HW + 255) / 256), B), dim3(256), 0, (hipStream_t)s, frames,
                       out_c8, B, HW, sigma);
    return launch_status("fastdvd_pack_kernel");
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

}  // extern "C"
