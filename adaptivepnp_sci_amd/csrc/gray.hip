// Grayscale (non-Bayer) PnP-ADMM mode, SURVEY 8(f) rank 4: the reference's one-stage loop
// (dvp_linear_inv_2_stage_ADMM_tensor_online.py:385-407, :500-509) with the Bayer split and the demosaic removed, the
// denoiser being Chambolle TV on the full frames or the model zoo's FFDNet-gray (model_zoo/ffdnet_gray.pth, hinted at
// two_stage_ADMM_Online_FFD_Warm.py:33-40).  The projection, the dual update and the PSNR partials are the plane-major
// kernels of sci_ops.hip (they are per pixel, any consistent layout works); this file holds the layout steps around
// the two priors.  With FFDNet-gray the state lives pixel-unshuffled, [B][4][M][N] -- the 2x2 pixel-unshuffle FFDNet
// starts with (network_ffdnet.py:60-62) IS that layout, so the network input is the state plus the sigma map.
#include "common.hpp"

namespace scipnp {

// in_c8[t][0][m][n][0..7] = { (x - b)[t][0..3][m][n], sigma, 0, 0, 0 }
__global__ void gray_net_input_kernel(const float* __restrict__ x, const float* __restrict__ b, float sigma,
                                      float* __restrict__ in_c8, size_t MN, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // (t, pixel)
    if (i >= total) return;
    const size_t t = i / MN, px = i - t * MN;
    const float* xs = x + t * 4 * MN + px;
    const float* bs = b + t * 4 * MN + px;
    float4* o = (float4*)(in_c8 + i * 8);
    o[0] = make_float4(xs[0] - bs[0], xs[MN] - bs[MN], xs[2 * MN] - bs[2 * MN], xs[3 * MN] - bs[3 * MN]);
    o[1] = make_float4(sigma, 0.f, 0.f, 0.f);
}

// theta_raw[t][ib][m][n] = out_c8[t][0][m][n][ib]
__global__ void gray_net_output_kernel(const float* __restrict__ out_c8, float* __restrict__ theta_raw, size_t MN, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t t = i / MN, px = i - t * MN;
    const float4 v = *(const float4*)(out_c8 + i * 8);
    float* o = theta_raw + t * 4 * MN + px;
    o[0] = v.x; o[MN] = v.y; o[2 * MN] = v.z; o[3 * MN] = v.w;
}

// (H,W,B) cube (frame index fastest) <-> [B][H][W] frames
template <bool TO_FRAMES>
__global__ void cube_frames_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, size_t HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // index into the [B][H][W] side
    if (i >= total) return;
    const size_t t = i / HW, px = i - t * HW;
    if (TO_FRAMES) dst[i] = src[px * B + t];
    else dst[px * B + t] = src[i];
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_gray_net_input(const float* x, const float* b, float sigma, float* in_c8, int M, int N, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && b && in_c8, "null pointer");
    SCIPNP_REQUIRE(M > 0 && N > 0 && B > 0, "bad shape");
    SCIPNP_ALIGNED(in_c8);
    const size_t MN = (size_t)M * N, total = MN * B;
    hipLaunchKernelGGL(gray_net_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, b, sigma,
                       in_c8, MN, total);
    return launch_status("gray_net_input_kernel");
}

int scipnp_gray_net_output(const float* out_c8, float* theta_raw, int M, int N, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(out_c8 && theta_raw, "null pointer");
    SCIPNP_REQUIRE(M > 0 && N > 0 && B > 0, "bad shape");
    SCIPNP_ALIGNED(out_c8);
    const size_t MN = (size_t)M * N, total = MN * B;
    hipLaunchKernelGGL(gray_net_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, out_c8,
                       theta_raw, MN, total);
    return launch_status("gray_net_output_kernel");
}

int scipnp_cube_to_frames(const float* cube, float* frames, int H, int W, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(cube && frames && H > 0 && W > 0 && B > 0, "null pointer or bad shape");
    const size_t HW = (size_t)H * W, total = HW * B;
    hipLaunchKernelGGL(cube_frames_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, cube,
                       frames, B, HW, total);
    return launch_status("cube_frames_kernel<true>");
}

int scipnp_frames_to_cube(const float* frames, float* cube, int H, int W, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(cube && frames && H > 0 && W > 0 && B > 0, "null pointer or bad shape");
    const size_t HW = (size_t)H * W, total = HW * B;
    hipLaunchKernelGGL(cube_frames_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, frames,
                       cube, B, HW, total);
    return launch_status("cube_frames_kernel<false>");
}

}  // extern "C"
