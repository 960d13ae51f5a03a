// Final-report image quality metrics on the device, per frame of the mosaic cube held as plane-major states:
// squared error (PSNR) and structural similarity as scikit-image 0.18 computes them for the reference
//   dvp_linear_inv_2_stage_ADMM_tensor_online.py:316-321 / :542-547:
//     peak_signal_noise_ratio(X, x, data_range=1.)  -> mean((X - x)^2) with the squares formed in fp32, summed in fp64
//     structural_similarity(X, x, data_range=1.)    -> win x win uniform window, sample covariance, K1 = 0.01,
//                                                       K2 = 0.03, fp64, mean over the image minus a (win-1)/2 border
#include "common.hpp"

namespace scipnp {

__device__ __forceinline__ float mosaic_at(const float* __restrict__ st, int r, int c, int M, int N) {
    return st[((size_t)((r & 1) * 2 + (c & 1)) * M + (r >> 1)) * N + (c >> 1)];
}

__global__ void __launch_bounds__(256)
frame_metrics_kernel(const float* __restrict__ ref, const float* __restrict__ img, double* __restrict__ part,
                     int M, int N, int win, double C1, double C2) {
    __shared__ double red[16];
    const int H = 2 * M, W = 2 * N, t = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float* a = ref + (size_t)t * 4 * M * N;
    const float* b = img + (size_t)t * 4 * M * N;
    double se = 0.0, ss = 0.0;
    if (p < (size_t)H * W) {
        const int r = (int)(p / W), c = (int)(p % W);
        const float e = mosaic_at(a, r, c, M, N) - mosaic_at(b, r, c, M, N);
        se = (double)(e * e);
        const int pad = (win - 1) / 2;
        if (r >= pad && r < H - pad && c >= pad && c < W - pad) {
            double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
            for (int i = -pad; i <= pad; ++i)
                for (int j = -pad; j <= pad; ++j) {
                    const double x = (double)mosaic_at(a, r + i, c + j, M, N);
                    const double y = (double)mosaic_at(b, r + i, c + j, M, N);
                    sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
                }
            const double npx = (double)(win * win), cov = npx / (npx - 1.0);
            const double ux = sx / npx, uy = sy / npx;
            const double vx = cov * (sxx / npx - ux * ux), vy = cov * (syy / npx - uy * uy);
            const double vxy = cov * (sxy / npx - ux * uy);
            ss = ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
        }
    }
    const double s0 = block_sum_double(se, red, threadIdx.x, blockDim.x);
    __syncthreads();
    const double s1 = block_sum_double(ss, red, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) {
        double* o = part + ((size_t)t * gridDim.x + blockIdx.x) * 2;
        o[0] = s0;
        o[1] = s1;
    }
}

}  // namespace scipnp

using namespace scipnp;

extern "C" int scipnp_frame_metrics(const float* ref_state, const float* img_state, double* part, int M, int N, int B,
                                    int win, double data_range, int* nblocks, scipnp_stream_t s) {
    SCIPNP_REQUIRE(M > 0 && N > 0 && B > 0 && B <= 65535 && win >= 3 && (win & 1), "bad shape / window");
    SCIPNP_REQUIRE(2 * M >= win && 2 * N >= win, "win_size exceeds image extent");
    const size_t HW = (size_t)4 * M * N;
    const unsigned nb = (unsigned)((HW + 255) / 256);
    if (nblocks) *nblocks = (int)nb;
    if (!part) return 0;                                   // size query
    SCIPNP_REQUIRE(ref_state && img_state, "null pointer");
    const double C1 = (0.01 * data_range) * (0.01 * data_range), C2 = (0.03 * data_range) * (0.03 * data_range);
    hipLaunchKernelGGL(frame_metrics_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)s, ref_state, img_state, part, M, N, win,
                       C1, C2);
    return launch_status("frame_metrics_kernel");
}
