// Pre- and post-denoiser fusions on the plane-major state.
//
// pre : mosaic = x + inv_rho*b  ->  Malvar-2004 demosaic (reflect-101 border of the torch port)
//       ->  x_rgb, x_rgb - w/tau  ->  planar and/or pixel-unshuffled c8 denoiser input.
// post: denoised RGB (planar, or FFDNet tail in c8 before pixel-shuffle) -> theta at the CFA sites,
//       clip, dual updates b and w, optional SSE partials.
// Round 6: between the two kernels only the MOSAIC has to travel (scipnp_pm_pre_denoise_mosaic / scipnp_pm_post_denoise_mosaic: 4 E bytes
// written + read instead of the 12 E of x_rgb): the post kernel demosaicks its quad again from the stored mosaic with the same
// operations (malvar_quad), so w += x_rgb - out sees bit for bit the x_rgb the pre kernel fed to the denoiser.  (It cannot take the
// mosaic from x and b themselves: it updates b -- and x in the first iteration -- in place, under its neighbours' 5x5 windows.)
// One thread = one Bayer quad (2x2 mosaic pixels) of one frame; consecutive lanes = consecutive n,
// so every plane access is a coalesced row segment.  Built with -ffp-contract=off; the 5x5
// correlations use explicit fmaf chains in row-major tap order.
#include "common.hpp"

namespace scipnp {

// 5x5 tap tables, already divided by 8 (reference malvar2004.py:174-208).  Row-major, ky then kx.
__device__ constexpr float K_G[25] = {0, 0, -0.125f, 0, 0, 0, 0, 0.25f, 0, 0, -0.125f, 0.25f, 0.5f, 0.25f, -0.125f,
                                       0, 0, 0.25f, 0, 0, 0, 0, -0.125f, 0, 0};
__device__ constexpr float K_ROW[25] = {0, 0, 0.0625f, 0, 0, 0, -0.125f, 0, -0.125f, 0, -0.125f, 0.5f, 0.625f, 0.5f, -0.125f,
                                         0, -0.125f, 0, -0.125f, 0, 0, 0, 0.0625f, 0, 0};
__device__ constexpr float K_DIAG[25] = {0, 0, -0.1875f, 0, 0, 0, 0.25f, 0, 0.25f, 0, -0.1875f, 0, 0.75f, 0, -0.1875f,
                                          0, 0.25f, 0, 0.25f, 0, 0, 0, -0.1875f, 0, 0};

// v is the 6x6 mosaic window whose element [2][2] is the quad's (dy=0,dx=0) pixel
template <int OY, int OX, bool TRANSPOSE>
__device__ __forceinline__ float corr5(const float (&v)[6][6], const float (&k)[25]) {
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float tap = TRANSPOSE ? k[kx * 5 + ky] : k[ky * 5 + kx];
            if (tap != 0.f) acc = fmaf(tap, v[OY + ky][OX + kx], acc);
        }
    return acc;
}

// reflect-101 index of mosaic row/col `r` (may be -2..-1 or H..H+1) expressed in plane space:
// parity is preserved, so plane dy stays and only the plane row changes.
__device__ __forceinline__ int reflect_plane(int mm, int d, int M) {
    // mosaic index r = 2*mm + d
    int r = 2 * mm + d;
    const int H = 2 * M;
    if (r < 0) r = -r;
    if (r >= H) r = 2 * (H - 1) - r;
    return r >> 1;  // r keeps parity d
}

// Malvar-2004 on the quad whose (dy=0,dx=0) pixel is v[2][2]: rgb[c][dy][dx] (malvar2004.py:213-240)
__device__ __forceinline__ void malvar_quad(const float (&v)[6][6], float (&rgb)[3][2][2]) {
    // R site (0,0)
    rgb[0][0][0] = v[2][2];
    rgb[1][0][0] = corr5<0, 0, false>(v, K_G);
    rgb[2][0][0] = corr5<0, 0, false>(v, K_DIAG);
    // G1 site (0,1): red row, blue column
    rgb[0][0][1] = corr5<0, 1, false>(v, K_ROW);
    rgb[1][0][1] = v[2][3];
    rgb[2][0][1] = corr5<0, 1, true>(v, K_ROW);
    // G2 site (1,0): blue row, red column
    rgb[0][1][0] = corr5<1, 0, true>(v, K_ROW);
    rgb[1][1][0] = v[3][2];
    rgb[2][1][0] = corr5<1, 0, false>(v, K_ROW);
    // B site (1,1)
    rgb[0][1][1] = corr5<1, 1, false>(v, K_DIAG);
    rgb[1][1][1] = corr5<1, 1, false>(v, K_G);
    rgb[2][1][1] = v[3][3];
}

// the 6x6 mosaic window of quad (m, n) of one frame from its four planes (plane-major, reflect-101 border); bt != nullptr: the mosaic
// is x + inv_rho * b, formed here
__device__ __forceinline__ void load_window(const float* __restrict__ xt, const float* __restrict__ bt, float inv_rho, int M, int N,
                                            int m, int n, float (&v)[6][6]) {
    const size_t plane = (size_t)M * N;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d_y = i & 1;                 // window row i <-> mosaic row 2m-2+i: parity = i&1
        const int mm = reflect_plane(m - 1 + (i >> 1), d_y, M);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int d_x = j & 1;
            const int nn = reflect_plane(n - 1 + (j >> 1), d_x, N);
            const size_t o = (size_t)(d_y * 2 + d_x) * plane + (size_t)mm * N + nn;
            v[i][j] = bt ? (xt[o] + inv_rho * bt[o]) : xt[o];
        }
    }
}

// Which (column block, quad row m, frame t) this workgroup computes.  Workgroups are dealt round-robin over the 8 XCDs in dispatch
// order, so with the plain (blockIdx.x, y, z) mapping the quad rows m - 1, m, m + 1 a workgroup's 6x6 windows read belong to three
// different XCDs and every L2 fetches every row of x, b (or the mosaic) three times (PMC: 134 MB per pre-denoiser launch against
// 98 algorithmic).  Remapped, XCD k works through a CONTIGUOUS eighth of the (t, m, column block) order and finds its neighbours'
// rows in its own L2.  The logical indices are what everything below uses (the squared-error partials keep their order).
__device__ __forceinline__ void quad_block(int& bx, int& m, int& t) {
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    unsigned lin = (blockIdx.z * ny + blockIdx.y) * nx + blockIdx.x;
    if ((total & 7u) == 0u) lin = (lin & 7u) * (total >> 3) + (lin >> 3);
    bx = (int)(lin % nx);
    const unsigned r = lin / nx;
    m = (int)(r % ny);
    t = (int)(r / ny);
}

// shared tail of the two pre-denoiser kernels: store x_rgb, form x_rgb - inv_tau*w and emit it in the requested
// layouts (planar rgb_w, fp32 c8 with the pixel-unshuffle + sigma map, split-fp16 c8s)
__device__ __forceinline__ void emit_pre_outputs(const float (&rgb)[3][2][2], const float* __restrict__ w,
                                                 float* __restrict__ x_rgb, float* __restrict__ rgb_w,
                                                 float* __restrict__ net_in, char* __restrict__ net_in_s, int M, int N,
                                                 int m, int n, int t, float inv_tau, float sigma) {
    const size_t plane = (size_t)M * N;
    const int W = 2 * N;
    const size_t HW = 4 * plane;
    float in[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const size_t o = ((size_t)t * 3 + c) * HW + (size_t)(2 * m + dy) * W + 2 * n;
            if (x_rgb) *(float2*)(x_rgb + o) = make_float2(rgb[c][dy][0], rgb[c][dy][1]);
            float2 wv = make_float2(0.f, 0.f);
            if (w) wv = *(const float2*)(w + o);
            in[c][dy][0] = w ? (rgb[c][dy][0] - inv_tau * wv.x) : rgb[c][dy][0];
            in[c][dy][1] = w ? (rgb[c][dy][1] - inv_tau * wv.y) : rgb[c][dy][1];
            if (rgb_w) *(float2*)(rgb_w + o) = make_float2(in[c][dy][0], in[c][dy][1]);
        }
    if (net_in) {
        // c8 layout [t][2][M][N][8]; channel = c*4 + dy*2 + dx; 12 = sigma; 13..15 = 0
        float4* dst0 = (float4*)(net_in + (((size_t)t * 2 + 0) * plane + (size_t)m * N + n) * 8);
        float4* dst1 = (float4*)(net_in + (((size_t)t * 2 + 1) * plane + (size_t)m * N + n) * 8);
        dst0[0] = make_float4(in[0][0][0], in[0][0][1], in[0][1][0], in[0][1][1]);
        dst0[1] = make_float4(in[1][0][0], in[1][0][1], in[1][1][0], in[1][1][1]);
        dst1[0] = make_float4(in[2][0][0], in[2][0][1], in[2][1][0], in[2][1][1]);
        dst1[1] = make_float4(sigma, 0.f, 0.f, 0.f);
    }
    if (net_in_s) {
        // c8s layout [t][2 groups][2 planes (hi, lo')][M][N][8 fp16] for the split-fp16 convolutions
        const float g0[8] = {in[0][0][0], in[0][0][1], in[0][1][0], in[0][1][1], in[1][0][0], in[1][0][1], in[1][1][0], in[1][1][1]};
        const float g1[8] = {in[2][0][0], in[2][0][1], in[2][1][0], in[2][1][1], sigma, 0.f, 0.f, 0.f};
        const size_t pix = (size_t)m * N + n;
        char* base = net_in_s + (size_t)t * 2 * (2 * plane * 16);
        split8_store(g0, base + pix * 16, base + plane * 16 + pix * 16);
        split8_store(g1, base + 2 * plane * 16 + pix * 16, base + 3 * plane * 16 + pix * 16);
    }
}

__global__ void __launch_bounds__(256)
pm_pre_denoise_kernel(const float* __restrict__ x, const float* __restrict__ b,
                      const float* __restrict__ w, float* __restrict__ x_rgb, float* __restrict__ mosaic_out, float* __restrict__ rgb_w,
                      float* __restrict__ net_in, char* __restrict__ net_in_s, int M, int N, int B, float inv_rho,
                      float inv_tau, float sigma) {
    int bx, m, t;
    quad_block(bx, m, t);
    const int n = bx * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const size_t plane = (size_t)M * N;
    float v[6][6];
    load_window(x + (size_t)t * 4 * plane, b ? b + (size_t)t * 4 * plane : nullptr, inv_rho, M, N, m, n, v);
    if (mosaic_out) {                          // the quad's own four mosaic values, plane-major like x
        float* mo = mosaic_out + (size_t)t * 4 * plane + (size_t)m * N + n;
        mo[0] = v[2][2]; mo[plane] = v[2][3]; mo[2 * plane] = v[3][2]; mo[3 * plane] = v[3][3];
    }
    float rgb[3][2][2];
    malvar_quad(v, rgb);
    emit_pre_outputs(rgb, w, x_rgb, rgb_w, net_in, net_in_s, M, N, m, n, t, inv_tau, sigma);
}

// closed-form RGB update of the reference's `close_form_demosaic` branch (dvp...:175-182 / :224-230), k > 0:
//   x_rgb = (rho*x3 + b3 + tau*out_prev + w) / (rho*mask + tau)   [clipped to [0,1] on the FFDNet branch]
// with x3, b3 the Bayer planes scattered to their CFA sites (zero elsewhere) and mask the CFA site mask.
__global__ void __launch_bounds__(256)
pm_pre_closed_form_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ w,
                          const float* __restrict__ out_prev, float* __restrict__ x_rgb, float* __restrict__ rgb_w,
                          float* __restrict__ net_in, char* __restrict__ net_in_s, int M, int N, int B, float rho,
                          float tau, float inv_tau, int clip, float sigma) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    const int t = blockIdx.z;
    if (n >= N) return;
    const size_t plane = (size_t)M * N;
    const int W = 2 * N;
    const size_t HW = 4 * plane;
    float rgb[3][2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ib = dy * 2 + dx;
            const int site = (ib == 0) ? 0 : (ib == 3 ? 2 : 1);
            const size_t so = ((size_t)t * 4 + ib) * plane + (size_t)m * N + n;
            const float xs = x[so], bs = b[so];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t o = ((size_t)t * 3 + c) * HW + (size_t)(2 * m + dy) * W + 2 * n + dx;
                const float x3 = (c == site) ? xs : 0.f, b3 = (c == site) ? bs : 0.f;
                const float num = ((rho * x3 + b3) + tau * out_prev[o]) + w[o];
                const float den = ((c == site) ? rho : 0.f) + tau;
                float v = num / den;
                if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
                rgb[c][dy][dx] = v;
            }
        }
    emit_pre_outputs(rgb, w, x_rgb, rgb_w, net_in, net_in_s, M, N, m, n, t, inv_tau, sigma);
}

// x_rgb produced elsewhere (deep demosaicking, reference :192-194 / :242-244): only the `x_rgb - w/tau` fusion and
// the denoiser-input layouts of the two kernels above
__global__ void __launch_bounds__(256)
pm_pre_rgb_kernel(const float* __restrict__ w, float* __restrict__ x_rgb, float* __restrict__ rgb_w,
                  float* __restrict__ net_in, char* __restrict__ net_in_s, int M, int N, int B, float inv_tau, float sigma) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    const int t = blockIdx.z;
    if (n >= N) return;
    const int W = 2 * N;
    const size_t HW = (size_t)4 * M * N;
    float rgb[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const float2 q = *(const float2*)(x_rgb + ((size_t)t * 3 + c) * HW + (size_t)(2 * m + dy) * W + 2 * n);
            rgb[c][dy][0] = q.x; rgb[c][dy][1] = q.y;
        }
    emit_pre_outputs(rgb, w, x_rgb, rgb_w, net_in, net_in_s, M, N, m, n, t, inv_tau, sigma);
}

constexpr int POST_THREADS = 256;

__global__ void __launch_bounds__(POST_THREADS)
pm_post_denoise_kernel(const float* __restrict__ out_rgb, const float* __restrict__ out_c8,
                       float* __restrict__ out_store, float* __restrict__ x,
                       const float* __restrict__ x_rgb, const float* __restrict__ mosaic, float* __restrict__ theta, float* __restrict__ b,
                       float* __restrict__ w, const float* __restrict__ orig, double* sse_part,
                       int first_iter_alias, int M, int N, int B) {
    __shared__ double red[16];
    int bx, m, t;
    quad_block(bx, m, t);
    const int n = bx * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (n < N) {
        const size_t plane = (size_t)M * N;
        const int W = 2 * N;
        const size_t HW = 4 * plane;
        float o[3][2][2];
        if (out_c8) {
            const float4* s0 = (const float4*)(out_c8 + (((size_t)t * 2 + 0) * plane + (size_t)m * N + n) * 8);
            const float4* s1 = (const float4*)(out_c8 + (((size_t)t * 2 + 1) * plane + (size_t)m * N + n) * 8);
            const float4 r = s0[0], g = s0[1], bl = s1[0];
            o[0][0][0] = r.x; o[0][0][1] = r.y; o[0][1][0] = r.z; o[0][1][1] = r.w;
            o[1][0][0] = g.x; o[1][0][1] = g.y; o[1][1][0] = g.z; o[1][1][1] = g.w;
            o[2][0][0] = bl.x; o[2][0][1] = bl.y; o[2][1][0] = bl.z; o[2][1][1] = bl.w;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const float2 q = *(const float2*)(out_rgb + ((size_t)t * 3 + c) * HW + (size_t)(2 * m + dy) * W + 2 * n);
                    o[c][dy][0] = q.x; o[c][dy][1] = q.y;
                }
        }
        float xm[3][2][2];
        if (w && mosaic) {                     // x_rgb of this quad again, from the mosaic the pre kernel stored (same operations)
            float v[6][6];
            load_window(mosaic + (size_t)t * 4 * plane, nullptr, 0.f, M, N, m, n, v);
            malvar_quad(v, xm);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const size_t off = ((size_t)t * 3 + c) * HW + (size_t)(2 * m + dy) * W + 2 * n;
                if (out_store) *(float2*)(out_store + off) = make_float2(o[c][dy][0], o[c][dy][1]);
                if (w) {
                    const float2 xr = mosaic ? make_float2(xm[c][dy][0], xm[c][dy][1]) : *(const float2*)(x_rgb + off);
                    float2 wv = *(float2*)(w + off);
                    wv.x = wv.x + (xr.x - o[c][dy][0]);
                    wv.y = wv.y + (xr.y - o[c][dy][1]);
                    *(float2*)(w + off) = wv;
                }
            }
        // CFA sites: R (0,0), G1 (0,1), G2 (1,0), B (1,1)   -- dvp...:206-209
        const float raw[4] = {o[0][0][0], o[1][0][1], o[1][1][0], o[2][1][1]};
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const size_t off = ((size_t)t * 4 + ib) * plane + (size_t)m * N + n;
            const float th = fminf(fmaxf(raw[ib], 0.f), 1.f);
            float xe;
            if (first_iter_alias) { xe = raw[ib]; x[off] = xe; }   // x IS theta in the reference's first iteration
            else xe = x[off];
            b[off] = b[off] + (xe - th);
            theta[off] = th;
            if (sse_part) {
                const float e = orig[off] - th;
                acc += (double)(e * e);
            }
        }
    }
    if (sse_part) {
        const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
        if (threadIdx.x == 0)
            sse_part[((size_t)t * gridDim.y + m) * gridDim.x + bx] = s;
    }
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_pm_pre_denoise_ex(const float* x, const float* b, const float* w, float* x_rgb, float* rgb_w,
                             float* net_in_c8, void* net_in_c8s, int M, int N, int B, float inv_rho, float inv_tau,
                             float sigma, scipnp_stream_t s);

int scipnp_pm_pre_denoise(const float* x, const float* b, const float* w, float* x_rgb, float* rgb_w,
                          float* net_in_c8, int M, int N, int B, float inv_rho, float inv_tau, float sigma,
                          scipnp_stream_t s) {
    return scipnp_pm_pre_denoise_ex(x, b, w, x_rgb, rgb_w, net_in_c8, nullptr, M, N, B, inv_rho, inv_tau, sigma, s);
}

int scipnp_pm_pre_denoise_ex(const float* x, const float* b, const float* w, float* x_rgb, float* rgb_w,
                             float* net_in_c8, void* net_in_c8s, int M, int N, int B, float inv_rho, float inv_tau,
                             float sigma, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && x_rgb, "null pointer");
    SCIPNP_REQUIRE(M >= 2 && N >= 2 && B > 0 && B <= 65535 && M <= 65535, "bad shape M=%d N=%d B=%d", M, N, B);
    SCIPNP_ALIGNED(x_rgb);
    if (w) SCIPNP_ALIGNED(w);
    if (rgb_w) SCIPNP_ALIGNED(rgb_w);
    if (net_in_c8) SCIPNP_ALIGNED(net_in_c8);
    if (net_in_c8s) SCIPNP_ALIGNED(net_in_c8s);
    const int threads = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
    const dim3 grid((N + threads - 1) / threads, M, B);
    hipLaunchKernelGGL(pm_pre_denoise_kernel, grid, dim3(threads), 0, (hipStream_t)s, x, b, w, x_rgb, (float*)nullptr, rgb_w,
                       net_in_c8, (char*)net_in_c8s, M, N, B, inv_rho, inv_tau, sigma);
    return launch_status("pm_pre_denoise_kernel");
}

int scipnp_pm_pre_denoise_mosaic(const float* x, const float* b, const float* w, float* mosaic, float* rgb_w,
                                 float* net_in_c8, void* net_in_c8s, int M, int N, int B, float inv_rho, float inv_tau,
                                 float sigma, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && mosaic, "null pointer");
    SCIPNP_REQUIRE(mosaic != x && mosaic != b, "the mosaic buffer must not alias x or b (its neighbours are still being read)");
    SCIPNP_REQUIRE(rgb_w || net_in_c8 || net_in_c8s, "no denoiser-input layout requested");
    SCIPNP_REQUIRE(M >= 2 && N >= 2 && B > 0 && B <= 65535 && M <= 65535, "bad shape M=%d N=%d B=%d", M, N, B);
    if (w) SCIPNP_ALIGNED(w);
    if (rgb_w) SCIPNP_ALIGNED(rgb_w);
    if (net_in_c8) SCIPNP_ALIGNED(net_in_c8);
    if (net_in_c8s) SCIPNP_ALIGNED(net_in_c8s);
    const int threads = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
    const dim3 grid((N + threads - 1) / threads, M, B);
    hipLaunchKernelGGL(pm_pre_denoise_kernel, grid, dim3(threads), 0, (hipStream_t)s, x, b, w, (float*)nullptr, mosaic, rgb_w,
                       net_in_c8, (char*)net_in_c8s, M, N, B, inv_rho, inv_tau, sigma);
    return launch_status("pm_pre_denoise_kernel");
}

int scipnp_pm_pre_closed_form(const float* x, const float* b, const float* w, const float* out_prev, float* x_rgb,
                              float* rgb_w, float* net_in_c8, void* net_in_c8s, int M, int N, int B, float rho, float tau,
                              float inv_tau, int clip, float sigma, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && b && w && out_prev && x_rgb, "null pointer");
    SCIPNP_REQUIRE(M >= 1 && N >= 1 && B > 0 && B <= 65535 && M <= 65535, "bad shape M=%d N=%d B=%d", M, N, B);
    SCIPNP_ALIGNED(x_rgb); SCIPNP_ALIGNED(w);
    if (rgb_w) SCIPNP_ALIGNED(rgb_w);
    if (net_in_c8) SCIPNP_ALIGNED(net_in_c8);
    if (net_in_c8s) SCIPNP_ALIGNED(net_in_c8s);
    const int threads = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
    const dim3 grid((N + threads - 1) / threads, M, B);
    hipLaunchKernelGGL(pm_pre_closed_form_kernel, grid, dim3(threads), 0, (hipStream_t)s, x, b, w, out_prev, x_rgb, rgb_w,
                       net_in_c8, (char*)net_in_c8s, M, N, B, rho, tau, inv_tau, clip, sigma);
    return launch_status("pm_pre_closed_form_kernel");
}

int scipnp_pm_pre_rgb(const float* w, float* x_rgb, float* rgb_w, float* net_in_c8, void* net_in_c8s, int M, int N, int B,
                      float inv_tau, float sigma, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x_rgb && (rgb_w || net_in_c8 || net_in_c8s), "null pointer");
    SCIPNP_REQUIRE(M >= 1 && N >= 1 && B > 0 && B <= 65535 && M <= 65535, "bad shape M=%d N=%d B=%d", M, N, B);
    SCIPNP_ALIGNED(x_rgb);
    if (w) SCIPNP_ALIGNED(w);
    if (rgb_w) SCIPNP_ALIGNED(rgb_w);
    if (net_in_c8) SCIPNP_ALIGNED(net_in_c8);
    if (net_in_c8s) SCIPNP_ALIGNED(net_in_c8s);
    const int threads = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
    const dim3 grid((N + threads - 1) / threads, M, B);
    hipLaunchKernelGGL(pm_pre_rgb_kernel, grid, dim3(threads), 0, (hipStream_t)s, w, x_rgb, rgb_w, net_in_c8,
                       (char*)net_in_c8s, M, N, B, inv_tau, sigma);
    return launch_status("pm_pre_rgb_kernel");
}

int scipnp_pm_post_denoise(const float* out_rgb, const float* out_c8, float* out_rgb_store, float* x,
                           const float* x_rgb, float* theta, float* b, float* w, const float* orig,
                           double* sse_part, int first_iter_alias, int M, int N, int B, int* nblocks,
                           scipnp_stream_t s) {
    SCIPNP_REQUIRE((out_rgb != nullptr) != (out_c8 != nullptr), "exactly one of out_rgb / out_c8 must be given");
    SCIPNP_REQUIRE(x && theta && b, "null pointer");
    SCIPNP_REQUIRE((w == nullptr) || (x_rgb != nullptr), "w update needs x_rgb");
    SCIPNP_REQUIRE((sse_part == nullptr) || (orig != nullptr), "sse_part needs orig");
    SCIPNP_REQUIRE(M >= 1 && N >= 1 && B > 0 && B <= 65535 && M <= 65535, "bad shape");
    if (out_c8) SCIPNP_ALIGNED(out_c8);
    const int threads = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
    const dim3 grid((N + threads - 1) / threads, M, B);
    if (nblocks) *nblocks = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(pm_post_denoise_kernel, grid, dim3(threads), 0, (hipStream_t)s, out_rgb, out_c8,
                       out_rgb_store, x, x_rgb, (const float*)nullptr, theta, b, w, orig, sse_part, first_iter_alias, M, N, B);
    return launch_status("pm_post_denoise_kernel");
}

int scipnp_pm_post_denoise_mosaic(const float* out_rgb, const float* out_c8, float* out_rgb_store, float* x,
                                  const float* mosaic, float* theta, float* b, float* w, const float* orig,
                                  double* sse_part, int first_iter_alias, int M, int N, int B, int* nblocks,
                                  scipnp_stream_t s) {
    SCIPNP_REQUIRE((out_rgb != nullptr) != (out_c8 != nullptr), "exactly one of out_rgb / out_c8 must be given");
    SCIPNP_REQUIRE(x && theta && b && w && mosaic, "null pointer");
    SCIPNP_REQUIRE(mosaic != x && mosaic != b && mosaic != theta, "the mosaic buffer must not alias the state it helps to update");
    SCIPNP_REQUIRE((sse_part == nullptr) || (orig != nullptr), "sse_part needs orig");
    SCIPNP_REQUIRE(M >= 2 && N >= 2 && B > 0 && B <= 65535 && M <= 65535, "bad shape");
    if (out_c8) SCIPNP_ALIGNED(out_c8);
    const int threads = N >= 256 ? 256 : (N >= 128 ? 128 : 64);
    const dim3 grid((N + threads - 1) / threads, M, B);
    if (nblocks) *nblocks = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(pm_post_denoise_kernel, grid, dim3(threads), 0, (hipStream_t)s, out_rgb, out_c8,
                       out_rgb_store, x, (const float*)nullptr, mosaic, theta, b, w, orig, sse_part, first_iter_alias, M, N, B);
    return launch_status("pm_post_denoise_kernel");
}

}  // extern "C"
