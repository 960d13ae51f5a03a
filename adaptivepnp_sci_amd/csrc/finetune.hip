// Online measurement-loss finetune of the denoiser (reference packages/ffdnet/test_ffdnet_ipol.py:248-300):
// the pieces that are not a plain forward convolution.
//
//   loss / dLoss : L = mean_{m,n,ib} ( sum_t Phi * bayer_sample(pixel_shuffle(net_out)) - y )^2 and its
//                  gradient with respect to the network tail output (c8, before pixel-shuffle);
//   wgrad        : dW[co][ci][ky][kx] = sum_{n,y,x} dZ[n][co][y][x] * A[n][ci][y+ky-1][x+kx-1] as an MFMA
//                  GEMM with the PIXELS as the K dimension (v_mfma_f32_32x32x2_f32), persistent workgroups
//                  accumulating in registers over their share of pixel tiles, fp32 slabs + a fixed-order
//                  slab reduction (deterministic, no atomics);
//   bgrad        : db[co] = sum dZ;
//   adam         : torch.optim.Adam single-tensor update (betas (0.9,0.999), eps 1e-8, no weight decay);
//   pack (device): OIHW master weights -> packed c8 slabs for conv3x3_c8 (forward, or transposed + flipped
//                  for the backward-data convolution).
// Backward-data itself is conv3x3_c8 with the transposed/flipped weights and the ReLU mask epilogue (flag bit 4).
#include "common.hpp"

namespace scipnp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// --------------------------------------------------------------------------------------------- loss
// out_c8 [B][2][M][N][8]: channel c*4+dy*2+dx; CFA sites: R -> ch 0, G1 -> ch 5, G2 -> ch 6, B -> ch 11
__global__ void __launch_bounds__(256)
ffdnet_loss_grad_kernel(const float* __restrict__ out_c8, const float* __restrict__ Phi, const float* __restrict__ y,
                        float* __restrict__ gout_c8, double* __restrict__ loss_part, int M, int N, int B) {
    __shared__ double red[16];
    const size_t plane = (size_t)M * N;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (q < plane) {
        const int chan[4] = {0, 5, 6, 11};
        float up[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < B; ++t) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const float o = out_c8[(((size_t)t * 2 + (chan[ib] >> 3)) * plane + q) * 8 + (chan[ib] & 7)];
                up[ib] = up[ib] + o * Phi[((size_t)t * 4 + ib) * plane + q];
            }
        }
        const float norm = 2.0f / (float)(4 * plane);
        float g[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float d = up[ib] - y[(size_t)ib * plane + q];
            acc += (double)(d * d);
            g[ib] = norm * d;
        }
        for (int t = 0; t < B; ++t) {
            float4* d0 = (float4*)(gout_c8 + (((size_t)t * 2 + 0) * plane + q) * 8);
            float4* d1 = (float4*)(gout_c8 + (((size_t)t * 2 + 1) * plane + q) * 8);
            const float p0 = Phi[((size_t)t * 4 + 0) * plane + q], p1 = Phi[((size_t)t * 4 + 1) * plane + q];
            const float p2 = Phi[((size_t)t * 4 + 2) * plane + q], p3 = Phi[((size_t)t * 4 + 3) * plane + q];
            d0[0] = make_float4(g[0] * p0, 0.f, 0.f, 0.f);
            d0[1] = make_float4(0.f, g[1] * p1, g[2] * p2, 0.f);
            d1[0] = make_float4(0.f, 0.f, 0.f, g[3] * p3);
            d1[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = s;
}

// --------------------------------------------------------------------------------------------- wgrad
constexpr int WG_TW = 32, WG_TR = 2;                 // pixel tile: 2 rows x 32 columns = 64 pixels = 32 k-steps
constexpr int WG_PX = WG_TW * WG_TR;
constexpr int WG_DPITCH = WG_PX + 1;                 // per channel-group pitch (in pixels) of the dZ tile
constexpr int WG_AW = WG_TW + 2, WG_AR = WG_TR + 2;
constexpr int WG_APITCH = WG_AW * WG_AR + 1;         // 137
constexpr int WG_THREADS = 9 * 64;                   // wave t <-> tap t

// grid = (nslab, Cin/32 blocks).  act: [n][CGin][h][w][8], dz: [n][CGout][h][w][8];
// slab layout: slabs[slab][tap][coP][ciP]  (coP = 32*COB, ciP = 32*gridDim.y)
template <int COB>
__global__ void __launch_bounds__(WG_THREADS)
conv3x3_wgrad_kernel(const float* __restrict__ act, const float* __restrict__ dz, float* __restrict__ slabs,
                     int n_img, int CGin, int CGout, int cg0 /* first output channel group of this launch */,
                     int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzs = smem;                                            // [4*COB][WG_DPITCH][8]
    float* as = smem + 4 * COB * WG_DPITCH * 8;                   // [4][WG_APITCH][8]
    const int tid = threadIdx.x;
    const int lane = tid & 63, tap = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int cib = blockIdx.y;
    const size_t HW = (size_t)H * W;
    const int tiles_x = (W + WG_TW - 1) / WG_TW, tiles_y = (H + WG_TR - 1) / WG_TR;
    const int tiles = n_img * tiles_y * tiles_x;

    f32x16 acc[COB];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

    const int a_lane = ((li >> 3) * WG_DPITCH) * 8 + (li & 7);    // + (cb*4*DPITCH + px)*8
    const int b_lane = ((li >> 3) * WG_APITCH) * 8 + (li & 7);    // + pos*8

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * WG_TW, y0 = ty * WG_TR;
        __syncthreads();   // previous tile's reads done
        // dZ tile: all COB*4 channel groups x 64 pixels x 8
        for (int e = tid; e < 4 * COB * WG_PX * 2; e += WG_THREADS) {
            const int half = e & 1, px = (e >> 1) % WG_PX, cgl = (e >> 1) / WG_PX;
            const int cg = cg0 + cgl;
            const int gy = y0 + px / WG_TW, gx = x0 + px % WG_TW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cg < CGout && gy < H && gx < W)
                v = *(const float4*)(dz + (((size_t)n * CGout + cg) * HW + (size_t)gy * W + gx) * 8 + 4 * half);
            *(float4*)(dzs + (cgl * WG_DPITCH + px) * 8 + 4 * half) = v;
        }
        // activation tile with halo: the 4 channel groups of this ci block
        for (int e = tid; e < 4 * WG_AW * WG_AR * 2; e += WG_THREADS) {
            const int half = e & 1, pos = (e >> 1) % (WG_AW * WG_AR), cgl = (e >> 1) / (WG_AW * WG_AR);
            const int cg = cib * 4 + cgl;
            const int gy = y0 - 1 + pos / WG_AW, gx = x0 - 1 + pos % WG_AW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cg < CGin && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *(const float4*)(act + (((size_t)n * CGin + cg) * HW + (size_t)gy * W + gx) * 8 + 4 * half);
            *(float4*)(as + (cgl * WG_APITCH + pos) * 8 + 4 * half) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < WG_PX / 2; ++s) {
            const int px = 2 * s + lh;
            const int r = px / WG_TW, c = px % WG_TW;
            const float bv = as[b_lane + ((r + ky) * WG_AW + c + kx) * 8];
#pragma unroll
            for (int cb = 0; cb < COB; ++cb) {
                const float av = dzs[a_lane + (cb * 4 * WG_DPITCH + px) * 8];
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[cb], 0, 0, 0);
            }
        }
    }
    // C[row = co_local][col = ci_local]: row = (r&3) + 8*(r>>2) + 4*lh, col = li
    const int coP = 32 * COB, ciP = 32 * gridDim.y;
    float* slab = slabs + ((size_t)blockIdx.x * 9 + tap) * coP * ciP;
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[(size_t)co * ciP + cib * 32 + li] = acc[cb][r];
        }
}

// dW[co][ci][tap] (OIHW, real channel counts) = sum over slabs in fixed order
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dW, int Cin_real, int co0,
                    int co_count /* real output channels of this chunk */, int coP, int ciP) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = co_count * Cin_real * 9;
    if (idx >= total) return;
    const int ci = idx % Cin_real, co = (idx / Cin_real) % co_count, tap = idx / (Cin_real * co_count);   // coalesced slab reads
    const size_t stride = (size_t)9 * coP * ciP;
    const float* p = slabs + ((size_t)tap * coP + co) * ciP + ci;
    // four interleaved partial sums (slab k -> accumulator k % 4), then ((s0+s1)+s2)+s3: a fixed order, and four loads in
    // flight per lane instead of a chain of nslab dependent ones
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nslab; k += 4) {
        s0 += p[(size_t)k * stride];
        s1 += p[(size_t)(k + 1) * stride];
        s2 += p[(size_t)(k + 2) * stride];
        s3 += p[(size_t)(k + 3) * stride];
    }
    for (; k < nslab; ++k) s0 += p[(size_t)k * stride];
    const float s = ((s0 + s1) + s2) + s3;
    dW[((size_t)(co0 + co) * Cin_real + ci) * 9 + tap] = s;
}

// db[co] = sum_{n,y,x} dz[n][co/8][y][x][co%8]: one block per (channel group, image chunk) -> partials, then reduce
__global__ void __launch_bounds__(256)
bgrad_partial_kernel(const float* __restrict__ dz, float* __restrict__ part, int n_img, int CG, size_t HW, int nchunk) {
    __shared__ float red[256 * 8 / 8];
    const int cg = blockIdx.y, chunk = blockIdx.x;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t total = (size_t)n_img * HW;
    for (size_t i = (size_t)chunk * blockDim.x + threadIdx.x; i < total; i += (size_t)nchunk * blockDim.x) {
        const size_t n = i / HW, p = i - n * HW;
        const float4* s = (const float4*)(dz + ((n * CG + cg) * HW + p) * 8);
        const float4 a = s[0], b = s[1];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    }
    // wave reduce then 4 waves through LDS
    for (int c = 0; c < 8; ++c)
        for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_down(acc[c], off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int c = 0; c < 8; ++c) red[wave * 8 + c] = acc[c];
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += red[w * 8 + threadIdx.x];
        part[((size_t)cg * nchunk + chunk) * 8 + threadIdx.x] = s;
    }
}

__device__ __forceinline__ void bgrad_reduce_body(const float* __restrict__ part, float* __restrict__ db, int Cout_real, int nchunk, int co) {
    if (co >= Cout_real) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[((size_t)(co >> 3) * nchunk + k) * 8 + (co & 7)];
    db[co] = s;
}
__global__ void bgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ db, int Cout_real, int nchunk) {
    bgrad_reduce_body(part, db, Cout_real, nchunk, blockIdx.x * blockDim.x + threadIdx.x);
}
// ... of MANY layers in one launch (blockIdx.y = the layer)
constexpr int BGRAD_MULTI_MAX = 32;
struct BgradJobs {
    const float* part[BGRAD_MULTI_MAX];
    float* db[BGRAD_MULTI_MAX];
    int cout_real[BGRAD_MULTI_MAX];
};
__global__ void bgrad_reduce_multi_kernel(const BgradJobs j, int nchunk) {
    const int q = blockIdx.y;
    bgrad_reduce_body(j.part[q], j.db[q], j.cout_real[q], nchunk, blockIdx.x * blockDim.x + threadIdx.x);
}

// --------------------------------------------------------------------------------------------- Adam
// torch.optim.Adam._single_tensor_adam (foreach=False, amsgrad=False, weight_decay=0, maximize=False)
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
            float one_minus_b1, float b2, float one_minus_b2, float neg_step_size, float bc2_sqrt, float eps) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + one_minus_b1 * (gi - m[i]);            // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + (one_minus_b2 * gi) * gi;          // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] + (neg_step_size * mi) / denom;                     // param.addcdiv_(exp_avg, denom, value=-step_size)
    m[i] = mi;
    v[i] = vi;
}

// --------------------------------------------------------------------------------------------- device pack
// forward : packed[cig][tap][coP][8] = W[co][ci][tap]                       (+ bias[coP] appended)
// backward: the conv that maps dZ (Cout ch) to dA (Cin ch):  W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]
//           -> packed'[cog][tap][ciP][8], no bias
__device__ __forceinline__ void pack_device_body(const float* __restrict__ w, const float* __restrict__ bias,
                                                 const float* __restrict__ scale, float* __restrict__ packed, int Cin_real,
                                                 int Cout_real, int Kin /*padded in-ch of the packed conv*/, int KoutP, int transpose,
                                                 size_t i) {
    const size_t nw = (size_t)(Kin / 8) * 9 * KoutP * 8;
    if (i >= nw + KoutP) return;
    if (i >= nw) {
        const int o = (int)(i - nw);
        packed[i] = (!transpose && bias && o < Cout_real) ? bias[o] : 0.f;
        return;
    }
    const int c8 = i & 7;
    const int o = (int)((i >> 3) % KoutP);
    const int tap = (int)((i / ((size_t)8 * KoutP)) % 9);
    const int ig = (int)(i / ((size_t)8 * KoutP * 9));
    const int in_ch = ig * 8 + c8;
    float val = 0.f;
    if (!transpose) {
        if (o < Cout_real && in_ch < Cin_real) {
            val = w[((size_t)o * Cin_real + in_ch) * 9 + tap];
            if (scale) val = val * scale[o];
        }
    } else {
        // packed conv: input channel = original co (in_ch), output channel = original ci (o), flipped tap
        if (o < Cin_real && in_ch < Cout_real) {
            val = w[((size_t)in_ch * Cin_real + o) * 9 + (8 - tap)];
            if (scale) val = val * scale[in_ch];
        }
    }
    packed[i] = val;
}

__global__ void __launch_bounds__(256)
pack_device_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ scale,
                   float* __restrict__ packed, int Cin_real, int Cout_real, int Kin, int KoutP, int transpose) {
    pack_device_body(w, bias, scale, packed, Cin_real, Cout_real, Kin, KoutP, transpose, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// the packs of MANY layers in one launch (round 5: a trainer repacks every layer in both directions after each Adam step -- 46
// launches of 4 us for FFDNet, each a dependent launch of its own; blockIdx.y = the job)
constexpr int PACK_MULTI_MAX = 32;
struct PackDeviceJobs {
    const float* w[PACK_MULTI_MAX];
    const float* bias[PACK_MULTI_MAX];
    const float* scale[PACK_MULTI_MAX];
    float* packed[PACK_MULTI_MAX];
    int cin_real[PACK_MULTI_MAX], cout_real[PACK_MULTI_MAX], kin[PACK_MULTI_MAX], koutp[PACK_MULTI_MAX], transpose[PACK_MULTI_MAX];
};
__global__ void __launch_bounds__(256) pack_device_multi_kernel(const PackDeviceJobs j) {
    const int q = blockIdx.y;
    pack_device_body(j.w[q], j.bias[q], j.scale[q], j.packed[q], j.cin_real[q], j.cout_real[q], j.kin[q], j.koutp[q], j.transpose[q],
                     (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - mean[c] * sc;
}

// one wave per output channel
__global__ void __launch_bounds__(64)
bn_fold_grads_kernel(const float* __restrict__ W, const float* __restrict__ G, const float* __restrict__ sdy,
                     const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ var,
                     float eps, float* __restrict__ dW, float* __restrict__ dgamma, float* __restrict__ dbeta, int K) {
    const int co = blockIdx.x;
    const float sd = sqrtf(var[co] + eps);
    const float sc = gamma[co] / sd;
    float dot = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) {
        const float g = G[(size_t)co * K + k];
        dot += W[(size_t)co * K + k] * g;
        dW[(size_t)co * K + k] = sc * g;
    }
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
    if (threadIdx.x == 0) {
        dgamma[co] = (dot - mean[co] * sdy[co]) / sd;
        dbeta[co] = sdy[co];
    }
}

static inline int round_up_i(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_pack_conv3x3_device_multi(int n, const float* const* w, const float* const* bias, const float* const* scale,
                                     float* const* packed, const int* Cin_real, const int* Cout_real, const int* Cin, const int* Cout,
                                     const int* transpose_flip, scipnp_stream_t s) {
    SCIPNP_REQUIRE(n >= 0 && (n == 0 || (w && packed && Cin_real && Cout_real && Cin && Cout && transpose_flip)), "bad arguments");
    for (int base = 0; base < n; base += PACK_MULTI_MAX) {
        const int m = n - base < PACK_MULTI_MAX ? n - base : PACK_MULTI_MAX;
        PackDeviceJobs j = {};
        size_t most = 0;
        for (int q = 0; q < m; ++q) {
            const int g = base + q;
            SCIPNP_REQUIRE(w[g] && packed[g] && Cin[g] % 8 == 0 && Cout[g] % 8 == 0 && Cin_real[g] <= Cin[g] && Cout_real[g] <= Cout[g],
                           "bad arguments in job %d", g);
            j.w[q] = w[g]; j.bias[q] = bias ? bias[g] : nullptr; j.scale[q] = scale ? scale[g] : nullptr; j.packed[q] = packed[g];
            j.cin_real[q] = Cin_real[g]; j.cout_real[q] = Cout_real[g]; j.transpose[q] = transpose_flip[g];
            j.kin[q] = transpose_flip[g] ? Cout[g] : Cin[g];
            j.koutp[q] = round_up_i(transpose_flip[g] ? Cin[g] : Cout[g], 32);
            const size_t total = (size_t)(j.kin[q] / 8) * 9 * j.koutp[q] * 8 + j.koutp[q];
            most = total > most ? total : most;
        }
        hipLaunchKernelGGL(pack_device_multi_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)m), dim3(256), 0, (hipStream_t)s, j);
    }
    return launch_status("pack_device_multi_kernel");
}

int scipnp_pack_conv3x3_device_scaled(const float* w, const float* bias, const float* scale, float* packed,
                                      int Cin_real, int Cout_real, int Cin, int Cout, int transpose_flip,
                                      scipnp_stream_t s);

int scipnp_ffdnet_loss_grad(const float* out_c8, const float* Phi, const float* y, float* gout_c8, double* loss_part,
                            int M, int N, int B, int* nblocks, scipnp_stream_t s) {
    SCIPNP_REQUIRE(nblocks && M > 0 && N > 0 && B > 0, "bad arguments");
    const size_t plane = (size_t)M * N;
    const unsigned blocks = (unsigned)((plane + 255) / 256);
    *nblocks = (int)blocks;
    if (loss_part == nullptr) return SCIPNP_OK;   // size query
    SCIPNP_REQUIRE(out_c8 && Phi && y && gout_c8, "null pointer");
    SCIPNP_ALIGNED(gout_c8);
    hipLaunchKernelGGL(ffdnet_loss_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out_c8, Phi, y, gout_c8,
                       loss_part, M, N, B);
    return launch_status("ffdnet_loss_grad_kernel");
}

size_t scipnp_conv3x3_wgrad_workspace_floats(int Cin, int Cout, int nslab) {
    if (Cin <= 0 || Cout <= 0 || nslab <= 0) return 0;
    const int chunk = round_up_i(Cout, 32) < 96 ? round_up_i(Cout, 32) : 96;
    return (size_t)nslab * 9 * chunk * round_up_i(Cin, 32);
}

int scipnp_conv3x3_wgrad(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                         int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(act_c8 && dz_c8 && dW && workspace, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin_real <= Cin && Cout_real <= Cout &&
                   nslab > 0 && nslab <= 65535, "bad shape");
    SCIPNP_ALIGNED(act_c8); SCIPNP_ALIGNED(dz_c8);
    const int ciP = round_up_i(Cin, 32);
    hipStream_t st = (hipStream_t)s;
    // output channels in chunks of <= 96 (3 MFMA row blocks per wave); one slab set + reduction per chunk
    for (int co0 = 0; co0 < Cout_real; co0 += 96) {
        const int left = round_up_i(Cout, 32) - co0;
        const int coP = left < 96 ? left : 96;
        const int COB = coP / 32;
        const dim3 grid(nslab, ciP / 32);
        const size_t lds = (size_t)(4 * COB * WG_DPITCH + 4 * WG_APITCH) * 8 * sizeof(float);
#define SCIPNP_GO(C)                                                                                           \
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<C>), grid, dim3(WG_THREADS), lds, st, act_c8, dz_c8, workspace, n, \
                       Cin / 8, Cout / 8, co0 / 8, h, w)
        if (COB == 1) SCIPNP_GO(1); else if (COB == 2) SCIPNP_GO(2); else SCIPNP_GO(3);
#undef SCIPNP_GO
        int rc = launch_status("conv3x3_wgrad_kernel");
        if (rc) return rc;
        const int co_count = (Cout_real - co0) < coP ? (Cout_real - co0) : coP;
        const int total = co_count * Cin_real * 9;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, nslab, dW,
                           Cin_real, co0, co_count, coP, ciP);
        rc = launch_status("wgrad_reduce_kernel");
        if (rc) return rc;
    }
    return SCIPNP_OK;
}

int scipnp_conv_bias_grad(const float* dz_c8, float* db, float* workspace, int n, int Cout_real, int Cout, int h, int w,
                          scipnp_stream_t s) {
    SCIPNP_REQUIRE(dz_c8 && workspace && n > 0 && Cout % 8 == 0 && Cout_real <= Cout, "bad arguments");   // (db == NULL: the partial
    SCIPNP_ALIGNED(dz_c8);                                                                              // sums only, see ..._reduce_multi)
    const int nchunk = 64;   // workspace: (Cout/8) * 64 * 8 floats
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(bgrad_partial_kernel, dim3(nchunk, Cout / 8), dim3(256), 0, st, dz_c8, workspace, n, Cout / 8,
                       (size_t)h * w, nchunk);
    if (db) hipLaunchKernelGGL(bgrad_reduce_kernel, dim3((Cout_real + 63) / 64), dim3(64), 0, st, workspace, db, Cout_real, nchunk);
    return launch_status("bgrad kernels");
}

int scipnp_conv_bias_grad_reduce_multi(int n, const float* const* workspace, float* const* db, const int* Cout_real, scipnp_stream_t s) {
    SCIPNP_REQUIRE(n >= 0 && (n == 0 || (workspace && db && Cout_real)), "bad arguments");
    for (int base = 0; base < n; base += BGRAD_MULTI_MAX) {
        const int m = n - base < BGRAD_MULTI_MAX ? n - base : BGRAD_MULTI_MAX;
        BgradJobs j = {};
        int most = 0;
        for (int q = 0; q < m; ++q) {
            SCIPNP_REQUIRE(workspace[base + q] && db[base + q] && Cout_real[base + q] > 0, "bad arguments in job %d", base + q);
            j.part[q] = workspace[base + q]; j.db[q] = db[base + q]; j.cout_real[q] = Cout_real[base + q];
            most = Cout_real[base + q] > most ? Cout_real[base + q] : most;
        }
        hipLaunchKernelGGL(bgrad_reduce_multi_kernel, dim3((unsigned)((most + 63) / 64), (unsigned)m), dim3(64), 0, (hipStream_t)s, j, 64);
    }
    return launch_status("bgrad_reduce_multi_kernel");
}

int scipnp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                     double beta1, double beta2, double eps, int step, scipnp_stream_t s) {
    SCIPNP_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1 && n > 0, "bad arguments");
    // lr, betas and eps are Python doubles in torch: the step-dependent scalars are evaluated in double and
    // reach the element-wise float32 arithmetic as float32, exactly like torch's Scalar arguments
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1;
    const double bc2_sqrt = sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg,
                       exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(-step_size),
                       (float)bc2_sqrt, (float)eps);
    return launch_status("adam_kernel");
}

int scipnp_pack_conv3x3_device(const float* w, const float* bias, float* packed, int Cin_real, int Cout_real, int Cin,
                               int Cout, int transpose_flip, scipnp_stream_t s) {
    return scipnp_pack_conv3x3_device_scaled(w, bias, nullptr, packed, Cin_real, Cout_real, Cin, Cout, transpose_flip, s);
}

int scipnp_pack_conv3x3_device_scaled(const float* w, const float* bias, const float* scale, float* packed,
                                      int Cin_real, int Cout_real, int Cin, int Cout, int transpose_flip,
                                      scipnp_stream_t s) {
    SCIPNP_REQUIRE(w && packed && Cin % 8 == 0 && Cout % 8 == 0 && Cin_real <= Cin && Cout_real <= Cout, "bad arguments");
    // forward: packed conv has Cin inputs, Cout outputs; backward-data: Cout inputs, Cin outputs
    const int Kin = transpose_flip ? Cout : Cin;
    const int KoutP = round_up_i(transpose_flip ? Cin : Cout, 32);
    const size_t total = (size_t)(Kin / 8) * 9 * KoutP * 8 + KoutP;
    hipLaunchKernelGGL(pack_device_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, bias,
                       scale, packed, Cin_real, Cout_real, Kin, KoutP, transpose_flip);
    return launch_status("pack_device_kernel");
}

// ---- eval-mode BatchNorm folded into the preceding bias-free conv:  y = conv(x; W)*s + t,
//      s = gamma/sqrt(var+eps), t = beta - mean*s.  Given G = wgrad(x, dy) and sdy = sum dy:
//      dW = s*G,  dgamma = (<W,G> - mean*sdy)/sqrt(var+eps),  dbeta = sdy.
int scipnp_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                   float* shift, int C, scipnp_stream_t s) {
    SCIPNP_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, gamma, beta, mean, var, eps, scale,
                       shift, C);
    return launch_status("bn_fold_kernel");
}

int scipnp_bn_fold_grads(const float* W, const float* G, const float* sdy, const float* gamma, const float* mean,
                         const float* var, float eps, float* dW, float* dgamma, float* dbeta, int Cout, int K,
                         scipnp_stream_t s) {
    SCIPNP_REQUIRE(W && G && sdy && gamma && mean && var && dW && dgamma && dbeta && Cout > 0 && K > 0, "bad arguments");
    hipLaunchKernelGGL(bn_fold_grads_kernel, dim3(Cout), dim3(64), 0, (hipStream_t)s, W, G, sdy, gamma, mean, var, eps, dW,
                       dgamma, dbeta, K);
    return launch_status("bn_fold_grads_kernel");
}

}  // extern "C"
