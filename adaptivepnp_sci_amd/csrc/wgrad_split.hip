// Weight / bias gradients of the online finetune on the fp16 MFMA with error-compensated operands
// (reference packages/ffdnet/test_ffdnet_ipol.py:296 `loss.backward()`):
//   dW[co][ci][ky][kx] = sum_{n,y,x} dZ[n][co][y][x] * A[n][ci][y+ky-1][x+kx-1]
// as a GEMM with the PIXELS as the K dimension on v_mfma_f32_32x32x16_f16.  Both operands are c8s tensors
// ([n][C/8][hi, lo'][h*w][8 fp16], conv_split.hip): the tiles go to LDS exactly as they lie in memory ([pixel][8 ch]
// rows of 16 bytes) and the MFMA operands, which need 8 consecutive PIXELS of one channel per lane, are fetched with
// the gfx950 transposing LDS read ds_read_b64_tr_b16 (4 pixel rows x 16 channels per 16-lane group, delivered
// channel-major) -- no transposing store, and the 3x3 tap is just a row offset of the activation tile.
//   value = hi*hi + (hi*lo' + lo'*hi)/2048  in two fp32 accumulators per tile (the lo'*lo' term is dropped);
//   dZ arrives pre-scaled by a power of two (fp16 range), the slab reduction un-scales exactly.
// Workgroup = 9 waves (wave t <-> tap t) x 32*COB output channels x 32 input channels, persistent over its share of
// 2x32-pixel tiles; next tile prefetched into registers while the current one is consumed; fp32 slabs + fixed-order
// reduction (deterministic, no atomics) as in the fp32 kernel (finetune.hip).
#include "common.hpp"

namespace scipnp {

typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4w __attribute__((ext_vector_type(4)));

constexpr int WS_TW = 32, WS_TR = 2, WS_PX = WS_TW * WS_TR;      // 64 pixels = 4 K-steps of 16
constexpr int WS_AW = WS_TW + 2, WS_AR = WS_TR + 2, WS_APX = WS_AW * WS_AR;   // 34 x 4 = 136
constexpr int WS_GD = WS_PX * 16 + 64;                           // dZ channel-group stride: 1088 B = 64 (mod 256)
constexpr int WS_GA = 2368;                                      // act channel-group stride >= 136*16, = 64 (mod 256)
constexpr int WS_THREADS = 9 * 64;
static_assert(WS_GA >= WS_APX * 16 && WS_GA % 256 == 64 && WS_GD % 256 == 64, "conflict-free group strides");

template <int COB>
struct WsCfg {
    static constexpr int DZ_PLANE = 4 * COB * WS_GD;
    static constexpr int A_PLANE = 4 * WS_GA;
    static constexpr int ACT_BASE = 2 * DZ_PLANE;
    static constexpr int LDS_BYTES = ACT_BASE + 2 * A_PLANE;
    static constexpr int DZ_UNITS = 2 * 4 * COB * WS_PX;          // 16-byte units
    static constexpr int A_UNITS = 2 * 4 * WS_APX;
    static constexpr int UNITS = DZ_UNITS + A_UNITS;
    static constexpr int ITERS = (UNITS + WS_THREADS - 1) / WS_THREADS;
};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ h16x8 tr_read8(const char* lds_addr) {
    // pixels (row block) +0..3 and +4..7 of this lane's channel: two 4x16 transposed blocks, 64 B apart
    auto p0 = (__attribute__((address_space(3))) s16x4w*)(lds_addr);
    auto p1 = (__attribute__((address_space(3))) s16x4w*)(lds_addr + 64);
    const s16x4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
    const s16x4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
    union { s16x4w s[2]; h16x8 h; } u;
    u.s[0] = a; u.s[1] = b;
    return u.h;
}
#endif

// grid = (nslab, Cin/32 blocks).  act: c8s [n][CGin][2][HW][8], dz: c8s [n][CGout][2][HW][8];
// slab layout: slabs[slab][tap][coP][ciP]  (coP = 32*COB, ciP = 32*gridDim.y)
template <int COB>
__global__ void __launch_bounds__(WS_THREADS)
conv3x3_wgrad_split_kernel(const char* __restrict__ act, const char* __restrict__ dz, float* __restrict__ slabs,
                           int n_img, int CGin, int CGout, int cg0, int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = WsCfg<COB>;
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, tap = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ky = tap / 3, kx = tap - 3 * ky;
    // (slab, input-channel block) of this workgroup.  The ncib blocks of one slab walk the same tiles at the same time and
    // read the same dZ; workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so when the slab count is a
    // multiple of 8 the linear id is remapped such that the blocks of a slab share an XCD and dZ is fetched from HBM once
    // per slab instead of once per block (measured before: 888 MB per launch against 402 MB algorithmic at 96 channels)
    int slab_id = blockIdx.x, cib = blockIdx.y;
    const int nslab = gridDim.x, ncib = gridDim.y;
    if (ncib > 1 && (nslab & 7) == 0) {
        const int L = blockIdx.x + nslab * blockIdx.y, j = L >> 3;
        slab_id = (L & 7) + 8 * (j / ncib);
        cib = j % ncib;
    }
    const size_t HW = (size_t)H * W;
    const int tiles_x = (W + WS_TW - 1) / WS_TW, tiles_y = (H + WS_TR - 1) / WS_TR;
    const int tiles = n_img * tiles_y * tiles_x;

    // ---- staging plan of this thread (constant over tiles): LDS byte offset, byte offset inside one image for tile
    // origin (0,0), and the (dy, dx) of the pixel for the bounds test; s_lds < 0 marks an idle / out-of-range slot
    int s_lds[Cfg::ITERS], s_off[Cfg::ITERS], s_dydx[Cfg::ITERS];
#pragma unroll
    for (int k = 0; k < Cfg::ITERS; ++k) {
        const int e = tid + k * WS_THREADS;
        s_lds[k] = -1; s_off[k] = 0; s_dydx[k] = 0;
        if (e < Cfg::DZ_UNITS) {
            const int pl = e / (4 * COB * WS_PX), rem = e - pl * (4 * COB * WS_PX);
            const int cgl = rem / WS_PX, px = rem - cgl * WS_PX;
            const int dy = px / WS_TW, dx = px % WS_TW;
            if (cg0 + cgl < CGout) {
                s_lds[k] = pl * Cfg::DZ_PLANE + cgl * WS_GD + px * 16;
                s_off[k] = (int)((((size_t)(cg0 + cgl) * 2 + pl) * HW + (size_t)dy * W + dx) * 16);
                s_dydx[k] = (dy << 16) | (dx & 0xffff);
            }
        } else if (e < Cfg::UNITS) {
            const int e2 = e - Cfg::DZ_UNITS;
            const int pl = e2 / (4 * WS_APX), rem = e2 - pl * (4 * WS_APX);
            const int cgl = rem / WS_APX, pos = rem - cgl * WS_APX;
            const int dy = pos / WS_AW - 1, dx = pos % WS_AW - 1;
            s_lds[k] = Cfg::ACT_BASE + pl * Cfg::A_PLANE + cgl * WS_GA + pos * 16;
            if (cib * 4 + cgl < CGin) {
                s_off[k] = (int)(((long long)((size_t)(cib * 4 + cgl) * 2 + pl) * (long long)HW + (long long)dy * W + dx) * 16);
                s_dydx[k] = (dy << 16) | (dx & 0xffff);
            } else {
                s_dydx[k] = (int)0x80008000;                    // far out of range: the slot stages zeros
            }
        }
    }
    constexpr int DZ_SLOTS_FULL = Cfg::DZ_UNITS / WS_THREADS;          // slots k < this are dZ for every thread
    uint4 stage[Cfg::ITERS];
    auto fetch = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int x0 = tx * WS_TW, y0 = ty * WS_TR;
        const size_t dz_img = (size_t)n * CGout * 2 * HW * 16, act_img = (size_t)n * CGin * 2 * HW * 16;
        const long long org = ((long long)y0 * W + x0) * 16;
#pragma unroll
        for (int k = 0; k < Cfg::ITERS; ++k) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int gy = y0 + (s_dydx[k] >> 16), gx = x0 + (int)(short)(s_dydx[k] & 0xffff);
            const bool isdz = (k < DZ_SLOTS_FULL) || (tid + k * WS_THREADS < Cfg::DZ_UNITS);
            if (s_lds[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const char* src = (isdz ? dz + dz_img : act + act_img) + (s_off[k] + org);
                v = *(const uint4*)src;
            }
            stage[k] = v;
        }
    };

    f32x16w acc_hh[COB], acc_x[COB];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_hh[cb][r] = 0.f; acc_x[cb][r] = 0.f; }

    // transposed-read lane addressing: 16-lane group G covers channels 16*(G&1).., K half lh = G>>1;
    // lane 4q+p of the group supplies pixel row q, channels 4p..4p+3 (channel group (p>>1), byte 8*(p&1))
    const int G = lane >> 4, l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    const int grp_sel = 2 * (G & 1) + (p >> 1);
    const int a_lane = grp_sel * WS_GD + (8 * lh + q) * 16 + 8 * (p & 1);
    const int b_lane = Cfg::ACT_BASE + grp_sel * WS_GA + ((ky * WS_AW + kx) + 8 * lh + q) * 16 + 8 * (p & 1);

    int tile = slab_id;
    if (tile < tiles) fetch(tile);
    for (; tile < tiles; tile += nslab) {
        __syncthreads();                                    // previous tile's reads are done
#pragma unroll
        for (int k = 0; k < Cfg::ITERS; ++k)
            if (s_lds[k] >= 0) *(uint4*)(smem_w + s_lds[k]) = stage[k];
        __syncthreads();
        if (tile + nslab < tiles) fetch(tile + nslab);      // in flight while this tile is consumed
#pragma unroll
        for (int ks = 0; ks < WS_PX / 16; ++ks) {
            const int r = ks >> 1, c0 = 16 * (ks & 1);
            const char* bp = smem_w + b_lane + (r * WS_AW + c0) * 16;
            const h16x8 b_hi = tr_read8(bp), b_lo = tr_read8(bp + Cfg::A_PLANE);
#pragma unroll
            for (int cb = 0; cb < COB; ++cb) {
                const char* ap = smem_w + a_lane + cb * 4 * WS_GD + ks * 256;
                const h16x8 a_hi = tr_read8(ap), a_lo = tr_read8(ap + Cfg::DZ_PLANE);
                acc_hh[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc_hh[cb], 0, 0, 0);
                acc_x[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc_x[cb], 0, 0, 0);
                acc_x[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc_x[cb], 0, 0, 0);
            }
        }
    }
    // C[row = co_local][col = ci_local]: row = (r&3) + 8*(r>>2) + 4*lh, col = li
    const int coP = 32 * COB, ciP = 32 * ncib;
    float* slab = slabs + ((size_t)slab_id * 9 + tap) * coP * ciP;
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            slab[(size_t)co * ciP + cib * 32 + li] = acc_hh[cb][r] + acc_x[cb][r] * (1.0f / 2048.0f);
        }
#endif
}

// dW[co][ci][tap] = scale * sum over slabs in fixed order (scale = exact power of two un-doing the dZ pre-scale)
__global__ void __launch_bounds__(256)
wgrad_reduce_scaled_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dW, int Cin_real, int co0,
                           int co_count, int coP, int ciP, float scale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = co_count * Cin_real * 9;
    if (idx >= total) return;
    // ci fastest: the lanes of a wave read consecutive floats of one slab row (the slab layout's fastest index); the
    // 36-byte-strided write of dW happens once, the slabs are read nslab times
    const int ci = idx % Cin_real, co = (idx / Cin_real) % co_count, tap = idx / (Cin_real * co_count);
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
    dW[((size_t)(co0 + co) * Cin_real + ci) * 9 + tap] = s * scale;
}

// db[co] = scale * sum dz over c8s (hi + lo'/2048)
__global__ void __launch_bounds__(256)
bgrad_partial_split_kernel(const char* __restrict__ dz, float* __restrict__ part, int n_img, int CG, size_t HW, int nchunk) {
    __shared__ float red[4 * 8];
    const int cg = blockIdx.y, chunk = blockIdx.x;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t total = (size_t)n_img * HW;
    for (size_t i = (size_t)chunk * blockDim.x + threadIdx.x; i < total; i += (size_t)nchunk * blockDim.x) {
        const size_t n = i / HW, px = i - n * HW;
        const char* g = dz + (n * CG + cg) * (2 * HW * 16);
        const h16x8 h = *(const h16x8*)(g + px * 16), l = *(const h16x8*)(g + HW * 16 + px * 16);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += (float)h[c] + (float)l[c] * (1.0f / 2048.0f);
    }
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

__global__ void bgrad_reduce_scaled_kernel(const float* __restrict__ part, float* __restrict__ db, int Cout_real,
                                           int nchunk, float scale) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= Cout_real) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[((size_t)(co >> 3) * nchunk + k) * 8 + (co & 7)];
    db[co] = s * scale;
}

static inline int ru32(int v) { return (v + 31) / 32 * 32; }

template <int COB>
static int launch_wgrad_split(const void* act, const void* dz, float* ws, int nslab, int ciP, int n, int Cin, int Cout,
                              int co0, int h, int w, hipStream_t st) {
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_wgrad_split_kernel<COB>, WsCfg<COB>::LDS_BYTES, "wgrad_split")) return rc;
    hipLaunchKernelGGL((conv3x3_wgrad_split_kernel<COB>), dim3(nslab, ciP / 32), dim3(WS_THREADS), WsCfg<COB>::LDS_BYTES, st,
                       (const char*)act, (const char*)dz, ws, n, Cin / 8, Cout / 8, co0 / 8, h, w);
    return launch_status("conv3x3_wgrad_split_kernel");
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_conv3x3_wgrad_split(const void* act_c8s, const void* dz_c8s, float* dW, float* workspace, int nslab, int n,
                               int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, float scale,
                               scipnp_stream_t s) {
    SCIPNP_REQUIRE(act_c8s && dz_c8s && dW && workspace, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin_real <= Cin && Cout_real <= Cout &&
                   nslab > 0 && nslab <= 65535, "bad shape");
    SCIPNP_REQUIRE((long long)h * w * 16 < (1ll << 31), "image too large");
    SCIPNP_ALIGNED(act_c8s); SCIPNP_ALIGNED(dz_c8s);
    const int ciP = ru32(Cin);
    hipStream_t st = (hipStream_t)s;
    for (int co0 = 0; co0 < Cout_real; co0 += 96) {
        const int left = ru32(Cout) - co0;
        const int coP = left < 96 ? left : 96;
        const int COB = coP / 32;
        int rc;
        if (COB == 1) rc = launch_wgrad_split<1>(act_c8s, dz_c8s, workspace, nslab, ciP, n, Cin, Cout, co0, h, w, st);
        else if (COB == 2) rc = launch_wgrad_split<2>(act_c8s, dz_c8s, workspace, nslab, ciP, n, Cin, Cout, co0, h, w, st);
        else rc = launch_wgrad_split<3>(act_c8s, dz_c8s, workspace, nslab, ciP, n, Cin, Cout, co0, h, w, st);
        if (rc) return rc;
        const int co_count = (Cout_real - co0) < coP ? (Cout_real - co0) : coP;
        const int total = co_count * Cin_real * 9;
        hipLaunchKernelGGL(wgrad_reduce_scaled_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, nslab, dW,
                           Cin_real, co0, co_count, coP, ciP, scale);
        rc = launch_status("wgrad_reduce_scaled_kernel");
        if (rc) return rc;
    }
    return SCIPNP_OK;
}

int scipnp_conv_bias_grad_split(const void* dz_c8s, float* db, float* workspace, int n, int Cout_real, int Cout, int h,
                                int w, float scale, scipnp_stream_t s) {
    SCIPNP_REQUIRE(dz_c8s && db && workspace && n > 0 && Cout % 8 == 0 && Cout_real <= Cout, "bad arguments");
    SCIPNP_ALIGNED(dz_c8s);
    const int nchunk = 64;   // workspace: (Cout/8) * 64 * 8 floats
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(bgrad_partial_split_kernel, dim3(nchunk, Cout / 8), dim3(256), 0, st, (const char*)dz_c8s, workspace, n,
                       Cout / 8, (size_t)h * w, nchunk);
    hipLaunchKernelGGL(bgrad_reduce_scaled_kernel, dim3((Cout_real + 63) / 64), dim3(64), 0, st, workspace, db, Cout_real,
                       nchunk, scale);
    return launch_status("bgrad split kernels");
}

}  // extern "C"
