// fp32 weight gradient of a 3x3 / pad 1 / stride 1 convolution in the Winograd F(2x2,3x3) domain
// (reference packages/ffdnet/test_ffdnet_ipol.py:296, packages/fastdvdnet/test_fastdvdnet.py:449 `loss.backward()`):
//   forward   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile
//   gradient  dg = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
// 16 exact fp32 products per tile and channel pair instead of the 36 of the direct form (csrc/finetune.hip), accumulated in
// fp32 on v_mfma_f32_32x32x2_f32 with the TILES as the K dimension:  dU_p[co][ci] = sum_t dM_p[co][t] V_p[ci][t].
//
// Workgroup = 16 waves, wave p <-> Winograd position p = 4 xi + nu, 32*COB output channels x 32 input channels; persistent
// over chunks of 8 tiles (16 x 2 output pixels).  One thread per (tile, channel) of a chunk loads the raw 2x2 output-gradient
// tile or the raw 4x4 input patch, transforms it in registers and writes its 16 transformed values to LDS
// [position][tile][channel] -- the operand a lane of the MFMA needs is then ONE ds_read_b32.  The raw loads of chunk k+1 are
// in flight under the MFMAs of chunk k, its transform + LDS writes follow them, one barrier per chunk, two LDS buffers.
// v_mfma_f32_32x32x2_f32 shares the fp32 vector lanes with the transform adds (tools/probes/mfma_valu_coissue.py), so the
// ~50 adds per thread and chunk are matrix time: (768 matrix cycles + ~250) per chunk and wave against 1728 in direct form.
// fp32 slabs + fixed-order reduction (deterministic, no atomics); the reduction applies G^T . G in double.
#include "common.hpp"
#include <type_traits>

namespace scipnp {

typedef float ww_f32x16 __attribute__((ext_vector_type(16)));
typedef float ww_f32x4 __attribute__((ext_vector_type(4)));

constexpr int WW_T = 8;                         // tiles per chunk, along x
constexpr int WW_MFMA_WAVES = 8;                // two Winograd positions per wave

template <int COB>
struct WwCfg {
    static constexpr int COP = 32 * COB;                           // output channels of the workgroup
    static constexpr int DZ_WAVES = 2 * COB;                       // producers of dM: one lane per (tile, channel pair)
    static constexpr int THREADS = 64 * (WW_MFMA_WAVES + DZ_WAVES + 2); // ... + 2 waves producing V
    static constexpr int M_FLOATS = 16 * WW_T * COP;               // dM [p][tile][co]
    static constexpr int V_FLOATS = 16 * WW_T * 32;                // V  [p][tile][ci]
    static constexpr int BUF_FLOATS = M_FLOATS + V_FLOATS;
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_FLOATS * sizeof(float);
};

// grid = (nslab, Cin/32 blocks).  act: [n][CGin][h][w][8], dz: [n][CGout][h][w][8];
// slab layout: slabs[slab][p 16][coP][ciP]  (coP = 32*COB, ciP = 32*gridDim.y)
//
// Waves are specialised: waves 0..7 only multiply (wave w <-> positions 2w, 2w+1, 96 x 32 accumulators each), waves 8.. only
// load and transform -- one lane per (tile, channel pair): 8-byte loads through a buffer descriptor whose per-lane offsets
// are constants of the thread (the chunk origin is a scalar offset: no vector arithmetic per chunk, which would be matrix
// time on this MFMA), the transform on both channels at once, sixteen 8-byte LDS writes [position][tile][channel].  The
// producers work one chunk ahead of the consumers (two LDS buffers, one barrier per chunk), their loads two chunks ahead.
// First version (every wave loading, transforming and multiplying in turn, one channel per lane): 943 us at 96 -> 96;
// roles per wave with a chunk loop per role, so that nothing is merged behind a load: 483 us; this form: 440 us.
template <int COB>
__global__ void __launch_bounds__(WwCfg<COB>::THREADS)
conv3x3_wgrad_wino_kernel(const float* __restrict__ act, const float* __restrict__ dz, float* __restrict__ slabs, int n_img,
                          int CGin, int CGout, int cg0 /* first output channel group of this launch */, int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = WwCfg<COB>;
    extern __shared__ __attribute__((aligned(16))) float smem_ww[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: the role branches below are uniform
    const int li = lane & 31, lh = lane >> 5;
    const int cib = blockIdx.y;
    const size_t HW = (size_t)H * W;
    const int tiles_x = (W + 1) / 2, tiles_y = (H + 1) / 2;
    const int chunks_x = (tiles_x + WW_T - 1) / WW_T;
    const int chunks = n_img * tiles_y * chunks_x;
    const unsigned dz_bytes = (unsigned)((size_t)n_img * CGout * HW * 32), act_bytes = (unsigned)((size_t)n_img * CGin * HW * 32);
    const int first = blockIdx.x, step = gridDim.x;
    const int coP = Cfg::COP, ciP = 32 * gridDim.y;

    if (wave < WW_MFMA_WAVES) {
        // ---------------------------------------------------------------- consumers
        ww_f32x16 acc[2][COB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.f;
        // MFMA operands: A[row = co][k = tile], B[k = tile][col = ci]; lane (li, lh) supplies row / column li of K index lh
        const int p0 = 2 * wave;
        const int a_lane = p0 * WW_T * Cfg::COP + lh * Cfg::COP + li;            // + i*WW_T*COP + (2 ks)*COP + cb*32
        const int b_lane = Cfg::M_FLOATS + p0 * WW_T * 32 + lh * 32 + li;        // + i*WW_T*32 + (2 ks)*32
        if (first < chunks) __syncthreads();                                     // chunk `first` is in buffer 0
        int cur = 0;
        for (int chunk = first; chunk < chunks; chunk += step) {
            const float* buf = smem_ww + cur * Cfg::BUF_FLOATS;
#pragma unroll
            for (int ks = 0; ks < WW_T / 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float bv = buf[b_lane + i * WW_T * 32 + 2 * ks * 32];
#pragma unroll
                    for (int cb = 0; cb < COB; ++cb) {
                        const float av = buf[a_lane + i * WW_T * Cfg::COP + 2 * ks * Cfg::COP + cb * 32];
                        acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][cb], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                    // the producers have filled the other buffer meanwhile
            cur ^= 1;
        }
        // C[row = co_local][col = ci_local]: row = 32 cb + (r&3) + 8*(r>>2) + 4*lh, col = li.  A running pointer (rows +1 +1 +1 +5,
        // also across cb): the row offsets held at once would spill
        {
            float* out = slabs + (((size_t)blockIdx.x * 16 + p0) * coP + 4 * lh) * ciP + cib * 32 + li;
            const size_t pos_stride = (size_t)coP * ciP - (size_t)(32 * COB) * ciP;      // the +5 behind a position's last row overshoots to row 32 COB
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int cb = 0; cb < COB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        *out = acc[i][cb][r];
                        out += ((r & 3) == 3 ? (size_t)5 * ciP : (size_t)ciP);
                    }
                out += pos_stride;
            }
        }
        return;
    }

    // -------------------------------------------------------------------- producers
    // lane <-> (tile tl of the chunk, four channels 4 q4 .. 4 q4 + 3) of the output gradient (waves 8 .. 8+COB-1) or of the
    // input (last wave)
    auto produce = [&](auto role) {
        constexpr bool DZ = decltype(role)::value;
        constexpr int NLD = DZ ? 4 : 16;                           // loads per lane: 2x2 tile or 4x4 patch
        constexpr int VW = 2;                                      // channels per lane: few vector instructions per producer wave
                                                                   // (each of them waits for a gap between the consumers' MFMAs),
                                                                   // many producer waves
        typedef float vec_t __attribute__((ext_vector_type(VW)));
        constexpr int QPT = (DZ ? 32 * COB : 32) / VW;             // lanes per tile
        const int t = tid - 64 * (WW_MFMA_WAVES + (DZ ? 0 : Cfg::DZ_WAVES));
        const int tl = t / QPT, q = t - tl * QPT;
        const int cgl = (q * VW) >> 3;
        const bool ch_on = DZ ? (cg0 + cgl < CGout) : (cib * 4 + cgl < CGin);
        const int w_off = DZ ? (tl * Cfg::COP + VW * q) : (Cfg::M_FLOATS + tl * 32 + VW * q);
        constexpr int w_step = DZ ? WW_T * Cfg::COP : WW_T * 32;
        // a load = (lane offset: channel pair + tile column) + (immediate: dx) + (scalar: chunk origin + row dy); the input's
        // origin is shifted by (-1, -1) through the descriptor's base.  Loads that fall off the image (first / last tile row,
        // first / last chunk column) get an out-of-range lane offset and return 0.
        constexpr unsigned OOB = 0x80000000u;                      // >= num_records, and stays so with an immediate added
        constexpr int sh = DZ ? 0 : -1;
        constexpr int ROWS = DZ ? 2 : 4;                           // 2x2 tile / 4x4 patch
        const unsigned lane_base = ch_on ? (unsigned)((size_t)(DZ ? (cg0 + cgl) : (cib * 4 + cgl)) * HW * 32 + 4 * ((q * VW) & 7) +
                                                      2 * tl * 32) : OOB;
        unsigned colL = 0, colR = 0, rowB = 0;                     // bit dx / dy: off the image in the first / last chunk column / row
        constexpr unsigned rowT = DZ ? 0u : 1u;
#pragma unroll
        for (int d = 0; d < ROWS; ++d) {
            if (2 * tl + d + sh < 0) colL |= 1u << d;
            if (2 * WW_T * (chunks_x - 1) + 2 * tl + d + sh >= W) colR |= 1u << d;
            if (2 * (tiles_y - 1) + d + sh >= H) rowB |= 1u << d;
        }
        // the bytes in front of the input tensor are only ever addressed by loads of the first tile row / chunk column of
        // image 0, which are masked (offset out of range)
        const auto rs = DZ ? __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, dz_bytes, 0x00020000)
                           : __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)act - ((size_t)W + 1) * 32), 0,
                                                               act_bytes + (unsigned)((W + 1) * 32), 0x00020000);
        vec_t raw[NLD];                                            // (a second set, requested two chunks ahead, changes nothing:
                                                                   //  446 against 441 us -- the loads' latency is not what the kernel waits for)
        auto load1 = [&](unsigned vo, unsigned so) {
            return __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0));
        };
        // chunk -> (image n, tile row ty, chunk column cx) without a division per chunk: integer division runs on the vector
        // ALU, where every instruction of a producer waits for a gap between the consumers' MFMAs.  The walker advances by
        // `step` chunks per call with carries; the decomposition of `step` is computed once.
        const int per_img = tiles_y * chunks_x;
        const int d_n = step / per_img, d_ty = (step - d_n * per_img) / chunks_x, d_cx = step - d_n * per_img - d_ty * chunks_x;
        int w_n = first / per_img, w_ty = (first - w_n * per_img) / chunks_x, w_cx = first - w_n * per_img - w_ty * chunks_x;
        int w_chunk = first;
        auto advance = [&]() {
            w_chunk += step;
            w_cx += d_cx;
            if (w_cx >= chunks_x) { w_cx -= chunks_x; ++w_ty; }
            w_ty += d_ty;
            if (w_ty >= tiles_y) { w_ty -= tiles_y; ++w_n; }
            w_n += d_n;
        };
        auto fetch = [&](vec_t* raw) {                             // the walker's chunk (clamped: past the end, the last one again)
            const int cx = w_cx, ty = w_ty, n = w_n;
            const bool edge = ty == 0 || ty == tiles_y - 1 || cx == 0 || cx == chunks_x - 1;        // wave-uniform
            const unsigned so = (unsigned)((long long)n * (DZ ? CGout : CGin) * (long long)HW * 32 +
                                           ((long long)2 * ty * W + (long long)2 * WW_T * cx) * 32);
            if (edge) {
                const unsigned rows = (ty == 0 ? rowT : 0u) | (ty == tiles_y - 1 ? rowB : 0u);          // scalar
                const unsigned cols = (cx == 0 ? colL : 0u) | (cx == chunks_x - 1 ? colR : 0u);         // per lane
#pragma unroll
                for (int dy = 0; dy < ROWS; ++dy) {
                    const unsigned row_base = ((rows >> dy) & 1u) ? OOB : lane_base;
#pragma unroll
                    for (int dx = 0; dx < ROWS; ++dx)
                        raw[dy * ROWS + dx] = load1((((cols >> dx) & 1u) ? OOB : row_base) + dx * 32, so + (unsigned)(dy * W * 32));
                }
            } else {
#pragma unroll
                for (int dy = 0; dy < ROWS; ++dy)
#pragma unroll
                    for (int dx = 0; dx < ROWS; ++dx)
                        raw[dy * ROWS + dx] = load1(lane_base + dx * 32, so + (unsigned)(dy * W * 32));
            }
        };
        auto transform_store = [&](float* buf, const vec_t* raw) {
            float* dst = buf + w_off;
            if constexpr (DZ) {     // A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
                vec_t rr[4][2];
                rr[0][0] = raw[0];          rr[0][1] = raw[1];
                rr[1][0] = raw[0] + raw[2]; rr[1][1] = raw[1] + raw[3];
                rr[2][0] = raw[0] - raw[2]; rr[2][1] = raw[1] - raw[3];
                rr[3][0] = -raw[2];         rr[3][1] = -raw[3];
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    *(vec_t*)(dst + (xi * 4 + 0) * w_step) = rr[xi][0];
                    *(vec_t*)(dst + (xi * 4 + 1) * w_step) = rr[xi][0] + rr[xi][1];
                    *(vec_t*)(dst + (xi * 4 + 2) * w_step) = rr[xi][0] - rr[xi][1];
                    *(vec_t*)(dst + (xi * 4 + 3) * w_step) = -rr[xi][1];
                }
            } else {                // B^T d B, as conv_wino.hip
                vec_t tt[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    tt[0][c] = raw[0 * 4 + c] - raw[2 * 4 + c];
                    tt[1][c] = raw[1 * 4 + c] + raw[2 * 4 + c];
                    tt[2][c] = raw[2 * 4 + c] - raw[1 * 4 + c];
                    tt[3][c] = raw[1 * 4 + c] - raw[3 * 4 + c];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    *(vec_t*)(dst + (r * 4 + 0) * w_step) = tt[r][0] - tt[r][2];
                    *(vec_t*)(dst + (r * 4 + 1) * w_step) = tt[r][1] + tt[r][2];
                    *(vec_t*)(dst + (r * 4 + 2) * w_step) = tt[r][2] - tt[r][1];
                    *(vec_t*)(dst + (r * 4 + 3) * w_step) = tt[r][1] - tt[r][3];
                }
            }
        };
        if (first >= chunks) return;
        fetch(raw);                                                // chunk `first`
        transform_store(smem_ww, raw);
        if (w_chunk + step < chunks) advance();                    // (past the end the last chunk is fetched again: no branch
        fetch(raw);                                                //  around loads, nothing selected behind them)
        __syncthreads();
        int cur = 0;
        for (int chunk = first; chunk < chunks; chunk += step) {
            // chunk + step goes to the other buffer (last read by the consumers before the previous barrier) while they
            // multiply `chunk`; its raw values were requested a whole chunk ago
            transform_store(smem_ww + (cur ^ 1) * Cfg::BUF_FLOATS, raw);
            if (w_chunk + step < chunks) advance();
            fetch(raw);
            __syncthreads();
            cur ^= 1;
        }
    };
    if (wave < WW_MFMA_WAVES + Cfg::DZ_WAVES) produce(std::true_type{}); else produce(std::false_type{});
#endif
}

// sum over slabs in fixed order, one thread per (position, co, ci): the sums replace slab 0 in place (every thread reads
// and writes only its own element)
__global__ void __launch_bounds__(256)
wgrad_wino_sum_kernel(float* __restrict__ slabs, int nslab, int coP, int ciP) {
    const size_t stride = (size_t)16 * coP * ciP;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= stride) return;
    float* pp = slabs + idx;
    // four interleaved partial sums (slab k -> accumulator k % 4), then ((s0+s1)+s2)+s3, as wgrad_reduce_kernel
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nslab; k += 4) {
        s0 += pp[(size_t)k * stride];
        s1 += pp[(size_t)(k + 1) * stride];
        s2 += pp[(size_t)(k + 2) * stride];
        s3 += pp[(size_t)(k + 3) * stride];
    }
    for (; k < nslab; ++k) s0 += pp[(size_t)k * stride];
    pp[0] = ((s0 + s1) + s2) + s3;
}

// dW[co][ci][ky][kx] (OIHW, real channel counts) = G^T S G from the summed slab, in double
__global__ void __launch_bounds__(256)
wgrad_wino_finish_kernel(const float* __restrict__ sums, float* __restrict__ dW, int Cin_real, int co0,
                         int co_count /* real output channels of this chunk */, int coP, int ciP) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = co_count * Cin_real;
    if (idx >= total) return;
    const int ci = idx % Cin_real, co = idx / Cin_real;                   // ci fastest: coalesced reads
    const size_t pstride = (size_t)coP * ciP;
    const float* base = sums + (size_t)co * ciP + ci;
    double u[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) u[q] = (double)base[q * pstride];
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    float* out = dW + ((size_t)(co0 + co) * Cin_real + ci) * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            double s = 0.0;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) s += G[xi][ky] * G[nu][kx] * u[xi * 4 + nu];
            out[ky * 3 + kx] = (float)s;
        }
}

static inline int ww_round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace scipnp

using namespace scipnp;

extern "C" {

size_t scipnp_conv3x3_wgrad_wino_workspace_floats(int Cin, int Cout, int nslab) {
    if (Cin <= 0 || Cout <= 0 || nslab <= 0) return 0;
    const int chunk = ww_round_up(Cout, 32) < 96 ? ww_round_up(Cout, 32) : 96;
    return (size_t)nslab * 16 * chunk * ww_round_up(Cin, 32);
}

int scipnp_conv3x3_wgrad_wino(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                              int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(act_c8 && dz_c8 && dW && workspace, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin_real <= Cin && Cout_real <= Cout &&
                   nslab > 0 && nslab <= 65535, "bad shape");
    SCIPNP_REQUIRE((long long)n * Cin * h * w * 4 < (1ll << 31) && (long long)n * Cout * h * w * 4 < (1ll << 31),
                   "tensors of 2 GiB or more: use scipnp_conv3x3_wgrad");
    SCIPNP_ALIGNED(act_c8); SCIPNP_ALIGNED(dz_c8);
    const int ciP = ww_round_up(Cin, 32);
    hipStream_t st = (hipStream_t)s;
    // output channels in chunks of <= 96 (3 MFMA row blocks per wave); one slab set + reduction per chunk
    for (int co0 = 0; co0 < Cout_real; co0 += 96) {
        const int left = ww_round_up(Cout, 32) - co0;
        const int coP = left < 96 ? left : 96;
        const int COB = coP / 32;
        const dim3 grid(nslab, ciP / 32);
#define SCIPNP_WW(C)                                                                                                  \
    do {                                                                                                              \
        static LdsAttrOnce attr;                                                                                      \
        if (int rc_ = attr.ensure((const void*)conv3x3_wgrad_wino_kernel<C>, WwCfg<C>::LDS_BYTES, "wgrad_wino")) return rc_; \
        hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<C>), grid, dim3(WwCfg<C>::THREADS), WwCfg<C>::LDS_BYTES, st, act_c8, \
                           dz_c8, workspace, n, Cin / 8, Cout / 8, co0 / 8, h, w);                                    \
    } while (0)
        if (COB == 1) SCIPNP_WW(1); else if (COB == 2) SCIPNP_WW(2); else SCIPNP_WW(3);
#undef SCIPNP_WW
        int rc = launch_status("conv3x3_wgrad_wino_kernel");
        if (rc) return rc;
        const int co_count = (Cout_real - co0) < coP ? (Cout_real - co0) : coP;
        const size_t per_slab = (size_t)16 * coP * ciP;
        hipLaunchKernelGGL(wgrad_wino_sum_kernel, dim3((unsigned)((per_slab + 255) / 256)), dim3(256), 0, st, workspace, nslab,
                           coP, ciP);
        const int total = co_count * Cin_real;
        hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, dW, Cin_real, co0,
                           co_count, coP, ciP);
        rc = launch_status("wgrad_wino_sum / finish kernels");
        if (rc) return rc;
    }
    return SCIPNP_OK;
}

}  // extern "C"
