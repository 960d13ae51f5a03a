// 3x3 convolution with ERROR-COMPENSATED SPLIT-FP16 operands on the CDNA4 matrix cores.
//
// Every fp32 operand v is carried as two fp16 numbers  v = hi + lo' * 2^-11,  hi = fp16(v),
// lo' = fp16((v - hi) * 2^11)  (22 significant bits; the 2^11 keeps lo' in fp16's NORMAL range so nothing
// depends on subnormal handling).  A product a*b is evaluated as
//     a_hi*b_hi  +  2^-11 * (a_lo'*b_hi + a_hi*b_lo')          (the a_lo*b_lo term, 2^-22 relative, is dropped)
// with v_mfma_f32_32x32x16_f16: fp16 x fp16 products are exact in fp32 and the accumulation is fp32, so the
// only error beyond an fp32 convolution is the 2^-22 operand representation -- measured in the PnP loop:
// <= 3.4e-6 relative L2 per iterate over the reference's 25-iteration FFDNet schedule (bar: 1e-5), where plain
// fp16/bf16 operands give 3e-3.  Cost: 14 MFMAs of 32 cycles per (8 input channels x 9 taps x 32x32 block)
// against 36 fp32 MFMAs of 64 cycles: 5.1x fewer matrix-pipe cycles.
//
//   acc += [2^11 w_hi(tap a) | 2^11 w_hi(tap b)] x [x_hi(tap a) | x_hi(tap b)]   taps paired along K = 16 (5 MFMAs)
//   acc += [w_lo'(tap)       | w_hi(tap)       ] x [x_hi(tap)   | x_lo'(tap)  ]   both cross terms in one K (9 MFMAs)
//   out  = 2^-11 * acc + bias
// (one accumulator: the hi*hi term is scaled UP by the exact factor 2^11 -- applied to the w_hi fragment with a
//  packed fp16 multiply, |w| < 31.9 is checked when packing -- instead of scaling the cross terms down.)
//
// Layout "c8s": activations [n][C/8][2][h][w][8] fp16 -- per 8-channel group a hi plane and a lo' plane, 16 bytes
// per pixel and plane: the same footprint as fp32 c8; lanes of a wave read/write 16 B at 16-B stride in both
// global memory and LDS (bank-conflict free).
//
// Structure (as conv.hip): workgroup = 4 waves = 8x32 output pixels x (32*COB) channels, wave = 2 rows; K loop over
// the input channel groups; per group the input halo tile (both planes) and the weight slab are staged into LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write; lanes outside the image fetch zeros through
// the buffer descriptor's bounds check); fragments are read with ds_read_b128.  Dispatched form for stride 1 (WS = 1):
// the input tile is double-buffered and fetched one group ahead, the weight slab has ONE buffer that is refilled
// between two barriers after each group -- 168 VGPRs, 51 KiB LDS -> 3 workgroups (12 waves) per CU.  Stride-2 layers and
// flag 0x200 keep both regions double-buffered (one barrier per group, 77 KiB for COB = 3 -> 2 workgroups per CU).
// Per group and wave: 84 MFMAs, 60 ds_read_b128, 60 v_pk_mul_f16 (the 2^11 of the hi x hi weight fragments).
// Epilogues: split again to c8s (+ReLU), or fp32 c8 (network tails), or fp32 c8 with PixelShuffle(2) folded into the
// store; stride 1 or 2.
#include "common.hpp"
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstring>

namespace scipnp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int CS_TW = 32;
constexpr float CS_LO_SCALE = 2048.f, CS_LO_INV = 1.f / 2048.f;

// PB = output rows per wave, NW = waves per workgroup: the workgroup tile is (NW*PB) rows x 32 columns.  <2, 4> = 8 x 32
// pixels, two workgroups per CU, is what is dispatched.  Measured alternative <4, 8> (32 x 32 pixels, one 8-wave workgroup
// per CU, 192 accumulator registers per lane, twice the matrix work per LDS operand read and staged weight byte):
// 284 us against 231-242 us for the FFDNet body layer -- 256 VGPRs with spills, and a lone workgroup per CU leaves
// its barrier waits uncovered.
template <int COB, int STRIDE, int PB = 2, int NW = 4, int WS = 0>
struct SplitCfg {
    static constexpr int TH = NW * PB;
    static constexpr int THREADS = NW * 64;
    static constexpr int TWP = (CS_TW - 1) * STRIDE + 3;      // input tile columns (34 / 65)
    static constexpr int THP = (TH - 1) * STRIDE + 3;         // input tile rows    (10 / 17; 34 for the 32-row tile)
    static constexpr int IN_PLANE = THP * TWP * 16;           // bytes of one plane of the input tile
    static constexpr int IN_VEC = 2 * THP * TWP;              // 16-byte units (both planes)
    static constexpr int IN_ITERS = (IN_VEC + THREADS - 1) / THREADS;
    static constexpr int COUTP = 32 * COB;
    static constexpr int W_VEC = 9 * 2 * COUTP;
    static constexpr int W_ITERS = (W_VEC + THREADS - 1) / THREADS;
    // every lane of every 64-lane LDS-DMA instruction writes 16 bytes: both regions are padded to whole 256-unit
    // rounds (out-of-range lanes fetch zeros through the buffer bounds check)
    static constexpr int IN_PAD = IN_ITERS * THREADS * 16;
    static constexpr int W_PAD = W_ITERS * THREADS * 16;
    // WS = 0: [input | weights] x 2 buffers, both prefetched one group ahead.  WS = 1: the input tile is double-buffered,
    // the weight slab is not (it is re-fetched between two barriers after each group): 51 KiB instead of 77 KiB for
    // COB = 3, i.e. three workgroups (12 waves) per CU cover each other's waits instead of two.
    static constexpr int STAGE = WS ? IN_PAD : IN_PAD + W_PAD;
    static constexpr int W_AT = WS ? 2 * IN_PAD : IN_PAD;                 // weights: after both input buffers / inside the stage
    static constexpr size_t LDS_BYTES = WS ? 2 * (size_t)IN_PAD + W_PAD : 2 * (size_t)STAGE;
    static constexpr int WAVES_PER_SIMD = WS ? 3 : (NW == 8 ? 2 : ((LDS_BYTES * 2 <= 160 * 1024 && COB <= 3) ? 2 : 1));
};

struct SplitArgs {
    const char* in;      // c8s activations
    const char* wpk;     // packed split weights: [cig][tap][2][CoutP][8] fp16, then bias fp32 [CoutP]
    char* out;           // c8s (fp16 split) or fp32 c8
    const char* mask;    // c8s tensor of the output's shape: output zeroed where it is <= 0 (flag bit4; backward-data conv)
    const char* res;     // c8s tensor of the output's shape added before the mask (flag bit1; skip-connection gradient)
    int CGin, CGout, CoutP_total, nsplit;
    int H, W;            // input size
    int Ho, Wo;          // conv output size (before any pixel shuffle)
    int flags;           // bit0 ReLU, bit1 (2) residual, bit4 (16) ReLU-mask, bit5 (32) fp32 c8 output, bit3 (8) pixel-shuffle
                         // store (fp32 c8, or c8s with bit6 (64), where the residual has the shuffled shape)
    int* ovf;            // range-guard word of this launch (see below)
};

// Range guard: a launch raises ITS word when a value leaves fp16's finite range on its way into the c8s format (results
// are then invalid and the host must rerun with the fp32 kernels).  The word is a kernel argument: the device int the
// calling thread bound with scipnp_bind_overflow_word (one per solve / engine, owned and zeroed by the caller), or -- for
// callers that never bound one -- the process-wide word below.  Overlapping solves on different host threads therefore
// never see each other's report.
__device__ int g_split_overflow = 0;

__device__ __forceinline__ void split_store(float v0, float v1, float v2, float v3, char* hi_ptr, char* lo_ptr, int* ovf) {
    f16x4 h, l;
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hh = (_Float16)v[e];
        h[e] = hh;
        l[e] = (_Float16)((v[e] - (float)hh) * CS_LO_SCALE);
    }
    if (!(fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))) < 65000.f)) *ovf = 1;   // also NaN
    *(f16x4*)hi_ptr = h;
    *(f16x4*)lo_ptr = l;
}

// the same split, stored as ONE 16-byte vector per lane: lanes l and l+32 of the wave hold the two channel quads of one
// pixel's 8-channel group; after exchanging a half each, lane l < 32 stores [hi quad 0 | hi quad 1] to the hi plane and lane
// l+32 [lo' quad 0 | lo' quad 1] to the lo' plane (`dst` = that lane's plane + pixel offset).  Both lanes of a pair must be active.
__device__ __forceinline__ void split_store16(float v0, float v1, float v2, float v3, char* dst, int* ovf) {
#if defined(__HIP_DEVICE_COMPILE__)
    f16x4 h, l;
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hh = (_Float16)v[e];
        h[e] = hh;
        l[e] = (_Float16)((v[e] - (float)hh) * CS_LO_SCALE);
    }
    if (!(fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))) < 65000.f)) *ovf = 1;   // also NaN
    union { f16x4 f; unsigned u[2]; } hu, lu;
    hu.f = h; lu.f = l;
    // permlane32_swap(x, y): x of lanes 32..63 <-> y of lanes 0..31
    const auto r0 = __builtin_amdgcn_permlane32_swap(hu.u[0], lu.u[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hu.u[1], lu.u[1], false, false);
    // lanes < 32: (r[0], r[1]) = (own hi, partner's hi);  lanes >= 32: (partner's lo', own lo')
#ifdef CS_STORE_NT                                    /* variant build (make csvariant): the c8s stores with the nt hint */
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((u4v){r0[0], r1[0], r0[1], r1[1]}, (u4v*)dst);
#else
    *(uint4*)dst = make_uint4(r0[0], r1[0], r0[1], r1[1]);
#endif
#endif
}

// TAG only changes the symbol name (1 = network head layer) so profiler statistics of the body layers stay clean.
template <int COB, int TAG, int STRIDE, int SHUF, int PB = 2, int NW = 4, int WS = 0>
__global__ void __launch_bounds__((SplitCfg<COB, STRIDE, PB, NW, WS>::THREADS), (SplitCfg<COB, STRIDE, PB, NW, WS>::WAVES_PER_SIMD))
conv3x3_c8s_kernel(const SplitArgs a) {
    using Cfg = SplitCfg<COB, STRIDE, PB, NW, WS>;
    constexpr int CS_THREADS = Cfg::THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the linear id is
    // remapped such that one XCD works through a CONTIGUOUS run of tiles (for the FFDNet body: one frame per XCD) and
    // the halo rows shared by vertically adjacent tiles hit in that XCD's L2 instead of being fetched by another one
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
        if (!(a.flags & 0x800) && (total & 7) == 0) {
            unsigned w = bx + nx * (by + ny * bz);
            w = (w & 7) * (total >> 3) + (w >> 3);
            bx = w % nx; by = (w / nx) % ny; bz = w / (nx * ny);
        }
    }
    const int x0 = bx * CS_TW, y0 = by * Cfg::TH;
    const int n = bz / a.nsplit, split = bz % a.nsplit;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const size_t grp_bytes = 2 * HW * 16;                      // one input channel group (both planes)

    // staging plan: unit e = tid + 256k of a region is lane (tid & 63) of the wave instruction that fills the
    // 1 KiB at byte 16*(wave*64 + 256k).  Byte offsets inside one channel group / weight slab; OOB -> zeros.
    const unsigned OOB = 0x80000000u;
    unsigned in_off[Cfg::IN_ITERS];
#pragma unroll
    for (int k = 0; k < Cfg::IN_ITERS; ++k) {
        const int e = tid + k * CS_THREADS;
        in_off[k] = OOB;
        if (e < Cfg::IN_VEC) {
            const int plane = e / (Cfg::THP * Cfg::TWP), pix = e - plane * (Cfg::THP * Cfg::TWP);
            const int r = pix / Cfg::TWP, c = pix - r * Cfg::TWP;
            const int gy = y0 * STRIDE - 1 + r, gx = x0 * STRIDE - 1 + c;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) in_off[k] = (unsigned)((plane * HW + (size_t)gy * W + gx) * 16);
        }
    }
    unsigned w_off[Cfg::W_ITERS];
#pragma unroll
    for (int k = 0; k < Cfg::W_ITERS; ++k) {
        const int e = tid + k * CS_THREADS;
        w_off[k] = OOB;
        if (e < Cfg::W_VEC) {
            const int tp = e / Cfg::COUTP, co = e - tp * Cfg::COUTP;       // tp = tap*2 + plane
            w_off[k] = (unsigned)((tp * a.CoutP_total + split * Cfg::COUTP + co) * 16);
        }
    }
    const char* in_g = a.in + (size_t)n * a.CGin * grp_bytes;
    const char* w_g = a.wpk;
    const size_t w_step = (size_t)9 * 2 * a.CoutP_total * 16;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    (void)wvu; (void)in_g; (void)w_g;      // only read by the device-only staging code below
    auto dma_in = [&](char* buf) {
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins: keep them out of the host pass that only emits the launch stub
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, (int)grp_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < Cfg::IN_ITERS; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_in, (__attribute__((address_space(3))) void*)(buf + 16 * (wvu * 64 + k * CS_THREADS)), 16, in_off[k], 0, 0, 0);
#endif
        in_g += grp_bytes;
    };
    auto dma_w = [&](char* wbuf) {
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, (int)w_step, 0x00020000);
#pragma unroll
        for (int k = 0; k < Cfg::W_ITERS; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_w, (__attribute__((address_space(3))) void*)(wbuf + 16 * (wvu * 64 + k * CS_THREADS)), 16, w_off[k], 0, 0, 0);
#endif
        w_g += w_step;
    };
    auto dma_stage = [&](char* buf) {
        dma_in(buf);
        dma_w(WS ? smem_s + Cfg::W_AT : buf + Cfg::W_AT);
    };

    f32x16 acc[PB][COB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int cb = 0; cb < COB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][cb][r] = 0.f;

    dma_stage(smem_s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // LDS byte offsets.  input: plane p at p*IN_PLANE, pixel (r,c) at (r*TWP + c)*16
    //                    weights: IN_PAD + ((tap*2 + plane)*COUTP + co)*16
    const int px_base = ((PB * wv * STRIDE) * Cfg::TWP + li * STRIDE) * 16;   // + ((pb*S+ky)*TWP + kx)*16
    // weight fragment offsets are relative to the weight region of the current buffer (WS: the one shared region)
    const int co_base = li * 16;                                             // + ((tap*2+plane)*COUTP + cb*32)*16
    // hi x hi tap pairs: lane half h handles tap 2p+h; per-lane offsets and scale (0 for the missing tap 9) hoisted
    int pair_px[5], pair_co[5];
    _Float16 pair_scale[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int tap_h = 2 * p + lh;
        const bool live = tap_h < 9;
        const int tp = live ? tap_h : 8;
        const int ky = tp / 3, kx = tp - 3 * ky;
        pair_px[p] = px_base + (ky * Cfg::TWP + kx) * 16;
        pair_co[p] = co_base + (tp * 2 * Cfg::COUTP) * 16;
        pair_scale[p] = (_Float16)(live ? CS_LO_SCALE : 0.f);
    }

    for (int cig = 0; cig < a.CGin; ++cig) {
        const char* buf = smem_s + (cig & 1) * Cfg::STAGE;
        const char* wb = WS ? smem_s + Cfg::W_AT : buf + Cfg::W_AT;
        if (cig + 1 < a.CGin) {
            if (WS) dma_in(smem_s + ((cig + 1) & 1) * Cfg::STAGE);
            else dma_stage(smem_s + ((cig + 1) & 1) * Cfg::STAGE);
        }
        // 14 MFMA steps per group: 5 hi x hi tap pairs (lane half h handles tap 2p+h, tap 9 -> zero weights)
        // then 9 cross-term taps (k 0..7 = w_lo' x_hi from lane half 0, k 8..15 = w_hi x_lo' from lane half 1).
        f16x8 bfA[PB], afA[COB], bfB[PB], afB[COB];
        auto load_step = [&](int s, f16x8 (&bf)[PB], f16x8 (&af)[COB]) {
            if (s < 5) {
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) bf[pb] = *(const f16x8*)(buf + pair_px[s] + pb * STRIDE * Cfg::TWP * 16);
#pragma unroll
                for (int cb = 0; cb < COB; ++cb) af[cb] = *(const f16x8*)(wb + pair_co[s] + cb * 32 * 16);
            } else {
                const int tap = s - 5, ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    bf[pb] = *(const f16x8*)(buf + lh * Cfg::IN_PLANE + px_base + ((pb * STRIDE + ky) * Cfg::TWP + kx) * 16);
#pragma unroll
                for (int cb = 0; cb < COB; ++cb)
                    af[cb] = *(const f16x8*)(wb + co_base + ((tap * 2 + (1 - lh)) * Cfg::COUTP + cb * 32) * 16);
            }
        };
        auto mma_step = [&](int s, f16x8 (&bf)[PB], f16x8 (&af)[COB]) {
            if (s < 5) {
#pragma unroll
                for (int cb = 0; cb < COB; ++cb) af[cb] = af[cb] * (f16x8)pair_scale[s];     // exact 2^11 (or 0)
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int cb = 0; cb < COB; ++cb)
                    acc[pb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb], bf[pb], acc[pb][cb], 0, 0, 0);
        };
        load_step(0, bfA, afA);
#pragma unroll
        for (int s2 = 0; s2 < 14; s2 += 2) {
            load_step(s2 + 1, bfB, afB);
            mma_step(s2, bfA, afA);
            if (s2 + 2 < 14) load_step(s2 + 2, bfA, afA);
            mma_step(s2 + 1, bfB, afB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next group's LDS-DMA has landed
        __syncthreads();
        if (WS && cig + 1 < a.CGin) {                            // every wave is done with the weight slab: refill it
            dma_w(smem_s + Cfg::W_AT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue
    const float* bias = (const float*)(a.wpk + (size_t)a.CGin * w_step);
    const bool relu = a.flags & 1, f32out = (a.flags & 32) || SHUF;
    const int Ho = a.Ho, Wo = a.Wo;
    const size_t HWo = (size_t)Ho * Wo;
    const int x = x0 + li;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int y = y0 + PB * wv + pb;
        if (SHUF && (a.flags & 64)) {
            // PixelShuffle(2) store straight into c8s, with the skip tensor added (flag bit1): a lane holds, per conv
            // channel group g, the four sub-pixels e of OUTPUT channel 2g+lh; v_permlane32_swap exchanges the bottom-row
            // values of lane (li, 0) with the top-row values of lane (li, 1), after which lane (li, lh) owns all eight
            // channels of output group (split*COB+cb) at the two pixels (2y+lh, 2x), (2y+lh, 2x+1).
            if (y < Ho && x < Wo) {
                const int CGs = a.CGout >> 2;
                const size_t HWs = 4 * HWo;
#pragma unroll
                for (int cb = 0; cb < COB; ++cb) {
                    const int og = split * COB + cb;
                    if (og < CGs) {
                        float o8[2][8];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 bs = *(const f32x4*)(bias + (og * 4 + g) * 8 + 4 * lh);
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = acc[pb][cb][4 * g + e] * CS_LO_INV + bs[e];
                                if (relu) v[e] = fmaxf(v[e], 0.f);
                            }
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[dx]), __float_as_uint(v[2 + dx]),
                                                                                false, false);
                                o8[dx][2 * g] = __uint_as_float(r[0]);
                                o8[dx][2 * g + 1] = __uint_as_float(r[1]);
                            }
                        }
                        const size_t pix = (size_t)(2 * y + lh) * (2 * Wo) + 2 * x;
                        const size_t grp = ((size_t)n * CGs + og) * (2 * HWs * 16);
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            if (a.flags & 2) {
                                const char* rg = a.res + grp + (pix + dx) * 16;
                                const f16x8 rh = *(const f16x8*)rg, rl = *(const f16x8*)(rg + HWs * 16);
#pragma unroll
                                for (int c = 0; c < 8; ++c) o8[dx][c] = o8[dx][c] + ((float)rh[c] + (float)rl[c] * CS_LO_INV);
                            }
                            char* og_ptr = a.out + grp + (pix + dx) * 16;
                            split_store(o8[dx][0], o8[dx][1], o8[dx][2], o8[dx][3], og_ptr, og_ptr + HWs * 16, a.ovf);
                            split_store(o8[dx][4], o8[dx][5], o8[dx][6], o8[dx][7], og_ptr + 8, og_ptr + HWs * 16 + 8, a.ovf);
                        }
                    }
                }
            }
        } else if (y < Ho && x < Wo) {
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cog = (split * COB + cb) * 4 + g;
                    if (cog < a.CGout) {
                        const f32x4 bs = *(const f32x4*)(bias + cog * 8 + 4 * lh);
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[pb][cb][4 * g + e] * CS_LO_INV + bs[e];
                            if (relu) v[e] = fmaxf(v[e], 0.f);
                        }
                        const size_t pix = (size_t)y * Wo + x;
                        if (a.flags & 2) {
                            const char* rg = a.res + ((size_t)n * a.CGout + cog) * (2 * HWo * 16) + pix * 16 + 8 * lh;
                            const f16x4 rh = *(const f16x4*)rg, rl = *(const f16x4*)(rg + HWo * 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] + ((float)rh[e] + (float)rl[e] * CS_LO_INV);
                        }
                        if (a.flags & 16) {
                            // ReLU'(activation): the stashed activation is >= 0, positive iff one of its halves is
                            const char* mg = a.mask + ((size_t)n * a.CGout + cog) * (2 * HWo * 16) + pix * 16 + 8 * lh;
                            const f16x4 mh = *(const f16x4*)mg, ml = *(const f16x4*)(mg + HWo * 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (!((float)mh[e] > 0.f || (float)ml[e] > 0.f)) v[e] = 0.f;
                        }
                        if (SHUF) {
                            // PixelShuffle(2): conv channel 8*cog + 4*lh + e -> channel c = 2*cog + lh of pixel
                            // (2y + (e>>1), 2x + (e&1)); fp32 c8 tensor [n][CGout/4][2Ho][2Wo][8]
                            const int CGs = a.CGout >> 2;
                            float* o32 = (float*)a.out;
                            const size_t base = (((size_t)n * CGs + (cog >> 2)) * (2 * Ho) + 2 * y) * (size_t)(2 * Wo) * 8 +
                                                (size_t)(2 * x) * 8 + (cog & 3) * 2 + lh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o32[base + (size_t)(e >> 1) * (2 * Wo) * 8 + (e & 1) * 8] = v[e];
                        } else if (f32out) {
                            f32x4 o = {v[0], v[1], v[2], v[3]};
                            *(f32x4*)(a.out + ((((size_t)n * a.CGout + cog) * HWo + pix) * 8 + 4 * lh) * 4) = o;
                        } else {
                            // c8s store: this lane holds channels 4*lh .. 4*lh+3 of the group, lane li+32 the other four.  The two
                            // exchange one half each (v_permlane32_swap), after which lane (li, 0) owns the eight hi values and
                            // lane (li, 1) the eight lo' values of the pixel: ONE 16-byte store per lane and plane instead of two
                            // 8-byte ones (8-byte vector stores run at 0.54-0.70 of the 16-byte rate; the head layers are store-bound)
                            char* grp = a.out + ((size_t)n * a.CGout + cog) * (2 * HWo * 16);
                            split_store16(v[0], v[1], v[2], v[3], grp + (lh ? HWo * 16 : 0) + pix * 16, a.ovf);
                        }
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same convolution on v_mfma_f32_16x16x32_f16 (stride 1, c8s / fp32 c8 store): K = 32 holds FOUR 8-channel
// sub-blocks, one per 16-lane group ks of the wave, each with its own (tap, operand plane) pair:
//   steps 0,1 : hi x hi of taps 4s+ks                                   (A = w_hi * 2^11, B = x_hi)
//   steps 2-5 : cross terms, sub-block j = 4(s-2)+ks -> tap j/2:  even j  w_lo' x x_hi,  odd j  w_hi x x_lo'
//   step  6   : ks 0,1 cross terms of tap 8;  ks 2 hi x hi of tap 8;  ks 3 empty (A scaled by 0)
// = 7 steps x 32 = 224 K-slots per 8 input channels, exactly the 14 x 16 of the 32x32x16 kernel; the wave tile
// (32*COB channels x 2 rows x 32 columns) is 2*COB x 4 blocks of 16x16, 4 accumulator registers each.  Same LDS image,
// same packed weights, same bytes read per flop.  On random data the chip holds a higher clock on this MFMA shape
// (MI355X_MICROARCH.md, DVFS give-back (7)).
template <int COB, int TAG>
__global__ void __launch_bounds__(256, (SplitCfg<COB, 1>::WAVES_PER_SIMD))
conv3x3_c8s_k32_kernel(const SplitArgs a) {
    using Cfg = SplitCfg<COB, 1>;
    constexpr int CS_THREADS = 256, NCB = 2 * COB;
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int lc = lane & 15, ks = lane >> 4;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the linear id is
    // remapped such that one XCD works through a CONTIGUOUS run of tiles (for the FFDNet body: one frame per XCD) and
    // the halo rows shared by vertically adjacent tiles hit in that XCD's L2 instead of being fetched by another one
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
        if (!(a.flags & 0x800) && (total & 7) == 0) {
            unsigned w = bx + nx * (by + ny * bz);
            w = (w & 7) * (total >> 3) + (w >> 3);
            bx = w % nx; by = (w / nx) % ny; bz = w / (nx * ny);
        }
    }
    const int x0 = bx * CS_TW, y0 = by * Cfg::TH;
    const int n = bz / a.nsplit, split = bz % a.nsplit;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const size_t grp_bytes = 2 * HW * 16;

    const unsigned OOB = 0x80000000u;
    unsigned in_off[Cfg::IN_ITERS];
#pragma unroll
    for (int k = 0; k < Cfg::IN_ITERS; ++k) {
        const int e = tid + k * CS_THREADS;
        in_off[k] = OOB;
        if (e < Cfg::IN_VEC) {
            const int plane = e / (Cfg::THP * Cfg::TWP), pix = e - plane * (Cfg::THP * Cfg::TWP);
            const int r = pix / Cfg::TWP, c = pix - r * Cfg::TWP;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) in_off[k] = (unsigned)((plane * HW + (size_t)gy * W + gx) * 16);
        }
    }
    unsigned w_off[Cfg::W_ITERS];
#pragma unroll
    for (int k = 0; k < Cfg::W_ITERS; ++k) {
        const int e = tid + k * CS_THREADS;
        w_off[k] = OOB;
        if (e < Cfg::W_VEC) {
            const int tp = e / Cfg::COUTP, co = e - tp * Cfg::COUTP;
            w_off[k] = (unsigned)((tp * a.CoutP_total + split * Cfg::COUTP + co) * 16);
        }
    }
    const char* in_g = a.in + (size_t)n * a.CGin * grp_bytes;
    const char* w_g = a.wpk;
    const size_t w_step = (size_t)9 * 2 * a.CoutP_total * 16;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    (void)wvu; (void)in_g; (void)w_g;      // only read by the device-only staging code below
    auto dma_stage = [&](char* buf) {
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, (int)grp_bytes, 0x00020000);
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, (int)w_step, 0x00020000);
#pragma unroll
        for (int k = 0; k < Cfg::IN_ITERS; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_in, (__attribute__((address_space(3))) void*)(buf + 16 * (wvu * 64 + k * CS_THREADS)), 16, in_off[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < Cfg::W_ITERS; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                r_w, (__attribute__((address_space(3))) void*)(buf + Cfg::IN_PAD + 16 * (wvu * 64 + k * CS_THREADS)), 16,
                w_off[k], 0, 0, 0);
#endif
        in_g += grp_bytes;
        w_g += w_step;
    };

    f32x4 acc[4][NCB];                                   // [pixel block = 2*pb + half][16-channel block]
#pragma unroll
    for (int pq = 0; pq < 4; ++pq)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[pq][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    dma_stage(smem_s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-step, per-lane-group operand selection (tap, A plane, B plane, A scale)
    int offA[7], offB[7];
    _Float16 sc6 = (_Float16)1.f;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        int tap, pa, pbl;
        if (s < 2) { tap = 4 * s + ks; pa = 0; pbl = 0; }
        else if (s < 6) { const int j = 4 * (s - 2) + ks; tap = j >> 1; pa = (j & 1) ? 0 : 1; pbl = (j & 1) ? 1 : 0; }
        else { tap = 8; pa = (ks == 0) ? 1 : 0; pbl = (ks == 1) ? 1 : 0; }
        const int ky = tap / 3, kx = tap - 3 * ky;
        offA[s] = Cfg::IN_PAD + ((tap * 2 + pa) * Cfg::COUTP + lc) * 16;                               // + cb*16*16
        offB[s] = pbl * Cfg::IN_PLANE + ((2 * wv + ky) * Cfg::TWP + lc + kx) * 16;                      // + (pb*TWP + 16*half)*16
    }
    if (ks == 2) sc6 = (_Float16)CS_LO_SCALE;
    if (ks == 3) sc6 = (_Float16)0.f;

    for (int cig = 0; cig < a.CGin; ++cig) {
        const char* buf = smem_s + (cig & 1) * Cfg::STAGE;
        if (cig + 1 < a.CGin) dma_stage(smem_s + ((cig + 1) & 1) * Cfg::STAGE);
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            f16x8 bf[4], af[NCB];
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) bf[pq] = *(const f16x8*)(buf + offB[s] + ((pq >> 1) * Cfg::TWP + 16 * (pq & 1)) * 16);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) af[cb] = *(const f16x8*)(buf + offA[s] + cb * 16 * 16);
            if (s < 2) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) af[cb] = af[cb] * (f16x8)(_Float16)CS_LO_SCALE;
            } else if (s == 6) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) af[cb] = af[cb] * (f16x8)sc6;
            }
#pragma unroll
            for (int pq = 0; pq < 4; ++pq)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    acc[pq][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[cb], bf[pq], acc[pq][cb], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane = pixel column lc of its block, channels 16*cb + 4*ks .. +3
    const float* bias = (const float*)(a.wpk + (size_t)a.CGin * w_step);
    const bool relu = a.flags & 1, f32out = a.flags & 32;
    const int Ho = a.Ho, Wo = a.Wo;
    const size_t HWo = (size_t)Ho * Wo;
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
        const int y = y0 + 2 * wv + (pq >> 1), x = x0 + 16 * (pq & 1) + lc;
        if (y < Ho && x < Wo) {
            const size_t pix = (size_t)y * Wo + x;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int cog = (split * COB) * 4 + 2 * cb + (ks >> 1), sub = 4 * (ks & 1);
                if (cog < a.CGout) {
                    const f32x4 bs = *(const f32x4*)(bias + cog * 8 + sub);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[pq][cb][e] * CS_LO_INV + bs[e];
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (a.flags & 2) {
                        const char* rg = a.res + ((size_t)n * a.CGout + cog) * (2 * HWo * 16) + pix * 16 + 2 * sub;
                        const f16x4 rh = *(const f16x4*)rg, rl = *(const f16x4*)(rg + HWo * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] + ((float)rh[e] + (float)rl[e] * CS_LO_INV);
                    }
                    if (a.flags & 16) {
                        const char* mg = a.mask + ((size_t)n * a.CGout + cog) * (2 * HWo * 16) + pix * 16 + 2 * sub;
                        const f16x4 mh = *(const f16x4*)mg, ml = *(const f16x4*)(mg + HWo * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!((float)mh[e] > 0.f || (float)ml[e] > 0.f)) v[e] = 0.f;
                    }
                    if (f32out) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *(f32x4*)(a.out + ((((size_t)n * a.CGout + cog) * HWo + pix) * 8 + sub) * 4) = o;
                    } else {
                        char* grp = a.out + ((size_t)n * a.CGout + cog) * (2 * HWo * 16);
                        split_store(v[0], v[1], v[2], v[3], grp + pix * 16 + 2 * sub, grp + HWo * 16 + pix * 16 + 2 * sub, a.ovf);
                    }
                }
            }
        }
    }
}

template <int COB, int TAG>
static int launch_split_k32(const SplitArgs& a, int n, hipStream_t st) {
    using Cfg = SplitCfg<COB, 1>;
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_c8s_k32_kernel<COB, TAG>, Cfg::LDS_BYTES, "conv3x3_c8s_k32")) return rc;
    const dim3 grid((a.Wo + CS_TW - 1) / CS_TW, (a.Ho + Cfg::TH - 1) / Cfg::TH, n * a.nsplit);
    hipLaunchKernelGGL((conv3x3_c8s_k32_kernel<COB, TAG>), grid, dim3(256), Cfg::LDS_BYTES, st, a);
    return launch_status("conv3x3_c8s_k32_kernel");
}

template <int COB, int TAG, int STRIDE, int SHUF, int PB = 2, int NW = 4, int WS = 0>
static int launch_split(const SplitArgs& a, int n, hipStream_t st) {
    using Cfg = SplitCfg<COB, STRIDE, PB, NW, WS>;
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_c8s_kernel<COB, TAG, STRIDE, SHUF, PB, NW, WS>, Cfg::LDS_BYTES, "conv3x3_c8s"))
        return rc;
    const dim3 grid((a.Wo + CS_TW - 1) / CS_TW, (a.Ho + Cfg::TH - 1) / Cfg::TH, n * a.nsplit);
    hipLaunchKernelGGL((conv3x3_c8s_kernel<COB, TAG, STRIDE, SHUF, PB, NW, WS>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, a);
    return launch_status("conv3x3_c8s_kernel");
}

template <int STRIDE, int SHUF>
static int dispatch_split(SplitArgs& a, int n, hipStream_t st) {
    const int CoutP = a.CoutP_total;
    // stride-1 layers (plain or PixelShuffle store) run in the single-buffered-weights form, three workgroups per CU
    // (2-16 % ahead of the fully double-buffered two-workgroup form across the layer shapes and boxes measured; flag
    // 0x200 selects the latter for tools/conv_bench.py); 128-channel multiples as two 64-channel halves (the 4-block
    // form needs 288 VGPRs = one workgroup per CU).  Stride-2 layers keep the double-buffered form (their 17 x 65
    // input tile leaves room for one workgroup either way).
    constexpr int WS = STRIDE == 1 ? 1 : 0;
    const bool ws = WS && !(a.flags & 0x200);
    if (CoutP % 96 == 0) {
        a.nsplit = CoutP / 96;
        if (STRIDE == 1 && !SHUF && (a.flags & 0x100)) return ws ? launch_split<3, 1, 1, 0, 2, 4, 1>(a, n, st) : launch_split<3, 1, 1, 0>(a, n, st);
        if (STRIDE == 1 && !SHUF && (a.flags & 0x400)) return launch_split_k32<3, 0>(a, n, st);     // 16x16x32 MFMA form
        if (ws) return launch_split<3, 0, STRIDE, SHUF, 2, 4, WS>(a, n, st);
        return launch_split<3, 0, STRIDE, SHUF>(a, n, st);
    }
    if (CoutP % 64 == 0) {
        if (ws) { a.nsplit = CoutP / 64; return launch_split<2, 0, STRIDE, SHUF, 2, 4, WS>(a, n, st); }
        if (CoutP % 128 == 0) { a.nsplit = CoutP / 128; return launch_split<4, 0, STRIDE, SHUF>(a, n, st); }
        a.nsplit = CoutP / 64;
        return launch_split<2, 0, STRIDE, SHUF>(a, n, st);
    }
    a.nsplit = CoutP / 32;
    if (ws) return launch_split<1, 0, STRIDE, SHUF, 2, 4, WS>(a, n, st);
    return launch_split<1, 0, STRIDE, SHUF>(a, n, st);
}

static inline int round_up_s(int v, int m) { return (v + m - 1) / m * m; }

// fp32 c8 (+ optional c8s residual) -> c8s
__global__ void __launch_bounds__(256)
c8_to_c8s_kernel(const float* __restrict__ in, const char* __restrict__ res, char* __restrict__ out, size_t HW,
                 size_t total /* n*CG*HW */, int* ovf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t grp = i / HW, pix = i - grp * HW;
    f32x4 a = *(const f32x4*)(in + i * 8), b = *(const f32x4*)(in + i * 8 + 4);
    char* g = out + grp * (2 * HW * 16);
    if (res) {
        const f16x8 rh = *(const f16x8*)(res + grp * (2 * HW * 16) + pix * 16);
        const f16x8 rl = *(const f16x8*)(res + grp * (2 * HW * 16) + HW * 16 + pix * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = a[e] + ((float)rh[e] + (float)rl[e] * CS_LO_INV);
            b[e] = b[e] + ((float)rh[4 + e] + (float)rl[4 + e] * CS_LO_INV);
        }
    }
    split_store(a[0], a[1], a[2], a[3], g + pix * 16, g + HW * 16 + pix * 16, ovf);
    split_store(b[0], b[1], b[2], b[3], g + pix * 16 + 8, g + HW * 16 + pix * 16 + 8, ovf);
}

// device-side packing of (updated) fp32 master weights for the online finetune: forward layout, or the backward-data
// convolution W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] (no bias).  One thread per (cig, tap, co, c8) element pair.
__global__ void __launch_bounds__(256)
pack_split_device_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ scale,
                         char* __restrict__ packed, int Cin_real, int Cout_real, int Kin, int KoutP, int transpose, int* ovf) {
    const size_t nw = (size_t)(Kin / 8) * 9 * KoutP * 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw + KoutP) return;
    if (i >= nw) {
        const int o = (int)(i - nw);
        ((float*)(packed + nw * 4))[o] = (!transpose && bias && o < Cout_real) ? bias[o] : 0.f;
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
        if (o < Cin_real && in_ch < Cout_real) {
            val = w[((size_t)in_ch * Cin_real + o) * 9 + (8 - tap)];
            if (scale) val = val * scale[in_ch];
        }
    }
    if (!(fabsf(val) < 31.9f)) *ovf = 1;
    const _Float16 hi = (_Float16)val;
    const _Float16 lo = (_Float16)((val - (float)hi) * CS_LO_SCALE);
    _Float16* p = (_Float16*)packed;
    const size_t base = ((size_t)ig * 9 + tap) * 2;
    p[((base + 0) * KoutP + o) * 8 + c8] = hi;
    p[((base + 1) * KoutP + o) * 8 + c8] = lo;
}

// c8s -> fp32 c8 with an exact power-of-two rescale (gradients travel through the split kernels pre-scaled)
__global__ void __launch_bounds__(256)
c8s_to_c8_kernel(const char* __restrict__ in, float* __restrict__ out, size_t HW, size_t total, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t grp = i / HW, pix = i - grp * HW;
    const f16x8 h = *(const f16x8*)(in + grp * (2 * HW * 16) + pix * 16);
    const f16x8 l = *(const f16x8*)(in + grp * (2 * HW * 16) + HW * 16 + pix * 16);
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = ((float)h[e] + (float)l[e] * CS_LO_INV) * scale;
        b[e] = ((float)h[4 + e] + (float)l[4 + e] * CS_LO_INV) * scale;
    }
    *(f32x4*)(out + i * 8) = a;
    *(f32x4*)(out + i * 8 + 4) = b;
}

__global__ void __launch_bounds__(256)
c8_scale_to_c8s_kernel(const float* __restrict__ in, char* __restrict__ out, size_t HW, size_t total, float scale, int* ovf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t grp = i / HW, pix = i - grp * HW;
    const f32x4 a = *(const f32x4*)(in + i * 8) * scale, b = *(const f32x4*)(in + i * 8 + 4) * scale;
    char* g = out + grp * (2 * HW * 16);
    split_store(a[0], a[1], a[2], a[3], g + pix * 16, g + HW * 16 + pix * 16, ovf);
    split_store(b[0], b[1], b[2], b[3], g + pix * 16 + 8, g + HW * 16 + pix * 16 + 8, ovf);
}

// the calling thread's bound word (scipnp_bind_overflow_word), else the process-wide one of the current device
static thread_local int* t_ovf_word = nullptr;

int* exchange_overflow_word(int* w) {
    int* prev = t_ovf_word;
    t_ovf_word = w;
    return prev;
}

static int* current_ovf_word() {
    if (t_ovf_word) return t_ovf_word;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_split_overflow)) != hipSuccess) return nullptr;
    return (int*)p;
}

}  // namespace scipnp


using namespace scipnp;

extern "C" {

int scipnp_c8_add_to_c8s(const float* in_c8, const void* residual_c8s, void* out_c8s, int n, int C, int h, int w,
                         scipnp_stream_t s);
int scipnp_pack_conv3x3_split_device_scaled(const float* w, const float* bias, const float* scale, void* packed, int Cin_real,
                                            int Cout_real, int Cin, int Cout, int transpose_flip, scipnp_stream_t s);
int scipnp_conv3x3_c8s_ex(const void* in_c8s, const void* packed_split, void* out, const void* residual_c8s,
                          const void* mask_c8s, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);
int scipnp_ffdnet_forward_c8s_2s(const void* in_c8s, float* out_c8, const void* const* packed_split, int nb, int nc,
                                 void* scratch0, void* scratch1, int B, int M, int N, scipnp_stream_t s,
                                 scipnp_stream_t side_stream, void* fork_event, void* join_event);

/* scipnp_conv3x3_split_packed_bytes / scipnp_pack_conv3x3_split[_bn] (host functions): csrc/host_pack.hip */

int scipnp_conv3x3_c8s(const void* in_c8s, const void* packed_split, void* out, int n, int Cin, int Cout, int h, int w,
                       int flags, scipnp_stream_t s) {
    SCIPNP_REQUIRE(!(flags & (16 | 2)), "ReLU-mask / residual epilogues need scipnp_conv3x3_c8s_ex");
    return scipnp_conv3x3_c8s_ex(in_c8s, packed_split, out, nullptr, nullptr, n, Cin, Cout, h, w, flags, s);
}

int scipnp_conv3x3_c8s_ex(const void* in_c8s, const void* packed_split, void* out, const void* residual_c8s,
                          const void* mask_c8s, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in_c8s && packed_split && out, "null pointer");
    SCIPNP_REQUIRE(!(flags & 16) || (mask_c8s && !(flags & (4 | 8 | 32))), "flag bit4 needs mask_c8s and a c8s stride-1 output");
    SCIPNP_REQUIRE(!(flags & 2) || (residual_c8s && !(flags & 4) && (!(flags & 8) || (flags & 64))),
                   "flag bit1 needs residual_c8s and a stride-1 output (with pixel shuffle: the c8s store, bit6)");
    SCIPNP_REQUIRE(!(flags & 64) || ((flags & 8) && !(flags & 32)), "flag bit6 (c8s pixel-shuffle store) needs bit3 and excludes bit5");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad shape");
    SCIPNP_ALIGNED(in_c8s); SCIPNP_ALIGNED(packed_split); SCIPNP_ALIGNED(out);
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 31), "image too large for 32-bit tile offsets");
    const bool stride2 = flags & 4, shuf = flags & 8;
    SCIPNP_REQUIRE(!(stride2 && shuf), "stride-2 and pixel-shuffle epilogue cannot be combined");
    SCIPNP_REQUIRE(!shuf || Cout % 32 == 0, "pixel-shuffle epilogue needs Cout %% 32 == 0 (got %d)", Cout);
    SplitArgs a;
    a.in = (const char*)in_c8s; a.wpk = (const char*)packed_split; a.out = (char*)out; a.mask = (const char*)mask_c8s; a.res = (const char*)residual_c8s;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.CoutP_total = round_up_s(Cout, 32); a.nsplit = 1;
    a.H = h; a.W = w;
    a.Ho = stride2 ? (h - 1) / 2 + 1 : h;
    a.Wo = stride2 ? (w - 1) / 2 + 1 : w;
    a.flags = flags;
    a.ovf = current_ovf_word();
    SCIPNP_REQUIRE(a.ovf, "no range-guard word (hipGetSymbolAddress failed)");
    hipStream_t st = (hipStream_t)s;
    SCIPNP_REQUIRE((long long)n * (a.CoutP_total / 32) <= 65535, "grid too large");
    if (stride2) return dispatch_split<2, 0>(a, n, st);
    if (shuf) return dispatch_split<1, 1>(a, n, st);
    return dispatch_split<1, 0>(a, n, st);
}

int scipnp_bind_overflow_word(int* dev_word) {
    t_ovf_word = dev_word;
    return SCIPNP_OK;
}

int scipnp_read_overflow_word(const int* dev_word, int reset, int* flag_out, scipnp_stream_t s) {
    // synchronises the stream (one call per reconstruction, next to the final read-back)
    hipStream_t st = (hipStream_t)s;
    int* word = dev_word ? (int*)dev_word : current_ovf_word();
    SCIPNP_REQUIRE(word, "no range-guard word");
    int v = 0;
    hipError_t e = hipMemcpyAsync(&v, word, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(SCIPNP_EHIP, "scipnp_read_overflow_word: %s", hipGetErrorString(e));
    if (flag_out) *flag_out = v;
    if (reset && v) {
        e = hipMemsetAsync(word, 0, sizeof(int), st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(SCIPNP_EHIP, "scipnp_read_overflow_word reset: %s", hipGetErrorString(e));
    }
    return SCIPNP_OK;
}

int scipnp_split_overflow(int reset, int* flag_out, scipnp_stream_t s) {
    return scipnp_read_overflow_word(nullptr, reset, flag_out, s);
}

int scipnp_c8_to_c8s(const float* in_c8, void* out_c8s, int n, int C, int h, int w, scipnp_stream_t s) {
    return scipnp_c8_add_to_c8s(in_c8, nullptr, out_c8s, n, C, h, w, s);
}

int scipnp_c8_add_to_c8s(const float* in_c8, const void* residual_c8s, void* out_c8s, int n, int C, int h, int w,
                         scipnp_stream_t s) {
    SCIPNP_REQUIRE(in_c8 && out_c8s && n > 0 && C % 8 == 0 && h > 0 && w > 0, "bad arguments");
    SCIPNP_ALIGNED(in_c8); SCIPNP_ALIGNED(out_c8s);
    if (residual_c8s) SCIPNP_ALIGNED(residual_c8s);
    const size_t HW = (size_t)h * w, total = (size_t)n * (C / 8) * HW;
    int* ovf = current_ovf_word();
    SCIPNP_REQUIRE(ovf, "no range-guard word");
    hipLaunchKernelGGL(c8_to_c8s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, in_c8,
                       (const char*)residual_c8s, (char*)out_c8s, HW, total, ovf);
    return launch_status("c8_to_c8s_kernel");
}

int scipnp_pack_conv3x3_split_device(const float* w, const float* bias, void* packed, int Cin_real, int Cout_real, int Cin,
                                     int Cout, int transpose_flip, scipnp_stream_t s) {
    return scipnp_pack_conv3x3_split_device_scaled(w, bias, nullptr, packed, Cin_real, Cout_real, Cin, Cout, transpose_flip, s);
}

int scipnp_pack_conv3x3_split_device_scaled(const float* w, const float* bias, const float* scale, void* packed, int Cin_real,
                                            int Cout_real, int Cin, int Cout, int transpose_flip, scipnp_stream_t s) {
    SCIPNP_REQUIRE(w && packed, "null pointer");
    SCIPNP_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && Cin_real > 0 && Cout_real > 0 && Cin_real <= Cin && Cout_real <= Cout,
                   "bad channel counts");
    const int Kin = transpose_flip ? Cout : Cin;
    const int KoutP = round_up_s(transpose_flip ? Cin : Cout, 32);
    const size_t total = (size_t)(Kin / 8) * 9 * KoutP * 8 + KoutP;
    int* ovf = current_ovf_word();
    SCIPNP_REQUIRE(ovf, "no range-guard word");
    hipLaunchKernelGGL(pack_split_device_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, bias,
                       scale, (char*)packed, Cin_real, Cout_real, Kin, KoutP, transpose_flip, ovf);
    return launch_status("pack_split_device_kernel");
}

int scipnp_c8s_to_c8(const void* in_c8s, float* out_c8, float scale, int n, int C, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in_c8s && out_c8 && n > 0 && C % 8 == 0 && C > 0 && h > 0 && w > 0, "bad arguments");
    SCIPNP_ALIGNED(in_c8s); SCIPNP_ALIGNED(out_c8);
    const size_t HW = (size_t)h * w, total = (size_t)n * (C / 8) * HW;
    hipLaunchKernelGGL(c8s_to_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       (const char*)in_c8s, out_c8, HW, total, scale);
    return launch_status("c8s_to_c8_kernel");
}

int scipnp_c8_scale_to_c8s(const float* in_c8, void* out_c8s, float scale, int n, int C, int h, int w, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in_c8 && out_c8s && n > 0 && C % 8 == 0 && C > 0 && h > 0 && w > 0, "bad arguments");
    SCIPNP_ALIGNED(in_c8); SCIPNP_ALIGNED(out_c8s);
    const size_t HW = (size_t)h * w, total = (size_t)n * (C / 8) * HW;
    int* ovf = current_ovf_word();
    SCIPNP_REQUIRE(ovf, "no range-guard word");
    hipLaunchKernelGGL(c8_scale_to_c8s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, in_c8,
                       (char*)out_c8s, HW, total, scale, ovf);
    return launch_status("c8_scale_to_c8s_kernel");
}

// frames [n0, n0 + nf) of the pass on stream st
static int ffdnet_c8s_frames(const char* in_c8s, float* out_c8, const void* const* packed_split, int nb, int nc, char* s0,
                             char* s1, int n0, int nf, int M, int N, scipnp_stream_t st) {
    const size_t px = (size_t)M * N;
    const size_t in_f = 2 * px * 32, act_f = (size_t)(nc / 8) * px * 32, out_f = 2 * px * 8;      // bytes, bytes, floats per frame
    char* buf[2] = {s0 + n0 * act_f, s1 + n0 * act_f};
    int rc = scipnp_conv3x3_c8s(in_c8s + n0 * in_f, packed_split[0], buf[0], nf, 16, nc, M, N, 1 | 0x100, st);
    if (rc) return rc;
    int cur = 0;
    for (int l = 1; l < nb - 1; ++l) {
        rc = scipnp_conv3x3_c8s(buf[cur], packed_split[l], buf[cur ^ 1], nf, nc, nc, M, N, 1, st);
        if (rc) return rc;
        cur ^= 1;
    }
    return scipnp_conv3x3_c8s(buf[cur], packed_split[nb - 1], out_c8 + n0 * out_f, nf, nc, 16, M, N, 32, st);
}

int scipnp_ffdnet_forward_c8s(const void* in_c8s, float* out_c8, const void* const* packed_split, int nb, int nc,
                              void* scratch0, void* scratch1, int B, int M, int N, scipnp_stream_t s) {
    return scipnp_ffdnet_forward_c8s_2s(in_c8s, out_c8, packed_split, nb, nc, scratch0, scratch1, B, M, N, s, nullptr, nullptr,
                                        nullptr);
}

// With a side stream: the second half of the frames runs there, forked from and joined to the caller's stream by the
// caller's two events (legal under hipGraph capture: the side stream joins before the call returns) -- its launches fill
// the CUs the last generation of the other half leaves idle.  The library creates no streams or events of its own.
int scipnp_ffdnet_forward_c8s_2s(const void* in_c8s, float* out_c8, const void* const* packed_split, int nb, int nc,
                                 void* scratch0, void* scratch1, int B, int M, int N, scipnp_stream_t s,
                                 scipnp_stream_t side_stream, void* fork_event, void* join_event) {
    SCIPNP_REQUIRE(in_c8s && out_c8 && packed_split && scratch0 && scratch1, "null pointer");
    SCIPNP_REQUIRE(nb >= 2 && nc % 8 == 0 && nc > 0 && B > 0, "bad network shape nb=%d nc=%d B=%d", nb, nc, B);
    SCIPNP_REQUIRE(!side_stream || (fork_event && join_event), "a side stream needs the caller's fork and join events");
    const char* in = (const char*)in_c8s;
    char *s0 = (char*)scratch0, *s1 = (char*)scratch1;
    if (!side_stream || B < 2)
        return ffdnet_c8s_frames(in, out_c8, packed_split, nb, nc, s0, s1, 0, B, M, N, s);
    const int h0 = (B + 1) / 2;
    hipStream_t cur = (hipStream_t)s, side = (hipStream_t)side_stream;
    hipEvent_t fork = (hipEvent_t)fork_event, join = (hipEvent_t)join_event;
    if (hipEventRecord(fork, cur) != hipSuccess || hipStreamWaitEvent(side, fork, 0) != hipSuccess)
        return fail(SCIPNP_EHIP, "fork of the side stream: %s", hipGetErrorString(hipGetLastError()));
    const int rc = ffdnet_c8s_frames(in, out_c8, packed_split, nb, nc, s0, s1, h0, B - h0, M, N, side_stream);
    const int rc0 = ffdnet_c8s_frames(in, out_c8, packed_split, nb, nc, s0, s1, 0, h0, M, N, s);
    // join even after a failed launch: the caller's stream must not run ahead of work already queued on the side stream
    if (hipEventRecord(join, side) != hipSuccess || hipStreamWaitEvent(cur, join, 0) != hipSuccess)
        return fail(SCIPNP_EHIP, "join of the side stream: %s", hipGetErrorString(hipGetLastError()));
    return rc ? rc : rc0;
}

}  // extern "C"
