// 3x3 / stride-1 / pad-1 convolution in fp32 arithmetic as Winograd F(2x2, 3x3) on the CDNA4 matrix cores.
//
//   Y(2x2) = A^T [ (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 products per output tile and channel pair instead of 36: 2.25x fewer multiply-adds than the direct form of
// conv.hip, every one of them an exact fp32 product accumulated in fp32 on v_mfma_f32_16x16x4_f32 (the fp32 MFMA runs
// at the fp32 VECTOR rate on gfx950, so the input / output transforms -- O(C) adds per tile against O(C^2) products --
// cost a few per cent; this is the same trade the reference's cuDNN makes for fp32 3x3 convolutions).  The filter
// transform U = G g G^T is done once per weight update, in double, by scipnp_pack_conv3x3_wino.
//
// GEMM view, per Winograd position p = (xi, nu) of 16:   M_p[co][tile] = sum_ci U_p[co][ci] * V_p[ci][tile]
//   A = U_p   16 (co) x 4 (ci)     lane l holds A[l & 15][l >> 4]
//   B = V_p    4 (ci) x 16 (tile)  lane l holds B[l >> 4][l & 15]
//   D         16 (co) x 16 (tile)  lane l holds D[4*(l >> 4) + r][l & 15], r = 0..3
// A lane therefore owns ONE tile and, per 8-channel group, the channel pair {2q, 2q+1}, q = l >> 4: it reads its own
// 4x4 patch of those two channels (16 ds_read_b64), transforms it in registers (32 adds per channel) and feeds the 16
// positions x 2 output-channel halves x 2 k-steps = 64 MFMAs of the group; the 32 accumulators (128 VGPRs) hold all 16
// positions of 32 output channels x 16 tiles, so the output transform happens once, after the K loop.
//
// Work decomposition: workgroup = 8 waves = 16 rows x 32 columns of output pixels x 32 output channels; wave w owns the
// tile row w (output rows 2w, 2w+1; 16 tiles along x).  Output-channel blocks of 32 are separate workgroups placed next
// to each other on ONE XCD (its L2 serves the re-read of the input tile).  K loop over input channel groups of 8; per
// group, double-buffered in LDS:
//   raw input halo tile ((rows + 2) x 34 pixels x 8 channels) -> four channel-pair planes [q][rows + 2][34][2], global ->
//     registers -> ds_write_b64; a lane reads a patch row (4 pixels x 2 channels, 32 contiguous bytes) as two ds_read_b128;
//   U slab (16 positions x 32 co x 8 ci = 16 KiB, laid out [xi][co half h][k-step j][lane][nu] by the packer: one conflict-free
//     ds_read_b128 per lane feeds four MFMAs on the four accumulators (xi, nu = 0..3) of one half and k-step) by LDS-DMA
//     (buffer_load_dwordx4 ... lds), no staging registers.
#include "common.hpp"
#ifdef SCIPNP_DIAG_BUILD
#include "../../include/scipnp_diag.h"
#endif

namespace scipnp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// a - b on a float2 as ONE v_pk_add_f32 with a negated operand: the compiler selects a float2 subtraction as two v_sub_f32
// (it turns a + (-b) back into a subtraction first), and every vector instruction beside v_mfma_f32_16x16x4_f32 is matrix
// time lost (tools/probes/mfma_valu_coissue.py).  Same IEEE result.
__device__ __forceinline__ f32x2 psub(f32x2 a, f32x2 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - b;
#endif
}

constexpr int WN_STAMP_WORDS = 80;                    // per workgroup, diagnostic instantiation only
constexpr int WN_SLAB = 16 * 32 * 8;                  // 4096 floats = 16 KiB: U of one (channel group, co block)

// NW = waves per workgroup = tile rows of 2 output rows each: the workgroup covers 2*NW rows x 32 columns x 32 co
template <int NW>
struct WinoCfg {
    static constexpr int TW = 32, TH = 2 * NW;
    static constexpr int THREADS = 64 * NW;
    static constexpr int TWP = TW + 2, THP = TH + 2;           // input halo tile
    static constexpr int PLANE = (THP * TWP * 2 + 63) / 64 * 64; // floats per channel-pair plane; a multiple of 64 keeps the
                                                               // lane groups of the 16-byte patch reads on distinct banks
    static constexpr int RAW = 4 * PLANE;                      // floats (a multiple of 4: every buffer stays 16-B aligned)
    static constexpr int UNITS = THP * TWP * 2;                // 16-byte half pixels of the halo tile
    static constexpr int IN_ITERS = (UNITS + THREADS - 1) / THREADS;
    static constexpr int U_ITERS = (WN_SLAB / 4) / THREADS;    // LDS-DMA pieces of 16 B per lane
    static constexpr size_t LDS_BYTES = (2 * (size_t)RAW + 2 * (size_t)WN_SLAB) * sizeof(float);
};

struct WinoArgs {
    const float* in;
    const float* wpk;        // [CGin][CoutP/32][4096] + bias[CoutP]
    float* out;
    const float* residual;
    const float* mask_src;
    int CGin, CGout, NCB;    // NCB = CoutP / 32
    int H, W;
    int ntx, nty;
    int flags;
    unsigned long long* dbg; // diagnostic build only (STAMP): 8 words per workgroup, see scipnp_conv3x3_c8w_stamped
};

#if defined(__HIP_DEVICE_COMPILE__)
// one clock stamp (cdna_hip_programming.md 7, In-kernel stamps): only ever executed by the STAMP instantiation
#define WINO_STAMP(t)                                                                        \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define WINO_STAMP(t) (t) = 0
#endif

// TAG only changes the symbol name (1 = first / last layer of a network) so profiler statistics of the body layers stay clean.
//
// Software pipeline, one barrier per channel group g:  the MFMAs of group g run on V_g (registers) and U_g (LDS) while
// the same wave reads the raw patch of group g+1 from LDS and transforms it into V_{g+1} between the MFMAs; the raw tile
// of group g+2 travels global -> registers -> LDS and U_{g+1} global -> LDS (LDS-DMA) under the same MFMAs.
// NH = 16-channel halves of the 32-channel output block that hold real channels (1 for layers with <= 16 outputs, e.g.
// the 12-channel FFDNet tail: the padding half is never multiplied).
template <int TAG, int NW, int NH = 2, bool STAMP = false, bool SHUF = false>
__global__ void __launch_bounds__(64 * NW, 2)
conv3x3_c8w_kernel(const WinoArgs a) {
    using K = WinoCfg<NW>;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, rt0 = 0;
    (void)rt0;
    if constexpr (STAMP) {
#if defined(__HIP_DEVICE_COMPILE__)
        rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        WINO_STAMP(ts0);
    }
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    float* const raw_lds = smem_w;                     // [2][RAW]
    float* const u_lds = smem_w + 2 * K::RAW;          // [2][SLAB]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int tn = lane & 15, q = lane >> 4;           // tile along x, channel pair
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; one XCD works through a contiguous run of
    // (tile, co-block) pairs with the co-blocks of a tile adjacent, so the input tile is fetched from HBM once
    unsigned lin = blockIdx.x;
    {
        const unsigned total = gridDim.x;
        if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    }
    const int split = lin % a.NCB;
    unsigned t = lin / a.NCB;
    const int bx = t % a.ntx;
    t /= a.ntx;
    const int by = t % a.nty, n = t / a.nty;
    const int x0 = bx * K::TW, y0 = by * K::TH;

    // ---- staging plan of the raw tile: unit e -> pixel (r, c) of the halo tile, 16-byte half `hf`.  Units past the
    // end of the tile wrap around (a few lanes fetch and store a unit twice, same bytes): no lane is ever masked, so
    // the staging code is straight-line; pixels outside the image get a byte offset past the buffer descriptor's
    // range and read as 0.
    unsigned in_off[K::IN_ITERS];
    int lds_off[K::IN_ITERS];
#pragma unroll
    for (int k = 0; k < K::IN_ITERS; ++k) {
        const int e = (tid + k * K::THREADS) % K::UNITS;
        const int pix = e >> 1, hf = e & 1;
        const int r = pix / K::TWP, c = pix - r * K::TWP;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        lds_off[k] = (2 * hf) * K::PLANE + pix * 2;
        in_off[k] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)((gy * W + gx) * 32 + 16 * hf) : 0xFFFFFF00u;
    }
    const float* in_g = a.in + (size_t)n * a.CGin * HW * 8;                  // advanced by HW*8 per group
    const float* w_g = a.wpk + (size_t)split * WN_SLAB;                      // advanced by NCB*4096 per group
    const size_t w_step = (size_t)a.NCB * WN_SLAB;
    const unsigned plane_bytes = (unsigned)(HW * 32);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    (void)wvu; (void)w_g; (void)plane_bytes; (void)in_g;

    // `last`: this is the last group that exists -- the pointer stays (the pipeline then re-fetches it, unused)
    auto issue_raw = [&](f32x4 (&st)[K::IN_ITERS], bool last) {
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins: keep them out of the host pass that only emits the launch stub
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_g, 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < K::IN_ITERS; ++k)
            st[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_in, in_off[k], 0, 0));
#endif
        if (!last) in_g += HW * 8;
    };
    auto issue_u_piece = [&](float* dst, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_g, 0, WN_SLAB * 4, 0x00020000);
        char* ub = (char*)dst;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            r_w, (__attribute__((address_space(3))) void*)(ub + 16 * (wvu * 64 + k * K::THREADS)), 16,
            (unsigned)(16 * (tid + k * K::THREADS)), 0, 0, 0);
#endif
    };
    auto issue_u = [&](float* dst, bool last) {
#pragma unroll
        for (int k = 0; k < K::U_ITERS; ++k) issue_u_piece(dst, k);
        if (!last) w_g += w_step;
    };
    auto write_raw = [&](float* dst, const f32x4 (&st)[K::IN_ITERS]) {
#pragma unroll
        for (int k = 0; k < K::IN_ITERS; ++k) {
            *(f32x2*)(dst + lds_off[k]) = f32x2{st[k][0], st[k][1]};
            *(f32x2*)(dst + lds_off[k] + K::PLANE) = f32x2{st[k][2], st[k][3]};
        }
    };

    f32x4 acc[16][NH];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS offsets (floats)
    const int b_off = q * K::PLANE + ((2 * wv) * K::TWP + 2 * tn) * 2;              // + (dy*TWP + dx)*2
    const int a_off = lane * 4;                                                     // + vector [xi][h][j] * 256: {nu 0..3}

    const int CG = a.CGin;
    // one patch row of this lane's tile: pixels 2tn .. 2tn+3 of halo row 2wv + dy, channels (2q, 2q+1)
    // B^T d B of one 4x4 patch for the lane's channel PAIR at once: the pair sits in adjacent registers, so every add of the
    // transform is one v_pk_add_f32 -- beside v_mfma_f32_16x16x4_f32 a packed add costs the issue time of a scalar one
    // (tools/probes/mfma_valu_coissue.py), 32 instructions per group instead of 64
    auto load_patch_row = [&](const float* rawp, int dy, f32x2 (&row)[4]) {
        const f32x4 lo = *(const f32x4*)(rawp + b_off + dy * K::TWP * 2), hi = *(const f32x4*)(rawp + b_off + dy * K::TWP * 2 + 4);
        row[0] = f32x2{lo[0], lo[1]}; row[1] = f32x2{lo[2], lo[3]};
        row[2] = f32x2{hi[0], hi[1]}; row[3] = f32x2{hi[2], hi[3]};
    };
    auto transform = [&](f32x2 (&d)[4][4], f32x2 (&Vo)[4][4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x2 t0 = psub(d[0][c], d[2][c]), t1 = d[1][c] + d[2][c], t2 = psub(d[2][c], d[1][c]), t3 = psub(d[1][c], d[3][c]);
            d[0][c] = t0; d[1][c] = t1; d[2][c] = t2; d[3][c] = t3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Vo[r][0] = psub(d[r][0], d[r][2]);
            Vo[r][1] = d[r][1] + d[r][2];
            Vo[r][2] = psub(d[r][2], d[r][1]);
            Vo[r][3] = psub(d[r][1], d[r][3]);
        }
    };
    f32x4 st_in[K::IN_ITERS];
    f32x2 Va[4][4], Vb[4][4];                // B^T d B of the current / next group: [xi][nu] x the channel pair
    {   // prologue: raw tiles of groups 0 and 1, U of group 0; then V_0
        f32x4 st_b[K::IN_ITERS];
        issue_raw(st_in, CG <= 1);
        issue_u(u_lds, CG <= 1);
        issue_raw(st_b, CG <= 2);
        write_raw(raw_lds, st_in);
        write_raw(raw_lds + K::RAW, st_b);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (STAMP) WINO_STAMP(ts1);
        f32x2 d[4][4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) load_patch_row(raw_lds, dy, d[dy]);
        transform(d, Va);
    }

    // The LAST slot of a group is only loaded before the group's barrier; its four MFMAs are issued right behind the barrier,
    // after the first U fragments of the next group have been requested from LDS (NH = 2): the matrix pipe has work while the
    // wave waits for those operands -- the previous group's V lives in the other V buffer until slot 8 of the next group.
    constexpr bool HOLD = NH == 2;
    f32x4 u_held = {0.f, 0.f, 0.f, 0.f};
    // one channel group: MFMAs on (V, U_cig), transform of group cig+1 into Vn, staging of group cig+2 / U_{cig+1}
    auto group_step = [&](int cig, const f32x2 (&V)[4][4], f32x2 (&Vn)[4][4]) {
        const int cur = cig & 1;
        const float* ucur = u_lds + cur * WN_SLAB;
        float* const udst = u_lds + (cur ^ 1) * WN_SLAB;        // U of group cig+1 -> LDS
        const bool ulast = cig + 2 >= CG;
        const float* rnext = raw_lds + (cur ^ 1) * K::RAW;      // raw tile of group cig+1 (stale after the last group: unused)
        issue_raw(st_in, cig + 3 >= CG);                        // raw tile of group cig+2 -> registers
        f32x2 d[4][4];
        // 16 slots per group.  A slot's U fragment is ONE 16-byte vector: the four positions (xi, nu = 0..3) of one 16-channel
        // half h and one k-step j, feeding four MFMAs on four different accumulators; the slot order within xi is
        // (h0, j0) (h1, j0) (h0, j1) (h1, j1), so an accumulator comes back EIGHT matrix instructions after its first use
        // (v_mfma_f32_16x16x4_f32 runs at 0.80 of its rate when it comes back after 1, 2 or 4 and at 0.98 from 8 on:
        // tools/probes/mfma_dep_probe.py).  NH = 1: eight fragments (xi, j) on the even slots, order (x, j0) (x+1, j0) (x, j1)
        // (x+1, j1); the odd slots only carry the staging work below.
        auto frag_vec = [](int sl) {                               // slab vector index [xi][h][j] of slot sl
            if (NH == 2) return (sl >> 2) * 4 + (sl & 1) * 2 + ((sl >> 1) & 1);
            const int k = sl >> 1, xi = (k >> 2) * 2 + (k & 1), j = (k >> 1) & 1;
            return xi * 4 + j;
        };
        constexpr int STEP = NH == 2 ? 1 : 2;                      // slots between two fragments
        f32x4 af[3];                                            // U fragments of three consecutive fragments (rotating)
        af[0] = *(const f32x4*)(ucur + a_off + frag_vec(0) * 256);
        af[1] = *(const f32x4*)(ucur + a_off + frag_vec(STEP) * 256);
        if (HOLD && cig > 0) {                                  // slot 15 of the previous group: (xi 3, h 1, j 1), V still in Vn
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
                acc[12 + nu][NH - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u_held[nu], Vn[3][nu][1], acc[12 + nu][NH - 1], 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const bool has_frag = (p % STEP) == 0;
            const int fi = p / STEP;                            // fragment number
            if (has_frag && p + 2 * STEP < 16) af[(fi + 2) % 3] = *(const f32x4*)(ucur + a_off + frag_vec(p + 2 * STEP) * 256);
            // the LDS-DMA pieces of the next group's U slab one at a time under slots 1, 3, 5, 7: four in a row at the
            // head of the group cost 3 % (tools/probes/wino_stamps.py: slot 0 took twice a middle slot's time)
            if ((p & 1) && (p >> 1) < K::U_ITERS) issue_u_piece(udst, p >> 1);
            if (p == 2 * K::U_ITERS - 1 && !ulast) w_g += w_step;
            // the next group's input transform as ONE block of 32 packed adds: v_mfma_f32_16x16x4_f32 shares the fp32 vector
            // lanes, so a vector add beside it is never hidden (tools/probes/mfma_valu_coissue.py: 32 cycles per MFMA alone,
            // 42 + 4 NV with NV adds behind each) -- the transform is issue time to minimise, not work to spread under MFMAs
            if (p < 4) {
                load_patch_row(rnext, p, d[p]);
            } else if (p == 8) {
                transform(d, Vn);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);             // ... and not interleaved with this slot's MFMAs either
#endif
            } else if (p == 13) {
                // raw tile of group cig+2: its LDS buffer held group cig, whose transform finished before the last barrier
                write_raw(raw_lds + cur * K::RAW, st_in);
            }
            if (HOLD && p == 15) {
                u_held = af[fi % 3];                            // (issued behind the barrier, see above)
            } else if (has_frag) {
                const f32x4 u = af[fi % 3];
                int xi, h, j;
                if (NH == 2) { xi = p >> 2; h = p & 1; j = (p >> 1) & 1; }
                else { const int k = p >> 1; xi = (k >> 2) * 2 + (k & 1); h = 0; j = (k >> 1) & 1; }
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
                    acc[4 * xi + nu][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[nu], V[xi][nu][j], acc[4 * xi + nu][h], 0, 0, 0);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);                 // keep every slice under its own four MFMAs
#endif
            if constexpr (STAMP) {                             // detail (flags bit11): after every slot of groups 4 and 5
                if ((a.flags & 0x800) && (cig == 4 || cig == 5)) {
                    unsigned long long tq;
                    WINO_STAMP(tq);
                    if (tid == 0) a.dbg[(size_t)blockIdx.x * WN_STAMP_WORDS + 32 + (cig - 4) * 16 + p] = tq;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next group's U slab (LDS-DMA) has landed
        __syncthreads();
        if constexpr (STAMP) {                                 // end of group cig, words 8.. of the workgroup's record
            unsigned long long tg;
            WINO_STAMP(tg);
            if (tid == 0 && a.dbg && cig < 24) a.dbg[(size_t)blockIdx.x * WN_STAMP_WORDS + 8 + cig] = tg;
        }
    };
    if constexpr (STAMP) WINO_STAMP(ts2);
    int cig = 0;
    for (; cig + 1 < CG; cig += 2) {
        group_step(cig, Va, Vb);
        group_step(cig + 1, Vb, Va);
    }
    if (cig < CG) group_step(cig, Va, Vb);
    if (HOLD) {                                                 // slot 15 of the last group (its V: Va after an odd group count)
        const f32x2 (&Vl)[4][4] = (CG & 1) ? Va : Vb;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
            acc[12 + nu][NH - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u_held[nu], Vl[3][nu][1], acc[12 + nu][NH - 1], 0, 0, 0);
    }
    if constexpr (STAMP) WINO_STAMP(ts3);

    // ---- output transform  Y = A^T M A, bias, epilogue; lane: tile (wv, tn), channels 32*split + 16*h + 4*q + r.
    // Straight-line: the four pixels of a lane go out through a buffer descriptor over the two 8-channel planes of this
    // 16-channel half -- pixels outside the image and padding channel groups get an offset past its range and are dropped
    // by the hardware, the flag-dependent parts are three wave-uniform branches per half instead of branches per store
    // (the epilogue runs beside the partner workgroup's matrix loop and every instruction of it is issue time taken there).
    const float* bias = a.wpk + (size_t)a.CGin * w_step;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
    (void)relu; (void)add_res; (void)mask;                     // (read by the device-only block below)
    if constexpr (SHUF) {
        // PixelShuffle(2) folded into the store (flags bit3, as scipnp_conv3x3_c8_ex): conv channel 4c + 2dy + dx -> channel
        // c of pixel (2y + dy, 2x + dx); the 32 conv channels of this workgroup are the 8 channels of output group `split`, and a
        // lane's four values are the 2x2 sub-pixels of ONE output channel c = 4h + q.  The shuffled 16 x 64-pixel tile is
        // assembled in LDS (the staging buffers are free after the loop's last barrier) and leaves in whole 128-byte lines:
        // one thread = four consecutive pixels x 8 channels, residual (same layout) added there.  out = relu?(conv + bias + res).
        static_assert(NW == 4 && NH == 2, "shuffle epilogue: 8 x 32-pixel tiles, 32 conv channels");
        constexpr int ROW = 64 * 8 + 16 * 4;                   // floats per shuffled row: 4 floats of padding per 4 pixels
        float* const tile = smem_w;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const f32x4 bs = *(const f32x4*)(bias + (split * 4 + h * 2 + (q >> 1)) * 8 + 4 * (q & 1));
            f32x4 tm[4][2];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                tm[xi][0] = (acc[xi * 4 + 0][h] + acc[xi * 4 + 1][h]) + acc[xi * 4 + 2][h];
                tm[xi][1] = (acc[xi * 4 + 1][h] - acc[xi * 4 + 2][h]) - acc[xi * 4 + 3][h];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 v = ((i == 0) ? (tm[0][j] + tm[1][j]) + tm[2][j] : (tm[1][j] - tm[2][j]) - tm[3][j]) + bs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int yl = 2 * (2 * wv + i) + (e >> 1), xl = 2 * (2 * tn + j) + (e & 1);
                        tile[yl * ROW + xl * 8 + (xl >> 2) * 4 + 4 * h + q] = v[e];
                    }
                }
        }
        __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
        const int H2 = 2 * H, W2 = 2 * W;
        const int yl = tid >> 4, xg = tid & 15;                // 256 threads: 16 rows x 16 groups of 4 pixels
        const int y2 = 2 * y0 + yl, x2 = 2 * x0 + 4 * xg;
        const size_t plane = ((size_t)n * a.NCB + split) * (size_t)H2 * W2 * 8;           // floats; NCB = Cout/32 output groups
        const unsigned pbytes = (unsigned)((size_t)H2 * W2 * 32);
        auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + plane), 0, pbytes, 0x00020000);
        const float* src = tile + yl * ROW + xg * 36;
        f32x4 px[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) px[k] = *(const f32x4*)(src + 4 * k);
        unsigned off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) off[k] = (y2 < H2 && x2 + k < W2) ? (unsigned)(((size_t)y2 * W2 + x2 + k) * 32) : 0x80000000u;
        if (add_res) {
            auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual + plane), 0, pbytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                px[k] = px[k] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off[k >> 1] + 16 * (k & 1), 0, 0));
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) px[k][e] = fmaxf(px[k][e], 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, px[k]), r_out, off[k >> 1] + 16 * (k & 1), 0, 0);
#endif
        return;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int cog0 = split * 4 + h * 2;                    // this lane's group: cog0 + (q >> 1)
        if (cog0 >= a.CGout) continue;                         // wave-uniform
        const bool lane_ok = cog0 + (q >> 1) < a.CGout;
        const f32x4 bs = *(const f32x4*)(bias + (cog0 + (q >> 1)) * 8 + 4 * (q & 1));          // bias holds CoutP entries
        f32x4 tm[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            tm[xi][0] = (acc[xi * 4 + 0][h] + acc[xi * 4 + 1][h]) + acc[xi * 4 + 2][h];
            tm[xi][1] = (acc[xi * 4 + 1][h] - acc[xi * 4 + 2][h]) - acc[xi * 4 + 3][h];
        }
        f32x4 v[2][2];
        unsigned off[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = y0 + 2 * wv + i, x = x0 + 2 * tn + j;
                v[i][j] = ((i == 0) ? (tm[0][j] + tm[1][j]) + tm[2][j] : (tm[1][j] - tm[2][j]) - tm[3][j]) + bs;
                off[i][j] = (lane_ok && y < H && x < W)
                                ? (unsigned)((y * W + x) * 32 + 16 * (q & 1)) + (unsigned)(q >> 1) * plane_bytes
                                : 0x80000000u;
            }
#if defined(__HIP_DEVICE_COMPILE__)
        const size_t half0 = ((size_t)n * a.CGout + cog0) * HW * 8;                            // floats
        if (add_res) {
            auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    v[i][j] = v[i][j] + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off[i][j], 0, 0));
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][j][e] = fmaxf(v[i][j][e], 0.f);
        }
        if (mask) {   // ReLU backward: pass the gradient where the forward activation was > 0
            auto r_m = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask_src + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 fw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_m, off[i][j], 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][j][e] = (fw[e] > 0.f) ? v[i][j][e] : 0.f;
                }
        }
        auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i][j]), r_out, off[i][j], 0, 0);
#endif
    }
    if constexpr (STAMP) {
        WINO_STAMP(ts4);                                       // output transform done, stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WINO_STAMP(ts5);                                       // stores acknowledged
#if defined(__HIP_DEVICE_COMPILE__)
        if (tid == 0 && a.dbg) {
            unsigned long long* d = a.dbg + (size_t)blockIdx.x * WN_STAMP_WORDS;
            d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3; d[4] = ts4; d[5] = ts5;
            d[6] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |      // XCC_ID
                   (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));                             // HW_ID
            d[7] = __builtin_amdgcn_s_memrealtime();
            d[31] = rt0;
        }
#endif
    }
}

// U = G g G^T from the fp32 direct packing [cig][tap][CoutP][8]; one thread per (cig, co-block, p, co32, ci8) element
__global__ void pack_wino_kernel(const float* __restrict__ pk, float* __restrict__ out, int CGin, int CoutP) {
    const size_t NCB = CoutP / 32;
    const size_t total = (size_t)CGin * NCB * WN_SLAB;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        // slab element index: [xi 4][h 2][j 2][lane 64 = q*16 + tn][nu 4]  ->  U_p[co = 32 cb + 16 h + tn][ci = 2 q + j],
        // p = 4 xi + nu: one 16-byte vector per lane = the four positions of (xi, half h, k-step j)
        const int e = (int)(i % WN_SLAB);
        const size_t sl = i / WN_SLAB;
        const int cb = (int)(sl % NCB), cig = (int)(sl / NCB);
        const int nu = e & 3, tnl = (e >> 2) & 15, ql = (e >> 6) & 3, j = (e >> 8) & 1, h = (e >> 9) & 1, xi = e >> 10;
        const int co = cb * 32 + h * 16 + tnl, ci = 2 * ql + j;
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        double u = 0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx)
                u += G[xi][ky] * G[nu][kx] * (double)pk[(((size_t)cig * 9 + ky * 3 + kx) * CoutP + co) * 8 + ci];
        out[i] = (float)u;
    }
    if (i < (size_t)CoutP) out[total + i] = pk[(size_t)CGin * 9 * CoutP * 8 + i];       // bias
}

static inline int round_up_w(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace scipnp

using namespace scipnp;

extern "C" {

#ifndef SCIPNP_DIAG_BUILD   /* ---- product entries (libscipnp.so) */

size_t scipnp_conv3x3_wino_packed_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return 0;
    const int CoutP = round_up_w(Cout, 32);
    return (size_t)(Cin / 8) * (CoutP / 32) * WN_SLAB + CoutP;
}

int scipnp_pack_conv3x3_wino(const float* packed_f32, float* packed_wino, int Cin, int Cout, scipnp_stream_t s) {
    SCIPNP_REQUIRE(packed_f32 && packed_wino, "null pointer");
    SCIPNP_REQUIRE(Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0, "bad channel counts Cin=%d Cout=%d", Cin, Cout);
    SCIPNP_ALIGNED(packed_f32); SCIPNP_ALIGNED(packed_wino);
    const int CoutP = round_up_w(Cout, 32);
    const size_t total = (size_t)(Cin / 8) * (CoutP / 32) * WN_SLAB;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, packed_f32,
                       packed_wino, Cin / 8, CoutP);
    return launch_status("pack_wino_kernel");
}

int scipnp_conv3x3_c8w(const float* in, const float* packed_wino, float* out, const float* residual, const float* mask_src,
                       int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_wino && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0,
                   "bad shape n=%d Cin=%d Cout=%d h=%d w=%d (channels must be multiples of 8)", n, Cin, Cout, h, w);
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed_wino); SCIPNP_ALIGNED(out);
    if (residual) SCIPNP_ALIGNED(residual);
    if (mask_src) SCIPNP_ALIGNED(mask_src);
    SCIPNP_REQUIRE(!(flags & 4), "the Winograd kernel is stride 1");
    SCIPNP_REQUIRE(!(flags & 8) || (Cout % 32 == 0 && !(flags & (16 | 0x200))),
                   "PixelShuffle store: Cout must be a multiple of 32, no ReLU-mask epilogue, 8-row workgroups");
    SCIPNP_REQUIRE(!(flags & 8) || (long long)h * w * 4 * 32 < (1ll << 30), "shuffled plane too large for 32-bit buffer offsets");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    WinoArgs a;
    a.in = in; a.wpk = packed_wino; a.out = out; a.residual = residual; a.mask_src = mask_src;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w(Cout, 32) / 32;
    a.H = h; a.W = w; a.dbg = nullptr;
    const bool big = (flags & 0x200) != 0;                      // 16-row workgroups of 8 waves (default: 8 rows, 4 waves)
    const int th = big ? WinoCfg<8>::TH : WinoCfg<4>::TH;
    a.ntx = (w + 31) / 32; a.nty = (h + th - 1) / th;
    a.flags = flags;
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    const int tag = (flags & 0x100) ? 1 : 0;
    const int vi = tag * 2 + (big ? 1 : 0);
    const void* fns[4] = {(const void*)conv3x3_c8w_kernel<0, 4>, (const void*)conv3x3_c8w_kernel<0, 8>,
                          (const void*)conv3x3_c8w_kernel<1, 4>, (const void*)conv3x3_c8w_kernel<1, 8>};
    const size_t lds = big ? WinoCfg<8>::LDS_BYTES : WinoCfg<4>::LDS_BYTES;
    static LdsAttrOnce attr[4];
    if (int rc = attr[vi].ensure(fns[vi], lds, "conv3x3_c8w")) return rc;
    const dim3 grid((unsigned)total), block(big ? 512 : 256);
    if (flags & 8) {                                            // PixelShuffle(2) store (UpBlocks of FastDVDnet / DDnet)
        static LdsAttrOnce shuf_attr;
        if (int rc = shuf_attr.ensure((const void*)conv3x3_c8w_kernel<0, 4, 2, false, true>, lds, "conv3x3_c8w shuffle")) return rc;
        hipLaunchKernelGGL((conv3x3_c8w_kernel<0, 4, 2, false, true>), grid, block, lds, (hipStream_t)s, a);
        return launch_status("conv3x3_c8w_kernel<shuffle>");
    }
    if (!big && Cout <= 16) {                                   // one real 16-channel half: TAG 1 (first / last layers) only
        static LdsAttrOnce half_attr;
        if (int rc = half_attr.ensure((const void*)conv3x3_c8w_kernel<1, 4, 1>, lds, "conv3x3_c8w half")) return rc;
        hipLaunchKernelGGL((conv3x3_c8w_kernel<1, 4, 1>), grid, block, lds, (hipStream_t)s, a);
        return launch_status("conv3x3_c8w_kernel<1,4,1>");
    }
    switch (vi) {
        case 0: hipLaunchKernelGGL((conv3x3_c8w_kernel<0, 4>), grid, block, lds, (hipStream_t)s, a); break;
        case 1: hipLaunchKernelGGL((conv3x3_c8w_kernel<0, 8>), grid, block, lds, (hipStream_t)s, a); break;
        case 2: hipLaunchKernelGGL((conv3x3_c8w_kernel<1, 4>), grid, block, lds, (hipStream_t)s, a); break;
        default: hipLaunchKernelGGL((conv3x3_c8w_kernel<1, 8>), grid, block, lds, (hipStream_t)s, a); break;
    }
    return launch_status("conv3x3_c8w_kernel");
}

int scipnp_ffdnet_forward_c8w(const float* in_c8, float* out_c8, const float* const* packed_wino, int nb, int nc,
                              float* scratch0, float* scratch1, int B, int M, int N, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in_c8 && out_c8 && packed_wino && scratch0 && scratch1, "null pointer");
    SCIPNP_REQUIRE(nb >= 2 && nc % 8 == 0 && nc > 0, "bad network shape nb=%d nc=%d", nb, nc);
    float* buf[2] = {scratch0, scratch1};
    int rc = scipnp_conv3x3_c8w(in_c8, packed_wino[0], buf[0], nullptr, nullptr, B, 16, nc, M, N, 1 | 0x100, s);
    if (rc) return rc;
    int cur = 0;
    for (int l = 1; l < nb - 1; ++l) {
        rc = scipnp_conv3x3_c8w(buf[cur], packed_wino[l], buf[cur ^ 1], nullptr, nullptr, B, nc, nc, M, N, 1, s);
        if (rc) return rc;
        cur ^= 1;
    }
    return scipnp_conv3x3_c8w(buf[cur], packed_wino[nb - 1], out_c8, nullptr, nullptr, B, nc, 16, M, N, 0x100, s);
}

#else   /* ---- SCIPNP_DIAG_BUILD: the stamped instantiation lives in libscipnp_diag.so only (include/scipnp_diag.h) */

int scipnp_conv3x3_c8w_stamped(const float* in, const float* packed_wino, float* out, int n, int Cin, int Cout, int h, int w,
                               int flags, unsigned long long* stamps, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_wino && out && stamps, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && Cin > 0 && Cout > 16 && Cin % 8 == 0 && Cout % 8 == 0, "bad shape");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    WinoArgs a;
    a.in = in; a.wpk = packed_wino; a.out = out; a.residual = nullptr; a.mask_src = nullptr; a.dbg = stamps;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.NCB = round_up_w(Cout, 32) / 32;
    a.H = h; a.W = w;
    a.ntx = (w + 31) / 32; a.nty = (h + WinoCfg<4>::TH - 1) / WinoCfg<4>::TH;
    a.flags = flags & (1 | 0x800);
    const long long total = (long long)a.ntx * a.nty * n * a.NCB;
    SCIPNP_REQUIRE(total < (1ll << 31), "grid too large");
    const size_t lds = WinoCfg<4>::LDS_BYTES;
    static LdsAttrOnce stamp_attr;
    if (int rc = stamp_attr.ensure((const void*)conv3x3_c8w_kernel<0, 4, 2, true>, lds, "conv3x3_c8w stamped")) return rc;
    hipLaunchKernelGGL((conv3x3_c8w_kernel<0, 4, 2, true>), dim3((unsigned)total), dim3(256), lds, (hipStream_t)s, a);
    return launch_status("conv3x3_c8w_kernel<stamped>");
}

#endif  /* SCIPNP_DIAG_BUILD */

}  // extern "C"
