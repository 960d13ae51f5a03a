// 3x3 / stride-1 / pad-1 convolution in fp32 Winograd F(2x2,3x3) arithmetic -- PERSISTENT form with the input transform
// SHARED by all output channels of a tile (round 3).  Same algebra, same products and the same summation order as
// conv_wino.hip (results are bit-identical to it); what changes is who does what:
//
//   conv_wino.hip : wave = 16 tiles x 32 output channels x 16 Winograd positions (128 accumulator registers -> two waves
//                   per SIMD); each of the Cout/32 workgroups of a tile redoes the whole input transform (B^T d B) in
//                   registers; one workgroup per unit -- prologue and epilogue run beside one partner workgroup only.
//   this file     : wave = 16 tiles x 16 output channels x 16 positions (64 accumulator registers -> THREE waves per SIMD);
//                   workgroup = 12 waves = 2 tile rows x 96 output channels; V = B^T d B is computed ONCE per tile and channel
//                   group -- as four "jobs" per step (tile row x position half, 16 packed adds each), one per SIMD, rotating over
//                   the three waves of that SIMD -- written to LDS and read from there by the six channel-slice waves of the tile
//                   row: 3x fewer vector-ALU instructions on the lanes the fp32 MFMA itself runs on (tools/probes/
//                   mfma_valu_coissue.py: v_mfma_f32_16x16x4_f32 hides none of them).  The workgroups are PERSISTENT (one per
//                   CU, a static list of units each) and the channel-group pipeline runs on ACROSS unit boundaries: the raw
//                   tile / U slab of the next unit's first groups are in flight while the current unit finishes, so there is
//                   no per-unit prologue; the output transform is per lane (no cross-wave reduction) and overlaps the other
//                   waves' MFMAs.
//
// GEMM per Winograd position p (as conv_wino.hip):  M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile],
//   A = U_p 16 (co) x 4 (ci): lane l holds A[l & 15][l >> 4];  B = V_p 4 (ci) x 16 (tile): lane l holds B[l >> 4][l & 15];
//   D 16 x 16: lane l holds D[4 (l >> 4) + r][l & 15].  Lane (tn = l & 15, q = l >> 4) owns tile tn and, per 8-channel group,
//   the channel pair (2q, 2q+1): k-step j multiplies channel 2q + j.
//
// LDS per workgroup (floats):  raw[2][4 pair planes][6 x 34 halo pixels][2]   (global -> registers -> ds_write_b64)
//                              V[2][2 tile rows][4 xi][2 k-steps][64 lanes][4 nu]   (jobs -> ds_write_b128)
//                              U[2][6 slices][4 xi][2 k-steps][64 lanes][4 nu]      (LDS-DMA, 48 KiB per step)
// The four positions (xi, nu = 0..3) of a lane for ONE k-step are one 16-byte vector in both V and U: 16 ds_read_b128 feed
// the 32 MFMAs of a wave and step, and the MFMA order (xi, j0) (xi+1, j0) (xi, j1) (xi+1, j1) puts the second use of every
// accumulator EIGHT matrix instructions behind the first -- v_mfma_f32_16x16x4_f32 runs at 0.80 of its rate when an
// accumulator comes back after 1, 2 or 4 instructions and at 0.98 from 8 on, whatever the number of waves per SIMD
// (tools/probes/mfma_dep_probe.py, profiles/r03_mfma_dep.txt).
#include "common.hpp"
#include "../scipnp_lab.h"

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

// NHS = 16-channel slices per workgroup (6: 96 output channels), TR = tile rows per workgroup; waves = NHS * TR
template <int NHS, int TR>
struct WinoPCfg {
    static constexpr int WAVES = NHS * TR, THREADS = 64 * WAVES;
    static constexpr int TW = 32, TH = 2 * TR, TWP = TW + 2, THP = TH + 2;
    static constexpr int PLANE = (THP * TWP * 2 + 63) / 64 * 64;      // floats per channel-pair plane
    static constexpr int RAW = 4 * PLANE;
    static constexpr int UNITS = THP * TWP * 2;                        // 16-byte half pixels of the halo tile
    static constexpr int VBUF = TR * 8 * 64 * 4;                       // floats
    static constexpr int SLAB = NHS * 8 * 64 * 4;                      // floats: U of one channel group (16 pos x 16 NHS co x 8 ci)
    static constexpr int U_ITERS = (SLAB / 4) / THREADS;               // LDS-DMA pieces of 16 B per lane
    static constexpr int JOBS = 2 * TR;                                // (tile row, position half) per step
    static constexpr size_t LDS_BYTES = (2 * (size_t)RAW + 2 * (size_t)VBUF + 2 * (size_t)SLAB) * sizeof(float);
    static_assert(UNITS <= THREADS, "one staging unit per thread");
    static_assert((SLAB / 4) % THREADS == 0, "U slab must split evenly over the threads");
    static_assert(JOBS == 4 && WAVES % 4 == 0, "one transform job per SIMD and step");
};

struct WinoPArgs {
    const float* in;
    const float* wpk;        // [CGin][SLAB] + bias[CoutP]
    float* out;
    const float* residual;
    const float* mask_src;
    int CGin, CGout;
    int H, W;
    int ntx, nty, nunits;    // units = frames x nty x ntx (ntx fastest)
    int flags;
};

// DIAG: timing-only ablations selected by a.flags bits 12.. (wrong results; tools/probes/winop_ablate.py) -- the product
// instantiation compiles none of it
template <int TAG, int NHS, int TR, bool DIAG = false>
__global__ void __launch_bounds__((WinoPCfg<NHS, TR>::THREADS))
conv3x3_c8p_kernel(const WinoPArgs a) {
    const int dg = DIAG ? (a.flags >> 12) : 0;
    const bool d_nodma = dg & 1, d_nojob = dg & 2, d_noraw = dg & 4, d_noepi = dg & 8, d_nolds = dg & 16, d_nobar = dg & 32;
    using K = WinoPCfg<NHS, TR>;
    extern __shared__ __attribute__((aligned(16))) float smem_p[];
    float* const raw_lds = smem_p;                              // [2][RAW]
    float* const v_lds = smem_p + 2 * K::RAW;                   // [2][VBUF]
    float* const u_lds = v_lds + 2 * K::VBUF;                   // [2][SLAB]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = lane & 15, q = lane >> 4;
    const int wr = wv / NHS, wh = wv - wr * NHS;                // this wave's tile row and 16-channel slice
    const int H = a.H, W = a.W, CG = a.CGin;
    const size_t HW = (size_t)H * W;
    const unsigned plane_bytes = (unsigned)(HW * 32);
    (void)plane_bytes;

    // static unit list: unit u = blockIdx.x + k * gridDim.x.  With ntx a multiple of 8 (or a divisor) a workgroup keeps its
    // column block, and its vertical neighbours -- the workgroups 8, 16, ... further on, dealt to the SAME XCD -- run at the
    // same time: the halo rows they share hit in that XCD's L2.
    const int G = gridDim.x;
    const int my_units = (a.nunits - (int)blockIdx.x + G - 1) / G;
    const int S = my_units * CG;                                // steps of this workgroup

    // ---- staging plan of the raw tile (thread-constant part): unit e -> halo pixel (r, c), 16-byte half hf
    const int e = tid % K::UNITS;
    const int st_pix = e >> 1, st_hf = e & 1;
    const int st_r = st_pix / K::TWP, st_c = st_pix - st_r * K::TWP;
    const int lds_off = (2 * st_hf) * K::PLANE + st_pix * 2;

    // prefetch cursor of the raw tile: the step whose tile is fetched next
    int pf_unit = blockIdx.x, pf_cig = 0;
    unsigned pf_off = 0;
    const float* pf_base = a.in;
    auto pf_set_unit = [&]() {
        int t = pf_unit;
        const int bx = t % a.ntx;
        t /= a.ntx;
        const int by = t % a.nty, n = t / a.nty;
        const int gy = by * K::TH - 1 + st_r, gx = bx * K::TW - 1 + st_c;
        pf_off = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)((gy * W + gx) * 32 + 16 * st_hf) : 0xFFFFFF00u;
        pf_base = a.in + (size_t)n * CG * HW * 8;
    };
    pf_set_unit();
    f32x4 st_in = {0.f, 0.f, 0.f, 0.f};
    auto issue_raw = [&]() {                                    // fetch the cursor's tile into st_in, advance the cursor
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_in = __builtin_amdgcn_make_buffer_rsrc((void*)(pf_base + (size_t)pf_cig * HW * 8), 0, plane_bytes, 0x00020000);
        st_in = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_in, pf_off, 0, 0));
#endif
        if (++pf_cig == CG) {
            pf_cig = 0;
            pf_unit += G;
            if (pf_unit < a.nunits) pf_set_unit();              // (past the last unit the cursor is never used again)
        }
    };
    auto write_raw = [&](float* dst) {
        *(f32x2*)(dst + lds_off) = f32x2{st_in[0], st_in[1]};
        *(f32x2*)(dst + lds_off + K::PLANE) = f32x2{st_in[2], st_in[3]};
    };
    // U slab of channel group `cig` -> LDS by LDS-DMA, piece k of U_ITERS
    auto issue_u_piece = [&](float* dst, int cig, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
        auto r_w = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)cig * K::SLAB), 0, K::SLAB * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            r_w, (__attribute__((address_space(3))) void*)((char*)dst + 16 * (wv * 64 + k * K::THREADS)), 16,
            (unsigned)(16 * (tid + k * K::THREADS)), 0, 0, 0);
#endif
    };

    // ---- transform job (tile row jr, position half xp): V[xi][nu] for xi in {2xp, 2xp+1} of this lane's tile and channel
    // pair, from the raw rows 0,1,2 (xp = 0) / 1,2,3 (xp = 1); written as four 16-byte vectors V[xi][0..3] of one k-step
    auto run_job = [&](const float* rawp, float* vdst, int jr, int xp) {
        const float* src = rawp + q * K::PLANE + ((2 * jr + xp) * K::TWP + 2 * tn) * 2;
        auto load_row = [&](int r, f32x2 (&d)[4]) {
            const f32x4 lo = *(const f32x4*)(src + r * K::TWP * 2), hi = *(const f32x4*)(src + r * K::TWP * 2 + 4);
            d[0] = __builtin_shufflevector(lo, lo, 0, 1); d[1] = __builtin_shufflevector(lo, lo, 2, 3);
            d[2] = __builtin_shufflevector(hi, hi, 0, 1); d[3] = __builtin_shufflevector(hi, hi, 2, 3);
        };
        float* dst = vdst + (jr * 8 + 4 * xp) * 256 + lane * 4;        // vector (xi, j) at ((jr * 4 + xi) * 2 + j) * 256
        auto finish = [&](int x, const f32x2 (&t)[4]) {
            f32x2 v0 = psub(t[0], t[2]), v1 = t[1] + t[2], v2 = psub(t[2], t[1]), v3 = psub(t[1], t[3]);
#if defined(__HIP_DEVICE_COMPILE__)
            // (keeps the four results in register PAIRS: without it the element-wise stores below make the compiler split
            // every packed add of the transform into two scalar ones)
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
#endif
            *(f32x4*)(dst + (2 * x) * 256) = f32x4{v0[0], v1[0], v2[0], v3[0]};
            *(f32x4*)(dst + (2 * x + 1) * 256) = f32x4{v0[1], v1[1], v2[1], v3[1]};
        };
        // one xi at a time (the rows are loaded as they are needed: 24 live registers instead of 48)
        f32x2 da[4], db[4], dc[4], t[4];
        if (xp == 0) {                                          // rows d0,d1,d2: xi 0 = d0 - d2, xi 1 = d1 + d2
            load_row(0, da); load_row(2, db);
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = psub(da[c], db[c]);
            load_row(1, dc);
            finish(0, t);
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = dc[c] + db[c];
            finish(1, t);
        } else {                                                // rows d1,d2,d3 (local 0,1,2): xi 2 = d2 - d1, xi 3 = d1 - d3
            load_row(0, da); load_row(1, db);
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = psub(db[c], da[c]);
            load_row(2, dc);
            finish(0, t);
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c] = psub(da[c], dc[c]);
            finish(1, t);
        }
    };
    // job of this wave at step s (a wave-uniform choice): job jb = wv & 3 runs on the wave (wv >> 2) == s % 3 of "its" SIMD
    // (a workgroup's waves are dealt to the SIMDs cyclically, so the waves wv, wv + 4, wv + 8 share one)
    const int job_id = wv & 3, job_slot = wv >> 2;
    constexpr int SLOTS = K::WAVES / 4;

    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue (once per workgroup): raw tiles of steps 0 and 1, U of step 0, then V of step 0
    issue_raw();
    write_raw(raw_lds);
#pragma unroll
    for (int k = 0; k < K::U_ITERS; ++k) issue_u_piece(u_lds, 0, k);
    if (S > 1) {
        issue_raw();
        write_raw(raw_lds + K::RAW);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the U slab (LDS-DMA) has landed
    __syncthreads();
    if (job_slot == SLOTS - 1) run_job(raw_lds, v_lds, job_id >> 1, job_id & 1);
    __syncthreads();

    const float* bias = a.wpk + (size_t)CG * K::SLAB;
    const bool relu = a.flags & 1, add_res = (a.flags & 2) && a.residual, mask = (a.flags & 16) && a.mask_src;
    (void)relu; (void)add_res; (void)mask; (void)bias;

    const int cog0 = wh * 2;                                    // this lane's output group: cog0 + (q >> 1)
    const bool wave_ok = cog0 < a.CGout, lane_ok = cog0 + (q >> 1) < a.CGout;
    (void)lane_ok;
    // (loaded once: a load inside the epilogue would be waited for right there, a trip to L2 per unit)
    const f32x4 bs = wave_ok ? *(const f32x4*)(bias + (cog0 + (q >> 1)) * 8 + 4 * (q & 1)) : f32x4{0.f, 0.f, 0.f, 0.f};

    const int v_off = wr * 8 * 256 + lane * 4;                  // + (xi * 2 + j) * 256
    const int u_off = wh * 8 * 256 + lane * 4;                  // + (xi * 2 + j) * 256
    // ---- output transform Y = A^T M A, bias, epilogue of unit `unit`; lane: tile (wr, tn), channels 16 wh + 4 q + r.
    // Straight-line: the four pixels of a lane go out through a buffer descriptor over the two 8-channel planes of this
    // 16-channel slice -- pixels outside the image and padding channel groups get an offset past its range
    auto half2 = [](const f32x4& x, int hh) {                    // channel pair hh of a float4 (constant shuffle indices)
        return hh == 0 ? __builtin_shufflevector(x, x, 0, 1) : __builtin_shufflevector(x, x, 2, 3);
    };
    auto epilogue = [&](int unit) {
        if (wave_ok && !d_noepi) {
            int t = unit;
            const int bx = t % a.ntx;
            t /= a.ntx;
            const int by = t % a.nty, n = t / a.nty;
            const int x0 = bx * K::TW, y0 = by * K::TH;
#if defined(__HIP_DEVICE_COMPILE__)
            const size_t half0 = ((size_t)n * a.CGout + cog0) * HW * 8;                        // floats
            auto r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + half0), 0, 2 * plane_bytes, 0x00020000);
            auto r_res = __builtin_amdgcn_make_buffer_rsrc((void*)((add_res ? a.residual : a.out) + half0), 0, 2 * plane_bytes, 0x00020000);
            auto r_m = __builtin_amdgcn_make_buffer_rsrc((void*)((mask ? a.mask_src : a.out) + half0), 0, 2 * plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {                        // one output column of the 2x2 tile at a time (register peak)
                // (on channel PAIRS: a subtraction of float2 values is one v_pk_add_f32 with a negated operand, of float4 values
                // four v_sub_f32 -- and every vector instruction here is issue time taken from the matrix pipe)
                f32x2 tm[4][2];
#pragma unroll
                for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x2 a0 = half2(acc[xi * 4 + 0], hh);
                        const f32x2 a1 = half2(acc[xi * 4 + 1], hh);
                        const f32x2 a2 = half2(acc[xi * 4 + 2], hh);
                        const f32x2 a3 = half2(acc[xi * 4 + 3], hh);
                        tm[xi][hh] = (j == 0) ? (a0 + a1) + a2 : psub(psub(a1, a2), a3);
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int y = y0 + 2 * wr + i, x = x0 + 2 * tn + j;
                    f32x2 vh[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        vh[hh] = ((i == 0) ? (tm[0][hh] + tm[1][hh]) + tm[2][hh] : psub(psub(tm[1][hh], tm[2][hh]), tm[3][hh])) +
                                 half2(bs, hh);
                    f32x4 v = __builtin_shufflevector(vh[0], vh[1], 0, 1, 2, 3);
                    const unsigned off = (lane_ok && y < H && x < W)
                                             ? (unsigned)((y * W + x) * 32 + 16 * (q & 1)) + (unsigned)(q >> 1) * plane_bytes
                                             : 0x80000000u;
                    if (add_res) v = v + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off, 0, 0));
                    if (relu) {
#pragma unroll
                        for (int el = 0; el < 4; ++el) v[el] = fmaxf(v[el], 0.f);
                    }
                    if (mask) {   // ReLU backward: pass the gradient where the forward activation was > 0
                        const f32x4 fw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_m, off, 0, 0));
#pragma unroll
                        for (int el = 0; el < 4; ++el) v[el] = (fw[el] > 0.f) ? v[el] : 0.f;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, off, 0, 0);
                }
            }
#endif
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // Fragment f = (xi, j) of a step in issue order (0,0) (1,0) (0,1) (1,1) (2,0) (3,0) (2,1) (3,1); its V and U vectors sit at
    // index xi * 2 + j of the wave's eight.  Four register slots (f & 3), three fragments of look-ahead.  The LAST fragment of a
    // step is only LOADED before the step's barrier; its four MFMAs are issued right behind the barrier, after the first loads of
    // the next step have been sent off: the matrix pipe has work while every wave of the CU waits for its first operands and
    // does its bookkeeping -- with nothing held back the pipes idled there for 7 % of the launch (tools/probes/winop_ablate.py)
    auto frag_at = [](int f) { return (((f >> 2) * 2 + (f & 1)) * 2 + ((f >> 1) & 1)); };
    auto frag_xi = [](int f) { return (f >> 2) * 2 + (f & 1); };
    f32x4 vf[4], uf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { vf[k] = f32x4{0.f, 0.f, 0.f, 0.f}; uf[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto mma_frag = [&](int f) {
        const f32x4 v = vf[f & 3], u = uf[f & 3];
        const int xi = frag_xi(f);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
            acc[4 * xi + nu] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[nu], v[nu], acc[4 * xi + nu], 0, 0, 0);
    };
    int cur_unit = blockIdx.x, cig = 0;
    bool unit_done = false;
    for (int s = 0; s < S; ++s) {
        const int cur = s & 1;
        const float* vcur = v_lds + cur * K::VBUF + v_off;
        const float* ucur = u_lds + cur * K::SLAB + u_off;
        if (!d_nolds) {
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                vf[f] = *(const f32x4*)(vcur + frag_at(f) * 256);
                uf[f] = *(const f32x4*)(ucur + frag_at(f) * 256);
            }
        }
        if (s > 0) {
            mma_frag(7);                                        // held back from the previous step (slot 3)
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (unit_done) {
                epilogue(cur_unit);
                cur_unit += G;
            }
        }
        // raw tile of step s + 2: fetched now, written behind this step's MFMAs into the buffer step s's tile has left (its
        // transform ran during step s - 1)
        const bool fetch = s + 2 < S && !d_noraw;
        if (fetch) issue_raw();
        const bool more = s + 1 < S;
        const int ncig = (cig + 1 == CG) ? 0 : cig + 1;
        const bool my_job = more && (job_slot == s % SLOTS) && !d_nojob;
#pragma unroll
        for (int f = 0; f < 7; ++f) {
            if (f + 3 < 8 && !d_nolds) {
                vf[(f + 3) & 3] = *(const f32x4*)(vcur + frag_at(f + 3) * 256);
                uf[(f + 3) & 3] = *(const f32x4*)(ucur + frag_at(f + 3) * 256);
            }
            // the next step's U slab, one LDS-DMA piece per fragment
            if (more && f < K::U_ITERS && !d_nodma) issue_u_piece(u_lds + (cur ^ 1) * K::SLAB, ncig, f);
            // the transform job EARLY in the step: its LDS round trips are covered by the other waves' MFMAs (late in the step
            // the partners have run out of work and the pipe idles while the job wave waits for its rows)
            if (f == 1 && my_job) {
                run_job(raw_lds + (cur ^ 1) * K::RAW, v_lds + (cur ^ 1) * K::VBUF, job_id >> 1, job_id & 1);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (f == 5 && fetch) {                              // (not last: the wait for the load would hold up the barrier)
                write_raw(raw_lds + cur * K::RAW);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            mma_frag(f);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        unit_done = false;
        if (++cig == CG) {
            cig = 0;
            unit_done = true;
        }
        // the next step's U slab (LDS-DMA) has landed, this wave's LDS writes (raw tile, V of a transform job) are done and
        // fragment 7 of this step sits in its registers before anyone passes the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (!d_nobar) __builtin_amdgcn_s_barrier();
    }
    mma_frag(7);
    epilogue(cur_unit);
}

// U = G g G^T from the fp32 direct packing [cig][tap][CoutP][8], in the slab layout of this file:
//   [cig][slice h = co / 16][xi][k-step j][lane = q * 16 + tn][nu]  ->  U_p[co = 16 h + tn][ci = 2 q + j], p = 4 xi + nu
__global__ void pack_winop_kernel(const float* __restrict__ pk, float* __restrict__ out, int CGin, int CoutP, int CoutS) {
    // CoutS = 16 * NHS channels of the slab (>= the real channel count, <= CoutP of the direct packing)
    const size_t slab = (size_t)CoutS * 16 * 8;
    const size_t total = (size_t)CGin * slab;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int el = (int)(i % slab), cig = (int)(i / slab);
        const int nu = el & 3, ln = (el >> 2) & 63, j = (el >> 8) & 1, xi = (el >> 9) & 3, h = el >> 11;
        const int co = 16 * h + (ln & 15), ci = 2 * (ln >> 4) + j;
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        double u = 0;
        if (co < CoutP)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    u += G[xi][ky] * G[nu][kx] * (double)pk[(((size_t)cig * 9 + ky * 3 + kx) * CoutP + co) * 8 + ci];
        out[i] = (float)u;
    }
    if (i < (size_t)CoutS) out[total + i] = (i < (size_t)CoutP) ? pk[(size_t)CGin * 9 * CoutP * 8 + i] : 0.f;       // bias
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

// Cout % 96 == 0 is what this form is built for (FFDNet body and head layers); 1 = supported
int scipnp_conv3x3_c8p_supported(int Cin, int Cout) { return Cin > 0 && Cin % 8 == 0 && Cout == 96; }

size_t scipnp_conv3x3_winop_packed_floats(int Cin, int Cout) {
    if (!scipnp_conv3x3_c8p_supported(Cin, Cout)) return 0;
    return (size_t)(Cin / 8) * WinoPCfg<6, 2>::SLAB + 96;
}

int scipnp_pack_conv3x3_winop(const float* packed_f32, float* packed_winop, int Cin, int Cout, scipnp_stream_t s) {
    SCIPNP_REQUIRE(packed_f32 && packed_winop, "null pointer");
    SCIPNP_REQUIRE(scipnp_conv3x3_c8p_supported(Cin, Cout), "persistent Winograd form: Cin %% 8 == 0 and Cout == 96 (got %d, %d)", Cin, Cout);
    SCIPNP_ALIGNED(packed_f32); SCIPNP_ALIGNED(packed_winop);
    const int CoutP = (Cout + 31) / 32 * 32;
    const size_t total = (size_t)(Cin / 8) * WinoPCfg<6, 2>::SLAB;
    hipLaunchKernelGGL(pack_winop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, packed_f32,
                       packed_winop, Cin / 8, CoutP, 96);
    return launch_status("pack_winop_kernel");
}

int scipnp_conv3x3_c8p(const float* in, const float* packed_winop, float* out, const float* residual, const float* mask_src,
                       int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_winop && out, "null pointer");
    SCIPNP_REQUIRE(n > 0 && h > 0 && w > 0 && scipnp_conv3x3_c8p_supported(Cin, Cout),
                   "bad shape n=%d Cin=%d Cout=%d h=%d w=%d (persistent Winograd form: Cin %% 8 == 0, Cout == 96)", n, Cin, Cout, h, w);
    SCIPNP_ALIGNED(in); SCIPNP_ALIGNED(packed_winop); SCIPNP_ALIGNED(out);
    if (residual) SCIPNP_ALIGNED(residual);
    if (mask_src) SCIPNP_ALIGNED(mask_src);
    SCIPNP_REQUIRE(!(flags & (4 | 8 | 0x200)), "the persistent Winograd kernel is stride 1, plain store");
    SCIPNP_REQUIRE(!(flags & 16) || mask_src, "flag bit4 needs mask_src");
    SCIPNP_REQUIRE(!(flags & 2) || residual, "flag bit1 needs residual");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large for 32-bit buffer offsets (h*w < 2^25)");
    using K = WinoPCfg<6, 2>;
    WinoPArgs a;
    a.in = in; a.wpk = packed_winop; a.out = out; a.residual = residual; a.mask_src = mask_src;
    a.CGin = Cin / 8; a.CGout = Cout / 8;
    a.H = h; a.W = w;
    a.ntx = (w + K::TW - 1) / K::TW; a.nty = (h + K::TH - 1) / K::TH;
    const long long units = (long long)a.ntx * a.nty * n;
    SCIPNP_REQUIRE(units < (1ll << 30), "too many units");
    a.nunits = (int)units;
    a.flags = flags;
    static std::atomic<int> cus{0};
    int ncu = cus.load(std::memory_order_relaxed);
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail(SCIPNP_EHIP, "hipGetDeviceProperties");
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cus.store(ncu, std::memory_order_relaxed);
    }
    const int grid = (int)(units < ncu ? units : ncu);          // one persistent workgroup per CU (141 KiB of LDS each)
    const int tag = (flags & 0x100) ? 1 : 0;
    if (tag) {
        static LdsAttrOnce attr1;
        if (int rc = attr1.ensure((const void*)conv3x3_c8p_kernel<1, 6, 2>, K::LDS_BYTES, "conv3x3_c8p")) return rc;
        hipLaunchKernelGGL((conv3x3_c8p_kernel<1, 6, 2>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, (hipStream_t)s, a);
    } else {
        static LdsAttrOnce attr0;
        if (int rc = attr0.ensure((const void*)conv3x3_c8p_kernel<0, 6, 2>, K::LDS_BYTES, "conv3x3_c8p")) return rc;
        hipLaunchKernelGGL((conv3x3_c8p_kernel<0, 6, 2>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, (hipStream_t)s, a);
    }
    return launch_status("conv3x3_c8p_kernel");
}

// timing-only ablations of the persistent kernel (diag bits: 1 no U LDS-DMA in the loop, 2 no transform jobs, 4 no raw-tile
// staging, 8 no output transform / stores, 16 no V / U fragment reads, 32 no barrier); results are wrong by construction
int scipnp_conv3x3_c8p_diag(const float* in, const float* packed_winop, float* out, int n, int Cin, int Cout, int h, int w,
                            int flags, int diag, scipnp_stream_t s) {
    SCIPNP_REQUIRE(in && packed_winop && out && n > 0 && h > 0 && w > 0 && scipnp_conv3x3_c8p_supported(Cin, Cout), "bad arguments");
    SCIPNP_REQUIRE((long long)h * w * 32 < (1ll << 30), "image too large");
    using K = WinoPCfg<6, 2>;
    WinoPArgs a;
    a.in = in; a.wpk = packed_winop; a.out = out; a.residual = nullptr; a.mask_src = nullptr;
    a.CGin = Cin / 8; a.CGout = Cout / 8; a.H = h; a.W = w;
    a.ntx = (w + K::TW - 1) / K::TW; a.nty = (h + K::TH - 1) / K::TH;
    a.nunits = a.ntx * a.nty * n;
    a.flags = (flags & 1) | ((diag & 63) << 12);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(SCIPNP_EHIP, "hipGetDeviceProperties");
    const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    static LdsAttrOnce attr;
    if (int rc = attr.ensure((const void*)conv3x3_c8p_kernel<0, 6, 2, true>, K::LDS_BYTES, "conv3x3_c8p diag")) return rc;
    hipLaunchKernelGGL((conv3x3_c8p_kernel<0, 6, 2, true>), dim3(a.nunits < ncu ? a.nunits : ncu), dim3(K::THREADS), K::LDS_BYTES,
                       (hipStream_t)s, a);
    return launch_status("conv3x3_c8p_kernel<diag>");
}

}  // extern "C"
