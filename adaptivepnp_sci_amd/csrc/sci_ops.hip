// SCI forward / transpose operators, Euclidean projections, Bayer layout conversions, dual
// updates and the SSE reduction.  All HBM-bound streaming kernels: one pass, 16-byte lanes where
// the layout allows, no intermediate tensors.  Built with -ffp-contract=off so that every
// multiply and add rounds separately, exactly like the reference's op-by-op PyTorch expressions.
#include "common.hpp"

namespace scipnp {

// ===================================================================== reference layout (M,N,B,4)
// One thread owns the four Bayer planes of one (quad, frame): a float4.  The B frames of a quad are
// B consecutive lanes, so for B = 8 a quad is one 128-byte line and Sigma_t is a wavefront shuffle.

template <int LOGB, bool CONTIG>
__device__ __forceinline__ float4 frame_sum_shuffle(float4 v, int lane) {
    // every lane of a quad's group receives the torch-order sum over the group's B frames
    // (CONTIG: the order of a contiguous reduced dim, else of a strided one -- common.hpp)
    constexpr int B = 1 << LOGB;
    const int base = lane & ~(B - 1);
    float4 r;
    float* rp = &r.x;
    const float* vp = &v.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float mine = vp[c];
        auto term = [&](int i) { return __shfl(mine, base + i, WAVE); };
        rp[c] = CONTIG ? torch_contig_sum<B>(B, term) : torch_strided_sum<B>(B, term);
    }
    return r;
}

template <int LOGB, int MODE>  // MODE 0 two-stage, 1 one-stage, 2 = A only, 3 = At only, 4 = Phi_sum
__global__ void __launch_bounds__(256)
ref_layout_kernel(const float4* __restrict__ theta, const float4* __restrict__ bb,
                  const float4* __restrict__ Phi, const float4* __restrict__ y,
                  const float4* __restrict__ Phisum, float4* xout, float4* yout,
                  long long nquad, float c0, float c1) {
    constexpr int B = 1 << LOGB;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // = quad*B + t
    const long long total = nquad * B;
    const bool live = gid < total;
    const long long idx = live ? gid : (total - 1);  // keep whole waves converged for the shuffles
    const long long quad = idx >> LOGB;
    const int lane = threadIdx.x & 63;
    const float4 ph = Phi[idx];
    if (MODE == 4) {
        float4 s = frame_sum_shuffle<LOGB, false>(ph, lane);
        s.x = (s.x == 0.f) ? 1.f : s.x;
        s.y = (s.y == 0.f) ? 1.f : s.y;
        s.z = (s.z == 0.f) ? 1.f : s.z;
        s.w = (s.w == 0.f) ? 1.f : s.w;
        if (live && (idx & (B - 1)) == 0) yout[quad] = s;
        return;
    }
    if (MODE == 3) {
        const float4 yy = y[quad];
        if (live) xout[idx] = make_float4(yy.x * ph.x, yy.y * ph.y, yy.z * ph.z, yy.w * ph.w);
        return;
    }
    const float4 th = theta[idx];
    if (MODE == 2) {
        float4 pr = make_float4(th.x * ph.x, th.y * ph.y, th.z * ph.z, th.w * ph.w);
        float4 s = frame_sum_shuffle<LOGB, true>(pr, lane);
        if (live && (idx & (B - 1)) == 0) yout[quad] = s;
        return;
    }
    const float4 bv = bb[idx];
    float4 p;
    if (MODE == 0) {
        p = make_float4(th.x - c0 * bv.x, th.y - c0 * bv.y, th.z - c0 * bv.z, th.w - c0 * bv.w);
    } else {
        p = make_float4(th.x + bv.x, th.y + bv.y, th.z + bv.z, th.w + bv.w);
    }
    float4 pr = make_float4(p.x * ph.x, p.y * ph.y, p.z * ph.z, p.w * ph.w);
    const float4 yb = frame_sum_shuffle<LOGB, true>(pr, lane);
    const float4 yy = y[quad];
    const float4 ps = Phisum[quad];
    float4 r;
    if (MODE == 0) {
        r = make_float4((yy.x - yb.x) / (c1 + ps.x), (yy.y - yb.y) / (c1 + ps.y),
                        (yy.z - yb.z) / (c1 + ps.z), (yy.w - yb.w) / (c1 + ps.w));
        r = make_float4(p.x + ph.x * r.x, p.y + ph.y * r.y, p.z + ph.z * r.z, p.w + ph.w * r.w);
    } else {
        r = make_float4((yy.x - yb.x) / (ps.x + c1), (yy.y - yb.y) / (ps.y + c1),
                        (yy.z - yb.z) / (ps.z + c1), (yy.w - yb.w) / (ps.w + c1));
        r = make_float4(p.x + c0 * (r.x * ph.x), p.y + c0 * (r.y * ph.y),
                        p.z + c0 * (r.z * ph.z), p.w + c0 * (r.w * ph.w));
    }
    if (live) xout[idx] = r;
}

// Any other frame count (B < 64): one thread per quad walks its B float4 entries (contiguous in this layout) with the
// run-time forms of the two torch summation orders; the projections read theta, b, Phi a second time for the output
// (cache-served).  Same expressions, operation by operation, as ref_layout_kernel.
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
template <int MODE>
__device__ __forceinline__ float4 ref_p(const float4* theta, const float4* bb, long long i, float c0) {
    const float4 th = theta[i];
    if (MODE == 2) return th;
    const float4 bv = bb[i];
    if (MODE == 0) return make_float4(th.x - c0 * bv.x, th.y - c0 * bv.y, th.z - c0 * bv.z, th.w - c0 * bv.w);
    return make_float4(th.x + bv.x, th.y + bv.y, th.z + bv.z, th.w + bv.w);
}

template <int MODE>
__global__ void __launch_bounds__(256)
ref_layout_anyB_kernel(const float4* __restrict__ theta, const float4* __restrict__ bb,
                       const float4* __restrict__ Phi, const float4* __restrict__ y,
                       const float4* __restrict__ Phisum, float4* xout, float4* yout,
                       long long nquad, int B, float c0, float c1) {
    const long long quad = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (quad >= nquad) return;
    const long long base = quad * B;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 4) {
        float4 s = torch_strided_sum_rt(B, zero, [&](int i) { return Phi[base + i]; });
        s.x = (s.x == 0.f) ? 1.f : s.x;
        s.y = (s.y == 0.f) ? 1.f : s.y;
        s.z = (s.z == 0.f) ? 1.f : s.z;
        s.w = (s.w == 0.f) ? 1.f : s.w;
        yout[quad] = s;
        return;
    }
    if (MODE == 3) {
        const float4 yy = y[quad];
        for (int t = 0; t < B; ++t) xout[base + t] = f4_mul(yy, Phi[base + t]);
        return;
    }
    const float4 yb = torch_contig_sum_rt(B, zero, [&](int i) { return f4_mul(ref_p<MODE>(theta, bb, base + i, c0), Phi[base + i]); });
    if (MODE == 2) {
        yout[quad] = yb;
        return;
    }
    const float4 yy = y[quad];
    const float4 ps = Phisum[quad];
    float4 r;
    if (MODE == 0)
        r = make_float4((yy.x - yb.x) / (c1 + ps.x), (yy.y - yb.y) / (c1 + ps.y), (yy.z - yb.z) / (c1 + ps.z),
                        (yy.w - yb.w) / (c1 + ps.w));
    else
        r = make_float4((yy.x - yb.x) / (ps.x + c1), (yy.y - yb.y) / (ps.y + c1), (yy.z - yb.z) / (ps.z + c1),
                        (yy.w - yb.w) / (ps.w + c1));
    for (int t = 0; t < B; ++t) {
        const float4 p = ref_p<MODE>(theta, bb, base + t, c0);
        const float4 ph = Phi[base + t];
        if (MODE == 0)
            xout[base + t] = make_float4(p.x + ph.x * r.x, p.y + ph.y * r.y, p.z + ph.z * r.z, p.w + ph.w * r.w);
        else
            xout[base + t] = make_float4(p.x + c0 * (r.x * ph.x), p.y + c0 * (r.y * ph.y), p.z + c0 * (r.z * ph.z),
                                         p.w + c0 * (r.w * ph.w));
    }
}

template <int MODE>
static int launch_ref_layout(const float* theta, const float* b, const float* Phi, const float* y,
                             const float* Phisum, float* xout, float* yout, int M, int N, int B,
                             float c0, float c1, hipStream_t st) {
    SCIPNP_REQUIRE(M > 0 && N > 0, "M,N must be positive (got %d,%d)", M, N);
    SCIPNP_REQUIRE(B >= 1 && B <= TORCH_SUM_RT_MAX, "B = %d frames: 1 <= B <= %d", B, TORCH_SUM_RT_MAX);
    const long long nquad = (long long)M * N;
    const long long total = nquad * B;
    const int threads = 256;
    if (!(B == 1 || B == 2 || B == 4 || B == 8 || B == 16)) {      // the shuffle kernel needs a power of two <= 16
        hipLaunchKernelGGL((ref_layout_anyB_kernel<MODE>), dim3((unsigned)((nquad + threads - 1) / threads)), dim3(threads), 0,
                           st, (const float4*)theta, (const float4*)b, (const float4*)Phi, (const float4*)y,
                           (const float4*)Phisum, (float4*)xout, (float4*)yout, nquad, B, c0, c1);
        return launch_status("ref_layout_anyB_kernel");
    }
    const unsigned blocks = (unsigned)((total + threads - 1) / threads);
#define SCIPNP_GO(LB)                                                                              \
    hipLaunchKernelGGL((ref_layout_kernel<LB, MODE>), dim3(blocks), dim3(threads), 0, st,           \
                       (const float4*)theta, (const float4*)b, (const float4*)Phi, (const float4*)y, \
                       (const float4*)Phisum, (float4*)xout, (float4*)yout, nquad, c0, c1)
    switch (B) {
        case 1: SCIPNP_GO(0); break;
        case 2: SCIPNP_GO(1); break;
        case 4: SCIPNP_GO(2); break;
        case 8: SCIPNP_GO(3); break;
        default: SCIPNP_GO(4); break;
    }
#undef SCIPNP_GO
    return launch_status("ref_layout_kernel");
}

// mosaic (H,W,B) <-> planes (M,N,B,4): thread = (quad, frame); the B frames of a mosaic pixel are
// contiguous, so each of the four plane values is a 4*B-byte coalesced run per quad.
template <bool SPLIT>
__global__ void __launch_bounds__(256)
bayer_reorder_kernel(const float* __restrict__ src, float* __restrict__ dst, int M, int N, int B) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)M * N * B;
    if (gid >= total) return;
    const int t = (int)(gid % B);
    const long long quad = gid / B;
    const int n = (int)(quad % N), m = (int)(quad / N);
    const long long W = 2LL * N;
    const long long r0 = ((2LL * m) * W + 2LL * n) * B + t;
    const long long r1 = r0 + W * B;
    float4* pl = (float4*)(SPLIT ? dst : const_cast<float*>(src));
    if (SPLIT) {
        pl[gid] = make_float4(src[r0], src[r0 + B], src[r1], src[r1 + B]);
    } else {
        const float4 v = pl[gid];
        dst[r0] = v.x; dst[r0 + B] = v.y; dst[r1] = v.z; dst[r1 + B] = v.w;
    }
}

// ===================================================================== plane-major conversions
// state[t][ib][m][n] <-> mosaic[(2m+dy)][(2n+dx)][t].  Tiled through LDS so that both sides move
// whole lines: a block handles one mosaic row pair (2 rows) x 32 quads x all B frames.
constexpr int CONV_TQ = 32;

constexpr int CONV_FC = 64;       // frames per block (LDS tile); more frames: blockIdx.z walks chunks of 64

template <bool TO_STATE>
__global__ void __launch_bounds__(256)
state_mosaic_kernel(const float* __restrict__ src, float* __restrict__ dst, int M, int N, int Bt) {
    extern __shared__ float tile[];  // [2 rows][CONV_TQ*2 px][B+1]
    const int m = blockIdx.y;
    const int n0 = blockIdx.x * CONV_TQ;
    const int nq = min(CONV_TQ, N - n0);
    const int t0 = blockIdx.z * CONV_FC, B = min(CONV_FC, Bt - t0);       // this block's frames t0 .. t0+B-1
    const int W = 2 * N;
    const int pitch = B + 1;
    const int per_row = 2 * nq * B;  // floats of one mosaic row segment (this chunk of frames)
    const size_t plane = (size_t)M * N;
    if (TO_STATE) {
        for (int i = threadIdx.x; i < 2 * per_row; i += blockDim.x) {
            const int dy = i / per_row, j = i % per_row;
            const int px = j / B, t = j % B;
            tile[(dy * 2 * CONV_TQ + px) * pitch + t] =
                src[((size_t)(2 * m + dy) * W + 2 * n0 + px) * Bt + t0 + t];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < B * 4 * nq; i += blockDim.x) {
            const int q = i % nq, ib = (i / nq) & 3, t = i / (4 * nq);
            const int dy = ib >> 1, dx = ib & 1;
            dst[((size_t)(t0 + t) * 4 + ib) * plane + (size_t)m * N + n0 + q] =
                tile[(dy * 2 * CONV_TQ + 2 * q + dx) * pitch + t];
        }
    } else {
        for (int i = threadIdx.x; i < B * 4 * nq; i += blockDim.x) {
            const int q = i % nq, ib = (i / nq) & 3, t = i / (4 * nq);
            const int dy = ib >> 1, dx = ib & 1;
            tile[(dy * 2 * CONV_TQ + 2 * q + dx) * pitch + t] =
                src[((size_t)(t0 + t) * 4 + ib) * plane + (size_t)m * N + n0 + q];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * per_row; i += blockDim.x) {
            const int dy = i / per_row, j = i % per_row;
            const int px = j / B, t = j % B;
            dst[((size_t)(2 * m + dy) * W + 2 * n0 + px) * Bt + t0 + t] = tile[(dy * 2 * CONV_TQ + px) * pitch + t];
        }
    }
}

__global__ void __launch_bounds__(256)
y_to_meas_kernel(const float* __restrict__ y, float* __restrict__ meas, int M, int N) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = 4LL * M * N;
    if (gid >= total) return;
    const int n = (int)(gid % N);
    const int m = (int)((gid / N) % M);
    const int ib = (int)(gid / ((long long)M * N));
    meas[gid] = y[(size_t)(2 * m + (ib >> 1)) * (2 * N) + 2 * n + (ib & 1)];
}

// planar rgb [B][3][H][W] <-> cube (H,W,3,B): block = one image row x 64 columns x up to 64 frames, through LDS
template <bool TO_CUBE>
__global__ void __launch_bounds__(256)
rgb_cube_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Bt) {
    extern __shared__ float tile[];  // [64 px][3*B + 1]
    const int r = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int nc = min(64, W - c0);
    const int t0 = blockIdx.z * CONV_FC, B = min(CONV_FC, Bt - t0);
    const int CB = 3 * B, pitch = CB + 1;
    const size_t HW = (size_t)H * W;
    if (TO_CUBE) {
        for (int i = threadIdx.x; i < CB * nc; i += blockDim.x) {
            const int px = i % nc, ch_t = i / nc;  // ch_t = t*3 + c in the planar source order
            const int t = ch_t / 3, c = ch_t % 3;
            tile[px * pitch + c * B + t] = src[((size_t)(t0 + t) * 3 + c) * HW + (size_t)r * W + c0 + px];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CB * nc; i += blockDim.x) {
            const int px = i / CB, ct = i % CB, c = ct / B, t = ct % B;
            dst[(((size_t)r * W + c0 + px) * 3 + c) * Bt + t0 + t] = tile[px * pitch + ct];
        }
    } else {
        for (int i = threadIdx.x; i < CB * nc; i += blockDim.x) {
            const int px = i / CB, ct = i % CB, c = ct / B, t = ct % B;
            tile[px * pitch + ct] = src[(((size_t)r * W + c0 + px) * 3 + c) * Bt + t0 + t];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CB * nc; i += blockDim.x) {
            const int px = i % nc, ch_t = i / nc;
            const int t = ch_t / 3, c = ch_t % 3;
            dst[((size_t)(t0 + t) * 3 + c) * HW + (size_t)r * W + c0 + px] = tile[px * pitch + c * B + t];
        }
    }
}

// ===================================================================== plane-major projection
// state[t][q], q in [0, Q = 4MN); meas[q].  One thread = VEC consecutive pixels, all B frames in
// registers (MAXB template) -> theta, b, Phi are each read exactly once, x written once:
// 16 B per (pixel,frame) + 8 B per pixel of measurement = the algorithmic minimum.
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

// the projection's streamed operands (theta, b, Phi: read once per launch).  NT: non-temporal loads -- on a state far larger than the 256 MB
// Infinity Cache (2048 x 2048 x 8: 570 MB per launch) the launch takes 94-96 us instead of 119 (6.0 against 4.8 TB/s,
// profiles/r06k_proj_stream.txt); on states the cache holds (<= 285 MB per launch, launches back to back) plain loads are 4-25 % faster,
// and non-temporal STORES of x lose at every size -- so the entry point asks for NT only above PROJ_NT_BYTES per launch.
constexpr double PROJ_NT_BYTES = 384e6;
template <typename V, bool NT>
__device__ __forceinline__ V proj_ld(const float* p) {
    if constexpr (NT && sizeof(V) == 16) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t v = __builtin_nontemporal_load((const f4_t*)p);
        return __builtin_bit_cast(V, v);
    } else {
        return *(const V*)p;
    }
}

template <int VEC, int MAXB, int MODE, bool NT = false>  // MODE 0 two-stage, 1 one-stage, 2 setup (Phi_sum, x0 = y*Phi)
__global__ void __launch_bounds__(256)
pm_project_kernel(const float* __restrict__ theta, const float* __restrict__ bb,
                  const float* __restrict__ Phi, const float* __restrict__ y,
                  const float* Phisum_in, float* Phisum_out, float* xout,
                  long long Q, int B, float c0, float c1) {
    using V = typename VecT<VEC>::type;
    const long long q = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (q >= Q) return;
    float p[MAXB][VEC], ph[MAXB][VEC];
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        if (t < B) {
            const size_t o = (size_t)t * Q + q;
            V phv = proj_ld<V, NT>(Phi + o);
            const float* php = (const float*)&phv;
            if (MODE == 2) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) { ph[t][v] = php[v]; p[t][v] = php[v]; }
            } else {
                V thv = proj_ld<V, NT>(theta + o);
                V bv = proj_ld<V, NT>(bb + o);
                const float* thp = (const float*)&thv;
                const float* bp = (const float*)&bv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    ph[t][v] = php[v];
                    p[t][v] = (MODE == 0) ? (thp[v] - c0 * bp[v]) : (thp[v] + bp[v]);
                }
            }
        }
    }
    V yv = *(const V*)(y + q);
    const float* yp = (const float*)&yv;
    float r[VEC];
    if (MODE == 2) {
        V so;
        float* sp = (float*)&so;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float s = torch_strided_sum<MAXB>(B, [&](int i) { return ph[i][v]; });
            s = (s == 0.f) ? 1.f : s;
            sp[v] = s;
            r[v] = yp[v];
        }
        *(V*)(Phisum_out + q) = so;
        if (xout == nullptr) return;
    } else {
        V sv = *(const V*)(Phisum_in + q);
        const float* sp = (const float*)&sv;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float yb = torch_contig_sum<MAXB>(B, [&](int i) { return p[i][v] * ph[i][v]; });
            r[v] = (MODE == 0) ? (yp[v] - yb) / (c1 + sp[v]) : (yp[v] - yb) / (sp[v] + c1);
        }
    }
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        if (t < B) {
            V ov;
            float* op = (float*)&ov;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (MODE == 0) op[v] = p[t][v] + ph[t][v] * r[v];
                else if (MODE == 1) op[v] = p[t][v] + c0 * (r[v] * ph[t][v]);
                else op[v] = r[v] * ph[t][v];
            }
            *(V*)(xout + (size_t)t * Q + q) = ov;
        }
    }
}

// 33..63 frames: the per-thread register arrays no longer fit; two passes over the frames (sum, then output), the second
// served from cache.  T = float4 (four consecutive pixels) or float.
template <typename T> __device__ __forceinline__ T pm_ld(const float* p, size_t o) { return *(const T*)(p + o); }
__device__ __forceinline__ float f4_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float4 f4_sub_scaled(float4 a, float c, float4 b) {
    return make_float4(a.x - c * b.x, a.y - c * b.y, a.z - c * b.z, a.w - c * b.w);
}
__device__ __forceinline__ float f4_sub_scaled(float a, float c, float b) { return a - c * b; }
template <typename T> __device__ __forceinline__ T f4_zero();
template <> __device__ __forceinline__ float f4_zero<float>() { return 0.f; }
template <> __device__ __forceinline__ float4 f4_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <typename T, int MODE>
__global__ void __launch_bounds__(256)
pm_project_anyB_kernel(const float* __restrict__ theta, const float* __restrict__ bb, const float* __restrict__ Phi,
                       const float* __restrict__ y, const float* Phisum_in, float* Phisum_out, float* xout, long long Q,
                       int B, float c0, float c1) {
    constexpr int VEC = sizeof(T) / sizeof(float);
    const long long q = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (q >= Q) return;
    auto p_at = [&](int t) -> T {
        const size_t o = (size_t)t * Q + q;
        if (MODE == 0) return f4_sub_scaled(pm_ld<T>(theta, o), c0, pm_ld<T>(bb, o));
        return f4_add(pm_ld<T>(theta, o), pm_ld<T>(bb, o));
    };
    T r;
    float* rp = (float*)&r;
    const T yv = pm_ld<T>(y, q);
    const float* yp = (const float*)&yv;
    if (MODE == 2) {
        T s = torch_strided_sum_rt(B, f4_zero<T>(), [&](int i) { return pm_ld<T>(Phi, (size_t)i * Q + q); });
        float* sp = (float*)&s;
#pragma unroll
        for (int v = 0; v < VEC; ++v) sp[v] = (sp[v] == 0.f) ? 1.f : sp[v];
        *(T*)(Phisum_out + q) = s;
        if (xout == nullptr) return;
        r = yv;
    } else {
        const T yb = torch_contig_sum_rt(B, f4_zero<T>(), [&](int i) { return f4_mul(p_at(i), pm_ld<T>(Phi, (size_t)i * Q + q)); });
        const float* ybp = (const float*)&yb;
        const T sv = pm_ld<T>(Phisum_in, q);
        const float* sp = (const float*)&sv;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            rp[v] = (MODE == 0) ? (yp[v] - ybp[v]) / (c1 + sp[v]) : (yp[v] - ybp[v]) / (sp[v] + c1);
    }
    for (int t = 0; t < B; ++t) {
        const size_t o = (size_t)t * Q + q;
        const T ph = pm_ld<T>(Phi, o);
        const float* php = (const float*)&ph;
        T ov;
        float* op = (float*)&ov;
        if (MODE == 2) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) op[v] = rp[v] * php[v];
        } else {
            const T p = p_at(t);
            const float* pp = (const float*)&p;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                op[v] = (MODE == 0) ? (pp[v] + php[v] * rp[v]) : (pp[v] + c0 * (rp[v] * php[v]));
        }
        *(T*)(xout + o) = ov;
    }
}

// units > 1: U independent problems of one shape in the UNIT-BATCHED layout, state [B][U][4][M][N], measurement [U][4][M][N] --
// the projection is per mosaic pixel with a sum over the pixel's own B frames, so the U units are simply 4 M N U pixels.
template <int MODE>
static int launch_pm_project(const float* theta, const float* b, const float* Phi, const float* y,
                             const float* Phisum_in, float* Phisum_out, float* x, int M, int N, int B,
                             float c0, float c1, hipStream_t st, int units = 1) {
    SCIPNP_REQUIRE(M > 0 && N > 0 && B > 0 && B <= TORCH_SUM_RT_MAX, "bad shape M=%d N=%d B=%d (B <= %d)", M, N, B,
                   TORCH_SUM_RT_MAX);
    SCIPNP_REQUIRE(units >= 1, "units must be >= 1 (got %d)", units);
    const long long Q = 4LL * M * N * units;
    const bool vec = (Q % 4 == 0) && aligned16(Phi) && aligned16(y) && (x == nullptr || aligned16(x)) &&
                     (MODE == 2 ? aligned16(Phisum_out) : (aligned16(theta) && aligned16(b) && aligned16(Phisum_in)));
    const int threads = 256;
    if (B > 32) {
        if (vec)
            hipLaunchKernelGGL((pm_project_anyB_kernel<float4, MODE>), dim3((unsigned)((Q / 4 + threads - 1) / threads)),
                               dim3(threads), 0, st, theta, b, Phi, y, Phisum_in, Phisum_out, x, Q, B, c0, c1);
        else
            hipLaunchKernelGGL((pm_project_anyB_kernel<float, MODE>), dim3((unsigned)((Q + threads - 1) / threads)),
                               dim3(threads), 0, st, theta, b, Phi, y, Phisum_in, Phisum_out, x, Q, B, c0, c1);
        return launch_status("pm_project_anyB_kernel");
    }
#define SCIPNP_GO(VEC, MAXB)                                                                        \
    hipLaunchKernelGGL((pm_project_kernel<VEC, MAXB, MODE>),                                         \
                       dim3((unsigned)((Q / VEC + threads - 1) / threads)), dim3(threads), 0, st,    \
                       theta, b, Phi, y, Phisum_in, Phisum_out, x, Q, B, c0, c1)
    // small states (a 256 x 256 mosaic is 64 workgroups of 4-pixel threads on 256 CUs): one pixel per thread -- the loads of
    // a wave stay contiguous, the launch gets four times the workgroups (ADMM-TV 256x256x8: 6.7 -> see profiles/r03d_*)
    const bool wide = Q / 4 >= 512LL * threads;
    // states beyond the Infinity Cache: theta, b, Phi are read once per launch and nothing of them is found again -- non-temporal loads
    const bool nt = vec && wide && MODE != 2 && 16.0 * (double)Q * B >= PROJ_NT_BYTES;
#define SCIPNP_GO_NT(VEC, MAXB)                                                                     \
    hipLaunchKernelGGL((pm_project_kernel<VEC, MAXB, MODE, true>),                                   \
                       dim3((unsigned)((Q / VEC + threads - 1) / threads)), dim3(threads), 0, st,    \
                       theta, b, Phi, y, Phisum_in, Phisum_out, x, Q, B, c0, c1)
    if (nt && B <= 8) SCIPNP_GO_NT(4, 8);
    else if (nt && B <= 16) SCIPNP_GO_NT(4, 16);
    else if (vec && wide && B <= 8) SCIPNP_GO(4, 8);
    else if (vec && wide && B <= 16) SCIPNP_GO(4, 16);
    else if (vec && wide) SCIPNP_GO(2, 32);      // 17..32 frames: 2 pixels per thread keep p and Phi in 128 registers
    else if (B <= 8) SCIPNP_GO(1, 8);
    else if (B <= 16) SCIPNP_GO(1, 16);
    else SCIPNP_GO(1, 32);
#undef SCIPNP_GO
#undef SCIPNP_GO_NT
    return launch_status("pm_project_kernel");
}

// ===================================================================== dual update of iteration k-1 + projection of iteration k
// One launch for the tail of one ADMM iteration and the head of the next (round 3: the ADMM-TV iteration is launch-bound --
// four launches of 3 - 8 us on a 2 MB state): per mosaic pixel and frame
//     theta = clip(theta_raw, 0, 1);  b += / -= x - theta;  [squared error of the reported iterate]      (pm_dual_update_kernel)
//     p = theta - c0 b  /  theta + b;  x = p + Phi^T((y - Phi p) / (c1 + Phisum))                        (pm_project_kernel)
// with exactly the expressions of the two stand-alone kernels, so theta, b and x are bit-identical to running them one
// after the other (x is updated in place: a thread owns its pixels in every frame).  Squared-error partials: one per
// workgroup, entries [gridDim.x, nfill) zeroed, so a caller that sums the nfill partials of pm_dual_update's grid gets the
// total (fp64 sums of exact fp32 squares; the association differs from the stand-alone kernel's in the last bit of the sum).
template <int VEC, int MAXB, int MODE>
__global__ void __launch_bounds__(256)
pm_dual_project_kernel(const float* __restrict__ theta_raw, const TvCandidates cd, int use_cd, double tv_weight, double tv_eps,
                       float* xio, float* theta, float* bb,
                       const float* __restrict__ Phi, const float* __restrict__ y, const float* __restrict__ Phisum,
                       const float* __restrict__ orig, double* sse_part, int nfill, long long Q, long long MN, int B, float c0,
                       float c1, int CH) {
    using V = typename VecT<VEC>::type;
    __shared__ double red[16];
    __shared__ int s_sel[4 * 32];                       // [plane of the workgroup's pixels - first one][frame] (host: <= 4 planes)
    const int P = (int)(Q / MN);                        // planes per frame: 4 Bayer planes x units (unit-batched layout)
    // a workgroup owns CH consecutive chunks of blockDim.x * VEC pixels (CH > 1 only where one workgroup per chunk would be
    // more squared-error partials than the caller's buffer holds)
    const long long per = (long long)blockDim.x * VEC;
    const long long qb = (long long)blockIdx.x * CH * per;                           // first pixel of the workgroup
    const int ib_lo = (int)(qb / MN);
    double acc = 0.0;
    for (int ch = 0; ch < CH; ++ch) {
        const long long q = qb + (long long)ch * per + (long long)threadIdx.x * VEC;
        const bool live = q < Q;
        // operands that do not depend on the stop test first: their latency covers it
        float xr[MAXB][VEC], br[MAXB][VEC], ph[MAXB][VEC];
        if (live) {
#pragma unroll
            for (int t = 0; t < MAXB; ++t) {
                if (t < B) {
                    const size_t o = (size_t)t * Q + q;
                    const V xv = *(const V*)(xio + o), bv = *(const V*)(bb + o), phv = *(const V*)(Phi + o);
                    const float *xp = (const float*)&xv, *bp = (const float*)&bv, *php = (const float*)&phv;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) { xr[t][v] = xp[v]; br[t][v] = bp[v]; ph[t][v] = php[v]; }
                }
            }
        }
        if (use_cd && ch == 0) {
            // TV step in its candidate form: the stop test of every channel this workgroup touches, here (the same sums in the
            // same order as tv_stop_kernel): channel (t, ib) kept the `out` of iteration s_sel -> candidate s_sel - 1
            long long qe = qb + (long long)CH * per - 1;
            if (qe > Q - 1) qe = Q - 1;
            const int nib = qe >= qb ? (int)(qe / MN) - ib_lo + 1 : 0;
            const int wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
            if (cd.nbands <= 8) {           // eight channels per wave
                for (int j0 = wave * 8; j0 < nib * B; j0 += nw * 8) {
                    const int j = j0 + ((threadIdx.x & 63) >> 3);
                    const bool on = j < nib * B;
                    const int t = on ? j % B : 0, ibl = on ? j / B : 0;
                    const int c = t * P + ib_lo + ibl;
                    const int st = tv_band_stop_test8(on ? cd.part + (size_t)c * cd.nbands * 2 * cd.n_iter : nullptr, cd.nbands,
                                                      cd.n_iter, cd.MN, tv_weight, tv_eps);
                    if (on && (threadIdx.x & 7) == 0) s_sel[ibl * 32 + t] = st;
                }
            } else {
                for (int j = wave; j < nib * B; j += nw) {
                    const int t = j % B, ibl = j / B;
                    const int c = t * P + ib_lo + ibl;
                    const int st = tv_band_stop_test(cd.part + (size_t)c * cd.nbands * 2 * cd.n_iter, cd.nbands, cd.n_iter, cd.MN,
                                                     tv_weight, tv_eps);
                    if ((threadIdx.x & 63) == 0) s_sel[ibl * 32 + t] = st;
                }
            }
            __syncthreads();
        }
        if (live) {
            float p[MAXB][VEC];
#pragma unroll
            for (int t = 0; t < MAXB; ++t) {
                if (t < B) {
                    const size_t o = (size_t)t * Q + q;
                    // candidate form of the TV step: channel (t, plane of q) kept the `out` of iteration s_sel (Q/4 = plane size;
                    // the VEC pixels of a thread lie in one plane: Q/4 % VEC == 0 on the vector paths)
                    const size_t ro = use_cd ? (size_t)(s_sel[((int)(q / MN) - ib_lo) * 32 + t] - 1) * ((size_t)B * Q) + o : o;
                    const V rv = *(const V*)(theta_raw + ro);
                    const float *rp = (const float*)&rv, *xp = xr[t], *bp = br[t];
                    V tho, bo;
                    float *thp = (float*)&tho, *bop = (float*)&bo;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float th = fminf(fmaxf(rp[v], 0.f), 1.f);
                        const float d = xp[v] - th;
                        const float bn = (MODE == 0) ? (bp[v] + d) : (bp[v] - d);
                        thp[v] = th;
                        bop[v] = bn;
                        p[t][v] = (MODE == 0) ? (th - c0 * bn) : (th + bn);
                    }
                    *(V*)(theta + o) = tho;
                    *(V*)(bb + o) = bo;
                    if (sse_part) {
                        const V ov = *(const V*)(orig + o);
                        const float* op = (const float*)&ov;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const float e = op[v] - (MODE == 0 ? thp[v] : xp[v]);
                            acc += (double)(e * e);
                        }
                    }
                }
            }
            const V yv = *(const V*)(y + q), sv = *(const V*)(Phisum + q);
            const float *yp = (const float*)&yv, *sp = (const float*)&sv;
            float r[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float yb = torch_contig_sum<MAXB>(B, [&](int i) { return p[i][v] * ph[i][v]; });
                r[v] = (MODE == 0) ? (yp[v] - yb) / (c1 + sp[v]) : (yp[v] - yb) / (sp[v] + c1);
            }
#pragma unroll
            for (int t = 0; t < MAXB; ++t) {
                if (t < B) {
                    V ov;
                    float* op = (float*)&ov;
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        op[v] = (MODE == 0) ? (p[t][v] + ph[t][v] * r[v]) : (p[t][v] + c0 * (r[v] * ph[t][v]));
                    *(V*)(xio + (size_t)t * Q + q) = ov;
                }
            }
        }
    }
    if (sse_part) {
        const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
        if (threadIdx.x == 0) sse_part[blockIdx.x] = s;
        if (blockIdx.x == 0)
            for (int i = gridDim.x + threadIdx.x; i < nfill; i += blockDim.x) sse_part[i] = 0.0;
    }
}

// ---- round 5: the same launch for SMALL states with the candidate form of the TV step (the ADMM-TV iteration at 256x256x8: one
// pixel per thread, a workgroup's 256 pixels inside one plane, at most 8 frames, at most 8 bands per channel), arranged so that
// the kernel is ONE memory round trip deep instead of three.  The general kernel above loads the stop-test operands, evaluates
// the test in wave 0, passes the result through LDS and a barrier, and only then knows which candidate to load (a second round
// trip; y and Phisum a third) -- 12.2 us for 28 MB on a 2 MB state (profiles/r03d_*).  Here every wave loads the stop-test
// operands FIRST, then every other operand including ALL (n_iter - 1) candidates of its pixel (6 MB of extra reads that never
// leave the memory-side cache), evaluates the test itself while those are in flight (no LDS, no barrier; E / MN as an exact
// scaling when M*N is a power of two) and picks the candidate in registers.  (Loading only the candidate the channel kept in the
// previous ADMM iteration, the right one after the test where that prediction fails, measured the same 8.7 us: the kernel is
// as deep as its one round trip, not as wide as its reads -- profiles/r05zf_*.)  Same expressions in the same order as the general
// kernel: theta, b, x and the squared-error partials are bit-identical to it.
// (mask & a) | (~mask & b) with a wave-uniform mask, one v_bfi_b32
__device__ __forceinline__ unsigned bit_select(unsigned mask, float a, unsigned b) {
    unsigned r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
}

#if defined(SCIPNP_TV_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define DPS_STAMP(slot)                                                                                                 \
    do {                                                                                                                \
        if (stamps) {                                                                                                   \
            unsigned long long t_;                                                                                      \
            const unsigned long long* p_ = stamps + (slot);                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : "=&s"(t_) : "s"(p_) : "memory"); \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
    } while (0)
#define DPS_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define DPS_STAMP(slot) (void)0
#define DPS_WAIT_VM() (void)0
#endif
// FULL: exactly 8 frames and 5 TV iterations (4 candidates), SSE: squared-error partials wanted -- compile-time, so that the
// kernel is straight-line code: with run-time `t < B` / `it < n_iter - 1` / `sse_part` tests every frame was its own basic block,
// its loads stayed inside it, and each block waited for the previous block's stores (stamps: 3700 clocks of arithmetic + stores)
template <int MODE, bool FULL, bool SSE>
__global__ void __launch_bounds__(256)
pm_dual_project_spec_kernel(const TvCandidates cd, double tv_weight, double tv_eps, double inv_mn, float* xio, float* theta, float* bb,
                            const float* __restrict__ Phi, const float* __restrict__ y, const float* __restrict__ Phisum,
                            const float* __restrict__ orig, double* sse_part, int nfill, long long Q, long long MN, int B, float c0,
                            float c1
#if defined(SCIPNP_TV_STAMPS)
                            , unsigned long long* stamp_buf          // (clock stamps of one wave: tools/probes/tv_band_stamps.py)
#endif
                            ) {
    constexpr int MAXB = 8;
    __shared__ double red[16];
#if defined(SCIPNP_TV_STAMPS)
    unsigned long long* const stamps = (stamp_buf && blockIdx.x == gridDim.x / 2 + 3 && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) ? stamp_buf : nullptr;
#endif
    DPS_STAMP(0);
    const int P = (int)(Q / MN);                                        // planes per frame
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;      // (host: Q % 256 == 0, MN % 256 == 0)
    const int ib = (int)(((long long)blockIdx.x * 256) / MN);           // the plane of this workgroup's pixels
    const int lane = threadIdx.x & 63, t8 = lane >> 3;
    double s1[TV_STOP_MAXIT], s2[TV_STOP_MAXIT];
    if (FULL) {                       // (8 frames: every lane has a channel; lanes of a band the channel does not have read band 0)
        const int k8 = lane & 7;
        const bool on = k8 < cd.nbands;
        tv_band_stop_load8_full(cd.part + (((size_t)t8 * P + ib) * cd.nbands + (on ? k8 : 0)) * 2 * 5, on, s1, s2);
    } else {
        tv_band_stop_load8(t8 < B ? cd.part + ((size_t)t8 * P + ib) * cd.nbands * 2 * cd.n_iter : nullptr, cd.nbands, cd.n_iter, s1, s2);
    }
    float xr[MAXB], br[MAXB], ph[MAXB], og[MAXB], cr[TV_STOP_MAXIT][MAXB];
    // frame by frame, in the order the arithmetic below consumes them: frame t's operands are back (s_waitcnt vmcnt counts down
    // in issue order) while the later frames' are still in flight
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        const bool on = FULL || t < B;
        const size_t o = (size_t)t * Q + q;
        if (on) {
            xr[t] = tv_handoff_load(xio + o);
            br[t] = tv_handoff_load(bb + o);
            ph[t] = Phi[o];
        }
#pragma unroll
        for (int it = 0; it < TV_STOP_MAXIT; ++it)
            cr[it][t] = (FULL || (on && it < cd.n_iter - 1)) ? tv_handoff_load(cd.cand + (size_t)it * ((size_t)B * Q) + o) : 0.f;
        if (SSE) og[t] = on ? orig[o] : 0.f;
        __builtin_amdgcn_sched_barrier(0);    // (keeps the issue order frame-major: the scheduler clusters loads by base pointer)
    }
    const float yv = y[q], sv = Phisum[q];
    __builtin_amdgcn_sched_barrier(0);        // every load is in flight before the stop test starts
    DPS_STAMP(1);                 // every load issued
    DPS_WAIT_VM();
    DPS_STAMP(2);                 // ... and back
    const int st = tv_band_stop_finish8(s1, s2, cd.n_iter, cd.MN, tv_weight, tv_eps, inv_mn);   // lane 8 t: channel (t, ib)
    DPS_STAMP(3);                 // stop test
    double acc = 0.0;
    float p[MAXB];
#pragma unroll
    for (int t = 0; t < MAXB; ++t) {
        if (FULL || t < B) {
            const size_t o = (size_t)t * Q + q;
            const int sel = __builtin_amdgcn_readlane(st, 8 * t);       // channel kept the `out` of iteration sel: candidate sel - 1
            // (v_bfi_b32 with the wave-uniform `sel` as a scalar mask: as a ?: chain the compiler builds a maze of scalar branches
            // around the waits for the candidates it can skip -- 270 clocks per frame; as and / or it picks v_cndmask on VCC, which
            // issues in 19.5 clocks on this part against 5.5: profiles/r05zh_valu_issue_cost.txt)
            unsigned rb = __builtin_bit_cast(unsigned, cr[3][t]);
            rb = bit_select(sel == 3 ? ~0u : 0u, cr[2][t], rb);
            rb = bit_select(sel == 2 ? ~0u : 0u, cr[1][t], rb);
            rb = bit_select(sel == 1 ? ~0u : 0u, cr[0][t], rb);
            const float raw = __builtin_bit_cast(float, rb);
            const float th = fminf(fmaxf(raw, 0.f), 1.f);
            const float d = xr[t] - th;
            const float bn = (MODE == 0) ? (br[t] + d) : (br[t] - d);
            p[t] = (MODE == 0) ? (th - c0 * bn) : (th + bn);
            tv_handoff_store(theta + o, th);
            tv_handoff_store(bb + o, bn);
            if (SSE) {
                const float e = og[t] - (MODE == 0 ? th : xr[t]);
                acc += (double)(e * e);
            }
        }
    }
    const float yb = torch_contig_sum<MAXB>(FULL ? MAXB : B, [&](int i) { return p[i] * ph[i]; });
    const float r = (MODE == 0) ? (yv - yb) / (c1 + sv) : (yv - yb) / (sv + c1);
#pragma unroll
    for (int t = 0; t < MAXB; ++t)
        if (FULL || t < B) tv_handoff_store(xio + (size_t)t * Q + q, (MODE == 0) ? (p[t] + ph[t] * r) : (p[t] + c0 * (r * ph[t])));
    DPS_STAMP(4);                 // arithmetic, stores issued
    if (SSE) {
        const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
        if (threadIdx.x == 0) tv_handoff_store(sse_part + blockIdx.x, s);
        if (blockIdx.x == 0)
            for (int i = gridDim.x + threadIdx.x; i < nfill; i += blockDim.x) sse_part[i] = 0.0;
    }
    DPS_STAMP(5);
    DPS_WAIT_VM();
    DPS_STAMP(6);                 // stores acknowledged
#if defined(SCIPNP_TV_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (stamps) asm volatile("s_dcache_wb" ::: "memory");
#endif
}

// launch shape of pm_dual_project_kernel: pixels per thread, chunks per workgroup (more than one only where one workgroup per
// chunk would be more squared-error partials than the caller's nfill entries), workgroups
void dual_project_shape(long long Q, int B, int nfill, bool vec_ok, int* VEC, int* CH, unsigned* grid, int units) {
    const int threads = 256;
    const bool wide = Q / 4 >= 512LL * threads;               // (as launch_pm_project: small states take one pixel per thread)
    const int v = (vec_ok && wide) ? (B <= 16 ? 4 : 2) : 1;
    const long long nchunks = (Q / v + threads - 1) / threads;
    int ch = nfill > 0 ? (int)((nchunks + nfill - 1) / nfill) : 1;
    // unit batches cut the squared-error partials at unit boundaries: a workgroup must not straddle two units, so its pixel
    // count (ch * 256 * v) is kept a power of two (B = 3 gave ch = 3: 768 pixels against units of 2048 k pixels)
    if (units > 1) { int p2 = 1; while (p2 < ch) p2 <<= 1; ch = p2; }
    *VEC = v;
    *CH = ch;
    *grid = (unsigned)((nchunks + ch - 1) / ch);
}

// ===================================================================== dual update (+ SSE partials)
constexpr int RED_THREADS = 256;
constexpr int RED_PER_THREAD = 8;

__global__ void __launch_bounds__(RED_THREADS)
pm_dual_update_kernel(const float* __restrict__ theta_raw, const int32_t* __restrict__ sel, long long MN,
                      const float* __restrict__ x,
                      float* theta, float* b, const float* __restrict__ orig, double* sse_part,
                      int which, float sign, long long total) {
    __shared__ double red[16];
    double acc = 0.0;
    const long long base = (long long)blockIdx.x * (RED_THREADS * RED_PER_THREAD);
#pragma unroll
    for (int k = 0; k < RED_PER_THREAD; ++k) {
        const long long i = base + (long long)k * RED_THREADS + threadIdx.x;
        if (i < total) {
            const float raw = sel ? theta_raw[(long long)(sel[i / MN] - 1) * total + i] : theta_raw[i];
            const float xv = x[i];
            const float th = fminf(fmaxf(raw, 0.f), 1.f);
            const float d = xv - th;
            b[i] = (sign > 0.f) ? (b[i] + d) : (b[i] - d);
            theta[i] = th;
            if (sse_part) {
                const float e = orig[i] - (which == 0 ? th : xv);
                acc += (double)(e * e);
            }
        }
    }
    if (sse_part) {
        const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
        if (threadIdx.x == 0) sse_part[blockIdx.x] = s;
    }
}

__global__ void __launch_bounds__(RED_THREADS)
sse_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, double* part) {
    __shared__ double red[16];
    double acc = 0.0;
    const size_t base = (size_t)blockIdx.x * (RED_THREADS * RED_PER_THREAD);
#pragma unroll
    for (int k = 0; k < RED_PER_THREAD; ++k) {
        const size_t i = base + (size_t)k * RED_THREADS + threadIdx.x;
        if (i < n) {
            const float e = a[i] - b[i];
            acc += (double)(e * e);
        }
    }
    const double s = block_sum_double(acc, red, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

int scipnp_A(const float* x, const float* Phi, float* y, int M, int N, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && Phi && y, "null pointer");
    SCIPNP_ALIGNED(x); SCIPNP_ALIGNED(Phi); SCIPNP_ALIGNED(y);
    return launch_ref_layout<2>(x, nullptr, Phi, nullptr, nullptr, nullptr, y, M, N, B, 0.f, 0.f, (hipStream_t)s);
}

int scipnp_At(const float* y, const float* Phi, float* x, int M, int N, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && Phi && y, "null pointer");
    SCIPNP_ALIGNED(x); SCIPNP_ALIGNED(Phi); SCIPNP_ALIGNED(y);
    return launch_ref_layout<3>(nullptr, nullptr, Phi, y, nullptr, x, nullptr, M, N, B, 0.f, 0.f, (hipStream_t)s);
}

int scipnp_phisum(const float* Phi, float* Phisum, int M, int N, int B, scipnp_stream_t s) {
    SCIPNP_REQUIRE(Phi && Phisum, "null pointer");
    SCIPNP_ALIGNED(Phi); SCIPNP_ALIGNED(Phisum);
    return launch_ref_layout<4>(nullptr, nullptr, Phi, nullptr, nullptr, nullptr, Phisum, M, N, B, 0.f, 0.f, (hipStream_t)s);
}

int scipnp_proj_twostage(const float* theta, const float* b, const float* Phi, const float* y,
                         const float* Phisum, float* x, int M, int N, int B, float inv_rho,
                         float alpha_rho, scipnp_stream_t s) {
    SCIPNP_REQUIRE(theta && b && Phi && y && Phisum && x, "null pointer");
    SCIPNP_ALIGNED(theta); SCIPNP_ALIGNED(b); SCIPNP_ALIGNED(Phi); SCIPNP_ALIGNED(y); SCIPNP_ALIGNED(Phisum); SCIPNP_ALIGNED(x);
    return launch_ref_layout<0>(theta, b, Phi, y, Phisum, x, nullptr, M, N, B, inv_rho, alpha_rho, (hipStream_t)s);
}

int scipnp_proj_onestage(const float* theta, const float* b, const float* Phi, const float* y,
                         const float* Phisum, float* x, int M, int N, int B, float lambda, float gamma,
                         scipnp_stream_t s) {
    SCIPNP_REQUIRE(theta && b && Phi && y && Phisum && x, "null pointer");
    SCIPNP_ALIGNED(theta); SCIPNP_ALIGNED(b); SCIPNP_ALIGNED(Phi); SCIPNP_ALIGNED(y); SCIPNP_ALIGNED(Phisum); SCIPNP_ALIGNED(x);
    return launch_ref_layout<1>(theta, b, Phi, y, Phisum, x, nullptr, M, N, B, lambda, gamma, (hipStream_t)s);
}

static int bayer_reorder(const float* src, float* dst, int M, int N, int B, bool split, hipStream_t st) {
    SCIPNP_REQUIRE(src && dst && M > 0 && N > 0 && B > 0, "bad arguments");
    SCIPNP_ALIGNED(split ? dst : src);
    const long long total = (long long)M * N * B;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (split) hipLaunchKernelGGL(bayer_reorder_kernel<true>, dim3(blocks), dim3(256), 0, st, src, dst, M, N, B);
    else hipLaunchKernelGGL(bayer_reorder_kernel<false>, dim3(blocks), dim3(256), 0, st, src, dst, M, N, B);
    return launch_status("bayer_reorder_kernel");
}

int scipnp_bayer_split(const float* mosaic, float* planes, int M, int N, int B, scipnp_stream_t s) {
    return bayer_reorder(mosaic, planes, M, N, B, true, (hipStream_t)s);
}
int scipnp_bayer_merge(const float* planes, float* mosaic, int M, int N, int B, scipnp_stream_t s) {
    return bayer_reorder(planes, mosaic, M, N, B, false, (hipStream_t)s);
}

static int state_mosaic(const float* src, float* dst, int M, int N, int B, bool to_state, hipStream_t st) {
    SCIPNP_REQUIRE(src && dst && M > 0 && N > 0 && M <= 65535 && B > 0 && B <= 64 * 65535, "bad arguments");
    const dim3 grid((N + CONV_TQ - 1) / CONV_TQ, M, (B + CONV_FC - 1) / CONV_FC);
    const size_t lds = (size_t)2 * 2 * CONV_TQ * ((B < CONV_FC ? B : CONV_FC) + 1) * sizeof(float);
    if (to_state) hipLaunchKernelGGL(state_mosaic_kernel<true>, grid, dim3(256), lds, st, src, dst, M, N, B);
    else hipLaunchKernelGGL(state_mosaic_kernel<false>, grid, dim3(256), lds, st, src, dst, M, N, B);
    return launch_status("state_mosaic_kernel");
}

int scipnp_mosaic_to_state(const float* mosaic, float* state, int M, int N, int B, scipnp_stream_t s) {
    return state_mosaic(mosaic, state, M, N, B, true, (hipStream_t)s);
}
int scipnp_state_to_mosaic(const float* state, float* mosaic, int M, int N, int B, scipnp_stream_t s) {
    return state_mosaic(state, mosaic, M, N, B, false, (hipStream_t)s);
}

int scipnp_y_to_meas(const float* y, float* meas, int M, int N, scipnp_stream_t s) {
    SCIPNP_REQUIRE(y && meas && M > 0 && N > 0, "bad arguments");
    const long long total = 4LL * M * N;
    hipLaunchKernelGGL(y_to_meas_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, y, meas, M, N);
    return launch_status("y_to_meas_kernel");
}

static int rgb_cube(const float* src, float* dst, int H, int W, int B, bool to_cube, hipStream_t st) {
    SCIPNP_REQUIRE(src && dst && H > 0 && W > 0 && H <= 65535 && B > 0 && B <= 64 * 65535, "bad arguments");
    const dim3 grid((W + 63) / 64, H, (B + CONV_FC - 1) / CONV_FC);
    const size_t lds = (size_t)64 * (3 * (B < CONV_FC ? B : CONV_FC) + 1) * sizeof(float);
    if (to_cube) hipLaunchKernelGGL(rgb_cube_kernel<true>, grid, dim3(256), lds, st, src, dst, H, W, B);
    else hipLaunchKernelGGL(rgb_cube_kernel<false>, grid, dim3(256), lds, st, src, dst, H, W, B);
    return launch_status("rgb_cube_kernel");
}
int scipnp_rgb_to_cube(const float* rgb, float* cube, int H, int W, int B, scipnp_stream_t s) {
    return rgb_cube(rgb, cube, H, W, B, true, (hipStream_t)s);
}
int scipnp_cube_to_rgb(const float* cube, float* rgb, int H, int W, int B, scipnp_stream_t s) {
    return rgb_cube(cube, rgb, H, W, B, false, (hipStream_t)s);
}

int scipnp_pm_setup(const float* Phi, const float* y, float* Phisum, float* x0, int M, int N, int B,
                    scipnp_stream_t s) {
    SCIPNP_REQUIRE(Phi && y && Phisum, "null pointer");
    return launch_pm_project<2>(nullptr, nullptr, Phi, y, nullptr, Phisum, x0, M, N, B, 0.f, 0.f, (hipStream_t)s);
}

int scipnp_pm_project(const float* theta, const float* b, const float* Phi, const float* y,
                      const float* Phisum, float* x, int M, int N, int B, int mode, float c0, float c1,
                      scipnp_stream_t s) {
    SCIPNP_REQUIRE(theta && b && Phi && y && Phisum && x, "null pointer");
    SCIPNP_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (two-stage) or 1 (one-stage)");
    if (mode == 0) return launch_pm_project<0>(theta, b, Phi, y, Phisum, nullptr, x, M, N, B, c0, c1, (hipStream_t)s);
    return launch_pm_project<1>(theta, b, Phi, y, Phisum, nullptr, x, M, N, B, c0, c1, (hipStream_t)s);
}

int scipnp_pm_dual_update(const float* theta_raw, const float* x, float* theta, float* b,
                          const float* orig, double* sse_part, int which, float sign, int M, int N, int B,
                          int* nblocks, scipnp_stream_t s) {
    return pm_dual_update_sel(theta_raw, nullptr, x, theta, b, orig, sse_part, which, sign, M, N, B, nblocks, (hipStream_t)s, 1);
}

/* unit-batched layout (include/scipnp.h, "Unit batches"): U problems of one shape, state [B][U][4][M][N], y / Phisum [U][4][M][N] */
int scipnp_pm_setup_units(const float* Phi, const float* y, float* Phisum, float* x0, int M, int N, int B, int units,
                          scipnp_stream_t s) {
    SCIPNP_REQUIRE(Phi && y && Phisum, "null pointer");
    return launch_pm_project<2>(nullptr, nullptr, Phi, y, nullptr, Phisum, x0, M, N, B, 0.f, 0.f, (hipStream_t)s, units);
}

int scipnp_pm_project_units(const float* theta, const float* b, const float* Phi, const float* y, const float* Phisum, float* x,
                            int M, int N, int B, int units, int mode, float c0, float c1, scipnp_stream_t s) {
    SCIPNP_REQUIRE(theta && b && Phi && y && Phisum && x, "null pointer");
    SCIPNP_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (two-stage) or 1 (one-stage)");
    if (mode == 0) return launch_pm_project<0>(theta, b, Phi, y, Phisum, nullptr, x, M, N, B, c0, c1, (hipStream_t)s, units);
    return launch_pm_project<1>(theta, b, Phi, y, Phisum, nullptr, x, M, N, B, c0, c1, (hipStream_t)s, units);
}

}  // extern "C"

namespace scipnp {
int pm_dual_update_sel(const float* theta_raw, const int32_t* sel, const float* x, float* theta, float* b, const float* orig,
                       double* sse_part, int which, float sign, int M, int N, int B, int* nblocks, hipStream_t st, int units) {
    SCIPNP_REQUIRE(theta_raw && x && theta && b, "null pointer");
    SCIPNP_REQUIRE((sse_part == nullptr) || (orig != nullptr), "sse_part needs orig");
    SCIPNP_REQUIRE(units >= 1, "units must be >= 1");
    const long long total = 4LL * M * N * B * units;
    const int per = RED_THREADS * RED_PER_THREAD;
    const unsigned blocks = (unsigned)((total + per - 1) / per);
    if (nblocks) *nblocks = (int)blocks;
    hipLaunchKernelGGL(pm_dual_update_kernel, dim3(blocks), dim3(RED_THREADS), 0, st, theta_raw, sel, (long long)M * N, x,
                       theta, b, orig, sse_part, which, sign, total);
    return launch_status("pm_dual_update_kernel");
}
}  // namespace scipnp

extern "C" {

int scipnp_pm_dual_project_fits(int M, int N, int B) { return M > 0 && N > 0 && B > 0 && B <= 32; }

/* workgroups (= squared-error partials actually written; the rest of the nfill entries are zeros) of the fused launch on
 * `units` problems of M x N x B with 16-byte aligned buffers -- for a caller that cuts the partials at unit boundaries */
int scipnp_pm_dual_project_blocks(int M, int N, int B, int units, int nfill) {
    if (M <= 0 || N <= 0 || B <= 0 || B > 32 || units < 1) return 0;
    int vec, ch;
    unsigned grid;
    scipnp::dual_project_shape(4LL * M * N * units, B, nfill, (long long)M * N % 4 == 0, &vec, &ch, &grid, units);
    return (int)grid;
}

int scipnp_pm_dual_project(const float* theta_raw, float* x, float* theta, float* b, const float* Phi, const float* y,
                           const float* Phisum, const float* orig, double* sse_part, int nfill, int M, int N, int B, int mode,
                           float c0, float c1, scipnp_stream_t s) {
    return pm_dual_project_sel(theta_raw, nullptr, 0.0, 0.0, x, theta, b, Phi, y, Phisum, orig, sse_part, nfill, M, N, B, mode, c0,
                               c1, (hipStream_t)s, 1);
}

}  // extern "C"

namespace scipnp {
int pm_dual_project_sel(const float* theta_raw, const TvCandidates* cdp, double tv_weight, double tv_eps, float* x, float* theta,
                        float* b, const float* Phi, const float* y, const float* Phisum, const float* orig, double* sse_part,
                        int nfill, int M, int N, int B, int mode, float c0, float c1, hipStream_t st, int units) {
    SCIPNP_REQUIRE(theta_raw && x && theta && b && Phi && y && Phisum, "null pointer");
    SCIPNP_REQUIRE(units >= 1, "units must be >= 1");
    TvCandidates cd = {};
    const int use_cd = cdp != nullptr;
    if (use_cd) cd = *cdp;
    SCIPNP_REQUIRE(!use_cd || cd.n_iter - 1 <= 4, "candidate form: at most 5 TV iterations");
    SCIPNP_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (two-stage) or 1 (one-stage)");
    SCIPNP_REQUIRE(scipnp_pm_dual_project_fits(M, N, B), "fused dual update + projection: 1 <= B <= 32 (got %d)", B);
    SCIPNP_REQUIRE((sse_part == nullptr) || (orig != nullptr), "sse_part needs orig");
    const long long Q = 4LL * M * N * units, MN = (long long)M * N;
    // (vector paths: every thread's pixels in one Bayer plane and every candidate plane set 16-byte aligned)
    const bool vec = ((long long)M * N % 4 == 0) && aligned16(Phi) && aligned16(y) && aligned16(x) && aligned16(theta) && aligned16(b) &&
                     aligned16(Phisum) && aligned16(theta_raw) && (orig == nullptr || aligned16(orig));
    const int threads = 256;
    int VECs, CH;
    unsigned grid;
    dual_project_shape(Q, B, sse_part ? nfill : 0, vec, &VECs, &CH, &grid, units);
    SCIPNP_REQUIRE(sse_part == nullptr || nfill > 0, "sse_part needs nfill > 0");
    // the candidate form keeps the stop iteration of at most 4 planes per workgroup (always true for one unit)
    SCIPNP_REQUIRE(!use_cd || units == 1 || ((long long)CH * threads * VECs - 1) / MN + 2 <= 4,
                   "unit-batched candidate form: planes of %lld pixels are too small for this launch shape", MN);
    // small states in the candidate form: the one-round-trip kernel (laboratory builds: SCIPNP_DUAL_PROJECT_GENERAL=1 keeps the general one, for A/B)
    static const bool general_only = [] { const char* e = lab_switch("SCIPNP_DUAL_PROJECT_GENERAL"); return e && e[0] == '1'; }();
    if (use_cd && !general_only && VECs == 1 && CH == 1 && B <= 8 && cd.nbands <= 8 && MN % 256 == 0 && Q % 256 == 0) {
        const double inv_mn = (MN & (MN - 1)) == 0 ? 1.0 / (double)MN : 0.0;
#if defined(SCIPNP_TV_STAMPS)
        static unsigned long long* const stamp_buf = [] { const char* e = getenv("SCIPNP_STAMP_PTR"); return e ? (unsigned long long*)strtoull(e, nullptr, 16) : nullptr; }();
#define SCIPNP_DPS_EXTRA , stamp_buf
#else
#define SCIPNP_DPS_EXTRA
#endif
        const bool full = B == 8 && cd.n_iter == 5, sse = sse_part != nullptr;
#define SCIPNP_DPS(MODE, FULL, SSE)                                                                                        \
        hipLaunchKernelGGL((pm_dual_project_spec_kernel<MODE, FULL, SSE>), dim3(grid), dim3(threads), 0, st, cd, tv_weight, tv_eps,  \
                           inv_mn, x, theta, b, Phi, y, Phisum, orig, sse_part, nfill, Q, MN, B, c0, c1 SCIPNP_DPS_EXTRA)
        if (mode == 0) {
            if (full) { if (sse) SCIPNP_DPS(0, true, true); else SCIPNP_DPS(0, true, false); }
            else { if (sse) SCIPNP_DPS(0, false, true); else SCIPNP_DPS(0, false, false); }
        } else {
            if (full) { if (sse) SCIPNP_DPS(1, true, true); else SCIPNP_DPS(1, true, false); }
            else { if (sse) SCIPNP_DPS(1, false, true); else SCIPNP_DPS(1, false, false); }
        }
#undef SCIPNP_DPS
#undef SCIPNP_DPS_EXTRA
        return launch_status("pm_dual_project_spec_kernel");
    }
#define SCIPNP_DP(VEC, MAXB)                                                                                     \
    do {                                                                                                         \
        if (mode == 0)                                                                                           \
            hipLaunchKernelGGL((pm_dual_project_kernel<VEC, MAXB, 0>), dim3(grid), dim3(threads), 0, st, theta_raw, cd, use_cd, \
                               tv_weight, tv_eps, x, theta, b, Phi, y, Phisum, orig, sse_part, nfill, Q, MN, B, c0, c1, CH); \
        else                                                                                                     \
            hipLaunchKernelGGL((pm_dual_project_kernel<VEC, MAXB, 1>), dim3(grid), dim3(threads), 0, st, theta_raw, cd, use_cd, \
                               tv_weight, tv_eps, x, theta, b, Phi, y, Phisum, orig, sse_part, nfill, Q, MN, B, c0, c1, CH); \
    } while (0)
    if (VECs == 4 && B <= 8) SCIPNP_DP(4, 8);
    else if (VECs == 4) SCIPNP_DP(4, 16);
    else if (VECs == 2) SCIPNP_DP(2, 32);
    else if (B <= 8) SCIPNP_DP(1, 8);
    else if (B <= 16) SCIPNP_DP(1, 16);
    else SCIPNP_DP(1, 32);
#undef SCIPNP_DP
    return launch_status("pm_dual_project_kernel");
}
}  // namespace scipnp

extern "C" {

int scipnp_sse_partials(const float* a, const float* b, size_t n, double* part, int* nblocks,
                        scipnp_stream_t s) {
    SCIPNP_REQUIRE(a && b && nblocks, "null pointer");
    const size_t per = (size_t)RED_THREADS * RED_PER_THREAD;
    const unsigned blocks = (unsigned)((n + per - 1) / per);
    *nblocks = (int)blocks;
    if (part == nullptr) return SCIPNP_OK;  // size query
    hipLaunchKernelGGL(sse_kernel, dim3(blocks), dim3(RED_THREADS), 0, (hipStream_t)s, a, b, n, part);
    return launch_status("sse_kernel");
}

}  // extern "C"
