// Chambolle total-variation prior on C independent M x N channels (plane-major: every channel is
// one contiguous image).  Replaces the reference's per-iteration device->host copy, single-thread
// NumPy `skimage.restoration.denoise_tv_chambolle(..., n_iter_max=5, multichannel=True)` and
// host->device copy (dvp_linear_inv_2_stage_ADMM_tensor_online.py:153-160, :403-407).
//
// One launch per inner iteration i (the data-dependent stop needs the global energy E_i of each
// channel).  Kernel i first reduces the fp64 block partials that kernel i-1 left for its channel,
// evaluates skimage's stop test for iteration i-1 exactly as the NumPy-1.x code does (two float32
// array sums combined in double, see oracle/tv_chambolle.py) and returns if the channel has
// stopped -- theta then already holds the `out` of the stopping iteration.  Every block of a
// channel computes the same decision from the same numbers in the same order: deterministic, no
// atomics, no host round trip.
//
// Arithmetic follows skimage operation by operation in float32 (-ffp-contract=off):
//   d   = -(p0+p1) (+ p0[r-1,c] for r>0) (+ p1[r,c-1] for c>0);   out = v + d        (i > 0)
//   g0  = out[r+1,c]-out[r,c] (0 on the last row),  g1 likewise;  nrm = sqrt(g0*g0+g1*g1)
//   p  <- (p - 0.25 g) / (1 + nrm * (0.25/weight))
#include "common.hpp"

namespace scipnp {

// Tile = 16 rows x 256 columns (block = 256 x 4 threads, 4 rows per thread).  Wide and flat on purpose: a halo ROW is one
// contiguous KiB per array, a halo COLUMN is one 128-byte line per element -- with 32 x 32 tiles the five column halos
// made the kernel fetch 2.65x its algorithmic reads (88.7 MB against 33.5 MB per launch at 512 x 512 x 8, rocprofv3
// FETCH_SIZE, profiles/r01h_pmc_traffic.json) at the HBM ceiling; quarter-resolution planes up to 256 wide now have no
// column halo at all.
constexpr int TV_TSX = 256;      // tile columns
constexpr int TV_TSY = 16;       // tile rows
constexpr int TV_TY = 4;
constexpr int TV_RPT = TV_TSY / TV_TY;

struct TvWorkspace {
    float* p[2];        // ping-pong dual field: [2][C][M][N] each (component-major)
    double* partial;    // [n_iter][C][nblk][2]   (sum d^2, sum nrm)
    double* energy;     // [n_iter][C]
    int* stopped;       // [n_iter][C]
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t tv_layout(int M, int N, int C, int n_iter, void* base, TvWorkspace* ws) {
    const size_t img = (size_t)C * M * N;
    const int nblk = ((M + TV_TSY - 1) / TV_TSY) * ((N + TV_TSX - 1) / TV_TSX);
    size_t off = 0;
    char* b = (char*)base;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return b ? b + o : nullptr; };
    float* p0 = (float*)take(2 * img * sizeof(float));
    float* p1 = (float*)take(2 * img * sizeof(float));
    const int nblk_part = nblk > (M + 7) / 8 ? nblk : (M + 7) / 8;                  // (the banded kernels: up to ceil(M/8) bands per channel)
    double* part = (double*)take((size_t)n_iter * C * nblk_part * 2 * sizeof(double));
    double* en = (double*)take((size_t)n_iter * C * sizeof(double));
    int* st = (int*)take((size_t)n_iter * C * sizeof(int));
    if (ws) { ws->p[0] = p0; ws->p[1] = p1; ws->partial = part; ws->energy = en; ws->stopped = st; }
    return off;
}

__device__ __forceinline__ float tv_input(const float* __restrict__ x, const float* __restrict__ b, float coef,
                                          size_t o) {
    return b ? (x[o] + coef * b[o]) : x[o];
}
// (the band kernels of round 5: x and b are the previous launch's hand-off state)
__device__ __forceinline__ float tv_input_handoff(const float* __restrict__ x, const float* __restrict__ b, float coef, size_t o) {
    return b ? (tv_handoff_load(x + o) + coef * tv_handoff_load(b + o)) : tv_handoff_load(x + o);
}

// `out` of the current iteration at (r,c) from the previous dual field (global memory, cache-served)
// (x, b, p0, p1 already point at this channel's image)
template <bool FIRST>
__device__ __forceinline__ float tv_out_at(const float* __restrict__ x, const float* __restrict__ b, float coef,
                                           const float* __restrict__ p0, const float* __restrict__ p1,
                                           int r, int c, int N, float* d_out) {
    const size_t o = (size_t)r * N + c;
    const float v = tv_input(x, b, coef, o);
    if (FIRST) { *d_out = 0.f; return v; }
    float d = -(p0[o] + p1[o]);
    if (r > 0) d = d + p0[o - N];
    if (c > 0) d = d + p1[o - 1];
    *d_out = d;
    return v + d;
}

template <bool FIRST>
__global__ void __launch_bounds__(TV_TSX* TV_TY)
tv_iter_kernel(const float* __restrict__ x, const float* __restrict__ b, float coef, float* __restrict__ theta,
               TvWorkspace ws, int it, int M, int N, int C, double weight, float tau_over_w, double eps,
               int32_t* stop_iter) {
    __shared__ float s_out[TV_TSY + 1][TV_TSX + 1];
    __shared__ double red[16];
    __shared__ int s_stop;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * TV_TSX + tx;
    const int c = blockIdx.z;
    const int ntx = gridDim.x, nty = gridDim.y;
    const int nblk = ntx * nty;
    const int blk = blockIdx.y * ntx + blockIdx.x;
    const size_t img = (size_t)M * N;
    const size_t chan = (size_t)c * img;

    const float* p0 = ws.p[(it + 1) & 1] + chan;          // written by iteration it-1
    const float* p1 = p0 + (size_t)C * img;
    float* q0 = ws.p[it & 1] + chan;
    float* q1 = q0 + (size_t)C * img;
    const int r0 = blockIdx.y * TV_TSY, c0 = blockIdx.x * TV_TSX;
    const int col = c0 + tx;
    const float* xc = x + chan;
    const float* bc = b ? b + chan : nullptr;

    float dloc[TV_RPT];
#pragma unroll
    for (int k = 0; k < TV_RPT; ++k) {
        const int lr = ty * TV_RPT + k, r = r0 + lr;
        dloc[k] = 0.f;
        if (r < M && col < N)
            s_out[lr][tx] = tv_out_at<FIRST>(xc, bc, coef, p0, p1, r, col, N, &dloc[k]);
    }
    // right halo column (tx == 0 threads of each row group) and bottom halo row (ty == 0 row)
    float dummy;
    if (tx < TV_RPT) {
        const int lr = ty * TV_RPT + tx, r = r0 + lr, cc = c0 + TV_TSX;
        if (r < M && cc < N) s_out[lr][TV_TSX] = tv_out_at<FIRST>(xc, bc, coef, p0, p1, r, cc, N, &dummy);
    }
    if (ty == TV_TY - 1) {
        const int r = r0 + TV_TSY;
        if (r < M && col < N) s_out[TV_TSY][tx] = tv_out_at<FIRST>(xc, bc, coef, p0, p1, r, col, N, &dummy);
    }
    if (!FIRST) {
        // ---- evaluate iteration it-1 for this channel (skimage's stop test), identically in every block: wave 0 sums the
        // channel's block partials (lane l takes k = l, l+64, ... in order, then a fixed shuffle tree) while the tile
        // loads issued above are in flight; one barrier publishes the tile and the decision
        if (tid < 64) {
            const double* part = ws.partial + ((size_t)(it - 1) * C + c) * nblk * 2;
            double s1 = 0.0, s2 = 0.0;
            for (int k = tid; k < nblk; k += 64) { s1 += part[2 * k]; s2 += part[2 * k + 1]; }
            for (int off = 32; off > 0; off >>= 1) {
                s1 += __shfl_down(s1, off, 64);
                s2 += __shfl_down(s2, off, 64);
            }
            if (tid == 0) {
                // float32 array sums (held exactly: rounded once to float) then double arithmetic, as NumPy 1.x does
                double E = (double)(float)s1;
                E += weight * (double)(float)s2;           // `weight` is the Python double of the reference: see host
                E /= (double)img;
                int stopped = 0;
                if (it - 1 >= 1) {
                    const int was = ws.stopped[(size_t)(it - 2) * C + c];
                    const double E0 = ws.energy[c];
                    const double Eprev = ws.energy[(size_t)(it - 2) * C + c];
                    stopped = was || (fabs(Eprev - E) < eps * E0);
                    if (stopped && !was && stop_iter && blk == 0) stop_iter[c] = it - 1;
                }
                if (blk == 0) {
                    ws.energy[(size_t)(it - 1) * C + c] = E;
                    ws.stopped[(size_t)(it - 1) * C + c] = stopped;
                }
                s_stop = stopped;
            }
        }
    } else if (stop_iter && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        stop_iter[c] = -1;  // patched to n_iter_max-1 by the host wrapper's last launch (see below)
    }
    __syncthreads();
    if (!FIRST && s_stop) return;

    double acc1 = 0.0, acc2 = 0.0;
#pragma unroll
    for (int k = 0; k < TV_RPT; ++k) {
        const int lr = ty * TV_RPT + k, r = r0 + lr;
        if (r < M && col < N) {
            const size_t o = (size_t)r * N + col;
            const float out = s_out[lr][tx];
            const float g0 = (r < M - 1) ? (s_out[lr + 1][tx] - out) : 0.f;
            const float g1 = (col < N - 1) ? (s_out[lr][tx + 1] - out) : 0.f;
            float nrm = sqrtf(g0 * g0 + g1 * g1);
            acc1 += (double)(dloc[k] * dloc[k]);
            acc2 += (double)nrm;
            nrm = nrm * tau_over_w;
            nrm = nrm + 1.f;
            const float pp0 = FIRST ? 0.f : p0[o];
            const float pp1 = FIRST ? 0.f : p1[o];
            q0[o] = (pp0 - 0.25f * g0) / nrm;
            q1[o] = (pp1 - 0.25f * g1) / nrm;
            theta[chan + o] = out;
        }
    }
    acc1 = block_sum_double(acc1, red, tid, TV_TSX * TV_TY);
    acc2 = block_sum_double(acc2, red, tid, TV_TSX * TV_TY);
    if (tid == 0) {
        double* part = ws.partial + (((size_t)it * C + c) * nblk + blk) * 2;
        part[0] = acc1;
        part[1] = acc2;
    }
}

// ---- round 5: the dual update of R pixels of one thread written out (used by the banded kernel tv_band_run2 and the whole-plane
// kernel): see the comment on tv_band_run2 below for what is shared and guarded
typedef float tv_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float tv_lane_prev(float v) {      // lane l <- lane l - 1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float tv_lane_next(float v) {      // lane l <- lane l + 1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// guard word of a non-negative-or-signed operand: (|bits| << 1) - 1 -- zero maps to the largest unsigned, anything in
// (0, 2^-60) below TVB_GUARD
constexpr unsigned TVB_GUARD = (0x21800000u << 1) - 1u;      // bits(2^-60) = 0x21800000
__device__ __forceinline__ unsigned tv_guard_word(float f) { return (__builtin_bit_cast(unsigned, f) << 1) - 1u; }
constexpr unsigned TVB_BIG = 0x71800000u + TVB_GUARD;     // bits(2^100) = 0x71800000; big <= 0x7FFFFFFF: no wrap-around

// the p-update of R pixels of one thread: g[k] = (g0, g1), pz[k] = (p0, p1) in, pz out; nrm[k] out; returns the guard word
template <int R>
__device__ __forceinline__ unsigned tv_p_update_fast(const tv_f2 (&g)[R], tv_f2 (&pz)[R], float (&nrm)[R], float tau_over_w) {
    float x[R], s[R];
    tv_f2 num[R];
    unsigned guard = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const tv_f2 sq = g[k] * g[k];
        x[k] = sq.x + sq.y;
        const tv_f2 t = g[k] * 0.25f;
        num[k] = pz[k] - t;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) s[k] = __builtin_amdgcn_sqrtf(x[k]);
    unsigned big = 0u;                       // largest gradient energy of the thread's pixels (x >= 0: the bit pattern is monotonic)
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const unsigned a = tv_guard_word(x[k]), b = tv_guard_word(num[k].x), c = tv_guard_word(num[k].y);
        guard = min(guard, min(a, min(b, c)));
        big = max(big, __builtin_bit_cast(unsigned, x[k]));
    }
    // upper side: an energy above 2^100 (unnormalised input of ~1e15 and more; infinities and NaNs of an overflowed g*g included)
    // would put the denominator where v_rcp_f32 returns a denormal or zero and the refinement makes NaNs of what the IEEE division
    // rounds to zero -- such a wave takes the general path too: TVB_BIG - big drops below TVB_GUARD exactly when big > bits(2^100)
    guard = min(guard, TVB_BIG - big);
#pragma unroll
    for (int k = 0; k < R; ++k) {            // v_sqrt_f32 is within 1 ulp: pick among s - 1 ulp, s, s + 1 ulp by the sign of the residuals
        const float sm = __builtin_bit_cast(float, __builtin_bit_cast(int, s[k]) - 1);
        const float sp = __builtin_bit_cast(float, __builtin_bit_cast(int, s[k]) + 1);
        const float r1 = __builtin_fmaf(-sm, s[k], x[k]);
        const float r2 = __builtin_fmaf(-sp, s[k], x[k]);
        float q = (0.f >= r1) ? sm : s[k];
        q = (0.f < r2) ? sp : q;
        nrm[k] = q;
    }
    float den[R], rc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        float d = nrm[k] * tau_over_w;
        den[k] = d + 1.f;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) rc[k] = __builtin_amdgcn_rcpf(den[k]);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const float e = __builtin_fmaf(-den[k], rc[k], 1.f);
        rc[k] = __builtin_fmaf(e, rc[k], rc[k]);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const tv_f2 r = {rc[k], rc[k]}, nd = {-den[k], -den[k]};
        tv_f2 q = num[k] * r;
        tv_f2 rem = __builtin_elementwise_fma(nd, q, num[k]);
        q = __builtin_elementwise_fma(rem, r, q);
        rem = __builtin_elementwise_fma(nd, q, num[k]);
        pz[k] = __builtin_elementwise_fma(rem, r, q);
    }
    return guard;
}

// ... and with the compiler's correctly rounded sqrtf and division (every operand range)
template <int R>
__device__ __forceinline__ void tv_p_update_ieee(const tv_f2 (&g)[R], tv_f2 (&pz)[R], float (&nrm)[R], float tau_over_w) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
        float n = sqrtf(g[k].x * g[k].x + g[k].y * g[k].y);
        nrm[k] = n;
        n = n * tau_over_w;
        n = n + 1.f;
        pz[k].x = (pz[k].x - 0.25f * g[k].x) / n;
        pz[k].y = (pz[k].y - 0.25f * g[k].y) / n;
    }
}

// ---- whole-plane variant for M, N <= 128: ONE launch for all n_iter_max iterations, one 1024-thread workgroup per
// channel.  Thread (column, strip) keeps its R = ceil(M/8) rows of image, dual field and `out` in registers; row
// neighbours inside a strip are register neighbours, column neighbours are wave neighbours (DPP shuffles); only the
// strip and wave seams go through LDS.  The stop test is evaluated by every thread from the same 16 wave partials in
// the same order.  Same float32 operations per pixel as tv_iter_kernel, so `out` is bit-identical; the fp64 energy
// sums associate differently (per wave instead of per tile) before they are rounded to float32.
// Templated on the column count (128, or 64 for narrow planes: 16 strips) and on the register rows per thread.
constexpr int TVP_THREADS = 1024, TVP_MAX = 128;

// DUAL: the ADMM dual update rides in the epilogue (sci_ops.hip pm_dual_update_kernel, same expressions): theta =
// clip(out, 0, 1), b +-= x - theta, and the channel's sum of (orig - report)^2 goes to sse_part[channel]; entries
// [C, nfill) are zeroed so that a caller summing the nfill partials of the stand-alone kernel's grid gets the same total.
struct TvDual {
    float* b;               // the same memory as the kernel's (read-only) `b` input, written in the epilogue
    const float* orig;
    double* sse_part;
    int which, nfill;
    float sign;
};

// V2 (round 5): the same iteration with the band kernel's instruction diet -- lane neighbours by DPP wave shifts, the wave seams
// as one broadcast LDS read per phase, the dual update through tv_p_update_fast in chunks of four rows (general-path fallback per
// chunk), the wave's energy sums by DPP row shifts + readlane instead of 24 ds_bpermute -- bit-identical `out`; the energy
// sums associate differently inside a wave (fp64, far below the float32 the stop test rounds them to).  SCIPNP_TV_PLANE_V1=1
// launches the round-2 form (A/B).
template <int TVP_COLS, int TVP_R, bool DUAL, bool V2>
__global__ void __launch_bounds__(TVP_THREADS)
tv_plane_kernel(const float* x, const float* b, float coef, float* theta, int M,
                int N, int n_iter, double weight, float tau_over_w, double eps, int32_t* __restrict__ stop_iter, TvDual dual) {
    constexpr int TVP_STRIPS = TVP_THREADS / TVP_COLS;
    __shared__ float s_p0e[TVP_STRIPS + 1][TVP_COLS];   // [s+1]: p0 on the last row of strip s
    __shared__ float s_oe[TVP_STRIPS + 1][TVP_COLS];    // [s]:   out on the first row of strip s
    __shared__ __attribute__((aligned(16))) float s_p1e[TVP_STRIPS][TVP_R];          // p1 of column 63 (read by column 64)
    __shared__ __attribute__((aligned(16))) float s_oce[TVP_STRIPS][TVP_R];          // out of column 64 (read by column 63)
    __shared__ double s_red[2][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = V2 ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int col = V2 ? (wave % (TVP_COLS / 64)) * 64 + lane : (tid & (TVP_COLS - 1));
    const int strip = V2 ? wave / (TVP_COLS / 64) : tid / TVP_COLS;
    const int R = (M + TVP_STRIPS - 1) / TVP_STRIPS;
    const int r0 = strip * R;
    const size_t chan = (size_t)blockIdx.x * M * N;
    const float* xc = x + chan;
    const float* bc = b ? b + chan : nullptr;

    float v[TVP_R], p0[TVP_R], p1[TVP_R], out[TVP_R];
    bool ok[TVP_R];
#pragma unroll
    for (int k = 0; k < TVP_R; ++k) {
        const int r = r0 + k;
        ok[k] = col < N && k < R && r < M;
        v[k] = ok[k] ? tv_input(xc, bc, coef, (size_t)r * N + col) : 0.f;
        p0[k] = 0.f;
        p1[k] = 0.f;
        out[k] = v[k];
    }

    double E0 = 0.0, Eprev = 0.0;
    int stop_at = n_iter - 1;
    for (int it = 0; it < n_iter; ++it) {
        double a1 = 0.0, a2 = 0.0;
        if (V2) {
            if (it > 0) {           // the seams of p were published before the barrier that closed iteration it-1
                const float up0 = s_p0e[strip][col];
#pragma unroll
                for (int k = 0; k < TVP_R; ++k) {
                    float sl[4];                                   // (one 16-byte broadcast read per four rows)
                    if (k % 4 == 0) *(float4*)&sl[0] = *(const float4*)&s_p1e[strip][k];
                    float left = tv_lane_prev(p1[k]);
                    if (lane == 0) left = sl[k % 4];
                    const float up = k > 0 ? p0[k > 0 ? k - 1 : 0] : up0;
                    float d = -(p0[k] + p1[k]);
                    if (k > 0 || r0 > 0) d = d + up;
                    if (col > 0) d = d + left;
                    out[k] = v[k] + d;
                    a1 += (double)(ok[k] ? d * d : 0.f);
                }
                if (it == n_iter - 1) break;      // only `out` of the last iteration is used
            }
            s_oe[strip][col] = out[0];
            if (col == 64) {
#pragma unroll
                for (int k = 0; k < TVP_R; k += 4) *(float4*)&s_oce[strip][k] = float4{out[k], out[k + 1], out[k + 2], out[k + 3]};
            }
            __syncthreads();
            const float downR = s_oe[strip + 1][col];
            const bool gcol = col < N - 1;
            constexpr int CH = TVP_R >= 16 ? 2 : 4;          // rows per pass of the dual update (16 rows per thread: 128 VGPRs)
            float sr[4];
#pragma unroll
            for (int c0 = 0; c0 < TVP_R; c0 += CH) {
                tv_f2 g[CH], pn[CH];
                float nrm[CH];
                if (c0 % 4 == 0) *(float4*)&sr[0] = *(const float4*)&s_oce[strip][c0];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int k = c0 + j;
                    float right = tv_lane_next(out[k]);
                    if (lane == 63) right = sr[k % 4];
                    float down = downR;
                    if (k + 1 < TVP_R && k + 1 < R) down = out[k + 1 < TVP_R ? k + 1 : k];
                    g[j].x = (r0 + k < M - 1) ? (down - out[k]) : 0.f;
                    g[j].y = gcol ? (right - out[k]) : 0.f;
                    pn[j] = tv_f2{p0[k], p1[k]};
                }
                const unsigned guard = tv_p_update_fast<CH>(g, pn, nrm, tau_over_w);
                if (__any(guard < TVB_GUARD)) {       // an operand in (0, 2^-60): the general sqrtf and division (wave-uniform)
#pragma unroll
                    for (int j = 0; j < CH; ++j) pn[j] = tv_f2{p0[c0 + j], p1[c0 + j]};
                    tv_p_update_ieee<CH>(g, pn, nrm, tau_over_w);
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int k = c0 + j;
                    p0[k] = pn[j].x;
                    p1[k] = pn[j].y;
                    a2 += (double)(ok[k] ? nrm[j] : 0.f);
                }
            }
#pragma unroll
            for (int k = 0; k < TVP_R; ++k)
                if (k == R - 1) s_p0e[strip + 1][col] = p0[k];
            if (col == 63) {
#pragma unroll
                for (int k = 0; k < TVP_R; k += 4) *(float4*)&s_p1e[strip][k] = float4{p1[k], p1[k + 1], p1[k + 2], p1[k + 3]};
            }
            a1 = wave_sum_f64(a1);
            a2 = wave_sum_f64(a2);
        } else {
        if (it > 0) {           // the seams of p were published before the barrier that closed iteration it-1
#pragma unroll
            for (int k = 0; k < TVP_R; ++k) {
                float left = __shfl_up(p1[k], 1, 64);
                if (lane == 0) left = s_p1e[strip][k];
                float up = s_p0e[strip][col];
                if (k > 0) up = p0[k > 0 ? k - 1 : 0];
                float d = -(p0[k] + p1[k]);
                if (r0 + k > 0) d = d + up;
                if (col > 0) d = d + left;
                out[k] = v[k] + d;
                if (ok[k]) a1 += (double)(d * d);
            }
            if (it == n_iter - 1) break;      // only `out` of the last iteration is used
        }
        s_oe[strip][col] = out[0];
        if (col == 64) {
#pragma unroll
            for (int k = 0; k < TVP_R; ++k) s_oce[strip][k] = out[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TVP_R; ++k) {
            float right = __shfl_down(out[k], 1, 64);
            if (lane == 63) right = s_oce[strip][k];
            float down = s_oe[strip + 1][col];
            if (k + 1 < TVP_R && k + 1 < R) down = out[k + 1 < TVP_R ? k + 1 : k];
            const float g0 = (r0 + k < M - 1) ? (down - out[k]) : 0.f;
            const float g1 = (col < N - 1) ? (right - out[k]) : 0.f;
            float nrm = sqrtf(g0 * g0 + g1 * g1);
            if (ok[k]) a2 += (double)nrm;
            nrm = nrm * tau_over_w;
            nrm = nrm + 1.f;
            p0[k] = (p0[k] - 0.25f * g0) / nrm;
            p1[k] = (p1[k] - 0.25f * g1) / nrm;
            if (k == R - 1) s_p0e[strip + 1][col] = p0[k];
            if (col == 63) s_p1e[strip][k] = p1[k];
        }
        for (int off = 32; off > 0; off >>= 1) {
            a1 += __shfl_down(a1, off, 64);
            a2 += __shfl_down(a2, off, 64);
        }
        }
        if (lane == 0) { s_red[0][wave] = a1; s_red[1][wave] = a2; }
        __syncthreads();
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s1 += s_red[0][i]; s2 += s_red[1][i]; }
        double E = (double)(float)s1;
        E += weight * (double)(float)s2;
        E /= (double)((size_t)M * N);
        if (it == 0) {
            E0 = E;
            Eprev = E;
        } else if (fabs(Eprev - E) < eps * E0) {
            stop_at = it;
            break;
        } else {
            Eprev = E;
        }
    }
    if (stop_iter && tid == 0) stop_iter[blockIdx.x] = stop_at;
    if (!DUAL) {
#pragma unroll
        for (int k = 0; k < TVP_R; ++k)
            if (ok[k]) theta[chan + (size_t)(r0 + k) * N + col] = out[k];
        return;
    }
    double acc = 0.0;
    // all loads first: b is read and written through the same pointer, so a load placed after a store would wait for it
    float xv[TVP_R], bo[TVP_R], og[TVP_R];
#pragma unroll
    for (int k = 0; k < TVP_R; ++k) {
        const size_t o = chan + (size_t)(r0 + k) * N + col;
        xv[k] = ok[k] ? x[o] : 0.f;
        bo[k] = ok[k] ? dual.b[o] : 0.f;
        og[k] = (ok[k] && dual.sse_part) ? dual.orig[o] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < TVP_R; ++k) {
        if (ok[k]) {
            const size_t o = chan + (size_t)(r0 + k) * N + col;
            const float th = fminf(fmaxf(out[k], 0.f), 1.f);
            const float d = xv[k] - th;
            dual.b[o] = (dual.sign > 0.f) ? (bo[k] + d) : (bo[k] - d);
            theta[o] = th;
            if (dual.sse_part) {
                const float e = og[k] - (dual.which == 0 ? th : xv[k]);
                acc += (double)(e * e);
            }
        }
    }
    if (dual.sse_part) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        __syncthreads();                      // (the stop test's readers of s_red are done)
        if (lane == 0) s_red[0][wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int i = 0; i < 16; ++i) t += s_red[0][i];
            dual.sse_part[blockIdx.x] = t;
        }
        if (blockIdx.x == 0)
            for (int i = (int)gridDim.x + tid; i < dual.nfill; i += TVP_THREADS) dual.sse_part[i] = 0.0;
    }
}

// ---- banded variant (round 2): ONE launch for all iterations with MANY workgroups per channel.  A workgroup owns a band
// of RB rows of one channel (all columns, N <= 256) and computes it together with a halo of TVB_HALO rows on either side:
// five Chambolle iterations have a dependency cone of 4 rows up (p at r-1) and 4 rows down (out at r+1), so the band's own
// rows come out exactly as the whole-plane computation gives them -- same float32 operations per pixel, bit-identical
// `out` -- while rows of the halo go stale one per iteration from the artificial boundary inwards and are never used.
// The stop test needs the channel's global energy of every iteration: kernel A runs all iterations unconditionally, leaves
// the band's own-row partial sums of every iteration in global memory and stores the `out` of the LAST iteration
// (speculating that the channel does not stop early: with eps = 2e-4 and 5 iterations it almost never does); kernel B
// evaluates skimage's stop test per channel from the partials of all its bands and, for a channel that did stop at
// iteration i* < n-1, recomputes its bands up to i* and overwrites theta.  Geometry: thread = (column, strip of R rows);
// row neighbours inside a strip are register neighbours, column neighbours lane neighbours, strip / wave seams go through LDS.
constexpr int TVB_HALO = 4;

template <int COLS, int R, int STRIPS>
__device__ __forceinline__ void tv_band_run(const float* xc, const float* bc, float coef, float* th, int M, int N,
                                            int n_iter, int a_lo, int a_hi, int ext_lo, float tau_over_w, double* part_out,
                                            float* cand = nullptr, size_t cand_stride = 0) {
    constexpr int WPS = COLS / 64;                       // waves per strip
    __shared__ float s_p0e[STRIPS + 1][COLS];            // [s+1]: p0 on the last row of strip s
    __shared__ float s_oe[STRIPS + 1][COLS];             // [s]:   out on the first row of strip s
    __shared__ float s_p1e[STRIPS][WPS][R];              // p1 of a wave's last column (read by the next wave's first)
    __shared__ float s_oce[STRIPS][WPS][R];              // out of a wave's first column (read by the previous wave's last)
    __shared__ double s_red[2][COLS * STRIPS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = tid & (COLS - 1), strip = tid / COLS, wcol = col >> 6;
    const int r0 = ext_lo + strip * R;

    float v[R], p0[R], p1[R], out[R];
    bool ok[R], own[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = r0 + k;
        ok[k] = col < N && r < M;
        own[k] = ok[k] && r >= a_lo && r < a_hi;
        v[k] = ok[k] ? tv_input(xc, bc, coef, (size_t)r * N + col) : 0.f;
        p0[k] = 0.f;
        p1[k] = 0.f;
        out[k] = v[k];
    }
    if (strip == 0) { s_p0e[0][col] = 0.f; s_oe[STRIPS][col] = 0.f; }      // artificial band edges: defined (unused) values

    for (int it = 0; it < n_iter; ++it) {
        double a1 = 0.0, a2 = 0.0;
        if (it > 0) {           // the seams of p were published before the barrier that closed iteration it-1
#pragma unroll
            for (int k = 0; k < R; ++k) {
                float left = __shfl_up(p1[k], 1, 64);
                if (lane == 0) left = s_p1e[strip][wcol > 0 ? wcol - 1 : 0][k];
                float up = s_p0e[strip][col];
                if (k > 0) up = p0[k > 0 ? k - 1 : 0];
                float d = -(p0[k] + p1[k]);
                if (r0 + k > 0) d = d + up;
                if (col > 0) d = d + left;
                out[k] = v[k] + d;
                if (own[k]) a1 += (double)(d * d);
            }
            if (cand) {         // candidate form: the band's own rows of EVERY iteration's `out` (the stop test picks one later)
#pragma unroll
                for (int k = 0; k < R; ++k)
                    if (own[k]) cand[(size_t)(it - 1) * cand_stride + (size_t)(r0 + k) * N + col] = out[k];
            }
        }
        if (it == n_iter - 1) break;          // only `out` of the last iteration is used (its energy decides nothing)
        s_oe[strip][col] = out[0];
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < R; ++k) s_oce[strip][wcol][k] = out[k];
        }
        __syncthreads();                      // (also orders thread 0's read of s_red below before the next writes)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            float right = __shfl_down(out[k], 1, 64);
            if (lane == 63) right = s_oce[strip][wcol + 1 < WPS ? wcol + 1 : wcol][k];
            float down = s_oe[strip + 1][col];
            if (k + 1 < R) down = out[k + 1 < R ? k + 1 : k];
            const float g0 = (r0 + k < M - 1) ? (down - out[k]) : 0.f;
            const float g1 = (col < N - 1) ? (right - out[k]) : 0.f;
            float nrm = sqrtf(g0 * g0 + g1 * g1);
            if (own[k]) a2 += (double)nrm;
            nrm = nrm * tau_over_w;
            nrm = nrm + 1.f;
            p0[k] = (p0[k] - 0.25f * g0) / nrm;
            p1[k] = (p1[k] - 0.25f * g1) / nrm;
            if (k == R - 1) s_p0e[strip + 1][col] = p0[k];
            if (lane == 63) s_p1e[strip][wcol][k] = p1[k];
        }
        if (part_out) {         // kernel A: the band's own-row partial energy sums of this iteration
            for (int off = 32; off > 0; off >>= 1) {
                a1 += __shfl_down(a1, off, 64);
                a2 += __shfl_down(a2, off, 64);
            }
            if (lane == 0) { s_red[0][wave] = a1; s_red[1][wave] = a2; }
        }
        __syncthreads();                      // publishes the seams of p (and the wave partials)
        if (part_out && tid == 0) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < COLS * STRIPS / 64; ++i) { s1 += s_red[0][i]; s2 += s_red[1][i]; }
            part_out[2 * it] = s1;
            part_out[2 * it + 1] = s2;
        }
    }
    if (th) {
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (own[k]) th[(size_t)(r0 + k) * N + col] = out[k];
    }
}

// ---- round 5: the same band computation rearranged (the kernel above measured 3.0 us per Chambolle iteration on the ADMM-TV
// shape -- 32 planes of 128 x 128, 6 pixels per thread, two waves per SIMD -- against 1.7 us of instruction issue for its 75
// vector instructions per pixel: profiles/r05x_tv_band_sweep.txt).  Identical results -- `out`, candidates and partial sums bit
// for bit -- from fewer and better ordered instructions:
//   * the energy sums of an iteration stay in per-thread fp64 registers and are reduced ONCE after the loop (the per-iteration
//     wave reduction was a chain of 12 dependent cross-lane permutes + thread 0 adding the wave partials between two barriers);
//   * lane neighbours come from DPP wave shifts (one VALU move) instead of ds_bpermute round trips, the wave-seam values from
//     one unconditional broadcast LDS read per phase instead of R branches; row conditions are wave-uniform (scalar);
//   * sqrt and the two divisions by the same denominator are written out: the compiler's correctly rounded expansions are 17 +
//     2 x 11 instructions, most of them range handling (operands below 2^-96, infinities) -- here v_sqrt + the +-1 ulp fix-up (9)
//     and ONE reciprocal refinement shared by both quotients, whose Newton / residual steps run as packed fp32 pairs (8): the
//     same instruction sequence the compiler emits minus the scaling, hence the same bits whenever no scaling would have
//     happened.  A wave checks exactly that for every operand (gradient energy and both numerators: zero or >= 2^-60) and
//     redoes the phase with the compiler's sqrtf and '/' otherwise (tests/test_gpu_ops.py drives both paths; the one case
//     the fast path does not reproduce is the SIGN of a zero quotient for a -0 numerator, which only an earlier underflow
//     can produce and which reaches `out` only through an input pixel that is itself -0);
//   * the pixels of a thread go through each stage together (gradients, sqrt, reciprocal, quotients), so the quarter-rate
//     v_sqrt / v_rcp results and the compare -> select hazards of one pixel are covered by the next pixel's instructions.
// clock stamps of one workgroup's first and last wave (build/variants/libscipnp_tvstamps.so, -DSCIPNP_TV_STAMPS: the candidate form
// writes them behind the first 256 bytes of its otherwise unused stop_iter argument; tools/probes/tv_band_stamps.py)
#if defined(SCIPNP_TV_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define TVB_STAMP(slot)                                                                                                 \
    do {                                                                                                                \
        if (stamps) {                                                                                                   \
            unsigned long long t_;                                                                                      \
            const unsigned long long* p_ = stamps + (slot);                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : "=&s"(t_) : "s"(p_) : "memory"); \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
    } while (0)
#define TVB_REALTIME(slot)                                                                                              \
    do {                                                                                                                \
        if (stamps) {                                                                                                   \
            unsigned long long t_;                                                                                      \
            const unsigned long long* p_ = stamps + (slot);                                                             \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_store_dwordx2 %0, %1, 0x0" : "=&s"(t_) : "s"(p_) : "memory"); \
        }                                                                                                               \
    } while (0)
#else
#define TVB_STAMP(slot) (void)0
#define TVB_REALTIME(slot) (void)0
#endif

template <int COLS, int R, int STRIPS, bool FULLW>       // FULLW: N == COLS (no column mask anywhere)
__device__ __forceinline__ void tv_band_run2(const float* xc, const float* bc, float coef, float* th, int M, int N,
                                             int n_iter, int a_lo, int a_hi, int ext_lo, float tau_over_w, double* part_out,
                                             float* cand = nullptr, size_t cand_stride = 0, unsigned long long* stamp_buf = nullptr) {
    constexpr int WPS = COLS / 64;                       // waves per strip
    constexpr int NW = COLS * STRIPS / 64;
    constexpr int RP = (R + 3) & ~3;                     // seam vectors padded to 16 bytes
    constexpr int NACC = 2 * TVB_HALO;
    static_assert(NW >= NACC, "one wave per energy sum in the final reduction");
    __shared__ float s_p0e[STRIPS + 1][COLS];            // [s+1]: p0 on the last row of strip s
    __shared__ float s_oe[STRIPS + 1][COLS];             // [s]:   out on the first row of strip s
    __shared__ __attribute__((aligned(16))) float s_p1e[STRIPS][WPS + 1][RP];   // [w+1]: p1 of wave w's last column
    __shared__ __attribute__((aligned(16))) float s_oce[STRIPS][WPS + 1][RP];   // [w]:   out of wave w's first column
    __shared__ double s_acc[NACC][COLS * STRIPS];        // every thread's energy sums (after the loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // scalar: strip and row conditions are wave-uniform
    const int strip = wave / WPS, wcol = wave - strip * WPS;
    const int col = wcol * 64 + lane;
    const int r0 = ext_lo + strip * R;
    const bool colok = FULLW || col < N, last_col = !FULLW && col >= N - 1;
    const int kM = M - 1 - r0;                           // the image's last row inside this strip (if 0 <= kM < R)
#if defined(SCIPNP_TV_STAMPS)
    unsigned long long* const stamps = (stamp_buf && (wave == 0 || wave == NW - 1)) ? stamp_buf + (wave == 0 ? 0 : 64) : nullptr;
#endif
    TVB_REALTIME(60);
    TVB_STAMP(0);

    float v[R], out[R];
    tv_f2 pz[R];
    bool rowown[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = r0 + k;
        rowown[k] = r < M && r >= a_lo && r < a_hi;
        v[k] = (colok && r < M) ? tv_input_handoff(xc, bc, coef, (size_t)r * N + col) : 0.f;
        pz[k] = tv_f2{0.f, 0.f};
        out[k] = v[k];
    }
    // The edges of the image enter as operands that leave the arithmetic unchanged instead of as per-pixel selects: the dual
    // field "above row 0" and "left of column 0" is -0 (d + -0 == d for every d, signed zeros included); the `out` "below the
    // last row" / "right of the last column" is the edge pixel's own value (x - x == +0, what the reference's zero gradient is).
    // Band edges that are not image edges get the same values: they only feed halo rows, which go stale by construction.
    if (strip == 0) { s_p0e[0][col] = -0.f; s_oe[STRIPS][col] = 0.f; }
    if (wcol == 0 && lane < RP) s_p1e[strip][0][lane] = -0.f;
    if (wcol == WPS - 1 && lane < RP) s_oce[strip][WPS][lane] = 0.f;      // (FULLW: the same wave's lane 63 overwrites it below)
#if defined(SCIPNP_TV_STAMPS)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TVB_STAMP(1);

#pragma unroll 1
    for (int it = 0; it < n_iter; ++it) {
        double a1 = 0.0, a2 = 0.0;
        TVB_STAMP(8 + 8 * it);
        if (it > 0) {           // the seams of p were published before the barrier that closed iteration it-1
            float sl[RP];
#pragma unroll
            for (int k = 0; k < RP; k += 4) *(float4*)&sl[k] = *(const float4*)&s_p1e[strip][wcol][k];
            const float up0 = s_p0e[strip][col];
            float left[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                left[k] = tv_lane_prev(pz[k].y);
                if (lane == 0) left[k] = sl[k];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const float up = k > 0 ? pz[k > 0 ? k - 1 : 0].x : up0;
                float d = -(pz[k].x + pz[k].y);
                d = d + up;
                d = d + left[k];
                out[k] = v[k] + d;
                a1 += (double)((rowown[k] && colok) ? d * d : 0.f);      // (+0.0 for halo rows and columns >= N: the sum is unchanged)
            }
            if (cand) {         // candidate form: the band's own rows of EVERY iteration's `out` (the stop test picks one later)
                float* cp = cand + (size_t)(it - 1) * cand_stride + (size_t)r0 * N + col;
#pragma unroll
                for (int k = 0; k < R; ++k)
                    if (rowown[k] && colok) tv_handoff_store(cp + (size_t)k * N, out[k]);
            }
        }
        TVB_STAMP(9 + 8 * it);
        if (it == n_iter - 1) break;          // only `out` of the last iteration is used (its energy decides nothing)
        if (kM >= 0 && kM < R - 1) {          // (one wave row of the launch) the row below the image's last row repeats it
#pragma unroll
            for (int k = 1; k < R; ++k)
                if (k == kM + 1) out[k] = out[k - 1];
        }
        s_oe[strip][col] = out[0];
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < RP; k += 4) {
                float4 q;
                q.x = out[k < R ? k : R - 1]; q.y = out[k + 1 < R ? k + 1 : R - 1];
                q.z = out[k + 2 < R ? k + 2 : R - 1]; q.w = out[k + 3 < R ? k + 3 : R - 1];
                *(float4*)&s_oce[strip][wcol][k] = q;
            }
        }
        if (FULLW && wcol == WPS - 1 && lane == 63) {      // the last column's right neighbour is the column itself
#pragma unroll
            for (int k = 0; k < RP; k += 4) {
                float4 q;
                q.x = out[k < R ? k : R - 1]; q.y = out[k + 1 < R ? k + 1 : R - 1];
                q.z = out[k + 2 < R ? k + 2 : R - 1]; q.w = out[k + 3 < R ? k + 3 : R - 1];
                *(float4*)&s_oce[strip][WPS][k] = q;
            }
        }
        __syncthreads();
        TVB_STAMP(10 + 8 * it);
        {
            float sr[RP];
#pragma unroll
            for (int k = 0; k < RP; k += 4) *(float4*)&sr[k] = *(const float4*)&s_oce[strip][wcol + 1][k];
            float downR = s_oe[strip + 1][col];
            if (kM == R - 1) downR = out[R - 1];          // (wave-uniform)
            tv_f2 g[R];
            float nrm[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                float right = tv_lane_next(out[k]);
                if (lane == 63) right = sr[k];
                const float down = k + 1 < R ? out[k + 1 < R ? k + 1 : k] : downR;
                g[k].x = down - out[k];
                g[k].y = last_col ? 0.f : (right - out[k]);
            }
            tv_f2 pn[R];
#pragma unroll
            for (int k = 0; k < R; ++k) pn[k] = pz[k];
            const unsigned guard = tv_p_update_fast<R>(g, pn, nrm, tau_over_w);
            if (__any(guard < TVB_GUARD)) {           // an operand in (0, 2^-60): the general sqrtf and division (wave-uniform)
#pragma unroll
                for (int k = 0; k < R; ++k) pn[k] = pz[k];
                tv_p_update_ieee<R>(g, pn, nrm, tau_over_w);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                pz[k] = pn[k];
                a2 += (double)((rowown[k] && colok) ? nrm[k] : 0.f);
            }
        }
        TVB_STAMP(11 + 8 * it);
        s_p0e[strip + 1][col] = pz[R - 1].x;
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < RP; k += 4) {
                float4 q;
                q.x = pz[k < R ? k : R - 1].y; q.y = pz[k + 1 < R ? k + 1 : R - 1].y;
                q.z = pz[k + 2 < R ? k + 2 : R - 1].y; q.w = pz[k + 3 < R ? k + 3 : R - 1].y;
                *(float4*)&s_p1e[strip][wcol + 1][k] = q;
            }
        }
        s_acc[2 * it][tid] = a1;              // (the thread's own words: read back after the loop; keeps the loop rolled without
        s_acc[2 * it + 1][tid] = a2;          // a register per iteration -- selecting one by `it` cost 16 v_cndmask on VCC)
        __syncthreads();                      // publishes the seams of p
        TVB_STAMP(12 + 8 * it);
    }
    TVB_STAMP(50);
    if (part_out) {
        // the band's own-row partial energy sums of every iteration.  Through LDS, one wave per sum: eight 64-lane permute trees
        // per wave (the round-2 form, 768 ds_bpermute per workgroup) measured 4100 clocks here -- the LDS pipe, not latency.
        // (fp64 sums of exact fp32 terms in a fixed order: deterministic; the association differs from the round-2 kernel's in
        // the last bits of a double, far below what the stop test -- which rounds the channel sums to float32 -- can see)
        TVB_STAMP(51);
        if (wave < 2 * (n_iter - 1)) {        // (the loop's last barrier has published every thread's sums)
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < NW; ++j) t += s_acc[wave][j * 64 + lane];
            t = wave_sum_f64(t);
            if (lane == 0) tv_handoff_store(part_out + wave, t);
        }
        TVB_STAMP(52);
    }
    if (th) {
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (rowown[k] && colok) th[(size_t)(r0 + k) * N + col] = out[k];
    }
    TVB_STAMP(53);
    TVB_REALTIME(61);
#if defined(SCIPNP_TV_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (stamps) asm volatile("s_dcache_wb" ::: "memory");      // (the scalar data cache is not written back at the end of a kernel)
#endif
}

template <int COLS, int R, int STRIPS, int V2, typename... A>      // V2: 0 round-2 schedule, 1 round 5, 2 round 5 with N == COLS
__device__ __forceinline__ void tv_band_go(A... a) {
    if (V2 == 2) tv_band_run2<COLS, R, STRIPS, true>(a...);
    else if (V2 == 1) tv_band_run2<COLS, R, STRIPS, false>(a...);
    else tv_band_run<COLS, R, STRIPS>(a...);
}

template <int COLS, int R, int STRIPS, int STAGE, int V2>
__global__ void __launch_bounds__(COLS* STRIPS)
tv_band_kernel(const float* __restrict__ x, const float* __restrict__ b, float coef, float* __restrict__ theta, int M, int N,
               int n_iter, int RB, int nbands, double weight, float tau_over_w, double eps, double* __restrict__ part,
               int32_t* __restrict__ stop_iter, size_t cand_stride) {
    const int c = blockIdx.x / nbands, band = blockIdx.x - c * nbands;
    const size_t chan = (size_t)c * M * N;
    const int a_lo = band * RB, a_hi = min(M, a_lo + RB);
    const int ext_lo = max(0, a_lo - TVB_HALO);
    const float* xc = x + chan;
    const float* bc = b ? b + chan : nullptr;
    if (STAGE == 0) {
        tv_band_go<COLS, R, STRIPS, V2>(xc, bc, coef, theta + chan, M, N, n_iter, a_lo, a_hi, ext_lo, tau_over_w,
                                     part + ((size_t)c * nbands + band) * 2 * n_iter);
        return;
    }
    if (STAGE == 3) {
        // CANDIDATE form -- ONE launch, nothing recomputed, no communication between the bands: every band stores its own rows
        // of the `out` of every iteration 1 .. n_iter-1 (cand[it-1], `theta` here; plane stride cand_stride) beside its partial
        // energy sums; whoever consumes the result evaluates the stop test per channel from the partials (tv_band_stop_test:
        // tv_stop_kernel below, or the ADMM dual-update kernels on the fly) and reads cand[stop - 1].  Early stops are common
        // in the ADMM loop (most planes of a converging reconstruction), so a form that recomputes stopped channels pays a
        // second pass almost every time; and a "last band of the channel decides" scheme needs a device-scope fence per
        // workgroup -- an L2 write-back across the XCDs -- that costs more than the launch it saves (measured: 82 us against
        // 17 us for 32 planes of 128 x 128).
#if defined(SCIPNP_TV_STAMPS)
        if (V2) {
            tv_band_run2<COLS, R, STRIPS, V2 == 2>(xc, bc, coef, nullptr, M, N, n_iter, a_lo, a_hi, ext_lo, tau_over_w,
                                          part + ((size_t)c * nbands + band) * 2 * n_iter, theta + chan, cand_stride,
                                          (stop_iter && blockIdx.x == gridDim.x / 2 + 3) ? (unsigned long long*)(stop_iter + 64) : nullptr);
            return;
        }
#endif
        tv_band_go<COLS, R, STRIPS, V2>(xc, bc, coef, nullptr, M, N, n_iter, a_lo, a_hi, ext_lo, tau_over_w,
                                     part + ((size_t)c * nbands + band) * 2 * n_iter, theta + chan, cand_stride);
        return;
    }
    // ---- stop test of the channel (every workgroup of the channel evaluates it identically, tv_band_stop_test in common.hpp)
    __shared__ int s_stop_at;
    if (threadIdx.x < 64) {
        const int stop_at = tv_band_stop_test(part + (size_t)c * nbands * 2 * n_iter, nbands, n_iter, (size_t)M * N, weight, eps);
        if (threadIdx.x == 0) {
            s_stop_at = stop_at;
            if (stop_iter && band == 0) stop_iter[c] = stop_at;
        }
    }
    __syncthreads();
    const int stop_at = s_stop_at;
    if (stop_at == n_iter - 1) return;                   // theta already holds the `out` of the last iteration
    tv_band_go<COLS, R, STRIPS, V2>(xc, bc, coef, theta + chan, M, N, stop_at + 1, a_lo, a_hi, ext_lo, tau_over_w, nullptr);
}

// band height for N <= 256: 16-row bands when 32-row bands would leave the chip short of workgroups
bool tv_band_fits(int M, int N, int n_iter) { return N <= 256 && n_iter >= 1 && n_iter - 1 <= TVB_HALO; }

// laboratory switch (lab_switch; SCIPNP_TV_TINY_BANDS=1): 8-row bands for problems that leave the chip short of workgroups even with 16-row
// bands (twice the workgroups, 16 computed rows per 8 useful ones)
static bool tv_tiny_bands() {
    static const bool on = [] { const char* e = lab_switch("SCIPNP_TV_TINY_BANDS"); return e && e[0] == '1'; }();
    return on;
}

// laboratory switch (lab_switch, -DSCIPNP_LAB_SWITCHES builds only; SCIPNP_TV_BAND_V1=1): the round-2 schedule of the band computation (tv_band_run) instead of round 5's
// (tv_band_run2) -- same results, for A/B timing
static bool tv_band_v1() {
    static const bool on = [] { const char* e = lab_switch("SCIPNP_TV_BAND_V1"); return e && e[0] == '1'; }();
    return on;
}

// band geometry of the banded kernels for (M, C): rows per band, bands per channel
static void tv_band_geometry(int M, int C, int* RB, int* nbands) {
    const bool small = (long long)C * ((M + 31) / 32) < 256;        // fewer than one workgroup per CU with 32-row bands
    const bool tiny = tv_tiny_bands() && (long long)C * ((M + 15) / 16) <= 256;   // ... at most one per CU with 16-row bands
    *RB = tiny ? 8 : small ? 16 : 32;
    *nbands = (M + *RB - 1) / *RB;
}

// candidates != nullptr: the one-launch candidate form (STAGE 3) -- `out` of iteration it goes to candidates[(it - 1) * C*M*N ..],
// the partial sums to `part`, nothing else.  Else kernel A + kernel B into theta.
static int tv_band_launch(const float* x, const float* b, float coef, float* theta, int M, int N, int C, double weight_d,
                          float tau_over_w, double eps_d, int n_iter, double* part, int32_t* stop_iter,
                          float* candidates, hipStream_t st) {
    int RB, nbands;
    tv_band_geometry(M, C, &RB, &nbands);
    const bool small = RB == 16, tiny = RB == 8;
    const bool v1 = tv_band_v1();
    const dim3 grid((unsigned)(C * nbands));
    const size_t img = (size_t)C * M * N;
#define SCIPNP_TVB_(COLS, R, STRIPS, RBV, V2)                                                                               \
    do {                                                                                                                \
        static_assert(STRIPS * R >= RBV + 2 * TVB_HALO, "a workgroup's rows must cover its band + both halos");        \
        if (candidates) {                                                                                               \
            hipLaunchKernelGGL((tv_band_kernel<COLS, R, STRIPS, 3, V2>), grid, dim3(COLS * STRIPS), 0, st, x, b, coef, candidates, M, \
                               N, n_iter, RB, nbands, weight_d, tau_over_w, eps_d, part, stop_iter, img);               \
        } else {                                                                                                        \
            hipLaunchKernelGGL((tv_band_kernel<COLS, R, STRIPS, 0, V2>), grid, dim3(COLS * STRIPS), 0, st, x, b, coef, theta, M, N, \
                               n_iter, RB, nbands, weight_d, tau_over_w, eps_d, part, stop_iter, img);                  \
            hipLaunchKernelGGL((tv_band_kernel<COLS, R, STRIPS, 1, V2>), grid, dim3(COLS * STRIPS), 0, st, x, b, coef, theta, M, N, \
                               n_iter, RB, nbands, weight_d, tau_over_w, eps_d, part, stop_iter, img);                  \
        }                                                                                                               \
    } while (0)
#define SCIPNP_TVB(COLS, R, STRIPS, RBV)                                                                                \
    do {                                                                                                                \
        if (v1) SCIPNP_TVB_(COLS, R, STRIPS, RBV, 0);                                                                   \
        else if (N == COLS) SCIPNP_TVB_(COLS, R, STRIPS, RBV, 2);                                                       \
        else SCIPNP_TVB_(COLS, R, STRIPS, RBV, 1);                                                                      \
    } while (0)
    if (N <= 64) {
        if (tiny) SCIPNP_TVB(64, 2, 8, 8); else if (small) SCIPNP_TVB(64, 3, 8, 16); else SCIPNP_TVB(64, 5, 8, 32);
    } else if (N <= 128) {
        if (tiny) SCIPNP_TVB(128, 4, 4, 8); else if (small) SCIPNP_TVB(128, 6, 4, 16); else SCIPNP_TVB(128, 10, 4, 32);
    } else {
        if (tiny) SCIPNP_TVB(256, 4, 4, 8); else if (small) SCIPNP_TVB(256, 6, 4, 16); else SCIPNP_TVB(256, 10, 4, 32);
    }
#undef SCIPNP_TVB
#undef SCIPNP_TVB_
    return launch_status("tv_band_kernel");
}

// stop test of the candidate form as its own launch: one wave per channel, stop[c] = the iteration whose `out` skimage returns
__global__ void __launch_bounds__(64)
tv_stop_kernel(const double* __restrict__ part, int32_t* __restrict__ stop, int nbands, int n_iter, size_t MN, double weight,
               double eps) {
    const int c = blockIdx.x;
    const int s = tv_band_stop_test(part + (size_t)c * nbands * 2 * n_iter, nbands, n_iter, MN, weight, eps);
    if (threadIdx.x == 0) stop[c] = s;
}

// theta[i] = candidates[(stop[channel] - 1) * C*M*N + i]: the selection the fused ADMM kernel does on the fly
__global__ void __launch_bounds__(256)
tv_select_kernel(const float* __restrict__ cand, const int32_t* __restrict__ stop_iter, float* __restrict__ theta, size_t MN,
                 size_t img) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= img) return;
    theta[i] = cand[(size_t)(stop_iter[i / MN] - 1) * img + i];
}

// the shortest decimal that round-trips a float32: weight and eps are Python floats (doubles) in the reference and
// promote its float32 sums to double; they reach this ABI as float32 (0.1f -> 0.1, 2e-4f -> 2e-4)
static double as_double(float f) {
    for (int prec = 1; prec <= 9; ++prec) {
        char buf[40];
        double d = 0.0;
        snprintf(buf, sizeof buf, "%.*g", prec, (double)f);
        sscanf(buf, "%lf", &d);
        if ((float)d == f) return d;
    }
    return (double)f;
}

static int tv_plane_launch(const float* x, const float* b, float coef, float* theta, int M, int N, int C, double weight_d,
                           float tau_over_w, double eps_d, int n_iter_max, int32_t* stop_iter, const TvDual* dual,
                           hipStream_t st) {
    // rows per thread: ceil(M / strips), strips = 1024 / columns
    const TvDual d = dual ? *dual : TvDual{};
    static const bool plane_v1 = [] { const char* e = lab_switch("SCIPNP_TV_PLANE_V1"); return e && e[0] == '1'; }();
#define SCIPNP_TVP_(COLS, R, V2)                                                                                        \
    do {                                                                                                                \
        if (dual)                                                                                                       \
            hipLaunchKernelGGL((tv_plane_kernel<COLS, R, true, V2>), dim3(C), dim3(TVP_THREADS), 0, st, x, b, coef, theta, M, \
                               N, n_iter_max, weight_d, tau_over_w, eps_d, stop_iter, d);                                \
        else                                                                                                            \
            hipLaunchKernelGGL((tv_plane_kernel<COLS, R, false, V2>), dim3(C), dim3(TVP_THREADS), 0, st, x, b, coef, theta, M, \
                               N, n_iter_max, weight_d, tau_over_w, eps_d, stop_iter, d);                                \
    } while (0)
#define SCIPNP_TVP(COLS, R)                                                                                             \
    do { if (plane_v1) SCIPNP_TVP_(COLS, R, false); else SCIPNP_TVP_(COLS, R, true); } while (0)
    if (N <= 64) {
        if (M <= 64) SCIPNP_TVP(64, 4); else SCIPNP_TVP(64, 8);
    } else {
        if (M <= 64) SCIPNP_TVP(128, 8); else SCIPNP_TVP(128, 16);
    }
#undef SCIPNP_TVP
#undef SCIPNP_TVP_
    return launch_status("tv_plane_kernel");
}

// TV step + dual update of one ADMM-TV iteration in ONE launch (iterate.hip); false if the planes do not fit the
// whole-plane kernel or the caller's sse_part (sized for pm_dual_update's grid, `nfill` entries) has fewer than C entries
bool tv_plane_dual_fits(int M, int N, int C, int nfill, bool want_sse) {
    return M <= TVP_MAX && N <= TVP_MAX && (!want_sse || C <= nfill);
}

int tv_plane_dual(const float* x, float* b, float coef, float* theta, int M, int N, int C, float weight, float eps,
                  int n_iter_max, const float* orig, double* sse_part, int which, float sign, int nfill, hipStream_t st) {
    const double weight_d = as_double(weight);
    TvDual d;
    d.b = b; d.orig = orig; d.sse_part = sse_part; d.which = which; d.nfill = nfill; d.sign = sign;
    return tv_plane_launch(x, b, coef, theta, M, N, C, weight_d, (float)(0.25 / weight_d), as_double(eps), n_iter_max,
                           nullptr, &d, st);
}

bool tv_candidates_fit(int M, int N, int n_iter) { return tv_band_fits(M, N, n_iter) && n_iter >= 2; }

// where the candidate form keeps its results in a workspace laid out for (M, N, C, n_iter): `out` of iteration it at
// cand[(it - 1) * C*M*N ..] (the tiled kernel's dual-field slots), the partial sums of channel c at part[c * nbands * 2 * n_iter ..],
// a stop-iteration array of C ints for tv_stop_test_launch
void tv_candidate_ptrs(int M, int N, int C, int n_iter, void* workspace, TvCandidates* out) {
    TvWorkspace ws;
    tv_layout(M, N, C, n_iter, workspace, &ws);
    int RB;
    tv_band_geometry(M, C, &RB, &out->nbands);
    out->cand = ws.p[0];
    out->part = ws.partial;
    out->stop = (int32_t*)ws.stopped;
    out->n_iter = n_iter;
    out->MN = (size_t)M * N;
}

int tv_band_candidates(const float* x, const float* b, float coef, int M, int N, int C, float weight, float eps, int n_iter,
                       void* workspace, size_t workspace_bytes, hipStream_t st) {
    SCIPNP_REQUIRE(x && workspace && tv_candidates_fit(M, N, n_iter) && C > 0 && C <= 65535, "bad arguments");
    SCIPNP_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "workspace must be 256-byte aligned");
    TvWorkspace ws;
    const size_t need = tv_layout(M, N, C, n_iter, workspace, &ws);
    if (workspace_bytes < need) return fail(SCIPNP_EWORKSPACE, "TV workspace too small: %zu < %zu", workspace_bytes, need);
    const double weight_d = as_double(weight);
    return tv_band_launch(x, b, coef, nullptr, M, N, C, weight_d, (float)(0.25 / weight_d), as_double(eps), n_iter, ws.partial,
                          nullptr, ws.p[0], st);
}

// the stop test of a finished candidate launch as its own launch (the flush of a deferred ADMM-TV iteration): fills cd.stop
int tv_stop_test_launch(const TvCandidates& cd, int C, float weight, float eps, hipStream_t st) {
    hipLaunchKernelGGL(tv_stop_kernel, dim3(C), dim3(64), 0, st, cd.part, cd.stop, cd.nbands, cd.n_iter, cd.MN, as_double(weight),
                       as_double(eps));
    return launch_status("tv_stop_kernel");
}

double tv_scalar_as_double(float f) { return as_double(f); }

__global__ void tv_fill_stop_kernel(int32_t* stop_iter, int C, int last) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C && stop_iter[c] < 0) stop_iter[c] = last;
}

}  // namespace scipnp

using namespace scipnp;

extern "C" {

size_t scipnp_tv_workspace_bytes(int M, int N, int C, int n_iter_max) {
    if (M <= 0 || N <= 0 || C <= 0 || n_iter_max <= 0) return 0;
    return tv_layout(M, N, C, n_iter_max, nullptr, nullptr);
}

int scipnp_tv_chambolle(const float* x, const float* b, float coef, float* theta, int M, int N, int C,
                        float weight, float eps, int n_iter_max, void* workspace, size_t workspace_bytes,
                        int32_t* stop_iter, scipnp_stream_t s) {
    return scipnp_tv_chambolle_ex(x, b, coef, theta, M, N, C, weight, eps, n_iter_max, workspace, workspace_bytes, stop_iter,
                                  0, s);
}

int scipnp_tv_chambolle_ex(const float* x, const float* b, float coef, float* theta, int M, int N, int C,
                           float weight, float eps, int n_iter_max, void* workspace, size_t workspace_bytes,
                           int32_t* stop_iter, int kernel, scipnp_stream_t s) {
    SCIPNP_REQUIRE(x && theta && workspace, "null pointer");
    SCIPNP_REQUIRE(M > 0 && N > 0 && C > 0 && C <= 65535 && n_iter_max > 0, "bad shape M=%d N=%d C=%d n_iter_max=%d", M, N, C, n_iter_max);
    SCIPNP_REQUIRE(weight > 0.f, "weight must be positive");
    SCIPNP_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "workspace must be 256-byte aligned");
    TvWorkspace ws;
    const size_t need = tv_layout(M, N, C, n_iter_max, workspace, &ws);
    if (workspace_bytes < need) return fail(SCIPNP_EWORKSPACE, "TV workspace too small: %zu < %zu", workspace_bytes, need);
    hipStream_t st = (hipStream_t)s;
    const dim3 block(TV_TSX, TV_TY);
    const dim3 grid((N + TV_TSX - 1) / TV_TSX, (M + TV_TSY - 1) / TV_TSY, C);
    const double eps_d = as_double(eps);
    const double weight_d = as_double(weight);
    const float tau_over_w = (float)(0.25 / weight_d);
    const bool fits_plane = M <= TVP_MAX && N <= TVP_MAX;
    const bool fits_band = tv_band_fits(M, N, n_iter_max);
    SCIPNP_REQUIRE(kernel >= 0 && kernel <= 4 && (kernel != 2 || fits_plane) && ((kernel != 3 && kernel != 4) || fits_band),
                   "kernel=%d not available for %d x %d planes, %d iterations", kernel, M, N, n_iter_max);
    SCIPNP_REQUIRE(kernel != 4 || (n_iter_max >= 2 && stop_iter), "kernel 4 (candidate form) needs n_iter_max >= 2 and stop_iter");
    if (kernel == 3 || (kernel == 0 && fits_band)) {
        // banded form, kernel A + kernel B; the partials live in the tiled kernel's slot of the workspace
        // (n_iter x C x ceil(M/16) x 2 doubles: at least as many as C x nbands x 2 x n_iter)
        return tv_band_launch(x, b, coef, theta, M, N, C, weight_d, tau_over_w, eps_d, n_iter_max, ws.partial, stop_iter,
                              nullptr, st);
    }
    if (kernel == 4) {
        // candidate form (what scipnp_admm_tv_iterate uses with a deferred dual update) + the selection as its own launch:
        // the candidates live in the tiled kernel's dual-field slots (4 x C x M x N floats >= n_iter_max - 1 planes sets)
        int rc = tv_band_launch(x, b, coef, nullptr, M, N, C, weight_d, tau_over_w, eps_d, n_iter_max, ws.partial, stop_iter,
                                ws.p[0], st);
        if (rc) return rc;
        const size_t img = (size_t)C * M * N;
        int RB, nbands;
        tv_band_geometry(M, C, &RB, &nbands);
        hipLaunchKernelGGL(tv_stop_kernel, dim3(C), dim3(64), 0, st, ws.partial, stop_iter, nbands, n_iter_max, (size_t)M * N,
                           weight_d, eps_d);
        hipLaunchKernelGGL(tv_select_kernel, dim3((unsigned)((img + 255) / 256)), dim3(256), 0, st, ws.p[0], stop_iter, theta,
                           (size_t)M * N, img);
        return launch_status("tv_select_kernel");
    }
    if (fits_plane && kernel != 1) {
        // rows per thread: ceil(M / strips), strips = 1024 / columns
        return tv_plane_launch(x, b, coef, theta, M, N, C, weight_d, tau_over_w, eps_d, n_iter_max, stop_iter, nullptr, st);
    }
    for (int it = 0; it < n_iter_max; ++it) {
        if (it == 0)
            hipLaunchKernelGGL(tv_iter_kernel<true>, grid, block, 0, st, x, b, coef, theta, ws, it, M, N, C, weight_d,
                               tau_over_w, eps_d, stop_iter);
        else
            hipLaunchKernelGGL(tv_iter_kernel<false>, grid, block, 0, st, x, b, coef, theta, ws, it, M, N, C, weight_d,
                               tau_over_w, eps_d, stop_iter);
    }
    if (stop_iter)
        hipLaunchKernelGGL(tv_fill_stop_kernel, dim3((C + 63) / 64), dim3(64), 0, st, stop_iter, C, n_iter_max - 1);
    return launch_status("tv_iter_kernel");
}

}  // extern "C"
