/* One complete reconstruction through the iteration-level C ABI (include/scipnp.h), as a function so that several host
 * threads can run one each: ADMM-TV warm start + two-stage PnP-ADMM with the FFDNet-colour denoiser, split-fp16 kernels
 * (default) or fp32 Winograd.  Everything the solve touches is its own: HIP stream, device buffers, range-guard word and --
 * optionally -- the side stream and events of the two-stream network pass.  Used by pnp_admm_ffdnet_host.c and
 * two_solves_host.c. */
#ifndef PNP_SOLVE_H
#define PNP_SOLVE_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scipnp.h"

#define HIPCHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); exit(2); } } while (0)
#define SCICHK(e) do { int r_ = (e); if (r_ != SCIPNP_OK) { fprintf(stderr, "scipnp %s:%d (%d): %s\n", __FILE__, __LINE__, r_, scipnp_last_error()); exit(3); } } while (0)

static void* dmalloc(size_t bytes) { void* p; HIPCHK(hipMalloc(&p, bytes)); HIPCHK(hipMemset(p, 0, bytes)); return p; }
static void rd(void* dst, size_t bytes, FILE* f) { if (fread(dst, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(4); } }

typedef struct {
    const char* problem;      /* problem.bin (format: pnp_admm_ffdnet_host.c) */
    const char* out;          /* out_mosaic.bin */
    int f32;                  /* fp32 Winograd kernels instead of split-fp16 */
    int two_streams;          /* split-fp16: half of the frames of every network pass on the solve's own side stream */
    float input_scale;        /* y is multiplied by this (1 = as given; 2e4 drives the first layers' activations past fp16's 65504) */
    int overflow;             /* result: the solve's own range-guard word after the last iteration */
    int H, W, B, tv_iters, iters;
} pnp_solve_t;

static void pnp_solve(pnp_solve_t* job) {
    const int f32 = job->f32;
    FILE* f = fopen(job->problem, "rb");
    if (!f) { perror(job->problem); exit(1); }
    int32_t hdr[6];
    float sigma;
    rd(hdr, sizeof hdr, f); rd(&sigma, 4, f);
    const int H = hdr[0], W = hdr[1], B = hdr[2], nb = hdr[3], tv_iters = hdr[4], iters = hdr[5];
    job->H = H; job->W = W; job->B = B; job->tv_iters = tv_iters; job->iters = iters;
    const int M = H / 2, N = W / 2, nc = 96;
    const size_t HW = (size_t)H * W, E = HW * B, RGB = E * 3;
    float* y_h = malloc(HW * 4); float* Phi_h = malloc(E * 4);
    rd(y_h, HW * 4, f); rd(Phi_h, E * 4, f);
    if (job->input_scale != 1.0f)
        for (size_t i = 0; i < HW; ++i) y_h[i] *= job->input_scale;

    hipStream_t st; HIPCHK(hipStreamCreate(&st));
    /* the solve's own range-guard word: every split-fp16 launch of THIS thread raises it, nobody else's */
    int* ovf_word = dmalloc(sizeof(int));
    SCICHK(scipnp_bind_overflow_word(ovf_word));
    float *mosaic = dmalloc(E * 4), *ymos = dmalloc(HW * 4);
    float *Phi = dmalloc(E * 4), *y = dmalloc(HW * 4), *Phisum = dmalloc(HW * 4);
    float *theta = dmalloc(E * 4), *b = dmalloc(E * 4), *x = dmalloc(E * 4), *theta_raw = dmalloc(E * 4);
    HIPCHK(hipMemcpy(mosaic, Phi_h, E * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ymos, y_h, HW * 4, hipMemcpyHostToDevice));
    SCICHK(scipnp_mosaic_to_state(mosaic, Phi, M, N, B, st));
    SCICHK(scipnp_y_to_meas(ymos, y, M, N, st));
    SCICHK(scipnp_pm_setup(Phi, y, Phisum, theta, M, N, B, st));                 /* Phi Phi^T and the start point Phi^T y */

    /* ---- ADMM-TV warm start (one-stage solver: lambda = 1, gamma = 0.01, TV weight 0.1, 5 Chambolle iterations) */
    scipnp_admm_tv_args tv;
    memset(&tv, 0, sizeof tv);
    tv.struct_size = sizeof tv;
    tv.M = M; tv.N = N; tv.B = B; tv.two_stage = 0;
    tv.theta = theta; tv.b = b; tv.x = x; tv.theta_raw = theta_raw; tv.Phi = Phi; tv.y = y; tv.Phisum = Phisum;
    tv.c0 = 1.0; tv.c1 = 0.01; tv.tv_weight = 0.1f; tv.tv_iters = 5;
    tv.tv_workspace_bytes = scipnp_tv_workspace_bytes(M, N, 4 * B, 5);
    tv.tv_workspace = dmalloc(tv.tv_workspace_bytes);
    for (int k = 0; k < tv_iters; ++k) SCICHK(scipnp_admm_tv_iterate(&tv, NULL, st));

    /* ---- FFDNet weights: pack on the host, upload (fp32 form: the Winograd-domain weights are derived on the device) */
    const void** packed = malloc(nb * sizeof(void*));
    const float** packed_w = malloc(nb * sizeof(float*));
    const float** packed_w4 = calloc(nb, sizeof(float*));      /* F(4x4,3x3) packings of the body layers (NULL: F(2x2,3x3)) */
    for (int l = 0; l < nb; ++l) {
        int32_t dims[2];
        rd(dims, sizeof dims, f);
        const int co = dims[0], ci = dims[1], Cin = l == 0 ? 16 : nc, Cout = l == nb - 1 ? 16 : nc;
        float* w = malloc((size_t)co * ci * 9 * 4); float* bias = malloc((size_t)co * 4);
        rd(w, (size_t)co * ci * 9 * 4, f); rd(bias, (size_t)co * 4, f);
        if (f32) {
            const size_t nf = scipnp_conv3x3_packed_floats(Cin, Cout);
            float* ph = malloc(nf * 4);
            SCICHK(scipnp_pack_conv3x3_weights(w, bias, NULL, NULL, ci, co, Cin, Cout, ph));
            float* pd = dmalloc(nf * 4);
            HIPCHK(hipMemcpy(pd, ph, nf * 4, hipMemcpyHostToDevice));
            float* pw = dmalloc(scipnp_conv3x3_wino_packed_floats(Cin, Cout) * 4);
            SCICHK(scipnp_pack_conv3x3_wino(pd, pw, Cin, Cout, st));
            packed_w[l] = pw;
            if (Cin >= 16 && Cout >= 32) {               /* head and body layers: 2.25 multiply-adds per output instead of 4 */
                float* p4 = dmalloc(scipnp_conv3x3_wino4_packed_floats(Cin, Cout) * 4);
                SCICHK(scipnp_pack_conv3x3_wino4(pd, p4, Cin, Cout, st));
                packed_w4[l] = p4;
            }
            packed[l] = NULL;
            free(ph);
        } else {
            const size_t bytes = scipnp_conv3x3_split_packed_bytes(Cin, Cout);
            void* ph = malloc(bytes);
            SCICHK(scipnp_pack_conv3x3_split(w, bias, ci, co, Cin, Cout, ph));
            void* pd = dmalloc(bytes);
            HIPCHK(hipMemcpy(pd, ph, bytes, hipMemcpyHostToDevice));
            packed[l] = pd;
            free(ph);
        }
        free(w); free(bias);
    }
    fclose(f);

    /* ---- two-stage PnP-ADMM + FFDNet from the warm start: theta = x_tv, b = 0, w = 0 */
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpyAsync(theta, x, E * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(b, 0, E * 4, st));
    scipnp_twostage_ffdnet_args a;
    memset(&a, 0, sizeof a);
    a.struct_size = sizeof a;
    a.M = M; a.N = N; a.B = B;
    a.theta = theta; a.b = b; a.x = x; a.Phi = Phi; a.y = y; a.Phisum = Phisum;
    a.w = dmalloc(RGB * 4); a.x_rgb = dmalloc(RGB * 4); a.out_rgb = NULL;
    a.net_out_c8 = dmalloc((size_t)B * 2 * M * N * 8 * 4);
    /* conv_form names the arithmetic the pointers must provide (a mismatch is refused, never run in another form) */
    if (f32) { a.net_in_c8 = dmalloc((size_t)B * 2 * M * N * 8 * 4); a.packed_wino = packed_w; a.packed_wino4 = packed_w4; a.conv_form = SCIPNP_CONV_F32_WINO_F4; }
    else { a.net_in_c8s = dmalloc((size_t)B * 2 * 2 * M * N * 8 * 2); a.packed_split = packed; a.conv_form = SCIPNP_CONV_SPLIT_F16; }
    a.nb = nb; a.nc = nc;
    a.scratch0 = dmalloc((size_t)B * nc * M * N * 4); a.scratch1 = dmalloc((size_t)B * nc * M * N * 4);
    a.rho = 1.0; a.alpha = 1.0; a.tau = 100.0; a.sigma = sigma;
    a.overflow_word = ovf_word;                       /* (the same word the thread bound above; named per call as well) */
    hipStream_t side = NULL;
    hipEvent_t ev_fork = NULL, ev_join = NULL;
    if (job->two_streams && !f32) {                   /* the caller owns the side stream and its events, not the library */
        HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        a.side_stream = side; a.side_fork_event = ev_fork; a.side_join_event = ev_join;
    }
    for (int k = 0; k < iters; ++k) {
        a.first_iter = (k == 0);
        SCICHK(scipnp_twostage_ffdnet_iterate(&a, NULL, st));
    }
    SCICHK(scipnp_read_overflow_word(ovf_word, 1, &job->overflow, st));

    SCICHK(scipnp_state_to_mosaic(theta, mosaic, M, N, B, st));
    HIPCHK(hipStreamSynchronize(st));
    float* out_h = malloc(E * 4);
    HIPCHK(hipMemcpy(out_h, mosaic, E * 4, hipMemcpyDeviceToHost));
    FILE* g = fopen(job->out, "wb");
    if (!g || fwrite(out_h, 4, E, g) != E) { perror(job->out); exit(1); }
    fclose(g);
    SCICHK(scipnp_bind_overflow_word(NULL));
    free(out_h); free(y_h); free(Phi_h);
}
#endif
