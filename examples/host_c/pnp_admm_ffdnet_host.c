/* A complete reconstruction from a plain C host of libscipnp.so (no Python, no PyTorch): ADMM-TV warm start followed
 * by two-stage PnP-ADMM with the FFDNet-colour denoiser on the split-fp16 kernels (default) or, with a third argument
 * `f32`, in fp32 arithmetic on the Winograd kernels -- the pipeline of the reference's
 * ADMM_TV_Warm_Start_save.py + two_stage_ADMM_Online_FFD_Warm.py (without the online finetune) -- driven through the
 * iteration-level entries scipnp_admm_tv_iterate / scipnp_twostage_ffdnet_iterate of include/scipnp.h.
 *
 *   pnp_admm_ffdnet_host <problem.bin> <out_mosaic.bin> [f32]
 * problem.bin (little endian): int32 H, W, B, nb, tv_iters, iters; float32 sigma; y[H*W]; Phi[H*W*B] ((H,W,B) order);
 *   then nb layers: int32 cout, cin; float32 weight[cout*cin*9] (OIHW); float32 bias[cout].
 * out_mosaic.bin: float32 (H,W,B) reconstruction.       tests/test_gpu_cabi_host.py builds, runs and checks it
 * bit-for-bit against the Python drop-in solver. */
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

int main(int argc, char** argv) {
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s problem.bin out.bin [f32]\n", argv[0]); return 1; }
    const int f32 = argc == 4 && strcmp(argv[3], "f32") == 0;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[6];
    float sigma;
    rd(hdr, sizeof hdr, f); rd(&sigma, 4, f);
    const int H = hdr[0], W = hdr[1], B = hdr[2], nb = hdr[3], tv_iters = hdr[4], iters = hdr[5];
    const int M = H / 2, N = W / 2, nc = 96;
    const size_t HW = (size_t)H * W, E = HW * B, RGB = E * 3;
    float* y_h = malloc(HW * 4); float* Phi_h = malloc(E * 4);
    rd(y_h, HW * 4, f); rd(Phi_h, E * 4, f);

    hipStream_t st; HIPCHK(hipStreamCreate(&st));
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
    tv.M = M; tv.N = N; tv.B = B; tv.two_stage = 0;
    tv.theta = theta; tv.b = b; tv.x = x; tv.theta_raw = theta_raw; tv.Phi = Phi; tv.y = y; tv.Phisum = Phisum;
    tv.c0 = 1.0; tv.c1 = 0.01; tv.tv_weight = 0.1f; tv.tv_iters = 5;
    tv.tv_workspace_bytes = scipnp_tv_workspace_bytes(M, N, 4 * B, 5);
    tv.tv_workspace = dmalloc(tv.tv_workspace_bytes);
    for (int k = 0; k < tv_iters; ++k) SCICHK(scipnp_admm_tv_iterate(&tv, NULL, st));

    /* ---- FFDNet weights: pack on the host, upload (fp32 form: the Winograd-domain weights are derived on the device) */
    const void** packed = malloc(nb * sizeof(void*));
    const float** packed_w = malloc(nb * sizeof(float*));
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
    a.M = M; a.N = N; a.B = B;
    a.theta = theta; a.b = b; a.x = x; a.Phi = Phi; a.y = y; a.Phisum = Phisum;
    a.w = dmalloc(RGB * 4); a.x_rgb = dmalloc(RGB * 4); a.out_rgb = NULL;
    a.net_out_c8 = dmalloc((size_t)B * 2 * M * N * 8 * 4);
    if (f32) { a.net_in_c8 = dmalloc((size_t)B * 2 * M * N * 8 * 4); a.packed_wino = packed_w; }
    else { a.net_in_c8s = dmalloc((size_t)B * 2 * 2 * M * N * 8 * 2); a.packed_split = packed; }
    a.nb = nb; a.nc = nc;
    a.scratch0 = dmalloc((size_t)B * nc * M * N * 4); a.scratch1 = dmalloc((size_t)B * nc * M * N * 4);
    a.rho = 1.0; a.alpha = 1.0; a.tau = 100.0; a.sigma = sigma;
    for (int k = 0; k < iters; ++k) {
        a.first_iter = (k == 0);
        SCICHK(scipnp_twostage_ffdnet_iterate(&a, NULL, st));
    }
    int overflow = 0;
    SCICHK(scipnp_split_overflow(1, &overflow, st));
    if (overflow) { fprintf(stderr, "activations left fp16 range\n"); return 5; }

    SCICHK(scipnp_state_to_mosaic(theta, mosaic, M, N, B, st));
    HIPCHK(hipStreamSynchronize(st));
    float* out_h = malloc(E * 4);
    HIPCHK(hipMemcpy(out_h, mosaic, E * 4, hipMemcpyDeviceToHost));
    FILE* g = fopen(argv[2], "wb");
    if (!g || fwrite(out_h, 4, E, g) != E) { perror(argv[2]); return 1; }
    fclose(g);
    printf("%s: %dx%dx%d, %d ADMM-TV + %d ADMM-FFDNet iterations (%s) done\n", scipnp_version(), H, W, B, tv_iters, iters,
           f32 ? "fp32 Winograd" : "split-fp16");
    return 0;
}
