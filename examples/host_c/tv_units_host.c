/* UNIT BATCH through the plain C ABI (include/scipnp.h, "Unit batches"): U independent ADMM-TV problems of one shape are stepped
 *   (a) one after the other -- the reference's loop over measurements (ADMM_TV_Warm_Start_save.py:112-178), U calls of
 *       scipnp_admm_tv_iterate per iteration, each on its own [B][4][M][N] state -- and
 *   (b) together: ONE scipnp_admm_tv_iterate per iteration with args.units = U on the unit-batched layout [B][U][4][M][N]
 *       (y / Phisum [U][4][M][N]; set up by scipnp_pm_setup_units),
 * and every unit's reconstruction must come out BIT-IDENTICAL.  The host only permutes memory (hipMemcpy2D) to build the
 * batched layout from the per-unit states and back.
 *
 *   tv_units_host <problem.bin> <out.bin>
 *   problem.bin: int32 H, W, B, U, iters; then per unit: float32 y[H][W], Phi[H][W][B]       out.bin: float32 x[U][H][W][B] (form b)
 */
#include "pnp_solve.h"

static void tv_args(scipnp_admm_tv_args* tv, int M, int N, int B, int units, float* theta, float* b, float* x, float* raw,
                    const float* Phi, const float* y, const float* Phisum) {
    memset(tv, 0, sizeof *tv);
    tv->struct_size = sizeof *tv;
    tv->M = M; tv->N = N; tv->B = B; tv->two_stage = 0; tv->units = units;
    tv->theta = theta; tv->b = b; tv->x = x; tv->theta_raw = raw; tv->Phi = Phi; tv->y = y; tv->Phisum = Phisum;
    tv->c0 = 1.0; tv->c1 = 0.01; tv->tv_weight = 0.1f; tv->tv_iters = 5;
    tv->tv_workspace_bytes = scipnp_tv_workspace_bytes(M, N, 4 * B * (units > 1 ? units : 1), 5);
    tv->tv_workspace = dmalloc(tv->tv_workspace_bytes);
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[5];
    rd(hdr, sizeof hdr, f);
    const int H = hdr[0], W = hdr[1], B = hdr[2], U = hdr[3], iters = hdr[4];
    const int M = H / 2, N = W / 2;
    const size_t HW = (size_t)H * W, E = HW * B, Q = HW;                     /* Q = 4 M N floats per frame of a unit */
    hipStream_t st; HIPCHK(hipStreamCreate(&st));
    float* y_h = malloc(HW * 4); float* Phi_h = malloc(E * 4);
    float *mosaic = dmalloc(E * 4), *ymos = dmalloc(HW * 4);
    /* ---- (a) unit after unit; the per-unit states are kept for the layout permutation of (b) */
    float** Phi_u = malloc(U * sizeof(float*)); float** y_u = malloc(U * sizeof(float*)); float** x_u = malloc(U * sizeof(float*));
    for (int u = 0; u < U; ++u) {
        rd(y_h, HW * 4, f); rd(Phi_h, E * 4, f);
        HIPCHK(hipMemcpy(ymos, y_h, HW * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(mosaic, Phi_h, E * 4, hipMemcpyHostToDevice));
        Phi_u[u] = dmalloc(E * 4); y_u[u] = dmalloc(HW * 4); x_u[u] = dmalloc(E * 4);
        float *Phisum = dmalloc(HW * 4), *theta = dmalloc(E * 4), *b = dmalloc(E * 4), *raw = dmalloc(E * 4);
        SCICHK(scipnp_mosaic_to_state(mosaic, Phi_u[u], M, N, B, st));
        SCICHK(scipnp_y_to_meas(ymos, y_u[u], M, N, st));
        SCICHK(scipnp_pm_setup(Phi_u[u], y_u[u], Phisum, theta, M, N, B, st));
        scipnp_admm_tv_args tv;
        tv_args(&tv, M, N, B, 0, theta, b, x_u[u], raw, Phi_u[u], y_u[u], Phisum);
        for (int k = 0; k < iters; ++k) SCICHK(scipnp_admm_tv_iterate(&tv, NULL, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    fclose(f);
    /* ---- (b) all units in ONE launch sequence: Phi [B][U][Q], y [U][Q] from the per-unit tensors (memory movement only) */
    float *PhiB = dmalloc(E * U * 4), *yB = dmalloc(HW * U * 4), *PhisumB = dmalloc(HW * U * 4);
    float *thetaB = dmalloc(E * U * 4), *bB = dmalloc(E * U * 4), *xB = dmalloc(E * U * 4), *rawB = dmalloc(E * U * 4);
    for (int u = 0; u < U; ++u) {
        /* frame t of unit u: Q floats at PhiB + (t * U + u) * Q  <-  Phi_u[u] + t * Q */
        HIPCHK(hipMemcpy2D(PhiB + (size_t)u * Q, (size_t)U * Q * 4, Phi_u[u], Q * 4, Q * 4, B, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(yB + (size_t)u * Q, y_u[u], Q * 4, hipMemcpyDeviceToDevice));
    }
    SCICHK(scipnp_pm_setup_units(PhiB, yB, PhisumB, thetaB, M, N, B, U, st));
    scipnp_admm_tv_args tv;
    tv_args(&tv, M, N, B, U, thetaB, bB, xB, rawB, PhiB, yB, PhisumB);
    for (int k = 0; k < iters; ++k) SCICHK(scipnp_admm_tv_iterate(&tv, NULL, st));
    HIPCHK(hipStreamSynchronize(st));
    /* ---- compare every unit's x (the one-stage solver reports x) bit for bit, write form (b) as (H,W,B) mosaics */
    float* a_h = malloc(E * 4); float* b_h = malloc(E * 4);
    float* xunit = dmalloc(E * 4);
    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 1; }
    int bad = 0;
    for (int u = 0; u < U; ++u) {
        HIPCHK(hipMemcpy2D(xunit, Q * 4, xB + (size_t)u * Q, (size_t)U * Q * 4, Q * 4, B, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(a_h, x_u[u], E * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b_h, xunit, E * 4, hipMemcpyDeviceToHost));
        if (memcmp(a_h, b_h, E * 4) != 0) { ++bad; fprintf(stderr, "unit %d differs\n", u); }
        SCICHK(scipnp_state_to_mosaic(xunit, mosaic, M, N, B, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(a_h, mosaic, E * 4, hipMemcpyDeviceToHost));
        if (fwrite(a_h, 4, E, g) != E) { perror(argv[2]); return 1; }
    }
    fclose(g);
    printf("%d units of %dx%dx%d, %d ADMM-TV iterations: batched run %s the unit-after-unit runs\n", U, H, W, B, iters,
           bad ? "DIFFERS from" : "is bit-identical to");
    return bad ? 7 : 0;
}
