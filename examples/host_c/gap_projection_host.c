/* A plain C host of libscipnp.so (no Python, no PyTorch): the GAP / ADMM Euclidean projection of video SCI,
 *     x = p + Phi^T( (y - Phi p) / (alpha*rho + Phi Phi^T) ),   p = theta - b/rho,
 * on the reference's own tensor layout -- Bayer planes (M,N,B,4), measurement (M,N,4) -- through the C ABI of
 * include/scipnp.h, checked against a direct host evaluation of the same formula.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include examples/host_c/gap_projection_host.c \
 *       -L adaptivepnp_sci_amd -lscipnp -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/adaptivepnp_sci_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/gap_projection_host && /tmp/gap_projection_host
 *
 * This is how a C / C++ application would bind the hot path; tests/test_gpu_cabi_host.py builds and runs it. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "scipnp.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP: %s\n", hipGetErrorString(r_)); return 2; } } while (0)
#define CHECK_SCI(e) do { int r_ = (e); if (r_ != SCIPNP_OK) { fprintf(stderr, "scipnp(%d): %s\n", r_, scipnp_last_error()); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFFFF) / 16777216.0f; }

int main(void) {
    const int M = 37, N = 50, B = 8;                    /* quarter-resolution planes of a 74 x 100 x 8 mosaic cube */
    const size_t nq = (size_t)M * N, ne = nq * B * 4, nm = nq * 4;
    const float rho = 0.55f, alpha = 1.0f;
    printf("scipnp %s for %s\n", scipnp_version(), scipnp_arch());

    float *theta = malloc(ne * 4), *b = malloc(ne * 4), *Phi = malloc(ne * 4), *y = malloc(nm * 4), *x = malloc(ne * 4);
    unsigned seed = 12345u;
    for (size_t i = 0; i < ne; ++i) { theta[i] = frand(&seed); b[i] = 0.1f * (frand(&seed) - 0.5f); Phi[i] = frand(&seed) < 0.5f ? 0.f : 1.f; }
    for (size_t i = 0; i < nm; ++i) y[i] = 4.0f * frand(&seed);

    float *d_theta, *d_b, *d_Phi, *d_y, *d_sum, *d_x, *d_Ax;
    CHECK_HIP(hipMalloc((void**)&d_theta, ne * 4)); CHECK_HIP(hipMalloc((void**)&d_b, ne * 4));
    CHECK_HIP(hipMalloc((void**)&d_Phi, ne * 4));   CHECK_HIP(hipMalloc((void**)&d_y, nm * 4));
    CHECK_HIP(hipMalloc((void**)&d_sum, nm * 4));   CHECK_HIP(hipMalloc((void**)&d_x, ne * 4));
    CHECK_HIP(hipMalloc((void**)&d_Ax, nm * 4));
    CHECK_HIP(hipMemcpy(d_theta, theta, ne * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_b, b, ne * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_Phi, Phi, ne * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_y, y, nm * 4, hipMemcpyHostToDevice));

    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_SCI(scipnp_phisum(d_Phi, d_sum, M, N, B, st));                                       /* Phi Phi^T, zeros -> 1 */
    CHECK_SCI(scipnp_proj_twostage(d_theta, d_b, d_Phi, d_y, d_sum, d_x, M, N, B, 1.0f / rho, alpha * rho, st));
    CHECK_SCI(scipnp_A(d_x, d_Phi, d_Ax, M, N, B, st));                                         /* forward model of the result */
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(x, d_x, ne * 4, hipMemcpyDeviceToHost));

    /* the same formula on the host, in double */
    double num = 0.0, den = 0.0;
    for (size_t q = 0; q < nq; ++q)
        for (int ib = 0; ib < 4; ++ib) {
            double s = 0.0, ps = 0.0;
            for (int t = 0; t < B; ++t) {
                const size_t i = (q * B + t) * 4 + ib;
                s += Phi[i];
                ps += ((double)theta[i] - (double)b[i] / rho) * Phi[i];
            }
            if (s == 0.0) s = 1.0;
            const double r = ((double)y[q * 4 + ib] - ps) / ((double)alpha * rho + s);
            for (int t = 0; t < B; ++t) {
                const size_t i = (q * B + t) * 4 + ib;
                const double want = ((double)theta[i] - (double)b[i] / rho) + Phi[i] * r;
                num += (x[i] - want) * (x[i] - want);
                den += want * want;
            }
        }
    const double rel = sqrt(num / den);
    printf("projection rel-L2 vs host double: %.3e\n", rel);

    /* error path: a misaligned pointer must be refused with a message, not crash */
    const int rc = scipnp_A(d_x + 1, d_Phi, d_Ax, M, N, B, st);
    printf("misaligned call -> %d (%s)\n", rc, rc ? scipnp_last_error() : "accepted");
    hipStreamDestroy(st);
    return (rel < 1e-6 && rc != SCIPNP_OK) ? 0 : 1;
}
