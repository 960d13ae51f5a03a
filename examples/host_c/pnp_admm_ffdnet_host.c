/* A complete reconstruction from a plain C host of libscipnp.so (no Python, no PyTorch): ADMM-TV warm start followed
 * by two-stage PnP-ADMM with the FFDNet-colour denoiser on the split-fp16 kernels (default) or, with a third argument
 * `f32`, in fp32 arithmetic on the Winograd kernels -- the pipeline of the reference's
 * ADMM_TV_Warm_Start_save.py + two_stage_ADMM_Online_FFD_Warm.py (without the online finetune) -- driven through the
 * iteration-level entries scipnp_admm_tv_iterate / scipnp_twostage_ffdnet_iterate of include/scipnp.h (pnp_solve.h).
 *
 *   pnp_admm_ffdnet_host <problem.bin> <out_mosaic.bin> [f32 | 2s]
 *     f32: fp32 Winograd kernels;  2s: split-fp16 with half of the frames of every network pass on the host's side stream
 * problem.bin (little endian): int32 H, W, B, nb, tv_iters, iters; float32 sigma; y[H*W]; Phi[H*W*B] ((H,W,B) order);
 *   then nb layers: int32 cout, cin; float32 weight[cout*cin*9] (OIHW); float32 bias[cout].
 * out_mosaic.bin: float32 (H,W,B) reconstruction.       tests/test_gpu_cabi_host.py builds, runs and checks it
 * bit-for-bit against the Python drop-in solver. */
#include "pnp_solve.h"

int main(int argc, char** argv) {
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s problem.bin out.bin [f32 | 2s]\n", argv[0]); return 1; }
    pnp_solve_t job;
    memset(&job, 0, sizeof job);
    job.problem = argv[1]; job.out = argv[2]; job.input_scale = 1.0f;
    job.f32 = argc == 4 && strcmp(argv[3], "f32") == 0;
    job.two_streams = argc == 4 && strcmp(argv[3], "2s") == 0;
    pnp_solve(&job);
    if (job.overflow) { fprintf(stderr, "activations left fp16 range\n"); return 5; }
    printf("%s: %dx%dx%d, %d ADMM-TV + %d ADMM-FFDNet iterations (%s) done\n", scipnp_version(), job.H, job.W, job.B,
           job.tv_iters, job.iters, job.f32 ? "fp32 Winograd" : job.two_streams ? "split-fp16, two streams" : "split-fp16");
    return 0;
}
