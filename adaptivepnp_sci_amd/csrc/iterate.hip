// One whole ADMM iteration as ONE C-ABI call: the launch sequence the Python stepper (solver.AdmmRun.step) issues,
// for hosts that run the solver loop natively.
//   two-stage + FFDNet-colour (Malvar demosaic):  dvp_linear_inv_2_stage_ADMM_tensor_online.py:121-271, one pass
//   ADMM-TV, either solver:                       :121-160 + :265-271  /  :385-407 + :500-509
#include "common.hpp"
#include <cstdlib>

using namespace scipnp;

static int device_cu_count() {
    static const int n = [] {
        int dev = 0, cu = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 256;
        return cu > 0 ? cu : 256;
    }();
    return n;
}

// which launches an ADMM-TV call on this argument block takes
struct TvPath {
    bool banded;    // the banded TV kernel (many workgroups per plane) + a dual update
    bool plane;     // ONE whole-plane kernel for all TV iterations, the stop test and the dual update (planes up to 128 x 128)
    bool defer;     // banded, in its candidate form, the dual update fused into the next call's projection
    int nstd;       // squared-error partials of the stand-alone dual update's grid (the size of sse_part)
};

static TvPath tv_path(const scipnp_admm_tv_args* a) {
    const int M = a->M, N = a->N, B = a->B, U = a->units > 1 ? a->units : 1, BU = B * U;
    TvPath p = {};
    scipnp_sse_partials(a->x, a->x, (size_t)4 * M * N * BU, nullptr, &p.nstd, nullptr);   // size query: pm_dual_update's grid
    const bool want_sse = a->sse_part && a->orig;
    // planes up to 256 columns take the banded TV kernel (many workgroups per plane, one launch) followed by the dual update;
    // narrower problems that leave the chip idle either way keep the one-launch whole-plane form with the dual update in its epilogue
    bool banded = tv_band_fits(M, N, a->tv_iters) && (long long)4 * BU * ((M + 31) / 32) >= 128;
    // unit batches of small planes: ONE workgroup per plane (all TV iterations in registers, the stop test and the dual update in
    // the same launch, no candidates written) fills the chip once there is a plane per CU -- 256 planes of 128 x 128 take 1 x
    // the whole-plane kernel's time where the banded form + fused projection pay per plane (tools/probes/tv_units_probe.py)
    if (banded && U > 1 && tv_plane_dual_fits(M, N, 4 * BU, p.nstd, want_sse)) {
        static const int force = [] { const char* e = lab_switch("SCIPNP_TV_PLANE_BATCH"); return e ? atoi(e) : -1; }();
        const int cus = device_cu_count();
        const long long planes = 4LL * BU, gens = (planes + cus - 1) / cus;
        const bool fills = planes * 4 >= gens * cus * 3;                    // the last generation leaves at most a quarter idle
        if (force == 1 || (force != 0 && fills)) banded = false;
    }
    p.banded = banded;
    p.plane = !banded && tv_plane_dual_fits(M, N, 4 * BU, p.nstd, want_sse);
    // deferred form: TV in its one-launch candidate form (no second launch, nothing recomputed; theta_raw is not written), the
    // dual update -- which then also picks every channel's candidate -- fused into the next call's projection
    // (unit batches: the fused launch keeps the stop iterations of at most 4 planes per workgroup -- planes of >= 1024 pixels)
    p.defer = banded && a->defer_state && scipnp_pm_dual_project_fits(M, N, B) && tv_candidates_fit(M, N, a->tv_iters) &&
              (U == 1 || (long long)M * N >= 1024);
    return p;
}

extern "C" {

/* 1 if scipnp_admm_tv_iterate on this block runs the whole-plane kernel (its squared-error partials: one per plane, then zeros) */
int scipnp_admm_tv_plane_path(const scipnp_admm_tv_args* a) {
    if (!a || a->struct_size != sizeof(scipnp_admm_tv_args)) return 0;
    return tv_path(a).plane ? 1 : 0;
}

int scipnp_twostage_ffdnet_iterate(const scipnp_twostage_ffdnet_args* a, int* nblocks, scipnp_stream_t s) {
    SCIPNP_REQUIRE(a, "null argument block");
    SCIPNP_REQUIRE(a->struct_size == sizeof(scipnp_twostage_ffdnet_args),
                   "scipnp_twostage_ffdnet_args.struct_size is %zu, this library's block has %zu bytes (set it to sizeof the "
                   "struct of the header you compile against, and rebuild against include/scipnp.h of this library)",
                   a->struct_size, sizeof(scipnp_twostage_ffdnet_args));
    SCIPNP_REQUIRE(a->conv_form >= 0 && a->conv_form <= SCIPNP_CONV_F32_WINO_F4, "conv_form %d: 0 (inferred), 1 split-fp16, 2 fp32 F(2x2), 3 fp32 F(4x4)", a->conv_form);
    SCIPNP_REQUIRE(a->conv_form != SCIPNP_CONV_SPLIT_F16 || (a->packed_split && a->net_in_c8s), "conv_form split-fp16 needs packed_split and net_in_c8s");
    SCIPNP_REQUIRE(a->conv_form < SCIPNP_CONV_F32_WINO_F2 || (a->packed_wino && a->net_in_c8), "conv_form fp32 needs packed_wino and net_in_c8");
    SCIPNP_REQUIRE(a->conv_form != SCIPNP_CONV_F32_WINO_F4 || a->packed_wino4, "conv_form fp32 F(4x4) needs packed_wino4");
    const bool f32 = a->conv_form ? a->conv_form != SCIPNP_CONV_SPLIT_F16 : a->packed_wino != nullptr;
    const float* const* const p4 = a->conv_form == SCIPNP_CONV_F32_WINO_F2 ? nullptr : a->packed_wino4;
    SCIPNP_REQUIRE(a->theta && a->b && a->x && a->Phi && a->y && a->Phisum && a->w && a->x_rgb && a->net_out_c8 && a->scratch0 &&
                   a->scratch1 && (f32 ? (a->net_in_c8 != nullptr) : (a->net_in_c8s && a->packed_split)),
                   "null pointer in argument block");
    SCIPNP_REQUIRE(a->rho > 0.0 && a->tau > 0.0, "rho and tau must be positive");
    const int M = a->M, N = a->N, B = a->B, U = a->units > 1 ? a->units : 1, BU = B * U;   // (unit batch: B*U frames, see scipnp.h)
    // one rounding from double, like the reference's Python scalars handed to PyTorch (1 / rou, alpha * rou, 1 / tau)
    const float inv_rho = (float)(1.0 / a->rho), inv_tau = (float)(1.0 / a->tau), alpha_rho = (float)(a->alpha * a->rho);
    // x = p + Phi^T((y - Phi p)/(alpha rho + Phi Phi^T)),  p = theta - b/rho                      (:128-140)
    int rc = scipnp_pm_project_units(a->theta, a->b, a->Phi, a->y, a->Phisum, a->x, M, N, B, U, 0, inv_rho, alpha_rho, s);
    if (rc) return rc;
    // mosaic of x + b/rho, Malvar demosaic, x_rgb - w/tau, FFDNet input (pixel-unshuffle + sigma map, c8s or fp32 c8)   (:168-198)
    rc = scipnp_pm_pre_denoise_ex(a->x, a->b, a->w, a->x_rgb, nullptr, f32 ? a->net_in_c8 : nullptr,
                                  f32 ? nullptr : a->net_in_c8s, M, N, BU, inv_rho, inv_tau, a->sigma, s);
    if (rc) return rc;
    if (f32) {
        rc = scipnp_ffdnet_forward_c8w4(a->net_in_c8, a->net_out_c8, a->packed_wino, p4, a->nb, a->nc,
                                        (float*)a->scratch0, (float*)a->scratch1, BU, M, N, s);
    } else {
        // the solve's own range-guard word for the launches of this call; the thread's binding is restored afterwards
        OverflowScope scope(a->overflow_word);
        rc = scipnp_ffdnet_forward_c8s_2s(a->net_in_c8s, a->net_out_c8, a->packed_split, a->nb, a->nc, a->scratch0, a->scratch1,
                                          BU, M, N, s, a->side_stream, a->side_fork_event, a->side_join_event);
    }
    if (rc) return rc;
    // theta = clip(CFA samples of the denoised frames), b += x - theta, w += x_rgb - out, PSNR partials   (:206-209, :265-281)
    return scipnp_pm_post_denoise(nullptr, a->net_out_c8, a->out_rgb, a->x, a->x_rgb, a->theta, a->b, a->w, a->orig,
                                  a->sse_part, a->first_iter, M, N, BU, nblocks, s);
}

int scipnp_admm_tv_iterate(const scipnp_admm_tv_args* a, int* nblocks, scipnp_stream_t s) {
    SCIPNP_REQUIRE(a, "null argument block");
    SCIPNP_REQUIRE(a->struct_size == sizeof(scipnp_admm_tv_args),
                   "scipnp_admm_tv_args.struct_size is %zu, this library's block has %zu bytes", a->struct_size,
                   sizeof(scipnp_admm_tv_args));
    SCIPNP_REQUIRE(a->theta && a->b && a->x && a->theta_raw && a->Phi && a->y && a->Phisum && a->tv_workspace,
                   "null pointer in argument block");
    const int M = a->M, N = a->N, B = a->B, U = a->units > 1 ? a->units : 1, BU = B * U;   // (unit batch: 4*B*U TV channels)
    int rc;
    // planes up to 256 columns take the banded TV kernel (many workgroups per plane, one launch) followed by the dual update;
    // narrower problems that leave the chip idle either way keep the one-launch whole-plane form with the dual update in its epilogue
    const TvPath path = tv_path(a);
    const int nstd = path.nstd;
    const bool want_sse = a->sse_part && a->orig, defer = path.defer;
    TvCandidates cd = {};
    if (defer) tv_candidate_ptrs(M, N, 4 * BU, a->tv_iters, a->tv_workspace, &cd);
    float coef, sign, pc0, pc1;
    int mode;
    if (a->two_stage) {
        SCIPNP_REQUIRE(a->c0 > 0.0, "rho must be positive");
        coef = (float)(1.0 / a->c0); sign = +1.0f;          // theta = TV(x + b/rho), b += x - theta
        mode = 0; pc0 = coef; pc1 = (float)(a->c1 * a->c0);
    } else {
        coef = -1.0f; sign = -1.0f;                         // theta = TV(x - b),     b -= x - theta
        mode = 1; pc0 = (float)a->c0; pc1 = (float)a->c1;
    }
    if (defer && *a->defer_state)       // the previous call's dual update and this call's projection in one launch
        rc = pm_dual_project_sel(cd.cand, &cd, tv_scalar_as_double(a->tv_weight), tv_scalar_as_double(2e-4f), a->x, a->theta, a->b,
                                 a->Phi, a->y, a->Phisum, a->orig, a->orig ? a->sse_part_prev : nullptr, nstd, M, N, B, mode, pc0,
                                 pc1, (hipStream_t)s, U);
    else
        rc = scipnp_pm_project_units(a->theta, a->b, a->Phi, a->y, a->Phisum, a->x, M, N, B, U, mode, pc0, pc1, s);
    if (rc) return rc;
    if (a->defer_state) *a->defer_state = 0;
    if (path.plane) {
        if (nblocks) *nblocks = nstd;
        return tv_plane_dual(a->x, a->b, coef, a->theta, M, N, 4 * BU, a->tv_weight, 2e-4f, a->tv_iters, a->orig,
                             want_sse ? a->sse_part : nullptr, a->two_stage ? 0 : 1, sign, nstd, (hipStream_t)s);
    }
    if (defer) {                        // this iteration's dual update rides at the head of the next call (or the flush)
        rc = tv_band_candidates(a->x, a->b, coef, M, N, 4 * BU, a->tv_weight, 2e-4f, a->tv_iters, a->tv_workspace,
                                a->tv_workspace_bytes, (hipStream_t)s);
        if (rc) return rc;
        *a->defer_state = 1;
        if (nblocks) *nblocks = nstd;
        return SCIPNP_OK;
    }
    rc = scipnp_tv_chambolle(a->x, a->b, coef, a->theta_raw, M, N, 4 * BU, a->tv_weight, 2e-4f, a->tv_iters, a->tv_workspace,
                             a->tv_workspace_bytes, nullptr, s);
    if (rc) return rc;
    return scipnp_pm_dual_update(a->theta_raw, a->x, a->theta, a->b, a->orig, a->sse_part, a->two_stage ? 0 : 1, sign, M, N, BU,
                                 nblocks, s);
}

int scipnp_admm_tv_flush(const scipnp_admm_tv_args* a, int* nblocks, scipnp_stream_t s) {
    SCIPNP_REQUIRE(a, "null argument block");
    SCIPNP_REQUIRE(a->struct_size == sizeof(scipnp_admm_tv_args),
                   "scipnp_admm_tv_args.struct_size is %zu, this library's block has %zu bytes", a->struct_size,
                   sizeof(scipnp_admm_tv_args));
    if (!a->defer_state || !*a->defer_state) return SCIPNP_OK;
    SCIPNP_REQUIRE(a->theta && a->b && a->x && a->theta_raw, "null pointer in argument block");
    *a->defer_state = 0;
    const int BU = a->B * (a->units > 1 ? a->units : 1);
    TvCandidates cd = {};
    tv_candidate_ptrs(a->M, a->N, 4 * BU, a->tv_iters, a->tv_workspace, &cd);
    int rc = tv_stop_test_launch(cd, 4 * BU, a->tv_weight, 2e-4f, (hipStream_t)s);         // (the rare path: an extra launch)
    if (rc) return rc;
    return pm_dual_update_sel(cd.cand, cd.stop, a->x, a->theta, a->b, a->orig, a->orig ? a->sse_part : nullptr,
                              a->two_stage ? 0 : 1, a->two_stage ? +1.0f : -1.0f, a->M, a->N, BU, nblocks, (hipStream_t)s);
}

}  // extern "C"
