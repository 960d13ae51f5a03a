/* scipnp_diag.h -- the LABORATORY beside the product ABI (libscipnp_diag.so, built by `make` next to libscipnp.so).
 *
 * Nothing here is on a reconstruction path and no source under adaptivepnp_sci_amd/ calls it: micro-benchmarks that measure
 * the ceilings the rooflines are divided by (csrc/peaks.hip), instantiations of the product's Winograd kernels with
 * s_memtime stamps or with parts switched off (the SCIPNP_DIAG_BUILD sections of csrc/conv_wino.hip / conv_wino4.hip /
 * wgrad_wino4.hip).  Users: bench.py's `measured_peaks`, tools/peaks_bench.py, tools/probes.  (The kernels that were built,
 * measured and NOT adopted -- conv_winop, conv_wino4x / 4n / 4p -- live under lab/ with their own header, library and tests; the
 * default build does not compile them.)
 * Same conventions as include/scipnp.h (return codes, streams, alignment); errors are read with scipnp_last_error() of
 * libscipnp.so, which this library links. */
#ifndef SCIPNP_DIAG_H
#define SCIPNP_DIAG_H
#include "scipnp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* DIAGNOSTIC instantiation of scipnp_conv3x3_c8w4 with s_memtime stamps of wave 0 of every workgroup, written by scalar stores
 * (128 words per workgroup, grid = ceil(w/64)*ceil(h/8)*n*ceil(Cout/32); tools/probes/wino4_stamps.py): [0] entry, [1] first
 * tiles / slab in LDS, [2] first column pass done, [8 + 4g + {0,1,2,3}] k-step (g,0) MFMAs issued | its barrier passed | k-step
 * (g,1) MFMAs issued | its barrier passed (g < 24), [3] loop left, [4] partial tiles exchanged, [5] stores issued, [6] stores
 * acknowledged, [7] XCC_ID << 32 | HW_ID.  No product path calls it; the product kernel executes no stamp. */
int scipnp_conv3x3_c8w4_stamped(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s);
/* diagnostic: the F(4x4) kernel with parts switched off (timing only, WRONG results) -- tools/probes/wino4_ablate.py.
 * diag: bit0 no input transform, bit1 no raw-tile staging, bit2 no U LDS-DMA, bit3 no barriers in the K loop, bit4 no MFMAs,
 * bit5 no output transform / stores */
int scipnp_conv3x3_c8w4_diag(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                             int flags, int diag, scipnp_stream_t s);

/* DIAGNOSTIC instantiation of scipnp_conv3x3_c8w (layers with more than 16 outputs, flags bit0 only): the same kernel
 * with six s_memtime stamps per workgroup; no product path calls it and the product kernel executes no stamp.
 * stamps: 80 words per workgroup (grid = ceil(w/32)*ceil(h/8)*n*ceil(Cout/32)), written by its first lane:
 * [0] kernel entry, [1] first raw tiles + U slab in LDS, [2] first input transform done, [3] channel-group loop done,
 * [4] output transform done and stores issued, [5] stores acknowledged, [6] XCC_ID << 32 | HW_ID, [7] s_memrealtime
 * (100 MHz) at the end ([31]: at entry), [8 + g] end of channel group g (g < 24); with flags bit11 also
 * [32 + 16(g - 4) + p] after Winograd position p of groups g = 4, 5.  tools/probes/wino_stamps.py reads them. */
int scipnp_conv3x3_c8w_stamped(const float* in, const float* packed_wino, float* out, int n, int Cin, int Cout, int h, int w,
                               int flags, unsigned long long* stamps, scipnp_stream_t s);


/* ---------------------------------------------------------------------------------------------------------------
 * Diagnostics: measured ceilings for the rooflines (tools/peaks_bench.py; not on the reconstruction path).
 * scipnp_bench_mfma: register-resident MFMA loop on pseudo-random operands, mode 0 v_mfma_f32_32x32x16_f16,
 * 1 v_mfma_f32_16x16x32_f16, 2 v_mfma_f32_32x32x2_f32; `blocks` workgroups of 4 waves, iters x 4 (mode 1: x 8)
 * independent MFMAs per wave; out: blocks*256 floats.  scipnp_bench_stream: mode 0 reads n floats (sink: blocks*256
 * floats), mode 1 copies n floats.
 * ------------------------------------------------------------------------------------------------------------- */
int scipnp_bench_mfma(float* out, int blocks, int iters, int mode, scipnp_stream_t s);
/* scipnp_bench_mfma_valu: how much vector-ALU issue a matrix instruction hides -- per MFMA (f32 != 0:
 * v_mfma_f32_16x16x4_f32, else v_mfma_f32_32x32x16_f16; 32 matrix-pipe cycles either way) nv (0, 1, 2, 4, 6, 8) independent
 * v_add_f32 of the same wave; cycles[blocks*4]: s_memtime ticks of each wave's loop of iters x 16 MFMAs. */
int scipnp_bench_mfma_valu(float* out, unsigned long long* cycles, int blocks, int iters, int nv, int f32, scipnp_stream_t s);
/* accumulation-chain issue patterns of v_mfma_f32_16x16x4_f32: second use of an accumulator `dist` MFMAs behind the first */
int scipnp_bench_mfma_dep(float* out, unsigned long long* cycles, int blocks, int iters, int dist, scipnp_stream_t s);
/* VGPR-bank placement of the A / B operands of v_mfma_f32_16x16x4_f32 (var 0..3, csrc/peaks.hip) */
int scipnp_bench_mfma_bank(float* out, unsigned long long* cycles, int blocks, int iters, int var, scipnp_stream_t s);
int scipnp_bench_stream(const float* in, float* out, size_t n, int mode, int blocks, float* sink, scipnp_stream_t s);

/* csrc/wgrad_wino4.hip, laboratory instantiation: the F(4x4)-domain weight-gradient kernel of scipnp_conv3x3_wgrad_wino4 with its
 * timing-only ablation switches -- dbg bits: 1 no MFMAs, 2 no raw loads / LDS stores of them, 4 no transform, 8 placement probe
 * (every wave writes its HW_ID register to workspace[12 * workgroup + wave] and returns), 32 loads without the LDS stores,
 * 64 LDS stores without the loads.  Results are meaningless unless dbg == 0.  Workspace: the product entry's size. */
int scipnp_diag_conv3x3_wgrad_wino4(const float* act_c8, const float* dz_c8, float* dW, float* workspace, int nslab, int n,
                                    int Cin_real, int Cout_real, int Cin, int Cout, int h, int w, int dbg, scipnp_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* SCIPNP_DIAG_H */
