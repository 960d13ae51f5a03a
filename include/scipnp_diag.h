/* scipnp_diag.h -- the LABORATORY beside the product ABI (libscipnp_diag.so, built by `make` next to libscipnp.so).
 *
 * Nothing here is on a reconstruction path and no source under adaptivepnp_sci_amd/ calls it: micro-benchmarks that measure
 * the ceilings the rooflines are divided by (csrc/peaks.hip), instantiations of the product's Winograd kernels with
 * s_memtime stamps or with parts switched off (the SCIPNP_DIAG_BUILD sections of csrc/conv_wino.hip / conv_wino4.hip), and the
 * persistent F(2x2,3x3) kernel that round 3 built, measured 10 % slower than the classic one and did not adopt
 * (csrc/conv_winop.hip).  Users: bench.py's `measured_peaks`, tools/peaks_bench.py, tools/probes, two -m gpu tests.
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

/* ---- three waves per SIMD (csrc/conv_wino4x.hip, round 5; measured, not adopted): scipnp_conv3x3_c8w4's convolution with the 36
 * positions of a tile split over THREE waves by rows of the transformed patch ({1,2}, {3,4}, {0,5}: four packed operations per
 * patch column each), 96 accumulator registers per wave, 164 VGPRs; workgroup = 12 waves = 4 tile rows x 3 thirds (16 x 64 pixels x
 * 32 channels, one per CU: a 6-wave workgroup reserves two wave slots on every SIMD and a second one does not fit).  The same
 * packed_wino4 buffer, the same products and summation orders: results BIT-IDENTICAL to scipnp_conv3x3_c8w4.
 * flags: bit0 ReLU, bit1 residual, bit4 ReLU-backward mask, bit8 head tag (no PixelShuffle store).  277 us against 249 us on the FFDNet
 * body layer (profiles/r05a_*): twelve waves in lockstep through one barrier per k-step leave the matrix pipe idle at every
 * k-step's start and end.  _stamped / _diag: stamp slots and masks as for scipnp_conv3x3_c8w4 (stamped builds: masks 0, 1, 6, 7). */
int scipnp_conv3x3_c8w6(const float* in, const float* packed_wino4, float* out, const float* residual,
                        const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);
int scipnp_conv3x3_c8w6_stamped(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s);
int scipnp_conv3x3_c8w6_diag(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                             int flags, int diag, scipnp_stream_t s);

/* ---- 16-channel workgroups, three per CU (csrc/conv_wino4n.hip, round 5; measured, see profiles/r05e_*): scipnp_conv3x3_c8w4's
 * workgroup shape computing ONE 16-channel half -- 72 accumulator registers, U slabs of 9 KB, a single raw-tile buffer: 39 KB of LDS
 * and <= 168 VGPRs, i.e. three INDEPENDENT 4-wave workgroups per CU; twice the input transform, patch reads and raw-tile requests
 * per MFMA.  Weights: scipnp_repack_wino4n(packed_wino4 -> packed_wino4n of scipnp_conv3x3_wino4n_packed_floats floats).  Results
 * BIT-IDENTICAL to scipnp_conv3x3_c8w4; flags: bit0 ReLU, bit1 residual, bit4 mask.  _stamped: stamp slots as scipnp_conv3x3_c8w4_stamped
 * (SCIPNP_WN_WGS_PER_CU = 1 / 2 pads the LDS request); _diag: masks 1, 2, 4, 8, 16, 6, 7, 15. */
size_t scipnp_conv3x3_wino4n_packed_floats(int Cin, int Cout);
int scipnp_repack_wino4n(const float* packed_wino4, float* packed_wino4n, int Cin, int Cout, scipnp_stream_t s);
int scipnp_conv3x3_c8wn(const float* in, const float* packed_wino4n, float* out, const float* residual, const float* mask_src,
                        int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);
int scipnp_conv3x3_c8wn_stamped(const float* in, const float* packed_wino4n, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s);
int scipnp_conv3x3_c8wn_diag(const float* in, const float* packed_wino4n, float* out, int n, int Cin, int Cout, int h, int w,
                             int flags, int diag, scipnp_stream_t s);

/* ---- producer / consumer waves (csrc/conv_wino4p.hip, round 5; measured at parity, not adopted): scipnp_conv3x3_c8w4's convolution
 * for Cout % 64 == 0 on 12-wave workgroups, one per CU -- 8 consumer waves that only multiply (U and V operands from LDS, 144
 * accumulators, 168 VGPRs) and 4 producer waves that own the raw-tile requests and the input transform, done ONCE for 64 output
 * channels and written to LDS per k-step (the U requests are spread over the consumers: a wave's LDS-DMA requests execute one after
 * the other).  The same packed_wino4 buffer, BIT-IDENTICAL results.  flags: bit0 ReLU, bit1 residual, bit4 mask.  Isolated layer -5 .. -9 %,
 * inside a 4-layer chain -0 .. -5 % (profiles/r05f_*): in lockstep through one barrier per k-step the two consumers of a SIMD take
 * 27.4 hundred cycles per k-step with the producers idle (23.0 = their MFMAs) and 31.4 with them -- a producer's 18 V stores complete
 * 2300 cycles after issue behind the consumers' operand reads.  _stamped: wave 0 (a consumer) [0] entry, [1] prologue barriers passed,
 * [8 + 2s] MFMAs of k-step s issued, [9 + 2s] its barrier passed, [3] loop left, [4] first tile image written, [5] stores issued,
 * [6] acknowledged, [7] XCC_ID << 32 | HW_ID; wave 8 (a producer) [56] opening tiles landed, [57] first transform in LDS,
 * [64 + 2s] work of k-step s issued and its LDS operations complete, [65 + 2s] its barrier passed; flags bits 12..14 switch parts off
 * (1 no V stores, 2 no raw staging, 4 no transform: timing only) */
int scipnp_conv3x3_c8wp(const float* in, const float* packed_wino4, float* out, const float* residual,
                        const float* mask_src, int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);
int scipnp_conv3x3_c8wp_stamped(const float* in, const float* packed_wino4, float* out, int n, int Cin, int Cout, int h, int w,
                                int flags, unsigned long long* stamps, scipnp_stream_t s);

/* ---- persistent form of the fp32 Winograd convolution for 96-output-channel layers (csrc/conv_winop.hip, round 3): same
 * arithmetic and summation order as scipnp_conv3x3_c8w (bit-identical results), the input transform computed once per tile and
 * shared through LDS by all 96 output channels, 12-wave workgroups that stay resident (one per CU) and walk a static list of
 * 4-row x 32-column units with the channel-group pipeline running across unit boundaries.  Weights in their own slab layout:
 * scipnp_pack_conv3x3_winop from the fp32 direct packing (scipnp_conv3x3_winop_packed_floats floats; 0 if unsupported).
 * flags: bit0 ReLU, bit1 residual (fp32 c8, output shape), bit4 ReLU mask (mask_src), bit8 head tag; stride 1 only.
 * scipnp_conv3x3_c8p_supported(Cin, Cout) = 1 for Cin % 8 == 0, Cout == 96. */
int scipnp_conv3x3_c8p_supported(int Cin, int Cout);
size_t scipnp_conv3x3_winop_packed_floats(int Cin, int Cout);
int scipnp_pack_conv3x3_winop(const float* packed_f32, float* packed_winop, int Cin, int Cout, scipnp_stream_t s);
int scipnp_conv3x3_c8p(const float* in, const float* packed_winop, float* out, const float* residual, const float* mask_src,
                       int n, int Cin, int Cout, int h, int w, int flags, scipnp_stream_t s);
/* diagnostic: the same kernel with parts switched off (timing only, WRONG results) -- tools/probes/winop_ablate.py */
int scipnp_conv3x3_c8p_diag(const float* in, const float* packed_winop, float* out, int n, int Cin, int Cout, int h, int w,
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
