/* scipnp_lab.h -- kernels that were built, measured on MI355X and NOT adopted (lab/csrc/*.hip -> lab/libscipnp_lab.so, built by
 * `make -C lab` only; nothing under adaptivepnp_sci_amd/, bench.py or tests/ uses them).  The measurements are in profiles/ (r03*, r05a_*,
 * r05e_*, r05f_*); every entry is bit-identical to the product kernel it re-arranges (lab/test_lab_kernels.py).
 * Same conventions as include/scipnp.h; errors are read with scipnp_last_error() of libscipnp.so, which this library links. */
#ifndef SCIPNP_LAB_H
#define SCIPNP_LAB_H
#include "../include/scipnp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- three waves per SIMD (lab/csrc/conv_wino4x.hip, round 5; measured, not adopted): scipnp_conv3x3_c8w4's convolution with the 36
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

/* ---- 16-channel workgroups, three per CU (lab/csrc/conv_wino4n.hip, round 5; measured, see profiles/r05e_*): scipnp_conv3x3_c8w4's
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

/* ---- producer / consumer waves (lab/csrc/conv_wino4p.hip, round 5; measured at parity, not adopted): scipnp_conv3x3_c8w4's convolution
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

/* ---- persistent form of the fp32 Winograd convolution for 96-output-channel layers (lab/csrc/conv_winop.hip, round 3): same
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


#ifdef __cplusplus
}
#endif
#endif
