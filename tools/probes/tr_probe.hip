// Probe of ds_read_b64_tr_b16 addressing as used by the split-fp16 weight-gradient kernel (csrc/wgrad_split.hip):
// LDS image [channel group][pixel][8 ch] fp16; expects lane (li, lh) to receive pixels 8*lh+4r .. +3 of channel li.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int NPX = 64, NG = 4;
__global__ void k(const _Float16* in, _Float16* out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[NG * NPX * 8];
    for (int i = threadIdx.x; i < NG * NPX * 8; i += 64) lds[i] = in[i];
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
    const int l = threadIdx.x, G = l >> 4, l16 = l & 15, q = l16 >> 2, p = l16 & 3, lh = l >> 5;
    for (int r = 0; r < 2; ++r) {
        const int px = 8 * lh + 4 * r + q, cg = 2 * (G & 1) + (p >> 1);
        auto ptr = (__attribute__((address_space(3))) s16x4*)(lds + ((cg * NPX + px) * 8 + 4 * (p & 1)));
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
        *(s16x4*)(out + (l * 2 + r) * 4) = v;
    }
#endif
}
int main() {
    _Float16 h[NG * NPX * 8], o[64 * 8];
    for (int g = 0; g < NG; ++g) for (int px = 0; px < NPX; ++px) for (int c = 0; c < 8; ++c)
        h[(g * NPX + px) * 8 + c] = (_Float16)(px * 32 + g * 8 + c);
    _Float16 *di, *dout;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 2; ++r) for (int j = 0; j < 4; ++j) {
        const int li = l & 31, lh = l >> 5;
        const float want = (8 * lh + 4 * r + j) * 32 + li;
        const float got = (float)o[(l * 2 + r) * 4 + j];
        if (want != got) { if (bad < 8) printf("lane %d r %d j %d: want %g got %g\n", l, r, j, want, got); ++bad; }
    }
    printf("tr_probe mismatches: %d\n", bad);
    return bad != 0;
}
