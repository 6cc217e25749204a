// MODE.FP16_OVFL (hwreg(HW_REG_MODE, 23, 1)) on gfx950: what do v_cvt_f16_f32 / v_cvt_pk_f16_f32 return for overflow, infinity and NaN with the bit set?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/f16_ovfl_probe.hip -o /tmp/f16_ovfl_probe && /tmp/f16_ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
__global__ void k(const float* in, unsigned short* out, int mode) {
    if (mode) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);
    const float a = in[threadIdx.x];
    const f2 v = {a, -a};
    const h2 h = __builtin_convertvector(v, h2);
    out[threadIdx.x * 3 + 0] = __builtin_bit_cast(unsigned short, h[0]);
    out[threadIdx.x * 3 + 1] = __builtin_bit_cast(unsigned short, h[1]);
    out[threadIdx.x * 3 + 2] = __builtin_bit_cast(unsigned short, (_Float16)a);
}
int main() {
    const float h_in[8] = {1.0f, 65504.0f, 65520.0f, 1e6f, 3e38f, INFINITY, NAN, 70000.0f};
    float* d_in; unsigned short* d_out; hipMalloc(&d_in, 32); hipMalloc(&d_out, 8 * 3 * 2);
    hipMemcpy(d_in, h_in, 32, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d_in, d_out, mode);
        unsigned short o[24]; hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
        for (int i = 0; i < 8; ++i) printf("FP16_OVFL=%d  %12g -> pk %04x  pk(-x) %04x  scalar %04x\n", mode, h_in[i], o[i * 3], o[i * 3 + 1], o[i * 3 + 2]);
    }
    return 0;
}
