// ds_read_b64_tr_b16 lane mapping probe (gfx950): LDS holds a [16 keys][256 channels] 16-bit image, element value = key * 256 + channel.
// Lane (n = l & 15, g = l >> 4) supplies the address of key 4 g + (n >> 2), channel 16 ct + 4 (n & 3); expected result (the B operand of
// v_mfma_f32_16x16x16_f16: 4 consecutive k of column n): keys 4 g .. 4 g + 3 at channel 16 ct + n.
//   hipcc --offload-arch=gfx950 tools/probes/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int ct) {
    __shared__ __attribute__((aligned(16))) short lds[16 * 256];
    for (int i = threadIdx.x; i < 16 * 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, n = l & 15, g = l >> 4;
    const short* p = lds + (4 * g + (n >> 2)) * 256 + 16 * ct + 4 * (n & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    *reinterpret_cast<s16x4*>(out + l * 4) = v;
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    int bad = 0;
    for (int ct = 0; ct < 16; ct += 5) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ct);
        short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int want = (4 * (l >> 4) + j) * 256 + 16 * ct + (l & 15);
                if (h[l * 4 + j] != want) { if (bad < 8) printf("ct %d lane %d elem %d: got key %d ch %d, want key %d ch %d\n", ct, l, j, h[l*4+j] / 256, h[l*4+j] % 256, want / 256, want % 256); ++bad; }
            }
    }
    printf("tr16 probe: %d mismatches\n", bad);
    return bad != 0;
}
