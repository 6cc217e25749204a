// OCP e4m3 conversion probe (gfx950): the 8-bit lo rows of the index-exact route (csrc/common.h lo8_*) rely on
//   decode: v_cvt_scalef32_pk_f16_fp8(src, 2^-12)  = fp16( e4m3fn(byte) * 2^-12 ), subnormal fp16 results kept
//   encode: v_cvt_pk_fp8_f32(clamp(lo * 2^12, +-448)) = round-to-nearest-even into e4m3fn
// Checks all 256 bytes (decode) and every fp16 bit pattern as lo (encode) against a host restatement of the OCP format.
//   hipcc --offload-arch=gfx950 tools/probes/fp8_probe.hip -o /tmp/fp8_probe && /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void dec(unsigned short* out) {               // thread t: byte t in all four positions of a dword
    const int b = threadIdx.x + blockIdx.x * blockDim.x;
    const int v = b * 0x01010101;
    const h2 lo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 0.000244140625f, false);
    const h2 hi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, 0.000244140625f, true);
    out[b * 4 + 0] = __builtin_bit_cast(unsigned short, lo[0]);
    out[b * 4 + 1] = __builtin_bit_cast(unsigned short, lo[1]);
    out[b * 4 + 2] = __builtin_bit_cast(unsigned short, hi[0]);
    out[b * 4 + 3] = __builtin_bit_cast(unsigned short, hi[1]);
}
__global__ void enc(unsigned char* out) {                // thread t: fp16 bit pattern t as the lo part
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const float f = (float)__builtin_bit_cast(_Float16, (unsigned short)t) * 4096.0f;
    const float c = fminf(fmaxf(f, -448.0f), 448.0f);    // (NaN: fmaxf / fminf return the other operand -> -448: checked below as "any NaN handling")
    int e = 0;
    e = __builtin_amdgcn_cvt_pk_fp8_f32(c, c, e, false);
    out[t] = (unsigned char)(e & 0xff);
}
static float e4m3fn(int b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
static unsigned short f32_to_f16_bits(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float f16_bits_to_f32(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
int main() {
    unsigned short* d; hipMalloc(&d, 256 * 4 * 2);
    hipLaunchKernelGGL(dec, dim3(1), dim3(256), 0, 0, d);
    unsigned short h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 256; ++b) {
        const float want = e4m3fn(b) * 0.000244140625f;
        for (int j = 0; j < 4; ++j) {
            const unsigned short got = h[b * 4 + j], w = f32_to_f16_bits(want);
            const bool ok = std::isnan(want) ? std::isnan(f16_bits_to_f32(got)) : got == w;
            if (!ok) { if (bad < 12) printf("decode byte %02x pos %d: got %04x (%g) want %04x (%g)\n", b, j, got, f16_bits_to_f32(got), w, want); ++bad; }
        }
    }
    printf("fp8 probe, decode: %d mismatches of 1024\n", bad);
    unsigned char* de; hipMalloc(&de, 65536);
    hipLaunchKernelGGL(enc, dim3(256), dim3(256), 0, 0, de);
    std::vector<unsigned char> he(65536); hipMemcpy(he.data(), de, 65536, hipMemcpyDeviceToHost);
    int bad2 = 0;
    for (int t = 0; t < 65536; ++t) {
        float f = f16_bits_to_f32((unsigned short)t) * 4096.0f;
        if (std::isnan(f)) continue;
        f = fminf(fmaxf(f, -448.0f), 448.0f);
        // nearest e4m3fn value, ties to even mantissa; sign of zero kept
        int best = -1; float bd = INFINITY;
        for (int b = 0; b < 256; ++b) {
            const float v = e4m3fn(b);
            if (std::isnan(v) || (std::signbit(v) != std::signbit(f))) continue;
            const float dd = fabsf(v - f);
            if (dd < bd || (dd == bd && (b & 1) == 0)) { bd = dd; best = b; }
        }
        if (he[t] != best) { if (bad2 < 12) printf("encode lo %04x (x 4096 = %g): got %02x (%g) want %02x (%g)\n", t, f, he[t], e4m3fn(he[t]), best, e4m3fn(best)); ++bad2; }
    }
    printf("fp8 probe, encode: %d mismatches of 65536 fp16 patterns\n", bad2);
    return (bad || bad2) != 0;
}
