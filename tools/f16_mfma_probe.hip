// Probe (gfx950): does v_mfma_f32_16x16x32_f16 / v_mfma_f32_16x16x16_f16 keep fp16 SUBNORMAL inputs under hipcc's default kernel mode?
// The split-precision key rows (x ~ hi + lo, both fp16) rely on it: lo = x - hi is subnormal for |x| < 0.25.
//   hipcc --offload-arch=gfx950 -O2 tools/f16_mfma_probe.hip -o mv2d_amd/lib/f16_mfma_probe && mv2d_amd/lib/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) _Float16 h4;
typedef __attribute__((ext_vector_type(4))) float f4;

__global__ void probe(float a_val, float b_val, float* out) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[row = lane & 15][k = 8 * (lane >> 4) + i], B[k][col = lane & 15]: put one non-zero product on k = 0 for every (row, col)
    if ((lane >> 4) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    h4 a4, b4;
    for (int i = 0; i < 4; ++i) { a4[i] = (_Float16)0.f; b4[i] = (_Float16)0.f; }
    if ((lane >> 4) == 0) { a4[0] = (_Float16)a_val; b4[0] = (_Float16)b_val; }
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d, 0, 0, 0);
    if (lane == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)(_Float16)a_val; out[3] = (float)(_Float16)b_val; }
}

int main() {
    float* d;
    hipMalloc(&d, 16);
    const float cases[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1024.f}, {5.96e-8f, 4096.f}, {1024.f, 3.0e-6f}, {3.0e-6f, 3.0e-6f}, {6.0e-5f, 1.f}, {65504.f, 1.f}, {70000.f, 1.f}};
    int bad = 0;
    for (auto& cs : cases) {
        probe<<<1, 64>>>(cs[0], cs[1], d);
        float h[4];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double want = (double)h[2] * (double)h[3];
        const bool ok = std::fabs(h[0] - want) <= 1e-6 * std::fabs(want) && std::fabs(h[1] - want) <= 1e-6 * std::fabs(want);
        printf("a=%.4g (fp16 %.6g) b=%.4g (fp16 %.6g): mfma16x16x32 %.9g  mfma16x16x16 %.9g  expected %.9g  %s\n", cs[0], h[2], cs[1], h[3], h[0], h[1], want,
               (ok || !std::isfinite(want)) ? "ok" : "MISMATCH");
        if (!ok && std::isfinite(want)) ++bad;
    }
    printf(bad ? "f16 MFMA probe: %d mismatching cases (subnormal inputs flushed?)\n" : "f16 MFMA probe: subnormal fp16 inputs are kept (%d mismatches)\n", bad);
    return bad ? 1 : 0;
}
