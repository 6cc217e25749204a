// standalone microbenchmark (hipcc --offload-arch=gfx950 -O3 -o l2bench_mfma tools/l2bench_mfma.hip): a PE-like k-step (16 MFMAs + 4 x 1 KB weight fragments per wave, ring 3 steps ahead) with and without the loads
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
union Frag { uint4 u; bf16x8 v; };
// PE-like step: 4 x 1 KB weight fragments requested 3 steps ahead (4-slot ring), NM MFMAs per step on the oldest slot; 4 waves per block
template <int NM, bool LOADS>
__global__ __launch_bounds__(256, 1) void step_kernel(const uint4* __restrict__ p, long long n16, int iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long per_wave = n16 / 4;
    const uint4* base = p + wave * per_wave + lane;
    const long long nstep = per_wave / 256;                 // 4 fragments of 64 uint4 per step
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    Frag wq[4][4], a[4];
    for (int i = 0; i < 4; ++i) a[i].u = base[i * 64];
    for (int s = 0; s < 4; ++s) for (int j = 0; j < 4; ++j) wq[s][j].u = base[(s * 4 + j) * 64];
    for (int it = 0; it < iters; ++it) {
        for (long long st = 0; st + 8 <= nstep; st += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (LOADS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wq[(u + 3) & 3][j].u = base[((st + u + 3) * 4 + j) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[u][m & 3].v, a[(m >> 2) & 3].v, acc[m & 15], 0, 0, 0);
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 1.2345f) *sink = t;
}
int main() {
    const long long bytes = 2560 * 1024;
    uint4* d; float* sink;
    hipMalloc(&d, bytes + 65536); hipMemset(d, 0, bytes + 65536); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10;
    const long long steps = (bytes / 16 / 4 / 256) / 4 * 4 - 4;
    auto run = [&](const char* name, auto kern, int blocks, int nm, bool loads) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, bytes / 16, iters, sink); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, bytes / 16, iters, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us_step = ms * 1e3 / (iters * (double)steps);
        printf("%-28s blocks %3d: %.3f us per step (%.0f cycles @2.4GHz)%s\n", name, blocks, us_step, us_step * 2400,
               loads ? "" : "  [no loads]");
        (void)nm;
    };
    for (int blocks : {8, 138, 256}) {
        run("16 MFMA + 4 KB loads / step", step_kernel<16, true>, blocks, 16, true);
        run("16 MFMA, no loads", step_kernel<16, false>, blocks, 16, false);
        run("0 MFMA + 4 KB loads / step", step_kernel<0, true>, blocks, 0, true);
        run("32 MFMA + 4 KB loads / step", step_kernel<32, true>, blocks, 32, true);
        run("32 MFMA, no loads", step_kernel<32, false>, blocks, 32, false);
    }
    return 0;
}
