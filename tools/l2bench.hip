// standalone microbenchmark (hipcc --offload-arch=gfx950 -O3 -o l2bench tools/l2bench.hip): per-CU bandwidth of streaming an L2-resident 2.5 MB region
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// every block streams the same `bytes` region (L2 resident) `iters` times: NW waves per block, DEPTH 1 KB loads in flight per wave
template <int DEPTH>
__global__ void stream_kernel(const uint4* __restrict__ p, long long n16, int iters, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const long long per_wave = n16 / nw;            // uint4 elements per wave
    const uint4* base = p + wave * per_wave + lane;
    for (int it = 0; it < iters; ++it) {
        for (long long i = 0; i + 64 * DEPTH <= per_wave; i += 64 * DEPTH) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = base[i + 64 * d];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = acc.x;
}
int main() {
    const long long bytes = 2560 * 1024;           // 2.5 MB like the PE weights
    uint4* d; unsigned int* sink;
    hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20;
    for (int blocks : {8, 256}) for (int nw : {4, 8, 16}) for (int depth : {4, 8, 16}) {
        auto launch = [&]() {
            if (depth == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(blocks), dim3(64 * nw), 0, 0, d, bytes / 16, iters, sink);
            else if (depth == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(64 * nw), 0, 0, d, bytes / 16, iters, sink);
            else hipLaunchKernelGGL(stream_kernel<16>, dim3(blocks), dim3(64 * nw), 0, 0, d, bytes / 16, iters, sink);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %3d waves/block %2d loads in flight/wave %2d: %.1f GB/s per CU (%.2f TB/s total)\n", blocks, nw, depth,
               (double)bytes * iters / (ms * 1e-3) / 1e9, (double)bytes * iters * blocks / (ms * 1e-3) / 1e12);
    }
    return 0;
}
