// What does HBM deliver for CHUNKED gathers?  A 384 MB region is read exactly once, in chunks of CH bytes visited in a random order; a wave has U KB in flight
// (U dwordx4 loads per lane issued before the first use), W waves per block, enough blocks for every CU to hold its maximum.  Prints TB/s per (CH, U, W).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(512) void gather(const char* __restrict__ src, const int* __restrict__ order, int chunk_kb, long long n_units, unsigned int* __restrict__ sink) {
    // unit = U KB = the piece one wave requests at once; a chunk of chunk_kb KB holds chunk_kb / U units (chunk_kb >= U) or a unit spans U / chunk_kb chunks
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (long long)gridDim.x * (blockDim.x >> 6);
    unsigned int acc = 0;
    for (long long u = wave; u < n_units; u += nw) {
        u32x4 v[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const long long kb = u * U + i;                       // the kb-th KB of the walk
            const long long chunk = kb / chunk_kb, off = kb - chunk * chunk_kb;
            const long long base = (long long)order[chunk] * chunk_kb + off;
            v[i] = *reinterpret_cast<const u32x4*>(src + base * 1024 + lane * 16);
        }
#pragma unroll
        for (int i = 0; i < U; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const long long MB = 384, total_kb = MB * 1024;
    char* src; hipMalloc(&src, total_kb * 1024); hipMemset(src, 1, total_kb * 1024);
    unsigned int* sink; hipMalloc(&sink, 4);
    int* d_order; hipMalloc(&d_order, total_kb * 4 * 2);           // up to half-KB chunks
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::mt19937 rng(1);
    for (int chunk_kb : {1, 2, 4, 8, 16, 64}) {
        const long long nchunk = total_kb / chunk_kb;
        std::vector<int> order(nchunk);
        for (long long i = 0; i < nchunk; ++i) order[i] = (int)i;
        std::shuffle(order.begin(), order.end(), rng);
        hipMemcpy(d_order, order.data(), nchunk * 4, hipMemcpyHostToDevice);
        for (int W : {4, 8}) {
            for (int U : {4, 8, 16}) {
                const long long n_units = total_kb / U;
                const int blocks = 256 * (W == 4 ? 8 : 4) * 2;
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEventRecord(e0);
                    if (U == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(64 * W), 0, 0, src, d_order, chunk_kb, n_units, sink);
                    if (U == 8) hipLaunchKernelGGL(gather<8>, dim3(blocks), dim3(64 * W), 0, 0, src, d_order, chunk_kb, n_units, sink);
                    if (U == 16) hipLaunchKernelGGL(gather<16>, dim3(blocks), dim3(64 * W), 0, 0, src, d_order, chunk_kb, n_units, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("chunk %3d KB  waves/block %d  in flight per wave %2d KB : %7.1f us = %5.2f TB/s\n", chunk_kb, W, U, best * 1e3, MB * 1.048576e6 / (best * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
