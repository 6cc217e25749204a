// Fused FFN of the MV2D decoder layer (mmcv FFN: x + W2.relu(W1.x + b1) + b2, 256 -> 2048 -> 256; configured at
// configs/mv2d/exp/*:78-79, called from mmcv BaseTransformerLayer inside MU/petr_transformer.py:269-311), exact fp32 MFMA.
//
// Round-1 profile: the two separate GEMMs were L2->CU bandwidth bound (every 16x16 output tile re-fetched 32 KB of
// operands: 164 MB for 0.63 GFLOP) and the [M,2048] hidden activations made a round trip through memory.
// Here one block owns 32 query rows x one 64-wide slice of the hidden layer:
//   phase 1: H = relu(X[32,256] . W1[slice]^T + b1[slice])      (X staged once in LDS, W1 slice straight to VGPRs)
//   phase 2: P = H[32,64] . W2[:, slice]^T                      (H stays in LDS, W2 slice straight to VGPRs)
// and writes the partial sum P as slab `slice`; the following row kernel adds the 32 slabs, b2 and the residual in a
// fixed order (deterministic) and applies LayerNorm.  Operand traffic drops to 160 KB per 2.1 MFLOP block (51 MB).
// The weights are static and are read FRAGMENT-MAJOR (mv2d_ffn_pack_weights): the float4 a lane needs for (tile, k chunk c) sits at
// [...][c][lane], so every weight load of a wave is one contiguous 1 KB instead of 16 rows x 64 B.
#include "common.h"

namespace {

constexpr int C = 256, HS = 64, BR = 32;      // channels, hidden slice, rows per block

__device__ __forceinline__ int xoff(int row, int slot) { return row * C + ((slot ^ (row & 15)) << 2); }        // floats, 64 slots/row
__device__ __forceinline__ int hoff(int row, int slot) { return row * HS + ((slot ^ (row & 15)) << 2); }       // floats, 16 slots/row

__global__ __launch_bounds__(256, 2) void ffn_fused_kernel(const float* __restrict__ X, const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, float* __restrict__ slabs, int M, int hidden) {
    __shared__ __attribute__((aligned(16))) float xs[BR * C];
    __shared__ __attribute__((aligned(16))) float hs_[BR * HS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int slice = blockIdx.x, m0 = blockIdx.y * BR;
    const int rt = wave >> 1, half = wave & 1;
    // ---- issue everything that does not depend on LDS: X rows (coalesced), W1 slice fragments
    float4 xr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, row = idx >> 6, slot = idx & 63;
        xr[i] = *reinterpret_cast<const float4*>(X + (long long)min(m0 + row, M - 1) * C + slot * 4);
    }
    float4 w1[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // W1p[slice][tile (4)][c (16)][lane][4]
        const float* wp = W1 + ((long long)((slice * 4 + 2 * half + t) * 16) * 64 + lane) * 4;
#pragma unroll
        for (int c = 0; c < 16; ++c) w1[t][c] = *reinterpret_cast<const float4*>(wp + c * 256);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, row = idx >> 6, slot = idx & 63;
        *reinterpret_cast<float4*>(xs + xoff(row, slot)) = xr[i];
    }
    __syncthreads();
    // W2 slice fragments: in flight while phase 1 computes
    float4 w2[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        // W2p[slice][tile (16)][c (4)][lane][4]
        const float* wp = W2 + ((long long)((slice * 16 + 8 * half + t) * 4) * 64 + lane) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) w2[t][c] = *reinterpret_cast<const float4*>(wp + c * 256);
    }
    // ---- phase 1: two 16x16 tiles of H per wave, K = 256 (k spread over (MFMA step, lane group) identically for A and W)
    f32x4_t h[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(xs + xoff(rt * 16 + fr, 4 * c + fg));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w1[t][c].x, h[t], 0, 0, 0);
            h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w1[t][c].y, h[t], 0, 0, 0);
            h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w1[t][c].z, h[t], 0, 0, 0);
            h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w1[t][c].w, h[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = (2 * half + t) * 16 + fr;
        const float bb = b1[slice * HS + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + 4 * fg + r;
            hs_[hoff(row, col >> 2) + (col & 3)] = relu_f(h[t][r] + bb);
        }
    }
    __syncthreads();
    // ---- phase 2: eight 16x16 tiles of P per wave, K = 64
    f32x4_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(hs_ + hoff(rt * 16 + fr, 4 * c + fg));
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w2[t][c].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w2[t][c].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w2[t][c].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w2[t][c].w, acc[t], 0, 0, 0);
        }
    }
    float* out = slabs + (long long)slice * M * C;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int n = (8 * half + t) * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + rt * 16 + 4 * fg + r;
            if (m < M) out[(long long)m * C + n] = acc[t][r];
        }
    }
}

// W1 [hidden,256], W2 [256,hidden] (nn.Linear layout) -> the fragment-major copies the kernel reads:
//   W1p[slice][tile][c][fr + 16 fg][e] = W1[64 slice + 16 tile + fr][16 c + 4 fg + e]          (4 tiles, 16 chunks)
//   W2p[slice][tile][c][fr + 16 fg][e] = W2[16 tile + fr][64 slice + 16 c + 4 fg + e]          (16 tiles, 4 chunks)
__global__ void ffn_pack_kernel(const float* __restrict__ W1, const float* __restrict__ W2, float* __restrict__ W1p, float* __restrict__ W2p, int hidden) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one float4 per thread and matrix
    const long long total = (long long)hidden * C / 4;
    if (idx >= total) return;
    const int lane = idx & 63, fr = lane & 15, fg = lane >> 4;
    long long t = idx >> 6;
    {   // W1p
        const int c = (int)(t % 16), tile = (int)((t / 16) % 4), slice = (int)(t / 64);
        *reinterpret_cast<float4*>(W1p + idx * 4) = *reinterpret_cast<const float4*>(W1 + (long long)(64 * slice + 16 * tile + fr) * C + 16 * c + 4 * fg);
    }
    {   // W2p
        const int c = (int)(t % 4), tile = (int)((t / 4) % 16), slice = (int)(t / 64);
        *reinterpret_cast<float4*>(W2p + idx * 4) = *reinterpret_cast<const float4*>(W2 + (long long)(16 * tile + fr) * hidden + 64 * slice + 16 * c + 4 * fg);
    }
}

}  // namespace

extern "C" int mv2d_ffn_pack_weights(const float* W1, const float* W2, float* W1p, float* W2p, int hidden, void* stream) {
    MV2D_CHECK_ARG(W1 && W2 && W1p && W2p && hidden > 0 && (hidden % HS) == 0, "mv2d_ffn_pack_weights: bad args");
    const long long total = (long long)hidden * C / 4;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W1, W2, W1p, W2p, hidden);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_ffn_fused(const float* X, const float* W1, const float* b1, const float* W2, float* slabs, int M, int hidden,
                              void* stream) {
    MV2D_CHECK_ARG(X && W1 && b1 && W2 && slabs, "mv2d_ffn_fused: null pointer");
    MV2D_CHECK_ARG(hidden > 0 && (hidden % HS) == 0, "mv2d_ffn_fused: hidden must be a multiple of 64");
    if (M == 0) return MV2D_OK;
    hipLaunchKernelGGL(ffn_fused_kernel, dim3(hidden / HS, cdiv(M, BR)), dim3(256), 0, (hipStream_t)stream, X, W1, b1, W2, slabs, M, hidden);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
