// Fused FFN of the MV2D decoder layer on the bf16 matrix cores in split precision ("bf16x3"), same block structure and slab
// output as ffn.hip (mmcv FFN 256 -> 2048 -> 256, configs/mv2d/exp/*:78-79; MU/petr_transformer.py:269-311).
//
// The exact-fp32 kernel (ffn.hip) is bound by v_mfma_f32_16x16x4_f32: 256 of them per wave at 32 cycles each, two blocks per
// CU.  Here every fp32 operand is carried as a bf16 pair x = x_hi + x_lo (x_hi = bf16(x), x_lo = bf16(x - x_hi)) and a
// product is a_hi.w_hi + a_lo.w_hi + a_hi.w_lo on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: every partial product is
// exact in fp32, the dropped a_lo.w_lo term is ~2^-18 relative -> ~1e-5 relative error (plain bf16: 4e-3), at 3/16 of the
// matrix-pipe time.  The weights are split once at load time (mv2d_split_bf16x2) and stored fragment-major (mv2d_pack_wfrag_bf16),
// X is split when it is staged into LDS, the hidden
// activations when they are written to LDS.  Both phases run swapped (D^T = W.A^T), so a lane always ends with 4 consecutive
// columns of one row: 8-byte LDS writes of H, 16-byte stores of the slab.
#include "common.h"

namespace {

constexpr int C = 256, HS = 64;               // channels, hidden slice (rows per block: template parameter, 8 per wave)
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
union Frag { uint4 u; mfma_bf16x8 v; };

// bf16 LDS images: X [32][256] (512 B rows, 32 slots of 16 B), H [32][64] (128 B rows, 8 slots); slot ^= row bits
__device__ __forceinline__ int xoff(int row, int slot) { return row * (C * 2) + ((slot ^ (row & 15)) << 4); }
__device__ __forceinline__ int hoff(int row, int slot) { return row * (HS * 2) + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ void split2(float a, float b, unsigned int& hi, unsigned int& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// G consecutive 64-wide hidden slices per block, accumulated in registers: hidden/(64 G) slabs instead of hidden/64.  With many rows
// (a batch of samples) there are enough blocks anyway, and the slab round trip through memory ([32, M, 256] fp32 written here, read
// by the row kernel that follows) was as expensive as the FFN itself.
// BR = 32 rows (4 waves) or 64 rows (8 waves) per block: the waves of the row tiles share the weight fragments of their column half
// through the L1, so 64-row blocks halve the L2 -> CU weight traffic -- which turned out not to be what limits the kernel (slower).
template <int G, int BR>
__global__ __launch_bounds__(BR * 8, (G == 1 && BR == 32) ? 2 : 1) void ffn_x3_kernel(const float* __restrict__ X, const unsigned short* __restrict__ W1h,
                                                        const unsigned short* __restrict__ W1l, const float* __restrict__ b1,
                                                        const unsigned short* __restrict__ W2h, const unsigned short* __restrict__ W2l,
                                                        float* __restrict__ slabs, int M, int hidden) {
    __shared__ __attribute__((aligned(16))) unsigned char xh[BR * C * 2], xl[BR * C * 2], hh[2][BR * HS * 2], hl[2][BR * HS * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int slab = blockIdx.x, m0 = blockIdx.y * BR;
    const int rt = wave >> 1, half = wave & 1;
    const int nt = hidden / 16;
    // ---- issue everything that does not depend on LDS: X rows (coalesced), W1 fragments (hi, lo) of the first slice
    float4 xr[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + (BR * 8) * i, row = idx >> 5, slot = idx & 31;       // 32 slots of 8 floats per row
        const float* xp = X + (long long)min(m0 + row, M - 1) * C + slot * 8;
        xr[i][0] = *reinterpret_cast<const float4*>(xp);
        xr[i][1] = *reinterpret_cast<const float4*>(xp + 4);
    }
    Frag w1h[2][8], w1l[2][8];
    auto load_w1 = [&](int slice) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // fragment-major W1 [hidden,256]: [k-step (8)][hidden/16 column tiles][lane][8]
            const int tile = slice * 4 + 2 * half + t;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const long long wo = (((long long)s * nt + tile) * 64 + lane) * 8;
                w1h[t][s].u = *reinterpret_cast<const uint4*>(W1h + wo);
                w1l[t][s].u = *reinterpret_cast<const uint4*>(W1l + wo);
            }
        }
    };
    load_w1(slab * G);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + (BR * 8) * i, row = idx >> 5, slot = idx & 31;
        uint4 h4, l4;
        split2(xr[i][0].x, xr[i][0].y, h4.x, l4.x); split2(xr[i][0].z, xr[i][0].w, h4.y, l4.y);
        split2(xr[i][1].x, xr[i][1].y, h4.z, l4.z); split2(xr[i][1].z, xr[i][1].w, h4.w, l4.w);
        *reinterpret_cast<uint4*>(xh + xoff(row, slot)) = h4;
        *reinterpret_cast<uint4*>(xl + xoff(row, slot)) = l4;
    }
    __syncthreads();
    f32x4_t a0[8], a1[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { a0[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; a1[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int slice = slab * G + g;
        unsigned char* hhg = hh[g & 1];
        unsigned char* hlg = hl[g & 1];
        // W2 slice fragments: in flight while phase 1 computes.  (interleaved so that a lane ends with 4 consecutive columns per
        // tile pair; see the store below)
        Frag w2h[8][2], w2l[8][2];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // fragment-major W2 [256,hidden]: [k-step (hidden/32)][16 column tiles][lane][8]
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const long long wo = (((long long)(slice * 2 + s) * 16 + 8 * half + t) * 64 + lane) * 8;
                w2h[t][s].u = *reinterpret_cast<const uint4*>(W2h + wo);
                w2l[t][s].u = *reinterpret_cast<const uint4*>(W2l + wo);
            }
        }
        // ---- phase 1 (swapped): lane (fr, fg) ends with hidden columns 4fg..4fg+3 of row fr for each of its two 16-wide tiles
        f32x4_t h0[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}}, h1[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            Frag ah, al;
            ah.u = *reinterpret_cast<const uint4*>(xh + xoff(rt * 16 + fr, 4 * s + fg));
            al.u = *reinterpret_cast<const uint4*>(xl + xoff(rt * 16 + fr, 4 * s + fg));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                h0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[t][s].v, ah.v, h0[t], 0, 0, 0);
                h1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[t][s].v, al.v, h1[t], 0, 0, 0);
                h1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1l[t][s].v, ah.v, h1[t], 0, 0, 0);
            }
        }
        if (g + 1 < G) load_w1(slice + 1);               // the next slice's W1: in flight during the epilogue and phase 2
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = (2 * half + t) * 16 + 4 * fg;                               // first of 4 hidden columns (local to the slice)
            const float4 bb = *reinterpret_cast<const float4*>(b1 + slice * HS + col);
            const float v0 = relu_f(h0[t][0] + h1[t][0] + bb.x), v1 = relu_f(h0[t][1] + h1[t][1] + bb.y);
            const float v2 = relu_f(h0[t][2] + h1[t][2] + bb.z), v3 = relu_f(h0[t][3] + h1[t][3] + bb.w);
            uint2 hi, lo;
            split2(v0, v1, hi.x, lo.x); split2(v2, v3, hi.y, lo.y);
            const int row = rt * 16 + fr, off = hoff(row, col >> 3) + (col & 4) * 2;
            *reinterpret_cast<uint2*>(hhg + off) = hi;
            *reinterpret_cast<uint2*>(hlg + off) = lo;
        }
        __syncthreads();     // (the H buffers alternate: a wave can only pass this barrier after every wave finished phase 2 of g - 1)
        // ---- phase 2 (swapped): eight 16-column tiles per wave, K = 64
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Frag gh, gl;
            gh.u = *reinterpret_cast<const uint4*>(hhg + hoff(rt * 16 + fr, 4 * s + fg));
            gl.u = *reinterpret_cast<const uint4*>(hlg + hoff(rt * 16 + fr, 4 * s + fg));
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                a0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2h[t][s].v, gh.v, a0[t], 0, 0, 0);
                a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2h[t][s].v, gl.v, a1[t], 0, 0, 0);
                a1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2l[t][s].v, gh.v, a1[t], 0, 0, 0);
            }
        }
    }
    const int m = m0 + rt * 16 + fr;
    if (m < M) {
        float* out = slabs + ((long long)slab * M + m) * C + 128 * half + 4 * fg;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            *reinterpret_cast<float4*>(out + 16 * t) = make_float4(a0[t][0] + a1[t][0], a0[t][1] + a1[t][1], a0[t][2] + a1[t][2], a0[t][3] + a1[t][3]);
    }
}

}  // namespace

extern "C" int mv2d_ffn_fused_x3(const float* X, const void* W1hi, const void* W1lo, const float* b1, const void* W2hi, const void* W2lo,
                                 float* slabs, int M, int hidden, int slices_per_block, void* stream) {
    MV2D_CHECK_ARG(X && W1hi && W1lo && b1 && W2hi && W2lo && slabs, "mv2d_ffn_fused_x3: null pointer");
    MV2D_CHECK_ARG(hidden > 0 && (hidden % HS) == 0, "mv2d_ffn_fused_x3: hidden must be a multiple of 64");
    MV2D_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W1hi & 15) == 0 && ((uintptr_t)W1lo & 15) == 0 && ((uintptr_t)W2hi & 15) == 0 &&
                       ((uintptr_t)W2lo & 15) == 0 && ((uintptr_t)slabs & 15) == 0 && ((uintptr_t)b1 & 15) == 0,
                   "mv2d_ffn_fused_x3: operands must be 16-byte aligned");
    const int G = slices_per_block;
    MV2D_CHECK_ARG((G == 1 || G == 2 || G == 4) && (hidden / HS) % G == 0, "mv2d_ffn_fused_x3: slices_per_block must be 1, 2 or 4 and divide hidden/64");
    if (M == 0) return MV2D_OK;
    const unsigned short *w1h = (const unsigned short*)W1hi, *w1l = (const unsigned short*)W1lo, *w2h = (const unsigned short*)W2hi, *w2l = (const unsigned short*)W2lo;
    static const int br_env = getenv("MV2D_FFN_BR") ? atoi(getenv("MV2D_FFN_BR")) : 0;
    const int br = br_env ? br_env : 32;            // measured at M = 1800: 64-row blocks 0.69 vs 0.658 ms per decoder pass -> experiment switch only
    if (G == 1 && br == 64) {
        hipLaunchKernelGGL((ffn_x3_kernel<1, 64>), dim3(hidden / HS, cdiv(M, 64)), dim3(512), 0, (hipStream_t)stream, X, w1h, w1l, b1, w2h, w2l, slabs, M, hidden);
        MV2D_LAUNCH_CHECK();
        return MV2D_OK;
    }
    const dim3 grid(hidden / HS / G, cdiv(M, 32));
    if (G == 1) hipLaunchKernelGGL((ffn_x3_kernel<1, 32>), grid, dim3(256), 0, (hipStream_t)stream, X, w1h, w1l, b1, w2h, w2l, slabs, M, hidden);
    else if (G == 2) hipLaunchKernelGGL((ffn_x3_kernel<2, 32>), grid, dim3(256), 0, (hipStream_t)stream, X, w1h, w1l, b1, w2h, w2l, slabs, M, hidden);
    else hipLaunchKernelGGL((ffn_x3_kernel<4, 32>), grid, dim3(256), 0, (hipStream_t)stream, X, w1h, w1l, b1, w2h, w2l, slabs, M, hidden);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
