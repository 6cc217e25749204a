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

constexpr int C = 256, HS = 64;               // channels, hidden slice
typedef q16x8_t mfma_bf16x8;      // common.h "q16": fp16 pairs since round 5
union Frag { uint4 u; mfma_bf16x8 v; };

// bf16 LDS images: X [32][256] (512 B rows, 32 slots of 16 B), H [32][64] (128 B rows, 8 slots); slot ^= row bits
__device__ __forceinline__ int xoff(int row, int slot) { return row * (C * 2) + ((slot ^ (row & 15)) << 4); }
__device__ __forceinline__ int hoff(int row, int slot) { return row * (HS * 2) + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ void split2(float a, float b, unsigned int& hi, unsigned int& lo) {
    split_q16x2(a, b, hi, lo);
}

// G consecutive 64-wide hidden slices per block, accumulated in registers: hidden/(64 G) slabs instead of hidden/64 (experiment switch:
// with one block per CU it loses what the smaller slab round trip saves).
//
// Work split inside a block (32 rows = 2 row tiles, 4 waves): a wave owns COLUMN tiles -- hidden tile `wave` in phase 1, output tiles
// 4 wave .. 4 wave + 3 in phase 2 -- for BOTH row tiles, so every weight fragment is fetched by exactly one wave and used for all 32
// rows.  (First version: waves = (row tile, column half); each fragment was fetched twice and used for 16 rows, and the kernel sat
// at the per-CU L1 fill rate: 93 % of a block's time was waiting for weights.)
// Row tiles per block RTB (round 5): 2 (32 rows) for small launches, 3 (48 rows) for batches -- 4800 rows are 600 blocks of 32 rows = 1.17 rounds of
// the 512 two-per-CU slots, i.e. two rounds (49.8 us); 400 blocks of 48 rows run in one (42.8 us; 2400 rows: 31.0 -> 27.8; 320 rows: 17.3 -> 22.3, so
// small launches keep 32).  A row's arithmetic does not depend on RTB (ONE accumulator per output tile, slices and products in a fixed order), so
// the choice by row count keeps "a sample's result does not depend on its batch" bit for bit.
template <int G, int RTB>
__global__ __launch_bounds__(256, 2) void ffn_x3_kernel(const float* __restrict__ X, const unsigned short* __restrict__ W1h,
                                                        const unsigned short* __restrict__ W1l, const float* __restrict__ b1,
                                                        const unsigned short* __restrict__ W2h, const unsigned short* __restrict__ W2l,
                                                        float* __restrict__ slabs, int M, int hidden) {
    constexpr int BR = 16 * RTB, NXR = BR / 8;           // rows per block, X staging rounds
    constexpr int NHB = G > 1 ? 2 : 1;                   // H buffers (they alternate between the slices of a block)
    __shared__ __attribute__((aligned(16))) unsigned char xh[BR * C * 2], xl[BR * C * 2], hh[NHB][BR * HS * 2], hl[NHB][BR * HS * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    // (natural block order: with 4 slabs and x fastest, XCD k only ever sees slab k % 4, i.e. a quarter of the weights; the XCD-chunked order
    //  of linear_x3 / heads -- all slabs of a row block on one XCD -- fetched 71 instead of 29 MB per launch here, round 4)
    const int slab = blockIdx.x, m0 = blockIdx.y * BR;
    const int nt = hidden / 16;
    // ---- issue everything that does not depend on LDS: X rows (coalesced), W1 fragments (hi, lo) of the first slice
    float4 xr[NXR][2];
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
        const int idx = tid + 256 * i, row = idx >> 5, slot = idx & 31;            // 32 slots of 8 floats per row
        const float* xp = X + (long long)min(m0 + row, M - 1) * C + slot * 8;
        xr[i][0] = *reinterpret_cast<const float4*>(xp);
        xr[i][1] = *reinterpret_cast<const float4*>(xp + 4);
    }
    Frag w1h[8], w1l[8];
    auto load_w1 = [&](int slice) {
        // fragment-major W1 [hidden,256]: [k-step (8)][hidden/16 column tiles][lane][8]
        const int tile = slice * 4 + wave;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const long long wo = (((long long)s * nt + tile) * 64 + lane) * 8;
            w1h[s].u = *reinterpret_cast<const uint4*>(W1h + wo);
            w1l[s].u = *reinterpret_cast<const uint4*>(W1l + wo);
        }
    };
    load_w1(slab * G);
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
        const int idx = tid + 256 * i, row = idx >> 5, slot = idx & 31;
        uint4 h4, l4;
        split2(xr[i][0].x, xr[i][0].y, h4.x, l4.x); split2(xr[i][0].z, xr[i][0].w, h4.y, l4.y);
        split2(xr[i][1].x, xr[i][1].y, h4.z, l4.z); split2(xr[i][1].z, xr[i][1].w, h4.w, l4.w);
        *reinterpret_cast<uint4*>(xh + xoff(row, slot)) = h4;
        *reinterpret_cast<uint4*>(xl + xoff(row, slot)) = l4;
    }
    __syncthreads();
    // ONE fp32 accumulator per output tile for the hi.hi product and the two correction products (rounds 1-4 kept the corrections apart: 32 more
    // registers, which 3 row tiles per block do not have; same 6e-7 against fp64 either way)
    constexpr int NACC = 1;
    f32x4_t acc[NACC][RTB][4];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < RTB; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[n][r][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int slice = slab * G + g;
        unsigned char* hhg = hh[g & (NHB - 1)];
        unsigned char* hlg = hl[g & (NHB - 1)];
        // W2 slice fragments of this wave's four output tiles: in flight while phase 1 computes
        Frag w2h[4][2], w2l[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // fragment-major W2 [256,hidden]: [k-step (hidden/32)][16 column tiles][lane][8]
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const long long wo = (((long long)(slice * 2 + s) * 16 + 4 * wave + t) * 64 + lane) * 8;
                w2h[t][s].u = *reinterpret_cast<const uint4*>(W2h + wo);
                w2l[t][s].u = *reinterpret_cast<const uint4*>(W2l + wo);
            }
        }
        // ---- phase 1 (swapped): lane (fr, fg) ends with hidden columns 16 wave + 4fg .. + 3 of rows 16 r + fr
        f32x4_t h0[RTB], h1[RTB];
#pragma unroll
        for (int r = 0; r < RTB; ++r) { h0[r] = f32x4_t{0.f, 0.f, 0.f, 0.f}; h1[r] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int r = 0; r < RTB; ++r) {
                Frag ah, al;
                ah.u = *reinterpret_cast<const uint4*>(xh + xoff(r * 16 + fr, 4 * s + fg));
                al.u = *reinterpret_cast<const uint4*>(xl + xoff(r * 16 + fr, 4 * s + fg));
                h0[r] = mfma_q16_16x16x32(w1h[s].v, ah.v, h0[r], 0, 0, 0);
                h1[r] = mfma_q16_16x16x32(w1h[s].v, al.v, h1[r], 0, 0, 0);
                h1[r] = mfma_q16_16x16x32(w1l[s].v, ah.v, h1[r], 0, 0, 0);
            }
        }
        if (g + 1 < G) load_w1(slice + 1);               // the next slice's W1: in flight during the epilogue and phase 2
        {
            const int col = wave * 16 + 4 * fg;                                         // first of 4 hidden columns (local to the slice)
            const float4 bb = *reinterpret_cast<const float4*>(b1 + slice * HS + col);
#pragma unroll
            for (int r = 0; r < RTB; ++r) {
                const float v0 = relu_f(h0[r][0] + h1[r][0] + bb.x), v1 = relu_f(h0[r][1] + h1[r][1] + bb.y);
                const float v2 = relu_f(h0[r][2] + h1[r][2] + bb.z), v3 = relu_f(h0[r][3] + h1[r][3] + bb.w);
                uint2 hi, lo;
                split2(v0, v1, hi.x, lo.x); split2(v2, v3, hi.y, lo.y);
                const int row = r * 16 + fr, off = hoff(row, col >> 3) + (col & 4) * 2;
                *reinterpret_cast<uint2*>(hhg + off) = hi;
                *reinterpret_cast<uint2*>(hlg + off) = lo;
            }
        }
        __syncthreads();     // (the H buffers alternate: a wave can only pass this barrier after every wave finished phase 2 of g - 1)
        // ---- phase 2 (swapped): four 16-column tiles x two row tiles per wave, K = 64
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int r = 0; r < RTB; ++r) {
                Frag gh, gl;
                gh.u = *reinterpret_cast<const uint4*>(hhg + hoff(r * 16 + fr, 4 * s + fg));
                gl.u = *reinterpret_cast<const uint4*>(hlg + hoff(r * 16 + fr, 4 * s + fg));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0][r][t] = mfma_q16_16x16x32(w2h[t][s].v, gh.v, acc[0][r][t], 0, 0, 0);
                    acc[NACC - 1][r][t] = mfma_q16_16x16x32(w2h[t][s].v, gl.v, acc[NACC - 1][r][t], 0, 0, 0);
                    acc[NACC - 1][r][t] = mfma_q16_16x16x32(w2l[t][s].v, gh.v, acc[NACC - 1][r][t], 0, 0, 0);
                }
            }
        }
    }
    // The MFMA layout gives a lane 4 columns of 16 different rows (a store instruction would touch 16 rows x 64 bytes): the wave's
    // 32 x 64 tile goes through its quarter of the X images (free since the barrier before the last phase 2) and is stored row-major,
    // 4 rows x 256 contiguous bytes per instruction.
    float4* ot = reinterpret_cast<float4*>(wave < 2 ? xh : xl) + (wave & 1) * (BR * 16);          // 8 KB per wave (RTB = 2)
    static_assert(RTB >= 1 && RTB <= 4, "output staging: a wave's BR x 64 tile in its quarter of the X images");
#pragma unroll
    for (int r = 0; r < RTB; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 16 * r + fr;
            ot[row * 16 + ((4 * t + fg) ^ (row & 15))] = NACC == 2
                ? make_float4(acc[0][r][t][0] + acc[NACC - 1][r][t][0], acc[0][r][t][1] + acc[NACC - 1][r][t][1],
                              acc[0][r][t][2] + acc[NACC - 1][r][t][2], acc[0][r][t][3] + acc[NACC - 1][r][t][3])
                : make_float4(acc[0][r][t][0], acc[0][r][t][1], acc[0][r][t][2], acc[0][r][t][3]);
        }
    __builtin_amdgcn_wave_barrier();                         // read back by the same wave only
#pragma unroll
    for (int k = 0; k < BR / 4; ++k) {
        const int row = 4 * k + (lane >> 4), c4 = lane & 15, m = m0 + row;
        const float4 v = ot[row * 16 + (c4 ^ (row & 15))];
        if (m < M) *reinterpret_cast<float4*>(slabs + ((long long)slab * M + m) * C + 64 * wave + 4 * c4) = v;
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
    MV2D_CHECK_ARG((G == 1 || G == 2 || G == 4 || G == 8) && (hidden / HS) % G == 0, "mv2d_ffn_fused_x3: slices_per_block must be 1, 2, 4 or 8 and divide hidden/64");
    if (M == 0) return MV2D_OK;
    const unsigned short *w1h = (const unsigned short*)W1hi, *w1l = (const unsigned short*)W1lo, *w2h = (const unsigned short*)W2hi, *w2l = (const unsigned short*)W2lo;
    const int rtb = M > 1024 ? 3 : 2;
    const dim3 grid(hidden / HS / G, cdiv(M, 16 * rtb));
#define MV2D_FFN(G_, R_) hipLaunchKernelGGL((ffn_x3_kernel<G_, R_>), grid, dim3(256), 0, (hipStream_t)stream, X, w1h, w1l, b1, w2h, w2l, slabs, M, hidden)
    if (rtb == 3) { if (G == 1) MV2D_FFN(1, 3); else if (G == 2) MV2D_FFN(2, 3); else if (G == 4) MV2D_FFN(4, 3); else MV2D_FFN(8, 3); }
    else { if (G == 1) MV2D_FFN(1, 2); else if (G == 2) MV2D_FFN(2, 2); else if (G == 4) MV2D_FFN(4, 2); else MV2D_FFN(8, 2); }
#undef MV2D_FFN
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
