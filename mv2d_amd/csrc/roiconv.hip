// QueryGenerator shared conv: Conv2d(256, 256, 3, padding=1) + ReLU on the 7x7 RoI features, followed by AvgPool2d(7)
// (RH/utils/query_generator.py:298-304, 322-331, 352-358) — one block per RoI, conv + bias + ReLU + pooling fused.
//
// The implicit-GEMM route (gemm_bf16.hip, a_mode = 1) re-gathers every RoI row 9 times through L2 in 64x64 tiles, writes the
// [R*49, 256] fp32 conv output and needs a pooling launch.  Here the block keeps its RoI (49 cells x 256 channels key16 = 25 KB)
// in LDS once and every tap is just a remapped row index into it (out-of-range neighbours -> a zero row); the weights
// (256 x 2304 key16) are never staged: each wave streams the 64 output columns it owns as MFMA fragments straight from L2 through
// a 4-deep register ring.  The weights are static, so they are stored FRAGMENT-MAJOR (mv2d_pack_wfrag_bf16: [k-step][column
// tile][lane][8]): a fragment load is one contiguous 1 KB per wave instead of 16 rows x 64 B (row-major fragment loads
// measured 74 us for this kernel: the TA walks 16 half-used lines per instruction and the L1 thrashes).  After the single staging barrier the waves run free — no barrier, no LDS write in the 72-step k loop
// (fully unrolled so that hipcc keeps exact vmcnt counts for the ring).  The epilogue pools in registers + two shuffles and
// writes [R, 256] fp32: the 15 MB conv output and the avgpool launch disappear.
// k order = (tap, channel) like the implicit GEMM, so the conv sums are bit-identical; only the 49-term pooling sum is
// re-associated (fp32, ~1e-7).
#include "common.h"

namespace {

constexpr int C = 256, CELLS = 49, KT = 9 * C;
struct Frag { uint4 u; };                     // one MFMA operand fragment: 8 key16 values (common.h: fp16 since round 4)
typedef unsigned int cv_u32x4 __attribute__((ext_vector_type(4)));

// NR RoIs per block: with NR = 2 the wave's weight fragments feed 8 instead of 4 row tiles (the kernel is bound by streaming the
// 1.18 MB of conv weights through L2 / L1 once per block: 2.8 GB per 2400 RoIs at NR = 1).  Every RoI keeps its own four row tiles
// (rows 64 rl .. 64 rl + 48 of the block), so the arithmetic per RoI — and the result, bit for bit — does not depend on NR or on which
// RoI it is paired with.
template <int NR>
__global__ __launch_bounds__(256, 2) void roi_conv_pool_kernel(const unsigned short* __restrict__ feat, const unsigned short* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ out, int ld_out, int R) {
    // LDS rows: RoI rl at rows RS rl .. RS rl + 48, its zero row at RS rl + 49 (RS = 64 for NR = 2: a multiple of 16, so the chunk
    // swizzle and with it every fragment address of RoI 1 is that of RoI 0 plus a constant -> an instruction offset, no registers)
    constexpr int RS = NR == 1 ? CELLS + 1 : 64, NROWS = RS * (NR - 1) + CELLS + 1, RING = NR == 1 ? 4 : 3, RT = 4 * NR;
    __shared__ __attribute__((aligned(16))) unsigned char xs[NROWS * C * 2];
    __shared__ float bs[C];
    const int roi0 = blockIdx.x * NR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    // ---- weight fragments of this wave's 64 output columns, register ring over the 72 k-steps (requested first: in flight while the RoIs are staged)
    // fragment-major weights: fragment (k-step ks, column tile jt) = 64 lanes x 16 B at ((ks * 16 + jt) * 64 + lane) * 8
    const unsigned short* w_src = W + ((long long)(wave * 4) * 64 + lane) * 8;
    constexpr int KS_STRIDE = 16 * 64 * 8, JT_STRIDE = 64 * 8;
    Frag wq[RING][4];
#pragma unroll
    for (int p = 0; p < RING - 1; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[p][j].u = *reinterpret_cast<const uint4*>(w_src + p * KS_STRIDE + j * JT_STRIDE);
    // ---- stage the RoIs: 16-byte chunks, chunk c of row r at slot c ^ (r & 15).  All loads of a thread are issued before the first
    // LDS write (a rolled loop is one exposed memory round trip per iteration: 7 of them were a fifth of the block's time).
    {
        constexpr int NST = (NROWS * 32 + 255) / 256;
        cv_u32x4 st[NST];
        float bv = 0.f;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = tid + 256 * i, row = c >> 5, slot = c & 31;
            const int rl = min(row / RS, NR - 1), cell = min(row - rl * RS, CELLS - 1);
            const bool ok = row - rl * RS < CELLS && roi0 + rl < R;
            st[i] = *reinterpret_cast<const cv_u32x4*>(feat + ((long long)min(roi0 + rl, R - 1) * CELLS + cell) * C + slot * 8);   // branch-free
            if (!ok) st[i] = cv_u32x4{0u, 0u, 0u, 0u};
        }
        bv = bias[tid];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = tid + 256 * i, row = c >> 5, slot = c & 31;
            if (row < NROWS) *reinterpret_cast<cv_u32x4*>(xs + row * (C * 2) + ((slot ^ (row & 15)) << 4)) = st[i];
        }
        bs[tid] = bv;                                    // the bias waits in LDS: a global load in the epilogue is another round trip
    }
    // cell coordinates of the 4 row tiles of a RoI for this lane (cell = 16 i + fr)
    int cy[4], cx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = 16 * i + fr; cy[i] = r / 7; cx[i] = r - cy[i] * 7; }
    f32x4_t acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < 72; ++ks) {
        const int tap = ks >> 3, s = ks & 7;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        if (ks + RING - 1 < 72) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[(ks + RING - 1) % RING][j].u = *reinterpret_cast<const uint4*>(w_src + (ks + RING - 1) * KS_STRIDE + j * JT_STRIDE);
        }
        __builtin_amdgcn_sched_barrier(0);           // keep the prefetch ahead (hipcc otherwise sinks the loads to their use)
#pragma unroll
        for (int rl = 0; rl < NR; ++rl) {               // one RoI's four row tiles at a time: 4 fragment registers, not 4 NR
            Frag a[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int y = cy[it] + dy, x = cx[it] + dx;
                const bool ok = (16 * it + fr) < CELLS && y >= 0 && y < 7 && x >= 0 && x < 7;
                const int src = ok ? y * 7 + x : CELLS;
                a[it].u = *reinterpret_cast<const uint4*>(xs + rl * (RS * C * 2) + src * (C * 2) + (((4 * s + fg) ^ (src & 15)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[4 * rl + it][j] = mfma_k16_16x16x32(a[it].u, wq[ks % RING][j].u, acc[4 * rl + it][j]);
        }
    }
    // ---- bias + ReLU + mean over the 49 cells: lane holds cells 16 i + 4 fg + r of column 64 wave + 16 j + fr
#pragma unroll
    for (int rl = 0; rl < NR; ++rl) {
        if (roi0 + rl >= R) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = wave * 64 + 16 * j + fr;
            const float b = bs[n];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * i + 4 * fg + r < CELLS) sum += relu_f(acc[4 * rl + i][j][r] + b);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (fg == 0) out[(long long)(roi0 + rl) * ld_out + n] = sum / 49.0f;
        }
    }
}

// Split-precision variant (index-exact route of the engine): the RoI cells come as key16 hi + lo pairs (mv2d_roi_align_ex: x ~ hi + lo), the
// weights as fragment-major key16 hi / lo copies (mv2d_split_key16 + mv2d_pack_wfrag_bf16); every product is a_lo w_hi + a_hi w_lo + a_hi w_hi
// (three MFMAs into one fp32 accumulator; the lo x lo term, 2^-18 relative, is dropped).  One RoI per block: two 25 KB LDS images, a
// 3-deep ring for both weight streams, the same tap remapping and epilogue as the bf16 kernel.  MFMA-bound (3 x 2304 MFMAs per wave
// against 2.4 MB of weight fragments per block).
__global__ __launch_bounds__(256, 2) void roi_conv_pool_x3_kernel(const unsigned short* __restrict__ feat_hi, const unsigned short* __restrict__ feat_lo,
                                                                  const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                                                  const float* __restrict__ bias, float* __restrict__ out, int ld_out, int R) {
    constexpr int NROWS = CELLS + 1, RING = 3;
    __shared__ __attribute__((aligned(16))) unsigned char xh[NROWS * C * 2], xl[NROWS * C * 2];
    __shared__ float bs[C];
    const int roi = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const long long w_off = ((long long)(wave * 4) * 64 + lane) * 8;
    constexpr int KS_STRIDE = 16 * 64 * 8, JT_STRIDE = 64 * 8;
    Frag wqh[RING][4], wql[RING][4];
#pragma unroll
    for (int p = 0; p < RING - 1; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wqh[p][j].u = *reinterpret_cast<const uint4*>(Wh + w_off + p * KS_STRIDE + j * JT_STRIDE);
            wql[p][j].u = *reinterpret_cast<const uint4*>(Wl + w_off + p * KS_STRIDE + j * JT_STRIDE);
        }
    {
        constexpr int NST = (NROWS * 32 + 255) / 256;
        cv_u32x4 sh[NST], sl[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = tid + 256 * i, row = c >> 5, slot = c & 31;
            const bool ok = row < CELLS;
            const long long src = ((long long)roi * CELLS + min(row, CELLS - 1)) * C + slot * 8;
            sh[i] = *reinterpret_cast<const cv_u32x4*>(feat_hi + src);
            sl[i] = *reinterpret_cast<const cv_u32x4*>(feat_lo + src);
            if (!ok) { sh[i] = cv_u32x4{0u, 0u, 0u, 0u}; sl[i] = cv_u32x4{0u, 0u, 0u, 0u}; }
        }
        const float bv = bias[tid];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int c = tid + 256 * i, row = c >> 5, slot = c & 31;
            if (row < NROWS) {
                *reinterpret_cast<cv_u32x4*>(xh + row * (C * 2) + ((slot ^ (row & 15)) << 4)) = sh[i];
                *reinterpret_cast<cv_u32x4*>(xl + row * (C * 2) + ((slot ^ (row & 15)) << 4)) = sl[i];
            }
        }
        bs[tid] = bv;
    }
    int cy[4], cx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = 16 * i + fr; cy[i] = r / 7; cx[i] = r - cy[i] * 7; }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < 72; ++ks) {
        const int tap = ks >> 3, s = ks & 7;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        if (ks + RING - 1 < 72) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wqh[(ks + RING - 1) % RING][j].u = *reinterpret_cast<const uint4*>(Wh + w_off + (ks + RING - 1) * KS_STRIDE + j * JT_STRIDE);
                wql[(ks + RING - 1) % RING][j].u = *reinterpret_cast<const uint4*>(Wl + w_off + (ks + RING - 1) * KS_STRIDE + j * JT_STRIDE);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        Frag ah[4], al[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int y = cy[it] + dy, x = cx[it] + dx;
            const bool ok = (16 * it + fr) < CELLS && y >= 0 && y < 7 && x >= 0 && x < 7;
            const int src = ok ? y * 7 + x : CELLS;
            const int off = src * (C * 2) + (((4 * s + fg) ^ (src & 15)) << 4);
            ah[it].u = *reinterpret_cast<const uint4*>(xh + off);
            al[it].u = *reinterpret_cast<const uint4*>(xl + off);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[it][j] = mfma_k16_16x16x32(al[it].u, wqh[ks % RING][j].u, acc[it][j]);
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[it][j] = mfma_k16_16x16x32(ah[it].u, wql[ks % RING][j].u, acc[it][j]);
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[it][j] = mfma_k16_16x16x32(ah[it].u, wqh[ks % RING][j].u, acc[it][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = wave * 64 + 16 * j + fr;
        const float b = bs[n];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * i + 4 * fg + r < CELLS) sum += relu_f(acc[i][j][r] + b);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (fg == 0) out[(long long)roi * ld_out + n] = sum / 49.0f;
    }
}

// row-major W [N, K] bf16 -> fragment-major Wp[K/32][N/16][64][8]: Wp[ks][jt][fr + 16 fg][e] = W[16 jt + fr][32 ks + 8 fg + e]
__global__ void pack_wfrag_kernel(const unsigned short* __restrict__ W, unsigned short* __restrict__ Wp, int N, int K) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte chunk per thread
    const long long total = (long long)N * K / 8;
    if (idx >= total) return;
    const int lane = idx & 63;
    const long long t = idx >> 6;
    const int jt = (int)(t % (N / 16)), ks = (int)(t / (N / 16));
    const int fr = lane & 15, fg = lane >> 4;
    *reinterpret_cast<uint4*>(Wp + idx * 8) = *reinterpret_cast<const uint4*>(W + (long long)(16 * jt + fr) * K + 32 * ks + 8 * fg);
}

}  // namespace

extern "C" int mv2d_pack_wfrag_bf16(const void* W, void* Wp, int N, int K, void* stream) {
    MV2D_CHECK_ARG(W && Wp && N > 0 && K > 0 && (N % 16) == 0 && (K % 32) == 0, "mv2d_pack_wfrag_bf16: N % 16 == 0 and K % 32 == 0 required");
    const long long total = (long long)N * K / 8;
    hipLaunchKernelGGL(pack_wfrag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)W, (unsigned short*)Wp, N, K);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_qg_conv_pool(const void* roi_feat, const void* W, const float* bias, float* out, int ld_out, int R, void* stream) {
    MV2D_CHECK_ARG(roi_feat && W && bias && out && ld_out >= C, "mv2d_qg_conv_pool: bad args");
    MV2D_CHECK_ARG(((uintptr_t)roi_feat & 15) == 0 && ((uintptr_t)W & 15) == 0, "mv2d_qg_conv_pool: operands must be 16-byte aligned");
    if (R == 0) return MV2D_OK;
    // two RoIs per block once the grid fills the chip either way (identical results, see the kernel)
    const int nr = R >= 1024 ? 2 : 1;
    if (nr == 2)
        hipLaunchKernelGGL(roi_conv_pool_kernel<2>, dim3(cdiv(R, 2)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)roi_feat,
                           (const unsigned short*)W, bias, out, ld_out, R);
    else
        hipLaunchKernelGGL(roi_conv_pool_kernel<1>, dim3(R), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)roi_feat,
                           (const unsigned short*)W, bias, out, ld_out, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_qg_conv_pool_x3(const void* roi_feat_hi, const void* roi_feat_lo, const void* W_hi, const void* W_lo, const float* bias, float* out,
                                    int ld_out, int R, void* stream) {
    MV2D_CHECK_ARG(roi_feat_hi && roi_feat_lo && W_hi && W_lo && bias && out && ld_out >= C, "mv2d_qg_conv_pool_x3: bad args");
    MV2D_CHECK_ARG(((uintptr_t)roi_feat_hi & 15) == 0 && ((uintptr_t)roi_feat_lo & 15) == 0 && ((uintptr_t)W_hi & 15) == 0 && ((uintptr_t)W_lo & 15) == 0,
                   "mv2d_qg_conv_pool_x3: operands must be 16-byte aligned");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(roi_conv_pool_x3_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)roi_feat_hi,
                       (const unsigned short*)roi_feat_lo, (const unsigned short*)W_hi, (const unsigned short*)W_lo, bias, out, ld_out, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
