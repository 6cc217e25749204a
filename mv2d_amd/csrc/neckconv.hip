// 3x3 convolution (padding 1, 256 -> 256 channels) over the stride-16 feature maps [V, h, w, 256] — the fpn_conv of the extra FPN
// level MV2D puts between the 2-D detector and the RoI head (neck=dict(type='FPN', start_level=2, end_level=2, num_outs=1),
// configs/mv2d/exp/*:32-39; mmdet FPN: lateral 1x1 conv + 3x3 conv, no norm, no activation; called from
// mmdet3d_plugin/models/detectors/mv2d.py:122-127, 256-258).  "Next" row f2 of SURVEY.md 8(f).
//
// Same pattern as roiconv.hip: a block owns a 4 x 16 patch of one map; the patch + halo (6 x 18 positions x 256 ch bf16 = 55 KB)
// is staged in LDS once with the zero padding filled in, every tap is an offset into it; the weights ([out][tap][cin], K = 2304)
// stream fragment-major from L2 through a 4-deep register ring; no barrier in the 72-step k loop.  MFMAs run swapped, so a lane
// ends with 4 consecutive output channels of one position: bias + one float4 store per tile into the position-major fp32 map
// that the RoI-head engine reads directly (no NCHW->NHWC pass afterwards).
#include "common.h"

namespace {

constexpr int C = 256, KT = 9 * C, TH = 4, TW = 16, HW_ = TW + 2, HH_ = TH + 2;
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
union Frag { uint4 u; mfma_bf16x8 v; };

__global__ __launch_bounds__(256, 2) void map_conv3x3_kernel(const unsigned short* __restrict__ in, const unsigned short* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out, int V, int h, int w) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[HH_ * HW_ * C * 2];         // 108 positions x 512 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int v = b / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    // ---- stage patch + halo: 16-byte chunks, chunk c of halo position q at slot c ^ (q & 15); outside the map -> zeros
    for (int c = tid; c < HH_ * HW_ * 32; c += 256) {
        const int q = c >> 5, slot = c & 31;
        const int qy = q / HW_, qx = q - qy * HW_;
        const int y = y0 + qy - 1, x = x0 + qx - 1;
        uint4 val = make_uint4(0u, 0u, 0u, 0u);
        if (y >= 0 && y < h && x >= 0 && x < w) val = *reinterpret_cast<const uint4*>(in + (((long long)v * h + y) * w + x) * C + slot * 8);
        *reinterpret_cast<uint4*>(xs + q * (C * 2) + ((slot ^ (q & 15)) << 4)) = val;
    }
    // fragment-major weights: fragment (k-step ks, column tile jt) at ((ks * 16 + jt) * 64 + lane) * 8
    const unsigned short* w_src = W + ((long long)(wave * 4) * 64 + lane) * 8;
    constexpr int KS_STRIDE = 16 * 64 * 8, JT_STRIDE = 64 * 8;
    Frag wq[4][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[p][j].u = *reinterpret_cast<const uint4*>(w_src + p * KS_STRIDE + j * JT_STRIDE);
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < 72; ++ks) {
        const int tap = ks >> 3, s = ks & 7;
        const int dy = tap / 3, dx = tap - (tap / 3) * 3;          // halo coordinates: output (i, fr) reads halo (i + dy, fr + dx)
        if (ks + 3 < 72) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[(ks + 3) & 3][j].u = *reinterpret_cast<const uint4*>(w_src + (ks + 3) * KS_STRIDE + j * JT_STRIDE);
        }
        __builtin_amdgcn_sched_barrier(0);
        Frag a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = (i + dy) * HW_ + fr + dx;
            a[i].u = *reinterpret_cast<const uint4*>(xs + q * (C * 2) + (((4 * s + fg) ^ (q & 15)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)        // swapped: D[n][pos] — lane (fr, fg) holds channels 4fg..4fg+3 of position fr
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks & 3][j].v, a[i].v, acc[i][j], 0, 0, 0);
    }
    const int x = x0 + fr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + i;
        if (y >= h || x >= w) continue;
        float* op = out + (((long long)v * h + y) * w + x) * C + wave * 64 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + wave * 64 + 16 * j + 4 * fg);
            *reinterpret_cast<float4*>(op + 16 * j) = make_float4(acc[i][j][0] + bb.x, acc[i][j][1] + bb.y, acc[i][j][2] + bb.z, acc[i][j][3] + bb.w);
        }
    }
}

// NCHW fp32 [V,C,h*w] -> position-major bf16 [V*h*w, C] (the lateral conv's input operand)
__global__ __launch_bounds__(256) void nchw_to_nhwc_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int V, int Cn, int HW) {
    __shared__ float tile[64][65];
    const int v = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 16 * i;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + tq * 4 + k;
            tile[ty + 16 * i][tq * 4 + k] = (c < Cn && p < HW) ? x[((long long)v * Cn + c) * HW + p] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = ty + 16 * i, p = p0 + pl, c = c0 + tq * 4;
        if (p < HW && c + 3 < Cn)
            *reinterpret_cast<uint2*>(y + ((long long)v * HW + p) * Cn + c) =
                make_uint2(pack_bf16x2(tile[tq * 4][pl], tile[tq * 4 + 1][pl]), pack_bf16x2(tile[tq * 4 + 2][pl], tile[tq * 4 + 3][pl]));
    }
}

}  // namespace

extern "C" int mv2d_map_conv3x3(const void* in, const void* Wp, const float* bias, float* out, int V, int h, int w, void* stream) {
    MV2D_CHECK_ARG(in && Wp && bias && out && V > 0 && h > 0 && w > 0, "mv2d_map_conv3x3: bad args");
    MV2D_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)Wp & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)bias & 15) == 0,
                   "mv2d_map_conv3x3: operands must be 16-byte aligned");
    const int blocks = V * ((h + TH - 1) / TH) * ((w + TW - 1) / TW);
    hipLaunchKernelGGL(map_conv3x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in,
                       (const unsigned short*)Wp, bias, out, V, h, w);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_nchw_to_nhwc_bf16(const float* x, void* y, int V, int Cn, int HW, void* stream) {
    MV2D_CHECK_ARG(x && y && V > 0 && Cn > 0 && (Cn % 4) == 0 && HW > 0, "mv2d_nchw_to_nhwc_bf16: bad args");
    dim3 grid(cdiv(HW, 64), cdiv(Cn, 64), V);
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, V, Cn, HW);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
