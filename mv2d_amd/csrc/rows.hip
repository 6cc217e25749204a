// Row-wise kernels over the [R, 256] query state of the MV2D decoder (gfx950, one wave64 per row,
// 4 channels per lane, wavefront reductions — no LDS).
#include "common.h"

namespace {

constexpr int C = 256;

struct LnParams {
    const float* parts; int n_parts; long long part_stride;   // sum of split-K partial slabs [n_parts][M][256]
    const float* bias;          // [256] or null
    const float* residual;      // [M,256] or null
    const float* ln_w; const float* ln_b;   // LayerNorm affine (null -> no LN, plain sum)
    int relu;                   // ReLU after LN
    float* out;                 // [M,256]  y
    const float* addvec; float* out_plus;   // out_plus = y + addvec (e.g. query_pos) or null
    const float* ln2_w; const float* ln2_b; float* out2;   // out2 = LN2(y) (shared post_norm) or null
    int M; float eps;
    int rows_per_group;         // > 0: bias / ln_w / ln_b of row r are at + (r / rows_per_group) * 256
};

__device__ __forceinline__ float4 ln4(float4 v, const float* w, const float* b, int c0, float eps) {
    float s = wave_sum(v.x + v.y + v.z + v.w);
    float mean = s * (1.0f / C);
    float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / C);
    float rstd = 1.0f / sqrtf(var + eps);
    float4 ww = *reinterpret_cast<const float4*>(w + c0);
    float4 bb = *reinterpret_cast<const float4*>(b + c0);
    return make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
}

// MU/petr_transformer.py:563-565,589-590 (norms / post_norm), mmcv BaseTransformerLayer residual rules,
// cross_attention_head.py:127-133 (Linear-LN-ReLU of the cls branch)
__global__ __launch_bounds__(256) void row_ln_kernel(LnParams p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int lane = threadIdx.x & 63, c0 = lane * 4;
    const int goff = p.rows_per_group > 0 ? (row / p.rows_per_group) * C : 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        // partial slabs: 8 loads in flight per step; the summation ORDER stays s = 0, 1, 2, ... (deterministic)
        const float* pp = p.parts + (long long)row * C + c0;
        int s = 0;
        for (; s + 8 <= p.n_parts; s += 8) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4*>(pp + (s + j) * p.part_stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
        }
        for (; s < p.n_parts; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(pp + s * p.part_stride);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
    }
    if (p.bias) {
        float4 t = *reinterpret_cast<const float4*>(p.bias + goff + c0);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (p.residual) {
        float4 t = *reinterpret_cast<const float4*>(p.residual + (long long)row * C + c0);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (p.ln_w) v = ln4(v, p.ln_w + goff, p.ln_b + goff, c0, p.eps);
    if (p.relu) { v.x = relu_f(v.x); v.y = relu_f(v.y); v.z = relu_f(v.z); v.w = relu_f(v.w); }
    if (p.out) *reinterpret_cast<float4*>(p.out + (long long)row * C + c0) = v;
    if (p.out_plus) {
        float4 t = *reinterpret_cast<const float4*>(p.addvec + (long long)row * C + c0);
        *reinterpret_cast<float4*>(p.out_plus + (long long)row * C + c0) = make_float4(v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w);
    }
    if (p.out2) *reinterpret_cast<float4*>(p.out2 + (long long)row * C + c0) = ln4(v, p.ln2_w, p.ln2_b, c0, p.eps);
}

// RH/utils/query_generator.py:322-331: AvgPool2d(7) over the 49 cells of relu(conv) -> [R,256]
__global__ __launch_bounds__(256) void avgpool49_kernel(const float* x, float* out, int ld_out, int R) {
    const int r = blockIdx.x, c = threadIdx.x;
    if (r >= R) return;
    float s = 0.f;
    for (int i = 0; i < 49; ++i) s += x[((long long)r * 49 + i) * C + c];
    out[(long long)r * ld_out + c] = s / 49.0f;
}

// fp32 -> bf16 copy (weights / activations), n multiple of 4 handled with a tail
__global__ void f32_to_bf16_kernel(const float* x, unsigned short* y, long long n) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<const float4*>(x + i);
        uint2 o = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        *reinterpret_cast<uint2*>(y + i) = o;
    } else {
        for (; i < n; ++i) y[i] = f32_to_bf16(x[i]);
    }
}

// fp32 -> (hi, lo) bf16 pair with x ~ hi + lo (static weights of the bf16x3 query-side kernels)
__global__ void split_bf16x2_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = x[i];
    const unsigned short h = f32_to_bf16(f);
    hi[i] = h;
    lo[i] = f32_to_bf16(f - __uint_as_float(((unsigned int)h) << 16));
}

// fp32 -> (hi, lo) pair in the query side's split format (common.h "q16": fp16 pairs since round 5; bf16 pairs in a -DMV2D_Q16_BF16 build):
// the static weights of the split-precision kernels (row-fused linears, FFN, heads, per-head maps, PE x3)
__global__ void split_q16x2_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned short h, l;
    split_q16(x[i], h, l);
    hi[i] = h;
    lo[i] = l;
}

// fp32 -> key16 (common.h: the key-side 16-bit format, fp16 since round 4): static weights of the key-side kernels; hi only, or hi + lo
__global__ void f32_to_key16_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const float a = x[i], b = i + 1 < n ? x[i + 1] : 0.f;
    unsigned int h, l;
    split_k16x2(a, b, h, l);
    hi[i] = (unsigned short)(h & 0xffffu);
    if (lo) lo[i] = (unsigned short)(l & 0xffffu);
    if (i + 1 < n) {
        hi[i + 1] = (unsigned short)(h >> 16);
        if (lo) lo[i + 1] = (unsigned short)(l >> 16);
    }
}

// NCHW fp32 [V,C,h*w] -> position-major [V*h*w, C] fp32 (LDS-tiled transpose, coalesced both ways)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, float* y, int V, int Cn, int HW) {
    __shared__ float tile[32][33];
    const int v = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < Cn && p < HW) ? x[((long long)v * Cn + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int p = p0 + i, c = c0 + tx;
        if (p < HW && c < Cn) y[((long long)v * HW + p) * Cn + c] = tile[tx][i];
    }
}

// same, 64 channels x 64 positions per block with 16-byte accesses on both sides (needs HW % 4 == 0, Cn % 4 == 0):
// every global row segment is 256 B, 4 loads + 4 stores in flight per thread instead of 4-byte accesses.
__global__ __launch_bounds__(256) void nchw_to_nhwc64_kernel(const float* __restrict__ x, float* __restrict__ y, int V, int Cn, int HW) {
    __shared__ float tile[64][65];
    const int v = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16
    float4 in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 16 * i, p = p0 + tq * 4;
        in[i] = (c < Cn && p < HW) ? *reinterpret_cast<const float4*>(x + ((long long)v * Cn + c) * HW + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* t = &tile[ty + 16 * i][tq * 4];
        t[0] = in[i].x; t[1] = in[i].y; t[2] = in[i].z; t[3] = in[i].w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = ty + 16 * i, p = p0 + pl, c = c0 + tq * 4;
        if (p < HW && c < Cn)
            *reinterpret_cast<float4*>(y + ((long long)v * HW + p) * Cn + c) =
                make_float4(tile[tq * 4][pl], tile[tq * 4 + 1][pl], tile[tq * 4 + 2][pl], tile[tq * 4 + 3][pl]);
    }
}

// The same transposition restricted to the positions somebody reads (round 5): mask [V * HW] bytes, 1 = the position is inside some RoI's rectangle
// (the engine's roi_mask: every RoIAlign tap, every key position and every row the PE block reads lies there).  A 64-position block without a listed
// position is skipped, a row is only written where the mask is set; the other rows of y keep whatever they held.  37 % (S path) / 50 % (T path) of
// the rows are listed: the 277 MB written per 16-sample cfg2_s frame become ~100 MB.
__global__ __launch_bounds__(256) void nchw_to_nhwc64_masked_kernel(const float* __restrict__ x, float* __restrict__ y, const unsigned char* __restrict__ mask,
                                                                   int V, int Cn, int HW) {
    __shared__ float tile[64][65];
    __shared__ unsigned char m[64];
    __shared__ int any;
    const int v = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int p = p0 + threadIdx.x;
        const unsigned char b = p < HW ? mask[(long long)v * HW + p] : 0;
        m[threadIdx.x] = b;
        if (b) any = 1;
    }
    __syncthreads();
    if (!any) return;
    const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float4 in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 16 * i, p = p0 + tq * 4;
        in[i] = (c < Cn && p < HW) ? *reinterpret_cast<const float4*>(x + ((long long)v * Cn + c) * HW + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* t = &tile[ty + 16 * i][tq * 4];
        t[0] = in[i].x; t[1] = in[i].y; t[2] = in[i].z; t[3] = in[i].w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = ty + 16 * i, p = p0 + pl, c = c0 + tq * 4;
        if (p < HW && c < Cn && m[pl])
            *reinterpret_cast<float4*>(y + ((long long)v * HW + p) * Cn + c) =
                make_float4(tile[tq * 4][pl], tile[tq * 4 + 1][pl], tile[tq * 4 + 2][pl], tile[tq * 4 + 3][pl]);
    }
}

// cross_attention_head.py:216-238 tail: reg[0:2] = sigmoid(reg[0:2] + isig(ref)[0:2]), reg[4] = sigmoid(reg[4] + isig(ref)[2]),
// de-normalise to metres with pc_range; RH/mv2d_t_head.py:136-140: reg[8:10] /= dt when dt != 0.  reg [L,R,10] in place.
__global__ void finalize_reg_kernel(float* reg, const float* __restrict__ ref, int L, int R, float pc0, float pc1, float pc2,
                                    float pd0, float pd1, float pd2, float dt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * R) return;
    const int r = i % R;
    float* t = reg + (long long)i * 10;
    float is[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float x = fminf(fmaxf(ref[r * 3 + k], 0.f), 1.f);
        is[k] = logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
    }
    const float s0 = 1.f / (1.f + expf(-(t[0] + is[0])));
    const float s1 = 1.f / (1.f + expf(-(t[1] + is[1])));
    const float s4 = 1.f / (1.f + expf(-(t[4] + is[2])));
    t[0] = s0 * pd0 + pc0;
    t[1] = s1 * pd1 + pc1;
    t[4] = s4 * pd2 + pc2;
    if (dt != 0.f) { t[8] = t[8] / dt; t[9] = t[9] / dt; }
}

}  // namespace

extern "C" int mv2d_finalize_reg(float* reg, const float* ref, int L, int R, const float* pc_range, float dt, void* stream) {
    MV2D_CHECK_ARG(reg && ref && pc_range && L > 0, "mv2d_finalize_reg: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(finalize_reg_kernel, dim3(cdiv(L * R, 256)), dim3(256), 0, (hipStream_t)stream, reg, ref, L, R, pc_range[0],
                       pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2], dt);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_row_ln(const float* parts, int n_parts, long long part_stride, const float* bias,
                           const float* residual, const float* ln_w, const float* ln_b, int relu, float* out,
                           const float* addvec, float* out_plus, const float* ln2_w, const float* ln2_b, float* out2,
                           int M, float eps, int rows_per_group, void* stream) {
    MV2D_CHECK_ARG(parts && n_parts >= 1 && (out || out_plus || out2), "mv2d_row_ln: null input/output");
    MV2D_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr), "mv2d_row_ln: ln_w/ln_b must both be set or null");
    MV2D_CHECK_ARG(!out_plus || addvec, "mv2d_row_ln: out_plus needs addvec");
    MV2D_CHECK_ARG(!out2 || (ln2_w && ln2_b), "mv2d_row_ln: out2 needs ln2_w/ln2_b");
    if (M == 0) return MV2D_OK;
    LnParams p{parts, n_parts, part_stride, bias, residual, ln_w, ln_b, relu, out, addvec, out_plus, ln2_w, ln2_b, out2, M, eps, rows_per_group};
    hipLaunchKernelGGL(row_ln_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_avgpool49(const float* x, float* out, int ld_out, int R, void* stream) {
    MV2D_CHECK_ARG(x && out && ld_out >= 256, "mv2d_avgpool49: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(avgpool49_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, out, ld_out, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
    MV2D_CHECK_ARG(x && y && n >= 0, "mv2d_f32_to_bf16: bad args");
    if (n == 0) return MV2D_OK;
    long long threads = (n + 3) / 4;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned short*)y, n);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_split_bf16x2(const float* x, void* hi, void* lo, long long n, void* stream) {
    MV2D_CHECK_ARG(x && hi && lo && n >= 0, "mv2d_split_bf16x2: bad args");
    if (n == 0) return MV2D_OK;
    hipLaunchKernelGGL(split_bf16x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)hi,
                       (unsigned short*)lo, n);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_split_q16x2(const float* x, void* hi, void* lo, long long n, void* stream) {
    MV2D_CHECK_ARG(x && hi && lo && n >= 0, "mv2d_split_q16x2: bad args");
    if (n == 0) return MV2D_OK;
    hipLaunchKernelGGL(split_q16x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)hi,
                       (unsigned short*)lo, n);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// 1 = IEEE fp16 pairs, 0 = bf16 pairs (a -DMV2D_Q16_BF16 build): the split format of the query side, see common.h
extern "C" int mv2d_q16_format(void) { return MV2D_Q16_IS_F16; }

// 1 = IEEE fp16, 0 = bf16 (a -DMV2D_KEY16_BF16 build): the 16-bit format of the key side, see common.h
extern "C" int mv2d_key16_format(void) { return MV2D_KEY16_IS_F16; }

extern "C" int mv2d_f32_to_key16(const float* x, void* hi, void* lo, long long n, void* stream) {
    MV2D_CHECK_ARG(x && hi && n >= 0, "mv2d_f32_to_key16: bad args");
    if (n == 0) return MV2D_OK;
    const long long threads = (n + 1) / 2;
    hipLaunchKernelGGL(f32_to_key16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)hi,
                       (unsigned short*)lo, n);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_nchw_to_nhwc_masked(const float* x, float* y, const unsigned char* mask, int V, int Cn, int HW, void* stream) {
    MV2D_CHECK_ARG(x && y && mask && V > 0 && Cn > 0 && HW > 0, "mv2d_nchw_to_nhwc_masked: bad args");
    MV2D_CHECK_ARG((HW % 4) == 0 && (Cn % 4) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
                   "mv2d_nchw_to_nhwc_masked: HW and C must be multiples of 4, x and y 16-byte aligned");
    dim3 grid(cdiv(HW, 64), cdiv(Cn, 64), V);
    hipLaunchKernelGGL(nchw_to_nhwc64_masked_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, mask, V, Cn, HW);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_nchw_to_nhwc(const float* x, float* y, int V, int Cn, int HW, void* stream) {
    MV2D_CHECK_ARG(x && y && V > 0 && Cn > 0 && HW > 0, "mv2d_nchw_to_nhwc: bad args");
    if ((HW % 4) == 0 && (Cn % 4) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
        dim3 grid(cdiv(HW, 64), cdiv(Cn, 64), V);
        hipLaunchKernelGGL(nchw_to_nhwc64_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, V, Cn, HW);
    } else {
        dim3 grid(cdiv(HW, 32), cdiv(Cn, 32), V);
        hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, V, Cn, HW);
    }
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
