// Attention kernels of the MV2D decoder (gfx950 / CDNA4, wave64).
//
//  * mv2d_self_attn_fwd   — FlattenMHSelfAttention core (MU/petr_transformer.py:317-370): dense softmax
//                           attention among all R queries, 8 heads x 32, exact-fp32 MFMA with the "swapped"
//                           S^T = K.Q^T formulation so the softmax row (keys) stays lane-local + 2 shuffles and
//                           P feeds the P.V MFMA straight from registers (no LDS round trip).
//  * mv2d_sparse_xattn_fwd — PETRMultiheadAttention core (MU/petr_transformer.py:426-513) restricted to the
//                           keys each query is allowed to see (CSR row_ptr/col_idx), instead of the reference's
//                           dense [8,R,S] logits + boolean mask.  Keys/values are the bf16 per-layer projections.
#include "common.h"

namespace {

constexpr int C = 256, HEADS = 8, HD = 32;

// ------------------------------------------------------------------------------------------------
// self attention: one wave per (16-query tile, head)
//   qkv: [R, 768] fp32 = (q | k | v) in_proj outputs (q NOT yet scaled), ctx: [R, 256] fp32
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void self_attn_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int R, float scale) {
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    const int q0 = blockIdx.x * 16, h = blockIdx.y;
    const int qrow = min(q0 + fr, R - 1);
    const float* qp = qkv + (long long)qrow * 768 + h * HD + 4 * fg;
    const float4 qa = *reinterpret_cast<const float4*>(qp);
    const float4 qb = *reinterpret_cast<const float4*>(qp + 16);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    const int ntiles = (R + 15) / 16;
    for (int t = 0; t < ntiles; ++t) {
        const int krow = min(t * 16 + fr, R - 1);
        const float* kp = qkv + (long long)krow * 768 + C + h * HD + 4 * fg;
        const float4 ka = *reinterpret_cast<const float4*>(kp);
        const float4 kb = *reinterpret_cast<const float4*>(kp + 16);
        // S^T[key = 16t + 4fg + reg][query = fr]; the d index is spread over (MFMA step, lane group) by the
        // same bijection for K and Q: step j of the first half contracts d = 4g + j, g = 0..3.
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.x, qa.x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.y, qa.y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.z, qa.z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.w, qa.w, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.x, qb.x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.y, qb.y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.z, qb.z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.w, qb.w, s, 0, 0, 0);
        float p[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = t * 16 + 4 * fg + r;
            p[r] = key < R ? s[r] * scale : -INFINITY;
            tmax = fmaxf(tmax, p[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = expf(p[r] - m_new); psum += p[r]; }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // O^T[d][query] += sum_key V[key][d] * P^T[key][query]; MFMA step r contracts keys 16t + 4g + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int vrow = min(t * 16 + 4 * fg + r, R - 1);
            const float* vp = qkv + (long long)vrow * 768 + 2 * C + h * HD + fr;
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[0], p[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16], p[r], o1, 0, 0, 0);
        }
    }
    if (q0 + fr < R) {
        const float inv = 1.0f / l_run;
        float* op = ctx + (long long)(q0 + fr) * C + h * HD + 4 * fg;
        *reinterpret_cast<float4*>(op) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
        *reinterpret_cast<float4*>(op + 16) = make_float4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
    }
}

// ------------------------------------------------------------------------------------------------
// sparse cross attention, one wave per query; lane l owns channels 4l..4l+3 (head = l >> 3).
//   q: [R,256] fp32 already scaled by 1/sqrt(32); K,V: [S,256] bf16; out ctx [R,256] fp32
//   A query with no allowed key gets ctx = 0 (the reference yields NaN there — DESIGN.md).
//   Optional debug outputs: logits [8][nnz] (pre-softmax, CSR order).
// ------------------------------------------------------------------------------------------------
constexpr int KCH = 8;   // keys per chunk (independent loads in flight)

__global__ __launch_bounds__(256) void sparse_xattn_kernel(const float* __restrict__ q, const unsigned short* __restrict__ K,
                                                           const unsigned short* __restrict__ V, const int* __restrict__ row_ptr,
                                                           const int* __restrict__ col_idx, float* __restrict__ ctx,
                                                           float* __restrict__ dbg_logits, long long dbg_stride, int R) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float4 q4 = *reinterpret_cast<const float4*>(q + (long long)r * C + 4 * lane);
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    float m_run = -INFINITY, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = beg; base < end; base += KCH) {
        uint2 kk[KCH], vv[KCH];
        float lg[KCH];
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int e = min(base + i, end - 1);
            const long long row = (long long)col_idx[e] * C + 4 * lane;
            kk[i] = *reinterpret_cast<const uint2*>(K + row);
            vv[i] = *reinterpret_cast<const uint2*>(V + row);
        }
        float cmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            float d = __uint_as_float(kk[i].x << 16) * q4.x;
            d = fmaf(__uint_as_float(kk[i].x & 0xffff0000u), q4.y, d);
            d = fmaf(__uint_as_float(kk[i].y << 16), q4.z, d);
            d = fmaf(__uint_as_float(kk[i].y & 0xffff0000u), q4.w, d);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (dbg_logits && (lane & 7) == 0 && base + i < end) dbg_logits[(long long)(lane >> 3) * dbg_stride + base + i] = d;
            lg[i] = (base + i < end) ? d : -INFINITY;
            cmax = fmaxf(cmax, lg[i]);
        }
        const float m_new = fmaxf(m_run, cmax);
        const float alpha = expf(m_run - m_new);
        acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const float pi = expf(lg[i] - m_new);
            psum += pi;
            acc.x = fmaf(pi, __uint_as_float(vv[i].x << 16), acc.x);
            acc.y = fmaf(pi, __uint_as_float(vv[i].x & 0xffff0000u), acc.y);
            acc.z = fmaf(pi, __uint_as_float(vv[i].y << 16), acc.z);
            acc.w = fmaf(pi, __uint_as_float(vv[i].y & 0xffff0000u), acc.w);
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
    }
    const float inv = (end > beg) ? 1.0f / l_run : 0.f;
    *reinterpret_cast<float4*>(ctx + (long long)r * C + 4 * lane) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

}  // namespace

extern "C" int mv2d_self_attn_fwd(const float* qkv, float* ctx, int R, void* stream) {
    MV2D_CHECK_ARG(qkv && ctx && R >= 0, "mv2d_self_attn_fwd: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(self_attn_kernel, dim3(cdiv(R, 16), HEADS), dim3(64), 0, (hipStream_t)stream, qkv, ctx, R,
                       1.0f / sqrtf((float)HD));
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_sparse_xattn_fwd(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx,
                                     float* ctx, float* dbg_logits, long long dbg_stride, int R, void* stream) {
    MV2D_CHECK_ARG(q && K && V && row_ptr && col_idx && ctx && R >= 0, "mv2d_sparse_xattn_fwd: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(sparse_xattn_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, q, (const unsigned short*)K,
                       (const unsigned short*)V, row_ptr, col_idx, ctx, dbg_logits, dbg_stride, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
