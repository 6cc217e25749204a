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
// self attention: one 4-wave block per (16-query tile, head); the key tiles are dealt round-robin to the
// 4 waves (flash-decoding style split), each wave prefetches its next tile's K/V fragments while the MFMAs of
// the current one run, and the 4 partial (m, l, O) states are merged through 8.5 KB of LDS.
//   qkv: [R, 768] fp32 = (q | k | v) in_proj outputs (q NOT yet scaled), ctx: [R, 256] fp32
// ------------------------------------------------------------------------------------------------
struct KVFrag { float4 ka, kb; float v0[4], v1[4]; };

__device__ __forceinline__ void load_kv(KVFrag& f, const float* __restrict__ qkv, int t, int h, int fr, int fg, int R) {
    const int krow = min(t * 16 + fr, R - 1);
    const float* kp = qkv + (long long)krow * 768 + C + h * HD + 4 * fg;
    f.ka = *reinterpret_cast<const float4*>(kp);
    f.kb = *reinterpret_cast<const float4*>(kp + 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int vrow = min(t * 16 + 4 * fg + r, R - 1);
        const float* vp = qkv + (long long)vrow * 768 + 2 * C + h * HD + fr;
        f.v0[r] = vp[0];
        f.v1[r] = vp[16];
    }
}

// Several samples in one launch (grp_start[n_grp + 1] = first query row of every sample): attention stays inside a sample, and the
// query / key tiles are counted from the sample's first row, so a sample's result does not depend on what else is in the batch.
// DN (training): the first dn_pad rows are denoising queries in groups of dn_single (prepare_for_dn's attn_mask,
// RH/mv2d_s_head.py:95-107): a key is visible iff it is a matched query (key >= dn_pad) or the query is a denoising query of the same group.
template <bool DN>
__global__ __launch_bounds__(256) void self_attn_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int R, float scale,
                                                        const int* __restrict__ grp_start, int n_grp, int dn_pad, int dn_single) {
    __shared__ float sm[4][16], sl[4][16], so[4][32][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    int tile = blockIdx.x;
    if (grp_start) {
        int g = 0, gs = 0, ge = 0;
        for (; g < n_grp; ++g) {
            gs = grp_start[g]; ge = grp_start[g + 1];
            const int nt = (ge - gs + 15) >> 4;
            if (tile < nt) break;
            tile -= nt;
        }
        if (g == n_grp) return;
        qkv += (long long)gs * 768; ctx += (long long)gs * C; R = ge - gs;
    } else if (tile * 16 >= R) return;
    const int q0 = tile * 16, h = blockIdx.y;
    const int qrow = min(q0 + fr, R - 1);
    const float* qp = qkv + (long long)qrow * 768 + h * HD + 4 * fg;
    const float4 qa = *reinterpret_cast<const float4*>(qp);
    const float4 qb = *reinterpret_cast<const float4*>(qp + 16);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    const int ntiles = (R + 15) / 16;
    KVFrag cur, nxt;
    if (wave < ntiles) load_kv(cur, qkv, wave, h, fr, fg, R);
    for (int t = wave; t < ntiles; t += 4) {
        if (t + 4 < ntiles) load_kv(nxt, qkv, t + 4, h, fr, fg, R);
        // S^T[key = 16t + 4fg + reg][query = fr]; the d index is spread over (MFMA step, lane group) by the
        // same bijection for K and Q: step j of the first half contracts d = 4g + j, g = 0..3.
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.x, qa.x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.y, qa.y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.z, qa.z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.w, qa.w, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.x, qb.x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.y, qb.y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.z, qb.z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.w, qb.w, s, 0, 0, 0);
        float p[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = t * 16 + 4 * fg + r;
            bool vis = key < R;
            if (DN) vis = vis && (key >= dn_pad || (q0 + fr < dn_pad && key / dn_single == (q0 + fr) / dn_single));
            p[r] = vis ? s[r] * scale : -INFINITY;
            tmax = fmaxf(tmax, p[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (DN && m_new == -INFINITY) ? 0.f : m_new;      // nothing visible to this query so far: (m, l, o) stay (-inf, 0, 0), no NaN
        const float alpha = expf(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = expf(p[r] - m_use); psum += p[r]; }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // O^T[d][query] += sum_key V[key][d] * P^T[key][query]; MFMA step r contracts keys 16t + 4g + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v0[r], p[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v1[r], p[r], o1, 0, 0, 0);
        }
        cur = nxt;
    }
    if (fg == 0) { sm[wave][fr] = m_run; sl[wave][fr] = l_run; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { so[wave][4 * fg + r][fr] = o0[r]; so[wave][16 + 4 * fg + r][fr] = o1[r]; }
    __syncthreads();
    // merge: thread -> query q = tid & 15, two d values
    const int q = tid & 15, d0 = (tid >> 4) * 2;
    if (q0 + q < R) {
        const float M = fmaxf(fmaxf(sm[0][q], sm[1][q]), fmaxf(sm[2][q], sm[3][q]));
        float den = 0.f, n0 = 0.f, n1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float e = expf(sm[w][q] - M);        // waves without a tile: exp(-inf) = 0
            den += sl[w][q] * e;
            n0 += so[w][d0][q] * e;
            n1 += so[w][d0 + 1][q] * e;
        }
        const float inv = 1.0f / den;
        *reinterpret_cast<float2*>(ctx + (long long)(q0 + q) * C + h * HD + d0) = make_float2(n0 * inv, n1 * inv);
    }
}

// ------------------------------------------------------------------------------------------------
// sparse cross attention: one NW-wave block per query; lane l of every wave owns channels 4l..4l+3 (head = l >> 3).
// The row's keys are dealt to the waves in chunks of KCH (round robin).  A row is a serial chain of dependent gathers
// (index -> K/V rows), so the kernel's time is the LONGEST row's chain: three chunks are in flight per wave — the indices of
// chunk c+2 (one coalesced load, broadcast with readlane), the K/V rows of chunk c+1 (512 B coalesced per key) and the
// arithmetic of chunk c — and long rows get 8 waves x 16 keys per round.  Partial (m, l, acc) states merge through LDS.
//   q: [R,256] fp32 already scaled by 1/sqrt(32); K,V: [S,256] bf16; out ctx [R,256] fp32
//   A query with no allowed key gets ctx = NaN like the reference (empty_nan = 1; the NaN then spreads to every query through
//   the next self attention) or ctx = 0 (empty_nan = 0) — DESIGN.md.
//   Optional debug output: logits [8][nnz] (pre-softmax, CSR order).
// ------------------------------------------------------------------------------------------------
template <int KCH>
struct KeyChunk { uint2 kk[KCH], vv[KCH]; };

template <int KCH>
__device__ __forceinline__ int load_idx(const int* __restrict__ col_idx, int base, int end, int lane) {
    const int e = min(base + (lane & (KCH - 1)), end - 1);
    return base < end ? col_idx[e] : 0;
}

template <int KCH>
__device__ __forceinline__ void load_rows(KeyChunk<KCH>& c, const unsigned short* __restrict__ K, const unsigned short* __restrict__ V,
                                          int idx, int lane) {
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const long long row = (long long)__builtin_amdgcn_readlane(idx, i) * C + 4 * lane;
        c.kk[i] = *reinterpret_cast<const uint2*>(K + row);
        c.vv[i] = *reinterpret_cast<const uint2*>(V + row);
    }
}

// Attention-probability dropout (nn.MultiheadAttention's `dropout`, MU/petr_transformer.py:404-418, in training): the keep decision of the
// probability of (allowed pair e, head h) is a counter-based hash of (seed, 8 e + h) (murmur3 finaliser) compared with thr = p * 2^32, so the
// forward and the backward kernels regenerate the same mask without storing it; kept probabilities are scaled by 1 / (1 - p).
// thr == 0: no dropout (every pair kept, scale 1).
struct AttnDrop { unsigned int thr, seed; float scale; };
__device__ __forceinline__ float attn_drop_factor(const AttnDrop& d, long long e, int h) {
    if (d.thr == 0u) return 1.f;
    unsigned int u = ((unsigned int)(e * 8 + h) * 0x9E3779B1u) ^ d.seed;
    u ^= u >> 16; u *= 0x85EBCA6Bu; u ^= u >> 13; u *= 0xC2B2AE35u; u ^= u >> 16;
    return u >= d.thr ? d.scale : 0.f;
}

template <int NW, int KCH>
__global__ __launch_bounds__(64 * NW) void sparse_xattn_kernel(const float* __restrict__ q, const unsigned short* __restrict__ K,
                                                               const unsigned short* __restrict__ V, const int* __restrict__ row_ptr,
                                                               const int* __restrict__ col_idx, float* __restrict__ ctx,
                                                               float* __restrict__ dbg_logits, long long dbg_stride, int R, int empty_nan,
                                                               AttnDrop drop) {
    __shared__ float sm[NW][8], sl[NW][8], sacc[NW][C];
    const int r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    constexpr int STEP = NW * KCH;
    int base = beg + wave * KCH;
    int idx_next = load_idx<KCH>(col_idx, base, end, lane);
    int idx_next2 = load_idx<KCH>(col_idx, base + STEP, end, lane);
    const float4 q4 = *reinterpret_cast<const float4*>(q + (long long)r * C + 4 * lane);
    float m_run = -INFINITY, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    KeyChunk<KCH> cur, nxt;
    if (base < end) load_rows<KCH>(cur, K, V, idx_next, lane);
    for (; base < end; base += STEP) {
        if (base + STEP < end) load_rows<KCH>(nxt, K, V, idx_next2, lane);
        idx_next2 = load_idx<KCH>(col_idx, base + 2 * STEP, end, lane);
        float lg[KCH];
        float cmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            float d = __uint_as_float(cur.kk[i].x << 16) * q4.x;
            d = fmaf(__uint_as_float(cur.kk[i].x & 0xffff0000u), q4.y, d);
            d = fmaf(__uint_as_float(cur.kk[i].y << 16), q4.z, d);
            d = fmaf(__uint_as_float(cur.kk[i].y & 0xffff0000u), q4.w, d);
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
            float pi = expf(lg[i] - m_new);
            psum += pi;                                                    // the softmax denominator sees every allowed key
            pi *= attn_drop_factor(drop, base + i, lane >> 3);             // dropped / rescaled probability enters the value sum only
            acc.x = fmaf(pi, __uint_as_float(cur.vv[i].x << 16), acc.x);
            acc.y = fmaf(pi, __uint_as_float(cur.vv[i].x & 0xffff0000u), acc.y);
            acc.z = fmaf(pi, __uint_as_float(cur.vv[i].y << 16), acc.z);
            acc.w = fmaf(pi, __uint_as_float(cur.vv[i].y & 0xffff0000u), acc.w);
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        cur = nxt;
    }
    if ((lane & 7) == 0) { sm[wave][lane >> 3] = m_run; sl[wave][lane >> 3] = l_run; }
    *reinterpret_cast<float4*>(&sacc[wave][4 * lane]) = acc;
    __syncthreads();
    if (tid < C) {                               // thread tid -> channel tid, head tid / 32
        const int hh = tid >> 5;
        float out = empty_nan ? __uint_as_float(0x7fc00000u) : 0.f;   // no allowed key: NaN like nn.MultiheadAttention, or 0
        if (end > beg) {
            float M = sm[0][hh];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, sm[w][hh]);
            float den = 0.f, num = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float e = expf(sm[w][hh] - M);
                den += sl[w][hh] * e;
                num += sacc[w][tid] * e;
            }
            out = num / den;
        }
        ctx[(long long)r * C + tid] = out;
    }
}

__device__ __forceinline__ float4 bf16x4(const uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// ------------------------------------------------------------------------------------------------
// Backward of the sparse cross attention ("next" row f3: the training path of the head needs it; forward = sparse_xattn_kernel).
//   s_j = q.k_j, p = softmax_j(s) per head, ctx = sum_j p_j v_j   over the keys j the CSR row allows
//   d ctx given:  dp_j = dctx.v_j,  ds_j = p_j (dp_j - D),  D = sum_j p_j dp_j = dctx.ctx (per head)
//   dq = sum_j ds_j k_j,   dK[j] += ds_j q,   dV[j] += p_j dctx        (dK / dV: fp32 atomics, several queries share a key)
// One 4-wave block per query, lane l = channels 4l..4l+3 (head l >> 3) like the forward kernel; the keys of the row are dealt to
// the waves round robin.  Pass 1 recomputes the softmax statistics (no forward state is kept), pass 2 forms the gradients.
// q is the SCALED query the forward kernel received (dq is the gradient with respect to it).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float head_sum(float d) {
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    return d;
}

// pass over the queries: dq, and per allowed pair e the probabilities / logit gradients of the 8 heads: pd[e][0..7] = p, pd[e][8..15] = ds
template <int NW>
__global__ __launch_bounds__(64 * NW) void sparse_xattn_bwd_q_kernel(const float* __restrict__ q, const unsigned short* __restrict__ K,
                                                                 const unsigned short* __restrict__ V, const int* __restrict__ row_ptr,
                                                                 const int* __restrict__ col_idx, const float* __restrict__ ctx,
                                                                 const float* __restrict__ dctx, float* __restrict__ dq, float* __restrict__ pd, int R,
                                                                 AttnDrop drop, float dq_scale) {
    __shared__ float sm[NW][8], sl[NW][8], sdq[NW][C];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 3;
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    const long long ro = (long long)r * C + 4 * lane;
    if (end <= beg) {                                        // no key: the forward output does not depend on q
        if (wave == 0) *reinterpret_cast<float4*>(dq + ro) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 q4 = *reinterpret_cast<const float4*>(q + ro);
    const float4 c4 = *reinterpret_cast<const float4*>(ctx + ro);
    const float4 d4 = *reinterpret_cast<const float4*>(dctx + ro);
    const float D = head_sum(d4.x * c4.x + d4.y * c4.y + d4.z * c4.z + d4.w * c4.w);
    // ---- pass 1: softmax statistics of the row (per head); the next key's row is requested before the current one is used
    float m_run = -INFINITY, l_run = 0.f;
    {
        int e = beg + wave;
        uint2 kn = e < end ? *reinterpret_cast<const uint2*>(K + (long long)col_idx[e] * C + 4 * lane) : make_uint2(0u, 0u);
        for (; e < end; e += NW) {
            const float4 k4 = bf16x4(kn);
            if (e + NW < end) kn = *reinterpret_cast<const uint2*>(K + (long long)col_idx[e + NW] * C + 4 * lane);
            const float sv = head_sum(k4.x * q4.x + k4.y * q4.y + k4.z * q4.z + k4.w * q4.w);
            const float m_new = fmaxf(m_run, sv);
            l_run = l_run * expf(m_run - m_new) + expf(sv - m_new);
            m_run = m_new;
        }
    }
    if ((lane & 7) == 0) { sm[wave][h] = m_run; sl[wave][h] = l_run; }
    __syncthreads();
    float M = sm[0][h], den = 0.f;
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, sm[w][h]);
#pragma unroll
    for (int w = 0; w < NW; ++w) den += sl[w][h] * expf(sm[w][h] - M);      // waves without a key: exp(-inf) = 0
    const float lse = M + logf(den);
    // ---- pass 2: p, ds per pair and head; dq
    float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        int e = beg + wave;
        uint2 kn = make_uint2(0u, 0u), vn = kn;
        if (e < end) {
            const long long ko = (long long)col_idx[e] * C + 4 * lane;
            kn = *reinterpret_cast<const uint2*>(K + ko); vn = *reinterpret_cast<const uint2*>(V + ko);
        }
        for (; e < end; e += NW) {
            const float4 k4 = bf16x4(kn), v4 = bf16x4(vn);
            if (e + NW < end) {
                const long long ko = (long long)col_idx[e + NW] * C + 4 * lane;
                kn = *reinterpret_cast<const uint2*>(K + ko); vn = *reinterpret_cast<const uint2*>(V + ko);
            }
            const float sv = head_sum(k4.x * q4.x + k4.y * q4.y + k4.z * q4.z + k4.w * q4.w);
            const float pj = expf(sv - lse);
            // with dropout ctx = sum_j (p_j m_j) v_j: dL/dp_j = m_j dctx.v_j, D = sum_j p_j m_j dctx.v_j = dctx.ctx still holds, dV_j = (p_j m_j) dctx
            const float ms = attn_drop_factor(drop, e, h);
            const float dp = head_sum(d4.x * v4.x + d4.y * v4.y + d4.z * v4.z + d4.w * v4.w) * ms;
            const float ds = pj * (dp - D);
            gq.x = fmaf(ds, k4.x, gq.x); gq.y = fmaf(ds, k4.y, gq.y); gq.z = fmaf(ds, k4.z, gq.z); gq.w = fmaf(ds, k4.w, gq.w);
            if ((lane & 7) == 0) { pd[(long long)e * 16 + h] = pj * ms; pd[(long long)e * 16 + 8 + h] = ds; }
        }
    }
    *reinterpret_cast<float4*>(&sdq[wave][4 * lane]) = gq;
    __syncthreads();
    if (wave == 0) {
        float4 o = *reinterpret_cast<float4*>(&sdq[0][4 * lane]);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float4 t = *reinterpret_cast<float4*>(&sdq[w][4 * lane]);
            o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
        }
        *reinterpret_cast<float4*>(dq + ro) = make_float4(o.x * dq_scale, o.y * dq_scale, o.z * dq_scale, o.w * dq_scale);
    }
}

// pass over the keys (one wave per key, deterministic, no atomics): dK[s] = sum over the pairs (r, s) of ds . q[r], dV[s] = sum p . dctx[r];
// key_ptr [S+1] / pair_idx [nnz] = the allowed pairs sorted by key (pair ids in CSR order), pair_row [nnz] = query of every pair
__global__ __launch_bounds__(256) void sparse_xattn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ dctx, const float* __restrict__ pd,
                                                                  const int* __restrict__ key_ptr, const int* __restrict__ pair_idx,
                                                                  const int* __restrict__ pair_row, float* __restrict__ dK, float* __restrict__ dV, int S) {
    const int lane = threadIdx.x & 63, h = lane >> 3;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    float4 gk = make_float4(0.f, 0.f, 0.f, 0.f), gv = gk;
    const int beg = key_ptr[s], end = key_ptr[s + 1];
    for (int i = beg; i < end; ++i) {
        const int e = pair_idx[i];
        const long long ro = (long long)pair_row[e] * C + 4 * lane;
        const float pj = pd[(long long)e * 16 + h], ds = pd[(long long)e * 16 + 8 + h];
        const float4 q4 = *reinterpret_cast<const float4*>(q + ro);
        const float4 d4 = *reinterpret_cast<const float4*>(dctx + ro);
        gk.x = fmaf(ds, q4.x, gk.x); gk.y = fmaf(ds, q4.y, gk.y); gk.z = fmaf(ds, q4.z, gk.z); gk.w = fmaf(ds, q4.w, gk.w);
        gv.x = fmaf(pj, d4.x, gv.x); gv.y = fmaf(pj, d4.y, gv.y); gv.z = fmaf(pj, d4.z, gv.z); gv.w = fmaf(pj, d4.w, gv.w);
    }
    *reinterpret_cast<float4*>(dK + (long long)s * C + 4 * lane) = gk;
    *reinterpret_cast<float4*>(dV + (long long)s * C + 4 * lane) = gv;
}


// The same pass for patterns with few keys and long key lists (the self attention: every query sees every query): one 4-wave block per
// key, the key's pairs dealt to the waves round robin (the next pair's indices requested one step ahead), the four partial sums added in
// fixed order through LDS.  (One wave per key walked 300 pairs through three dependent loads each: 168 us for 300 keys.)
__global__ __launch_bounds__(256) void sparse_xattn_bwd_kv_split_kernel(const float* __restrict__ q, const float* __restrict__ dctx,
                                                                        const float* __restrict__ pd, const int* __restrict__ key_ptr,
                                                                        const int* __restrict__ pair_idx, const int* __restrict__ pair_row,
                                                                        float* __restrict__ dK, float* __restrict__ dV, int S) {
    __shared__ float sk[4][C], sv[4][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 3;
    const int s = blockIdx.x;
    float4 gk = make_float4(0.f, 0.f, 0.f, 0.f), gv = gk;
    const int beg = key_ptr[s], end = key_ptr[s + 1];
    int i = beg + wave;
    int e = i < end ? pair_idx[i] : 0;
    int row = i < end ? pair_row[e] : 0;
    for (; i < end; i += 4) {
        const int e_cur = e;
        const long long ro = (long long)row * C + 4 * lane;
        if (i + 4 < end) { e = pair_idx[i + 4]; row = pair_row[e]; }
        const float pj = pd[(long long)e_cur * 16 + h], ds = pd[(long long)e_cur * 16 + 8 + h];
        const float4 q4 = *reinterpret_cast<const float4*>(q + ro);
        const float4 d4 = *reinterpret_cast<const float4*>(dctx + ro);
        gk.x = fmaf(ds, q4.x, gk.x); gk.y = fmaf(ds, q4.y, gk.y); gk.z = fmaf(ds, q4.z, gk.z); gk.w = fmaf(ds, q4.w, gk.w);
        gv.x = fmaf(pj, d4.x, gv.x); gv.y = fmaf(pj, d4.y, gv.y); gv.z = fmaf(pj, d4.z, gv.z); gv.w = fmaf(pj, d4.w, gv.w);
    }
    *reinterpret_cast<float4*>(&sk[wave][4 * lane]) = gk;
    *reinterpret_cast<float4*>(&sv[wave][4 * lane]) = gv;
    __syncthreads();
    const int c = threadIdx.x;
    dK[(long long)s * C + c] = (sk[0][c] + sk[1][c]) + (sk[2][c] + sk[3][c]);
    dV[(long long)s * C + c] = (sv[0][c] + sv[1][c]) + (sv[2][c] + sv[3][c]);
}

}  // namespace

extern "C" int mv2d_self_attn_fwd(const float* qkv, float* ctx, int R, const int* grp_start, int n_grp, void* stream) {
    MV2D_CHECK_ARG(qkv && ctx && R >= 0 && (!grp_start || n_grp >= 1), "mv2d_self_attn_fwd: bad args");
    if (R == 0) return MV2D_OK;
    const int tiles = grp_start ? cdiv(R, 16) + n_grp - 1 : cdiv(R, 16);       // upper bound of sum_g ceil(R_g / 16)
    hipLaunchKernelGGL(self_attn_kernel<false>, dim3(tiles, HEADS), dim3(256), 0, (hipStream_t)stream, qkv, ctx, R,
                       1.0f / sqrtf((float)HD), grp_start, n_grp, 0, 1);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_self_attn_dn_fwd(const float* qkv, float* ctx, int R, int dn_pad, int dn_single, void* stream) {
    MV2D_CHECK_ARG(qkv && ctx && R >= 0, "mv2d_self_attn_dn_fwd: bad args");
    MV2D_CHECK_ARG(dn_pad >= 0 && dn_pad <= R && (dn_pad == 0 || (dn_single >= 1 && dn_pad % dn_single == 0)),
                   "mv2d_self_attn_dn_fwd: dn_pad must be a multiple of dn_single and at most R");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(self_attn_kernel<true>, dim3(cdiv(R, 16), HEADS), dim3(256), 0, (hipStream_t)stream, qkv, ctx, R,
                       1.0f / sqrtf((float)HD), (const int*)nullptr, 0, dn_pad, dn_pad ? dn_single : 1);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

static inline AttnDrop make_drop(float p_drop, unsigned int seed) {
    AttnDrop d{0u, seed, 1.f};
    if (p_drop > 0.f) {
        const double t = (double)p_drop * 4294967296.0;
        d.thr = t >= 4294967295.0 ? 0xffffffffu : (unsigned int)t;
        if (d.thr == 0u) d.thr = 1u;
        d.scale = 1.f / (1.f - p_drop);
    }
    return d;
}

extern "C" int mv2d_sparse_xattn_fwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx,
                                          float* ctx, float* dbg_logits, long long dbg_stride, int R, int empty_nan, float p_drop,
                                          unsigned int seed, void* stream);

extern "C" int mv2d_sparse_xattn_fwd(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx,
                                     float* ctx, float* dbg_logits, long long dbg_stride, int R, int empty_nan, void* stream) {
    return mv2d_sparse_xattn_fwd_drop(q, K, V, row_ptr, col_idx, ctx, dbg_logits, dbg_stride, R, empty_nan, 0.f, 0u, stream);
}

extern "C" int mv2d_sparse_xattn_fwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx,
                                          float* ctx, float* dbg_logits, long long dbg_stride, int R, int empty_nan, float p_drop,
                                          unsigned int seed, void* stream) {
    MV2D_CHECK_ARG(q && K && V && row_ptr && col_idx && ctx && R >= 0, "mv2d_sparse_xattn_fwd: bad args");
    MV2D_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mv2d_sparse_xattn_fwd_drop: p_drop in [0, 1)");
    const AttnDrop drop = make_drop(p_drop, seed);
    if (R == 0) return MV2D_OK;
    // 8 waves x 4-key chunks: best of {2,4,8} waves x {4,8,16} keys with several samples per launch (cfg2_s decoder 0.671 -> 0.655 ms per
    // 6-sample batch, cfg3_t 0.845 -> 0.817); with one sample per launch 8 x 8 was marginally ahead (LOG.md section 8)
    hipLaunchKernelGGL((sparse_xattn_kernel<8, 4>), dim3(R), dim3(64 * 8), 0, (hipStream_t)stream, q, (const unsigned short*)K, (const unsigned short*)V,
                       row_ptr, col_idx, ctx, dbg_logits, dbg_stride, R, empty_nan, drop);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_sparse_xattn_bwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                                          const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                                          float* dq, float* dK, float* dV, int R, int S, float p_drop, unsigned int seed, void* stream);

extern "C" int mv2d_sparse_xattn_bwd(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                                     const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                                     float* dq, float* dK, float* dV, int R, int S, void* stream) {
    return mv2d_sparse_xattn_bwd_drop(q, K, V, row_ptr, col_idx, ctx, dctx, key_ptr, pair_idx, pair_row, pair_ws, dq, dK, dV, R, S, 0.f, 0u, stream);
}

extern "C" int mv2d_sparse_xattn_bwd_ex(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                                        const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                                        float* dq, float* dK, float* dV, int R, int S, float p_drop, unsigned int seed, int long_rows, float dq_scale, void* stream);

extern "C" int mv2d_sparse_xattn_bwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                                          const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                                          float* dq, float* dK, float* dV, int R, int S, float p_drop, unsigned int seed, void* stream) {
    return mv2d_sparse_xattn_bwd_ex(q, K, V, row_ptr, col_idx, ctx, dctx, key_ptr, pair_idx, pair_row, pair_ws, dq, dK, dV, R, S, p_drop, seed, 0, 1.f, stream);
}

// long_rows != 0: the pattern has hundreds of keys per query and of queries per key (the decoder's self attention): 16 waves per query in
// the query pass, one block per key in the key pass.  The results differ from long_rows = 0 by summation order only.
// dq_scale: dq is multiplied by it (the 1 / sqrt(d) of q = (x W^T + b) / sqrt(d): the gradient w.r.t. the unscaled projection).
extern "C" int mv2d_sparse_xattn_bwd_ex(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                                        const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                                        float* dq, float* dK, float* dV, int R, int S, float p_drop, unsigned int seed, int long_rows, float dq_scale, void* stream) {
    MV2D_CHECK_ARG(q && K && V && row_ptr && col_idx && ctx && dctx && key_ptr && pair_idx && pair_row && pair_ws && dq && dK && dV,
                   "mv2d_sparse_xattn_bwd: null pointer");
    MV2D_CHECK_ARG(R >= 0 && S >= 0 && p_drop >= 0.f && p_drop < 1.f, "mv2d_sparse_xattn_bwd: bad sizes / p_drop");
    const AttnDrop drop = make_drop(p_drop, seed);
    if (R > 0) {
        if (long_rows)
            hipLaunchKernelGGL(sparse_xattn_bwd_q_kernel<16>, dim3(R), dim3(1024), 0, (hipStream_t)stream, q, (const unsigned short*)K,
                               (const unsigned short*)V, row_ptr, col_idx, ctx, dctx, dq, pair_ws, R, drop, dq_scale);
        else
            hipLaunchKernelGGL(sparse_xattn_bwd_q_kernel<4>, dim3(R), dim3(256), 0, (hipStream_t)stream, q, (const unsigned short*)K,
                               (const unsigned short*)V, row_ptr, col_idx, ctx, dctx, dq, pair_ws, R, drop, dq_scale);
    }
    if (S > 0) {
        if (long_rows)
            hipLaunchKernelGGL(sparse_xattn_bwd_kv_split_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, q, dctx, pair_ws, key_ptr, pair_idx, pair_row,
                               dK, dV, S);
        else
            hipLaunchKernelGGL(sparse_xattn_bwd_kv_kernel, dim3(cdiv(S, 4)), dim3(256), 0, (hipStream_t)stream, q, dctx, pair_ws, key_ptr, pair_idx, pair_row,
                               dK, dV, S);
    }
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
