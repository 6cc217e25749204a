// FlattenMHSelfAttention core (MU/petr_transformer.py:317-370) on bf16 MFMAs in split precision, keys / values through LDS.
//
// The round-1 kernel (attention.hip: exact fp32 v_mfma_f32_16x16x4_f32, one block per (16 queries, head), K / V fragments straight
// from the [R,768] qkv rows) spent its 18 us per layer on fragment-shaped gathers of 3 KB-strided rows: every (16-query, head) block
// re-read its sample's K / V as 16-byte pieces, 42 MB through the L2s per call for 7.4 MB of qkv.  Here:
//   * block = (64 queries, head, sample): the sample's K_h / V_h rows are staged ONCE per block in chunks of 128 keys with whole
//     128-byte row loads, split into bf16 hi / lo pairs (w = hi + lo, 2^-17 relative) and written as MFMA-ready LDS images:
//     K [key][32 d] with the 16-byte slots XOR-swizzled, V transposed [d][key] so that the P.V contraction reads keys contiguously;
//   * S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_16x16x32_bf16 with the three cross terms hi.hi + lo.hi + hi.lo (fp32-class,
//     1e-5 against fp64 like the exact-fp32 kernel, 3/16 of its matrix-pipe time); a wave owns 16 queries and walks the keys in
//     steps of 32 with an online softmax; the k index of the P.V MFMA is permuted (keys 4g..4g+3 | 16+4g..16+4g+3 per lane group)
//     so that P feeds it straight from the S^T accumulators;
//   * attention stays inside a sample (grp_start), tiles are counted from the sample's first row: a sample's result does not depend
//     on the batch it is in.  DN: the denoising mask of prepare_for_dn (RH/mv2d_s_head.py:95-107) evaluated from two integers.
#include "common.h"

namespace {

constexpr int C = 256, HD = 32, KC = 128;                 // channels, head dim, keys per LDS chunk
constexpr int VPITCH = KC * 2 + 16;                       // bytes per row of the transposed V images (pad: conflict-free 8-byte reads)
typedef q16x8_t sa_bf16x8;      // common.h "q16": fp16 pairs since round 5
union SFrag { uint4 u; sa_bf16x8 v; uint2 h[2]; };

__device__ __forceinline__ void sa_split4(const float4& v, uint2& hi, uint2& lo) {
    split_q16x4(v, hi, lo);
}

template <bool DN>
__global__ __launch_bounds__(256) void self_attn_x3_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int R, float scale,
                                                           const int* __restrict__ grp_start, int dn_pad, int dn_single, int nx, int npair) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KC * 64 + 2 * HD * VPITCH];
    unsigned char* Kh = smem;                              // [KC][64 B]: 16-byte slot c of key k at slot c ^ ((k >> 2) & 3)
    unsigned char* Kl = smem + KC * 64;
    unsigned char* Vh = smem + 2 * KC * 64;                // [32 d][VPITCH]: key k of channel d at d * VPITCH + 2 k
    unsigned char* Vl = Vh + HD * VPITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    // XCD-aware block map (block b runs on XCD b % 8, each XCD has its own L2): the nx query blocks of one (sample, head) pair read the
    // same K / V slice, so pair p goes to XCD p % 8 and its query blocks to consecutive slots there.  Speed only; any map is correct.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, qb = slot % nx, pair = (slot / nx) * 8 + xcd;
    if (pair >= npair) return;                             // (block-uniform; the pair count is padded to a multiple of 8)
    const int h = pair & 7, grp = pair >> 3;
    if (grp_start) {
        const int gs = grp_start[grp], ge = grp_start[grp + 1];
        qkv += (long long)gs * 768; ctx += (long long)gs * C; R = ge - gs;
    }
    const int q0 = qb * 64 + wave * 16;                    // this wave's 16 queries (rows of the sample)
    if (qb * 64 >= R) return;                              // (block-uniform)
    const bool wave_on = q0 < R;
    // B operand of S^T = K.Q^T: lane (query n, g): q[query][32 h + 8 g .. + 7] * scale, hi / lo
    SFrag qh, ql;
    {
        const float* qp = qkv + (long long)min(q0 + n, R - 1) * 768 + h * HD + 8 * g;
        float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
        a = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
        b = make_float4(b.x * scale, b.y * scale, b.z * scale, b.w * scale);
        sa_split4(a, qh.h[0], ql.h[0]);
        sa_split4(b, qh.h[1], ql.h[1]);
    }
    float m_run = -INFINITY, l_run = 0.f;                  // of query n (identical in the four lanes sharing n after the reductions)
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};     // O^T: lane (query n, channels 4g..4g+3 of d-tile 0 / 1)
    const int qrow = q0 + n;
    // denoising group of this lane's query as a key range [glo, ghi) (empty for a matched query)
    const int qg = (DN && qrow < dn_pad) ? (int)(((float)qrow + 0.5f) / (float)dn_single) : 0;
    const int glo = (DN && qrow < dn_pad) ? qg * dn_single : 0, ghi = (DN && qrow < dn_pad) ? glo + dn_single : 0;
    // global loads of a chunk: thread -> (row tid >> 3 + 32 i, channels 4 (tid & 7) ..): whole 128-byte rows per 8 threads.  The loads of
    // chunk c + 1 are in flight while chunk c is computed (qkv was just written by another kernel: an Infinity-Cache round trip).
    float4 kreg[KC / 32], vreg[KC / 32];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KC / 32; ++i) {
            const long long row = (long long)min(k0 + (tid >> 3) + 32 * i, R - 1) * 768 + h * HD + 4 * (tid & 7);
            kreg[i] = *reinterpret_cast<const float4*>(qkv + row + C);
            vreg[i] = *reinterpret_cast<const float4*>(qkv + row + 2 * C);
        }
    };
    load_chunk(0);
    for (int k0 = 0; k0 < R; k0 += KC) {
        if (k0) __syncthreads();                           // everybody is done with the previous chunk
#pragma unroll
        for (int i = 0; i < KC / 32; ++i) {
            const int kr = (tid >> 3) + 32 * i, dq = tid & 7;
            uint2 hi, lo;
            sa_split4(kreg[i], hi, lo);
            const int koff = kr * 64 + ((((dq >> 1) ^ ((kr >> 2) & 3))) << 4) + (dq & 1) * 8;
            *reinterpret_cast<uint2*>(Kh + koff) = hi;
            *reinterpret_cast<uint2*>(Kl + koff) = lo;
            sa_split4(vreg[i], hi, lo);
            const int voff = (4 * dq) * VPITCH + 2 * kr;
            *reinterpret_cast<unsigned short*>(Vh + voff) = (unsigned short)(hi.x & 0xffffu);
            *reinterpret_cast<unsigned short*>(Vh + voff + VPITCH) = (unsigned short)(hi.x >> 16);
            *reinterpret_cast<unsigned short*>(Vh + voff + 2 * VPITCH) = (unsigned short)(hi.y & 0xffffu);
            *reinterpret_cast<unsigned short*>(Vh + voff + 3 * VPITCH) = (unsigned short)(hi.y >> 16);
            *reinterpret_cast<unsigned short*>(Vl + voff) = (unsigned short)(lo.x & 0xffffu);
            *reinterpret_cast<unsigned short*>(Vl + voff + VPITCH) = (unsigned short)(lo.x >> 16);
            *reinterpret_cast<unsigned short*>(Vl + voff + 2 * VPITCH) = (unsigned short)(lo.y & 0xffffu);
            *reinterpret_cast<unsigned short*>(Vl + voff + 3 * VPITCH) = (unsigned short)(lo.y >> 16);
        }
        if (k0 + KC < R) load_chunk(k0 + KC);
        __syncthreads();
        if (!wave_on) continue;
        const int kend = min(KC, R - k0);
        for (int ks = 0; ks < kend; ks += 32) {
            // ---- S^T of two key tiles: rows = keys ks + 16 t + 4g + i, column = query n
            f32x4_t s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kr = ks + 16 * t + n;             // A operand: lane (key n of the tile, g): channels 8g..8g+7
                SFrag kh, kl;
                const int off = kr * 64 + ((g ^ ((kr >> 2) & 3)) << 4);
                kh.u = *reinterpret_cast<const uint4*>(Kh + off);
                kl.u = *reinterpret_cast<const uint4*>(Kl + off);
                f32x4_t a = {0.f, 0.f, 0.f, 0.f};
                a = mfma_q16_16x16x32(kl.v, qh.v, a, 0, 0, 0);
                a = mfma_q16_16x16x32(kh.v, ql.v, a, 0, 0, 0);
                a = mfma_q16_16x16x32(kh.v, qh.v, a, 0, 0, 0);
                s[t] = a;
            }
            float p[8], tmax = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = k0 + ks + 16 * t + 4 * g + i;
                    bool vis = key < R;
                    // (bitwise, not short-circuit: the mask stays a lane mask, the MFMAs around it run in uniform control flow)
                    if (DN) vis = vis & ((key >= dn_pad) | ((key >= glo) & (key < ghi)));
                    p[4 * t + i] = vis ? s[t][i] : -INFINITY;
                    tmax = fmaxf(tmax, p[4 * t + i]);
                }
            {   // over the four 16-lane rows (the key groups of a step): v_permlane16_swap / v_permlane32_swap, no LDS crossbar trips
                const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                tmax = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
                const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
                tmax = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
            }
            const float m_new = fmaxf(m_run, tmax);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;    // nothing visible so far: (m, l, o) stay (-inf, 0, 0), no NaN
            const float alpha = __expf(m_run - m_use);
            float psum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { p[j] = __expf(p[j] - m_use); psum += p[j]; }
            {
                const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(psum), __float_as_uint(psum), false, false);
                psum = __uint_as_float(a[0]) + __uint_as_float(a[1]);
                const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(psum), __float_as_uint(psum), false, false);
                psum = __uint_as_float(b[0]) + __uint_as_float(b[1]);
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 4; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            // ---- O^T += V^T . P^T: B operand = P of this lane's 8 keys (k slot (g, e) <-> key 4g + e | 16 + 4g + e - 4), hi / lo
            SFrag ph, pl;
            sa_split4(make_float4(p[0], p[1], p[2], p[3]), ph.h[0], pl.h[0]);
            sa_split4(make_float4(p[4], p[5], p[6], p[7]), ph.h[1], pl.h[1]);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                SFrag vh, vl;                               // A operand: lane (channel 16 m + n, g): the same 8 keys
                const int off = (16 * m + n) * VPITCH + 2 * (ks + 4 * g);
                vh.h[0] = *reinterpret_cast<const uint2*>(Vh + off);
                vh.h[1] = *reinterpret_cast<const uint2*>(Vh + off + 32);
                vl.h[0] = *reinterpret_cast<const uint2*>(Vl + off);
                vl.h[1] = *reinterpret_cast<const uint2*>(Vl + off + 32);
                f32x4_t& o = m ? o1 : o0;
                o = mfma_q16_16x16x32(vl.v, ph.v, o, 0, 0, 0);
                o = mfma_q16_16x16x32(vh.v, pl.v, o, 0, 0, 0);
                o = mfma_q16_16x16x32(vh.v, ph.v, o, 0, 0, 0);
            }
        }
    }
    if (wave_on && qrow < R) {
        const float inv = 1.0f / l_run;
        float* op = ctx + (long long)qrow * C + h * HD + 4 * g;
        *reinterpret_cast<float4*>(op) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
        *reinterpret_cast<float4*>(op + 16) = make_float4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
    }
}

}  // namespace

// max_grp_rows: upper bound of the rows of one sample (sizes the grid; 0: R)
extern "C" int mv2d_self_attn_x3_fwd(const float* qkv, float* ctx, int R, const int* grp_start, int n_grp, int max_grp_rows, int dn_pad,
                                     int dn_single, void* stream) {
    MV2D_CHECK_ARG(qkv && ctx && R >= 0 && (!grp_start || n_grp >= 1), "mv2d_self_attn_x3_fwd: bad args");
    MV2D_CHECK_ARG(dn_pad == 0 || (!grp_start && dn_single >= 1 && dn_pad % dn_single == 0 && dn_pad <= R),
                   "mv2d_self_attn_x3_fwd: the denoising mask needs one sample, dn_pad a multiple of dn_single and at most R");
    MV2D_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ctx & 15) == 0, "mv2d_self_attn_x3_fwd: operands must be 16-byte aligned");
    if (R == 0) return MV2D_OK;
    const int rows = (grp_start && max_grp_rows > 0) ? min(max_grp_rows, R) : R;
    const int nx = cdiv(rows, 64), npair = 8 * (grp_start ? n_grp : 1);
    const dim3 grid(nx * 8 * cdiv(npair, 8));
    const float scale = 1.0f / sqrtf((float)HD);
    if (dn_pad > 0)
        hipLaunchKernelGGL(self_attn_x3_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, qkv, ctx, R, scale, grp_start, dn_pad, dn_single, nx, npair);
    else
        hipLaunchKernelGGL(self_attn_x3_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, qkv, ctx, R, scale, grp_start, 0, 1, nx, npair);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
