// Exact-fp32 MFMA GEMM for the small-M (M = #queries) per-query ops of the MV2D decoder (gfx950).
//
//   C[M,N] = epilogue( A[M,K] (fp32) x W[N,K]^T (fp32, nn.Linear layout) + bias )
//
// v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (exact f32, the f32 vector rate), so the
// query-side state of the decoder (in_proj of q, out_proj, FFN, cls/reg branches, query-generator fcs:
// MU/petr_transformer.py:358-363,503-508, cross_attention_head.py:118-146, query_generator.py:318,196)
// keeps the reference's fp32 numerics; only the K/V side runs in bf16.
//
// M is tiny (300-900 rows): the problem is a latency problem, not a bandwidth or MFMA problem (round-1 profile:
// an LDS-staged 64x32x32 tile loop paid one dependent global-load round trip per 32-wide K step, ~1 us each).
// So there is NO LDS and no barrier here: every wave owns one 16x16 output tile and issues the loads of a whole
// 256-wide K pass (16 + 16 float4 per lane, fragment-shaped: lane (r, g) reads row r, k = 16c + 4g .. +3) before
// the first MFMA; the k index is spread over (MFMA step, lane group) by the same bijection for A and W.
// Grid: (N/32, M/32, groups*split_k) blocks of 2x2 waves -> 304 waves for a 300x256 output.
// split-K slices are written as separate partial slabs that the following row kernel sums in a fixed order.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int KP = 256;                       // K elements per pass (all in flight)
constexpr int NCH = KP / 16;                  // float4 chunks per lane per operand per pass

struct Params {
    const float* A; const float* A2; int n_split;   // columns n >= n_split read A2
    const float* W; const float* bias;
    int M, N, K, lda, ldw;
    int k_chunk;                  // K range per split-K slice, multiple of 32
    int act;                      // 0 none, 1 relu
    float scale;                  // v = (acc + bias) * scale
    float clamp;                  // > 0: v = min(max(v, -clamp), clamp)  (query_generator.py:369)
    void* C; int c_bf16; int ldc; long long c_slice_stride;   // slice z writes C + z * c_slice_stride
    int split_k;                  // blockIdx.z = group * split_k + slice
    long long a_gs, w_gs, b_gs, c_gs;   // per-group element strides (grouped GEMM: one weight set per decoder layer)
    int dbg;                      // MV2D_F32_DBG ablation: 1 = loads only (no MFMA), 2 = MFMA only (no loads), 3 = neither
};

template <int NT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(Params p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.y * 32 + (wave >> 1) * 16, n0 = blockIdx.x * (32 * NT) + (wave & 1) * (16 * NT);
    if (m0 >= p.M || n0 >= p.N) return;
    const int grp = blockIdx.z / p.split_k, slice = blockIdx.z - grp * p.split_k;
    const int kbeg = slice * p.k_chunk, kend = min(p.K, kbeg + p.k_chunk);
    const float* Abase = ((p.n_split > 0 && n0 >= p.n_split) ? p.A2 : p.A) + grp * p.a_gs;
    const int arow = min(m0 + fr, p.M - 1);
    const float* ap = Abase + (long long)arow * p.lda + 4 * fg;
    const float* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int wrow = n0 + 16 * t + fr;
        wp[t] = p.W + grp * p.w_gs + (long long)(wrow < p.N ? wrow : p.N - 1) * p.ldw + 4 * fg;
    }

    // two accumulator chains per tile (even / odd 16-wide k chunks): the dependent-accumulator latency of
    // v_mfma_f32_16x16x4_f32 is 40 cycles against a 32-cycle issue interval
    f32x4_t acc[NT], accb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accb[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int k0 = kbeg; k0 < kend; k0 += KP) {
        const int nch = min(NCH, (kend - k0) / 16);     // k ranges are multiples of 32 -> whole 16-chunks
        float4 a[NCH], w[NT][NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < nch && !(p.dbg & 2)) {
                a[c] = *reinterpret_cast<const float4*>(ap + k0 + 16 * c);
#pragma unroll
                for (int t = 0; t < NT; ++t) w[t][c] = *reinterpret_cast<const float4*>(wp[t] + k0 + 16 * c);
            } else {
                a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < NT; ++t) w[t][c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (p.dbg & 1) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][0] += a[c].x * w[t][c].x + a[c].y * w[t][c].y + a[c].z * w[t][c].z + a[c].w * w[t][c].w;
            continue;
        }
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, w[t][c].x, acc[t], 0, 0, 0);
                accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c + 1].x, w[t][c + 1].x, accb[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, w[t][c].y, acc[t], 0, 0, 0);
                accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c + 1].y, w[t][c + 1].y, accb[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, w[t][c].z, acc[t], 0, 0, 0);
                accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c + 1].z, w[t][c + 1].z, accb[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, w[t][c].w, acc[t], 0, 0, 0);
                accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c + 1].w, w[t][c + 1].w, accb[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += accb[t][r];

    // lane holds C[m = m0 + 4*fg + r][n = n0 + 16t + fr]
    unsigned char* Cz = reinterpret_cast<unsigned char*>(p.C) + ((long long)slice * p.c_slice_stride + grp * p.c_gs) * (p.c_bf16 ? 2 : 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 16 * t + fr;
        if (n >= p.N) continue;
        const float bn = (p.bias && slice == 0) ? p.bias[grp * p.b_gs + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + fg * 4 + r;
            if (m >= p.M) continue;
            float v = (acc[t][r] + bn) * p.scale;
            if (p.act == 1) v = relu_f(v);
            if (p.clamp > 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
            const long long o = (long long)m * p.ldc + n;
            if (p.c_bf16) reinterpret_cast<unsigned short*>(Cz)[o] = f32_to_bf16(v);
            else reinterpret_cast<float*>(Cz)[o] = v;
        }
    }
}

}  // namespace

extern "C" int mv2d_gemm_f32(const float* A, const float* A2, int n_split, const float* W, const float* bias,
                             int M, int N, int K, int lda, int ldw, int split_k, int act, float scale, float clamp, void* C,
                             int c_bf16, int ldc, long long c_slice_stride, int groups, long long a_gs, long long w_gs,
                             long long b_gs, long long c_gs, void* stream) {
    MV2D_CHECK_ARG(A && W && C, "mv2d_gemm_f32: null A/W/C");
    MV2D_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 32) == 0, "mv2d_gemm_f32: K must be a positive multiple of 32");
    MV2D_CHECK_ARG((lda % 4) == 0 && (ldw % 4) == 0, "mv2d_gemm_f32: lda/ldw must be multiples of 4 (16-byte rows)");
    MV2D_CHECK_ARG(split_k >= 1 && (K % (split_k * 32)) == 0, "mv2d_gemm_f32: K must divide into split_k slices of multiples of 32");
    MV2D_CHECK_ARG(n_split == 0 || (A2 && (n_split % 32) == 0), "mv2d_gemm_f32: n_split must be a multiple of 32 with A2 set");
    MV2D_CHECK_ARG(groups >= 1 && (groups == 1 || split_k == 1), "mv2d_gemm_f32: groups > 1 needs split_k == 1");
    if (M == 0) return MV2D_OK;
    Params p;
    p.clamp = clamp; p.split_k = split_k; p.a_gs = a_gs; p.w_gs = w_gs; p.b_gs = b_gs; p.c_gs = c_gs;
    p.A = A; p.A2 = A2; p.n_split = n_split; p.W = W; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
    p.k_chunk = K / split_k; p.act = act; p.scale = scale; p.C = C; p.c_bf16 = c_bf16; p.ldc = ldc;
    p.c_slice_stride = c_slice_stride;
    p.dbg = 0;
    // (two column tiles per wave sharing one A fragment -- gemm_f32_kernel<2> -- measured no gain in round 1; one tile per wave)
    dim3 grid(cdiv(N, 32), cdiv(M, 32), split_k * groups);
    hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
