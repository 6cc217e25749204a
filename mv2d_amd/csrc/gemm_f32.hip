// Exact-fp32 MFMA GEMM for the small-M (M = #queries) per-query ops of the MV2D decoder (gfx950).
//
//   C[M,N] = epilogue( A[M,K] (fp32) x W[N,K]^T (fp32, nn.Linear layout) + bias )
//
// v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (exact f32, the f32 vector rate), so the
// query-side state of the decoder (in_proj of q, out_proj, FFN, cls/reg branches, query-generator fcs:
// MU/petr_transformer.py:358-363,503-508, cross_attention_head.py:118-146, query_generator.py:318,196)
// keeps the reference's fp32 numerics; only the K/V side runs in bf16.
//
// Tile 64x32x32, 4 waves, wave w owns rows 16w..16w+15 x 32 columns.  M is tiny (300-900 rows) so the
// grid is made of many small tiles (and optional split-K slices written as separate partial slabs that the
// following row kernel sums in a fixed order -> deterministic).
#include "common.h"

namespace {

constexpr int BM = 64, BN = 32, BK = 32;
constexpr int ROW_BYTES = BK * 4;            // 128 B
constexpr int A_BYTES = BM * ROW_BYTES;      // 8 KiB
constexpr int B_BYTES = BN * ROW_BYTES;      // 4 KiB

struct Params {
    const float* A; const float* A2; int n_split;   // columns n >= n_split read A2
    const float* W; const float* bias;
    int M, N, K, lda, ldw;
    int k_chunk;                  // K range per blockIdx.z slice (split-K), multiple of BK
    int act;                      // 0 none, 1 relu
    float scale;                  // v = (acc + bias) * scale
    float clamp;                  // > 0: v = min(max(v, -clamp), clamp)  (query_generator.py:369)
    void* C; int c_bf16; int ldc; long long c_slice_stride;   // slice z writes C + z * c_slice_stride
    int split_k;                  // blockIdx.z = group * split_k + slice
    long long a_gs, w_gs, b_gs, c_gs;   // per-group element strides (grouped GEMM: one weight set per decoder layer)
};

__device__ __forceinline__ int lds_off(int row, int slot) { return row * ROW_BYTES + ((slot ^ (row & 7)) << 4); }

__global__ __launch_bounds__(256) void gemm_f32_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * A_BYTES;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int grp = blockIdx.z / p.split_k, slice = blockIdx.z - grp * p.split_k;
    const int kbeg = slice * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Abase = ((p.n_split > 0 && n0 >= p.n_split) ? p.A2 : p.A) + grp * p.a_gs;
    const float* Wbase = p.W + grp * p.w_gs;

    int a_row[2], a_slot[2];
    long long a_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = tid + 256 * i;
        a_row[i] = c >> 3; a_slot[i] = c & 7;
        int m = m0 + a_row[i];
        a_src[i] = (long long)(m < p.M ? m : p.M - 1) * p.lda;
    }
    const int b_row = tid >> 3, b_slot = tid & 7;
    const bool b_ok = (n0 + b_row) < p.N;
    const long long b_src = (long long)(b_ok ? n0 + b_row : p.N - 1) * p.ldw;

    float4 ra[2], rb;
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(Abase + a_src[i] + k0 + a_slot[i] * 4);
        rb = *reinterpret_cast<const float4*>(Wbase + b_src + k0 + b_slot * 4);
        if (!b_ok) rb = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<float4*>(As + buf * A_BYTES + lds_off(a_row[i], a_slot[i])) = ra[i];
        *reinterpret_cast<float4*>(Bs + buf * B_BYTES + lds_off(b_row, b_slot)) = rb;
    };

    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const int nk = (kend - kbeg) / BK;
    const int fr = lane & 15, fg = lane >> 4;
    if (nk > 0) {
        load_tile(kbeg);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kbeg + (kt + 1) * BK);
        const unsigned char* a_t = As + buf * A_BYTES;
        const unsigned char* b_t = Bs + buf * B_BYTES;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            // lane group g holds k = kc*16 + 4g .. +3; MFMA step j contracts {kc*16 + 4g + j : g = 0..3}
            // (any bijection of k onto (step, lane-group) is valid as long as A and B agree)
            float4 a = *reinterpret_cast<const float4*>(a_t + lds_off(wave * 16 + fr, kc * 4 + fg));
            float4 b0 = *reinterpret_cast<const float4*>(b_t + lds_off(fr, kc * 4 + fg));
            float4 b1 = *reinterpret_cast<const float4*>(b_t + lds_off(16 + fr, kc * 4 + fg));
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc[1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    unsigned char* Cz = reinterpret_cast<unsigned char*>(p.C) + ((long long)slice * p.c_slice_stride + grp * p.c_gs) * (p.c_bf16 ? 2 : 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + fr;
        if (n >= p.N) continue;
        const float bn = (p.bias && slice == 0) ? p.bias[grp * p.b_gs + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + fg * 4 + r;
            if (m >= p.M) continue;
            float v = (acc[j][r] + bn) * p.scale;
            if (p.act == 1) v = fmaxf(v, 0.f);
            if (p.clamp > 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
            long long o = (long long)m * p.ldc + n;
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
    MV2D_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % BK) == 0, "mv2d_gemm_f32: K must be a positive multiple of 32");
    MV2D_CHECK_ARG((lda % 4) == 0 && (ldw % 4) == 0, "mv2d_gemm_f32: lda/ldw must be multiples of 4 (16-byte rows)");
    MV2D_CHECK_ARG(split_k >= 1 && (K % (split_k * BK)) == 0, "mv2d_gemm_f32: K must divide into split_k slices of multiples of 32");
    MV2D_CHECK_ARG(n_split == 0 || (A2 && (n_split % BN) == 0), "mv2d_gemm_f32: n_split must be a multiple of 32 with A2 set");
    MV2D_CHECK_ARG(groups >= 1 && (groups == 1 || split_k == 1), "mv2d_gemm_f32: groups > 1 needs split_k == 1");
    if (M == 0) return MV2D_OK;
    Params p;
    p.clamp = clamp; p.split_k = split_k; p.a_gs = a_gs; p.w_gs = w_gs; p.b_gs = b_gs; p.c_gs = c_gs;
    p.A = A; p.A2 = A2; p.n_split = n_split; p.W = W; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw;
    p.k_chunk = K / split_k; p.act = act; p.scale = scale; p.C = C; p.c_bf16 = c_bf16; p.ldc = ldc;
    p.c_slice_stride = c_slice_stride;
    dim3 grid(cdiv(N, BN), cdiv(M, BM), split_k * groups);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
