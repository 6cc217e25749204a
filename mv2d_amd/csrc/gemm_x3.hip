// fp32-class GEMM on the bf16 matrix cores ("bf16x3" split precision) for the small-M per-query ops (gfx950).
//
//   C[M,N] = epilogue( A[M,K] (fp32) x W[N,K]^T (fp32, given as a bf16 hi/lo pair) + bias )
//
// a = a_hi + a_lo, w = w_hi + w_lo with x_hi = bf16(x), x_lo = bf16(x - x_hi);  a.w ~= a_hi.w_hi + a_lo.w_hi + a_hi.w_lo:
// three v_mfma_f32_16x16x32_bf16 with fp32 accumulation, every partial product exact in fp32, dropped term ~2^-18 relative
// -> ~1e-5 relative error (vs 4e-3 for plain bf16), at 3/16 of the matrix-core time of v_mfma_f32_16x16x4_f32.
// Round-1 measurements: the exact-f32 MFMA chain (64 dependent 40-cycle MFMAs per 16x16x256 tile) was the critical path of
// every per-query GEMM once the load latency had been removed.
// Same structure as gemm_f32.hip: no LDS, one 16x16 output tile per wave, a whole 256-wide K pass in flight.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
constexpr int KP = 256;

struct Params {
    const float* A; const float* A2; int n_split;
    const unsigned short* Whi; const unsigned short* Wlo; const float* bias;
    int M, N, K, lda, ldw;
    int k_chunk, act;
    float scale, clamp;
    void* C; int c_bf16; int ldc; long long c_slice_stride;
    int split_k;
    long long a_gs, w_gs, b_gs, c_gs;
};

union Pack8 { uint4 u; mfma_bf16x8 v; };

// split 8 fp32 into bf16 hi / lo fragments
__device__ __forceinline__ void split8(const float4& x0, const float4& x1, mfma_bf16x8& hi, mfma_bf16x8& lo) {
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = f32_to_bf16(f[i]);
        l[i] = f32_to_bf16(f[i] - __uint_as_float(h[i] << 16));
    }
    Pack8 ph, pl;
    ph.u = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    pl.u = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    hi = ph.v; lo = pl.v;
}

__global__ __launch_bounds__(256) void gemm_x3_kernel(Params p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.y * 32 + (wave >> 1) * 16, n0 = blockIdx.x * 32 + (wave & 1) * 16;
    if (m0 >= p.M || n0 >= p.N) return;
    const int grp = blockIdx.z / p.split_k, slice = blockIdx.z - grp * p.split_k;
    const int kbeg = slice * p.k_chunk, kend = min(p.K, kbeg + p.k_chunk);
    const float* Abase = ((p.n_split > 0 && n0 >= p.n_split) ? p.A2 : p.A) + grp * p.a_gs;
    const float* ap = Abase + (long long)min(m0 + fr, p.M - 1) * p.lda + 8 * fg;
    const long long wrow = grp * p.w_gs + (long long)min(n0 + fr, p.N - 1) * p.ldw + 8 * fg;
    const unsigned short* whp = p.Whi + wrow;
    const unsigned short* wlp = p.Wlo + wrow;

    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // two chains: hi.hi  and the two correction terms
    for (int k0 = kbeg; k0 < kend; k0 += KP) {
        const int nst = min(KP / 32, (kend - k0) / 32);
        float4 a0[KP / 32], a1[KP / 32];
        Pack8 wh[KP / 32], wl[KP / 32];
#pragma unroll
        for (int j = 0; j < KP / 32; ++j) {
            if (j < nst) {
                a0[j] = *reinterpret_cast<const float4*>(ap + k0 + 32 * j);
                a1[j] = *reinterpret_cast<const float4*>(ap + k0 + 32 * j + 4);
                wh[j].u = *reinterpret_cast<const uint4*>(whp + k0 + 32 * j);
                wl[j].u = *reinterpret_cast<const uint4*>(wlp + k0 + 32 * j);
            }
        }
#pragma unroll
        for (int j = 0; j < KP / 32; ++j) {
            if (j < nst) {
                mfma_bf16x8 ah, al;
                split8(a0[j], a1[j], ah, al);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[j].v, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[j].v, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[j].v, acc1, 0, 0, 0);
            }
        }
    }
    const int n = n0 + fr;
    if (n >= p.N) return;
    unsigned char* Cz = reinterpret_cast<unsigned char*>(p.C) + ((long long)slice * p.c_slice_stride + grp * p.c_gs) * (p.c_bf16 ? 2 : 4);
    const float bn = (p.bias && slice == 0) ? p.bias[grp * p.b_gs + n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + fg * 4 + r;
        if (m >= p.M) continue;
        float v = ((acc0[r] + acc1[r]) + bn) * p.scale;
        if (p.act == 1) v = relu_f(v);
        if (p.clamp > 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
        const long long o = (long long)m * p.ldc + n;
        if (p.c_bf16) reinterpret_cast<unsigned short*>(Cz)[o] = f32_to_bf16(v);
        else reinterpret_cast<float*>(Cz)[o] = v;
    }
}

__global__ void split_bf16x2_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = x[i];
    const unsigned short h = f32_to_bf16(f);
    hi[i] = h;
    lo[i] = f32_to_bf16(f - __uint_as_float(((unsigned int)h) << 16));
}

}  // namespace

extern "C" int mv2d_split_bf16x2(const float* x, void* hi, void* lo, long long n, void* stream) {
    MV2D_CHECK_ARG(x && hi && lo && n >= 0, "mv2d_split_bf16x2: bad args");
    if (n == 0) return MV2D_OK;
    hipLaunchKernelGGL(split_bf16x2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)hi,
                       (unsigned short*)lo, n);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_gemm_x3(const float* A, const float* A2, int n_split, const void* Whi, const void* Wlo, const float* bias, int M, int N,
                            int K, int lda, int ldw, int split_k, int act, float scale, float clamp, void* C, int c_bf16, int ldc,
                            long long c_slice_stride, int groups, long long a_gs, long long w_gs, long long b_gs, long long c_gs,
                            void* stream) {
    MV2D_CHECK_ARG(A && Whi && Wlo && C, "mv2d_gemm_x3: null A/W/C");
    MV2D_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 32) == 0, "mv2d_gemm_x3: K must be a positive multiple of 32");
    MV2D_CHECK_ARG((lda % 4) == 0 && (ldw % 8) == 0, "mv2d_gemm_x3: lda must be a multiple of 4 and ldw of 8 (16-byte rows)");
    MV2D_CHECK_ARG(split_k >= 1 && (K % (split_k * 32)) == 0, "mv2d_gemm_x3: K must divide into split_k slices of multiples of 32");
    MV2D_CHECK_ARG(n_split == 0 || (A2 && (n_split % 32) == 0), "mv2d_gemm_x3: n_split must be a multiple of 32 with A2 set");
    MV2D_CHECK_ARG(groups >= 1 && (groups == 1 || split_k == 1), "mv2d_gemm_x3: groups > 1 needs split_k == 1");
    if (M == 0) return MV2D_OK;
    Params p;
    p.A = A; p.A2 = A2; p.n_split = n_split; p.Whi = (const unsigned short*)Whi; p.Wlo = (const unsigned short*)Wlo; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.k_chunk = K / split_k; p.act = act; p.scale = scale; p.clamp = clamp;
    p.C = C; p.c_bf16 = c_bf16; p.ldc = ldc; p.c_slice_stride = c_slice_stride; p.split_k = split_k;
    p.a_gs = a_gs; p.w_gs = w_gs; p.b_gs = b_gs; p.c_gs = c_gs;
    dim3 grid(cdiv(N, 32), cdiv(M, 32), split_k * groups);
    hipLaunchKernelGGL(gemm_x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
