// Dense building blocks of the head's TRAINING route: forward and backward of the decoder's linears / layer norms /
// FFN / prediction branches run on the hand-written bf16 tile GEMM (gemm_bf16.hip) in split precision by K-concatenation,
//     [a_hi | a_lo | a_hi] . [b_hi | b_hi | b_lo]^T = a_hi b_hi + a_lo b_hi + a_hi b_lo          (fp32 accumulation, ~1e-5 relative),
// instead of torch autograd over rocBLAS.  The three products of a linear layer y = x W^T (MU/petr_transformer.py:195-311,
// RH/bbox_heads/cross_attention_head.py:118-142; mmcv FFN) all have the form C = A B^T once the operands are laid out right:
//     forward   y  [M,N] = x  [M,K] . W   [N,K]^T
//     backward  dx [M,K] = dy [M,N] . W^T [K,N]^T          (the B operand is the TRANSPOSE of W)
//               dW [N,K] = dy^T [N,M] . x^T [K,M]^T        (both operands transposed; the contraction runs over the M rows)
// so the only new kernels are the operand builders below (fp32 matrix, optionally transposed, zero-padded -> bf16 [rows, 3 K'] in the
// [hi | lo | hi] (A side) or [hi | hi | lo] (B side) form), the layer-norm backward and a deterministic column sum (bias gradients).
#include "common.h"

namespace {

__device__ __forceinline__ void tr_split(float v, unsigned short& hi, unsigned short& lo) {
    hi = f32_to_bf16(v);
    lo = f32_to_bf16(v - bf16_to_f32(hi));
}

// dst [rows_out, 3 * kp] bf16 from src fp32: element (r, k) = transpose ? src[k * ld + r] : src[r * ld + k] for r < rows, k < kk, else 0.
// side 0 (A operand): [hi | lo | hi]; side 1 (B operand): [hi | hi | lo].  32 x 32 tiles through LDS so that both the reads and the writes
// are row-contiguous in either orientation.
__global__ __launch_bounds__(256) void split3_op_kernel(const float* __restrict__ src, long long ld, int rows, int kk, int transpose,
                                                        unsigned short* __restrict__ dst, int rows_out, int kp, int side) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
    if (transpose) {
        // src is [kk, rows]: read rows of src (index k) contiguous in r
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + ty + 8 * j, r = r0 + tx;
            tile[ty + 8 * j][tx] = (k < kk && r < rows) ? src[(long long)k * ld + r] : 0.f;       // tile[k_local][r_local]
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 8 * j, k = k0 + tx;
            tile[tx][ty + 8 * j] = (r < rows && k < kk) ? src[(long long)r * ld + k] : 0.f;       // tile[k_local][r_local]
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, k = k0 + tx;
        if (r < rows_out && k < kp) {
            unsigned short hi, lo;
            tr_split(tile[tx][ty + 8 * j], hi, lo);
            unsigned short* o = dst + (long long)r * (3LL * kp) + k;
            o[0] = hi;
            o[kp] = side ? hi : lo;
            o[2 * kp] = side ? lo : hi;
        }
    }
}

// out[c] = sum_r x[r, c]  (fp32, fixed order: deterministic): one block per 64 columns, 4 row groups of 64 lanes each, partial sums
// through LDS.
// (blockIdx.y = chunk of CS_ROWS rows, written to out + blockIdx.y * out_stride: long matrices are summed in two passes, both in fixed order)
constexpr int CS_ROWS = 512;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ld, int rows, int cols, float* __restrict__ out,
                                                     long long out_stride = 0, int chunk = 1 << 30) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int r_begin = blockIdx.y * chunk, r_end = min(rows, r_begin + chunk);
    out += (long long)blockIdx.y * out_stride;
    float s = 0.f;
    if (c < cols)
        for (int r = r_begin + g; r < r_end; r += 4) s += x[(long long)r * ld + c];
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < cols) out[c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// LayerNorm backward over rows of 256 (nn.LayerNorm, eps inside the square root): y = xhat * w + b, xhat = (x - mean) * rstd.
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w;   dw_part[block] = sum over the block's rows of dy * xhat,  db_part = sum dy
// One wave per row (4 channels per lane), 4 rows per block pass, ROWS_PER_BLOCK rows per block; the per-block partial column sums are
// reduced by colsum_kernel (fixed order).
constexpr int LNB_ROWS = 64;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                     float* __restrict__ dx, float* __restrict__ dw_part, float* __restrict__ db_part, int M, float eps) {
    __shared__ float sw[4][256], sb[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 wv = *reinterpret_cast<const float4*>(w + 4 * lane);
    float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
    const int r_begin = blockIdx.x * LNB_ROWS;
    for (int r = r_begin + wave; r < min(r_begin + LNB_ROWS, M); r += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (long long)r * 256 + 4 * lane);
        const float4 dv = *reinterpret_cast<const float4*>(dy + (long long)r * 256 + 4 * lane);
        const float mean = wave_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.f / 256.f);
        const float4 c = make_float4(xv.x - mean, xv.y - mean, xv.z - mean, xv.w - mean);
        const float var = wave_sum((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.f / 256.f);
        const float rstd = 1.f / sqrtf(var + eps);
        const float4 xh = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
        const float4 g = make_float4(dv.x * wv.x, dv.y * wv.y, dv.z * wv.z, dv.w * wv.w);
        const float mg = wave_sum((g.x + g.y) + (g.z + g.w)) * (1.f / 256.f);
        const float mgx = wave_sum((g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w)) * (1.f / 256.f);
        *reinterpret_cast<float4*>(dx + (long long)r * 256 + 4 * lane) =
            make_float4(rstd * (g.x - mg - xh.x * mgx), rstd * (g.y - mg - xh.y * mgx), rstd * (g.z - mg - xh.z * mgx), rstd * (g.w - mg - xh.w * mgx));
        aw.x += dv.x * xh.x; aw.y += dv.y * xh.y; aw.z += dv.z * xh.z; aw.w += dv.w * xh.w;
        ab.x += dv.x; ab.y += dv.y; ab.z += dv.z; ab.w += dv.w;
    }
    *reinterpret_cast<float4*>(&sw[wave][4 * lane]) = aw;
    *reinterpret_cast<float4*>(&sb[wave][4 * lane]) = ab;
    __syncthreads();
    const int c = threadIdx.x;
    dw_part[(long long)blockIdx.x * 256 + c] = (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]);
    db_part[(long long)blockIdx.x * 256 + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
}

// g = dy where y > 0 else 0 (the ReLU of a linear layer's forward, applied to the incoming gradient)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ g, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 d = *reinterpret_cast<const float4*>(dy + i), v = *reinterpret_cast<const float4*>(y + i);
        *reinterpret_cast<float4*>(g + i) = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
    } else
        for (long long j = i; j < n; ++j) g[j] = y[j] > 0.f ? dy[j] : 0.f;
}

}  // namespace

extern "C" int mv2d_split3_operand(const float* src, long long ld, int rows, int k, int transpose, void* dst, int rows_out, int k_pad, int side,
                                   void* stream) {
    MV2D_CHECK_ARG(src && dst && rows >= 0 && k > 0 && rows_out >= rows && k_pad >= k && (k_pad % 8) == 0 && (side == 0 || side == 1),
                   "mv2d_split3_operand: bad args (k_pad >= k, a multiple of 8; side 0 = [hi|lo|hi], 1 = [hi|hi|lo])");
    if (rows_out == 0) return MV2D_OK;
    hipLaunchKernelGGL(split3_op_kernel, dim3(cdiv(k_pad, 32), cdiv(rows_out, 32)), dim3(256), 0, (hipStream_t)stream, src, ld, rows, k, transpose,
                       (unsigned short*)dst, rows_out, k_pad, side);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_colsum_scratch_rows(int rows) { return rows > 2 * CS_ROWS ? cdiv(rows, CS_ROWS) : 0; }

extern "C" int mv2d_colsum(const float* x, long long ld, int rows, int cols, float* out, float* scratch /* [mv2d_colsum_scratch_rows(rows), cols] or NULL */,
                           void* stream) {
    MV2D_CHECK_ARG(x && out && rows >= 0 && cols > 0, "mv2d_colsum: bad args");
    const int nch = mv2d_colsum_scratch_rows(rows);
    if (nch > 0 && scratch) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64), nch), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, scratch, (long long)cols, CS_ROWS);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, (long long)cols, nch, cols, out, 0LL, 1 << 30);
    } else
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, out, 0LL, 1 << 30);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_layer_norm_bwd_blocks(int M) { return cdiv(M, LNB_ROWS); }

extern "C" int mv2d_layer_norm_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw_part, float* db_part, float* dw, float* db,
                                   int M, float eps, void* stream) {
    MV2D_CHECK_ARG(x && dy && w && dx && dw_part && db_part && dw && db && M >= 0, "mv2d_layer_norm_bwd: bad args");
    MV2D_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)w & 15) == 0,
                   "mv2d_layer_norm_bwd: operands must be 16-byte aligned (rows of 256 fp32)");
    const int nb = cdiv(M, LNB_ROWS);
    if (nb > 0) hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, dy, w, dx, dw_part, db_part, M, eps);
    hipLaunchKernelGGL(colsum_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, (const float*)dw_part, 256LL, nb, 256, dw, 0LL, 1 << 30);
    hipLaunchKernelGGL(colsum_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, (const float*)db_part, 256LL, nb, 256, db, 0LL, 1 << 30);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Composite entries (round 3): one call = the whole launch sequence of a split-precision product / of a linear layer's backward, on a
// caller-provided workspace.  The training step was launch-bound from Python (~1500 launches of ~20 us host time each); the sequences
// below are issued from C at a few microseconds per launch.
// ---------------------------------------------------------------------------------------------------------------------------------------
extern "C" int mv2d_gemm_bf16_ex(const void* A, const void* A2, int n_split, int a_mode, const void* W, const float* bias, int M, int N, int K, int lda,
                                 const int* m_dev, int act, const float* mul, int ldmul, const float* add, int ldadd, void* C, int c_bf16, int ldc,
                                 long long c_blk_stride, int c_blk_cols, void* C2, const float* add2, int ldc2, int ldadd2, int c_split3,
                                 const int* add_idx, int add_period, int k_splits, long long c_split_stride, void* stream);

static inline long long al256(long long b) { return (b + 255) & ~255LL; }
static inline int pad_to(int n, int m) { return (n + m - 1) / m * m; }
static int mm_splits(int M, int Np, int kp, int act) {
    // few output tiles with a long contraction (weight gradients: K = the rows of the layer input): split K over the grid's y dimension
    const long long tiles = (long long)cdiv(M, 64) * cdiv(Np, 64);
    if (act != 0 || tiles >= 256 || kp < 1024) return 1;
    long long s = 512 / tiles;
    if (s < 1) s = 1;
    if (s > 3LL * kp / 256) s = 3LL * kp / 256;
    if (s > 64) s = 64;
    return (int)s;
}

// bytes of workspace mv2d_matmul_nt_x3 needs for C[M,N] = op(A) op(B)^T with a contraction of K
extern "C" long long mv2d_matmul_nt_x3_ws_bytes(int M, int N, int K) {
    const int kp = pad_to(K, 64), Np = pad_to(N, 8);
    const int splits = mm_splits(M, Np, kp, 0);
    long long b = al256((long long)M * 3 * kp * 2) + al256((long long)Np * 3 * kp * 2) + al256((long long)Np * 4);
    if (splits > 1) b += al256((long long)splits * M * Np * 4) + al256((long long)mv2d_colsum_scratch_rows(splits) * M * Np * 4);
    return b;
}

// C [M, ldc >= pad8(N)] fp32 = act(op(A) op(B)^T + bias): A fp32 [M,K] (or [K,M] with trans_a), B fp32 [N,K] (or [K,N] with trans_b), unit
// column stride, row strides lda / ldb; bias [N] or NULL; act 0 none / 1 ReLU.  Columns N .. pad8(N) of C are written too (zeros + nothing).
extern "C" int mv2d_matmul_nt_x3(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                                 float* C, int ldc, int M, int N, int K, void* ws, long long ws_bytes, void* stream) {
    MV2D_CHECK_ARG(A && B && C && M >= 0 && N > 0 && K > 0 && (act == 0 || act == 1), "mv2d_matmul_nt_x3: bad args");
    if (M == 0) return MV2D_OK;
    const int kp = pad_to(K, 64), Np = pad_to(N, 8);
    MV2D_CHECK_ARG(ldc >= Np && (ldc % 4) == 0, "mv2d_matmul_nt_x3: ldc >= N rounded up to 8, a multiple of 4");
    MV2D_CHECK_ARG(ws && ws_bytes >= mv2d_matmul_nt_x3_ws_bytes(M, N, K) && ((uintptr_t)ws & 255) == 0, "mv2d_matmul_nt_x3: workspace too small / misaligned");
    const int splits = mm_splits(M, Np, kp, act);
    char* w = (char*)ws;
    void* a3 = w; w += al256((long long)M * 3 * kp * 2);
    void* b3 = w; w += al256((long long)Np * 3 * kp * 2);
    float* bias_p = (float*)w; w += al256((long long)Np * 4);
    float* slabs = (float*)w; if (splits > 1) w += al256((long long)splits * M * Np * 4);
    float* scratch = (float*)w;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = mv2d_split3_operand(A, lda, M, K, trans_a, a3, M, kp, 0, stream)) != MV2D_OK) return rc;
    if ((rc = mv2d_split3_operand(B, ldb, N, K, trans_b, b3, Np, kp, 1, stream)) != MV2D_OK) return rc;
    const float* bp = bias;
    if (bias && Np != N) {
        if (hipMemsetAsync(bias_p, 0, (size_t)Np * 4, st) != hipSuccess || hipMemcpyAsync(bias_p, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return MV2D_ERR_LAUNCH;
        bp = bias_p;
    }
    if (splits > 1) {
        MV2D_CHECK_ARG(ldc == Np, "mv2d_matmul_nt_x3: the split-K route writes a dense C (ldc = N rounded up to 8)");
        if ((rc = mv2d_gemm_bf16_ex(a3, nullptr, 0, 0, b3, bp, M, Np, 3 * kp, 3 * kp, nullptr, 0, nullptr, 0, nullptr, 0, slabs, 0, Np, 0, 0, nullptr, nullptr,
                                    0, 0, 0, nullptr, 0, splits, (long long)M * Np, stream)) != MV2D_OK) return rc;
        return mv2d_colsum(slabs, (long long)M * Np, splits, M * Np, C, mv2d_colsum_scratch_rows(splits) ? scratch : nullptr, stream);
    }
    return mv2d_gemm_bf16_ex(a3, nullptr, 0, 0, b3, bp, M, Np, 3 * kp, 3 * kp, nullptr, act, nullptr, 0, nullptr, 0, C, 0, ldc, 0, 0, nullptr, nullptr, 0, 0, 0,
                             nullptr, 0, 1, 0, stream);
}

extern "C" long long mv2d_gemm_f32x3_ws_bytes(int M, int N, int K);
extern "C" int mv2d_gemm_f32x3(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                               float* C, long long ldc, int M, int N, int K, void* ws, long long ws_bytes, void* stream);

extern "C" long long mv2d_linear_bwd_x3_ws_bytes(int M, int N, int K) {
    const long long a = al256(mv2d_gemm_f32x3_ws_bytes(M, K, N)), b = al256(mv2d_gemm_f32x3_ws_bytes(N, K, M));
    return al256((long long)M * N * 4) + (a > b ? a : b) + 256 + al256((long long)mv2d_colsum_scratch_rows(M) * N * 4);
}

// Backward of y = act(x W^T + b), x [M,K], W [N,K], dy / y [M,N] (dense rows): g = dy (masked by y > 0 when y is given);
// dx [M,K] = g W (skipped when NULL), dW [N,K] = g^T x (skipped when NULL), db [N] = column sums of g (skipped when NULL).  The two
// products run on mv2d_gemm_f32x3 (fp32 operands read in place, transposed where the product needs it; split-K for dW over many rows).
extern "C" int mv2d_linear_bwd_x3(const float* x, const float* W, const float* y, const float* dy, float* dx, float* dW, float* db, int M, int N, int K,
                                  void* ws, long long ws_bytes, void* stream) {
    MV2D_CHECK_ARG(x && W && dy && M >= 0 && N > 0 && K > 0, "mv2d_linear_bwd_x3: bad args");
    MV2D_CHECK_ARG(ws && ws_bytes >= mv2d_linear_bwd_x3_ws_bytes(M, N, K) && ((uintptr_t)ws & 255) == 0, "mv2d_linear_bwd_x3: workspace too small / misaligned");
    if (M == 0) return MV2D_OK;                      // (the caller zero-fills dW / db for an empty batch)
    char* w = (char*)ws;
    const float* g = dy;
    if (y) {
        float* gm = (float*)w;
        const long long n = (long long)M * N;
        hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)cdiv((int)((n + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, dy, y, gm, n);
        MV2D_LAUNCH_CHECK();
        g = gm;
    }
    w += al256((long long)M * N * 4);
    const long long a = al256(mv2d_gemm_f32x3_ws_bytes(M, K, N)), b = al256(mv2d_gemm_f32x3_ws_bytes(N, K, M)), mm = (a > b ? a : b) + 256;
    void* mws = w; w += mm;
    int rc;
    if (dx && (rc = mv2d_gemm_f32x3(g, N, 0, W, K, 1, nullptr, 0, dx, K, M, K, N, mws, mm, stream)) != MV2D_OK) return rc;          // g [M,N] . (W^T [K,N])^T
    if (dW && (rc = mv2d_gemm_f32x3(g, N, 1, x, K, 1, nullptr, 0, dW, K, N, K, M, mws, mm, stream)) != MV2D_OK) return rc;          // g^T [N,M] . (x^T [K,M])^T
    if (db) return mv2d_colsum(g, N, M, N, db, mv2d_colsum_scratch_rows(M) ? (float*)w : nullptr, stream);
    return MV2D_OK;
}
