// Dense building blocks of the head's TRAINING route: the backward of a linear layer y = act(x W^T + b)
// (MU/petr_transformer.py:195-311, RH/bbox_heads/cross_attention_head.py:118-142; mmcv FFN) as ONE call (ReLU mask of the gradient,
//     dx [M,K] = g W,   dW [N,K] = g^T x,   db [N] = column sums of g,
// both products on mv2d_gemm_f32x3 (csrc/gemm_f32x3.hip: fp32 operands read in place in either orientation, split into bf16 hi / lo while a
// tile is staged, three MFMAs per product), the layer-norm backward and a deterministic column sum (bias gradients, split-K slabs).
// (Round 3's first build of the products -- operand images [hi | lo | hi] in HBM + the bf16 tile GEMM, mv2d_split3_operand /
//  mv2d_matmul_nt_x3 -- was retired in round 4.)
#include "common.h"

namespace {

// out[c] = sum_r x[r, c]  (fp32, fixed order: deterministic): one block per 16 columns, 16 row groups of 16 lanes each (a wave reads
// 64-byte row pieces of 4 rows), 4 loads in flight per lane, partial sums through LDS.  (Rounds 3-4: 64 columns x 4 row groups -- 75
// dependent iterations for the 300-row gradients of a training step, 9.4 us for 300 KB; the bias gradients were 15 % of the step's kernel time.)
// (blockIdx.y = chunk of CS_ROWS rows, written to out + blockIdx.y * out_stride: long matrices are summed in two passes, both in fixed order)
constexpr int CS_ROWS = 512, CS_COLS = 16, CS_GROUPS = 16;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ld, int rows, int cols, float* out,
                                                     long long out_stride = 0, int chunk = 1 << 30, const float* add = nullptr) {
    __shared__ float part[CS_GROUPS][CS_COLS + 1];
    const int cl = threadIdx.x & (CS_COLS - 1), g = threadIdx.x / CS_COLS;
    const int c = blockIdx.x * CS_COLS + cl;
    const int r_begin = blockIdx.y * chunk, r_end = min(rows, r_begin + chunk);
    out += (long long)blockIdx.y * out_stride;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int r = r_begin + g;
        for (; r + 3 * CS_GROUPS < r_end; r += 4 * CS_GROUPS) {
            s0 += x[(long long)r * ld + c];
            s1 += x[(long long)(r + CS_GROUPS) * ld + c];
            s2 += x[(long long)(r + 2 * CS_GROUPS) * ld + c];
            s3 += x[(long long)(r + 3 * CS_GROUPS) * ld + c];
        }
        for (; r < r_end; r += CS_GROUPS) s0 += x[(long long)r * ld + c];
    }
    part[g][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CS_GROUPS; ++k) t += part[k][cl];
        out[c] = add ? add[c] + t : t;           // (add may alias out: each element is read and written by this thread only)
    }
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
    // (dy may be a contiguous slice with a storage offset, e.g. rows of 10 floats: 16-byte accesses only when all three pointers allow them)
    const bool vec = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    if (vec && i + 3 < n) {
        const float4 d = *reinterpret_cast<const float4*>(dy + i), v = *reinterpret_cast<const float4*>(y + i);
        *reinterpret_cast<float4*>(g + i) = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
    } else
        for (long long j = i; j < n && j < i + 4; ++j) g[j] = y[j] > 0.f ? dy[j] : 0.f;
}

// Softmax backward of a dense attention block, row by row: dS = P * (dPm - sum_j P_j dPm_j), dPm = dP * (Pd != 0 ? keep_scale : 0)
// (Pd = dropout(P): a kept probability is non-zero, and where P itself is 0 the product is 0 either way).  dP is overwritten with dS.
// One block per row, two passes over the row (the second one finds it in L2).
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, const float* __restrict__ Pd, float* __restrict__ dP,
                                                               long long ld, int cols, float keep_scale) {
    __shared__ float red[4];
    const long long o = (long long)blockIdx.x * ld;
    const bool drop = Pd != P;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float m = drop ? (Pd[o + c] != 0.f ? keep_scale : 0.f) : 1.f;
        s += P[o + c] * (dP[o + c] * m);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float m = drop ? (Pd[o + c] != 0.f ? keep_scale : 0.f) : 1.f;
        dP[o + c] = P[o + c] * (dP[o + c] * m - r);
    }
}

}  // namespace

// dP [rows, ld] (first `cols` columns of every row) <- P * (dP m - rowsum(P dP m)), m = keep_scale where Pd != 0 else 0 (Pd == P: no dropout, m = 1)
extern "C" int mv2d_softmax_bwd_rows(const float* P, const float* Pd, float* dP, long long ld, int rows, int cols, float keep_scale, void* stream) {
    MV2D_CHECK_ARG(P && Pd && dP && rows >= 0 && cols >= 0 && ld >= cols, "mv2d_softmax_bwd_rows: bad args");
    if (rows == 0 || cols == 0) return MV2D_OK;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, P, Pd, dP, ld, cols, keep_scale);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_colsum_scratch_rows(int rows) { return rows > 2 * CS_ROWS ? cdiv(rows, CS_ROWS) : 0; }

extern "C" int mv2d_colsum_add(const float* x, long long ld, int rows, int cols, float* out, float* scratch, const float* add, void* stream);
extern "C" int mv2d_colsum(const float* x, long long ld, int rows, int cols, float* out, float* scratch /* [mv2d_colsum_scratch_rows(rows), cols] or NULL */,
                           void* stream) {
    return mv2d_colsum_add(x, ld, rows, cols, out, scratch, nullptr, stream);
}

// out[c] = add[c] + sum_r x[r, c] (add NULL: the plain sum; add == out accumulates)
extern "C" int mv2d_colsum_add(const float* x, long long ld, int rows, int cols, float* out, float* scratch, const float* add, void* stream) {
    MV2D_CHECK_ARG(x && out && rows >= 0 && cols > 0, "mv2d_colsum: bad args");
    const int nch = mv2d_colsum_scratch_rows(rows);
    if (nch > 0 && scratch) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, CS_COLS), nch), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, scratch, (long long)cols, CS_ROWS, (const float*)nullptr);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, CS_COLS)), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, (long long)cols, nch, cols, out, 0LL, 1 << 30, add);
    } else
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, CS_COLS)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, out, 0LL, 1 << 30, add);
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
    hipLaunchKernelGGL(colsum_kernel, dim3(256 / CS_COLS), dim3(256), 0, (hipStream_t)stream, (const float*)dw_part, 256LL, nb, 256, dw, 0LL, 1 << 30);
    hipLaunchKernelGGL(colsum_kernel, dim3(256 / CS_COLS), dim3(256), 0, (hipStream_t)stream, (const float*)db_part, 256LL, nb, 256, db, 0LL, 1 << 30);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Composite entry: one call = the whole launch sequence of a linear layer's backward, on a caller-provided workspace (the training step was
// launch-bound from Python; the sequence below is issued from C at a few microseconds per launch).
// ---------------------------------------------------------------------------------------------------------------------------------------
static inline long long al256(long long b) { return (b + 255) & ~255LL; }

extern "C" long long mv2d_gemm_f32x3_ws_bytes(int M, int N, int K);
extern "C" int mv2d_gemm_f32x3(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                               float* C, long long ldc, int M, int N, int K, void* ws, long long ws_bytes, void* stream);

extern "C" int mv2d_wgrad_f32x3(const float* g, const float* x, float* dW, float* db, int M, int N, int K, void* ws, long long ws_bytes,
                                float* cs_scratch, void* stream);

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
    // dW = g^T x with db = column sums of g from the same kernel when the product runs in one pass (mv2d_wgrad_f32x3)
    if (dW) return mv2d_wgrad_f32x3(g, x, dW, db, M, N, K, mws, mm, mv2d_colsum_scratch_rows(M) ? (float*)w : nullptr, stream);
    if (db) return mv2d_colsum(g, N, M, N, db, mv2d_colsum_scratch_rows(M) ? (float*)w : nullptr, stream);
    return MV2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The query generator's 3 x 3 convolution over the 7 x 7 RoI features (RH/utils/query_generator.py:352-366, padding 1) runs as ONE product
// over the unfolded input: cols [R * 49, 9 * 256], column order (tap = 3 ky + kx, channel).  Rounds 3-4 unfolded with torch (pad + nine
// shifted views + cat: 11 launches forward, ~30 backward with nine zero-filled gradients); here one kernel per direction.
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {

// cols[(r, y, x), tap, c] = x[r, (y + ky - 1, x + kx - 1), c] or 0 outside the 7 x 7 cell grid
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ in, float* __restrict__ cols, int R) {
    const int pos = blockIdx.x * 4 + (threadIdx.x >> 6), c = 4 * (threadIdx.x & 63);
    if (pos >= R * 49) return;
    const int r = pos / 49, p = pos % 49, y = p / 7, x = p % 7;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yy >= 0 && yy < 7 && xx >= 0 && xx < 7) v = *reinterpret_cast<const float4*>(in + ((long long)r * 49 + yy * 7 + xx) * 256 + c);
        *reinterpret_cast<float4*>(cols + (long long)pos * 2304 + tap * 256 + c) = v;
    }
}
// dx[r, (y, x), c] = sum over the taps of dcols[(r, y - ky + 1, x - kx + 1), tap, c] (fixed order)
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int R) {
    const int pos = blockIdx.x * 4 + (threadIdx.x >> 6), c = 4 * (threadIdx.x & 63);
    if (pos >= R * 49) return;
    const int r = pos / 49, p = pos % 49, y = p / 7, x = p % 7;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y - tap / 3 + 1, xx = x - tap % 3 + 1;
        if (yy >= 0 && yy < 7 && xx >= 0 && xx < 7) {
            const float4 v = *reinterpret_cast<const float4*>(dcols + ((long long)r * 49 + yy * 7 + xx) * 2304 + tap * 256 + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(dx + (long long)pos * 256 + c) = s;
}

}  // namespace

// x [R, 49, 256] fp32 (7 x 7 cells, row-major) -> cols [R * 49, 2304]
extern "C" int mv2d_im2col3x3(const float* x, float* cols, int R, void* stream) {
    MV2D_CHECK_ARG(x && cols && R >= 0, "mv2d_im2col3x3: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(cdiv(R * 49, 4)), dim3(256), 0, (hipStream_t)stream, x, cols, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// gradient of mv2d_im2col3x3: dcols [R * 49, 2304] -> dx [R, 49, 256]
extern "C" int mv2d_col2im3x3(const float* dcols, float* dx, int R, void* stream) {
    MV2D_CHECK_ARG(dcols && dx && R >= 0, "mv2d_col2im3x3: bad args");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(cdiv(R * 49, 4)), dim3(256), 0, (hipStream_t)stream, dcols, dx, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// center2lidar + the normalisation of the reference points (RH/utils/query_generator.py:333-341, RH/mv2d_s_head.py:146-152), forward and
// backward: c = (u, v, depth) per RoI -> hom = (u d, v d, d, 1) -> xyz = (M^-1 hom)[:3] -> ref = (xyz - low) / range.  One launch each
// (the torch expression was ~9 element-wise launches forward and ~15 backward on [R, 4] tensors).
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {

struct C2LRange { float lo[3], span[3]; };

__global__ __launch_bounds__(256) void center2lidar_fwd_kernel(const float* __restrict__ c, const float* __restrict__ minv, float* __restrict__ ref, int R,
                                                               C2LRange rg) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float d = c[r * 3 + 2];
    const float hom[4] = {c[r * 3] * d, c[r * 3 + 1] * d, d, 1.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float* m = minv + r * 16 + 4 * i;
        const float xyz = ((m[0] * hom[0] + m[1] * hom[1]) + m[2] * hom[2]) + m[3] * hom[3];
        ref[r * 3 + i] = (xyz - rg.lo[i]) / rg.span[i];
    }
}

__global__ __launch_bounds__(256) void center2lidar_bwd_kernel(const float* __restrict__ g, const float* __restrict__ c, const float* __restrict__ minv,
                                                               float* __restrict__ dc, int R, C2LRange rg) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float dh[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float dx = g[r * 3 + i] / rg.span[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) dh[j] += minv[r * 16 + 4 * i + j] * dx;
    }
    const float u = c[r * 3], v = c[r * 3 + 1], d = c[r * 3 + 2];
    dc[r * 3] = dh[0] * d;
    dc[r * 3 + 1] = dh[1] * d;
    dc[r * 3 + 2] = (dh[0] * u + dh[1] * v) + dh[2];
}

}  // namespace

// c [R,3] (u, v, depth), minv [R,16] = inverse(K_roi E^T) row-major, pc_range: 6 HOST floats -> ref [R,3] normalised reference points
extern "C" int mv2d_center2lidar_fwd(const float* c, const float* minv, float* ref, int R, const float* pc_range, void* stream) {
    MV2D_CHECK_ARG(c && minv && ref && pc_range && R >= 0, "mv2d_center2lidar_fwd: bad args");
    if (R == 0) return MV2D_OK;
    C2LRange rg;
    for (int k = 0; k < 3; ++k) { rg.lo[k] = pc_range[k]; rg.span[k] = pc_range[3 + k] - pc_range[k]; }
    hipLaunchKernelGGL(center2lidar_fwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, c, minv, ref, R, rg);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// g [R,3] = gradient of ref -> dc [R,3]
extern "C" int mv2d_center2lidar_bwd(const float* g, const float* c, const float* minv, float* dc, int R, const float* pc_range, void* stream) {
    MV2D_CHECK_ARG(g && c && minv && dc && pc_range && R >= 0, "mv2d_center2lidar_bwd: bad args");
    if (R == 0) return MV2D_OK;
    C2LRange rg;
    for (int k = 0; k < 3; ++k) { rg.lo[k] = pc_range[k]; rg.span[k] = pc_range[3 + k] - pc_range[k]; }
    hipLaunchKernelGGL(center2lidar_bwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, g, c, minv, dc, R, rg);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
