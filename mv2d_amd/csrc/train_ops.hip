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
