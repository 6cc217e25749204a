// Split-precision GEMM on fp32 operands in either orientation (gfx950 / CDNA4, wave64) -- the dense products of the head's TRAINING
// route (nn.Linear / mmcv FFN forward and backward, MU/petr_transformer.py:195-311,
// RH/bbox_heads/cross_attention_head.py:118-142):
//
//     C[m, n] = act( sum_k opA[m, k] * opB[n, k] + bias[n] ),     opA[m, k] = TA ? A[k * lda + m] : A[m * lda + k]   (same for B)
//
//     forward   y  = x W^T      : A = x  [M,K],             B = W [N,K]
//     backward  dx = g W        : A = g  [M,N],             B = W [N,K] read TRANSPOSED (contraction over N)
//               dW = g^T x      : A = g  [M,N] TRANSPOSED,  B = x [M,K] TRANSPOSED (contraction over the M rows of the layer input)
//
// Round 3 first built these from three launches each (mv2d_split3_operand x 2 -> bf16 [hi | lo | hi] / [hi | hi | lo] images in HBM, transposed
// where needed, then the bf16 tile GEMM over the 3 K concatenation): 757 operand builds per training step = 19 % of its kernel time, and the
// weight-gradient products over the 14.7 k key rows read their transposed operands as 128-byte pieces 88 KB apart (287 us for 5.8 GFLOP).
// Here the operands stay fp32 in HBM: a tile is loaded in its NATURAL orientation (row-contiguous float4 loads either way), split into
// bf16 hi / lo in registers (x = hi + lo, hi = bf16(x), lo = bf16(x - hi)) and written to LDS in the k-contiguous layout the MFMA fragments
// want -- a transposed operand through a 4 x 4 register transpose, 8-byte LDS stores.  Per k-step three v_mfma_f32_16x16x32_bf16
// (a_hi b_hi + a_lo b_hi + a_hi b_lo, fp32 accumulation: ~1e-5 relative, the dropped a_lo b_lo term is 2^-18).
//
// Block = 64 x 64 outputs, 64 of the contraction per stage, 4 waves as 2 x 2 (32 x 32 per wave).  One LDS stage, the next stage's global
// loads in flight during the MFMAs (these products are small: 3..36 stages).  Split-K over blockIdx.y into fp32 slabs for the products with
// few output tiles and a long contraction (the caller sums the slabs in fixed order, mv2d_colsum).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 fx_bf16x8;
union FxFrag { uint4 u; uint2 h[2]; fx_bf16x8 v; };

constexpr int BM = 64, BN = 64, BK = 64;
constexpr int IMG = 64 * BK * 2;                      // one bf16 image [64 rows][64 k]: 8 KB

struct FxParams {
    const float* A; long long lda;
    const float* B; long long ldb;
    const float* bias;
    float* C; long long ldc;
    int M, N, K, act;
    int k_tiles_per_split; long long c_split_stride;
    float alpha;                // C = act((sum + bias) * alpha)
    int accumulate;             // C += ... (one-pass products only; the split-K slabs are summed onto C by the caller's column sum)
    int out_bf16;               // C holds bf16 (the keys / values the sparse attention kernels read)
    const float* relu_y;        // fp32 [M, ldc] or NULL: C = relu_y > 0 ? value : 0 (the ReLU of the forward applied to a gradient)
    long long batch_a, batch_b, batch_c;   // element strides of blockIdx.z (a batch of products with the same shape: the heads of an attention)
    float* rowsum;              // trans_a products in one pass: rowsum[m] = sum_k opA[m, k] (the bias gradient next to dW = g^T x), or NULL
};

// byte offset of k-group `kq` (4 consecutive k = 8 bytes) of row `row` in an image; 16-byte slots XOR-swizzled so that both the
// fragment reads (16 consecutive rows, one slot) and the transposed stores (rows 4 apart, one slot) spread over the banks
__device__ __forceinline__ int fx_off(int row, int kq) {
    const int slot = (kq >> 1) ^ ((row ^ (row >> 3)) & 7);
    return row * (BK * 2) + (slot << 4) + ((kq & 1) << 3);
}

__device__ __forceinline__ void fx_split4(const float4& v, uint2& hi, uint2& lo) {
    hi.x = pack_bf16x2(v.x, v.y); hi.y = pack_bf16x2(v.z, v.w);
    lo.x = pack_bf16x2(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack_bf16x2(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
}

__device__ __forceinline__ float4 fx_add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// 4 consecutive floats from p (element index e of a row of `limit` valid elements), zero beyond; vec: the row base and stride keep 16-byte alignment
__device__ __forceinline__ float4 fx_load4(const float* __restrict__ row, int e, int limit, bool vec) {
    if (e + 3 < limit && vec) return *reinterpret_cast<const float4*>(row + e);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < limit) v.x = row[e];
    if (e + 1 < limit) v.y = row[e + 1];
    if (e + 2 < limit) v.z = row[e + 2];
    if (e + 3 < limit) v.w = row[e + 3];
    return v;
}

// One operand tile: `rows` = its 64 output rows (m or n) starting at r0 (< R valid), contraction range [k0, k0 + 64) (< K valid).
// T = false: memory [R, K] row stride ld.  T = true: memory [K, R] row stride ld.
template <bool T>
struct FxTile {
    float4 v[4];
    __device__ __forceinline__ void load(const float* __restrict__ P, long long ld, int r0, int R, int k0, int K, bool vec, int tid) {
        if (!T) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i, row = idx >> 4, c4 = idx & 15;
                const int r = r0 + row;
                v[i] = r < R ? fx_load4(P + (long long)r * ld, k0 + 4 * c4, K, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int kg = tid >> 4, rg = tid & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 4 * kg + j;
                v[j] = k < K ? fx_load4(P + (long long)k * ld, r0 + 4 * rg, R, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* hi_img, unsigned char* lo_img, int tid) const {
        if (!T) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i, row = idx >> 4, c4 = idx & 15;
                uint2 h, l;
                fx_split4(v[i], h, l);
                const int o = fx_off(row, c4);
                *reinterpret_cast<uint2*>(hi_img + o) = h;
                *reinterpret_cast<uint2*>(lo_img + o) = l;
            }
        } else {
            const int kg = tid >> 4, rg = tid & 15;
            const float a[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                                   {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};      // a[k][row]
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint2 h, l;
                fx_split4(make_float4(a[0][c], a[1][c], a[2][c], a[3][c]), h, l);
                const int o = fx_off(4 * rg + c, kg);
                *reinterpret_cast<uint2*>(hi_img + o) = h;
                *reinterpret_cast<uint2*>(lo_img + o) = l;
            }
        }
    }
};

// PF = stages of global loads in flight (a register ring).  PF = 1: one stage ahead, few registers -- the products with hundreds of
// blocks, where other blocks hide the latency.  PF = 4: the products of a training step with 4..160 blocks (300 query rows): their
// duration is the chain stage -> stage of ONE block, so all (up to four) stages are requested before the first one is used
// (10.8 -> ~6 us for 300 x 256 x 256).  Same k order either way: the results do not depend on PF.
template <bool TA, bool TB, int PF>
__global__ __launch_bounds__(256) void gemm_f32x3_kernel(FxParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * IMG];
    unsigned char *Ah = smem, *Al = smem + IMG, *Bh = smem + 2 * IMG, *Bl = smem + 3 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    p.A += blockIdx.z * p.batch_a; p.B += blockIdx.z * p.batch_b; p.C += blockIdx.z * p.batch_c;
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / n_tiles) * BM, n0 = (blockIdx.x % n_tiles) * BN;
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = blockIdx.y * p.k_tiles_per_split, kt1 = min(nk_all, kt0 + p.k_tiles_per_split);
    const bool vecA = ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0) && (p.lda % 4) == 0;
    const bool vecB = ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0) && (p.ldb % 4) == 0;

    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    FxTile<TA> ta[PF];
    FxTile<TB> tb[PF];
    const bool want_rs = TA && p.rowsum != nullptr && n0 == 0;
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (kt0 + u < kt1) {
            ta[u].load(p.A, p.lda, m0, p.M, (kt0 + u) * BK, p.K, vecA, tid);
            tb[u].load(p.B, p.ldb, n0, p.N, (kt0 + u) * BK, p.K, vecB, tid);
        }
    for (int ktb = kt0; ktb < kt1; ktb += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = ktb + u;
            if (kt >= kt1) break;                           // (uniform)
            ta[u].store(Ah, Al, tid);
            tb[u].store(Bh, Bl, tid);
            if (TA && want_rs) rs = fx_add4(rs, fx_add4(fx_add4(ta[u].v[0], ta[u].v[1]), fx_add4(ta[u].v[2], ta[u].v[3])));
            if (kt + PF < kt1) {                            // this register set is free again: the stage PF ahead
                ta[u].load(p.A, p.lda, m0, p.M, (kt + PF) * BK, p.K, vecA, tid);
                tb[u].load(p.B, p.ldb, n0, p.N, (kt + PF) * BK, p.K, vecB, tid);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                FxFrag ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wr * 32 + i * 16 + fr, o = fx_off(row, 2 * (4 * ks + fg));
                    ah[i].u = *reinterpret_cast<const uint4*>(Ah + o);
                    al[i].u = *reinterpret_cast<const uint4*>(Al + o);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wc * 32 + j * 16 + fr, o = fx_off(row, 2 * (4 * ks + fg));
                    bh[j].u = *reinterpret_cast<const uint4*>(Bh + o);
                    bl[j].u = *reinterpret_cast<const uint4*>(Bl + o);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i].v, bh[j].v, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i].v, bh[j].v, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i].v, bl[j].v, acc[i][j], 0, 0, 0);
                    }
            }
            __syncthreads();                                // every wave is done reading this stage
        }
    }
    if (TA && want_rs) {
        // thread (kg, rg) holds the sums over its k of rows 4 rg .. 4 rg + 3: reduce over the 16 kg in fixed order through LDS
        float* red = reinterpret_cast<float*>(smem);                // (every wave is past the last MFMA stage: the images are free)
        const int kg = tid >> 4, rg = tid & 15;
        *reinterpret_cast<float4*>(red + kg * 64 + 4 * rg) = rs;
        __syncthreads();
        if (tid < 64 && m0 + tid < p.M) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g * 64 + tid];
            p.rowsum[m0 + tid] = t;
        }
    }
    // epilogue: the MFMA leaves C[row 4 fg + r][col fr] of every 16 x 16 tile in lane (fr, fg): 16 consecutive columns per store instruction
    float* Cb = p.C + (long long)blockIdx.y * p.c_split_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 32 + j * 16 + fr;
        if (n >= p.N) continue;
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + 4 * fg + r;
                if (m < p.M) {
                    float v = (acc[i][j][r] + bv) * p.alpha;
                    if (p.act == 1) v = relu_f(v);
                    if (p.relu_y && !(p.relu_y[(long long)m * p.ldc + n] > 0.f)) v = 0.f;
                    if (p.out_bf16) { reinterpret_cast<unsigned short*>(Cb)[(long long)m * p.ldc + n] = f32_to_bf16(v); continue; }
                    if (p.accumulate) v += Cb[(long long)m * p.ldc + n];
                    Cb[(long long)m * p.ldc + n] = v;
                }
            }
    }
}

}  // namespace

extern "C" int mv2d_colsum_scratch_rows(int rows);
extern "C" int mv2d_colsum(const float* x, long long ld, int rows, int cols, float* out, float* scratch, void* stream);
extern "C" int mv2d_colsum_add(const float* x, long long ld, int rows, int cols, float* out, float* scratch, const float* add, void* stream);

static int fx_splits(int M, int N, int K, int act) {
    // few output tiles with a long contraction (weight gradients: K = the rows of the layer input): split K over blockIdx.y
    const long long tiles = (long long)cdiv(M, BM) * cdiv(N, BN);
    const int nk = cdiv(K, BK);
    if (act != 0 || tiles >= 256 || nk < 8) return 1;
    long long s = 512 / tiles;
    if (s > nk / 2) s = nk / 2;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

// bytes of workspace mv2d_gemm_f32x3 needs (split-K slabs + the scratch of their column sum); 0 when the product runs in one pass
extern "C" long long mv2d_gemm_f32x3_ws_bytes(int M, int N, int K) {
    const int s = fx_splits(M, N, K, 0);
    if (s <= 1) return 0;
    return (long long)s * M * N * 4 + (long long)mv2d_colsum_scratch_rows(s) * M * N * 4 + 512;
}

// C [M, ldc] fp32 = act(op(A) op(B)^T + bias): see the file header.  A [M,K] (trans_a: [K,M]), B [N,K] (trans_b: [K,N]), unit column strides,
// row strides lda / ldb (any; 16-byte aligned rows take the float4 path); bias [N] or NULL; act 0 none / 1 ReLU.  ws: workspace of
// mv2d_gemm_f32x3_ws_bytes(M, N, K) bytes, 256-byte aligned; without it (or with ldc != N) a long contraction runs in one pass instead of split-K.
extern "C" int mv2d_gemm_f32x3_ex(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                                  float alpha, int accumulate, int out_bf16, void* C, long long ldc, int M, int N, int K, void* ws,
                                  long long ws_bytes, void* stream);
static int gemm_f32x3_impl(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                           float alpha, int accumulate, int out_bf16, void* Cv, long long ldc, int M, int N, int K, void* ws, long long ws_bytes,
                           float* rowsum, const float* relu_y, void* stream);

extern "C" int mv2d_gemm_f32x3(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                               float* C, long long ldc, int M, int N, int K, void* ws, long long ws_bytes, void* stream) {
    return mv2d_gemm_f32x3_ex(A, lda, trans_a, B, ldb, trans_b, bias, act, 1.f, 0, 0, C, ldc, M, N, K, ws, ws_bytes, stream);
}

// The same product with C = act((op(A) op(B)^T + bias) * alpha); accumulate: C += (fp32 C); out_bf16: C is bf16 [M, ldc] (one pass, no accumulate).
extern "C" int mv2d_gemm_f32x3_ex(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                                  float alpha, int accumulate, int out_bf16, void* Cv, long long ldc, int M, int N, int K, void* ws,
                                  long long ws_bytes, void* stream) {
    return gemm_f32x3_impl(A, lda, trans_a, B, ldb, trans_b, bias, act, alpha, accumulate, out_bf16, Cv, ldc, M, N, K, ws, ws_bytes, nullptr, nullptr, stream);
}

// dx [M,K] = (g [M,N] W [N,K]) * alpha, zeroed where relu_y [M,K] <= 0 (relu_y NULL: no mask): the input gradient of a linear layer whose input
// came out of a ReLU (and a dropout: alpha = 1 / keep rate), mask applied in the product's epilogue.  One pass.
extern "C" int mv2d_dgrad_relu_f32x3(const float* g, const float* W, const float* relu_y, float alpha, float* dx, int M, int N, int K, void* stream) {
    MV2D_CHECK_ARG(g && W && dx && M >= 0 && N > 0 && K > 0, "mv2d_dgrad_relu_f32x3: bad args");
    if (M == 0) return MV2D_OK;
    return gemm_f32x3_impl(g, N, 0, W, K, 1, nullptr, 0, alpha, 0, 0, dx, K, M, K, N, nullptr, 0, nullptr, relu_y, stream);
}

// sum of the split-K slabs of a batched product: out[m * ldc + b * batch_c + n] = sum_s slab[s][b][m][n] (fixed order)
__global__ __launch_bounds__(256) void fx_batched_slab_sum_kernel(const float* __restrict__ slab, int splits, int batch, int M, int N, float* __restrict__ out,
                                                                  long long ldc, long long batch_c) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, per = (long long)batch * M * N;
    if (i >= per) return;
    float t = 0.f;
    for (int s_ = 0; s_ < splits; ++s_) t += slab[s_ * per + i];
    const int n = (int)(i % N), m = (int)((i / N) % M), b = (int)(i / ((long long)M * N));
    out[(long long)m * ldc + b * batch_c + n] = t;
}

static int fx_batched_splits(int M, int N, int K, int batch) {
    const long long tiles = (long long)cdiv(M, BM) * cdiv(N, BN) * batch;
    const int nk = cdiv(K, BK);
    if (tiles >= 256 || nk < 8) return 1;
    long long s_ = 768 / tiles;
    if (s_ > nk / 2) s_ = nk / 2;
    if (s_ > 64) s_ = 64;
    return s_ < 1 ? 1 : (int)s_;
}

// bytes of workspace the batched product wants for split-K (0: it runs in one pass)
extern "C" long long mv2d_gemm_f32x3_batched_ws_bytes(int M, int N, int K, int batch) {
    const int s_ = fx_batched_splits(M, N, K, batch);
    return s_ <= 1 ? 0 : (long long)s_ * batch * M * N * 4;
}

// `batch` products of one shape in one launch (blockIdx.z): C_b [M, ldc] = op(A_b) op(B_b)^T with A_b = A + b * batch_a (elements), likewise B_b, C_b
// -- the per-head products of a dense attention block (head b = a 32-column slice of [rows, 256] operands: batch stride 32, row stride 256).
// No bias / activation; C = alpha * product.  Few output tiles with a long contraction (P V and dS K of the denoising rows: 400 x 32 outputs per head over 16 k
// keys) are split over K into slabs in `ws` (mv2d_gemm_f32x3_batched_ws_bytes; NULL: one pass) and summed in fixed order.
extern "C" int mv2d_gemm_f32x3_batched(const float* A, long long lda, long long batch_a, int trans_a, const float* B, long long ldb, long long batch_b,
                                       int trans_b, float* C, long long ldc, long long batch_c, int M, int N, int K, int batch, float alpha, void* ws,
                                       long long ws_bytes, void* stream) {
    MV2D_CHECK_ARG(A && B && C && M >= 0 && N > 0 && K > 0 && batch >= 1 && batch <= 65535, "mv2d_gemm_f32x3_batched: bad args");
    if (M == 0) return MV2D_OK;
    int splits = fx_batched_splits(M, N, K, batch);
    if (splits > 1 && (!ws || ws_bytes < mv2d_gemm_f32x3_batched_ws_bytes(M, N, K, batch) || ((uintptr_t)ws & 15) != 0)) splits = 1;
    FxParams p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.bias = nullptr; p.M = M; p.N = N; p.K = K; p.act = 0; p.alpha = alpha; p.out_bf16 = 0; p.rowsum = nullptr;
    p.relu_y = nullptr; p.accumulate = 0; p.batch_a = batch_a; p.batch_b = batch_b;
    const int nk = cdiv(K, BK);
    p.k_tiles_per_split = cdiv(nk, splits);
    splits = cdiv(nk, p.k_tiles_per_split);
    if (splits > 1) { p.C = (float*)ws; p.ldc = N; p.batch_c = (long long)M * N; p.c_split_stride = (long long)batch * M * N; }
    else { p.C = C; p.ldc = ldc; p.batch_c = batch_c; p.c_split_stride = 0; }
    const dim3 grid(cdiv(M, BM) * cdiv(N, BN), splits, batch);
    hipStream_t st = (hipStream_t)stream;
    const bool deep = (long long)grid.x * splits * batch <= 1024;
#define MV2D_FX_LAUNCH(TA_, TB_) do { \
        if (deep) hipLaunchKernelGGL((gemm_f32x3_kernel<TA_, TB_, 4>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((gemm_f32x3_kernel<TA_, TB_, 1>), grid, dim3(256), 0, st, p); } while (0)
    if (trans_a && trans_b) MV2D_FX_LAUNCH(true, true);
    else if (trans_a) MV2D_FX_LAUNCH(true, false);
    else if (trans_b) MV2D_FX_LAUNCH(false, true);
    else MV2D_FX_LAUNCH(false, false);
#undef MV2D_FX_LAUNCH
    if (splits > 1) {
        const long long per = (long long)batch * M * N;
        hipLaunchKernelGGL(fx_batched_slab_sum_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, (const float*)ws, splits, batch, M, N, C, ldc, batch_c);
    }
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// Weight + bias gradient of a linear layer in one call: dW [N,K] = g^T x (g [M,N], x [M,K] dense rows) and db [N] = column sums of g -- inside the
// product's kernel when it runs in one pass (rows of op(A) = g^T summed by the blocks of the first column tile, fixed order), as a
// separate mv2d_colsum after a split-K product (many rows).  cs_scratch: [mv2d_colsum_scratch_rows(M), N] floats or NULL.
extern "C" int mv2d_wgrad_f32x3(const float* g, const float* x, float* dW, float* db, int M, int N, int K, void* ws, long long ws_bytes,
                                float* cs_scratch, void* stream) {
    MV2D_CHECK_ARG(g && x && dW && M >= 0 && N > 0 && K > 0, "mv2d_wgrad_f32x3: bad args");
    if (M == 0) return MV2D_OK;
    const bool one_pass = fx_splits(N, K, M, 0) <= 1 || !ws || ws_bytes < mv2d_gemm_f32x3_ws_bytes(N, K, M) || ((uintptr_t)ws & 255) != 0;
    const int rc = gemm_f32x3_impl(g, N, 1, x, K, 1, nullptr, 0, 1.f, 0, 0, dW, K, N, K, M, ws, ws_bytes, (db && one_pass) ? db : nullptr, nullptr, stream);
    if (rc != MV2D_OK || !db || one_pass) return rc;
    return mv2d_colsum_add(g, N, M, N, db, mv2d_colsum_scratch_rows(M) ? cs_scratch : nullptr, nullptr, stream);
}

static int gemm_f32x3_impl(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                           float alpha, int accumulate, int out_bf16, void* Cv, long long ldc, int M, int N, int K, void* ws, long long ws_bytes,
                           float* rowsum, const float* relu_y, void* stream) {
    float* C = (float*)Cv;
    MV2D_CHECK_ARG(!(out_bf16 && accumulate), "mv2d_gemm_f32x3_ex: accumulate needs an fp32 C");
    MV2D_CHECK_ARG(A && B && C && M >= 0 && N > 0 && K > 0 && (act == 0 || act == 1) && ldc >= N, "mv2d_gemm_f32x3: bad args");
    if (M == 0) return MV2D_OK;
    int splits = (out_bf16 || relu_y) ? 1 : fx_splits(M, N, K, act);
    if (splits > 1 && (ldc != N || !ws || ws_bytes < mv2d_gemm_f32x3_ws_bytes(M, N, K) || ((uintptr_t)ws & 255) != 0)) splits = 1;   // no slabs: one pass
    FxParams p;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.bias = bias; p.M = M; p.N = N; p.K = K; p.act = act; p.alpha = alpha; p.out_bf16 = out_bf16; p.rowsum = rowsum; p.relu_y = relu_y; p.batch_a = p.batch_b = p.batch_c = 0;
    const int nk = cdiv(K, BK);
    p.k_tiles_per_split = cdiv(nk, splits);
    splits = cdiv(nk, p.k_tiles_per_split);                 // (no empty split)
    float* slabs = (float*)ws;
    if (splits > 1) { p.C = slabs; p.ldc = N; p.c_split_stride = (long long)M * N; p.accumulate = 0; }
    else { p.C = C; p.ldc = ldc; p.c_split_stride = 0; p.accumulate = accumulate; }
    const dim3 grid(cdiv(M, BM) * cdiv(N, BN), splits);
    hipStream_t st = (hipStream_t)stream;
    // few blocks: the product's duration is one block's chain of stages -> four stages of loads in flight
    const bool deep = (long long)grid.x * grid.y <= 512;
#define MV2D_FX_LAUNCH(TA_, TB_) do { \
        if (deep) hipLaunchKernelGGL((gemm_f32x3_kernel<TA_, TB_, 4>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((gemm_f32x3_kernel<TA_, TB_, 1>), grid, dim3(256), 0, st, p); } while (0)
    if (trans_a && trans_b) MV2D_FX_LAUNCH(true, true);
    else if (trans_a) MV2D_FX_LAUNCH(true, false);
    else if (trans_b) MV2D_FX_LAUNCH(false, true);
    else MV2D_FX_LAUNCH(false, false);
#undef MV2D_FX_LAUNCH
    MV2D_LAUNCH_CHECK();
    if (splits > 1) {
        float* scratch = slabs + (long long)splits * M * N;
        return mv2d_colsum_add(slabs, (long long)M * N, splits, M * N, C, mv2d_colsum_scratch_rows(splits) ? scratch : nullptr, accumulate ? C : nullptr, stream);
    }
    return MV2D_OK;
}
