// The decoder of the head's TRAINING route as two C entries: forward and backward of the six post-norm layers of
// PETRTransformerDecoder (self attention, norm, cross attention, norm, FFN, norm; shared post_norm on every intermediate output;
// MU/petr_transformer.py:195-311,404-418,501-508,563-590; mmcv BaseTransformerLayer / FFN / MultiheadAttention residual + dropout rules).
//
// Rounds 3-4 drove this from torch autograd, one Python call per operator: ~1500 launches per step at ~11 us of host time each, and every
// linear layer's backward ran its three products one after the other on one stream.  Here the whole launch sequence is issued from C:
//   * the chain that the next operator waits for (dx) runs on the caller's stream;
//   * everything that only ends in a parameter gradient (dW = g^T x, db = column sums, the norms' partial sums) and the key-side work
//     (K / V projections of all layers in the forward; dK / dV, their projections' backward in the backward) is forked onto side streams and
//     joined once at the end -- these kernels are 4..160 workgroups each, so they run beside the chain on the other CUs;
//   * residual add + dropout + LayerNorm (+ query_pos add, + the shared post_norm) is one kernel in the forward, the LayerNorm backward takes
//     up to three incoming gradients and also writes the dropout-masked copy that the preceding linear layer's backward reads.
// The dense products are mv2d_gemm_f32x3_ex (fp32 operands, bf16 hi / lo split in the kernel), the attention is mv2d_sparse_xattn_*_drop
// (CSR; K / V in bf16 -- written as bf16 straight from the projection's epilogue).  Dropout masks are counter hashes of (seed, element),
// regenerated in the backward, never stored.  No buffer of the backward is reused inside one call (side streams read them asynchronously).
#include "common.h"
#include <stdlib.h>

extern "C" long long mv2d_gemm_f32x3_ws_bytes(int M, int N, int K);
extern "C" int mv2d_gemm_f32x3_ex(const float* A, long long lda, int trans_a, const float* B, long long ldb, int trans_b, const float* bias, int act,
                                  float alpha, int accumulate, int out_bf16, void* C, long long ldc, int M, int N, int K, void* ws,
                                  long long ws_bytes, void* stream);
extern "C" int mv2d_wgrad_f32x3(const float* g, const float* x, float* dW, float* db, int M, int N, int K, void* ws, long long ws_bytes,
                                float* cs_scratch, void* stream);
extern "C" int mv2d_colsum_scratch_rows(int rows);
extern "C" int mv2d_colsum_add(const float* x, long long ld, int rows, int cols, float* out, float* scratch, const float* add, void* stream);
extern "C" int mv2d_sparse_xattn_fwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx,
                                          float* ctx, float* dbg_logits, long long dbg_stride, int R, int empty_nan, float p_drop,
                                          unsigned int seed, void* stream);
extern "C" int mv2d_sparse_xattn_bwd_drop(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                                          const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                                          float* dq, float* dK, float* dV, int R, int S, float p_drop, unsigned int seed, void* stream);
extern "C" int mv2d_sparse_xattn_bwd_ex(const float* q, const void* K, const void* V, const int* row_ptr, const int* col_idx, const float* ctx,
                                        const float* dctx, const int* key_ptr, const int* pair_idx, const int* pair_row, float* pair_ws,
                                        float* dq, float* dK, float* dV, int R, int S, float p_drop, unsigned int seed, int long_rows, float dq_scale, void* stream);
extern "C" int mv2d_f32_to_bf16(const float* x, void* y, long long n, void* stream);
extern "C" long long mv2d_dense_attn_ws_bytes(int n, int nk, int backward);
extern "C" int mv2d_dense_attn_fwd(const float* q, const float* k, const float* v, int n, int nk, float p_drop, unsigned int seed, float* ctx, float* lse,
                                   void* ws, void* stream);
extern "C" int mv2d_dense_attn_bwd_parts(const float* q, const float* k, const float* v, const float* ctx, const float* dctx, const float* lse, int n,
                                         int nk, float p_drop, unsigned int seed, float dq_scale, float* dq, float* dk, float* dv, void* ws, int parts,
                                         void* stream);
extern "C" int mv2d_dgrad_relu_f32x3(const float* g, const float* W, const float* relu_y, float alpha, float* dx, int M, int N, int K, void* stream);

// the scalar arguments of both entries (mirrored by mv2d_amd/_lib.py: TdDims)
struct mv2d_td_dims {
    int T, S, L, F;                          // query rows, key rows, layers, FFN width
    int sa_nnz, ca_nnz;                      // allowed pairs of the self / cross attention pattern
    float p_sa_attn, p_sa_out, p_ca_attn, p_ca_out, p_ffn_act, p_ffn_out;      // dropout probabilities (0 in eval mode)
    unsigned int seed;
    float eps;
    int pad, nk;                             // leading denoising rows (they see the nk rows dn_keys of the key side, a dense block); 0, 0: none
};

namespace {

constexpr int C = 256, NPL = 18, NH = 8, HD = 32;             // channels; parameter tensors per layer
enum { SA_W, SA_B, SA_OW, SA_OB, N0_W, N0_B, CA_W, CA_B, CA_OW, CA_OB, N1_W, N1_B, F1_W, F1_B, F2_W, F2_B, N2_W, N2_B };

struct Drop { unsigned int thr, seed; float scale; };
static inline Drop mk_drop(float p, unsigned int seed) {
    Drop d{0u, seed, 1.f};
    if (p > 0.f) {
        const double t = (double)p * 4294967296.0;
        d.thr = t >= 4294967295.0 ? 0xffffffffu : (unsigned int)t;
        if (d.thr == 0u) d.thr = 1u;
        d.scale = 1.f / (1.f - p);
    }
    return d;
}
static inline unsigned int site_seed(unsigned int base, int layer, int site) { return base + 0x632BE5ABu * (unsigned int)(layer * 8 + site + 1); }

// keep / scale factor of element idx (murmur3 finaliser of (idx, seed), as the attention kernels' probability dropout)
__device__ __forceinline__ float drop_factor(const Drop& d, unsigned int idx) {
    if (d.thr == 0u) return 1.f;
    unsigned int u = (idx * 0x9E3779B1u) ^ d.seed;
    u ^= u >> 16; u *= 0x85EBCA6Bu; u ^= u >> 13; u *= 0xC2B2AE35u; u ^= u >> 16;
    return u >= d.thr ? d.scale : 0.f;
}
__device__ __forceinline__ float4 drop4(const Drop& d, const float4& v, unsigned int idx) {
    return make_float4(v.x * drop_factor(d, idx), v.y * drop_factor(d, idx + 1), v.z * drop_factor(d, idx + 2), v.w * drop_factor(d, idx + 3));
}
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// (the arithmetic order of row_ln_kernel's ln4, csrc/rows.hip: the two training routes then produce the same bits, so that no ReLU unit
//  switches between them)
__device__ __forceinline__ float4 ln_row(const float4& v, const float* w, const float* b, int c0, float eps) {
    const float s = wave_sum(v.x + v.y + v.z + v.w);
    const float mean = s * (1.0f / C);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / C);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 ww = ld4(w + c0), bb = ld4(b + c0);
    return make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
}

// s = x + dropout(o);  y = LN(s);  yq = y + qpos (optional);  y2 = LN2(y) (optional: the shared post_norm).  One wave per row.
struct ResLnArgs {
    const float *x, *o, *w, *b, *qpos, *w2, *b2;
    float *s, *y, *yq, *y2;
    int M; float eps; Drop d;
};
__global__ __launch_bounds__(256) void res_ln_kernel(ResLnArgs p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), c0 = 4 * (threadIdx.x & 63);
    if (row >= p.M) return;
    const long long e = (long long)row * C + c0;
    const float4 s = add4(ld4(p.x + e), drop4(p.d, ld4(p.o + e), (unsigned int)e));
    st4(p.s + e, s);
    const float4 y = ln_row(s, p.w, p.b, c0, p.eps);
    st4(p.y + e, y);
    if (p.yq) st4(p.yq + e, add4(y, ld4(p.qpos + e)));
    if (p.y2) st4(p.y2 + e, ln_row(y, p.w2, p.b2, c0, p.eps));
}

// LayerNorm backward over rows of 256 with up to three incoming gradients (dy = dy0 + dy1 + dy2) and, optionally, the dropout-masked copy
// of the result (the gradient of the residual branch's last linear layer): dx, dx_drop; partial column sums of dy * xhat / dy per block.
constexpr int LNB_ROWS = 16;
struct LnBwdArgs {
    const float *x, *dy0, *dy1, *dy2, *w;
    float *dx, *dx_drop, *dw_part, *db_part;
    int M; float eps; Drop d;
    const float* relu_y;        // y = relu(LN(x)) of the forward: dy counts only where y > 0 (the Linear-LN-ReLU of the class branch), or NULL
};
__global__ __launch_bounds__(256) void ln_bwd_ex_kernel(LnBwdArgs p) {
    __shared__ float sw[4][C], sb[4][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 wv = ld4(p.w + 4 * lane);
    float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
    const int r_begin = blockIdx.x * LNB_ROWS;
    for (int r = r_begin + wave; r < min(r_begin + LNB_ROWS, p.M); r += 4) {
        const long long e = (long long)r * C + 4 * lane;
        const float4 xv = ld4(p.x + e);
        float4 dv = ld4(p.dy0 + e);
        if (p.dy1) dv = add4(dv, ld4(p.dy1 + e));
        if (p.dy2) dv = add4(dv, ld4(p.dy2 + e));
        if (p.relu_y) {
            const float4 yv = ld4(p.relu_y + e);
            dv = make_float4(yv.x > 0.f ? dv.x : 0.f, yv.y > 0.f ? dv.y : 0.f, yv.z > 0.f ? dv.z : 0.f, yv.w > 0.f ? dv.w : 0.f);
        }
        const float mean = wave_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.f / C);
        const float4 c = make_float4(xv.x - mean, xv.y - mean, xv.z - mean, xv.w - mean);
        const float var = wave_sum((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.f / C);
        const float rstd = 1.f / sqrtf(var + p.eps);
        const float4 xh = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
        const float4 g = make_float4(dv.x * wv.x, dv.y * wv.y, dv.z * wv.z, dv.w * wv.w);
        const float mg = wave_sum((g.x + g.y) + (g.z + g.w)) * (1.f / C);
        const float mgx = wave_sum((g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w)) * (1.f / C);
        const float4 dx = make_float4(rstd * (g.x - mg - xh.x * mgx), rstd * (g.y - mg - xh.y * mgx), rstd * (g.z - mg - xh.z * mgx),
                                      rstd * (g.w - mg - xh.w * mgx));
        st4(p.dx + e, dx);
        if (p.dx_drop) st4(p.dx_drop + e, drop4(p.d, dx, (unsigned int)e));
        aw.x += dv.x * xh.x; aw.y += dv.y * xh.y; aw.z += dv.z * xh.z; aw.w += dv.w * xh.w;
        ab.x += dv.x; ab.y += dv.y; ab.z += dv.z; ab.w += dv.w;
    }
    st4(&sw[wave][4 * lane], aw);
    st4(&sb[wave][4 * lane], ab);
    __syncthreads();
    const int c = threadIdx.x;
    p.dw_part[(long long)blockIdx.x * C + c] = (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]);
    p.db_part[(long long)blockIdx.x * C + c] = (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]);
}

// y = relu(LN(x))   (Linear-LN-ReLU of the class branch, cross_attention_head.py:127-133)
__global__ __launch_bounds__(256) void ln_relu_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                      float* __restrict__ y, int M, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), c0 = 4 * (threadIdx.x & 63);
    if (row >= M) return;
    const long long e = (long long)row * C + c0;
    const float4 v = ln_row(ld4(x + e), w, b, c0, eps);
    st4(y + e, make_float4(relu_f(v.x), relu_f(v.y), relu_f(v.z), relu_f(v.w)));
}

// out = a + b
__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st4(out + 4 * i, add4(ld4(a + 4 * i), ld4(b + 4 * i)));
}
// x *= s
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long n4, float s) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { float4 v = ld4(x + 4 * i); st4(x + 4 * i, make_float4(v.x * s, v.y * s, v.z * s, v.w * s)); }
}
// h = dropout(h)   (after the FFN's ReLU)
__global__ __launch_bounds__(256) void drop_kernel(float* __restrict__ h, long long n4, Drop d) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st4(h + 4 * i, drop4(d, ld4(h + 4 * i), (unsigned int)(4 * i)));
}
// g = h > 0 ? g * scale : 0   (h = dropout(relu(.)) is positive exactly where the unit was active AND kept)
__global__ __launch_bounds__(256) void relu_mask_scale_kernel(float* __restrict__ g, const float* __restrict__ h, long long n4, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4 d = ld4(g + 4 * i), v = ld4(h + 4 * i);
        st4(g + 4 * i, make_float4(v.x > 0.f ? d.x * scale : 0.f, v.y > 0.f ? d.y * scale : 0.f, v.z > 0.f ? d.z * scale : 0.f, v.w > 0.f ? d.w * scale : 0.f));
    }
}
// out0[c] = sum_r x0[r, c], out1[c] = sum_r x1[r, c] over [rows, 256] partial sums (blockIdx.y picks the pair): 16 columns x 16 row groups
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int rows, float* __restrict__ out0,
                                                      float* __restrict__ out1) {
    __shared__ float part[16][17];
    const float* x = blockIdx.y ? x1 : x0;
    float* out = blockIdx.y ? out1 : out0;
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    float s = 0.f;
    for (int r = g; r < rows; r += 16) s += x[(long long)r * C + c];
    part[g][cl] = s;
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][cl];
        out[c] = t;
    }
}
// dst[i, :] = src[idx[i], :]  (rows of 256; idx NULL: the identity)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), c = 4 * (threadIdx.x & 63);
    if (i >= n) return;
    st4(dst + (long long)i * C + c, ld4(src + (long long)(idx ? idx[i] : i) * C + c));
}
// dst[idx[i], :] += src[i, :]  (idx holds every row at most once: no atomics; idx NULL: the identity)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* __restrict__ dst, const int* __restrict__ idx, const float* __restrict__ src, int n) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), c = 4 * (threadIdx.x & 63);
    if (i >= n) return;
    float* d = dst + (long long)(idx ? idx[i] : i) * C + c;
    st4(d, add4(ld4(d), ld4(src + (long long)i * C + c)));
}
// out = sum of n buffers (fixed order)
struct SumArgs { const float* src[16]; int n; };
__global__ __launch_bounds__(256) void sum_n_kernel(SumArgs a, float* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = ld4(a.src[0] + 4 * i);
    for (int k = 1; k < a.n; ++k) s = add4(s, ld4(a.src[k] + 4 * i));
    st4(out + 4 * i, s);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int NSIDE = 3, NEV = 128;
struct Pool { bool ready = false; hipStream_t side[NSIDE]; hipEvent_t ev[NEV]; int next = 0; };
Pool g_pool[16];

// MV2D_TD_SERIAL=1 (diagnostics): every "side stream" is the caller's stream -- the same launches without any overlap
static bool td_serial() {
    static const bool v = [] { const char* e = getenv("MV2D_TD_SERIAL"); return e && e[0] == '1'; }();
    return v;
}

static Pool* pool() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    Pool& p = g_pool[dev];
    if (!p.ready) {
        for (int i = 0; i < NSIDE; ++i)
            if (hipStreamCreateWithFlags(&p.side[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 0; i < NEV; ++i)
            if (hipEventCreateWithFlags(&p.ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        p.ready = true;
    }
    return &p;
}
// `to` continues only after everything issued to `from` so far
static void after(Pool* p, hipStream_t from, hipStream_t to) {
    if (from == to) return;
    hipEvent_t e = p->ev[p->next];
    p->next = (p->next + 1) % NEV;
    (void)hipEventRecord(e, from);
    (void)hipStreamWaitEvent(to, e, 0);
}

// Error path of the entries below (TD_RC returns as soon as a launch fails): the side streams were forked off the caller's stream and may still read / write
// act and ws; the caller frees them as soon as it sees the error.  The guard joins every side stream back into the caller's stream on any exit that did not
// reach the regular join at the end.
struct SideJoin {
    Pool* p; hipStream_t st; bool armed = true;
    ~SideJoin() {
        if (!armed || !p) return;
        for (int i = 0; i < NSIDE; ++i) after(p, p->side[i], st);
    }
};

static inline long long al256(long long b) { return (b + 255) & ~255LL; }

struct Carver {
    char* base; long long off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <class T> T* take(long long count) { T* r = (T*)(base + off); off += al256(count * (long long)sizeof(T)); return r; }
};

static long long gemm_ws_max(const mv2d_td_dims& d) {
    const int T = d.T, S = d.S, F = d.F;
    const int shapes[][3] = {{T, C, C}, {T, F, C}, {T, C, F}, {C, C, T}, {F, C, T}, {C, F, T}, {C, C, S}, {S, C, C}};
    long long m = 0;
    for (auto& s : shapes) { const long long b = mv2d_gemm_f32x3_ws_bytes(s[0], s[1], s[2]); if (b > m) m = b; }
    return al256(m) + 256;
}

// a stream with its own split-K workspace and column-sum scratch
struct Lane {
    hipStream_t st; void* ws; long long ws_bytes; float* cs;
    int gemm(const float* A, long long lda, int ta, const float* B, long long ldb, int tb, const float* bias, int act, float alpha, int acc, int bf16,
             void* Cm, long long ldc, int M, int N, int K) const {
        return mv2d_gemm_f32x3_ex(A, lda, ta, B, ldb, tb, bias, act, alpha, acc, bf16, Cm, ldc, M, N, K, ws, ws_bytes, st);
    }
    // y [M,N] = act((x [M,K] W[N,K]^T + b) * alpha)
    int linear(const float* x, const float* W, const float* b, int act, float alpha, int bf16, void* y, int M, int N, int K) const {
        return gemm(x, K, 0, W, K, 0, b, act, alpha, 0, bf16, y, N, M, N, K);
    }
    // dx [M,K] (+)= g [M,N] W [N,K]
    int dgrad(const float* g, const float* W, float* dx, int M, int N, int K, int acc) const { return gemm(g, N, 0, W, K, 1, nullptr, 0, 1.f, acc, 0, dx, K, M, K, N); }
    // dW [N,K] = g^T x, db [N] = column sums of g
    int wgrad(const float* g, const float* x, float* dW, float* db, int M, int N, int K) const {
        return mv2d_wgrad_f32x3(g, x, dW, db, M, N, K, ws, ws_bytes, cs, st);
    }
    // the two parameter gradients of a LayerNorm from the per-block partial sums of ln_bwd_ex_kernel: one launch
    void ln_params(const float* pw, const float* pb, int nb, float* gw, float* gb) const;
};

void Lane::ln_params(const float* pw, const float* pb, int nb, float* gw, float* gb) const {
    hipLaunchKernelGGL(colsum2_kernel, dim3(C / 16, 2), dim3(256), 0, st, pw, pb, nb, gw, gb);
}

static inline unsigned int blocks4(long long n) { return (unsigned int)((n / 4 + 255) / 256); }

// the activations the backward needs, per layer (fp32 unless noted)
struct Act {
    float *xq, *q_sa, *ctx_sa, *s1, *x1, *xq1, *q_ca, *ctx_ca, *s2, *x2, *h, *s3, *x3;
    unsigned short *k_sa, *v_sa, *K, *V;
    float *kd, *vd, *lse;                   // denoising rows: projected key / value rows they see [nk, C] fp32, log-sum-exp of their rows [NH, pad]
};
struct ActLayout {
    float* x0; Act a[8];
    long long bytes;
    ActLayout(const mv2d_td_dims& d, void* base) {
        Carver c(base);
        const long long TC = (long long)d.T * C;
        x0 = c.take<float>(TC);
        for (int l = 0; l < d.L; ++l) {
            Act& a_ = a[l];
            a_.xq = c.take<float>(TC); a_.q_sa = c.take<float>(TC); a_.ctx_sa = c.take<float>(TC); a_.s1 = c.take<float>(TC); a_.x1 = c.take<float>(TC);
            a_.xq1 = c.take<float>(TC); a_.q_ca = c.take<float>(TC); a_.ctx_ca = c.take<float>(TC); a_.s2 = c.take<float>(TC); a_.x2 = c.take<float>(TC);
            a_.h = c.take<float>((long long)d.T * d.F); a_.s3 = c.take<float>(TC); a_.x3 = c.take<float>(TC);
            a_.k_sa = c.take<unsigned short>(TC); a_.v_sa = c.take<unsigned short>(TC);
            a_.K = c.take<unsigned short>((long long)d.S * C); a_.V = c.take<unsigned short>((long long)d.S * C);
            a_.kd = a_.vd = a_.lse = nullptr;
            if (d.nk > 0) {
                a_.kd = c.take<float>((long long)d.nk * C); a_.vd = c.take<float>((long long)d.nk * C);
                a_.lse = c.take<float>((long long)NH * d.pad);
            }
        }
        bytes = c.off;
    }
};

static bool dims_ok(const mv2d_td_dims* d) {
    return d && d->T > 0 && d->S > 0 && d->L >= 1 && d->L <= 8 && d->F > 0 && d->F % 4 == 0 && d->sa_nnz >= 0 && d->ca_nnz >= 0 && d->pad >= 0 &&
           d->pad <= d->T && d->nk >= 0 && d->nk <= d->S && ((d->pad > 0) == (d->nk > 0)) &&
           (long long)NH * d->pad * ((d->nk + 31) & ~31) < (1LL << 32);        // (the dropout counter of the dense block is 32 bits wide)
}

#define TD_RC(x) do { const int rc_ = (x); if (rc_ != MV2D_OK) return rc_; } while (0)

}  // namespace

extern "C" long long mv2d_train_decoder_act_bytes(const mv2d_td_dims* d) {
    if (!dims_ok(d)) return -1;
    return ActLayout(*d, nullptr).bytes + 256;
}

extern "C" long long mv2d_train_decoder_ws_bytes(const mv2d_td_dims* d, int backward) {
    if (!dims_ok(d)) return -1;
    const long long TC = (long long)d->T * C, SC = (long long)d->S * C, TF = (long long)d->T * d->F;
    const long long lanes = (1 + NSIDE) * (gemm_ws_max(*d) + al256((long long)(mv2d_colsum_scratch_rows(d->S > d->T ? d->S : d->T) + 1) * d->F * 4));
    const long long KB = al256((long long)d->nk * C * 4);                                                                 // 0 without denoising rows
    const long long DF = d->nk > 0 ? al256(mv2d_dense_attn_ws_bytes(d->pad, d->nk, 0)) : 0, DB = d->nk > 0 ? al256(mv2d_dense_attn_ws_bytes(d->pad, d->nk, 1)) : 0;
    if (!backward) return lanes + al256(TC * 4) + (d->nk > 0 ? DF + 2 * al256(SC * 4) : 0) + 4096;
    const int nb = cdiv(d->T, LNB_ROWS);
    const long long per_layer = 16 * al256(TC * 4) + 2 * al256(TF * 4) + 2 * al256(SC * 4) + al256((long long)(d->sa_nnz > 0 ? d->sa_nnz : 1) * 64) +
                                al256((long long)(d->ca_nnz > 0 ? d->ca_nnz : 1) * 64) + 10 * al256((long long)nb * C * 4) + DB + 2 * KB;
    return lanes + d->L * per_layer + 4096;
}

// params: 18 L + 2 device pointers (per layer: self-attention in_proj weight [3C,C] / bias, out_proj weight / bias, norm 0 weight / bias; the same
// for the cross attention + norm 1; FFN linear 1 weight [F,C] / bias, linear 2 weight [C,F] / bias, norm 2; then post_norm weight / bias).
// qpos [T,C], key_in / val_in [S,C] fp32; CSR patterns of the two attentions; outs [L,T,C]: post_norm of every layer's output.
// act: mv2d_train_decoder_act_bytes, kept by the caller until the backward; ws: mv2d_train_decoder_ws_bytes(d, 0).  256-byte aligned.
extern "C" int mv2d_train_decoder_fwd(const mv2d_td_dims* d, const float* const* params, const float* qpos, const float* key_in, const float* val_in,
                                      const int* sa_row_ptr, const int* sa_col, const int* ca_row_ptr, const int* ca_col, const int* dn_keys, float* outs,
                                      void* act, void* ws, void* stream) {
    MV2D_CHECK_ARG(dims_ok(d) && params && qpos && key_in && val_in && sa_row_ptr && sa_col && ca_row_ptr && ca_col && outs && act && ws,
                   "mv2d_train_decoder_fwd: bad args");
    MV2D_CHECK_ARG(dn_keys || d->nk == 0 || d->nk == d->S, "mv2d_train_decoder_fwd: dn_keys may only be NULL when the denoising rows see every key");
    MV2D_CHECK_ARG((((uintptr_t)act | (uintptr_t)ws) & 255) == 0, "mv2d_train_decoder_fwd: act / ws must be 256-byte aligned");
    Pool* pl = pool();
    MV2D_CHECK_ARG(pl != nullptr, "mv2d_train_decoder_fwd: could not create the side streams");
    SideJoin side_join{pl, (hipStream_t)stream};
    const int T = d->T, S = d->S, L = d->L, F = d->F;
    const long long TC = (long long)T * C;
    const float qs = 1.f / sqrtf((float)(C / 8));
    hipStream_t st = (hipStream_t)stream;
    ActLayout al(*d, act);
    Carver cw(ws);
    const long long gws = gemm_ws_max(*d), csb = (long long)(mv2d_colsum_scratch_rows(S > T ? S : T) + 1) * F;
    Lane mainl{st, cw.take<char>(gws), gws, cw.take<float>(csb)};
    Lane side[NSIDE];
    for (int i = 0; i < NSIDE; ++i) side[i] = Lane{td_serial() ? st : pl->side[i], cw.take<char>(gws), gws, cw.take<float>(csb)};
    float* tmp = cw.take<float>(TC);
    const int pad = d->pad, nk = d->nk;
    float *k32 = nullptr, *v32 = nullptr;
    char* dn_ws = nullptr;
    if (nk > 0) { dn_ws = cw.take<char>(mv2d_dense_attn_ws_bytes(pad, nk, 0)); k32 = cw.take<float>((long long)S * C); v32 = cw.take<float>((long long)S * C); }

    // the key side of all layers does not depend on the queries: K_l = key_in Wk_l^T + bk_l, V_l = val_in Wv_l^T + bv_l (bf16) on two side streams
    // (with denoising rows: in fp32 first -- their dense block reads the rows dn_keys of it unrounded, as the per-operator graph does)
    after(pl, st, side[1].st);
    after(pl, st, side[2].st);
    hipEvent_t kv_ready[8][2];
    for (int l = 0; l < L; ++l) {
        const float* const* P = params + l * NPL;
        if (nk > 0) {
            float* kf = dn_keys ? k32 : al.a[l].kd;
            float* vf = dn_keys ? v32 : al.a[l].vd;
            TD_RC(side[1].linear(key_in, P[CA_W] + (long long)C * C, P[CA_B] + C, 0, 1.f, 0, kf, S, C, C));
            TD_RC(mv2d_f32_to_bf16(kf, al.a[l].K, (long long)S * C, side[1].st));
            TD_RC(side[2].linear(val_in, P[CA_W] + 2LL * C * C, P[CA_B] + 2 * C, 0, 1.f, 0, vf, S, C, C));
            TD_RC(mv2d_f32_to_bf16(vf, al.a[l].V, (long long)S * C, side[2].st));
            if (dn_keys) {
                hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(nk, 4)), dim3(256), 0, side[1].st, (const float*)kf, dn_keys, al.a[l].kd, nk);
                hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(nk, 4)), dim3(256), 0, side[2].st, (const float*)vf, dn_keys, al.a[l].vd, nk);
            }
        } else {
            TD_RC(side[1].linear(key_in, P[CA_W] + (long long)C * C, P[CA_B] + C, 0, 1.f, 1, al.a[l].K, S, C, C));
            TD_RC(side[2].linear(val_in, P[CA_W] + 2LL * C * C, P[CA_B] + 2 * C, 0, 1.f, 1, al.a[l].V, S, C, C));
        }
        for (int j = 0; j < 2; ++j) {
            kv_ready[l][j] = pl->ev[pl->next];
            pl->next = (pl->next + 1) % NEV;
            (void)hipEventRecord(kv_ready[l][j], side[1 + j].st);
        }
    }
    (void)hipMemsetAsync(al.x0, 0, TC * 4, st);
    const float* x_in = al.x0;
    for (int l = 0; l < L; ++l) {
        const float* const* P = params + l * NPL;
        const Act& a = al.a[l];
        // ---- self attention: q = k = x + query_pos, v = x
        if (l == 0) hipLaunchKernelGGL(add2_kernel, dim3(blocks4(TC)), dim3(256), 0, st, x_in, qpos, a.xq, TC / 4);
        TD_RC(mainl.linear(a.xq, P[SA_W], P[SA_B], 0, qs, 0, a.q_sa, T, C, C));
        TD_RC(mainl.linear(a.xq, P[SA_W] + (long long)C * C, P[SA_B] + C, 0, 1.f, 1, a.k_sa, T, C, C));
        TD_RC(mainl.linear(x_in, P[SA_W] + 2LL * C * C, P[SA_B] + 2 * C, 0, 1.f, 1, a.v_sa, T, C, C));
        TD_RC(mv2d_sparse_xattn_fwd_drop(a.q_sa, a.k_sa, a.v_sa, sa_row_ptr, sa_col, a.ctx_sa, nullptr, 0, T, 0, d->p_sa_attn, site_seed(d->seed, l, 0), st));
        TD_RC(mainl.linear(a.ctx_sa, P[SA_OW], P[SA_OB], 0, 1.f, 0, tmp, T, C, C));
        {
            ResLnArgs r{x_in, tmp, P[N0_W], P[N0_B], qpos, nullptr, nullptr, a.s1, a.x1, a.xq1, nullptr, T, d->eps, mk_drop(d->p_sa_out, site_seed(d->seed, l, 1))};
            hipLaunchKernelGGL(res_ln_kernel, dim3(cdiv(T, 4)), dim3(256), 0, st, r);
        }
        // ---- cross attention: q = x + query_pos; keys / values from the side streams
        TD_RC(mainl.linear(a.xq1, P[CA_W], P[CA_B], 0, qs, 0, a.q_ca, T, C, C));
        (void)hipStreamWaitEvent(st, kv_ready[l][0], 0);
        (void)hipStreamWaitEvent(st, kv_ready[l][1], 0);
        if (nk > 0) {
            // the denoising rows: a dense block over the rows dn_keys (csrc/dense_attn.hip: the logits never leave the MFMA accumulators)
            TD_RC(mv2d_dense_attn_fwd(a.q_ca, a.kd, a.vd, pad, nk, d->p_ca_attn, site_seed(d->seed, l, 6), a.ctx_ca, a.lse, dn_ws, st));
        }
        if (T > pad)
            TD_RC(mv2d_sparse_xattn_fwd_drop(a.q_ca + (long long)pad * C, a.K, a.V, ca_row_ptr, ca_col, a.ctx_ca + (long long)pad * C, nullptr, 0, T - pad, 0,
                                             d->p_ca_attn, site_seed(d->seed, l, 2), st));
        TD_RC(mainl.linear(a.ctx_ca, P[CA_OW], P[CA_OB], 0, 1.f, 0, tmp, T, C, C));
        {
            ResLnArgs r{a.x1, tmp, P[N1_W], P[N1_B], nullptr, nullptr, nullptr, a.s2, a.x2, nullptr, nullptr, T, d->eps, mk_drop(d->p_ca_out, site_seed(d->seed, l, 3))};
            hipLaunchKernelGGL(res_ln_kernel, dim3(cdiv(T, 4)), dim3(256), 0, st, r);
        }
        // ---- FFN: Linear-ReLU-Dropout, Linear-Dropout, + identity
        TD_RC(mainl.linear(a.x2, P[F1_W], P[F1_B], 1, 1.f, 0, a.h, T, F, C));
        if (d->p_ffn_act > 0.f)
            hipLaunchKernelGGL(drop_kernel, dim3(blocks4((long long)T * F)), dim3(256), 0, st, a.h, (long long)T * F / 4, mk_drop(d->p_ffn_act, site_seed(d->seed, l, 4)));
        TD_RC(mainl.linear(a.h, P[F2_W], P[F2_B], 0, 1.f, 0, tmp, T, C, F));
        {
            // the next layer's q = k input (x3 + query_pos) and this layer's intermediate output (post_norm) come out of the same kernel
            ResLnArgs r{a.x2, tmp, P[N2_W], P[N2_B], qpos, params[L * NPL], params[L * NPL + 1], a.s3, a.x3, l + 1 < L ? al.a[l + 1].xq : nullptr,
                        outs + (long long)l * TC, T, d->eps, mk_drop(d->p_ffn_out, site_seed(d->seed, l, 5))};
            hipLaunchKernelGGL(res_ln_kernel, dim3(cdiv(T, 4)), dim3(256), 0, st, r);
        }
        x_in = a.x3;
    }
    MV2D_LAUNCH_CHECK();
    for (int i = 1; i < NSIDE; ++i) after(pl, side[i].st, st);
    side_join.armed = false;
    return MV2D_OK;
}

// Backward of mv2d_train_decoder_fwd.  grads: 18 L + 2 device pointers, the gradient of every parameter (written, not accumulated); d_outs
// [L,T,C]; the transposed patterns (ops.csr_transpose: key_ptr [.+1], pair_idx [nnz], pair_row [nnz]) of both attentions; d_qpos [T,C],
// d_key_in / d_val_in [S,C] (written).  ws: mv2d_train_decoder_ws_bytes(d, 1).
extern "C" int mv2d_train_decoder_bwd(const mv2d_td_dims* d, const float* const* params, float* const* grads, const float* qpos, const float* key_in,
                                      const float* val_in, const int* sa_row_ptr, const int* sa_col, const int* sa_key_ptr, const int* sa_pair_idx,
                                      const int* sa_pair_row, const int* ca_row_ptr, const int* ca_col, const int* ca_key_ptr, const int* ca_pair_idx,
                                      const int* ca_pair_row, const int* dn_keys, const float* d_outs, const void* act, void* ws, float* d_qpos,
                                      float* d_key_in, float* d_val_in, void* stream) {
    MV2D_CHECK_ARG(dims_ok(d) && params && grads && qpos && key_in && val_in && sa_row_ptr && sa_col && sa_key_ptr && sa_pair_idx && sa_pair_row &&
                   ca_row_ptr && ca_col && ca_key_ptr && ca_pair_idx && ca_pair_row && d_outs && act && ws && d_qpos && d_key_in && d_val_in,
                   "mv2d_train_decoder_bwd: bad args");
    MV2D_CHECK_ARG((((uintptr_t)act | (uintptr_t)ws) & 255) == 0, "mv2d_train_decoder_bwd: act / ws must be 256-byte aligned");
    MV2D_CHECK_ARG(2 * d->L <= 16, "mv2d_train_decoder_bwd: too many layers");
    MV2D_CHECK_ARG(dn_keys || d->nk == 0 || d->nk == d->S, "mv2d_train_decoder_bwd: dn_keys may only be NULL when the denoising rows see every key");
    Pool* pl = pool();
    MV2D_CHECK_ARG(pl != nullptr, "mv2d_train_decoder_bwd: could not create the side streams");
    SideJoin side_join{pl, (hipStream_t)stream};
    const int T = d->T, S = d->S, L = d->L, F = d->F;
    const long long TC = (long long)T * C, SC = (long long)S * C, TF = (long long)T * F;
    const float qs = 1.f / sqrtf((float)(C / 8));
    hipStream_t st = (hipStream_t)stream;
    ActLayout al(*d, const_cast<void*>(act));
    Carver cw(ws);
    const long long gws = gemm_ws_max(*d), csb = (long long)(mv2d_colsum_scratch_rows(S > T ? S : T) + 1) * F;
    Lane mainl{st, cw.take<char>(gws), gws, cw.take<float>(csb)};
    Lane side[NSIDE];
    for (int i = 0; i < NSIDE; ++i) side[i] = Lane{td_serial() ? st : pl->side[i], cw.take<char>(gws), gws, cw.take<float>(csb)};
    const int nb = cdiv(T, LNB_ROWS);
    const float* post_w = params[L * NPL];

    struct Bufs {
        float *dxa, *A, *Q1, *ds3, *dyf, *dffn, *ds2, *do2, *dctx2, *dq2, *ds1, *do1, *dctx1, *dq1, *dk1, *dv1, *dh, *dK, *dV, *pw_sa, *pw_ca;
        float* part[8];
        float *dkd, *dvd; char* dws;           // denoising rows: gradients of the key / value rows they see [nk, C]; scratch of the dense block's backward
    } B[8];
    const int pad = d->pad, nk = d->nk;
    for (int l = 0; l < L; ++l) {
        Bufs& b = B[l];
        float** t16[] = {&b.dxa, &b.A, &b.Q1, &b.ds3, &b.dyf, &b.dffn, &b.ds2, &b.do2, &b.dctx2, &b.dq2, &b.ds1, &b.do1, &b.dctx1, &b.dq1, &b.dk1, &b.dv1};
        for (float** t : t16) *t = cw.take<float>(TC);
        b.dh = cw.take<float>(TF);
        (void)cw.take<float>(TF);
        b.dK = cw.take<float>(SC); b.dV = cw.take<float>(SC);
        b.pw_sa = cw.take<float>((long long)(d->sa_nnz > 0 ? d->sa_nnz : 1) * 16);
        b.pw_ca = cw.take<float>((long long)(d->ca_nnz > 0 ? d->ca_nnz : 1) * 16);
        for (int j = 0; j < 8; ++j) b.part[j] = cw.take<float>((long long)nb * C);
        b.dkd = b.dvd = nullptr; b.dws = nullptr;
        if (nk > 0) { b.dws = cw.take<char>(mv2d_dense_attn_ws_bytes(pad, nk, 1)); b.dkd = cw.take<float>((long long)nk * C); b.dvd = cw.take<float>((long long)nk * C); }
    }
    // the small weight-gradient products go to side[0]; the key side of the cross attention to side[1] (keys) and side[2] (values)
    for (int i = 0; i < NSIDE; ++i) after(pl, st, side[i].st);

    auto ln_bwd = [&](const float* x, const float* dy0, const float* dy1, const float* dy2, const float* w, float* dx, float* dx_drop, const Drop& dr,
                      float* pw, float* pb) {
        LnBwdArgs a{x, dy0, dy1, dy2, w, dx, dr.thr ? dx_drop : nullptr, pw, pb, T, d->eps, dr, nullptr};
        hipLaunchKernelGGL(ln_bwd_ex_kernel, dim3(nb), dim3(256), 0, st, a);
    };

    // post_norm of every intermediate output: independent of the chain below
    float* post_pw = cw.take<float>((long long)L * nb * C);
    float* post_pb = cw.take<float>((long long)L * nb * C);
    for (int l = 0; l < L; ++l) {
        LnBwdArgs a{al.a[l].x3, d_outs + (long long)l * TC, nullptr, nullptr, post_w, B[l].dxa, nullptr, post_pw + (long long)l * nb * C,
                    post_pb + (long long)l * nb * C, T, d->eps, mk_drop(0.f, 0u), nullptr};
        hipLaunchKernelGGL(ln_bwd_ex_kernel, dim3(nb), dim3(256), 0, st, a);
    }
    after(pl, st, side[0].st);
    side[0].ln_params(post_pw, post_pb, L * nb, grads[L * NPL], grads[L * NPL + 1]);

    const float *g0 = nullptr, *g1 = nullptr, *g2 = nullptr;           // the gradient w.r.t. the output of layer l: g0 + g1 + g2
    bool first_kv = true;
    for (int l = L - 1; l >= 0; --l) {
        const float* const* P = params + l * NPL;
        float* const* G = grads + l * NPL;
        const Act& a = al.a[l];
        Bufs& b = B[l];
        const float* x_in = l > 0 ? al.a[l - 1].x3 : al.x0;
        if (l == L - 1) { g0 = b.dxa; g1 = g2 = nullptr; }
        // ================= the chain (caller's stream): every kernel the next one waits for
        // ---- norm 2 and the FFN
        const Drop d5 = mk_drop(d->p_ffn_out, site_seed(d->seed, l, 5));
        ln_bwd(a.s3, g0, g1, g2, P[N2_W], b.ds3, b.dyf, d5, b.part[0], b.part[1]);
        const float* dyf = d5.thr ? b.dyf : b.ds3;
        // dh = (dy W2) / keep rate where h > 0 (h = dropout(relu(.)) is positive exactly where the unit was active AND kept): mask in the epilogue
        TD_RC(mv2d_dgrad_relu_f32x3(dyf, P[F2_W], a.h, d->p_ffn_act > 0.f ? 1.f / (1.f - d->p_ffn_act) : 1.f, b.dh, T, C, F, st));
        TD_RC(mainl.dgrad(b.dh, P[F1_W], b.dffn, T, F, C, 0));
        // ---- norm 1 and the cross attention (queries; the key pass goes to the side streams below)
        const Drop d3 = mk_drop(d->p_ca_out, site_seed(d->seed, l, 3));
        ln_bwd(a.s2, b.ds3, b.dffn, nullptr, P[N1_W], b.ds2, b.do2, d3, b.part[2], b.part[3]);
        const float* do2 = d3.thr ? b.do2 : b.ds2;
        TD_RC(mainl.dgrad(do2, P[CA_OW], b.dctx2, T, C, C, 0));
        if (nk > 0)      // the denoising rows' dense block, query side (dq, scaled by 1 / sqrt(d)); its key side follows on the side stream below
            TD_RC(mv2d_dense_attn_bwd_parts(a.q_ca, a.kd, a.vd, a.ctx_ca, b.dctx2, a.lse, pad, nk, d->p_ca_attn, site_seed(d->seed, l, 6), qs, b.dq2, b.dkd, b.dvd,
                                            b.dws, 1, st));
        const long long po = (long long)pad * C;
        TD_RC(mv2d_sparse_xattn_bwd_ex(a.q_ca + po, a.K, a.V, ca_row_ptr, ca_col, a.ctx_ca + po, b.dctx2 + po, ca_key_ptr, ca_pair_idx, ca_pair_row, b.pw_ca,
                                       b.dq2 + po, b.dK, b.dV, T - pad, 0, d->p_ca_attn, site_seed(d->seed, l, 2), 0, qs, st));
        // the key pass of the cross attention (dK, dV) and the projections behind it: side streams 1 (keys) and 2 (values)
        after(pl, st, side[1].st);
        TD_RC(mv2d_sparse_xattn_bwd_ex(a.q_ca + po, a.K, a.V, ca_row_ptr, ca_col, a.ctx_ca + po, b.dctx2 + po, ca_key_ptr, ca_pair_idx, ca_pair_row, b.pw_ca,
                                       b.dq2 + po, b.dK, b.dV, 0, S, d->p_ca_attn, site_seed(d->seed, l, 2), 0, qs, side[1].st));
        if (nk > 0) {
            // dk / dv of the dense block (one kernel), added to the rows dn_keys of dK / dV
            TD_RC(mv2d_dense_attn_bwd_parts(a.q_ca, a.kd, a.vd, a.ctx_ca, b.dctx2, a.lse, pad, nk, d->p_ca_attn, site_seed(d->seed, l, 6), qs, b.dq2, b.dkd, b.dvd,
                                            b.dws, 2, side[1].st));
            hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(cdiv(nk, 4)), dim3(256), 0, side[1].st, b.dK, dn_keys, (const float*)b.dkd, nk);
            hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(cdiv(nk, 4)), dim3(256), 0, side[1].st, b.dV, dn_keys, (const float*)b.dvd, nk);
        }
        after(pl, side[1].st, side[2].st);
        TD_RC(side[1].dgrad(b.dK, P[CA_W] + (long long)C * C, d_key_in, S, C, C, first_kv ? 0 : 1));
        TD_RC(side[1].wgrad(b.dK, key_in, G[CA_W] + (long long)C * C, G[CA_B] + C, S, C, C));
        TD_RC(side[2].dgrad(b.dV, P[CA_W] + 2LL * C * C, d_val_in, S, C, C, first_kv ? 0 : 1));
        TD_RC(side[2].wgrad(b.dV, val_in, G[CA_W] + 2LL * C * C, G[CA_B] + 2 * C, S, C, C));
        first_kv = false;
        TD_RC(mainl.dgrad(b.dq2, P[CA_W], b.Q1, T, C, C, 0));
        // ---- norm 0 and the self attention
        const Drop d1 = mk_drop(d->p_sa_out, site_seed(d->seed, l, 1));
        ln_bwd(a.s1, b.ds2, b.Q1, nullptr, P[N0_W], b.ds1, b.do1, d1, b.part[4], b.part[5]);
        const float* do1 = d1.thr ? b.do1 : b.ds1;
        TD_RC(mainl.dgrad(do1, P[SA_OW], b.dctx1, T, C, C, 0));
        TD_RC(mv2d_sparse_xattn_bwd_ex(a.q_sa, a.k_sa, a.v_sa, sa_row_ptr, sa_col, a.ctx_sa, b.dctx1, sa_key_ptr, sa_pair_idx, sa_pair_row, b.pw_sa,
                                       b.dq1, b.dk1, b.dv1, T, T, d->p_sa_attn, site_seed(d->seed, l, 0), d->sa_nnz >= 64LL * T ? 1 : 0, qs, st));
        // d(x + query_pos) of the q / k inputs; d x of the value input joins the post_norm gradient of the layer below
        TD_RC(mainl.dgrad(b.dq1, P[SA_W], b.A, T, C, C, 0));
        TD_RC(mainl.dgrad(b.dk1, P[SA_W] + (long long)C * C, b.A, T, C, C, 1));
        if (l > 0) {
            TD_RC(mainl.dgrad(b.dv1, P[SA_W] + 2LL * C * C, B[l - 1].dxa, T, C, C, 1));
            g0 = b.ds1; g1 = b.A; g2 = B[l - 1].dxa;
        }
        // ================= this layer's parameter gradients (side stream 0): nothing on the chain waits for them
        after(pl, st, side[0].st);
        side[0].ln_params(b.part[0], b.part[1], nb, G[N2_W], G[N2_B]);
        side[0].ln_params(b.part[2], b.part[3], nb, G[N1_W], G[N1_B]);
        side[0].ln_params(b.part[4], b.part[5], nb, G[N0_W], G[N0_B]);
        TD_RC(side[0].wgrad(dyf, a.h, G[F2_W], G[F2_B], T, C, F));
        TD_RC(side[0].wgrad(b.dh, a.x2, G[F1_W], G[F1_B], T, F, C));
        TD_RC(side[0].wgrad(do2, a.ctx_ca, G[CA_OW], G[CA_OB], T, C, C));
        TD_RC(side[0].wgrad(b.dq2, a.xq1, G[CA_W], G[CA_B], T, C, C));
        TD_RC(side[0].wgrad(do1, a.ctx_sa, G[SA_OW], G[SA_OB], T, C, C));
        TD_RC(side[0].wgrad(b.dq1, a.xq, G[SA_W], G[SA_B], T, C, C));
        TD_RC(side[0].wgrad(b.dk1, a.xq, G[SA_W] + (long long)C * C, G[SA_B] + C, T, C, C));
        TD_RC(side[0].wgrad(b.dv1, x_in, G[SA_W] + 2LL * C * C, G[SA_B] + 2 * C, T, C, C));
    }
    // query_pos entered every layer twice (x + query_pos before each attention)
    SumArgs sa;
    sa.n = 0;
    for (int l = 0; l < L; ++l) { sa.src[sa.n++] = B[l].A; sa.src[sa.n++] = B[l].Q1; }
    hipLaunchKernelGGL(sum_n_kernel, dim3(blocks4(TC)), dim3(256), 0, st, sa, d_qpos, TC / 4);
    MV2D_LAUNCH_CHECK();
    for (int i = 0; i < NSIDE; ++i) after(pl, side[i].st, st);
    side_join.armed = false;
    return MV2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The classification / regression branches of every intermediate output (RH/bbox_heads/cross_attention_head.py:118-142,200-218):
//   cls_l = Linear(ReLU(LN(Linear(ReLU(LN(Linear(out_l)))))))      reg_l = Linear(ReLU(Linear(ReLU(Linear(out_l)))))      (raw box code)
// The L layers are independent: layer l runs on stream l mod 4 (the caller's stream and the three side streams).
// ---------------------------------------------------------------------------------------------------------------------------------------
struct mv2d_th_dims { int T, L, NC; float eps; };

namespace {

constexpr int NPH = 16, NREG = 10;
enum { C0_W, C0_B, C1_W, C1_B, C3_W, C3_B, C4_W, C4_B, C6_W, C6_B, R0_W, R0_B, R2_W, R2_B, R4_W, R4_B };

struct HAct { float *y0, *y1, *y3, *y4, *t0, *t2; };
struct HActLayout {
    HAct a[8]; long long bytes;
    HActLayout(const mv2d_th_dims& d, void* base) {
        Carver c(base);
        for (int l = 0; l < d.L; ++l) {
            float** f[] = {&a[l].y0, &a[l].y1, &a[l].y3, &a[l].y4, &a[l].t0, &a[l].t2};
            for (float** q : f) *q = c.take<float>((long long)d.T * C);
        }
        bytes = c.off;
    }
};
static bool th_ok(const mv2d_th_dims* d) { return d && d->T > 0 && d->L >= 1 && d->L <= 8 && d->NC >= 1 && d->NC <= 64; }
static long long th_gemm_ws(const mv2d_th_dims& d) {
    const int shapes[][3] = {{d.T, C, C}, {C, C, d.T}, {d.NC, C, d.T}, {NREG, C, d.T}};
    long long m = 0;
    for (auto& s : shapes) { const long long b = mv2d_gemm_f32x3_ws_bytes(s[0], s[1], s[2]); if (b > m) m = b; }
    return al256(m) + 256;
}
static long long th_cs(const mv2d_th_dims& d) { return (long long)(mv2d_colsum_scratch_rows(d.T) + 1) * C; }

}  // namespace

extern "C" long long mv2d_train_heads_act_bytes(const mv2d_th_dims* d) { return th_ok(d) ? HActLayout(*d, nullptr).bytes + 256 : -1; }

extern "C" long long mv2d_train_heads_ws_bytes(const mv2d_th_dims* d, int backward) {
    if (!th_ok(d)) return -1;
    const long long lanes = (1 + NSIDE) * (th_gemm_ws(*d) + al256(th_cs(*d) * 4));
    if (!backward) return lanes + 4096;
    const int nb = cdiv(d->T, LNB_ROWS);
    return lanes + d->L * (8 * al256((long long)d->T * C * 4) + 4 * al256((long long)nb * C * 4)) + 4096;
}

// params: 16 L device pointers -- per layer cls_branches.{0,1,3,4,6}.{weight,bias} then reg_branches.{0,2,4}.{weight,bias};
// outs [L,T,256] -> cls [L,T,NC], reg [L,T,10] (the raw code: the reference point / range arithmetic stays with the caller).
extern "C" int mv2d_train_heads_fwd(const mv2d_th_dims* d, const float* const* params, const float* outs, float* cls, float* reg, void* act, void* ws,
                                    void* stream) {
    MV2D_CHECK_ARG(th_ok(d) && params && outs && cls && reg && act && ws, "mv2d_train_heads_fwd: bad args");
    MV2D_CHECK_ARG((((uintptr_t)act | (uintptr_t)ws) & 255) == 0, "mv2d_train_heads_fwd: act / ws must be 256-byte aligned");
    Pool* pl = pool();
    MV2D_CHECK_ARG(pl != nullptr, "mv2d_train_heads_fwd: could not create the side streams");
    SideJoin side_join{pl, (hipStream_t)stream};
    const int T = d->T, L = d->L, NC = d->NC;
    hipStream_t st = (hipStream_t)stream;
    HActLayout al(*d, act);
    Carver cw(ws);
    const long long gws = th_gemm_ws(*d), csb = th_cs(*d);
    Lane lane[1 + NSIDE];
    lane[0] = Lane{st, cw.take<char>(gws), gws, cw.take<float>(csb)};
    for (int i = 0; i < NSIDE; ++i) { lane[1 + i] = Lane{td_serial() ? st : pl->side[i], cw.take<char>(gws), gws, cw.take<float>(csb)}; after(pl, st, lane[1 + i].st); }
    for (int l = 0; l < L; ++l) {
        const Lane& ln = lane[l % (1 + NSIDE)];
        const float* const* P = params + l * NPH;
        const HAct& a = al.a[l];
        const float* x = outs + (long long)l * T * C;
        TD_RC(ln.linear(x, P[C0_W], P[C0_B], 0, 1.f, 0, a.y0, T, C, C));
        hipLaunchKernelGGL(ln_relu_kernel, dim3(cdiv(T, 4)), dim3(256), 0, ln.st, a.y0, P[C1_W], P[C1_B], a.y1, T, d->eps);
        TD_RC(ln.linear(a.y1, P[C3_W], P[C3_B], 0, 1.f, 0, a.y3, T, C, C));
        hipLaunchKernelGGL(ln_relu_kernel, dim3(cdiv(T, 4)), dim3(256), 0, ln.st, a.y3, P[C4_W], P[C4_B], a.y4, T, d->eps);
        TD_RC(ln.linear(a.y4, P[C6_W], P[C6_B], 0, 1.f, 0, cls + (long long)l * T * NC, T, NC, C));
        TD_RC(ln.linear(x, P[R0_W], P[R0_B], 1, 1.f, 0, a.t0, T, C, C));
        TD_RC(ln.linear(a.t0, P[R2_W], P[R2_B], 1, 1.f, 0, a.t2, T, C, C));
        TD_RC(ln.linear(a.t2, P[R4_W], P[R4_B], 0, 1.f, 0, reg + (long long)l * T * NREG, T, NREG, C));
    }
    MV2D_LAUNCH_CHECK();
    for (int i = 0; i < NSIDE; ++i) after(pl, lane[1 + i].st, st);
    side_join.armed = false;
    return MV2D_OK;
}

// grads: 16 L device pointers (written); d_cls [L,T,NC], d_reg [L,T,10] -> d_outs [L,T,256] (written).
extern "C" int mv2d_train_heads_bwd(const mv2d_th_dims* d, const float* const* params, float* const* grads, const float* outs, const float* d_cls,
                                    const float* d_reg, const void* act, void* ws, float* d_outs, void* stream) {
    MV2D_CHECK_ARG(th_ok(d) && params && grads && outs && d_cls && d_reg && act && ws && d_outs, "mv2d_train_heads_bwd: bad args");
    MV2D_CHECK_ARG((((uintptr_t)act | (uintptr_t)ws) & 255) == 0, "mv2d_train_heads_bwd: act / ws must be 256-byte aligned");
    Pool* pl = pool();
    MV2D_CHECK_ARG(pl != nullptr, "mv2d_train_heads_bwd: could not create the side streams");
    SideJoin side_join{pl, (hipStream_t)stream};
    const int T = d->T, L = d->L, NC = d->NC;
    const long long TC = (long long)T * C;
    hipStream_t st = (hipStream_t)stream;
    HActLayout al(*d, const_cast<void*>(act));
    Carver cw(ws);
    const long long gws = th_gemm_ws(*d), csb = th_cs(*d);
    Lane lane[1 + NSIDE];
    lane[0] = Lane{st, cw.take<char>(gws), gws, cw.take<float>(csb)};
    for (int i = 0; i < NSIDE; ++i) { lane[1 + i] = Lane{td_serial() ? st : pl->side[i], cw.take<char>(gws), gws, cw.take<float>(csb)}; after(pl, st, lane[1 + i].st); }
    const int nb = cdiv(T, LNB_ROWS);
    for (int l = 0; l < L; ++l) {
        const Lane& ln = lane[l % (1 + NSIDE)];
        const float* const* P = params + l * NPH;
        float* const* G = grads + l * NPH;
        const HAct& a = al.a[l];
        const float* x = outs + (long long)l * TC;
        float* dx = d_outs + (long long)l * TC;
        float *dy4 = cw.take<float>(TC), *dy3 = cw.take<float>(TC), *dy1 = cw.take<float>(TC), *dy0 = cw.take<float>(TC), *dt2 = cw.take<float>(TC),
              *dt0 = cw.take<float>(TC);
        (void)cw.take<float>(2 * TC);
        float* part[4];
        for (float*& q : part) q = cw.take<float>((long long)nb * C);
        // class branch
        const float* g6 = d_cls + (long long)l * T * NC;
        TD_RC(ln.wgrad(g6, a.y4, G[C6_W], G[C6_B], T, NC, C));
        TD_RC(ln.dgrad(g6, P[C6_W], dy4, T, NC, C, 0));
        {
            LnBwdArgs b{a.y3, dy4, nullptr, nullptr, P[C4_W], dy3, nullptr, part[0], part[1], T, d->eps, mk_drop(0.f, 0u), a.y4};
            hipLaunchKernelGGL(ln_bwd_ex_kernel, dim3(nb), dim3(256), 0, ln.st, b);
            ln.ln_params(part[0], part[1], nb, G[C4_W], G[C4_B]);
        }
        TD_RC(ln.wgrad(dy3, a.y1, G[C3_W], G[C3_B], T, C, C));
        TD_RC(ln.dgrad(dy3, P[C3_W], dy1, T, C, C, 0));
        {
            LnBwdArgs b{a.y0, dy1, nullptr, nullptr, P[C1_W], dy0, nullptr, part[2], part[3], T, d->eps, mk_drop(0.f, 0u), a.y1};
            hipLaunchKernelGGL(ln_bwd_ex_kernel, dim3(nb), dim3(256), 0, ln.st, b);
            ln.ln_params(part[2], part[3], nb, G[C1_W], G[C1_B]);
        }
        TD_RC(ln.wgrad(dy0, x, G[C0_W], G[C0_B], T, C, C));
        TD_RC(ln.dgrad(dy0, P[C0_W], dx, T, C, C, 0));
        // regression branch
        const float* g4 = d_reg + (long long)l * T * NREG;
        TD_RC(ln.wgrad(g4, a.t2, G[R4_W], G[R4_B], T, NREG, C));
        TD_RC(mv2d_dgrad_relu_f32x3(g4, P[R4_W], a.t2, 1.f, dt2, T, NREG, C, ln.st));
        TD_RC(ln.wgrad(dt2, a.t0, G[R2_W], G[R2_B], T, C, C));
        TD_RC(mv2d_dgrad_relu_f32x3(dt2, P[R2_W], a.t0, 1.f, dt0, T, C, C, ln.st));
        TD_RC(ln.wgrad(dt0, x, G[R0_W], G[R0_B], T, C, C));
        TD_RC(ln.dgrad(dt0, P[R0_W], dx, T, C, C, 1));
    }
    MV2D_LAUNCH_CHECK();
    for (int i = 0; i < NSIDE; ++i) after(pl, lane[1 + i].st, st);
    side_join.armed = false;
    return MV2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Box code of every intermediate output (cross_attention_head.py:216-238; RH/mv2d_t_head.py:136-140), forward and backward:
//   out[0,1,4] = sigmoid(t[0,1,4] + isig(ref)[0,1,2]) * range + low,  isig(x) = log(max(clamp(x, 0, 1), 1e-5) / max(1 - clamp(x, 0, 1), 1e-5));
//   out[8,9] = t[8,9] / dt for the rows >= pad when dt != 0 (the denoising rows keep theirs); the other entries pass through.
// The backward also returns the gradient w.r.t. the reference points (summed over the layers in fixed order): the query generator trains
// through them.
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {

struct BoxRange { float lo[3], span[3]; };

__device__ __forceinline__ float isig_f(float x) {
    const float r = fminf(fmaxf(x, 0.f), 1.f);
    return logf(fmaxf(r, 1e-5f) / fmaxf(1.f - r, 1e-5f));
}

__global__ __launch_bounds__(256) void box_code_fwd_kernel(const float* __restrict__ t, const float* __restrict__ ref, float* __restrict__ out, int L,
                                                           int T, int pad, float dt, BoxRange rg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L * T) return;
    const int r = i % T;
    float v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = t[(long long)i * 10 + k];
    const int col[3] = {0, 1, 4};
#pragma unroll
    for (int k = 0; k < 3; ++k) v[col[k]] = 1.f / (1.f + expf(-(v[col[k]] + isig_f(ref[r * 3 + k])))) * rg.span[k] + rg.lo[k];
    if (dt != 0.f && r >= pad) { v[8] = v[8] / dt; v[9] = v[9] / dt; }
#pragma unroll
    for (int k = 0; k < 10; ++k) out[(long long)i * 10 + k] = v[k];
}

// d t (every layer) and, per row, d ref = sum over the layers of d z * d isig / d ref
__global__ __launch_bounds__(256) void box_code_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out, const float* __restrict__ ref,
                                                           float* __restrict__ dt_out, float* __restrict__ dref, int L, int T, int pad, float dt,
                                                           BoxRange rg) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= T) return;
    float acc[3] = {0.f, 0.f, 0.f};
    const int col[3] = {0, 1, 4};
    for (int l = 0; l < L; ++l) {
        const long long o = ((long long)l * T + r) * 10;
        float v[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) v[k] = g[o + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s = (out[o + col[k]] - rg.lo[k]) / rg.span[k];           // the sigmoid of the forward
            v[col[k]] = v[col[k]] * rg.span[k] * s * (1.f - s);
            acc[k] += v[col[k]];
        }
        if (dt != 0.f && r >= pad) { v[8] = v[8] / dt; v[9] = v[9] / dt; }
#pragma unroll
        for (int k = 0; k < 10; ++k) dt_out[o + k] = v[k];
    }
    if (dref) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = ref[r * 3 + k];
            float d = 0.f;
            if (x > 0.f && x < 1.f) {                                           // clamp(x, 0, 1) passes the gradient inside the interval only
                if (x > 1e-5f) d += 1.f / x;                                    // log(max(x, 1e-5))
                if (1.f - x > 1e-5f) d += 1.f / (1.f - x);                      // -log(max(1 - x, 1e-5))
            }
            dref[r * 3 + k] = acc[k] * d;
        }
    }
}

}  // namespace

extern "C" int mv2d_box_code_fwd(const float* t, const float* ref, float* out, int L, int T, int pad, float dt, const float* pc_range, void* stream) {
    MV2D_CHECK_ARG(t && ref && out && pc_range && L > 0 && T >= 0, "mv2d_box_code_fwd: bad args");
    if (T == 0) return MV2D_OK;
    BoxRange rg;
    for (int k = 0; k < 3; ++k) { rg.lo[k] = pc_range[k]; rg.span[k] = pc_range[3 + k] - pc_range[k]; }
    hipLaunchKernelGGL(box_code_fwd_kernel, dim3(cdiv(L * T, 256)), dim3(256), 0, (hipStream_t)stream, t, ref, out, L, T, pad, dt, rg);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// g: gradient of the boxes [L,T,10], out: the boxes of the forward -> d_t [L,T,10], d_ref [T,3] (NULL: not wanted)
extern "C" int mv2d_box_code_bwd(const float* g, const float* out, const float* ref, float* d_t, float* d_ref, int L, int T, int pad, float dt,
                                 const float* pc_range, void* stream) {
    MV2D_CHECK_ARG(g && out && ref && d_t && pc_range && L > 0 && T >= 0, "mv2d_box_code_bwd: bad args");
    if (T == 0) return MV2D_OK;
    BoxRange rg;
    for (int k = 0; k < 3; ++k) { rg.lo[k] = pc_range[k]; rg.span[k] = pc_range[3 + k] - pc_range[k]; }
    hipLaunchKernelGGL(box_code_bwd_kernel, dim3(cdiv(T, 256)), dim3(256), 0, (hipStream_t)stream, g, out, ref, d_t, d_ref, L, T, pad, dt, rg);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
