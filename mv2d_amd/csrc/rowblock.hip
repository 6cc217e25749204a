// Row-block fused kernels of the MV2D decoder (gfx950, exact fp32 MFMA): one 4-wave block owns 16 query rows and
// chains several 256x256 linears / LayerNorms / residual adds on an activation tile that never leaves LDS.
//
//   mv2d_attn_out_fused  : ctx -> out_proj + bias + residual -> LayerNorm -> x_out  [-> (+query_pos) -> q in_proj * scale -> q]
//                          (mmcv BaseTransformerLayer steps 'self_attn' tail + 'norm' + 'cross_attn' head, and 'cross_attn' tail + 'norm';
//                           MU/petr_transformer.py:358-370, 487-513, norms of :269-311)
//   mv2d_heads_fused     : per decoder layer, cls branch (Linear-LN-ReLU x2 + Linear) and reg branch (Linear-ReLU x2 + Linear) + the
//                          reference-point / sigmoid / pc_range tail (RH/bbox_heads/cross_attention_head.py:216-238) and velocity / dt
//                          (RH/mv2d_t_head.py:136-140)
//
// Why: the round-1 profile showed ~5 us of fixed cost per dependent kernel; these chains were 3-9 launches of 7 us each.
// A 256x256 fp32 linear on one CU costs ~3.4 us of MFMA time (256 FLOP/clk/CU), so chains of up to ~3 linears fit.
// Weights stream from L2 as MFMA fragments (16 float4 per lane per 16-column tile, next tile in flight behind the current one).
#include "common.h"

namespace {

constexpr int C = 256;

__device__ __forceinline__ int toff(int row, int col) { return row * C + ((((col >> 2) ^ (row & 15))) << 2) + (col & 3); }

struct Frag { float4 v[16]; };

__device__ __forceinline__ void load_w(Frag& f, const float* __restrict__ W, int ldw, int nrow, int nmax, int fg) {
    const float* wp = W + (long long)min(nrow, nmax - 1) * ldw + 4 * fg;
#pragma unroll
    for (int c = 0; c < 16; ++c) f.v[c] = *reinterpret_cast<const float4*>(wp + 16 * c);
}

// the same fragment from a FRAGMENT-MAJOR weight copy (mv2d_pack_wfrag_f32: [16-row tile][k chunk c][lane][4]): one contiguous
// 1 KB per load instead of 16 rows x 64 B
__device__ __forceinline__ void load_w_frag(Frag& f, const float* __restrict__ Wp, int tile, int lane) {
    const float* wp = Wp + ((long long)tile * 16 * 64 + lane) * 4;
#pragma unroll
    for (int c = 0; c < 16; ++c) f.v[c] = *reinterpret_cast<const float4*>(wp + c * 256);
}

__device__ __forceinline__ f32x4_t tile_mma(const float* __restrict__ As, const Frag& f, int fr, int fg) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(As + fr * C + (((4 * c + fg) ^ fr) << 2));
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f.v[c].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f.v[c].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f.v[c].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f.v[c].w, acc, 0, 0, 0);
    }
    return acc;
}

// Out[16,256] = As[16,256] . W[256,256]^T: wave w computes columns 64w .. 64w+63 (4 tiles), results returned in registers.
// All 64 weight-fragment loads of the wave (4 tiles x 16 float4 = 256 VGPRs; the kernels run one wave per SIMD) are issued before
// the first MFMA: the weights come from the Infinity Cache (~2.5 us away) and must be waited for once, not once per tile.
template <bool FRAG = false>
__device__ __forceinline__ void linear256(const float* __restrict__ As, const float* __restrict__ W, int wave, int fr, int fg, f32x4_t acc[4]) {
    Frag f0, f1, f2, f3;
    if (FRAG) {
        const int lane = fr + 16 * fg;
        load_w_frag(f0, W, wave * 4, lane);
        load_w_frag(f1, W, wave * 4 + 1, lane);
        load_w_frag(f2, W, wave * 4 + 2, lane);
        load_w_frag(f3, W, wave * 4 + 3, lane);
    } else {
        load_w(f0, W, C, wave * 64 + fr, C, fg);
        load_w(f1, W, C, wave * 64 + 16 + fr, C, fg);
        load_w(f2, W, C, wave * 64 + 32 + fr, C, fg);
        load_w(f3, W, C, wave * 64 + 48 + fr, C, fg);
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the compiler from sinking the loads back next to their MFMAs
    acc[0] = tile_mma(As, f0, fr, fg);
    acc[1] = tile_mma(As, f1, fr, fg);
    acc[2] = tile_mma(As, f2, fr, fg);
    acc[3] = tile_mma(As, f3, fr, fg);
}

// write the wave's 4 tiles (+bias, optional relu, optional scale) into an LDS tile
__device__ __forceinline__ void store_tile(float* __restrict__ Ds, const f32x4_t acc[4], const float* __restrict__ bias, int wave, int fr, int fg,
                                           bool relu, float scale) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = wave * 64 + 16 * t + fr;
        const float b = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = (acc[t][r] + b) * scale;
            if (relu) v = relu_f(v);
            Ds[toff(4 * fg + r, col)] = v;
        }
    }
}

// load 16 rows x 256 floats (rows m0.., clamped) into an LDS tile; optional elementwise add of a second global tile
__device__ __forceinline__ void load_rows(float* __restrict__ Ds, const float* __restrict__ G, const float* __restrict__ G2, int m0, int M, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i, row = idx >> 6, slot = idx & 63;
        const long long g = (long long)min(m0 + row, M - 1) * C + slot * 4;
        float4 v = *reinterpret_cast<const float4*>(G + g);
        if (G2) { const float4 u = *reinterpret_cast<const float4*>(G2 + g); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        *reinterpret_cast<float4*>(Ds + row * C + ((slot ^ (row & 15)) << 2)) = v;
    }
}

// in-place LayerNorm (+ optional residual rows added first, + optional relu) of a 16x256 LDS tile; wave w owns rows 4w..4w+3.
// Optionally writes the result rows to global (out) and a second tile / global with `addvec` rows added.
__device__ __forceinline__ void ln_tile(float* __restrict__ Ds, const float* __restrict__ resid, const float* __restrict__ lw, const float* __restrict__ lb,
                                        bool relu, float* __restrict__ out, const float* __restrict__ addvec, float* __restrict__ Ds_plus,
                                        int m0, int M, int wave, int lane, float eps) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr, m = m0 + row;
        float* p = Ds + row * C + ((lane ^ (row & 15)) << 2);
        float4 v = *reinterpret_cast<const float4*>(p);
        const long long g = (long long)min(m, M - 1) * C + lane * 4;
        if (resid) { const float4 u = *reinterpret_cast<const float4*>(resid + g); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / C);
        const float rstd = 1.0f / sqrtf(var + eps);
        const float4 ww = *reinterpret_cast<const float4*>(lw + lane * 4), bb = *reinterpret_cast<const float4*>(lb + lane * 4);
        v = make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
        if (relu) { v.x = relu_f(v.x); v.y = relu_f(v.y); v.z = relu_f(v.z); v.w = relu_f(v.w); }
        *reinterpret_cast<float4*>(p) = v;
        if (out && m < M) *reinterpret_cast<float4*>(out + g) = v;
        if (Ds_plus) {
            const float4 u = *reinterpret_cast<const float4*>(addvec + g);
            *reinterpret_cast<float4*>(Ds_plus + row * C + ((lane ^ (row & 15)) << 2)) = make_float4(v.x + u.x, v.y + u.y, v.z + u.z, v.w + u.w);
        }
    }
}

struct AttnOutParams {
    const float* ctx; const float* resid; const float* Wo; const float* bo; const float* lw; const float* lb;
    float* x_out;
    const float* qpos; const float* Wq; const float* bq; float qscale; float* q_out;     // optional second stage (Wq null -> skipped)
    int M; float eps;
};

// 16 waves per block: wave w owns the 16-column tile w of BOTH linears (one 16x16x256 tile each = exactly the work a wave of the
// N-parallel GEMM kernel does, so the chain is not longer than two separate launches), then row w of the LayerNorm.
// (The first version of this kernel gave a wave four tiles per linear: 19 blocks x 4 waves chained 8 tiles each and lost to three
//  separate launches.)  VGPR budget at 4 waves per SIMD is 128: the activation tile is staged through LDS by one 16-byte load
// per thread, only the weight fragment (64 VGPRs) is held in registers; the Wq fragment is fetched into the same registers
// behind the first tile's MFMAs and arrives under the LayerNorm.
__global__ __launch_bounds__(1024) void attn_out_fused_kernel(AttnOutParams p) {
    __shared__ __attribute__((aligned(16))) float ta[16 * C];
    __shared__ __attribute__((aligned(16))) float tb[16 * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * 16;
    // activation tile: thread -> (row = wave, 16-byte slot = lane), weight fragment of column tile `wave`
    const long long grow = (long long)min(m0 + wave, p.M - 1) * C + lane * 4;
    const float4 av = *reinterpret_cast<const float4*>(p.ctx + grow);
    Frag f;
    load_w(f, p.Wo, C, wave * 16 + fr, C, fg);
    *reinterpret_cast<float4*>(ta + wave * C + ((lane ^ (wave & 15)) << 2)) = av;
    __syncthreads();
    f32x4_t acc = tile_mma(ta, f, fr, fg);
    if (p.Wq) load_w(f, p.Wq, C, wave * 16 + fr, C, fg);        // in flight during the LayerNorm
    {
        const int col = wave * 16 + fr;
        const float b = p.bo[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[toff(4 * fg + r, col)] = acc[r] + b;
    }
    __syncthreads();
    {   // row `wave`: residual + LayerNorm exactly like row_ln_kernel (lane = 4 consecutive columns, wavefront reductions)
        const int row = wave, m = m0 + row;
        float4 v = *reinterpret_cast<const float4*>(tb + row * C + ((lane ^ (row & 15)) << 2));
        const float4 u = *reinterpret_cast<const float4*>(p.resid + grow);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / C);
        const float rstd = 1.0f / sqrtf(var + p.eps);
        const float4 ww = *reinterpret_cast<const float4*>(p.lw + lane * 4), bb = *reinterpret_cast<const float4*>(p.lb + lane * 4);
        v = make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
        if (m < p.M) *reinterpret_cast<float4*>(p.x_out + grow) = v;
        if (p.Wq) {
            const float4 qp = *reinterpret_cast<const float4*>(p.qpos + grow);
            *reinterpret_cast<float4*>(ta + row * C + ((lane ^ (row & 15)) << 2)) = make_float4(v.x + qp.x, v.y + qp.y, v.z + qp.z, v.w + qp.w);
        }
    }
    if (!p.Wq) return;
    __syncthreads();
    acc = tile_mma(ta, f, fr, fg);
    const int col = wave * 16 + fr;
    const float b = p.bq[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * fg + r;
        if (m < p.M) p.q_out[(long long)m * C + col] = (acc[r] + b) * p.qscale;
    }
}

// The same fused chain in split precision ("bf16x3", cf. ffn_x3.hip): both weights come as bf16 hi/lo pairs in
// fragment-major order (mv2d_split_bf16x2 + mv2d_pack_wfrag_bf16), the activation tile is split when it is staged into LDS.
// 24 v_mfma_f32_16x16x32_bf16 per tile instead of 64 v_mfma_f32_16x16x4_f32 (408 vs 2048 matrix-pipe cycles per wave, and a
// block's 16 waves share one CU), ~1e-5 relative error.
struct AttnOutX3Params {
    const float* ctx; const float* resid; const unsigned short* Woh; const unsigned short* Wol; const float* bo; const float* lw; const float* lb;
    float* x_out;
    const float* qpos; const unsigned short* Wqh; const unsigned short* Wql; const float* bq; float qscale; float* q_out;
    int M; float eps;
    // ZMAP (cross attention, csrc/xattn_tile.hip): the context tile is computed here from the tile attention's z [M,8,256]:
    //   ctx[:, 32 h + d] = Wv_h z_h + bv (packed WB, see xattn_ctxmap_kernel); rows without a key (row_ptr) -> NaN / 0
    const float* z; const uint4* WBh; const uint4* WBl; const float* bv; const int* row_ptr; int empty_nan;
    // QMAP: the query tile goes on into the per-head query maps Qt (packed WA, see xattn_qmap_kernel) instead of q_out
    const uint4* WAh; const uint4* WAl; uint4* Qt;
};

typedef q16x8_t mfma_bf16x8;      // the query side's 16-bit split format (common.h "q16": fp16 pairs since round 5)
union BFrag { uint4 u; mfma_bf16x8 v; };

__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
    split_q16x4(v, hi, lo);
}

// one 16x16 tile: sum over 8 k-steps of a_hi.w_hi + a_lo.w_hi + a_hi.w_lo; activation rows from the bf16 LDS images (512 B rows,
// 16-byte chunk c of row r at c ^ r), weight fragments (hi, lo) already in registers
__device__ __forceinline__ f32x4_t tile_mma_x3(const unsigned char* __restrict__ ah, const unsigned char* __restrict__ al, const BFrag wh[8],
                                               const BFrag wl[8], int fr, int fg) {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        BFrag xh, xl;
        const int off = fr * 512 + (((4 * s + fg) ^ fr) << 4);
        xh.u = *reinterpret_cast<const uint4*>(ah + off);
        xl.u = *reinterpret_cast<const uint4*>(al + off);
        a0 = mfma_q16_16x16x32(xh.v, wh[s].v, a0, 0, 0, 0);
        a1 = mfma_q16_16x16x32(xl.v, wh[s].v, a1, 0, 0, 0);
        a1 = mfma_q16_16x16x32(xh.v, wl[s].v, a1, 0, 0, 0);
    }
    return f32x4_t{a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]};
}

__device__ __forceinline__ void load_w_x3(BFrag wh[8], BFrag wl[8], const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                          int tile, int lane) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {                 // fragment-major [k-step][16 column tiles][lane][8]
        const long long o = (((long long)s * 16 + tile) * 64 + lane) * 8;
        wh[s].u = *reinterpret_cast<const uint4*>(Wh + o);
        wl[s].u = *reinterpret_cast<const uint4*>(Wl + o);
    }
}

// RT row tiles (16 rows each) per block share one fetch of the weight fragments.  A block is bound by fetching its 0.5 MB of weights
// through one CU's L1 (~57 GB/s), whatever the number of rows: with a batch of samples (M > 512) two row tiles per block take the
// same time on half as many CUs, which the other streams' wide kernels can use.
__device__ __forceinline__ void split8(const float4& x0, const float4& x1, BFrag& hi, BFrag& lo) {
    uint2 h0, l0, h1, l1;
    split4(x0, h0, l0);
    split4(x1, h1, l1);
    hi.u = make_uint4(h0.x, h0.y, h1.x, h1.y);
    lo.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

template <int RT, bool ZMAP, bool QMAP>
__global__ __launch_bounds__(1024) void attn_out_fused_x3_kernel(AttnOutX3Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char ah[RT * 16 * 512], al[RT * 16 * 512];
    __shared__ __attribute__((aligned(16))) float tb[RT * 16 * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int mb = blockIdx.x * (16 * RT);
    long long grow[RT];
    float4 av[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        grow[t] = (long long)min(mb + 16 * t + wave, p.M - 1) * C + lane * 4;
        if (!ZMAP) av[t] = *reinterpret_cast<const float4*>(p.ctx + grow[t]);
    }
    BFrag wh[8], wl[8];
    if (ZMAP) {
        // context map of the tile cross attention: wave w = (head w >> 1, column half w & 1) computes the 16 x 16 block of ctx it would
        // otherwise have loaded as rows, in xattn_ctxmap_kernel's arithmetic (same fragments, same MFMA order), and writes it into the images
        const int h = wave >> 1, nt = wave & 1;
        const uint4* wbh = p.WBh + (long long)h * 16 * 64 + nt * 64 + lane;
        const uint4* wbl = p.WBl + (long long)h * 16 * 64 + nt * 64 + lane;
        const float bias = p.bv[16 * wave + fr];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const float* zp = p.z + ((long long)min(mb + 16 * t + fr, p.M - 1) * 8 + h) * C + 8 * fg;
            f32x4_t ca = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                BFrag zh, zl, bh, bl;
                split8(*reinterpret_cast<const float4*>(zp + 32 * s), *reinterpret_cast<const float4*>(zp + 32 * s + 4), zh, zl);
                bh.u = wbh[s * 128];
                bl.u = wbl[s * 128];
                ca = mfma_q16_16x16x32(zl.v, bh.v, ca, 0, 0, 0);
                ca = mfma_q16_16x16x32(zh.v, bl.v, ca, 0, 0, 0);
                ca = mfma_q16_16x16x32(zh.v, bh.v, ca, 0, 0, 0);
            }
            const int col = 16 * wave + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * fg + r, m = min(mb + 16 * t + row, p.M - 1);
                float v = ca[r] + bias;
                if (p.row_ptr[m + 1] <= p.row_ptr[m]) v = p.empty_nan ? __uint_as_float(0x7fc00000u) : 0.f;
                unsigned short hb, lb;
                split_q16(v, hb, lb);
                const int off = t * 8192 + row * 512 + (((col >> 3) ^ row) << 4) + (col & 7) * 2;
                *reinterpret_cast<unsigned short*>(ah + off) = hb;
                *reinterpret_cast<unsigned short*>(al + off) = lb;
            }
        }
        load_w_x3(wh, wl, p.Woh, p.Wol, wave, lane);
    } else {
        load_w_x3(wh, wl, p.Woh, p.Wol, wave, lane);
    }
    const int aoff = wave * 512 + (((lane >> 1) ^ wave) << 4) + (lane & 1) * 8;     // this thread's 4 values in the bf16 images of a tile
    if (!ZMAP) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            uint2 hi, lo;
            split4(av[t], hi, lo);
            *reinterpret_cast<uint2*>(ah + t * 8192 + aoff) = hi;
            *reinterpret_cast<uint2*>(al + t * 8192 + aoff) = lo;
        }
    }
    // everything the epilogue reads from global memory is requested BEFORE the barrier and the MFMAs (each of these was a dependent round
    // trip of its own behind them: bias, residual rows; the LayerNorm parameters do not fit the 128-register budget of 16 waves per block)
    float4 ures[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) ures[t] = *reinterpret_cast<const float4*>(p.resid + grow[t]);
    const float bo_c = p.bo[wave * 16 + fr];
    __syncthreads();
    f32x4_t acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = tile_mma_x3(ah + t * 8192, al + t * 8192, wh, wl, fr, fg);
    if (p.Wqh) load_w_x3(wh, wl, p.Wqh, p.Wql, wave, lane);        // in flight during the LayerNorm
    {
        const int col = wave * 16 + fr;
        const float b = bo_c;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[t * 16 * C + toff(4 * fg + r, col)] = acc[t][r] + b;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = wave, m = mb + 16 * t + row;
        float4 v = *reinterpret_cast<float4*>(tb + t * 16 * C + row * C + ((lane ^ (row & 15)) << 2));
        const float4 u = ures[t];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / C);
        const float rstd = 1.0f / sqrtf(var + p.eps);
        const float4 ww = *reinterpret_cast<const float4*>(p.lw + lane * 4), bb = *reinterpret_cast<const float4*>(p.lb + lane * 4);
        v = make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
        if (m < p.M) *reinterpret_cast<float4*>(p.x_out + grow[t]) = v;
        if (p.Wqh) {
            const float4 qp = *reinterpret_cast<const float4*>(p.qpos + grow[t]);
            uint2 hi, lo;
            split4(make_float4(v.x + qp.x, v.y + qp.y, v.z + qp.z, v.w + qp.w), hi, lo);
            *reinterpret_cast<uint2*>(ah + t * 8192 + aoff) = hi;
            *reinterpret_cast<uint2*>(al + t * 8192 + aoff) = lo;
        }
    }
    if (!p.Wqh) return;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = tile_mma_x3(ah + t * 8192, al + t * 8192, wh, wl, fr, fg);
    const int col = wave * 16 + fr;
    const float b = p.bq[col];
    if (!QMAP) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb + 16 * t + 4 * fg + r;
                if (m < p.M) p.q_out[(long long)m * C + col] = (acc[t][r] + b) * p.qscale;
            }
        return;
    }
    // query map of the tile cross attention (xattn_qmap_kernel's arithmetic): the q tile goes through LDS (tb is free: every wave has
    // passed the barrier behind the LayerNorm reads), wave w = (head w >> 1, channel tiles 8 (w & 1) ..): Qt[m][h][s][g][hi | lo]
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[t * 16 * C + toff(4 * fg + r, col)] = (acc[t][r] + b) * p.qscale;
    __syncthreads();
    {
        const int h = wave >> 1, half = wave & 1;
        const uint4* wah = p.WAh + ((long long)h * 16 + 8 * half) * 64 + lane;
        const uint4* wal = p.WAl + ((long long)h * 16 + 8 * half) * 64 + lane;
        BFrag ahf[8], alf[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ahf[k].u = wah[k * 64]; alf[k].u = wal[k * 64]; }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            BFrag bh, bl;
            const float* qrow = tb + t * 16 * C;
            const int c0 = 32 * h + 8 * fg;
            split8(*reinterpret_cast<const float4*>(qrow + toff(fr, c0)), *reinterpret_cast<const float4*>(qrow + toff(fr, c0 + 4)), bh, bl);
            const int m = mb + 16 * t + fr;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                a0 = mfma_q16_16x16x32(alf[2 * u].v, bh.v, a0, 0, 0, 0);
                a0 = mfma_q16_16x16x32(ahf[2 * u].v, bl.v, a0, 0, 0, 0);
                a0 = mfma_q16_16x16x32(ahf[2 * u].v, bh.v, a0, 0, 0, 0);
                a1 = mfma_q16_16x16x32(alf[2 * u + 1].v, bh.v, a1, 0, 0, 0);
                a1 = mfma_q16_16x16x32(ahf[2 * u + 1].v, bl.v, a1, 0, 0, 0);
                a1 = mfma_q16_16x16x32(ahf[2 * u + 1].v, bh.v, a1, 0, 0, 0);
                BFrag hi, lo;
                {
                    // the operand of the tile cross attention is stored in the KEY-SIDE format (common.h key16: fp16), like xattn_qmap_kernel
                    unsigned int hh[4], ll[4];
                    split_k16x2(a0[0], a0[1], hh[0], ll[0]); split_k16x2(a0[2], a0[3], hh[1], ll[1]);
                    split_k16x2(a1[0], a1[1], hh[2], ll[2]); split_k16x2(a1[2], a1[3], hh[3], ll[3]);
                    hi.u = make_uint4(hh[0], hh[1], hh[2], hh[3]);
                    lo.u = make_uint4(ll[0], ll[1], ll[2], ll[3]);
                }
                if (m < p.M) {
                    uint4* out = p.Qt + (long long)m * 512 + h * 64 + (4 * half + u) * 8 + fg * 2;
                    out[0] = hi.u;
                    out[1] = lo.u;
                }
            }
        }
    }
}

__device__ __forceinline__ float4 ln_row(float4 v, const float* __restrict__ w, const float* __restrict__ b, int c0, float eps);

// Self-attention core + out_proj + residual + LayerNorm + cross-attention q projection for 16 queries in one block (16 waves):
// the attention part is self_attn_kernel's (attention.hip: exact fp32 MFMA, S^T = K.Q^T, online softmax, P feeds P.V from registers)
// with two waves per head splitting the key tiles; the merged context tile goes straight into the bf16 hi/lo LDS images that
// the out_proj reads — it never visits global memory, and one dependent launch per layer disappears.
//   qkv [R,768] fp32 = in_proj outputs (q | k | v), q not yet scaled.
struct SaKV { float4 ka, kb; float v0[4], v1[4]; };

__device__ __forceinline__ void sa_load_kv(SaKV& f, const float* __restrict__ qkv, int t, int h, int fr, int fg, int R) {
    const int krow = min(t * 16 + fr, R - 1);
    const float* kp = qkv + (long long)krow * 768 + C + h * 32 + 4 * fg;
    f.ka = *reinterpret_cast<const float4*>(kp);
    f.kb = *reinterpret_cast<const float4*>(kp + 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int vrow = min(t * 16 + 4 * fg + r, R - 1);
        const float* vp = qkv + (long long)vrow * 768 + 2 * C + h * 32 + fr;
        f.v0[r] = vp[0];
        f.v1[r] = vp[16];
    }
}

__global__ __launch_bounds__(1024) void sa_block_fused_x3_kernel(AttnOutX3Params p, const float* __restrict__ qkv, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char ah[16 * 512], al[16 * 512];
    __shared__ __attribute__((aligned(16))) float tb[16 * C];
    __shared__ float sm[16][16], sl[16][16], so[16][32][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * 16, R = p.M;
    // ---- self attention: head h = wave / 2, key tiles t = part, part + 2, ...
    {
        const int h = wave >> 1, part = wave & 1;
        const int qrow = min(m0 + fr, R - 1);
        const float* qp = qkv + (long long)qrow * 768 + h * 32 + 4 * fg;
        const float4 qa = *reinterpret_cast<const float4*>(qp);
        const float4 qb = *reinterpret_cast<const float4*>(qp + 16);
        float m_run = -INFINITY, l_run = 0.f;
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        const int ntiles = (R + 15) / 16;
        SaKV cur, nxt;
        if (part < ntiles) sa_load_kv(cur, qkv, part, h, fr, fg, R);
        for (int t = part; t < ntiles; t += 2) {
            if (t + 2 < ntiles) sa_load_kv(nxt, qkv, t + 2, h, fr, fg, R);
            f32x4_t sc = {0.f, 0.f, 0.f, 0.f};
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.x, qa.x, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.y, qa.y, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.z, qa.z, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ka.w, qa.w, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.x, qb.x, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.y, qb.y, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.z, qb.z, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kb.w, qb.w, sc, 0, 0, 0);
            float pr[4];
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * 16 + 4 * fg + r;
                pr[r] = key < R ? sc[r] * scale : -INFINITY;
                tmax = fmaxf(tmax, pr[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = expf(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[r] = expf(pr[r] - m_new); psum += pr[r]; }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v0[r], pr[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.v1[r], pr[r], o1, 0, 0, 0);
            }
            cur = nxt;
        }
        if (fg == 0) { sm[wave][fr] = m_run; sl[wave][fr] = l_run; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { so[wave][4 * fg + r][fr] = o0[r]; so[wave][16 + 4 * fg + r][fr] = o1[r]; }
    }
    // out_proj fragments of column tile `wave`: in flight during the merge
    BFrag wh[8], wl[8];
    load_w_x3(wh, wl, p.Woh, p.Wol, wave, lane);
    __syncthreads();
    {   // merge the two partial states of a head: thread -> (head, query, 4 consecutive d) -> bf16 hi/lo images of the context tile
        const int h = tid >> 7, q = tid & 15, d0 = ((tid >> 4) & 7) * 4;
        const float ma = sm[2 * h][q], mb = sm[2 * h + 1][q];
        const float Mx = fmaxf(ma, mb);
        const float ea = expf(ma - Mx), eb = expf(mb - Mx);          // a wave without a tile: exp(-inf) = 0
        const float inv = 1.0f / (sl[2 * h][q] * ea + sl[2 * h + 1][q] * eb);
        float4 c;
        c.x = (so[2 * h][d0][q] * ea + so[2 * h + 1][d0][q] * eb) * inv;
        c.y = (so[2 * h][d0 + 1][q] * ea + so[2 * h + 1][d0 + 1][q] * eb) * inv;
        c.z = (so[2 * h][d0 + 2][q] * ea + so[2 * h + 1][d0 + 2][q] * eb) * inv;
        c.w = (so[2 * h][d0 + 3][q] * ea + so[2 * h + 1][d0 + 3][q] * eb) * inv;
        const int col = h * 32 + d0;                                 // 4 consecutive columns of row q
        const int off = q * 512 + (((col >> 3) ^ q) << 4) + (col & 4) * 2;
        uint2 hi, lo;
        split4(c, hi, lo);
        *reinterpret_cast<uint2*>(ah + off) = hi;
        *reinterpret_cast<uint2*>(al + off) = lo;
    }
    __syncthreads();
    // ---- out_proj + residual + LayerNorm + q projection: identical to attn_out_fused_x3_kernel from here on
    const long long grow = (long long)min(m0 + wave, p.M - 1) * C + lane * 4;
    const int aoff = wave * 512 + (((lane >> 1) ^ wave) << 4) + (lane & 1) * 8;
    f32x4_t acc = tile_mma_x3(ah, al, wh, wl, fr, fg);
    if (p.Wqh) load_w_x3(wh, wl, p.Wqh, p.Wql, wave, lane);
    {
        const int col = wave * 16 + fr;
        const float b = p.bo[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[toff(4 * fg + r, col)] = acc[r] + b;
    }
    __syncthreads();
    {
        const int row = wave, m = m0 + row;
        float4 v = *reinterpret_cast<const float4*>(tb + row * C + ((lane ^ (row & 15)) << 2));
        const float4 u = *reinterpret_cast<const float4*>(p.resid + grow);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        v = ln_row(v, p.lw, p.lb, lane * 4, p.eps);
        if (m < p.M) *reinterpret_cast<float4*>(p.x_out + grow) = v;
        if (p.Wqh) {
            const float4 qp = *reinterpret_cast<const float4*>(p.qpos + grow);
            uint2 hi, lo;
            split4(make_float4(v.x + qp.x, v.y + qp.y, v.z + qp.z, v.w + qp.w), hi, lo);
            *reinterpret_cast<uint2*>(ah + aoff) = hi;
            *reinterpret_cast<uint2*>(al + aoff) = lo;
        }
    }
    if (!p.Wqh) return;
    __syncthreads();
    acc = tile_mma_x3(ah, al, wh, wl, fr, fg);
    const int col = wave * 16 + fr;
    const float b = p.bq[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * fg + r;
        if (m < p.M) p.q_out[(long long)m * C + col] = (acc[r] + b) * p.qscale;
    }
}

// FFN tail + the NEXT layer's self-attention in_proj in one row-fused kernel (16 rows per block, 16 waves):
//   y   = LayerNorm(sum of the FFN slabs + b2 + residual)            (mmcv FFN identity + 'norm', MU/petr_transformer.py:269-311)
//   out = post_norm(y)                                               (decoder post_norm of the intermediate output, :563-565)
//   x = y, xq = y + query_pos, qkv = [xq.Wq^T + bq | xq.Wk^T + bk | x.Wv^T + bv]   (FlattenMHSelfAttention in_proj, :346-363)
// The in_proj runs in bf16x3 split precision like attn_out_fused_x3: wave w owns column tile w of q, of k (activation xq) and of v
// (activation x); its weight fragments stream as half-tiles (4 k-steps, hi + lo = 32 VGPRs) through a double buffer, the next
// half-tile in flight behind the MFMAs of the current one (VGPR budget 128 at 4 waves per SIMD).
struct FfnOutParams {
    const float* parts; int n_parts; long long part_stride; const float* b2; const float* resid;
    const float* lw; const float* lb; const float* pw; const float* pb;
    float* x_out; const float* qpos; float* xq_out; float* outs;
    const unsigned short* Wh; const unsigned short* Wl; const float* b_in; float* qkv;     // Wh null -> no in_proj (last layer)
    int M; float eps;
};

__device__ __forceinline__ float4 ln_row(float4 v, const float* __restrict__ w, const float* __restrict__ b, int c0, float eps) {
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / C);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 ww = *reinterpret_cast<const float4*>(w + c0), bb = *reinterpret_cast<const float4*>(b + c0);
    return make_float4(dx * rstd * ww.x + bb.x, dy * rstd * ww.y + bb.y, dz * rstd * ww.z + bb.z, dw * rstd * ww.w + bb.w);
}

// (RT row tiles per block share the in_proj weight fragments, as in attn_out_fused_x3_kernel)
template <int RT>
__global__ __launch_bounds__(1024) void ffn_out_fused_x3_kernel(FfnOutParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char qh[RT * 16 * 512], ql[RT * 16 * 512], xh[RT * 16 * 512], xl[RT * 16 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int mb = blockIdx.x * (16 * RT);
    // first weight half-tile (q tile of this wave, k-steps 0..3) goes out before anything else
    BFrag wh[2][4], wl[2][4];
    auto load_half = [&](int buf, int st) {                      // stage st = 2 * tile_kind + half; tile_kind 0/1/2 = q/k/v
        const int tile = (st >> 1) * 16 + wave, s0 = (st & 1) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {                            // fragment-major [k-step][48 column tiles][lane][8]
            const long long o = (((long long)(s0 + s) * 48 + tile) * 64 + lane) * 8;
            wh[buf][s].u = *reinterpret_cast<const uint4*>(p.Wh + o);
            wl[buf][s].u = *reinterpret_cast<const uint4*>(p.Wl + o);
        }
    };
    if (p.Wh) load_half(0, 0);
    const int aoff = wave * 512 + (((lane >> 1) ^ wave) << 4) + (lane & 1) * 8;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = mb + 16 * rt + wave;                       // LayerNorm stage: row = wave, lane = 4 consecutive columns
        const long long grow = (long long)min(m, p.M - 1) * C + lane * 4;
        // ---- sum of the slabs (fixed order) + b2 + residual -> LN -> x, xq, post-norm output
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        // bias, residual row and query_pos row are requested together with the slabs (each was a dependent round trip of its own behind them)
        const float4 t_b2 = *reinterpret_cast<const float4*>(p.b2 + lane * 4), u_res = *reinterpret_cast<const float4*>(p.resid + grow);
        const float4 qp = *reinterpret_cast<const float4*>(p.qpos + grow);
        {
            // the slabs were just written by other XCDs (they come from the Infinity Cache, ~2 us away): 16 loads in flight per round trip,
            // summed in the fixed order s = 0, 1, 2, ... (bit-identical to mv2d_row_ln)
            const float* pp = p.parts + grow;
            int s = 0;
            for (; s + 16 <= p.n_parts; s += 16) {
                float4 t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = *reinterpret_cast<const float4*>(pp + (s + j) * p.part_stride);
#pragma unroll
                for (int j = 0; j < 16; ++j) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
            }
            for (; s + 4 <= p.n_parts; s += 4) {               // (8 or 4 slabs when the FFN accumulates several slices per block)
                float4 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const float4*>(pp + (s + j) * p.part_stride);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
            }
            for (; s < p.n_parts; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(pp + s * p.part_stride);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
        }
        {
            const float4 t = t_b2, u = u_res;
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        v = ln_row(v, p.lw, p.lb, lane * 4, p.eps);
        const float4 vq = make_float4(v.x + qp.x, v.y + qp.y, v.z + qp.z, v.w + qp.w);
        if (m < p.M) {
            *reinterpret_cast<float4*>(p.x_out + grow) = v;
            if (p.xq_out) *reinterpret_cast<float4*>(p.xq_out + grow) = vq;       // (optional: with the next in_proj fused nobody reads x + qpos)
            if (p.outs) *reinterpret_cast<float4*>(p.outs + grow) = ln_row(v, p.pw, p.pb, lane * 4, p.eps);
        } else if (p.outs) {
            (void)ln_row(v, p.pw, p.pb, lane * 4, p.eps);        // keep the wave-wide reductions convergent
        }
        if (p.Wh) {
            uint2 hi, lo;
            split4(vq, hi, lo);
            *reinterpret_cast<uint2*>(qh + rt * 8192 + aoff) = hi; *reinterpret_cast<uint2*>(ql + rt * 8192 + aoff) = lo;
            split4(v, hi, lo);
            *reinterpret_cast<uint2*>(xh + rt * 8192 + aoff) = hi; *reinterpret_cast<uint2*>(xl + rt * 8192 + aoff) = lo;
        }
    }
    if (!p.Wh) return;
    __syncthreads();
    // ---- in_proj: 6 half-tile stages (q0 q1 k0 k1 v0 v1), double-buffered weight fragments
#pragma unroll
    for (int kind = 0; kind < 3; ++kind) {
        const unsigned char* ah = kind < 2 ? qh : xh;
        const unsigned char* al = kind < 2 ? ql : xl;
        f32x4_t a0[RT], a1[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { a0[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; a1[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int st = 2 * kind + h, buf = st & 1;
            if (st + 1 < 6) load_half(buf ^ 1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    BFrag ya, yb;
                    const int off = rt * 8192 + fr * 512 + (((4 * (4 * h + s) + fg) ^ fr) << 4);
                    ya.u = *reinterpret_cast<const uint4*>(ah + off);
                    yb.u = *reinterpret_cast<const uint4*>(al + off);
                    a0[rt] = mfma_q16_16x16x32(ya.v, wh[buf][s].v, a0[rt], 0, 0, 0);
                    a1[rt] = mfma_q16_16x16x32(yb.v, wh[buf][s].v, a1[rt], 0, 0, 0);
                    a1[rt] = mfma_q16_16x16x32(ya.v, wl[buf][s].v, a1[rt], 0, 0, 0);
                }
            }
        }
        const int col = kind * 256 + wave * 16 + fr;
        const float b = p.b_in[col];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = mb + 16 * rt + 4 * fg + r;
                if (mm < p.M) p.qkv[(long long)mm * 768 + col] = (a0[rt][r] + a1[rt][r]) + b;
            }
    }
}

// Tail of the query generator + the query positional embedding, row-fused (16 RoIs per block, 16 waves):
//   center = fc_center(enc2)                       (RH/utils/query_generator.py:404; exact fp32, wave = RoI, wavefront dot products)
//   xyz = center2lidar(center), ref = (xyz - pc_min) / pc_extent   (query_generator.py:333-341, RH/mv2d_t_head.py:51-57 — no clamp)
//   posemb = pos2posemb3d(ref)                     (MU/pe.py:21-33, (y | x | z) blocks of sin/cos pairs)
//   qpos = query_embedding(posemb) = Linear(384,256)-ReLU-Linear(256,256)   (cross_attention_head.py:118-125; bf16x3)
// replaces 4 dependent launches (GEMM N=3, refpoint kernel, 2 GEMMs).
struct QEmbParams {
    const float* enc2; const float* Wc; const float* bc; const float* minv; const float* dim_t;
    float pc0, pc1, pc2, pd0, pd1, pd2;
    const unsigned short* W0h; const unsigned short* W0l; const float* b0; const unsigned short* W2h; const unsigned short* W2l; const float* b2;
    float* center; float* xyz; float* ref; float* posemb; float* qpos; int R;
};

__global__ __launch_bounds__(1024) void query_embed_fused_x3_kernel(QEmbParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char eh[16 * 1024], el[16 * 1024];      // posemb tile, K = 384 (1 KB pitch)
    __shared__ __attribute__((aligned(16))) unsigned char hh[16 * 512], hl[16 * 512];        // hidden tile, K = 256
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int r = min(m0 + wave, p.R - 1);
    const bool live = m0 + wave < p.R;
    // first-layer weight fragments of column tile `wave` (12 k-steps, hi + lo, fragment-major [k-step][16 tiles][lane][8]): the first
    // 8 now, the last 4 into the slots of the first 4 once those are consumed (VGPR budget 128)
    BFrag wh[8], wl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const long long o = (((long long)s * 16 + wave) * 64 + lane) * 8;
        wh[s].u = *reinterpret_cast<const uint4*>(p.W0h + o);
        wl[s].u = *reinterpret_cast<const uint4*>(p.W0l + o);
    }
    // ---- fc_center: three 256-long dot products per RoI
    const float4 e = *reinterpret_cast<const float4*>(p.enc2 + (long long)r * C + lane * 4);
    float cp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(p.Wc + k * C + lane * 4);
        cp[k] = wave_sum(e.x * w.x + e.y * w.y + e.z * w.z + e.w * w.w) + p.bc[k];
    }
    // ---- center2lidar + normalisation (same operation order as refpoint_posemb_kernel)
    const float cc[4] = {cp[0] * cp[2], cp[1] * cp[2], cp[2], 1.0f};
    float pt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = acc + p.minv[r * 16 + i * 4 + k] * cc[k];
        pt[i] = acc;
    }
    const float n0 = (pt[0] - p.pc0) / p.pd0, n1 = (pt[1] - p.pc1) / p.pd1, n2 = (pt[2] - p.pc2) / p.pd2;
    if (live && lane < 3) {
        p.center[r * 3 + lane] = cp[lane];
        p.xyz[r * 3 + lane] = pt[lane];
        p.ref[r * 3 + lane] = lane == 0 ? n0 : (lane == 1 ? n1 : n2);
    }
    // ---- pos2posemb3d: 384 channels per RoI -> global + bf16 hi/lo images (row = wave)
    {
        const float two_pi = 6.283185307179586f;
        const float py = n1 * two_pi, px = n0 * two_pi, pz = n2 * two_pi;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ch = lane + 64 * j, axis = ch >> 7, i = ch & 127;
            const float pos = axis == 0 ? py : (axis == 1 ? px : pz);
            const float a = pos / p.dim_t[i];
            const float v = (i & 1) ? cosf(a) : sinf(a);
            if (live) p.posemb[(long long)r * 384 + ch] = v;
            unsigned short hi, lo;
            split_q16(v, hi, lo);
            const int off = wave * 1024 + (((ch >> 3) ^ wave) << 4) + (ch & 7) * 2;
            *reinterpret_cast<unsigned short*>(eh + off) = hi;
            *reinterpret_cast<unsigned short*>(el + off) = lo;
        }
    }
    __syncthreads();
    // ---- query_embedding.0 + ReLU -> hidden tile images
    {
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            BFrag xh, xl;
            const int off = fr * 1024 + (((4 * s + fg) ^ fr) << 4);
            xh.u = *reinterpret_cast<const uint4*>(eh + off);
            xl.u = *reinterpret_cast<const uint4*>(el + off);
            a0 = mfma_q16_16x16x32(xh.v, wh[s & 7].v, a0, 0, 0, 0);
            a1 = mfma_q16_16x16x32(xl.v, wh[s & 7].v, a1, 0, 0, 0);
            a1 = mfma_q16_16x16x32(xh.v, wl[s & 7].v, a1, 0, 0, 0);
            if (s == 3) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long long o = (((long long)(8 + t) * 16 + wave) * 64 + lane) * 8;
                    wh[t].u = *reinterpret_cast<const uint4*>(p.W0h + o);
                    wl[t].u = *reinterpret_cast<const uint4*>(p.W0l + o);
                }
            }
        }
        // second-layer fragments into the same registers, in flight during the hidden-tile write
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const long long o = (((long long)s * 16 + wave) * 64 + lane) * 8;
            wh[s].u = *reinterpret_cast<const uint4*>(p.W2h + o);
            wl[s].u = *reinterpret_cast<const uint4*>(p.W2l + o);
        }
        const int col = wave * 16 + fr;
        const float b = p.b0[col];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 4 * fg + q;
            const float v = relu_f((a0[q] + a1[q]) + b);
            unsigned short hi, lo;
            split_q16(v, hi, lo);
            const int off = row * 512 + (((col >> 3) ^ row) << 4) + (col & 7) * 2;
            *reinterpret_cast<unsigned short*>(hh + off) = hi;
            *reinterpret_cast<unsigned short*>(hl + off) = lo;
        }
    }
    __syncthreads();
    // ---- query_embedding.2
    {
        const f32x4_t acc = tile_mma_x3(hh, hl, wh, wl, fr, fg);
        const int col = wave * 16 + fr;
        const float b = p.b2[col];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + 4 * fg + q;
            if (m < p.R) p.qpos[(long long)m * C + col] = acc[q] + b;
        }
    }
}

struct HeadsParams {
    const float* outs;            // [L, M, 256]
    const float* w0; const float* b0; const float* lnw1; const float* lnb1; const float* w3; const float* b3; const float* lnw4; const float* lnb4;
    const float* w6; const float* b6;                                   // cls branch, stacked per layer
    const float* r0; const float* rb0; const float* r2; const float* rb2; const float* r4; const float* rb4;   // reg branch
    const float* ref;             // [M,3]
    float* cls; float* reg;       // [L, M, 10]
    int M, L; float eps; float pc0, pc1, pc2, pd0, pd1, pd2, dt;
    const float* dt_rows;         // optional [M]: per-row time step (a batch of samples), overrides dt
};

__global__ __launch_bounds__(256, 1) void heads_fused_kernel(HeadsParams p) {
    __shared__ __attribute__((aligned(16))) float ta[16 * C];
    __shared__ __attribute__((aligned(16))) float tb[16 * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * 16, l = blockIdx.y, branch = blockIdx.z;
    const long long wl = (long long)l * C * C, bl = (long long)l * C;
    load_rows(ta, p.outs + (long long)l * p.M * C, nullptr, m0, p.M, tid);
    __syncthreads();
    f32x4_t acc[4];
    const float* wlast; const float* blast; float* outp;
    if (branch == 0) {
        linear256<true>(ta, p.w0 + wl, wave, fr, fg, acc);
        store_tile(tb, acc, p.b0 + bl, wave, fr, fg, false, 1.0f);
        __syncthreads();
        ln_tile(tb, nullptr, p.lnw1 + bl, p.lnb1 + bl, true, nullptr, nullptr, nullptr, m0, p.M, wave, lane, p.eps);
        __syncthreads();
        linear256<true>(tb, p.w3 + wl, wave, fr, fg, acc);
        store_tile(ta, acc, p.b3 + bl, wave, fr, fg, false, 1.0f);
        __syncthreads();
        ln_tile(ta, nullptr, p.lnw4 + bl, p.lnb4 + bl, true, nullptr, nullptr, nullptr, m0, p.M, wave, lane, p.eps);
        __syncthreads();
        wlast = p.w6 + (long long)l * 10 * C; blast = p.b6 + l * 10; outp = p.cls;
    } else {
        linear256<true>(ta, p.r0 + wl, wave, fr, fg, acc);
        store_tile(tb, acc, p.rb0 + bl, wave, fr, fg, true, 1.0f);
        __syncthreads();
        linear256<true>(tb, p.r2 + wl, wave, fr, fg, acc);
        store_tile(ta, acc, p.rb2 + bl, wave, fr, fg, true, 1.0f);
        __syncthreads();
        wlast = p.r4 + (long long)l * 10 * C; blast = p.rb4 + l * 10; outp = p.reg;
    }
    if (wave != 0) return;
    // final Linear(256 -> 10): one 16x16 tile, weight rows >= 10 clamped and masked
    Frag f;
    load_w(f, wlast, C, fr, 10, fg);
    const f32x4_t o = tile_mma(ta, f, fr, fg);
    if (fr >= 10) return;
    const float b = blast[fr];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * fg + r;
        if (m >= p.M) continue;
        float v = o[r] + b;
        if (branch == 1) {
            // cross_attention_head.py:219-238: add inverse_sigmoid(ref) to (cx, cy) and cz, sigmoid, de-normalise; T head: v / dt
            if (fr == 0 || fr == 1 || fr == 4) {
                const int k = fr == 4 ? 2 : fr;
                const float x = fminf(fmaxf(p.ref[m * 3 + k], 0.f), 1.f);
                const float is = logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
                const float s = 1.f / (1.f + expf(-(v + is)));
                v = fr == 0 ? s * p.pd0 + p.pc0 : (fr == 1 ? s * p.pd1 + p.pc1 : s * p.pd2 + p.pc2);
            } else if (fr >= 8) {
                const float dt = p.dt_rows ? p.dt_rows[m] : p.dt;
                if (dt != 0.f) v = v / dt;
            }
        }
        outp[((long long)l * p.M + m) * 10 + fr] = v;
    }
}

// W [N, ldw] fp32 -> fragment-major Wp[ceil(N/16)][K/16][64][4]: Wp[tile][c][fr + 16 fg][e] = W[min(16 tile + fr, N-1)][16 c + 4 fg + e]
__global__ void pack_wfrag_f32_kernel(const float* __restrict__ W, float* __restrict__ Wp, int N, int K, int ldw) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one float4 per thread
    const int tiles = (N + 15) / 16, kc = K / 16;
    if (idx >= (long long)tiles * kc * 64) return;
    const int lane = idx & 63, fr = lane & 15, fg = lane >> 4;
    const long long t = idx >> 6;
    const int c = (int)(t % kc), tile = (int)(t / kc);
    const int row = min(16 * tile + fr, N - 1);
    *reinterpret_cast<float4*>(Wp + idx * 4) = *reinterpret_cast<const float4*>(W + (long long)row * ldw + 16 * c + 4 * fg);
}


// Plain linear layer C = act(A . W^T + b) in split precision for the per-query MLPs with several hundred to a few thousand rows
// (query generator: 256 -> 1024, 1056 -> 512, 512 -> 256; first self-attention in_proj): the per-wave-tile kernels (gemm_f32 /
// the retired round-2 gemm_x3 kernel: no LDS, every 16x16 tile fetches its own operands) are latency kernels for a few hundred rows and L2-bound beyond.
// Block = RT x 16 rows x 128 columns, 8 waves (wave w = column tile w for all row tiles); K runs in chunks of 256: the fp32 A chunk
// is split into bf16 hi / lo LDS images (double buffered), the fragment-major weight chunk (hi, lo) sits in registers, both are
// requested one chunk ahead.  Columns >= n_split read A2 instead of A (in_proj of (q, k | v) from two inputs).
struct LinX3Params {
    const float* A; const float* A2; int n_split; int lda; const unsigned short* Wh; const unsigned short* Wl; const float* bias;
    float* C; int ldc; int M, N, K; int act; float clamp;
    long long a_gs, w_gs, b_gs, c_gs;       // element strides between the groups of a batched call (blockIdx.z = group)
    // extensions (mv2d_linear_x3_ex; the index-exact route of the engine): device-side row count, implicit 3x3 convolution over
    // [R,49,256] RoI cells (chunk c of K = 2304 is tap c: the A row of (RoI, cell) is the neighbouring cell's 256 channels or zero),
    // sigmoid (act == 2), C = v * mul + add with fp32 operands [M, ld_ma]
    const int* m_dev; int conv3x3; const float* mul; const float* add; int ld_ma;
};

template <int RT>
__global__ __launch_bounds__(512) void linear_x3_kernel(LinX3Params p) {
    {   // group g of a batched call: its own A columns / weights / bias / C columns
        const long long g = blockIdx.z;
        p.A += g * p.a_gs; if (p.A2) p.A2 += g * p.a_gs;
        p.Wh += g * p.w_gs; p.Wl += g * p.w_gs;
        if (p.bias) p.bias += g * p.b_gs;
        p.C += g * p.c_gs;
    }
    __shared__ __attribute__((aligned(16))) unsigned char ah[2][RT * 16 * 512], al[2][RT * 16 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    // the column blocks of one row block run on ONE XCD (they read the same A rows: round 4, PMC: 4-8 x the A bytes fetched before)
    const int lin = xcd_chunked(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int mb = (lin / gridDim.x) * (16 * RT), n0 = (lin % gridDim.x) * 128;
    if (p.m_dev) {                                                              // device-side M (block-uniform exit before any barrier)
        p.M = min(p.M, *p.m_dev);
        if (mb >= p.M) return;
    }
    const int ntile = p.N >> 4, tile = min((n0 >> 4) + wave, ntile - 1);       // clamped: the extra waves of a ragged last block recompute
    const float* A = (p.n_split > 0 && n0 >= p.n_split) ? p.A2 : p.A;
    const int nchunk = (p.K + 255) >> 8;
    constexpr int NA = RT * 2;                                                  // float4 per thread and chunk
    float4 ar[NA];
    BFrag wh[8], wl[8], wh2[8], wl2[8];
    auto load_a = [&](int c) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 512 * i, row = idx >> 6, q = idx & 63, k = 256 * c + 4 * q;
            if (p.conv3x3) {
                const int m = min(mb + row, p.M - 1), r = m / 49, cell = m - 49 * r;
                const int yy = cell / 7 + c / 3 - 1, xx = cell % 7 + c % 3 - 1;
                ar[i] = (yy >= 0 && yy < 7 && xx >= 0 && xx < 7) ? *reinterpret_cast<const float4*>(A + ((long long)r * 49 + yy * 7 + xx) * 256 + 4 * q)
                                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                continue;
            }
            ar[i] = k < p.K ? *reinterpret_cast<const float4*>(A + (long long)min(mb + row, p.M - 1) * p.lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 512 * i, row = idx >> 6, q = idx & 63;
            const int off = row * 512 + (((q >> 1) ^ (row & 15)) << 4) + (q & 1) * 8;
            uint2 hi, lo;
            split4(ar[i], hi, lo);
            *reinterpret_cast<uint2*>(ah[buf] + off) = hi;
            *reinterpret_cast<uint2*>(al[buf] + off) = lo;
        }
    };
    auto load_w = [&](BFrag (&h)[8], BFrag (&l)[8], int c) {
        const int ns = min(8, (p.K - 256 * c) >> 5);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < ns) {                                                       // fragment-major [K/32][N/16][lane][8]
                const long long o = (((long long)(8 * c + s) * ntile + tile) * 64 + lane) * 8;
                h[s].u = *reinterpret_cast<const uint4*>(p.Wh + o);
                l[s].u = *reinterpret_cast<const uint4*>(p.Wl + o);
            }
        }
    };
    load_a(0);
    load_w(wh, wl, 0);
    store_a(0);
    __syncthreads();
    f32x4_t a0[RT], a1[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { a0[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; a1[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) { load_a(c + 1); load_w(wh2, wl2, c + 1); }
        const int ns = min(8, (p.K - 256 * c) >> 5);
        const unsigned char* bh = ah[c & 1];
        const unsigned char* bl = al[c & 1];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < ns) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    BFrag xh, xl;
                    const int off = t * 8192 + fr * 512 + (((4 * s + fg) ^ fr) << 4);
                    xh.u = *reinterpret_cast<const uint4*>(bh + off);
                    xl.u = *reinterpret_cast<const uint4*>(bl + off);
                    a0[t] = mfma_q16_16x16x32(xh.v, wh[s].v, a0[t], 0, 0, 0);
                    a1[t] = mfma_q16_16x16x32(xl.v, wh[s].v, a1[t], 0, 0, 0);
                    a1[t] = mfma_q16_16x16x32(xh.v, wl[s].v, a1[t], 0, 0, 0);
                }
            }
        }
        if (more) {
            store_a((c + 1) & 1);              // the other buffer: its last readers passed the previous barrier
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 8; ++s) { wh[s] = wh2[s]; wl[s] = wl2[s]; }
        }
    }
    const int col = tile * 16 + fr;
    if ((n0 >> 4) + wave >= ntile) return;
    const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mb + 16 * t + 4 * fg + r;
            if (m >= p.M) continue;
            float v = (a0[t][r] + a1[t][r]) + b;
            if (p.act == 1) v = relu_f(v);
            else if (p.act == 2) v = 1.f / (1.f + expf(-v));
            if (p.clamp > 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
            if (p.mul) v = v * p.mul[(long long)m * p.ld_ma + col];
            if (p.add) v = v + p.add[(long long)m * p.ld_ma + col];
            p.C[(long long)m * p.ldc + col] = v;
        }
}

// (hi, lo) key16 rows (common.h: fp16) of a (+ b): the key / value rows of the index-exact route.  n4 = number of float4; rows beyond *m_dev are skipped.
__global__ void split_rows_kernel(const float4* __restrict__ a, const float4* __restrict__ b, uint2* __restrict__ hi, uint2* __restrict__ lo,
                                  long long n4, const int* __restrict__ m_dev, int row4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4 || (m_dev && i >= (long long)*m_dev * row4)) return;
    float4 v = a[i];
    if (b) { const float4 w = b[i]; v = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w); }
    uint2 h, l;
    split_k16x2(v.x, v.y, h.x, l.x);
    split_k16x2(v.z, v.w, h.y, l.y);
    hi[i] = h;
    lo[i] = l;
}

// The prediction branches in split precision (bf16x3): same chain as heads_fused_kernel with the four 256x256 linears of a
// (layer, branch) on v_mfma_f32_16x16x32_bf16 (hi/lo pairs, ~1e-5 relative), 16 waves per block: wave w owns column tile w of a
// linear and row w of the LayerNorm / ReLU stage; the 256 -> 10 output layer stays exact fp32.  (Profile of a 4-sample batch: the
// exact-fp32 kernel spent 93 us in 2048-cycle MFMA chains per wave.)
struct HeadsX3Params {
    const float* outs;
    const unsigned short* w0h; const unsigned short* w0l; const float* b0; const float* lnw1; const float* lnb1;
    const unsigned short* w3h; const unsigned short* w3l; const float* b3; const float* lnw4; const float* lnb4;
    const float* w6; const float* b6;
    const unsigned short* r0h; const unsigned short* r0l; const float* rb0; const unsigned short* r2h; const unsigned short* r2l; const float* rb2;
    const float* r4; const float* rb4;
    const float* ref; float* cls; float* reg;
    int M, L; float eps; float pc0, pc1, pc2, pd0, pd1, pd2, dt; const float* dt_rows;
};

// RT row tiles (16 rows each) per block share one load of the weight fragments: with many rows (a batch of samples) the kernel is
// bound by the L2 -> CU traffic of the weights (0.5 MB per block), not by the matrix pipe.
template <int RT>
__global__ __launch_bounds__(1024) void heads_fused_x3_kernel(HeadsX3Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char ah[RT * 16 * 512], al[RT * 16 * 512];
    __shared__ __attribute__((aligned(16))) float tb[RT * 16 * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    // the row blocks of one (layer, branch) run on ONE XCD: its 1 MB of weights is fetched by ~2 XCDs instead of all 8
    const int lin = xcd_chunked(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int mb = (lin % gridDim.x) * (16 * RT), l = (lin / gridDim.x) % gridDim.y, branch = lin / (gridDim.x * gridDim.y);
    const long long wo = (long long)l * C * C, bl = (long long)l * C;
    const unsigned short* W1h = (branch == 0 ? p.w0h : p.r0h) + wo;
    const unsigned short* W1l = (branch == 0 ? p.w0l : p.r0l) + wo;
    const unsigned short* W2h = (branch == 0 ? p.w3h : p.r2h) + wo;
    const unsigned short* W2l = (branch == 0 ? p.w3l : p.r2l) + wo;
    const float* B1 = (branch == 0 ? p.b0 : p.rb0) + bl;
    const float* B2 = (branch == 0 ? p.b3 : p.rb2) + bl;
    float4 av[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
        av[t] = *reinterpret_cast<const float4*>(p.outs + ((long long)l * p.M + min(mb + 16 * t + wave, p.M - 1)) * C + lane * 4);
    BFrag wh[8], wl[8];
    load_w_x3(wh, wl, W1h, W1l, wave, lane);
    const int aoff = wave * 512 + (((lane >> 1) ^ wave) << 4) + (lane & 1) * 8;     // this thread's 4 values in the bf16 images of a tile
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        uint2 hi, lo;
        split4(av[t], hi, lo);
        *reinterpret_cast<uint2*>(ah + t * 8192 + aoff) = hi;
        *reinterpret_cast<uint2*>(al + t * 8192 + aoff) = lo;
    }
    __syncthreads();
    const int col = wave * 16 + fr;
    float* trow = tb + wave * C + ((lane ^ (wave & 15)) << 2);          // row = wave of a tile, columns 4 lane ..
    // ---- linear 1 -> (LayerNorm) -> ReLU
    f32x4_t acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = tile_mma_x3(ah + t * 8192, al + t * 8192, wh, wl, fr, fg);
    load_w_x3(wh, wl, W2h, W2l, wave, lane);                            // in flight during the row stage
    {
        const float b = B1[col];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[t * 16 * C + toff(4 * fg + r, col)] = acc[t][r] + b;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float4 v = *reinterpret_cast<float4*>(trow + t * 16 * C);
        if (branch == 0) v = ln_row(v, p.lnw1 + bl, p.lnb1 + bl, lane * 4, p.eps);
        v = make_float4(relu_f(v.x), relu_f(v.y), relu_f(v.z), relu_f(v.w));
        uint2 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<uint2*>(ah + t * 8192 + aoff) = hi;
        *reinterpret_cast<uint2*>(al + t * 8192 + aoff) = lo;
    }
    __syncthreads();
    // ---- linear 2 -> (LayerNorm) -> ReLU, kept as an fp32 tile for the output layer
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = tile_mma_x3(ah + t * 8192, al + t * 8192, wh, wl, fr, fg);
    {
        const float b = B2[col];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[t * 16 * C + toff(4 * fg + r, col)] = acc[t][r] + b;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float4 v = *reinterpret_cast<float4*>(trow + t * 16 * C);
        if (branch == 0) v = ln_row(v, p.lnw4 + bl, p.lnb4 + bl, lane * 4, p.eps);
        *reinterpret_cast<float4*>(trow + t * 16 * C) = make_float4(relu_f(v.x), relu_f(v.y), relu_f(v.z), relu_f(v.w));
    }
    __syncthreads();
    if (wave >= RT) return;
    const int m0 = mb + 16 * wave;                                      // wave t finishes row tile t
    if (m0 >= p.M) return;
    const float* tbt = tb + wave * 16 * C;
    // ---- final Linear(256 -> 10): one 16x16 tile, weight rows >= 10 clamped and masked
    const float* wlast = branch == 0 ? p.w6 + (long long)l * 10 * C : p.r4 + (long long)l * 10 * C;
    const float* blast = branch == 0 ? p.b6 + l * 10 : p.rb4 + l * 10;
    float* outp = branch == 0 ? p.cls : p.reg;
    Frag f;
    load_w(f, wlast, C, fr, 10, fg);
    const f32x4_t o = tile_mma(tbt, f, fr, fg);
    if (fr >= 10) return;
    const float b = blast[fr];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * fg + r;
        if (m >= p.M) continue;
        float v = o[r] + b;
        if (branch == 1) {
            // cross_attention_head.py:219-238: add inverse_sigmoid(ref) to (cx, cy) and cz, sigmoid, de-normalise; T head: v / dt
            if (fr == 0 || fr == 1 || fr == 4) {
                const int k = fr == 4 ? 2 : fr;
                const float x = fminf(fmaxf(p.ref[m * 3 + k], 0.f), 1.f);
                const float is = logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
                const float sg = 1.f / (1.f + expf(-(v + is)));
                v = fr == 0 ? sg * p.pd0 + p.pc0 : (fr == 1 ? sg * p.pd1 + p.pc1 : sg * p.pd2 + p.pc2);
            } else if (fr >= 8) {
                const float dt = p.dt_rows ? p.dt_rows[m] : p.dt;
                if (dt != 0.f) v = v / dt;
            }
        }
        outp[((long long)l * p.M + m) * 10 + fr] = v;
    }
}

}  // namespace

extern "C" int mv2d_pack_wfrag_f32(const float* W, float* Wp, int N, int K, int ldw, void* stream) {
    MV2D_CHECK_ARG(W && Wp && N > 0 && K > 0 && (K % 16) == 0 && (ldw % 4) == 0 && ldw >= K, "mv2d_pack_wfrag_f32: bad args");
    const long long total = (long long)((N + 15) / 16) * (K / 16) * 64;
    hipLaunchKernelGGL(pack_wfrag_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, Wp, N, K, ldw);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_attn_out_fused(const float* ctx, const float* resid, const float* Wo, const float* bo, const float* ln_w, const float* ln_b,
                                   float* x_out, const float* qpos, const float* Wq, const float* bq, float qscale, float* q_out, int M,
                                   float eps, void* stream) {
    MV2D_CHECK_ARG(ctx && resid && Wo && bo && ln_w && ln_b && x_out, "mv2d_attn_out_fused: null pointer");
    MV2D_CHECK_ARG(!Wq || (qpos && bq && q_out), "mv2d_attn_out_fused: the q stage needs qpos, bq and q_out");
    if (M == 0) return MV2D_OK;
    AttnOutParams p{ctx, resid, Wo, bo, ln_w, ln_b, x_out, qpos, Wq, bq, qscale, q_out, M, eps};
    hipLaunchKernelGGL(attn_out_fused_kernel, dim3(cdiv(M, 16)), dim3(1024), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_attn_out_fused_x3(const float* ctx, const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo,
                                      const float* ln_w, const float* ln_b, float* x_out, const float* qpos, const void* Wq_hi,
                                      const void* Wq_lo, const float* bq, float qscale, float* q_out, int M, float eps, void* stream) {
    MV2D_CHECK_ARG(ctx && resid && Wo_hi && Wo_lo && bo && ln_w && ln_b && x_out, "mv2d_attn_out_fused_x3: null pointer");
    MV2D_CHECK_ARG(!Wq_hi || (Wq_lo && qpos && bq && q_out), "mv2d_attn_out_fused_x3: the q stage needs Wq_lo, qpos, bq and q_out");
    if (M == 0) return MV2D_OK;
    AttnOutX3Params p{ctx, resid, (const unsigned short*)Wo_hi, (const unsigned short*)Wo_lo, bo, ln_w, ln_b, x_out, qpos,
                      (const unsigned short*)Wq_hi, (const unsigned short*)Wq_lo, bq, qscale, q_out, M, eps,
                      nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
    if (M <= 512) hipLaunchKernelGGL((attn_out_fused_x3_kernel<1, false, false>), dim3(cdiv(M, 16)), dim3(1024), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_out_fused_x3_kernel<2, false, false>), dim3(cdiv(M, 32)), dim3(1024), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// The two row kernels around the tile cross attention with its per-head maps fused in (csrc/xattn_tile.hip):
//   mv2d_attn_out_qmap_x3: out_proj + residual + LayerNorm of the SELF attention, cross-attention q projection, and the query map
//                          Qt (= mv2d_attn_out_fused_x3 + mv2d_xattn_qmap, bitwise; q itself is not written)
//   mv2d_attn_out_zmap_x3: context map of z (= mv2d_xattn_ctxmap) + out_proj + residual + LayerNorm of the CROSS attention
extern "C" int mv2d_attn_out_qmap_x3(const float* ctx, const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo,
                                     const float* ln_w, const float* ln_b, float* x_out, const float* qpos, const void* Wq_hi,
                                     const void* Wq_lo, const float* bq, float qscale, const void* WA_hi, const void* WA_lo, void* Qt, int M,
                                     float eps, void* stream) {
    MV2D_CHECK_ARG(ctx && resid && Wo_hi && Wo_lo && bo && ln_w && ln_b && x_out && qpos && Wq_hi && Wq_lo && bq && WA_hi && WA_lo && Qt,
                   "mv2d_attn_out_qmap_x3: null pointer");
    if (M == 0) return MV2D_OK;
    AttnOutX3Params p{ctx, resid, (const unsigned short*)Wo_hi, (const unsigned short*)Wo_lo, bo, ln_w, ln_b, x_out, qpos,
                      (const unsigned short*)Wq_hi, (const unsigned short*)Wq_lo, bq, qscale, nullptr, M, eps,
                      nullptr, nullptr, nullptr, nullptr, nullptr, 0, (const uint4*)WA_hi, (const uint4*)WA_lo, (uint4*)Qt};
    if (M <= 512) hipLaunchKernelGGL((attn_out_fused_x3_kernel<1, false, true>), dim3(cdiv(M, 16)), dim3(1024), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_out_fused_x3_kernel<2, false, true>), dim3(cdiv(M, 32)), dim3(1024), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_attn_out_zmap_x3(const float* z, const void* WB_hi, const void* WB_lo, const float* bv, const int* row_ptr, int empty_nan,
                                     const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo, const float* ln_w,
                                     const float* ln_b, float* x_out, int M, float eps, void* stream) {
    MV2D_CHECK_ARG(z && WB_hi && WB_lo && bv && row_ptr && resid && Wo_hi && Wo_lo && bo && ln_w && ln_b && x_out, "mv2d_attn_out_zmap_x3: null pointer");
    if (M == 0) return MV2D_OK;
    AttnOutX3Params p{nullptr, resid, (const unsigned short*)Wo_hi, (const unsigned short*)Wo_lo, bo, ln_w, ln_b, x_out, nullptr,
                      nullptr, nullptr, nullptr, 1.0f, nullptr, M, eps,
                      z, (const uint4*)WB_hi, (const uint4*)WB_lo, bv, row_ptr, empty_nan, nullptr, nullptr, nullptr};
    if (M <= 512) hipLaunchKernelGGL((attn_out_fused_x3_kernel<1, true, false>), dim3(cdiv(M, 16)), dim3(1024), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_out_fused_x3_kernel<2, true, false>), dim3(cdiv(M, 32)), dim3(1024), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_sa_block_fused_x3(const float* qkv, const float* resid, const void* Wo_hi, const void* Wo_lo, const float* bo,
                                      const float* ln_w, const float* ln_b, float* x_out, const float* qpos, const void* Wq_hi,
                                      const void* Wq_lo, const float* bq, float qscale, float* q_out, int M, float eps, void* stream) {
    MV2D_CHECK_ARG(qkv && resid && Wo_hi && Wo_lo && bo && ln_w && ln_b && x_out, "mv2d_sa_block_fused_x3: null pointer");
    MV2D_CHECK_ARG(!Wq_hi || (Wq_lo && qpos && bq && q_out), "mv2d_sa_block_fused_x3: the q stage needs Wq_lo, qpos, bq and q_out");
    if (M == 0) return MV2D_OK;
    AttnOutX3Params p{nullptr, resid, (const unsigned short*)Wo_hi, (const unsigned short*)Wo_lo, bo, ln_w, ln_b, x_out, qpos,
                      (const unsigned short*)Wq_hi, (const unsigned short*)Wq_lo, bq, qscale, q_out, M, eps};
    hipLaunchKernelGGL(sa_block_fused_x3_kernel, dim3(cdiv(M, 16)), dim3(1024), 0, (hipStream_t)stream, p, qkv, 1.0f / sqrtf(32.0f));
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_query_embed_fused_x3(const float* enc2, const float* Wc, const float* bc, const float* minv, const float* dim_t,
                                        const float* pc_range, const void* W0_hi, const void* W0_lo, const float* b0, const void* W2_hi,
                                        const void* W2_lo, const float* b2, float* center, float* xyz, float* ref, float* posemb,
                                        float* qpos, int R, void* stream) {
    MV2D_CHECK_ARG(enc2 && Wc && bc && minv && dim_t && pc_range && W0_hi && W0_lo && b0 && W2_hi && W2_lo && b2 && center && xyz && ref &&
                       posemb && qpos, "mv2d_query_embed_fused_x3: null pointer");
    if (R == 0) return MV2D_OK;
    QEmbParams p{enc2, Wc, bc, minv, dim_t, pc_range[0], pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1],
                 pc_range[5] - pc_range[2], (const unsigned short*)W0_hi, (const unsigned short*)W0_lo, b0, (const unsigned short*)W2_hi,
                 (const unsigned short*)W2_lo, b2, center, xyz, ref, posemb, qpos, R};
    hipLaunchKernelGGL(query_embed_fused_x3_kernel, dim3(cdiv(R, 16)), dim3(1024), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_ffn_out_fused_x3(const float* parts, int n_parts, long long part_stride, const float* b2, const float* resid,
                                     const float* ln_w, const float* ln_b, const float* post_w, const float* post_b, float* x_out,
                                     const float* qpos, float* xq_out, float* outs, const void* Win_hi, const void* Win_lo,
                                     const float* b_in, float* qkv, int M, float eps, void* stream) {
    MV2D_CHECK_ARG(parts && n_parts > 0 && b2 && resid && ln_w && ln_b && x_out && qpos, "mv2d_ffn_out_fused_x3: null pointer");
    MV2D_CHECK_ARG(!outs || (post_w && post_b), "mv2d_ffn_out_fused_x3: outs needs the post_norm parameters");
    MV2D_CHECK_ARG(!Win_hi || (Win_lo && b_in && qkv), "mv2d_ffn_out_fused_x3: the in_proj stage needs Win_lo, b_in and qkv");
    if (M == 0) return MV2D_OK;
    FfnOutParams p{parts, n_parts, part_stride, b2, resid, ln_w, ln_b, post_w, post_b, x_out, qpos, xq_out, outs,
                   (const unsigned short*)Win_hi, (const unsigned short*)Win_lo, b_in, qkv, M, eps};
    // 32 rows per block above 512 rows (bitwise the same rows either way, tests/test_gpu_engine.py batch == single): no gain at 8 samples per launch
    // (the slab sum doubles per block), +1.2 % samples/s at 16 per launch (150 instead of 300 blocks, half the weight stream)
    if (M <= 512) hipLaunchKernelGGL(ffn_out_fused_x3_kernel<1>, dim3(cdiv(M, 16)), dim3(1024), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(ffn_out_fused_x3_kernel<2>, dim3(cdiv(M, 32)), dim3(1024), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_heads_fused(const float* outs, const float* const* cls_w, const float* const* reg_w, const float* ref, float* cls, float* reg,
                                int M, int L, float eps, const float* pc_range, float dt, const float* dt_rows, void* stream) {
    // cls_w: {w0,b0,lnw1,lnb1,w3,b3,lnw4,lnb4,w6,b6} device pointers (each stacked over L layers); reg_w: {w0,b0,w2,b2,w4,b4}
    MV2D_CHECK_ARG(outs && cls_w && reg_w && ref && cls && reg && pc_range && L > 0, "mv2d_heads_fused: null pointer");
    for (int i = 0; i < 10; ++i) MV2D_CHECK_ARG(cls_w[i] != nullptr, "mv2d_heads_fused: null cls weight");
    for (int i = 0; i < 6; ++i) MV2D_CHECK_ARG(reg_w[i] != nullptr, "mv2d_heads_fused: null reg weight");
    if (M == 0) return MV2D_OK;
    HeadsParams p{outs, cls_w[0], cls_w[1], cls_w[2], cls_w[3], cls_w[4], cls_w[5], cls_w[6], cls_w[7], cls_w[8], cls_w[9],
                  reg_w[0], reg_w[1], reg_w[2], reg_w[3], reg_w[4], reg_w[5], ref, cls, reg, M, L, eps,
                  pc_range[0], pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2], dt, dt_rows};
    hipLaunchKernelGGL(heads_fused_kernel, dim3(cdiv(M, 16), L, 2), dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_heads_fused_x3(const float* outs, const void* const* cls_w, const void* const* reg_w, const float* ref, float* cls, float* reg,
                                   int M, int L, float eps, const float* pc_range, float dt, const float* dt_rows, void* stream) {
    // cls_w: {w0_hi,w0_lo,b0,lnw1,lnb1,w3_hi,w3_lo,b3,lnw4,lnb4,w6,b6}; reg_w: {w0_hi,w0_lo,b0,w2_hi,w2_lo,b2,w4,b4} device pointers,
    // every tensor stacked over the L layers; the *_hi/_lo matrices are per-layer mv2d_split_bf16x2 + mv2d_pack_wfrag_bf16 copies
    MV2D_CHECK_ARG(outs && cls_w && reg_w && ref && cls && reg && pc_range && L > 0, "mv2d_heads_fused_x3: null pointer");
    for (int i = 0; i < 12; ++i) MV2D_CHECK_ARG(cls_w[i] != nullptr, "mv2d_heads_fused_x3: null cls weight");
    for (int i = 0; i < 8; ++i) MV2D_CHECK_ARG(reg_w[i] != nullptr, "mv2d_heads_fused_x3: null reg weight");
    if (M == 0) return MV2D_OK;
    typedef const unsigned short* U; typedef const float* Fp;
    HeadsX3Params p{outs, (U)cls_w[0], (U)cls_w[1], (Fp)cls_w[2], (Fp)cls_w[3], (Fp)cls_w[4], (U)cls_w[5], (U)cls_w[6], (Fp)cls_w[7], (Fp)cls_w[8],
                    (Fp)cls_w[9], (Fp)cls_w[10], (Fp)cls_w[11],
                    (U)reg_w[0], (U)reg_w[1], (Fp)reg_w[2], (U)reg_w[3], (U)reg_w[4], (Fp)reg_w[5], (Fp)reg_w[6], (Fp)reg_w[7],
                    ref, cls, reg, M, L, eps,
                    pc_range[0], pc_range[1], pc_range[2], pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2], dt, dt_rows};
    if (M <= 512) hipLaunchKernelGGL(heads_fused_x3_kernel<1>, dim3(cdiv(M, 16), L, 2), dim3(1024), 0, (hipStream_t)stream, p);
    else if (M <= 1024) hipLaunchKernelGGL(heads_fused_x3_kernel<2>, dim3(cdiv(M, 32), L, 2), dim3(1024), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(heads_fused_x3_kernel<4>, dim3(cdiv(M, 64), L, 2), dim3(1024), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_linear_x3_ex(const float* A, const float* A2, int n_split, int lda, const void* Whi, const void* Wlo, const float* bias,
                                 float* C, int ldc, int M, int N, int K, int act, float clamp, int groups, long long a_gs, long long w_gs,
                                 long long b_gs, long long c_gs, const int* m_dev, int conv3x3, const float* mul, const float* add, int ld_ma,
                                 void* stream);

extern "C" int mv2d_linear_x3(const float* A, const float* A2, int n_split, int lda, const void* Whi, const void* Wlo, const float* bias,
                              float* C, int ldc, int M, int N, int K, int act, float clamp, int groups, long long a_gs, long long w_gs,
                              long long b_gs, long long c_gs, void* stream) {
    return mv2d_linear_x3_ex(A, A2, n_split, lda, Whi, Wlo, bias, C, ldc, M, N, K, act, clamp, groups, a_gs, w_gs, b_gs, c_gs, nullptr, 0,
                             nullptr, nullptr, 0, stream);
}

extern "C" int mv2d_split_rows_key16(const float* a, const float* b, void* hi, void* lo, int M, int cols, const int* m_dev, void* stream) {
    MV2D_CHECK_ARG(a && hi && lo && M >= 0 && cols > 0 && (cols % 4) == 0, "mv2d_split_rows_key16: bad args (cols % 4 == 0)");
    MV2D_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)hi & 7) == 0 && ((uintptr_t)lo & 7) == 0,
                   "mv2d_split_rows_key16: operands must be 16-byte aligned");
    if (M == 0) return MV2D_OK;
    const long long n4 = (long long)M * (cols / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b,
                       (uint2*)hi, (uint2*)lo, n4, m_dev, cols / 4);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_linear_x3_ex(const float* A, const float* A2, int n_split, int lda, const void* Whi, const void* Wlo, const float* bias,
                                 float* C, int ldc, int M, int N, int K, int act, float clamp, int groups, long long a_gs, long long w_gs,
                                 long long b_gs, long long c_gs, const int* m_dev, int conv3x3, const float* mul, const float* add, int ld_ma,
                                 void* stream) {
    MV2D_CHECK_ARG(A && Whi && Wlo && C, "mv2d_linear_x3: null pointer");
    MV2D_CHECK_ARG(!conv3x3 || (K == 2304 && n_split == 0 && groups == 1), "mv2d_linear_x3_ex: conv3x3 reads [R,49,256] cells, K = 9 * 256");
    MV2D_CHECK_ARG((!mul && !add) || ld_ma >= N, "mv2d_linear_x3_ex: ld_ma < N");
    MV2D_CHECK_ARG(M >= 0 && N > 0 && (N % 16) == 0 && K > 0 && (K % 32) == 0, "mv2d_linear_x3: N % 16 == 0 and K % 32 == 0 required");
    MV2D_CHECK_ARG((lda % 4) == 0 && (conv3x3 || lda >= K) && ldc >= N && ((uintptr_t)A & 15) == 0, "mv2d_linear_x3: A rows must be 16-byte aligned");
    MV2D_CHECK_ARG(n_split == 0 || (A2 && (n_split % 128) == 0 && ((uintptr_t)A2 & 15) == 0), "mv2d_linear_x3: n_split must be a multiple of 128 with A2 set");
    MV2D_CHECK_ARG(groups >= 1 && (groups == 1 || ((a_gs % 4) == 0 && (w_gs % 8) == 0)), "mv2d_linear_x3: group strides must keep 16-byte alignment");
    if (M == 0) return MV2D_OK;
    LinX3Params p{A, A2, n_split, lda, (const unsigned short*)Whi, (const unsigned short*)Wlo, bias, C, ldc, M, N, K, act, clamp,
                  a_gs, w_gs, b_gs, c_gs, m_dev, conv3x3, mul, add, ld_ma};
    if (M <= 512) hipLaunchKernelGGL(linear_x3_kernel<1>, dim3(cdiv(N, 128), cdiv(M, 16), groups), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(linear_x3_kernel<2>, dim3(cdiv(N, 128), cdiv(M, 32), groups), dim3(512), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
