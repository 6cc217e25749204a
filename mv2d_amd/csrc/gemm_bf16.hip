// bf16 MFMA GEMM for the big-M dense ops of the MV2D hot path (gfx950 / CDNA4, wave64).
//
//   C[M,N] = epilogue( A[M,K] (bf16) x W[N,K]^T (bf16, nn.Linear layout) + bias )      fp32 accumulate
//
// Used for: PE MLPs (MU/pe.py:64-77,36-48), K/V projections of all 6 decoder layers at once
// (MU/petr_transformer.py:503-508 in_proj of key/value), QueryGenerator conv3x3 as implicit GEMM
// (RH/utils/query_generator.py:298-304).
//
// Tile BMxBNx64 (128x128 or 64x64), 256 threads = 4 waves as 2x2, v_mfma_f32_16x16x32_bf16.
// A/W tiles are register-staged (16 B per lane, one 128 B row per 8 lanes -> full-line coalesced reads)
// into an XOR-swizzled LDS image (byte ^= (row&7)<<4) so the ds_read_b128 fragment reads are <=2-way.
// ONE LDS stage: the next tile's global loads are issued into registers before the MFMAs of the current
// tile and written to LDS after them (issue-early / write-late), several blocks resident per CU cover each
// other's load latency (K is only 192..2304 here: 3..36 steps).  The 64x64 variant exists because the N=256
// GEMMs over ~9-15k rows would otherwise be < 256 blocks (one per CU, nothing to overlap with).
// Epilogue (round-1 profile: 2-byte scattered stores + per-element address math cost more than the K loop at
// K=256): every 16-row MFMA slab is staged per wave through padded fp32 LDS and leaves as whole rows — 8
// consecutive columns per lane, float4 reads of the fused mul/add operands, 16-byte bf16 / 2x16-byte fp32 stores.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK_MIN = 64;

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

struct Params {
    const unsigned short* A;       // bf16 [M, lda]  (plain)  |  [R, 49, 256] (conv3x3 mode)
    const unsigned short* A2;      // optional: columns n >= n_split read A2 instead of A (same lda)
    const unsigned short* W;       // bf16 [N, K]
    const float* bias;             // [N] or null
    int M, N, K, lda;
    int n_tiles;
    const int* m_dev;              // optional device-side row count (<= M); tiles beyond it exit
    int a_mode;                    // 0 plain, 1 conv3x3 over [R,49,256]
    int n_split;
    int act;                       // 0 none, 1 relu, 2 sigmoid
    const float* mul; int ldmul;   // v *= mul[m, n]
    const float* add; int ldadd;   // v += add[m, n]
    void* C; int c_bf16; int ldc;  // primary output
    long long c_blk_stride; int c_blk_cols;   // out = C + (n / c_blk_cols) * c_blk_stride + m * ldc + n % c_blk_cols
    unsigned short* C2; const float* add2; int ldc2; int ldadd2;   // C2 = bf16(v + add2[m, n])
    const int* add_idx; int add_period;   // optional: the add operand's row of output row m is add_idx[m] % add_period (a gathered table)
    int k_splits; long long c_split_stride;   // split-K: blockIdx.y = split s of the K range, output slab C + s * c_split_stride (fp32, no act; bias in slab 0)
    int c_split3;                  // bf16 output as the split-precision operand of a following GEMM: [hi | lo | hi] in column blocks of N (ldc >= 3 N)
};

// LDS image: one tile row = BK bf16 (128 or 256 B) in 16-byte slots, slot index XOR-ed with the low row bits
template <int BK>
__device__ __forceinline__ int lds_off(int row, int slot) { return row * (BK * 2) + ((slot ^ (row & (BK / 8 - 1))) << 4); }

template <int BM, int BN, int BK>
struct Cfg {
    static constexpr int ROW_BYTES = BK * 2;
    static constexpr int SPR = BK / 8;                        // 16-byte slots per row
    static constexpr int WM = BM / 2, WN = BN / 2;            // wave tile
    static constexpr int TI = WM / 16, TJ = WN / 16;          // MFMA tiles per wave
    static constexpr int CA = BM * SPR / 256, CB = BN * SPR / 256;   // 16-byte staging chunks per thread
    static constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    static constexpr int SLD = WN + 4;                        // padded fp32 row of the epilogue stage
    static constexpr int STAGE_BYTES = 16 * SLD * 4;          // one 16-row slab per wave
    static constexpr int LDS_BYTES = (A_BYTES + B_BYTES) > 4 * STAGE_BYTES ? (A_BYTES + B_BYTES) : 4 * STAGE_BYTES;
    static constexpr int CPR = WN / 8;                        // 8-column chunks per slab row
    static constexpr int EIT = 16 * CPR / 64;                 // epilogue chunks per lane per slab (2 or 1)
};

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256, (BM == 128 ? 3 : 4)) void gemm_bf16_kernel(Params p) {
    using G = Cfg<BM, BN, BK>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
    unsigned char* As = smem;
    unsigned char* Bs = smem + G::A_BYTES;

    int M = p.M;
    if (p.m_dev) { int md = *p.m_dev; M = md < M ? md : M; }
    // XCD-aware tile order (block b runs on XCD b % 8): the 8 XCDs take 8 consecutive M tiles and each
    // walks all N tiles of ITS M tile back to back, so an A tile is fetched into one L2 only.
    const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
    const int n_tile = q % p.n_tiles, m_tile = (q / p.n_tiles) * 8 + xcd;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const unsigned short* Abase = (p.n_split > 0 && n0 >= p.n_split) ? p.A2 : p.A;

    // ---- per-thread staging assignment
    int a_off[G::CA], b_off[G::CB];
    long long a_src[G::CA], b_src[G::CB];
    int c_py[G::CA], c_px[G::CA];
    bool a_ok[G::CA], b_ok[G::CB];
#pragma unroll
    for (int i = 0; i < G::CA; ++i) {
        const int c = tid + 256 * i, row = c / G::SPR, slot = c % G::SPR;
        a_off[i] = lds_off<BK>(row, slot);
        const int m = m0 + row;
        a_ok[i] = m < M;
        const int mc = a_ok[i] ? m : (M - 1);
        if (p.a_mode == 0) {
            a_src[i] = (long long)mc * p.lda + slot * 8;
            c_py[i] = c_px[i] = 0;
        } else {
            const int r = mc / 49, cell = mc - r * 49;
            c_py[i] = cell / 7; c_px[i] = cell - c_py[i] * 7;
            a_src[i] = (long long)r * 49 * p.lda + slot * 8;          // conv mode: lda = channels per cell (256, or 768 for [hi | lo | hi] cells)
        }
    }
#pragma unroll
    for (int i = 0; i < G::CB; ++i) {
        const int c = tid + 256 * i, row = c / G::SPR, slot = c % G::SPR;
        b_off[i] = lds_off<BK>(row, slot);
        const int n = n0 + row;
        b_ok[i] = n < p.N;
        b_src[i] = (long long)(b_ok[i] ? n : (p.N - 1)) * p.K + slot * 8;
    }

    uint4 ra[G::CA], rb[G::CB];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < G::CA; ++i) {
            const unsigned short* src;
            bool ok = a_ok[i];
            if (p.a_mode == 0) {
                src = Abase + a_src[i] + k0;
            } else {
                const int tap = k0 / p.lda, c0 = k0 - tap * p.lda;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int y = c_py[i] + dy, x = c_px[i] + dx;
                ok = ok && y >= 0 && y < 7 && x >= 0 && x < 7;
                const int yy = ok ? y : 0, xx = ok ? x : 0;
                src = Abase + a_src[i] + (yy * 7 + xx) * p.lda + c0;
            }
            uint4 v = *reinterpret_cast<const uint4*>(src);
            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < G::CB; ++i) {
            uint4 w = *reinterpret_cast<const uint4*>(p.W + b_src[i] + k0);
            if (!b_ok[i]) w = make_uint4(0u, 0u, 0u, 0u);
            rb[i] = w;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < G::CA; ++i) *reinterpret_cast<uint4*>(As + a_off[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < G::CB; ++i) *reinterpret_cast<uint4*>(Bs + b_off[i]) = rb[i];
    };

    f32x4_t acc[G::TI][G::TJ];
#pragma unroll
    for (int i = 0; i < G::TI; ++i)
#pragma unroll
        for (int j = 0; j < G::TJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // split-K (blockIdx.y): this block's range of k tiles; a split without tiles writes zeros
    const int nk_all = p.K / BK, per = (nk_all + p.k_splits - 1) / p.k_splits;
    const int kt0 = blockIdx.y * per, nk = min(nk_all, kt0 + per);
    if (kt0 < nk) {
        load_tile(kt0 * BK);
        store_tile();
    }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = kt0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);          // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            mfma_bf16x8 af[G::TI], bfr[G::TJ];
#pragma unroll
            for (int i = 0; i < G::TI; ++i)
                af[i] = *reinterpret_cast<const mfma_bf16x8*>(As + lds_off<BK>(wr * G::WM + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int j = 0; j < G::TJ; ++j)
                bfr[j] = *reinterpret_cast<const mfma_bf16x8*>(Bs + lds_off<BK>(wc * G::WN + j * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < G::TI; ++i)
#pragma unroll
                for (int j = 0; j < G::TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                    // every wave is done reading this stage
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue.  MFMA layout: lane holds C[row = 4*fg + reg][col = fr] of each 16x16 tile.
    // Per 16-row slab i: all TJ tiles go to the wave's private padded fp32 stage, then each lane takes CPR-aligned
    // chunks of 8 consecutive columns of one row and applies bias / mul / add / act / casts with vector accesses.
    float* stage = reinterpret_cast<float*>(smem + wave * G::STAGE_BYTES);
    const int chunk = lane % G::CPR;                        // column chunk of this lane (same in every iteration)
    const int ncol = n0 + wc * G::WN + chunk * 8;           // first of its 8 global columns
    const bool col_ok = ncol < p.N;                         // N is a multiple of 8 (host-checked)
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (p.bias && col_ok && blockIdx.y == 0) ? p.bias[ncol + e] : 0.f;
    const int nb = p.c_blk_cols > 0 ? ncol / p.c_blk_cols : 0;
    const long long c_col = (long long)nb * p.c_blk_stride + (p.c_blk_cols > 0 ? ncol - nb * p.c_blk_cols : ncol);
#pragma unroll
    for (int i = 0; i < G::TI; ++i) {
#pragma unroll
        for (int j = 0; j < G::TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(fg * 4 + r) * G::SLD + j * 16 + fr] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): own LDS writes landed (wave-private region)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < G::EIT; ++it) {
            const int lrow = (it * 64 + lane) / G::CPR;
            const int m = m0 + wr * G::WM + i * 16 + lrow;
            const float4 s0 = *reinterpret_cast<const float4*>(stage + lrow * G::SLD + chunk * 8);
            const float4 s1 = *reinterpret_cast<const float4*>(stage + lrow * G::SLD + chunk * 8 + 4);
            if (m < M && col_ok) {
                float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
                if (p.mul) {
                    const float4 a0 = *reinterpret_cast<const float4*>(p.mul + (long long)m * p.ldmul + ncol);
                    const float4 a1 = *reinterpret_cast<const float4*>(p.mul + (long long)m * p.ldmul + ncol + 4);
                    v[0] *= a0.x; v[1] *= a0.y; v[2] *= a0.z; v[3] *= a0.w; v[4] *= a1.x; v[5] *= a1.y; v[6] *= a1.z; v[7] *= a1.w;
                }
                if (p.add) {
                    const long long ar = p.add_idx ? (long long)(p.add_idx[m] % p.add_period) : (long long)m;
                    const float4 a0 = *reinterpret_cast<const float4*>(p.add + ar * p.ldadd + ncol);
                    const float4 a1 = *reinterpret_cast<const float4*>(p.add + ar * p.ldadd + ncol + 4);
                    v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
                } else if (p.act == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
                }
                if (p.C) {
                    const long long o = c_col + (long long)m * p.ldc;
                    if (p.c_split3) {
                        unsigned int h[4], l[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            h[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                            l[e] = pack_bf16x2(v[2 * e] - __uint_as_float(h[e] << 16), v[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u));
                        }
                        unsigned short* cp = reinterpret_cast<unsigned short*>(p.C) + o;
                        *reinterpret_cast<uint4*>(cp) = make_uint4(h[0], h[1], h[2], h[3]);
                        *reinterpret_cast<uint4*>(cp + p.N) = make_uint4(l[0], l[1], l[2], l[3]);
                        *reinterpret_cast<uint4*>(cp + 2 * p.N) = make_uint4(h[0], h[1], h[2], h[3]);
                    } else if (p.c_bf16) {
                        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p.C) + o) =
                            make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                    } else {
                        float* cp = reinterpret_cast<float*>(p.C) + o + (long long)blockIdx.y * p.c_split_stride;
                        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
                if (p.C2) {
                    if (p.add2) {
                        const float4 a0 = *reinterpret_cast<const float4*>(p.add2 + (long long)m * p.ldadd2 + ncol);
                        const float4 a1 = *reinterpret_cast<const float4*>(p.add2 + (long long)m * p.ldadd2 + ncol + 4);
                        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
                    }
                    *reinterpret_cast<uint4*>(p.C2 + (long long)m * p.ldc2 + ncol) =
                        make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // own LDS reads retired before the next slab overwrites the stage
        asm volatile("" ::: "memory");
    }
}

// (hi | lo | hi) bf16 row of a (+ b): the A operand of a split-precision product through the plain bf16 GEMM,
//   [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo]^T = a_hi w_hi + a_lo w_hi + a_hi w_lo   (K' = 3 K, fp32 accumulation in the MFMA accumulators)
__global__ void split3_rows_kernel(const float4* __restrict__ a, const float4* __restrict__ b, unsigned short* __restrict__ out, long long n4,
                                   const int* __restrict__ m_dev, int row4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4 || (m_dev && i >= (long long)*m_dev * row4)) return;
    float4 v = a[i];
    if (b) { const float4 w = b[i]; v = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w); }
    const unsigned int h0 = pack_bf16x2(v.x, v.y), h1 = pack_bf16x2(v.z, v.w);
    const unsigned int l0 = pack_bf16x2(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
    const unsigned int l1 = pack_bf16x2(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
    const long long row = i / row4, c4 = i - row * row4;
    unsigned short* o = out + row * (12LL * row4) + 4 * c4;
    *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(o + 4 * row4) = make_uint2(l0, l1);
    *reinterpret_cast<uint2*>(o + 8 * row4) = make_uint2(h0, h1);
}

}  // namespace

extern "C" int mv2d_split3_rows(const float* a, const float* b, void* out, int M, int cols, const int* m_dev, void* stream) {
    MV2D_CHECK_ARG(a && out && M >= 0 && cols > 0 && (cols % 4) == 0, "mv2d_split3_rows: bad args (cols % 4 == 0)");
    MV2D_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)out & 7) == 0, "mv2d_split3_rows: operands must be 16-byte aligned");
    if (M == 0) return MV2D_OK;
    const long long n4 = (long long)M * (cols / 4);
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b,
                       (unsigned short*)out, n4, m_dev, cols / 4);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_gemm_bf16_ex(const void* A, const void* A2, int n_split, int a_mode, const void* W,
                                 const float* bias, int M, int N, int K, int lda, const int* m_dev, int act,
                                 const float* mul, int ldmul, const float* add, int ldadd, void* C, int c_bf16,
                                 int ldc, long long c_blk_stride, int c_blk_cols, void* C2, const float* add2,
                                 int ldc2, int ldadd2, int c_split3, const int* add_idx, int add_period, int k_splits, long long c_split_stride,
                                 void* stream);

// C-ABI: see include/mv2d_hip.h
extern "C" int mv2d_gemm_bf16(const void* A, const void* A2, int n_split, int a_mode, const void* W,
                              const float* bias, int M, int N, int K, int lda, const int* m_dev, int act,
                              const float* mul, int ldmul, const float* add, int ldadd, void* C, int c_bf16,
                              int ldc, long long c_blk_stride, int c_blk_cols, void* C2, const float* add2,
                              int ldc2, int ldadd2, void* stream) {
    return mv2d_gemm_bf16_ex(A, A2, n_split, a_mode, W, bias, M, N, K, a_mode == 1 ? 256 : lda, m_dev, act, mul, ldmul, add, ldadd, C, c_bf16, ldc,
                             c_blk_stride, c_blk_cols, C2, add2, ldc2, ldadd2, 0, nullptr, 0, 1, 0, stream);
}

extern "C" int mv2d_gemm_bf16_ex(const void* A, const void* A2, int n_split, int a_mode, const void* W,
                                 const float* bias, int M, int N, int K, int lda, const int* m_dev, int act,
                                 const float* mul, int ldmul, const float* add, int ldadd, void* C, int c_bf16,
                                 int ldc, long long c_blk_stride, int c_blk_cols, void* C2, const float* add2,
                                 int ldc2, int ldadd2, int c_split3, const int* add_idx, int add_period, int k_splits, long long c_split_stride,
                                 void* stream) {
    MV2D_CHECK_ARG(A && W && (C || C2), "mv2d_gemm_bf16: null A/W/C");
    MV2D_CHECK_ARG(!c_split3 || (C && c_bf16 && !C2 && c_blk_cols == 0 && ldc >= 3 * N), "mv2d_gemm_bf16_ex: c_split3 writes bf16 [hi | lo | hi], ldc >= 3 N");
    MV2D_CHECK_ARG(k_splits >= 1 && (k_splits == 1 || (C && !c_bf16 && !C2 && act == 0 && !mul && !add && c_blk_cols == 0 && (c_split_stride % 4) == 0)),
                   "mv2d_gemm_bf16_ex: split-K writes plain fp32 partial slabs (no activation / fused operands)");
    MV2D_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % BK_MIN) == 0, "mv2d_gemm_bf16: K must be a positive multiple of 64");
    MV2D_CHECK_ARG((N % 8) == 0, "mv2d_gemm_bf16: N must be a multiple of 8");
    MV2D_CHECK_ARG(a_mode == 0 || (a_mode == 1 && (lda % 128) == 0 && K == 9 * lda && (M % 49) == 0), "mv2d_gemm_bf16: conv3x3 mode needs K = 9 * lda (channels per cell, a multiple of 128), M=R*49");
    MV2D_CHECK_ARG((lda % 8) == 0, "mv2d_gemm_bf16: lda must be a multiple of 8 (16-byte rows)");
    MV2D_CHECK_ARG(n_split == 0 || (A2 && (n_split % 128) == 0), "mv2d_gemm_bf16: n_split must be a multiple of 128 with A2 set");
    MV2D_CHECK_ARG(!(C2 && C && c_bf16), "mv2d_gemm_bf16: with a second (bf16) output the primary output must be fp32");
    MV2D_CHECK_ARG(c_blk_cols == 0 || (c_blk_cols % 8) == 0, "mv2d_gemm_bf16: c_blk_cols must be a multiple of 8");
    MV2D_CHECK_ARG(((uintptr_t)C & 15) == 0 && ((uintptr_t)C2 & 15) == 0 && ((uintptr_t)mul & 15) == 0 && ((uintptr_t)add & 15) == 0 &&
                       ((uintptr_t)add2 & 15) == 0,
                   "mv2d_gemm_bf16: outputs and fused operands must be 16-byte aligned");
    MV2D_CHECK_ARG((ldc % (c_bf16 ? 8 : 4)) == 0 && (ldc2 % 8) == 0 && (ldmul % 4) == 0 && (ldadd % 4) == 0 && (ldadd2 % 4) == 0 &&
                       (c_blk_stride % 8) == 0,
                   "mv2d_gemm_bf16: row strides must keep 16-byte alignment");
    if (M == 0) return MV2D_OK;
    Params p;
    p.A = (const unsigned short*)A; p.A2 = (const unsigned short*)A2; p.W = (const unsigned short*)W; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.m_dev = m_dev; p.a_mode = a_mode; p.n_split = n_split; p.act = act;
    p.mul = mul; p.ldmul = ldmul; p.add = add; p.ldadd = ldadd; p.C = C; p.c_bf16 = c_bf16; p.ldc = ldc;
    p.c_blk_stride = c_blk_stride; p.c_blk_cols = c_blk_cols; p.C2 = (unsigned short*)C2; p.add2 = add2;
    p.ldc2 = ldc2; p.ldadd2 = ldadd2; p.c_split3 = c_split3; p.add_idx = add_idx; p.add_period = add_idx ? (add_period > 0 ? add_period : 1) : 0;
    p.k_splits = k_splits; p.c_split_stride = c_split_stride;
    // tile choice: 128x128 when that already gives >= 3 blocks per CU, else 64x64 (4x the blocks)
    const long long big_blocks = (long long)cdiv(M, 128) * cdiv(N, 128) * k_splits;
    if (big_blocks >= 768) {
        p.n_tiles = cdiv(N, 128);
        dim3 grid(((cdiv(M, 128) + 7) / 8) * 8 * p.n_tiles, k_splits);
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        p.n_tiles = cdiv(N, 64);
        dim3 grid(((cdiv(M, 64) + 7) / 8) * 8 * p.n_tiles, k_splits);
        // deep K: 128-wide k tiles (twice the bytes in flight per block, half the barriers)
        if ((K % 128) == 0 && K >= 512) hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, 128>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
