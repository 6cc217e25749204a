// bf16 MFMA GEMM for the big-M dense ops of the MV2D hot path (gfx950 / CDNA4, wave64).
//
//   C[M,N] = epilogue( A[M,K] (bf16) x W[N,K]^T (bf16, nn.Linear layout) + bias )      fp32 accumulate
//
// Used for: PE MLPs (MU/pe.py:64-77,36-48), K/V projections of all 6 decoder layers at once
// (MU/petr_transformer.py:503-508 in_proj of key/value), QueryGenerator conv3x3 as implicit GEMM
// (RH/utils/query_generator.py:298-304).
//
// Tile 128x128x64, 256 threads = 4 waves as 2x2, each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// A/W tiles are register-staged (16 B per lane, one 128 B row per 8 lanes -> full-line coalesced reads)
// into an XOR-swizzled LDS image (byte ^= (row&7)<<4) so the ds_read_b128 fragment reads are <=2-way.
// ONE 32 KiB LDS stage (A 16 KiB + W 16 KiB): the next tile's global loads are issued into registers before the
// MFMAs of the current tile and written to LDS after them (T14 issue-early / write-late), so 3 blocks are
// resident per CU and cover each other's load latency (K is only 192..2304 here: 3..36 steps).
// bf16 outputs leave through the same LDS: each wave stages its 64x64 sub-tile (XOR-swizzled) and writes
// whole 128-byte row segments with 16-byte stores instead of 2-byte scattered stores.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int ROW_BYTES = BK * 2;              // 128 B per LDS row
constexpr int TILE_BYTES = BM * ROW_BYTES;     // 16 KiB

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
};

__device__ __forceinline__ int lds_off(int row, int slot) { return row * ROW_BYTES + ((slot ^ (row & 7)) << 4); }

__global__ __launch_bounds__(256, 3) void gemm_bf16_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* As = smem;
    unsigned char* Bs = smem + TILE_BYTES;

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

    // ---- per-thread staging assignment: 4 chunks of 16 B per operand
    int a_row[4], a_slot[4];
    long long a_src[4];            // element offset of the row start (plain) / roi base (conv)
    int c_py[4], c_px[4];
    bool a_ok[4];
    long long b_src[4];
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int row = c >> 3, slot = c & 7;
        a_row[i] = row; a_slot[i] = slot;
        int m = m0 + row;
        a_ok[i] = m < M;
        int mc = a_ok[i] ? m : (M - 1);
        if (p.a_mode == 0) {
            a_src[i] = (long long)mc * p.lda;
            c_py[i] = c_px[i] = 0;
        } else {
            int r = mc / 49, cell = mc - r * 49;
            c_py[i] = cell / 7; c_px[i] = cell - c_py[i] * 7;
            a_src[i] = (long long)r * 49 * 256;
        }
        int n = n0 + row;
        b_ok[i] = n < p.N;
        b_src[i] = (long long)(b_ok[i] ? n : (p.N - 1)) * p.K;
    }

    uint4 ra[4], rb[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned short* src;
            bool ok = a_ok[i];
            if (p.a_mode == 0) {
                src = Abase + a_src[i] + k0 + a_slot[i] * 8;
            } else {
                int tap = k0 >> 8, c0 = (k0 & 255) + a_slot[i] * 8;
                int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                int y = c_py[i] + dy, x = c_px[i] + dx;
                ok = ok && y >= 0 && y < 7 && x >= 0 && x < 7;
                int yy = ok ? y : 0, xx = ok ? x : 0;
                src = Abase + a_src[i] + (yy * 7 + xx) * 256 + c0;
            }
            uint4 v = *reinterpret_cast<const uint4*>(src);
            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
            ra[i] = v;
            uint4 w = *reinterpret_cast<const uint4*>(p.W + b_src[i] + k0 + a_slot[i] * 8);
            if (!b_ok[i]) w = make_uint4(0u, 0u, 0u, 0u);
            rb[i] = w;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int off = lds_off(a_row[i], a_slot[i]);
            *reinterpret_cast<uint4*>(As + off) = ra[i];
            *reinterpret_cast<uint4*>(Bs + off) = rb[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);          // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            mfma_bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wr * 64 + i * 16 + fr;
                af[i] = *reinterpret_cast<const mfma_bf16x8*>(As + lds_off(row, ks * 4 + fg));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = wc * 64 + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const mfma_bf16x8*>(Bs + lds_off(row, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                    // every wave is done reading this stage
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds C[m = .. + fg*4 + reg][n = .. + fr]
    // The bf16 result (C when c_bf16, else C2) is staged per wave in LDS (64 rows x 128 B, 32-B units XOR-swizzled by
    // the 4-row group so the four lane groups of an MFMA column store land on different banks) and written out as
    // whole 128-byte row segments.  fp32 results are stored directly (64-byte segments per lane group).
    unsigned short* stage = reinterpret_cast<unsigned short*>(smem + wave * 8192);
    const bool stage_c = p.C && p.c_bf16, stage_c2 = p.C2 != nullptr;
    const bool staged = (stage_c || stage_c2) && ((p.N & 63) == 0) && (p.c_blk_cols == 0 || (p.c_blk_cols & 63) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + fr;
        const bool n_ok = n < p.N;
        const float bn = (p.bias && n_ok) ? p.bias[n] : 0.f;
        const int nb = p.c_blk_cols > 0 ? n / p.c_blk_cols : 0;
        const int nc = p.c_blk_cols > 0 ? n - nb * p.c_blk_cols : n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = i * 16 + fg * 4 + r;                   // row inside the wave's 64x64 sub-tile
                const int m = m0 + wr * 64 + lrow;
                const bool ok = n_ok && m < M;
                float v = acc[i][j][r] + bn;
                if (ok && p.mul) v *= p.mul[(long long)m * p.ldmul + n];
                if (ok && p.add) v += p.add[(long long)m * p.ldadd + n];
                if (p.act == 1) v = fmaxf(v, 0.f);
                else if (p.act == 2) v = 1.f / (1.f + __expf(-v));
                float v2 = v;
                if (ok && p.C2 && p.add2) v2 = v + p.add2[(long long)m * p.ldadd2 + n];
                if (ok && p.C && !p.c_bf16) {
                    long long o = (long long)nb * p.c_blk_stride + (long long)m * p.ldc + nc;
                    reinterpret_cast<float*>(p.C)[o] = v;
                }
                if (staged) {
                    const int lcol = j * 16 + fr;
                    const int byte = lrow * 128 + ((lcol * 2) ^ (((lrow >> 2) & 3) << 5));
                    stage[byte >> 1] = f32_to_bf16(stage_c2 ? v2 : v);
                } else if (ok) {
                    if (stage_c) {
                        long long o = (long long)nb * p.c_blk_stride + (long long)m * p.ldc + nc;
                        reinterpret_cast<unsigned short*>(p.C)[o] = f32_to_bf16(v);
                    }
                    if (stage_c2) p.C2[(long long)m * p.ldc2 + n] = f32_to_bf16(v2);
                }
            }
        }
    }
    if (staged) {
        // wave-local hand-off through LDS (each wave reads back only its own 8 KiB): LDS ops of one wave are ordered
        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
        const int nsub = n0 + wc * 64;          // first column of this wave's sub-tile
        unsigned short* dst;
        long long ldd;
        if (stage_c2) { dst = p.C2 + nsub; ldd = p.ldc2; }
        else {
            const int nb = p.c_blk_cols > 0 ? nsub / p.c_blk_cols : 0;
            const int nc = p.c_blk_cols > 0 ? nsub - nb * p.c_blk_cols : nsub;
            dst = reinterpret_cast<unsigned short*>(p.C) + (long long)nb * p.c_blk_stride + nc;
            ldd = p.ldc;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int lrow = it * 8 + (lane >> 3), chunk = lane & 7;
            const int m = m0 + wr * 64 + lrow;
            const int byte = lrow * 128 + ((chunk * 16) ^ (((lrow >> 2) & 3) << 5));
            const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(stage) + byte);
            if (m < M) *reinterpret_cast<uint4*>(dst + (long long)m * ldd + chunk * 8) = v;
        }
    }
}

}  // namespace

// C-ABI: see include/mv2d_hip.h
extern "C" int mv2d_gemm_bf16(const void* A, const void* A2, int n_split, int a_mode, const void* W,
                              const float* bias, int M, int N, int K, int lda, const int* m_dev, int act,
                              const float* mul, int ldmul, const float* add, int ldadd, void* C, int c_bf16,
                              int ldc, long long c_blk_stride, int c_blk_cols, void* C2, const float* add2,
                              int ldc2, int ldadd2, void* stream) {
    MV2D_CHECK_ARG(A && W && (C || C2), "mv2d_gemm_bf16: null A/W/C");
    MV2D_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % BK) == 0, "mv2d_gemm_bf16: K must be a positive multiple of 64");
    MV2D_CHECK_ARG(a_mode == 0 || (a_mode == 1 && K == 9 * 256 && (M % 49) == 0), "mv2d_gemm_bf16: conv3x3 mode needs K=2304, M=R*49");
    MV2D_CHECK_ARG(a_mode == 1 || (lda % 8) == 0, "mv2d_gemm_bf16: lda must be a multiple of 8 (16-byte rows)");
    MV2D_CHECK_ARG(n_split == 0 || (A2 && (n_split % BN) == 0), "mv2d_gemm_bf16: n_split must be a multiple of 128 with A2 set");
    MV2D_CHECK_ARG(!(C2 && C && c_bf16), "mv2d_gemm_bf16: with a second (bf16) output the primary output must be fp32");
    MV2D_CHECK_ARG(((uintptr_t)C2 & 15) == 0 && (!c_bf16 || ((uintptr_t)C & 15) == 0) && (ldc2 % 8) == 0 && (!c_bf16 || (ldc % 8) == 0),
                   "mv2d_gemm_bf16: bf16 outputs need 16-byte aligned base and row stride");
    if (M == 0) return MV2D_OK;
    Params p;
    p.A = (const unsigned short*)A; p.A2 = (const unsigned short*)A2; p.W = (const unsigned short*)W; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.m_dev = m_dev; p.a_mode = a_mode; p.n_split = n_split; p.act = act;
    p.mul = mul; p.ldmul = ldmul; p.add = add; p.ldadd = ldadd; p.C = C; p.c_bf16 = c_bf16; p.ldc = ldc;
    p.c_blk_stride = c_blk_stride; p.c_blk_cols = c_blk_cols; p.C2 = (unsigned short*)C2; p.add2 = add2;
    p.ldc2 = ldc2; p.ldadd2 = ldadd2;
    p.n_tiles = cdiv(N, BN);
    dim3 grid(((cdiv(M, BM) + 7) / 8) * 8 * p.n_tiles);
    hipLaunchKernelGGL(gemm_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
