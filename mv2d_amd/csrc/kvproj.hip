// K/V in_proj of all decoder layers: C[M, N] = A[M,256] (bf16) x W[N,256]^T (bf16) + bias, bf16 out, layer-major blocks.
// (MU/petr_transformer.py:503-508 — the key / value halves of nn.MultiheadAttention's in_proj, 2 x 6 layers, N = 3072.)
//
// With K = 256 the generic tile kernel (gemm_bf16.hip) spends its time in per-tile prologues and epilogues: 2880 blocks each
// fetch 128 KB, run 8 k-steps and leave.  This kernel is built around the shape instead (gfx950, wave64):
//   * the A rows are the stationary operand and live in REGISTERS: a block owns 128 rows (4 waves as 2x2, 64 rows per wave =
//     4 row tiles x 8 k-steps x 16 B = 128 VGPRs per lane, loaded once, fragment shaped);
//   * the weights stream: 64 output columns x 256 k (32 KB) per step through a 2-stage LDS ring filled by the LDS-DMA
//     (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass); the copy of step t+1 is in flight during the MFMAs and the
//     stores of step t.  LDS image: row n = 512 B, 16-byte chunk c stored at c ^ (n & 15) — the DMA destination is lane-linear,
//     so the permutation is applied to the per-lane SOURCE address and again to the ds_read_b128 fragment reads (conflict free
//     for the 4 x 16 lane groups of ds_read_b128);
//   * a block walks `nr` columns (12 steps at nr = 768), so the pipeline stays full for ~12 tiles instead of 1;
//   * the MFMAs run swapped (D^T = W_tile . A_tile^T), so a lane ends up with 4 consecutive output COLUMNS of one row per tile;
//     the two column tiles of a wave take interleaved weight rows (tile j, row 4g+r -> column 8g + 4j + r), which makes that 8
//     consecutive columns = one 16-byte bf16 store per lane straight from the accumulators — no LDS staging in the epilogue
//     (64 B per row per wave; the two column waves complete the 128-B line).
// One raw s_barrier per step; a counted vmcnt before it covers this wave's DMA pieces (the younger stores stay in flight).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int KD = 256;                      // K extent (fixed)
constexpr int BM = 128, BN = 64;
constexpr int ROWB = KD * 2;                 // 512 B per weight row
constexpr int STAGE_BYTES = BN * ROWB;       // 32 KB
constexpr int NR_MAX = 1536;                 // most columns per block (bias slice in LDS)
constexpr int LDS_BYTES = 2 * STAGE_BYTES + NR_MAX * 4;

// LDS image of a weight stage: row n = 512 B, 16-byte chunk c at position c ^ key(n), key = row bits {0,1,3,4}: the 16 rows a
// fragment read touches (8*(fr>>2) + 4j + (fr&3)) get 16 distinct keys -> conflict-free ds_read_b128 lane groups.
__device__ __forceinline__ int swz_key(int row) { return (row & 3) | (((row >> 3) & 3) << 2); }

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct KvParams {
    const unsigned short* A; const unsigned short* A2; int n_split; int lda;
    const unsigned short* W; const float* bias;
    int M, N, nr; const int* m_dev;
    unsigned short* C; int ldc; long long c_blk_stride; int c_blk_cols;
};

union Frag { uint4 u; mfma_bf16x8 v; };

#ifdef MV2D_KV_TRACE
__device__ long long g_kv_trace[64 * 8 * 32 * 8];   // [block][wave][step][stamp]
#define KV_STAMP(k) do { if (blockIdx.x < 64 && lane == 0) g_kv_trace[((blockIdx.x * 8 + wave) * 32 + t) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define KV_STAMP(k)
#endif

// LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base, lds_base + 1 KB), lane linear.
// Issued from inline asm on purpose: hipcc treats a *visible* LDS-DMA as a pending LDS write that may alias every later
// ds_read and drains it (s_waitcnt vmcnt(0)) before the first fragment read — which serialises copy and MFMAs.  The copy is
// ordered by hand instead: counted vmcnt + s_barrier at the top of every step.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}

// NG wave groups of 4 waves (2x2) share one weight stage; group g owns rows [128g, 128g + 128) of the block's row tile.
// NG = 2 runs the groups half a step apart ("ping-pong"): between two barriers group 0 does MFMA(t) then epilogue(t),
// group 1 does epilogue(t-1) then MFMA(t) — one group's stores / VALU overlap the other's matrix work on every SIMD, and the
// weight DMA (whose per-piece issue cost is paid by the issuing wave) is split over 8 waves and feeds 256 rows instead of 128.
template <int NG>
__global__ __launch_bounds__(256 * NG, 2) void kvproj_kernel(KvParams p) {
    constexpr int BMT = BM * NG;                                   // rows per block
    constexpr int PW = 8 / NG;                                     // DMA pieces per wave per stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    int M = p.M;
    if (p.m_dev) { const int md = *p.m_dev; M = md < M ? md : M; }
    // XCD-aware order (block b runs on XCD b % 8): the column ranges of one row tile run on the same XCD back to back
    const int nranges = p.N / p.nr;
    const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
    const int range = q % nranges, m_tile = (q / nranges) * 8 + xcd;
    const int m0 = m_tile * BMT;
    if (m0 >= M) return;
    const int n_begin = range * p.nr, steps = p.nr / BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1, fr = lane & 15, fg = lane >> 4;
    const int mrow0 = m0 + grp * BM + wr * 64;                     // first row of this wave
    const unsigned short* Abase = (p.n_split > 0 && n_begin >= p.n_split) ? p.A2 : p.A;

    // ---- weight DMA: 32 wave pieces of 1 KB (2 rows) per stage, PW per wave
    const int dma_half = lane >> 5, dma_p = lane & 31;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * PW * 1024);
    auto issue = [&](int t) {
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int R = 2 * (wave * PW + j) + dma_half;
            const unsigned short* src = p.W + (long long)(n_begin + t * BN + R) * KD + ((dma_p ^ swz_key(R)) << 3);
            dma16(src, lds_wave + (t & 1) * STAGE_BYTES + j * 1024);
        }
    };
    issue(0);
    // bias of this block's columns -> LDS (no ordinary load may be waited for inside the loop: vmcnt is in order, waiting for a
    // load younger than the DMA pieces would drain them)
    float* bias_s = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);
    for (int c = tid; c < p.nr; c += 256 * NG) bias_s[c] = p.bias ? p.bias[n_begin + c] : 0.f;
    const bool full_tile = m0 + BMT <= M;                          // every store is issued by every wave

    // ---- stationary A fragments: lane (fr, fg) holds A[row 16i + fr][k = 32s + 8fg .. +7]
    Frag af[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = mrow0 + i * 16 + fr;
        row = row < M ? row : M - 1;
        const unsigned short* ap = Abase + (long long)row * p.lda + fg * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) af[i][s].u = *reinterpret_cast<const uint4*>(ap + s * 32);
    }

    // weight rows of this lane's two fragment reads: tile j, fragment row fr -> stage row wc*32 + 8*(fr>>2) + 4j + (fr&3)
    const int b_row = wc * 32 + 8 * (fr >> 2) + (fr & 3);
    const int b_row_off = b_row * ROWB;                            // swz_key(b_row) == swz_key(b_row + 4) == fr

    f32x4_t acc[4][2];
    auto mfma_phase = [&](int t) {
        const unsigned char* stage = smem + (t & 1) * STAGE_BYTES + b_row_off;
        const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int coff = ((s * 4 + fg) ^ fr) << 4;
            Frag b0, b1;
            b0.u = *reinterpret_cast<const uint4*>(stage + coff);
            b1.u = *reinterpret_cast<const uint4*>(stage + 4 * ROWB + coff);
#pragma unroll
            for (int i = 0; i < 4; ++i) {          // swapped: D[n][m] — lane (fr, fg) holds columns n = 4fg..4fg+3 of row m = fr
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0.v, af[i][s].v, s == 0 ? zero : acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1.v, af[i][s].v, s == 0 ? zero : acc[i][1], 0, 0, 0);
            }
        }
    };
    // epilogue of step t straight from the accumulators: 8 consecutive columns of one row per lane -> one 16-byte store.
    // c_blk_cols is a multiple of 64 (host-checked), so the output block of a step is wave uniform: no per-lane division.
    unsigned short* crow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) crow[i] = p.C + (long long)(mrow0 + i * 16 + fr) * p.ldc + wc * 32 + 8 * fg;
    auto epilogue = [&](int t) {
        const int col0 = n_begin + t * BN;                         // first column of the step (uniform)
        const int nb = p.c_blk_cols > 0 ? col0 / p.c_blk_cols : 0;
        const long long c_col = (long long)nb * p.c_blk_stride + (p.c_blk_cols > 0 ? col0 - nb * p.c_blk_cols : col0);
        const int lcol = t * BN + wc * 32 + 8 * fg;
        const float4 bl = *reinterpret_cast<const float4*>(bias_s + lcol);
        const float4 bh = *reinterpret_cast<const float4*>(bias_s + lcol + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (full_tile || mrow0 + i * 16 + fr < M) {
                *reinterpret_cast<uint4*>(crow[i] + c_col) =
                    make_uint4(pack_bf16x2(acc[i][0][0] + bl.x, acc[i][0][1] + bl.y), pack_bf16x2(acc[i][0][2] + bl.z, acc[i][0][3] + bl.w),
                               pack_bf16x2(acc[i][1][0] + bh.x, acc[i][1][1] + bh.y), pack_bf16x2(acc[i][1][2] + bh.z, acc[i][1][3] + bh.w));
            }
        }
    };

    for (int t = 0; t < steps; ++t) {
        // this wave's DMA pieces of step t have landed; the 4 stores issued after them (younger) may stay in flight —
        // group g issues its first stores in interval g, so the counted wait is only valid from t > g on
        KV_STAMP(0);
        if (full_tile && t > grp) __builtin_amdgcn_s_waitcnt(0x0F74); else __builtin_amdgcn_s_waitcnt(0x0F70);
        KV_STAMP(1);
        __builtin_amdgcn_s_barrier();                              // ... everybody's, and the other stage is free again
        asm volatile("" ::: "memory");
        KV_STAMP(2);
        if (t + 1 < steps) issue(t + 1);
        KV_STAMP(3);
        if (grp == 0) {
            mfma_phase(t);
#ifdef MV2D_KV_TRACE
            if (acc[3][1][3] == 1.2345e30f) KV_STAMP(5);
#endif
            KV_STAMP(4);
            epilogue(t);
            KV_STAMP(5);
        } else {
            if (t > 0) epilogue(t - 1);
            KV_STAMP(4);
            mfma_phase(t);
#ifdef MV2D_KV_TRACE
            if (acc[3][1][3] == 1.2345e30f) KV_STAMP(4);
#endif
            KV_STAMP(5);
        }
    }
    if (grp != 0) epilogue(steps - 1);
}

}  // namespace

// C-ABI: see include/mv2d_hip.h
extern "C" int mv2d_kv_proj(const void* A, const void* A2, int n_split, int lda, const void* W, const float* bias, int M, int N,
                            const int* m_dev, void* C, int ldc, long long c_blk_stride, int c_blk_cols, void* stream) {
    MV2D_CHECK_ARG(A && W && C, "mv2d_kv_proj: null A/W/C");
    MV2D_CHECK_ARG(M >= 0 && N > 0 && (N % 256) == 0, "mv2d_kv_proj: N must be a positive multiple of 256");
    MV2D_CHECK_ARG((lda % 8) == 0 && (ldc % 8) == 0 && (c_blk_stride % 8) == 0 && (c_blk_cols % 64) == 0,
                   "mv2d_kv_proj: strides must keep 16-byte alignment, c_blk_cols a multiple of 64");
    MV2D_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)A2 & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0,
                   "mv2d_kv_proj: operands must be 16-byte aligned");
    if (M == 0) return MV2D_OK;
    // columns per block: the largest of 768 / 512 / 256 that divides N and does not straddle n_split
    // (1536 once there are rows for several rounds of blocks: the A rows are fetched N / nr times; measured 233 -> 215 us at M = 88k,
    //  but 35 -> 45 us at M = 14.7k where the wider ranges leave CUs without a block)
    int nr = 256;
    for (int cand : {1536, 768, 512}) {
        if (cand == 1536 && M < 40000) continue;
        if ((N % cand) == 0 && (n_split == 0 || (n_split % cand) == 0)) { nr = cand; break; }
    }
    static const int nr_env = getenv("MV2D_KV_NR") ? atoi(getenv("MV2D_KV_NR")) : 0;
    if (nr_env) nr = nr_env;
    MV2D_CHECK_ARG(nr <= NR_MAX && (N % nr) == 0 && (nr % BN) == 0, "mv2d_kv_proj: columns per block must divide N, be a multiple of 64 and <= 1536");
    MV2D_CHECK_ARG(n_split == 0 || (A2 && (n_split % nr) == 0), "mv2d_kv_proj: n_split must be a multiple of 256 with A2 set");
    KvParams p;
    p.A = (const unsigned short*)A; p.A2 = (const unsigned short*)A2; p.n_split = n_split; p.lda = lda;
    p.W = (const unsigned short*)W; p.bias = bias; p.M = M; p.N = N; p.nr = nr; p.m_dev = m_dev;
    p.C = (unsigned short*)C; p.ldc = ldc; p.c_blk_stride = c_blk_stride; p.c_blk_cols = c_blk_cols;
    // two wave groups per block (256 rows, ping-pong) once that still gives every CU a block; else one group (128 rows)
    static const int ng_env = getenv("MV2D_KV_NG") ? atoi(getenv("MV2D_KV_NG")) : 0;
    const int nranges = N / nr;
    const int ng = ng_env ? ng_env : ((long long)cdiv(M, 2 * BM) * nranges >= 200 ? 2 : 1);
    const int m_tiles = cdiv(M, BM * ng);
    dim3 grid(((m_tiles + 7) / 8) * 8 * nranges);
    if (ng == 2) hipLaunchKernelGGL(kvproj_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(kvproj_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

#ifdef MV2D_KV_TRACE
extern "C" int mv2d_kv_trace_read(long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_kv_trace), n * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#endif
