// Cross attention of the MV2D decoder on MFMA tiles, fused with the key / value projections (gfx950 / CDNA4, wave64).
//
// Replaces  kvproj_kernel (K/V of six layers written to HBM) + sparse_xattn_kernel (one VALU dot product per allowed pair)
// for PETRMultiheadAttention's core (MU/petr_transformer.py:426-513, nn.MultiheadAttention with an attn_mask).
//
// The in_proj of the keys / values is folded into the QUERY side, so the key side of every layer and head reads the same two
// key16 row arrays (common.h: fp16 since round 4; Xk = feat + pe, Xv = feat at the gathered positions / RoI cells) and nothing per layer is
// written for the keys:
//
//   logit_h[j] = q_h . (Wk_h x_j + bk_h)      = (Wk_h^T q_h) . x_j + const_h      (const_h cancels in the softmax over j)
//   ctx_h      = sum_j p_hj (Wv_h v_j + bv_h) = Wv_h (sum_j p_hj v_j) + bv_h      (sum_j p_hj = 1)
//
// Three launches per layer:
//   xattn_qmap_kernel    Qt[r] = (Wk_h^T q_h)_h as a 16 x 256 key16 MFMA operand per query: rows 0-7 = the "hi" parts of the
//                        eight heads, rows 8-15 = the "lo" remainders (fp32-class query side), stored fragment-major.  (The map
//                        itself is a bf16x3 product on the fp32 query; only its RESULT is split into the key-side format.)
//   xattn_tile_kernel    one block per query, the keys of its CSR row dealt to the waves in tiles of 16: the tile's Xk rows are
//                        gathered with whole-row (512 B) coalesced loads into an XOR-swizzled LDS tile, logits S[16 x 16] =
//                        Qt . Xk_tile^T on v_mfma_f32_16x16x32_f16 (hi and lo rows of Qt in one instruction: the sum of the
//                        two row groups is the fp32-class logit), online softmax per head, P split hi / lo into the same 16
//                        operand rows, z += P . Xv_tile on v_mfma_f32_16x16x16_f16 with the Xv rows loaded row-contiguous
//                        (16 B per lane) and transposed in registers (v_perm) into key-major B fragments.
//   xattn_ctxmap_kernel  ctx = Wv_h z_h + bv (bf16x3), empty CSR rows -> NaN (the reference's behaviour) or 0.
#include "common.h"

namespace {

constexpr int C = 256, HEADS = 8;
constexpr float LOG2E = 1.4426950408889634f;

typedef q16x8_t xt_bf16x8;      // the maps run in the query side's split format (common.h "q16": fp16 pairs since round 5)
union XtFrag { uint4 u; xt_bf16x8 v; };

__device__ __forceinline__ void xt_split8(const float4& x0, const float4& x1, XtFrag& hi, XtFrag& lo) {
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_q16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
    hi.u = make_uint4(h[0], h[1], h[2], h[3]);
    lo.u = make_uint4(l[0], l[1], l[2], l[3]);
}

// fp32 x 8 -> key16 hi / lo fragments (the format the tile kernel's MFMAs read)
__device__ __forceinline__ void xt_split8_k16(const float4& x0, const float4& x1, XtFrag& hi, XtFrag& lo) {
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_k16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
    hi.u = make_uint4(h[0], h[1], h[2], h[3]);
    lo.u = make_uint4(l[0], l[1], l[2], l[3]);
}

// ------------------------------------------------------------------------------------------------
// query map: Qt[r][h][s][g][part][e] = part( sum_d q[r][32 h + d] Wk[32 h + d][32 s + 8 g + e] ), part = hi / lo  (operand row n of the
// tile kernel = head n & 7, part n >> 3; one (query, head) = 1 KB contiguous: a wave of this kernel fills whole lines, a fragment load of
// the tile kernel reads 8 whole lines)
// Block = 16 queries, wave = head.  "Swapped" product D[channel][query] so that a lane ends up with 8 CONSECUTIVE channels of one
// query (rows 4g..4g+3 of two channel tiles whose row -> channel assignment is interleaved by the weight packing) = exactly one
// 16-byte chunk of the operand the tile kernel reads.
//   WA_hi / WA_lo [8 heads][16 tiles][64 lanes][8]: lane (m = l & 15, g = l >> 4), e: Wk[32 h + 8 g + e][chan(t, m)],
//   chan(t, m) = 32 (t >> 1) + 8 (m >> 2) + 4 (t & 1) + (m & 3)                                 (ops.pack_xattn_maps)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void xattn_qmap_kernel(const float* __restrict__ q, const uint4* __restrict__ WA_hi,
                                                         const uint4* __restrict__ WA_lo, uint4* __restrict__ Qt, int R) {
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 16;
    const int row = min(q0 + n, R - 1);
    const float* qp = q + (long long)row * C + 32 * h + 8 * g;
    XtFrag bh, bl;
    xt_split8(*reinterpret_cast<const float4*>(qp), *reinterpret_cast<const float4*>(qp + 4), bh, bl);
    const uint4* wh = WA_hi + (long long)h * 16 * 64 + lane;
    const uint4* wl = WA_lo + (long long)h * 16 * 64 + lane;
    f32x4_t acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        XtFrag ah, al;
        ah.u = wh[t * 64];
        al.u = wl[t * 64];
        f32x4_t a = {0.f, 0.f, 0.f, 0.f};
        a = mfma_q16_16x16x32(al.v, bh.v, a, 0, 0, 0);
        a = mfma_q16_16x16x32(ah.v, bl.v, a, 0, 0, 0);
        a = mfma_q16_16x16x32(ah.v, bh.v, a, 0, 0, 0);
        acc[t] = a;
    }
    // the wave's 16 rows x 1 KB go through LDS (two halves of 8 rows) so that every store instruction writes ONE row's 1 KB contiguously: the MFMA
    // leaves a lane 32-byte pieces 128 bytes apart (round 3: 18.2 -> 15.3 us for 4800 rows; the kernel is bound by its 39 MB of output, 6.2 us without)
    __shared__ uint4 stage[8][8 * 64];                           // [wave][8 rows x 64 chunks of 16 B], chunk index XOR-ed with the row
    uint4* st = stage[h];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((n >> 3) == half) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                XtFrag hi, lo;
                xt_split8_k16(make_float4(acc[2 * u][0], acc[2 * u][1], acc[2 * u][2], acc[2 * u][3]),
                              make_float4(acc[2 * u + 1][0], acc[2 * u + 1][1], acc[2 * u + 1][2], acc[2 * u + 1][3]), hi, lo);
                const int c = u * 8 + g * 2, r8 = n & 7;
                st[r8 * 64 + (c ^ (r8 << 1))] = hi.u;
                st[r8 * 64 + ((c + 1) ^ (r8 << 1))] = lo.u;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the wave's LDS writes have landed
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            const int row = q0 + 8 * half + r8;
            const uint4 v = st[r8 * 64 + (lane ^ (r8 << 1))];
            if (row < R) Qt[(long long)row * 512 + h * 64 + lane] = v;
        }
        __builtin_amdgcn_wave_barrier();                         // before the second half overwrites the stage
    }
}

// ------------------------------------------------------------------------------------------------
// context map: ctx[r][32 h + d] = sum_c Wv[32 h + d][c] z[r][h][c] + bv[32 h + d]   (bf16x3: z split hi / lo, Wv split hi / lo)
// Block = 16 queries, wave = head.  WB_hi / WB_lo [8 heads][8 k-steps][2 column tiles][64 lanes][8]:
//   lane (n, g), e: Wv[32 h + 16 nt + n][32 s + 8 g + e]                                          (ops.pack_xattn_maps)
// A CSR row without a key: ctx = NaN (empty_nan, like nn.MultiheadAttention) or 0 — the value bias must not reach such a query.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void xattn_ctxmap_kernel(const float* __restrict__ z, const uint4* __restrict__ WB_hi,
                                                           const uint4* __restrict__ WB_lo, const float* __restrict__ bv,
                                                           const int* __restrict__ row_ptr, float* __restrict__ ctx, int R, int empty_nan) {
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 16;
    const int row = min(q0 + n, R - 1);
    const float* zp = z + ((long long)row * HEADS + h) * C + 8 * g;
    const uint4* wh = WB_hi + (long long)h * 16 * 64 + lane;
    const uint4* wl = WB_lo + (long long)h * 16 * 64 + lane;
    f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float4 x0[8], x1[8];
#ifdef MV2D_CTXMAP_DIRECT
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        x0[s] = *reinterpret_cast<const float4*>(zp + 32 * s);
        x1[s] = *reinterpret_cast<const float4*>(zp + 32 * s + 4);
    }
#else
    {
        // a lane needs 32-byte pieces 128 bytes apart of its row's 1 KB: the rows are read whole (one load instruction = one row's 1 KB) and
        // redistributed through LDS, two halves of 8 rows (chunk index XOR-ed with the row)
        (void)zp;
        __shared__ float4 stage[8][8 * 64];
        float4* st = stage[h];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 v[8];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int rr = min(q0 + 8 * half + r8, R - 1);
                v[r8] = *reinterpret_cast<const float4*>(z + ((long long)rr * HEADS + h) * C + 4 * lane);
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) st[r8 * 64 + (lane ^ (r8 << 1))] = v[r8];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if ((n >> 3) == half) {
                const int r8 = n & 7;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int c = s * 8 + g * 2;
                    x0[s] = st[r8 * 64 + (c ^ (r8 << 1))];
                    x1[s] = st[r8 * 64 + ((c + 1) ^ (r8 << 1))];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#endif
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        XtFrag ah, al;
        xt_split8(x0[s], x1[s], ah, al);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            XtFrag bh, bl;
            bh.u = wh[(s * 2 + nt) * 64];
            bl.u = wl[(s * 2 + nt) * 64];
            acc[nt] = mfma_q16_16x16x32(al.v, bh.v, acc[nt], 0, 0, 0);
            acc[nt] = mfma_q16_16x16x32(ah.v, bl.v, acc[nt], 0, 0, 0);
            acc[nt] = mfma_q16_16x16x32(ah.v, bh.v, acc[nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = q0 + 4 * g + i;
        if (r < R) {
            const bool empty = row_ptr[r + 1] <= row_ptr[r];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = 32 * h + 16 * nt + n;
                float v = acc[nt][i] + bv[col];
                if (empty) v = empty_nan ? __uint_as_float(0x7fc00000u) : 0.f;
                ctx[(long long)r * C + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tile attention: see the file header.  One block (NW waves) per query; wave w takes the key tiles w, w + NW, ...
//   Qt  [R][8 heads][8 k-steps][4][hi | lo][8] key16 (xattn_qmap_kernel), Xk / Xv [S][256] key16, CSR row_ptr / col_idx, z [R][8][256] fp32
// LDS (one array): per wave an 8 KB key tile (16 rows x 32 chunks of 16 B, chunk c of row r at slot c ^ (r & 15): the
// fragment reads of 16 different rows hit 16 different bank slots) that later holds the wave's partial z, 512 B of P, and the
// softmax statistics of the merge.
// ------------------------------------------------------------------------------------------------
// (v_perm_b32: bytes 0-3 of the selector index {S1 = a: 0..3, S0 = b: 4..7}; the shift / mask formulation compiled to two VALU ops per pair)
__device__ __forceinline__ unsigned int xt_lo_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }     // (a.lo16, b.lo16)
__device__ __forceinline__ unsigned int xt_hi_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }     // (a.hi16, b.hi16)

// XLO (the engine's index-exact route): the key / value rows come as key16 hi + lo pairs (fp32-class key side):
// logits += Qt_hi . Xk_lo, z += P_hi . Xv_lo (the lo x lo terms, 2^-18 relative, are dropped).
// maximum over the 16 lanes of a DPP row (lane & 15 = the key of a tile): two quad permutes, then the half-row and the row mirror.  Four
// v_max with a DPP operand instead of four dependent ds_bpermute round trips per softmax row.
#define XT_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float xt_row16_max(float v) {
    v = fmaxf(v, XT_DPP(v, 0xB1));      // quad_perm [1,0,3,2]
    v = fmaxf(v, XT_DPP(v, 0x4E));      // quad_perm [2,3,0,1]
    v = fmaxf(v, XT_DPP(v, 0x141));     // row_half_mirror
    v = fmaxf(v, XT_DPP(v, 0x140));     // row_mirror
    return v;
}

typedef unsigned int xt_u32x4 __attribute__((ext_vector_type(4)));      // staging registers (arrays of HIP's uint4 STRUCT that live across a loop end up in scratch)
typedef unsigned int xt_u32x2 __attribute__((ext_vector_type(2)));
// XLO: 0 = key16 rows alone (default route), 1 = hi + key16 lo rows, 2 = hi + e4m3 lo rows (common.h "lo8": 256-byte rows, converted to key16 in registers
// on their way into the LDS tile / the v_perm transposes -- the MFMAs and everything behind them are those of XLO = 1)
template <int NW, bool DBG, int XLO>
__global__ __launch_bounds__(64 * NW, 2) void xattn_tile_kernel(const uint4* __restrict__ Qt, const unsigned short* __restrict__ Xk,
                                                             const unsigned short* __restrict__ Xv, const unsigned short* __restrict__ Xk_lo,
                                                             const unsigned short* __restrict__ Xv_lo, const int* __restrict__ row_ptr,
                                                             const int* __restrict__ col_idx, float* __restrict__ z,
                                                             float* __restrict__ dbg_logits, long long dbg_stride, int R, int empty_nan,
                                                             const int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NW * 8192 + NW * 512 + NW * 64 + (XLO ? NW * 8192 : 0)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    // XCD-aware block -> query map (block b runs on XCD b % 8): every XCD gets one contiguous range of queries, so neighbouring
    // queries (T path: overlapping key sets) share an L2.  Speed only; any map is correct.
    int r;
    {
        const int b = blockIdx.x, x = b & 7, qn = R >> 3, rem = R & 7;
        r = (x < rem ? x * (qn + 1) : rem * (qn + 1) + (x - rem) * qn) + (b >> 3);
        // optional query order (T path: the queries of a sample sorted by their smallest key, mv2d_xattn_query_order's perm): blocks that run
        // side by side on an XCD then read overlapping key sets and share its L2.  Speed only.
        if (order) r = order[r];
    }
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    float* zr = z + (long long)r * (HEADS * C);
    if (end <= beg) {
        const float v = empty_nan ? __uint_as_float(0x7fc00000u) : 0.f;
        for (int i = tid; i < HEADS * C; i += 64 * NW) zr[i] = v;
        return;
    }
    uint4* kt = reinterpret_cast<uint4*>(smem) + wave * 512;
    float* pl = reinterpret_cast<float*>(smem + NW * 8192) + wave * 128;
    float* sst = reinterpret_cast<float*>(smem + NW * 8192 + NW * 512);              // [NW][8] running max, [NW][8] sums
    uint4* kt2 = reinterpret_cast<uint4*>(smem + NW * 8192 + NW * 512 + NW * 64) + wave * 512;      // XLO: the lo parts of the key tile

    // this lane's rows of S / z: operand rows 4g + i; rows 0-7 carry the hi parts, 8-15 the lo parts of head (4 (g & 1) + i)
    float m_run[4], l_run[4];
    f32x4_t Z[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m_run[i] = -INFINITY; l_run[i] = 0.f; }
#pragma unroll
    for (int u = 0; u < 16; ++u) Z[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int ntile = (end - beg + 15) >> 4;
    XtFrag qa[8];
    if (wave < ntile) {
        const uint4* qp = Qt + (long long)r * 512 + (n & 7) * 64 + g * 2 + (n >> 3);
#pragma unroll
        for (int s = 0; s < 8; ++s) qa[s].u = qp[s * 8];
    }
    // ---- the pieces of a tile.  Gather: Xk rows whole (lanes 0-31 one row, 32-63 the next), Xv rows as 16-byte column chunks of keys 4g..4g+3
    // (byte offsets as 32-bit unsigned: scalar base + vector offset addressing instead of 64-bit address arithmetic per row;
    //  the row arrays must stay below 4 GB = 2^23 rows of 512 B, include/mv2d_hip.h).  `myidx`: lane (n, *) holds the index of key n of the tile.
    auto load_v = [&](const unsigned short* V_, int myidx, xt_u32x4 (&dst)[4][2]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int vidx = (unsigned int)__shfl(myidx, 4 * g + e, 64);
            const char* vp = reinterpret_cast<const char*>(V_) + ((vidx << 9) + 16u * (unsigned)n);
            dst[e][0] = *reinterpret_cast<const xt_u32x4*>(vp);
            dst[e][1] = *reinterpret_cast<const xt_u32x4*>(vp + 256);
        }
    };
    auto load_k = [&](const unsigned short* K_, int myidx, xt_u32x4 (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned int ridx = (unsigned int)__shfl(myidx, 2 * i + (lane >> 5), 64);
            dst[i] = *reinterpret_cast<const xt_u32x4*>(reinterpret_cast<const char*>(K_) + ((ridx << 9) + (unsigned)(lane & 31) * 16u));
        }
    };
    auto store_k = [&](uint4* tile, const xt_u32x4 (&src)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rowi = 2 * i + (lane >> 5);
            reinterpret_cast<xt_u32x4*>(tile)[rowi * 32 + ((lane & 31) ^ (rowi & 15))] = src[i];
        }
    };
    // e4m3 lo rows: the same lane -> (row, channels) assignment at half the bytes (8 per lane and row: a half wave reads one 256-byte row)
    auto load_v8 = [&](const unsigned short* V_, int myidx, xt_u32x2 (&dst)[4][2]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int vidx = (unsigned int)__shfl(myidx, 4 * g + e, 64);
            const char* vp = reinterpret_cast<const char*>(V_) + ((vidx << 8) + 8u * (unsigned)n);
            dst[e][0] = *reinterpret_cast<const xt_u32x2*>(vp);
            dst[e][1] = *reinterpret_cast<const xt_u32x2*>(vp + 128);
        }
    };
    auto load_k8 = [&](const unsigned short* K_, int myidx, xt_u32x2 (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned int ridx = (unsigned int)__shfl(myidx, 2 * i + (lane >> 5), 64);
            dst[i] = *reinterpret_cast<const xt_u32x2*>(reinterpret_cast<const char*>(K_) + ((ridx << 8) + (unsigned)(lane & 31) * 8u));
        }
    };
    auto store_k8 = [&](uint4* tile, const xt_u32x2 (&src)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rowi = 2 * i + (lane >> 5);
            tile[rowi * 32 + ((lane & 31) ^ (rowi & 15))] = lo8_chunk(make_uint2(src[i].x, src[i].y));
        }
    };
    // logits, online softmax and P . V of tile tt: the key tile is in LDS (kt, XLO: kt2), the value rows in registers
    auto compute = [&](int tt, const xt_u32x4 (&vreg)[4][2], const auto& vlo) {
        const int kbase = beg + 16 * tt;
        // ---- logits of the tile: D[row 4g+i][key n] = sum_c Qt[row][c] Xk[key][c]
        f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            XtFrag kb;
            kb.u = kt[n * 32 + ((4 * s + g) ^ n)];
            sacc = mfma_k16_16x16x32(qa[s].u, kb.u, sacc);
            if (XLO) {
                XtFrag kl, qh;
                kl.u = kt2[n * 32 + ((4 * s + g) ^ n)];
                qh.u = n < 8 ? qa[s].u : make_uint4(0u, 0u, 0u, 0u);              // hi rows only
                sacc = mfma_k16_16x16x32(qh.u, kl.u, sacc);
            }
        }
        const bool valid = kbase + n < end;
        float sv[4], p[4], alpha[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // hi rows + lo rows (head 4 (g & 1) + i) sit 32 lanes apart: one v_permlane32_swap instead of a trip through the LDS crossbar
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sacc[i]), __float_as_uint(sacc[i]), false, false);
            const float full = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            if (DBG && dbg_logits && g < 2 && valid) dbg_logits[(long long)(4 * g + i) * dbg_stride + kbase + n] = full;
            sv[i] = valid ? full * LOG2E : -INFINITY;                               // the softmax runs in base 2 (v_exp_f32)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float tm = sv[i];
            tm = xt_row16_max(tm);                                                   // over the 16 key lanes, DPP (no LDS round trips)
            const float m_new = fmaxf(m_run[i], tm);
            alpha[i] = __builtin_amdgcn_exp2f(m_run[i] - m_new);
            p[i] = __builtin_amdgcn_exp2f(sv[i] - m_new);
            l_run[i] = l_run[i] * alpha[i] + p[i];                                   // per-lane share of the row sum (reduced at the end)
            m_run[i] = m_new;
        }
        // ---- P as the A operand of the 16x16x16 MFMA: lane (row n, g): keys 4g..4g+3 of head n & 7, hi (n < 8) or lo part
        if (g < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pl[(4 * g + i) * 16 + n] = p[i];
        }
        __builtin_amdgcn_wave_barrier();
        uint2 pa, pah;
        {
            const float4 pv = *reinterpret_cast<const float4*>(pl + (n & 7) * 16 + 4 * g);
            unsigned int h0, h1, l0, l1;
            split_k16x2_bounded(pv.x, pv.y, h0, l0);                               // probabilities: inside the fp16 range, no clamp
            split_k16x2_bounded(pv.z, pv.w, h1, l1);
            pa = n < 8 ? make_uint2(h0, h1) : make_uint2(l0, l1);
            pah = n < 8 ? make_uint2(h0, h1) : make_uint2(0u, 0u);
        }
        // ---- z = alpha z + P . Xv_tile; column tile (H, w): output column n <-> channel 128 H + 8 n + w
        // The rescaling runs unconditionally.  In the first tile of a wave alpha = 2^-inf = 0 multiplies rows that are still 0; a guard `if (not the
        // first tile)` is wave-uniform but not provably so and compiled to 64 v_cndmask per tile (a quarter of the loop's vector instructions:
        // 65 -> 60 us per cfg2_s layer without it).  Skipping the 64 multiplications behind a ballot when no head's maximum moved is exact but
        // slower (61.3 -> 62.2 us: the branch costs more than the multiplications it saves).
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) Z[u][i] *= alpha[i];
#pragma unroll
        for (int H = 0; H < 2; ++H) {
            const unsigned int r0[4] = {vreg[0][H].x, vreg[0][H].y, vreg[0][H].z, vreg[0][H].w};
            const unsigned int r1[4] = {vreg[1][H].x, vreg[1][H].y, vreg[1][H].z, vreg[1][H].w};
            const unsigned int r2[4] = {vreg[2][H].x, vreg[2][H].y, vreg[2][H].z, vreg[2][H].w};
            const unsigned int r3[4] = {vreg[3][H].x, vreg[3][H].y, vreg[3][H].z, vreg[3][H].w};
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int d = w >> 1;
                const uint2 vb = (w & 1) ? make_uint2(xt_hi_pair(r0[d], r1[d]), xt_hi_pair(r2[d], r3[d]))
                                         : make_uint2(xt_lo_pair(r0[d], r1[d]), xt_lo_pair(r2[d], r3[d]));
                f32x4_t zc = Z[H * 8 + w];
                zc = mfma_k16_16x16x16(pa, vb, zc);
                if constexpr (XLO != 0) {
                    // channel pair d of key e: a dword of the key16 lo row, or two bytes of the e4m3 row converted here (one v_cvt per pair and key)
                    auto lo_pair_of = [&](int e) -> unsigned int {
                        if constexpr (XLO == 2) {
                            const unsigned int b = vlo[e][H][d >> 1];
                            return (d & 1) ? lo8_pair<1>(b) : lo8_pair<0>(b);
                        } else {
                            return vlo[e][H][d];
                        }
                    };
                    const unsigned int q0 = lo_pair_of(0), q1 = lo_pair_of(1), q2 = lo_pair_of(2), q3 = lo_pair_of(3);
                    const uint2 vl = (w & 1) ? make_uint2(xt_hi_pair(q0, q1), xt_hi_pair(q2, q3)) : make_uint2(xt_lo_pair(q0, q1), xt_lo_pair(q2, q3));
                    zc = mfma_k16_16x16x16(pah, vl, zc);
                }
                Z[H * 8 + w] = zc;
            }
        }
        __builtin_amdgcn_wave_barrier();                                             // before the next tile overwrites kt / pl
    };
    if constexpr (XLO == 2) {
        // index-exact route with e4m3 lo rows (round 6): the order of XLO = 1 below; the lo halves are 8-byte loads and 16 + 16 staging registers
        int idx_next = wave < ntile ? col_idx[min(beg + 16 * wave + n, end - 1)] : 0;
        for (int tt = wave; tt < ntile; tt += NW) {
            const int myidx = idx_next;
            if (tt + NW < ntile) idx_next = col_idx[min(beg + 16 * (tt + NW) + n, end - 1)];
            xt_u32x4 kreg[8], vreg[4][2];
            xt_u32x2 klo[8], vlo[4][2];
            load_k(Xk, myidx, kreg);
            load_k8(Xk_lo, myidx, klo);
            store_k(kt, kreg);
            store_k8(kt2, klo);
            load_v(Xv, myidx, vreg);
            load_v8(Xv_lo, myidx, vlo);
            __builtin_amdgcn_wave_barrier();
            compute(tt, vreg, vlo);
        }
    } else if constexpr (XLO == 1) {
        // index-exact route, TWO PHASES per tile (round 4): the hi and lo halves of the 16 key rows are requested together (16 loads in flight) and
        // go to LDS; only then the hi and lo value rows are requested -- into the registers the key rows just left -- and arrive while the
        // logits and the softmax run.  (Round 3 requested K hi, V hi up front and the lo halves behind the first LDS writes, in 32 more
        // registers: 314 us instead of 92 us per layer at cfg3_t for twice the bytes.)  The key indices of a tile are requested one tile AHEAD.
        int idx_next = wave < ntile ? col_idx[min(beg + 16 * wave + n, end - 1)] : 0;
        for (int tt = wave; tt < ntile; tt += NW) {
            const int myidx = idx_next;
            if (tt + NW < ntile) idx_next = col_idx[min(beg + 16 * (tt + NW) + n, end - 1)];
            xt_u32x4 kreg[8], klo[8], vreg[4][2], vlo[4][2];
            load_k(Xk, myidx, kreg);
            load_k(Xk_lo, myidx, klo);
            store_k(kt, kreg);
            store_k(kt2, klo);
            load_v(Xv, myidx, vreg);
            load_v(Xv_lo, myidx, vlo);
            __builtin_amdgcn_wave_barrier();
            compute(tt, vreg, vlo);
        }
    } else {
        // default route: the key indices of a tile are requested one tile AHEAD (round 4: a tile is two dependent round trips, index then rows;
        // the index trip of the next tile runs under the current tile's gather and arithmetic: cfg3_t 94.5 -> 87.3 us per layer).  Requesting
        // the ROWS of the next tile ahead as well (software pipelining: key rows through a second register set, or key + value rows with two
        // named value buffers, 230 / 256 registers) does not pay: cfg3_t 89.9 -> 87.9 / 91.3 us, cfg5_t 85.0 -> 81.8 / 84.6, cfg2_s 63.2 ->
        // 66.3 / 68.2 (same box, round 4) -- twice the bytes in flight per wave buy nothing, the kernel sits at what the memory system
        // delivers for 512-byte rows (4-5 TB/s from HBM, 10-12 TB/s where L2 serves the repeats), not at a per-wave latency chain.
        int idx_next = wave < ntile ? col_idx[min(beg + 16 * wave + n, end - 1)] : 0;
        for (int tt = wave; tt < ntile; tt += NW) {
            const int myidx = idx_next;
            if (tt + NW < ntile) idx_next = col_idx[min(beg + 16 * (tt + NW) + n, end - 1)];
            xt_u32x4 kreg[8], vreg[4][2];
            load_k(Xk, myidx, kreg);
            load_v(Xv, myidx, vreg);
            store_k(kt, kreg);
            __builtin_amdgcn_wave_barrier();
            compute(tt, vreg, vreg);
        }
    }
    // ---- row sums over the 16 key lanes; partial (m, l, z) of the wave -> LDS (z into the wave's own key-tile region)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float l = l_run[i];
        l += __shfl_xor(l, 1, 64);
        l += __shfl_xor(l, 2, 64);
        l += __shfl_xor(l, 4, 64);
        l += __shfl_xor(l, 8, 64);
        l_run[i] = l;
    }
    if (n == 0 && g < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sst[wave * 8 + 4 * g + i] = m_run[i];
            sst[NW * 8 + wave * 8 + 4 * g + i] = l_run[i];
        }
    }
    {
        // hi rows (lanes 0-31) + lo rows (lanes 32-63) of z with ONE half-wave exchange per register pair (v_permlane32_swap): afterwards
        // lanes g < 2 hold the sums of column tiles w = 0..3 and lanes g >= 2 those of w = 4..7 (for head 4 (g & 1) + i), so that every lane
        // stores one float4 per (half, row)
        float* szw = reinterpret_cast<float*>(smem) + wave * (HEADS * C);
#pragma unroll
        for (int H = 0; H < 2; ++H)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Z[H * 8 + w][i]), __float_as_uint(Z[H * 8 + w + 4][i]), false, false);
                    v[w] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                }
                float* dst = szw + (4 * (g & 1) + i) * C + 128 * H + 8 * n + 4 * (g >> 1);
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
    __syncthreads();
    // ---- merge the waves: thread -> (channel, half of the heads)
    {
        const float* sz = reinterpret_cast<const float*>(smem);
        for (int idx = tid; idx < HEADS * C; idx += 64 * NW) {
            const int h = idx >> 8, c = idx & (C - 1);
            float M = sst[h];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, sst[w * 8 + h]);
            float den = 0.f, num = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float e = __builtin_amdgcn_exp2f(sst[w * 8 + h] - M);        // waves without a tile: 2^(-inf) = 0
                den += sst[NW * 8 + w * 8 + h] * e;
                num += sz[(w * HEADS + h) * C + c] * e;
            }
            zr[h * C + c] = num * __builtin_amdgcn_rcpf(den);
        }
    }
}

}  // namespace

extern "C" int mv2d_xattn_qmap(const float* q, const void* WA_hi, const void* WA_lo, void* Qt, int R, void* stream) {
    MV2D_CHECK_ARG(q && WA_hi && WA_lo && Qt && R >= 0, "mv2d_xattn_qmap: bad args");
    MV2D_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)WA_hi & 15) == 0 && ((uintptr_t)WA_lo & 15) == 0 && ((uintptr_t)Qt & 15) == 0,
                   "mv2d_xattn_qmap: operands must be 16-byte aligned");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(xattn_qmap_kernel, dim3(cdiv(R, 16)), dim3(512), 0, (hipStream_t)stream, q, (const uint4*)WA_hi, (const uint4*)WA_lo,
                       (uint4*)Qt, R);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_xattn_ctxmap(const float* z, const void* WB_hi, const void* WB_lo, const float* bv, const int* row_ptr, float* ctx, int R,
                                 int empty_nan, void* stream) {
    MV2D_CHECK_ARG(z && WB_hi && WB_lo && bv && row_ptr && ctx && R >= 0, "mv2d_xattn_ctxmap: bad args");
    MV2D_CHECK_ARG(((uintptr_t)z & 15) == 0 && ((uintptr_t)WB_hi & 15) == 0 && ((uintptr_t)WB_lo & 15) == 0, "mv2d_xattn_ctxmap: operands must be 16-byte aligned");
    if (R == 0) return MV2D_OK;
    hipLaunchKernelGGL(xattn_ctxmap_kernel, dim3(cdiv(R, 16)), dim3(512), 0, (hipStream_t)stream, z, (const uint4*)WB_hi, (const uint4*)WB_lo, bv,
                       row_ptr, ctx, R, empty_nan);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_xattn_tile_fwd_ordered(const void* Qt, const void* Xk, const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr,
                                           const int* col_idx, float* z, float* dbg_logits, long long dbg_stride, int R, int empty_nan, int waves,
                                           const int* order, int lo_fmt, void* stream);

extern "C" int mv2d_xattn_tile_fwd(const void* Qt, const void* Xk, const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr,
                                   const int* col_idx, float* z, float* dbg_logits, long long dbg_stride, int R, int empty_nan, int waves, void* stream) {
    return mv2d_xattn_tile_fwd_ordered(Qt, Xk, Xv, Xk_lo, Xv_lo, row_ptr, col_idx, z, dbg_logits, dbg_stride, R, empty_nan, waves, nullptr, 0, stream);
}

extern "C" int mv2d_xattn_tile_fwd_ordered(const void* Qt, const void* Xk, const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr,
                                           const int* col_idx, float* z, float* dbg_logits, long long dbg_stride, int R, int empty_nan, int waves,
                                           const int* order, int lo_fmt, void* stream) {
    MV2D_CHECK_ARG(Qt && Xk && Xv && row_ptr && col_idx && z && R >= 0, "mv2d_xattn_tile_fwd: bad args");
    MV2D_CHECK_ARG(lo_fmt == 0 || (lo_fmt == 1 && Xk_lo), "mv2d_xattn_tile_fwd: lo_fmt is 0 (key16 lo rows) or 1 (e4m3 lo rows; needs the lo rows)");
    MV2D_CHECK_ARG((Xk_lo == nullptr) == (Xv_lo == nullptr), "mv2d_xattn_tile_fwd: Xk_lo and Xv_lo come together");
    MV2D_CHECK_ARG(((uintptr_t)Qt & 15) == 0 && ((uintptr_t)Xk & 15) == 0 && ((uintptr_t)Xv & 15) == 0 && ((uintptr_t)z & 15) == 0 &&
                       ((uintptr_t)Xk_lo & 15) == 0 && ((uintptr_t)Xv_lo & 15) == 0, "mv2d_xattn_tile_fwd: operands must be 16-byte aligned");
    MV2D_CHECK_ARG(waves == 0 || waves == 1 || waves == 2 || waves == 4 || waves == 8, "mv2d_xattn_tile_fwd: waves per query must be 1, 2, 4 or 8 (0: default)");
    if (R == 0) return MV2D_OK;
    const int nw = waves ? waves : 2;      // the engine passes its own choice (2: see engine.py)
#define MV2D_XT(NW, DBG, XLO) hipLaunchKernelGGL((xattn_tile_kernel<NW, DBG, XLO>), dim3(R), dim3(64 * NW), 0, (hipStream_t)stream, (const uint4*)Qt, \
                                                 (const unsigned short*)Xk, (const unsigned short*)Xv, (const unsigned short*)Xk_lo, (const unsigned short*)Xv_lo, \
                                                 row_ptr, col_idx, z, dbg_logits, dbg_stride, R, empty_nan, order)
    // index-exact route (hi + lo rows): two waves per SIMD like the default (round 3: the 312-register build ran one wave per SIMD, 133 us per layer)
    if (Xk_lo && lo_fmt == 1) { if (dbg_logits) MV2D_XT(4, true, 2); else if (nw == 4) MV2D_XT(4, false, 2); else if (nw == 1) MV2D_XT(1, false, 2); else MV2D_XT(2, false, 2); }
    else if (Xk_lo) { if (dbg_logits) MV2D_XT(4, true, 1); else if (nw == 4) MV2D_XT(4, false, 1); else if (nw == 1) MV2D_XT(1, false, 1); else MV2D_XT(2, false, 1); }
    else if (dbg_logits) { if (nw == 8) MV2D_XT(8, true, 0); else if (nw == 2) MV2D_XT(2, true, 0); else if (nw == 1) MV2D_XT(1, true, 0); else MV2D_XT(4, true, 0); }
    else { if (nw == 8) MV2D_XT(8, false, 0); else if (nw == 2) MV2D_XT(2, false, 0); else if (nw == 1) MV2D_XT(1, false, 0); else MV2D_XT(4, false, 0); }
#undef MV2D_XT
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
