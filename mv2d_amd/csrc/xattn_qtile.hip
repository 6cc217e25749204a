// T-path cross attention with SHARED KEY TILES (gfx950 / CDNA4, wave64): one workgroup per tile of 16 queries, the union of the
// tile's key lists streamed ONCE through LDS, per-pair masks as 16-bit words (round 3; SURVEY section 7 step 3 "one workgroup per
// query tile").
//
// Why: on the masked-map path (MV2DTHead, RH/mv2d_t_head.py:79-109; PETRMultiheadAttention with a boolean attn_mask,
// MU/petr_transformer.py:501-508) every key row is read by 2.9 (cfg3_t) to 6.2 (cfg5_t) queries.  xattn_tile_kernel (one block per
// query) re-gathers those rows once per query through L2 / Infinity Cache: 424 MB per launch against 169 MB of distinct rows.  Queries
// that share keys are neighbours once the queries of a sample are ORDERED BY THEIR SMALLEST KEY INDEX (the key list is in (view, y, x)
// order: a query and the queries of the RoIs it is epipolar-matched with start at the same cells); tiles of 16 consecutive queries in
// that order read 1.37 x (cfg3_t) the distinct rows.
//
// Three kernels:
//   qt_order_kernel   per sample: rank of every query by (smallest key, query index) -> perm; tile table (first slot / queries per tile)
//   qt_build_kernel   per query tile: OR of the queries' cell bitmasks (mv2d_mask_compact's `bits`) -> ascending union key list (padded
//                     to a multiple of 16) + for every (16-key union tile, query) a 16-bit mask of the allowed pairs
//   xattn_qtile_kernel<NW>   block = 2 NW queries (NW waves, TWO queries per wave), the union's K / V tiles (16 rows x 512 B each) through a
//                     double-buffered LDS ring filled two tiles ahead.  Same arithmetic as xattn_tile_kernel (raw key space: Qt = per-head
//                     query maps as bf16 hi + lo, base-2 online softmax, P as bf16 hi + lo), with the hi / lo parts folded into the K
//                     dimension of the MFMAs instead of their M rows, so that the 16 MFMA rows carry 2 queries x 8 heads:
//                       logits  S[(q, h)][key]  = sum over 512 k = [Qt_hi | Qt_lo][(q, h)][c] . [Xk | Xk][key][c]       16 x v_mfma_f32_16x16x32_bf16
//                       z[(q, h)][ch]          += sum over 32 k = [P_hi | P_lo][(q, h)][key] . [Xv ; Xv][key][ch]         16 x v_mfma_f32_16x16x32_bf16
//                     A wave whose two queries have no allowed key in a union tile skips it (its mask word is zero).
// The result differs from xattn_tile_kernel's only in the order of the fp32 sums (tests compare both with fp64).
#include <cstdlib>
#include "common.h"

namespace {

constexpr int C = 256, HEADS = 8;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int QT_MAX = 16;                   // queries per tile: 16 (8 waves x 2) or 8 (4 waves x 2)
constexpr int UT_MAX = 512;                  // union tiles (of 16 keys) a query tile may have: 8192 keys
constexpr int GRP_MAX = 4096;                // queries per sample the ordering kernel ranks in LDS

typedef __attribute__((ext_vector_type(8))) __bf16 qt_bf16x8;
typedef unsigned int qt_u32x4 __attribute__((ext_vector_type(4)));      // native vector: arrays of HIP's uint4 STRUCT end up in scratch
union QFrag { uint4 u; qt_bf16x8 v; qt_u32x4 n; };

#define QT_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float qt_row16_max(float v) {
    v = fmaxf(v, QT_DPP(v, 0xB1));
    v = fmaxf(v, QT_DPP(v, 0x4E));
    v = fmaxf(v, QT_DPP(v, 0x141));
    v = fmaxf(v, QT_DPP(v, 0x140));
    return v;
}
__device__ __forceinline__ unsigned int qt_lo_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ unsigned int qt_hi_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ------------------------------------------------------------------------------------------------
// groups: sample b = rows [grp_start[b], grp_start[b+1]), b < n_grp; the bucket-padding rows [grp_start[n_grp], R) form one more group.
// perm[slot] = query of sorted slot `slot` (slots of a group = its row range); tiles of QT consecutive slots inside a group.
// ------------------------------------------------------------------------------------------------
constexpr int ORD_CHUNK = 128;               // queries ranked per block of qt_order_kernel (8 threads per query)
__global__ __launch_bounds__(1024) void qt_order_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const int* __restrict__ grp_start,
                                                        int n_grp, int R, int* __restrict__ perm, int* __restrict__ tile_q0, int* __restrict__ tile_qn,
                                                        int* __restrict__ n_tiles, int* __restrict__ flags, int QT) {
    // grid (groups, chunks of ORD_CHUNK queries): every block holds the group's keys in LDS and ranks its chunk, 8 threads per query
    // (round 3: one block per group took 39 us for the ~950-query samples of cfg5_t)
    __shared__ int key[GRP_MAX];
    const int g = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const int lo = g < n_grp ? grp_start[g] : grp_start[n_grp];
    const int hi = g < n_grp ? grp_start[g + 1] : R;
    const int n = hi - lo;
    if (chunk == 0 && tile_q0) {                               // (null: only the order is wanted, mv2d_xattn_query_order)
        int tb = 0;                                            // tiles of the groups before this one
        for (int k = 0; k < g; ++k) {
            const int nk = (k < n_grp ? grp_start[k + 1] : R) - (k < n_grp ? grp_start[k] : grp_start[n_grp]);
            tb += (nk + QT - 1) / QT;
        }
        const int nt = (n + QT - 1) / QT;
        for (int i = tid; i < nt; i += 1024) { tile_q0[tb + i] = lo + i * QT; tile_qn[tb + i] = min(QT, n - i * QT); }
        if (g == n_grp && tid == 0) *n_tiles = tb + nt;
    }
    if (n > GRP_MAX) {                                         // too many queries in one sample for the LDS ranking: keep the natural order
        if (chunk == 0) {
            if (tid == 0) flags[0] = 1;
            for (int i = tid; i < n; i += 1024) perm[lo + i] = lo + i;
        }
        return;
    }
    if (chunk * ORD_CHUNK >= n) return;
    for (int i = tid; i < n; i += 1024) {
        const int b = row_ptr[lo + i], e = row_ptr[lo + i + 1];
        key[i] = e > b ? col_idx[b] : 0x7fffffff;              // the CSR rows are ascending: the first entry is the smallest key
    }
    __syncthreads();
    const int i = chunk * ORD_CHUNK + (tid >> 3), sub = tid & 7;
    int rank = 0;
    if (i < n) {
        const int ki = key[i];
        for (int j = sub; j < n; j += 8) { const int kj = key[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    rank += __shfl_xor(rank, 4);
    if (i < n && sub == 0) perm[lo + rank] = lo + i;
}

// ------------------------------------------------------------------------------------------------
// bits [R][nwords]: per-query bitmask over the cells of the query's sample (word (pos >> 5) - w0(sample), bit pos & 31), as written by
// csr_count_kernel; pos2s: cell -> key index.  Per tile: ukeys[uptr .. uptr + ceil16(ucnt)) ascending (the padding repeats the last key),
// qmask[(uptr >> 4) + t][w] = (mask of query slot 2w) | (mask of slot 2w + 1) << 16: the allowed keys of the slots among the 16 keys of union tile t.  Storage is handed out by an atomic bump
// counter (alloc[0], zeroed per frame): the placement of a tile is arbitrary, its contents are not.
// ------------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(256) void qt_build_kernel(const unsigned int* __restrict__ bits, int nwords, const int* __restrict__ rect, int V,
                                                       int cells_per_sample, const int* __restrict__ pos2s, const int* __restrict__ perm,
                                                       const int* __restrict__ tile_q0, const int* __restrict__ tile_qn, const int* __restrict__ n_tiles,
                                                       int* __restrict__ uptr, int* __restrict__ ucnt, int* __restrict__ ukeys, int ucap,
                                                       unsigned int* __restrict__ qmask, int* __restrict__ alloc, int* __restrict__ flags) {
    extern __shared__ unsigned int dyn[];                       // uw[nwords] | wbase[nwords]
    __shared__ unsigned int qm[UT_MAX * QT];
    __shared__ int wsum[4], carry, sbase;
    __shared__ int qs[QT];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (t >= *n_tiles) return;
    unsigned int* uw = dyn;
    int* wbase = reinterpret_cast<int*>(dyn + nwords);
    const int q0 = tile_q0[t], qn = tile_qn[t];
    if (tid < QT) qs[tid] = tid < qn ? perm[q0 + tid] : -1;
    if (tid == 0) carry = 0;
    for (int i = tid; i < UT_MAX * QT; i += 256) qm[i] = 0u;
    __syncthreads();
    const int w0 = (int)(((long long)(rect[qs[0] * 5] / V) * cells_per_sample) >> 5);
    // union words + exclusive scan of their popcounts (256 words per pass)
    for (int wb = 0; wb < nwords; wb += 256) {
        const int wi = wb + tid;
        unsigned int u = 0u;
        if (wi < nwords)
            for (int j = 0; j < qn; ++j) u |= bits[(long long)qs[j] * nwords + wi];
        const int c = __popc(u);
        int s = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(s, o, 64); if (lane >= o) s += x; }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        int off = carry + s - c;
        for (int k = 0; k < wv; ++k) off += wsum[k];
        if (wi < nwords) { uw[wi] = u; wbase[wi] = off; }
        __syncthreads();
        if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    int U = carry;
    if (U > UT_MAX * 16) { if (tid == 0) flags[0] = 1; U = UT_MAX * 16; }
    const int Up = (U + 15) & ~15;
    if (tid == 0) {
        const int b = atomicAdd(alloc, Up);
        if (b + Up > ucap) { flags[0] = 1; sbase = -1; } else sbase = b;
        uptr[t] = b + Up > ucap ? 0 : b;
        ucnt[t] = b + Up > ucap ? 0 : U;
    }
    __syncthreads();
    const int base = sbase;
    if (base < 0) return;
    // expand: key of every set bit + the per-query masks
    for (int wi = tid; wi < nwords; wi += 256) {
        unsigned int u = uw[wi];
        if (!u) continue;
        unsigned int bq[QT];
#pragma unroll
        for (int j = 0; j < QT; ++j) bq[j] = j < qn ? bits[(long long)qs[j] * nwords + wi] : 0u;
        int k = wbase[wi];
        while (u) {
            const int b = __ffs(u) - 1;
            u &= u - 1;
            if (k < U) {
                ukeys[base + k] = pos2s[(w0 + wi) * 32 + b];
#pragma unroll
                for (int j = 0; j < QT; ++j)
                    if ((bq[j] >> b) & 1u) atomicOr(&qm[(k >> 4) * QT + j], 1u << (k & 15));
            }
            ++k;
        }
    }
    __syncthreads();
    // padding keys: repeat the last key (their mask bits stay zero)
    if (tid < Up - U) {
        // (the last key was written by some thread of this block: read it back after the barrier through the same pointer)
        ukeys[base + U + tid] = U > 0 ? ukeys[base + U - 1] : 0;
    }
    // one dword per (union tile, wave of the attention kernel): the masks of the wave's two query slots
    const int nut = Up >> 4;
    unsigned int* out = qmask + (long long)(base >> 4) * (QT / 2);
    for (int i = tid; i < nut * (QT / 2); i += 256) {
        const int ut = i / (QT / 2), w = i - ut * (QT / 2);
        out[i] = (qm[ut * QT + 2 * w] & 0xffffu) | (qm[ut * QT + 2 * w + 1] << 16);
    }
}

// ------------------------------------------------------------------------------------------------
// attention over a query tile: see the file header.  NW waves, wave w = query slots 2w, 2w + 1 of the tile.
//   Qt [R][8 heads][8 k-steps][4][hi | lo][8] bf16 (xattn_qmap_kernel), Xk / Xv [S][256] bf16, z [R][8][256] fp32
// MFMA rows m = 8 * (query of the wave) + head.
// ------------------------------------------------------------------------------------------------
// LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base, lds_base + 1 KB), lane linear (the swizzle of the tile
// image is applied to the SOURCE chunk).  Issued from inline asm and ordered by hand (counted vmcnt + barrier), like csrc/kvproj.hip: a
// compiler-visible DMA is drained before every following ds_read.
__device__ __forceinline__ void qt_dma16(const void* gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int CNT>
__device__ __forceinline__ void qt_wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (CNT & 15) | ((CNT >> 4) << 14)); }

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void xattn_qtile_kernel(const uint4* __restrict__ Qt, const unsigned short* __restrict__ Xk,
                                                                 const unsigned short* __restrict__ Xv, const int* __restrict__ perm,
                                                                 const int* __restrict__ tile_q0, const int* __restrict__ tile_qn,
                                                                 const int* __restrict__ n_tiles, const int* __restrict__ uptr, const int* __restrict__ ucnt,
                                                                 const int* __restrict__ ukeys, const unsigned int* __restrict__ qmask,
                                                                 float* __restrict__ z, int empty_nan, int dbg_mode) {
    constexpr int QT = 2 * NW;
    constexpr int NT = 64 * NW;
    constexpr int NB = 4;                                       // ring of K / V tile buffers: NB - 1 tiles in flight behind the one being computed
    constexpr int PW = 16 / NW;                                 // 1 KB DMA pieces per wave and tile (8 of the K tile + 8 of the V tile per block)
    constexpr int SEG = 128;                                    // union tiles whose key indices / masks are resident in LDS at a time
    constexpr int TILE_B = 16384, RING_B = NB * TILE_B, PL_B = NW * 1024, UK_B = SEG * 16 * 4, MK_B = SEG * NW * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING_B + PL_B + UK_B + MK_B];
    float* pls = reinterpret_cast<float*>(smem + RING_B);
    int* uk = reinterpret_cast<int*>(smem + RING_B + PL_B);
    unsigned int* mk = reinterpret_cast<unsigned int*>(smem + RING_B + PL_B + UK_B);
    // XCD-aware block -> tile map (block b runs on XCD b % 8): every XCD takes one contiguous range of the (ordered) tiles, neighbours share its L2
    int tb;
    {
        const int nt_all = *n_tiles, b = blockIdx.x, x = b & 7, qn_ = nt_all >> 3, rem = nt_all & 7, k = b >> 3;
        if (k >= qn_ + (x < rem ? 1 : 0)) return;
        tb = (x < rem ? x * (qn_ + 1) : rem * (qn_ + 1) + (x - rem) * qn_) + k;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, g = lane >> 4;
    const int q0 = tile_q0[tb], qn = tile_qn[tb], u0 = uptr[tb], U = ucnt[tb];
    const int ntile = (U + 15) >> 4;
    const int j0 = 2 * wave, j1 = 2 * wave + 1;          // (mask dword of a union tile: slot j0 in the low half, j1 in the high half)
    const int r0 = j0 < qn ? perm[q0 + j0] : -1, r1 = j1 < qn ? perm[q0 + j1] : -1;
    float* pl = pls + wave * 256;

    // this wave's A operand of the logits: rows m = 8 qsel + h, 16 k-steps (8 of the hi parts, 8 of the lo parts)
    QFrag qa[16];
    {
        const int rq = max((n >> 3) ? r1 : r0, 0);
        const uint4* qp = Qt + (long long)rq * 512 + (n & 7) * 64 + g * 2;
#pragma unroll
        for (int s = 0; s < 8; ++s) { qa[s].u = qp[s * 8]; qa[8 + s].u = qp[s * 8 + 1]; }
    }
    float m_run[4], l_run[4];
    f32x4_t Z[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m_run[i] = -INFINITY; l_run[i] = 0.f; }
#pragma unroll
    for (int u = 0; u < 16; ++u) Z[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- tile ring (LDS-DMA): tile image = [16 rows][32 slots of 16 B], chunk c of row r in slot c ^ r; K image then V image.
    // Piece p (0..15) of a tile = rows 2 (p & 7), 2 (p & 7) + 1 of the K (p < 8) or V (p >= 8) image; wave w issues pieces w PW .. w PW + PW - 1.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int dma_half = lane >> 5, dma_slot = lane & 31;
    auto issue = [&](int t) {
        const unsigned bufb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(t % NB) * TILE_B);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int p = wave * PW + j, row = 2 * (p & 7) + dma_half;
            const unsigned int ridx = dbg_mode == 2 ? (unsigned int)(row + 16 * (blockIdx.x & 255)) : (unsigned int)uk[16 * t + row];
            const unsigned int off = ridx * (unsigned)(C * 2) + (unsigned)((dma_slot ^ row) << 4);
            const char* src = reinterpret_cast<const char*>(p < 8 ? Xk : Xv) + off;
            qt_dma16(src, bufb + (unsigned)p * 1024u);
        }
    };
    const unsigned int* qm = qmask + (long long)(u0 >> 4) * NW;

    auto compute = [&](int t) {
        // the two queries' masks for this union tile (wave-uniform)
        const unsigned int mm = __builtin_amdgcn_readfirstlane(mk[t * NW + wave]);
        if (mm == 0u || dbg_mode == 1) return;
        const qt_u32x4* ktb = reinterpret_cast<const qt_u32x4*>(smem + (t % NB) * TILE_B);
        const qt_u32x4* vtb = ktb + 512;
        f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            QFrag kb;
            kb.n = ktb[n * 32 + ((4 * s + g) ^ n)];
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[8 + s].v, kb.v, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[s].v, kb.v, sacc, 0, 0, 0);
        }
        const bool allowed = (mm >> (16 * (g >> 1) + n)) & 1u;          // rows 4g..4g+3 belong to query g >> 1; this lane's key is n
        float p[4], alpha[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sv = allowed ? sacc[i] * LOG2E : -INFINITY;
            const float tm = qt_row16_max(sv);
            const float m_new = fmaxf(m_run[i], tm);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;        // a query without a key so far: everything stays 0
            alpha[i] = __builtin_amdgcn_exp2f(m_run[i] - m_use);
            p[i] = __builtin_amdgcn_exp2f(sv - m_use);
            l_run[i] = l_run[i] * alpha[i] + p[i];
            m_run[i] = m_new;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) pl[(4 * g + i) * 16 + n] = p[i];
        __builtin_amdgcn_wave_barrier();
        // A operand of P . V: lane (row n, g): k = 8g..8g+7 = keys 8 (g & 1) + e, hi parts (g < 2) or lo parts (g >= 2)
        QFrag pa;
        {
            const float4 pv0 = *reinterpret_cast<const float4*>(pl + n * 16 + 8 * (g & 1));
            const float4 pv1 = *reinterpret_cast<const float4*>(pl + n * 16 + 8 * (g & 1) + 4);
            const float f[8] = {pv0.x, pv0.y, pv0.z, pv0.w, pv1.x, pv1.y, pv1.z, pv1.w};
            unsigned int h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
                l[e] = pack_bf16x2(f[2 * e] - __uint_as_float(h[e] << 16), f[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u));
            }
            pa.u = g < 2 ? make_uint4(h[0], h[1], h[2], h[3]) : make_uint4(l[0], l[1], l[2], l[3]);
        }
        // (once a row's running maximum has settled alpha is exactly 1: skip the 64 rescaling multiplies when that holds for the whole wave)
        const bool rescale = __builtin_amdgcn_ballot_w64((alpha[0] != 1.f) | (alpha[1] != 1.f) | (alpha[2] != 1.f) | (alpha[3] != 1.f)) != 0ull;
        // z = alpha z + P . Xv_tile; column tile (H, w): output column n <-> channel 128 H + 8 n + w; B[k = 8g + e][n] = Xv[key 8 (g & 1) + e][channel]
#pragma unroll
        for (int H = 0; H < 2; ++H) {
            qt_u32x4 vr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = 8 * (g & 1) + e;
                vr[e] = vtb[row * 32 + ((16 * H + n) ^ row)];
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int d = w >> 1;
                QFrag vb;
                vb.u = (w & 1) ? make_uint4(qt_hi_pair(vr[0][d], vr[1][d]), qt_hi_pair(vr[2][d], vr[3][d]), qt_hi_pair(vr[4][d], vr[5][d]), qt_hi_pair(vr[6][d], vr[7][d]))
                               : make_uint4(qt_lo_pair(vr[0][d], vr[1][d]), qt_lo_pair(vr[2][d], vr[3][d]), qt_lo_pair(vr[4][d], vr[5][d]), qt_lo_pair(vr[6][d], vr[7][d]));
                f32x4_t zc = Z[H * 8 + w];
                if (rescale) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) zc[i] *= alpha[i];
                }
                Z[H * 8 + w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa.v, vb.v, zc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();                                   // pl is rewritten by the next tile
    };

    for (int seg0 = 0; seg0 < ntile; seg0 += SEG) {
        const int nseg = min(SEG, ntile - seg0);
        __syncthreads();                                                    // (the previous segment's readers of uk / mk / the ring are done)
        for (int i = tid; i < nseg * 16; i += NT) uk[i] = ukeys[u0 + 16 * seg0 + i];
        for (int i = tid; i < nseg * NW; i += NT) mk[i] = qm[seg0 * NW + i];
        __syncthreads();
        qt_wait_vm<0>();                                                    // no ordinary load may be in flight beside the counted DMA pieces
        for (int t = 0; t < NB - 1 && t < nseg; ++t) issue(t);
        for (int t = 0; t < nseg; ++t) {
            // this wave's pieces of tile t have landed (younger pieces, of the tiles issued after it, may stay in flight); then everybody's
            const int ahead = min(nseg, t + NB - 1) - t - 1;                // tiles issued behind tile t
            if (ahead >= 2) qt_wait_vm<2 * PW>(); else if (ahead == 1) qt_wait_vm<PW>(); else qt_wait_vm<0>();
            __builtin_amdgcn_s_barrier();                                   // ... and every wave is done with tile t - 1: its buffer is free
            asm volatile("" ::: "memory");
            if (t + NB - 1 < nseg) issue(t + NB - 1);
            compute(t);
        }
    }
    // ---- row sums over the 16 key lanes, normalise, store: row 4g + i = (query g >> 1, head 4 (g & 1) + i); lane n holds channels
    // 128 H + 8 n + w (w = 0..7) of that row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float l = l_run[i];
        l += __shfl_xor(l, 1, 64);
        l += __shfl_xor(l, 2, 64);
        l += __shfl_xor(l, 4, 64);
        l += __shfl_xor(l, 8, 64);
        l_run[i] = l;
    }
    const int rq = (g >> 1) ? r1 : r0;
    if (rq < 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool empty = !(l_run[i] > 0.f);
        const float inv = empty ? 0.f : __builtin_amdgcn_rcpf(l_run[i]);
        const float ev = empty_nan ? __uint_as_float(0x7fc00000u) : 0.f;
        float* dst = z + ((long long)rq * HEADS + 4 * (g & 1) + i) * C + 8 * n;
#pragma unroll
        for (int H = 0; H < 2; ++H) {
            float v[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) v[w] = empty ? ev : Z[H * 8 + w][i] * inv;
            *reinterpret_cast<float4*>(dst + 128 * H) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 128 * H + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

}  // namespace

static inline int qt_order_chunks(int R) { const int n = R < GRP_MAX ? R : GRP_MAX; return (n + ORD_CHUNK - 1) / ORD_CHUNK; }
static inline int qt_size(int queries_per_tile) { return queries_per_tile == 16 ? 16 : queries_per_tile == 2 ? 2 : queries_per_tile == 4 ? 4 : 8; }
static inline bool qt_ok(int q) { return q == 2 || q == 4 || q == 8 || q == 16; }
extern "C" long long mv2d_xattn_qtile_max_tiles(int R, int n_samples, int queries_per_tile) {
    const int QT = qt_size(queries_per_tile);
    return (long long)(R + QT - 1) / QT + n_samples + 1;
}

// Query order alone (T path): perm [R] = the queries of every sample sorted by their smallest key (bucket-padding rows behind the last sample keep
// their places as a group of their own).  xattn_tile_kernel launched in this order reads overlapping key sets from neighbouring blocks.
extern "C" int mv2d_xattn_query_order(const int* row_ptr, const int* col_idx, const int* grp_start, int n_samples, int R, int* perm, int* flags, void* stream) {
    MV2D_CHECK_ARG(row_ptr && col_idx && grp_start && perm && flags && R > 0 && n_samples >= 1, "mv2d_xattn_query_order: bad args");
    hipLaunchKernelGGL(qt_order_kernel, dim3(n_samples + 1, qt_order_chunks(R)), dim3(1024), 0, (hipStream_t)stream, row_ptr, col_idx, grp_start, n_samples, R, perm, (int*)nullptr,
                       (int*)nullptr, (int*)nullptr, flags, 8);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

// Per frame, after mv2d_mask_compact (same stream): query order + tile table + union key lists + pair masks.  alloc / flags: int32
// device words zeroed by the caller before the call (flags[0] != 0 afterwards: a capacity was exceeded and the tables are unusable).
extern "C" int mv2d_xattn_qtile_build(const int* row_ptr, const int* col_idx, const int* grp_start, int n_samples, int R, const void* bits, int nwords,
                                      const int* rect, int V, int cells_per_sample, const int* pos2s, int* perm, int* tile_q0, int* tile_qn,
                                      int* n_tiles, int* uptr, int* ucnt, int* ukeys, int ucap, void* qmask, int* alloc, int* flags, int queries_per_tile,
                                      void* stream) {
    MV2D_CHECK_ARG(row_ptr && col_idx && grp_start && bits && rect && pos2s && perm && tile_q0 && tile_qn && n_tiles && uptr && ucnt && ukeys && qmask &&
                       alloc && flags, "mv2d_xattn_qtile_build: null pointer");
    MV2D_CHECK_ARG(R > 0 && n_samples >= 1 && nwords > 0 && nwords <= 8192 && ucap > 0 && (ucap % 16) == 0, "mv2d_xattn_qtile_build: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    MV2D_CHECK_ARG(qt_ok(queries_per_tile), "mv2d_xattn_qtile_build: 2, 4, 8 or 16 queries per tile");
    const int QT = qt_size(queries_per_tile);
    hipLaunchKernelGGL(qt_order_kernel, dim3(n_samples + 1, qt_order_chunks(R)), dim3(1024), 0, st, row_ptr, col_idx, grp_start, n_samples, R, perm, tile_q0, tile_qn, n_tiles, flags, QT);
    const int ntmax = (int)mv2d_xattn_qtile_max_tiles(R, n_samples, QT);
#define MV2D_QB(Q) hipLaunchKernelGGL(qt_build_kernel<Q>, dim3(ntmax), dim3(256), nwords * 8, st, (const unsigned int*)bits, nwords, rect, V, cells_per_sample, pos2s, \
                                      perm, tile_q0, tile_qn, n_tiles, uptr, ucnt, ukeys, ucap, (unsigned int*)qmask, alloc, flags)
    if (QT == 16) MV2D_QB(16); else if (QT == 8) MV2D_QB(8); else if (QT == 4) MV2D_QB(4); else MV2D_QB(2);
#undef MV2D_QB
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_xattn_qtile_fwd(const void* Qt, const void* Xk, const void* Xv, const int* perm, const int* tile_q0, const int* tile_qn,
                                    const int* n_tiles, const int* uptr, const int* ucnt, const int* ukeys, const void* qmask, float* z, int R,
                                    int n_samples, int empty_nan, int queries_per_tile, void* stream) {
    MV2D_CHECK_ARG(Qt && Xk && Xv && perm && tile_q0 && tile_qn && n_tiles && uptr && ucnt && ukeys && qmask && z && R >= 0, "mv2d_xattn_qtile_fwd: bad args");
    MV2D_CHECK_ARG(((uintptr_t)Qt & 15) == 0 && ((uintptr_t)Xk & 15) == 0 && ((uintptr_t)Xv & 15) == 0 && ((uintptr_t)z & 15) == 0,
                   "mv2d_xattn_qtile_fwd: operands must be 16-byte aligned");
    if (R == 0) return MV2D_OK;
    MV2D_CHECK_ARG(qt_ok(queries_per_tile), "mv2d_xattn_qtile_fwd: 2, 4, 8 or 16 queries per tile (as built)");
    static const int dbg_mode = getenv("MV2D_QTILE_DBG") ? atoi(getenv("MV2D_QTILE_DBG")) : 0;      // timing experiments: 1 = no arithmetic, 2 = the same 16 rows every tile
    const int ntmax = (int)mv2d_xattn_qtile_max_tiles(R, n_samples, queries_per_tile);
#define MV2D_QA(W) hipLaunchKernelGGL((xattn_qtile_kernel<W>), dim3(ntmax), dim3(64 * W), 0, (hipStream_t)stream, (const uint4*)Qt, (const unsigned short*)Xk, \
                                      (const unsigned short*)Xv, perm, tile_q0, tile_qn, n_tiles, uptr, ucnt, ukeys, (const unsigned int*)qmask, z, empty_nan, dbg_mode)
    if (queries_per_tile == 16) MV2D_QA(8); else if (queries_per_tile == 8) MV2D_QA(4); else if (queries_per_tile == 4) MV2D_QA(2); else MV2D_QA(1);
#undef MV2D_QA
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
