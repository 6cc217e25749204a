// Cross attention of a decoder layer with KEY TILES SHARED BETWEEN QUERIES (round 6; gfx950 / CDNA4, wave64).
//
// PETRMultiheadAttention's core (MU/petr_transformer.py:426-513) with the boolean masks of the two heads (RH/mv2d_t_head.py:79-109: the
// union of the correlated RoIs' rectangles; RH/mv2d_s_head.py:184-192: the cells of the correlated RoIs): a key row is allowed for 2.9 (cfg3_t)
// to 6.2 (cfg5_t / the overlapping S rig) queries.  The per-query kernels (xattn_tile.hip, xattn_fused.hip) request it once per query and leave
// the dedup to the L2s: 1.75-2.4 x the distinct rows cross the fabric.  Here a block owns a GROUP of up to 8 queries and walks the UNION of their
// key lists ONCE: a 16-key tile (hi + lo key rows, hi + lo value rows = 32 KB) is brought into LDS by the LDS-DMA (global_load_lds_dwordx4, a
// 4-deep ring, counted vmcnt + one barrier per tile) and all eight queries run their logits / online softmax / P.V against it.
//
// Round 3 built the idea with wave = query on position-ordered unions and lost 3 x: a wave idled on tiles none of its query's keys were in and
// did masked work on the others (LOG.md).  Here WAVE = HEAD: the 16 rows of a wave's MFMAs are (query, hi | lo part) of ONE head, the raw-key-space
// logits of that head for all 8 queries are one MFMA chain, so the eight waves do the same work on every tile -- a (query, key) slot the query
// does not list is masked at no cost in time -- and the per-head query map (phase A) and context map (phase C) of xattn_fused.hip stay inside
// the wave: one launch per layer, Qt and z never leave the wave's LDS scratch.  The union of a group is sorted by membership signature
// (the byte of member bits) inside a window of positions; the order of a softmax row's keys is free at fp32 rounding level.
//
// mv2d_xattn_group_tables builds the per-group tables once per frame from the CSR and a query order (groups = runs of 8 consecutive slots of
// a sample's order: queries that share keys should be neighbours in it -- mv2d_xattn_query_order, or mv2d_xattn_cluster_order below).
#include "common.h"

namespace {

constexpr int C = 256, HEADS = 8, QB = 8;
constexpr float LOG2E = 1.4426950408889634f;

typedef q16x8_t xg_q16x8;
union XgFrag { uint4 u; xg_q16x8 v; };
typedef unsigned int xg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void xg_split8(const float4& x0, const float4& x1, XgFrag& hi, XgFrag& lo) {
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_q16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
    hi.u = make_uint4(h[0], h[1], h[2], h[3]);
    lo.u = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void xg_split8_k16(const float4& x0, const float4& x1, XgFrag& hi, XgFrag& lo) {
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_k16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
    hi.u = make_uint4(h[0], h[1], h[2], h[3]);
    lo.u = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ unsigned int xg_lo_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ unsigned int xg_hi_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
#define XG_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float xg_row16_max(float v) {
    v = fmaxf(v, XG_DPP(v, 0xB1));
    v = fmaxf(v, XG_DPP(v, 0x4E));
    v = fmaxf(v, XG_DPP(v, 0x141));
    v = fmaxf(v, XG_DPP(v, 0x140));
    return v;
}

// ------------------------------------------------------------------------------------------------------------------------------
// group tables.  One block per group g: members = the (up to 8) query rows order[slot0 .. slot0 + cnt) of ONE sample (a sample's result must not
// depend on the batch it is in: groups never straddle samples; the bucket-padding rows behind the last sample form groups of their own).
//   g_slot / g_cnt [ng_max]: first slot / members (0: no such group), g_ptr / g_len: the group's union list in ucol / umask (g_ptr a multiple
//   of 16, the list padded to a multiple of 16 with mask 0 entries that name a valid row), sorted by (signature, position) inside every
//   window of GW consecutive positions (a sample's keys span one or two windows).
// Allocation of the lists: one atomic per group (the PLACE of a list varies from run to run, its content does not).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int GW = 16384;                  // positions per window
constexpr int GT = 1024;                   // threads of the table kernel

__device__ __forceinline__ int xg_block_sum(int v, int* red /* [16] */, int tid) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < GT / 64; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(GT) void xattn_group_tables_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const int* __restrict__ order,
                                                                const int* __restrict__ grp_start, int n_grp, int R, int* __restrict__ g_slot,
                                                                int* __restrict__ g_cnt, int* __restrict__ g_ptr, int* __restrict__ g_len,
                                                                int* __restrict__ ucol, unsigned char* __restrict__ umask, int* __restrict__ u_total,
                                                                int ucap, int* __restrict__ flags) {
    __shared__ unsigned int win[GW / 4];                   // membership byte per position of the window
    __shared__ int lpos[GW];
    __shared__ unsigned char lsig[GW];
    __shared__ int red[16], wscan[16];
    __shared__ int meta[8];                                // slot0, cnt, base, n_w
    __shared__ int mrow[QB], mbeg[QB], mend[QB];
    __shared__ unsigned int pres[8];
    __shared__ int cls[256], ccount[256], cbase[256];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int rem = g, slot0 = -1, cnt = 0;
        for (int b = 0; b <= n_grp; ++b) {
            const int lo = grp_start[b], hi = b < n_grp ? grp_start[b + 1] : R;
            const int ngb = (hi - lo + QB - 1) / QB;
            if (rem < ngb) { slot0 = lo + QB * rem; cnt = min(QB, hi - slot0); break; }
            rem -= ngb;
        }
        meta[0] = slot0; meta[1] = cnt;
    }
    __syncthreads();
    const int slot0 = meta[0], cnt = meta[1];
    if (cnt <= 0) {
        if (tid == 0) { g_slot[g] = 0; g_cnt[g] = 0; g_ptr[g] = 0; g_len[g] = 0; }
        return;
    }
    if (tid < QB) {
        const int r = tid < cnt ? (order ? order[slot0 + tid] : slot0 + tid) : -1;
        mrow[tid] = r;
        mbeg[tid] = r >= 0 ? row_ptr[r] : 0;
        mend[tid] = r >= 0 ? row_ptr[r + 1] : 0;
    }
    __syncthreads();
    // range of positions the members list
    int pmin = 0x7fffffff, pmax = -1;
    for (int j = 0; j < cnt; ++j)
        for (int i = mbeg[j] + tid; i < mend[j]; i += GT) { const int p = col_idx[i]; pmin = min(pmin, p); pmax = max(pmax, p); }
    for (int o = 32; o > 0; o >>= 1) { pmin = min(pmin, __shfl_xor(pmin, o, 64)); pmax = max(pmax, __shfl_xor(pmax, o, 64)); }
    if (lane == 0) { red[wave] = pmin; wscan[wave] = pmax; }
    __syncthreads();
    pmin = red[0]; pmax = wscan[0];
#pragma unroll
    for (int w = 1; w < GT / 64; ++w) { pmin = min(pmin, red[w]); pmax = max(pmax, wscan[w]); }
    __syncthreads();
    if (pmax < 0) {                                        // every member's row is empty
        if (tid == 0) { g_slot[g] = slot0; g_cnt[g] = cnt; g_ptr[g] = 0; g_len[g] = 0; }
        return;
    }
    auto mark = [&](int w0) {
        for (int i = tid; i < GW / 4; i += GT) win[i] = 0u;
        __syncthreads();
        for (int j = 0; j < cnt; ++j)
            for (int i = mbeg[j] + tid; i < mend[j]; i += GT) {
                const int off = col_idx[i] - w0;
                if (off >= 0 && off < GW) atomicOr(&win[off >> 2], (1u << j) << (8 * (off & 3)));
            }
        __syncthreads();
    };
    constexpr int PER = GW / GT;                           // 16 positions = 4 words per thread
    // ---- pass 1: the length of the union
    int n_total = 0;
    for (int w0 = pmin; w0 <= pmax; w0 += GW) {
        mark(w0);
        int c = 0;
#pragma unroll
        for (int k = 0; k < PER / 4; ++k) {
            const unsigned int w = win[tid * (PER / 4) + k];
            c += ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
        }
        n_total += xg_block_sum(c, red, tid);
        __syncthreads();
    }
    const int n_pad = (n_total + 15) & ~15;
    if (tid == 0) {
        int base = atomicAdd(u_total, n_pad);
        if (base + n_pad > ucap) { flags[0] = 1; base = -1; }
        meta[2] = base;
    }
    __syncthreads();
    const int base = meta[2];
    if (base < 0) {                                        // capacity exceeded (flagged): the group is left without keys
        if (tid == 0) { g_slot[g] = slot0; g_cnt[g] = cnt; g_ptr[g] = 0; g_len[g] = 0; }
        return;
    }
    // ---- pass 2: per window: compact in position order, then a stable counting sort by signature
    int out_off = 0;
    for (int w0 = pmin; w0 <= pmax; w0 += GW) {
        mark(w0);
        unsigned int wd[PER / 4];
        int c = 0;
#pragma unroll
        for (int k = 0; k < PER / 4; ++k) {
            wd[k] = win[tid * (PER / 4) + k];
            c += ((wd[k] & 0xffu) != 0) + ((wd[k] & 0xff00u) != 0) + ((wd[k] & 0xff0000u) != 0) + ((wd[k] & 0xff000000u) != 0);
        }
        // exclusive scan of c over the block (wave scan, then the 16 wave totals)
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wscan[wave] = inc;
        if (tid < 8) pres[tid] = 0u;
        __syncthreads();
        int wbase = 0, n_w = 0;
#pragma unroll
        for (int w = 0; w < GT / 64; ++w) { if (w < wave) wbase += wscan[w]; n_w += wscan[w]; }
        int dst = wbase + inc - c;
#pragma unroll
        for (int k = 0; k < PER / 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned int s = (wd[k] >> (8 * b)) & 0xffu;
                if (s) {
                    lpos[dst] = w0 + tid * PER + 4 * k + b;
                    lsig[dst] = (unsigned char)s;
                    atomicOr(&pres[s >> 5], 1u << (s & 31));
                    ++dst;
                }
            }
        __syncthreads();
        // the signatures present, ascending
        if (tid < 256) {
            if ((pres[tid >> 5] >> (tid & 31)) & 1u) {
                int idx = __popc(pres[tid >> 5] & ((1u << (tid & 31)) - 1u));
                for (int k = 0; k < (tid >> 5); ++k) idx += __popc(pres[k]);
                cls[idx] = tid;
            }
        }
        int K = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) K += __popc(pres[k]);
        __syncthreads();
        for (int k = wave; k < K; k += GT / 64) {
            const int s = cls[k];
            int cc = 0;
            for (int i = lane; i < n_w; i += 64) cc += lsig[i] == s;
            for (int o = 32; o > 0; o >>= 1) cc += __shfl_xor(cc, o, 64);
            if (lane == 0) ccount[k] = cc;
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < K; ++k) { cbase[k] = run; run += ccount[k]; }
        }
        __syncthreads();
        for (int k = wave; k < K; k += GT / 64) {
            const int s = cls[k];
            int run = base + out_off + cbase[k];
            for (int i0 = 0; i0 < n_w; i0 += 64) {
                const int i = i0 + lane;
                const bool m = i < n_w && lsig[i] == s;
                const unsigned long long bm = __ballot(m);
                if (m) {
                    const int d = run + __popcll(bm & ((1ull << lane) - 1ull));
                    ucol[d] = lpos[i];
                    umask[d] = (unsigned char)s;
                }
                run += __popcll(bm);
            }
        }
        out_off += n_w;
        __syncthreads();
    }
    // padding to a whole tile: mask 0, a valid row
    if (tid < n_pad - n_total) { ucol[base + n_total + tid] = pmax; umask[base + n_total + tid] = 0; }
    if (tid == 0) { g_slot[g] = slot0; g_cnt[g] = cnt; g_ptr[g] = base; g_len[g] = n_total; }
}

// ------------------------------------------------------------------------------------------------------------------------------
// the attention kernel.  Block = one group (up to 8 queries), 8 waves, WAVE = HEAD: the 16 MFMA rows of a wave are (query j, hi | lo part) of
// its head's mapped query Qt_h, so every wave works on every tile of the union (no wave idles while another walks "its" keys; a query that
// does not list a key has that (row, key) slot masked) and the three phases never leave the wave: the head's query map writes the wave's own
// A operand, its context sums z_h feed its own context map.  What the waves share is the ring of key / value tiles.
//   LDS: ring of NST stages [K hi 8 KB | K lo | V hi | V lo] (rows 512 B; key rows XOR-swizzled for the B-fragment reads, value rows
//   chunk-swizzled for the transposing reads -- both permutations ride on the DMA's source addresses), 8 x 8 KB of wave-private scratch
//   (Qt_h, later z_h; aliases stages 2-3 on the hi + lo route), P staging, the union's (row, mask) table of the current chunk.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int NST = 4;                                     // ring stages
constexpr int CH = 4096;                                   // union entries whose (row, mask) a block keeps in LDS at a time
#ifndef MV2D_XG_DBG
#define MV2D_XG_DBG 0                                      // timing experiments only: 1 = no arithmetic in the tile loop, 2 = no DMA
#endif

// LDS-DMA of 16 bytes per lane: lane l of the wave writes LDS bytes [lds_dst + 16 l, + 16) from gbase + voff.  Inline asm: hipcc would put a
// vmcnt(0) in front of every LDS read that follows a __builtin_amdgcn_global_load_lds (it cannot tell which stage the read touches), which
// drains the ring once per tile; an asm load is absent from its bookkeeping, the waits below are counted by hand.
__device__ __forceinline__ void xg_dma16(const void* gbase, unsigned int voff, unsigned int lds_dst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
template <int N> __device__ __forceinline__ void xg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef short xg_s16x4 __attribute__((ext_vector_type(4)));
// B fragment of v_mfma_f32_16x16x16_f16 straight from a row-major [16 keys][256 channels] image: the 16 lanes of group g hand in the addresses of
// keys 4 g + (n >> 2), channels c0 + 4 (n & 3) .. + 3 and get back keys 4 g .. 4 g + 3 of channel c0 + n (ds_read_b64_tr_b16; tools/probes/tr16_probe.hip)
__device__ __forceinline__ uint2 xg_tr_read(const unsigned char* p) {
    const xg_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) xg_s16x4*)p);
    return __builtin_bit_cast(uint2, v);
}

template <bool XLO>
struct XgCfg {
    static constexpr int ST = XLO ? 32768 : 16384;         // bytes of a stage: K hi | (K lo) | V hi | (V lo), 8 KB each
    static constexpr int RING = NST * ST;
    static constexpr int OFF_SCR = 65536;                  // wave-private scratch, 8 KB per wave (hi + lo route: stages 2 and 3)
    static constexpr int AREA = RING > OFF_SCR + QB * 8192 ? RING : OFF_SCR + QB * 8192;
    static constexpr int OFF_PL = AREA, OFF_RQ = OFF_PL + QB * 512, OFF_UC = OFF_RQ + 64, OFF_UM = OFF_UC + CH * 4, SMEM = OFF_UM + CH;
    static constexpr int DPT = XLO ? 4 : 2;                // DMA instructions per wave and tile
    static constexpr int PRE = XLO ? 2 : 3;                // tiles that may be requested while the scratch holds Qt
};

template <bool XLO>
__global__ __launch_bounds__(64 * QB, 2) void xattn_group_kernel(const float* __restrict__ q, const uint4* __restrict__ WA_hi, const uint4* __restrict__ WA_lo,
                                                                const uint4* __restrict__ WB_hi, const uint4* __restrict__ WB_lo, const float* __restrict__ bv,
                                                                const unsigned short* __restrict__ Xk, const unsigned short* __restrict__ Xv,
                                                                const unsigned short* __restrict__ Xk_lo, const unsigned short* __restrict__ Xv_lo,
                                                                const int* __restrict__ row_ptr, const int* __restrict__ order,
                                                                const int* __restrict__ g_slot, const int* __restrict__ g_cnt, const int* __restrict__ g_ptr,
                                                                const int* __restrict__ g_len, const int* __restrict__ ucol,
                                                                const unsigned char* __restrict__ umask, float* __restrict__ ctx, int empty_nan, int ng) {
    using Cfg = XgCfg<XLO>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int h = wave;                                      // this wave's head
    int* rq = reinterpret_cast<int*>(smem + Cfg::OFF_RQ);
    int* uc = reinterpret_cast<int*>(smem + Cfg::OFF_UC);
    unsigned char* um = smem + Cfg::OFF_UM;
    unsigned char* scr = smem + Cfg::OFF_SCR + wave * 8192;
    const int grp = xcd_chunked(blockIdx.x, ng);
    const int nq = g_cnt[grp];
    if (nq <= 0) return;
    const int slot0 = g_slot[grp], ubase = g_ptr[grp], ulen = g_len[grp];
    const int nchunk = (ulen + CH - 1) / CH;
    if (tid < QB) {
        const int s = slot0 + min(tid, nq - 1);
        rq[tid] = order ? order[s] : s;
    }
    auto load_tables = [&](int c) {
        const int ne = ((min(CH, ulen - c * CH) + 15) >> 4) * 16;
        for (int i = tid; i < ne; i += 64 * QB) {
            uc[i] = ucol[ubase + c * CH + i];
            um[i] = umask[ubase + c * CH + i];
        }
    };
    if (nchunk > 0) load_tables(0);
    __syncthreads();
    const unsigned int lds0 = (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int tt) {
#if MV2D_XG_DBG != 2
        // rows 2 w, 2 w + 1 of the tile: lanes 0-31 one row, 32-63 the next; 16-byte chunk c of key row r lands at slot c ^ r, of value row r at
        // slot c ^ 2 (r & 7) (the source address carries the permutation, the DMA writes lane-linear)
        const int row = 2 * wave + (lane >> 5);
        const unsigned int k = (unsigned int)uc[16 * tt + row];
        const unsigned int sl = (unsigned int)(lane & 31);
        const unsigned int koff = (k << 9) + ((sl ^ (unsigned int)row) << 4), voff = (k << 9) + ((sl ^ (2u * (unsigned int)(row & 7))) << 4);
        const unsigned int dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)((tt & (NST - 1)) * Cfg::ST + wave * 1024));
        xg_dma16(Xk, koff, dst);
        if constexpr (XLO) {
            xg_dma16(Xk_lo, koff, dst + 8192);
            xg_dma16(Xv, voff, dst + 16384);
            xg_dma16(Xv_lo, voff, dst + 24576);
        } else {
            xg_dma16(Xv, voff, dst + 8192);
        }
#endif
    };
    int ntile = nchunk > 0 ? (min(CH, ulen) + 15) >> 4 : 0;
#pragma unroll
    for (int t = 0; t < Cfg::PRE; ++t)
        if (t < ntile) issue(t);
    // ---------------------------------------------------------------- phase A: Qt_h of the group's queries (split-precision MFMAs on the fp32 query,
    // packed weights WA from L2; xattn_fused.hip) -> the wave's scratch -> its A operand: row n = (query n & 7, part n >> 3)
    XgFrag qa[8];
    {
        const int r = rq[n & 7];
        const float* qp = q + (long long)r * C + 32 * h + 8 * g;
        XgFrag bh, bl;
        xg_split8(*reinterpret_cast<const float4*>(qp), *reinterpret_cast<const float4*>(qp + 4), bh, bl);
        const uint4* wh = WA_hi + (long long)h * 16 * 64 + lane;
        const uint4* wl = WA_lo + (long long)h * 16 * 64 + lane;
        uint4* qt = reinterpret_cast<uint4*>(scr) + (n & 7) * 64;
        xg_u32x4 wa_h[16], wa_l[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            wa_h[t] = *reinterpret_cast<const xg_u32x4*>(wh + t * 64);
            wa_l[t] = *reinterpret_cast<const xg_u32x4*>(wl + t * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4_t a[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4_t c = {0.f, 0.f, 0.f, 0.f};
                c = mfma_q16_16x16x32(wa_l[2 * u + k], bh.v, c);
                c = mfma_q16_16x16x32(wa_h[2 * u + k], bl.v, c);
                c = mfma_q16_16x16x32(wa_h[2 * u + k], bh.v, c);
                a[k] = c;
            }
            XgFrag hi, lo;
            xg_split8_k16(make_float4(a[0][0], a[0][1], a[0][2], a[0][3]), make_float4(a[1][0], a[1][1], a[1][2], a[1][3]), hi, lo);
            if (n < 8) {
                qt[u * 8 + g * 2] = hi.u;
                qt[u * 8 + g * 2 + 1] = lo.u;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint4* qr = reinterpret_cast<const uint4*>(scr) + (n & 7) * 64 + g * 2 + (n >> 3);
#pragma unroll
        for (int s = 0; s < 8; ++s) qa[s].u = qr[s * 8];
    }
    // ---------------------------------------------------------------- phase B: the union of the group, tile by tile
    float* pl = reinterpret_cast<float*>(smem + Cfg::OFF_PL) + wave * 128;
    float m_run[4], l_run[4];
    f32x4_t Z[16];                                           // Z[ct][i]: row 4 g + i, channel 16 ct + n
#pragma unroll
    for (int i = 0; i < 4; ++i) { m_run[i] = -INFINITY; l_run[i] = 0.f; }
#pragma unroll
    for (int u = 0; u < 16; ++u) Z[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int qbit = 4 * (g & 1);                            // row 4 g + i belongs to query 4 (g & 1) + i
    // per-lane offsets of the operand reads inside a stage
    const int koffs = n * 512;                               // key row n; chunk (4 s + g) ^ n
    const int vkey = 4 * g + (n >> 2);
    const int voffs = vkey * 512 + (n & 1) * 8, vsw = 2 * (vkey & 7), vch = (n & 3) >> 1;

    auto compute = [&](int tt) {
        const unsigned char* stg = smem + (tt & (NST - 1)) * Cfg::ST;
        const unsigned char* kh = stg + koffs;
        const unsigned char* vh = stg + (XLO ? 16384 : 8192) + voffs;
        const unsigned int mb = um[16 * tt + n];
        f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int sl = ((4 * s + g) ^ n) << 4;
            const uint4 kb = *reinterpret_cast<const uint4*>(kh + sl);
            sacc = mfma_k16_16x16x32(qa[s].u, kb, sacc);
            if (XLO) {
                const uint4 kl = *reinterpret_cast<const uint4*>(kh + 8192 + sl);
                sacc = mfma_k16_16x16x32(qa[s].u, kl, sacc);      // (hi + lo rows of Qt) x K lo: the lo x lo term rides along
            }
        }
        float sv[4], p[4], alpha[4];
        bool resc = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sacc[i]), __float_as_uint(sacc[i]), false, false);
            const float full = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            sv[i] = ((mb >> (qbit + i)) & 1u) ? full * LOG2E : -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float tm = xg_row16_max(sv[i]);
            const float m_new = fmaxf(m_run[i], tm);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;          // a query that lists no key so far: p = 0, nothing to rescale
            const bool moved = m_new != m_run[i];
            alpha[i] = moved ? __builtin_amdgcn_exp2f(m_run[i] - m_use) : 1.f;
            resc = resc || moved;
            p[i] = __builtin_amdgcn_exp2f(sv[i] - m_use);
            l_run[i] = l_run[i] * alpha[i] + p[i];
            m_run[i] = m_new;
        }
        if (g < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pl[(4 * g + i) * 16 + n] = p[i];
        }
        __builtin_amdgcn_wave_barrier();
        uint2 pa;
        {
            const float4 pv = *reinterpret_cast<const float4*>(pl + (n & 7) * 16 + 4 * g);
            unsigned int h0, h1, l0, l1;
            split_k16x2_bounded(pv.x, pv.y, h0, l0);
            split_k16x2_bounded(pv.z, pv.w, h1, l1);
            pa = n < 8 ? make_uint2(h0, h1) : make_uint2(l0, l1);
        }
        if (__ballot(resc) != 0ull) {                         // (wave-uniform: a scalar branch; the maxima stop moving after a row's first tiles)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) Z[u][i] *= alpha[i];
        }
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const int sl = ((2 * ct + vch) ^ vsw) << 4;
            f32x4_t zc = Z[ct];
            zc = mfma_k16_16x16x16(pa, xg_tr_read(vh + sl), zc);
            if (XLO) zc = mfma_k16_16x16x16(pa, xg_tr_read(vh + 8192 + sl), zc);      // (hi + lo rows of P) x V lo
            Z[ct] = zc;
        }
        __builtin_amdgcn_wave_barrier();
    };

    __syncthreads();                                         // every wave holds its operand: the scratch (stages 2, 3 on the hi + lo route) may be overwritten
    for (int c = 0; c < nchunk; ++c) {
        if (c > 0) {
            __syncthreads();                                 // every wave is done with the table and the ring
            load_tables(c);
            ntile = (min(CH, ulen - c * CH) + 15) >> 4;
            __syncthreads();
#pragma unroll
            for (int t = 0; t < Cfg::PRE; ++t)
                if (t < ntile) issue(t);
        }
        if (Cfg::PRE < 3 && ntile > 2) issue(2);
        for (int tt = 0; tt < ntile; ++tt) {
            const int rem = ntile - 1 - tt;
            // this wave's pieces of tile tt have landed (loads return in order; tiles tt + 1, tt + 2 may stay in flight) ...
            if (rem >= 2) xg_wait_vm<2 * Cfg::DPT>();
            else if (rem == 1) xg_wait_vm<Cfg::DPT>();
            else xg_wait_vm<0>();
            __builtin_amdgcn_s_barrier();                    // ... and everybody's; every wave is past tile tt - 1, whose stage tile tt + 3 takes
            if (tt + 3 < ntile) issue(tt + 3);
#if MV2D_XG_DBG != 1
            compute(tt);
#endif
        }
    }
    __syncthreads();                                         // the ring is idle: the scratch takes z
    // ---- z_h / l of the 8 queries -> the wave's scratch ([query][256] fp32)
    {
        float rl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float l = l_run[i];
            l += __shfl_xor(l, 1, 64);
            l += __shfl_xor(l, 2, 64);
            l += __shfl_xor(l, 4, 64);
            l += __shfl_xor(l, 8, 64);
            rl[i] = __builtin_amdgcn_rcpf(l);
        }
        float* zs = reinterpret_cast<float*>(scr);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // hi rows (lanes 0-31) + lo rows (lanes 32-63): afterwards lanes g < 2 hold column tile ct, lanes g >= 2 column tile ct + 8 of query 4 (g & 1) + i
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Z[ct][i]), __float_as_uint(Z[ct + 8][i]), false, false);
                const float v = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * rl[i];
                zs[(qbit + i) * C + 16 * (ct + 8 * (g >> 1)) + n] = v;
            }
        __builtin_amdgcn_wave_barrier();
    }
    // ---------------------------------------------------------------- phase C: ctx[:, 32 h .. 32 h + 31] = Wv_h z_h + bv (xattn_fused.hip)
    {
        const int j = n & 7;
        const float* zp = reinterpret_cast<const float*>(scr) + j * C + 8 * g;
        const uint4* wh = WB_hi + (long long)h * 16 * 64 + lane;
        const uint4* wl = WB_lo + (long long)h * 16 * 64 + lane;
        f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        xg_u32x4 wb_h[16], wb_l[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            wb_h[t] = *reinterpret_cast<const xg_u32x4*>(wh + t * 64);
            wb_l[t] = *reinterpret_cast<const xg_u32x4*>(wl + t * 64);
        }
        int rr_[4], rp0_[4], rp1_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rr_[i] = rq[(4 * g + i) & 7];
            rp0_[i] = row_ptr[rr_[i]];
            rp1_[i] = row_ptr[rr_[i] + 1];
        }
        const float bv0 = bv[32 * h + n], bv1 = bv[32 * h + 16 + n];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 x0 = *reinterpret_cast<const float4*>(zp + 32 * s);
            const float4 x1 = *reinterpret_cast<const float4*>(zp + 32 * s + 4);
            XgFrag ah, al;
            xg_split8(x0, x1, ah, al);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[nt] = mfma_q16_16x16x32(al.v, wb_h[s * 2 + nt], acc[nt]);
                acc[nt] = mfma_q16_16x16x32(ah.v, wb_l[s * 2 + nt], acc[nt]);
                acc[nt] = mfma_q16_16x16x32(ah.v, wb_h[s * 2 + nt], acc[nt]);
            }
        }
        if (g < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int js = 4 * g + i;
                if (js < nq) {
                    const int rr = rr_[i];
                    const bool empty = rp1_[i] <= rp0_[i];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int col = 32 * h + 16 * nt + n;
                        float v = acc[nt][i] + (nt ? bv1 : bv0);
                        if (empty) v = empty_nan ? __uint_as_float(0x7fc00000u) : 0.f;
                        ctx[(long long)rr * C + col] = v;
                    }
                }
            }
        }
    }
}

}  // namespace

// C-ABI: include/mv2d_hip.h
extern "C" int mv2d_xattn_group_max(int R, int n_samples) { return (R + QB - 1) / QB + n_samples + 1; }

extern "C" int mv2d_xattn_group_tables(const int* row_ptr, const int* col_idx, const int* order, const int* grp_start, int n_samples, int R, int ng_max,
                                       int* g_slot, int* g_cnt, int* g_ptr, int* g_len, int* ucol, unsigned char* umask, int ucap, int* u_total,
                                       int* flags, void* stream) {
    MV2D_CHECK_ARG(row_ptr && col_idx && grp_start && g_slot && g_cnt && g_ptr && g_len && ucol && umask && u_total && flags,
                   "mv2d_xattn_group_tables: null pointer");
    MV2D_CHECK_ARG(R > 0 && n_samples >= 1 && ucap >= 16, "mv2d_xattn_group_tables: bad sizes");
    MV2D_CHECK_ARG(ng_max >= mv2d_xattn_group_max(R, n_samples), "mv2d_xattn_group_tables: ng_max < mv2d_xattn_group_max(R, n_samples)");
    hipLaunchKernelGGL(xattn_group_tables_kernel, dim3(ng_max), dim3(GT), 0, (hipStream_t)stream, row_ptr, col_idx, order, grp_start, n_samples, R, g_slot,
                       g_cnt, g_ptr, g_len, ucol, umask, u_total, ucap, flags);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_xattn_group_fwd(const float* q, const void* WA_hi, const void* WA_lo, const void* WB_hi, const void* WB_lo, const float* bv,
                                    const void* Xk, const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr, const int* order,
                                    const int* g_slot, const int* g_cnt, const int* g_ptr, const int* g_len, const int* ucol, const unsigned char* umask,
                                    float* ctx, int ng_max, int empty_nan, void* stream) {
    MV2D_CHECK_ARG(q && WA_hi && WA_lo && WB_hi && WB_lo && bv && Xk && Xv && row_ptr && g_slot && g_cnt && g_ptr && g_len && ucol && umask && ctx,
                   "mv2d_xattn_group_fwd: null pointer");
    MV2D_CHECK_ARG((Xk_lo == nullptr) == (Xv_lo == nullptr), "mv2d_xattn_group_fwd: Xk_lo and Xv_lo come together");
    MV2D_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)WA_hi & 15) == 0 && ((uintptr_t)WA_lo & 15) == 0 && ((uintptr_t)WB_hi & 15) == 0 &&
                       ((uintptr_t)WB_lo & 15) == 0 && ((uintptr_t)Xk & 15) == 0 && ((uintptr_t)Xv & 15) == 0 && ((uintptr_t)Xk_lo & 15) == 0 &&
                       ((uintptr_t)Xv_lo & 15) == 0, "mv2d_xattn_group_fwd: operands must be 16-byte aligned");
    if (ng_max <= 0) return MV2D_OK;
    const dim3 grid(ng_max), block(64 * QB);
    if (Xk_lo)
        hipLaunchKernelGGL((xattn_group_kernel<true>), grid, block, 0, (hipStream_t)stream, q, (const uint4*)WA_hi, (const uint4*)WA_lo, (const uint4*)WB_hi,
                           (const uint4*)WB_lo, bv, (const unsigned short*)Xk, (const unsigned short*)Xv, (const unsigned short*)Xk_lo,
                           (const unsigned short*)Xv_lo, row_ptr, order, g_slot, g_cnt, g_ptr, g_len, ucol, umask, ctx, empty_nan, ng_max);
    else
        hipLaunchKernelGGL((xattn_group_kernel<false>), grid, block, 0, (hipStream_t)stream, q, (const uint4*)WA_hi, (const uint4*)WA_lo, (const uint4*)WB_hi,
                           (const uint4*)WB_lo, bv, (const unsigned short*)Xk, (const unsigned short*)Xv, (const unsigned short*)Xk_lo,
                           (const unsigned short*)Xv_lo, row_ptr, order, g_slot, g_cnt, g_ptr, g_len, ucol, umask, ctx, empty_nan, ng_max);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
