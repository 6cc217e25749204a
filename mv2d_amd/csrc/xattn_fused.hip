// Cross attention of a decoder layer in ONE launch (round 5): query map -> tile attention -> context map, fused per block of 8 queries.
//
// Replaces the three launches of csrc/xattn_tile.hip (xattn_qmap_kernel, xattn_tile_kernel, xattn_ctxmap_kernel) for
// PETRMultiheadAttention's core (MU/petr_transformer.py:426-513) on rows of similar length (the S path: a query reads the 49 cells of its own
// RoI and of the few RoIs matched to it).  The arithmetic is the same, operation for operation (raw key space, see xattn_tile.hip: the K / V
// in_proj ride on the query side, the key side reads the unprojected key16 hi + lo rows), so the results are BITWISE those of the three
// kernels with one wave per query (tests/test_gpu_kernels.py::test_xattn_fused_equals_the_three_kernels); what disappears are the two
// intermediates of that decomposition -- Qt (8 KB per query: the mapped query as a 16 x 256 key16 MFMA operand) and z (8 KB: the per-head
// context sums in the key space) -- which the three kernels write to and read back from HBM: 16 of the ~121 KB a query of the index-exact
// route moves per layer, and two launches of six.
//
// Block = 8 queries, 8 waves (two per SIMD), LDS 132 KB:
//   phase A  wave = head h: Qt_h of the 8 queries (split-precision MFMAs on the fp32 query, packed weights WA from L2) -> LDS [query][head][1 KB]
//   phase B  wave = query w: its Qt operand into 32 registers, then the tile loop of xattn_tile_kernel<1, false, XLO> over its CSR row (key
//            tiles of 16 rows through the wave's 16 KB of LDS -- which alias the Qt area once every wave holds its operand); the un-normalised
//            context sums z and the softmax denominators go to LDS
//   phase C  wave = head h: ctx[:, 32 h .. 32 h + 31] = Wv_h z_h / l_h + bv for the 8 queries (packed weights WB from L2), NaN / 0 for a row
//            without a key, written to HBM as the [R, 256] fp32 rows the out-projection kernel reads.
// A block streams the packed weights of both maps (512 KB of hi + lo fragments) from L2 once per 8 queries = 64 KB per query against the
// ~105 KB of key / value rows it gathers from HBM; the blocks of a launch are out of phase after the first round, so the map phases of one
// CU overlap the gathers of the others.  Rows of very different length (the T path: 1 .. 400 keys) would wait for the longest of the 8 at
// the barrier between B and C: the engine keeps the three kernels there.
// Round 6 experiment, OFF: MODE.FP16_OVFL for the whole kernel (mv2d_set_f16_ovfl at its entry; -DMV2D_XF_OVFL=1): the fp32 -> fp16 conversions of the hi / lo splits in
// the two map phases then clamp an overflow themselves, which takes 3 of the ~11 vector instructions per split pair away: 98.0 -> 95.8 us per cfg2_s launch, in-range
// results bit for bit.  But with the bit set a NaN in a key row, a value row or the query no longer reaches the output (the conversions do keep it --
// tools/probes/f16_ovfl_probe.hip -- so it is lost in the f16 MFMAs), and a poisoned input frame must stay visible (tests/test_gpu_engine.py::
// test_poisoned_feature_cell_stays_visible, tools/gpu_jobs/nan_dbg.py).  Toggling the bit around every split would fence the MFMAs off from the splits.
#ifndef MV2D_XF_OVFL
#define MV2D_XF_OVFL 0
#endif
#if MV2D_XF_OVFL
#define MV2D_F16_OVFL_MODE 1
#endif
#include "common.h"
#include <stdlib.h>
#ifndef MV2D_XF_QB_DEFAULT
#define MV2D_XF_QB_DEFAULT 8
#endif
#ifndef MV2D_XF_WBATCH
#define MV2D_XF_WBATCH 0
#endif
#ifndef MV2D_XF_PIPE
#define MV2D_XF_PIPE 0          // 1: e4m3 lo rows, the key rows of tile t + 1 requested in front of tile t's arithmetic -- measured SLOWER (see the loop), off
#endif

#ifdef MV2D_XF_TRACE
__device__ long long g_xf_trace[32];
#define XF_STAMP(i) do { if (blockIdx.x == MV2D_XF_TRACE && threadIdx.x == 0) g_xf_trace[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define XF_STAMP(i) do {} while (0)
#endif

namespace {

constexpr int C = 256, HEADS = 8;                            // (QB = queries per block = waves per block: a template parameter, 8 or 4)
constexpr float LOG2E = 1.4426950408889634f;

typedef q16x8_t xf_q16x8;
union XfFrag { uint4 u; xf_q16x8 v; };
typedef unsigned int xf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int xf_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void xf_split8(const float4& x0, const float4& x1, XfFrag& hi, XfFrag& lo) {
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_q16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
    hi.u = make_uint4(h[0], h[1], h[2], h[3]);
    lo.u = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void xf_split8_k16(const float4& x0, const float4& x1, XfFrag& hi, XfFrag& lo) {
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    unsigned int h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_k16x2(f[2 * i], f[2 * i + 1], h[i], l[i]);
    hi.u = make_uint4(h[0], h[1], h[2], h[3]);
    lo.u = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ unsigned int xf_lo_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ unsigned int xf_hi_pair(unsigned int a, unsigned int b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
#define XF_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float xf_row16_max(float v) {
    v = fmaxf(v, XF_DPP(v, 0xB1));
    v = fmaxf(v, XF_DPP(v, 0x4E));
    v = fmaxf(v, XF_DPP(v, 0x141));
    v = fmaxf(v, XF_DPP(v, 0x140));
    return v;
}

constexpr int WAVE_LDS = 16384;                              // per wave: key tile hi (8 KB) | key tile lo (8 KB); phase A / C: Qt / z of query `wave` in the first 8 KB

// QB = 8: one block per CU (132 KB of LDS); QB = 4 (round 6): 66 KB, two blocks per CU -- the map phases of one block (packed weights from L2, the
// gather idle) run under the tile loop of the other; a wave then maps two heads in phases A and C
// XLO: 0 = key16 rows alone, 1 = hi + key16 lo rows, 2 = hi + e4m3 lo rows (common.h "lo8"; see xattn_tile_kernel)
template <int XLO, int QB>
__global__ __launch_bounds__(64 * QB, 2) void xattn_fused_kernel(const float* __restrict__ q, const uint4* __restrict__ WA_hi, const uint4* __restrict__ WA_lo,
                                                                const uint4* __restrict__ WB_hi, const uint4* __restrict__ WB_lo, const float* __restrict__ bv,
                                                                const unsigned short* __restrict__ Xk, const unsigned short* __restrict__ Xv,
                                                                const unsigned short* __restrict__ Xk_lo, const unsigned short* __restrict__ Xv_lo,
                                                                const int* __restrict__ row_ptr, const int* __restrict__ col_idx, float* __restrict__ ctx,
                                                                int R, int empty_nan, const int* __restrict__ order, int nblk) {
    constexpr int SMEM = QB * WAVE_LDS + QB * 512 + QB * HEADS * 4 + 3 * QB * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
#if MV2D_XF_OVFL
    mv2d_set_f16_ovfl();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    float* lsum = reinterpret_cast<float*>(smem + QB * WAVE_LDS + QB * 512);         // [query][head] softmax denominators
    int* rq = reinterpret_cast<int*>(smem + QB * WAVE_LDS + QB * 512 + QB * HEADS * 4);       // [query slot] -> query row, then the ends of its CSR row
    int* rbeg = rq + QB;
    int* rend = rq + 2 * QB;
    // XCD-chunked block order (block b runs on XCD b % 8): every XCD works through one contiguous range of query slots; optional launch order
    // of the queries (the S path ranks them by the smallest RoI they list, so that matched RoIs share an L2).  Speed only.
    // The R query slots are dealt EVENLY to nblk >= ceil(R / 8) blocks (the host rounds nblk up to a multiple of the CU count when that leaves >= 4
    // queries per block: 4800 queries = 600 blocks of 8 are 2.34 rounds of 256 one-block-per-CU slots, and the third round costs nearly a full one;
    // 768 blocks of 6-7 queries are three full rounds of 0.78 x the work each: 140 -> 112 us per layer).
    const int blk = xcd_chunked(blockIdx.x, nblk);
    const int slot0 = (int)((long long)blk * R / nblk), nq = (int)((long long)(blk + 1) * R / nblk) - slot0;
    if (nq <= 0) return;
    XF_STAMP(0);
    if (tid < QB) {
        const int s = slot0 + min(tid, nq - 1);
        const int r_ = order ? order[s] : s;
        rq[tid] = r_;
        rbeg[tid] = row_ptr[r_];
        rend[tid] = row_ptr[r_ + 1];
    }
    __syncthreads();
    XF_STAMP(1);
    // Round 6 (per-phase stamps, tools/xf_trace.py: 43 % of a block's time was the wait for a tile's key rows, the first tile's behind three dependent round
    // trips -- row ends, key indices, rows -- that only started after phase A): the wave's CSR row ends come with the slot table, the indices of its first
    // tile are requested in front of phase A and the hi + lo key rows of that tile in the middle of it, so that they travel under the query maps.
    const int r = rq[wave];
    const int beg = rbeg[wave], end = rend[wave];
    const int ntile = wave < nq ? (end - beg + 15) >> 4 : 0;          // (waves beyond the block's queries: no tiles; their z / l are never read)
    int idx_next = ntile > 0 ? col_idx[min(beg + n, end - 1)] : 0;
    int idx_nx1 = (XLO == 2 && MV2D_XF_PIPE && ntile > 1) ? col_idx[min(beg + 16 + n, end - 1)] : 0;      // (pipelined tile loop: the indices run two tiles ahead)
    xf_u32x4 kreg0[8], klo0[XLO == 1 ? 8 : 1];
    xf_u32x2 klo0b[XLO == 2 ? 8 : 1];
    auto load_k0 = [&](const unsigned short* K_, int myidx, xf_u32x4 (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned int ridx = (unsigned int)__shfl(myidx, 2 * i + (lane >> 5), 64);
            dst[i] = *reinterpret_cast<const xf_u32x4*>(reinterpret_cast<const char*>(K_) + ((ridx << 9) + (unsigned)(lane & 31) * 16u));
        }
    };
    // e4m3 lo rows: the same lane -> (row, channels) assignment at half the bytes (a half wave reads one 256-byte row)
    auto load_k8 = [&](const unsigned short* K_, int myidx, xf_u32x2 (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned int ridx = (unsigned int)__shfl(myidx, 2 * i + (lane >> 5), 64);
            dst[i] = *reinterpret_cast<const xf_u32x2*>(reinterpret_cast<const char*>(K_) + ((ridx << 8) + (unsigned)(lane & 31) * 8u));
        }
    };
    // ---------------------------------------------------------------- phase A: query maps, wave = head
#ifdef MV2D_XF_NOMAPS      // timing experiment: the tile loops alone (no map phases, garbage results).  Round 6, e4m3 lo rows, idle GPU: 76.4 us per cfg2_s launch
                           // against 99.3 with the maps (cfg2_s_nc6: 153.6 / 178.4): the gathers alone move their 371 MB at 4.86 TB/s -- the rate every row-gather
                           // kernel of this library tops out at (0.77 of the 6.3 TB/s a streaming read reaches) -- and the two map phases cost 23-25 us per launch
    if (ntile > 0) {
        load_k0(Xk, idx_next, kreg0);
        if constexpr (XLO == 1) load_k0(Xk_lo, idx_next, klo0);
        if constexpr (XLO == 2) load_k8(Xk_lo, idx_next, klo0b);
    }
    for (int h = wave; h < 0; h += QB) {
#else
    for (int h = wave; h < HEADS; h += QB) {
#endif
        const int r = rq[n & (QB - 1)];
        const float* qp = q + (long long)r * C + 32 * h + 8 * g;
        // (the query values are REQUESTED here and split behind the weight requests: split first, hipcc waited for them -- a full round trip -- before it issued
        //  the first weight load: 3.4 k of a block's 62 k cycles, round 6 stamps)
        const float4 q0 = *reinterpret_cast<const float4*>(qp), q1 = *reinterpret_cast<const float4*>(qp + 4);
        const uint4* wh = WA_hi + (long long)h * 16 * 64 + lane;
        const uint4* wl = WA_lo + (long long)h * 16 * 64 + lane;
        uint4* qt = reinterpret_cast<uint4*>(smem + (n & (QB - 1)) * WAVE_LDS) + h * 64;      // this lane's query, this head: 64 chunks of 16 B
        // ALL 32 weight fragments of the head are requested before the first MFMA (128 registers, free in this phase): the first build left the
        // loads next to their MFMAs and the ISA showed 24 serialised L2 round trips per block and phase (tools/isa_waits.sh)
        xf_u32x4 wa_h[16], wa_l[16];
        auto request = [&](int t0, int t1) {
#pragma unroll
            for (int t = t0; t < t1; ++t) {
                wa_h[t] = *reinterpret_cast<const xf_u32x4*>(wh + t * 64);
                wa_l[t] = *reinterpret_cast<const xf_u32x4*>(wl + t * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#if MV2D_XF_WBATCH
        // (experiment: the fragments in three batches, the later ones requested between the MFMA groups -- issuing 32 x 1 KB per wave is itself 3.7 k cycles at the
        //  L1's 64 B / clk, during which the wave computes nothing)
        request(0, 8);
#else
        request(0, 16);
#endif
        XF_STAMP(20);
        XfFrag bh, bl;
        xf_split8(q0, q1, bh, bl);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u == 1) XF_STAMP(21);
            if (u == 4) XF_STAMP(22);
#if MV2D_XF_WBATCH
            if (u == 0) request(8, 12);
            if (u == 2) request(12, 16);
#endif
            f32x4_t a[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4_t c = {0.f, 0.f, 0.f, 0.f};
                c = mfma_q16_16x16x32(wa_l[2 * u + k], bh.v, c);
                c = mfma_q16_16x16x32(wa_h[2 * u + k], bl.v, c);
                c = mfma_q16_16x16x32(wa_h[2 * u + k], bh.v, c);
                a[k] = c;
            }
            XfFrag hi, lo;
            xf_split8_k16(make_float4(a[0][0], a[0][1], a[0][2], a[0][3]), make_float4(a[1][0], a[1][1], a[1][2], a[1][3]), hi, lo);
            if (n < QB) {
                qt[u * 8 + g * 2] = hi.u;
                qt[u * 8 + g * 2 + 1] = lo.u;
            }
            if (u == (XLO ? 5 : 3) && h == wave && ntile > 0) {   // (all weight fragments of the head have been requested: the rows queue behind them; late enough for their 64 registers)
                load_k0(Xk, idx_next, kreg0);
                if constexpr (XLO == 1) load_k0(Xk_lo, idx_next, klo0);
                if constexpr (XLO == 2) load_k8(Xk_lo, idx_next, klo0b);
            }
        }
    }
    XF_STAMP(23);
    // (LDS only: __syncthreads() would also wait for the key rows that are still in flight -- vmcnt(0) -- and put their latency back in front of phase B)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    XF_STAMP(2);
    // ---------------------------------------------------------------- phase B: tile attention, wave = query
    (void)r;
    XfFrag qa[8];
    {
        const uint4* qp = reinterpret_cast<const uint4*>(smem + wave * WAVE_LDS) + (n & 7) * 64 + g * 2 + (n >> 3);
#pragma unroll
        for (int s = 0; s < 8; ++s) qa[s].u = qp[s * 8];
    }
    // (a wave's operand lies in its OWN 16 KB area, which only the wave itself overwrites with key tiles below: no barrier needed here)
    uint4* kt = reinterpret_cast<uint4*>(smem + wave * WAVE_LDS);
    uint4* kt2 = kt + 512;
    float* pl = reinterpret_cast<float*>(smem + QB * WAVE_LDS) + wave * 128;
    float m_run[4], l_run[4];
    f32x4_t Z[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m_run[i] = -INFINITY; l_run[i] = 0.f; }
#pragma unroll
    for (int u = 0; u < 16; ++u) Z[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        auto load_v = [&](const unsigned short* V_, int myidx, xf_u32x4 (&dst)[4][2]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int vidx = (unsigned int)__shfl(myidx, 4 * g + e, 64);
                const char* vp = reinterpret_cast<const char*>(V_) + ((vidx << 9) + 16u * (unsigned)n);
                dst[e][0] = *reinterpret_cast<const xf_u32x4*>(vp);
                dst[e][1] = *reinterpret_cast<const xf_u32x4*>(vp + 256);
            }
        };
        auto load_k = [&](const unsigned short* K_, int myidx, xf_u32x4 (&dst)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned int ridx = (unsigned int)__shfl(myidx, 2 * i + (lane >> 5), 64);
                dst[i] = *reinterpret_cast<const xf_u32x4*>(reinterpret_cast<const char*>(K_) + ((ridx << 9) + (unsigned)(lane & 31) * 16u));
            }
        };
        auto store_k = [&](uint4* tile, const xf_u32x4 (&src)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rowi = 2 * i + (lane >> 5);
                reinterpret_cast<xf_u32x4*>(tile)[rowi * 32 + ((lane & 31) ^ (rowi & 15))] = src[i];
            }
        };
        auto load_v8 = [&](const unsigned short* V_, int myidx, xf_u32x2 (&dst)[4][2]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int vidx = (unsigned int)__shfl(myidx, 4 * g + e, 64);
                const char* vp = reinterpret_cast<const char*>(V_) + ((vidx << 8) + 8u * (unsigned)n);
                dst[e][0] = *reinterpret_cast<const xf_u32x2*>(vp);
                dst[e][1] = *reinterpret_cast<const xf_u32x2*>(vp + 128);
            }
        };
        auto store_k8 = [&](uint4* tile, const xf_u32x2 (&src)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rowi = 2 * i + (lane >> 5);
                tile[rowi * 32 + ((lane & 31) ^ (rowi & 15))] = lo8_chunk(make_uint2(src[i].x, src[i].y));
            }
        };
        auto compute = [&](int tt, const xf_u32x4 (&vreg)[4][2], const auto& vlo) {
            const int kbase = beg + 16 * tt;
            f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                XfFrag kb;
                kb.u = kt[n * 32 + ((4 * s + g) ^ n)];
                sacc = mfma_k16_16x16x32(qa[s].u, kb.u, sacc);
                if (XLO) {
                    XfFrag kl, qh;
                    kl.u = kt2[n * 32 + ((4 * s + g) ^ n)];
                    qh.u = n < 8 ? qa[s].u : make_uint4(0u, 0u, 0u, 0u);
                    sacc = mfma_k16_16x16x32(qh.u, kl.u, sacc);
                }
            }
            const bool valid = kbase + n < end;
            float sv[4], p[4], alpha[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sacc[i]), __float_as_uint(sacc[i]), false, false);
                const float full = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                sv[i] = valid ? full * LOG2E : -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float tm = sv[i];
                tm = xf_row16_max(tm);
                const float m_new = fmaxf(m_run[i], tm);
                alpha[i] = __builtin_amdgcn_exp2f(m_run[i] - m_new);
                p[i] = __builtin_amdgcn_exp2f(sv[i] - m_new);
                l_run[i] = l_run[i] * alpha[i] + p[i];
                m_run[i] = m_new;
            }
            if (g < 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pl[(4 * g + i) * 16 + n] = p[i];
            }
            __builtin_amdgcn_wave_barrier();
            uint2 pa, pah;
            {
                const float4 pv = *reinterpret_cast<const float4*>(pl + (n & 7) * 16 + 4 * g);
                unsigned int h0, h1, l0, l1;
                split_k16x2_bounded(pv.x, pv.y, h0, l0);
                split_k16x2_bounded(pv.z, pv.w, h1, l1);
                pa = n < 8 ? make_uint2(h0, h1) : make_uint2(l0, l1);
                pah = n < 8 ? make_uint2(h0, h1) : make_uint2(0u, 0u);
            }
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
                    const uint2 vb = (w & 1) ? make_uint2(xf_hi_pair(r0[d], r1[d]), xf_hi_pair(r2[d], r3[d]))
                                             : make_uint2(xf_lo_pair(r0[d], r1[d]), xf_lo_pair(r2[d], r3[d]));
                    f32x4_t zc = Z[H * 8 + w];
                    zc = mfma_k16_16x16x16(pa, vb, zc);
                    if constexpr (XLO != 0) {
                        auto lo_pair_of = [&](int e) -> unsigned int {
                            if constexpr (XLO == 2) {
                                const unsigned int b = vlo[e][H][d >> 1];
                                return (d & 1) ? lo8_pair<1>(b) : lo8_pair<0>(b);
                            } else {
                                return vlo[e][H][d];
                            }
                        };
                        const unsigned int q0 = lo_pair_of(0), q1 = lo_pair_of(1), q2 = lo_pair_of(2), q3 = lo_pair_of(3);
                        const uint2 vl = (w & 1) ? make_uint2(xf_hi_pair(q0, q1), xf_hi_pair(q2, q3)) : make_uint2(xf_lo_pair(q0, q1), xf_lo_pair(q2, q3));
                        zc = mfma_k16_16x16x16(pah, vl, zc);
                    }
                    Z[H * 8 + w] = zc;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        // the first tile's key rows were requested during phase A: into LDS right away (the wave's Qt operand sits in qa by now)
        if (ntile > 0) {
            store_k(kt, kreg0);
            if constexpr (XLO == 1) store_k(kt2, klo0);
            if constexpr (XLO == 2) store_k8(kt2, klo0b);
        }
        if constexpr (XLO == 2 && MV2D_XF_PIPE != 0) {
            // Round 6, e4m3 lo rows: a tile's rows are 48 + 48 staging registers instead of 64 + 64, so the key rows of tile t + 1 (hi + lo: 16 loads) are
            // requested right behind the value rows of tile t and travel under tile t's logits, softmax and P.V; they go to LDS when the tile is done with
            // its own.  One exposed round trip per tile (the value rows, partly under the logits) instead of two; the indices run two tiles ahead.
            // MEASURED (same box, idle GPU): 110.6 us per cfg2_s launch against 98.6 us for the two-phase order below, 194.8 against 178.3 at cfg2_s_nc6 (256
            // registers with 16 loop-invariant values in scratch) -- like round 4's software pipelining of the tile kernel, more rows in flight per wave buy
            // nothing: the launch sits at what the memory system delivers for row gathers, not at a per-wave latency chain.  Compiled out (MV2D_XF_PIPE=1).
            int idx_cur = idx_next, idx_nx = idx_nx1;
            for (int tt = 0; tt < ntile; ++tt) {
                const int idx_nn = tt + 2 < ntile ? col_idx[min(beg + 16 * (tt + 2) + n, end - 1)] : 0;
                xf_u32x4 vreg[4][2], kreg[8];
                xf_u32x2 vlo[4][2], klo[8];
                load_v(Xv, idx_cur, vreg);
                load_v8(Xv_lo, idx_cur, vlo);
                const bool more = tt + 1 < ntile;
                if (more) {
                    load_k(Xk, idx_nx, kreg);
                    load_k8(Xk_lo, idx_nx, klo);
                }
                __builtin_amdgcn_wave_barrier();
                XF_STAMP(3 + 2 * min(tt, 5));
                compute(tt, vreg, vlo);
                XF_STAMP(4 + 2 * min(tt, 5));
                if (more) {
                    store_k(kt, kreg);
                    store_k8(kt2, klo);
                }
                idx_cur = idx_nx;
                idx_nx = idx_nn;
            }
        } else
        for (int tt = 0; tt < ntile; ++tt) {
            const int myidx = idx_next;
            if (tt + 1 < ntile) idx_next = col_idx[min(beg + 16 * (tt + 1) + n, end - 1)];
            if constexpr (XLO == 2) {
                xf_u32x4 vreg[4][2];
                xf_u32x2 vlo[4][2];
                if (tt > 0) {
                    xf_u32x4 kreg[8];
                    xf_u32x2 klo[8];
                    load_k(Xk, myidx, kreg);
                    load_k8(Xk_lo, myidx, klo);
                    store_k(kt, kreg);
                    store_k8(kt2, klo);
                }
                load_v(Xv, myidx, vreg);
                load_v8(Xv_lo, myidx, vlo);
                __builtin_amdgcn_wave_barrier();
                XF_STAMP(3 + 2 * min(tt, 5));
                compute(tt, vreg, vlo);
                XF_STAMP(4 + 2 * min(tt, 5));
            } else if constexpr (XLO == 1) {
                xf_u32x4 vreg[4][2], vlo[4][2];
                if (tt > 0) {
                    xf_u32x4 kreg[8], klo[8];
                    load_k(Xk, myidx, kreg);
                    load_k(Xk_lo, myidx, klo);
                    store_k(kt, kreg);
                    store_k(kt2, klo);
                }
                load_v(Xv, myidx, vreg);
                load_v(Xv_lo, myidx, vlo);
                __builtin_amdgcn_wave_barrier();
                XF_STAMP(3 + 2 * min(tt, 5));
                compute(tt, vreg, vlo);
                XF_STAMP(4 + 2 * min(tt, 5));
            } else {
                xf_u32x4 vreg[4][2];
                if (tt > 0) {
                    xf_u32x4 kreg[8];
                    load_k(Xk, myidx, kreg);
                    load_v(Xv, myidx, vreg);
                    store_k(kt, kreg);
                } else {
                    load_v(Xv, myidx, vreg);
                }
                __builtin_amdgcn_wave_barrier();
                compute(tt, vreg, vreg);
            }
        }
    }
    // ---- denominators (row sums over the 16 key lanes) and the un-normalised z of the query into the wave's own LDS area ([head][256] fp32 = 8 KB)
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
        for (int i = 0; i < 4; ++i) lsum[wave * HEADS + 4 * g + i] = l_run[i];
    }
    {
        float* szw = reinterpret_cast<float*>(smem + wave * WAVE_LDS);
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
    XF_STAMP(15);
    __syncthreads();
    XF_STAMP(16);
    // ---------------------------------------------------------------- phase C: context maps, wave = head
#ifdef MV2D_XF_NOMAPS
    if (tid == 0) ctx[(long long)r * C] = reinterpret_cast<const float*>(smem)[lane];
    for (int h = wave; h < 0; h += QB) {
#else
    for (int h = wave; h < HEADS; h += QB) {
#endif
        const int j = n & (QB - 1);
        const float* zp = reinterpret_cast<const float*>(smem + j * WAVE_LDS) + h * C + 8 * g;
        // (xattn_tile_kernel normalises when it merges its waves: num * rcp(den), the factor of the single wave being exp2(0) = 1)
        const float rl = __builtin_amdgcn_rcpf(lsum[j * HEADS + h]);
        const uint4* wh = WB_hi + (long long)h * 16 * 64 + lane;
        const uint4* wl = WB_lo + (long long)h * 16 * 64 + lane;
        f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        xf_u32x4 wb_h[16], wb_l[16];                          // (all fragments of the head first, like phase A; and what the epilogue reads)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            wb_h[t] = *reinterpret_cast<const xf_u32x4*>(wh + t * 64);
            wb_l[t] = *reinterpret_cast<const xf_u32x4*>(wl + t * 64);
        }
        int rr_[4], rp0_[4], rp1_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rr_[i] = rq[(4 * g + i) & (QB - 1)];
            rp0_[i] = rbeg[(4 * g + i) & (QB - 1)];
            rp1_[i] = rend[(4 * g + i) & (QB - 1)];
        }
        const float bv0 = bv[32 * h + n], bv1 = bv[32 * h + 16 + n];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float4 x0 = *reinterpret_cast<const float4*>(zp + 32 * s);
            float4 x1 = *reinterpret_cast<const float4*>(zp + 32 * s + 4);
            x0 = make_float4(x0.x * rl, x0.y * rl, x0.z * rl, x0.w * rl);
            x1 = make_float4(x1.x * rl, x1.y * rl, x1.z * rl, x1.w * rl);
            XfFrag ah, al;
            xf_split8(x0, x1, ah, al);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[nt] = mfma_q16_16x16x32(al.v, wb_h[s * 2 + nt], acc[nt]);
                acc[nt] = mfma_q16_16x16x32(ah.v, wb_l[s * 2 + nt], acc[nt]);
                acc[nt] = mfma_q16_16x16x32(ah.v, wb_h[s * 2 + nt], acc[nt]);
            }
        }
        if (4 * g < QB) {
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
    XF_STAMP(17);
}

}  // namespace

#ifdef MV2D_XF_TRACE
extern "C" int mv2d_xf_trace_read(long long* host, int n) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_xf_trace), n * sizeof(long long)) == hipSuccess ? 0 : -2; }
#endif

// C-ABI: include/mv2d_hip.h
extern "C" int mv2d_xattn_fused_fwd(const float* q, const void* WA_hi, const void* WA_lo, const void* WB_hi, const void* WB_lo, const float* bv,
                                    const void* Xk, const void* Xv, const void* Xk_lo, const void* Xv_lo, const int* row_ptr, const int* col_idx,
                                    float* ctx, int R, int empty_nan, const int* order, int lo_fmt, void* stream) {
    MV2D_CHECK_ARG(lo_fmt == 0 || (lo_fmt == 1 && Xk_lo), "mv2d_xattn_fused_fwd: lo_fmt is 0 (key16 lo rows) or 1 (e4m3 lo rows; needs the lo rows)");
    MV2D_CHECK_ARG(q && WA_hi && WA_lo && WB_hi && WB_lo && bv && Xk && Xv && row_ptr && col_idx && ctx && R >= 0, "mv2d_xattn_fused_fwd: bad args");
    MV2D_CHECK_ARG((Xk_lo == nullptr) == (Xv_lo == nullptr), "mv2d_xattn_fused_fwd: Xk_lo and Xv_lo come together");
    MV2D_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)WA_hi & 15) == 0 && ((uintptr_t)WA_lo & 15) == 0 && ((uintptr_t)WB_hi & 15) == 0 &&
                       ((uintptr_t)WB_lo & 15) == 0 && ((uintptr_t)Xk & 15) == 0 && ((uintptr_t)Xv & 15) == 0 && ((uintptr_t)Xk_lo & 15) == 0 &&
                       ((uintptr_t)Xv_lo & 15) == 0, "mv2d_xattn_fused_fwd: operands must be 16-byte aligned");
    if (R == 0) return MV2D_OK;
    static int n_cu = 0;
    if (n_cu == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    // queries per block: 8 (one block per CU) or 4 (two per CU: the map phases of one block under the tile loop of the other).  MV2D_XF_QB=4 / 8 forces it.
    static const int qb_env = [] { const char* e = getenv("MV2D_XF_QB"); return e ? atoi(e) : 0; }();
    const int QB = qb_env == 4 || qb_env == 8 ? qb_env : MV2D_XF_QB_DEFAULT;
    const int slots = n_cu * (QB == 4 ? 2 : 1);               // blocks of a round
    int nblk = (R + QB - 1) / QB;
    if (nblk > slots) {                                       // whole rounds, as long as a block keeps >= QB / 2 queries
        const int up = (nblk + slots - 1) / slots * slots;
        if ((long long)up * (QB / 2) <= R) nblk = up;
    } else {
        // a small launch (one sample: 300 queries = 38 blocks of 8 on a 256-CU chip): spread it, down to two queries per block -- the blocks stream
        // the map weights from L2 either way, and the launch is bound by the longest block (26.5 -> ~14 us per layer at R = 300)
        const int spread = R / 2 < n_cu ? R / 2 : n_cu;
        if (spread > nblk) nblk = spread;
    }
    const dim3 grid(nblk), block(64 * QB);
#define MV2D_XF(XLO_, QB_) hipLaunchKernelGGL((xattn_fused_kernel<XLO_, QB_>), grid, block, 0, (hipStream_t)stream, q, (const uint4*)WA_hi, (const uint4*)WA_lo, \
                                              (const uint4*)WB_hi, (const uint4*)WB_lo, bv, (const unsigned short*)Xk, (const unsigned short*)Xv,                     \
                                              (const unsigned short*)Xk_lo, (const unsigned short*)Xv_lo, row_ptr, col_idx, ctx, R, empty_nan, order, nblk)
    if (Xk_lo && lo_fmt == 1) { if (QB == 4) MV2D_XF(2, 4); else MV2D_XF(2, 8); }
    else if (Xk_lo) { if (QB == 4) MV2D_XF(1, 4); else MV2D_XF(1, 8); }
    else { if (QB == 4) MV2D_XF(0, 4); else MV2D_XF(0, 8); }
#undef MV2D_XF
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
