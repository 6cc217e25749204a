// The PE block of the key side in SPLIT PRECISION, second shape (round 6; index-exact route; gfx950 / CDNA4, wave64), one launch:
//   P1 = position_encoder(A1)                                   192 -> 1024 -> 256     (MU/pe.py:64-77, 158-160)
//   G  = sigmoid(conv_expand(relu(conv_reduce(feat))))          256 -> 256 -> 256      (MU/pe.py:36-48, 162-166)
//   pe = tab[position] + P1 * G;  T path: key rows Xk = pe + feat, value rows Xv = feat as key16 hi + lo pairs
// The arithmetic of pe_x3_kernel (csrc/pe_x3.hip), product for product and in the same k order -- the outputs are BITWISE the same
// (tests/test_gpu_kernels.py::test_pe_fused_x3_kernel) -- on a different division of the work.  pe_x3_kernel keeps hi / lo images of a 64-row input tile
// and of a 256-column part of the hidden layer in LDS (135 KB: one block of 4 waves per CU, one wave per SIMD) and streams the weights from L2 straight
// into registers; its matrix pipe issues 42 % of the time (profiles/r05_px_trace_pe_x3.txt).  Here:
//   * a WAVE OWNS ITS ROWS through the whole chain.  With the swapped product D[column][row] a lane ends a layer holding, for its row, the columns
//     4 fg .. 4 fg + 3 of every 16-column tile; when the first layers' weight rows are packed so that tiles 2 s, 2 s + 1 hold the hidden columns
//     32 s + 8 fg + {0..3}, {4..7} (ops.pack_x3_rowperm), the bias + ReLU + hi / lo split of two accumulator tiles IS the B fragment of k-step s
//     of the next layer: the hidden layer never leaves the registers, no LDS image, no barrier between the layers;
//   * the WEIGHTS go through LDS instead -- one k-step of one layer (16 column tiles, hi + lo = 32 KB) per stage of a 4-deep ring filled by the
//     LDS-DMA (global_load_lds_dwordx4 as inline asm, counted vmcnt + one barrier per k-step), read by every wave of the 128-row block: half the
//     L2 -> CU weight stream per row of pe_x3_kernel.
// OPT-IN (HeadEngine.pe_rows_in_waves), measured on 250 k rows (tools/pe_time.py; pe_x3_kernel: 1132-1244 us):
//   8 waves x 16 rows, two waves per SIMD: 1056 us.  Without the DMA 881 us, without the DMA and the weight reads 466 us (76 % of the MFMA peak): every
//     wave reads every weight fragment for ONE row tile, 8 x 32 KB of ds_read_b128 per k-step -- the LDS read path is the bound (~85 B/clk per CU).
//   4 waves x 32 rows, one wave per SIMD (a fragment feeds two row tiles; the shape kept here): 1213 us.  Without the DMA 1062 us, without DMA and weight
//     reads 684 us: with one wave per SIMD the fragment reads, the hi / lo conversions of the hidden layer and the MFMAs of a step share one instruction
//     stream, and what the halved LDS traffic saves is lost again.
// Neither beats the kernel it was to replace; the file stays as the tested record of the experiment (LOG.md, round 6).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int C = 256;
constexpr int NS = 4, STG = 32768, NSTEP = 80;
#ifndef MV2D_PB_DBG
#define MV2D_PB_DBG 0            // timing experiments only: 1 = no MFMAs / weight reads, 2 = no DMA, 5 = no DMA and no weight reads
#endif
enum { B_R = 0, B_E = 256, B_1A = 512, B_1B = 1536, B_FLOATS = 1792 };
constexpr int OFF_B = NS * STG, SMEM = OFF_B + B_FLOATS * 4;

struct XFrag { uint4 h, l; };

struct PeX3bParams {
    const float* A1; const float* Xmap; const int* row_index; const int* m_dev; int M;
    const unsigned short* W1a_h; const unsigned short* W1a_l; const float* b1a; const unsigned short* W1b_h; const unsigned short* W1b_l; const float* b1b;
    const unsigned short* Wr_h; const unsigned short* Wr_l; const float* br; const unsigned short* We_h; const unsigned short* We_l; const float* be;
    const float* sine_tab; int tab_period;
    float* pe; unsigned short* Xk_hi; unsigned short* Xk_lo; unsigned short* Xv_hi; unsigned short* Xv_lo;
    int lo8;                                                 // the lo row outputs are 256-byte e4m3 rows (common.h "lo8")
    int* lo8_flag;                                           // |= 1 when a lo remainder leaves the e4m3 range (may be NULL)
    int pe_at_index;                                         // pe row m is written at row row_index[m] (a position-indexed map) instead of row m
};

// ---- the 80 steps of a block: parts 0..3 = hidden columns 256 p .. of the frustum MLP (6 + 8 k-steps each), part 4 = the gate: 8 k-steps of its
// first layer, then its second layer TWICE over its 8 k-steps, once per half of the output columns (steps 64..71: column tiles 0..7, 72..79: tiles 8..15;
// half stages of 16 KB) -- a full-width gate accumulator beside the P1 accumulator and the hidden tile would not fit the register file
__host__ __device__ constexpr int part_of(int T) { return T < 56 ? T / 14 : 4; }
__host__ __device__ constexpr int tin_of(int T) { return T < 64 ? T - part_of(T) * 14 : 8 + (T - 64) % 8; }
__host__ __device__ constexpr bool first_layer(int T) { return part_of(T) < 4 ? tin_of(T) < 6 : tin_of(T) < 8; }
__host__ __device__ constexpr int tile0_of(int T) { return T >= 72 ? 8 : 0; }
__host__ __device__ constexpr int ntile_of(int T) { return T >= 64 ? 8 : 16; }
// element offset of the stage's first tile inside its packed array ([k-step][tile][64 lanes][8])
__host__ __device__ constexpr long long stage_off(int T) {
    const int p = part_of(T), t = tin_of(T);
    return p < 4 ? (t < 6 ? (long long)(t * 64 + p * 16) * 512 : (long long)((p * 8 + (t - 6)) * 16) * 512)
                 : (t < 8 ? (long long)(t * 16) * 512 : (long long)((t - 8) * 16 + tile0_of(T)) * 512);
}

__device__ __forceinline__ void pb_dma16(const void* gbase, unsigned int voff, unsigned int lds_dst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ XFrag hidden_frag(const f32x4_t& a, const f32x4_t& b, const float4& ba, const float4& bb) {
    XFrag f;
    unsigned int h[4], l[4];
    split_q16x2(relu_f(a[0] + ba.x), relu_f(a[1] + ba.y), h[0], l[0]);
    split_q16x2(relu_f(a[2] + ba.z), relu_f(a[3] + ba.w), h[1], l[1]);
    split_q16x2(relu_f(b[0] + bb.x), relu_f(b[1] + bb.y), h[2], l[2]);
    split_q16x2(relu_f(b[2] + bb.z), relu_f(b[3] + bb.w), h[3], l[3]);
    f.h = make_uint4(h[0], h[1], h[2], h[3]);
    f.l = make_uint4(l[0], l[1], l[2], l[3]);
    return f;
}
__device__ __forceinline__ XFrag input_frag(const float4& a, const float4& b) {
    XFrag f;
    unsigned int h[4], l[4];
    split_q16x2(a.x, a.y, h[0], l[0]);
    split_q16x2(a.z, a.w, h[1], l[1]);
    split_q16x2(b.x, b.y, h[2], l[2]);
    split_q16x2(b.z, b.w, h[3], l[3]);
    f.h = make_uint4(h[0], h[1], h[2], h[3]);
    f.l = make_uint4(l[0], l[1], l[2], l[3]);
    return f;
}

// The kernel as a class template over its shape: NW waves per block, each owning RT tiles of 16 rows (block = 16 RT NW rows).
//   <8, 1>: two waves per SIMD (<= 256 registers); every wave reads every weight fragment for ONE row tile -- 8 x 32 KB of LDS reads per k-step,
//           measured LDS-bound (tools/pe_time.py: 881 us of 250 k rows without the DMA, 466 us without the weight reads)
//   <4, 2>: one wave per SIMD (<= 512 registers), a weight fragment feeds two row tiles: half the LDS reads per MFMA, the next group's fragments
//           are read while the current group's 24 MFMAs issue
template <int NW, int RT>
struct Pe {
    static constexpr int NTHR = 64 * NW, ROWS_W = 16 * RT, BM = ROWS_W * NW;
    static constexpr int PCS = 16 / NW;                      // 1 KB pieces of a 16 KB half stage per wave
    // ---- what is in flight: step u issues the DMA pieces of stage u + 3.  Loads return in order, so "stage T has landed" = at most the pieces issued
    // after its own are outstanding.  (hipcc's own loads -- the rows of the two first layers -- are requested outside the step loops; its waits for them
    // count fewer operations than are in flight and therefore drain the ring: once per block, in front of the gate.)
    static constexpr int pieces(int T) { return PCS * ntile_of(T) / 16; }      // 1 KB pieces per wave and half (hi | lo) of stage T
    static constexpr int n_dma(int u) { return (u + 3 >= 0 && u + 3 < NSTEP) ? 2 * pieces(u + 3) : 0; }
    static constexpr int wait_count(int T) { return n_dma(T - 2) + n_dma(T - 1); }

    struct Ctx {
        const PeX3bParams* p;
        unsigned char* smem;
        unsigned int lds0;         // LDS byte address of smem
        unsigned int voff;         // this lane's byte offset inside a 16 KB half stage: the wave's first piece
        int wave, lane;
    };

    template <int T>
    static __device__ __forceinline__ void dma_stage(const Ctx& c) {
        if constexpr (T < NSTEP && (MV2D_PB_DBG < 2)) {
            constexpr int p = part_of(T);
            constexpr bool l1 = first_layer(T);
            const unsigned short* bh = p < 4 ? (l1 ? c.p->W1a_h : c.p->W1b_h) : (l1 ? c.p->Wr_h : c.p->We_h);
            const unsigned short* bl = p < 4 ? (l1 ? c.p->W1a_l : c.p->W1b_l) : (l1 ? c.p->Wr_l : c.p->We_l);
            bh += stage_off(T);
            bl += stage_off(T);
            // tile j of the stage lands at j KB of the slot's hi / lo half (a half stage keeps the places of its tiles); the wave moves `pieces` tiles of each
            constexpr int NP = pieces(T);
            const unsigned int src = (unsigned int)(c.wave * (NP * 1024) + c.lane * 16);
            const unsigned int dst = __builtin_amdgcn_readfirstlane(c.lds0 + (unsigned int)((T % NS) * STG + (tile0_of(T) + c.wave * NP) * 1024));
#pragma unroll
            for (int i = 0; i < NP; ++i) pb_dma16(bh, src + 1024u * i, dst + 1024u * i);
#pragma unroll
            for (int i = 0; i < NP; ++i) pb_dma16(bl, src + 1024u * i, dst + 16384u + 1024u * i);
        }
    }
    template <int T>
    static __device__ __forceinline__ void kstep_wait(const Ctx& c) {
        // this wave's pieces of stage T have landed (stages T + 1, T + 2 may stay in flight) ...
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(wait_count(T)) : "memory");
        __builtin_amdgcn_s_barrier();                        // ... and everybody's; every wave is past k-step T - 1, whose slot stage T + 3 takes
    }
    // step T: the stage's column tiles x RT row tiles x (w_lo x_hi + w_hi x_lo + w_hi x_hi), the weight fragments from ring slot T % NS in groups of
    // 4 tiles; the fragments of group g + 1 are requested before the MFMAs of group g issue
    template <int T>
    static __device__ __forceinline__ void kstep_mma(const Ctx& c, f32x4_t (&acc)[RT][16], const XFrag (&x)[RT]) {
#if MV2D_PB_DBG == 1
        return;
#endif
        constexpr int J0 = tile0_of(T), NG = ntile_of(T) / 4;
        const unsigned char* st = c.smem + (T % NS) * STG + J0 * 1024 + c.lane * 16;
        uint4 wh[2][4], wl[2][4];
        auto load = [&](int g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#if MV2D_PB_DBG == 5
                wh[g & 1][j] = make_uint4(0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + T);
                wl[g & 1][j] = make_uint4(0x1c001c00u + j, 0x1c001c00u, 0x1c001c00u, 0x1c001c00u + T);
#else
                wh[g & 1][j] = *reinterpret_cast<const uint4*>(st + (4 * g + j) * 1024);
                wl[g & 1][j] = *reinterpret_cast<const uint4*>(st + 16384 + (4 * g + j) * 1024);
#endif
            }
        };
        load(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load(g + 1);
            // product-major: 4 RT independent MFMAs between two that accumulate into the same tile
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][J0 + 4 * g + j] = mfma_q16_16x16x32(t == 0 ? wl[g & 1][j] : wh[g & 1][j], t == 1 ? x[i].l : x[i].h, acc[i][J0 + 4 * g + j]);
        }
    }
    static __device__ __forceinline__ void zero(f32x4_t (&acc)[RT][16]) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // a first layer (N k-steps from T0) on fragments that are in registers
    template <int T0, int N, int S = 0>
    static __device__ __forceinline__ void input_steps(const Ctx& c, f32x4_t (&acc)[RT][16], const XFrag (&x)[N][RT]) {
        if constexpr (S < N) {
            kstep_wait<T0 + S>(c);
            dma_stage<T0 + S + 3>(c);
            kstep_mma<T0 + S>(c, acc, x[S]);
            input_steps<T0, N, S + 1>(c, acc, x);
        }
    }
    // a second layer: the hidden fragments are made on the way (two accumulator tiles per k-step and row tile)
    template <int T0, int S = 0>
    static __device__ __forceinline__ void hidden_steps(const Ctx& c, f32x4_t (&acc)[RT][16], const f32x4_t (&hid)[RT][16], const float* bias /* LDS, + 8 fg */) {
        if constexpr (S < 8) {
            const float4 ba = *reinterpret_cast<const float4*>(bias + 32 * S);
            const float4 bb = *reinterpret_cast<const float4*>(bias + 32 * S + 4);
            XFrag hb[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) hb[i] = hidden_frag(hid[i][2 * S], hid[i][2 * S + 1], ba, bb);
            kstep_wait<T0 + S>(c);
            dma_stage<T0 + S + 3>(c);
            kstep_mma<T0 + S>(c, acc, hb);
            hidden_steps<T0, S + 1>(c, acc, hid, bias);
        }
    }

    static __device__ __forceinline__ void run(const PeX3bParams& p, unsigned char* smem) {
        float* Bs = reinterpret_cast<float*>(smem + OFF_B);
        int M = p.M;
        if (p.m_dev) { const int md = *p.m_dev; M = md < M ? md : M; }
        const int m0 = blockIdx.x * BM;
        if (m0 >= M) return;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
        Ctx c{&p, smem, (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem, (unsigned int)(wave * (PCS * 1024) + lane * 16), wave, lane};
        dma_stage<0>(c);
        dma_stage<1>(c);
        dma_stage<2>(c);
        // biases -> LDS: [br | be | b1a | b1b] as 448 float4
        for (int t = tid; t < B_FLOATS / 4; t += NTHR) {
            const float* src = t < 64 ? p.br + 4 * t : t < 128 ? p.be + 4 * (t - 64) : t < 384 ? p.b1a + 4 * (t - 128) : p.b1b + 4 * (t - 384);
            *reinterpret_cast<float4*>(Bs + 4 * t) = *reinterpret_cast<const float4*>(src);
        }
        // this lane's rows (the B operand columns of the swapped products) and their frustum fragments: 6 k-steps x 8 values each
        XFrag xin[6][RT];
        const float* xp[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int mrow = min(m0 + ROWS_W * wave + 16 * i + fr, M - 1);
            const long long xrow = p.row_index ? p.row_index[mrow] : mrow;
            xp[i] = p.Xmap + xrow * C + 8 * fg;
            const float* a1 = p.A1 + (long long)mrow * 192 + 8 * fg;
#pragma unroll
            for (int s_ = 0; s_ < 6; ++s_)
                xin[s_][i] = input_frag(*reinterpret_cast<const float4*>(a1 + 32 * s_), *reinterpret_cast<const float4*>(a1 + 32 * s_ + 4));
        }
        __syncthreads();                                     // the biases are in LDS
        // ---- 1. P1 = position_encoder(A1) in four parts of 256 hidden columns
        f32x4_t accf[RT][16];
        zero(accf);
        {
            f32x4_t hid[RT][16];
            zero(hid);
            input_steps<0, 6>(c, hid, xin);
            hidden_steps<6>(c, accf, hid, Bs + B_1A + 8 * fg);
            zero(hid);
            input_steps<14, 6>(c, hid, xin);
            hidden_steps<20>(c, accf, hid, Bs + B_1A + 256 + 8 * fg);
            zero(hid);
            input_steps<28, 6>(c, hid, xin);
            hidden_steps<34>(c, accf, hid, Bs + B_1A + 512 + 8 * fg);
            zero(hid);
            input_steps<42, 6>(c, hid, xin);
            hidden_steps<48>(c, accf, hid, Bs + B_1A + 768 + 8 * fg);
        }
        // ---- 2. the gate on the feature rows; 3. (P1 + b) * sigmoid(gate + b) per half of the columns: accumulator tile j, element e = column
        // 16 j + 4 fg + e of row fr.  (accurate exp: this route is compared at fp32 rounding level)
        {
            XFrag xf[8][RT];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_)
                    xf[s_][i] = input_frag(*reinterpret_cast<const float4*>(xp[i] + 32 * s_), *reinterpret_cast<const float4*>(xp[i] + 32 * s_ + 4));
            f32x4_t hid[RT][16];
            zero(hid);
            input_steps<56, 8>(c, hid, xf);
            f32x4_t accg[RT][16];                            // (only the 8 tiles of the current half are live)
            auto apply = [&](int j0) {
#pragma unroll
                for (int j = j0; j < j0 + 8; ++j) {
                    const float4 eb = *reinterpret_cast<const float4*>(Bs + B_E + 16 * j + 4 * fg);
                    const float4 fb = *reinterpret_cast<const float4*>(Bs + B_1B + 16 * j + 4 * fg);
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const f32x4_t gt{1.f / (1.f + expf(-(accg[i][j][0] + eb.x))), 1.f / (1.f + expf(-(accg[i][j][1] + eb.y))),
                                         1.f / (1.f + expf(-(accg[i][j][2] + eb.z))), 1.f / (1.f + expf(-(accg[i][j][3] + eb.w)))};
                        accf[i][j] = f32x4_t{(accf[i][j][0] + fb.x) * gt[0], (accf[i][j][1] + fb.y) * gt[1], (accf[i][j][2] + fb.z) * gt[2],
                                             (accf[i][j][3] + fb.w) * gt[3]};
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) accg[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            hidden_steps<64>(c, accg, hid, Bs + B_R + 8 * fg);
            apply(0);
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 8; j < 16; ++j) accg[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            hidden_steps<72>(c, accg, hid, Bs + B_R + 8 * fg);
            apply(8);
        }
        __syncthreads();                                     // every wave is done with the ring: it becomes the waves' output tiles
        // ---- 4. through a wave-private [16 RT rows][256] fp32 tile (16-byte chunk c of row r at slot c ^ (r & 15)), then whole rows: + table, T path:
        // + feature row -> key16 pairs
        float* ot = reinterpret_cast<float*>(smem + wave * (ROWS_W * 1024));
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                *reinterpret_cast<float4*>(ot + (16 * i + fr) * C + (((4 * j + fg) ^ fr) << 2)) = make_float4(accf[i][j][0], accf[i][j][1], accf[i][j][2], accf[i][j][3]);
        __builtin_amdgcn_wave_barrier();
        const bool rows16 = p.Xk_hi != nullptr;
#pragma unroll 4
        for (int r = 0; r < ROWS_W; ++r) {
            const int m = m0 + ROWS_W * wave + r;
            if (m >= M) break;                               // (wave-uniform)
            const long long ri = p.row_index ? p.row_index[m] : m;
            float4 v = *reinterpret_cast<const float4*>(ot + r * C + ((lane ^ (r & 15)) << 2));
            const float4 tv = *reinterpret_cast<const float4*>(p.sine_tab + (long long)((int)ri % p.tab_period) * C + 4 * lane);
            v = make_float4(v.x + tv.x, v.y + tv.y, v.z + tv.z, v.w + tv.w);
            if (p.pe) *reinterpret_cast<float4*>(p.pe + (p.pe_at_index ? ri : (long long)m) * C + 4 * lane) = v;
            if (rows16) {
                const float4 f = *reinterpret_cast<const float4*>(p.Xmap + ri * C + 4 * lane);
                uint2 h, l;
                split_k16x2(v.x + f.x, v.y + f.y, h.x, l.x);
                split_k16x2(v.z + f.z, v.w + f.w, h.y, l.y);
                *reinterpret_cast<uint2*>(p.Xk_hi + (long long)m * C + 4 * lane) = h;
                if (p.lo8) *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(p.Xk_lo) + (long long)m * C + 4 * lane) = lo8_pack4_flag(l.x, l.y, p.lo8_flag);
                else *reinterpret_cast<uint2*>(p.Xk_lo + (long long)m * C + 4 * lane) = l;
                split_k16x2(f.x, f.y, h.x, l.x);
                split_k16x2(f.z, f.w, h.y, l.y);
                *reinterpret_cast<uint2*>(p.Xv_hi + (long long)m * C + 4 * lane) = h;
                if (p.lo8) *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(p.Xv_lo) + (long long)m * C + 4 * lane) = lo8_pack4_flag(l.x, l.y, p.lo8_flag);
                else *reinterpret_cast<uint2*>(p.Xv_lo + (long long)m * C + 4 * lane) = l;
            }
        }
    }
};

__global__ __launch_bounds__(256, 1) void pe_x3b_kernel_4x2(PeX3bParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    Pe<4, 2>::run(p, smem);
}

}  // namespace

// C-ABI: include/mv2d_hip.h
extern "C" int mv2d_pe_fused_x3b(const float* A1, const float* Xmap, const int* row_index, const int* m_dev, int M,
                                 const void* W1a_hi, const void* W1a_lo, const float* b1a, const void* W1b_hi, const void* W1b_lo, const float* b1b,
                                 const void* Wr_hi, const void* Wr_lo, const float* br, const void* We_hi, const void* We_lo, const float* be,
                                 const float* sine_tab, int tab_period, float* pe, void* Xk_hi, void* Xk_lo, void* Xv_hi, void* Xv_lo, int lo_fmt, int pe_at_index, int* lo8_flag, void* stream) {
    MV2D_CHECK_ARG(A1 && Xmap && W1a_hi && W1a_lo && b1a && W1b_hi && W1b_lo && b1b && Wr_hi && Wr_lo && br && We_hi && We_lo && be && sine_tab,
                   "mv2d_pe_fused_x3b: null pointer");
    MV2D_CHECK_ARG(pe || Xk_hi, "mv2d_pe_fused_x3b: no output");
    MV2D_CHECK_ARG(!pe_at_index || (pe && row_index), "mv2d_pe_fused_x3b: pe_at_index needs pe and row_index");
    MV2D_CHECK_ARG((Xk_hi != nullptr) == (Xk_lo != nullptr) && (Xk_hi != nullptr) == (Xv_hi != nullptr) && (Xk_hi != nullptr) == (Xv_lo != nullptr),
                   "mv2d_pe_fused_x3b: the four key / value row outputs come together");
    MV2D_CHECK_ARG(M >= 0 && tab_period > 0, "mv2d_pe_fused_x3b: M must be >= 0 and tab_period > 0");
    MV2D_CHECK_ARG(lo_fmt == 0 || lo_fmt == 1, "mv2d_pe_fused_x3b: lo_fmt is 0 (key16 lo rows) or 1 (e4m3 lo rows)");
    MV2D_CHECK_ARG(((uintptr_t)A1 & 15) == 0 && ((uintptr_t)Xmap & 15) == 0 && ((uintptr_t)sine_tab & 15) == 0, "mv2d_pe_fused_x3b: rows must be 16-byte aligned");
    MV2D_CHECK_ARG((((uintptr_t)W1a_hi | (uintptr_t)W1a_lo | (uintptr_t)W1b_hi | (uintptr_t)W1b_lo | (uintptr_t)Wr_hi | (uintptr_t)Wr_lo | (uintptr_t)We_hi |
                     (uintptr_t)We_lo) & 15) == 0, "mv2d_pe_fused_x3b: packed weights must be 16-byte aligned");
    if (M == 0) return MV2D_OK;
    PeX3bParams p{A1, Xmap, row_index, m_dev, M, (const unsigned short*)W1a_hi, (const unsigned short*)W1a_lo, b1a, (const unsigned short*)W1b_hi,
                  (const unsigned short*)W1b_lo, b1b, (const unsigned short*)Wr_hi, (const unsigned short*)Wr_lo, br, (const unsigned short*)We_hi,
                  (const unsigned short*)We_lo, be, sine_tab, tab_period, pe, (unsigned short*)Xk_hi, (unsigned short*)Xk_lo, (unsigned short*)Xv_hi,
                  (unsigned short*)Xv_lo, lo_fmt, lo8_flag, pe_at_index};
    hipLaunchKernelGGL(pe_x3b_kernel_4x2, dim3(cdiv(M, 128)), dim3(256), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
