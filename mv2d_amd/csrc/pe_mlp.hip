// The whole position-encoding block of MV2D's PE module for one tile of 64 key positions in ONE launch:
//   P1 = position_encoder(A1):  Linear(192->1024)-ReLU-Linear(1024->256)      (MU/pe.py:64-77, 158-160: frustum coordinates)
//   G  = sigmoid(conv_expand(relu(conv_reduce(feat))))                        (SELayer gate, MU/pe.py:36-48, 162-166)
//   P2 = adapt_pos3d(A2):       Linear(384->1024)-ReLU-Linear(1024->256)      (MU/pe.py:150-156: sine embedding)
//   pe = P2 + P1 * G ,   Xk = bf16(pe + feat)                                 (MU/pe.py:166; key input of the cross attention)
// (all "1x1 convs" of the reference are per-position linears).
//
// Before: six bf16 GEMM launches with the two [S,1024] hidden activations going through memory, 64x64 tiles that are L2-port and
// LDS-read bound (100 us per frame at S = 8.8k, 8 % of the CU-time it occupies is MFMA).  Here, in the pattern of roiconv.hip:
//   * the activation operand of each layer is RESIDENT in LDS (input tile 64 x K1, hidden 64 x 512 bf16), written once, read by
//     all waves; two barriers per half of the hidden layer, none inside the k loops; every wave owns 1/NW of the columns
//     of a layer for all 64 rows, so each weight fragment is read by exactly one wave (measured: 4 waves x 64 rows = 8 waves x 64
//     rows = 4 waves x 32 rows with two blocks per CU, within 5 %);
//   * the weights are never staged: fragment-major copies (mv2d_pack_wfrag_bf16) stream straight from L2 into a 4-deep register
//     ring, one contiguous 1 KB per fragment load, prefetched 3 steps ahead through all layer / part / MLP boundaries;
//   * hidden activations never leave the CU; P1, G, P2 are combined in registers (LDS allows one block per CU anyway, so a wave
//     may use all 512 registers); one fp32 + one bf16 store per output.
// MFMAs run swapped (D^T = W.A^T): a lane ends with 4 consecutive columns of one row -> 8-byte LDS writes of the hidden layer,
// 16-byte global stores.  Same k order and the same bf16 rounding of the hidden layers as the GEMM route: results are bit-identical.
#include <cstdlib>
#include "common.h"

namespace {

#ifdef MV2D_PE_TRACE
__device__ long long g_pe_trace[64];
#define PE_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_pe_trace[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PE_STAMP(i) do {} while (0)
#endif

#ifndef MV2D_PE_BM
#define MV2D_PE_BM 64
#endif
#ifndef MV2D_PE_NW
#define MV2D_PE_NW 4
#endif
constexpr int C = 256, BM = MV2D_PE_BM, RT = BM / 16;     // rows / row tiles per block
constexpr int NW = MV2D_PE_NW;                            // waves per block; every wave owns 1/NW of the columns of a layer
constexpr int CT2 = 16 / NW;                              // output column tiles per wave (layer 2)
constexpr int HP = 512;                                   // hidden columns resident at a time
constexpr int PITCH_H = HP * 2;                           // 1 KB rows
constexpr int A_BYTES = BM * 1024, H_BYTES = BM * PITCH_H;
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
union Frag { uint4 u; mfma_bf16x8 v; };

struct PeParams {
    const unsigned short* A1; const unsigned short* A2; const unsigned short* Xfb; const float* Xf32; const int* row_index; const int* m_dev; int M;
    const unsigned short* W1a; const float* b1a; const unsigned short* W1b; const float* b1b;
    const unsigned short* W2a; const float* b2a; const unsigned short* W2b; const float* b2b;
    const unsigned short* Wr; const float* br; const unsigned short* We; const float* be;
    float* pe; unsigned short* Xk;
    const float* sine_tab; int tab_period;      // TAB variant: adapt_pos3d(sine) + b2b per map position (a constant of weights + padding geometry)
};

// ---- building blocks -------------------------------------------------------------------------------------------------------
// The kernel is ONE software pipeline over the 152 k-steps of the three MLPs (gate 16, frustum 2 x 28, sine 2 x 40): every step
// consumes four 1 KB weight fragments from a 4-slot register ring that is fed three steps ahead, also across the layer / part /
// MLP boundaries (the successor's first three steps are requested in the last three steps of a part), so the ring never drains.
// Everything else a step needs is requested earlier as well: the LDS fragments of step t+1 during step t, the hidden-layer bias
// at the first k-step of a pass, the next MLP's input tile / the epilogue operands at the start of the last layer 2.  (Round-1
// profile of the previous version: 950 cycles per step against 256 cycles of MFMA work; the input staging loop and the
// epilogues were one exposed memory round trip per 16 bytes, the bias loads drained the ring once per pass.)
// All step indices are compile-time (one function instance per part, unrolled step loops): hipcc keeps exact vmcnt counts.
template <int K1_, int HID_>
struct Mlp {
    static constexpr int K1 = K1_, HID = HID_;
    static constexpr int PITCH_A = K1 <= 256 ? 512 : 1024;       // power-of-two row pitch so that the chunk XOR stays inside the row
    static constexpr int CPR = K1 / 8;                           // 16-byte chunks per input row
    static constexpr int HPL = HID >= HP ? HP : HID;             // hidden columns per part (512, or 256 for the gate)
    static constexpr int NH = HID / HPL;
    static constexpr int TPW = HPL / 16 / NW;                    // hidden column tiles per wave and part ...
    static constexpr int PT1 = 4, NPASS = TPW / PT1;             // ... in passes of 4
    static constexpr int KS1 = K1 / 32, L1 = NPASS * KS1, L2 = HPL / 32, NSTEP = L1 + L2;
    static constexpr int NT1 = HID / 16;
    static constexpr int NST = BM * CPR / (64 * NW) / 2;         // staging loads per thread and half (two register arrays: hipcc
                                                                 // leaves a 12 x 16-byte array in scratch)
    static_assert(TPW % 4 == 0 && CT2 == 4, "four fragments per step");
    static_assert(NSTEP % 4 == 0 && L1 % 4 == 0, "ring slots stay aligned across parts");
    static_assert(BM * CPR % (2 * 64 * NW) == 0, "whole staging rounds");
};
using MlpG = Mlp<256, 256>;
using MlpA = Mlp<192, 1024>;
using MlpB = Mlp<384, 1024>;

struct WPtr { const unsigned short* w1; const unsigned short* w2; };     // per-wave, per-lane bases of the two fragment-major matrices

// fragment group (4 consecutive column tiles = 4 KB) of step t of part H
template <class ML, int H>
__device__ __forceinline__ const unsigned short* step_ptr(const WPtr& w, int t) {
    if (t < ML::L1) {
        const int pass = t / ML::KS1, ks = t - pass * ML::KS1;
        return w.w1 + (long long)(ks * ML::NT1 + H * (ML::HPL / 16) + pass * ML::PT1) * 512;
    }
    return w.w2 + (long long)((H * (ML::HPL / 32) + (t - ML::L1)) * 16) * 512;
}
__device__ __forceinline__ void load4(Frag (&dst)[4], const unsigned short* wp) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j].u = *reinterpret_cast<const uint4*>(wp + j * 512);
}
#define MMA(w, a, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((w).v, (a).v, c, 0, 0, 0)
__device__ __forceinline__ void load_a(Frag (&a)[RT], const unsigned char* S, int pitch, int kstep, int fr, int fg) {
#pragma unroll
    for (int i = 0; i < RT; ++i) a[i].u = *reinterpret_cast<const uint4*>(S + (16 * i + fr) * pitch + (((4 * kstep + fg) ^ fr) << 4));
}

// input tile rows m0.. of `Ag` (ld = K1): all loads of a thread are issued together (rows beyond M clamp to the last valid row; their
// results are never stored), the LDS writes follow when the tile buffer is free
template <class ML>
__device__ __forceinline__ void stage_issue(uint4 (&sa)[ML::NST], uint4 (&sb)[ML::NST], const unsigned short* __restrict__ Ag, int m0, int M, int tid) {
#pragma unroll
    for (int i = 0; i < 2 * ML::NST; ++i) {
        const int c = tid + i * (64 * NW), row = c / ML::CPR, chunk = c - row * ML::CPR;
        const uint4 v = *reinterpret_cast<const uint4*>(Ag + (long long)min(m0 + row, M - 1) * ML::K1 + chunk * 8);
        if (i < ML::NST) sa[i] = v; else sb[i - ML::NST] = v;
    }
}
template <class ML>
__device__ __forceinline__ void stage_write(unsigned char* As, const uint4 (&sa)[ML::NST], const uint4 (&sb)[ML::NST], int tid) {
#pragma unroll
    for (int i = 0; i < 2 * ML::NST; ++i) {
        const int c = tid + i * (64 * NW), row = c / ML::CPR, chunk = c - row * ML::CPR;
        *reinterpret_cast<uint4*>(As + row * ML::PITCH_A + ((chunk ^ (row & 15)) << 4)) = i < ML::NST ? sa[i] : sb[i - ML::NST];
    }
}

// what a part requests at the start of its layer 2 (the first point where the input tile buffer / the dead layer-1 accumulators
// are free), for use after its last step:
enum Mid { MID_NONE = 0, MID_STAGE = 1, MID_FINAL = 3 };
// the feature rows of the last epilogue: requested at the start of the last layer 2 (64 more live registers) or in the epilogue itself
constexpr bool F_EARLY = false;
// layer 1: LDS fragments of step t+1 requested during step t (16 more live registers where the pressure peaks) or at the start of step t
constexpr bool A_AHEAD_L1 = false;
struct Ctx { const unsigned short* next_in; const float* Xf32; int m0, M, tid, n0; int stamp; };
// all biases live in LDS (a global load in the middle of the pipeline would have to be waited for through the whole ring)
enum { B_R = 0, B_E = 256, B_1A = 512, B_1B = 1536, B_2A = 1792, B_2B = 2816, B_FLOATS = 3072 };

// One resident part H of MLP `ML` on the block's rows: acc2 += relu(A . W1[part]^T + b1[part]) . W2[:, part]^T.
// On entry the ring holds steps 0..2 of this part; its last three steps request steps 0..2 of part NXH of `NX` (HAS_NEXT).
template <class ML, int H, class NX, int NXH, bool HAS_NEXT, int MID>
__device__ __forceinline__ void mlp_part(const unsigned char* As, unsigned char* Hs, const WPtr& w, const float* b1 /* LDS */, const WPtr& nw,
                                         f32x4_t (&acc2)[RT][CT2], Frag (&wq)[4][4], int lane, int wave, const Ctx& cx,
                                         uint4 (&sa)[NX::NST], uint4 (&sb)[NX::NST], float4 (&f)[RT][CT2]) {
    const int fr = lane & 15, fg = lane >> 4;
    f32x4_t acc1[RT][4];
    Frag a[2][RT];
    load_a(a[0], As, ML::PITCH_A, 0, fr, fg);
#pragma unroll
    for (int t = 0; t < ML::L1; ++t) {
        if (t + 3 < ML::NSTEP) load4(wq[(t + 3) & 3], step_ptr<ML, H>(w, t + 3));
        const int pass = t / ML::KS1, ks = t - pass * ML::KS1;
        if (ks == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < RT; ++i) acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        if (A_AHEAD_L1 && t + 1 < ML::L1) load_a(a[(t + 1) & 1], As, ML::PITCH_A, (t + 1) % ML::KS1, fr, fg);
        if (!A_AHEAD_L1 && t > 0) load_a(a[t & 1], As, ML::PITCH_A, t % ML::KS1, fr, fg);
        __builtin_amdgcn_sched_barrier(0);               // keep the requests ahead of this step's MFMAs
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc1[i][j] = MMA(wq[t & 3][j], a[t & 1][i], acc1[i][j]);
        if (ks == ML::KS1 - 1) {
            // hidden tile of this pass: lane (fr, fg) holds hidden columns 4fg..4fg+3 of row 16i + fr -> bias, ReLU, bf16, 8-byte write
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lcol = (wave * ML::TPW + pass * 4 + j) * 16 + 4 * fg;             // column inside the resident part
                const float4 bb = *reinterpret_cast<const float4*>(b1 + H * ML::HPL + lcol);
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const uint2 hv = make_uint2(pack_bf16x2(relu_f(acc1[i][j][0] + bb.x), relu_f(acc1[i][j][1] + bb.y)),
                                                pack_bf16x2(relu_f(acc1[i][j][2] + bb.z), relu_f(acc1[i][j][3] + bb.w)));
                    *reinterpret_cast<uint2*>(Hs + (16 * i + fr) * PITCH_H + (((lcol >> 3) ^ fr) << 4) + (lcol & 4) * 2) = hv;
                }
            }
        }
    }
    PE_STAMP(cx.stamp);                                // end of layer 1 (before the barrier)
    __syncthreads();                                   // the resident part of the hidden layer is complete
    PE_STAMP(cx.stamp + 1);
    if (MID == MID_STAGE) stage_issue<NX>(sa, sb, cx.next_in, cx.m0, cx.M, cx.tid);
    if (MID == MID_FINAL && F_EARLY) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT2; ++j)
                f[i][j] = *reinterpret_cast<const float4*>(cx.Xf32 + (long long)min(cx.m0 + 16 * i + fr, cx.M - 1) * C + cx.n0 + 16 * j);
    }
    load_a(a[0], Hs, PITCH_H, 0, fr, fg);
#pragma unroll
    for (int t2 = 0; t2 < ML::L2; ++t2) {
        const int t = ML::L1 + t2;
        if (t + 3 < ML::NSTEP) load4(wq[(t + 3) & 3], step_ptr<ML, H>(w, t + 3));
        else if (HAS_NEXT) load4(wq[(t + 3) & 3], step_ptr<NX, NXH>(nw, t + 3 - ML::NSTEP));
        if (t2 + 1 < ML::L2) load_a(a[(t2 + 1) & 1], Hs, PITCH_H, t2 + 1, fr, fg);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT2; ++j)
                acc2[i][j] = MMA(wq[t & 3][j], a[t2 & 1][i], acc2[i][j]);
    }
    PE_STAMP(cx.stamp + 2);                            // end of layer 2 (before the barrier)
    __syncthreads();                                   // everybody is done reading this part (and, after the last one, As / Hs are free)
    PE_STAMP(cx.stamp + 3);
    const_cast<Ctx&>(cx).stamp += 4;
}

__device__ __forceinline__ void zero_acc(f32x4_t (&acc)[RT][CT2]) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// TAB: the sine branch (stage 3, 80 of the 152 k-steps) is read from p.sine_tab[position % tab_period] instead of being evaluated.
template <bool TAB>
__global__ __launch_bounds__(64 * NW, 1) void pe_fused_kernel(PeParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + H_BYTES + B_FLOATS * 4];
    unsigned char* As = smem;
    unsigned char* Hs = smem + A_BYTES;
    float* Bs = reinterpret_cast<float*>(smem + A_BYTES + H_BYTES);
    int M = p.M;
    if (p.m_dev) { const int md = *p.m_dev; M = md < M ? md : M; }
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int n0 = wave * (CT2 * 16) + 4 * fg;          // this lane's 4 output columns of column tile j start at n0 + 16 j

    const long long lo = (long long)lane * 8;
    const WPtr wG{p.Wr + (long long)(wave * MlpG::TPW) * 512 + lo, p.We + (long long)(wave * CT2) * 512 + lo};
    const WPtr wA{p.W1a + (long long)(wave * MlpA::TPW) * 512 + lo, p.W1b + (long long)(wave * CT2) * 512 + lo};
    const WPtr wB{p.W2a + (long long)(wave * MlpB::TPW) * 512 + lo, p.W2b + (long long)(wave * CT2) * 512 + lo};
    Frag wq[4][4];
#pragma unroll
    for (int t = 0; t < 3; ++t) load4(wq[t], step_ptr<MlpG, 0>(wG, t));
    {
        // biases -> LDS: float4 index tid + 256 r of [br | be | b1a | b1b | b2a | b2b]
        float4 bv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int q = tid + 256 * r;
            const float* src = q < 64 ? p.br + 4 * q : q < 128 ? p.be + 4 * (q - 64) : q < 384 ? p.b1a + 4 * (q - 128)
                             : q < 448 ? p.b1b + 4 * (q - 384) : q < 704 ? p.b2a + 4 * (q - 448) : p.b2b + 4 * (q - 704);
            bv[r] = *reinterpret_cast<const float4*>(src);
        }
        uint4 sa[MlpG::NST], sb[MlpG::NST];
        stage_issue<MlpG>(sa, sb, p.Xfb, m0, M, tid);
#pragma unroll
        for (int r = 0; r < 3; ++r) *reinterpret_cast<float4*>(Bs + 4 * (tid + 256 * r)) = bv[r];
        stage_write<MlpG>(As, sa, sb, tid);
    }
    __syncthreads();
    f32x4_t acc[RT][CT2];
    f32x4_t g[RT][CT2];                                 // the gate, then P1 * gate: carried in registers through the next MLP
    float4 f[RT][CT2];                                  // feature rows
    Ctx cx{p.A1, p.Xf32, m0, M, tid, n0, 1};
    PE_STAMP(0);

    // 1. gate = sigmoid(conv_expand(relu(conv_reduce(feat))))
    zero_acc(acc);
    {
        uint4 sa[MlpA::NST], sb[MlpA::NST];
        mlp_part<MlpG, 0, MlpA, 0, true, MID_STAGE>(As, Hs, wG, Bs + B_R, wA, acc, wq, lane, wave, cx, sa, sb, f);
        stage_write<MlpA>(As, sa, sb, tid);
    }
#pragma unroll
    for (int j = 0; j < CT2; ++j) {
        const float4 eb = *reinterpret_cast<const float4*>(Bs + B_E + n0 + 16 * j);
#pragma unroll
        for (int i = 0; i < RT; ++i)
            g[i][j] = f32x4_t{1.f / (1.f + __expf(-(acc[i][j][0] + eb.x))), 1.f / (1.f + __expf(-(acc[i][j][1] + eb.y))),
                              1.f / (1.f + __expf(-(acc[i][j][2] + eb.z))), 1.f / (1.f + __expf(-(acc[i][j][3] + eb.w)))};
    }
    __syncthreads();
    // 2. P1 = position_encoder(A1);  g = (P1 + b) * gate
    zero_acc(acc);
    {
        uint4 s0[MlpA::NST], s1[MlpA::NST];
        mlp_part<MlpA, 0, MlpA, 1, true, MID_NONE>(As, Hs, wA, Bs + B_1A, wA, acc, wq, lane, wave, cx, s0, s1, f);
    }
    cx.next_in = p.A2;
    if (TAB) {
        uint4 s0[MlpA::NST], s1[MlpA::NST];
        mlp_part<MlpA, 1, MlpA, 1, false, MID_FINAL>(As, Hs, wA, Bs + B_1A, wA, acc, wq, lane, wave, cx, s0, s1, f);
    } else {
        uint4 sa[MlpB::NST], sb[MlpB::NST];
        mlp_part<MlpA, 1, MlpB, 0, true, MID_STAGE>(As, Hs, wA, Bs + B_1A, wB, acc, wq, lane, wave, cx, sa, sb, f);
        stage_write<MlpB>(As, sa, sb, tid);
    }
#pragma unroll
    for (int j = 0; j < CT2; ++j) {
        const float4 eb = *reinterpret_cast<const float4*>(Bs + B_1B + n0 + 16 * j);
#pragma unroll
        for (int i = 0; i < RT; ++i)
            g[i][j] = f32x4_t{(acc[i][j][0] + eb.x) * g[i][j][0], (acc[i][j][1] + eb.y) * g[i][j][1],
                              (acc[i][j][2] + eb.z) * g[i][j][2], (acc[i][j][3] + eb.w) * g[i][j][3]};
    }
    __syncthreads();
    // 3. P2 = adapt_pos3d(A2);  pe = (P2 + b) + g;  Xk = bf16(pe + feat)
    zero_acc(acc);
    if (!TAB) {
        uint4 s0[MlpB::NST], s1[MlpB::NST];
        mlp_part<MlpB, 0, MlpB, 1, true, MID_NONE>(As, Hs, wB, Bs + B_2A, wB, acc, wq, lane, wave, cx, s0, s1, f);
        mlp_part<MlpB, 1, MlpB, 1, false, MID_FINAL>(As, Hs, wB, Bs + B_2A, wB, acc, wq, lane, wave, cx, s0, s1, f);
    }
    // ---- final epilogue.  The MFMA layout gives a lane 4 columns of 16 different rows: stored directly, every store instruction touches
    // 16 rows x 64 bytes and the epilogue took 20 % of the kernel (store-issue bound, 19k cycles).  The tile goes through LDS (free
    // after the last barrier) and comes back row-major: a wave stores 4 rows x 256 contiguous bytes per instruction, and the feature
    // rows of Xk = bf16(pe + feat) are read the same way.
    (void)f;
    float* ot = reinterpret_cast<float*>(smem) + wave * (BM * 68);          // [BM rows][64 columns of this wave], pitch 68 floats
#pragma unroll
    for (int j = 0; j < CT2; ++j) {
        const float4 eb = TAB ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(Bs + B_2B + n0 + 16 * j);
#pragma unroll
        for (int i = 0; i < RT; ++i)
            *reinterpret_cast<float4*>(ot + (16 * i + fr) * 68 + 16 * j + 4 * fg) =
                TAB ? make_float4(g[i][j][0], g[i][j][1], g[i][j][2], g[i][j][3])
                    : make_float4((acc[i][j][0] + eb.x) + g[i][j][0], (acc[i][j][1] + eb.y) + g[i][j][1],
                                  (acc[i][j][2] + eb.z) + g[i][j][2], (acc[i][j][3] + eb.w) + g[i][j][3]);
    }
    __builtin_amdgcn_wave_barrier();                         // the tile is read back by the same wave only
    {
        const int c4 = (lane & 15) * 4, r0 = lane >> 4;
        const long long gcol = wave * (CT2 * 16) + c4;
        float4 fv[BM / 4], tv[TAB ? BM / 4 : 1];
        if (TAB) {
            int ri[BM / 4];
#pragma unroll
            for (int k = 0; k < BM / 4; ++k) {               // map position of key row m: row_index[m], or m itself (the whole map)
                const int m = min(m0 + 4 * k + r0, M - 1);
                ri[k] = p.row_index ? p.row_index[m] : m;
            }
#pragma unroll
            for (int k = 0; k < BM / 4; ++k) {
                if (p.Xk) fv[k] = *reinterpret_cast<const float4*>(p.Xf32 + (long long)(p.row_index ? ri[k] : min(m0 + 4 * k + r0, M - 1)) * C + gcol);
                tv[TAB ? k : 0] = *reinterpret_cast<const float4*>(p.sine_tab + (long long)(ri[k] % p.tab_period) * C + gcol);
            }
        } else if (p.row_index) {                            // feature rows straight from the position-major map (row of key m = row_index[m])
            int ri[BM / 4];
#pragma unroll
            for (int k = 0; k < BM / 4; ++k) ri[k] = p.row_index[min(m0 + 4 * k + r0, M - 1)];
#pragma unroll
            for (int k = 0; k < BM / 4; ++k) fv[k] = *reinterpret_cast<const float4*>(p.Xf32 + (long long)ri[k] * C + gcol);
        } else {
#pragma unroll
            for (int k = 0; k < BM / 4; ++k)
                fv[k] = *reinterpret_cast<const float4*>(p.Xf32 + (long long)min(m0 + 4 * k + r0, M - 1) * C + gcol);
        }
#pragma unroll
        for (int k = 0; k < BM / 4; ++k) {
            const int row = 4 * k + r0, m = m0 + row;
            float4 v = *reinterpret_cast<const float4*>(ot + row * 68 + c4);
            if (TAB) v = make_float4(v.x + tv[TAB ? k : 0].x, v.y + tv[TAB ? k : 0].y, v.z + tv[TAB ? k : 0].z, v.w + tv[TAB ? k : 0].w);
            if (m < M) {
                if (!TAB || p.pe) *reinterpret_cast<float4*>(p.pe + (long long)m * C + gcol) = v;
                if (!TAB || p.Xk)
                    *reinterpret_cast<uint2*>(p.Xk + (long long)m * C + gcol) =
                        make_uint2(pack_bf16x2(v.x + fv[k].x, v.y + fv[k].y), pack_bf16x2(v.z + fv[k].z, v.w + fv[k].w));
            }
        }
    }
    PE_STAMP(cx.stamp);
}

}  // namespace

// C-ABI: see include/mv2d_hip.h
extern "C" int mv2d_pe_fused(const void* A1, const void* A2, const void* Xfb, const float* Xf32, const int* row_index, const int* m_dev, int M,
                             const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                             const void* W2a, const float* b2a, const void* W2b, const float* b2b,
                             const void* Wr, const float* br, const void* We, const float* be,
                             float* pe, void* Xk, void* stream) {
    MV2D_CHECK_ARG(A1 && A2 && Xfb && Xf32 && W1a && b1a && W1b && b1b && W2a && b2a && W2b && b2b && Wr && br && We && be && pe && Xk,
                   "mv2d_pe_fused: null pointer");
    MV2D_CHECK_ARG(M >= 0, "mv2d_pe_fused: M must be >= 0");
    if (M == 0) return MV2D_OK;
    PeParams p{(const unsigned short*)A1, (const unsigned short*)A2, (const unsigned short*)Xfb, Xf32, row_index, m_dev, M,
               (const unsigned short*)W1a, b1a, (const unsigned short*)W1b, b1b, (const unsigned short*)W2a, b2a,
               (const unsigned short*)W2b, b2b, (const unsigned short*)Wr, br, (const unsigned short*)We, be, pe, (unsigned short*)Xk, nullptr, 1};
    hipLaunchKernelGGL(pe_fused_kernel<false>, dim3(cdiv(M, BM)), dim3(64 * NW), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_pe_fused_tab2(const void* A1, const void* Xfb, const float* Xf32, const int* row_index, const int* m_dev, int M,
                                  const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                                  const void* Wr, const float* br, const void* We, const float* be,
                                  const float* sine_tab, int tab_period, float* pe, void* Xk, int shape, void* stream);   // pe_tab96.hip

extern "C" int mv2d_pe_fused_tab(const void* A1, const void* Xfb, const float* Xf32, const int* row_index, const int* m_dev, int M,
                                 const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                                 const void* Wr, const float* br, const void* We, const float* be,
                                 const float* sine_tab, int tab_period, float* pe, void* Xk, void* stream) {
    MV2D_CHECK_ARG(A1 && Xfb && (Xf32 || !Xk) && W1a && b1a && W1b && b1b && Wr && br && We && be && sine_tab && (pe || Xk), "mv2d_pe_fused_tab: null pointer");
    MV2D_CHECK_ARG(M >= 0 && tab_period > 0, "mv2d_pe_fused_tab: M must be >= 0 and tab_period > 0");
    if (M == 0) return MV2D_OK;
    // read per call (a host-side launch decision; graphs keep what they captured).  Default: pe_tab96.hip's kernel in its 96-row /
    // 8-wave shape; MV2D_PE_TAB_KERNEL=2: its two-64-row-blocks-per-CU shape; =64: this file's kernel (one wave per SIMD).  All three
    // give bit-identical results (110 / 134 / 145 us on 70 k rows).
    const char* sel = getenv("MV2D_PE_TAB_KERNEL");
    const int ksel = sel ? atoi(sel) : 96;
    if (ksel != 64)
        return mv2d_pe_fused_tab2(A1, Xfb, Xf32, row_index, m_dev, M, W1a, b1a, W1b, b1b, Wr, br, We, be, sine_tab, tab_period, pe, Xk,
                                  ksel == 2 ? 0 : 1, stream);
    // the bias staging reads b2a / b2b too: point them at valid memory (b1a has 1024 floats, b1b 256)
    PeParams p{(const unsigned short*)A1, (const unsigned short*)A1, (const unsigned short*)Xfb, Xf32, row_index, m_dev, M,
               (const unsigned short*)W1a, b1a, (const unsigned short*)W1b, b1b, (const unsigned short*)W1a, b1a,
               (const unsigned short*)W1b, b1b, (const unsigned short*)Wr, br, (const unsigned short*)We, be, pe, (unsigned short*)Xk, sine_tab,
               tab_period};
    hipLaunchKernelGGL(pe_fused_kernel<true>, dim3(cdiv(M, BM)), dim3(64 * NW), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

#ifdef MV2D_PE_TRACE
extern "C" int mv2d_pe_trace_read(long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pe_trace), n * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#endif
