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
//     ring, one contiguous 1 KB per fragment load, prefetched 3 steps ahead across the layer-1 / layer-2 boundary of a part (its
//     step list is unrolled at compile time, so hipcc keeps exact vmcnt counts);
//   * hidden activations never leave the CU; P1, G, P2 are combined in registers; one fp32 + one bf16 store per output.
// MFMAs run swapped (D^T = W.A^T): a lane ends with 4 consecutive columns of one row -> 8-byte LDS writes of the hidden layer,
// 16-byte global stores.  Same k order and the same bf16 rounding of the hidden layers as the GEMM route: results are bit-identical.
#include "common.h"

namespace {

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
    const unsigned short* A1; const unsigned short* A2; const unsigned short* Xfb; const float* Xf32; const int* m_dev; int M;
    const unsigned short* W1a; const float* b1a; const unsigned short* W1b; const float* b1b;
    const unsigned short* W2a; const float* b2a; const unsigned short* W2b; const float* b2b;
    const unsigned short* Wr; const float* br; const unsigned short* We; const float* be;
    float* pe; unsigned short* Xk;
};

// two-layer MLP on the block's 64 rows: acc2[i][j] (row tile i, column tile j of this wave's 64 output columns) =
// relu(A . W1^T + b1) . W2^T, A [64, K1] bf16 rows m0.. of `Ag` (ld = K1), hidden HID, output 256.
template <int K1, int HID>
__device__ __forceinline__ void run_mlp(unsigned char* __restrict__ As, unsigned char* __restrict__ Hs, const unsigned short* __restrict__ Ag,
                                        int m0, int M, const unsigned short* __restrict__ W1, const float* __restrict__ b1,
                                        const unsigned short* __restrict__ W2, f32x4_t acc2[RT][CT2], int tid) {
    constexpr int PITCH_A = K1 <= 256 ? 512 : 1024;       // power-of-two row pitch so that the chunk XOR stays inside the row
    constexpr int CPR = K1 / 8;                           // 16-byte chunks per input row
    constexpr int HPL = HID >= HP ? HP : HID;             // hidden columns per part (512, or 256 for the gate)
    constexpr int NH = HID / HPL;
    constexpr int TPW = HPL / 16 / NW;                    // hidden column tiles per wave and part ...
    constexpr int PT1 = TPW >= 4 ? 4 : TPW, NPASS = TPW / PT1;   // ... in passes of PT1
    constexpr int KS1 = K1 / 32, L1 = NPASS * KS1, L2 = HPL / 32;
    static_assert(L2 >= 4 && TPW >= 1 && CT2 >= 1, "the register ring runs through layer 1 into layer 2");
    constexpr int NT1 = HID / 16;
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;

    // ---- stage the input tile (rows beyond M clamp to the last valid row; their results are never stored)
    for (int c = tid; c < BM * CPR; c += 64 * NW) {
        const int row = c / CPR, chunk = c - row * CPR;
        const int m = min(m0 + row, M - 1);
        *reinterpret_cast<uint4*>(As + row * PITCH_A + ((chunk ^ (row & 15)) << 4)) =
            *reinterpret_cast<const uint4*>(Ag + (long long)m * K1 + chunk * 8);
    }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT2; ++j) acc2[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t acc1[RT][PT1];
    __syncthreads();

#pragma unroll 1
    for (int h = 0; h < NH; ++h) {                     // one resident part of the hidden layer at a time (runtime loop)
        // fragment group (4 consecutive column tiles = 4 KB) of step t of this part: compile-time offsets from two bases
        const unsigned short* w1b = W1 + ((long long)(h * HPL / 16 + wave * TPW) * 64 + lane) * 8;
        const unsigned short* w2b = W2 + (((long long)(h * (HPL / 32)) * 16 + wave * CT2) * 64 + lane) * 8;
        auto wptr = [&](int t) -> const unsigned short* {
            if (t < L1) { const int pass = t / KS1, ks = t - pass * KS1; return w1b + ((long long)ks * NT1 + pass * PT1) * 512; }
            return w2b + (long long)(t - L1) * 16 * 512;
        };
        Frag wq[4][4];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const unsigned short* wp = wptr(t);
#pragma unroll
            for (int j = 0; j < (t < L1 ? PT1 : CT2); ++j) wq[t & 3][j].u = *reinterpret_cast<const uint4*>(wp + j * 512);
        }
        // ---- layer 1 of this part (L1 steps), then layer 2 (L2 steps); the ring runs through (L1 is a multiple of 4)
#pragma unroll
        for (int t = 0; t < L1; ++t) {
            {
                const unsigned short* wp = wptr(t + 3);
#pragma unroll
                for (int j = 0; j < (t + 3 < L1 ? PT1 : CT2); ++j) wq[(t + 3) & 3][j].u = *reinterpret_cast<const uint4*>(wp + j * 512);
            }
            __builtin_amdgcn_sched_barrier(0);           // keep the prefetch 3 steps ahead
            const int pass = t / KS1, ks = t - pass * KS1;
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < PT1; ++j) acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            Frag a[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i)
                a[i].u = *reinterpret_cast<const uint4*>(As + (16 * i + fr) * PITCH_A + (((4 * ks + fg) ^ fr) << 4));
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < PT1; ++j)
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[t & 3][j].v, a[i].v, acc1[i][j], 0, 0, 0);
            if (ks == KS1 - 1) {
                // hidden tile of this pass: lane (fr, fg) holds hidden columns 4fg..4fg+3 of row 16i + fr -> bias, ReLU, bf16, 8-byte write
#pragma unroll
                for (int j = 0; j < PT1; ++j) {
                    const int lcol = (wave * TPW + pass * PT1 + j) * 16 + 4 * fg;             // column inside the resident part
                    const float4 bb = *reinterpret_cast<const float4*>(b1 + h * HPL + lcol);
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const uint2 hv = make_uint2(pack_bf16x2(relu_f(acc1[i][j][0] + bb.x), relu_f(acc1[i][j][1] + bb.y)),
                                                    pack_bf16x2(relu_f(acc1[i][j][2] + bb.z), relu_f(acc1[i][j][3] + bb.w)));
                        *reinterpret_cast<uint2*>(Hs + (16 * i + fr) * PITCH_H + (((lcol >> 3) ^ fr) << 4) + (lcol & 4) * 2) = hv;
                    }
                }
            }
        }
        __syncthreads();                               // the resident part of the hidden layer is complete
#pragma unroll
        for (int t2 = 0; t2 < L2; ++t2) {
            if (t2 + 3 < L2) {
                const unsigned short* wp = wptr(L1 + t2 + 3);
#pragma unroll
                for (int j = 0; j < CT2; ++j) wq[(L1 + t2 + 3) & 3][j].u = *reinterpret_cast<const uint4*>(wp + j * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
            Frag a[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i)
                a[i].u = *reinterpret_cast<const uint4*>(Hs + (16 * i + fr) * PITCH_H + (((4 * t2 + fg) ^ fr) << 4));
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT2; ++j)
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[(L1 + t2) & 3][j].v, a[i].v, acc2[i][j], 0, 0, 0);
        }
        __syncthreads();                               // everybody is done reading this part (and, after the last one, As / Hs are free)
    }
}

__global__ __launch_bounds__(64 * NW, 2) void pe_fused_kernel(PeParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + H_BYTES];
    unsigned char* As = smem;
    unsigned char* Hs = smem + A_BYTES;
    int M = p.M;
    if (p.m_dev) { const int md = *p.m_dev; M = md < M ? md : M; }
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;

    // The three MLPs hand their results over through the output buffer itself (every lane re-reads exactly the addresses it
    // wrote): carrying P1 and the gate in registers across the next MLP costs 64-128 more VGPRs and pushes the kernel into spills.
    f32x4_t acc[RT][CT2];
    // 1. gate = sigmoid(conv_expand(relu(conv_reduce(feat))))            -> pe (temporary)
    run_mlp<256, 256>(As, Hs, p.Xfb, m0, M, p.Wr, p.br, p.We, acc, tid);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = m0 + 16 * i + fr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < CT2; ++j) {
            const int n = wave * (CT2 * 16) + 16 * j + 4 * fg;
            const float4 bb = *reinterpret_cast<const float4*>(p.be + n);
            *reinterpret_cast<float4*>(p.pe + (long long)m * C + n) =
                make_float4(1.f / (1.f + __expf(-(acc[i][j][0] + bb.x))), 1.f / (1.f + __expf(-(acc[i][j][1] + bb.y))),
                            1.f / (1.f + __expf(-(acc[i][j][2] + bb.z))), 1.f / (1.f + __expf(-(acc[i][j][3] + bb.w))));
        }
    }
    // 2. P1 = position_encoder(A1);  Pg = (P1 + b) * gate                 -> pe (temporary)
    run_mlp<192, 1024>(As, Hs, p.A1, m0, M, p.W1a, p.b1a, p.W1b, acc, tid);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = m0 + 16 * i + fr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < CT2; ++j) {
            const int n = wave * (CT2 * 16) + 16 * j + 4 * fg;
            const float4 bb = *reinterpret_cast<const float4*>(p.b1b + n);
            float4* dst = reinterpret_cast<float4*>(p.pe + (long long)m * C + n);
            const float4 g = *dst;
            *dst = make_float4((acc[i][j][0] + bb.x) * g.x, (acc[i][j][1] + bb.y) * g.y, (acc[i][j][2] + bb.z) * g.z, (acc[i][j][3] + bb.w) * g.w);
        }
    }
    // 3. P2 = adapt_pos3d(A2);  pe = (P2 + b) + Pg;  Xk = bf16(pe + feat)
    run_mlp<384, 1024>(As, Hs, p.A2, m0, M, p.W2a, p.b2a, p.W2b, acc, tid);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = m0 + 16 * i + fr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < CT2; ++j) {
            const int n = wave * (CT2 * 16) + 16 * j + 4 * fg;
            const float4 bb = *reinterpret_cast<const float4*>(p.b2b + n);
            float4* dst = reinterpret_cast<float4*>(p.pe + (long long)m * C + n);
            const float4 pg = *dst;
            const float4 v = make_float4((acc[i][j][0] + bb.x) + pg.x, (acc[i][j][1] + bb.y) + pg.y, (acc[i][j][2] + bb.z) + pg.z, (acc[i][j][3] + bb.w) + pg.w);
            *dst = v;
            const float4 f = *reinterpret_cast<const float4*>(p.Xf32 + (long long)m * C + n);
            *reinterpret_cast<uint2*>(p.Xk + (long long)m * C + n) = make_uint2(pack_bf16x2(v.x + f.x, v.y + f.y), pack_bf16x2(v.z + f.z, v.w + f.w));
        }
    }
}

}  // namespace

// C-ABI: see include/mv2d_hip.h
extern "C" int mv2d_pe_fused(const void* A1, const void* A2, const void* Xfb, const float* Xf32, const int* m_dev, int M,
                             const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                             const void* W2a, const float* b2a, const void* W2b, const float* b2b,
                             const void* Wr, const float* br, const void* We, const float* be,
                             float* pe, void* Xk, void* stream) {
    MV2D_CHECK_ARG(A1 && A2 && Xfb && Xf32 && W1a && b1a && W1b && b1b && W2a && b2a && W2b && b2b && Wr && br && We && be && pe && Xk,
                   "mv2d_pe_fused: null pointer");
    MV2D_CHECK_ARG(M >= 0, "mv2d_pe_fused: M must be >= 0");
    if (M == 0) return MV2D_OK;
    PeParams p{(const unsigned short*)A1, (const unsigned short*)A2, (const unsigned short*)Xfb, Xf32, m_dev, M,
               (const unsigned short*)W1a, b1a, (const unsigned short*)W1b, b1b, (const unsigned short*)W2a, b2a,
               (const unsigned short*)W2b, b2b, (const unsigned short*)Wr, br, (const unsigned short*)We, be, pe, (unsigned short*)Xk};
    hipLaunchKernelGGL(pe_fused_kernel, dim3(cdiv(M, BM)), dim3(64 * NW), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
