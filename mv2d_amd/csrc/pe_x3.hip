// The PE block of the key side in SPLIT PRECISION (index-exact route; gfx950 / CDNA4, wave64), one launch:
//   P1 = position_encoder(A1)                                   192 -> 1024 -> 256     (MU/pe.py:64-77, 158-160)
//   G  = sigmoid(conv_expand(relu(conv_reduce(feat))))          256 -> 256 -> 256      (MU/pe.py:36-48, 162-166)
//   pe = tab[position] + P1 * G                                  tab = adapt_pos3d(sine) + bias, constant per (weights, padding geometry)
//   T path: key rows Xk = pe + feat and value rows Xv = feat, each as a key16 hi + lo pair (what xattn_tile_kernel<.., XLO> gathers)
// on UNROUNDED fp32 inputs (frustum rows from pe_inputs_kernel<true>, feature rows read from the map): every product is
// a_hi w_hi + a_lo w_hi + a_hi w_lo on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (operands split into bf16 hi / lo, 2^-17 per operand,
// the hidden layer split when it is written to LDS) -- the arithmetic of the K-concatenated GEMM chain it replaces
// ([a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo]^T on the plain tile GEMM, round 3), which moved the 1024-wide hidden layer through HBM as
// [hi | lo | hi] (860 MB out + 860 MB back per 16-sample launch) in four launches + two operand-split passes + two row-split passes:
// 977 + 116 us (S path) / 1733 + 202 + 237 us (T path) per 16-sample launch.
//
// Structure = pe_tab_kernel's (pe_tab96.hip) 64-row shape: 4 waves, wave w owns column tiles 4w..4w+3 of every 256-column part for all 4 row
// tiles, fragment-major weights streamed straight from L2 through a register ring (hi and lo streams), the hidden layer in parts of 256
// columns through LDS, the gate LAST.  hi and lo images of the input tile and of the hidden tile live side by side in LDS (128 KB): one
// block per CU, one wave per SIMD (<= 512 registers), 48 MFMAs per k-step per wave -- the kernel is matrix-pipe bound by construction
// (3 x the MFMAs of the default kernel on the same loads).
#include "common.h"

#ifdef MV2D_PX_TRACE
__device__ long long g_px_trace[32];
#define PX_STAMP(i) do { if (blockIdx.x == MV2D_PX_TRACE && threadIdx.x == 0) g_px_trace[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PX_STAMP(i) do {} while (0)
#endif

namespace {

constexpr int C = 256;
constexpr int PITCH = 512;                                  // bytes per row of an LDS image (256 bf16), 16-byte chunk c of row r at c ^ (r & 15)
#ifndef MV2D_PX_RING
#define MV2D_PX_RING 3
#endif
constexpr int RT = 4, NW = 4, CT = 4, BM = 16 * RT, NTHR = 64 * NW, RING = MV2D_PX_RING;
constexpr int IMG = BM * PITCH;                             // one 64-row image: 32 KB
enum { B_R = 0, B_E = 256, B_1A = 512, B_1B = 1536, B_FLOATS = 1792 };
constexpr int OT_PITCH = 36;                                // floats per row of a wave's output tile [BM][32 columns]
constexpr int SMEM = 4 * IMG + B_FLOATS * 4;                // A hi | A lo | H hi | H lo | biases = 135 KB
static_assert(NW * BM * OT_PITCH * 4 <= 4 * IMG, "the waves' output tiles fit into the LDS images they replace");

typedef q16x8_t px_bf16x8;      // common.h "q16": fp16 pairs since round 5
struct XFrag { uint4 h, l; };

struct PeX3Params {
    const float* A1; const float* Xmap; const int* row_index; const int* m_dev; int M;
    const unsigned short* W1a_h; const unsigned short* W1a_l; const float* b1a; const unsigned short* W1b_h; const unsigned short* W1b_l; const float* b1b;
    const unsigned short* Wr_h; const unsigned short* Wr_l; const float* br; const unsigned short* We_h; const unsigned short* We_l; const float* be;
    const float* sine_tab; int tab_period;
    float* pe; unsigned short* Xk_hi; unsigned short* Xk_lo; unsigned short* Xv_hi; unsigned short* Xv_lo;
    int lo8;                                                 // the lo row outputs are 256-byte e4m3 rows (common.h "lo8")
    int* lo8_flag;                                           // |= 1 when a lo remainder leaves the e4m3 range (may be NULL)
    int pe_at_index;                                         // pe row m is written at row row_index[m] (a position-indexed map) instead of row m
};

// ---- the 72 k-steps of a block as one compile-time schedule (pe_tab96.hip): parts 0..3 = hidden columns 256 p .. of the frustum MLP
// (6 + 8 steps each), part 4 = the gate (8 + 8 steps).  Step T consumes CT weight fragments of the hi and of the lo stream.
constexpr int NSTEP = 4 * 14 + 16;
__host__ __device__ constexpr int part_of(int T) { return T < 56 ? T / 14 : 4; }
__host__ __device__ constexpr int first_of(int p) { return p * 14; }
__host__ __device__ constexpr int ks1_of(int p) { return p == 4 ? 8 : 6; }

struct WBase { const unsigned short* wr[2]; const unsigned short* we[2]; const unsigned short* w1a[2]; const unsigned short* w1b[2]; };   // [hi, lo], + lane * 8 + wave * CT tiles

template <int T>
__device__ __forceinline__ long long step_off() {
    constexpr int p = part_of(T), t = T - first_of(p), ks1 = ks1_of(p);
    if constexpr (p == 4) {
        if constexpr (t < ks1) return (long long)(t * 16) * 512;                               // Wr  [ks][16 tiles]
        else return (long long)((t - ks1) * 16) * 512;                                         // We  [ks][16 tiles]
    } else {
        if constexpr (t < ks1) return (long long)(t * 64 + p * 16) * 512;                      // W1a [ks][64 tiles], this part's 16 tiles
        else return (long long)((p * 8 + (t - ks1)) * 16) * 512;                               // W1b [32 k-steps][16 tiles]
    }
}
template <int T>
__device__ __forceinline__ const unsigned short* step_base(const WBase& w, int part) {
    constexpr int p = part_of(T), t = T - first_of(p), ks1 = ks1_of(p);
    if constexpr (p == 4) return t < ks1 ? w.wr[part] : w.we[part];
    else return t < ks1 ? w.w1a[part] : w.w1b[part];
}

template <int T>
__device__ __forceinline__ void ring_load(XFrag (&wq)[RING][CT], const WBase& w) {
    if constexpr (T < NSTEP) {
        const unsigned short* ph = step_base<T>(w, 0) + step_off<T>();
        const unsigned short* pl = step_base<T>(w, 1) + step_off<T>();
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            wq[T % RING][j].h = *reinterpret_cast<const uint4*>(ph + 512 * j);
            wq[T % RING][j].l = *reinterpret_cast<const uint4*>(pl + 512 * j);
        }
    }
}

__device__ __forceinline__ void load_a(XFrag (&a)[RT], const unsigned char* Lh, const unsigned char* Ll, int kstep, int fr, int fg) {
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int off = (16 * i + fr) * PITCH + (((4 * kstep + fg) ^ fr) << 4);
        a[i].h = *reinterpret_cast<const uint4*>(Lh + off);
        a[i].l = *reinterpret_cast<const uint4*>(Ll + off);
    }
}

// N k-steps of one layer.  The activation fragments of step K + 1 are read from LDS before the MFMAs of step K issue.
template <int T0, int N, int K = 0>
__device__ __forceinline__ void steps(f32x4_t (&acc)[RT][CT], XFrag (&wq)[RING][CT], XFrag (&a)[2][RT], const WBase& w, const unsigned char* Lh,
                                      const unsigned char* Ll, int fr, int fg) {
    if constexpr (K < N) {
        if constexpr (K == 0) load_a(a[0], Lh, Ll, 0, fr, fg);
        ring_load<T0 + K + RING - 1>(wq, w);
        if constexpr (K + 1 < N) load_a(a[(K + 1) & 1], Lh, Ll, K + 1, fr, fg);
        __builtin_amdgcn_sched_barrier(0);             // the loads stay ahead of the MFMAs
        // product-major: 16 independent MFMAs between two that accumulate into the same tile (a dependent MFMA waits ~8 passes for its input)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const XFrag& wf = wq[(T0 + K) % RING][j];
                    const XFrag& af = a[K & 1][i];
                    acc[i][j] = mfma_q16_16x16x32(t == 0 ? wf.l : wf.h, t == 1 ? af.l : af.h, acc[i][j]);
                }
        __builtin_amdgcn_sched_barrier(0);
        steps<T0, N, K + 1>(acc, wq, a, w, Lh, Ll, fr, fg);
    }
}

__device__ __forceinline__ void zero_acc(f32x4_t (&acc)[RT][CT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ void split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
    split_q16x2(a, b, hi.x, lo.x);
    split_q16x2(c, d, hi.y, lo.y);
}

// layer 1 of part P into the hidden images: lane (fr, fg) holds hidden columns lcol..lcol+3 of row 16 i + fr -> bias, ReLU, hi / lo split,
// two 8-byte writes.  A barrier before the stores waits for the previous part's layer 2 (one hidden buffer), one after completes the tile.
template <int P>
__device__ __forceinline__ void layer1(XFrag (&wq)[RING][CT], XFrag (&a)[2][RT], const WBase& w, const unsigned char* Ah, const unsigned char* Al,
                                       unsigned char* Hh, unsigned char* Hl, const float* bias /* LDS, this part's 256 */, int wave, int fr, int fg) {
    f32x4_t acc1[RT][CT];
    zero_acc(acc1);
    steps<first_of(P), ks1_of(P)>(acc1, wq, a, w, Ah, Al, fr, fg);
    if constexpr (P > 0 && P < 4) __syncthreads();   // (the gate's layer 1 follows a block barrier anyway)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int lcol = (wave * CT + j) * 16 + 4 * fg;
        const float4 bb = *reinterpret_cast<const float4*>(bias + lcol);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            uint2 hv, lv;
            split4(relu_f(acc1[i][j][0] + bb.x), relu_f(acc1[i][j][1] + bb.y), relu_f(acc1[i][j][2] + bb.z), relu_f(acc1[i][j][3] + bb.w), hv, lv);
            const int off = (16 * i + fr) * PITCH + (((lcol >> 3) ^ fr) << 4) + (lcol & 4) * 2;
            *reinterpret_cast<uint2*>(Hh + off) = hv;
            *reinterpret_cast<uint2*>(Hl + off) = lv;
        }
    }
    __syncthreads();
}

// fp32 rows -> hi / lo LDS images: NCH 16-byte chunks (8 columns) per row, thread t moves float4 pieces (half a chunk each).  Two phases, so
// that the loads can be in flight under MFMA work: stage_load issues them, stage_commit splits and writes the images.
template <int NCH>
struct Stage {
    static constexpr int PIECES = BM * NCH * 2, PER = PIECES / NTHR;
    static_assert(PIECES % NTHR == 0, "");
    float4 v[PER];
    __device__ __forceinline__ void load(const float* __restrict__ src, long long ld, const int* __restrict__ ridx, int m0, int M, int tid) {
        // all row indices first, then all rows: written as one loop, the compiler waited for index i AND row i - 1 (vmcnt(0)) in front of every row -- PER
        // dependent round trips per block instead of two (round 6, tools/isa_waits.sh)
#ifdef MV2D_PX_ROUND5_STAGE      // (timing A/B: the round-5 form)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + NTHR * i, row = c / (2 * NCH), piece = c - row * (2 * NCH);
            const int m = min(m0 + row, M - 1);
            const long long r = ridx ? ridx[m] : m;
            v[i] = *reinterpret_cast<const float4*>(src + r * ld + piece * 4);
        }
#else
        int r[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + NTHR * i, row = c / (2 * NCH);
            const int m = min(m0 + row, M - 1);
            r[i] = ridx ? ridx[m] : m;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + NTHR * i, row = c / (2 * NCH), piece = c - row * (2 * NCH);
            v[i] = *reinterpret_cast<const float4*>(src + (long long)r[i] * ld + piece * 4);
        }
#endif
    }
    __device__ __forceinline__ void commit(unsigned char* Lh, unsigned char* Ll, int tid) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + NTHR * i, row = c / (2 * NCH), piece = c - row * (2 * NCH), chunk = piece >> 1;
            uint2 hv, lv;
            split4(v[i].x, v[i].y, v[i].z, v[i].w, hv, lv);
            const int off = row * PITCH + ((chunk ^ (row & 15)) << 4) + (piece & 1) * 8;
            *reinterpret_cast<uint2*>(Lh + off) = hv;
            *reinterpret_cast<uint2*>(Ll + off) = lv;
        }
    }
};

// The feature rows of a tile are a GATHER (64 rows x 1 KB through row_index) that all blocks of a round request at the same moment; loads return in
// order, so the weight-ring wait behind them stalls for the whole gather (round-5 stamps: 18 k of a block's 141 k cycles in front of layer 1 of
// part 3).  MV2D_PX_TOUCH: every thread reads ONE word of two of the tile's 512 cache lines early -- 1 = in the prologue (the frustum rows are waited
// for there anyway), 2 = in front of part 2 -- so that the real loads find their lines in L2.
#ifndef MV2D_PX_TOUCH
#define MV2D_PX_TOUCH 2
#endif
#define PX_TOUCH_ISSUE()                                                                                   \
    do {                                                                                                   \
        const int c0_ = tid, c1_ = tid + NTHR;                                                             \
        const int ma_ = min(m0 + (c0_ >> 3), M - 1), mb_ = min(m0 + (c1_ >> 3), M - 1);                    \
        const long long ra_ = p.row_index ? p.row_index[ma_] : ma_, rb_ = p.row_index ? p.row_index[mb_] : mb_; \
        touch0 = p.Xmap[ra_ * C + (c0_ & 7) * 32];                                                         \
        touch1 = p.Xmap[rb_ * C + (c1_ & 7) * 32];                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)

__global__ __launch_bounds__(NTHR, 1) void pe_x3_kernel(PeX3Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* Ah = smem;
    unsigned char* Al = smem + IMG;
    unsigned char* Hh = smem + 2 * IMG;
    unsigned char* Hl = smem + 3 * IMG;
    float* Bs = reinterpret_cast<float*>(smem + 4 * IMG);
    int M = p.M;
    if (p.m_dev) { const int md = *p.m_dev; M = md < M ? md : M; }
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const long long lo = (long long)lane * 8 + (long long)wave * CT * 512;
    const WBase w{{p.Wr_h + lo, p.Wr_l + lo}, {p.We_h + lo, p.We_l + lo}, {p.W1a_h + lo, p.W1a_l + lo}, {p.W1b_h + lo, p.W1b_l + lo}};
    XFrag wq[RING][CT], a[2][RT];
    float touch0 = 0.f, touch1 = 0.f;
    PX_STAMP(0);
    ring_load<0>(wq, w);
    ring_load<1>(wq, w);
    if constexpr (RING > 3) ring_load<2>(wq, w);
    if constexpr (RING > 4) ring_load<3>(wq, w);
    if constexpr (RING > 5) ring_load<4>(wq, w);
    {
        // biases -> LDS: [br | be | b1a | b1b] as 448 float4
        constexpr int NB = (B_FLOATS / 4 + NTHR - 1) / NTHR;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int t = tid + NTHR * i;
            if (t < B_FLOATS / 4) {
                const float* src = t < 64 ? p.br + 4 * t : t < 128 ? p.be + 4 * (t - 64) : t < 384 ? p.b1a + 4 * (t - 128) : p.b1b + 4 * (t - 384);
                *reinterpret_cast<float4*>(Bs + 4 * t) = *reinterpret_cast<const float4*>(src);
            }
        }
        // the frustum rows of the tile (192 channels = 24 chunks per row), fp32 -> hi / lo images
        Stage<24> st;
        st.load(p.A1, 192, nullptr, m0, M, tid);
        st.commit(Ah, Al, tid);
    }
#if MV2D_PX_TOUCH == 1
    PX_TOUCH_ISSUE();
#endif
    __syncthreads();
    PX_STAMP(1);
    const int n0 = wave * CT * 16 + 4 * fg;             // this lane's 4 output columns of column tile j start at n0 + 16 j
    f32x4_t accf[RT][CT];                               // P1 = position_encoder(A1), bias added at the end

    // ---- 1. P1 in four parts of 256 hidden columns
    zero_acc(accf);
    layer1<0>(wq, a, w, Ah, Al, Hh, Hl, Bs + B_1A, wave, fr, fg);
    PX_STAMP(2);
    steps<first_of(0) + 6, 8>(accf, wq, a, w, Hh, Hl, fr, fg);
    PX_STAMP(3);
    layer1<1>(wq, a, w, Ah, Al, Hh, Hl, Bs + B_1A + 256, wave, fr, fg);
    PX_STAMP(4);
    steps<first_of(1) + 6, 8>(accf, wq, a, w, Hh, Hl, fr, fg);
    PX_STAMP(5);
#if MV2D_PX_TOUCH == 2
    PX_TOUCH_ISSUE();
#endif
    layer1<2>(wq, a, w, Ah, Al, Hh, Hl, Bs + B_1A + 512, wave, fr, fg);
    PX_STAMP(6);
    steps<first_of(2) + 6, 8>(accf, wq, a, w, Hh, Hl, fr, fg);
    PX_STAMP(7);
    // the feature rows of the tile (256 channels = 32 chunks per row, gathered through row_index) are requested a whole part ahead: they travel
    // under the MFMAs of layer 1 (stamps of the first version: 22 k of a block's 145 k cycles waited for them right here)
    Stage<32> fs;
    fs.load(p.Xmap, C, p.row_index, m0, M, tid);
#ifndef MV2D_PX_ROUND5_STAGE
    __builtin_amdgcn_sched_barrier(0);                 // (without it hipcc sinks the 16 row loads to fs.commit below: 16 dependent round trips per block, tools/isa_waits.sh)
#endif
#if MV2D_PX_TOUCH
    asm volatile("" ::"v"(touch0), "v"(touch1));      // the touch loads are complete at the latest here (their lines sit in L2 for the loads above)
#endif
    layer1<3>(wq, a, w, Ah, Al, Hh, Hl, Bs + B_1A + 768, wave, fr, fg);        // after its barrier nobody reads the frustum images any more
    PX_STAMP(8);
    fs.commit(Ah, Al, tid);                            // other waves may still run the last layer 2 (hidden images only)
    PX_STAMP(9);
    steps<first_of(3) + 6, 8>(accf, wq, a, w, Hh, Hl, fr, fg);
    PX_STAMP(10);
    __syncthreads();                                   // the feature tile is in the A images (and the last layer 2 is done with the hidden tile)
    PX_STAMP(11);
    // ---- 2. the gate
    f32x4_t acc[RT][CT];
    {
        // layer 1 of the gate: no barrier needed in front of its stores (the barrier above), P = 4
        f32x4_t acc1[RT][CT];
        zero_acc(acc1);
        steps<first_of(4), 8>(acc1, wq, a, w, Ah, Al, fr, fg);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int lcol = (wave * CT + j) * 16 + 4 * fg;
            const float4 bb = *reinterpret_cast<const float4*>(Bs + B_R + lcol);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                uint2 hv, lv;
                split4(relu_f(acc1[i][j][0] + bb.x), relu_f(acc1[i][j][1] + bb.y), relu_f(acc1[i][j][2] + bb.z), relu_f(acc1[i][j][3] + bb.w), hv, lv);
                const int off = (16 * i + fr) * PITCH + (((lcol >> 3) ^ fr) << 4) + (lcol & 4) * 2;
                *reinterpret_cast<uint2*>(Hh + off) = hv;
                *reinterpret_cast<uint2*>(Hl + off) = lv;
            }
        }
        __syncthreads();
    }
    PX_STAMP(12);
    // read-back mapping of the output phase: lane -> (row r0 + 8 k, columns c4..c4+3 of 32); the row indices travel under layer 2
    constexpr int NK = BM / 8;
    const int c4 = (lane & 7) * 4, r0 = lane >> 3;
    int ri[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int m = min(m0 + 8 * k + r0, M - 1);
        ri[k] = p.row_index ? p.row_index[m] : m;
    }
    // the table rows (and, T path, the fp32 feature rows) of the first 32 output columns are requested before the gate's second layer
    const bool rows16 = p.Xk_hi != nullptr;
    float4 tvq[NK], fvq[NK];
    auto request = [&](int jp) {
        const long long gcol = wave * CT * 16 + jp * 32 + c4;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            tvq[k] = *reinterpret_cast<const float4*>(p.sine_tab + (long long)(ri[k] % p.tab_period) * C + gcol);
            if (rows16) fvq[k] = *reinterpret_cast<const float4*>(p.Xmap + (long long)ri[k] * C + gcol);
        }
    };
    request(0);
    zero_acc(acc);
    steps<first_of(4) + 8, 8>(acc, wq, a, w, Hh, Hl, fr, fg);
    PX_STAMP(13);
    // ---- 3. pe = tab + (P1 + b) * gate; T path: Xk = pe + feat, Xv = feat as key16 hi + lo pairs.  Through a wave-private LDS tile
    // [BM rows][32 columns], then whole 128-byte row pieces.  The sigmoid in place, the bias of P1:
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const float4 eb = *reinterpret_cast<const float4*>(Bs + B_E + n0 + 16 * j);
        const float4 fb = *reinterpret_cast<const float4*>(Bs + B_1B + n0 + 16 * j);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            // (accurate exp: this route is compared at fp32 rounding level)
            const f32x4_t g{1.f / (1.f + expf(-(acc[i][j][0] + eb.x))), 1.f / (1.f + expf(-(acc[i][j][1] + eb.y))),
                            1.f / (1.f + expf(-(acc[i][j][2] + eb.z))), 1.f / (1.f + expf(-(acc[i][j][3] + eb.w)))};
            accf[i][j] = f32x4_t{(accf[i][j][0] + fb.x) * g[0], (accf[i][j][1] + fb.y) * g[1], (accf[i][j][2] + fb.z) * g[2], (accf[i][j][3] + fb.w) * g[3]};
        }
    }
    __syncthreads();                                   // all LDS images free: they become the waves' output tiles
    PX_STAMP(14);
    float* ot = reinterpret_cast<float*>(smem) + wave * (BM * OT_PITCH);
#pragma unroll
    for (int jp = 0; jp < CT / 2; ++jp) {               // 32 columns (two column tiles) at a time
        if (jp > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < RT; ++i)
                *reinterpret_cast<float4*>(ot + (16 * i + fr) * OT_PITCH + 16 * j + 4 * fg) =
                    make_float4(accf[i][2 * jp + j][0], accf[i][2 * jp + j][1], accf[i][2 * jp + j][2], accf[i][2 * jp + j][3]);
        __builtin_amdgcn_wave_barrier();               // the tile is read back by the same wave only
        const long long gcol = wave * CT * 16 + jp * 32 + c4;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int row = 8 * k + r0, m = m0 + row;
            float4 v = *reinterpret_cast<const float4*>(ot + row * OT_PITCH + c4);
            const float4 tv = tvq[k];
            v = make_float4(v.x + tv.x, v.y + tv.y, v.z + tv.z, v.w + tv.w);
            if (m < M) {
                if (p.pe) *reinterpret_cast<float4*>(p.pe + (long long)(p.pe_at_index ? ri[k] : m) * C + gcol) = v;
                if (rows16) {
                    const float4 f = fvq[k];
                    uint2 h, l;
                    split_k16x2(v.x + f.x, v.y + f.y, h.x, l.x);
                    split_k16x2(v.z + f.z, v.w + f.w, h.y, l.y);
                    *reinterpret_cast<uint2*>(p.Xk_hi + (long long)m * C + gcol) = h;
                    if (p.lo8) *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(p.Xk_lo) + (long long)m * C + gcol) = lo8_pack4_flag(l.x, l.y, p.lo8_flag);
                    else *reinterpret_cast<uint2*>(p.Xk_lo + (long long)m * C + gcol) = l;
                    split_k16x2(f.x, f.y, h.x, l.x);
                    split_k16x2(f.z, f.w, h.y, l.y);
                    *reinterpret_cast<uint2*>(p.Xv_hi + (long long)m * C + gcol) = h;
                    if (p.lo8) *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(p.Xv_lo) + (long long)m * C + gcol) = lo8_pack4_flag(l.x, l.y, p.lo8_flag);
                    else *reinterpret_cast<uint2*>(p.Xv_lo + (long long)m * C + gcol) = l;
                }
            }
        }
        if (jp + 1 < CT / 2) request(jp + 1);
        PX_STAMP(15 + jp);
    }
}

}  // namespace

#ifdef MV2D_PX_TRACE
extern "C" int mv2d_px_trace_read(long long* host, int n) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_px_trace), n * sizeof(long long)) == hipSuccess ? 0 : -2; }
#endif

// C-ABI: include/mv2d_hip.h
extern "C" int mv2d_pe_fused_x3(const float* A1, const float* Xmap, const int* row_index, const int* m_dev, int M,
                                const void* W1a_hi, const void* W1a_lo, const float* b1a, const void* W1b_hi, const void* W1b_lo, const float* b1b,
                                const void* Wr_hi, const void* Wr_lo, const float* br, const void* We_hi, const void* We_lo, const float* be,
                                const float* sine_tab, int tab_period, float* pe, void* Xk_hi, void* Xk_lo, void* Xv_hi, void* Xv_lo, int lo_fmt, int pe_at_index, int* lo8_flag, void* stream) {
    MV2D_CHECK_ARG(A1 && Xmap && W1a_hi && W1a_lo && b1a && W1b_hi && W1b_lo && b1b && Wr_hi && Wr_lo && br && We_hi && We_lo && be && sine_tab,
                   "mv2d_pe_fused_x3: null pointer");
    MV2D_CHECK_ARG(pe || Xk_hi, "mv2d_pe_fused_x3: no output");
    MV2D_CHECK_ARG(!pe_at_index || (pe && row_index), "mv2d_pe_fused_x3: pe_at_index needs pe and row_index");
    MV2D_CHECK_ARG((Xk_hi != nullptr) == (Xk_lo != nullptr) && (Xk_hi != nullptr) == (Xv_hi != nullptr) && (Xk_hi != nullptr) == (Xv_lo != nullptr),
                   "mv2d_pe_fused_x3: the four key / value row outputs come together");
    MV2D_CHECK_ARG(M >= 0 && tab_period > 0, "mv2d_pe_fused_x3: M must be >= 0 and tab_period > 0");
    MV2D_CHECK_ARG(lo_fmt == 0 || lo_fmt == 1, "mv2d_pe_fused_x3: lo_fmt is 0 (key16 lo rows) or 1 (e4m3 lo rows)");
    MV2D_CHECK_ARG(((uintptr_t)A1 & 15) == 0 && ((uintptr_t)Xmap & 15) == 0 && ((uintptr_t)sine_tab & 15) == 0, "mv2d_pe_fused_x3: rows must be 16-byte aligned");
    if (M == 0) return MV2D_OK;
    PeX3Params p{A1, Xmap, row_index, m_dev, M, (const unsigned short*)W1a_hi, (const unsigned short*)W1a_lo, b1a, (const unsigned short*)W1b_hi,
                 (const unsigned short*)W1b_lo, b1b, (const unsigned short*)Wr_hi, (const unsigned short*)Wr_lo, br, (const unsigned short*)We_hi,
                 (const unsigned short*)We_lo, be, sine_tab, tab_period, pe, (unsigned short*)Xk_hi, (unsigned short*)Xk_lo, (unsigned short*)Xv_hi,
                 (unsigned short*)Xv_lo, lo_fmt, lo8_flag, pe_at_index};
    hipLaunchKernelGGL(pe_x3_kernel, dim3(cdiv(M, BM)), dim3(NTHR), 0, (hipStream_t)stream, p);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
