// The PE block of the key side, with the sine branch folded into a table (MU/pe.py:36-48,64-77,150-166; the branch adapt_pos3d(sine)
// depends only on the weights and the padding geometry, the engine evaluates it once per (weights, geometry) into `sine_tab`):
//   P1 = position_encoder(A1)                                   192 -> 1024 -> 256     (MU/pe.py:64-77, 158-160)
//   G  = sigmoid(conv_expand(relu(conv_reduce(feat))))          256 -> 256 -> 256      (MU/pe.py:36-48, 162-166)
//   pe = tab[position] + P1 * G ,  Xk = key16(pe + feat)        tab = adapt_pos3d(sine) + bias, constant per (weights, padding geometry)
//   (Xk optional: the S path's keys are RoI-aligned rows, it only needs pe -- no feature-row read, a third less traffic;
//    pe optional: the T path's keys are the Xk rows, nothing reads pe there)
//
// Its round-1/2 predecessor (pe_fused_kernel: 4 waves, 64 rows, ONE wave per SIMD, one block per CU; retired in round 4) spent less than half
// of a block's life in its MFMA loops: prologue, the staging of the second input tile and above all the output phase (1.5 KB written and 2 KB
// read per row) ran with the matrix pipe idle, and since all blocks of a round move in lockstep those phases hit HBM as a burst (11 B/clk/CU)
// while the MFMA phases left it idle.  With the table this kernel is HBM-bound, not MFMA-bound (356 MB per 70 k rows against 83 GFLOP).
// All 16-bit operands (input rows, weights, hidden layer, Xk) are in the key-side format of common.h (key16 = fp16 since round 4).
// Two shapes of one template, same fragment-major weights and k order (bit-identical results):
//   * <RT=4, NW=4, CT=4>: 64 rows, 4 waves, 71 KB of LDS and <= 256 registers -> TWO INDEPENDENT BLOCKS PER CU: one block's memory
//     phases run under the other's MFMA loops and the blocks drift out of lockstep;
//   * <RT=6, NW=8, CT=2>: 96 rows, 8 waves (2 per SIMD) in one block per CU; a weight fragment feeds 6 MFMAs instead of 4 and the
//     hidden tile is double-buffered (one barrier per part).
// The gate comes LAST so that only two accumulator sets are ever live (P1 and the running one).
#include "common.h"

#ifdef MV2D_PE_TRACE
__device__ long long g_pe96_trace[64];
#define PE_STAMP(i) do { if (blockIdx.x == MV2D_PE_TRACE && threadIdx.x == 0) g_pe96_trace[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PE_STAMP(i) do {} while (0)
#endif

namespace {

constexpr int C = 256;
constexpr int PITCH = 512;                                  // bytes per row of the LDS tiles (256 bf16), 16-byte chunk c of row r at c ^ (r & 15)
enum { B_R = 0, B_E = 256, B_1A = 512, B_1B = 1536, B_FLOATS = 1792 };
constexpr int OT_PITCH = 36;                                // floats per row of a wave's output tile [BM][32 columns]
struct PFrag { uint4 u; };
typedef unsigned int pt_u32x4 __attribute__((ext_vector_type(4)));   // staging registers (arrays of HIP's uint4 struct end up in scratch)

struct PeTabParams {
    const unsigned short* A1; const unsigned short* Xfb; const float* Xf32; const int* row_index; const int* m_dev; int M;
    const unsigned short* W1a; const float* b1a; const unsigned short* W1b; const float* b1b;
    const unsigned short* Wr; const float* br; const unsigned short* We; const float* be;
    const float* sine_tab; int tab_period; float* pe; unsigned short* Xk;
};

// Shape of a block: RT row tiles of 16 rows, NW waves, CT column tiles of 16 per wave (NW * CT = 16: a part = 256 columns),
// HB hidden buffers, RING weight steps in flight, EXP = experiment bits (1: no weight loads after the first RING steps, 2: no activation reads)
template <int RT_, int NW_, int CT_, int HB_, int RING_, int EXP_>
struct Shape {
    static constexpr int RT = RT_, NW = NW_, CT = CT_, HB = HB_, RING = RING_, EXP = EXP_;
    static constexpr int BM = 16 * RT, NTHR = 64 * NW, A_BYTES = BM * PITCH, H_BYTES = BM * PITCH;
    static constexpr int SMEM = A_BYTES + HB * H_BYTES + B_FLOATS * 4;
    static_assert(NW * CT == 16, "a part is 256 columns");
    static_assert(NW * BM * OT_PITCH * 4 <= A_BYTES + HB * H_BYTES, "the waves' output tiles fit into the LDS tiles they replace");
};

// ---- the 72 k-steps of a block as one compile-time schedule: parts 0..3 = hidden columns 256 p .. of the frustum MLP (6 + 8 steps
// each), part 4 = the gate (8 + 8 steps).  Step T consumes CT weight fragments (this wave's column tiles).
constexpr int NSTEP = 4 * 14 + 16;
__host__ __device__ constexpr int part_of(int T) { return T < 56 ? T / 14 : 4; }
__host__ __device__ constexpr int first_of(int p) { return p * 14; }
__host__ __device__ constexpr int ks1_of(int p) { return p == 4 ? 8 : 6; }

struct WBase { const unsigned short* wr; const unsigned short* we; const unsigned short* w1a; const unsigned short* w1b; };   // + lane * 8 + wave * CT tiles

template <int T>
__device__ __forceinline__ const unsigned short* step_ptr(const WBase& w) {
    constexpr int p = part_of(T), t = T - first_of(p), ks1 = ks1_of(p);
    if constexpr (p == 4) {
        if constexpr (t < ks1) return w.wr + (long long)(t * 16) * 512;                       // Wr  [ks][16 tiles]
        else return w.we + (long long)((t - ks1) * 16) * 512;                                  // We  [ks][16 tiles]
    } else {
        if constexpr (t < ks1) return w.w1a + (long long)(t * 64 + p * 16) * 512;              // W1a [ks][64 tiles], this part's 16 tiles
        else return w.w1b + (long long)((p * 8 + (t - ks1)) * 16) * 512;                       // W1b [32 k-steps][16 tiles]
    }
}

template <class S, int T>
__device__ __forceinline__ void ring_load(PFrag (&wq)[S::RING][S::CT], const WBase& w) {
    if constexpr (T < NSTEP && !((S::EXP & 1) && T >= S::RING)) {
        const unsigned short* ptr = step_ptr<T>(w);
#pragma unroll
        for (int j = 0; j < S::CT; ++j) wq[T % S::RING][j].u = *reinterpret_cast<const uint4*>(ptr + 512 * j);
    }
}

template <class S>
__device__ __forceinline__ void load_a(PFrag (&a)[S::RT], const unsigned char* L, int kstep, int fr, int fg) {
#pragma unroll
    for (int i = 0; i < S::RT; ++i) a[i].u = *reinterpret_cast<const uint4*>(L + (16 * i + fr) * PITCH + (((4 * kstep + fg) ^ fr) << 4));
}

// N k-steps of one layer: D^T[column][row] += W . A^T (swapped: a lane ends with 4 consecutive columns of one row).  The activation
// fragments of step K + 1 are read from LDS before the MFMAs of step K issue.
template <class S, int T0, int N, int K = 0>
__device__ __forceinline__ void steps(f32x4_t (&acc)[S::RT][S::CT], PFrag (&wq)[S::RING][S::CT], PFrag (&a)[2][S::RT], const WBase& w, const unsigned char* L,
                                      int fr, int fg) {
    if constexpr (K < N) {
        if constexpr (K == 0) load_a<S>(a[0], L, 0, fr, fg);
        ring_load<S, T0 + K + S::RING - 1>(wq, w);
        if constexpr (K + 1 < N && !(S::EXP & 2)) load_a<S>(a[(K + 1) & 1], L, K + 1, fr, fg);
        __builtin_amdgcn_sched_barrier(0);             // the loads stay ahead of the MFMAs (the scheduler otherwise sinks them to their uses)
#pragma unroll
        for (int i = 0; i < S::RT; ++i)
#pragma unroll
            for (int j = 0; j < S::CT; ++j)
                acc[i][j] = mfma_k16_16x16x32(wq[(T0 + K) % S::RING][j].u, a[(S::EXP & 2) ? 0 : (K & 1)][i].u, acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        steps<S, T0, N, K + 1>(acc, wq, a, w, L, fr, fg);
    }
}

// layer 1 of part P into a hidden buffer: lane (fr, fg) holds hidden columns lcol..lcol+3 of row 16 i + fr -> bias, ReLU, key16, 8-byte
// write.  Barrier after: the tile is complete.  With two hidden buffers the one written here was last read two barriers ago; with one,
// a barrier before the stores waits for the previous part's layer 2.
template <class S, int P>
__device__ __forceinline__ void layer1(PFrag (&wq)[S::RING][S::CT], PFrag (&a)[2][S::RT], const WBase& w, const unsigned char* As, unsigned char* Hb,
                                       const float* bias /* LDS, this part's 256 */, int wave, int fr, int fg) {
    // the accumulators START at the bias (lane (fr, fg) holds hidden columns lcol..lcol+3 of column tile j for every row tile): one vector add
    // per hidden value less than adding it in the epilogue
    f32x4_t acc1[S::RT][S::CT];
#pragma unroll
    for (int j = 0; j < S::CT; ++j) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + (wave * S::CT + j) * 16 + 4 * fg);
#pragma unroll
        for (int i = 0; i < S::RT; ++i) acc1[i][j] = f32x4_t{bb.x, bb.y, bb.z, bb.w};
    }
    steps<S, first_of(P), ks1_of(P)>(acc1, wq, a, w, As, fr, fg);
    if constexpr (S::HB == 1 && P > 0 && P < 4) __syncthreads();   // (the gate's layer 1 follows a block barrier anyway)
#pragma unroll
    for (int j = 0; j < S::CT; ++j) {
        const int lcol = (wave * S::CT + j) * 16 + 4 * fg;
#pragma unroll
        for (int i = 0; i < S::RT; ++i) {
            const uint2 hv = make_uint2(pack_k16x2_relu(acc1[i][j][0], acc1[i][j][1]), pack_k16x2_relu(acc1[i][j][2], acc1[i][j][3]));
            *reinterpret_cast<uint2*>(Hb + (16 * i + fr) * PITCH + (((lcol >> 3) ^ fr) << 4) + (lcol & 4) * 2) = hv;
        }
    }
    __syncthreads();
}

template <class S>
__global__ __launch_bounds__(S::NTHR, 2) void pe_tab_kernel(PeTabParams p) {
    constexpr int RT = S::RT, CT = S::CT, BM = S::BM, NTHR = S::NTHR;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S::SMEM];
    unsigned char* As = smem;
    unsigned char* Hs0 = smem + S::A_BYTES;
    unsigned char* Hs1 = S::HB == 2 ? smem + S::A_BYTES + S::H_BYTES : Hs0;
    float* Bs = reinterpret_cast<float*>(smem + S::A_BYTES + S::HB * S::H_BYTES);
    int M = p.M;
    if (p.m_dev) { const int md = *p.m_dev; M = md < M ? md : M; }
    const int m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const long long lo = (long long)lane * 8 + (long long)wave * CT * 512;
    const WBase w{p.Wr + lo, p.We + lo, p.W1a + lo, p.W1b + lo};
    PE_STAMP(0);
    PFrag wq[S::RING][CT], a[2][RT];
    ring_load<S, 0>(wq, w);
    ring_load<S, 1>(wq, w);
    if constexpr (S::RING > 3) ring_load<S, 2>(wq, w);
    {
        // biases -> LDS: [br | be | b1a | b1b] as 448 float4; the frustum rows of the tile (192 channels = 24 chunks per row) -> As
        constexpr int NB = (B_FLOATS / 4 + NTHR - 1) / NTHR, NA = (BM * 24 + NTHR - 1) / NTHR;
        float4 bv[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int t = tid + NTHR * i;
            if (t < B_FLOATS / 4) {
                const float* src = t < 64 ? p.br + 4 * t : t < 128 ? p.be + 4 * (t - 64) : t < 384 ? p.b1a + 4 * (t - 128) : p.b1b + 4 * (t - 384);
                bv[i] = *reinterpret_cast<const float4*>(src);
            }
        }
        pt_u32x4 sa[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + NTHR * i, row = c / 24, chunk = c - row * 24;
            if (c < BM * 24) sa[i] = *reinterpret_cast<const pt_u32x4*>(p.A1 + (long long)min(m0 + row, M - 1) * 192 + chunk * 8);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid + NTHR * i < B_FLOATS / 4) *reinterpret_cast<float4*>(Bs + 4 * (tid + NTHR * i)) = bv[i];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + NTHR * i, row = c / 24, chunk = c - row * 24;
            if (c < BM * 24) *reinterpret_cast<pt_u32x4*>(As + row * PITCH + ((chunk ^ (row & 15)) << 4)) = sa[i];
        }
    }
    __syncthreads();
    PE_STAMP(1);
    const int n0 = wave * CT * 16 + 4 * fg;             // this lane's 4 output columns of column tile j start at n0 + 16 j
    f32x4_t accf[RT][CT];                               // P1 = position_encoder(A1), bias added at the end

    // ---- 1. P1 in four parts of 256 hidden columns
#pragma unroll
    for (int j = 0; j < CT; ++j) {                      // (P1 starts at its bias b1b, the gate accumulator below at be)
        const float4 fb = *reinterpret_cast<const float4*>(Bs + B_1B + n0 + 16 * j);
#pragma unroll
        for (int i = 0; i < RT; ++i) accf[i][j] = f32x4_t{fb.x, fb.y, fb.z, fb.w};
    }
    layer1<S, 0>(wq, a, w, As, Hs0, Bs + B_1A, wave, fr, fg);
    PE_STAMP(2);
    steps<S, first_of(0) + 6, 8>(accf, wq, a, w, Hs0, fr, fg);
    PE_STAMP(3);
    layer1<S, 1>(wq, a, w, As, Hs1, Bs + B_1A + 256, wave, fr, fg);
    PE_STAMP(4);
    steps<S, first_of(1) + 6, 8>(accf, wq, a, w, Hs1, fr, fg);
    PE_STAMP(5);
    layer1<S, 2>(wq, a, w, As, Hs0, Bs + B_1A + 512, wave, fr, fg);
    PE_STAMP(6);
    steps<S, first_of(2) + 6, 8>(accf, wq, a, w, Hs0, fr, fg);
    PE_STAMP(7);
    layer1<S, 3>(wq, a, w, As, Hs1, Bs + B_1A + 768, wave, fr, fg);        // after its barrier nobody reads As any more
    PE_STAMP(8);
    {
        // the feature rows of the tile (256 channels = 32 chunks per row) travel while the last layer 2 runs
        constexpr int NX = BM * 32 / NTHR;
        pt_u32x4 sa[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + NTHR * i, row = c >> 5, chunk = c & 31;
            sa[i] = *reinterpret_cast<const pt_u32x4*>(p.Xfb + (long long)min(m0 + row, M - 1) * C + chunk * 8);
        }
        steps<S, first_of(3) + 6, 8>(accf, wq, a, w, Hs1, fr, fg);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + NTHR * i, row = c >> 5, chunk = c & 31;
            *reinterpret_cast<pt_u32x4*>(As + row * PITCH + ((chunk ^ (row & 15)) << 4)) = sa[i];
        }
    }
    PE_STAMP(9);
    __syncthreads();                                   // the feature tile is in As (and the last layer 2 is done with the hidden tile)
    PE_STAMP(10);
    // ---- 2. the gate
    f32x4_t acc[RT][CT];
    layer1<S, 4>(wq, a, w, As, Hs0, Bs + B_R, wave, fr, fg);
    PE_STAMP(11);
    // read-back mapping of the output phase: lane -> (row r0 + 8 k, columns c4..c4+3 of 32); the row indices travel under layer 2
    constexpr int NK = BM / 8;
    const int c4 = (lane & 7) * 4, r0 = lane >> 3;
    int ri[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int m = min(m0 + 8 * k + r0, M - 1);
        ri[k] = p.row_index ? p.row_index[m] : m;
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const float4 eb = *reinterpret_cast<const float4*>(Bs + B_E + n0 + 16 * j);
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i][j] = f32x4_t{eb.x, eb.y, eb.z, eb.w};
    }
    steps<S, first_of(4) + 8, 8>(acc, wq, a, w, Hs0, fr, fg);
    PE_STAMP(12);
    // ---- 3. pe = tab + (P1 + b) * gate, Xk = key16(pe + feat): through a wave-private LDS tile [BM rows][32 columns], then whole
    // 128-byte row pieces (the MFMA layout would store 16 rows x 16 bytes per instruction).  The feature and table rows of the first
    // 32 columns are requested before the gate math (their latency is this phase's floor).
    float4 fv[NK], tv[NK];
    auto request = [&](int jp) {
        const long long gcol = wave * CT * 16 + jp * 32 + c4;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (p.Xk) fv[k] = *reinterpret_cast<const float4*>(p.Xf32 + (long long)ri[k] * C + gcol);   // (uniform: the S path has no use for Xk)
            tv[k] = *reinterpret_cast<const float4*>(p.sine_tab + (long long)(ri[k] % p.tab_period) * C + gcol);
        }
    };
    request(0);
    // the sigmoid in place (both accumulators started at their biases)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            // (v_rcp_f32, 1 ulp, instead of the IEEE division sequence: ~8 instructions less per value; the gate multiplies a value that is rounded
            //  to key16 or added to an fp32 table row right after)
            const f32x4_t g{__builtin_amdgcn_rcpf(1.f + __expf(-acc[i][j][0])), __builtin_amdgcn_rcpf(1.f + __expf(-acc[i][j][1])),
                            __builtin_amdgcn_rcpf(1.f + __expf(-acc[i][j][2])), __builtin_amdgcn_rcpf(1.f + __expf(-acc[i][j][3]))};
            accf[i][j] = f32x4_t{accf[i][j][0] * g[0], accf[i][j][1] * g[1], accf[i][j][2] * g[2], accf[i][j][3] * g[3]};
        }
    }
    __syncthreads();                                   // all LDS tiles free: they become the waves' output tiles
    PE_STAMP(13);
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
            v = make_float4(v.x + tv[k].x, v.y + tv[k].y, v.z + tv[k].z, v.w + tv[k].w);
            if (m < M) {
                if (p.pe) *reinterpret_cast<float4*>(p.pe + (long long)m * C + gcol) = v;
                if (p.Xk)
                    *reinterpret_cast<uint2*>(p.Xk + (long long)m * C + gcol) =
                        make_uint2(pack_k16x2(v.x + fv[k].x, v.y + fv[k].y), pack_k16x2(v.z + fv[k].z, v.w + fv[k].w));
            }
        }
        if (jp + 1 < CT / 2) request(jp + 1);
        PE_STAMP(14 + jp);
    }
}

template <class S>
void launch(const PeTabParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(pe_tab_kernel<S>, dim3(cdiv(p.M, S::BM)), dim3(S::NTHR), 0, stream, p);
}

}  // namespace

// mv2d_pe_fused_tab (include/mv2d_hip.h) = shape 1; mv2d_pe_fused_tab2 exposes the shape for the kernel tests:
// shape 1 = 96 rows x 8 waves, one block per CU (default); 0 = 64 rows x 4 waves, two blocks per CU (bit-identical, slower: 134 vs 110 us on 70 k rows)
extern "C" int mv2d_pe_fused_tab2(const void* A1, const void* Xfb, const float* Xf32, const int* row_index, const int* m_dev, int M,
                                  const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                                  const void* Wr, const float* br, const void* We, const float* be,
                                  const float* sine_tab, int tab_period, float* pe, void* Xk, int shape, void* stream) {
    MV2D_CHECK_ARG(A1 && Xfb && (Xf32 || !Xk) && W1a && b1a && W1b && b1b && Wr && br && We && be && sine_tab && (pe || Xk), "mv2d_pe_fused_tab: null pointer");
    MV2D_CHECK_ARG(M >= 0 && tab_period > 0, "mv2d_pe_fused_tab: M must be >= 0 and tab_period > 0");
    if (M == 0) return MV2D_OK;
    PeTabParams p{(const unsigned short*)A1, (const unsigned short*)Xfb, Xf32, row_index, m_dev, M, (const unsigned short*)W1a, b1a,
                  (const unsigned short*)W1b, b1b, (const unsigned short*)Wr, br, (const unsigned short*)We, be, sine_tab, tab_period, pe,
                  (unsigned short*)Xk};
    hipStream_t st = (hipStream_t)stream;
    if (shape == 1) launch<Shape<6, 8, 2, 2, 4, 0>>(p, st);
    else launch<Shape<4, 4, 4, 1, 3, 0>>(p, st);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}

extern "C" int mv2d_pe_fused_tab(const void* A1, const void* Xfb, const float* Xf32, const int* row_index, const int* m_dev, int M,
                                 const void* W1a, const float* b1a, const void* W1b, const float* b1b,
                                 const void* Wr, const float* br, const void* We, const float* be,
                                 const float* sine_tab, int tab_period, float* pe, void* Xk, void* stream) {
    return mv2d_pe_fused_tab2(A1, Xfb, Xf32, row_index, m_dev, M, W1a, b1a, W1b, b1b, Wr, br, We, be, sine_tab, tab_period, pe, Xk, 1, stream);
}

#ifdef MV2D_PE_TRACE
extern "C" int mv2d_pe96_trace_read(long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pe96_trace), n * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#endif
