// Shared device helpers for the MV2D gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MV2D_OK 0
#define MV2D_ERR_ARG (-1)
#define MV2D_ERR_LAUNCH (-2)

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one MFMA 16x16x32 A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // one MFMA 16x16 C/D fragment

extern "C" void mv2d_set_error(const char* msg);

#define MV2D_CHECK_ARG(cond, msg)                 \
    do {                                          \
        if (!(cond)) {                            \
            mv2d_set_error(msg);                  \
            return MV2D_ERR_ARG;                  \
        }                                         \
    } while (0)

#define MV2D_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) {                              \
            mv2d_set_error(hipGetErrorString(e__));           \
            return MV2D_ERR_LAUNCH;                           \
        }                                                     \
    } while (0)

// XCD-chunked block order: the dispatcher deals consecutive workgroups (x fastest, then y, z) round robin to the 8 XCDs, each with its own
// L2.  xcd_chunked(b, total) turns the dispatch index b into a logical index such that every XCD works through ONE contiguous range of
// logical indices: blocks that share operands (the column blocks of one row block, the row blocks of one weight set) then share an L2
// instead of fetching the operand once per XCD.  Speed / traffic only: any bijection is correct.
__device__ __forceinline__ int xcd_chunked(int b, int total) {
    const int x = b & 7, q = total >> 3, rem = total & 7;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + (b >> 3);
}

__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned int)h) << 16);
}

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// two fp32 -> packed bf16 pair with the gfx950 hardware convert (v_cvt_pk_bf16_f32, round-to-nearest-even)
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_cv;
    const f32x2_cv v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_cv));
}

// ------------------------------------------------------------------------------------------------------------------------------
// "key16": the 16-bit storage / MFMA operand format of the KEY SIDE of the hot path -- the gathered key / value rows of the cross
// attention, the RoI cells of the query generator's conv, the operands and the hidden layer of the PE MLPs, and the weights those
// kernels multiply them with.  Round 4: IEEE fp16 (11 significand bits; v_mfma_f32_16x16x32_f16 runs at the bf16 rate on the same
// bytes) instead of bf16 (8 bits): the key-side rounding, which dominated the deviation from the fp32 reference, shrinks 8 x.  Range
// guard: conversions SATURATE at +-65504 (FPN features / PE activations are O(1..100); a saturated element is finite and visible, an
// inf would poison the softmax row).  fp16 subnormals are kept by the MFMA under hipcc's default kernel mode (denorm mode 3; probed on
// MI355X by tools/f16_mfma_probe.hip), so a hi + lo pair carries ~22 bits down to an absolute 2^-24.
// -DMV2D_KEY16_BF16 builds the round-3 format for A/B runs (tools/build_variant.sh); mv2d_key16_format() reports which one a library has.
// The QUERY side (bf16x3 split precision on fp32 operands) and the generic tile GEMM (gemm_bf16.hip) are bf16 either way.
// ------------------------------------------------------------------------------------------------------------------------------
#ifdef MV2D_KEY16_BF16
#define MV2D_KEY16_IS_F16 0
typedef __attribute__((ext_vector_type(8))) __bf16 k16x8_t;
typedef __attribute__((ext_vector_type(4))) short k16x4_t;
__device__ __forceinline__ unsigned int pack_k16x2(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ unsigned short f32_to_k16(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ float k16_to_f32(unsigned short h) { return bf16_to_f32(h); }
__device__ __forceinline__ float k16_lo_of_pair(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float k16_hi_of_pair(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ f32x4_t mfma_k16_16x16x32(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(k16x8_t, a), __builtin_bit_cast(k16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma_k16_16x16x16(const uint2& a, const uint2& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(k16x4_t, a), __builtin_bit_cast(k16x4_t, b), c, 0, 0, 0);
}
#else
#define MV2D_KEY16_IS_F16 1
typedef __attribute__((ext_vector_type(8))) _Float16 k16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 k16x4_t;
// two fp32 -> packed fp16 pair, round-to-nearest-even, saturating at the largest finite fp16; NaN stays NaN (the clamp is written with
// comparisons, which are false for NaN)
// (v_med3_f32 returns the MINIMUM of its operands when one is NaN, so the NaN is put back by one compare + select: three instructions per value
//  instead of two compares + two selects; the qmap kernel's epilogue was 70 % clamp instructions, 16.3 -> 22.5 us when fp16 arrived)
__device__ __forceinline__ float k16_sat(float v) {
    const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    return v != v ? v : c;
}
__device__ __forceinline__ unsigned int pack_k16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_cv;
    const f32x2_cv v = {k16_sat(lo), k16_sat(hi)};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_cv));
}
__device__ __forceinline__ unsigned short f32_to_k16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)k16_sat(f)); }
__device__ __forceinline__ float k16_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float k16_lo_of_pair(unsigned int p) { return k16_to_f32((unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float k16_hi_of_pair(unsigned int p) { return k16_to_f32((unsigned short)(p >> 16)); }
__device__ __forceinline__ f32x4_t mfma_k16_16x16x32(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(k16x8_t, a), __builtin_bit_cast(k16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma_k16_16x16x16(const uint2& a, const uint2& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(k16x4_t, a), __builtin_bit_cast(k16x4_t, b), c, 0, 0, 0);
}
#endif
// key16 pair of relu(a), relu(b) for the hidden layers of the key-side MLPs: ReLU and the range clamp are ONE median (0 <= x <= 65504) per value,
// three instructions per pair with the conversion (the first fp16 build spent seven: compare + select ReLU, compare + select clamp, NaN put back;
// the PE kernel's hidden-layer epilogues were a quarter of its vector instructions).  A NaN becomes 0 here (v_med3_f32 returns the minimum of its
// operands for a NaN) -- the hidden layer is not where a poisoned input stays visible: the feature row itself goes into the key / value rows.
__device__ __forceinline__ unsigned int pack_k16x2_relu(float a, float b) {
#if MV2D_KEY16_IS_F16
    typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_cv;
    const f32x2_cv v = {__builtin_amdgcn_fmed3f(a, 0.f, 65504.f), __builtin_amdgcn_fmed3f(b, 0.f, 65504.f)};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_cv));
#else
    return pack_k16x2(a < 0.f ? 0.f : a, b < 0.f ? 0.f : b);
#endif
}
// the same for values KNOWN to lie inside the fp16 range (softmax probabilities): no clamp
__device__ __forceinline__ void split_k16x2_bounded(float a, float b, unsigned int& hi, unsigned int& lo) {
#if MV2D_KEY16_IS_F16
    typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_cv;
    const f32x2_cv v = {a, b};
    hi = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_cv));
    const f32x2_cv r = {a - k16_lo_of_pair(hi), b - k16_hi_of_pair(hi)};
    lo = __builtin_bit_cast(unsigned int, __builtin_convertvector(r, f16x2_cv));
#else
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - k16_lo_of_pair(hi), b - k16_hi_of_pair(hi));
#endif
}
// hi + lo split of two fp32 values into key16 pairs: x ~ hi + lo.  The remainder is taken from the CLAMPED value: a saturated (or infinite) input
// has hi = +-65504 exactly and therefore remainder 0, a NaN stays NaN in both parts, and the remainder of anything else is at most half an fp16 ulp
// (<= 16), so the second conversion needs no clamp.
// MV2D_F16_OVFL_MODE (a kernel-local #define, with mv2d_set_f16_ovfl() at the kernel's entry): MODE.FP16_OVFL makes every fp32 -> fp16 conversion of the
// wave clamp an overflow to +-65504 itself (NaN kept, a true infinity stays infinite), so the three clamp instructions per value go away.
__device__ __forceinline__ void mv2d_set_f16_ovfl() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }      // hwreg(HW_REG_MODE, 23, 1) = 1
__device__ __forceinline__ void split_k16x2(float a, float b, unsigned int& hi, unsigned int& lo) {
#if MV2D_KEY16_IS_F16
    typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_cv;
#ifdef MV2D_F16_OVFL_MODE
    const f32x2_cv v = {a, b};
#else
    const f32x2_cv v = {k16_sat(a), k16_sat(b)};
#endif
    hi = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_cv));
    const f32x2_cv r = {v[0] - k16_lo_of_pair(hi), v[1] - k16_hi_of_pair(hi)};
    lo = __builtin_bit_cast(unsigned int, __builtin_convertvector(r, f16x2_cv));
#else
    hi = pack_k16x2(a, b);
    lo = pack_k16x2(a - k16_lo_of_pair(hi), b - k16_hi_of_pair(hi));
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// "lo8" (round 6): the lo halves of the key / value rows of the index-exact route as 8-BIT floats -- OCP e4m3 (4 exponent, 3 mantissa bits, bias 7,
// max 448) of lo * 2^12 -- in 256-byte rows next to the 512-byte key16 (hi) rows: a (query, key) pair gathers 1.5 KB instead of 2 KB.  The remainder
// of an fp16 rounding is at most half an ulp, |lo| <= 2^-12 |x| 2^frac <= 2^-11 |x|, so lo * 2^12 <= 2 |x| lies in the normal range of the format
// [2^-6, 448] for 2^-7 <= |x| <= 224 (subnormal steps of 2^-9 below: absolute error <= 2^-22 there; saturation above, never reached by the head's
// feature rows) under ONE FIXED scale: no reduction over the rows, the producers write the bytes in their epilogues.  The lo part enters a product
// with relative weight 2^-12, its own rounding (2^-4 relative) therefore with 2^-16: the class logits move by <= 3e-7 of their range and all 17
// reference parity cases keep their ranked indices (profiles/r06_ablate_exact.txt, tools/ablate_lo8_cases.py; e5m2 -- 2 mantissa bits -- does not).
// The conversions are the hardware's (tools/probes/fp8_probe.hip checks both against a restatement of the OCP format: every byte, every fp16 pattern):
//   lo8_pack4:   two key16 lo PAIRS -> 4 bytes, round-to-nearest-even of fp16(lo) * 2^12 clamped to +-448 (a NaN stays NaN: 0x7f / 0xff)
//   lo8_pair:    bytes (2 w, 2 w + 1) of a dword -> the key16 pair fp16(e4m3 * 2^-12), subnormal results kept (v_cvt_scalef32_pk_f16_fp8)
// The route's results are those of fp16 lo rows that hold the dequantised values, bit for bit (tests/test_gpu_kernels.py).
// ------------------------------------------------------------------------------------------------------------------------------
#if MV2D_KEY16_IS_F16
__device__ __forceinline__ float lo8_clamp(float v) {
    const float c = __builtin_amdgcn_fmed3f(v, -448.f, 448.f);
    return v != v ? v : c;
}
__device__ __forceinline__ unsigned int lo8_pack4(unsigned int lo01, unsigned int lo23) {
    int e = 0;
    e = __builtin_amdgcn_cvt_pk_fp8_f32(lo8_clamp(k16_lo_of_pair(lo01) * 4096.f), lo8_clamp(k16_hi_of_pair(lo01) * 4096.f), e, false);
    e = __builtin_amdgcn_cvt_pk_fp8_f32(lo8_clamp(k16_lo_of_pair(lo23) * 4096.f), lo8_clamp(k16_hi_of_pair(lo23) * 4096.f), e, true);
    return (unsigned int)e;
}
// the same, and *flag |= 1 (device memory, may be NULL) when one of the four remainders leaves the e4m3 range (|lo| 2^12 > 448, i.e. |x| beyond ~224: that
// element keeps its hi half's precision only) -- the engine reports it (HeadEngine._check_capacity) instead of degrading silently
__device__ __forceinline__ unsigned int lo8_pack4_flag(unsigned int lo01, unsigned int lo23, int* flag) {
    const float a = k16_lo_of_pair(lo01) * 4096.f, b = k16_hi_of_pair(lo01) * 4096.f, c = k16_lo_of_pair(lo23) * 4096.f, d = k16_hi_of_pair(lo23) * 4096.f;
    // (the flag is read first: a frame full of such values would otherwise send one atomic per lane and store)
    if (flag && fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))) > 448.f && *reinterpret_cast<volatile int*>(flag) == 0) atomicOr(flag, 1);
    return lo8_pack4(lo01, lo23);
}
template <int W>
__device__ __forceinline__ unsigned int lo8_pair(unsigned int src) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_cv;
    const f16x2_cv v = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)src, 0.000244140625f, W != 0);
    return __builtin_bit_cast(unsigned int, v);
}
// 8 bytes -> the 16-byte key16 chunk of the same 8 channels
__device__ __forceinline__ uint4 lo8_chunk(const uint2& b) { return make_uint4(lo8_pair<0>(b.x), lo8_pair<1>(b.x), lo8_pair<0>(b.y), lo8_pair<1>(b.y)); }
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// "q16": the 16-bit SPLIT format of the query side (and of the split-precision PE kernel): every fp32 operand x is carried as a pair
// x ~ hi + lo and a product is a_hi.w_hi + a_lo.w_hi + a_hi.w_lo on three 16-bit MFMAs with fp32 accumulation.  Rounds 1-4: bf16 pairs
// (8 + 8 significand bits, 2^-17 per operand: 4.5e-6 relative on a 256-term dot product, the floor of the index-exact route's class-logit
// error).  Round 5: IEEE fp16 pairs (11 + 11 bits, 2^-23 per operand: 2.7e-7 on the same product -- tighter than an fp32 accumulation of
// the exact products, 7.6e-7) at the same MFMA rate and bytes.  Range: the conversions saturate at +-65504 (NaN kept) like the key side's;
// a lo part below 2^-14 is an fp16 subnormal, which the MFMA keeps (tools/f16_mfma_probe.hip): absolute error 2^-25 there.
// -DMV2D_Q16_BF16 builds the round-4 format; mv2d_q16_format() reports which one a library has (the weights are split by
// mv2d_split_q16x2 of the same library, so the two can never be mixed).
// ------------------------------------------------------------------------------------------------------------------------------
#if defined(MV2D_Q16_BF16) || !MV2D_KEY16_IS_F16
#define MV2D_Q16_IS_F16 0
typedef __attribute__((ext_vector_type(8))) __bf16 q16x8_t;
__device__ __forceinline__ unsigned short f32_to_q16(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ float q16_to_f32(unsigned short h) { return bf16_to_f32(h); }
__device__ __forceinline__ void split_q16x2(float a, float b, unsigned int& hi, unsigned int& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
template <class A, class B>
__device__ __forceinline__ f32x4_t mfma_q16_16x16x32(const A& a, const B& b, f32x4_t c, int = 0, int = 0, int = 0) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(q16x8_t, a), __builtin_bit_cast(q16x8_t, b), c, 0, 0, 0);
}
#else
#define MV2D_Q16_IS_F16 1
typedef __attribute__((ext_vector_type(8))) _Float16 q16x8_t;
__device__ __forceinline__ unsigned short f32_to_q16(float f) { return f32_to_k16(f); }
__device__ __forceinline__ float q16_to_f32(unsigned short h) { return k16_to_f32(h); }
__device__ __forceinline__ void split_q16x2(float a, float b, unsigned int& hi, unsigned int& lo) { split_k16x2(a, b, hi, lo); }
template <class A, class B>
__device__ __forceinline__ f32x4_t mfma_q16_16x16x32(const A& a, const B& b, f32x4_t c, int = 0, int = 0, int = 0) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(q16x8_t, a), __builtin_bit_cast(q16x8_t, b), c, 0, 0, 0);
}
#endif
// hi / lo of ONE value (LDS images written element-wise); the remainder is taken from the saturated value (see split_k16x2)
__device__ __forceinline__ void split_q16(float v, unsigned short& hi, unsigned short& lo) {
#if MV2D_Q16_IS_F16
    const float c = k16_sat(v);
    hi = __builtin_bit_cast(unsigned short, (_Float16)c);
    lo = __builtin_bit_cast(unsigned short, (_Float16)(c - (float)__builtin_bit_cast(_Float16, hi)));
#else
    hi = f32_to_bf16(v);
    lo = f32_to_bf16(v - bf16_to_f32(hi));
#endif
}
__device__ __forceinline__ void split_q16x4(const float4& v, uint2& hi, uint2& lo) {
    split_q16x2(v.x, v.y, hi.x, lo.x);
    split_q16x2(v.z, v.w, hi.y, lo.y);
}

// ReLU that keeps NaN like torch.relu (fmaxf(NaN, 0) would return 0 and hide a poisoned row).  (An integer maximum with 0 on the bit pattern is one
// instruction instead of two, but zeroes a NaN whose sign bit is set -- and the NaN rows of a fully masked query do carry it after the
// LayerNorm: tests/test_gpu_golden.py::test_fully_masked_row_golden fails with it.  Round 4.)
__device__ __forceinline__ float relu_f(float v) { return v < 0.f ? 0.f : v; }

// Reductions over the 64 lanes of a wave without the LDS crossbar, in the order of the xor butterfly they replace (32, 16, 8, 4, 2, 1):
// v_permlane32_swap, v_permlane16_swap, then DPP row rotations by 8 and 4 (after the step before them lanes j and j ^ 8, resp. j ^ 4, hold
// the same value, so the rotated partner carries exactly the xor partner's value) and the two quad permutes.  Bit for bit the result of six
// dependent ds_bpermute round trips.
#define MV2D_DPP_F(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float wave_sum(float v) {
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    v += MV2D_DPP_F(v, 0x128);                       // row_ror:8
    v += MV2D_DPP_F(v, 0x124);                       // row_ror:4
    v += MV2D_DPP_F(v, 0x4E);                        // quad_perm [2,3,0,1]
    v += MV2D_DPP_F(v, 0xB1);                        // quad_perm [1,0,3,2]
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    v = fmaxf(v, MV2D_DPP_F(v, 0x128));
    v = fmaxf(v, MV2D_DPP_F(v, 0x124));
    v = fmaxf(v, MV2D_DPP_F(v, 0x4E));
    v = fmaxf(v, MV2D_DPP_F(v, 0xB1));
    return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
