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

// ReLU that keeps NaN like torch.relu (fmaxf(NaN, 0) would return 0 and hide a poisoned row)
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
