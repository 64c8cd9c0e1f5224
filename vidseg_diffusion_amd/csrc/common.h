// Shared host/device helpers for libvidseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#define VS_OK 0
#define VS_ERR_ARG (-1)
#define VS_ERR_HIP (-2)
#define VS_ERR_UNSUPPORTED (-3)

extern thread_local char g_vs_err[512];

#define VS_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(g_vs_err, sizeof(g_vs_err), __VA_ARGS__);   \
        return (code);                                       \
    } while (0)

#define VS_REQUIRE(cond, ...)                                \
    do {                                                     \
        if (!(cond)) VS_FAIL(VS_ERR_ARG, __VA_ARGS__);       \
    } while (0)

#define VS_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) VS_FAIL(VS_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// Activation / weight storage format of the UNet path: IEEE fp16 by default -- what the reference computes in under CUDA
// autocast (11 significand bits; the dumped Q/K "attention maps" are fp16 anyway) -- or bf16 when built with
// -DVIDSEG_ACT_BF16.  Both are 16-bit and feed the same-rate MFMA (v_mfma_f32_32x32x16_{f16,bf16}); the kernels only touch
// the format through the four helpers below, so `bf16_t` simply means "raw 16-bit activation bits".
typedef unsigned short bf16_t;

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef __attribute__((ext_vector_type(2))) float vs_f32x2;
typedef __attribute__((ext_vector_type(8))) short vs_s16x8;
#ifdef VIDSEG_ACT_BF16
#define VIDSEG_ACT_IS_F16 0
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {       // round-to-nearest-even: v_cvt_pk_bf16_f32 on gfx950
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
typedef __attribute__((ext_vector_type(2))) __bf16 vs_bf16x2;
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(vs_f32x2{lo, hi}, vs_bf16x2));
}
__device__ __forceinline__ f32x16 mfma_32x32x16(vs_s16x8 a, vs_s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_16x16x32(vs_s16x8 a, vs_s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#else
#define VIDSEG_ACT_IS_F16 1
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {       // round-to-nearest-even v_cvt_f16_f32
    return __builtin_bit_cast(bf16_t, (_Float16)f);
}
typedef __attribute__((ext_vector_type(2))) _Float16 vs_f16x2;
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {   // RNE pair conversion
    return __builtin_bit_cast(unsigned, __builtin_convertvector(vs_f32x2{lo, hi}, vs_f16x2));
}
__device__ __forceinline__ f32x16 mfma_32x32x16(vs_s16x8 a, vs_s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_16x16x32(vs_s16x8 a, vs_s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#endif

// acc + a.lo * b.lo + a.hi * b.hi on a packed pair of activations (one v_dot2c_f32_{f16,bf16})
__device__ __forceinline__ float dot2_acc(unsigned a, unsigned b, float acc) {
#if VIDSEG_ACT_IS_F16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(vs_f16x2, a), __builtin_bit_cast(vs_f16x2, b), acc, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(vs_bf16x2, a), __builtin_bit_cast(vs_bf16x2, b), acc, false);
#endif
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Blocks reach the 8 XCDs round-robin by linear block id; give every XCD one contiguous chunk of the (head, query block) space so
// the query blocks of a (sample, head) share its K/V through one L2 instead of fetching it on all eight.
__device__ __forceinline__ void attn_block(int& bh, int& qb) {
    const int gx = gridDim.x;
    const long long tot = (long long)gx * gridDim.y;
    const long long lin = (long long)blockIdx.y * gx + blockIdx.x;
    const long long q = tot / 8, r = tot % 8, xcd = lin % 8, idx = lin / 8;
    const long long id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bh = (int)(id / gx);
    qb = (int)(id % gx);
}

// A split operand image has rows [hi | lo | third] of C values each (row stride 3 C).  Every GEMM consumer takes widths C % 64 == 0
// only and never reads the third plane (k_gemm_p7x / k_gemm_phx address planes 0, 1; the 3 C walk of the other kernels folds its last
// third back onto plane 0: GemmParams::a_fold), so the producers leave it unwritten there -- 4 instead of 6 bytes per value on
// HBM-bound passes (GroupNorm apply 27.7 -> 23.8 us, LayerNorm 35.1 -> 30.5 us per launch, about 2 ms per window).  Other widths keep
// [hi | lo | hi].
#define VS_THIRD_PLANE(C) (((C) & 63) != 0)

// Host-side override of a kernel-selection default, for tests and same-box A/B runs only: VAR="key=value,key=value".  The product
// path sets none of them; every kernel choice is a fixed function of the problem.  (VIDSEG_GEMM: csrc/gemm_conv.hip, VIDSEG_ATTN: the
// attention kernels of csrc/unet_ops.hip.)
static inline int vs_knob(const char* var, const char* key, int dflt) {
    const char* e = getenv(var);
    const size_t kl = strlen(key);
    while (e && *e) {
        const char* eq = strchr(e, '=');
        if (!eq) break;
        if ((size_t)(eq - e) == kl && !strncmp(e, key, kl)) return atoi(eq + 1);
        const char* c = strchr(eq, ',');
        if (!c) break;
        e = c + 1;
    }
    return dflt;
}
