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

// One-time setup that is PER DEVICE (hipFuncSetAttribute: a process that drives several GPUs needs it on each one) and safe to race on:
// two threads may both run the setup, which is idempotent; nobody launches before `mark()` of its own pass.
struct VsOncePerDevice {
    unsigned long long done = 0;
    static unsigned long long bit() {
        int d = 0;
        (void)hipGetDevice(&d);
        return 1ull << (d & 63);
    }
    bool needs() const { return !(__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit()); }
    void mark() { __atomic_fetch_or(&done, bit(), __ATOMIC_RELEASE); }
};
// A per-device tri-state cache (-1 unknown, 0 no, 1 yes) for "does this device accept the attribute".
struct VsPerDeviceFlag {
    signed char v[64];
    VsPerDeviceFlag() { memset(v, -1, sizeof(v)); }
    signed char& here() {
        int d = 0;
        (void)hipGetDevice(&d);
        return v[d & 63];
    }
};

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

#if VIDSEG_ACT_IS_F16
// ---- the exact mode's operand split: x -> hi = fp16(x), lo = fp16(x - hi)
// x must be ONE fp32 value for both lines.  Without the barrier hipcc (-ffp-contract=fast) fuses the conversion with x's producer
// per use: for x = a * s it emitted hi = v_cvt_pk_f16_f32(v_mul_f32(a, s)) for the operand image (two roundings) but
// v_fma_mixlo_f16(a, s, 0) (one rounding of the exact product) as the hi that lo is taken against -- the two differ when the fp32
// product sits on an fp16 rounding tie, and hi + lo is then one fp16 ulp (2^-12 relative) off: 6 of 10240 query rows of a
// 1024-token attention came out 5e-5 wrong (tools/lab/x_attn_diag2.py), every one holding such a tie.
__device__ __forceinline__ void split_hl(float x, f16& hi, f16& lo) {
    asm volatile("" : "+v"(x));
    hi = (f16)x;
    lo = (f16)(x - (float)hi);
}
// Two values at once, packed the way the MFMA operand wants them (x0 in the low half): the same roundings as split_hl in 4 VALU
// instructions instead of 11 -- v_cvt_pk_f16_f32 for the hi pair, x - hi as v_fma_mix_f32 (hi read as an fp16 half, the fma exact),
// v_cvt_pk_f16_f32 for the lo pair.  For values with no foldable producer (the attention's probabilities come out of v_exp_f32).
typedef __attribute__((ext_vector_type(2))) float xf32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 xf16x2;
__device__ __forceinline__ void split_hl2(float x0, float x1, unsigned& hi2, unsigned& lo2) {
    asm volatile("" : "+v"(x0), "+v"(x1));
    const xf16x2 h = __builtin_convertvector(xf32x2{x0, x1}, xf16x2);
    const unsigned hu = __builtin_bit_cast(unsigned, h);
    float r0, r1;                                               // (hipcc turns fma(ext(h), -1, x) back into a conversion and a subtraction)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hu), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hu), "v"(x1));
    const xf16x2 l = __builtin_convertvector(xf32x2{r0, r1}, xf16x2);
    hi2 = hu;
    lo2 = __builtin_bit_cast(unsigned, l);
}
// Four / eight values at once as the image's 8- / 16-byte cells (the same roundings as split_hl, pair by pair through split_hl2).
typedef __attribute__((ext_vector_type(2))) unsigned int vs_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int vs_u32x4;
__device__ __forceinline__ void split_hl4(const float (&x)[4], f16x4& hi, f16x4& lo) {
    unsigned h0, l0, h1, l1;
    split_hl2(x[0], x[1], h0, l0);
    split_hl2(x[2], x[3], h1, l1);
    hi = __builtin_bit_cast(f16x4, vs_u32x2{h0, h1});
    lo = __builtin_bit_cast(f16x4, vs_u32x2{l0, l1});
}
__device__ __forceinline__ void split_hl8(const float (&x)[8], f16x8& hi, f16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_hl2(x[2 * e], x[2 * e + 1], h[e], l[e]);
    hi = __builtin_bit_cast(f16x8, vs_u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(f16x8, vs_u32x4{l[0], l[1], l[2], l[3]});
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

// erf in fp32, branch-free, for the exact mode's GELU (sgm/modules/attention.py:89-96 -> F.gelu, erf form).  Two minimax fits --
// |a| <= 0.9277: a + a p(a^2); beyond: 1 - exp(q(|a|)) -- both evaluated and selected per lane (a wave of gate values straddles the
// switch point anyway, so libm's branches ran both sides under exec masks: ~45 VALU instructions per value in the GEGLU epilogue,
// which made that epilogue, not the K loop, the longer half of a K' = 960 tile).  Max error 0.994 ulp against float64 erf over all
// of [0, 6] (checked exhaustively on the host with the same fmaf sequence; libm's erff: 0.90 ulp); ~20 instructions.  The
// exponential is v_exp_f32 of the product with log2 e: its argument's rounding is <= 0.5 ulp of a value whose result is <= 0.4, i.e.
// < 3e-8 absolute on erf.
__device__ __forceinline__ float erf_f32(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896340736f), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? big : q;
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
