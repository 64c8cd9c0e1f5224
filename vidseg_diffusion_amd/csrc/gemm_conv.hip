// bf16 MFMA implicit-GEMM for gfx950: one kernel family serves every dense contraction of the UNet
//   * 3x3 convolutions (stride 1/2, optional fused nearest-2x upsample, optional two-source channel concat)
//   * 1x1 convolutions / nn.Linear (A[m][k] rows, optional two-source concat)
// with fused epilogues (bias, per-sample embedding vector, residual add, SiLU, GEGLU, fp16 tap copy,
// fp32 output).  out[m][n] = sum_k A(m,k) * W[n][k];  W is [N][K] row-major (K contiguous) -- for a conv
// the weight is pre-packed to k = (kh*3+kw)*Cin + c so that with NHWC activations every 64-wide K chunk is
// one contiguous 128-byte run of one input pixel.
//
// Structure: 256 threads = 4 waves, wave tile 64x64 built from v_mfma_f32_32x32x16_bf16 (2x2 fragments),
// block tile 128x128 (2x2 waves) or 256x64 (4x1 waves), BK = 64, LDS double buffer with a 16-byte-slot XOR
// swizzle, global->register prefetch of the next K chunk issued before the MFMAs of the current one.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct GemmParams {
    // A operand
    const bf16_t* x0;
    const bf16_t* x1;        // second concat source or nullptr
    int C0, C1;              // channels of each source (Cin = C0 + C1)
    int Hin, Win;            // stored input height/width (conv) ; unused for ksize == 1
    int Hout, Wout;
    int ksize, stride, up;   // ksize 1 or 3; stride 1/2; up 1/2 (nearest upsample of the input before the conv)
    // B operand
    const bf16_t* w;         // [N][K]
    int N, K;
    long long M;
    // epilogue
    const float* bias;       // [N] (already permuted for GEGLU) or nullptr
    const float* rowvec;     // per-sample vector: rowvec[(m / rows_per_sample) * rv_stride + n] or nullptr
    int rv_stride, rows_per_sample;
    const bf16_t* residual;  // [M][ldr] or nullptr
    int ldr;
    bf16_t* out;             // [M][ldo] or nullptr
    int ldo;
    float* out_f32;          // [M][ldo] or nullptr
    f16* tap;                // fp16 copy of columns [0, tap_cols) with leading dim tap_ld, or nullptr
    f16* tap2;               // fp16 copy of columns [tap_cols, 2*tap_cols) (same leading dim), or nullptr
    int tap_cols, tap_ld;
    int act;                 // 0 none, 1 SiLU, 2 GEGLU (32-column interleaved x|gate groups)
};

#define BK 64

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

template <int BM, int BN>
__global__ void __launch_bounds__(256) k_gemm_conv(GemmParams p) {
    constexpr int WN = BN / 64;                       // waves along N
    constexpr int AR = BM / 32;                       // A rows per thread (16-byte chunks)
    constexpr int BR = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: buf b: A[BM][64] bf16 then B[BN][64] bf16
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF = A_BYTES + B_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: consecutive tiles along N share the A rows; keep them on one XCD's L2
    const long long tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.N + BN - 1) / BN;
    const long long nwg = tiles_m * tiles_n;
    long long bid = blockIdx.x;
    {
        const long long q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long tm = bid / tiles_n;
    const int tn = (int)(bid % tiles_n);
    const long long m0 = tm * BM;
    const int n0 = tn * BN;

    const int Cin = p.C0 + p.C1;
    const int chunk = tid & 7;                        // 16-byte chunk inside the 128-byte K run
    const int rbase = tid >> 3;                       // 0..31

    // per-thread A row descriptors
    int a_b[AR], a_oh[AR], a_ow[AR];
    bool a_ok[AR];
    const int HWo = p.Hout * p.Wout;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const long long m = m0 + rbase + 32 * i;
        a_ok[i] = m < p.M;
        const long long mm = a_ok[i] ? m : 0;
        if (p.ksize == 1) {
            a_b[i] = 0;
            a_oh[i] = 0;
            a_ow[i] = 0;
        } else {
            a_b[i] = (int)(mm / HWo);
            const int rem = (int)(mm % HWo);
            a_oh[i] = rem / p.Wout;
            a_ow[i] = rem % p.Wout;
        }
    }
    u32x4 ra[AR], rb[BR];

    auto load_regs = [&](int k0) {
        const int tap = k0 / Cin;                      // uniform
        const int c0 = k0 - tap * Cin;
        const bf16_t* src = p.x0;
        int Cs = p.C0, cc = c0;
        if (c0 >= p.C0) {
            src = p.x1;
            Cs = p.C1;
            cc = c0 - p.C0;
        }
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (a_ok[i]) {
                long long pix;
                bool ok = true;
                if (p.ksize == 1) {
                    pix = m0 + rbase + 32 * i;
                } else {
                    const int ih = a_oh[i] * p.stride + kh - 1, iw = a_ow[i] * p.stride + kw - 1;
                    ok = ih >= 0 && iw >= 0 && ih < p.Hin * p.up && iw < p.Win * p.up;
                    pix = ((long long)a_b[i] * p.Hin + (ih / p.up)) * p.Win + (iw / p.up);
                }
                if (ok) v = *reinterpret_cast<const u32x4*>(src + pix * Cs + cc + chunk * 8);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + rbase + 32 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n < p.N) v = *reinterpret_cast<const u32x4*>(p.w + (long long)n * p.K + k0 + chunk * 8);
            rb[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
        char* A = smem + buf * BUF;
        char* B = A + A_BYTES;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int r = rbase + 32 * i;
            *reinterpret_cast<u32x4*>(A + r * 128 + ((chunk ^ (r & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int r = rbase + 32 * i;
            *reinterpret_cast<u32x4*>(B + r * 128 + ((chunk ^ (r & 7)) << 4)) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    load_regs(0);
    store_lds(0);
    __syncthreads();
    const int l31 = lane & 31, hi = lane >> 5;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) load_regs((ks + 1) * BK);
        const char* A = smem + (ks & 1) * BUF;
        const char* B = A + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t fa[2], fb[2];
            const int ch = s * 2 + hi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + l31;
                fa[i] = *reinterpret_cast<const bf16x8_t*>(A + r * 128 + ((ch ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn * 64 + j * 32 + l31;
                fb[j] = *reinterpret_cast<const bf16x8_t*>(B + r * 128 + ((ch ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) store_lds((ks + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------------------------
    if (p.act == 2) {
        // GEGLU: fragment j=0 holds x, j=1 holds gate for output column (n0 + wn*64)/2 + l31
        const int oc = (n0 + wn * 64) / 2 + l31;
        const bool cok = (n0 + wn * 64 + 32 + l31) < p.N;
        const float bx = (p.bias && cok) ? p.bias[n0 + wn * 64 + l31] : 0.f;
        const float bg = (p.bias && cok) ? p.bias[n0 + wn * 64 + 32 + l31] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < p.M && cok) {
                    const float xv = acc[i][0][r] + bx, gv = acc[i][1][r] + bg;
                    p.out[m * p.ldo + oc] = f32_to_bf16(xv * gelu_erf(gv));
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        const bool cok = n < p.N;
        const float bv = (p.bias && cok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < p.M && cok) {
                    float v = acc[i][j][r] + bv;
                    if (p.rowvec) v += p.rowvec[(m / p.rows_per_sample) * p.rv_stride + n];
                    if (p.act == 1) v = silu_f(v);
                    if (p.tap && n < p.tap_cols) p.tap[m * p.tap_ld + n] = (f16)v;
                    if (p.tap2 && n >= p.tap_cols && n < 2 * p.tap_cols) p.tap2[m * p.tap_ld + (n - p.tap_cols)] = (f16)v;
                    if (p.residual) v += bf16_to_f32(p.residual[m * p.ldr + n]);
                    if (p.out) p.out[m * p.ldo + n] = f32_to_bf16(v);
                    if (p.out_f32) p.out_f32[m * p.ldo + n] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Direct convolution for the two tiny-channel convs (input conv Cin=4/8, output conv Cout=4).
// x: NHWC fp32 or bf16 source, w: [Cout][3][3][Cin] fp32.  One thread per (pixel, cout).
// ---------------------------------------------------------------------------------------------
template <typename TI>
__device__ __forceinline__ float ld_in(const TI* p, long long i);
template <>
__device__ __forceinline__ float ld_in<float>(const float* p, long long i) { return p[i]; }
template <>
__device__ __forceinline__ float ld_in<bf16_t>(const bf16_t* p, long long i) { return bf16_to_f32(p[i]); }

template <typename TI>
__global__ void k_conv3x3_direct(const TI* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int B, int H,
                                 int W, int Cin, int Cout, bf16_t* __restrict__ out_bf16, float* __restrict__ out_nchw_f32) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * Cout;
    if (idx >= total) return;
    const int co = (int)(idx % Cout);
    const long long pix = idx / Cout;
    const int ow = (int)(pix % W), oh = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    float acc = bias ? bias[co] : 0.f;
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh + kh - 1;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow + kw - 1;
            if (iw < 0 || iw >= W) continue;
            const long long base = (((long long)b * H + ih) * W + iw) * Cin;
            const float* wp = w + ((co * 3 + kh) * 3 + kw) * Cin;
            for (int c = 0; c < Cin; ++c) acc = fmaf(ld_in<TI>(x, base + c), wp[c], acc);
        }
    }
    if (out_bf16) out_bf16[pix * Cout + co] = f32_to_bf16(acc);
    if (out_nchw_f32) out_nchw_f32[(((long long)b * Cout + co) * H + oh) * W + ow] = acc;
}

// ---- live HIP-event timing of this kernel family (bench.py roofline) -------------------------------------
// When enabled, every k_gemm_conv launch is bracketed by two events recorded on the launch stream; collect()
// synchronises, sums the elapsed times and the algorithmic FLOPs (2*M*N*K per launch).
#include <vector>
struct GemmProf {
    bool on = false;
    std::vector<hipEvent_t> ev;      // pairs
    size_t used = 0;
    double flops = 0.0;
    long long launches = 0;
};
static GemmProf g_prof;

static inline hipEvent_t prof_event() {
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_prof.ev.push_back(e);
    }
    return g_prof.ev[g_prof.used++];
}

extern "C" {

int vidseg_gemm_profile_begin(void) {
    g_prof.on = true;
    g_prof.used = 0;
    g_prof.flops = 0.0;
    g_prof.launches = 0;
    return VS_OK;
}

// out[0] = total kernel milliseconds, out[1] = algorithmic FLOPs, out[2] = launches
int vidseg_gemm_profile_end(double* out) {
    g_prof.on = false;
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(g_prof.ev[i + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
        if (e != hipSuccess) VS_FAIL(VS_ERR_HIP, "gemm_profile_end: %s", hipGetErrorString(e));
        ms += t;
    }
    out[0] = ms;
    out[1] = g_prof.flops;
    out[2] = (double)g_prof.launches;
    return VS_OK;
}

static int launch_gemm(const GemmParams& p, hipStream_t st) {
    VS_REQUIRE(p.K % BK == 0, "gemm: K=%d must be a multiple of %d", p.K, BK);
    VS_REQUIRE(p.C0 % BK == 0 && p.C1 % BK == 0, "gemm: source channels (%d,%d) must be multiples of %d", p.C0, p.C1, BK);
    VS_REQUIRE(p.M > 0 && p.N > 0, "gemm: empty problem");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_gemm_conv<128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)k_gemm_conv<256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        attr = true;
    }
    const bool narrow = (p.N % 128 != 0) && (p.N % 128 <= 64) && p.act != 2 && p.M >= 256;
    if (g_prof.on) {
        (void)hipEventRecord(prof_event(), st);
        g_prof.flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
        g_prof.launches++;
    }
    if (narrow) {
        const long long tiles = ((p.M + 255) / 256) * ((p.N + 63) / 64);
        k_gemm_conv<256, 64><<<dim3((unsigned)tiles), 256, 2 * (256 + 64) * BK * 2, st>>>(p);
    } else {
        const long long tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
        k_gemm_conv<128, 128><<<dim3((unsigned)tiles), 256, 2 * (128 + 128) * BK * 2, st>>>(p);
    }
    if (g_prof.on) (void)hipEventRecord(prof_event(), st);
    VS_CHECK_LAUNCH("gemm_conv");
    return VS_OK;
}

// out[M][N] = A[M][K] @ W[N][K]^T (+bias)(+rowvec)(act)(+residual); A may be the channel concat of two [M][C] tensors.
int vidseg_linear_bf16(const void* a0, const void* a1, int C0, int C1, long long M, const void* w, int N, const float* bias,
                       const float* rowvec, int rv_stride, int rows_per_sample, const void* residual, int ldr, void* out,
                       float* out_f32, int ldo, void* tap, void* tap2, int tap_cols, int tap_ld, int act, hipStream_t st) {
    GemmParams p{};
    p.x0 = (const bf16_t*)a0;
    p.x1 = (const bf16_t*)a1;
    p.C0 = C0;
    p.C1 = a1 ? C1 : 0;
    p.ksize = 1;
    p.stride = 1;
    p.up = 1;
    p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.w = (const bf16_t*)w;
    p.N = N;
    p.K = p.C0 + p.C1;
    p.M = M;
    p.bias = bias;
    p.rowvec = rowvec;
    p.rv_stride = rv_stride;
    p.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
    p.residual = (const bf16_t*)residual;
    p.ldr = ldr;
    p.out = (bf16_t*)out;
    p.out_f32 = out_f32;
    p.ldo = ldo;
    p.tap = (f16*)tap;
    p.tap2 = (f16*)tap2;
    p.tap_cols = tap_cols;
    p.tap_ld = tap_ld;
    p.act = act;
    if (act == 2) VS_REQUIRE(N % 64 == 0 && out, "linear: GEGLU needs N %% 64 == 0 and a bf16 output");
    return launch_gemm(p, st);
}

// 3x3 convolution, padding 1, NHWC bf16 activations, weight packed [Cout][(kh*3+kw)*Cin + c].
int vidseg_conv3x3_bf16(const void* x0, const void* x1, int C0, int C1, int B, int Hin, int Win, int stride, int up, const void* w,
                        int Cout, const float* bias, const float* rowvec, int rv_stride, const void* residual, void* out,
                        hipStream_t st) {
    VS_REQUIRE((stride == 1 || stride == 2) && (up == 1 || up == 2), "conv3x3: stride=%d up=%d", stride, up);
    GemmParams p{};
    p.x0 = (const bf16_t*)x0;
    p.x1 = (const bf16_t*)x1;
    p.C0 = C0;
    p.C1 = x1 ? C1 : 0;
    p.Hin = Hin;
    p.Win = Win;
    p.ksize = 3;
    p.stride = stride;
    p.up = up;
    p.Hout = (Hin * up + 2 - 3) / stride + 1;
    p.Wout = (Win * up + 2 - 3) / stride + 1;
    p.w = (const bf16_t*)w;
    p.N = Cout;
    p.K = 9 * (p.C0 + p.C1);
    p.M = (long long)B * p.Hout * p.Wout;
    p.bias = bias;
    p.rowvec = rowvec;
    p.rv_stride = rv_stride;
    p.rows_per_sample = p.Hout * p.Wout;
    p.residual = (const bf16_t*)residual;
    p.ldr = Cout;
    p.out = (bf16_t*)out;
    p.ldo = Cout;
    return launch_gemm(p, st);
}

// Tiny-channel 3x3 convs.  in_is_f32: x is NHWC fp32 (sampler latent) else NHWC bf16.
int vidseg_conv3x3_direct(const void* x, int in_is_f32, const float* w, const float* bias, int B, int H, int W, int Cin, int Cout,
                          void* out_bf16_nhwc, float* out_f32_nchw, hipStream_t st) {
    const long long total = (long long)B * H * W * Cout;
    if (total == 0) return VS_OK;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (in_is_f32)
        k_conv3x3_direct<float><<<dim3(blocks), 256, 0, st>>>((const float*)x, w, bias, B, H, W, Cin, Cout, (bf16_t*)out_bf16_nhwc,
                                                             out_f32_nchw);
    else
        k_conv3x3_direct<bf16_t><<<dim3(blocks), 256, 0, st>>>((const bf16_t*)x, w, bias, B, H, W, Cin, Cout, (bf16_t*)out_bf16_nhwc,
                                                              out_f32_nchw);
    VS_CHECK_LAUNCH("conv3x3_direct");
    return VS_OK;
}

}  // extern "C"
