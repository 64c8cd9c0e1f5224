// HIPCC_FLAGS: -fno-slp-vectorize
// The arithmetic of the conditioner's OpenCLIP ViT-H towers that the UNet path's kernels do not already cover (SURVEY.md §8(f) rank 4;
// sgm/modules/encoders/modules.py:498-567 text, :570-728 image, :1028-1046 SVD's image prediction embedder).  The towers run ONCE per
// clip (SD: the empty prompt, a constant; SVD: one conditioning frame), on 77 / 257 tokens: every kernel here is latency-bound fp32
// work on a few hundred KB, written for clarity and a fixed summation order, not for a roofline.  The projections of the towers go
// through the exact mode's split-operand GEMM (vidseg_linear_a16_rf32), so the embeddings are fp32-accurate like the reference's.
//
//   vidseg_clip_attention_f32      softmax(q k^T * scale [+ causal mask]) v, any head width <= 128 (ViT-H image tower: 80), fp32
//   vidseg_clip_layernorm_f32      torch.nn.LayerNorm, fp32 in / out (ln_pre, ln_post, ln_final)
//   vidseg_clip_gelu_split3        erf GELU of the MLP's hidden layer written as the next GEMM's operand image
//   vidseg_clip_blur_axis          one pass of kornia's separable gaussian_blur2d (reflect border) -- the antialias of kornia.geometry.resize
//   vidseg_clip_resize_patches     bicubic (align_corners) resize to S x S, (x + 1) / 2, CLIP mean / std, laid out as the patch
//                                  matrix of the 14 x 14 stride-14 convolution (rows = patches, K = 3 * 14 * 14 padded to a GEMM width)
#include "common.h"

#define CA_MAXD 128
#define CA_MAXK 1024
// One wave per query row; lanes over keys for q k^T (K rows through L2: a few hundred KB in all), lanes over channels for p v.
__global__ void __launch_bounds__(256) k_clip_attention(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                        const float* __restrict__ v, int ldv, float* __restrict__ out, int ldo, int Nq, int Nk,
                                                        int d, float scale, int causal) {
    __shared__ float Qs[4][CA_MAXD];
    __shared__ float Ps[4][CA_MAXK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.y, b = blockIdx.z;
    const int i = blockIdx.x * 4 + wave;
    if (i >= Nq) return;                                        // no block-wide barrier below: waves are independent
    const float* qr = q + ((long long)b * Nq + i) * ldq + h * d;
    for (int c = lane; c < d; c += 64) Qs[wave][c] = qr[c] * scale;
    __builtin_amdgcn_wave_barrier();
    const int nk = causal ? min(Nk, i + 1) : Nk;                 // keys 0 .. i of a causal row (open_clip's build_attention_mask: -inf above the diagonal)
    float m = -INFINITY;
    for (int j = lane; j < nk; j += 64) {
        const float* kr = k + ((long long)b * Nk + j) * ldk + h * d;
        float s = 0.f;
        for (int c = 0; c < d; c += 4) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + c);
            s = fmaf(Qs[wave][c], kv[0], s);
            s = fmaf(Qs[wave][c + 1], kv[1], s);
            s = fmaf(Qs[wave][c + 2], kv[2], s);
            s = fmaf(Qs[wave][c + 3], kv[3], s);
        }
        Ps[wave][j] = s;
        m = fmaxf(m, s);
    }
    m = wave_max_f32(m);
    float l = 0.f;
    for (int j = lane; j < nk; j += 64) {
        const float p = expf(Ps[wave][j] - m);
        Ps[wave][j] = p;
        l += p;
    }
    l = wave_sum_f32(l);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / l;
    for (int c = lane; c < d; c += 64) {
        const float* vc = v + (long long)b * Nk * ldv + h * d + c;
        float o = 0.f;
        for (int j = 0; j < nk; ++j) o = fmaf(Ps[wave][j], vc[(long long)j * ldv], o);
        out[((long long)b * Nq + i) * ldo + h * d + c] = o * inv;
    }
}

// LayerNorm of fp32 rows (C <= 2048, C % 4 == 0), one wave per row, the arithmetic of k_x_layernorm_split3 with an fp32 result
__global__ void __launch_bounds__(256) k_clip_layernorm(const float* __restrict__ x, long long M, int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    constexpr int MAXCH = 8;
    f32x4 v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        const int c = lane * 4 + ch * 256;
        if (c < C) {
            v[ch] = *reinterpret_cast<const f32x4*>(x + row * C + c);
            s += (v[ch][0] + v[ch][1]) + (v[ch][2] + v[ch][3]);
        }
    }
    const float mean = wave_sum_f32(s) / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        const int c = lane * 4 + ch * 256;
        if (c < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dlt = v[ch][j] - mean;
                qq = fmaf(dlt, dlt, qq);
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_f32(qq) / (float)C + eps);
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
        const int c = lane * 4 + ch * 256;
        if (c < C) {
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf((v[ch][j] - mean) * rstd, ga[j], be[j]);
            *reinterpret_cast<f32x4*>(out + row * C + c) = o;
        }
    }
}

// One axis of kornia.filters.gaussian_blur2d(separable=True, border_type="reflect") on [planes][H][W] fp32: taps `w` (ks, odd) along
// x (axis 1) or y (axis 0); reflect = index -1 -> 1, n -> n - 2 (torch.nn.functional.pad mode "reflect").
__global__ void __launch_bounds__(256) k_clip_blur_axis(const float* __restrict__ x, long long planes, int H, int W, const float* __restrict__ w,
                                                        int ks, int axis, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * H * W) return;
    const int xx = (int)(i % W), yy = (int)((i / W) % H);
    const long long p = i / ((long long)W * H);
    const int n = axis ? W : H, pos = axis ? xx : yy, r = ks / 2;
    const float* base = x + p * H * W;
    float s = 0.f;
    for (int t = 0; t < ks; ++t) {
        int j = pos + t - r;
        if (j < 0) j = -j;
        if (j >= n) j = 2 * n - 2 - j;
        s = fmaf(w[t], axis ? base[(long long)yy * W + j] : base[(long long)j * W + xx], s);
    }
    out[i] = s;
}

// torch.nn.functional.interpolate(mode="bicubic", align_corners=True) to S x S (A = -0.75 cubic convolution, border indices clamped),
// then (x + 1) / 2 and kornia.enhance.normalize(mean, std) (modules.py:621-633), written as rows of the patch matrix:
// out[(b * G + py) * G + px][c * P * P + ky * P + kx], G = S / P patches a side, row stride ldo (columns >= 3 P P stay untouched: the
// caller zero-fills the pad once).
__device__ __forceinline__ void cubic_weights(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// ATen's area_pixel_compute_source_index -> floorf -> subtraction in fp32, WITHOUT contracting the product into the subtraction: the
// fraction has to belong to the same rounded coordinate whose floor is taken.  (hipcc contracts by default, and __fmul_rn is a plain
// product to it; the fused form moves the taps by up to 3e-5 of a pixel at x ~ 1000: 1e-4 on a noise image.  torch's CPU kernels --
// the oracle -- do not fuse; a CUDA build of the same ATen source may.)
__device__ __forceinline__ void source_coord(float scale, int o, int& i0, float& t) {
#pragma clang fp contract(off)
    const float r = scale * (float)o;
    const float fl = floorf(r);
    i0 = (int)fl;
    t = r - fl;
}

__global__ void __launch_bounds__(256) k_clip_resize_patches(const float* __restrict__ x, int B, int H, int W, int S, int P, float m0, float m1,
                                                             float m2, float s0, float s1, float s2, float* __restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * 3 * S * S) return;
    const int ox = (int)(i % S), oy = (int)((i / S) % S), c = (int)((i / ((long long)S * S)) % 3), b = (int)(i / ((long long)3 * S * S));
    const float sy = S > 1 ? (float)(H - 1) / (float)(S - 1) : 0.f, sx = S > 1 ? (float)(W - 1) / (float)(S - 1) : 0.f;
    int iy, ix;
    float ty, tx;
    source_coord(sy, oy, iy, ty);
    source_coord(sx, ox, ix, tx);
    float wy[4], wx[4];
    cubic_weights(ty, wy);
    cubic_weights(tx, wx);
    const float* base = x + ((long long)b * 3 + c) * H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), H - 1);
        float row = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) row = fmaf(wx[e], base[(long long)yy * W + min(max(ix - 1 + e, 0), W - 1)], row);
        acc = fmaf(wy[a], row, acc);
    }
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const float val = ((acc + 1.0f) / 2.0f - mean) / sd;
    const int G = S / P, py = oy / P, ky = oy % P, px = ox / P, kx = ox % P;
    out[(((long long)b * G + py) * G + px) * ldo + (c * P + ky) * P + kx] = val;
}

#if VIDSEG_ACT_IS_F16
// split3(GELU(x)), erf form (open_clip's nn.GELU; the same 0.5 g (1 + erf(g / sqrt 2)) as the GEGLU epilogue)
__global__ void __launch_bounds__(256) k_clip_gelu_split3(const float* __restrict__ x, long long M, int C, f16* __restrict__ out) {
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4n = C / 4;
    if (i4 >= M * c4n) return;
    const long long m = i4 / c4n;
    const int c = (int)(i4 - m * c4n) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * C + c);
    f16x4 h, l;
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = 0.5f * v[j] * (1.0f + erf_f32(v[j] * 0.70710678118654752440f));
    split_hl4(f, h, l);
    f16* o = out + m * 3 * C + c;
    *reinterpret_cast<f16x4*>(o) = h;
    *reinterpret_cast<f16x4*>(o + C) = l;
    if (VS_THIRD_PLANE(C)) *reinterpret_cast<f16x4*>(o + 2 * C) = h;
}
#endif

extern "C" {

int vidseg_clip_attention_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B, int H,
                              int Nq, int Nk, int d, float scale, int causal, hipStream_t st) {
    VS_REQUIRE(d > 0 && d <= CA_MAXD && d % 4 == 0 && Nk > 0 && Nk <= CA_MAXK, "clip_attention: d=%d (<= %d, %% 4) Nk=%d (1..%d)", d, CA_MAXD, Nk,
               CA_MAXK);
    VS_REQUIRE(ldk % 4 == 0 && !(causal && Nq != Nk), "clip_attention: ldk=%d causal=%d Nq=%d Nk=%d", ldk, causal, Nq, Nk);
    if ((long long)B * H * Nq == 0) return VS_OK;
    k_clip_attention<<<dim3((unsigned)((Nq + 3) / 4), H, B), 256, 0, st>>>(q, ldq, k, ldk, v, ldv, out, ldo, Nq, Nk, d, scale, causal);
    VS_CHECK_LAUNCH("clip_attention_f32");
    return VS_OK;
}

int vidseg_clip_layernorm_f32(const float* x, long long M, int C, const float* gamma, const float* beta, float eps, float* out, hipStream_t st) {
    VS_REQUIRE(C % 4 == 0 && C > 0 && C <= 2048, "clip_layernorm: C=%d", C);
    if (M == 0) return VS_OK;
    k_clip_layernorm<<<dim3((unsigned)((M + 3) / 4)), 256, 0, st>>>(x, M, C, gamma, beta, eps, out);
    VS_CHECK_LAUNCH("clip_layernorm_f32");
    return VS_OK;
}

int vidseg_clip_blur_axis(const float* x, long long planes, int H, int W, const float* taps, int ks, int axis, float* out, hipStream_t st) {
    VS_REQUIRE(ks >= 1 && ks % 2 == 1 && (axis == 0 || axis == 1) && ks / 2 < (axis ? W : H), "clip_blur_axis: ks=%d axis=%d H=%d W=%d", ks, axis, H,
               W);
    VS_REQUIRE(x != out, "clip_blur_axis: in place is not supported");
    const long long n = planes * H * W;
    if (n == 0) return VS_OK;
    k_clip_blur_axis<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(x, planes, H, W, taps, ks, axis, out);
    VS_CHECK_LAUNCH("clip_blur_axis");
    return VS_OK;
}

int vidseg_clip_resize_patches(const float* x, int B, int H, int W, int S, int P, const float* mean3_host, const float* std3_host, float* out,
                               int ldo, hipStream_t st) {
    VS_REQUIRE(S > 0 && P > 0 && S % P == 0 && ldo >= 3 * P * P && H > 0 && W > 0, "clip_resize_patches: S=%d P=%d ldo=%d H=%d W=%d", S, P, ldo, H,
               W);
    const long long n = (long long)B * 3 * S * S;
    if (n == 0) return VS_OK;
    k_clip_resize_patches<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(x, B, H, W, S, P, mean3_host[0], mean3_host[1], mean3_host[2],
                                                                           std3_host[0], std3_host[1], std3_host[2], out, ldo);
    VS_CHECK_LAUNCH("clip_resize_patches");
    return VS_OK;
}

#if VIDSEG_ACT_IS_F16
int vidseg_clip_gelu_split3(const float* x, long long M, int C, void* out16, hipStream_t st) {
    VS_REQUIRE(C % 4 == 0, "clip_gelu_split3: C=%d", C);
    if (M == 0) return VS_OK;
    k_clip_gelu_split3<<<dim3((unsigned)((M * (C / 4) + 255) / 256)), 256, 0, st>>>(x, M, C, (f16*)out16);
    VS_CHECK_LAUNCH("clip_gelu_split3");
    return VS_OK;
}
#else
int vidseg_clip_gelu_split3(const float*, long long, int, void*, hipStream_t) {
    VS_FAIL(VS_ERR_UNSUPPORTED, "clip_gelu_split3: the exact (split-fp16) mode needs the fp16 build of the library");
}
#endif

}  // extern "C"
